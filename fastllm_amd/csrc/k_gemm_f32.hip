// k_gemm_f32.hip -- the fp32 mode's prompt GEMM on the matrix cores: Y[T,N] = X[T,K] . W[N,K]^T with fp32 operands and fp32 accumulation
// (v_mfma_f32_32x32x2_f32: exact fp32 products, the sum over k an fmaf chain in k order -- bit for bit what gemm_generic_kernel's
// scalar fmaf loop computes, at the matrix pipe's 64 FLOP / clk / SIMD with two operand VGPRs per 2048 FMAs instead of one per FMA).
//
// The fp32 mode is the one the literal parity bar is stated on (logits within 1e-3 of the CPU reference, identical greedy ids:
// tests/test_gpu_literal_configs.py); its 512-token Mistral-7B prefill spent 238 of 243 ms in the 64 x 64 VALU kernel at 30 TFLOP/s.
//
// Tile BM x 128 (BM = 128, or 64 where 128-row tiles would not fill the chip), BK = 16, four waves as 2 x 2, each (BM / 2) x 64 =
// MB x 2 blocks of 32 x 32 (16 accumulator VGPRs per block).  Operands go global -> registers -> LDS, K-major ([k][row], rows padded
// to BM + 4 floats: the transposing stores are conflict-free, a fragment read is 32 consecutive floats per k), double-buffered: the
// next K tile's global loads are issued before the 8 k-steps of MFMAs on the current one and stored to the other buffer behind them;
// one barrier per K tile.  A lane's A / B operand of k-step s is element [2 s + (lane >> 5)][lane & 31] of the tile: natural k order.
// Epilogues as gemm_generic_kernel (row scale, bias; SiLU(gate) * up on the interleaved 16-column groups -- a 32-column block holds a
// gate group in lanes 0-15 and its up group in lanes 16-31: one shuffle).
// Measured (Mistral-7B, 512 tokens): gate/up on 128-row tiles 109 TFLOP/s, the 64-row grids (256-384 workgroups) 82-84 -- neither 32-deep K
// tiles nor eight waves per workgroup moved the latter (80 / 83); below ~128 tokens the grids are a fraction of the chip (no K slices here).
#include <algorithm>

#include "kernels.h"

namespace fl {

typedef float float16v __attribute__((ext_vector_type(16)));

template <int BM>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const float *__restrict__ W, const float *__restrict__ X,
                                                            const float *__restrict__ bias, void *__restrict__ out,
                                                            int T, int N, int K, int epi, const float *__restrict__ row_scale) {
    constexpr int BN = 128, BK = 16, LDA = BM + 4, LDB = BN + 4, MB = BM / 64;
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lr = tid >> 2, kq = tid & 3;                          // loader: row lr (+ 64 h), floats 4 kq .. 4 kq + 3 of the K tile

    float16v acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    float4v ra[MB], rb[2];
    const float *xp[MB], *wp[2];
#pragma unroll
    for (int h = 0; h < MB; h++) xp[h] = X + (size_t)min(m0 + lr + 64 * h, T - 1) * K + 4 * kq;
#pragma unroll
    for (int h = 0; h < 2; h++) wp[h] = W + (size_t)min(n0 + lr + 64 * h, N - 1) * K + 4 * kq;
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < MB; h++) ra[h] = *reinterpret_cast<const float4v *>(xp[h] + k0);
#pragma unroll
        for (int h = 0; h < 2; h++) rb[h] = *reinterpret_cast<const float4v *>(wp[h] + k0);
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
#pragma unroll
            for (int h = 0; h < MB; h++) As[buf][4 * kq + e][lr + 64 * h] = ra[h][e];
#pragma unroll
            for (int h = 0; h < 2; h++) Bs[buf][4 * kq + e][lr + 64 * h] = rb[h][e];
        }
    };
    const int nk = K / BK;
    gload(0);
    sstore(0);
    __syncthreads();
    const int kh = lane >> 5, l32 = lane & 31;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int s = 0; s < BK / 2; s++) {
            float a[MB], b[2];
#pragma unroll
            for (int i = 0; i < MB; i++) a[i] = As[buf][2 * s + kh][wr * (BM / 2) + i * 32 + l32];
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = Bs[buf][2 * s + kh][wn * 64 + j * 32 + l32];
#pragma unroll
            for (int i = 0; i < MB; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    // C/D map of the 32 x 32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < MB; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + wr * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float rs = (row_scale && m < T) ? row_scale[m] : 1.0f;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int n = n0 + wn * 64 + j * 32 + l32;
                float v = acc[i][j][r];
                if (row_scale) v *= rs;
                if (epi == EPI_GATEUP) {
                    const float up = __shfl_down(v, 16);             // lanes 0-15 of a half: the gate column; 16 further: its up column
                    if (m < T && l32 < 16 && n + 16 < N) {
                        const int q = (n >> 5) * 16 + (n & 15);
                        reinterpret_cast<float *>(out)[(size_t)m * (N / 2) + q] = v / (1.0f + expf(-v)) * up;
                    }
                } else if (m < T && n < N) {
                    reinterpret_cast<float *>(out)[(size_t)m * N + n] = v + (bias ? bias[n] : 0.f);
                }
            }
        }
}

// ---- short prompts and decode batches in fp32 (2 ... 64 rows): a weight stream, not matrix work ----------------------------------
// The MFMA kernel above runs N / 128 workgroups below 64 tokens -- 32 of 256 CUs for a 4096-row matrix -- and a 4-stream fp32 decode
// step took 45 ms where one stream's takes 5.9.  Here a workgroup of eight waves owns 16 weight rows and ALL of up to TB token rows:
// the x rows of a 1024-float K chunk are staged in LDS (TB x 4 KB), every wave streams ITS two weight rows as coalesced float4s (1 KB
// per wave per load, eight loads in flight per lane) and every x float4 it reads from LDS serves both rows; fp32 FMAs, one butterfly
// per (row, token) sum at the end.  More than TB tokens: one pass over the weights per block of 16 rows (the host loops).
// gate/up: a workgroup takes 8 gate rows and their 8 up rows (16 further down the interleaved matrix), a wave holds a pair.
template <int TB>
__global__ __launch_bounds__(512) void gemv_f32_rows_kernel(const float *__restrict__ W, const float *__restrict__ X, const float *__restrict__ bias,
                                                            float *__restrict__ out, int T, int t0, int N, int K, int epi,
                                                            const float *__restrict__ row_scale) {
    constexpr int KC = 1024;
    __shared__ float4v xs[TB][KC / 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int rows[2];
    if (epi == EPI_GATEUP) {
        const int g = blockIdx.x >> 1, p = (blockIdx.x & 1) * 8 + wave;
        rows[0] = g * 32 + p; rows[1] = g * 32 + 16 + p;
    } else {
        rows[0] = blockIdx.x * 16 + wave * 2; rows[1] = rows[0] + 1;
    }
    float acc[2][TB];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int t = 0; t < TB; t++) acc[r][t] = 0.f;
    const float4v zero = {0.f, 0.f, 0.f, 0.f};
    const float *wr0 = W + (size_t)min(rows[0], N - 1) * K, *wr1 = W + (size_t)min(rows[1], N - 1) * K;
    // K is whole 256-float pieces (one float4 per lane): every bound below is wave-uniform
    float4v w[2][4];
    const int kl = lane * 4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (j * 256 < K) {
            w[0][j] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(wr0 + j * 256 + kl));
            w[1][j] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(wr1 + j * 256 + kl));
        } else { w[0][j] = zero; w[1][j] = zero; }
    }
    for (int k0 = 0; k0 < K; k0 += KC) {
        __syncthreads();                                      // the previous chunk has been consumed
        const int kc = min(KC, K - k0) / 4;                   // float4s per row in this chunk
#pragma unroll 2
        for (int idx = tid; idx < TB * (KC / 4); idx += 512) {
            const int t = idx / (KC / 4), c = idx % (KC / 4), row = t0 + t;
            if (c < kc) xs[t][c] = row < T ? *reinterpret_cast<const float4v *>(X + (size_t)row * K + k0 + 4 * c) : zero;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (k0 + j * 256 < K) {
                const float4v w0 = w[0][j], w1 = w[1][j];
                if (k0 + KC + j * 256 < K) {                 // this slot's registers are free: the next chunk's piece goes out now
                    w[0][j] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(wr0 + k0 + KC + j * 256 + kl));
                    w[1][j] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(wr1 + k0 + KC + j * 256 + kl));
                }
#pragma unroll
                for (int t = 0; t < TB; t++) {
                    const float4v xv = xs[t][j * 64 + lane];
                    acc[0][t] = fmaf(w0[3], xv[3], fmaf(w0[2], xv[2], fmaf(w0[1], xv[1], fmaf(w0[0], xv[0], acc[0][t]))));
                    acc[1][t] = fmaf(w1[3], xv[3], fmaf(w1[2], xv[2], fmaf(w1[1], xv[1], fmaf(w1[0], xv[0], acc[1][t]))));
                }
            }
        }
    }
    float res[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int t = 0; t < TB; t++) {
            const float v = wave_sum(acc[r][t]);
            if (lane == t) res[r] = v;
        }
    const int m = t0 + lane;
    if (lane < TB && m < T) {
        const float rs = row_scale ? row_scale[m] : 1.0f;
        if (epi == EPI_GATEUP) {
            const int n = rows[0];                            // gate row; its up row is the pair's second
            if (n + 16 < N) {
                const float gt = res[0] * rs, up = res[1] * rs;
                out[(size_t)m * (N / 2) + (n >> 5) * 16 + (n & 15)] = gt / (1.0f + expf(-gt)) * up;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int n = rows[r];
                if (n < N) out[(size_t)m * N + n] = res[r] * rs + (bias ? bias[n] : 0.f);
            }
        }
    }
}

bool gemv_f32_rows_supported(int64_t T, int64_t N, int64_t K, int epi) {
    const int64_t maxt = tune(TK_F32_ROWS_MAX);
    return maxt > 0 && T > 1 && T <= std::min<int64_t>(maxt, 64) && K % 256 == 0 && N >= 1 && (epi == EPI_F32 || (epi == EPI_GATEUP && N % 32 == 0));
}

int launch_gemv_f32_rows(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K, int epi,
                         const float *row_scale) {
    if (!gemv_f32_rows_supported(T, N, K, epi)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemv_f32_rows: 2-64 rows, K a multiple of 256");
    if (epi == EPI_GATEUP && bias) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemv_f32_rows: gate/up has no bias");
    char tag[40];
    snprintf(tag, sizeof tag, "f32rows,%lldx%lld", (long long)N, (long long)K);
    Launcher LL = L; LL.tag = tag;
    const unsigned grid = (unsigned)((N + 15) / 16);
    for (int64_t t0 = 0; t0 < T; t0 += 16) {
        const int64_t tb = std::min<int64_t>(16, T - t0);
        const double bytes = ((double)N * K + (double)tb * K) * 4.0, flops = 2.0 * tb * N * K;
        int rc;
        if (tb <= 4) rc = LL.launch(KC_GEMM_GENERIC, bytes, flops, gemv_f32_rows_kernel<4>, dim3(grid), dim3(512), 0, (const float *)W, (const float *)x, bias, (float *)y, (int)T, (int)t0, (int)N, (int)K, epi, row_scale);
        else if (tb <= 8) rc = LL.launch(KC_GEMM_GENERIC, bytes, flops, gemv_f32_rows_kernel<8>, dim3(grid), dim3(512), 0, (const float *)W, (const float *)x, bias, (float *)y, (int)T, (int)t0, (int)N, (int)K, epi, row_scale);
        else rc = LL.launch(KC_GEMM_GENERIC, bytes, flops, gemv_f32_rows_kernel<16>, dim3(grid), dim3(512), 0, (const float *)W, (const float *)x, bias, (float *)y, (int)T, (int)t0, (int)N, (int)K, epi, row_scale);
        FL_TRY(rc);
    }
    return FL_OK;
}

bool gemm_f32_mfma_supported(int64_t T, int64_t N, int64_t K) {
    return tune(TK_GEMM_F32_MFMA) != 0 && T > 1 && N >= 1 && K % 16 == 0 && K >= 16;
}

int launch_gemm_f32_mfma(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                         int epi, const float *row_scale) {
    if (!gemm_f32_mfma_supported(T, N, K)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_f32_mfma: K must be a multiple of 16");
    if (epi != EPI_F32 && epi != EPI_GATEUP) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_f32_mfma: fp32 or gate/up epilogue");
    if (epi == EPI_GATEUP && (N % 32 || bias)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_f32_mfma: gate/up rows in groups of 32, no bias");
    const int64_t tn = (N + 127) / 128, t128 = ((T + 127) / 128) * tn;
    const double bytes = ((double)N * K + (double)T * K) * 4.0;
    char tag[40];
    snprintf(tag, sizeof tag, "f32mfma,%lldx%lld", (long long)N, (long long)K);
    Launcher LL = L; LL.tag = tag;
    // 128-row tiles once they give every CU two workgroups (two waves per SIMD hide the global loads), else 64-row tiles
    if ((T + 63) / 64 > 65535) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_f32_mfma: too many row tiles");
    if (t128 >= 512)
        return LL.launch(KC_GEMM_GENERIC, bytes, 2.0 * T * N * K, gemm_f32_mfma_kernel<128>, dim3((unsigned)tn, (unsigned)((T + 127) / 128)), dim3(256), 0,
                         (const float *)W, (const float *)x, bias, y, (int)T, (int)N, (int)K, epi, row_scale);
    return LL.launch(KC_GEMM_GENERIC, bytes, 2.0 * T * N * K, gemm_f32_mfma_kernel<64>, dim3((unsigned)tn, (unsigned)((T + 63) / 64)), dim3(256), 0,
                     (const float *)W, (const float *)x, bias, y, (int)T, (int)N, (int)K, epi, row_scale);
}

}  // namespace fl
