// k_linear.hip -- projection kernels: y = x . W^T (+bias) for the QKV / O / gate-up / down /
// lm_head matmuls of the decoder (candle Linear::forward).
//
//   gemv_kernel      T == 1 (decode): the HBM-bound weight stream (k_gemv.hip).
//   gemm_generic     any T, any shape: 64x64 LDS-tiled fp32-FMA kernel (fp32 parity mode and odd
//                    shapes).
//   gemm_mfma        T > 1, bf16: MFMA-tiled kernel (k_gemm_mfma.hip).
#include <stdlib.h>

#include <atomic>

#include <algorithm>

#include "kernels.h"

namespace fl {


// =============================================================================== generic GEMM
// 64x64 output tile, 16-deep K slices through LDS (as fp32), 256 threads, 4x4 per thread with
// column stride 16 so that a thread owns gate column c and up column c+16 of the interleaved
// gate/up layout.  Any T, N, K.
template <typename WT, typename XT>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const WT *__restrict__ W, const XT *__restrict__ X,
                                                           const float *__restrict__ bias, void *__restrict__ out,
                                                           int T, int N, int K, int epi,
                                                           const float *__restrict__ row_scale) {
    __shared__ float xs[16][64 + 1];
    __shared__ float ws[16][64 + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            int r = i >> 4, kk = i & 15;
            int m = m0 + r, n = n0 + r, k = k0 + kk;
            xs[kk][r] = (m < T && k < K) ? elem<XT>::ld(X + (size_t)m * K + k) : 0.f;
            ws[kk][r] = (n < N && k < K) ? elem<WT>::ld(W + (size_t)n * K + k) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { a[i] = xs[kk][ty * 4 + i]; b[i] = ws[kk][tx + 16 * i]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int m = m0 + ty * 4 + i;
        if (m >= T) continue;
        if (row_scale) {
            const float rs = row_scale[m];
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] *= rs;
        }
        if (epi == EPI_GATEUP) {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                int n = n0 + tx + 16 * j;                 // gate column; up is n + 16
                if (n + 16 < N) {
                    int q = (n >> 5) * 16 + (n & 15);
                    float gt = acc[i][j], up = acc[i][j + 1];
                    float a = gt / (1.0f + expf(-gt)) * up;
                    elem<XT>::st(reinterpret_cast<XT *>(out) + (size_t)m * (N / 2) + q, a);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int n = n0 + tx + 16 * j;
                if (n < N) reinterpret_cast<float *>(out)[(size_t)m * N + n] = acc[i][j] + (bias ? bias[n] : 0.f);
            }
        }
    }
}

template <typename WT, typename XT>
static int launch_gemm_generic(Launcher &L, const void *W, const void *x, const float *bias, void *y,
                               int64_t T, int64_t N, int64_t K, int epi, const float *row_scale) {
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((T + 63) / 64));
    double bytes = (double)N * K * sizeof(WT) + (double)T * K * sizeof(XT);
    return L.launch(KC_GEMM_GENERIC, bytes, 2.0 * T * N * K, gemm_generic_kernel<WT, XT>, grid, dim3(256), 0,
                    (const WT *)W, (const XT *)x, bias, y, (int)T, (int)N, (int)K, epi, row_scale);
}

int launch_gemm_mfma(Launcher &L, const void *W, const void *x, const float *bias, void *y,
                     int64_t T, int64_t N, int64_t K, int epi, const float *row_scale, int ksplit);   // k_gemm_mfma.hip
int gemm_mfma_ksplit(int64_t T, int64_t N, int64_t K, int epi, int max_split);

static GemvArgs plain_args(const void *W, const void *x, const float *bias, void *y, int64_t N, int64_t K, int epi,
                           const float *scale) {
    GemvArgs a; a.W = W; a.x = x; a.bias = bias; a.out = y; a.N = (int)N; a.K = (int)K; a.epi = epi; a.pro = PRO_X;
    a.x_scale = scale;
    return a;
}

int launch_linear(Launcher &L, int dtype, const void *W, const void *x, const float *bias, void *y,
                  int64_t T, int64_t N, int64_t K, int epi, const float *row_scale, int max_split, int *n_split_out) {
    if (n_split_out) *n_split_out = 1;
    if (T <= 0 || N <= 0 || K <= 0) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_linear: bad shape");
    if (epi == EPI_GATEUP && N % 32) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gate/up matrix rows must be a multiple of 32");
    const int force_generic = tune(TK_FORCE_GENERIC_GEMM);
    // Launcher::rsp (row scales as partial sums): only the kernels gemm_takes_rs_parts() names read it -- any other path would
    // silently use a stale vector, so the sums are finished into the vector first (kernels.h, rs_parts_to_vector)
    auto no_parts = [&]() -> int { return rs_parts_to_vector(L, row_scale, T); };
    if (dtype == FL_DTYPE_BF16) {
        if (T == 1 && gemv_supported(dtype, N, K)) { FL_TRY(no_parts()); return launch_gemv(L, dtype, plain_args(W, x, bias, y, N, K, epi, row_scale)); }
        // mid-size prompts: 128 x 256 tiles, K slices summed inside the launch (k_gemm_h4.hip) -- one complete output, no slabs
        if (!force_generic && T > 1) {
            const int ks = gemm_h4_plan(T, N, K, epi);
            if (ks > 0) return launch_gemm_h4(L, W, x, bias, y, T, N, K, epi, row_scale, ks);
            // a tensor-parallel rank's projections (no slabs: the all-reduce wants the sum): K slices that meet inside the launch
            if (L.tp > 1 && max_split <= 1) {
                const int kw = gemm_h4_plan_whole(T, N, K, epi);
                if (kw > 0) return launch_gemm_h4(L, W, x, bias, y, T, N, K, epi, row_scale, kw);
            }
            // a few tokens past an even number of 256-row tiles: 3 x 128 tiles are a round and a half of the chip and cost two (Mistral-7B gate/up
            // 512 / 513 tokens: 92.6 / 170.3 us).  The even part keeps its whole rounds and the last rows go as a launch of their own on
            // whatever serves that many rows (ring kernel 36-46 us up to 32 rows, short-prompt GEMM 47-56 us up to 128).  Whole Mistral-7B
            // prefills, split / one launch: 513 tokens 10.25 / 10.87 ms, 545 10.76 / 11.01, 600 11.26 / 11.37, 1025 16.49 / 16.83, 1100
            // 16.91 / 17.08 -- and 700 12.48 / 12.29, 768 12.68 / 12.50, 1280 18.46 / 18.17: the second weight pass stops paying near 100 rows
            if (epi == EPI_GATEUP && T > 512 && tune(TK_GATEUP_ROWSPLIT) && !bias) {
                const int64_t tm = (T + 255) / 256, T0 = (tm - 1) * 256;
                if ((tm & 1) && T - T0 <= 96 && gemm_w14_plan(T0, N, K, epi)) {
                    FL_TRY(launch_gemm_w14(L, W, x, bias, y, T0, N, K, epi, row_scale));
                    Launcher L2 = L;
                    if (L2.rsp.part) L2.rsp.part += (size_t)T0 * L2.rsp.np;
                    if (T - T0 <= 32) FL_TRY(rs_parts_to_vector(L2, row_scale ? row_scale + T0 : nullptr, T - T0));     // (the ring kernel reads the vector)
                    return launch_linear(L2, dtype, W, (const char *)x + (size_t)T0 * K * 2, bias, (char *)y + (size_t)T0 * (N / 2) * 2, T - T0, N, K, epi,
                                         row_scale ? row_scale + T0 : nullptr, max_split, nullptr);
                }
            }
            // 224-column tiles where they fill the chip and 256-column ones do not (k_gemm_w14.hip)
            if (gemm_w14_plan(T, N, K, epi)) return launch_gemm_w14(L, W, x, bias, y, T, N, K, epi, row_scale);
        }
        // short prompts / decode batches on the kernel whose K slices meet inside the launch (k_gemm_skf.hip): a tensor-parallel rank's
        // complete outputs (no slabs for its all-reduce) up to 64 rows -- tp = 4, 32 rows: down_proj 22.6 -> 11.6 us, 128 rows: o_proj
        // 11.1 -> 18.1 (the last arriver's tail grows with the tile) --; gate/up of the opt-in five-launch layer; every shape when forced
        if (!force_generic && T > 1 && T <= 128) {
            const int skf = tune(TK_GEMM_SKF);
            const bool whole = epi == EPI_F32 && ((skf >= 1 && L.tp > 1 && max_split <= 1 && T <= 64) || skf >= 3);
            if ((epi == EPI_GATEUP && skf >= 2) || whole) {
                const int ks = gemm_skf_plan(T, N, K, epi);
                if (ks > 0 && (ks > 1 || epi == EPI_GATEUP || skf >= 3)) return launch_gemm_skf(L, W, x, bias, y, T, N, K, epi, row_scale, ks);
            }
        }
        // prompts of 2-32 tokens: the wide gate/up stream on the LDS-DMA ring kernel of the decode batches (5.8 / 5.3 TB/s at <= 16 / 32 rows against 4.9)
        if (!force_generic && T > 1 && T <= 32 && epi == EPI_GATEUP && N >= 8192 && !L.rsp.part && tune(TK_PREFILL_DMA) &&
            gemv_dma_supported((int)T, N, K, epi, 0) && gemv_dma_ksplit(K, 0, epi) == 1) {
            GemvBatchArgs ga;
            ga.W = W; ga.x = x; ga.x_scale = row_scale; ga.out = y; ga.N = (int)N; ga.K = (int)K; ga.epi = epi; ga.pro = PRO_X; ga.B = (int)T; ga.nks = 1;
            return launch_gemv_dma(L, ga);
        }
        const int use_skinny = tune(TK_GEMM_SKINNY);
        if (!force_generic && use_skinny && gemm_skinny_supported(T, N, K)) {     // short prompts: a weight stream
            FL_TRY(no_parts());
            const int ks = (n_split_out && !bias) ? gemm_skinny_ksplit(T, N, K, epi, std::min(max_split, 4)) : 1;   // (more slabs cost the summing launch more than they save here)
            if (n_split_out) *n_split_out = ks;
            return launch_gemm_skinny(L, W, x, bias, y, T, N, K, epi, row_scale, ks);
        }
        if (!force_generic && gemm_mfma_supported(dtype, T, N, K)) {
            const int ks = (n_split_out && !bias) ? gemm_mfma_ksplit(T, N, K, epi, max_split) : 1;
            if (n_split_out) *n_split_out = ks;
            return launch_gemm_mfma(L, W, x, bias, y, T, N, K, epi, row_scale, ks);
        }
        FL_TRY(no_parts());
        return launch_gemm_generic<bf16_t, bf16_t>(L, W, x, bias, y, T, N, K, epi, row_scale);
    }
    FL_TRY(no_parts());
    if (dtype == FL_DTYPE_F32) {
        if (T == 1 && gemv_supported(dtype, N, K)) return launch_gemv(L, dtype, plain_args(W, x, bias, y, N, K, epi, row_scale));
        if (gemv_f32_rows_supported(T, N, K, epi) && !tune(TK_FORCE_GENERIC_GEMM)) return launch_gemv_f32_rows(L, W, x, bias, y, T, N, K, epi, row_scale);
        if (gemm_f32_mfma_supported(T, N, K) && !tune(TK_FORCE_GENERIC_GEMM)) return launch_gemm_f32_mfma(L, W, x, bias, y, T, N, K, epi, row_scale);
        return launch_gemm_generic<float, float>(L, W, x, bias, y, T, N, K, epi, row_scale);
    }
    FL_FAIL(FL_ERR_UNSUPPORTED, "launch_linear: unsupported dtype %d", dtype);
}

}  // namespace fl
