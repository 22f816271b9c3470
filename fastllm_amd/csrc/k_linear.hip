// k_linear.hip -- projection kernels: y = x . W^T (+bias) for the QKV / O / gate-up / down /
// lm_head matmuls of the decoder (candle Linear::forward).
//
//   gemv_kernel      T == 1 (decode).  HBM-bound weight stream: every weight byte is read once,
//                    16 B per lane, non-temporal, straight to VGPRs (no LDS round trip for the
//                    streamed operand); x is staged once per workgroup in LDS; fp32 accumulate;
//                    wave-reduce; fused bias / SiLU-gate epilogue.
//   gemm_generic     any T, any shape: 64x64 LDS-tiled fp32-FMA kernel (fp32 parity mode and odd
//                    shapes).
//   gemm_mfma        T > 1, bf16: MFMA-tiled kernel (k_gemm_mfma.hip).
#include <stdlib.h>

#include <atomic>

#include "kernels.h"

namespace fl {

// =============================================================================== GEMV (T = 1)
// One wave computes R rows at a time; rows of a group share the x fragment read from LDS.
// U = chunks (of 512 K-elements per wave) whose loads are issued before any FMA.
template <typename WT, typename XT, int R, int U>
__global__ __launch_bounds__(256) void gemv_kernel(const WT *__restrict__ W, const XT *__restrict__ x,
                                                   const float *__restrict__ bias, void *__restrict__ out,
                                                   int N, int K, int epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    XT *xs = reinterpret_cast<XT *>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = K >> 3;                       // 8-element chunks; K % 8 == 0
    // stage x: 16 B per thread per step
    for (int c = tid; c < nchunk; c += 256) {
        if constexpr (sizeof(XT) == 2) {
            *reinterpret_cast<uint4v *>(xs + c * 8) = *reinterpret_cast<const uint4v *>(x + c * 8);
        } else {
            *reinterpret_cast<float4v *>(xs + c * 8) = *reinterpret_cast<const float4v *>(x + c * 8);
            *reinterpret_cast<float4v *>(xs + c * 8 + 4) = *reinterpret_cast<const float4v *>(x + c * 8 + 4);
        }
    }
    __syncthreads();

    const int ngroups = (N + R - 1) / R;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    for (int g = gw; g < ngroups; g += nw) {
        int rows[R];
        const WT *wp[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            int row;
            if (epi == EPI_GATEUP) { int q = g * (R / 2) + (r >> 1); row = (q >> 4) * 32 + (q & 15) + ((r & 1) << 4); }
            else row = g * R + r;
            rows[r] = row;
            wp[r] = W + (size_t)(row < N ? row : N - 1) * K;
        }
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = 0.f;

        int c0 = lane;
        // full blocks of U chunks: no predicates
        for (; c0 + 64 * (U - 1) < nchunk; c0 += 64 * U) {
            float w[R][U][8];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int r = 0; r < R; r++) load8_nt(wp[r] + (size_t)(c0 + 64 * u) * 8, w[r][u]);
#pragma unroll
            for (int u = 0; u < U; u++) {
                float xv[8];
                load8(xs + (c0 + 64 * u) * 8, xv);
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[r] = fmaf(w[r][u][j], xv[j], acc[r]);
            }
        }
        for (; c0 < nchunk; c0 += 64) {              // tail
            float xv[8];
            load8(xs + c0 * 8, xv);
#pragma unroll
            for (int r = 0; r < R; r++) {
                float w[8];
                load8_nt(wp[r] + (size_t)c0 * 8, w);
#pragma unroll
                for (int j = 0; j < 8; j++) acc[r] = fmaf(w[j], xv[j], acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
            if (epi == EPI_GATEUP) {
#pragma unroll
                for (int r = 0; r < R; r += 2) {
                    int q = g * (R / 2) + (r >> 1);
                    if (rows[r + 1] < N) {
                        float gt = acc[r], up = acc[r + 1];
                        float a = gt / (1.0f + expf(-gt)) * up;          // candle silu(g) * u
                        elem<XT>::st(reinterpret_cast<XT *>(out) + q, a);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (rows[r] < N) reinterpret_cast<float *>(out)[rows[r]] = acc[r] + (bias ? bias[rows[r]] : 0.f);
            }
        }
    }
}

static int env_int(const char *name, int dflt) {
    const char *s = getenv(name);
    return s && *s ? atoi(s) : dflt;
}

bool gemv_supported(int dtype, int64_t N, int64_t K) {
    (void)N;
    if (dtype != FL_DTYPE_BF16 && dtype != FL_DTYPE_F32) return false;
    size_t lds = (size_t)K * (dtype == FL_DTYPE_BF16 ? 2 : 4);
    return K % 8 == 0 && K >= 8 && lds <= 160 * 1024 - 256;
}

template <typename WT, typename XT, int R, int U>
static int launch_gemv_t(Launcher &L, const void *W, const void *x, const float *bias, void *y,
                         int64_t N, int64_t K, int epi) {
    auto kern = gemv_kernel<WT, XT, R, U>;
    size_t lds = ((size_t)K * sizeof(XT) + 15) & ~(size_t)15;
    if (lds > 64 * 1024) {
        static std::atomic<size_t> raised{0};      // per instantiation, process-wide
        if (raised.load() < lds) {
            FL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            raised.store(lds);
        }
    }
    int64_t ngroups = (N + R - 1) / R;
    int64_t blocks = (ngroups + 3) / 4;
    // grid-stride above ~8 workgroups per CU so x is staged at most 2048 times
    int64_t cap = env_int("FL_GEMV_MAXBLOCKS", 2048);
    if (blocks > cap) blocks = cap;
    double bytes = (double)N * K * sizeof(WT);
    return L.launch(KC_GEMV, bytes, 2.0 * N * K, kern, dim3((unsigned)blocks), dim3(256), lds,
                    (const WT *)W, (const XT *)x, bias, y, (int)N, (int)K, epi);
}

template <typename WT, typename XT>
static int launch_gemv(Launcher &L, const void *W, const void *x, const float *bias, void *y,
                       int64_t N, int64_t K, int epi) {
    static const int R = env_int("FL_GEMV_R", 2), U = env_int("FL_GEMV_U", 4);
    if (R == 4 && U == 2) return launch_gemv_t<WT, XT, 4, 2>(L, W, x, bias, y, N, K, epi);
    if (R == 4 && U == 4) return launch_gemv_t<WT, XT, 4, 4>(L, W, x, bias, y, N, K, epi);
    if (R == 2 && U == 8) return launch_gemv_t<WT, XT, 2, 8>(L, W, x, bias, y, N, K, epi);
    if (R == 2 && U == 2) return launch_gemv_t<WT, XT, 2, 2>(L, W, x, bias, y, N, K, epi);
    return launch_gemv_t<WT, XT, 2, 4>(L, W, x, bias, y, N, K, epi);
}

// =============================================================================== generic GEMM
// 64x64 output tile, 16-deep K slices through LDS (as fp32), 256 threads, 4x4 per thread with
// column stride 16 so that a thread owns gate column c and up column c+16 of the interleaved
// gate/up layout.  Any T, N, K.
template <typename WT, typename XT>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const WT *__restrict__ W, const XT *__restrict__ X,
                                                           const float *__restrict__ bias, void *__restrict__ out,
                                                           int T, int N, int K, int epi) {
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
                               int64_t T, int64_t N, int64_t K, int epi) {
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((T + 63) / 64));
    double bytes = (double)N * K * sizeof(WT) + (double)T * K * sizeof(XT);
    return L.launch(KC_GEMM_GENERIC, bytes, 2.0 * T * N * K, gemm_generic_kernel<WT, XT>, grid, dim3(256), 0,
                    (const WT *)W, (const XT *)x, bias, y, (int)T, (int)N, (int)K, epi);
}

int launch_gemm_mfma(Launcher &L, const void *W, const void *x, const float *bias, void *y,
                     int64_t T, int64_t N, int64_t K, int epi);   // k_gemm_mfma.hip

int launch_linear(Launcher &L, int dtype, const void *W, const void *x, const float *bias, void *y,
                  int64_t T, int64_t N, int64_t K, int epi) {
    if (T <= 0 || N <= 0 || K <= 0) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_linear: bad shape");
    if (epi == EPI_GATEUP && N % 32) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gate/up matrix rows must be a multiple of 32");
    static const int force_generic = env_int("FL_FORCE_GENERIC_GEMM", 0);
    if (dtype == FL_DTYPE_BF16) {
        if (T == 1 && gemv_supported(dtype, N, K)) return launch_gemv<bf16_t, bf16_t>(L, W, x, bias, y, N, K, epi);
        if (!force_generic && gemm_mfma_supported(dtype, T, N, K)) return launch_gemm_mfma(L, W, x, bias, y, T, N, K, epi);
        return launch_gemm_generic<bf16_t, bf16_t>(L, W, x, bias, y, T, N, K, epi);
    }
    if (dtype == FL_DTYPE_F32) {
        if (T == 1 && gemv_supported(dtype, N, K)) return launch_gemv<float, float>(L, W, x, bias, y, N, K, epi);
        return launch_gemm_generic<float, float>(L, W, x, bias, y, T, N, K, epi);
    }
    FL_FAIL(FL_ERR_UNSUPPORTED, "launch_linear: unsupported dtype %d", dtype);
}

}  // namespace fl
