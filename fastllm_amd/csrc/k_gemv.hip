// k_gemv.hip -- the decode (T = 1) weight-streaming kernel: y = W[N,K] . x  with fused prologue
// and epilogues.  This is the dominant kernel of the hot path: 97.5 % of a decode step's HBM
// bytes are linear-layer weights read exactly once.
//
// Streaming (HBM-bound, cdna guide "GEMV / M <= 16" row): each wave owns R rows at a time and
// reads them 16 B per lane (1 KiB per wave instruction), non-temporal, straight to VGPRs, U
// chunks of all R rows in flight before the first FMA; fp32 accumulate; 64-lane shuffle reduce.
// x (the activation vector) is staged once per workgroup in LDS and re-read with ds_read_b128.
//
// Prologue PRO_NORM (fused K2/K9, and K1 for layer 0): RMSNorm is folded around the dot product,
//     W . (v / m * w)  =  (1/m) * W . (v * w),     v = x_in + delta (or the token's embedding row),
// so the workgroup stages x' = v * w in LDS with no dependence on m = sqrt(mean(v^2) + eps): the
// sum of squares rides along in the same pass and is combined behind the SAME barrier as the
// staging; 1/m is applied to the accumulator in the epilogue.  No separate norm kernel, no xn
// round trip, no extra barrier.  Workgroup 0 also writes the updated residual v to x_out (a
// different buffer than x_in: other workgroups still read x_in).  The first weight loads are
// issued before the staging so HBM is busy from the first cycle.
//
// Epilogues: EPI_F32 (+bias) -> fp32;  EPI_GATEUP: silu(gate)*up on the 16-interleaved layout;
// EPI_QKV_ROPE (fused K4/K5): rows are paired (j, j+d/2) per head, RoPE is applied with the
// position from the device step state and q / rotated k / v go straight to the q buffer and
// the KV cache slot `len`.
#include <stdlib.h>

#include <atomic>

#include "kernels.h"

namespace fl {

__device__ inline void load_raw_nt(const bf16_t *p, uint4v (&r)[1]) {
    r[0] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(p));
}
__device__ inline void load_raw_nt(const float *p, uint4v (&r)[2]) {
    r[0] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(p));
    r[1] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(p + 4));
}
template <typename WT> __device__ inline void unpack_raw(const uint4v (&r)[sizeof(WT) == 2 ? 1 : 2], float (&o)[8]) {
    if constexpr (sizeof(WT) == 2) {
        unpack8(r[0], o);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) { o[i] = __uint_as_float(r[0][i]); o[4 + i] = __uint_as_float(r[1][i]); }
    }
}

template <typename WT, typename XT, int R, int U, int PRO>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float red[4];
    XT *xs = reinterpret_cast<XT *>(lds_raw);
    const WT *__restrict__ W = reinterpret_cast<const WT *>(a.W);
    const int N = a.N, K = a.K, epi = a.epi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = K >> 3;                       // 8-element chunks; K % 8 == 0
    const int half = a.d >> 1;
    const int ngroups = (N + R - 1) / R;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;

    auto row_of = [&](int g, int r) -> int {
        if (epi == EPI_GATEUP) { int q = g * (R / 2) + (r >> 1); return (q >> 4) * 32 + (q & 15) + ((r & 1) << 4); }
        if (epi == EPI_QKV_ROPE) { int q = g * (R / 2) + (r >> 1); int hd = q / half, j = q - hd * half; return hd * a.d + j + (r & 1) * half; }
        return g * R + r;
    };

    uint4v pre[R][U][sizeof(WT) == 2 ? 1 : 2];      // raw 16-B loads; unpacked at the FMA
    bool have_pre = false;
    float inv_m = 1.0f;
    if constexpr (PRO == PRO_NORM) {
        constexpr int NCH = 3;                       // K <= 6144 (host-checked)
        float v[NCH][8], wn[NCH][8];
        const WT *erow = nullptr;
        if (a.embed) erow = reinterpret_cast<const WT *>(a.embed) + (size_t)a.st->token * K;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = tid + 256 * i;
            if (c < nchunk) {
                if (erow) load8(erow + c * 8, v[i]); else load8(a.x_in + c * 8, v[i]);
                load8(a.norm_w + c * 8, wn[i]);
                if (a.delta) {
                    float dl[8];
                    load8(a.delta + c * 8, dl);
#pragma unroll
                    for (int j = 0; j < 8; j++) v[i][j] += dl[j];
                }
            }
        }
        // first weight block of this wave: in flight while x' is staged
        if (gw < ngroups && lane + 64 * (U - 1) < nchunk) {
            have_pre = true;
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    int row = row_of(gw, r);
                    load_raw_nt(W + (size_t)(row < N ? row : N - 1) * K + (size_t)(lane + 64 * u) * 8, pre[r][u]);
                }
        }
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = tid + 256 * i;
            if (c < nchunk) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { ss = fmaf(v[i][j], v[i][j], ss); o[j] = v[i][j] * wn[i][j]; }
                store8(xs + c * 8, o);
                if (blockIdx.x == 0 && a.x_out) store8(a.x_out + c * 8, v[i]);
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        ss = red[0] + red[1] + red[2] + red[3];
        inv_m = 1.0f / sqrtf(ss / (float)K + a.eps);           // candle rms_norm (App. A.2)
    } else {
        const XT *__restrict__ x = reinterpret_cast<const XT *>(a.x);
        if (a.x_scale) inv_m = *a.x_scale;
        for (int c = tid; c < nchunk; c += 256) {
            if constexpr (sizeof(XT) == 2) {
                *reinterpret_cast<uint4v *>(xs + c * 8) = *reinterpret_cast<const uint4v *>(x + c * 8);
            } else {
                *reinterpret_cast<float4v *>(xs + c * 8) = *reinterpret_cast<const float4v *>(x + c * 8);
                *reinterpret_cast<float4v *>(xs + c * 8 + 4) = *reinterpret_cast<const float4v *>(x + c * 8 + 4);
            }
        }
        __syncthreads();
    }

    for (int g = gw; g < ngroups; g += nw) {
        int rows[R];
        const WT *wp[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            rows[r] = row_of(g, r);
            wp[r] = W + (size_t)(rows[r] < N ? rows[r] : N - 1) * K;
        }
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = 0.f;

        int c0 = lane;
        if (PRO == PRO_NORM && have_pre && g == gw) {            // consume the prefetched block
#pragma unroll
            for (int u = 0; u < U; u++) {
                float xv[8];
                load8(xs + (c0 + 64 * u) * 8, xv);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    float wv[8];
                    unpack_raw<WT>(pre[r][u], wv);
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[r] = fmaf(wv[j], xv[j], acc[r]);
                }
            }
            c0 += 64 * U;
        }
        for (; c0 + 64 * (U - 1) < nchunk; c0 += 64 * U) {      // full blocks of U chunks: no predicates
            uint4v w[R][U][sizeof(WT) == 2 ? 1 : 2];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int r = 0; r < R; r++) load_raw_nt(wp[r] + (size_t)(c0 + 64 * u) * 8, w[r][u]);
#pragma unroll
            for (int u = 0; u < U; u++) {
                float xv[8];
                load8(xs + (c0 + 64 * u) * 8, xv);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    float wv[8];
                    unpack_raw<WT>(w[r][u], wv);
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[r] = fmaf(wv[j], xv[j], acc[r]);
                }
            }
        }
        for (; c0 < nchunk; c0 += 64) {                          // tail
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
        for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]) * inv_m;
        if (lane == 0) {
            if (epi == EPI_GATEUP) {
#pragma unroll
                for (int r = 0; r < R; r += 2) {
                    const int q = g * (R / 2) + (r >> 1);
                    if (rows[r + 1] < N) {
                        const float gt = acc[r], up = acc[r + 1];
                        const float act = gt / (1.0f + expf(-gt)) * up;          // candle silu(g) * u
                        elem<XT>::st(reinterpret_cast<XT *>(a.out) + q, act);
                    }
                }
            } else if (epi == EPI_QKV_ROPE) {
                const uint32_t pos = a.st->pos, slot = a.st->len;
                const uint32_t p = pos < (uint32_t)a.max_pos ? pos : (uint32_t)a.max_pos - 1;
#pragma unroll
                for (int r = 0; r < R; r += 2) {
                    if (rows[r + 1] >= N) continue;
                    const int q = g * (R / 2) + (r >> 1);
                    const int hd = q / half, j = q - hd * half;
                    float x0 = acc[r], x1 = acc[r + 1];
                    if (a.bias) { x0 += a.bias[rows[r]]; x1 += a.bias[rows[r + 1]]; }
                    XT *dst;
                    if (hd < a.H + a.Hkv) {                                   // rotate-half RoPE (App. A.4)
                        const float c = a.cos_tab[(size_t)p * half + j], s = a.sin_tab[(size_t)p * half + j];
                        const float r0 = x0 * c - x1 * s, r1 = x0 * s + x1 * c;
                        x0 = r0; x1 = r1;
                        dst = hd < a.H ? reinterpret_cast<XT *>(a.q_out) + (size_t)hd * a.d
                                       : reinterpret_cast<XT *>(a.k_cache) + ((size_t)(hd - a.H) * a.max_seq + slot) * a.d;
                    } else {
                        dst = reinterpret_cast<XT *>(a.v_cache) + ((size_t)(hd - a.H - a.Hkv) * a.max_seq + slot) * a.d;
                    }
                    elem<XT>::st(dst + j, x0);
                    elem<XT>::st(dst + j + half, x1);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (rows[r] < N) reinterpret_cast<float *>(a.out)[rows[r]] = acc[r] + (a.bias ? a.bias[rows[r]] : 0.f);
            }
        }
    }
}

static std::atomic<int> g_gemv_r{0}, g_gemv_u{0}, g_gemv_maxblocks{0}, g_gemv_maxblocks_norm{0};
static int env_int(const char *name, int dflt) { const char *s = getenv(name); return s && *s ? atoi(s) : dflt; }

void gemv_set_tuning(int R, int U, int maxblocks, int maxblocks_norm) {
    if (R > 0) g_gemv_r = R;
    if (U > 0) g_gemv_u = U;
    if (maxblocks > 0) g_gemv_maxblocks = maxblocks;
    if (maxblocks_norm > 0) g_gemv_maxblocks_norm = maxblocks_norm;
}

bool gemv_supported(int dtype, int64_t N, int64_t K) {
    (void)N;
    if (dtype != FL_DTYPE_BF16 && dtype != FL_DTYPE_F32) return false;
    size_t lds = (size_t)K * (dtype == FL_DTYPE_BF16 ? 2 : 4);
    return K % 8 == 0 && K >= 8 && lds <= 160 * 1024 - 256;
}
bool gemv_norm_supported(int dtype, int64_t N, int64_t K) { return gemv_supported(dtype, N, K) && K <= 6144; }

template <typename WT, typename XT, int R, int U, int PRO>
static int launch_gemv_t(Launcher &L, const GemvArgs &a) {
    auto kern = gemv_kernel<WT, XT, R, U, PRO>;
    const int64_t N = a.N, K = a.K;
    size_t lds = ((size_t)K * sizeof(XT) + 15) & ~(size_t)15;
    if (lds > 64 * 1024) {
        static std::atomic<size_t> raised{0};      // per instantiation, process-wide
        if (raised.load() < lds) {
            FL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            raised.store(lds);
        }
    }
    // Grid: every wave should get the same number of row groups (a ragged last round costs up to
    // 1/rounds of the kernel), and all workgroups should be resident at once (<= cap).
    const int64_t ngroups = (N + R - 1) / R;
    int mb = g_gemv_maxblocks.load(), mbn = g_gemv_maxblocks_norm.load();
    if (!mb) { mb = env_int("FL_GEMV_MAXBLOCKS", 768); g_gemv_maxblocks = mb; }
    if (!mbn) { mbn = env_int("FL_GEMV_MAXBLOCKS_NORM", 768); g_gemv_maxblocks_norm = mbn; }
    const int64_t cap = PRO == PRO_NORM ? mbn : mb;
    int64_t blocks = (ngroups + 3) / 4;
    if (blocks > cap) {
        double best_eff = 0.0; int64_t best_b = cap;
        for (int64_t b = cap; b >= cap / 2 && b >= 1; b--) {
            const int64_t per_wave = (ngroups + 4 * b - 1) / (4 * b);
            const double eff = (double)ngroups / (double)(per_wave * 4 * b);
            if (eff > best_eff + 1e-9) { best_eff = eff; best_b = b; }
        }
        blocks = best_b;
    }
    double bytes = (double)N * K * sizeof(WT);
    return L.launch(KC_GEMV, bytes, 2.0 * N * K, kern, dim3((unsigned)blocks), dim3(256), lds, a);
}

template <typename WT, typename XT, int PRO>
static int launch_gemv_ru(Launcher &L, const GemvArgs &a) {
    int R = g_gemv_r.load(), U = g_gemv_u.load();
    if (!R) { R = env_int("FL_GEMV_R", 2); g_gemv_r = R; }
    if (!U) { U = env_int("FL_GEMV_U", 4); g_gemv_u = U; }
    if (R == 4 && U == 2) return launch_gemv_t<WT, XT, 4, 2, PRO>(L, a);
    if (R == 4 && U == 4) return launch_gemv_t<WT, XT, 4, 4, PRO>(L, a);
    if (R == 2 && U == 8) return launch_gemv_t<WT, XT, 2, 8, PRO>(L, a);
    if (R == 2 && U == 2) return launch_gemv_t<WT, XT, 2, 2, PRO>(L, a);
    return launch_gemv_t<WT, XT, 2, 4, PRO>(L, a);
}

int launch_gemv(Launcher &L, int dtype, const GemvArgs &a) {
    if (a.N <= 0 || a.K <= 0) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv: bad shape");
    if (!gemv_supported(dtype, a.N, a.K)) FL_FAIL(FL_ERR_UNSUPPORTED, "launch_gemv: K=%d unsupported", a.K);
    if (a.pro == PRO_NORM && !gemv_norm_supported(dtype, a.N, a.K)) FL_FAIL(FL_ERR_UNSUPPORTED, "fused norm needs K <= 6144");
    if (a.epi == EPI_GATEUP && a.N % 32) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gate/up matrix rows must be a multiple of 32");
    if (a.epi == EPI_QKV_ROPE && (a.d <= 0 || a.d % 2 || a.N != (a.H + 2 * a.Hkv) * a.d)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "bad qkv shape");
    if (dtype == FL_DTYPE_BF16)
        return a.pro == PRO_NORM ? launch_gemv_ru<bf16_t, bf16_t, PRO_NORM>(L, a) : launch_gemv_ru<bf16_t, bf16_t, PRO_X>(L, a);
    return a.pro == PRO_NORM ? launch_gemv_ru<float, float, PRO_NORM>(L, a) : launch_gemv_ru<float, float, PRO_X>(L, a);
}

}  // namespace fl
