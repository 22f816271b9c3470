// k_gemv.hip -- the decode (T = 1) weight-streaming kernel: y = W[N,K] . x  with fused prologue
// and epilogues.  This is the dominant kernel of the hot path: 97.5 % of a decode step's HBM
// bytes are linear-layer weights read exactly once.
//
// Streaming (HBM-bound, cdna guide "GEMV / M <= 16" row): each wave owns R rows at a time and
// reads them 16 B per lane (1 KiB per wave instruction), non-temporal, straight to VGPRs -- no
// LDS round trip for read-once bytes.  bf16 weights: the wave's (row group, K block) items form
// ONE stream, a block (U chunks x R rows = 4 or 8 KiB) ahead in a second register buffer, every
// load unconditional so that hipcc keeps counted vmcnt waits; the arithmetic is v_dot2c_f32_bf16
// (common.h: inline asm, with the wait the gfx940+ dot-result hazard asks for behind every run
// of dots; tests/test_dot_hazard.py checks the shipped code objects).  A projection short enough
// for one row group per wave requests its whole share before the prologue (SMALL).  fp32 weights:
// straight-line blocks of U chunks x R rows with FMAs, the K tail as one more block.  8..12 waves
// per CU overlap each other's load and multiply phases, and the first block is requested BEFORE
// x is staged so HBM is busy during the prologue.  fp32 accumulate; 64-lane shuffle reduce.  x is
// staged once per workgroup in LDS (ds_read_b128).  Against a bare read stream of the same bytes
// (tools/micro/stream_ceiling.hip, 6.76 TB/s) the largest launch of a Mistral-7B step is 6 % slower.
//
// Geometry: one workgroup per CU (or two), 4..12 waves each, chosen so that every wave gets the
// same number of row groups (a ragged last round costs 1/rounds of the kernel) and everything is
// resident at once.
//
// Prologue PRO_NORM (fused K2/K9, and K1 for layer 0): RMSNorm is folded around the dot product,
//     W . (v / m * w)  =  (1/m) * W . (v * w),     v = x_in + delta (or the token's embedding row),
// so the workgroup stages x' = v * w in LDS with no dependence on m = sqrt(mean(v^2) + eps): the
// sum of squares rides along in the same pass and is combined behind the SAME barrier as the
// staging; 1/m is applied to the accumulator in the epilogue.  Workgroup 0 also writes the updated
// residual v to x_out (a different buffer than x_in: other workgroups still read x_in).
//
// Epilogues: EPI_F32 (+bias) -> fp32, or -- GemvArgs::ll, the row-parallel o_proj / down_proj of a tensor-parallel
// group -- summed over the ranks right here (comm_ll.h);  EPI_GATEUP: silu(gate)*up on the 16-interleaved layout;
// EPI_QKV_ROPE (fused K4/K5): rows are paired (j, j+d/2) per head, RoPE is applied with the
// position from the device step state and q / rotated k / v go straight to the q buffer and
// the KV cache slot `len`.
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "comm_ll.h"
#include "kernels.h"

namespace fl {

constexpr int kGemvMaxThreads = 768;     // 12 waves: 170 VGPRs per lane available
constexpr int kMaxDevices = 64;

template <typename WT> struct RawChunk { uint4v v[sizeof(WT) == 2 ? 1 : 2]; };

#ifndef FL_GEMV_DOT2
#define FL_GEMV_DOT2 1
#endif
constexpr bool kGemvDot2 = FL_GEMV_DOT2 != 0;
#ifndef FL_GEMV_PIPE
#define FL_GEMV_PIPE 1
#endif
constexpr bool kGemvPipe = FL_GEMV_PIPE != 0 && kGemvDot2;

#if !defined(FL_GEMV_PLAIN_LOADS) || !defined(FL_EXPERIMENTAL)
#undef FL_GEMV_PLAIN_LOADS
#define FL_GEMV_PLAIN_LOADS 0              // (-DFL_GEMV_PLAIN_LOADS=1, EXPERIMENTAL build only: the weight stream through plain instead of non-temporal loads)
#endif
__device__ inline void load_raw_nt(const bf16_t *p, RawChunk<bf16_t> &r) {
    if constexpr (FL_GEMV_PLAIN_LOADS != 0) r.v[0] = *reinterpret_cast<const uint4v *>(p);
    else r.v[0] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(p));
}
__device__ inline void load_raw_nt(const float *p, RawChunk<float> &r) {
    r.v[0] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(p));
    r.v[1] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(p + 4));
}
__device__ inline void unpack_raw(const RawChunk<bf16_t> &r, float (&o)[8]) { unpack8(r.v[0], o); }
__device__ inline void unpack_raw(const RawChunk<float> &r, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; i++) { o[i] = __uint_as_float(r.v[0][i]); o[4 + i] = __uint_as_float(r.v[1][i]); }
}

// MAXT = 768: up to 12 waves per workgroup (170 VGPRs per lane).
// SMALL: every wave owns at most ONE row group and K <= 2 blocks of U chunks (host-checked): the whole
// weight share of the wave (2*U*R KiB) is requested before x is staged -- one HBM round trip instead of
// two for the short QKV / o_proj launches, where ramp-up is most of the kernel.
template <typename WT, typename XT, int R, int U, int PRO, int MAXT, bool SMALL, int EPI, bool FUSE_AR>
__global__ __launch_bounds__(MAXT) void gemv_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float red[kGemvMaxThreads / 64];
    __shared__ float cv[kGemvMaxThreads / 64];       // ArgMax candidates of the waves (GemvArgs::amax)
    __shared__ int ci[kGemvMaxThreads / 64];
    XT *xs = reinterpret_cast<XT *>(lds_raw);
    const WT *__restrict__ W = reinterpret_cast<const WT *>(a.W);
    const int N = a.N, K = a.K;
    constexpr int epi = EPI;                         // compile-time: the epilogue's operands do not occupy registers of the others
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nwv = nthr >> 6;
    const int nchunk = K >> 3;                       // 8-element chunks; K % 8 == 0
    const int half = a.d >> 1;
    const int ngroups = (N + R - 1) / R;
    const int gw = blockIdx.x * nwv + wave, nw = gridDim.x * nwv;

    auto row_of = [&](int g, int r) -> int {
        if (epi == EPI_GATEUP) { int q = g * (R / 2) + (r >> 1); return (q >> 4) * 32 + (q & 15) + ((r & 1) << 4); }
        if (epi == EPI_QKV_ROPE) { int q = g * (R / 2) + (r >> 1); int hd = q / half, j = q - hd * half; return hd * a.d + j + (r & 1) * half; }
        return g * R + r;
    };
    typedef RawChunk<WT> Buf[R][U];
    // first K block of this wave's first row group: requested before x is staged, so HBM is busy
    // during the prologue
    constexpr int NPRE = SMALL ? 2 : 1;              // K blocks requested ahead of the prologue
    Buf pre[NPRE];
    bool have_pre[NPRE];
#pragma unroll
    for (int i = 0; i < NPRE; i++) have_pre[i] = gw < ngroups && lane + 64 * (U * (i + 1) - 1) < nchunk;
    // PIPE (bf16, every launch that is not SMALL): the wave's (row group, K block) items form ONE stream, requested a block
    // ahead into two register buffers -- while a block is multiplied the next one is in flight, also across row groups and
    // their epilogues.  Every load of the stream is unconditional (no control flow around loads: hipcc then keeps COUNTED
    // vmcnt waits): past the end of K a lane re-reads the last chunk (its x is zeroed); the last one or two items are
    // peeled off the loop, so that the loop itself never requests past the wave's last item.
    constexpr bool PIPE = kGemvPipe && !SMALL && sizeof(WT) == 2 && sizeof(XT) == 2;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);                       // wave-uniform copies for the stream's bookkeeping
    const int gw_u = blockIdx.x * nwv + wave_u;
    const int nb = (nchunk + 64 * U - 1) / (64 * U);                                // K blocks per row group (the last may be partial)
    const int n_items = gw_u < ngroups ? (ngroups - gw_u + nw - 1) / nw * nb : 0;  // this wave's items
    int lg = gw_u, lb = 0;                                                          // load stream: next item = (row group, block)
    auto load_next = [&](Buf &buf) {
        const int g = min(lg, ngroups - 1);                                         // (a wave without work reads the last group's first block once)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int row = row_of(g, r);
            const WT *wr = W + (size_t)(row < N ? row : N - 1) * K;
#pragma unroll
            for (int u = 0; u < U; u++) load_raw_nt(wr + (size_t)min(lane + 64 * (U * lb + u), nchunk - 1) * 8, buf[r][u]);
        }
        if (++lb == nb) { lb = 0; lg += nw; }
    };
    auto prefetch = [&]() {
        if constexpr (PIPE) { load_next(pre[0]); return; }
#pragma unroll
        for (int i = 0; i < NPRE; i++) {
            if (!have_pre[i]) continue;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int row = row_of(gw, r);
                const WT *wp = W + (size_t)(row < N ? row : N - 1) * K;
#pragma unroll
                for (int u = 0; u < U; u++) load_raw_nt(wp + (size_t)(lane + 64 * (U * i + u)) * 8, pre[i][r][u]);
            }
        }
    };
    float inv_m = 1.0f;
    if constexpr (PRO == PRO_NORM) {
        constexpr int NCH = SMALL ? 1 : 3;           // nthr * NCH * 8 >= K (host-checked)
        float v[NCH][8], wn[NCH][8], dl[NCH][8];
        const WT *erow = nullptr;
        if (a.embed) erow = reinterpret_cast<const WT *>(a.embed) + (size_t)a.st->token * K;
#pragma unroll
        for (int i = 0; i < NCH; i++) {                  // requests only: nothing here waits (an add of delta in this loop made the
            const int c = tid + nthr * i;                //  weight stream below start a round trip late)
            if (c < nchunk) {
                if (erow) load8(erow + c * 8, v[i]); else load8(a.x_in + c * 8, v[i]);
                load8(a.norm_w + c * 8, wn[i]);
                if (a.delta) {
                    load8(a.delta + c * 8, dl[i]);
                    for (int s0 = 1; s0 < a.delta_nslab; s0 += 7) {        // partial vectors of a K-sliced producer (the opt-in fused attention
                        float ds[7][8];                                     // + o_proj launch): up to seven per round trip, fixed order; this path
#pragma unroll                                                              // does wait before the weights are requested
                        for (int u = 0; u < 7; u++)
                            if (s0 + u < a.delta_nslab) load8(a.delta + (size_t)(s0 + u) * K + c * 8, ds[u]);
#pragma unroll
                        for (int u = 0; u < 7; u++)
                            if (s0 + u < a.delta_nslab) {
#pragma unroll
                                for (int j = 0; j < 8; j++) dl[i][j] += ds[u][j];
                            }
                    }
                }
            }
        }
        // the weight stream starts now: issued after the (short) activation loads so that the counted
        // wait for those does not have to drain the long weight loads
        prefetch();
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = tid + nthr * i;
            if (c < nchunk) {
                if (a.delta) {
#pragma unroll
                    for (int j = 0; j < 8; j++) v[i][j] += dl[i][j];
                }
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
        ss = 0.f;
        for (int w = 0; w < nwv; w++) ss += red[w];
        inv_m = 1.0f / sqrtf(ss / (float)K + a.eps);           // candle rms_norm (App. A.2)
    } else {
        const XT *__restrict__ x = reinterpret_cast<const XT *>(a.x);
        constexpr int NXR = 4;
        uint4v xr[NXR][sizeof(XT) == 2 ? 1 : 2];
        // stage x through registers: up to NXR chunks per thread are requested before the weights
#pragma unroll
        for (int i = 0; i < NXR; i++) {
            const int c = tid + nthr * i;
            if (c < nchunk) {
                xr[i][0] = *reinterpret_cast<const uint4v *>(x + c * 8);
                if constexpr (sizeof(XT) == 4) xr[i][1] = *reinterpret_cast<const uint4v *>(x + c * 8 + 4);
            }
        }
        if (a.x_scale) inv_m = *a.x_scale;
        prefetch();
#pragma unroll
        for (int i = 0; i < NXR; i++) {
            const int c = tid + nthr * i;
            if (c < nchunk) {
                *reinterpret_cast<uint4v *>(xs + c * 8) = xr[i][0];
                if constexpr (sizeof(XT) == 4) *reinterpret_cast<uint4v *>(xs + c * 8 + 4) = xr[i][1];
            }
        }
        for (int c = tid + NXR * nthr; c < nchunk; c += nthr) {        // very long K: the rest
            if constexpr (sizeof(XT) == 2) {
                *reinterpret_cast<uint4v *>(xs + c * 8) = *reinterpret_cast<const uint4v *>(x + c * 8);
            } else {
                *reinterpret_cast<float4v *>(xs + c * 8) = *reinterpret_cast<const float4v *>(x + c * 8);
                *reinterpret_cast<float4v *>(xs + c * 8 + 4) = *reinterpret_cast<const float4v *>(x + c * 8 + 4);
            }
        }
        __syncthreads();
    }

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.f;

    // RoPE epilogue operands are requested up front (position from the step state, then this wave's cos/sin
    // pairs) so that their two dependent round trips overlap the weight stream instead of trailing it
    uint32_t rope_p = 0, rope_slot = 0;
    float rope_c[R / 2], rope_s[R / 2], rope_b0[R / 2], rope_b1[R / 2];            // (bias of the pair's two rows: Qwen2)
    auto rope_prefetch = [&](int g) {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            const int q = g * (R / 2) + (r >> 1);
            const int hd = q / half, j = q - hd * half;
            const bool rot = hd < a.H + a.Hkv;
            rope_c[r >> 1] = rot ? a.cos_tab[(size_t)rope_p * half + j] : 1.f;
            rope_s[r >> 1] = rot ? a.sin_tab[(size_t)rope_p * half + j] : 0.f;
            const int r0w = row_of(g, r), r1w = row_of(g, r + 1);
            rope_b0[r >> 1] = a.bias && r1w < N ? a.bias[r0w] : 0.f;                // requested with the tables: in the epilogue it was a
            rope_b1[r >> 1] = a.bias && r1w < N ? a.bias[r1w] : 0.f;                //  round trip at the very end of the launch
        }
    };
    if (epi == EPI_QKV_ROPE) {
        const uint32_t pos = a.st->pos;
        rope_slot = a.st->len;
        rope_p = pos < (uint32_t)a.max_pos ? pos : (uint32_t)a.max_pos - 1;
        if (gw < ngroups) rope_prefetch(gw);
    }

    float best_v = -INFINITY; int best_i = -1;                                      // running ArgMax of this wave's rows (lane 0)
    auto finish_group = [&](int g) {
        float sum[R];
#pragma unroll
        for (int r = 0; r < R; r++) { sum[r] = wave_sum(acc[r]) * inv_m; acc[r] = 0.f; }
        if constexpr (FUSE_AR) {                           // row-parallel projection of a tensor-parallel group (EPI_F32, no bias: host-checked)
            ll_allreduce_rows<R>(a.ll, a.ll_slot, g * R, N, sum, reinterpret_cast<float *>(a.out), lane);
            return;
        }
        if (lane != 0) return;
        if (epi == EPI_GATEUP) {
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const int q = g * (R / 2) + (r >> 1);
                if (row_of(g, r + 1) < N) {
                    const float gt = sum[r], up = sum[r + 1];
                    const float act = gt / (1.0f + expf(-gt)) * up;          // candle silu(g) * u
                    elem<XT>::st(reinterpret_cast<XT *>(a.out) + q, act);
                }
            }
        } else if (epi == EPI_QKV_ROPE) {
            const uint32_t slot = rope_slot;
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const int r1w = row_of(g, r + 1);
                if (r1w >= N) continue;
                const int q = g * (R / 2) + (r >> 1);
                const int hd = q / half, j = q - hd * half;
                float x0 = sum[r], x1 = sum[r + 1];
                if (a.bias) { x0 += rope_b0[r >> 1]; x1 += rope_b1[r >> 1]; }
                XT *dst;
                size_t stride = 1;                                        // element stride between j and j+1
                if (hd < a.H + a.Hkv) {                                   // rotate-half RoPE (App. A.4)
                    const float c = rope_c[r >> 1], s = rope_s[r >> 1];
                    float t0, t1;
                    rope_rotate(x0, x1, c, s, t0, t1);
                    x0 = t0; x1 = t1;
                    dst = hd < a.H ? reinterpret_cast<XT *>(a.q_out) + (size_t)hd * a.d
                                   : reinterpret_cast<XT *>(a.k_cache) + ((size_t)(hd - a.H) * a.max_seq + slot) * a.d;
                } else if (a.v_ld > 0) {                                  // transposed value cache [Hkv][d][v_ld]
                    dst = reinterpret_cast<XT *>(a.v_cache) + (size_t)(hd - a.H - a.Hkv) * a.d * a.v_ld + slot;
                    stride = (size_t)a.v_ld;
                } else {
                    dst = reinterpret_cast<XT *>(a.v_cache) + ((size_t)(hd - a.H - a.Hkv) * a.max_seq + slot) * a.d;
                }
                elem<XT>::st(dst + (size_t)j * stride, x0);
                elem<XT>::st(dst + (size_t)(j + half) * stride, x1);
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int row = row_of(g, r);
                if (row < N) {
                    const float y = sum[r] + (a.bias ? a.bias[row] : 0.f);
                    reinterpret_cast<float *>(a.out)[row] = y;
                    if (a.amax && (best_i < 0 || y > best_v || (y == best_v && row > best_i))) { best_v = y; best_i = row; }   // (lane 0; as argmax_last)
                }
            }
        }
    };
    // the workgroup's ArgMax candidate (EPI_F32 with GemvArgs::amax): lane 0 of every wave holds the best of its rows
    auto leave_candidate = [&]() {
        if constexpr (EPI != EPI_F32 || FUSE_AR) return;
        if (!a.amax) return;                                                          // (kernel argument: uniform)
        if (lane == 0) { cv[wave] = best_v; ci[wave] = best_i; }
        __syncthreads();
        if (tid == 0) {
            float bv = cv[0]; int bi = ci[0];
            for (int w = 1; w < nwv; w++)
                if (ci[w] >= 0 && (bi < 0 || cv[w] > bv || (cv[w] == bv && ci[w] > bi))) { bv = cv[w]; bi = ci[w]; }
            a.amax[1 + blockIdx.x] = ArgmaxCand{bv, bi};
            if (blockIdx.x == 0) a.amax[0] = ArgmaxCand{0.f, (int)gridDim.x};
        }
    };
    auto fma_block = [&](const Buf &buf, int c0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if constexpr (kGemvDot2 && sizeof(WT) == 2 && sizeof(XT) == 2) {
                // bf16 x bf16: four v_dot2c_f32_bf16 per 16 bytes of weights (exact products, fp32 accumulate) instead of eight
                // unpacks + eight FMAs -- the wave is back at its loads sooner
                const uint4v xr = *reinterpret_cast<const uint4v *>(xs + (c0 + 64 * u) * 8);
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[r] = dot2c_bf16(buf[r][u].v[0][j], xr[j], acc[r]);
            } else {
                float xv[8];
                load8(xs + (c0 + 64 * u) * 8, xv);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    float wv[8];
                    unpack_raw(buf[r][u], wv);
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[r] = fmaf(wv[j], xv[j], acc[r]);
                }
            }
        }
        if constexpr (kGemvDot2 && sizeof(WT) == 2 && sizeof(XT) == 2) {
#pragma unroll
            for (int r = 0; r < R; r++) dot2c_settle(acc[r]);
        }
    };

    if constexpr (SMALL) {
        if (gw < ngroups) {
#pragma unroll
            for (int i = 0; i < NPRE; i++)
                if (have_pre[i]) fma_block(pre[i], lane + 64 * U * i);
            finish_group(gw);
        }
        leave_candidate();
        return;
    }
    if constexpr (PIPE) {
        int cg = gw_u, cb = 0;                                                      // consume stream
        const bool ragged = nchunk % (64 * U) != 0;
        auto consume = [&](const Buf &buf) {
            if (epi == EPI_QKV_ROPE && cb == 0 && cg != gw_u) rope_prefetch(cg);
            const int c0 = lane + 64 * U * cb;
            if (ragged && cb == nb - 1) {                                           // wave-uniform: the partial last block of K
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int ci = c0 + 64 * u;
                    uint4v xr = *reinterpret_cast<const uint4v *>(xs + min(ci, nchunk - 1) * 8);
                    if (ci >= nchunk) xr = uint4v{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int r = 0; r < R; r++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[r] = dot2c_bf16(buf[r][u].v[0][j], xr[j], acc[r]);
                }
#pragma unroll
                for (int r = 0; r < R; r++) dot2c_settle(acc[r]);
            } else {
                fma_block(buf, c0);
            }
            if (++cb == nb) { finish_group(cg); cb = 0; cg += nw; }
        };
        Buf nxt;
        int t = 0;
#pragma nounroll
        for (; t + 2 < n_items; t += 2) {
            load_next(nxt);
            consume(pre[0]);
            load_next(pre[0]);
            consume(nxt);
        }
        if (n_items - t == 2) { load_next(nxt); consume(pre[0]); consume(nxt); }
        else if (n_items - t == 1) consume(pre[0]);
        leave_candidate();
        return;
    }
#pragma nounroll
    for (int g = gw; g < ngroups; g += nw) {
        if (epi == EPI_QKV_ROPE && g != gw) rope_prefetch(g);
        const WT *wp[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int row = row_of(g, r);
            wp[r] = W + (size_t)(row < N ? row : N - 1) * K;
        }
        int c0 = lane;
        if (g == gw) {
#pragma unroll
            for (int i = 0; i < NPRE; i++)
                if (have_pre[i]) { fma_block(pre[i], c0); c0 += 64 * U; }
        }
#pragma nounroll
        for (; c0 + 64 * (U - 1) < nchunk; c0 += 64 * U) {      // full blocks of U chunks: straight-line, counted waits
            Buf w;
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int r = 0; r < R; r++) load_raw_nt(wp[r] + (size_t)(c0 + 64 * u) * 8, w[r][u]);
            fma_block(w, c0);
        }
        // K tail: the remaining 1..U-1 whole 64-lane chunks as ONE straight-line block (K = 3584, 5632,
        // 18944 ... all leave such a remainder), then a generic loop for a ragged last chunk
        // ... and the tensor-parallel slices (K = 1792, 4736, 896 ...) end in a PARTIAL 64-lane chunk: the lanes past
        // the end load a clamped (valid) address and contribute zeros, so the block stays straight-line as well
        {
            const int remaining = nchunk - (c0 - lane);          // wave-uniform
            const int nblk = (remaining + 63) >> 6;
            if (remaining > 0 && nblk < U) {
                auto tail_block = [&](auto TU) {
                    constexpr int NT = decltype(TU)::value;
                    RawChunk<WT> w[R][NT];
#pragma unroll
                    for (int u = 0; u < NT; u++) {
                        const int ci = c0 + 64 * u, cc = ci < nchunk ? ci : nchunk - 1;
#pragma unroll
                        for (int r = 0; r < R; r++) load_raw_nt(wp[r] + (size_t)cc * 8, w[r][u]);
                    }
#pragma unroll
                    for (int u = 0; u < NT; u++) {
                        const int ci = c0 + 64 * u;
                        const bool ok = ci < nchunk;
                        float xv[8];
                        load8(xs + (ok ? ci : nchunk - 1) * 8, xv);
#pragma unroll
                        for (int j = 0; j < 8; j++) xv[j] = ok ? xv[j] : 0.f;
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            float wv[8];
                            unpack_raw(w[r][u], wv);
#pragma unroll
                            for (int j = 0; j < 8; j++) acc[r] = fmaf(wv[j], xv[j], acc[r]);
                        }
                    }
                    c0 += 64 * NT;
                };
                if (nblk == 1) tail_block(std::integral_constant<int, 1>{});
                else if (nblk == 2) tail_block(std::integral_constant<int, 2>{});
                else if (nblk == 3) tail_block(std::integral_constant<int, 3>{});
                else if (U > 4 && nblk == 4) tail_block(std::integral_constant<int, 4>{});
            }
        }
#pragma nounroll
        for (; c0 < nchunk; c0 += 64) {                          // ragged tail of K
            float xv[8];
            load8(xs + c0 * 8, xv);
#pragma unroll
            for (int r = 0; r < R; r++) {
                float wv[8];
                load8_nt(wp[r] + (size_t)c0 * 8, wv);
#pragma unroll
                for (int j = 0; j < 8; j++) acc[r] = fmaf(wv[j], xv[j], acc[r]);
            }
        }
        finish_group(g);
    }
    leave_candidate();
}

// forced grid / waves per workgroup (fl_tune "gemv_blocks" / "gemv_waves"; 0 = automatic).  Rows per wave pass and chunks per block
// are rows of the switch table (TK_GEMV_R, TK_GEMV_U): read per launch, so "reload_env" reaches them like every other switch.
static std::atomic<int> g_force_blocks{0}, g_force_waves{0};

void gemv_set_tuning(int blocks, int waves) {
    if (blocks >= 0) g_force_blocks = blocks;
    if (waves >= 0) g_force_waves = waves;
}

bool gemv_supported(int dtype, int64_t N, int64_t K) {
    (void)N;
    if (dtype != FL_DTYPE_BF16 && dtype != FL_DTYPE_F32) return false;
    size_t lds = (size_t)K * (dtype == FL_DTYPE_BF16 ? 2 : 4);
    return K % 8 == 0 && K >= 8 && lds <= 160 * 1024 - 256;
}
// fused-norm prologue: every staging thread holds at most 3 chunks of 8 in registers, and the
// smallest workgroup is 256 threads
bool gemv_norm_supported(int dtype, int64_t N, int64_t K) { return gemv_supported(dtype, N, K) && K <= 6144; }

static int cu_count() {                      // of the current device (the shards of a group may sit on different ones)
    static std::atomic<int> cached[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    int n = cached[dev].load();
    if (!n) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
        cached[dev].store(n);
    }
    return n;
}

// Pick (workgroups, waves per workgroup): equal row groups per wave, everything resident at once
// (<= 12 waves per CU at 170 VGPRs), as many waves per CU as that allows.
static void pick_geometry(int64_t ngroups, size_t lds_bytes, int *blocks_out, int *waves_out) {
    const int cus = cu_count();
    int fb = g_force_blocks.load(), fw = g_force_waves.load();
    if (fb > 0 && fw > 0) { *blocks_out = fb; *waves_out = fw; return; }
    if (ngroups <= (int64_t)cus * 4) {                 // small matrix: 4-wave workgroups, one group per wave
        *waves_out = 4; *blocks_out = (int)((ngroups + 3) / 4);
        return;
    }
    static const int kWaves[] = {12, 11, 10, 9, 8, 7, 6, 5, 4};
    double best = -1.0; int bb = cus, bw = 8;
    for (int mult = 1; mult <= 2; mult++) {
        for (int w : kWaves) {
            if (mult * w > 12) continue;
            if ((size_t)mult * lds_bytes > 150 * 1024) continue;
            const int64_t wt = (int64_t)cus * mult * w;
            const int64_t per = (ngroups + wt - 1) / wt;
            const double eff = (double)ngroups / (double)(per * wt);
            // prefer balance, then more waves per CU (latency hiding), then fewer workgroups
            const double score = eff + 1e-3 * (mult * w) / 12.0 - 1e-4 * mult;
            if (score > best) { best = score; bb = cus * mult; bw = w; }
        }
    }
    *blocks_out = bb; *waves_out = bw;
}

template <typename WT, typename XT, int R, int U, int PRO, int MAXT, bool SMALL, int EPI, bool FUSE_AR = false>
static int launch_gemv_ke(Launcher &L, const GemvArgs &a, int blocks, int waves, size_t lds) {
    auto kern = gemv_kernel<WT, XT, R, U, PRO, MAXT, SMALL, EPI, FUSE_AR>;
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    double bytes = (double)a.N * a.K * sizeof(WT);
    char tag[32];
    snprintf(tag, sizeof tag, "%dx%d%s%s", a.N, a.K, PRO == PRO_NORM ? ",norm" : "", a.epi == EPI_GATEUP ? ",glu" : (a.epi == EPI_QKV_ROPE ? ",rope" : ""));
    Launcher LL = L; LL.tag = tag;
    return LL.launch(KC_GEMV, bytes, 2.0 * a.N * a.K, kern, dim3((unsigned)blocks), dim3((unsigned)waves * 64), lds, a);
}

template <typename WT, typename XT, int R, int U, int PRO, int MAXT, bool SMALL>
static int launch_gemv_k(Launcher &L, const GemvArgs &a, int blocks, int waves, size_t lds) {
    if (a.epi == EPI_GATEUP) return launch_gemv_ke<WT, XT, R, U, PRO, MAXT, SMALL, EPI_GATEUP>(L, a, blocks, waves, lds);
    if (a.epi == EPI_QKV_ROPE) return launch_gemv_ke<WT, XT, R, U, PRO, MAXT, SMALL, EPI_QKV_ROPE>(L, a, blocks, waves, lds);
    if (a.ll) return launch_gemv_ke<WT, XT, R, U, PRO, MAXT, SMALL, EPI_F32, true>(L, a, blocks, waves, lds);
    return launch_gemv_ke<WT, XT, R, U, PRO, MAXT, SMALL, EPI_F32>(L, a, blocks, waves, lds);
}

template <typename WT, typename XT, int R, int U, int PRO>
static int launch_gemv_t(Launcher &L, const GemvArgs &a) {
    const int64_t N = a.N, K = a.K;
    size_t lds = ((size_t)K * sizeof(XT) + 15) & ~(size_t)15;
    const int64_t ngroups = (N + R - 1) / R;
    int blocks = 1, waves = 4;
    pick_geometry(ngroups, lds, &blocks, &waves);
    if (a.amax && blocks + 1 > kMaxArgmaxCand) {           // (callers ask gemv_leaves_candidates first; a tuned grid may still get here)
        FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv: %d workgroups exceed the ArgMax candidate buffer (gemv_leaves_candidates not consulted)", blocks);
    }
    if (PRO == PRO_NORM && (int64_t)waves * 64 * 3 * 8 < K) waves = (int)((K + 64 * 3 * 8 - 1) / (64 * 3 * 8));   // staging capacity
    if (waves < 4) waves = 4;
    const int allow_small = tune(TK_GEMV_SMALL);
    const bool small = allow_small && sizeof(WT) == 2 && ngroups <= (int64_t)blocks * waves && (K >> 3) % (64 * U) == 0 &&
                       (K >> 3) <= 2 * 64 * U && (PRO != PRO_NORM || (int64_t)waves * 64 * 8 >= K);
    if (small) return launch_gemv_k<WT, XT, R, U, PRO, 768, true>(L, a, blocks, waves, lds);
    return launch_gemv_k<WT, XT, R, U, PRO, 768, false>(L, a, blocks, waves, lds);
}

template <typename WT, typename XT, int PRO>
static int launch_gemv_ru(Launcher &L, const GemvArgs &a) {
    int R = tune(TK_GEMV_R), U = tune(TK_GEMV_U);
    // default (no FL_GEMV_U / fl_tune): four 1-KiB chunks in flight per row, two for the plain and RoPE epilogues at
    // K <= 4096 (Mistral-7B: o_proj 7.96 -> 7.48 us, lm_head 43.8 -> 42.0, QKV 11.95 -> 11.82; down_proj at K = 14336
    // 19.9 vs 20.9 and gate/up 38.2 vs 38.1 stay on four)
    if (U <= 0) {
        // fp32 weights: two 2-KiB chunks per row -- the four-chunk instantiations spill 20-48 VGPRs under the 168-register cap of a
        // 768-thread workgroup (Mistral-7B fp32 decode: gate/up 5.11 -> 5.41 TB/s, down_proj 5.03 -> 5.69, step 6.12 -> 5.78 ms)
        U = 2;
        // bf16 (two register buffers of the stream, dot2 arithmetic: 152-166 VGPRs at U = 8, no spills): 8 KiB per row group
        // and request round from K = 4096 (Mistral-7B: down_proj 19.9 -> 19.05 us, the others equal or better), 4 below
        if (sizeof(WT) == 2) U = (a.K >> 3) >= 512 ? 8 : 4;
        // the QKV projection is short enough for every wave to own ONE row group: with U = K / 512 its whole share is one block,
        // requested before the norm prologue runs (the SMALL form): Mistral-7B 11.8 -> 10.5 us
        // (bf16 only: with fp32 weights eight chunks per row are 128 VGPRs of stream buffers -- the fp32 mode's QKV ran at 1.4 TB/s from scratch spills)
        if (sizeof(WT) == 2 && R == 2 && a.epi == EPI_QKV_ROPE && a.pro == PRO_NORM && (a.K == 4096 || a.K == 3584)) U = a.K / 512;
    }
    if (R == 4 && U == 2) return launch_gemv_t<WT, XT, 4, 2, PRO>(L, a);
    if (R == 2 && U == 7) return launch_gemv_t<WT, XT, 2, 7, PRO>(L, a);
    if (R == 2 && U == 8) return launch_gemv_t<WT, XT, 2, 8, PRO>(L, a);
    if (R == 2 && U == 2) return launch_gemv_t<WT, XT, 2, 2, PRO>(L, a);
    return launch_gemv_t<WT, XT, 2, 4, PRO>(L, a);
}

// Would this launch's grid fit the ArgMax candidate buffer (GemvArgs::amax)?  A tuned grid (fl_tune "gemv_blocks") or a
// device with >= 512 CUs may not: the caller then leaves amax null and token selection scans the logits instead.
bool gemv_leaves_candidates(int dtype, const GemvArgs &a) {
    int R = tune(TK_GEMV_R);
    if (R != 4) R = 2;                                       // (the instantiations launch_gemv_ru picks from)
    const size_t es = dtype == FL_DTYPE_BF16 ? 2 : 4;
    int blocks = 1, waves = 4;
    pick_geometry(((int64_t)a.N + R - 1) / R, ((size_t)a.K * es + 15) & ~(size_t)15, &blocks, &waves);
    return blocks + 1 <= kMaxArgmaxCand;
}

int64_t gemv_owner_chunk(int dtype, int64_t N, int64_t K) {
    int R = tune(TK_GEMV_R);
    if (R != 4) R = 2;
    const size_t es = dtype == FL_DTYPE_BF16 ? 2 : 4;
    int blocks = 1, waves = 4;
    pick_geometry((N + R - 1) / R, ((size_t)K * es + 15) & ~(size_t)15, &blocks, &waves);
    if (waves < 4) waves = 4;
    return (int64_t)waves * R * K * (int64_t)es;
}

int launch_gemv(Launcher &L, int dtype, const GemvArgs &a) {
    if (a.N <= 0 || a.K <= 0) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv: bad shape");
    if (!gemv_supported(dtype, a.N, a.K)) FL_FAIL(FL_ERR_UNSUPPORTED, "launch_gemv: K=%d unsupported", a.K);
    if (a.pro == PRO_NORM && !gemv_norm_supported(dtype, a.N, a.K)) FL_FAIL(FL_ERR_UNSUPPORTED, "fused norm needs K <= 6144");
    if (a.epi == EPI_GATEUP && a.N % 32) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gate/up matrix rows must be a multiple of 32");
    if (a.ll && (a.epi != EPI_F32 || a.bias || a.ll_slot <= 0)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "fused all-reduce: plain fp32 epilogue without bias only");
    if (a.epi == EPI_QKV_ROPE && (a.d <= 0 || a.d % 2 || a.N != (a.H + 2 * a.Hkv) * a.d)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "bad qkv shape");
    if (dtype == FL_DTYPE_BF16)
        return a.pro == PRO_NORM ? launch_gemv_ru<bf16_t, bf16_t, PRO_NORM>(L, a) : launch_gemv_ru<bf16_t, bf16_t, PRO_X>(L, a);
    return a.pro == PRO_NORM ? launch_gemv_ru<float, float, PRO_NORM>(L, a) : launch_gemv_ru<float, float, PRO_X>(L, a);
}

}  // namespace fl
