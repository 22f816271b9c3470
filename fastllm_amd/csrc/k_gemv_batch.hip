// k_gemv_batch.hip -- the decode weight-streaming kernel for B <= 8 independent sequences at once:
// Y[b] = W[N,K] . x[b].  Row N4 of the scope table ("batching across concurrent streams": the reference
// runs every stream as its own single-sequence loop, mod.rs:137-238, so each stream pays for the whole
// weight read; here one read of W serves all B streams).
//
// Same streaming scheme as k_gemv.hip: a wave owns R = 2 rows at a time, reads them 16 B per lane,
// non-temporal, straight to VGPRs as ONE double-buffered stream of U = 4 chunk blocks with unconditional loads (counted
// vmcnt waits; round 3: one and two streams 3.03 / 3.10 -> 2.93 / 3.01 ms per step on Mistral-7B), fp32
// accumulate, 64-lane butterfly.  What changes is the activation side: the B vectors sit in LDS as
// [B][Ks] and every weight chunk is multiplied with all of them (8 unpacks + 8*B FMAs per 16-byte load,
// still far below the VALU rate needed to keep up with HBM at B = 8).  The per-lane accumulation order
// over K and the butterfly are those of the single-sequence kernel, so a row's dot product is the same
// number whichever kernel computed it.
//
// LDS holds B * Ks * 2 bytes, so a long K is cut into `nks` slices handled by different workgroups
// (blockIdx.y); the slices write separate fp32 slabs [nks][B][N] that the next norm prologue sums in a
// fixed order -- only the row-parallel EPI_F32 products (down_proj: K = 14336 / 18944) need it.
//
// Prologue PRO_NORM per sequence: v = x_in[b] + sum of delta slabs (or the embedding row of the
// sequence's current token), x' = bf16(v * w) staged, 1/rms applied to the accumulator (as k_gemv.hip).
// Epilogues: EPI_F32 (+bias), EPI_GATEUP (silu(gate)*up), EPI_QKV_ROPE with each sequence's own RoPE
// position and KV slot from its device step state, written into its own cache.
#include <stdlib.h>

#include <atomic>

#include "kernels.h"

namespace fl {

constexpr int kBR = 2, kBThreads = 512;

// kBU: 1-KiB chunks per row and block: 8 from K = 4096 (16 KiB per request round and wave, as the single-sequence kernel), else 4
template <int NB, int PRO, int EPI, int kBU>
__global__ __launch_bounds__(kBThreads) void gemv_batch_kernel(const GemvBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float red[kBThreads / 64][NB];
    bf16_t *xs = reinterpret_cast<bf16_t *>(lds_raw);                 // [NB][Ks]
    const bf16_t *__restrict__ W = reinterpret_cast<const bf16_t *>(a.W);
    const int N = a.N, K = a.K, B = a.B;
    constexpr int epi = EPI;                          // compile-time, like gemv_kernel's: the other epilogues' operands cost no registers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nthr = kBThreads, nwv = kBThreads / 64;
    const int ks = blockIdx.y;
    // K slice of this workgroup, in 8-element chunks; slices are multiples of 64 chunks except the last
    const int nchunk_all = K >> 3;
    const int per = ((nchunk_all + a.nks - 1) / a.nks + 63) & ~63;
    const int cs0 = min(nchunk_all, ks * per), cs1 = min(nchunk_all, cs0 + per);
    const int nchunk = cs1 - cs0;                                      // chunks of the slice
    const int Ks = per * 8;                                            // LDS row stride (elements)
    const int half = a.d >> 1;
    const int ngroups = (N + kBR - 1) / kBR;
    const int nw = gridDim.x * nwv;

    auto row_of = [&](int g, int r) -> int {
        if (epi == EPI_GATEUP) { int q = g; return (q >> 4) * 32 + (q & 15) + (r << 4); }
        if (epi == EPI_QKV_ROPE) { int q = g; int hd = q / half, j = q - hd * half; return hd * a.d + j + r * half; }
        return g * kBR + r;
    };
    typedef uint4v Buf[kBR][kBU];
    // The wave's (row group, K block) items form ONE stream, requested a block ahead into two register buffers -- the form of the
    // single-sequence kernel (k_gemv.hip, round 2), ported in round 3: while a block is multiplied the next one is in flight, also
    // across row groups and their epilogues.  Every load of the stream is unconditional (past the end of K a lane re-reads the last
    // chunk and its x is zeroed; a wave without work reads the last group's first block once): control flow around loads makes
    // hipcc fall back from counted vmcnt waits to vmcnt(0).  The first block is requested before the activations are staged.
    Buf pre;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int gw_u = blockIdx.x * nwv + wave_u;
    const int nb = max(1, (nchunk + 64 * kBU - 1) / (64 * kBU));                       // K blocks per row group (the last may be partial)
    const int n_items = gw_u < ngroups ? (ngroups - gw_u + nw - 1) / nw * nb : 0;
    const bool ragged = nchunk % (64 * kBU) != 0;
    int lg = gw_u, lb = 0;                                                             // load stream: next item = (row group, block)
    auto load_next = [&](Buf &buf) {
        const int g = min(lg, ngroups - 1);
#pragma unroll
        for (int r = 0; r < kBR; r++) {
            const int row = row_of(g, r);
            const bf16_t *wp = W + (size_t)(row < N ? row : N - 1) * K + (size_t)cs0 * 8;
#pragma unroll
            for (int u = 0; u < kBU; u++)
                buf[r][u] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(wp + (size_t)min(lane + 64 * (kBU * lb + u), max(nchunk, 1) - 1) * 8));
        }
        if (++lb == nb) { lb = 0; lg += nw; }
    };
    auto prefetch = [&]() { load_next(pre); };
    float inv_m[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) inv_m[b] = 1.0f;
    if constexpr (PRO == PRO_NORM) {
        prefetch();
        float ss[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) ss[b] = 0.f;
        // one pass over K in steps of nthr chunks; the loads of all B rows of a step are issued together
        for (int c = tid; c < nchunk_all; c += nthr) {
            float v[NB][8];
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (b < B) {
                    if (a.embed) load8(reinterpret_cast<const bf16_t *>(a.embed) + (size_t)a.seqs[b].st->token * K + c * 8, v[b]);
                    else load8(a.x_in + (size_t)b * K + c * 8, v[b]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) v[b][j] = 0.f;
                }
            }
            if (a.delta) {
                for (int s = 0; s < a.n_slab; s++) {
#pragma unroll
                    for (int b = 0; b < NB; b++) {
                        if (b >= B) continue;
                        float dl[8];
                        load8(a.delta + (size_t)s * a.slab_stride + (size_t)b * K + c * 8, dl);
#pragma unroll
                        for (int j = 0; j < 8; j++) v[b][j] += dl[j];
                    }
                }
            }
            const bool mine = c >= cs0 && c < cs1;
            float wn[8];
            if (mine) load8(a.norm_w + c * 8, wn);
#pragma unroll
            for (int b = 0; b < NB; b++) {
#pragma unroll
                for (int j = 0; j < 8; j++) ss[b] = fmaf(v[b][j], v[b][j], ss[b]);
                if (mine) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) o[j] = v[b][j] * wn[j];
                    store8(xs + (size_t)b * Ks + (size_t)(c - cs0) * 8, o);
                }
                if (b < B && blockIdx.x == 0 && ks == 0 && a.x_out) store8(a.x_out + (size_t)b * K + c * 8, v[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) ss[b] = wave_sum(ss[b]);
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < NB; b++) red[wave][b] = ss[b];
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; b++) {
            float s = 0.f;
            for (int w = 0; w < nwv; w++) s += red[w][b];
            inv_m[b] = 1.0f / sqrtf(s / (float)K + a.eps);             // candle rms_norm (App. A.2)
        }
    } else {
        const bf16_t *__restrict__ x = reinterpret_cast<const bf16_t *>(a.x);
        prefetch();
#pragma unroll
        for (int b = 0; b < NB; b++) {
            for (int c = tid; c < nchunk; c += nthr) {
                uint4v v = uint4v{0, 0, 0, 0};
                if (b < B) v = *reinterpret_cast<const uint4v *>(x + (size_t)b * K + (size_t)(cs0 + c) * 8);
                *reinterpret_cast<uint4v *>(xs + (size_t)b * Ks + (size_t)c * 8) = v;
            }
        }
        __syncthreads();
    }

    float acc[kBR][NB];
#pragma unroll
    for (int r = 0; r < kBR; r++)
#pragma unroll
        for (int b = 0; b < NB; b++) acc[r][b] = 0.f;

    auto fma_block = [&](const Buf &buf, int c0) {                     // bf16 x bf16 on v_dot2c_f32_bf16, as the single-sequence GEMV
#pragma unroll
        for (int u = 0; u < kBU; u++) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const uint4v xr = *reinterpret_cast<const uint4v *>(xs + (size_t)b * Ks + (size_t)(c0 + 64 * u) * 8);
#pragma unroll
                for (int r = 0; r < kBR; r++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[r][b] = dot2c_bf16(buf[r][u][j], xr[j], acc[r][b]);
            }
        }
#pragma unroll
        for (int r = 0; r < kBR; r++)
#pragma unroll
            for (int b = 0; b < NB; b++) dot2c_settle(acc[r][b]);
    };
    auto finish_group = [&](int g) {
        float sum[kBR][NB];
#pragma unroll
        for (int r = 0; r < kBR; r++)
#pragma unroll
            for (int b = 0; b < NB; b++) { sum[r][b] = wave_sum(acc[r][b]) * inv_m[b]; acc[r][b] = 0.f; }
        if (lane != 0) return;
        const int r0w = row_of(g, 0), r1w = row_of(g, 1);
        if (epi == EPI_GATEUP) {
            if (r1w >= N) return;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (b >= B) break;
                const float gt = sum[0][b], up = sum[1][b];
                const float act = gt / (1.0f + expf(-gt)) * up;              // candle silu(g) * u
                elem<bf16_t>::st(reinterpret_cast<bf16_t *>(a.out) + (size_t)b * (N / 2) + g, act);
            }
        } else if (epi == EPI_QKV_ROPE) {
            if (r1w >= N) return;
            const int hd = g / half, j = g - hd * half;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (b >= B) break;
                const SeqRef &sq = a.seqs[b];
                float x0 = sum[0][b], x1 = sum[1][b];
                if (a.bias) { x0 += a.bias[r0w]; x1 += a.bias[r1w]; }
                const uint32_t pos = sq.st->pos, slot = sq.st->len;
                bf16_t *dst;
                size_t stride = 1;
                if (hd < a.H + a.Hkv) {                                       // rotate-half RoPE (App. A.4)
                    const uint32_t p = pos < (uint32_t)a.max_pos ? pos : (uint32_t)a.max_pos - 1;
                    const float c = a.cos_tab[(size_t)p * half + j], s = a.sin_tab[(size_t)p * half + j];
                    float t0, t1;
                    rope_rotate(x0, x1, c, s, t0, t1);
                    x0 = t0; x1 = t1;
                    dst = hd < a.H ? reinterpret_cast<bf16_t *>(a.q_out) + ((size_t)b * a.H + hd) * a.d
                                   : reinterpret_cast<bf16_t *>(sq.k) + a.kv_layer_off * sq.seq_alloc + ((size_t)(hd - a.H) * sq.seq_alloc + slot) * a.d;
                } else {                                                      // transposed value cache [Hkv][d][seq_alloc]
                    dst = reinterpret_cast<bf16_t *>(sq.v) + a.kv_layer_off * sq.seq_alloc + (size_t)(hd - a.H - a.Hkv) * a.d * sq.seq_alloc + slot;
                    stride = (size_t)sq.seq_alloc;
                }
                elem<bf16_t>::st(dst + (size_t)j * stride, x0);
                elem<bf16_t>::st(dst + (size_t)(j + half) * stride, x1);
            }
        } else {
            float *out = reinterpret_cast<float *>(a.out) + (size_t)ks * B * N;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (b >= B) break;
                if (r0w < N) out[(size_t)b * N + r0w] = sum[0][b] + (a.bias && ks == 0 ? a.bias[r0w] : 0.f);
                if (r1w < N) out[(size_t)b * N + r1w] = sum[1][b] + (a.bias && ks == 0 ? a.bias[r1w] : 0.f);
            }
        }
    };

    int cg = gw_u, cb = 0;                                                             // consume stream
    auto consume = [&](const Buf &buf) {
        const int c0 = lane + 64 * kBU * cb;
        if (ragged && cb == nb - 1) {                                                  // wave-uniform: the partial last block of K
#pragma unroll
            for (int u = 0; u < kBU; u++) {
                const int ci = c0 + 64 * u;
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    uint4v xr = *reinterpret_cast<const uint4v *>(xs + (size_t)b * Ks + (size_t)min(ci, max(nchunk, 1) - 1) * 8);
                    if (ci >= nchunk) xr = uint4v{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int r = 0; r < kBR; r++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[r][b] = dot2c_bf16(buf[r][u][j], xr[j], acc[r][b]);
                }
            }
#pragma unroll
            for (int r = 0; r < kBR; r++)
#pragma unroll
                for (int b = 0; b < NB; b++) dot2c_settle(acc[r][b]);
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
        consume(pre);
        load_next(pre);
        consume(nxt);
    }
    if (n_items - t == 2) { load_next(nxt); consume(pre); consume(nxt); }
    else if (n_items - t == 1) consume(pre);
}

// ------------------------------------------------------------------------------- MFMA variant (B >= 3)
// At B >= 4 the VALU formulation runs out of FMA rate (8*B FMAs per 16-byte load).  Here the B vectors are
// the 16 columns of a v_mfma_f32_16x16x32_bf16 B operand (read from LDS, rows >= B are duplicates that
// are never stored) and 16 weight rows x 32 k are the A operand, loaded from HBM straight into the A
// fragment layout (lane = row l%16, k group l/16: 64 contiguous bytes per row per instruction, the next
// k step takes the adjacent 64).  A workgroup owns a unit of 32 rows (two MFMA tiles that share the
// activation fragment -- gate and up rows of the same channels, or the rotate-half partners j, j+d/2 of a
// head) and its 8 waves split K; the 8 partial tiles are summed through LDS in wave order.  VALU work per
// byte is nil, so the kernel stays on the HBM roofline for any B <= 8.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#ifndef FL_MFMA_NT
#define FL_MFMA_NT 0
#endif
#if FL_MFMA_NT
#define FL_MLOAD(p) __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(p))
#else
#define FL_MLOAD(p) (*reinterpret_cast<const uint4v *>(p))
#endif
// MODE 0: per-unit loop; 1: pipelined across units (PIPE); 2: PIPE + per-wave LDS transpose (STAGED).
// Tried and dropped (profiles/r01/README.md): K steps dealt round-robin to the waves (3.5 vs 4.05 TB/s), and
// workgroup-cooperative 32-row x 512-k stages with 1 KiB-per-row loads and a barrier per stage (3.3 TB/s).
template <int PRO, int MU, int MODE, int EPI>
__global__ __launch_bounds__(kBThreads) void gemv_batch_mfma_kernel(const GemvBatchArgs a) {
    constexpr bool PIPE = MODE == 1 || MODE == 2, STAGED = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float red[kBThreads / 64][8];
    __shared__ float inv_lds[8];
    const bf16_t *__restrict__ W = reinterpret_cast<const bf16_t *>(a.W);
    const int N = a.N, K = a.K, B = a.B;
    constexpr int epi = EPI;                          // compile-time, like gemv_kernel's: the other epilogues' operands cost no registers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nthr = kBThreads, nwv = kBThreads / 64;
    const int ks = blockIdx.y;
    const int nchunk_all = K >> 3;
    const int per = ((nchunk_all + a.nks - 1) / a.nks + 63) & ~63;
    const int cs0 = min(nchunk_all, ks * per), cs1 = min(nchunk_all, cs0 + per);
    const int nchunk = cs1 - cs0;
    const int XS = per * 8 + 8;                                        // padded LDS row stride (elements): conflict-free b128 reads
    bf16_t *xs = reinterpret_cast<bf16_t *>(lds_raw);                 // [8][XS]
    float *part = reinterpret_cast<float *>(lds_raw + (size_t)8 * XS * 2);   // [8 waves][2 tiles][16 cols][16 rows]
    float *fin = part + nwv * 2 * 256;                                 // [2][16][16] summed tiles for the paired epilogues
    unsigned char *stage_base = reinterpret_cast<unsigned char *>(fin + 2 * 256);
    unsigned char *stage = stage_base + (size_t)wave * 8192;          // STAGED: 32 rows x 256 B per wave
    const int half = a.d >> 1;
    const int m = lane & 15, kg = lane >> 4;

    // ---- stage the activations (rows >= B zero) ----
    if constexpr (PRO == PRO_NORM) {
        float ss[8];
#pragma unroll
        for (int b = 0; b < 8; b++) ss[b] = 0.f;
        for (int c = tid; c < nchunk_all; c += nthr) {
            float v[8][8];
#pragma unroll
            for (int b = 0; b < 8; b++) {
                if (b < B) {
                    if (a.embed) load8(reinterpret_cast<const bf16_t *>(a.embed) + (size_t)a.seqs[b].st->token * K + c * 8, v[b]);
                    else load8(a.x_in + (size_t)b * K + c * 8, v[b]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) v[b][j] = 0.f;
                }
            }
            if (a.delta) {
                for (int s = 0; s < a.n_slab; s++) {
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        if (b >= B) continue;
                        float dl[8];
                        load8(a.delta + (size_t)s * a.slab_stride + (size_t)b * K + c * 8, dl);
#pragma unroll
                        for (int j = 0; j < 8; j++) v[b][j] += dl[j];
                    }
                }
            }
            const bool mine = c >= cs0 && c < cs1;
            float wn[8];
            if (mine) load8(a.norm_w + c * 8, wn);
#pragma unroll
            for (int b = 0; b < 8; b++) {
#pragma unroll
                for (int j = 0; j < 8; j++) ss[b] = fmaf(v[b][j], v[b][j], ss[b]);
                if (mine) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) o[j] = v[b][j] * wn[j];
                    store8(xs + (size_t)b * XS + (size_t)(c - cs0) * 8, o);
                }
                if (b < B && blockIdx.x == 0 && ks == 0 && a.x_out) store8(a.x_out + (size_t)b * K + c * 8, v[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < 8; b++) ss[b] = wave_sum(ss[b]);
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < 8; b++) red[wave][b] = ss[b];
        }
        __syncthreads();
        if (tid < 8) {
            float s = 0.f;
            for (int w = 0; w < nwv; w++) s += red[w][tid];
            inv_lds[tid] = 1.0f / sqrtf(s / (float)K + a.eps);         // candle rms_norm (App. A.2)
        }
    } else {
        const bf16_t *__restrict__ x = reinterpret_cast<const bf16_t *>(a.x);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            for (int c = tid; c < nchunk; c += nthr) {
                uint4v v = uint4v{0, 0, 0, 0};
                if (b < B) v = *reinterpret_cast<const uint4v *>(x + (size_t)b * K + (size_t)(cs0 + c) * 8);
                *reinterpret_cast<uint4v *>(xs + (size_t)b * XS + (size_t)c * 8) = v;
            }
        }
        if (tid < 8) inv_lds[tid] = 1.0f;
    }
    __syncthreads();

    // ---- units of 32 rows: tile A = rows ra0.., tile B = rows rb0.. ----
    const int nunits = epi == EPI_QKV_ROPE ? N / 32 : (N + 31) / 32;
    const int nsteps = nchunk >> 2;                                    // 32-element k steps of the slice
    const int spw = (nsteps + nwv - 1) / nwv;
    const int s0 = min(nsteps, wave * spw), s1 = min(nsteps, s0 + spw);
    const bf16_t *xb = xs + (size_t)(m & 7) * XS + kg * 8;
    float4v ca = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
    auto unit_rows = [&](int unit, int &ra0, int &rb0) {
        if (epi == EPI_QKV_ROPE) {
            const int upr = half / 16;                                 // units per head
            const int hd = unit / upr, i = unit - hd * upr;
            ra0 = hd * a.d + 16 * i; rb0 = ra0 + half;
        } else { ra0 = unit * 32; rb0 = ra0 + 16; }
    };
    // sum the 8 waves' partial tiles in wave order, apply 1/rms, run the epilogue; clears ca / cb
    auto finish_unit = [&](int unit) {
        int ra0, rb0;
        unit_rows(unit, ra0, rb0);
        // C fragment: column n = m (sequence), rows 4*kg .. 4*kg+3  ->  part[wave][tile][n][row]
        *reinterpret_cast<float4v *>(part + ((wave * 2 + 0) * 16 + m) * 16 + 4 * kg) = ca;
        *reinterpret_cast<float4v *>(part + ((wave * 2 + 1) * 16 + m) * 16 + 4 * kg) = cb;
        ca = float4v{0.f, 0.f, 0.f, 0.f}; cb = float4v{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        const int tile = tid >> 8, n = (tid >> 4) & 15, r = tid & 15;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < nwv; w++) sum += part[((w * 2 + tile) * 16 + n) * 16 + r];
        sum *= inv_lds[n & 7];
        const int row = (tile ? rb0 : ra0) + r;
        if (epi == EPI_F32) {
            if (n < B && row < N)
                reinterpret_cast<float *>(a.out)[(size_t)ks * B * N + (size_t)n * N + row] = sum + (a.bias && ks == 0 ? a.bias[row] : 0.f);
            __syncthreads();                                           // part[] is rewritten by the next unit
            return;
        }
        if (epi == EPI_QKV_ROPE && a.bias && row < N) sum += a.bias[row];
        fin[(tile * 16 + n) * 16 + r] = sum;
        __syncthreads();
        if (tile == 0 && n < B && rb0 + r < N) {
            float x0 = fin[n * 16 + r], x1 = fin[(16 + n) * 16 + r];
            if (epi == EPI_GATEUP) {
                const float act = x0 / (1.0f + expf(-x0)) * x1;              // candle silu(g) * u
                elem<bf16_t>::st(reinterpret_cast<bf16_t *>(a.out) + (size_t)n * (N / 2) + unit * 16 + r, act);
            } else {
                const SeqRef &sq = a.seqs[n];
                const int hd = ra0 / a.d, j = ra0 - hd * a.d + r;
                const uint32_t pos = sq.st->pos, slot = sq.st->len;
                bf16_t *dst;
                size_t stride = 1;
                if (hd < a.H + a.Hkv) {                                       // rotate-half RoPE (App. A.4)
                    const uint32_t p = pos < (uint32_t)a.max_pos ? pos : (uint32_t)a.max_pos - 1;
                    const float c = a.cos_tab[(size_t)p * half + j], sn = a.sin_tab[(size_t)p * half + j];
                    float t0, t1;
                    rope_rotate(x0, x1, c, sn, t0, t1);
                    x0 = t0; x1 = t1;
                    dst = hd < a.H ? reinterpret_cast<bf16_t *>(a.q_out) + ((size_t)n * a.H + hd) * a.d
                                   : reinterpret_cast<bf16_t *>(sq.k) + a.kv_layer_off * sq.seq_alloc + ((size_t)(hd - a.H) * sq.seq_alloc + slot) * a.d;
                } else {
                    dst = reinterpret_cast<bf16_t *>(sq.v) + a.kv_layer_off * sq.seq_alloc + (size_t)(hd - a.H - a.Hkv) * a.d * sq.seq_alloc + slot;
                    stride = (size_t)sq.seq_alloc;
                }
                elem<bf16_t>::st(dst + (size_t)j * stride, x0);
                elem<bf16_t>::st(dst + (size_t)(j + half) * stride, x1);
            }
        }
        __syncthreads();
    };

    if constexpr (PIPE) {
        // Host-checked: every wave has the same number nbw >= 1 of MU-step blocks per unit.  The blocks of all
        // the workgroup's units form one flat sequence; block f+1 is requested before block f is consumed, so
        // the weight stream does not drain while a unit's tiles are reduced and written out.
        const int nbw = (s1 - s0) / MU;
        const int myunits = (int)blockIdx.x < nunits ? (nunits - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
        const int total = myunits * nbw;
        int iu = 0, ib = 0;                                            // issue cursor: unit index (of mine), block in unit
        int cu = 0, cbk = 0;                                           // compute cursor
        auto issue = [&](uint4v (&wa)[MU], uint4v (&wb)[MU]) {
            int ra0, rb0;
            unit_rows((int)blockIdx.x + iu * (int)gridDim.x, ra0, rb0);
            if constexpr (STAGED) {
                // MU == 4: a block is 32 rows x 128 k.  Instruction i (wa[u] = 2u, wb[u] = 2u+1) reads rows 4i..4i+3
                // of the unit, 256 contiguous bytes (two full cache lines) per row: lane = (row & 3) * 16 + chunk.
                // 64 bytes per row per instruction -- the MFMA A-fragment pattern -- costs ~17 % of the stream rate.
                const size_t koff = (size_t)cs0 * 8 + (size_t)(s0 + ib * MU) * 32 + (size_t)(lane & 15) * 8;
#pragma unroll
                for (int u = 0; u < MU; u++) {
                    const int r0 = 8 * u + (lane >> 4), r1 = r0 + 4;                  // unit rows 0..31 (16.. = tile B)
                    const int g0 = (r0 < 16 ? ra0 + r0 : rb0 + r0 - 16), g1 = (r1 < 16 ? ra0 + r1 : rb0 + r1 - 16);
                    wa[u] = FL_MLOAD(W + (size_t)min(g0, N - 1) * K + koff);
                    wb[u] = FL_MLOAD(W + (size_t)min(g1, N - 1) * K + koff);
                }
            } else {
                const size_t koff = (size_t)cs0 * 8 + kg * 8 + (size_t)(s0 + ib * MU) * 32;
                const bf16_t *pa = W + (size_t)min(ra0 + m, N - 1) * K + koff;
                const bf16_t *pb = W + (size_t)min(rb0 + m, N - 1) * K + koff;
#pragma unroll
                for (int u = 0; u < MU; u++) { wa[u] = FL_MLOAD(pa + (size_t)u * 32); wb[u] = FL_MLOAD(pb + (size_t)u * 32); }
            }
            if (++ib == nbw) { ib = 0; iu++; }
        };
        auto compute = [&](const uint4v (&wa)[MU], const uint4v (&wb)[MU]) {
            if constexpr (STAGED) {
                // registers (row-major pieces) -> this wave's LDS stage, chunk c of row r at position c ^ (r & 15);
                // back out as A fragments: lane (m, kg) of k step t takes row m, chunk 4t + kg.  Wave-private: no barrier.
                const int c = lane & 15;
#pragma unroll
                for (int u = 0; u < MU; u++) {
                    const int r0 = 8 * u + (lane >> 4), r1 = r0 + 4;
                    *reinterpret_cast<uint4v *>(stage + r0 * 256 + ((c ^ (r0 & 15)) << 4)) = wa[u];
                    *reinterpret_cast<uint4v *>(stage + r1 * 256 + ((c ^ (r1 & 15)) << 4)) = wb[u];
                }
                const int sb = s0 + cbk * MU;
#pragma unroll
                for (int t = 0; t < MU; t++) {
                    const bf16x8_t fa = *reinterpret_cast<const bf16x8_t *>(stage + m * 256 + (((4 * t + kg) ^ m) << 4));
                    const bf16x8_t fb = *reinterpret_cast<const bf16x8_t *>(stage + (16 + m) * 256 + (((4 * t + kg) ^ m) << 4));
                    const bf16x8_t xf = *reinterpret_cast<const bf16x8_t *>(xb + (size_t)(sb + t) * 32);
                    ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, xf, ca, 0, 0, 0);
                    cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, xf, cb, 0, 0, 0);
                }
            } else {
                const int sb = s0 + cbk * MU;
#pragma unroll
                for (int u = 0; u < MU; u++) {
                    const bf16x8_t xf = *reinterpret_cast<const bf16x8_t *>(xb + (size_t)(sb + u) * 32);
                    ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa[u]), xf, ca, 0, 0, 0);
                    cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb[u]), xf, cb, 0, 0, 0);
                }
            }
            if (++cbk == nbw) { cbk = 0; finish_unit((int)blockIdx.x + cu * (int)gridDim.x); cu++; }
        };
        uint4v a0[MU], b0[MU], a1[MU], b1[MU];
        if (total > 0) issue(a0, b0);
#pragma nounroll
        for (int f = 0; f < total; f += 2) {
            if (f + 1 < total) issue(a1, b1);
            compute(a0, b0);
            if (f + 2 < total) issue(a0, b0);
            if (f + 1 < total) compute(a1, b1);
        }
    } else {
        for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
            int ra0, rb0;
            unit_rows(unit, ra0, rb0);
            const bf16_t *pa = W + (size_t)min(ra0 + m, N - 1) * K + (size_t)cs0 * 8 + kg * 8;
            const bf16_t *pb = W + (size_t)min(rb0 + m, N - 1) * K + (size_t)cs0 * 8 + kg * 8;
            int s = s0;
#pragma nounroll
            for (; s + MU <= s1; s += MU) {                            // straight-line block, counted waits
                uint4v wa[MU], wb[MU];
#pragma unroll
                for (int u = 0; u < MU; u++) { wa[u] = FL_MLOAD(pa + (size_t)(s + u) * 32); wb[u] = FL_MLOAD(pb + (size_t)(s + u) * 32); }
#pragma unroll
                for (int u = 0; u < MU; u++) {
                    const bf16x8_t xf = *reinterpret_cast<const bf16x8_t *>(xb + (size_t)(s + u) * 32);
                    ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa[u]), xf, ca, 0, 0, 0);
                    cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb[u]), xf, cb, 0, 0, 0);
                }
            }
#pragma nounroll
            for (; s < s1; s++) {
                const uint4v wa = FL_MLOAD(pa + (size_t)s * 32), wb = FL_MLOAD(pb + (size_t)s * 32);
                const bf16x8_t xf = *reinterpret_cast<const bf16x8_t *>(xb + (size_t)s * 32);
                ca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), xf, ca, 0, 0, 0);
                cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb), xf, cb, 0, 0, 0);
            }
            finish_unit(unit);
        }
    }
}

static int cu_count_b() {
    static int n = 0;
    if (!n) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// K slices: enough that B vectors of one slice fit in LDS (bf16), and -- for the fp32 epilogue, whose
// slabs the consumer sums anyway -- enough that a short matrix (N / 32 units < CUs) still covers the chip
int gemv_batch_ksplit(int B, int64_t K, int64_t N, int epi) {
    // N <= 0: the LDS minimum only (a consumer that cannot sum slabs, e.g. the logits)
    const int NB = B <= 2 ? 2 : (B <= 4 ? 4 : 8);
    const size_t budget = 132 * 1024;           // the MFMA variant adds 18 KB of partial tiles + row padding
    int nks = 1;
    while (true) {
        const int64_t per = (((K / 8) + nks - 1) / nks + 63) & ~(int64_t)63;
        if ((size_t)NB * per * 8 * 2 <= budget) break;
        nks++;
    }
    if (epi == EPI_F32 && B >= 3 && N > 0) {
        const int64_t units = (N + 31) / 32;
        while (units * nks < cu_count_b() && nks < 4 && (K / 8 + nks) / (nks + 1) >= 128) nks++;
    }
    return nks;
}

template <int NB, int PRO, int EPI, int U>
static int launch_gemv_batch_u(Launcher &L, const GemvBatchArgs &a) {
    auto kern = gemv_batch_kernel<NB, PRO, EPI, U>;
    const int64_t per = (((a.K / 8) + a.nks - 1) / a.nks + 63) & ~(int64_t)63;
    const size_t lds = (size_t)NB * per * 8 * 2;
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    const int64_t ngroups = (a.N + kBR - 1) / kBR;
    const int nwv = kBThreads / 64;
    int blocks = (int)std::min<int64_t>((ngroups + nwv - 1) / nwv, std::max(1, cu_count_b() / a.nks));
    char tag[32];
    snprintf(tag, sizeof tag, "b%d:%dx%d%s%s", a.B, a.N, a.K, PRO == PRO_NORM ? ",norm" : "", a.epi == EPI_GATEUP ? ",glu" : (a.epi == EPI_QKV_ROPE ? ",rope" : ""));
    Launcher LL = L; LL.tag = tag;
    return LL.launch(KC_GEMV, (double)a.N * a.K * 2, 2.0 * a.N * a.K * a.B, kern, dim3((unsigned)blocks, (unsigned)a.nks), dim3(kBThreads), lds, a);
}

template <int NB, int PRO, int EPI>
static int launch_gemv_batch_e(Launcher &L, const GemvBatchArgs &a) {
    // FL_BATCH_U=8: 16-KiB blocks as the single-sequence kernel uses from K = 4096 -- measured no better here (one and two streams:
    // 2.97 / 3.04 ms per step against 2.93 / 3.01 with 8-KiB blocks), so 4 is the default; NB <= 2 only (registers)
    const int u_env = tune(TK_BATCH_U);
    if constexpr (NB <= 2) {
        if (u_env == 8) return launch_gemv_batch_u<NB, PRO, EPI, 8>(L, a);
    }
    return launch_gemv_batch_u<NB, PRO, EPI, 4>(L, a);
}

template <int NB, int PRO>
static int launch_gemv_batch_t(Launcher &L, const GemvBatchArgs &a) {
    if (a.epi == EPI_GATEUP) return launch_gemv_batch_e<NB, PRO, EPI_GATEUP>(L, a);
    if (a.epi == EPI_QKV_ROPE) return launch_gemv_batch_e<NB, PRO, EPI_QKV_ROPE>(L, a);
    return launch_gemv_batch_e<NB, PRO, EPI_F32>(L, a);
}

template <int PRO, int MU, int MODE, int EPI>
static int launch_gemv_batch_mfma_e(Launcher &L, const GemvBatchArgs &a) {
    auto kern = gemv_batch_mfma_kernel<PRO, MU, MODE, EPI>;
    const int64_t per = (((a.K / 8) + a.nks - 1) / a.nks + 63) & ~(int64_t)63;
    const size_t lds = (size_t)8 * (per * 8 + 8) * 2 + (size_t)(8 * 2 * 256 + 2 * 256) * 4 + (MODE >= 2 ? 8 * 8192 : 0);
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    const int64_t nunits = (a.N + 31) / 32;
    const int blocks = (int)std::min<int64_t>(nunits, std::max(1, cu_count_b() / a.nks));
    char tag[32];
    snprintf(tag, sizeof tag, "b%dm:%dx%d%s%s", a.B, a.N, a.K, PRO == PRO_NORM ? ",norm" : "", a.epi == EPI_GATEUP ? ",glu" : (a.epi == EPI_QKV_ROPE ? ",rope" : ""));
    Launcher LL = L; LL.tag = tag;
    return LL.launch(KC_GEMV, (double)a.N * a.K * 2, 2.0 * a.N * a.K * a.B, kern, dim3((unsigned)blocks, (unsigned)a.nks), dim3(kBThreads), lds, a);
}

template <int PRO, int MU, int MODE>
static int launch_gemv_batch_mfma_k(Launcher &L, const GemvBatchArgs &a) {
    if (a.epi == EPI_GATEUP) return launch_gemv_batch_mfma_e<PRO, MU, MODE, EPI_GATEUP>(L, a);
    if (a.epi == EPI_QKV_ROPE) return launch_gemv_batch_mfma_e<PRO, MU, MODE, EPI_QKV_ROPE>(L, a);
    return launch_gemv_batch_mfma_e<PRO, MU, MODE, EPI_F32>(L, a);
}

template <int PRO>
static int launch_gemv_batch_mfma_t(Launcher &L, const GemvBatchArgs &a) {
    // pipelined variant: every K slice must give each of the 8 waves the same whole number of MU-step blocks
    const int64_t nchunk_all = a.K / 8;
    const int64_t per = ((nchunk_all + a.nks - 1) / a.nks + 63) & ~(int64_t)63;
    int mu = 4;
    bool pipe = tune(TK_BATCH_MODE) != 0;
    for (int k = 0; k < a.nks; k++) {
        const int64_t c0 = std::min(nchunk_all, k * per), c1 = std::min(nchunk_all, c0 + per);
        const int64_t steps = (c1 - c0) / 4;
        if (steps == 0 || steps % 8) { pipe = false; break; }
        while (mu > 1 && (steps / 8) % mu) mu /= 2;
    }
    // staging variants need room for 64 KB next to the activations
    const size_t lds_staged = (size_t)8 * (per * 8 + 8) * 2 + (size_t)(8 * 2 * 256 + 2 * 256) * 4 + 8 * 8192;
    const int mode_env = tune(TK_BATCH_MODE);
    if (pipe) {
        if (mu == 4 && mode_env >= 2 && lds_staged <= 160 * 1024 - 1024) return launch_gemv_batch_mfma_k<PRO, 4, 2>(L, a);
        if (mu == 4) return launch_gemv_batch_mfma_k<PRO, 4, 1>(L, a);
        if (mu == 2) return launch_gemv_batch_mfma_k<PRO, 2, 1>(L, a);
        return launch_gemv_batch_mfma_k<PRO, 1, 1>(L, a);
    }
    return launch_gemv_batch_mfma_k<PRO, 4, 0>(L, a);
}

int launch_gemv_batch(Launcher &L, const GemvBatchArgs &a) {
    if (a.N <= 0 || a.K <= 0 || a.K % 8 || a.B < 1 || a.B > 8) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv_batch: bad shape / batch");
    if (a.nks < 1 || a.nks < gemv_batch_ksplit(a.B, a.K, 0, a.epi)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv_batch: too few K slices for LDS");
    if (a.nks > 1 && a.epi != EPI_F32) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv_batch: K slices only for the fp32 epilogue");
    if (a.epi == EPI_GATEUP && a.N % 32) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gate/up matrix rows must be a multiple of 32");
    if (a.epi == EPI_QKV_ROPE && (a.d <= 0 || a.d % 2 || a.N != (a.H + 2 * a.Hkv) * a.d || !a.seqs)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "bad qkv shape");
    const int mfma_min = tune(TK_BATCH_MFMA_MIN);
    const bool mfma_ok = a.K % 32 == 0 && (a.epi != EPI_QKV_ROPE || (a.d % 32 == 0)) && (a.epi != EPI_GATEUP || a.N % 32 == 0);
    if (a.B >= mfma_min && mfma_ok)
        return a.pro == PRO_NORM ? launch_gemv_batch_mfma_t<PRO_NORM>(L, a) : launch_gemv_batch_mfma_t<PRO_X>(L, a);
#define FL_GO(NBV)                                                                             \
    return a.pro == PRO_NORM ? launch_gemv_batch_t<NBV, PRO_NORM>(L, a) : launch_gemv_batch_t<NBV, PRO_X>(L, a);
    if (a.B <= 2) { FL_GO(2) }
    if (a.B <= 4) { FL_GO(4) }
    FL_GO(8)
#undef FL_GO
}

}  // namespace fl
