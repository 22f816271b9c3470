// k_engine.hip -- the persistent decode engine: a CHAIN of weight-streaming projections of one decode step in ONE launch
// (llama.rs:147-149 / mistral.rs:223-226 / qwen.rs:142: the T = 1 forward every generated token pays, mod.rs:446-451).
//
// Why: a 1.1B decode layer is five launches of 4-9 us each, and ~half of each is not streaming -- the boundary, the first HBM
// round trip (ramp), the drain, the tail of the slowest workgroup -- so HBM idles for half the step (TinyLlama at
// 0.39 of roofline, a tp = 8 rank of Mistral-7B on the same 4-8 us floor per launch).  Here one workgroup per CU stays resident
// across o_proj -> gate/up -> down_proj -> (the NEXT layer's QKV projection | lm_head):
//   * 8 STREAMER waves per CU (ENG_STREAM_WAVES, kernels.h) run the same double-buffered register stream as k_gemv.hip (16 B per lane non-temporal loads,
//     v_dot2c_f32_bf16, counted vmcnt) over a static, per-CU-balanced share of every op's rows; a wave that has finished its
//     rows of op k requests its first block of op k+1 (8 KiB per wave, 64 KB per CU; FL_ENGINE_PF blocks) BEFORE it waits for op k+1's
//     input, so the weight stream runs on across the dependency (cdna guide 5.6 "prefetch-credit");
//   * the op's output vector crosses the chip as 8-byte {value, tag} GRANULES (guide Guideline 16 R2: the data is the flag --
//     one relaxed agent-scope store, no fence, no counter): fp32 deltas for o_proj / down_proj, packed bf16 pairs for silu(g)*u;
//   * 4 GATHERER waves per CU (ENG_GATHER_WAVES; no weight loads of their own, so their polls do not queue behind a refill burst) sweep the
//     whole vector once the CU's own streamers are done, apply residual add + RMSNorm weight (the fused K2/K9 prologue of
//     k_gemv.hip: W.(v/m*w) = (1/m).W.(v*w)), keep the fp32 residual stream in LDS and hand x to the streamers through LDS behind
//     one workgroup barrier per op.
// Tags are (token epoch << 8 | launch, edge): unique per use, so nothing is re-zeroed between launches; the epoch word is
// advanced by the step's select_advance launch.  Every wait is bounded (StepState::error, then the launch drains).
// Attention stays its own launch between two engine launches (stage 1): per layer 2 launches instead of 5.
//
// Residency: the grid is one workgroup per CU and workgroups wait for each other, so all of them must be resident at once:
// FL_ENGINE=1 is for a process that has the GPU to itself -- a second stream's kernels, a batch decode or another process
// on the card turn lost co-residency into a bounded-wait FL_ERR_HIP (never a hang; the bound is read per call, FL_ENGINE_TIMEOUT_MS).
// EXPERIMENTAL build only (Makefile): it measured 5 % slower than the launches it replaces.
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "comm_ll.h"
#include "kernels.h"

namespace fl {
#ifdef FL_EXPERIMENTAL


typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int E_NG = ENG_GATHER_WAVES, E_NS = ENG_STREAM_WAVES, E_THREADS = (E_NG + E_NS) * 64;
constexpr int E_PF_WAVES = 8;                        // streamer waves per CU that request their first block ahead of the op's input
constexpr int E_MISC_BYTES = 512;                  // LDS words ahead of the vectors: done counter, abort, ss partials, ArgMax candidates
constexpr int E_MAXG = 8;                         // granules a gatherer lane requests per sweep (E_MAXG * 128 = 4096 of an edge)

struct RawW { uint4v v; };

__device__ inline unsigned long long ld_granule(const unsigned long long *p) {
    return __hip_atomic_load((const gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // global_load_dwordx2 sc1
}
__device__ inline void st_granule(unsigned long long *p, uint32_t tag, uint32_t value) {
    __hip_atomic_store((gu64 *)p, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline uint32_t lds_load_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct EngCtx {
    unsigned char *lds;
    uint32_t *misc;            // [0] streamer waves done (monotonic), [1] abort, [8..32) ArgMax cv / ci, [32..48) ss partials [2 parities][8]
    bf16_t *xs0, *xs1;         // x of even / odd ops
    float *res;
    int cu, ncu, wave, lane;
    uint32_t ebase;
    unsigned long long *stamp;   // diagnostics (null: off)
};

// ---- gatherer side: bring op `o`'s input vector into xs[o & 1] ------------------------------------------------------
__device__ inline bool eng_wait_local(const EngArgs &a, EngCtx &c, uint32_t want) {
    const long long t0 = wall_clock64();
    while (lds_load_u32(c.misc + 0) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (lds_load_u32(c.misc + 1)) return false;
        if (wall_clock64() - t0 > a.timeout_ticks) return false;
    }
    return true;
}

__device__ inline void eng_go(EngCtx &c) {                 // one add per gatherer wave and op: "the streamers may request op o's first block"
    if (c.lane == 0) __hip_atomic_fetch_add(c.misc + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One pass requests EVERY granule of a lane's share at once (one round trip per pass, whatever the vector's length); a
// further pass re-reads only what was not there yet.  false: gave up (error word set, the launch drains).
template <bool HALF, typename F>    // HALF: a granule is two 4-byte {bf16 value, 16-bit tag} words (the silu(g)*u edge), else {fp32 value, 32-bit tag}
__device__ inline bool eng_sweep(const EngArgs &a, EngCtx &c, const unsigned long long *edge, int n, int base, uint32_t tag, int gl, bool signal, F &&take) {
    const long long t0 = wall_clock64();
    unsigned long long g[E_MAXG];
    unsigned missing = 0;
#pragma unroll
    for (int k = 0; k < E_MAXG; k++) if (base + k * (E_NG * 64) + gl < n) missing |= 1u << k;
    for (;;) {
#pragma unroll
        for (int k = 0; k < E_MAXG; k++)
            if (missing >> k & 1) g[k] = ld_granule(edge + base + k * (E_NG * 64) + gl);
        // the sweep's requests are in the CU's memory queue: NOW the streamers may add their prefetch behind them (a CU's loads
        // return in order: requested first, the prefetch burst delayed every sweep by its own landing time)
        if (signal) { eng_go(c); signal = false; }
#pragma unroll
        for (int k = 0; k < E_MAXG; k++)
            if (missing >> k & 1) {
                const bool ready = HALF ? ((uint32_t)(g[k] >> 16) & 0xffffu) == (tag & 0xffffu) && (uint32_t)(g[k] >> 48) == (tag & 0xffffu)
                                        : (uint32_t)(g[k] >> 32) == tag;
                if (ready) {
                    take(base + k * (E_NG * 64) + gl, k, HALF ? ((uint32_t)g[k] & 0xffffu) | ((uint32_t)(g[k] >> 32) << 16) : (uint32_t)g[k]);
                    missing &= ~(1u << k);
                }
            }
        if (__builtin_amdgcn_ballot_w64(missing != 0) == 0) return true;
        if (lds_load_u32(c.misc + 1)) return false;
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > a.timeout_ticks) {
            if (c.lane == 0) { a.st_rw->error = 0xE9610000u | (uint32_t)(tag & 0xffff); __hip_atomic_store(c.misc + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            return false;
        }
    }
}

__device__ inline void eng_gather(const EngArgs &a, const EngOp &op, int o, EngCtx &c) {
    const int gl = c.wave * 64 + c.lane;                     // 0 .. E_NG*64-1
    bf16_t *xs = (o & 1) ? c.xs1 : c.xs0;
    const int K = op.K;
    if (op.in == ENG_IN_X) {
        // plain vector from the previous launch (attention output) + this launch's residual stream
        eng_go(c);                                            // nothing to sweep: the streamers request their first block at once
        const uint4v *src = reinterpret_cast<const uint4v *>(op.x);
        for (int i = gl; i < (K >> 3); i += E_NG * 64) reinterpret_cast<uint4v *>(xs)[i] = src[i];
        if (a.x_res_in) {
            const float4v *rs = reinterpret_cast<const float4v *>(a.x_res_in);
            for (int i = gl; i < (a.h >> 2); i += E_NG * 64) reinterpret_cast<float4v *>(c.res)[i] = rs[i];
        }
        return;
    }
    const uint32_t tag = c.ebase | (uint32_t)op.tag_in;
    constexpr int SPAN = E_MAXG * E_NG * 64;                  // granules per sweep
    if (op.in == ENG_IN_NORM) {
        // the next norm's weights first: they do not depend on anything
        float w[E_MAXG];
#pragma unroll
        for (int k = 0; k < E_MAXG; k++) { const int i = k * (E_NG * 64) + gl; w[k] = i < K ? op.norm_w[i] : 0.f; }
        // this CU's own streamers next: their rows of the producing op are among the granules, and polling before they are
        // done would only take bandwidth from them
        eng_wait_local(a, c, (uint32_t)(E_NS * o));
        if (c.stamp && c.wave == 0 && c.lane == 0) c.stamp[4 + 4 * o] = wall_clock64();
        for (int i = 0; i < a.gather_delay; i++) __builtin_amdgcn_s_sleep(1);     // the other CUs' last stores are still on their way
        float ss = 0.f;
        float *res_out = c.cu == 0 ? op.res_out : nullptr;
        for (int base = 0; base < K; base += SPAN) {
            if (base) {
#pragma unroll
                for (int k = 0; k < E_MAXG; k++) { const int i = base + k * (E_NG * 64) + gl; w[k] = i < K ? op.norm_w[i] : 0.f; }
            }
            eng_sweep<false>(a, c, op.in_edge, K, base, tag, gl, base == 0, [&](int i, int k, uint32_t bits) {
                const float v = c.res[i] + __uint_as_float(bits);
                c.res[i] = v;
                xs[i] = float_to_bf16_bits(v * w[k]);
                if (res_out) res_out[i] = v;                  // the residual stream the next launch starts from
            });
        }
        // sum of squares in a FIXED order (granules arrive in any order; every CU and every run must get the same 1/rms)
        for (int i = gl; i < K; i += E_NG * 64) { const float v = c.res[i]; ss = fmaf(v, v, ss); }
        ss = wave_sum(ss);
        if (c.lane == 0) reinterpret_cast<float *>(c.misc)[32 + (o & 1) * 8 + c.wave] = ss;
    } else {                                                  // ENG_IN_ACT: bf16 values, two 4-byte granules per load
        eng_wait_local(a, c, (uint32_t)(E_NS * o));
        if (c.stamp && c.wave == 0 && c.lane == 0) c.stamp[4 + 4 * o] = wall_clock64();
        for (int i = 0; i < a.gather_delay; i++) __builtin_amdgcn_s_sleep(1);
        const int n = K >> 1;
        uint32_t *xs32 = reinterpret_cast<uint32_t *>(xs);
        for (int base = 0; base < n; base += SPAN)
            eng_sweep<true>(a, c, op.in_edge, n, base, tag, gl, base == 0, [&](int i, int, uint32_t bits) { xs32[i] = bits; });
    }
}

// ---- streamer side ---------------------------------------------------------------------------------------------------
template <int R, int U, int OUT>
__device__ inline void eng_stream(const EngArgs &a, const EngOp &op, int o, EngCtx &c, float &best_v, int &best_i) {
    const bf16_t *__restrict__ W = reinterpret_cast<const bf16_t *>(op.W);
    const bf16_t *xs = (o & 1) ? c.xs1 : c.xs0;
    const int N = op.N, K = op.K, lane = c.lane;
    const int nchunk = K >> 3;
    const int half = a.d >> 1;
    const int ngroups = (N + R - 1) / R;
    const int sw = c.wave - E_NG;
    const int gw = c.cu + c.ncu * sw, nw = c.ncu * E_NS;                  // items interleaved by CU: every CU gets the same number (+-1)
    const int nb = (nchunk + 64 * U - 1) / (64 * U);
    const int n_items = gw < ngroups ? (ngroups - gw + nw - 1) / nw * nb : 0;
    const bool ragged = nchunk % (64 * U) != 0;
    typedef RawW Buf[R][U];

    auto row_of = [&](int g, int r) -> int {
        if (OUT == ENG_OUT_EDGE_ACT) { const int q = g * (R / 2) + (r >> 1); return (q >> 4) * 32 + (q & 15) + ((r & 1) << 4); }
        if (OUT == ENG_OUT_QKV) { const int q = g * (R / 2) + (r >> 1); const int hd = q / half, j = q - hd * half; return hd * a.d + j + (r & 1) * half; }
        return g * R + r;
    };
    int lg = gw, lb = 0;
    auto load_next = [&](Buf &buf) {
        const int g = min(lg, ngroups - 1);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int row = row_of(g, r);
            const bf16_t *wr = W + (size_t)(row < N ? row : N - 1) * K;
#pragma unroll
            for (int u = 0; u < U; u++)
                buf[r][u].v = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(wr + (size_t)min(lane + 64 * (U * lb + u), nchunk - 1) * 8));
        }
        if (++lb == nb) { lb = 0; lg += nw; }
    };
    // the first block of this op (8 KiB per wave, <= 48 KB per CU) is requested before the op's input exists: the weight stream
    // runs on across the dependency.  No more than that: a CU's loads return in order, so the gatherers' sweep queues behind
    // whatever is outstanding (stamps: behind two 16-KiB blocks per wave the first edge took 7.5 us; profiles/r03/README.md)
    Buf b0, b1;
    const bool ahead = sw < E_PF_WAVES;
    if (ahead && n_items >= 1) {
        const long long t0 = wall_clock64();
        while (lds_load_u32(c.misc + 2) < (uint32_t)(E_NG * (o + 1))) {    // ... and behind the gatherers' first sweep (eng_go)
            __builtin_amdgcn_s_sleep(1);
            if (lds_load_u32(c.misc + 1) || wall_clock64() - t0 > a.timeout_ticks) break;
        }
        load_next(b0);
        if (a.pf_blocks >= 2 && n_items >= 2) load_next(b1);
    }

    // RoPE operands of this wave's first group (position from the step state): also ahead of the barrier
    uint32_t rope_p = 0, rope_slot = 0;
    float rope_c[(R + 1) / 2], rope_s[(R + 1) / 2], rope_b0[(R + 1) / 2], rope_b1[(R + 1) / 2];
    auto rope_prefetch = [&](int g) {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            const int q = g * (R / 2) + (r >> 1);
            const int hd = q / half, j = q - hd * half;
            const bool rot = hd < a.H + a.Hkv;
            rope_c[r >> 1] = rot ? a.cos_tab[(size_t)rope_p * half + j] : 1.f;
            rope_s[r >> 1] = rot ? a.sin_tab[(size_t)rope_p * half + j] : 0.f;
            const int r0w = row_of(g, r), r1w = row_of(g, r + 1);
            rope_b0[r >> 1] = op.bias && r1w < N ? op.bias[r0w] : 0.f;
            rope_b1[r >> 1] = op.bias && r1w < N ? op.bias[r1w] : 0.f;
        }
    };
    if (OUT == ENG_OUT_QKV) {
        const uint32_t pos = a.st->pos;
        rope_slot = a.st->len;
        rope_p = pos < (uint32_t)a.max_pos ? pos : (uint32_t)a.max_pos - 1;
        if (gw < ngroups) rope_prefetch(gw);
    }

    __builtin_amdgcn_s_barrier();                                         // x of this op is in LDS (the gatherers waited for their writes)
    asm volatile("" ::: "memory");
    if (c.stamp && c.wave == E_NG && lane == 0) c.stamp[3 + 4 * o] = wall_clock64();              // streamer wave 0 past the barrier of op o
    if (!ahead && n_items >= 1) load_next(b0);
    if (n_items >= 2 && !(ahead && a.pf_blocks >= 2)) load_next(b1);
    float inv_m = 1.0f;
    if (op.in == ENG_IN_NORM) {
        const float *ssp = reinterpret_cast<const float *>(c.misc) + 32 + (o & 1) * 8;
        float ss = 0.f;
#pragma unroll
        for (int w = 0; w < E_NG; w++) ss += ssp[w];
        inv_m = 1.0f / sqrtf(ss / (float)K + a.eps);                      // candle rms_norm (App. A.2)
    }
    const uint32_t tag = c.ebase | (uint32_t)op.tag_out;

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.f;
    auto finish_group = [&](int g) {
        float sum[R];
#pragma unroll
        for (int r = 0; r < R; r++) { sum[r] = wave_sum(acc[r]) * inv_m; acc[r] = 0.f; }
        if constexpr (OUT == ENG_OUT_EDGE_F32) {
            // rows g*R + r -> lane r publishes {fp32 value, tag}
            float v = sum[0];
#pragma unroll
            for (int r = 1; r < R; r++) v = lane == r ? sum[r] : v;
            const int row = g * R + lane;
            if (lane < R && row < N) st_granule(op.out_edge + row, tag, __float_as_uint(v));
        } else if constexpr (OUT == ENG_OUT_EDGE_ACT) {
            // one gate/up pair per item -> one 4-byte {bf16 silu(g)*u, 16-bit tag} granule (the tag's low half: consecutive steps
            // differ in it, and a word only ever holds the last step's value); two-row items keep every wave of the CU streaming
            static_assert(R == 2, "one channel per item");
            const float a0 = sum[0] / (1.0f + expf(-sum[0])) * sum[1];            // candle silu(g) * u
            if (lane == 0 && row_of(g, 1) < N)
                __hip_atomic_store((__attribute__((address_space(1))) uint32_t *)(reinterpret_cast<uint32_t *>(op.out_edge) + g),
                                   ((tag & 0xffffu) << 16) | (uint32_t)float_to_bf16_bits(a0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if constexpr (OUT == ENG_OUT_QKV) {
            if (lane != 0) return;
            const uint32_t slot = rope_slot;
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const int r1w = row_of(g, r + 1);
                if (r1w >= N) continue;
                const int q = g * (R / 2) + (r >> 1);
                const int hd = q / half, j = q - hd * half;
                float x0 = sum[r], x1 = sum[r + 1];
                if (op.bias) { x0 += rope_b0[r >> 1]; x1 += rope_b1[r >> 1]; }
                bf16_t *dst;
                size_t stride = 1;
                if (hd < a.H + a.Hkv) {                                   // rotate-half RoPE (App. A.4)
                    float t0, t1;
                    rope_rotate(x0, x1, rope_c[r >> 1], rope_s[r >> 1], t0, t1);
                    x0 = t0; x1 = t1;
                    dst = hd < a.H ? reinterpret_cast<bf16_t *>(a.q_out) + (size_t)hd * a.d
                                   : reinterpret_cast<bf16_t *>(a.k_cache) + ((size_t)(hd - a.H) * a.max_seq + slot) * a.d;
                } else if (a.v_ld > 0) {                                  // transposed value cache [Hkv][d][v_ld]
                    dst = reinterpret_cast<bf16_t *>(a.v_cache) + (size_t)(hd - a.H - a.Hkv) * a.d * a.v_ld + slot;
                    stride = (size_t)a.v_ld;
                } else {
                    dst = reinterpret_cast<bf16_t *>(a.v_cache) + ((size_t)(hd - a.H - a.Hkv) * a.max_seq + slot) * a.d;
                }
                dst[(size_t)j * stride] = float_to_bf16_bits(x0);
                dst[(size_t)(j + half) * stride] = float_to_bf16_bits(x1);
            }
        } else {                                                          // ENG_OUT_LOGITS
            if (lane != 0) return;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int row = g * R + r;
                if (row < N) {
                    const float y = sum[r] + (op.bias ? op.bias[row] : 0.f);
                    reinterpret_cast<float *>(op.dst)[row] = y;
                    if (a.amax && (best_i < 0 || y > best_v || (y == best_v && row > best_i))) { best_v = y; best_i = row; }
                }
            }
        }
    };
    int cg = gw, cb = 0;
    auto consume = [&](const Buf &buf) {
        if (OUT == ENG_OUT_QKV && cb == 0 && cg != gw) rope_prefetch(cg);
        const int c0 = lane + 64 * U * cb;
        if (ragged && cb == nb - 1) {                                     // wave-uniform: the partial last block of K
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int ci = c0 + 64 * u;
                uint4v xr = *reinterpret_cast<const uint4v *>(xs + min(ci, nchunk - 1) * 8);
                if (ci >= nchunk) xr = uint4v{0u, 0u, 0u, 0u};
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[r] = dot2c_bf16(buf[r][u].v[j], xr[j], acc[r]);
            }
#pragma unroll
            for (int r = 0; r < R; r++) dot2c_settle(acc[r]);             // (inside the branch: the hazard window closes before the join)
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint4v xr = *reinterpret_cast<const uint4v *>(xs + (c0 + 64 * u) * 8);
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[r] = dot2c_bf16(buf[r][u].v[j], xr[j], acc[r]);
            }
#pragma unroll
            for (int r = 0; r < R; r++) dot2c_settle(acc[r]);
        }
        if (++cb == nb) { finish_group(cg); cb = 0; cg += nw; }
    };
    int t = 0;
#pragma nounroll
    for (; t + 3 < n_items; t += 2) {                                     // two blocks in flight; every load of the loop unconditional
        consume(b0);
        load_next(b0);
        consume(b1);
        load_next(b1);
    }
    const int rest = n_items - t;
    if (rest == 3) { consume(b0); load_next(b0); consume(b1); consume(b0); }
    else if (rest == 2) { consume(b0); consume(b1); }
    else if (rest == 1) consume(b0);
}

template <bool STAMPS>
__global__ __launch_bounds__(E_THREADS) void engine_kernel(const EngArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    EngCtx c;
    c.lds = lds_raw;
    c.misc = reinterpret_cast<uint32_t *>(lds_raw);
    c.xs0 = reinterpret_cast<bf16_t *>(lds_raw + E_MISC_BYTES);
    c.xs1 = reinterpret_cast<bf16_t *>(lds_raw + E_MISC_BYTES + a.xs0_bytes);
    c.res = reinterpret_cast<float *>(lds_raw + E_MISC_BYTES + a.xs0_bytes + a.xs1_bytes);
    c.cu = blockIdx.x; c.ncu = gridDim.x;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.lane = threadIdx.x & 63;
    c.ebase = *a.epoch << 8;
    unsigned long long *stamp = STAMPS && a.stamps ? a.stamps + (size_t)blockIdx.x * 32 : nullptr;
    c.stamp = stamp;
    if (threadIdx.x < 8) c.misc[threadIdx.x] = 0;
    if (STAMPS && stamp && threadIdx.x == 0) stamp[0] = wall_clock64();
    __syncthreads();

    float best_v = -INFINITY; int best_i = -1;
    for (int o = 0; o < a.nops; o++) {
        const EngOp &op = a.op[o];
        if (c.wave < E_NG) {
            eng_gather(a, op, o, c);
            if (STAMPS && stamp && threadIdx.x == 0) stamp[1 + 4 * o] = wall_clock64();          // input of op o complete on this CU
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the LDS writes have landed before the barrier releases the streamers
            __builtin_amdgcn_s_barrier();
        } else {
            // 8-KiB blocks everywhere: two rows (a gate/up pair, a RoPE pair) x four 1-KiB chunks
            if (op.out == ENG_OUT_EDGE_ACT) eng_stream<2, 4, ENG_OUT_EDGE_ACT>(a, op, o, c, best_v, best_i);
            else if (op.out == ENG_OUT_EDGE_F32) {
                // few rows per CU (o_proj, down_proj of a small model): one row per wave, so that every wave of the CU streams
                if (op.R == 1) eng_stream<1, 4, ENG_OUT_EDGE_F32>(a, op, o, c, best_v, best_i);
                else eng_stream<2, 4, ENG_OUT_EDGE_F32>(a, op, o, c, best_v, best_i);
            }
            else if (op.out == ENG_OUT_QKV) eng_stream<2, 4, ENG_OUT_QKV>(a, op, o, c, best_v, best_i);
            else eng_stream<2, 4, ENG_OUT_LOGITS>(a, op, o, c, best_v, best_i);
            if (STAMPS && stamp && threadIdx.x == E_NG * 64) stamp[2 + 4 * o] = wall_clock64();  // streamer wave 0 done with op o
            if (c.lane == 0) __hip_atomic_fetch_add(c.misc + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    // the workgroup's ArgMax candidate (lm_head as the last op): lane 0 of every streamer wave holds the best of its rows
    if (a.amax) {
        float *cv = reinterpret_cast<float *>(c.misc) + 8;
        int *ci = reinterpret_cast<int *>(c.misc) + 8 + (E_NG + E_NS);
        if (c.lane == 0) { cv[c.wave] = best_v; ci[c.wave] = best_i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float bv = -INFINITY; int bi = -1;
            for (int w = E_NG; w < E_NG + E_NS; w++)
                if (ci[w] >= 0 && (bi < 0 || cv[w] > bv || (cv[w] == bv && ci[w] > bi))) { bv = cv[w]; bi = ci[w]; }
            a.amax[1 + blockIdx.x] = ArgmaxCand{bv, bi};
            if (blockIdx.x == 0) a.amax[0] = ArgmaxCand{0.f, (int)gridDim.x};
        }
    }
    if (STAMPS && stamp && threadIdx.x == 0) stamp[31] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------ host side
bool engine_shape_ok(int64_t h, int64_t Hd, int64_t I, int64_t N_last) {
    // vectors: x of o_proj / down_proj in xs[0], of gate/up / the last op in xs[1], the fp32 residual stream
    if (h % 8 || Hd % 8 || I % 8 || h < 8 || Hd < 8 || I < 8 || N_last < 1) return false;
    const size_t lds = E_MISC_BYTES + (size_t)std::max(Hd, I) * 2 + (size_t)h * 2 + (size_t)h * 4;
    return lds <= 150 * 1024;
}

static std::atomic<int> g_engine_grid{0};
void engine_set_grid(int workgroups) { g_engine_grid = workgroups > 0 ? workgroups : 0; }

int launch_engine(Launcher &L, const EngArgs &a_in) {
    EngArgs a = a_in;
    if (a.grid <= 0) a.grid = g_engine_grid.load();
    if (a.nops < 1 || a.nops > ENG_MAX_OPS) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: 1..%d ops", ENG_MAX_OPS);
    if (!a.epoch || !a.st || !a.st_rw) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: epoch word and step state are required");
    size_t k0 = 0, k1 = 0;
    double bytes = 0;
    for (int o = 0; o < a.nops; o++) {
        const EngOp &op = a.op[o];
        if (op.N <= 0 || op.K < 8 || op.K % 8 || !op.W) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: bad shape of op %d", o);
        if (op.in == ENG_IN_X ? !op.x : !op.in_edge) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: op %d has no input", o);
        if (op.in == ENG_IN_NORM && (!op.norm_w || op.K != a.h || !a.x_res_in || o == 0)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: op %d: the norm prologue runs over the residual stream behind an edge", o);
        if (op.in == ENG_IN_ACT && (o == 0 || a.op[o - 1].out != ENG_OUT_EDGE_ACT || a.op[o - 1].N != 2 * op.K)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: op %d: the activation edge follows a gate/up op of twice its width", o);
        if (o > 0 && op.in == ENG_IN_X) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: only the first op reads a plain vector");
        if (o > 0 && op.in == ENG_IN_NORM && (a.op[o - 1].out != ENG_OUT_EDGE_F32 || a.op[o - 1].N != op.K || a.op[o - 1].out_edge != op.in_edge || a.op[o - 1].tag_out != op.tag_in))
            FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: op %d does not read the edge its predecessor writes", o);
        if ((op.out == ENG_OUT_EDGE_F32 || op.out == ENG_OUT_EDGE_ACT) && !op.out_edge) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: op %d has no output edge", o);
        if (op.out == ENG_OUT_EDGE_ACT && op.N % 32) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: gate/up rows must be a multiple of 32");
        if (op.out == ENG_OUT_QKV && (a.d <= 0 || a.d % 2 || op.N != (a.H + 2 * a.Hkv) * a.d)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: bad qkv shape");
        if (op.out == ENG_OUT_LOGITS && !op.dst) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: logits need a buffer");
        if ((op.out == ENG_OUT_QKV || op.out == ENG_OUT_LOGITS) && o != a.nops - 1) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: a global output ends the chain");
        if (op.tag_in < 0 || op.tag_in > 255 || op.tag_out < 0 || op.tag_out > 255) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: tags are 1..255");
        if (op.out == ENG_OUT_EDGE_F32) a.op[o].R = (op.N / 2 < (a.grid > 0 ? a.grid : 256) * E_NS) ? 1 : 2;
        ((o & 1) ? k1 : k0) = std::max((o & 1) ? k1 : k0, (size_t)op.K);
        bytes += (double)op.N * op.K * 2;
    }
    a.xs0_bytes = (int)((k0 * 2 + 15) & ~(size_t)15);
    a.xs1_bytes = (int)((k1 * 2 + 15) & ~(size_t)15);
    const size_t lds = E_MISC_BYTES + (size_t)a.xs0_bytes + a.xs1_bytes + (size_t)a.h * 4;
    if (lds > 150 * 1024) FL_FAIL(FL_ERR_UNSUPPORTED, "engine: vectors of %zu bytes do not fit the LDS", lds);
    int dev = 0;
    FL_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    static int cus[64];
    if (dev < 0 || dev >= 64) FL_FAIL(FL_ERR_HIP, "engine: device index");
    if (!cus[dev]) { FL_HIP(hipGetDeviceProperties(&p, dev)); cus[dev] = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }
    const int blocks = a.grid > 0 ? a.grid : cus[dev];
    const int delay = tune(TK_ENGINE_DELAY);
    a.gather_delay = delay;
    const int pfb = tune(TK_ENGINE_PF);
    a.pf_blocks = pfb;
    if (a.amax && blocks + 1 > kMaxArgmaxCand) FL_FAIL(FL_ERR_BAD_ARGUMENT, "engine: %d workgroups exceed the ArgMax candidate buffer", blocks);
    char tag[32];
    snprintf(tag, sizeof tag, "eng%d:%dx%d..%dx%d", a.nops, a.op[0].N, a.op[0].K, a.op[a.nops - 1].N, a.op[a.nops - 1].K);
    Launcher LL = L; LL.tag = tag;
    // FL_ENGINE_STAMPS=<file>: the diagnostic instantiation -- every workgroup records wall-clock stamps (100 MHz) at its op
    // boundaries; the launch is synchronous and one line per launch is appended to the file (tools/engine_stamps.py reads it)
    const char *stamp_path = env_str("FL_ENGINE_STAMPS");
    if (stamp_path && *stamp_path && !a.stamps) {
        static unsigned long long *dbuf = nullptr;
        if (!dbuf) FL_HIP(hipMalloc(&dbuf, (size_t)1024 * 32 * 8));
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(L.stream, &cs);
        if (cs == hipStreamCaptureStatusNone && blocks <= 1024) {
            a.stamps = dbuf;
            FL_HIP(hipMemsetAsync(dbuf, 0, (size_t)blocks * 32 * 8, L.stream));
            auto kern = engine_kernel<true>;
            FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
            FL_TRY(LL.launch(KC_GEMV, bytes, bytes, kern, dim3((unsigned)blocks), dim3(E_THREADS), lds, a));
            std::vector<unsigned long long> hs((size_t)blocks * 32);
            FL_HIP(hipStreamSynchronize(L.stream));
            FL_HIP(hipMemcpy(hs.data(), dbuf, hs.size() * 8, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(stamp_path, "a")) {
                fprintf(f, "launch %s nops %d blocks %d\n", tag, a.nops, blocks);
                for (int b = 0; b < blocks; b++) {
                    fprintf(f, "wg %d", b);
                    for (int i = 0; i < 32; i++) fprintf(f, " %lld", hs[(size_t)b * 32 + i] ? (long long)(hs[(size_t)b * 32 + i] - hs[(size_t)b * 32]) : -1ll);
                    fprintf(f, "\n");
                }
                fclose(f);
            }
            return FL_OK;
        }
    }
    if (a.stamps) {
        auto kern = engine_kernel<true>;
        FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
        return LL.launch(KC_GEMV, bytes, bytes, kern, dim3((unsigned)blocks), dim3(E_THREADS), lds, a);
    }
    auto kern = engine_kernel<false>;
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    return LL.launch(KC_GEMV, bytes, bytes, kern, dim3((unsigned)blocks), dim3(E_THREADS), lds, a);
}

#else
// Default build: the engine measured 5 % slower than the launches it replaces (profiles/r03/README.md) and is not compiled in;
// `make EXPERIMENTAL=1` builds it (fastllm_amd/lib/libfastllm_mi355x_exp.so) and tests/test_gpu_engine.py runs against that library.
bool engine_shape_ok(int64_t, int64_t, int64_t, int64_t) { return false; }
void engine_set_grid(int) {}
int launch_engine(Launcher &, const EngArgs &) { FL_FAIL(FL_ERR_UNSUPPORTED, "the decode engine is not in this build (make EXPERIMENTAL=1)"); }
#endif
}  // namespace fl
