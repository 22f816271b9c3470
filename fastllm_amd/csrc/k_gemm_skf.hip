// k_gemm_skf.hip -- the short-prompt projection GEMM (T <= 128 tokens: a weight stream, k_gemm_skinny.hip) with its K slices met INSIDE
// the launch and the layer's element-wise work in its epilogue, so that a short prompt's -- and a decode batch's -- layer is five
// launches like a decode step's (QKV + RoPE + KV append | attention | o_proj + residual + norm | gate/up | down_proj + residual + norm)
// instead of eight (rmsnorm_add, QKV slabs, rope_kv, attention, o_proj slabs, rmsnorm_add, gate/up, down_proj slabs):
//
//   Y[T,N] = X[T,K] . W[N,K]^T      bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16), Mistral-7B: 14.5 GB of W per pass
//
// K loop: gemm_skinny_kernel's, unchanged (a workgroup owns all T tokens and a 32 / 64 / 128-row strip of W; four LDS stages filled by
// LDS-DMA three K tiles ahead, one raw barrier per K tile, counted vmcnt).
// Where the slices meet: every workgroup publishes its accumulators (write-through sc1 stores, lane-major float4s), waits for its
// stores and draws a ticket on the tile's word; the one that draws the last ticket adds the others' tiles to its own IN K ORDER (its
// own at its place: the sum does not depend on who came last) and runs the epilogue.  Nobody waits for anybody, so the grid need not be
// co-resident (processes may share the GPU, a captured decode-batch graph may replay next to anything); the word is reset by the last
// arriver, so a launch leaves the workspace as it found it and there is no per-launch host state (graph capture).  At T <= 128 a tile
// is at most 32 KB: the serial tail of the last arriver is one L2 round trip plus the epilogue it would have run anyway (the 128 x 256
// kernel's static block ownership, k_gemm_h4.hip, exists because ITS tiles are 128 KB).
// Epilogues (the last arriver's, or every workgroup's when K is not sliced).  The DEFAULT library holds EPI_F32 only -- what it runs: a
// tensor-parallel rank's complete outputs; the other three are the five-launch layer, which measured slower than the launches it
// replaces (profiles/r05/README.md) and lives in the EXPERIMENTAL build with the other kernels that lost:
//   EPI_F32       row scale, bias -> fp32 (a complete output: what a tensor-parallel rank's all-reduce wants)
//   EPI_GATEUP    silu(gate) * up -> bf16; the row scale may come from a residual epilogue's partial sums (RsParts)
//   EPI_RESID     h += y; xn = bf16((h + y) * w_next); one partial sum of squares per (row, 64-column strip)  (kernels.h, ResidEpi)
//   EPI_QKV_ROPE  row scale, bias, rotate-half RoPE of the q / k heads, q -> activation buffer, k / v -> the cache (V transposed for
//                 the MFMA attention): a lane holds column c of a head in one accumulator tile and its partner c + d/2 in the other
//                 (the strip's fragment rows are mapped that way), so the rotation is lane-local.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#include "attn_common.h"
#include "kernels.h"

namespace fl {

typedef __bf16 bf16x8f __attribute__((ext_vector_type(8)));
constexpr int F_BK = 64;
constexpr size_t kSkfPartBytes = (size_t)32 << 20;       // partial tiles of one launch, per stream
constexpr int kSkfMaxTiles = 4096;                        // ticket words

struct SkfSpace { float *part = nullptr; unsigned *cnt = nullptr; };
struct SkfArgs { SkfSpace ws; ResidEpi re; RopeEpi ro; RsParts rsp; };

__device__ inline void glds16f(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ inline void glds16f_nt(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 2);
}
__device__ inline bf16x8f frag_f(const unsigned char *tile, int row, int chunk) {
    return *reinterpret_cast<const bf16x8f *>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
template <int N> __device__ inline void wait_vmcnt_f() {
    static_assert(N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int MAXY, int PW> __device__ inline void wait_tail_f(int younger) {
    if constexpr (MAXY <= 0) { wait_vmcnt_f<0>(); }
    else { if (younger >= MAXY) wait_vmcnt_f<MAXY * PW>(); else wait_tail_f<MAXY - 1, PW>(younger); }
}

// BM tokens x (NW * 16 * NT) weight rows per workgroup of NWM x NW waves (see gemm_skinny_kernel).  NT = 2: the wave's two accumulator
// tiles are a gate / up pair (rows r, r + 16) or, for the RoPE epilogue, a column of a head and its rotate-half partner (r, r + d/2).
template <int BM, int NT, int NW, int S_NSTG, bool WNT, int NWM>
__global__ __launch_bounds__(NW * NWM * 64) void gemm_skf_kernel(const bf16_t *__restrict__ W, const bf16_t *__restrict__ X,
                                                                 const float *__restrict__ bias, void *__restrict__ out,
                                                                 int T, int N, int K, int epi, const float *__restrict__ row_scale,
                                                                 int ksplit, const SkfArgs a) {
    constexpr int BN = NW * 16 * NT, MT = BM / 16 / NWM, NTHR = NW * NWM * 64;
    constexpr int XB = BM * 128, STG = XB + BN * 128;
    constexpr int NI = (BM + BN) / 8, PW = NI / (NW * NWM);
    static_assert(NI % (NW * NWM) == 0, "stage instructions must divide evenly over the waves");
    static_assert(BM % (16 * NWM) == 0, "token tiles must divide over the wave rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
    const int wave = wave_all % NW, wm = wave_all / NW;
    const int m16 = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.z * BM;
    const int nk_all = K / F_BK, kz = blockIdx.y;
    const int kt0 = (int)((long long)nk_all * kz / ksplit), nk = (int)((long long)nk_all * (kz + 1) / ksplit) - kt0;
    X += (size_t)kt0 * F_BK; W += (size_t)kt0 * F_BK;

    // rows of the strip (0 .. BN) this wave's accumulator tiles j = 0 / 1 multiply: consecutive blocks of 16, a gate / up pair, or --
    // RoPE -- a head's column block and its partner half a head further (d = 128: the 128-row strip is one head; d = 64: two, or
    // with two waves one)
    int wrow[NT];
    if constexpr (NT == 2) {
        if (epi == EPI_QKV_ROPE) {
            const int half = a.ro.d >> 1, per_head = half / 16;                 // wave columns per head: 4 (d = 128) or 2 (d = 64)
            wrow[0] = (wave / per_head) * a.ro.d + (wave % per_head) * 16;
            wrow[1] = wrow[0] + half;
        } else { wrow[0] = wave * 32; wrow[1] = wave * 32 + 16; }
    } else { wrow[0] = wave * 16; }

    float4v acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int kt) {
        unsigned char *base = lds + (kt % S_NSTG) * STG;
#pragma unroll
        for (int s = 0; s < PW; s++) {
            const int q = wave_all * PW + s;
            const int rb = q * 8, r = rb + (lane >> 3), pc = lane & 7, c = pc ^ ((r >> 1) & 7);
            if (rb < BM) {
                int gr = m0 + r; if (gr > T - 1) gr = T - 1;
                glds16f(X + (size_t)gr * K + kt * F_BK + c * 8, base + rb * 128);
            } else {
                int gr = n0 + r - BM; if (gr > N - 1) gr = N - 1;
                if constexpr (WNT) glds16f_nt(W + (size_t)gr * K + kt * F_BK + c * 8, base + rb * 128);
                else glds16f(W + (size_t)gr * K + kt * F_BK + c * 8, base + rb * 128);
            }
        }
    };
#pragma unroll
    for (int p = 0; p < S_NSTG - 1; p++)
        if (p < nk) stage(p);
    for (int kt = 0; kt < nk; kt++) {
        {
            const int younger = nk - 1 - kt;
            if (younger >= S_NSTG - 2) wait_vmcnt_f<(S_NSTG - 2) * PW>();
            else wait_tail_f<S_NSTG - 3, PW>(younger);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + S_NSTG - 1 < nk) stage(kt + S_NSTG - 1);
        const unsigned char *xt = lds + (kt % S_NSTG) * STG, *wt = xt + XB;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int chunk = ks * 4 + kg;
            bf16x8f b[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) b[j] = frag_f(wt, wrow[j] + m16, chunk);
#pragma unroll
            for (int i = 0; i < MT; i++) {
                const bf16x8f af = frag_f(xt, (wm * MT + i) * 16 + m16, chunk);
#pragma unroll
                for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                                    // every wave is done with the ring: LDS is the epilogue's now
    float *rs_lds = reinterpret_cast<float *>(lds);                     // [BM] row scales
    float *ss_lds = rs_lds + BM;                                        // [BM][NW] partial sums of squares (EPI_RESID)
    int *flag_lds = reinterpret_cast<int *>(ss_lds + BM * NW);

    // ---- the K slices meet: publish, ticket, and the last arriver sums in K order ----
    if (ksplit > 1) {
        const int tile = blockIdx.z * gridDim.x + blockIdx.x;
        float4v *slab0 = reinterpret_cast<float4v *>(a.ws.part) + (size_t)tile * ksplit * (MT * NT * NTHR) + tid;
        float4v *mine = slab0 + (size_t)kz * (MT * NT * NTHR);
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NT; j++) st_sc1_x4(reinterpret_cast<float *>(mine + (size_t)(i * NT + j) * NTHR), acc[i][j]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned *cnt = a.ws.cnt + tile;
            const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = t == (unsigned)ksplit - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (behind the acquire's wait, as attn_common.h)
            }
            flag_lds[0] = last;
        }
        __syncthreads();
        if (!flag_lds[0]) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                 // every wave reads the peers' tiles: behind its own acquire
        // The (up to three) peers' tiles: ALL requested before the first is used, unconditionally (a slice index past the last re-reads
        // the last tile and is dropped from the sum) -- a load inside a loop with a run-time bound is waited for where it stands, and a
        // tile of MT x NT float4s per peer, one round trip each, was most of this kernel's tail (profiles/r05/README.md)
        float4v pp[3][MT][NT];
#pragma unroll
        for (int q3 = 0; q3 < 3; q3++) {
            const int q = min(q3 + (q3 >= kz ? 1 : 0), ksplit - 1);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++) pp[q3][i][j] = slab0[(size_t)q * (MT * NT * NTHR) + (size_t)(i * NT + j) * NTHR];
        }
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NT; j++) {
                // K order: slice q is this workgroup's own (q == kz) or peer q (q < kz: pp[q]) / (q > kz: pp[q - 1])
                float4v sum = kz == 0 ? acc[i][j] : pp[0][i][j];
#pragma unroll
                for (int q = 1; q < 4; q++) {
                    const float4v v = q == kz ? acc[i][j] : pp[q < kz ? q : q - 1][i][j];
                    if (q < ksplit) sum += v;
                }
                acc[i][j] = sum;
            }
    }

    // ---- row scales of the tile's tokens: the caller's vector, a residual epilogue's partial sums, or 1 ----
    if (tid < BM) {
        const int row = min(m0 + tid, T - 1);
        float rs = 1.0f;
        if (a.rsp.part) {
            const float4v *p = reinterpret_cast<const float4v *>(a.rsp.part + (size_t)row * a.rsp.np);
            float ss = 0.f;
            for (int i = 0; i < a.rsp.np / 4; i++) { const float4v v = p[i]; ss += v[0]; ss += v[1]; ss += v[2]; ss += v[3]; }
            rs = 1.0f / sqrtf(ss * a.rsp.inv_h + a.rsp.eps);
        } else if (row_scale) rs = row_scale[row];
        rs_lds[tid] = rs;
    }
    __syncthreads();

    const int cn = lane & 15, rm = (lane >> 4) * 4;
    const int mw = wm * MT * 16;                                        // first token (within the tile) of this wave's rows
#ifdef FL_EXPERIMENTAL                                                  // (the five-launch layer's epilogues: see the header)
    if (epi == EPI_GATEUP) {
        if constexpr (NT == 2) {
            const int n = n0 + wrow[0] + cn;                            // gate row; up = n + 16
            const int qq = (n >> 5) * 16 + (n & 15);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int ml = mw + i * 16 + rm + rg, m = m0 + ml;
                    if (m >= T || n + 16 >= N) continue;
                    const float rs = rs_lds[ml];
                    const float gt = acc[i][0][rg] * rs, up = acc[i][1][rg] * rs;
                    const float av = gt / (1.0f + expf(-gt)) * up;
                    reinterpret_cast<bf16_t *>(out)[(size_t)m * (N / 2) + qq] = float_to_bf16_bits(av);
                }
        }
        return;
    }
    if (epi == EPI_QKV_ROPE) {
        if constexpr (NT == 2) {
            const RopeEpi &ro = a.ro;
            const int D = ro.d, half = D >> 1;
            const int n1 = n0 + wrow[0] + cn;                           // this lane's column; its partner is n1 + half
            if (n1 >= N) return;
            const int hd = n1 / D, c1 = n1 % D;                         // head (q heads, then k, then v) and index within it (< half)
            const uint32_t pos0 = ro.seqs ? 0u : ro.st->pos, len0 = ro.seqs ? 0u : ro.st->len;
            const float b1 = bias ? bias[n1] : 0.f, b2 = bias ? bias[n1 + half] : 0.f;
            const bool rot = hd < ro.H + ro.Hkv;
            // One row's rotation and stores.  Everything the stores depend on -- cos / sin, and a decode batch's per-row step states --
            // is requested for ALL of the lane's rows before the first store (the compiler cannot move a load over a store that may
            // alias it: row after row, each waited for).
            auto put = [&](int i, int rg, uint32_t slot, size_t sa, bf16_t *kb, bf16_t *vb, float cs, float sn) {
                const int ml = mw + i * 16 + rm + rg, t = m0 + ml;
                if (t >= T) return;
                const float rs = rs_lds[ml];
                const float x1 = acc[i][0][rg] * rs + b1, x2 = acc[i][1][rg] * rs + b2;
                float y1 = x1, y2 = x2;
                if (rot) rope_rotate(x1, x2, cs, sn, y1, y2);
                if (hd < ro.H) {
                    bf16_t *o = reinterpret_cast<bf16_t *>(ro.q_out) + ((size_t)t * ro.H + hd) * D + c1;
                    o[0] = float_to_bf16_bits(y1); o[half] = float_to_bf16_bits(y2);
                } else if (rot || !ro.v_transposed) {
                    bf16_t *o = (rot ? kb + (size_t)(hd - ro.H) * sa * D : vb + (size_t)(hd - ro.H - ro.Hkv) * sa * D) + (size_t)slot * D + c1;
                    o[0] = float_to_bf16_bits(y1); o[half] = float_to_bf16_bits(y2);
                } else {                                                // transposed value cache [Hkv][D][seq]
                    bf16_t *o = vb + ((size_t)(hd - ro.H - ro.Hkv) * D + c1) * sa + slot;
                    o[0] = float_to_bf16_bits(y1); o[(size_t)half * sa] = float_to_bf16_bits(y2);
                }
            };
            auto trig = [&](uint32_t pos, float &cs, float &sn) {
                const uint32_t p = pos < (uint32_t)ro.max_pos ? pos : (uint32_t)ro.max_pos - 1;   // host validates range
                cs = rot ? ro.cos_tab[(size_t)p * half + c1] : 1.f;
                sn = rot ? ro.sin_tab[(size_t)p * half + c1] : 0.f;
            };
            float cs[MT][4], sn[MT][4];
            if (!ro.seqs) {
                // a prompt: token t at position pos0 + t, KV slot len0 + t of the one cache
                bf16_t *kb = reinterpret_cast<bf16_t *>(ro.k_cache), *vb = reinterpret_cast<bf16_t *>(ro.v_cache);
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) trig(pos0 + (uint32_t)min(m0 + mw + i * 16 + rm + rg, T - 1), cs[i][rg], sn[i][rg]);
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) put(i, rg, len0 + (uint32_t)(m0 + mw + i * 16 + rm + rg), (size_t)ro.max_seq, kb, vb, cs[i][rg], sn[i][rg]);
            } else {
                // a decode batch: row t is sequence t's one new token -- its own position, slot and caches
                uint32_t pos[MT][4], slot[MT][4];
                int sa[MT][4];
                bf16_t *kb[MT][4], *vb[MT][4];
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) {
                        const SeqRef &sq = ro.seqs[min(m0 + mw + i * 16 + rm + rg, T - 1)];
                        pos[i][rg] = sq.st->pos; slot[i][rg] = sq.st->len; sa[i][rg] = sq.seq_alloc;
                        kb[i][rg] = reinterpret_cast<bf16_t *>(sq.k); vb[i][rg] = reinterpret_cast<bf16_t *>(sq.v);
                    }
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) trig(pos[i][rg], cs[i][rg], sn[i][rg]);
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int rg = 0; rg < 4; rg++)
                        put(i, rg, slot[i][rg], (size_t)sa[i][rg], kb[i][rg] + ro.kv_layer_off * (size_t)sa[i][rg], vb[i][rg] + ro.kv_layer_off * (size_t)sa[i][rg], cs[i][rg], sn[i][rg]);
            }
        }
        return;
    }
    if (epi == EPI_RESID) {
        // h += y; xn = bf16((h + y) * w_next); this strip's share of every row's sum of squares: 16 lanes of a row, then the NW waves
        // (fixed order), one slot per row and strip
        const ResidEpi &re = a.re;
        // (all of the lane's h values are requested before the first store: a load cannot move over a store to the same array)
        float hv[MT][4][NT], wn[NT], bn_[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int n = min(n0 + wrow[j] + cn, N - 1);
            wn[j] = re.w[n]; bn_[j] = bias ? bias[n] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++)
#pragma unroll
                for (int j = 0; j < NT; j++)
                    hv[i][rg][j] = re.h[(size_t)min(m0 + mw + i * 16 + rm + rg, T - 1) * N + min(n0 + wrow[j] + cn, N - 1)];
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const int ml = mw + i * 16 + rm + rg, m = m0 + ml;
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const int n = n0 + wrow[j] + cn;
                    if (m < T && n < N) {
                        const float hn = hv[i][rg][j] + (acc[i][j][rg] + bn_[j]);
                        re.h[(size_t)m * N + n] = hn;
                        reinterpret_cast<bf16_t *>(re.xn)[(size_t)m * N + n] = float_to_bf16_bits(hn * wn[j]);
                        ss = fmaf(hn, hn, ss);
                    }
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o);
                if (cn == 0) ss_lds[ml * NW + wave] = ss;
            }
        __syncthreads();
        if (tid < BM && m0 + tid < T) {
            float s = ss_lds[tid * NW];
#pragma unroll
            for (int w = 1; w < NW; w++) s += ss_lds[tid * NW + w];
            re.part[(size_t)(m0 + tid) * re.np + blockIdx.x] = s;
            // (np counts four slots per 256 columns: a width that is not a multiple of 256 leaves slots behind the last strip, which the
            // consumer adds too)
            if (blockIdx.x == gridDim.x - 1)
                for (int sl = (int)gridDim.x; sl < re.np; sl++) re.part[(size_t)(m0 + tid) * re.np + sl] = 0.f;
        }
        return;
    }
#endif
    // EPI_F32: row scale and bias -> fp32
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int ml = mw + i * 16 + rm + rg, m = m0 + ml;
            if (m >= T) continue;
            const float rs = rs_lds[ml];
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int n = n0 + wrow[j] + cn;
                if (n < N) reinterpret_cast<float *>(out)[(size_t)m * N + n] = acc[i][j][rg] * rs + (bias ? bias[n] : 0.f);
            }
        }
}

// ------------------------------------------------------------------------------------------------ host side
// workspace of a stream: 32 MB of partial tiles + the ticket words (zeroed once: the last arriver of a tile resets its word)
struct SkfWs { float *part = nullptr; unsigned *cnt = nullptr; };
static std::mutex g_skf_mu;
static std::map<std::pair<int, hipStream_t>, SkfWs> g_skf_spaces;

static int skf_space(hipStream_t stream, SkfSpace *out) {
    int dev = 0;
    FL_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_skf_mu);
    SkfWs &sp = g_skf_spaces[{dev, stream}];
    if (!sp.part) {
        // (first sliced launch on this stream.  Never inside a stream capture: a decode batch's first step runs eagerly, model.hip)
        FL_HIP(hipMalloc((void **)&sp.part, kSkfPartBytes));
        const hipError_t e = hipMalloc((void **)&sp.cnt, kSkfMaxTiles * sizeof(unsigned));
        if (e != hipSuccess) { (void)hipFree(sp.part); sp.part = nullptr; FL_HIP(e); }
        FL_HIP(hipMemsetAsync(sp.cnt, 0, kSkfMaxTiles * sizeof(unsigned), stream));
    }
    out->part = sp.part; out->cnt = sp.cnt;
    return FL_OK;
}
void gemm_skf_release_stream(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_skf_mu);
    auto it = g_skf_spaces.find({dev, stream});
    if (it == g_skf_spaces.end()) return;
    if (it->second.part) (void)hipFree(it->second.part);
    if (it->second.cnt) (void)hipFree(it->second.cnt);
    g_skf_spaces.erase(it);
}

// strip width of a shape: gate/up pairs and RoPE heads take 128-row strips (64 for RoPE heads of 64 on small matrices); plain /
// residual outputs 64 (one partial-sum slot per strip, as the 128 x 256 kernel's: gemm_resid_partials)
static int skf_bn(int64_t N, int epi, int d) {
    if (epi == EPI_GATEUP) return 128;
    if (epi == EPI_QKV_ROPE) return (d == 64 && N / 128 < 64) ? 64 : 128;
    return 64;
}

// Which shapes run here, and in how many K slices (0: not here).  Who asks is the callers' matter (FL_GEMM_SKF, common.h: by default
// only a tensor-parallel rank's complete outputs).  d: head_dim (RoPE only).
int gemm_skf_plan(int64_t T, int64_t N, int64_t K, int epi, int d) {
    if (tune(TK_GEMM_SKF) <= 0 || T < 2 || T > 128 || K % F_BK || K / F_BK < 4 || N < 64) return 0;
    if (epi != EPI_F32 && epi != EPI_GATEUP && epi != EPI_RESID && epi != EPI_QKV_ROPE) return 0;
#ifndef FL_EXPERIMENTAL
    if (epi != EPI_F32) return 0;          // the five-launch layer's epilogues measured slower than the launches they replace: EXPERIMENTAL build only
#endif
    if (epi == EPI_GATEUP && N % 32) return 0;
    if (epi == EPI_RESID && N % 64) return 0;                       // (one partial-sum slot per 64-column strip, whole strips)
    if (epi == EPI_QKV_ROPE && ((d != 64 && d != 128) || N % 128)) return 0;
    const int bn = skf_bn(N, epi, d);
    const int64_t strips = (N + bn - 1) / bn, nk = K / F_BK;
    if (strips > kSkfMaxTiles) return 0;
    int ks = 1;
    if (epi != EPI_GATEUP) {
        const int forced = tune(TK_SKF_SPLIT);
        if (forced > 0) ks = (int)std::min<int64_t>(std::min(forced, 4), nk);
        else while (ks < 4 && strips * ks < 512 && nk / (ks + 1) >= 8) ks++;      // (gemm_skinny_ksplit's rule: cover the chip twice, >= 8 K tiles per slice)
    }
    // partial tiles of the launch must fit the stream's workspace
    const int64_t bm = T <= 32 ? 32 : (T <= 64 ? 64 : 128);
    while (ks > 1 && (size_t)strips * ks * bm * bn * 4 > kSkfPartBytes) ks--;
    return ks;
}

template <int BM, int NT, int NW, int NSTG, bool WNT, int NWM>
static int launch_skf_s(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K, int epi,
                        const float *row_scale, int ksplit, const SkfArgs &a) {
    constexpr int BN = NW * 16 * NT;
    constexpr size_t lds = (size_t)NSTG * (BM * 128 + BN * 128);
    static_assert(lds <= 160 * 1024, "LDS ring exceeds the CU");
    static_assert((NSTG - 2) * ((BM + BN) / 8 / (NW * NWM)) < 64, "vmcnt field");
    static_assert(lds >= (size_t)(BM + BM * NW) * 4 + 16, "the epilogue's scratch lives in the ring");
    auto kern = gemm_skf_kernel<BM, NT, NW, NSTG, WNT, NWM>;
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    const double bytes = ((double)N * K + (double)T * K) * 2.0;
    char tag[40];
    snprintf(tag, sizeof tag, "skf,%lldx%lld%s%s", (long long)N, (long long)K, ksplit > 1 ? ",sliced" : "",
             epi == EPI_RESID ? ",resid" : epi == EPI_QKV_ROPE ? ",rope" : epi == EPI_GATEUP ? ",glu" : "");
    Launcher LL = L; LL.tag = tag;
    const dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)ksplit, (unsigned)((T + BM - 1) / BM));
    return LL.launch(KC_GEMM_MFMA, bytes, 2.0 * T * N * K, kern, grid, dim3(NW * NWM * 64), lds,
                     (const bf16_t *)W, (const bf16_t *)x, bias, y, (int)T, (int)N, (int)K, epi, row_scale, ksplit, a);
}

template <int BM, int NT, int NW>
static int launch_skf_t(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K, int epi,
                        const float *row_scale, int ksplit, const SkfArgs &a) {
    constexpr int BN = NW * 16 * NT, STG = (BM + BN) * 128;
    // non-temporal W pieces: the rule of launch_skinny_t (up to 32 tokens, and on the large matrices at any length)
    const int ntm = tune(TK_SKINNY_NT);
    const bool nt = (ntm == 1 && (T <= 32 || N * K >= ((int64_t)32 << 20))) || ntm == 2;
    if constexpr (BM >= 64) {
        constexpr int NI = (BM + BN) / 8;
        constexpr int WM = NW == 4 ? 2 : (BM == 128 && NI % 8 == 0 ? 4 : 2);
        static_assert(NI % (NW * WM) == 0 && BM % (16 * WM) == 0, "wave rows must divide the stage and the token tiles");
        constexpr int kStg = 4 * STG <= 160 * 1024 ? 4 : 3;
        return nt ? launch_skf_s<BM, NT, NW, kStg, true, WM>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit, a)
                  : launch_skf_s<BM, NT, NW, kStg, false, WM>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit, a);
    } else {
        return nt ? launch_skf_s<BM, NT, NW, 4, true, 1>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit, a)
                  : launch_skf_s<BM, NT, NW, 4, false, 1>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit, a);
    }
}

int launch_gemm_skf(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K, int epi,
                    const float *row_scale, int ksplit, const ResidEpi *resid, const RopeEpi *rope) {
    const int d = rope ? rope->d : 0;
    if (T < 2 || T > 128 || K % F_BK || K / F_BK < 4 || N < 64 || ksplit < 1 || ksplit > 4 || K / F_BK < ksplit)
        FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_skf: shape / K slices not supported");
    if ((epi == EPI_RESID) != (resid != nullptr) || (resid && (!resid->h || !resid->w || !resid->xn || !resid->part || N % 64 || resid->np < N / 64 || resid->np % 4)))
        FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_skf: the residual epilogue takes its operands, whole 64-column strips");
    if ((epi == EPI_QKV_ROPE) != (rope != nullptr) ||
        (rope && ((d != 64 && d != 128) || N % 128 || N != (int64_t)(rope->H + 2 * rope->Hkv) * d || rope->col_base || !rope->cos_tab || !rope->sin_tab || !rope->q_out ||
                  (rope->seqs ? !rope->v_transposed : (!rope->st || !rope->k_cache || !rope->v_cache)))))
        FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_skf: the RoPE epilogue takes its operands, the whole q | k | v matrix, head_dim 64 / 128");
    if (epi == EPI_GATEUP && (bias || ksplit != 1 || N % 32)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_skf: gate/up takes no bias and the whole K");
    if (epi != EPI_F32 && epi != EPI_GATEUP && epi != EPI_RESID && epi != EPI_QKV_ROPE) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_skf: epilogue %d", epi);
#ifndef FL_EXPERIMENTAL
    if (epi != EPI_F32) FL_FAIL(FL_ERR_UNSUPPORTED, "gemm_skf: the gate/up, residual and RoPE epilogues (the five-launch layer: measured slower) are in the EXPERIMENTAL build only");
#endif
    SkfArgs a;
    if (resid) a.re = *resid;
    if (rope) { a.ro = *rope; a.ro.on = 1; }
    a.rsp = L.rsp;
    const int bn = skf_bn(N, epi, d);
    if (ksplit > 1) {
        const int64_t bm = T <= 32 ? 32 : (T <= 64 ? 64 : 128), strips = (N + bn - 1) / bn;
        if (strips > kSkfMaxTiles || (size_t)strips * ksplit * bm * bn * 4 > kSkfPartBytes) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_skf: partial tiles exceed the workspace");
        FL_TRY(skf_space(L.stream, &a.ws));
    }
#ifdef FL_EXPERIMENTAL
#define FL_SKF(BMV)                                                                                                       \
    if (bn == 128) return launch_skf_t<BMV, 2, 4>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit, a);                  \
    if (epi == EPI_QKV_ROPE) return launch_skf_t<BMV, 2, 2>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit, a);        \
    return launch_skf_t<BMV, 1, 4>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit, a);
#else
#define FL_SKF(BMV) return launch_skf_t<BMV, 1, 4>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit, a);
#endif
    if (T <= 32) { FL_SKF(32) }
    if (T <= 64) { FL_SKF(64) }
    FL_SKF(128)
#undef FL_SKF
}

}  // namespace fl
