// k_attn_mfma.hip -- bf16 attention on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16), for the
// KV-cached single sequence of this path: decode (one token, G = H/Hkv query heads per kv head) and
// prefill (16 query tokens of one head per wave).  Flash-style: scores never leave registers.
//
// One wave processes 32 keys per step with 16 "query columns" (decode: the G heads of the kv group,
// padded to 16; prefill: 16 consecutive tokens of one head):
//   S^T = K . Q^T   A = K rows straight from the cache [key][d] (16 B per lane), B = Q fragments kept
//                   in registers; the 16 MFMA rows of the two row-tiles are mapped to keys as
//                   key = base + 8*(i>>2) + 4*tile + (i&3), so that afterwards lane (col q, group g4)
//                   holds the scores of keys base + 8*g4 + 0..7 -- exactly the A-operand layout of ...
//   O  += P . V     A = P (softmaxed scores, packed to bf16 in registers, no LDS, no transposition),
//                   B = V^T read 16 B per lane from the TRANSPOSED value cache [d][key].
// The value cache is therefore stored transposed in bf16 mode ([Hkv][d][seq_alloc]); the append of one
// token is d strided 2-byte stores (done by the QKV kernel's epilogue), a prefill appends rows of keys.
// Online softmax runs per column on the lanes that own it (max over the lane's 8 keys + two xor
// shuffles across the four lane groups); the rescale factor reaches the O accumulators (whose rows
// are columns of S^T) through four wave shuffles.
//
// Decode: grid (Hkv, nsplit), 4 waves per workgroup, 32-key tiles round-robin over the waves; the
// cross-wave and cross-split merge is shared with the VALU kernel (attn_common.h).
// Prefill: each wave owns one (16-token tile, head); causal + sliding-window mask per column.
#include <stdlib.h>

#include <algorithm>
#include <atomic>

#include "attn_mfma.h"

namespace fl {

// ------------------------------------------------------------------------------- decode
template <int D, int GMAX, int NW>
__device__ __forceinline__ void attn_decode_mfma_body(const bf16_t *__restrict__ q, const bf16_t *__restrict__ kc,
                                             const bf16_t *__restrict__ vT, const StepState *__restrict__ st,
                                             float *__restrict__ part_m, float *__restrict__ part_l,
                                             float *__restrict__ part_o, unsigned *__restrict__ counters,
                                             bf16_t *__restrict__ out, int H, int Hkv, int seq_alloc,
                                             float scale, int nsplit, float *lds, int *is_last) {
    const int hk = blockIdx.x, split = blockIdx.y;
    const int G = H / Hkv, hq0 = hk * G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g4 = lane >> 4;
    const int S = (int)st->len + 1;
    int per = (S + nsplit - 1) / nsplit;
    per = (per + 32 * NW - 1) / (32 * NW) * (32 * NW);
    const int lo = split * per, hi = min(S, lo + per);

    bf16x8 qf[D / 32];
#pragma unroll
    for (int dk = 0; dk < D / 32; dk++) {
        if (i < G) qf[dk] = ld_bf16x8(q + (size_t)(hq0 + i) * D + dk * 32 + g4 * 8);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) qf[dk][j] = (__bf16)0.f;
        }
    }
    MfmaAttnState<D> s; s.init();
    const bf16_t *kb = kc + (size_t)hk * seq_alloc * D;
    const bf16_t *vb = vT + (size_t)hk * D * seq_alloc;
    // A wave's tiles are 32 * NW keys apart and every fragment is one trip to HBM.  The next tile's K fragments are
    // requested before the current tile is computed (its V fragments are requested inside the step, behind the K.Q^T
    // MFMAs, and land under the softmax); both sets of V would not fit the register budget of three waves per SIMD.
    constexpr int STEP = 32 * NW;
    struct KregVglobal {
        bf16x8 k[2][D / 32];
        const bf16_t *vT; int ldv, kbase, i, g4;
        __device__ void load_k(const bf16_t *kb_, int kbase_, int i_, int g4_) {
#pragma unroll
            for (int dk = 0; dk < D / 32; dk++)
#pragma unroll
                for (int t = 0; t < 2; t++) k[t][dk] = ld_bf16x8(kb_ + (size_t)(kbase_ + 8 * (i_ >> 2) + (i_ & 3) + 4 * t) * D + dk * 32 + g4_ * 8);
        }
        __device__ bf16x8 k_frag(int tile, int dk) const { return k[tile][dk]; }
        __device__ bf16x8 v_frag(int db) const { return ld_bf16x8(vT + (size_t)(db * 16 + i) * ldv + kbase + 8 * g4); }
    };
    if constexpr (NW > 8) {               // 16-wave workgroups have 128 VGPRs per lane: no room for a second K set,
        int kbase = lo + 32 * wave;       // but the FIRST tile's K and V^T fragments are requested together here too
        if (kbase < hi) {
            RegKV<D> r0;
            r0.load(kb, vb, seq_alloc, kbase, i, g4);
            attn_tile<D>(s, qf, r0, kbase, 0, lo, hi, scale, lane);
            kbase += STEP;
        }
        for (; kbase < hi; kbase += STEP) {
            const GlobalKV<D> kv{kb, vb, seq_alloc, kbase, i, g4};
            attn_tile<D>(s, qf, kv, kbase, 0, lo, hi, scale, lane);
        }
    } else {
    KregVglobal ra, rb;
    ra.vT = rb.vT = vb; ra.ldv = rb.ldv = seq_alloc; ra.i = rb.i = i; ra.g4 = rb.g4 = g4;
    int kbase = lo + 32 * wave;
    if (kbase < hi) {
        // the FIRST tile's K and V^T fragments are requested together (one round trip instead of two): at the context
        // lengths where a wave has a single tile (S <= 128 * nsplit) that is the whole chain
        RegKV<D> r0;
        r0.load(kb, vb, seq_alloc, kbase, i, g4);
        if (kbase + STEP < hi) rb.load_k(kb, kbase + STEP, i, g4);
        attn_tile<D>(s, qf, r0, kbase, 0, lo, hi, scale, lane);
        kbase += STEP;
    }
    while (kbase < hi) {
        if (kbase + STEP < hi) ra.load_k(kb, kbase + STEP, i, g4);
        rb.kbase = kbase;
        attn_tile<D>(s, qf, rb, kbase, 0, lo, hi, scale, lane);
        kbase += STEP;
        if (kbase >= hi) break;
        if (kbase + STEP < hi) rb.load_k(kb, kbase + STEP, i, g4);
        ra.kbase = kbase;
        attn_tile<D>(s, qf, ra, kbase, 0, lo, hi, scale, lane);
        kbase += STEP;
    }
    }

    mfma_state_to_lds<D, GMAX>(s, lds, wave, G, lane);
    __syncthreads();
    decode_tail<bf16_t, D, GMAX, NW>(lds, is_last, G, hq0, hk, split, nsplit, part_m, part_l, part_o, counters, out);
}

// Prefetch workgroups (blockIdx.y >= nsplit): every 128-byte line of the next launch's weights is touched once and dropped,
// while the attention workgroups wait on their own round trips and HBM idles.  The lines land in the Infinity Cache and in
// the L2 of the XCD that touched them -- so a prefetch workgroup takes the chunks [j chunk, (j + 1) chunk) whose consumer
// workgroup (j mod its grid; workgroups go round the XCDs in launch order) will run on ITS XCD (HW_REG_XCC_ID).  They take no
// part in the splits' combine.  Yardstick: the o_proj GEMV with its weights cache-resident (FL_OP_HOT=1, tools/skinny_probe.py)
// runs in 4.5 us instead of 7.7 (Mistral-7B), TinyLlama's 2048 x 2048 in ~3 instead of 4.4.
__device__ __forceinline__ void prefetch_chunks(float *dump, const void *pf, unsigned chunk_lines, unsigned nchunks, unsigned row_lines, unsigned row_take, unsigned w, unsigned P, unsigned nthr, unsigned delay) {
    if (delay) { for (unsigned d = 0; d < delay; d++) __builtin_amdgcn_s_sleep(16); }   // (x 1024 cycles: let the attention's own K / V requests go first)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    const unsigned Px = max(1u, (P + 7) >> 3), r = (w >> 3) % Px;
    const unsigned char *base = reinterpret_cast<const unsigned char *>(pf);
    for (unsigned j = xcc + 8u * r; j < nchunks; j += 8u * Px) {
        const unsigned char *cb = base + (size_t)j * chunk_lines * 128;
        // (the first row_take of every row's row_lines lines: each consumer wave then has the same share left to fetch)
        const unsigned rows = chunk_lines / row_lines;
        for (unsigned l = threadIdx.x; l < rows * row_take; l += nthr) {
            const unsigned rr = l / row_take, ll = l - rr * row_take;
            // an LDS-DMA dword per lane: no register destination (an asm load into a scratch VGPR lands after hipcc has given that
            // register to the next address -- a memory fault, seen); the 256 bytes fall on the attention's unused LDS state
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(cb + ((size_t)rr * row_lines + ll) * 128),
                                             (__attribute__((address_space(3))) void *)dump, 4, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int D, int GMAX, int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_mfma_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ kc,
                                                               const bf16_t *__restrict__ vT, const StepState *__restrict__ st,
                                                               float *__restrict__ part_m, float *__restrict__ part_l,
                                                               float *__restrict__ part_o, unsigned *__restrict__ counters,
                                                               bf16_t *__restrict__ out, int H, int Hkv, int seq_alloc,
                                                               float scale, int nsplit, const void *__restrict__ pf, unsigned pf_chunk_lines, unsigned pf_nchunks,
                                                               unsigned pf_row_lines, unsigned pf_row_take, unsigned pf_delay) {
    __shared__ float lds[decode_lds_floats<D, GMAX, NW>()];
    __shared__ int is_last;
    if ((int)blockIdx.y >= nsplit) {                                // workgroup-uniform
        prefetch_chunks(lds, pf, pf_chunk_lines, pf_nchunks, pf_row_lines, pf_row_take, (blockIdx.y - nsplit) * gridDim.x + blockIdx.x, (gridDim.y - nsplit) * gridDim.x, NW * 64, pf_delay);
        return;
    }
    attn_decode_mfma_body<D, GMAX, NW>(q, kc, vT, st, part_m, part_l, part_o, counters, out, H, Hkv, seq_alloc, scale, nsplit, lds, &is_last);
}

// The same for B sequences in one launch (blockIdx.z = sequence): each reads its own cache, length and
// split scratch from its SeqRef; q / out are [B][H*D].  grid.y = the largest split count of the batch.
template <int D, int GMAX, int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_mfma_batch_kernel(const bf16_t *__restrict__ q, const SeqRef *__restrict__ seqs,
                                                                     size_t kv_layer_off, bf16_t *__restrict__ out, int H, int Hkv,
                                                                     float scale, int nsplit_cap) {
    __shared__ float lds[decode_lds_floats<D, GMAX, NW>()];
    __shared__ int is_last;
    SeqRef sq = seqs[blockIdx.z];
    sq.nsplit = min(sq.nsplit, nsplit_cap);                         // a batch already fills the chip: fewer, longer splits
    if ((int)blockIdx.y >= sq.nsplit) return;                       // workgroup-uniform
    const size_t off = kv_layer_off * (size_t)sq.seq_alloc;
    attn_decode_mfma_body<D, GMAX, NW>(q + (size_t)blockIdx.z * H * D, reinterpret_cast<const bf16_t *>(sq.k) + off,
                                       reinterpret_cast<const bf16_t *>(sq.v) + off, sq.st, sq.part_m, sq.part_l, sq.part_o,
                                       sq.counters, out + (size_t)blockIdx.z * H * D, H, Hkv, sq.seq_alloc, scale, sq.nsplit, lds, &is_last);
}

template <int D, int GMAX, int NW>
static int launch_decode_mfma_t(Launcher &L, const void *q, const void *kc, const void *vT, const StepState *st, void *out,
                                const AttnScratch &sc, int64_t H, int64_t Hkv, int64_t seq_alloc, float scale) {
    // Prefetch workgroups (FL_ATTN_PREFETCH=1; off by default): ~1 KiB per thread (eight lines), as extra rows of the grid.
    // Measured, round 3 (tools/decode_probe.py): the consumer gets what the yardstick promised -- Mistral-7B o_proj 7.35 -> 4.4 us
    // (7.6 TB/s of its weights), Qwen2-7B 6.2 -> 4.4 -- and the attention launch pays it back: 7.3 -> 9.5 us (Mistral, S = 530),
    // 11.9 -> 13.7 (Qwen2, S = 4100): its chain of round trips (K, V, the splits' partials) slows down under the prefetch traffic
    // by about what the consumer gains, whatever share of the rows is taken (FL_ATTN_PREFETCH_PCT 60 / 75 / 100) and however late
    // the prefetchers start (FL_ATTN_PREFETCH_DELAY).  Tokens/s: Mistral-7B 369.8 -> 371.9 (four runs each), Qwen2-7B 360.1 ->
    // 356.9, TinyLlama-1.1B unchanged (its o_proj is not memory-bound: 4.2-4.4 us either way).  (read per call: tests switch it)
#ifdef FL_EXPERIMENTAL
    const int pf_on = tune(TK_ATTN_PREFETCH);
#else
    const int pf_on = 0;                                           // (measured neutral: EXPERIMENTAL build only)
#endif
    const int pf_per_thread = std::max(1, tune(TK_ATTN_PREFETCH_LINES));
    const int pf_pct = std::min(100, std::max(1, tune(TK_ATTN_PREFETCH_PCT)));
    const int pf_delay = std::max(0, tune(TK_ATTN_PREFETCH_DELAY));
    unsigned pf_rows = 0, chunk_lines = 0, nchunks = 0, row_lines = 1, row_take = 1;
    if (pf_on && sc.prefetch && sc.prefetch_chunk >= 128 && sc.prefetch_bytes >= sc.prefetch_chunk && sc.prefetch_row >= 128 && sc.prefetch_chunk % sc.prefetch_row == 0) {
        chunk_lines = (unsigned)(sc.prefetch_chunk / 128);
        row_lines = (unsigned)(sc.prefetch_row / 128);
        row_take = std::max(1u, row_lines * (unsigned)pf_pct / 100u);
        nchunks = (unsigned)std::min<int64_t>(sc.prefetch_bytes / sc.prefetch_chunk, 1 << 20);
        const int64_t lines = (int64_t)chunk_lines / row_lines * row_take * nchunks, per_wg = (int64_t)NW * 64 * pf_per_thread;
        int64_t wgs = (lines + per_wg - 1) / per_wg;
        pf_rows = (unsigned)((wgs + Hkv - 1) / Hkv);
        // a multiple of eight workgroups in the whole launch: the dispatcher's walk over the XCDs carries on from launch to
        // launch, and the consumer's workgroup b is assumed on XCD b mod 8 (the prefetchers read their own XCC id)
        while ((((unsigned)sc.nsplit + pf_rows) * (unsigned)Hkv) % 8) pf_rows++;
    }
    dim3 grid((unsigned)Hkv, (unsigned)sc.nsplit + pf_rows, 1);
    double kvbytes = 2.0 * (double)sc.kv_len_hint * Hkv * D * 2;
    return L.launch(KC_ATTN_DECODE, kvbytes, 4.0 * (double)sc.kv_len_hint * H * D, attn_decode_mfma_kernel<D, GMAX, NW>, grid,
                    dim3(NW * 64), 0, (const bf16_t *)q, (const bf16_t *)kc, (const bf16_t *)vT, st, sc.part_m, sc.part_l,
                    sc.part_o, sc.counters, (bf16_t *)out, (int)H, (int)Hkv, (int)seq_alloc, scale, sc.nsplit, sc.prefetch, chunk_lines, nchunks, row_lines, row_take, (unsigned)pf_delay);
}

bool attn_mfma_supported(int dtype, int64_t H, int64_t Hkv, int64_t d) {
    // decode packs the G = H/Hkv heads into the 16 MFMA columns; prefill runs one wave per head of the group
    return dtype == FL_DTYPE_BF16 && (d == 64 || d == 128) && Hkv > 0 && H % Hkv == 0 && H / Hkv <= 8;
}

int launch_attn_decode_mfma(Launcher &L, const void *q, const void *k_cache, const void *v_cache_T, const StepState *st,
                            void *out, const AttnScratch &sc, int64_t H, int64_t Hkv, int64_t d, int64_t seq_alloc,
                            float scale) {
    const int G = (int)(H / Hkv);
    if (sc.nsplit > 64) FL_FAIL(FL_ERR_BAD_ARGUMENT, "attention: at most 64 splits");
    // nsplit == 1: one wide workgroup per kv head (no cross-workgroup combine); else 4-wave workgroups
    const bool wide = sc.nsplit == 1;
#define FL_GO(DD)                                                                                                   \
    if (G <= 4) return wide ? launch_decode_mfma_t<DD, 4, 16>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale)  \
                            : launch_decode_mfma_t<DD, 4, 4>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale);  \
    if (G <= 8) return wide ? launch_decode_mfma_t<DD, 8, 16>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale)  \
                            : launch_decode_mfma_t<DD, 8, 4>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale);  \
    return wide ? launch_decode_mfma_t<DD, 16, 8>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale)              \
                : launch_decode_mfma_t<DD, 16, 4>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale);
    if (d == 128) { FL_GO(128) }
    if (d == 64) { FL_GO(64) }
#undef FL_GO
    FL_FAIL(FL_ERR_UNSUPPORTED, "mfma attention: head_dim %lld", (long long)d);
}

// batched decode: every sequence of the batch uses 4-wave split workgroups (its own nsplit >= 1)
int launch_attn_decode_mfma_batch(Launcher &L, const void *q, const SeqRef *seqs_dev, int B, int max_nsplit, size_t kv_layer_off,
                                  void *out, int64_t H, int64_t Hkv, int64_t d, float scale, double kv_bytes_hint) {
    const int G = (int)(H / Hkv);
    if (max_nsplit > 64 || max_nsplit < 1) FL_FAIL(FL_ERR_BAD_ARGUMENT, "attention: 1..64 splits");
    // measured (Mistral-7B, 8 sequences at S ~ 600), with a release fence per split workgroup: 128 workgroups (2 splits)
    // 15.3 us, 448 (7 splits) 24.3 us; with write-through publication (attn_common.h): 128 -> 11.4 us, 256 -> 10.9, 512 -> 12.2
    const int wg_cap = tune(TK_ATTN_BATCH_WGS);
    const int cap = (int)std::max<int64_t>(1, wg_cap / (Hkv * B));
    max_nsplit = std::min(max_nsplit, cap);
    dim3 grid((unsigned)Hkv, (unsigned)max_nsplit, (unsigned)B);
#define FL_GOB(DD, GM)                                                                                                          \
    return L.launch(KC_ATTN_DECODE, kv_bytes_hint, 0, attn_decode_mfma_batch_kernel<DD, GM, 4>, grid, dim3(256), 0, (const bf16_t *)q, \
                    seqs_dev, kv_layer_off, (bf16_t *)out, (int)H, (int)Hkv, scale, cap);
    if (d == 128) { if (G <= 4) { FL_GOB(128, 4) } if (G <= 8) { FL_GOB(128, 8) } FL_GOB(128, 16) }
    if (d == 64) { if (G <= 4) { FL_GOB(64, 4) } if (G <= 8) { FL_GOB(64, 8) } FL_GOB(64, 16) }
#undef FL_GOB
    FL_FAIL(FL_ERR_UNSUPPORTED, "mfma attention: head_dim %lld", (long long)d);
}

// ------------------------------------------------------------------------------- prefill
// grid (ceil(T / (16*TT)), Hkv); workgroup = G*TT waves: wave w serves query head g = w % G of kv head
// blockIdx.y for the 16 tokens of sub-tile w / G.  All waves walk the same key tiles, which are staged
// ONCE per workgroup in LDS (K tile + V^T tile, 16 KB per 32 keys at d = 128, double-buffered LDS-DMA),
// so every K/V byte fetched from L2 feeds G*TT x 16 query rows instead of 16.
// Mask (App. A.5): cached prefix [0,len) visible; in-call key j visible to token t iff j <= t and
// (window < 0 or j + window >= t).
__device__ inline void glds16(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate)
__device__ inline void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // always safe
    }
}

// every wave issues the same number of 1-KiB pieces, per = ceil((NK + NV) / nwv) (wrapping: a duplicate piece writes
// the same bytes), so that a counted vmcnt can tell which tiles have landed
template <int D>
__device__ inline void stage_kv(const bf16_t *__restrict__ kb, const bf16_t *__restrict__ vT, int ldv, int kbase,
                                unsigned char *buf, int wave, int nwv, int lane, int per) {
    constexpr int CPR = D / 8, NK = CPR / 2, NV = D / 16;          // wave-instructions (1 KiB each) per tile
    unsigned char *kt = buf, *vt = buf + 32 * D * 2;
    for (int k = 0; k < per; k++) {
        const int e = (wave + k * nwv) % (NK + NV);
        if (e < NK) {
            const int p = e * 64 + lane, row = p / CPR, pc = p % CPR, c = pc ^ (row & (CPR - 1));
            glds16(kb + (size_t)(kbase + row) * D + c * 8, kt + e * 1024);
        } else {
            const int ev = e - NK, p = ev * 64 + lane, row = p >> 2, pc = p & 3, c = pc ^ ((row >> 2) & 3);
            glds16(vT + (size_t)row * ldv + kbase + c * 8, vt + ev * 1024);
        }
    }
}

template <int D>
__global__ __launch_bounds__(1024) void attn_prefill_mfma_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ kc,
                                                                const bf16_t *__restrict__ vT, const StepState *__restrict__ st,
                                                                bf16_t *__restrict__ out, int T, int H, int Hkv, int seq_alloc,
                                                                float scale, int window, int TT, int nst, int ks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // nst x (K tile | V^T tile)
    constexpr int TILE = 2 * 32 * D * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int i = lane & 15, g4 = lane >> 4;
    const int G = H / Hkv, hk = blockIdx.y;
    // ks == 2 (short prompts): two waves per (head, token sub-tile) take alternate key tiles and merge at the end --
    // a key step is a chain of dependent latencies, and this puts two such chains on every SIMD
    const int wq = wave % (G * TT), half = wave / (G * TT);
    const int hq = hk * G + wq % G;
    const int tb0 = blockIdx.x * TT * 16;                        // first token of the workgroup
    const int t0 = tb0 + (wq / G) * 16;                          // first token of this wave
    const int len = (int)st->len;
    const int t = t0 + i;                                        // this lane's column
    const bool col_ok = t < T;

    bf16x8 qf[D / 32];
#pragma unroll
    for (int dk = 0; dk < D / 32; dk++) {
        if (col_ok) qf[dk] = ld_bf16x8(q + ((size_t)t * H + hq) * D + dk * 32 + g4 * 8);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) qf[dk][j] = (__bf16)0.f;
        }
    }
    // keys cached before the API call are an unmasked prefix; the causal + sliding-window mask runs over
    // the call's own tokens even when the library cut the call into chunks (call0 <= len)
    const int c0 = (int)st->call0;
    int lo = c0;
    if (window >= 0 && len + t - window > c0) lo = len + t - window;
    const int lo_q = col_ok ? lo : 0, hi_q = col_ok ? len + t + 1 : 0;
    const int pre_hi = col_ok ? c0 : 0;
    // this wave's useful key range (uniform per wave): tiles outside are skipped, staging is not
    int wstart = 0;
    if (c0 == 0 && window >= 0 && len + t0 - window > 0) wstart = ((len + t0 - window) / 32) * 32;
    const int wend = t0 < T ? len + min(T, t0 + 16) : 0;

    MfmaAttnState<D> s; s.init();
    const bf16_t *kb = kc + (size_t)hk * seq_alloc * D;
    const bf16_t *vb = vT + (size_t)hk * D * seq_alloc;
    int kstart = 0;
    if (c0 == 0 && window >= 0 && len + tb0 - window > 0) kstart = ((len + tb0 - window) / 32) * 32;
    const int kend = len + min(T, tb0 + 16 * TT);
    const int nsteps = (kend - kstart + 31) / 32;
    // ring of nst tiles, nst - 1 of them in flight
    constexpr int NI = D / 16 + D / 16;
    const int per = (NI + nwv - 1) / nwv;
    if (ks == 2) {
        // super-steps of two tiles (buffers [stage][half]); one barrier per super-step, two LDS stages
        const int nsuper = (nsteps + 1) / 2;
        auto stage2 = [&](int ss, int stg) {
            const int k0 = kstart + 64 * ss;
            stage_kv<D>(kb, vb, seq_alloc, k0, lds + (stg * 2) * TILE, wave, nwv, lane, per);
            if (k0 + 32 < kend) stage_kv<D>(kb, vb, seq_alloc, k0 + 32, lds + (stg * 2 + 1) * TILE, wave, nwv, lane, per);
        };
        stage2(0, 0);
        for (int ss = 0; ss < nsuper; ss++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (ss + 1 < nsuper) stage2(ss + 1, (ss + 1) & 1);
            const int kbase = kstart + 64 * ss + 32 * half;
            if (kbase < kend && kbase + 32 > wstart && kbase < wend) {               // wave-uniform
                const unsigned char *cur = lds + ((ss & 1) * 2 + half) * TILE;
                const LdsKV<D> kv{cur, cur + 32 * D * 2, i, g4};
                attn_tile<D>(s, qf, kv, kbase, pre_hi, lo_q, hi_q, scale, lane);
            }
        }
        // merge the two key halves through LDS (the tiles are dead after the barrier)
        __syncthreads();
        float *slab = reinterpret_cast<float *>(lds) + (size_t)wq * (16 * D + 32);
        float lt1 = s.l;
        lt1 += __shfl_xor(lt1, 16, 64);
        lt1 += __shfl_xor(lt1, 32, 64);
        if (half == 1) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int db = 0; db < D / 16; db++) slab[(4 * g4 + r) * D + db * 16 + i] = s.O[db][r];
            if (g4 == 0) { slab[16 * D + i] = s.m; slab[16 * D + 16 + i] = lt1; }
        }
        __syncthreads();
        if (half == 1) return;
        const float m1 = slab[16 * D + i], l1 = slab[16 * D + 16 + i];
        const float mm = fmaxf(s.m, m1);
        const float w0 = s.m == -INFINITY ? 0.f : __expf(s.m - mm), w1 = m1 == -INFINITY ? 0.f : __expf(m1 - mm);
        const float ltot = lt1 * w0 + l1 * w1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int qq = 4 * g4 + r;
            const float lr = __shfl(ltot, qq, 64), a0 = __shfl(w0, qq, 64), a1 = __shfl(w1, qq, 64);
            const int tt = t0 + qq;
            if (tt < T) {
                const float inv = 1.0f / lr;
                bf16_t *o = out + ((size_t)tt * H + hq) * D;
#pragma unroll
                for (int db = 0; db < D / 16; db++)
                    o[db * 16 + i] = float_to_bf16_bits((s.O[db][r] * a0 + slab[qq * D + db * 16 + i] * a1) * inv);
            }
        }
        return;
    }
    for (int a = 0; a < nst - 1 && a < nsteps; a++) stage_kv<D>(kb, vb, seq_alloc, kstart + 32 * a, lds + a * TILE, wave, nwv, lane, per);
    for (int sidx = 0; sidx < nsteps; sidx++) {
        wait_vmcnt(per * min(nst - 2, nsteps - 1 - sidx));       // tile sidx has landed; the younger ones may still fly
        __builtin_amdgcn_s_barrier();
        const int kbase = kstart + 32 * sidx;
        if (sidx + nst - 1 < nsteps)
            stage_kv<D>(kb, vb, seq_alloc, kbase + 32 * (nst - 1), lds + ((sidx + nst - 1) % nst) * TILE, wave, nwv, lane, per);
        if (kbase + 32 > wstart && kbase < wend) {               // wave-uniform
            const unsigned char *cur = lds + (sidx % nst) * TILE;
            const LdsKV<D> kv{cur, cur + 32 * D * 2, i, g4};
            attn_tile<D>(s, qf, kv, kbase, pre_hi, lo_q, hi_q, scale, lane);
        }
    }

    float lt = s.l;
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int qq = 4 * g4 + r;
        const float lr = __shfl(lt, qq, 64);
        const int tt = t0 + qq;
        if (tt < T) {
            const float inv = 1.0f / lr;
            bf16_t *o = out + ((size_t)tt * H + hq) * D;
#pragma unroll
            for (int db = 0; db < D / 16; db++) o[db * 16 + i] = float_to_bf16_bits(s.O[db][r] * inv);
        }
    }
}

// ------------------------------------------------------------------------------- prefill, 32 query rows per wave
// The 16-row kernel above re-reads every staged K/V tile from LDS once per 16 query rows and does its softmax
// bookkeeping per 16 x 32 scores: LDS bandwidth and VALU, not the matrix pipe, set its pace (MfmaUtil 12 %).  Here a
// wave owns 32 tokens of one query head and both products run "swapped" on v_mfma_f32_32x32x16_bf16 so that a query
// row lives on a lane (pair) from start to finish:
//   S^T = K . Q^T     A = K rows from the LDS tile, B = Q^T fragments kept in registers (8 x bf16x8 at d = 128).
//                     MFMA row m carries key pi(m) = m with bits 2 and 3 swapped, so lane (q, hi) ends up with the
//                     scores of keys 8*hi + 0..7 (registers 0..7) and 16 + 8*hi + 0..7 (registers 8..15) -- exactly
//                     the B-operand layout of the second product, no cross-lane traffic.
//   O^T += V^T . P^T  A = V^T rows (d) straight from the transposed-V tile, B = P^T = the exponentiated scores packed
//                     to bf16 in place.  O^T keeps q on the lanes: max / sum / rescale are per-lane scalars (one
//                     xor-32 shuffle joins the two half-rows), the rescale is skipped when no row's max moved.
// Scores are kept in the exp2 domain (scale * log2 e folded into one FMA per score).  K/V tiles (32 keys) arrive by
// LDS-DMA into a 3-deep ring, two tiles ahead, behind a counted vmcnt and one raw barrier per tile; tiles that are
// fully visible to all 32 rows of a wave skip the mask arithmetic.
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned short ushort4v __attribute__((ext_vector_type(4)));

// KS2 (key split, prompts too short to balance otherwise): a (head, 32-token block) is served by TWO waves that take alternate
// key tiles -- a step stages two tiles -- and meet through LDS at the end: the causal chain of the longest block, which sets the
// launch's duration when there are too few (block, kv head) items for the snake to even out, is half as long, and the 32-token
// blocks make twice as many items.  (Mistral-7B T = 2048: 64 blocks x 8 kv heads = 512 items, two balanced rounds.)
// KSF = 4 (since round 4): FOUR waves per (head, token block) on every fourth key tile, two ring slots of four tiles -- with subgroups of
// two query heads (gsub = 2) the items double again: prompts below ~2048 tokens, whose (block, kv head) items are too few for the
// snake, halve the longest block's chain once more and fill the chip (Mistral-7B T = 1024: 512 items instead of 256).
template <int D, int NW, int KSF>
__global__ __launch_bounds__(NW * 64) void attn_prefill32_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ kc,
                                                                 const bf16_t *__restrict__ vT, const StepState *__restrict__ st,
                                                                 bf16_t *__restrict__ out, int T, int H, int Hkv, int seq_alloc,
                                                                 float scale_log2e, int window, int paired, int gsub) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // NSLOT x KSF x (K tile | V^T tile)
    constexpr bool KS2 = KSF > 1;                                            // (the key-split form: partner waves meet through LDS)
    constexpr int TILE = 2 * 32 * D * 2, SLOT = KSF * TILE, NSLOT = KSF == 4 ? 2 : 3;
    constexpr int NK = D / 16, NV = D / 16, NI = NK + NV;                  // 1-KiB wave-instructions per tile
    constexpr int PER = (NI + NW - 1) / NW;                                // issued by every wave (wrapping: duplicates are benign)
    constexpr int CPR = D / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m32 = lane & 31, hi = lane >> 5;
    // gsub > 0 (groups of 7 or 8 query heads, which leave no room for wave pairs): a workgroup serves gsub of a kv head's G query
    // heads, the group is dealt as nsg = ceil(G / gsub) items (Qwen2-7B: 4 + 3 heads; one wave pair of the second item idles)
    const int G = H / Hkv, Gs = gsub > 0 ? gsub : G, nsg = (G + Gs - 1) / Gs;
    const int TB = NW / Gs / KSF;                                           // 32-token blocks per workgroup
    const int sub = wave / Gs, tok_sub = sub / KSF, ksub = sub % KSF;      // this wave's token block and key-tile parity
    const int len = (int)st->len;
    const int c0 = (int)st->call0;
    const int krow = (m32 & 19) | ((m32 & 4) << 1) | ((m32 & 8) >> 1);      // pi(m): bits 2 and 3 swapped
    // Causal: token block b walks ~b+1 key tiles.  paired = 1: a workgroup takes block nb-1-x and then block x, so every
    // workgroup walks the same number of tiles (one balanced round when nb/2 x Hkv covers the chip); 0: the long blocks
    // are dispatched first and the short ones fill the tail; 2 ("snake", long prompts): gridDim.x persistent workgroups
    // deal themselves the (block, kv head) items -- longest first -- boustrophedon: round r goes 0..P-1, round r+1
    // P-1..0, so consecutive rounds add up to the same work for everyone whatever nb x Hkv is (4608 tokens of Qwen2-7B are
    // 288 paired workgroups: 1.125 rounds that took as long as two).
    const int nb = (T + 32 * TB - 1) / (32 * TB);
    const int nhs = Hkv * nsg;                                              // (kv head, head subgroup) pairs
    const int npass = paired == 2 ? (nb * nhs + (int)gridDim.x - 1) / (int)gridDim.x : (paired ? 2 : 1);
    for (int pass = 0; pass < npass; pass++) {
    int blk, hs;
    if (paired == 2) {
        const int item = pass * (int)gridDim.x + ((pass & 1) ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x);
        if (item >= nb * nhs) break;                                        // (only the last round can be short)
        blk = nb - 1 - item / nhs; hs = item % nhs;
    } else {
        hs = blockIdx.y;
        blk = pass == 0 ? nb - 1 - (int)blockIdx.x : (int)blockIdx.x;
        if (pass == 1 && blk >= nb - 1 - (int)blockIdx.x) break;            // odd count: the middle block was pass 0
    }
    const int hk = hs / nsg, g = (hs % nsg) * Gs + wave % Gs;             // kv head; query head within its group
    const int hq = hk * G + min(g, G - 1);
    const bf16_t *kb = kc + (size_t)hk * seq_alloc * D;
    const bf16_t *vb = vT + (size_t)hk * D * seq_alloc;
    const int tb0 = blk * TB * 32;
    const int t0 = tb0 + tok_sub * 32;
    const bool wave_on = wave < TB * Gs * KSF && g < G;                     // NW need not be a multiple of Gs; a short last subgroup
    const int t = t0 + m32;
    const bool col_ok = wave_on && t < T;

    bf16x8 qf[D / 16];
#pragma unroll
    for (int dk = 0; dk < D / 16; dk++) {
        if (col_ok) qf[dk] = ld_bf16x8(q + ((size_t)t * H + hq) * D + dk * 16 + hi * 8);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) qf[dk][j] = (__bf16)0.f;
        }
    }
    int lo = c0;
    if (window >= 0 && len + t - window > c0) lo = len + t - window;
    const int lo_q = col_ok ? lo : 0, hi_q = col_ok ? len + t + 1 : 0;
    const int pre_hi = col_ok ? c0 : 0;
    int wstart = 0;
    if (c0 == 0 && window >= 0 && len + t0 - window > 0) wstart = ((len + t0 - window) / 32) * 32;
    const int wend = (wave_on && t0 < T) ? len + min(T, t0 + 32) : 0;
    // a tile [kb, kb+32) needs no mask for this wave when every EXISTING row sees all of it.  Rows past the prompt's end
    // (a ragged last block) have q = 0 and no store: whatever they compute stays on their own lanes.  (Requiring 32
    // existing rows sent the longest block of a T % 32 != 0 prompt down the masked path for all of its ~T/32 tiles: Qwen2-7B
    // T = 5000 456 us per layer against 320 at T = 4992.)
    const bool rows_full = wave_on && t0 < T;
    int lo_max = c0;                                                        // largest lower bound among the wave's rows
    if (window >= 0 && len + t0 + 31 - window > c0) lo_max = len + t0 + 31 - window;
    const int hi_min = len + t0 + 1;                                        // smallest upper bound

    float mrow = -INFINITY, lrow = 0.f;
    float16v O[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; db++)
#pragma unroll
        for (int r = 0; r < 16; r++) O[db][r] = 0.f;

    int kstart = 0;
    if (c0 == 0 && window >= 0 && len + tb0 - window > 0) kstart = ((len + tb0 - window) / 32) * 32;
    const int kend = len + min(T, tb0 + 32 * TB);
    const int ntiles = (kend - kstart + 31) / 32;
    const int nsteps = (ntiles + KSF - 1) / KSF;                            // a step = KSF key tiles, one per wave of a pair

    auto stage = [&](int kbase, unsigned char *buf) {
        unsigned char *kt = buf, *vt = buf + 32 * D * 2;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            int e = wave + k * NW;
            if (e >= NI) e -= NI;                                           // PER * NW < 2 * NI
            if (e < NK) {
                const int p = e * 64 + lane, row = p / CPR, pc = p % CPR, c = pc ^ (row & (CPR - 1));
                glds16(kb + (size_t)(kbase + row) * D + c * 8, kt + e * 1024);
            } else {
                const int ev = e - NK, p = ev * 64 + lane, row = p >> 2, pc = p & 3, c = pc ^ ((row >> 2) & 3);
                glds16(vb + (size_t)row * seq_alloc + kbase + c * 8, vt + ev * 1024);
            }
        }
    };
    // (a step always stages KSF tiles so that the counted wait below is the same for every step: the tile behind the
    // last one is the last one again)
    auto stage_step = [&](int s_, unsigned char *slot) {
#pragma unroll
        for (int f = 0; f < KSF; f++) stage(kstart + 32 * min(s_ * KSF + f, ntiles - 1), slot + f * TILE);
    };
    stage_step(0, lds);
    if (NSLOT == 3 && nsteps > 1) stage_step(1, lds + SLOT);

    for (int sidx = 0; sidx < nsteps; sidx++) {
        // three slots: two steps in flight, the younger one's loads stay outstanding; two slots (KSF = 4): one step in flight
        if (NSLOT == 3 && sidx + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * KSF) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (sidx + NSLOT - 1 < nsteps) stage_step(sidx + NSLOT - 1, lds + ((sidx + NSLOT - 1) % NSLOT) * SLOT);
        const int jt = sidx * KSF + ksub;                                   // this wave's key tile of the step
        const int kbase = kstart + 32 * jt;
        if (jt >= ntiles || !(kbase + 32 > wstart && kbase < wend)) continue;   // wave-uniform
        const unsigned char *kt = lds + (sidx % NSLOT) * SLOT + ksub * TILE, *vt = kt + 32 * D * 2;

        // all K fragments are requested before the first MFMA (one register set per fragment: a shared one would
        // expose the LDS latency D/16 times per tile), the V fragments right behind them
        bf16x8 kf[D / 16];
#pragma unroll
        for (int dk = 0; dk < D / 16; dk++) {
            const int c = dk * 2 + hi;
            kf[dk] = *reinterpret_cast<const bf16x8 *>(kt + krow * (D * 2) + ((c ^ (krow & (CPR - 1))) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 vf[2][D / 32];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int db = 0; db < D / 32; db++) {
                const int row = db * 32 + m32, c = 2 * ks + hi;
                vf[ks][db] = *reinterpret_cast<const bf16x8 *>(vt + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
            }
        __builtin_amdgcn_sched_barrier(0);
        float16v S;
#pragma unroll
        for (int r = 0; r < 16; r++) S[r] = 0.f;
#pragma unroll
        for (int dk = 0; dk < D / 16; dk++) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[dk], qf[dk], S, 0, 0, 0);

        const bool full = rows_full && (kbase + 32 <= c0 || (kbase >= lo_max && kbase + 32 <= hi_min));
        float alpha;
        bf16x8 pa[2];
        if (full) {
            float mt = S[0];
#pragma unroll
            for (int r = 1; r < 16; r++) mt = fmaxf(mt, S[r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * scale_log2e;           // scale > 0: max commutes with it
            const float mn = fmaxf(mrow, mt);
            alpha = __builtin_amdgcn_exp2f(mrow - mn);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(S[r], scale_log2e, -mn));
                ps += pv;
                pa[r >> 3][r & 7] = (__bf16)pv;
            }
            lrow = lrow * alpha + ps;
            mrow = mn;
        } else {
            float x[16];
            float mt = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int key = kbase + (r & 7) + 8 * hi + 16 * (r >> 3);
                const bool ok = key < pre_hi || (key >= lo_q && key < hi_q);
                x[r] = ok ? S[r] * scale_log2e : -INFINITY;
                mt = fmaxf(mt, x[r]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float mn = fmaxf(mrow, mt);
            const bool dead = mn == -INFINITY;                              // this row has seen no visible key yet
            alpha = dead ? 1.0f : __builtin_amdgcn_exp2f(mrow - mn);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float pv = dead ? 0.f : __builtin_amdgcn_exp2f(x[r] - mn);
                ps += pv;
                pa[r >> 3][r & 7] = (__bf16)pv;
            }
            lrow = lrow * alpha + ps;
            mrow = mn;
        }
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {                   // some row's running max moved
#pragma unroll
            for (int db = 0; db < D / 32; db++)
#pragma unroll
                for (int r = 0; r < 16; r++) O[db][r] *= alpha;
        }
#pragma unroll
        for (int db = 0; db < D / 32; db++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) O[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks][db], pa[ks], O[db], 0, 0, 0);
    }

    if constexpr (KS2) {
        // the two waves of a (head, token block) meet: the odd one hands (m, l, O) over through the ring's memory
        __builtin_amdgcn_s_barrier();                                       // every wave is done with the key tiles
        constexpr int MXF = 64 * (D / 2 + 2);                               // floats of one partner's (m, l, O)
        float *mx0 = reinterpret_cast<float *>(lds) + (size_t)(tok_sub * Gs + wave % Gs) * (KSF - 1) * MXF;
        if (wave_on && ksub >= 1) {
            float *mx = mx0 + (size_t)(ksub - 1) * MXF;
            mx[lane] = mrow; mx[64 + lane] = lrow;
#pragma unroll
            for (int db = 0; db < D / 32; db++)
#pragma unroll
                for (int r = 0; r < 16; r++) mx[(2 + db * 16 + r) * 64 + lane] = O[db][r];
        }
        // s_barrier waits for no counter (gfx950 back-off barriers): the hand-over's ds_writes must have landed before the
        // partner wave may read them
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave_on && ksub == 0) {
#pragma nounroll
            for (int p = 0; p < KSF - 1; p++) {                             // partners in key-tile order: the sum does not depend on timing
                const float *mx = mx0 + (size_t)p * MXF;
                const float m1 = mx[lane], l1 = mx[64 + lane];
                const float mn = fmaxf(mrow, m1);
                const float a0 = mrow == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mrow - mn), a1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m1 - mn);
                lrow = lrow * a0 + l1 * a1;
                mrow = mn;
#pragma unroll
                for (int db = 0; db < D / 32; db++)
#pragma unroll
                    for (int r = 0; r < 16; r++) O[db][r] = O[db][r] * a0 + mx[(2 + db * 16 + r) * 64 + lane] * a1;
            }
        }
    }
    const float ltot = lrow + __shfl_xor(lrow, 32, 64);
    if (col_ok && ksub == 0) {
        const float inv = 1.0f / ltot;
        bf16_t *o = out + ((size_t)t * H + hq) * D;
#pragma unroll
        for (int db = 0; db < D / 32; db++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int d0 = db * 32 + 8 * rr + 4 * hi;
                ushort4v pk;
#pragma unroll
                for (int j = 0; j < 4; j++) pk[j] = float_to_bf16_bits(O[db][4 * rr + j] * inv);
                *reinterpret_cast<ushort4v *>(o + d0) = pk;
            }
    }
    __builtin_amdgcn_s_barrier();        // every wave is done with the ring before the next pass refills it
    }
}

template <int D, int NW, int KSF = 1>
static int launch_pf32(Launcher &L, const void *q, const void *k_cache, const void *v_cache_T, const StepState *st, void *out,
                       int64_t T, int64_t H, int64_t Hkv, int64_t seq_alloc, float scale, int64_t window, int TB, int paired, int gsub = 0) {
    const int64_t nb = (T + 32 * TB - 1) / (32 * TB);
    const int64_t G = H / Hkv, nsg = gsub > 0 ? (G + gsub - 1) / gsub : 1;
    dim3 grid((unsigned)(paired ? (nb + 1) / 2 : nb), (unsigned)(Hkv * nsg));
    if (paired == 2) {                                              // one persistent workgroup per CU of the CURRENT device
        static std::atomic<int> cached[64];
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            cus = cached[dev].load();
            if (!cus) {
                hipDeviceProp_t prop;
                cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
                cached[dev].store(cus);
            }
        }
        grid = dim3((unsigned)cus, 1);
    }
    const size_t lds = (KSF == 4 ? 2 : 3) * (size_t)(2 * 32 * D * 2) * KSF;
    const double flops = 2.0 * (double)T * T * H * D;
    Launcher LL = L; LL.tag = KSF == 4 ? "32row,ks4" : KSF == 2 ? "32row,ks2" : "32row";
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(attn_prefill32_kernel<D, NW, KSF>), lds));
    return LL.launch(KC_ATTN_PREFILL, 0, flops, attn_prefill32_kernel<D, NW, KSF>, grid, dim3(NW * 64), lds, (const bf16_t *)q,
                     (const bf16_t *)k_cache, (const bf16_t *)v_cache_T, st, (bf16_t *)out, (int)T, (int)H, (int)Hkv, (int)seq_alloc,
                     scale * 1.44269504088896340736f, (int)window, paired, gsub);
}

static std::atomic<int> g_prefill_force{0};
void attn_prefill_force(int which) { g_prefill_force = which; }

int launch_attn_prefill_mfma(Launcher &L, const void *q, const void *k_cache, const void *v_cache_T, const StepState *st,
                             void *out, int64_t T, int64_t H, int64_t Hkv, int64_t d, int64_t seq_alloc, float scale,
                             int64_t window) {
    const int G = (int)(H / Hkv);
    if (G > 8) FL_FAIL(FL_ERR_UNSUPPORTED, "mfma prefill attention: at most 8 query heads per kv head");
    // 32-row waves (attn_prefill32_kernel) once the prompt is long enough to fill the chip with their workgroups
    // (with wave pairs from 640 tokens for G = 1, 2, 4: Mistral-7B T = 768 31.9 -> 24.8 us per layer, T = 512 a tie, below slower;
    // from 256 tokens for the groups dealt as subgroups of four heads: Qwen2-7B T = 512 23.2 -> 18.3, TinyLlama 16.5 -> 13.2, T = 256 a tie)
    const int pf32_min_env = tune(TK_ATTN_PF32_MIN_T);
    // (since round 4, with four waves per (head, block) up to 576 tokens: from 352 tokens for G = 1, 2, 4 as well)
    const int pf32_min_t = pf32_min_env > 0 ? pf32_min_env : ((G == 1 || G == 2 || G == 4) ? 352 : (G > 4 ? 256 : 1024));
    const int force = g_prefill_force.load();                        // fl_op_attention pins one kernel (unit tests)
    // (577-639 tokens, G = 1, 2, 4: the four-wave form ends at 576 tokens, the wave pairs pay from 640: the 16-row kernel in between)
    const bool gap = pf32_min_env <= 0 && (G == 1 || G == 2 || G == 4) && T > 576 && T < 640;
    if ((force == 3 || (force == 0 && T >= pf32_min_t && !gap)) && scale > 0.f) {
        // waves per workgroup: 8 (G = 1, 2, 4), 6 (G = 3), else G.  Paired (balanced) grids win as soon as they cover
        // ~3/4 of the chip -- Mistral-7B per layer: T = 3072 134 us paired (192 workgroups) vs 197 unpaired, T = 4096
        // 177 vs 305, T = 8192 681 vs 693; T = 2048 (128 paired workgroups) 98 vs 85.  4-wave workgroups (half the K/V
        // reuse, one wave per SIMD) lost everywhere: T = 2048 105 / 180 us, T = 4096 371 / 515.
        const int NW = G <= 4 ? (8 / G) * G : G;
        int TB = NW / G;
        // too few (block, kv head) items to balance: halve the token blocks and split every block's keys over a wave pair
        const int ks2_mode = tune(TK_ATTN_PF32_KS2);
        bool ks2 = NW == 8 && TB % 2 == 0 && (ks2_mode >= 0 ? ks2_mode != 0 : ((T + 32 * TB - 1) / (32 * TB)) * Hkv < 2 * 256);
        // groups of 5..8 heads: wave pairs only fit when the group is dealt in subgroups of four heads (8 waves = 4 heads x a pair)
        int gsub = 0;
        if (!ks2 && G > 4 && (ks2_mode >= 0 ? ks2_mode != 0 : ((T + 31) / 32) * Hkv < 2 * 256)) { ks2 = true; gsub = 4; TB = 2; }
        if (ks2) TB /= 2;
        // ... and when even the wave pairs leave fewer than two rounds of items: FOUR waves per (head, block), groups of four or more
        // heads dealt two heads at a time -- twice the items again, the longest block's chain a quarter (attn_pf32_ks2 = 4 forces it)
        int ksf = ks2 ? 2 : 1;
        if (ks2 && G != 3) {
            const int64_t items2 = ((T + 32 * TB - 1) / (32 * TB)) * Hkv * (gsub ? (G + gsub - 1) / gsub : 1);
            // Whole prefills, wave pairs / four waves: Mistral-7B 384 / 512 tokens x 0.994 / 0.988 (against the 16-row kernel it replaces
            // there), 600-1536 a tie, 2048 x 1.019; Qwen2-7B 257-512 x 0.986-0.993, 640-1024 a tie; TinyLlama 300 / 512 x 0.986 / 0.972,
            // 640 a tie: on up to 576 tokens.  (Per launch at 512 tokens, Mistral-7B: 16-row kernel 18.6 us, wave pairs 19.2, four waves 15.6.)
            if (ks2_mode == 4 || (ks2_mode < 0 && items2 < 2 * 256 && T <= 576)) {
                ksf = 4;
                gsub = G >= 4 ? 2 : 0;
                TB = 8 / (gsub ? gsub : G) / 4;
            }
        }
        const int64_t nb = (T + 32 * TB - 1) / (32 * TB);
        const int64_t nhs = Hkv * (gsub ? (G + gsub - 1) / gsub : 1);
        const int force_pair = tune(TK_ATTN_PF32_PAIRED);
        // two rounds or more of (block, kv head) items: persistent workgroups, snake order (launch_pf32)
        const int paired = force_pair >= 0 ? force_pair : (nb * nhs >= 2 * 256 ? 2 : ((nb + 1) / 2 * nhs >= 180 ? 1 : 0));
        if (ksf == 4 && d == 128) return launch_pf32<128, 8, 4>(L, q, k_cache, v_cache_T, st, out, T, H, Hkv, seq_alloc, scale, window, TB, paired, gsub);
        if (ksf == 4 && d == 64) return launch_pf32<64, 8, 4>(L, q, k_cache, v_cache_T, st, out, T, H, Hkv, seq_alloc, scale, window, TB, paired, gsub);
        if (ks2 && d == 128) return launch_pf32<128, 8, 2>(L, q, k_cache, v_cache_T, st, out, T, H, Hkv, seq_alloc, scale, window, TB, paired, gsub);
        if (ks2 && d == 64) return launch_pf32<64, 8, 2>(L, q, k_cache, v_cache_T, st, out, T, H, Hkv, seq_alloc, scale, window, TB, paired, gsub);
#define FL_PF32(DD, WW) if (d == DD && NW == WW) return launch_pf32<DD, WW>(L, q, k_cache, v_cache_T, st, out, T, H, Hkv, seq_alloc, scale, window, TB, paired);
        FL_PF32(128, 8) FL_PF32(128, 7) FL_PF32(128, 6) FL_PF32(128, 5) FL_PF32(64, 8) FL_PF32(64, 7) FL_PF32(64, 6) FL_PF32(64, 5)
#undef FL_PF32
    }
    // token sub-tiles per workgroup: every staged K/V tile then serves TT x G x 16 query rows.  Measured: G = 4
    // (Mistral, T = 2048) 4.75 ms with 4 waves, 4.25 with 8, 4.06 with 16; G = 7 (Qwen2, T = 4096) 10.8 ms with 7
    // waves, 12.3 with 14 (126 VGPRs: 16 waves per CU either way, and a longer token block is more lopsided under
    // the causal mask)
    // ... but only while the grid still covers the chip: at T = 512 (Mistral) 4-wave workgroups (256 of them) take
    // 26 us per layer against 36 us for 64 16-wave ones; at T = 1024 8 waves 49 us against 57 (4) and 67 (16)
    const int tt_waves = tune(TK_ATTN_PF_WAVES);
    int TT = tt_waves > 0 ? std::max(1, std::min(16, tt_waves) / G) : (G <= 4 ? 16 / G : 1);
    if (tt_waves <= 0)
        while (TT > 1 && ((T + 16 * TT - 1) / (16 * TT)) * Hkv < 256) TT >>= 1;
    dim3 grid((unsigned)((T + 16 * TT - 1) / (16 * TT)), (unsigned)Hkv);
    dim3 block((unsigned)(G * TT * 64));
    // LDS tiles in the ring (FL_ATTN_PF_STAGES, 2..4).  Measured: a deeper ring buys nothing -- Mistral T = 512 25 / 26 /
    // 25 us per layer with 2 / 3 / 4 tiles, T = 768 41 / 40 / 42: with one wave per SIMD a key step is a chain of
    // dependent LDS read -> MFMA -> shuffle -> exp -> MFMA latencies (~1.5 us), not a wait for the tile
    const int pf_stages = tune(TK_ATTN_PF_STAGES);
    int nst = std::max(2, std::min(4, G * TT <= 8 ? pf_stages : 2));
    // key split (two waves per head and token sub-tile) while the workgroup stays within 8 waves: short prompts
    const int pf_ksplit = tune(TK_ATTN_PF_KSPLIT);
    // (four waves per query tile measured slower than two: Mistral T = 512 23 vs 18 us per layer, T = 768 45 vs 33)
    // ... while the workgroup stays within 16 waves and its merge slabs within the 64 KB of LDS a launch gets by default
    const size_t slabs = (size_t)G * TT * (16 * d + 32) * 4;
    const int ks = (pf_ksplit >= 2 && G * TT * 2 <= 16 && slabs <= 64 * 1024) ? 2 : 1;
    if (ks == 2) { nst = 2; block = dim3((unsigned)(G * TT * 2 * 64)); }
    const size_t lds = std::max((size_t)nst * ks * (size_t)(2 * 32 * d * 2), ks == 2 ? slabs : (size_t)0);
    double flops = 2.0 * (double)T * T * H * d;
    if (d == 128)
        return L.launch(KC_ATTN_PREFILL, 0, flops, attn_prefill_mfma_kernel<128>, grid, block, lds, (const bf16_t *)q,
                        (const bf16_t *)k_cache, (const bf16_t *)v_cache_T, st, (bf16_t *)out, (int)T, (int)H, (int)Hkv,
                        (int)seq_alloc, scale, (int)window, TT, nst, ks);
    if (d == 64)
        return L.launch(KC_ATTN_PREFILL, 0, flops, attn_prefill_mfma_kernel<64>, grid, block, lds, (const bf16_t *)q,
                        (const bf16_t *)k_cache, (const bf16_t *)v_cache_T, st, (bf16_t *)out, (int)T, (int)H, (int)Hkv,
                        (int)seq_alloc, scale, (int)window, TT, nst, ks);
    FL_FAIL(FL_ERR_UNSUPPORTED, "mfma attention: head_dim %lld", (long long)d);
}

}  // namespace fl
