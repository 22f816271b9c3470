// k_attn_oproj.hip -- decode attention and o_proj in ONE launch (bf16), with every dependency kept inside a kv head's group.
//
// Decode attention is a latency chain that touches a few MB of K/V and leaves HBM idle for its whole duration, while
// the o_proj GEMV that follows needs attention outputs before its first FMA but its WEIGHTS depend on nothing.  An
// all-to-all hand-off inside a launch (every workgroup waits for every head) costs 5.9-8.6 us on this chip -- more
// than the 3.1 us kernel boundary it would replace (tools/micro/grid_handoff.hip) -- but a hand-off among the 32
// workgroups of ONE XCD costs 1.9 us.  So o_proj is cut along K by kv head:
//   * workgroup b belongs to kv head hk = b % Hkv (the workgroups the dispatcher places on one XCD when Hkv = 8) and
//     owns a block of output rows x the G*d columns of W_o that multiply THAT kv head's query heads; it pulls this
//     slice (<= ~150 KB) into LDS with LDS-DMA the moment it starts (no VGPRs, nothing waits on it);
//   * the first `nsplit` workgroups of each group run the attention step for their key range (same MFMA code as
//     k_attn_mfma.hip), publish their (m, l, o) partials with write-through stores and bump the group's word -- they
//     never wait for anybody;
//   * every workgroup with rows polls ITS group's word (one lane, relaxed loads, bounded), reads the group's partials
//     with coherent loads (no fences anywhere), combines the splits, and finishes its rows from LDS on the matrix cores;
//   * the result is Hkv partial vectors [Hkv][h] (fp32), summed in fixed order by the consumer's norm prologue
//     (GemvArgs::delta_nslab).
// All workgroups are co-resident by construction (grid <= CU count, one workgroup per CU by LDS footprint), the poll
// is bounded (on give-up an error word is set in the step state and the host reports FL_ERR_HIP).  The group words
// (Hkv per layer) are zeroed by set_state, and the target is (step+1)*nsplit, so a captured graph replays without
// per-step memset nodes.
#include <algorithm>
#include <atomic>
#include <vector>

#include "attn_mfma.h"

namespace fl {
#ifdef FL_EXPERIMENTAL


struct AttnOprojArgs {
    const bf16_t *q, *kc, *vT; const StepState *st;
    float *part;                   // [H][nsplit][D + 4]: a split's unnormalised o[D], m, l (16-byte aligned rows)
    unsigned *done;                // this layer's words: [Hkv] split partials published since set_state
    const bf16_t *Wo;              // [h][K], K = H*D
    float *slabs;                  // [Hkv][h] fp32: partial o_proj outputs, one per kv head
    int H, Hkv, seq_alloc, nsplit, h, K;
    int rows_attn, rows_other;     // row split inside a group: attention workgroups own rows_attn rows each
    float scale;
    StepState *st_rw;              // for the error word
    unsigned long long *stamps;    // FL_AO_STAMPS diagnostic: [workgroup][12] s_memrealtime stamps, or null
    int attn_waves;                // waves of an attention workgroup that take key tiles (a CU pulls only ~40 GB/s: a split's K/V must stay small)
    int dma_delay;                 // x ~0.25 us before the non-attention workgroups start their W_o stream
};

__device__ inline void glds16_w(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
// coherent (sc0 sc1) load: past the L1 and this XCD's L2, which may hold the previous step's line
__device__ inline float ld_coherent(const float *p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

template <int D, int GMAX>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_oproj_kernel(const AttnOprojArgs a) {
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // [W rows | x | attention slabs]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g4 = lane >> 4;
    const int hk = blockIdx.x % a.Hkv, mb = blockIdx.x / a.Hkv;            // kv head (group), member
    const int G = a.H / a.Hkv, seg = G * D, cpr = seg >> 3;                 // W_o columns of this group; 16-byte chunks per row
    const bool attn_block = mb < a.nsplit;
    const int nrows_want = attn_block ? a.rows_attn : a.rows_other;
    const int row0 = attn_block ? mb * a.rows_attn : a.nsplit * a.rows_attn + (mb - a.nsplit) * a.rows_other;
    const int nrows = max(0, min(nrows_want, a.h - row0));
    const size_t wbytes = ((size_t)nrows_want * seg * 2 + 1023) & ~(size_t)1023;
    unsigned char *wl = lds_raw;
    bf16_t *xs = reinterpret_cast<bf16_t *>(lds_raw + wbytes);
    float *slab = reinterpret_cast<float *>(lds_raw + wbytes + (((size_t)seg * 2 + 15) & ~(size_t)15));
    auto stamp = [&](int k) { if (a.stamps && tid == 0) a.stamps[blockIdx.x * 12 + k] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);

    // W_o slice -> LDS: 64 chunks of 16 bytes per wave instruction, linear image [row][seg]
    auto pull_rows = [&]() {
        const int total = nrows * cpr, ninstr = (total + 63) >> 6;
        const unsigned char *src = reinterpret_cast<const unsigned char *>(a.Wo) + ((size_t)row0 * a.K + (size_t)hk * seg) * 2;
        for (int e = wave; e < ninstr; e += NW) {
            const int c = min(e * 64 + lane, total - 1);                 // (the lanes past the end land in the rounded-up tail)
            const int r = c / cpr, p = c - r * cpr;
            int cc = p - r % cpr;                                         // slot p of row r holds chunk (p - r) mod cpr: the 16 rows of an
            if (cc < 0) cc += cpr;                                        // MFMA fragment read then fall into 16 different bank groups
            glds16_w(src + (size_t)r * a.K * 2 + (size_t)cc * 16, wl + (size_t)e * 1024);
        }
    };

    if (attn_block) {
        const int S = (int)a.st->len + 1;
        int per = (S + a.nsplit - 1) / a.nsplit;
        const int AW = a.attn_waves;
        per = (per + 32 * AW - 1) / (32 * AW) * (32 * AW);
        const int lo = mb * per, hi = min(S, lo + per);
        if (a.stamps && tid == 0 && hi > -5) a.stamps[blockIdx.x * 12 + 7] = __builtin_amdgcn_s_memrealtime();
        const bf16_t *kb = a.kc + (size_t)hk * a.seq_alloc * D;
        const bf16_t *vb = a.vT + (size_t)hk * D * a.seq_alloc;
        // everything this wave needs for its first 32-key step is requested at once (q for the group's heads, K and V^T
        // fragments): two dependent round trips in all.  (Loads left inside attn_tile were serialised by hipcc -- one
        // wait per fragment, a dozen round trips.)
        bf16x8 qf[D / 32];
#pragma unroll
        for (int dk = 0; dk < D / 32; dk++) qf[dk] = ld_bf16x8(a.q + (size_t)(hk * G + min(i, G - 1)) * D + dk * 32 + g4 * 8);
        RegKV<D> kv0;
        const bool tile0 = wave < AW && lo + 32 * wave < hi;
        if (tile0) kv0.load(kb, vb, a.seq_alloc, lo + 32 * wave, i, g4);
        if (i >= G) {
#pragma unroll
            for (int dk = 0; dk < D / 32; dk++)
#pragma unroll
                for (int j = 0; j < 8; j++) qf[dk][j] = (__bf16)0.f;
        }
        MfmaAttnState<D> s; s.init();
        if (tile0) attn_tile<D>(s, qf, kv0, lo + 32 * wave, 0, lo, hi, a.scale, lane);
        for (int kbase = lo + 32 * wave + 32 * AW; wave < AW && kbase < hi; kbase += 32 * AW) {
            const GlobalKV<D> kv{kb, vb, a.seq_alloc, kbase, i, g4};
            attn_tile<D>(s, qf, kv, kbase, 0, lo, hi, a.scale, lane);
        }
        if (a.stamps && tid == 0 && s.m > -1e38f) a.stamps[blockIdx.x * 12 + 8] = __builtin_amdgcn_s_memrealtime();
        mfma_state_to_lds<D, GMAX>(s, slab, wave, G, lane);
        __syncthreads();
        stamp(1);
        // publish this split's partials (write-through: visible to every XCD once vmcnt has drained), then count it
        for (int e = tid; e < G * (D / 4); e += NW * 64) {
            const int g = e / (D / 4), j4 = (e % (D / 4)) * 4;
            float M, L, O[4];
            combine_lds<D, GMAX, NW>(slab, g, j4, M, L, O);
            float *pp = a.part + ((size_t)(hk * G + g) * a.nsplit + mb) * (D + 4);
            st_sc1_x4(pp + j4, float4v{O[0], O[1], O[2], O[3]});
            if (j4 == 0) st_sc1_pair(pp + D, M, L);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.done + hk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(2);
        if (nrows <= 0) return;                                             // workgroup-uniform
    }
    if (nrows <= 0) return;                                                 // (more workgroups than row blocks)
    // the weight stream of the other workgroups waits a moment: 33 MB of requests ahead of the attention workgroups' first
    // loads would put their two dependent round trips behind a saturated memory system
    if (!attn_block) for (int z = 0; z < a.dma_delay; z++) __builtin_amdgcn_s_sleep(8);
    pull_rows();       // (attention workgroups: behind their own loads, which an in-order vmcnt would otherwise wait for)

    // ---- wait for the splits of THIS kv head (one lane polls the group's word, relaxed, bounded; its loads return behind
    // the wave's share of the stream, i.e. about when the stream has landed -- two waves kept free of weight loads to
    // poll earlier saw the word LATER: their polls queue behind the CU's own burst all the same, and six loaders are slower)
    if (tid == 0) {
        const unsigned target = (a.st->step + 1u) * (unsigned)a.nsplit;
        unsigned spins = 0;
        while (__hip_atomic_load(a.done + hk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { a.st_rw->error = 0xA77E; break; }       // bounded: never hang the GPU
        }
    }
    __syncthreads();
    stamp(3);
    // read the partials with coherent loads (no fences anywhere) and combine them into x (bf16, as the two-launch path
    // rounds it): one (head, 4 d-elements) slot per thread, at most 8 splits, one round trip
    for (int e = tid; e < G * (D / 4); e += NW * 64) {
        const int g = e / (D / 4), j4 = (e % (D / 4)) * 4;
        const float *pp = a.part + (size_t)(hk * G + g) * a.nsplit * (D + 4);
        float pm[8], pl[8], po[8][4];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float *q = pp + (size_t)min(u, a.nsplit - 1) * (D + 4);
            pm[u] = ld_coherent(q + D); pl[u] = ld_coherent(q + D + 1);
#pragma unroll
            for (int j = 0; j < 4; j++) po[u][j] = ld_coherent(q + j4 + j);
        }
        float M = -INFINITY;
#pragma unroll
        for (int u = 0; u < 8; u++) if (u < a.nsplit) M = fmaxf(M, pm[u]);
        float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float w = (u < a.nsplit && pm[u] != -INFINITY) ? __expf(pm[u] - M) : 0.f;
            L = fmaf(pl[u], w, L);
#pragma unroll
            for (int j = 0; j < 4; j++) O[j] = fmaf(po[u][j], w, O[j]);
        }
        const float inv = 1.0f / L;
#pragma unroll
        for (int j = 0; j < 4; j++) elem<bf16_t>::st(xs + g * D + j4 + j, O[j] * inv);
    }
    stamp(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the W_o slice has landed
    __syncthreads();
    stamp(5);

    // ---- rows of this workgroup from LDS on the matrix cores: a 16-row tile per wave step, x broadcast to all 16
    // columns (column 0 is kept) -- as VALU dot products the bf16 unpacking made this the longest section of the launch
    // (3 us; 16-way bank conflicts of an unswizzled image: 2.8)
    constexpr int KSM = GMAX * D / 32;                                 // K steps of a full group (G == GMAX: Mistral, TinyLlama)
    auto store_tile = [&](int t, const float4v &v) {
        if (i != 0) return;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = t * 16 + 4 * g4 + q;
            if (row < nrows) a.slabs[(size_t)hk * a.h + row0 + row] = v[q];
        }
    };
    if (seg == KSM * 32) {
        // x stays in registers for all row tiles (LDS moves 128 bytes per clock: reading x again per tile doubled the section),
        // a tile's W fragments are requested in one go
        bf16x8 xf[KSM];
#pragma unroll
        for (int ks = 0; ks < KSM; ks++) xf[ks] = ld_bf16x8(xs + ks * 32 + g4 * 8);
        for (int t = wave; t * 16 < nrows; t += NW) {
            const int r = min(t * 16 + i, nrows - 1);                  // (rows past the end repeat the last one; not stored)
            const bf16_t *wr = reinterpret_cast<const bf16_t *>(wl) + (size_t)r * seg;
            const int rot = r % cpr;
            bf16x8 wf[KSM];
#pragma unroll
            for (int ks = 0; ks < KSM; ks++) {
                int slot = ks * 4 + g4 + rot;                           // chunk (ks * 4 + g4) of row r sits in slot (chunk + r) mod cpr
                if (slot >= cpr) slot -= cpr;
                wf[ks] = ld_bf16x8(wr + slot * 8);
            }
            float4v acc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) acc[u] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSM; ks++) acc[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf[ks], acc[ks & 3], 0, 0, 0);
            store_tile(t, (acc[0] + acc[2]) + (acc[1] + acc[3]));
        }
    } else {
        for (int t = wave; t * 16 < nrows; t += NW) {
            const int r = min(t * 16 + i, nrows - 1);
            const bf16_t *wr = reinterpret_cast<const bf16_t *>(wl) + (size_t)r * seg;
            const bf16_t *xr = xs + g4 * 8;
            int slot = g4 + r % cpr;
            if (slot >= cpr) slot -= cpr;
            auto next4 = [&]() { slot += 4; if (slot >= cpr) slot -= cpr; };
            float4v acc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) acc[u] = float4v{0.f, 0.f, 0.f, 0.f};
            int k0 = 0;
            for (; k0 + 128 <= seg; k0 += 128) {                       // four K steps per trip to LDS, four independent accumulators
                bf16x8 wf[4], xf[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { wf[u] = ld_bf16x8(wr + slot * 8); next4(); xf[u] = ld_bf16x8(xr + k0 + 32 * u); }
#pragma unroll
                for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[u], acc[u], 0, 0, 0);
            }
            for (; k0 < seg; k0 += 32) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_bf16x8(wr + slot * 8), ld_bf16x8(xr + k0), acc[0], 0, 0, 0);
                next4();
            }
            store_tile(t, (acc[0] + acc[2]) + (acc[1] + acc[3]));
        }
    }
    stamp(6);
}

// LDS budget and row split; returns false when the fused launch does not apply
bool attn_oproj_plan(int64_t H, int64_t Hkv, int64_t d, int64_t h, int nsplit, int cus, int *n_blocks, int *rows_attn,
                     int *rows_other, size_t *lds_bytes) {
    if (Hkv <= 0 || H % Hkv) return false;
    const int64_t G = H / Hkv, seg_b = G * d * 2;
    if (!(d == 64 || d == 128) || G > 8 || Hkv > 8) return false;                        // (Hkv partial vectors are summed by the consumer)
    const int mg = (int)(cus / Hkv);                                                       // workgroups per kv head
    if (nsplit < 1 || nsplit > 8 || nsplit >= mg) return false;                           // (one pass of at most 8 splits per combining thread)
    const int gmax = G <= 4 ? 4 : 8;
    const size_t xb = ((size_t)seg_b + 15) & ~(size_t)15, slab = (size_t)8 * gmax * (d + 2) * 4 + 256;
    const size_t budget = 160 * 1024 - 2048;
    if (xb + slab + 1024 >= budget) return false;
    const int64_t max_other = (int64_t)((budget - xb - 1024) / (size_t)seg_b);            // no slabs there
    const int64_t max_attn = (int64_t)((budget - xb - slab - 1024) / (size_t)seg_b);
    const int n_other = mg - nsplit;
    // attention workgroups should own as few rows as possible: the others take up to their LDS limit
    const int ro = (int)std::min<int64_t>(max_other, (h + n_other - 1) / n_other);
    const int64_t left = h - (int64_t)ro * n_other;
    const int ra = left > 0 ? (int)((left + nsplit - 1) / nsplit) : 0;
    if (ro < 1 || ra > max_attn) return false;
    auto wb = [&](int rows) { return ((size_t)rows * seg_b + 1023) & ~(size_t)1023; };
    *n_blocks = mg * (int)Hkv; *rows_attn = ra; *rows_other = ro;
    *lds_bytes = std::max(wb(ra) + xb + slab, wb(ro) + xb);
    return true;
}

int launch_attn_oproj(Launcher &L, const void *q, const void *k_cache, const void *v_cache_T, const StepState *st,
                      StepState *st_rw, float *partials, unsigned *done, int nsplit, int attn_waves, int64_t kv_len_hint,
                      const void *Wo, float *slabs, int64_t H, int64_t Hkv, int64_t d, int64_t h, int64_t seq_alloc, float scale) {
    static std::atomic<int> cus_cached{0};                // (all shards run on the same kind of GPU)
    int cus = cus_cached.load();
    if (!cus) {
        int dev = 0; hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
        cus_cached.store(cus);
    }
    int nb = 0, ra = 0, ro = 0; size_t lds = 0;
    if (!attn_oproj_plan(H, Hkv, d, h, nsplit, cus, &nb, &ra, &ro, &lds))
        FL_FAIL(FL_ERR_UNSUPPORTED, "fused attention+o_proj launch does not fit this shape");
    AttnOprojArgs a;
    a.q = (const bf16_t *)q; a.kc = (const bf16_t *)k_cache; a.vT = (const bf16_t *)v_cache_T; a.st = st; a.st_rw = st_rw;
    a.part = partials; a.done = done; a.Wo = (const bf16_t *)Wo; a.slabs = slabs;
    a.H = (int)H; a.Hkv = (int)Hkv; a.seq_alloc = (int)seq_alloc; a.nsplit = nsplit; a.h = (int)h; a.K = (int)(H * d);
    a.rows_attn = ra; a.rows_other = ro; a.scale = scale;
    const int G = (int)(H / Hkv);
    const double bytes = (double)h * H * d * 2 + 2.0 * (double)kv_len_hint * Hkv * d * 2;
    const double flops = 2.0 * h * H * d + 4.0 * (double)kv_len_hint * H * d;
    a.attn_waves = std::max(1, std::min(8, attn_waves));
    const int delay = std::max(0, tune(TK_AO_DELAY));    // x ~0.25 us (measured: 0 -> 13.9 us, 6 -> 12.0, 12 -> 12.5)
    a.dma_delay = delay;
    // FL_AO_STAMPS=1 (eager launches only: run with FL_GRAPH=0): every workgroup records s_memrealtime at its section
    // boundaries; launches 201-203 of the process are read back and summarised on stderr (tools/ao_stamps.sh)
    const bool want_stamps = env_str("FL_AO_STAMPS") != nullptr;
    static unsigned long long *stamps = nullptr;
    static std::atomic<int> launches{0};
    a.stamps = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool stamped = want_stamps && hipStreamIsCapturing(L.stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone &&
                         (stamps || hipMalloc(&stamps, (size_t)cus * 12 * 8) == hipSuccess) &&
                         hipMemsetAsync(stamps, 0, (size_t)cus * 12 * 8, L.stream) == hipSuccess;
    if (stamped) a.stamps = stamps;
    auto go = [&](auto kern) -> int {
        FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
        FL_TRY(L.launch(KC_ATTN_OPROJ, bytes, flops, kern, dim3((unsigned)nb), dim3(512), lds, a));
        if (!stamped) return FL_OK;
        const int n = ++launches;
        if (n <= 200 || n > 203) return FL_OK;
        std::vector<unsigned long long> hst((size_t)nb * 12);
        FL_HIP(hipStreamSynchronize(L.stream));
        FL_HIP(hipMemcpy(hst.data(), stamps, hst.size() * 8, hipMemcpyDeviceToHost));
        static const char *names[12] = {"start", "attention: slabs in LDS", "attention: published", "poll passed", "splits combined",
                                        "weights landed", "rows done", "attention: S known", "attention: wave 0 through its keys", "", "", ""};
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nb; b++) t0 = std::min(t0, hst[(size_t)b * 12]);
        fprintf(stderr, "attn_oproj (%d workgroups, %d + %d rows, %zu B LDS, %d splits), us since the first workgroup started, mean / max (n):\n", nb, ra, ro, lds, nsplit);
        for (int k = 0; k < 9; k++) {
            double mx = 0, sm = 0; int cn = 0;
            for (int b = 0; b < nb; b++)
                if (hst[(size_t)b * 12 + k]) { const double v = (double)(hst[(size_t)b * 12 + k] - t0) / 100.0; mx = std::max(mx, v); sm += v; cn++; }
            fprintf(stderr, "  %-36s %6.2f / %6.2f (%d)\n", names[k], cn ? sm / cn : 0., mx, cn);
        }
        return FL_OK;
    };
    if (d == 128) return G <= 4 ? go(attn_oproj_kernel<128, 4>) : go(attn_oproj_kernel<128, 8>);
    return G <= 4 ? go(attn_oproj_kernel<64, 4>) : go(attn_oproj_kernel<64, 8>);
}

#else
// Default build: 0-1.5 % on Mistral-7B, slower elsewhere (profiles/r02/README.md): not compiled in; `make EXPERIMENTAL=1` builds it.
bool attn_oproj_plan(int64_t, int64_t, int64_t, int64_t, int, int, int *, int *, int *, size_t *) { return false; }
int launch_attn_oproj(Launcher &, const void *, const void *, const void *, const StepState *, StepState *, float *, unsigned *, int, int, int64_t,
                      const void *, float *, int64_t, int64_t, int64_t, int64_t, int64_t, float) {
    FL_FAIL(FL_ERR_UNSUPPORTED, "the fused attention + o_proj launch is not in this build (make EXPERIMENTAL=1)");
}
#endif
}  // namespace fl
