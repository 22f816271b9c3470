// k_attn_oproj.hip -- decode attention and o_proj in ONE launch (bf16).
//
// Decode attention is a latency chain that touches ~2-8 MB of K/V and leaves HBM idle for its whole
// duration, while the o_proj GEMV that follows needs every attention output before its first FMA but
// its WEIGHTS depend on nothing.  So this launch runs one workgroup per CU:
//   * every workgroup immediately pulls its slice of W_o (rows x K bf16, up to ~150 KB) into LDS with
//     LDS-DMA (global_load_lds, no VGPRs, nothing waits on it);
//   * the first Hkv*nsplit workgroups do the attention step (same MFMA code as k_attn_mfma.hip; their
//     first 32-key tile is loaded into registers BEFORE their own weight DMA is issued, and they own
//     few or no W_o rows, so the DMA does not delay them), publish the head outputs and bump a
//     "heads done" word with an agent-scope release;
//   * every workgroup then polls that word (one lane, relaxed loads, bounded), acquires, stages the
//     attention output vector in LDS and finishes its rows from LDS in about a microsecond.
// The W_o stream (32 MB for Mistral-7B) is thereby hidden behind the attention latency: the pair costs
// about what attention alone did.  All workgroups are co-resident by construction (grid <= CU count, one
// workgroup per CU by LDS footprint), attention workgroups never wait on the others, and the poll is
// bounded (on give-up an error word is set in the step state and the host reports FL_ERR_HIP).
// The heads-done words (one per layer) are zeroed by set_state, and the target is (step+1)*Hkv, so a
// captured graph replays without per-step memset nodes.
#include <algorithm>
#include <atomic>

#include "attn_mfma.h"

namespace fl {

struct AttnOprojArgs {
    const bf16_t *q, *kc, *vT; const StepState *st;
    float *part_m, *part_l, *part_o; unsigned *tickets;
    bf16_t *ao;                    // [H*D] attention output (global; read back by every workgroup)
    unsigned *heads_done;          // this layer's word
    const bf16_t *Wo;              // [h][K], K = H*D
    float *delta;                  // [h] fp32
    int H, Hkv, seq_alloc, nsplit, h, K;
    int n_attn, rows_attn, rows_other;     // row split: attention workgroups own rows_attn rows each
    float scale;
    StepState *st_rw;              // for the error word
};

__device__ inline void glds16_w(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int D, int GMAX>
__global__ __launch_bounds__(512) void attn_oproj_kernel(const AttnOprojArgs a) {
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // [W rows | x | attention slabs]
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g4 = lane >> 4;
    const int K = a.K, bid = blockIdx.x;
    const bool attn_block = bid < a.n_attn;
    const int nrows_want = attn_block ? a.rows_attn : a.rows_other;
    const int row0 = attn_block ? bid * a.rows_attn : a.n_attn * a.rows_attn + (bid - a.n_attn) * a.rows_other;
    const int nrows = max(0, min(nrows_want, a.h - row0));
    // per-workgroup LDS layout: [own W rows][x][attention slabs (attention workgroups only)]
    unsigned char *wl = lds_raw;
    bf16_t *xs = reinterpret_cast<bf16_t *>(lds_raw + (size_t)nrows_want * K * 2);
    float *slab = reinterpret_cast<float *>(lds_raw + (size_t)nrows_want * K * 2 + (size_t)K * 2);

    // ---- attention workgroups: first tile into registers before anything else is requested
    const int G = a.H / a.Hkv;
    int hk = 0, split = 0, lo = 0, hi = 0;
    bf16x8 qf[D / 32];
    RegKV<D> kv0;
    bool tile0 = false;
    const bf16_t *kb = nullptr, *vb = nullptr;
    if (attn_block) {
        hk = bid / a.nsplit; split = bid % a.nsplit;
        const int S = (int)a.st->len + 1;
        int per = (S + a.nsplit - 1) / a.nsplit;
        per = (per + 32 * NW - 1) / (32 * NW) * (32 * NW);
        lo = split * per; hi = min(S, lo + per);
        kb = a.kc + (size_t)hk * a.seq_alloc * D;
        vb = a.vT + (size_t)hk * D * a.seq_alloc;
#pragma unroll
        for (int dk = 0; dk < D / 32; dk++) {
            if (i < G) qf[dk] = ld_bf16x8(a.q + (size_t)(hk * G + i) * D + dk * 32 + g4 * 8);
            else {
#pragma unroll
                for (int j = 0; j < 8; j++) qf[dk][j] = (__bf16)0.f;
            }
        }
        tile0 = lo + 32 * wave < hi;
        if (tile0) kv0.load(kb, vb, a.seq_alloc, lo + 32 * wave, i, g4);
    }
    // ---- W_o slice -> LDS (1 KiB per wave instruction, linear image [row][K])
    {
        const int ninstr = (nrows * K * 2) >> 10;                       // K*2 is a multiple of 1 KiB (host-checked)
        const unsigned char *src = reinterpret_cast<const unsigned char *>(a.Wo + (size_t)row0 * K);
        for (int e = wave; e < ninstr; e += NW) glds16_w(src + (size_t)e * 1024 + lane * 16, wl + (size_t)e * 1024);
    }
    // ---- attention
    if (attn_block) {
        MfmaAttnState<D> s; s.init();
        if (tile0) attn_tile<D>(s, qf, kv0, lo + 32 * wave, 0, lo, hi, a.scale, lane);
        for (int kbase = lo + 32 * wave + 32 * NW; kbase < hi; kbase += 32 * NW) {
            const GlobalKV<D> kv{kb, vb, a.seq_alloc, kbase, i, g4};
            attn_tile<D>(s, qf, kv, kbase, 0, lo, hi, a.scale, lane);
        }
        mfma_state_to_lds<D, GMAX>(s, slab, wave, G, lane);
        __syncthreads();
        const bool wrote = decode_tail<bf16_t, D, GMAX, NW>(slab, &is_last, G, hk * G, hk, split, a.nsplit, a.part_m,
                                                             a.part_l, a.part_o, a.tickets, a.ao);
        if (wrote) {                                                    // publish this kv head's outputs
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(a.heads_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (nrows <= 0) return;                                             // workgroup-uniform

    // ---- wait for every kv head of THIS launch
    if (threadIdx.x == 0) {
        const unsigned target = (a.st->step + 1u) * (unsigned)a.Hkv;
        unsigned spins = 0;
        while (__hip_atomic_load(a.heads_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) { a.st_rw->error = 0xA77E; break; }       // bounded: never hang the GPU
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int c = threadIdx.x; c * 8 < K; c += NW * 64)
        *reinterpret_cast<uint4v *>(xs + c * 8) = *reinterpret_cast<const uint4v *>(a.ao + c * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // W_o slice has landed
    __syncthreads();

    // ---- rows of this workgroup from LDS
    const int nchunk = K >> 3;
    for (int r = wave; r < nrows; r += NW) {
        const bf16_t *wr = reinterpret_cast<const bf16_t *>(wl) + (size_t)r * K;
        float acc = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            float wv[8], xv[8];
            load8(wr + c * 8, wv);
            load8(xs + c * 8, xv);
#pragma unroll
            for (int j = 0; j < 8; j++) acc = fmaf(wv[j], xv[j], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) a.delta[row0 + r] = acc;
    }
}

// LDS budget and row split; returns false when the fused launch does not apply
bool attn_oproj_plan(int64_t H, int64_t Hkv, int64_t d, int64_t h, int nsplit, int cus, int *n_blocks, int *rows_attn,
                     int *rows_other, size_t *lds_bytes) {
    const int64_t K = H * d, G = H / Hkv;
    if (!(d == 64 || d == 128) || G > 8 || (K * 2) % 1024 || K % 8) return false;
    const int gmax = G <= 4 ? 4 : 8;
    const size_t xb = (size_t)K * 2, slab = (size_t)8 * gmax * (d + 2) * 4 + 256;        // x ; attention slabs
    const size_t budget = 160 * 1024 - 1024;
    if (xb + slab >= budget) return false;
    const int64_t max_other = (int64_t)((budget - xb) / (size_t)(K * 2));                 // no slabs there
    const int64_t max_attn = (int64_t)((budget - xb - slab) / (size_t)(K * 2));
    const int n_attn = (int)(Hkv * nsplit);
    if (max_other < 1 || n_attn > cus) return false;
    const int nb = cus;
    int ro, ra;
    if (nb > n_attn) {
        // attention workgroups should own as few rows as possible: the others take up to their LDS limit
        ro = (int)std::min<int64_t>(max_other, (h + (nb - n_attn) - 1) / (nb - n_attn));
        const int64_t left = h - (int64_t)ro * (nb - n_attn);
        ra = left > 0 ? (int)((left + n_attn - 1) / n_attn) : 0;
    } else {
        ro = 0; ra = (int)((h + n_attn - 1) / n_attn);
    }
    if (ra > max_attn) return false;
    *n_blocks = nb; *rows_attn = ra; *rows_other = ro;
    *lds_bytes = std::max((size_t)ra * K * 2 + xb + slab, (size_t)ro * K * 2 + xb);
    return true;
}

int launch_attn_oproj(Launcher &L, const void *q, const void *k_cache, const void *v_cache_T, const StepState *st,
                      StepState *st_rw, const AttnScratch &sc, void *ao, unsigned *heads_done, const void *Wo,
                      float *delta, int64_t H, int64_t Hkv, int64_t d, int64_t h, int64_t seq_alloc, float scale) {
    static int cus = 0;                                   // queried once (all shards run on the same kind of GPU)
    if (!cus) {
        int dev = 0; hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
    }
    int nb = 0, ra = 0, ro = 0; size_t lds = 0;
    if (!attn_oproj_plan(H, Hkv, d, h, sc.nsplit, cus, &nb, &ra, &ro, &lds))
        FL_FAIL(FL_ERR_UNSUPPORTED, "fused attention+o_proj launch does not fit this shape");
    AttnOprojArgs a;
    a.q = (const bf16_t *)q; a.kc = (const bf16_t *)k_cache; a.vT = (const bf16_t *)v_cache_T; a.st = st; a.st_rw = st_rw;
    a.part_m = sc.part_m; a.part_l = sc.part_l; a.part_o = sc.part_o; a.tickets = sc.counters;
    a.ao = (bf16_t *)ao; a.heads_done = heads_done; a.Wo = (const bf16_t *)Wo; a.delta = delta;
    a.H = (int)H; a.Hkv = (int)Hkv; a.seq_alloc = (int)seq_alloc; a.nsplit = sc.nsplit; a.h = (int)h; a.K = (int)(H * d);
    a.n_attn = (int)(Hkv * sc.nsplit); a.rows_attn = ra; a.rows_other = ro; a.scale = scale;
    const int G = (int)(H / Hkv);
    const double bytes = (double)h * H * d * 2 + 2.0 * (double)sc.kv_len_hint * Hkv * d * 2;
    const double flops = 2.0 * h * H * d + 4.0 * (double)sc.kv_len_hint * H * d;
    auto go = [&](auto kern) -> int {
        FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
        return L.launch(KC_ATTN_OPROJ, bytes, flops, kern, dim3((unsigned)nb), dim3(512), lds, a);
    };
    if (d == 128) return G <= 4 ? go(attn_oproj_kernel<128, 4>) : go(attn_oproj_kernel<128, 8>);
    return G <= 4 ? go(attn_oproj_kernel<64, 4>) : go(attn_oproj_kernel<64, 8>);
}

}  // namespace fl
