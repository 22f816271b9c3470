// attn_common.h -- pieces shared by the VALU (k_attn.hip) and MFMA (k_attn_mfma.hip) attention kernels:
// the LDS slab format, the cross-wave combine and the in-launch split-S combine.
//
// LDS slab: lds[wave][head][D + 2] floats = unnormalised o[D], running max m, running sum l.
#pragma once
#include "kernels.h"

namespace fl {

// agent-scope write-through stores (global_store_dword / dwordx2 ... sc1): visible to every XCD once the storing wave's
// vmcnt has drained, no L2 write-back fence
__device__ inline void st_sc1(float *p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void st_sc1_x4(float *p, float4v v) {                       // p 16-byte aligned (one fabric write, not two)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ inline void st_sc1_pair(float *p, float a, float b) {              // p 8-byte aligned
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// floats of LDS a decode-attention workgroup needs: its NW wave slabs, or -- in the workgroup that combines the splits --
// the (m, l) of up to 64 splits plus one 5-float partial per thread
template <int D, int GMAX, int NW>
constexpr int decode_lds_floats() {
    return NW * GMAX * (D + 2) > 2 * GMAX * 64 + 5 * NW * 64 ? NW * GMAX * (D + 2) : 2 * GMAX * 64 + 5 * NW * 64;
}

// combine the NW wave slabs in LDS for (head g, 4 d-elements at j4): returns M, L and O[4] (unnormalised)
template <int D, int GMAX, int NW>
__device__ __forceinline__ void combine_lds(const float *lds, int g, int j4, float &M, float &L, float (&O)[4]) {
    constexpr int STR = D + 2;
    M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; w++) M = fmaxf(M, lds[((size_t)w * GMAX + g) * STR + D]);
    L = 0.f; O[0] = O[1] = O[2] = O[3] = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const float *p = lds + ((size_t)w * GMAX + g) * STR;
        const float wt = p[D] == -INFINITY ? 0.f : __expf(p[D] - M);
        L = fmaf(p[D + 1], wt, L);                                 // explicit FMAs: hipcc contracted the two inlined copies of this
#pragma unroll                                                     // loop differently (1-ulp differences between call sites)
        for (int j = 0; j < 4; j++) O[j] = fmaf(p[j4 + j], wt, O[j]);
    }
}

// Tail of a decode-attention workgroup once its NW wave slabs are in LDS (and a barrier has passed).
// nsplit == 1: normalise and write the output.  Otherwise publish this split's (m, l, o) slab and
// let the workgroup that draws the last ticket of its (kv head, q-group) combine all splits:
//   publishers:      write-through (sc1) stores -> per-wave vmcnt(0) -> barrier -> ticket   (no release: that fence
//                    writes back the XCD's whole L2, and hundreds of split workgroups each paid for one);
//   last arriver:    agent-scope acquire -> barrier -> plain loads, splits spread over all threads
// (cdna guide Guideline 16, counter form and R1 store forms; placement-independent).  The ticket word is reset by
// the last arriver, so a captured graph replays without a memset node.
// Returns true in the workgroup that wrote the final output of its heads.
template <typename CT, int D, int GMAX, int NW>
__device__ __forceinline__ bool decode_tail(float *lds, int *is_last, int G, int hq0, int ticket_idx, int split, int nsplit,
                                   float *__restrict__ part_m, float *__restrict__ part_l, float *__restrict__ part_o,
                                   unsigned *__restrict__ counters, CT *__restrict__ out) {
    if (nsplit == 1) {
        for (int e = threadIdx.x; e < G * (D / 4); e += NW * 64) {
            const int g = e / (D / 4), j4 = (e % (D / 4)) * 4;
            float M, L, O[4];
            combine_lds<D, GMAX, NW>(lds, g, j4, M, L, O);
            const float inv = 1.0f / L;
#pragma unroll
            for (int j = 0; j < 4; j++) elem<CT>::st(out + (size_t)(hq0 + g) * D + j4 + j, O[j] * inv);
        }
        return true;
    }
    for (int e = threadIdx.x; e < G * (D / 4); e += NW * 64) {
        const int g = e / (D / 4), j4 = (e % (D / 4)) * 4;
        float M, L, O[4];
        combine_lds<D, GMAX, NW>(lds, g, j4, M, L, O);
        const size_t idx = (size_t)(hq0 + g) * nsplit + split;
        // write-through (sc1) stores: no release fence is needed before the ticket (cdna guide Guideline 16, R1 store
        // forms).  The fence is a write-back of the XCD's whole L2; one per split workgroup made a long context's
        // attention (512 workgroups at S = 16 384) wait on 512 of them, and at a handful of splits -- where plain stores +
        // one release fence used to measure faster -- the write-through form is ahead as well since the combine's loads are
        // batched (7.3 -> 7.0 us at six splits).
        st_sc1_x4(part_o + idx * D + j4, float4v{O[0], O[1], O[2], O[3]});
        if (j4 == 0) { st_sc1(part_m + idx, M); st_sc1(part_l + idx, L); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *cnt = counters + ticket_idx;
        const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (unsigned)nsplit - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // everyone has arrived: the word is reset for the next launch BEHIND the acquire's wait (before it, the wait
            // also sat out this store's trip to memory)
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *is_last = last;
    }
    __syncthreads();
    if (!*is_last) return false;
    // last arriver: the (m, l) of all splits go to LDS, the o slabs are combined in parallel.
    // Long contexts have up to 64 splits and every partial o row is a trip to L2 / HBM.  The (head, 4-d) slots are spread
    // over ALL threads -- `parts` threads per slot take the splits round-robin -- and the parts meet in LDS.  Fixed order
    // (part 0, 1, ...), so results stay reproducible.  With at most eight splits per thread (the short contexts) they are
    // requested at once, unconditionally (past the last split a thread re-reads the last one with weight 0) and BEFORE the
    // (m, l) staging and its barrier: one round trip in all, where the remainder iterations of the streaming loop below wait
    // for their loads one by one (7.65 -> 7.45 us at six splits).  Longer contexts keep that loop (hoisting their first
    // eight loads cost Qwen2-7B at S = 4100 1.4 us per launch).
    const int slots = G * (D / 4), nthr = NW * 64;
    int parts = 1;
    while (parts * 2 * slots <= nthr && parts * 2 * 8 <= nsplit) parts *= 2;      // (a handful of splits: one pass, no LDS round)
    const int slot = threadIdx.x % slots, part = threadIdx.x / slots;
    const bool active = threadIdx.x < parts * slots;
    const int g = slot / (D / 4), j4 = (slot % (D / 4)) * 4;
    const size_t hb = (size_t)(hq0 + g) * nsplit;
    auto load_parts = [&](int s0, float4v (&o4)[8]) {
#pragma unroll
        for (int u = 0; u < 8; u++)
            o4[u] = *reinterpret_cast<const float4v *>(part_o + (hb + min(s0 + u * parts, nsplit - 1)) * D + j4);
    };
    const bool one_batch = nsplit <= 8 * parts;                       // (the short-context case; longer ones keep the streaming loop)
    float4v o4[8];
    if (one_batch && active) load_parts(part, o4);
    float *lm = lds, *ll = lds + GMAX * nsplit;                       // 2 * GMAX * nsplit floats (nsplit <= 64)
    for (int e = threadIdx.x; e < G * nsplit; e += NW * 64) {
        const int gg = e / nsplit, sp = e % nsplit;
        const size_t idx = (size_t)(hq0 + gg) * nsplit + sp;
        lm[gg * nsplit + sp] = part_m[idx]; ll[gg * nsplit + sp] = part_l[idx];
    }
    __syncthreads();
    float *red = ll + GMAX * nsplit;                                  // [parts][slots][5] floats behind lm / ll
    float M = -INFINITY;
    if (active)
        for (int sp = 0; sp < nsplit; sp++) M = fmaxf(M, lm[g * nsplit + sp]);
    if (active) {
        float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
        if (one_batch) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int sp = part + u * parts;
                const bool ok = sp < nsplit;
                const float mm = lm[g * nsplit + (ok ? sp : 0)];
                const float w = (!ok || mm == -INFINITY) ? 0.f : __expf(mm - M);
                L = fmaf(ll[g * nsplit + (ok ? sp : 0)], w, L);
#pragma unroll
                for (int j = 0; j < 4; j++) O[j] = fmaf(o4[u][j], w, O[j]);
            }
        } else {
#pragma unroll 8
            for (int sp = part; sp < nsplit; sp += parts) {
                const float mm = lm[g * nsplit + sp];
                const float w = mm == -INFINITY ? 0.f : __expf(mm - M);
                L = fmaf(ll[g * nsplit + sp], w, L);
                const float4v o = *reinterpret_cast<const float4v *>(part_o + (hb + sp) * D + j4);
#pragma unroll
                for (int j = 0; j < 4; j++) O[j] = fmaf(o[j], w, O[j]);
            }
        }
        if (parts > 1) {
            float *r = red + ((size_t)part * slots + slot) * 5;
            r[0] = L; r[1] = O[0]; r[2] = O[1]; r[3] = O[2]; r[4] = O[3];
        } else {
            const float inv = 1.0f / L;
#pragma unroll
            for (int j = 0; j < 4; j++) elem<CT>::st(out + (size_t)(hq0 + g) * D + j4 + j, O[j] * inv);
        }
    }
    if (parts > 1) {
        __syncthreads();
        if (threadIdx.x < slots) {
            float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
            for (int pt = 0; pt < parts; pt++) {
                const float *r = red + ((size_t)pt * slots + slot) * 5;
                L += r[0]; O[0] += r[1]; O[1] += r[2]; O[2] += r[3]; O[3] += r[4];
            }
            const float inv = 1.0f / L;
#pragma unroll
            for (int j = 0; j < 4; j++) elem<CT>::st(out + (size_t)(hq0 + g) * D + j4 + j, O[j] * inv);
        }
    }
    return true;
}

}  // namespace fl
