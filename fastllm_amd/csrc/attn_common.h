// attn_common.h -- pieces shared by the VALU (k_attn.hip) and MFMA (k_attn_mfma.hip) attention kernels:
// the LDS slab format, the cross-wave combine and the in-launch split-S combine.
//
// LDS slab: lds[wave][head][D + 2] floats = unnormalised o[D], running max m, running sum l.
#pragma once
#include "kernels.h"

namespace fl {

// combine the NW wave slabs in LDS for (head g, 4 d-elements at j4): returns M, L and O[4] (unnormalised)
template <int D, int GMAX, int NW>
__device__ inline void combine_lds(const float *lds, int g, int j4, float &M, float &L, float (&O)[4]) {
    constexpr int STR = D + 2;
    M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; w++) M = fmaxf(M, lds[((size_t)w * GMAX + g) * STR + D]);
    L = 0.f; O[0] = O[1] = O[2] = O[3] = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const float *p = lds + ((size_t)w * GMAX + g) * STR;
        const float wt = p[D] == -INFINITY ? 0.f : __expf(p[D] - M);
        L += p[D + 1] * wt;
#pragma unroll
        for (int j = 0; j < 4; j++) O[j] += p[j4 + j] * wt;
    }
}

// Tail of a decode-attention workgroup once its NW wave slabs are in LDS (and a barrier has passed).
// nsplit == 1: normalise and write the output.  Otherwise publish this split's (m, l, o) slab and
// let the workgroup that draws the last ticket of its (kv head, q-group) combine all splits:
//   plain stores -> per-wave vmcnt(0) -> barrier -> agent-scope release -> vmcnt(0) -> ticket;
//   last arriver: agent-scope acquire -> barrier -> plain loads
// (cdna guide Guideline 16, counter form; placement-independent).  The ticket word is reset by the last
// arriver, so a captured graph replays without a memset node.
// Returns true in the workgroup that wrote the final output of its heads.
template <typename CT, int D, int GMAX, int NW>
__device__ inline bool decode_tail(float *lds, int *is_last, int G, int hq0, int ticket_idx, int split, int nsplit,
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
        *reinterpret_cast<float4v *>(part_o + idx * D + j4) = float4v{O[0], O[1], O[2], O[3]};
        if (j4 == 0) { part_m[idx] = M; part_l[idx] = L; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned *cnt = counters + ticket_idx;
        const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (unsigned)nsplit - 1;
        if (last) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everyone has arrived
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *is_last = last;
    }
    __syncthreads();
    if (!*is_last) return false;
    // last arriver: stage the (m, l) of all splits in LDS, then combine the o slabs in parallel
    float *lm = lds, *ll = lds + GMAX * nsplit;                       // 2 * GMAX * nsplit floats (nsplit <= 64)
    for (int e = threadIdx.x; e < G * nsplit; e += NW * 64) {
        const int g = e / nsplit, sp = e % nsplit;
        const size_t idx = (size_t)(hq0 + g) * nsplit + sp;
        lm[g * nsplit + sp] = part_m[idx]; ll[g * nsplit + sp] = part_l[idx];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < G * (D / 4); e += NW * 64) {
        const int g = e / (D / 4), j4 = (e % (D / 4)) * 4;
        const size_t hb = (size_t)(hq0 + g) * nsplit;
        float M = -INFINITY;
        for (int sp = 0; sp < nsplit; sp++) M = fmaxf(M, lm[g * nsplit + sp]);
        float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int sp = 0; sp < nsplit; sp++) {
            const float mm = lm[g * nsplit + sp];
            const float w = mm == -INFINITY ? 0.f : __expf(mm - M);
            L += ll[g * nsplit + sp] * w;
            const float4v o4 = *reinterpret_cast<const float4v *>(part_o + (hb + sp) * D + j4);
#pragma unroll
            for (int j = 0; j < 4; j++) O[j] += o4[j] * w;
        }
        const float inv = 1.0f / L;
#pragma unroll
        for (int j = 0; j < 4; j++) elem<CT>::st(out + (size_t)(hq0 + g) * D + j4 + j, O[j] * inv);
    }
    return true;
}

}  // namespace fl
