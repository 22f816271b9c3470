// attn_mfma.h -- device code shared by the MFMA attention kernels (k_attn_mfma.hip, k_attn_oproj.hip):
// the per-wave 32-key step on v_mfma_f32_16x16x32_bf16 and its operand sources.  See k_attn_mfma.hip
// for the layout derivation.
#pragma once
#include "attn_common.h"

namespace fl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline bf16x8 ld_bf16x8(const bf16_t *p) { return *reinterpret_cast<const bf16x8 *>(p); }

template <int D>
struct MfmaAttnState {
    float m, l;                 // of column q = lane & 15 (m identical in the four lane groups, l partial)
    float4v O[D / 16];          // rows q = 4*(lane>>4) + r, column d = db*16 + (lane & 15)
    __device__ void init() {
        m = -INFINITY; l = 0.f;
#pragma unroll
        for (int i = 0; i < D / 16; i++) O[i] = float4v{0.f, 0.f, 0.f, 0.f};
    }
};

// Operand sources of one 32-key step.  Global: straight from the caches (decode: every byte is used once).
template <int D>
struct GlobalKV {
    const bf16_t *kb, *vT; int ldv, kbase, i, g4;
    __device__ bf16x8 k_frag(int tile, int dk) const {
        const int key = kbase + 8 * (i >> 2) + (i & 3) + 4 * tile;
        return ld_bf16x8(kb + (size_t)key * D + dk * 32 + g4 * 8);
    }
    __device__ bf16x8 v_frag(int db) const { return ld_bf16x8(vT + (size_t)(db * 16 + i) * ldv + kbase + 8 * g4); }
};
// LDS: a K tile [32 keys][D] and a V^T tile [D][32 keys] staged by LDS-DMA and shared by the waves of a
// workgroup (prefill).  16-B chunks are XOR-swizzled (K: chunk ^ (row & (CPR-1)); V^T: chunk ^ ((row>>2)&3))
// on the DMA source address and on the read address, so the 16-lane ds_read_b128 groups spread over banks.
template <int D>
struct LdsKV {
    const unsigned char *kt, *vt; int i, g4;
    static constexpr int CPR = D / 8;
    __device__ bf16x8 k_frag(int tile, int dk) const {
        const int row = 8 * (i >> 2) + (i & 3) + 4 * tile, c = dk * 4 + g4;
        return *reinterpret_cast<const bf16x8 *>(kt + row * (D * 2) + ((c ^ (row & (CPR - 1))) << 4));
    }
    __device__ bf16x8 v_frag(int db) const {
        const int row = db * 16 + i;
        return *reinterpret_cast<const bf16x8 *>(vt + row * 64 + ((g4 ^ ((row >> 2) & 3)) << 4));
    }
};

// One 32-key step.  Column q sees key k iff  k < pre_hi  ||  (lo_q <= k && k < hi_q).
template <int D, typename KV>
__device__ __forceinline__ void attn_tile(MfmaAttnState<D> &s, const bf16x8 (&qf)[D / 32], const KV &kv, int kbase, int pre_hi,
                                 int lo_q, int hi_q, float scale, int lane) {
    const int g4 = lane >> 4;
    float4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dk = 0; dk < D / 32; dk++) {
        const bf16x8 a0 = kv.k_frag(0, dk);
        const bf16x8 a1 = kv.k_frag(1, dk);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[dk], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[dk], s1, 0, 0, 0);
    }
    // value fragments for this step: issued now, consumed after the softmax
    bf16x8 vb[D / 16];
#pragma unroll
    for (int db = 0; db < D / 16; db++) vb[db] = kv.v_frag(db);

    float p[8];
    float mt = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int key = kbase + 8 * g4 + j;
        const float sc = (j < 4 ? s0[j & 3] : s1[j & 3]) * scale;
        const bool ok = key < pre_hi || (key >= lo_q && key < hi_q);
        p[j] = ok ? sc : -INFINITY;
        mt = fmaxf(mt, p[j]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(s.m, mt);
    const bool dead = mn == -INFINITY;                         // this column has seen no visible key yet
    const float alpha = dead ? 1.0f : __expf(s.m - mn);
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) { p[j] = dead ? 0.f : __expf(p[j] - mn); ps += p[j]; }
    s.l = s.l * alpha + ps;
    s.m = mn;
    bf16x8 pa;
#pragma unroll
    for (int j = 0; j < 8; j++) pa[j] = (__bf16)p[j];
    float ar[4];
#pragma unroll
    for (int r = 0; r < 4; r++) ar[r] = __shfl(alpha, 4 * g4 + r, 64);     // factor of the O rows this lane holds
#pragma unroll
    for (int db = 0; db < D / 16; db++) {
#pragma unroll
        for (int r = 0; r < 4; r++) s.O[db][r] *= ar[r];
        s.O[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, vb[db], s.O[db], 0, 0, 0);
    }
}

// Registers: fragments of ONE 32-key step loaded ahead of time (so that other memory traffic can be
// issued behind them without delaying the step).
template <int D>
struct RegKV {
    bf16x8 k[2][D / 32], v[D / 16];
    __device__ void load(const bf16_t *kb, const bf16_t *vT, int ldv, int kbase, int i, int g4) {
        const GlobalKV<D> g{kb, vT, ldv, kbase, i, g4};
#pragma unroll
        for (int dk = 0; dk < D / 32; dk++) { k[0][dk] = g.k_frag(0, dk); k[1][dk] = g.k_frag(1, dk); }
#pragma unroll
        for (int db = 0; db < D / 16; db++) v[db] = g.v_frag(db);
    }
    __device__ bf16x8 k_frag(int tile, int dk) const { return k[tile][dk]; }
    __device__ bf16x8 v_frag(int db) const { return v[db]; }
};

// wave slab -> LDS in the shared format [wave][head][o[D], m, l] (attn_common.h)
template <int D, int GMAX>
__device__ __forceinline__ void mfma_state_to_lds(const MfmaAttnState<D> &s, float *lds, int wave, int G, int lane) {
    const int i = lane & 15, g4 = lane >> 4;
    float lt = s.l;
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    constexpr int STR = D + 2;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int qq = 4 * g4 + r;
        if (qq < G) {
            float *p = lds + ((size_t)wave * GMAX + qq) * STR;
#pragma unroll
            for (int db = 0; db < D / 16; db++) p[db * 16 + i] = s.O[db][r];
        }
    }
    if (g4 == 0 && i < G) {
        float *p = lds + ((size_t)wave * GMAX + i) * STR;
        p[D] = s.m; p[D + 1] = lt;
    }
}

}  // namespace fl
