// k_attn_rep.hip -- decode attention + o_proj in ONE launch for SHORT caches: every workgroup computes the attention of ALL
// query heads itself (replicated), keeps the result in LDS and multiplies its own rows of W_o with it.
// (The T = 1 step of llama.rs:147-149 / mistral.rs:223-226 / qwen.rs:142)
//
// Why: at a few hundred cached positions the whole K / V of a layer is a few hundred KB (TinyLlama: 1 KB per position; one rank of
// a tensor-parallel Mistral-7B: 0.5-1 KB), the attention launch is a chain of round trips on a few dozen workgroups (4.3-7.8 us with
// HBM idle) and the o_proj launch behind it sits on the launch floor as well (4.4-4.5 us).  Here the 256 workgroups of the o_proj
// grid each read that K / V from their XCD's L2 (it comes from HBM once per XCD) -- redundant arithmetic, a few MFMA tiles per wave --
// while their rows of W_o, requested first, are on their way from HBM: no split-S partials, no ticket, no second launch, no
// hand-off between workgroups.  One dependent step of the layer's chain is gone (5 launches -> 4).
// The work grows with the cache length -- and a CU reads L2 at only ~40 GB/s -- so the host uses this form only for caches of up to
// ~96 KB of K / V per layer (attn_oproj_rep_supported; FL_ATTN_REP=0 turns it off).
//
// Workgroup: 8 waves.  Wave w attends for kv head w / nslice (its G query heads packed into the 16 MFMA columns, as in
// k_attn_mfma.hip) over the 32-key tiles t = w % nslice (mod nslice), nslice = 8 / Hkv; the fragments of up to TB tiles are requested
// in one burst (one round trip per TB tiles).  The slices of a head meet in LDS in slice order (fixed: reproducible), the normalised
// output is rounded to bf16 there -- the same rounding point as the attention launch's store -- and every wave finishes RPW rows:
// 16-byte weight chunks straight to registers (non-temporal), v_dot2c_f32_bf16, wave reduce, fp32 store (or the tensor-parallel
// exchange of comm_ll.h in the epilogue).
#include <stdlib.h>

#include "attn_mfma.h"
#include "comm_ll.h"

namespace fl {

typedef unsigned short ushort4v __attribute__((ext_vector_type(4)));

constexpr int R_NW = 8;                      // waves per workgroup
constexpr int R_KC = 4;                      // 512-element chunks of a W_o row a lane holds (K = H * d <= 2048)

template <int D, int GMAX, int RPW>
__global__ __launch_bounds__(R_NW * 64) void attn_oproj_rep_kernel(const AttnRepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float *slabs = reinterpret_cast<float *>(lds_raw);                                   // [8 waves][GMAX][D + 2]
    bf16_t *xs = reinterpret_cast<bf16_t *>(lds_raw + (size_t)R_NW * GMAX * (D + 2) * 4); // [H * D] attention output, bf16
    constexpr int TB = D == 64 ? 4 : 2;                                                  // tiles requested per burst (registers)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g4 = lane >> 4;
    const int H = a.H, Hkv = a.Hkv, G = H / Hkv, nslice = R_NW / Hkv;
    const int K = a.K, N = a.N, nchunk = K >> 3;

    // ---- this wave's rows of W_o: requested first, consumed last ----
    const int row0 = ((int)blockIdx.x * R_NW + wave) * RPW;
    uint4v w[RPW][R_KC];
#pragma unroll
    for (int r = 0; r < RPW; r++) {
        const bf16_t *wr = reinterpret_cast<const bf16_t *>(a.Wo) + (size_t)min(row0 + r, N - 1) * K;
#pragma unroll
        for (int u = 0; u < R_KC; u++) {
            const int ci = lane + 64 * u;
            w[r][u] = uint4v{0u, 0u, 0u, 0u};
            if (ci < nchunk && row0 + r < N) w[r][u] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(wr + (size_t)ci * 8));
        }
    }

    // ---- attention of kv head hk over the tiles sl, sl + nslice, ... ----
    const int hk = wave / nslice, sl = wave - hk * nslice;
    const int S = (int)a.st->len + 1;
    const bf16_t *q = reinterpret_cast<const bf16_t *>(a.q);
    bf16x8 qf[D / 32];
#pragma unroll
    for (int dk = 0; dk < D / 32; dk++) {
        if (i < G) qf[dk] = ld_bf16x8(q + (size_t)(hk * G + i) * D + dk * 32 + g4 * 8);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) qf[dk][j] = (__bf16)0.f;
        }
    }
    MfmaAttnState<D> s; s.init();
    const bf16_t *kb = reinterpret_cast<const bf16_t *>(a.kc) + (size_t)hk * a.seq_alloc * D;
    const bf16_t *vb = reinterpret_cast<const bf16_t *>(a.vT) + (size_t)hk * D * a.seq_alloc;
    for (int t0 = sl; t0 * 32 < S; t0 += nslice * TB) {
        RegKV<D> r[TB];
#pragma unroll
        for (int j = 0; j < TB; j++) {                       // (keys past S inside the cache's allocation are read and masked)
            // a tile is requested only if it exists: the unconditional form (every burst TB tiles, clamped) measured SLOWER at
            // every length but one -- the launch's time follows the bytes a CU requests, not the number of round trips
            const int kbase = (t0 + j * nslice) * 32;
            if (kbase < S) r[j].load(kb, vb, a.seq_alloc, kbase, i, g4);
        }
#pragma unroll
        for (int j = 0; j < TB; j++) {
            const int kbase = (t0 + j * nslice) * 32;
            if (kbase < S) attn_tile<D>(s, qf, r[j], kbase, 0, 0, S, a.scale, lane);
        }
    }
    mfma_state_to_lds<D, GMAX>(s, slabs, wave, G, lane);
    __syncthreads();
    // the slices of every head, in slice order -> normalised output, bf16, in LDS
    constexpr int STR = D + 2;
    for (int e = threadIdx.x; e < H * (D / 4); e += R_NW * 64) {
        const int hq = e / (D / 4), j4 = (e % (D / 4)) * 4;
        const int hk2 = hq / G, g = hq - hk2 * G;
        float M = -INFINITY;
        for (int z = 0; z < nslice; z++) M = fmaxf(M, slabs[((size_t)(hk2 * nslice + z) * GMAX + g) * STR + D]);
        float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < nslice; z++) {
            const float *p = slabs + ((size_t)(hk2 * nslice + z) * GMAX + g) * STR;
            const float wt = p[D] == -INFINITY ? 0.f : __expf(p[D] - M);
            L = fmaf(p[D + 1], wt, L);
#pragma unroll
            for (int j = 0; j < 4; j++) O[j] = fmaf(p[j4 + j], wt, O[j]);
        }
        const float inv = 1.0f / L;
        ushort4v pk;
#pragma unroll
        for (int j = 0; j < 4; j++) pk[j] = float_to_bf16_bits(O[j] * inv);
        *reinterpret_cast<ushort4v *>(xs + (size_t)hq * D + j4) = pk;
    }
    __syncthreads();

    // ---- the rows ----
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < R_KC; u++) {
        const int ci = lane + 64 * u;
        if (64 * u < nchunk) {                               // wave-uniform
            uint4v xr = *reinterpret_cast<const uint4v *>(xs + (size_t)min(ci, nchunk - 1) * 8);
            if (ci >= nchunk) xr = uint4v{0u, 0u, 0u, 0u};
#pragma unroll
            for (int r = 0; r < RPW; r++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[r] = dot2c_bf16(w[r][u][j], xr[j], acc[r]);
#pragma unroll
            for (int r = 0; r < RPW; r++) dot2c_settle(acc[r]);       // (a branch follows: the dot-result hazard window must close here)
        }
    }
    float sum[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) sum[r] = wave_sum(acc[r]);
    if (a.ll) {                                              // row-parallel projection of a tensor-parallel group: summed over the ranks here
        if (row0 < N) ll_allreduce_rows<RPW>(a.ll, a.ll_slot, row0, N, sum, a.out, lane);
        return;
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RPW; r++)
            if (row0 + r < N) a.out[row0 + r] = sum[r];
    }
}

// Would launch_attn_oproj_rep take this shape, with a cache of this capacity?
bool attn_oproj_rep_supported(int64_t H, int64_t Hkv, int64_t d, int64_t h, int64_t max_seq, bool any_size) {
    if (Hkv < 1 || H % Hkv || R_NW % Hkv) return false;
    const int64_t G = H / Hkv, K = H * d;
    if ((d != 64 && d != 128) || G > 8 || K % 8 || K > 512 * R_KC || h < 1) return false;
    const int64_t nslice = R_NW / Hkv, tiles = (max_seq + 31) / 32;
    // Measured (profiles/r03/README.md): a CU pulls the replicated K / V from its XCD's L2 at only ~40 GB/s (its own outstanding
    // requests, not the L2, are the limit), so the launch costs 6.5 us at 64 KB of K / V per layer, 10.3 at 160 KB, 13.7 at 330 KB,
    // against 8.6 us for the two launches it replaces: it pays up to ~96 KB (TinyLlama: 96 cached positions; one rank of an
    // 8-way Mistral-7B: 192), and that is where the host uses it.
    if (any_size) return (tiles + nslice - 1) / nslice <= 32;          // (FL_ATTN_REP=2: probes)
    return (tiles + nslice - 1) / nslice <= 8 && 2 * Hkv * max_seq * d * 2 <= 96 * 1024;
}

template <int D, int GMAX, int RPW>
static int launch_rep_t(Launcher &L, const AttnRepArgs &a, int blocks, size_t lds) {
    auto kern = attn_oproj_rep_kernel<D, GMAX, RPW>;
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    char tag[32];
    snprintf(tag, sizeof tag, "rep,%dx%d", a.N, a.K);
    Launcher LL = L; LL.tag = tag;
    return LL.launch(KC_ATTN_OPROJ, (double)a.N * a.K * 2, 2.0 * a.N * a.K, kern, dim3((unsigned)blocks), dim3(R_NW * 64), lds, a);
}

int launch_attn_oproj_rep(Launcher &L, const AttnRepArgs &a) {
    if (!a.q || !a.kc || !a.vT || !a.st || !a.Wo || !a.out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "attn_oproj_rep: null argument");
    const int D = a.K / std::max(1, a.H);
    if (a.H <= 0 || a.K != a.H * D || !attn_oproj_rep_supported(a.H, a.Hkv, D, a.N, 32, false)) FL_FAIL(FL_ERR_UNSUPPORTED, "attn_oproj_rep: unsupported shape");
    if (a.ll && a.ll_slot <= 0) FL_FAIL(FL_ERR_BAD_ARGUMENT, "attn_oproj_rep: fused all-reduce needs its slot");
    // rows per wave: one while that fills the chip's CUs, else two
    const int rpw = a.N > 256 * R_NW ? 2 : 1;
    const int blocks = (a.N + R_NW * rpw - 1) / (R_NW * rpw);
    const int G = a.H / a.Hkv, gmax = G <= 4 ? 4 : 8;
    const size_t lds = (size_t)R_NW * gmax * (D + 2) * 4 + (size_t)a.K * 2;
    if (D == 64) {
        if (gmax == 4) return rpw == 1 ? launch_rep_t<64, 4, 1>(L, a, blocks, lds) : launch_rep_t<64, 4, 2>(L, a, blocks, lds);
        return rpw == 1 ? launch_rep_t<64, 8, 1>(L, a, blocks, lds) : launch_rep_t<64, 8, 2>(L, a, blocks, lds);
    }
    if (gmax == 4) return rpw == 1 ? launch_rep_t<128, 4, 1>(L, a, blocks, lds) : launch_rep_t<128, 4, 2>(L, a, blocks, lds);
    return rpw == 1 ? launch_rep_t<128, 8, 1>(L, a, blocks, lds) : launch_rep_t<128, 8, 2>(L, a, blocks, lds);
}

}  // namespace fl
