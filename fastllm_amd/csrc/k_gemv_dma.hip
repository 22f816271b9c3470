// k_gemv_dma.hip -- the WIDE projections of a batched decode step or a very short prompt (3 <= B <= 32 rows; gate/up and lm_head: thousands
// of 16-row tiles): Y[b] = W[N,K] . x[b] on the matrix cores with the weights streamed by LDS-DMA into wave-private
// rings.  Row N4 of the scope table (mod.rs:137-238: the reference runs every stream as its own loop and pays for the
// whole weight read per stream; here one read serves the batch).
//
// Why another kernel (profiles/r02/README.md has the numbers): the register-staged MFMA GEMV (k_gemv_batch.hip) moves
// 4.3-4.6 TB/s and the short-prompt GEMM (k_gemm_skinny.hip) 4.6-5.3 with a barrier per K tile, an X tile re-staged
// per K tile (+25 % DMA traffic) and at most 48 KB of weights in flight per CU.  This one is organised like the
// single-sequence GEMV -- every wave is an independent streamer -- but with the MFMA doing the arithmetic (5.6-5.9 TB/s):
//   * a UNIT is one 16-row MFMA tile of W; a workgroup (8 waves, one per CU) walks its units; the 8 waves split a
//     unit's K tiles (64 k each) round-robin, so a row's consecutive 128-byte pieces are requested by sibling waves
//     at about the same time;
//   * a wave's K tiles never change, so its activation fragments (B rows x its K tiles) are loaded ONCE, into registers
//     (8 VGPRs per K tile), and no activation byte sits in LDS: all of it is ring;
//   * every wave owns 16 KB of ring filled by `global_load_lds` (non-temporal: each weight byte is read once by one CU):
//     four stages of 16 rows x 128 k where K allows (round 3: a 1-KB instruction then carries 4 rows x 256 contiguous bytes
//     instead of 8 x 128 -- longer runs per row are served better: gate/up 40.5 -> 39.6 us, lm_head 51.2 -> 47.4 at 8 streams),
//     else eight of 16 rows x 64 k; all but one stage in flight behind a COUNTED vmcnt; the ring is
//     wave-private, so the loop has no barrier at all, and it runs on across unit boundaries;
//   * per unit the 8 partial 16x8 tiles meet in LDS (4 KB, double-buffered: one barrier per unit) and one wave, in
//     rotation, sums them in wave order and runs the epilogue.
// A gate/up unit holds gate rows of 8 channels in slots 0-7 and their up rows in slots 8-15, exchanged with one xor-8
// shuffle for silu(gate) * up.
//
// The narrow projections of the step (QKV, o_proj, down_proj: one or two units per CU, so a unit's fixed costs are the
// kernel) run FASTER as K slices of the short-prompt GEMM, and a norm prologue built per lane costs ~10 us of L1
// throughput (8 cache lines per load instruction): both were built here, measured and removed (profiles/r02/README.md).
// Inputs: x = bf16 [B][K] (the norm launch's x * w), x_scale = its 1/rms per sequence.  EPI_F32 (K slices -> fp32 slabs)
// and EPI_GATEUP.
#include <stdlib.h>

#include <algorithm>

#include "kernels.h"

namespace fl {

typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));

constexpr int D_WAVES = 8, D_THREADS = D_WAVES * 64, D_STAGES = 8, D_STAGE_BYTES = 2048;
constexpr int D_RING_BYTES = D_WAVES * D_STAGES * D_STAGE_BYTES;      // 128 KB
constexpr int D_RED_FLOATS = 2 * D_WAVES * 64 * 4;                    // two buffers of 8 partial tiles (16 rows x 16 slots: round 5, was half a tile for <= 8 rows)

__device__ inline void glds16d(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 2);   // aux 2: nt
}
template <int N> __device__ inline void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// at most `n` (0 .. MAXN, wave-uniform) of this wave's youngest loads may still be in flight
template <int MAXN, int STEP = 2> __device__ inline void wait_vm_upto(int n) {
    if constexpr (MAXN <= 0) { wait_vm<0>(); }
    else { if (n >= MAXN) wait_vm<MAXN>(); else wait_vm_upto<MAXN - STEP, STEP>(n); }
}

// global row of slot i (0..15) of unit u
template <int EPI>
__device__ inline int unit_row(const GemvBatchArgs &a, int u, int i) {
    if constexpr (EPI == EPI_GATEUP) {
        const int c = u * 8 + (i & 7);                                 // channel
        return (c >> 4) * 32 + (c & 15) + ((i >> 3) << 4);             // 16-interleaved gate / up layout
    } else {
        return u * 16 + i;
    }
}

// NKT: K tiles per wave (compile-time: the activation fragments live in registers).
// KT: k per tile, 64 or 128 -- a stage is 16 rows x KT k (2 or 4 KB), so one 1-KB LDS-DMA instruction carries 8 rows x 128 B or
// 4 rows x 256 B: the longer the contiguous run per row, the better the HBM serves it (round 3; FL_DMA_KT).
// RB: blocks of 16 activation rows (1: up to 16 rows; 2: up to 32 -- two accumulator tiles per weight fragment, the partial tiles of a
// unit then meet in a single LDS buffer behind one more barrier per unit: the ring keeps its 128 KB)
template <int NKT, int EPI, int KT, int RB>
__global__ __launch_bounds__(D_THREADS) void gemv_dma_kernel(const GemvBatchArgs a) {
    constexpr int STAGE_BYTES = 16 * KT * 2, STAGES = (D_STAGES * D_STAGE_BYTES) / STAGE_BYTES;   // 16 KB of ring per wave either way
    constexpr int DPS = STAGE_BYTES / 1024;                     // LDS-DMA instructions per stage
    constexpr int RPI = 16 / DPS, LPR = 64 / RPI;               // rows per instruction, lanes (16-byte chunks) per row
    constexpr int SUB = KT / 32;                                // MFMA k steps per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [8 waves][8 stages][2 KB] | red | inv
    float *red = reinterpret_cast<float *>(lds + D_RING_BYTES);              // [2 units in flight][RB][8 waves][64 lanes][4]: 16 KB per row block -- with two, the CU's 160 KB to the byte
    const bf16_t *__restrict__ W = reinterpret_cast<const bf16_t *>(a.W);
    const int N = a.N, K = a.K, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // provably wave-uniform: task bookkeeping stays scalar
    const int m16 = lane & 15, kg = lane >> 4;
    const int ks_slice = blockIdx.y;
    // K slice of this workgroup in 64-wide tiles
    const int nt_all = K / KT;
    const int per = (nt_all + a.nks - 1) / a.nks;
    const int t0s = min(nt_all, ks_slice * per), nts = min(nt_all, t0s + per) - t0s;       // first tile, tiles of the slice
    const int nunits = EPI == EPI_F32 ? (N + 15) / 16 : N / 16;
    const int my_units = (int)blockIdx.x < nunits ? (nunits - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    unsigned char *ring = lds + (size_t)wave * STAGES * STAGE_BYTES;

    // wave-uniform: does local tile t of this wave exist?  (tile index within the slice: t * 8 + wave.  A contiguous range of
    // tiles per wave instead measured the same within noise: gate/up 39.7 -> 39.3 us, lm_head 47.1 -> 49.1)
    auto tile_ok = [&](int t) { return t * D_WAVES + wave < nts; };
    int my_tiles = 0;
#pragma unroll
    for (int t = 0; t < NKT; t++) my_tiles += tile_ok(t) ? 1 : 0;
    const int total = my_units * my_tiles;                          // this wave's (unit, tile) tasks, unit-major

    // ---- the weight stream starts first: tasks 0 .. S-2 of this wave ----
    // task f -> (unit index f / my_tiles, local tile f % my_tiles); stage f % S; two 1-KB instructions (8 rows each)
    int iu = 0, it = 0;                                                 // issue cursor: unit index (of mine), local tile
    auto issue = [&](int f) {
        const int u = (int)blockIdx.x + iu * (int)gridDim.x;
        const int kt = t0s + it * D_WAVES + wave;
        if (++it == my_tiles) { it = 0; iu++; }
        unsigned char *st = ring + (f & (STAGES - 1)) * STAGE_BYTES;
#pragma unroll
        for (int h = 0; h < DPS; h++) {
            const int slot = RPI * h + lane / LPR, pc = lane % LPR;
            const int c = KT == 64 ? pc ^ ((slot >> 1) & 7) : pc ^ (slot & 15);      // the read side's XOR swizzle, applied to the source
            int row = unit_row<EPI>(a, u, slot);
            if (row > N - 1) row = N - 1;
            glds16d(W + (size_t)row * K + (size_t)kt * KT + c * 8, st + h * 1024);
        }
    };
    int issued = 0;
    for (; issued < STAGES - 1 && issued < total; issued++) issue(issued);

    // ---- activation fragments of this wave's K tiles: lane (token m16, k group kg) ----
    uint4v xf[RB][NKT][SUB];
    const bf16_t *__restrict__ x = reinterpret_cast<const bf16_t *>(a.x);
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int t = 0; t < NKT; t++) {
#pragma unroll
            for (int s2 = 0; s2 < SUB; s2++) {
                xf[rb][t][s2] = uint4v{0, 0, 0, 0};
                if (tile_ok(t) && rb * 16 + m16 < B)
                    xf[rb][t][s2] = *reinterpret_cast<const uint4v *>(x + (size_t)(rb * 16 + m16) * K + (size_t)(t0s + t * D_WAVES + wave) * KT + s2 * 32 + kg * 8);
            }
        }
    // 1/rms of this lane's rows (tokens 16 rb + 4 kg + rg): registers, the LDS is all ring and partial tiles
    float inv[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int b = rb * 16 + kg * 4 + rg;
            inv[rb][rg] = (a.x_scale && b < B) ? a.x_scale[b] : 1.0f;
        }

    // ---- epilogue of one unit: the summed tile (tokens 4 * kg + reg, slot m16); lanes whose four tokens are all past B have nothing to do ----
    auto epilogue = [&](int u, float4v sum, int rb) {                 // rb: the row block (tokens 16 rb ..)
        if (rb * 16 + kg * 4 >= B) return;
        const int i = m16;
        float other[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            sum[rg] *= inv[rb][rg];
            other[rg] = __shfl_xor(sum[rg], 8, 64);                   // the partner slot's value (gate <-> up)
        }
        if constexpr (EPI == EPI_F32) {
            const int row = unit_row<EPI>(a, u, i);
            if (row >= N) return;
            float *out = reinterpret_cast<float *>(a.out) + (size_t)ks_slice * B * N;
            const float bv = (a.bias && ks_slice == 0) ? a.bias[row] : 0.f;
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const int b = rb * 16 + kg * 4 + rg;
                if (b < B) out[(size_t)b * N + row] = sum[rg] + bv;
            }
        } else {
            if (i >= 8) return;
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const int b = rb * 16 + kg * 4 + rg;
                if (b >= B) continue;
                const float gt = sum[rg], up = other[rg];
                const float act = gt / (1.0f + expf(-gt)) * up;        // candle silu(g) * u
                elem<bf16_t>::st(reinterpret_cast<bf16_t *>(a.out) + (size_t)b * (N / 2) + u * 8 + i, act);
            }
        }
    };

    // ---- main loop: units outer, this wave's K tiles inner (unrolled: xf[t] is a register name) ----
    int f = 0;                                                          // next task to consume
    for (int ui = 0; ui < my_units; ui++) {
        float4v acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; rb++) acc[rb] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NKT; t++) {
            if (!tile_ok(t)) continue;
            // task f has landed when at most the (issued - f - 1) younger tasks' loads (2 each) are outstanding
            wait_vm_upto<DPS * (STAGES - 2), DPS>(DPS * (issued - f - 1));
            const unsigned char *st = ring + (f & (STAGES - 1)) * STAGE_BYTES;
            bf16x8d wf[SUB];
#pragma unroll
            for (int s2 = 0; s2 < SUB; s2++) {
                const int chunk = s2 * 4 + kg;
                wf[s2] = *reinterpret_cast<const bf16x8d *>(st + m16 * (KT * 2) + ((chunk ^ (KT == 64 ? (m16 >> 1) & 7 : m16 & 15)) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the stage may be refilled from here on
            __builtin_amdgcn_sched_barrier(0);
            if (issued < total) { issue(issued); issued++; }           // into the stage task f - 1 used: its reads are done
#pragma unroll
            for (int rb = 0; rb < RB; rb++)
#pragma unroll
                for (int s2 = 0; s2 < SUB; s2++)
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8d, xf[rb][t][s2]), wf[s2], acc[rb], 0, 0, 0);
            f++;
        }
        // the 8 partial tiles meet in LDS; wave (ui % 8) sums them in wave order and runs the epilogue
        float *buf = red + (ui & 1) * (RB * D_WAVES * 64 * 4);          // (rewritten two units later: one barrier per unit orders it)
#pragma unroll
        for (int rb = 0; rb < RB; rb++)
            if (rb * 16 + kg * 4 < B) *reinterpret_cast<float4v *>(buf + ((rb * D_WAVES + wave) * 64 + lane) * 4) = acc[rb];
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < RB; rb++)
            if (wave == ((RB * ui + rb) & 7)) {                         // (two row blocks: two waves, side by side)
                float4v sum = {0.f, 0.f, 0.f, 0.f};
                if (rb * 16 + kg * 4 < B) {
#pragma unroll
                    for (int w = 0; w < D_WAVES; w++) {
                        const float4v p = *reinterpret_cast<const float4v *>(buf + ((rb * D_WAVES + w) * 64 + lane) * 4);
                        sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2]; sum[3] += p[3];
                    }
                }
                epilogue((int)blockIdx.x + ui * (int)gridDim.x, sum, rb);
            }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int cu_count_d() {
    int dev = 0; hipDeviceProp_t p;
    static int cached[64];
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) cached[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    return cached[dev];
}

// K tiles per wave for a slice count; 0 = the shape does not fit the kernel
static int dma_kt(int64_t K) {                                  // k per ring stage: 128 (256-byte row segments) where K allows it
    const int want = tune(TK_DMA_KT);
    return want == 128 && K % 128 == 0 ? 128 : 64;
}
static int dma_tiles_per_wave(int64_t K, int nks) {
    if (K % 64) return 0;
    const int kt = dma_kt(K);
    const int64_t nt = K / kt, per = (nt + nks - 1) / nks;
    const int64_t pw = (per + D_WAVES - 1) / D_WAVES;
    return pw <= 1024 / kt ? (int)pw : 0;                       // 16 tiles of 64 k or 8 of 128: 128 VGPRs of activation fragments
}

bool gemv_dma_supported(int B, int64_t N, int64_t K, int epi, int d) {
    (void)d;
    if (B < 1 || B > 32 || K % 64 || N < 1) return false;
    // 17-32 rows: two sets of activation fragments per wave -- half as many K tiles fit its registers (the whole K, unsliced)
    if (B > 16 && (dma_tiles_per_wave(K, 1) == 0 || dma_tiles_per_wave(K, 1) * (dma_kt(K) / 64) > 8)) return false;
    if (epi == EPI_GATEUP) return N % 32 == 0;
    return epi == EPI_F32;                                            // (its epilogue clamps a ragged last unit)
}

// K slices: the fewest that bring a wave's K tiles down to 16 (registers); for the fp32 epilogue (slabs summed by the
// consumer) more when the matrix has fewer units than the chip has CUs
int gemv_dma_ksplit(int64_t K, int64_t N, int epi) {
    int nks = 1;
    while (nks < 16 && dma_tiles_per_wave(K, nks) == 0) nks++;
    if (epi == EPI_F32 && N > 0) {
        const int64_t units = (N + 15) / 16;
        while (units * nks < cu_count_d() && nks < 4 && (K / 64) / (nks + 1) >= 16) nks++;
    }
    return nks;
}

template <int NKT, int EPI, int KT, int RB = 1>
static int launch_dma_e(Launcher &L, const GemvBatchArgs &a) {
    auto kern = gemv_dma_kernel<NKT, EPI, KT, RB>;
    const size_t lds = (size_t)D_RING_BYTES + (size_t)D_RED_FLOATS * 4 * RB;
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    const int64_t nunits = EPI == EPI_F32 ? (a.N + 15) / 16 : a.N / 16;
    const int cus = cu_count_d();
    // one workgroup per CU; with K slices the slices of a unit run side by side
    const int blocks = (int)std::min<int64_t>(nunits, std::max(1, cus / a.nks));
    char tag[32];
    snprintf(tag, sizeof tag, "b%dd:%dx%d%s", a.B, a.N, a.K, EPI == EPI_GATEUP ? ",glu" : "");
    Launcher LL = L; LL.tag = tag;
    return LL.launch(KC_GEMV, (double)a.N * a.K * 2, 2.0 * a.N * a.K * a.B, kern, dim3((unsigned)blocks, (unsigned)a.nks), dim3(D_THREADS), lds, a);
}

template <int NKT, int KT>
static int launch_dma_n(Launcher &L, const GemvBatchArgs &a) {
    return a.epi == EPI_GATEUP ? launch_dma_e<NKT, EPI_GATEUP, KT>(L, a) : launch_dma_e<NKT, EPI_F32, KT>(L, a);
}
template <int NKT, int KT>
static int launch_dma_n2(Launcher &L, const GemvBatchArgs &a) {        // 17-32 rows
    return a.epi == EPI_GATEUP ? launch_dma_e<NKT, EPI_GATEUP, KT, 2>(L, a) : launch_dma_e<NKT, EPI_F32, KT, 2>(L, a);
}

int launch_gemv_dma(Launcher &L, const GemvBatchArgs &a) {
    if (!gemv_dma_supported(a.B, a.N, a.K, a.epi, a.d)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv_dma: unsupported shape");
    if (a.pro != PRO_X || !a.x) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv_dma: activations are bf16 [B][K] (PRO_X)");
    if (a.nks < 1 || (a.nks > 1 && a.epi != EPI_F32)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv_dma: K slices only for the plain fp32 projection");
    const int pw = dma_tiles_per_wave(a.K, a.nks);
    if (pw == 0) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv_dma: too few K slices (a wave holds at most 16 K tiles of activations)");
    if (a.B > 16) {
        if (a.nks != 1) FL_FAIL(FL_ERR_BAD_ARGUMENT, "launch_gemv_dma: 17-32 rows take the whole K");
        if (dma_kt(a.K) == 128) return pw <= 2 ? launch_dma_n2<2, 128>(L, a) : launch_dma_n2<4, 128>(L, a);
        return pw <= 4 ? launch_dma_n2<4, 64>(L, a) : launch_dma_n2<8, 64>(L, a);
    }
    if (dma_kt(a.K) == 128) {
        if (pw <= 2) return launch_dma_n<2, 128>(L, a);
        if (pw <= 4) return launch_dma_n<4, 128>(L, a);
        if (pw <= 6) return launch_dma_n<6, 128>(L, a);
        return launch_dma_n<8, 128>(L, a);
    }
    if (pw <= 4) return launch_dma_n<4, 64>(L, a);
    if (pw <= 8) return launch_dma_n<8, 64>(L, a);
    if (pw <= 12) return launch_dma_n<12, 64>(L, a);
    return launch_dma_n<16, 64>(L, a);
}

}  // namespace fl
