// k_gemm_mfma.hip -- prefill projection GEMM on the gfx950 matrix cores.
//
//   Y[T,N] = X[T,K] . W[N,K]^T      bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
//
// 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
// tiles), BK = 64.  Both operands are K-contiguous, so X and W tiles have the same LDS image:
// [128 rows][64 bf16] = 128-B rows, filled by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 rows
// per wave-instruction) into two buffers; the 16-B chunk index is XOR-swizzled with (row>>1)&7
// on the SOURCE address and on the ds_read_b128 address (linear LDS destination), which makes
// the 16-lane groups of ds_read_b128 conflict-free for the 16x16x32 fragment pattern.
// Workgroup ids are remapped so that the tiles sharing a W panel run on one XCD (shared L2).
//
// Epilogues: fp32 store (+bias) or SiLU-gate on the 16-interleaved gate/up layout (tile column
// blocks alternate gate/up, so one lane holds gate[j] and up[j]).
#include <stdlib.h>

#include <algorithm>

#include "kernels.h"

namespace fl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile

__device__ inline void glds16(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
// non-temporal: a W panel one workgroup reads once (profiles/r04/README.md, ingest path)
__device__ inline void glds16_nt(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 2);
}

// stage one 128x64 tile: rows [row0, row0+128) of a [nrows][K] matrix at columns [k0, k0+64)
__device__ inline void stage_tile(const bf16_t *__restrict__ M, int nrows, int K, int row0, int k0,
                                  unsigned char *lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int r = wave * 32 + i * 8 + (lane >> 3);         // tile row written by this lane
        int pc = lane & 7;                               // physical 16-B chunk
        int c = pc ^ ((r >> 1) & 7);                     // logical chunk it must hold
        int gr = row0 + r; if (gr > nrows - 1) gr = nrows - 1;
        const bf16_t *src = M + (size_t)gr * K + k0 + c * 8;
        glds16(src, lds_tile + (wave * 32 + i * 8) * 128);
    }
}

__device__ inline bf16x8 read_frag(const unsigned char *lds_tile, int row, int chunk) {
    int pc = chunk ^ ((row >> 1) & 7);
    return *reinterpret_cast<const bf16x8 *>(lds_tile + row * 128 + pc * 16);
}

__global__ __launch_bounds__(256) void gemm_mfma_kernel(const bf16_t *__restrict__ W, const bf16_t *__restrict__ X,
                                                        const float *__restrict__ bias, void *__restrict__ out,
                                                        int T, int N, int K, int epi, int tiles_m, int tiles_n,
                                                        const float *__restrict__ row_scale, int ksplit, int ldc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [2 buffers][X tile | W tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware remap (bijective for any grid size): ids that share an XCD get consecutive tiles
    const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int li = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tn = li / tiles_m, tm = li % tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    float4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    // split-K: blockIdx.y owns K tiles [kt0, kt0 + nk) and writes its own fp32 slab (summed by the consumer)
    const int nk_all = K / BK, kz = blockIdx.y;
    const int kt0 = (int)((long long)nk_all * kz / ksplit), nk = (int)((long long)nk_all * (kz + 1) / ksplit) - kt0;
    X += (size_t)kt0 * BK; W += (size_t)kt0 * BK;
    if (ksplit > 1) out = reinterpret_cast<float *>(out) + (size_t)kz * T * N;
    stage_tile(X, T, K, m0, 0, lds, wave, lane);
    stage_tile(W, N, K, n0, 0, lds + TILE_BYTES, wave, lane);
    for (int kt = 0; kt < nk; kt++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned char *cur = lds + (kt & 1) * 2 * TILE_BYTES;
        if (kt + 1 < nk) {
            unsigned char *nxt = lds + ((kt + 1) & 1) * 2 * TILE_BYTES;
            stage_tile(X, T, K, m0, (kt + 1) * BK, nxt, wave, lane);
            stage_tile(W, N, K, n0, (kt + 1) * BK, nxt + TILE_BYTES, wave, lane);
        }
        const unsigned char *xt = cur, *wt = cur + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int chunk = ks * 4 + (lane >> 4);
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = read_frag(xt, wm * 64 + i * 16 + (lane & 15), chunk);
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = read_frag(wt, wn * 64 + j * 16 + (lane & 15), chunk);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // C/D map of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int cn = lane & 15, rm = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int m = m0 + wm * 64 + i * 16 + rm + rg;
            if (m >= T) continue;
            const float rs = row_scale ? row_scale[m] : 1.0f;
            if (epi == EPI_GATEUP) {
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int n = n0 + wn * 64 + j * 16 + cn;        // gate column; up = n + 16
                    if (n + 16 < N) {
                        const int qq = (n >> 5) * 16 + (n & 15);
                        float gt = acc[i][j][rg] * rs, up = acc[i][j + 1][rg] * rs;
                        float a = gt / (1.0f + expf(-gt)) * up;
                        reinterpret_cast<bf16_t *>(out)[(size_t)m * (ldc / 2) + qq] = float_to_bf16_bits(a);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int n = n0 + wn * 64 + j * 16 + cn;
                    if (n < N) reinterpret_cast<float *>(out)[(size_t)m * ldc + n] = acc[i][j][rg] * rs + (bias ? bias[n] : 0.f);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 256x128 tile, 8 waves (4 along M x 2 along N, 64x64 per wave: the same fragment code), BK = 64,
// THREE LDS stages (3 x 48 KiB).  The LDS-DMA of tile s+2 is issued while tile s is computed and is
// still in flight across the next workgroup barrier: the per-step wait is a COUNTED s_waitcnt vmcnt(6)
// (the six DMA instructions of the younger tile stay outstanding) followed by a raw s_barrier -- a
// __syncthreads() here would drain the DMA (its fence waits vmcnt(0)).  One workgroup per CU; each X
// row block is shared by two N-waves and each W tile by four M-waves, which also cuts the L2->LDS
// bytes per MAC by 25 % against the 128x128 kernel.  Used when T and the grid are large enough.
constexpr int BM2 = 256, STAGE2 = (BM2 + BN) * BK * 2;       // 48 KiB per stage

// stage `rows` x 64 of a [nrows][K] matrix; 8 waves, rows/64 LDS-DMA instructions per wave
template <int ROWS, bool NT = false>
__device__ inline void stage_rows8(const bf16_t *__restrict__ M, int nrows, int K, int row0, int k0,
                                   unsigned char *lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < ROWS / 64; i++) {
        const int rb = (i * 8 + wave) * 8;                       // 8 rows (1 KiB) per instruction
        const int r = rb + (lane >> 3), pc = lane & 7, c = pc ^ ((r >> 1) & 7);
        int gr = row0 + r; if (gr > nrows - 1) gr = nrows - 1;
        if constexpr (NT) glds16_nt(M + (size_t)gr * K + k0 + c * 8, lds_tile + rb * 128);
        else glds16(M + (size_t)gr * K + k0 + c * 8, lds_tile + rb * 128);
    }
}

__global__ __launch_bounds__(512) void gemm_mfma256_kernel(const bf16_t *__restrict__ W, const bf16_t *__restrict__ X,
                                                           const float *__restrict__ bias, void *__restrict__ out,
                                                           int T, int N, int K, int epi, int tiles_m, int tiles_n,
                                                           const float *__restrict__ row_scale, int ksplit, int ldc, int wnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [3 stages][X 256x64 | W 128x64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int li = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tn = li / tiles_m, tm = li % tiles_m;
    const int m0 = tm * BM2, n0 = tn * BN;

    float4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    // split-K: blockIdx.y owns K tiles [kt0, kt0 + nk) and writes its own fp32 slab (summed by the consumer)
    const int nk_all = K / BK, kz = blockIdx.y;
    const int kt0 = (int)((long long)nk_all * kz / ksplit), nk = (int)((long long)nk_all * (kz + 1) / ksplit) - kt0;
    X += (size_t)kt0 * BK; W += (size_t)kt0 * BK;
    if (ksplit > 1) out = reinterpret_cast<float *>(out) + (size_t)kz * T * N;
    auto stage = [&](int kt, int slot) {                          // 4 + 2 = 6 LDS-DMA instructions per wave
        unsigned char *base = lds + slot * STAGE2;
        stage_rows8<BM2>(X, T, K, m0, kt * BK, base, wave, lane);
        if (wnt) stage_rows8<BN, true>(W, N, K, n0, kt * BK, base + BM2 * BK * 2, wave, lane);   // (one row tile: nobody re-reads the panel)
        else stage_rows8<BN>(W, N, K, n0, kt * BK, base + BM2 * BK * 2, wave, lane);
    };
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    for (int kt = 0; kt < nk; kt++) {
        // tile kt has landed for this wave: its loads are older than the (at most) 6 of tile kt+1
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // everyone's tile kt landed; slot (kt+2)%3 is free
        if (kt + 2 < nk) stage(kt + 2, (kt + 2) % 3);
        const unsigned char *xt = lds + (kt % 3) * STAGE2, *wt = xt + BM2 * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int chunk = ks * 4 + (lane >> 4);
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = read_frag(xt, wm * 64 + i * 16 + (lane & 15), chunk);
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = read_frag(wt, wn * 64 + j * 16 + (lane & 15), chunk);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    const int cn = lane & 15, rm = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int m = m0 + wm * 64 + i * 16 + rm + rg;
            if (m >= T) continue;
            const float rs = row_scale ? row_scale[m] : 1.0f;
            if (epi == EPI_GATEUP) {
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int n = n0 + wn * 64 + j * 16 + cn;
                    if (n + 16 < N) {
                        const int qq = (n >> 5) * 16 + (n & 15);
                        float gt = acc[i][j][rg] * rs, up = acc[i][j + 1][rg] * rs;
                        float av = gt / (1.0f + expf(-gt)) * up;
                        reinterpret_cast<bf16_t *>(out)[(size_t)m * (ldc / 2) + qq] = float_to_bf16_bits(av);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int n = n0 + wn * 64 + j * 16 + cn;
                    if (n < N) reinterpret_cast<float *>(out)[(size_t)m * ldc + n] = acc[i][j][rg] * rs + (bias ? bias[n] : 0.f);
                }
            }
        }
    }
}

bool gemm_mfma_supported(int dtype, int64_t T, int64_t N, int64_t K) {
    return dtype == FL_DTYPE_BF16 && T > 1 && K % BK == 0 && K >= BK && N >= 1;
}

// ---- which kernel, how many K slices.  Three tiled kernels; the choice is the cheapest by a model fitted to the probes
// (tools/gemm_probe.py, prefill_profile.py, prefill_ragged.py), microseconds:
//   256x256 (k_gemm_8p.hip)  rounds of 256 tiles x (K steps x (1.0 + 0.6 fill) + 6)   one workgroup per CU
//   256x128                  rounds of 256 tiles x (K steps x 1.21 + 5)      where its own rules admit it (below)
//   128x128                  rounds of 512 tiles x (K steps x (0.95 + 0.33 fill) + 4)  two workgroups per CU
//   + K slices: the summing launch's read of the fp32 slabs at ~6 TB/s.
// Rounds 1 and 2 chose by fill thresholds (>= 192 tiles and 75-82 % of the last round, later half a round of useful
// fill); sweeping prompt lengths that are not round numbers showed what those miss: one token past a tile boundary put
// Mistral-7B's gate/up on the 128x128 kernel in two rounds (T = 257: 134 us against 83 at T = 256 and ~106 on the 256x256
// kernel), Qwen2-7B's at T = 255 likewise (151 us against 89 at 256).  The model reproduces the measured launches within
// ~10 % (T = 129 gate/up 128x128 86 / 82 measured; T = 256 256x128 82 / 83; Qwen2 T = 255 128x128 151 / 151, 256x256 93 / 89).
enum { GK_128 = 0, GK_256 = 1, GK_8P = 2 };
static const double kNoKernel = 1e30;
// (the write is this launch's ordinary epilogue, spread over ks times the workgroups; what is extra is the summing launch's read:
// rmsnorm_add with eight 8.4 MB slabs 13.3 us against 5-6 with one)
static double cost_slabs(int64_t T, int64_t N, int ks) { return ks > 1 ? (double)ks * T * N * 4.0 / 6.0e6 : 0.0; }
static double cost_8p(int64_t T, int64_t N, int64_t K, int ks) {
    const int mink = tune(TK_8P_MINK);   // K steps per slice: keep the pipeline long enough to pay for its ramp
    if (tune(TK_GEMM_8P) == 0) return kNoKernel;
    if (T <= 128 || K % 64 || ks < 1 || (K / 64) / ks < mink) return kNoKernel;
    const int64_t t8 = ((T + 255) / 256) * ((N + 255) / 256) * ks, rounds = (t8 + 255) / 256;
    // a K step takes 1.3-1.4 us on a grid that leaves a quarter of the chip idle and 1.5-1.6 us on a full one (clock and memory
    // contention): 1.0 + 0.6 x fill, per round; below one row tile every tile takes the bounds-checked epilogue (+8 us)
    const double last = (double)(t8 - (rounds - 1) * 256) / 256.0, steps = (double)(K / 64) / ks;
    return (double)(rounds - 1) * (steps * 1.6 + 6.0) + (steps * (1.0 + 0.6 * last) + 6.0) + (T < 256 ? 8.0 : 0.0) + cost_slabs(T, N, ks);
}
// the 256x128 kernel keeps its measured niches: very large grids that do not suit the 256x256 one, and -- with K slices -- a
// grid of exactly one round (224..256 workgroups, >= 8 K steps per slice: Mistral-7B T = 512 down_proj 80 -> 71.5 us in round 1)
static double cost_256(int64_t T, int64_t N, int64_t K, int ks) {
    const int use256 = tune(TK_GEMM_256);
    const int split256 = tune(TK_GEMM_256_SPLIT);
    if (!use256 || T < 192 || K % BK || K / BK < 3 || ks < 1) return kNoKernel;
    const int64_t tiles = ((T + BM2 - 1) / BM2) * ((N + BN - 1) / BN) * ks, rounds = (tiles + 255) / 256;
    const bool one_round = split256 && tiles >= 224 && tiles <= 256 && (K / BK) / ks >= 8;
    if (!((ks == 1 && tiles >= 1024) || one_round)) return kNoKernel;
    return (double)rounds * ((double)(K / BK) / ks * 1.21 + 5.0) + cost_slabs(T, N, ks);
}
static double cost_128(int64_t T, int64_t N, int64_t K, int ks) {
    if (ks < 1 || (ks > 1 && (K / BK) / ks < 8)) return kNoKernel;
    const int64_t tiles = ((T + BM - 1) / BM) * ((N + BN - 1) / BN) * ks, rounds = (tiles + 511) / 512;
    // the same dependence on fill as the 256x256 kernel: 1.12 us per K step at half a round (TinyLlama gate/up, T = 384), 1.28 full
    const double last = (double)(tiles - (rounds - 1) * 512) / 512.0, steps = (double)(K / BK) / ks;
    return (double)(rounds - 1) * (steps * 1.28 + 4.0) + (steps * (0.95 + 0.33 * last) + 4.0) + cost_slabs(T, N, ks);
}
static int pick_kernel(int64_t T, int64_t N, int64_t K, int ks, double *cost_out = nullptr) {
    const double c8 = cost_8p(T, N, K, ks), c2 = cost_256(T, N, K, ks), c1 = cost_128(T, N, K, ks);
    int k = GK_128; double c = c1;
    if (c2 < c) { k = GK_256; c = c2; }
    if (c8 < c) { k = GK_8P; c = c8; }
    if (cost_out) *cost_out = c;
    return k;
}

// how many K splits the launcher will use for this shape when the caller allows up to max_split slabs
static bool gemm_streamk_whole(int64_t T, int64_t N, int64_t K, int epi);
static bool peel_plan(int64_t T, int64_t N, int64_t K, int64_t *n_main_out);
int gemm_mfma_ksplit(int64_t T, int64_t N, int64_t K, int epi, int max_split) {
    if (gemm_streamk_whole(T, N, K, epi)) return 1;                 // one launch, partials meet inside it
    {   // whole rounds + a stream-K tail beat K slices of a grid that large (Mistral-7B o_proj at T = 4100: 272 tiles)
        int64_t n_main = 0;
        if (peel_plan(T, N, K, &n_main)) return 1;
    }
    int best = 1;
    double best_us = kNoKernel;
    for (int ks = 1; ks <= (epi == EPI_F32 ? std::max(1, max_split) : 1); ks++) {
        double c;
        (void)pick_kernel(T, N, K, ks, &c);
        if (c < best_us) { best = ks; best_us = c; }               // (ties: fewer slices)
    }
    return best;
}

// One launch over the column range this call was given (all of N, or a piece of a peeled matrix): ldc = the row stride
// of the full output; allow8p = false keeps the 256x256 kernel out (the tail of a peeled matrix).
static int launch_gemm_mfma_impl(Launcher &L, const void *W, const void *x, const float *bias, void *y,
                                 int64_t T, int64_t N, int64_t K, int epi, const float *row_scale, int ksplit, int64_t ldc, bool allow8p);

// A long prompt's grid of 256x256 tiles rarely fills whole rounds of the chip: Qwen2-7B at T = 4096 has 288 QKV tiles
// (1.125 rounds: the kernel was rejected and the projection ran on 128x128 tiles at 0.65 PFLOP/s), 2368 gate/up tiles
// (9.25 rounds: the last quarter round costs a full one) and 224 o_proj / down tiles (an eighth of the chip idle).  So the
// matrix is PEELED by columns: as many whole rounds as fit run as plain tiles, the remaining columns as a second launch
// that writes its own column range of the same output (ldc) -- in stream-K form (k_gemm_8p.hip: one workgroup per CU,
// the (tile, K step) line cut into equal pieces), or, with FL_GEMM_STREAMK=0, on the smaller tiles when the remainder
// is at most a quarter round.  A grid of less than one round runs stream-K whole when that saves more than it costs.
static int streamk_on() {
    return tune(TK_GEMM_STREAMK);
}
// Whole-matrix stream-K (FL_GEMM_STREAMK=2: grids of 96..255 tiles; =3: every shape the kernel takes -- tests).  Off by
// default: measured on Qwen2-7B's down_proj at T = 4096 (224 tiles, an eighth of the chip idle) it LOST 440 -> 535 us --
// pieces that start inside a tile take the workgroups that share a W or X panel out of lock step (their L2 hits become
// MALL/HBM reads), and every split tile moves its fp32 accumulators through memory twice (profiles/r02/README.md).
static bool gemm_streamk_whole(int64_t T, int64_t N, int64_t K, int epi) {
    const int sk = streamk_on();
    (void)epi;
    if (sk < 2 || tune(TK_GEMM_8P) != 1 || T < 256 || K % 64 || K / 64 < 8) return false;
    if (sk >= 3) return true;
    const int64_t t8 = ((T + 255) / 256) * ((N + 255) / 256);
    return T >= 1024 && K / 64 >= 32 && t8 >= 96 && t8 < 256;
}

// column peeling of a ragged 256x256 grid: whole rounds first (n_main columns), the rest as a stream-K (or 128x128) tail
static bool peel_plan(int64_t T, int64_t N, int64_t K, int64_t *n_main_out) {
    const int peel = tune(TK_GEMM_PEEL);
    const int use8p = tune(TK_GEMM_8P);
    if (!(peel && use8p == 1 && T >= 256 && K % 64 == 0 && (K / 64) >= 16)) return false;
    const int64_t tm = (T + 255) / 256, tn = (N + 255) / 256, t8 = tm * tn;
    const int64_t full = t8 / 256;
    const int64_t n_main_tiles = full * 256 / tm;                 // whole column tiles inside the full rounds
    const int64_t tail_tiles = t8 - tm * n_main_tiles;
    if (!(full >= 1 && tail_tiles > 0 && tail_tiles <= (streamk_on() ? 128 : 64) && n_main_tiles >= 1 && n_main_tiles < tn && tm * n_main_tiles >= 224))
        return false;
    *n_main_out = n_main_tiles * 256;                              // (256 | 32: gate/up pairs stay whole)
    return true;
}

int launch_gemm_mfma(Launcher &L, const void *W, const void *x, const float *bias, void *y,
                     int64_t T, int64_t N, int64_t K, int epi, const float *row_scale, int ksplit) {
    const int sk = streamk_on();
    if (ksplit == 1 && gemm_streamk_whole(T, N, K, epi))
        return launch_gemm_8p(L, W, x, bias, y, T, N, K, epi, row_scale, 1, N, true);
    int64_t n_main = 0;
    if (ksplit == 1 && peel_plan(T, N, K, &n_main)) {
        const int64_t n_tail = N - n_main;
        const size_t es_out = epi == EPI_GATEUP ? 2 : 4;
        const int64_t col_main = epi == EPI_GATEUP ? n_main / 2 : n_main;
        FL_TRY(launch_gemm_mfma_impl(L, W, x, bias, y, T, n_main, K, epi, row_scale, 1, N, true));
        const bf16_t *Wt = (const bf16_t *)W + (size_t)n_main * K;
        const float *bt = bias ? bias + n_main : nullptr;
        void *yt = (char *)y + (size_t)col_main * es_out;
        if (const int ks = gemm_h4_tail_slices(T, n_tail, K)) return launch_gemm_h4(L, Wt, x, bt, yt, T, n_tail, K, epi, row_scale, ks, N);
        if (sk) return launch_gemm_8p(L, Wt, x, bt, yt, T, n_tail, K, epi, row_scale, 1, N, true);
        return launch_gemm_mfma_impl(L, Wt, x, bt, yt, T, n_tail, K, epi, row_scale, 1, N, false);
    }
    return launch_gemm_mfma_impl(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit, N, true);
}

// ---- a long prompt's QKV projection with the RoPE / bias / KV-append epilogue (EPI_QKV_ROPE) on the four-wave kernel: where launch_linear
// would run it as ONE plain grid of 256 x 256 tiles, or as whole rounds + tail columns on the 128 x 256 kernel (1, 2 or 4 in-launch
// slices).  K slabs and stream-K pieces keep the fp32 output and the rope_kv_append launch (which sums the slabs anyway).
static bool qkv_rope_long_parts(int64_t T, int64_t N, int64_t K, int max_split, int64_t *n_main, int *tail_ks) {
    *n_main = N; *tail_ks = 0;
    if (!tune(TK_GEMM_ROPE_4W) || T < 768 || tune(TK_FORCE_GENERIC_GEMM) || tune(TK_GEMM_8P) != 1 || !gemm_mfma_supported(FL_DTYPE_BF16, T, N, K)) return false;
    if (gemm_streamk_whole(T, N, K, EPI_F32)) return false;
    int64_t nm = 0;
    if (peel_plan(T, N, K, &nm)) {
        const int ks = gemm_h4_tail_slices(T, N - nm, K);
        if (!ks || ks == 3 || nm % 128 || (N - nm) % 128) return false;
        *n_main = nm; *tail_ks = ks;
    } else if (gemm_mfma_ksplit(T, N, K, EPI_F32, max_split) != 1 || pick_kernel(T, N, K, 1) != GK_8P) {
        return false;
    }
    return gemm_4w_rope_supported(T, *n_main, K) && gemm_4w_rule(T, *n_main, K, K / 64, false);
}
bool gemm_qkv_rope_long_plan(int64_t T, int64_t N, int64_t K, int max_split) {
    int64_t nm; int ks;
    return qkv_rope_long_parts(T, N, K, max_split, &nm, &ks);
}
int launch_gemm_qkv_rope_long(Launcher &L, const void *W, const void *x, const float *bias, int64_t T, int64_t N, int64_t K, const float *row_scale,
                              const RopeEpi &rope, int max_split) {
    int64_t nm; int ks;
    if (!qkv_rope_long_parts(T, N, K, max_split, &nm, &ks)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_qkv_rope_long: not planned for this shape");
    FL_TRY(launch_gemm_4w_rope(L, W, x, bias, T, nm, K, row_scale, rope));
    if (nm == N) return FL_OK;
    RopeEpi rt = rope;
    rt.col_base = (int)nm;
    return launch_gemm_h4(L, (const bf16_t *)W + (size_t)nm * K, x, bias ? bias + nm : nullptr, nullptr, T, N - nm, K, EPI_QKV_ROPE, row_scale, ks, N - nm, nullptr, &rt);
}

// Would the kernel launch_linear picks for this projection take its row scales from a residual epilogue's partial sums (Launcher::rsp)?
// Mirrors launch_linear / launch_gemm_mfma: the 128 x 256 kernel, or a plain (no stream-K, no peeled tail) launch of the 256 x 256 ones.
bool gemm_takes_rs_parts(int dtype, int64_t T, int64_t N, int64_t K, int epi, int max_split) {
    if (dtype != FL_DTYPE_BF16 || T <= 1 || tune(TK_FORCE_GENERIC_GEMM)) return false;
    if (tune(TK_DEBUG_RS_PARTS)) return true;                        // (tests: a plan that is wrong on purpose)
    if (tune(TK_GEMM_SKF) >= 2 && gemm_skf_plan(T, N, K, epi) > 0 && (epi == EPI_GATEUP || tune(TK_GEMM_SKF) >= 3)) return true;   // short prompts: k_gemm_skf.hip sums the partials itself
    if (gemm_h4_plan(T, N, K, epi) > 0 || gemm_w14_plan(T, N, K, epi)) return true;
    if (tune(TK_GEMM_SKINNY) && gemm_skinny_supported(T, N, K)) return false;
    if (!gemm_mfma_supported(dtype, T, N, K)) return false;
    const int ks = epi == EPI_F32 ? gemm_mfma_ksplit(T, N, K, epi, max_split) : 1;
    if (ks == 1 && gemm_streamk_whole(T, N, K, epi)) return false;
    int64_t n_main = 0;
    if (ks == 1 && peel_plan(T, N, K, &n_main)) return false;
    return tune(TK_GEMM_8P) >= 1 && (tune(TK_GEMM_8P) >= 2 ? (K % 64 == 0 && (K / 64) / ks >= 2) : pick_kernel(T, N, K, ks) == GK_8P);
}

// ---- residual epilogue (EPI_RESID, kernels.h): only where the 256x256 kernel takes the whole K in one launch (or a peeled
// pair of launches): long prompts.  FL_GEMM_RESID=0 keeps the rmsnorm_add launches.
int gemm_resid_partials(int64_t N) { return (int)((N + 255) / 256) * 4; }
bool gemm_resid_supported(int dtype, int64_t T, int64_t N, int64_t K, int max_split) {
    if (tune(TK_GEMM_RESID) == 0) return false;
    if (dtype == FL_DTYPE_BF16 && tune(TK_GEMM_SKF) >= 2 && gemm_skf_plan(T, N, K, EPI_RESID) > 0) return true;   // short prompts: k_gemm_skf.hip (opt-in)
    if (dtype == FL_DTYPE_BF16 && N % 16 == 0 && gemm_h4_plan(T, N, K, EPI_RESID) > 0) return true;   // mid-size prompts: k_gemm_h4.hip
    if (dtype != FL_DTYPE_BF16 || tune(TK_GEMM_8P) != 1 || T < 256 || K % 64 || K / 64 < 2 || N % 16) return false;
    if (gemm_streamk_whole(T, N, K, EPI_F32)) return false;
    int64_t n_main = 0;
    if (peel_plan(T, N, K, &n_main)) return streamk_on() != 0;     // main launch + stream-K tail, both with the residual epilogue
    if (pick_kernel(T, N, K, 1) != GK_8P) return false;
    // against K slices + the rmsnorm_add launch that sums them (reads ks slabs and h, writes h and xn): the residual epilogue
    // reads h and writes xn itself and leaves a 4 us finalize.  The model is good to ~10 %, the epilogue has won every
    // measured tie (Mistral-7B down_proj at T = 3000: 314 + 4 us against 306 + 47 in four slices): it gets 15 %.
    const double tn = (double)T * N;
    double best_split = kNoKernel;
    for (int ks = 1; ks <= std::max(1, max_split); ks++) {
        double c;
        (void)pick_kernel(T, N, K, ks, &c);
        best_split = std::min(best_split, c - cost_slabs(T, N, ks) + tn * (4.0 * (ks + 2) + 2.0) / 6.0e6 + 3.0);
    }
    return cost_8p(T, N, K, 1) + tn * 6.0 / 6.0e6 + 4.3 <= 1.15 * best_split;
}
int launch_gemm_resid(Launcher &L, const void *W, const void *x, int64_t T, int64_t N, int64_t K, const ResidEpi &re) {
    if (re.np != gemm_resid_partials(N)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_resid: partial-sum layout");
    if (const int ks = tune(TK_GEMM_SKF) >= 2 ? gemm_skf_plan(T, N, K, EPI_RESID) : 0) return launch_gemm_skf(L, W, x, nullptr, nullptr, T, N, K, EPI_RESID, nullptr, ks, &re);
    if (const int ks = gemm_h4_plan(T, N, K, EPI_RESID)) return launch_gemm_h4(L, W, x, nullptr, nullptr, T, N, K, EPI_RESID, nullptr, ks, N, &re);
    int64_t n_main = 0;
    if (peel_plan(T, N, K, &n_main)) {
        FL_TRY(launch_gemm_8p(L, W, x, nullptr, nullptr, T, n_main, K, EPI_RESID, nullptr, 1, N, false, &re));
        ResidEpi rt = re;                                         // the tail's columns: same rows, later column tiles
        rt.h += n_main; rt.w += n_main; rt.xn = (bf16_t *)rt.xn + n_main; rt.part += (n_main / 256) * 4;
        if (const int ks = gemm_h4_tail_slices(T, N - n_main, K))
            return launch_gemm_h4(L, (const bf16_t *)W + (size_t)n_main * K, x, nullptr, nullptr, T, N - n_main, K, EPI_RESID, nullptr, ks, N, &rt);
        return launch_gemm_8p(L, (const bf16_t *)W + (size_t)n_main * K, x, nullptr, nullptr, T, N - n_main, K, EPI_RESID, nullptr, 1, N, true, &rt);
    }
    return launch_gemm_8p(L, W, x, nullptr, nullptr, T, N, K, EPI_RESID, nullptr, 1, N, false, &re);
}

static int launch_gemm_mfma_impl(Launcher &L, const void *W, const void *x, const float *bias, void *y,
                                 int64_t T, int64_t N, int64_t K, int epi, const float *row_scale, int ksplit, int64_t ldc, bool allow8p) {
    // 256x256 phase-interleaved kernel (k_gemm_8p.hip).  FL_GEMM_8P: 0 off, 1 where the model above prefers it, 2 always
    const int use8p = !allow8p ? 0 : tune(TK_GEMM_8P);
    const bool splittable = ksplit == 1 || (!bias && epi == EPI_F32);
    int kern = splittable ? pick_kernel(T, N, K, ksplit) : GK_128;
    if (!use8p && kern == GK_8P) kern = cost_256(T, N, K, ksplit) < cost_128(T, N, K, ksplit) ? GK_256 : GK_128;
    if (use8p >= 2 && K % 64 == 0 && (K / 64) / ksplit >= 2 && splittable) kern = GK_8P;
    if (kern == GK_8P) return launch_gemm_8p(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit, ldc);
    FL_TRY(rs_parts_to_vector(L, row_scale, T));                    // (the 128-column kernels take their row scales as a vector)
    if (kern == GK_256) {
        const int tm2 = (int)((T + BM2 - 1) / BM2), tn2 = (int)((N + BN - 1) / BN);
        const size_t lds2 = 3 * (size_t)STAGE2;                     // 144 KiB
        FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(gemm_mfma256_kernel), lds2));
        double bytes2 = ((double)N * K + (double)T * K) * 2.0;
        char tag2[32];
        snprintf(tag2, sizeof tag2, "256x128,%lldx%lld%s", (long long)N, (long long)K, ksplit > 1 ? ",splitK" : "");
        Launcher L2 = L; L2.tag = tag2;
        return L2.launch(KC_GEMM_MFMA, bytes2, 2.0 * T * N * K, gemm_mfma256_kernel, dim3((unsigned)(tm2 * tn2), (unsigned)ksplit), dim3(512),
                        lds2, (const bf16_t *)W, (const bf16_t *)x, bias, y, (int)T, (int)N, (int)K, epi, tm2, tn2, row_scale, ksplit, (int)ldc,
                        (int)(tune(TK_H4_NT) == 1 || (tune(TK_H4_NT) < 0 && tm2 == 1 && T >= 176)));
    }
    const int tiles_m = (int)((T + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
    if (ksplit > 1 && bias) FL_FAIL(FL_ERR_BAD_ARGUMENT, "split-K GEMM cannot add a bias");
    const size_t lds = 4 * TILE_BYTES;     // 64 KiB
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(gemm_mfma_kernel), lds));
    double bytes = ((double)N * K + (double)T * K) * 2.0;
    char tag[32];
    snprintf(tag, sizeof tag, "128x128,%lldx%lld%s", (long long)N, (long long)K, ksplit > 1 ? ",splitK" : "");
    Launcher LL = L; LL.tag = tag;
    return LL.launch(KC_GEMM_MFMA, bytes, 2.0 * T * N * K, gemm_mfma_kernel, dim3((unsigned)(tiles_m * tiles_n), (unsigned)ksplit),
                    dim3(256), lds, (const bf16_t *)W, (const bf16_t *)x, bias, y, (int)T, (int)N, (int)K, epi,
                    tiles_m, tiles_n, row_scale, ksplit, (int)ldc);
}

}  // namespace fl
