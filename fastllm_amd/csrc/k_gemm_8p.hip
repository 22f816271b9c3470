// k_gemm_8p.hip -- prefill projection GEMM, 256x256 tile, phase-interleaved (8 phases per 2 K tiles).
//
//   Y[T,N] = X[T,K] . W[N,K]^T      bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
//
// The 128x128 / 256x128 kernels (k_gemm_mfma.hip) read 0.5 LDS fragments per MFMA and run every wave
// through "read fragments, then multiply" in lock step: LDS time ~ MFMA time, little of it overlapped,
// ~0.8 PFLOP/s.  This kernel follows the cdna programming guide's 256^2 structure:
//   * 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave (8 x 4 accumulator tiles, 128 VGPRs): 0.375
//     fragment reads per MFMA;
//   * a K tile (BK = 64) is consumed in 4 phases of 16 MFMAs, one output quadrant each, in the order
//     (m0,n0) (m0,n1) (m1,n1) (m1,n0), so that only one operand half is (re)read per phase
//     (12 / 4 / 8 / 0 ds_read_b128) and the B fragments of n0 stay in registers for the last phase;
//   * the two M wave groups run half a phase apart (one extra barrier for group 1 up front): one
//     group's MFMAs cover the other's fragment reads, both SIMD-resident waves alternate on the matrix core;
//   * LDS is cut into 16 KB half tiles by quadrant -- A0/A1 = the rows an m0 / m1 phase reads (of both M
//     groups), B0/B1 likewise -- two parities of 4 halves = 128 KB; every phase re-stages exactly one half
//     by LDS-DMA (global_load_lds x 2 per lane), at least two phases after its last ds_read and five
//     phases before its next use, so four half tiles (64 KB per CU) are always in flight;
//   * the loads cross barriers: every phase waits with a COUNTED s_waitcnt vmcnt(8) (the four youngest
//     half tiles stay outstanding) and raw s_barrier -- never vmcnt(0) in the steady state.
// Schedule (tile u, parity u&1):   phase:  1          2          3          4
//                                  reads:  A0,B0      B1         A1         --
//                                  stage:  B1(u+1)    A1(u+1)    A0(u+2)    B0(u+2)
// LDS image of a half tile: [128 rows][64 bf16], 16-byte chunk index XOR-swizzled with (row >> 1) & 7 on
// the DMA source address and on the read address (linear DMA destination): conflict-free ds_read_b128.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "attn_common.h"
#include "gemm_w4.h"
#include "kernels.h"

namespace fl {

typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));

constexpr int P_LDS = 2 * 4 * P_HALF;                  // 128 KiB

__device__ inline void glds16p(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ inline bf16x8p frag(const unsigned char *half, int row, int chunk) {
    const int pc = chunk ^ ((row >> 1) & 7);
    return *reinterpret_cast<const bf16x8p *>(half + row * 128 + pc * 16);
}
// tile row of local row r of half h:  A halves: M group r>>6, 64 rows each;  B halves: N group r>>5, 32 rows each
__device__ inline int b_row(int h, int r) { return (r >> 5) * 64 + h * 32 + (r & 31); }

// one half tile = 16 wave-instructions of 1 KiB (8 rows): two per wave
__device__ inline void glds16p_nt(const void *g, unsigned char *lds_wave_base) {   // non-temporal: a W panel nobody re-reads (one row tile)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 2);
}
template <bool IS_A, bool NT = false>
__device__ inline void stage_half(const bf16_t *__restrict__ M, int nrows, int K, int row0, int k0, int h,
                                  unsigned char *half, int wave, int lane) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int rb = (wave * 2 + s) * 8;
        const int r = rb + (lane >> 3), pc = lane & 7, c = pc ^ ((r >> 1) & 7);
        int gr = row0 + (IS_A ? a_row(h, r) : b_row(h, r));
        if (gr > nrows - 1) gr = nrows - 1;
        if constexpr (NT) glds16p_nt(M + (size_t)gr * K + k0 + c * 8, half + rb * 128);
        else glds16p(M + (size_t)gr * K + k0 + c * 8, half + rb * 128);
    }
}

// Stream-K workspace (SK instantiation; otherwise the grid is tiles x K slices, one segment per workgroup)
struct StreamK {
    float *part;          // [workgroups][2][256 * 256] fp32 partial accumulators, lane-major (16-B coalesced)
};

// STAMP: diagnostic instantiation (FL_8P_STAMPS=file): every workgroup records the 100 MHz wall clock and the core clock at
// entry / after the (first) prologue / after the (last) K loop / at exit; the production instantiation holds no stamp code.
//
// SK (stream-K): the grid is ONE workgroup per CU and the work is the line of (tile, K step) units, cut into equal
// pieces; a workgroup walks its piece as up to three segments -- the end of a tile another workgroup began, whole tiles,
// the beginning of a tile the next workgroup(s) finish.  A piece of a split tile only PUBLISHES its fp32 partial accumulators
// (slot 0: the piece ends the tile or lies inside it = the workgroup's first segment; slot 1: it begins the tile = its last
// segment); gemm_8p_fixup_kernel, launched behind it, adds the pieces of every split tile in K order and runs the tile's
// epilogue with one workgroup per 32-row block -- 8 x the tiles in parallel.  (A first version had the last arriver of a
// tile do this inside the launch, on a ticket: 17-32 workgroups then read 2 MB each while the rest of the chip idled --
// Qwen2-7B's 32-tile QKV tail 67 us.)  Nobody waits for another workgroup: the grid need not be co-resident.
template <bool STAMP, bool SK>
__global__ __launch_bounds__(512) void gemm_8p_kernel(const bf16_t *__restrict__ W, const bf16_t *__restrict__ X,
                                                      const float *__restrict__ bias, void *__restrict__ out,
                                                      int T, int N, int K, int epi, int tiles_m, int tiles_n,
                                                      const float *__restrict__ row_scale, int ksplit, int ldc,
                                                      unsigned long long *__restrict__ stamps, StreamK sk, ResidEpi re, RsParts rsp, int wnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [parity][A0 A1 B0 B1]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform values live in SGPRs)
    const int wr = wave >> 2, wc = wave & 3;
    const int m16 = lane & 15, kg = lane >> 4;
    unsigned long long st[8];
    if (STAMP) { st[0] = __builtin_amdgcn_s_memrealtime(); st[1] = __builtin_amdgcn_s_memtime(); }

    // XCD-aware remap (bijective): ids that share an XCD get consecutive tiles (same W panel in its L2)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    const int wid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int nk_all = K / P_BK;
    const int U = tiles_m * tiles_n * nk_all;                                     // stream-K: units on the line (host: < 2^31 / grid)
    auto cut = [&](int w) { return (int)((long long)U * w / nwg); };
    int u = SK ? cut(wid) : 0;
    const int u_end = SK ? cut(wid + 1) : 0;
    bool first_seg = true;

    for (;;) {
        // ---- this segment: tile li, K tiles [kt0, kt0 + nk) ----
        int li, kt0, nk;
        if (SK) {
            if (u >= u_end) break;
            li = u / nk_all; kt0 = u - li * nk_all;
            nk = min(nk_all - kt0, u_end - u);
            u += nk;
        } else {                                                                   // split-K: blockIdx.y owns a K slice and its own fp32 slab
            const int kz = blockIdx.y;
            li = wid;
            kt0 = (int)((long long)nk_all * kz / ksplit); nk = (int)((long long)nk_all * (kz + 1) / ksplit) - kt0;
        }
        const int tn = li / tiles_m, tm = li % tiles_m;
        const int m0 = tm * P_BM, n0 = tn * P_BN;
        const bf16_t *Xs = X + (size_t)kt0 * P_BK, *Ws = W + (size_t)kt0 * P_BK;

        float4v acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

        auto hbuf = [&](int tile, int which) -> unsigned char * { return lds + ((tile & 1) * 4 + which) * P_HALF; };   // which: 0 A0, 1 A1, 2 B0, 3 B1
        auto stA = [&](int h, int tile) { if (tile < nk) stage_half<true>(Xs, T, K, m0, tile * P_BK, h, hbuf(tile, h), wave, lane); };
        auto stB = [&](int h, int tile) {
            if (tile >= nk) return;
            if (wnt) stage_half<false, true>(Ws, N, K, n0, tile * P_BK, h, hbuf(tile, 2 + h), wave, lane);
            else stage_half<false>(Ws, N, K, n0, tile * P_BK, h, hbuf(tile, 2 + h), wave, lane);
        };

        // prologue: tile 0 whole, A0 / B0 of tile 1 (what phases 3, 4 of "tile -1" would have staged)
        stA(0, 0); stB(0, 0); stB(1, 0); stA(1, 0); stA(0, 1); stB(0, 1);
        if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();                    // group 1 runs half a phase behind (+10 %, gemm_probe)
        if (STAMP && first_seg) { st[2] = __builtin_amdgcn_s_memrealtime(); st[3] = __builtin_amdgcn_s_memtime(); }

        bf16x8p fa[4][2], fb0[2][2], fb1[2][2];
        const int arow = wr * 64 + m16, brow = wc * 32 + m16;
#define P_WAIT(TAIL)                                                                   \
        if (TAIL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#define P_MFMA(MI0, NJ0, FB)                                                           \
        __builtin_amdgcn_s_barrier();                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
        __builtin_amdgcn_sched_barrier(0);   /* pins the cluster between the barriers; s_setprio around it: -2.5 % */ \
        _Pragma("unroll") for (int ks = 0; ks < 2; ks++)                               \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                              \
                _Pragma("unroll") for (int j = 0; j < 2; j++)                          \
                    acc[MI0 + i][NJ0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], FB[j][ks], acc[MI0 + i][NJ0 + j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                             \
        __builtin_amdgcn_s_barrier();

        for (int t = 0; t < nk; t++) {
            const bool tail = t + 2 >= nk;                            // fewer than four young half tiles behind us
            const unsigned char *A0 = hbuf(t, 0), *A1 = hbuf(t, 1), *B0 = hbuf(t, 2), *B1 = hbuf(t, 3);
            // ---- phase 1: quadrant (m0, n0) ----
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) fb0[j][ks] = frag(B0, brow + j * 16, ks * 4 + kg);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) fa[i][ks] = frag(A0, arow + i * 16, ks * 4 + kg);
            stB(1, t + 1);
            P_WAIT(tail)
            P_MFMA(0, 0, fb0)
            // ---- phase 2: (m0, n1) ----
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) fb1[j][ks] = frag(B1, brow + j * 16, ks * 4 + kg);
            stA(1, t + 1);
            P_WAIT(tail)
            P_MFMA(0, 2, fb1)
            // ---- phase 3: (m1, n1) ----
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) fa[i][ks] = frag(A1, arow + i * 16, ks * 4 + kg);
            stA(0, t + 2);
            P_WAIT(tail)
            P_MFMA(4, 2, fb1)
            // ---- phase 4: (m1, n0) ----
            stB(0, t + 2);
            P_WAIT(tail)
            P_MFMA(4, 0, fb0)
        }
#undef P_MFMA
#undef P_WAIT
        if (wr == 0) __builtin_amdgcn_s_barrier();                    // balances group 1's extra barrier
        if (STAMP) { st[4] = __builtin_amdgcn_s_memrealtime(); st[5] = __builtin_amdgcn_s_memtime(); }
        __builtin_amdgcn_s_barrier();                                  // every wave is past its last fragment read: LDS is free
        first_seg = false;

        // (lane-derived addresses of the code below must not be hoisted over the K loop, where every register counts:
        // they hang off a thread id the optimiser cannot see through)
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));

        // ---- epilogue (store_rows above) or, for a piece of a split tile, publication of the partial accumulators ----
        if (!SK || nk == nk_all) {
            void *outz = out;
            if (!SK && ksplit > 1) outz = reinterpret_cast<float *>(out) + (size_t)blockIdx.y * T * N;
            float *rs_lds = reinterpret_cast<float *>(lds);
            if (tid_e < P_BM) rs_lds[tid_e] = row_scale_of(row_scale, rsp, min(m0 + tid_e, T - 1));
            __syncthreads();
            EpiCtx ctx;
            epi_ctx_init(ctx, outz, bias, rs_lds, T, N, epi, ldc, m0, n0, tn, wr, wc, tid_e, re);
            if (m0 + P_BM <= T && n0 + P_BN <= N) {
#pragma unroll
                for (int i = 0; i < 8; i++) store_rows<0>(ctx, i, acc[i]);
            } else if (n0 + P_BN <= N) {
#pragma unroll
                for (int i = 0; i < 8; i++) store_rows<2>(ctx, i, acc[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) store_rows<1>(ctx, i, acc[i]);
            }
        } else {
            float4v *pw = reinterpret_cast<float4v *>(sk.part + ((size_t)wid * 2 + (kt0 > 0 ? 0 : 1)) * (P_BM * P_BN)) + tid_e;
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) pw[(size_t)(i * 4 + j) * 512] = acc[i][j];
        }
        if (!SK) break;
        __syncthreads();                                               // the row scales are read: the next prologue may write LDS
    }
    if (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st[6] = __builtin_amdgcn_s_memrealtime(); st[7] = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xccid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xccid));
            unsigned long long *o = stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 10;
            for (int i = 0; i < 8; i++) o[i] = st[i];
            o[8] = hwid; o[9] = xccid;
        }
    }
}

// ---- the same tile on FOUR waves (2 x 2), one per SIMD, 128 x 128 outputs each -------------------------------------------
// Why: at eight waves a K tile costs 192 KB of ds_read_b128 per CU against 2048 cycles of MFMA -- 75 % of the LDS port -- and
// every phase pays two barriers for the hand-over between the two wave groups.  With 128 x 128 per wave (the 256 accumulator
// registers live in the AGPR half of a 512-register wave) the same tile reads 128 KB (0.25 fragment per MFMA) and a wave
// covers its own loads: while the 32 MFMAs of one output quadrant issue, the wave requests the 8 fragments the NEXT phase
// needs (into the register set the previous phase freed) and re-stages, by LDS-DMA, the half tile the previous phase read.
//   tile t (even):  phase   computes   with        requests            re-stages (read one phase ago)
//                   1       (m0,n0)    fa , fb0    fb1 <- B1(t)        B0(t+2)
//                   2       (m0,n1)    fa , fb1    fa2 <- A1(t)        B1(t+2)
//                   3       (m1,n1)    fa2, fb1    fa  <- A0(t+1)      A1(t+2)
//                   4       (m1,n0)    fa2, fb0    fb1 <- B1(t+1)      A0(t+3)
//   odd tiles run the mirror image (n0 <-> n1, fb0 <-> fb1, B0 <-> B1), so four fragment sets (128 VGPRs) suffice.
// One s_barrier per phase.  Before it every wave has waited for its fragment reads (lgkmcnt(0): the half may be re-staged
// behind the barrier) and for all but its 24 youngest DMA pieces (vmcnt(24): a half tile is read seven phases after it was
// requested, six halves of four pieces per wave stay in flight).  Past the last K tile the re-stage requests go on, clamped to
// the last tile (into halves nobody reads again), so the count holds to the end; vmcnt(0) + barrier before the epilogue.
// Where the time goes (Mistral-7B gate/up shape at T = 4096, random operands, TFLOP/s; timing-only builds, round 3):
//   eight-wave kernel 1168 | this one 1298-1337 | without its DMA pieces 1546 | without its fragment reads 1417 | neither 1695
//   (= the MFMA stream alone at the clock the chip holds on this data) | without the barrier 1302 | without the vmcnt 1293.
// The fillers cost ISSUE time: a wave's MFMAs leave 8 of every 16 cycles free, a ds_read_b128 takes ~14 and a DMA piece ~33
// (a plain global_load_dwordx4 in its place costs the same: 1287).  Tried and slower: the pieces in back-to-back pairs (1297),
// one wave issuing all four of a row block (1177), a copy of the loop per wave with the piece in a different gap (1182), the
// 32x32x16 MFMA shape with the same fillers in its 24-cycle gaps (1261: the chip holds a lower clock on that shape).
// hipBLASLt on the same operands: 1430-1470 (tools/gemm_vs_library.py).  Neutral: the pieces as buffer loads with one M0 write per
// phase and the K offset as the scalar offset (6 instructions fewer per phase: 1328 against 1311-1337) -- kept, the loop is cleaner.
// The four 1-KiB DMA pieces a wave adds to a half tile are buffer loads into LDS: lane offset (32-bit, precomputed) + the K tile's
// byte offset as the scalar offset + the piece's number x 1024 as the instruction offset, which moves the global address AND the
// LDS address -- so M0 (the LDS address of the wave's piece 0 in that half) is written once per phase and a piece is ONE
// instruction.  (The lane offsets carry 3072 - 1024 s and the descriptor's base lies 3072 bytes before the matrix, so that the
// instruction offset cancels out of the global address.)  hipcc does not count them: every wait is an explicit vmcnt.
template <bool SK>
__global__ __launch_bounds__(256) void gemm_4w_kernel(const bf16_t *__restrict__ W, const bf16_t *__restrict__ X,
                                                      const float *__restrict__ bias, void *__restrict__ out,
                                                      int T, int N, int K, int epi, int tiles_m, int tiles_n,
                                                      const float *__restrict__ row_scale, int ksplit, int ldc,
                                                      StreamK sk, ResidEpi re, int group_m, RsParts rsp, RopeEpi ro) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [parity][A0 A1 B0 B1]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wn = wave & 1;
    const int m16 = lane & 15, kg = lane >> 4;
    asm volatile("" : : : W4_AGPRS);                                    // the kernel descriptor allocates a[0:255]

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    const int wid = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
    const int nk_all = K / P_BK;
    const int U = tiles_m * tiles_n * nk_all;
    auto cut = [&](int w) { return (int)((long long)U * w / nwg); };
    int u = SK ? cut(wid) : 0;
    const int u_end = SK ? cut(wid + 1) : 0;

    for (;;) {
        int li, kt0, nk;
        if (SK) {
            if (u >= u_end) break;
            li = u / nk_all; kt0 = u - li * nk_all;
            nk = min(nk_all - kt0, u_end - u);
            u += nk;
        } else {
            const int kz = blockIdx.y;
            li = wid;
            kt0 = (int)((long long)nk_all * kz / ksplit); nk = (int)((long long)nk_all * (kz + 1) / ksplit) - kt0;
        }
        // tile order: groups of group_m row tiles, columns next, so that the 32 tiles an XCD works on together are a
        // group_m x (32 / group_m) block -- fewer distinct X / W panels per K step in its L2 than a 16 x 2 strip
        const int per_g = group_m * tiles_n, g0 = (li / per_g) * group_m, gm = min(tiles_m - g0, group_m), lr = li % per_g;
        const int tm = g0 + lr % gm, tn = lr / gm;
        const int m0 = tm * P_BM, n0 = tn * P_BN;
        const bf16_t *Xs = X + (size_t)kt0 * P_BK, *Ws = W + (size_t)kt0 * P_BK;

        // byte offsets of this lane's 16 bytes in each of its four 1-KiB pieces of the four half-tile kinds (k = 0)
        unsigned offA[2][4], offB[2][4];
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int r = (wave * 4 + s) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
                offA[h][s] = (unsigned)(((size_t)min(m0 + a_row(h, r), T - 1) * K + c * 8) * 2) + (W4_BIAS - 1024 * s);
                offB[h][s] = (unsigned)(((size_t)min(n0 + a_row(h, r), N - 1) * K + c * 8) * 2) + (W4_BIAS - 1024 * s);
            }
        w4_for<16>([&](auto c) { w4_zero16<decltype(c)::value * 16>(); });

        auto hbuf = [&](int tile, int which) -> unsigned char * { return lds + ((tile & 1) * 4 + which) * P_HALF; };   // 0 A0, 1 A1, 2 B0, 3 B1
        const int4w rsX = w4_rsrc(Xs), rsW = w4_rsrc(Ws);
        auto koff = [&](int tile) { return (unsigned)(min(tile, nk - 1) * (P_BK * 2)); };   // byte offset of a K tile (clamped past the end)
        const unsigned lds0 = (unsigned)(size_t)lds + wave * 4096;      // this wave's piece 0 of half tile 0 (LDS byte address)
        auto m0of = [&](int tile, int which) { return lds0 + ((tile & 1) * 4 + which) * P_HALF; };
        // all four pieces of a half (prologue), or piece q alone after the phase's w4_set_m0
        auto half4 = [&](bool isA, int h, int tile) {
            w4_set_m0(m0of(tile, isA ? h : 2 + h));
            const unsigned ko = koff(tile);
            const int4w rs = isA ? rsX : rsW;
            const unsigned *of = isA ? offA[h] : offB[h];
            w4_piece<0>(rs, of[0], ko); w4_piece<1>(rs, of[1], ko); w4_piece<2>(rs, of[2], ko); w4_piece<3>(rs, of[3], ko);
        };

        // prologue: tiles 0 and 1 whole, in the order they are read
        half4(true, 0, 0); half4(false, 0, 0); half4(false, 1, 0); half4(true, 1, 0);
        half4(true, 0, 1); half4(false, 1, 1); half4(false, 0, 1); half4(true, 1, 1);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");               // A0(0), B0(0) of this wave have landed
        __builtin_amdgcn_s_barrier();

        int4w fa[4][2], fa2[4][2], fb0[4][2], fb1[4][2];
        const int arow = wr * 64 + m16, brow = wn * 64 + m16;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                fa[i][ks] = frag4(hbuf(0, 0), arow + i * 16, ks * 4 + kg);
                fb0[i][ks] = frag4(hbuf(0, 2), brow + i * 16, ks * 4 + kg);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // everybody has read A0(0), B0(0)
        half4(true, 0, 2);                                              // ("phase 8 of tile -1")
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");

        // one phase: the 32 MFMAs of quadrant (NH, MI0) with 8 fragment requests and 4 DMA pieces between them, in this order
#define W4_FENCE __builtin_amdgcn_sched_barrier(0);
#define W4_PHASE(NH, MI0, FA, FB, RD, RHALF, RROW, ST_A, ST_H, ST_T)                                        \
        __builtin_amdgcn_s_barrier();                                                                       \
        W4_FENCE                                                                                            \
        const unsigned ko_##ST_A##ST_H = koff(ST_T);                                                        \
        w4_set_m0(m0of(ST_T, (ST_A) ? (ST_H) : 2 + (ST_H)));                                                \
        W4_FENCE                                                                                            \
        w4_for<4>([&](auto qc) {                                                                            \
            constexpr int q = decltype(qc)::value, a0 = 16 * (8 * (NH) + (MI0) + q);                        \
            w4_mfma<a0>(FA[q][0], FB[0][0]); W4_FENCE                                                       \
            RD[q][0] = frag4(RHALF, RROW + q * 16, kg);                                                     \
            W4_FENCE                                                                                        \
            w4_mfma<a0>(FA[q][1], FB[0][1]); W4_FENCE                                                       \
            w4_mfma<a0 + 4>(FA[q][0], FB[1][0]); W4_FENCE                                                   \
            RD[q][1] = frag4(RHALF, RROW + q * 16, 4 + kg);                                                 \
            W4_FENCE                                                                                        \
            w4_mfma<a0 + 4>(FA[q][1], FB[1][1]); W4_FENCE                                                   \
            w4_mfma<a0 + 8>(FA[q][0], FB[2][0]); W4_FENCE                                                   \
            w4_piece<q>((ST_A) ? rsX : rsW, (ST_A) ? offA[ST_H][q] : offB[ST_H][q], ko_##ST_A##ST_H);        \
            W4_FENCE                                                                                        \
            w4_mfma<a0 + 8>(FA[q][1], FB[2][1]); W4_FENCE                                                   \
            w4_mfma<a0 + 12>(FA[q][0], FB[3][0]); W4_FENCE                                                  \
            w4_mfma<a0 + 12>(FA[q][1], FB[3][1]); W4_FENCE                                                  \
        });                                                                                                 \
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);      /* lgkmcnt(0), as a builtin: hipcc then knows the fragments are in */ \
        W4_FENCE

#define W4_TILE(TT, NX, NY, FBX, FBY)                                                                       \
        {                                                                                                   \
            const int tt = (TT);                                                                            \
            W4_PHASE(NX, 0, fa, FBX, FBY, hbuf(tt, 2 + NY), brow, false, NX, tt + 2)                        \
            W4_PHASE(NY, 0, fa, FBY, fa2, hbuf(tt, 1), arow, false, NY, tt + 2)                             \
            W4_PHASE(NY, 4, fa2, FBY, fa, hbuf(tt + 1, 0), arow, true, 1, tt + 2)                           \
            W4_PHASE(NX, 4, fa2, FBX, FBY, hbuf(tt + 1, 2 + NY), brow, true, 0, tt + 3)                     \
        }

        int t = 0;
        for (; t + 1 < nk; t += 2) {
            W4_TILE(t, 0, 1, fb0, fb1)
            W4_TILE(t + 1, 1, 0, fb1, fb0)
        }
        if (t < nk) W4_TILE(t, 0, 1, fb0, fb1)
#undef W4_TILE
#undef W4_PHASE
#undef W4_FENCE
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the clamped re-stages past the last tile; the last MFMAs' results
        __builtin_amdgcn_s_barrier();                                   // LDS is free

        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        if (!SK || nk == nk_all) {
            void *outz = out;
            if (!SK && ksplit > 1) outz = reinterpret_cast<float *>(out) + (size_t)blockIdx.y * T * N;
            float *rs_lds = reinterpret_cast<float *>(lds);
            rs_lds[tid_e] = row_scale_of(row_scale, rsp, min(m0 + tid_e, T - 1));       // 256 threads, 256 rows
            __syncthreads();
            const bool whole = m0 + P_BM <= T && n0 + P_BN <= N;
            if (!SK && ro.on) {
                // EPI_QKV_ROPE (gemm_w4.h): a wave's 128 columns are one head of 128 or two of 64; 16 rows at a time through the
                // wave's own LDS staging tile (the K loop's ring is free), out as q / appended K / appended (transposed) V in bf16
                const RopeLane rl{&ro, rs_lds + wr * 128, bias, reinterpret_cast<float *>(lds + 4096) + (tid_e >> 6) * (16 * 132), T, N, m0 + wr * 128, tid_e & 63,
                                  ro.st->pos, ro.st->len};
                const int colw = n0 + wn * 128;
                w4_for<8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const float4v lo[4] = {w4_read<16 * i>(), w4_read<16 * i + 4>(), w4_read<16 * i + 8>(), w4_read<16 * i + 12>()};
                    const float4v hi[4] = {w4_read<16 * (8 + i)>(), w4_read<16 * (8 + i) + 4>(), w4_read<16 * (8 + i) + 8>(), w4_read<16 * (8 + i) + 12>()};
                    if (ro.d == 128) rope_rows128(rl, i, colw, lo, hi);
                    else { rope_rows64(rl, i, colw, lo); rope_rows64(rl, i, colw + 64, hi); }
                });
                return;
            }
            w4_for<2>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                EpiCtx ctx;
                epi_ctx_init(ctx, outz, bias, rs_lds, T, N, epi, ldc, m0, n0, tn, wr, wn * 2 + h, tid_e, re);
                w4_for<8>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, a0 = 16 * (8 * h + i);
                    const float4v v[4] = {w4_read<a0>(), w4_read<a0 + 4>(), w4_read<a0 + 8>(), w4_read<a0 + 12>()};
                    if (whole) store_rows<0>(ctx, i, v);
                    else if (n0 + P_BN <= N) store_rows<2>(ctx, i, v);
                    else store_rows<1>(ctx, i, v);
                });
            });
        } else {
            // the eight-wave kernel's lane-major layout (the fix-up kernel reads it): wave (wr, wc = 2 wn + h) of that kernel
            float4v *pw = reinterpret_cast<float4v *>(sk.part + ((size_t)wid * 2 + (kt0 > 0 ? 0 : 1)) * (P_BM * P_BN));
            w4_for<64>([&](auto c) {
                constexpr int e = decltype(c)::value, h = e >> 5, i = (e >> 2) & 7, j = e & 3;
                pw[(size_t)(i * 4 + j) * 512 + (wr * 4 + wn * 2 + h) * 64 + (tid_e & 63)] = w4_read<16 * (8 * h + i) + 4 * j>();
            });
        }
        if (!SK) break;
        __syncthreads();
    }
}

// Fix-up of a stream-K launch (same stream, right behind it): workgroup (tile, 32-row block i) adds the published pieces
// of a split tile in K order -- each thread the elements its lane held in the GEMM -- and runs the epilogue.
__global__ __launch_bounds__(512) void gemm_8p_fixup_kernel(const float *__restrict__ bias, void *__restrict__ out, int T, int N, int K, int epi,
                                                            int tiles_m, int tiles_n, const float *__restrict__ row_scale, int ldc, int nwg,
                                                            StreamK sk, ResidEpi re, int group_m) {
    __shared__ __attribute__((aligned(16))) float rs_lds[P_BM];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 2, wc = wave & 3;
    const int li = blockIdx.x >> 3, i = blockIdx.x & 7;
    const int nk_all = K / P_BK, U = tiles_m * tiles_n * nk_all;
    auto cut = [&](int w) { return (int)((long long)U * w / nwg); };
    const int t0 = li * nk_all, t1 = t0 + nk_all;
    int w0 = (int)((long long)t0 * nwg / U);                          // the workgroup whose piece holds the tile's K step 0
    while (cut(w0 + 1) <= t0) w0++;
    while (cut(w0) > t0) w0--;
    int nseg = 1;
    while (cut(w0 + nseg) < t1) nseg++;
    if (nseg == 1) return;                                            // the tile was computed whole: its owner ran the epilogue
    // (the GEMM's tile order: group_m = tiles_m is the eight-wave kernel's plain column-major walk)
    const int per_g = group_m * tiles_n, g0 = (li / per_g) * group_m, gm = min(tiles_m - g0, group_m), lr = li % per_g;
    const int tm = g0 + lr % gm, tn = lr / gm, m0 = tm * P_BM, n0 = tn * P_BN;
    if (tid < P_BM) rs_lds[tid] = row_scale ? row_scale[min(m0 + tid, T - 1)] : 1.0f;
    __syncthreads();
    float4v v[4] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
    const float4v *p0 = reinterpret_cast<const float4v *>(sk.part) + tid + (size_t)(i * 4) * 512;
    for (int sg = 0; sg < nseg; sg++) {
        const int wq = w0 + sg, slot = cut(wq) > t0 ? 0 : 1;
        const float4v *pq = p0 + ((size_t)wq * 2 + slot) * (P_BM * P_BN / 4);
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] += pq[(size_t)j * 512];
    }
    EpiCtx ctx;
    epi_ctx_init(ctx, out, bias, rs_lds, T, N, epi, ldc, m0, n0, tn, wr, wc, tid, re);
    if (m0 + P_BM <= T && n0 + P_BN <= N) store_rows<0>(ctx, i, v);
    else if (n0 + P_BN <= N) store_rows<2>(ctx, i, v);
    else store_rows<1>(ctx, i, v);
}

// Stream-K workspace of one stream: two partial tiles per workgroup
struct SkSpace { float *part = nullptr; int nwg = 0; };
static std::mutex g_sk_mu;
static std::map<std::pair<int, hipStream_t>, SkSpace> g_sk_spaces;    // 128 MiB per stream that runs long-prompt GEMMs, until the stream goes

// bytes of stream-K workspace held for `stream` on the current device (allocated by the first long-prompt forward)
int64_t gemm_8p_workspace_bytes(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(g_sk_mu);
    auto it = g_sk_spaces.find({dev, stream});
    return it == g_sk_spaces.end() ? 0 : (int64_t)it->second.nwg * 2 * P_BM * P_BN * (int64_t)sizeof(float);
}

// the owner of a stream calls this before destroying it (work on the stream has completed)
void gemm_8p_release_stream(hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_sk_mu);
    for (auto it = g_sk_spaces.begin(); it != g_sk_spaces.end();) {
        if (it->first.second == stream) {
            (void)hipFree(it->second.part);
            it = g_sk_spaces.erase(it);
        } else {
            ++it;
        }
    }
}

static int cu_count();
static int streamk_space(hipStream_t stream, int nwg, StreamK *sk) {
    int dev = 0;
    FL_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_sk_mu);
    SkSpace &sp = g_sk_spaces[{dev, stream}];
    if (sp.nwg < nwg) {
        // sized ONCE for a workgroup per CU (launch_gemm_8p never asks for more): no synchronise-and-free in the middle of a
        // launch sequence; gemm_8p_workspace_bytes() reports it to fl_model_info
        nwg = std::max(nwg, cu_count());
        if (sp.part) { FL_HIP(hipStreamSynchronize(stream)); (void)hipFree(sp.part); }
        sp = SkSpace{};
        FL_HIP(hipMalloc(&sp.part, (size_t)nwg * 2 * P_BM * P_BN * sizeof(float)));
        sp.nwg = nwg;
    }
    *sk = StreamK{sp.part};
    return FL_OK;
}

static int cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) cached[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    return cached[dev];
}

// does the four-wave form take this launch (the rule of the comment in launch_gemm_8p)?
bool gemm_4w_rule(int64_t T, int64_t N, int64_t K, int64_t ksteps, bool streamk) {
    const int four = tune(TK_GEMM_4W);
    return four && (four > 1 || (T >= 768 && ksteps >= 10 && K >= 3072 && (streamk || N >= 3072)));
}

int launch_gemm_8p(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                   int epi, const float *row_scale, int ksplit, int64_t ldc, bool streamk, const ResidEpi *resid) {
    if (ldc <= 0) ldc = N;
    if ((epi == EPI_RESID) != (resid != nullptr) || (resid && (ksplit != 1 || !resid->h || !resid->w || !resid->xn || !resid->part)))
        FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_8p: the residual epilogue takes its operands, whole K");
    const ResidEpi re = resid ? *resid : ResidEpi{};
    if (ksplit > 1 && ldc != N) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_8p: K slices write whole slabs (ldc == N)");
    if (streamk && ksplit != 1) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_8p: stream-K takes the whole K");
    if (streamk) FL_TRY(rs_parts_to_vector(L, row_scale, T));       // (a stream-K launch's fix-up takes its row scales as a vector)
    const int tiles_m = (int)((T + P_BM - 1) / P_BM), tiles_n = (int)((N + P_BN - 1) / P_BN);
    if (K % P_BK || K / P_BK / ksplit < 2) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_8p: K must give at least two 64-wide tiles per slice");
    if (ksplit > 1 && (bias || epi != EPI_F32)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "split-K GEMM: fp32 epilogue without bias only");
    const double bytes = ((double)N * K + (double)T * K) * 2.0;
    char tag[32];
    snprintf(tag, sizeof tag, "8p,%lldx%lld%s%s", (long long)N, (long long)K, streamk ? ",streamK" : ksplit > 1 ? ",splitK" : "", resid ? ",resid" : "");
    Launcher LL = L; LL.tag = tag;
    StreamK sk{nullptr};
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)ksplit);
    if (streamk) {
        // one workgroup per CU, but never more than one per two K steps of work
        const int64_t units = (int64_t)tiles_m * tiles_n * (K / P_BK);
        if (units >= (int64_t)1 << 30) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_8p: stream-K line too long");
        // pieces aligned with the tiles keep the workgroups that share a W or X panel in lock step (its L2 hits): split every
        // tile into the same number of pieces, at most eight, while the grid fits the chip
        const int64_t nt = (int64_t)tiles_m * tiles_n, cus = cu_count();
        const int sk_minsteps = std::max(2, tune(TK_SK_MINSTEPS));   // K steps per piece, at least (Qwen2-7B 4k QKV tail: piece launch + fix-up 58.2 us at 4, 53.8 at 8, 53.0 at 12)
        const int64_t split = std::max<int64_t>(1, std::min<int64_t>({(int64_t)8, cus / std::max<int64_t>(1, nt), (K / P_BK) / sk_minsteps}));
        const int nwg = (int)std::max<int64_t>(1, std::min<int64_t>(nt * split, units / 2));
        FL_TRY(streamk_space(L.stream, nwg, &sk));
        grid = dim3((unsigned)nwg, 1);
    }
    int group_m = tiles_m;                                          // tile order of the launch (the four-wave kernel groups row tiles)
    auto fixup = [&]() -> int {                                     // behind a stream-K launch: the split tiles' sums and epilogues
        if (!streamk) return FL_OK;
        Launcher LF = L; LF.tag = "8p,fixup";
        return LF.launch(KC_GEMM_MFMA, 0.0, 0.0, gemm_8p_fixup_kernel, dim3((unsigned)(tiles_m * tiles_n * 8)), dim3(512), 0, bias, y, (int)T, (int)N,
                         (int)K, epi, tiles_m, tiles_n, row_scale, (int)ldc, (int)grid.x, sk, re, group_m);
    };
    // the four-wave form of the same tile (same grid, workspace and fix-up).  FL_GEMM_4W: 0 never, 2 always, 1 (default) from 768
    // tokens, stream-K pieces and K slices of ten or more steps.  The rule comes from an A/B inside one process, whole prefills back
    // to back (tools/prefill_ab.py; ms, eight waves / four waves): Mistral-7B T = 512 8.75 / 8.83, 768 12.19 / 12.11, 1024 14.92 /
    // 14.14, 2048 27.47 / 25.47, 4096 52.09 / 48.11; Qwen2-7B 512 8.85 / 9.16, 4096 49.52 / 46.11.  Launch by launch the four-wave
    // kernel is 8-15 % faster at 512 tokens too (gate/up 116.7 -> 98.6 us between event pairs), but a sustained prefill of single-round
    // grids gives that back in clock; from two rounds of tiles on it keeps 5-8 %.
    // (its DMA pieces address a lane's bytes as a 32-bit offset from the matrix base: matrices of 4 GiB or more stay on eight waves)
    const bool fits32 = (double)std::max(T, N) * (double)K * 2.0 + (double)K * 2.0 + 8192.0 < 4294967296.0;   // (+ the K offset and the 3072-byte bias of the lane offsets)
    // ... on matrices of K >= 3072 and N >= 3072 only: TinyLlama-1.1B (K = 2048; down_proj 2048 x 5632, eight column tiles) lost
    // 3.5-5 % at 768-2048 tokens with it on either projection; with this rule it is untouched, Mistral-7B and Qwen2-7B keep
    // their gains (768 tokens -0.6 / -3.6 %, 1024 -5.2 / -4.2 %, 4096 -7.7..-8.6 / -6.9..-7.9 %)
    const int64_t ksteps = streamk ? K / P_BK : (K / P_BK) / ksplit;
    if (fits32 && !env_str("FL_8P_STAMPS") && gemm_4w_rule(T, N, K, ksteps, streamk)) {   // (N of a peeled tail is small: stream-K pieces go by K alone)
        auto k4 = streamk ? gemm_4w_kernel<true> : gemm_4w_kernel<false>;
        FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(k4), P_LDS));
        snprintf(tag, sizeof tag, "4w,%lldx%lld%s%s", (long long)N, (long long)K, streamk ? ",streamK" : ksplit > 1 ? ",splitK" : "", resid ? ",resid" : "");
        LL.tag = tag;
        const int group_env = tune(TK_GEMM_GROUPM);
        group_m = std::max(1, std::min(tiles_m, group_env > 0 ? group_env : 4));   // 4: +3-4 % at T = 4096 over the 16 x 2 strip (8: +2-3, 2: +1)
        FL_TRY(LL.launch(KC_GEMM_MFMA, bytes, 2.0 * T * N * K, k4, grid, dim3(256), P_LDS, (const bf16_t *)W, (const bf16_t *)x,
                         bias, y, (int)T, (int)N, (int)K, epi, tiles_m, tiles_n, row_scale, ksplit, (int)ldc, sk, re, group_m, L.rsp, RopeEpi{}));
        return fixup();
    }
    const int wnt8 = !streamk && tiles_m == 1 && T >= 176 && tune(TK_H4_NT) != 0;   // one row tile: the W panels are read once -- non-temporal
    const bool stamp = env_str("FL_8P_STAMPS") != nullptr;           // diagnostics only: synchronous, appends one record per launch
    auto kern = streamk ? (stamp ? gemm_8p_kernel<true, true> : gemm_8p_kernel<false, true>) : (stamp ? gemm_8p_kernel<true, false> : gemm_8p_kernel<false, false>);
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), P_LDS));
    if (stamp) {
        const char *path = env_str("FL_8P_STAMPS");
        const size_t nwg = (size_t)grid.x * grid.y;
        unsigned long long *d = nullptr;
        FL_HIP(hipMalloc(&d, nwg * 80));
        const int rc = LL.launch(KC_GEMM_MFMA, bytes, 2.0 * T * N * K, kern, grid, dim3(512), P_LDS, (const bf16_t *)W,
                                 (const bf16_t *)x, bias, y, (int)T, (int)N, (int)K, epi, tiles_m, tiles_n, row_scale, ksplit, (int)ldc, d, sk, re, L.rsp, wnt8);
        if (rc == FL_OK) FL_TRY(fixup());
        std::vector<unsigned long long> h(nwg * 10);
        FL_HIP(hipStreamSynchronize(L.stream));
        FL_HIP(hipMemcpy(h.data(), d, nwg * 80, hipMemcpyDeviceToHost));
        (void)hipFree(d);
        if (FILE *f = fopen(path, "a")) {
            fprintf(f, "launch %lld %lld %lld %d %zu\n", (long long)T, (long long)N, (long long)K, epi, nwg);
            for (size_t i = 0; i < nwg; i++) {
                for (int j = 0; j < 10; j++) fprintf(f, "%llu ", h[i * 10 + j]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
        return rc;
    }
    FL_TRY(LL.launch(KC_GEMM_MFMA, bytes, 2.0 * T * N * K, kern, grid, dim3(512), P_LDS, (const bf16_t *)W, (const bf16_t *)x,
                     bias, y, (int)T, (int)N, (int)K, epi, tiles_m, tiles_n, row_scale, ksplit, (int)ldc, (unsigned long long *)nullptr, sk, re, L.rsp, wnt8));
    return fixup();
}

// The QKV projection of a long prompt with EPI_QKV_ROPE on the four-wave kernel: whole K, plain tiles, no fp32 output -- the epilogue
// writes q, the appended K and the appended V (k_gemm_h4.hip does the same for mid-size prompts and for the peeled tail columns).
// N here is the column range this launch covers (whole heads: a multiple of 128); its heads are numbered from column 0 of W.
bool gemm_4w_rope_supported(int64_t T, int64_t N, int64_t K) {
    const bool fits32 = (double)std::max(T, N) * (double)K * 2.0 + (double)K * 2.0 + 8192.0 < 4294967296.0;
    return tune(TK_GEMM_4W) >= 1 && tune(TK_GEMM_8P) >= 1 && fits32 && K % P_BK == 0 && K / P_BK >= 2 && N % 128 == 0 && T >= 1 && !env_str("FL_8P_STAMPS");
}
int launch_gemm_4w_rope(Launcher &L, const void *W, const void *x, const float *bias, int64_t T, int64_t N, int64_t K, const float *row_scale,
                        const RopeEpi &rope) {
    if (!gemm_4w_rope_supported(T, N, K)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_4w: RoPE epilogue: shape not supported");
    if ((rope.d != 64 && rope.d != 128) || !rope.st || !rope.cos_tab || !rope.sin_tab || !rope.q_out || !rope.k_cache || !rope.v_cache)
        FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_4w: the RoPE epilogue takes its operands, head_dim 64 / 128");
    RopeEpi ro = rope;
    ro.on = 1;
    const int tiles_m = (int)((T + P_BM - 1) / P_BM), tiles_n = (int)((N + P_BN - 1) / P_BN);
    auto k4 = gemm_4w_kernel<false>;
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(k4), P_LDS));
    char tag[32];
    snprintf(tag, sizeof tag, "4w,%lldx%lld,rope", (long long)N, (long long)K);
    Launcher LL = L; LL.tag = tag;
    const int group_env = tune(TK_GEMM_GROUPM);
    const int group_m = std::max(1, std::min(tiles_m, group_env > 0 ? group_env : 4));
    return LL.launch(KC_GEMM_MFMA, ((double)N * K + (double)T * K) * 2.0, 2.0 * T * N * K, k4, dim3((unsigned)(tiles_m * tiles_n), 1), dim3(256), P_LDS,
                     (const bf16_t *)W, (const bf16_t *)x, bias, (void *)nullptr, (int)T, (int)N, (int)K, (int)EPI_QKV_ROPE, tiles_m, tiles_n, row_scale, 1, (int)N,
                     StreamK{nullptr}, ResidEpi{}, group_m, L.rsp, ro);
}

}  // namespace fl
