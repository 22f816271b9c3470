// k_gemm_h4.hip -- prefill projection GEMM for mid-size prompts: 128 x 256 tile on four waves, K slices summed INSIDE the launch.
//
//   Y[T,N] = X[T,K] . W[N,K]^T      bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
//
// Why: at 129..1024 tokens the row-parallel projections (o_proj, down_proj: N = h) and the QKV projection are 32-48 tiles of
// 256 x 256 on 256 CUs.  Round 2/3 cut K eight ways into fp32 slabs that the NEXT launch (rmsnorm_add, rope_kv) summed: 67 MB
// written and re-read per projection at T = 512, 13 us of every rmsnorm_add and ~12 us inside the GEMM's own epilogue.  Here the
// tile is half the size (twice the tiles, so at most FOUR K slices fill the chip: half the partial bytes) and the partials meet
// inside the launch, by STATIC OWNERSHIP (the protocol is written out where it runs, "the slices of a tile meet" below):
//   * a tile is eight blocks (an eighth of the tile each); slice kz of S owns blocks kz, kz + S, ...;
//   * when its K loop ends a workgroup publishes the blocks it does NOT own (write-through sc1 stores, lane-major 16-byte
//     layout), waits for its stores and counts itself on the tile's `published` word (one agent-scope atomic add);
//   * once all S slices are counted it loads the peers' copies of its OWN blocks, adds them to the accumulators it kept in
//     registers in K ORDER (its own at its place in the order: the sum does not depend on timing) and runs the epilogue of those
//     blocks: plain fp32 / bias / row scale, SiLU-gate, the residual epilogue of kernels.h (h += y, xn = (h + y) * w_next,
//     partial sums of squares: no rmsnorm_add launch) or RoPE + KV append (gemm_w4.h);
//   * nobody waits without bound for a workgroup that may not have started (processes may share the GPU): an early slice
//     waits `wait_ticks`, then ABANDONS its blocks -- publishes them too, sets their bits in the tile's flag word -- and leaves;
//     the slice counted last CLOSES the flag word after its own blocks and finishes whatever bits it finds, from memory alone; a
//     slice whose abandoning fetch_or finds the word closed finishes its blocks itself.  Every block is finished exactly once.
// The two words of a tile live in two sets used by alternate launches on a workspace; a launch zeroes the OTHER set, so nothing is
// reset inside a launch and a late waiter never sees a recycled word.
//
// Tile and K loop: four waves as 2 (M) x 2 (N), 64 x 128 outputs per wave (32 accumulator tiles = a[0:127], owned by inline asm
// as in gemm_4w_kernel).  A K tile (BK = 64) is three 16-KiB half tiles -- A (128 rows of X), B0 / B1 (the first / second 64 of
// each wave column's 128 rows of W) -- in a ring of THREE slots (144 KiB), and is consumed in two phases of 32 MFMAs:
//     tile t:  phase   computes        requests (fragments)                     re-stages by LDS-DMA
//              1       (n0) fa, fb0    fb1 <- B1(t); fa' rows 0-31 <- A(t+1)     B1(t+2) x 4 pieces, A(t+3) x 2
//              2       (n1) fa, fb1    fa' rows 32-63 <- A(t+1); fb0 <- B0(t+1)  A(t+3) x 2, B0(t+3) x 4
// (fa / fa' alternate by tile parity: four fragment sets = 128 VGPRs).  One s_barrier per phase; before it every wave has
// waited for its fragment reads (lgkmcnt(0)) and for all but its 18 (phase 1) / 16 (phase 2) youngest DMA pieces: a half tile
// is read three to four phases after it was requested.  The order of requests is the same from the prologue on (A(0) B0(0)
// B1(0) A(1) B0(1) B1(1) A(2) B0(2) | B1(2) A(3) B0(3) ...), so the counts hold from the first tile; past the last K tile the
// requests go on, clamped to the last tile, into slots nobody reads again.
// 1.5 x the fragment reads and DMA pieces per MFMA of the 256 x 256 tile (24 + 12 per 64 MFMAs): the price of the smaller tile.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "attn_common.h"
#include "gemm_w4.h"
#include "kernels.h"

namespace fl {

constexpr int H4_BM = 128, H4_BN = 256;
constexpr int H4_SLOT = 3 * P_HALF;                    // 48 KiB: A | B0 | B1
constexpr int H4_LDS = 3 * H4_SLOT;                    // 144 KiB
constexpr int H4_TILE_F4 = H4_BM * H4_BN / 4;          // float4 per partial tile (128 KiB)
constexpr int H4_MAX_TILES = 4096;                     // tiles of one launch (words of the tile protocol: two sets of this many pairs)

#define H4_AGPRS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135"

// (EPI_QKV_ROPE, the RoPE / bias / KV-append epilogue: gemm_w4.h, shared with the four-wave 256 x 256 kernel)

template <int S, bool WNT>
__global__ __launch_bounds__(256) void gemm_h4_kernel(const bf16_t *__restrict__ W, const bf16_t *__restrict__ X,
                                                      const float *__restrict__ bias, void *__restrict__ out,
                                                      int T, int N, int K, int epi, int tiles_m, int tiles_n,
                                                      const float *__restrict__ row_scale, int ldc,
                                                      H4Space ws, ResidEpi re, RopeEpi ro, int group_m, int pf_mode, int set, int wait_ticks, unsigned long long *stamps, RsParts rsp) {
    constexpr int ksplit = S;
    const int pf_dist = pf_mode & 255;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [slot][A B0 B1]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wn = wave & 1;
    // diagnostics (FL_H4_STAMPS): thread 0 leaves the 100 MHz wall clock at eight points of the workgroup's life
    auto stamp = [&](int i) { if (stamps && threadIdx.x == 0) stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    const int m16 = lane & 15, kg = lane >> 4;
    asm volatile("" : : : H4_AGPRS);                                    // the kernel descriptor allocates a[0:135]: accumulators + the prefetch sink a128

    // XCD-aware remap (bijective): ids that share an XCD get consecutive tiles; the slices of a tile share blockIdx.x
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    const int li = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
    const int kz = blockIdx.y;
    const int nk_all = K / P_BK;
    const int kt0 = (int)((long long)nk_all * kz / ksplit), nk = (int)((long long)nk_all * (kz + 1) / ksplit) - kt0;
    // tile order: groups of group_m row tiles, columns next (the tiles an XCD works on together share few X / W panels)
    const int per_g = group_m * tiles_n, g0 = (li / per_g) * group_m, gm = min(tiles_m - g0, group_m), lr = li % per_g;
    const int tm = g0 + lr % gm, tn = lr / gm;
    const int m0 = tm * H4_BM, n0 = tn * H4_BN;
    const bf16_t *Xs = X + (size_t)kt0 * P_BK, *Ws = W + (size_t)kt0 * P_BK;

    // byte offsets of this lane's 16 bytes in each of its four 1-KiB pieces of the three half-tile kinds (k = 0)
    unsigned offA[4], offB[2][4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int r = (wave * 4 + s) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        offA[s] = (unsigned)(((size_t)min(m0 + r, T - 1) * K + c * 8) * 2) + (W4_BIAS - 1024 * s);
#pragma unroll
        for (int h = 0; h < 2; h++)
            offB[h][s] = (unsigned)(((size_t)min(n0 + a_row(h, r), N - 1) * K + c * 8) * 2) + (W4_BIAS - 1024 * s);
    }
#ifdef FL_EXPERIMENTAL
    const bool blocked = (pf_mode >> 16) & 1;     // timing probe of the EXPERIMENTAL build only (WRONG RESULTS): W read as if stored K-tile-major, [N/256][K/64][256 rows][64]
#else
    constexpr bool blocked = false;
#endif
    if (blocked) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int r = (wave * 4 + s) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
            for (int h = 0; h < 2; h++)
                offB[h][s] = (unsigned)((size_t)n0 * K * 2 + (size_t)a_row(h, r) * 128 + c * 16) + (W4_BIAS - 1024 * s);
        }
    }
    // L2 prefetch of the W panel H4_PF K tiles ahead (one 128-byte line per row and K tile): the gm workgroups that share the panel
    // take 256 / gm rows each, a wave a quarter of those, one line per lane (spare lanes repeat the last one).  The load's data is
    // never used: it goes to a128, a register nothing else touches.
    unsigned offP;
    {
        const int lines_wg = (256 + gm - 1) / gm, lines_wave = (lines_wg + 3) / 4;
        const int sect = pf_mode >> 8 ? (pf_mode >> 8) : 1;                    // lanes per 128-byte line (1, 2 or 4: sectors of 128 / 64 / 32 bytes)
        const int ln = lane / sect, sc = lane % sect;
        const int rl = min(255, (tm - g0) * lines_wg + wave * lines_wave + min(ln, lines_wave - 1));
        offP = (unsigned)(((size_t)min(n0 + rl, N - 1) * K) * 2) + W4_BIAS + sc * (128 / sect);
        if ((pf_mode >> 8) == 0) offP = (unsigned)(((size_t)min(n0, N - 1) * K) * 2) + W4_BIAS;   // off: every lane the same line
    }
    w4_for<8>([&](auto c) { w4_zero16<decltype(c)::value * 16>(); });

    const int4w rsX = w4_rsrc(Xs), rsW = w4_rsrc(Ws);
#define H4_WP(S_, OFF_, KO_) w4_piece<S_, WNT>(rsW, OFF_, KO_)      /* a W piece: non-temporal where few row tiles share the panel */
    // byte offset of a K tile, clamped at both ends: past the last tile the requests go on into slots nobody reads, and a prefetch
    // distance below 3 asks for tiles before the first (an unclamped negative index became a ~4 GiB soffset, outside the buffer's
    // range check: ADVICE r4)
    auto koff = [&](int tile) { return (unsigned)(max(0, min(tile, nk - 1)) * (P_BK * 2)); };
    auto koffW = [&](int tile) { return blocked ? (unsigned)((kt0 + max(0, min(tile, nk - 1))) * 32768 - kt0 * (P_BK * 2)) : koff(tile); };
    const unsigned lds0 = (unsigned)(size_t)lds + wave * 4096;          // this wave's piece 0 of half 0 of slot 0 (LDS byte address)
    // which: 0 A, 1 B0, 2 B1
    auto m0of = [&](int slot, int which) { return lds0 + slot * H4_SLOT + which * P_HALF; };
    auto prefetch = [&](int tile) {
        asm volatile("buffer_load_dword a128, %0, %1, %2 offen" : : "v"(offP), "s"(rsW), "s"(koff(tile)) : "memory");
    };
    auto half4 = [&](int which, int slot, int tile) {                   // all four pieces of a half (prologue)
        w4_set_m0(m0of(slot, which));
        const unsigned ko = which == 0 ? koff(tile) : koffW(tile);
        const int4w rs = which == 0 ? rsX : rsW;
        const unsigned *of = which == 0 ? offA : offB[which - 1];
        if (which == 0) { w4_piece<0>(rs, of[0], ko); w4_piece<1>(rs, of[1], ko); w4_piece<2>(rs, of[2], ko); w4_piece<3>(rs, of[3], ko); }
        else { w4_piece<0, WNT>(rs, of[0], ko); w4_piece<1, WNT>(rs, of[1], ko); w4_piece<2, WNT>(rs, of[2], ko); w4_piece<3, WNT>(rs, of[3], ko); }
    };

    // prologue: the first eight half tiles of the request order
    prefetch(pf_dist - 3);
    half4(0, 0, 0); half4(1, 0, 0); prefetch(pf_dist - 2); half4(2, 0, 0);
    half4(0, 1, 1); half4(1, 1, 1); prefetch(pf_dist - 1); half4(2, 1, 1);
    half4(0, 2, 2); half4(1, 2, 2);
    asm volatile("s_waitcnt vmcnt(26)" ::: "memory");                   // A(0), B0(0) of this wave have landed
    __builtin_amdgcn_s_barrier();

    // fragment addresses: lane part (row, swizzled 16-byte chunk for ks = 0 / 1), then slot / half / row block as offsets
    const int arow = wr * 64 + m16, brow = wn * 64 + m16;
    const unsigned sw = (unsigned)((m16 >> 1) & 7);
    const unsigned la0 = arow * 128 + (((unsigned)kg ^ sw) << 4), la1 = arow * 128 + (((unsigned)(4 + kg) ^ sw) << 4);
    const unsigned lb0 = brow * 128 + (((unsigned)kg ^ sw) << 4), lb1 = brow * 128 + (((unsigned)(4 + kg) ^ sw) << 4);
    auto rdA = [&](int slot, int i, int ks) { return *reinterpret_cast<const int4w *>(lds + slot * H4_SLOT + (ks ? la1 : la0) + i * 2048); };
    auto rdB = [&](int slot, int h, int j, int ks) { return *reinterpret_cast<const int4w *>(lds + slot * H4_SLOT + (1 + h) * P_HALF + (ks ? lb1 : lb0) + j * 2048); };

    int4w fa[4][2], fa2[4][2], fb0[4][2], fb1[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            fa[i][ks] = rdA(0, i, ks);
            fb0[i][ks] = rdB(0, 0, i, ks);
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(17)" ::: "memory");                   // B1(0), A(1) too (the first phase reads them)

#define H4_FENCE __builtin_amdgcn_sched_barrier(0);
    // the 8 MFMAs of row block Q of n half NH, with three fragment requests and up to two DMA pieces between them
#define H4_Q(NH, Q, FA, FB, RD0, RD1, RD2, PC0, PC1)                                                        \
        {                                                                                                   \
            constexpr int a0 = 16 * (4 * (NH) + (Q));                                                       \
            w4_mfma<a0>(FA[Q][0], FB[0][0]); H4_FENCE                                                       \
            RD0; H4_FENCE                                                                                   \
            w4_mfma<a0>(FA[Q][1], FB[0][1]); H4_FENCE                                                       \
            w4_mfma<a0 + 4>(FA[Q][0], FB[1][0]); H4_FENCE                                                   \
            RD1; H4_FENCE                                                                                   \
            w4_mfma<a0 + 4>(FA[Q][1], FB[1][1]); H4_FENCE                                                   \
            w4_mfma<a0 + 8>(FA[Q][0], FB[2][0]); H4_FENCE                                                   \
            PC0; H4_FENCE                                                                                   \
            w4_mfma<a0 + 8>(FA[Q][1], FB[2][1]); H4_FENCE                                                   \
            RD2; H4_FENCE                                                                                   \
            w4_mfma<a0 + 12>(FA[Q][0], FB[3][0]); H4_FENCE                                                  \
            PC1; H4_FENCE                                                                                   \
            w4_mfma<a0 + 12>(FA[Q][1], FB[3][1]); H4_FENCE                                                  \
        }
#define H4_NONE (void)0
    // one K tile: S0 / S1 / S2 = ring slots of tiles t, t + 1, t + 2 (tile t + 3 goes to S0); FA current, FN next A set
#define H4_TILE(TT, S0, S1, S2, FA, FN)                                                                     \
        {                                                                                                   \
            const int tt = (TT);                                                                            \
            const unsigned ko2 = koffW(tt + 2), ko3 = koff(tt + 3), ko3w = koffW(tt + 3);                                          \
            /* ---- phase 1: n half 0 ---- */                                                               \
            __builtin_amdgcn_s_barrier();                                                                   \
            H4_FENCE                                                                                        \
            prefetch(tt + pf_dist);                                                                          \
            w4_set_m0(m0of(S2, 2));                                                                         \
            H4_FENCE                                                                                        \
            H4_Q(0, 0, FA, fb0, fb1[0][0] = rdB(S0, 1, 0, 0), fb1[0][1] = rdB(S0, 1, 0, 1), fb1[1][0] = rdB(S0, 1, 1, 0),          \
                 H4_WP(0, offB[1][0], ko2), H4_WP(1, offB[1][1], ko2))                      \
            H4_Q(0, 1, FA, fb0, fb1[1][1] = rdB(S0, 1, 1, 1), fb1[2][0] = rdB(S0, 1, 2, 0), fb1[2][1] = rdB(S0, 1, 2, 1),          \
                 H4_WP(2, offB[1][2], ko2), H4_WP(3, offB[1][3], ko2))                      \
            w4_set_m0(m0of(S0, 0));                                                                         \
            H4_FENCE                                                                                        \
            H4_Q(0, 2, FA, fb0, fb1[3][0] = rdB(S0, 1, 3, 0), fb1[3][1] = rdB(S0, 1, 3, 1), FN[0][0] = rdA(S1, 0, 0),              \
                 w4_piece<0>(rsX, offA[0], ko3), H4_NONE)                                                   \
            H4_Q(0, 3, FA, fb0, FN[0][1] = rdA(S1, 0, 1), FN[1][0] = rdA(S1, 1, 0), FN[1][1] = rdA(S1, 1, 1),                      \
                 w4_piece<1>(rsX, offA[1], ko3), H4_NONE)                                                   \
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");                                               \
            __builtin_amdgcn_s_waitcnt(0xC07F);      /* lgkmcnt(0), as a builtin: hipcc then knows the fragments are in */ \
            H4_FENCE                                                                                        \
            /* ---- phase 2: n half 1 ---- */                                                               \
            __builtin_amdgcn_s_barrier();                                                                   \
            H4_FENCE                                                                                        \
            H4_Q(1, 0, FA, fb1, FN[2][0] = rdA(S1, 2, 0), FN[2][1] = rdA(S1, 2, 1), FN[3][0] = rdA(S1, 3, 0),                      \
                 w4_piece<2>(rsX, offA[2], ko3), H4_NONE)                                                   \
            H4_Q(1, 1, FA, fb1, FN[3][1] = rdA(S1, 3, 1), fb0[0][0] = rdB(S1, 0, 0, 0), fb0[0][1] = rdB(S1, 0, 0, 1),              \
                 w4_piece<3>(rsX, offA[3], ko3), H4_NONE)                                                   \
            w4_set_m0(m0of(S0, 1));                                                                         \
            H4_FENCE                                                                                        \
            H4_Q(1, 2, FA, fb1, fb0[1][0] = rdB(S1, 0, 1, 0), fb0[1][1] = rdB(S1, 0, 1, 1), fb0[2][0] = rdB(S1, 0, 2, 0),          \
                 H4_WP(0, offB[0][0], ko3w), H4_WP(1, offB[0][1], ko3w))                      \
            H4_Q(1, 3, FA, fb1, fb0[2][1] = rdB(S1, 0, 2, 1), fb0[3][0] = rdB(S1, 0, 3, 0), fb0[3][1] = rdB(S1, 0, 3, 1),          \
                 H4_WP(2, offB[0][2], ko3w), H4_WP(3, offB[0][3], ko3w))                      \
            asm volatile("s_waitcnt vmcnt(17)" ::: "memory");                                               \
            __builtin_amdgcn_s_waitcnt(0xC07F);                                                             \
            H4_FENCE                                                                                        \
        }

    stamp(1);
    // six tiles per round of the loop: the ring slots and the two A fragment sets are then compile-time names
    int t = 0;
    for (; t + 5 < nk; t += 6) {
        H4_TILE(t, 0, 1, 2, fa, fa2)
        H4_TILE(t + 1, 1, 2, 0, fa2, fa)
        H4_TILE(t + 2, 2, 0, 1, fa, fa2)
        H4_TILE(t + 3, 0, 1, 2, fa2, fa)
        H4_TILE(t + 4, 1, 2, 0, fa, fa2)
        H4_TILE(t + 5, 2, 0, 1, fa2, fa)
    }
    // (the remainder: up to five tiles, same order)
    if (t < nk) { H4_TILE(t, 0, 1, 2, fa, fa2) }
    if (t + 1 < nk) { H4_TILE(t + 1, 1, 2, 0, fa2, fa) }
    if (t + 2 < nk) { H4_TILE(t + 2, 2, 0, 1, fa, fa2) }
    if (t + 3 < nk) { H4_TILE(t + 3, 0, 1, 2, fa2, fa) }
    if (t + 4 < nk) { H4_TILE(t + 4, 1, 2, 0, fa, fa2) }
#undef H4_TILE
#undef H4_Q
#undef H4_NONE
#undef H4_FENCE
#undef H4_WP
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the clamped re-stages past the last tile; the last MFMAs' results
    __builtin_amdgcn_s_barrier();                                       // LDS is free
    stamp(2);

    // (lane-derived addresses of the code below must not be hoisted over the K loop: they hang off a thread id the optimiser cannot see through)
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    float *rs_lds = reinterpret_cast<float *>(lds);                     // [128] row scales
    int *flag_lds = reinterpret_cast<int *>(lds + 1024);

    if (tid_e < H4_BM) rs_lds[tid_e] = row_scale_of(row_scale, rsp, min(m0 + tid_e, T - 1));
    const bool whole = m0 + H4_BM <= T && n0 + H4_BN <= N;

    RopeLane rl{&ro, rs_lds + wr * 64, bias, reinterpret_cast<float *>(lds + 4096) + (tid_e >> 6) * (16 * 132), T, N, m0 + wr * 64, tid_e & 63, 0u, 0u};
    if (ro.on) { rl.pos0 = ro.st->pos; rl.len = ro.st->len; }
    const int colw = n0 + wn * 128;                                     // first column of this wave's 128
    if constexpr (S == 1) {
        __syncthreads();
        if (ro.on) {
            w4_for<4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const float4v lo[4] = {w4_read<16 * i>(), w4_read<16 * i + 4>(), w4_read<16 * i + 8>(), w4_read<16 * i + 12>()};
                const float4v hi[4] = {w4_read<16 * (4 + i)>(), w4_read<16 * (4 + i) + 4>(), w4_read<16 * (4 + i) + 8>(), w4_read<16 * (4 + i) + 12>()};
                if (ro.d == 128) rope_rows128(rl, i, colw, lo, hi);
                else { rope_rows64(rl, i, colw, lo); rope_rows64(rl, i, colw + 64, hi); }
            });
            return;
        }
        w4_for<2>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            EpiCtx ctx;
            epi_ctx_init(ctx, out, bias, rs_lds, T, N, epi, ldc, m0, n0, tn, wr, wn * 2 + h, tid_e, re, 64);
            w4_for<4>([&](auto ic) {
                constexpr int i = decltype(ic)::value, a0 = 16 * (4 * h + i);
                const float4v v[4] = {w4_read<a0>(), w4_read<a0 + 4>(), w4_read<a0 + 8>(), w4_read<a0 + 12>()};
                if (whole) store_rows<0>(ctx, i, v);
                else if (n0 + H4_BN <= N) store_rows<2>(ctx, i, v);
                else store_rows<1>(ctx, i, v);
            });
        });
        return;
    } else {
        // ---- the slices of a tile meet ----
        // A tile is eight BLOCKS (block b = 4 h + i: rows 16 i .. 16 i + 15 of every wave's n half h -- an eighth of the tile; in the
        // lane-major partial layout 4 x 4 KiB contiguous per slice).  Slice kz OWNS blocks kz, kz + S, ...: it publishes the other
        // blocks (write-through stores), counts itself on the tile's `done` word, and once all S slices are counted adds the other
        // slices' copies of its own blocks to the accumulators it kept in registers -- in K order, its own at its place -- and runs
        // their epilogue.  3/4 of the partial bytes cross memory at four slices, every workgroup works on the tile's end, and in
        // the ordinary case the protocol is one atomic add per workgroup.
        // Nobody waits without bound for a workgroup that may not have started: a slice that is early waits `wait_ticks`, then
        // ABANDONS its blocks -- publishes them too and sets their bits in the tile's flag word -- and leaves.  The slice counted
        // last (for which everything is there by definition) CLOSES the word after its own blocks and finishes whatever bits it
        // finds, from memory alone.  A slice whose abandoning fetch_or finds the word closed came too late to be rescued, but then
        // everything is there: it finishes its blocks itself.  So every block is finished exactly once whatever the timing.
        // The words live in two sets used by alternate launches; a launch zeroes the other set, so nothing is reset inside a launch
        // and a late waiter can never see a recycled word.
        constexpr unsigned CLOSED = 0x100u;
        unsigned *done = ws.ctr + ((size_t)(set * H4_MAX_TILES) + li) * 2, *flagw = done + 1;
        {   // the other set, for the next launch on this workspace (words of every tile a launch may have: the grids differ)
            unsigned *other = ws.ctr + (size_t)((set ^ 1) * H4_MAX_TILES) * 2;
            const int nthr = gridDim.x * gridDim.y * 256, me = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid_e;
            for (int w = me; w < H4_MAX_TILES * 2; w += nthr) __hip_atomic_store(other + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // publish: tile (h, i, j) of this wave at float4 index (16 h + 4 i + j) * 256 + tid -- coalesced 16-byte write-through stores
        float4v *slab = reinterpret_cast<float4v *>(ws.part) + ((size_t)li * S + kz) * H4_TILE_F4 + tid_e;
        auto publish = [&](bool own_blocks) {
            w4_for<8>([&](auto c) {
                constexpr int blk = decltype(c)::value;
                if ((blk % S == kz) == own_blocks)
                    w4_for<4>([&](auto jc) {
                        constexpr int e = blk * 4 + decltype(jc)::value;
                        st_sc1_x4(reinterpret_cast<float *>(slab + (size_t)e * 256), w4_read<4 * e>());
                    });
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        publish(false);
        __syncthreads();
        stamp(3);
        if (tid_e == 0) {
            const unsigned o = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int state = o == (unsigned)S - 1 ? 2 : 0;             // 2: counted last; 1: all there in time; 0: gave up waiting
            if (!state) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)S) { state = 1; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)wait_ticks) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            flag_lds[0] = state;
        }
        __syncthreads();
        stamp(4);
        int state = __builtin_amdgcn_readfirstlane(flag_lds[0]);
        unsigned own_bits = 0;
#pragma unroll
        for (int bb = 0; bb < 8; bb++) own_bits |= (bb % S == kz) ? 1u << bb : 0u;
        if (state == 0) {
            publish(true);
            __syncthreads();
            if (tid_e == 0) flag_lds[1] = (int)__hip_atomic_fetch_or(flagw, own_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (!((unsigned)__builtin_amdgcn_readfirstlane(flag_lds[1]) & CLOSED)) return;      // the last slice will find the bits
            state = 1;                                                                           // closed already: all there, nobody comes back
        }
        const float4v *parts = reinterpret_cast<const float4v *>(ws.part) + (size_t)li * S * H4_TILE_F4 + tid_e;
        auto finish = [&](int b, const float4v (&v)[4]) {
            EpiCtx ctx;
            epi_ctx_init(ctx, out, bias, rs_lds, T, N, epi, ldc, m0, n0, tn, wr, wn * 2 + (b >> 2), tid_e, re, 64);
            if (whole) store_rows<0>(ctx, b & 3, v);
            else if (n0 + H4_BN <= N) store_rows<2>(ctx, b & 3, v);
            else store_rows<1>(ctx, b & 3, v);
        };
        // The other slices' copies of the own blocks: all requested at once, as write-through-coherent (sc1) loads -- with sc1 stores
        // on the other side, the storing waves' vmcnt(0) before their count, and this workgroup's barrier behind thread 0's poll,
        // no acquire fence is needed (cdna guide, Guideline 16 table, first row).
        constexpr int NB = (8 + S - 1) / S;
        float4v ld[NB][S - 1][4];
        {
            // (buffer loads with the sc1 policy through the builtin: hipcc counts them like its own loads -- asm loads it does not,
            // and it copied their destination registers before the data was there)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(ws.part) + (size_t)li * S * (H4_BM * H4_BN), 0, S * H4_BM * H4_BN * 4, 0x00020000);
            const int lane_off = tid_e * 16;
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int bq = min(kz + k * S, 7);
#pragma unroll
                for (int p = 0; p < S - 1; p++) {
                    const int base = ((p + (p >= kz ? 1 : 0)) * H4_TILE_F4 + bq * 4 * 256) * 16;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        ld[k][p][j] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, base + j * 4096, 16));
                }
            }
        }
        // own accumulators: the register numbers depend on kz, so one copy of these reads per slice index
        float4v own[NB][4];
        w4_for<S>([&](auto kc) {
            constexpr int KZ = decltype(kc)::value;
            if (kz == KZ)
                w4_for<NB>([&](auto qc) {
                    constexpr int bq = KZ + decltype(qc)::value * S < 8 ? KZ + decltype(qc)::value * S : 7;
                    own[decltype(qc)::value][0] = w4_read<16 * bq>(); own[decltype(qc)::value][1] = w4_read<16 * bq + 4>();
                    own[decltype(qc)::value][2] = w4_read<16 * bq + 8>(); own[decltype(qc)::value][3] = w4_read<16 * bq + 12>();
                });
        });
        stamp(6);
        // (the sums replace the own accumulators in place: one array less -- at three slices the tail sat at the 256-VGPR limit and
        // hipcc parked two registers in a0 / a1, which it believes free: wrong sums.  tests/test_dot_hazard.py audits the build.)
        float4v (&vs)[NB][4] = own;
#pragma unroll
        for (int k = 0; k < NB; k++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // K order: slice q is this workgroup's own (q == kz), or peer q (q < kz) / peer q - 1 (q > kz)
                float4v acc = kz == 0 ? own[k][j] : ld[k][0][j];
#pragma unroll
                for (int q = 1; q < S; q++) {
                    const float4v below = ld[k][q - 1][j];
                    acc += q == kz ? own[k][j] : (q < kz ? ld[k][q < S - 1 ? q : q - 1][j] : below);
                }
                vs[k][j] = acc;
            }
        }
        if (ro.on) {
            // (host: 2 or 4 slices here, so that blocks b and b + 4 -- the two halves of a 128-wide head -- have one owner:
            // own block k < NB / 2 is (n half 0, row block kz + k S), own block k + NB / 2 its partner in n half 1)
            if constexpr (NB % 2 == 0) {
#pragma unroll
                for (int k = 0; k < NB / 2; k++) {
                    const int i = kz + k * S;                                                    // < 4
                    if (ro.d == 128) rope_rows128(rl, i, colw, vs[k], vs[k + NB / 2]);
                    else { rope_rows64(rl, i, colw, vs[k]); rope_rows64(rl, i, colw + 64, vs[k + NB / 2]); }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int bq = kz + k * S;
                if (bq < 8) finish(bq, vs[k]);
            }
        }
        stamp(7);
        if (state != 2) return;
        // counted last: close the tile's word; whatever bits are there belong to slices that gave up -- their blocks are in memory
        __syncthreads();
        if (tid_e == 0) flag_lds[1] = (int)(0xFFu & __hip_atomic_fetch_or(flagw, CLOSED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __syncthreads();
        const unsigned rest = (unsigned)__builtin_amdgcn_readfirstlane(flag_lds[1]);
        if (!rest) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                 // (rare path: plain loads behind an acquire)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        auto from_memory = [&](int bq, float4v (&v)[4]) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[j] = parts[(size_t)(bq * 4 + j) * 256];
#pragma unroll
                for (int sl = 1; sl < S; sl++) v[j] += parts[(size_t)sl * H4_TILE_F4 + (size_t)(bq * 4 + j) * 256];
            }
        };
        for (int bq = 0; bq < (ro.on ? 4 : 8); bq++) {
            if (!((rest >> bq) & 1)) continue;
            float4v v[4];
            from_memory(bq, v);
            if (!ro.on) { finish(bq, v); continue; }
            float4v v2[4];                                                  // (an owner abandons both halves of its heads: bit bq + 4 is set too)
            from_memory(bq + 4, v2);
            if (ro.d == 128) rope_rows128(rl, bq, colw, v, v2);
            else { rope_rows64(rl, bq, colw, v); rope_rows64(rl, bq, colw + 64, v2); }
        }
    }
}

// ---- workspace of one stream: partial tiles of the sliced launches + the per-tile ticket / done words ------------------------
struct H4Ws { float *part = nullptr; size_t part_tiles = 0; unsigned *ctr = nullptr; unsigned launches = 0; };
static std::mutex g_h4_mu;
static std::map<std::pair<int, hipStream_t>, H4Ws> g_h4_spaces;

int64_t gemm_h4_workspace_bytes(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(g_h4_mu);
    auto it = g_h4_spaces.find({dev, stream});
    return it == g_h4_spaces.end() ? 0 : (int64_t)it->second.part_tiles * H4_BM * H4_BN * 4;
}
void gemm_h4_release_stream(hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_h4_mu);
    for (auto it = g_h4_spaces.begin(); it != g_h4_spaces.end();) {
        if (it->first.second == stream) {
            (void)hipFree(it->second.part); (void)hipFree(it->second.ctr);
            it = g_h4_spaces.erase(it);
        } else {
            ++it;
        }
    }
}
// *set: which of the two word sets this launch uses (the launch zeroes the other one for the next)
static int h4_space(hipStream_t stream, size_t tiles, H4Space *out, int *set) {
    int dev = 0;
    FL_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_h4_mu);
    H4Ws &sp = g_h4_spaces[{dev, stream}];
    if (!sp.ctr) {
        FL_HIP(hipMalloc((void **)&sp.ctr, (size_t)H4_MAX_TILES * 2 * 2 * sizeof(unsigned)));
        FL_HIP(hipMemsetAsync(sp.ctr, 0, (size_t)H4_MAX_TILES * 2 * 2 * sizeof(unsigned), stream));
    }
    if (sp.part_tiles < tiles) {
        tiles = std::max<size_t>(tiles, 320);                      // 40 MiB: a chip's worth of sliced tiles; grows only for forced shapes
        if (sp.part) { FL_HIP(hipStreamSynchronize(stream)); (void)hipFree(sp.part); sp.part = nullptr; sp.part_tiles = 0; }
        FL_HIP(hipMalloc((void **)&sp.part, tiles * H4_BM * H4_BN * sizeof(float)));
        sp.part_tiles = tiles;
    }
    *set = (int)(sp.launches & 1);                                 // (advanced by h4_launched once the launch is in the stream)
    *out = H4Space{sp.part, sp.ctr};
    return FL_OK;
}
// the launch that took word set `launches & 1` is in the stream: the next sliced launch takes the other set.  A launch that failed
// never ran, so it zeroed nothing: the set it would have used is still clean and is handed out again.
static void h4_launched(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_h4_mu);
    auto it = g_h4_spaces.find({dev, stream});
    if (it != g_h4_spaces.end()) it->second.launches++;
}

bool gemm_h4_supported(int64_t T, int64_t N, int64_t K, int ksplit) {
    if (T < 1 || N < 1 || K % P_BK || ksplit < 1 || ksplit > H4_MAXS || K / P_BK < ksplit) return false;
    const int64_t tiles = ((T + H4_BM - 1) / H4_BM) * ((N + H4_BN - 1) / H4_BN);
    if (tiles > H4_MAX_TILES || tiles * ksplit > 65535 * 8) return false;
    // a lane's bytes are addressed as a 32-bit offset from the matrix base (+ the K offset and the 3072-byte bias of the lane offsets)
    return (double)std::max(T, N) * (double)K * 2.0 + (double)K * 2.0 + 8192.0 < 4294967296.0;
}

// Which shapes run here, and in how many K slices (0: not here).  TK_GEMM_H4 = 2 / TK_H4_SPLIT pin the choice (tests, probes).
int gemm_h4_plan(int64_t T, int64_t N, int64_t K, int epi) {
    const int mode = tune(TK_GEMM_H4);
    if (mode <= 0 || T <= 16) return 0;           // (T <= 16: decode batches, whose steps are captured graphs: no host-side launch state)
    const int forced = tune(TK_H4_SPLIT);
    const int64_t tiles = ((T + H4_BM - 1) / H4_BM) * ((N + H4_BN - 1) / H4_BN), nk = K / P_BK;
    int ks = forced > 0 ? forced : 1;
    if (forced <= 0) {
        // slices while the grid still fits one round of the chip and a slice keeps >= 8 K steps
        while (ks < H4_MAXS && tiles * (ks + 1) <= 256 && nk / (ks + 1) >= 8) ks++;
    }
    ks = (int)std::min<int64_t>(ks, std::max<int64_t>(1, nk));
    if (epi == EPI_QKV_ROPE && ks == 3) ks = 2;                     // (the two halves of a 128-wide head need one owner: 1, 2 or 4 slices)
    if (!gemm_h4_supported(T, N, K, ks)) return 0;
    if (mode >= 2) return ks;
    // mode 1: the row-parallel projections (residual epilogue: no slabs, no rmsnorm_add launch) of prompts of 257-640 tokens, where
    // the 256 x 256 grid is 32-48 tiles and needs eight K slices.  A/B inside one process, whole Mistral-7B prefills back to back
    // (tools/tune_ab.py, ms without / with): T = 256 6.86 / 7.22, 384 9.00 / 8.55, 512 9.22 / 8.89, 768 12.52 / 13.36, 1024 14.98 / 14.94.
    const int64_t t8 = ((T + 255) / 256) * ((N + 255) / 256);
    // (Qwen2-7B 384 / 512 / 640: 0.987 / 1.000 / 0.958; TinyLlama-1.1B, N = K = 2048: 1.025 / 0.995 / 0.986 -- left where it was)
    if ((epi == EPI_RESID || epi == EPI_QKV_ROPE) && T > 256 && T <= 640 && t8 < 128 && tiles * ks <= 256 && ks >= 2 && N >= 3072 && K >= 3072) return ks;
    // 641-1024 tokens: o_proj alone (short K: <= 32 steps per slice; down_proj's 112-step slices and the unsliced QKV lose to 256 x 256),
    // and only where 256-row tiles would waste >= 64 rows (whole Mistral-7B prefills, with / without: 700 tokens 0.981, 896 0.959;
    // 768 and 1024 -- multiples of 256 -- 1.002 / 1.010)
    if (epi == EPI_RESID && T > 640 && T <= 1024 && tune(TK_H4_OPROJ_1K) && ((T + 255) / 256) * 256 - T >= 64 && t8 < 128 && ks >= 2 && tiles * ks >= 192 && tiles * ks <= 256 && nk / ks <= 32 &&
        N >= 3072 && K >= 3072) return ks;
    // 129-256 tokens (one row tile of 256 x 256, two of this kernel): per projection, by how much of the chip the grid fills.  Mistral-7B
    // at 256 tokens, us per launch, old path / this kernel: gate/up (224 tiles, no slices) 83.4 / 76.7; QKV + RoPE 29.8 + 6.5 / 32.2;
    // o_proj + rmsnorm_add 24.9 + 6.8 / 28.0; down_proj 46.2 + 6.8 / 67.1 (128 workgroups with 56 K steps each: stays where it was).
    // ... and, at 257-640 tokens, the same per-projection rule for matrices below 3072 in N or K (TinyLlama-1.1B at 512 tokens, old path /
    // this kernel: gate/up 44.7 / 37.7 us; QKV + RoPE 25.9 / 21.1; o_proj + rmsnorm_add 25.6 / 21.1; down_proj 33.1 / 34.3; whole prefill
    // 3.28 -> 2.92 ms; at 384 tokens every grid is below the fill bounds and nothing changes)
    if (T > 128 && T <= 640 && (T <= 256 || N < 3072 || K < 3072) && t8 < 128 && tiles * ks <= 256) {
        const int64_t fill = tiles * ks;
        if (epi == EPI_GATEUP && ks == 1 && fill >= 160) return 1;
        if (epi == EPI_QKV_ROPE && ks >= 2 && fill >= 128) return ks;
        if (epi == EPI_RESID && ks >= 2 && fill >= 128 && nk / ks <= 24) return ks;
    }
    return 0;
}

// A projection whose caller needs ONE complete output and cannot hand K slabs to the next launch: a tensor-parallel rank's row-parallel
// projections (the all-reduce wants the sum) and its shard of gate/up.  Their grids are a fraction of the single-GPU ones -- Mistral-7B
// at tp = 2, 512 tokens: o_proj 4096 x 2048 and down_proj 4096 x 7168 are 128 tiles of 128 x 128 with 32 / 112 K steps, which that
// kernel ran unsliced at 220-280 TFLOP/s (profiles/r05/tp_prefill_before.txt: a rank's 512-token prefill took LONGER at tp = 2 than the
// whole model on one GPU) -- so K slices that meet inside the launch are what fills the chip.  0: not here.
int gemm_h4_plan_whole(int64_t T, int64_t N, int64_t K, int epi) {
    if (tune(TK_GEMM_H4) <= 0 || T <= 128 || T > 1024 || (epi != EPI_F32 && epi != EPI_GATEUP && epi != EPI_QKV_ROPE)) return 0;
    const int64_t tiles = ((T + H4_BM - 1) / H4_BM) * ((N + H4_BN - 1) / H4_BN), nk = K / P_BK;
    const int64_t t8 = ((T + 255) / 256) * ((N + 255) / 256);
    if (t8 >= 128 || tiles > 256) return 0;                          // (half a round of 256 x 256 tiles and more: the large kernels' shapes)
    int ks = 1;
    while (ks < H4_MAXS && tiles * (ks + 1) <= 256 && nk / (ks + 1) >= 8) ks++;
    if (epi == EPI_QKV_ROPE) {                                       // a rank's q | k | v rows: RoPE, bias and the KV append in the epilogue (1, 2 or 4 slices)
        if (ks == 3) ks = 2;
        // (tp = 4: 1536 rows = 24 tiles x 4 slices; against 128 x 128 tiles in K slabs + the rope_kv launch that sums them)
        return gemm_h4_supported(T, N, K, ks) && tiles * ks >= 96 ? ks : 0;       // (tp = 8: 12 tiles x 4 = a tie, stays)
    }
    if (!gemm_h4_supported(T, N, K, ks) || tiles * ks < 96) return 0;
    return ks;
}

// The tail columns of a column-peeled GEMM (k_gemm_mfma.hip: whole rounds of 256 x 256 tiles + at most half a round more): instead of the
// stream-K launch and its fix-up launch, ONE launch of this kernel when the tail's 128 x 256 tiles times 2-4 K slices fill most of the
// chip -- any epilogue, the slices summed inside.  (Qwen2-7B: the 5120 gate/up columns past 32768 at 512 tokens 31.3 + 13.2 us as
// stream-K + fix-up; the 512 QKV / 1024 gate/up columns at 4096 tokens 27 + 14 / 36 + 14 us.)
int gemm_h4_tail_slices(int64_t T, int64_t N, int64_t K) {
    if (tune(TK_GEMM_H4) <= 0 || !tune(TK_H4_TAIL) || T <= 16 || K % P_BK) return 0;
    const int64_t tiles = ((T + H4_BM - 1) / H4_BM) * ((N + H4_BN - 1) / H4_BN), nk = K / P_BK;
    int ks = 1;
    while (ks < H4_MAXS && tiles * (ks + 1) <= 256 && nk / (ks + 1) >= 8) ks++;
    // (h4_tail = 2: also unsliced when the tail's tiles alone fill the chip -- Mistral-7B's 2048 QKV tail columns at 4096 tokens, 256 tiles)
    if ((ks < 2 && !(tune(TK_H4_TAIL) >= 2 && tiles >= 224)) || tiles * ks < 160 || !gemm_h4_supported(T, N, K, ks)) return 0;
    return ks;
}

int launch_gemm_h4(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                   int epi, const float *row_scale, int ksplit, int64_t ldc, const ResidEpi *resid, const RopeEpi *rope) {
    if (ldc <= 0) ldc = N;
    if (!gemm_h4_supported(T, N, K, ksplit)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_h4: shape / K slices not supported");
    if ((epi == EPI_RESID) != (resid != nullptr) || (resid && (!resid->h || !resid->w || !resid->xn || !resid->part)))
        FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_h4: the residual epilogue takes its operands");
    if ((epi == EPI_QKV_ROPE) != (rope != nullptr) ||
        (rope && (ksplit == 3 || (rope->d != 64 && rope->d != 128) || N % rope->d || rope->col_base % 128 || rope->col_base + N > (int64_t)(rope->H + 2 * rope->Hkv) * rope->d || !rope->st || !rope->cos_tab ||
                  !rope->sin_tab || !rope->q_out || !rope->k_cache || !rope->v_cache)))
        FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_h4: the RoPE epilogue takes its operands, head_dim 64 / 128, and 1, 2 or 4 K slices");
    if (epi == EPI_GATEUP && bias) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_h4: gate/up takes no bias");
    const ResidEpi re = resid ? *resid : ResidEpi{};
    RopeEpi ro = rope ? *rope : RopeEpi{};
    ro.on = rope ? 1 : 0;
    const int tiles_m = (int)((T + H4_BM - 1) / H4_BM), tiles_n = (int)((N + H4_BN - 1) / H4_BN);
    H4Space ws{nullptr, nullptr};
    int set = 0;
    if (ksplit > 1) FL_TRY(h4_space(L.stream, (size_t)tiles_m * tiles_n * ksplit, &ws, &set));
    // W pieces non-temporal (h4_nt: 1 always, 0 never, -1 at 176-256 tokens: two nearly full row tiles per panel): an HBM stream through LDS-DMA
    // is 15 % faster with nt (tools/micro/ingest_bench.hip) and does not push the activations out of the caches, but the second
    // reader of a panel then misses more often.  Whole prefills, plain / nt: Mistral-7B 200 / 256 tokens 6.40 / 6.24, 6.58 / 6.42 ms;
    // 300 / 384 / 512 / 640 tokens (3-5 row tiles) x 1.013 / 1.007 / 1.009 / 1.050; Qwen2-7B 384-640 x 1.04-1.06
    const int nt_mode = tune(TK_H4_NT);
    const bool wnt = nt_mode == 1 || (nt_mode < 0 && T >= 176 && T <= 256);   // (Mistral-7B: 130-150 tokens x 1.02-1.03 with nt, 160 a tie, 176-256 x 0.98-0.96; Qwen2-7B +-1 %)
    auto kern = wnt ? (ksplit == 1 ? gemm_h4_kernel<1, true> : ksplit == 2 ? gemm_h4_kernel<2, true> : ksplit == 3 ? gemm_h4_kernel<3, true> : gemm_h4_kernel<4, true>)
                    : (ksplit == 1 ? gemm_h4_kernel<1, false> : ksplit == 2 ? gemm_h4_kernel<2, false> : ksplit == 3 ? gemm_h4_kernel<3, false> : gemm_h4_kernel<4, false>);
    static_assert(H4_MAXS == 4, "one instantiation per slice count");
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), H4_LDS));
    char tag[32];
    snprintf(tag, sizeof tag, "h4,%lldx%lld%s%s", (long long)N, (long long)K, ksplit > 1 ? ",sliced" : "", resid ? ",resid" : rope ? ",rope" : "");
    Launcher LL = L; LL.tag = tag;
    // groups of four row tiles x columns: the eight tiles an XCD works on per K slice at T = 512 are 4 x 2 (four X panels, two W panels)
    const int group_m = std::max(1, std::min(tiles_m, 4));
    const double bytes = ((double)N * K + (double)T * K) * 2.0;
    const char *stamp_path = env_str("FL_H4_STAMPS");        // diagnostics only: synchronous, appends one record per launch
    unsigned long long *d_st = nullptr;
    const size_t nwg = (size_t)tiles_m * tiles_n * ksplit;
    if (stamp_path) { FL_HIP(hipMalloc((void **)&d_st, nwg * 64)); FL_HIP(hipMemsetAsync(d_st, 0, nwg * 64, L.stream)); }
    const int rc = LL.launch(KC_GEMM_MFMA, bytes, 2.0 * T * N * K, kern, dim3((unsigned)(tiles_m * tiles_n), (unsigned)ksplit), dim3(256), H4_LDS,
                     (const bf16_t *)W, (const bf16_t *)x, bias, y, (int)T, (int)N, (int)K, epi, tiles_m, tiles_n, row_scale, (int)ldc,
                     ws, re, ro, group_m, tune(TK_H4_PF), set, tune(TK_H4_WAIT_US) * 100, d_st, L.rsp);
    if (rc == FL_OK && ksplit > 1) h4_launched(L.stream);
    if (stamp_path) {
        std::vector<unsigned long long> h(nwg * 8);
        FL_HIP(hipStreamSynchronize(L.stream));
        FL_HIP(hipMemcpy(h.data(), d_st, nwg * 64, hipMemcpyDeviceToHost));
        (void)hipFree(d_st);
        if (FILE *f = fopen(stamp_path, "a")) {
            fprintf(f, "launch %lld %lld %lld %d %d %zu\n", (long long)T, (long long)N, (long long)K, epi, ksplit, nwg);
            for (size_t i = 0; i < nwg; i++) {
                for (int j = 0; j < 8; j++) fprintf(f, "%llu ", h[i * 8 + j]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    return rc;
}

}  // namespace fl
