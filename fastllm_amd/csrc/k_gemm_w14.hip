// k_gemm_w14.hip -- prefill projection GEMM, 256 x 224 tile on four waves: the four-wave 256 x 256 kernel of k_gemm_8p.hip with
// fourteen column blocks instead of sixteen, for matrices whose width is whole 224-column tiles but whose 256-column grid leaves CUs idle.
//
//   Y[T,N] = X[T,K] . W[N,K]^T      bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
//
// Why: Mistral-7B / Llama-3-8B's fused gate/up matrix is 28672 rows = 112 tiles of 256 = 128 tiles of 224.  At 257-512 tokens (two row
// tiles) the 256-wide grid is 224 workgroups on 256 CUs -- an eighth of the chip idle for the largest projection of the prefill (38 % of
// its time); 224-wide it is exactly 256, each with 7/8 of the work.  The same holds wherever both grids take the same number of rounds
// (768-1024 tokens: 448 / 512 tiles, two rounds either way; 2048 tokens: 896 / 1024, four).
//
// Tile and K loop: waves as 4 (M) x 1 (N), 64 rows x 224 columns each: 4 x 14 accumulator tiles = a[0:223], owned by inline asm as in
// gemm_4w_kernel.  A K tile (BK = 64) is four half tiles in LDS -- A0 / A1 (rows 0-127 / 128-255 of X), B0 (columns 0-127 of the tile:
// 16 KiB), B1 (columns 128-223: 12 KiB) -- in two slots by K-tile parity (2 x 64 KiB), and is consumed in FOUR phases, one per column
// chunk c0..c3 = 4, 4, 3, 3 column blocks (32, 32, 24, 24 MFMAs):
//     tile t:  phase   computes     requests (fragments)                         re-stages by LDS-DMA (into tile t's slot)
//              1       fa x c0      c1 <- B0(t)                       (8)        A0(t+2) x 4 pieces per wave
//              2       fa x c1      c2 <- B1(t); fa' row block 0      (6 + 2)    A1(t+2) x 4
//              3       fa x c2      c3 <- B1(t); fa' row blocks 1, 2  (6 + 4)    B0(t+2) x 4
//              4       fa x c3      c0 <- B0(t+1); fa' row block 3    (8 + 2)    B1(t+2) x 3
// (fa' = the A fragments of tile t + 1; fa / fa' alternate by tile parity, the two B fragment sets by phase: 128 VGPRs.)  One s_barrier
// per phase; before it every wave has waited for its fragment reads (lgkmcnt(0)) and for all but its 11 / 15 / 15 / 18 youngest DMA
// pieces: a half tile is read four to six phases after it was requested.  The request order is the same from the prologue on (tile 0's
// four halves, tile 1's), so the counts hold from the first tile; past the last K tile the requests go on, clamped to the last tile.
// 36 fragment reads and 15 DMA pieces per 112 MFMAs (the 256 x 256 tile: 32 + 16 per 128): every wave reads the whole B half tiles.
#include <stdlib.h>

#include <algorithm>

#include "attn_common.h"
#include "gemm_w4.h"
#include "kernels.h"

namespace fl {

constexpr int W14_BM = 256, W14_BN = 224;
constexpr int W14_SLOT = 4 * P_HALF;                   // 64 KiB: A0 | A1 | B0 | B1 (B1 uses 12 of its 16 KiB)
constexpr int W14_LDS = 2 * W14_SLOT;                  // 128 KiB

#define W14_AGPRS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223"

template <bool WNT>
__global__ __launch_bounds__(256) void gemm_w14_kernel(const bf16_t *__restrict__ W, const bf16_t *__restrict__ X,
                                                       const float *__restrict__ bias, void *__restrict__ out,
                                                       int T, int N, int K, int epi, int tiles_m, int tiles_n,
                                                       const float *__restrict__ row_scale, int ldc, int group_m, RsParts rsp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [parity][A0 A1 B0 B1]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, kg = lane >> 4;
    asm volatile("" : : : W14_AGPRS);                                   // the kernel descriptor allocates a[0:223]

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    const int li = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
    const int nk = K / P_BK;
    const int per_g = group_m * tiles_n, g0 = (li / per_g) * group_m, gm = min(tiles_m - g0, group_m), lr = li % per_g;
    const int tm = g0 + lr % gm, tn = lr / gm;
    const int m0 = tm * W14_BM, n0 = tn * W14_BN;

    // byte offsets of this lane's 16 bytes in each of its 1-KiB pieces (k = 0): A0 / A1 / B0: pieces 4 w + s; B1 (12 pieces): 3 w + s
    unsigned offA[2][4], offB0[4], offB1[3];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int r = (wave * 4 + s) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; h++) offA[h][s] = (unsigned)(((size_t)min(m0 + h * 128 + r, T - 1) * K + c * 8) * 2) + (W4_BIAS - 1024 * s);
        offB0[s] = (unsigned)(((size_t)min(n0 + r, N - 1) * K + c * 8) * 2) + (W4_BIAS - 1024 * s);
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int r = (wave * 3 + s) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        offB1[s] = (unsigned)(((size_t)min(n0 + 128 + r, N - 1) * K + c * 8) * 2) + (W4_BIAS - 1024 * s);
    }
    w4_for<14>([&](auto c) { w4_zero16<decltype(c)::value * 16>(); });

    const int4w rsX = w4_rsrc(X), rsW = w4_rsrc(W);
#define W14_WP(S_, OFF_) w4_piece<S_, WNT>(rsW, OFF_, ko)      /* a W piece: non-temporal where no other workgroup re-reads the panel */
    auto koff = [&](int tile) { return (unsigned)(min(tile, nk - 1) * (P_BK * 2)); };   // byte offset of a K tile (clamped past the end)
    const unsigned ldsw4 = (unsigned)(size_t)lds + wave * 4096, ldsw3 = (unsigned)(size_t)lds + wave * 3072;
    // which: 0 A0, 1 A1, 2 B0, 3 B1
    auto m0of = [&](int slot, int which) { return (which == 3 ? ldsw3 : ldsw4) + slot * W14_SLOT + which * P_HALF; };
    auto stage_half = [&](int which, int slot, int tile) {              // every piece of this wave of one half tile (prologue)
        w4_set_m0(m0of(slot, which));
        const unsigned ko = koff(tile);
        if (which < 2) { w4_piece<0>(rsX, offA[which][0], ko); w4_piece<1>(rsX, offA[which][1], ko); w4_piece<2>(rsX, offA[which][2], ko); w4_piece<3>(rsX, offA[which][3], ko); }
        else if (which == 2) { W14_WP(0, offB0[0]); W14_WP(1, offB0[1]); W14_WP(2, offB0[2]); W14_WP(3, offB0[3]); }
        else { W14_WP(0, offB1[0]); W14_WP(1, offB1[1]); W14_WP(2, offB1[2]); }
    };
    // prologue: tiles 0 and 1 whole, in the order of the loop's requests (15 pieces per wave and tile)
    stage_half(0, 0, 0); stage_half(1, 0, 0); stage_half(2, 0, 0); stage_half(3, 0, 0);
    stage_half(0, 1, 1); stage_half(1, 1, 1); stage_half(2, 1, 1); stage_half(3, 1, 1);
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");                   // A0(0), A1(0), B0(0) of this wave have landed
    __builtin_amdgcn_s_barrier();

    // fragment addresses: lane part (row, swizzled 16-byte chunk for ks = 0 / 1), then slot / half / block as offsets
    const unsigned sw = (unsigned)((m16 >> 1) & 7);
    const unsigned la0 = ((wave & 1) * 64 + m16) * 128 + (((unsigned)kg ^ sw) << 4), la1 = ((wave & 1) * 64 + m16) * 128 + (((unsigned)(4 + kg) ^ sw) << 4);
    const unsigned lb0 = m16 * 128 + (((unsigned)kg ^ sw) << 4), lb1 = m16 * 128 + (((unsigned)(4 + kg) ^ sw) << 4);
    const unsigned ahalf = (wave >> 1) * P_HALF;
    auto rdA = [&](int slot, int q, int ks) { return *reinterpret_cast<const int4w *>(lds + slot * W14_SLOT + ahalf + (ks ? la1 : la0) + q * 2048); };
    // column block cb of the tile (0 .. 13): B0 holds 0 .. 7, B1 8 .. 13
    auto rdB = [&](int slot, int cb, int ks) { return *reinterpret_cast<const int4w *>(lds + slot * W14_SLOT + (cb < 8 ? 2 : 3) * P_HALF + (ks ? lb1 : lb0) + (cb < 8 ? cb : cb - 8) * 2048); };

    int4w fa[4][2], fa2[4][2], fbx[4][2], fby[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            fa[i][ks] = rdA(0, i, ks);
            fbx[i][ks] = rdB(0, i, ks);
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);                                 // lgkmcnt(0)

#define W14_FENCE __builtin_amdgcn_sched_barrier(0);
#define W14_NONE (void)0
    // row block Q against four column blocks CB0 .. CB0 + 3 (8 MFMAs), fillers F0 .. F3 between them
#define W14_Q4(Q, CB0, FA, FB, F0, F1, F2, F3)                                                              \
        {                                                                                                   \
            constexpr int a0 = 4 * ((Q) * 14 + (CB0));                                                      \
            w4_mfma<a0>(FA[Q][0], FB[0][0]); W14_FENCE                                                      \
            F0; W14_FENCE                                                                                   \
            w4_mfma<a0>(FA[Q][1], FB[0][1]); W14_FENCE                                                      \
            w4_mfma<a0 + 4>(FA[Q][0], FB[1][0]); W14_FENCE                                                  \
            F1; W14_FENCE                                                                                   \
            w4_mfma<a0 + 4>(FA[Q][1], FB[1][1]); W14_FENCE                                                  \
            w4_mfma<a0 + 8>(FA[Q][0], FB[2][0]); W14_FENCE                                                  \
            F2; W14_FENCE                                                                                   \
            w4_mfma<a0 + 8>(FA[Q][1], FB[2][1]); W14_FENCE                                                  \
            w4_mfma<a0 + 12>(FA[Q][0], FB[3][0]); W14_FENCE                                                 \
            F3; W14_FENCE                                                                                   \
            w4_mfma<a0 + 12>(FA[Q][1], FB[3][1]); W14_FENCE                                                 \
        }
    // ... against three column blocks (6 MFMAs)
#define W14_Q3(Q, CB0, FA, FB, F0, F1, F2, F3)                                                              \
        {                                                                                                   \
            constexpr int a0 = 4 * ((Q) * 14 + (CB0));                                                      \
            w4_mfma<a0>(FA[Q][0], FB[0][0]); W14_FENCE                                                      \
            F0; W14_FENCE                                                                                   \
            w4_mfma<a0>(FA[Q][1], FB[0][1]); W14_FENCE                                                      \
            w4_mfma<a0 + 4>(FA[Q][0], FB[1][0]); W14_FENCE                                                  \
            F1; W14_FENCE                                                                                   \
            w4_mfma<a0 + 4>(FA[Q][1], FB[1][1]); W14_FENCE                                                  \
            w4_mfma<a0 + 8>(FA[Q][0], FB[2][0]); W14_FENCE                                                  \
            F2; W14_FENCE                                                                                   \
            w4_mfma<a0 + 8>(FA[Q][1], FB[2][1]); W14_FENCE                                                  \
            F3; W14_FENCE                                                                                   \
        }
#define W14_END(CNT)                                                                                        \
        asm volatile("s_waitcnt vmcnt(" #CNT ")" ::: "memory");                                             \
        __builtin_amdgcn_s_waitcnt(0xC07F);      /* lgkmcnt(0), as a builtin: hipcc then knows the fragments are in */ \
        W14_FENCE
    // one K tile: S0 = its slot (tile t + 2 goes there), S1 = the slot of tile t + 1; FA current, FN next A fragment set
#define W14_TILE(TT, S0, S1, FA, FN)                                                                        \
        {                                                                                                   \
            const int tt = (TT);                                                                            \
            const unsigned ko = koff(tt + 2);                                                               \
            /* ---- phase 1: columns 0-63 ---- */                                                           \
            __builtin_amdgcn_s_barrier();                                                                   \
            W14_FENCE                                                                                       \
            w4_set_m0(m0of(S0, 0));                                                                         \
            W14_FENCE                                                                                       \
            W14_Q4(0, 0, FA, fbx, fby[0][0] = rdB(S0, 4, 0), fby[0][1] = rdB(S0, 4, 1), w4_piece<0>(rsX, offA[0][0], ko), fby[1][0] = rdB(S0, 5, 0)) \
            W14_Q4(1, 0, FA, fbx, fby[1][1] = rdB(S0, 5, 1), fby[2][0] = rdB(S0, 6, 0), w4_piece<1>(rsX, offA[0][1], ko), W14_NONE)    \
            W14_Q4(2, 0, FA, fbx, fby[2][1] = rdB(S0, 6, 1), fby[3][0] = rdB(S0, 7, 0), w4_piece<2>(rsX, offA[0][2], ko), W14_NONE)    \
            W14_Q4(3, 0, FA, fbx, fby[3][1] = rdB(S0, 7, 1), W14_NONE, w4_piece<3>(rsX, offA[0][3], ko), W14_NONE)                     \
            W14_END(11)                                                                                     \
            /* ---- phase 2: columns 64-127 ---- */                                                         \
            __builtin_amdgcn_s_barrier();                                                                   \
            W14_FENCE                                                                                       \
            w4_set_m0(m0of(S0, 1));                                                                         \
            W14_FENCE                                                                                       \
            W14_Q4(0, 4, FA, fby, fbx[0][0] = rdB(S0, 8, 0), fbx[0][1] = rdB(S0, 8, 1), w4_piece<0>(rsX, offA[1][0], ko), fbx[1][0] = rdB(S0, 9, 0)) \
            W14_Q4(1, 4, FA, fby, fbx[1][1] = rdB(S0, 9, 1), fbx[2][0] = rdB(S0, 10, 0), w4_piece<1>(rsX, offA[1][1], ko), W14_NONE)   \
            W14_Q4(2, 4, FA, fby, fbx[2][1] = rdB(S0, 10, 1), FN[0][0] = rdA(S1, 0, 0), w4_piece<2>(rsX, offA[1][2], ko), W14_NONE)    \
            W14_Q4(3, 4, FA, fby, FN[0][1] = rdA(S1, 0, 1), W14_NONE, w4_piece<3>(rsX, offA[1][3], ko), W14_NONE)                      \
            W14_END(15)                                                                                     \
            /* ---- phase 3: columns 128-175 ---- */                                                        \
            __builtin_amdgcn_s_barrier();                                                                   \
            W14_FENCE                                                                                       \
            w4_set_m0(m0of(S0, 2));                                                                         \
            W14_FENCE                                                                                       \
            W14_Q3(0, 8, FA, fbx, fby[0][0] = rdB(S0, 11, 0), fby[0][1] = rdB(S0, 11, 1), W14_WP(0, offB0[0]), fby[1][0] = rdB(S0, 12, 0)) \
            W14_Q3(1, 8, FA, fbx, fby[1][1] = rdB(S0, 12, 1), fby[2][0] = rdB(S0, 13, 0), W14_WP(1, offB0[1]), fby[2][1] = rdB(S0, 13, 1)) \
            W14_Q3(2, 8, FA, fbx, FN[1][0] = rdA(S1, 1, 0), FN[1][1] = rdA(S1, 1, 1), W14_WP(2, offB0[2]), W14_NONE)        \
            W14_Q3(3, 8, FA, fbx, FN[2][0] = rdA(S1, 2, 0), FN[2][1] = rdA(S1, 2, 1), W14_WP(3, offB0[3]), W14_NONE)        \
            W14_END(15)                                                                                     \
            /* ---- phase 4: columns 176-223 ---- */                                                        \
            __builtin_amdgcn_s_barrier();                                                                   \
            W14_FENCE                                                                                       \
            w4_set_m0(m0of(S0, 3));                                                                         \
            W14_FENCE                                                                                       \
            W14_Q3(0, 11, FA, fby, fbx[0][0] = rdB(S1, 0, 0), fbx[0][1] = rdB(S1, 0, 1), W14_WP(0, offB1[0]), fbx[1][0] = rdB(S1, 1, 0)) \
            W14_Q3(1, 11, FA, fby, fbx[1][1] = rdB(S1, 1, 1), fbx[2][0] = rdB(S1, 2, 0), W14_WP(1, offB1[1]), fbx[2][1] = rdB(S1, 2, 1)) \
            W14_Q3(2, 11, FA, fby, fbx[3][0] = rdB(S1, 3, 0), fbx[3][1] = rdB(S1, 3, 1), W14_WP(2, offB1[2]), W14_NONE)     \
            W14_Q3(3, 11, FA, fby, FN[3][0] = rdA(S1, 3, 0), FN[3][1] = rdA(S1, 3, 1), W14_NONE, W14_NONE)                             \
            W14_END(18)                                                                                     \
        }

    int t = 0;
    for (; t + 1 < nk; t += 2) {
        W14_TILE(t, 0, 1, fa, fa2)
        W14_TILE(t + 1, 1, 0, fa2, fa)
    }
    if (t < nk) W14_TILE(t, 0, 1, fa, fa2)
#undef W14_TILE
#undef W14_END
#undef W14_Q3
#undef W14_Q4
#undef W14_NONE
#undef W14_FENCE
#undef W14_WP
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the clamped re-stages past the last tile; the last MFMAs' results
    __builtin_amdgcn_s_barrier();                                       // LDS is free

    // (lane-derived addresses of the code below must not be hoisted over the K loop: they hang off a thread id the optimiser cannot see through)
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    float *rs_lds = reinterpret_cast<float *>(lds);
    rs_lds[tid_e] = row_scale_of(row_scale, rsp, min(m0 + tid_e, T - 1));       // 256 threads, 256 rows
    __syncthreads();
    const bool whole = m0 + W14_BM <= T;                                // (N is whole tiles)
    // four column groups of 64 (the last one 32 wide: two column blocks) x four row blocks per wave
    w4_for<4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        EpiCtx ctx;
        epi_ctx_init(ctx, out, bias, rs_lds, T, N, epi, ldc, m0, n0, tn, wave, g, tid_e, ResidEpi{}, 64);
        w4_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value, cb = g * 4;
            const float4v z{0.f, 0.f, 0.f, 0.f};
            const float4v v[4] = {w4_read<4 * (q * 14 + cb)>(), w4_read<4 * (q * 14 + cb + 1)>(),
                                  g == 3 ? z : w4_read<4 * (q * 14 + (g == 3 ? 0 : cb + 2))>(), g == 3 ? z : w4_read<4 * (q * 14 + (g == 3 ? 0 : cb + 3))>()};
            if (whole) store_rows<0, g == 3 ? 2 : 4>(ctx, q, v);
            else store_rows<2, g == 3 ? 2 : 4>(ctx, q, v);
        });
    });
}

// Would a 224-column grid beat the 256-column one?  The width must be whole 224-column tiles (and the fused gate/up layout's 32-row
// pairs stay whole: 224 = 7 x 32); both grids must take the same number of rounds of the chip, so that every round is 7/8 of the work.
bool gemm_w14_plan(int64_t T, int64_t N, int64_t K, int epi) {
    const int mode = tune(TK_GEMM_W14);
    if (mode <= 0 || (epi != EPI_F32 && epi != EPI_GATEUP) || T <= 256 || N % W14_BN || K % P_BK || K / P_BK < 2) return false;
    if ((double)std::max(T, N) * (double)K * 2.0 + (double)K * 2.0 + 8192.0 >= 4294967296.0) return false;     // 32-bit lane offsets
    if (mode >= 2) return true;
    const int64_t tm = (T + 255) / 256, t224 = tm * (N / W14_BN), t256 = tm * ((N + 255) / 256);
    if (N < 8192 || (t224 + 255) / 256 > (t256 + 255) / 256) return false;
    // ... and the 224-column grid whole rounds -- or three row tiles below 768 tokens, where the 256-column grid would run on the
    // eight-wave kernel (whole Mistral-7B prefills, 256- / 224-column gate/up: 513 tokens 11.03 / 10.73 ms, 600 11.65 / 11.32,
    // 700 12.30 / 12.15; against the four-wave kernel from 768 tokens on it loses: 768 12.51 / 12.67, 1025 16.79 / 17.09)
    return t224 % 256 == 0 || (tm == 3 && T < 768);
}

int launch_gemm_w14(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                    int epi, const float *row_scale, int64_t ldc) {
    if (ldc <= 0) ldc = N;
    if (N % W14_BN || K % P_BK || K / P_BK < 2 || (epi != EPI_F32 && epi != EPI_GATEUP)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_w14: N must be whole 224-column tiles, fp32 or gate/up epilogue");
    if ((double)std::max(T, N) * (double)K * 2.0 + (double)K * 2.0 + 8192.0 >= 4294967296.0) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_w14: matrices below 4 GiB");
    const int tiles_m = (int)((T + W14_BM - 1) / W14_BM), tiles_n = (int)(N / W14_BN);
    const bool wnt = tune(TK_W14_NT) == 1 || (tune(TK_W14_NT) < 0 && T <= 512);   // (two row tiles: 512 tokens 94.6 -> 90.4 us; from three on the re-reads want the panel in L2: 1024 tokens 195 -> 202)
    auto kern = wnt ? gemm_w14_kernel<true> : gemm_w14_kernel<false>;
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), W14_LDS));
    char tag[32];
    snprintf(tag, sizeof tag, "w14,%lldx%lld", (long long)N, (long long)K);
    Launcher LL = L; LL.tag = tag;
    const int group_env = tune(TK_GEMM_GROUPM);
    const int group_m = std::max(1, std::min(tiles_m, group_env > 0 ? group_env : 4));
    return LL.launch(KC_GEMM_MFMA, ((double)N * K + (double)T * K) * 2.0, 2.0 * T * N * K, kern, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), W14_LDS,
                     (const bf16_t *)W, (const bf16_t *)x, bias, y, (int)T, (int)N, (int)K, epi, tiles_m, tiles_n, row_scale, (int)ldc, group_m, L.rsp);
}

}  // namespace fl
