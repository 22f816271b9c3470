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

#include "kernels.h"

namespace fl {

typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));

constexpr int P_BM = 256, P_BN = 256, P_BK = 64;
constexpr int P_HALF = 128 * P_BK * 2;                 // 16 KiB
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
__device__ inline int a_row(int h, int r) { return (r >> 6) * 128 + h * 64 + (r & 63); }
__device__ inline int b_row(int h, int r) { return (r >> 5) * 64 + h * 32 + (r & 31); }

// one half tile = 16 wave-instructions of 1 KiB (8 rows): two per wave
template <bool IS_A>
__device__ inline void stage_half(const bf16_t *__restrict__ M, int nrows, int K, int row0, int k0, int h,
                                  unsigned char *half, int wave, int lane) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int rb = (wave * 2 + s) * 8;
        const int r = rb + (lane >> 3), pc = lane & 7, c = pc ^ ((r >> 1) & 7);
        int gr = row0 + (IS_A ? a_row(h, r) : b_row(h, r));
        if (gr > nrows - 1) gr = nrows - 1;
        glds16p(M + (size_t)gr * K + k0 + c * 8, half + rb * 128);
    }
}

__global__ __launch_bounds__(512) void gemm_8p_kernel(const bf16_t *__restrict__ W, const bf16_t *__restrict__ X,
                                                      const float *__restrict__ bias, void *__restrict__ out,
                                                      int T, int N, int K, int epi, int tiles_m, int tiles_n,
                                                      const float *__restrict__ row_scale, int ksplit, int ldc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [parity][A0 A1 B0 B1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int m16 = lane & 15, kg = lane >> 4;

    // XCD-aware remap (bijective): ids that share an XCD get consecutive tiles (same W panel in its L2)
    const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    const int li = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int tn = li / tiles_m, tm = li % tiles_m;
    const int m0 = tm * P_BM, n0 = tn * P_BN;

    // split-K: blockIdx.y owns K tiles [kt0, kt0 + nk) and writes its own fp32 slab
    const int nk_all = K / P_BK, kz = blockIdx.y;
    const int kt0 = (int)((long long)nk_all * kz / ksplit), nk = (int)((long long)nk_all * (kz + 1) / ksplit) - kt0;
    X += (size_t)kt0 * P_BK; W += (size_t)kt0 * P_BK;
    if (ksplit > 1) out = reinterpret_cast<float *>(out) + (size_t)kz * T * N;

    float4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    auto hbuf = [&](int tile, int which) -> unsigned char * { return lds + ((tile & 1) * 4 + which) * P_HALF; };   // which: 0 A0, 1 A1, 2 B0, 3 B1
    auto stA = [&](int h, int tile) { if (tile < nk) stage_half<true>(X, T, K, m0, tile * P_BK, h, hbuf(tile, h), wave, lane); };
    auto stB = [&](int h, int tile) { if (tile < nk) stage_half<false>(W, N, K, n0, tile * P_BK, h, hbuf(tile, 2 + h), wave, lane); };

    // prologue: tile 0 whole, A0 / B0 of tile 1 (what phases 3, 4 of "tile -1" would have staged)
    stA(0, 0); stB(0, 0); stB(1, 0); stA(1, 0); stA(0, 1); stB(0, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                    // group 1 runs half a phase behind (+10 %, gemm_probe)

    bf16x8p fa[4][2], fb0[2][2], fb1[2][2];
    const int arow = wr * 64 + m16, brow = wc * 32 + m16;
#define P_WAIT(TAIL)                                                                   \
    if (TAIL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         \
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#define P_MFMA(MI0, NJ0, FB)                                                           \
    __builtin_amdgcn_s_barrier();                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                 \
    __builtin_amdgcn_sched_barrier(0);   /* pins the cluster between the barriers; s_setprio around it: -2.5 % */ \
    _Pragma("unroll") for (int ks = 0; ks < 2; ks++)                                   \
        _Pragma("unroll") for (int i = 0; i < 4; i++)                                  \
            _Pragma("unroll") for (int j = 0; j < 2; j++)                              \
                acc[MI0 + i][NJ0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], FB[j][ks], acc[MI0 + i][NJ0 + j], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    __builtin_amdgcn_s_barrier();

    for (int u = 0; u < nk; u++) {
        const bool tail = u + 2 >= nk;                            // fewer than four young half tiles behind us
        const unsigned char *A0 = hbuf(u, 0), *A1 = hbuf(u, 1), *B0 = hbuf(u, 2), *B1 = hbuf(u, 3);
        // ---- phase 1: quadrant (m0, n0) ----
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) fb0[j][ks] = frag(B0, brow + j * 16, ks * 4 + kg);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) fa[i][ks] = frag(A0, arow + i * 16, ks * 4 + kg);
        stB(1, u + 1);
        P_WAIT(tail)
        P_MFMA(0, 0, fb0)
        // ---- phase 2: (m0, n1) ----
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) fb1[j][ks] = frag(B1, brow + j * 16, ks * 4 + kg);
        stA(1, u + 1);
        P_WAIT(tail)
        P_MFMA(0, 2, fb1)
        // ---- phase 3: (m1, n1) ----
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) fa[i][ks] = frag(A1, arow + i * 16, ks * 4 + kg);
        stA(0, u + 2);
        P_WAIT(tail)
        P_MFMA(4, 2, fb1)
        // ---- phase 4: (m1, n0) ----
        stB(0, u + 2);
        P_WAIT(tail)
        P_MFMA(4, 0, fb0)
    }
#undef P_MFMA
#undef P_WAIT
    if (wr == 0) __builtin_amdgcn_s_barrier();                    // balances group 1's extra barrier

    // C/D map of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int cn = lane & 15, rm = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int m = m0 + wr * 128 + i * 16 + rm + rg;
            if (m >= T) continue;
            const float rs = row_scale ? row_scale[m] : 1.0f;
            if (epi == EPI_GATEUP) {
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int n = n0 + wc * 64 + j * 16 + cn;        // gate column; up = n + 16
                    if (n + 16 < N) {
                        const int qq = (n >> 5) * 16 + (n & 15);
                        const float gt = acc[i][j][rg] * rs, up = acc[i][j + 1][rg] * rs;
                        const float a = gt / (1.0f + expf(-gt)) * up;
                        reinterpret_cast<bf16_t *>(out)[(size_t)m * (ldc / 2) + qq] = float_to_bf16_bits(a);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int n = n0 + wc * 64 + j * 16 + cn;
                    if (n < N) reinterpret_cast<float *>(out)[(size_t)m * ldc + n] = acc[i][j][rg] * rs + (bias ? bias[n] : 0.f);
                }
            }
        }
    }
}

int launch_gemm_8p(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                   int epi, const float *row_scale, int ksplit, int64_t ldc) {
    if (ldc <= 0) ldc = N;
    if (ksplit > 1 && ldc != N) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_8p: K slices write whole slabs (ldc == N)");
    const int tiles_m = (int)((T + P_BM - 1) / P_BM), tiles_n = (int)((N + P_BN - 1) / P_BN);
    if (K % P_BK || K / P_BK / ksplit < 2) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_8p: K must give at least two 64-wide tiles per slice");
    if (ksplit > 1 && (bias || epi != EPI_F32)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "split-K GEMM: fp32 epilogue without bias only");
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(gemm_8p_kernel), P_LDS));
    const double bytes = ((double)N * K + (double)T * K) * 2.0;
    char tag[32];
    snprintf(tag, sizeof tag, "8p,%lldx%lld%s", (long long)N, (long long)K, ksplit > 1 ? ",splitK" : "");
    Launcher LL = L; LL.tag = tag;
    return LL.launch(KC_GEMM_MFMA, bytes, 2.0 * T * N * K, gemm_8p_kernel, dim3((unsigned)(tiles_m * tiles_n), (unsigned)ksplit), dim3(512),
                     P_LDS, (const bf16_t *)W, (const bf16_t *)x, bias, y, (int)T, (int)N, (int)K, epi, tiles_m, tiles_n, row_scale, ksplit, (int)ldc);
}

}  // namespace fl
