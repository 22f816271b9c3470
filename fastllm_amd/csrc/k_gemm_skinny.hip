// k_gemm_skinny.hip -- projection GEMM for SHORT prompts (T <= 128): Y[T,N] = X[T,K] . W[N,K]^T.
//
// With a few dozen tokens the projection is a weight stream (Mistral-7B: 14.5 GB per pass), not MFMA work,
// and the 128x128 kernel moves it at ~2 TB/s: 48..224 workgroups with two 32 KB stages each do not keep
// enough bytes in flight.  This kernel is built for that regime:
//   * a workgroup owns ALL T tokens (BM = 32 / 64 / 128 rows of X) and a narrow strip of W rows
//     (BN = 64 or 128), so the grid is N / BN workgroups -- times a K split for the fp32 epilogue;
//   * four LDS stages, LDS-DMA three K tiles ahead, ONE raw barrier per K tile and a counted vmcnt (the two
//     youngest tiles stay in flight across it): up to 48 KB of W per workgroup outstanding;
//   * 4 waves (2 when N is small), each owning 16 or 32 (a gate/up pair) W rows against every token
//     tile, so the W bytes are read from LDS exactly once and the X tile (L2-resident) by all four waves.
// LDS image as the other GEMM kernels: [rows][64 bf16], chunk index XOR (row >> 1) & 7 on DMA source and read.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "kernels.h"

namespace fl {

typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
constexpr int S_BK = 64;
// (BM = 256 with three stages was tried for 128 < T <= 256: 9.6 ms vs 8.9 ms for the 128x128 kernel on the
// Mistral-7B prefill -- the X tile, re-read by every workgroup, then costs more than the stream gains.)

__device__ inline void glds16s(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
// non-temporal form for the weight strip: every byte is read once by one CU (cdna guide, nt-weights)
__device__ inline void glds16s_nt(const void *g, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 2);
}
__device__ inline bf16x8s frag_s(const unsigned char *tile, int row, int chunk) {
    return *reinterpret_cast<const bf16x8s *>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// epilogue of a wave's MT x NT accumulator tiles: tokens mw .. mw + 16 MT, W rows nw .. nw + 16 NT.
// C/D map of 16x16 MFMA: col = lane & 15 (W row), row = (lane >> 4) * 4 + reg (token)
template <int MT, int NT>
__device__ inline void skinny_epilogue(const float4v (&acc)[MT][NT], void *__restrict__ out, const float *__restrict__ bias,
                                       const float *__restrict__ row_scale, int T, int N, int epi, int mw, int nw, int lane) {
    const int cn = lane & 15, rm = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MT; i++) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int m = mw + i * 16 + rm + rg;
            if (m >= T) continue;
            const float rs = row_scale ? row_scale[m] : 1.0f;
            if (epi == EPI_GATEUP) {
                if constexpr (NT == 2) {
                    const int n = nw + cn;                           // gate row; up = n + 16
                    if (n + 16 < N) {
                        const int qq = (n >> 5) * 16 + (n & 15);
                        const float gt = acc[i][0][rg] * rs, up = acc[i][1][rg] * rs;
                        const float a = gt / (1.0f + expf(-gt)) * up;
                        reinterpret_cast<bf16_t *>(out)[(size_t)m * (N / 2) + qq] = float_to_bf16_bits(a);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const int n = nw + j * 16 + cn;
                    if (n < N) reinterpret_cast<float *>(out)[(size_t)m * N + n] = acc[i][j][rg] * rs + (bias ? bias[n] : 0.f);
                }
            }
        }
    }
}

// beyond BM = 128 tokens the kernel can run ceil(T / 128) token blocks per strip (FL_GEMM_SKINNY_MAXT moves the
// limit), but the 128x128 kernel is then faster: Mistral-7B prefill T = 256 9.8 vs 9.2 ms, T = 512 18.3 vs 11.7 ms
static inline int kSkinnyMaxT_() { return tune(TK_GEMM_SKINNY_MAXT); }

template <int N> __device__ inline void wait_vmcnt() {
    static_assert(N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// the last tiles of a K loop: `younger` (0 .. MAXY) tiles of PW loads each may stay outstanding
template <int MAXY, int PW> __device__ inline void wait_tail(int younger) {
    if constexpr (MAXY <= 0) { wait_vmcnt<0>(); }
    else { if (younger >= MAXY) wait_vmcnt<MAXY * PW>(); else wait_tail<MAXY - 1, PW>(younger); }
}

// BM tokens x (NW * 16 * NT) weight rows per workgroup of NWM x NW waves; NT n-tiles of 16 rows per wave column; with
// NWM = 2 the token tiles are split over two wave rows (8 waves, two per SIMD): an LDS-DMA instruction costs its wave
// ~100 cycles of issue, so at BM = 128 (32 KB per K tile = 8 instructions per wave of a 4-wave workgroup) staging, not
// the MFMAs or HBM, set the pace; eight waves halve it and let one wave's MFMAs run under its SIMD partner's issue.
// STAMP (FL_SKINNY_STAMPS=file, diagnostics): every wave sums the core-clock cycles it spends per K step in the counted
// wait, at the barrier, issuing its LDS-DMA and in fragment reads + MFMAs (tools/stamps_skinny.py)
template <int BM, int NT, int NW, int S_NSTG, bool WNT, int NWM, bool STAMP = false>
__global__ __launch_bounds__(NW * NWM * 64) void gemm_skinny_kernel(const bf16_t *__restrict__ W, const bf16_t *__restrict__ X,
                                                          const float *__restrict__ bias, void *__restrict__ out,
                                                          int T, int N, int K, int epi, const float *__restrict__ row_scale,
                                                          int ksplit, unsigned long long *__restrict__ stamps) {
    constexpr int BN = NW * 16 * NT, MT = BM / 16 / NWM;         // MT: token tiles per wave
    constexpr int XB = BM * 128, STG = XB + BN * 128;             // bytes per stage
    constexpr int NI = (BM + BN) / 8;                              // 1 KiB DMA instructions per stage
    constexpr int PW = NI / (NW * NWM);                            // ... per wave
    static_assert(NI % (NW * NWM) == 0, "stage instructions must divide evenly over the waves");
    static_assert(BM % (16 * NWM) == 0, "token tiles must divide over the wave rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
    const int wave = wave_all % NW, wm = wave_all / NW;          // wave column (W rows) / wave row (token tiles)
    const int m16 = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.z * BM;       // blockIdx.z: token block (T > BM: W strips are re-read through L2)
    const int nk_all = K / S_BK, kz = blockIdx.y;
    const int kt0 = (int)((long long)nk_all * kz / ksplit), nk = (int)((long long)nk_all * (kz + 1) / ksplit) - kt0;
    X += (size_t)kt0 * S_BK; W += (size_t)kt0 * S_BK;
    if (ksplit > 1) out = reinterpret_cast<float *>(out) + (size_t)kz * T * N;

    float4v acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int kt) {
        unsigned char *base = lds + (kt % S_NSTG) * STG;
#pragma unroll
        for (int s = 0; s < PW; s++) {
            const int q = wave_all * PW + s;                       // instruction index: 8 rows each
            const int rb = q * 8, r = rb + (lane >> 3), pc = lane & 7, c = pc ^ ((r >> 1) & 7);
            if (rb < BM) {                                         // X rows (uniform per instruction)
                int gr = m0 + r; if (gr > T - 1) gr = T - 1;
                glds16s(X + (size_t)gr * K + kt * S_BK + c * 8, base + rb * 128);
            } else {
                int gr = n0 + r - BM; if (gr > N - 1) gr = N - 1;
                if constexpr (WNT) glds16s_nt(W + (size_t)gr * K + kt * S_BK + c * 8, base + rb * 128);
                else glds16s(W + (size_t)gr * K + kt * S_BK + c * 8, base + rb * 128);
            }
        }
    };
    // note: the swizzle of a W row uses its row index inside the W tile: r - BM keeps (r >> 1) & 7 since BM % 16 == 0
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_comp = 0, t_prev = 0, t_begin = 0;
    auto tick = [&](unsigned long long &acc_c) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            acc_c += t - t_prev; t_prev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (STAMP) { t_begin = __builtin_amdgcn_s_memtime(); }
#pragma unroll
    for (int p = 0; p < S_NSTG - 1; p++)
        if (p < nk) stage(p);
    if constexpr (STAMP) { t_prev = __builtin_amdgcn_s_memtime(); }
    for (int kt = 0; kt < nk; kt++) {
        // tile kt has landed for this wave when at most the S_NSTG - 2 younger tiles' loads are outstanding
        {
            const int younger = nk - 1 - kt;                     // tiles staged after kt (wave-uniform)
            if (younger >= S_NSTG - 2) wait_vmcnt<(S_NSTG - 2) * PW>();
            else wait_tail<S_NSTG - 3, PW>(younger);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tick(c_wait);
        __builtin_amdgcn_s_barrier();                              // everyone's tile kt landed; slot (kt+3)%4 is free
        tick(c_bar);
        if (kt + S_NSTG - 1 < nk) stage(kt + S_NSTG - 1);
        tick(c_issue);
        const unsigned char *xt = lds + (kt % S_NSTG) * STG, *wt = xt + XB;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int chunk = ks * 4 + kg;
            bf16x8s b[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) b[j] = frag_s(wt, wave * 16 * NT + j * 16 + m16, chunk);
#pragma unroll
            for (int i = 0; i < MT; i++) {
                const bf16x8s a = frag_s(xt, (wm * MT + i) * 16 + m16, chunk);
#pragma unroll
                for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr (STAMP) { asm volatile("s_nop 0" ::: "memory"); tick(c_comp); }
    }

    if constexpr (STAMP) {
        if (lane == 0) {
            unsigned long long *o = stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (NW * NWM) * 8 + wave_all * 8;
            o[0] = c_wait; o[1] = c_bar; o[2] = c_issue; o[3] = c_comp; o[4] = t_prev - t_begin; o[5] = nk; o[6] = __builtin_amdgcn_s_memrealtime(); o[7] = 0;
        }
    }
    skinny_epilogue<MT, NT>(acc, out, bias, row_scale, T, N, epi, m0 + wm * MT * 16, n0 + wave * 16 * NT, lane);
}

// ---- loader waves (64 / 128 tokens).  Measured from inside (FL_SKINNY_STAMPS, tools/stamps_skinny.py; Mistral-7B gate/up at
// T = 128, 8 waves, per K step and wave): counted vmcnt wait 140 cycles (the tile has all but landed: the kernel is NOT short
// of bytes in flight -- two rings with 12 W stages ran 10-20 % SLOWER), barrier 80-340, issuing four 1-KiB LDS-DMA
// instructions 400-620, fragment reads + 16 MFMAs 730-750: a wave does these one after the other, ~1630 cycles per K step,
// 3.9 TB/s.  Here the staging moves to NLW extra waves that do nothing else (a few VGPRs each); the NW x NWM compute waves
// only meet the barrier and multiply.  Same ring, same single raw barrier per K step.
template <int BM, int NT, int NW, int NWM, int NLW, int S_NSTG, bool WNT, bool STAMP>
__global__ __launch_bounds__((NW * NWM + NLW) * 64) void gemm_skinny_ld_kernel(const bf16_t *__restrict__ W, const bf16_t *__restrict__ X,
                                                                   const float *__restrict__ bias, void *__restrict__ out,
                                                                   int T, int N, int K, int epi, const float *__restrict__ row_scale,
                                                                   int ksplit, unsigned long long *__restrict__ stamps) {
    constexpr int BN = NW * 16 * NT, MT = BM / 16 / NWM, NWV = NW * NWM;
    constexpr int XB = BM * 128, STG = XB + BN * 128;             // bytes per stage
    constexpr int NI = (BM + BN) / 8;                              // 1 KiB DMA instructions per stage
    constexpr int PW = NI / NLW;                                   // ... per loader wave
    static_assert(NI % NLW == 0, "stage instructions must divide evenly over the loader waves");
    static_assert(BM % (16 * NWM) == 0, "token tiles must divide over the wave rows");
    static_assert((S_NSTG - 2) * PW < 64, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave_all >= NWV;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.z * BM;
    const int nk_all = K / S_BK, kz = blockIdx.y;
    const int kt0 = (int)((long long)nk_all * kz / ksplit), nk = (int)((long long)nk_all * (kz + 1) / ksplit) - kt0;
    X += (size_t)kt0 * S_BK; W += (size_t)kt0 * S_BK;
    if (ksplit > 1) out = reinterpret_cast<float *>(out) + (size_t)kz * T * N;
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_comp = 0, t_prev = 0, t_begin = 0;
    auto tick = [&](unsigned long long &acc_c) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            acc_c += t - t_prev; t_prev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto dump = [&]() {
        if constexpr (STAMP) {
            if (lane == 0) {
                unsigned long long *o = stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (NWV + NLW) * 8 + wave_all * 8;
                o[0] = c_wait; o[1] = c_bar; o[2] = c_issue; o[3] = c_comp; o[4] = t_prev - t_begin; o[5] = nk; o[6] = __builtin_amdgcn_s_memrealtime(); o[7] = loader;
            }
        }
    };
    if constexpr (STAMP) { t_begin = t_prev = __builtin_amdgcn_s_memtime(); }

    if (loader) {
        const int li = wave_all - NWV;
        auto stage = [&](int kt) {
            unsigned char *base = lds + (kt % S_NSTG) * STG;
#pragma unroll
            for (int s = 0; s < PW; s++) {
                const int rb = (li * PW + s) * 8, r = rb + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);   // (r - BM keeps the swizzle: BM % 16 == 0)
                if (rb < BM) {                                     // X rows (uniform per instruction)
                    int gr = m0 + r; if (gr > T - 1) gr = T - 1;
                    glds16s(X + (size_t)gr * K + kt * S_BK + c * 8, base + rb * 128);
                } else {
                    int gr = n0 + r - BM; if (gr > N - 1) gr = N - 1;
                    if constexpr (WNT) glds16s_nt(W + (size_t)gr * K + kt * S_BK + c * 8, base + rb * 128);
                    else glds16s(W + (size_t)gr * K + kt * S_BK + c * 8, base + rb * 128);
                }
            }
        };
#pragma unroll
        for (int p = 0; p < S_NSTG - 1; p++)
            if (p < nk) stage(p);
        for (int kt = 0; kt < nk; kt++) {
            const int younger = nk - 1 - kt;                       // tiles staged after kt (wave-uniform)
            if (younger >= S_NSTG - 2) wait_vmcnt<(S_NSTG - 2) * PW>();
            else wait_tail<S_NSTG - 3, PW>(younger);
            tick(c_wait);
            __builtin_amdgcn_s_barrier();                          // tile kt is in LDS; the compute waves are done with tile kt - 1
            tick(c_bar);
            if (kt + S_NSTG - 1 < nk) stage(kt + S_NSTG - 1);
            tick(c_issue);
        }
        dump();
        return;
    }

    const int wave = wave_all % NW, wm = wave_all / NW;          // wave column (W rows) / wave row (token tiles)
    const int m16 = lane & 15, kg = lane >> 4;
    float4v acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; kt++) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tick(c_wait);
        __builtin_amdgcn_s_barrier();
        tick(c_bar);
        const unsigned char *xt = lds + (kt % S_NSTG) * STG, *wt = xt + XB;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int chunk = ks * 4 + kg;
            bf16x8s b[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) b[j] = frag_s(wt, wave * 16 * NT + j * 16 + m16, chunk);
#pragma unroll
            for (int i = 0; i < MT; i++) {
                const bf16x8s a = frag_s(xt, (wm * MT + i) * 16 + m16, chunk);
#pragma unroll
                for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr (STAMP) { asm volatile("s_nop 0" ::: "memory"); tick(c_comp); }
    }
    dump();
    skinny_epilogue<MT, NT>(acc, out, bias, row_scale, T, N, epi, m0 + wm * MT * 16, n0 + wave * 16 * NT, lane);
}

// diagnostics (FL_SKINNY_STAMPS=file): run a stamped instantiation synchronously and append its per-wave counters to the file
template <typename Kern>
static int run_stamped(Launcher &LL, Kern kst, dim3 grid, dim3 block, size_t lds, double bytes, double flops, const char *path, int waves_per_wg,
                       const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K, int epi, const float *row_scale, int ksplit) {
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kst), lds));
    const size_t nw = (size_t)grid.x * grid.y * grid.z * waves_per_wg;
    unsigned long long *d = nullptr;
    FL_HIP(hipMalloc(&d, nw * 64));
    const int rc = LL.launch(KC_GEMM_MFMA, bytes, flops, kst, grid, block, lds, (const bf16_t *)W, (const bf16_t *)x, bias, y, (int)T, (int)N, (int)K,
                             epi, row_scale, ksplit, d);
    std::vector<unsigned long long> h(nw * 8);
    FL_HIP(hipStreamSynchronize(LL.stream));
    FL_HIP(hipMemcpy(h.data(), d, nw * 64, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    if (FILE *f = fopen(path, "a")) {
        fprintf(f, "launch %lld %lld %lld %d %zu %d\n", (long long)T, (long long)N, (long long)K, epi, nw, waves_per_wg);
        for (size_t i = 0; i < nw; i++) {
            for (int j = 0; j < 8; j++) fprintf(f, "%llu ", h[i * 8 + j]);
            fprintf(f, "\n");
        }
        fclose(f);
    }
    return rc;
}

template <int BM, int NT, int NW, int NWM, bool WNT>
static int launch_skinny_ld_s(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                              int epi, const float *row_scale, int ksplit) {
    constexpr int BN = NW * 16 * NT, NI = (BM + BN) / 8, NWV = NW * NWM;
    constexpr int NLW = (NI % 8 == 0 && NWV + 8 <= 16) ? 8 : 4;    // loader waves: <= 4 (5) instructions each per K step
    constexpr int STG = (BM + BN) * 128;
    constexpr int kFit = 156 * 1024 / STG, NSTG = kFit < 6 ? kFit : 6;
    static_assert(NSTG >= 3 && NI % NLW == 0, "loader geometry");
    constexpr size_t lds = (size_t)NSTG * STG;
    const dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)ksplit, (unsigned)((T + BM - 1) / BM)), block((NWV + NLW) * 64);
    const double bytes = ((double)N * K + (double)T * K) * 2.0;
    char tag[32];
    snprintf(tag, sizeof tag, "skinny,ld,%lldx%lld%s", (long long)N, (long long)K, ksplit > 1 ? ",splitK" : "");
    Launcher LL = L; LL.tag = tag;
    if (const char *path = env_str("FL_SKINNY_STAMPS"))
        return run_stamped(LL, gemm_skinny_ld_kernel<BM, NT, NW, NWM, NLW, NSTG, WNT, true>, grid, block, lds, bytes, 2.0 * T * N * K, path, NWV + NLW,
                           W, x, bias, y, T, N, K, epi, row_scale, ksplit);
    auto kern = gemm_skinny_ld_kernel<BM, NT, NW, NWM, NLW, NSTG, WNT, false>;
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    return LL.launch(KC_GEMM_MFMA, bytes, 2.0 * T * N * K, kern, grid, block, lds, (const bf16_t *)W, (const bf16_t *)x, bias, y, (int)T, (int)N, (int)K,
                     epi, row_scale, ksplit, (unsigned long long *)nullptr);
}

template <int BM, int NT, int NW, int NSTG, bool WNT, int NWM>
static int launch_skinny_s(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                           int epi, const float *row_scale, int ksplit) {
    constexpr int BN = NW * 16 * NT;
    constexpr size_t lds = (size_t)NSTG * (BM * 128 + BN * 128);
    static_assert(lds <= 160 * 1024, "LDS ring exceeds the CU");
    static_assert((NSTG - 2) * ((BM + BN) / 8 / (NW * NWM)) < 64, "vmcnt field");
    auto kern = gemm_skinny_kernel<BM, NT, NW, NSTG, WNT, NWM>;
    FL_TRY(raise_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    const double bytes = ((double)N * K + (double)T * K) * 2.0;
    char tag[32];
    snprintf(tag, sizeof tag, "skinny,%lldx%lld%s", (long long)N, (long long)K, ksplit > 1 ? ",splitK" : "");
    Launcher LL = L; LL.tag = tag;
    const dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)ksplit, (unsigned)((T + BM - 1) / BM));
    if constexpr (BM == 128 && !WNT && NSTG == 4) {                // (the shipped kernel's own counters: FL_SKINNY_STAMPS with FL_SKINNY_LOADERS=0)
        if (const char *path = env_str("FL_SKINNY_STAMPS"))
            return run_stamped(LL, gemm_skinny_kernel<BM, NT, NW, NSTG, WNT, NWM, true>, grid, dim3(NW * NWM * 64), lds, bytes, 2.0 * T * N * K, path,
                               NW * NWM, W, x, bias, y, T, N, K, epi, row_scale, ksplit);
    }
    return LL.launch(KC_GEMM_MFMA, bytes, 2.0 * T * N * K, kern, grid, dim3(NW * NWM * 64), lds,
                     (const bf16_t *)W, (const bf16_t *)x, bias, y, (int)T, (int)N, (int)K, epi, row_scale, ksplit, (unsigned long long *)nullptr);
}

// ring depth: FL_SKINNY_STAGES (default 4; deeper rings measured SLOWER: profiles/r02/README.md); FL_SKINNY_NT: non-temporal
// weight loads for T <= 32 (-5..10 % there, +3..8 % at T = 128); FL_SKINNY_WM: two wave rows for BM >= 64
static inline int kSkinnyStages_() { return tune(TK_SKINNY_STAGES); }
static inline int kSkinnyNt_() { return tune(TK_SKINNY_NT); }
static inline int kSkinnyWm_() { return tune(TK_SKINNY_WM); }
// 64 / 128 tokens: dedicated staging waves (gemm_skinny_ld_kernel).  Off: per projection it is -1..6 % (TinyLlama down -8..15 %)
// and +10..19 % on short K loops, end to end (prefill_sweep) within the noise.  Read per call: tests and the stamps tool switch it.
static int skinny_loaders() { const int v = tune(TK_SKINNY_LOADERS); return v >= 0 ? v : (env_str("FL_SKINNY_STAMPS") ? 1 : 0); }

template <int BM, int NT, int NW>
static int launch_skinny_t(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                           int epi, const float *row_scale, int ksplit) {
    constexpr int BN = NW * 16 * NT, STG = (BM + BN) * 128, PW = (BM + BN) / 8 / NW;
    // deepest ring: LDS (<= 152 KB) and the 6-bit vmcnt field
    constexpr int kFit = 152 * 1024 / STG, kCnt = 63 / PW + 2;
    constexpr int kDeep = kFit < kCnt ? (kFit < 12 ? kFit : 12) : (kCnt < 12 ? kCnt : 12);
    // non-temporal W pieces (skinny_nt: 0 never, 2 always, 1 = by rule): up to 32 tokens, and on the large matrices at any length -- per
    // launch, cold weights, plain / nt at 64 and 128 tokens (tools/skinny_nt_probe.py): Mistral-7B gate/up 54.7 / 50.9, 62.9 / 58.2 us,
    // down_proj 27.2 / 26.0, 37.0 / 36.6; the 16-24 Mi-element matrices a tie; TinyLlama's (4-22 Mi) +1 ... +26 % (short launches:
    // the higher first-byte latency of the streaming path shows).  Whole Mistral-7B prefills 33-64 tokens -5 %, 96-128 -2.6 %.
    const bool nt = (kSkinnyNt_() == 1 && (T <= 32 || N * K >= ((int64_t)32 << 20))) || kSkinnyNt_() == 2;
    if constexpr (BM >= 64) {
        // wave rows: 2 for the 4-column workgroups; the narrow (2-column) strips of small matrices get 4 (128 tokens, gate/up
        // pairs: 24 staging instructions per K tile over 8 waves instead of 2) or 2 where the instruction count divides
        constexpr int NI = (BM + BN) / 8;
        constexpr int WM = NW == 4 ? 2 : (BM == 128 && NI % 8 == 0 ? 4 : 2);
        static_assert(NI % (NW * WM) == 0 && BM % (16 * WM) == 0, "wave rows must divide the stage and the token tiles");
        if (kSkinnyWm_()) {
            // (A/B on one box, three passes each: down_proj -1..3 %, gate/up -2..6 %, TinyLlama down -8..15 %; but a 64-token
            // workgroup with only 16 K steps -- Mistral's QKV in four slices -- loses 10-19 %: the loaders' start-up is not amortised)
#ifdef FL_EXPERIMENTAL
            if (skinny_loaders() && (BM == 128 || (K / S_BK) / ksplit >= 20))
                return nt ? launch_skinny_ld_s<BM, NT, NW, WM, true>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit)
                          : launch_skinny_ld_s<BM, NT, NW, WM, false>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit);
#endif
            constexpr int kStg = 4 * STG <= 160 * 1024 ? 4 : 3;
            return nt ? launch_skinny_s<BM, NT, NW, kStg, true, WM>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit)
                      : launch_skinny_s<BM, NT, NW, kStg, false, WM>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit);
        }
    }
    if constexpr (kDeep > 4 && BM <= 32) {
        if (kSkinnyStages_() > 4)
            return nt ? launch_skinny_s<BM, NT, NW, kDeep, true, 1>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit)
                      : launch_skinny_s<BM, NT, NW, kDeep, false, 1>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit);
    }
    if constexpr (4 * STG <= 160 * 1024)
        return nt ? launch_skinny_s<BM, NT, NW, 4, true, 1>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit)
                  : launch_skinny_s<BM, NT, NW, 4, false, 1>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit);
    FL_FAIL(FL_ERR_UNSUPPORTED, "gemm_skinny: this tile needs the 8-wave form (FL_SKINNY_WM=1)");
}

// T <= 128: always (one token block).  129..256 tokens (two token blocks, the strip re-read through L2) only for matrices
// below 24 Mi elements, where the tiled kernels' grids are far short of the chip -- measured per projection at T = 256,
// 8-wave workgroups: TinyLlama QKV 18.2 -> 15.3 us, gate/up 38.6 -> 27.1, down 30.7 -> 15.8, Mistral o_proj 27.0 -> 24.2, but
// Mistral QKV 31.0 -> 36.2, gate/up 91 -> 112, lm_head 94 -> 172 (TinyLlama T = 256 prefill 3.05 -> 2.0 ms).
bool gemm_skinny_supported(int64_t T, int64_t N, int64_t K) {
    if (!(T > 1 && K % S_BK == 0 && K / S_BK >= 4 && N >= 64)) return false;
    if (T <= kSkinnyMaxT_()) return true;
    const int max2 = tune(TK_GEMM_SKINNY_MAXT2);
    return T <= max2 && N * K < ((int64_t)24 << 20);
}

// K slices for the fp32 epilogue: enough workgroups to cover the chip twice, at least 8 K tiles per slice
int gemm_skinny_ksplit(int64_t T, int64_t N, int64_t K, int epi, int max_split) {
    if (epi != EPI_F32 || max_split <= 1) return 1;
    const int64_t wgs = (N + 63) / 64;
    int ks = 1;
    while (ks < max_split && wgs * ks < 512 && (K / S_BK) / (ks + 1) >= 8) ks++;
    return ks;
}

int launch_gemm_skinny(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                       int epi, const float *row_scale, int ksplit) {
    if (!gemm_skinny_supported(T, N, K)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_skinny: unsupported shape");
    if (ksplit > 1 && (bias || epi != EPI_F32)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "split-K GEMM: fp32 epilogue without bias only");
    if ((K / S_BK) / ksplit < 1) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gemm_skinny: too many K slices");
    const bool pair = epi == EPI_GATEUP;                          // a wave must hold gate and up rows
    // narrow strips (2 waves) when 4-wave strips would leave most CUs without a workgroup
    const bool narrow = (N + (pair ? 127 : 63)) / (pair ? 128 : 64) * ksplit * ((T + 127) / 128) < 160;
#define FL_SK(BMV)                                                                                             \
    if (narrow) return pair ? launch_skinny_t<BMV, 2, 2>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit)    \
                            : launch_skinny_t<BMV, 1, 2>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit);   \
    return pair ? launch_skinny_t<BMV, 2, 4>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit)                \
                : launch_skinny_t<BMV, 1, 4>(L, W, x, bias, y, T, N, K, epi, row_scale, ksplit);
    if (T <= 32) { FL_SK(32) }
    if (T <= 64) { FL_SK(64) }
    FL_SK(128)
#undef FL_SK
}

}  // namespace fl
