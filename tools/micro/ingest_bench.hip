// ingest_bench.hip -- how fast can a CU pull bytes INTO LDS with buffer_load ... lds (the GEMM kernels' staging path)?
// One workgroup per CU (grid = CUs x wgs_per_cu), 4 waves, each wave keeps P 1-KiB pieces in flight (counted vmcnt), no MFMA.
//   mode 0: every workgroup streams its own region of a buffer far larger than the caches (HBM)
//   mode 1: every workgroup re-reads a 128-KiB region of its own (L2-resident after the first pass)
//   mode 2: 3/8 of the pieces as mode 0, 5/8 as mode 1 (a GEMM's W : X mix at 512 tokens)
// build: hipcc --offload-arch=gfx950 -O3 -o build/ingest_bench tools/micro/ingest_bench.hip ; run: build/ingest_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int int4w __attribute__((ext_vector_type(4)));

__device__ inline int4w rsrc_of(const void *p) {
    const unsigned long long a = (unsigned long long)p;
    return int4w{(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)0x7fffffff, 0x00020000};
}

template <int P, bool NT = false>
__global__ __launch_bounds__(256) void ingest_kernel(const unsigned char *buf, size_t region_bytes, int pieces_per_wave, int mode, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 4 waves x P KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char *hbm = buf + ((size_t)blockIdx.x * 4 + wave) * region_bytes;               // this wave's stream
    const unsigned char *l2r = buf + (size_t)gridDim.x * 4 * region_bytes + ((size_t)blockIdx.x * 4 + wave) * 32768;   // this wave's 32-KiB hot region
    const int4w rh = rsrc_of(hbm), rl = rsrc_of(l2r);
    const unsigned ldsw = (unsigned)(size_t)lds + wave * (P * 1024);
    const unsigned voff = lane * 16;
    auto piece = [&](int i) {
        const unsigned m0 = ldsw + (i % P) * 1024;
        asm volatile("s_mov_b32 m0, %0" ::"s"(m0) : "memory");
        //   mode 3: every wave alternates HBM / L2 pieces (1 : 1); mode 4: waves 0, 1 stream HBM only, waves 2, 3 L2 only (the same bytes)
        const bool hot = mode == 1 || (mode == 2 && (i & 7) >= 3) || (mode == 3 && (i & 1)) || (mode == 4 && wave >= 2);
        if (hot) {
            const unsigned so = (unsigned)((i * 1024) & 32767);
            asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rl), "s"(so) : "memory");
        } else {
            const unsigned so = (unsigned)(((size_t)i * 1024) % region_bytes);
            if (NT) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen nt lds" ::"v"(voff), "s"(rh), "s"(so) : "memory");
            else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rh), "s"(so) : "memory");
        }
    };
    for (int i = 0; i < P; i++) piece(i);
    for (int i = P; i < pieces_per_wave; i++) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P - 1) : "memory");
        piece(i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned *>(lds + 16);
}

// the same streams through global_load_dwordx4 into VGPRs (P loads of 16 B per lane in flight, xor-folded into a sink register)
template <int P, bool NT = true>
__global__ __launch_bounds__(256) void vgpr_kernel(const unsigned char *buf, size_t region_bytes, int pieces_per_wave, int mode, unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char *hbm = buf + ((size_t)blockIdx.x * 4 + wave) * region_bytes + lane * 16;
    const unsigned char *l2r = buf + (size_t)gridDim.x * 4 * region_bytes + ((size_t)blockIdx.x * 4 + wave) * 32768 + lane * 16;
    int4w acc{0, 0, 0, 0};
    for (int i0 = 0; i0 < pieces_per_wave; i0 += P) {
        int4w v[P];
#pragma unroll
        for (int j = 0; j < P; j++) {
            const int i = i0 + j;
            const bool hot = mode == 1 || (mode == 2 && (i & 7) >= 3);
            const unsigned char *p = hot ? l2r + ((i * 1024) & 32767) : hbm + ((size_t)i * 1024) % region_bytes;
            v[j] = NT ? __builtin_nontemporal_load(reinterpret_cast<const int4w *>(p)) : *reinterpret_cast<const int4w *>(p);
        }
#pragma unroll
        for (int j = 0; j < P; j++) acc ^= v[j];
    }
    if (acc.x == 0x12345678) sink[blockIdx.x] = acc.y ^ acc.z ^ acc.w;
}

template <int P, bool NT = true>
static void run_v(const unsigned char *buf, size_t region, int ppw, int mode, int nwg, unsigned *sink, const char *what) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    vgpr_kernel<P, NT><<<nwg, 256>>>(buf, region, ppw, mode, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) vgpr_kernel<P, NT><<<nwg, 256>>>(buf, region, ppw, mode, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nwg * 4 * ppw * 1024.0 * reps;
    printf("%-28s VGPR loads%s, %2d in flight, wgs=%4d: %7.1f us per launch, %6.2f TB/s total, %6.1f GB/s per workgroup\n", what, NT ? " nt" : "   ", P, nwg, ms * 1e3 / reps,
           bytes / ms / 1e9, bytes / ms / 1e6 / nwg);
}

template <int P, bool NT = false>
static void run(const unsigned char *buf, size_t region, int ppw, int mode, int nwg, unsigned *sink, const char *what) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ingest_kernel<P, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * P * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ingest_kernel<P, NT><<<nwg, 256, 4 * P * 1024>>>(buf, region, ppw, mode, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) ingest_kernel<P, NT><<<nwg, 256, 4 * P * 1024>>>(buf, region, ppw, mode, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nwg * 4 * ppw * 1024.0 * reps;
    printf("%-32s LDS-DMA%s P=%2d wgs=%4d: %7.1f us per launch, %6.2f TB/s total, %6.1f GB/s per workgroup\n", what, NT ? " nt" : "   ", P, nwg, ms * 1e3 / reps, bytes / ms / 1e9,
           bytes / ms / 1e6 / nwg);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t region = 2u << 20;                                   // 2 MiB per wave -> 2 GiB for 256 x 4 waves: beyond every cache
    const int ppw = 2048;                                             // pieces per wave and launch (2 MiB)
    unsigned char *buf; unsigned *sink;
    const size_t total = (size_t)cus * 2 * 4 * region + (size_t)cus * 2 * 4 * 32768 + 4096;
    CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&sink, 4096 * 4));
    const char *names[3] = {"HBM stream", "L2-resident (32 KiB / wave)", "3/8 HBM + 5/8 L2"};
    for (int mode = 0; mode < 3; mode++) {
        for (int wpc = 1; wpc <= 2; wpc++) {
            run<8>(buf, region, ppw, mode, cus * wpc, sink, names[mode]);
            run<16>(buf, region, ppw, mode, cus * wpc, sink, names[mode]);
            if (wpc == 1) run<32>(buf, region, ppw, mode, cus * wpc, sink, names[mode]);
        }
    }
    run<16>(buf, region, ppw, 3, cus, sink, "1/2 HBM + 1/2 L2, every wave");
    run<16>(buf, region, ppw, 4, cus, sink, "1/2 HBM + 1/2 L2, by wave");
    run<16, true>(buf, region, ppw, 3, cus, sink, "1/2 HBM + 1/2 L2, every wave");
    run<16, true>(buf, region, ppw, 4, cus, sink, "1/2 HBM + 1/2 L2, by wave");
    run<16, true>(buf, region, ppw, 0, cus, sink, names[0]);
    run<16, true>(buf, region, ppw, 2, cus, sink, names[2]);
    run_v<16, false>(buf, region, ppw, 0, cus, sink, names[0]);
    run_v<16, false>(buf, region, ppw, 1, cus, sink, names[1]);
    run_v<16, false>(buf, region, ppw, 2, cus, sink, names[2]);
    for (int mode = 0; mode < 3; mode++) {
        run_v<8>(buf, region, ppw, mode, cus, sink, names[mode]);
        run_v<16>(buf, region, ppw, mode, cus, sink, names[mode]);
        run_v<16>(buf, region, ppw, mode, cus * 2, sink, names[mode]);
    }
    return 0;
}
