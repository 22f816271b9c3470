// What a bare read stream reaches on MI355X, by request form -- the ceiling the decode GEMV is measured against.
// Every launch reads one 224 MiB buffer exactly once (the gate/up matrix of Mistral-7B is 235 MB); the launches rotate
// over five buffers (1.1 GiB: more than the 256 MiB Infinity Cache), so every byte comes from HBM.
//   mode 0: global_load_dwordx4 nt to registers, 4 x 1 KiB per wave in flight (the GEMV's request form), 12 waves / CU
//   mode 1: the same with the default cache policy
//   mode 2: global_load_lds_dwordx4 nt into a wave-private LDS ring, 7 x 2 KiB per wave in flight, 8 waves / CU
//   mode 3: the same with the default cache policy
//   mode 4: mode 0 with 8 x 1 KiB per wave in flight
//   mode 5: mode 0 on two 6-wave workgroups per CU
// Prints us per launch and TB/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned uint4v __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(768) void reg_stream(const uint4v *__restrict__ p, size_t n16, unsigned *sink) {
    const size_t nthr = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // a wave reads 1 KiB contiguous per instruction; its U instructions of a round are U KiB apart in one block
    const size_t wave = t >> 6, lane = t & 63, nwave = nthr >> 6;
    unsigned acc = 0;
    for (size_t b = wave * U * 64; b + U * 64 <= n16; b += nwave * U * 64) {
        uint4v r[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = NT ? __builtin_nontemporal_load(p + b + u * 64 + lane) : p[b + u * 64 + lane];
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= r[u][0] ^ r[u][3];
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <bool NT>
__global__ __launch_bounds__(512) void dma_stream(const unsigned char *__restrict__ p, size_t bytes, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];      // 8 waves x 8 stages x 2 KiB
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned char *mine = ring + wave * 8 * 2048;
    const size_t nwave = (size_t)gridDim.x * 8, w = (size_t)blockIdx.x * 8 + wave;
    const size_t nstage = bytes / 2048;
    int slot = 0;
    for (size_t s = w; s < nstage; s += nwave) {
        const unsigned char *g = p + s * 2048 + lane * 16;
        unsigned char *l = mine + slot * 2048;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, NT ? 2 : 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + 1024), (__attribute__((address_space(3))) void *)(l + 1024), 16, 0, NT ? 2 : 0);
        slot = (slot + 1) & 7;
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                       // seven stages (14 instructions) stay in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ring[threadIdx.x * 16] == 0x5a && bytes == 1) *sink = 1;
}

int main() {
    const size_t bytes = (size_t)224 << 20;
    const int nbuf = 5;
    unsigned char *buf[nbuf]; unsigned *sink;
    for (int i = 0; i < nbuf; i++) { CHECK(hipMalloc(&buf[i], bytes)); CHECK(hipMemset(buf[i], 0x3c + i, bytes)); }
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(dma_stream<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(dma_stream<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[] = {"registers nt, 4 KiB / wave, 12 waves", "registers default, 4 KiB / wave, 12 waves", "LDS-DMA nt, 14 KiB / wave, 8 waves",
                           "LDS-DMA default, 14 KiB / wave, 8 waves", "registers nt, 8 KiB / wave, 12 waves", "registers nt, 4 KiB / wave, 2 x 6 waves"};
    for (int mode = 0; mode < 6; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            const int iters = 40;
            for (int it = -5; it < iters; it++) {
                if (it == 0) CHECK(hipEventRecord(e0));
                const unsigned char *p = buf[(it + 5) % nbuf];
                switch (mode) {
                    case 0: hipLaunchKernelGGL((reg_stream<4, true>), dim3(256), dim3(768), 0, 0, (const uint4v *)p, bytes / 16, sink); break;
                    case 1: hipLaunchKernelGGL((reg_stream<4, false>), dim3(256), dim3(768), 0, 0, (const uint4v *)p, bytes / 16, sink); break;
                    case 2: hipLaunchKernelGGL((dma_stream<true>), dim3(256), dim3(512), 128 * 1024, 0, p, bytes, sink); break;
                    case 3: hipLaunchKernelGGL((dma_stream<false>), dim3(256), dim3(512), 128 * 1024, 0, p, bytes, sink); break;
                    case 4: hipLaunchKernelGGL((reg_stream<8, true>), dim3(256), dim3(768), 0, 0, (const uint4v *)p, bytes / 16, sink); break;
                    default: hipLaunchKernelGGL((reg_stream<4, true>), dim3(512), dim3(384), 0, 0, (const uint4v *)p, bytes / 16, sink); break;
                }
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("mode %d (%s): %.2f us per launch, %.2f TB/s\n", mode, names[mode], ms * 1e3 / iters, bytes * (double)iters / ms / 1e9);
        }
    }
    return 0;
}
