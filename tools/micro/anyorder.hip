// Can dependent weight-streaming kernels of ONE stream overlap their ramp-up / drain on MI355X?
//
// A decode step is a chain of ~160 dependent launches; each pays ~1.5-3 us of kernel boundary + ramp.  The weights a
// kernel streams do not depend on its predecessor, only the (tiny) activation vector does.  This experiment launches
// the chain with hipExtAnyOrderLaunch (AQL packets without the barrier bit: the command processor dispatches packet
// i+1 as soon as packet i's workgroups have all been PLACED, not finished) and carries the real dependency in device
// memory: every workgroup requests its first weight block, then waits for flag[i-1], then streams.
//
//   mode 0: plain in-order launches (the baseline: what model.hip does today)
//   mode 1: hipExtAnyOrderLaunch + device flags
//   mode 2: plain launches + device flags (the cost of the flag protocol alone)
// each also replayed from a captured hipGraph.  Prints us per kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned uint4v __attribute__((ext_vector_type(4)));

struct Args {
    const uint4v *w;        // this kernel's weights: nwg * per_wg uint4
    size_t per_wg;          // uint4 per workgroup (multiple of 768 * 4)
    float *vec;             // [2][256]: hand-off vector (kernel i reads slot (i-1)&1, writes slot i&1)
    unsigned *flag;         // [nk]: completions of kernel i
    unsigned *cnt;          // [nk]: arrival counters
    unsigned *bad;
    int idx, nk, use_flags;
};

__global__ __launch_bounds__(768) void chain_kernel(const Args a) {
    const int tid = threadIdx.x, wg = blockIdx.x;
    const uint4v *p = a.w + (size_t)wg * a.per_wg;
    // first block of the weight stream: requested BEFORE the dependency is waited for
    uint4v r[4];
#pragma unroll
    for (int u = 0; u < 4; u++) r[u] = __builtin_nontemporal_load(p + tid + 768 * u);
    __shared__ float xin;
    if (a.use_flags) {
        if (tid == 0) {
            const int prev = a.idx == 0 ? a.nk - 1 : a.idx - 1;
            const unsigned own = __hip_atomic_load(a.flag + a.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = a.idx == 0 ? own : own + 1;
            long long t0 = wall_clock64();
            while (__hip_atomic_load(a.flag + prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 100000000LL) { *a.bad = 1; break; }          // 1 s
            }
            xin = __hip_atomic_load(a.vec + ((a.idx + 1) & 1) * 256 + (wg & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (tid == 0) {
        xin = a.vec[((a.idx + 1) & 1) * 256 + (wg & 255)];
    }
    __syncthreads();
    float acc = xin * 1e-30f;
    for (size_t c = 768 * 4; ; c += 768 * 4) {
        uint4v n[4];
        const bool more = c < a.per_wg;
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; u++) n[u] = __builtin_nontemporal_load(p + c + tid + 768 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += __uint_as_float(r[u][0] & 0x3f800000u) + __uint_as_float(r[u][3] & 0x3f800000u);
        if (!more) break;
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = n[u];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ float red[12];
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < 12; i++) s += red[i];
        float *dst = a.vec + (a.idx & 1) * 256 + (wg & 255);
        if (a.use_flags) {
            __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // sc1: write-through
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(a.cnt + a.idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                __hip_atomic_store(a.cnt + a.idx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned own = __hip_atomic_load(a.flag + a.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.flag + a.idx, own + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            *dst = s;
        }
    }
}

int main(int argc, char **argv) {
    const int nk = argc > 1 ? atoi(argv[1]) : 64;
    const size_t mb = argc > 2 ? atoi(argv[2]) : 48;                  // MB per kernel
    const int nwg = 256;
    size_t per_wg = mb * 1024 * 1024 / 16 / nwg;
    per_wg = per_wg / (768 * 4) * (768 * 4);
    if (per_wg < 768 * 4) per_wg = 768 * 4;                          // at least one block per workgroup (the kernel reads it unconditionally)
    const size_t per_k = per_wg * nwg;
    uint4v *w; float *vec; unsigned *flag, *cnt, *bad;
    CHECK(hipMalloc(&w, per_k * 16 * nk));
    CHECK(hipMemset(w, 0x3f, per_k * 16 * nk));
    CHECK(hipMalloc(&vec, 2 * 256 * 4)); CHECK(hipMemset(vec, 0, 2 * 256 * 4));
    CHECK(hipMalloc(&flag, nk * 4)); CHECK(hipMalloc(&cnt, nk * 4)); CHECK(hipMalloc(&bad, 4));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%d kernels x %.1f MB, %d workgroups x 768 threads\n", nk, per_k * 16 / 1e6, nwg);
    auto enqueue = [&](int mode) {
        for (int i = 0; i < nk; i++) {
            Args a{w + (size_t)i * per_k, per_wg, vec, flag, cnt, bad, i, nk, mode != 0};
            const unsigned fl = (mode == 1 && i > 0) ? hipExtAnyOrderLaunch : 0;       // kernel 0 keeps the barrier bit
            hipExtLaunchKernelGGL(chain_kernel, dim3(nwg), dim3(768), 0, s, nullptr, nullptr, fl, a);
        }
    };
    for (int mode = 0; mode < 3; mode++) {
        CHECK(hipMemset(flag, 0, nk * 4)); CHECK(hipMemset(cnt, 0, nk * 4)); CHECK(hipMemset(bad, 0, 4));
        for (int rep = 0; rep < 3; rep++) {
            enqueue(mode);                                                                // warm
            CHECK(hipStreamSynchronize(s));
            CHECK(hipEventRecord(e0, s));
            enqueue(mode);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned hb; CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            printf("mode %d eager : %.2f us per kernel (%.2f TB/s) bad=%u\n", mode, ms * 1e3 / nk, per_k * 16.0 * nk / ms / 1e9, hb);
        }
        // graph replay of the same chain
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        enqueue(mode);
        hipError_t ce = hipStreamEndCapture(s, &g);
        if (ce != hipSuccess || !g) { printf("mode %d: capture failed: %s\n", mode, hipGetErrorString(ce)); (void)hipGetLastError(); continue; }
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipGraphLaunch(ge, s));
            CHECK(hipStreamSynchronize(s));
            CHECK(hipEventRecord(e0, s));
            CHECK(hipGraphLaunch(ge, s));
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned hb; CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            printf("mode %d graph : %.2f us per kernel (%.2f TB/s) bad=%u\n", mode, ms * 1e3 / nk, per_k * 16.0 * nk / ms / 1e9, hb);
        }
        CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    }
    return 0;
}
