// Cost of an all-to-all hand-off between phases of one persistent launch on MI355X (256 workgroups, one per CU):
// every workgroup writes its slice of a 16 KB vector, signals, waits for all, then reads the WHOLE vector.
// Variants: (0) agent-scope release / acquire fences + relaxed counter;  (1) sc1 (write-through) stores, relaxed
// counter, sc1 loads, no fences;  (2) sc1 data + per-workgroup flags polled in parallel;  (3) / (4) the hand-off a
// QKV + attention fusion would need: only the 32 workgroups that hold one kv head's rows exchange (768 floats, one
// counter per group) -- (3) group = blockIdx % 8 (the workgroups the dispatcher places on one XCD), (4) group =
// blockIdx / 32 (spread over all XCDs).  Prints microseconds per phase; compare with a kernel boundary (~3.5 us + ramp).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void handoff_kernel(float *vec /* 2 x 4096 */, unsigned *counters, int nphase, float *sink, unsigned *bad) {
    __shared__ float xs[4096];
    const int tid = threadIdx.x, nwg = gridDim.x, wg = blockIdx.x;
    float acc = 0.f;
    for (int p = 0; p < nphase; p++) {
        float *cur = vec + (p & 1) * 4096;
        // produce: this workgroup's 16 floats
        if (tid < 4096 / 256) {
            const float v = (float)(p + 1) + 0.001f * (wg * 16 + tid);
            // MODE 1: system-scope relaxed atomic store = write-through past the XCD's L2 (sc0 sc1)
            if (MODE >= 1) __hip_atomic_store(cur + wg * 16 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else cur[wg * 16 + tid] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (MODE >= 3) {
            const int grp = MODE == 3 ? wg % 8 : wg / 32;
            if (tid == 0) {
                __hip_atomic_fetch_add(counters + (size_t)p * 8 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                long long t0 = wall_clock64();
                while (__hip_atomic_load(counters + (size_t)p * 8 + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32u) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > 200000000LL) { *bad = 1; break; }
                }
            }
            __syncthreads();
            // consume: the group's 32 x 16 floats
            if (tid < 512) {
                const int m = tid >> 4, src = MODE == 3 ? m * 8 + grp : grp * 32 + m;
                xs[tid] = __hip_atomic_load(cur + src * 16 + (tid & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __syncthreads();
            acc += xs[(tid * 7 + p) & 511];
            const int me = MODE == 3 ? wg / 8 : wg % 32, oth = (me + 13) % 32, osrc = MODE == 3 ? oth * 8 + grp : grp * 32 + oth;
            if (xs[me * 16] != (float)(p + 1) + 0.001f * (wg * 16)) *bad = 2;
            if (xs[oth * 16 + 3] != (float)(p + 1) + 0.001f * (osrc * 16 + 3)) *bad = 3;
            __syncthreads();
            continue;
        }
        if (MODE == 2) {
            // per-workgroup flags: no read-modify-write, every thread polls one producer's flag
            if (tid == 0) __hip_atomic_store(counters + (size_t)p * nwg + wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (tid < nwg) {
                long long t0 = wall_clock64();
                while (__hip_atomic_load(counters + (size_t)p * nwg + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
                    if (wall_clock64() - t0 > 200000000LL) { *bad = 1; break; }
                }
            }
        } else if (tid == 0) {
            if (MODE == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_fetch_add(counters + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long t0 = wall_clock64();
            while (__hip_atomic_load(counters + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 200000000LL) { *bad = 1; break; }
            }
            if (MODE == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        __syncthreads();
        // consume: the whole vector into LDS (as a GEMV prologue would)
        for (int i = tid; i < 4096; i += 512) {
            if (MODE >= 1) xs[i] = __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else xs[i] = cur[i];
        }
        __syncthreads();
        acc += xs[(tid * 7 + p) & 4095];
        if (xs[wg * 16] != (float)(p + 1) + 0.001f * (wg * 16)) *bad = 2;      // own slot of THIS phase
        if (xs[((wg + 97) % nwg) * 16 + 3] != (float)(p + 1) + 0.001f * (((wg + 97) % nwg) * 16 + 3)) *bad = 3;   // someone else's
        __syncthreads();
    }
    sink[wg * 512 + tid] = acc;
}

int main() {
    const int nwg = 256, nphase = 2000;
    float *vec, *sink; unsigned *cnt, *bad;
    CHECK(hipMalloc(&vec, 2 * 4096 * 4)); CHECK(hipMalloc(&sink, nwg * 512 * 4));
    CHECK(hipMalloc(&cnt, (size_t)nphase * nwg * 4)); CHECK(hipMalloc(&bad, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemset(cnt, 0, (size_t)nphase * nwg * 4)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(vec, 0, 2 * 4096 * 4));
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(handoff_kernel<0>, dim3(nwg), dim3(512), 0, 0, vec, cnt, nphase, sink, bad);
            else if (mode == 1) hipLaunchKernelGGL(handoff_kernel<1>, dim3(nwg), dim3(512), 0, 0, vec, cnt, nphase, sink, bad);
            else if (mode == 2) hipLaunchKernelGGL(handoff_kernel<2>, dim3(nwg), dim3(512), 0, 0, vec, cnt, nphase, sink, bad);
            else if (mode == 3) hipLaunchKernelGGL(handoff_kernel<3>, dim3(nwg), dim3(512), 0, 0, vec, cnt, nphase, sink, bad);
            else hipLaunchKernelGGL(handoff_kernel<4>, dim3(nwg), dim3(512), 0, 0, vec, cnt, nphase, sink, bad);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned b; CHECK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
            printf("mode %d (%s): %.3f us per phase  bad=%u\n", mode, mode == 4 ? "groups of 32 by blockIdx / 32" : mode == 3 ? "groups of 32 by blockIdx % 8 (one XCD)" : mode == 2 ? "sc1 data + per-workgroup flags, parallel poll" : mode ? "sc1 stores/loads, no fences" : "release/acquire fences", ms * 1e3 / nphase, b);
        }
    }
    // reference: empty kernel boundaries
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(handoff_kernel<0>, dim3(nwg), dim3(512), 0, 0, vec, cnt, 0, sink, bad);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("back-to-back launches of a 256 x 512 kernel with no phases: %.3f us per launch\n", ms * 1e3 / 2000);
    return 0;
}
