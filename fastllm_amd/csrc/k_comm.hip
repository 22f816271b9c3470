// k_comm.hip -- one-shot all-reduce / all-gather over peer-mapped HBM (xGMI), for the small
// messages of tensor-parallel decode.
//
// a decode step all-reduces one [h] fp32 vector (16 KB for Mistral-7B) twice per
// layer.  A ring collective pays several xGMI hops of latency for a message that fits in one packet
// burst, so every rank instead PUSHES its vector straight into an inbox slot in every peer's HBM
// (xGMI is point-to-point: seven independent links, one 16 KB store burst on each), raises a flag
// there, waits for the tp flags in its own inbox and sums the tp slots in rank order -- the same
// order on every rank, so all ranks hold bit-identical sums and stay in lock step on the argmax.
//
// Memory: the inbox and the flags live in one hipDeviceMallocUncached allocation per rank (peer
// stores must not be hidden by the owner's L2), exported with hipIpcGetMemHandle and mapped by the
// peers (comm.hip: comm_ipc_export / comm_connect_impl).  Two inbox halves alternate by epoch parity: a rank can run at most one
// collective ahead of the slowest peer (it needs that peer's flag to finish), so the half it
// overwrites is never still being read.  Epochs count collectives and live on the device, so a
// captured decode graph replays with no argument update.
#include <algorithm>

#include "comm_ll.h"
#include "kernels.h"

namespace fl {

template <bool GATHER>
__global__ __launch_bounds__(1024) void oneshot_kernel(const float *__restrict__ in, float *__restrict__ out, CommTable tab,
                                                       int rank, int tp, int n, int nmax, long long out_stride, uint32_t *__restrict__ epoch_ctr,
                                                       uint32_t *__restrict__ err, long long timeout_ticks, uint32_t *__restrict__ abort_flag) {
    __shared__ uint32_t s_bad;
    const int tid = threadIdx.x;
    const uint32_t e = *epoch_ctr + 1;
    const size_t half = (size_t)(e & 1) * tp * nmax;
    if (tid == 0) s_bad = 0;

    // push: my vector into slot [rank] of every rank's inbox (own inbox included: one code path).  Loopback (CommTable::loop): every
    // entry is the own inbox, and the store that would go to rank p's slot [rank] goes to the own slot [p]
    const size_t my_slot = half + (size_t)rank * nmax;
    const size_t slot_step = tab.loop ? (size_t)nmax : 0, slot0 = tab.loop ? half : my_slot;
    if ((n & 3) == 0) {
        for (int i = tid * 4; i < n; i += 4096) {
            const float4v v = *reinterpret_cast<const float4v *>(in + i);
            for (int p = 0; p < tp; p++) *reinterpret_cast<float4v *>(tab.inbox[p] + slot0 + p * slot_step + i) = v;
        }
    } else {
        for (int i = tid; i < n; i += 1024) {
            const float v = in[i];
            for (int p = 0; p < tp; p++) tab.inbox[p][slot0 + p * slot_step + i] = v;
        }
    }
    __threadfence_system();                       // every store has been acknowledged by its peer before a flag is raised
    __syncthreads();
    if (tid < tp) {
        __hip_atomic_store(tab.flags[tid] + (tab.loop ? tid : rank), e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // wait for rank `tid`'s flag in my own inbox; bounded so a dead peer cannot hang the GPU
        const uint32_t *f = tab.flags[rank] + tid;
        const long long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
            __builtin_amdgcn_s_sleep(2);
            const long long waited = wall_clock64() - t0;
            if (waited > timeout_ticks || (waited > 100000 && abort_flag && *abort_flag)) { s_bad = 1; break; }   // (1 ms once a wait has given up)
        }
    }
    __syncthreads();
    __threadfence_system();                       // acquire: the slots are read only after all tp flags were seen
    if (s_bad) {                                  // give up: report, do not touch `out`
        if (tid == 0) { if (*err == 0) *err = 0xA11D0000u | (uint32_t)rank; if (abort_flag) *abort_flag = 1; *epoch_ctr = e; }      // the first report stays
        return;
    }
    const float *mine = tab.inbox[rank] + half;
    if ((n & 3) == 0) {
        for (int i = tid * 4; i < n; i += 4096) {
            if (GATHER) {
                for (int s = 0; s < tp; s++)
                    *reinterpret_cast<float4v *>(out + (size_t)s * out_stride + i) = *reinterpret_cast<const float4v *>(mine + (size_t)s * nmax + i);
            } else {
                float4v acc = *reinterpret_cast<const float4v *>(mine + i);
                for (int s = 1; s < tp; s++) acc += *reinterpret_cast<const float4v *>(mine + (size_t)s * nmax + i);
                *reinterpret_cast<float4v *>(out + i) = acc;
            }
        }
    } else {
        for (int i = tid; i < n; i += 1024) {
            if (GATHER) {
                for (int s = 0; s < tp; s++) out[(size_t)s * out_stride + i] = mine[(size_t)s * nmax + i];
            } else {
                float acc = mine[i];
                for (int s = 1; s < tp; s++) acc += mine[(size_t)s * nmax + i];
                out[i] = acc;
            }
        }
    }
    __syncthreads();
    if (tid == 0) *epoch_ctr = e;
}

// The same collective for LARGE messages (a decode batch's [B, h] deltas, a short prompt's span, a batch's logits blocks): the single
// 1024-thread workgroup above pushes 16 KB in a microsecond but 256 KB x tp peers in 65 us (one CU's store stream) -- 8.3 of the 10.4 ms
// of a 32-stream step on an 8-way group (profiles/r05/README.md).  Here G workgroups take a slice of the vector each: push it to every
// peer, raise the SLICE's flag there (flag words [64 + 8 g + source rank] of the 16 KB flag page: G <= 256 workgroups of 256 threads,
// a kilo-float slice or more each: the uncached inbox is read at a CU's ~30 GB/s, so the slices want many CUs), wait for the tp flags of
// their own slice, sum / gather it.  No workgroup waits for another workgroup of its own launch.  All of them read the epoch at their
// start; the one that FINISHES last (a ticket) advances it, so every workgroup of a launch has seen the same epoch.
constexpr int kWideFlag0 = 64, kWideMaxG = 256, kWideThreads = 256;       // (the flag page is 16 KB: 64 + 256 x 8 words)
template <bool GATHER>
__global__ __launch_bounds__(kWideThreads) void oneshot_wide_kernel(const float *__restrict__ in, float *__restrict__ out, CommTable tab,
                                                            int rank, int tp, int n, int nmax, long long out_stride, uint32_t *__restrict__ epoch_ctr,
                                                            uint32_t *__restrict__ err, long long timeout_ticks, uint32_t *__restrict__ abort_flag,
                                                            uint32_t *__restrict__ done_ctr) {
    // no LDS in this kernel: its workgroups sit on CUs waiting for peers, and a CU that holds even one LDS granule of theirs cannot
    // take a workgroup that wants the whole LDS (the decode ring kernels; on a test rig with every rank on one card that was a deadlock)
    const int tid = threadIdx.x, g = blockIdx.x, G = gridDim.x;
    const uint32_t e = *epoch_ctr + 1;
    const size_t half = (size_t)(e & 1) * tp * nmax;
    const int per = (((n + G - 1) / G) + 3) & ~3, i0 = g * per, i1 = min(n, i0 + per);     // this workgroup's slice (n % 4 == 0: whole float4s)
    const size_t my_slot = half + (size_t)rank * nmax;
    const size_t slot_step = tab.loop ? (size_t)nmax : 0, slot0 = tab.loop ? half : my_slot;
    for (int i = i0 + tid * 4; i < i1; i += kWideThreads * 4) {
        const float4v v = *reinterpret_cast<const float4v *>(in + i);
        for (int p = 0; p < tp; p++) *reinterpret_cast<float4v *>(tab.inbox[p] + slot0 + p * slot_step + i) = v;
    }
    // every wave waits until ITS pushes have left the CU (the barrier alone does not: a workgroup-scope release keeps vmcnt open on this
    // target), the barrier orders that before the flag stores below, and their system-scope RELEASE (buffer_wbl2 sc0 sc1 + vmcnt(0)) then
    // writes out whatever of them an L2 still holds: one L2 write-back per workgroup instead of one per wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int fw = kWideFlag0 + g * FL_MAX_TP;
    if (tid < tp) {
        // (fence + relaxed store rather than a release store: the fence's write-back is WAITED for -- buffer_wbl2 sc0 sc1; s_waitcnt
        // vmcnt(0) -- before the flag leaves; the compiler's release store issued the flag right behind the write-back)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(tab.flags[tid] + fw + (tab.loop ? tid : rank), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t *f = tab.flags[rank] + fw + tid;
        const long long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
            __builtin_amdgcn_s_sleep(2);
            const long long waited = wall_clock64() - t0;
            if (waited > timeout_ticks || (waited > 100000 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (__hip_atomic_exchange(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && *err == 0) *err = 0xA11D0000u | (uint32_t)rank | ((uint32_t)tid << 4) | ((uint32_t)(g & 255) << 8);   // the first report stays: who waited (bits 0-3), for whom (4-7), which slice (8-15)
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // system scope: drop what this CU and its L2 hold of the inbox; nothing of ours to write back
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {      // else: a wait gave up (here or in an earlier collective): the group is broken, `out` stays untouched
        const float *mine = tab.inbox[rank] + half;
        for (int i = i0 + tid * 4; i < i1; i += kWideThreads * 4) {
            if (GATHER) {
                for (int sl = 0; sl < tp; sl++)
                    *reinterpret_cast<float4v *>(out + (size_t)sl * out_stride + i) = *reinterpret_cast<const float4v *>(mine + (size_t)sl * nmax + i);
            } else {
                float4v acc = *reinterpret_cast<const float4v *>(mine + i);
                for (int sl = 1; sl < tp; sl++) acc += *reinterpret_cast<const float4v *>(mine + (size_t)sl * nmax + i);
                *reinterpret_cast<float4v *>(out + i) = acc;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        // the workgroup that finishes last moves the epoch (every workgroup of this launch read it before it could finish)
        if (__hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)G - 1) {
            __hip_atomic_store(done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *epoch_ctr = e;
        }
    }
}

// One collective over n <= nmax floats.  all-reduce: out[n] = sum_r in_r[n] (in == out allowed: every
// thread reads its elements of `in` before any write of `out`, and writes only its own elements).
// all-gather: out[r*out_stride + i] = in_r[i].
int launch_oneshot(Launcher &L, bool gather, const float *in, float *out, const CommTable &tab, int rank, int tp,
                   int64_t n, int64_t nmax, int64_t out_stride, uint32_t *epoch_ctr, uint32_t *err, long long timeout_ticks,
                   uint32_t *abort_flag, uint32_t *done_ctr, int max_wgs) {
    if (n <= 0 || n > nmax || tp < 1 || tp > FL_MAX_TP) FL_FAIL(FL_ERR_BAD_ARGUMENT, "one-shot collective: bad size %lld (max %lld), tp %d", (long long)n, (long long)nmax, tp);
    const double bytes = (double)n * 4 * (2.0 * tp + 1);
    L.tag = gather ? "allgather" : "allreduce";
    int rc;
    // large messages (whole float4s): slices over up to max_wgs <= 256 workgroups; the decode step's [h] vector stays on the one-workgroup form
    if (done_ctr && abort_flag && max_wgs > 1 && n >= 16384 && n % 4 == 0) {
        const int maxg = std::min(kWideMaxG, max_wgs);
        const int G = (int)std::min<int64_t>(maxg, (n + 1023) / 1024);
        L.tag = gather ? "allgather,wide" : "allreduce,wide";
        if (gather)
            rc = L.launch(KC_COMM, bytes, 0, oneshot_wide_kernel<true>, dim3((unsigned)G), dim3(kWideThreads), 0, in, out, tab, rank, tp, (int)n, (int)nmax, (long long)out_stride, epoch_ctr, err, timeout_ticks, abort_flag, done_ctr);
        else
            rc = L.launch(KC_COMM, bytes, 0, oneshot_wide_kernel<false>, dim3((unsigned)G), dim3(kWideThreads), 0, in, out, tab, rank, tp, (int)n, (int)nmax, (long long)out_stride, epoch_ctr, err, timeout_ticks, abort_flag, done_ctr);
        L.tag = "";
        return rc;
    }
    if (gather)
        rc = L.launch(KC_COMM, bytes, 0, oneshot_kernel<true>, dim3(1), dim3(1024), 0, in, out, tab, rank, tp, (int)n, (int)nmax, (long long)out_stride, epoch_ctr, err, timeout_ticks, abort_flag);
    else
        rc = L.launch(KC_COMM, bytes, 0, oneshot_kernel<false>, dim3(1), dim3(1024), 0, in, out, tab, rank, tp, (int)n, (int)nmax, (long long)out_stride, epoch_ctr, err, timeout_ticks, abort_flag);
    L.tag = "";
    return rc;
}

// The epilogue exchange of comm_ll.h on its own: four waves per workgroup, two rows per wave, like the GEMV that
// normally carries it.  in == out is allowed (a wave reads its rows before it writes them).
__global__ __launch_bounds__(256) void ll_allreduce_kernel(const LLTable *__restrict__ ll, int slot, const float *__restrict__ in,
                                                           float *__restrict__ out, int n) {
    const int lane = threadIdx.x & 63, row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= n) return;
    const float s[2] = {in[row0], row0 + 1 < n ? in[row0 + 1] : 0.f};
    ll_allreduce_rows<2>(ll, slot, row0, n, s, out, lane);
}

int launch_ll_allreduce(Launcher &L, const LLTable *ll_dev, int slot, const float *in, float *out, int64_t n) {
    if (!ll_dev || n <= 0 || slot <= 0) FL_FAIL(FL_ERR_BAD_ARGUMENT, "fused all-reduce: bad arguments");
    L.tag = "ll_allreduce";
    const int rc = L.launch(KC_COMM, (double)n * 8 * 2, 0, ll_allreduce_kernel, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, ll_dev, slot, in, out, (int)n);
    L.tag = "";
    return rc;
}

}  // namespace fl
