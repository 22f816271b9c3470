// comm_ll.h -- device side of the all-reduce that rides in a GEMV epilogue (LLTable, kernels.h).
//
// tensor-parallel decode all-reduces one [h] fp32 vector after o_proj and after down_proj.  As a
// kernel of its own (k_comm.hip) each of those costs a launch boundary plus push -> fence -> flag -> poll -> sum
// on one workgroup; at tp = 8 the GEMVs themselves are a few microseconds, so the 64 collectives per token are
// the larger part of the step.  Here the wave that has just reduced R output rows exchanges them itself:
//   lane p < tp:  store {value, epoch} (8 bytes, single-copy atomic) into slot [rank] of rank p's region,
//                 then poll slot [p] of the own region until its epoch matches;
//   all lanes:    add the tp values in rank order (same order, hence same bits, on every rank).
// No fence: data and validity are one store.  No deadlock: workgroups are dispatched in index order and a
// spinning wave only waits for rows whose peers' waves have an index no larger than its own.  Slot reuse: a
// half (epoch parity) is rewritten two all-reduces later; a rank reaches that kernel only after every peer has
// pushed the all-reduce in between, which each of them did in a LATER kernel than the one that read this half.
#pragma once
#include "kernels.h"

namespace fl {

template <int R>
__device__ inline void ll_allreduce_rows(const LLTable *__restrict__ t, int slot, int row0, int N, const float (&sum)[R],
                                         float *__restrict__ out, int lane) {
    const int tp = t->tp, rank = t->rank, n = t->n;
    const uint32_t e = *t->epoch_ctr * (uint32_t)t->slots + (uint32_t)slot;
    const size_t half = (size_t)(e & 1) * tp * n;
    float got[R];
#pragma unroll
    for (int r = 0; r < R; r++) got[r] = 0.f;
    if (lane < tp) {
        uint64_t *dst = t->peer[lane] + half + (size_t)(t->loop ? lane : rank) * n + row0;      // (loopback: the own region's slot [lane])
#pragma unroll
        for (int r = 0; r < R; r++)
            if (row0 + r < N)
                __hip_atomic_store(dst + r, ((uint64_t)e << 32) | (uint64_t)__float_as_uint(sum[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t *src = t->peer[rank] + half + (size_t)lane * n + row0;
        const long long t0 = wall_clock64();
        bool alive = true;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (row0 + r >= N) continue;
            uint64_t w = __hip_atomic_load(src + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            while (alive && (uint32_t)(w >> 32) != e) {
                __builtin_amdgcn_s_sleep(1);
                const long long waited = wall_clock64() - t0;
                // bounded: a dead peer must not hang the GPU; once one wait of this rank has given up, the others of the
                // broken step follow after 1 ms instead of the full bound each
                if (waited > t->timeout_ticks || (waited > 100000 && *t->abort_flag)) {
                    if (*t->err == 0) *t->err = 0xA11E0000u | (uint32_t)rank;
                    *t->abort_flag = 1;
                    alive = false;
                }
                w = __hip_atomic_load(src + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            got[r] = __uint_as_float((uint32_t)w);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        float acc = __shfl(got[r], 0, 64);
        for (int p = 1; p < tp; p++) acc += __shfl(got[r], p, 64);
        if (lane == 0 && row0 + r < N) out[row0 + r] = acc;
    }
}

}  // namespace fl
