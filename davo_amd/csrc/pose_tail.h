// pose_tail.h — "last workgroup" tails: the final, tiny reduction of a launch done by whichever workgroup of that
// launch finishes last, instead of by a launch of its own (a 5-7 us launch for a few hundred additions).
//
// Protocol (the classic threadfence reduction): a workgroup writes its partial results, each writing thread fences
// (release at agent scope: the XCDs' L2s are separate), the workgroup synchronises, one thread takes a ticket from an
// agent-scope atomic counter.  The holder of the last ticket fences again (acquire) and reads every workgroup's
// partials with agent-scope loads, in a FIXED order - the sums are bitwise the same as the separate kernel's and do
// not depend on which workgroup happens to be last.  It leaves the counter at zero for the next launch on the stream.
#pragma once
#include <hip/hip_runtime.h>

#include "params.h"

namespace davo {

__device__ __forceinline__ float agent_load(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every thread of the workgroup calls this after the partial results were written and fenced by their writers.
// ticket_lds: one free word of the workgroup's LDS (the kernels' LDS budgets are exact, so no static __shared__ here).
// -> true in every thread of the one workgroup that took the last of `total` tickets.
__device__ __forceinline__ bool last_workgroup(unsigned* counter, unsigned total, unsigned* ticket_lds) {
    __syncthreads();
    if (threadIdx.x == 0) *ticket_lds = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = *ticket_lds == total - 1;
    if (last) __threadfence();
    return last;
}

// pose[n][head*3+k] = 0.01 * (bias + (1/P) * sum over the tiles that cover image n); same order as pose_from_tiles
// (prologue.h): tile-major, N tile minor.  Called by every thread of the last workgroup of a y_mode 2 launch.
template <int THREADS>
__device__ __forceinline__ void pose_from_tiles_tail(const ConvParamsH& p) {
    const int P = p.pose_P, bm = p.pose_bm, mtiles = p.pose_mt, ntn = p.ntiles_n;
    for (int i = threadIdx.x; i < p.pose_NB * 6; i += THREADS) {
        const int n = i / 6, hk = i - n * 6, head = hk / 3, k = hk - head * 3;
        const int t0 = (n * P) / bm, t1 = ((n + 1) * P - 1) / bm;
        float tot = 0.f;
        for (int t = t0; t <= t1 && t < mtiles; ++t) {
            const int slot = (t * bm) / P == n ? 0 : 1;
            const float* pp = p.pose_partial + (((long)head * mtiles + t) * ntn) * 6 + slot * 3 + k;
            for (int nt = 0; nt < ntn; ++nt) tot += agent_load(pp + nt * 6);
        }
        p.pose_out[i] = 0.01f * (tot / (float)P + p.pose_bias[hk]);
    }
    if (threadIdx.x == 0) __hip_atomic_store(p.pose_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace davo
