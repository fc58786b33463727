// pose_tail.h — "last workgroup" tails: the final, tiny reduction of a launch done by whichever workgroup of that
// launch finishes last, instead of by a launch of its own (a 5-7 us launch for a few hundred additions).
//
// Protocol.  MI355X has one L2 per XCD and they are not coherent with each other, so the textbook form (plain stores,
// __threadfence(), atomic ticket) costs a release fence = a write-back of the XCD's whole L2 (buffer_wbl2) per workgroup:
// measured +70 us on the squeeze launch and +19 us on cnv7.  Instead the few words that cross workgroups are moved with
// AGENT-SCOPE accesses, which go through to the memory side on their own: the partial results are written with agent-scope
// atomic stores, the writing wave waits for their acknowledgement (s_waitcnt vmcnt(0): gfx950 counts stores there), the
// workgroup synchronises, and one thread takes a ticket from an agent-scope atomic counter.  The holder of the last ticket
// reads every workgroup's partials with agent-scope loads, in a FIXED order - the sums are bitwise the same as the separate
// kernel's and do not depend on which workgroup happens to be last.  It leaves the counter at zero for the next launch on
// the stream.  No fence, no cache write-back.
#pragma once
#include <hip/hip_runtime.h>

#include "params.h"

namespace davo {

__device__ __forceinline__ float agent_load(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void agent_store(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the calling wave's agent-scope stores have been acknowledged (vmcnt(0); expcnt / lgkmcnt untouched)
__device__ __forceinline__ void agent_stores_done() {
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));
    asm volatile("" ::: "memory");
}

// Every thread of the workgroup calls this after the partial results were written with agent_store and their writers
// passed agent_stores_done().
// ticket_lds: one free word of the workgroup's LDS (the kernels' LDS budgets are exact, so no static __shared__ here).
// -> true in every thread of the one workgroup that took the last of `total` tickets.
__device__ __forceinline__ bool last_workgroup(unsigned* counter, unsigned total, unsigned* ticket_lds) {
    __syncthreads();
    if (threadIdx.x == 0) *ticket_lds = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return *ticket_lds == total - 1;
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
