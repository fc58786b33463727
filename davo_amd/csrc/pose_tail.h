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

// One wave's sum over the per-tile partial sums that cover image n, head / component hk (partial[head][mtile][ntile][slot][k],
// slot 0 = the tile's first image, 1 = the next one): entry j = (tile t0 + j / ntn, N tile j % ntn) goes to lane j % 64 (a
// lane adds its entries in order), then a fixed butterfly over the lanes — the same bits whoever computes it (the
// pose_from_tiles kernel or the cnv7 launch's last workgroup) and however many lanes hold an entry.  Every lane returns the sum.
template <bool AGENT>
__device__ __forceinline__ float pose_tile_sum(const float* __restrict__ partial, int n, int hk, int P, int bm, int mtiles, int ntn, int lane) {
    const int head = hk / 3, k = hk - head * 3;
    const int t0 = (n * P) / bm;
    int t1 = ((n + 1) * P - 1) / bm;
    if (t1 > mtiles - 1) t1 = mtiles - 1;
    const int count = (t1 - t0 + 1) * ntn;
    float tot = 0.f;
    for (int j = lane; j < count; j += 64) {
        const int t = t0 + j / ntn, nt = j - (j / ntn) * ntn;
        const int slot = (t * bm) / P == n ? 0 : 1;
        const float* pp = partial + (((long)head * mtiles + t) * ntn + nt) * 6 + slot * 3 + k;
        tot += AGENT ? agent_load(pp) : *pp;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    return tot;
}

// pose[n][head*3+k] = 0.01 * (bias + (1/P) * sum over the tiles that cover image n); one wave per output, like the
// pose_from_tiles kernel (prologue.h).  Called by every thread of the last workgroup of a y_mode 2 launch.
template <int THREADS>
__device__ __forceinline__ void pose_from_tiles_tail(const ConvParamsH& p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < p.pose_NB * 6; i += THREADS / 64) {
        const int n = i / 6, hk = i - n * 6;
        const float tot = pose_tile_sum<true>(p.pose_partial, n, hk, p.pose_P, p.pose_bm, p.pose_mt, p.ntiles_n, lane);
        if (lane == 0) p.pose_out[i] = 0.01f * (tot / (float)p.pose_P + p.pose_bias[hk]);
    }
    if (threadIdx.x == 0) __hip_atomic_store(p.pose_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace davo
