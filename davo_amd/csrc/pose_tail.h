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


// ---- split-K: the fix-up folded into the launch (batch 1: two 7 us launches fewer per forward) -----------------------------
// The S parts of a tile are the workgroups (x, 0..S-1) of a grid whose x extent is a multiple of 8, and workgroup ids go round-robin
// over the 8 XCDs (tools/exp/dispatch_probe.hip; checked once per context by xcd_round_robin_probe): ALL parts of a tile run on ONE
// XCD and their partial sums meet in that XCD's L2.  So, unlike the tails above, nothing has to cross to the memory side: plain
// stores, a wait for their acknowledgement (the L2 has them), a ticket, and the part that takes the last ticket reads the tile's
// partial sums back through the same L2 (its CU's vector cache is invalidated first), adds them in the fix-up kernel's fixed order
// (prologue.h: splitk_fixup - the same bits), applies ReLU, writes the stored form and notes the range.
template <int THREADS, int BMH, int BNH>
__device__ __forceinline__ void splitk_tail(const ConvParamsH& p, int mtile, int ntile, unsigned* ticket_lds) {
    agent_stores_done();                                       // this thread's partial sums are in the L2
    const int tile = (mtile - p.mtile0) * p.ntiles_n + ntile;
    if (!last_workgroup(p.sk_counter + tile, (unsigned)p.sk_parts, ticket_lds)) return;
    asm volatile("buffer_inv sc1" ::: "memory");               // nothing stale in this CU's vector cache
    const float* __restrict__ part = reinterpret_cast<const float*>(p.y);
    const int S = p.sk_parts, N = p.Cout;
    const int m0 = mtile * BMH, n0 = ntile * BNH;
    const float lo_clamp = p.sk_relu ? 0.f : -65504.f;
    float vmax = 0.f;
    // A thread owns four channels (one float4) of every (THREADS / (BNH / 4))-th row.  The loop is all latency if a row's S loads are
    // issued one after the other (first version: +16 us per layer instead of -7): RU rows x S parts are requested before the first
    // addition.  Per element the additions are the fix-up kernel's: ((p0 + p1) + p2) + p3.
    constexpr int C4 = BNH / 4, RSTEP = THREADS / C4, RU = 4;
    static_assert(THREADS % C4 == 0 && BMH % (RSTEP * RU) == 0, "split-K tail: tile / workgroup shape");
    const int c4 = threadIdx.x % C4, rbase = threadIdx.x / C4;
    const int n = n0 + 4 * c4;
    const bool n_ok = n + 3 < N;                                     // N is a multiple of 32: four channels are in or out together
    for (int r0 = rbase; r0 < BMH; r0 += RSTEP * RU) {
        float4 v[RU][4];
#pragma unroll
        for (int i = 0; i < RU; ++i) {
            const long m = m0 + r0 + i * RSTEP;
            const bool ok = n_ok && m < p.M;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                v[i][s] = (ok && s < S) ? *reinterpret_cast<const float4*>(part + (m * S + s) * N + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < RU; ++i) {
            const long m = m0 + r0 + i * RSTEP;
            if (!(n_ok && m < p.M)) continue;
            float x[4] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w};
#pragma unroll
            for (int s = 1; s < 4; ++s)
                if (s < S) { x[0] += v[i][s].x; x[1] += v[i][s].y; x[2] += v[i][s].z; x[3] += v[i][s].w; }
            unsigned short hi[4], lo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = fmaxf(x[k], lo_clamp);
                vmax = fmaxf(vmax, fabsf(t));
                t = fminf(t, 65504.f);
                const _Float16 h = (_Float16)t;
                hi[k] = __builtin_bit_cast(unsigned short, h);
                lo[k] = __builtin_bit_cast(unsigned short, (_Float16)(t - (float)h));
            }
            uint8_t* o = p.sk_y + m * (long)N * 4 + (n >> 5) * 128 + (n & 31) * 2;
            *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
            *reinterpret_cast<uint2*>(o + 64) = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
        }
    }
    if (p.sk_range) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        range_note(p.sk_range, vmax, (threadIdx.x & 63) == 0);
    }
    if (threadIdx.x == 0) __hip_atomic_store(p.sk_counter + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace davo
