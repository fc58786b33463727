// conv_patch_f32.h — cnv1, cnv2 and cnv3 of the FLOAT32 mode (davo_set_precision 0: the reference's own arithmetic,
// nets/posenn.py:205-213) from an LDS-staged input patch, on v_mfma_f32_16x16x4_f32.
//
// As implicit GEMMs (conv_igemm.h) these layers ran at 0.56 / 0.51 / 0.57 of the FP32-MFMA peak (profiles/r03d, r04a): per chunk and
// thread four tap decompositions, bounds tests and 64-bit addresses, five LDS writes and a barrier around 16 matrix instructions -
// and a second set of staging registers (loads two chunks ahead) changed nothing (profiles/r04_f32_prefetch2_ab.md): they do not
// wait for memory, they pay for addressing.  The f16x3 path removed exactly that with the patch kernels of conv_patch_h3.h; this
// is the same recipe in float32.  A float32 pixel has the same bytes as a split-fp16 one (16 channels x 4 B = 64 B at cnv2's
// input, 32 x 4 = 128 B at cnv3's), so the patch geometry, the LDS-DMA staging, the bank-conflict-free region layout, the tile
// walk and the double buffering are conv_patch_cnv2_h3's / conv_patch_cnv3_h3's unchanged (cp2:: / cp3:: constants); what
// differs is the matrix phase:
//   * a 16-byte unit is FOUR float32 channels; region q of the patch holds unit q of every pixel; lane (r, kq) reads unit kq
//     (+ 4 j) of pixel r at the tap - one ds_read_b128 - and feeds element t of it to the t-th of four v_mfma_f32_16x16x4_f32:
//     instruction t contracts input channels {t, 4 + t, 8 + t, 12 + t} (+ 16 j) of the tap;
//   * the weights are the MFMA's A operand and live in registers for the whole persistent kernel (cnv2: 25 taps x 4 = 100
//     per lane; cnv3: 9 taps x 8 = 72), packed on the host in exactly that order (weights.hip);
//   * C[channel][pixel]: a lane's accumulator quad is four consecutive output channels of one pixel = one 16-byte store.
// Per output the float32 fma chain runs tap by tap (ky major), inside a tap in the channel order above: another fixed order of
// the same chain as conv_igemm_f32's ("f32_n16" in include/davo_hip.h says the same of cnv1's tile), not the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_patch_h3.h"

namespace davo {

// ---- cnv2 (5x5, stride 2, 16 -> 32 channels), float32 --------------------------------------------------------------------
// Wave w owns output channels 16 (w & 1) .. +15 and the pixel groups 2 (w >> 1), +1 of the 8 x 8 output tile (a group = 2 rows
// x 8 columns); it stages region w of the patch.  Two workgroups per CU (100 weight registers per lane; three measured 4 % slower).
__global__ __launch_bounds__(cp2::THREADS, 2) void conv_patch_cnv2_f32(ConvPatchParams p) {
    using namespace cp2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_f2[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ng = wave & 1, g0 = 2 * (wave >> 1);
    const int r = lane & 15, kq = lane >> 4;
    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + 16 * ng + 4 * kq);
    const f32x4 bv4 = {b4.x, b4.y, b4.z, b4.w};
    // unit kq of pixel (2 (g0 + G) + (r >> 3), r & 7) of the tile at tap (ky, kx): region kq, patch row 2 * row + ky, column
    // 2 (r & 7) + kx = parity kx & 1, unit (r & 7) + (kx >> 1).  Lanes 0-7 of a service group read 8 consecutive units, lanes
    // 8-15 the same units two patch rows (640 B = 128 mod 256) on: the 64 banks once (conv_patch_cnv2_h3's layout).
    const int a_lane = kq * REGION + (r >> 3) * (2 * ROWB) + (r & 7) * 16 + g0 * (4 * ROWB);

    auto issue_patch = [&](const TileCoord& tc, int buf) {
        const int iy_base = tc.ty * TH * 2 - p.pad_t, ix_base = tc.tx * TW * 2 - p.pad_l;
        const uint8_t* xin = p.x + (size_t)tc.n * p.H * p.W * 64 + wave * 16;
        uint8_t* dst = smem_f2 + buf * PATCH + wave * REGION;
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int L = k * 64 + lane;
            const int py = (L * 3277) >> 16;                  // L / 20 for L < 704
            const int rem = L - py * ROW_UNITS;
            const int par = rem >= PWU ? 1 : 0, idx = rem - par * PWU;
            const int iy = iy_base + py, ix = ix_base + 2 * idx + par;
            const bool ok = py < PH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint8_t* src = patch_src(ok, xin, (unsigned)(iy * p.W + ix) * 64u, p.zeros);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + k * 1024), 16, 0, 0);
        }
    };

    const TileWalk tw = tile_walk(p.ntiles);
    int t = tw.first, buf = 0;
    bool stores_counted = false;       // the previous tile issued exactly 2 stores per lane after this tile's patch DMA (interior tile)
    TileCoord tc = tile_coord(t, p.tiles_x, p.tiles_y);
    const TileCoord ts = tile_coord(tw.step, p.tiles_x, p.tiles_y);
    if (t < tw.end) issue_patch(tc, 0);
    // the wave's weights: [tap][N group][t][lane] float32 (weights.hip), fetched behind the first patch's DMA
    constexpr int NT = KS * KS;
    float wreg[NT][4];
    const float* wsrc = reinterpret_cast<const float*>(p.w);
#pragma unroll
    for (int tp = 0; tp < NT; ++tp)
#pragma unroll
        for (int q = 0; q < 4; ++q) wreg[tp][q] = wsrc[((size_t)(tp * 2 + ng) * 4 + q) * 64 + lane];
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));         // as conv_patch_cnv2_h3: no compiler-placed vmcnt wait inside the tile loop
    float* const yout = reinterpret_cast<float*>(p.y);
    while (t < tw.end) {
        const int n = tc.n;
        const int oy0 = tc.ty * TH, ox0 = tc.tx * TW;
        if (stores_counted) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tnext = t + tw.step;
        const TileCoord tcn = tile_next(tc, ts, p.tiles_x, p.tiles_y);
        if (tnext < tw.end) issue_patch(tcn, buf ^ 1);

        // matrix phase: the two fragments of tap s + 1 are requested before the eight matrix instructions of tap s are queued
        // (counted wait: LDS returns in order); the two groups' accumulator chains alternate (a 16x16x4 result is ready 40
        // cycles after issue, the next instruction on the same accumulator comes 64 cycles later)
        const unsigned a_u32 = lds_u32(smem_f2 + buf * PATCH + a_lane);
        f32x4 acc[2] = {bv4, bv4};
        f32x4 fr[2][2];                                        // [ring slot][group]
#define F2_RD(dst_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(a_u32), "n"(off_) : "memory")
#define F2_OFF(S_, G_) (((S_) / KS) * ROWB + (((S_) % KS) & 1) * (PWU * 16) + (((S_) % KS) >> 1) * 16 + (G_) * (4 * ROWB))
#define F2_ISSUE(S_, B_) { F2_RD(fr[B_][0], F2_OFF(S_, 0)); F2_RD(fr[B_][1], F2_OFF(S_, 1)); }
#define F2_WAIT(N_, B_) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fr[B_][0]), "+v"(fr[B_][1]) : "n"(N_))
#define F2_MM(S_, B_)                                                                              \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                         \
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[S_][q_], fr[B_][0][q_], acc[0], 0, 0, 0); \
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[S_][q_], fr[B_][1][q_], acc[1], 0, 0, 0); \
        }
#define F2_STEP(S_)                                                                                \
        {   if constexpr ((S_) + 1 < NT) { F2_ISSUE((S_) + 1, ((S_) + 1) & 1) F2_WAIT(2, (S_) & 1); } \
            else { F2_WAIT(0, (S_) & 1); }                                                         \
            F2_MM(S_, (S_) & 1)                                                                    \
            __builtin_amdgcn_sched_barrier(0); }
        F2_ISSUE(0, 0)
        F2_STEP(0) F2_STEP(1) F2_STEP(2) F2_STEP(3) F2_STEP(4) F2_STEP(5) F2_STEP(6) F2_STEP(7) F2_STEP(8) F2_STEP(9)
        F2_STEP(10) F2_STEP(11) F2_STEP(12) F2_STEP(13) F2_STEP(14) F2_STEP(15) F2_STEP(16) F2_STEP(17) F2_STEP(18) F2_STEP(19)
        F2_STEP(20) F2_STEP(21) F2_STEP(22) F2_STEP(23) F2_STEP(24)
        static_assert(NT == 25, "unrolled by hand");
#undef F2_RD
#undef F2_OFF
#undef F2_ISSUE
#undef F2_WAIT
#undef F2_MM
#undef F2_STEP

        // epilogue: lane (r, kq) holds channels 16 ng + 4 kq .. +3 of pixel r of each group: one 16-byte store per group
        const bool interior = oy0 + TH <= p.Ho && ox0 + TW <= p.Wo;   // uniform
        stores_counted = interior;
        const int c0 = 16 * ng + 4 * kq;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int oy = oy0 + 2 * (g0 + g) + (r >> 3), ox = ox0 + (r & 7);
            const bool ok = interior || (oy < p.Ho && ox < p.Wo);
            float4 v;
            v.x = fmaxf(acc[g][0], 0.f); v.y = fmaxf(acc[g][1], 0.f); v.z = fmaxf(acc[g][2], 0.f); v.w = fmaxf(acc[g][3], 0.f);
            if (ok) *reinterpret_cast<float4*>(yout + (((size_t)n * p.Ho + oy) * p.Wo + ox) * 32 + c0) = v;
        }
        t = tnext;
        tc = tcn;
        buf ^= 1;
    }
}

// ---- cnv3 (3x3, dilation 2, 32 -> 64 channels), float32 ------------------------------------------------------------------
// conv_patch_cnv3_h3's geometry: the 12 x 12-pixel patch of an 8 x 8 output tile in eight regions (unit u = channels 4 u .. +3 of
// every pixel), four pixel groups of two output rows TWO apart x 8 columns (conflict-free reads).  Wave w owns output channels
// 16 w .. +15 with its 72 weights in registers; a tap is two ds_read_b128 (units kq and kq + 4) and eight matrix instructions per
// group: instruction (j, t) contracts input channels {4 kq' + t + 16 j}.  Three workgroups per CU.
__global__ __launch_bounds__(cp3::THREADS, 3) void conv_patch_cnv3_f32(ConvPatchParams p) {
    using namespace cp3;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_f3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // = N group: channels 16 wave .. +15
    const int r = lane & 15, kq = lane >> 4;
    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + 16 * wave + 4 * kq);
    const f32x4 bv4 = {b4.x, b4.y, b4.z, b4.w};
    // unit kq (+ 4 j) of pixel r of group g at tap (ky, kx): region kq + 4 j, patch row (4 (g >> 1) + (g & 1) + 2 (r >> 3)) + 2 ky,
    // pixel (r & 7) + 2 kx
    const int a_lane = kq * REGION + (r >> 3) * (2 * ROWB) + (r & 7) * 16;

    auto issue_patch = [&](const TileCoord& tc, int buf) {
        const int iy_base = tc.ty * TH - p.pad_t, ix_base = tc.tx * TW - p.pad_l;
        const uint8_t* xin = p.x + (size_t)tc.n * p.H * p.W * 128;
        uint8_t* dst = smem_f3 + buf * PATCH;
#pragma unroll
        for (int kk = 0; kk < (NDMA + 3) / 4; ++kk) {
            const int k = wave + 4 * kk;
            if (k < NDMA) {
                const int L = k * 64 + lane;
                const int reg = L / (PH * PW), rem = L - reg * (PH * PW);
                const int py = rem / PW, px = rem - py * PW;
                const int iy = iy_base + py, ix = ix_base + px;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const uint8_t* src = patch_src(ok, xin, (unsigned)(iy * p.W + ix) * 128u + reg * 16, p.zeros);
                __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + k * 1024), 16, 0, 0);
            }
        }
    };

    const TileWalk tw = tile_walk(p.ntiles);
    int t = tw.first, buf = 0;
    bool stores_counted = false;       // the previous tile issued exactly 4 stores per lane after this tile's patch DMA (interior tile)
    TileCoord tc = tile_coord(t, p.tiles_x, p.tiles_y);
    const TileCoord ts = tile_coord(tw.step, p.tiles_x, p.tiles_y);
    if (t < tw.end) issue_patch(tc, 0);
    // the wave's weights: [tap][wave][instruction 4 j + t][lane] float32 (weights.hip)
    float wreg[STEPS][8];
    const float* wsrc = reinterpret_cast<const float*>(p.w);
#pragma unroll
    for (int tp = 0; tp < STEPS; ++tp)
#pragma unroll
        for (int q = 0; q < 8; ++q) wreg[tp][q] = wsrc[((size_t)(tp * 4 + wave) * 8 + q) * 64 + lane];
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));
    float* const yout = reinterpret_cast<float*>(p.y);
    while (t < tw.end) {
        const int n = tc.n;
        const int oy0 = tc.ty * TH, ox0 = tc.tx * TW;
        if (stores_counted) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tnext = t + tw.step;
        const TileCoord tcn = tile_next(tc, ts, p.tiles_x, p.tiles_y);
        if (tnext < tw.end) issue_patch(tcn, buf ^ 1);

        // matrix phase in half-steps of two pixel groups (as conv_patch_cnv3_h3): the four fragments of half-step h + 1 are
        // requested before the sixteen matrix instructions of half-step h are queued; the pair's accumulator chains alternate
        const unsigned a_u32 = lds_u32(smem_f3 + buf * PATCH + a_lane);
        f32x4 acc[4] = {bv4, bv4, bv4, bv4};
        f32x4 fr[2][2][2];                                     // [ring slot][group of the pair][channel half j]
#define F3_RD(dst_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(a_u32), "n"(off_) : "memory")
#define F3_OFF(H_, G_, J_) ((((H_) >> 1) / 3) * (RATE * ROWB) + (((H_) >> 1) % 3) * (RATE * 16) + (4 * ((H_) & 1) + (G_)) * ROWB + (J_) * (4 * REGION))
#define F3_ISSUE(H_, B_)                                                                           \
        { F3_RD(fr[B_][0][0], F3_OFF(H_, 0, 0)); F3_RD(fr[B_][1][0], F3_OFF(H_, 1, 0));            \
          F3_RD(fr[B_][0][1], F3_OFF(H_, 0, 1)); F3_RD(fr[B_][1][1], F3_OFF(H_, 1, 1)); }
#define F3_WAIT(N_, B_) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fr[B_][0][0]), "+v"(fr[B_][1][0]), "+v"(fr[B_][0][1]), "+v"(fr[B_][1][1]) : "n"(N_))
#define F3_MM(H_, B_)                                                                              \
        {   constexpr int g_ = 2 * ((H_) & 1), tp_ = (H_) >> 1;                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                       \
                _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                 \
                    acc[g_] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tp_][4 * j_ + q_], fr[B_][0][j_][q_], acc[g_], 0, 0, 0); \
                    acc[g_ + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tp_][4 * j_ + q_], fr[B_][1][j_][q_], acc[g_ + 1], 0, 0, 0); \
                } }
#define F3_HALF(H_)                                                                                \
        {   if constexpr ((H_) + 1 < 2 * STEPS) { F3_ISSUE((H_) + 1, ((H_) + 1) & 1) F3_WAIT(4, (H_) & 1); }   \
            else { F3_WAIT(0, (H_) & 1); }                                                         \
            F3_MM(H_, (H_) & 1)                                                                    \
            __builtin_amdgcn_sched_barrier(0); }
        F3_ISSUE(0, 0)
        F3_HALF(0) F3_HALF(1) F3_HALF(2) F3_HALF(3) F3_HALF(4) F3_HALF(5) F3_HALF(6) F3_HALF(7) F3_HALF(8)
        F3_HALF(9) F3_HALF(10) F3_HALF(11) F3_HALF(12) F3_HALF(13) F3_HALF(14) F3_HALF(15) F3_HALF(16) F3_HALF(17)
        static_assert(STEPS == 9, "unrolled by hand");
#undef F3_RD
#undef F3_OFF
#undef F3_ISSUE
#undef F3_WAIT
#undef F3_MM
#undef F3_HALF

        // epilogue: lane (r, kq) holds channels 16 wave + 4 kq .. +3 of pixel r of each group: row 4 (g >> 1) + (g & 1) + 2 (r >> 3),
        // column r & 7: one 16-byte store per group
        const bool interior = oy0 + TH <= p.Ho && ox0 + TW <= p.Wo;   // uniform
        stores_counted = interior;
        const int c0 = 16 * wave + 4 * kq;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int oy = oy0 + 4 * (g >> 1) + (g & 1) + 2 * (r >> 3), ox = ox0 + (r & 7);
            const bool ok = interior || (oy < p.Ho && ox < p.Wo);
            float4 v;
            v.x = fmaxf(acc[g][0], 0.f); v.y = fmaxf(acc[g][1], 0.f); v.z = fmaxf(acc[g][2], 0.f); v.w = fmaxf(acc[g][3], 0.f);
            if (ok) *reinterpret_cast<float4*>(yout + (((size_t)n * p.Ho + oy) * p.Wo + ox) * 64 + c0) = v;
        }
        t = tnext;
        tc = tcn;
        buf ^= 1;
    }
}

// ---- cnv1 (7x7, stride 2, 8 packed input channels -> 16), float32 --------------------------------------------------------
// conv_patch_cnv1_h3's geometry (21 x 37-pixel patch of an 8 x 16 output tile, one buffer, rows padded to 32 units): a float32
// pixel is 32 B = two 16-byte units (channels 0-3 | 4-7) where the split form had a hi and a lo unit, so the staging is the same
// bytes.  Lane (r, kq) is output column r and tap slot kq of a step (ky, h): kx = 4 h + kq (kx = 7 is a zero-weight dummy); it
// reads the tap's eight channels (two ds_read_b128) and feeds channel c to the c-th of eight v_mfma_f32_16x16x4_f32, whose four
// k are the step's four taps.  All 16 output channels in every wave (112 weights per lane); a wave owns two output rows.
// Three workgroups per CU (166 registers, 42 KB of LDS each).
__global__ __launch_bounds__(cp1::THREADS, 3) void conv_patch_cnv1_f32(ConvPatchParams p) {
    using namespace cp1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_f1[];
    uint8_t* patch = smem_f1;                  // [2 channel halves][PH][2 parities][UNITS] x 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the wave's weights: [step][channel c][lane] float32: lane (r' = output channel, kq) holds W[ky][4 h + kq][c][r'] (weights.hip)
    float wreg[STEPS][8];
    const float* wsrc = reinterpret_cast<const float*>(p.w);
#pragma unroll
    for (int st = 0; st < STEPS; ++st)
#pragma unroll
        for (int q = 0; q < 8; ++q) wreg[st][q] = wsrc[((size_t)st * 8 + q) * 64 + lane];
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));
    const int r = lane & 15, kq = lane >> 4;
    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + 4 * kq);
    const f32x4 bv4 = {b4.x, b4.y, b4.z, b4.w};
    // fragment address of subtile row oy_l: py = 2 oy_l + ky, px = 2 r + 4 h + kq: parity kq & 1, unit r + 2 h + (kq >> 1)
    const int a_lane = (kq & 1) * ROWB + (r + (kq >> 1)) * 16;
    const uint8_t* a0 = patch + (2 * (2 * wave) * 2) * ROWB + a_lane;

    auto issue_patch = [&](const TileCoord& tc) {
        const int iy_base = tc.ty * TH * 2 - p.pad_t, ix_base = tc.tx * TW * 2 - p.pad_l;
        const uint8_t* xin = p.x + (size_t)tc.n * p.H * p.W * 32;
        const int par = lane >> 5, px2 = lane & 31;
        const int ix = ix_base + 2 * px2 + par;
        const bool okx = px2 * 2 + par < PW && (unsigned)ix < (unsigned)p.W;
        const unsigned offx = (unsigned)ix * 32u;
        for (int k = wave; k < 2 * PH; k += 4) {
            const int plane = k >= PH ? 1 : 0;
            const int iy = iy_base + k - plane * PH;                    // uniform
            const bool ok = okx && (unsigned)iy < (unsigned)p.H;
            const uint8_t* src = patch_src(ok, xin, (unsigned)(iy * p.W) * 32u + offx + plane * 16, p.zeros);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(patch + k * 1024), 16, 0, 0);
        }
    };

    const TileWalk tw = tile_walk(p.ntiles);
    int t = tw.first;
    TileCoord tc = tile_coord(t, p.tiles_x, p.tiles_y);
    const TileCoord ts = tile_coord(tw.step, p.tiles_x, p.tiles_y);
    if (t < tw.end) issue_patch(tc);
    float* const yout = reinterpret_cast<float*>(p.y);
    while (t < tw.end) {
        const int n = tc.n;
        const int oy0 = tc.ty * TH, ox0 = tc.tx * TW;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 acc0 = bv4, acc1 = bv4;
        const unsigned a_u32 = lds_u32(a0);                    // subtile row 1 = a0 + 4 ROWB
        f32x4 fr[2][2][2];                                     // [ring slot][subtile row][channel half]
#define F1_RD(dst_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(a_u32), "n"(off_) : "memory")
#define F1_OFF(S_, G_, PL_) (((S_) >> 1) * 2 * ROWB + ((S_) & 1) * 32 + (G_) * (4 * ROWB) + (PL_) * PLANE)
#define F1_ISSUE(S_, B_)                                                                           \
        { F1_RD(fr[B_][0][0], F1_OFF(S_, 0, 0)); F1_RD(fr[B_][1][0], F1_OFF(S_, 1, 0));            \
          F1_RD(fr[B_][0][1], F1_OFF(S_, 0, 1)); F1_RD(fr[B_][1][1], F1_OFF(S_, 1, 1)); }
#define F1_WAIT(N_, B_) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fr[B_][0][0]), "+v"(fr[B_][1][0]), "+v"(fr[B_][0][1]), "+v"(fr[B_][1][1]) : "n"(N_))
#define F1_MM(S_, B_)                                                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                           \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                     \
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[S_][4 * j_ + q_], fr[B_][0][j_][q_], acc0, 0, 0, 0); \
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[S_][4 * j_ + q_], fr[B_][1][j_][q_], acc1, 0, 0, 0); \
            }
#define F1_STEP(S_)                                                                                \
        {   if constexpr ((S_) + 1 < STEPS) { F1_ISSUE((S_) + 1, ((S_) + 1) & 1) F1_WAIT(4, (S_) & 1); }   \
            else { F1_WAIT(0, (S_) & 1); }                                                         \
            F1_MM(S_, (S_) & 1)                                                                    \
            __builtin_amdgcn_sched_barrier(0); }
        F1_ISSUE(0, 0)
        F1_STEP(0) F1_STEP(1) F1_STEP(2) F1_STEP(3) F1_STEP(4) F1_STEP(5) F1_STEP(6)
        F1_STEP(7) F1_STEP(8) F1_STEP(9) F1_STEP(10) F1_STEP(11) F1_STEP(12) F1_STEP(13)
        static_assert(STEPS == 14, "unrolled by hand");
#undef F1_RD
#undef F1_OFF
#undef F1_ISSUE
#undef F1_WAIT
#undef F1_MM
#undef F1_STEP
        __syncthreads();                                       // every wave is done reading the patch
        const int tnext = t + tw.step;
        tc = tile_next(tc, ts, p.tiles_x, p.tiles_y);
        if (tnext < tw.end) issue_patch(tc);                   // the refill flies under this tile's stores

        // epilogue: lane (r, kq) holds output channels 4 kq .. +3 of pixel (2 wave + sub, r) of the tile: one 16-byte store per row
        const bool interior = oy0 + TH <= p.Ho && ox0 + TW <= p.Wo;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int oy = oy0 + 2 * wave + sub, ox = ox0 + r;
            const bool ok = interior || (oy < p.Ho && ox < p.Wo);
            const f32x4 a = sub == 0 ? acc0 : acc1;
            float4 v;
            v.x = fmaxf(a[0], 0.f); v.y = fmaxf(a[1], 0.f); v.z = fmaxf(a[2], 0.f); v.w = fmaxf(a[3], 0.f);
            if (ok) *reinterpret_cast<float4*>(yout + (((size_t)n * p.Ho + oy) * p.Wo + ox) * 16 + 4 * kq) = v;
        }
        t = tnext;
    }
}

}  // namespace davo
