// conv_patch_h3.h — cnv1 (7x7, stride 2, 8 packed input channels -> 16) for the f16x3 path.
//
// As an implicit GEMM with a global-memory gather, cnv1 re-reads every input pixel ~12 times
// (49 taps / stride^2): 1.4 GB of L2 traffic per 32-triplet batch for a 109 MB input, which made
// it load-bound at 4x its matrix-pipe time.  Here a workgroup stages the input PATCH of its
// 8 x 16 output tile (21 x 37 pixels, 25 KB) in LDS once by LDS-DMA and builds the MFMA A
// fragments straight from the patch: the 8 channels of one tap of one pixel are exactly one
// 16-byte fragment of v_mfma_f32_16x16x32_f16 (K = 32 = 4 taps x 8 channels, N = 16 = Cout, so
// no padded output columns either).
//
// Patch layout: [plane hi|lo][py][column parity][px/2] x 16 B, rows padded to 32 units, so the
// 16 lanes of a ds_read_b128 group (consecutive output columns, stride-2 input columns of one
// parity) read 16 consecutive units = all 64 banks once; the two taps sharing a lane group differ
// only by multiples of 256 B.  Taps are enumerated 8 per filter row (kx = 7 is a zero-weight
// dummy) so the 4 taps of an MFMA step always lie in one filter row: 14 steps.
//
// Arithmetic is the same fp16 hi/lo split as conv_igemm_h3.h (3 MFMAs per step, one float32
// accumulator, weights pre-scaled by a power of two).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_igemm_h3.h"
#include "prologue.h"

namespace davo {

#ifdef DAVO_TUNING
#define CP_DBG(bit_) ((p.dbg & (bit_)) != 0)
#else
#define CP_DBG(bit_) false
#endif

// Tile walk of the persistent patch kernels.  Workgroup ids go round-robin over the 8 XCDs, each with an L2 of its own, so
// XCD x = id % 8 walks the CONTIGUOUS tile range [x * per, (x + 1) * per): tiles whose patches overlap (the halo is 1.5x - 2x
// the input) are staged by workgroups that share an L2.  Every tile is visited exactly once whatever the hardware's actual
// placement; a grid that is not a multiple of 8 walks t, t + grid, ... as before.
struct TileWalk { int first, step, end; };
__device__ __forceinline__ TileWalk tile_walk(int ntiles) {
    TileWalk w;
    if ((gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7, per = (ntiles + 7) >> 3;
        w.first = x * per + (blockIdx.x >> 3);
        w.step = gridDim.x >> 3;
        w.end = (x + 1) * per < ntiles ? (x + 1) * per : ntiles;
    } else {
        w.first = blockIdx.x; w.step = gridDim.x; w.end = ntiles;
    }
    return w;
}

// (image, tile row, tile column) of a tile index, advanced by a constant stride without divisions: every wave of a
// workgroup runs this scalar arithmetic on the CU's one scalar unit, and two runtime divisions per tile and use were a
// quarter of the loop's scalar instructions.
struct TileCoord { int n, ty, tx; };
__device__ __forceinline__ TileCoord tile_coord(int t, int tiles_x, int tiles_y) {
    TileCoord c;
    const int per_img = tiles_x * tiles_y;
    c.n = t / per_img;
    const int tt = t - c.n * per_img;
    c.ty = tt / tiles_x;
    c.tx = tt - c.ty * tiles_x;
    return c;
}
__device__ __forceinline__ TileCoord tile_next(TileCoord c, const TileCoord& s, int tiles_x, int tiles_y) {
    c.tx += s.tx;
    if (c.tx >= tiles_x) { c.tx -= tiles_x; ++c.ty; }
    c.ty += s.ty;
    if (c.ty >= tiles_y) { c.ty -= tiles_y; ++c.n; }
    c.n += s.n;
    return c;
}
// source of one 16-byte patch unit: its pixel when inside the image, else the zero line — as two selects, not a branch
__device__ __forceinline__ const uint8_t* patch_src(bool ok, const uint8_t* img, unsigned off, const uint8_t* zeros) {
    const unsigned long a = reinterpret_cast<unsigned long>(img) + off, z = reinterpret_cast<unsigned long>(zeros);
    return reinterpret_cast<const uint8_t*>(ok ? a : z);
}

// Two activations -> their stored form: v = min(max(x * scale, 0), 65504) (ReLU; 65504 = the fp16 range),
// H = {fp16(v0), fp16(v1)}, L = {fp16(v0 - H0), fp16(v1 - H1)}; vm collects max v before the upper clamp (range monitor).
// Scalar float arithmetic on purpose: no device code of this library uses packed float32 instructions (v_pk_mul_f32 /
// v_pk_add_f32 / v_pk_fma_f32; davo_amd/_lib.py, HIPCC_FLAGS).  Same roundings as the 2-vector form it replaces.
__device__ __forceinline__ void split_pair_relu(float x0, float x1, float scale, float& vm, unsigned& H, unsigned& L) {
    float v0 = fmaxf(x0 * scale, 0.f), v1 = fmaxf(x1 * scale, 0.f);
    vm = fmaxf(vm, fmaxf(v0, v1));
    v0 = fminf(v0, 65504.f);
    v1 = fminf(v1, 65504.f);
    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
    const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
    H = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    L = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
}

// Persistent form: the grid is 3 workgroups per CU; each loads the 28 KB of B fragments into registers
// once and then walks its share of the output tiles (tile t, t + gridDim.x, ...), re-filling only the
// 42 KB input patch per tile.  The co-resident workgroups overlap each other's fill / matrix / store phases.
template <bool FUSED>
__global__ __launch_bounds__(cp1::THREADS, 3) void conv_patch_cnv1_h3(ConvPatchParams p) {
    using namespace cp1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_p[];
    uint8_t* patch = smem_p;                   // [2][PH][2][UNITS] x 16 B

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the B (weight) fragments of all 14 steps, hi and lo, live in registers for the whole kernel: [step][plane][lane] x 16 B.
    // (Staged in LDS they were a third of the kernel's fragment reads and 28 KB that now admit a third workgroup per CU.)
    half8 wreg[STEPS][2];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        wreg[st][0] = *reinterpret_cast<const half8*>(p.w + ((size_t)(st * 2 + 0) * 64 + lane) * 16);
        wreg[st][1] = *reinterpret_cast<const half8*>(p.w + ((size_t)(st * 2 + 1) * 64 + lane) * 16);
    }
    // the weight loads are waited for HERE, with the builtin the compiler's wait-count pass understands: an inline-asm wait is
    // opaque to it, and it would otherwise put its own vmcnt(0) at the weights' first use inside the tile loop - behind the
    // next patch's DMA, serialising the prefetch (tools/check_isa.py guards the loops of all three kernels)
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));

    const int r = lane & 15, kq = lane >> 4;
    const float bv = p.bias[r] * p.bias_scale;                 // C/D layout: col = lane & 15
    // A fragment address of subtile row oy_l: py = 2*oy_l + ky, px = 2*r + 4*h + kq
    const int a_lane = (kq & 1) * ROWB + (r + (kq >> 1)) * 16;
    const uint8_t* a0 = patch + (2 * (2 * wave) * 2) * ROWB + a_lane;

    // stage one tile's patch.  !FUSED: LDS-DMA from the packed tensor, one wave-instruction = 64 units
    // = one py, both parities.  FUSED: every thread builds 3-4 patch pixels from the raw inputs
    // (same arithmetic as mask_pack<16>) and writes their hi / lo units.
    auto issue_patch = [&](const TileCoord& tc) {
        const int n = tc.n, ty = tc.ty, tx = tc.tx;
        const int iy_base = ty * TH * 2 - p.pad_t, ix_base = tx * TW * 2 - p.pad_l;
        if constexpr (!FUSED) {
            const uint8_t* xin = p.x + (size_t)n * p.H * p.W * 32;
            // the lane's column is the same in every piece: its bounds test and byte offset are computed once per tile
            const int par = lane >> 5, px2 = lane & 31;
            const int ix = ix_base + 2 * px2 + par;
            const bool okx = px2 * 2 + par < PW && (unsigned)ix < (unsigned)p.W && !CP_DBG(1);
            const unsigned offx = (unsigned)ix * 32u;
            for (int k = wave; k < 2 * PH; k += 4) {
                const int plane = k >= PH ? 1 : 0;
                const int iy = iy_base + k - plane * PH;                    // uniform
                const bool ok = okx && (unsigned)iy < (unsigned)p.H;
                const uint8_t* src = patch_src(ok, xin, (unsigned)(iy * p.W) * 32u + offx + plane * 16, p.zeros);
                __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(patch + k * 1024), 16, 0, 0);
            }
        } else {
            const int b = n >> 1, s = n & 1;
            const Variant& v = p.v;
            const size_t HW = (size_t)p.H * p.W;
            const float* tab_s = p.tab + ((size_t)b * 3 + 1 + s) * NCLS;
            const float* tab_t = p.tab + (size_t)b * 3 * NCLS;
            // PW + 1 columns: unit (parity 1, px2 = 18) is read by the zero-weight dummy tap kx = 7 of
            // the last output column and must hold finite data (0 x NaN would poison the sum)
            constexpr int PWF = PW + 1;
            for (int idx = tid; idx < PH * PWF; idx += THREADS) {
                const int py = idx / PWF, px = idx - py * PWF;
                const int iy = iy_base + py, ix = ix_base + px;
                float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                    const uint8_t* row = p.img + ((size_t)b * p.H + iy) * (size_t)(9 * p.W);
                    const uint8_t* pt = row + (size_t)(p.W + ix) * 3;
                    const uint8_t* ps = row + (size_t)((s ? 2 * p.W : 0) + ix) * 3;
                    const size_t pix = (size_t)iy * p.W + ix;
                    float as = 1.f, at = 1.f;
                    if (v.att_source != 0) as = att_lookup(tab_s, p.seg[((size_t)b * 3 + (s ? 2 : 0)) * HW + pix]);
                    if (v.att_source == 3) at = att_lookup(tab_t, p.seg[((size_t)b * 3 + 1) * HW + pix]);
                    r[0] = u8_to_unit(pt[0]); r[1] = u8_to_unit(pt[1]); r[2] = u8_to_unit(pt[2]);
                    r[3] = u8_to_unit(ps[0]); r[4] = u8_to_unit(ps[1]); r[5] = u8_to_unit(ps[2]);
                    if (v.mask_rgb) {
                        r[0] *= at; r[1] *= at; r[2] *= at;
                        r[3] *= as; r[4] *= as; r[5] *= as;
                    }
                    if (v.cin_per_frame == 5) {
                        const float2 f = *reinterpret_cast<const float2*>(p.flow + (((size_t)b * 4 + s) * HW + pix) * 2);
                        r[6] = v.mask_info ? f.x * as : f.x;
                        r[7] = v.mask_info ? f.y * as : f.y;
                    }
                }
                _Float16 hl[16];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const _Float16 h = (_Float16)r[k];
                    hl[k] = h;
                    hl[8 + k] = (_Float16)(r[k] - (float)h);
                }
                uint8_t* dst = patch + (py * 2 + (px & 1)) * ROWB + (px >> 1) * 16;
                *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(&hl[0]);
                *reinterpret_cast<float4*>(dst + PLANE) = *reinterpret_cast<const float4*>(&hl[8]);
            }
        }
    };

    const TileWalk tw = tile_walk(p.ntiles);
    int t = tw.first;
    float vmax = 0.f;
    TileCoord tc = tile_coord(t, p.tiles_x, p.tiles_y);
    const TileCoord ts = tile_coord(tw.step, p.tiles_x, p.tiles_y);
    if (t < tw.end) issue_patch(tc);
    while (t < tw.end) {
        const int n = tc.n;
        const int oy0 = tc.ty * TH, ox0 = tc.tx * TW;
        // explicit drain: when ordinary loads are mixed with LDS-DMA (FUSED fill after the weight DMA)
        // hipcc's own vmcnt bookkeeping does not reliably cover the DMA before the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // Matrix phase, software-pipelined by hand (conv_patch_cnv2_h3 has the reasoning): the four fragments of step s+1 are
        // requested before the six MFMAs of step s are queued, counted waits, alternating accumulator chains.
        f32x4 acc0 = {bv, bv, bv, bv}, acc1 = acc0;
#ifdef DAVO_CNV1_AUTO
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            const int ky = step >> 1, h = step & 1;
            const int aoff = ky * 2 * ROWB + h * 32;
            const half8 bh = wreg[step][0], bl = wreg[step][1];
            const half8 ah0 = lds_frag(a0 + aoff), al0 = lds_frag(a0 + aoff + PLANE);
            const half8 ah1 = lds_frag(a0 + 4 * ROWB + aoff), al1 = lds_frag(a0 + 4 * ROWB + aoff + PLANE);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh, acc1, 0, 0, 0);
        }
#else
        const unsigned a_u32 = lds_u32(a0);                    // subtile row 1 = a0 + 4 ROWB
        half8 fh[2][2], fl[2][2];                              // [ring slot][subtile row]
#define C1_RD(dst_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(a_u32), "n"(off_) : "memory")
#define C1_OFF(S_, G_, PL_) (((S_) >> 1) * 2 * ROWB + ((S_) & 1) * 32 + (G_) * (4 * ROWB) + (PL_) * PLANE)
#define C1_ISSUE(S_, B_)                                                                           \
        { C1_RD(fh[B_][0], C1_OFF(S_, 0, 0)); C1_RD(fh[B_][1], C1_OFF(S_, 1, 0));                  \
          C1_RD(fl[B_][0], C1_OFF(S_, 0, 1)); C1_RD(fl[B_][1], C1_OFF(S_, 1, 1)); }
#define C1_WAIT(N_, B_) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fh[B_][0]), "+v"(fh[B_][1]), "+v"(fl[B_][0]), "+v"(fl[B_][1]) : "n"(N_))
#define C1_MM(S_, B_)                                                                              \
        {   const half8 bh = wreg[S_][0], bl = wreg[S_][1];                                        \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[B_][0], bh, acc0, 0, 0, 0);           \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[B_][1], bh, acc1, 0, 0, 0);           \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[B_][0], bl, acc0, 0, 0, 0);           \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[B_][1], bl, acc1, 0, 0, 0);           \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[B_][0], bh, acc0, 0, 0, 0);           \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[B_][1], bh, acc1, 0, 0, 0); }
#define C1_STEP(S_)                                                                                \
        {   if constexpr ((S_) + 1 < STEPS) { C1_ISSUE((S_) + 1, ((S_) + 1) & 1) C1_WAIT(4, (S_) & 1); }   \
            else { C1_WAIT(0, (S_) & 1); }                                                         \
            C1_MM(S_, (S_) & 1)                                                                    \
            __builtin_amdgcn_sched_barrier(0); }
        C1_ISSUE(0, 0)
        C1_STEP(0) C1_STEP(1) C1_STEP(2) C1_STEP(3) C1_STEP(4) C1_STEP(5) C1_STEP(6)
        C1_STEP(7) C1_STEP(8) C1_STEP(9) C1_STEP(10) C1_STEP(11) C1_STEP(12) C1_STEP(13)
        static_assert(STEPS == 14, "unrolled by hand");
#undef C1_RD
#undef C1_OFF
#undef C1_ISSUE
#undef C1_WAIT
#undef C1_MM
#undef C1_STEP
#endif
        __syncthreads();                                       // every wave is done reading the patch
        const int tnext = t + tw.step;
        tc = tile_next(tc, ts, p.tiles_x, p.tiles_y);
        if (tnext < tw.end) issue_patch(tc);                   // the refill flies under this tile's stores

        // ---- epilogue: C/D of 16x16x32: col = lane & 15 (channel), row = 4*(lane>>4) + i (pixel).
        // even lanes store the hi halves of channels (n, n+1), odd lanes the lo halves of (n-1, n): one 4-byte store per value
        const bool odd = r & 1;
        const unsigned sel = odd ? 0x03020706u : 0x05040100u;   // perm(xn, x, sel): even = {x.lo16, xn.lo16}, odd = {xn.hi16, x.hi16}
        const int choff = odd ? 32 + (r - 1) * 2 : r * 2;
        const bool interior = oy0 + TH <= p.Ho && ox0 + TW <= p.Wo;        // uniform: every tile of a 128x416 / 256x832 frame
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int oy = oy0 + 2 * wave + sub;
            if (!interior && oy >= p.Ho) continue;
            uint8_t* __restrict__ orow = p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox0 + 4 * kq) * 64 + choff;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = fmaxf((sub == 0 ? acc0[i] : acc1[i]) * p.out_scale, 0.f);
                const bool ok = interior || ox0 + 4 * kq + i < p.Wo;
                if (ok) vmax = fmaxf(vmax, v);
                v = fminf(v, 65504.f);
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                const unsigned x = (unsigned)__builtin_bit_cast(unsigned short, hi) |
                                   ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
                const unsigned xn = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                if (ok && !CP_DBG(2)) *reinterpret_cast<unsigned*>(orow + i * 64) = __builtin_amdgcn_perm(xn, x, sel);
            }
        }
        t = tnext;
    }
    if (p.range) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        // one address for the whole launch: only a wave that would raise the record pays for the atomic
        range_note(p.range, vmax, lane == 0);
    }
}

// ---- cnv2 (5x5, stride 2, 16 -> 32 channels) from an LDS patch ------------------------------------
// As an implicit GEMM cnv2 was bound by its gather (13 chunks x 128-byte rows assembled from 16-byte pieces of two taps:
// 0.043 ms at B=32, 0.028 with the loads pointed at one line, 0.003 of it matrix time).  Here a workgroup stages the
// 19 x 20-pixel input patch of an 8 x 8 output tile once (24 KB, double-buffered: the next tile's patch flies under this
// tile's matrix phase and stores) and builds the A fragments of v_mfma_f32_16x16x32_f16 straight from it: K = 32 = 2 taps
// x 16 channels, so a lane's fragment is 8 channels (one 16-byte unit) of one tap of one pixel.
// Patch layout: four regions [plane hi|lo][channel half], each [patch row][column parity][10 units] x 16 B.  A pixel group
// is 2 output rows x 8 columns: lanes 0-7 read 8 consecutive units, lanes 8-15 the same units two patch rows (640 B = 128
// mod 256) further on, and region bases are multiples of 256 B, so the 16 lanes of every ds_read_b128 service group
// cover the 64 banks once whatever mix of k-quarters the group holds.  Taps are enumerated 6 per filter row (kx = 5 is a
// zero-weight dummy): 15 steps.  Wave w owns output channels 16 (w & 1) .. +15 (its 30 weight fragments live in registers
// for the whole kernel) and the pixel groups 2 (w >> 1), +1.  Persistent: each workgroup walks tiles t, t + grid, ...
__global__ __launch_bounds__(cp2::THREADS, 2) void conv_patch_cnv2_h3(ConvPatchParams p) {
    using namespace cp2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_p2[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ng = wave & 1, g0 = 2 * (wave >> 1);
    const int r = lane & 15, kq = lane >> 4;
    // The weights are the MFMA's A operand, the pixels its B operand: C[channel][pixel], so a lane's accumulator quad is
    // FOUR CONSECUTIVE CHANNELS 16 ng + 4 kq .. +3 of pixel r of the group, and the split store is one 8-byte store of hi
    // halves and one of lo halves per quad, with no lane exchange (half the stores and less than half the epilogue
    // arithmetic of the pixel-major form; same products in the same order per accumulator).
    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + 16 * ng + 4 * kq);
    const f32x4 bv4 = {b4.x * p.bias_scale, b4.y * p.bias_scale, b4.z * p.bias_scale, b4.w * p.bias_scale};
    // pixel fragment of group g, step (ky, h), plane pl:  region (pl, kq & 1), patch row 2 (2 g + (r >> 3)) + ky, parity kq >> 1
    // (kx = 2 h + (kq >> 1)), unit (r & 7) + h
    const int a_lane = (kq & 1) * REGION + (kq >> 1) * (PWU * 16) + (r >> 3) * (2 * ROWB) + (r & 7) * 16 + g0 * (4 * ROWB);

    // this wave stages region `wave` (plane = wave >> 1, channel half = wave & 1): piece k, lane l = linear unit 64 k + l of the region
    auto issue_patch = [&](const TileCoord& tc, int buf) {
        const int iy_base = tc.ty * TH * 2 - p.pad_t, ix_base = tc.tx * TW * 2 - p.pad_l;
        const uint8_t* xin = p.x + (size_t)tc.n * p.H * p.W * 64 + (wave >> 1) * 32 + (wave & 1) * 16;
        uint8_t* dst = smem_p2 + buf * PATCH + wave * REGION;
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int L = k * 64 + lane;
            const int py = (L * 3277) >> 16;                  // L / 20 for L < 704
            const int rem = L - py * ROW_UNITS;
            const int par = rem >= PWU ? 1 : 0, idx = rem - par * PWU;
            const int iy = iy_base + py, ix = ix_base + 2 * idx + par;
            const bool ok = py < PH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && !CP_DBG(1);
            const uint8_t* src = patch_src(ok, xin, (unsigned)(iy * p.W + ix) * 64u, p.zeros);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + k * 1024), 16, 0, 0);
        }
    };

    const TileWalk tw = tile_walk(p.ntiles);
    int t = tw.first, buf = 0;
    float vmax = 0.f;
    bool stores_counted = false;       // the previous tile issued exactly 4 stores per lane after this tile's patch DMA (interior tile)
    TileCoord tc = tile_coord(t, p.tiles_x, p.tiles_y);
    const TileCoord ts = tile_coord(tw.step, p.tiles_x, p.tiles_y);
    if (t < tw.end) issue_patch(tc, 0);
    // the wave's 30 weight fragments (30 KB) are fetched behind the first patch's DMA
    half8 wreg[STEPS][2];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        wreg[st][0] = *reinterpret_cast<const half8*>(p.w + ((size_t)((st * 2 + ng) * 2 + 0) * 64 + lane) * 16);
        wreg[st][1] = *reinterpret_cast<const half8*>(p.w + ((size_t)((st * 2 + ng) * 2 + 1) * 64 + lane) * 16);
    }
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));         // see conv_patch_cnv1_h3: no compiler-placed vmcnt wait inside the tile loop
    while (t < tw.end) {
        const int n = tc.n;
        const int oy0 = tc.ty * TH, ox0 = tc.tx * TW;
        // this tile's patch has landed (vmcnt counts in issue order and the previous tile's stores are younger than this
        // patch's DMA: they may stay in flight); behind the barrier every wave has also left the previous tile's matrix
        // phase, so the other buffer may be refilled: the next tile's patch flies under this tile's matrix phase and stores
        if (stores_counted && !CP_DBG(2)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tnext = t + tw.step;
        const TileCoord tcn = tile_next(tc, ts, p.tiles_x, p.tiles_y);
        if (tnext < tw.end) issue_patch(tcn, buf ^ 1);

        // Matrix phase, software-pipelined by hand (left to the compiler every step read its fragments, waited for
        // lgkmcnt(0) and ran three dependent MFMAs per accumulator): the four fragments of step s+1 are requested before the
        // six MFMAs of step s are queued, the wait is counted (LDS returns in order) and names the fragments it releases, and
        // the two groups' accumulator chains alternate.  Per accumulator the products still arrive as hi*hi, hi*lo, lo*hi.
        const unsigned a_u32 = lds_u32(smem_p2 + buf * PATCH + a_lane);
        f32x4 acc[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] = bv4;
        half8 fh[2][2], fl[2][2];                              // [ring slot][group]
#define C2_RD(dst_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(a_u32), "n"(off_) : "memory")
#define C2_OFF(S_, G_, PL_) (((S_) / 3) * ROWB + ((S_) % 3) * 16 + (G_) * (4 * ROWB) + (PL_) * (2 * REGION))
#define C2_ISSUE(S_, B_)                                                                           \
        { C2_RD(fh[B_][0], C2_OFF(S_, 0, 0)); C2_RD(fh[B_][1], C2_OFF(S_, 1, 0));                  \
          C2_RD(fl[B_][0], C2_OFF(S_, 0, 1)); C2_RD(fl[B_][1], C2_OFF(S_, 1, 1)); }
#define C2_WAIT(N_, B_) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fh[B_][0]), "+v"(fh[B_][1]), "+v"(fl[B_][0]), "+v"(fl[B_][1]) : "n"(N_))
#define C2_MM(S_, B_)                                                                              \
        {   const half8 bh = wreg[S_][0], bl = wreg[S_][1];                                        \
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, fh[B_][0], acc[0], 0, 0, 0);       \
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, fh[B_][1], acc[1], 0, 0, 0);       \
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, fh[B_][0], acc[0], 0, 0, 0);       \
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, fh[B_][1], acc[1], 0, 0, 0);       \
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, fl[B_][0], acc[0], 0, 0, 0);       \
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, fl[B_][1], acc[1], 0, 0, 0); }
#define C2_STEP(S_)                                                                                \
        {   if constexpr ((S_) + 1 < STEPS) { C2_ISSUE((S_) + 1, ((S_) + 1) & 1) C2_WAIT(4, (S_) & 1); }   \
            else { C2_WAIT(0, (S_) & 1); }                                                         \
            C2_MM(S_, (S_) & 1)                                                                    \
            __builtin_amdgcn_sched_barrier(0); }
        C2_ISSUE(0, 0)
        C2_STEP(0) C2_STEP(1) C2_STEP(2) C2_STEP(3) C2_STEP(4) C2_STEP(5) C2_STEP(6) C2_STEP(7)
        C2_STEP(8) C2_STEP(9) C2_STEP(10) C2_STEP(11) C2_STEP(12) C2_STEP(13) C2_STEP(14)
        static_assert(STEPS == 15, "unrolled by hand");
#undef C2_RD
#undef C2_OFF
#undef C2_ISSUE
#undef C2_WAIT
#undef C2_MM
#undef C2_STEP

        // ---- epilogue: C/D of 16x16x32 with the weights as A: row = 4 kq + i = channel 16 ng + 4 kq + i, col = lane & 15 =
        // pixel r of the group: output row 2 g + (r >> 3), column r & 7.  A pixel is [32 hi | 32 lo] halves: the quad's hi halves
        // are 8 bytes at 2 c0, its lo halves 8 bytes at 64 + 2 c0.
        const bool interior = oy0 + TH <= p.Ho && ox0 + TW <= p.Wo;   // uniform; every tile of a 128x416 / 256x832 frame
        stores_counted = interior;
        const int c0 = 16 * ng + 4 * kq;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int oy = oy0 + 2 * (g0 + g) + (r >> 3), ox = ox0 + (r & 7);
            const bool ok = interior || (oy < p.Ho && ox < p.Wo);
            uint8_t* __restrict__ o = p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * 128 + c0 * 2;
            float vm = 0.f;
            uint2 hw, lw;
            split_pair_relu(acc[g][0], acc[g][1], p.out_scale, vm, hw.x, lw.x);
            split_pair_relu(acc[g][2], acc[g][3], p.out_scale, vm, hw.y, lw.y);
            if (ok) {
                vmax = fmaxf(vmax, vm);
                if (!CP_DBG(2)) {
                    *reinterpret_cast<uint2*>(o) = hw;
                    *reinterpret_cast<uint2*>(o + 64) = lw;
                }
            }
        }
        t = tnext;
        tc = tcn;
        buf ^= 1;
    }
    if (p.range) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        range_note(p.range, vmax, lane == 0);
    }
}

// ---- cnv3 (3x3, dilation 2, 32 -> 64 channels) from an LDS patch ----------------------------------
// As an implicit GEMM cnv3 has 9 chunks of 24 matrix instructions per wave between barriers: 0.046 ms at B=32 of which
// 0.004 is matrix time, 0.025 the barrier / wait skeleton and 0.015 the stores all CUs issue together.  Same recipe as
// conv_patch_cnv2_h3: the 12 x 12-pixel input patch of an 8 x 8 output tile is staged once (18 KB, double-buffered), the
// A fragments come straight from it (K = 32 = one tap x 32 channels: a lane's fragment is channel quarter lane>>4 of one
// pixel), wave w owns output channels 16 w .. +15 with its 18 weight fragments in registers, one barrier per tile.
// Patch layout: eight regions [plane hi|lo][channel quarter], each [patch row][12 pixels] x 16 B.  A pixel group is two
// output rows TWO apart x 8 columns (rows {0,2}, {1,3}, {4,6}, {5,7} of the tile): lanes 0-7 read 8 consecutive units,
// lanes 8-15 the same units 384 B = 128 mod 256 further on; regions are multiples of 256 B: conflict-free.
__global__ __launch_bounds__(cp3::THREADS, 3) void conv_patch_cnv3_h3(ConvPatchParams p) {
    using namespace cp3;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_p3[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // = N group: channels 16 wave .. +15
    const int r = lane & 15, kq = lane >> 4;
    const float bv = p.bias[16 * wave + r] * p.bias_scale;              // C/D layout: col = lane & 15 = channel
    // A fragment of group g, tap (ky, kx), plane pl: region (pl, kq), patch row (4 (g >> 1) + (g & 1) + 2 (r >> 3)) + 2 ky,
    // pixel (r & 7) + 2 kx
    const int a_lane = kq * REGION + (r >> 3) * (2 * ROWB) + (r & 7) * 16;

    // piece k, lane l = linear 16-byte unit 64 k + l of the patch [region][row][pixel]; the waves take pieces k = wave, wave + 4, ...
    auto issue_patch = [&](const TileCoord& tc, int buf) {
        const int iy_base = tc.ty * TH - p.pad_t, ix_base = tc.tx * TW - p.pad_l;
        const uint8_t* xin = p.x + (size_t)tc.n * p.H * p.W * 128;
        uint8_t* dst = smem_p3 + buf * PATCH;
#pragma unroll
        for (int kk = 0; kk < (NDMA + 3) / 4; ++kk) {
            const int k = wave + 4 * kk;
            if (k < NDMA) {
                const int L = k * 64 + lane;
                const int reg = L / (PH * PW), rem = L - reg * (PH * PW);
                const int py = rem / PW, px = rem - py * PW;
                const int iy = iy_base + py, ix = ix_base + px;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && !CP_DBG(1);
                const uint8_t* src = patch_src(ok, xin, (unsigned)(iy * p.W + ix) * 128u + (reg >> 2) * 64 + (reg & 3) * 16, p.zeros);
                __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + k * 1024), 16, 0, 0);
            }
        }
    };

    const TileWalk tw = tile_walk(p.ntiles);
    int t = tw.first, buf = 0;
    float vmax = 0.f;
    bool stores_counted = false;       // the previous tile issued exactly 16 stores per lane after this tile's patch DMA (interior tile)
    TileCoord tc = tile_coord(t, p.tiles_x, p.tiles_y);
    const TileCoord ts = tile_coord(tw.step, p.tiles_x, p.tiles_y);
    if (t < tw.end) issue_patch(tc, 0);
    // the wave's 18 weight fragments (18 KB) are fetched behind the first patch's DMA
    half8 wreg[STEPS][2];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        wreg[st][0] = *reinterpret_cast<const half8*>(p.w + ((size_t)((st * 4 + wave) * 2 + 0) * 64 + lane) * 16);
        wreg[st][1] = *reinterpret_cast<const half8*>(p.w + ((size_t)((st * 4 + wave) * 2 + 1) * 64 + lane) * 16);
    }
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));         // see conv_patch_cnv1_h3: no compiler-placed vmcnt wait inside the tile loop
    while (t < tw.end) {
        const int n = tc.n;
        const int oy0 = tc.ty * TH, ox0 = tc.tx * TW;
        // this tile's patch has landed (the previous tile's stores are younger than its DMA and may stay in flight);
        // behind the barrier every wave has also left the previous tile's matrix phase, so the other buffer may be
        // refilled: the next tile's patch flies under this tile's matrix phase and stores
        if (stores_counted && !CP_DBG(2)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tnext = t + tw.step;
        const TileCoord tcn = tile_next(tc, ts, p.tiles_x, p.tiles_y);
        if (tnext < tw.end) issue_patch(tcn, buf ^ 1);

        // Matrix phase, software-pipelined by hand as in conv_patch_cnv2_h3, in half-steps of two pixel groups: the four
        // fragments of half-step h+1 are requested before the six MFMAs of half-step h are queued (a ring of 2 x 4 fragment
        // registers: three waves per SIMD need <= 168 registers, 72 of them hold weights).
        const unsigned a_u32 = lds_u32(smem_p3 + buf * PATCH + a_lane);
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{bv, bv, bv, bv};
        half8 fh[2][2], fl[2][2];                              // [ring slot][group of the pair]
#define C3_RD(dst_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(a_u32), "n"(off_) : "memory")
        // half-step H_ = (step H_ >> 1, group pair H_ & 1): groups 2 (H_ & 1) and 2 (H_ & 1) + 1 = tile rows 4 (H_ & 1) + {0,2} / {1,3}
#define C3_OFF(H_, G_, PL_) ((((H_) >> 1) / 3) * (RATE * ROWB) + (((H_) >> 1) % 3) * (RATE * 16) + (4 * ((H_) & 1) + (G_)) * ROWB + (PL_) * (4 * REGION))
#define C3_ISSUE(H_, B_)                                                                           \
        { C3_RD(fh[B_][0], C3_OFF(H_, 0, 0)); C3_RD(fh[B_][1], C3_OFF(H_, 1, 0));                  \
          C3_RD(fl[B_][0], C3_OFF(H_, 0, 1)); C3_RD(fl[B_][1], C3_OFF(H_, 1, 1)); }
#define C3_WAIT(N_, B_) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fh[B_][0]), "+v"(fh[B_][1]), "+v"(fl[B_][0]), "+v"(fl[B_][1]) : "n"(N_))
#define C3_MM(H_, B_)                                                                              \
        {   const half8 bh = wreg[(H_) >> 1][0], bl = wreg[(H_) >> 1][1];                          \
            constexpr int g_ = 2 * ((H_) & 1);                                                     \
            acc[g_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[B_][0], bh, acc[g_], 0, 0, 0);     \
            acc[g_ + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[B_][1], bh, acc[g_ + 1], 0, 0, 0); \
            acc[g_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[B_][0], bl, acc[g_], 0, 0, 0);     \
            acc[g_ + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[B_][1], bl, acc[g_ + 1], 0, 0, 0); \
            acc[g_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[B_][0], bh, acc[g_], 0, 0, 0);     \
            acc[g_ + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[B_][1], bh, acc[g_ + 1], 0, 0, 0); }
#define C3_HALF(H_)                                                                                \
        {   if constexpr ((H_) + 1 < 2 * STEPS) { C3_ISSUE((H_) + 1, ((H_) + 1) & 1) C3_WAIT(4, (H_) & 1); }   \
            else { C3_WAIT(0, (H_) & 1); }                                                         \
            C3_MM(H_, (H_) & 1)                                                                    \
            __builtin_amdgcn_sched_barrier(0); }
        C3_ISSUE(0, 0)
        C3_HALF(0) C3_HALF(1) C3_HALF(2) C3_HALF(3) C3_HALF(4) C3_HALF(5) C3_HALF(6) C3_HALF(7) C3_HALF(8)
        C3_HALF(9) C3_HALF(10) C3_HALF(11) C3_HALF(12) C3_HALF(13) C3_HALF(14) C3_HALF(15) C3_HALF(16) C3_HALF(17)
        static_assert(STEPS == 9, "unrolled by hand");
#undef C3_RD
#undef C3_OFF
#undef C3_ISSUE
#undef C3_WAIT
#undef C3_MM
#undef C3_HALF

        // ---- epilogue: C/D of 16x16x32: col = lane & 15 (channel 16 wave + r), row = 4 kq + i = pixel of the group:
        // output row 4 (g >> 1) + (g & 1) + 2 (kq >> 1), column 4 (kq & 1) + i.  The 64 output channels are two blocks of
        // [32 hi | 32 lo] halves; even lanes store the hi halves of channels (c, c+1), odd lanes the lo halves of (c-1, c)
        const bool odd = r & 1;
        const unsigned sel = odd ? 0x03020706u : 0x05040100u;
        const int cfull = 16 * wave + r, cin = cfull & 31;
        const int choff = (cfull >> 5) * 128 + (odd ? 64 + (cin - 1) * 2 : cin * 2);
        const bool interior = oy0 + TH <= p.Ho && ox0 + TW <= p.Wo;   // uniform; every tile of a 128x416 / 256x832 frame
        stores_counted = interior;
        if (interior) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int oy = oy0 + 4 * (g >> 1) + (g & 1) + 2 * (kq >> 1);
                uint8_t* __restrict__ orow = p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox0 + 4 * (kq & 1)) * 256 + choff;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = fmaxf(acc[g][i] * p.out_scale, 0.f);
                    vmax = fmaxf(vmax, v);
                    v = fminf(v, 65504.f);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const unsigned x = (unsigned)__builtin_bit_cast(unsigned short, hi) |
                                       ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
                    const unsigned xn = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                    if (!CP_DBG(2)) *reinterpret_cast<unsigned*>(orow + i * 256) = __builtin_amdgcn_perm(xn, x, sel);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int oy = oy0 + 4 * (g >> 1) + (g & 1) + 2 * (kq >> 1);
                if (oy >= p.Ho) continue;
                const int oxb = ox0 + 4 * (kq & 1);
                uint8_t* __restrict__ orow = p.y + (((size_t)n * p.Ho + oy) * p.Wo + oxb) * 256 + choff;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = fmaxf(acc[g][i] * p.out_scale, 0.f);
                    const bool ok = oxb + i < p.Wo;
                    if (ok) vmax = fmaxf(vmax, v);
                    v = fminf(v, 65504.f);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const unsigned x = (unsigned)__builtin_bit_cast(unsigned short, hi) |
                                       ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
                    const unsigned xn = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
                    if (ok) *reinterpret_cast<unsigned*>(orow + i * 256) = __builtin_amdgcn_perm(xn, x, sel);
                }
            }
        }
        t = tnext;
        tc = tcn;
        buf ^= 1;
    }
    if (p.range) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        range_note(p.range, vmax, lane == 0);
    }
}

#undef CP_DBG

}  // namespace davo
