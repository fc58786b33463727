// conv_igemm_h3.h — the same implicit-GEMM convolution as conv_igemm.h, on the fp16 matrix
// cores with float32-grade accuracy ("f16x3").
//
// FP32 MFMA runs at 1/16 of the fp16 rate on gfx950 and there is no xf32/TF32 path, so every
// float32 operand is split into two fp16 numbers
//
//     a = ah + al,   ah = fp16(a),   al = fp16(a - ah)                     (22 significant bits)
//
// and a product is evaluated as three fp16 MFMA products accumulated in ONE float32 accumulator:
//
//     a*b  ~=  ah*bh + ah*bl + al*bh                              (the dropped al*bl term is 2^-22)
//
// Each fp16 product is exact in float32 (11 x 11 bits), so the only loss is the representation
// error of the operands.  The residual al is ~2^-11 |a| and may be an fp16 subnormal; that costs
// absolute, not relative, accuracy (<= 2^-25 per operand), which is harmless for activations.
// Weights are small (|w| ~ 0.05), so each layer's weights are pre-multiplied by a power of two
// (max|w| -> [128,256), exact) before the split, which keeps their residuals normal, and the
// epilogue multiplies the accumulator by the inverse power of two.  Measured end to end the 6-DoF
// outputs differ from the float64 restatement by ~1e-7 relative, like the FP32-MFMA path, at
// 3 MFMA passes of 32 cycles per K=16 instead of 8 passes of 64 cycles (5.3x less matrix-pipe time).
//
// Activations are STORED in the split form by the producing kernel's epilogue — 4 bytes per
// element, the same HBM bytes as float32 — in a channel-blocked layout: per pixel, per block of
// CB = min(C,32) channels, CB hi halves followed by CB lo halves.  A K-chunk of 32 k-elements
// (one tap x 32 channels, or 32/C taps when C < 32) is then one 128-byte LDS row
// [32 hi | 32 lo], read as four ds_read_b128 fragments per 32-row MFMA tile.  The k order is
// channel-block major, tap minor, so a workgroup revisits the same 32-channel slice of its
// pixels for all taps before moving on (L2 / L1 locality of the dilated gather).
//
// Range: a layer's stored activations carry an exact power-of-two scale 2^shift (davo_calibrate; 0 by default)
// so that their fp16 pairs stay inside [2^-11, 65504); out_scale / bias_scale fold it in, the storing
// epilogues record the largest value written (range monitor, include/davo_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_igemm.h"
#include "pose_tail.h"

// measurement-only switches of the kernels (ConvParamsH::dbg) exist in a -DDAVO_TUNING build only; the product
// library compiles them out
#ifdef DAVO_TUNING
#define H3_DBG(bit_) ((p.dbg & (bit_)) != 0)
#else
#define H3_DBG(bit_) false
#endif

#ifndef DAVO_H3_NDG8
#define DAVO_H3_NDG8 2          // 8-group chunks: matrix groups that carry the next chunk's DMA (2: four loads each, groups 1-2; 4 measured 0.5 % slower)
#endif

#ifndef DAVO_POSE_EXP
#define DAVO_POSE_EXP 0
#endif
namespace davo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef __attribute__((address_space(3))) const uint8_t lds_u8_t;
// 32-bit LDS address of a pointer into shared memory (operand of ds_read_*)
__device__ __forceinline__ unsigned lds_u32(const uint8_t* p) {
    return (unsigned)(unsigned long)(lds_u8_t*)p;
}

__device__ __forceinline__ half8 lds_frag(const uint8_t* p) {
    return *reinterpret_cast<const half8*>(p);
}

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// Staging is LDS-DMA only (DMA must be true): global_load_lds_dwordx4 straight into LDS, no staging
// registers, no ds_write pass.  An LDS-DMA wave-instruction writes 1 KiB linearly (lane i -> base +
// 16 i = 8 rows of 128 B), so rows cannot be padded; bank conflicts are avoided by XOR-swizzling the
// 16-byte unit index with (row>>1)&7 — applied to the per-lane SOURCE address here and to the
// fragment reads.  (A register-staged variant with rows padded to 144 B measured 10-15 % slower.)
// SMALLC: Cin < 32, a chunk spans several taps, so the tap (and its bounds test) differs per lane.
// With Cin >= 32 the tap is uniform over the workgroup and everything about a chunk except the
// image-bounds test is scalar: the loop then costs ~20 VALU instructions per 24 MFMAs.
// M16: use v_mfma_f32_16x16x32_f16 (K = 32 = the whole chunk per instruction) instead of 32x32x16.  Same LDS
// bytes and matrix cycles per chunk; under matrix-dense load the chip holds a higher clock on the 16x16 shape
// (MI355X_MICROARCH.md, DVFS give-back item 7), so the faster one is chosen by measurement.
// RATE > 0 ("XS", 3x3 stride-1 layers, 16x16x32 form): the three kx taps of a filter row share ONE staged pixel patch.
// In flattened pixel order the taps kx = 0, 1, 2 of a dilated 3x3 filter read the same pixels shifted by RATE, so
// per (channel block, ky) the workgroup stages BMH + 2*RATE consecutive pixels once (LDS row i = flattened pixel
// tile_first - RATE + i of input row y + (ky-1)*RATE) and the chunk of tap kx reads its A fragments RATE*kx rows
// further down: the pixel-operand DMA falls 2.8-2.95x (all DMA instructions and L2 -> LDS bytes by a third).
// Where the shift leaves the image row (x < RATE for kx = 0, x >= W - RATE for kx = 2) TF's zero padding is restored
// by AND-ing the fragment with a per-lane mask; vertical padding is the DMA's bounds test as before.  Products and
// their order per accumulator are unchanged: results are bit-identical to RATE = 0.
// The kernel body takes its workgroup id as parameters (wg_x of nwg_x, group wg_y) so that conv_igemm_h3_mainrem below can
// run two tile shapes of one layer in ONE launch; conv_igemm_h3 itself passes blockIdx / gridDim.
template <int KS, int STRIDE, int WM, int WN, int TM, int TN, int LAYER, bool DMA, bool SMALLC, bool M16 = false, int NSTG = 2, int RATE = 0>
__device__ __forceinline__ void conv_igemm_h3_body(const ConvParamsH& p, const int wg_x, const int nwg_x, const int wg_y) {
    using T = TileH<WM, WN, TM, TN, NSTG>;
    using TX = TileX<WM, WN, TM, TN, NSTG, (RATE > 0 ? RATE : 1)>;
    constexpr bool XS = RATE > 0;
    static_assert(!XS || (KS == 3 && STRIDE == 1 && M16 && !SMALLC && DMA), "shared-tap staging: 3x3 stride-1 layers, 16x16x32 form");
    constexpr int BMH = T::BMH, BNH = T::BNH;
    constexpr int ROWB = DMA ? 128 : LDB;              // LDS bytes per row
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_h[];
    constexpr int NSTAGE = DMA ? T::DMA_STAGES : 2;
    constexpr int A_BUFS = XS ? 2 : NSTAGE, A_ROWS = XS ? TX::PR : BMH;
    uint8_t* As = smem_h;                              // [NSTAGE][BMH][ROWB]   (XS: [2][PR][128] pixel patches)
    uint8_t* Bs = smem_h + A_BUFS * A_ROWS * ROWB;     // [NSTAGE][BNH][ROWB]
    uint8_t* xdummy = Bs + NSTAGE * BNH * ROWB;        // XS: 1 KB per wave, where the DMA slots past the patch's end land
    const unsigned xzero = lds_u32(xdummy + WM * WN * 1024);    // XS: 128 zero bytes (what a tap reads outside its image row)
    if constexpr (XS) {
        if (threadIdx.x < 8) *reinterpret_cast<float4*>(xdummy + WM * WN * 1024 + threadIdx.x * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;

#ifdef DAVO_TUNING
    // tuning build only (8192): the workgroups that fill a CU's 2nd, 3rd ... slot in the first round start late by
    // slot x dbg[23:16] x ~4 us (dbg[27:24] = slots per CU), so that co-resident workgroups run out of phase
    if (H3_DBG(8192)) {
        const int slot_i = blockIdx.x / 256, nslot_ = (p.dbg >> 24) & 0xf;
        if (slot_i >= 1 && slot_i < nslot_ && blockIdx.y == 0)
            for (int i = 0; i < slot_i * ((p.dbg >> 16) & 0xff); ++i) __builtin_amdgcn_s_sleep(127);
    }
    // 16384 / 32768: the same with "second slot" = odd workgroup of its XCD / odd workgroup id (first 512 workgroups)
    if (H3_DBG(16384 | 32768) && blockIdx.x < 512 && blockIdx.y == 0 && (H3_DBG(16384) ? ((blockIdx.x >> 3) & 1) : (blockIdx.x & 1)))
        for (int i = 0; i < ((p.dbg >> 16) & 0xff); ++i) __builtin_amdgcn_s_sleep(127);
#endif
    const int tile_i = xcd_remap(wg_x, nwg_x);
    const int tile = p.tile_order ? __builtin_amdgcn_readfirstlane(p.tile_order[tile_i]) : tile_i;
    const int ntile = tile % p.ntiles_n, mtile = p.mtile0 + tile / p.ntiles_n;
    const int grp = wg_y;
    const uint8_t* __restrict__ xg = p.x + p.x_boff + grp * p.g_x_boff;
    const uint8_t* __restrict__ wg = p.w + grp * p.g_w + (long)ntile * BNH * p.w_row_bytes;
    const float* __restrict__ bg = p.bias + grp * p.g_bias + ntile * BNH;
    // Filter rows that reach inside the image for at least one output pixel of this tile: [ky0, ky0 + nky).  The other rows
    // multiply zero padding only (dilation 8 on a 32-row map: the tiles of image rows 0..7 never see ky = 0, those of rows
    // 24..31 never ky = 2), so the whole workgroup walks the chunks of the valid rows alone: at 128x416 15 % of cnv5's and
    // 8 % of cnv4's chunks.  Sums lose terms that are exactly zero; the order of the others is unchanged.
    int ky0 = 0, nky = KS;
    if constexpr (KS == 3 && !SMALLC) {
        const FilterRows fr = valid_filter_rows(mtile * BMH, min(mtile * BMH + BMH, p.M) - 1, p.Hout, p.Wout, p.Hin, STRIDE, p.pad_t, p.rate);
        ky0 = __builtin_amdgcn_readfirstlane(fr.ky0);
        nky = __builtin_amdgcn_readfirstlane(fr.nky);
    }
    if constexpr (KS == 3 && !SMALLC && !XS) {
        // the plain ring's prologue puts DMA_STAGES - 1 chunks in flight: a tile left with fewer (split-K parts of one channel
        // block that keep a single filter row: 3 chunks against a ring of 6) walks all rows as before
        if ((p.nchunks / p.cpb) * KS * nky < T::DMA_STAGES - 1) { ky0 = 0; nky = KS; }
    }
    const int tq0 = (KS == 3 && !SMALLC) ? KS * ky0 : 0, tq1 = (KS == 3 && !SMALLC) ? KS * (ky0 + nky) : p.cpb;   // chunk (tap) range inside a channel block
    const int nch = (p.nchunks / p.cpb) * (tq1 - tq0);             // chunks this tile walks

    // ---- staging assignment: thread -> (row r0 + ROWS_PER_PASS*j, 16-byte unit u of the 128-byte row)
    // unit u: plane = u>>2 (0 = hi, 1 = lo), k-elements 8*(u&3) .. +7 of the chunk
    const int r0 = tid >> 3;
    const int u = DMA ? ((tid & 7) ^ ((r0 >> 1) & 7)) : (tid & 7);   // logical unit this thread fetches
    int iy0[T::A_LOADS], ix0[T::A_LOADS], pix0[T::A_LOADS];
#pragma unroll
    for (int j = 0; j < T::A_LOADS; ++j) {
        const int m = mtile * BMH + r0 + T::ROWS_PER_PASS * j;
        if (m < p.M) {
            const int hw = p.Hout * p.Wout;
            const int n = m / hw, rem = m - n * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            iy0[j] = oy * STRIDE - p.pad_t;
            ix0[j] = ox * STRIDE - p.pad_l;
            pix0[j] = n * p.Hin * p.Win;
        } else {
            iy0[j] = -(1 << 28);
            ix0[j] = 0;
            pix0[j] = 0;
        }
    }
    const int cb = 1 << p.cb_log2;
    const int e0 = (u & 3) * 8;                              // first k-element of this unit
    const int tap_in_chunk = e0 >> p.cb_log2;                // which of the chunk's taps it belongs to
    const int unit_boff = (u >> 2) * (cb * 2) + (e0 & (cb - 1)) * 2;   // plane + channel offset in block
    const int wave_u = __builtin_amdgcn_readfirstlane(wid);          // provably wave-uniform (LDS-DMA base -> M0)

    // ---- LDS-DMA issue of one chunk ----------------------------------------------------------
    // per-thread invariants: abase[j] = address of this thread's 16-byte unit at tap (0,0) of row j
    // (a virtual address when that tap is padding: only dereferenced when in bounds);
    // boff[j] = byte offset of its weight unit from the chunk's (uniform) weight base.
    const uint8_t* abase[T::A_LOADS];
    unsigned boff[T::B_LOADS];
#pragma unroll
    for (int j = 0; j < T::A_LOADS; ++j)
        abase[j] = xg + ((long)pix0[j] + (long)iy0[j] * p.Win + ix0[j]) * p.x_pix_bytes + unit_boff;
#pragma unroll
    for (int j = 0; j < T::B_LOADS; ++j)
        boff[j] = (unsigned)((r0 + T::ROWS_PER_PASS * j) * (int)p.w_row_bytes + u * 16);
    // XS invariants.  Patch slot j of this thread = LDS row r0 + ROWS_PER_PASS*j = flattened pixel mtile*BMH - RATE + row.
    constexpr int XSLOTS = XS ? TX::A_SLOTS : 1;
    int xpix[XSLOTS];                                        // pixel index (n, y - RATE, x) of the slot at ky = 0 (only used when valid)
    int xyv[XSLOTS];                                         // input row at ky = 0 (y - RATE), or far out of range: never valid
    unsigned xkeep = 0xffffu;                                // bit i: row group i keeps tap kx = 0 (x >= RATE); bit 8+i: keeps kx = 2
    const int xrow0 = wm * TM * 32 + (lane & 15);            // first fragment row of this lane inside the tile
    // The shared patch is read kx*RATE rows down, and under the (row >> 1) & 7 swizzle of the other LDS images a shift of
    // 2 (mod 4) rows puts two k-quarters of a ds_read_b128 service group on the same banks (dilation 2: 6 M conflict cycles
    // per cnv6 launch).  The patch therefore has its own swizzle, unit ^ (row & 6): conflict-free at every even shift.
    const int ux = (tid & 7) ^ (r0 & 6);
    const uint8_t* xgu = xg + ((ux >> 2) * (cb * 2) + (((ux & 3) * 8) & (cb - 1)) * 2);
    if constexpr (XS) {
        const int hw = p.Hout * p.Wout;
#pragma unroll
        for (int j = 0; j < XSLOTS; ++j) {
            const int row = r0 + T::ROWS_PER_PASS * j;
            const int f = mtile * BMH - RATE + row;
            const bool ok = row < TX::PR && f >= 0 && f < p.Mtot;
            const int fc = ok ? f : 0;
            const int n = fc / hw, rem = fc - n * hw;
            const int y = rem / p.Wout, x = rem - y * p.Wout;
            xpix[j] = (n * p.Hin + y - RATE) * p.Win + x;
            xyv[j] = ok ? y - RATE : -(1 << 28);
        }
        xkeep = 0;
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) {
            const int f = mtile * BMH + xrow0 + i * 16;
            const int x = f % p.Wout;
            if (x >= RATE) xkeep |= 1u << i;
            if (x < p.Wout - RATE) xkeep |= 1u << (8 + i);
        }
    }
    // A fragment byte offset (plane 0 = hi, 1 = lo) of tap kx from the patch base: row xrow0 + kx*RATE, unit swizzled by its row
#define H3_XFRAG(KX_, PL_) ((unsigned)((xrow_ + (KX_) * RATE) * 128 + ((((PL_) * 4 + (lane >> 4)) ^ ((xrow_ + (KX_) * RATE) & 6)) * 16)))
    // A fragment of row group I_ for tap KX_: the patch row, or the zero row for the lanes the tap carries out of the image row
#define H3_XRD(arr_, I_, base_, KX_)                                                               \
    {                                                                                              \
        if constexpr ((KX_) == 1) { H3_RD(arr_[I_], base_, (I_) * 16 * 128); }                     \
        else {                                                                                     \
            /* the zero row is read at the 16-byte unit the patch row would have been read at: the lanes of a service group \
               then cover the banks once, whichever of them read zeros (all at unit 0 they collided with the patch lanes:     \
               1.27 M conflict cycles per cnv6 launch) */                                                                     \
            const unsigned ad_ = (xkeep >> (((KX_) == 0 ? 0 : 8) + (I_))) & 1u ? (base_) + (I_) * 16 * 128 : (xzero | ((base_) & 112u)); \
            H3_RD(arr_[I_], ad_, 0);                                                               \
        }                                                                                          \
    }
    // patch slot j_ of the super-chunk (xcblk, xky) into patch buffer abuf_; a slot past the patch's end parks in xdummy
#define H3_XDMA_A(j_, abuf_)                                                                       \
    {                                                                                              \
        if constexpr ((j_) < XSLOTS) {                                                             \
            const int iy = xyv[j_] + xdy;                                                          \
            const bool ok = (unsigned)iy < (unsigned)p.Hin;                                        \
            uint8_t* dst_ = (8 * wave_u + (j_) * T::ROWS_PER_PASS) < TX::PR                        \
                                ? As + (abuf_) * TX::PR * 128 + ((j_) * T::ROWS_PER_PASS + 8 * wave_u) * 128 \
                                : xdummy + wave_u * 1024;                                          \
            const uint8_t* src_ = xgu + (((long)(xpix[j_] + xdp_)) << p.x_pix_log2) + xcoff;       \
            __builtin_amdgcn_global_load_lds((gptr_t*)(ok ? src_ : p.zeros), (lptr_t*)dst_, 16, 0, 0); \
        } else {                                                                                   \
            __builtin_amdgcn_global_load_lds((gptr_t*)p.zeros, (lptr_t*)(xdummy + wave_u * 1024), 16, 0, 0); \
        }                                                                                          \
    }
#define H3_DMA_A(j_)                                                                               \
    if constexpr (T::A_LOADS > j_) {                                                               \
        const int iy = iy0[j_] + dy, ix = ix0[j_] + dx;                                            \
        const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;          \
        __builtin_amdgcn_global_load_lds((gptr_t*)(ok ? abase[j_] + delta : p.zeros),              \
                                         (lptr_t*)(a_ + (j_ * T::ROWS_PER_PASS + 8 * wave_u) * 128), 16, 0, 0); \
    }
#define H3_DMA_B(j_)                                                                               \
    if constexpr (T::B_LOADS > j_)                                                                 \
        __builtin_amdgcn_global_load_lds((gptr_t*)(wq + boff[j_]),                                 \
                                         (lptr_t*)(b_ + (j_ * T::ROWS_PER_PASS + 8 * wave_u) * 128), 16, 0, 0);
    // (cblk, tq) walk the chunks in order: scalar counters instead of a division per chunk
#define H3_DMA_SETUP(q_, buf_)                                                                     \
        const int tap = SMALLC ? (dma_tq << p.tpc_log2) + tap_in_chunk : dma_tq;                   \
        const int ky = tap / KS, kx = tap - ky * KS;                                               \
        /* a chunk with nothing to fetch (padding tap; filler DMA of the last iteration) fails every bounds test */ \
        const bool tap_ok = (SMALLC ? tap < p.ntaps : true) && dma_on;                             \
        const int dy = tap_ok ? ky * p.rate : -(1 << 28), dx = kx * p.rate;                        \
        const long delta = ((long)dy * p.Win + dx) * p.x_pix_bytes + (long)dma_cblk * (cb * 4);    \
        uint8_t* a_ = As + (buf_) * BMH * 128;                                                     \
        uint8_t* b_ = Bs + (buf_) * BNH * 128;                                                     \
        const int rq_ = dma_cblk * p.cpb + dma_tq;          /* the chunk's index in the layer's weight rows */ \
        const uint8_t* wq = wg + (long)(rq_ < p.nchunks ? rq_ : p.nchunks - 1) * 128;
#define H3_DMA_ADVANCE if (++dma_tq == tq1) { dma_tq = tq0; ++dma_cblk; }
    // slot s of the 8 DMA issue slots of a chunk: 0..3 = A rows, 4..7 = B rows
#define H3_DMA_SLOT(s_)                                                                            \
    {                                                                                              \
        if constexpr ((s_) == 0) H3_DMA_A(0)                                                       \
        if constexpr ((s_) == 1) H3_DMA_A(1)                                                       \
        if constexpr ((s_) == 2) H3_DMA_A(2)                                                       \
        if constexpr ((s_) == 3) H3_DMA_A(3)                                                       \
        if constexpr ((s_) == 4) H3_DMA_B(0)                                                       \
        if constexpr ((s_) == 5) H3_DMA_B(1)                                                       \
        if constexpr ((s_) == 6) H3_DMA_B(2)                                                       \
        if constexpr ((s_) == 7) H3_DMA_B(3)                                                       \
    }
#define H3_DMA_CHUNK(q_, buf_)                                                                     \
    {                                                                                              \
        constexpr bool dma_on = true;                                                              \
        H3_DMA_SETUP(q_, buf_)                                                                     \
        H3_DMA_A(0) H3_DMA_A(1) H3_DMA_A(2) H3_DMA_A(3)                                            \
        H3_DMA_B(0) H3_DMA_B(1) H3_DMA_B(2) H3_DMA_B(3)                                            \
        H3_DMA_ADVANCE                                                                             \
    }
    int dma_cblk = 0, dma_tq = tq0;
    // one 32-k chunk = two K=16 MFMA steps; per step and output tile: hi*hi, hi*lo, lo*hi
#define H3_STEP32(buf_, s_)                                                                        \
    {                                                                                              \
        const uint8_t* a = As + (buf_) * BMH * ROWB + (wm * TM * 32 + li) * ROWB;                  \
        const uint8_t* b = Bs + (buf_) * BNH * ROWB + (wn * TN * 32 + li) * ROWB;                  \
        half8 ah[TM], al[TM], bh[TN], bl[TN];                                                      \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                           \
            ah[i] = lds_frag(a + i * 32 * ROWB + foff[0][s_]);                                     \
            al[i] = lds_frag(a + i * 32 * ROWB + foff[1][s_]);                                     \
        }                                                                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                           \
            bh[j] = lds_frag(b + j * 32 * ROWB + foff[0][s_]);                                     \
            bl[j] = lds_frag(b + j * 32 * ROWB + foff[1][s_]);                                     \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0); \
            }                                                                                      \
    }
    // 16x16x32 form: "step" s_ is the N half of the wave tile; the A fragments (whole chunk depth) are read
    // in step 0 and reused in step 1
#define H3_STEP16(buf_, s_)                                                                        \
    {                                                                                              \
        const uint8_t* a = As + (buf_) * BMH * ROWB + (wm * TM * 32 + l16) * ROWB;                 \
        const uint8_t* b = Bs + (buf_) * BNH * ROWB + (wn * TN * 32 + (s_) * TN * 16 + l16) * ROWB; \
        if ((s_) == 0) {                                                                           \
            _Pragma("unroll") for (int i = 0; i < 2 * TM; ++i) {                                   \
                a16h[i] = lds_frag(a + i * 16 * ROWB + foff16[0]);                                 \
                a16l[i] = lds_frag(a + i * 16 * ROWB + foff16[1]);                                 \
            }                                                                                      \
        }                                                                                          \
        half8 bh[TN], bl[TN];                                                                      \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                           \
            bh[j] = lds_frag(b + j * 16 * ROWB + foff16[0]);                                       \
            bl[j] = lds_frag(b + j * 16 * ROWB + foff16[1]);                                       \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 2 * TM; ++i)                                         \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
                f32x4& c_ = acc16[i][(s_) * TN + j];                                               \
                c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16h[i], bh[j], c_, 0, 0, 0);          \
                c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16h[i], bl[j], c_, 0, 0, 0);          \
                c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16l[i], bh[j], c_, 0, 0, 0);          \
            }                                                                                      \
    }
    // 16x16x32 form, whole chunk, software-pipelined by hand.  The A fragments (NI row groups, hi and lo) are
    // read once per chunk; the B fragments of column group j + 2 are requested before the MFMAs of group j are
    // queued, so their LDS latency hides behind 6*NI matrix instructions and only the first reads of a chunk
    // are exposed.  (Left to the compiler, the loop nest reads a whole half-step's fragments, waits for
    // lgkmcnt(0) and only then starts the matrix pipe - twice per chunk; with the reads written as builtins it
    // still waited for lgkmcnt(0) at every group.)  The reads are inline asm so that the waits can be counted:
    // LDS returns in order, so "lgkmcnt(n)" = everything but the newest n reads has landed; each wait names the
    // fragments it releases as in/out operands, which keeps the matrix instructions that use them behind it.
    // Per accumulator the products still arrive as hi*hi, hi*lo, lo*hi per chunk: bitwise the same sums as before.
#define H3_RD(dst_, addr_, off_) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(off_) : "memory")
#define H3_WAIT_B(n_, x_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x_) : "n"(n_))
#define H3_WAIT_A(n_, arr_)                                                                        \
    {                                                                                              \
        if constexpr (NI == 4)                                                                     \
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(arr_[0]), "+v"(arr_[1]), "+v"(arr_[2]), "+v"(arr_[3]) : "n"(n_)); \
        else                                                                                       \
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(arr_[0]), "+v"(arr_[1]) : "n"(n_));        \
    }
#define H3_MFMA_ROW(aarr_, bfrag_, J_)                                                             \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                 \
        acc16[i][J_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aarr_[i], bfrag_, acc16[i][J_], 0, 0, 0);
    // 12 (or 3*NI) matrix instructions with the group's DMA arithmetic threaded between them: one MFMA, then up to
    // NV_ vector/scalar instructions in its shadow, a DMA issue after every third MFMA
#define H3_ILV(NV_, VE_)                                                                           \
    _Pragma("unroll") for (int r_ = 0; r_ < 3 * NI; ++r_) {                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x006, NV_, 0);                                       \
        if (r_ % (VE_) == (VE_) - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);            \
    }
#define H3_GROUP(J_)                                                                               \
    if constexpr ((J_) < NJ) {                                                                     \
        if constexpr ((J_) + 2 < NJ) {                                                             \
            H3_RD(bh[(J_) + 2 < NJ ? (J_) + 2 : 0], b_h, ((J_) + 2) * 16 * ROWB);                  \
            H3_RD(bl[(J_) + 2 < NJ ? (J_) + 2 : 0], b_l, ((J_) + 2) * 16 * ROWB);                  \
        }                                                                                          \
        constexpr int after = 2 * ((J_) + 1 < NJ) + 2 * ((J_) + 2 < NJ);     /* B reads newer than group J_'s */ \
        if constexpr ((J_) == 0) {                                                                 \
            H3_WAIT_A(1 + NI + after, a16h)                                                        \
            H3_WAIT_B(1 + NI + after, bh[0]);                                                      \
            H3_MFMA_ROW(a16h, bh[0], 0)                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            H3_WAIT_B(NI + after, bl[0]);                                                          \
            H3_MFMA_ROW(a16h, bl[0], 0)                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            H3_WAIT_A(after, a16l)                                                                 \
            H3_MFMA_ROW(a16l, bh[0], 0)                                                            \
            if constexpr (D0 == 0) {                                                               \
                H3_DMA_SLOT(0) H3_DMA_SLOT(1) H3_DMA_SLOT(2) H3_DMA_SLOT(3)                        \
            }                                                                                      \
        } else {                                                                                   \
            H3_WAIT_B(after, bh[J_]);                                                              \
            H3_WAIT_B(after, bl[J_]);                                                              \
            constexpr bool dma_here = (J_) >= D0 && (J_) < D0 + NDG;                               \
            if constexpr (dma_here) {                /* this group's share of the next chunk's DMA */ \
                H3_DMA_SLOT(((J_) - D0) * DPG) H3_DMA_SLOT(((J_) - D0) * DPG + 1)                  \
                if constexpr (DPG >= 4) { H3_DMA_SLOT(((J_) - D0) * DPG + 2) H3_DMA_SLOT(((J_) - D0) * DPG + 3) } \
                if constexpr (DPG == 8) {                                                          \
                    H3_DMA_SLOT(4) H3_DMA_SLOT(5) H3_DMA_SLOT(6) H3_DMA_SLOT(7)                    \
                }                                                                                  \
            }                                                                                      \
            H3_MFMA_ROW(a16h, bh[J_], J_)                                                          \
            H3_MFMA_ROW(a16h, bl[J_], J_)                                                          \
            H3_MFMA_ROW(a16l, bh[J_], J_)                                                          \
            if constexpr (dma_here) H3_ILV(DPG == 2 ? 2 : (DPG == 4 ? 4 : 6), DPG == 8 ? 1 : 3)    \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
#define H3_CHUNK16(buf_)                                                                           \
    {                                                                                              \
        constexpr int NI = 2 * TM, NJ = 2 * TN;                                                    \
        static_assert((NI == 2 || NI == 4) && NJ >= 2 && NJ <= 8, "wave tile shape");              \
        /* DMA issue slots: groups D0 .. D0+NDG-1, DPG slots each, all in the first half of the chunk */ \
        constexpr int D0 = NJ >= 4 ? 1 : 0, NDG = NJ >= 8 ? DAVO_H3_NDG8 : 2, DPG = 8 / NDG;                  \
        const unsigned a0 = lds_u32(As + (buf_) * BMH * ROWB + (wm * TM * 32 + l16) * ROWB);       \
        const unsigned b0 = lds_u32(Bs + (buf_) * BNH * ROWB + (wn * TN * 32 + l16) * ROWB);       \
        const unsigned a_h = a0 + foff16[0], a_l = a0 + foff16[1], b_h = b0 + foff16[0], b_l = b0 + foff16[1]; \
        half8 bh[NJ], bl[NJ];                                                                      \
        H3_RD(a16h[0], a_h, 0);                                                                    \
        H3_RD(a16h[1], a_h, 16 * ROWB);                                                            \
        if constexpr (NI == 4) { H3_RD(a16h[NI - 2], a_h, 32 * ROWB); H3_RD(a16h[NI - 1], a_h, 48 * ROWB); } \
        H3_RD(bh[0], b_h, 0);                                                                      \
        H3_RD(bl[0], b_l, 0);                                                                      \
        H3_RD(a16l[0], a_l, 0);                                                                    \
        H3_RD(a16l[1], a_l, 16 * ROWB);                                                            \
        if constexpr (NI == 4) { H3_RD(a16l[NI - 2], a_l, 32 * ROWB); H3_RD(a16l[NI - 1], a_l, 48 * ROWB); } \
        H3_RD(bh[1], b_h, 16 * ROWB);                                                              \
        H3_RD(bl[1], b_l, 16 * ROWB);                                                              \
        /* the scalar walk to the next chunk (tap, channel block, addresses) runs behind the reads it does not feed */ \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        H3_DMA_SETUP(q + NST - 1, nslot)                                                           \
        DAVO_PRIO_UP(DAVO_MMPRIO_H3);                                                              \
        H3_GROUP(0) H3_GROUP(1) H3_GROUP(2) H3_GROUP(3) H3_GROUP(4) H3_GROUP(5) H3_GROUP(6) H3_GROUP(7)    \
        DAVO_PRIO_DOWN(DAVO_MMPRIO_H3);                                                            \
        H3_DMA_ADVANCE                                                                             \
    }
    // ---- XS: the same hand-scheduled chunk with the A fragments read RATE*kx rows down the shared patch -------------
    // DMA slot s of a chunk: the chunk's share of the next patch first, then the next weight chunk
#define H3_XSLOT(s_, KX_, nabuf_)                                                                  \
    {                                                                                              \
        if constexpr ((s_) < TX::APC) { H3_XDMA_A(KX_ * TX::APC + (s_), nabuf_) }                  \
        else if constexpr ((s_) - TX::APC == 0) { H3_DMA_B(0) }                                    \
        else if constexpr ((s_) - TX::APC == 1) { H3_DMA_B(1) }                                    \
        else if constexpr ((s_) - TX::APC == 2) { H3_DMA_B(2) }                                    \
        else if constexpr ((s_) - TX::APC == 3) { H3_DMA_B(3) }                                    \
    }
#define H3_XSLOTS_OF_GROUP(g_, KX_, nabuf_)                                                                     \
    {                                                                                                           \
        H3_XSLOT((g_) * DPGX, KX_, nabuf_)                                                                      \
        if constexpr (DPGX >= 2) H3_XSLOT((g_) * DPGX + 1, KX_, nabuf_)                                         \
        if constexpr (DPGX >= 3) H3_XSLOT((g_) * DPGX + 2, KX_, nabuf_)                                         \
    }
#define H3_GROUPX(J_, KX_, nabuf_)                                                                              \
    if constexpr ((J_) < NJ) {                                                                     \
        if constexpr ((J_) + 2 < NJ) {                                                             \
            H3_RD(bh[(J_) + 2 < NJ ? (J_) + 2 : 0], b_h, ((J_) + 2) * 16 * ROWB);                  \
            H3_RD(bl[(J_) + 2 < NJ ? (J_) + 2 : 0], b_l, ((J_) + 2) * 16 * ROWB);                  \
        }                                                                                          \
        constexpr int after = 2 * ((J_) + 1 < NJ) + 2 * ((J_) + 2 < NJ);     /* B reads newer than group J_'s */ \
        if constexpr ((J_) == 0) {                                                                 \
            H3_WAIT_A(1 + NI + after, a16h)                                                        \
            H3_WAIT_B(1 + NI + after, bh[0]);                                                      \
            H3_MFMA_ROW(a16h, bh[0], 0)                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            H3_WAIT_B(NI + after, bl[0]);                                                          \
            H3_MFMA_ROW(a16h, bl[0], 0)                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            H3_WAIT_A(after, a16l)                                                                 \
            H3_MFMA_ROW(a16l, bh[0], 0)                                                            \
            if constexpr (D0 == 0) H3_XSLOTS_OF_GROUP(0, KX_, nabuf_)                                         \
        } else {                                                                                   \
            H3_WAIT_B(after, bh[J_]);                                                              \
            H3_WAIT_B(after, bl[J_]);                                                              \
            constexpr bool dma_here = (J_) >= D0 && (J_) < D0 + 2;                                 \
            if constexpr (dma_here) H3_XSLOTS_OF_GROUP((J_) - D0, KX_, nabuf_)                                \
            H3_MFMA_ROW(a16h, bh[J_], J_)                                                          \
            H3_MFMA_ROW(a16h, bl[J_], J_)                                                          \
            H3_MFMA_ROW(a16l, bh[J_], J_)                                                          \
            if constexpr (dma_here) H3_ILV(4, (3 * NI) / DPGX)                                     \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
    // chunk of tap KX_ (compile time: the fragment addresses of the three taps live in registers), weights in ring slot
    // bslot_, patch buffer abuf_; its DMA slots fetch the next weight chunk into nslot and patch slots of the next
    // super-chunk into nabuf_
#define H3_CHUNK16X(KX_, bslot_, abuf_, nabuf_)                                                    \
    {                                                                                              \
        constexpr int NI = 2 * TM, NJ = 2 * TN;                                                    \
        constexpr int D0 = NJ >= 4 ? 1 : 0;                                                        \
        constexpr int NSL = TX::APC + T::B_LOADS, DPGX = (NSL + 1) / 2;                            \
        static_assert(DPGX <= 3 && NSL <= 6, "DMA slots per group");                               \
        /* per-chunk copies behind an empty asm: without them the compiler computes the three taps' fragment offsets and   \
           the five patch slots' 64-bit addresses once per super-chunk and keeps them live (16 registers: it spilled) */   \
        int xrow_ = xrow0, xdp_ = xdpix;                                                           \
        asm volatile("" : "+v"(xrow_), "+s"(xdp_));                                                \
        const unsigned pb_ = lds_u32(As + (abuf_) * TX::PR * 128);                                 \
        const unsigned b0 = lds_u32(Bs + (bslot_) * BNH * ROWB + (wn * TN * 32 + l16) * ROWB);     \
        const unsigned a_h = pb_ + H3_XFRAG(KX_, 0), a_l = pb_ + H3_XFRAG(KX_, 1), b_h = b0 + foff16[0], b_l = b0 + foff16[1]; \
        half8 bh[NJ], bl[NJ];                                                                      \
        /* where a shifted tap leaves the image row (kx = 0: x < RATE, kx = 2: x >= W - RATE) TF pads with zeros: those   \
           lanes read the zero row instead of the patch - no arithmetic on the fragments, no register held for it */      \
        H3_XRD(a16h, 0, a_h, KX_) H3_XRD(a16h, 1, a_h, KX_)                                        \
        if constexpr (NI == 4) { H3_XRD(a16h, NI - 2, a_h, KX_) H3_XRD(a16h, NI - 1, a_h, KX_) }   \
        H3_RD(bh[0], b_h, 0);                                                                      \
        H3_RD(bl[0], b_l, 0);                                                                      \
        H3_XRD(a16l, 0, a_l, KX_) H3_XRD(a16l, 1, a_l, KX_)                                        \
        if constexpr (NI == 4) { H3_XRD(a16l, NI - 2, a_l, KX_) H3_XRD(a16l, NI - 1, a_l, KX_) }   \
        H3_RD(bh[1], b_h, 16 * ROWB);                                                              \
        H3_RD(bl[1], b_l, 16 * ROWB);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        uint8_t* b_ = Bs + nslot * BNH * 128;                                                      \
        const uint8_t* wq = wg + (long)h3_real_chunk(q + NST - 1 < nch ? q + NST - 1 : nch - 1, ky0, nky) * 128; \
        DAVO_PRIO_UP(DAVO_MMPRIO_H3);                                                              \
        H3_GROUPX(0, KX_, nabuf_) H3_GROUPX(1, KX_, nabuf_) H3_GROUPX(2, KX_, nabuf_) H3_GROUPX(3, KX_, nabuf_)    \
        H3_GROUPX(4, KX_, nabuf_) H3_GROUPX(5, KX_, nabuf_) H3_GROUPX(6, KX_, nabuf_) H3_GROUPX(7, KX_, nabuf_)    \
        DAVO_PRIO_DOWN(DAVO_MMPRIO_H3);                                                            \
    }
#define H3_STEP(buf_, s_)                                                                          \
    {                                                                                              \
        if constexpr (M16) H3_STEP16(buf_, s_) else H3_STEP32(buf_, s_)                            \
    }
#define H3_COMPUTE(buf_) { H3_STEP(buf_, 0) H3_STEP(buf_, 1) }

    // fragment byte offsets inside an LDS row for (plane, k-step): logical unit plane*4 + 2s + lh,
    // XOR-swizzled with (row>>1)&7 = (li>>1)&7 in the DMA layout (tile row bases are multiples of 32)
    int foff[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int unit = pl * 4 + 2 * s + lh;
            foff[pl][s] = (DMA ? (unit ^ ((li >> 1) & 7)) : unit) * 16;
        }

    // 16x16x32 operand map: lane l holds row/col l&15, k-elements 8*(l>>4).. of the 32-chunk = logical unit
    // plane*4 + (l>>4), swizzled with (row>>1)&7 = ((l&15)>>1)&7 (16-row tile bases are multiples of 16)
    const int l16 = lane & 15, q16 = lane >> 4;
    int foff16[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) foff16[pl] = ((pl * 4 + q16) ^ ((l16 >> 1) & 7)) * 16;
    half8 a16h[2 * TM], a16l[2 * TM];
    f32x4 acc16[2 * TM][2 * TN];

    // accumulators start at bias * bias_scale (exact: a power of two) so the epilogue issues no load
    // (see conv_igemm.h)
    f32x16 acc[TM][TN];
    if constexpr (M16) {
        const float inv = p.bias_scale;
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) {
            const float bv = bg[wn * TN * 32 + j * 16 + l16] * inv;
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc16[i][j][r] = bv;
        }
    } else {
        const float inv = p.bias_scale;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float bv = bg[wn * TN * 32 + j * 32 + li] * inv;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
        }
    }

    static_assert(DMA, "the register-staged variant was retired: LDS-DMA staging measured 10-15 % faster");
    {
        // LDS ring of NST slots, NST-1 chunks of LDS-DMA in flight.  The wait in front of each
        // barrier is a COUNTED vmcnt (everything but this thread's newest NST-2 chunks), the barrier
        // a raw s_barrier (__syncthreads() would always drain with vmcnt(0)), and a slot is read one
        // iteration after the wait + barrier that retired it.  The slot refilled in iteration q was
        // last read in iteration q-1, which every wave left through the previous barrier.
        constexpr int NST = T::DMA_STAGES;
        constexpr int NDMA = T::A_LOADS + T::B_LOADS;         // DMA instructions per thread per chunk
        constexpr int KEEP = (NST - 2) * NDMA;                // instructions allowed to stay in flight
        constexpr int WAIT_KEEP = (KEEP & 15) | (7 << 4) | (15 << 8) | ((KEEP >> 4) << 14);   // vmcnt(KEEP) only
        constexpr int WAIT_ALL = (7 << 4) | (15 << 8);                                        // vmcnt(0) only
        if constexpr (XS) {
            // Super-chunk sc = (channel block, ky) = three chunks kx = 0, 1, 2 on ONE pixel patch (buffer sc & 1); the next
            // super-chunk's patch arrives in the other buffer, its DMA slots spread over the three chunks, and every chunk
            // fetches the next weight chunk into the ring as before.  Patch slots past the super-chunks (and weight chunks
            // past the end) are issued all the same, reading the zero line: the per-chunk instruction count stays constant
            // for the counted wait, and no branch cuts the interleaved code.
            const int nsc = nch / 3;
            int xcblk = 0, xky = ky0;
            {
                const int xdy = ky0 * RATE, xdp_ = ky0 * RATE * p.Win, xcoff = 0;
                H3_XDMA_A(0, 0) H3_XDMA_A(1, 0) H3_XDMA_A(2, 0)
                if constexpr (XSLOTS > 3) H3_XDMA_A(3, 0)
                if constexpr (XSLOTS > 4) H3_XDMA_A(4, 0)
                static_assert(XSLOTS <= 5, "patch slots per thread");
                for (int c0 = 0; c0 < NST - 1 && c0 < nch; ++c0) {
                    uint8_t* b_ = Bs + c0 * BNH * 128;
                    const uint8_t* wq = wg + (long)h3_real_chunk(c0, ky0, nky) * 128;
                    H3_DMA_B(0) H3_DMA_B(1) H3_DMA_B(2) H3_DMA_B(3)
                }
            }
            __builtin_amdgcn_s_waitcnt(WAIT_ALL);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the zero row's ds_write
            __builtin_amdgcn_s_barrier();
            constexpr int XKEEP = (NST - 2) * (TX::APC + T::B_LOADS);
            constexpr int XWAIT_KEEP = (XKEEP & 15) | (7 << 4) | (15 << 8) | ((XKEEP >> 4) << 14);
            int slot = 0, q = 0;
#define H3_XSTEP(KX_)                                                                              \
            {                                                                                      \
                const int nslot = slot == 0 ? NST - 1 : slot - 1;                                  \
                H3_CHUNK16X(KX_, slot, abuf, nabuf)                                                \
                if (q + 1 < nch) {                                                                 \
                    /* the patch of the next super-chunk must have landed when its first chunk starts: drain after kx = 2 */ \
                    if (NST >= 3 && KX_ != 2) __builtin_amdgcn_s_waitcnt(XWAIT_KEEP);              \
                    else __builtin_amdgcn_s_waitcnt(WAIT_ALL);                                     \
                    __builtin_amdgcn_s_barrier();                                                  \
                }                                                                                  \
                slot = slot == NST - 1 ? 0 : slot + 1;                                             \
                ++q;                                                                               \
            }
            for (int sc = 0; sc < nsc; ++sc) {
                if (++xky == ky0 + nky) { xky = ky0; ++xcblk; }       // the super-chunk whose patch this one fetches
                const int xdy = sc + 1 < nsc ? xky * RATE : -(1 << 28);
                const int xdpix = xky * RATE * p.Win, xcoff = xcblk * 128;
                const int abuf = sc & 1, nabuf = abuf ^ 1;
                H3_XSTEP(0) H3_XSTEP(1) H3_XSTEP(2)
            }
#undef H3_XSTEP
            __builtin_amdgcn_s_waitcnt(WAIT_ALL);
        } else {
        H3_DMA_CHUNK(0, 0)
        if constexpr (NST >= 3) {
            static_assert(NST <= 6, "ring depth");
            if (nch >= NST - 1) {
                H3_DMA_CHUNK(1, 1)
                if constexpr (NST >= 4) H3_DMA_CHUNK(2, 2)
                if constexpr (NST >= 5) H3_DMA_CHUNK(3, 3)
                if constexpr (NST >= 6) H3_DMA_CHUNK(4, 4)
            }
        }
        if (NST >= 3 && nch >= NST - 1) __builtin_amdgcn_s_waitcnt(WAIT_KEEP);
        else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
        __builtin_amdgcn_s_barrier();
        int slot = 0;                                         // slot of chunk q
        for (int q = 0; q < nch; ++q) {
            const int nslot = slot == 0 ? NST - 1 : slot - 1; // (q + NST - 1) % NST
            const bool more = q + NST - 1 < nch;
            if constexpr (M16) {
                // the next chunk's DMA is issued from inside the matrix groups (H3_GROUP), two or four instructions
                // per group over the first half of the chunk: its address arithmetic runs in the shadow of queued
                // MFMAs instead of in front of them, and every load still has half a chunk to land
                // (the last chunk has nothing to prefetch: its slots re-load the last weight chunk and zero rows
                // into the idle ring slot instead of branching around the interleaved code)
                const bool dma_on = more && !H3_DBG(1);
                if (!H3_DBG(2)) H3_CHUNK16(slot)
                else if (dma_on) H3_DMA_CHUNK(q + NST - 1, nslot)
            } else {
                // Stagger (32x32x16 form): the two waves that share a SIMD would otherwise run the same phases in
                // lockstep; the second half of the workgroup issues its DMA after its first k-step
                const bool late_dma = H3_DBG(4) ? false : (WM * WN >= 8 && wave_u >= (WM * WN) / 2);
                if (more && !late_dma && !H3_DBG(1)) H3_DMA_CHUNK(q + NST - 1, nslot)
                if (!H3_DBG(2)) H3_STEP32(slot, 0)
                if (more && late_dma && !H3_DBG(1)) H3_DMA_CHUNK(q + NST - 1, nslot)
                if (!H3_DBG(2)) H3_STEP32(slot, 1)
            }
            if (q + 1 < nch) {
                // M16: the filler DMA keeps the count of younger instructions constant, so the counted wait always holds
                if (NST >= 3 && (more || M16) && nch >= NST - 1) __builtin_amdgcn_s_waitcnt(WAIT_KEEP);
                else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
                __builtin_amdgcn_s_barrier();
            }
            slot = slot == NST - 1 ? 0 : slot + 1;
        }
        if constexpr (M16) __builtin_amdgcn_s_waitcnt(WAIT_ALL);   // the last chunk's filler DMA must land before LDS is reused
        }
    }

    if (H3_DBG(32)) return;                                       // measurement only: no epilogue
    // ---- epilogues.  Both accumulator layouts are walked through the same three helpers:
    //   column group jj -> column inside the wave tile; (row group ii, register r) -> row inside the wave tile
    constexpr int NCG = M16 ? 2 * TN : TN, NRG = M16 ? 2 * TM : TM, NREG = M16 ? 4 : 16;
    auto col_of = [&](int jj) { return M16 ? jj * 16 + l16 : jj * 32 + li; };
    auto row_of = [&](int ii, int r) { return M16 ? ii * 16 + 4 * q16 + r : ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh; };
    auto val_of = [&](int ii, int jj, int r) -> float {
        if constexpr (M16) return acc16[ii][jj][r];
        else return acc[ii][jj][r];
    };

    // pose head fused (y_mode 2): pred is 1x1 linear and the spatial mean is linear
    // (nets/posenn.py:240-241), so a tile only has to deliver sum_rows sum_cols relu(x) * Wpred[col][k],
    // split by image.  Fixed summation order -> bitwise reproducible; pose_from_tiles adds the tiles.
    if (p.y_mode == 2) {
        const int row0 = mtile * BMH;
        const int img0 = row0 / p.pose_P;
        const int split_row = (img0 + 1) * p.pose_P - row0;          // tile rows >= split_row: next image
        float q[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef DAVO_POSE_DEBUG
        float dbg_v[12];
#endif
#pragma unroll
        for (int jj = 0; jj < NCG; ++jj) {
            const int n = ntile * BNH + wn * TN * 32 + col_of(jj);
            const float* wp = p.pose_w + ((long)grp * p.Cout + (n < p.Cout ? n : 0)) * 3;
            const float w0 = n < p.Cout ? wp[0] : 0.f, w1 = n < p.Cout ? wp[1] : 0.f, w2 = n < p.Cout ? wp[2] : 0.f;
            float s0 = 0.f, s1 = 0.f;
            // out_scale is a positive power of two: max(x*s, 0) = s*max(x, 0) exactly, so the sums run on the raw
            // accumulators and are scaled once.  A tile inside one image and inside M (uniform test) needs no
            // per-value row tests: two instructions per value.
            if (split_row >= BMH && row0 + BMH <= p.M) {
#pragma unroll
                for (int ii = 0; ii < NRG; ++ii)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) s0 += fmaxf(val_of(ii, jj, r), 0.f);
            } else {
#pragma unroll
                for (int ii = 0; ii < NRG; ++ii)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        const int row = wm * TM * 32 + row_of(ii, r);
                        float v = fmaxf(val_of(ii, jj, r), 0.f);
                        if (row0 + row >= p.M) v = 0.f;
                        if (row < split_row) s0 += v; else s1 += v;
                    }
            }
            s0 *= p.out_scale; s1 *= p.out_scale;
            // (No drain, no idle cycles here any more: the round-3 flaky sums were one instruction form, below.)
#if DAVO_POSE_EXP == 0
            q[0] += s0 * w0; q[1] += s0 * w1; q[2] += s0 * w2;
            q[3] += s1 * w0; q[4] += s1 * w1; q[5] += s1 * w2;
#else
            // Reproducer builds only (tools/build_variant.py _e6 -DDAVO_POSE_EXP=6; tools/exp/flake_count.py, flake_lanes.py;
            // DESIGN.md section 4; profiles/r04_flake*_variants.log).  The six products as the packed sequence hipcc's SLP
            // vectoriser formed here.  6: (q4, q5) by v_pk_fma_f32 ... op_sel:[0,1,0] - the LOW result lane takes the HIGH register
            // of the 64-bit src1 pair: with three or more waves per SIMD and neighbours keeping the matrix pipe busy the selected
            // operand sporadically reads as 0 in lanes 48-63 (the product vanishes, the lane returns its addend): 299 of 299
            // forwards wrong at B = 4.  8: the same products from an (s1, s1) pair, no select: 0 of 299.  9: the selecting
            // instruction with a constant-0 addend and scalar sums: 74 of 299.  10: the select on src0 (op_sel:[1,0,0]): 0 of 299.
            {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 sp = {s0, s1}, w01 = {w0, w1}, w20 = {w2, w0}, w12 = {w1, w2};
                f2 q01 = {q[0], q[1]}, q23 = {q[2], q[3]}, q45 = {q[4], q[5]};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(q01) : "v"(w01), "v"(sp));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q23) : "v"(w20), "v"(sp));
#if DAVO_POSE_EXP == 6
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(q45) : "v"(w12), "v"(sp));
#elif DAVO_POSE_EXP == 8
                { f2 shh = {s1, s1}; asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q45) : "v"(w12), "v"(shh)); }
#elif DAVO_POSE_EXP == 9
                { f2 t45; asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(t45) : "v"(w12), "v"(sp));
                  q45.x += t45.x; asm volatile("" : "+v"(q45)); q45.y += t45.y; asm volatile("" : "+v"(q45)); }
#elif DAVO_POSE_EXP == 10
                asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0]" : "+v"(q45) : "v"(w12), "v"(sp));
#else
#error "DAVO_POSE_EXP: 6, 8, 9 or 10"
#endif
                q[0] = q01.x; q[1] = q01.y; q[2] = q23.x; q[3] = q23.y; q[4] = q45.x; q[5] = q45.y;
            }
#endif
#ifdef DAVO_POSE_DEBUG
            if constexpr (NCG == 2) { if (jj == 0) { dbg_v[10] = q[4]; dbg_v[11] = q[5]; } dbg_v[jj * 5 + 0] = s0; dbg_v[jj * 5 + 1] = s1; dbg_v[jj * 5 + 2] = w0; dbg_v[jj * 5 + 3] = w1; dbg_v[jj * 5 + 4] = w2; }
#endif
        }
#ifdef DAVO_POSE_DEBUG
        // experiment build only (tools/exp/flake_lanes.py): every lane's operands and its six sums before the wave reduction
        if constexpr (NCG == 2) {
            float* d = p.pose_partial + 4l * 2 * p.pose_mt * p.ntiles_n * 6 +
                       ((((long)grp * p.pose_mt + (mtile - p.mtile0)) * p.ntiles_n + ntile) * (WM * WN * 64) + tid) * 20;
#pragma unroll
            for (int k = 0; k < 10; ++k) d[k] = dbg_v[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) d[10 + k] = q[k];
            d[16] = dbg_v[10]; d[17] = dbg_v[11];
        }
#endif
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                q[k] += __shfl_down(q[k], o, 64);
#if DAVO_POSE_EXP                                 /* reproducer builds: keep the butterfly scalar whatever the flags */
                asm volatile("" : "+v"(q[k]));
#endif
            }
        __syncthreads();                                             // every wave is done with the staging slots
        float* red = reinterpret_cast<float*>(smem_h);               // [waves][6]
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[wid * 6 + k] = q[k];
        __syncthreads();
        if (tid < 6) {
            float t = 0.f;
            for (int w = 0; w < WM * WN; ++w) t += red[w * 6 + tid];
            float* dst = p.pose_partial + (((long)grp * p.pose_mt + (mtile - p.mtile0)) * p.ntiles_n + ntile) * 6 + tid;
            if (p.pose_counter) { agent_store(dst, t); agent_stores_done(); }      // read by the launch's last workgroup
            else *dst = t;
        }
        // the workgroup that finishes last adds the tiles in fixed order and writes the poses (pose_tail.h)
        if (p.pose_counter && last_workgroup(p.pose_counter, (unsigned)p.pose_total, reinterpret_cast<unsigned*>(red + 64))) pose_from_tiles_tail<WM * WN * 64>(p);
        return;
    }

    // combine, bias (already in the accumulator), ReLU; store float32 or re-split for the next layer
    const int ocb_log2 = p.y_ld >= 32 ? 5 : (p.y_ld == 16 ? 4 : 3);
    const int ocb = 1 << ocb_log2;
    float vmax = 0.f;                                                // largest |stored value| of this lane (range monitor)
    const int mt_store = H3_DBG(64) ? (mtile & 255) : mtile;       // 64: measurement only, stores fold onto 256 tiles

    // Split store, interior tile (every row < M, every column < Cout: all but the last tile row of a launch).
    // The general loop below costs ~35 instructions and three branches per value (64-bit address products, bounds
    // masks, mode tests) - a fifth to a third of a layer's time.  Here: 32-bit offsets from a uniform tile base,
    // no bounds tests, ReLU folded into the lower clamp, the lane-pair exchange as one DPP move + one byte permute.
    if (p.y_mode == 1 && (mtile + 1) * BMH <= p.M && (ntile + 1) * BNH <= p.Cout) {
        uint8_t* __restrict__ tbase = p.y + (long)mt_store * BMH * p.y_ld * 4;
        const unsigned rowb = (unsigned)p.y_ld * 4u;
        const bool odd = lane & 1;
        // word = even lane: hi halves of channels (n, n+1); odd lane: lo halves of (n-1, n).  perm(xn, x, sel): bytes 0-3 = x
        const unsigned sel = odd ? 0x03020706u : 0x05040100u;
        const float lo_clamp = p.relu ? 0.f : -65504.f;
        // column byte offsets: the wave's first channel is a multiple of 32 (checked: blocks of 32 channels, 64 B of hi
        // halves then 64 B of lo halves), so column group jj sits a compile-time distance from group 0 and the
        // distance folds into the store's immediate offset
        const int ng0 = p.y_coff + grp * p.g_y_coff + ntile * BNH + wn * TN * 32 + col_of(0);
        const bool regular = ocb == 32 && ((ng0 - col_of(0)) & 31) == 0;
        const unsigned coff0 = (unsigned)((ng0 >> ocb_log2) * (ocb * 4) + (ng0 & (ocb - 1)) * 2 + (odd ? ocb * 2 - 2 : 0));
        unsigned coff[NCG];
#pragma unroll
        for (int jj = 0; jj < NCG; ++jj) {
            const int ng = ng0 - col_of(0) + col_of(jj);
            coff[jj] = (unsigned)((ng >> ocb_log2) * (ocb * 4) + (ng & (ocb - 1)) * 2 + (odd ? ocb * 2 - 2 : 0));
        }
        constexpr int CW = M16 ? 16 : 32;                      // channels per column group
/* experiment (-DDAVO_STORE_SC1): write-through stores (agent scope, global_store_dword ... sc1) leave no dirty lines for the
   end-of-kernel release to write back - measured 7 % slower end to end (cnv4 +23 us, cnv5 +37, cnv6 +25 at B = 32:
   profiles/r03_writethrough_stores_ab.log): the L2 merges the 64-byte pieces of a row that write-through sends out one by one */
#ifdef DAVO_STORE_SC1
#define H3_STORE_U32(ptr_, val_) __hip_atomic_store(reinterpret_cast<unsigned*>(ptr_), (val_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define H3_STORE_U32(ptr_, val_) (*reinterpret_cast<unsigned*>(ptr_) = (val_))
#endif
#define H3_STORE_ROWS(COFF_)                                                                        \
        _Pragma("unroll") for (int ii = 0; ii < NRG; ++ii)                                          \
            _Pragma("unroll") for (int r = 0; r < NREG; ++r) {                                      \
                uint8_t* __restrict__ rowp = tbase + ((unsigned)(wm * TM * 32 + row_of(ii, r)) * rowb + coff0); \
                _Pragma("unroll") for (int jj = 0; jj < NCG; ++jj) {                                \
                    float v = fmaxf(val_of(ii, jj, r) * p.out_scale, lo_clamp);                     \
                    vmax = fmaxf(vmax, fabsf(v));                                                   \
                    v = fminf(v, 65504.f);                              /* fp16 range; see DESIGN.md */ \
                    const _Float16 hi = (_Float16)v;                                                \
                    const _Float16 lo = (_Float16)(v - (float)hi);                                  \
                    const unsigned x = (unsigned)__builtin_bit_cast(unsigned short, hi) |           \
                                       ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);    \
                    const unsigned xn = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   /* quad_perm [1,0,3,2] */ \
                    H3_STORE_U32(rowp + (COFF_), __builtin_amdgcn_perm(xn, x, sel));                   \
                }                                                                                   \
            }
        if (regular) { H3_STORE_ROWS((jj * CW >> 5) * 128 + ((jj * CW) & 31) * 2) }
        else { H3_STORE_ROWS(coff[jj] - coff0) }
#undef H3_STORE_ROWS
#undef H3_STORE_U32
    } else {
#pragma unroll
    for (int jj = 0; jj < NCG; ++jj) {
        const int n = ntile * BNH + wn * TN * 32 + col_of(jj);
        const bool n_ok = n < p.Cout;
        const int ng = p.y_coff + grp * p.g_y_coff + n;              // channel in the output tensor
        const long cbyte = (long)(ng >> ocb_log2) * (ocb * 4) + (ng & (ocb - 1)) * 2;
#pragma unroll
        for (int ii = 0; ii < NRG; ++ii)
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const int m = mt_store * BMH + wm * TM * 32 + row_of(ii, r);
                float v = val_of(ii, jj, r) * p.out_scale;
                if (p.relu) v = fmaxf(v, 0.f);
                if (n_ok && m < p.M) {
                    if (p.y_mode == 0) {
                        reinterpret_cast<float*>(p.y)[(long)m * p.y_ld + ng] = v;
                    } else {
                        // split, then pair up with the neighbouring lane (= neighbouring channel) so that
                        // every lane issues ONE 4-byte store instead of two 2-byte ones: even lanes store
                        // the hi halves of channels (n, n+1), odd lanes the lo halves of (n-1, n)
                        vmax = fmaxf(vmax, fabsf(v));
                        v = fminf(fmaxf(v, -65504.f), 65504.f);     // fp16 range; see DESIGN.md
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        const unsigned x = (unsigned)__builtin_bit_cast(unsigned short, hi) |
                                           ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
                        const unsigned xn = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                        const bool odd = lane & 1;
                        const unsigned word = odd ? ((xn >> 16) | (x & 0xffff0000u)) : ((x & 0xffffu) | (xn << 16));
                        uint8_t* o = p.y + (long)m * p.y_ld * 4 + cbyte + (odd ? ocb * 2 - 2 : 0);
                        *reinterpret_cast<unsigned*>(o) = word;
                    }
                }
            }
    }
    }
    if (p.y_mode == 1 && p.range) {      // non-negative floats order like their bit patterns; inf = overflow
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        range_note(p.range, vmax, lane == 0);      // params.h: only a wave that would raise the record pays for the atomic
    }
    // split-K part with the fix-up folded in: the tile's last part adds the partial sums and writes the stored form (pose_tail.h)
    if (p.y_mode == 0 && p.sk_counter) splitk_tail<WM * WN * 64, BMH, BNH>(p, mtile, ntile, reinterpret_cast<unsigned*>(smem_h));
}

template <int KS, int STRIDE, int WM, int WN, int TM, int TN, int LAYER, bool DMA, bool SMALLC, bool M16 = false, int NSTG = 2, int RATE = 0>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN) >= 8 ? (WM * WN) / 4 : 2)
void conv_igemm_h3(ConvParamsH p) {
    conv_igemm_h3_body<KS, STRIDE, WM, WN, TM, TN, LAYER, DMA, SMALLC, M16, NSTG, RATE>(p, blockIdx.x, gridDim.x, blockIdx.y);
}

// Main launch (256x256 tiles, shared-tap staging) and remainder launch (128x128 tiles, three ring slots) of one layer as ONE
// grid.  Workgroups of a launch finish together, so every CU stores its tile at the same moment and nothing computes
// meanwhile (HISTORY.md round 2: a few per cent of cnv5 / cnv6); started half a round apart, the two halves of the chip would hide
// each other's burst, but a late start idles.  The remainder supplies the offset for free: it is a quarter round of short
// tiles, so half of the CUs take theirs FIRST and the other half LAST - every CU still runs three long tiles and one
// short one, the halves' bursts no longer coincide, and the boundary between the two launches is gone.
// The id order follows how the hardware hands workgroups out (tools/exp/dispatch_probe.hip: id % 8 = XCD; the workgroups of
// an XCD go to its four shader engines round-robin, strictly in order, and the queue WAITS when the engine whose turn it is
// has no free CU - so every engine must hold the same mix, or free CUs idle behind a blocked head: the first attempt, short
// and long alternating per workgroup, gave two engines only short tiles and ran 10 % slower).  Ids [0, 2 h): of every 64
// consecutive ids (8 per XCD = 2 per engine) the first 32 are remainder tiles, the last 32 main tiles (h = half of the
// remainder's workgroups, a multiple of 32): each engine's eight CUs start with four of each.  Then the rest of the main
// tiles, then the second half of the remainder.  Ordinals keep id % 8, so xcd_remap still gives each XCD a contiguous run
// of each kind.
// TNM: N extent of the main tile in 64-column units per wave pair: 4 = 256x256 (cnv5, cnv6), 2 = 256x128 (cnv4, 128 output channels).
// order 1 (round 3): the offset is taken per XCD instead of inside every XCD - the even XCDs run all their short tiles first, the
// odd XCDs all theirs last (x = id % 8 is the XCD, i = id / 8 its i-th workgroup in dispatch order).  The store bursts of the two
// halves of the chip still do not coincide (HBM is shared), but the CUs of one XCD stay in step on neighbouring tiles, so the
// halo rows one tile fetched are still in that XCD's L2 when its neighbour reads them.
template <int LAYER, int RATE, int TNM = 4>
__global__ __launch_bounds__(512, 2) void conv_igemm_h3_mainrem(ConvParamsH pm, ConvParamsH pr, int n_main, int n_rem, int order) {
    const int b = blockIdx.x, h = n_rem >> 1;
    int ord;
    bool rem;
    if (order == 1) {
        const int x = b & 7, i = b >> 3, ns = n_rem >> 3, nl = n_main >> 3;       // per XCD: ns short, nl long tiles
        if (x & 1) { rem = i >= nl; ord = ((rem ? i - nl : i) << 3) | x; }
        else { rem = i < ns; ord = ((rem ? i : i - ns) << 3) | x; }
    }
    else if (order == 2) { rem = b >= n_main; ord = rem ? b - n_main : b; }     // long tiles first (pm.tile_order), the short remainder tiles fill the end
    else if (b < 2 * h) { rem = ((b >> 5) & 1) == 0; ord = ((b >> 6) << 5) | (b & 31); }
    else if (b < h + n_main) { rem = false; ord = b - h; }
    else { rem = true; ord = b - n_main; }
    if (rem) conv_igemm_h3_body<3, 1, 4, 2, 1, 2, LAYER, true, false, true, 3, 0>(pr, ord, n_rem, 0);
    else conv_igemm_h3_body<3, 1, 4, 2, 2, TNM, LAYER, true, false, true, 2, RATE>(pm, ord, n_main, 0);
}

#undef H3_DBG
#undef H3_DMA_A
#undef H3_XDMA_A
#undef H3_XFRAG
#undef H3_XRD
#undef H3_XSLOT
#undef H3_XSLOTS_OF_GROUP
#undef H3_GROUPX
#undef H3_CHUNK16X
#undef H3_DMA_B
#undef H3_DMA_CHUNK
#undef H3_DMA_SLOT
#undef H3_DMA_SETUP
#undef H3_DMA_ADVANCE
#undef H3_COMPUTE
#undef H3_STEP
#undef H3_STEP16
#undef H3_CHUNK16
#undef H3_GROUP
#undef H3_ILV
#undef H3_MFMA_ROW
#undef H3_WAIT_A
#undef H3_WAIT_B
#undef H3_RD
#undef H3_STEP32

}  // namespace davo
