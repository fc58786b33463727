// conv_igemm_h3s.h — the f16x3 implicit-GEMM convolution (see conv_igemm_h3.h for the arithmetic,
// the split-fp16 activation layout and the LDS-DMA staging) on a 208-pixel x 256-channel tile.
//
// Why 208.  Workgroups of a launch take equal time, so a grid that is not a whole number of rounds
// of the 256 CUs pays a full last round.  The PoseNN feature maps are 32x104 = 16 x 208 pixels (cnv3..
// cnv6), 16x52 = 4 x 208 (cnv7) and 64x208 at 256x832 inputs, so with 208-pixel tiles a batch of B
// triplets is exactly 2B*16 (cnv5, cnv6) or 2B*4*2 (cnv7, two heads) tiles: B = 32 gives 4, 4 and 2 whole
// rounds where 256-pixel tiles gave 3.25, 3.25 and 1.625 (a remainder launch each, cnv7 a half-empty
// round).  A tile is also whole image rows, never two images.
//
// Shape of the work inside the tile.  13 = 208/16 is prime, so the eight waves split the CHANNELS, not
// the pixels: wave w owns output channels [32w, 32w+32) for all 208 pixels.  The weights are the MFMA A
// operand (two 16-channel groups, hi and lo fragments held for the whole chunk), the pixels the B operand
// (13 groups, fragments streamed from LDS two groups ahead of their use):
//     C[channel][pixel] += W[channel][k] * X[pixel][k]            v_mfma_f32_16x16x32_f16
// so a lane's four accumulator registers are four CONSECUTIVE CHANNELS of one pixel, and the split-fp16
// store of the epilogue is two 8-byte stores per register quad (hi halves, lo halves) with no lane exchange.
// Per accumulator the products still arrive as x_hi*w_hi, x_hi*w_lo, x_lo*w_hi per chunk, chunk after chunk.
//
// Per chunk and wave: 4 + 26 fragment reads for 78 matrix instructions (0.38 reads per MFMA against 0.25 on
// the 256x256 tile: every wave reads all pixels), still < 40 % of the LDS read rate at full matrix rate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_igemm_h3.h"

namespace davo {

#ifdef DAVO_TUNING
#define HS_DBG(bit_) ((p.dbg & (bit_)) != 0)
#else
#define HS_DBG(bit_) false
#endif

// NSA = ring slots of the PIXEL operand (the weights always have two).  3 for cnv7: its stride-2 gather misses L2 on the
// first touch of every 128-byte pixel slice (0.151 -> 0.140 ms with the input pinned in L2), so its pixel DMA is issued two
// chunks ahead; the weight DMA of chunk q+1 is issued BEFORE the pixel DMA of chunk q+2, so that "all but the newest
// four" (vmcnt counts in issue order) is exactly what chunk q+1 needs.
// WAVES = 4 (round 4): the 208 x 128 tile for cnv4 (128 output channels).  The per-wave code is the 208x256 tile's - a wave owns 32
// channels for all 208 pixels - with four waves, one per SIMD, one workgroup per CU (116 KB of LDS with three pixel ring slots), and
// 2B*16 tiles are whole rounds of the 256 CUs where 128-pixel tiles give 3.25.  Bit-identical to the 128x128 kernel
// (test_tile_208x128_forced_for_cnv4) and MEASURED SLOWER: cnv4 0.100 -> 0.108 ms at B = 32, 0.394 -> 0.408 at B = 128, level at
// B = 16 (profiles/r04_cnv4_208x128_ab.log) - one wave per SIMD issues its 11 DMA instructions and their address arithmetic into
// its own matrix stream, with no second wave to cover the chunk boundary.  Offered to the planner only with "tile_208x128" 1.
template <int KS, int STRIDE, int LAYER, int NSA = 2, int WAVES = 8>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void conv_igemm_h3s(ConvParamsH p) {
    using T = TileSW<WAVES>;
    constexpr int NP = T::NP, NAJ = T::NAJ, RPP = T::RPP;
    static_assert(NSA == 2 || NSA == 3, "pixel ring slots");
    static_assert(WAVES == 8 || NSA == 3, "the four-wave tile issues weights first, pixels two chunks ahead");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_s[];
    uint8_t* As = smem_s;                              // [NSA][208][128]  pixels
    uint8_t* Bs = smem_s + NSA * T::A_SLOT;            // [2][256][128]  weights
    uint8_t* dummy = smem_s + NSA * T::A_SLOT + 2 * T::B_SLOT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, q16 = lane >> 4;

#ifdef DAVO_TUNING
    // tuning build only: de-phase the rounds of a launch - half of the first round's workgroups (1024: by XCD parity,
    // 2048: half of every XCD's CUs) start dbg[23:16] x ~4 us late, so that the halves' store bursts no longer coincide
    if (HS_DBG(1024 | 2048) && blockIdx.x < 256 && blockIdx.y == 0 &&
        (HS_DBG(1024) ? (blockIdx.x & 1) : ((blockIdx.x >> 3) & 1))) {
        const int nsl = (p.dbg >> 16) & 0xff;
        for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = tile % p.ntiles_n, mtile = p.mtile0 + tile / p.ntiles_n;
    const int grp = blockIdx.y;
    const uint8_t* __restrict__ xg = p.x + p.x_boff + grp * p.g_x_boff;
    const uint8_t* __restrict__ wg = p.w + grp * p.g_w + (long)ntile * T::BN * p.w_row_bytes;
    const float* __restrict__ bg = p.bias + grp * p.g_bias + ntile * T::BN;

    // ---- staging assignment: thread -> rows r0 + 64 j, logical 16-byte unit u of the 128-byte row --------
    // pixel rows: j = 0..NAJ-2 for every wave, j = NAJ-1 (rows 192..207) for waves 0 and 1 only; weight rows: j = 0..3
    const int r0 = tid >> 3;
    const int u = (tid & 7) ^ ((r0 >> 1) & 7);
    const uint8_t* abase[NAJ];
    int iy0[NAJ], ix0[NAJ];
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
        const int row = r0 + RPP * j;
        const int m = mtile * T::BM + row;
        int pix0 = 0;
        if (row < T::BM && m < p.M) {
            const int hw = p.Hout * p.Wout;
            const int n = m / hw, rem = m - n * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            iy0[j] = oy * STRIDE - p.pad_t;
            ix0[j] = ox * STRIDE - p.pad_l;
            pix0 = HS_DBG(4096) ? 0 : n * p.Hin * p.Win;     // 4096, tuning build only: every tile reads image 0 (input stays in L2)
        } else {
            iy0[j] = -(1 << 28);                       // every tap out of bounds -> the zero line
            ix0[j] = 0;
        }
        const int unit_boff = (u >> 2) * 64 + (u & 3) * 16;          // plane (hi | lo) + 8 channels x 2 B
        abase[j] = xg + ((long)pix0 + (long)iy0[j] * p.Win + ix0[j]) * p.x_pix_bytes + unit_boff;
    }
    // Weight rows are staged in a permuted order: LDS row R = 32 w + 16 c + l (wave w, MFMA row group c, row l of the
    // group = 4 (lane>>4) + r) holds output channel 32 w + 8 (l>>2) + 4 c + (l&3), so that a lane's registers r = 0..3 of
    // groups c = 0, 1 are EIGHT CONSECUTIVE CHANNELS 32 w + 8 (lane>>4) + 4 c + r: their hi halves are one 16-byte store,
    // their lo halves another (the store burst of the epilogue is bound by the number of store requests, not by bytes).
    unsigned boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int R = r0 + RPP * j;
        const int chan = (R & ~31) + 8 * ((R & 15) >> 2) + 4 * ((R >> 4) & 1) + (R & 3);
        boff[j] = (unsigned)(chan * (int)p.w_row_bytes + u * 16);
    }
    // LDS destinations of this wave's DMA instructions (wave-uniform): row block 8*wave + RPP j
    const int a_row_off = 8 * wave_u * 128;
    uint8_t* const a3_dst_fixed = dummy + wave_u * 1024;             // waves >= 2: the 4th pixel load lands here

    int dma_cblk = 0, dma_tq = 0;
    // chunk q = (channel block cblk, tap tq): the tap is uniform over the workgroup, the walk is scalar
    // the pixel walk (dma_cblk, dma_tq) points at the chunk whose pixels are fetched next (NSA - 1 ahead of the chunk being
    // computed); qb_ = the chunk whose weights are fetched (one ahead)
#define HS_DMA_SETUP(qb_, abuf_, bbuf_)                                                            \
        const int ky = dma_tq / KS, kx = dma_tq - ky * KS;                                         \
        const int dy = dma_on ? ky * p.rate : -(1 << 28), dx = kx * p.rate;                        \
        const long delta = ((long)dy * p.Win + dx) * p.x_pix_bytes + (long)dma_cblk * 128;         \
        uint8_t* a_ = As + (abuf_) * T::A_SLOT + a_row_off;                                        \
        uint8_t* b_ = Bs + (bbuf_) * T::B_SLOT + a_row_off;                                        \
        uint8_t* a3_ = wave_u < 2 ? a_ + 192 * 128 : a3_dst_fixed;     /* the last pixel pass */       \
        const uint8_t* wq = wg + (long)((qb_) < p.nchunks ? (qb_) : p.nchunks - 1) * 128;
#define HS_DMA_A(j_)                                                                               \
    {                                                                                              \
        const int iy = iy0[j_] + dy, ix = ix0[j_] + dx;                                            \
        const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;          \
        __builtin_amdgcn_global_load_lds((gptr_t*)(ok ? abase[j_] + delta : p.zeros),              \
                                         (lptr_t*)((j_) == NAJ - 1 ? a3_ : a_ + (j_) * RPP * 128), 16, 0, 0); \
    }
#define HS_DMA_B(j_)                                                                               \
    __builtin_amdgcn_global_load_lds((gptr_t*)(wq + boff[j_]), (lptr_t*)(b_ + (j_) * RPP * 128), 16, 0, 0);
    // NSA == 3: the four weight loads first, so the counted wait at the end of the chunk can leave the pixel loads in flight
#define HS_DMA_SLOT(s_)                                                                            \
    {                                                                                              \
        if constexpr (NSA == 2) {                                                                  \
            if constexpr ((s_) == 0) HS_DMA_A(0)                                                   \
            if constexpr ((s_) == 1) HS_DMA_B(0)                                                   \
            if constexpr ((s_) == 2) HS_DMA_A(1)                                                   \
            if constexpr ((s_) == 3) HS_DMA_B(1)                                                   \
            if constexpr ((s_) == 4) HS_DMA_A(2)                                                   \
            if constexpr ((s_) == 5) HS_DMA_B(2)                                                   \
            if constexpr ((s_) == 6) HS_DMA_A(3)                                                   \
            if constexpr ((s_) == 7) HS_DMA_B(3)                                                   \
        } else {                                                                                   \
            if constexpr ((s_) == 0) HS_DMA_B(0)                                                   \
            if constexpr ((s_) == 1) HS_DMA_B(1)                                                   \
            if constexpr ((s_) == 2) HS_DMA_B(2)                                                   \
            if constexpr ((s_) == 3) HS_DMA_B(3)                                                   \
            if constexpr ((s_) == 4) HS_DMA_A(0)                                                   \
            if constexpr ((s_) == 5) HS_DMA_A(1)                                                   \
            if constexpr ((s_) == 6) HS_DMA_A(2)                                                   \
            if constexpr ((s_) == 7) HS_DMA_A(3)                                                   \
            if constexpr ((s_) == 8 && NAJ > 4) HS_DMA_A(NAJ > 4 ? 4 : 0)                          \
            if constexpr ((s_) == 9 && NAJ > 5) HS_DMA_A(NAJ > 5 ? 5 : 0)                          \
            if constexpr ((s_) == 10 && NAJ > 6) HS_DMA_A(NAJ > 6 ? 6 : 0)                         \
        }                                                                                          \
    }
    // all pixel loads of a chunk (the prologue's)
#define HS_DMA_A_ALL                                                                               \
    {                                                                                              \
        HS_DMA_A(0) HS_DMA_A(1) HS_DMA_A(2) HS_DMA_A(3)                                            \
        if constexpr (NAJ > 4) { HS_DMA_A(NAJ > 4 ? 4 : 0) HS_DMA_A(NAJ > 5 ? 5 : 0) HS_DMA_A(NAJ > 6 ? 6 : 0) } \
    }
#define HS_DMA_ADVANCE if (++dma_tq == p.cpb) { dma_tq = 0; ++dma_cblk; }

    // ---- accumulators: acc[c][i] = 16 channels (group c of this wave) x 16 pixels (group i); they start at
    // bias * bias_scale (exact: a power of two) so the epilogue issues no load (conv_igemm_h3.h)
    f32x4 acc[T::NC][NP];
#pragma unroll
    for (int c = 0; c < T::NC; ++c) {
        const float4 bv4 = *reinterpret_cast<const float4*>(bg + wave_u * 32 + 8 * q16 + 4 * c);
        const float s = p.bias_scale;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            acc[c][i][0] = bv4.x * s; acc[c][i][1] = bv4.y * s; acc[c][i][2] = bv4.z * s; acc[c][i][3] = bv4.w * s;
        }
    }

    // fragment byte offsets inside an LDS row for plane hi / lo: logical unit plane*4 + (lane>>4), XOR-swizzled with
    // (row>>1)&7 = ((lane&15)>>1)&7 (16-row group bases are multiples of 16)
    const int f_hi = ((0 + q16) ^ ((l16 >> 1) & 7)) * 16, f_lo = ((4 + q16) ^ ((l16 >> 1) & 7)) * 16;

    // ---- one chunk: the weights' four fragments are read once, the pixel fragments of group i + 2 are requested
    // before the six MFMAs of group i are queued.  Reads are inline asm so the waits can be COUNTED (LDS returns in
    // order: "lgkmcnt(n)" = all but the newest n reads have landed); each wait names the fragments it releases.
#define HS_RD(dst_, addr_, off_) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(off_) : "memory")
#define HS_WAIT1(n_, x_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x_) : "n"(n_))
#define HS_WAIT2(n_, x_, y_) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x_), "+v"(y_) : "n"(n_))
#define HS_MFMA(w_, x_, C_, I_) acc[C_][I_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_, x_, acc[C_][I_], 0, 0, 0);
    // group I_ of the chunk; XH_/XL_ = this group's pixel fragments, NH_/NL_ = the ring registers group I_+2 loads into
#define HS_GROUP(I_, XH_, XL_, NH_, NL_)                                                           \
    {                                                                                              \
        constexpr int ahead = ((I_) + 1 < NP ? 2 : 0) + ((I_) + 2 < NP ? 2 : 0);  /* reads newer than this group's */ \
        if constexpr ((I_) + 2 < NP) {                                                             \
            HS_RD(NH_, x_h, ((I_) + 2) * 16 * 128);                                                \
            HS_RD(NL_, x_l, ((I_) + 2) * 16 * 128);                                                \
        }                                                                                          \
        if constexpr ((I_) == 0) {                                                                 \
            /* issue order of the chunk's first reads: wh0 xh0 wl0 xl0 wh1 wl1 | xh1 xl1 | xh2 xl2 */ \
            HS_WAIT2(8, wh0, XH_);                                                                 \
            HS_MFMA(wh0, XH_, 0, 0)                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            HS_WAIT1(7, wl0);                                                                      \
            HS_MFMA(wl0, XH_, 0, 0)                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            HS_WAIT1(6, XL_);                                                                      \
            HS_MFMA(wh0, XL_, 0, 0)                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            HS_WAIT2(4, wh1, wl1);                                                                 \
            HS_MFMA(wh1, XH_, 1, 0) HS_MFMA(wl1, XH_, 1, 0) HS_MFMA(wh1, XL_, 1, 0)                \
        } else {                                                                                   \
            HS_WAIT1(ahead + 1, XH_);                                                              \
            if constexpr ((I_) >= 1 && (I_) <= 4 + NAJ) HS_DMA_SLOT((I_) - 1)   /* this group's share of the next chunk's DMA */ \
            HS_MFMA(wh0, XH_, 0, I_) HS_MFMA(wh1, XH_, 1, I_) HS_MFMA(wl0, XH_, 0, I_) HS_MFMA(wl1, XH_, 1, I_) \
            HS_WAIT1(ahead, XL_);                                                                  \
            HS_MFMA(wh0, XL_, 0, I_) HS_MFMA(wh1, XL_, 1, I_)                                      \
            if constexpr ((I_) >= 1 && (I_) <= 4 + NAJ) {                                          \
                _Pragma("unroll") for (int r_ = 0; r_ < 6; ++r_) {                                 \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
                    __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);                             \
                    if (r_ == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
#define HS_CHUNK(abuf_, anbuf_, bbuf_, q_)                                                         \
    {                                                                                              \
        const unsigned x0 = lds_u32(As + (abuf_) * T::A_SLOT + l16 * 128);                         \
        const unsigned w0 = lds_u32(Bs + (bbuf_) * T::B_SLOT + (wave_u * 32 + l16) * 128);         \
        const unsigned x_h = x0 + f_hi, x_l = x0 + f_lo, w_h = w0 + f_hi, w_l = w0 + f_lo;         \
        half8 wh0, wl0, wh1, wl1, xa_h, xa_l, xb_h, xb_l, xc_h, xc_l;                              \
        HS_RD(wh0, w_h, 0);                                                                        \
        HS_RD(xa_h, x_h, 0);                                                                       \
        HS_RD(wl0, w_l, 0);                                                                        \
        HS_RD(xa_l, x_l, 0);                                                                       \
        HS_RD(wh1, w_h, 16 * 128);                                                                 \
        HS_RD(wl1, w_l, 16 * 128);                                                                 \
        HS_RD(xb_h, x_h, 16 * 128);                                                                \
        HS_RD(xb_l, x_l, 16 * 128);                                                                \
        /* the scalar walk to the next chunk (tap, channel block, addresses) runs behind the reads it does not feed */ \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        HS_DMA_SETUP((q_) + 1, anbuf_, (bbuf_) ^ 1)                                                \
        HS_GROUP(0, xa_h, xa_l, xc_h, xc_l)   HS_GROUP(1, xb_h, xb_l, xa_h, xa_l)   HS_GROUP(2, xc_h, xc_l, xb_h, xb_l) \
        HS_GROUP(3, xa_h, xa_l, xc_h, xc_l)   HS_GROUP(4, xb_h, xb_l, xa_h, xa_l)   HS_GROUP(5, xc_h, xc_l, xb_h, xb_l) \
        HS_GROUP(6, xa_h, xa_l, xc_h, xc_l)   HS_GROUP(7, xb_h, xb_l, xa_h, xa_l)   HS_GROUP(8, xc_h, xc_l, xb_h, xb_l) \
        HS_GROUP(9, xa_h, xa_l, xc_h, xc_l)   HS_GROUP(10, xb_h, xb_l, xa_h, xa_l)  HS_GROUP(11, xc_h, xc_l, xb_h, xb_l) \
        HS_GROUP(12, xa_h, xa_l, xc_h, xc_l)                                                       \
        HS_DMA_ADVANCE                                                                             \
    }

    constexpr int WAIT_ALL = (7 << 4) | (15 << 8);                   // vmcnt(0) only
    constexpr int WAIT_KEEP_A = NAJ | (7 << 4) | (15 << 8);          // vmcnt(NAJ): this iteration's pixel loads stay in flight
    {
        // LDS rings: 2 weight slots, NSA pixel slots.  Chunks 0 (.. NSA-2) are fetched up front; in iteration q the weights
        // of chunk q+1 and the pixels of chunk q+NSA-1 are issued from inside the matrix groups of chunk q (address
        // arithmetic in the shadow of queued MFMAs); the end of the iteration waits for everything chunk q+1 reads and
        // publishes it with the raw s_barrier.  A slot refilled in iteration q was last read in iteration q-1, which every
        // wave left through the previous barrier.
        {
            constexpr bool dma_on = true;
            {
                HS_DMA_SETUP(0, 0, 0)
                HS_DMA_A_ALL HS_DMA_B(0) HS_DMA_B(1) HS_DMA_B(2) HS_DMA_B(3)
                HS_DMA_ADVANCE
            }
            if constexpr (NSA == 3) {
                if (p.nchunks > 1) {
                    HS_DMA_SETUP(0, 1, 0)
                    (void)b_; (void)wq;
                    HS_DMA_A_ALL
                }
                HS_DMA_ADVANCE
            }
        }
        __builtin_amdgcn_s_waitcnt(WAIT_ALL);
        __builtin_amdgcn_s_barrier();
        int aslot = 0;
        for (int q = 0; q < p.nchunks; ++q) {
            // the last chunk(s) have nothing to prefetch: their DMA slots fail every bounds test (zero line) and re-load
            // the last weight chunk into the idle slot instead of branching around the interleaved code
            const bool dma_on = q + NSA - 1 < p.nchunks && !HS_DBG(1);
            const int anext = NSA == 2 ? (aslot ^ 1) : (aslot == 0 ? 2 : aslot - 1);     // slot of chunk q + NSA - 1 = (q - 1) mod NSA
            HS_CHUNK(aslot, anext, q & 1, q)
            if constexpr (NSA == 3) __builtin_amdgcn_s_waitcnt(WAIT_KEEP_A);
            else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
            __builtin_amdgcn_s_barrier();
            aslot = NSA == 2 ? (aslot ^ 1) : (aslot == 2 ? 0 : aslot + 1);
        }
        if constexpr (NSA == 3) {
            __builtin_amdgcn_s_waitcnt(WAIT_ALL);                    // the last iteration's filler loads must land before LDS is reused
            __builtin_amdgcn_s_barrier();
        }
    }

    if (HS_DBG(32)) return;                                          // tuning build only: no epilogue
    // ---- epilogues.  acc[c][i][r]: channel = 32 wave + 8 (lane>>4) + 4 c + r, pixel row of the tile = 16 i + (lane&15)
    const int ch0 = wave_u * 32 + 8 * q16;                           // this lane's first channel inside the N tile (c = 0, r = 0)
    const int row0 = mtile * T::BM;

    // pose head fused (y_mode 2, cnv7): pred 1x1 and the spatial mean are linear (nets/posenn.py:240-241), so the
    // tile delivers sum_pixels sum_channels relu(x) * Wpred[channel][k], split by image; pose_from_tiles adds the
    // tiles in a fixed order.  Fixed summation order here too -> bitwise reproducible run to run.
    if (p.y_mode == 2) {
        const int img0 = row0 / p.pose_P;
        const int split_row = (img0 + 1) * p.pose_P - row0;          // tile rows >= split_row belong to the next image
        float qv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < T::NC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = ntile * T::BN + ch0 + c * 4 + r;
                const bool n_ok = n < p.Cout;
                const float* wp = p.pose_w + ((long)grp * p.Cout + (n_ok ? n : 0)) * 3;
                const float w0 = n_ok ? wp[0] : 0.f, w1 = n_ok ? wp[1] : 0.f, w2 = n_ok ? wp[2] : 0.f;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int row = 16 * i + l16;
                    float v = fmaxf(acc[c][i][r], 0.f);
                    if (row0 + row >= p.M) v = 0.f;
                    if (row < split_row) s0 += v; else s1 += v;
                }
                s0 *= p.out_scale; s1 *= p.out_scale;                // a positive power of two: exact
                qv[0] += s0 * w0; qv[1] += s0 * w1; qv[2] += s0 * w2;
                qv[3] += s1 * w0; qv[4] += s1 * w1; qv[5] += s1 * w2;
            }
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) qv[k] += __shfl_down(qv[k], o, 64);
        float* red = reinterpret_cast<float*>(smem_s);               // [waves][6]; every wave left the last barrier
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[wave_u * 6 + k] = qv[k];
        __syncthreads();
        if (tid < 6) {
            float t = 0.f;
            for (int w = 0; w < T::WAVES; ++w) t += red[w * 6 + tid];
            float* dst = p.pose_partial + (((long)grp * p.pose_mt + (mtile - p.mtile0)) * p.ntiles_n + ntile) * 6 + tid;
            if (p.pose_counter) { agent_store(dst, t); agent_stores_done(); }      // read by the launch's last workgroup
            else *dst = t;
        }
        // the workgroup that finishes last adds the tiles in fixed order and writes the poses (pose_tail.h)
        if (p.pose_counter && last_workgroup(p.pose_counter, (unsigned)p.pose_total, reinterpret_cast<unsigned*>(red + 64))) pose_from_tiles_tail<T::THREADS>(p);
        return;
    }

    float vmax = 0.f;                                                // largest stored value of this lane (range monitor)
    const int ng0 = p.y_coff + grp * p.g_y_coff + ntile * T::BN + ch0;          // this lane's first channel in the output tensor
    if (p.y_mode == 1 && row0 + T::BM <= p.M && (ntile + 1) * T::BN <= p.Cout && p.y_ld >= 32 && (ng0 & 7) == 0) {
        // split store, interior tile: channels are blocked by 32 per pixel ([32 hi | 32 lo] halves); this lane's eight
        // consecutive channels (inside one block since ng0 % 8 == 0) are 16 bytes of hi halves and 16 bytes of lo halves
        // 64 bytes further on.  32-bit offsets from a uniform tile base, ReLU folded into the lower clamp.
        uint8_t* __restrict__ tbase = p.y + (long)row0 * p.y_ld * 4;
        const unsigned rowb = (unsigned)p.y_ld * 4u;
        const float lo_clamp = p.relu ? 0.f : -65504.f;
        const unsigned coff = (unsigned)((ng0 >> 5) * 128 + (ng0 & 31) * 2);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            uint8_t* __restrict__ rowp = tbase + ((unsigned)(16 * i + l16) * rowb + coff);
            unsigned short hb[8], lb[8];
#pragma unroll
            for (int c = 0; c < T::NC; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = fmaxf(acc[c][i][r] * p.out_scale, lo_clamp);
                    vmax = fmaxf(vmax, fabsf(v));
                    v = fminf(v, 65504.f);                           // fp16 range; see DESIGN.md
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    hb[4 * c + r] = __builtin_bit_cast(unsigned short, hi);
                    lb[4 * c + r] = __builtin_bit_cast(unsigned short, lo);
                }
            uint4 hw, lw;
            hw.x = (unsigned)hb[0] | ((unsigned)hb[1] << 16); hw.y = (unsigned)hb[2] | ((unsigned)hb[3] << 16);
            hw.z = (unsigned)hb[4] | ((unsigned)hb[5] << 16); hw.w = (unsigned)hb[6] | ((unsigned)hb[7] << 16);
            lw.x = (unsigned)lb[0] | ((unsigned)lb[1] << 16); lw.y = (unsigned)lb[2] | ((unsigned)lb[3] << 16);
            lw.z = (unsigned)lb[4] | ((unsigned)lb[5] << 16); lw.w = (unsigned)lb[6] | ((unsigned)lb[7] << 16);
            if (HS_DBG(64)) {                                         // tuning build only: the arithmetic, no stores
                asm volatile("" :: "v"(hw.x), "v"(hw.y), "v"(hw.z), "v"(hw.w), "v"(lw.x), "v"(lw.y), "v"(lw.z), "v"(lw.w));
                continue;
            }
            if (HS_DBG(512)) {       // tuning build only (WRONG placement, same bytes): whole 1-KB rows per store instruction
                uint8_t* r1 = tbase + (unsigned)(wave_u * 26 + 2 * i) * rowb + lane * 16;
                *reinterpret_cast<uint4*>(r1) = hw;
                *reinterpret_cast<uint4*>(r1 + rowb) = lw;
                continue;
            }
            if (HS_DBG(256)) {                                        // tuning build only: non-temporal stores
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 hv = {hw.x, hw.y, hw.z, hw.w}, lv = {lw.x, lw.y, lw.z, lw.w};
                __builtin_nontemporal_store(hv, reinterpret_cast<u32x4*>(rowp));
                __builtin_nontemporal_store(lv, reinterpret_cast<u32x4*>(rowp + 64));
                continue;
            }
            *reinterpret_cast<uint4*>(rowp) = hw;
            *reinterpret_cast<uint4*>(rowp + 64) = lw;
        }
    } else {
        // edge tiles (rows past M, channels past Cout), narrow tensors, float32 output: one guarded store per value
        const int ocb_log2 = p.y_ld >= 32 ? 5 : (p.y_ld == 16 ? 4 : 3);
        const int ocb = 1 << ocb_log2;
#pragma unroll
        for (int c = 0; c < T::NC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = ntile * T::BN + ch0 + c * 4 + r;
                const bool n_ok = n < p.Cout;
                const int ng = p.y_coff + grp * p.g_y_coff + n;
                const long cbyte = (long)(ng >> ocb_log2) * (ocb * 4) + (ng & (ocb - 1)) * 2;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int m = row0 + 16 * i + l16;
                    float v = acc[c][i][r] * p.out_scale;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (n_ok && m < p.M) {
                        if (p.y_mode == 0) {
                            reinterpret_cast<float*>(p.y)[(long)m * p.y_ld + ng] = v;
                        } else {
                            vmax = fmaxf(vmax, fabsf(v));
                            v = fminf(fmaxf(v, -65504.f), 65504.f);
                            const _Float16 hi = (_Float16)v;
                            const _Float16 lo = (_Float16)(v - (float)hi);
                            uint8_t* o = p.y + (long)m * p.y_ld * 4 + cbyte;
                            *reinterpret_cast<_Float16*>(o) = hi;
                            *reinterpret_cast<_Float16*>(o + ocb * 2) = lo;
                        }
                    }
                }
            }
    }
    if (p.y_mode == 1 && p.range) {      // non-negative floats order like their bit patterns; inf = overflow
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        range_note(p.range, vmax, lane == 0);
    }
}

#undef HS_DBG
#undef HS_DMA_SETUP
#undef HS_DMA_A
#undef HS_DMA_B
#undef HS_DMA_SLOT
#undef HS_DMA_A_ALL
#undef HS_DMA_ADVANCE
#undef HS_RD
#undef HS_WAIT1
#undef HS_WAIT2
#undef HS_MFMA
#undef HS_GROUP
#undef HS_CHUNK

}  // namespace davo
