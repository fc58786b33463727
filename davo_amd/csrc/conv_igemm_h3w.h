// conv_igemm_h3w.h — the f16x3 implicit GEMM of the dilated 3x3 stride-1 layers on tiles of FOUR waves, one per SIMD (round 5,
// option "wave128"): conv_igemm_h3w (cnv5 / cnv6, 256 x 256 tiles, 128 x 128 outputs per wave), conv_igemm_h3w64 (their remainder rows,
// 256 x 64 tiles), conv_igemm_h3w128 (cnv4, 256 x 128 tiles).  All three: conv_igemm_h3's arithmetic, activation layout and shared-tap
// staging, the same products in the same order per accumulator - bit-identical results.
//
// Why: under the board power cap the matrix pipe's sustained rate depends on how many LDS fragment reads feed an MFMA
// (tools/exp/mfma_peak_probe.hip: 1,602 TFLOP/s at 0.25 reads per MFMA, 1,731 at 0.125).  conv_igemm_h3's 256x256 tile gives
// each of its eight waves 64 x 128 outputs: (4 + 8) x 2 fragments per 96 MFMAs = 0.25.  A 128 x 128 wave tile reads
// (8 + 8) x 2 per 192 = 0.167.  The price: 256 accumulator registers per lane, so one wave per SIMD and nothing but the wave's
// own instruction stream to hide LDS and DMA latency behind.  Hence:
//   * the matrix instructions are inline asm with the accumulator tied in the accumulation registers, and every other instruction of
//     the loop sits in a slot behind one of them, pinned by scheduling barriers (W_SLOT);
//   * ONE barrier per chunk, in front of the last column group's MFMAs (not at the chunk's end): by then every wave has read the
//     last fragments of the chunk, the next chunk's weights (and, after kx = 2, the next patch) have landed, so the next chunk's
//     A fragments and first B fragments are requested under the last 24 MFMAs of this one, into a second register set;
//   * the DMA is buffer_load ... lds: 32-bit offsets, a row outside the image is an offset outside the tensor (zeros); the next
//     chunk's weights and a third of the next super-chunk's patch go out from column groups 1 and 2.
// (The header's second and third kernels say what differs for them; DESIGN.md section 9 has the measurements.)
#pragma once
#include "conv_igemm_h3.h"

namespace davo {

template <int RATE> struct TileW {
    static constexpr int BM = 256, BN = 256, THREADS = 256, NW = 4, RPP = 32;
    static constexpr int PR = (BM + 2 * RATE + 7) / 8 * 8;          // patch rows (a DMA wave-instruction = 8 rows)
    static constexpr int A_SLOTS = (PR / 8 + NW - 1) / NW;           // patch DMA instructions per thread per super-chunk (9)
    static constexpr int B_LOADS = BN / RPP;                         // weight DMA instructions per thread per chunk (8)
    static constexpr int PATCH = PR * 128, BSLOT = BN * 128;
    static constexpr int LDS_BYTES = 2 * PATCH + 2 * BSLOT + NW * 1024 + 128;     // + parking + zero row
    static_assert(A_SLOTS == 9 && LDS_BYTES <= 160 * 1024, "staging shape");
};

typedef __attribute__((address_space(3))) void w_lds_t;

#define W_SB __builtin_amdgcn_sched_barrier(0)
#define W_RD(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(off_) : "memory")
// fragment of row group I_ for tap KX_: the patch row, or the zero row for the lanes the tap carries out of the image row
#define W_XRD(dst_, I_, base_, KX_)                                                                 \
    {                                                                                              \
        if constexpr ((KX_) == 1) { W_RD(dst_, base_, (I_) * 2048); }                              \
        else {                                                                                     \
            const unsigned ad_ = (xkeep >> (((KX_) == 0 ? 0 : 8) + (I_))) & 1u ? (base_) + (I_) * 2048 : (xzero | ((base_) & 112u)); \
            W_RD(dst_, ad_, 0);                                                                    \
        }                                                                                          \
    }
#define W_WAIT8(n_, a_) asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a_[0]), "+v"(a_[1]), "+v"(a_[2]), "+v"(a_[3]), "+v"(a_[4]), "+v"(a_[5]), "+v"(a_[6]), "+v"(a_[7]) : "n"(n_))
#define W_WAIT1(n_, x_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x_) : "n"(n_))
#define W_WAIT2(n_, x_, y_) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x_), "+v"(y_) : "n"(n_))
// The matrix instructions are inline asm with the accumulator as ONE in/out operand in the accumulation registers: the 64 accumulator
// quads fill all 256 of them, and left to choose, the compiler writes results out of place, then shuffles and spills accumulators at the
// unrolled loop's back edge (first build: 254 scratch accesses and 1,016 register moves inside the loop).  The scheduler does not
// know an inline-asm MFMA for one, so nothing is left to it: every other instruction of the loop sits in a SLOT behind one matrix
// instruction, pinned by scheduling barriers - at most three vector instructions, the 12 cycles the matrix pipe is busy past the
// instruction's own issue.  (Second build, each column group's DMA arithmetic in front of its 24 MFMAs: level with conv_igemm_h3.)
#ifndef DAVO_W_CHMAJOR
#define DAVO_W_CHMAJOR 0    /* 1: the weights as the matrix instruction's A operand - a lane's accumulator quad is four consecutive CHANNELS of one pixel, 8-byte stores, no lane exchange.  Bit-identical and SLOWER: cnv5 / cnv6 main launches 0.219 / 0.428 -> 0.238 / 0.450 ms, same box (profiles/r05bj_chmajor_ab.log) */
#endif
#if DAVO_W_CHMAJOR
#define W_MFMA1(a_, b_, I_, J_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %1, %0" : "+a"(acc[I_][J_]) : "v"(a_[I_]), "v"(b_))
#else
#define W_MFMA1(a_, b_, I_, J_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[I_][J_]) : "v"(a_[I_]), "v"(b_))
#endif
// bias into the accumulators and the split store of one wave tile of NI_ x NJ_ quads whose first row / channel are row0_ / ch0_ (tile
// row, output channel); channel-major quads store 8 bytes of hi halves and 8 of lo halves per quad, with no lane exchange
#if DAVO_W_CHMAJOR
#define W_BIAS(NI_, NJ_, ch0_)                                                                     \
    _Pragma("unroll") for (int j = 0; j < (NJ_); ++j) {                                            \
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + (ch0_) + j * 16 + 4 * q16);     \
        _Pragma("unroll") for (int i = 0; i < (NI_); ++i) {                                        \
            acc[i][j][0] = b4.x * p.bias_scale; acc[i][j][1] = b4.y * p.bias_scale;                \
            acc[i][j][2] = b4.z * p.bias_scale; acc[i][j][3] = b4.w * p.bias_scale;                \
        }                                                                                          \
    }
#define W_STORE(NI_, NJ_, row0_, ch0_)                                                             \
    {                                                                                              \
        const int cg0 = p.y_coff + (ch0_) + 4 * q16;          /* a multiple of 4; blocks of 32 channels: 64 B of hi halves, 64 B of lo halves */ \
        const unsigned coff0 = (unsigned)((cg0 >> 5) * 128 + (cg0 & 31) * 2);                      \
        _Pragma("unroll") for (int ii = 0; ii < (NI_); ++ii) {                                     \
            uint8_t* __restrict__ rowp = tbase + ((unsigned)((row0_) + ii * 16 + l16) * rowb + coff0);          \
            _Pragma("unroll") for (int jj = 0; jj < (NJ_); ++jj) {                                 \
                unsigned short h_[4], l_[4];                                                       \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                    \
                    float v = fmaxf(acc[ii][jj][r] * p.out_scale, lo_clamp);                       \
                    vmax = fmaxf(vmax, fabsf(v));                                                  \
                    v = fminf(v, 65504.f);                                                         \
                    const _Float16 hi = (_Float16)v;                                               \
                    const _Float16 lo = (_Float16)(v - (float)hi);                                 \
                    h_[r] = __builtin_bit_cast(unsigned short, hi); l_[r] = __builtin_bit_cast(unsigned short, lo); \
                }                                                                                  \
                uint8_t* q_ = rowp + ((jj * 16 >> 5) * 128 + ((jj * 16) & 31) * 2);                \
                *reinterpret_cast<uint2*>(q_) = make_uint2((unsigned)h_[0] | ((unsigned)h_[1] << 16), (unsigned)h_[2] | ((unsigned)h_[3] << 16));       \
                *reinterpret_cast<uint2*>(q_ + 64) = make_uint2((unsigned)l_[0] | ((unsigned)l_[1] << 16), (unsigned)l_[2] | ((unsigned)l_[3] << 16));  \
            }                                                                                      \
        }                                                                                          \
    }
#else
#define W_BIAS(NI_, NJ_, ch0_)                                                                     \
    _Pragma("unroll") for (int j = 0; j < (NJ_); ++j) {                                            \
        const float bv = p.bias[(ch0_) + j * 16 + l16] * p.bias_scale;                             \
        _Pragma("unroll") for (int i = 0; i < (NI_); ++i)                                          \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[i][j][r] = bv;                       \
    }
#define W_STORE(NI_, NJ_, row0_, ch0_)                                                             \
    {                                                                                              \
        const bool odd = lane & 1;                                                                 \
        const unsigned sel = odd ? 0x03020706u : 0x05040100u;                                      \
        const int ng0 = p.y_coff + (ch0_) + l16;                                                   \
        const unsigned coff0 = (unsigned)((ng0 >> 5) * 128 + (ng0 & 31) * 2 + (odd ? 62 : 0));     \
        _Pragma("unroll") for (int ii = 0; ii < (NI_); ++ii)                                       \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                        \
                uint8_t* __restrict__ rowp = tbase + ((unsigned)((row0_) + ii * 16 + 4 * q16 + r) * rowb + coff0);          \
                _Pragma("unroll") for (int jj = 0; jj < (NJ_); ++jj) {                             \
                    float v = fmaxf(acc[ii][jj][r] * p.out_scale, lo_clamp);                       \
                    vmax = fmaxf(vmax, fabsf(v));                                                  \
                    v = fminf(v, 65504.f);                                                         \
                    const _Float16 hi = (_Float16)v;                                               \
                    const _Float16 lo = (_Float16)(v - (float)hi);                                 \
                    const unsigned x = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16); \
                    const unsigned xn = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   /* quad_perm [1,0,3,2] */ \
                    *reinterpret_cast<unsigned*>(rowp + ((jj * 16 >> 5) * 128 + ((jj * 16) & 31) * 2)) = __builtin_amdgcn_perm(xn, x, sel); \
                }                                                                                  \
            }                                                                                      \
    }
#endif
#define W_SLOT(a_, b_, I_, J_, STMT_) W_MFMA1(a_, b_, I_, J_); W_SB; STMT_; W_SB;
#define W_PASS(a_, b_, J_)                                                                         \
    { W_MFMA1(a_, b_, 0, J_); W_MFMA1(a_, b_, 1, J_); W_MFMA1(a_, b_, 2, J_); W_MFMA1(a_, b_, 3, J_);   \
      W_MFMA1(a_, b_, 4, J_); W_MFMA1(a_, b_, 5, J_); W_MFMA1(a_, b_, 6, J_); W_MFMA1(a_, b_, 7, J_); }
// a pass of eight matrix instructions with a statement behind each
#define W_PASS8(a_, b_, J_, S0_, S1_, S2_, S3_, S4_, S5_, S6_, S7_)                                \
    W_SLOT(a_, b_, 0, J_, S0_) W_SLOT(a_, b_, 1, J_, S1_) W_SLOT(a_, b_, 2, J_, S2_) W_SLOT(a_, b_, 3, J_, S3_) \
    W_SLOT(a_, b_, 4, J_, S4_) W_SLOT(a_, b_, 5, J_, S5_) W_SLOT(a_, b_, 6, J_, S6_) W_SLOT(a_, b_, 7, J_, S7_)
#define W_NOP ((void)0)
// Patch slot j_ (name tag t_) of a super-chunk into patch buffer abuf_, in three steps of at most two vector instructions: a buffer
// load whose offset lies outside the tensor returns zeros, which is what a row above or below the image (TF's padding) and a slot
// outside the layer's pixels must read.  Slots past the patch's end (wave-uniform) issue nothing.
#define W_XA(t_, j_) const int xiy##t_ = xyv[j_] + xdy; const bool xok##t_ = (unsigned)xiy##t_ < (unsigned)p.Hin
#define W_XB(t_, j_) const unsigned xvo##t_ = xok##t_ ? poff[j_] + xsoff : 0xFFFFFF00u
#define W_XC(t_, j_, abuf_)                                                                        \
    if ((j_) < 8 || (8 * wave_u + (j_) * 32) < PR)                                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (w_lds_t*)(As + (abuf_) * TW::PATCH + ((j_) * 32 + 8 * wave_u) * 128), 16, xvo##t_, 0, 0, 0)
#define W_BDMA(j_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (w_lds_t*)(Bs + (slot_) * TW::BSLOT + ((j_) * 32 + 8 * wave_u) * 128), 16, boff[j_], wsoff, 0, 0)

// one chunk: tap KX_ of the super-chunk in patch buffer abuf, weights in ring slot bslot, A fragments in set CUR_ (requested by
// the chunk before); requests the next chunk's first fragments (tap NKX_, weights in slot bslot ^ 1) into set CUR_ ^ 1 and issues the
// DMA of the next chunk's weights and this chunk's third of the next super-chunk's patch
#define W_BODY(KX_, CUR_)                                                                          \
    {                                                                                              \
        constexpr int NKX_ = ((KX_) + 1) % 3;                                                      \
        const unsigned b0_ = lds_u32(Bs + bslot * TW::BSLOT + (wn * 128 + l16) * 128);             \
        const unsigned b_h = b0_ + foff16[0], b_l = b0_ + foff16[1];                               \
        /* weights of the next chunk -> slot bslot ^ 1 (free since the barrier of the chunk before this one) */ \
        const unsigned wsoff = (KX_) == 2 ? wnext : wcur + ((KX_) + 1) * 128;                      \
        /* ---- column group 0 */                                                                  \
        W_RD(bh[2], b_h, 2 * 2048); W_RD(bl[2], b_l, 2 * 2048);                                    \
        W_WAIT8(13, AH[CUR_]); W_WAIT1(13, bh[0]);                                                 \
        W_PASS(AH[CUR_], bh[0], 0)                                                                 \
        W_SB;                                                                                      \
        W_WAIT1(12, bl[0]);                                                                        \
        W_PASS(AH[CUR_], bl[0], 0)                                                                 \
        W_SB;                                                                                      \
        W_WAIT8(4, AL[CUR_]);                                                                      \
        W_PASS(AL[CUR_], bh[0], 0)                                                                 \
        W_SB;                                                                                      \
        /* ---- column group 1: this chunk's three patch slots (they come from furthest away), then two weight pieces */ \
        W_RD(bh[3], b_h, 3 * 2048); W_RD(bl[3], b_l, 3 * 2048);                                    \
        W_WAIT2(4, bh[1], bl[1]);                                                                  \
        W_PASS8(AH[CUR_], bh[1], 1, W_XA(0, (KX_) * 3 + 0), W_XB(0, (KX_) * 3 + 0), W_XC(0, (KX_) * 3 + 0, nabuf),     \
                W_XA(1, (KX_) * 3 + 1), W_XB(1, (KX_) * 3 + 1), W_XC(1, (KX_) * 3 + 1, nabuf), W_XA(2, (KX_) * 3 + 2), W_XB(2, (KX_) * 3 + 2)) \
        W_PASS8(AH[CUR_], bl[1], 1, W_XC(2, (KX_) * 3 + 2, nabuf), W_BDMA(0, bslot ^ 1), W_BDMA(1, bslot ^ 1), W_NOP, W_NOP, W_NOP, W_NOP, W_NOP) \
        W_PASS(AL[CUR_], bh[1], 1)                                                                 \
        W_SB;                                                                                      \
        /* ---- column group 2: the other six weight pieces */                                     \
        W_RD(bh[4], b_h, 4 * 2048); W_RD(bl[4], b_l, 4 * 2048);                                    \
        W_WAIT2(4, bh[2], bl[2]);                                                                  \
        W_PASS8(AH[CUR_], bh[2], 2, W_BDMA(2, bslot ^ 1), W_BDMA(3, bslot ^ 1), W_BDMA(4, bslot ^ 1), W_BDMA(5, bslot ^ 1),     \
                W_BDMA(6, bslot ^ 1), W_BDMA(7, bslot ^ 1), W_NOP, W_NOP)                          \
        W_PASS(AH[CUR_], bl[2], 2) W_PASS(AL[CUR_], bh[2], 2)                                      \
        W_SB;                                                                                      \
        /* ---- column groups 3..6 */                                                              \
        W_RD(bh[5], b_h, 5 * 2048); W_RD(bl[5], b_l, 5 * 2048);                                    \
        W_WAIT2(4, bh[3], bl[3]);                                                                  \
        W_PASS(AH[CUR_], bh[3], 3) W_PASS(AH[CUR_], bl[3], 3) W_PASS(AL[CUR_], bh[3], 3)           \
        W_SB;                                                                                      \
        W_RD(bh[6], b_h, 6 * 2048); W_RD(bl[6], b_l, 6 * 2048);                                    \
        W_WAIT2(4, bh[4], bl[4]);                                                                  \
        W_PASS(AH[CUR_], bh[4], 4) W_PASS(AH[CUR_], bl[4], 4) W_PASS(AL[CUR_], bh[4], 4)           \
        W_SB;                                                                                      \
        W_RD(bh[7], b_h, 7 * 2048); W_RD(bl[7], b_l, 7 * 2048);                                    \
        W_WAIT2(4, bh[5], bl[5]);                                                                  \
        W_PASS(AH[CUR_], bh[5], 5) W_PASS(AH[CUR_], bl[5], 5) W_PASS(AL[CUR_], bh[5], 5)           \
        W_SB;                                                                                      \
        W_WAIT2(2, bh[6], bl[6]);                                                                  \
        W_PASS(AH[CUR_], bh[6], 6) W_PASS(AH[CUR_], bl[6], 6) W_PASS(AL[CUR_], bh[6], 6)           \
        W_SB;                                                                                      \
        W_WAIT2(0, bh[7], bl[7]);                                                                  \
        /* ---- column group 7 (the last chunk runs the barrier and the requests too: no accumulator is defined in two branches) */ \
        {                                                                                          \
            /* every wave has read this chunk's last fragments; what this wave fetched for the next chunk has landed */ \
            __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));                   /* vmcnt(0) */       \
            __builtin_amdgcn_s_barrier();                                                          \
            const unsigned pb_ = lds_u32(As + ((KX_) == 2 ? nabuf : abuf) * TW::PATCH);            \
            const int nrow_ = xrow0 + NKX_ * RATE;                                                 \
            const unsigned na_h = pb_ + (unsigned)(nrow_ * 128 + ((q16 ^ (nrow_ & 6)) * 16));      \
            const unsigned na_l = pb_ + (unsigned)(nrow_ * 128 + (((4 + q16) ^ (nrow_ & 6)) * 16)); \
            const unsigned nb0_ = lds_u32(Bs + (bslot ^ 1) * TW::BSLOT + (wn * 128 + l16) * 128);  \
            const unsigned nb_h = nb0_ + foff16[0], nb_l = nb0_ + foff16[1];                       \
            W_SB;                                                                                  \
            W_PASS8(AH[CUR_], bh[7], 7, W_XRD(AH[(CUR_) ^ 1][0], 0, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][1], 1, na_h, NKX_),       \
                    W_XRD(AH[(CUR_) ^ 1][2], 2, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][3], 3, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][4], 4, na_h, NKX_), \
                    W_XRD(AH[(CUR_) ^ 1][5], 5, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][6], 6, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][7], 7, na_h, NKX_)) \
            W_PASS8(AH[CUR_], bl[7], 7, W_RD(bh[0], nb_h, 0), W_RD(bl[0], nb_l, 0),                \
                    W_XRD(AL[(CUR_) ^ 1][0], 0, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][1], 1, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][2], 2, na_l, NKX_), \
                    W_XRD(AL[(CUR_) ^ 1][3], 3, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][4], 4, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][5], 5, na_l, NKX_)) \
            W_PASS8(AL[CUR_], bh[7], 7, W_XRD(AL[(CUR_) ^ 1][6], 6, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][7], 7, na_l, NKX_),       \
                    W_RD(bh[1], nb_h, 2048), W_RD(bl[1], nb_l, 2048), W_NOP, W_NOP, W_NOP, W_NOP)  \
        }                                                                                          \
        W_SB;                                                                                      \
        bslot ^= 1;                                                                                \
    }

// The body takes its workgroup id as parameters (wg_x of nwg_x) so that conv_igemm_h3w_mainrem below can run the layer's remainder
// tiles in the same launch; conv_igemm_h3w passes blockIdx / gridDim.
template <int LAYER, int RATE>
__device__ __forceinline__ void conv_igemm_h3w_body(const ConvParamsH& p, const int wg_x, const int nwg_x) {
    using TW = TileW<RATE>;
    constexpr int PR = TW::PR;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_w[];
    uint8_t* As = smem_w;                               // [2][PR][128] pixel patches
    uint8_t* Bs = smem_w + 2 * TW::PATCH;               // [2][256][128] weight ring
    uint8_t* xdummy = Bs + 2 * TW::BSLOT;               // 1 KB per wave: where the DMA slots past the patch's end land
    const unsigned xzero = lds_u32(xdummy + 4 * 1024);  // 128 zero bytes (what a tap reads outside its image row)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave_u >> 1, wn = wave_u & 1;
    const int l16 = lane & 15, q16 = lane >> 4;
    if (tid < 8) *reinterpret_cast<float4*>(xdummy + 4 * 1024 + tid * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

    const int tile_i = xcd_remap(wg_x, nwg_x);
    const int mtile = p.mtile0 + (p.tile_order ? __builtin_amdgcn_readfirstlane(p.tile_order[tile_i]) : tile_i);    // one N tile: N = 256
    const uint8_t* __restrict__ xg = p.x + p.x_boff;
    const uint8_t* __restrict__ wg = p.w;
    int ky0, nky;
    {
        const FilterRows fr = valid_filter_rows(mtile * 256, mtile * 256 + 255, p.Hout, p.Wout, p.Hin, 1, p.pad_t, p.rate);
        ky0 = __builtin_amdgcn_readfirstlane(fr.ky0);
        nky = __builtin_amdgcn_readfirstlane(fr.nky);
    }
    const int nsc = (p.nchunks / 9) * nky;               // super-chunks (channel block, filter row) this tile walks: three chunks each

    // ---- staging assignment (as conv_igemm_h3: thread -> row r0 + 32 j, 16-byte unit of the 128-byte row, swizzled)
    const int r0 = tid >> 3;
    unsigned boff[8];
    {
        const int u = (tid & 7) ^ ((r0 >> 1) & 7);
#pragma unroll
        for (int j = 0; j < 8; ++j) boff[j] = (unsigned)((r0 + 32 * j) * (int)p.w_row_bytes + u * 16);
    }
    const int ux = (tid & 7) ^ (r0 & 6);
    // byte offset of this thread's 16-byte unit of patch slot j at filter row ky = 0 (modulo 2^32: rows above the first image are
    // negative there and come back into range with the filter row's own offset), and the input row it holds
    unsigned poff[9];
    int xyv[9];
    {
        // (n, y, x) of slots 0 and 1 by division, of the others by stepping 32 pixels on (one wrap per step: the launcher checks the map)
        const int hw = p.Hout * p.Wout;
        const unsigned uoff = (unsigned)((ux >> 2) * 64 + (ux & 3) * 16);        // 32-channel blocks: 64 B of hi halves, 64 B of lo halves
        const int q32 = 32 / p.Wout, r32 = 32 - q32 * p.Wout;
        int n = 0, y = 0, x = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int row = r0 + 32 * j;
            const int f = mtile * 256 - RATE + row;
            const bool ok = row < PR && f >= 0 && f < p.Mtot;
            if (j < 2) {
                const int fc = f >= 0 ? f : 0;
                n = fc / hw;
                const int rem = fc - n * hw;
                y = rem / p.Wout;
                x = rem - y * p.Wout;
            } else {
                x += r32; y += q32;
                if (x >= p.Wout) { x -= p.Wout; ++y; }
                if (y >= p.Hout) { y -= p.Hout; ++n; }
            }
            poff[j] = ((unsigned)((n * p.Hin + y - RATE) * p.Win + x) << p.x_pix_log2) + uoff;
            xyv[j] = ok ? y - RATE : -(1 << 28);
        }
    }
    // buffer resources: offsets are range-checked by the hardware (an out-of-range load returns zeros)
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(xg), 0, (int)((unsigned)p.Mtot << p.x_pix_log2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wg), 0, (int)(256u * (unsigned)p.w_row_bytes), 0x00020000);
    const int xrow0 = wm * 128 + l16;                     // first fragment row of this lane inside the tile
    unsigned xkeep = 0;                                   // bit i: row group i keeps tap kx = 0 (x >= RATE); bit 8 + i: keeps kx = 2
    {
        int x = (mtile * 256 + xrow0) % p.Wout;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (x >= RATE) xkeep |= 1u << i;
            if (x < p.Wout - RATE) xkeep |= 1u << (8 + i);
            x += 16;
            if (x >= p.Wout) x -= p.Wout;                 // Wout > 16 (launcher)
        }
    }
    int foff16[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) foff16[pl] = ((pl * 4 + q16) ^ ((l16 >> 1) & 7)) * 16;

    f32x4 acc[8][8];
    W_BIAS(8, 8, wn * 128)
    half8 AH[2][8], AL[2][8], bh[8], bl[8];

    // ---- prologue: the first patch, the first weight chunk; then the first chunk's first fragments
    {
        const int xdy = ky0 * RATE;
        const unsigned xsoff = (unsigned)(ky0 * RATE * p.Win) << p.x_pix_log2;
        const unsigned wsoff = (unsigned)(ky0 * 3) * 128u;
        W_XA(0, 0); W_XB(0, 0); W_XC(0, 0, 0); W_XA(1, 1); W_XB(1, 1); W_XC(1, 1, 0); W_XA(2, 2); W_XB(2, 2); W_XC(2, 2, 0);
        W_XA(3, 3); W_XB(3, 3); W_XC(3, 3, 0); W_XA(4, 4); W_XB(4, 4); W_XC(4, 4, 0); W_XA(5, 5); W_XB(5, 5); W_XC(5, 5, 0);
        W_XA(6, 6); W_XB(6, 6); W_XC(6, 6, 0); W_XA(7, 7); W_XB(7, 7); W_XC(7, 7, 0); W_XA(8, 8); W_XB(8, 8); W_XC(8, 8, 0);
        W_BDMA(0, 0); W_BDMA(1, 0); W_BDMA(2, 0); W_BDMA(3, 0); W_BDMA(4, 0); W_BDMA(5, 0); W_BDMA(6, 0); W_BDMA(7, 0);
    }
    // the 256 accumulator registers are written HERE, while the first patch and weights are on their way (left alone, the compiler
    // writes them in front of the first matrix instruction, behind the barrier: ~0.4 us per tile with nothing else to do)
#ifndef DAVO_W_EARLYINIT
#define DAVO_W_EARLYINIT 1
#endif
#if DAVO_W_EARLYINIT
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
#endif
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));             // vmcnt(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the zero row's ds_write
    __builtin_amdgcn_s_barrier();
    {
        const unsigned pb_ = lds_u32(As);
        const unsigned na_h = pb_ + (unsigned)(xrow0 * 128 + ((q16 ^ (xrow0 & 6)) * 16));
        const unsigned na_l = pb_ + (unsigned)(xrow0 * 128 + (((4 + q16) ^ (xrow0 & 6)) * 16));
        const unsigned nb0_ = lds_u32(Bs + (wn * 128 + l16) * 128);
        W_XRD(AH[0][0], 0, na_h, 0) W_XRD(AH[0][1], 1, na_h, 0) W_XRD(AH[0][2], 2, na_h, 0) W_XRD(AH[0][3], 3, na_h, 0)
        W_XRD(AH[0][4], 4, na_h, 0) W_XRD(AH[0][5], 5, na_h, 0) W_XRD(AH[0][6], 6, na_h, 0) W_XRD(AH[0][7], 7, na_h, 0)
        W_RD(bh[0], nb0_ + foff16[0], 0); W_RD(bl[0], nb0_ + foff16[1], 0);
        W_XRD(AL[0][0], 0, na_l, 0) W_XRD(AL[0][1], 1, na_l, 0) W_XRD(AL[0][2], 2, na_l, 0) W_XRD(AL[0][3], 3, na_l, 0)
        W_XRD(AL[0][4], 4, na_l, 0) W_XRD(AL[0][5], 5, na_l, 0) W_XRD(AL[0][6], 6, na_l, 0) W_XRD(AL[0][7], 7, na_l, 0)
        W_RD(bh[1], nb0_ + foff16[0], 2048); W_RD(bl[1], nb0_ + foff16[1], 2048);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- main loop.  Chunk q = 3 sc + kx alternates between the two fragment sets, so two super-chunks are unrolled.
    int bslot = 0;
    int ccblk = 0, cky = ky0;                             // the current super-chunk's channel block and filter row
#define W_SUPER(P0_, P1_, P2_)                                                                     \
    {                                                                                              \
        int xcblk = ccblk, xky = cky + 1;                  /* the super-chunk whose patch this one fetches */ \
        if (xky == ky0 + nky) { xky = ky0; ++xcblk; }                                              \
        const bool xmore = sc + 1 < nsc;                                                           \
        const int xdy = xmore ? xky * RATE : -(1 << 28);                                           \
        const unsigned xsoff = ((unsigned)(xky * RATE * p.Win) << p.x_pix_log2) + (unsigned)xcblk * 128u;     \
        const unsigned wcur = (unsigned)((ccblk * 3 + cky) * 3) * 128u;                            \
        const unsigned wnext = xmore ? (unsigned)((xcblk * 3 + xky) * 3) * 128u : wcur;            \
        const int abuf = sc & 1, nabuf = abuf ^ 1;                                                 \
        W_BODY(0, P0_) W_BODY(1, P1_) W_BODY(2, P2_)                                               \
        ccblk = xcblk; cky = xky;                                                                  \
        ++sc;                                                                                      \
    }
    int sc = 0;
    while (sc + 2 <= nsc) {
        W_SUPER(0, 1, 0)
        W_SUPER(1, 0, 1)
    }
    if (sc < nsc) W_SUPER(0, 1, 0)
#undef W_SUPER
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));             // the last chunk's filler DMA must land before the LDS is given back
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // ... and its filler fragment requests before their registers are reused

    // ---- epilogue: interior tile, blocks of 32 channels (checked by the launcher): the split store
    uint8_t* __restrict__ tbase = p.y + (long)mtile * 256 * p.y_ld * 4;
    const unsigned rowb = (unsigned)p.y_ld * 4u;
    const float lo_clamp = p.relu ? 0.f : -65504.f;
    float vmax = 0.f;
    W_STORE(8, 8, wm * 128, wn * 128)
    if (p.range) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        range_note(p.range, vmax, lane == 0);
    }
}

template <int LAYER, int RATE>
__global__ __launch_bounds__(256, 1) void conv_igemm_h3w(ConvParamsH p) {
    conv_igemm_h3w_body<LAYER, RATE>(p, blockIdx.x, gridDim.x);
}


// ---- the layer's remainder rows: 256 x 64 tiles (conv_igemm_h3w64) ------------------------------------------------------------
// At B = 32 cnv5 / cnv6 are 3.25 rounds of 256-row tiles; the last quarter round ran as 256 workgroups of conv_igemm_h3's 128x128 tile
// (three ring slots): 33 / 57 us for 7.7 % of the rows, MFMA busy 0.26 / 0.38 - a chain of 72 chunks at 0.79 us each, which is what a CU
// takes in from L2 (32 KB per chunk at ~45 GB/s), not what its matrix pipe needs (0.32 us).  Same rows, same 256 workgroups, less to
// stage: a 256-row x 64-channel tile stages the shared pixel patch (11 KB per chunk's share) and 8 KB of weights per chunk - 19 KB for the
// same 3.1 MFLOP.  Four waves of 64 x 64 outputs (wave w = rows 64 w ..), conv_igemm_h3w's slots and barrier placement; a chunk is only
// 48 matrix instructions per wave (0.32 us), so the rings are deep: weights four chunks ahead in a ring of six, patches two super-chunks
// ahead in a ring of three, and the wait in front of the barrier is COUNTED (everything but the DMA of the newest three chunks).
template <int RATE> struct TileW64 {
    static constexpr int PR = TileW<RATE>::PR, PATCH = PR * 128, BSLOT = 64 * 128, NB = 6, NP = 3, AHEAD = 4;
    static constexpr int LDS_BYTES = NP * PATCH + NB * BSLOT + 4 * 1024 + 128;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per workgroup");
};

#define W64_WAIT4(n_, a_) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a_[0]), "+v"(a_[1]), "+v"(a_[2]), "+v"(a_[3]) : "n"(n_))
#define W64_PASS(a_, b_, J_) { W_MFMA1(a_, b_, 0, J_); W_MFMA1(a_, b_, 1, J_); W_MFMA1(a_, b_, 2, J_); W_MFMA1(a_, b_, 3, J_); }
#define W64_PASS4(a_, b_, J_, S0_, S1_, S2_, S3_) \
    W_SLOT(a_, b_, 0, J_, S0_) W_SLOT(a_, b_, 1, J_, S1_) W_SLOT(a_, b_, 2, J_, S2_) W_SLOT(a_, b_, 3, J_, S3_)
// patch slot (always issued: a slot past the patch's end reads outside the tensor into the wave's parking KB, so that every wave's
// vmcnt counts the same five DMA instructions per chunk)
#define W64_XC(t_, j_, abuf_)                                                                      \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (w_lds_t*)(((j_) < 8 || (8 * wave_u + (j_) * 32) < PR)                     \
                                                     ? As + (abuf_) * TW::PATCH + ((j_) * 32 + 8 * wave_u) * 128 : xdummy + wave_u * 1024), \
                                             16, ((j_) < 8 || (8 * wave_u + (j_) * 32) < PR) ? xvo##t_ : 0xFFFFFF00u, 0, 0, 0)
#define W64_BDMA(j_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (w_lds_t*)(Bs + (slot_) * TW::BSLOT + ((j_) * 32 + 8 * wave_u) * 128), 16, boff[j_], wsoff, 0, 0)
#define W64_BODY(KX_, CUR_)                                                                        \
    {                                                                                              \
        constexpr int NKX_ = ((KX_) + 1) % 3;                                                      \
        const unsigned b0_ = lds_u32(Bs + wslot * TW::BSLOT + l16 * 128);                          \
        const unsigned b_h = b0_ + foff16[0], b_l = b0_ + foff16[1];                               \
        /* weights of chunk q + 4 = tap (KX_ + 1) % 3 of the next super-chunk (the one after it for KX_ = 2) -> ring slot wslot + 4 */ \
        const unsigned wsoff = ((KX_) == 2 ? w2 : w1) + NKX_ * 128;                                \
        const int wdst = wslot + 4 >= TW::NB ? wslot + 4 - TW::NB : wslot + 4;                     \
        /* ---- column group 0 */                                                                  \
        W_RD(bh[2], b_h, 2 * 2048); W_RD(bl[2], b_l, 2 * 2048);                                    \
        W64_WAIT4(9, AH[CUR_]); W_WAIT1(9, bh[0]);                                                 \
        W64_PASS(AH[CUR_], bh[0], 0)                                                               \
        W_SB;                                                                                      \
        W_WAIT1(8, bl[0]);                                                                         \
        W64_PASS(AH[CUR_], bl[0], 0)                                                               \
        W_SB;                                                                                      \
        W64_WAIT4(4, AL[CUR_]);                                                                    \
        W64_PASS(AL[CUR_], bh[0], 0)                                                               \
        W_SB;                                                                                      \
        /* ---- column group 1: the chunk's five DMA instructions */                               \
        W_RD(bh[3], b_h, 3 * 2048); W_RD(bl[3], b_l, 3 * 2048);                                    \
        W_WAIT2(4, bh[1], bl[1]);                                                                  \
        W64_PASS4(AH[CUR_], bh[1], 1, W_XA(0, (KX_) * 3 + 0), W_XB(0, (KX_) * 3 + 0), W64_XC(0, (KX_) * 3 + 0, pbuf2), W_XA(1, (KX_) * 3 + 1)) \
        W64_PASS4(AH[CUR_], bl[1], 1, W_XB(1, (KX_) * 3 + 1), W64_XC(1, (KX_) * 3 + 1, pbuf2), W_XA(2, (KX_) * 3 + 2), W_XB(2, (KX_) * 3 + 2)) \
        W64_PASS4(AL[CUR_], bh[1], 1, W64_XC(2, (KX_) * 3 + 2, pbuf2), W64_BDMA(0, wdst), W64_BDMA(1, wdst), W_NOP)                        \
        W_SB;                                                                                      \
        /* ---- column group 2 */                                                                  \
        W_WAIT2(2, bh[2], bl[2]);                                                                  \
        W64_PASS(AH[CUR_], bh[2], 2) W64_PASS(AH[CUR_], bl[2], 2) W64_PASS(AL[CUR_], bh[2], 2)     \
        W_SB;                                                                                      \
        W_WAIT2(0, bh[3], bl[3]);                                                                  \
        /* ---- column group 3: everything but the DMA of the newest three chunks has landed (the next chunk's weights were issued three    \
           chunks ago, the next super-chunk's patch during the super-chunk before this one) */     \
        {                                                                                          \
            __builtin_amdgcn_s_waitcnt((15 & 15) | (7 << 4) | (15 << 8) | ((15 >> 4) << 14));      /* vmcnt(15) */ \
            __builtin_amdgcn_s_barrier();                                                          \
            const unsigned pb_ = lds_u32(As + ((KX_) == 2 ? pbuf1 : pbuf0) * TW::PATCH);           \
            const int nrow_ = xrow0 + NKX_ * RATE;                                                 \
            const unsigned na_h = pb_ + (unsigned)(nrow_ * 128 + ((q16 ^ (nrow_ & 6)) * 16));      \
            const unsigned na_l = pb_ + (unsigned)(nrow_ * 128 + (((4 + q16) ^ (nrow_ & 6)) * 16)); \
            const int nws_ = wslot + 1 == TW::NB ? 0 : wslot + 1;                                  \
            const unsigned nb0_ = lds_u32(Bs + nws_ * TW::BSLOT + l16 * 128);                      \
            const unsigned nb_h = nb0_ + foff16[0], nb_l = nb0_ + foff16[1];                       \
            W_SB;                                                                                  \
            W64_PASS4(AH[CUR_], bh[3], 3, W_XRD(AH[(CUR_) ^ 1][0], 0, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][1], 1, na_h, NKX_),               \
                      W_XRD(AH[(CUR_) ^ 1][2], 2, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][3], 3, na_h, NKX_))                                    \
            W64_PASS4(AH[CUR_], bl[3], 3, W_RD(bh[0], nb_h, 0), W_RD(bl[0], nb_l, 0),               \
                      W_XRD(AL[(CUR_) ^ 1][0], 0, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][1], 1, na_l, NKX_))                                    \
            W64_PASS4(AL[CUR_], bh[3], 3, W_XRD(AL[(CUR_) ^ 1][2], 2, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][3], 3, na_l, NKX_),               \
                      W_RD(bh[1], nb_h, 2048), W_RD(bl[1], nb_l, 2048))                            \
        }                                                                                          \
        W_SB;                                                                                      \
        wslot = wslot + 1 == TW::NB ? 0 : wslot + 1;                                               \
    }

template <int LAYER, int RATE>
__global__ __launch_bounds__(256, 1) void conv_igemm_h3w64(ConvParamsH p) {
    using TW = TileW64<RATE>;
    constexpr int PR = TW::PR;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_w6[];
    uint8_t* As = smem_w6;                              // [3][PR][128] pixel patches
    uint8_t* Bs = smem_w6 + TW::NP * TW::PATCH;         // [6][64][128] weight ring
    uint8_t* xdummy = Bs + TW::NB * TW::BSLOT;          // 1 KB per wave: where a patch slot past the patch's end lands
    const unsigned xzero = lds_u32(xdummy + 4 * 1024);  // 128 zero bytes (what a tap reads outside its image row)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, q16 = lane >> 4;
    if (tid < 8) *reinterpret_cast<float4*>(xdummy + 4 * 1024 + tid * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = tile & 3, mtile = p.mtile0 + (tile >> 2);          // four N tiles of 64 channels per 256-row tile
    const uint8_t* __restrict__ xg = p.x + p.x_boff;
    const uint8_t* __restrict__ wg = p.w + (long)ntile * 64 * p.w_row_bytes;
    int ky0, nky;
    {
        const FilterRows fr = valid_filter_rows(mtile * 256, mtile * 256 + 255, p.Hout, p.Wout, p.Hin, 1, p.pad_t, p.rate);
        ky0 = __builtin_amdgcn_readfirstlane(fr.ky0);
        nky = __builtin_amdgcn_readfirstlane(fr.nky);
    }
    const int nsc = (p.nchunks / 9) * nky;

    const int r0 = tid >> 3;
    unsigned boff[2];
    {
        const int u = (tid & 7) ^ ((r0 >> 1) & 7);
#pragma unroll
        for (int j = 0; j < 2; ++j) boff[j] = (unsigned)((r0 + 32 * j) * (int)p.w_row_bytes + u * 16);
    }
    const int ux = (tid & 7) ^ (r0 & 6);
    unsigned poff[9];
    int xyv[9];
    {
        const int hw = p.Hout * p.Wout;
        const unsigned uoff = (unsigned)((ux >> 2) * 64 + (ux & 3) * 16);
        const int q32 = 32 / p.Wout, r32 = 32 - q32 * p.Wout;
        int n = 0, y = 0, x = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int row = r0 + 32 * j;
            const int f = mtile * 256 - RATE + row;
            const bool ok = row < PR && f >= 0 && f < p.Mtot;
            if (j < 2) {
                const int fc = f >= 0 ? f : 0;
                n = fc / hw;
                const int rem = fc - n * hw;
                y = rem / p.Wout;
                x = rem - y * p.Wout;
            } else {
                x += r32; y += q32;
                if (x >= p.Wout) { x -= p.Wout; ++y; }
                if (y >= p.Hout) { y -= p.Hout; ++n; }
            }
            poff[j] = ((unsigned)((n * p.Hin + y - RATE) * p.Win + x) << p.x_pix_log2) + uoff;
            xyv[j] = ok ? y - RATE : -(1 << 28);
        }
    }
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(xg), 0, (int)((unsigned)p.Mtot << p.x_pix_log2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wg), 0, (int)(64u * (unsigned)p.w_row_bytes), 0x00020000);
    const int xrow0 = wave_u * 64 + l16;
    unsigned xkeep = 0;
    {
        int x = (mtile * 256 + xrow0) % p.Wout;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (x >= RATE) xkeep |= 1u << i;
            if (x < p.Wout - RATE) xkeep |= 1u << (8 + i);
            x += 16;
            if (x >= p.Wout) x -= p.Wout;
        }
    }
    int foff16[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) foff16[pl] = ((pl * 4 + q16) ^ ((l16 >> 1) & 7)) * 16;

    f32x4 acc[4][4];
    W_BIAS(4, 4, ntile * 64)
    half8 AH[2][4], AL[2][4], bh[4], bl[4];

    // (channel block, filter row) of a super-chunk; super-chunks past the end repeat the last one (their DMA is never read)
    auto sc_coords = [&](int s, int& cblk, int& ky) {
        const int sv = s < nsc ? s : nsc - 1;
        cblk = sv / nky;
        ky = ky0 + (sv - cblk * nky);
    };
    auto w_at = [](int cb, int ky) { return (unsigned)((cb * 3 + ky) * 3) * 128u; };      // byte offset of its first weight chunk
    auto w_of = [&](int s) { int cb, ky; sc_coords(s, cb, ky); return w_at(cb, ky); };

    // ---- prologue: patches of super-chunks 0 and 1, weights of chunks 0..3; then the first chunk's first fragments
#define W64_PROLOGUE_PATCH(S_, BUF_)                                                               \
    {                                                                                              \
        int cb_, ky_; sc_coords(S_, cb_, ky_);                                                     \
        const int xdy = (S_) < nsc ? ky_ * RATE : -(1 << 28);                                      \
        const unsigned xsoff = ((unsigned)(ky_ * RATE * p.Win) << p.x_pix_log2) + (unsigned)cb_ * 128u;      \
        W_XA(0, 0); W_XB(0, 0); W64_XC(0, 0, BUF_); W_XA(1, 1); W_XB(1, 1); W64_XC(1, 1, BUF_); W_XA(2, 2); W_XB(2, 2); W64_XC(2, 2, BUF_); \
        W_XA(3, 3); W_XB(3, 3); W64_XC(3, 3, BUF_); W_XA(4, 4); W_XB(4, 4); W64_XC(4, 4, BUF_); W_XA(5, 5); W_XB(5, 5); W64_XC(5, 5, BUF_); \
        W_XA(6, 6); W_XB(6, 6); W64_XC(6, 6, BUF_); W_XA(7, 7); W_XB(7, 7); W64_XC(7, 7, BUF_); W_XA(8, 8); W_XB(8, 8); W64_XC(8, 8, BUF_); \
    }
    W64_PROLOGUE_PATCH(0, 0)
    W64_PROLOGUE_PATCH(1, 1)
#undef W64_PROLOGUE_PATCH
    {
        const unsigned wa = w_of(0), wb = w_of(1);
        { const unsigned wsoff = wa; W64_BDMA(0, 0); W64_BDMA(1, 0); }
        { const unsigned wsoff = wa + 128; W64_BDMA(0, 1); W64_BDMA(1, 1); }
        { const unsigned wsoff = wa + 256; W64_BDMA(0, 2); W64_BDMA(1, 2); }
        { const unsigned wsoff = wb; W64_BDMA(0, 3); W64_BDMA(1, 3); }
    }
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));             // vmcnt(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the zero row's ds_write
    __builtin_amdgcn_s_barrier();
    {
        const unsigned pb_ = lds_u32(As);
        const unsigned na_h = pb_ + (unsigned)(xrow0 * 128 + ((q16 ^ (xrow0 & 6)) * 16));
        const unsigned na_l = pb_ + (unsigned)(xrow0 * 128 + (((4 + q16) ^ (xrow0 & 6)) * 16));
        const unsigned nb0_ = lds_u32(Bs + l16 * 128);
        W_XRD(AH[0][0], 0, na_h, 0) W_XRD(AH[0][1], 1, na_h, 0) W_XRD(AH[0][2], 2, na_h, 0) W_XRD(AH[0][3], 3, na_h, 0)
        W_RD(bh[0], nb0_ + foff16[0], 0); W_RD(bl[0], nb0_ + foff16[1], 0);
        W_XRD(AL[0][0], 0, na_l, 0) W_XRD(AL[0][1], 1, na_l, 0) W_XRD(AL[0][2], 2, na_l, 0) W_XRD(AL[0][3], 3, na_l, 0)
        W_RD(bh[1], nb0_ + foff16[0], 2048); W_RD(bl[1], nb0_ + foff16[1], 2048);
    }
    W_SB;

    // ---- main loop: chunk q = 3 sc + kx alternates between the two fragment sets, so two super-chunks are unrolled
    int wslot = 0;                                        // ring slot of the current chunk's weights
    int pbuf0 = 0, pbuf1 = 1, pbuf2 = 2;                  // patch buffers of super-chunks sc, sc + 1, sc + 2
    // coordinates of super-chunks sc + 1 and sc + 2, walked without divisions
    int cb1, ky1, cb2, ky2;
    sc_coords(1, cb1, ky1);
    sc_coords(2, cb2, ky2);
#define W64_SUPER(P0_, P1_, P2_)                                                                   \
    {                                                                                              \
        const int xdy = sc + 2 < nsc ? ky2 * RATE : -(1 << 28);      /* the super-chunk whose patch this one fetches */ \
        const unsigned xsoff = ((unsigned)(ky2 * RATE * p.Win) << p.x_pix_log2) + (unsigned)cb2 * 128u;       \
        const unsigned w1 = w_at(cb1, ky1), w2 = w_at(cb2, ky2);                                   \
        W64_BODY(0, P0_) W64_BODY(1, P1_) W64_BODY(2, P2_)                                         \
        { const int t_ = pbuf0; pbuf0 = pbuf1; pbuf1 = pbuf2; pbuf2 = t_; }                        \
        cb1 = cb2; ky1 = ky2;                                                                      \
        if (sc + 3 < nsc) { if (++ky2 == ky0 + nky) { ky2 = ky0; ++cb2; } }                        \
        ++sc;                                                                                      \
    }
    int sc = 0;
    while (sc + 2 <= nsc) {
        W64_SUPER(0, 1, 0)
        W64_SUPER(1, 0, 1)
    }
    if (sc < nsc) W64_SUPER(0, 1, 0)
#undef W64_SUPER
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));             // the last chunks' filler DMA must land before the LDS is given back
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // ... and the filler fragment requests before their registers are reused

    // ---- epilogue: interior tile, blocks of 32 channels (checked by the launcher): the split store
    uint8_t* __restrict__ tbase = p.y + (long)mtile * 256 * p.y_ld * 4;
    const unsigned rowb = (unsigned)p.y_ld * 4u;
    const float lo_clamp = p.relu ? 0.f : -65504.f;
    float vmax = 0.f;
    W_STORE(4, 4, wave_u * 64, ntile * 64)
    if (p.range) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        range_note(p.range, vmax, lane == 0);
    }
}
#undef W64_WAIT4
#undef W64_PASS
#undef W64_PASS4
#undef W64_XC
#undef W64_BDMA
#undef W64_BODY




// ---- cnv4 (3x3, dilation 4, 64 -> 128 channels) on 256 x 128 tiles (conv_igemm_h3w128) -----------------------------------------
// cnv4 on conv_igemm_h3's 128x128 tiles (two workgroups per CU, two ring slots) takes 1.6 us per chunk and workgroup for 0.38 us of
// matrix work: 32 KB staged per 3.1 MFLOP, at what a CU takes in from L2.  A 256-row tile with the shared pixel patch stages 11 KB of
// pixels and 16 KB of weights per chunk for twice the work - 2.3 x fewer bytes per FLOP.  (conv_igemm_h3's own 256x128 tile has that
// ratio too and measured level: one workgroup per CU with two ring slots has one chunk in flight.)  Four waves of 128 x 64 outputs,
// conv_igemm_h3w's slots and barrier placement, and as many bytes in flight as the LDS holds: two patch buffers (the next super-chunk's
// patch is issued during the first two chunks of this one) and a weight ring of FIVE slots, four chunks ahead; the wait in front of the
// barrier is counted per tap (first build: three patch buffers, weights two ahead, one chunk's DMA in flight: 1.2 us per chunk).
template <int RATE> struct TileW128 {
    static constexpr int PR = TileW<RATE>::PR, PATCH = PR * 128, BSLOT = 128 * 128, NB = 5, NP = 2;
    static constexpr int LDS_BYTES = NP * PATCH + NB * BSLOT + 4 * 1024 + 128;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per workgroup");
};

#define W128_BDMA(j_, slot_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (w_lds_t*)(Bs + (slot_) * TW::BSLOT + ((j_) * 32 + 8 * wave_u) * 128), 16, boff[j_], wsoff, 0, 0)
#define W128_XC(t_, j_, abuf_)                                                                     \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (w_lds_t*)(((j_) < 8 || (8 * wave_u + (j_) * 32) < PR)                     \
                                                     ? As + (abuf_) * TW::PATCH + ((j_) * 32 + 8 * wave_u) * 128 : xdummy + wave_u * 1024), \
                                             16, ((j_) < 8 || (8 * wave_u + (j_) * 32) < PR) ? xvo##t_ : 0xFFFFFF00u, 0, 0, 0)
// the DMA of one chunk, by tap: kx = 0 five patch slots of the next super-chunk, kx = 1 the other four, kx = 2 none; then the four
// weight pieces of the chunk four ahead.  24 slots of column group 1.
#define W128_DMA_0                                                                                  \
        W_PASS8(AH[CUR__], bh[1], 1, W_XA(0, 0), W_XB(0, 0), W128_XC(0, 0, nabuf), W_XA(1, 1), W_XB(1, 1), W128_XC(1, 1, nabuf), W_XA(2, 2), W_XB(2, 2)) \
        W_PASS8(AH[CUR__], bl[1], 1, W128_XC(2, 2, nabuf), W_XA(3, 3), W_XB(3, 3), W128_XC(3, 3, nabuf), W_XA(4, 4), W_XB(4, 4), W128_XC(4, 4, nabuf), W128_BDMA(0, wdst)) \
        W_PASS8(AL[CUR__], bh[1], 1, W128_BDMA(1, wdst), W128_BDMA(2, wdst), W128_BDMA(3, wdst), W_NOP, W_NOP, W_NOP, W_NOP, W_NOP)
#define W128_DMA_1                                                                                  \
        W_PASS8(AH[CUR__], bh[1], 1, W_XA(5, 5), W_XB(5, 5), W128_XC(5, 5, nabuf), W_XA(6, 6), W_XB(6, 6), W128_XC(6, 6, nabuf), W_XA(7, 7), W_XB(7, 7)) \
        W_PASS8(AH[CUR__], bl[1], 1, W128_XC(7, 7, nabuf), W_XA(8, 8), W_XB(8, 8), W128_XC(8, 8, nabuf), W128_BDMA(0, wdst), W128_BDMA(1, wdst), W128_BDMA(2, wdst), W128_BDMA(3, wdst)) \
        W_PASS(AL[CUR__], bh[1], 1)
#define W128_DMA_2                                                                                  \
        W_PASS8(AH[CUR__], bh[1], 1, W128_BDMA(0, wdst), W128_BDMA(1, wdst), W128_BDMA(2, wdst), W128_BDMA(3, wdst), W_NOP, W_NOP, W_NOP, W_NOP) \
        W_PASS(AH[CUR__], bl[1], 1) W_PASS(AL[CUR__], bh[1], 1)
// vmcnt allowed in front of the barrier of tap kx: the next chunk's weights were issued three chunks ago (the DMA of the newest three
// chunks may be in flight: 8 + 4 + 9 instructions); after kx = 2 the next patch - its last pieces are chunk kx = 1's first DMA
// instructions - must have landed as well: only the weight pieces of chunks kx = 1 and kx = 2 may be in flight
#define W128_VM_0 21
#define W128_VM_1 21
#define W128_VM_2 8
#define W128_BODY(KX_, CUR_)                                                                       \
    {                                                                                              \
        constexpr int NKX_ = ((KX_) + 1) % 3;                                                      \
        constexpr int CUR__ = CUR_;                                                                \
        const unsigned b0_ = lds_u32(Bs + wslot * TW::BSLOT + (wn * 64 + l16) * 128);              \
        const unsigned b_h = b0_ + foff16[0], b_l = b0_ + foff16[1];                               \
        /* weights of chunk q + 4: tap (kx + 1) % 3 of the next super-chunk (of the one after it for kx = 2) -> the slot the chunk before used */ \
        const unsigned wsoff = ((KX_) == 2 ? w2 : w1) + NKX_ * 128;                                \
        const int wdst = wslot == 0 ? TW::NB - 1 : wslot - 1;                                      \
        /* ---- column group 0 */                                                                  \
        W_RD(bh[2], b_h, 2 * 2048); W_RD(bl[2], b_l, 2 * 2048);                                    \
        W_WAIT8(13, AH[CUR_]); W_WAIT1(13, bh[0]);                                                 \
        W_PASS(AH[CUR_], bh[0], 0)                                                                 \
        W_SB;                                                                                      \
        W_WAIT1(12, bl[0]);                                                                        \
        W_PASS(AH[CUR_], bl[0], 0)                                                                 \
        W_SB;                                                                                      \
        W_WAIT8(4, AL[CUR_]);                                                                      \
        W_PASS(AL[CUR_], bh[0], 0)                                                                 \
        W_SB;                                                                                      \
        /* ---- column group 1: the chunk's DMA */                                                 \
        W_RD(bh[3], b_h, 3 * 2048); W_RD(bl[3], b_l, 3 * 2048);                                    \
        W_WAIT2(4, bh[1], bl[1]);                                                                  \
        W128_DMA_##KX_                                                                             \
        W_SB;                                                                                      \
        /* ---- column group 2 */                                                                  \
        W_WAIT2(2, bh[2], bl[2]);                                                                  \
        W_PASS(AH[CUR_], bh[2], 2) W_PASS(AH[CUR_], bl[2], 2) W_PASS(AL[CUR_], bh[2], 2)           \
        W_SB;                                                                                      \
        W_WAIT2(0, bh[3], bl[3]);                                                                  \
        /* ---- column group 3 */                                                                  \
        {                                                                                          \
            __builtin_amdgcn_s_waitcnt((W128_VM_##KX_ & 15) | (7 << 4) | (15 << 8) | ((W128_VM_##KX_ >> 4) << 14));             \
            __builtin_amdgcn_s_barrier();                                                          \
            const unsigned pb_ = lds_u32(As + ((KX_) == 2 ? nabuf : abuf) * TW::PATCH);            \
            const int nrow_ = xrow0 + NKX_ * RATE;                                                 \
            const unsigned na_h = pb_ + (unsigned)(nrow_ * 128 + ((q16 ^ (nrow_ & 6)) * 16));      \
            const unsigned na_l = pb_ + (unsigned)(nrow_ * 128 + (((4 + q16) ^ (nrow_ & 6)) * 16)); \
            const int nws_ = wslot + 1 == TW::NB ? 0 : wslot + 1;                                  \
            const unsigned nb0_ = lds_u32(Bs + nws_ * TW::BSLOT + (wn * 64 + l16) * 128);          \
            const unsigned nb_h = nb0_ + foff16[0], nb_l = nb0_ + foff16[1];                       \
            W_SB;                                                                                  \
            W_PASS8(AH[CUR_], bh[3], 3, W_XRD(AH[(CUR_) ^ 1][0], 0, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][1], 1, na_h, NKX_),       \
                    W_XRD(AH[(CUR_) ^ 1][2], 2, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][3], 3, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][4], 4, na_h, NKX_), \
                    W_XRD(AH[(CUR_) ^ 1][5], 5, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][6], 6, na_h, NKX_), W_XRD(AH[(CUR_) ^ 1][7], 7, na_h, NKX_)) \
            W_PASS8(AH[CUR_], bl[3], 3, W_RD(bh[0], nb_h, 0), W_RD(bl[0], nb_l, 0),                \
                    W_XRD(AL[(CUR_) ^ 1][0], 0, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][1], 1, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][2], 2, na_l, NKX_), \
                    W_XRD(AL[(CUR_) ^ 1][3], 3, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][4], 4, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][5], 5, na_l, NKX_)) \
            W_PASS8(AL[CUR_], bh[3], 3, W_XRD(AL[(CUR_) ^ 1][6], 6, na_l, NKX_), W_XRD(AL[(CUR_) ^ 1][7], 7, na_l, NKX_),       \
                    W_RD(bh[1], nb_h, 2048), W_RD(bl[1], nb_l, 2048), W_NOP, W_NOP, W_NOP, W_NOP)  \
        }                                                                                          \
        W_SB;                                                                                      \
        wslot = wslot + 1 == TW::NB ? 0 : wslot + 1;                                               \
    }

template <int LAYER, int RATE>
__global__ __launch_bounds__(256, 1) void conv_igemm_h3w128(ConvParamsH p) {
    using TW = TileW128<RATE>;
    constexpr int PR = TW::PR;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_w8[];
    uint8_t* As = smem_w8;                              // [2][PR][128] pixel patches
    uint8_t* Bs = smem_w8 + TW::NP * TW::PATCH;         // [5][128][128] weight ring
    uint8_t* xdummy = Bs + TW::NB * TW::BSLOT;          // 1 KB per wave: where a patch slot past the patch's end lands
    const unsigned xzero = lds_u32(xdummy + 4 * 1024);  // 128 zero bytes (what a tap reads outside its image row)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave_u >> 1, wn = wave_u & 1;
    const int l16 = lane & 15, q16 = lane >> 4;
    if (tid < 8) *reinterpret_cast<float4*>(xdummy + 4 * 1024 + tid * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

    const int mtile = p.mtile0 + xcd_remap(blockIdx.x, gridDim.x);       // one N tile: N = 128
    const uint8_t* __restrict__ xg = p.x + p.x_boff;
    const uint8_t* __restrict__ wg = p.w;
    int ky0, nky;
    {
        const FilterRows fr = valid_filter_rows(mtile * 256, mtile * 256 + 255, p.Hout, p.Wout, p.Hin, 1, p.pad_t, p.rate);
        ky0 = __builtin_amdgcn_readfirstlane(fr.ky0);
        nky = __builtin_amdgcn_readfirstlane(fr.nky);
    }
    const int nsc = (p.nchunks / 9) * nky;

    const int r0 = tid >> 3;
    unsigned boff[4];
    {
        const int u = (tid & 7) ^ ((r0 >> 1) & 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) boff[j] = (unsigned)((r0 + 32 * j) * (int)p.w_row_bytes + u * 16);
    }
    const int ux = (tid & 7) ^ (r0 & 6);
    unsigned poff[9];
    int xyv[9];
    {
        const int hw = p.Hout * p.Wout;
        const unsigned uoff = (unsigned)((ux >> 2) * 64 + (ux & 3) * 16);
        const int q32 = 32 / p.Wout, r32 = 32 - q32 * p.Wout;
        int n = 0, y = 0, x = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int row = r0 + 32 * j;
            const int f = mtile * 256 - RATE + row;
            const bool ok = row < PR && f >= 0 && f < p.Mtot;
            if (j < 2) {
                const int fc = f >= 0 ? f : 0;
                n = fc / hw;
                const int rem = fc - n * hw;
                y = rem / p.Wout;
                x = rem - y * p.Wout;
            } else {
                x += r32; y += q32;
                if (x >= p.Wout) { x -= p.Wout; ++y; }
                if (y >= p.Hout) { y -= p.Hout; ++n; }
            }
            poff[j] = ((unsigned)((n * p.Hin + y - RATE) * p.Win + x) << p.x_pix_log2) + uoff;
            xyv[j] = ok ? y - RATE : -(1 << 28);
        }
    }
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(xg), 0, (int)((unsigned)p.Mtot << p.x_pix_log2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wg), 0, (int)(128u * (unsigned)p.w_row_bytes), 0x00020000);
    const int xrow0 = wm * 128 + l16;
    unsigned xkeep = 0;
    {
        int x = (mtile * 256 + xrow0) % p.Wout;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (x >= RATE) xkeep |= 1u << i;
            if (x < p.Wout - RATE) xkeep |= 1u << (8 + i);
            x += 16;
            if (x >= p.Wout) x -= p.Wout;
        }
    }
    int foff16[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) foff16[pl] = ((pl * 4 + q16) ^ ((l16 >> 1) & 7)) * 16;

    f32x4 acc[8][4];
    W_BIAS(8, 4, wn * 64)
    half8 AH[2][8], AL[2][8], bh[4], bl[4];

    auto sc_coords = [&](int s, int& cblk, int& ky) {
        const int sv = s < nsc ? s : nsc - 1;
        cblk = sv / nky;
        ky = ky0 + (sv - cblk * nky);
    };
    auto w_at = [](int cb, int ky) { return (unsigned)((cb * 3 + ky) * 3) * 128u; };

    // ---- prologue: the first patch, the weights of chunks 0..3; then the first chunk's first fragments
    int cb1, ky1, cb2, ky2;                               // coordinates of super-chunks sc + 1 and sc + 2, walked without divisions
    sc_coords(1, cb1, ky1);
    sc_coords(2, cb2, ky2);
    {
        const int xdy = ky0 * RATE;
        const unsigned xsoff = (unsigned)(ky0 * RATE * p.Win) << p.x_pix_log2;
        W_XA(0, 0); W_XB(0, 0); W128_XC(0, 0, 0); W_XA(1, 1); W_XB(1, 1); W128_XC(1, 1, 0); W_XA(2, 2); W_XB(2, 2); W128_XC(2, 2, 0);
        W_XA(3, 3); W_XB(3, 3); W128_XC(3, 3, 0); W_XA(4, 4); W_XB(4, 4); W128_XC(4, 4, 0); W_XA(5, 5); W_XB(5, 5); W128_XC(5, 5, 0);
        W_XA(6, 6); W_XB(6, 6); W128_XC(6, 6, 0); W_XA(7, 7); W_XB(7, 7); W128_XC(7, 7, 0); W_XA(8, 8); W_XB(8, 8); W128_XC(8, 8, 0);
        const unsigned wa = w_at(0, ky0), wb = w_at(cb1, ky1);
        { const unsigned wsoff = wa; W128_BDMA(0, 0); W128_BDMA(1, 0); W128_BDMA(2, 0); W128_BDMA(3, 0); }
        { const unsigned wsoff = wa + 128; W128_BDMA(0, 1); W128_BDMA(1, 1); W128_BDMA(2, 1); W128_BDMA(3, 1); }
        { const unsigned wsoff = wa + 256; W128_BDMA(0, 2); W128_BDMA(1, 2); W128_BDMA(2, 2); W128_BDMA(3, 2); }
        { const unsigned wsoff = wb; W128_BDMA(0, 3); W128_BDMA(1, 3); W128_BDMA(2, 3); W128_BDMA(3, 3); }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));             // vmcnt(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the zero row's ds_write
    __builtin_amdgcn_s_barrier();
    {
        const unsigned pb_ = lds_u32(As);
        const unsigned na_h = pb_ + (unsigned)(xrow0 * 128 + ((q16 ^ (xrow0 & 6)) * 16));
        const unsigned na_l = pb_ + (unsigned)(xrow0 * 128 + (((4 + q16) ^ (xrow0 & 6)) * 16));
        const unsigned nb0_ = lds_u32(Bs + (wn * 64 + l16) * 128);
        W_XRD(AH[0][0], 0, na_h, 0) W_XRD(AH[0][1], 1, na_h, 0) W_XRD(AH[0][2], 2, na_h, 0) W_XRD(AH[0][3], 3, na_h, 0)
        W_XRD(AH[0][4], 4, na_h, 0) W_XRD(AH[0][5], 5, na_h, 0) W_XRD(AH[0][6], 6, na_h, 0) W_XRD(AH[0][7], 7, na_h, 0)
        W_RD(bh[0], nb0_ + foff16[0], 0); W_RD(bl[0], nb0_ + foff16[1], 0);
        W_XRD(AL[0][0], 0, na_l, 0) W_XRD(AL[0][1], 1, na_l, 0) W_XRD(AL[0][2], 2, na_l, 0) W_XRD(AL[0][3], 3, na_l, 0)
        W_XRD(AL[0][4], 4, na_l, 0) W_XRD(AL[0][5], 5, na_l, 0) W_XRD(AL[0][6], 6, na_l, 0) W_XRD(AL[0][7], 7, na_l, 0)
        W_RD(bh[1], nb0_ + foff16[0], 2048); W_RD(bl[1], nb0_ + foff16[1], 2048);
    }
    W_SB;

    // ---- main loop
    int wslot = 0;                                        // ring slot of the current chunk's weights
#define W128_SUPER(P0_, P1_, P2_)                                                                  \
    {                                                                                              \
        const int xdy = sc + 1 < nsc ? ky1 * RATE : -(1 << 28);      /* the super-chunk whose patch this one fetches */ \
        const unsigned xsoff = ((unsigned)(ky1 * RATE * p.Win) << p.x_pix_log2) + (unsigned)cb1 * 128u;       \
        const unsigned w1 = w_at(cb1, ky1), w2 = w_at(cb2, ky2);                                   \
        const int abuf = sc & 1, nabuf = abuf ^ 1;                                                 \
        W128_BODY(0, P0_) W128_BODY(1, P1_) W128_BODY(2, P2_)                                      \
        cb1 = cb2; ky1 = ky2;                                                                      \
        if (sc + 3 < nsc) { if (++ky2 == ky0 + nky) { ky2 = ky0; ++cb2; } }                        \
        ++sc;                                                                                      \
    }
    int sc = 0;
    while (sc + 2 <= nsc) {
        W128_SUPER(0, 1, 0)
        W128_SUPER(1, 0, 1)
    }
    if (sc < nsc) W128_SUPER(0, 1, 0)
#undef W128_SUPER
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));             // the last chunks' filler DMA must land before the LDS is given back
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // ... and the filler fragment requests before their registers are reused

    // ---- epilogue: interior tile, blocks of 32 channels (checked by the launcher): the split store
    uint8_t* __restrict__ tbase = p.y + (long)mtile * 256 * p.y_ld * 4;
    const unsigned rowb = (unsigned)p.y_ld * 4u;
    const float lo_clamp = p.relu ? 0.f : -65504.f;
    float vmax = 0.f;
    W_STORE(8, 4, wm * 128, wn * 64)
    if (p.range) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        range_note(p.range, vmax, lane == 0);
    }
}
#undef W128_BDMA
#undef W128_XC
#undef W128_DMA_0
#undef W128_DMA_1
#undef W128_DMA_2
#undef W128_VM_0
#undef W128_VM_1
#undef W128_VM_2
#undef W128_BODY

#undef W_RD
#undef W_XRD
#undef W_WAIT8
#undef W_WAIT1
#undef W_WAIT2
#undef W_PASS
#undef W_MFMA1
#undef W_BIAS
#undef W_STORE
#undef W_SLOT
#undef W_PASS8
#undef W_NOP
#undef W_XA
#undef W_XB
#undef W_XC
#undef W_SB
#undef W_BDMA
#undef W_BODY

}  // namespace davo
