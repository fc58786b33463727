// conv_igemm.h — slim.conv2d(padding='SAME') as an implicit GEMM on the gfx950 FP32 matrix
// cores (v_mfma_f32_32x32x2_f32).  One template covers all eight PoseNN layers
// (reference nets/posenn.py:211-215,238-240):
//
//     Y[m][n] = relu( sum_k A[m][k] * Wp[n][k] + bias[n] )
//       m = (image, oy, ox) flattened over the whole batch     (GEMM M)
//       n = output channel                                      (GEMM N)
//       k = (tap, ci), tap = ky*KS+kx, ci < Cin (power of two)   (GEMM K, padded to 32)
//
// A is never materialised: a k-chunk of 32 is one (or several, when Cin < 32) filter taps
// of contiguous NHWC channels, gathered straight from the input activation with TF's
// asymmetric SAME padding turned into a bounds test (zero fill).  Weights are re-laid-out
// once at load time to Wp[Cout_pad][K_pad] (k contiguous) so both operands are staged with
// the same 16-byte rows.
//
// Tile: 128 (M) x BN (N) per 256-thread workgroup (4 waves, one per SIMD), K step 32,
// LDS double-buffered with one barrier per step; rows padded to 36 floats so the
// ds_read_b128 fragment reads are bank-conflict free (36 = 4*9, 9 odd -> 16 distinct slots).
// Each lane reads 4 consecutive k of its row with one ds_read_b128 and feeds them to four
// successive MFMAs (lane half h supplies k = 8g+4h+j to MFMA j of group g for both operands,
// so A and B agree on the contraction order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "params.h"

#ifndef DAVO_F32_ABLATE_A
#define DAVO_F32_ABLATE_A 0    /* timing experiments only: every pixel load reads the zero line / every weight load the first chunk */
#define DAVO_F32_ABLATE_B 0
#endif
#ifndef DAVO_F32_EARLY_STORE
#define DAVO_F32_EARLY_STORE 1      /* 128-column tile: two staging register sets, LDS stores ahead of the matrix phase (below) */
#endif
#ifndef DAVO_F32_FRAG0
#define DAVO_F32_FRAG0 1            /* early-store loop: a chunk's first fragments are requested ahead of its LDS stores */
#endif
#ifndef DAVO_F32_FAST_EPILOGUE
#define DAVO_F32_FAST_EPILOGUE 1
#endif
#ifndef DAVO_F32_FRAG_AHEAD
#define DAVO_F32_FRAG_AHEAD 0       /* (experiment) group g + 1's fragments pinned ahead of group g's MFMAs with sched_barriers: +1.4 % SLOWER than the compiler's own interleaving */
#endif
#ifndef DAVO_F32_EARLY_STORE_MIN_BN
#define DAVO_F32_EARLY_STORE_MIN_BN 32      /* every tile but cnv1's 16-column one: measured on the narrow remainder tiles too (-2..4 %) */
#endif
#ifndef DAVO_F32_ACC_AGPR
#define DAVO_F32_ACC_AGPR 0         /* (experiment) accumulators pinned into the accumulation registers: inline-asm matrix instructions */
#endif
#if DAVO_F32_ACC_AGPR
#define DAVO_MFMA32(acc_, a_, b_) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc_) : "v"(a_), "v"(b_))
#else
#define DAVO_MFMA32(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, acc_, 0, 0, 0)
#endif
// the 4 k of a fragment group x the wave's TM x TN accumulators.  Builtins: the compiler orders them (it interleaves accumulators);
// inline asm keeps the written order, so the accumulators go innermost there (a dependent matrix instruction straight behind its
// producer needs a wait state)
#if DAVO_F32_ACC_AGPR
#define DAVO_F4C(v_, c_) ((c_) == 0 ? (v_).x : (c_) == 1 ? (v_).y : (c_) == 2 ? (v_).z : (v_).w)
#define DAVO_MFMA_TILE(FA_, FB_)                                                                   \
            _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_)                                       \
                _Pragma("unroll") for (int i_ = 0; i_ < T::TM; ++i_)                               \
                    _Pragma("unroll") for (int j_ = 0; j_ < T::TN; ++j_)                           \
                        DAVO_MFMA32(acc[i_][j_], DAVO_F4C(FA_, c_), DAVO_F4C(FB_, c_));
#else
#define DAVO_MFMA_TILE(FA_, FB_)                                                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < T::TM; ++i_)                                   \
                _Pragma("unroll") for (int j_ = 0; j_ < T::TN; ++j_) {                             \
                    DAVO_MFMA32(acc[i_][j_], (FA_).x, (FB_).x);                                    \
                    DAVO_MFMA32(acc[i_][j_], (FA_).y, (FB_).y);                                    \
                    DAVO_MFMA32(acc[i_][j_], (FA_).z, (FB_).z);                                    \
                    DAVO_MFMA32(acc[i_][j_], (FA_).w, (FB_).w);                                    \
                }
#endif
#ifndef DAVO_F32_ABLATE_BARRIER
#define DAVO_F32_ABLATE_BARRIER 0   /* timing experiments only */
#endif

namespace davo {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// XCD-aware, bijective remap of the linear workgroup id: workgroups b, b+8, b+16, ... share an
// XCD (and its 4 MiB L2), so give each XCD a contiguous run of tiles = neighbouring pixels of
// the same images, whose dilated taps overlap.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

// LAYER only names the instantiation (0 = generic, 1..7 = cnv1..cnv7) so that rocprofv3's
// per-kernel statistics separate the layers that share a tile shape (cnv4/cnv5/cnv6).
template <int KS, int STRIDE, int BN, int LAYER>
__device__ __forceinline__ void conv_igemm_f32_body(const ConvParams& p, const int wg_x, const int nwg_x, const int wg_y) {
    using T = Tile<BN>;
    constexpr bool N16 = BN == 16;             // 16 output columns: four waves of 32 x 16 on v_mfma_f32_16x16x4_f32
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;           // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wid / T::WN, wn = wid % T::WN;

    const int tile_i = xcd_remap(wg_x, nwg_x);
    const int tile = p.tile_order ? __builtin_amdgcn_readfirstlane(p.tile_order[tile_i]) : tile_i;
    const int ntile = tile % p.ntiles_n, mtile = p.mtile0 + tile / p.ntiles_n;
    const int grp = wg_y;
    const float* __restrict__ xg = p.x + p.x_coff + grp * p.g_x_coff;
    const float* __restrict__ wg = p.w + grp * p.g_w + (long)ntile * BN * p.Kpad;
    const float* __restrict__ bg = p.bias + grp * p.g_bias + ntile * BN;
    float* __restrict__ yg = p.y + p.y_coff + grp * p.g_y_coff;

    // ---- staging assignment: thread -> (row r0+32j, 4 consecutive k at kk) ---------------
    const int r0 = tid >> 3, kk = (tid & 7) * 4;
    constexpr int RPP = T::RPP;                // rows per staging pass: 32 (four waves) | 64 (eight)
    int iy0[4], ix0[4], pix0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j >= T::A_LOADS) { iy0[j] = -(1 << 28); ix0[j] = 0; pix0[j] = 0; continue; }
        const int m = mtile * BM + r0 + RPP * j;
        if (m < p.M) {
            const int hw = p.Hout * p.Wout;
            const int n = m / hw, rem = m - n * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            iy0[j] = oy * STRIDE - p.pad_t;
            ix0[j] = ox * STRIDE - p.pad_l;
            pix0[j] = n * p.Hin * p.Win;
        } else {
            iy0[j] = -(1 << 28);               // every tap lands out of bounds -> zero rows
            ix0[j] = 0;
            pix0[j] = 0;
        }
    }
    const int cmask = (1 << p.cin_log2) - 1;

    // Staging registers are named scalars, not arrays: hipcc (ROCm 7.2) keeps a conditionally
    // written float4 array in scratch and serialises every load behind a scratch store.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    // second set (EARLY_STORE below): chunk q + 2 is in flight while chunk q + 1 goes from the first set into LDS
    float4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
    sa0 = sa1 = sa2 = sa3 = sb0 = sb1 = sb2 = sb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    // Pixel addresses are carried from chunk to chunk: with Cin >= 32 a tap spans Cin / 32 consecutive chunks whose loads differ
    // by 32 channels (128 bytes) only, so the ~17 instructions per row of the bounds test and the 64-bit offset (several
    // quarter-rate integer multiplies) are paid on a tap's first chunk and the others add a step (0 for a row that reads the
    // zero line).  cnv6 (8 chunks per tap): 68 -> ~20 address instructions per chunk on average.
    const float *pa0 = p.zeros, *pa1 = p.zeros, *pa2 = p.zeros, *pa3 = p.zeros;
    int st0 = 0, st1 = 0, st2 = 0, st3 = 0;                       // floats to the same pixel's next 32 channels, or 0
    const int cpt = p.cin_log2 >= 5 ? 1 << (p.cin_log2 - 5) : 1;  // chunks per tap
    int l_cb = 0;                                                 // chunk inside the current tap (every launch starts on a tap boundary)
#define DAVO_ADDR_A(j_, ptr_, st_)                                                                 \
    {                                                                                              \
        const int iy = iy0[j_] + dy, ix = ix0[j_] + dx;                                            \
        const bool ok = tap_ok && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win && !DAVO_F32_ABLATE_A; \
        const long off = (long)(pix0[j_] + iy * p.Win + ix) * p.x_ld + c;                          \
        /* unconditional load; a padded tap reads a zero line: no branch around the load */       \
        ptr_ = ok ? xg + off : p.zeros;                                                            \
        st_ = ok ? BK : 0;                                                                         \
    }
#define DAVO_LOAD_CHUNK(q_) DAVO_LOAD_CHUNK_R(q_, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3)
#define DAVO_LOAD_CHUNK_R(q_, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3)                              \
    {                                                                                              \
        const int kg = (q_) * BK + kk;                                                             \
        if (l_cb == 0) {                                                                           \
            const int tap = kg >> p.cin_log2, c = kg & cmask;                                      \
            const int ky = tap / KS, kx = tap - ky * KS;                                           \
            const int dy = ky * p.rate, dx = kx * p.rate;                                          \
            const bool tap_ok = tap < p.ntaps;                                                     \
            DAVO_ADDR_A(0, pa0, st0) DAVO_ADDR_A(1, pa1, st1)                                      \
            if constexpr (T::A_LOADS > 2) { DAVO_ADDR_A(2, pa2, st2) DAVO_ADDR_A(3, pa3, st3) }    \
        } else {                                                                                   \
            pa0 += st0; pa1 += st1;                                                                \
            if constexpr (T::A_LOADS > 2) { pa2 += st2; pa3 += st3; }                              \
        }                                                                                          \
        l_cb = l_cb + 1 == cpt ? 0 : l_cb + 1;                                                     \
        ra0 = *reinterpret_cast<const float4*>(pa0);                                               \
        ra1 = *reinterpret_cast<const float4*>(pa1);                                               \
        if constexpr (T::A_LOADS > 2) {                                                            \
            ra2 = *reinterpret_cast<const float4*>(pa2);                                           \
            ra3 = *reinterpret_cast<const float4*>(pa3);                                           \
        }                                                                                          \
        const float* wrow = wg + (long)r0 * p.Kpad + (DAVO_F32_ABLATE_B ? kk : kg);               \
        if (!N16 || r0 < 16) rb0 = *reinterpret_cast<const float4*>(wrow);                         \
        if constexpr (T::NB_LOADS > 1) rb1 = *reinterpret_cast<const float4*>(wrow + (long)RPP * p.Kpad); \
        if constexpr (T::NB_LOADS > 2) {                                                           \
            rb2 = *reinterpret_cast<const float4*>(wrow + 2L * RPP * p.Kpad);                      \
            rb3 = *reinterpret_cast<const float4*>(wrow + 3L * RPP * p.Kpad);                      \
        }                                                                                          \
    }
#define DAVO_STORE_CHUNK(buf_) DAVO_STORE_CHUNK_R(buf_, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3)
#define DAVO_STORE_CHUNK_R(buf_, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3)                           \
    {                                                                                              \
        float* a_ = As + (buf_) * BM * LDK + r0 * LDK + kk;                                        \
        float* b_ = Bs + (buf_) * BN * LDK + r0 * LDK + kk;                                        \
        *reinterpret_cast<float4*>(a_) = ra0;                                                      \
        *reinterpret_cast<float4*>(a_ + RPP * LDK) = ra1;                                          \
        if constexpr (T::A_LOADS > 2) {                                                            \
            *reinterpret_cast<float4*>(a_ + 2 * RPP * LDK) = ra2;                                  \
            *reinterpret_cast<float4*>(a_ + 3 * RPP * LDK) = ra3;                                  \
        }                                                                                          \
        if (!N16 || r0 < 16) *reinterpret_cast<float4*>(b_) = rb0;                                 \
        if constexpr (T::NB_LOADS > 1) *reinterpret_cast<float4*>(b_ + RPP * LDK) = rb1;           \
        if constexpr (T::NB_LOADS > 2) {                                                           \
            *reinterpret_cast<float4*>(b_ + 2 * RPP * LDK) = rb2;                                  \
            *reinterpret_cast<float4*>(b_ + 3 * RPP * LDK) = rb3;                                  \
        }                                                                                          \
    }

    // The accumulators start at the bias (a lane's 16 results of a tile share one output channel),
    // so the epilogue issues no load: a load there would make hipcc put `s_waitcnt vmcnt(0)` in
    // front of every guarded store, and on gfx950 vmcnt counts stores too — the epilogue's stores
    // would complete one by one.
    f32x16 acc[T::TM][T::TN];
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t acc16[2];                          // N16: rows 16 i + 4 (lane >> 4) + r of the wave's 32, column lane & 15
    const int l16 = lane & 15, q16 = lane >> 4;
    if constexpr (N16) {
        const float bv = bg[l16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[i][r] = bv;
    } else {
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const float bv = bg[wn * T::TN * 32 + j * 32 + li];
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }
    }

    // chunks of filter rows that only see zero padding for every pixel of this tile are not walked (params.h,
    // valid_filter_rows): here chunks run tap by tap, so a filter row is one contiguous run of 3 * Cin / 32 chunks.
    // The skipped terms are exact zeros: the float32 fma chain of every output is the oracle's without them.
    int q0 = 0, q1 = p.nchunks;
    if constexpr (KS == 3) {
        if (p.cin_log2 >= 5) {
            const FilterRows fr = valid_filter_rows(mtile * BM, min(mtile * BM + BM, p.M) - 1, p.Hout, p.Wout, p.Hin, STRIDE, p.pad_t, p.rate);
            const int per_row = 3 << (p.cin_log2 - 5);
            q0 = __builtin_amdgcn_readfirstlane(fr.ky0 * per_row);
            q1 = __builtin_amdgcn_readfirstlane((fr.ky0 + fr.nky) * per_row);
        }
    }
    DAVO_LOAD_CHUNK(q0)
    DAVO_STORE_CHUNK(0)
    __syncthreads();

    // N16: lane (l16, q16) feeds row / column l16 and k = 16 g + 4 q16 + t of the chunk to step t of group g (any assignment of the
    // chunk's 32 k to (group, step, lane quarter) is a valid order of the same float32 fma chain, fixed for every batch size)
#define DAVO_COMPUTE16(buf_)                                                                       \
    {                                                                                              \
        const float* a = As + (buf_) * BM * LDK + (wm * 32 + l16) * LDK + 4 * q16;                 \
        const float* b = Bs + (buf_) * BN * LDK + l16 * LDK + 4 * q16;                             \
        _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                            \
            const float4 fb = *reinterpret_cast<const float4*>(b + g * 16);                        \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                        \
                const float4 fa = *reinterpret_cast<const float4*>(a + i * 16 * LDK + g * 16);     \
                acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb.x, acc16[i], 0, 0, 0);    \
                acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb.y, acc16[i], 0, 0, 0);    \
                acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, fb.z, acc16[i], 0, 0, 0);    \
                acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, fb.w, acc16[i], 0, 0, 0);    \
            }                                                                                      \
        }                                                                                          \
    }
#define DAVO_COMPUTE(buf_)                                                                         \
    if constexpr (N16) DAVO_COMPUTE16(buf_) else                                                   \
    {                                                                                              \
        const float* a = As + (buf_) * BM * LDK + (wm * T::TM * 32 + li) * LDK + 4 * lh;           \
        const float* b = Bs + (buf_) * BN * LDK + (wn * T::TN * 32 + li) * LDK + 4 * lh;           \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                            \
            float4 fa[T::TM], fb[T::TN];                                                           \
            _Pragma("unroll") for (int i = 0; i < T::TM; ++i)                                      \
                fa[i] = *reinterpret_cast<const float4*>(a + i * 32 * LDK + g * 8);                \
            _Pragma("unroll") for (int j = 0; j < T::TN; ++j)                                      \
                fb[j] = *reinterpret_cast<const float4*>(b + j * 32 * LDK + g * 8);                \
            DAVO_MFMA_TILE(fa[i_], fb[j_])                                                         \
        }                                                                                          \
    }

    // the same with the fragments of group 0 requested ahead (DAVO_FRAG0): the LDS serves its queue in order, so behind the eight stores
    // of the early-store loop the first fragments of a chunk arrived ~100 cycles late; now they are requested first
#define DAVO_FRAG0(buf_)                                                                           \
    {                                                                                              \
        const float* a = As + (buf_) * BM * LDK + (wm * T::TM * 32 + li) * LDK + 4 * lh;           \
        const float* b = Bs + (buf_) * BN * LDK + (wn * T::TN * 32 + li) * LDK + 4 * lh;           \
        _Pragma("unroll") for (int i = 0; i < T::TM; ++i) pf_a[i] = *reinterpret_cast<const float4*>(a + i * 32 * LDK); \
        _Pragma("unroll") for (int j = 0; j < T::TN; ++j) pf_b[j] = *reinterpret_cast<const float4*>(b + j * 32 * LDK); \
    }
#define DAVO_COMPUTE_PF(buf_) DAVO_COMPUTE_PF_G(buf_, 0, 4)
#define DAVO_COMPUTE_PF_G(buf_, G0_, G1_)                                                           \
    {                                                                                              \
        const float* a = As + (buf_) * BM * LDK + (wm * T::TM * 32 + li) * LDK + 4 * lh;           \
        const float* b = Bs + (buf_) * BN * LDK + (wn * T::TN * 32 + li) * LDK + 4 * lh;           \
        float4 fa[2][T::TM], fb[2][T::TN];                                                         \
        _Pragma("unroll") for (int i = 0; i < T::TM; ++i) fa[0][i] = pf_a[i];                      \
        _Pragma("unroll") for (int j = 0; j < T::TN; ++j) fb[0][j] = pf_b[j];                      \
        _Pragma("unroll") for (int g = (G0_); g < (G1_); ++g) {                                    \
            /* group g + 1's fragments are requested before group g's MFMAs are queued (DAVO_F32_FRAG_AHEAD) */ \
            if (DAVO_F32_FRAG_AHEAD && g + 1 < 4) {                                                \
                _Pragma("unroll") for (int i = 0; i < T::TM; ++i)                                  \
                    fa[(g + 1) & 1][i] = *reinterpret_cast<const float4*>(a + i * 32 * LDK + (g + 1) * 8); \
                _Pragma("unroll") for (int j = 0; j < T::TN; ++j)                                  \
                    fb[(g + 1) & 1][j] = *reinterpret_cast<const float4*>(b + j * 32 * LDK + (g + 1) * 8); \
                __builtin_amdgcn_sched_barrier(0);                                                 \
            }                                                                                      \
            if (!DAVO_F32_FRAG_AHEAD && g > 0) {                                                   \
                _Pragma("unroll") for (int i = 0; i < T::TM; ++i)                                  \
                    fa[g & 1][i] = *reinterpret_cast<const float4*>(a + i * 32 * LDK + g * 8);     \
                _Pragma("unroll") for (int j = 0; j < T::TN; ++j)                                  \
                    fb[g & 1][j] = *reinterpret_cast<const float4*>(b + j * 32 * LDK + g * 8);     \
            }                                                                                      \
            DAVO_MFMA_TILE(fa[g & 1][i_], fb[g & 1][j_])                                           \
            if (DAVO_F32_FRAG_AHEAD) __builtin_amdgcn_sched_barrier(0);                            \
        }                                                                                          \
    }
    float4 pf_a[T::TM], pf_b[T::TN];

    // EARLY_STORE (the 128-column tile, cnv4..cnv7's main launches): two sets of staging registers.  The loads of chunk q + 2 are issued
    // at the top of chunk q, and chunk q + 1 - loaded a whole chunk ago - goes into the idle LDS buffer BEFORE chunk q's matrix phase
    // instead of behind it: the eight LDS stores and their latency pass under the wave's own 64 MFMAs, and what is left between the last
    // MFMA and the barrier is nothing.  (One set: loads at the top, matrix phase, then a staircase of vmcnt waits with the stores, an
    // lgkmcnt(0), the barrier - ~500 cycles per chunk in which the SIMD only works if its other wave happens to be in ITS matrix
    // phase.)  The idle buffer was last read in chunk q - 1, and every wave has passed the barrier behind that chunk.
    constexpr bool EARLY_STORE = DAVO_F32_EARLY_STORE && BN >= DAVO_F32_EARLY_STORE_MIN_BN;
    if constexpr (EARLY_STORE) {
        int q = q0;
        if (q + 1 < q1) DAVO_LOAD_CHUNK_R(q + 1, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3)        // chunk q0 + 1 -> second set
#define DAVO_EARLY_STEP(buf_, RS_, RL_)                                                             \
        {                                                                                          \
            if (DAVO_F32_FRAG0) { DAVO_FRAG0(buf_) __builtin_amdgcn_sched_barrier(0); }            \
            DAVO_STORE_CHUNK_R((buf_) ^ 1, RS_)                        /* chunk q + 1, landed long ago */ \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (q + 2 < q1) DAVO_LOAD_CHUNK_R(q + 2, RL_)              /* flies under this chunk and the next */ \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            DAVO_PRIO_UP(DAVO_MMPRIO);                                                             \
            if (DAVO_F32_FRAG0) DAVO_COMPUTE_PF(buf_) else DAVO_COMPUTE(buf_)                      \
            DAVO_PRIO_DOWN(DAVO_MMPRIO);                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (!DAVO_F32_ABLATE_BARRIER) __syncthreads();                                         \
            ++q;                                                                                   \
        }
#define DAVO_SET_S sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3
#define DAVO_SET_R ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3
        while (q + 2 < q1) {
            DAVO_EARLY_STEP(0, DAVO_SET_S, DAVO_SET_R)
            DAVO_EARLY_STEP(1, DAVO_SET_R, DAVO_SET_S)
        }
        // q + 2 >= q1 here and (q - q0) is even: one or two chunks are left
        if (q + 1 < q1) {
            DAVO_EARLY_STEP(0, DAVO_SET_S, DAVO_SET_R)
            DAVO_COMPUTE(1)
        } else {
            DAVO_COMPUTE(0)
        }
#undef DAVO_EARLY_STEP
#undef DAVO_SET_S
#undef DAVO_SET_R
    } else {
    for (int q = q0; q + 1 < q1; ++q) {
        const int buf = (q - q0) & 1;
        DAVO_LOAD_CHUNK(q + 1)                           // global loads fly under the MFMAs
        __builtin_amdgcn_sched_barrier(0);               // keep hipcc from sinking them to the stores
        DAVO_PRIO_UP(DAVO_MMPRIO);                       // the wave that has MFMAs to issue wins the arbitration (params.h)
        DAVO_COMPUTE(buf)
        DAVO_PRIO_DOWN(DAVO_MMPRIO);
        __builtin_amdgcn_sched_barrier(0);
        DAVO_STORE_CHUNK(buf ^ 1)
        if (!DAVO_F32_ABLATE_BARRIER) __syncthreads();
    }
    DAVO_COMPUTE((q1 - 1 - q0) & 1)
    }

#if DAVO_F32_ACC_AGPR
    if constexpr (!N16) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j) asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[i][j]));
    }
#endif
    if constexpr (N16) {       // C/D layout of 16x16x4: col = lane & 15, row = 4 (lane >> 4) + r
        const int n = ntile * BN + l16;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mtile * BM + wm * 32 + i * 16 + 4 * q16 + r;
                float v = acc16[i][r];
                if (p.relu) v = fmaxf(v, 0.f);
                if (n < p.Cout && m < p.M) yg[(long)m * p.y_ld + n] = v;
            }
        return;
    }
    // ---- pose head in the epilogue (cnv7, float32 mode): fixed summation order -> bitwise reproducible; pose_from_tiles adds the tiles
    if (p.pose_w) {
        const int row0 = mtile * BM;
        const int img0 = row0 / p.pose_P;
        const int split_row = (img0 + 1) * p.pose_P - row0;          // tile rows >= split_row: next image
        float q[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            const int n = ntile * BN + wn * T::TN * 32 + j * 32 + li;
            const bool n_ok = n < p.Cout;
            const float* wp = p.pose_w + ((long)grp * p.Cout + (n_ok ? n : 0)) * 3;
            const float w0 = n_ok ? wp[0] : 0.f, w1 = n_ok ? wp[1] : 0.f, w2 = n_ok ? wp[2] : 0.f;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * T::TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = fmaxf(acc[i][j][r], 0.f);
                    if (row0 + row >= p.M) v = 0.f;
                    if (row < split_row) s0 += v; else s1 += v;
                }
            q[0] += s0 * w0; q[1] += s0 * w1; q[2] += s0 * w2;
            q[3] += s1 * w0; q[4] += s1 * w1; q[5] += s1 * w2;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q[k] += __shfl_down(q[k], o, 64);
        __syncthreads();                                             // every wave is done with the staging buffers
        float* red = smem;                                           // [4 waves][6]
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[wid * 6 + k] = q[k];
        __syncthreads();
        constexpr int UNITS = BN >= 32 ? BN / 32 : 1;                // 32-column units of this tile
        if (tid < 6 * UNITS) {
            const int unit = tid / 6, k = tid - unit * 6;
            float t = 0.f;
            if (unit == 0)
                for (int w = 0; w < T::WAVES; ++w) t += red[w * 6 + k];
            p.pose_partial[(((long)grp * p.pose_mt + mtile) * 8 + ntile * UNITS + unit) * 6 + k] = t;
        }
        return;
    }
    // ---- epilogue: bias + ReLU, C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // Interior tile (every row < M, every column < Cout: all but a launch's last tile row): a uniform 64-bit tile base and 32-bit
    // offsets, no bounds tests - the general loop below pays a 64-bit address product and two compares per value, 64 values per lane:
    // a tenth of cnv4's 18-chunk tiles.
    if (DAVO_F32_FAST_EPILOGUE && (mtile + 1) * BM <= p.M && (ntile + 1) * BN <= p.Cout) {
        float* __restrict__ tbase = yg + (long)mtile * BM * p.y_ld + ntile * BN;
        const unsigned ld = (unsigned)p.y_ld;
        const bool relu = p.relu != 0;
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned roff = (unsigned)(wm * T::TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ld + (unsigned)(wn * T::TN * 32 + li);
#pragma unroll
                for (int j = 0; j < T::TN; ++j) {
                    float v = acc[i][j][r];
                    if (relu) v = fmaxf(v, 0.f);
                    tbase[roff + j * 32] = v;
                }
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int ncol = wn * T::TN * 32 + j * 32 + li;          // column inside the N tile
        const int n = ntile * BN + ncol;
        const bool n_ok = n < p.Cout;
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * T::TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = mtile * BM + row;
                float v = acc[i][j][r];
                if (p.relu) v = fmaxf(v, 0.f);
                if (n_ok && m < p.M) yg[(long)m * p.y_ld + n] = v;
            }
    }
}

template <int KS, int STRIDE, int BN, int LAYER>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32(ConvParams p) {
    conv_igemm_f32_body<KS, STRIDE, BN, LAYER>(p, blockIdx.x, gridDim.x, blockIdx.y);
}

// 128 x 256 tile, eight waves (64 x 64 each), one workgroup per CU (round 5 experiment, "f32_n256"): a pixel tile is staged once for
// all 256 output channels of cnv5 / cnv6 instead of once per 128-column tile
template <int KS, int STRIDE, int LAYER>
__global__ __launch_bounds__(512, 1) void conv_igemm_f32_n256(ConvParams p) {
    conv_igemm_f32_body<KS, STRIDE, 256, LAYER>(p, blockIdx.x, gridDim.x, blockIdx.y);
}

// A layer's main launch (whole rounds of 128-column tiles) and its remainder launch (narrower tiles, forward.hip / plan.hip) as ONE
// grid: the first n_main workgroups take the main tiles, the others the remainder's.  Workgroups are handed out strictly in id
// order, so the remainder's tiles start on whichever CUs finish their last main tile first: the main launch's ragged tail (tiles
// differ in length since the padding rows of the filter are skipped) and the remainder's start overlap instead of meeting at a
// launch boundary.  Same tiles, same kernels' arithmetic: bit-identical to the two launches.  n_main (and, for a grouped layer,
// n_rem) is a multiple of 8, so a workgroup's ordinal keeps its id % 8 = its XCD.
// A grouped layer (cnv7: two groups) keeps its groups in the x extent too - all main tiles of every group first, then all remainder
// tiles: with the groups in grid.y, group 0's remainder ran in the middle of the grid.
template <int KS, int STRIDE, int RBN, int LAYER>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32_mainrem(ConvParams pm, ConvParams pr, int n_main, int n_rem, int groups) {
    const int b = blockIdx.x, nm = groups * n_main;
    if (b < nm) conv_igemm_f32_body<KS, STRIDE, 128, LAYER>(pm, b % n_main, n_main, b / n_main);
    else conv_igemm_f32_body<KS, STRIDE, RBN, LAYER>(pr, (b - nm) % n_rem, n_rem, (b - nm) / n_rem);
}

#undef DAVO_ADDR_A
#undef DAVO_LOAD_CHUNK
#undef DAVO_LOAD_CHUNK_R
#undef DAVO_STORE_CHUNK_R
#undef DAVO_STORE_CHUNK
#undef DAVO_COMPUTE
#undef DAVO_COMPUTE16
#undef DAVO_FRAG0
#undef DAVO_COMPUTE_PF
#undef DAVO_COMPUTE_PF_G

}  // namespace davo
