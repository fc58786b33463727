// params.h — kernel parameter blocks, tile geometry and constants shared by the kernels
// (conv_igemm.h, conv_igemm_h3.h, conv_patch_h3.h, prologue.h) and the host code that plans and
// launches them.  No kernels here (two small inline helpers of the range guard apart), so host-only translation units include it cheaply.
#pragma once
#include <stdint.h>

// Matrix phases raise the wave's issue priority (s_setprio): where a SIMD holds waves of several workgroups, the one that
// has MFMAs to issue then wins the arbitration against the others' address arithmetic, LDS traffic and stores, and the matrix
// pipe starves less (conv_igemm_f32 cnv6 main launch 1.706 -> 1.672 ms; levels 1, 2, 3 measured alike; the patch kernels of both
// modes measured level with it and do without).  0 = off.
#ifndef DAVO_MMPRIO
#define DAVO_MMPRIO 1
#endif
#ifndef DAVO_MMPRIO_H3
#define DAVO_MMPRIO_H3 0          /* f16x3 kernels: see the measurement in profiles/r04n_setprio.md */
#endif
#define DAVO_PRIO_UP(level_) do { if ((level_) > 0) __builtin_amdgcn_s_setprio(level_); } while (0)
#define DAVO_PRIO_DOWN(level_) do { if ((level_) > 0) __builtin_amdgcn_s_setprio(0); } while (0)

namespace davo {

constexpr int NCLS = 19;            // Cityscapes train ids (utils/seg_utils/labels.py:64-101)
constexpr int SQ_CHUNKS = 32;       // SE squeeze: partial sums per (triplet, source) plane
constexpr int PH_SPLIT = 8;         // pose head: partial sums per (image, head)

struct Variant {
    int cin_per_frame, cnv6_out, se_act, norm_flow, abs_mode, att_source, mask_rgb, mask_info;
};

// ---- filter rows that only see padding ------------------------------------------------------
// 3x3 layers: the filter rows ky that land inside the image for at least one output pixel of the flattened pixel range
// [m0, m1] are [ky0, ky0 + nky); a range that crosses an image boundary keeps all three.  A tile whose pixels all sit in
// the top `rate` rows of the map never sees ky = 0 (TF pads with zeros above the image: nets/posenn.py:213-215, dilation 2,
// 4, 8 on 32-row maps), one in the bottom rows never ky = 2: the convolution kernels skip those rows' chunks per tile.
struct FilterRows { int ky0, nky; };
__host__ __device__ inline FilterRows valid_filter_rows(int m0, int m1, int Hout, int Wout, int Hin, int stride, int pad_t, int rate) {
    FilterRows fr = {0, 3};
    const int hw = Hout * Wout;
    const int n0 = m0 / hw;
    if (m1 < m0 || m1 / hw != n0) return fr;
    const int ymin = (m0 - n0 * hw) / Wout, ymax = (m1 - n0 * hw) / Wout;
    int lo = 0, hi = 2;
    while (lo < hi && ymax * stride - pad_t + lo * rate < 0) ++lo;
    while (hi > lo && ymin * stride - pad_t + hi * rate > Hin - 1) --hi;
    fr.ky0 = lo;
    fr.nky = hi - lo + 1;
    return fr;
}
// f16x3 kernels walk chunks channel block by channel block, nine taps each: the v-th chunk of a tile that keeps filter rows
// [ky0, ky0 + nky) is chunk (block * 3 + ky) * 3 + kx of the layer's weight rows
__host__ __device__ inline int h3_real_chunk(int v, int ky0, int nky) {
    const int sc = v / 3, kx = v - 3 * sc;
    const int blk = nky == 3 ? sc / 3 : (nky == 2 ? sc >> 1 : sc);
    return (blk * 3 + ky0 + (sc - blk * nky)) * 3 + kx;
}

// ---- conv_igemm.h (FP32 MFMA) ------------------------------------------------------------
struct ConvParams {
    const float* x;       // input activation, pixel-major NHWC
    const float* w;       // packed weights [Npad][Kpad]
    const float* bias;    // [Npad]
    float* y;             // output activation
    const float* zeros;   // >= 16 bytes of zeros: what a padded (out-of-image) tap reads
    int Hin, Win, Hout, Wout;
    int cin_log2;         // Cin = 1 << cin_log2 (channels per tap in the packed k order)
    int x_ld, x_coff;     // floats per input pixel, first channel used
    int y_ld, y_coff;     // floats per output pixel, first channel written
    int Cout;             // valid output channels (per group)
    int pad_t, pad_l, rate;
    const int* tile_order;   // or null: tile_order[i] = the tile the i-th workgroup (after xcd_remap) takes (long tiles first, forward.hip)
    int M;                // images * Hout * Wout
    int nchunks, Kpad, ntaps;
    int ntiles_n;
    int mtile0;           // first 128-row M tile of this launch (a layer may be split in two launches)
    int relu;
    // grouped launch (blockIdx.y = group): per-group strides
    int g_x_coff, g_y_coff;
    long g_w, g_bias;
    // cnv7 with the pose head in the epilogue (round 4; pred 1x1 + spatial mean are linear, nets/posenn.py:240-241): nothing is
    // stored but the tile's sum_pixels sum_channels relu(x) * Wpred[c][k], split by image, per 32-column unit:
    // pose_partial[group][mtile][8 units][2 image slots][3]; a tile wider than 32 columns writes its sum into its first unit and
    // zeros into the others, so the main (128-column) and the remainder (32-column) launch share one layout for pose_from_tiles
    const float* pose_w;    // [groups][256][3] pred kernels; null = store the activation as before
    float* pose_partial;
    int pose_P, pose_mt;    // output pixels per image (>= 128), M tiles of the whole layer
};

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDK = 36;          // padded LDS row (floats)

// BN = 16 (round 3): cnv1 has 16 output channels; on the 32-column tile half of every matrix instruction was padding.  Four waves of
// 32 rows x 16 columns on v_mfma_f32_16x16x4_f32 (conv_igemm.h).
template <int BN> struct Tile;
template <> struct Tile<16> {
    static constexpr int WN = 1, WM = 4, TM = 1, TN = 1, NB_LOADS = 1, WAVES = 4, RPP = 32, A_LOADS = 4;
    static constexpr int LDS_BYTES = 2 * (128 + 16) * 36 * 4;
};
template <int BN> struct Tile {
    static constexpr int WAVES = BN == 256 ? 8 : 4;    // 256 columns (round 5 experiment): eight waves, one workgroup per CU
    static constexpr int WN = BN == 256 ? 4 : (BN >= 64 ? 2 : 1);    // waves along N
    static constexpr int WM = WAVES / WN;          // waves along M
    static constexpr int TM = BM / WM / 32;        // 32x32 MFMA tiles per wave along M
    static constexpr int TN = BN / WN / 32;
    static constexpr int RPP = WAVES * 8;          // tile rows one staging pass of the workgroup covers (a thread: 4 floats of one row)
    static constexpr int A_LOADS = BM / RPP;       // float4 pixel loads per thread per chunk (4 | 2)
    static constexpr int NB_LOADS = BN / RPP;      // float4 weight loads per thread per chunk
    static constexpr int LDS_BYTES = 2 * (BM + BN) * LDK * 4;
};

// ---- conv_igemm_h3.h (f16x3) --------------------------------------------------------------
struct ConvParamsH {
    const uint8_t* x;       // split-fp16 blocked activation
    const uint8_t* w;       // packed weights: [Npad][nchunks][32 hi | 32 lo] halves (128 B per chunk)
    const float* bias;      // [Npad]
    uint8_t* y;             // output: float32 NHWC (y_mode 0) or split-fp16 blocked (y_mode 1)
    const uint8_t* zeros;   // >= 16 zero bytes: what a padded (out-of-image) tap reads
    int Hin, Win, Hout, Wout;
    long x_pix_bytes;       // bytes per input pixel (all channels of the tensor x 4)
    int x_pix_log2;         // log2 of it when it is a power of two (shared-tap staging needs one), else -1
    int x_boff;             // byte offset inside a pixel of the first channel block used
    int cb_log2;            // CB = channels per block = min(Cin, 32)
    int tpc_log2;           // taps per chunk = 32 / CB
    int cpb;                // chunks per channel block = ceil(ntaps / taps per chunk)
    int nchunks;
    long w_row_bytes;       // nchunks * 128
    int y_mode, y_ld, y_coff, Cout;
    int pad_t, pad_l, rate;
    int M, ntaps, ntiles_n, mtile0, relu;
    const int* tile_order;  // or null: as ConvParams::tile_order
    int Mtot;               // output rows of the whole layer (M is this launch's upper row bound): shared-tap staging reads the
                            // pixels RATE before and after a tile, which may belong to the layer's other launch
    int xs;                 // host: issue the shared-tap instantiation (conv_igemm_h3 RATE > 0) where the layer has one
    int deep;               // host: the launch has at most one workgroup per CU - issue the deep-ring instantiation of its tile
    // split-K with the fix-up folded into the launch (pose_tail.h, splitk_tail): the part that finishes last on a tile adds the tile's
    // sk_parts partial sums (y = float32 [M][sk_parts][Cout]) in fixed order and writes the stored form to sk_y; null = no fold
    unsigned* sk_counter;   // one ticket counter per tile of the launch (zero before and after)
    uint8_t* sk_y;          // the layer's split-fp16 blocked output
    unsigned* sk_range;     // range record word of the layer (or null)
    int sk_parts, sk_relu;
    int g_x_boff, g_y_coff;
    long g_w, g_bias;
    float out_scale;        // 2^(shift_out - shift_in) / (power of two the layer's weights were multiplied by)
    float bias_scale;       // accumulator init = bias * bias_scale (weight scale * 2^shift_in; a power of two)
    unsigned* range;        // y_mode 1: atomicMax of the stored (scaled) activations' bit patterns; may be null
    // y_mode 2 (cnv7 only): the pose head is fused into the epilogue, nothing is stored but this tile's
    // contribution to sum_pixels sum_channels relu(cnv7) * Wpred[c][k] for the (at most two) images it touches
    const float* pose_w;    // [groups][256][3] pred kernels
    float* pose_partial;    // [groups][pose_mt][ntiles_n][2 image slots][3]
    int pose_P, pose_mt;    // output pixels per image (>= tile height), M tiles in the launch
    // the tail of the pose head inside the same launch (pose_tail.h): the workgroup that finishes LAST adds the tiles'
    // partial sums in a fixed order and writes the poses; null pose_counter = a separate pose_from_tiles launch does it
    unsigned* pose_counter; // one counter per in-flight slot, zero between launches
    const float* pose_bias; // [groups][3] pred biases
    float* pose_out;        // [NB][6]
    int pose_NB, pose_bm, pose_total;   // pair images, tile height, workgroups of the launch
    int dbg;                // measurement only (DAVO_DBG; results are wrong with any bit set): 1 DMA reads the zero line,
                            // 2 no matrix phase, 4 no wave-half stagger (32x32x16 form), 32 no epilogue, 64 stores fold onto 256 tiles
};

constexpr int LDB = 144;        // LDS row: 128 data bytes + 16 pad (conflict-free b128 fragment reads)

template <int WM, int WN, int TM, int TN, int NSTG = 2> struct TileH {
    static constexpr int THREADS = WM * WN * 64;
    static constexpr int BMH = WM * TM * 32;
    static constexpr int BNH = WN * TN * 32;
    static constexpr int A_LOADS = BMH * 8 / THREADS;
    static constexpr int B_LOADS = BNH * 8 / THREADS;
    static constexpr int ROWS_PER_PASS = THREADS / 8;
    static constexpr int LDS_BYTES = 2 * (BMH + BNH) * LDB;          // register-staged: padded rows
#ifndef DAVO_H3_STAGES
#define DAVO_H3_STAGES 2
#endif
    // LDS ring slots: 2; the 128x128 tile of a remainder launch (one workgroup per CU, nothing else to hide the DMA
    // latency behind) takes 3
    static constexpr int DMA_STAGES = DAVO_H3_STAGES == 2 ? NSTG : DAVO_H3_STAGES;
    static constexpr int LDS_BYTES_DMA = DMA_STAGES * (BMH + BNH) * 128;   // LDS-DMA ring: linear rows, XOR-swizzled units
    static_assert(A_LOADS >= 1 && A_LOADS <= 4 && B_LOADS >= 1 && B_LOADS <= 4, "staging shape");
};

// conv_igemm_h3 with RATE > 0: the three kx taps of a filter row read one staged patch of BMH + 2*RATE pixels
template <int WM, int WN, int TM, int TN, int NSTG, int RATE> struct TileX {
    using T = TileH<WM, WN, TM, TN, NSTG>;
    static constexpr int PR = (T::BMH + 2 * RATE + 7) / 8 * 8;           // patch rows (a DMA wave-instruction = 8 rows)
    static constexpr int NWV = WM * WN;
    static constexpr int A_SLOTS = (PR / 8 + NWV - 1) / NWV;             // patch DMA instructions per thread per super-chunk
    static constexpr int APC = (A_SLOTS + 2) / 3;                        // ... issued per chunk
    static constexpr int LDS_BYTES = 2 * PR * 128 + T::DMA_STAGES * T::BNH * 128 + NWV * 1024 + 128;   // + parking + zero row
};

// ---- conv_igemm_h3s.h (f16x3, 208-pixel x 256-channel tile) ------------------------------
// W waves of 32 output channels each: 8 = the 208x256 tile (cnv5, cnv6, cnv7), 4 = the 208x128 tile (cnv4, 128 output channels:
// four waves, one per SIMD, one workgroup per CU)
template <int W> struct TileSW {
    static constexpr int BM = 208, BN = 32 * W, THREADS = 64 * W, WAVES = W;
    static constexpr int NP = BM / 16;                 // 13 pixel groups
    static constexpr int NC = 2;                       // 16-channel groups per wave
    static constexpr int RPP = THREADS / 8;            // LDS rows one DMA pass of the workgroup covers (8 per wave)
    static constexpr int NAJ = (BM + RPP - 1) / RPP;   // pixel-row loads per thread and chunk (4 | 7): the last pass holds rows 192..207, waves 0 and 1 only
    static constexpr int NBJ = BN / RPP;               // weight-row loads per thread and chunk (4)
    static constexpr int A_SLOT = BM * 128, B_SLOT = BN * 128;
    static constexpr int DUMMY = WAVES * 1024;         // where the waves without a last pixel-row load park theirs
    static constexpr int LDS_BYTES = 2 * (A_SLOT + B_SLOT) + DUMMY;
    static constexpr int lds_bytes(int nsa) { return nsa * A_SLOT + 2 * B_SLOT + DUMMY; }   // nsa pixel ring slots (conv_igemm_h3s NSA)
    static_assert(NBJ == 4 && 192 + 8 * 2 == BM, "staging shape");
};
typedef TileSW<8> TileS;

// ---- conv_patch_h3.h (cnv1 from an LDS patch) ---------------------------------------------
namespace cp1 {
constexpr int KS = 7, TH = 8, TW = 16;                 // filter, output tile
constexpr int PH = 2 * TH + KS - 2, PW = 2 * TW + KS - 2;      // 21 x 37 input pixels
constexpr int UNITS = 32;                              // 16-byte units per (py, parity) row (19 used)
constexpr int ROWB = UNITS * 16;                       // 512 B
constexpr int PLANE = PH * 2 * ROWB;                   // 21,504 B per plane (hi / lo)
constexpr int STEPS = 2 * KS;                          // 14 MFMA steps (8 taps per filter row)
constexpr int WBYTES = STEPS * 2 * 64 * 16;            // 28,672 B: [step][plane][lane] x 16 B
constexpr int LDS_BYTES = 2 * PLANE;                   // 43,008 B: the patch (the weight fragments live in registers)
constexpr int THREADS = 256;
}  // namespace cp1

// ---- conv_patch_h3.h (cnv2 from an LDS patch) ---------------------------------------------
namespace cp2 {
constexpr int KS = 5, TH = 8, TW = 8;                  // filter, output tile: 4 pixel groups of 2 rows x 8 columns
constexpr int PH = 2 * TH + KS - 2;                    // 19 input rows
constexpr int PWU = TW + 2;                            // 16-byte units per column parity of a patch row (columns 0..19)
constexpr int ROWB = 2 * PWU * 16;                     // 320 B per patch row: [even columns | odd columns], = 64 mod 128
constexpr int ROW_UNITS = 2 * PWU;
constexpr int REGION = (PH * ROWB + 255) / 256 * 256;  // 6,144 B per (plane, channel half) region, a multiple of 256
constexpr int NDMA = (PH * ROW_UNITS + 63) / 64;       // 6 LDS-DMA wave-instructions per region
constexpr int PATCH = 4 * REGION;                      // 24,576 B
constexpr int LDS_BYTES = 2 * PATCH;                   // double-buffered: 49,152 B, three workgroups per CU
constexpr int STEPS = KS * 3;                          // 15 MFMA steps: filter row x tap pair (kx = 5 is a zero-weight dummy)
constexpr int WBYTES = STEPS * 2 * 2 * 64 * 16;        // [step][N group][plane][lane] x 16 B
constexpr int THREADS = 256;
static_assert(NDMA * 1024 <= REGION, "the last DMA piece stays inside its region");
}  // namespace cp2

// ---- conv_patch_h3.h (cnv3 from an LDS patch) ---------------------------------------------
namespace cp3 {
constexpr int RATE = 2, TH = 8, TW = 8;                // 3x3, dilation 2, stride 1; output tile: 4 pixel groups of 2 rows x 8 columns
constexpr int PH = TH + 2 * RATE, PW = TW + 2 * RATE;  // 12 x 12 input pixels
constexpr int ROWB = PW * 16;                          // 192 B per patch row of a region; two rows = 384 B = 128 mod 256
constexpr int REGION = PH * ROWB;                      // 2,304 B per (plane, channel quarter) region, a multiple of 256
constexpr int PATCH = 8 * REGION;                      // 18,432 B = 18 LDS-DMA wave-instructions exactly
constexpr int NDMA = PATCH / 1024;
constexpr int LDS_BYTES = 2 * PATCH;                   // double-buffered: 36,864 B
constexpr int STEPS = 9;                               // one tap x 32 channels per MFMA step
constexpr int WBYTES = STEPS * 4 * 2 * 64 * 16;        // [step][N group][plane][lane] x 16 B
constexpr int THREADS = 256;
static_assert(REGION % 256 == 0 && PATCH % 1024 == 0 && (2 * ROWB) % 256 == 128, "bank-conflict-free fragment reads, whole DMA pieces");
}  // namespace cp3

struct ConvPatchParams {
    const uint8_t* x;       // packed split-fp16 input [NB][H][W][8 hi | 8 lo]
    const uint8_t* w;       // [14][2][64][8] halves: B fragments in lane order, pre-scaled
    const float* bias;      // [16]
    uint8_t* y;             // split-fp16 blocked output [NB][Ho][Wo][16 hi | 16 lo]
    const uint8_t* zeros;
    int H, W, Ho, Wo, pad_t, pad_l;
    int tiles_x, tiles_y, ntiles;
    float out_scale;
    float bias_scale;           // accumulator init = bias * bias_scale
    unsigned* range;            // atomicMax of the stored activations' bit patterns (range monitor); may be null
    // FUSED = true: the patch is built from the raw inputs (mask + pack fused in, the packed tensor
    // is never materialised): davo.py:1519-1522 (u8 -> f32), :1115,1178 (LUT attention), :1404-1442
    const uint8_t* img;     // u8 [B][H][3W][3]
    const float* flow;      // [B][4][H][W][2]
    const float* seg;       // [B][3][H][W][1]
    const float* tab;       // [B][3][19] attention tables (se_excite)
    Variant v;
    int dbg;                // measurement only (-DDAVO_TUNING, DAVO_PDBG): 1 = every patch load reads the zero line, 2 = no stores
};

// ---- f16x3 range guard ---------------------------------------------------------------------------------------------------
// A range record is RANGE_WORDS unsigned words: [layer] = bit pattern of the largest magnitude a storing epilogue of cnv1..cnv6 has
// written (atomicMax; non-negative floats order like their bit patterns), [RANGE_SNAP] = 1 once a batch's inputs were copied for
// a re-issue, [RANGE_SEQ] = sequence number of the last batch whose final kernel has seen the record (host mirror only).
// The maxima are RUNNING maxima over the batches that share a record: zeroed when a verdict failed or the scales changed, not
// per batch.  Why: a wave only pays for the atomic if it would raise the record, and a record that starts every batch at zero is
// raised by every wave of the first round - measured at batch 1 with per-batch zeroing: cnv1 8 -> 29 us, cnv5 28 -> 133 us,
// cnv6 37 -> 141 us (13,000 serialised atomics from the split-K fix-up alone); per-batch flags beside a running maximum cost the
// same while every workgroup kept them, and still 1 us per storing kernel when one workgroup in 64 did (six memory-side round
// trips at the end of a 10 us kernel).  So "clamped" is judged per batch exactly (the batch that first pushes the maximum past
// 65504 fails its verdict and the record is reset), and "too small" on everything stored since the last reset: a checkpoint
// whose scale does not fit shows in the first batch; activations that collapse by 2^-20 between two batches of one network on
// bounded inputs do not occur (ReLU layers are homogeneous in the input, the strip is u8), and the host path - one record per
// call - still sees them.  This is round 3's guard between two davo_synchronize calls, without its reset at each of them.
constexpr int RANGE_WORDS = 8, RANGE_SNAP = 6, RANGE_SEQ = 7;
// the verdict on one layer's maximum, shared by the host (forward.hip: check_range) and the kernel that keeps the inputs when it will fail
__host__ __device__ inline bool range_value_fails(float vmax) {
    return !(vmax < 65504.f) || (vmax > 0.f && vmax < 0x1p-11f);      // clamped / inf, or too small for the fp16 pairs' low halves
}
// a storing epilogue's note in the record: vmax = the wave's largest stored magnitude, leader = one lane of the wave
// (compiler builtins only, so that this header still needs no HIP header)
__device__ inline void range_note(unsigned* __restrict__ rec_layer, float vmax, bool leader) {
    if (!leader) return;
    const unsigned u = __builtin_bit_cast(unsigned, vmax);
    if (u > __atomic_load_n(rec_layer, __ATOMIC_RELAXED)) (void)__hip_atomic_fetch_max(rec_layer, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// What the last kernel of a ticketed device-path batch does for the range guard (api.hip, prologue.h): mirror the batch's finished
// record into page-locked host memory (the host's verdict then costs no copy), and keep the batch's inputs in the context's
// ring slot if the record fails.  record null = nothing to do; s_img null = no copy wanted ("stable_inputs", "auto_range" 0).
struct SnapArgs {
    unsigned* record;                       // the batch's range record (RANGE_WORDS words, layout above)
    unsigned* host_mirror;                  // device-visible address of its page-locked host copy
    unsigned seq;                           // this batch's sequence number (host_mirror[RANGE_SEQ] = seq tells the host the record is final)
    const uint8_t *img, *flow, *seg;        // the caller's buffers (16-byte aligned)
    uint8_t *s_img, *s_flow, *s_seg;        // the ring slot's
    unsigned img_vec, flow_vec_half, flow_vec, seg_vec;      // per window, in 16-byte units: strip, flow planes 0-1, whole flow block, seg
    int B;
};

}  // namespace davo
