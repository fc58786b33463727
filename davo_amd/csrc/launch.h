// launch.h — the kernel launch functions, one translation unit per kernel family so the device
// code compiles in parallel.  Each launcher sets the dynamic-LDS attribute of its kernel once per
// (device, kernel) and returns the launch's error.
#pragma once
#include <hip/hip_runtime.h>

#include "params.h"
#include "plan.h"

namespace davo {

// hipFuncAttributeMaxDynamicSharedMemorySize for `kernel` on the CURRENT device, set once per (device, kernel):
// a function attribute belongs to the device's code object, so a second context on another GPU of the same
// process needs its own call (launch_misc.hip).
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);

// ---- launch_f32.hip: conv_igemm_f32 (FP32 MFMA, bit-exact fmaf chains) ---------------------------
// generic shapes (davo_conv2d_same and non-default cnv6 widths)
hipError_t launch_conv(int KS, int stride, int BN, const ConvParams& p, dim3 grid, hipStream_t s);
// the seven PoseNN layers (layer 0..6 = cnv1..cnv7), each under its own kernel name; BN is chosen per launch
hipError_t launch_layer(int layer, int BN, const ConvParams& p, dim3 grid, hipStream_t s);
hipError_t launch_layer_n256(int layer, const ConvParams& p, dim3 grid, hipStream_t s);
hipError_t launch_layer_mainrem(int layer, int rbn, const ConvParams& pm, const ConvParams& pr, int n_main, int n_rem, int groups, hipStream_t s);

// ---- launch_h3.hip: conv_igemm_h3 (f16x3) -----------------------------------------------------------
hipError_t launch_layer_h3(int layer, int tile, const ConvParamsH& p, dim3 grid, hipStream_t s);
// main launch (256x256 tiles, shared-tap staging) + remainder launch (128x128 tiles) of cnv5 / cnv6 (layer 4 / 5) as one grid
// (conv_igemm_h3_mainrem); hipErrorNotSupported when the two launches do not have that shape: the caller issues them separately
// whether launch_layer_h3_mainrem would issue this shape (asked BEFORE the profiling scope of the launch is opened)
bool layer_h3_mainrem_supported(int layer, const ConvParamsH& pm, int n_main, int n_rem);
// order: 0 = the short tiles' offset inside every XCD (round 2), 1 = per XCD (even XCDs short tiles first, odd XCDs last)
hipError_t launch_layer_h3_mainrem(int layer, const ConvParamsH& pm, int n_main, const ConvParamsH& pr, int n_rem, int order, hipStream_t s);
// launch_h3s.hip: conv_igemm_h3s (TILE_208x256) for layer 4..6 = cnv5, cnv6, cnv7
hipError_t launch_layer_h3s(int layer, const ConvParamsH& p, dim3 grid, hipStream_t s);
// conv_igemm_h3w.h: 256x256 tiles on four waves of 128x128 (cnv5, cnv6; option "wave128")
bool layer_h3w_supported(int layer, const ConvParamsH& p);
hipError_t launch_layer_h3w(int layer, const ConvParamsH& p, dim3 grid, hipStream_t s);
bool layer_h3w128_supported(const ConvParamsH& p);
hipError_t launch_layer_h3w128(const ConvParamsH& p, hipStream_t s);      // cnv4: 256x128 tiles, p.mtile0 in 256-row tiles
hipError_t launch_layer_h3w64(int layer, const ConvParamsH& p, hipStream_t s);      // remainder rows: 256x64 tiles, p.ntiles_n = 4, p.mtile0 in 256-row tiles
hipError_t launch_h3_generic(int KS, int stride, int tile, const ConvParamsH& p, dim3 grid, hipStream_t s);

// ---- launch_misc.hip: prologue, pose head, cnv1 patch kernel, direct convolution ---------------------
hipError_t launch_se_squeeze(const float* d_flow, int B, int HW, const Variant& v, float* d_partial, hipStream_t s);
hipError_t launch_se_excite(const float* d_partial, int B, int HW, const Variant& v, const float* w1, const float* b1,
                            const float* w2, const float* b2, const float* wstatic, float* d_tab, unsigned* d_range_reset, hipStream_t s);
// squeeze + excitation in one launch: the last workgroup of each triplet evaluates its tables (prologue.h, pose_tail.h);
// -> 1 if the parts (x, 0..3) of a 64 x 4 grid ran on one XCD each for every x (what the folded split-K fix-up relies on), 0 if not
hipError_t xcd_round_robin_probe(hipStream_t s, int* ok);
// d_counters: one zeroed unsigned per triplet, left at zero
hipError_t launch_se_squeeze_excite(const float* d_flow, int B, int HW, const Variant& v, float* d_partial, unsigned* d_counters,
                                    const float* w1, const float* b1, const float* w2, const float* b2, const float* wstatic,
                                    float* d_tab, unsigned* d_range_reset, hipStream_t s);
// ld: 16 = split-fp16 8-channel layout (f16x3), 8 = float32 8-channel, 10 = the reference's 10-channel layout
hipError_t launch_mask_pack(int ld, const uint8_t* d_img, const float* d_flow, const float* d_seg, const float* d_tab,
                            const Variant& v, int B, int H, int W, float* d_packed, hipStream_t s);
hipError_t launch_cnv1_patch(bool fused, const ConvPatchParams& p, int nblk, hipStream_t s);
hipError_t launch_cnv2_patch(const ConvPatchParams& p, int nblk, hipStream_t s);
hipError_t launch_cnv3_patch(const ConvPatchParams& p, int nblk, hipStream_t s);
// float32 mode (conv_patch_f32.h)
hipError_t launch_cnv1_patch_f32(const ConvPatchParams& p, int nblk, hipStream_t s);
hipError_t launch_cnv2_patch_f32(const ConvPatchParams& p, int nblk, hipStream_t s);
hipError_t launch_cnv3_patch_f32(const ConvPatchParams& p, int nblk, hipStream_t s);
// split-K fix-up: d_part [M][S][N] float32 partial sums -> the layer's stored activation (ReLU, fp16 hi/lo pairs, range monitor)
hipError_t launch_splitk_fixup(const float* d_part, long M, int N, int S, int relu, uint8_t* d_y, unsigned* d_range, hipStream_t s);
hipError_t launch_pose_from_tiles(const float* d_tiles, int NB, int P, int bm, int mtiles, int ntiles_n,
                                  const float* d_bpred, float* d_pose, const SnapArgs& snap, hipStream_t s);
// the range guard's conditional copy of the batch's inputs as a launch of its own (prologue.h)
hipError_t launch_range_guard_snapshot(const SnapArgs& snap, hipStream_t s);
hipError_t launch_pose_head(const float* d_c7, int NB, int P, const float* d_wpred, const float* d_bpred,
                            float* d_partial, float* d_pose, hipStream_t s);
hipError_t launch_conv_direct(const float* x, int N, int Hin, int Win, int cin, int x_ld, int x_coff, const float* w, int KS,
                              int cout, const float* bias, int stride, int rate, int pt, int pl, int Ho, int Wo, int relu,
                              float* y, int y_ld, int y_coff, hipStream_t s);

}  // namespace davo
