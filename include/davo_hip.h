/* davo_hip.h — C ABI of libdavo_hip.so: DAVO frame-to-frame pose inference on MI355X (gfx950).
 *
 * The reference (BassyKuo/DAVO, TensorFlow 1.13) has no plugin/FFI layer: this path sits
 * behind the Python class DAVO (reference davo.py:30).  Each entry point below names the
 * reference interface it stands in for; davo_amd/davo.py is the ctypes host side that keeps
 * the reference's call surface on top of it.  Plain pointers and sizes only — no torch, no
 * TF types.  A context is NOT thread-safe: one context per GPU per host thread (the reference
 * drives sess.run from a single thread, test_kitti_pose.py:133-145).
 *
 * Every function returns 0 on success and a negative davo_status on failure; the message is
 * available from davo_last_error().  Tensor layouts are the reference's: NHWC, float32 unless
 * stated, HWIO convolution kernels, [in,out] dense kernels.
 */
#ifndef DAVO_HIP_H
#define DAVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct davo_ctx davo_ctx;

typedef enum {
    DAVO_OK = 0,
    DAVO_ERR_INVALID = -1,      /* bad argument / shape / unsupported variant */
    DAVO_ERR_HIP = -2,          /* a HIP runtime call failed                   */
    DAVO_ERR_NOT_READY = -3,    /* forward before every weight was loaded      */
    DAVO_ERR_NOMEM = -4,
    DAVO_ERR_RANGE = -5,        /* f16x3: a layer's activations left the fp16-pair storage range */
    DAVO_ERR_COMM = -6          /* librccl could not be loaded, or an RCCL call failed           */
} davo_status;

/* What the reference derives from the --version string (davo.py:1010-1102,1117-1450);
 * davo_amd/version.py:parse_version() fills it.  Field order is ABI. */
typedef struct {
    int32_t cin_per_frame;  /* 5: rgb+flow ("v1", davo.py:1062-1065) | 3: rgb only ("v0")        */
    int32_t cnv6_out;       /* -cnv6_(\d+), default 128 (davo.py:1052-1053); multiple of 32       */
    int32_t se_act;         /* 0 relu | 1 tanh | 2 leaky_relu(0.2)   (davo.py:1077-1085)          */
    int32_t norm_flow;      /* (f-0.32140523)/15.384229 on the SE input (davo.py:1089-1091)       */
    int32_t abs_mode;       /* 0 none | 1 |f_x| | 2 |f_y| | 3 |f|      (davo.py:1094-1102)        */
    int32_t att_source;     /* 0 ones (-no_segmask) | 1 se_flow | 2 static, tgt=1 | 3 static, all */
    int32_t mask_rgb;       /* rgb_k *= att_k                          (davo.py:1419-1423)        */
    int32_t mask_info;      /* flow_k *= att_k  (-segmask_all)         (davo.py:1430-1434)        */
} davo_variant;

/* Replaces DAVO(version).setup_inference(img_height, img_width, 'davo', 3, batch_size, ...)
 * (davo.py:1533-1551 -> build_pose_test_graph_davo, davo.py:955-1494): creates the device
 * context (workspace for up to max_batch triplets of H x W frames) on HIP device `device`.
 * H and W must be multiples of 4. */
int davo_create(davo_ctx** out, int device, int H, int W, int max_batch, const davo_variant* v);

/* Replaces tf.train.Saver(tf.trainable_variables()).restore(sess, ckpt)
 * (test_kitti_pose.py:129-131), one tensor at a time, keyed by the TF variable name
 * (e.g. "pose_exp_net/cnv1/weights", "pose_exp_net/pose/rotation/cnv6/biases",
 * "pose_exp_net/se_flow/bottleneck_fc/kernel").  `data` is a host pointer; the context keeps
 * its own (re-laid-out) device copy. */
int davo_load_weight(davo_ctx* ctx, const char* tf_name, const float* data,
                     const int64_t* shape, int ndim);

/* Number of weight tensors the variant still needs (0 = ready); names of the missing ones are
 * left in davo_last_error(). */
int davo_weights_missing(davo_ctx* ctx);

/* Replaces DAVO.inference(sess, mode='pose') (davo.py:1553-1569 -> sess.run({'pose': pred_poses})).
 * Host pointers, shapes as the reference's test driver sets them (test_kitti_pose.py:110-114):
 *   img  uint8 [B,H,3W,3]  (strip src0|tgt|src1, data_loader.py:537-557)
 *   flow f32   [B,4,H,W,2] (planes 0,1 used, davo.py:978-982)
 *   seg  f32   [B,3,H,W,1] (file order src0,tgt,src1, davo.py:998-1004)
 *   pose_out f32 [B,2,6]   rows = (tgt->src0, tgt->src1), each [rz,ry,rx,tx,ty,tz]
 * Synchronous: returns after pose_out is written.  1 <= B <= max_batch. */
int davo_forward(davo_ctx* ctx, int B, const uint8_t* img, const float* flow, const float* seg,
                 float* pose_out);

/* Same computation on device-resident buffers (the path bench.py times; multi-GPU shards keep
 * their windows in HBM).  Asynchronous on the context's stream unless elapsed_ms != NULL, in
 * which case it is bracketed by HIP events, synchronised, and the device time is returned.
 * With davo_set_inflight(ctx, n > 1) the timed form still waits for its own batch only and the
 * batches of all slots are judged together at davo_synchronize().
 * f16x3 range guard: the asynchronous form cannot know its own result, so every batch gets a slot of a ring of range
 * records (running maxima: "clamped" is exact per batch, "too small" covers what the slot has stored since its record was last
 * zeroed - at a failed verdict, a change of scales, and for every 256th batch) and is judged later: when its slot of a ring of 8 is needed again (eight batches on), at davo_synchronize(), or before
 * anything that changes the storage scales.  A failed verdict RE-ISSUES that batch (see "auto_range" below) and
 * rewrites its pose buffer, so:
 *   - the OUTPUT buffer of a batch must stay alive until davo_synchronize() has returned, and its poses are final only
 *     then (davo_range_stats / davo_range_report say whether anything was re-issued).  A pose buffer may be handed to a
 *     later batch before that (two alternating buffers; one buffer overwritten every step): the newest batch issued into a
 *     range of memory always wins - a re-issue writes into a buffer of the context first and is copied to the batch's own
 *     pose buffer only if no batch issued after it targets an overlapping range.  What such a caller cannot have is FINAL
 *     poses per batch before it synchronises (a D2H copy ordered on the stream behind batch n reads batch n's first issue):
 *     streaming callers use davo_submit / davo_wait below, where the library owns the pose buffers and delivers final poses;
 *   - the INPUT buffers (16-byte aligned) may be overwritten or recycled as soon as work ordered behind the call on
 *     the context's stream may run (e.g. the next H2D on that stream, or anything behind an event recorded there):
 *     the batch's last kernel sees the finished range record and, if the batch will have to be re-issued, copies
 *     the inputs it was issued on into buffers of the context (1.97 MB per 128x416 window, eight batches deep,
 *     allocated at the first such call); the re-issue reads that copy.  A batch in range - every batch of a
 *     well-ranged checkpoint - copies nothing.  A caller that keeps its inputs unchanged until davo_synchronize()
 *     anyway can save the buffers with davo_set_option(ctx, "stable_inputs", 1); re-issues then read the caller's.
 * The timed, synchronous form judges (and re-issues) its own batch; elapsed_ms is then the first issue's time. */
int davo_forward_device(davo_ctx* ctx, int B, const void* d_img, const void* d_flow,
                        const void* d_seg, void* d_pose, float* elapsed_ms);

/* The sequence loop as a stream.  The reference's driver pulls batches through tf.data's prefetch(8*B) while the session runs
 * (test_kitti_pose.py:133-145, data_loader.py:321-324), so input transfer, compute and result delivery of neighbouring batches
 * overlap; davo_forward returns poses and cannot.  davo_submit takes host pointers shaped as davo_forward's and rotates through the
 * in-flight slots (davo_set_inflight) like davo_forward_device: on the slot's own stream it queues the H2D copies of the planes the
 * path reads into the slot's staging set, the forward, and the D2H of the poses into a page-locked ring entry of the context.  With
 * two or more slots the copies of batch n+1 run under the kernels of batch n (one slot: copy and kernels alternate, still without
 * the host waiting for either).  It returns once the input buffers of the batch submitted `hold` calls earlier have been copied:
 * hold = 0 - on return THIS batch's img / flow / seg may be overwritten (a loader that recycles a batch when the next one is
 * asked for); hold = k - the caller keeps a batch's inputs unchanged for k more submits and the call waits for a copy that is
 * probably done already (k >= 8 never waits for a copy and records no event).  Page-locked inputs (davo_host_alloc /
 * davo_host_register) copy as DMA; pageable ones work, synchronously.
 * pose_out [B,2,6] is written BY THE HOST when the batch is delivered: after its f16x3 range ticket has been judged and, if it
 * failed, the batch re-issued (re-calibrated / float32 kernels, from the context's own copy of its inputs into the context's own
 * pose buffer).  Delivery happens in submission order inside later davo_submit calls (whatever has finished by then; the oldest
 * is waited for once eight batches are undelivered), davo_wait and davo_synchronize.  pose_out must stay valid until then; it may
 * be pageable.  davo_wait(ctx, n) returns once at most n submitted batches are undelivered (n = 0: all delivered; the streams may
 * still be busy with range bookkeeping, davo_synchronize also idles them).  davo_pending = undelivered batches (>= 0).
 * davo_forward, davo_set_stream and davo_set_inflight deliver everything first. */
int davo_submit(davo_ctx* ctx, int B, const uint8_t* img, const float* flow, const float* seg, float* pose_out, int hold);
int davo_wait(davo_ctx* ctx, int leave_pending);
int davo_pending(davo_ctx* ctx);

const char* davo_last_error(const davo_ctx* ctx);
void davo_destroy(davo_ctx* ctx);

/* ---- device memory / stream plumbing for hosts without a HIP binding (ctypes) ---------- */
int davo_device_malloc(davo_ctx* ctx, size_t bytes, void** out);
int davo_device_free(davo_ctx* ctx, void* p);
/* Page-locked host memory for the buffers handed to davo_forward (what tf.data's prefetch-to-pinned
 * staging does under sess.run, data_loader.py:319-325): H2D then runs as DMA at the PCIe rate and
 * overlaps the kernels of the previous sub-batch.  Pageable buffers work too, just slower. */
int davo_host_alloc(int device, size_t bytes, void** out);
int davo_host_free(void* p);
/* Page-lock memory the caller already owns (hipHostRegister), e.g. the shared-memory batch buffers that the input
 * pipeline's worker processes decode into (davo_amd/loader.py: ProcessWindowLoader); undo with davo_host_unregister
 * before the memory is unmapped. */
int davo_host_register(int device, void* p, size_t bytes);
int davo_host_unregister(void* p);
int davo_memcpy_h2d(davo_ctx* ctx, void* dst, const void* src, size_t bytes);
int davo_memcpy_d2h(davo_ctx* ctx, void* dst, const void* src, size_t bytes);
/* Gives every f16x3 batch issued through davo_forward_device and not judged yet its range verdict (each on its own
 * record), then waits for every stream of the context.  A batch that left the fp16-pair storage range is re-issued
 * before the call returns (re-calibrated, or on the float32 kernels: "auto_range"), so on DAVO_OK every pose buffer
 * holds float32-grade results.  With "auto_range" 0 the first failed verdict is returned as DAVO_ERR_RANGE instead
 * (the poses of that batch are not float32-grade; all batches have been judged, nothing stays pending). */
int davo_synchronize(davo_ctx* ctx);
/* Run on a caller-owned hipStream_t (NULL restores the context's own stream).  Batches issued on the stream the context leaves
 * are judged (and, if need be, re-issued) inside this call, i.e. it waits for them: the old stream may be destroyed afterwards. */
int davo_set_stream(davo_ctx* ctx, void* hip_stream);
/* Number of batches davo_forward_device keeps in flight (1..4, default 1).  With n > 1 the context
 * owns n streams and n activation workspaces and successive calls rotate through them, so the small
 * kernels of one batch overlap the large convolutions of another (the counterpart of the
 * reference's tf.data prefetch, data_loader.py:321-324; +3...7 % throughput at n = 2, B = 32).  The
 * caller must give concurrent calls distinct input/output buffers and call davo_synchronize()
 * before reading results.  davo_forward (host buffers) stays synchronous. */
int davo_set_inflight(davo_ctx* ctx, int n);

/* ---- measurement ---------------------------------------------------------
 * With profiling on, every kernel launch of davo_forward[_device] is bracketed by HIP events
 * on the launch stream.  davo_profile_entry(i) returns the kernel's label, number of timed
 * launches and total device milliseconds since the last davo_profile_reset(); it returns
 * DAVO_ERR_INVALID once i is past the last entry.  Reading synchronises the stream. */
int davo_profile_enable(davo_ctx* ctx, int on);   /* 0 off | 1 every kernel | 2 only the dominant kernel (main cnv6 launch) */
int davo_profile_reset(davo_ctx* ctx);
int davo_profile_entry(davo_ctx* ctx, int i, char* name, int name_len, int* launches,
                       double* total_ms);
/* Per-launch record of one label since the last reset, in issue order: which = 0 the launches' own durations (ms),
 * which = 1 the time from the previous bracketed launch's START to this one's (with every batch bracketed, one step's
 * period on the stream; -1 for the first sample).  Copies up to cap floats, returns the number recorded (>= 0) or a
 * negative davo_status.  bench.py reports the clock ramp of its timed region from it. */
int davo_profile_samples(davo_ctx* ctx, const char* name, int which, float* out, int cap);

/* How the LAST forward issued conv layer `layer` (0..6 = cnv1..cnv7): launch 0 is the main launch,
 * launch 1 the remainder launch with a narrower N tile (mtiles = 0 if there was none).  A launch
 * covers `mtiles` 128-row M tiles x all output channels, in N tiles of `bn`.  bench.py uses it to
 * price the FLOPs of the launch it puts on the roofline. */
int davo_last_plan(davo_ctx* ctx, int layer, int launch, int* mtiles, int* bn);

/* Arithmetic of the convolution stack:
 *   1 (default) "f16x3": every float32 operand is split into two fp16 halves (22 significant
 *      bits) and each product is three fp16 MFMA products accumulated in float32 — float32-grade
 *      results (measured ~1e-7 relative on the 6-DoF outputs) at 5.3x less matrix-pipe time;
 *      a layer's activations must stay within the fp16-pair range (see davo_calibrate below).
 *   0 "f32": v_mfma_f32_32x32x2_f32, bit-for-bit float32 fmaf chains, no range restriction. */
int davo_set_precision(davo_ctx* ctx, int precision);

/* f16x3 range management.  Activations between layers are stored as fp16 (hi, lo) pairs; a layer's values must
 * stay below 65504 (they are clamped there) and its largest value above ~2^-11 for the pairs to carry float32-grade
 * precision.  Each storing kernel records the largest value it wrote (running maxima: a clamped value is caught in the very call
 * that stores it, "too small" on what has been stored since the record was last zeroed - by a recovery, a change of scales, and for
 * every 256th call / batch); davo_forward judges the record at the end of its call.
 * The reference's float32 graph never fails on a finite network (davo.py:1553-1569), so by default neither does this
 * library: a batch that left the range is re-issued with the scales re-calibrated ON THAT BATCH, and if it still
 * leaves the range (no per-layer power of two covers it) on the float32 kernels — mode 0 below, for that batch only;
 * later batches run f16x3 with the new scales.  davo_range_stats counts what happened.  davo_set_option(ctx,
 * "auto_range", 0) turns the recovery off: the entry points then return DAVO_ERR_RANGE (poses are still written, not
 * float32-grade) and the caller calibrates or changes mode itself.  davo_calibrate runs the path on
 * a sample batch (device buffers, as davo_forward_device) and gives every layer an exact power-of-two storage
 * scale that puts its largest value in [512, 1024); results inside the safe range do not depend on the scales
 * beyond rounding noise (~1e-8).  The reference's float32 graph needs none of this (TF conv2d, nets/posenn.py:205-215);
 * mode 0 (f32) ignores the scales.
 *   davo_activation_range: max_abs[6] = largest |activation| written by cnv1..cnv6 since the last reset (true
 *     magnitudes), shifts[6] = log2 of the storage scales; either may be NULL.  Synchronises.
 *   davo_set_activation_shifts: install scales from an earlier calibration (NULL = all zero). */
int davo_calibrate(davo_ctx* ctx, int batch, const void* d_img, const void* d_flow, const void* d_seg, int* shifts_out);
int davo_activation_range(davo_ctx* ctx, float* max_abs, int* shifts, int reset);
int davo_set_activation_shifts(davo_ctx* ctx, const int* shifts);
/* Range recoveries since davo_create: re-calibrations triggered by a failed verdict, batches that ran on the float32
 * kernels because no scale covered them, batches re-issued in total.  Any pointer may be NULL. */
int davo_range_stats(davo_ctx* ctx, long long* recalibrations, long long* f32_batches, long long* reissued);
/* What the range management last did, in words ("" if nothing yet): which layer's verdict triggered a re-calibration or
 * a float32 batch, or which weight tensor's per-input-channel spread keeps the network on the float32 kernels.  The
 * string lives until the next call on the context. */
const char* davo_range_report(const davo_ctx* ctx);

/* Options.
 *   "auto_range" (default 1): f16x3 batches that leave the fp16-pair storage range are re-issued (see above); 0 = the
 *       failed verdict is returned as DAVO_ERR_RANGE.
 *   "stable_inputs" (default 0): 1 = the caller keeps the input buffers of every davo_forward_device batch unchanged
 *       until davo_synchronize(); the library then takes no copies of them (see davo_forward_device).
 * Kernel-fusion switches of the f16x3 path (both modes give the same poses to ~1e-7):
 *   "fuse_pose" (default 1): pred 1x1 + spatial mean + 0.01 (nets/posenn.py:240-250) run in cnv7's
 *       epilogue; the cnv7 activation is never written to HBM.  0 = cnv7 stored, separate pose-head kernels.
 *   "fold_tails" (default -1 = auto): 1 = the 2->8->19 excitation MLP runs in the workgroup of the squeeze launch that
 *       delivers a triplet's last partial sum, and the pose head's sum over cnv7's tiles in the workgroup of the cnv7
 *       launch that finishes last (csrc/pose_tail.h: ticket counters, agent-scope loads and stores, fixed summation order):
 *       two launches fewer per batch; 2 = the excitation only; 0 = se_excite / pose_from_tiles as launches of their own.
 *       The memory-side round trips cost more than the launches they save except at the smallest batches (B = 32: squeeze
 *       +18 us, cnv7 +3 us against 13 us of launches), so auto folds the excitation at batch <= 2 only.  Bit-identical.
 *   "deep_ring" (default 1): launches of at most one workgroup per CU (batch 1..4) run the same tiles on LDS rings of 3..6
 *       slots instead of 2 (more chunks of LDS-DMA in flight behind a counted wait).  Bit-identical results.
 *   "split_k" (default 1): cnv5 / cnv6 launches of at most half a workgroup per CU (batch 1 at 128x416) run the two halves
 *       of their input channels as two groups of the same kernel into float32 partial sums; a fix-up kernel adds them,
 *       applies ReLU and writes the stored form (cnv6 at batch 1: 52 -> 39 us).  Two partial sums instead of one chain: the
 *       layer's outputs - and so the poses - agree with the single chain to float32 rounding (~1e-7 relative), not to the
 *       bit; 0 = one K chain at every batch size (poses then do not depend on how windows are batched, with "fuse_pose" 0
 *       to the bit).
 *   "fold_fixup" (default 0): with "split_k", the part of a tile that finishes last adds the tile's partial sums and writes the stored
 *       form itself, so the two fix-up launches of a batch-1 forward (7 us each) go: the parts of a tile are the workgroups (x, 0..S-1)
 *       of a grid whose x extent is a multiple of 8, workgroup ids go round-robin over the XCDs (checked on the device once per
 *       context), so the partial sums meet in one XCD's L2.  Same additions in the same order: bit-identical to 0.  Measured SLOWER
 *       (batch 1: 0.141 against 0.132 ms per forward, profiles/r05k_fold_fixup_ab.md): one workgroup per tile adds what the fix-up
 *       launch spreads over the whole chip, behind a ticket's memory-side round trip; kept for experiments.
 *   "f32_n16" (default 1; float32 mode): cnv1 (16 output channels) on a 128x16 tile with v_mfma_f32_16x16x4_f32 instead of the
 *       128x32 tile whose matrix instructions were half padding.  Another order of the same float32 fma chain per output.
 *   "f32_n256" (default 0; float32 mode): cnv5 / cnv6 on a 128 x 256 tile of eight waves, one workgroup per CU (csrc/conv_igemm.h,
 *       conv_igemm_f32_n256): a pixel tile is staged once for all 256 output channels.  Bit-identical; measured 4 % slower per step
 *       than the 128-column tiles (eight waves behind one barrier stay in step: MEASURED_AND_REJECTED.md); kept for experiments.
 *   "merge_rem_f32" (default 1; float32 mode): where cnv4 / cnv5 / cnv6 are planned as a main launch of whole rounds of 128-column tiles
 *       plus a remainder launch of narrower ones, both run as ONE grid (csrc/conv_igemm.h, conv_igemm_f32_mainrem): workgroups are
 *       handed out in id order, so the remainder's tiles start on the CUs that finish their last main tile first (float32 step
 *       -1.8 % at B = 32; cnv7 measured slower merged and keeps two launches).  Bit-identical to 0.
 *   "patch_f32" (default 1; float32 mode): cnv1, cnv2 and cnv3 from an LDS-staged input patch on v_mfma_f32_16x16x4_f32
 *       (csrc/conv_patch_f32.h: the f16x3 patch kernels' recipe in float32) instead of the implicit GEMM: 0.153 / 0.070 / 0.091
 *       -> 0.108 / 0.049 / 0.069 ms at B = 32.  Another fixed order of the same float32 fma chain per output.
 *   "tile_208x128" (default 0): 1 = cnv4 may run on a 208-pixel x 128-channel tile of four waves (csrc/conv_igemm_h3s.h): whole
 *       rounds of the 256 CUs at every batch that is a multiple of 8, bit-identical results, measured 8 % slower than the
 *       128x128 tile at B = 32 and level at B = 16; kept for experiments.
 *   "merge_order" (default -1 = 2 where the launch has a tile order, see "skip_order", else 0): how the merged grid interleaves
 *       its two tile shapes: 0 = offset inside every XCD (half of each XCD's CUs run their short tile first), 1 = per XCD (even
 *       XCDs first, odd XCDs last: an XCD's CUs stay in step, a fifth fewer L2 misses, 0.5 % slower), 2 = all main tiles, long
 *       ones first, then the remainder tiles.  Bit-identical results.
 *   "skip_order" (default 1): the 3x3 kernels do not walk the chunks of a filter row that is zero padding for every pixel of
 *       a tile (davo_tile_filter_rows), so the tiles of a launch differ in length; every XCD's run of tiles can be handed out
 *       long tiles first (a device table per launch shape): 0 = never, 1 = in the float32 launches (cnv5 850 -> 793 us),
 *       2 = in the f16x3 merged grids too (cnv5 259.9 -> 256.8 us, a third more HBM reads, same step time under the power cap).
 *       Bit-identical results.
 *   "merge_cnv4" (default 0): cnv4 as whole rounds of 256x128 tiles + 128x128 remainder tiles in one grid like cnv5 / cnv6
 *       ("merge_rem"); measured level with the single launch of 128x128 tiles.  Bit-identical results.
 *   "fuse_pack" (default -1 = auto, which is off: measured level at every batch): 1 = cnv1 builds its input patch from
 *       the raw inputs (mask + pack fused in).
 *   "patch_cnv2", "patch_cnv3" (default 1): cnv2 (5x5 stride 2) / cnv3 (3x3 dilation 2) read their taps from an LDS-staged
 *       input patch (csrc/conv_patch_h3.h); 0 = the implicit-GEMM kernel.  The two sum a pixel's taps in different orders:
 *       poses agree to float32 rounding.
 *   "merge_rem" (default 1): where cnv5 / cnv6 are planned as a main launch of 256x256 tiles plus a remainder launch of
 *       128x128 tiles (B = 32), both run as ONE grid in which half of the CUs take their remainder tile first and the other
 *       half last (csrc/conv_igemm_h3.h, conv_igemm_h3_mainrem): the halves' store bursts no longer coincide.  0 = two
 *       launches.  Same tiles, same kernels' arithmetic: bit-identical results.
 *   "share_taps" (default 1): cnv3..cnv6 on the 256-row tiles stage ONE pixel patch per filter row for its three kx taps
 *       (csrc/conv_igemm_h3.h, RATE > 0); 0 = a staged chunk per tap.  Bit-identical results.
 *   "wave128" (default 2, round 5): 1 = cnv5 / cnv6's launches of whole 256x256 tiles run on FOUR waves of 128x128 outputs each
 *       (csrc/conv_igemm_h3w.h: 0.167 LDS fragment reads per matrix instruction instead of 0.25, which is what the power-capped
 *       matrix pipe's rate depends on; one wave per SIMD, every other instruction of the loop in a slot behind one MFMA); the
 *       layer's remainder rows then run as a launch of their own ("merge_rem" does not apply).  Same products in the same order:
 *       bit-identical to 0 = conv_igemm_h3's eight waves of 64x128.  -4.6 % per step at B = 32 (profiles/r05bd_w128_ab.log).
 *       2 = the remainder rows too leave conv_igemm_h3's 128x128 tiles: 256x64 tiles on four waves of 64x64 outputs, the shared pixel
 *       patch and deep DMA rings (conv_igemm_h3w64: 19 KB staged per chunk instead of 32; cnv5 / cnv6 remainder 34 / 59 -> 31 / 50 us,
 *       profiles/r05bh_w64_ab.log).  Bit-identical as well.
 *       With 2, cnv4 too runs on four-wave tiles (conv_igemm_h3w128: 256x128, shared patch, five-slot weight ring) where its 256-row
 *       tiles fill whole rounds of the CUs to within 12 % (B = 128: 0.385 -> 0.318 ms; B = 32: level, so not taken); 3 forces that kernel
 *       wherever it fits (test hook).  Bit-identical.
 *   "cu_partition" (default 0; with davo_set_inflight(ctx, n > 1)): slot i's stream is CU-masked to its own 1/n of every
 *       XCD's compute units (hipExtStreamCreateWithCUMask) and its launches are planned for that many CUs.  Measured
 *       without gain (HISTORY.md round 2); kept for experiments.  Results do not change.
 *   "force_tile" (test hook, default -1 = the launch planner decides): tile id of csrc/plan.h (0 128x32, 1 256x64,
 *       2 256x128, 3 128x256, 4 128x128, 5 256x256, 6 208x256); every f16x3 layer the tile fits is issued as one
 *       launch of that shape.  Poses do not depend on it beyond float32 rounding of the fused pose head's sums.
 *   "profile_stride" (default 1): with davo_profile_enable(ctx, 2) (only the dominant kernel bracketed by HIP events) the
 *       events go around every n-th batch's launch only: a pair costs two ~6 us pipeline bubbles around the launch it times.
 * Host-buffer entry point (davo_forward):
 *   "host_chunk" (default 8): windows per sub-batch; the H2D copy of sub-batch i+1 overlaps the kernels of
 *       sub-batch i when B >= 2*host_chunk.  0 = copy the whole batch, then compute.  Results do not change. */
int davo_set_option(davo_ctx* ctx, const char* key, int value);

/* ---- multi-GPU: the pose gather of the window-sharded sequence driver, on RCCL ---------------
 * The reference runs one process on one GPU (test_kitti_pose.py:133-149: window loop, then the
 * sequential 4x4 chain).  Here one process per GPU takes a contiguous window range; the only
 * exchange is one all-gather of [n,2,6] float32 before the chain.  librccl is loaded on the first
 * call of this group.  Protocol: rank 0 calls davo_comm_unique_id and ships the DAVO_COMM_ID_BYTES
 * to the other ranks by any side channel (davo_amd/comm.py: a file), then every rank calls
 * davo_comm_init with its own context (one GPU per rank: RCCL refuses two ranks on one device).
 *   davo_allgather_poses: local [n_local,2,6] host floats (n_local <= n_per_rank; the rest of the
 *     rank's slot is zero-filled) -> all [nranks*n_per_rank,2,6] host floats on every rank, rank r's
 *     windows at [r*n_per_rank, ...).  elapsed_ms (may be NULL) = device time of the collective alone.
 *   davo_allgather_poses_device: the same on device buffers (n_per_rank windows in, nranks*n_per_rank
 *     out), after the context's own streams have drained.
 *   davo_comm_allreduce: one double, in place; op 0 sum | 1 max | 2 min (bench.py: max-over-ranks).
 *   davo_comm_barrier: every stream of this context idle, then a rendezvous of all ranks.
 *   davo_comm_preload: only loads librccl (573 MB to map, its code objects to register: most of a second); needs no context and
 *     may run on any thread, e.g. at process start behind the host's own imports.
 * Threads: davo_comm_preload, davo_comm_unique_id and davo_comm_init may run on a second host thread while the context's owner
 * thread issues forwards (a communicator takes seconds to build and is not needed before the gather); davo_comm_init publishes
 * the communicator into the context as its last step.  Every other call of this group belongs to the owner thread.
 * Errors: DAVO_ERR_COMM (message names the RCCL call); there is no fallback transport. */
#define DAVO_COMM_ID_BYTES 128
int davo_comm_preload(char* err, int err_len);
int davo_comm_unique_id(void* id_out, char* err, int err_len);
int davo_comm_init(davo_ctx* ctx, int nranks, int rank, const void* id);
int davo_comm_size(davo_ctx* ctx, int* nranks, int* rank);
int davo_allgather_poses(davo_ctx* ctx, const float* local, int n_local, int n_per_rank, float* all,
                         float* elapsed_ms);
int davo_allgather_poses_device(davo_ctx* ctx, const void* d_local, int n_per_rank, void* d_all,
                                float* elapsed_ms);
int davo_comm_allreduce(davo_ctx* ctx, double* value, int op);
int davo_comm_barrier(davo_ctx* ctx);
int davo_comm_destroy(davo_ctx* ctx);

/* ---- test hooks -------------------------------------------------------------------------
 * impl 0 = MFMA implicit-GEMM kernels (default, the product path);
 * impl 1 = one-thread-per-output direct convolution in HIP on the reference's own tensor
 *          layouts (10-channel input, HWIO weights) — an on-device cross-check, never timed. */
int davo_set_impl(davo_ctx* ctx, int impl);

/* Copy an intermediate of the LAST forward to host (float32, NHWC, pair-image major = 2B images):
 * "att_table" [B,3,19], "packed" [2B,H,W,8|10], "cnv1".."cnv5", "cnv6" [.., 2*cnv6_out]
 * (rotation | translation), "cnv7" [.., 512].  n_floats must equal the tensor size. */
int davo_debug_read(davo_ctx* ctx, const char* tensor, float* host_out, size_t n_floats);

/* Stand-alone slim.conv2d(padding='SAME') (nets/posenn.py:205-215) through the same MFMA
 * kernels, host pointers: x [N,H,W,Cin] float32 (Cin a power of two >= 4; >= 8 for f16x3),
 * w HWIO [k,k,Cin,Cout], y [N,ceil(H/stride),ceil(W/stride),Cout] float32.
 * k in {1,3,5,7}, stride in {1,2}; precision as davo_set_precision. */
int davo_conv2d_same(int device, const float* x, int N, int H, int W, int Cin,
                     const float* w, int k, int Cout, const float* bias,
                     int stride, int rate, int relu, int precision, float* y, char* err, int err_len);

/* The launch planner on its own (no GPU needed): how an f16x3 layer of M output rows x npad output
 * channels (x groups) is issued.  Up to two launches: rows[i] output rows in tiles of tile_bm[i] x
 * tile_bn[i]; returns the number of launches (1 or 2) or a negative davo_status. */
int davo_plan_layer(int M, int npad, int groups, int* rows, int* tile_bm, int* tile_bn);

/* The per-tile rule of the 3x3 kernels on its own (no GPU needed): a tile that covers the flattened output pixels [m0, m1]
 * of a [*, Hout, Wout] map walks only the filter rows [*ky0, *ky0 + *nky) that reach inside the Hin-row input for at
 * least one of its pixels (slim.conv2d pads with zeros: nets/posenn.py:205-215); the chunks of the other rows multiply
 * padding only and are skipped.  chunk_map[v], v < 3 * nky * nblocks (optional, may be NULL): index, in the layer's
 * weight rows, of the v-th chunk the f16x3 kernels walk for such a tile.  Returns DAVO_OK or a negative davo_status. */
int davo_tile_filter_rows(int m0, int m1, int Hout, int Wout, int Hin, int stride, int pad_t, int rate,
                          int* ky0, int* nky, int nblocks, int* chunk_map);

#ifdef __cplusplus
}
#endif
#endif /* DAVO_HIP_H */
