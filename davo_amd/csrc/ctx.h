// ctx.h — host-side state of one davo_ctx (include/davo_hip.h) and the small helpers every
// translation unit of libdavo_hip.so shares.  Host code only; the kernels live in conv_igemm.h,
// conv_igemm_h3.h, conv_igemm_h3s.h, conv_patch_h3.h and prologue.h and are launched through launch.h.
//
// Translation units (built in parallel by davo_amd/_lib.py, linked into one shared library):
//   api.hip         extern "C" entry points (context, weights, forward, calibration, test hooks)
//   forward.hip     the forward plan of the pose path (which kernel, which buffers, in what order)
//   plan.hip        launch planning: tile shapes and whole-round launch splits (pure host logic)
//   weights.hip     weight re-layout: HWIO float32 -> packed f32 / split-fp16 operands
//   launch_f32.hip  conv_igemm_f32 instantiations + dispatch
//   launch_h3.hip   conv_igemm_h3 instantiations + dispatch (the f16x3 path, the long compile)
//   launch_h3s.hip  conv_igemm_h3s instantiations (208x256 tile); launch_h3_generic.hip: davo_conv2d_same's shapes
//   launch_misc.hip prologue / pose head / cnv1..cnv3 patch / direct-convolution kernels + dispatch
//   comm.hip        RCCL communicator behind the C ABI (pose gather of the window-sharded driver)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <map>
#include <string>
#include <vector>

#include "../../include/davo_hip.h"
#include "params.h"

namespace davo {

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    float* dev = nullptr;          // raw copy in the reference layout (impl 1, pose_head, SE)
};

struct ConvLayer {
    const char* label;
    int KS, stride, rate;
    int cin, cin_log2, cout;       // packed input channels per tap (power of two), valid outputs
    int BN, npad, kpad, nchunks, groups;
    float* d_w = nullptr;          // [groups][npad][kpad]
    float* d_b = nullptr;          // [groups][npad]
    // f16x3 path (conv_igemm_h3.h): channel-blocked k order, split-fp16 packed weights
    int cb_log2 = 0, tpc_log2 = 0, cpb = 0, nchunks_h = 0, npad_h = 0, tile_h = 0;
    float wscale = 1.f;            // power of two the packed fp16 weights are multiplied by
    uint8_t* d_wh = nullptr;       // [groups][npad_h][nchunks_h][32 hi | 32 lo] halves
    float* d_bh = nullptr;         // [groups][npad_h]
};

struct ProfEntry {
    std::string name;
    int launches = 0;
    double total_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    // per-launch record since the last reset (davo_profile_samples; at most PROF_SAMPLES_CAP kept): the launch's own duration
    // and the time from the previous bracketed launch's start to this one's (-1 when the previous start is not known)
    std::vector<float> dur_ms, period_ms;
};
constexpr size_t PROF_SAMPLES_CAP = 8192;

// One in-flight batch: its own HIP stream and activation workspace.  Weights are shared.
struct Slot {
    hipStream_t stream = nullptr;
    float *d_partial = nullptr, *d_tab = nullptr, *d_packed = nullptr, *d_pose_partial = nullptr;
    float* d_act[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned* d_counters = nullptr;              // "last workgroup" tickets (pose_tail.h): [0] cnv7's pose tail, [1 + b] triplet b's squeeze, [1 + max_batch + t] tile t of a split-K launch
};

// A batch davo_forward_device has issued whose f16x3 range record has not been judged yet.  Every such batch owns one slot of a
// small ring: a range record of its own and - unless the caller declared its inputs stable - room for a context-owned copy of
// its inputs, which the batch's last kernel fills if (and only if) the record will fail the verdict, so that the re-issue reads
// exactly what was issued whatever the caller has done to its buffers since (api.hip, prologue.h).
constexpr int RANGE_RING = 8;
constexpr int SK_TILE_COUNTERS = 256;            // tiles of a split-K launch whose fix-up is folded in (forward.hip): at most one per CU
struct Ticket {
    int B;
    const void *img, *flow, *seg;              // what a re-issue reads: the ring slot's snapshot, or the caller's buffers ("stable_inputs")
    void* pose;
    int ring;
    bool snap;                                 // img / flow / seg are the ring slot's copy
    bool frozen;                               // raw holds the batch's record (read before the ring's records were reset)
    unsigned raw[RANGE_WORDS];
    unsigned seq;                              // the batch's sequence number: its last kernel writes it into the slot's host mirror
    hipStream_t stream;                        // the stream it was issued on
    int shifts[6];                             // storage scales the batch was issued under (davo_activation_range reports true magnitudes)
    unsigned long long issue;                  // issue number of the batch (pose_superseded, api.hip)
    float* h_pose;                             // davo_submit batches: page-locked host copy of `pose`, refreshed after a re-issue (else null)
};

// A batch davo_submit has issued whose poses have not been delivered to the caller's array yet (api.hip: streaming entry point).
// Its inputs live in the staging set of its in-flight slot (at most STREAM_SETS slots), its poses in entry `pr` of a ring of STREAM_POSES device
// buffers with page-locked host twins, so neither a re-issue nor the caller's buffer recycling can touch another batch's data.
constexpr int STREAM_SETS = 4, STREAM_POSES = 8;
struct StreamJob {
    int B;
    float* pose_out;                           // the caller's [B,2,6] (pageable is fine: written by the host at delivery)
    int pr;                                    // pose ring entry
    bool ticketed;                             // has a range ticket (f16x3) that must be judged before delivery
    unsigned seq;                              // ... its sequence number
};

struct PoseSpan { uintptr_t lo, hi; unsigned long long issue; };      // the pose buffer range of an issued batch (api.hip: pose_superseded)

struct Comm;                                     // comm.hip: RCCL communicator state

}  // namespace davo

struct davo_ctx {
    int device = 0, H = 0, W = 0, max_batch = 0;
    std::vector<davo::Slot> slots;             // slots[0] is created by davo_create
    int inflight = 1, next_slot = 0;
    int dev_cus = 256;                         // compute units of the device (hipDeviceProp_t::multiProcessorCount, read by davo_create)
    int ncu = 256;                             // compute units a launch of this context may use (CU-masked slot streams: dev_cus / slots)
    bool cu_partition = false;                 // davo_set_option "cu_partition": slot i's stream is masked to its own share of every XCD's CUs
    bool user_stream = false;
    bool opt_fuse_pose = true;                 // f16x3: pose head fused into cnv7's epilogue (davo_set_option)
    int opt_fuse_pack = -1;                    // f16x3: mask+pack fused into cnv1's patch fill: 0 off | 1 on | -1 where it pays (small batches: one launch fewer)
    bool opt_patch_cnv2 = true;                // f16x3: cnv2 from an LDS-staged input patch (conv_patch_cnv2_h3) instead of the implicit GEMM
    bool opt_patch_cnv3 = true;                // f16x3: cnv3 likewise (conv_patch_cnv3_h3)
    bool opt_merge_rem = true;                 // f16x3: cnv5 / cnv6 main + remainder launches as one grid (conv_igemm_h3_mainrem)
    int opt_fold_tails = -1;                   // 0 off | 1 both tails | 2 the excitation only | -1 auto: the excitation at small batches (pose_tail.h: what it costs)               // the excitation MLP and the pose head's tile sum run in the last workgroup of the squeeze / cnv7 launch
    int opt_wave128 = 2;                       // f16x3: cnv5 / cnv6 256x256 tiles on four waves of 128x128 outputs (conv_igemm_h3w.h: -3..5 % per step, bit-identical); 2: their remainder rows on 256x64 tiles too (-1.2 %)
    bool opt_deep_ring = true;                 // f16x3: launches of at most one workgroup per CU (batch 1..4) run on LDS rings of 3..6 slots
    int opt_merge_order = -1;                  // merged grids: 0 = short tiles offset inside every XCD, 1 = per XCD, 2 = main tiles (long first) then the remainder; -1 = 2 where a tile order exists, else 0
    int opt_skip_order = 1;                    // launches whose tiles skip different numbers of padding rows of the filter hand out the long tiles first (tile_order_for, forward.hip): 0 = never, 1 = float32 launches, 2 = the f16x3 merged grids too
    std::map<std::vector<int>, int*> tile_orders;   // device tables of those launches, by (layer, tile rows, tiles, ...); nullptr = uniform
    bool opt_tile_208x128 = false;             // f16x3: cnv4 may run on the four-wave 208x128 tile (conv_igemm_h3s.h; measured 8 % behind the 128x128 tile at B = 32: off)
    bool opt_merge_cnv4 = false;               // f16x3: cnv4 as whole rounds of 256x128 tiles + 128x128 remainder tiles in one grid where the batch allows
    bool opt_share_taps = true;                // f16x3: cnv3..cnv6 stage one pixel patch per filter row for its three taps
    bool opt_f32_n256 = false;                 // f32 mode experiment: cnv5 / cnv6 on the 128 x 256 tile (eight waves, one workgroup per CU)
    int opt_merge_rem_f32 = 1;             // f32 mode: cnv4..cnv7 main + remainder launches as one grid (conv_igemm_f32_mainrem)
    bool opt_f32_n16 = true;                   // f32 mode: cnv1 (16 output channels) on the 128x16 tile / v_mfma_f32_16x16x4_f32 instead of the padded 128x32 one
    bool opt_fold_fixup = false;               // f16x3 split-K: the part that finishes a tile last adds its partial sums (no splitk_fixup launch); needs xcd_rr > 0.  Measured slower (batch 1: 0.141 against 0.132 ms): off
    int xcd_rr = -1;                           // workgroups (x, y) of a grid whose x extent is a multiple of 8 share an XCD for every y: -1 not probed yet | 0 no | 1 yes
    bool opt_split_k = true;                   // f16x3: cnv5 / cnv6 launches of at most half a workgroup per CU split their K loop in two (forward.hip)
    float* d_splitk = nullptr;                 // split-K partial sums [4 slots][M][2][N] float32
    size_t splitk_floats = 0;                  // ... per slot
    float* d_pose_tiles = nullptr;             // per-tile partial sums of the fused pose head
    size_t pose_tiles_floats = 0;
    bool cnv7_valid = true;
    davo::Variant v{};
    int impl = 0;
    int precision = 1;                         // 0 = FP32 MFMA (bit-exact fmaf chains), 1 = f16x3 split (default)
    bool packed_h_ready = false;
    int weight_channel_spread_log2 = 0;        // largest log2 spread of the per-input-channel weight norms over cnv2..cnv7 (weights.hip)
    std::string weight_channel_spread_layer;   // ... and the tensor that has it
    int last_precision = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;
    std::map<std::string, davo::HostTensor> weights;
    std::vector<std::string> needed;
    bool packed_ready = false;                 // float32 convolution weights packed (built at the first float32 forward)
    bool pred_ready = false;                   // pose head kernels on the device (every mode)
    davo::ConvLayer L[7];                      // cnv1..cnv5, cnv6 (fused), cnv7 (grouped)
    float *d_wpred = nullptr, *d_bpred = nullptr;
    uint8_t* d_w1patch = nullptr;              // cnv1 B fragments for conv_patch_cnv1_h3
    uint8_t* d_w2patch = nullptr;              // cnv2 B fragments for conv_patch_cnv2_h3
    float *d_w1patch_f32 = nullptr, *d_w2patch_f32 = nullptr, *d_w3patch_f32 = nullptr;      // float32 mode: cnv1 / cnv2 / cnv3 weights in the patch kernels' register order (conv_patch_f32.h)
    bool opt_patch_f32 = true;                 // float32 mode: cnv1 / cnv2 / cnv3 from an LDS-staged input patch (conv_patch_f32.h) instead of the implicit GEMM
    uint8_t* d_w3patch = nullptr;              // cnv3 B fragments for conv_patch_cnv3_h3
    // geometry
    int H1, W1, H2, W2, H3, W3;
    // workspace
    float *d_partial = nullptr, *d_tab = nullptr, *d_packed = nullptr, *d_zeros = nullptr, *d_pose_partial = nullptr;
    unsigned* d_counters = nullptr;            // the active slot's ticket counters
    float* d_act[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t act_floats_per_img[7];
    int act_ch[7];
    int packed_ld = 8;
    int last_B = 0;
    bool packed_valid = true;                  // false when cnv1 consumed the raw inputs directly (fused)
    const void *last_img = nullptr, *last_flow = nullptr, *last_seg = nullptr;
    int last_plan[7][2] = {};                  // per layer, per launch: 128-row M tiles * 1000 + tile id / BN (reported by the bench)
    // host-API staging
    void *s_img = nullptr, *s_flow = nullptr, *s_seg = nullptr, *s_pose = nullptr;
    float* h_sync_pose = nullptr;              // davo_forward: page-locked bounce buffer of the poses
    hipStream_t copy_stream = nullptr;         // H2D of the next sub-batch runs here while the previous one computes
    std::vector<hipEvent_t> copy_done;
    // f16x3 range management: activations are stored as fp16 pairs scaled by 2^act_shift[layer] (davo_calibrate);
    // every storing epilogue atomicMax-es the largest stored magnitude into d_range[layer]
    int act_shift[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned* d_range_base = nullptr;          // [1 + RANGE_RING][RANGE_WORDS] (params.h): record 0 serves the host path, calibration and re-issues; 1.. the ring
    unsigned* d_range = nullptr;               // the record the next launches write to (one of the above)
    bool range_zero = false;                   // the next forward's first kernel zeroes that record itself (ticketed device-path batches)
    // Range recovery (davo_set_option "auto_range", default on): every device-path batch is judged on a record of its own, at the
    // latest when its ring slot is needed again (RANGE_RING batches later) or at davo_synchronize; a failed verdict re-issues that
    // batch from the context's own copy of its inputs - recalibrated, or on the float32 kernels (api.hip)
    bool opt_auto_range = true;
    bool opt_stable_inputs = false;            // "stable_inputs": the caller keeps inputs unchanged until the verdict, no copies are taken
    std::deque<davo::Ticket> tickets;
    bool ring_busy[davo::RANGE_RING] = {};
    int ring_next = 0;
    void *snap_img[davo::RANGE_RING] = {}, *snap_flow[davo::RANGE_RING] = {}, *snap_seg[davo::RANGE_RING] = {};
    davo::SnapArgs snap{};                     // set around a ticketed batch: its last kernel copies the inputs if the record fails (prologue.h)
    hipStream_t read_stream = nullptr;
    unsigned batch_seq = 0, snap_seq_issued = 0;   // sequence number of the last ticketed batch (never 0 for a batch)
    unsigned* h_range = nullptr;               // page-locked: [0] landing pad of a record read, [1 + r] the mirror ring slot r's last kernel writes
    unsigned* h_range_dev = nullptr;           // ... as the device sees it
    float range_seen[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // largest true |activation| judged since the last reset
    int sticky_range_rc = 0;                   // "auto_range" 0: a failed verdict met while issuing is reported by the next davo_synchronize
    std::string sticky_range_err;
    bool f32_fallback = false;                 // the last forward_device ran the float32 kernels because of the weight guard
    std::string range_report;                  // what the range management last did, in words (davo_range_report)
    long long n_recalibrations = 0, n_f32_batches = 0, n_reissued = 0;
    int host_chunk = 8;                        // davo_forward: windows per sub-batch (davo_set_option "host_chunk"; 0 = whole batch)
    // streaming host entry (davo_submit / davo_wait): staging input sets, pose ring, undelivered batches in issue order
    void *st_img[davo::STREAM_SETS] = {}, *st_flow[davo::STREAM_SETS] = {}, *st_seg[davo::STREAM_SETS] = {};     // one staging set per in-flight slot
    hipEvent_t st_copied[davo::STREAM_POSES] = {};             // "the H2D copies of the batch in pose ring entry k are done" (recorded only for hold < STREAM_POSES)
    bool copy_tracked[davo::STREAM_POSES] = {};
    float *d_pose_ring[davo::STREAM_POSES] = {}, *h_pose_ring[davo::STREAM_POSES] = {};
    hipEvent_t pose_done[davo::STREAM_POSES] = {};
    std::deque<davo::StreamJob> jobs;
    unsigned long long n_submitted = 0;
    float* d_reissue_pose = nullptr;           // a re-issued batch writes here first; copied to its own pose buffer unless a later batch has taken that
    std::deque<davo::PoseSpan> pose_spans;     // pose buffer ranges of the batches issued since the oldest pending ticket
    unsigned long long n_issued = 0;
    int host_since_fresh = 1 << 30;            // davo_forward calls since the base record was last zeroed (api.hip); the first call starts afresh
    int since_fresh_record = 0;                // ticketed batches since one last started from a zeroed range record (api.hip: ticket_begin)
    // profiling
    bool prof = false;
    bool prof_dominant_only = false;           // profile mode 2: bracket only the main cnv6 launch
    int prof_stride = 1, prof_tick = 0;        // ... of every prof_stride-th batch (davo_set_option "profile_stride"): an event pair
                                               // costs two ~6 us bubbles around the launch it brackets
    std::vector<davo::ProfEntry> prof_entries;
    std::vector<hipEvent_t> event_pool;
    // multi-GPU (comm.hip)
    davo::Comm* comm = nullptr;
};

namespace davo {

inline int fail(davo_ctx* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ::davo::fail(c, DAVO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                __FILE__, __LINE__);                                           \
    } while (0)

// TF `SAME` padding (SURVEY.md note P): out = ceil(in/stride), pad_before = total // 2
inline void same_pad(int in, int k, int stride, int rate, int* out, int* before) {
    const int o = (in + stride - 1) / stride;
    const int keff = (k - 1) * rate + 1;
    int total = (o - 1) * stride + keff - in;
    if (total < 0) total = 0;
    *out = o;
    *before = total / 2;
}

inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// ---- profiling: HIP events around a launch on the launch stream -----------------------------
struct ProfScope {
    davo_ctx* c;
    ProfEntry* e = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(davo_ctx* ctx, const char* name);
    ~ProfScope();
};
int prof_collect(davo_ctx* c);
int sync_all_slots(davo_ctx* c);

// ---- weights.hip ----------------------------------------------------------------------------
void init_layer(ConvLayer& L, const char* label, int KS, int stride, int rate, int cin, int cout, int groups);
std::vector<std::string> needed_names(const Variant& v);
bool expected_shape(const davo_ctx* c, const std::string& name, std::vector<int64_t>* sh);
int upload(davo_ctx* c, const std::vector<float>& host, float** dev);
int build_packed_weights(davo_ctx* c);
int build_pred_weights(davo_ctx* c);
int build_packed_weights_h3(davo_ctx* c);
int missing_weights(davo_ctx* c, std::string* names);
float weight_prescale(const float* w, size_t n);
void pack_conv_weights(const float* w_tf, int KS, int cin_tf, int cout, const int* chmap, int cin_packed,
                       int npad, int kpad, float* out);
void pack_conv_weights_h3(const float* w_tf, int KS, int cin_tf, int cout, const int* chmap, int cin_packed,
                          int cb_log2, int tpc_log2, int cpb, int nchunks, float scale, _Float16* out);
inline void split_f16(float v, _Float16* hi, _Float16* lo) {
    const _Float16 h = (_Float16)v;
    *hi = h;
    *lo = (_Float16)(v - (float)h);
}

// ---- forward.hip ----------------------------------------------------------------------------
void activate_slot(davo_ctx* c, int i);
int forward_device(davo_ctx* c, int B, const void* d_img, const void* d_flow, const void* d_seg, void* d_pose);
// f16x3: verdict on the range record (d_range) read back from the device; DAVO_ERR_RANGE names the layer
int check_range(davo_ctx* c, const unsigned* raw /*[RANGE_WORDS]*/, const int* shifts = nullptr);

// ---- comm.hip -------------------------------------------------------------------------------
void comm_release(davo_ctx* c);

}  // namespace davo
