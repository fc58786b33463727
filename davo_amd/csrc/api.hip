// api.hip — the extern "C" entry points of libdavo_hip.so (include/davo_hip.h): context life
// cycle, weight loading, the host- and device-buffer forward calls, f16x3 range management,
// measurement and test hooks.  The forward plan itself is forward.hip; kernels are reached
// through launch.h; the RCCL communicator is comm.hip.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>

#include "ctx.h"
#include "launch.h"
#include "plan.h"

using namespace davo;

namespace {

// hipMemset runs on the null stream and may return before the fill has run; the context's streams are non-blocking, so nothing
// orders them behind it.  (Found as one wrong batch in twenty streamed runs: the zero fill of a slot's new staging set landed on
// top of the first batch's freshly copied flow planes, profiles/r05f_stream_flake.log.)  Returns when the bytes ARE zero.
int zero_now(davo_ctx* c, void* p, size_t bytes) {
    HIP_TRY(c, hipMemset(p, 0, bytes));
    HIP_TRY(c, hipStreamSynchronize(nullptr));
    return DAVO_OK;
}

void free_slot(Slot& s) {
    for (auto p : s.d_act) if (p) (void)hipFree(p);
    void* misc[] = {s.d_partial, s.d_tab, s.d_packed, s.d_pose_partial, s.d_counters};
    for (auto p : misc) if (p) (void)hipFree(p);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    s = Slot();
}

// allocate one in-flight slot (stream + activation workspace for max_batch triplets)
int alloc_slot(davo_ctx* c, Slot* s) {
    const size_t NB = 2 * (size_t)c->max_batch;
    HIP_TRY(c, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    for (int i = 0; i < 7; ++i)
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&s->d_act[i]), NB * c->act_floats_per_img[i] * sizeof(float)));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&s->d_packed), NB * (size_t)c->H * c->W * 10 * sizeof(float)));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&s->d_partial), (size_t)c->max_batch * 2 * SQ_CHUNKS * 2 * sizeof(float)));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&s->d_tab), (size_t)c->max_batch * 3 * NCLS * sizeof(float)));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&s->d_pose_partial), NB * 2 * PH_SPLIT * 3 * sizeof(float)));
    { int rc = zero_now(c, s->d_partial, (size_t)c->max_batch * 2 * SQ_CHUNKS * 2 * sizeof(float)); if (rc) return rc; }
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&s->d_counters), ((size_t)c->max_batch + 1 + SK_TILE_COUNTERS) * sizeof(unsigned)));
    { int rc = zero_now(c, s->d_counters, ((size_t)c->max_batch + 1 + SK_TILE_COUNTERS) * sizeof(unsigned)); if (rc) return rc; }
    return DAVO_OK;
}

}  // namespace

// ============================================================================================
extern "C" {

int davo_create(davo_ctx** out, int device, int H, int W, int max_batch, const davo_variant* v) {
    if (!out || !v) return DAVO_ERR_INVALID;
    *out = nullptr;
    davo_ctx* c = new davo_ctx();
    *out = c;                                   // returned even on failure so the message is readable
    c->device = device; c->H = H; c->W = W; c->max_batch = max_batch;
    c->v = Variant{v->cin_per_frame, v->cnv6_out, v->se_act, v->norm_flow, v->abs_mode, v->att_source,
                   v->mask_rgb, v->mask_info};
    if (H < 16 || W < 16 || H % 4 || W % 4) return fail(c, DAVO_ERR_INVALID, "H and W must be multiples of 4 and >= 16 (got %dx%d)", H, W);
    if (max_batch < 1) return fail(c, DAVO_ERR_INVALID, "max_batch must be >= 1");
    if (v->cin_per_frame != 5 && v->cin_per_frame != 3) return fail(c, DAVO_ERR_INVALID, "cin_per_frame must be 3 or 5");
    if (ilog2_exact(v->cnv6_out) < 5 || v->cnv6_out > 256) return fail(c, DAVO_ERR_INVALID, "cnv6_out must be 32, 64, 128 or 256");
    if (v->se_act < 0 || v->se_act > 2 || v->abs_mode < 0 || v->abs_mode > 3 || v->att_source < 0 || v->att_source > 3)
        return fail(c, DAVO_ERR_INVALID, "variant field out of range");
    int ndev = 0;
    HIP_TRY(c, hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(c, DAVO_ERR_INVALID, "device %d not present (%d visible)", device, ndev);
    HIP_TRY(c, hipSetDevice(device));
    {   // the launch planner sizes whole rounds for the CUs this device really has (a partitioned MI355X exposes fewer than 256)
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, device));
        c->dev_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        c->ncu = c->dev_cus;
    }
    c->needed = needed_names(c->v);

    c->H1 = (H + 1) / 2; c->W1 = (W + 1) / 2;
    c->H2 = (c->H1 + 1) / 2; c->W2 = (c->W1 + 1) / 2;
    c->H3 = (c->H2 + 1) / 2; c->W3 = (c->W2 + 1) / 2;
    const int c6 = c->v.cnv6_out;
    init_layer(c->L[0], "cnv1", 7, 2, 1, 8, 16, 1);
    init_layer(c->L[1], "cnv2", 5, 2, 1, 16, 32, 1);
    init_layer(c->L[2], "cnv3", 3, 1, 2, 32, 64, 1);
    init_layer(c->L[3], "cnv4", 3, 1, 4, 64, 128, 1);
    init_layer(c->L[4], "cnv5", 3, 1, 8, 128, 256, 1);
    init_layer(c->L[5], "cnv6", 3, 1, 2, 256, 2 * c6, 1);
    init_layer(c->L[6], "cnv7", 3, 2, 1, c6, 256, 2);

    const int ch[7] = {16, 32, 64, 128, 256, 2 * c6, 512};
    const size_t px[7] = {(size_t)c->H1 * c->W1, (size_t)c->H2 * c->W2, (size_t)c->H2 * c->W2, (size_t)c->H2 * c->W2,
                          (size_t)c->H2 * c->W2, (size_t)c->H2 * c->W2, (size_t)c->H3 * c->W3};
    for (int i = 0; i < 7; ++i) {
        c->act_ch[i] = ch[i];
        c->act_floats_per_img[i] = px[i] * ch[i];
    }
    c->slots.resize(1);
    { int rc = alloc_slot(c, &c->slots[0]); if (rc) return rc; }
    c->own_stream = c->slots[0].stream;
    activate_slot(c, 0);
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_zeros), 256));
    { int rc = zero_now(c, c->d_zeros, 256); if (rc) return rc; }
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_range_base), (1 + RANGE_RING) * RANGE_WORDS * sizeof(unsigned)));
    { int rc = zero_now(c, c->d_range_base, (1 + RANGE_RING) * RANGE_WORDS * sizeof(unsigned)); if (rc) return rc; }
    c->d_range = c->d_range_base;
    return DAVO_OK;
}

int davo_load_weight(davo_ctx* c, const char* tf_name, const float* data, const int64_t* shape, int ndim) {
    if (!c || !tf_name || !data || !shape || ndim < 1 || ndim > 4) return fail(c, DAVO_ERR_INVALID, "bad argument to davo_load_weight");
    std::vector<int64_t> want;
    if (!expected_shape(c, tf_name, &want)) return fail(c, DAVO_ERR_INVALID, "unknown variable `%s'", tf_name);
    bool listed = false;
    for (auto& n : c->needed) listed |= (n == tf_name);
    if (!listed) return fail(c, DAVO_ERR_INVALID, "variable `%s' is not part of this variant", tf_name);
    std::vector<int64_t> got(shape, shape + ndim);
    if (got != want) {
        std::string g, w;
        for (auto d : got) g += std::to_string(d) + ",";
        for (auto d : want) w += std::to_string(d) + ",";
        return fail(c, DAVO_ERR_INVALID, "`%s': shape [%s] does not match expected [%s]", tf_name, g.c_str(), w.c_str());
    }
    size_t n = 1;
    for (auto d : got) n *= (size_t)d;
    HostTensor& t = c->weights[tf_name];
    t.shape = got;
    t.data.assign(data, data + n);
    HIP_TRY(c, hipSetDevice(c->device));
    // the kernels read the small dense tensors (SE fully-connected layers, static channel weights) in the reference's own layout;
    // the convolution tensors are re-laid-out at the first forward (weights.hip) and their raw device copy is only the test hook's
    // (impl 1: uploaded on demand, forward.hip) - 20 hipMalloc + copies less in front of a rank's first batch
    const std::string nm = tf_name;
    const bool dense = nm.find("se_flow") != std::string::npos || nm.find("seg_channel_weight") != std::string::npos;
    if (t.dev) { (void)hipFree(t.dev); t.dev = nullptr; }
    if (dense) { int rc = upload(c, t.data, &t.dev); if (rc) return rc; }
    c->packed_ready = false;
    c->packed_h_ready = false;
    c->pred_ready = false;
    return DAVO_OK;
}

int davo_weights_missing(davo_ctx* c) {
    if (!c) return DAVO_ERR_INVALID;
    std::string names;
    const int n = missing_weights(c, &names);
    if (n) c->err = "weights not loaded: " + names;
    return n;
}

// ---- f16x3 range verdicts and recovery -----------------------------------------------------------------
// The reference's float32 graph (nets/posenn.py:205-215, davo.py:1553-1569) never fails on a finite network, so the
// default f16x3 arithmetic must not either: a batch whose range record fails the verdict is re-issued here - first
// with the storage scales re-calibrated on that very batch, and if it still leaves the fp16-pair range, on the
// library's own float32 kernels (davo_set_precision(ctx, 0) for that batch only).  "auto_range" 0 restores the plain
// DAVO_ERR_RANGE verdict.
//
// Device path (round 4).  Every batch davo_forward_device issues owns one slot of a ring of RANGE_RING: a record of its own
// (zeroed by the forward's first kernel) and, unless the caller declared "stable_inputs", room for a copy of its inputs.  The
// batch's LAST kernel reads the finished record and, if a layer left the range, copies the inputs it was issued on into the
// slot (prologue.h: snapshot_inputs_if_range_fails) - in stream order behind the kernels that read them and ahead of anything
// the caller orders behind the batch, e.g. the next H2D into the same buffers.  A batch in range costs six loads per thread and
// no copy.  A batch is judged when its slot is needed again, at davo_synchronize, or before anything that changes the scales;
// a failed verdict re-issues THAT batch from the slot's copy, so a streaming caller that recycles its input buffers still gets
// float32-grade poses for every batch.  (First built with an unconditional side-stream copy: +1.6 % of the step at B = 32,
// profiles/r04_snapshot_ab.log.)
namespace {

constexpr int RING = RANGE_RING;
constexpr int FRESH_EVERY = 256;

size_t img_bytes(const davo_ctx* c) { return (size_t)c->H * c->W * 9; }
size_t flow_bytes(const davo_ctx* c) { return (size_t)c->H * c->W * 8 * sizeof(float); }
size_t seg_bytes(const davo_ctx* c) { return (size_t)c->H * c->W * 3 * sizeof(float); }

int ensure_ring(davo_ctx* c, bool snapshots) {
    if (!c->read_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->read_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_range), (1 + RANGE_RING) * RANGE_WORDS * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_range_dev), c->h_range, 0));
        memset(c->h_range, 0, (1 + RANGE_RING) * RANGE_WORDS * sizeof(unsigned));
    }
    if (snapshots && !c->snap_img[0]) {
        for (int r = 0; r < RING; ++r) {
            HIP_TRY(c, hipMalloc(&c->snap_img[r], img_bytes(c) * c->max_batch));
            HIP_TRY(c, hipMalloc(&c->snap_flow[r], flow_bytes(c) * c->max_batch));
            HIP_TRY(c, hipMalloc(&c->snap_seg[r], seg_bytes(c) * c->max_batch));
        }
    }
    return DAVO_OK;
}

// a record -> host, on a stream of its own (never behind queued batches, never through the null stream)
int read_record(davo_ctx* c, const unsigned* d_rec, unsigned raw[RANGE_WORDS]) {
    if (!c->read_stream) { int rc = ensure_ring(c, false); if (rc) return rc; }
    HIP_TRY(c, hipMemcpyAsync(c->h_range, d_rec, RANGE_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, c->read_stream));
    HIP_TRY(c, hipStreamSynchronize(c->read_stream));
    memcpy(raw, c->h_range, RANGE_WORDS * sizeof(unsigned));
    return DAVO_OK;
}

unsigned* ring_record(davo_ctx* c, int r) { return c->d_range_base + RANGE_WORDS * (1 + r); }

void note_seen(davo_ctx* c, const unsigned raw[6], const int* shifts) {
    for (int i = 0; i < 6; ++i) {
        float v;
        memcpy(&v, &raw[i], sizeof v);
        const float t = ldexpf(v, -shifts[i]);
        if (!(t <= c->range_seen[i])) c->range_seen[i] = t;          // NaN / inf records stay visible
    }
}

int zero_base_record(davo_ctx* c, hipStream_t s) {
    HIP_TRY(c, hipMemsetAsync(c->d_range_base, 0, RANGE_WORDS * sizeof(unsigned), s));
    return DAVO_OK;
}

// Every stream idle.  The ring's running maxima (params.h) are about to lose their meaning - a failed verdict, or the scales are
// going to change: read the record of every batch that is still waiting for its verdict into its ticket first, then reset the ring.
int freeze_pending_and_reset_ring(davo_ctx* c) {
    for (Ticket& t : c->tickets)
        if (!t.frozen) {
            // every stream the context knows is idle, so the mirrors are final - unless the batch went out on a caller's stream the
            // context no longer runs on (davo_set_stream judges its tickets before a switch; this is the belt to those braces)
            const unsigned* m = c->h_range + RANGE_WORDS * (1 + t.ring);
            if (__atomic_load_n(&m[RANGE_SEQ], __ATOMIC_ACQUIRE) != t.seq) {
                HIP_TRY(c, hipStreamSynchronize(t.stream));
                if (__atomic_load_n(&m[RANGE_SEQ], __ATOMIC_ACQUIRE) != t.seq) return fail(c, DAVO_ERR_HIP, "a batch finished without reporting its range record");
            }
            memcpy(t.raw, m, sizeof t.raw);
            t.frozen = true;
        }
    { int rc = zero_now(c, c->d_range_base + RANGE_WORDS, RANGE_RING * RANGE_WORDS * sizeof(unsigned)); if (rc) return rc; }
    return DAVO_OK;
}

// power-of-two storage scales from a sample batch: each pass runs the path and moves every layer's largest stored
// value into [512, 1024).  A layer computed from badly ranged inputs still has about the right magnitude, so each
// pass fixes at least the first badly ranged layer exactly and the later ones to within a few powers of two.
// Runs on the base record; every stream must be idle.
int calibrate_on(davo_ctx* c, int B, const void* d_img, const void* d_flow, const void* d_seg, void* d_pose) {
    int rc = DAVO_OK;
    const int save_precision = c->precision, save_impl = c->impl;
    c->precision = 1; c->impl = 0;
    c->d_range = c->d_range_base;
    for (int pass = 0; pass < 8 && rc == DAVO_OK; ++pass) {
        if ((rc = zero_base_record(c, c->stream))) break;
        rc = forward_device(c, B, d_img, d_flow, d_seg, d_pose);
        if (rc) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(c, DAVO_ERR_HIP, "hipStreamSynchronize failed"); break; }
        unsigned raw[RANGE_WORDS];
        if ((rc = read_record(c, c->d_range_base, raw))) break;
        bool changed = false;
        for (int i = 0; i < 6; ++i) {
            float v;
            memcpy(&v, &raw[i], sizeof v);
            int delta = 0;
            if (!std::isfinite(v)) delta = -32;
            else if (v > 0.f) { int e; (void)frexpf(v, &e); delta = 10 - e; }        // stored max -> [2^9, 2^10): 64x headroom
            int ns = c->act_shift[i] + delta;
            ns = ns < -60 ? -60 : (ns > 60 ? 60 : ns);
            if (ns != c->act_shift[i]) { c->act_shift[i] = ns; changed = true; }
        }
        if (!changed) break;
    }
    c->precision = save_precision; c->impl = save_impl;
    (void)zero_base_record(c, c->stream);
    return rc;
}

// one batch, synchronously, on the base record; -> DAVO_OK, DAVO_ERR_RANGE (the verdict) or a hard error
int run_judged(davo_ctx* c, const Ticket& b) {
    c->d_range = c->d_range_base;
    { int rc = zero_base_record(c, c->stream); if (rc) return rc; }
    int rc = forward_device(c, b.B, b.img, b.flow, b.seg, b.pose);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->last_precision != 1) return DAVO_OK;
    unsigned raw[RANGE_WORDS];
    if ((rc = read_record(c, c->d_range_base, raw))) return rc;
    rc = check_range(c, raw);
    if (rc == DAVO_OK) note_seen(c, raw, c->act_shift);
    return rc;
}

// A failed verdict: re-issue the batch - as issued if the scales have moved since and now hold it, re-calibrated on itself if
// not, on the float32 kernels if even that leaves the range (per-layer scales cannot cover e.g. an inf / NaN producing net).
// Drains every stream first: the re-issue uses slot 0's workspace and the base record.
// A caller may have handed the batch's pose buffer to a LATER batch since (two alternating buffers, one buffer overwritten every
// step): the re-issue therefore writes into a pose buffer of the context and is copied to the caller's only if no batch
// issued after this one targets an overlapping range - the newest writer of a buffer always wins (pose_spans: davo_forward_device).
bool pose_superseded(const davo_ctx* c, const Ticket& t) {
    const uintptr_t lo = (uintptr_t)t.pose, hi = lo + (size_t)t.B * 12 * sizeof(float);
    for (const PoseSpan& sp : c->pose_spans)
        if (sp.issue > t.issue && sp.lo < hi && lo < sp.hi) return true;
    return false;
}

int recover_batch(davo_ctx* c, const Ticket& orig) {
    { int rc = sync_all_slots(c); if (rc) return rc; }
    const std::string verdict = c->err;
    { int rc = freeze_pending_and_reset_ring(c); if (rc) return rc; }       // the failed slot's maximum must go; the scales may move
    if (!c->d_reissue_pose) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_reissue_pose), (size_t)c->max_batch * 12 * sizeof(float)));
    Ticket b = orig;
    b.pose = c->d_reissue_pose;
    activate_slot(c, 0);
    int rc = run_judged(c, b);
    if (rc == DAVO_ERR_RANGE) {
        if ((rc = calibrate_on(c, b.B, b.img, b.flow, b.seg, b.pose))) return rc;
        ++c->n_recalibrations;
        rc = run_judged(c, b);
        c->range_report = "re-calibrated: " + verdict;
    }
    if (rc == DAVO_ERR_RANGE) {
        const int save = c->precision;
        c->precision = 0;
        rc = forward_device(c, b.B, b.img, b.flow, b.seg, b.pose);
        c->precision = save;
        if (rc == DAVO_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, DAVO_ERR_HIP, "hipStreamSynchronize failed");
        ++c->n_f32_batches;
        c->range_report = "float32 kernels for one batch: " + verdict;
    }
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!pose_superseded(c, orig))
        HIP_TRY(c, hipMemcpy(orig.pose, c->d_reissue_pose, (size_t)orig.B * 12 * sizeof(float), hipMemcpyDeviceToDevice));
    ++c->n_reissued;
    c->err.clear();
    return DAVO_OK;
}

// verdict on the oldest unjudged device-path batch (waits for that batch only)
int judge_front(davo_ctx* c) {
    const Ticket t = c->tickets.front();
    c->tickets.pop_front();
    unsigned raw[RANGE_WORDS];
    int rc = DAVO_OK;
    if (t.frozen) memcpy(raw, t.raw, sizeof raw);        // read when the ring was reset (every stream was idle then)
    else {
        // the batch's last kernel writes its sequence number into the slot's host mirror behind the maxima (prologue.h): poll that,
        // with the stream's own state as the way out if the device has failed
        volatile const unsigned* m = c->h_range + RANGE_WORDS * (1 + t.ring);
        for (unsigned spin = 0; m[RANGE_SEQ] != t.seq; ++spin) {
            if ((spin & 1023u) == 1023u) {
                const hipError_t q = hipStreamQuery(t.stream);
                if (q == hipSuccess) {                                    // the stream is idle: the store is on its way or the kernel never ran
                    if (m[RANGE_SEQ] == t.seq) break;
                    HIP_TRY(c, hipStreamSynchronize(t.stream));
                    if (m[RANGE_SEQ] != t.seq) return fail(c, DAVO_ERR_HIP, "a batch finished without reporting its range record");
                    break;
                }
                if (q != hipErrorNotReady) return fail(c, DAVO_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        for (int i = 0; i < RANGE_WORDS; ++i) raw[i] = m[i];
    }
    if (rc == DAVO_OK) {
        rc = check_range(c, raw, t.shifts);
        if (rc == DAVO_OK) note_seen(c, raw, t.shifts);
        else if (rc == DAVO_ERR_RANGE && c->opt_auto_range) {
            // the batch's last kernel reached the same verdict on the same record and kept the inputs (prologue.h)
            if (t.snap && raw[RANGE_SNAP] != 1u) rc = fail(c, DAVO_ERR_INVALID, "internal: a batch failed its range verdict but its inputs were not kept");
            else {
                rc = recover_batch(c, t);
                // a davo_submit batch: the re-issue rewrote the pose ring entry, its page-locked twin follows (every stream is idle)
                if (rc == DAVO_OK && t.h_pose && hipMemcpy(t.h_pose, t.pose, (size_t)t.B * 12 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
                    rc = fail(c, DAVO_ERR_HIP, "copying re-issued poses to the host failed");
            }
        } else if (rc == DAVO_ERR_RANGE && !t.frozen) {
            // no recovery ("auto_range" 0): the slot's running maximum has served its verdict - the next batch starts afresh
            const std::string keep = c->err;
            if (hipMemsetAsync(ring_record(c, t.ring), 0, RANGE_WORDS * sizeof(unsigned), c->read_stream) != hipSuccess ||
                hipStreamSynchronize(c->read_stream) != hipSuccess) rc = fail(c, DAVO_ERR_HIP, "resetting a range record failed");
            else c->err = keep;
        }
    }
    c->ring_busy[t.ring] = false;              // after the re-issue: it read the slot's copy of the inputs
    return rc;
}

// every unjudged batch; a failed verdict with "auto_range" 0 does not stop the others from being judged
int judge_all(davo_ctx* c) {
    int first = c->sticky_range_rc;
    std::string first_err = c->sticky_range_err;
    c->sticky_range_rc = 0; c->sticky_range_err.clear();
    while (!c->tickets.empty()) {
        const int rc = judge_front(c);
        if (rc == DAVO_ERR_RANGE) { if (!first) { first = rc; first_err = c->err; } }
        else if (rc) { for (auto& t : c->tickets) c->ring_busy[t.ring] = false; c->tickets.clear(); return rc; }
    }
    { int rc = sync_all_slots(c); if (rc) return rc; }
    if (first) { c->err = first_err; return first; }
    return DAVO_OK;
}

// the batch about to be issued takes ring slot ring_next: judge what still holds it (may re-issue: before the slot rotation)
// every batch davo_forward_device issues leaves the range of its pose buffer here (pose_superseded); entries no pending ticket can
// be older than are dropped
void note_pose_span(davo_ctx* c, const void* d_pose, int B) {
    ++c->n_issued;
    const unsigned long long oldest = c->tickets.empty() ? c->n_issued : c->tickets.front().issue;
    while (!c->pose_spans.empty() && c->pose_spans.front().issue <= oldest) c->pose_spans.pop_front();
    if (!c->tickets.empty()) c->pose_spans.push_back(PoseSpan{(uintptr_t)d_pose, (uintptr_t)d_pose + (size_t)B * 12 * sizeof(float), c->n_issued});
}

int ticket_reserve(davo_ctx* c) {
    while (c->ring_busy[c->ring_next]) {
        const int rc = judge_front(c);
        if (rc == DAVO_ERR_RANGE && !c->opt_auto_range) {          // reported by the next davo_synchronize; this batch is issued all the same
            if (!c->sticky_range_rc) { c->sticky_range_rc = rc; c->sticky_range_err = c->err; }
        } else if (rc) return rc;
    }
    return DAVO_OK;
}

// ... the batch's kernels record into the slot's record; its last kernel keeps the inputs there if the record fails
int ticket_begin(davo_ctx* c, int B, const void* d_img, const void* d_flow, const void* d_seg, bool* snap, bool own_inputs = false) {
    const int r = c->ring_next;
    // own_inputs: the batch reads a staging set of the context (davo_submit), which the next batches overwrite whatever the caller declared
    *snap = c->opt_auto_range && (!c->opt_stable_inputs || own_inputs);
    if (*snap && (((uintptr_t)d_img | (uintptr_t)d_flow | (uintptr_t)d_seg) & 15)) return fail(c, DAVO_ERR_INVALID, "device input buffers must be 16-byte aligned");
    { int rc = ensure_ring(c, *snap); if (rc) return rc; }
    c->d_range = ring_record(c, r);
    c->range_zero = true;
    // The records hold RUNNING maxima (params.h): "clamped" is exact per batch, "too small" is judged on everything a slot has stored
    // since its record was last zeroed.  So that a long stream that never synchronises still notices activations that collapse,
    // every FRESH_EVERY-th batch starts from a zeroed record (a memset in stream order ahead of the batch's kernels; that batch pays
    // its first round's atomics, ~0.2 ms, once in FRESH_EVERY batches).
    if (++c->since_fresh_record >= FRESH_EVERY) {
        c->since_fresh_record = 0;
        HIP_TRY(c, hipMemsetAsync(c->d_range, 0, 6 * sizeof(unsigned), c->stream));
    }
    if (++c->batch_seq == 0) c->batch_seq = 1;
    c->snap_seq_issued = c->batch_seq;
    c->snap = SnapArgs{c->d_range, c->h_range_dev + RANGE_WORDS * (1 + r), c->batch_seq,
                       static_cast<const uint8_t*>(d_img), static_cast<const uint8_t*>(d_flow), static_cast<const uint8_t*>(d_seg),
                       *snap ? static_cast<uint8_t*>(c->snap_img[r]) : nullptr, *snap ? static_cast<uint8_t*>(c->snap_flow[r]) : nullptr,
                       *snap ? static_cast<uint8_t*>(c->snap_seg[r]) : nullptr,
                       (unsigned)(img_bytes(c) / 16), (unsigned)(flow_bytes(c) / 32), (unsigned)(flow_bytes(c) / 16), (unsigned)(seg_bytes(c) / 16), B};
    return DAVO_OK;
}

int ticket_end(davo_ctx* c, int rc, int B, const void* d_img, const void* d_flow, const void* d_seg, void* d_pose, bool snap, float* h_pose = nullptr) {
    const int r = c->ring_next;
    c->d_range = c->d_range_base;
    c->range_zero = false;
    c->snap = SnapArgs{};
    if (rc) return rc;
    if (c->f32_fallback) ++c->n_f32_batches;
    if (c->last_precision != 1) return DAVO_OK;                                  // float32 kernels (weight guard): no record, no verdict
    Ticket t{};
    t.B = B; t.img = snap ? c->snap_img[r] : d_img; t.flow = snap ? c->snap_flow[r] : d_flow; t.seg = snap ? c->snap_seg[r] : d_seg;
    t.pose = d_pose; t.ring = r; t.snap = snap; t.seq = c->snap_seq_issued; t.stream = c->stream; t.h_pose = h_pose; t.issue = c->n_issued;
    for (int i = 0; i < 6; ++i) t.shifts[i] = c->act_shift[i];
    c->tickets.push_back(t);
    c->ring_busy[r] = true;
    c->ring_next = (r + 1) % RING;
    return DAVO_OK;
}

}  // namespace

int davo_forward_device(davo_ctx* c, int B, const void* d_img, const void* d_flow, const void* d_seg,
                        void* d_pose, float* elapsed_ms) {
    if (!c) return DAVO_ERR_INVALID;
    if (B < 1 || B > c->max_batch) return fail(c, DAVO_ERR_INVALID, "batch %d outside [1,%d]", B, c->max_batch);
    if (!d_img || !d_flow || !d_seg || !d_pose) return fail(c, DAVO_ERR_INVALID, "null device pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool ticketed = c->impl == 0 && c->precision == 1;      // f16x3: the batch gets a record (and a copy of its inputs) of its own
    if (ticketed) { int rc = ticket_reserve(c); if (rc) return rc; }
    // rotate through the in-flight slots: this batch runs on its own stream and workspace
    activate_slot(c, c->next_slot);
    c->next_slot = (c->next_slot + 1) % c->inflight;
    bool snap = false;
    if (ticketed) { int rc = ticket_begin(c, B, d_img, d_flow, d_seg, &snap); if (rc) return rc; }
    if (!ticketed && c->pose_spans.size() > 64) {          // float32 batches behind pending f16x3 tickets: bounded
        const int rc = judge_all(c);
        if (rc == DAVO_ERR_RANGE) { c->sticky_range_rc = rc; c->sticky_range_err = c->err; }       // "auto_range" 0: reported by the next davo_synchronize
        else if (rc) return rc;
    }
    note_pose_span(c, d_pose, B);
    if (!elapsed_ms) {
        int rc = forward_device(c, B, d_img, d_flow, d_seg, d_pose);
        if (ticketed) rc = ticket_end(c, rc, B, d_img, d_flow, d_seg, d_pose, snap);
        else if (rc == DAVO_OK && c->f32_fallback) ++c->n_f32_batches;
        return rc;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = DAVO_OK;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == DAVO_OK) rc = fail(c, DAVO_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (hip_ok(hipEventCreate(&e0), "hipEventCreate") && hip_ok(hipEventCreate(&e1), "hipEventCreate") &&
        hip_ok(hipEventRecord(e0, c->stream), "hipEventRecord")) {
        rc = forward_device(c, B, d_img, d_flow, d_seg, d_pose);
        if (rc == DAVO_OK && hip_ok(hipEventRecord(e1, c->stream), "hipEventRecord") &&
            hip_ok(hipEventSynchronize(e1), "hipEventSynchronize"))
            hip_ok(hipEventElapsedTime(elapsed_ms, e0, e1), "hipEventElapsedTime");
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (ticketed) rc = ticket_end(c, rc, B, d_img, d_flow, d_seg, d_pose, snap);
    // the timed form is synchronous, so it can judge (and, if need be, re-issue) its own batch; elapsed_ms is the first issue's
    if (rc == DAVO_OK && c->inflight == 1) rc = judge_all(c);
    return rc;
}


// ---- streaming host entry: davo_submit / davo_wait ----------------------------------------------------------------
// The reference's driver pulls batches through tf.data's prefetch(8B) while the session runs (test_kitti_pose.py:133-145,
// data_loader.py:321-324): input transfer, compute and result delivery of neighbouring batches overlap.  davo_forward cannot (it
// returns poses), so the sequence driver uses this pair.  davo_submit rotates through the in-flight slots like
// davo_forward_device; on the slot's OWN stream it queues the H2D copies of the batch into the slot's staging set, the forward,
// and the D2H of the poses into a page-locked ring entry.  A stream runs in order, so the staging set is safe to refill without an
// event, and with two or more slots the copies of batch n+1 run under the kernels of batch n.  (First built with a copy stream
// and events between it and the slots: each hipEventRecord costs 6 us of host time and a marker on the queue,
// profiles/r05a_hip_call_cost.log, and the batch-1 loop ran at 134-168 us per window.)  davo_wait / davo_synchronize judge the
// batch's range ticket (re-issuing it if need be) and only then write the poses into the caller's array.  Inputs, poses and range
// snapshots of a batch all live in context-owned memory.
namespace {

int ensure_stream_state(davo_ctx* c, int slot) {
    if (!c->d_pose_ring[0])
        for (int k = 0; k < STREAM_POSES; ++k) {
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_pose_ring[k]), (size_t)c->max_batch * 12 * sizeof(float)));
            HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_pose_ring[k]), (size_t)c->max_batch * 12 * sizeof(float), hipHostMallocDefault));
            HIP_TRY(c, hipEventCreateWithFlags(&c->pose_done[k], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->st_copied[k], hipEventDisableTiming));
        }
    if (!c->st_img[slot]) {
        HIP_TRY(c, hipMalloc(&c->st_img[slot], img_bytes(c) * c->max_batch));
        HIP_TRY(c, hipMalloc(&c->st_flow[slot], flow_bytes(c) * c->max_batch));
        HIP_TRY(c, hipMalloc(&c->st_seg[slot], seg_bytes(c) * c->max_batch));
        // planes the path never reads (flow 2,3; the target frame's label map) are never copied either: defined contents all the same
        { int rc = zero_now(c, c->st_flow[slot], flow_bytes(c) * c->max_batch); if (rc) return rc; }
        { int rc = zero_now(c, c->st_seg[slot], seg_bytes(c) * c->max_batch); if (rc) return rc; }
    }
    return DAVO_OK;
}

bool ticket_pending(const davo_ctx* c, unsigned seq) {
    for (const Ticket& t : c->tickets) if (t.seq == seq) return true;
    return false;
}

// the oldest undelivered batch: verdict (and re-issue) first, then its poses go to the caller's array
int deliver_front(davo_ctx* c) {
    const StreamJob j = c->jobs.front();
    c->jobs.pop_front();
    while (j.ticketed && ticket_pending(c, j.seq)) {
        const int rc = judge_front(c);
        if (rc == DAVO_ERR_RANGE && !c->opt_auto_range) {          // "auto_range" 0: reported by the next davo_synchronize; the poses are delivered as they are
            if (!c->sticky_range_rc) { c->sticky_range_rc = rc; c->sticky_range_err = c->err; }
        } else if (rc) return rc;
    }
    HIP_TRY(c, hipEventSynchronize(c->pose_done[j.pr]));
    memcpy(j.pose_out, c->h_pose_ring[j.pr], (size_t)j.B * 12 * sizeof(float));
    return DAVO_OK;
}

int deliver_all(davo_ctx* c) {
    while (!c->jobs.empty()) { const int rc = deliver_front(c); if (rc) { c->jobs.clear(); return rc; } }
    return DAVO_OK;
}

}  // namespace

int davo_submit(davo_ctx* c, int B, const uint8_t* img, const float* flow, const float* seg, float* pose_out, int hold) {
    if (!c) return DAVO_ERR_INVALID;
    if (!img || !flow || !seg || !pose_out) return fail(c, DAVO_ERR_INVALID, "null host pointer");
    if (B < 1 || B > c->max_batch) return fail(c, DAVO_ERR_INVALID, "batch %d outside [1,%d]", B, c->max_batch);
    if (hold < 0) return fail(c, DAVO_ERR_INVALID, "hold must be >= 0");
    HIP_TRY(c, hipSetDevice(c->device));
    // deliver what has finished (never blocks), and make room in the pose ring (blocks on the oldest batch only when the ring is full)
    while (!c->jobs.empty() && ((int)c->jobs.size() >= STREAM_POSES || hipEventQuery(c->pose_done[c->jobs.front().pr]) == hipSuccess)) {
        const int rc = deliver_front(c);
        if (rc) return rc;
    }
    const bool ticketed = c->impl == 0 && c->precision == 1;
    if (ticketed) { int rc = ticket_reserve(c); if (rc) return rc; }          // may judge - and re-issue - an older batch: before the slot rotation
    const int slot = c->next_slot, pr = (int)(c->n_submitted % STREAM_POSES);
    { int rc = ensure_stream_state(c, slot); if (rc) return rc; }
    activate_slot(c, slot);
    c->next_slot = (c->next_slot + 1) % c->inflight;
    hipStream_t s = c->stream;
    const size_t nb_img = img_bytes(c), nb_flow = flow_bytes(c), nb_seg = seg_bytes(c);
    // H2D on the slot's stream, in order behind the forward that last read this staging set.  Only what the path reads crosses PCIe:
    // flow planes 0,1 (davo.py:978-982) and, unless the variant reads the target frame's label map too (-segmask_all-static),
    // the two source frames' maps (davo.py:998-1004, 1408-1412).
    HIP_TRY(c, hipMemcpyAsync(c->st_img[slot], img, nb_img * B, hipMemcpyHostToDevice, s));
    if (B == 1) HIP_TRY(c, hipMemcpyAsync(c->st_flow[slot], flow, nb_flow / 2, hipMemcpyHostToDevice, s));
    else HIP_TRY(c, hipMemcpy2DAsync(c->st_flow[slot], nb_flow, flow, nb_flow, nb_flow / 2, B, hipMemcpyHostToDevice, s));
    if (c->v.att_source == 3 || B < 4) HIP_TRY(c, hipMemcpyAsync(c->st_seg[slot], seg, nb_seg * B, hipMemcpyHostToDevice, s));
    else
        for (int plane = 0; plane < 3; plane += 2)
            HIP_TRY(c, hipMemcpy2DAsync((uint8_t*)c->st_seg[slot] + plane * (nb_seg / 3), nb_seg, (const uint8_t*)seg + plane * (nb_seg / 3), nb_seg,
                                        nb_seg / 3, B, hipMemcpyHostToDevice, s));
    // the caller keeps a batch's inputs unchanged for `hold` more submits.  With hold >= STREAM_POSES the pose ring already implies it
    // (a batch is delivered - so its copies are long done - before the eighth submit after it returns): no event then
    const bool track_copy = hold < STREAM_POSES;
    if (track_copy) HIP_TRY(c, hipEventRecord(c->st_copied[pr], s));

    bool snap = false;
    if (ticketed) { int rc = ticket_begin(c, B, c->st_img[slot], c->st_flow[slot], c->st_seg[slot], &snap, true); if (rc) return rc; }
    ++c->n_issued;                    // (no pose span: a pose ring entry is not reused before its batch has been delivered)
    int rc = forward_device(c, B, c->st_img[slot], c->st_flow[slot], c->st_seg[slot], c->d_pose_ring[pr]);
    const unsigned seq = c->snap_seq_issued;
    bool has_ticket = false;
    if (ticketed) {
        const size_t before = c->tickets.size();
        rc = ticket_end(c, rc, B, c->st_img[slot], c->st_flow[slot], c->st_seg[slot], c->d_pose_ring[pr], snap, c->h_pose_ring[pr]);
        has_ticket = c->tickets.size() > before;            // (the weight guard's float32 batches get no ticket)
    } else if (rc == DAVO_OK && c->f32_fallback) ++c->n_f32_batches;
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->h_pose_ring[pr], c->d_pose_ring[pr], (size_t)B * 12 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipEventRecord(c->pose_done[pr], s));
    c->jobs.push_back(StreamJob{B, pose_out, pr, has_ticket, seq});
    c->copy_tracked[pr] = track_copy;
    ++c->n_submitted;
    if (track_copy && (unsigned long long)hold < c->n_submitted) {
        const int q = (int)((c->n_submitted - 1 - hold) % STREAM_POSES);
        if (c->copy_tracked[q]) HIP_TRY(c, hipEventSynchronize(c->st_copied[q]));
        // (a batch submitted with hold >= 8 recorded no event: it has been delivered by now if it is 8 or more submits back, and a
        // caller that lowers `hold` from one call to the next keeps the larger promise for the batches it made it for)
    }
    return DAVO_OK;
}

int davo_wait(davo_ctx* c, int leave_pending) {
    if (!c) return DAVO_ERR_INVALID;
    if (leave_pending < 0) return fail(c, DAVO_ERR_INVALID, "leave_pending must be >= 0");
    HIP_TRY(c, hipSetDevice(c->device));
    while ((int)c->jobs.size() > leave_pending) { const int rc = deliver_front(c); if (rc) return rc; }
    return DAVO_OK;
}

int davo_pending(davo_ctx* c) { return c ? (int)c->jobs.size() : DAVO_ERR_INVALID; }

int davo_forward(davo_ctx* c, int B, const uint8_t* img, const float* flow, const float* seg, float* pose_out) {
    if (!c) return DAVO_ERR_INVALID;
    if (!img || !flow || !seg || !pose_out) return fail(c, DAVO_ERR_INVALID, "null host pointer");
    if (B < 1 || B > c->max_batch) return fail(c, DAVO_ERR_INVALID, "batch %d outside [1,%d]", B, c->max_batch);
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = deliver_all(c); if (rc) return rc; }          // davo_submit batches still under way: delivered first
    { int rc = sync_all_slots(c); if (rc) return rc; }       // the host path owns the single staging buffer set
    activate_slot(c, 0);
    const size_t HW = (size_t)c->H * c->W;
    const size_t nb_img = HW * 9, nb_flow = HW * 8 * sizeof(float), nb_seg = HW * 3 * sizeof(float);
    if (!c->s_img) {
        HIP_TRY(c, hipMalloc(&c->s_img, nb_img * c->max_batch));
        HIP_TRY(c, hipMalloc(&c->s_flow, nb_flow * c->max_batch));
        HIP_TRY(c, hipMalloc(&c->s_seg, nb_seg * c->max_batch));
        HIP_TRY(c, hipMalloc(&c->s_pose, (size_t)c->max_batch * 12 * sizeof(float)));
    }
    if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    { int rc = judge_all(c); if (rc) return rc; }               // device-path batches issued before this call
    activate_slot(c, 0);
    c->d_range = c->d_range_base;
    // The base record holds RUNNING maxima like the ring's records (params.h): a record that starts at zero is raised by every wave of
    // every kernel's first round, and at batch 1 - the reference's operating point - those serialised atomics were 200 us of this
    // call's 417.  "Clamped" stays exact per call (the call that pushes a maximum past 65504 fails, and every recovery path zeroes
    // the record); "too small" is judged on what has been stored since the record was last zeroed: by a recovery, a change of scales,
    // and every FRESH_EVERY-th call.
    if (++c->host_since_fresh >= FRESH_EVERY) {
        c->host_since_fresh = 0;
        HIP_TRY(c, hipMemsetAsync(c->d_range_base, 0, RANGE_WORDS * sizeof(unsigned), c->stream));
    }
    // Sub-batches: the copy of chunk i+1 (copy_stream) overlaps the kernels of chunk i (compute stream).
    // Results do not depend on the split (windows are independent; tests/test_hip_parity.py batch invariance).
    // Only flow planes 0 and 1 are read by the path (davo.py:978-982), so only those cross PCIe.
    const int chunk = (c->host_chunk > 0 && B >= 2 * c->host_chunk) ? c->host_chunk : B;
    const int nchunks = (B + chunk - 1) / chunk;
    while ((int)c->copy_done.size() < nchunks) {
        hipEvent_t e;
        HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->copy_done.push_back(e);
    }
    bool f32_fallback = false;
    // Batch 1 is the reference's own operating point (run_inference.sh:44-51), and there this call was 474 us around 129 us of kernels
    // (round 5).  What went: the copy stream and its event for a call that is a single sub-batch (the copies go on the compute
    // stream itself); the read-back of the range record on a stream of its own behind the synchronise (the call's last kernel mirrors
    // the record into page-locked host memory like a ticketed batch's, prologue.h); the pose copy into pageable memory (a page-locked
    // bounce buffer, then memcpy).
    const bool h3_call = c->impl == 0 && c->precision == 1;
    if (h3_call) { int rc = ensure_ring(c, false); if (rc) return rc; }
    if (!c->h_sync_pose) HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_sync_pose), (size_t)c->max_batch * 12 * sizeof(float), hipHostMallocDefault));
    unsigned seq = 0;
    for (int i = 0; i < nchunks; ++i) {
        const int b0 = i * chunk, nb = std::min(chunk, B - b0);
        uint8_t* di = (uint8_t*)c->s_img + nb_img * b0;
        uint8_t* df = (uint8_t*)c->s_flow + nb_flow * b0;
        uint8_t* ds = (uint8_t*)c->s_seg + nb_seg * b0;
        hipStream_t cs = nchunks == 1 ? c->stream : c->copy_stream;
        HIP_TRY(c, hipMemcpyAsync(di, img + nb_img * b0, nb_img * nb, hipMemcpyHostToDevice, cs));
        if (nb == 1) HIP_TRY(c, hipMemcpyAsync(df, (const uint8_t*)flow + nb_flow * b0, nb_flow / 2, hipMemcpyHostToDevice, cs));
        else HIP_TRY(c, hipMemcpy2DAsync(df, nb_flow, (const uint8_t*)flow + nb_flow * b0, nb_flow, nb_flow / 2, nb, hipMemcpyHostToDevice, cs));
        HIP_TRY(c, hipMemcpyAsync(ds, (const uint8_t*)seg + nb_seg * b0, nb_seg * nb, hipMemcpyHostToDevice, cs));
        if (nchunks > 1) {
            HIP_TRY(c, hipEventRecord(c->copy_done[i], c->copy_stream));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_done[i], 0));
        }
        if (h3_call && i == nchunks - 1) {       // the call's last kernel mirrors the finished record (all sub-batches) to the host
            if (++c->batch_seq == 0) c->batch_seq = 1;
            seq = c->batch_seq;
            c->snap = SnapArgs{};
            c->snap.record = c->d_range_base; c->snap.host_mirror = c->h_range_dev; c->snap.seq = seq; c->snap.B = nb;
        }
        int rc = forward_device(c, nb, di, (const float*)df, (const float*)ds, (float*)c->s_pose + (size_t)b0 * 12);
        c->snap = SnapArgs{};
        if (rc) return rc;
        f32_fallback |= c->f32_fallback;
    }
    if (f32_fallback) ++c->n_f32_batches;                      // once per call, not per sub-batch
    HIP_TRY(c, hipMemcpyAsync(c->h_sync_pose, c->s_pose, (size_t)B * 12 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(pose_out, c->h_sync_pose, (size_t)B * 12 * sizeof(float));
    if (c->last_precision != 1) return DAVO_OK;
    unsigned raw[RANGE_WORDS];
    int rc = DAVO_OK;
    if (seq && __atomic_load_n(&c->h_range[RANGE_SEQ], __ATOMIC_ACQUIRE) == seq) memcpy(raw, c->h_range, sizeof raw);      // the stream is idle: the mirror is final
    else rc = read_record(c, c->d_range_base, raw);
    if (rc) return rc;
    rc = check_range(c, raw);
    if (rc == DAVO_OK) note_seen(c, raw, c->act_shift);
    if (rc == DAVO_ERR_RANGE && c->opt_auto_range) {
        // the staged copy of the batch is still in HBM: re-issue it whole (recalibrated, or on the float32 kernels)
        rc = recover_batch(c, [&] { Ticket t{}; t.B = B; t.img = c->s_img; t.flow = c->s_flow; t.seg = c->s_seg; t.pose = c->s_pose; t.ring = -1; t.stream = c->stream; t.issue = ~0ull; return t; }());
        if (rc == DAVO_OK) HIP_TRY(c, hipMemcpy(pose_out, c->s_pose, (size_t)B * 12 * sizeof(float), hipMemcpyDeviceToHost));
    }
    return rc;
}

int davo_range_stats(davo_ctx* c, long long* recalibrations, long long* f32_batches, long long* reissued) {
    if (!c) return DAVO_ERR_INVALID;
    if (recalibrations) *recalibrations = c->n_recalibrations;
    if (f32_batches) *f32_batches = c->n_f32_batches;
    if (reissued) *reissued = c->n_reissued;
    return DAVO_OK;
}

int davo_activation_range(davo_ctx* c, float* max_abs, int* shifts, int reset) {
    if (!c) return DAVO_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = judge_all(c); if (rc && rc != DAVO_ERR_RANGE) return rc; }      // every issued batch's record is in range_seen now
    for (int i = 0; i < 6; ++i) {
        if (max_abs) max_abs[i] = c->range_seen[i];
        if (shifts) shifts[i] = c->act_shift[i];
    }
    if (reset) for (int i = 0; i < 6; ++i) c->range_seen[i] = 0.f;
    return DAVO_OK;
}

const char* davo_range_report(const davo_ctx* c) { return c ? c->range_report.c_str() : ""; }

int davo_set_activation_shifts(davo_ctx* c, const int* shifts) {
    if (!c) return DAVO_ERR_INVALID;
    { int rc = judge_all(c); if (rc) return rc; }             // batches issued under the old scales get their verdict first
    { int rc = freeze_pending_and_reset_ring(c); if (rc) return rc; }      // maxima stored under the old scales say nothing about the new
    c->host_since_fresh = FRESH_EVERY;                                       // ... the host path's record included: its next call starts afresh
    for (int i = 0; i < 6; ++i) {
        const int s = shifts ? shifts[i] : 0;
        if (s < -60 || s > 60) return fail(c, DAVO_ERR_INVALID, "activation shift %d outside [-60,60]", s);
        c->act_shift[i] = s;
    }
    return DAVO_OK;
}

int davo_calibrate(davo_ctx* c, int B, const void* d_img, const void* d_flow, const void* d_seg, int* shifts_out) {
    if (!c) return DAVO_ERR_INVALID;
    if (B < 1 || B > c->max_batch) return fail(c, DAVO_ERR_INVALID, "batch %d outside [1,%d]", B, c->max_batch);
    if (!d_img || !d_flow || !d_seg) return fail(c, DAVO_ERR_INVALID, "null device pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = judge_all(c); if (rc) return rc; }             // batches issued under the old scales get their verdict first
    { int rc = freeze_pending_and_reset_ring(c); if (rc) return rc; }
    activate_slot(c, 0);
    float* d_pose = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&d_pose), (size_t)B * 12 * sizeof(float)));
    const int rc = calibrate_on(c, B, d_img, d_flow, d_seg, d_pose);
    (void)hipFree(d_pose);
    if (rc == DAVO_OK && shifts_out) for (int i = 0; i < 6; ++i) shifts_out[i] = c->act_shift[i];
    return rc;
}

const char* davo_last_error(const davo_ctx* c) { return c ? c->err.c_str() : "null context"; }

void davo_destroy(davo_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_all_slots(c);
    comm_release(c);
    for (auto& kv : c->weights) if (kv.second.dev) (void)hipFree(kv.second.dev);
    for (auto& L : c->L) {
        if (L.d_w) (void)hipFree(L.d_w);
        if (L.d_b) (void)hipFree(L.d_b);
        if (L.d_wh) (void)hipFree(L.d_wh);
        if (L.d_bh) (void)hipFree(L.d_bh);
    }
    for (auto e : c->copy_done) (void)hipEventDestroy(e);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (int k = 0; k < STREAM_SETS; ++k)
        for (void* q : {c->st_img[k], c->st_flow[k], c->st_seg[k]}) if (q) (void)hipFree(q);
    for (int k = 0; k < STREAM_POSES; ++k) {
        if (c->d_pose_ring[k]) (void)hipFree(c->d_pose_ring[k]);
        if (c->h_pose_ring[k]) (void)hipHostFree(c->h_pose_ring[k]);
        if (c->pose_done[k]) (void)hipEventDestroy(c->pose_done[k]);
        if (c->st_copied[k]) (void)hipEventDestroy(c->st_copied[k]);
    }
    if (c->h_sync_pose) (void)hipHostFree(c->h_sync_pose);
    void* misc[] = {c->d_reissue_pose, c->d_range_base, c->d_splitk, c->d_pose_tiles, c->d_w1patch, c->d_w2patch, c->d_w3patch, c->d_w1patch_f32, c->d_w2patch_f32, c->d_w3patch_f32, c->d_zeros, c->d_wpred, c->d_bpred, c->s_img, c->s_flow, c->s_seg, c->s_pose};
    for (auto p : misc) if (p) (void)hipFree(p);
    for (auto& kv : c->tile_orders) if (kv.second) (void)hipFree(kv.second);
    for (auto& pe : c->prof_entries)
        for (auto& ab : pe.pending) { (void)hipEventDestroy(ab.first); (void)hipEventDestroy(ab.second); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto& sl : c->slots) free_slot(sl);
    delete c;
}

int davo_host_alloc(int device, size_t bytes, void** out) {
    if (!out || bytes == 0) return DAVO_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return DAVO_ERR_HIP;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? DAVO_OK : DAVO_ERR_HIP;
}
int davo_host_free(void* p) { return hipHostFree(p) == hipSuccess ? DAVO_OK : DAVO_ERR_HIP; }
int davo_host_register(int device, void* p, size_t bytes) {
    if (!p || bytes == 0) return DAVO_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return DAVO_ERR_HIP;
    return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? DAVO_OK : DAVO_ERR_HIP;
}
int davo_host_unregister(void* p) { return hipHostUnregister(p) == hipSuccess ? DAVO_OK : DAVO_ERR_HIP; }

int davo_device_malloc(davo_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return DAVO_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMalloc(out, bytes));
    return DAVO_OK;
}
int davo_device_free(davo_ctx* c, void* p) {
    if (!c) return DAVO_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipFree(p));
    return DAVO_OK;
}
int davo_memcpy_h2d(davo_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return DAVO_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DAVO_OK;
}
int davo_memcpy_d2h(davo_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return DAVO_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DAVO_OK;
}
int davo_synchronize(davo_ctx* c) {
    if (!c) return DAVO_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    // f16x3: the batches davo_forward_device issued since the last synchronize are judged here (the asynchronous
    // entry point cannot know its own result).  A failed verdict re-issues that batch (recover_batch); with "auto_range" 0
    // it is returned as DAVO_ERR_RANGE = some layer left the fp16-pair storage range.  davo_submit batches are delivered first.
    { int rc = deliver_all(c); if (rc) return rc; }
    return judge_all(c);
}
int davo_set_stream(davo_ctx* c, void* hip_stream) {
    if (!c) return DAVO_ERR_INVALID;
    if (hip_stream && c->inflight > 1) return fail(c, DAVO_ERR_INVALID, "a caller-owned stream needs davo_set_inflight(ctx, 1)");
    // batches issued on the stream the context is about to leave get their verdict (and any re-issue) while that stream is still
    // the one the context synchronises; the caller may destroy it afterwards
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = deliver_all(c); if (rc) return rc; }
    { int rc = judge_all(c); if (rc) return rc; }
    c->user_stream = hip_stream != nullptr;
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return DAVO_OK;
}

int davo_profile_enable(davo_ctx* c, int on) {
    if (!c) return DAVO_ERR_INVALID;
    if (!on && c->prof) { int rc = prof_collect(c); if (rc) return rc; }
    c->prof = on != 0;
    c->prof_dominant_only = on == 2;
    return DAVO_OK;
}
int davo_profile_reset(davo_ctx* c) {
    if (!c) return DAVO_ERR_INVALID;
    int rc = prof_collect(c);
    if (rc) return rc;
    for (auto& pe : c->prof_entries) { pe.launches = 0; pe.total_ms = 0.0; pe.dur_ms.clear(); pe.period_ms.clear(); }
    return DAVO_OK;
}
int davo_profile_samples(davo_ctx* c, const char* name, int which, float* out, int cap) {
    if (!c || !name || which < 0 || which > 1 || cap < 0 || (cap > 0 && !out)) return DAVO_ERR_INVALID;
    int rc = prof_collect(c);
    if (rc) return rc;
    for (const auto& pe : c->prof_entries)
        if (pe.name == name) {
            const std::vector<float>& v = which == 0 ? pe.dur_ms : pe.period_ms;
            const int n = (int)std::min(v.size(), (size_t)cap);
            for (int i = 0; i < n; ++i) out[i] = v[i];
            return (int)v.size();
        }
    return 0;
}
int davo_profile_entry(davo_ctx* c, int i, char* name, int name_len, int* launches, double* total_ms) {
    if (!c) return DAVO_ERR_INVALID;
    int rc = prof_collect(c);
    if (rc) return rc;
    if (i < 0 || i >= (int)c->prof_entries.size()) return DAVO_ERR_INVALID;
    const ProfEntry& pe = c->prof_entries[i];
    if (name && name_len > 0) { strncpy(name, pe.name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    if (launches) *launches = pe.launches;
    if (total_ms) *total_ms = pe.total_ms;
    return DAVO_OK;
}

int davo_last_plan(davo_ctx* c, int layer, int launch, int* mtiles, int* bn) {
    if (!c || layer < 0 || layer > 6 || launch < 0 || launch > 1) return DAVO_ERR_INVALID;
    const int v = c->last_plan[layer][launch];
    if (mtiles) *mtiles = v / 1000;
    if (bn) *bn = v % 1000;
    return DAVO_OK;
}

// (Re)create the slots' streams.  With cu_partition on, slot i of n gets the CUs [i*32/n, (i+1)*32/n) of EVERY XCD
// (hipExtStreamCreateWithCUMask; mask bit b = CU b/8 of XCD b%8, and a mask must leave no XCD empty — probed with
// tools/exp/cumask_probe.hip), so batches in different slots run side by side on disjoint CUs and the write bursts of
// one overlap the matrix phases of the other.  The launch planner then sizes rounds for 256/n CUs.
static int rebuild_slot_streams(davo_ctx* c, int n) {
    for (int i = 0; i < (int)c->slots.size(); ++i) {
        Slot& s = c->slots[i];
        if (s.stream) { HIP_TRY(c, hipStreamDestroy(s.stream)); s.stream = nullptr; }
        if (c->cu_partition && n > 1 && i < n && c->dev_cus == 256) {          // the mask layout below is the 8 XCD x 32 CU part's
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int lo = i * 32 / n, hi = (i + 1) * 32 / n;                 // CU indices inside an XCD
            for (int b = 0; b < 256; ++b)
                if (b / 8 >= lo && b / 8 < hi) mask[b / 32] |= 1u << (b % 32);
            HIP_TRY(c, hipExtStreamCreateWithCUMask(&s.stream, 8, mask));
        } else {
            HIP_TRY(c, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        }
    }
    c->own_stream = c->slots[0].stream;
    c->ncu = (c->cu_partition && n > 1 && c->dev_cus == 256) ? 256 / n : c->dev_cus;
    activate_slot(c, 0);
    return DAVO_OK;
}

int davo_set_inflight(davo_ctx* c, int n) {
    if (!c || n < 1 || n > 4) return fail(c, DAVO_ERR_INVALID, "inflight must be 1..4");
    if (n > 1 && c->user_stream) return fail(c, DAVO_ERR_INVALID, "in-flight slots use the context's own streams: clear davo_set_stream first");
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = deliver_all(c); if (rc) return rc; }
    { int rc = sync_all_slots(c); if (rc) return rc; }
    while ((int)c->slots.size() < n) {
        c->slots.emplace_back();
        int rc = alloc_slot(c, &c->slots.back());
        if (rc) return rc;
    }
    c->inflight = n;
    c->next_slot = 0;
    if (c->cu_partition || c->ncu != c->dev_cus) { int rc = rebuild_slot_streams(c, n); if (rc) return rc; }
    return DAVO_OK;
}

int davo_set_option(davo_ctx* c, const char* key, int value) {
    if (!c || !key) return DAVO_ERR_INVALID;
    const std::string k = key;
    if (k == "fuse_pose") c->opt_fuse_pose = value != 0;
    else if (k == "fuse_pack") c->opt_fuse_pack = value < 0 ? -1 : (value != 0);
    else if (k == "share_taps") c->opt_share_taps = value != 0;
    else if (k == "merge_rem") c->opt_merge_rem = value != 0;
    else if (k == "merge_cnv4") c->opt_merge_cnv4 = value != 0;
    else if (k == "tile_208x128") c->opt_tile_208x128 = value != 0;
    else if (k == "merge_order") c->opt_merge_order = value < 0 ? -1 : (value > 2 ? 0 : value);
    else if (k == "skip_order") c->opt_skip_order = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (k == "patch_cnv2") c->opt_patch_cnv2 = value != 0;
    else if (k == "patch_cnv3") c->opt_patch_cnv3 = value != 0;
    else if (k == "fold_tails") c->opt_fold_tails = value < 0 ? -1 : (value > 2 ? 1 : value);
    else if (k == "deep_ring") c->opt_deep_ring = value != 0;
    else if (k == "wave128") c->opt_wave128 = value < 0 ? 0 : (value > 3 ? 3 : value);      // 3: cnv4 on conv_igemm_h3w128 whatever the round count (test hook)
    else if (k == "split_k") c->opt_split_k = value != 0;
    else if (k == "fold_fixup") c->opt_fold_fixup = value != 0;
    else if (k == "f32_n16") c->opt_f32_n16 = value != 0;
    else if (k == "f32_n256") c->opt_f32_n256 = value != 0;
    else if (k == "merge_rem_f32") c->opt_merge_rem_f32 = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (k == "patch_f32") c->opt_patch_f32 = value != 0;
    else if (k == "auto_range") { int rc = judge_all(c); if (rc) return rc; c->opt_auto_range = value != 0; }
    else if (k == "stable_inputs") { int rc = judge_all(c); if (rc) return rc; c->opt_stable_inputs = value != 0; }
    else if (k == "force_tile") {
        // test hook: every f16x3 layer the tile fits runs as ONE launch of that tile shape (plan.h tile ids; -1 = planner)
        if (value < -1 || value >= NUM_TILES) return fail(c, DAVO_ERR_INVALID, "force_tile must be -1..%d", NUM_TILES - 1);
        for (auto& L : c->L) L.tile_h = (value >= 0 && L.cout >= tile_shape(value).bn) ? value : -1;
    }
    else if (k == "cu_partition") {
        if (c->user_stream) return fail(c, DAVO_ERR_INVALID, "cu_partition uses the context's own streams: clear davo_set_stream first");
        HIP_TRY(c, hipSetDevice(c->device));
        { int rc = sync_all_slots(c); if (rc) return rc; }
        c->cu_partition = value != 0;
        int rc = rebuild_slot_streams(c, c->inflight);
        if (rc) return rc;
    }
    else if (k == "profile_stride") { if (value < 1) return fail(c, DAVO_ERR_INVALID, "profile_stride must be >= 1"); c->prof_stride = value; c->prof_tick = 0; }
    else if (k == "host_chunk") { if (value < 0) return fail(c, DAVO_ERR_INVALID, "host_chunk must be >= 0"); c->host_chunk = value; }
    else return fail(c, DAVO_ERR_INVALID, "unknown option `%s'", key);
    return DAVO_OK;
}

int davo_set_precision(davo_ctx* c, int precision) {
    if (!c || (precision != 0 && precision != 1)) return fail(c, DAVO_ERR_INVALID, "precision must be 0 (f32) or 1 (f16x3)");
    c->precision = precision;
    return DAVO_OK;
}

int davo_set_impl(davo_ctx* c, int impl) {
    if (!c || (impl != 0 && impl != 1)) return fail(c, DAVO_ERR_INVALID, "impl must be 0 (mfma) or 1 (direct)");
    c->impl = impl;
    return DAVO_OK;
}

int davo_debug_read(davo_ctx* c, const char* tensor, float* host_out, size_t n_floats) {
    if (!c || !tensor || !host_out) return DAVO_ERR_INVALID;
    if (c->last_B < 1) return fail(c, DAVO_ERR_NOT_READY, "no forward has run yet");
    const size_t NB = 2 * (size_t)c->last_B;
    const float* src = nullptr;
    size_t n = 0;
    const std::string t = tensor;
    if (t == "att_table") { src = c->d_tab; n = (size_t)c->last_B * 3 * NCLS; }
    else if (t == "packed") {
        if (!c->packed_valid) {          // fused path: materialise the packed tensor on demand from the last inputs
            HIP_TRY(c, launch_mask_pack(16, static_cast<const uint8_t*>(c->last_img), static_cast<const float*>(c->last_flow),
                                        static_cast<const float*>(c->last_seg), c->d_tab, c->v, c->last_B, c->H, c->W, c->d_packed, c->stream));
            c->packed_valid = true;
        }
        src = c->d_packed; n = NB * c->H * c->W * c->packed_ld;
    }
    else {
        const char* names[7] = {"cnv1", "cnv2", "cnv3", "cnv4", "cnv5", "cnv6", "cnv7"};
        for (int i = 0; i < 7; ++i)
            if (t == names[i]) { src = c->d_act[i]; n = NB * c->act_floats_per_img[i]; }
    }
    if (t == "pose_tiles") {          // the fused pose head's per-tile partial sums of the last batch (slot 0's region)
        if (c->cnv7_valid || !c->d_pose_tiles) return fail(c, DAVO_ERR_NOT_READY, "the pose head did not run fused");
#ifndef DAVO_POSE_DEBUG
        if (n_floats > c->pose_tiles_floats) return fail(c, DAVO_ERR_INVALID, "pose_tiles holds %zu floats", c->pose_tiles_floats);
#endif
        return davo_memcpy_d2h(c, host_out, c->d_pose_tiles, n_floats * sizeof(float));
    }
    if (t == "cnv7" && !c->cnv7_valid)
        return fail(c, DAVO_ERR_NOT_READY, "cnv7 was not materialised: the pose head ran fused (davo_set_option(ctx, \"fuse_pose\", 0))");
    if (!src) return fail(c, DAVO_ERR_INVALID, "unknown tensor `%s'", tensor);
    if (n != n_floats) return fail(c, DAVO_ERR_INVALID, "`%s' holds %zu floats, caller asked for %zu", tensor, n, n_floats);
    int rc = davo_memcpy_d2h(c, host_out, src, n * sizeof(float));
    if (rc) return rc;
    const bool split = c->last_precision == 1 && t != "att_table" && t != "cnv7";
    if (split) {       // split-fp16 blocked -> plain float32 NHWC
        int ch = 8;
        const char* names[6] = {"cnv1", "cnv2", "cnv3", "cnv4", "cnv5", "cnv6"};
        for (int i = 0; i < 6; ++i) if (t == names[i]) ch = c->act_ch[i];
        const int cb = ch < 32 ? ch : 32;
        int shift = 0;
        for (int i = 0; i < 6; ++i) if (t == names[i]) shift = c->act_shift[i];
        std::vector<float> tmp(ch);
        const size_t npix = n / ch;
        for (size_t px = 0; px < npix; ++px) {
            const _Float16* raw = reinterpret_cast<const _Float16*>(host_out + px * ch);
            for (int k = 0; k < ch; ++k) {
                const _Float16* blk = raw + (size_t)(k / cb) * cb * 2;
                tmp[k] = ldexpf((float)blk[k % cb] + (float)blk[cb + k % cb], -shift);
            }
            memcpy(host_out + px * ch, tmp.data(), ch * sizeof(float));
        }
    }
    return DAVO_OK;
}

int davo_tile_filter_rows(int m0, int m1, int Hout, int Wout, int Hin, int stride, int pad_t, int rate,
                          int* ky0, int* nky, int nblocks, int* chunk_map) {
    if (!ky0 || !nky || Hout < 1 || Wout < 1 || Hin < 1 || stride < 1 || rate < 1 || m0 < 0 || nblocks < 0) return DAVO_ERR_INVALID;
    const FilterRows fr = valid_filter_rows(m0, m1, Hout, Wout, Hin, stride, pad_t, rate);
    *ky0 = fr.ky0;
    *nky = fr.nky;
    if (chunk_map)
        for (int v = 0; v < 3 * fr.nky * nblocks; ++v) chunk_map[v] = h3_real_chunk(v, fr.ky0, fr.nky);
    return DAVO_OK;
}

int davo_plan_layer(int M, int npad, int groups, int* rows, int* tile_bm, int* tile_bn) {
    if (M < 1 || npad < 32 || npad % 32 || groups < 1 || !rows || !tile_bm || !tile_bn) return DAVO_ERR_INVALID;
    const std::vector<LaunchH> plan = plan_layer_h3(M, npad, groups, -1, npad == 256);
    if (plan.empty() || plan.size() > 2) return DAVO_ERR_INVALID;
    for (size_t i = 0; i < plan.size(); ++i) {
        const TileShape ts = tile_shape(plan[i].tile);
        rows[i] = plan[i].rows; tile_bm[i] = ts.bm; tile_bn[i] = ts.bn;
    }
    return (int)plan.size();
}

int davo_conv2d_same(int device, const float* x, int N, int H, int W, int Cin, const float* w, int k, int Cout,
                     const float* bias, int stride, int rate, int relu, int precision, float* y, char* err, int err_len) {
    auto bad = [&](const char* m, int code) {
        if (err && err_len > 0) { strncpy(err, m, err_len - 1); err[err_len - 1] = 0; }
        return code;
    };
    const int cl = ilog2_exact(Cin);
    if (!x || !w || !bias || !y) return bad("null pointer", DAVO_ERR_INVALID);
    if (cl < 2) return bad("Cin must be a power of two >= 4", DAVO_ERR_INVALID);
    if (precision == 1 && cl < 3) return bad("f16x3 needs Cin >= 8", DAVO_ERR_INVALID);
    if (!(k == 1 || k == 3 || k == 5 || k == 7) || !(stride == 1 || stride == 2) || rate < 1)
        return bad("k in {1,3,5,7}, stride in {1,2}, rate >= 1", DAVO_ERR_INVALID);
    if (hipSetDevice(device) != hipSuccess) return bad("hipSetDevice failed", DAVO_ERR_HIP);
    ConvLayer L;
    init_layer(L, "conv", k, stride, rate, Cin, Cout, 1);
    int Ho, Wo, pt, pl;
    same_pad(H, k, stride, rate, &Ho, &pt);
    same_pad(W, k, stride, rate, &Wo, &pl);
    const size_t nx = (size_t)N * H * W * Cin, ny = (size_t)N * Ho * Wo * Cout;
    std::vector<float> bp(precision == 1 ? L.npad_h : L.npad, 0.f);
    memcpy(bp.data(), bias, Cout * sizeof(float));
    std::vector<float> wp;
    std::vector<_Float16> wph, xh;
    const void *hx = x, *hw = nullptr;
    size_t wbytes = 0;
    if (precision == 1) {
        wph.assign((size_t)L.npad_h * L.nchunks_h * 64, (_Float16)0.0f);
        L.wscale = weight_prescale(w, (size_t)k * k * Cin * Cout);
        pack_conv_weights_h3(w, k, Cin, Cout, nullptr, Cin, L.cb_log2, L.tpc_log2, L.cpb, L.nchunks_h, L.wscale, wph.data());
        hw = wph.data(); wbytes = wph.size() * sizeof(_Float16);
        const int cb = 1 << L.cb_log2;                       // float32 NHWC -> split-fp16 blocked
        xh.resize(nx * 2);
        for (size_t px = 0; px < nx / Cin; ++px)
            for (int ch = 0; ch < Cin; ++ch) {
                _Float16* blk = xh.data() + px * Cin * 2 + (size_t)(ch / cb) * cb * 2;
                split_f16(x[px * Cin + ch], blk + ch % cb, blk + cb + ch % cb);
            }
        hx = xh.data();
    } else {
        wp.assign((size_t)L.npad * L.kpad, 0.f);
        pack_conv_weights(w, k, Cin, Cout, nullptr, Cin, L.npad, L.kpad, wp.data());
        hw = wp.data(); wbytes = wp.size() * sizeof(float);
    }
    void *dx = nullptr, *dw = nullptr, *db = nullptr, *dy = nullptr, *dz = nullptr;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    chk(hipMalloc(&dx, nx * 4)); chk(hipMalloc(&dw, wbytes));
    chk(hipMalloc(&db, bp.size() * 4)); chk(hipMalloc(&dy, ny * 4));
    chk(hipMalloc(&dz, 256));
    if (e == hipSuccess) {
        chk(hipMemset(dz, 0, 256));
        chk(hipMemcpy(dx, hx, nx * 4, hipMemcpyHostToDevice));
        chk(hipMemcpy(dw, hw, wbytes, hipMemcpyHostToDevice));
        chk(hipMemcpy(db, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
        if (precision == 1) {
            const int tile = Cout <= 32 ? TILE_128x32 : Cout <= 64 ? TILE_256x64 : Cout <= 128 ? TILE_256x128 : TILE_128x256;
            const TileShape ts = tile_shape(tile);
            ConvParamsH p{};
            p.x = static_cast<const uint8_t*>(dx); p.w = static_cast<const uint8_t*>(dw);
            p.bias = static_cast<const float*>(db); p.y = static_cast<uint8_t*>(dy);
            p.zeros = static_cast<const uint8_t*>(dz);
            p.Hin = H; p.Win = W; p.Hout = Ho; p.Wout = Wo; p.x_pix_bytes = (long)Cin * 4; p.x_pix_log2 = -1;
            p.cb_log2 = L.cb_log2; p.tpc_log2 = L.tpc_log2; p.cpb = L.cpb; p.nchunks = L.nchunks_h;
            p.w_row_bytes = (long)L.nchunks_h * 128; p.y_mode = 0; p.y_ld = Cout; p.Cout = Cout;
            p.pad_t = pt; p.pad_l = pl; p.rate = rate; p.M = N * Ho * Wo; p.ntaps = k * k;
            p.Mtot = p.M; p.xs = 0; p.ntiles_n = L.npad_h / ts.bn; p.relu = relu; p.out_scale = 1.0f / L.wscale; p.bias_scale = L.wscale; p.range = nullptr;
            dim3 grid((p.M + ts.bm - 1) / ts.bm * p.ntiles_n, 1);
            const hipError_t le = launch_h3_generic(k, stride, tile, p, grid, nullptr);
            chk(le);
        } else {
            ConvParams p{};
            p.x = static_cast<const float*>(dx); p.w = static_cast<const float*>(dw);
            p.bias = static_cast<const float*>(db); p.y = static_cast<float*>(dy);
            p.zeros = static_cast<const float*>(dz);
            p.Hin = H; p.Win = W; p.Hout = Ho; p.Wout = Wo; p.cin_log2 = cl; p.x_ld = Cin; p.y_ld = Cout;
            p.Cout = Cout; p.pad_t = pt; p.pad_l = pl; p.rate = rate; p.M = N * Ho * Wo;
            p.nchunks = L.nchunks; p.Kpad = L.kpad; p.ntaps = k * k; p.ntiles_n = L.npad / L.BN; p.relu = relu;
            dim3 grid((p.M + BM - 1) / BM * p.ntiles_n, 1);
            chk(launch_conv(k, stride, L.BN, p, grid, nullptr));
        }
        chk(hipDeviceSynchronize());
        chk(hipMemcpy(y, dy, ny * 4, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dz); (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(db); (void)hipFree(dy);
    if (e != hipSuccess) return bad(hipGetErrorString(e), DAVO_ERR_HIP);
    return DAVO_OK;
}

}  // extern "C"
