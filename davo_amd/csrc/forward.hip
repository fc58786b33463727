// forward.hip — the forward plan of the pose path on one context:
//
//   se_squeeze_partial -> se_excite -> mask_pack -> cnv1..cnv5 -> cnv6 (rotation|translation
//   fused into one N = 2*cnv6_out GEMM, both read cnv5: nets/posenn.py:222-238)
//   -> cnv7 (grouped x2) -> pose head.
//
// The two PoseNN calls of a triplet (davo.py:1456-1457, shared weights) run as one batch of
// 2B pair images.  Host code only: kernels are reached through launch.h.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "ctx.h"
#include "launch.h"
#include "plan.h"

namespace davo {

// ---- profiling ------------------------------------------------------------------------------------
ProfScope::ProfScope(davo_ctx* ctx, const char* name) : c(ctx) {
    if (!c->prof) return;
    if (c->prof_dominant_only) {
        if (strcmp(name, "cnv6") != 0) return;
        if (c->prof_tick++ % c->prof_stride != 0) return;
    }
    for (auto& pe : c->prof_entries)
        if (pe.name == name) { e = &pe; break; }
    if (!e) {
        c->prof_entries.emplace_back();
        e = &c->prof_entries.back();
        e->name = name;
    }
    auto get = [&]() {
        hipEvent_t ev = nullptr;
        if (!c->event_pool.empty()) { ev = c->event_pool.back(); c->event_pool.pop_back(); }
        else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr;
        return ev;
    };
    a = get(); b = get();
    if (a) (void)hipEventRecord(a, c->stream);
}

ProfScope::~ProfScope() {
    if (!e) return;
    if (b) (void)hipEventRecord(b, c->stream);
    if (a && b) e->pending.emplace_back(a, b);
    else {                                    // half a pair is of no use: back to the pool
        if (a) c->event_pool.push_back(a);
        if (b) c->event_pool.push_back(b);
    }
}

int sync_all_slots(davo_ctx* c) {
    for (auto& s : c->slots) HIP_TRY(c, hipStreamSynchronize(s.stream));
    if (c->user_stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DAVO_OK;
}

int prof_collect(davo_ctx* c) {
    { int rc = sync_all_slots(c); if (rc) return rc; }
    for (auto& pe : c->prof_entries) {
        hipEvent_t prev_start = nullptr;
        for (auto& ab : pe.pending) {
            float ms = 0.f, gap = -1.f;
            if (hipEventElapsedTime(&ms, ab.first, ab.second) == hipSuccess) {
                pe.total_ms += ms;
                pe.launches += 1;
                if (prev_start && hipEventElapsedTime(&gap, prev_start, ab.first) != hipSuccess) gap = -1.f;
                if (pe.dur_ms.size() < PROF_SAMPLES_CAP) { pe.dur_ms.push_back(ms); pe.period_ms.push_back(gap); }
            }
            if (prev_start) c->event_pool.push_back(prev_start);
            prev_start = ab.first;
            c->event_pool.push_back(ab.second);
        }
        if (prev_start) c->event_pool.push_back(prev_start);
        pe.pending.clear();
    }
    return DAVO_OK;
}

// bind a slot's stream and workspace to the members every launch helper uses
void activate_slot(davo_ctx* c, int i) {
    const Slot& s = c->slots[i];
    if (!(c->user_stream && i == 0)) c->stream = s.stream;
    c->d_partial = s.d_partial; c->d_tab = s.d_tab; c->d_packed = s.d_packed; c->d_pose_partial = s.d_pose_partial;
    c->d_counters = s.d_counters;
    for (int k = 0; k < 7; ++k) c->d_act[k] = s.d_act[k];
}

// f16x3 only.  Stored activations (fp16 hi/lo pairs) are float32-grade while the layer's largest stored value is
// below the fp16 maximum (above it values were clamped) and not so small that the pairs lose their low bits
// (tools/exp_activation_scale.py: the 1e-4 bar holds down to ~2^-16 of O(1) activations; 2^-11 is the guard).
int check_range(davo_ctx* c, const unsigned* raw, const int* shifts) {
    static const char* names[6] = {"cnv1", "cnv2", "cnv3", "cnv4", "cnv5", "cnv6"};
    if (!shifts) shifts = c->act_shift;              // the scales the judged batch was issued under
    for (int i = 0; i < 6; ++i) {
        float v;
        memcpy(&v, &raw[i], sizeof v);
        if (!range_value_fails(v)) continue;         // params.h: the test the batch's last kernel applies too
        const float actual = ldexpf(v, -shifts[i]);
        if (!(v < 65504.f))
            return fail(c, DAVO_ERR_RANGE, "%s activations reach %.4g: outside the fp16-pair storage range at scale 2^%d "
                        "(values were clamped) - run davo_calibrate() or davo_set_precision(ctx, 0)", names[i], (double)actual, shifts[i]);
        return fail(c, DAVO_ERR_RANGE, "%s activations are at most %.4g: too small for the fp16-pair storage at scale 2^%d "
                        "- run davo_calibrate() or davo_set_precision(ctx, 0)", names[i], (double)actual, shifts[i]);
    }
    return DAVO_OK;
}

namespace {

constexpr int MAX_WEIGHT_CHANNEL_SPREAD_LOG2 = 14;     // f16x3 per-channel guard (weights.hip)
constexpr int FOLD_EXCITE_MAX_BATCH = 2;       // auto modes: largest batch that folds the excitation / fuses mask + pack into cnv1
constexpr int FUSE_PACK_MAX_BATCH = 0;        // measured level at every batch (cnv1 +5 us for mask_pack's 6.7): nowhere by default

// ---- long tiles first ------------------------------------------------------------------------------
// The 3x3 kernels skip the chunks of filter rows that are all padding for a tile (params.h, valid_filter_rows), so the tiles
// of one launch differ in length (dilation 8 on a 32-row map: 6 of an image's 13 tiles of 256 pixels walk two thirds of the
// chunks).  Workgroups are handed out strictly in id order, an XCD's to its four shader engines round-robin
// (tools/exp/dispatch_probe.hip), and a launch is three to six rounds of tiles: in natural order some CUs draw long tiles
// every round and the launch is as long as before.  The table puts, inside every XCD's contiguous run of tiles (xcd_remap),
// the long tiles first (stable, so neighbours stay neighbours): every engine then sees the same long-first sequence and
// the short tiles pack the end.  Returns nullptr when all tiles of the launch cost the same.
const int* tile_order_for(davo_ctx* c, int li, int kind, int bm, int mtile0, int mtiles, int ntiles_n, int M,
                          int Hout, int Wout, int Hin, int stride, int pad_t, int rate) {
    if (!c->opt_skip_order) return nullptr;
    const std::vector<int> key = {li, kind, bm, mtile0, mtiles, ntiles_n, M, Hout, Wout};
    auto it = c->tile_orders.find(key);
    if (it != c->tile_orders.end()) return it->second;
    const int nt = mtiles * ntiles_n;
    std::vector<int> cost(nt), order(nt);
    bool uniform = true;
    for (int t = 0; t < nt; ++t) {
        const int m0 = (mtile0 + t / ntiles_n) * bm, m1 = std::min(m0 + bm, M) - 1;
        cost[t] = valid_filter_rows(m0, m1, Hout, Wout, Hin, stride, pad_t, rate).nky;
        uniform = uniform && cost[t] == cost[0];
    }
    int* dev = nullptr;
    if (!uniform) {
        const int q = nt >> 3, r = nt & 7;
        for (int x = 0; x < 8; ++x) {                       // xcd_remap: XCD x runs tiles [start, start + len)
            const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, len = q + (x < r ? 1 : 0);
            int w = start;
            for (int want = 3; want >= 1; --want)
                for (int t = start; t < start + len; ++t)
                    if (cost[t] == want) order[w++] = t;
        }
        if (hipMalloc(reinterpret_cast<void**>(&dev), nt * sizeof(int)) != hipSuccess ||
            hipMemcpy(dev, order.data(), nt * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            if (dev) (void)hipFree(dev);
            dev = nullptr;                                   // not fatal: natural order
            (void)hipGetLastError();
        }
    }
    c->tile_orders[key] = dev;
    return dev;
}

// ---- one conv layer, FP32-MFMA path ---------------------------------------------------------------
// fuse_pose (cnv7): the pose head runs in the epilogue (conv_igemm.h); *pose_mt receives the layer's M tiles
int run_conv_layer(davo_ctx* c, int li, const float* x, int x_ld, int Hin, int Win, float* y, int y_ld, int NB, bool fuse_pose = false, int* pose_mt = nullptr) {
    const ConvLayer& L = c->L[li];
    ConvParams p{};
    int Ho, Wo, pt, pl;
    same_pad(Hin, L.KS, L.stride, L.rate, &Ho, &pt);
    same_pad(Win, L.KS, L.stride, L.rate, &Wo, &pl);
    p.x = x; p.w = L.d_w; p.bias = L.d_b; p.y = y; p.zeros = c->d_zeros;
    p.Hin = Hin; p.Win = Win; p.Hout = Ho; p.Wout = Wo;
    p.cin_log2 = L.cin_log2; p.x_ld = x_ld; p.x_coff = 0; p.y_ld = y_ld; p.y_coff = 0;
    p.Cout = L.cout; p.pad_t = pt; p.pad_l = pl; p.rate = L.rate;
    p.M = NB * Ho * Wo; p.nchunks = L.nchunks; p.Kpad = L.kpad; p.ntaps = L.KS * L.KS;
    p.ntiles_n = L.npad / L.BN; p.relu = 1;
    if (L.groups == 2) {
        p.g_x_coff = L.cin; p.g_y_coff = L.cout;
        p.g_w = (long)L.npad * L.kpad; p.g_bias = L.npad;
    }
    const int mtiles = (p.M + BM - 1) / BM;
    std::vector<Launch> plan = plan_layer(mtiles, L.npad, L.groups, c->ncu);
    if (li == 0 && L.cout <= 16 && c->opt_f32_n16) plan = {{0, mtiles, 16}};      // cnv1 on the 128x16 tile at every batch size (conv_igemm.h, N16)
    if (fuse_pose) {
        const size_t need = (size_t)L.groups * mtiles * 8 * 6;
        if (need > c->pose_tiles_floats) {
            if (c->d_pose_tiles) { int rs = sync_all_slots(c); if (rs) return rs; HIP_TRY(c, hipFree(c->d_pose_tiles)); c->d_pose_tiles = nullptr; }
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_pose_tiles), need * sizeof(float) * 4));   // x4: one region per in-flight slot
            c->pose_tiles_floats = need;
        }
        const int slot_idx = (c->next_slot + c->inflight - 1) % c->inflight;
        p.pose_w = c->d_wpred; p.pose_partial = c->d_pose_tiles + (size_t)slot_idx * c->pose_tiles_floats;
        p.pose_P = Ho * Wo; p.pose_mt = mtiles;
        if (pose_mt) *pose_mt = mtiles;
    }
    c->last_plan[li][0] = c->last_plan[li][1] = 0;
    auto order_for = [&](const Launch& l, int ntn) {
        return (L.KS == 3 && L.cin_log2 >= 5) ? tile_order_for(c, li, 0, BM, l.mtile0, l.mtiles, ntn, p.M, Ho, Wo, Hin, L.stride, pt, L.rate) : nullptr;
    };
    // experiment ("f32_n256"): cnv5 / cnv6 as whole rounds of 128 x 256 tiles (eight waves, one workgroup per CU) + a remainder launch of
    // 128 x 64 tiles
    if (c->opt_f32_n256 && (li == 4 || li == 5) && L.npad == 256 && L.groups == 1 && !fuse_pose && mtiles >= c->ncu) {
        const int main_m = (mtiles / c->ncu) * c->ncu;
        ConvParams pm = p;
        pm.mtile0 = 0; pm.ntiles_n = 1;
        pm.tile_order = tile_order_for(c, li, 2, BM, 0, main_m, 1, p.M, Ho, Wo, Hin, L.stride, pt, L.rate);
        {
            ProfScope ps(c, L.label);
            HIP_TRY(c, launch_layer_n256(li, pm, dim3(main_m, 1), c->stream));
        }
        c->last_plan[li][0] = main_m * 1000 + 256;
        if (main_m < mtiles) {
            ConvParams pr = p;
            pr.mtile0 = main_m; pr.ntiles_n = L.npad / 64;
            pr.tile_order = tile_order_for(c, li, 0, BM, main_m, mtiles - main_m, pr.ntiles_n, p.M, Ho, Wo, Hin, L.stride, pt, L.rate);
            const std::string label = std::string(L.label) + ".rem";
            ProfScope ps(c, label.c_str());
            HIP_TRY(c, launch_layer(li, 64, pr, dim3((mtiles - main_m) * pr.ntiles_n, 1), c->stream));
            c->last_plan[li][1] = (mtiles - main_m) * 1000 + 64;
        }
        return DAVO_OK;
    }
    // main + remainder as one grid (conv_igemm.h, conv_igemm_f32_mainrem; "merge_rem_f32"): the remainder's tiles start on the CUs
    // that finish their last main tile first, instead of behind a launch boundary.  Same tiles, same arithmetic.
    // (cnv7 keeps its two launches: merged it measured 0.532 against 0.439 + 0.052 ms - its 32-column remainder tiles then run two per CU
    // on the main tile's LDS and register budget instead of three, and 0.545 with a 64-column remainder; cnv4 / cnv5 / cnv6: -5 / -4 / -1 %,
    // profiles/r05q_f32_mainrem_ab.md)
    if (c->opt_merge_rem_f32 && li >= 3 && (li <= 5 || c->opt_merge_rem_f32 > 1) && plan.size() == 2 && plan[0].BN == 128 && (plan[1].BN == 32 || plan[1].BN == 64)) {
        ConvParams pm = p, pr = p;
        pm.mtile0 = plan[0].mtile0; pm.ntiles_n = L.npad / 128; pm.tile_order = order_for(plan[0], pm.ntiles_n);
        pr.mtile0 = plan[1].mtile0; pr.ntiles_n = L.npad / plan[1].BN; pr.tile_order = order_for(plan[1], pr.ntiles_n);
        const int n_main = plan[0].mtiles * pm.ntiles_n, n_rem = plan[1].mtiles * pr.ntiles_n;
        if (n_main % 8 == 0 && (L.groups == 1 || n_rem % 8 == 0)) {
            ProfScope ps(c, L.label);
            HIP_TRY(c, launch_layer_mainrem(li, plan[1].BN, pm, pr, n_main, n_rem, L.groups, c->stream));
            c->last_plan[li][0] = (plan[0].mtiles + plan[1].mtiles) * 1000 + 128;      // one launch covers the layer
            return DAVO_OK;
        }
    }
    for (size_t i = 0; i < plan.size(); ++i) {
        p.mtile0 = plan[i].mtile0;
        p.ntiles_n = plan[i].BN == 16 ? 1 : L.npad / plan[i].BN;
        dim3 grid(plan[i].mtiles * p.ntiles_n, L.groups);
        p.tile_order = (L.KS == 3 && L.cin_log2 >= 5) ? tile_order_for(c, li, 0, BM, plan[i].mtile0, plan[i].mtiles, p.ntiles_n, p.M, Ho, Wo, Hin, L.stride, pt, L.rate) : nullptr;
        const std::string label = i == 0 ? std::string(L.label) : std::string(L.label) + ".rem";
        ProfScope ps(c, label.c_str());
        HIP_TRY(c, launch_layer(li, plan[i].BN, p, grid, c->stream));
        c->last_plan[li][i] = plan[i].mtiles * 1000 + plan[i].BN;
    }
    return DAVO_OK;
}

// ---- one conv layer, f16x3 path: x and y are split-fp16 blocked tensors (y float32 when y_f32) -----
int run_conv_layer_h3(davo_ctx* c, int li, const void* x, int x_ch, int Hin, int Win, void* y, int y_ld,
                      bool y_f32, int NB, bool fuse_pose = false, int* pose_bm = nullptr, int* pose_mt = nullptr,
                      int* pose_ntn = nullptr, float* pose_out = nullptr) {
    const ConvLayer& L = c->L[li];
    ConvParamsH p{};
    int Ho, Wo, pt, pl;
    same_pad(Hin, L.KS, L.stride, L.rate, &Ho, &pt);
    same_pad(Win, L.KS, L.stride, L.rate, &Wo, &pl);
    p.x = static_cast<const uint8_t*>(x); p.w = L.d_wh; p.bias = L.d_bh; p.y = static_cast<uint8_t*>(y);
    p.zeros = reinterpret_cast<const uint8_t*>(c->d_zeros);
    p.Hin = Hin; p.Win = Win; p.Hout = Ho; p.Wout = Wo;
    p.x_pix_bytes = (long)x_ch * 4; p.x_boff = 0; p.x_pix_log2 = ilog2_exact(x_ch * 4);
    p.cb_log2 = L.cb_log2; p.tpc_log2 = L.tpc_log2; p.cpb = L.cpb; p.nchunks = L.nchunks_h;
    p.w_row_bytes = (long)L.nchunks_h * 128;
    p.y_mode = y_f32 ? 0 : 1; p.y_ld = y_ld; p.y_coff = 0; p.Cout = L.cout;
    p.pad_t = pt; p.pad_l = pl; p.rate = L.rate;
    p.M = NB * Ho * Wo; p.ntaps = L.KS * L.KS; p.mtile0 = 0; p.relu = 1;
    p.Mtot = p.M; p.xs = c->opt_share_taps ? 1 : 0;
    {   // stored activations carry 2^act_shift (exact); cnv7 feeds the float32 pose head unscaled
        const int sin = li == 0 ? 0 : c->act_shift[li - 1], sout = li == 6 ? 0 : c->act_shift[li];
        p.out_scale = ldexpf(1.0f / L.wscale, sout - sin);
        p.bias_scale = ldexpf(L.wscale, sin);
        p.range = c->d_range ? c->d_range + li : nullptr;
    }
    if (L.groups == 2) {
        p.g_x_boff = L.cin * 4; p.g_y_coff = L.cout;
        p.g_w = (long)L.npad_h * p.w_row_bytes; p.g_bias = L.npad_h;
    }
    if (const char* e = tuning_env("DAVO_DBG")) p.dbg = atoi(e);       // tuning build only
    // cnv5, cnv6, cnv7 (256 channels per group) and cnv4 (128): conv_igemm_h3s.h is instantiated for them
    const bool allow_208 = (li >= 4 && L.npad_h == 256) || (li == 3 && L.npad_h == 128 && c->opt_tile_208x128);
    // cnv5 / cnv6 with "wave128": the 256x256 tile is conv_igemm_h3w's, 4.7 % faster than the one the planner's table was fitted on
    // (measured across 21 batch sizes with the table scaled: only B = 24 and 256x832 B = 6 change plan, -2.7 / -3.2 %: profiles/r05bn_planner_scale.log)
    const double others_scale = (c->opt_wave128 && (li == 4 || li == 5) && L.npad_h == 256 && L.groups == 1 && !fuse_pose) ? 0.955 : 1.0;
    std::vector<LaunchH> plan = plan_layer_h3(p.M, L.npad_h, L.groups, L.tile_h, allow_208, c->ncu, others_scale);
    if (fuse_pose) {      // one launch, one tile shape no taller than an image, so a tile touches <= 2 images
        const int P = Ho * Wo;
        const int best = plan_single_tile_h3(p.M, L.npad_h, L.groups, P, L.tile_h, allow_208, c->ncu);
        if (best < 0) return fail(c, DAVO_ERR_INVALID, "no tile fits the fused pose head");
        plan = {{0, p.M, best}};
        const TileShape ts = tile_shape(best);
        const int mt = (p.M + ts.bm - 1) / ts.bm, ntn = L.npad_h / ts.bn;
        const size_t need = (size_t)L.groups * mt * ntn * 6;
        if (need > c->pose_tiles_floats) {
            if (c->d_pose_tiles) { int rs = sync_all_slots(c); if (rs) return rs; HIP_TRY(c, hipFree(c->d_pose_tiles)); c->d_pose_tiles = nullptr; }
#ifdef DAVO_POSE_DEBUG
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_pose_tiles), need * sizeof(float) * 4 + (size_t)L.groups * mt * ntn * 512 * 20 * sizeof(float)));
#else
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_pose_tiles), need * sizeof(float) * 4));   // x4: one region per in-flight slot
#endif
            c->pose_tiles_floats = need;
        }
        const int slot_idx = (c->next_slot + c->inflight - 1) % c->inflight;      // the slot this batch runs in
        p.y_mode = 2; p.pose_w = c->d_wpred; p.pose_partial = c->d_pose_tiles + (size_t)slot_idx * c->pose_tiles_floats;
        p.pose_P = P; p.pose_mt = mt;
        if (pose_out) {      // the launch's last workgroup adds the tiles and writes the poses (pose_tail.h)
            p.pose_counter = c->d_counters; p.pose_bias = c->d_bpred; p.pose_out = pose_out;
            p.pose_NB = NB; p.pose_bm = ts.bm; p.pose_total = L.groups * mt * ntn;
        }
        if (pose_bm) *pose_bm = ts.bm;
        if (pose_mt) *pose_mt = mt;
        if (pose_ntn) *pose_ntn = ntn;
    }
    c->last_plan[li][0] = c->last_plan[li][1] = 0;
    // cnv4 on conv_igemm_h3w128's 256 x 128 tiles, every row in one launch - where the 256-row tiles fill whole rounds of the CUs or
    // nearly so: a round of them takes 24.4 us against 29.2 for the same rows on conv_igemm_h3's tiles (B = 128: 0.385 -> 0.318 ms),
    // which a last round that is three quarters empty gives back (B = 32, 3.25 rounds: 0.0925 -> 0.0938 ms)
    const long t256 = (p.M + 255) / 256;
    const bool rounds_ok = t256 >= c->ncu && (double)((t256 + c->ncu - 1) / c->ncu) * c->ncu <= 1.12 * (double)t256;
    if ((c->opt_wave128 >= 3 || (c->opt_wave128 >= 2 && rounds_ok)) && li == 3 && !fuse_pose && L.groups == 1 && L.npad_h == 128 && L.tile_h < 0 &&
        p.M >= 256 * c->ncu) {
        ConvParamsH pw = p;
        pw.ntiles_n = 1; pw.mtile0 = 0;
        if (layer_h3w128_supported(pw)) {
            {
                ProfScope ps(c, L.label);
                HIP_TRY(c, launch_layer_h3w128(pw, c->stream));
            }
            c->last_plan[li][0] = ((p.M + 127) / 128) * 1000 + TILE_256x128;
            return DAVO_OK;
        }
    }
    if (c->opt_merge_cnv4 && c->opt_merge_rem && li == 3 && !fuse_pose && L.npad_h == 128 && L.groups == 1 && c->ncu == 256 && L.tile_h < 0 &&
        plan.size() == 1 && plan[0].tile == TILE_128x128) {
        // cnv4 (N = 128): whole rounds of 256x128 tiles (shared-tap staging, twice the matrix work per staged byte of the
        // 128x128 tile) + the remaining rows on 128x128 tiles, as one grid like cnv5 / cnv6 below
        const int rows1 = (p.M / 256 / c->ncu) * c->ncu * 256;
        const int rem = p.M - rows1, n_rem = (rem + 127) / 128;
        if (rows1 > 0 && rem > 0 && rem % 128 == 0 && n_rem % 64 == 0 && n_rem <= 256)
            plan = {{0, rows1, TILE_256x128}, {rows1, rem, TILE_128x128}};
    }
    const bool merge_256 = plan.size() == 2 && plan[0].tile == TILE_256x256 && plan[1].tile == TILE_128x128 && L.npad_h == 256;
    const bool merge_128 = plan.size() == 2 && plan[0].tile == TILE_256x128 && plan[1].tile == TILE_128x128 && L.npad_h == 128;
    if (c->opt_merge_rem && !fuse_pose && (merge_256 || merge_128) && L.groups == 1 && c->ncu == 256 && !tuning_env("DAVO_NO_MERGE")) {
        // main + remainder as one grid (conv_igemm_h3_mainrem): same tiles, same arithmetic, de-phased store bursts
        ConvParamsH pm = p, pr = p;
        const int rem_ntn = L.npad_h / 128;                       // N tiles of a 128x128 remainder row block
        pm.ntiles_n = 1; pm.mtile0 = 0; pm.M = plan[0].rows;
        pr.ntiles_n = rem_ntn; pr.mtile0 = plan[1].row0 / 128; pr.M = plan[1].row0 + plan[1].rows;
        const int n_main = plan[0].rows / 256, n_rem = ((plan[1].rows + 127) / 128) * rem_ntn;
        // the shape test comes first: a profiling scope is opened only around a launch that is really issued
        // (an empty event pair under the layer's label would halve its average and advance the stride counter twice)
        // "wave128": the four-wave main tile (conv_igemm_h3w.h) runs as a launch of its own - merged with four-wave remainder tiles it
        // measured 0.262 / 0.497 ms for cnv5 / cnv6 against 0.258 / 0.487 as two launches (profiles/r05bc_w128_ab.log)
        const bool own_launch = c->opt_wave128 && merge_256 && layer_h3w_supported(li, pm);
        if (!own_launch && layer_h3_mainrem_supported(li, pm, n_main, n_rem)) {
            // f16x3: long-first measured 256.8 against 259.9 us on cnv5 with a third more HBM reads (205 -> 272 MB: neighbours no longer
            // run side by side) and no change of the power-capped step: natural order unless "skip_order" is 2
            pm.tile_order = c->opt_skip_order >= 2 ? tile_order_for(c, li, 1, 256, 0, n_main, 1, pm.M, Ho, Wo, Hin, L.stride, pt, L.rate) : nullptr;
            const int order = c->opt_merge_order >= 0 ? c->opt_merge_order : (pm.tile_order ? 2 : 0);
            {
                ProfScope ps(c, L.label);
                HIP_TRY(c, launch_layer_h3_mainrem(li, pm, n_main, pr, n_rem, order, c->stream));
            }
            c->last_plan[li][0] = ((plan[0].rows + plan[1].rows + 127) / 128) * 1000 + 7;     // 7: 256x256 + 128x128 in one grid
            return DAVO_OK;
        }
    }
    // Split-K.  A launch of at most half a workgroup per CU (cnv6 at batch 1: 104 tiles of 128x128, each a serial chain of 72
    // chunks at ~0.7 us) leaves half of the chip idle and is as long as its chain: the two halves of the input channels run
    // as two "groups" of the same kernel (the grouped-convolution path cnv7 uses: own channel offset, weight offset and
    // output slot; the second half starts from a zero bias) into float32 partial sums, and splitk_fixup adds the two,
    // applies ReLU and writes the stored form.  Two partial sums instead of one chain change the float32 rounding of the
    // layer's outputs (not their value: ~1e-7 relative), so batch sizes that split and batch sizes that do not agree to
    // rounding, not to the bit ("split_k" 0 restores the single chain).  (With S parts: S groups, S partial sums.)
    if (c->opt_split_k && !fuse_pose && (li == 4 || li == 5) && L.groups == 1 && plan.size() == 1 && p.y_mode == 1 && L.cout % 32 == 0 &&
        !is_208(plan[0].tile)) {
        const TileShape ts = tile_shape(plan[0].tile);
        const int mtiles = (p.M + ts.bm - 1) / ts.bm, ntn = L.npad_h / ts.bn;
        const long tiles = (long)mtiles * ntn;
        // parts: whole channel blocks each, whole filter rows (shared-tap staging walks a row's three taps together).  Four parts
        // where the workgroups then still fit side by side (two per CU on the two-slot ring), else two (one per CU, deep ring)
        auto fits = [&](int S) { return L.nchunks_h % (S * L.cpb) == 0 && ((L.nchunks_h / S) % 3) == 0; };
        const int per_cu = ts.lds * 2 <= 160 * 1024 ? 2 : 1;
        const int S = (fits(4) && tiles * 4 <= (long)per_cu * c->ncu && tiles * 2 <= c->ncu) ? 4 : (fits(2) && tiles * 2 <= c->ncu ? 2 : 1);       // two parts at two per CU (B = 2) measured slower: 54 -> 57 us
        if (S > 1) {
            const size_t need = (size_t)p.M * S * L.cout;
            if (need > c->splitk_floats) {
                if (c->d_splitk) { int rs = sync_all_slots(c); if (rs) return rs; HIP_TRY(c, hipFree(c->d_splitk)); c->d_splitk = nullptr; }
                HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_splitk), need * sizeof(float) * 4));   // x4: one region per in-flight slot
                c->splitk_floats = need;
            }
            const int slot_idx = (c->next_slot + c->inflight - 1) % c->inflight;
            float* part = c->d_splitk + (size_t)slot_idx * c->splitk_floats;
            ConvParamsH ps = p;
            ps.nchunks = L.nchunks_h / S;
            ps.g_x_boff = (L.cin / S) * 4; ps.g_w = (long)ps.nchunks * 128; ps.g_bias = L.npad_h; ps.g_y_coff = L.cout;
            ps.y = reinterpret_cast<uint8_t*>(part); ps.y_mode = 0; ps.y_ld = S * L.cout; ps.relu = 0; ps.range = nullptr;
            ps.ntiles_n = ntn; ps.mtile0 = 0;
            dim3 grid(mtiles * ntn, S);
            ps.deep = c->opt_deep_ring && tiles * S <= c->ncu;
            c->last_plan[li][0] = ((p.M + 127) / 128) * 1000 + plan[0].tile;
            // The fix-up folded into the launch (pose_tail.h, splitk_tail): with an x extent that is a multiple of 8 the S parts of a
            // tile run on one XCD, so the part that finishes last finds the others' partial sums in its own L2 and writes the stored
            // form itself - no splitk_fixup launch (7 us each at batch 1, two per forward).  Checked once per context on the device.
            bool fold = c->opt_fold_fixup && S <= 4 && grid.x % 8 == 0 && (int)grid.x <= SK_TILE_COUNTERS && c->ncu == c->dev_cus && !c->cu_partition &&
                        !c->user_stream;
            if (fold && c->xcd_rr < 0) {
                int ok = 0;
                HIP_TRY(c, xcd_round_robin_probe(c->stream, &ok));
                c->xcd_rr = ok;
            }
            fold = fold && c->xcd_rr > 0;
            if (fold) {
                ps.sk_counter = c->d_counters + 1 + c->max_batch;
                ps.sk_y = p.y; ps.sk_range = p.range; ps.sk_parts = S; ps.sk_relu = 1;
            }
            ProfScope pscope(c, L.label);
            HIP_TRY(c, launch_layer_h3(li, plan[0].tile, ps, grid, c->stream));
            if (!fold) HIP_TRY(c, launch_splitk_fixup(part, p.M, L.cout, S, 1, p.y, p.range, c->stream));
            return DAVO_OK;
        }
    }
    for (size_t i = 0; i < plan.size() && i < 2; ++i) {
        const TileShape ts = tile_shape(plan[i].tile);
        p.ntiles_n = L.npad_h / ts.bn;
        p.mtile0 = plan[i].row0 / ts.bm;
        const int full_m = p.M;
        p.M = plan[i].row0 + plan[i].rows;                    // rows past this launch's range are not its job
        const int mtiles = (plan[i].rows + ts.bm - 1) / ts.bm;
        dim3 grid(mtiles * p.ntiles_n, L.groups);
        p.deep = c->opt_deep_ring && plan.size() == 1 && (long)grid.x * grid.y <= c->ncu;       // at most one workgroup per CU
        c->last_plan[li][i] = ((plan[i].rows + 127) / 128) * 1000 + plan[i].tile;
        // no tile order here: cnv4's single launch of 128x128 tiles (two per CU, 6.5 rounds) measured 94.4 us in natural order and
        // 97.4 long-first (profiles/r04e_skip_padding_rows.md); the merged grids above and the float32 launches take one
        const std::string label = i == 0 ? std::string(L.label) : std::string(L.label) + ".rem";
        {
            ProfScope ps(c, label.c_str());
            bool done = false;
            if (c->opt_wave128 && L.groups == 1 && !fuse_pose) {
                if (plan[i].tile == TILE_256x256 && layer_h3w_supported(li, p)) {
                    HIP_TRY(c, launch_layer_h3w(li, p, grid, c->stream));
                    done = true;
                } else if (c->opt_wave128 >= 2 && i == 1 && plan.size() == 2 && plan[0].tile == TILE_256x256 && plan[1].tile == TILE_128x128 &&
                           L.npad_h == 256 && plan[1].row0 % 256 == 0 && plan[1].rows % 256 == 0) {
                    // the remainder rows of a layer whose main launch ran on conv_igemm_h3w: 256 x 64 tiles (conv_igemm_h3w64)
                    ConvParamsH pr = p;
                    pr.ntiles_n = 4; pr.mtile0 = plan[1].row0 / 256;
                    const hipError_t e = launch_layer_h3w64(li, pr, c->stream);
                    if (e == hipSuccess) done = true;
                    else if (e != hipErrorNotSupported) HIP_TRY(c, e);
                }
            }
            if (!done) HIP_TRY(c, launch_layer_h3(li, plan[i].tile, p, grid, c->stream));
        }
        p.M = full_m;
    }
    return DAVO_OK;
}

// cnv1 of the f16x3 path from an LDS-staged input patch (conv_patch_h3.h).  fused: the patch is built
// from the raw inputs (mask + pack fused in); otherwise it is copied from the packed tensor.
int run_cnv1_patch(davo_ctx* c, bool fused, const void* d_img, const void* d_flow, const void* d_seg, void* y, int NB) {
    const ConvLayer& L = c->L[0];
    ConvPatchParams p{};
    int Ho, Wo, pt, pl;
    same_pad(c->H, 7, 2, 1, &Ho, &pt);
    same_pad(c->W, 7, 2, 1, &Wo, &pl);
    p.x = reinterpret_cast<const uint8_t*>(c->d_packed); p.w = c->d_w1patch; p.bias = L.d_bh; p.y = static_cast<uint8_t*>(y);
    p.zeros = reinterpret_cast<const uint8_t*>(c->d_zeros);
    p.H = c->H; p.W = c->W; p.Ho = Ho; p.Wo = Wo; p.pad_t = pt; p.pad_l = pl;
    p.tiles_x = (Wo + cp1::TW - 1) / cp1::TW; p.tiles_y = (Ho + cp1::TH - 1) / cp1::TH;
    p.out_scale = ldexpf(1.0f / L.wscale, c->act_shift[0]);
    p.bias_scale = L.wscale;
    p.range = c->d_range;
    p.ntiles = NB * p.tiles_x * p.tiles_y;
    p.img = static_cast<const uint8_t*>(d_img); p.flow = static_cast<const float*>(d_flow);
    p.seg = static_cast<const float*>(d_seg); p.tab = c->d_tab; p.v = c->v;
    c->last_plan[0][0] = ((NB * Ho * Wo + 127) / 128) * 1000 + 99; c->last_plan[0][1] = 0;
    const int nblk = p.ntiles < 3 * c->ncu ? p.ntiles : 3 * c->ncu;   // 3 workgroups per CU, each walks its tiles
    if (const char* e = tuning_env("DAVO_PDBG")) p.dbg = atoi(e);      // tuning build only
    ProfScope ps(c, "cnv1");
    HIP_TRY(c, launch_cnv1_patch(fused, p, nblk, c->stream));
    return DAVO_OK;
}

// cnv2 of the f16x3 path from an LDS-staged input patch (conv_patch_h3.h, conv_patch_cnv2_h3)
int run_cnv2_patch(davo_ctx* c, const void* x, void* y, int NB) {
    const ConvLayer& L = c->L[1];
    ConvPatchParams p{};
    int Ho, Wo, pt, pl;
    same_pad(c->H1, 5, 2, 1, &Ho, &pt);
    same_pad(c->W1, 5, 2, 1, &Wo, &pl);
    p.x = static_cast<const uint8_t*>(x); p.w = c->d_w2patch; p.bias = L.d_bh; p.y = static_cast<uint8_t*>(y);
    p.zeros = reinterpret_cast<const uint8_t*>(c->d_zeros);
    p.H = c->H1; p.W = c->W1; p.Ho = Ho; p.Wo = Wo; p.pad_t = pt; p.pad_l = pl;
    p.tiles_x = (Wo + cp2::TW - 1) / cp2::TW; p.tiles_y = (Ho + cp2::TH - 1) / cp2::TH;
    p.out_scale = ldexpf(1.0f / L.wscale, c->act_shift[1] - c->act_shift[0]);
    p.bias_scale = ldexpf(L.wscale, c->act_shift[0]);
    p.range = c->d_range ? c->d_range + 1 : nullptr;
    p.ntiles = NB * p.tiles_x * p.tiles_y;
    c->last_plan[1][0] = ((NB * Ho * Wo + 127) / 128) * 1000 + 98; c->last_plan[1][1] = 0;
    const int nblk = p.ntiles < 2 * c->ncu ? p.ntiles : 2 * c->ncu;   // 2 workgroups per CU (120 weight registers per lane), each walks its tiles
    if (const char* e = tuning_env("DAVO_PDBG")) p.dbg = atoi(e);      // tuning build only
    ProfScope ps(c, "cnv2");
    HIP_TRY(c, launch_cnv2_patch(p, nblk, c->stream));
    return DAVO_OK;
}

// cnv1 of the float32 mode from an LDS-staged input patch (conv_patch_f32.h); x: the packed float32 input [NB][H][W][8], y: [NB][H1][W1][16]
int run_cnv1_patch_f32(davo_ctx* c, const void* x, void* y, int NB) {
    const ConvLayer& L = c->L[0];
    ConvPatchParams p{};
    int Ho, Wo, pt, pl;
    same_pad(c->H, 7, 2, 1, &Ho, &pt);
    same_pad(c->W, 7, 2, 1, &Wo, &pl);
    p.x = static_cast<const uint8_t*>(x); p.w = reinterpret_cast<const uint8_t*>(c->d_w1patch_f32); p.bias = L.d_b; p.y = static_cast<uint8_t*>(y);
    p.zeros = reinterpret_cast<const uint8_t*>(c->d_zeros);
    p.H = c->H; p.W = c->W; p.Ho = Ho; p.Wo = Wo; p.pad_t = pt; p.pad_l = pl;
    p.tiles_x = (Wo + cp1::TW - 1) / cp1::TW; p.tiles_y = (Ho + cp1::TH - 1) / cp1::TH;
    p.ntiles = NB * p.tiles_x * p.tiles_y;
    c->last_plan[0][0] = ((NB * Ho * Wo + 127) / 128) * 1000 + 99; c->last_plan[0][1] = 0;
    const int nblk = p.ntiles < 3 * c->ncu ? p.ntiles : 3 * c->ncu;   // three workgroups per CU, each walks its tiles
    ProfScope ps(c, "cnv1");
    HIP_TRY(c, launch_cnv1_patch_f32(p, nblk, c->stream));
    return DAVO_OK;
}

// cnv2 of the float32 mode from an LDS-staged input patch (conv_patch_f32.h); x: float32 NHWC [NB][H1][W1][16], y: [NB][H2][W2][32]
int run_cnv2_patch_f32(davo_ctx* c, const void* x, void* y, int NB) {
    const ConvLayer& L = c->L[1];
    ConvPatchParams p{};
    int Ho, Wo, pt, pl;
    same_pad(c->H1, 5, 2, 1, &Ho, &pt);
    same_pad(c->W1, 5, 2, 1, &Wo, &pl);
    p.x = static_cast<const uint8_t*>(x); p.w = reinterpret_cast<const uint8_t*>(c->d_w2patch_f32); p.bias = L.d_b; p.y = static_cast<uint8_t*>(y);
    p.zeros = reinterpret_cast<const uint8_t*>(c->d_zeros);
    p.H = c->H1; p.W = c->W1; p.Ho = Ho; p.Wo = Wo; p.pad_t = pt; p.pad_l = pl;
    p.tiles_x = (Wo + cp2::TW - 1) / cp2::TW; p.tiles_y = (Ho + cp2::TH - 1) / cp2::TH;
    p.ntiles = NB * p.tiles_x * p.tiles_y;
    c->last_plan[1][0] = ((NB * Ho * Wo + 127) / 128) * 1000 + 98; c->last_plan[1][1] = 0;
    const int nblk = p.ntiles < 2 * c->ncu ? p.ntiles : 2 * c->ncu;   // two workgroups per CU (three measured 4 % slower), each walks its tiles
    ProfScope ps(c, "cnv2");
    HIP_TRY(c, launch_cnv2_patch_f32(p, nblk, c->stream));
    return DAVO_OK;
}

// cnv3 of the float32 mode from an LDS-staged input patch (conv_patch_f32.h); x: float32 NHWC [NB][H2][W2][32], y: [NB][H2][W2][64]
int run_cnv3_patch_f32(davo_ctx* c, const void* x, void* y, int NB) {
    const ConvLayer& L = c->L[2];
    ConvPatchParams p{};
    p.x = static_cast<const uint8_t*>(x); p.w = reinterpret_cast<const uint8_t*>(c->d_w3patch_f32); p.bias = L.d_b; p.y = static_cast<uint8_t*>(y);
    p.zeros = reinterpret_cast<const uint8_t*>(c->d_zeros);
    p.H = c->H2; p.W = c->W2; p.Ho = c->H2; p.Wo = c->W2; p.pad_t = cp3::RATE; p.pad_l = cp3::RATE;
    p.tiles_x = (p.Wo + cp3::TW - 1) / cp3::TW; p.tiles_y = (p.Ho + cp3::TH - 1) / cp3::TH;
    p.ntiles = NB * p.tiles_x * p.tiles_y;
    c->last_plan[2][0] = ((NB * p.Ho * p.Wo + 127) / 128) * 1000 + 97; c->last_plan[2][1] = 0;
    const int nblk = p.ntiles < 3 * c->ncu ? p.ntiles : 3 * c->ncu;
    ProfScope ps(c, "cnv3");
    HIP_TRY(c, launch_cnv3_patch_f32(p, nblk, c->stream));
    return DAVO_OK;
}

// cnv3 of the f16x3 path from an LDS-staged input patch (conv_patch_h3.h, conv_patch_cnv3_h3)
int run_cnv3_patch(davo_ctx* c, const void* x, void* y, int NB) {
    const ConvLayer& L = c->L[2];
    ConvPatchParams p{};
    p.x = static_cast<const uint8_t*>(x); p.w = c->d_w3patch; p.bias = L.d_bh; p.y = static_cast<uint8_t*>(y);
    p.zeros = reinterpret_cast<const uint8_t*>(c->d_zeros);
    p.H = c->H2; p.W = c->W2; p.Ho = c->H2; p.Wo = c->W2; p.pad_t = cp3::RATE; p.pad_l = cp3::RATE;
    p.tiles_x = (p.Wo + cp3::TW - 1) / cp3::TW; p.tiles_y = (p.Ho + cp3::TH - 1) / cp3::TH;
    p.out_scale = ldexpf(1.0f / L.wscale, c->act_shift[2] - c->act_shift[1]);
    p.bias_scale = ldexpf(L.wscale, c->act_shift[1]);
    p.range = c->d_range ? c->d_range + 2 : nullptr;
    p.ntiles = NB * p.tiles_x * p.tiles_y;
    c->last_plan[2][0] = ((NB * p.Ho * p.Wo + 127) / 128) * 1000 + 97; c->last_plan[2][1] = 0;
    const int nblk = p.ntiles < 3 * c->ncu ? p.ntiles : 3 * c->ncu;   // 3 workgroups per CU, each walks its tiles
    if (const char* e = tuning_env("DAVO_PDBG")) p.dbg = atoi(e);      // tuning build only
    ProfScope ps(c, "cnv3");
    HIP_TRY(c, launch_cnv3_patch(p, nblk, c->stream));
    return DAVO_OK;
}

int run_direct(davo_ctx* c, const char* label, const float* x, int N, int Hin, int Win, int cin, int x_ld,
               int x_coff, const std::string& wname, const std::string& bname, int KS, int cout, int stride,
               int rate, float* y, int y_ld, int y_coff) {
    int Ho, Wo, pt, pl;
    same_pad(Hin, KS, stride, rate, &Ho, &pt);
    same_pad(Win, KS, stride, rate, &Wo, &pl);
    for (const std::string* n : {&wname, &bname}) {            // impl 1 is the only reader of the raw device copies: made on demand
        HostTensor& t = c->weights.at(*n);
        if (!t.dev) { int rc = upload(c, t.data, &t.dev); if (rc) return rc; }
    }
    ProfScope ps(c, label);
    HIP_TRY(c, launch_conv_direct(x, N, Hin, Win, cin, x_ld, x_coff, c->weights.at(wname).dev, KS, cout,
                                  c->weights.at(bname).dev, stride, rate, pt, pl, Ho, Wo, 1, y, y_ld, y_coff, c->stream));
    return DAVO_OK;
}

}  // namespace

int forward_device(davo_ctx* c, int B, const void* d_img, const void* d_flow, const void* d_seg, void* d_pose) {
    if (B < 1 || B > c->max_batch) return fail(c, DAVO_ERR_INVALID, "batch %d outside [1,%d]", B, c->max_batch);
    if (!d_img || !d_flow || !d_seg || !d_pose) return fail(c, DAVO_ERR_INVALID, "null device pointer");
    {
        std::string names;
        if (missing_weights(c, &names)) return fail(c, DAVO_ERR_NOT_READY, "weights not loaded: %s", names.c_str());
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->pred_ready) { int rc = build_pred_weights(c); if (rc) return rc; }
    bool h3 = c->impl == 0 && c->precision == 1;
    c->f32_fallback = false;
    if (h3 && !c->packed_h_ready) { int rc = build_packed_weights_h3(c); if (rc) return rc; }
    if (h3 && c->weight_channel_spread_log2 > MAX_WEIGHT_CHANNEL_SPREAD_LOG2) {
        // a consumer's per-input-channel weight norms span more than 2^14: per-layer storage scales cannot keep every channel's fp16
        // pair float32-grade (weights.hip).  The reference's float32 graph has no such limit (nets/posenn.py:205-215): float32 kernels.
        if (!c->opt_auto_range)
            return fail(c, DAVO_ERR_RANGE, "`%s': per-input-channel weight norms span 2^%d (> 2^%d): the f16x3 storage cannot hold every channel "
                        "float32-grade - davo_set_precision(ctx, 0), or leave \"auto_range\" on", c->weight_channel_spread_layer.c_str(),
                        c->weight_channel_spread_log2, MAX_WEIGHT_CHANNEL_SPREAD_LOG2);
        h3 = false;
        c->f32_fallback = true;              // counted once per API call by the caller (api.hip)
        char note[256];
        snprintf(note, sizeof note, "float32 kernels: per-input-channel weight norms of `%s' span 2^%d (> 2^%d)",
                 c->weight_channel_spread_layer.c_str(), c->weight_channel_spread_log2, MAX_WEIGHT_CHANNEL_SPREAD_LOG2);
        c->range_report = note;
    }
    if (!h3 && c->impl == 0 && !c->packed_ready) { int rc = build_packed_weights(c); if (rc) return rc; }      // float32 kernels: packed at their first use
    unsigned* const range_reset = (h3 && c->range_zero) ? c->d_range : nullptr;

    const int H = c->H, W = c->W, HW = H * W, NB = 2 * B;
    const Variant& v = c->v;
    hipStream_t s = c->stream;
    auto wdev = [&](const char* n) -> const float* {
        auto it = c->weights.find(n);
        return it == c->weights.end() ? nullptr : it->second.dev;
    };
    // Launches are ~6 us each whatever they do; at batch 1 the path is 11 of them around 0.1 ms of work.  Where a launch can be
    // folded into its neighbour at less than that, small batches do it (measured per batch: profiles/):
    //   the excitation MLP in the squeeze launch's last workgroup (costs two memory-side round trips per workgroup: +2 us at
    //   B = 1, +18 us at B = 32), mask + pack inside cnv1's patch fill (level at B = 32).
    const bool fold_excite = v.att_source == 1 && (c->opt_fold_tails >= 1 || (c->opt_fold_tails < 0 && B <= FOLD_EXCITE_MAX_BATCH));
    const bool fold_pose = c->opt_fold_tails == 1;
    if (fold_excite) {
        // squeeze + excitation in one launch: the workgroup that delivers a triplet's last partial sum evaluates its tables
        ProfScope ps(c, "se_squeeze_partial");
        HIP_TRY(c, launch_se_squeeze_excite(static_cast<const float*>(d_flow), B, HW, v, c->d_partial, c->d_counters + 1,
                                            wdev("pose_exp_net/se_flow/bottleneck_fc/kernel"), wdev("pose_exp_net/se_flow/bottleneck_fc/bias"),
                                            wdev("pose_exp_net/se_flow/recover_fc/kernel"), wdev("pose_exp_net/se_flow/recover_fc/bias"),
                                            wdev("pose_exp_net/pose_exp_net/seg_channel_weight/weight"), c->d_tab, range_reset, s));
    } else if (v.att_source == 1) {
        ProfScope ps(c, "se_squeeze_partial");
        HIP_TRY(c, launch_se_squeeze(static_cast<const float*>(d_flow), B, HW, v, c->d_partial, s));
    }
    if (!fold_excite) {
        ProfScope ps(c, "se_excite");
        HIP_TRY(c, launch_se_excite(c->d_partial, B, HW, v,
                                    wdev("pose_exp_net/se_flow/bottleneck_fc/kernel"), wdev("pose_exp_net/se_flow/bottleneck_fc/bias"),
                                    wdev("pose_exp_net/se_flow/recover_fc/kernel"), wdev("pose_exp_net/se_flow/recover_fc/bias"),
                                    wdev("pose_exp_net/pose_exp_net/seg_channel_weight/weight"), c->d_tab, range_reset, s));
    }
    // f16x3, fuse_pack: cnv1 builds its input patch straight from the raw inputs (mask + pack fused in,
    // the packed tensor never touches HBM).  Measured equal in time to mask_pack + cnv1 (the fused fill is bound
    // by its byte loads), so the two-kernel form stays the default.  Tuning build: DAVO_FUSE_PACK=1 / DAVO_CNV1_PATCH=0.
    const char* fe = tuning_env("DAVO_FUSE_PACK");
    const char* pe = tuning_env("DAVO_CNV1_PATCH");
    const bool fuse_env = (fe && atoi(fe) == 1) || c->opt_fuse_pack == 1 || (c->opt_fuse_pack < 0 && B <= FUSE_PACK_MAX_BATCH);
    const bool patch1 = !(pe && atoi(pe) == 0);
    const bool fused = h3 && patch1 && fuse_env;
    c->packed_valid = !fused;
    c->last_img = d_img; c->last_flow = d_flow; c->last_seg = d_seg;
    c->packed_ld = c->impl == 0 ? 8 : 10;
    if (!fused) {
        ProfScope ps(c, "mask_pack");
        HIP_TRY(c, launch_mask_pack(h3 ? 16 : (c->impl == 0 ? 8 : 10), static_cast<const uint8_t*>(d_img),
                                    static_cast<const float*>(d_flow), static_cast<const float*>(d_seg), c->d_tab, v, B, H, W,
                                    c->d_packed, s));
    }
    const int c6 = v.cnv6_out;
    float** a = c->d_act;
    int rc;
    bool pose_fused = false;
    int pose_bm = 0, pose_mt = 0, pose_ntn = 0;
    c->cnv7_valid = true;
    if (h3) {
        if (patch1) { if ((rc = run_cnv1_patch(c, fused, d_img, d_flow, d_seg, a[0], NB))) return rc; }
        else if ((rc = run_conv_layer_h3(c, 0, c->d_packed, 8, H, W, a[0], 16, false, NB))) return rc;
        const char* p2e = tuning_env("DAVO_CNV2_PATCH");
        if (c->opt_patch_cnv2 && c->L[1].tile_h < 0 && !(p2e && atoi(p2e) == 0)) { if ((rc = run_cnv2_patch(c, a[0], a[1], NB))) return rc; }
        else if ((rc = run_conv_layer_h3(c, 1, a[0], 16, c->H1, c->W1, a[1], 32, false, NB))) return rc;
        const char* p3e = tuning_env("DAVO_CNV3_PATCH");
        if (c->opt_patch_cnv3 && c->L[2].tile_h < 0 && !(p3e && atoi(p3e) == 0)) { if ((rc = run_cnv3_patch(c, a[1], a[2], NB))) return rc; }
        else if ((rc = run_conv_layer_h3(c, 2, a[1], 32, c->H2, c->W2, a[2], 64, false, NB))) return rc;
        if ((rc = run_conv_layer_h3(c, 3, a[2], 64, c->H2, c->W2, a[3], 128, false, NB))) return rc;
        if ((rc = run_conv_layer_h3(c, 4, a[3], 128, c->H2, c->W2, a[4], 256, false, NB))) return rc;
        if ((rc = run_conv_layer_h3(c, 5, a[4], 256, c->H2, c->W2, a[5], 2 * c6, false, NB))) return rc;
        pose_fused = c->opt_fuse_pose && c->H3 * c->W3 >= 128;
        if ((rc = run_conv_layer_h3(c, 6, a[5], 2 * c6, c->H2, c->W2, a[6], 512, true, NB, pose_fused, &pose_bm, &pose_mt, &pose_ntn,
                                    fold_pose ? static_cast<float*>(d_pose) : nullptr))) return rc;
        c->cnv7_valid = !pose_fused;
    } else if (c->impl == 0) {
        if (c->opt_patch_f32 && c->d_w1patch_f32 && c->packed_ld == 8) { if ((rc = run_cnv1_patch_f32(c, c->d_packed, a[0], NB))) return rc; }
        else if ((rc = run_conv_layer(c, 0, c->d_packed, 8, H, W, a[0], 16, NB))) return rc;
        if (c->opt_patch_f32 && c->d_w2patch_f32) { if ((rc = run_cnv2_patch_f32(c, a[0], a[1], NB))) return rc; }
        else if ((rc = run_conv_layer(c, 1, a[0], 16, c->H1, c->W1, a[1], 32, NB))) return rc;
        if (c->opt_patch_f32 && c->d_w3patch_f32) { if ((rc = run_cnv3_patch_f32(c, a[1], a[2], NB))) return rc; }
        else if ((rc = run_conv_layer(c, 2, a[1], 32, c->H2, c->W2, a[2], 64, NB))) return rc;
        if ((rc = run_conv_layer(c, 3, a[2], 64, c->H2, c->W2, a[3], 128, NB))) return rc;
        if ((rc = run_conv_layer(c, 4, a[3], 128, c->H2, c->W2, a[4], 256, NB))) return rc;
        if ((rc = run_conv_layer(c, 5, a[4], 256, c->H2, c->W2, a[5], 2 * c6, NB))) return rc;
        // float32 mode, round 4: the pose head in cnv7's epilogue like the f16x3 path's (the 109 MB activation is neither written nor
        // read back, pose_head_partial + pose_finish become pose_from_tiles); tiles of 128 rows must not span more than two images
        pose_fused = c->opt_fuse_pose && c->H3 * c->W3 >= 128 && c->L[6].npad == 256;
        pose_bm = 128; pose_ntn = 8;
        if ((rc = run_conv_layer(c, 6, a[5], 2 * c6, c->H2, c->W2, a[6], 512, NB, pose_fused, &pose_mt))) return rc;
        c->cnv7_valid = !pose_fused;
    } else {
        const std::string P = "pose_exp_net/";
        const int c10 = 2 * v.cin_per_frame;
        if (v.cin_per_frame != 5) return fail(c, DAVO_ERR_INVALID, "impl 1 supports the 10-channel (v1) input only");
        if ((rc = run_direct(c, "cnv1", c->d_packed, NB, H, W, c10, 10, 0, P + "cnv1/weights", P + "cnv1/biases", 7, 16, 2, 1, a[0], 16, 0))) return rc;
        if ((rc = run_direct(c, "cnv2", a[0], NB, c->H1, c->W1, 16, 16, 0, P + "cnv2/weights", P + "cnv2/biases", 5, 32, 2, 1, a[1], 32, 0))) return rc;
        if ((rc = run_direct(c, "cnv3", a[1], NB, c->H2, c->W2, 32, 32, 0, P + "cnv3/weights", P + "cnv3/biases", 3, 64, 1, 2, a[2], 64, 0))) return rc;
        if ((rc = run_direct(c, "cnv4", a[2], NB, c->H2, c->W2, 64, 64, 0, P + "cnv4/weights", P + "cnv4/biases", 3, 128, 1, 4, a[3], 128, 0))) return rc;
        if ((rc = run_direct(c, "cnv5", a[3], NB, c->H2, c->W2, 128, 128, 0, P + "cnv5/weights", P + "cnv5/biases", 3, 256, 1, 8, a[4], 256, 0))) return rc;
        const char* heads[2] = {"rotation", "translation"};
        for (int h = 0; h < 2; ++h) {
            const std::string hp = P + "pose/" + heads[h] + "/";
            if ((rc = run_direct(c, "cnv6", a[4], NB, c->H2, c->W2, 256, 256, 0, hp + "cnv6/weights", hp + "cnv6/biases", 3, c6, 1, 2, a[5], 2 * c6, h * c6))) return rc;
            if ((rc = run_direct(c, "cnv7", a[5], NB, c->H2, c->W2, c6, 2 * c6, h * c6, hp + "cnv7/weights", hp + "cnv7/biases", 3, 256, 2, 1, a[6], 512, h * 256))) return rc;
        }
    }
    bool snap_done = false;
    if (!(pose_fused && fold_pose)) {
        ProfScope ps(c, "pose_head");
        if (pose_fused) {
            const int slot_idx = (c->next_slot + c->inflight - 1) % c->inflight;
            HIP_TRY(c, launch_pose_from_tiles(c->d_pose_tiles + (size_t)slot_idx * c->pose_tiles_floats, NB, c->H3 * c->W3, pose_bm,
                                              pose_mt, pose_ntn, c->d_bpred, static_cast<float*>(d_pose), h3 ? c->snap : SnapArgs{}, s));
            snap_done = true;
        } else {
            HIP_TRY(c, launch_pose_head(a[6], NB, c->H3 * c->W3, c->d_wpred, c->d_bpred, c->d_pose_partial, static_cast<float*>(d_pose), s));
        }
    }
    // the range guard's conditional copy of this batch's inputs (api.hip: tickets) rides in pose_from_tiles; other pose heads get a launch
    if (h3 && c->snap.record && !snap_done) HIP_TRY(c, launch_range_guard_snapshot(c->snap, s));
    c->last_B = B;
    c->last_precision = h3 ? 1 : 0;
    return DAVO_OK;
}

}  // namespace davo
