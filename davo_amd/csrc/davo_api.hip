// davo_api.hip — C ABI of libdavo_hip.so (see include/davo_hip.h) and the host-side launch
// plan of the pose path:
//
//   se_squeeze_partial -> se_excite -> mask_pack -> cnv1..cnv5 -> cnv6 (rotation|translation
//   fused into one N = 2*cnv6_out GEMM, both read cnv5: nets/posenn.py:222-238)
//   -> cnv7 (grouped x2) -> pose_head.
//
// The two PoseNN calls of a triplet (davo.py:1456-1457, shared weights) run as one batch of
// 2B pair images.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/davo_hip.h"
#include "conv_igemm.h"
#include "conv_igemm_h3.h"
#include "conv_patch_h3.h"
#include "prologue.h"

using namespace davo;

namespace {

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
};

void same_pad(int in, int k, int stride, int rate, int* out, int* before) {
    const int o = (in + stride - 1) / stride;
    const int keff = (k - 1) * rate + 1;
    int total = (o - 1) * stride + keff - in;
    if (total < 0) total = 0;
    *out = o;
    *before = total / 2;
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace

// One in-flight batch: its own HIP stream and activation workspace.  Weights are shared.
struct Slot {
    hipStream_t stream = nullptr;
    float *d_partial = nullptr, *d_tab = nullptr, *d_packed = nullptr, *d_pose_partial = nullptr;
    float* d_act[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

struct davo_ctx {
    int device = 0, H = 0, W = 0, max_batch = 0;
    std::vector<Slot> slots;                   // slots[0] is created by davo_create
    int inflight = 1, next_slot = 0;
    bool user_stream = false;
    bool opt_fuse_pose = true;                 // f16x3: pose head fused into cnv7's epilogue (davo_set_option)
    bool opt_fuse_pack = false;                // f16x3: mask+pack fused into cnv1's patch fill
    float* d_pose_tiles = nullptr;             // per-tile partial sums of the fused pose head
    size_t pose_tiles_floats = 0;
    bool cnv7_valid = true;
    Variant v{};
    int impl = 0;
    int precision = 1;                         // 0 = FP32 MFMA (bit-exact fmaf chains), 1 = f16x3 split (default)
    bool packed_h_ready = false;
    int last_precision = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;
    std::map<std::string, HostTensor> weights;
    std::vector<std::string> needed;
    bool packed_ready = false;
    ConvLayer L[7];                            // cnv1..cnv5, cnv6 (fused), cnv7 (grouped)
    float *d_wpred = nullptr, *d_bpred = nullptr;
    uint8_t* d_w1patch = nullptr;               // cnv1 B fragments for conv_patch_cnv1_h3
    // geometry
    int H1, W1, H2, W2, H3, W3;
    // workspace
    float *d_partial = nullptr, *d_tab = nullptr, *d_packed = nullptr, *d_zeros = nullptr, *d_pose_partial = nullptr;
    float* d_act[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t act_floats_per_img[7];
    int act_ch[7];
    int packed_ld = 8;
    int last_B = 0;
    bool packed_valid = true;                  // false when cnv1 consumed the raw inputs directly (fused)
    const void *last_img = nullptr, *last_flow = nullptr, *last_seg = nullptr;
    int last_plan[7][2] = {};                  // per layer, per launch: mtiles*1000 + BN (reported by the bench)
    // host-API staging
    void *s_img = nullptr, *s_flow = nullptr, *s_seg = nullptr, *s_pose = nullptr;
    hipStream_t copy_stream = nullptr;         // H2D of the next sub-batch runs here while the previous one computes
    std::vector<hipEvent_t> copy_done;
    // f16x3 range management: activations are stored as fp16 pairs scaled by 2^act_shift[layer] (davo_calibrate);
    // every storing epilogue atomicMax-es the largest stored magnitude into d_range[layer]
    int act_shift[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned* d_range = nullptr;               // [8]
    int host_chunk = 8;                        // davo_forward: windows per sub-batch (davo_set_option "host_chunk"; 0 = whole batch)
    // profiling
    bool prof = false;
    bool prof_dominant_only = false;           // profile mode 2: bracket only the main cnv6 launch
    std::vector<ProfEntry> prof_entries;
    std::vector<hipEvent_t> event_pool;
};

namespace {

int fail(davo_ctx* c, int code, const char* fmt, ...);

// bind a slot's stream and workspace to the members every launch helper uses
void activate_slot(davo_ctx* c, int i) {
    const Slot& s = c->slots[i];
    if (!(c->user_stream && i == 0)) c->stream = s.stream;
    c->d_partial = s.d_partial; c->d_tab = s.d_tab; c->d_packed = s.d_packed; c->d_pose_partial = s.d_pose_partial;
    for (int k = 0; k < 7; ++k) c->d_act[k] = s.d_act[k];
}

void free_slot(Slot& s) {
    for (auto p : s.d_act) if (p) (void)hipFree(p);
    void* misc[] = {s.d_partial, s.d_tab, s.d_packed, s.d_pose_partial};
    for (auto p : misc) if (p) (void)hipFree(p);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    s = Slot();
}

int fail(davo_ctx* c, int code, const char* fmt, ...) {
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
            return fail(c, DAVO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// ---- conv dispatch ----------------------------------------------------------------------
template <int KS, int STRIDE, int BN, int LAYER>
hipError_t launch_conv_t(const ConvParams& p, dim3 grid, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_igemm_f32<KS, STRIDE, BN, LAYER>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Tile<BN>::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), Tile<BN>::LDS_BYTES, s, p);
    return hipGetLastError();
}

template <int KS, int STRIDE>
hipError_t launch_conv_bn(int BN, const ConvParams& p, dim3 grid, hipStream_t s) {
    switch (BN) {
        case 32: return launch_conv_t<KS, STRIDE, 32, 0>(p, grid, s);
        case 64: return launch_conv_t<KS, STRIDE, 64, 0>(p, grid, s);
        case 128: return launch_conv_t<KS, STRIDE, 128, 0>(p, grid, s);
    }
    return hipErrorInvalidValue;
}

// generic shapes (davo_conv2d_same and non-default cnv6 widths)
hipError_t launch_conv(int KS, int stride, int BN, const ConvParams& p, dim3 grid, hipStream_t s) {
    if (stride == 1) {
        switch (KS) {
            case 1: return launch_conv_bn<1, 1>(BN, p, grid, s);
            case 3: return launch_conv_bn<3, 1>(BN, p, grid, s);
            case 5: return launch_conv_bn<5, 1>(BN, p, grid, s);
            case 7: return launch_conv_bn<7, 1>(BN, p, grid, s);
        }
    } else if (stride == 2) {
        switch (KS) {
            case 1: return launch_conv_bn<1, 2>(BN, p, grid, s);
            case 3: return launch_conv_bn<3, 2>(BN, p, grid, s);
            case 5: return launch_conv_bn<5, 2>(BN, p, grid, s);
            case 7: return launch_conv_bn<7, 2>(BN, p, grid, s);
        }
    }
    return hipErrorInvalidValue;
}

// the seven PoseNN layers, each under its own kernel name (LAYER tag); BN is chosen per launch
template <int KS, int STRIDE, int LAYER>
hipError_t launch_tagged(int BN, const ConvParams& p, dim3 grid, hipStream_t s) {
    switch (BN) {
        case 32: return launch_conv_t<KS, STRIDE, 32, LAYER>(p, grid, s);
        case 64: return launch_conv_t<KS, STRIDE, 64, LAYER>(p, grid, s);
        case 128: return launch_conv_t<KS, STRIDE, 128, LAYER>(p, grid, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_layer(int layer, int BN, const ConvParams& p, dim3 grid, hipStream_t s) {
    switch (layer) {
        case 0: return launch_tagged<7, 2, 1>(BN, p, grid, s);
        case 1: return launch_tagged<5, 2, 2>(BN, p, grid, s);
        case 2: return launch_tagged<3, 1, 3>(BN, p, grid, s);
        case 3: return launch_tagged<3, 1, 4>(BN, p, grid, s);
        case 4: return launch_tagged<3, 1, 5>(BN, p, grid, s);
        case 5: return launch_tagged<3, 1, 6>(BN, p, grid, s);
        case 6: return launch_tagged<3, 2, 7>(BN, p, grid, s);
    }
    return hipErrorInvalidValue;
}

// ---- launch planning ----------------------------------------------------------------------
// Workgroups of one launch all take the same time, and the dispatcher refills both slots of a
// CU together, so a grid that is not a whole number of rounds (256 CUs x resident workgroups)
// pays for a full last round: at B = 32 the 3,328 tiles of cnv5/cnv6 are 6.5 rounds of 512 and
// ran in the time of 7.  A layer is therefore issued as a main launch of whole rounds at the
// widest N tile plus, when it pays, a remainder launch with a narrower N tile (more, shorter
// workgroups) that again fills whole rounds.  Costs are in units of one round of 128x128 tiles.
struct Launch { int mtile0, mtiles, BN; };

int slots_for(int BN) { return 256 * (BN == 32 ? 3 : 2); }         // LDS 46 / 55 / 74 KB per workgroup
// time of one round (every CU full) relative to a round of 128x128 tiles: resident workgroups
// per CU x tile area / measured relative efficiency of the narrower tiles
double tile_cost(int BN) { return BN == 128 ? 1.0 : BN == 64 ? 0.5 / 0.92 : 0.375 / 0.75; }

double rounds_cost(long tiles, int BN) {
    const long s = slots_for(BN);
    return (double)((tiles + s - 1) / s) * tile_cost(BN);
}

std::vector<Launch> plan_layer(int mtiles, int npad, int groups) {
    int bmax = npad % 128 == 0 ? 128 : npad % 64 == 0 ? 64 : 32;
    // tuning overrides (measurement only): DAVO_FORCE_BN=32|64|128 -> one launch at that N tile,
    // DAVO_PLAN=single -> one launch at the widest N tile
    if (const char* e = getenv("DAVO_FORCE_BN")) {
        const int bn = atoi(e);
        if ((bn == 32 || bn == 64 || bn == 128) && npad % bn == 0) return {{0, mtiles, bn}};
    }
    if (const char* e = getenv("DAVO_PLAN"))
        if (!strcmp(e, "single")) return {{0, mtiles, bmax}};
    std::vector<Launch> best;
    double best_cost = 1e30;
    for (int bn = bmax; bn >= 32; bn >>= 1) {                          // one launch
        const double c = rounds_cost((long)mtiles * (npad / bn) * groups, bn);
        if (c < best_cost - 1e-9) { best_cost = c; best = {{0, mtiles, bn}}; }
    }
    const long per_m = (long)(npad / bmax) * groups;                   // tiles per M tile at bmax
    const long s = slots_for(bmax);
    const int main_m = (int)(((long)mtiles * per_m / s) * s / per_m);  // whole rounds only
    if (main_m > 0 && main_m < mtiles && (main_m * per_m) % s == 0) {
        const double cm = rounds_cost(main_m * per_m, bmax);
        for (int bn = bmax; bn >= 32; bn >>= 1) {
            const double c = cm + rounds_cost((long)(mtiles - main_m) * (npad / bn) * groups, bn) + 0.02;
            if (c < best_cost - 1e-9) { best_cost = c; best = {{0, main_m, bmax}, {main_m, mtiles - main_m, bn}}; }
        }
    }
    return best;
}

// ---- f16x3 path: tile shapes, weight packing, dispatch --------------------------------------
// tile id -> (WM, WN, TM, TN): BM = WM*TM*32, BN = WN*TN*32, threads = WM*WN*64
enum { TILE_128x32 = 0, TILE_256x64 = 1, TILE_256x128 = 2, TILE_128x256 = 3, TILE_128x128 = 4, TILE_256x256 = 5, NUM_TILES = 6 };
struct TileShape { int bm, bn, threads, lds; };
TileShape tile_shape(int t) {
    switch (t) {
        case TILE_128x32: return {128, 32, 256, TileH<4, 1, 1, 1>::LDS_BYTES_DMA};
        case TILE_256x64: return {256, 64, 512, TileH<4, 2, 2, 1>::LDS_BYTES_DMA};
        case TILE_256x128: return {256, 128, 512, TileH<4, 2, 2, 2>::LDS_BYTES_DMA};
        case TILE_128x256: return {128, 256, 512, TileH<2, 4, 2, 2>::LDS_BYTES_DMA};
        case TILE_256x256: return {256, 256, 512, TileH<4, 2, 2, 4>::LDS_BYTES_DMA};
        default: return {128, 128, 512, TileH<4, 2, 1, 2>::LDS_BYTES_DMA};
    }
}

#ifndef DAVO_REM_STAGES
#define DAVO_REM_STAGES 3
#endif
// All f16x3 launches are LDS-DMA staged.  SMALLC (Cin < 32) is a property of the layer.
// v_mfma 16x16x32 instead of 32x32x16 for the large tiles (higher held clock under matrix-dense load);
// DAVO_H3_M16=0 selects the 32x32x16 form (A/B measurements)
bool h3_use_m16() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DAVO_H3_M16"); v = e ? atoi(e) : 1; }
    return v != 0;
}

template <int KS, int STRIDE, int WM, int WN, int TM, int TN, int LAYER, bool SMALLC, bool M16, int NSTG = 2>
hipError_t launch_h3_m(const ConvParamsH& p, dim3 grid, hipStream_t s) {
    static bool attr_set = false;
    using T = TileH<WM, WN, TM, TN, NSTG>;
    auto kern = conv_igemm_h3<KS, STRIDE, WM, WN, TM, TN, LAYER, true, SMALLC, M16, NSTG>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES_DMA);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(T::THREADS), T::LDS_BYTES_DMA, s, p);
    return hipGetLastError();
}

// All f16x3 launches are LDS-DMA staged.  SMALLC (Cin < 32) is a property of the layer.  The 16x16x32
// form is used for EVERY tile shape of cnv3..cnv7, so that an output element is summed in the same order
// whatever tile the launch plan gives it (batch-size invariance to the bit).
template <int KS, int STRIDE, int WM, int WN, int TM, int TN, int LAYER, bool SMALLC>
hipError_t launch_h3_c(const ConvParamsH& p, dim3 grid, hipStream_t s) {
    if constexpr (LAYER >= 3 && !SMALLC) {
        if (h3_use_m16()) return launch_h3_m<KS, STRIDE, WM, WN, TM, TN, LAYER, SMALLC, true>(p, grid, s);
    }
    return launch_h3_m<KS, STRIDE, WM, WN, TM, TN, LAYER, SMALLC, false>(p, grid, s);
}

// MAXBN bounds the instantiations to the N tiles a layer can use (its padded Cout)
template <int KS, int STRIDE, int LAYER, bool SMALLC, int MAXBN>
hipError_t launch_h3_tile(int tile, const ConvParamsH& p, dim3 grid, hipStream_t s) {
    if (tile == TILE_128x32) return launch_h3_c<KS, STRIDE, 4, 1, 1, 1, LAYER, SMALLC>(p, grid, s);
    if constexpr (MAXBN >= 64)
        if (tile == TILE_256x64) return launch_h3_c<KS, STRIDE, 4, 2, 2, 1, LAYER, SMALLC>(p, grid, s);
    if constexpr (MAXBN >= 128) {
        if (tile == TILE_256x128) return launch_h3_c<KS, STRIDE, 4, 2, 2, 2, LAYER, SMALLC>(p, grid, s);
        if (tile == TILE_128x128) {
            // cnv7 (stride 2, pose head in the epilogue): four waves of 64x64 measured 7 % faster than eight of
            // 32x64 (fewer LDS fragment reads per MFMA); the stride-1 layers measured the other way round
            if constexpr (LAYER == 7) return launch_h3_c<KS, STRIDE, 2, 2, 2, 2, LAYER, SMALLC>(p, grid, s);
            else {
                // a launch of at most one workgroup per CU (remainder rows) has no second workgroup to hide the
                // DMA latency behind: three ring slots instead of two (cnv6.rem 0.073 -> 0.061 ms, cnv5.rem 0.041 -> 0.036)
                if constexpr (LAYER >= 3 && !SMALLC)
                    if ((long)grid.x * grid.y <= 256 && h3_use_m16())
                        return launch_h3_m<KS, STRIDE, 4, 2, 1, 2, LAYER, SMALLC, true, DAVO_REM_STAGES>(p, grid, s);
                return launch_h3_c<KS, STRIDE, 4, 2, 1, 2, LAYER, SMALLC>(p, grid, s);
            }
        }
    }
    if constexpr (MAXBN >= 256 && LAYER != 0) {
        if (tile == TILE_128x256) return launch_h3_c<KS, STRIDE, 2, 4, 2, 2, LAYER, SMALLC>(p, grid, s);
        if (tile == TILE_256x256) return launch_h3_c<KS, STRIDE, 4, 2, 2, 4, LAYER, SMALLC>(p, grid, s);
    }
    if constexpr (MAXBN >= 256 && LAYER == 0)
        if (tile == TILE_128x256) return launch_h3_c<KS, STRIDE, 2, 4, 2, 2, LAYER, SMALLC>(p, grid, s);
    return hipErrorInvalidValue;
}

// generic shapes (davo_conv2d_same): SMALLC follows the channel-block size
template <int KS, int STRIDE>
hipError_t launch_h3_generic(int tile, const ConvParamsH& p, dim3 grid, hipStream_t s) {
    return p.cb_log2 < 5 ? launch_h3_tile<KS, STRIDE, 0, true, 256>(tile, p, grid, s)
                         : launch_h3_tile<KS, STRIDE, 0, false, 256>(tile, p, grid, s);
}

hipError_t launch_layer_h3(int layer, int tile, const ConvParamsH& p, dim3 grid, hipStream_t s) {
    switch (layer) {
        case 0: return launch_h3_tile<7, 2, 1, true, 32>(tile, p, grid, s);
        case 1: return launch_h3_tile<5, 2, 2, true, 32>(tile, p, grid, s);
        case 2: return launch_h3_tile<3, 1, 3, false, 64>(tile, p, grid, s);
        case 3: return launch_h3_tile<3, 1, 4, false, 128>(tile, p, grid, s);
        case 4: return launch_h3_tile<3, 1, 5, false, 256>(tile, p, grid, s);
        case 5: return launch_h3_tile<3, 1, 6, false, 256>(tile, p, grid, s);
        case 6: return launch_h3_tile<3, 2, 7, false, 256>(tile, p, grid, s);
    }
    return hipErrorInvalidValue;
}

inline void split_f16(float v, _Float16* hi, _Float16* lo) {
    const _Float16 h = (_Float16)v;
    *hi = h;
    *lo = (_Float16)(v - (float)h);
}

// power of two that moves max|w| into [128, 256): keeps the fp16 residuals of small weights normal
float weight_prescale(const float* w, size_t n) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) m = std::fmax(m, std::fabs(w[i]));
    if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
    int e;
    std::frexp(m, &e);                 // m = f * 2^e, f in [0.5, 1)
    return std::ldexp(1.0f, 8 - e);    // m * scale in [128, 256)
}

// HWIO float32 [KS,KS,Cin_tf,Cout] -> [npad][nchunks][32 hi | 32 lo] halves in the kernel's k
// order: chunk q = (channel block cblk, tap group tq), element e -> tap = tq*tpc + e/CB,
// channel = cblk*CB + e%CB.
void pack_conv_weights_h3(const float* w_tf, int KS, int cin_tf, int cout, const int* chmap, int cin_packed,
                          int cb_log2, int tpc_log2, int cpb, int nchunks, float scale, _Float16* out /*zeroed*/) {
    const int cb = 1 << cb_log2, ntaps = KS * KS;
    for (int q = 0; q < nchunks; ++q) {
        const int cblk = q / cpb, tq = q % cpb;
        for (int e = 0; e < 32; ++e) {
            const int tap = (tq << tpc_log2) + (e >> cb_log2);
            const int cp = cblk * cb + (e & (cb - 1));
            if (tap >= ntaps || cp >= cin_packed) continue;
            const int ci = chmap ? chmap[cp] : cp;
            if (ci < 0 || ci >= cin_tf) continue;
            const float* src = w_tf + ((size_t)tap * cin_tf + ci) * cout;
            for (int n = 0; n < cout; ++n) {
                _Float16* o = out + ((size_t)n * nchunks + q) * 64;
                split_f16(src[n] * scale, o + e, o + 32 + e);
            }
        }
    }
}

int pick_bn(int cout) { return cout <= 32 ? 32 : (cout % 128 == 0 ? 128 : (cout <= 64 ? 64 : 128)); }

// Re-lay-out HWIO weights [KS,KS,Cin_tf,Cout] -> Wp[npad][kpad], k = tap*cin_packed + c, where
// packed channel c reads TF input channel chmap[c] (or nothing: -1).  Zero padded.
void pack_conv_weights(const float* w_tf, int KS, int cin_tf, int cout, const int* chmap, int cin_packed,
                       int npad, int kpad, float* out /*npad*kpad, zeroed*/) {
    for (int tap = 0; tap < KS * KS; ++tap)
        for (int c = 0; c < cin_packed; ++c) {
            const int ci = chmap ? chmap[c] : c;
            if (ci < 0 || ci >= cin_tf) continue;
            const float* src = w_tf + ((size_t)tap * cin_tf + ci) * cout;
            const size_t k = (size_t)tap * cin_packed + c;
            for (int n = 0; n < cout; ++n) out[(size_t)n * kpad + k] = src[n];
        }
}

// ---- profiling --------------------------------------------------------------------------
struct ProfScope {
    davo_ctx* c;
    ProfEntry* e = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(davo_ctx* ctx, const char* name) : c(ctx) {
        if (!c->prof) return;
        if (c->prof_dominant_only && strcmp(name, "cnv6") != 0) return;
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
    ~ProfScope() {
        if (!e) return;
        if (b) (void)hipEventRecord(b, c->stream);
        if (a && b) e->pending.emplace_back(a, b);
    }
};

int sync_all_slots(davo_ctx* c);

int prof_collect(davo_ctx* c) {
    { int rc = sync_all_slots(c); if (rc) return rc; }
    for (auto& pe : c->prof_entries) {
        for (auto& ab : pe.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ab.first, ab.second) == hipSuccess) {
                pe.total_ms += ms;
                pe.launches += 1;
            }
            c->event_pool.push_back(ab.first);
            c->event_pool.push_back(ab.second);
        }
        pe.pending.clear();
    }
    return DAVO_OK;
}

// ---- weights ----------------------------------------------------------------------------
std::vector<std::string> needed_names(const Variant& v) {
    std::vector<std::string> n;
    const char* trunk[] = {"cnv1", "cnv2", "cnv3", "cnv4", "cnv5"};
    for (auto l : trunk) {
        n.push_back(std::string("pose_exp_net/") + l + "/weights");
        n.push_back(std::string("pose_exp_net/") + l + "/biases");
    }
    const char* heads[] = {"rotation", "translation"};
    const char* hl[] = {"cnv6", "cnv7", "pred"};
    for (auto h : heads)
        for (auto l : hl) {
            n.push_back(std::string("pose_exp_net/pose/") + h + "/" + l + "/weights");
            n.push_back(std::string("pose_exp_net/pose/") + h + "/" + l + "/biases");
        }
    if (v.att_source == 1) {
        n.push_back("pose_exp_net/se_flow/bottleneck_fc/kernel");
        n.push_back("pose_exp_net/se_flow/bottleneck_fc/bias");
        n.push_back("pose_exp_net/se_flow/recover_fc/kernel");
        n.push_back("pose_exp_net/se_flow/recover_fc/bias");
    } else if (v.att_source == 2 || v.att_source == 3) {
        n.push_back("pose_exp_net/pose_exp_net/seg_channel_weight/weight");
    }
    return n;
}

bool expected_shape(const davo_ctx* c, const std::string& name, std::vector<int64_t>* sh) {
    const int c10 = 2 * c->v.cin_per_frame, c6 = c->v.cnv6_out;
    auto is = [&](const char* s) { return name == s; };
    auto ends = [&](const char* s) {
        const size_t n = strlen(s);
        return name.size() >= n && name.compare(name.size() - n, n, s) == 0;
    };
    if (is("pose_exp_net/cnv1/weights")) *sh = {7, 7, c10, 16};
    else if (is("pose_exp_net/cnv1/biases")) *sh = {16};
    else if (is("pose_exp_net/cnv2/weights")) *sh = {5, 5, 16, 32};
    else if (is("pose_exp_net/cnv2/biases")) *sh = {32};
    else if (is("pose_exp_net/cnv3/weights")) *sh = {3, 3, 32, 64};
    else if (is("pose_exp_net/cnv3/biases")) *sh = {64};
    else if (is("pose_exp_net/cnv4/weights")) *sh = {3, 3, 64, 128};
    else if (is("pose_exp_net/cnv4/biases")) *sh = {128};
    else if (is("pose_exp_net/cnv5/weights")) *sh = {3, 3, 128, 256};
    else if (is("pose_exp_net/cnv5/biases")) *sh = {256};
    else if (ends("/cnv6/weights")) *sh = {3, 3, 256, c6};
    else if (ends("/cnv6/biases")) *sh = {c6};
    else if (ends("/cnv7/weights")) *sh = {3, 3, c6, 256};
    else if (ends("/cnv7/biases")) *sh = {256};
    else if (ends("/pred/weights")) *sh = {1, 1, 256, 3};
    else if (ends("/pred/biases")) *sh = {3};
    else if (is("pose_exp_net/se_flow/bottleneck_fc/kernel")) *sh = {2, 8};
    else if (is("pose_exp_net/se_flow/bottleneck_fc/bias")) *sh = {8};
    else if (is("pose_exp_net/se_flow/recover_fc/kernel")) *sh = {8, NCLS};
    else if (is("pose_exp_net/se_flow/recover_fc/bias")) *sh = {NCLS};
    else if (is("pose_exp_net/pose_exp_net/seg_channel_weight/weight")) *sh = {NCLS};
    else return false;
    return true;
}

int upload(davo_ctx* c, const std::vector<float>& host, float** dev) {
    if (*dev) { HIP_TRY(c, hipFree(*dev)); *dev = nullptr; }
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(dev), host.size() * sizeof(float)));
    HIP_TRY(c, hipMemcpy(*dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    return DAVO_OK;
}

void init_layer(ConvLayer& L, const char* label, int KS, int stride, int rate, int cin, int cout, int groups) {
    L.label = label; L.KS = KS; L.stride = stride; L.rate = rate;
    L.cin = cin; L.cin_log2 = ilog2_exact(cin); L.cout = cout; L.groups = groups;
    L.BN = pick_bn(cout);
    L.npad = (cout + L.BN - 1) / L.BN * L.BN;
    L.kpad = (KS * KS * cin + BK - 1) / BK * BK;
    L.nchunks = L.kpad / BK;
    const int cb = cin < 32 ? cin : 32;
    L.cb_log2 = ilog2_exact(cb);
    L.tpc_log2 = ilog2_exact(32 / cb);
    L.cpb = (KS * KS + (32 / cb) - 1) / (32 / cb);
    L.nchunks_h = (cin / cb) * L.cpb;
    L.tile_h = -1;                                           // -1: the planner picks per launch
    if (const char* e = getenv("DAVO_H3_TILE")) {            // measurement only: force a tile where it fits
        const int t = atoi(e);
        const char* only = getenv("DAVO_H3_TILE_LABEL");     // restrict the force to one layer ("cnv4")
        if (t >= 0 && t < NUM_TILES && cout >= tile_shape(t).bn && (!only || strcmp(only, L.label) == 0)) L.tile_h = t;
    }
    const int gran = cout > 128 ? 256 : cout > 64 ? 128 : cout > 32 ? 64 : 32;    // widest N tile a launch may use
    L.npad_h = (cout + gran - 1) / gran * gran;
}

int build_packed_weights(davo_ctx* c) {
    const int c6 = c->v.cnv6_out, cpf = c->v.cin_per_frame;
    auto W = [&](const std::string& n) -> const HostTensor& { return c->weights.at(n); };
    // cnv1: packed input channel -> TF input channel.  v1: [t r,g,b | 0,0 | s r,g,b | fx,fy];
    // the two zero channels (davo.py:979,1065) are dropped.  v0: [t rgb | s rgb].
    int chmap1[8];
    if (cpf == 5) { const int m[8] = {0, 1, 2, 5, 6, 7, 8, 9}; memcpy(chmap1, m, sizeof m); }
    else { const int m[8] = {0, 1, 2, 3, 4, 5, -1, -1}; memcpy(chmap1, m, sizeof m); }

    struct Src { const char* w; const char* b; const int* chmap; int cin_tf; };
    const Src trunk[5] = {{"pose_exp_net/cnv1/weights", "pose_exp_net/cnv1/biases", chmap1, 2 * cpf},
                          {"pose_exp_net/cnv2/weights", "pose_exp_net/cnv2/biases", nullptr, 16},
                          {"pose_exp_net/cnv3/weights", "pose_exp_net/cnv3/biases", nullptr, 32},
                          {"pose_exp_net/cnv4/weights", "pose_exp_net/cnv4/biases", nullptr, 64},
                          {"pose_exp_net/cnv5/weights", "pose_exp_net/cnv5/biases", nullptr, 128}};
    for (int i = 0; i < 5; ++i) {
        ConvLayer& L = c->L[i];
        std::vector<float> wp((size_t)L.npad * L.kpad, 0.f), bp(L.npad, 0.f);
        pack_conv_weights(W(trunk[i].w).data.data(), L.KS, trunk[i].cin_tf, L.cout, trunk[i].chmap, L.cin,
                          L.npad, L.kpad, wp.data());
        memcpy(bp.data(), W(trunk[i].b).data.data(), L.cout * sizeof(float));
        int rc = upload(c, wp, &L.d_w); if (rc) return rc;
        rc = upload(c, bp, &L.d_b); if (rc) return rc;
    }
    const char* heads[2] = {"rotation", "translation"};
    {   // cnv6: one GEMM, N = [rotation c6 | translation c6]
        ConvLayer& L = c->L[5];
        std::vector<float> wp((size_t)L.npad * L.kpad, 0.f), bp(L.npad, 0.f);
        for (int h = 0; h < 2; ++h) {
            const std::string p = std::string("pose_exp_net/pose/") + heads[h] + "/cnv6/";
            pack_conv_weights(W(p + "weights").data.data(), 3, 256, c6, nullptr, 256, c6, L.kpad,
                              wp.data() + (size_t)h * c6 * L.kpad);
            memcpy(bp.data() + h * c6, W(p + "biases").data.data(), c6 * sizeof(float));
        }
        int rc = upload(c, wp, &L.d_w); if (rc) return rc;
        rc = upload(c, bp, &L.d_b); if (rc) return rc;
    }
    {   // cnv7: two groups (blockIdx.y), group g reads cnv6 channels [g*c6, (g+1)*c6)
        ConvLayer& L = c->L[6];
        std::vector<float> wp((size_t)2 * L.npad * L.kpad, 0.f), bp((size_t)2 * L.npad, 0.f);
        for (int h = 0; h < 2; ++h) {
            const std::string p = std::string("pose_exp_net/pose/") + heads[h] + "/cnv7/";
            pack_conv_weights(W(p + "weights").data.data(), 3, c6, 256, nullptr, c6, L.npad, L.kpad,
                              wp.data() + (size_t)h * L.npad * L.kpad);
            memcpy(bp.data() + (size_t)h * L.npad, W(p + "biases").data.data(), 256 * sizeof(float));
        }
        int rc = upload(c, wp, &L.d_w); if (rc) return rc;
        rc = upload(c, bp, &L.d_b); if (rc) return rc;
    }
    {   // pred: [2][256][3] + [2][3]
        std::vector<float> wp(2 * 256 * 3), bp(2 * 3);
        for (int h = 0; h < 2; ++h) {
            const std::string p = std::string("pose_exp_net/pose/") + heads[h] + "/pred/";
            memcpy(wp.data() + h * 768, W(p + "weights").data.data(), 768 * sizeof(float));
            memcpy(bp.data() + h * 3, W(p + "biases").data.data(), 3 * sizeof(float));
        }
        int rc = upload(c, wp, &c->d_wpred); if (rc) return rc;
        rc = upload(c, bp, &c->d_bpred); if (rc) return rc;
    }
    c->packed_ready = true;
    return DAVO_OK;
}

int upload_bytes(davo_ctx* c, const void* host, size_t bytes, void** dev) {
    if (*dev) { HIP_TRY(c, hipFree(*dev)); *dev = nullptr; }
    HIP_TRY(c, hipMalloc(dev, bytes));
    HIP_TRY(c, hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return DAVO_OK;
}

// split-fp16 weights for the f16x3 path, same layer structure as build_packed_weights
int build_packed_weights_h3(davo_ctx* c) {
    const int c6 = c->v.cnv6_out, cpf = c->v.cin_per_frame;
    auto W = [&](const std::string& n) -> const HostTensor& { return c->weights.at(n); };
    int chmap1[8];
    if (cpf == 5) { const int m[8] = {0, 1, 2, 5, 6, 7, 8, 9}; memcpy(chmap1, m, sizeof m); }
    else { const int m[8] = {0, 1, 2, 3, 4, 5, -1, -1}; memcpy(chmap1, m, sizeof m); }
    const char* heads[2] = {"rotation", "translation"};
    for (int li = 0; li < 7; ++li) {
        ConvLayer& L = c->L[li];
        const size_t per_group = (size_t)L.npad_h * L.nchunks_h * 64;
        std::vector<_Float16> wp(per_group * L.groups, (_Float16)0.0f);
        std::vector<float> bp((size_t)L.npad_h * L.groups, 0.f);
        {   // one power-of-two scale per layer (both heads share the launch)
            float sc = 1e30f;
            if (li < 5) {
                const char* names[5] = {"cnv1", "cnv2", "cnv3", "cnv4", "cnv5"};
                const HostTensor& t = W(std::string("pose_exp_net/") + names[li] + "/weights");
                sc = weight_prescale(t.data.data(), t.data.size());
            } else {
                for (int h = 0; h < 2; ++h) {
                    const HostTensor& t = W(std::string("pose_exp_net/pose/") + heads[h] + (li == 5 ? "/cnv6/weights" : "/cnv7/weights"));
                    sc = std::fmin(sc, weight_prescale(t.data.data(), t.data.size()));
                }
            }
            L.wscale = sc;
        }
        auto pack = [&](const std::string& wname, const std::string& bname, int cin_tf, int cout, const int* chmap,
                        _Float16* wdst, float* bdst) {
            pack_conv_weights_h3(W(wname).data.data(), L.KS, cin_tf, cout, chmap, L.cin, L.cb_log2, L.tpc_log2, L.cpb,
                                 L.nchunks_h, L.wscale, wdst);
            memcpy(bdst, W(bname).data.data(), cout * sizeof(float));
        };
        if (li < 5) {
            const char* names[5] = {"cnv1", "cnv2", "cnv3", "cnv4", "cnv5"};
            const std::string p = std::string("pose_exp_net/") + names[li] + "/";
            pack(p + "weights", p + "biases", li == 0 ? 2 * cpf : L.cin, L.cout, li == 0 ? chmap1 : nullptr, wp.data(), bp.data());
        } else if (li == 5) {       // rotation | translation stacked along N
            for (int h = 0; h < 2; ++h) {
                const std::string p = std::string("pose_exp_net/pose/") + heads[h] + "/cnv6/";
                pack(p + "weights", p + "biases", 256, c6, nullptr, wp.data() + (size_t)h * c6 * L.nchunks_h * 64, bp.data() + h * c6);
            }
        } else {                    // cnv7: one group per head
            for (int h = 0; h < 2; ++h) {
                const std::string p = std::string("pose_exp_net/pose/") + heads[h] + "/cnv7/";
                pack(p + "weights", p + "biases", c6, 256, nullptr, wp.data() + (size_t)h * per_group, bp.data() + (size_t)h * L.npad_h);
            }
        }
        int rc = upload_bytes(c, wp.data(), wp.size() * sizeof(_Float16), reinterpret_cast<void**>(&L.d_wh));
        if (rc) return rc;
        rc = upload(c, bp, &L.d_bh);
        if (rc) return rc;
    }
    {   // cnv1 patch kernel: [14 steps][hi|lo][64 lanes][8 channels] halves, lane = (n = l&15, tap slot kq = l>>4)
        const ConvLayer& L = c->L[0];
        const HostTensor& t = W("pose_exp_net/cnv1/weights");
        const int cin_tf = 2 * cpf;
        std::vector<_Float16> wp((size_t)cp1::STEPS * 2 * 64 * 8, (_Float16)0.0f);
        for (int step = 0; step < cp1::STEPS; ++step)
            for (int l = 0; l < 64; ++l) {
                const int n = l & 15, kq = l >> 4, ky = step >> 1, kx = 4 * (step & 1) + kq;
                if (kx >= 7) continue;
                for (int j = 0; j < 8; ++j) {
                    const int ci = chmap1[j];
                    if (ci < 0 || ci >= cin_tf) continue;
                    const float v = t.data[(((size_t)ky * 7 + kx) * cin_tf + ci) * 16 + n] * L.wscale;
                    split_f16(v, &wp[((size_t)(step * 2 + 0) * 64 + l) * 8 + j], &wp[((size_t)(step * 2 + 1) * 64 + l) * 8 + j]);
                }
            }
        int rc = upload_bytes(c, wp.data(), wp.size() * sizeof(_Float16), reinterpret_cast<void**>(&c->d_w1patch));
        if (rc) return rc;
    }
    c->packed_h_ready = true;
    return DAVO_OK;
}

int missing_weights(davo_ctx* c, std::string* names) {
    int n = 0;
    for (auto& nm : c->needed)
        if (!c->weights.count(nm)) {
            ++n;
            if (names) { if (!names->empty()) *names += ", "; *names += nm; }
        }
    return n;
}

// ---- the forward plan -------------------------------------------------------------------
int run_conv_layer(davo_ctx* c, int li, const float* x, int x_ld, int Hin, int Win, float* y, int y_ld,
                   int NB) {
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
    const std::vector<Launch> plan = plan_layer(mtiles, L.npad, L.groups);
    c->last_plan[li][0] = c->last_plan[li][1] = 0;
    for (size_t i = 0; i < plan.size(); ++i) {
        p.mtile0 = plan[i].mtile0;
        p.ntiles_n = L.npad / plan[i].BN;
        dim3 grid(plan[i].mtiles * p.ntiles_n, L.groups);
        const std::string label = i == 0 ? std::string(L.label) : std::string(L.label) + ".rem";
        ProfScope ps(c, label.c_str());
        HIP_TRY(c, launch_layer(li, plan[i].BN, p, grid, c->stream));
        c->last_plan[li][i] = plan[i].mtiles * 1000 + plan[i].BN;
    }
    return DAVO_OK;
}

// ---- f16x3 launch planning -------------------------------------------------------------------
// Same idea as plan_layer: whole rounds of the most efficient tile, then a remainder launch with a
// smaller tile that again fills whole rounds.  Costs are in units of one round of 256x256 tiles;
// eff = measured throughput of the tile relative to 256x256 at full occupancy (B=128: cnv5, cnv6, cnv7 forced to
// one tile shape each, re-measured after the matrix loop was software-pipelined).  The two narrow tiles keep the
// figures fitted on the K < 600 layers that use them (cnv3, cnv4: A/B on one box, profiles/ r01f notes).
struct TileInfo { int id, per_cu; double eff; };
TileInfo kTiles[] = {{TILE_256x256, 1, 1.00}, {TILE_128x256, 1, 0.84}, {TILE_256x128, 1, 0.84},
                     {TILE_128x128, 2, 0.87}, {TILE_256x64, 1, 0.62}, {TILE_128x32, 3, 0.40}};
// measurement only: DAVO_H3_EFF="e0,e1,e2,e3,e4,e5[,p4]" overrides the efficiencies (table order) and 256x64's per_cu
void tiles_from_env() {
    static bool done = false;
    if (done) return;
    done = true;
    const char* e = getenv("DAVO_H3_EFF");
    if (!e) return;
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    const int n = sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf,%lf", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6);
    for (int i = 0; i < 6 && i < n; ++i) if (v[i] > 0) kTiles[i].eff = v[i];
    if (n >= 7 && v[6] >= 1) kTiles[4].per_cu = (int)v[6];
}
struct LaunchH { int row0, rows, tile; };

// whole rounds run per_cu workgroups per CU side by side; in the last, partial round a CU holds
// ceil(rest / 256) of them (the dispatcher spreads a short tail one per CU)
double h3_cost(const TileInfo& t, long ntiles) {
    const TileShape ts = tile_shape(t.id);
    const long slots = 256L * t.per_cu;
    const double one = (ts.bm * ts.bn / 65536.0) / t.eff;
    const long full = ntiles / slots, rest = ntiles % slots;
    return (double)full * t.per_cu * one + (double)((rest + 255) / 256) * one;
}

std::vector<LaunchH> plan_layer_h3(int M, int npad, int groups, int forced_tile) {
    tiles_from_env();
    auto ntiles = [&](const TileInfo& t, int rows) {
        const TileShape ts = tile_shape(t.id);
        return (long)((rows + ts.bm - 1) / ts.bm) * (npad / ts.bn) * groups;
    };
    auto fits = [&](const TileInfo& t) { const int bn = tile_shape(t.id).bn; return bn <= npad && npad % bn == 0; };
    if (forced_tile >= 0) return {{0, M, forced_tile}};
    std::vector<LaunchH> best;
    double best_cost = 1e30;
    for (const TileInfo& t1 : kTiles) {
        if (!fits(t1)) continue;
        const double c1 = h3_cost(t1, ntiles(t1, M));
        if (c1 < best_cost - 1e-9) { best_cost = c1; best = {{0, M, t1.id}}; }
        const TileShape s1 = tile_shape(t1.id);
        const long per_round = 256L * t1.per_cu, per_m = (long)(npad / s1.bn) * groups;
        if (per_round % per_m) continue;
        const long m_per_round = per_round / per_m;                       // M tiles of t1 in one round
        const long rounds = ((long)M / s1.bm) / m_per_round;
        int rows1 = (int)(rounds * m_per_round * s1.bm);
        rows1 -= rows1 % 256;                                             // every tile height divides 256
        if (rows1 <= 0 || rows1 >= M) continue;
        const double cm = h3_cost(t1, ntiles(t1, rows1));
        static const int rem_force = getenv("DAVO_H3_REM_TILE") ? atoi(getenv("DAVO_H3_REM_TILE")) : -1;
        for (const TileInfo& t2 : kTiles) {
            if (!fits(t2)) continue;
            if (rem_force >= 0 && t2.id != rem_force) continue;
            const double c = cm + h3_cost(t2, ntiles(t2, M - rows1)) + 0.01;
            if (c < best_cost - 1e-9) { best_cost = c; best = {{0, rows1, t1.id}, {rows1, M - rows1, t2.id}}; }
        }
    }
    return best;
}

// f16x3 launch of conv layer li: x and y are split-fp16 blocked tensors (y float32 when y_f32)
int run_conv_layer_h3(davo_ctx* c, int li, const void* x, int x_ch, int Hin, int Win, void* y, int y_ld,
                      bool y_f32, int NB, bool fuse_pose = false, int* pose_bm = nullptr, int* pose_mt = nullptr,
                      int* pose_ntn = nullptr) {
    const ConvLayer& L = c->L[li];
    ConvParamsH p{};
    int Ho, Wo, pt, pl;
    same_pad(Hin, L.KS, L.stride, L.rate, &Ho, &pt);
    same_pad(Win, L.KS, L.stride, L.rate, &Wo, &pl);
    p.x = static_cast<const uint8_t*>(x); p.w = L.d_wh; p.bias = L.d_bh; p.y = static_cast<uint8_t*>(y);
    p.zeros = reinterpret_cast<const uint8_t*>(c->d_zeros);
    p.Hin = Hin; p.Win = Win; p.Hout = Ho; p.Wout = Wo;
    p.x_pix_bytes = (long)x_ch * 4; p.x_boff = 0;
    p.cb_log2 = L.cb_log2; p.tpc_log2 = L.tpc_log2; p.cpb = L.cpb; p.nchunks = L.nchunks_h;
    p.w_row_bytes = (long)L.nchunks_h * 128;
    p.y_mode = y_f32 ? 0 : 1; p.y_ld = y_ld; p.y_coff = 0; p.Cout = L.cout;
    p.pad_t = pt; p.pad_l = pl; p.rate = L.rate;
    p.M = NB * Ho * Wo; p.ntaps = L.KS * L.KS; p.mtile0 = 0; p.relu = 1;
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
    if (const char* e = getenv("DAVO_DBG")) p.dbg = atoi(e);
    std::vector<LaunchH> plan = plan_layer_h3(p.M, L.npad_h, L.groups, L.tile_h);
    if (fuse_pose) {      // one launch, one tile shape no taller than an image, so a tile touches <= 2 images
        const int P = Ho * Wo;
        tiles_from_env();
        int best = -1; double bc = 1e30;
        for (const TileInfo& t : kTiles) {
            const TileShape ts = tile_shape(t.id);
            if (ts.bn > L.npad_h || L.npad_h % ts.bn || ts.bm > P) continue;
            if (L.tile_h >= 0 && t.id != L.tile_h) continue;             // measurement only (DAVO_H3_TILE)
            const double cst = h3_cost(t, (long)((p.M + ts.bm - 1) / ts.bm) * (L.npad_h / ts.bn) * L.groups);
            if (cst < bc) { bc = cst; best = t.id; }
        }
        if (best < 0) return fail(c, DAVO_ERR_INVALID, "no tile fits the fused pose head");
        plan = {{0, p.M, best}};
        const TileShape ts = tile_shape(best);
        const int mt = (p.M + ts.bm - 1) / ts.bm, ntn = L.npad_h / ts.bn;
        const size_t need = (size_t)L.groups * mt * ntn * 6;
        if (need > c->pose_tiles_floats) {
            if (c->d_pose_tiles) { int rs = sync_all_slots(c); if (rs) return rs; HIP_TRY(c, hipFree(c->d_pose_tiles)); c->d_pose_tiles = nullptr; }
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_pose_tiles), need * sizeof(float) * 4));   // x4: one region per in-flight slot
            c->pose_tiles_floats = need;
        }
        const int slot_idx = (c->next_slot + c->inflight - 1) % c->inflight;      // the slot this batch runs in
        p.y_mode = 2; p.pose_w = c->d_wpred; p.pose_partial = c->d_pose_tiles + (size_t)slot_idx * c->pose_tiles_floats;
        p.pose_P = P; p.pose_mt = mt;
        if (pose_bm) *pose_bm = ts.bm;
        if (pose_mt) *pose_mt = mt;
        if (pose_ntn) *pose_ntn = ntn;
    }
    c->last_plan[li][0] = c->last_plan[li][1] = 0;
    for (size_t i = 0; i < plan.size() && i < 2; ++i) {
        const TileShape ts = tile_shape(plan[i].tile);
        p.ntiles_n = L.npad_h / ts.bn;
        p.mtile0 = plan[i].row0 / ts.bm;
        const int full_m = p.M;
        p.M = plan[i].row0 + plan[i].rows;                    // rows past this launch's range are not its job
        const int mtiles = (plan[i].rows + ts.bm - 1) / ts.bm;
        dim3 grid(mtiles * p.ntiles_n, L.groups);
        c->last_plan[li][i] = ((plan[i].rows + 127) / 128) * 1000 + plan[i].tile;
        const std::string label = i == 0 ? std::string(L.label) : std::string(L.label) + ".rem";
        {
            ProfScope ps(c, label.c_str());
            HIP_TRY(c, launch_layer_h3(li, plan[i].tile, p, grid, c->stream));
        }
        p.M = full_m;
    }
    return DAVO_OK;
}

// cnv1 of the f16x3 path from an LDS-staged input patch (conv_patch_h3.h).  fused: the patch is built
// from the raw inputs (mask + pack fused in); otherwise it is copied from the packed tensor.
int run_cnv1_patch(davo_ctx* c, bool fused, const void* d_img, const void* d_flow, const void* d_seg, void* y, int NB) {
    static bool attr_set = false;
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
    if (!attr_set) {
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_patch_cnv1_h3<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, cp1::LDS_BYTES));
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_patch_cnv1_h3<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, cp1::LDS_BYTES));
        attr_set = true;
    }
    c->last_plan[0][0] = ((NB * Ho * Wo + 127) / 128) * 1000 + 99; c->last_plan[0][1] = 0;
    const int nblk = p.ntiles < 512 ? p.ntiles : 512;          // 2 workgroups per CU, each walks its tiles
    ProfScope ps(c, "cnv1");
    if (fused) hipLaunchKernelGGL(conv_patch_cnv1_h3<true>, dim3(nblk), dim3(cp1::THREADS), cp1::LDS_BYTES, c->stream, p);
    else hipLaunchKernelGGL(conv_patch_cnv1_h3<false>, dim3(nblk), dim3(cp1::THREADS), cp1::LDS_BYTES, c->stream, p);
    HIP_TRY(c, hipGetLastError());
    return DAVO_OK;
}

int run_direct(davo_ctx* c, const char* label, const float* x, int N, int Hin, int Win, int cin, int x_ld,
               int x_coff, const std::string& wname, const std::string& bname, int KS, int cout, int stride,
               int rate, float* y, int y_ld, int y_coff) {
    int Ho, Wo, pt, pl;
    same_pad(Hin, KS, stride, rate, &Ho, &pt);
    same_pad(Win, KS, stride, rate, &Wo, &pl);
    const long total = (long)N * Ho * Wo * cout;
    ProfScope ps(c, label);
    hipLaunchKernelGGL(conv_direct, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, x, N, Hin,
                       Win, cin, x_ld, x_coff, c->weights.at(wname).dev, KS, cout, c->weights.at(bname).dev,
                       stride, rate, pt, pl, Ho, Wo, 1, y, y_ld, y_coff);
    HIP_TRY(c, hipGetLastError());
    return DAVO_OK;
}

int forward_device(davo_ctx* c, int B, const void* d_img, const void* d_flow, const void* d_seg, void* d_pose) {
    if (B < 1 || B > c->max_batch) return fail(c, DAVO_ERR_INVALID, "batch %d outside [1,%d]", B, c->max_batch);
    if (!d_img || !d_flow || !d_seg || !d_pose) return fail(c, DAVO_ERR_INVALID, "null device pointer");
    {
        std::string names;
        if (missing_weights(c, &names)) return fail(c, DAVO_ERR_NOT_READY, "weights not loaded: %s", names.c_str());
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->packed_ready) { int rc = build_packed_weights(c); if (rc) return rc; }
    const bool h3 = c->impl == 0 && c->precision == 1;
    if (h3 && !c->packed_h_ready) { int rc = build_packed_weights_h3(c); if (rc) return rc; }

    const int H = c->H, W = c->W, HW = H * W, NB = 2 * B;
    const Variant& v = c->v;
    hipStream_t s = c->stream;
    auto wdev = [&](const char* n) -> const float* {
        auto it = c->weights.find(n);
        return it == c->weights.end() ? nullptr : it->second.dev;
    };
    if (v.att_source == 1) {
        ProfScope ps(c, "se_squeeze_partial");
        hipLaunchKernelGGL(se_squeeze_partial, dim3(SQ_CHUNKS, 2, B), dim3(256), 0, s,
                           static_cast<const float*>(d_flow), HW, v.norm_flow, v.abs_mode, c->d_partial);
        HIP_TRY(c, hipGetLastError());
    }
    {
        ProfScope ps(c, "se_excite");
        hipLaunchKernelGGL(se_excite, dim3(B), dim3(64), 0, s, c->d_partial, HW, v,
                           wdev("pose_exp_net/se_flow/bottleneck_fc/kernel"), wdev("pose_exp_net/se_flow/bottleneck_fc/bias"),
                           wdev("pose_exp_net/se_flow/recover_fc/kernel"), wdev("pose_exp_net/se_flow/recover_fc/bias"),
                           wdev("pose_exp_net/pose_exp_net/seg_channel_weight/weight"), c->d_tab);
        HIP_TRY(c, hipGetLastError());
    }
    const long nthreads = (long)NB * H * (W / 4);
    // f16x3, DAVO_FUSE_PACK=1: cnv1 builds its input patch straight from the raw inputs (mask + pack fused in,
    // the packed tensor never touches HBM).  Measured equal in time to mask_pack + cnv1 (the fused fill is bound
    // by its byte loads), so the two-kernel form stays the default.
    static const bool fuse_env_default = getenv("DAVO_FUSE_PACK") && atoi(getenv("DAVO_FUSE_PACK")) == 1;
    const bool fuse_env = fuse_env_default || c->opt_fuse_pack;
    static const bool patch1 = !(getenv("DAVO_CNV1_PATCH") && atoi(getenv("DAVO_CNV1_PATCH")) == 0);
    const bool fused = h3 && patch1 && fuse_env;
    c->packed_valid = !fused;
    c->last_img = d_img; c->last_flow = d_flow; c->last_seg = d_seg;
    c->packed_ld = c->impl == 0 ? 8 : 10;
    if (!fused) {
        ProfScope ps(c, "mask_pack");
        if (h3)
            hipLaunchKernelGGL(mask_pack<16>, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s,
                               static_cast<const uint8_t*>(d_img), static_cast<const float*>(d_flow),
                               static_cast<const float*>(d_seg), c->d_tab, v, B, H, W, c->d_packed);
        else if (c->impl == 0)
            hipLaunchKernelGGL(mask_pack<8>, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s,
                               static_cast<const uint8_t*>(d_img), static_cast<const float*>(d_flow),
                               static_cast<const float*>(d_seg), c->d_tab, v, B, H, W, c->d_packed);
        else
            hipLaunchKernelGGL(mask_pack<10>, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s,
                               static_cast<const uint8_t*>(d_img), static_cast<const float*>(d_flow),
                               static_cast<const float*>(d_seg), c->d_tab, v, B, H, W, c->d_packed);
        HIP_TRY(c, hipGetLastError());
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
        if ((rc = run_conv_layer_h3(c, 1, a[0], 16, c->H1, c->W1, a[1], 32, false, NB))) return rc;
        if ((rc = run_conv_layer_h3(c, 2, a[1], 32, c->H2, c->W2, a[2], 64, false, NB))) return rc;
        if ((rc = run_conv_layer_h3(c, 3, a[2], 64, c->H2, c->W2, a[3], 128, false, NB))) return rc;
        if ((rc = run_conv_layer_h3(c, 4, a[3], 128, c->H2, c->W2, a[4], 256, false, NB))) return rc;
        if ((rc = run_conv_layer_h3(c, 5, a[4], 256, c->H2, c->W2, a[5], 2 * c6, false, NB))) return rc;
        pose_fused = c->opt_fuse_pose && c->H3 * c->W3 >= 128;
        if ((rc = run_conv_layer_h3(c, 6, a[5], 2 * c6, c->H2, c->W2, a[6], 512, true, NB, pose_fused, &pose_bm, &pose_mt, &pose_ntn))) return rc;
        c->cnv7_valid = !pose_fused;
    } else if (c->impl == 0) {
        if ((rc = run_conv_layer(c, 0, c->d_packed, 8, H, W, a[0], 16, NB))) return rc;
        if ((rc = run_conv_layer(c, 1, a[0], 16, c->H1, c->W1, a[1], 32, NB))) return rc;
        if ((rc = run_conv_layer(c, 2, a[1], 32, c->H2, c->W2, a[2], 64, NB))) return rc;
        if ((rc = run_conv_layer(c, 3, a[2], 64, c->H2, c->W2, a[3], 128, NB))) return rc;
        if ((rc = run_conv_layer(c, 4, a[3], 128, c->H2, c->W2, a[4], 256, NB))) return rc;
        if ((rc = run_conv_layer(c, 5, a[4], 256, c->H2, c->W2, a[5], 2 * c6, NB))) return rc;
        if ((rc = run_conv_layer(c, 6, a[5], 2 * c6, c->H2, c->W2, a[6], 512, NB))) return rc;
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
    if (pose_fused) {
        ProfScope ps(c, "pose_head");
        const int slot_idx = (c->next_slot + c->inflight - 1) % c->inflight;
        hipLaunchKernelGGL(pose_from_tiles, dim3((NB * 6 + 63) / 64), dim3(64), 0, s,
                           c->d_pose_tiles + (size_t)slot_idx * c->pose_tiles_floats, NB, c->H3 * c->W3, pose_bm, pose_mt,
                           pose_ntn, c->d_bpred, static_cast<float*>(d_pose));
        HIP_TRY(c, hipGetLastError());
    } else {
        ProfScope ps(c, "pose_head");
        hipLaunchKernelGGL(pose_head_partial, dim3(PH_SPLIT, NB, 2), dim3(256), 0, s, a[6], c->H3 * c->W3, c->d_wpred,
                           c->d_pose_partial);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL(pose_finish, dim3((NB * 6 + 63) / 64), dim3(64), 0, s, c->d_pose_partial, NB, c->H3 * c->W3,
                           c->d_bpred, static_cast<float*>(d_pose));
        HIP_TRY(c, hipGetLastError());
    }
    c->last_B = B;
    c->last_precision = h3 ? 1 : 0;
    return DAVO_OK;
}

}  // namespace

namespace {

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
    HIP_TRY(c, hipMemset(s->d_partial, 0, (size_t)c->max_batch * 2 * SQ_CHUNKS * 2 * sizeof(float)));
    return DAVO_OK;
}

int sync_all_slots(davo_ctx* c) {
    for (auto& s : c->slots) HIP_TRY(c, hipStreamSynchronize(s.stream));
    if (c->user_stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
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
    HIP_TRY(c, hipMemset(c->d_zeros, 0, 256));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_range), 8 * sizeof(unsigned)));
    HIP_TRY(c, hipMemset(c->d_range, 0, 8 * sizeof(unsigned)));
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
    int rc = upload(c, t.data, &t.dev);
    if (rc) return rc;
    c->packed_ready = false;
    c->packed_h_ready = false;
    return DAVO_OK;
}

int davo_weights_missing(davo_ctx* c) {
    if (!c) return DAVO_ERR_INVALID;
    std::string names;
    const int n = missing_weights(c, &names);
    if (n) c->err = "weights not loaded: " + names;
    return n;
}

int davo_forward_device(davo_ctx* c, int B, const void* d_img, const void* d_flow, const void* d_seg,
                        void* d_pose, float* elapsed_ms) {
    if (!c) return DAVO_ERR_INVALID;
    // rotate through the in-flight slots: this batch runs on its own stream and workspace
    activate_slot(c, c->next_slot);
    c->next_slot = (c->next_slot + 1) % c->inflight;
    if (!elapsed_ms) return forward_device(c, B, d_img, d_flow, d_seg, d_pose);
    HIP_TRY(c, hipSetDevice(c->device));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    HIP_TRY(c, hipEventRecord(e0, c->stream));
    int rc = forward_device(c, B, d_img, d_flow, d_seg, d_pose);
    if (rc == DAVO_OK) {
        HIP_TRY(c, hipEventRecord(e1, c->stream));
        HIP_TRY(c, hipEventSynchronize(e1));
        HIP_TRY(c, hipEventElapsedTime(elapsed_ms, e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// f16x3 only.  Stored activations (fp16 hi/lo pairs) are float32-grade while the layer's largest stored value is
// below the fp16 maximum (above it values were clamped) and not so small that the pairs lose their low bits
// (tools/exp_activation_scale.py: the 1e-4 bar holds down to ~2^-16 of O(1) activations; 2^-11 is the guard).
static int check_range(davo_ctx* c, const unsigned raw[6]) {
    static const char* names[6] = {"cnv1", "cnv2", "cnv3", "cnv4", "cnv5", "cnv6"};
    for (int i = 0; i < 6; ++i) {
        float v;
        memcpy(&v, &raw[i], sizeof v);
        const float actual = ldexpf(v, -c->act_shift[i]);
        if (!(v < 65504.f))
            return fail(c, DAVO_ERR_RANGE, "%s activations reach %.4g: outside the fp16-pair storage range at scale 2^%d "
                        "(values were clamped) - run davo_calibrate() or davo_set_precision(ctx, 0)", names[i], (double)actual, c->act_shift[i]);
        if (v > 0.f && v < 0x1p-11f)
            return fail(c, DAVO_ERR_RANGE, "%s activations are at most %.4g: too small for the fp16-pair storage at scale 2^%d "
                        "- run davo_calibrate() or davo_set_precision(ctx, 0)", names[i], (double)actual, c->act_shift[i]);
    }
    return DAVO_OK;
}

int davo_forward(davo_ctx* c, int B, const uint8_t* img, const float* flow, const float* seg, float* pose_out) {
    if (!c) return DAVO_ERR_INVALID;
    if (!img || !flow || !seg || !pose_out) return fail(c, DAVO_ERR_INVALID, "null host pointer");
    if (B < 1 || B > c->max_batch) return fail(c, DAVO_ERR_INVALID, "batch %d outside [1,%d]", B, c->max_batch);
    HIP_TRY(c, hipSetDevice(c->device));
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
    HIP_TRY(c, hipMemsetAsync(c->d_range, 0, 8 * sizeof(unsigned), c->stream));      // the monitor covers this call
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
    for (int i = 0; i < nchunks; ++i) {
        const int b0 = i * chunk, nb = std::min(chunk, B - b0);
        uint8_t* di = (uint8_t*)c->s_img + nb_img * b0;
        uint8_t* df = (uint8_t*)c->s_flow + nb_flow * b0;
        uint8_t* ds = (uint8_t*)c->s_seg + nb_seg * b0;
        HIP_TRY(c, hipMemcpyAsync(di, img + nb_img * b0, nb_img * nb, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipMemcpy2DAsync(df, nb_flow, (const uint8_t*)flow + nb_flow * b0, nb_flow, nb_flow / 2, nb,
                                    hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipMemcpyAsync(ds, (const uint8_t*)seg + nb_seg * b0, nb_seg * nb, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipEventRecord(c->copy_done[i], c->copy_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_done[i], 0));
        int rc = forward_device(c, nb, di, (const float*)df, (const float*)ds, (float*)c->s_pose + (size_t)b0 * 12);
        if (rc) return rc;
    }
    HIP_TRY(c, hipMemcpyAsync(pose_out, c->s_pose, (size_t)B * 12 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    unsigned raw[6];
    HIP_TRY(c, hipMemcpyAsync(raw, c->d_range, sizeof raw, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return c->last_precision == 1 ? check_range(c, raw) : DAVO_OK;
}

int davo_activation_range(davo_ctx* c, float* max_abs, int* shifts, int reset) {
    if (!c) return DAVO_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = sync_all_slots(c); if (rc) return rc; }
    unsigned raw[6];
    HIP_TRY(c, hipMemcpy(raw, c->d_range, sizeof raw, hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) {
        float v;
        memcpy(&v, &raw[i], sizeof v);
        if (max_abs) max_abs[i] = ldexpf(v, -c->act_shift[i]);
        if (shifts) shifts[i] = c->act_shift[i];
    }
    if (reset) HIP_TRY(c, hipMemset(c->d_range, 0, 8 * sizeof(unsigned)));
    return DAVO_OK;
}

int davo_set_activation_shifts(davo_ctx* c, const int* shifts) {
    if (!c) return DAVO_ERR_INVALID;
    { int rc = sync_all_slots(c); if (rc) return rc; }
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
    { int rc = sync_all_slots(c); if (rc) return rc; }
    activate_slot(c, 0);
    float* d_pose = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&d_pose), (size_t)B * 12 * sizeof(float)));
    int rc = DAVO_OK;
    const int save_precision = c->precision, save_impl = c->impl;
    c->precision = 1; c->impl = 0;
    // A layer computed from badly ranged inputs still has about the right magnitude, so each pass fixes at
    // least the first badly ranged layer exactly and the later ones to within a few powers of two.
    for (int pass = 0; pass < 8 && rc == DAVO_OK; ++pass) {
        if (hipMemset(c->d_range, 0, 8 * sizeof(unsigned)) != hipSuccess) { rc = fail(c, DAVO_ERR_HIP, "hipMemset failed"); break; }
        rc = forward_device(c, B, d_img, d_flow, d_seg, d_pose);
        if (rc) break;
        unsigned raw[6];
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(raw, c->d_range, sizeof raw, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(c, DAVO_ERR_HIP, "reading the activation ranges failed");
            break;
        }
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
    (void)hipMemset(c->d_range, 0, 8 * sizeof(unsigned));
    (void)hipFree(d_pose);
    if (rc == DAVO_OK && shifts_out) for (int i = 0; i < 6; ++i) shifts_out[i] = c->act_shift[i];
    return rc;
}

const char* davo_last_error(const davo_ctx* c) { return c ? c->err.c_str() : "null context"; }

void davo_destroy(davo_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_all_slots(c);
    for (auto& kv : c->weights) if (kv.second.dev) (void)hipFree(kv.second.dev);
    for (auto& L : c->L) {
        if (L.d_w) (void)hipFree(L.d_w);
        if (L.d_b) (void)hipFree(L.d_b);
        if (L.d_wh) (void)hipFree(L.d_wh);
        if (L.d_bh) (void)hipFree(L.d_bh);
    }
    for (auto e : c->copy_done) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    void* misc[] = {c->d_range, c->d_pose_tiles, c->d_w1patch, c->d_zeros, c->d_wpred, c->d_bpred, c->s_img, c->s_flow, c->s_seg, c->s_pose};
    for (auto p : misc) if (p) (void)hipFree(p);
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
    return sync_all_slots(c);
}
int davo_set_stream(davo_ctx* c, void* hip_stream) {
    if (!c) return DAVO_ERR_INVALID;
    if (hip_stream && c->inflight > 1) return fail(c, DAVO_ERR_INVALID, "a caller-owned stream needs davo_set_inflight(ctx, 1)");
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
    for (auto& pe : c->prof_entries) { pe.launches = 0; pe.total_ms = 0.0; }
    return DAVO_OK;
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

int davo_set_inflight(davo_ctx* c, int n) {
    if (!c || n < 1 || n > 4) return fail(c, DAVO_ERR_INVALID, "inflight must be 1..4");
    if (n > 1 && c->user_stream) return fail(c, DAVO_ERR_INVALID, "in-flight slots use the context's own streams: clear davo_set_stream first");
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = sync_all_slots(c); if (rc) return rc; }
    while ((int)c->slots.size() < n) {
        c->slots.emplace_back();
        int rc = alloc_slot(c, &c->slots.back());
        if (rc) return rc;
    }
    c->inflight = n;
    c->next_slot = 0;
    return DAVO_OK;
}

int davo_set_option(davo_ctx* c, const char* key, int value) {
    if (!c || !key) return DAVO_ERR_INVALID;
    const std::string k = key;
    if (k == "fuse_pose") c->opt_fuse_pose = value != 0;
    else if (k == "fuse_pack") c->opt_fuse_pack = value != 0;
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
            const long nthreads = (long)NB * c->H * (c->W / 4);
            hipLaunchKernelGGL(mask_pack<16>, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream,
                               static_cast<const uint8_t*>(c->last_img), static_cast<const float*>(c->last_flow),
                               static_cast<const float*>(c->last_seg), c->d_tab, c->v, c->last_B, c->H, c->W, c->d_packed);
            HIP_TRY(c, hipGetLastError());
            c->packed_valid = true;
        }
        src = c->d_packed; n = NB * c->H * c->W * c->packed_ld;
    }
    else {
        const char* names[7] = {"cnv1", "cnv2", "cnv3", "cnv4", "cnv5", "cnv6", "cnv7"};
        for (int i = 0; i < 7; ++i)
            if (t == names[i]) { src = c->d_act[i]; n = NB * c->act_floats_per_img[i]; }
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
            p.Hin = H; p.Win = W; p.Hout = Ho; p.Wout = Wo; p.x_pix_bytes = (long)Cin * 4;
            p.cb_log2 = L.cb_log2; p.tpc_log2 = L.tpc_log2; p.cpb = L.cpb; p.nchunks = L.nchunks_h;
            p.w_row_bytes = (long)L.nchunks_h * 128; p.y_mode = 0; p.y_ld = Cout; p.Cout = Cout;
            p.pad_t = pt; p.pad_l = pl; p.rate = rate; p.M = N * Ho * Wo; p.ntaps = k * k;
            p.ntiles_n = L.npad_h / ts.bn; p.relu = relu; p.out_scale = 1.0f / L.wscale; p.bias_scale = L.wscale; p.range = nullptr;
            dim3 grid((p.M + ts.bm - 1) / ts.bm * p.ntiles_n, 1);
            hipError_t le = hipErrorInvalidValue;
            if (stride == 1) {
                if (k == 1) le = launch_h3_generic<1, 1>(tile, p, grid, nullptr);
                if (k == 3) le = launch_h3_generic<3, 1>(tile, p, grid, nullptr);
                if (k == 5) le = launch_h3_generic<5, 1>(tile, p, grid, nullptr);
                if (k == 7) le = launch_h3_generic<7, 1>(tile, p, grid, nullptr);
            } else {
                if (k == 1) le = launch_h3_generic<1, 2>(tile, p, grid, nullptr);
                if (k == 3) le = launch_h3_generic<3, 2>(tile, p, grid, nullptr);
                if (k == 5) le = launch_h3_generic<5, 2>(tile, p, grid, nullptr);
                if (k == 7) le = launch_h3_generic<7, 2>(tile, p, grid, nullptr);
            }
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
