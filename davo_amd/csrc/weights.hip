// weights.hip — weight tensors of the pose path: the TF variable names and shapes a variant needs
// (SURVEY.md table W), and their one-time re-layout into the kernels' operand formats:
//   f32 path    HWIO [KS,KS,Cin,Cout] -> Wp[Cout_pad][K_pad] floats, k = tap * Cin_packed + c
//   f16x3 path  HWIO -> [Cout_pad][chunk][32 hi | 32 lo] halves, channel-block major / tap minor,
//               pre-multiplied by a power of two so the fp16 residuals of small weights stay normal
// rotation/cnv6 and translation/cnv6 are stacked along N (one GEMM, both read cnv5:
// nets/posenn.py:222-238); cnv7 is one group per head.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "ctx.h"
#include "plan.h"

namespace davo {

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

static int pick_bn(int cout) { return cout <= 32 ? 32 : (cout % 128 == 0 ? 128 : (cout <= 64 ? 64 : 128)); }

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
    if (const char* e = tuning_env("DAVO_H3_TILE")) {        // tuning build only: force a tile where it fits
        const int t = atoi(e);
        const char* only = tuning_env("DAVO_H3_TILE_LABEL"); // restrict the force to one layer ("cnv4")
        if (t >= 0 && t < NUM_TILES && cout >= tile_shape(t).bn && (!only || strcmp(only, L.label) == 0)) L.tile_h = t;
    }
    const int gran = cout > 128 ? 256 : cout > 64 ? 128 : cout > 32 ? 64 : 32;    // widest N tile a launch may use
    L.npad_h = (cout + gran - 1) / gran * gran;
}

// the pose head's 1x1 kernels [2][256][3] + [2][3]: every arithmetic mode reads them, so they are built at the first forward whatever
// the mode; the float32 convolution weights (below) only when a float32 forward is issued (round 5: a rank that runs f16x3 packs
// and uploads 13 MB less in front of its first batch)
int build_pred_weights(davo_ctx* c) {
    auto W = [&](const std::string& n) -> const HostTensor& { return c->weights.at(n); };
    const char* heads[2] = {"rotation", "translation"};
    std::vector<float> wp(2 * 256 * 3), bp(2 * 3);
    for (int h = 0; h < 2; ++h) {
        const std::string p = std::string("pose_exp_net/pose/") + heads[h] + "/pred/";
        memcpy(wp.data() + h * 768, W(p + "weights").data.data(), 768 * sizeof(float));
        memcpy(bp.data() + h * 3, W(p + "biases").data.data(), 3 * sizeof(float));
    }
    int rc = upload(c, wp, &c->d_wpred); if (rc) return rc;
    rc = upload(c, bp, &c->d_bpred); if (rc) return rc;
    c->pred_ready = true;
    return DAVO_OK;
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
    {   // float32 patch kernels (conv_patch_f32.h): the weights as the MFMA's A operand, in the order the kernels consume them.
        // cnv2: [25 taps][2 N groups][4 instructions t][64 lanes]; lane (r = l & 15, kq = l >> 4) of instruction t holds
        // W[tap][input channel 4 kq + t][output channel 16 ng + r]
        const HostTensor& t2 = W("pose_exp_net/cnv2/weights");           // [5][5][16][32]
        std::vector<float> w2((size_t)25 * 2 * 4 * 64);
        for (int tap = 0; tap < 25; ++tap)
            for (int ng = 0; ng < 2; ++ng)
                for (int q = 0; q < 4; ++q)
                    for (int l = 0; l < 64; ++l)
                        w2[((size_t)(tap * 2 + ng) * 4 + q) * 64 + l] = t2.data[((size_t)tap * 16 + 4 * (l >> 4) + q) * 32 + 16 * ng + (l & 15)];
        int rc = upload(c, w2, &c->d_w2patch_f32); if (rc) return rc;
        // cnv3: [9 taps][4 waves][8 instructions 4 j + t][64 lanes]: W[tap][input channel 4 kq + t + 16 j][output channel 16 wave + r]
        const HostTensor& t3 = W("pose_exp_net/cnv3/weights");           // [3][3][32][64]
        std::vector<float> w3((size_t)9 * 4 * 8 * 64);
        for (int tap = 0; tap < 9; ++tap)
            for (int wv = 0; wv < 4; ++wv)
                for (int q = 0; q < 8; ++q)
                    for (int l = 0; l < 64; ++l)
                        w3[((size_t)(tap * 4 + wv) * 8 + q) * 64 + l] = t3.data[((size_t)tap * 32 + 4 * (l >> 4) + (q & 3) + 16 * (q >> 2)) * 64 + 16 * wv + (l & 15)];
        rc = upload(c, w3, &c->d_w3patch_f32); if (rc) return rc;
        // cnv1: [14 steps = ky x h][8 packed channels c][64 lanes]: lane (r' = output channel, kq) holds W[ky][kx = 4 h + kq][chmap1[c]][r'],
        // zero for the dummy tap kx = 7 and for packed channels the variant does not have
        const HostTensor& t1 = W("pose_exp_net/cnv1/weights");           // [7][7][2 cpf][16]
        const int cin_tf = 2 * cpf;
        std::vector<float> w1((size_t)14 * 8 * 64, 0.f);
        for (int st = 0; st < 14; ++st)
            for (int q = 0; q < 8; ++q)
                for (int l = 0; l < 64; ++l) {
                    const int ky = st >> 1, kx = 4 * (st & 1) + (l >> 4), ci = chmap1[q];
                    if (kx >= 7 || ci < 0 || ci >= cin_tf) continue;
                    w1[((size_t)st * 8 + q) * 64 + l] = t1.data[(((size_t)ky * 7 + kx) * cin_tf + ci) * 16 + (l & 15)];
                }
        rc = upload(c, w1, &c->d_w1patch_f32); if (rc) return rc;
    }
    c->packed_ready = true;
    return DAVO_OK;
}

static int upload_bytes(davo_ctx* c, const void* host, size_t bytes, void** dev) {
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
        // a single-group layer's bias is followed by 3 x npad_h zeros: the later parts of a split-K launch (forward.hip) start from them
        std::vector<float> bp((size_t)L.npad_h * (L.groups > 1 ? L.groups : 4), 0.f);
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
    {   // cnv2 patch kernel: [15 steps = ky x tap pair h][N group][hi|lo][64 lanes][8 halves]; lane = (channel c = l&15,
        // k-quarter kq = l>>4): tap kx = 2 h + (kq >> 1) (kx = 5: zero), input channels 8 (kq & 1) .. +7
        const ConvLayer& L = c->L[1];
        const HostTensor& t = W("pose_exp_net/cnv2/weights");            // [5][5][16][32]
        std::vector<_Float16> wp((size_t)cp2::WBYTES / 2, (_Float16)0.0f);
        for (int step = 0; step < cp2::STEPS; ++step)
            for (int g = 0; g < 2; ++g)
                for (int l = 0; l < 64; ++l) {
                    const int n = 16 * g + (l & 15), kq = l >> 4, ky = step / 3, kx = 2 * (step % 3) + (kq >> 1);
                    if (kx >= 5) continue;
                    for (int j = 0; j < 8; ++j) {
                        const int ci = 8 * (kq & 1) + j;
                        const float v = t.data[(((size_t)ky * 5 + kx) * 16 + ci) * 32 + n] * L.wscale;
                        const size_t base = ((size_t)(step * 2 + g) * 2) * 64;
                        split_f16(v, &wp[(base + l) * 8 + j], &wp[(base + 64 + l) * 8 + j]);
                    }
                }
        int rc = upload_bytes(c, wp.data(), wp.size() * sizeof(_Float16), reinterpret_cast<void**>(&c->d_w2patch));
        if (rc) return rc;
    }
    {   // cnv3 patch kernel: [9 taps][4 N groups][hi|lo][64 lanes][8 halves]; lane = (channel c = l&15, channel quarter kq = l>>4)
        const ConvLayer& L = c->L[2];
        const HostTensor& t = W("pose_exp_net/cnv3/weights");            // [3][3][32][64]
        std::vector<_Float16> wp((size_t)cp3::WBYTES / 2, (_Float16)0.0f);
        for (int step = 0; step < cp3::STEPS; ++step)
            for (int g = 0; g < 4; ++g)
                for (int l = 0; l < 64; ++l) {
                    const int n = 16 * g + (l & 15), kq = l >> 4;
                    for (int j = 0; j < 8; ++j) {
                        const float v = t.data[((size_t)step * 32 + 8 * kq + j) * 64 + n] * L.wscale;
                        const size_t base = ((size_t)(step * 4 + g) * 2) * 64;
                        split_f16(v, &wp[(base + l) * 8 + j], &wp[(base + 64 + l) * 8 + j]);
                    }
                }
        int rc = upload_bytes(c, wp.data(), wp.size() * sizeof(_Float16), reinterpret_cast<void**>(&c->d_w3patch));
        if (rc) return rc;
    }
    {   // Per-channel guard of the f16x3 arithmetic.  One power-of-two scale per layer keeps the LAYER's largest
        // activation at [512, 1024); a channel whose activations are 2^-r of that keeps its hi half but loses its lo half below
        // r ~ 12 (absolute error 2^-25 of the stored value), which only matters if the consuming layer multiplies that
        // channel by weights 2^r larger than the others' - visible here, in the weights: the spread of the per-input-channel
        // weight norms of cnv2..cnv7.  Measured (tools/exp_dynamic_range.py): r = 14 -> 2.6e-6, r = 18 -> 5e-5, r = 22 -> 9e-4
        // against the 1e-4 bar.  Beyond 2^14 the network runs on the float32 kernels (auto_range) or is refused (DAVO_ERR_RANGE).
        c->weight_channel_spread_log2 = 0;
        c->weight_channel_spread_layer.clear();
        const char* lw[6] = {"pose_exp_net/cnv2/weights", "pose_exp_net/cnv3/weights", "pose_exp_net/cnv4/weights", "pose_exp_net/cnv5/weights",
                             "/cnv6/weights", "/cnv7/weights"};
        for (int k = 0; k < 6; ++k) {
            for (int h = 0; h < (k < 4 ? 1 : 2); ++h) {
                const std::string name = k < 4 ? std::string(lw[k]) : std::string("pose_exp_net/pose/") + heads[h] + lw[k];
                const HostTensor& t = W(name);
                const int64_t taps = t.shape[0] * t.shape[1], cin = t.shape[2], cout = t.shape[3];
                // Upward spread only (round 4): what matters is a channel whose weights are 2^r LARGER than the typical channel's (its
                // small activations then carry the output).  A dead or pruned channel with tiny weights is harmless - its lost
                // low bits are multiplied by nothing - so the reference point is the lower quartile of the per-channel norms,
                // not their minimum: up to a quarter of the channels may be arbitrarily small without moving it.
                std::vector<float> norms;
                for (int64_t ci = 0; ci < cin; ++ci) {
                    float m = 0.f;
                    for (int64_t tp = 0; tp < taps; ++tp)
                        for (int64_t co = 0; co < cout; ++co) m = std::fmax(m, std::fabs(t.data[(tp * cin + ci) * cout + co]));
                    if (std::isfinite(m)) norms.push_back(m);
                }
                if (norms.empty()) continue;
                std::sort(norms.begin(), norms.end());
                const float mx = norms.back(), typical = norms[norms.size() / 4];
                if (mx > 0.f && typical > 0.f) {
                    int e1, e0;
                    (void)frexpf(mx, &e1); (void)frexpf(typical, &e0);
                    if (e1 - e0 > c->weight_channel_spread_log2) { c->weight_channel_spread_log2 = e1 - e0; c->weight_channel_spread_layer = name; }
                } else if (mx > 0.f) {          // more than a quarter of the channels are all-zero: compare with the smallest non-zero norm
                    float mn = mx;
                    for (float m : norms) if (m > 0.f) { mn = m; break; }
                    int e1, e0;
                    (void)frexpf(mx, &e1); (void)frexpf(mn, &e0);
                    if (e1 - e0 > c->weight_channel_spread_log2) { c->weight_channel_spread_log2 = e1 - e0; c->weight_channel_spread_layer = name; }
                }
            }
        }
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

}  // namespace davo
