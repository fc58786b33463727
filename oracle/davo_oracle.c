/* CPU ORACLE (test infrastructure, not product code) — float32 C restatement of DAVO's
 * frame-to-frame pose inference path; also the timed CPU baseline ("port") of bench.py.
 *
 * PARITY UNPINNED: the reference computes this path with tensorflow-gpu==1.13.1
 * (requirements.txt:1), absent offline, and ships no test/golden vector/checkpoint for it
 * (SURVEY.md §4, §8c).  This file is pinned by agreement with oracle/davo_oracle.py
 * (float64 numpy) and torch-CPU conv2d (tests/test_oracle.py), not by TF outputs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Restates (paths relative to /root/reference):
 *   davo.py:1519-1522            preprocess_image
 *   data_loader.py:537-557       strip = src0 | tgt | src1
 *   davo.py:978-982,998-1004     flow planes 0,1; seg file planes (src0,tgt,src1)
 *   davo.py:1088-1102            SE input transform
 *   nets/attention_module.py:54-103   se(mode='gp')
 *   davo.py:1115,1178            one_hot(int32(seg)) . w  == LUT gather (out of range -> 0)
 *   nets/posenn.py:380-394       static seg weights
 *   davo.py:1404-1442            masking + concat
 *   nets/posenn.py:189-254       decouple_sharednet_v0_dilation
 *   davo.py:1453-1458            two shared-weight calls -> [B,2,6]
 * TF semantics: SAME padding (pad_before = total/2), float->int32 cast truncates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NCLS 19

typedef struct {
    int cin_per_frame, cnv6_out, se_act, norm_flow, abs_mode, att_source, mask_rgb, mask_info;
} oracle_variant;

/* canonical weight order (oracle/c_oracle.py builds the pointer table) */
enum { W_CNV1 = 0, W_CNV2 = 2, W_CNV3 = 4, W_CNV4 = 6, W_CNV5 = 8,
       W_ROT = 10, W_TRANS = 16,          /* each: cnv6 w,b, cnv7 w,b, pred w,b */
       W_SE = 22,                         /* w1[2,8] b1[8] w2[8,19] b2[19] */
       W_STATIC = 26, W_COUNT = 27 };

static void same_pad(int in, int k, int stride, int rate, int* out, int* before, int* after) {
    int o = (in + stride - 1) / stride;
    int keff = (k - 1) * rate + 1;
    int total = (o - 1) * stride + keff - in;
    if (total < 0) total = 0;
    *out = o; *before = total / 2; *after = total - total / 2;
}

/* slim.conv2d(padding='SAME'): NHWC x, HWIO w, + bias, optional ReLU (posenn.py:205-215,238-240) */
void oracle_conv2d_same(const float* x, int N, int H, int W, int Cin,
                        const float* w, int kh, int kw, int Cout, const float* b,
                        int stride, int rate, int relu, float* y) {
    int Ho, Wo, pt, pb, pl, pr;
    same_pad(H, kh, stride, rate, &Ho, &pt, &pb);
    same_pad(W, kw, stride, rate, &Wo, &pl, &pr);
    const int Hp = H + pt + pb, Wp = W + pl + pr;
    float* xp = (float*)calloc((size_t)N * Hp * Wp * Cin, sizeof(float));
    for (int n = 0; n < N; ++n)
        for (int iy = 0; iy < H; ++iy)
            memcpy(xp + (((size_t)n * Hp + iy + pt) * Wp + pl) * Cin,
                   x + ((size_t)n * H + iy) * W * Cin, (size_t)W * Cin * sizeof(float));
    enum { PB = 4 };                                   /* output pixels per register block */
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy) {
            float acc[PB][256];
            for (int ox0 = 0; ox0 < Wo; ox0 += PB) {
                const int np = (Wo - ox0 < PB) ? Wo - ox0 : PB;
                for (int p = 0; p < PB; ++p)
                    for (int co = 0; co < Cout; ++co) acc[p][co] = 0.f;
                for (int ky = 0; ky < kh; ++ky)
                    for (int kx = 0; kx < kw; ++kx) {
                        const float* wt = w + (size_t)(ky * kw + kx) * Cin * Cout;
                        const float* xr[PB];
                        for (int p = 0; p < PB; ++p) {
                            int ox = ox0 + (p < np ? p : 0);
                            xr[p] = xp + (((size_t)n * Hp + oy * stride + ky * rate) * Wp
                                          + ox * stride + kx * rate) * Cin;
                        }
                        for (int ci = 0; ci < Cin; ++ci) {
                            const float* wr = wt + (size_t)ci * Cout;
                            const float x0 = xr[0][ci], x1 = xr[1][ci], x2 = xr[2][ci], x3 = xr[3][ci];
#pragma omp simd
                            for (int co = 0; co < Cout; ++co) {
                                const float wv = wr[co];
                                acc[0][co] += x0 * wv; acc[1][co] += x1 * wv;
                                acc[2][co] += x2 * wv; acc[3][co] += x3 * wv;
                            }
                        }
                    }
                for (int p = 0; p < np; ++p) {
                    float* yo = y + (((size_t)n * Ho + oy) * Wo + ox0 + p) * Cout;
                    for (int co = 0; co < Cout; ++co) {
                        float v = acc[p][co] + b[co];
                        yo[co] = (relu && v < 0.f) ? 0.f : v;
                    }
                }
            }
        }
    free(xp);
}

static float se_act(int kind, float v) {
    if (kind == 1) return tanhf(v);
    if (kind == 2) return v > 0.f ? v : 0.2f * v;
    return v > 0.f ? v : 0.f;
}

/* [B,3,19] attention tables for (tgt,src0,src1): davo.py:1175-1180,1385-1412 */
static void attention_tables(const oracle_variant* v, int B, int H, int W, const float* flow,
                             const float* const* wts, float* tab) {
    for (int i = 0; i < B * 3 * NCLS; ++i) tab[i] = 1.f;
    if (v->att_source == 1) {
        const float *w1 = wts[W_SE], *b1 = wts[W_SE + 1], *w2 = wts[W_SE + 2], *b2 = wts[W_SE + 3];
        const size_t plane = (size_t)H * W * 2;
        for (int b = 0; b < B; ++b)
            for (int s = 0; s < 2; ++s) {
                const float* f = flow + ((size_t)b * 4 + s) * plane;
                double sum[2] = {0, 0};
                for (size_t i = 0; i < (size_t)H * W; ++i)
                    for (int c = 0; c < 2; ++c) {
                        float t = f[2 * i + c];
                        if (v->norm_flow) t = (t - 0.32140523f) / 15.384229f;
                        if (v->abs_mode == 3 || (v->abs_mode == 1 && c == 0) || (v->abs_mode == 2 && c == 1))
                            t = fabsf(t);
                        sum[c] += t;
                    }
                float sq[2] = {(float)(sum[0] / ((double)H * W)), (float)(sum[1] / ((double)H * W))};
                float e[8];
                for (int j = 0; j < 8; ++j)
                    e[j] = se_act(v->se_act, sq[0] * w1[j] + sq[1] * w1[8 + j] + b1[j]);
                float* t = tab + ((size_t)b * 3 + 1 + s) * NCLS;
                for (int c = 0; c < NCLS; ++c) {
                    float z = b2[c];
                    for (int j = 0; j < 8; ++j) z += e[j] * w2[j * NCLS + c];
                    t[c] = 1.f / (1.f + expf(-z));
                }
            }
    } else if (v->att_source == 2 || v->att_source == 3) {
        for (int b = 0; b < B; ++b)
            for (int k = (v->att_source == 3 ? 0 : 1); k < 3; ++k)
                for (int c = 0; c < NCLS; ++c)
                    tab[((size_t)b * 3 + k) * NCLS + c] = 1.f / (1.f + expf(-wts[W_STATIC][c]));
    }
}

static inline float att_at(const float* tab19, float seg) {
    /* tf.cast(float->int32) truncates toward zero; NaN / inf / beyond-int32 inputs are platform-defined there and
     * pinned here (and in the product) to "no class": only finite values in (-1, 19) select a row */
    if (!(seg > -1.0f && seg < (float)NCLS)) return 0.f;   /* tf.one_hot: out of range -> zero row */
    return tab19[(int)seg];
}

/* packed: [B,2,H,W,2*cin] (davo.py:961-1004,1404-1442; posenn.py:198) */
void oracle_pack(const oracle_variant* v, int B, int H, int W, const uint8_t* img,
                 const float* flow, const float* seg, const float* const* wts, float* packed) {
    const int c = v->cin_per_frame, C = 2 * c;
    float* tab = (float*)malloc(sizeof(float) * (size_t)B * 3 * NCLS);
    attention_tables(v, B, H, W, flow, wts, tab);
    const float inv255 = 1.0f / 255.0f;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int s = 0; s < 2; ++s)
                for (int x = 0; x < W; ++x) {
                    float* o = packed + ((((size_t)b * 2 + s) * H + y) * W + x) * C;
                    const uint8_t* row = img + ((size_t)b * H + y) * 3 * W * 3;
                    const uint8_t* pt = row + (size_t)(W + x) * 3;
                    const uint8_t* ps = row + (size_t)((s ? 2 * W : 0) + x) * 3;
                    /* tf.ones_like overrides are ones everywhere, ignore pixels included:
                     * tgt unless static_all (davo.py:1394,1411); every frame for -no_segmask (:1387) */
                    const float at = v->att_source == 3
                        ? att_at(tab + ((size_t)b * 3 + 0) * NCLS, seg[(((size_t)b * 3 + 1) * H + y) * W + x])
                        : 1.f;
                    const float as = v->att_source == 0 ? 1.f
                        : att_at(tab + ((size_t)b * 3 + 1 + s) * NCLS,
                                 seg[(((size_t)b * 3 + (s ? 2 : 0)) * H + y) * W + x]);
                    for (int k = 0; k < C; ++k) o[k] = 0.f;
                    for (int k = 0; k < 3; ++k) {
                        float t = (float)pt[k] * inv255 * 2.0f - 1.0f;
                        float r = (float)ps[k] * inv255 * 2.0f - 1.0f;
                        o[k] = v->mask_rgb ? t * at : t;
                        o[c + k] = v->mask_rgb ? r * as : r;
                    }
                    if (c == 5) {
                        const float* f = flow + ((((size_t)b * 4 + s) * H + y) * W + x) * 2;
                        o[c + 3] = v->mask_info ? f[0] * as : f[0];
                        o[c + 4] = v->mask_info ? f[1] * as : f[1];
                    }
                }
    free(tab);
}

/* decouple_sharednet_v0_dilation (posenn.py:189-254) on N pair images -> pose[N,6] */
void oracle_posenet(const oracle_variant* v, int N, int H, int W, const float* x,
                    const float* const* wts, float* pose) {
    const int C0 = 2 * v->cin_per_frame, c6 = v->cnv6_out;
    const int H1 = (H + 1) / 2, W1 = (W + 1) / 2, H2 = (H1 + 1) / 2, W2 = (W1 + 1) / 2;
    const int H3 = (H2 + 1) / 2, W3 = (W2 + 1) / 2;
    float* a1 = (float*)malloc(sizeof(float) * (size_t)N * H1 * W1 * 16);
    float* a2 = (float*)malloc(sizeof(float) * (size_t)N * H2 * W2 * 32);
    float* a3 = (float*)malloc(sizeof(float) * (size_t)N * H2 * W2 * 64);
    float* a4 = (float*)malloc(sizeof(float) * (size_t)N * H2 * W2 * 128);
    float* a5 = (float*)malloc(sizeof(float) * (size_t)N * H2 * W2 * 256);
    float* a6 = (float*)malloc(sizeof(float) * (size_t)N * H2 * W2 * c6);
    float* a7 = (float*)malloc(sizeof(float) * (size_t)N * H3 * W3 * 256);
    float* a8 = (float*)malloc(sizeof(float) * (size_t)N * H3 * W3 * 3);
    oracle_conv2d_same(x, N, H, W, C0, wts[W_CNV1], 7, 7, 16, wts[W_CNV1 + 1], 2, 1, 1, a1);
    oracle_conv2d_same(a1, N, H1, W1, 16, wts[W_CNV2], 5, 5, 32, wts[W_CNV2 + 1], 2, 1, 1, a2);
    oracle_conv2d_same(a2, N, H2, W2, 32, wts[W_CNV3], 3, 3, 64, wts[W_CNV3 + 1], 1, 2, 1, a3);
    oracle_conv2d_same(a3, N, H2, W2, 64, wts[W_CNV4], 3, 3, 128, wts[W_CNV4 + 1], 1, 4, 1, a4);
    oracle_conv2d_same(a4, N, H2, W2, 128, wts[W_CNV5], 3, 3, 256, wts[W_CNV5 + 1], 1, 8, 1, a5);
    for (int head = 0; head < 2; ++head) {
        const float* const* hw = wts + (head ? W_TRANS : W_ROT);
        oracle_conv2d_same(a5, N, H2, W2, 256, hw[0], 3, 3, c6, hw[1], 1, 2, 1, a6);
        oracle_conv2d_same(a6, N, H2, W2, c6, hw[2], 3, 3, 256, hw[3], 2, 1, 1, a7);
        oracle_conv2d_same(a7, N, H3, W3, 256, hw[4], 1, 1, 3, hw[5], 1, 1, 0, a8);
        for (int n = 0; n < N; ++n)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int p = 0; p < H3 * W3; ++p) s += a8[((size_t)n * H3 * W3 + p) * 3 + j];
                pose[n * 6 + head * 3 + j] = 0.01f * (float)(s / (H3 * W3));    /* posenn.py:241,250 */
            }
    }
    free(a1); free(a2); free(a3); free(a4); free(a5); free(a6); free(a7); free(a8);
}

/* DAVO.inference(mode='pose') (davo.py:1553-1569): pose_out [B,2,6] */
int oracle_forward(const oracle_variant* v, int B, int H, int W, const uint8_t* img,
                   const float* flow, const float* seg, const float* const* wts,
                   float* pose_out, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    float* packed = (float*)malloc(sizeof(float) * (size_t)B * 2 * H * W * 2 * v->cin_per_frame);
    if (!packed) return -1;
    oracle_pack(v, B, H, W, img, flow, seg, wts, packed);
    oracle_posenet(v, 2 * B, H, W, packed, wts, pose_out);
    free(packed);
    return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
