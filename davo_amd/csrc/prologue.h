// prologue.h — the HBM-bound front of the pose path: SE squeeze, the 2->8->19 excitation
// MLP, and the mask + pack pass that builds the PoseNN input.
//
// Reference (all under /root/reference): davo.py:1519-1522 (u8 -> f32 * (1/255) * 2 - 1),
// data_loader.py:537-557 (strip = src0 | tgt | src1), davo.py:978-982 / 998-1004 (flow planes
// 0,1; seg file planes src0,tgt,src1), davo.py:1088-1102 (SE input transform),
// nets/attention_module.py:54-103 (se, mode 'gp'), davo.py:1115,1178 (one_hot . weights ==
// 19-entry LUT gather, out-of-range id -> 0), nets/posenn.py:380-394 (static weights),
// davo.py:1404-1442 (masking, concat).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "params.h"
#include "pose_tail.h"

namespace davo {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// SE squeeze, pass 1: partial[b][s][chunk][2] = sum over the chunk's pixels of t(flow[b][s]).
// grid (SQ_CHUNKS, 2, B), 256 threads; float4 = two pixels x (fx, fy).  Fixed chunking and a
// fixed reduction tree -> bitwise reproducible run to run.
__global__ __launch_bounds__(256) void se_squeeze_partial(const float* __restrict__ flow, int HW,
                                                          int norm_flow, int abs_mode,
                                                          float* __restrict__ partial) {
    const int chunk = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
    const float4* f = reinterpret_cast<const float4*>(flow + ((size_t)b * 4 + s) * HW * 2);
    const int nvec = HW / 2;                                   // HW is a multiple of 16
    const int per = (nvec + SQ_CHUNKS - 1) / SQ_CHUNKS;
    const int beg = chunk * per, end = min(beg + per, nvec);
    float sx = 0.f, sy = 0.f;
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        float4 v = f[i];
        if (norm_flow) {
            v.x = (v.x - 0.32140523f) / 15.384229f; v.y = (v.y - 0.32140523f) / 15.384229f;
            v.z = (v.z - 0.32140523f) / 15.384229f; v.w = (v.w - 0.32140523f) / 15.384229f;
        }
        if (abs_mode & 1) { v.x = fabsf(v.x); v.z = fabsf(v.z); }
        if (abs_mode & 2) { v.y = fabsf(v.y); v.w = fabsf(v.w); }
        sx += v.x + v.z;
        sy += v.y + v.w;
    }
    __shared__ float red[2][4];
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[0][wid] = sx; red[1][wid] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = partial + (((size_t)b * 2 + s) * SQ_CHUNKS + chunk) * 2;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// One wave's share of the excitation: tab[b][frame][19] for frame = (tgt, src0, src1).
// Lanes 0..31 fetch the squeeze partials (fixed butterfly: bitwise reproducible), every lane evaluates the 8 bottleneck
// units, lane c < 19 the class c of the recovery layer — the three dependent steps of the MLP cost three memory round
// trips instead of the ~200 a single lane needed.  AGENT: the partials were written by other workgroups of the SAME
// launch (se_squeeze_excite's last workgroup): read them past the non-coherent caches.
template <bool AGENT>
__device__ __forceinline__ void se_excite_wave(const float* __restrict__ partial, int HW, const Variant& v, int b, int frame, int lane,
                                               const float* __restrict__ w1, const float* __restrict__ b1,
                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                               const float* __restrict__ wstatic, float* __restrict__ tab) {
    float* t = tab + ((size_t)b * 3 + frame) * NCLS;
    if (v.att_source == 1 && frame >= 1) {
        const float* pp = partial + ((size_t)b * 2 + (frame - 1)) * SQ_CHUNKS * 2;
        static_assert(SQ_CHUNKS == 32, "one partial pair per lane of the lower half-wave");
        float sx = 0.f, sy = 0.f;
        if (lane < SQ_CHUNKS) {
            sx = AGENT ? agent_load(pp + 2 * lane) : pp[2 * lane];
            sy = AGENT ? agent_load(pp + 2 * lane + 1) : pp[2 * lane + 1];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
        const float inv = 1.0f / (float)HW;
        sx *= inv; sy *= inv;                                  // tf.reduce_mean(axis=[1,2])
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = sx * w1[j] + sy * w1[8 + j] + b1[j];     // dense [2,8]
            e[j] = v.se_act == 1 ? tanhf(z) : v.se_act == 2 ? (z > 0.f ? z : 0.2f * z) : fmaxf(z, 0.f);
        }
        if (lane < NCLS) {
            float z = b2[lane];
#pragma unroll
            for (int j = 0; j < 8; ++j) z += e[j] * w2[j * NCLS + lane];   // dense [8,19]
            t[lane] = 1.0f / (1.0f + expf(-z));
        }
    } else if (lane < NCLS) {
        if ((v.att_source == 2 && frame >= 1) || v.att_source == 3) t[lane] = 1.0f / (1.0f + expf(-wstatic[lane]));
        else t[lane] = 1.0f;                                   // davo.py:1408-1412 / 1385-1389
    }
}

// SE pass 2 + excitation as a launch of its own: one 64-lane wave per (triplet, frame).  Used by the variants without a
// squeeze (static / no attention) and with davo_set_option "fold_tails" 0.
__global__ __launch_bounds__(64) void se_excite(const float* __restrict__ partial, int HW, Variant v,
                                                const float* __restrict__ w1, const float* __restrict__ b1,
                                                const float* __restrict__ w2, const float* __restrict__ b2,
                                                const float* __restrict__ wstatic,
                                                float* __restrict__ tab, unsigned* __restrict__ range_reset) {
    // a ticketed device-path batch starts its own f16x3 range record (api.hip): first kernel of the forward, so in stream
    // order before any storing epilogue; a null pointer = the record is the caller's business
    if (range_reset && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) range_reset[RANGE_SNAP] = 0u;   // the per-batch word (params.h)
    se_excite_wave<false>(partial, HW, v, blockIdx.x, blockIdx.y, threadIdx.x, w1, b1, w2, b2, wstatic, tab);
}

// Squeeze and excitation in ONE launch (attention_module.py:64-67 then :89-101): se_squeeze_partial's body, and the
// workgroup that delivers the LAST of a triplet's 2 x SQ_CHUNKS partial sums evaluates that triplet's three tables
// (pose_tail.h: ticket counter per triplet, agent-scope fences).  Same sums in the same order as the two launches.
__global__ __launch_bounds__(256) void se_squeeze_excite(const float* __restrict__ flow, int HW, Variant v,
                                                         float* __restrict__ partial, unsigned* __restrict__ counters,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         const float* __restrict__ wstatic, float* __restrict__ tab,
                                                         unsigned* __restrict__ range_reset) {
    const int chunk = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
    if (range_reset && chunk == 0 && s == 0 && b == 0 && threadIdx.x == 0) range_reset[RANGE_SNAP] = 0u;      // as se_excite
    const float4* f = reinterpret_cast<const float4*>(flow + ((size_t)b * 4 + s) * HW * 2);
    const int nvec = HW / 2;                                   // HW is a multiple of 16
    const int per = (nvec + SQ_CHUNKS - 1) / SQ_CHUNKS;
    const int beg = chunk * per, end = min(beg + per, nvec);
    float sx = 0.f, sy = 0.f;
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        float4 q = f[i];
        if (v.norm_flow) {
            q.x = (q.x - 0.32140523f) / 15.384229f; q.y = (q.y - 0.32140523f) / 15.384229f;
            q.z = (q.z - 0.32140523f) / 15.384229f; q.w = (q.w - 0.32140523f) / 15.384229f;
        }
        if (v.abs_mode & 1) { q.x = fabsf(q.x); q.z = fabsf(q.z); }
        if (v.abs_mode & 2) { q.y = fabsf(q.y); q.w = fabsf(q.w); }
        sx += q.x + q.z;
        sy += q.y + q.w;
    }
    __shared__ float red[2][4];
    __shared__ unsigned ticket;
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[0][wid] = sx; red[1][wid] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = partial + (((size_t)b * 2 + s) * SQ_CHUNKS + chunk) * 2;
        agent_store(o, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        agent_store(o + 1, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
        agent_stores_done();
    }
    if (!last_workgroup(counters + b, 2u * SQ_CHUNKS, &ticket)) return;
    if (wid < 3) se_excite_wave<true>(partial, HW, v, b, wid, lane, w1, b1, w2, b2, wstatic, tab);
    if (threadIdx.x == 0) __hip_atomic_store(counters + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float att_lookup(const float* tab19, float seg) {
    // tf.cast(float -> int32) truncates toward zero; one_hot of an out-of-range id is a zero row.  NaN / inf /
    // beyond-int32 labels are platform-defined in the cast (x86: INT_MIN, GPUs: 0 or saturation) and pinned to
    // "no class" here: only finite values in (-1, 19) select a row (the comparison is false for NaN).
    return (seg > -1.0f && seg < (float)NCLS) ? tab19[(int)seg] : 0.f;
}

__device__ __forceinline__ float u8_to_unit(uint32_t byte) {
    return (float)byte * (1.0f / 255.0f) * 2.0f - 1.0f;        // davo.py:1521-1522
}

// Mask + pack: one thread per 4 horizontally adjacent pixels of one pair image.
// out[(b*2+s)][y][x][0..LD): LD = 8 : tgt rgb | src rgb*att | flow*att      (product layout; the
//                                      two identically-zero tgt-flow channels are dropped)
//                            LD = 10: tgt rgb | 0 0 | src rgb*att | flow*att (reference layout)
// v0 variants (rgb only) leave the flow slots zero.
template <int LD>
__global__ __launch_bounds__(256) void mask_pack(const uint8_t* __restrict__ img, const float* __restrict__ flow,
                                                 const float* __restrict__ seg, const float* __restrict__ tab,
                                                 Variant v, int B, int H, int W, float* __restrict__ out) {
    const int W4 = W >> 2;
    const long total = (long)B * 2 * H * W4;
    const long gid_raw = (long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = gid_raw < total;
    constexpr bool STAGED = LD == 16 || LD == 8;               // 32 B per pixel: the stores leave through LDS (below)
    if (!STAGED && !valid) return;
    const long gid = valid ? gid_raw : total - 1;             // staged: every thread reaches the barrier below
    const int x4 = (int)(gid % W4);
    long t = gid / W4;
    const int y = (int)(t % H); t /= H;
    const int s = (int)(t & 1), b = (int)(t >> 1);
    const int x = x4 * 4;

    const uint8_t* row = img + ((size_t)b * H + y) * (size_t)(9 * W);
    const uint32_t* pt = reinterpret_cast<const uint32_t*>(row + (size_t)(W + x) * 3);
    const uint32_t* ps = reinterpret_cast<const uint32_t*>(row + (size_t)((s ? 2 * W : 0) + x) * 3);
    const uint32_t t0 = pt[0], t1 = pt[1], t2 = pt[2];
    const uint32_t s0 = ps[0], s1 = ps[1], s2 = ps[2];
    const uint8_t tb[12] = {(uint8_t)t0, (uint8_t)(t0 >> 8), (uint8_t)(t0 >> 16), (uint8_t)(t0 >> 24),
                            (uint8_t)t1, (uint8_t)(t1 >> 8), (uint8_t)(t1 >> 16), (uint8_t)(t1 >> 24),
                            (uint8_t)t2, (uint8_t)(t2 >> 8), (uint8_t)(t2 >> 16), (uint8_t)(t2 >> 24)};
    const uint8_t sb[12] = {(uint8_t)s0, (uint8_t)(s0 >> 8), (uint8_t)(s0 >> 16), (uint8_t)(s0 >> 24),
                            (uint8_t)s1, (uint8_t)(s1 >> 8), (uint8_t)(s1 >> 16), (uint8_t)(s1 >> 24),
                            (uint8_t)s2, (uint8_t)(s2 >> 8), (uint8_t)(s2 >> 16), (uint8_t)(s2 >> 24)};

    const size_t pix = (size_t)y * W + x;
    const float4 sg = *reinterpret_cast<const float4*>(seg + ((size_t)b * 3 + (s ? 2 : 0)) * H * W + pix);
    const float* tab_s = tab + ((size_t)b * 3 + 1 + s) * NCLS;
    // tf.ones_like overrides are ones everywhere, ignore-label pixels included: every frame for
    // -no_segmask (davo.py:1387), the tgt frame unless static_all (davo.py:1394,1411).
    float as[4] = {1.f, 1.f, 1.f, 1.f};
    if (v.att_source != 0) {
        as[0] = att_lookup(tab_s, sg.x); as[1] = att_lookup(tab_s, sg.y);
        as[2] = att_lookup(tab_s, sg.z); as[3] = att_lookup(tab_s, sg.w);
    }
    float at[4] = {1.f, 1.f, 1.f, 1.f};
    if (v.att_source == 3) {                                   // static_all: tgt frame is masked too
        const float4 tg = *reinterpret_cast<const float4*>(seg + ((size_t)b * 3 + 1) * H * W + pix);
        const float* tab_t = tab + (size_t)b * 3 * NCLS;
        at[0] = att_lookup(tab_t, tg.x); at[1] = att_lookup(tab_t, tg.y);
        at[2] = att_lookup(tab_t, tg.z); at[3] = att_lookup(tab_t, tg.w);
    }
    float fl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (v.cin_per_frame == 5) {
        const float4* fp = reinterpret_cast<const float4*>(flow + (((size_t)b * 4 + s) * H * W + pix) * 2);
        const float4 f0 = fp[0], f1 = fp[1];
        fl[0] = f0.x; fl[1] = f0.y; fl[2] = f0.z; fl[3] = f0.w;
        fl[4] = f1.x; fl[5] = f1.y; fl[6] = f1.z; fl[7] = f1.w;
    }
    constexpr int OUT_FLOATS = LD == 16 ? 8 : LD;          // LD 16 = split-fp16 layout, 8 floats' worth per pixel
    float* o = out + (((size_t)(b * 2 + s) * H + y) * W + x) * OUT_FLOATS;
    // LD 16 / 8: a thread's 4 pixels are 8 x 16 B, 128 B apart from its neighbour's - stored directly, every store
    // instruction would touch 64 separate lines.  The workgroup's 32 KB (contiguous: output offset = gid * 128 B)
    // go through LDS instead and leave as 1 KiB per wave instruction.  Unit t*8 + (k ^ (t&7)): 2-way bank conflicts
    // on the way in, none on the way out.  (Round 4: the float32 layout too - mask_pack<8> stored directly until then:
    // 46 us against mask_pack<16>'s 31 for the same bytes.)
    __shared__ float4 stage[STAGED ? 256 * 8 : 1];
#pragma unroll
    for (int px = 0; px < 4; ++px) {
        const float mt = v.mask_rgb ? at[px] : 1.f, ms = v.mask_rgb ? as[px] : 1.f;
        const float mi = v.mask_info ? as[px] : 1.f;
        float r[10];
        r[0] = u8_to_unit(tb[3 * px]); r[1] = u8_to_unit(tb[3 * px + 1]); r[2] = u8_to_unit(tb[3 * px + 2]);
        r[3] = u8_to_unit(sb[3 * px]); r[4] = u8_to_unit(sb[3 * px + 1]); r[5] = u8_to_unit(sb[3 * px + 2]);
        if (v.mask_rgb) {
            r[0] *= mt; r[1] *= mt; r[2] *= mt;
            r[3] *= ms; r[4] *= ms; r[5] *= ms;
        }
        r[6] = v.mask_info ? fl[2 * px] * mi : fl[2 * px];
        r[7] = v.mask_info ? fl[2 * px + 1] * mi : fl[2 * px + 1];
        if (LD == 16) {
            // split-fp16 form for the f16x3 convolutions: per pixel [8 hi halves | 8 lo halves]
            // (32 bytes, the same as 8 floats): x = hi + lo
            _Float16 hl[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const _Float16 h = (_Float16)r[k];
                hl[k] = h;
                hl[8 + k] = (_Float16)(r[k] - (float)h);
            }
            const int t8 = threadIdx.x * 8, sw = threadIdx.x & 7;
            stage[t8 + ((2 * px) ^ sw)] = *reinterpret_cast<const float4*>(&hl[0]);
            stage[t8 + ((2 * px + 1) ^ sw)] = *reinterpret_cast<const float4*>(&hl[8]);
        } else if (LD == 8) {
            const int t8 = threadIdx.x * 8, sw = threadIdx.x & 7;
            stage[t8 + ((2 * px) ^ sw)] = make_float4(r[0], r[1], r[2], r[3]);
            stage[t8 + ((2 * px + 1) ^ sw)] = make_float4(r[4], r[5], r[6], r[7]);
        } else {
            float* q = o + px * 10;
            q[0] = r[0]; q[1] = r[1]; q[2] = r[2]; q[3] = 0.f; q[4] = 0.f;
            q[5] = r[3]; q[6] = r[4]; q[7] = r[5]; q[8] = r[6]; q[9] = r[7];
        }
    }
    if (STAGED) {
        __syncthreads();
        float4* og = reinterpret_cast<float4*>(out) + (size_t)blockIdx.x * (256 * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = i * 256 + threadIdx.x;              // physical unit; owner thread t = j >> 3
            const int k = (j & 7) ^ ((j >> 3) & 7);           // its logical 16-byte piece
            if ((long)blockIdx.x * 256 + (j >> 3) < total) og[(j & ~7) + k] = stage[j];
        }
    }
}

// Pose head tail: pred (1x1, 256 -> 3, linear) + mean over H3 x W3 + 0.01 scale
// (nets/posenn.py:240-241,248-250).  pred and the mean are both linear, so
//   pose[n][head*3+j] = 0.01 * ( b[j] + (1/P) * sum_c ( sum_p cnv7[n][p][head][c] ) * Wp[c][j] ).
// Pass 1: grid (PH_SPLIT, 2B images, 2 heads), 256 threads = one per cnv7 channel (coalesced rows);
// block s sums its slice of the P pixels and writes 3 partial dot products.  Pass 2 (pose_finish)
// adds the PH_SPLIT partials in a fixed order -> bitwise reproducible.

__global__ __launch_bounds__(256) void pose_head_partial(const float* __restrict__ c7, int P,
                                                         const float* __restrict__ wpred /*[2][256][3]*/,
                                                         float* __restrict__ partial /*[2B][2][PH_SPLIT][3]*/) {
    const int sp = blockIdx.x, n = blockIdx.y, head = blockIdx.z, c = threadIdx.x;
    const int per = (P + PH_SPLIT - 1) / PH_SPLIT;
    const int p0 = sp * per, p1 = min(p0 + per, P);
    const float* src = c7 + (size_t)n * P * 512 + head * 256 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = p0;
    for (; p + 4 <= p1; p += 4) {
        s0 += src[(size_t)p * 512]; s1 += src[(size_t)(p + 1) * 512];
        s2 += src[(size_t)(p + 2) * 512]; s3 += src[(size_t)(p + 3) * 512];
    }
    for (; p < p1; ++p) s0 += src[(size_t)p * 512];
    const float sc = (s0 + s1) + (s2 + s3);
    const float* wp = wpred + ((size_t)head * 256 + c) * 3;
    float v[3] = {sc * wp[0], sc * wp[1], sc * wp[2]};
    __shared__ float red[3][4];
    const int lane = c & 63, wid = c >> 6;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        v[j] = wave_sum(v[j]);
        if (lane == 0) red[j][wid] = v[j];
    }
    __syncthreads();
    if (c < 3)
        partial[(((size_t)n * 2 + head) * PH_SPLIT + sp) * 3 + c] = (red[c][0] + red[c][1]) + (red[c][2] + red[c][3]);
}

__global__ __launch_bounds__(64) void pose_finish(const float* __restrict__ partial, int NB, int P,
                                                  const float* __restrict__ bpred /*[2][3]*/,
                                                  float* __restrict__ pose /*[2B][6]*/) {
    const int i = blockIdx.x * 64 + threadIdx.x;           // (n, head, j)
    if (i >= NB * 6) return;
    const int n = i / 6, hj = i - n * 6, head = hj / 3, j = hj - head * 3;
    const float* pp = partial + ((size_t)n * 2 + head) * PH_SPLIT * 3 + j;
    float tot = 0.f;
#pragma unroll
    for (int s = 0; s < PH_SPLIT; ++s) tot += pp[s * 3];
    pose[i] = 0.01f * (tot / (float)P + bpred[hj]);
}

// Pose from the per-tile partial sums cnv7's fused epilogue wrote (conv_igemm_h3.h, y_mode 2):
// partial[head][mtile][ntile][slot][k], slot 0 = the tile's first image, slot 1 = the next one.  One 64-lane wave per
// output (n, head, k): the tiles' terms are spread over the lanes and added by a fixed butterfly (pose_tail.h) — a
// single thread walking 56 dependent loads (128x32 tiles at batch 1) took 12 us.
// Range guard, device side (api.hip: tickets).  The batch's range record is complete when its last kernel runs (stream order), so
// that kernel can see the verdict the host will reach later: if a layer left the fp16-pair range the batch will be re-issued,
// and the inputs it was issued on are copied into the context's ring slot NOW - behind the kernels that read them, ahead of
// anything the caller orders behind the batch (e.g. the H2D of the next batch into the same buffers).  A batch in range -
// every batch of a well-ranged checkpoint - costs six loads per thread and no copy.  Only flow planes 0 and 1 are read by the
// path (davo.py:978-982): the first half of every window's block.
// phase 1 (top of the kernel, so that the memory-side round trips pass behind the kernel's own work): the maxima, and the first half of the mirror
struct RangePeek { unsigned w[6]; bool fails; };
__device__ __forceinline__ RangePeek range_peek(const SnapArgs& a, unsigned worker) {
    RangePeek r{};
    if (!a.record) return r;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        r.w[i] = __atomic_load_n(a.record + i, __ATOMIC_RELAXED);
        r.fails |= range_value_fails(__uint_as_float(r.w[i]));
    }
    if (worker == 0) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store((v4u){r.w[0], r.w[1], r.w[2], r.w[3]}, reinterpret_cast<v4u*>(a.host_mirror));
    }
    return r;
}
// phase 2 (end of the kernel)
__device__ __forceinline__ void snapshot_inputs_if_range_fails(const SnapArgs& a, const RangePeek& r, unsigned worker, unsigned nworkers) {
    if (!a.record) return;
    const bool copy = r.fails && a.s_img != nullptr;
    if (worker == 0) {
        // The host's verdict reads this mirror: no device-to-host copy, no event, no stream to wait for (a synchronous read per
        // batch made batch 1 host-bound, 0.124 -> 0.36 ms per window; an event per batch costs a marker on the GPU's queue).  Two
        // 16-byte stores to coherent host memory; the second carries the sequence number and is issued after the first has been
        // acknowledged, so a host that sees this batch's number sees this batch's maxima.
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_nontemporal_store((v4u){r.w[4], r.w[5], copy ? 1u : 0u, a.seq}, reinterpret_cast<v4u*>(a.host_mirror) + 1);
        if (copy) a.record[RANGE_SNAP] = 1u;
    }
    if (!copy) return;
    const uint4* si = reinterpret_cast<const uint4*>(a.img);
    const uint4* sf = reinterpret_cast<const uint4*>(a.flow);
    const uint4* ss = reinterpret_cast<const uint4*>(a.seg);
    uint4* di = reinterpret_cast<uint4*>(a.s_img);
    uint4* df = reinterpret_cast<uint4*>(a.s_flow);
    uint4* ds = reinterpret_cast<uint4*>(a.s_seg);
    const size_t n_img = (size_t)a.img_vec * a.B, n_seg = (size_t)a.seg_vec * a.B, n_flow = (size_t)a.flow_vec_half * a.B;
    for (size_t i = worker; i < n_img; i += nworkers) di[i] = si[i];
    for (size_t i = worker; i < n_seg; i += nworkers) ds[i] = ss[i];
    for (size_t i = worker; i < n_flow; i += nworkers) {
        const size_t b = i / a.flow_vec_half, o = i - b * a.flow_vec_half;
        df[b * a.flow_vec + o] = sf[b * a.flow_vec + o];
    }
}

__global__ __launch_bounds__(64) void pose_from_tiles(const float* __restrict__ partial, int NB, int P, int bm,
                                                      int mtiles, int ntiles_n, const float* __restrict__ bpred,
                                                      float* __restrict__ pose /*[2B][6]*/, SnapArgs snap) {
    const int i = blockIdx.x;                              // (n, head, k)
    const int n = i / 6, hk = i - n * 6;
    const RangePeek peek = range_peek(snap, blockIdx.x * 64 + threadIdx.x);
    const float tot = pose_tile_sum<false>(partial, n, hk, P, bm, mtiles, ntiles_n, threadIdx.x);
    if (threadIdx.x == 0) pose[i] = 0.01f * (tot / (float)P + bpred[hk]);
    snapshot_inputs_if_range_fails(snap, peek, blockIdx.x * 64 + threadIdx.x, gridDim.x * 64);
}

// the same guard as a launch of its own, behind the pose heads that are not pose_from_tiles ("fuse_pose" 0, "fold_tails" 1, tiny maps)
__global__ __launch_bounds__(256) void range_guard_snapshot(SnapArgs snap) {
    const RangePeek peek = range_peek(snap, blockIdx.x * 256 + threadIdx.x);
    snapshot_inputs_if_range_fails(snap, peek, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// Split-K fix-up (forward.hip: small batches): out[m][n] = stored(relu(part[m][0][n] + part[m][1][n] + ...)) in the f16x3
// activation layout (per pixel, per 32 channels: 32 hi halves | 32 lo halves).  The partial sums carry the layer's
// out_scale already; fixed order of the S terms.  One thread per channel pair; N a multiple of 32.
__global__ __launch_bounds__(256) void splitk_fixup(const float* __restrict__ part, long total_pairs, int N, int S, int relu,
                                                    uint8_t* __restrict__ y, unsigned* __restrict__ range) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    float vmax = 0.f;
    if (idx < total_pairs) {
        const int half_n = N >> 1;
        const long m = idx / half_n;
        const int n = (int)(idx - m * half_n) * 2;
        const float2* src = reinterpret_cast<const float2*>(part + (m * S) * N + n);
        float2 v = src[0];
        for (int s = 1; s < S; ++s) {
            const float2 t = src[(long)s * half_n];
            v.x += t.x; v.y += t.y;
        }
        const float lo_clamp = relu ? 0.f : -65504.f;
        v.x = fmaxf(v.x, lo_clamp); v.y = fmaxf(v.y, lo_clamp);
        vmax = fmaxf(fabsf(v.x), fabsf(v.y));
        v.x = fminf(v.x, 65504.f); v.y = fminf(v.y, 65504.f);
        const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y;
        const _Float16 l0 = (_Float16)(v.x - (float)h0), l1 = (_Float16)(v.y - (float)h1);
        uint8_t* o = y + m * (long)N * 4 + (n >> 5) * 128 + (n & 31) * 2;
        *reinterpret_cast<unsigned*>(o) = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
        *reinterpret_cast<unsigned*>(o + 64) = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
    }
    if (range) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        range_note(range, vmax, (threadIdx.x & 63) == 0);
    }
}

// ---- on-device cross-check (impl 1): one thread per output element, reference layouts ----
__global__ __launch_bounds__(256) void conv_direct(const float* __restrict__ x, int N, int Hin, int Win, int Cin,
                                                   int x_ld, int x_coff,
                                                   const float* __restrict__ w /*HWIO*/, int KS, int Cout,
                                                   const float* __restrict__ bias, int stride, int rate,
                                                   int pad_t, int pad_l, int Hout, int Wout, int relu,
                                                   float* __restrict__ y, int y_ld, int y_coff) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * Hout * Wout * Cout;
    if (gid >= total) return;
    const int co = (int)(gid % Cout);
    long m = gid / Cout;
    const int ox = (int)(m % Wout); long t = m / Wout;
    const int oy = (int)(t % Hout); const int n = (int)(t / Hout);
    float acc = 0.f;
    for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy * stride - pad_t + ky * rate;
        if ((unsigned)iy >= (unsigned)Hin) continue;
        for (int kx = 0; kx < KS; ++kx) {
            const int ix = ox * stride - pad_l + kx * rate;
            if ((unsigned)ix >= (unsigned)Win) continue;
            const float* xp = x + ((size_t)(n * Hin + iy) * Win + ix) * x_ld + x_coff;
            const float* wp = w + ((size_t)(ky * KS + kx) * Cin) * Cout + co;
            for (int ci = 0; ci < Cin; ++ci) acc += xp[ci] * wp[(size_t)ci * Cout];
        }
    }
    acc += bias[co];
    if (relu) acc = fmaxf(acc, 0.f);
    y[m * y_ld + y_coff + co] = acc;
}

}  // namespace davo
