"""CPU ORACLE (test infrastructure, not product code) — numpy restatement of DAVO's
frame-to-frame pose inference path.

PARITY UNPINNED: the reference's arithmetic lives in tensorflow-gpu==1.13.1
(requirements.txt:1), which is absent offline (``import tensorflow`` ->
ModuleNotFoundError), the reference ships no test, golden vector or checkpoint for this
path, so this restatement cannot be checked against TF outputs.  It
is pinned instead by agreement of three independent implementations: this file
(float64, tap-by-tap matmul), oracle/davo_oracle.c (float32, direct loops) and
torch.nn.functional.conv2d on CPU with explicit asymmetric padding (tests/test_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

What is restated (all paths relative to /root/reference):
  davo.py:1519-1522   preprocess_image            u8 -> f32 * (1/255) * 2 - 1
  data_loader.py:537-557 batch_unpack_image_sequence  strip = src0 | tgt | src1
  davo.py:978-982, 998-1004  flow planes 0,1; seg planes (tgt,src0,src1) = file planes (1,0,2)
  davo.py:1088-1102   SE input transform (norm / abs_h / abs_v / abs)
  nets/attention_module.py:54-103  se(mode='gp'): mean_{h,w} -> dense(8,act) -> dense(19,sigmoid)
  davo.py:1115,1178   one_hot(int32(seg),19) . weights  == LUT gather, out-of-range id -> 0
  nets/posenn.py:380-394  build_seg_channel_weight (static attention variants)
  davo.py:1404-1442   masking + per-frame concat(rgb, info)
  nets/posenn.py:189-254  decouple_sharednet_v0_dilation (slim.conv2d: SAME, +bias, ReLU)
  davo.py:1453-1458   two calls with shared weights, concat on axis -2 -> [B,2,6]
TF semantics restated because the reference only composes TF ops: SAME padding
(SURVEY note P), float->int32 cast truncates toward zero, tf.one_hot of an
out-of-range index is an all-zero row.
"""
import numpy as np

NUM_SEG_CLASSES = 19


def same_pad(in_size, k, stride, rate):
    """TF 'SAME': out = ceil(in/stride); total pad = max((out-1)*stride + k_eff - in, 0);
    before = total // 2 (the smaller half goes first)."""
    out = -(-in_size // stride)
    k_eff = (k - 1) * rate + 1
    total = max((out - 1) * stride + k_eff - in_size, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, w, b, stride=1, rate=1, relu=True):
    """slim.conv2d(padding='SAME') on NHWC x with HWIO w: conv + bias (+ ReLU)
    (nets/posenn.py:205-215,238-240)."""
    N, H, W, Cin = x.shape
    kh, kw, ci, Cout = w.shape
    assert ci == Cin
    Ho, pt, pb = same_pad(H, kh, stride, rate)
    Wo, pl, pr = same_pad(W, kw, stride, rate)
    xp = np.zeros((N, H + pt + pb, W + pl + pr, Cin), x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    y = np.zeros((N, Ho, Wo, Cout), x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            y0, x0 = ky * rate, kx * rate
            patch = xp[:, y0:y0 + (Ho - 1) * stride + 1:stride, x0:x0 + (Wo - 1) * stride + 1:stride, :]
            y += patch.reshape(-1, Cin).dot(w[ky, kx].astype(x.dtype)).reshape(N, Ho, Wo, Cout)
    y += b.astype(x.dtype)
    if relu:
        np.maximum(y, 0, out=y)
    return y


def _act(name, x):
    if name == "tanh":
        return np.tanh(x)
    if name == "lrelu":
        return np.where(x > 0, x, 0.2 * x)            # tf.nn.leaky_relu default alpha
    return np.maximum(x, 0)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def preprocess_image(img_u8, dtype):
    """davo.py:1519-1522; convert_image_dtype(u8->f32) multiplies by float32(1/255)."""
    if dtype == np.float32:
        return img_u8.astype(np.float32) * np.float32(1.0 / 255.0) * np.float32(2.0) - np.float32(1.0)
    return img_u8.astype(dtype) * (1.0 / 255.0) * 2.0 - 1.0


def attention_tables(cfg, flow, weights, dtype):
    """[B,3,19] class-attention tables for (tgt, src0, src1).

    se_flow: davo.py:1175-1180 + :1408-1412 (tgt table := ones);
    static:  nets/posenn.py:380-394, davo.py:1390-1400; ones: davo.py:1385-1389."""
    B = flow.shape[0]
    tab = np.ones((B, 3, NUM_SEG_CLASSES), dtype)
    if cfg.att_source == "se_flow":
        w1 = weights["pose_exp_net/se_flow/bottleneck_fc/kernel"].astype(dtype)
        b1 = weights["pose_exp_net/se_flow/bottleneck_fc/bias"].astype(dtype)
        w2 = weights["pose_exp_net/se_flow/recover_fc/kernel"].astype(dtype)
        b2 = weights["pose_exp_net/se_flow/recover_fc/bias"].astype(dtype)
        for s in range(2):
            f = flow[:, s].astype(dtype)                                  # [B,H,W,2]
            if cfg.norm_flow:
                f = (f - dtype(0.32140523)) / dtype(15.384229)
            if cfg.abs_mode == "h":
                f = np.stack([np.abs(f[..., 0]), f[..., 1]], -1)
            elif cfg.abs_mode == "v":
                f = np.stack([f[..., 0], np.abs(f[..., 1])], -1)
            elif cfg.abs_mode == "all":
                f = np.abs(f)
            sq = f.astype(np.float64).mean(axis=(1, 2)).astype(dtype)     # attention_module.py:66
            e = _act(cfg.se_act, sq.dot(w1) + b1)
            tab[:, 1 + s] = _sigmoid(e.dot(w2) + b2)
    elif cfg.att_source in ("static_src", "static_all"):
        w = _sigmoid(weights["pose_exp_net/pose_exp_net/seg_channel_weight/weight"].astype(dtype))
        tab[:, 1] = w
        tab[:, 2] = w
        if cfg.att_source == "static_all":
            tab[:, 0] = w
    return tab


def attention_map(table, seg):
    """sum_c one_hot(int32(seg),19)[...,c] * table[c]  (davo.py:1115,1178)."""
    s = seg[..., 0]
    # tf.cast(float -> int32) truncates toward zero; what it gives for NaN / inf / beyond int32 is the platform's
    # (INT_MIN on x86, 0 or saturation on GPUs).  Pinned here and in the product: anything that is not a finite
    # value in (-1, 19) selects no class (zero row of the one_hot), like every other out-of-range id.
    inside = np.isfinite(s) & (s > -1.0) & (s < float(NUM_SEG_CLASSES))
    ids = np.trunc(np.where(inside, s, -1.0)).astype(np.int64)            # [B,H,W]
    ok = inside & (ids >= 0) & (ids < NUM_SEG_CLASSES)
    safe = np.where(ok, ids, 0)
    att = np.take_along_axis(table[:, None, None, :], safe[..., None], axis=-1)[..., 0]
    return np.where(ok, att, 0).astype(table.dtype)[..., None]           # [B,H,W,1]


def pack_inputs(cfg, img_u8, flow, seg, weights, dtype=np.float64):
    """Masked, concatenated PoseNN inputs for the two pairs: [B,2,H,W,2*cin_per_frame]
    (davo.py:961-1004, 1404-1442; nets/posenn.py:198)."""
    B, H, W3, _ = img_u8.shape
    W = W3 // 3
    x = preprocess_image(img_u8, dtype)
    src0, tgt, src1 = x[:, :, :W], x[:, :, W:2 * W], x[:, :, 2 * W:]       # data_loader.py:537-557
    tab = attention_tables(cfg, flow, weights, dtype)
    ones = np.ones((B, H, W, 1), dtype)
    # seg file order is (src0,tgt,src1).  tf.ones_like(map) overrides are ones EVERYWHERE,
    # ignore-label pixels included: tgt for se_flow (davo.py:1411) and static_src (:1394),
    # all three frames for -no_segmask (:1387).
    att_tgt = attention_map(tab[:, 0], seg[:, 1]) if cfg.att_source == "static_all" else ones
    if cfg.att_source == "ones":
        att = [ones, ones]
    else:
        att = [attention_map(tab[:, 1], seg[:, 0]), attention_map(tab[:, 2], seg[:, 2])]
    out = np.zeros((B, 2, H, W, 2 * cfg.cin_per_frame), dtype)
    c = cfg.cin_per_frame
    for s, src in enumerate((src0, src1)):
        t_rgb = tgt * att_tgt if cfg.mask_rgb else tgt
        s_rgb = src * att[s] if cfg.mask_rgb else src
        out[:, s, ..., 0:3] = t_rgb
        out[:, s, ..., c:c + 3] = s_rgb
        if cfg.use_flow_info:
            f = flow[:, s].astype(dtype)                                   # raw flow (davo.py:1064)
            out[:, s, ..., c + 3:c + 5] = f * att[s] if cfg.mask_info else f
            # tgt "flow" is zeros (davo.py:979) -> channels 3,4 stay zero
    return out


_TRUNK = (("cnv1", 2, 1), ("cnv2", 2, 1), ("cnv3", 1, 2), ("cnv4", 1, 4), ("cnv5", 1, 8))


def posenet(x, weights, dtype=np.float64, keep=None):
    """decouple_sharednet_v0_dilation (nets/posenn.py:189-254): x [N,H,W,10] -> [N,6]."""
    h = x.astype(dtype)
    for name, stride, rate in _TRUNK:
        h = conv2d_same(h, weights["pose_exp_net/%s/weights" % name],
                        weights["pose_exp_net/%s/biases" % name], stride, rate)
        if keep is not None:
            keep[name] = h
    outs = []
    for head in ("rotation", "translation"):
        p = "pose_exp_net/pose/%s/" % head
        c6 = conv2d_same(h, weights[p + "cnv6/weights"], weights[p + "cnv6/biases"], 1, 2)
        c7 = conv2d_same(c6, weights[p + "cnv7/weights"], weights[p + "cnv7/biases"], 2, 1)
        pred = conv2d_same(c7, weights[p + "pred/weights"], weights[p + "pred/biases"], 1, 1, relu=False)
        if keep is not None:
            keep[head + "/cnv6"], keep[head + "/cnv7"], keep[head + "/pred"] = c6, c7, pred
        outs.append(pred.astype(np.float64).mean(axis=(1, 2)).astype(dtype))   # posenn.py:241
    return dtype(0.01) * np.concatenate(outs, axis=-1)                    # posenn.py:248-250


def forward(cfg, img_u8, flow, seg, weights, dtype=np.float64, keep=None):
    """DAVO.inference(mode='pose') (davo.py:1553-1569): -> float [B,2,6]."""
    packed = pack_inputs(cfg, img_u8, flow, seg, weights, dtype)
    B, _, H, W, C = packed.shape
    if keep is not None:
        keep["packed"] = packed
    poses = posenet(packed.reshape(B * 2, H, W, C), weights, dtype, keep)
    return poses.reshape(B, 2, 6)
