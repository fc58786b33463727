#!/usr/bin/env python
"""Error of the two arithmetic modes against the C oracle when the dynamic range INSIDE a layer is wide:
  * per-channel: every other output channel of cnv3 and of cnv5 x 2^-s, the consumer's matching input-channel weights x 2^s
    (the same network by ReLU homogeneity), with and without calibration;
  * heavy tails: n weights per layer x t (another network; the oracle runs on the same weights).
Run on the GPU box: python tools/exp_dynamic_range.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION   # noqa: E402
from oracle import c_oracle                                           # noqa: E402  (checker only)

cfg = parse_version(FLAGSHIP_VERSION)
B, H, W = 2, 128, 416
img, flow, seg = synth.make_inputs(B, H, W)
weights = synth.make_weights(cfg)
want = c_oracle.forward(cfg, img, flow, seg, weights)
scale = np.abs(want).max()


def scale_channels(w, producer, consumers, idx, shift):
    k = np.float32(2.0 ** shift)
    w2 = dict(w)
    pw, pb = w["pose_exp_net/%s/weights" % producer].copy(), w["pose_exp_net/%s/biases" % producer].copy()
    pw[..., idx] *= k
    pb[idx] *= k
    w2["pose_exp_net/%s/weights" % producer], w2["pose_exp_net/%s/biases" % producer] = pw, pb
    for c in consumers:
        cw = w["pose_exp_net/%s/weights" % c].copy()
        cw[:, :, idx, :] /= k
        w2["pose_exp_net/%s/weights" % c] = cw
    return w2


def run(w, precision, calibrate=False):
    e = Engine(cfg, H, W, B)
    e.load_weights(w)
    e.set_precision(precision)
    if calibrate:
        e.calibrate(img, flow, seg)
    got = e.forward(img, flow, seg)
    st = e.range_stats()
    e.close()
    return got, st


print("max|ref| %.4g; bar 1e-4 abs and 1e-4 relative to max|ref|" % scale)
for s in (0, 6, 10, 14, 18, 22):
    w2 = scale_channels(weights, "cnv3", ["cnv4"], np.arange(0, 64, 2), -s)
    w2 = scale_channels(w2, "cnv5", ["pose/rotation/cnv6", "pose/translation/cnv6"], np.arange(1, 256, 2), -s)
    a, sa = run(w2, "f16x3")
    b, sb = run(w2, "f16x3", True)
    c, _ = run(w2, "f32")
    print("per-channel 2^-%-2d  max abs err: f16x3 %.3g (f32 batches %d)  calibrated %.3g  f32 %.3g" %
          (s, np.abs(a - want).max(), sa["f32_batches"], np.abs(b - want).max(), np.abs(c - want).max()), flush=True)
rng = np.random.RandomState(7)
for n, t in ((6, 100.0), (6, 1000.0), (64, 100.0)):
    w2 = dict(weights)
    for name in list(w2):
        if name.endswith("/weights") and "/pred/" not in name:
            w = w2[name].copy()
            flat = w.reshape(-1)
            flat[rng.choice(flat.size, n, replace=False)] *= t
            w2[name] = w
    ref = c_oracle.forward(cfg, img, flow, seg, w2)
    a, sa = run(w2, "f16x3")
    c, _ = run(w2, "f32")
    print("heavy tails %d x %g per layer: max|ref| %.4g  max abs err: f16x3 %.3g (recal %d, f32 batches %d)  f32 %.3g" %
          (n, t, np.abs(ref).max(), np.abs(a - ref).max(), sa["recalibrations"], sa["f32_batches"], np.abs(c - ref).max()), flush=True)
