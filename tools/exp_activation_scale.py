#!/usr/bin/env python
"""Error of the two arithmetic modes when one layer's activations are scaled by 2^s: cnv3
(weights, bias) * 2^s, cnv4 weights * 2^-s — the same network by ReLU homogeneity."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION   # noqa: E402
from oracle import c_oracle                                           # noqa: E402  (checker only)

cfg = parse_version(FLAGSHIP_VERSION)
img, flow, seg = synth.make_inputs(4, 128, 416)
weights = synth.make_weights(cfg)
want = c_oracle.forward(cfg, img, flow, seg, weights)
for shift in (0, -4, -8, -12, -16, -20, 4, 8, 10):
    k = np.float32(2.0 ** shift)
    w2 = dict(weights)
    w2["pose_exp_net/cnv3/weights"] = weights["pose_exp_net/cnv3/weights"] * k
    w2["pose_exp_net/cnv3/biases"] = weights["pose_exp_net/cnv3/biases"] * k
    w2["pose_exp_net/cnv4/weights"] = weights["pose_exp_net/cnv4/weights"] / k
    row = []
    for precision in ("f16x3", "f32"):
        e = Engine(cfg, 128, 416, 4)
        e.load_weights(w2)
        e.set_precision(precision)
        row.append(np.abs(e.forward(img, flow, seg) - want).max())
        e.close()
    print("cnv3 activations x 2^%-3d  max abs err  f16x3 %.3g   f32 %.3g   (max|ref| %.3g)" % (shift, row[0], row[1], np.abs(want).max()), flush=True)
