#!/usr/bin/env python
"""Maximum-size check on the GPU: 176 triplets of 256x832 in one batch — the cnv5 / cnv6 activations are 4.8 GB each,
past 2^32 bytes, so every 64-bit address path is exercised — with every window compared against the C oracle
(the batch repeats 8 oracle-checked windows 22 times).  ~14 GB of HBM, a few seconds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION   # noqa: E402
from oracle import c_oracle                                           # noqa: E402  (checker only)

cfg = parse_version(FLAGSHIP_VERSION)
w = synth.make_weights(cfg)
H, W, R = 256, 832, 22
B = 8 * R
img8, flow8, seg8 = synth.make_inputs(8, H, W)
img, flow, seg = np.tile(img8, (R, 1, 1, 1)), np.tile(flow8, (R, 1, 1, 1, 1)), np.tile(seg8, (R, 1, 1, 1, 1))
want = c_oracle.forward(cfg, img8, flow8, seg8, w)
rc = 0
for prec in ("f16x3", "f32"):
    e = Engine(cfg, H, W, B)
    e.load_weights(w)
    e.set_precision(prec)
    got = e.forward(img, flow, seg)
    e.close()
    d = [float(np.abs(got[k * 8:(k + 1) * 8] - want).max()) for k in range(R)]
    print("%s B=%d %dx%d: max abs err over all windows %.3g (first block %.3g, last block %.3g)" % (prec, B, H, W, max(d), d[0], d[-1]), flush=True)
    rc |= int(max(d) > 1e-4)
sys.exit(rc)
