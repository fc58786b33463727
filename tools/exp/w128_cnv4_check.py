#!/usr/bin/env python
"""cnv4 on conv_igemm_h3w128 (option "wave128" 3) against conv_igemm_h3's tiles: activation and poses bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402

cfg = parse_version(FLAGSHIP_VERSION)
w = synth.make_weights(cfg)
bad = 0
for H, W, B in ((128, 416, 32), (128, 416, 40), (256, 832, 8), (128, 416, 128)):
    img, flow, seg = synth.make_inputs(min(B, 8), H, W, first_window=2)
    reps = -(-B // img.shape[0])
    img, flow, seg = (np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:B] for a in (img, flow, seg))
    e = Engine(cfg, H, W, B)
    e.load_weights(w)
    e.set_option("host_chunk", 0)
    e.set_option("fuse_pose", 0)
    out, act = {}, {}
    for opt in (0, 3):
        e.set_option("wave128", opt)
        out[opt] = e.forward(img, flow, seg).copy()
        act[opt] = e.debug_read("cnv4", (2 * B, H // 4, W // 4, 128)).copy()
        plan = e.last_plan(3)
    same = np.array_equal(out[0], out[3]) and np.array_equal(act[0], act[3])
    bad += not same
    print("%dx%d B=%d: %s (cnv4 plan with wave128=3: %s; max |act diff| %.3g)" % (H, W, B, "bit-identical" if same else "DIFFERENT", plan,
          float(np.abs(act[0] - act[3]).max())), flush=True)
    e.close()
print("CNV4_CHECK", "OK" if not bad else "FAILED")
sys.exit(1 if bad else 0)
