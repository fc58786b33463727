#!/usr/bin/env python
"""Tuning build only: what bounds the patch kernels (cnv1..cnv3)?  DAVO_PDBG: 1 = patch loads read the zero line, 2 = no stores.

    DAVO_LIB_SUFFIX=_tuning python tools/exp/patch_ablation.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402

cfg = parse_version(FLAGSHIP_VERSION)
B, H, W = 32, 128, 416
e = Engine(cfg, H, W, B)
e.load_weights(synth.make_weights(cfg))
img, flow, seg = synth.make_inputs(8, H, W)
img, flow, seg = np.tile(img, (4, 1, 1, 1)), np.tile(flow, (4, 1, 1, 1, 1)), np.tile(seg, (4, 1, 1, 1, 1))
d = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
arms = [("base", 0), ("zero-line loads", 1), ("no stores", 2), ("both", 3)]
res = {a[0]: {} for a in arms}
for rnd in range(3):
    for name, dbg in arms:
        os.environ["DAVO_PDBG"] = str(dbg)
        for _ in range(3):
            e.forward_device(B, *d)
        e.synchronize()
        e.profile(1)
        e.profile_reset()
        for _ in range(8):
            e.forward_device(B, *d)
        for k, (n, ms) in e.profile_entries().items():
            res[name].setdefault(k, []).append(ms / max(n, 1))
        e.profile(0)
keys = ["mask_pack", "cnv1", "cnv2", "cnv3", "cnv4"]
print("%-20s" % "arm" + "".join(" %9s" % k for k in keys))
for name, _ in arms:
    print("%-20s" % name + "".join(" %9.4f" % np.median(res[name][k]) for k in keys))
