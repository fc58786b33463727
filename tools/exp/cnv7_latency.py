#!/usr/bin/env python
"""Tuning build only: how much of cnv5..cnv7's time on the 208x256 tile is the latency of pixel loads that miss L2?
Arms: base; dbg 4096 = every tile reads image 0 (same DMA instructions, input L2-resident); dbg 1 = DMA reads the zero line.

    DAVO_LIB_SUFFIX=_tuning python tools/exp/cnv7_latency.py
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
e.set_option("force_tile", 6)
arms = [("base", 0), ("input in L2", 4096), ("zero-line DMA", 1), ("no stores", 64), ("L2 + no stores", 4096 | 64)]
res = {a[0]: {} for a in arms}
for rnd in range(3):
    for name, dbg in arms:
        os.environ["DAVO_DBG"] = str(dbg)
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
print("%-22s %9s %9s %9s" % ("arm", "cnv5", "cnv6", "cnv7"))
for name, _ in arms:
    print("%-22s" % name + "".join(" %9.4f" % np.median(res[name][k]) for k in ("cnv5", "cnv6", "cnv7")))
