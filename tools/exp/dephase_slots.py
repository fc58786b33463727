#!/usr/bin/env python
"""Tuning build only: co-resident workgroups of cnv2 (3 per CU) / cnv4 (2 per CU) started out of phase (dbg 8192).

    DAVO_LIB_SUFFIX=_tuning python tools/exp/dephase_slots.py
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
arms = [("base", 0)]
for slots in (2,):
    for n in (2, 4):
        arms.append(("%d slots, +%d us/slot" % (slots, 4 * n), 8192 | (n << 16) | (slots << 24)))
for bit, nm in ((16384, "odd in XCD"), (32768, "odd id")):
    for n in (1, 2, 3, 4, 6):
        arms.append(("%s +%d us" % (nm, 4 * n), bit | (n << 16)))
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
keys = ["cnv2", "cnv3", "cnv3.rem", "cnv4", "cnv5", "cnv5.rem", "cnv6", "cnv6.rem"]
print("%-22s" % "arm" + "".join(" %9s" % k for k in keys))
for name, _ in arms:
    print("%-22s" % name + "".join(" %9.4f" % (np.median(res[name][k]) if k in res[name] else float("nan")) for k in keys))
