#!/usr/bin/env python
"""Tuning build only (DAVO_LIB_SUFFIX=_tuning): do the store bursts of a launch cost less when half of the chip's
workgroups run out of phase with the other half?  cnv5..cnv7 forced onto the 208x256 tile (whole rounds at B=32);
half of the first round's workgroups start N x ~4 us late (conv_igemm_h3s.h, dbg bits 1024 / 2048).

    DAVO_LIB_SUFFIX=_tuning python tools/exp/dephase.py
"""
import os
import sys
import time

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
arms = [("base", 0)]
for bit in (1024, 2048):
    for n in (8, 15, 23):
        arms.append(("%s +%dus" % ("xcd-parity" if bit == 1024 else "in-xcd", 4 * n), bit | (n << 16)))
arms.append(("no stores", 64))
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
