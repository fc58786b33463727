#!/usr/bin/env python
"""Tuning build only (DAVO_LIB_SUFFIX=_tuning): what do the epilogue and the staging of cnv4..cnv7 cost at B = 32 under the default plan?
Arms interleaved in one process: base | no epilogue (dbg 32) | no DMA issue (dbg 1) | both.

    python tools/build_variant.py _tuning -DDAVO_TUNING && DAVO_LIB_SUFFIX=_tuning python tools/exp/epilogue_ablation.py [--batch 32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--options", default="")
a = ap.parse_args()
cfg = parse_version(FLAGSHIP_VERSION)
B, H, W = a.batch, 128, 416
e = Engine(cfg, H, W, B)
e.load_weights(synth.make_weights(cfg))
nu = min(8, B)
img, flow, seg = synth.make_inputs(nu, H, W)
r = -(-B // nu)
img, flow, seg = np.tile(img, (r, 1, 1, 1))[:B], np.tile(flow, (r, 1, 1, 1, 1))[:B], np.tile(seg, (r, 1, 1, 1, 1))[:B]
d = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
for kv in filter(None, a.options.split(",")):
    k, v = kv.split("=")
    e.set_option(k, int(v))
arms = [("base", 0), ("no epilogue", 32), ("no DMA", 1), ("neither", 33)]
res = {n: {} for n, _ in arms}
for rnd in range(4):
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
keys = [k for k in res["base"] if k.startswith("cnv")]
print("B=%d; ms per launch, median of 4 rounds" % B)
print("%-14s" % "arm" + "".join(" %9s" % k for k in keys))
for name, _ in arms:
    print("%-14s" % name + "".join(" %9.4f" % np.median(res[name].get(k, [float("nan")])) for k in keys))
print("plans: " + "; ".join("cnv%d %s" % (li + 1, e.last_plan(li)) for li in range(3, 7)))
