#!/usr/bin/env python
"""Which tensor differs between repeated forwards (or between two option settings) of one engine: run on the GPU box.
    python tools/exp/determinism_probe.py --batch 4 --tile 0 [--toggle deep_ring]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--height", type=int, default=128)
ap.add_argument("--width", type=int, default=416)
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--toggle", default="")
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
cfg = parse_version(FLAGSHIP_VERSION)
B, H, W = a.batch, a.height, a.width
img, flow, seg = synth.make_inputs(B, H, W)
e = Engine(cfg, H, W, B)
e.load_weights(synth.make_weights(cfg))
e.set_option("force_tile", a.tile)
shapes = {"packed": (2 * B, H, W, 8), "cnv1": (2 * B, (H + 1) // 2, (W + 1) // 2, 16), "cnv2": (2 * B, (H + 3) // 4, (W + 3) // 4, 32),
          "cnv3": (2 * B, (H + 3) // 4, (W + 3) // 4, 64), "cnv4": (2 * B, (H + 3) // 4, (W + 3) // 4, 128),
          "cnv5": (2 * B, (H + 3) // 4, (W + 3) // 4, 256), "cnv6": (2 * B, (H + 3) // 4, (W + 3) // 4, 256)}
ref = None
for r in range(a.reps):
    if a.toggle:
        e.set_option(a.toggle, r & 1)
    pose = e.forward(img, flow, seg).copy()
    cur = {k: e.debug_read(k, s).copy() for k, s in shapes.items()}
    cur["pose"] = pose
    cur["plans"] = [e.last_plan(li) for li in range(7)]
    if ref is None:
        ref = cur
        print("plans", cur["plans"])
        continue
    diff = [k for k in list(shapes) + ["pose"] if not np.array_equal(cur[k], ref[k])]
    print("rep %d%s: differs in %s" % (r, " (%s=%d)" % (a.toggle, r & 1) if a.toggle else "", diff or "nothing"),
          {k: float(np.abs(cur[k] - ref[k]).max()) for k in diff})
e.close()
