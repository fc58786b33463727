#!/usr/bin/env python
"""One batch in flight or two?  Interleaved A/B in ONE process (arms alternate every round), wall time per step.

    python tools/exp/pipelined_ab.py [--batch 32] [--rounds 10] [--steps 40]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
cfg = parse_version(FLAGSHIP_VERSION)
B, H, W = a.batch, 128, 416
e = Engine(cfg, H, W, B)
e.load_weights(synth.make_weights(cfg))
sets = []
for k in range(2):
    nu = min(8, B)
    img, flow, seg = synth.make_inputs(nu, H, W, first_window=k * B)
    r = -(-B // nu)
    img, flow, seg = np.tile(img, (r, 1, 1, 1))[:B], np.tile(flow, (r, 1, 1, 1, 1))[:B], np.tile(seg, (r, 1, 1, 1, 1))[:B]
    sets.append((e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48)))
e.set_inflight(2)
for i in range(100):                                                 # clock settle
    e.forward_device(B, *sets[i % 2])
e.synchronize()
res = {1: [], 2: []}
for rnd in range(a.rounds):
    for n in ((1, 2) if rnd % 2 == 0 else (2, 1)):
        e.set_inflight(n)
        for i in range(6):
            e.forward_device(B, *sets[i % 2])
        e.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            e.forward_device(B, *sets[i % 2])
        e.synchronize()
        res[n].append((time.perf_counter() - t0) / a.steps * 1e3)
for n in (1, 2):
    v = np.array(res[n])
    print("B=%d, %d in flight: ms/step median %.4f  mean %.4f  min %.4f  max %.4f  (%d rounds of %d steps) -> %.0f triplets/s"
          % (B, n, np.median(v), v.mean(), v.min(), v.max(), a.rounds, a.steps, B / np.median(v) * 1e3))
d = np.array(res[1]) / np.array(res[2])
print("B=%d: per-round ratio (1 in flight / 2 in flight) median %.4f, min %.4f, max %.4f; two in flight wins %d of %d rounds"
      % (B, np.median(d), d.min(), d.max(), int((d > 1).sum()), a.rounds))
e.close()
