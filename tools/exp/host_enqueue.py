#!/usr/bin/env python
"""Is the device path host-bound at this batch?  Time to ENQUEUE n forwards (the loop alone) vs time until they are done.
    python tools/exp/host_enqueue.py [--batch 1] [--n 400]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.getcwd())
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--n", type=int, default=400)
a = ap.parse_args()
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402

cfg = parse_version(FLAGSHIP_VERSION)
B, H, W = a.batch, 128, 416
e = Engine(cfg, H, W, B)
e.load_weights(synth.make_weights(cfg))
img, flow, seg = synth.make_inputs(B, H, W)
d = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
for _ in range(200):
    e.forward_device(B, *d)
e.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(a.n):
        e.forward_device(B, *d)
    t1 = time.perf_counter()
    e.synchronize()
    t2 = time.perf_counter()
    print("B=%d: enqueue %.1f us per batch, done after %.1f us per batch" % (B, (t1 - t0) / a.n * 1e6, (t2 - t0) / a.n * 1e6), flush=True)
e.close()
