#!/usr/bin/env python
"""Experiment: two contexts (two HIP streams, two workspaces) on one GPU, alternating batches, versus one
context — does overlapping the small kernels of one batch with the big convolutions of the other pay?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = parse_version(FLAGSHIP_VERSION)
w = synth.make_weights(cfg)
img, flow, seg = synth.make_inputs(4, 128, 416)
reps = B // 4
img, flow, seg = np.tile(img, (reps, 1, 1, 1)), np.tile(flow, (reps, 1, 1, 1, 1)), np.tile(seg, (reps, 1, 1, 1, 1))


def mk():
    e = Engine(cfg, 128, 416, B)
    e.load_weights(w)
    bufs = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
    return e, bufs


def run(engines, steps):
    for e, b in engines:
        for _ in range(3):
            e.forward_device(B, *b)
        e.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        e, b = engines[i % len(engines)]
        e.forward_device(B, *b)
    for e, _ in engines:
        e.synchronize()
    return B * steps / (time.perf_counter() - t0)


one = [mk()]
two = [one[0], mk()]
three = two + [mk()]
for rnd in range(3):
    print("round %d: 1 ctx %.0f | 2 ctx %.0f | 3 ctx %.0f triplets/s" % (rnd, run(one, 40), run(two, 40), run(three, 42)))
