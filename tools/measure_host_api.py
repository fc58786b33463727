#!/usr/bin/env python
"""PCIe-inclusive rate of the host-buffer entry point davo_forward: numpy arrays in
pageable host memory -> poses on the host, B=32, 128x416.  Not the bench metric (bench.py keeps the
inputs resident in HBM); reported for information."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION   # noqa: E402

B = 32
cfg = parse_version(FLAGSHIP_VERSION)
img, flow, seg = synth.make_inputs(8, 128, 416)
img, flow, seg = np.tile(img, (4, 1, 1, 1)), np.tile(flow, (4, 1, 1, 1, 1)), np.tile(seg, (4, 1, 1, 1, 1))
from davo_amd import pinned_empty                                     # noqa: E402

e = Engine(cfg, 128, 416, B)
e.load_weights(synth.make_weights(cfg))
pinned = tuple(pinned_empty(a.shape, a.dtype) for a in (img, flow, seg))
for d, a in zip(pinned, (img, flow, seg)):
    d[...] = a
mb = (img.nbytes + flow.nbytes // 2 + seg.nbytes) / 1e6              # flow planes 2,3 are never read and never copied
n = 30
for label, bufs in (("pageable", (img, flow, seg)), ("pinned", pinned)):
    for chunk in (0, 8):
        e.set_option("host_chunk", chunk)
        for _ in range(3):
            ref = e.forward(*bufs)
        t0 = time.perf_counter()
        for _ in range(n):
            e.forward(*bufs)
        dt = time.perf_counter() - t0
        print("davo_forward host buffers %-8s host_chunk=%d: %8.1f triplets/s, %.3f ms per batch of %d, %.1f MB H2D per batch -> %.1f GB/s"
              % (label, chunk, B * n / dt, dt / n * 1e3, B, mb, mb * n / dt / 1e3), flush=True)

# round 5: the streaming entry point on the same batch (davo_submit, four slots; pinned arrays held for the whole run: hold = 8)
for prec in ("f16x3", "f32"):
    e.set_precision(prec)
    outs = [np.empty((B, 2, 6), np.float32) for _ in range(n)]
    for slots in (1, 2, 3, 4):
        e.set_inflight(slots)
        for rep in range(2):
            t0 = time.perf_counter()
            for o in outs:
                e.submit(*pinned, o, hold=8)
            e.synchronize()
            dt = time.perf_counter() - t0
        print("davo_submit   host buffers pinned   %-5s %d slot(s)    : %8.1f triplets/s, %.3f ms per batch of %d (PCIe-inclusive)"
              % (prec, slots, B * n / dt, dt / n * 1e3, B), flush=True)
    e.set_inflight(1)
    e.set_option("host_chunk", 8)
    e.forward(*pinned)
    t0 = time.perf_counter()
    for _ in range(n):
        e.forward(*pinned)
    dt = time.perf_counter() - t0
    print("davo_forward  host buffers pinned   %-5s host_chunk=8 : %8.1f triplets/s, %.3f ms per batch of %d (PCIe-inclusive)"
          % (prec, B * n / dt, dt / n * 1e3, B), flush=True)
