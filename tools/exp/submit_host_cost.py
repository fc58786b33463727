#!/usr/bin/env python
"""Measurement only: the host time of one davo_submit at batch 1 when a slot is free (bursts of `slots` submits after a drain),
against the per-window period of a long streamed run - is the streamed batch-1 rate bound by the host's issue cost or by the GPU?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import synth                                           # noqa: E402
from davo_amd.davo import Engine, pinned_empty                       # noqa: E402
from davo_amd.version import parse_version, FLAGSHIP_VERSION         # noqa: E402


def main():
    H, W, B = 128, 416, 1
    for prec in ("f16x3", "f32"):
        e = Engine(parse_version(FLAGSHIP_VERSION), H, W, B, device=0)
        e.load_weights(synth.make_weights(FLAGSHIP_VERSION))
        e.set_precision(prec)
        img, flow, seg = synth.make_inputs(64, H, W, seed=3)
        pin = [pinned_empty(a.shape, a.dtype, 0) for a in (img, flow, seg)]
        for p, a in zip(pin, (img, flow, seg)):
            p[...] = a
        img, flow, seg = pin
        out = np.zeros((4096, 1, 2, 6), np.float32)
        for slots in (1, 2, 3, 4):
            e.set_inflight(slots)
            for i in range(32):                                      # warm: code objects, staging sets
                e.submit(img[i % 64:i % 64 + 1], flow[i % 64:i % 64 + 1], seg[i % 64:i % 64 + 1], out[i], 8)
            e.synchronize()
            burst = []
            for r in range(200):
                for k in range(slots):
                    i = (r * slots + k) % 64
                    t0 = time.perf_counter()
                    e.submit(img[i:i + 1], flow[i:i + 1], seg[i:i + 1], out[r * slots + k], 8)
                    burst.append(time.perf_counter() - t0)
                e.synchronize()
            n = 3200
            t0 = time.perf_counter()
            for j in range(n):
                i = j % 64
                e.submit(img[i:i + 1], flow[i:i + 1], seg[i:i + 1], out[j], 8)
            e.synchronize()
            per = (time.perf_counter() - t0) / n
            print("%s %d slot(s): un-blocked submit %.1f us (median %.1f), streamed period %.1f us per window = %.0f windows/s" % (
                prec, slots, 1e6 * np.mean(burst), 1e6 * np.median(burst), 1e6 * per, 1 / per), flush=True)
        del e


if __name__ == "__main__":
    main()
