#!/usr/bin/env python
"""Measurement only: what a ProcessWindowLoader's start costs in front of its first batch - the fork server warm-up, creating the
ring, starting the workers, and the wait for batch 0 - on a synthetic dump (batch 64, 128x416, 568 windows: one rank's shard)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
T0 = time.perf_counter()
import numpy as np                                                   # noqa: E402,F401
from davo_amd import loader as L                                     # noqa: E402


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    H, W, B, n = 128, 416, 64, 568
    with tempfile.TemporaryDirectory() as d:
        L.write_synthetic_dump(d, 0, n + 2, H, W, images="scene")
        for rep in range(3):
            t = [time.perf_counter()]
            L.warm_workers()
            t.append(time.perf_counter())
            ld = L.ProcessWindowLoader(d, 0, H, W, 0, n, B, procs=procs, hold=1)
            t.append(time.perf_counter())
            ld.start()
            t.append(time.perf_counter())
            it = iter(ld)
            first = next(it)
            t.append(time.perf_counter())
            for _ in it:
                pass
            t.append(time.perf_counter())
            ld.close()
            names = ("warm_workers", "construct", "start (ring + forks)", "first batch ready", "rest of the shard")
            print("rep %d: " % rep + ", ".join("%s %.3f" % (nm, t[i + 1] - t[i]) for i, nm in enumerate(names)), flush=True)


if __name__ == "__main__":
    main()
