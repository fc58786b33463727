#!/usr/bin/env python
"""Which part of a window's host work stops scaling with worker processes on the GPU box (16-CPU cgroup share)?
N processes each loop over: JPEG decode only / .npy plane reads only / both into a private buffer."""
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def work(mode, d, secs, q):
    import numpy as np
    from PIL import Image
    from davo_amd import loader as L
    H, W = 128, 416
    img = np.empty((H, 3 * W, 3), np.uint8); flow = np.empty((4, H, W, 2), np.float32); seg = np.empty((3, H, W, 1), np.float32)
    n, w, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        jpg, flo, sg = L.window_paths(d, 9, (w % 62) + 1)
        if mode in ("jpeg", "both"):
            with Image.open(jpg) as im:
                img[...] = np.asarray(im.convert("RGB"), np.uint8)
        if mode in ("npy", "both"):
            L._read_npy_into(flo, flow, (0, 1))
            L._read_npy_into(sg, seg, (0, 2))
        n += 1; w += 1
    q.put(n / (time.perf_counter() - t0))


if __name__ == "__main__":
    from davo_amd import loader as L
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        L.write_synthetic_dump(d, 9, 64, 128, 416)
        ctx = mp.get_context("spawn")
        for mode in ("jpeg", "npy", "both"):
            for n in (1, 4, 8, 12, 16):
                q = ctx.Queue()
                ps = [ctx.Process(target=work, args=(mode, d, 3.0, q)) for _ in range(n)]
                for p in ps: p.start()
                rates = [q.get() for _ in ps]
                for p in ps: p.join()
                print("%-5s %2d procs: %8.0f windows/s total, %6.0f per process" % (mode, n, sum(rates), sum(rates) / n), flush=True)
