#!/usr/bin/env python
"""Why does a pool worker need twice the time per window of a plain process (tools/exp/loader_breakdown.py)?  The same decode loop
in 14 plain processes, with the destination varied: (A) one private 1.8 MB buffer, rewritten (cache-hot); (B) a private 1.26 GB
ring walked slot by slot (cold memory, anonymous pages); (C) a 1.26 GB shared-memory ring created by the parent, every process
writing the slots of "its" windows round-robin (cold memory + a process's first touch of a page that exists); (D) = C after
madvise(MADV_POPULATE_WRITE) in every process; (E) = C where process k only ever writes slots k, k+P, ... (static assignment)."""
import multiprocessing as mp
import os
import sys
import tempfile
import time
from multiprocessing import shared_memory

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import loader as L                                     # noqa: E402

H, W, B, NRING = 128, 416, 64, 7
SLOTS = B * NRING


def views(bufs):
    return (np.ndarray((SLOTS, H, 3 * W, 3), np.uint8, buffer=bufs[0]), np.ndarray((SLOTS, 4, H, W, 2), np.float32, buffer=bufs[1]),
            np.ndarray((SLOTS, 3, H, W, 1), np.float32, buffer=bufs[2]))


def worker(mode, d, k, P, nwin, names, q):
    if mode == "A":
        img = np.empty((1, H, 3 * W, 3), np.uint8); flow = np.empty((1, 4, H, W, 2), np.float32); seg = np.empty((1, 3, H, W, 1), np.float32)
        slot = lambda i, w: 0                                  # noqa: E731
    elif mode == "B":
        img = np.empty((SLOTS, H, 3 * W, 3), np.uint8); flow = np.empty((SLOTS, 4, H, W, 2), np.float32); seg = np.empty((SLOTS, 3, H, W, 1), np.float32)
        slot = lambda i, w: w % SLOTS                          # noqa: E731
    else:
        segs = [shared_memory.SharedMemory(name=n) for n in names]
        if mode == "D":
            for sm in segs:
                sm._mmap.madvise(23)
        img, flow, seg = views([sm.buf for sm in segs])
        slot = (lambda i, w: w % SLOTS) if mode != "E" else (lambda i, w: (k + (i % (SLOTS // P)) * P) % SLOTS)       # noqa: E731
    L.load_window_into(d, 0, 1, H, W, img[0], flow[0], seg[0], None, L.FLOW_PLANES_USED, L.SEG_PLANES_SOURCES)
    q.put("up")
    t0 = time.perf_counter()
    n = 0
    for i, w in enumerate(range(k, nwin, P)):
        sl = slot(i, w)
        L.load_window_into(d, 0, w + 1, H, W, img[sl], flow[sl], seg[sl], None, L.FLOW_PLANES_USED, L.SEG_PLANES_SOURCES)
        n += 1
    q.put((n, time.perf_counter() - t0))


def main():
    N, real, P = 4541, 642, int(os.environ.get("P", "14"))
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        L.write_synthetic_dump(d, 0, real, H, W, images="scene")
        for w in range(real - 2, N - 2):
            for src, dst in zip(L.window_paths(d, 0, (w % (real - 2)) + 1), L.window_paths(d, 0, w + 1)):
                os.symlink(src, dst)
        sizes = (SLOTS * H * 3 * W * 3, SLOTS * 4 * H * W * 2 * 4, SLOTS * 3 * H * W * 4)
        segs = [shared_memory.SharedMemory(create=True, size=n) for n in sizes]
        for sm in segs:
            np.frombuffer(sm.buf, np.uint8)[::4096] = 0          # the parent allocates every page (as its hipHostRegister would)
        ctx = mp.get_context("fork")
        try:
            for mode in "ABCDEA":
                q = ctx.Queue()
                ps = [ctx.Process(target=worker, args=(mode, d, k, P, N - 2, [sm.name for sm in segs], q)) for k in range(P)]
                t0 = time.perf_counter()
                for p in ps:
                    p.start()
                res = [q.get() for _ in range(2 * P)]
                dt = time.perf_counter() - t0
                for p in ps:
                    p.join()
                res = [r for r in res if r != "up"]
                print("mode %s, %2d processes: %7.0f windows/s, %.3f ms per window per process (wall incl. start %.3f s)" % (
                    mode, P, sum(r[0] for r in res) / max(r[1] for r in res), 1e3 * sum(r[1] / r[0] for r in res) / P, dt), flush=True)
        finally:
            for sm in segs:
                sm.close(); sm.unlink()


if __name__ == "__main__":
    main()
