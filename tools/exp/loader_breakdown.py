#!/usr/bin/env python
"""Where the input pipeline's time goes on this host (measurement only): per-window CPU cost of the parts of load_window_into on
photograph-like strips, the ProcessWindowLoader's rate at several worker counts, and the rate of the same number of plain
processes that each decode their share into a private buffer with no task queue at all (the CPU floor of that worker count)."""
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import loader as L                                     # noqa: E402

H, W = 128, 416


def floor_worker(d, lo, hi, step, q):
    img = np.empty((H, 3 * W, 3), np.uint8); flow = np.empty((4, H, W, 2), np.float32); seg = np.empty((3, H, W, 1), np.float32)
    L.load_window_into(d, 0, lo + 1, H, W, img, flow, seg, None, L.FLOW_PLANES_USED, L.SEG_PLANES_SOURCES)
    q.put("up")
    t0 = time.perf_counter()
    n = 0
    for w in range(lo, hi, step):
        L.load_window_into(d, 0, w + 1, H, W, img, flow, seg, None, L.FLOW_PLANES_USED, L.SEG_PLANES_SOURCES)
        n += 1
    q.put((n, time.perf_counter() - t0))


def main():
    from PIL import Image
    N, real, B = 4541, 642, 64
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        L.write_synthetic_dump(d, 0, real, H, W, images="scene")
        for w in range(real - 2, N - 2):
            for src, dst in zip(L.window_paths(d, 0, (w % (real - 2)) + 1), L.window_paths(d, 0, w + 1)):
                os.symlink(src, dst)
        img = np.empty((H, 3 * W, 3), np.uint8); flow = np.empty((4, H, W, 2), np.float32); seg = np.empty((3, H, W, 1), np.float32)

        def t(fn, n=200):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for w in range(n):
                    fn(w)
                best = min(best, (time.perf_counter() - t0) / n)
            return best * 1e3

        def dec_load(w):
            with Image.open(L.window_paths(d, 0, w + 1)[0]) as im:
                im.load()
        print("one process, per window: load_window_into %.3f ms | jpeg open+load %.3f | flow planes 0,1 %.3f | seg planes 0,2 %.3f" % (
            t(lambda w: L.load_window_into(d, 0, w + 1, H, W, img, flow, seg, None, L.FLOW_PLANES_USED, L.SEG_PLANES_SOURCES)), t(dec_load),
            t(lambda w: L._read_npy_into(L.window_paths(d, 0, w + 1)[1], flow, L.FLOW_PLANES_USED)),
            t(lambda w: L._read_npy_into(L.window_paths(d, 0, w + 1)[2], seg, L.SEG_PLANES_SOURCES))), flush=True)
        ctx = mp.get_context("fork")
        for P in [int(x) for x in os.environ.get("FLOOR_P", "8,14,16").split(",")]:
            q = ctx.Queue()
            ps = [ctx.Process(target=floor_worker, args=(d, k, N - 2, P, q)) for k in range(P)]
            t0 = time.perf_counter()
            for p in ps:
                p.start()
            res = [q.get() for _ in range(2 * P)]
            dt = time.perf_counter() - t0
            for p in ps:
                p.join()
            tot = sum(r[0] for r in res if r != "up")
            per = [r[1] / r[0] for r in res if r != "up"]
            print("floor, %2d plain processes: %7.0f windows/s (wall incl. start %.3f s); per window per process %.3f ms" % (
                P, tot / max(r[1] for r in res if r != "up"), dt, 1e3 * sum(per) / len(per)), flush=True)
        for P in [int(x) for x in os.environ.get("LOADER_P", "14,16").split(",")]:
            for chunk in (4, 8):
                t0 = time.perf_counter()
                stamps = []
                ld = L.ProcessWindowLoader(d, 0, H, W, 0, N - 2, B, procs=P, prefetch=2, chunk=chunk)
                for s, e, _ in ld:
                    stamps.append((time.perf_counter(), e))
                half = len(stamps) // 2
                steady = (stamps[-1][1] - stamps[half][1]) / (stamps[-1][0] - stamps[half][0])
                print("ProcessWindowLoader, %2d workers, chunk %2d: %7.0f windows/s over the second half, %7.0f incl. start; workers busy %.3f s "
                      "= %.3f ms per window, utilisation %.2f of %d x %.3f s" % (P, chunk, steady, stamps[-1][1] / (stamps[-1][0] - t0), ld.busy_s,
                                                                               1e3 * ld.busy_s / (N - 2), ld.busy_s / P / (stamps[-1][0] - t0), P, stamps[-1][0] - t0), flush=True)


if __name__ == "__main__":
    main()
