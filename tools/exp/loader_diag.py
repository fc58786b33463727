#!/usr/bin/env python
"""Inside a pool worker (measurement only): the ProcessWindowLoader's own pool set-up (fork-server context, shared ring) with a task
that reports, per task, the seconds spent in the strip decode, the two .npy reads, and the process's page faults and context
switches - against the same task run by plain forked processes on the same shared ring."""
import multiprocessing as mp
import os
import resource
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor
from multiprocessing import shared_memory

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import loader as L                                     # noqa: E402

H, W, B, NRING = 128, 416, 64, 7
_S = {}


def init(names, d):
    segs = [shared_memory.SharedMemory(name=n) for n in names]
    _S["segs"] = segs
    _S["v"] = (np.ndarray((B * NRING, H, 3 * W, 3), np.uint8, buffer=segs[0].buf), np.ndarray((B * NRING, 4, H, W, 2), np.float32, buffer=segs[1].buf),
               np.ndarray((B * NRING, 3, H, W, 1), np.float32, buffer=segs[2].buf))
    _S["d"] = d


def task(slot0, w0, n):
    from PIL import Image
    img, flow, seg = _S["v"]
    d = _S["d"]
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    c0 = time.process_time()
    t = [0.0, 0.0, 0.0]
    for j in range(n):
        jpg, flo, sg = L.window_paths(d, 0, w0 + j + 1)
        a = time.perf_counter()
        with Image.open(jpg) as im:
            arr = np.asarray(im.convert("RGB"), np.uint8)
        img[slot0 + j][...] = arr
        b = time.perf_counter()
        L._read_npy_into(flo, flow[slot0 + j], L.FLOW_PLANES_USED)
        c = time.perf_counter()
        L._read_npy_into(sg, seg[slot0 + j], L.SEG_PLANES_SOURCES)
        e = time.perf_counter()
        t[0] += b - a; t[1] += c - b; t[2] += e - c
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    return (n, t[0], t[1], t[2], time.process_time() - c0, r1.ru_minflt - r0.ru_minflt, r1.ru_nvcsw - r0.ru_nvcsw, r1.ru_nivcsw - r0.ru_nivcsw, os.getpid())


def report(name, res, wall, P):
    n = sum(r[0] for r in res)
    print("%-34s %6.0f windows/s | per window: jpeg %.3f flow %.3f seg %.3f ms, cpu %.3f ms | faults %.0f, vol.cs %.2f, invol.cs %.2f | %d pids" % (
        name, n / wall, 1e3 * sum(r[1] for r in res) / n, 1e3 * sum(r[2] for r in res) / n, 1e3 * sum(r[3] for r in res) / n,
        1e3 * sum(r[4] for r in res) / n, sum(r[5] for r in res) / n, sum(r[6] for r in res) / n, sum(r[7] for r in res) / n, len({r[8] for r in res})), flush=True)


def plain(k, P, nwin, names, d, q):
    init(names, d)
    out = []
    for c in range(k, nwin // 4, P):
        out.append(task((4 * c) % (B * NRING), 4 * c, 4))
    q.put(out)


def main():
    N, real, P = 4541, 642, int(os.environ.get("P", "14"))
    nwin = (N - 2) // 4 * 4
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        L.write_synthetic_dump(d, 0, real, H, W, images="scene")
        for w in range(real - 2, N - 2):
            for src, dst in zip(L.window_paths(d, 0, (w % (real - 2)) + 1), L.window_paths(d, 0, w + 1)):
                os.symlink(src, dst)
        sizes = (B * NRING * H * 3 * W * 3, B * NRING * 4 * H * W * 2 * 4, B * NRING * 3 * H * W * 4)
        segs = [shared_memory.SharedMemory(create=True, size=n) for n in sizes]
        names = [sm.name for sm in segs]
        for sm in segs:
            np.frombuffer(sm.buf, np.uint8)[::4096] = 0
        try:
            for rep in range(2):
                ctx = mp.get_context("fork")
                q = ctx.Queue()
                ps = [ctx.Process(target=plain, args=(k, P, nwin, names, d, q)) for k in range(P)]
                t0 = time.perf_counter()
                for p in ps:
                    p.start()
                res = sum((q.get() for _ in range(P)), [])
                wall = time.perf_counter() - t0
                for p in ps:
                    p.join()
                report("plain forked processes", res, wall, P)
                for cname, c in (("pool, fork-server workers", L.worker_context()), ("pool, forked workers", mp.get_context("fork"))):
                    pool = ProcessPoolExecutor(P, mp_context=c, initializer=init, initargs=(names, d))
                    list(pool.map(task, [0] * P, [0] * P, [1] * P))              # workers up
                    t0 = time.perf_counter()
                    futs = [pool.submit(task, (4 * c) % (B * NRING), 4 * c, 4) for c in range(nwin // 4)]
                    res = [f.result() for f in futs]
                    wall = time.perf_counter() - t0
                    pool.shutdown()
                    report(cname, res, wall, P)
        finally:
            for sm in segs:
                sm.close(); sm.unlink()


if __name__ == "__main__":
    main()
