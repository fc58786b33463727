#!/usr/bin/env python
"""Row f2 rates: files on disk -> decoded batches (threaded loader alone) and files ->
trajectory (CLI, loader + H2D + kernels + stitch), on a synthetic dump in the reference's on-disk format."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import loader as L, synth, parse_version, FLAGSHIP_VERSION   # noqa: E402

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 322                   # frames -> N-2 windows
    H, W, B = 128, 416, 32
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        t0 = time.perf_counter()
        real = min(N, 642)                                                # windows past these are links to them: same files, same decode work
        L.write_synthetic_dump(d, 9, real, H, W)
        for w in range(real - 2, N - 2):
            for src, dst in zip(L.window_paths(d, 9, (w % (real - 2)) + 1), L.window_paths(d, 9, w + 1)):
                os.symlink(src, dst)
        print("wrote %d windows (+ %d links to them) in %.1f s" % (real - 2, N - real, time.perf_counter() - t0), flush=True)
        for workers, procs in ((1, 0), (4, 0), (8, 8)):
            t0 = t1 = time.perf_counter()
            n = n1 = 0
            for s, e, _ in L.kitti_loader(d, 9, H, W, 0, N - 2, B, workers=workers, prefetch=2, decode_procs=procs):
                if n == 0:
                    t1, n1 = time.perf_counter(), e - s                # steady state starts after the first batch
                n += e - s
            t2 = time.perf_counter()
            print("loader alone, %2d threads, %2d decode processes: %7.1f windows/s steady, %7.1f incl. start (%.2f s to first batch)"
                  % (workers, procs, (n - n1) / (t2 - t1), n / (t2 - t0), t1 - t0), flush=True)
        for procs in (8, 12, 14):
            t0 = time.perf_counter()
            stamps = []
            for s, e, _ in L.ProcessWindowLoader(d, 9, H, W, 0, N - 2, B, procs=procs, prefetch=2):
                stamps.append((time.perf_counter(), e))
            half = len(stamps) // 2                                          # the workers are all up by then (spawn + imports: ~0.5 s)
            steady = (stamps[-1][1] - stamps[half][1]) / (stamps[-1][0] - stamps[half][0])
            print("process loader alone, %2d worker processes (shared batch buffers, used planes only): %7.1f windows/s over the second half, "
                  "%7.1f incl. start (%.2f s to first batch)" % (procs, steady, stamps[-1][1] / (stamps[-1][0] - t0), stamps[0][0] - t0), flush=True)
        if "--cli" in sys.argv:
            from davo_amd import run_kitti_pose
            np.savez(os.path.join(d, "w.npz"), **synth.make_weights(parse_version(FLAGSHIP_VERSION)))
            for rep in range(2):                                         # second run: page cache warm, context creation still included
                t0 = time.perf_counter()
                run_kitti_pose.main(["--concat_img_dir", d, "--ckpt_file", os.path.join(d, "w.npz"), "--output_dir", d,
                                     "--test_seq", "9", "--batch_size", str(B), "--force_comm", "--report", os.path.join(d, "report.json")])
                print("CLI files -> trajectory: %.2f s for %d windows; report %s" % (time.perf_counter() - t0, N - 2,
                                                                                     open(os.path.join(d, "report.json")).read()), flush=True)


if __name__ == "__main__":          # the decode processes are spawned and re-import this module
    main()
