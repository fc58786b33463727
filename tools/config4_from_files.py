#!/usr/bin/env python
"""BASELINE configs[3] on one rank, from files: KITTI seq 00's shape (4541 frames -> 4539 windows, batch 64) as a dump in the
reference's on-disk format -> CLI (worker-process loader, RCCL all-gather forced at world size 1) -> trajectory, with the run's
time split.  The dump holds 640 distinct windows; the rest are links to them (same files, same decode work, warm page cache).

    python tools/config4_from_files.py [out.json] [--shard r/R] [--images noise|scene] [--sync] [--fresh] [--no-comm]
        --shard 3/8: only what rank 3 of 8 would do (568 windows); --sync: the synchronous driver (one davo_forward per batch)
        instead of the streaming entry point; --fresh: every run is a process of its own (`python -m davo_amd.run_kitti_pose`),
        so the report's start-up split (process start -> first batch) is a rank's real one"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from davo_amd import loader as L, synth, parse_version, FLAGSHIP_VERSION, run_kitti_pose
    N, real, H, W, B = 4541, 642, 128, 416, 64
    argv = sys.argv[1:]
    shard = argv.pop(argv.index("--shard") + 1) if "--shard" in argv else None
    if shard:
        argv.remove("--shard")
    procs = argv.pop(argv.index("--procs") + 1) if "--procs" in argv else None
    if procs:
        argv.remove("--procs")
    images = argv.pop(argv.index("--images") + 1) if "--images" in argv else "noise"
    if "--images" in argv:
        argv.remove("--images")
    sync = "--sync" in argv
    fresh = "--fresh" in argv
    no_comm = "--no-comm" in argv          # one GPU, no RCCL communicator (the start-up split without librccl)
    argv = [a for a in argv if a not in ("--sync", "--fresh", "--no-comm")]
    out = argv[0] if argv else None
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        L.write_synthetic_dump(d, 0, real, H, W, images=images)
        jpg_bytes = sum(os.path.getsize(L.window_paths(d, 0, w + 1)[0]) for w in range(real - 2)) // (real - 2)
        for w in range(real - 2, N - 2):
            for src, dst in zip(L.window_paths(d, 0, (w % (real - 2)) + 1), L.window_paths(d, 0, w + 1)):
                os.symlink(src, dst)
        np.savez(os.path.join(d, "w.npz"), **synth.make_weights(parse_version(FLAGSHIP_VERSION)))
        runs = []
        for rep in range(3):                                           # run 0 pays the library's first load and the cold page cache
            t0 = time.perf_counter()
            cli = (["--concat_img_dir", d, "--ckpt_file", os.path.join(d, "w.npz"), "--output_dir", d, "--test_seq", "0",
                    "--batch_size", str(B), "--report", os.path.join(d, "report.json")] + ([] if no_comm else ["--force_comm"]) +
                   (["--emulate_shard", shard] if shard else []) + (["--loader_procs", procs] if procs else []) + (["--sync_driver"] if sync else []))
            if fresh:
                import subprocess
                subprocess.check_call([sys.executable, "-m", "davo_amd.run_kitti_pose"] + cli, cwd=ROOT)
            else:
                run_kitti_pose.main(cli)
            r = json.load(open(os.path.join(d, "report.json")))
            r["process_wall_s_incl_context_and_weights"] = round(time.perf_counter() - t0, 3)
            runs.append(r)
        assert len(open(os.path.join(d, "00-pred_kitti_pose.txt")).read().splitlines()) == N
    rec = {"what": "BASELINE configs[3] shape on ONE rank from files (seq 00: 4541 frames, 4539 windows, batch 64, forced RCCL gather); "
                   "8 ranks would each take 568 of these windows" + (": THIS run is the work of rank %s" % shard if shard else ""),
           "driver": "synchronous (davo_forward per batch)" if sync else "streamed (davo_submit, three batches in flight (four up to batch 2))",
           "fresh_process_per_run": fresh, "rccl_communicator": not no_comm, "images": images, "mean_jpeg_bytes_per_strip": jpg_bytes, "runs": runs}
    print(json.dumps(rec, indent=1))
    if out:
        json.dump(rec, open(out, "w"), indent=1)


if __name__ == "__main__":          # the loader's worker processes are spawned and re-import this module
    main()
