#!/usr/bin/env python
"""BASELINE configs[0] at the reference's own operating point - KITTI seq 03's shape (801 frames -> 799 windows), batch 1,
128x416 (run_inference.sh:44-51, test_kitti_pose.py:133-145) - from memory: windows/s of the sequence loop with the time split,
synchronous driver (one davo_forward per window) against the streaming entry point (davo_submit, two windows in flight).

    python tools/config1_stream.py [out.json]

The 799 windows are 64 distinct synthetic ones in page-locked memory (generating 799 takes minutes of host time); every window
is its own submit, trajectory stitched at the end as the CLI does."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION, pinned_empty
    from davo_amd import sequence as S
    H, W, NF, n = 128, 416, 801, 64
    cfg = parse_version(FLAGSHIP_VERSION)
    src = synth.make_inputs(n, H, W)
    pinned = [pinned_empty(a.shape, a.dtype) for a in src]
    for p, a in zip(pinned, src):
        p[...] = a

    def load(s, e):
        k = (s * 37) % n
        return tuple(p[k:k + 1] for p in pinned)         # views of page-locked memory: no host copy per window
    eng = Engine(cfg, H, W, 1)
    eng.load_weights(synth.make_weights(cfg))
    opts = [a for a in sys.argv[1:] if "=" in a]                      # e.g. fold_tails=1 fuse_pack=1 (davo_set_option)
    sys.argv = [a for a in sys.argv if "=" not in a]
    for kv in opts:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    eng.calibrate(*load(0, 1))
    rec = {"what": "BASELINE configs[0] shape from memory: 799 windows, batch 1, 128x416, f16x3 (default arithmetic) and f32", "options": opts, "runs": []}
    ref = None
    for precision in ("f16x3", "f32"):
        eng.set_precision(precision)
        for name, mk in (("synchronous", lambda: None), ("streamed_hold0", lambda: S.PoseStream(eng, 2, hold=0)),
                         ("streamed_hold8", lambda: S.PoseStream(eng, 2, hold=8)),
                         ("streamed_hold0_inflight3", lambda: S.PoseStream(eng, 3, hold=0)), ("streamed_hold8_inflight3", lambda: S.PoseStream(eng, 3, hold=8)),
                         ("streamed_hold8_inflight4", lambda: S.PoseStream(eng, 4, hold=8))):
            best = None
            for rep in range(3):
                eng.set_inflight(1)
                timing = {}
                t0 = time.perf_counter()
                traj, poses = S.run_sequence(eng.forward, load, NF, 1, timing=timing, stream=mk())
                dt = time.perf_counter() - t0
                if best is None or dt < best["total_s"]:
                    best = dict(driver=name, precision=precision, total_s=round(dt, 4), windows_per_s=round((NF - 2) / dt, 1),
                                **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in timing.items()})
            if precision == "f16x3":
                if ref is None:
                    ref = poses.copy()
                best["bit_identical_to_synchronous"] = bool(np.array_equal(poses, ref))
            rec["runs"].append(best)
            print(json.dumps(best), flush=True)
    # the device-resident loop of the same 799 batches (no copies, poses stay in HBM): what the GPU side alone sustains on this box
    eng.set_precision("f16x3")
    bufs = [eng.alloc(a[:1].nbytes).upload(a[:1]) for a in src] + [eng.alloc(48)]
    for infl in (1, 2):
        eng.set_inflight(infl)
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(NF - 2):
                eng.forward_device(1, *bufs)
            t_issue = time.perf_counter() - t0
            eng.synchronize()
            dt = time.perf_counter() - t0
        rec["runs"].append(dict(driver="device_resident_forward_device", inflight=infl, windows_per_s=round((NF - 2) / dt, 1),
                                host_issue_us_per_window=round(1e6 * t_issue / (NF - 2), 1)))
        print(json.dumps(rec["runs"][-1]), flush=True)
    eng.set_inflight(1)
    rec["note"] = ("best of 3 per driver; total_s = loop + stitch (no file write); load_wait_s = building the window views; streamed: "
                   "forward_s = time inside davo_submit (issue + wait for the copy of the window `hold` submits back), drain_s = wait "
                   "for the last windows")
    print(json.dumps(rec, indent=1))
    if len(sys.argv) > 1:
        json.dump(rec, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
