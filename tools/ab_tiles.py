#!/usr/bin/env python
"""A/B of launch plans in ONE process on one GPU (cdna_hip_programming.md §5.4 rule 24): per-kernel HIP-event times
of the flagship path with the f16x3 layers forced onto one tile shape (davo_set_option "force_tile") or planned.

    python tools/ab_tiles.py [--batch 32] [--tiles -1,5,6] [--rounds 3] [--steps 10] [--options k=v,...]
    python tools/ab_tiles.py --arms "patch_cnv2=1;patch_cnv2=0"      # arms = option sets instead of tile shapes
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--tiles", default="-1,5,6")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--options", default="", help="extra davo_set_option pairs applied to every arm, k=v,k=v")
    ap.add_argument("--arms", default="", help="arms as option sets 'k=v,k=v;k=v,...' (interleaved in one process) instead of --tiles")
    ap.add_argument("--inflight", type=int, default=1, help="batches in flight (consecutive calls rotate through that many streams)")
    a = ap.parse_args()
    cfg = parse_version(FLAGSHIP_VERSION)
    B, H, W = a.batch, a.height, a.width
    e = Engine(cfg, H, W, B)
    e.load_weights(synth.make_weights(cfg))
    nu = min(8, B)
    img, flow, seg = synth.make_inputs(nu, H, W)
    reps = -(-B // nu)
    img, flow, seg = np.tile(img, (reps, 1, 1, 1))[:B], np.tile(flow, (reps, 1, 1, 1, 1))[:B], np.tile(seg, (reps, 1, 1, 1, 1))[:B]
    sets = [(e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
            for _ in range(a.inflight)]
    e.set_inflight(a.inflight)
    for kv in filter(None, a.options.split(",")):
        k, v = kv.split("=")
        e.set_option(k, int(v))
    tiles = [int(t) for t in a.tiles.split(",")]
    arm_opts = {}
    if a.arms:
        tiles = []
        for spec in a.arms.split(";"):
            tiles.append(spec)
            arm_opts[spec] = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in filter(None, spec.split(","))]
    acc = {t: {} for t in tiles}
    wall = {t: [] for t in tiles}
    import time
    for rnd in range(a.rounds):
        for t in tiles:
            if a.arms:
                for k, v in arm_opts[t]:
                    e.set_option(k, v)
            else:
                e.set_option("force_tile", t)
            for i in range(3 * a.inflight):
                e.forward_device(B, *sets[i % a.inflight])
            e.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                e.forward_device(B, *sets[i % a.inflight])
            e.synchronize()
            wall[t].append((time.perf_counter() - t0) / a.steps * 1e3)
            e.profile(1)
            e.profile_reset()
            for i in range(a.steps):
                e.forward_device(B, *sets[i % a.inflight])
            for k, (n, ms) in e.profile_entries().items():
                acc[t].setdefault(k, []).append(ms / max(n, 1))
            e.profile(0)
    names = []
    for t in tiles:
        for k in acc[t]:
            if k not in names:
                names.append(k)
    print("B=%d %dx%d; ms per launch, median over %d rounds; columns = %s %s" % (B, H, W, a.rounds, "options" if a.arms else "force_tile", tiles))
    for k in names:
        print("%-20s" % k + "".join("%10.4f" % (np.median(acc[t][k]) if k in acc[t] else float("nan")) for t in tiles))
    names_t = {0: "128x32", 1: "256x64", 2: "256x128", 3: "128x256", 4: "128x128", 5: "256x256", 6: "208x256", 7: "256x256+128x128 merged"}
    print("last arm's launch plans (mtiles of 128 rows x tile): " + "; ".join(
        "cnv%d %s" % (li + 1, [(m, names_t.get(t, t)) for m, t in e.last_plan(li)]) for li in range(3, 7)))
    print("%-20s" % "step (wall, no events)" + "".join("%10.4f" % np.median(wall[t]) for t in tiles))
    print("%-20s" % "triplets/s" + "".join("%10.0f" % (B / np.median(wall[t]) * 1e3) for t in tiles))


if __name__ == "__main__":
    main()
