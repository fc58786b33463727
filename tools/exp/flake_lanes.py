#!/usr/bin/env python
"""Which lane of the fused pose epilogue goes wrong, and what does it hold?  (GPU box; needs a -DDAVO_POSE_DEBUG build)

    python tools/build_variant.py _fd -fslp-vectorize -DDAVO_POSE_NO_DRAIN -DDAVO_POSE_DEBUG
    DAVO_LIB_SUFFIX=_fd python tools/exp/flake_lanes.py --batch 4 --tile 0 --reps 300

The debug build stores, per lane of every cnv7 workgroup (128x32 tiles: two column groups), the operands of the six products
(s0, s1, w0, w1, w2 per column group) and the six sums before the wave reduction.  Every record is checked against its own
operands (float64), so a wrong lane is found absolutely, not only against the first forward."""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--show", type=int, default=12)
a = ap.parse_args()
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION, _lib  # noqa: E402

cfg = parse_version(FLAGSHIP_VERSION)
B, H, W = a.batch, 128, 416
img, flow, seg = synth.make_inputs(B, H, W)
e = Engine(cfg, H, W, B)
e.load_weights(synth.make_weights(cfg))
e.set_option("force_tile", a.tile)
bufs = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
mt, ntn, T = (2 * B * 16 * 52 + 127) // 128, 8, 256
need = 2 * mt * ntn * 6
nwg = 2 * mt * ntn
names = ["s0a", "s1a", "w0a", "w1a", "w2a", "s0b", "s1b", "w0b", "w1b", "w2b", "q0", "q1", "q2", "q3", "q4", "q5", "q4a", "q5a", "-", "-"]
NR = len(names)
bad_forwards, shown, hist = 0, 0, {}
n_first = n_second = lost_a = lost_b = 0
quarter, waves = [0, 0, 0, 0], [0, 0, 0, 0]
ref_pose = None
for r in range(a.reps):
    e.forward_device(B, *bufs)
    e.synchronize()
    pose = bufs[3].download((B, 2, 6))
    raw = e.debug_read("pose_tiles", (4 * need + nwg * T * NR,))
    rec = raw[4 * need:].reshape(nwg, T, NR).astype(np.float64)
    if ref_pose is None:
        ref_pose = pose
    sa, sb = rec[..., 0:2], rec[..., 5:7]
    wa, wb = rec[..., 2:5], rec[..., 7:10]
    want = np.concatenate([sa[..., 0:1] * wa + sb[..., 0:1] * wb, sa[..., 1:2] * wa + sb[..., 1:2] * wb], axis=-1)
    mag = np.concatenate([np.abs(sa[..., 0:1] * wa) + np.abs(sb[..., 0:1] * wb), np.abs(sa[..., 1:2] * wa) + np.abs(sb[..., 1:2] * wb)], axis=-1)
    got = rec[..., 10:16]
    wrong = np.abs(got - want) > 1e-5 * mag + 1e-30
    # which of the two instructions lost it: was group a's own result (q4 after the first column group) already wrong?
    first_wrong = np.abs(rec[..., 16] - sa[..., 1] * wa[..., 1]) > 1e-5 * np.abs(sa[..., 1] * wa[..., 1]) + 1e-30
    n_first += int((wrong[..., 4] & first_wrong).sum()); n_second += int((wrong[..., 4] & ~first_wrong).sum())
    lost_a += int((wrong[..., 4] & (np.abs(got[..., 4] - sb[..., 1] * wb[..., 1]) <= 1e-5 * np.abs(sb[..., 1] * wb[..., 1]) + 1e-30)).sum())
    lost_b += int((wrong[..., 4] & (np.abs(got[..., 4] - sa[..., 1] * wa[..., 1]) <= 1e-5 * np.abs(sa[..., 1] * wa[..., 1]) + 1e-30)).sum())
    for wg, t in zip(*np.nonzero(wrong[..., 4])):
        quarter[t % 64 // 16] += 1
        waves[t // 64] += 1
    nw = int(wrong.sum())
    differs = not np.array_equal(pose, ref_pose)
    if nw or differs:
        bad_forwards += 1
    for wg, t, k in zip(*np.nonzero(wrong)):
        hist[int(k)] = hist.get(int(k), 0) + 1
        if shown >= a.show:
            continue
        shown += 1
        v = rec[wg, t]
        grp, rest = divmod(int(wg), mt * ntn)
        print("rep %d: head %d mtile %d ntile %d wave %d lane %d  q%d = %.9g, operands give %.9g" %
              (r, grp, rest // ntn, rest % ntn, t // 64, t % 64, k, got[wg, t, k], want[wg, t, k]))
        print("   " + "  ".join("%s=%.9g" % (n, x) for n, x in zip(names, v)))
        s = 0 if k < 3 else 1
        ga, gb = v[s] * v[2 + k % 3], v[5 + s] * v[7 + k % 3]
        xa, xb = got[wg, t, k] - gb, got[wg, t, k] - ga        # what column group a / b would have had to contribute
        print("   group a term %.9g, group b term %.9g; if only a is wrong it contributed %.9g (ratio %.6g), if only b: %.9g (ratio %.6g)" %
              (ga, gb, xa, xa / ga if ga else np.nan, xb, xb / gb if gb else np.nan))
        pool = {n: x for n, x in zip(names, v)}
        pool.update({"q%d_want" % j: want[wg, t, j] for j in range(6)})
        pool["one"] = 1.0
        for target, label in ((xa, "a"), (xb, "b"), (got[wg, t, k], "q")):
            hits = []
            for (n1, x1), (n2, x2) in itertools.combinations_with_replacement(pool.items(), 2):
                if target != 0 and abs(x1 * x2 - target) <= 2e-6 * abs(target):
                    hits.append("%s*%s" % (n1, n2))
            for (n1, x1), (n2, x2), (n3, x3) in itertools.combinations_with_replacement(pool.items(), 3):
                if target != 0 and "one" not in (n1, n2, n3) and abs(x1 * x2 * x3 - target) <= 2e-6 * abs(target):
                    hits.append("%s*%s*%s" % (n1, n2, n3))
            if hits:
                print("   %s = %s" % (label, " = ".join(hits[:6])))
        sys.stdout.flush()
    if differs and not nw:
        print("rep %d: poses differ from the first forward's but every lane record is consistent with its operands" % r, flush=True)
print("lib %s B=%d tile %d: %d of %d forwards hold a wrong lane or differ from the first; wrong lanes per output q0..q5: %s; plan cnv7 %s" %
      (os.path.basename(_lib.LIB_PATH), B, a.tile, bad_forwards, a.reps, [hist.get(k, 0) for k in range(6)], e.last_plan(6)), flush=True)
print("wrong q4 lanes: %d with group a's own q4 already wrong after the first instruction, %d with it right (lost by the second); final q4 = group b's product alone in %d, = group a's alone in %d; "
      "by lane quarter 0..3: %s; by wave 0..3: %s" % (n_first, n_second, lost_a, lost_b, quarter, waves), flush=True)
e.close()
