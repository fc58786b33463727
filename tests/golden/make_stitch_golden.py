#!/usr/bin/env python
"""Generate tests/golden/stitch_golden.json from the REFERENCE's own numpy pose utilities.

Runs in the build container only (it imports /root/reference/data/kitti/pose_evaluation_utils.py, pure numpy — the
one piece of the reference that is importable here); the reference module stays where it is, only the
JSON of inputs and expected outputs travels.  It pins row f1 (trajectory stitch) to reference-held code:

* ``pose_vec2mat(vec, False)`` / ``euler2mat(rz, ry, rx)`` of the reference (pose_evaluation_utils.py:218-311,359-370:
  R = Rx.Ry.Rz on ``[rz,ry,rx,tx,ty,tz]``, the same convention as utils/geo_utils.py:12-63,105-119) on seeded
  vectors: KITTI-sized motions, large angles over the whole [-pi, pi) range the reference function accepts (it
  asserts instead of clipping; the clip of the TF twin, geo_utils.py:30-32, is covered by a hand case in
  tests/test_sequence.py), zero angles (the function's skipped-factor branches) and -pi exactly;
* the driver's chain (test_kitti_pose.py:141-149) evaluated with those reference functions in float64 on a seeded
  40-window pose tensor: first window contributes T(tgt->src0), every window inv(T(tgt->src1)).

    python tests/golden/make_stitch_golden.py          # rewrites tests/golden/stitch_golden.json
"""
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data/kitti/pose_evaluation_utils.py"


def main():
    spec = importlib.util.spec_from_file_location("ref_pose_evaluation_utils", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    rng = np.random.RandomState(8964)
    vecs = []
    for _ in range(24):                                   # KITTI-sized frame-to-frame motion
        vecs.append(np.concatenate([rng.uniform(-0.05, 0.05, 3), rng.uniform(-1.5, 1.5, 3)]))
    for _ in range(32):                                   # the whole angle range the reference accepts
        vecs.append(np.concatenate([rng.uniform(-np.pi, np.pi, 3) * 0.999, rng.uniform(-3, 3, 3)]))
    vecs += [np.array([0.0, 0.0, 0.0, 1.0, -2.0, 3.0]), np.array([0.7, 0.0, 0.0, 0.0, 0.0, 0.0]),
             np.array([0.0, -1.1, 0.0, 0.0, 0.0, 0.0]), np.array([0.0, 0.0, 2.9, 0.0, 0.0, 0.0]),
             np.array([-np.pi, 0.3, -np.pi, 0.5, 0.5, 0.5]), np.array([3.0, -3.0, 3.1, -1.0, 0.0, 1.0])]
    vecs = np.array(vecs)
    mats = np.array([ref.pose_vec2mat(v, False) for v in vecs])
    rots = np.array([ref.euler2mat(v[0], v[1], v[2]) for v in vecs])
    assert np.array_equal(mats[:, :3, :3], rots)

    nw = 40
    poses = np.concatenate([rng.uniform(-0.04, 0.04, (nw, 2, 3)), rng.uniform(-1.2, 1.2, (nw, 2, 3))], -1)
    poses = poses.astype(np.float32).astype(np.float64)       # the network emits float32 poses (davo.py:1553-1569)
    steps = []
    for w in range(nw):                                   # test_kitti_pose.py:141-145
        if w == 0:
            steps.append(ref.pose_vec2mat(poses[w, 0], False))
        steps.append(np.linalg.inv(ref.pose_vec2mat(poses[w, 1], False)))
    prev = np.eye(4)
    traj = [prev]
    for p in steps:                                       # :147-149
        prev = np.dot(prev, p)
        traj.append(prev)
    out = {"generator": "tests/golden/make_stitch_golden.py",
           "reference": "data/kitti/pose_evaluation_utils.py:218-311 (euler2mat), :359-370 (pose_vec2mat)",
           "vectors": vecs.tolist(), "matrices": mats.tolist(),
           "chain_poses": poses.tolist(), "chain_trajectory": np.array(traj).tolist()}
    with open(os.path.join(HERE, "stitch_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote %d vectors, a %d-window chain" % (len(vecs), nw))


if __name__ == "__main__":
    main()
