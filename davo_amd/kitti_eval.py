"""KITTI odometry evaluation — row f4: numpy restatement of the devkit evaluator the reference
ships as C++ (kitti_benchmark/cpp/test_odometry_all.cpp: trajectoryDistances :44-56,
lastFrameFromSegmentLength :58-63, rotationError/translationError :65-78, calcSequenceErrors
:81-127, saveStats :385-407) and of its reporting script (kitti_benchmark/show_errors.py:38-39).

It is a CPU file-to-file tool after the path (O(N) on <= 4.5 k poses), kept on the CPU.  The
devkit accumulates distances and errors in float32; this restatement does the same so its rows
agree with the compiled reference evaluator (tests/test_kitti_eval.py builds and runs it).
"""
import os

import numpy as np

LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)      # test_odometry_all.cpp:12
STEP_SIZE = 10                                          # :87, "every second"


def load_poses(path):
    """3x4 row-major text lines -> [N,4,4] float64 (loadPoses, :26-42)."""
    a = np.loadtxt(path, dtype=np.float64).reshape(-1, 3, 4)
    m = np.tile(np.eye(4), (a.shape[0], 1, 1))
    m[:, :3, :] = a
    return m


def trajectory_distances(poses):
    d = (poses[1:, :3, 3] - poses[:-1, :3, 3]).astype(np.float32)
    step = np.sqrt((d * d).sum(axis=1, dtype=np.float32), dtype=np.float32)
    dist = np.zeros(poses.shape[0], np.float32)
    acc = np.float32(0)
    for i, s in enumerate(step):                         # sequential float32 accumulation, as the devkit
        acc = np.float32(acc + s)
        dist[i + 1] = acc
    return dist


def calc_sequence_errors(poses_gt, poses_result):
    """-> float array [n,5]: first_frame, r_err/len (rad/m), t_err/len, len, speed (calcSequenceErrors)."""
    dist = trajectory_distances(poses_gt)
    inv_gt, inv_res = np.linalg.inv(poses_gt), np.linalg.inv(poses_result)
    rows = []
    n = poses_gt.shape[0]
    for first in range(0, n, STEP_SIZE):
        for ln in LENGTHS:
            target = np.float32(dist[first] + np.float32(ln))
            idx = np.nonzero(dist[first:] > target)[0]
            if idx.size == 0:
                continue
            last = first + int(idx[0])
            delta_gt = inv_gt[first] @ poses_gt[last]
            delta_res = inv_res[first] @ poses_result[last]
            err = np.linalg.inv(delta_res) @ delta_gt
            d = np.float32(0.5 * (np.float32(err[0, 0]) + np.float32(err[1, 1]) + np.float32(err[2, 2]) - 1.0))
            r_err = np.float32(np.arccos(np.clip(d, np.float32(-1), np.float32(1))))
            t = err[:3, 3].astype(np.float32)
            t_err = np.float32(np.sqrt(np.float32((t * t).sum(dtype=np.float32))))
            speed = np.float32(ln / (0.1 * float(last - first + 1)))
            rows.append((first, np.float32(r_err / np.float32(ln)), np.float32(t_err / np.float32(ln)), float(ln), speed))
    return np.array(rows, np.float64).reshape(-1, 5)


def sequence_stats(err):
    """mean t_err, mean r_err of a sequence's rows (saveStats :385-407)."""
    return float(err[:, 2].mean()), float(err[:, 1].mean())


def summary(errors_by_seq):
    """show_errors.py:38-39: per sequence t_rel (%) = mean(t_err)*100, r_rel = mean(r_err)*57.3
    (deg/m), plus the average over sequences -> {seq: (t_rel, r_rel)}."""
    out = {}
    for seq, err in errors_by_seq.items():
        out[seq] = (float(err[:, 2].mean() * 100.0), float(err[:, 1].mean() * 57.3))
    if out:
        out["ave"] = (float(np.mean([v[0] for v in out.values()])), float(np.mean([v[1] for v in out.values()])))
    return out


def evaluate(gt_dir, result_dir, seqs):
    """Evaluate `<result_dir>/NN.txt` (or NN-pred_kitti_pose.txt) against `<gt_dir>/NN.txt`."""
    errs = {}
    for s in seqs:
        gt = load_poses(os.path.join(gt_dir, "%02d.txt" % s))
        cand = [os.path.join(result_dir, "%02d.txt" % s), os.path.join(result_dir, "%02d-pred_kitti_pose.txt" % s)]
        path = next((c for c in cand if os.path.exists(c)), None)
        if path is None:
            raise FileNotFoundError("no result trajectory for sequence %02d under %s" % (s, result_dir))
        res = load_poses(path)
        if res.shape[0] != gt.shape[0]:
            raise ValueError("sequence %02d: %d poses, ground truth has %d" % (s, res.shape[0], gt.shape[0]))
        errs["%02d" % s] = calc_sequence_errors(gt, res)
    return errs
