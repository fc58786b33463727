"""Row f4: the numpy restatement of the KITTI devkit evaluator vs the reference's own C++ program
(compiled from /root/reference into oracle/_ref/ by oracle/Makefile) on trajectories the reference
ships (ORB-SLAM2 stereo results + ground truth).  Skipped where the reference is absent."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from davo_amd import kitti_eval as K
from davo_amd import sequence as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/kitti_benchmark"
BIN = os.path.join(ROOT, "oracle", "_ref", "test_odometry_all")
GT03 = os.path.join(ROOT, "tests", "golden", "kitti_gt_poses_03.txt")


def test_perfect_trajectory_has_zero_error():
    gt = K.load_poses(GT03)
    err = K.calc_sequence_errors(gt, gt.copy())
    assert err.shape[1] == 5 and err.shape[0] > 50
    assert np.all(err[:, 1] < 1e-3) and np.all(err[:, 2] < 1e-6)      # acos(1-eps) noise only
    assert set(err[:, 3]) <= set(float(x) for x in K.LENGTHS)
    # a stitched perfect prediction evaluates to ~0 drift: f1 (stitch) and f4 (evaluator) agree
    traj = np.array(S.stitch_trajectory(S.relative_pose_vectors(gt), mat_dtype=np.float64))
    err2 = K.calc_sequence_errors(np.linalg.inv(gt[0]) @ gt, traj)
    assert err2[:, 2].max() < 1e-5


def test_known_drift():
    gt = K.load_poses(GT03)
    res = gt.copy()
    res[:, :3, 3] *= 1.05                                              # 5 % scale error
    t_rel, r_rel = K.summary({"03": K.calc_sequence_errors(gt, res)})["03"]
    assert 4.5 < t_rel < 5.5 and r_rel < 1e-2


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not present")
def test_matches_reference_evaluator(tmp_path):
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/test_odometry_all"])
    work = tmp_path
    os.makedirs(work / "data" / "odometry")
    os.symlink(os.path.join(REF, "data", "odometry", "poses"), work / "data" / "odometry" / "poses")
    os.makedirs(work / "results" / "orb" / "data")
    for s in range(11):
        shutil.copy(os.path.join(REF, "data", "odometry", "poses_from_ORBSLAM2-S", "%02d-ORB-SLAM2-S.txt" % s),
                    work / "results" / "orb" / "data" / ("%02d.txt" % s))
    subprocess.run([BIN, "orb"], cwd=str(work), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    checked = 0
    for s in (3, 4, 7, 10, 0):
        ref_file = work / "results" / "orb" / "errors" / ("%02d.txt" % s)
        if not ref_file.exists():
            continue
        ref = np.loadtxt(str(ref_file)).reshape(-1, 5)
        mine = K.calc_sequence_errors(K.load_poses(os.path.join(REF, "data", "odometry", "poses", "%02d.txt" % s)),
                                      K.load_poses(str(work / "results" / "orb" / "data" / ("%02d.txt" % s))))
        assert mine.shape == ref.shape, (s, mine.shape, ref.shape)
        assert np.array_equal(mine[:, 0], ref[:, 0]) and np.array_equal(mine[:, 3], ref[:, 3])
        assert np.abs(mine[:, 1] - ref[:, 1]).max() < 2e-6          # the file is written with %f (6 decimals)
        assert np.abs(mine[:, 2] - ref[:, 2]).max() < 2e-6
        assert np.abs(mine[:, 4] - ref[:, 4]).max() < 1e-3
        stats = np.loadtxt(str(work / "results" / "orb" / ("%02d-stats.txt" % s)))
        t_mean, r_mean = K.sequence_stats(mine)
        assert abs(stats[0] - t_mean) < 2e-6 and abs(stats[1] - r_mean) < 2e-6
        checked += 1
    assert checked >= 3
