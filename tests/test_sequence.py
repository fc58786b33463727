"""Row f1 (stitch + KITTI writer) and row e (window sharding + gather) on CPU.

tests/golden/kitti_gt_poses_03.txt is a DATA fixture: the KITTI ground-truth poses of sequence 03
that the reference ships under kitti_benchmark/data/odometry/poses/03.txt (801 lines x 12 floats)."""
import os

import numpy as np
import pytest

from davo_amd import sequence as S
from davo_amd import synth, parse_version, FLAGSHIP_VERSION

HERE = os.path.dirname(os.path.abspath(__file__))
GT03 = os.path.join(HERE, "golden", "kitti_gt_poses_03.txt")


def test_euler_convention_and_roundtrip():
    # R = Rx.Ry.Rz, vector order [rz,ry,rx,tx,ty,tz] (geo_utils.py:12-63,105-119)
    v = np.array([[0.3, -0.2, 0.1, 1.0, 2.0, 3.0]])
    m = S.pose_vec2mat(v, np.float64)[0]
    cz, sz, cy, sy, cx, sx = np.cos(.3), np.sin(.3), np.cos(-.2), np.sin(-.2), np.cos(.1), np.sin(.1)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    assert np.allclose(m[:3, :3], Rx @ Ry @ Rz) and np.allclose(m[:3, 3], [1, 2, 3]) and np.allclose(m[3], [0, 0, 0, 1])
    assert np.allclose(S.mat2pose_vec(m)[0], v[0])
    # angles are clipped to [-pi, pi] before use (geo_utils.py:30-32)
    a = S.pose_vec2mat(np.array([[4.0, 0, 0, 0, 0, 0]]), np.float64)
    b = S.pose_vec2mat(np.array([[np.pi, 0, 0, 0, 0, 0]]), np.float64)
    assert np.allclose(a, b)


def test_stitch_reproduces_kitti_gt_seq03(tmp_path):
    gt = S.read_kitti_poses(GT03)
    assert gt.shape[0] == 801
    rel = S.relative_pose_vectors(gt)                       # what a perfect net would output
    assert rel.shape == (799, 2, 6)
    traj = S.stitch_trajectory(rel, mat_dtype=np.float64)
    assert len(traj) == 801                                 # one line per frame (test_kitti_pose.py:119-120,143-149)
    ref = np.linalg.inv(gt[0]) @ gt
    assert np.abs(np.array(traj) - ref).max() < 2e-4
    # float32 matrices, as the reference's TF graph builds them: still within KITTI file precision
    traj32 = S.stitch_trajectory(rel.astype(np.float32))
    assert np.abs(np.array(traj32) - ref).max() < 5e-2
    out = tmp_path / "03-pred_kitti_pose.txt"
    S.write_kitti_poses(str(out), traj)
    lines = open(out).read().splitlines()
    assert len(lines) == 801 and all(len(l.split(" ")) == 12 for l in lines)
    assert lines[0] == "1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0 0.0"        # str(float(x)) format
    assert np.abs(S.read_kitti_poses(str(out)) - np.array(traj)).max() < 1e-12


def test_first_window_special_case():
    p = np.zeros((2, 2, 6), np.float32)
    p[0, 0, 5] = 1.0        # tgt->src0 of window 0: +1 in z
    p[0, 1, 5] = -2.0       # tgt->src1: inv -> +2
    p[1, 0, 5] = 7.0        # ignored: only the first window's src0 pose is used
    p[1, 1, 5] = -3.0
    t = S.stitch_trajectory(p)
    assert [m[2, 3] for m in t] == [0.0, 1.0, 3.0, 6.0]


def test_shard_windows_cover_exactly():
    for nw, world in ((799, 8), (4539, 8), (5, 8), (16, 4), (1, 2)):
        got = []
        for r in range(world):
            lo, hi = S.shard_windows(nw, world, r)
            assert 0 <= lo <= hi <= nw
            got += list(range(lo, hi))
        assert got == list(range(nw))
    assert S.shard_windows(4539, 8, 0) == (0, 568) and S.shard_windows(4539, 8, 7) == (3976, 4539)
    assert [S.is_valid_sample(5, i) for i in range(5)] == [False, True, True, True, False]


def test_run_shard_pads_and_drops():
    calls = []

    def infer(img, flow, seg):
        calls.append(img.shape[0])
        return np.tile(img[:, 0, 0, 0].astype(np.float32)[:, None, None], (1, 2, 6))

    def load(s, e):
        n = e - s
        img = np.zeros((n, 4, 12, 3), np.uint8)
        img[:, 0, 0, 0] = np.arange(s, e)
        return img, np.zeros((n, 4, 4, 4, 2), np.float32), np.zeros((n, 3, 4, 4, 1), np.float32)
    out = S.run_shard(infer, load, 3, 10, 4)
    assert calls == [4, 4]                                   # the ragged last batch is padded to B
    assert np.array_equal(out[:, 0, 0], np.arange(3, 10))    # and the padded outputs are dropped


def _rank_main(rank, world, port, n_frames, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import c_oracle
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    infer = lambda img, flow, seg: c_oracle.forward(cfg, img, flow, seg, weights, nthreads=2)   # noqa: E731
    traj, poses = S.run_sequence(infer, S.synthetic_window_loader(32, 64), n_frames, 3, rank, world)
    if rank == 0:
        q.put((np.array(traj), poses))
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(c_oracle):
    """world_size-2 run of the sharded driver (gloo on CPU; the oracle stands in for the GPU
    engine, which is the only thing that differs on the real path)."""
    import torch.multiprocessing as mp
    n_frames = 13                                            # 11 windows: shards of 6 and 5, ragged batches of 3
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    infer = lambda img, flow, seg: c_oracle.forward(cfg, img, flow, seg, weights, nthreads=2)   # noqa: E731
    traj1, poses1 = S.run_sequence(infer, S.synthetic_window_loader(32, 64), n_frames, 3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    traj2, poses2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert poses2.shape == (11, 2, 6)
    assert np.array_equal(poses1, poses2)                    # sharding changes nothing, bit for bit
    assert np.array_equal(np.array(traj1), traj2)
