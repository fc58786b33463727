"""Row f2: the reference's on-disk window format and the threaded, prefetching loader."""
import os

import numpy as np
import pytest

from davo_amd import loader as L
from davo_amd import sequence as S
from davo_amd import synth


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("dump"))
    assert L.write_synthetic_dump(d, 3, 9, 32, 64) == 7                     # 9 frames -> 7 windows
    return d


def test_dump_layout_and_frame_count(dump):
    files = sorted(os.listdir(os.path.join(dump, "03")))
    assert files[0] == "000001-flownet2.npy" and "000001.jpg" in files and "000007-seglabel.npy" in files
    assert L.count_frames(dump, 3) == 9                                       # #jpg + 2 (test_kitti_pose.py:81-82)
    img, flow, seg = L.load_window(dump, 3, 1, 32, 64)
    assert img.shape == (32, 192, 3) and img.dtype == np.uint8
    assert flow.shape == (4, 32, 64, 2) and seg.shape == (3, 32, 64, 1)
    _, flow0, seg0 = synth.make_inputs(1, 32, 64, first_window=0)
    assert np.array_equal(flow, flow0[0]) and np.array_equal(seg, seg0[0])   # npy arrays are exact; jpg is lossy
    with pytest.raises(ValueError, match="expected"):
        L.load_window(dump, 3, 1, 32, 65)


def test_threaded_loader_order_and_content(dump):
    ld = L.kitti_loader(dump, 3, 32, 64, 1, 7, batch_size=4, workers=3, prefetch=1)
    got = list(ld)
    assert [(s, e) for s, e, _ in got] == [(1, 5), (5, 7)] and len(ld) == 2
    for s, e, (img, flow, seg) in got:
        assert img.shape[0] == e - s
        for k, w in enumerate(range(s, e)):
            ref = L.load_window(dump, 3, w + 1, 32, 64)
            assert np.array_equal(img[k], ref[0]) and np.array_equal(flow[k], ref[1]) and np.array_equal(seg[k], ref[2])


def test_decode_processes_match_inline_decode(dump):
    got = list(L.kitti_loader(dump, 3, 32, 64, 0, 7, batch_size=3, workers=2, prefetch=1, decode_procs=2))
    assert [(s, e) for s, e, _ in [(g[0], g[1], None) for g in got]] == [(0, 3), (3, 6), (6, 7)]
    s, e, (img, flow, seg) = got[-1]
    ref = L.load_window(dump, 3, 7, 32, 64)
    assert np.array_equal(img[0], ref[0]) and np.array_equal(flow[0], ref[1]) and np.array_equal(seg[0], ref[2])


def test_npy_readinto_and_fallback(tmp_path):
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    np.save(str(tmp_path / "a.npy"), a)
    dst = np.empty((2, 3, 4), np.float32)
    L._read_npy_into(str(tmp_path / "a.npy"), dst)
    assert np.array_equal(dst, a)
    np.save(str(tmp_path / "b.npy"), a.astype(np.float64))               # other dtype: np.load + cast
    dst[...] = 0
    L._read_npy_into(str(tmp_path / "b.npy"), dst)
    assert np.array_equal(dst, a)
    with open(str(tmp_path / "a.npy"), "r+b") as f:
        f.truncate(130)
    with pytest.raises(ValueError, match="truncated"):
        L._read_npy_into(str(tmp_path / "a.npy"), dst)


def test_loader_buffer_ring(dump):
    """alloc=...: batches live in a ring of prefetch+4 caller-provided buffer sets and stay valid until the next one is taken."""
    made = []

    def alloc(shape, dtype):
        made.append(np.empty(shape, dtype))
        return made[-1]
    ld = L.kitti_loader(dump, 3, 32, 64, 0, 7, batch_size=2, workers=2, prefetch=1, alloc=alloc)
    for s, e, (img, flow, seg) in ld:
        held = (img.copy(), flow.copy())
        import time
        time.sleep(0.05)                                        # let the producer run ahead
        assert np.array_equal(img, held[0]) and np.array_equal(flow, held[1])
        ref = L.load_window(dump, 3, s + 1, 32, 64)
        assert np.array_equal(img[0], ref[0]) and np.array_equal(seg[0], ref[2]) and img.shape[0] == e - s
        assert any(np.shares_memory(img, m) for m in made)
    assert len(made) == 4 * 3                                   # 7 windows = 4 batches < prefetch + 4 sets, x 3 tensors


def test_loader_propagates_errors_and_stops_early(dump):
    def bad(w):
        if w == 3:
            raise IOError("boom %d" % w)
        return L.load_window(dump, 3, w + 1, 32, 64)
    it = iter(L.ThreadedWindowLoader(bad, 0, 7, 2, workers=2, prefetch=1))
    assert next(it)[0] == 0
    with pytest.raises(IOError, match="boom 3"):
        for _ in it:
            pass
    it2 = iter(L.kitti_loader(dump, 3, 32, 64, 0, 7, 1, workers=2, prefetch=1))
    next(it2)
    it2.close()                                                              # consumer gives up: producer must exit


def test_run_sequence_with_prefetching_loader_matches_inline(dump, c_oracle):
    from davo_amd import parse_version, FLAGSHIP_VERSION
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    infer = lambda img, flow, seg: c_oracle.forward(cfg, img, flow, seg, weights, nthreads=2)   # noqa: E731
    fac = S.kitti_window_loader(dump, 3, 9, 32, 64, workers=2, prefetch=2)
    traj_a, poses_a = S.run_sequence(infer, fac, 9, 3)                       # threaded loader (factory)
    traj_b, poses_b = S.run_sequence(infer, fac.__call__, 9, 3)              # inline (callable) form
    assert np.array_equal(poses_a, poses_b) and len(traj_a) == 9


def test_process_loader_fills_shared_buffers_with_the_planes_the_path_reads(dump):
    """ProcessWindowLoader: worker processes decode whole windows into shared-memory batch buffers; batches arrive in
    order, the strip and the consumed planes (flow 0,1; label maps of the two source frames) equal the inline loads,
    the planes the path never reads (davo.py:978-982,998-1004) are not touched, and the segments are gone afterwards."""
    d, n_windows, H, W = dump, 7, 32, 64
    ref = {w: L.load_window(d, 3, w + 1, H, W) for w in range(n_windows)}
    for B, procs, chunk in ((4, 2, None), (5, 3, 2)):
        ld = L.ProcessWindowLoader(d, 3, H, W, 0, n_windows, B, procs=procs, prefetch=1, chunk=chunk)
        seen = []
        for s, e, (img, flow, seg) in ld:
            assert img.shape == (e - s, H, 3 * W, 3) and s == (seen[-1] if seen else 0)
            for i, w in enumerate(range(s, e)):
                assert np.array_equal(img[i], ref[w][0])
                assert np.array_equal(flow[i, :2], ref[w][1][:2]) and np.array_equal(seg[i, [0, 2]], ref[w][2][[0, 2]])
                assert not flow[i, 2:].any() and not seg[i, 1].any()
            seen.append(e)
        assert seen[-1] == n_windows and len(seen) == -(-n_windows // B)
        assert ld._pool is None and not any(os.path.exists("/dev/shm/" + sm.name.lstrip("/")) for trio in ld._segs for sm in trio)
        ld.close()
        assert ld._segs == []
    # every plane on request (the -segmask_all-static variant reads the target frame's label map too)
    ld = L.ProcessWindowLoader(d, 3, H, W, 2, 7, 5, procs=2, seg_planes=(0, 1, 2), flow_planes=None)
    (s, e, (img, flow, seg)), = list(ld)
    assert (s, e) == (2, 7) and all(np.array_equal(flow[i], ref[2 + i][1]) and np.array_equal(seg[i], ref[2 + i][2]) for i in range(5))


def test_process_loader_hold_keeps_earlier_batches_valid(dump):
    """hold = k: a batch stays valid until k + 1 further ones have been asked for (the streaming driver submits batch n while its
    copy of batch n - 1 may still be running); the ring is k entries longer for the same number of batches in flight."""
    d, n_windows, H, W = dump, 7, 32, 64
    ref = {w: L.load_window(d, 3, w + 1, H, W) for w in range(n_windows)}
    plain = L.ProcessWindowLoader(d, 3, H, W, 0, n_windows, 1, procs=2, prefetch=1)
    ld = L.ProcessWindowLoader(d, 3, H, W, 0, n_windows, 1, procs=2, prefetch=1, hold=2)
    assert ld.nring == plain.nring + 2
    plain.close()
    import time
    kept = []
    for s, e, (img, flow, seg) in ld:
        kept.append((s, img, flow, seg))
        time.sleep(0.05)                                                     # the workers run ahead as far as the ring allows
        for s0, i0, f0, g0 in kept[-3:]:                                     # this batch and the two before it are intact
            assert np.array_equal(i0[0], ref[s0][0]) and np.array_equal(f0[0, :2], ref[s0][1][:2]) and np.array_equal(g0[0, 0], ref[s0][2][0])
    assert [k[0] for k in kept] == list(range(n_windows))
    ld.close()
    with pytest.raises(L.ShmBudgetError):
        per = H * 3 * W * 3 + 4 * H * W * 2 * 4 + 3 * H * W * 4
        L.ProcessWindowLoader(d, 3, H, W, 0, n_windows, 1, procs=2, hold=2, shm_budget=5 * per)      # 4 + hold entries are the minimum


def test_process_loader_propagates_a_missing_file(dump, tmp_path):
    import shutil
    d, n_windows, H, W = dump, 7, 32, 64
    d2 = str(tmp_path / "broken")
    shutil.copytree(d, d2)
    os.remove(L.window_paths(d2, 3, 4)[1])
    with pytest.raises(Exception):
        for _ in L.ProcessWindowLoader(d2, 3, H, W, 0, n_windows, 4, procs=2):
            pass


def test_run_sequence_with_process_loader_reports_the_time_split(dump, c_oracle):
    """run_sequence over the process loader == over inline loads (poses bit for bit: the unread planes do not enter the
    path), and the time split it reports adds up."""
    from davo_amd import sequence as S, parse_version, FLAGSHIP_VERSION
    d, n_windows, H, W = dump, 7, 32, 64
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    infer = lambda img, flow, seg: c_oracle.forward(cfg, img, flow, seg, weights)      # noqa: E731 — the checker stands in for the GPU
    fac = S.kitti_window_loader(d, 3, n_windows + 2, H, W)
    ref_traj, ref_poses = S.run_sequence(infer, fac, n_windows + 2, 4)
    timing = {}
    fac2 = S.kitti_window_loader(d, 3, n_windows + 2, H, W, procs=2)
    traj, poses = S.run_sequence(infer, fac2, n_windows + 2, 4, timing=timing)
    assert np.array_equal(poses, ref_poses) and np.array_equal(np.array(traj), np.array(ref_traj))
    assert timing["windows_this_rank"] == n_windows and timing["forward_s"] > 0 and timing["load_wait_s"] >= 0
    assert timing["gather_s"] >= 0 and timing["stitch_s"] > 0


def test_process_loader_started_early_and_never_iterated_cleans_up(dump):
    """start() fills batches ahead of the first next(); a loader that is started and then dropped (or closed) stops its
    producer and leaves nothing in /dev/shm."""
    d, n_windows, H, W = dump, 7, 32, 64
    ld = L.ProcessWindowLoader(d, 3, H, W, 0, n_windows, 4, procs=2).start()
    names = [sm.name.lstrip("/") for trio in ld._segs for sm in trio]
    assert names and ld.start() is ld                                     # idempotent
    got = [(s, e) for s, e, _ in ld]
    assert got == [(0, 4), (4, 7)]
    ld.close()
    ld2 = L.ProcessWindowLoader(d, 3, H, W, 0, n_windows, 4, procs=2).start()
    names += [sm.name.lstrip("/") for trio in ld2._segs for sm in trio]
    ld2.close()                                                           # never iterated
    assert not any(os.path.exists("/dev/shm/" + n) for n in names)
    fac = S.kitti_window_loader(d, 3, n_windows + 2, H, W, procs=2)
    fac.prestart(0, n_windows, 4)
    assert [(s, e) for s, e, _ in fac.for_range(0, n_windows, 4)] == [(0, 4), (4, 7)]


def test_process_loader_sizes_its_ring_in_bytes_and_falls_back_when_shm_is_too_small(dump, capsys):
    """the shared batch buffers are budgeted against /dev/shm (a quarter of its free space, at most 2 GiB).  A ring
    that does not fit whole is shortened (fewer batches in flight, same batches out); one that does not fit four buffer sets is
    refused with a clear error, and the loader factory then decodes with threads into its own buffers instead."""
    d, n_windows, H, W = dump, 7, 32, 64
    B = 2
    per_batch = B * (H * 3 * W * 3 + 4 * H * W * 2 * 4 + 3 * H * W * 4)
    full = L.ProcessWindowLoader(d, 3, H, W, 0, n_windows, B, procs=2, prefetch=2)
    assert full.nring == full.prefetch + full.fill + 2 and L.shm_budget_bytes() > 0
    full.close()
    small = L.ProcessWindowLoader(d, 3, H, W, 0, n_windows, B, procs=2, prefetch=2, shm_budget=4 * per_batch + 100)
    assert small.nring == 4 and small.prefetch + small.fill + 2 == 4
    ref = {w: L.load_window(d, 3, w + 1, H, W) for w in range(n_windows)}
    got = [(s, e, img.copy()) for s, e, (img, _, _) in small]
    assert [(s, e) for s, e, _ in got] == [(0, 2), (2, 4), (4, 6), (6, 7)]
    assert all(np.array_equal(img[i], ref[s + i][0]) for s, e, img in got for i in range(e - s))
    small.close()
    with pytest.raises(L.ShmBudgetError, match="do not fit"):
        L.ProcessWindowLoader(d, 3, H, W, 0, n_windows, B, procs=2, shm_budget=3 * per_batch)
    # the factory's fallback
    real = L.shm_budget_bytes
    L.shm_budget_bytes = lambda: 3 * per_batch
    try:
        fac = S.kitti_window_loader(d, 3, n_windows + 2, H, W, procs=2)
        ld = fac.for_range(0, n_windows, B)
        assert isinstance(ld, L.ThreadedWindowLoader)
        assert [(s, e) for s, e, _ in ld] == [(0, 2), (2, 4), (4, 6), (6, 7)]
        fac.prestart(0, n_windows, B)                                     # nothing to start early: no error either
    finally:
        L.shm_budget_bytes = real
    assert "falling back to the threaded loader" in capsys.readouterr().err


def test_emulated_shard_does_one_ranks_work(dump, c_oracle):
    """run_sequence(emulate=(r, R)): rank r's windows carry the network's poses, the others zero motion."""
    from davo_amd import parse_version, FLAGSHIP_VERSION
    d, n_windows, H, W = dump, 7, 32, 64
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    infer = lambda img, flow, seg: c_oracle.forward(cfg, img, flow, seg, weights)      # noqa: E731
    fac = S.kitti_window_loader(d, 3, n_windows + 2, H, W)
    _, ref = S.run_sequence(infer, fac, n_windows + 2, 2)
    timing = {}
    traj, poses = S.run_sequence(infer, fac, n_windows + 2, 2, timing=timing, emulate=(1, 3))
    lo, hi = S.shard_windows(n_windows, 3, 1)
    assert (lo, hi) == (3, 6) and timing["windows_this_rank"] == 3 and len(traj) == n_windows + 2
    assert np.array_equal(poses[lo:hi], ref[lo:hi]) and not poses[:lo].any() and not poses[hi:].any()


def test_scene_like_dump_is_reproducible_and_far_smaller_than_noise(tmp_path):
    """images="scene": photograph-like strips at the quality the reference's dumps were written with (scipy.misc.imsave's
    default 75, data/preprocess.py:65) — same layout, same arrays, a fraction of the noise strips' bytes."""
    a, b, n = str(tmp_path / "a"), str(tmp_path / "b"), str(tmp_path / "n")
    assert L.write_synthetic_dump(a, 0, 5, 128, 416, images="scene") == 3
    L.write_synthetic_dump(b, 0, 5, 128, 416, images="scene")
    L.write_synthetic_dump(n, 0, 5, 128, 416)
    ia, fa, sa = L.load_window(a, 0, 2, 128, 416)
    ib, fb, sb = L.load_window(b, 0, 2, 128, 416)
    i_n, fn, sn = L.load_window(n, 0, 2, 128, 416)
    assert np.array_equal(ia, ib) and ia.shape == (128, 1248, 3)
    assert np.array_equal(fa, fn) and np.array_equal(sa, sn)                 # only the strips differ
    assert not np.array_equal(L.load_window(a, 0, 1, 128, 416)[0], ia)
    size = lambda d: os.path.getsize(L.window_paths(d, 0, 2)[0])
    assert 20_000 < size(a) < 80_000 < size(n)
    with pytest.raises(ValueError, match="images"):
        L.write_synthetic_dump(a, 1, 5, 128, 416, images="photo")
