"""Sequence driver around the pose path: window sharding across GPUs, the one pose gather,
and the trajectory stitch + KITTI writer that follow the path in the reference's test driver.

Reference: ``test_kitti_pose.py:75-154`` (driver; hot loop ``:133-145``, chain ``:147-149``,
writer ``:150-153``), ``utils/geo_utils.py:12-63,93-119`` (``euler2mat`` clips to [-pi,pi],
R = Rx.Ry.Rz; ``pose_vec2mat`` on ``[rz,ry,rx,tx,ty,tz]``), ``utils/common_utils.py:8-28``.

The windows of a sequence are independent (each ``[2,6]`` output depends only on its own
3 frames), so ranks take contiguous window ranges and run them with no data-path collective;
the only exchange is one all-gather of ``[n,2,6]`` float32 before the sequential 4x4 chain
(48 B per window: 218 KB for KITTI seq 00 — latency only, on RCCL over xGMI; davo_amd/comm.py).
"""
import os

import time

import numpy as np


# ---- geometry (utils/geo_utils.py) --------------------------------------------------------
def euler2mat(z, y, x, dtype=np.float32):
    """R = Rx(x).Ry(y).Rz(z) with angles clipped to [-pi,pi] (geo_utils.py:12-63).
    The reference evaluates this in a float32 TF graph (test_kitti_pose.py:122-123)."""
    z = np.clip(np.asarray(z, dtype), -np.pi, np.pi)
    y = np.clip(np.asarray(y, dtype), -np.pi, np.pi)
    x = np.clip(np.asarray(x, dtype), -np.pi, np.pi)
    n = z.shape[0]
    cz, sz, cy, sy, cx, sx = np.cos(z), np.sin(z), np.cos(y), np.sin(y), np.cos(x), np.sin(x)
    zmat = np.zeros((n, 3, 3), dtype); ymat = np.zeros((n, 3, 3), dtype); xmat = np.zeros((n, 3, 3), dtype)
    zmat[:, 0, 0] = cz; zmat[:, 0, 1] = -sz; zmat[:, 1, 0] = sz; zmat[:, 1, 1] = cz; zmat[:, 2, 2] = 1
    ymat[:, 0, 0] = cy; ymat[:, 0, 2] = sy; ymat[:, 1, 1] = 1; ymat[:, 2, 0] = -sy; ymat[:, 2, 2] = cy
    xmat[:, 0, 0] = 1; xmat[:, 1, 1] = cx; xmat[:, 1, 2] = -sx; xmat[:, 2, 1] = sx; xmat[:, 2, 2] = cx
    return np.matmul(np.matmul(xmat, ymat), zmat)


def pose_vec2mat(vec, dtype=np.float32):
    """[n,6] ``[rz,ry,rx,tx,ty,tz]`` -> [n,4,4] (geo_utils.py:93-119)."""
    vec = np.asarray(vec, dtype).reshape(-1, 6)
    n = vec.shape[0]
    m = np.zeros((n, 4, 4), dtype)
    m[:, :3, :3] = euler2mat(vec[:, 0], vec[:, 1], vec[:, 2], dtype)
    m[:, :3, 3] = vec[:, 3:6]
    m[:, 3, 3] = 1
    return m


def mat2pose_vec(m):
    """Inverse of pose_vec2mat for |ry| < pi/2 (float64): with R = Rx.Ry.Rz,
    R[0,2] = sin y, R[0,1] = -cos y sin z, R[1,2] = -sin x cos y."""
    m = np.asarray(m, np.float64).reshape(-1, 4, 4)
    y = np.arcsin(np.clip(m[:, 0, 2], -1.0, 1.0))
    z = np.arctan2(-m[:, 0, 1], m[:, 0, 0])
    x = np.arctan2(-m[:, 1, 2], m[:, 2, 2])
    return np.concatenate([np.stack([z, y, x], -1), m[:, :3, 3]], -1)


# ---- stitch + writer (test_kitti_pose.py:136-153) -----------------------------------------
def stitch_trajectory(poses, mat_dtype=np.float32):
    """poses [Nw,2,6] (row 0 = tgt->src0, row 1 = tgt->src1 of window w, tgt = frame w+1)
    -> list of Nw+2 float64 4x4 camera poses, first = identity.

    First window contributes T(tgt->src0); every window contributes inv(T(tgt->src1))
    (test_kitti_pose.py:141-145); the chain is float64 (``np.eye(4).astype(float)``, :118)."""
    poses = np.asarray(poses, np.float32)
    if poses.shape[0] == 0:
        return [np.eye(4).astype(float)]
    # all windows' matrices at once (the per-window form of :141-145 spent 19 us of interpreter time per window: 87 ms of a
    # 0.96 s KITTI seq-00 run); same float32 elementwise arithmetic and the same LAPACK inverse per matrix
    first = pose_vec2mat(poses[:1, 0], mat_dtype)[0]                               # :144
    steps = [first] + list(np.linalg.inv(pose_vec2mat(poses[:, 1], mat_dtype)))    # :142,145
    prev = np.eye(4).astype(float)
    out = [prev]
    for p in steps:                                                                # :147-149
        prev = np.dot(prev, p)
        out.append(prev)
    return out


def write_kitti_poses(path, mats):
    """first 3 rows, 12 x str(float) per line (test_kitti_pose.py:150-153)."""
    with open(path, "w") as f:
        for p in mats:
            f.write("%s\n" % " ".join([str(float(x)) for x in np.asarray(p)[:3, :].reshape(12)]))


def read_kitti_poses(path):
    a = np.loadtxt(path).reshape(-1, 3, 4)
    m = np.tile(np.eye(4), (a.shape[0], 1, 1))
    m[:, :3, :] = a
    return m


def relative_pose_vectors(abs_poses):
    """Ground-truth poses -> the [Nw,2,6] tensor a perfect network would emit, i.e. the inverse
    of stitch_trajectory: M(tgt->src0) = P_w^-1 P_{w+1}, M(tgt->src1) = P_{w+2}^-1 P_{w+1}."""
    P = np.asarray(abs_poses, np.float64)
    nw = P.shape[0] - 2
    out = np.zeros((nw, 2, 6))
    for w in range(nw):
        out[w, 0] = mat2pose_vec(np.linalg.inv(P[w]).dot(P[w + 1]))[0]
        out[w, 1] = mat2pose_vec(np.linalg.inv(P[w + 2]).dot(P[w + 1]))[0]
    return out


# ---- window sharding + gather --------------------------------------------------------------
def is_valid_sample(n_frames, tgt_idx, seq_length=3):
    """utils/common_utils.py:16-28 for a single drive."""
    off = int((seq_length - 1) / 2)
    return tgt_idx - off >= 0 and tgt_idx + off < n_frames


def shard_windows(n_windows, world, rank):
    """contiguous range of rank: [r*ceil(Nw/R), min((r+1)*ceil(Nw/R), Nw))"""
    per = -(-n_windows // world)
    lo = min(rank * per, n_windows)
    return lo, min(lo + per, n_windows)


def _pad_batch(img, flow, seg, batch_size):
    n = img.shape[0]
    if n < batch_size:
        pad = batch_size - n
        img = np.concatenate([img, np.repeat(img[-1:], pad, 0)])
        flow = np.concatenate([flow, np.repeat(flow[-1:], pad, 0)])
        seg = np.concatenate([seg, np.repeat(seg[-1:], pad, 0)])
    return img, flow, seg, n


class PoseStream:
    """The library's streaming entry point (include/davo_hip.h: davo_submit / davo_wait) as run_shard's ``stream``: batches
    are issued without waiting for their poses, so the H2D copy and the kernels of batch n+1 run while batch n computes and
    while the loader is asked for batch n+2 - what tf.data's prefetch does for the reference's loop
    (test_kitti_pose.py:133-145, data_loader.py:321-324).  ``hold``: batches whose input arrays the source keeps valid after
    yielding the next one (0 for davo_amd.loader's loaders; arrays that are never recycled can take 8: no copy is waited for).
    ``inflight``: slots = streams the batches rotate through; a slot's stream runs copy -> kernels -> pose copy in order, so at
    batch 1 (44 us of copy latency in front of 127 us of kernels) more slots keep the GPU busier: 8.8 k windows/s with two, 10.6 k with
    three, 12.4 k with four (the library's maximum) on 799 windows; at batch 32 three are best (f16x3 from page-locked host memory:
    14.7 k / 23.1 k / 26.7 k / 22.6 k triplets/s with one to four slots - four forwards of 3,000-workgroup launches at once break up
    the launch plan's whole rounds; float32 8.9 k from two slots on = the HBM-resident rate).  Default: four up to batch 2, else three."""

    def __init__(self, engine, inflight=None, hold=0):
        self.engine, self.hold = engine, hold
        if inflight is None:            # measured: profiles/r05ac_config1_b1.json (batch 1), profiles/r05af_host_api.log (batch 32)
            inflight = 4 if engine.max_batch <= 2 else 3
        engine.set_inflight(inflight)

    def submit(self, img, flow, seg, out):
        self.engine.submit(img, flow, seg, out, self.hold)

    def drain(self):
        self.engine.synchronize()


def run_shard(infer_fn, load_windows, lo, hi, batch_size, timing=None, stream=None):
    """Run windows [lo,hi) in batches; the last batch is padded by repeating its last window
    and the padded outputs are dropped (the reference's complete_batch_size,
    utils/common_utils.py:8-13, would append duplicate poses for B>1; parity is defined on
    B=1 semantics).

    ``stream`` (a PoseStream) replaces ``infer_fn``: batches are submitted and their poses collected at the end, so input
    wait, copies and kernels overlap instead of adding up; the poses are the same bits (same kernels on the same batches).

    ``load_windows`` is either a callable ``(s, e) -> (img, flow, seg)`` or an iterable of
    ``(s, e, (img, flow, seg))`` in window order (davo_amd.loader.ThreadedWindowLoader: the next
    batches are decoded while this one is on the GPU).  ``timing`` (a dict) receives the seconds spent waiting for
    input and inside ``infer_fn``."""
    out = np.zeros((hi - lo, 2, 6), np.float32)
    tails = []                                            # streamed partial batches: (padded poses, where they go)
    if callable(load_windows):
        batches = ((s, min(s + batch_size, hi), load_windows(s, min(s + batch_size, hi))) for s in range(lo, hi, batch_size))
    else:
        batches = iter(load_windows)
    t_load = t_fwd = 0.0
    while True:
        t0 = time.perf_counter()
        item = next(batches, None)                       # with a prefetching loader: the time the GPU side WAITED for input
        t1 = time.perf_counter()
        t_load += t1 - t0
        if item is None:
            break
        s, e, (img, flow, seg) = item
        img, flow, seg, n = _pad_batch(img, flow, seg, batch_size)
        if stream is None:
            out[s - lo:e - lo] = np.asarray(infer_fn(img, flow, seg))[:n]
        elif n == batch_size:
            stream.submit(img, flow, seg, out[s - lo:e - lo])          # delivered straight into its rows of `out`
        else:
            full = np.empty((batch_size, 2, 6), np.float32)
            stream.submit(img, flow, seg, full)
            tails.append((full, s - lo, n, (img, flow, seg)))           # the padded copies stay alive until the drain
        t_fwd += time.perf_counter() - t1
    if stream is not None:
        t1 = time.perf_counter()
        stream.drain()
        for full, at, n, _ in tails:
            out[at:at + n] = full[:n]
        t_drain = time.perf_counter() - t1
    if timing is not None:
        timing["load_wait_s"] = timing.get("load_wait_s", 0.0) + t_load
        # synchronous driver: H2D copies + kernels + D2H of the poses (davo_forward).  Streamed: the time inside davo_submit - issuing
        # the batch and waiting for ITS H2D copy, while the previous batches compute - and drain_s, the wait for the last batches
        timing["forward_s"] = timing.get("forward_s", 0.0) + t_fwd
        if stream is not None:
            timing["drain_s"] = timing.get("drain_s", 0.0) + t_drain
            timing["streamed"] = True
    return out


def gather_poses(local, n_windows, world, rank, comm=None):
    """All ranks -> [Nw,2,6] on every rank: one all-gather of equal padded counts.

    ``comm`` is the run's communicator: ``davo_amd.comm.RcclComm`` (librccl through the C ABI,
    include/davo_hip.h: davo_allgather_poses) in the product; anything with the same
    ``allgather(local, n_per_rank) -> (all, ms)`` in the CPU tests of the sharding logic."""
    if world == 1 and comm is None:
        return np.asarray(local, np.float32)
    if comm is None:
        raise ValueError("world size %d needs a communicator (davo_amd.comm.RcclComm)" % world)
    per = -(-n_windows // world)
    full, _ = comm.allgather(np.ascontiguousarray(local, np.float32), per)
    # rank r's slot holds its hi-lo windows first; slots are in rank order = window order
    parts = []
    for r in range(world):
        lo, hi = shard_windows(n_windows, world, r)
        parts.append(full[r * per:r * per + (hi - lo)])
    return np.concatenate(parts, 0)


def run_sequence(infer_fn, load_windows, n_frames, batch_size, rank=0, world=1, comm=None, timing=None, emulate=None, stream=None):
    """The driver loop of test_kitti_pose.py:133-149, sharded: returns the Nf 4x4 poses on
    every rank (the stitch is cheap and sequential; rank 0 writes the file).  ``timing`` (a dict) receives this
    rank's seconds per stage: load_wait_s, forward_s, gather_s, stitch_s.

    ``emulate=(r, R)`` (measurement aid, one process): do exactly what rank r of R would do - its window shard, the gather
    (through ``comm`` at its real world size, normally 1), the stitch of the whole sequence - with the other ranks' windows left
    at zero motion.  ``stream`` (a PoseStream): the shard runs through the library's streaming entry point (run_shard)."""
    n_windows = n_frames - 2
    lo, hi = shard_windows(n_windows, *((world, rank) if emulate is None else (emulate[1], emulate[0])))
    if hasattr(load_windows, "for_range"):               # a loader factory: build this rank's prefetching loader
        load_windows = load_windows.for_range(lo, hi, batch_size)
    local = run_shard(infer_fn, load_windows, lo, hi, batch_size, timing, stream)
    t0 = time.perf_counter()
    if emulate is None:
        poses = gather_poses(local, n_windows, world, rank, comm)
    else:
        poses = np.zeros((n_windows, 2, 6), np.float32)
        poses[lo:hi] = gather_poses(local, hi - lo, world, rank, comm)
    t1 = time.perf_counter()
    traj = stitch_trajectory(poses)
    if timing is not None:
        timing.update(gather_s=t1 - t0, stitch_s=time.perf_counter() - t1, windows_this_rank=hi - lo)
    return traj, poses


# ---- on-disk inputs (data_loader.py:241-325; doc/preprocessing.md:50-114) -------------------
class kitti_window_loader:
    """Loader factory over the reference's dump (davo_amd/loader.py): ``for_range(lo, hi, B)`` gives the
    threaded, prefetching batch iterator of a rank's shard; calling it ``(s, e)`` loads one batch inline."""

    def __init__(self, concat_img_dir, seq, n_frames, H, W, workers=4, prefetch=2, alloc=None, decode_procs=0, procs=0,
                 pin=None, unpin=None, seg_planes=None, hold=0):
        self.dir, self.seq, self.n_frames, self.H, self.W = concat_img_dir, seq, n_frames, H, W
        self.workers, self.prefetch, self.alloc, self.decode_procs = workers, prefetch, alloc, decode_procs
        self.procs, self.pin, self.unpin, self.seg_planes = procs, pin, unpin, seg_planes
        self.hold = hold          # batches the process loader keeps valid beyond the one being asked for (PoseStream(hold=...)); the threaded loader gives 0

    def prestart(self, lo, hi, batch_size):
        """Build this rank's process loader now and let it start filling batches: the workers fork, attach the buffers and
        decode the first batches behind the caller's GPU set-up instead of in front of the first batch.  for_range hands it out."""
        if self.procs > 0:
            ld = self.for_range(lo, hi, batch_size)
            if hasattr(ld, "start"):
                self._early = ((lo, hi, batch_size), ld.start())

    def for_range(self, lo, hi, batch_size):
        from . import loader as L
        early = getattr(self, "_early", None)
        if early is not None and early[0] == (lo, hi, batch_size):
            self._early = None
            return early[1]
        if self.procs > 0:      # worker processes fill shared (page-locked) batch buffers: davo_amd/loader.py, ProcessWindowLoader
            try:
                return L.ProcessWindowLoader(self.dir, self.seq, self.H, self.W, lo, hi, batch_size, self.procs, self.prefetch,
                                             pin=self.pin, unpin=self.unpin, hold=self.hold,
                                             seg_planes=L.SEG_PLANES_SOURCES if self.seg_planes is None else self.seg_planes)
            except L.ShmBudgetError as e:      # e.g. a container with the usual 64 MB /dev/shm: decode threads into pinned buffers instead
                import sys
                print("davo_amd: %s - falling back to the threaded loader" % e, file=sys.stderr)
        return L.kitti_loader(self.dir, self.seq, self.H, self.W, lo, hi, batch_size, self.workers, self.prefetch, self.alloc, self.decode_procs)

    def __call__(self, s, e):
        from .loader import load_window
        parts = [load_window(self.dir, self.seq, w + 1, self.H, self.W) for w in range(s, e)]
        return tuple(np.stack([p[k] for p in parts]) for k in range(3))


def synthetic_window_loader(H, W, seed=None):
    from . import synth

    def load(s, e):
        return synth.make_inputs(e - s, H, W, seed=synth.SEED if seed is None else seed, first_window=s)
    return load
