"""GPU tests of the streaming entry point (include/davo_hip.h: davo_submit / davo_wait), of the sequence driver on top of it,
of davo_set_stream, and of the rank launcher with more than one rank on the box's GPU.  The oracle is the checker only.

Reference: the loop being streamed is test_kitti_pose.py:133-145 behind data_loader.py:321-324's prefetch."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from davo_amd import DAVO, Engine, synth, parse_version, FLAGSHIP_VERSION
from davo_amd import sequence as S

from helpers import assert_pose_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(cfg, H, W, B, weights, precision="f16x3"):
    e = Engine(cfg, H, W, B)
    e.load_weights(weights)
    e.set_precision(precision)
    return e


def _rescaled(weights, shift):
    """the same network with cnv3's activations 2^shift larger (ReLU is homogeneous): trips the f16x3 range guard"""
    w = dict(weights)
    s = np.float32(2.0 ** shift)
    w["pose_exp_net/cnv3/weights"] = weights["pose_exp_net/cnv3/weights"] * s
    w["pose_exp_net/cnv3/biases"] = weights["pose_exp_net/cnv3/biases"] * s
    w["pose_exp_net/cnv4/weights"] = weights["pose_exp_net/cnv4/weights"] / s
    return w


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("inflight,hold", [(1, 0), (2, 0), (2, 2), (3, 9)])
def test_submit_equals_the_synchronous_forward_to_the_bit(precision, inflight, hold):
    """Eleven batches (more than the pose ring of eight and the four staging sets) of three different sizes through
    davo_submit: same bits as davo_forward on the same batches.  With hold = 0 ONE set of host arrays is overwritten with the
    next batch right after each submit returns (the loader contract: a batch is valid until the next one is asked for)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    H, W, Bmax, n = 64, 96, 3, 11
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, Bmax, weights, precision)
    data = [synth.make_inputs((3, 1, 2)[k % 3], H, W, first_window=2 * k) for k in range(n)]
    want = [e.forward(*d) for d in data]
    e.set_inflight(inflight)
    outs = [np.full((d[0].shape[0], 2, 6), np.nan, np.float32) for d in data]
    if hold == 0:
        bufs = [np.empty((Bmax,) + a.shape[1:], a.dtype) for a in data[0]]
        for k, d in enumerate(data):
            b = d[0].shape[0]
            for buf, a in zip(bufs, d):
                buf[:b] = a
            e.submit(bufs[0][:b], bufs[1][:b], bufs[2][:b], outs[k])
            for buf in bufs:
                buf[...] = 255 if buf.dtype == np.uint8 else np.nan      # consumed: the caller may do anything to them now
    else:
        for k, d in enumerate(data):
            e.submit(*d, outs[k], hold=hold)
    assert 0 < e.pending() <= 8
    e.wait(2)
    assert e.pending() <= 2 and not np.isnan(outs[n - 3]).any()
    e.synchronize()
    assert e.pending() == 0
    for k in range(n):
        assert np.array_equal(outs[k], want[k]), (k, np.abs(outs[k] - want[k]).max())
    # the synchronous entry point after a stream, and a stream after it
    assert np.array_equal(e.forward(*data[1]), want[1])
    e.submit(*data[0], outs[0])
    assert np.array_equal(e.forward(*data[2]), want[2]) and e.pending() == 0        # davo_forward delivers what is under way first
    assert np.array_equal(outs[0], want[0])
    e.close()


def test_submit_argument_checks():
    cfg = parse_version(FLAGSHIP_VERSION)
    e = _engine(cfg, 64, 96, 2, synth.make_weights(cfg))
    img, flow, seg = synth.make_inputs(2, 64, 96)
    with pytest.raises(ValueError):
        e.submit(img, flow, seg, np.empty((2, 2, 6), np.float64))
    with pytest.raises(ValueError):
        e.submit(img, flow, seg, np.empty((3, 2, 6), np.float32))
    with pytest.raises(ValueError):
        e.submit(img[:, :, ::2], flow, seg, np.empty((2, 2, 6), np.float32))
    with pytest.raises(ValueError):
        e.submit(img.astype(np.float32), flow, seg, np.empty((2, 2, 6), np.float32), hold=1)     # a converted copy cannot be held
    big = synth.make_inputs(3, 64, 96)
    with pytest.raises(ValueError):
        e.submit(*big, np.empty((3, 2, 6), np.float32))                                         # batch > max_batch
    assert e.pending() == 0
    e.close()


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 128, 416)])
def test_streamed_batches_of_a_guard_tripping_checkpoint_are_reissued_from_the_contexts_copy(c_oracle, B, H, W):
    """A checkpoint whose cnv3 activations leave the fp16-pair range, thirteen batches through davo_submit with ONE recycled set
    of host arrays: every batch that went out on the old scales is re-issued at its verdict from the context's own snapshot into
    the context's own pose buffer, and the caller's arrays receive oracle-grade poses - whatever "stable_inputs" says about the
    CALLER's buffers (the staging sets are the library's, and they are overwritten four batches later)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    n = 13
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, _rescaled(weights, 16), "f16x3")
    e.set_option("stable_inputs", 1)
    e.set_inflight(2)
    data = [synth.make_inputs(B, H, W, first_window=3 * k) for k in range(n)]
    wants = [c_oracle.forward(cfg, *d, weights) for d in data]
    outs = [np.full((B, 2, 6), np.nan, np.float32) for _ in range(n)]
    bufs = [np.empty_like(a) for a in data[0]]
    for k in range(n):
        for buf, a in zip(bufs, data[k]):
            buf[...] = a
        e.submit(*bufs, outs[k])
    e.synchronize()
    for k in range(n):
        assert_pose_close(outs[k], wants[k], "streamed, guard tripped, batch %d" % k)
    st = e.range_stats()
    assert 1 <= st["reissued"] <= 9 and st["recalibrations"] >= 1 and st["f32_batches"] == 0, st
    for k in range(n):                                                   # scales settled: nothing more is re-issued
        e.submit(*data[k], outs[k], hold=8)
    e.synchronize()
    assert e.range_stats() == st
    for k in range(n):
        assert_pose_close(outs[k], wants[k], "streamed, settled, batch %d" % k)
    e.close()


class _CachedWindows:
    def __init__(self, H, W, n=64, first_window=0):
        self.n = n
        self.img, self.flow, self.seg = synth.make_inputs(n, H, W, first_window=first_window)

    def __call__(self, s, e):
        idx = [(w * 37) % self.n for w in range(s, e)]
        return self.img[idx], self.flow[idx], self.seg[idx]


@pytest.mark.parametrize("nw,B", [(799, 1), (4539, 64)])
def test_streamed_driver_equals_the_synchronous_driver_to_the_bit(nw, B):
    """configs[0] (seq 03: 799 windows, batch 1) and configs[3] (seq 00: 4,539 windows, batch 64, ragged last batch) through
    run_shard: the streaming entry point (two in flight) delivers the same bits as one davo_forward per batch."""
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    load = _CachedWindows(128, 416, 64)
    e = _engine(cfg, 128, 416, B, weights, "f16x3")
    e.set_option("host_chunk", 0)                 # davo_forward in one piece like davo_submit (sub-batches of 8 sum the pose head's tiles per sub-batch)
    sync = S.run_shard(e.forward, load, 0, nw, B)
    timing = {}
    streamed = S.run_shard(None, load, 0, nw, B, timing, S.PoseStream(e, inflight=2))
    assert np.array_equal(streamed, sync)
    assert timing["streamed"] and e.pending() == 0
    lo, hi = S.shard_windows(nw, 8, 7)            # a rank's (short) shard of eight
    assert np.array_equal(S.run_shard(None, load, lo, hi, B, None, S.PoseStream(e)), S.run_shard(e.forward, load, lo, hi, B))
    e.close()


def test_inference_from_an_iterator_runs_one_batch_ahead(c_oracle):
    """DAVO.setup_inference(input_img_uint8=<iterator>): inference() returns batch n while batch n+1 is already on the GPU
    (the tf.data pull model of davo.py:1533-1569 / data_loader.py:321-324); same bits as the array form, StopIteration at the end."""
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    data = [synth.make_inputs(2, 64, 96, first_window=2 * k) for k in range(5)]
    pulled = []

    def gen():
        for k, d in enumerate(data):
            pulled.append(k)
            yield d
    a = DAVO(FLAGSHIP_VERSION)
    a.load_weights(weights)
    a.setup_inference(64, 96, "davo", 3, 2, input_img_uint8=gen())
    b = DAVO(FLAGSHIP_VERSION)
    b.load_weights(weights)
    b.setup_inference(64, 96, "davo", 3, 2)
    for k in range(5):
        got = a.inference(None, "pose")["pose"]
        assert pulled == list(range(min(k + 2, 5)))                      # one batch ahead of what has been returned
        assert np.array_equal(got, b.inference(None, "pose", inputs=data[k])["pose"])
        if k == 0:
            assert_pose_close(got, c_oracle.forward(cfg, *data[0], weights), "iterator batch 0")
    with pytest.raises(StopIteration):
        a.inference(None, "pose")
    a.engine.close(); b.engine.close()


# ---- davo_set_stream ---------------------------------------------------------------------------------------------------
class _Hip:
    """the three HIP calls a test needs to own a stream (tests only: the product binds HIP through libdavo_hip.so)"""

    def __init__(self):
        self.L = ctypes.CDLL("libamdhip64.so")
        self.L.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.L.hipStreamDestroy.argtypes = [ctypes.c_void_p]
        self.L.hipStreamSynchronize.argtypes = [ctypes.c_void_p]

    def stream(self):
        s = ctypes.c_void_p()
        assert self.L.hipStreamCreate(ctypes.byref(s)) == 0
        return s

    def destroy(self, s):
        assert self.L.hipStreamSynchronize(s) == 0 and self.L.hipStreamDestroy(s) == 0


def test_caller_owned_stream_with_pending_tickets_and_a_switch(c_oracle):
    """davo_set_stream: ten batches of a guard-tripping checkpoint on the CALLER's stream, the inputs recycled (overwritten in
    stream order) behind every batch; then the context is switched to a second stream and back to its own while tickets are
    still pending, and the first stream is destroyed right after the switch.  Every batch's poses come out oracle-grade: the
    switch judges (and re-issues) what was issued on the stream the context leaves."""
    cfg = parse_version(FLAGSHIP_VERSION)
    B, H, W, n = 2, 64, 96, 10
    hip = _Hip()
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, _rescaled(weights, 16), "f16x3")
    s1, s2 = hip.stream(), hip.stream()
    L = e._L
    assert L.davo_set_stream(e._ctx, s1) == 0
    data = [synth.make_inputs(B, H, W, first_window=3 * k) for k in range(n + 4)]
    wants = [c_oracle.forward(cfg, *d, weights) for d in data]
    d_in = [e.alloc(a.nbytes) for a in data[0]]
    poses = [e.alloc(B * 48) for _ in range(n + 4)]
    for k in range(n):                                                  # uploads run on the context's current stream = s1
        for buf, a in zip(d_in, data[k]):
            buf.upload(a)
        e.forward_device(B, *d_in, poses[k])
    assert L.davo_set_stream(e._ctx, s2) == 0                           # tickets of batches 2..9 are pending on s1 here
    hip.destroy(s1)
    st = e.range_stats()
    assert st["reissued"] >= 8 and st["recalibrations"] >= 1, st        # all judged by the switch (batches 0, 1 when their ring slots came round)
    for k in range(n):
        assert_pose_close(poses[k].download((B, 2, 6)), wants[k], "caller's stream, batch %d" % k)
    for k in range(n, n + 2):                                           # on the second stream, scales settled
        for buf, a in zip(d_in, data[k]):
            buf.upload(a)
        e.forward_device(B, *d_in, poses[k])
    assert L.davo_set_stream(e._ctx, None) == 0                         # back to the context's own stream, two tickets pending on s2
    hip.destroy(s2)
    for k in range(n + 2, n + 4):
        for buf, a in zip(d_in, data[k]):
            buf.upload(a)
        e.forward_device(B, *d_in, poses[k])
    e.synchronize()
    assert e.range_stats() == st
    for k in range(n, n + 4):
        assert_pose_close(poses[k].download((B, 2, 6)), wants[k], "after the switch, batch %d" % k)
    # a caller-owned stream and several batches in flight exclude each other, both ways round
    s3 = hip.stream()
    e.set_inflight(2)
    assert L.davo_set_stream(e._ctx, s3) == -1 and b"inflight" in L.davo_last_error(e._ctx)
    e.set_inflight(1)
    assert L.davo_set_stream(e._ctx, s3) == 0
    with pytest.raises(ValueError):
        e.set_inflight(2)
    assert L.davo_set_stream(e._ctx, None) == 0
    hip.destroy(s3)
    e.close()


def test_recycled_pose_buffer_never_receives_a_late_reissue(c_oracle):
    """Two alternating pose buffers (bench.py's two-in-flight leg, any double-buffered caller) with a guard-tripping checkpoint,
    eleven batches: batches 0..7 go out on the old scales; issuing batch 8 needs batch 0's ring slot, so batch 0 is judged and
    the scales re-calibrated; batches 8, 9, 10 run in range.  At synchronize batches 1..7 are re-issued - AFTER batches 9 and 10
    wrote the two buffers.  A re-issue goes to a buffer of the context and is copied out only if no later batch has taken
    its pose buffer: each buffer ends up with the LAST batch issued into it (round 4: batches 6 and 7 landed there)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    B, H, W, n = 2, 64, 96, 11
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, _rescaled(weights, 16), "f16x3")
    e.set_inflight(2)
    data = [synth.make_inputs(B, H, W, first_window=3 * k) for k in range(n)]
    wants = [c_oracle.forward(cfg, *d, weights) for d in data]
    sets = [tuple(e.alloc(a.nbytes).upload(a) for a in d) for d in data]
    poses = [e.alloc(B * 48), e.alloc(B * 48)]
    for k in range(n):
        e.forward_device(B, *sets[k], poses[k % 2])
    e.synchronize()
    assert np.abs(wants[n - 1] - wants[n - 3]).max() > 1e-3              # the batches are told apart by their poses
    assert_pose_close(poses[(n - 1) % 2].download((B, 2, 6)), wants[n - 1], "last batch into buffer %d" % ((n - 1) % 2))
    assert_pose_close(poses[(n - 2) % 2].download((B, 2, 6)), wants[n - 2], "last batch into buffer %d" % ((n - 2) % 2))
    st = e.range_stats()
    assert st["reissued"] == 8 and st["recalibrations"] >= 1, st
    # a buffer that was NOT taken by a later batch does receive its re-issue (batch 7 of eight, all on the old scales)
    e2 = _engine(cfg, H, W, B, _rescaled(weights, 16), "f16x3")
    sets2 = [tuple(e2.alloc(a.nbytes).upload(a) for a in d) for d in data[:8]]
    own = [e2.alloc(B * 48) for _ in range(8)]
    for k in range(8):
        e2.forward_device(B, *sets2[k], own[k])
    e2.synchronize()
    for k in range(8):
        assert_pose_close(own[k].download((B, 2, 6)), wants[k], "own buffer, batch %d" % k)
    e.close(); e2.close()


# ---- the rank launcher with two ranks on the box's GPU ----------------------------------------------------------------
_RANK = r"""
import os, sys
sys.path.insert(0, @ROOT@)
import davo_amd                                    # binds this rank to DAVO_CPU_SLICE before numpy starts a thread
import numpy as np
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION
from oracle import c_oracle
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
want_cpus = sorted(int(c) for c in os.environ["DAVO_CPU_SLICE"].split(","))
assert sorted(os.sched_getaffinity(0)) == want_cpus, (sorted(os.sched_getaffinity(0)), want_cpus)
if len(sys.argv) > 2 and int(sys.argv[2]) == rank:
    sys.exit(9)                                    # the failing-rank case
cfg = parse_version(FLAGSHIP_VERSION)
weights = synth.make_weights(cfg)
e = Engine(cfg, 64, 96, 2, device=0)               # both ranks on the one GPU of the box: no communicator (RCCL refuses that), HIP does not mind
e.load_weights(weights)
data = [synth.make_inputs(2, 64, 96, first_window=10 * rank + 2 * k) for k in range(3)]
outs = [np.empty((2, 2, 6), np.float32) for _ in data]
e.set_inflight(2)
for d, o in zip(data, outs):
    e.submit(*d, o)
e.synchronize()
err = max(float(np.abs(o - c_oracle.forward(cfg, *d, weights)).max()) for d, o in zip(data, outs))
assert err < 1e-4, err
open(os.path.join(sys.argv[1], "ok%d" % rank), "w").write("%s %.3g" % (",".join(map(str, want_cpus)), err))
e.close()
"""


def test_two_ranks_through_the_launcher_on_one_gpu(tmp_path, c_oracle):
    """spawn_ranks with nprocs = 2 on the GPU box: each rank binds itself to its CPU slice (no exec hop), builds
    an engine on device 0, streams three batches and checks them against the oracle; then a run whose rank 1 fails returns its
    exit code and stops the other rank."""
    from davo_amd.launch import spawn_ranks, cpu_slices
    script = tmp_path / "rank.py"
    script.write_text(_RANK.replace("@ROOT@", repr(ROOT)))
    assert spawn_ranks([str(script), str(tmp_path)], 2, timeout=600) == 0
    got = [open(tmp_path / ("ok%d" % r)).read().split()[0] for r in range(2)]
    want = cpu_slices(2)
    if len(os.sched_getaffinity(0)) >= 2:
        assert [set(int(c) for c in g.split(",")) for g in got] == want and not (want[0] & want[1])
    assert spawn_ranks([str(script), str(tmp_path / "x"), "1"], 2, timeout=600) == 9
    # the parent of the ranks never loaded the HIP library ... in a process of its own (this test process has, long ago)
    code = ("import sys; sys.path.insert(0, %r); from davo_amd.launch import spawn_ranks; from davo_amd import _lib; "
            "rc = spawn_ranks([%r, %r], 2, timeout=600); assert _lib._lib is None; "
            "assert not any('libdavo_hip' in l or 'libamdhip64' in l for l in open('/proc/self/maps')); sys.exit(rc)"
            % (ROOT, str(script), str(tmp_path)))
    assert subprocess.call([sys.executable, "-c", code]) == 0
