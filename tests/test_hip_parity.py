"""GPU parity tests: the HIP path (through the C ABI) vs the CPU oracle and the committed
golden outputs.  Bar: max|pose - ref| <= 1e-4 AND <= 1e-4*max|ref|.
The oracle is the checker only; nothing here falls back to it."""
import os

import numpy as np
import pytest

from davo_amd import DAVO, Engine, conv2d_same, synth, parse_version, FLAGSHIP_VERSION, DavoError
from oracle import davo_oracle as O

from helpers import load_golden, case_inputs, assert_pose_close, assert_layer_close

pytestmark = pytest.mark.gpu


# ---- the MFMA implicit-GEMM kernel alone ---------------------------------------------------
@pytest.mark.parametrize("k,stride,rate,cin,cout,N,H,W", [
    (7, 2, 1, 8, 16, 2, 32, 48),        # cnv1 shape class: Cin < 32 (4 taps per k-chunk), N tile half empty
    (5, 2, 1, 16, 32, 2, 33, 47),       # cnv2 class, odd sizes -> asymmetric SAME pad
    (3, 1, 2, 32, 64, 3, 32, 104),      # cnv3
    (3, 1, 4, 64, 128, 1, 32, 104),     # cnv4
    (3, 1, 8, 128, 256, 1, 32, 104),    # cnv5: rate 8, two N tiles
    (3, 1, 2, 256, 256, 1, 16, 40),     # cnv6 fused width
    (3, 2, 1, 128, 256, 2, 32, 104),    # cnv7: stride 2, pad (0,1)
    (1, 1, 1, 256, 3, 1, 16, 52),       # pred: 1x1, N = 3
    (3, 1, 1, 4, 5, 1, 9, 11),          # ragged everything: M tail, K tail, N tail
    (3, 2, 3, 16, 40, 2, 21, 19),
])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_conv_kernel_vs_oracle(k, stride, rate, cin, cout, N, H, W, precision):
    if precision == "f16x3" and cin < 8:
        pytest.skip("f16x3 k-chunks are built from 8-channel units")
    rng = np.random.RandomState(1000 * k + 100 * stride + 10 * rate + cin)
    x = rng.randn(N, H, W, cin).astype(np.float32)
    w = (rng.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    b = (rng.randn(cout) * 0.1).astype(np.float32)
    for relu in (True, False):
        want = O.conv2d_same(x.astype(np.float64), w, b, stride, rate, relu)
        got = conv2d_same(x, w, b, stride, rate, relu, precision=precision)
        assert_layer_close(got, want, "conv %s k%d s%d r%d %d->%d relu=%s" % (precision, k, stride, rate, cin, cout, relu), rtol=5e-6)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_conv_kernel_asymmetric_operand(precision):
    """A = identity-like input with an asymmetric kernel catches a transposed C/D map."""
    x = np.zeros((1, 8, 8, 32), np.float32)
    x[0, 3, 5, :] = np.arange(1, 33)
    w = np.zeros((1, 1, 32, 64), np.float32)
    for ci in range(32):
        w[0, 0, ci, (3 * ci + 1) % 64] = 1.0 + ci
    got = conv2d_same(x, w, np.zeros(64, np.float32), 1, 1, False, precision=precision)
    want = O.conv2d_same(x.astype(np.float64), w, np.zeros(64), 1, 1, False)
    assert np.array_equal(got, want.astype(np.float32))


# ---- whole path, layer by layer ------------------------------------------------------------
def _engine(cfg, H, W, B, weights, precision="f16x3"):
    e = Engine(cfg, H, W, B)
    e.load_weights(weights)
    e.set_precision(precision)
    return e


PRECISIONS = ["f16x3", "f32"]


def _repack10_to_8(p10):
    return np.concatenate([p10[..., 0:3], p10[..., 5:10]], axis=-1)


@pytest.mark.parametrize("impl,precision", [("mfma", "f16x3"), ("mfma", "f32"), ("direct", "f32")])
def test_layers_flagship_128x416(impl, precision):
    cfg = parse_version(FLAGSHIP_VERSION)
    B, H, W = 1, 128, 416
    img, flow, seg = synth.make_inputs(B, H, W)
    weights = synth.make_weights(cfg)
    keep = {}
    want = O.forward(cfg, img, flow, seg, weights, np.float64, keep)
    e = _engine(cfg, H, W, B, weights, precision)
    e.set_impl(impl)
    e.set_option("fuse_pose", 0)                 # keep cnv7 materialised so it can be inspected
    got = e.forward(img, flow, seg)
    tab = e.debug_read("att_table", (B, 3, 19))
    assert_layer_close(tab[:, 1:], O.attention_tables(cfg, flow, weights, np.float64)[:, 1:], "att_table", rtol=2e-6)
    p10 = keep["packed"].reshape(2 * B, H, W, 10)
    if impl == "mfma":
        assert_layer_close(e.debug_read("packed", (2 * B, H, W, 8)), _repack10_to_8(p10), "packed", rtol=2e-6)
    else:
        assert_layer_close(e.debug_read("packed", (2 * B, H, W, 10)), p10, "packed10", rtol=2e-6)
    for name, shp in (("cnv1", (64, 208, 16)), ("cnv2", (32, 104, 32)), ("cnv3", (32, 104, 64)),
                      ("cnv4", (32, 104, 128)), ("cnv5", (32, 104, 256))):
        assert_layer_close(e.debug_read(name, (2 * B,) + shp), keep[name], name)
    c6 = e.debug_read("cnv6", (2 * B, 32, 104, 256))
    assert_layer_close(c6[..., :128], keep["rotation/cnv6"], "rot cnv6")
    assert_layer_close(c6[..., 128:], keep["translation/cnv6"], "trans cnv6")
    c7 = e.debug_read("cnv7", (2 * B, 16, 52, 512))
    assert_layer_close(c7[..., :256], keep["rotation/cnv7"], "rot cnv7")
    assert_layer_close(c7[..., 256:], keep["translation/cnv7"], "trans cnv7")
    assert_pose_close(got, want, "pose (%s, %s)" % (impl, precision))
    e.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_golden_cases_all_variants(c_oracle, precision):
    g = load_golden()
    for name, case in g["cases"].items():
        cfg, img, flow, seg, weights = case_inputs(case)
        e = _engine(cfg, case["H"], case["W"], case["B"], weights, precision)
        got = e.forward(img, flow, seg)
        assert_pose_close(got, np.array(case["pose"]), name + " vs golden")
        if case["H"] * case["W"] <= 128 * 416:
            assert_pose_close(got, c_oracle.forward(cfg, img, flow, seg, weights), name + " vs C oracle")
        e.close()


@pytest.mark.parametrize("precision,fuse_pose", [("f16x3", 1), ("f16x3", 0), ("f32", 0)])
def test_batching_is_per_sample_and_deterministic(precision, fuse_pose):
    """A window's pose must not depend on what else is in the batch or on max_batch.  With separate
    pose-head kernels every sum has a batch-independent order, so the poses are bit-identical wherever
    the window sits; with the pose head fused into cnv7's tiles (f16x3 default) the spatial sum is cut
    at tile boundaries, which move with the window's position: same value to float32 rounding
    (asserted at 1e-6 relative), still bitwise reproducible run to run."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(5, 128, 416)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, 128, 416, 8, weights, precision)
    e.set_option("fuse_pose", fuse_pose)
    if not fuse_pose:
        e.set_option("split_k", 0)      # bit-identity across batch sizes is a property of the single K chain (split_k: to rounding)

    def same(a, b):
        if fuse_pose:
            return np.abs(a - b).max() <= 1e-6 * np.abs(b).max()
        return np.array_equal(a, b)
    all5 = e.forward(img, flow, seg)
    again = e.forward(img, flow, seg)
    assert np.array_equal(all5, again)                       # bitwise reproducible
    for i in (0, 3, 4):
        one = e.forward(img[i:i + 1], flow[i:i + 1], seg[i:i + 1])
        assert same(one[0], all5[i])
    rev = e.forward(img[::-1], flow[::-1], seg[::-1])
    assert same(rev[::-1], all5)
    e.close()


def test_f16x3_close_to_f32_path():
    """The two arithmetic modes agree far inside the parity bar (the split keeps 22 bits)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(2, 128, 416)
    w = synth.make_weights(cfg)
    e = _engine(cfg, 128, 416, 2, w, "f32")
    p32 = e.forward(img, flow, seg)
    e.set_precision("f16x3")
    p16 = e.forward(img, flow, seg)
    assert np.abs(p16 - p32).max() <= 2e-6 * np.abs(p32).max()
    e.close()


def test_linearity_of_pose_head():
    """pred + mean are linear: scaling the pred kernel and bias by a scales the pose by a."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(1, 64, 96)
    w = synth.make_weights(cfg)
    e = _engine(cfg, 64, 96, 1, w)
    p1 = e.forward(img, flow, seg)
    w2 = dict(w)
    for h in ("rotation", "translation"):
        for k in ("weights", "biases"):
            n = "pose_exp_net/pose/%s/pred/%s" % (h, k)
            w2[n] = w[n] * np.float32(0.5)
    e.load_weights(w2)
    p2 = e.forward(img, flow, seg)
    assert np.allclose(p2, 0.5 * p1, rtol=2e-6, atol=1e-9)
    e.close()


def test_ignore_label_everywhere_zeroes_sources():
    """seg = 255 everywhere -> src rgb and flow are fully masked: the pose equals the pose of a
    window whose source frames are black(-ish: value 0 after masking) with zero flow."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(1, 64, 96)
    w = synth.make_weights(cfg)
    seg255 = np.full_like(seg, 255.0)
    want = O.forward(cfg, img, flow, seg255, w)
    e = _engine(cfg, 64, 96, 1, w)
    got = e.forward(img, flow, seg255)
    assert_pose_close(got, want, "all-ignore")
    flow2 = flow.copy(); flow2[:, :2] *= 3.0               # masked out -> no effect at all
    assert np.array_equal(e.forward(img, flow2, seg255), got)
    e.close()


# ---- the reference call surface ------------------------------------------------------------
def test_davo_class_call_surface():
    g = load_golden()["cases"]["flagship_b2_128x416"]
    cfg, img, flow, seg, weights = case_inputs(g)
    system = DAVO(version=FLAGSHIP_VERSION)
    system.setup_inference(128, 416, "davo", 3, 2, img, input_flow=flow, input_depth=seg, input_seglabel=seg)
    system.load_weights(weights)
    pred = system.inference(None, mode='pose')
    assert set(pred) == {'pose'} and pred['pose'].shape == (2, 2, 6) and pred['pose'].dtype == np.float32
    assert_pose_close(pred['pose'], np.array(g["pose"]), "DAVO.inference")
    with pytest.raises(NotImplementedError):
        system.inference(None, mode='feature')

    # pull-model: an iterator of batches stands in for the tf.data get_next()
    def batches():
        for i in range(2):
            yield img[i:i + 1], flow[i:i + 1], seg[i:i + 1]
    s2 = DAVO(version=FLAGSHIP_VERSION)
    s2.load_weights(weights)
    s2.setup_inference(128, 416, "davo", 3, 1, batches())
    for i in range(2):
        assert_pose_close(s2.inference(None, 'pose')['pose'][0], np.array(g["pose"])[i], "window %d" % i)


def test_error_behaviour():
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    img, flow, seg = synth.make_inputs(2, 64, 96)
    e = Engine(cfg, 64, 96, 1)
    with pytest.raises(DavoError, match="weights not loaded"):
        e.forward(img[:1], flow[:1], seg[:1])
    bad = dict(weights)
    bad["pose_exp_net/cnv1/weights"] = np.zeros((7, 7, 6, 16), np.float32)
    with pytest.raises(ValueError, match="does not match expected"):
        e.load_weights(bad)
    e.load_weights(weights)
    with pytest.raises(ValueError, match="batch 2 outside"):
        e.forward(img, flow, seg)
    with pytest.raises(ValueError):
        e.forward(img[:1], flow[:1, :2], seg[:1])
    e.close()
    with pytest.raises(ValueError, match="multiples of 4"):
        Engine(cfg, 62, 96, 1)
    with pytest.raises(NameError):
        DAVO("v1-sharedNN-dilatedPoseNN-se_spp_flow").setup_inference(64, 96, "davo")


def test_sequence_driver_on_gpu(tmp_path, c_oracle):
    """config-1 plumbing on the GPU: a 13-frame synthetic sequence through run_kitti_pose ->
    13-line trajectory file that matches the oracle-driven stitch."""
    from davo_amd import run_kitti_pose, sequence as S
    run_kitti_pose.main(["--synthetic", "13", "--output_dir", str(tmp_path), "--test_seq", "3",
                         "--batch_size", "4", "--img_height", "64", "--img_width", "96"])
    got = S.read_kitti_poses(str(tmp_path / "03-pred_kitti_pose.txt"))
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    infer = lambda img, flow, seg: c_oracle.forward(cfg, img, flow, seg, weights)   # noqa: E731
    want, _ = S.run_sequence(infer, S.synthetic_window_loader(64, 96), 13, 4)
    assert got.shape == (13, 4, 4)
    assert np.abs(got - np.array(want)).max() < 2e-4


def test_cli_from_files_on_disk(tmp_path, c_oracle):
    """Row f2 end to end: a dump in the reference's on-disk format (jpg strip + flownet2/seglabel npy) and an
    .npz of the weights -> CLI with the threaded loader -> trajectory == oracle on the same decoded files."""
    from davo_amd import run_kitti_pose, sequence as S, loader as L
    dump = str(tmp_path / "dump")
    L.write_synthetic_dump(dump, 9, 11, 64, 96)
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    np.savez(str(tmp_path / "w.npz"), **weights)
    run_kitti_pose.main(["--concat_img_dir", dump, "--ckpt_file", str(tmp_path / "w.npz"), "--output_dir", str(tmp_path),
                         "--test_seq", "9", "--batch_size", "4", "--img_height", "64", "--img_width", "96"])
    got = S.read_kitti_poses(str(tmp_path / "09-pred_kitti_pose.txt"))
    infer = lambda img, flow, seg: c_oracle.forward(cfg, img, flow, seg, weights)   # noqa: E731
    want, _ = S.run_sequence(infer, S.kitti_window_loader(dump, 9, 11, 64, 96).__call__, 11, 4)
    assert got.shape == (11, 4, 4)
    assert np.abs(got - np.array(want)).max() < 2e-4


def test_host_entry_chunked_copy_and_pinned_buffers():
    """davo_forward: sub-batch copy/compute overlap (host_chunk), page-locked buffers and the skipped flow
    planes 2,3 (never read: davo.py:978-982) leave the poses bit-identical."""
    from davo_amd import pinned_empty
    cfg = parse_version(FLAGSHIP_VERSION)
    B = 20                                                    # 8 + 8 + 4: ragged last sub-batch
    img, flow, seg = synth.make_inputs(B, 64, 96)
    e = _engine(cfg, 64, 96, B, synth.make_weights(cfg), "f16x3")
    e.set_option("host_chunk", 0)
    whole = e.forward(img, flow, seg).copy()
    e.set_option("host_chunk", 8)
    # default plan: a sub-batch of 8 is small enough for split-K (two partial sums per cnv5 / cnv6 output), the whole batch
    # is not: the same poses to float32 rounding, like the fused pose head's tile sums
    assert np.abs(e.forward(img, flow, seg) - whole).max() <= 1e-6 * np.abs(whole).max()
    e.set_option("split_k", 0)                                # one K chain whatever the batch: bit-identical
    e.set_option("host_chunk", 0)
    want = e.forward(img, flow, seg).copy()
    e.set_option("host_chunk", 8)
    assert np.array_equal(e.forward(img, flow, seg), want)
    pin = tuple(pinned_empty(a.shape, a.dtype) for a in (img, flow, seg))
    for d, a in zip(pin, (img, flow, seg)):
        d[...] = a
    pin[1][:, 2:] = np.nan                                    # unused planes must not matter (and are not copied)
    assert np.array_equal(e.forward(*pin), want)
    assert np.array_equal(e.forward(pin[0][:5], pin[1][:5], pin[2][:5]), want[:5])
    e.close()
    del pin


@pytest.mark.parametrize("shift", [-8, 6])
def test_activation_scale_robustness(c_oracle, shift):
    """ReLU is positively homogeneous: cnv3 (weights, bias) * 2^s and cnv4 weights * 2^-s is the same network with
    cnv3's activations scaled by 2^s.  The split-fp16 operands must hold the 1e-4 bar when a layer's activations
    are small (lo parts in the fp16 subnormal range, s=-8) or large (s=6), as a trained checkpoint may have them."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(2, 64, 96)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    w2 = dict(weights)
    k = np.float32(2.0 ** shift)
    w2["pose_exp_net/cnv3/weights"] = weights["pose_exp_net/cnv3/weights"] * k
    w2["pose_exp_net/cnv3/biases"] = weights["pose_exp_net/cnv3/biases"] * k
    w2["pose_exp_net/cnv4/weights"] = weights["pose_exp_net/cnv4/weights"] / k
    for precision in PRECISIONS:
        e = _engine(cfg, 64, 96, 2, w2, precision)
        assert_pose_close(e.forward(img, flow, seg), want, "cnv3 activations x 2^%d, %s" % (shift, precision))
        e.close()


def _rescaled(weights, shift):
    k = np.float32(2.0 ** shift)
    w2 = dict(weights)
    w2["pose_exp_net/cnv3/weights"] = weights["pose_exp_net/cnv3/weights"] * k
    w2["pose_exp_net/cnv3/biases"] = weights["pose_exp_net/cnv3/biases"] * k
    w2["pose_exp_net/cnv4/weights"] = weights["pose_exp_net/cnv4/weights"] / k
    return w2


@pytest.mark.parametrize("shift", [-22, 16])
def test_range_guard_and_calibration(c_oracle, shift):
    """A checkpoint whose cnv3 activations are ~2^-22 (fp16 pairs lose their bits) or ~2^16 (beyond the fp16
    maximum), with the automatic recovery switched off: the host entry point refuses loudly (DAVO_ERR_RANGE),
    calibrate() moves the layer's storage scale, and the poses then meet the bar like any other checkpoint."""
    from davo_amd import DavoRangeError
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(2, 64, 96)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    e = _engine(cfg, 64, 96, 2, _rescaled(weights, shift), "f16x3")
    e.set_option("auto_range", 0)                                         # the plain verdict, no recovery
    with pytest.raises(DavoRangeError, match="cnv3 activations"):
        e.forward(img, flow, seg)
    shifts = e.calibrate(img, flow, seg)
    e0 = _engine(cfg, 64, 96, 2, weights, "f16x3")
    plain = e0.calibrate(img, flow, seg)
    e0.close()
    plain["cnv3"] -= shift                                                # cnv3's storage scale absorbs the 2^shift
    assert shifts == plain, (shifts, plain)
    assert_pose_close(e.forward(img, flow, seg), want, "calibrated, cnv3 x 2^%d" % shift)
    mx, sh = e.activation_range()
    assert sh == shifts and all(256 <= mx[k] * 2.0 ** sh[k] < 2048 for k in mx), (mx, sh)
    e.set_precision("f32")                                                # the f32 mode never needed it
    assert_pose_close(e.forward(img, flow, seg), want, "f32, cnv3 x 2^%d" % shift)
    e.close()


def test_calibration_is_neutral_for_a_well_ranged_checkpoint(c_oracle):
    """Power-of-two storage scales are exact: calibrating the ordinary checkpoint moves the poses by rounding noise
    only, activations read back unscaled, and the scales can be saved and re-installed."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(3, 64, 96)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, 64, 96, 3, weights, "f16x3")
    before = e.forward(img, flow, seg).copy()
    act_before = e.debug_read("cnv4", (6, 16, 24, 128)).copy()
    shifts = e.calibrate(img, flow, seg)
    after = e.forward(img, flow, seg).copy()
    assert np.abs(after - before).max() <= 1e-6 * np.abs(before).max()       # rounding noise of the re-ranged lo halves
    assert_layer_close(e.debug_read("cnv4", (6, 16, 24, 128)), act_before, "cnv4 after calibration", rtol=1e-6)
    assert_pose_close(after, c_oracle.forward(cfg, img, flow, seg, weights), "calibrated")
    e.set_activation_shifts(None)
    assert np.array_equal(e.forward(img, flow, seg), before)
    e.set_activation_shifts(shifts)
    assert np.array_equal(e.forward(img, flow, seg), after)
    e.close()


def test_c_caller(tmp_path):
    """The C ABI stands on its own: examples/c_abi_pose.c (gcc, no Python, no HIP headers) gives the same bits as the
    Python host class over the same entry points."""
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_pose")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), "-o", exe, os.path.join(root, "examples", "c_abi_pose.c"),
                           "-L", os.path.join(root, "davo_amd"), "-ldavo_hip", "-Wl,-rpath," + os.path.join(root, "davo_amd"), "-lm"])
    cfg = parse_version(FLAGSHIP_VERSION)
    B, H, W = 3, 64, 96
    img, flow, seg = synth.make_inputs(B, H, W)
    weights = synth.make_weights(cfg)
    with open(tmp_path / "w.bin", "wb") as f:
        for name, a in weights.items():
            a = np.ascontiguousarray(a, np.float32)
            f.write(struct.pack("<I", len(name)) + name.encode() + struct.pack("<I", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<3i", B, H, W) + img.tobytes() + flow.tobytes() + seg.tobytes())
    out = subprocess.run([exe, str(tmp_path / "w.bin"), str(tmp_path / "in.bin"), str(tmp_path / "poses.bin")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = np.fromfile(tmp_path / "poses.bin", np.float32).reshape(B, 2, 6)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    assert np.array_equal(got, e.forward(img, flow, seg))
    e.close()


def test_random_shape_sweep():
    """tools/stress_shapes.py, 12 seeded random (H, W, B, cnv6 width, variant) cases: f16x3 vs the bit-exact f32 path,
    a second batch size on the same windows, every sixth case against the C oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_shapes.py"), "12", "20261003"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("H,W,B", [(16, 16, 1), (16, 20, 3), (20, 16, 2), (16, 416, 1), (128, 16, 2)])
def test_minimum_sizes(c_oracle, H, W, B):
    """The smallest frames davo_create accepts (cnv7 output 2x2 or a single row / column of tiles): every GEMM is
    smaller than one tile."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    for precision in PRECISIONS:
        e = _engine(cfg, H, W, B, weights, precision)
        assert_pose_close(e.forward(img, flow, seg), want, "%dx%d B=%d %s" % (H, W, B, precision))
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,B", [(32, 416, 4), (32, 416, 16), (64, 832, 8), (128, 416, 9)])
def test_tiles_that_skip_padding_rows_of_the_filter(c_oracle, H, W, B):
    """The 3x3 kernels do not walk the chunks of a filter row that only sees zero padding for every pixel of a tile
    (davo_tile_filter_rows).  32-row frames have 8-row maps, where dilation 8 leaves the centre row alone (one of three
    filter rows kept, every tile) and dilation 4 keeps two; 64x832 and 128x416 mix tiles that keep two and three rows, at
    batch sizes that give single-tile-shape, split-K and main + remainder plans.  Per-layer activations and poses against
    the oracle, both arithmetic modes; the shared-tap kernels' tiles against the plain ones to the bit."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W, first_window=3)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    for precision in PRECISIONS:
        e = _engine(cfg, H, W, B, weights, precision)
        got = e.forward(img, flow, seg)
        assert_pose_close(got, want, "%dx%d B=%d %s" % (H, W, B, precision))
        if precision == "f16x3":
            e.set_option("share_taps", 0)
            assert np.array_equal(e.forward(img, flow, seg), got)
        e.close()


@pytest.mark.gpu
def test_long_tiles_first_is_bit_identical():
    """"skip_order": which workgroup takes which tile never changes a tile's arithmetic.  B = 32 at 128x416 (cnv5's merged grid:
    6 of every 13 main tiles skip a filter row; float32 cnv5 / cnv6 launches): natural order, the default (float32 launches
    long-first) and everything long-first give the same poses to the bit in both modes."""
    cfg = parse_version(FLAGSHIP_VERSION)
    B = 32
    img, flow, seg = synth.make_inputs(8, 128, 416, first_window=40)
    img, flow, seg = np.tile(img, (4, 1, 1, 1)), np.tile(flow, (4, 1, 1, 1, 1)), np.tile(seg, (4, 1, 1, 1, 1))
    weights = synth.make_weights(cfg)
    for precision in PRECISIONS:
        e = _engine(cfg, 128, 416, B, weights, precision)
        got = {}
        for order in (1, 0, 2):
            e.set_option("skip_order", order)
            got[order] = e.forward(img, flow, seg)
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2]), precision
        assert np.array_equal(got[1][:8], got[1][8:16])          # the same windows in other tiles of the batch
        e.close()


# ---- BASELINE.json configurations at full size ---------------------------------------------------
def test_config2_batch32_full_size(c_oracle):
    """configs[1]: B=32, 128x416 — every window against the C oracle (multi-launch plan, remainder tiles)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    B = 32
    img, flow, seg = synth.make_inputs(B, 128, 416, first_window=100)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, 128, 416, B, weights, "f16x3")
    got = e.forward(img, flow, seg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    assert_pose_close(got, want, "B=32 f16x3")
    e.set_precision("f32")
    assert_pose_close(e.forward(img, flow, seg), want, "B=32 f32")
    e.close()


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_config3_batch128_properties(c_oracle, precision):
    """configs[2]: B=128 — 16 distinct windows tiled 8x: duplicates must produce identical poses
    wherever they sit in the batch (any tile, any launch of the plan), and a sample matches the oracle.  Both arithmetic modes
    (float32 = the reference's own, with its per-launch tile orders and filter-row ranges at this batch), and both ways of
    issuing the batch: as davo_forward's sub-batches of eight and in ONE piece (the launch plan of a resident batch of 128)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img16, flow16, seg16 = synth.make_inputs(16, 128, 416, first_window=500)
    weights = synth.make_weights(cfg)
    img, flow, seg = np.tile(img16, (8, 1, 1, 1)), np.tile(flow16, (8, 1, 1, 1, 1)), np.tile(seg16, (8, 1, 1, 1, 1))
    want = c_oracle.forward(cfg, img16[:4], flow16[:4], seg16[:4], weights)
    e = _engine(cfg, 128, 416, 128, weights, precision)
    for chunk in (8, 0):
        e.set_option("host_chunk", chunk)
        got = e.forward(img, flow, seg).reshape(8, 16, 2, 6)
        for r in range(1, 8):                                    # fused pose head: equal to float32 rounding (see above)
            assert np.abs(got[r] - got[0]).max() <= 1e-6 * np.abs(got[0]).max()
        assert_pose_close(got[0, :4], want, "B=128 sample, %s, host_chunk %d" % (precision, chunk))
        if chunk == 0:
            assert sum(m for m, _ in e.last_plan(5)) == 128 * 2 * 32 * 104 // 128       # cnv6's launches cover all 256 pair images of ONE batch
    e.close()


def test_config5_256x832(c_oracle):
    """configs[4] shape: 256x832 inputs, a small batch against the oracle in both modes."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(2, 256, 832, first_window=7)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    for prec in PRECISIONS:
        e = _engine(cfg, 256, 832, 2, weights, prec)
        assert_pose_close(e.forward(img, flow, seg), want, "256x832 %s" % prec)
        e.close()


def test_fused_pack_cnv1_variant(c_oracle):
    """davo_set_option("fuse_pack", 1): cnv1 builds its patch from the raw inputs (mask + pack fused in; the packed
    tensor never touches HBM)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    w = synth.make_weights(cfg)
    for B, H, W in ((3, 128, 416), (2, 36, 100)):
        img, flow, seg = synth.make_inputs(B, H, W)
        e = _engine(cfg, H, W, B, w, "f16x3")
        e.set_option("fuse_pack", 1)
        e.profile(1)
        got = e.forward(img, flow, seg)
        assert_pose_close(got, c_oracle.forward(cfg, img, flow, seg, w), "fuse_pack %dx%d" % (H, W))
        assert "mask_pack" not in e.profile_entries() and "cnv1" in e.profile_entries()
        e.close()


def test_two_batches_in_flight(c_oracle):
    """davo_set_inflight(2): consecutive device calls run on separate streams / workspaces; both results
    must equal the one-at-a-time results bit for bit."""
    cfg = parse_version(FLAGSHIP_VERSION)
    B = 4
    weights = synth.make_weights(cfg)
    batches = [synth.make_inputs(B, 128, 416, first_window=10 * k) for k in range(3)]
    e = _engine(cfg, 128, 416, B, weights, "f16x3")
    ref = [e.forward(*b) for b in batches]
    bufs = []
    for img, flow, seg in batches:
        bufs.append((e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48)))
    e.set_inflight(2)
    for rep in range(3):
        for b in bufs:
            e.forward_device(B, *b)
    e.synchronize()
    for k, b in enumerate(bufs):
        assert np.array_equal(b[3].download((B, 2, 6)), ref[k])
    assert_pose_close(ref[0], c_oracle.forward(cfg, *batches[0], weights), "in-flight batch 0")
    e.set_inflight(1)
    assert np.array_equal(e.forward(*batches[1]), ref[1])
    e.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_seg_label_edge_values(precision):
    """tf.cast(float->int32) truncates toward zero and tf.one_hot of an out-of-range id is a zero row
    (davo.py:1115): fractional, negative and ignore labels must mask exactly like the oracle says."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(2, 64, 96)
    # ... and labels the cast leaves to the platform (NaN, inf, beyond int32) select no class, here as in the oracle
    vals = np.array([0.0, 0.9, 18.0, 18.99, 19.0, 255.0, -0.5, -3.0, 7.5, 100.0,
                     np.nan, np.inf, -np.inf, 3e9, -3e9, -1.0, -0.999], np.float32)
    rng = np.random.RandomState(3)
    seg = vals[rng.randint(0, len(vals), size=seg.shape)].astype(np.float32)
    w = synth.make_weights(cfg)
    with np.errstate(invalid="ignore"):
        want = O.forward(cfg, img, flow, seg, w)
    e = _engine(cfg, 64, 96, 2, w, precision)
    assert_pose_close(e.forward(img, flow, seg), want, "edge labels %s" % precision)
    e.close()


def test_fused_pose_head_equals_unfused(c_oracle):
    """cnv7 epilogue with the pose head fused in (default) vs cnv7 stored + separate pose-head kernels."""
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    for B, H, W in ((5, 128, 416), (3, 36, 100), (2, 64, 96)):
        img, flow, seg = synth.make_inputs(B, H, W, first_window=3)
        e = _engine(cfg, H, W, B, weights, "f16x3")
        fused = e.forward(img, flow, seg)
        assert np.array_equal(fused, e.forward(img, flow, seg))          # reproducible
        e.set_option("fuse_pose", 0)
        unfused = e.forward(img, flow, seg)
        assert np.abs(fused - unfused).max() <= 2e-6 * np.abs(unfused).max()
        assert_pose_close(fused, c_oracle.forward(cfg, img, flow, seg, weights), "fused pose head %dx%d" % (H, W))
        e.close()


# ---- device path: range guard at synchronize --------------------------------------------------------
@pytest.mark.parametrize("shift", [-22, 16])
def test_device_path_range_guard_at_synchronize(c_oracle, shift):
    """davo_forward_device is asynchronous and cannot judge its own batch; the batches issued since the last
    davo_synchronize are judged there.  A checkpoint whose cnv3 activations sit at 2^-22 / 2^16 must come back as
    DAVO_ERR_RANGE from synchronize (and from the timed, synchronous form), never as silently clamped poses."""
    from davo_amd import DavoRangeError
    cfg = parse_version(FLAGSHIP_VERSION)
    B = 2
    img, flow, seg = synth.make_inputs(B, 64, 96)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, 64, 96, B, _rescaled(weights, shift), "f16x3")
    e.set_option("auto_range", 0)                            # the plain verdict, no recovery
    bufs = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
    e.forward_device(B, *bufs)
    with pytest.raises(DavoRangeError, match="cnv3 activations"):
        e.synchronize()
    e.synchronize()                                          # the record was consumed: nothing pending, no error
    with pytest.raises(DavoRangeError, match="cnv3 activations"):
        e.forward_device(B, *bufs, timed=True)               # synchronous form judges its own batch
    # calibrated, the same checkpoint passes on the device path and meets the bar
    e.calibrate(img, flow, seg)
    e.forward_device(B, *bufs)
    e.synchronize()
    assert_pose_close(bufs[3].download((B, 2, 6)), c_oracle.forward(cfg, img, flow, seg, weights), "device path, calibrated")
    # the float32 mode leaves no record
    e.set_activation_shifts(None)
    e.set_precision("f32")
    e.forward_device(B, *bufs)
    e.synchronize()
    assert_pose_close(bufs[3].download((B, 2, 6)), c_oracle.forward(cfg, img, flow, seg, weights), "device path, f32")
    e.close()


# ---- RCCL: the gather of the sharded driver, through librccl behind the C ABI -------------------------
def test_rccl_single_rank_communicator(tmp_path, monkeypatch):
    """An nranks=1 RCCL communicator on this GPU (one GPU = one rank; RCCL refuses two ranks on one device):
    davo_comm_unique_id -> file -> davo_comm_init, then all-gather [n,2,6] with a short (padded) shard, the
    device-buffer form, barrier and max all-reduce.  Proves librccl loads and the collectives run."""
    from davo_amd.comm import RcclComm
    monkeypatch.setenv("DAVO_COMM_DIR", str(tmp_path))
    cfg = parse_version(FLAGSHIP_VERSION)
    e = Engine(cfg, 64, 96, 2)
    comm = RcclComm(e, 0, 1)
    assert not os.path.exists(str(tmp_path / "rccl_id"))                # rank 0 removed the id after the collective init
    local = np.arange(5 * 12, dtype=np.float32).reshape(5, 2, 6)
    got, ms = comm.allgather(local, 8)                                   # 5 windows in a slot of 8
    assert got.shape == (8, 2, 6) and np.array_equal(got[:5], local) and not got[5:].any() and ms >= 0.0
    d_in, d_out = e.alloc(local.nbytes).upload(local), e.alloc(local.nbytes)
    comm.allgather_device(d_in, 5, d_out)
    assert np.array_equal(d_out.download((5, 2, 6)), local)
    comm.barrier()
    assert comm.allreduce(3.25, "max") == 3.25 and comm.allreduce(2.0, "sum") == 2.0
    with pytest.raises(Exception):
        RcclComm(e, 0, 1)                                                # one communicator per context
    comm.close()
    e.close()


def test_sharded_driver_with_rccl_world_of_one(tmp_path, monkeypatch, c_oracle):
    """run_sequence with a real RcclComm (world 1 forced through the gather) == the plain single-rank run."""
    from davo_amd import sequence as S
    from davo_amd.comm import RcclComm
    monkeypatch.setenv("DAVO_COMM_DIR", str(tmp_path))
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, 64, 96, 4, weights, "f16x3")
    comm = RcclComm(e, 0, 1)
    load = S.synthetic_window_loader(64, 96)
    local = S.run_shard(e.forward, load, 0, 7, 4)
    full, _ = comm.allgather(local, 7)
    assert np.array_equal(full, local)
    comm.close()
    e.close()


# ---- BASELINE.json configurations as tests (shapes of configs[0], [3], [4]) ---------------------------
def test_config1_seq03_shape_801_frames(tmp_path, c_oracle):
    """configs[0] plumbing at full length: 801 frames -> 799 windows, batch 1, 128x416 -> an 801-line
    03-pred_kitti_pose.txt (test_kitti_pose.py:133-153, run_inference.sh:44-51).  The file is checked window by
    window on a sample: consecutive trajectory poses differ by inv(T(tgt->src1)) of that window (:145-149), and the
    first step is T(tgt->src0) of window 0 (:143-144), both from the oracle's poses of the same synthetic windows."""
    from davo_amd import run_kitti_pose, sequence as S
    run_kitti_pose.main(["--synthetic", "801", "--output_dir", str(tmp_path), "--test_seq", "3", "--batch_size", "1"])
    lines = open(str(tmp_path / "03-pred_kitti_pose.txt")).read().splitlines()
    assert len(lines) == 801 and all(len(l.split(" ")) == 12 for l in lines)
    assert lines[0] == "1.0 0.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0 1.0 0.0"
    traj = S.read_kitti_poses(str(tmp_path / "03-pred_kitti_pose.txt"))
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    load = S.synthetic_window_loader(128, 416)
    for w in (0, 1, 137, 400, 798):
        want = c_oracle.forward(cfg, *load(w, w + 1), weights)[0]           # [2,6]
        step = np.linalg.inv(traj[w + 1]) @ traj[w + 2]
        assert np.abs(step - np.linalg.inv(S.pose_vec2mat(want[1:2], np.float64)[0])).max() < 2e-4, w
        if w == 0:
            assert np.abs(traj[1] - S.pose_vec2mat(want[0:1], np.float64)[0]).max() < 2e-4


class _CachedWindows:
    """window w -> one of `n` distinct synthetic windows (generating 4,539 distinct 128x416 windows takes minutes
    of host time and is not what these tests are about)."""

    def __init__(self, H, W, n=64, first_window=0):
        self.n = n
        self.img, self.flow, self.seg = synth.make_inputs(n, H, W, first_window=first_window)

    def index(self, w):
        return (w * 37) % self.n

    def __call__(self, s, e):
        idx = [self.index(w) for w in range(s, e)]
        return self.img[idx], self.flow[idx], self.seg[idx]


def test_config4_eight_shards_equal_one(c_oracle):
    """configs[3] shape (seq 00: 4541 frames -> 4539 windows, 8 contiguous shards of 568/563, batch 64): the eight
    shards run one after another on this GPU and concatenated equal the 1-shard run; windows either side of shard
    boundaries match the oracle.  (The 8-GPU run itself is the driver's.)"""
    from davo_amd import sequence as S
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    H, W, B, nw = 128, 416, 64, 4539
    load = _CachedWindows(H, W, 64)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    one = S.run_shard(e.forward, load, 0, nw, B)
    parts = []
    for r in range(8):
        lo, hi = S.shard_windows(nw, 8, r)
        parts.append(S.run_shard(e.forward, load, lo, hi, B))
    assert [p.shape[0] for p in parts] == [568] * 7 + [563]
    eight = np.concatenate(parts, 0)
    # same windows in other batch positions: the fused pose head sums tiles in a fixed order per image, so equal to
    # float32 rounding (tests above), not necessarily to the bit
    assert np.abs(eight - one).max() <= 1e-6 * np.abs(one).max()
    assert np.array(S.stitch_trajectory(eight)).shape == (4541, 4, 4)
    for w in (0, 567, 568, 3975, 3976, 4538):
        assert_pose_close(eight[w:w + 1], c_oracle.forward(cfg, *load(w, w + 1), weights), "window %d" % w)
    # one rank's shard in the reference's own arithmetic, batches issued whole (the float32 launch plan at batch 64 and at the
    # shard's ragged last batch of 56 -> padded to 64)
    e.set_precision("f32")
    e.set_option("host_chunk", 0)
    lo, hi = S.shard_windows(nw, 8, 3)
    shard = S.run_shard(e.forward, load, lo, hi, B)
    assert np.abs(shard - one[lo:hi]).max() <= 2e-6 * np.abs(one).max()             # the two modes agree to float32 rounding
    for w in (lo, lo + 63, lo + 64, hi - 1):
        assert_pose_close(shard[w - lo:w - lo + 1], c_oracle.forward(cfg, *load(w, w + 1), weights), "float32, window %d" % w)
    e.close()


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_config5_per_gpu_plan_256x832_batch64(c_oracle, precision):
    """configs[4] per-GPU shape: 256x832, B=64 (activations past 2^32 bytes, the full launch plan, issued in one piece): 8 distinct
    windows tiled 8x — duplicates must agree wherever they sit in the batch, and two windows match the oracle; in both modes."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img8, flow8, seg8 = synth.make_inputs(8, 256, 832, first_window=40)
    weights = synth.make_weights(cfg)
    img, flow, seg = np.tile(img8, (8, 1, 1, 1)), np.tile(flow8, (8, 1, 1, 1, 1)), np.tile(seg8, (8, 1, 1, 1, 1))
    e = _engine(cfg, 256, 832, 64, weights, precision)
    e.set_option("host_chunk", 0)
    got = e.forward(img, flow, seg).reshape(8, 8, 2, 6)
    for r in range(1, 8):
        assert np.abs(got[r] - got[0]).max() <= 1e-6 * np.abs(got[0]).max()
    assert_pose_close(got[0, :2], c_oracle.forward(cfg, img8[:2], flow8[:2], seg8[:2], weights), "256x832 B=64 sample, %s" % precision)
    e.close()


# ---- row f3 end to end: a TF V2 checkpoint bundle -> CLI -> HIP engine ----------------------------------
def test_cli_restores_a_tf_bundle(tmp_path, c_oracle):
    """test_kitti_pose.py:129-131 / run_inference.sh:28-40: the flagship weights written as a TF V2 tensor bundle
    the way a training run leaves it (two data shards, global_step and Adam slot variables beside the trainable
    ones, a `checkpoint` state file) are restored by `run_kitti_pose --ckpt_file <dir>` into the HIP engine."""
    from davo_amd import run_kitti_pose, sequence as S, loader as L, tf_checkpoint as T
    dump = str(tmp_path / "dump")
    L.write_synthetic_dump(dump, 9, 9, 64, 96)
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    extra = {"global_step": np.array(1600000, np.int64)}
    for k, v in list(weights.items())[:6]:
        extra[k + "/Adam"] = np.zeros_like(v)
        extra[k + "/Adam_1"] = np.ones_like(v)
    ck = tmp_path / "ckpt"
    ck.mkdir()
    T.write_checkpoint(str(ck / "model-1600000"), dict(weights, **extra), num_shards=2)
    assert sorted(os.listdir(str(ck))) == ["checkpoint", "model-1600000.data-00000-of-00002", "model-1600000.data-00001-of-00002",
                                           "model-1600000.index"]
    run_kitti_pose.main(["--concat_img_dir", dump, "--ckpt_file", str(ck), "--output_dir", str(tmp_path),
                         "--test_seq", "9", "--batch_size", "4", "--img_height", "64", "--img_width", "96"])
    got = S.read_kitti_poses(str(tmp_path / "09-pred_kitti_pose.txt"))
    infer = lambda img, flow, seg: c_oracle.forward(cfg, img, flow, seg, weights)   # noqa: E731
    want, _ = S.run_sequence(infer, S.kitti_window_loader(dump, 9, 9, 64, 96).__call__, 9, 4)
    assert got.shape == (9, 4, 4) and np.abs(got - np.array(want)).max() < 2e-4


# ---- the 208-pixel x 256-channel tile (conv_igemm_h3s.h) -------------------------------------------------
@pytest.mark.parametrize("B,H,W", [(1, 128, 416), (5, 128, 416), (3, 64, 96), (2, 36, 100), (1, 256, 832)])
def test_tile_208x256_forced(c_oracle, B, H, W):
    """cnv5, cnv6 and cnv7 forced onto the 208x256 tile (waves split the channels, weights as the MFMA A operand) at
    shapes where its tiles are whole image rows (128x416, 256x832) and where they straddle images and end in a ragged
    tile (64x96, 36x100): layer by layer against the 256x256-tile kernel, poses against the oracle, with the pose head
    fused into cnv7's tiles and separate."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W, first_window=3)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    e.set_option("split_k", 0)                               # tile-independent bits are a property of the single K chain
    h2, w2 = -(-H // 4), -(-W // 4)
    for fuse_pose in (0, 1):
        e.set_option("fuse_pose", fuse_pose)
        e.set_option("force_tile", 4)                        # 128x128 tiles everywhere they fit
        base = e.forward(img, flow, seg).copy()
        l5 = e.debug_read("cnv5", (2 * B, h2, w2, 256)).copy()
        l6 = e.debug_read("cnv6", (2 * B, h2, w2, 256)).copy()
        e.set_option("force_tile", 6)
        got = e.forward(img, flow, seg)
        assert [p[1] for p in e.last_plan(4)] == [6] and [p[1] for p in e.last_plan(5)] == [6]
        assert_pose_close(got, want, "208x256 tile, fuse_pose=%d" % fuse_pose)
        # the same products reach every accumulator in the same order: the activations are the other kernel's, bit for bit
        assert np.array_equal(e.debug_read("cnv5", (2 * B, h2, w2, 256)), l5)
        assert np.array_equal(e.debug_read("cnv6", (2 * B, h2, w2, 256)), l6)
        if not fuse_pose:
            assert np.array_equal(got, base)
        else:
            assert np.abs(got - base).max() <= 1e-6 * np.abs(base).max()
    e.set_option("force_tile", -1)
    assert_pose_close(e.forward(img, flow, seg), want, "planner")
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(2, 128, 416), (3, 64, 96), (2, 36, 100), (1, 256, 832)])
def test_f32_patch_kernels_match_the_implicit_gemm(c_oracle, B, H, W):
    """Float32 mode, round 4: cnv1, cnv2 and cnv3 from an LDS-staged input patch on v_mfma_f32_16x16x4_f32 (csrc/conv_patch_f32.h)
    instead of the implicit GEMM: another fixed order of the same float32 fma chain per output, so the layers agree to float32
    rounding (not to the bit), at shapes with whole tiles (128x416, 256x832) and with partial tiles in both directions (64x96:
    cnv1 32x48, cnv2/3 16x24; 36x100: 18x50, 9x25); poses against the oracle; "patch_f32" 0 restores the implicit GEMM."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W, first_window=2)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, weights, "f32")
    h1, w1, h2, w2 = -(-H // 2), -(-W // 2), -(-H // 4), -(-W // 4)
    shapes = {"cnv1": (2 * B, h1, w1, 16), "cnv2": (2 * B, h2, w2, 32), "cnv3": (2 * B, h2, w2, 64)}
    e.set_option("patch_f32", 0)
    base = e.forward(img, flow, seg).copy()
    ref = {k: e.debug_read(k, s).copy() for k, s in shapes.items()}
    assert [p[1] for p in e.last_plan(1)] != [98]
    e.set_option("patch_f32", 1)
    got = e.forward(img, flow, seg)
    assert [p[1] for p in e.last_plan(0)] == [99] and [p[1] for p in e.last_plan(1)] == [98] and [p[1] for p in e.last_plan(2)] == [97]
    for k, s in shapes.items():
        assert_layer_close(e.debug_read(k, s), ref[k], "%s: patch kernel vs implicit GEMM (float32)" % k, rtol=2e-6)
    assert np.abs(got - base).max() <= 2e-6 * np.abs(base).max()
    if H * W <= 128 * 416:
        assert_pose_close(got, c_oracle.forward(cfg, img, flow, seg, weights), "float32 patch kernels %dx%d" % (H, W))
    # the pose head in cnv7's epilogue (float32 mode, round 4; where an image has >= 128 output pixels) against the stored
    # activation + pose_head_partial / pose_finish: the same sums in another order
    e.set_option("fuse_pose", 0)
    unfused = e.forward(img, flow, seg).copy()
    c7 = e.debug_read("cnv7", (2 * B, -(-H // 8), -(-W // 8), 512))
    assert np.isfinite(c7).all() and np.abs(unfused - got).max() <= 2e-6 * np.abs(got).max()
    e.set_option("fuse_pose", 1)
    again = e.forward(img, flow, seg)
    assert np.array_equal(again, got)                            # run to run identical
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(2, 128, 416), (3, 64, 96), (1, 36, 100)])
def test_tile_208x128_forced_for_cnv4(c_oracle, B, H, W):
    """cnv4 on the four-wave 208x128 tile (round 4: conv_igemm_h3s with WAVES = 4, one wave per SIMD, three pixel ring slots):
    whole image rows at 128x416, tiles that straddle images and a ragged last tile at the small shapes.  Same products in the
    same order per accumulator as the 128x128 kernel: cnv4's activations bit for bit, poses against the oracle."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W, first_window=5)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    e.set_option("split_k", 0)
    h2, w2 = -(-H // 4), -(-W // 4)
    e.set_option("tile_208x128", 1)                          # off by default: measured 8 % behind the 128x128 tile at B = 32
    e.set_option("force_tile", 4)
    base = e.forward(img, flow, seg).copy()
    l4 = e.debug_read("cnv4", (2 * B, h2, w2, 128)).copy()
    e.set_option("force_tile", 8)                            # fits cnv4 only (128 output channels); the other layers keep their plan
    got = e.forward(img, flow, seg)
    assert [p[1] for p in e.last_plan(3)] == [8], e.last_plan(3)
    assert np.array_equal(e.debug_read("cnv4", (2 * B, h2, w2, 128)), l4)
    assert_pose_close(got, want, "cnv4 on the 208x128 tile")
    e.set_option("force_tile", -1)
    e.set_option("tile_208x128", 0)
    off = e.forward(img, flow, seg)
    assert [p[1] for p in e.last_plan(3)] != [8]
    assert_pose_close(off, want, "tile_208x128 = 0")
    e.close()


# ---- main + remainder launch of cnv5 / cnv6 as one grid (conv_igemm_h3_mainrem) ----------------------------
@pytest.mark.gpu
def test_merged_main_and_remainder_launch_is_bit_identical(c_oracle):
    """At B=32 cnv5 and cnv6 are 3.25 rounds of 256x256 tiles: a main launch of 3 whole rounds plus a remainder launch of
    128x128 tiles.  `merge_rem` (default) runs both tile shapes in ONE grid, half of the CUs taking their remainder tile
    first and the other half last, so that the halves' store bursts do not coincide.  Same tiles, same arithmetic: every
    activation and pose is bit-identical to the two-launch plan; the plan reports tile id 7 for the merged layers."""
    cfg = parse_version(FLAGSHIP_VERSION)
    B, H, W = 32, 128, 416
    img, flow, seg = synth.make_inputs(B, H, W, first_window=5)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    e.set_option("host_chunk", 0)                            # the whole batch as one step (the host entry's default chunks of 8 plan otherwise)
    e.set_option("fuse_pose", 0)
    e.set_option("wave128", 0)                               # (round 5's four-wave main tile runs as a launch of its own: test_wave128_...)
    e.set_option("merge_rem", 0)
    base = e.forward(img, flow, seg).copy()
    assert [t for _, t in e.last_plan(4)] == [5, 4] and [t for _, t in e.last_plan(5)] == [5, 4]
    a5 = e.debug_read("cnv5", (2 * B, 32, 104, 256)).copy()
    a6 = e.debug_read("cnv6", (2 * B, 32, 104, 256)).copy()
    e.set_option("merge_rem", 1)
    got = e.forward(img, flow, seg)
    assert [t for _, t in e.last_plan(4)] == [7] and [t for _, t in e.last_plan(5)] == [7]
    assert np.array_equal(e.debug_read("cnv5", (2 * B, 32, 104, 256)), a5)
    assert np.array_equal(e.debug_read("cnv6", (2 * B, 32, 104, 256)), a6)
    assert np.array_equal(got, base)
    assert_pose_close(got[:4], c_oracle.forward(cfg, img[:4], flow[:4], seg[:4], weights), "merged launch")
    # the other workgroup-id orders of the same grid, and cnv4 as a merged grid (256x128 + 128x128): the same bits
    a4 = e.debug_read("cnv4", (2 * B, 32, 104, 128)).copy()
    for key, val in (("merge_order", 1), ("merge_cnv4", 1)):
        e.set_option(key, val)
        assert np.array_equal(e.forward(img, flow, seg), base), key
        assert np.array_equal(e.debug_read("cnv4", (2 * B, 32, 104, 128)), a4), key
        assert np.array_equal(e.debug_read("cnv5", (2 * B, 32, 104, 256)), a5) and np.array_equal(e.debug_read("cnv6", (2 * B, 32, 104, 256)), a6), key
    assert [t for _, t in e.last_plan(3)] == [7]
    e.close()


# ---- repeatability under load: the hand-counted waits of the f16x3 kernels ---------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(32, 128, 416), (16, 128, 416), (7, 52, 172)])
def test_repeated_forwards_are_bit_identical(B, H, W):
    """LDS-DMA rings, counted vmcnt / lgkmcnt waits, double-buffered patches with one barrier per tile: a miscounted wait is
    a race that one parity run can pass.  The same two batches 120 times with the queue kept deep, then 60 times checked
    one by one: every result equals the first, bit for bit (`tools/soak.py` is the long form: 4,800 forwards, 6 shapes)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    sets, refs = [], []
    for k in range(2):
        img, flow, seg = synth.make_inputs(B, H, W, first_window=53 * k + 3)
        d = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
        e.forward_device(B, *d)
        e.synchronize()
        sets.append(d)
        refs.append(d[3].download((B, 2, 6)).copy())
    for i in range(120):
        e.forward_device(B, *sets[i % 2])
    e.synchronize()
    assert np.array_equal(sets[1][3].download((B, 2, 6)), refs[1])
    for i in range(60):
        e.forward_device(B, *sets[i % 2])
        assert np.array_equal(sets[i % 2][3].download((B, 2, 6)), refs[i % 2]), i
    e.close()


# ---- cnv2 / cnv3 from LDS-staged input patches (conv_patch_cnv2_h3, conv_patch_cnv3_h3) -------------------
@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(2, 128, 416), (3, 64, 96), (2, 36, 100), (1, 52, 172), (5, 20, 48), (1, 256, 832)])
def test_patch_kernels_match_implicit_gemm(c_oracle, B, H, W):
    """cnv2 and cnv3 read their taps from a double-buffered LDS patch of an 8x8 output tile (weights in registers, one
    barrier per tile) instead of gathering 13 / 9 chunks per tile.  Same products, another summation order: the
    activations agree with the implicit-GEMM kernels' to float32 rounding and the poses with the oracle, at whole tiles
    (128x416, 256x832) and at maps that leave partial tiles in both directions (9x25, 13x43, 5x12 pixels)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W, first_window=23)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    h2, w2 = -(-H // 4), -(-W // 4)
    e.set_option("patch_cnv2", 0)
    e.set_option("patch_cnv3", 0)
    base = e.forward(img, flow, seg).copy()
    assert e.last_plan(1)[0][1] < 90 and e.last_plan(2)[0][1] < 90          # tile ids of the implicit-GEMM kernel
    a2 = e.debug_read("cnv2", (2 * B, h2, w2, 32)).copy()
    a3 = e.debug_read("cnv3", (2 * B, h2, w2, 64)).copy()
    for k2, k3 in ((1, 0), (0, 1), (1, 1)):
        e.set_option("patch_cnv2", k2)
        e.set_option("patch_cnv3", k3)
        got = e.forward(img, flow, seg)
        assert (e.last_plan(1)[0][1] == 98) == bool(k2) and (e.last_plan(2)[0][1] == 97) == bool(k3)
        b2 = e.debug_read("cnv2", (2 * B, h2, w2, 32))
        b3 = e.debug_read("cnv3", (2 * B, h2, w2, 64))
        assert np.abs(b2 - a2).max() <= 2e-6 * np.abs(a2).max(), (k2, k3)
        assert np.abs(b3 - a3).max() <= 2e-6 * np.abs(a3).max(), (k2, k3)
        if not k2:
            assert np.array_equal(b2, a2)
        assert np.abs(got - base).max() <= 2e-6 * np.abs(base).max()
        assert_pose_close(got, want, "patch kernels %d%d" % (k2, k3))
    e.close()


# ---- shared-tap staging (conv_igemm_h3 RATE > 0) ---------------------------------------------------------
@pytest.mark.parametrize("B,H,W", [(1, 128, 416), (5, 128, 416), (3, 64, 96), (2, 36, 100), (2, 20, 48), (1, 256, 832)])
def test_shared_tap_staging_is_bit_identical(c_oracle, B, H, W):
    """cnv3..cnv6 stage ONE pixel patch per filter row for its three kx taps (a third fewer DMA instructions and L2->LDS
    bytes); lanes whose shifted tap leaves the image row read a zero row.  Same products, same order: every activation
    and the poses are bit-identical to the per-tap staging, at whole-row tiles (128x416, 256x832), at tiles straddling
    image rows and images (64x96, 36x100), and where the map is narrower than the dilation (20x48: falls back per layer)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W, first_window=11)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    e.set_option("fuse_pose", 0)
    h2, w2 = -(-H // 4), -(-W // 4)
    shapes = {"cnv3": 64, "cnv4": 128, "cnv5": 256, "cnv6": 256}
    for tile in (-1, 4, 5):                                 # planner (main + remainder launches), 128x128, 256x256
        e.set_option("force_tile", tile)
        e.set_option("share_taps", 0)
        base = e.forward(img, flow, seg).copy()
        acts = {k: e.debug_read(k, (2 * B, h2, w2, c)).copy() for k, c in shapes.items()}
        e.set_option("share_taps", 1)
        got = e.forward(img, flow, seg)
        for k, c in shapes.items():
            assert np.array_equal(e.debug_read(k, (2 * B, h2, w2, c)), acts[k]), (k, tile)
        assert np.array_equal(got, base)
    assert_pose_close(got, c_oracle.forward(cfg, img, flow, seg, weights), "shared taps")
    e.close()


# ---- f16x3 at the reference's call surface: never an error on a finite float32 network (davo.py:1553-1569) ----------
def _scale_channels(weights, producer, consumers, idx, shift):
    """ReLU homogeneity per channel: output channels `idx` of `producer` (weights + bias) x 2^shift and the matching input
    channels of every consumer x 2^-shift is the same network with those activation channels scaled by 2^shift."""
    k = np.float32(2.0 ** shift)
    w2 = dict(weights)
    pw = weights["pose_exp_net/%s/weights" % producer].copy()
    pb = weights["pose_exp_net/%s/biases" % producer].copy()
    pw[..., idx] *= k
    pb[idx] *= k
    w2["pose_exp_net/%s/weights" % producer], w2["pose_exp_net/%s/biases" % producer] = pw, pb
    for cons in consumers:
        cw = weights["pose_exp_net/%s/weights" % cons].copy()
        cw[:, :, idx, :] /= k
        w2["pose_exp_net/%s/weights" % cons] = cw
    return w2


@pytest.mark.parametrize("shift", [-22, 16])
def test_auto_range_host_path_never_raises(c_oracle, shift):
    """The checkpoints of test_range_guard_and_calibration through the DEFAULT engine: the first batch leaves the
    fp16-pair range, the library re-calibrates on it and re-issues it; the caller sees float32-grade poses, no error."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(2, 64, 96)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    e = _engine(cfg, 64, 96, 2, _rescaled(weights, shift), "f16x3")
    assert_pose_close(e.forward(img, flow, seg), want, "auto range, cnv3 x 2^%d" % shift)
    st = e.range_stats()
    assert st == {"recalibrations": 1, "f32_batches": 0, "reissued": 1}, st
    assert_pose_close(e.forward(img, flow, seg), want, "second batch, new scales")
    assert e.range_stats() == st                                          # the new scales hold: nothing re-issued
    e.close()
    d = DAVO(version=FLAGSHIP_VERSION)                                    # and through the reference's call surface
    d.load_weights(_rescaled(weights, shift))
    d.setup_inference(64, 96, "davo", 3, 2, img, None, flow, None, seg)
    assert_pose_close(d.inference(None, "pose")["pose"], want, "DAVO.inference, cnv3 x 2^%d" % shift)
    d.engine.close()


def test_auto_range_device_path_reissues_at_synchronize(c_oracle):
    """Asynchronous batches on two buffer sets, three in flight before the verdict: synchronize() re-issues them and
    returns with every pose buffer float32-grade."""
    cfg = parse_version(FLAGSHIP_VERSION)
    B = 2
    weights = synth.make_weights(cfg)
    e = _engine(cfg, 64, 96, B, _rescaled(weights, 16), "f16x3")
    sets, wants = [], []
    for k in range(2):
        img, flow, seg = synth.make_inputs(B, 64, 96, first_window=5 * k)
        sets.append((e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48)))
        wants.append(c_oracle.forward(cfg, img, flow, seg, weights))
    for k in (0, 1, 0):
        e.forward_device(B, *sets[k])
    e.synchronize()
    for k in range(2):
        assert_pose_close(sets[k][3].download((B, 2, 6)), wants[k], "device path, set %d" % k)
    st = e.range_stats()
    assert st["reissued"] == 3 and st["recalibrations"] >= 1 and st["f32_batches"] == 0, st     # every issued batch once, on its own record
    assert "re-calibrated" in e.range_report() and "cnv3" in e.range_report(), e.range_report()
    for k in (0, 1):
        e.forward_device(B, *sets[k])
    e.synchronize()
    assert e.range_stats() == st
    assert e.forward_device(B, *sets[0], timed=True) > 0.0
    e.close()


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 128, 416)])
def test_streaming_caller_recycles_its_input_buffers_before_the_verdict(c_oracle, B, H, W):
    """A double-buffered H2D loop (the shape of a streaming caller): ONE set of input buffers, overwritten with the next
    batch's data - stream-ordered, behind the forward that read them - long before davo_synchronize.  With a checkpoint that
    trips the range guard every batch is re-issued at its verdict: from the context's own copy of what was issued, so every
    batch's poses are oracle-grade although the caller's buffers hold later data by then (round 3 re-issued from the caller's
    pointers: VERDICT r3 item 4).  Eleven batches: more than the ring of eight, so slots are judged and reused while issuing."""
    cfg = parse_version(FLAGSHIP_VERSION)
    n = 11                     # 64x96: the pose head is two launches and the guard's copy a launch of its own; 128x416: both ride in pose_from_tiles
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, _rescaled(weights, 16), "f16x3")
    data = [synth.make_inputs(B, H, W, first_window=3 * k) for k in range(n)]
    wants = [c_oracle.forward(cfg, *d, weights) for d in data]
    d_img, d_flow, d_seg = e.alloc(data[0][0].nbytes), e.alloc(data[0][1].nbytes), e.alloc(data[0][2].nbytes)
    poses = [e.alloc(B * 48) for _ in range(n)]
    for k in range(n):
        d_img.upload(data[k][0]); d_flow.upload(data[k][1]); d_seg.upload(data[k][2])      # on the context's stream, behind batch k - 1
        e.forward_device(B, d_img, d_flow, d_seg, poses[k])
    poison = np.full_like(data[0][0], 255)
    d_img.upload(poison)                                                   # and the buffers do not even hold the last batch any more
    e.synchronize()
    for k in range(n):
        assert_pose_close(poses[k].download((B, 2, 6)), wants[k], "recycled inputs, batch %d" % k)
    st = e.range_stats()
    # batches 0..7 went out on the old scales; issuing batch 8 needed batch 0's ring slot, so batch 0 was judged and the scales
    # re-calibrated there: batches 8..10 were in range from the start
    assert st["reissued"] == 8 and st["recalibrations"] >= 1 and st["f32_batches"] == 0, st
    # once the scales hold, the same loop re-issues nothing
    for k in range(n):
        d_img.upload(data[k][0]); d_flow.upload(data[k][1]); d_seg.upload(data[k][2])
        e.forward_device(B, d_img, d_flow, d_seg, poses[k])
    e.synchronize()
    assert e.range_stats() == st
    for k in range(n):
        assert_pose_close(poses[k].download((B, 2, 6)), wants[k], "recycled inputs, scales settled, batch %d" % k)
    e.close()


def test_range_tickets_with_two_batches_in_flight(c_oracle):
    """davo_set_inflight(2): consecutive batches run on two streams and two workspaces, each with a ticket of its own (its
    record, its copy of the inputs if the verdict fails, its stream to poll).  A checkpoint that trips the guard, ten batches on
    buffers of their own (with batches in flight on two streams the API gives a caller no point at which a buffer may be
    recycled short of davo_synchronize): every batch's poses come out oracle-grade, through the ring of eight."""
    cfg = parse_version(FLAGSHIP_VERSION)
    B, n = 2, 10
    weights = synth.make_weights(cfg)
    e = _engine(cfg, 64, 96, B, _rescaled(weights, 16), "f16x3")
    data = [synth.make_inputs(B, 64, 96, first_window=2 * k) for k in range(n)]
    wants = [c_oracle.forward(cfg, *d, weights) for d in data]
    sets = [tuple(e.alloc(a.nbytes).upload(a) for a in d) + (e.alloc(B * 48),) for d in data]
    e.set_inflight(2)
    for k in range(n):
        e.forward_device(B, *sets[k])
    e.synchronize()
    for k in range(n):
        assert_pose_close(sets[k][3].download((B, 2, 6)), wants[k], "two in flight, batch %d" % k)
    st = e.range_stats()
    assert st["reissued"] == 8 and st["recalibrations"] >= 1 and st["f32_batches"] == 0, st      # batches 8, 9 went out on the new scales
    for k in range(n):                                                  # settled: nothing is re-issued, same poses
        e.forward_device(B, *sets[k])
    e.synchronize()
    assert e.range_stats() == st
    for k in range(n):
        assert_pose_close(sets[k][3].download((B, 2, 6)), wants[k], "two in flight, settled, batch %d" % k)
    e.set_inflight(1)
    e.close()


def test_stable_inputs_reissue_from_the_callers_buffers(c_oracle):
    """"stable_inputs" 1: the caller promises unchanged inputs until the verdict, the library takes no copies and a re-issue
    reads the caller's buffers (the round-3 behaviour, now opt-in)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    B = 2
    weights = synth.make_weights(cfg)
    e = _engine(cfg, 64, 96, B, _rescaled(weights, 16), "f16x3")
    e.set_option("stable_inputs", 1)
    sets, wants = [], []
    for k in range(3):
        img, flow, seg = synth.make_inputs(B, 64, 96, first_window=4 * k)
        sets.append((e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48)))
        wants.append(c_oracle.forward(cfg, img, flow, seg, weights))
    for k in (0, 1, 2, 0, 1, 2):
        e.forward_device(B, *sets[k])
    e.synchronize()
    for k in range(3):
        assert_pose_close(sets[k][3].download((B, 2, 6)), wants[k], "stable inputs, set %d" % k)
    assert e.range_stats()["reissued"] == 6                # all six went out before the first verdict (ring of eight)
    e.close()


def test_auto_range_falls_back_to_the_f32_kernels(c_oracle):
    """cnv3 activations at 2^-80: no storage scale in [-60, 60] brings them into the fp16-pair range, so the batch runs
    on the library's own float32 kernels (never the oracle) and still meets the bar."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(2, 64, 96)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    e = _engine(cfg, 64, 96, 2, _rescaled(weights, -80), "f16x3")
    assert_pose_close(e.forward(img, flow, seg), want, "f32 fallback")
    st = e.range_stats()
    assert st["f32_batches"] == 1 and st["reissued"] == 1, st
    e.close()


@pytest.mark.parametrize("H,W,B", [(64, 96, 2), (128, 416, 1)])
def test_per_channel_dynamic_range(c_oracle, H, W, B):
    """Wide dynamic range INSIDE a layer: every other output channel of cnv3 and of cnv5 carries 2^-10 of its
    neighbours' magnitude (the next layer's matching input-channel weights x 2^10: the same network).  One power-of-two
    scale per layer cannot lift those channels, so their hi/lo pairs sit 10 bits lower in the fp16 range; the bar must
    hold with and without calibration (the bound)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    w2 = _scale_channels(weights, "cnv3", ["cnv4"], np.arange(0, 64, 2), -10)
    w2 = _scale_channels(w2, "cnv5", ["pose/rotation/cnv6", "pose/translation/cnv6"], np.arange(1, 256, 2), -10)
    assert_pose_close(c_oracle.forward(cfg, img, flow, seg, w2), want, "oracle: the rescaled net is the same net")
    for precision in PRECISIONS:
        e = _engine(cfg, H, W, B, w2, precision)
        assert_pose_close(e.forward(img, flow, seg), want, "per-channel 2^-10, %s" % precision)
        if precision == "f16x3":
            e.calibrate(img, flow, seg)
            assert_pose_close(e.forward(img, flow, seg), want, "per-channel 2^-10, calibrated")
            assert e.range_stats()["reissued"] == 0
        e.close()


def test_heavy_tailed_weights(c_oracle):
    """A few weights per layer 100x larger than the rest (the layer's power-of-two weight scale is set by them, the
    ordinary weights' lo halves sit 7 bits lower): another network, so the oracle runs on the same weights."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(2, 64, 96)
    weights = dict(synth.make_weights(cfg))
    rng = np.random.RandomState(7)
    for name in list(weights):
        if name.endswith("/weights") and "/pred/" not in name:
            w = weights[name].copy()
            flat = w.reshape(-1)
            idx = rng.choice(flat.size, 6, replace=False)
            flat[idx] *= 100.0
            weights[name] = w
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    assert np.isfinite(want).all() and np.abs(want).max() > 1e-2
    for precision in PRECISIONS:
        e = _engine(cfg, 64, 96, 2, weights, precision)
        assert_pose_close(e.forward(img, flow, seg), want, "heavy-tailed weights, %s" % precision)
        e.close()


def test_later_windows_louder_than_the_calibration_window(c_oracle):
    """A sequence through DAVO.inference (iterator inputs, davo.py:1553-1569) whose windows 3.. carry 2^8 larger flow
    than the window the scales were calibrated on: the loud batch is re-calibrated and re-issued inside the call, the
    windows after it run on the new scales, every pose meets the bar."""
    cfg = parse_version(FLAGSHIP_VERSION)
    H, W, B, nb = 64, 96, 1, 6
    weights = synth.make_weights(cfg)
    batches = []
    for i in range(nb):
        img, flow, seg = synth.make_inputs(B, H, W, first_window=i)
        if i >= 3:
            flow = (flow * np.float32(256.0)).astype(np.float32)
        batches.append((img, flow, seg))
    d = DAVO(version=FLAGSHIP_VERSION)
    d.load_weights(weights)
    d.setup_inference(H, W, "davo", 3, B, iter(batches))
    d.calibrate(batches[0])
    for i in range(nb):
        got = d.inference(None, "pose")["pose"]
        assert_pose_close(got, c_oracle.forward(cfg, *batches[i], weights), "window %d" % i)
    st = d.engine.range_stats()
    assert st["recalibrations"] == 1 and st["reissued"] == 1 and st["f32_batches"] == 0, st
    d.engine.close()


@pytest.mark.parametrize("B", [11, 32])
def test_profile_counts_one_launch_per_forward(B):
    """Every profiled layer records exactly one event pair per forward (per issued launch), also at a batch where the
    merged cnv5 / cnv6 grid does not apply and the layer falls back to main + remainder launches (B = 11)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    H, W = 128, 416
    img, flow, seg = synth.make_inputs(1, H, W)
    img, flow, seg = (np.repeat(a, B, axis=0) for a in (img, flow, seg))
    e = _engine(cfg, H, W, B, synth.make_weights(cfg), "f16x3")
    bufs = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
    e.forward_device(B, *bufs)
    e.synchronize()
    n = 4
    e.profile(1)
    e.profile_reset()
    for _ in range(n):
        e.forward_device(B, *bufs)
    ent = e.profile_entries()
    for layer in ("cnv5", "cnv6"):
        launches = len(e.last_plan({"cnv5": 4, "cnv6": 5}[layer]))
        assert ent[layer][0] == n and ent[layer][1] > 0.0, (layer, ent)
        assert (layer + ".rem" in ent) == (launches == 2), (layer, launches, sorted(ent))
        if launches == 2:
            assert ent[layer + ".rem"][0] == n
    e.profile(2)
    e.set_option("profile_stride", 2)
    e.profile_reset()
    for _ in range(n):
        e.forward_device(B, *bufs)
    ent = e.profile_entries()
    assert ent["cnv6"][0] == n // 2 and ent["cnv6"][1] > 0.0, ent
    e.profile(False)
    e.close()


def test_bench_multi_gpu_code_path_on_one_gpu(tmp_path):
    """bench.py's N > 1 path (RCCL communicator, barrier-bracketed timed region, max all-reduce over ranks, pose
    all-gather) executed at world size 1 through the rank launcher: `--force-comm`.  The driver's 8-GPU run then is not
    the first execution of that code."""
    import json
    import sys
    from davo_amd.launch import spawn_ranks
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "bench.out"
    with open(out, "wb") as f:
        rc = spawn_ranks([os.path.join(root, "bench.py"), "--gpus", "1", "--force-comm", "--steps", "3", "--warmup", "1", "--settle", "2",
                          "--batch", "4", "--no-cpu-baseline", "--no-f32", "--no-pipelined"], 1, stdout=f, timeout=600)
    assert rc == 0
    res = json.loads(open(out).read().strip().splitlines()[-1])
    g = res["gather"]
    assert g and g["ranks"] == 1 and g["forced_at_world_1"] and g["bytes_per_rank"] == 4 * 48 and g["collective_ms"] >= 0.0
    assert res["n_gpus"] == 1 and res["steps"] == 3 and res["value"] > 0 and res["timing"]["settle_steps"] == 2
    assert res["timing"]["dominant_launch_ms"]["n"] == 3 and res["timing"]["step_ms_on_stream"]["n"] == 2
    assert res["max_abs_err_vs_oracle"] <= 1e-4 * res["max_abs_ref"]
    assert res["config"]["workload"].startswith("none of BASELINE.json's configs")


@pytest.mark.parametrize("B,H,W", [(32, 128, 416), (5, 128, 416), (3, 64, 96), (1, 128, 416)])
def test_folded_tails_are_bit_identical(B, H, W):
    """"fold_tails": the excitation MLP in the squeeze launch's last workgroup (per triplet) and the pose head's tile sum in
    the cnv7 launch's last workgroup give the bits of the separate se_excite / pose_from_tiles launches — whichever
    workgroup happens to be last, launch after launch, and with two batches in flight."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(min(B, 4), H, W)
    reps = -(-B // img.shape[0])
    img, flow, seg = (np.concatenate([a] * reps)[:B] for a in (img, flow, seg))
    e = _engine(cfg, H, W, B, synth.make_weights(cfg), "f16x3")
    sets = [(e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
            for _ in range(2)]
    e.set_option("fold_tails", 0)
    e.forward_device(B, *sets[0])
    e.synchronize()
    want = sets[0][3].download((B, 2, 6))
    tab = e.debug_read("att_table", (B, 3, 19)).copy()
    assert np.isfinite(want).all() and np.abs(want).max() > 1e-3
    e.set_option("fold_tails", 1)
    for _ in range(3):
        sets[0][3].upload(np.zeros((B, 2, 6), np.float32))
        e.forward_device(B, *sets[0])
        e.synchronize()
        assert np.array_equal(sets[0][3].download((B, 2, 6)), want)
        assert np.array_equal(e.debug_read("att_table", (B, 3, 19)), tab)
    e.set_inflight(2)
    for i in range(8):
        e.forward_device(B, *sets[i % 2])
    e.synchronize()
    for st in sets:
        assert np.array_equal(st[3].download((B, 2, 6)), want)
    e.close()


@pytest.mark.parametrize("B,H,W", [(1, 128, 416), (2, 128, 416), (4, 128, 416), (1, 64, 96), (3, 36, 100), (1, 256, 832)])
def test_deep_ring_is_bit_identical(c_oracle, B, H, W):
    """"deep_ring": launches of at most one workgroup per CU (small batches) run the same tiles on LDS rings of 3..6 slots
    instead of 2 (more chunks of LDS-DMA in flight): same products in the same order, so every activation and pose is
    bit-identical — also with every tile shape forced in turn."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    for tile in (-1, 0, 1, 2, 3, 4):
        e.set_option("force_tile", tile)
        e.set_option("deep_ring", 0)
        want = e.forward(img, flow, seg).copy()
        acts = {k: e.debug_read(k, (2 * B, (H + 3) // 4, (W + 3) // 4, ch)).copy() for k, ch in (("cnv4", 128), ("cnv5", 256), ("cnv6", 256))}
        e.set_option("deep_ring", 1)
        got = e.forward(img, flow, seg)
        assert np.array_equal(got, want), ("tile", tile)
        for k, ch in (("cnv4", 128), ("cnv5", 256), ("cnv6", 256)):
            assert np.array_equal(e.debug_read(k, (2 * B, (H + 3) // 4, (W + 3) // 4, ch)), acts[k]), (k, "tile", tile)
    if H * W <= 128 * 416:
        assert_pose_close(got, c_oracle.forward(cfg, img, flow, seg, weights), "deep ring %dx%d B=%d" % (H, W, B))
    e.close()


@pytest.mark.parametrize("B,tile", [(4, 0), (8, 0), (3, -1), (1, -1), (5, 4)])
def test_pose_is_identical_launch_after_launch(B, tile, c_oracle):
    """Regression: with 128x32 tiles on cnv7 (what the planner picks at batch 1) and three or more workgroups per CU
    (batch >= 3), the fused pose head's sums over tiles that straddle two images came out wrong in one forward in three, by up
    to 7e-3.  Cause (round 4): compiler-formed `v_pk_fma_f32 ... op_sel:[0,1,0]` - the low result lane takes
    the high register of src1 - sporadically reads the selected operand as 0 in lanes 48-63 under that occupancy; the library
    holds no packed float32 instruction (tools/check_isa.py).  Two hundred forwards of one batch must agree to the bit, and the
    first of them with the oracle (every forward of the flaking builds could be wrong, the first included)."""
    cfg = parse_version(FLAGSHIP_VERSION)
    H, W = 128, 416
    img, flow, seg = synth.make_inputs(B, H, W)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    e.set_option("force_tile", tile)
    bufs = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
    ref, bad = None, 0
    for _ in range(200):
        e.forward_device(B, *bufs)
        e.synchronize()
        pose = bufs[3].download((B, 2, 6))
        if ref is None:
            ref = pose
        elif not np.array_equal(pose, ref):
            bad += 1
    assert bad == 0, "%d of 199 forwards differ from the first" % bad
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    err, scale = float(np.abs(ref - want).max()), float(np.abs(want).max())
    assert err <= 1e-4 and err <= 1e-4 * scale, "first forward vs oracle: %.3g (max|ref| %.3g)" % (err, scale)
    e.close()


def test_config4_shape_from_files_one_rank_with_forced_gather(tmp_path, c_oracle):
    """BASELINE configs[3] plumbing on one GPU: a dump on disk -> worker processes decoding into shared page-locked batch
    buffers (the default loader of the CLI) -> batch 64 -> the RCCL all-gather at world size 1 (--force_comm) -> stitch ->
    trajectory file == the oracle on the same decoded files; the run's time split is reported."""
    import json
    from davo_amd import run_kitti_pose, sequence as S, loader as L
    dump = str(tmp_path / "dump")
    n_frames = 150
    L.write_synthetic_dump(dump, 0, n_frames, 64, 96)
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    np.savez(str(tmp_path / "w.npz"), **weights)
    report = str(tmp_path / "report.json")
    run_kitti_pose.main(["--concat_img_dir", dump, "--ckpt_file", str(tmp_path / "w.npz"), "--output_dir", str(tmp_path),
                         "--test_seq", "0", "--batch_size", "64", "--img_height", "64", "--img_width", "96", "--loader_procs", "3",
                         "--force_comm", "--report", report])
    got = S.read_kitti_poses(str(tmp_path / "00-pred_kitti_pose.txt"))
    infer = lambda img, flow, seg: c_oracle.forward(cfg, img, flow, seg, weights)   # noqa: E731
    want, _ = S.run_sequence(infer, S.kitti_window_loader(dump, 0, n_frames, 64, 96).__call__, n_frames, 64)
    assert got.shape == (n_frames, 4, 4)
    assert np.abs(got - np.array(want)).max() < 2e-3                      # a 148-step chain of float32-grade poses
    r = json.load(open(report))
    assert r["windows"] == n_frames - 2 and r["world"] == 1 and r["batch_size"] == 64
    for k in ("load_wait_s", "forward_s", "gather_s", "stitch_s", "write_s", "total_s"):
        assert r[k] >= 0.0
    assert r["forward_s"] > 0 and r["total_s"] >= r["forward_s"] and r["range_recovery"]["f32_batches"] == 0
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("psm_")]


@pytest.mark.parametrize("B,H,W", [(1, 128, 416), (2, 64, 96), (1, 256, 832), (1, 36, 100)])
def test_split_k_small_batches(c_oracle, B, H, W):
    """"split_k": cnv5 / cnv6 launches of at most half a workgroup per CU run the two halves of their input channels as
    two groups into float32 partial sums, a fix-up kernel adds them (ReLU, fp16 pairs, range record).  Against the single
    chain: the same activations to float32 rounding; against the oracle: the bar; launch after launch: the same bits."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    shp = (2 * B, (H + 3) // 4, (W + 3) // 4, 256)
    e.set_option("split_k", 0)
    one = e.forward(img, flow, seg).copy()
    a5, a6 = e.debug_read("cnv5", shp).copy(), e.debug_read("cnv6", shp).copy()
    e.set_option("split_k", 1)
    two = e.forward(img, flow, seg).copy()
    assert_layer_close(e.debug_read("cnv5", shp), a5, "cnv5 split-K vs single chain", rtol=2e-6)
    assert_layer_close(e.debug_read("cnv6", shp), a6, "cnv6 split-K vs single chain", rtol=2e-6)
    assert np.abs(two - one).max() <= 2e-6 * np.abs(one).max()
    for _ in range(3):
        assert np.array_equal(e.forward(img, flow, seg), two)
    mx, _ = e.activation_range()
    assert mx["cnv5"] > 0 and mx["cnv6"] > 0                               # the fix-up kernel keeps the range record
    if H * W <= 128 * 416:
        assert_pose_close(two, c_oracle.forward(cfg, img, flow, seg, weights), "split-K %dx%d B=%d" % (H, W, B))
    e.close()


def test_folded_split_k_fixup_is_bit_identical(c_oracle):
    """"fold_fixup": at batch 1 (128x416: 104 tiles x 4 parts per layer, an x extent that is a multiple of 8) the part of a tile that
    finishes last adds the tile's partial sums itself - found in its own XCD's L2 - instead of a splitk_fixup launch.  Same
    additions in the same order: cnv5, cnv6 and the poses carry the same bits as with the launch, forward after forward, with
    three batches in flight (the slots' partial-sum regions and ticket counters are their own), and the range record is kept."""
    cfg = parse_version(FLAGSHIP_VERSION)
    H, W = 128, 416
    weights = synth.make_weights(cfg)
    data = [synth.make_inputs(1, H, W, first_window=5 * k) for k in range(4)]
    e = _engine(cfg, H, W, 1, weights, "f16x3")
    shp = (2, 32, 104, 256)
    e.set_option("fold_fixup", 0)
    want, acts = [], []
    for d in data:
        want.append(e.forward(*d).copy())
        acts.append((e.debug_read("cnv5", shp).copy(), e.debug_read("cnv6", shp).copy()))
    e.profile(1); e.profile_reset()
    e.forward(*data[0])
    assert any("cnv5" == k or "cnv6" == k for k in e.profile_entries())
    e.profile(0)
    e.set_option("fold_fixup", 1)                 # (off by default: measured slower than the launch it saves, davo_hip.h)
    e.activation_range(reset=True)
    for rep in range(10):
        for k, d in enumerate(data):
            got = e.forward(*d)
            assert np.array_equal(got, want[k]), (rep, k, np.abs(got - want[k]).max())
            if rep == 0:
                assert np.array_equal(e.debug_read("cnv5", shp), acts[k][0]) and np.array_equal(e.debug_read("cnv6", shp), acts[k][1])
    mx, _ = e.activation_range()
    assert mx["cnv5"] > 0 and mx["cnv6"] > 0                               # the folded tail keeps the range record
    assert_pose_close(want[0], c_oracle.forward(cfg, *data[0], weights), "batch 1, folded fix-up")
    e.set_inflight(3)
    outs = [np.empty((1, 2, 6), np.float32) for _ in range(40)]
    for i, o in enumerate(outs):
        e.submit(*data[i % 4], o)
    e.synchronize()
    for i, o in enumerate(outs):
        assert np.array_equal(o, want[i % 4]), i
    e.close()


def test_per_channel_spread_beyond_the_pair_format_runs_in_float32(c_oracle):
    """Half of cnv3's and cnv5's channels at 2^-22 of their neighbours (consumer weights x 2^22): per-layer storage scales
    cannot keep those channels' fp16 pairs float32-grade (measured 9e-4 against the 1e-4 bar).  The spread shows in the
    consumer's per-input-channel weight norms, so the library sees it when the weights are packed and runs this network
    on its float32 kernels — the caller gets the reference's result, as with any float32 network (davo.py:1553-1569);
    with auto_range off it refuses instead."""
    from davo_amd import DavoRangeError
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(2, 64, 96)
    weights = synth.make_weights(cfg)
    want = c_oracle.forward(cfg, img, flow, seg, weights)
    w2 = _scale_channels(weights, "cnv3", ["cnv4"], np.arange(0, 64, 2), -22)
    w2 = _scale_channels(w2, "cnv5", ["pose/rotation/cnv6", "pose/translation/cnv6"], np.arange(1, 256, 2), -22)
    e = _engine(cfg, 64, 96, 2, w2, "f16x3")
    assert_pose_close(e.forward(img, flow, seg), want, "per-channel 2^-22 (float32 kernels)")
    assert e.range_stats()["f32_batches"] == 1                                        # once per call, not per host sub-batch
    assert "float32 kernels" in e.range_report() and "weights" in e.range_report(), e.range_report()
    e.set_option("auto_range", 0)
    with pytest.raises(DavoRangeError, match="per-input-channel weight norms"):
        e.forward(img, flow, seg)
    e.close()
    # 2^-12 stays on the f16x3 kernels and inside the bar
    w3 = _scale_channels(weights, "cnv3", ["cnv4"], np.arange(0, 64, 2), -12)
    e = _engine(cfg, 64, 96, 2, w3, "f16x3")
    assert_pose_close(e.forward(img, flow, seg), want, "per-channel 2^-12 (f16x3)")
    assert e.range_stats()["f32_batches"] == 0
    e.close()
    # a dead input channel (tiny weights in the consumer, nothing else changed) is harmless and must NOT cost the fast path:
    # the guard measures the spread upwards from the lower quartile of the channel norms, not from their minimum
    w4 = {k: v.copy() for k, v in weights.items()}
    w4["pose_exp_net/cnv4/weights"][:, :, 5, :] *= 2.0 ** -20
    w4["pose_exp_net/pose/rotation/cnv6/weights"][:, :, 7:11, :] *= 2.0 ** -24
    e = _engine(cfg, 64, 96, 2, w4, "f16x3")
    assert_pose_close(e.forward(img, flow, seg), c_oracle.forward(cfg, img, flow, seg, w4), "dead input channels (f16x3)")
    assert e.range_stats()["f32_batches"] == 0 and e.range_report() == ""
    e.close()


# ---- 256x256 tiles on four waves of 128x128 (conv_igemm_h3w.h, option "wave128") ---------------------------------
@pytest.mark.parametrize("B,H,W", [(1, 128, 416), (3, 128, 416), (32, 128, 416), (2, 64, 96), (1, 256, 832), (2, 36, 100)])
def test_wave128_tile_is_bit_identical(c_oracle, B, H, W):
    """cnv5 / cnv6 on conv_igemm_h3w (one wave per SIMD, 128x128 outputs each, next chunk's fragments requested under the last
    column group): same staging, same products in the same order per accumulator as conv_igemm_h3's 256x256 tile - activations and
    poses bit-identical, as a single launch (force_tile 5) and as the main launch of the planner's split; 36x100 has no whole
    256-row tiles per launch and must fall back."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(B, H, W, first_window=5)
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    e.set_option("fuse_pose", 0)
    e.set_option("host_chunk", 0)                            # B = 32 as one step: the plan with a remainder (and, without wave128, the merged grid)
    h2, w2 = -(-H // 4), -(-W // 4)
    shapes = {"cnv5": 256, "cnv6": 256}
    for tile, merge in ((5, 0), (-1, 0), (-1, 1)):          # one launch; main + remainder launches; the two as one grid (where the plan has one)
        e.set_option("force_tile", tile)
        e.set_option("merge_rem", merge)
        e.set_option("wave128", 0)
        base = e.forward(img, flow, seg).copy()
        acts = {k: e.debug_read(k, (2 * B, h2, w2, c)).copy() for k, c in shapes.items()}
        for w in (1, 2):                                     # 2: the remainder rows on conv_igemm_h3w64's 256x64 tiles as well
            e.set_option("wave128", w)
            got = e.forward(img, flow, seg)
            for k, a in acts.items():
                assert np.array_equal(e.debug_read(k, a.shape), a), (k, tile, merge, w)
            assert np.array_equal(got, base)
    assert_pose_close(got, c_oracle.forward(cfg, img, flow, seg, weights), "wave128")
    e.close()


@pytest.mark.parametrize("B,H,W,opt", [(48, 128, 416, 2), (8, 128, 416, 3), (32, 128, 416, 3), (2, 256, 832, 3)])
def test_wave128_cnv4_on_256x128_tiles_is_bit_identical(c_oracle, B, H, W, opt):
    """cnv4 on conv_igemm_h3w128 (256 x 128 tiles, four waves of 128 x 64, shared patch, five-slot weight ring four chunks ahead, per-tap
    counted waits): the default where its tiles fill whole rounds of the CUs or nearly so (B = 48: 4.875 rounds), "wave128" 3 forces it
    elsewhere.  Same products in the same order as conv_igemm_h3's tiles: activation and poses bit for bit."""
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(min(B, 8), H, W, first_window=4)
    reps = -(-B // img.shape[0])
    img, flow, seg = (np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:B] for a in (img, flow, seg))
    weights = synth.make_weights(cfg)
    e = _engine(cfg, H, W, B, weights, "f16x3")
    e.set_option("host_chunk", 0)
    e.set_option("wave128", 0)
    base = e.forward(img, flow, seg).copy()
    a4 = e.debug_read("cnv4", (2 * B, H // 4, W // 4, 128)).copy()
    e.set_option("wave128", opt)
    got = e.forward(img, flow, seg).copy()
    if 2 * B * (H // 4) * (W // 4) >= 256 * 256:           # at least one 256-row tile per CU (the kernel's own condition; below it: fallback)
        assert [t for _, t in e.last_plan(3)] == [2], e.last_plan(3)
    assert np.array_equal(e.debug_read("cnv4", a4.shape), a4)
    assert np.array_equal(got, base)
    assert_pose_close(got[:4], c_oracle.forward(cfg, img[:4], flow[:4], seg[:4], weights), "cnv4 on 256x128 tiles")
    e.close()

