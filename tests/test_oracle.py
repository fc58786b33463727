"""The oracle itself: numpy-f64 vs C-f32 vs torch-CPU conv2d, TF SAME-padding facts, and the
committed golden outputs.  (Parity with TF1 is unpinned — see oracle/davo_oracle.py.)"""
import os

import numpy as np
import pytest

from davo_amd import synth
from davo_amd.version import parse_version, FLAGSHIP_VERSION
from oracle import davo_oracle as O

from helpers import load_golden, case_inputs, assert_pose_close, assert_layer_close, checksum_matches


def test_same_padding_note_p():
    # SURVEY note P: cnv1 (2,3), cnv2 (1,2), cnv7 (0,1); dilated stride-1 layers pad `rate` each side
    assert O.same_pad(128, 7, 2, 1) == (64, 2, 3) and O.same_pad(416, 7, 2, 1) == (208, 2, 3)
    assert O.same_pad(64, 5, 2, 1) == (32, 1, 2) and O.same_pad(208, 5, 2, 1) == (104, 1, 2)
    assert O.same_pad(32, 3, 2, 1) == (16, 0, 1) and O.same_pad(104, 3, 2, 1) == (52, 0, 1)
    for r in (2, 4, 8):
        assert O.same_pad(32, 3, 1, r) == (32, r, r)
    assert O.same_pad(256, 7, 2, 1) == (128, 2, 3)          # identical at 256x832
    assert O.same_pad(5, 3, 2, 1) == (3, 1, 1)              # odd size: symmetric


def test_flop_count_table_l():
    # SURVEY table L: 3,890,085,888 MAC per pair at 128x416
    layers = [(7, 10, 16, 64 * 208), (5, 16, 32, 32 * 104), (3, 32, 64, 32 * 104), (3, 64, 128, 32 * 104),
              (3, 128, 256, 32 * 104)] + 2 * [(3, 256, 128, 32 * 104), (3, 128, 256, 16 * 52), (1, 256, 3, 16 * 52)]
    assert sum(k * k * ci * co * px for k, ci, co, px in layers) == 3890085888


@pytest.mark.parametrize("k,stride,rate,cin,cout,H,W", [
    (7, 2, 1, 10, 16, 20, 28), (5, 2, 1, 16, 32, 17, 23), (3, 1, 2, 32, 64, 12, 20),
    (3, 1, 8, 8, 16, 12, 20), (3, 2, 1, 16, 8, 13, 15), (1, 1, 1, 32, 3, 5, 7)])
def test_conv_three_way(c_oracle, k, stride, rate, cin, cout, H, W):
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F
    rng = np.random.RandomState(k * 100 + stride * 10 + rate)
    x = rng.randn(2, H, W, cin).astype(np.float32)
    w = (rng.randn(k, k, cin, cout) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.randn(cout).astype(np.float32)
    y64 = O.conv2d_same(x.astype(np.float64), w, b, stride, rate)
    y32 = c_oracle.conv2d_same(x, w, b, stride, rate)
    # independent third opinion: torch conv2d with explicit asymmetric TF padding
    Ho, pt, pb = O.same_pad(H, k, stride, rate)
    Wo, pl, pr = O.same_pad(W, k, stride, rate)
    xt = F.pad(torch.from_numpy(x).double().permute(0, 3, 1, 2), (pl, pr, pt, pb))
    yt = F.conv2d(xt, torch.from_numpy(w).double().permute(3, 2, 0, 1), torch.from_numpy(b).double(),
                  stride=stride, dilation=rate)
    yt = torch.relu(yt).permute(0, 2, 3, 1).numpy()
    assert yt.shape == y64.shape == y32.shape == (2, Ho, Wo, cout)
    assert np.abs(yt - y64).max() < 1e-12
    assert_layer_close(y32, y64, "C vs numpy", rtol=5e-6)


def test_attention_lut_semantics():
    # one_hot(int32(seg)): truncation toward zero, out-of-range ids (255, -3, 19) -> 0
    tab = np.arange(1, 20, dtype=np.float64)[None] / 100.0
    seg = np.array([0.0, 0.9, 18.0, 18.99, 19.0, 255.0, -0.5, -3.0], np.float32).reshape(1, 1, 8, 1)
    att = O.attention_map(tab, seg)[0, 0, :, 0]
    assert np.allclose(att, [0.01, 0.01, 0.19, 0.19, 0.0, 0.0, 0.01, 0.0])
    # labels the float -> int32 cast leaves to the platform: pinned to "no class"
    odd = np.array([np.nan, np.inf, -np.inf, 3e9, -3e9, -1.0, -0.999, 18.999], np.float32).reshape(1, 1, 8, 1)
    with np.errstate(invalid="ignore"):
        att = O.attention_map(tab, odd)[0, 0, :, 0]
    assert np.allclose(att, [0, 0, 0, 0, 0, 0, 0.01, 0.19])


def test_pack_layout_flagship(c_oracle):
    cfg = parse_version(FLAGSHIP_VERSION)
    img, flow, seg = synth.make_inputs(1, 64, 96)
    w = synth.make_weights(cfg)
    p = O.pack_inputs(cfg, img, flow, seg, w)
    assert p.shape == (1, 2, 64, 96, 10)
    assert np.all(p[..., 3:5] == 0)                          # tgt "flow" is zeros (davo.py:979)
    tgt = img[:, :, 96:192].astype(np.float64) / 255 * 2 - 1   # tgt is NOT masked for se_flow (davo.py:1411)
    assert np.allclose(p[:, 0, ..., 0:3], tgt) and np.allclose(p[:, 1, ..., 0:3], tgt)
    ign = seg[:, 0, ..., 0] == 255                           # ignore label zeroes src0 rgb+flow
    assert ign.any() and np.all(p[:, 0][ign][:, 5:] == 0)
    assert_layer_close(c_oracle.pack_inputs(cfg, img, flow, seg, w), p, "pack C vs numpy", rtol=1e-6)


def test_golden_vectors(c_oracle):
    g = load_golden()
    for name, case in g["cases"].items():
        if case["H"] * case["W"] > 128 * 416:
            continue                                         # the 256x832 case is covered on the GPU box
        cfg, img, flow, seg, weights = case_inputs(case)
        for k in ("img", "flow", "seg"):                     # the seeded inputs regenerate bit-for-bit
            checksum_matches({"img": img, "flow": flow, "seg": seg}[k], case["inputs"][k], rtol=1e-12)
        want = np.array(case["pose"])
        assert_pose_close(c_oracle.forward(cfg, img, flow, seg, weights), want, name + " C-f32")
        if case["H"] * case["W"] <= 64 * 96:
            keep = {}
            assert_pose_close(O.forward(cfg, img, flow, seg, weights, np.float64, keep), want, name + " numpy")
            for lname, ck in case["layers"].items():
                checksum_matches(keep[lname], ck)
        assert np.abs(want).max() > 0.02, "outputs too small for the absolute bar to mean anything"


def test_numpy_f32_mode_close_to_f64():
    g = load_golden()["cases"]["flagship_b3_32x64_smallflow"]
    cfg, img, flow, seg, weights = case_inputs(g)
    assert_pose_close(O.forward(cfg, img, flow, seg, weights, np.float32), np.array(g["pose"]), "numpy f32")


def test_c_oracle_thread_count_invariant(c_oracle):
    g = load_golden()["cases"]["flagship_b3_32x64_smallflow"]
    cfg, img, flow, seg, weights = case_inputs(g)
    a = c_oracle.forward(cfg, img, flow, seg, weights, nthreads=1)
    b = c_oracle.forward(cfg, img, flow, seg, weights, nthreads=4)
    assert np.array_equal(a, b)


def test_whole_posenet_vs_torch():
    """Third opinion on the whole PoseNN (not just one conv): torch-CPU float64 conv2d with explicit TF
    SAME padding, both heads, pred + mean + 0.01 — against oracle.posenet on the oracle's packed input."""
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F
    g = load_golden()["cases"]["flagship_b3_36x100_ragged"]
    cfg, img, flow, seg, w = case_inputs(g)
    packed = O.pack_inputs(cfg, img, flow, seg, w)
    B, _, H, W, C = packed.shape
    x = torch.from_numpy(packed.reshape(B * 2, H, W, C)).permute(0, 3, 1, 2)

    def conv(x, name, stride, rate, relu=True):
        wt = torch.from_numpy(w[name + "/weights"]).double().permute(3, 2, 0, 1)
        k = wt.shape[-1]
        _, pt, pb = O.same_pad(x.shape[2], k, stride, rate)
        _, pl, pr = O.same_pad(x.shape[3], k, stride, rate)
        y = F.conv2d(F.pad(x, (pl, pr, pt, pb)), wt, torch.from_numpy(w[name + "/biases"]).double(), stride=stride, dilation=rate)
        return torch.relu(y) if relu else y
    h = x
    for name, stride, rate in (("cnv1", 2, 1), ("cnv2", 2, 1), ("cnv3", 1, 2), ("cnv4", 1, 4), ("cnv5", 1, 8)):
        h = conv(h, "pose_exp_net/" + name, stride, rate)
    outs = []
    for head in ("rotation", "translation"):
        p = "pose_exp_net/pose/%s/" % head
        c7 = conv(conv(h, p + "cnv6", 1, 2), p + "cnv7", 2, 1)
        outs.append(conv(c7, p + "pred", 1, 1, relu=False).mean(dim=(2, 3)))
    pose = (0.01 * torch.cat(outs, dim=1)).numpy().reshape(B, 2, 6)
    assert np.abs(pose - np.array(g["pose"])).max() < 1e-12


def test_c_oracle_asan(tmp_path):
    """oracle/Makefile's sanitizer build (-fsanitize=address,undefined) of the C restatement on the shapes that
    stress its indexing: minimum frame, ragged sizes with asymmetric SAME padding, ignore labels, both variants'
    channel counts.  The checker itself must be memory-clean (sanitizers on the CPU build only)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "libdavo_oracle_asan.so"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from davo_amd import synth, parse_version, FLAGSHIP_VERSION\n"
        "from oracle import c_oracle, davo_oracle as O\n"
        "for version, (B, H, W) in ((FLAGSHIP_VERSION, (1, 16, 16)), (FLAGSHIP_VERSION, (2, 20, 36)),\n"
        "                           ('v1-decay100k-sharedNN-dilatedPoseNN-cnv6_64-no_segmask', (1, 16, 20))):\n"
        "    cfg = parse_version(version); w = synth.make_weights(cfg)\n"
        "    img, flow, seg = synth.make_inputs(B, H, W)\n"
        "    seg[0, :, :4, :4] = 255.0; seg[0, 0, 5, 5] = -3.0; seg[0, 2, 6, 6] = np.nan\n"
        "    got = c_oracle.forward(cfg, img, flow, seg, w, nthreads=2)\n"
        "    want = O.forward(cfg, img, flow, seg, w)\n"
        "    assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-3), version\n"
        "x = np.random.RandomState(3).randn(2, 9, 11, 4).astype(np.float32)\n"
        "k = np.random.RandomState(4).randn(3, 3, 4, 5).astype(np.float32)\n"
        "y = c_oracle.conv2d_same(x, k, np.zeros(5, np.float32), 2, 3, True); assert y.shape == (2, 5, 6, 5)\n"
        "print('asan ok')\n" % root)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               DAVO_ORACLE_SO=os.path.join(root, "oracle", "libdavo_oracle_asan.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "asan ok" in out.stdout, (out.stdout + out.stderr)[-3000:]
