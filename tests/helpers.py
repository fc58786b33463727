import json
import os

import numpy as np

from davo_amd import synth
from davo_amd.version import parse_version

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose_golden.json")

# The bar: max|pose - oracle| <= 1e-4 AND <= 1e-4 * max|oracle| on [B,2,6].
ABS_TOL = 1e-4
REL_TOL = 1e-4


def load_golden():
    with open(GOLDEN) as f:
        return json.load(f)


def case_inputs(case):
    cfg = parse_version(case["version"])
    img, flow, seg = synth.make_inputs(case["B"], case["H"], case["W"])
    flow = (flow * np.float32(case["flow_scale"])).astype(np.float32)
    return cfg, img, flow, seg, synth.make_weights(cfg)


def assert_pose_close(got, want, what=""):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    err = np.abs(got - want).max()
    scale = np.abs(want).max()
    assert err <= ABS_TOL, "%s max abs err %.3g > %.1g" % (what, err, ABS_TOL)
    assert err <= REL_TOL * scale, "%s max abs err %.3g > %.1g * max|ref| (%.3g)" % (what, err, REL_TOL, scale)
    return err


def assert_layer_close(got, want, what="", rtol=2e-5):
    """activations: max abs err relative to the tensor's max magnitude."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want).max()
    scale = max(np.abs(want).max(), 1e-30)
    assert err <= rtol * scale, "%s: max abs err %.3g vs scale %.3g (rel %.3g > %.1g)" % (what, err, scale, err / scale, rtol)
    return err / scale


def checksum_matches(arr, ck, rtol=1e-5):
    a = np.asarray(arr, np.float64).ravel()
    assert list(np.asarray(arr).shape) == ck["shape"]
    scale = max(ck["l2"] / np.sqrt(a.size), 1e-30)
    assert abs(np.sqrt((a * a).sum()) - ck["l2"]) <= rtol * max(ck["l2"], 1e-30)
    assert abs(a.mean() - ck["mean"]) <= rtol * scale
    for i, v in ck["samples"]:
        assert abs(a[i] - v) <= 10 * rtol * max(abs(v), scale), (i, a[i], v)
