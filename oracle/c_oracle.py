"""ctypes wrapper of oracle/libdavo_oracle.so (CPU ORACLE — test infrastructure only;
see the header of oracle/davo_oracle.c).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DAVO_ORACLE_SO: another build of the same file (the sanitizer build of tests/test_oracle.py::test_c_oracle_asan)
_SO = os.environ.get("DAVO_ORACLE_SO") or os.path.join(_HERE, "libdavo_oracle.so")

WEIGHT_ORDER = (
    ["pose_exp_net/%s/%s" % (l, k) for l in ("cnv1", "cnv2", "cnv3", "cnv4", "cnv5")
     for k in ("weights", "biases")]
    + ["pose_exp_net/pose/%s/%s/%s" % (h, l, k) for h in ("rotation", "translation")
       for l in ("cnv6", "cnv7", "pred") for k in ("weights", "biases")]
    + ["pose_exp_net/se_flow/bottleneck_fc/kernel", "pose_exp_net/se_flow/bottleneck_fc/bias",
       "pose_exp_net/se_flow/recover_fc/kernel", "pose_exp_net/se_flow/recover_fc/bias",
       "pose_exp_net/pose_exp_net/seg_channel_weight/weight"])


class _Variant(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("cin_per_frame", "cnv6_out", "se_act", "norm_flow",
                                            "abs_mode", "att_source", "mask_rgb", "mask_info")]


def build(force=False):
    src = os.path.join(_HERE, "davo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libdavo_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_forward.restype = ctypes.c_int
        _lib.oracle_max_threads.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _weight_table(weights):
    keep, ptrs = [], (ctypes.POINTER(ctypes.c_float) * len(WEIGHT_ORDER))()
    for i, name in enumerate(WEIGHT_ORDER):
        if name in weights:
            a = np.ascontiguousarray(weights[name], np.float32)
            keep.append(a)
            ptrs[i] = _fp(a)
    return keep, ptrs


def max_threads():
    return lib().oracle_max_threads()


def conv2d_same(x, w, b, stride, rate, relu=True):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    N, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    Ho, Wo = -(-H // stride), -(-W // stride)
    y = np.empty((N, Ho, Wo, Cout), np.float32)
    lib().oracle_conv2d_same(_fp(x), N, H, W, Cin, _fp(w), kh, kw, Cout, _fp(b),
                             stride, rate, int(relu), _fp(y))
    return y


def pack_inputs(cfg, img, flow, seg, weights):
    img = np.ascontiguousarray(img, np.uint8); flow = np.ascontiguousarray(flow, np.float32)
    seg = np.ascontiguousarray(seg, np.float32)
    B, H, W3, _ = img.shape
    W = W3 // 3
    keep, ptrs = _weight_table(weights)
    v = _Variant(*cfg.as_c_ints())
    out = np.empty((B, 2, H, W, 2 * cfg.cin_per_frame), np.float32)
    lib().oracle_pack(ctypes.byref(v), B, H, W, img.ctypes.data_as(ctypes.c_void_p), _fp(flow),
                      _fp(seg), ptrs, _fp(out))
    return out


def forward(cfg, img, flow, seg, weights, nthreads=0):
    """float32 [B,2,6] poses; nthreads=0 -> OpenMP default (all cores)."""
    img = np.ascontiguousarray(img, np.uint8); flow = np.ascontiguousarray(flow, np.float32)
    seg = np.ascontiguousarray(seg, np.float32)
    B, H, W3, _ = img.shape
    W = W3 // 3
    assert flow.shape == (B, 4, H, W, 2) and seg.shape == (B, 3, H, W, 1)
    keep, ptrs = _weight_table(weights)
    v = _Variant(*cfg.as_c_ints())
    out = np.empty((B, 2, 6), np.float32)
    rc = lib().oracle_forward(ctypes.byref(v), B, H, W, img.ctypes.data_as(ctypes.c_void_p),
                              _fp(flow), _fp(seg), ptrs, _fp(out), int(nthreads))
    if rc != 0:
        raise MemoryError("oracle_forward failed (%d)" % rc)
    return out
