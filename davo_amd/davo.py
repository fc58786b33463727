"""Host side of the pose path: the reference's ``DAVO`` call surface over libdavo_hip.so.

Mirrors ``DAVO.__init__`` (reference davo.py:31-33), ``setup_inference`` (davo.py:1533-1551)
and ``inference`` (davo.py:1553-1569).  The TF reference takes tensors of a tf.data iterator
and pulls the next batch inside ``sess.run``; here the inputs are numpy arrays, or an iterator
yielding ``(img_u8, flow, seg)`` batches, pulled once per ``inference`` call.  Weights enter
through ``load_weights`` — the stand-in for ``tf.train.Saver(...).restore``
(test_kitti_pose.py:129-131) — keyed by the TF variable names.
"""
import ctypes

import numpy as np

from . import _lib
from .version import parse_version, weight_shapes


class DavoError(RuntimeError):
    pass


class DavoRangeError(DavoError):
    """f16x3: a layer's activations left the fp16-pair storage range (include/davo_hip.h: DAVO_ERR_RANGE).
    Only raised with ``set_option("auto_range", 0)``: by default the library re-issues such a batch itself
    (re-calibrated, or on its float32 kernels), as the reference's float32 graph never fails on a finite network."""


_PY_ERR = {-1: ValueError, -2: DavoError, -3: DavoError, -4: MemoryError, -5: DavoRangeError}


class _PinnedBlock:
    """hipHostMalloc'd block; freed when the last numpy view of it goes away."""

    def __init__(self, device, nbytes):
        p = ctypes.c_void_p()
        if _lib.lib().davo_host_alloc(int(device), int(nbytes), ctypes.byref(p)) != 0:
            raise DavoError("davo_host_alloc(%d bytes) failed on device %d" % (nbytes, device))
        self.ptr, self.nbytes = p, int(nbytes)

    def __del__(self):
        if getattr(self, "ptr", None):
            _lib.lib().davo_host_free(self.ptr)
            self.ptr = None


def pinned_empty(shape, dtype, device=0):
    """numpy array over page-locked host memory (include/davo_hip.h: davo_host_alloc): buffers filled by the
    loader and handed to DAVO.inference / Engine.forward copy at the PCIe DMA rate."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    blk = _PinnedBlock(device, max(n, 1))
    buf = (ctypes.c_uint8 * blk.nbytes).from_address(blk.ptr.value)
    buf._owner = blk                                   # the ctypes object is the array's base; it keeps the block alive
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def pin_array(arr, device=0):
    """Page-lock the memory of a C-contiguous numpy array in place (include/davo_hip.h: davo_host_register)."""
    if not arr.flags.c_contiguous:
        raise ValueError("only a C-contiguous array can be page-locked in place")
    if _lib.lib().davo_host_register(int(device), ctypes.c_void_p(arr.ctypes.data), arr.nbytes) != 0:
        raise DavoError("davo_host_register(%d bytes) failed on device %d" % (arr.nbytes, device))


def unpin_array(arr):
    if _lib.lib().davo_host_unregister(ctypes.c_void_p(arr.ctypes.data)) != 0:
        raise DavoError("davo_host_unregister failed")


class DeviceBuffer:
    """A hipMalloc'd buffer owned through a context (bench / multi-GPU shards keep inputs in HBM)."""

    def __init__(self, engine, nbytes):
        self.engine, self.nbytes = engine, int(nbytes)
        p = ctypes.c_void_p()
        engine._check(_lib.lib().davo_device_malloc(engine._ctx, self.nbytes, ctypes.byref(p)))
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.engine._check(_lib.lib().davo_memcpy_h2d(self.engine._ctx, self.ptr, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))
        return self

    def download(self, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.engine._check(_lib.lib().davo_memcpy_d2h(self.engine._ctx, out.ctypes.data_as(ctypes.c_void_p), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            _lib.lib().davo_device_free(self.engine._ctx, self.ptr)
            self.ptr = None


class Engine:
    """Thin object wrapper of one ``davo_ctx`` (one GPU, one host thread)."""

    def __init__(self, cfg, img_height, img_width, max_batch, device=0):
        self.cfg, self.H, self.W, self.max_batch, self.device = cfg, img_height, img_width, max_batch, device
        self._L = _lib.lib()
        self._ctx = ctypes.c_void_p()
        v = _lib.DavoVariant(*cfg.as_c_ints())
        rc = self._L.davo_create(ctypes.byref(self._ctx), device, img_height, img_width, max_batch, ctypes.byref(v))
        if rc != 0:
            msg = self._L.davo_last_error(self._ctx).decode() if self._ctx else "davo_create failed"
            if self._ctx:
                self._L.davo_destroy(self._ctx)
                self._ctx = ctypes.c_void_p()
            raise _PY_ERR.get(rc, DavoError)(msg)

    def _check(self, rc):
        if rc != 0:
            raise _PY_ERR.get(rc, DavoError)(self._L.davo_last_error(self._ctx).decode())

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.davo_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    __del__ = close

    def load_weights(self, weights):
        want = weight_shapes(self.cfg)
        for name in want:
            if name not in weights:
                raise KeyError("checkpoint has no variable `%s'" % name)
        for name in want:
            a = np.ascontiguousarray(weights[name], np.float32)
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            self._check(self._L.davo_load_weight(self._ctx, name.encode(), a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), shape, a.ndim))

    def forward(self, img, flow, seg):
        img = np.ascontiguousarray(img, np.uint8)
        flow = np.ascontiguousarray(flow, np.float32)
        seg = np.ascontiguousarray(seg, np.float32)
        B = img.shape[0]
        if img.shape != (B, self.H, 3 * self.W, 3):
            raise ValueError("img shape %s != %s" % (img.shape, (B, self.H, 3 * self.W, 3)))
        if flow.shape != (B, 4, self.H, self.W, 2):
            raise ValueError("flow shape %s != %s" % (flow.shape, (B, 4, self.H, self.W, 2)))
        if seg.shape != (B, 3, self.H, self.W, 1):
            raise ValueError("seg shape %s != %s" % (seg.shape, (B, 3, self.H, self.W, 1)))
        out = np.empty((B, 2, 6), np.float32)
        vp = ctypes.c_void_p
        self._check(self._L.davo_forward(self._ctx, B, img.ctypes.data_as(vp), flow.ctypes.data_as(vp),
                                         seg.ctypes.data_as(vp), out.ctypes.data_as(vp)))
        return out

    def _check_batch(self, img, flow, seg):
        B = img.shape[0]
        if img.shape != (B, self.H, 3 * self.W, 3):
            raise ValueError("img shape %s != %s" % (img.shape, (B, self.H, 3 * self.W, 3)))
        if flow.shape != (B, 4, self.H, self.W, 2):
            raise ValueError("flow shape %s != %s" % (flow.shape, (B, 4, self.H, self.W, 2)))
        if seg.shape != (B, 3, self.H, self.W, 1):
            raise ValueError("seg shape %s != %s" % (seg.shape, (B, 3, self.H, self.W, 1)))
        return B

    def submit(self, img, flow, seg, out, hold=0):
        """Streaming form of forward (include/davo_hip.h: davo_submit): issue the batch and return; ``out`` - a C-contiguous
        float32 [B,2,6] array the caller keeps alive - receives the poses when the batch is delivered (a later submit, wait()
        or synchronize()).  On return the input arrays of the batch submitted ``hold`` calls ago may be overwritten."""
        if not (isinstance(img, np.ndarray) and img.dtype == np.uint8 and img.flags.c_contiguous and
                isinstance(flow, np.ndarray) and flow.dtype == np.float32 and flow.flags.c_contiguous and
                isinstance(seg, np.ndarray) and seg.dtype == np.float32 and seg.flags.c_contiguous):
            if hold:
                raise ValueError("submit(hold > 0) needs C-contiguous uint8 / float32 / float32 arrays (a converted copy would not outlive the call)")
            img = np.ascontiguousarray(img, np.uint8)
            flow = np.ascontiguousarray(flow, np.float32)
            seg = np.ascontiguousarray(seg, np.float32)
        B = self._check_batch(img, flow, seg)
        if not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.flags.c_contiguous and out.shape == (B, 2, 6)):
            raise ValueError("out must be a C-contiguous float32 array of shape (%d, 2, 6)" % B)
        vp = ctypes.c_void_p
        self._check(self._L.davo_submit(self._ctx, B, img.ctypes.data_as(vp), flow.ctypes.data_as(vp), seg.ctypes.data_as(vp),
                                        out.ctypes.data_as(vp), int(hold)))

    def wait(self, leave_pending=0):
        """Deliver submitted batches until at most ``leave_pending`` are outstanding (include/davo_hip.h: davo_wait)."""
        self._check(self._L.davo_wait(self._ctx, int(leave_pending)))

    def pending(self):
        return self._L.davo_pending(self._ctx)

    LAYERS = ("cnv1", "cnv2", "cnv3", "cnv4", "cnv5", "cnv6")

    def calibrate(self, img, flow, seg):
        """Choose the power-of-two storage scales of the f16x3 activations from a sample batch
        (include/davo_hip.h: davo_calibrate).  -> {layer: log2 scale}."""
        img = np.ascontiguousarray(img, np.uint8)
        flow = np.ascontiguousarray(flow, np.float32)
        seg = np.ascontiguousarray(seg, np.float32)
        B = img.shape[0]
        if img.shape != (B, self.H, 3 * self.W, 3) or flow.shape != (B, 4, self.H, self.W, 2) or seg.shape != (B, 3, self.H, self.W, 1):
            raise ValueError("calibration batch shapes %s %s %s do not match the engine" % (img.shape, flow.shape, seg.shape))
        bufs = [self.alloc(a.nbytes).upload(a) for a in (img, flow, seg)]
        shifts = (ctypes.c_int * 6)()
        try:
            self._check(self._L.davo_calibrate(self._ctx, B, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, shifts))
        finally:
            for b in bufs:
                b.free()
        return dict(zip(self.LAYERS, list(shifts)))

    def activation_range(self, reset=False):
        """-> ({layer: largest |activation| stored since the last reset}, {layer: log2 storage scale})."""
        mx, sh = (ctypes.c_float * 6)(), (ctypes.c_int * 6)()
        self._check(self._L.davo_activation_range(self._ctx, mx, sh, int(bool(reset))))
        return dict(zip(self.LAYERS, list(mx))), dict(zip(self.LAYERS, list(sh)))

    def range_stats(self):
        """{'recalibrations', 'f32_batches', 'reissued'}: what the f16x3 range recovery has done so far
        (include/davo_hip.h: davo_range_stats)."""
        a, b, c = ctypes.c_longlong(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
        self._check(self._L.davo_range_stats(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"recalibrations": a.value, "f32_batches": b.value, "reissued": c.value}

    def range_report(self):
        """What the range management last did, in words ('' if nothing): the verdict behind a re-calibration or a float32
        batch, or the weight tensor whose per-input-channel spread keeps the network on the float32 kernels."""
        return (self._L.davo_range_report(self._ctx) or b"").decode()

    def set_activation_shifts(self, shifts=None):
        """Install storage scales from an earlier calibrate() (dict or sequence of 6 ints; None = none)."""
        if shifts is None:
            self._check(self._L.davo_set_activation_shifts(self._ctx, None))
            return
        vals = [shifts[k] for k in self.LAYERS] if isinstance(shifts, dict) else list(shifts)
        self._check(self._L.davo_set_activation_shifts(self._ctx, (ctypes.c_int * 6)(*vals)))

    def forward_device(self, B, d_img, d_flow, d_seg, d_pose, timed=False):
        ms = ctypes.c_float(0.0)
        self._check(self._L.davo_forward_device(self._ctx, B, d_img.ptr, d_flow.ptr, d_seg.ptr, d_pose.ptr,
                                                ctypes.byref(ms) if timed else None))
        return ms.value if timed else None

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def synchronize(self):
        self._check(self._L.davo_synchronize(self._ctx))

    def set_option(self, key, value):
        """'auto_range' (default 1), 'fuse_pose' (default 1), 'fuse_pack' (default 0), 'host_chunk' (default 8), ...:
        see include/davo_hip.h."""
        self._check(self._L.davo_set_option(self._ctx, key.encode(), int(value)))

    def set_inflight(self, n):
        """Batches kept in flight by forward_device (1..4): n streams + n workspaces, rotated per call."""
        self._check(self._L.davo_set_inflight(self._ctx, int(n)))

    def set_impl(self, impl):
        self._check(self._L.davo_set_impl(self._ctx, {"mfma": 0, "direct": 1}.get(impl, impl)))

    def set_precision(self, precision):
        """'f16x3' (default: split-fp16 MFMA, float32-grade) or 'f32' (FP32 MFMA, bit-exact fmaf chains)."""
        self._check(self._L.davo_set_precision(self._ctx, {"f32": 0, "f16x3": 1}.get(precision, precision)))

    def profile(self, on):
        """0/False off, 1/True every kernel, 2 only the dominant kernel (main cnv6 launch)."""
        self._check(self._L.davo_profile_enable(self._ctx, int(on)))

    def profile_reset(self):
        self._check(self._L.davo_profile_reset(self._ctx))

    def profile_entries(self):
        out, i = {}, 0
        name = ctypes.create_string_buffer(64)
        n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        while self._L.davo_profile_entry(self._ctx, i, name, 64, ctypes.byref(n), ctypes.byref(ms)) == 0:
            out[name.value.decode()] = (n.value, ms.value)
            i += 1
        return out

    def profile_samples(self, name, cap=8192):
        """(durations_ms, periods_ms) of every bracketed launch of `name` since the last profile_reset, in issue order
        (include/davo_hip.h: davo_profile_samples); period = start of the previous bracketed launch to this one's start."""
        out = []
        for which in (0, 1):
            buf = (ctypes.c_float * cap)()
            n = self._L.davo_profile_samples(self._ctx, name.encode(), which, buf, cap)
            if n < 0:
                self._check(n)
            out.append(np.array(buf[:min(n, cap)], np.float32))
        return out[0], out[1]

    def last_plan(self, layer):
        """[(mtiles, BN), ...] of the launches the last forward used for conv layer 0..6."""
        out = []
        for k in (0, 1):
            m, bn = ctypes.c_int(0), ctypes.c_int(0)
            self._check(self._L.davo_last_plan(self._ctx, layer, k, ctypes.byref(m), ctypes.byref(bn)))
            if m.value:
                out.append((m.value, bn.value))
        return out

    def debug_read(self, tensor, shape):
        out = np.empty(shape, np.float32)
        self._check(self._L.davo_debug_read(self._ctx, tensor.encode(), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), out.size))
        return out


def conv2d_same(x, w, b, stride=1, rate=1, relu=True, device=0, precision="f32"):
    """slim.conv2d(padding='SAME') through the MFMA implicit-GEMM kernel (test hook)."""
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    N, H, W, Cin = x.shape
    k, k2, ci, Cout = w.shape
    if k != k2 or ci != Cin:
        raise ValueError("weights %s do not fit input %s" % (w.shape, x.shape))
    y = np.empty((N, -(-H // stride), -(-W // stride), Cout), np.float32)
    err = ctypes.create_string_buffer(256)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = _lib.lib().davo_conv2d_same(device, x.ctypes.data_as(fp), N, H, W, Cin, w.ctypes.data_as(fp), k, Cout,
                                     b.ctypes.data_as(fp), stride, rate, int(relu),
                                     {"f32": 0, "f16x3": 1}[precision], y.ctypes.data_as(fp), err, 256)
    if rc != 0:
        raise _PY_ERR.get(rc, DavoError)(err.value.decode())
    return y


class DAVO(object):
    """Drop-in for the reference class on the inference path (reference davo.py:30)."""

    def __init__(self, version=None, att_19=None, device=0):
        self.version = version          # davo.py:32
        self.att_19 = att_19            # davo.py:33 (unused on the pose path)
        self.device = device
        self.engine = None
        self._weights = None

    def setup_inference(self, img_height, img_width, mode, seq_length=3, batch_size=1,
                        input_img_uint8=None, input_pose=None, input_flow=None, input_depth=None,
                        input_seglabel=None):
        """davo.py:1533-1551.  ``input_*`` are numpy arrays ([B,H,3W,3] u8, [B,4,H,W,2] f32,
        [B,3,H,W,1] f32) or ``input_img_uint8`` is an iterator yielding (img, flow, seg);
        ``input_pose`` / ``input_depth`` are accepted and ignored like the reference does for
        this variant."""
        self.img_height, self.img_width, self.mode, self.batch_size = img_height, img_width, mode, batch_size
        if self.mode != 'davo':
            return                                        # davo.py:1548: other modes do nothing
        if seq_length != 3:
            raise ValueError("seq_length %d: the pose path is built for 3-frame windows" % seq_length)
        self.seq_length, self.num_source = seq_length, seq_length - 1
        assert self.version is not None                   # davo.py:959
        self.cfg = parse_version(self.version)
        self.engine = Engine(self.cfg, img_height, img_width, batch_size, self.device)
        if self._weights is not None:
            self.engine.load_weights(self._weights)
        self._ahead = []                                   # iterator inputs: batches submitted and not returned yet
        if input_img_uint8 is not None and not isinstance(input_img_uint8, np.ndarray) and input_flow is None:
            self._inputs = iter(input_img_uint8)
            # the counterpart of the tf.data iterator the reference's graph pulls from (data_loader.py:321-324, prefetch): batches
            # go through the library's streaming entry point, so while inference() call n waits for its poses, batch n+1 is
            # already copied and running (two in flight on the GPU)
            self.engine.set_inflight(2)
        else:
            self._inputs = (input_img_uint8, input_flow, input_seglabel)

    def load_weights(self, weights):
        """Stand-in for tf.train.Saver(tf.trainable_variables()).restore (test_kitti_pose.py:129-131)."""
        self._weights = weights
        if self.engine is not None:
            self.engine.load_weights(weights)

    def calibrate(self, inputs):
        """Range-calibrate the f16x3 arithmetic on a sample batch (img, flow, seg); see Engine.calibrate.  Optional:
        ``inference`` re-calibrates by itself on the first batch that leaves the range (the reference's float32
        graph has no counterpart and never fails, davo.py:1553-1569); calling it up front only saves that re-issue."""
        if self.engine is None:
            raise DavoError("setup_inference(..., mode='davo') has not been called")
        return self.engine.calibrate(*inputs)

    def inference(self, sess=None, mode='pose', inputs=None):
        """davo.py:1553-1569: -> {'pose': float32 [B,2,6]}; ``sess`` is accepted and ignored."""
        if mode != 'pose':
            raise NotImplementedError("mode `%s': only 'pose' is built (davo.py:1555-1556)" % mode)
        if self.engine is None:
            raise DavoError("setup_inference(..., mode='davo') has not been called")
        if inputs is None and not isinstance(self._inputs, tuple):
            return {'pose': self._next_from_iterator()}
        if inputs is not None:
            img, flow, seg = inputs
        else:
            img, flow, seg = self._inputs
        if img is None or flow is None or seg is None:
            raise ValueError("image, flow and seglabel inputs are all required for version `%s'" % self.version)
        return {'pose': self.engine.forward(img, flow, seg)}

    def _next_from_iterator(self):
        """Poses of the iterator's next batch; the batch after it is submitted before this one is waited for.  A batch the
        iterator yields must stay valid until the iterator is asked for the next one (davo_amd.loader's contract)."""
        while len(self._ahead) < 2:
            item = next(self._inputs, None)
            if item is None:
                break
            img, flow, seg = item
            out = np.empty((np.shape(img)[0], 2, 6), np.float32)
            self.engine.submit(img, flow, seg, out)             # hold = 0: the batch is copied when this returns
            self._ahead.append(out)
        if not self._ahead:
            raise StopIteration("the input iterator is exhausted")      # tf.errors.OutOfRangeError's counterpart
        out = self._ahead.pop(0)
        self.engine.wait(len(self._ahead))
        return out
