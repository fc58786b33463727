"""Input pipeline in front of the path — row f2.

The reference feeds the graph from a ``tf.data`` pipeline (``data_loader.py:241-325``): file names ->
``read_file + decode_jpeg`` on 4 threads and ``tf.py_func(np.load)`` for the flow / seglabel arrays ->
``zip`` -> ``batch(B)`` -> ``prefetch(8B)``.  On disk (``doc/preprocessing.md:50-114``,
``data/preprocess.py:61-66``, ``test_kitti_pose.py:44-49``) a window with target frame ``FFFFFF`` of
sequence ``SS`` is

    <dump>/SS/FFFFFF.jpg              128 x 1248 RGB strip  src0 | tgt | src1
    <dump>/SS/FFFFFF-flownet2.npy     float32 (4, H, W, 2)
    <dump>/SS/FFFFFF-seglabel.npy     float32 (3, H, W, 1)   (file order src0, tgt, src1)

``ThreadedWindowLoader`` is the same pipeline with a thread pool and a bounded prefetch queue: batches
come out in window order while the next ones are being decoded, so file IO overlaps the GPU.
JPEG decoders differ by +-1 LSB between libjpeg builds (TF's vs Pillow's), which is why parity of the
path is defined from decoded tensors, not from .jpg files.
"""
import collections
import os
import queue
import threading
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np


def window_paths(dump_dir, seq, tgt_idx):
    stem = os.path.join(dump_dir, "%.2d" % seq, "%.6d" % tgt_idx)        # test_kitti_pose.py:44-49
    return stem + ".jpg", stem + "-flownet2.npy", stem + "-seglabel.npy"


def load_window(dump_dir, seq, tgt_idx, H, W):
    """One window -> (img u8 [H,3W,3], flow f32 [4,H,W,2], seg f32 [3,H,W,1])."""
    from PIL import Image
    jpg, flo, sg = window_paths(dump_dir, seq, tgt_idx)
    with Image.open(jpg) as im:
        img = np.asarray(im.convert("RGB"), np.uint8)
    if img.shape != (H, 3 * W, 3):
        raise ValueError("%s is %s, expected %s" % (jpg, img.shape, (H, 3 * W, 3)))
    # memory-mapped: the one copy of these arrays is the worker's write into its slot of the batch buffer
    flow = np.load(flo, mmap_mode="r").astype(np.float32, copy=False).reshape(4, H, W, 2)
    seg = np.load(sg, mmap_mode="r").astype(np.float32, copy=False).reshape(3, H, W, 1)
    return img, flow, seg


def _read_npy_into(path, dst):
    """Read a C-ordered .npy of dst's dtype and size straight into dst (no intermediate array); anything else
    goes through np.load + cast."""
    with open(path, "rb") as f:
        major, _ = np.lib.format.read_magic(f)
        shape, fortran, dtype = (np.lib.format.read_array_header_1_0 if major == 1 else np.lib.format.read_array_header_2_0)(f)
        if not fortran and dtype == dst.dtype and int(np.prod(shape)) == dst.size and dst.flags.c_contiguous:
            if f.readinto(memoryview(dst).cast("B")) != dst.nbytes:
                raise ValueError("%s is truncated" % path)
            return
    dst[...] = np.load(path).astype(dst.dtype, copy=False).reshape(dst.shape)


def _decode_jpeg(path):
    """Runs in a decode process: file -> (H, W, raw RGB bytes)."""
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert("RGB")
        return im.height, im.width, im.tobytes()


class JpegDecodePool:
    """Decode processes for the strips.  This Pillow build keeps the GIL while decoding, so threads alone give
    one core's worth of JPEG decode; the reference's tf.image.decode_jpeg runs on 4 native threads
    (data_loader.py:283-288).  Start method "spawn": the parent may already hold a HIP context, which must
    not be forked."""

    def __init__(self, nprocs):
        import multiprocessing as mp
        self.pool = ProcessPoolExecutor(nprocs, mp_context=mp.get_context("spawn"))

    def decode(self, path):
        h, w, raw = self.pool.submit(_decode_jpeg, path).result()
        return np.frombuffer(raw, np.uint8).reshape(h, w, 3)

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)


def load_window_into(dump_dir, seq, tgt_idx, H, W, img, flow, seg, decoder=None):
    """load_window writing into the caller's [H,3W,3] / [4,H,W,2] / [3,H,W,1] slots."""
    jpg, flo, sg = window_paths(dump_dir, seq, tgt_idx)
    if decoder is None:
        from PIL import Image
        with Image.open(jpg) as im:
            a = np.asarray(im.convert("RGB"), np.uint8)
    else:
        a = decoder.decode(jpg)
    if a.shape != (H, 3 * W, 3):
        raise ValueError("%s is %s, expected %s" % (jpg, a.shape, (H, 3 * W, 3)))
    img[...] = a
    _read_npy_into(flo, flow)
    _read_npy_into(sg, seg)


def count_frames(dump_dir, seq, seq_length=3):
    """N = #jpg + 2*max_src_offset (test_kitti_pose.py:81-82)."""
    d = os.path.join(dump_dir, "%.2d" % seq)
    n = len([f for f in os.listdir(d) if f.endswith(".jpg")])
    return n + 2 * int((seq_length - 1) / 2)


class ThreadedWindowLoader:
    """Iterate batches of windows [lo, hi) in order; ``workers`` decode threads, ``prefetch`` batches ahead.

    ``load_one(w)`` returns the tensors of window w (target frame w + 1).  A failed load is re-raised in
    the consumer at the position of its batch.

    Batches live in a ring of ``prefetch + 4`` buffer sets (two being filled, one ready, ``prefetch`` queued,
    one with the consumer): a batch is valid until the consumer asks for the next one — copy it to keep it longer.
    Each decode thread writes its window straight into its slot of the batch.  ``alloc(shape, dtype)``
    provides the buffers (default ``np.empty``; ``davo_amd.pinned_empty`` for page-locked memory)."""

    def __init__(self, load_one, lo, hi, batch_size, workers=4, prefetch=2, alloc=None, load_into=None, on_close=None):
        self.load_one, self.lo, self.hi, self.B = load_one, lo, hi, batch_size
        self.workers, self.prefetch = max(1, workers), max(1, prefetch)
        self.alloc, self.load_into, self.on_close = alloc, load_into, on_close

    def __len__(self):
        return -(-(self.hi - self.lo) // self.B)

    def __iter__(self):
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        FILL = 2                                       # batches being decoded at once (no idle threads at batch ends)
        nring = self.prefetch + FILL + 2               # + one waiting in q.put, + one with the consumer
        ring, turn = [], 0
        alloc = self.alloc or np.empty

        def fill(w, bufs, i, parts=None):              # runs in a pool thread: decode + the single copy into the batch
            if parts is None and self.load_into is not None:
                self.load_into(w, bufs[0][i], bufs[1][i], bufs[2][i])
                return
            parts = self.load_one(w) if parts is None else parts
            for k in range(3):
                bufs[k][i] = parts[k]

        def producer():
            nonlocal turn
            try:
                with ThreadPoolExecutor(self.workers) as pool:
                    first = self.load_one(self.lo) if self.lo < self.hi else None   # shapes and dtypes of a window
                    starts = iter(range(self.lo, self.hi, self.B))
                    pending = collections.deque()
                    while True:
                        while len(pending) < FILL:
                            s = next(starts, None)
                            if s is None:
                                break
                            e = min(s + self.B, self.hi)
                            if len(ring) < nring:
                                ring.append(tuple(alloc((self.B,) + p.shape, p.dtype) for p in first))
                            bufs = ring[turn % nring]
                            turn += 1
                            pending.append((s, e, bufs, [pool.submit(fill, w, bufs, w - s, first if w == self.lo else None)
                                                         for w in range(s, e)]))
                        if not pending or stop.is_set():
                            break
                        s, e, bufs, futs = pending.popleft()
                        for f in futs:
                            f.result()
                        q.put((s, e, tuple(b[:e - s] for b in bufs), None))
                    for _, _, _, futs in pending:
                        for f in futs:
                            f.cancel()
            except BaseException as exc:            # noqa: BLE001 — hand the failure to the consumer
                q.put((None, None, None, exc))
                return
            q.put(None)

        t = threading.Thread(target=producer, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                s, e, batch, exc = item
                if exc is not None:
                    raise exc
                yield s, e, batch
        finally:
            stop.set()
            while t.is_alive():                      # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.05)
            if self.on_close is not None:
                self.on_close()


def kitti_loader(dump_dir, seq, H, W, lo, hi, batch_size, workers=4, prefetch=2, alloc=None, decode_procs=0):
    """Windows [lo, hi) of a sequence dump; window w has target frame w + 1.  ``decode_procs`` > 0 decodes the
    strips in that many processes (the ``workers`` threads then only wait for them and read the .npy files)."""
    dec = JpegDecodePool(decode_procs) if decode_procs > 0 else None
    return ThreadedWindowLoader(
        lambda w: load_window(dump_dir, seq, w + 1, H, W), lo, hi, batch_size,
        max(workers, 2 * decode_procs), prefetch, alloc,
        load_into=lambda w, i, f, s: load_window_into(dump_dir, seq, w + 1, H, W, i, f, s, dec),
        on_close=dec.close if dec else None)


def write_synthetic_dump(dump_dir, seq, n_frames, H, W, seed=None, quality=95):
    """Write a dump in the reference's on-disk format from the seeded synthetic tensors (no KITTI offline).
    Returns the number of windows written."""
    from PIL import Image
    from . import synth
    d = os.path.join(dump_dir, "%.2d" % seq)
    os.makedirs(d, exist_ok=True)
    for w in range(n_frames - 2):
        img, flow, seg = synth.make_inputs(1, H, W, seed=synth.SEED if seed is None else seed, first_window=w)
        jpg, flo, sg = window_paths(dump_dir, seq, w + 1)
        Image.fromarray(img[0]).save(jpg, quality=quality)
        np.save(flo, flow[0])
        np.save(sg, seg[0])
    return n_frames - 2
