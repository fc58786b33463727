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
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

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
    flow = np.load(flo).astype(np.float32, copy=False).reshape(4, H, W, 2)
    seg = np.load(sg).astype(np.float32, copy=False).reshape(3, H, W, 1)
    return img, flow, seg


def count_frames(dump_dir, seq, seq_length=3):
    """N = #jpg + 2*max_src_offset (test_kitti_pose.py:81-82)."""
    d = os.path.join(dump_dir, "%.2d" % seq)
    n = len([f for f in os.listdir(d) if f.endswith(".jpg")])
    return n + 2 * int((seq_length - 1) / 2)


class ThreadedWindowLoader:
    """Iterate batches of windows [lo, hi) in order; ``workers`` decode threads, ``prefetch`` batches ahead.

    ``load_one(w)`` returns the tensors of window w (target frame w + 1).  A failed load is re-raised in
    the consumer at the position of its batch.

    ``alloc(shape, dtype)`` (e.g. ``davo_amd.pinned_empty``) makes the batches live in a ring of
    ``prefetch + 2`` caller-provided buffer sets (one being filled, ``prefetch`` queued, one with the
    consumer), so a batch stays valid until the consumer asks for the next one."""

    def __init__(self, load_one, lo, hi, batch_size, workers=4, prefetch=2, alloc=None):
        self.load_one, self.lo, self.hi, self.B = load_one, lo, hi, batch_size
        self.workers, self.prefetch = max(1, workers), max(1, prefetch)
        self.alloc = alloc

    def __len__(self):
        return -(-(self.hi - self.lo) // self.B)

    def __iter__(self):
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        ring, turn = [], 0

        def producer():
            nonlocal turn
            try:
                with ThreadPoolExecutor(self.workers) as pool:
                    for s in range(self.lo, self.hi, self.B):
                        if stop.is_set():
                            return
                        e = min(s + self.B, self.hi)
                        parts = list(pool.map(self.load_one, range(s, e)))
                        if self.alloc is None:
                            batch = tuple(np.stack([p[k] for p in parts]) for k in range(3))
                        else:
                            if len(ring) < self.prefetch + 2:
                                ring.append(tuple(self.alloc((self.B,) + parts[0][k].shape, parts[0][k].dtype) for k in range(3)))
                            bufs = ring[turn % (self.prefetch + 2)]
                            turn += 1
                            batch = tuple(np.stack([p[k] for p in parts], out=bufs[k][:e - s]) for k in range(3))
                        q.put((s, e, batch, None))
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


def kitti_loader(dump_dir, seq, H, W, lo, hi, batch_size, workers=4, prefetch=2, alloc=None):
    """Windows [lo, hi) of a sequence dump; window w has target frame w + 1."""
    return ThreadedWindowLoader(lambda w: load_window(dump_dir, seq, w + 1, H, W), lo, hi, batch_size, workers, prefetch, alloc)


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
