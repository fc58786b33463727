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


def _read_npy_into(path, dst, planes=None):
    """Read a C-ordered .npy of dst's dtype and size straight into dst (no intermediate array); anything else
    goes through np.load + cast.  ``planes``: indices along the first axis that are wanted — the others are skipped
    in the file (seek) and left as they are in dst (the path reads flow planes 0,1 and the source frames' label maps only:
    davo.py:978-982,998-1004, so 1.3 of a window's 2.3 MB never have to leave the page cache)."""
    with open(path, "rb") as f:
        major, _ = np.lib.format.read_magic(f)
        shape, fortran, dtype = (np.lib.format.read_array_header_1_0 if major == 1 else np.lib.format.read_array_header_2_0)(f)
        if not fortran and dtype == dst.dtype and int(np.prod(shape)) == dst.size and dst.flags.c_contiguous:
            if planes is None:
                if f.readinto(memoryview(dst).cast("B")) != dst.nbytes:
                    raise ValueError("%s is truncated" % path)
                return
            base, per = f.tell(), dst[0].nbytes
            for k in planes:
                f.seek(base + k * per)
                if f.readinto(memoryview(dst[k]).cast("B")) != per:
                    raise ValueError("%s is truncated" % path)
            return
    a = np.load(path).astype(dst.dtype, copy=False).reshape(dst.shape)
    if planes is None:
        dst[...] = a
    else:
        for k in planes:
            dst[k] = a[k]


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


FLOW_PLANES_USED = (0, 1)           # davo.py:978-982: pred_flows = [0, flow[:,0], flow[:,1]]
SEG_PLANES_SOURCES = (0, 2)         # davo.py:998-1004 + 1408-1412: the target frame's attention is overwritten by ones


def load_window_into(dump_dir, seq, tgt_idx, H, W, img, flow, seg, decoder=None, flow_planes=None, seg_planes=None):
    """load_window writing into the caller's [H,3W,3] / [4,H,W,2] / [3,H,W,1] slots; ``flow_planes`` / ``seg_planes``
    restrict the .npy reads to the planes the variant consumes (None = all)."""
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
    _read_npy_into(flo, flow, flow_planes)
    _read_npy_into(sg, seg, seg_planes)


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


# ---- worker processes that fill shared batch buffers --------------------------------------------------------------
_W = {}                              # per worker process: the attached buffer ring and the dump it reads


def worker_context():
    """The multiprocessing context the loader's workers are started from: a FORK SERVER that has this module, numpy and Pillow
    imported already, so a worker is a fork of a warm interpreter (a few ms) instead of a spawned one that imports them
    (~0.3-0.5 s each, twelve at once).  The server itself is a spawned process that never touches a GPU, so forking it is safe
    whatever the parent holds (a HIP context must not be forked); call ``warm_workers()`` as early as possible - before the
    parent's own heavy imports and its GPU set-up - and the server's imports pass behind them."""
    import multiprocessing as mp
    ctx = mp.get_context("forkserver")
    ctx.set_forkserver_preload(["davo_amd.loader", "numpy", "PIL.Image", "PIL.JpegImagePlugin"])
    return ctx


def warm_workers():
    """Start the fork server now (idempotent, returns at once: its imports run in the server process)."""
    worker_context()
    from multiprocessing import forkserver
    forkserver.ensure_running()


def shm_budget_bytes():
    """Bytes of shared, page-locked batch buffers a loader may create: a quarter of what /dev/shm has free, at most 2 GiB
    (containers often mount 64 MB there: the first touch of a larger segment is a SIGBUS, not an exception)."""
    try:
        st = os.statvfs("/dev/shm")
        free = st.f_bavail * st.f_frsize
    except OSError:
        return 0
    return int(min(free // 4, 2 << 30))


def _keep_freed_memory_mapped():
    """A window's decode allocates and frees the strip four times over (Pillow's image core, the RGB copy, its bytes, the array:
    480 KB each).  In a worker forked from the fresh fork server glibc serves each of them by mmap and returns it by munmap, so
    every window pays ~470 minor faults on brand-new zero pages (177 k faults per worker and run, 546 per window, measured:
    profiles/r05d_loader_diag2.log) and a worker needs 1.16 ms per window where the same loop in a process whose allocator has
    already learnt to keep such blocks needs 0.66.  Tell the allocator up front: blocks up to 32 MB come from the heap, and the heap
    is not trimmed."""
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 32 << 20)            # M_MMAP_THRESHOLD (glibc's maximum)
        libc.mallopt(-1, 1 << 30)             # M_TRIM_THRESHOLD
        libc.mallopt(-2, 16 << 20)            # M_TOP_PAD
    except (OSError, AttributeError):
        pass                                  # another libc: slower, not wrong


def _chunk_owner(c, chunks_per_ring, P):
    """Worker that fills chunk c.  A chunk's place in the ring of batch buffers is c % chunks_per_ring and that place is always
    filled by the same worker, so a worker only ever writes its own pages of the shared buffers: its first touch of a page (a minor
    fault per page and process although the page exists) happens once, in the first trip round the ring, not on every window."""
    return (c % chunks_per_ring) % P


def _slot_worker(k, P, names, ctrl_name, B, H, W, dump_dir, seq, flow_planes, seg_planes, lo, hi, chunk, nring, sem, errq):
    """Worker k of P (a fork of the warm fork server): fills its chunks of windows [lo, hi) in order, straight into the shared batch
    buffers.  No task queue: which chunks are its own follows from k (_chunk_owner); ctrl[0] = number of batches that may be filled
    (the consumer's progress + the ring), ctrl[1] = stop, ctrl[2 + k] = chunks this worker has finished, ctrl[2 + P + k] = ns it
    worked.  One semaphore post per finished chunk wakes the consumer."""
    import time
    from multiprocessing import shared_memory
    _keep_freed_memory_mapped()
    ctrl_shm = shared_memory.SharedMemory(name=ctrl_name)
    ctrl = np.ndarray((2 + 2 * P,), np.int64, buffer=ctrl_shm.buf)
    segs = []
    try:
        ring = []
        for trio in names:
            # the names are registered with the parent's resource tracker already; the parent's close() is what unlinks them
            t = [shared_memory.SharedMemory(name=n) for n in trio]
            segs += t
            ring.append((np.ndarray((B, H, 3 * W, 3), np.uint8, buffer=t[0].buf), np.ndarray((B, 4, H, W, 2), np.float32, buffer=t[1].buf),
                         np.ndarray((B, 3, H, W, 1), np.float32, buffer=t[2].buf)))
        cpb = -(-B // chunk)                               # chunks per batch (the last one of a batch may be short)
        nbatches = -(-(hi - lo) // B)
        for c in range(nbatches * cpb):
            if _chunk_owner(c, cpb * nring, P) != k:
                continue
            bi, j = divmod(c, cpb)
            w0 = lo + bi * B + j * chunk
            w1 = min(w0 + chunk, lo + bi * B + B, hi)
            if w0 < w1:
                while ctrl[0] <= bi:                       # the ring entry still holds a batch the consumer has not released
                    if ctrl[1]:
                        return
                    time.sleep(0.0002)
                if ctrl[1]:
                    return
                t0 = time.perf_counter_ns()
                img, flow, seg = ring[bi % nring]
                for w in range(w0, w1):
                    i = w - lo - bi * B
                    load_window_into(dump_dir, seq, w + 1, H, W, img[i], flow[i], seg[i], None, flow_planes, seg_planes)
                ctrl[2 + P + k] += time.perf_counter_ns() - t0
            ctrl[2 + k] += 1
            sem.release()
        if os.environ.get("DAVO_LOADER_DIAG"):             # measurement aid (tools/exp/loader_breakdown.py): where this worker's time went
            import resource
            import sys
            r = resource.getrusage(resource.RUSAGE_SELF)
            sys.stderr.write("loader worker %2d: busy %.3f s wall, %.3f s cpu in total, %d minor faults, %d voluntary / %d involuntary context "
                             "switches, cpu %s\n" % (k, ctrl[2 + P + k] * 1e-9, time.process_time(), r.ru_minflt, r.ru_nvcsw, r.ru_nivcsw,
                                                     open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36]))
    except BaseException as exc:                           # noqa: BLE001 - hand the failure to the consumer
        try:
            errq.put(exc)
        except Exception:                                  # noqa: BLE001 - an exception that does not pickle
            errq.put(RuntimeError("loader worker %d: %r" % (k, exc)))
        ctrl[1] = 1
        sem.release()
    finally:
        del ctrl
        for sm in segs + [ctrl_shm]:
            try:
                sm.close()
            except (OSError, BufferError):
                pass


class ShmBudgetError(RuntimeError):
    """The shared batch buffers of a ProcessWindowLoader do not fit /dev/shm (see shm_budget_bytes)."""


class ProcessWindowLoader:
    """Batches of windows [lo, hi) in order, filled by ``procs`` worker PROCESSES that decode the strip and read the
    .npy planes of whole windows straight into shared-memory batch buffers (data_loader.py:241-325's pipeline with
    processes where TF has native threads: Pillow decodes under the GIL, so threads give one core's worth of JPEG).
    The parent never touches a pixel and runs no task queue: every worker knows which chunks of ``chunk`` windows are its own
    (``_chunk_owner``: always the same places of the ring, so it writes only pages it has touched before), fills them in order as
    far as the consumer's progress allows, and counts them in a small shared control block; the consumer yields a batch when the
    counts say its chunks are done.  (Round 3/4 handed chunks out through a ProcessPoolExecutor: a worker then met every page of
    every buffer for the first time sooner or later - 460 minor faults per window - and needed 1.3 ms per window where a plain
    process needs 0.66: 12.3 k windows/s on 14 workers against 20.6 k, profiles/r05b_loader_breakdown.log, r05c_loader_diag.log.)
    ``pin(array)`` / ``unpin(array)`` page-lock the buffers for the H2D DMA (davo_amd.pin_array: hipHostRegister over the shared
    mapping).  Only the flow planes and label maps the variant consumes are read; the rest of a slot keeps its zeros.  Workers
    are forks of a warm fork server (worker_context): the parent may hold a HIP context, the server never does.

    A batch is valid until the consumer has asked for ``hold`` + 1 further ones (``hold`` = 0: until the next one); the last
    ones until ``close()`` (or the loader's deletion), which unpins and unmaps the buffers.  The end of iteration stops the
    workers and removes the segments' names."""

    def __init__(self, dump_dir, seq, H, W, lo, hi, batch_size, procs=8, prefetch=2, chunk=None, pin=None, unpin=None,
                 flow_planes=FLOW_PLANES_USED, seg_planes=SEG_PLANES_SOURCES, shm_budget=None, hold=0):
        self.args = (dump_dir, seq, H, W)
        self.lo, self.hi, self.B = lo, hi, batch_size
        self.procs, self.prefetch, self.hold = max(1, procs), max(1, prefetch), max(0, hold)
        self.chunk = max(1, min(chunk or 4, batch_size))
        # batches being filled at once: enough windows in flight (three chunks per worker) that no worker idles at a batch's end
        self.fill = max(2, -(-3 * self.procs * self.chunk // batch_size))
        # the ring is sized in BYTES: prefetch + fill + 2 (+ hold) batch buffer trios if they fit the budget (shm_budget_bytes),
        # fewer batches in flight if not, and a clear error - the caller falls back to the threaded loader - if not even four fit
        per_batch = batch_size * (H * 3 * W * 3 + 4 * H * W * 2 * 4 + 3 * H * W * 4)
        budget = shm_budget_bytes() if shm_budget is None else shm_budget
        want = self.prefetch + self.fill + 2 + self.hold
        self.nring = min(want, budget // per_batch)
        if self.nring < 4 + self.hold:     # one being filled, two ready / in flight, one (+ hold) with the consumer
            raise ShmBudgetError("batch buffers of %.0f MB each do not fit /dev/shm's budget of %.0f MB %d times "
                                 "(ProcessWindowLoader needs shared memory: use the threaded loader, --loader_procs 0)"
                                 % (per_batch / 2 ** 20, budget / 2 ** 20, 4 + self.hold))
        if self.nring < want:
            self.fill = max(1, self.nring - 3 - self.hold)
            self.prefetch = self.nring - 2 - self.fill - self.hold
        self.pin, self.unpin, self.fp, self.sp = pin, unpin, flow_planes, seg_planes
        self._segs, self._views, self._pool, self._pinned, self._unlinked = [], [], None, [], []
        self._ctrl_shm = self._ctrl = self._sem = self._errq = None
        self._started = False
        self.busy_s = 0.0

    def __len__(self):
        return -(-(self.hi - self.lo) // self.B)

    def _open(self):
        self.close()                                      # a second iteration starts from fresh buffers
        from multiprocessing import shared_memory
        dump_dir, seq, H, W = self.args
        B, P = self.B, self.procs
        sizes = (B * H * 3 * W * 3, B * 4 * H * W * 2 * 4, B * 3 * H * W * 4)
        for _ in range(self.nring):
            trio = [shared_memory.SharedMemory(create=True, size=max(n, 1)) for n in sizes]
            self._segs.append(trio)
            views = (np.ndarray((B, H, 3 * W, 3), np.uint8, buffer=trio[0].buf),
                     np.ndarray((B, 4, H, W, 2), np.float32, buffer=trio[1].buf),
                     np.ndarray((B, 3, H, W, 1), np.float32, buffer=trio[2].buf))
            self._views.append(views)
        self._ctrl_shm = shared_memory.SharedMemory(create=True, size=8 * (2 + 2 * P))
        self._ctrl = np.ndarray((2 + 2 * P,), np.int64, buffer=self._ctrl_shm.buf)
        self._ctrl[:] = 0
        self._ctrl[0] = self.nring                        # nothing is with the consumer yet: every ring entry may be filled
        names = [[sm.name for sm in trio] for trio in self._segs]
        ctx = worker_context()
        self._sem, self._errq = ctx.Semaphore(0), ctx.SimpleQueue()
        self._pool = [ctx.Process(target=_slot_worker, daemon=True,
                                  args=(k, P, names, self._ctrl_shm.name, B, H, W, dump_dir, seq, self.fp, self.sp, self.lo, self.hi,
                                        self.chunk, self.nring, self._sem, self._errq)) for k in range(P)]
        for p in self._pool:                              # forks of the warm server: they attach the buffers and start filling at once
            p.start()
        # Page-locking (hipHostRegister) allocates and pins every page: 0.3 s for the 1.4 GB of a batch-64 ring, which used to sit in
        # front of a rank's first batch.  It runs on a thread of its own now, ring entry by ring entry, behind the workers' start; the
        # consumer waits only for the entry of the batch it is about to hand out.
        self._pin_ready = [threading.Event() for _ in self._views]
        self._pin_thread = None
        if self.pin is not None:
            def pin_all():
                try:
                    for i, views in enumerate(self._views):
                        for v in views:
                            self.pin(v)
                            self._pinned.append(v)
                        self._pin_ready[i].set()
                except BaseException as exc:               # noqa: BLE001 - re-raised by the consumer
                    self._pin_exc = exc
                    for ev in self._pin_ready:
                        ev.set()
            self._pin_exc = None
            self._pin_thread = threading.Thread(target=pin_all, name="davo-loader-pin", daemon=True)
            self._pin_thread.start()
        else:
            for ev in self._pin_ready:
                ev.set()
        # what batch bi needs: per worker, the number of its chunks among the chunks of batches 0..bi (its chunks run in order)
        self._cpb = -(-B // self.chunk)
        self._need = np.zeros(P, np.int64)
        self._need_upto = 0                               # chunks counted into _need so far

    def _stop(self):
        """end of iteration: the workers stop; the names leave /dev/shm (nothing leaks if the process dies from here on) while
        the mappings - and with them the last batches the consumer may still hold - stay valid until close()"""
        if self._pool is not None:
            if self._ctrl is not None:
                self.busy_s = float(self._ctrl[2 + self.procs:2 + 2 * self.procs].sum()) * 1e-9
                self._ctrl[1] = 1
            for p in self._pool:
                p.join(timeout=5.0)
            for p in self._pool:
                if p.is_alive():
                    p.terminate()
                    p.join(timeout=1.0)
            self._pool = None
        for sm in [sm for trio in self._segs for sm in trio] + ([self._ctrl_shm] if self._ctrl_shm is not None else []):
            if sm not in self._unlinked:
                self._unlinked.append(sm)
                try:
                    sm.unlink()
                except OSError:
                    pass

    def close(self):
        """unpin and unmap the batch buffers: every array this loader has yielded is invalid afterwards (numpy does not keep
        a shared-memory mapping alive).  Called by __del__; iteration itself only stops the workers (_stop)."""
        self._stop()
        self._started = False
        t = getattr(self, "_pin_thread", None)
        if t is not None:
            t.join()
            self._pin_thread = None
        if self.unpin is not None:
            for v in self._pinned:
                try:
                    self.unpin(v)
                except Exception:                         # noqa: BLE001 — teardown must reach the unmap below
                    pass
        self._pinned = []
        self._views = []
        self._ctrl = None
        for sm in [sm for trio in self._segs for sm in trio] + ([self._ctrl_shm] if self._ctrl_shm is not None else []):
            try:
                sm.close()
            except (OSError, BufferError):
                pass
        self._segs, self._unlinked, self._ctrl_shm = [], [], None

    def __del__(self):
        try:
            self.close()
        except Exception:                                 # noqa: BLE001 — interpreter teardown
            pass

    def start(self):
        """Create the buffers, start the workers and begin filling batches now (idempotent): the first ``nring`` batches are
        decoded while the caller is still setting up its GPU context."""
        if not self._started:
            self._open()
            self._started = True
        return self

    def _wait_for(self, bi):
        """block until every chunk of batch bi is in its buffer; re-raises a worker's failure"""
        P, total = self.procs, len(self) * self._cpb
        upto = min((bi + 1) * self._cpb, total)
        for c in range(self._need_upto, upto):
            self._need[_chunk_owner(c, self._cpb * self.nring, P)] += 1
        self._need_upto = max(self._need_upto, upto)
        prog = self._ctrl[2:2 + P]
        while not (prog >= self._need).all():
            if not self._sem.acquire(timeout=0.05):
                for p in self._pool:
                    if not p.is_alive() and p.exitcode not in (0, None):
                        raise RuntimeError("a loader worker died (exit code %s)" % p.exitcode)
            if self._ctrl[1]:
                if not self._errq.empty():
                    raise self._errq.get()
                raise RuntimeError("the loader's workers stopped")

    def __iter__(self):
        self.start()
        nb = len(self)
        try:
            for bi in range(nb):
                # asking for batch bi releases the batches up to bi - 1 - hold: their ring entries may be refilled
                self._ctrl[0] = max(0, bi - self.hold) + self.nring
                self._wait_for(bi)
                self._pin_ready[bi % self.nring].wait()
                if getattr(self, "_pin_exc", None) is not None:
                    raise self._pin_exc
                s = self.lo + bi * self.B
                e = min(s + self.B, self.hi)
                yield s, e, tuple(v[:e - s] for v in self._views[bi % self.nring])
        finally:
            self._started = False
            self._stop()


def kitti_loader(dump_dir, seq, H, W, lo, hi, batch_size, workers=4, prefetch=2, alloc=None, decode_procs=0):
    """Windows [lo, hi) of a sequence dump; window w has target frame w + 1.  ``decode_procs`` > 0 decodes the
    strips in that many processes (the ``workers`` threads then only wait for them and read the .npy files)."""
    dec = JpegDecodePool(decode_procs) if decode_procs > 0 else None
    return ThreadedWindowLoader(
        lambda w: load_window(dump_dir, seq, w + 1, H, W), lo, hi, batch_size,
        max(workers, 2 * decode_procs), prefetch, alloc,
        load_into=lambda w, i, f, s: load_window_into(dump_dir, seq, w + 1, H, W, i, f, s, dec),
        on_close=dec.close if dec else None)


def scene_like_strip(H, W, window, seed=0):
    """A u8 [H,3W,3] strip with the statistics of a photograph rather than of white noise: four octaves of smooth random
    fields per frame plus sensor-like noise.  JPEG cost depends on content: uniform noise at quality 95 (the parity inputs of
    ``synth.make_inputs``) is 188 KB and 1.3-2.0 ms of libjpeg per 128x1248 strip, the worst case; a strip like this one at
    quality 75 — the default of ``scipy.misc.imsave``, which wrote the reference's dumps (data/preprocess.py:65) — is ~50 KB: tens of KB,
    the order of a photographic strip of this size (no KITTI frame is available offline to pin it closer)."""
    from PIL import Image
    rng = np.random.default_rng(1000003 * seed + window)
    out = np.empty((H, 3 * W, 3), np.float32)
    for f in range(3):
        acc = np.zeros((H, W, 3), np.float32)
        for gh, gw, amp in ((3, 6, 70.0), (9, 27, 35.0), (33, 105, 16.0), (65, 209, 12.0)):
            g = Image.fromarray(rng.integers(0, 256, (gh, gw, 3), dtype=np.uint8))
            acc += amp * (np.asarray(g.resize((W, H), Image.BILINEAR), np.float32) / 127.5 - 1.0)
        out[:, f * W:(f + 1) * W] = acc
    out += 128.0 + rng.normal(0.0, 20.0, out.shape)
    return np.clip(out, 0, 255).astype(np.uint8)


def write_synthetic_dump(dump_dir, seq, n_frames, H, W, seed=None, quality=None, images="noise"):
    """Write a dump in the reference's on-disk format from the seeded synthetic tensors (no KITTI offline).
    ``images``: "noise" = the parity inputs' uniform-noise strips at quality 95 (worst case for the decoder), "scene" =
    ``scene_like_strip`` at quality 75 (a real dump's file size and decode cost).  Returns the number of windows written."""
    from PIL import Image
    from . import synth
    if images not in ("noise", "scene"):
        raise ValueError("images must be 'noise' or 'scene'")
    if quality is None:
        quality = 95 if images == "noise" else 75
    d = os.path.join(dump_dir, "%.2d" % seq)
    os.makedirs(d, exist_ok=True)
    for w in range(n_frames - 2):
        img, flow, seg = synth.make_inputs(1, H, W, seed=synth.SEED if seed is None else seed, first_window=w)
        jpg, flo, sg = window_paths(dump_dir, seq, w + 1)
        strip = img[0] if images == "noise" else scene_like_strip(H, W, w, synth.SEED if seed is None else seed)
        Image.fromarray(strip).save(jpg, quality=quality)
        np.save(flo, flow[0])
        np.save(sg, seg[0])
    return n_frames - 2
