"""RCCL communicator of a window-sharded run, over the C ABI (include/davo_hip.h: ``davo_comm_*``).

One process per GPU; the reference has no multi-GPU code at all (``test_kitti_pose.py:133-149`` is
one process, one device), so this is new design: ranks run contiguous window ranges and meet once,
in an all-gather of ``[n,2,6]`` float32, before the sequential trajectory chain.  The communicator
is created on librccl itself — no ``torch.distributed``, no fallback transport: if RCCL cannot be
loaded or initialised the run fails.

The 128-byte ``ncclUniqueId`` travels from rank 0 to the other ranks through a small file:
``$DAVO_COMM_FILE``, or ``$DAVO_COMM_DIR/rccl_id[.$DAVO_COMM_NONCE]`` (``davo_amd.launch.spawn_ranks`` creates a fresh
private directory and a random nonce per run), or — under ``python -m torch.distributed.run``, which sets neither — a
file in a private per-user directory whose name carries the launcher's PID, ``MASTER_PORT`` and the elastic agent's
run id and restart count, so a restarted worker group never reads the id of the group before it.  Rank 0 creates the
file exclusively (``O_EXCL``, mode 0600); a reader accepts it only if it is the reader's own (same uid) and not older
than the launcher process itself.
"""
import ctypes
import os
import stat
import tempfile
import time

import numpy as np

from . import _lib


class CommError(RuntimeError):
    pass


def _private_dir():
    """per-user 0700 directory under the temp dir (another local user can neither read nor pre-plant an id file)"""
    d = os.path.join(tempfile.gettempdir(), "davo_comm_uid%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise CommError("%s is not a private directory of uid %d" % (d, os.getuid()))
    return d


def rendezvous_path():
    f = os.environ.get("DAVO_COMM_FILE")
    if f:
        return f
    d = os.environ.get("DAVO_COMM_DIR")
    if d:
        nonce = os.environ.get("DAVO_COMM_NONCE")
        return os.path.join(d, "rccl_id" + ("." + nonce if nonce else ""))
    tag = "_".join(str(os.environ.get(k, "0")) for k in ("MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"))
    tag = "".join(ch if ch.isalnum() or ch in "_-" else "-" for ch in tag)
    return os.path.join(_private_dir(), "rccl_%d_%s.id" % (os.getppid(), tag))


def _launcher_start_time():
    """wall-clock start of the parent process (the launcher every rank shares): an id file older than it belongs to an
    earlier run.  Falls back to 0 (no age test) where /proc is not readable."""
    try:
        with open("/proc/%d/stat" % os.getppid()) as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])          # field 22: starttime in clock ticks since boot
        with open("/proc/stat") as f:
            btime = next(int(l.split()[1]) for l in f if l.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK") - 1.0             # btime is whole seconds
    except (OSError, ValueError, StopIteration, IndexError):
        return 0.0


def _own_start_time():
    try:
        with open("/proc/self/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/stat") as f:
            btime = next(int(l.split()[1]) for l in f if l.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK") - 1.0
    except (OSError, ValueError, StopIteration, IndexError):
        return 0.0


def id_not_before():
    """Oldest mtime an id file may have: not older than the launcher process, and never more than a minute older than this
    rank itself.  The second bound matters when the 'launcher' is a long-lived shell or scheduler step (ranks started by hand
    with DAVO_COMM_FILE): a file left by a crashed earlier run is younger than that shell but not than this rank."""
    own = _own_start_time()
    return max(_launcher_start_time(), own - 60.0 if own > 0 else 0.0)


def world_from_env():
    """(rank, local_rank, world) as the launch contract sets them (RANK / LOCAL_RANK / WORLD_SIZE)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def publish_id(path, ident):
    """rank 0: the 128 id bytes appear at `path` all at once (exclusive create of a temporary, then rename)"""
    tmp = "%s.%d.tmp" % (path, os.getpid())
    for stale in (tmp, path):                       # a file of this very name can only be a leftover of this launcher's
        try:
            os.remove(stale)
        except OSError:
            pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, "wb") as f:
        f.write(ident)
    os.replace(tmp, path)                           # readers see all 128 bytes or no file


def wait_for_id(path, rank, timeout=180.0, not_before=None):
    """ranks > 0: the id rank 0 published under THIS launcher (own uid, complete, not older than the launcher process)"""
    t0 = time.time()
    not_before = id_not_before() if not_before is None else not_before
    while True:
        try:
            st = os.stat(path)
            if st.st_size == _lib.COMM_ID_BYTES and st.st_uid == os.getuid() and st.st_mtime >= not_before:
                with open(path, "rb") as f:
                    raw = f.read()
                if len(raw) == _lib.COMM_ID_BYTES:
                    return raw
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise CommError("rank %d: no RCCL id at %s after %.0f s (rank 0 did not start?)" % (rank, path, timeout))
        time.sleep(0.02)


def preload_in_background():
    """Start loading librccl on a thread of its own (include/davo_hip.h: davo_comm_preload) and return at once: mapping its 573 MB
    and registering its code objects is most of a second that a rank can spend behind its own imports, weight file and GPU
    set-up.  -> the thread (nothing needs to join it: the first communicator call waits for the load by itself)."""
    import threading

    def load():
        err = ctypes.create_string_buffer(512)
        _lib.lib().davo_comm_preload(err, 512)          # a failure is reported by the communicator's constructor

    t = threading.Thread(target=load, name="davo-rccl-preload", daemon=True)
    t.start()
    return t


class PendingComm:
    """A communicator that is being built on a second thread (``RcclComm.from_env_async``): the id exchange and
    ``ncclCommInitRank`` - seconds - run while the rank's windows are already on the GPU; the first use (the pose gather) joins."""

    def __init__(self, engine, rank, world, path=None, timeout=180.0):
        import threading
        self._comm = self._exc = None
        self.t_ready = None

        def build():
            try:
                self._comm = RcclComm(engine, rank, world, path, timeout)
            except BaseException as e:                      # noqa: BLE001 - re-raised in the thread that uses the communicator
                self._exc = e
            self.t_ready = time.time()

        self._thread = threading.Thread(target=build, name="davo-rccl-init", daemon=True)
        self._thread.start()

    def get(self):
        self._thread.join()
        if self._exc is not None:
            raise self._exc
        return self._comm

    def allgather(self, local, n_per_rank=None):
        return self.get().allgather(local, n_per_rank)

    def allgather_device(self, d_local, n_per_rank, d_all):
        return self.get().allgather_device(d_local, n_per_rank, d_all)

    def allreduce(self, value, op="max"):
        return self.get().allreduce(value, op)

    def barrier(self):
        self.get().barrier()

    def close(self):
        self._thread.join()
        if self._comm is not None:
            self._comm.close()


class RcclComm:
    """``ncclCommInitRank`` on the engine's GPU.  Collective: every rank constructs it."""

    OPS = {"sum": 0, "max": 1, "min": 2}

    def __init__(self, engine, rank, world, path=None, timeout=180.0):
        self.engine, self.rank, self.world = engine, int(rank), int(world)
        self._L = _lib.lib()
        path = path or rendezvous_path()
        ident = (ctypes.c_uint8 * _lib.COMM_ID_BYTES)()
        if self.rank == 0:
            err = ctypes.create_string_buffer(512)
            rc = self._L.davo_comm_unique_id(ident, err, 512)
            if rc != 0:
                raise CommError("davo_comm_unique_id: %s" % err.value.decode())
            publish_id(path, bytes(ident))
        else:
            raw = wait_for_id(path, self.rank, timeout)
            ctypes.memmove(ident, raw, len(raw))
        engine._check(self._L.davo_comm_init(engine._ctx, self.world, self.rank, ident))
        self._open = True
        if self.rank == 0:                              # every rank has read the id once the collective init returned
            try:
                os.remove(path)
            except OSError:
                pass

    @classmethod
    def from_env(cls, engine):
        rank, _, world = world_from_env()
        return cls(engine, rank, world)

    @classmethod
    def from_env_async(cls, engine):
        """-> PendingComm: the communicator is built on a second thread and joined by its first use."""
        rank, _, world = world_from_env()
        return PendingComm(engine, rank, world)

    def allgather(self, local, n_per_rank=None):
        """local [n,2,6] (n <= n_per_rank; the rank's slot is zero-padded) -> ([world*n_per_rank,2,6], collective ms)."""
        local = np.ascontiguousarray(local, np.float32).reshape(-1, 2, 6)
        n = local.shape[0]
        per = n if n_per_rank is None else int(n_per_rank)
        out = np.empty((self.world * per, 2, 6), np.float32)
        ms = ctypes.c_float(0.0)
        fp = ctypes.POINTER(ctypes.c_float)
        self.engine._check(self._L.davo_allgather_poses(self.engine._ctx, local.ctypes.data_as(fp), n, per, out.ctypes.data_as(fp), ctypes.byref(ms)))
        return out, float(ms.value)

    def allgather_device(self, d_local, n_per_rank, d_all):
        ms = ctypes.c_float(0.0)
        self.engine._check(self._L.davo_allgather_poses_device(self.engine._ctx, d_local.ptr, int(n_per_rank), d_all.ptr, ctypes.byref(ms)))
        return float(ms.value)

    def allreduce(self, value, op="max"):
        v = ctypes.c_double(float(value))
        self.engine._check(self._L.davo_comm_allreduce(self.engine._ctx, ctypes.byref(v), self.OPS[op]))
        return v.value

    def barrier(self):
        self.engine._check(self._L.davo_comm_barrier(self.engine._ctx))

    def close(self):
        if getattr(self, "_open", False) and getattr(self.engine, "_ctx", None):
            self._L.davo_comm_destroy(self.engine._ctx)
        self._open = False
