"""RCCL communicator of a window-sharded run, over the C ABI (include/davo_hip.h: ``davo_comm_*``).

One process per GPU; the reference has no multi-GPU code at all (``test_kitti_pose.py:133-149`` is
one process, one device), so this is new design: ranks run contiguous window ranges and meet once,
in an all-gather of ``[n,2,6]`` float32, before the sequential trajectory chain.  The communicator
is created on librccl itself — no ``torch.distributed``, no fallback transport: if RCCL cannot be
loaded or initialised the run fails.

The 128-byte ``ncclUniqueId`` travels from rank 0 to the other ranks through a small file:
``$DAVO_COMM_FILE``, or ``$DAVO_COMM_DIR/rccl_id`` (``davo_amd.launch.spawn_ranks`` creates a fresh
directory per run), or — under ``python -m torch.distributed.run``, which sets neither — a name built
from the launcher's PID and ``MASTER_PORT`` in the temp directory.
"""
import ctypes
import os
import tempfile
import time

import numpy as np

from . import _lib

_START = time.time()


def rendezvous_path():
    f = os.environ.get("DAVO_COMM_FILE")
    if f:
        return f
    d = os.environ.get("DAVO_COMM_DIR")
    if d:
        return os.path.join(d, "rccl_id")
    return os.path.join(tempfile.gettempdir(), "davo_comm_%d_%d_%s.id" % (os.getuid(), os.getppid(), os.environ.get("MASTER_PORT", "0")))


def world_from_env():
    """(rank, local_rank, world) as the launch contract sets them (RANK / LOCAL_RANK / WORLD_SIZE)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


class CommError(RuntimeError):
    pass


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
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "wb") as f:
                f.write(bytes(ident))
            os.replace(tmp, path)                       # readers see all 128 bytes or no file
        else:
            t0 = time.time()
            while True:
                try:
                    # a file left behind by an earlier, crashed run under the same name is older than this process
                    if os.path.getsize(path) == _lib.COMM_ID_BYTES and os.path.getmtime(path) >= _START - 60.0:
                        with open(path, "rb") as f:
                            raw = f.read()
                        if len(raw) == _lib.COMM_ID_BYTES:
                            ctypes.memmove(ident, raw, len(raw))
                            break
                except OSError:
                    pass
                if time.time() - t0 > timeout:
                    raise CommError("rank %d: no RCCL id at %s after %.0f s (rank 0 did not start?)" % (self.rank, path, timeout))
                time.sleep(0.02)
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
