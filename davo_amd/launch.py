"""Rank launcher: one process per GPU of the node, started by a parent that never touches a GPU.

``python bench.py --gpus N`` and ``python -m davo_amd.run_kitti_pose --gpus N`` call
``spawn_ranks``: N children of the same command with RANK / LOCAL_RANK / WORLD_SIZE set (the same
variables ``python -m torch.distributed.run`` sets, so a command started by either launcher reads
its place the same way) and a fresh ``DAVO_COMM_DIR`` for the RCCL id (davo_amd/comm.py).  The
parent only waits: it returns the first non-zero exit code and stops the remaining ranks, so a failed
rank fails the run (no rank is left waiting in a collective).  Nothing here imports the HIP library
or initialises a device — a process that has done so must not be replaced or forked into ranks.
"""
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time


def spawn_ranks(argv, nprocs, env_extra=None, timeout=None):
    """Run ``sys.executable argv...`` as ``nprocs`` ranks; -> exit code (0 only if every rank returned 0)."""
    comm_dir = tempfile.mkdtemp(prefix="davo_comm_")
    procs = []
    try:
        for r in range(nprocs):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nprocs), "LOCAL_WORLD_SIZE": str(nprocs),
                        "DAVO_COMM_DIR": comm_dir, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
            env.setdefault("MASTER_ADDR", "127.0.0.1")
            if env_extra:
                env.update(env_extra)
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
        t0 = time.time()
        code = 0
        live = list(procs)
        while live and code == 0:
            for p in list(live):
                rc = p.poll()
                if rc is not None:
                    live.remove(p)
                    if rc != 0:
                        code = rc
            if timeout is not None and time.time() - t0 > timeout:
                code = 124
            if live and code == 0:
                time.sleep(0.05)
        if code != 0:                                   # stop exactly the ranks this call started
            for p in live:
                p.send_signal(signal.SIGTERM)
            t1 = time.time()
            while any(p.poll() is None for p in live) and time.time() - t1 < 10.0:
                time.sleep(0.05)
            for p in live:
                if p.poll() is None:
                    p.kill()
        for p in procs:
            p.wait()
        return code
    finally:
        shutil.rmtree(comm_dir, ignore_errors=True)
