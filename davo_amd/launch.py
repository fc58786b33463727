"""Rank launcher: one process per GPU of the node, started by a parent that never touches a GPU.

``python bench.py --gpus N`` and ``python -m davo_amd.run_kitti_pose --gpus N`` call
``spawn_ranks``: N children of the same command with RANK / LOCAL_RANK / WORLD_SIZE set (the same
variables ``python -m torch.distributed.run`` sets, so a command started by either launcher reads
its place the same way) and a fresh ``DAVO_COMM_DIR`` for the RCCL id (davo_amd/comm.py).  The
parent only waits: it returns the first non-zero exit code and stops the remaining ranks, so a failed
rank fails the run (no rank is left waiting in a collective).  Nothing here imports the HIP library
or initialises a device — a process that has done so must not be replaced or forked into ranks.

Each rank is given its own contiguous slice of the CPUs this process may use (``sched_getaffinity``) in ``DAVO_CPU_SLICE``
and binds ITSELF to it in its first statements (``bind_rank_cpus``: the top of ``davo_amd/__init__.py`` and of ``bench.py``,
before numpy or the HIP library start a thread), so the ranks' loader / decode threads do not migrate over each other's cores
(and stay on one NUMA side where the allowed set spans several).  No process here ever replaces its own program: an ``exec``
from a process that a preloaded library (``rocprofv3 -- python bench.py --gpus N``) has already made a GPU process takes the
machine down on this pool.  ``HSA_ENABLE_IPC_MODE_LEGACY=0`` is passed on (and defaulted) because RCCL shares device
buffers between the ranks' processes through HIP IPC handles, and the hosts' driver only supports the dmabuf form of
them: with the legacy mode ``ncclCommInitRank`` fails in ``hipIpcGetMemHandle: invalid argument``.
"""
import os
import secrets
import shutil
import signal
import subprocess
import sys
import tempfile
import time


CPU_SLICE_ENV = "DAVO_CPU_SLICE"


def bind_rank_cpus(environ=None):
    """A rank's first statement: bind this process to the CPU slice its launcher chose (``DAVO_CPU_SLICE``: comma-separated
    CPU numbers), before anything has started a thread - every thread and worker process the rank ever starts inherits it.
    Returns the set it bound to, or None (variable unset or empty, or the kernel refused).  Needs nothing but ``os``."""
    spec = (os.environ if environ is None else environ).get(CPU_SLICE_ENV, "")
    if not spec:
        return None
    try:
        cpus = {int(c) for c in spec.split(",")}
        os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, ValueError):
        return None


def cpu_slices(nprocs, cpus=None):
    """the allowed CPUs cut into nprocs contiguous slices (rank r gets slice r); fewer CPUs than ranks: everyone keeps all"""
    cpus = sorted(os.sched_getaffinity(0) if cpus is None else cpus)
    if len(cpus) < nprocs:
        return [set(cpus)] * nprocs
    return [set(cpus[r * len(cpus) // nprocs:(r + 1) * len(cpus) // nprocs]) for r in range(nprocs)]


def spawn_ranks(argv, nprocs, env_extra=None, timeout=None, bind_cpus=True, stdout=None):
    """Run ``sys.executable argv...`` as ``nprocs`` ranks; -> exit code (0 only if every rank returned 0).
    ``stdout``: a file the ranks' standard output goes to instead of this process's."""
    comm_dir = tempfile.mkdtemp(prefix="davo_comm_")        # 0700, this run's alone
    nonce = secrets.token_hex(8)                             # the id file's name is not predictable either
    slices = cpu_slices(nprocs) if bind_cpus and nprocs > 1 else None
    procs = []
    try:
        for r in range(nprocs):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nprocs), "LOCAL_WORLD_SIZE": str(nprocs),
                        "DAVO_COMM_DIR": comm_dir, "DAVO_COMM_NONCE": nonce,
                        "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
            env.setdefault("MASTER_ADDR", "127.0.0.1")
            if env_extra:
                env.update(env_extra)
            cmd = [sys.executable] + list(argv)
            if slices is not None:
                # the rank binds itself in its first statements (bind_rank_cpus).  Setting the affinity of p.pid from here after
                # Popen raced with the child's first threads; a bare interpreter that bound itself and then exec'ed the
                # command was one program replacement too many on this pool (ADVICE / VERDICT r4)
                env[CPU_SLICE_ENV] = ",".join(str(c) for c in sorted(slices[r]))
            else:
                env.pop(CPU_SLICE_ENV, None)
            procs.append(subprocess.Popen(cmd, env=env, stdout=stdout))
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
