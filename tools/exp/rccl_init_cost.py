#!/usr/bin/env python
"""Measurement only: what ncclCommInitRank costs a fresh process (world size 1 on device 0), under a few environment settings a
single-node job may choose.  One child process per setting and repetition; the parent never touches the GPU."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
t0 = time.time()
from davo_amd import _lib
from davo_amd.davo import Engine
from davo_amd.comm import RcclComm
import numpy as np
_lib.lib()
from davo_amd.version import parse_version, FLAGSHIP_VERSION
e = Engine(parse_version(FLAGSHIP_VERSION), 128, 416, 4, device=0)
t1 = time.time()
c = RcclComm(e, 0, 1)
t2 = time.time()
out, ms = c.allgather(np.zeros((4, 2, 6), np.float32))
t3 = time.time()
c.close()
print("RESULT %%.3f %%.3f %%.3f" %% (t1 - t0, t2 - t1, t3 - t2), flush=True)
""" % ROOT

SETTINGS = [
    ("default", {}),
    ("msccl off", {"RCCL_MSCCL_ENABLE": "0", "RCCL_MSCCLPP_ENABLE": "0"}),
    ("no net probing", {"NCCL_IB_DISABLE": "1", "NCCL_NET_PLUGIN": "none", "NCCL_SOCKET_IFNAME": "lo"}),
    ("ras off", {"NCCL_RAS_ENABLE": "0"}),
    ("all of the above", {"RCCL_MSCCL_ENABLE": "0", "RCCL_MSCCLPP_ENABLE": "0", "NCCL_IB_DISABLE": "1", "NCCL_NET_PLUGIN": "none",
                          "NCCL_SOCKET_IFNAME": "lo", "NCCL_RAS_ENABLE": "0"}),
    ("all + 1 channel", {"RCCL_MSCCL_ENABLE": "0", "RCCL_MSCCLPP_ENABLE": "0", "NCCL_IB_DISABLE": "1", "NCCL_NET_PLUGIN": "none",
                         "NCCL_SOCKET_IFNAME": "lo", "NCCL_RAS_ENABLE": "0", "NCCL_MAX_NCHANNELS": "2", "NCCL_MIN_NCHANNELS": "1"}),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rows = []
    for name, env_add in SETTINGS:
        res = []
        for _ in range(reps):
            env = dict(os.environ)
            env.update(env_add)
            with tempfile.TemporaryDirectory() as d:
                env["DAVO_COMM_DIR"] = d
                p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=120)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
            if p.returncode != 0 or not line:
                res.append(("failed", p.stderr[-300:]))
            else:
                res.append(tuple(float(x) for x in line[0].split()[1:]))
        rows.append({"setting": name, "env": env_add, "context_s|comm_init_s|first_allgather_s": res})
        print(name, res, flush=True)
    if len(sys.argv) > 2:
        json.dump(rows, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
