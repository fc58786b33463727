"""Sequence inference driver — the counterpart of the reference's ``test_kitti_pose.py``
(flags ``:20-29``, main ``:75-154``): runs the pose path over a KITTI odometry sequence and
writes ``<seq>-pred_kitti_pose.txt``.

    python -m davo_amd.run_kitti_pose --test_seq 3 --concat_img_dir DUMP --ckpt_file W.npz \
        --output_dir out --version v1-...                    # one GPU
    python -m davo_amd.run_kitti_pose ... --batch_size 64 --gpus 8     # windows sharded over 8 GPUs

With ``--gpus N`` the command starts N ranks of itself, one per GPU (davo_amd/launch.py; a launcher that
sets RANK / LOCAL_RANK / WORLD_SIZE itself, e.g. ``python -m torch.distributed.run``, works too); the ranks'
poses meet in one RCCL all-gather (davo_amd/comm.py) and rank 0 writes the trajectory.

``--ckpt_file`` is a TF V2 checkpoint as the reference's Saver wrote it (prefix, ``.index`` file or the
directory holding ``checkpoint``; davo_amd/tf_checkpoint.py reads it without TensorFlow) or an ``.npz``
keyed by the TF variable names (SURVEY table W); with
``--synthetic N`` the inputs and weights are the seeded synthetic ones (no KITTI dump or
checkpoint exists offline) and N is the frame count (801 = seq 03, 4541 = seq 00).
"""
import argparse
import os
import sys
import time

import numpy as np

from . import sequence as S
from .davo import DAVO
from .version import FLAGSHIP_VERSION


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=1)          # test_kitti_pose.py:21
    ap.add_argument("--img_height", type=int, default=128)
    ap.add_argument("--img_width", type=int, default=416)
    ap.add_argument("--seq_length", type=int, default=3)
    ap.add_argument("--test_seq", type=int, default=9)
    ap.add_argument("--concat_img_dir", default=None)
    ap.add_argument("--output_dir", required=True)
    ap.add_argument("--ckpt_file", default=None)
    ap.add_argument("--version", default=FLAGSHIP_VERSION)
    ap.add_argument("--synthetic", type=int, default=0, help="frame count of a synthetic sequence")
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node to shard the windows over (one rank per GPU)")
    ap.add_argument("--no_calibrate", action="store_true",
                    help="skip the activation-range calibration of the f16x3 arithmetic (include/davo_hip.h: davo_calibrate)")
    ap.add_argument("--loader_threads", type=int, default=4, help="decode/read threads of the input pipeline (as data_loader.py:283-288; more threads contend on the GIL)")
    ap.add_argument("--decode_procs", type=int, default=0,
                    help="(threaded loader) extra JPEG decode processes; 0 = decode in the loader threads")
    ap.add_argument("--loader_procs", type=int, default=-1,
                    help="worker processes that decode the strips and read the .npy planes straight into shared, page-locked batch "
                         "buffers (davo_amd/loader.py: ProcessWindowLoader).  -1 = this rank's CPU share minus two, at most 16; "
                         "0 = the threaded loader (--loader_threads)")
    ap.add_argument("--force_comm", action="store_true",
                    help="build the RCCL communicator and run the pose all-gather at world size 1 too (exercises the multi-GPU path on one GPU)")
    ap.add_argument("--emulate_shard", default=None, metavar="r/R",
                    help="measurement aid: do what rank r of R would do (its window shard, the gather, the whole stitch) in this one process")
    ap.add_argument("--report", default=None, help="write the run's time split (load wait / forward / gather / stitch / write) as JSON here")
    a = ap.parse_args(argv)

    from .comm import RcclComm, world_from_env
    rank, local_rank, world = world_from_env()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the parent only starts the ranks and waits; it never touches a GPU
        from .launch import spawn_ranks
        raise SystemExit(spawn_ranks(["-m", "davo_amd.run_kitti_pose"] + list(sys.argv[1:] if argv is None else argv), a.gpus))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    device_index = local_rank
    emulate = tuple(int(x) for x in a.emulate_shard.split("/")) if a.emulate_shard else None
    if not a.synthetic and a.loader_procs != 0:
        # first thing this rank does: the fork server the loader's workers come from starts importing numpy / Pillow now, behind
        # this process's own imports, the checkpoint read and the GPU set-up (davo_amd/loader.py: worker_context)
        from .loader import warm_workers
        warm_workers()

    H, W = a.img_height, a.img_width
    if a.synthetic:
        from . import synth
        n_frames = a.synthetic
        load = S.synthetic_window_loader(H, W)
        weights = synth.make_weights(a.version)
    else:
        from glob import glob
        d = os.path.join(a.concat_img_dir, "%.2d" % a.test_seq)
        n_frames = len(glob(d + "/*.jpg")) + 2 * int((a.seq_length - 1) / 2)      # test_kitti_pose.py:81-82
        from .davo import pinned_empty, pin_array, unpin_array    # batches are decoded straight into page-locked memory
        procs = a.loader_procs
        if procs < 0:
            cores = len(os.sched_getaffinity(0))
            try:
                q, period = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    cores = min(cores, max(1, int(int(q) / int(period))))
            except (OSError, ValueError):
                pass
            procs = max(1, min(16, cores - 2))
        from .version import parse_version
        static_all = parse_version(a.version).att_source == "static_all"   # only -segmask_all-static reads the target frame's label map
        load = S.kitti_window_loader(a.concat_img_dir, a.test_seq, n_frames, H, W,
                                     alloc=lambda shape, dtype: pinned_empty(shape, dtype, device_index),
                                     workers=a.loader_threads, decode_procs=a.decode_procs, procs=procs,
                                     pin=lambda arr: pin_array(arr, device_index), unpin=unpin_array,
                                     seg_planes=(0, 1, 2) if static_all else None)
        # the loader's workers come up (spawn + imports: ~0.5 s) while the checkpoint is read and the GPU context is built
        load.prestart(*S.shard_windows(n_frames - 2, *((world, rank) if emulate is None else (emulate[1], emulate[0]))), a.batch_size)
        from .tf_checkpoint import load_weights
        weights = load_weights(a.ckpt_file)        # TF V2 checkpoint (prefix / .index / directory) or .npz

    system = DAVO(version=a.version, device=device_index)
    system.load_weights(weights)
    system.setup_inference(H, W, "davo", a.seq_length, a.batch_size)
    infer = lambda img, flow, seg: system.inference(None, "pose", inputs=(img, flow, seg))["pose"]   # noqa: E731
    if not a.no_calibrate:
        # every rank calibrates on the same first windows, so the trajectory does not depend on the world size
        system.calibrate(load(0, min(a.batch_size, n_frames - 2)))

    comm = RcclComm.from_env(system.engine) if (world > 1 or a.force_comm) else None      # collective; fails loudly, no other transport
    t0 = time.perf_counter()
    timing = {}
    # this rank's prefetching loader (started above): handed in ready-made and closed only after the trajectory is written -
    # unpinning and unmapping ~1 GB of batch buffers takes 0.15 s and used to run inside run_sequence when the last reference died
    lo_hi = S.shard_windows(n_frames - 2, *((world, rank) if emulate is None else (emulate[1], emulate[0])))
    ld = load.for_range(*lo_hi, a.batch_size) if hasattr(load, "for_range") else load
    traj, poses = S.run_sequence(infer, ld, n_frames, a.batch_size, rank, world, comm, timing, emulate)
    dt = time.perf_counter() - t0
    if rank == 0:
        os.makedirs(a.output_dir, exist_ok=True)
        out = os.path.join(a.output_dir, "%.2d-pred_kitti_pose.txt" % a.test_seq)   # :116
        tw = time.perf_counter()
        S.write_kitti_poses(out, traj)
        timing["write_s"] = time.perf_counter() - tw
        timing.update(total_s=dt + timing["write_s"], windows=n_frames - 2, world=world, batch_size=a.batch_size,
                      windows_per_s=(n_frames - 2) / dt, range_recovery=system.engine.range_stats(),
                      note="rank 0's seconds; load_wait_s = time the GPU side waited for the input pipeline, forward_s = H2D + kernels + "
                           "pose D2H inside DAVO.inference, gather_s = the RCCL all-gather incl. staging")
        print("Done. Please check %s  (%d windows on %d GPU(s) in %.2f s incl. input generation/IO: input wait %.2f, forward %.2f, "
              "gather %.3f, stitch %.2f, write %.2f)" % (out, n_frames - 2, world, dt, timing["load_wait_s"], timing["forward_s"],
                                                         timing["gather_s"], timing["stitch_s"], timing["write_s"]))
        if a.report:
            import json
            with open(a.report, "w") as f:
                json.dump({k: (round(v, 4) if isinstance(v, float) else v) for k, v in timing.items()}, f)
    if hasattr(ld, "close"):
        ld.close()
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
