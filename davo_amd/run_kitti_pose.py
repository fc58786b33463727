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

_T_IMPORT = time.time()          # before numpy and the package: the first mark of the start-up split (--report)

import numpy as np               # noqa: E402

from . import sequence as S              # noqa: E402
from .davo import DAVO                   # noqa: E402
from .version import FLAGSHIP_VERSION    # noqa: E402


_FIRST_MAIN = True


def _process_start_time():
    """wall-clock time this process was created: its age from /proc (start time in clock ticks since boot against the uptime, 10 ms
    resolution - btime in /proc/stat is whole seconds and put up to a second of error into round 5's first start-up splits)"""
    try:
        ticks = int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[19])
        uptime = float(open("/proc/uptime").read().split()[0])
        return time.time() - (uptime - ticks / os.sysconf("SC_CLK_TCK"))
    except (OSError, ValueError, IndexError):
        return None


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
    ap.add_argument("--calibrate_on_first_windows", action="store_true",
                    help="calibrate every rank on windows 0..7 of the sequence (loaded inline) instead of on its own first batch: the storage "
                         "scales, and so the trajectory's last bits, then do not depend on the number of GPUs")
    ap.add_argument("--sync_driver", action="store_true",
                    help="one synchronous davo_forward per batch (input wait + copy + kernels + pose copy add up) instead of the streaming "
                         "entry point (davo_submit: three batches in flight (four up to batch 2), copies and input wait overlapped with the kernels)")
    a = ap.parse_args(argv)
    # start-up split (wall clock).  Only the first main() of a process can say what the process start cost
    global _FIRST_MAIN
    fresh, _FIRST_MAIN = _FIRST_MAIN, False
    marks = [("interpreter_up", _T_IMPORT), ("imports_done", time.time())] if fresh else [("main_entered", time.time())]

    def mark(name):
        marks.append((name, time.time()))

    from .comm import RcclComm, world_from_env, preload_in_background
    rank, local_rank, world = world_from_env()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the parent only starts the ranks and waits; it never touches a GPU
        from .launch import spawn_ranks
        raise SystemExit(spawn_ranks(["-m", "davo_amd.run_kitti_pose"] + list(sys.argv[1:] if argv is None else argv), a.gpus))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    device_index = local_rank
    emulate = tuple(int(x) for x in a.emulate_shard.split("/")) if a.emulate_shard else None
    need_comm = world > 1 or a.force_comm
    # Start-up is a rank's whole run on a sharded sequence (a 568-window shard is 0.1 s of work), so everything that does not depend
    # on each other starts at once: librccl loads on a thread of its own (573 MB to map: most of a second, needed only at the
    # gather), the input pipeline's workers fork and its buffers are created and page-locked on another, while this thread reads
    # the checkpoint and builds the GPU context.  (Round 4 did these one after the other: 2.9 s from process start to the first
    # batch in a fresh process, 1.7 s of it the communicator: profiles/r05b_config4_scene_streamed.json.)
    if need_comm:
        preload_in_background()
    import threading
    loader_thread = None
    gpu_ready = threading.Event()
    H, W = a.img_height, a.img_width
    load = weights = None
    if a.synthetic:
        n_frames = a.synthetic
    else:
        if a.loader_procs != 0:
            # the fork server the loader's workers come from starts importing numpy / Pillow now (davo_amd/loader.py: worker_context);
            # it is a spawned interpreter that never touches a GPU, and its 0.2 s of imports pass behind HIP's initialisation
            from .loader import warm_workers
            warm_workers()
        d = os.path.join(a.concat_img_dir, "%.2d" % a.test_seq)
        n_frames = sum(1 for f in os.listdir(d) if f.endswith(".jpg")) + 2 * int((a.seq_length - 1) / 2)      # test_kitti_pose.py:81-82
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
                                     pin=lambda arr: (gpu_ready.wait(), pin_array(arr, device_index)), unpin=unpin_array,
                                     seg_planes=(0, 1, 2) if static_all else None, hold=0 if a.sync_driver else 1)
        # the loader's buffers are created and its workers start filling them on a thread of its own, before anything else: the
        # workers need no GPU.  Page-locking the buffers (1.4 GB at batch 64: 0.3 s) follows on the loader's own thread, entry by
        # entry, once the context below exists: HIP serialises hipHostRegister with the context's own allocations, and pinning
        # beside them made the context 0.25 s slower (profiles/r05an_config4_scene_shard_nocomm.json against r05ao).
        shard = S.shard_windows(n_frames - 2, *((world, rank) if emulate is None else (emulate[1], emulate[0])))
        loader_thread = threading.Thread(target=load.prestart, args=(shard[0], shard[1], a.batch_size), name="davo-loader-start")
        loader_thread.start()
        # ... and the checkpoint is read and parsed (0.06-0.1 s of file and numpy work, no GPU) on a third
        from .tf_checkpoint import load_weights
        ckpt = {}

        def read_checkpoint():
            try:
                ckpt["weights"] = load_weights(a.ckpt_file)        # TF V2 checkpoint (prefix / .index / directory) or .npz
            except BaseException as exc:                           # noqa: BLE001 - re-raised by the main thread below
                ckpt["error"] = exc
        weights_thread = threading.Thread(target=read_checkpoint, name="davo-checkpoint-read")
        weights_thread.start()
    # the GPU context: the communicator's thread needs it, and HIP's own initialisation (0.2-0.4 s) is on every path
    from . import _lib
    try:
        _lib.lib()
        mark("library_loaded")
        system = DAVO(version=a.version, device=device_index)
        system.setup_inference(H, W, "davo", a.seq_length, a.batch_size)
        mark("gpu_context_created")
    finally:
        gpu_ready.set()                        # also on failure: the loader's pinning thread must not wait for ever
    # the communicator is not needed before the gather: the id exchange and ncclCommInitRank (1.5-1.7 s, most of it inside HIP's
    # code-object loading, which other HIP calls queue behind) run on a second thread from here on (collective; fails loudly at the
    # gather, no other transport)
    comm = RcclComm.from_env_async(system.engine) if need_comm else None

    if a.synthetic:
        from . import synth
        load = S.synthetic_window_loader(H, W)
        weights = synth.make_weights(a.version)
    else:
        weights_thread.join()
        if "error" in ckpt:
            raise ckpt["error"]
        weights = ckpt["weights"]
    mark("inputs_and_weights_ready")
    system.load_weights(weights)
    mark("weights_on_gpu")
    infer = lambda img, flow, seg: system.inference(None, "pose", inputs=(img, flow, seg))["pose"]   # noqa: E731
    if loader_thread is not None:
        loader_thread.join()
    mark("loader_started")
    lo_hi = S.shard_windows(n_frames - 2, *((world, rank) if emulate is None else (emulate[1], emulate[0])))
    ld = load.for_range(*lo_hi, a.batch_size) if hasattr(load, "for_range") else load
    batches = ld
    if not a.no_calibrate:
        if a.calibrate_on_first_windows or a.synthetic or not hasattr(ld, "__iter__") or lo_hi[0] >= lo_hi[1]:
            # windows 0..7 of the sequence on every rank: the storage scales - and with them the trajectory's last bits - do not
            # depend on the world size.  Loaded inline: Pillow's import and eight decodes in front of this rank's first batch
            system.calibrate(load(0, min(a.batch_size, 8, n_frames - 2)))
        else:
            # the first eight windows of THIS rank's first batch, which its loader's workers are decoding anyway (round 5: the inline
            # load was 0.35 s of a rank's 0.8 s start-up, most of it importing Pillow into this process).  The scales are exact
            # powers of two with 64x headroom: ranks that calibrate on different windows agree to float32 rounding
            # (test_calibration_is_neutral_for_a_well_ranged_checkpoint), not to the bit: --calibrate_on_first_windows restores that
            import itertools
            it = iter(ld)
            first = next(it)
            n8 = min(8, first[1] - first[0])
            system.calibrate(tuple(x[:n8] for x in first[2]))
            batches = itertools.chain([first], it)
    mark("calibrated_first_forward_done")
    # streamed: a batch of the process loader stays valid while the next one is asked for (hold = 1), so davo_submit does not wait for
    # its own copy; the threaded loader and the synthetic windows give no such promise (hold = 0)
    from .loader import ProcessWindowLoader
    stream_hold = lambda ld: 1 if isinstance(ld, ProcessWindowLoader) else 0      # noqa: E731
    t0 = time.perf_counter()
    timing = {}
    # this rank's prefetching loader (started above): handed in ready-made and closed only after the trajectory is written -
    # unpinning and unmapping ~1 GB of batch buffers takes 0.15 s and used to run inside run_sequence when the last reference died
    stream = None if a.sync_driver else S.PoseStream(system.engine, hold=stream_hold(ld))
    traj, poses = S.run_sequence(infer, batches, n_frames, a.batch_size, rank, world, comm, timing, emulate, stream)
    dt = time.perf_counter() - t0
    if rank == 0:
        os.makedirs(a.output_dir, exist_ok=True)
        out = os.path.join(a.output_dir, "%.2d-pred_kitti_pose.txt" % a.test_seq)   # :116
        tw = time.perf_counter()
        S.write_kitti_poses(out, traj)
        timing["write_s"] = time.perf_counter() - tw
        t_proc = _process_start_time()
        names = [m[0] for m in marks]
        times = [m[1] for m in marks]
        startup = {"%s_s" % names[i]: times[i] - times[i - 1] for i in range(1, len(marks))}
        if t_proc is not None and fresh:
            startup["process_start_to_interpreter_up_s"] = times[0] - t_proc
            startup["process_start_to_first_batch_s"] = times[-1] - t_proc
        startup["first_batch_to_trajectory_written_s"] = time.time() - times[-1]
        if getattr(comm, "t_ready", None) is not None and t_proc is not None and fresh:
            startup["process_start_to_communicator_ready_s"] = comm.t_ready - t_proc      # built on a second thread; the gather waited for it
            startup["process_start_to_trajectory_written_s"] = time.time() - t_proc
        timing["startup"] = {k: round(v, 4) for k, v in startup.items()}
        timing.update(total_s=dt + timing["write_s"], windows=n_frames - 2, world=world, batch_size=a.batch_size,
                      windows_per_s=(n_frames - 2) / dt, range_recovery=system.engine.range_stats(),
                      note="rank 0's seconds; load_wait_s = time the GPU side waited for the input pipeline, forward_s = H2D + kernels + "
                           "pose D2H inside DAVO.inference (streamed: the time inside davo_submit), gather_s = the RCCL all-gather incl. staging and the "
                           "wait for the communicator, which is built on a second thread from the moment the GPU context exists")
        print("Done. Please check %s  (%d windows on %d GPU(s) in %.2f s incl. input generation/IO: input wait %.2f, forward %.2f, "
              "gather %.3f, stitch %.2f, write %.2f%s)" % (out, n_frames - 2, world, dt, timing["load_wait_s"], timing["forward_s"],
                                                           timing["gather_s"], timing["stitch_s"], timing["write_s"],
                                                           "; streamed: forward = time inside davo_submit, drain %.3f" % timing["drain_s"]
                                                           if "drain_s" in timing else ""))
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
