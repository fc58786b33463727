#!/usr/bin/env python
"""bench.py — pose-net triplets/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the hot path (SE squeeze + excite, mask/pack, 8 conv layers, pose head)
over one batch of synthetic triplets that is already resident in HBM.  At N=1 the workload is
BASELINE.json configs[1]: batch 32, 128x416, flagship variant
(dilatedPoseNN-cnv6_128 + se_flow + fc_tanh).  For N>1 every rank (one process per GPU) runs the
same per-GPU batch on its own windows (weak scaling, no data-path collective: windows are
independent, test_kitti_pose.py:134-145); after the timed region the ranks' poses meet once in an
RCCL all-gather (davo_amd/comm.py), reported separately and not part of `value`.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--settle S] [--batch B] [--height H --width W] [--force-comm]

Untimed before the timed region: W warm-up steps, then S settle steps (default: whatever brings W + S to 100) — the
chip's clock needs ~25 steps (30 ms) of this load to settle after idle.  The JSON's `warmup` is the number of untimed
steps that really ran (W + S; `warmup_arg` and `settle_steps` beside it) and the timed region is exactly K full forwards.  `timing` in the JSON shows the ramp that is left: per-step periods
and the dominant launch's durations at the start and at the end of the timed region.

`--gpus N` with N > 1 starts the N ranks itself (davo_amd/launch.py: a parent that touches no GPU and
returns non-zero if any rank does); started under `python -m torch.distributed.run --nproc-per-node N`
the script finds RANK / LOCAL_RANK / WORLD_SIZE already set and runs as that rank.  The timed region is
bracketed on both sides by davo_comm_barrier (every stream of the rank's context drained, then an RCCL
all-reduce rendezvous) — the launch contract's barrier + device synchronize without PyTorch — and
`value` uses the MAX elapsed time over the ranks (RCCL all-reduce).  A failed RCCL init or collective
is a non-zero exit: there is no other transport.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from davo_amd.launch import bind_rank_cpus      # noqa: E402  (standard library only)
BOUND_CPUS = bind_rank_cpus()      # a rank started by davo_amd.launch binds itself to its CPU slice before numpy starts a thread

import numpy as np      # noqa: E402

# FLOPs of one PoseNN pair evaluation at 128x416, SURVEY.md §8a table L (MACs x 2)
MACS_PER_PAIR_128x416 = 3890085888
CNV6_MACS_PER_PAIR_128x416 = 2 * 981467136          # rotation + translation cnv6, fused into one launch
PEAK_F32_MFMA_TFLOPS = 157.3                         # MI355X_MICROARCH.md: FP32 matrix, dense
PEAK_F16_MFMA_TFLOPS = 2500.0                        # MI355X_MICROARCH.md: BF16/FP16 matrix, dense
PEAK_HBM_GBS = 8000.0
# compulsory HBM bytes of the prologue kernels per triplet at 128x416: se_squeeze reads the two flow
# planes it reduces; mask_pack reads the u8 strip + 2 flow planes + 2 seg planes and writes the packed PoseNN input
# (2 pair images x 8 channels x 4 B)
SQUEEZE_BYTES_128x416 = 851968
PACK_BYTES_128x416 = 479232 + 851968 + 425984 + 2 * 128 * 416 * 32


def usable_cores():
    """host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(period)
            n = max(1, min(n, int(quota)))
    except (OSError, ValueError):
        pass
    return n, quota


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def workload_name(B, H, W, world):
    """the BASELINE.json configuration the arguments select (or none of them)"""
    net = "dilatedPoseNN-cnv6_128 + se_flow + fc_tanh (segmask_all + abs_flow)"
    shape = "batch=%d/GPU synthetic %dx%d RGB+flow+seg triplets, %s" % (B, H, W, net)
    if (B, H, W) == (32, 128, 416):
        return "BASELINE.json configs[1]: single MI355X, batch=32 synthetic 128x416 triplets" + (
            "" if world == 1 else " (x%d window-sharded replicas)" % world) + ", " + net
    if (B, H, W) == (128, 128, 416):
        return "BASELINE.json configs[2]: single MI355X, batch=128, same net" + ("" if world == 1 else " (x%d replicas)" % world)
    if (B, H, W) == (64, 256, 832):
        return "BASELINE.json configs[4] per-GPU shape: 256x832, batch=64/GPU, segmask_all + abs_flow (%d GPU%s)" % (world, "" if world == 1 else "s")
    return "none of BASELINE.json's configs: " + shape


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # Defaults: the chip's clock takes ~25 steps (30 ms) to settle after idle - kernel durations fall 10 % over them
    # (profiles/: per-launch trace of the dominant kernel) - and then holds (W = 50 ... 3000 measure the same to 0.7 %,
    # DESIGN.md section 6a); 5 warm-up steps time the ramp, not the path.
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--settle", type=int, default=-1,
                    help="untimed steps after the warm-up, before the timed region (clock settle; disclosed as settle_steps). "
                         "Default -1: max(0, 100 - warmup)")
    ap.add_argument("--force-comm", action="store_true",
                    help="build the RCCL communicator at world size 1 too, so the barrier / max all-reduce / gather code of the "
                         "multi-GPU run executes on one GPU")
    ap.add_argument("--batch", type=int, default=32, help="triplets per GPU per step")
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic windows generated per rank (tiled to the batch); 0 = the whole batch distinct")
    ap.add_argument("--precision", choices=["f16x3", "f32"], default="f32",
                    help="arithmetic of the top-level value / roofline.  f32 (default): FP32 MFMA, bit-exact fmaf chains - the reference's own "
                         "arithmetic (slim.conv2d in float32, nets/posenn.py:205-215); f16x3: split-fp16 MFMA, float32-grade, the library's "
                         "default mode.  The other one is measured in the same run and nested (fast_mode / reference_arithmetic)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches kept in flight on the GPU in the timed region (davo_set_inflight).  Default 1: one "
                         "batch at a time, so the event-bracketed kernel durations are the kernels' own")
    ap.add_argument("--cu-partition", action="store_true",
                    help="with --inflight n: slot i's stream is CU-masked to its own 1/n of every XCD's CUs (davo_set_option cu_partition)")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra 2-in-flight throughput measurement (used for the rocprofv3 passes, so that "
                         "per-kernel statistics are not mixed with overlapped launches)")
    ap.add_argument("--no-f32", "--no-second-leg", dest="no_f32", action="store_true",
                    help="skip the leg in the OTHER arithmetic (fast_mode when the top level is float32, reference_arithmetic when it is f16x3)")
    ap.add_argument("--profile-stride", type=int, default=0,
                    help="HIP events around the dominant kernel of every n-th timed step (default: 4 when steps >= 40, else 1)")
    ap.add_argument("--options", default="", help="davo_set_option pairs applied before the first step, k=v,k=v (A/B of launch plans under the profiler)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8, help="triplets per CPU-baseline pass")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # parent: start one rank per GPU and wait; this process never touches a GPU
        from davo_amd.launch import spawn_ranks
        raise SystemExit(spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    from davo_amd.comm import RcclComm, world_from_env
    rank, local_rank, world = world_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    device_index = local_rank                        # one rank per GPU

    from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION
    cfg = parse_version(FLAGSHIP_VERSION)
    B, H, W = args.batch, args.height, args.width
    scale_px = (H * W) / float(128 * 416)
    flops_per_triplet = 2 * 2 * MACS_PER_PAIR_128x416 * scale_px
    cnv6_flops_per_launch = 2 * CNV6_MACS_PER_PAIR_128x416 * scale_px * (2 * B)

    weights = synth.make_weights(cfg)
    eng = Engine(cfg, H, W, B, device=device_index)  # raises without a GPU: the HIP path has no CPU fallback
    eng.load_weights(weights)
    eng.set_precision(args.precision)
    for kv in filter(None, args.options.split(",")):
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    # raises on any RCCL failure: no gloo, no TCP stand-in
    comm = RcclComm(eng, rank, world) if (world > 1 or args.force_comm) else None

    # synthetic windows of this rank's shard, resident in HBM before the timed region; one buffer set per
    # in-flight slot (consecutive steps work on different batches of the shard)
    nu = B if args.unique <= 0 else max(1, min(args.unique, B))
    nset = max(1, args.inflight)
    sets = []
    for k in range(nset):
        img_u, flow_u, seg_u = synth.make_inputs(nu, H, W, first_window=(rank * nset + k) * B)
        reps = -(-B // nu)
        img_k = np.tile(img_u, (reps, 1, 1, 1))[:B]
        flow_k = np.tile(flow_u, (reps, 1, 1, 1, 1))[:B]
        seg_k = np.tile(seg_u, (reps, 1, 1, 1, 1))[:B]
        sets.append((eng.alloc(img_k.nbytes).upload(img_k), eng.alloc(flow_k.nbytes).upload(flow_k),
                     eng.alloc(seg_k.nbytes).upload(seg_k), eng.alloc(B * 12 * 4)))
        if k == 0:
            img, flow, seg = img_k, flow_k, seg_k
    d_img, d_flow, d_seg, d_pose = sets[0]
    eng.set_inflight(nset)
    if args.cu_partition:
        eng.set_option("cu_partition", 1)

    def sync_all():
        """barrier + device synchronize: every stream of this rank's context idle (and its f16x3 range record
        judged: DavoRangeError here means a batch left the fp16-pair range), then all ranks rendezvous"""
        eng.synchronize()
        if comm is not None:
            comm.barrier()

    def max_over_ranks(x):
        return comm.allreduce(x, "max") if comm is not None else x

    def timed(run_steps, steps):
        sync_all()
        t0 = time.perf_counter()
        run_steps(steps)
        sync_all()
        return max_over_ranks(time.perf_counter() - t0)

    settle = args.settle if args.settle >= 0 else max(0, 100 - args.warmup)
    for i in range(args.warmup + settle):
        eng.forward_device(B, *sets[i % nset])
    eng.synchronize()

    # timed region: K steps; only the dominant kernel (main cnv6 launch) is bracketed by HIP events
    # on the launch stream, so the event records do not perturb the other launches
    # (every 4th step: an event pair puts two ~6 us bubbles around the launch it brackets - 1 % of the step if every launch is timed)
    stride = args.profile_stride or (4 if args.steps >= 40 else 1)
    eng.profile(2)
    eng.set_option("profile_stride", stride)
    eng.profile_reset()
    elapsed_max = timed(lambda k: [eng.forward_device(B, *sets[i % nset]) for i in range(k)], args.steps)
    dominant = eng.profile_entries()
    dom_dur, dom_period = eng.profile_samples("cnv6")

    def head_tail(x):
        x = [float(v) for v in x]
        k = max(1, min(5, len(x) // 2))
        return {"n": len(x), "first%d_mean" % k: round(sum(x[:k]) / k, 4), "last%d_mean" % k: round(sum(x[-k:]) / k, 4),
                "min": round(min(x), 4), "max": round(max(x), 4)} if x else None
    # the ramp that is left inside the timed region, from the events that bracket the dominant launch anyway: a bracketed
    # launch's own duration, and the stream time between the starts of consecutive bracketed launches (= stride steps)
    timing = {"warmup_steps": args.warmup, "settle_steps": settle, "timed_steps": args.steps, "bracketed_every": stride,
              "dominant_launch_ms": head_tail(dom_dur),
              "step_ms_on_stream": head_tail([p / stride for p in dom_period if p >= 0]),
              "note": "untimed = warmup + settle (clock ramp after idle, HISTORY.md round 2); the timed region is exactly `steps` full "
                      "forwards; step_ms_on_stream = time between the starts of consecutive bracketed cnv6 launches / stride"}
    plan6 = eng.last_plan(5)                          # [(128-row M tiles, N tile | f16x3 tile id), ...]
    # untimed extra pass, one batch in flight: per-kernel breakdown (every launch bracketed)
    eng.set_inflight(1)
    eng.profile(1)
    eng.profile_reset()
    for _ in range(max(3, args.steps // 4)):
        eng.forward_device(B, d_img, d_flow, d_seg, d_pose)
    kernels = eng.profile_entries()
    eng.profile(False)
    eng.synchronize()
    poses = d_pose.download((B, 2, 6))

    def roofline_block(precision, dom, plan, avg_key="cnv6"):
        total_mtiles = sum(m for m, _ in plan)
        flops_main = cnv6_flops_per_launch * plan[0][0] / total_mtiles
        n6, ms6 = dom.get(avg_key, (0, 0.0))
        avg6 = ms6 / max(n6, 1)
        achieved = flops_main / (avg6 * 1e-3) / 1e12 if avg6 > 0 else 0.0
        if precision == "f32":
            peak = PEAK_F32_MFMA_TFLOPS
            merged32 = len(plan) == 1 and H * W >= 128 * 416 and B >= 8
            kname = ("davo::conv_igemm_f32_mainrem<3,1,64,6> (cnv6 whole layer, one launch: whole rounds of 128x128 tiles, then the remainder's "
                     "128x64 tiles; rotation|translation fused, N=256, K=2304)" if merged32 else
                     "davo::conv_igemm_f32<3,1,128,6> (cnv6 main launch: rotation|translation fused, N=256, K=2304)")
            peak_note = "FP32 MFMA dense peak (v_mfma_f32_32x32x2_f32)"
            key = "conv_igemm_f32"
        else:
            # every algorithmic FLOP costs three fp16 MFMA FLOPs (hi*hi, hi*lo, lo*hi), so the
            # matrix-pipe roofline of this algorithm is the fp16 dense peak / 3
            peak = PEAK_F16_MFMA_TFLOPS / 3.0
            tiles = {0: "128x32", 1: "256x64", 2: "256x128", 3: "128x256", 4: "128x128", 5: "256x256", 6: "208x256",
                     7: "256x256 + 128x128 remainder tiles in one grid"}
            merged = plan[0][1] == 7
            # round 5: launches of whole 256x256 tiles run on conv_igemm_h3w (four waves of 128x128 outputs) unless "wave128" is 0
            wave128 = plan[0][1] == 5 and "wave128=0" not in (getattr(args, "options", "") or "").replace(" ", "")
            kname = ("davo::%s (cnv6 %s: rotation|translation fused, N=256, K=2304, %s tile, LDS-DMA staged, "
                     "v_mfma_f32_16x16x32_f16)" % ("conv_igemm_h3_mainrem<6,2>" if merged else
                                                   ("conv_igemm_h3w<6,2>" if wave128 else "conv_igemm_h3<3,1,...,6,true,false,true>"),
                                                   "whole layer, one launch" if merged else "main launch",
                                                   tiles.get(plan[0][1], "?") + (" on four waves of 128x128 outputs" if wave128 else "")))
            peak_note = ("fp16 MFMA dense peak 2500 TFLOP/s / 3 products per algorithmic FLOP.  What loops of nothing but this matrix "
                         "instruction sustain under the board power cap is measured in profiles/r04d_mfma_peak_probe{,2}.log and DESIGN.md section 2")
            key = "conv_igemm_h3_mainrem<6" if merged else ("conv_igemm_h3w<6" if wave128 else "conv_igemm_h3<3, 1, ")
        # HBM traffic of the dominant kernel: from the most recent committed PMC pass (profiles/), not live —
        # rocprofv3 counter collection cannot run inside the timed process
        traffic, traffic_src, busy = None, None, None
        try:
            import glob
            for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))):
                tj = json.load(open(f))
                if tj.get("batch", 32) != B or (H, W) != (128, 416):
                    continue
                hit = [v for k, v in tj["kernels"].items() if key in k and ((precision == "f32" and k.rstrip().endswith(", 6>")) or
                                                                              (precision != "f32" and ("mainrem" in key or "h3w<" in key or ", 6, true, false" in k)))]
                if hit:
                    hit.sort(key=lambda v: -(v["read_bytes"] + v["write_bytes"]))
                    traffic = hit[0]["read_bytes"] + hit[0]["write_bytes"]
                    busy = round(hit[0]["mfma_busy"], 4) if "mfma_busy" in hit[0] else None
                    traffic_src = "profiles/" + os.path.basename(f)
                    break
        except (OSError, ValueError, KeyError):
            pass
        return {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "mfma_busy": busy, "mfma_busy_source": traffic_src if busy is not None else None,
                "traffic": traffic, "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                "peak_note": peak_note, "avg_launch_ms": round(avg6, 4), "launches_timed": n6, "flops_per_launch": flops_main,
                "frac_note": "achieved = NOMINAL dense FLOPs of the launch (SURVEY 8d: padded taps included) / its duration.  The 3x3 kernels "
                             "skip filter rows that are zero padding for a whole tile, so frac can exceed the fraction of time the matrix "
                             "pipe is busy: mfma_busy (SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles, from the committed rocprofv3 PMC pass named "
                             "beside it, another box) is the utilisation",
                "launch_plan": "cnv6 as %s (mtiles of 128 rows, N tile)" % plan}

    # The OTHER arithmetic, in the same process on the same batch: same K steps, same barrier + synchronize bracketing, max over
    # ranks, its own roofline against the peak of the matrix instruction it uses, its own untimed run-in (the two modes load the chip
    # differently: float32 runs at full clock, f16x3 at the board power cap).  Top level float32 (default) -> `fast_mode` = f16x3,
    # the library's default mode; top level f16x3 -> `reference_arithmetic` = float32.
    other = "f16x3" if args.precision == "f32" else "f32"
    other_block = None
    other_kernels = None
    if not args.no_f32:
        eng.set_precision(other)
        for _ in range(args.warmup + settle if other == "f16x3" else max(2, (args.warmup + settle) // 10)):
            eng.forward_device(B, d_img, d_flow, d_seg, d_pose)
        eng.profile(2)
        eng.set_option("profile_stride", stride)
        eng.profile_reset()
        o_elapsed = timed(lambda k: [eng.forward_device(B, d_img, d_flow, d_seg, d_pose) for _ in range(k)], args.steps)
        dom_o = eng.profile_entries()
        plan_o = eng.last_plan(5)
        eng.profile(1)
        eng.profile_reset()
        for _ in range(max(3, args.steps // 4)):
            eng.forward_device(B, d_img, d_flow, d_seg, d_pose)
        other_kernels = {k: round(v[1] / max(v[0], 1), 4) for k, v in eng.profile_entries().items()}
        eng.profile(False)
        eng.synchronize()
        whole_o = flops_per_triplet * B * args.steps / o_elapsed / 1e12
        peak_o = PEAK_F32_MFMA_TFLOPS if other == "f32" else PEAK_F16_MFMA_TFLOPS / 3.0
        poses_o = d_pose.download((B, 2, 6))
        other_block = {"value": round(world * B * args.steps / o_elapsed, 2), "unit": "triplets/s",
                       "ms_per_step": round(o_elapsed / args.steps * 1e3, 4), "steps": args.steps,
                       "dtype": "f32" if other == "f32" else "f16x3 (fp16 hi/lo split operands, 3 MFMA products, f32 accumulate)",
                       "roofline": roofline_block(other, dom_o, plan_o),
                       "whole_path_tflops_per_gpu": round(whole_o, 2), "whole_path_frac_of_mfma_peak": round(whole_o / peak_o, 4),
                       "kernel_avg_ms": other_kernels,
                       "max_abs_diff_f16x3_vs_f32": float(np.abs(poses_o - poses).max()),
                       "note": ("davo_set_precision(1), the library's default: every float32 operand as an fp16 (hi, lo) pair, three "
                                "v_mfma_f32_16x16x32_f16 products per product, float32 accumulate - float32-grade poses (max_abs_diff beside "
                                "this note; the north star's tolerance is 1e-4), narrower arithmetic than the reference's by the letter"
                                if other == "f16x3" else
                                "davo_set_precision(0): v_mfma_f32_32x32x2_f32, bit-for-bit float32 fmaf chains - the reference's arithmetic")
                               + "; same batch, steps and bracketing as `value`"}
        eng.set_precision(args.precision)
        eng.forward_device(B, d_img, d_flow, d_seg, d_pose)      # d_pose holds the top-level mode's result again
        eng.synchronize()

    # extra, outside `value`: throughput with two batches in flight (the next batch's small kernels overlap this
    # batch's large convolutions; kernel wall durations are then shared time, so no roofline is quoted for it)
    pipelined = None
    if not args.no_pipelined and nset == 1:
        eng.set_precision("f16x3")                    # the library's default mode: what a streaming caller (davo_submit) runs
        img2, flow2, seg2 = synth.make_inputs(nu, H, W, first_window=(world + rank) * B)
        reps2 = -(-B // nu)
        set2 = (eng.alloc(img.nbytes).upload(np.tile(img2, (reps2, 1, 1, 1))[:B]),
                eng.alloc(flow.nbytes).upload(np.tile(flow2, (reps2, 1, 1, 1, 1))[:B]),
                eng.alloc(seg.nbytes).upload(np.tile(seg2, (reps2, 1, 1, 1, 1))[:B]), eng.alloc(B * 12 * 4))
        both = [sets[0], set2]
        eng.set_inflight(2)
        # the same untimed run-in as the main leg: this leg follows the float32 leg and the per-kernel pass, whose load is another
        # one (round 3 gave it 4 steps and timed it on the clock ramp: 26.07 k against 26.46 k for one in flight in BENCH_r03,
        # where an interleaved A/B in one process has two in flight ahead in 10 rounds of 10, +5.5 %: profiles/r04_pipelined_ab.log)
        for i in range(args.warmup + settle):
            eng.forward_device(B, *both[i % 2])
        pdt = timed(lambda k: [eng.forward_device(B, *both[i % 2]) for i in range(k)], args.steps)
        pipelined = {"batches_in_flight": 2, "dtype": "f16x3", "value": round(world * B * args.steps / pdt, 2), "unit": "triplets/s",
                     "ms_per_step": round(pdt / args.steps * 1e3, 4), "untimed_steps_before": args.warmup + settle,
                     "note": "davo_set_inflight(ctx, 2); same K steps, same untimed run-in, barrier + synchronize on both sides, max over "
                             "ranks; the next batch's small kernels overlap this batch's large convolutions"}
        eng.set_inflight(1)
        eng.set_precision(args.precision)
        eng.forward_device(B, *sets[0])
        eng.synchronize()
        sets.append(set2)

    # one all-gather of the shard poses (config 4's stitch input) over RCCL, outside the timed region
    gather = None
    if comm is not None:
        comm.barrier()
        g0 = time.perf_counter()
        allp, coll_ms = comm.allgather(poses)
        wall_ms = (time.perf_counter() - g0) * 1e3
        if allp.shape != (world * B, 2, 6) or not np.array_equal(allp[rank * B:(rank + 1) * B], poses):
            raise SystemExit("rank %d: the RCCL all-gather did not return this rank's own poses" % rank)
        # every rank ran the same weights on its own windows: the shards must all be finite and (world > 1) differ
        n_distinct = len({allp[r * B:(r + 1) * B].tobytes() for r in range(world)})
        if not np.isfinite(allp).all() or n_distinct != world:
            raise SystemExit("rank %d: gathered shards are not %d distinct finite pose blocks" % (rank, world))
        gather = {"backend": "rccl (librccl via davo_allgather_poses)", "ranks": world, "bytes_per_rank": int(poses.nbytes),
                  "collective_ms": round(coll_ms, 4), "wall_ms_incl_staging": round(wall_ms, 3),
                  "distinct_shards": n_distinct, "forced_at_world_1": bool(args.force_comm and world == 1)}

    if rank == 0:
        value = world * B * args.steps / elapsed_max
        kern_ms = {k: round(v[1] / max(v[0], 1), 4) for k, v in kernels.items()}
        whole = flops_per_triplet * B * args.steps / elapsed_max / 1e12
        roof = roofline_block(args.precision, dominant, plan6)
        roof["whole_path_frac_of_mfma_peak"] = round(whole / roof["peak"], 4)
        if other_block is not None:
            # flat copies inside `roofline` (a reader that keeps only this object's scalars still sees both arithmetics)
            tag = "fast_mode" if other == "f16x3" else "reference_arithmetic"
            roof.update({tag + "_value": other_block["value"], tag + "_ms_per_step": other_block["ms_per_step"],
                         tag + "_frac": other_block["roofline"]["frac"], tag + "_mfma_busy": other_block["roofline"]["mfma_busy"],
                         tag + "_whole_path_frac_of_mfma_peak": other_block["whole_path_frac_of_mfma_peak"],
                         tag + "_dtype": other, tag + "_max_abs_diff_vs_top_level": other_block["max_abs_diff_f16x3_vs_f32"]})
        dtype = ("f32 (v_mfma_f32_32x32x2_f32: the reference's arithmetic, slim.conv2d in float32)" if args.precision == "f32"
                 else "f16x3 (fp16 hi/lo split operands, 3 MFMA products, f32 accumulate)")
        # the HBM-bound front of the path: algorithmic bytes / kernel time from the per-kernel breakdown pass
        sq_ms, mp_ms = kern_ms.get("se_squeeze_partial", 0.0), kern_ms.get("mask_pack", 0.0)
        pro_bytes = (SQUEEZE_BYTES_128x416 + PACK_BYTES_128x416) * scale_px * B
        pro = None
        if sq_ms > 0 and mp_ms > 0:
            gbs = pro_bytes / ((sq_ms + mp_ms) * 1e-3) / 1e9
            pro = {"bound": "hbm", "kernels": "se_squeeze_partial + mask_pack<%d>" % (8 if args.precision == "f32" else 16), "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                   "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "bytes_per_step": int(pro_bytes),
                   "ms": round(sq_ms + mp_ms, 4),
                   "parts": {"se_squeeze_partial": {"bytes": int(SQUEEZE_BYTES_128x416 * scale_px * B), "ms": sq_ms,
                                                    "GB/s": round(SQUEEZE_BYTES_128x416 * scale_px * B / (sq_ms * 1e-3) / 1e9, 1)},
                             "mask_pack": {"bytes": int(PACK_BYTES_128x416 * scale_px * B), "ms": mp_ms,
                                           "GB/s": round(PACK_BYTES_128x416 * scale_px * B / (mp_ms * 1e-3) / 1e9, 1)}},
                   "note": "algorithmic bytes: flow planes 0,1 (squeeze); u8 strip + 2 flow + 2 seg planes in, packed "
                           "[2B,H,W,8] float32 (f16x3: 8 hi | 8 lo halves, the same 32 B per pixel) out (mask_pack); peak = 8 TB/s HBM3E spec"}
        res = {
            "metric": "pose-net triplets/sec (128x416x3-frame)" if (H, W) == (128, 416)
                      else "pose-net triplets/sec (%dx%dx3-frame)" % (H, W),
            "value": round(value, 2), "unit": "triplets/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup + settle, "warmup_arg": args.warmup, "settle_steps": settle,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic (splitmix64 seed 8964; random-init He-uniform weights; no KITTI/ckpt offline)",
            "config": {"workload": workload_name(B, H, W, world),
                       "version": FLAGSHIP_VERSION, "batch_per_gpu": B, "height": H, "width": W,
                       "parallelism": "window-sharded replicas x%d" % world, "batches_in_flight": nset,
                       **({"options": args.options} if args.options else {})},
            "roofline": roof,
            ("fast_mode" if other == "f16x3" else "reference_arithmetic"): other_block,
            "timing": timing,
            "whole_path_tflops_per_gpu": round(whole, 2),
            "whole_path_frac_of_mfma_peak": round(whole / roof["peak"], 4),
            "roofline_prologue": pro,
            "kernel_avg_ms": kern_ms,
            "range_recovery": eng.range_stats(),
            "pipelined": pipelined,
            "gather": gather,
        }
        # parity on the bench's own batch (bounded: the first 2 windows) + CPU baseline beside it
        from oracle import c_oracle
        ns = min(2, B)
        want = c_oracle.forward(cfg, img[:ns], flow[:ns], seg[:ns], weights)
        res["max_abs_err_vs_oracle"] = float(np.abs(poses[:ns] - want).max())
        res["max_abs_ref"] = float(np.abs(want).max())
        res["oracle_note"] = "CPU restatement (TF1 itself cannot run offline: parity unpinned vs TF)"
        if world == 1 and not args.no_cpu_baseline:
            nb = max(1, min(args.cpu_sample, B))
            cores, quota = usable_cores()
            c_oracle.forward(cfg, img[:1], flow[:1], seg[:1], weights, nthreads=cores)   # warm-up
            c0 = time.perf_counter()
            passes = 0
            while passes < 3 or time.perf_counter() - c0 < 10.0:
                c_oracle.forward(cfg, img[:nb], flow[:nb], seg[:nb], weights, nthreads=cores)
                passes += 1
                if time.perf_counter() - c0 > 30.0:
                    break
            cdt = time.perf_counter() - c0
            n1 = min(2, B)                                      # one thread: ~0.3 s per triplet
            s0 = time.perf_counter()
            c_oracle.forward(cfg, img[:n1], flow[:n1], seg[:n1], weights, nthreads=1)
            sdt = time.perf_counter() - s0
            res["cpu_baseline"] = {"value": round(nb * passes / cdt, 3), "unit": "triplets/s", "cores": cores,
                                   "kind": "port", "cpu_model": cpu_model(),
                                   "host_logical_cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)),
                                   "cgroup_cpu_quota": quota,
                                   "one_core": {"value": round(n1 / sdt, 3), "unit": "triplets/s", "cores": 1,
                                                "sample": "1 pass of %d triplets, 1 thread" % n1},
                                   "sample": "%d passes of %d triplets (%dx%d) of the bench batch through "
                                             "oracle/davo_oracle.c (f32, OpenMP, -O3 -march=x86-64-v3); `cores` = threads "
                                             "used = min(affinity mask, cgroup quota) of this process on the GPU box's host"
                                             % (passes, nb, H, W)}
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)

    if comm is not None:
        comm.barrier()
        comm.close()
    for st in sets:
        for b in st:
            b.free()
    eng.close()


if __name__ == "__main__":
    main()
