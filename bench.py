#!/usr/bin/env python
"""bench.py — pose-net triplets/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the hot path (SE squeeze + excite, mask/pack, 8 conv layers, pose head)
over one batch of synthetic triplets that is already resident in HBM.  At N=1 the workload is
BASELINE.json configs[1]: batch 32, 128x416, flagship variant
(dilatedPoseNN-cnv6_128 + se_flow + fc_tanh).  For N>1 every rank runs the same per-GPU batch on
its own windows (weak scaling, no data-path collective: windows are independent,
test_kitti_pose.py:134-145); after the timed region the ranks' poses are gathered once over
RCCL, which is reported separately and is not part of `value`.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--height H --width W]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# FLOPs of one PoseNN pair evaluation at 128x416, SURVEY.md §8a table L (MACs x 2)
MACS_PER_PAIR_128x416 = 3890085888
CNV6_MACS_PER_PAIR_128x416 = 2 * 981467136          # rotation + translation cnv6, fused into one launch
PEAK_F32_MFMA_TFLOPS = 157.3                         # MI355X_MICROARCH.md: FP32 matrix, dense
PEAK_F16_MFMA_TFLOPS = 2500.0                        # MI355X_MICROARCH.md: BF16/FP16 matrix, dense
PEAK_HBM_GBS = 8000.0


def usable_cores():
    """host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="triplets per GPU per step")
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic windows generated per rank (tiled to the batch)")
    ap.add_argument("--precision", choices=["f16x3", "f32"], default="f16x3",
                    help="f16x3: split-fp16 MFMA, float32-grade (default); f32: FP32 MFMA, bit-exact fmaf chains")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches kept in flight on the GPU in the timed region (davo_set_inflight).  Default 1: one "
                         "batch at a time, so the event-bracketed kernel durations are the kernels' own")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra 2-in-flight throughput measurement (used for the rocprofv3 passes, so that "
                         "per-kernel statistics are not mixed with overlapped launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8, help="triplets per CPU-baseline pass")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch                                     # plumbing only: barrier / synchronize / RCCL gather
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    # one rank per GPU; DAVO_BENCH_SHARE_GPU=1 (rehearsal on a 1-GPU box) folds ranks onto the devices present
    device_index = local_rank % ndev if os.environ.get("DAVO_BENCH_SHARE_GPU") == "1" else local_rank
    if device_index >= ndev:
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, ndev))
    torch.cuda.set_device(device_index)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DAVO_BENCH_BACKEND", "nccl")        # "nccl" is RCCL on ROCm
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", device_index))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        except Exception as e:                       # noqa: BLE001 — keep the measurement alive on gloo
            sys.stderr.write("rank %d: %s init failed (%s); falling back to gloo\n" % (rank, backend, e))
            backend = "gloo"
            dist.init_process_group("gloo", rank=rank, world_size=world)
    comm_dev = torch.device("cuda", device_index) if backend == "nccl" else torch.device("cpu")

    from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION
    cfg = parse_version(FLAGSHIP_VERSION)
    B, H, W = args.batch, args.height, args.width
    scale_px = (H * W) / float(128 * 416)
    flops_per_triplet = 2 * 2 * MACS_PER_PAIR_128x416 * scale_px
    cnv6_flops_per_launch = 2 * CNV6_MACS_PER_PAIR_128x416 * scale_px * (2 * B)

    weights = synth.make_weights(cfg)
    eng = Engine(cfg, H, W, B, device=device_index)
    eng.load_weights(weights)
    eng.set_precision(args.precision)

    # synthetic windows of this rank's shard, resident in HBM before the timed region; one buffer set per
    # in-flight slot (consecutive steps work on different batches of the shard)
    nu = max(1, min(args.unique, B))
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

    def sync_all():
        eng.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        eng.synchronize()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        eng.forward_device(B, *sets[i % nset])
    eng.synchronize()

    # timed region: K steps; only the dominant kernel (main cnv6 launch) is bracketed by HIP events
    # on the launch stream, so the event records do not perturb the other 15 launches
    eng.profile(2)
    eng.profile_reset()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.forward_device(B, *sets[i % nset])
    sync_all()
    elapsed = time.perf_counter() - t0
    dominant = eng.profile_entries()
    # untimed extra pass, one batch in flight: per-kernel breakdown (every launch bracketed)
    eng.set_inflight(1)
    eng.profile(1)
    eng.profile_reset()
    for _ in range(max(3, args.steps // 4)):
        eng.forward_device(B, d_img, d_flow, d_seg, d_pose)
    kernels = eng.profile_entries()
    eng.profile(False)

    # for reference: the bit-exact FP32-MFMA mode on the same batch (short, untimed for `value`)
    f32_mode = None
    if args.precision == "f16x3" and world == 1:
        eng.set_precision("f32")
        for _ in range(2):
            eng.forward_device(B, d_img, d_flow, d_seg, d_pose)
        eng.profile(2)
        eng.profile_reset()
        eng.synchronize()
        f0 = time.perf_counter()
        for _ in range(5):
            eng.forward_device(B, d_img, d_flow, d_seg, d_pose)
        eng.synchronize()
        fdt = time.perf_counter() - f0
        n32, ms32 = eng.profile_entries().get("cnv6", (0, 0.0))
        plan32 = eng.last_plan(5)
        share = plan32[0][0] / float(sum(m for m, _ in plan32))
        tf32 = cnv6_flops_per_launch * share / (ms32 / max(n32, 1) * 1e-3) / 1e12 if ms32 > 0 else 0.0
        f32_mode = {"value": round(B * 5 / fdt, 2), "unit": "triplets/s", "cnv6_tflops": round(tf32, 2),
                    "cnv6_frac_of_f32_mfma_peak": round(tf32 / PEAK_F32_MFMA_TFLOPS, 4),
                    "note": "davo_set_precision(0): v_mfma_f32_32x32x2_f32, bit-exact fmaf chains"}
        eng.profile(False)
        eng.set_precision("f16x3")
        eng.forward_device(B, d_img, d_flow, d_seg, d_pose)      # d_pose holds the f16x3 result again
        eng.synchronize()

    # extra, outside `value`: throughput with two batches in flight (the next batch's small kernels overlap this
    # batch's large convolutions; kernel wall durations are then shared time, so no roofline is quoted for it)
    pipelined = None
    if not args.no_pipelined and nset == 1:
        img2, flow2, seg2 = synth.make_inputs(nu, H, W, first_window=(world + rank) * B)
        reps2 = -(-B // nu)
        set2 = (eng.alloc(img.nbytes).upload(np.tile(img2, (reps2, 1, 1, 1))[:B]),
                eng.alloc(flow.nbytes).upload(np.tile(flow2, (reps2, 1, 1, 1, 1))[:B]),
                eng.alloc(seg.nbytes).upload(np.tile(seg2, (reps2, 1, 1, 1, 1))[:B]), eng.alloc(B * 12 * 4))
        both = [sets[0], set2]
        eng.set_inflight(2)
        for i in range(4):
            eng.forward_device(B, *both[i % 2])
        sync_all()
        p0 = time.perf_counter()
        for i in range(args.steps):
            eng.forward_device(B, *both[i % 2])
        sync_all()
        pdt = time.perf_counter() - p0
        tp = torch.tensor([pdt], dtype=torch.float64, device=comm_dev)
        if world > 1:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        pipelined = {"batches_in_flight": 2, "value": round(world * B * args.steps / float(tp.item()), 2), "unit": "triplets/s",
                     "ms_per_step": round(float(tp.item()) / args.steps * 1e3, 4),
                     "note": "davo_set_inflight(ctx, 2); same K steps, barrier + synchronize on both sides, max over ranks"}
        eng.set_inflight(1)
        eng.forward_device(B, *sets[0])
        eng.synchronize()
        sets.append(set2)

    t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())

    poses = d_pose.download((B, 2, 6))

    # one gather of the shard poses (config 4's stitch input) over RCCL, outside the timed region
    gather_ms, gather_err = None, None
    if world > 1:
        try:
            mine = torch.from_numpy(poses).to(comm_dev)
            out = [torch.empty_like(mine) for _ in range(world)]
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            dist.all_gather(out, mine)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) * 1e3
            assert torch.equal(out[rank].cpu(), torch.from_numpy(poses))
        except Exception as e:                       # noqa: BLE001
            gather_err = "%s: %s" % (type(e).__name__, e)

    if rank == 0:
        value = world * B * args.steps / elapsed_max
        # the dominant kernel = the main cnv6 launch (whole rounds of 128x128 tiles); a remainder
        # launch with narrower tiles, if the planner issued one, is listed as "cnv6.rem"
        plan6 = eng.last_plan(5)                      # [(128-row M tiles, N tile | f16x3 tile id), ...]
        total_mtiles6 = sum(m for m, _ in plan6)
        cnv6_flops_main = cnv6_flops_per_launch * plan6[0][0] / total_mtiles6
        n6, ms6 = dominant.get("cnv6", (0, 0.0))
        avg6 = ms6 / max(n6, 1)
        achieved = cnv6_flops_main / (avg6 * 1e-3) / 1e12 if avg6 > 0 else 0.0
        kern_ms = {k: round(v[1] / max(v[0], 1), 4) for k, v in kernels.items()}
        whole = flops_per_triplet * B * args.steps / elapsed_max / 1e12
        if args.precision == "f32":
            peak, dtype = PEAK_F32_MFMA_TFLOPS, "f32"
            kname = "davo::conv_igemm_f32<3,1,128,6> (cnv6 main launch: rotation|translation fused, N=256, K=2304)"
            peak_note = "FP32 MFMA dense peak (v_mfma_f32_32x32x2_f32)"
        else:
            # every algorithmic FLOP costs three fp16 MFMA FLOPs (hi*hi, hi*lo, lo*hi), so the
            # matrix-pipe roofline of this algorithm is the fp16 dense peak / 3
            peak, dtype = PEAK_F16_MFMA_TFLOPS / 3.0, "f16x3 (fp16 hi/lo split operands, 3 MFMA products, f32 accumulate)"
            tiles = {0: "128x32", 1: "256x64", 2: "256x128", 3: "128x256", 4: "128x128", 5: "256x256"}
            kname = ("davo::conv_igemm_h3<3,1,...,6,true,false,true> (cnv6 main launch: rotation|translation fused, N=256, "
                     "K=2304, %s tile, LDS-DMA staged, v_mfma_f32_16x16x32_f16)" % tiles.get(plan6[0][1], "?"))
            peak_note = "fp16 MFMA dense peak 2500 TFLOP/s / 3 products per algorithmic FLOP; frac = matrix-pipe utilisation"
        # HBM traffic of the dominant kernel: from the most recent committed PMC pass (profiles/), not live —
        # rocprofv3 counter collection cannot run inside the timed process
        traffic, traffic_src = None, None
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
            key = "conv_igemm_f32<3, 1, 128, 6>" if args.precision == "f32" else "conv_igemm_h3<3, 1, 4, 2, 2, 4, 6, true, false"
            for f in reversed(cands):
                tj = json.load(open(f))
                hit = [v for k, v in tj["kernels"].items() if key in k]
                if hit and (B, H, W) == (32, 128, 416):
                    traffic = hit[0]["read_bytes"] + hit[0]["write_bytes"]
                    traffic_src = "profiles/" + os.path.basename(f)
                    break
        except (OSError, ValueError, KeyError):
            pass
        res = {
            "metric": "pose-net triplets/sec (128x416x3-frame)" if (H, W) == (128, 416)
                      else "pose-net triplets/sec (%dx%dx3-frame)" % (H, W),
            "value": round(value, 2), "unit": "triplets/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic (splitmix64 seed 8964; random-init He-uniform weights; no KITTI/ckpt offline)",
            "config": {"workload": "BASELINE.json configs[1]: single MI355X, batch=%d synthetic %dx%d RGB+flow+seg "
                                   "triplets, dilatedPoseNN-cnv6_128 + se_flow + fc_tanh" % (B, H, W),
                       "version": FLAGSHIP_VERSION, "batch_per_gpu": B, "height": H, "width": W,
                       "parallelism": "window-sharded replicas x%d" % world, "batches_in_flight": nset},
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": traffic_src, "peak_note": peak_note,
                         "avg_launch_ms": round(avg6, 4), "flops_per_launch": cnv6_flops_main,
                         "launch_plan": "cnv6 as %s (mtiles of 128 rows, N tile)" % plan6},
            "whole_path_tflops_per_gpu": round(whole, 2),
            "whole_path_frac_of_mfma_peak": round(whole / peak, 4),
            "kernel_avg_ms": kern_ms,
            "f32_exact_mode": f32_mode,
            "pipelined": pipelined,
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "gather_backend": backend, "gather_error": gather_err,
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
            cores = usable_cores()
            c_oracle.forward(cfg, img[:1], flow[:1], seg[:1], weights, nthreads=cores)   # warm-up
            c0 = time.perf_counter()
            passes = 0
            while passes < 3 or time.perf_counter() - c0 < 10.0:
                c_oracle.forward(cfg, img[:nb], flow[:nb], seg[:nb], weights, nthreads=cores)
                passes += 1
                if time.perf_counter() - c0 > 30.0:
                    break
            cdt = time.perf_counter() - c0
            res["cpu_baseline"] = {"value": round(nb * passes / cdt, 3), "unit": "triplets/s", "cores": cores,
                                   "kind": "port",
                                   "sample": "%d passes of %d triplets (%dx%d) of the bench batch through "
                                             "oracle/davo_oracle.c (f32, OpenMP, -O3 -march=x86-64-v3)" % (passes, nb, H, W)}
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)

    for st in sets:
        for b in st:
            b.free()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
