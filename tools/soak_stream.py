#!/usr/bin/env python
"""Soak of the streaming entry point on the GPU: thousands of davo_submit calls with random batch sizes, in-flight depths and `hold`
values, one recycled set of host arrays, both arithmetic modes and a guard-tripping checkpoint - every delivered pose block compared
BIT FOR BIT with what the synchronous entry point returned for the same batch (a stream that re-issues batches: with the oracle's
bar against the CPU oracle instead).

    python tools/soak_stream.py [submits per configuration, default 600]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402
from oracle import c_oracle                                          # noqa: E402  (checker only)


def rescaled(weights, shift):
    w = dict(weights)
    k = np.float32(2.0 ** shift)
    w["pose_exp_net/cnv3/weights"] = weights["pose_exp_net/cnv3/weights"] * k
    w["pose_exp_net/cnv3/biases"] = weights["pose_exp_net/cnv3/biases"] * k
    w["pose_exp_net/cnv4/weights"] = weights["pose_exp_net/cnv4/weights"] / k
    return w


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    rng = np.random.RandomState(5)
    bad = 0
    for H, W, Bmax, precision, shift in ((128, 416, 4, "f16x3", 0), (64, 96, 6, "f16x3", 0), (128, 416, 2, "f32", 0), (128, 416, 16, "f16x3", 0),
                                         (64, 96, 3, "f16x3", 16)):
        t0 = time.time()
        e = Engine(cfg, H, W, Bmax)
        e.load_weights(rescaled(weights, shift) if shift else weights)
        e.set_precision(precision)
        e.set_option("host_chunk", 0)
        pool = [synth.make_inputs(Bmax, H, W, first_window=11 * k) for k in range(6)]
        ref = {}
        if shift:                                   # scales move while the stream runs: the CPU oracle is the reference
            for k in range(6):
                for b in range(1, Bmax + 1):
                    ref[(k, b)] = c_oracle.forward(cfg, pool[k][0][:b], pool[k][1][:b], pool[k][2][:b], weights)
        else:
            for k in range(6):
                for b in range(1, Bmax + 1):
                    ref[(k, b)] = e.forward(pool[k][0][:b], pool[k][1][:b], pool[k][2][:b]).copy()
        bufs = [np.empty_like(a) for a in pool[0]]
        outs, keys = [], []
        mism = 0
        for i in range(n):
            if i % 97 == 0:
                e.set_inflight(int(rng.randint(1, 5)))          # delivers what is under way first
            k, b = int(rng.randint(6)), int(rng.randint(1, Bmax + 1))
            hold = int(rng.choice([0, 0, 0, 9]))
            o = np.full((b, 2, 6), np.nan, np.float32)
            if hold == 0:
                for buf, a in zip(bufs, pool[k]):
                    buf[:b] = a[:b]
                e.submit(bufs[0][:b], bufs[1][:b], bufs[2][:b], o)
                bufs[0][:b] = 255                                 # the arrays are the caller's again
            else:
                e.submit(pool[k][0][:b], pool[k][1][:b], pool[k][2][:b], o, hold=hold)
            outs.append(o); keys.append((k, b))
            if i % 53 == 52:
                e.wait(int(rng.randint(0, 3)))
        e.synchronize()
        for o, key in zip(outs, keys):
            if shift:
                ok = np.abs(o - ref[key]).max() <= 1e-4
            else:
                ok = np.array_equal(o, ref[key])
            if not ok:
                mism += 1
                if mism <= 3:
                    print("  MISMATCH %dx%d %s batch %s: max abs diff %.3g" % (H, W, precision, key, np.nanmax(np.abs(o - ref[key]))))
        print("%dx%d Bmax %d %s%s: %d submits, %d mismatches, range stats %s, %.1f s" % (
            H, W, Bmax, precision, " guard-tripping" if shift else "", n, mism, e.range_stats(), time.time() - t0), flush=True)
        bad += mism
        e.close()
    print("SOAK_STREAM", "FAILED" if bad else "OK")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
