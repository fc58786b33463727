#!/usr/bin/env python
"""Soak test on the GPU: the same batches again and again, every result compared BIT FOR BIT with the first one.

The f16x3 path leans on hand-counted waits (LDS-DMA rings, counted vmcnt / lgkmcnt, double-buffered patches with one barrier
per tile): a miscounted wait is a race that a single parity run can pass.  Here each of several (batch size, frame size)
shapes runs `iters` forwards over rotating input sets with the GPU kept busy; any forward whose poses differ from that input
set's first result - by a single bit - fails.  The first results are also checked against the CPU oracle.

    python tools/soak.py [iters per shape, default 400]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402
from oracle import c_oracle                                          # noqa: E402  (checker only)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    bad = 0
    # round 3: the small batches too (deep LDS rings, split-K + fix-up kernel, 128x32 tiles with the fused pose head)
    for B, H, W, nsets, precision in ((32, 128, 416, 3, "f16x3"), (16, 128, 416, 3, "f16x3"), (5, 128, 416, 2, "f16x3"),
                                      (8, 256, 832, 2, "f16x3"), (7, 52, 172, 3, "f16x3"), (32, 128, 416, 2, "f32"),
                                      (1, 128, 416, 3, "f16x3"), (2, 128, 416, 3, "f16x3"), (3, 128, 416, 3, "f16x3"),
                                      (4, 128, 416, 2, "f16x3"), (1, 256, 832, 2, "f16x3"), (3, 52, 172, 3, "f16x3")):
        e = Engine(cfg, H, W, B)
        e.load_weights(weights)
        e.set_precision(precision)
        sets, refs = [], []
        for k in range(nsets):
            img, flow, seg = synth.make_inputs(B, H, W, first_window=101 * k + 7)
            d = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
            e.forward_device(B, *d)
            e.synchronize()
            ref = d[3].download((B, 2, 6)).copy()
            if k == 0:
                want = c_oracle.forward(cfg, img[:2], flow[:2], seg[:2], weights)
                err = float(np.abs(ref[:2] - want).max())
                assert err <= 1e-4 and err <= 1e-4 * float(np.abs(want).max()) * 10, err
            sets.append(d)
            refs.append(ref)
        mism = 0
        for i in range(iters):
            k = i % nsets
            e.forward_device(B, *sets[k])
            # every result is checked: the download synchronises, the next forwards are queued right behind it
            got = sets[k][3].download((B, 2, 6))
            if not np.array_equal(got, refs[k]):
                mism += 1
                if mism <= 3:
                    print("  MISMATCH B=%d %dx%d %s iter %d set %d: max abs diff %.3g" % (B, H, W, precision, i, k, np.abs(got - refs[k]).max()))
        # and a burst with the queue kept deep (no per-forward synchronisation), checked at the end
        for i in range(iters):
            e.forward_device(B, *sets[i % nsets])
        e.synchronize()
        last = (iters - 1) % nsets
        if not np.array_equal(sets[last][3].download((B, 2, 6)), refs[last]):
            mism += 1
            print("  MISMATCH after the queued burst, B=%d %dx%d %s" % (B, H, W, precision))
        print("B=%-3d %dx%-4d %-6s %d + %d forwards, %d mismatches" % (B, H, W, precision, iters, iters, mism))
        bad += mism
        e.close()
    print("soak: %s" % ("ok" if bad == 0 else "%d MISMATCHES" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
