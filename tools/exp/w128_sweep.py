#!/usr/bin/env python
"""Batch sweep of the four-wave tiles (option "wave128"): at every batch size the poses with wave128 = 2 (conv_igemm_h3w for whole
256x256 tiles, conv_igemm_h3w64 for the remainder rows) must equal wave128 = 0 (conv_igemm_h3 everywhere) to the bit; prints the
launch plans and the device-path step time of both."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402


def main():
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    bad = 0
    for H, W, batches in ((128, 416, (5, 8, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128)), (256, 832, (2, 4, 6, 8, 16, 33)), (64, 96, (16, 64))):
        for B in batches:
            img, flow, seg = synth.make_inputs(min(B, 8), H, W, first_window=3)
            reps = -(-B // img.shape[0])
            img, flow, seg = (np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:B] for a in (img, flow, seg))
            e = Engine(cfg, H, W, B)
            e.load_weights(weights)
            e.set_option("host_chunk", 0)
            bufs = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
            out, ms = {}, {}
            for w in (0, 2, 0, 2):                       # twice, alternating: the first pass also warms the chip up; the second is reported
                e.set_option("wave128", w)
                for _ in range(3):
                    e.forward_device(B, *bufs)
                e.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    e.forward_device(B, *bufs)
                e.synchronize()
                ms[w] = (time.perf_counter() - t0) / 5 * 1e3
                out[w] = bufs[3].download((B, 2, 6)).copy()
            same = np.array_equal(out[0], out[2])
            bad += not same
            print("%3dx%3d B=%3d  %s  cnv4 %s cnv5 %s cnv6 %s  step %.3f -> %.3f ms (%+.1f %%)" % (
                H, W, B, "bit-identical" if same else "DIFFERENT", e.last_plan(3), e.last_plan(4), e.last_plan(5), ms[0], ms[2], 100 * (ms[2] / ms[0] - 1)), flush=True)
            e.close()
    print("W128_SWEEP", "OK" if not bad else "FAILED (%d)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
