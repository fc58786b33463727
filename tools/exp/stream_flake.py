#!/usr/bin/env python
"""Reproduce a flaky delivery in the streamed guard-tripping test (tools/exp: diagnosis only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402
from oracle import c_oracle                                          # noqa: E402
from test_stream import _rescaled                                    # noqa: E402


def main():
    c_oracle.build()
    cfg = parse_version(FLAGSHIP_VERSION)
    B, H, W, n = 2, 64, 96, 13
    weights = synth.make_weights(cfg)
    data = [synth.make_inputs(B, H, W, first_window=3 * k) for k in range(n)]
    wants = [c_oracle.forward(cfg, *d, weights) for d in data]
    bad_runs = 0
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
        e = Engine(cfg, H, W, B)
        e.load_weights(_rescaled(weights, 16))
        e.set_option("stable_inputs", 1)
        e.set_inflight(2)
        outs = [np.full((B, 2, 6), np.nan, np.float32) for _ in range(n)]
        bufs = [np.empty_like(a) for a in data[0]]
        for k in range(n):
            for buf, a in zip(bufs, data[k]):
                buf[...] = a
            e.submit(*bufs, outs[k])
        e.synchronize()
        errs = [float(np.abs(outs[k] - wants[k]).max()) for k in range(n)]
        bad = [k for k in range(n) if not errs[k] < 1e-4]
        if bad:
            bad_runs += 1
            print("rep %d: wrong batches %s errs %s stats %s report %r" % (rep, bad, ["%.3g" % errs[k] for k in bad], e.range_stats(), e.range_report()), flush=True)
        e.close()
    print("%d bad runs" % bad_runs)


if __name__ == "__main__":
    main()
