#!/usr/bin/env python
"""Per-layer comparison of the HIP path with the float64 oracle at a chosen shape (GPU box; a debugging aid).
    python tools/exp/dbg_layers.py B H W [k=v,k=v options]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402
from oracle import davo_oracle as O                                  # noqa: E402  (checker only)

B, H, W = (int(a) for a in sys.argv[1:4])
opts = sys.argv[4] if len(sys.argv) > 4 else ""
cfg = parse_version(FLAGSHIP_VERSION)
img, flow, seg = synth.make_inputs(B, H, W)
weights = synth.make_weights(cfg)
keep = {}
want = O.forward(cfg, img, flow, seg, weights, np.float64, keep)
for precision in ("f16x3", "f32"):
    e = Engine(cfg, H, W, B)
    e.load_weights(weights)
    e.set_precision(precision)
    e.set_option("fuse_pose", 0)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        e.set_option(k, int(v))
    got = e.forward(img, flow, seg)
    h2, w2 = (H + 1) // 2, (W + 1) // 2
    h4, w4 = (h2 + 1) // 2, (w2 + 1) // 2
    for name, shp in (("cnv1", (h2, w2, 16)), ("cnv2", (h4, w4, 32)), ("cnv3", (h4, w4, 64)), ("cnv4", (h4, w4, 128)), ("cnv5", (h4, w4, 256))):
        a = e.debug_read(name, (2 * B,) + shp)
        d = np.abs(a - keep[name])
        print(precision, name, "max err %.3g (max ref %.3g)" % (d.max(), np.abs(keep[name]).max()), "worst image", int(d.reshape(2 * B, -1).max(1).argmax()), flush=True)
    c6 = e.debug_read("cnv6", (2 * B, h4, w4, 256))
    print(precision, "cnv6 max err %.3g" % max(np.abs(c6[..., :128] - keep["rotation/cnv6"]).max(), np.abs(c6[..., 128:] - keep["translation/cnv6"]).max()))
    print(precision, "pose max err %.3g" % np.abs(got - want).max(), "plan", [e.last_plan(i) for i in range(7)] if hasattr(e, "last_plan") else "")
    e.close()
