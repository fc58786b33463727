#!/usr/bin/env python
"""float32 mode: poses and cnv4..cnv6 activations of this library build as .npy files, for a bit-for-bit comparison between builds
(tools/exp: e.g. the product against `tools/build_variant.py _r4f32 -DDAVO_F32_EARLY_STORE=0 -DDAVO_F32_FAST_EPILOGUE=0` with
"merge_rem_f32" 0 = round 4's loop, epilogue and launch plan).   python tools/exp/f32_bits.py OUT_DIR [merge_rem_f32]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402

out, merge = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
os.makedirs(out, exist_ok=True)
cfg = parse_version(FLAGSHIP_VERSION)
w = synth.make_weights(cfg)
for B, H, W in ((1, 128, 416), (3, 64, 96), (5, 128, 416), (32, 128, 416), (2, 256, 832), (7, 96, 200)):
    e = Engine(cfg, H, W, B)
    e.load_weights(w)
    e.set_precision("f32")
    e.set_option("merge_rem_f32", merge)
    e.set_option("host_chunk", 0)
    d = synth.make_inputs(B, H, W, first_window=B)
    np.save(os.path.join(out, "pose_%d_%d_%d.npy" % (B, H, W)), e.forward(*d))
    for name, ch in (("cnv4", 128), ("cnv5", 256), ("cnv6", 256)):
        np.save(os.path.join(out, "%s_%d_%d_%d.npy" % (name, B, H, W)), e.debug_read(name, (2 * B, (H + 3) // 4, (W + 3) // 4, ch)))
    e.close()
print("written", out)
