#!/usr/bin/env python
"""How often does the pose of one fixed batch differ between repeated forwards?  (GPU box)
    python tools/exp/flake_count.py --batch 1 [--tile -1] [--reps 200] [--lib path/to/other/libdavo_hip.so]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--height", type=int, default=128)
ap.add_argument("--width", type=int, default=416)
ap.add_argument("--tile", type=int, default=-1)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--lib", default="")
ap.add_argument("--options", default="")
a = ap.parse_args()
from davo_amd import _lib                                            # noqa: E402
if a.lib:                                                            # an older build: entry points it lacks become inert stand-ins
    import ctypes
    _lib.LIB_PATH = os.path.abspath(a.lib)
    real = ctypes.CDLL(_lib.LIB_PATH)

    class Shim:
        def __getattr__(self, name):
            try:
                return getattr(real, name)
            except AttributeError:
                class Missing:
                    argtypes = restype = None

                    def __call__(self, *x):
                        return 0
                m = Missing()
                setattr(self, name, m)
                return m
    ctypes.CDLL = lambda path, *k, **kw: Shim()
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402

cfg = parse_version(FLAGSHIP_VERSION)
B, H, W = a.batch, a.height, a.width
img, flow, seg = synth.make_inputs(B, H, W)
e = Engine(cfg, H, W, B)
e.load_weights(synth.make_weights(cfg))
e.set_option("force_tile", a.tile)
for kv in filter(None, a.options.split(",")):
    k, v = kv.split("=")
    e.set_option(k, int(v))
bufs = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
ref, bad, worst, tref, shown = None, 0, 0.0, None, 0
nt = 2 * ((2 * B * ((H + 7) // 8) * ((W + 7) // 8) + 127) // 128) * 8 * 6 if a.tile == 0 and not a.lib else 0
for r in range(a.reps):
    e.forward_device(B, *bufs)
    e.synchronize()
    pose = bufs[3].download((B, 2, 6))
    tiles = e.debug_read("pose_tiles", (nt,)) if nt else None
    if ref is None:
        ref, tref = pose, tiles
    elif not np.array_equal(pose, ref):
        bad += 1
        worst = max(worst, float(np.abs(pose - ref).max()))
        if nt and shown < 6:
            shown += 1
            d = np.nonzero(tiles != tref)[0]
            print("  rep %d: %d tile words differ; (region [0 = tile sums, 1..3 = wave 0..2's own], head, mtile, ntile, slot*3+k): %s  values %s vs %s" %
                  (r, d.size, [(int(i // nt), int(i % nt // (52 * B // 4 * 8 * 6)), int(i % nt // 48 % (52 * B // 4)), int(i // 6 % 8), int(i % 6)) for i in d[:8]],
                   tiles[d[:4]], tref[d[:4]]), flush=True)
            for i in d[:2]:
                j = int(i % nt)
                print("    word %d: tile sum %r (ref %r); waves 0..2 now %s ref %s" % (j, tiles[j], tref[j],
                      [float(tiles[(w + 1) * nt + j]) for w in range(3)] if tiles.size > nt else "-",
                      [float(tref[(w + 1) * nt + j]) for w in range(3)] if tiles.size > nt else "-"), flush=True)
print("lib %s B=%d tile %d %s: %d of %d forwards differ from the first (worst %.3g); plan cnv7 %s" %
      (os.path.basename(_lib.LIB_PATH), B, a.tile, a.options, bad, a.reps - 1, worst, e.last_plan(6)), flush=True)
e.close()
