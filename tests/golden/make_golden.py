"""Regenerates tests/golden/pose_golden.json from the float64 numpy oracle.

The reference cannot run offline (TensorFlow 1.13 absent), so these are NOT TF outputs:
they are the oracle's own outputs, committed so that (a) the GPU box — which has no
/root/reference and regenerates only the seeded inputs — checks the HIP path against fixed
numbers, and (b) any later drift of the oracle itself is caught.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from davo_amd import synth                                    # noqa: E402
from davo_amd.version import parse_version, FLAGSHIP_VERSION  # noqa: E402
from oracle import davo_oracle as O                           # noqa: E402

CASES = [
    # name, version, B, H, W, flow_scale
    ("flagship_b2_128x416", FLAGSHIP_VERSION, 2, 128, 416, 1.0),
    ("flagship_b1_256x832", FLAGSHIP_VERSION, 1, 256, 832, 1.0),
    ("flagship_b3_32x64_smallflow", FLAGSHIP_VERSION, 3, 32, 64, 0.02),
    ("flagship_b3_36x100_ragged", FLAGSHIP_VERSION, 3, 36, 100, 0.3),      # odd map sizes: every tile edge is ragged
    ("relu_normflow_absh_b2_64x96", "v1-sharedNN-dilatedPoseNN-segmask_all-se_flow-norm_flow-abs_flow_h", 2, 64, 96, 1.0),
    ("lrelu_segmask_rgb_b2_64x96", "v1-sharedNN-dilatedPoseNN-segmask_rgb-se_flow-fc_lrelu-abs_flow_v", 2, 64, 96, 0.05),
    ("no_segmask_b2_64x96", "v1-sharedNN-dilatedPoseNN-cnv6_64-no_segmask", 2, 64, 96, 1.0),
    ("static_src_b2_64x96", "v1-sharedNN-dilatedPoseNN-segmask_all-static", 2, 64, 96, 1.0),
    ("static_all_b2_64x96", "v1-sharedNN-dilatedPoseNN-segmask_all", 2, 64, 96, 1.0),
    ("v0_rgb_only_b2_64x96", "v0-sharedNN-dilatedPoseNN-segmask-se_flow-abs_flow-fc_tanh", 2, 64, 96, 1.0),
]


def checksum(a):
    a = np.asarray(a, np.float64)
    flat = a.ravel()
    idx = (np.arange(8, dtype=np.int64) * 2654435761 + 12345) % flat.size
    return {"shape": list(a.shape), "mean": float(flat.mean()), "l2": float(np.sqrt((flat * flat).sum())),
            "samples": [[int(i), float(flat[i])] for i in idx]}


def main():
    out = {"_note": "float64 numpy oracle outputs (oracle/davo_oracle.py); NOT TensorFlow outputs — "
                    "the reference cannot run offline (SURVEY.md 8c: parity unpinned)",
           "seed": synth.SEED, "cases": {}}
    for name, version, B, H, W, fscale in CASES:
        cfg = parse_version(version)
        img, flow, seg = synth.make_inputs(B, H, W)
        flow = (flow * np.float32(fscale)).astype(np.float32)
        weights = synth.make_weights(cfg)
        keep = {}
        pose = O.forward(cfg, img, flow, seg, weights, np.float64, keep)
        out["cases"][name] = {"version": version, "B": B, "H": H, "W": W, "flow_scale": fscale,
                              "pose": pose.tolist(),
                              "layers": {k: checksum(v) for k, v in keep.items()},
                              "inputs": {"img": checksum(img), "flow": checksum(flow), "seg": checksum(seg)}}
        print(name, np.abs(pose).max())
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
