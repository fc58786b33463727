#!/usr/bin/env python
"""Build an experimental variant of libdavo_hip.so next to the product library.

    python tools/build_variant.py _tuning -DDAVO_TUNING          # measurement knobs compiled in
    DAVO_LIB_SUFFIX=_tuning DAVO_H3_TILE=4 python bench.py ...    # ... and used

The product library (no suffix) reads no environment variable and carries no measurement switch in its kernels;
`-DDAVO_TUNING` enables the DAVO_* knobs named in csrc/plan.hip, csrc/weights.hip, csrc/forward.hip,
csrc/launch_h3_impl.h and the `dbg` bits of csrc/conv_igemm_h3.h.  Other -D flags select kernel variants
(e.g. -DDAVO_REM_STAGES=2)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    if len(sys.argv) < 2 or not sys.argv[1].startswith("_"):
        raise SystemExit(__doc__)
    from davo_amd import _lib
    print(_lib.build(force="--force" in sys.argv, verbose=True, suffix=sys.argv[1],
                     extra_flags=[a for a in sys.argv[2:] if a not in ("--force", "--allow-packed-f32")],
                     allow_packed_f32="--allow-packed-f32" in sys.argv))      # e.g. the reproducers of DESIGN.md section 4 (fa, e5...)
