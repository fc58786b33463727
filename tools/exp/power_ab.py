#!/usr/bin/env python
"""Board power and shader clock under two option sets, arms alternated in ONE process (GPU box).

    python tools/exp/power_ab.py --arms "merge_order=0;merge_order=1" [--batch 32] [--seconds 3] [--rounds 4]

Power / clock come from the amdgpu hwmon files of the card (power1_average or power1_input in uW, freq1_input in Hz), read by a
sampler thread every 20 ms while the arm's forwards run back to back; `rocm-smi --json -P -c` is the fallback."""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                                   # noqa: E402
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION  # noqa: E402


def find_hwmon():
    """every amdgpu hwmon with a power file (a host holds many cards; the one this process loads is the one whose power moves:
    chosen after the fact as the card with the highest mean power over the loaded runs)"""
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        p = [f for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, f))]
        if p:
            out.append((os.path.join(d, p[0]), os.path.join(d, "freq1_input") if os.path.exists(os.path.join(d, "freq1_input")) else None))
    return out


def read_smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--json", "-P", "-c"], capture_output=True, text=True, timeout=5).stdout
        j = json.loads(out)
        card = next(iter(j.values()))
        pw = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k), float("nan"))
        ck = next((float(str(v).strip("()Mhz ")) for k, v in card.items() if k.startswith("sclk clock speed")), float("nan"))
        return pw, ck
    except Exception:                                                # noqa: BLE001
        return float("nan"), float("nan")


ap = argparse.ArgumentParser()
ap.add_argument("--arms", default="merge_order=0;merge_order=1")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=3.0)
ap.add_argument("--rounds", type=int, default=4)
a = ap.parse_args()
cfg = parse_version(FLAGSHIP_VERSION)
B, H, W = a.batch, 128, 416
e = Engine(cfg, H, W, B)
e.load_weights(synth.make_weights(cfg))
img, flow, seg = synth.make_inputs(8, H, W)
r = -(-B // 8)
img, flow, seg = np.tile(img, (r, 1, 1, 1))[:B], np.tile(flow, (r, 1, 1, 1, 1))[:B], np.tile(seg, (r, 1, 1, 1, 1))[:B]
d = (e.alloc(img.nbytes).upload(img), e.alloc(flow.nbytes).upload(flow), e.alloc(seg.nbytes).upload(seg), e.alloc(B * 48))
arms = [[(kv.split("=")[0], int(kv.split("=")[1])) for kv in spec.split(",")] for spec in a.arms.split(";")]
hw = find_hwmon()
print("power sources: %d hwmon cards" % len(hw) if hw else "rocm-smi --json -P -c", flush=True)
res = {i: {"ms": [], "w": [], "mhz": []} for i in range(len(arms))}
for rnd in range(a.rounds):
    for i in (range(len(arms)) if rnd % 2 == 0 else reversed(range(len(arms)))):
        for k, v in arms[i]:
            e.set_option(k, v)
        for _ in range(100):
            e.forward_device(B, *d)
        e.synchronize()
        stop, pw, ck = threading.Event(), [], []

        def sampler():
            while not stop.is_set():
                if hw:
                    rowp, rowc = [], []
                    for pf, cf in hw:
                        try:
                            rowp.append(int(open(pf).read()) / 1e6)
                            rowc.append(int(open(cf).read()) / 1e6 if cf else float("nan"))
                        except (OSError, ValueError):
                            rowp.append(float("nan")); rowc.append(float("nan"))
                    pw.append(rowp); ck.append(rowc)
                    time.sleep(0.02)
                else:
                    w, c = read_smi()
                    pw.append(w); ck.append(c)
        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < a.seconds:
            for _ in range(50):
                e.forward_device(B, *d)
            e.synchronize()
            n += 50
        dt = time.perf_counter() - t0
        stop.set(); th.join()
        res[i]["ms"].append(dt / n * 1e3)
        res[i]["w"].append(np.nanmean(np.array(pw, float), axis=0) if pw else float("nan"))
        res[i]["mhz"].append(np.nanmean(np.array(ck, float), axis=0) if ck else float("nan"))
card = 0
if hw:
    allw = np.array([w for i in res for w in res[i]["w"]])
    card = int(np.nanargmax(np.nanmean(allw, axis=0)))
    print("card under load: %s (mean %.0f W; the others %s W)" % (hw[card][0], np.nanmean(allw, axis=0)[card],
          [int(x) for k, x in enumerate(np.nanmean(allw, axis=0)) if k != card][:8]), flush=True)
for i, spec in enumerate(a.arms.split(";")):
    x = res[i]
    w = np.array([np.atleast_1d(v)[card] for v in x["w"]])
    c = np.array([np.atleast_1d(v)[card] for v in x["mhz"]])
    print("%-24s ms/step %.4f (%.4f..%.4f)  board power %.0f W (%.0f..%.0f)  sclk %.0f MHz (%.0f..%.0f)  [%d rounds of %.0f s]"
          % (spec, np.median(x["ms"]), min(x["ms"]), max(x["ms"]), np.nanmedian(w), np.nanmin(w), np.nanmax(w),
             np.nanmedian(c), np.nanmin(c), np.nanmax(c), a.rounds, a.seconds), flush=True)
e.close()
