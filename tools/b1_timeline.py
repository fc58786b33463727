#!/usr/bin/env python
"""Launch-by-launch timeline of the batch-1 forward (the reference's operating point, run_inference.sh:44-51).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_b1 -- python3 $R/tools/b1_timeline.py run
    python3 tools/b1_timeline.py table gpurun_out/prof_b1 [out.md]

`run`: 200 device-resident batch-1 forwards, one batch in flight (f16x3, then float32).  `table`: from the kernel trace, per launch of
a forward - averaged over the last 100 forwards of each mode - its start relative to the forward's first kernel, its duration, and the
gap since the previous kernel ended (what a dependent launch costs on the GPU's front end)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION
    cfg = parse_version(FLAGSHIP_VERSION)
    e = Engine(cfg, 128, 416, 1)
    e.load_weights(synth.make_weights(cfg))
    data = synth.make_inputs(1, 128, 416)
    bufs = [e.alloc(a.nbytes).upload(a) for a in data] + [e.alloc(48)]
    for prec in ("f16x3", "f32"):
        e.set_precision(prec)
        for _ in range(200):
            e.forward_device(1, *bufs)
        e.synchronize()
    e.close()


def table(d, out=None):
    f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0]) for r in csv.DictReader(open(f))]
    rows.sort()
    rows = [r for r in rows if r[2].startswith("davo::")]
    # a forward starts at every squeeze kernel
    fwd, cur = [], []
    for r in rows:
        if "se_squeeze" in r[2] and cur:
            fwd.append(cur)
            cur = []
        cur.append(r)
    fwd.append(cur)
    lines = []
    modes = {}
    for fw in fwd:
        key = tuple(k for _, _, k in fw)
        modes.setdefault(key, []).append(fw)
    for key, fws in sorted(modes.items(), key=lambda kv: -len(kv[1])):
        if len(fws) < 50:
            continue
        fws = fws[-100:]
        n = len(fws)
        period = (fws[-1][0][0] - fws[0][0][0]) / (n - 1) / 1e3
        lines += ["", "### %d launches per forward (%s), mean of the last %d forwards; forward-to-forward period %.1f us" % (
            len(key), "float32" if any("f32" in k for k in key) else "f16x3", n, period), "",
            "| # | kernel | start us | duration us | gap before us |", "|---|---|---|---|---|"]
        tot_d = tot_g = 0.0
        for i, k in enumerate(key):
            st = sum(fw[i][0] - fw[0][0] for fw in fws) / n / 1e3
            du = sum(fw[i][1] - fw[i][0] for fw in fws) / n / 1e3
            gp = sum((fw[i][0] - fw[i - 1][1]) if i else 0 for fw in fws) / n / 1e3
            tot_d += du; tot_g += gp
            lines.append("| %d | `%s` | %.1f | %.1f | %.1f |" % (i, k, st, du, gp))
        end = sum(fw[-1][1] - fw[0][0] for fw in fws) / n / 1e3
        lines.append("| | **first start to last end** | | **%.1f** = %.1f in kernels + %.1f in gaps | |" % (end, tot_d, tot_g))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write("# Batch-1 forward, launch by launch (rocprofv3 --kernel-trace; tools/b1_timeline.py)\n" + txt + "\n")


if __name__ == "__main__":
    if sys.argv[1:2] == ["run"]:
        run()
    else:
        table(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
