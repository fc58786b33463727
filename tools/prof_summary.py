#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small summaries kept under profiles/.

    python tools/prof_summary.py --stats gpurun_out/prof_stats --fetch gpurun_out/prof_fetch \
        --write gpurun_out/prof_write --sq gpurun_out/prof_sq --out profiles/r01_bench_b32

Writes <out>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim) and <out>_summary.md
(per-kernel average duration + PMC-derived HBM traffic and MFMA utilisation).
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB, collected in
separate --pmc passes; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane)
coalesced reads, so the read side is doubled; WRITE_SIZE is exact for 16-B/lane stores."""
import argparse
import collections
import csv
import glob
import os
import shutil


def find(d, pat):
    g = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return max(g, key=os.path.getmtime) if g else None      # newest run if several were merged


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    f = find(d, "*counter_collection.csv") if d else None
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats"); ap.add_argument("--fetch"); ap.add_argument("--write"); ap.add_argument("--sq")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    ap.add_argument("--batch", type=int, default=32, help="triplets per step of the profiled run (recorded in the traffic JSON)")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    lines = ["# rocprofv3 summary: %s" % os.path.basename(a.out), "", a.note, ""]
    stats = find(a.stats, "*kernel_stats.csv") if a.stats else None
    if stats:
        shutil.copy(stats, a.out + "_kernel_stats.csv")
        lines += ["## rocprofv3 --kernel-trace --stats (verbatim copy: %s_kernel_stats.csv)" % os.path.basename(a.out), "",
                  "| kernel | calls | avg us | total % |", "|---|---|---|---|"]
        for r in csv.DictReader(open(stats)):
            lines.append("| `%s` | %s | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
        lines.append("")
    fe, wr, sq = counters(a.fetch), counters(a.write), counters(a.sq)
    if fe or wr:
        lines += ["## HBM traffic per launch (PMC; separate --pmc passes)", "",
                  "| kernel | FETCH_SIZE KiB (raw) | read MB (x2 gfx950 correction) | WRITE_SIZE KiB | write MB | total MB |",
                  "|---|---|---|---|---|---|"]
        for k in sorted(set(fe) | set(wr)):
            f = fe[k].get("FETCH_SIZE", []); w = wr[k].get("WRITE_SIZE", [])
            fa = sum(f) / len(f) if f else 0.0
            wa = sum(w) / len(w) if w else 0.0
            rmb, wmb = 2 * fa * 1024 / 1e6, wa * 1024 / 1e6
            lines.append("| `%s` | %.0f | %.1f | %.0f | %.1f | %.1f |" % (k, fa, rmb, wa, wmb, rmb + wmb))
        lines.append("")
    if sq:
        lines += ["## SQ counters per launch", "",
                  "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8 XCDs); "
                  "clock = GRBM_GUI_ACTIVE/8 / duration.", "",
                  "| kernel | MFMA busy cycles | GRBM_GUI_ACTIVE | MFMA util | SQ_WAIT_ANY / SQ_WAVE_CYCLES | SQ_LDS_BANK_CONFLICT |",
                  "|---|---|---|---|---|---|"]
        for k in sorted(sq):
            c = {n: sum(v) / len(v) for n, v in sq[k].items()}
            gui = c.get("GRBM_GUI_ACTIVE", 0.0)
            util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * gui / 8) if gui else 0.0
            wait = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0.0
            lines.append("| `%s` | %.3g | %.3g | %.3f | %.3f | %.0f |" % (k, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), gui, util, wait,
                                                                      c.get("SQ_LDS_BANK_CONFLICT", 0)))
        lines.append("")
    if fe or wr:
        import json
        tr = {}
        for k in sorted(set(fe) | set(wr)):
            f = fe[k].get("FETCH_SIZE", []); w = wr[k].get("WRITE_SIZE", [])
            tr[k] = {"read_bytes": 2 * (sum(f) / len(f) if f else 0.0) * 1024, "write_bytes": (sum(w) / len(w) if w else 0.0) * 1024}
            c = {n: sum(v) / len(v) for n, v in sq.get(k, {}).items()} if sq else {}
            if c.get("GRBM_GUI_ACTIVE"):        # matrix pipe busy fraction of the launch (SQ pass), beside its traffic: bench.py quotes both
                tr[k]["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
        json.dump({"source": os.path.basename(a.out), "batch": a.batch, "note": "per launch; FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE exact",
                   "kernels": tr}, open(a.out + "_traffic.json", "w"), indent=1)
    open(a.out + "_summary.md", "w").write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
