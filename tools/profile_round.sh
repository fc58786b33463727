#!/bin/bash
# Round-end measurement set, run ON the GPU box:   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02a'
# Leaves under gpurun_out/: bench_<tag>32.json (full bench line: float32 top level, f16x3 nested as fast_mode, cpu_baseline), bench_<tag>128.json,
# bench_<tag>64_256x832.json, and the rocprofv3 directories prof_{stats,fetch,write,sq}[_b128], which
#   python tools/prof_summary.py --stats gpurun_out/prof_stats --fetch gpurun_out/prof_fetch \
#       --write gpurun_out/prof_write --sq gpurun_out/prof_sq --out profiles/<tag>_bench_b32_f16x3
#   python tools/prof_summary.py --batch 128 --stats gpurun_out/prof_stats_b128 --fetch gpurun_out/prof_fetch_b128 \
#       --write gpurun_out/prof_write_b128 --sq gpurun_out/prof_sq_b128 --out profiles/<tag>_bench_b128_f16x3
# condense into profiles/.  Counters are collected in their own runs, one --pmc pass each (MI355X_MICROARCH.md, HBM);
# the program after `--` is python3 itself (no env / bash hop under the profiler).
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
python3 "$R/bench.py" > "$O/bench_${TAG}32.json" 2> "$O/bench_${TAG}32.err"; echo "bench rc=$?"
python3 "$R/bench.py" --batch 128 --no-cpu-baseline > "$O/bench_${TAG}128.json" 2>/dev/null
python3 "$R/bench.py" --height 256 --width 832 --batch 64 --no-cpu-baseline > "$O/bench_${TAG}64_256x832.json" 2>/dev/null
for f in 32 128 64_256x832; do
  python3 -c "import json; d=json.loads(open('$O/bench_${TAG}$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['whole_path_frac_of_mfma_peak'], d['pipelined']['value'], (d.get('fast_mode') or d.get('reference_arithmetic') or {}).get('value'), (d.get('fast_mode') or d.get('reference_arithmetic') or {}).get('whole_path_frac_of_mfma_peak'))"
done
cd /tmp
rm -rf "$O"/prof_*
for B in 32 128; do
  SFX=""; [ $B = 128 ] && SFX="_b128"
  S="--batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_stats$SFX" -- python3 "$R/bench.py" --batch $B --no-cpu-baseline --no-pipelined > "$O/prof_stats$SFX.log" 2>&1; echo "stats B=$B rc=$?"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/prof_fetch$SFX" -- python3 "$R/bench.py" $S > "$O/prof_fetch$SFX.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/prof_write$SFX" -- python3 "$R/bench.py" $S > "$O/prof_write$SFX.log" 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$O/prof_sq$SFX" -- python3 "$R/bench.py" $S > "$O/prof_sq$SFX.log" 2>&1
  echo "pmc B=$B done"
done
