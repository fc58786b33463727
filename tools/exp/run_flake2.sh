set -e
mkdir -p gpurun_out
L=gpurun_out/r04_flake2.log
: > $L
for v in e8 e9 e10 e11; do
  DAVO_LIB_SUFFIX=_$v timeout -k 10 200 python tools/exp/flake_count.py --batch 4 --tile 0 --reps 300 >> $L 2>&1
done
DAVO_LIB_SUFFIX=_fd timeout -k 10 300 python tools/exp/flake_lanes.py --batch 4 --tile 0 --reps 200 --show 4 >> $L 2>&1
DAVO_LIB_SUFFIX=_e6d timeout -k 10 300 python tools/exp/flake_lanes.py --batch 4 --tile 0 --reps 200 --show 4 >> $L 2>&1
grep -n "^lib\|^wrong\|Error" $L
