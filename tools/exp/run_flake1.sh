set -e
mkdir -p gpurun_out
L=gpurun_out/r04_flake1.log
: > $L
for v in fa fb e1 e2 e5 e6 e7; do
  DAVO_LIB_SUFFIX=_$v timeout -k 10 200 python tools/exp/flake_count.py --batch 4 --tile 0 --reps 300 >> $L 2>&1
done
DAVO_LIB_SUFFIX=_fd timeout -k 10 300 python tools/exp/flake_lanes.py --batch 4 --tile 0 --reps 200 >> $L 2>&1
tail -60 $L
