set -e
mkdir -p gpurun_out
L=gpurun_out/r04_flake3_product.log
: > $L
timeout -k 10 300 python tools/exp/flake_count.py --batch 4 --tile 0 --reps 1000 >> $L 2>&1
timeout -k 10 300 python tools/exp/flake_count.py --batch 8 --tile 0 --reps 300 >> $L 2>&1
timeout -k 10 300 python tools/exp/flake_count.py --batch 3 --tile 0 --reps 300 >> $L 2>&1
timeout -k 10 300 python tools/exp/flake_count.py --batch 32 --tile -1 --reps 300 >> $L 2>&1
cat $L
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a_tests.log 2>&1; tail -5 gpurun_out/r04a_tests.log
