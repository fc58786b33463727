# A/B of libdavo_hip.so against variant libraries built by tools/build_variant.py, float32 top level, alternating rounds on one box:
#   gpurun -- 'bash tools/exp/f32_variant_ab.sh r05de_f32_agpr_ab _agpr'
set -u
TAG=$1; shift
O=gpurun_out
: > $O/$TAG.log
for rnd in 1 2 3; do
for v in "" "$@"; do
  DAVO_LIB_SUFFIX=$v python bench.py --steps 100 --warmup 5 --no-pipelined --no-f32 --no-cpu-baseline > $O/_v.json 2>>$O/$TAG.err || exit 1
  python - "$v" $rnd >> $O/$TAG.log <<'P'
import json,sys
d=json.load(open('gpurun_out/_v.json'))
k=d['kernel_avg_ms']
print("round %s lib '%s': %.4f ms/step  %8.1f triplets/s  cnv4 %.4f cnv5 %.4f cnv6 %.4f cnv7 %s  max abs err vs oracle %.3g" % (sys.argv[2], sys.argv[1] or 'product', d['ms_per_step'], d['value'], k.get('cnv4',0), k.get('cnv5',0), k.get('cnv6',0), k.get('cnv7'), d['max_abs_err_vs_oracle']))
P
done; done
cat $O/$TAG.log
