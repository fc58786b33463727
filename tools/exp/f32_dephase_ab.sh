# A/B of conv_igemm_f32 with de-phased starts (round 5; measured level, the macro was removed again).  The variant libraries were built with
#   python tools/build_variant.py _dp71 -DDAVO_F32_DEPHASE=71      (conv_igemm_f32_body, in front of the first chunk's loads:
#   if (BN == 128 && LAYER >= 5 && wg_x >= 256 && wg_x < 512) __builtin_amdgcn_s_sleep(DAVO_F32_DEPHASE);)
# Run on the GPU box: gpurun -- 'bash tools/exp/f32_dephase_ab.sh'
set -u
O=gpurun_out
: > $O/r05dd_f32_dephase_ab.log
for rnd in 1 2 3; do
for v in "" _dp71 _dp36; do
  DAVO_LIB_SUFFIX=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-pipelined --no-f32 > $O/_dp.json 2>>$O/r05dd.err || exit 1
  python - "$v" $rnd >> $O/r05dd_f32_dephase_ab.log <<'P'
import json,sys
d=json.load(open('gpurun_out/_dp.json'))
k=d['kernel_avg_ms']
print("round %s lib '%s': %.4f ms/step  %8.1f triplets/s  cnv5 %.4f cnv6 %.4f cnv4 %.4f cnv7 %s" % (sys.argv[2], sys.argv[1] or 'product', d['ms_per_step'], d['value'], k.get('cnv5',0), k.get('cnv6',0), k.get('cnv4',0), k.get('cnv7')))
P
done; done
cat $O/r05dd_f32_dephase_ab.log
