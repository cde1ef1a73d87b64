#!/bin/bash
# Everything the profiles/ directory records for the library as built, in one go on one GPU box:
# micro-benchmarks, counters and issue budget, kernel statistics, phase stamps, bench lines of the four configurations.
# usage (on the GPU box): tools/final_records.sh <tag>      (results under gpurun_out/<tag>_*)
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
set -o pipefail
step() { echo "== $*"; }
step micro-benchmarks
timeout -k 10 120 $R/tools/micro/valu_classes 4 > $O/${tag}_valu_classes.txt || exit 1
for w in 1 2 3; do timeout -k 10 120 $R/tools/micro/valu_classes $w | grep -E "^(v_fma_f32|v_mul_f32|v_add_f32|v_cvt_f32_ubyte0|v_cvt_u32_f32|v_add_u32|v_lshlrev_b32|v_pk_mul_f32|v_pk_min_u16|s_|two_s|ds_|v_mul_f32_then|v_readfirstlane_then)" >> $O/${tag}_valu_classes_fewer_waves.txt || exit 1; done
timeout -k 10 120 $R/tools/micro/ifetch > $O/${tag}_ifetch.txt || exit 1
step counters
$R/tools/pmc_coded.sh gpurun_out/${tag}_pmc > /dev/null || exit 1
cp $O/${tag}_pmc/summary.txt $O/${tag}_coded_pmc_summary.txt
step kernel statistics
cd $R
tools/prof_bench.sh ${tag} || exit 1
tools/prof_bench.sh ${tag}_config2 --config 2 || exit 1
tools/prof_bench.sh ${tag}_config4 --config 4 || exit 1
step phase stamps
JPEG_HIP_LIB=$R/jpeg-encoder_amd/variants/stamps0.so timeout -k 10 100 python tools/phase_stamps.py > $O/${tag}_phase_stamps.txt 2>/dev/null || exit 1
JPEG_HIP_LIB=$R/jpeg-encoder_amd/variants/stamps0.so timeout -k 10 100 python tools/phase_stamps.py --config 2 >> $O/${tag}_phase_stamps.txt 2>/dev/null || exit 1
JPEG_HIP_LIB=$R/jpeg-encoder_amd/variants/stamps0.so timeout -k 10 100 python tools/phase_stamps.py --config 4 --images 64 >> $O/${tag}_phase_stamps.txt 2>/dev/null || exit 1
step bench lines
for c in 3 2 4 5; do
  timeout -k 10 400 python bench.py --config $c > $O/${tag}_bench_config$c.json 2> $O/${tag}_bench_config$c.err || { tail -5 $O/${tag}_bench_config$c.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/${tag}_bench_config$c.json')); r=d['roofline']; print('config $c', d['value'], d['unit'], r['kernel'], r['kernel_ms'], 'frac', r['frac'], 'valu', (d.get('roofline_valu') or {}).get('frac'))"
done
