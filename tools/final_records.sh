#!/bin/bash
# Everything the profiles/ directory records for the library as built, in one go on one GPU box: micro-benchmarks,
# counters of the coding kernel (three configurations; the ablated builds for the per-phase table), kernel
# statistics, phase stamps, the stripe-chain timelines, bench lines of the five configurations.
# Needs jpeg-encoder_amd/variants/{stamps0,ablate1,ablate2,ablate3}.so (tools/build_variant.py) and the built
# micro-benchmarks.   usage (on the GPU box): tools/final_records.sh <tag>      (results under gpurun_out/<tag>_*)
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
V=$R/jpeg-encoder_amd/variants
set -o pipefail
step() { echo "== $*"; }
step micro-benchmarks and counters of the three configurations
$R/tools/pmc_configs.sh $tag || exit 1
for w in 1 2 3; do timeout -k 10 200 $R/tools/micro/valu_classes $w | grep -E "^(v_fma_f32|v_mul_f32|v_add_f32|v_cvt_f32_ubyte0|v_cvt_u32_f32|v_add_u32|v_lshlrev_b32|v_pk_mul_f32|v_pk_min_u16|s_|two_s|ds_|v_mul_f32_then|v_readfirstlane_then)" >> $O/${tag}_valu_classes_fewer_waves.txt || exit 1; done
timeout -k 10 120 $R/tools/micro/ifetch > $O/${tag}_ifetch.txt || exit 1
step counters of the ablated builds
for a in 1 2 3; do
  JPEG_HIP_LIB=$V/ablate$a.so $R/tools/pmc_coded.sh gpurun_out/${tag}_pmc_ablate$a > /dev/null || exit 1
  cp $O/${tag}_pmc_ablate$a/summary.txt $O/${tag}_coded_pmc_summary_ablate$a.txt
  rm -rf $O/${tag}_pmc_ablate$a
done
python3 $R/tools/phase_counters.py $O/${tag}_coded_pmc_summary_config3.txt $O/${tag}_coded_pmc_summary_ablate1.txt $O/${tag}_coded_pmc_summary_ablate2.txt $O/${tag}_coded_pmc_summary_ablate3.txt > $O/${tag}_phase_counters.txt || exit 1
step kernel statistics
cd $R
tools/prof_bench.sh ${tag} || exit 1
tools/prof_bench.sh ${tag}_config2 --config 2 || exit 1
tools/prof_bench.sh ${tag}_config4 --config 4 || exit 1
step phase stamps
for c in 3 2 4; do
  JPEG_HIP_LIB=$V/stamps0.so timeout -k 10 100 python tools/phase_stamps.py --config $c --images $([ $c = 4 ] && echo 64 || echo 32) >> $O/${tag}_phase_stamps.txt 2>/dev/null || exit 1
done
step stripe chain timelines
tools/trace_chain.sh ${tag}_8 8 || exit 1
tools/trace_chain.sh ${tag}_4 4 || exit 1
tools/trace_chain.sh ${tag}_1 1 || exit 1
step bench lines
for c in 3 2 4 5 1; do
  timeout -k 10 400 python bench.py --config $c > $O/${tag}_bench_config$c.json 2> $O/${tag}_bench_config$c.err || { tail -5 $O/${tag}_bench_config$c.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/${tag}_bench_config$c.json')); r=d['roofline']; print('config $c', d['value'], d['unit'], r['kernel'], r['kernel_ms'], 'frac', r['frac'], 'valu', (d.get('roofline_valu') or {}).get('frac'))"
done
