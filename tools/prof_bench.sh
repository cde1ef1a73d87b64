#!/bin/bash
# rocprofv3 --kernel-trace --stats over bench.py; keeps the kernel stats csv and the printed JSON line.
# usage (on the GPU box): tools/prof_bench.sh <tag> [bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python3 $R/bench.py --no-cpu-baseline --no-e2e "$@" > $R/gpurun_out/${tag}_prof.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_prof.log; exit 1; }
f=$(ls $R/gpurun_out/${tag}_prof/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/${tag}_kernel_stats.csv
grep '^{' $R/gpurun_out/${tag}_prof.log > $R/gpurun_out/${tag}_kernel_stats_bench_line.json
python3 $R/tools/prof_summary.py $R/gpurun_out/${tag}_prof.log $(dirname $f)
rm -rf $R/gpurun_out/${tag}_prof
