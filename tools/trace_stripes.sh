#!/bin/bash
# Kernel trace of one image coded in 4 stripes by 4 contexts on one GPU (tools/trace_stripes.py): the last of three
# rounds, start and end of every kernel relative to the round's first.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02_stripe_trace
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02_stripe_trace -- python3 $R/tools/trace_stripes.py > $R/gpurun_out/r02_stripe_trace.log 2>&1 || { tail -5 $R/gpurun_out/r02_stripe_trace.log; exit 1; }
d=$(dirname $(ls $R/gpurun_out/r02_stripe_trace/*/*kernel_trace.csv | head -1))
python3 $R/tools/trace_overlap.py $d 32 > $R/gpurun_out/r02_stripe_overlap.txt
cat $R/gpurun_out/r02_stripe_overlap.txt
rm -rf $R/gpurun_out/r02_stripe_trace
