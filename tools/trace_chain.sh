#!/bin/bash
# Kernel trace of one 32768x32768 image coded in 8 stripes by 8 contexts on one GPU through the stripe chain
# (tools/trace_chain.py): every kernel of the last pass, start and end relative to the pass's first kernel.
# usage (on the GPU box): tools/trace_chain.sh <tag> [stripes]
tag=$1; n=${2:-8}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/tools/trace_chain.py $n > $R/gpurun_out/${tag}_trace.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_trace.log; exit 1; }
d=$(dirname $(ls $R/gpurun_out/${tag}_trace/*/*kernel_trace.csv | head -1))
python3 - "$d" $n > $R/gpurun_out/${tag}_chain_timeline.txt <<'PY'
import csv, glob, sys
rows = []
for p in glob.glob(sys.argv[1] + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(p)):
        if "jpg::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("jpg::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0][:40], r.get("Queue_Id", "?")))
rows.sort()
n = int(sys.argv[2])
# kernels of one pass: 9 per stripe (k_mcu_dc, k_coded, k_tail, k_chain_place, k_group_sums, k_stream, k_chain_offset,
# k_chain_copy, k_chain_done); the last pass starts at the n-th last k_mcu_dc
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_mcu_dc")]
last = rows[starts[-n]:]
t0 = last[0][0]
print("kernels of the last pass of %d stripes under rocprofv3 --kernel-trace (us from the first kernel's start): start, end, duration, queue, kernel" % n)
for s, e, name, q in last:
    print("%9.1f %9.1f  %7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
print("pass: %.1f us from the first kernel's start to the last kernel's end" % ((max(e for _, e, _, _ in last) - t0) / 1e3))
PY
cat $R/gpurun_out/${tag}_trace.log | tail -2
tail -1 $R/gpurun_out/${tag}_chain_timeline.txt
rm -rf $R/gpurun_out/${tag}_trace
