#!/bin/bash
# On the GPU box: kernel-trace the benchmark once per library variant and print the kernel averages.
#   tools/prof_variants.sh <name> [<name> ...]      (variants/libjpeg_hip_<name>.so; "base" = the in-tree build)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset JPEG_HIP_LIB; else export JPEG_HIP_LIB=$R/variants/libjpeg_hip_$v.so; fi
  rm -rf /tmp/pv_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$v -o p -- python $R/bench.py --steps 10 --no-cpu-baseline ${BENCH_ARGS} > /tmp/pv_$v.log 2>&1 || { echo "$v: failed"; tail -5 /tmp/pv_$v.log; exit 1; }
  echo "== $v"; python $R/tools/prof_summary.py /tmp/pv_$v.log /tmp/pv_$v
done
