#!/bin/bash
# A/B timing of library variants on one GPU box (boxes differ by a per cent or two, runs on one box do not):
# usage (on the GPU box): tools/ab_bench.sh "<bench args>" variantA.so variantB.so@NAME=VALUE ...
# (files under jpeg-encoder_amd/variants, built by tools/build_variant.py; @NAME=VALUE sets an environment
# variable for that run).  Runs the bench (no CPU legs) with each variant in turn, twice over, and prints value
# and kernel times.
args=$1; shift
for round in 1 2; do
  for spec in "$@"; do
    v=${spec%%@*}; e=""
    [ "$spec" != "$v" ] && e=${spec#*@}
    env $e JPEG_HIP_LIB=$PWD/jpeg-encoder_amd/variants/$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-e2e $args > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python - "$spec" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
r = d["roofline"]
print("%-36s value %8.0f  %s mean %.4f min %.4f median %.4f ms" % (sys.argv[1], d["value"], r["kernel"], r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"]))
PY
  done
done
