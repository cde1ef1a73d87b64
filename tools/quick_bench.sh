#!/bin/bash
# parity tests of the fused kernels, then the headline bench without the CPU legs; prints value and kernel times
# usage (on the GPU box): tools/quick_bench.sh <tag> [bench args]
tag=$1; shift
timeout -k 10 400 python -u -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${tag}_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --no-cpu-baseline --no-e2e "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
r = d["roofline"]
print("value %.0f MP/s  %s mean %.4f min %.4f median %.4f ms  frac %.4f" % (d["value"], r["kernel"], r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"], r["frac"]))
f = d.get("roofline_front_kernel")
if f: print("k_front %.2f us (min %.2f) frac %.3f" % (f["kernel_ms"] * 1e3, f["kernel_ms_min"] * 1e3, f["frac"]))
PY
