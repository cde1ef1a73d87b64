#!/bin/bash
# Counters of the coding kernel for the named configurations 3, 2 and 4 (tools/pmc_coded.sh each: one rocprofv3 --pmc
# process per counter set) and the instruction micro-benchmark, on one GPU box.
# usage (on the GPU box): tools/pmc_configs.sh <tag>      (results under gpurun_out/<tag>_*)
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout -k 10 300 $R/tools/micro/valu_classes 4 > $O/${tag}_valu_classes.txt || exit 1
# images per launch of the counter runs: enough tiles per launch that the device is full for most of it
# (1920x1080 4:2:2 has 270 tiles per image: 64 images; 3840x2160 4:4:4 2160: 16; 7680x4320 4:2:0 8100: 4)
for c in 3 2 4; do
  case $c in 3) extra="";; 2) extra="--batch 32 --resident 16 --per-launch 16";; 4) extra="--batch 128 --resident 64 --per-launch 64";; esac
  $R/tools/pmc_coded.sh gpurun_out/${tag}_pmc_config$c --config $c $extra > /dev/null || exit 1
  cp $O/${tag}_pmc_config$c/summary.txt $O/${tag}_coded_pmc_summary_config$c.txt
  rm -rf $O/${tag}_pmc_config$c
done
