#!/bin/bash
# Builds an experimental copy of libjpeg_hip.so with extra compiler flags:
#   tools/build_variant.sh <name> [flags...]  ->  gpurun_out/variants/libjpeg_hip_<name>.so
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/variants; mkdir -p $O/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -I$R/include -I$R/jpeg-encoder_amd/csrc"
for f in front coded staged entropy api; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $R/jpeg-encoder_amd/csrc/$f.hip -o $O/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libjpeg_hip_$name.so $O/obj_$name/*.o
rm -rf $O/obj_$name
echo built $O/libjpeg_hip_$name.so
