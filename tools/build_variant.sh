#!/bin/bash
# usage: tools/build_variant.sh <name> [extra compiler flags...]
# Builds rust-raytracer_amd/librtx_hip_<name>.so from the same sources with extra flags (experiments).
set -e
name=$1; shift
cd "$(dirname "$0")/../rust-raytracer_amd/csrc"
obj=/tmp/rtx_variant_$name
mkdir -p $obj
for f in rtx_api rtx_bvh rtx_trace rtx_shade rtx_comm; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wall -Wno-unused-function "$@" -c $f.hip -o $obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librtx_hip_$name.so $obj/*.o -ldl
ls -la ../librtx_hip_$name.so
