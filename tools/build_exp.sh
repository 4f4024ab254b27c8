#!/bin/bash
# usage: tools/build_exp.sh <name> [-DMACRO ...]   -> tools/exp/libjpeg_amd_<name>.so
# An experimental build of the library with extra macros (e.g. -DJA_PHASE_PROFILE); run with
#   JPEG_AMD_LIBRARY=tools/exp/libjpeg_amd_<name>.so python tools/bench_variants.py ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=tools/exp/obj_$name; mkdir -p $out
for f in kernels_stage.hip kernels_fused.hip kernels_quad.hip kernels_encode.hip kernels_generic.hip capi.hip entropy.cpp entropy_encode.cpp; do
  extra=""; case $f in kernels_generic.hip|kernels_encode.hip|kernels_stage.hip) extra="-fno-slp-vectorize";; esac   # (jpeg_amd/build.py EXTRA_FLAGS)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $extra -I include "$@" -c jpeg_amd/csrc/$f -o $out/${f%.*}.o 2> $out/${f%.*}.log &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libjpeg_amd_$name.so $out/*.o
ls -la tools/exp/libjpeg_amd_$name.so
