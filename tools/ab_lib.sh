#!/bin/bash
# usage (GPU box): bash tools/ab_lib.sh <exp-name> [run_c3 args]  -- tools/run_c3.py alternately with the product library and
# tools/exp/libjpeg_amd_<exp-name>.so (three rounds each, same box, same call)
name=$1; shift
for r in 1 2 3; do
  JPEG_AMD_LIBRARY=tools/exp/libjpeg_amd_$name.so python3 tools/run_c3.py "$@" | sed "s/^/$name: /"
  python3 tools/run_c3.py "$@" | sed "s/^/product: /"
done
