#!/bin/bash
# usage (GPU box): bash tools/ab_encode.sh "<bench_encode args>" name... -- tools/bench_encode.py alternately with each build
# (tools/exp/libjpeg_amd_<name>.so; "product" = the product build), three rounds on one box
args=$1; shift
for r in 1 2 3; do
  for l in "$@"; do
    lib=""; [ "$l" != product ] && lib=tools/exp/libjpeg_amd_$l.so
    JPEG_AMD_LIBRARY=$lib python3 tools/bench_encode.py $args 2>/dev/null | sed "s/^/$l: /"
  done
done
