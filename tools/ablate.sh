#!/bin/bash
# sustained (N-rep) step time of the 8192 x 8192 4:2:0 decode for a list of experimental builds (tools/exp/libjpeg_amd_<name>.so),
# the product build first and last; usage (GPU box): tools/ablate.sh [reps] name...
reps=${1:-200}; shift
export JPEG_AMD_DYNAMIC=${JPEG_AMD_DYNAMIC:-0}
for round in 1 2; do
  echo -n "product: "; python tools/run_c3.py $reps 2>/dev/null
  for n in "$@"; do echo -n "$n: "; JPEG_AMD_LIBRARY=tools/exp/libjpeg_amd_$n.so python tools/run_c3.py $reps 2>/dev/null; done
done
