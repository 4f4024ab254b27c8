#!/bin/bash
# hipcc the stand-alone hardware probes next to their sources (run on the build host; the
# binaries travel to the GPU box with the working tree):  tools/build_probes.sh [name ...]
set -e
cd "$(dirname "$0")"
names=${@:-probe_bw probe_phase probe_pk probe_xcd probe_store_pattern probe_fetch probe_idct probe_isa probe_mix probe_rates probe_rtz probe_sdwa verify_div probe_stream probe_clock}
for n in $names; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ../jpeg_amd/csrc -I ../include -o $n $n.hip
  echo built tools/$n
done
