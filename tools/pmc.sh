#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   (run on the GPU box; output under gpurun_out/pmc_<tag>)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_${tag}_a -o a -- python3 $R/tools/bench_variants.py "$@" > $R/gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${tag}_b -o b -- python3 $R/tools/bench_variants.py "$@" > $R/gpurun_out/pmc_${tag}_b.log 2>&1
