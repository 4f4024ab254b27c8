#!/bin/bash
# usage (GPU box): bash tools/pmc_generic.sh   -- SQ counter passes of tools/bench_generic.py --only "4:2:0 12-bit" (k_generic_fused / k_generic_encode, 8192 x 8192)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_gen; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/a -o a -- python3 $R/tools/bench_generic.py --only "4:2:0 12-bit" --reps 5 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_VALU_CVT --output-format csv -d $O/b -o b -- python3 $R/tools/bench_generic.py --only "4:2:0 12-bit" --reps 5 > $O/b.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/a $O/b
