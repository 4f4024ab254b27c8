#!/bin/bash
# usage (GPU box): bash tools/pmc_c3.sh <tag> [run_c3 args]   -- SQ counter passes of tools/run_c3.py of the library JPEG_AMD_LIBRARY selects (default: the product build)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_$tag; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/run_c3.py "$@" > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/a -o a -- python3 $R/tools/run_c3.py "$@" > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_VALU_CVT --output-format csv -d $O/b -o b -- python3 $R/tools/run_c3.py "$@" > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $O/c -o c -- python3 $R/tools/run_c3.py "$@" > $O/c.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/tools/run_c3.py "$@" > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/tools/run_c3.py "$@" > $O/w.log 2>&1
cd $R
grep -h "us per call" $O/*.log
grep -E "k_band|k_luma|k_chroma" $O/kt/kt_kernel_stats.csv | cut -c1-160
python3 tools/pmc_summary.py $O/a $O/b $O/c $O/f $O/w
