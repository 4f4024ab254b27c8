#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag>
# Produces under gpurun_out/prof_<tag>/: kernel_stats.csv (rocprofv3 --kernel-trace --stats of the
# bench command), pmc_fetch_size.csv / pmc_write_size.csv (separate --pmc passes), bench.json.
# tools/make_traffic.py turns the two PMC files into profiles/traffic_latest.json.
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --steps 50 --warmup 10 --no-extras --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $BENCH > $O/kt.log 2>&1
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python3 $BENCH > $O/pf.log 2>&1
cp $O/pf/pf_counter_collection.csv $O/pmc_fetch_size.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- python3 $BENCH > $O/pw.log 2>&1
cp $O/pw/pw_counter_collection.csv $O/pmc_write_size.csv
cd $R && python3 tools/make_traffic.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv > $O/traffic_latest.json
cp $O/traffic_latest.json $R/profiles/traffic_latest.json   # so that the bench line below carries it
python3 bench.py > $O/bench.json 2> $O/bench.log
grep -E "k_luma|k_chroma|Name" $O/kernel_stats.csv | cut -c1-200
cat $O/traffic_latest.json | head -30
cut -c1-900 $O/bench.json
