#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag>      e.g. r02
# The round's committed evidence, for EVERY number bench.py reports:
#   c3  bench.py --workload c3                (k_quad420<1, 32, true, false>: the static walk, the headline)
#   c5  bench.py --workload c5 (4096 x 1080p) (k_quad420<1, 32, true, true> + <1, 16, true, true>: the mixed cut, ticket walk)
#   c2  tools/bench_idct.py --units 2048      (k_idct_plane: IDCT + dequant only, 2^22 blocks)
#   c4  tools/bench_encode.py --only 4:2:0    (k_encode_fused, 4096 x 4096)
# For each: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes.
# Output under gpurun_out/prof_<tag>/<cfg>/{kernel_stats.csv,pmc_fetch_size.csv,pmc_write_size.csv,run.log};
# tools/make_traffic.py turns them into profiles/<tag>_traffic.json (+ profiles/traffic_latest.json for c3).
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_cfg() {
  cfg=$1; shift
  mkdir -p $O/$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$cfg/kt -o kt -- python3 "$@" > $O/$cfg/run.log 2>&1
  cp $O/$cfg/kt/kt_kernel_stats.csv $O/$cfg/kernel_stats.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/$cfg/pf -o pf -- python3 "$@" > $O/$cfg/pf.log 2>&1
  cp $O/$cfg/pf/pf_counter_collection.csv $O/$cfg/pmc_fetch_size.csv
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/$cfg/pw -o pw -- python3 "$@" > $O/$cfg/pw.log 2>&1
  cp $O/$cfg/pw/pw_counter_collection.csv $O/$cfg/pmc_write_size.csv
  rm -rf $O/$cfg/kt $O/$cfg/pf $O/$cfg/pw
}
run_cfg c3 $R/bench.py --workload c3 --steps 50 --warmup 10 --no-extras --no-cpu --traffic none --valu none --no-c5-job --no-parity
run_cfg c5 $R/bench.py --workload c5 --steps 5 --warmup 2 --no-extras --no-cpu --traffic none --valu none --no-parity
run_cfg c2 $R/tools/bench_idct.py --units 2048 --only-main
run_cfg c4 $R/tools/bench_encode.py --only 4:2:0
cd $R
python3 tools/make_traffic.py $O $tag > $O/traffic.json
cp $O/traffic.json profiles/${tag}_traffic.json
python3 - <<PY
import json
t = json.load(open("$O/traffic.json"))
c3 = dict(t["configs"]["c3"]); c3.update({k: t[k] for k in ("commit", "kernel_source_sha16", "bench_sha16", "method")}); c3["workload"] = "c3"
c3["c5"] = dict(t["configs"]["c5"], images=4096)     # bench.py --workload c5 scales it by the rank's share of the images
json.dump(c3, open("profiles/traffic_latest.json", "w"), indent=1)
PY
for c in c3 c5 c2 c4; do cp $O/$c/kernel_stats.csv profiles/${tag}_${c}_kernel_stats.csv; cp $O/$c/pmc_fetch_size.csv profiles/${tag}_${c}_pmc_fetch_size.csv; cp $O/$c/pmc_write_size.csv profiles/${tag}_${c}_pmc_write_size.csv; done
python3 bench.py > $O/bench.json 2> $O/bench.log
cp $O/bench.json profiles/${tag}_bench.json
python3 - <<PY
# the record has to agree with itself: the profiler's mean duration of the headline kernel against the wall time per step of the bench line
import csv, json
b = json.load(open("$O/bench.json"))
k = next(float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open("$O/c3/kernel_stats.csv")) if "k_quad420" in r["Name"])
print(f"C3: rocprofv3 mean {k * 1e3:.1f} us, bench.py {b['ms_per_step'] * 1e3:.1f} us per step:", "agree" if abs(k / b["ms_per_step"] - 1) < 0.05 else "DISAGREE (a noisy box: run the record again)")
PY
cat $O/traffic.json | head -80
cut -c1-1500 $O/bench.json
# profiles/ is not copied back from the GPU box, gpurun_out/ is: leave the files to commit there
mkdir -p $O/to_profiles && cp profiles/${tag}_* profiles/traffic_latest.json $O/to_profiles/
