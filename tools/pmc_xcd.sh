#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export JPEG_AMD_XCD_IMAGES=$v
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_xcd$v -o f -- python3 $R/tools/run_c3.py 6 1920 1080 512 > $R/gpurun_out/pmc_xcd$v.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, collections, glob
for v in (0, 1):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_xcd{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "jpeg_amd" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, vals in acc.items(): print("XCD_IMAGES=%d" % v, k, "launches", len(vals), "mean FETCH_SIZE KiB", round(sum(vals) / len(vals)))
PY
