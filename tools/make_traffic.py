#!/usr/bin/env python3
"""profiles/traffic_latest.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
bench.py's C3 step:  python tools/make_traffic.py <fetch counter_collection.csv> <write ...csv>
Unit and gfx950 correction as prescribed by MI355X_MICROARCH.md (HBM section): both counters are
in KiB; FETCH_SIZE tallies the 128-byte requests of 16-B/lane streaming reads at 64 B, so it is
doubled.  Per-dispatch means are summed over the kernels of one fused decode step."""
import csv, json, re, sys, collections

def means(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "jpeg_amd" not in r["Kernel_Name"]:
            continue
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        acc[m.group(1) if m else r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}

f, w = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
per = {k: {"FETCH_SIZE": round(f.get(k, 0.0), 2), "WRITE_SIZE": round(w.get(k, 0.0), 2)} for k in sorted(set(f) | set(w))}
total = sum((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 for v in per.values())
print(json.dumps({
    "workload": "c3",
    "hbm_bytes_per_step": int(round(total)),
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 50 --warmup 10 "
              "--no-extras --no-cpu` (tools/profile_round.sh); per-dispatch means summed over the step's two kernels; "
              "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE reports half of a 16-B/lane streaming read; "
              "the 4-B/lane chroma-row DMA of k_luma_fused is uncalibrated)",
    "per_kernel_KiB": per,
    "algorithmic_bytes_per_step": 402653184,
}, indent=1))
