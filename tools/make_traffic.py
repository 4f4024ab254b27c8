#!/usr/bin/env python3
"""Per-configuration HBM traffic from tools/profile_round.sh's rocprofv3 passes:
    python tools/make_traffic.py <gpurun_out/prof_<tag>> <tag>   ->  JSON on stdout
For every configuration (c3, c5, c2, c4) and every jpeg_amd kernel of it: launches, mean duration from the
kernel trace, mean FETCH_SIZE / WRITE_SIZE per launch, and the bytes they stand for.  Unit and gfx950 correction as
MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB; FETCH_SIZE tallies the 128-byte requests of
16-B/lane streaming reads at 64 B, so it is doubled.
The record is stamped with the commit and with digests of bench.py and of the decode kernels' sources: bench.py only
reports `roofline.traffic` while the digest still matches (otherwise null)."""
import csv, json, re, sys, os, collections, hashlib, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
root, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")


def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def pmc(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "jpeg_amd" in r["Kernel_Name"]:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        if "jpeg_amd" in r["Name"]:
            out[short(r["Name"])] = {"launches": int(r["Calls"]), "mean_us": round(float(r["AverageNs"]) / 1e3, 2)}
    return out


ALG = {"c3": 402653184, "c5": 4096 * (128 * 48720 + 3 * 1920 * 1080), "c2": 256 * 2048 * 2048,
       "c4": 3 * 4096 * 4096 + 128 * (512 * 512 + 2 * 256 * 256)}
configs = {}
for cfg in ("c3", "c5", "c2", "c4"):
    d = os.path.join(root, cfg)
    if not os.path.exists(os.path.join(d, "kernel_stats.csv")):
        continue
    st = stats(os.path.join(d, "kernel_stats.csv"))
    f, w = pmc(os.path.join(d, "pmc_fetch_size.csv"), "FETCH_SIZE"), pmc(os.path.join(d, "pmc_write_size.csv"), "WRITE_SIZE")
    per = {}
    for k in sorted(set(st) | set(f) | set(w)):
        fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
        per[k] = dict(st.get(k, {}), FETCH_SIZE_KiB=round(fk, 1), WRITE_SIZE_KiB=round(wk, 1), hbm_bytes=int(round((2 * fk + wk) * 1024)))
        if per[k].get("mean_us"):
            per[k]["hbm_GB_per_s"] = round(per[k]["hbm_bytes"] / per[k]["mean_us"] / 1e3, 1)
    # c2 / c4 tools run several shapes: the headline shape is the kernel's largest launch; bench.py's steps are uniform
    rec = {"per_kernel": per, "algorithmic_bytes_per_step": ALG[cfg]}
    if cfg in ("c3", "c5"):
        rec["hbm_bytes_per_step"] = sum(v["hbm_bytes"] for v in per.values())
        rec["step_us_sum_of_kernel_means"] = round(sum(v.get("mean_us", 0.0) for v in per.values()), 2)
    configs[cfg] = rec


def sha16(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(open(os.path.join(ROOT, p), "rb").read())
    return h.hexdigest()[:16]


try:
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    # the GPU box has no .git: `git rev-parse --short HEAD > .commit_stamp` before the gpurun call carries the commit over
    stamp = os.path.join(ROOT, ".commit_stamp")
    commit = os.environ.get("JPEG_AMD_COMMIT") or (open(stamp).read().strip() + " (+ working tree at profiling time)" if os.path.exists(stamp)
                                                   else "unknown (no .git on the GPU box; see the commit that adds this file)")
print(json.dumps({
    "tag": tag, "commit": commit,
    "kernel_source_sha16": sha16(["jpeg_amd/csrc/" + f for f in ("kernels_quad.hip", "kernels_fused.hip", "fused_common.hpp", "dct.hpp", "upsample.hpp", "kernels.hpp", "capi.hip")]),   # = bench.DECODE_SOURCES
    "bench_sha16": sha16(["bench.py"]),
    "method": "rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over each "
              "configuration's command (tools/profile_round.sh); per-launch means; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 "
              "(gfx950: FETCH_SIZE reports half of a 16-B/lane streaming read, MI355X_MICROARCH.md)",
    "configs": configs,
}, indent=1))
