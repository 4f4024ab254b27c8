#!/usr/bin/env python3
"""A/B of two builds of the library on the 4:2:0 decode: alternating child processes (JPEG_AMD_LIBRARY), several
rounds, median and minimum per build and case.  usage (GPU box): tools/ab_lib.py <exp-name>[,<exp-name>...] [rounds]
(<exp-name>: tools/exp/libjpeg_amd_<exp-name>.so from tools/build_exp.sh; the other side is the product build)"""
import sys, os, subprocess, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("200", "8192", "8192", "1"), ("30", "1920", "1080", "512"), ("300", "4096", "4096", "1"), ("100", "1920", "1080", "64")]
if os.environ.get("AB_CASES"):   # e.g. AB_CASES="100:1920:1080:128,50:2048:2048:32"  (reps:W:H:N)
    CASES = [tuple(c.split(":")) for c in os.environ["AB_CASES"].split(",")]
name = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
libs = {n: os.path.join(ROOT, "tools", "exp", f"libjpeg_amd_{n}.so") for n in name.split(",")}
libs["product"] = ""
res = {}
for r in range(rounds):
    for case in CASES:
        for lib, path in libs.items():
            env = dict(os.environ)
            if path: env["JPEG_AMD_LIBRARY"] = path
            else: env.pop("JPEG_AMD_LIBRARY", None)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_c3.py"), *case], env=env, capture_output=True, text=True).stdout
            us = float(out.split("RGB:")[1].split("us")[0])
            res.setdefault((case, lib), []).append(us)
for case in CASES:
    line = f"{case[3]:>4s} x {case[1]}x{case[2]}:"
    for lib in libs:
        v = res[(case, lib)]
        line += f"  {lib} median {statistics.median(v):8.1f} min {min(v):8.1f} us"
    print(line, flush=True)
