#!/usr/bin/env python3
"""VGPRs / spills / scratch / LDS / occupancy of every kernel in a .hip file (hipcc -Rpass-analysis=kernel-resource-usage):
    tools/kernel_resources.py jpeg_amd/csrc/kernels_encode.hip [-DMACRO ...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from jpeg_amd.build import EXTRA_FLAGS   # per-source flags of the product build
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", *EXTRA_FLAGS.get(os.path.basename(sys.argv[1]), []),
       "-I", os.path.join(root, "include"),
       "--cuda-device-only", "-c", sys.argv[1], "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name": cur = v; rows[cur] = {}
    elif cur: rows[cur][k] = v
for name, r in rows.items():
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = short.replace("jpeg_amd::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print("%-64s VGPR %4s spill %3s scratch %4s LDS %6s occ %s" % (short, r.get("VGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
                                                                 r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
