#!/usr/bin/env python3
"""Turn a run of tools/probe_valu_classes under `rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES` into the cost
table of tools/valu_roofline.py: issue cycles per instruction and SIMD IN SQ CYCLES -- the unit `cycles elapsed`
(SQ_BUSY_CYCLES / 32) of a profiled kernel is counted in -- for every probed opcode at 8, 4 and 3 waves per SIMD.
    tools/valu_calibrate.py <pmc dir> <iterations> [out.json]
(the probe launches, per opcode and waves per SIMD, a 50-iteration warm-up and the measured launch: the second dispatch of every
(kernel, grid size) pair is taken; instructions per SIMD = waves per SIMD x iterations x 32)"""
import csv, glob, json, os, re, sys
from collections import defaultdict

d, iters = sys.argv[1], int(sys.argv[2])
out = sys.argv[3] if len(sys.argv) > 3 else None
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_valu_classes.hip")).read()
names = re.search(r"kNames\[kOps\] = \{(.*?)\};", src, re.S).group(1)
names = [n.strip().strip('"') for n in names.replace("\n", " ").split(",") if n.strip()]
rows = defaultdict(lambda: defaultdict(dict))   # (op id, grid) -> dispatch id -> counter -> value
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_op<(\d+)>", r["Kernel_Name"])
        if m:
            rows[(int(m.group(1)), int(r["Grid_Size"]))][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[(int(m.group(1)), int(r["Grid_Size"]))][int(r["Dispatch_Id"])]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
grids = sorted({g for _, g in rows}, reverse=True)          # 8, 4, 3 waves per SIMD
cus = grids[0] // (8 * 256)
table = {}
print(f"{cus} CUs; SQ cycles per instruction and SIMD = (SQ_BUSY_CYCLES / 32) / (waves per SIMD x {iters} x 32)")
print(f"{'id':>3s} {'opcode':22s} {'@8 waves':>9s} {'@4 waves':>9s} {'@3 waves':>9s} {'GHz @8':>7s}")
for i, name in enumerate(names):
    cyc = []
    ghz = None
    for g in grids:
        disp = rows.get((i, g))
        if not disp: cyc.append(None); continue
        last = disp[max(disp)]
        wps = g // (cus * 256)
        elapsed = last["SQ_BUSY_CYCLES"] / 32.0
        cyc.append(elapsed / (wps * iters * 32))
        if g == grids[0]: ghz = elapsed / last["ns"]
    table[name] = {"w8": cyc[0], "w4": cyc[1], "w3": cyc[2]}
    print(f"{i:3d} {name:22s} " + " ".join(f"{c:9.3f}" if c else "        -" for c in cyc) + (f" {ghz:7.3f}" if ghz else ""))
if out:
    json.dump({"unit": "SQ cycles per wave64 instruction and SIMD (SQ_BUSY_CYCLES / 32 of the probe kernel / instructions per SIMD)",
               "source": "tools/probe_valu_classes.hip under rocprofv3 --pmc SQ_BUSY_CYCLES, tools/valu_calibrate.py", "ops": table}, open(out, "w"), indent=1)
