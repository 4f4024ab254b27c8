#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)."""
import csv, sys, collections, glob
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "jpeg_amd" not in k: continue
            import re
            m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", k); k = m.group(1) if m else k[:40]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k)
            for c, v in sorted(cs.items()):
                print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
