#!/usr/bin/env python3
"""Per-kernel means of every counter in one or more rocprofv3 --pmc output directories:  python tools/pmc_table.py dir [dir ...]"""
import csv, glob, os, sys
from collections import defaultdict
tab = defaultdict(lambda: defaultdict(list))
for root in sys.argv[1:]:
    per = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per.items():
        for c, v in cs.items():
            tab[k][c].append(v)
for k in sorted(tab):
    if not any(x in k for x in ("x3", "wgrad", "render", "dgrad", "finish")):
        continue
    print(k)
    for c in sorted(tab[k]):
        v = tab[k][c]
        print(f"   {c:34s} {sum(v) / len(v):16.0f}   (n={len(v)})")
