#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/prof_pmc.sh output): per kernel, per counter, mean per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "staged"
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if want not in k:
                continue
            short = k.split("(")[0].split("::")[-1]
            acc[short][(row["Dispatch_Id"], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, d in acc.items():
    per = defaultdict(list)
    for (disp, c), vals in d.items():
        per[c].append(sum(vals))
    print(k)
    for c in sorted(per):
        v = per[c]
        print(f"  {c:28s} mean/dispatch {sum(v)/len(v):18.1f}   (n={len(v)})")
