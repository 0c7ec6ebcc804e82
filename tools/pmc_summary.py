#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs per kernel name: sum over dispatches, print per-dispatch means."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for path in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            short = name.split("(")[0].replace("void ", "").replace("ptrk::", "")
            cnt = row.get("Counter_Name")
            val = float(row.get("Counter_Value", 0) or 0)
            tot[short][cnt] += val
            calls[short][cnt] += 1
for k in sorted(tot):
    if not k.startswith("k_"):
        continue
    print("==", k)
    for c in sorted(tot[k]):
        n = calls[k][c]
        print("   %-44s total %.6g   per-dispatch %.6g   (%d dispatches)" % (c, tot[k][c], tot[k][c] / max(n, 1), n))
