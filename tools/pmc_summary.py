#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs for the pair kernel: per-launch averages of every counter."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "pair_kernel" not in k:
            continue
        a = acc[(k.split("(")[0][-48:], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted(acc)
for k in names:
    s, n = acc[k]
    print(f"{k[0]:50s} {k[1]:28s} per-launch avg {s / n:18.1f}  (n={n})")
