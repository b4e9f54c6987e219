#!/usr/bin/env python3
"""Condense rocprofv3 csv output (tools/prof.sh) into one text summary per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = []
def newest(pattern):
    """the newest file only: a merged gpurun_out/ keeps earlier runs' files beside the last one's"""
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1:]


for f in newest(os.path.join(root, "trace", "**", "*kernel_stats.csv")):
    out.append(f"== {os.path.relpath(f, root)}")
    out.append(open(f).read().strip())
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in newest(os.path.join(d, "**", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out.append(f"== {os.path.basename(d)} (mean per dispatch; n dispatches)")
    for k, cs in acc.items():
        for c, v in sorted(cs.items()):
            out.append(f"{k:60s} {c:32s} {sum(v)/len(v):18.1f}  n={len(v)}")
print("\n".join(out))
