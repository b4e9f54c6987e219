#!/usr/bin/env python3
"""Join the rocprofv3 passes of tools/sweep_pmc.sh with the configuration list."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep"
cfg = json.load(open("gpurun_out/sweep_configs.json"))
L, names, algo = cfg["launches"], cfg["configs"], cfg["algo_bytes"]
table = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(root, "pass_*"))):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "csr_" in r["Kernel_Name"] and "long" not in r["Kernel_Name"]]
    by_counter = defaultdict(list)
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        by_counter[r["Counter_Name"]].append(r)
    for counter, rs in by_counter.items():
        if len(rs) != L * len(names):
            print(f"# {counter}: {len(rs)} dispatches, expected {L * len(names)}", file=sys.stderr)
            continue
        for i, name in enumerate(names):
            grp = rs[i * L + 1:(i + 1) * L]  # drop the first launch of each configuration
            table[name][counter] = sum(float(r["Counter_Value"]) for r in grp) / len(grp)
            table[name]["us"] = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp) / len(grp) / 1e3
counters = sorted({c for v in table.values() for c in v} - {"us"})
print(f"{'config':36s} {'us':>7s} {'GB/s':>7s} " + " ".join(f"{c[:14]:>14s}" for c in counters))
for name in names:
    v = table.get(name, {})
    if "us" not in v:
        continue
    print(f"{name:36s} {v['us']:7.1f} {algo / v['us'] / 1e3:7.0f} " + " ".join(f"{v.get(c, float('nan')):14.0f}" for c in counters))
