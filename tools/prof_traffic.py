#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/prof.sh into profiles/traffic.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024:
  - both counters are in KiB (rocprofv3 basic counters);
  - on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read
    (it is TCC_EA0_RDREQ x 64 B while the requests are 128 B; MI355X_MICROARCH.md, HBM),
    so it is doubled; WRITE_SIZE is exact for streaming stores.
Infinity-Cache hits are counted by these fabric-side counters, so for a matrix that does
not fit the 256 MiB cache this is an upper bound on DRAM traffic.

usage: prof_traffic.py <prof dir> <workload name> [kernel-name-substring=short-name ...]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, workload = sys.argv[1], sys.argv[2]
names = dict(a.split("=") for a in sys.argv[3:]) or {"csr_stream<": "csr_stream"}


def mean_counter(sub, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = mean_counter("pmc_fetch", "FETCH_SIZE")
write = mean_counter("pmc_write", "WRITE_SIZE")
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
table = json.load(open(out_path)) if os.path.exists(out_path) else {}
for kernel, f in fetch.items():
    for sub, short in names.items():
        if sub in kernel:
            w = write.get(kernel, 0.0)
            table[f"{short}|{workload}"] = int((2 * f + w) * 1024)
            print(f"{short}|{workload}: FETCH_SIZE={f:.0f} KiB WRITE_SIZE={w:.0f} KiB -> {(2 * f + w) * 1024 / 1e6:.1f} MB")
json.dump(table, open(out_path, "w"), indent=1, sort_keys=True)
