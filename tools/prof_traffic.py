#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/prof.sh into profiles/traffic.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024:
  - both counters are in KiB (rocprofv3 basic counters);
  - on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read
    (it is TCC_EA0_RDREQ x 64 B while the requests are 128 B; MI355X_MICROARCH.md, HBM),
    so it is doubled; WRITE_SIZE is exact for streaming stores.
Infinity-Cache hits are counted by these fabric-side counters, so for a matrix that does
not fit the 256 MiB cache this is an upper bound on DRAM traffic.

Every entry is stamped with what it was measured ON -- the sha of the kernels header, the bytes of
the kernel's own format and its workgroup count, all taken from the JSON line the profiled bench
printed (trace.log) -- so that bench.py hands it out only for that same kernel revision and plan,
and `null` for anything else (bench.measured_traffic).

usage: prof_traffic.py <prof dir> <profile tag>
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha  # noqa: E402

root, tag = sys.argv[1], sys.argv[2]


def bench_line(log):
    for line in reversed(open(log).read().splitlines()):
        if line.lstrip().startswith("{") and '"roofline"' in line:
            return json.loads(line)
    raise SystemExit(f"no bench JSON line in {log}")


def per_product(sub, counter, products):
    """Counter value per SpMV: the sum over ALL dispatches of a kernel name divided by the number of products the
    profiled bench ran (a kernel that is launched twice per product -- csr_tile: ordinary tiles + long rows' tiles --
    must be added up, not averaged)."""
    acc = defaultdict(float)
    files = sorted(glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # the newest pass only (a merged gpurun_out/ keeps earlier runs' files beside it)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]] += float(row["Counter_Value"])
    return {k: v / products for k, v in acc.items()}


line = bench_line(os.path.join(root, "trace.log"))
kernel = line["roofline"]["kernel"]
workload = line["config"]["workload_key"]
products = int(line["steps"]) + int(line["warmup"]) + 1   # bench.py at N = 1: W warm-ups + 1, then the K timed steps
fetch = per_product("pmc_fetch", "FETCH_SIZE", products)
write = per_product("pmc_write", "WRITE_SIZE", products)
out_path = os.path.join(ROOT, "profiles", "traffic.json")
table = json.load(open(out_path)) if os.path.exists(out_path) else {}
table = {k: v for k, v in table.items() if isinstance(v, dict)}  # round-1 entries carried no stamp
# a product of csr_tile is several kernels: the tiers' csr_tile launches, tile_expand ahead of an expanded plan, the slab
# sums and the remainder behind them -- all of them move the product's bytes
COMPANIONS = {"csr_tile": ("tile_expand", "tile_slab_finish", "tile_remainder", "csr_long_pieces", "csr_long_finish")}
names = (kernel.split("<")[0],) + COMPANIONS.get(kernel.split("<")[0], ())
hits = [k for k in fetch if k.split("(")[0].split("<")[0].strip().endswith(names)]
if not hits:
    raise SystemExit(f"kernel {kernel} not found among {sorted(fetch)}")
# a kernel may run as several instantiations per product (csr_tile: ordinary tiles without the packed decode + the
# long rows' tiles with it): their traffic adds up
f = sum(fetch[k] for k in hits)
w = sum(write.get(k, 0.0) for k in hits)
table[f"{kernel}|{workload}"] = {
    "bytes": int((2 * f + w) * 1024), "fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1),
    "source": tag, "kernel_src_sha": kernel_source_sha(kernel),
    "format_bytes": int(line["roofline"]["format_bytes_per_launch"]),
    "blocks": int(line["config"]["workgroups"]), "instantiations": len(hits), "kernels": sorted({k.split("(")[0].split("<")[0].split("::")[-1].strip() for k in hits}),
    "kernel_ms_mean_unprofiled": line["roofline"]["kernel_ms_mean"]}
print(f"{kernel}|{workload}: FETCH_SIZE={f:.0f} KiB WRITE_SIZE={w:.0f} KiB -> {(2 * f + w) * 1024 / 1e6:.1f} MB ({len(hits)} instantiation(s))")
json.dump(table, open(out_path, "w"), indent=1, sort_keys=True)
