"""Join 'STATE k us' lines of tools/placement_pmc.py with the per-dispatch counters of the rocprofv3 --pmc pass that ran it.
Usage: python tools/placement_pmc_report.py <pass dir> <log with the STATE lines>"""
import csv
import glob
import sys
from collections import defaultdict

pass_dir, log = sys.argv[1], sys.argv[2]
states = [(int(l.split()[1]), float(l.split()[2])) for l in open(log) if l.startswith("STATE")]
files = glob.glob(pass_dir + "/**/*counter_collection.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "csr_stream_local" in r.get("Kernel_Name", "")]
by_dispatch = defaultdict(dict)
for r in rows:
    by_dispatch[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by_dispatch)
per = 15
names = sorted({n for d in by_dispatch.values() for n in d})
print(f"{'state':>5s} {'us':>7s} " + " ".join(f"{n[-26:]:>26s}" for n in names))
for k, (state, us) in enumerate(states):
    mine = ids[k * per + 5:(k + 1) * per]
    if not mine:
        break
    print(f"{state:5d} {us:7.1f} " + " ".join(f"{sum(by_dispatch[i].get(n, 0) for i in mine) / len(mine):26.0f}" for n in names))
