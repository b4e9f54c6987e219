#!/usr/bin/env python3
"""End to end through the C driver at a realistic size: write the cant-like stand-in as a Matrix
Market file (symmetric storage, like the SuiteSparse original), then run spmv_bench on it three
ways: host HLL builder; HLL built on the device; device HLL + the .csrbin sidecar (second run
reads the sidecar instead of parsing).  Prints wall times of each run and the driver's own lines."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsematrixvectormultiplication_amd import synth  # noqa: E402

grid = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else synth.FEM_GRID
M, row_ptr, col, val = synth.fem_like(grid, 1)
rows = np.repeat(np.arange(M, dtype=np.int64), np.diff(row_ptr))
keep = col <= rows                      # lower triangle: the generator's matrix is symmetric
work = tempfile.mkdtemp(prefix="spmv_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
path = os.path.join(work, "cant_like.mtx")
t = time.perf_counter()
with open(path, "w") as f:
    f.write("%%MatrixMarket matrix coordinate real symmetric\n")
    f.write(f"{M} {M} {int(keep.sum())}\n")
    np.savetxt(f, np.column_stack([rows[keep] + 1, col[keep] + 1, val[keep]]), fmt="%d %d %.17g")
print(f"wrote {path}: {os.path.getsize(path) / 1e6:.0f} MB, {int(keep.sum())} stored entries "
      f"({int(row_ptr[-1])} after symmetric expansion) in {time.perf_counter() - t:.1f} s", flush=True)
driver = os.path.join(ROOT, "sparsematrixvectormultiplication_amd", "spmv_bench")
out = os.path.join(work, "result")
for label, flags in (("host HLL builder", []), ("HLL on device", ["--hll-on-device"]),
                     ("HLL on device + sidecar (writes it)", ["--hll-on-device", "--cache"]),
                     ("HLL on device + sidecar (reads it)", ["--hll-on-device", "--cache"])):
    t = time.perf_counter()
    p = subprocess.run([driver, "--out", out, "--iters", "95", *flags, path], capture_output=True, text=True)
    dt = time.perf_counter() - t
    lines = [ln for ln in p.stdout.splitlines() if "us |" in ln or "sidecar" in ln]
    print(f"--- {label}: exit {p.returncode}, {dt:.2f} s wall\n" + "\n".join(lines), flush=True)
    if p.returncode:
        print(p.stdout[-1500:], p.stderr[-1500:])
print(open(os.path.join(out, "spmv_results_hip_roofline.csv")).read())
