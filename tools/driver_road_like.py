#!/usr/bin/env python3
"""The C driver on a matrix that gets the csr_tile plan: a road-like band of roadNet-PA's size (1.09 M rows, 3 per row,
general storage) written as a Matrix Market file, parsed, converted, uploaded and run by spmv_bench (HLL built on the
device); prints the driver's result lines and its roofline CSV (kernel names included)."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(7)
n, per_row, sigma = 1_090_920, 3, 3000.0
r = np.repeat(np.arange(n, dtype=np.int64), per_row)
c = np.clip(r + np.rint(rng.normal(0, sigma, len(r))).astype(np.int64), 0, n - 1)
key = np.unique(r * n + c)
r, c = key // n, key % n
v = rng.uniform(-1, 1, len(r))
work = tempfile.mkdtemp(prefix="spmv_road_", dir=os.environ.get("TMPDIR", "/tmp"))
path = os.path.join(work, "road_like.mtx")
t = time.perf_counter()
with open(path, "w") as f:
    f.write("%%MatrixMarket matrix coordinate real general\n")
    f.write(f"{n} {n} {len(r)}\n")
    np.savetxt(f, np.column_stack([r + 1, c + 1, v]), fmt="%d %d %.17g")
print(f"wrote {path}: {os.path.getsize(path) / 1e6:.0f} MB, {len(r)} entries in {time.perf_counter() - t:.1f} s", flush=True)
driver = os.path.join(ROOT, "sparsematrixvectormultiplication_amd", "spmv_bench")
out = os.path.join(work, "result")
t = time.perf_counter()
p = subprocess.run([driver, "--out", out, "--iters", "95", "--hll-on-device", path], capture_output=True, text=True)
print(f"exit {p.returncode}, {time.perf_counter() - t:.2f} s wall")
print("\n".join(ln for ln in p.stdout.splitlines() if "us |" in ln))
if p.returncode:
    print(p.stdout[-1500:], p.stderr[-1500:])
print(open(os.path.join(out, "spmv_results_hip_roofline.csv")).read())
