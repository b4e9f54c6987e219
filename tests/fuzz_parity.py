#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: many small matrices of random shape (banded / scattered /
mixed, empty rows, long rows, unsorted and repeated columns, fp64 and fp32, row blocks), the fast
path of both formats against the oracle.  Usage: python tests/fuzz_parity.py [cases] [seed]
(lives under tests/ because it checks against the oracle; run by test_randomised_parity_sweep)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from _util import assert_parity  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
oracle = Oracle()
sp.hip_init(0)
stats = {"plan": 0, "no_plan": 0, "hll_plan": 0, "long": 0}
for case in range(cases):
    M = int(rng.choice([1, 7, 31, 32, 33, 500, 3000, 20000]))
    N = int(rng.choice([1, 5, 17, 100, 1000, 3001, 30000, 200000]))
    shape = rng.choice(["banded", "scattered", "mixed"])
    mean = float(rng.choice([0.3, 2, 9, 30, 90, 400]))
    lens = rng.poisson(mean, M).astype(np.int64)
    lens[rng.random(M) < rng.choice([0.0, 0.1, 0.6])] = 0
    if M > 3 and rng.random() < 0.3:
        lens[rng.integers(0, M, 2)] = rng.integers(1500, 9000)       # rows longer than the stage
    lens = np.minimum(lens, N if shape != "banded" else max(1, min(N, 2 * 64 + 1)))
    dup = rng.random() < 0.15                                        # repeated columns allowed
    unsorted = rng.random() < 0.25
    cols = []
    for r in range(M):
        n = int(lens[r])
        if n == 0:
            cols.append(np.zeros(0, np.int32))
            continue
        centre = int(r * (N - 1) / max(M - 1, 1))
        if shape == "banded" or (shape == "mixed" and rng.random() < 0.9):
            lo = max(0, min(centre - 64, N - 129))
            pool = np.arange(lo, min(N, lo + 129))
        else:
            pool = None
        if pool is not None:
            c = rng.choice(pool, n, replace=dup or n > len(pool))
        else:
            c = rng.integers(0, N, n) if dup else rng.choice(N, n, replace=n > N)
        c = c.astype(np.int32)
        cols.append(c if unsorted else np.sort(c))
    lens = np.array([len(c) for c in cols], dtype=np.int64)
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate(cols).astype(np.int32) if M else np.zeros(0, np.int32)
    dtype = np.float32 if rng.random() < 0.3 else np.float64
    val = rng.uniform(-1, 1, len(col)).astype(dtype)
    x = rng.uniform(-1, 1, N).astype(dtype)
    what = f"case {case}: {shape} M={M} N={N} mean={mean} nnz={len(col)} {np.dtype(dtype).name} dup={dup} unsorted={unsorted}"
    if dtype == np.float64:
        y_ref = oracle.csr_serial(row_ptr, col, val, x)
    else:
        y_ref = oracle.csr_f32_accum64(row_ptr, col, val, x)
    r0 = int(rng.integers(0, M)) if rng.random() < 0.3 else 0
    r1 = int(rng.integers(r0, M + 1)) if r0 else M
    with sp.CsrDevice(M, N, row_ptr, col, val, row0=r0, row1=r1) as dev:
        info = dev.info()
        stats["plan" if info["local_blocks"] else "no_plan"] += 1
        stats["long"] += info["long_rows"] > 0
        y = dev.spmv(x, sp.CSR_AUTO)
        if r1 == r0:
            pass
        elif dtype == np.float64:
            assert_parity(y[r0:r1], y_ref[r0:r1], row_ptr[r0:r1 + 1] - row_ptr[r0],
                          col[row_ptr[r0]:row_ptr[r1]], val[row_ptr[r0]:row_ptr[r1]], x, what=what)
        else:
            scale = max(float(np.max(np.abs(y_ref))), 1e-30)
            assert np.max(np.abs(y[r0:r1].astype(np.float64) - y_ref[r0:r1])) <= 1e-5 * scale, what
        if dtype == np.float64 and r0 == 0 and r1 == M:
            with sp.HllDevice.from_csr_device(dev) as h:
                stats["hll_plan"] += h.info()["local_blocks"] > 0
                assert_parity(h.spmv(x, sp.HLL_AUTO), y_ref, row_ptr, col, val, x, what="HLL " + what)
            keys = np.repeat(np.arange(M, dtype=np.int64), np.diff(row_ptr)) * N + col
            if not unsorted and len(np.unique(keys)) == len(keys):   # no (row, column) pair repeats
                # the same matrix built on the device from shuffled triplets: identical CSR arrays
                rows_of = np.repeat(np.arange(M, dtype=np.int32), np.diff(row_ptr))
                perm = rng.permutation(len(col))
                with sp.CsrDevice.from_coo(M, N, rows_of[perm], col[perm], val[perm]) as built:
                    rp_b, col_b, val_b = built.download()
                    assert np.array_equal(rp_b, row_ptr) and np.array_equal(col_b, col) and \
                        val_b.tobytes() == val.tobytes(), "from_coo " + what
                    assert built.spmv(x, sp.CSR_AUTO).tobytes() == y.tobytes(), "from_coo y " + what
                stats["coo"] = stats.get("coo", 0) + 1
    if case % 25 == 24:
        print(f"{case + 1} cases ok  {stats}", flush=True)
print(f"all {cases} cases passed (seed {seed}): {stats}")
