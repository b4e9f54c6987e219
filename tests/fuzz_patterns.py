#!/usr/bin/env python3
"""Randomised check of the pattern plan (csr_stream_local<.., PAT>): matrices with an x-window plan of random shape --
stencil-like rows (shifted copies), random bands, long rows, empty rows, row blocks, fp64 / fp32 -- with the plan
FORCED: the result must equal, bit for bit, what the same handle gives through the slot stream, and the oracle within
the gate.  usage: fuzz_patterns.py [cases] [seed]   (also run by tests/test_gpu_parity.py with few cases)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd.device import set_tuning  # noqa: E402


def make_case(rng, case):
    M = int(rng.integers(50, 40000))
    N = M if case % 3 else int(rng.integers(M, 3 * M))
    kind = case % 5
    if kind in (0, 1, 2):  # stencil-like: a fixed set of offsets, rows shifted copies; some rows perturbed
        k = int(rng.choice([3, 7, 15, 28, 60, 130, 300]))
        span = int(rng.choice([k + 2, 4 * k, 40 * k]))
        offs = np.unique(rng.integers(-span, span + 1, k))
        rows = np.repeat(np.arange(M), len(offs))
        cols = rows * (N - 1) // max(M - 1, 1) + np.tile(offs, M)
        keep = (cols >= 0) & (cols < N)
        if kind == 1:
            keep &= rng.random(len(cols)) > 0.02   # a few entries missing: rows that break the chain
        if kind == 2:
            keep &= (rows % 11 != 5)               # empty rows
        rows, cols = rows[keep], cols[keep]
    elif kind == 3:  # random band
        per = int(rng.choice([2, 9, 40]))
        rows = np.repeat(np.arange(M), per)
        cols = np.clip(rows * (N - 1) // max(M - 1, 1) + rng.integers(-60, 61, len(rows)), 0, N - 1)
    else:  # mostly short rows and a few long ones (longer than the stage: split-row kernels beside the blocks)
        lens = rng.poisson(6, M)
        lens[rng.integers(0, M, 3)] = rng.integers(1500, 5000, 3)
        rows = np.repeat(np.arange(M), lens)
        cols = np.clip(rows * (N - 1) // max(M - 1, 1) + rng.integers(-300, 301, len(rows)), 0, N - 1)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    uniq = np.ones(len(rows), bool)
    uniq[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
    rows, cols = rows[uniq], cols[uniq]
    rp = np.zeros(M + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    return M, N, rp, cols.astype(np.int32)


def run(cases=60, seed=2027, oracle=None):
    rng = np.random.default_rng(seed)
    with_plan = 0
    hll_with_plan = [0]
    for case in range(cases):
        dtype = np.float32 if case % 4 == 3 else np.float64
        M, N, rp, col = make_case(rng, case)
        val = rng.uniform(-1, 1, len(col)).astype(dtype)
        x = rng.uniform(-1, 1, N).astype(dtype)
        r0, r1 = (0, M) if case % 6 else (M // 4, 3 * M // 4)
        set_tuning("local_patterns", 1)
        try:
            with sp.CsrDevice(M, N, rp, col, val, row0=r0, row1=r1) as dev:
                info = dev.info()
                if not info["local_blocks"]:
                    continue
                with_plan += info["pattern_slots"] > 0
                sp.lib().spmv_hip_memset(dev.y_ptr, 0xFF, M * x.itemsize)
                y1 = dev.spmv(x, sp.CSR_STREAM)[r0:r1].copy()
                set_tuning("local_patterns", 0)
                y0 = dev.spmv(x, sp.CSR_STREAM)[r0:r1].copy()
                assert y1.tobytes() == y0.tobytes(), f"case {case}: M={M} N={N} nnz={len(col)} {np.dtype(dtype).name} rows [{r0}, {r1})"
                if oracle is not None:
                    ref = (oracle.csr_serial if dtype == np.float64 else oracle.csr_f32_accum64)(rp, col, val, x)[r0:r1]
                    tol = 1e-10 if dtype == np.float64 else 1e-5
                    assert np.max(np.abs(y1.astype(np.float64) - ref)) <= tol * max(np.max(np.abs(ref)), 1e-300), f"case {case}"
                # the HLL twin on the slab built from this handle (fp64, whole matrices)
                if dtype == np.float64 and (r0, r1) == (0, M) and case % 2 == 0:
                    set_tuning("local_patterns", 1)
                    with sp.HllDevice.from_csr_device(dev) as hdev:
                        if hdev.info()["local_blocks"] and hdev.info()["pattern_slots"]:
                            h1 = hdev.spmv(x, sp.HLL_LDS).copy()
                            set_tuning("local_patterns", 0)
                            h0 = hdev.spmv(x, sp.HLL_LDS).copy()
                            assert h1.tobytes() == h0.tobytes(), f"HLL case {case}: M={M} N={N} nnz={len(col)}"
                            hll_with_plan[0] += 1
        finally:
            set_tuning("local_patterns", -1)
    run.hll_with_plan = hll_with_plan[0]
    return with_plan


if __name__ == "__main__":
    sp.hip_init(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2027
    print(f"{run(n, seed)} of {n} cases had a pattern plan ({run.hll_with_plan} HLL slabs of them as well); all equal to the slot stream bit for bit")
