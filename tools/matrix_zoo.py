#!/usr/bin/env python3
"""Breadth check of the fast path beyond the two BASELINE shapes: structural classes that occur in
the reference's own result list (stencils, FEM blocks, road-like graphs, wide bands, uniformly
random columns), each at >= 10 M nonzeros (well above the Infinity Cache), built with scipy on the
host.  For every class: which kernel upload picked, kernel time, algorithmic GB/s, % of 8 TB/s for
CSR and HLL, and parity against scipy's CSR product (row-wise 1e-10 gate)."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from _util import assert_parity  # noqa: E402

rng = np.random.default_rng(2026)


def stencil(dims, offsets):
    n = int(np.prod(dims))
    idx = np.arange(n, dtype=np.int64)
    coords = np.unravel_index(idx, dims)
    rows, cols = [], []
    for off in offsets:
        ok = np.ones(n, bool)
        lin = idx.copy()
        stride = 1
        for d in range(len(dims) - 1, -1, -1):
            c = coords[d] + off[d]
            ok &= (c >= 0) & (c < dims[d])
            lin += off[d] * stride
            stride *= dims[d]
        rows.append(idx[ok])
        cols.append(lin[ok])
    r, c = np.concatenate(rows), np.concatenate(cols)
    return sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))


def banded_random(n, per_row, sigma):
    r = np.repeat(np.arange(n, dtype=np.int64), per_row)
    c = np.clip(r + np.rint(rng.normal(0, sigma, len(r))).astype(np.int64), 0, n - 1)
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))
    a.sum_duplicates()
    return a


def uniform_random(n, per_row):
    r = np.repeat(np.arange(n, dtype=np.int64), per_row)
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, rng.integers(0, n, len(r)))), shape=(n, n))
    a.sum_duplicates()
    return a


off5 = [(0, 0), (0, 1), (0, -1), (1, 0), (-1, 0)]
off7 = [(0, 0, 0)] + [tuple(s * (1 if k == d else 0) for k in range(3)) for d in range(3) for s in (1, -1)]
off27 = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]
zoo = [
    ("2-D 5-point stencil 4000 x 4000", lambda: stencil((4000, 4000), off5)),
    ("3-D 7-point stencil 256^3", lambda: stencil((256, 256, 256), off7)),
    ("3-D 27-point stencil 160^3", lambda: stencil((160, 160, 160), off27)),
    ("road-like: 3 neighbours, sigma 2000", lambda: banded_random(12_000_000, 3, 2000.0)),
    ("wide random band: 30 per row, sigma 20000", lambda: banded_random(2_000_000, 30, 20000.0)),
    ("uniformly random columns: 20 per row, n = 4 M", lambda: uniform_random(4_000_000, 20)),
]
sp.hip_init(0)
print("| class | rows | nnz | CSR kernel | us | GB/s (algorithmic) | % of 8 TB/s | HLL kernel | us | % of 8 TB/s (HLL bytes) | parity |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for name, make in zoo:
    t = time.perf_counter()
    a = make()
    a.sort_indices()
    M, N = a.shape
    row_ptr, col, val = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    x = rng.uniform(-1, 1, N)
    y_ref = a @ x
    with sp.CsrDevice(M, N, row_ptr, col, val) as dev:
        info = dev.info()
        y = dev.spmv(x, sp.CSR_AUTO)
        lo, hi = M // 3, M // 3 + 20000
        assert_parity(y[lo:hi], y_ref[lo:hi], row_ptr[lo:hi + 1] - row_ptr[lo], col[row_ptr[lo]:row_ptr[hi]],
                      val[row_ptr[lo]:row_ptr[hi]], x, what=name)
        assert np.max(np.abs(y - y_ref)) <= 1e-10 * np.max(np.abs(y_ref)), name
        ms = dev.time(sp.CSR_AUTO, 3, 20, zero_y=False)
        with sp.HllDevice.from_csr_device(dev) as h:
            hi_info = h.info()
            yh = h.spmv(x, sp.HLL_AUTO)
            assert np.max(np.abs(yh - y_ref)) <= 1e-10 * np.max(np.abs(y_ref)), "HLL " + name
            hms = h.time(sp.HLL_AUTO, 3, 20, zero_y=False)
    gb = info["algo_bytes"] / (ms.mean() * 1e-3) / 1e9
    hgb = hi_info["algo_bytes"] / (hms.mean() * 1e-3) / 1e9
    print(f"| {name} | {M} | {info['nz']} | {sp.device.CSR_STREAM_KERNELS[info['stream_kernel']]} | "
          f"{ms.mean() * 1e3:.1f} | {gb:.0f} | {gb / 80:.1f} | {sp.device.HLL_LDS_KERNELS[hi_info['stream_kernel']]} | "
          f"{hms.mean() * 1e3:.1f} | {hgb / 80:.1f} | ok |", flush=True)
