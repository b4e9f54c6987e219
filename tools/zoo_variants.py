#!/usr/bin/env python3
"""All CSR / HLL variants on the two matrix classes of tools/matrix_zoo.py that get no x-window plan
and are not uniformly random (road-like, wide random band): is AUTO's choice the best one there?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
import scipy.sparse as sps  # noqa: E402

rng = np.random.default_rng(2026)


def banded_random(n, per_row, sigma):
    r = np.repeat(np.arange(n, dtype=np.int64), per_row)
    c = np.clip(r + np.rint(rng.normal(0, sigma, len(r))).astype(np.int64), 0, n - 1)
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    return a


sp.hip_init(0)
for name, a in (("road-like: 3 neighbours, sigma 2000", banded_random(12_000_000, 3, 2000.0)),
                ("road-like: 3 neighbours, sigma 200", banded_random(12_000_000, 3, 200.0)),
                ("wide random band: 30 per row, sigma 20000", banded_random(2_000_000, 30, 20000.0)),
                ("random band: 30 per row, sigma 2000", banded_random(2_000_000, 30, 2000.0))):
    M, N = a.shape
    rp, col, val = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    with sp.CsrDevice(M, N, rp, col, val) as dev:
        dev.set_x(np.ones(N))
        info = dev.info()
        print(f"{name}: M={M} nnz={info['nz']} plan blocks={info['local_blocks']} lanes_per_row={info['lanes_per_row']}")
        for vname, v in sp.CSR_VARIANTS.items():
            ms = dev.time(v, 2, 10, zero_y=False)
            print(f"   csr {vname:12s} {ms.mean() * 1e3:9.1f} us  {info['algo_bytes'] / ms.mean() / 1e6 / 80:5.1f} % of 8 TB/s")
        with sp.HllDevice.from_csr_device(dev) as h:
            h.set_x(np.ones(N))
            hi = h.info()
            for vname, v in sp.HLL_VARIANTS.items():
                ms = h.time(v, 2, 10, zero_y=False)
                print(f"   hll {vname:12s} {ms.mean() * 1e3:9.1f} us  {hi['algo_bytes'] / ms.mean() / 1e6 / 80:5.1f} % (plan {hi['local_blocks']})")
