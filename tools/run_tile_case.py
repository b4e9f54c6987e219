#!/usr/bin/env python3
"""One matrix class of tools/time_tile.py through csr_tile, a few launches (to be run under rocprofv3 --pmc)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TILE_QUIET", "1")
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
import scipy.sparse as sps  # noqa: E402

rng = np.random.default_rng(2026)
case = sys.argv[1] if len(sys.argv) > 1 else "road"
n, per_row, sigma = (12_000_000, 3, 2000.0) if case == "road" else (2_000_000, 30, 20000.0)
r = np.repeat(np.arange(n, dtype=np.int64), per_row)
c = np.clip(r + np.rint(rng.normal(0, sigma, len(r))).astype(np.int64), 0, n - 1)
a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))
a.sum_duplicates()
a.sort_indices()
sp.hip_init(0)
with sp.CsrDevice(n, n, a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data) as dev:
    dev.set_x(np.ones(n))
    info = dev.info()
    ms = dev.time(sp.CSR_AUTO, 2, 8, zero_y=False)
    print(f"{case}: {sp.device.CSR_STREAM_KERNELS[info['stream_kernel']]} {ms.mean() * 1e3:.1f} us blocks={info['tile_blocks']} passes={info['tile_passes']}")
