#!/usr/bin/env python3
"""Experiment: does visiting the rows of the nlpkkt-like matrix in 3-D bricks (rows that share
their x neighbourhood) instead of natural order speed up csr_stream?  Emulated by permuting the
ROWS of the CSR matrix on the host (columns / x untouched; y comes out permuted)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402
from sparsematrixvectormultiplication_amd.device import set_tuning  # noqa: E402

nx, ny, nz = synth.KKT_GRID
sp.hip_init(0)
M, row_ptr, col, val = synth.kkt_like()
n1 = M // 2
lens = np.diff(row_ptr)


def permuted(A, B, C):
    p = np.arange(M) % n1
    half = np.arange(M) // n1
    i, j, k = p % nx, (p // nx) % ny, p // (nx * ny)
    key = (half, k // C, j // B, i // A, k % C, j % B, i % A)
    order = np.lexsort(key[::-1])
    new_len = lens[order]
    new_rp = np.zeros(M + 1, np.int32)
    np.cumsum(new_len, out=new_rp[1:])
    # entry index array: for new row q, old entries row_ptr[order[q]] .. +len
    starts = np.repeat(row_ptr[order].astype(np.int64) - new_rp[:-1], new_len)
    idx = starts + np.arange(new_rp[-1], dtype=np.int64)
    return new_rp, col[idx], val[idx], order


x = np.ones(M)
set_tuning("stream_cap", 4096)
results = []
for name, brick in (("natural", None), ("16x4x4", (16, 4, 4)), ("8x8x4", (8, 8, 4)), ("32x2x2", (32, 2, 2)), ("120x1x1", (120, 1, 1)), ("8x8x8", (8, 8, 8))):
    if brick is None:
        rp, c, v = row_ptr, col, val
    else:
        rp, c, v, order = permuted(*brick)
    with sp.CsrDevice(M, M, rp, c, v) as dev:
        dev.set_x(x)
        best = []
        for kind, nm in ((0, "prod"), (1, "walk")):
            set_tuning("stream_kind", kind)
            ms = np.concatenate([dev.time(sp.CSR_STREAM, 2, 20, zero_y=False) for _ in range(3)])
            best.append(f"{nm} {ms.mean() * 1e3:6.1f} us ({dev.info()['algo_bytes'] / ms.mean() / 1e6 / 80:.1f} %)")
        print(f"rows in {name:8s} order: " + "   ".join(best), flush=True)
