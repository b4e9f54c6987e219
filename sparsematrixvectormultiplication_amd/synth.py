"""Seeded stand-in matrices (bench / test workload tooling, include/synth_matrix.h).

SuiteSparse `cant` and `nlpkkt120` are not available offline; these generate
shape-matched symmetric stand-ins directly as CSR, any row range at a time.
"""
from __future__ import annotations

import numpy as np

from . import _native as nat

KKT_GRID = (120, 120, 123)  # M = 3 542 400, like nlpkkt120
FEM_GRID = (9, 9, 257)      # M = 62 451, like cant


def _ip(a):
    return a.ctypes.data_as(nat.c_int_p)


def _gen(kind, dims, seed, row0, row1, dtype=np.float64):
    L = nat.lib()
    M = getattr(L, f"synth_{kind}_rows")(*dims)
    if M < 0:
        raise ValueError(f"bad {kind} grid {dims}")
    row_ptr = np.zeros(M + 1, dtype=np.int32)
    if getattr(L, f"synth_{kind}_row_ptr")(*dims, _ip(row_ptr)) != 0:
        raise ValueError(f"synth_{kind}_row_ptr failed")
    row1 = M if row1 is None else row1
    n = int(row_ptr[row1] - row_ptr[row0])
    col = np.empty(n, dtype=np.int32)
    val = np.empty(n, dtype=dtype)
    if getattr(L, f"synth_{kind}_fill")(*dims, seed, row0, row1, _ip(row_ptr), _ip(col),
                                        val.ctypes.data_as(nat.c_double_p)) != 0:
        raise ValueError(f"synth_{kind}_fill failed")
    return M, row_ptr, col, val


def kkt_like(grid=KKT_GRID, seed=2, row0=0, row1=None):
    """(M, row_ptr[M+1] of the WHOLE matrix, col, val of rows [row0, row1))."""
    return _gen("kkt", tuple(grid), seed, row0, row1)


def fem_like(grid=FEM_GRID, seed=1, row0=0, row1=None):
    return _gen("fem", tuple(grid), seed, row0, row1)


def powerlaw(n=1 << 24, max_degree=1 << 20, seed=5, row0=0, row1=None):
    """fp32 power-law CSR; same return convention as kkt_like."""
    L = nat.lib()
    row_ptr = np.zeros(n + 1, dtype=np.int32)
    if L.synth_powerlaw_row_ptr(n, max_degree, seed, _ip(row_ptr)) != 0:
        raise ValueError("synth_powerlaw_row_ptr failed")
    row1 = n if row1 is None else row1
    cnt = int(row_ptr[row1] - row_ptr[row0])
    col = np.empty(cnt, dtype=np.int32)
    val = np.empty(cnt, dtype=np.float32)
    if L.synth_powerlaw_fill(n, max_degree, seed, row0, row1, _ip(row_ptr), _ip(col),
                             val.ctypes.data_as(nat.c_float_p)) != 0:
        raise ValueError("synth_powerlaw_fill failed")
    return n, row_ptr, col, val
