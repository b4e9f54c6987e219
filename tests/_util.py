"""Shared helpers for the parity tests."""
import numpy as np

# BASELINE.json north_star: "within 1e-10 relative error (fp64)" against the
# reference's serial CSR result.  A re-ordered summation (lane-strided partial
# sums + butterfly, and FMA contraction on the GPU) cannot reproduce the serial
# left-to-right sum bit for bit, and for rows whose exact sum cancels to ~0
# (Laplacians / KKT blocks with x = 1) an element-wise relative error is
# unbounded, so the gate is ("Tolerance vs summation order"):
#   norm-wise:  max|y - y_ref| <= 1e-10 * max|y_ref|
#   row-wise:   |y_i - y_ref_i| <= 1e-10 * sum_j |a_ij x_j|
FP64_RTOL = 1e-10
FP32_NORMWISE_RTOL = 1e-5  # config 5 (fp32 data) against the fp64-accumulated oracle


def row_abs_sums(row_ptr, col_idx, values, x):
    p = np.abs(values * x[col_idx])
    out = np.zeros(len(row_ptr) - 1)
    nonempty = np.flatnonzero(np.diff(row_ptr) > 0)
    if len(nonempty):
        out[nonempty] = np.add.reduceat(p, row_ptr[nonempty])
    return out


def assert_parity(y, y_ref, row_ptr, col_idx, values, x, rtol=FP64_RTOL, what=""):
    y = np.asarray(y, dtype=np.float64)
    y_ref = np.asarray(y_ref, dtype=np.float64)
    assert y.shape == y_ref.shape, f"{what}: shape {y.shape} vs {y_ref.shape}"
    if y.size == 0:
        return
    assert np.all(np.isfinite(y)), f"{what}: non-finite result"
    d = np.abs(y - y_ref)
    scale = np.max(np.abs(y_ref))
    assert d.max() <= rtol * scale + 0.0 or scale == 0 and d.max() == 0, \
        f"{what}: norm-wise {d.max() / max(scale, 1e-300):.3e} > {rtol}"
    bound = rtol * row_abs_sums(np.asarray(row_ptr), np.asarray(col_idx),
                                np.asarray(values, dtype=np.float64),
                                np.asarray(x, dtype=np.float64))
    bad = np.flatnonzero(d > bound)
    assert bad.size == 0, (f"{what}: {bad.size} rows beyond {rtol} * sum|a_ij x_j|; first row "
                           f"{bad[0]}: |d| = {d[bad[0]]:.3e}, bound = {bound[bad[0]]:.3e}")


def random_csr(rng, M, N, mean_row, max_row=None, empty_frac=0.0, dtype=np.float64):
    """Random CSR with sorted, distinct columns per row."""
    max_row = min(N, max_row or max(1, 4 * mean_row))
    lens = np.minimum(rng.poisson(mean_row, M), max_row).astype(np.int64)
    lens[rng.random(M) < empty_frac] = 0
    row_ptr = np.zeros(M + 1, dtype=np.int32)
    np.cumsum(lens, out=row_ptr[1:])
    col = np.empty(row_ptr[-1], dtype=np.int32)
    for r in range(M):
        if lens[r]:
            col[row_ptr[r]:row_ptr[r + 1]] = np.sort(rng.choice(N, lens[r], replace=False))
    val = rng.uniform(-1, 1, row_ptr[-1]).astype(dtype)
    return row_ptr, col, val


def coo_from_csr(row_ptr, col, val, rng=None):
    rows = np.repeat(np.arange(len(row_ptr) - 1, dtype=np.int32), np.diff(row_ptr))
    if rng is not None:
        order = rng.permutation(len(rows))
        return rows[order], col[order], val[order]
    return rows, col, val


def banded_csr(rng, M, N, mean_row, band, empty_frac=0.0, dtype=np.float64, far_frac=0.0):
    """Random CSR whose columns stay within `band` of the diagonal (plus, for far_frac of the
    rows, one cluster far away), sorted and distinct per row: few x lines per row block."""
    lens = np.minimum(rng.poisson(mean_row, M), min(N, 2 * band)).astype(np.int64)
    lens[rng.random(M) < empty_frac] = 0
    row_ptr = np.zeros(M + 1, dtype=np.int32)
    np.cumsum(lens, out=row_ptr[1:])
    col = np.empty(row_ptr[-1], dtype=np.int32)
    for r in range(M):
        n = lens[r]
        if not n:
            continue
        centre = int(r * (N - 1) / max(M - 1, 1))
        lo = max(0, min(centre - band, N - 2 * band))
        cand = np.arange(lo, min(N, lo + 2 * band))
        c = rng.choice(cand, n, replace=False)
        if far_frac and rng.random() < far_frac:
            k = max(1, n // 4)
            c[:k] = (c[:k] + N // 2) % N
            c = np.unique(c)
            c = np.concatenate([c, rng.choice(np.setdiff1d(cand, c), n - len(c), replace=False)]) if len(c) < n else c
        col[row_ptr[r]:row_ptr[r + 1]] = np.sort(c)
    val = rng.uniform(-1, 1, row_ptr[-1]).astype(dtype)
    return row_ptr, col, val
