#!/usr/bin/env python3
"""The reference's own matrix list (result/result_cuda.csv: 30 SuiteSparse matrices; none of the files is available
offline) as seeded STAND-INS of the same size and structural class: rows, entries per row and the kind of column
pattern are the matrix's, the entries are not.  For every one: which kernel AUTO resolves to, its time, GFLOP/s,
% of 8 TB/s by algorithmic bytes, the best of all CSR variants (is AUTO's choice the best one?), the same for HLL,
parity against scipy -- plus, for context only, the GFLOP/s the reference published for its best CUDA kernel on the
real matrix (other hardware, other data: not a comparison).
usage: reference_list.py [substring of a name ...]"""
import os
import sys

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402

rng = np.random.default_rng(30)


def band(n, nnz, sigma, sym=False):
    per = max(1, round(nnz / n))
    r = np.repeat(np.arange(n, dtype=np.int64), per)
    c = np.clip(r + np.rint(rng.normal(0, sigma, len(r))).astype(np.int64), 0, n - 1)
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))
    if sym:
        a = a + a.T
    a.sum_duplicates()
    a.sort_indices()
    return a.tocsr()


def scattered(n, nnz):
    per = max(1, round(nnz / n))
    r = np.repeat(np.arange(n, dtype=np.int64), per)
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, rng.integers(0, n, len(r)))), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    return a


def powerlaw(n, nnz, longest):
    # row lengths ~ Zipf clipped to [1, longest], scaled to the entry count; columns half preferential, half uniform
    lens = np.minimum((1.0 / rng.random(n)) ** 0.9, longest)
    lens = np.maximum(1, np.rint(lens * nnz / lens.sum())).astype(np.int64)
    r = np.repeat(np.arange(n, dtype=np.int64), lens)
    hot = np.minimum((n * rng.random(len(r)) ** 3).astype(np.int64), n - 1)
    c = np.where(rng.random(len(r)) < 0.5, hot, rng.integers(0, n, len(r)))
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    return a


def circuit(n, nnz):
    # a sparse band plus a handful of rows and columns that touch a large part of the matrix (supply nets)
    a = band(n, nnz * 0.7, 50).tolil()
    for k in rng.integers(0, n, 4):
        cols = rng.integers(0, n, int(nnz * 0.04))
        a[k, cols] = rng.uniform(-1, 1, len(cols))
        a[cols, k] = rng.uniform(-1, 1, len(cols))
    a = a.tocsr()
    a.sort_indices()
    return a


def kkt(n):
    g = round((n / 2.05) ** (1 / 3))
    M, rp, col, val = synth.kkt_like((g, g, g + 3))
    return sps.csr_matrix((val, col, rp), shape=(M, M))


# name, rows, nnz, generator, best published CUDA GFLOP/s of the reference on the real matrix (result_cuda.csv)
LIST = [
    ("cage4", 9, 49, lambda: band(9, 49, 2), None),
    ("mhda416", 416, 8562, lambda: band(416, 8562, 12), None),
    ("mcfe", 765, 24382, lambda: band(765, 24382, 40), None),
    ("olm1000", 1000, 3996, lambda: band(1000, 3996, 2), None),
    ("adder_dcop_32", 1813, 11246, lambda: circuit(1813, 11246), None),
    ("west2021", 2021, 7353, lambda: scattered(2021, 7353), None),
    ("cavity10", 2597, 76367, lambda: band(2597, 76367, 60), None),
    ("rdist2", 3198, 56934, lambda: band(3198, 56934, 100), None),
    ("raefsky2", 3242, 294276, lambda: band(3242, 294276, 120), None),
    ("mhd4800a", 4800, 102252, lambda: band(4800, 102252, 30), None),
    ("bcsstk17", 10974, 428650, lambda: band(10974, 428650, 150), None),
    ("olafu", 16146, 1015156, lambda: band(16146, 1015156, 250), None),
    ("FEM_3D_thermal1", 17880, 430740, lambda: band(17880, 430740, 600), None),
    ("af23560", 23560, 484256, lambda: band(23560, 484256, 300), None),
    ("cant", 62451, 4007383, lambda: sps.csr_matrix((lambda t: (t[3], t[2], t[1]))(synth.fem_like(synth.FEM_GRID, 1)), shape=(62451, 62451)), 44.7),
    ("thermal1", 82654, 574458, lambda: band(82654, 574458, 3000), None),
    ("thermomech_TK", 102158, 711558, lambda: band(102158, 711558, 2000), None),
    ("lung2", 109460, 492564, lambda: band(109460, 492564, 3), None),
    ("dc1", 116835, 766396, lambda: circuit(116835, 766396), None),
    ("cop20k_A", 121192, 2624331, lambda: scattered(121192, 2624331), None),
    ("PR02R", 161070, 8185136, lambda: band(161070, 8185136, 1500), None),
    ("mac_econ_fwd500", 206500, 1273389, lambda: band(206500, 1273389, 20000), None),
    ("amazon0302", 262111, 1234877, lambda: scattered(262111, 1234877), None),
    ("ML_Laplace", 377002, 27689972, lambda: band(377002, 27689972, 700), None),
    ("af_1_k101", 503625, 16739943, lambda: band(503625, 16739943, 900), None),
    ("webbase-1M", 1000005, 3105536, lambda: powerlaw(1000005, 3105536, 4700), None),
    ("nlpkkt80", 1062400, 28704672, lambda: kkt(1062400), None),
    ("roadNet-PA", 1090920, 3083796, lambda: band(1090920, 3083796, 3000), 31.5),
    ("thermal2", 1228045, 8580313, lambda: band(1228045, 8580313, 20000), None),
    ("Cube_Coup_dt0", 2164760, 127206144, lambda: band(2164760, 127206144, 2500), 47.1),
]

want = sys.argv[1:]
sp.hip_init(0)
print("| stand-in for | rows | nnz | AUTO kernel | us | GFLOP/s | % of 8 TB/s | best CSR variant | its us | HLL AUTO us | HLL GFLOP/s | "
      "reference's best CUDA GFLOP/s on the real matrix | parity |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, rows, nnz, make, pub in LIST:
    if want and not any(w in name for w in want):
        continue
    a = make().tocsr()
    a.sort_indices()
    M, N = a.shape
    rp, col, val = a.indptr.astype(np.int32), a.indices.astype(np.int32), np.ascontiguousarray(a.data, dtype=np.float64)
    x = rng.uniform(-1, 1, N)
    y_ref = a @ x
    scale = max(np.max(np.abs(y_ref)), 1e-300)
    with sp.CsrDevice(M, N, rp, col, val) as dev:
        info = dev.info()
        y = dev.spmv(x, sp.CSR_AUTO)
        assert np.max(np.abs(y - y_ref)) <= 1e-10 * scale, name
        dev.set_x(x)
        iters = 200 if info["nz"] < 5_000_000 else 30
        t_auto = dev.time(sp.CSR_AUTO, 5, iters, zero_y=False).mean()
        best = ("auto", t_auto)
        for vname, v in sp.CSR_VARIANTS.items():
            if vname == "wave_row" and M > 4_000_000:
                continue
            yv = dev.spmv(x, v)
            assert np.max(np.abs(yv - y_ref)) <= 1e-10 * scale, (name, vname)
            dev.set_x(x)
            t = dev.time(v, 3, max(10, iters // 4), zero_y=False).mean()
            if t < best[1] * 0.97:
                best = (vname, t)
        with sp.HllDevice.from_csr_device(dev) as h:
            yh = h.spmv(x, sp.HLL_AUTO)
            assert np.max(np.abs(yh - y_ref)) <= 1e-10 * scale, "HLL " + name
            h.set_x(x)
            t_hll = h.time(sp.HLL_AUTO, 5, iters, zero_y=False).mean()
    kern = sp.device.CSR_STREAM_KERNELS[info["stream_kernel"]] if info["auto_variant"] == sp.CSR_STREAM else \
        [k for k, v in sp.CSR_VARIANTS.items() if v == info["auto_variant"]][0]
    gf = 2.0 * info["nz"] / (t_auto * 1e-3) / 1e9
    print(f"| {name} | {M} | {info['nz']} | {kern} | {t_auto * 1e3:.1f} | {gf:.0f} | {info['algo_bytes'] / t_auto / 1e6 / 80:.1f} | "
          f"{best[0]} | {best[1] * 1e3:.1f} | {t_hll * 1e3:.1f} | {2.0 * info['nz'] / (t_hll * 1e-3) / 1e9:.0f} | "
          f"{pub if pub else ''} | ok |", flush=True)
