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

want = [w for w in sys.argv[1:] if not w.startswith("--")]
sp.hip_init(0)


def rounds(dev, variants, iters, nrounds=3):
    """mean time of every variant, measured in interleaved rounds (AUTO first in every round: whoever runs first on a cold
    chip looked up to 6 % slower than the same kernel measured later in round 2's version of this table); the minimum of a
    variant's round means is its time"""
    best = {}
    for _ in range(nrounds):
        for name, v in variants:
            t = float(dev.time(v, 3, iters, zero_y=False).mean())
            best[name] = min(best.get(name, 1e9), t)
    return best


print("| stand-in for | rows | nnz | AUTO kernel | us | GFLOP/s | % of 8 TB/s | best CSR variant | its us | AUTO / best | HLL AUTO kernel | HLL AUTO us | "
      "best HLL variant | its us | HLL AUTO / best | reference's best CUDA GFLOP/s on the real matrix | parity |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
worst_csr = worst_hll = 0.0
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
        iters = 60 if info["nz"] < 5_000_000 else 12
        variants = [("auto", sp.CSR_AUTO)] + [(k, v) for k, v in sp.CSR_VARIANTS.items()
                                              if not (k in ("wave_row", "thread_row") and info["nz"] > 40_000_000)]
        for vname, v in variants:
            yv = dev.spmv(x, v)
            assert np.max(np.abs(yv - y_ref)) <= 1e-10 * scale, (name, vname)
        dev.set_x(x)
        tc = rounds(dev, variants, iters)
        t_auto = tc["auto"]
        best = min(tc.items(), key=lambda kv: kv[1])
        with sp.HllDevice.from_csr_device(dev) as h:
            hinfo = h.info()
            hvariants = [("auto", sp.HLL_AUTO)] + [(k, v) for k, v in sp.HLL_VARIANTS.items()
                                                   if not (k == "thread_row" and hinfo["slots"] > 40_000_000)]
            for vname, v in hvariants:
                yh = h.spmv(x, v)
                assert np.max(np.abs(yh - y_ref)) <= 1e-10 * scale, ("HLL", name, vname)
            h.set_x(x)
            th = rounds(h, hvariants, iters)
            hbest = min(th.items(), key=lambda kv: kv[1])
    kern = sp.device.CSR_STREAM_KERNELS[info["stream_kernel"]] if info["auto_variant"] == sp.CSR_STREAM else \
        [k for k, v in sp.CSR_VARIANTS.items() if v == info["auto_variant"]][0]
    hkern = sp.device.HLL_LDS_KERNELS[hinfo["stream_kernel"]] if hinfo["auto_variant"] == sp.HLL_LDS else \
        [k for k, v in sp.HLL_VARIANTS.items() if v == hinfo["auto_variant"]][0]
    gf = 2.0 * info["nz"] / (t_auto * 1e-3) / 1e9
    worst_csr = max(worst_csr, t_auto / best[1])
    worst_hll = max(worst_hll, th["auto"] / hbest[1])
    print(f"| {name} | {M} | {info['nz']} | {kern} | {t_auto * 1e3:.1f} | {gf:.0f} | {info['algo_bytes'] / t_auto / 1e6 / 80:.1f} | "
          f"{best[0]} | {best[1] * 1e3:.1f} | {t_auto / best[1]:.2f} | {hkern} | {th['auto'] * 1e3:.1f} | {hbest[0]} | {hbest[1] * 1e3:.1f} | "
          f"{th['auto'] / hbest[1]:.2f} | {pub if pub else ''} | ok |", flush=True)
print(f"\nworst AUTO / best over the list: CSR {worst_csr:.3f}, HLL {worst_hll:.3f}")
