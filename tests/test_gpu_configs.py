"""BASELINE.json's configurations at THEIR sizes on one GPU (the real SuiteSparse files are not available
offline: seeded shape-matched stand-ins, or the real file when $SPMV_MTX_DIR holds it).

  C2 / C3  cant-size CSR / HLL: every kernel variant, the full y against the oracle
  C4       the nlpkkt120-size matrix cut into the 8 row blocks (CSR, reference's row partitioner) / hack
           ranges (HLL, reference's hack partitioner) an 8-GPU run uses, each block uploaded and run in turn
           on the one GPU, the re-assembled y against the oracle and against the single-handle result
  C5       the power-law fp32 matrix at full size (2^24 rows, 2.6e8 nnz): linearity, bit-reproducibility,
           row samples against the fp64-accumulated oracle loop (fp32 has no reference counterpart:
           pinned by this repo's oracle only), and its 8-way row partition
"""
import os

import numpy as np
import pytest

import sparsematrixvectormultiplication_amd as sp
from _util import FP32_NORMWISE_RTOL, assert_parity
from sparsematrixvectormultiplication_amd import synth

pytestmark = pytest.mark.gpu


def _coo(row_ptr, col):
    return np.repeat(np.arange(len(row_ptr) - 1, dtype=np.int32), np.diff(row_ptr)), np.asarray(col, np.int32)


def test_c2_c3_cant_size_every_variant_full_y(gpu, oracle):
    M, row_ptr, col, val = synth.fem_like(synth.FEM_GRID, 1)
    assert M == 62451 and 4.0e6 < row_ptr[-1] < 4.6e6      # cant: 62 451 rows, 4 007 383 nnz
    rng = np.random.default_rng(21)
    for x in (np.ones(M), rng.uniform(-1, 1, M)):           # the reference's x = 1 and a seeded one
        y_ref = oracle.csr_serial(row_ptr, col, val, x)
        with sp.CsrDevice(M, M, row_ptr, col, val) as dev:
            assert dev.info()["stream_kernel"] == 1         # the x-window kernel
            for vname, variant in sorted(sp.CSR_VARIANTS.items()) + [("auto", sp.CSR_AUTO)]:
                assert_parity(dev.spmv(x, variant), y_ref, row_ptr, col, val, x, what=f"cant-size csr-{vname}")
            with sp.HllDevice.from_csr_device(dev) as built:   # slab built on the device
                assert_parity(built.spmv(x, sp.HLL_AUTO), y_ref, row_ptr, col, val, x, what="cant-size hll (device-built)")
        r, c = _coo(row_ptr, col)
        hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, M, r, c, val))
        assert hll.num_blocks == (M + 31) // 32             # hack = 32
        with sp.HllDevice(hll) as hdev:
            for vname, variant in sorted(sp.HLL_VARIANTS.items()) + [("auto", sp.HLL_AUTO)]:
                assert_parity(hdev.spmv(x, variant), y_ref, row_ptr, col, val, x, what=f"cant-size hll-{vname}")


def test_c4_nlpkkt_size_eight_row_blocks_and_hack_ranges(gpu, oracle):
    M, row_ptr, col, val = synth.kkt_like()
    assert M == 3_542_400 and row_ptr[-1] > 9.0e7
    rng = np.random.default_rng(22)
    x = rng.uniform(-1, 1, M)
    y_ref = oracle.csr_serial(row_ptr, col, val, x)
    with sp.CsrDevice(M, M, row_ptr, col, val) as whole:
        y_whole = whole.spmv(x, sp.CSR_AUTO)
    assert_parity(y_whole, y_ref, row_ptr, col, val, x, what="nlpkkt-size single handle")
    # CSR: the reference's nnz-balanced row partition for 8 ranks
    bounds = sp.partition_rows(row_ptr, 8)
    assert bounds[0] == 0 and bounds[-1] == M and np.all(np.diff(bounds) > 0)
    nnz_part = np.diff(row_ptr[bounds].astype(np.int64))
    assert nnz_part.max() <= 1.02 * nnz_part.mean()
    y = np.full(M, np.nan)
    for p in range(8):
        lo, hi = int(bounds[p]), int(bounds[p + 1])
        with sp.CsrDevice(M, M, row_ptr, col, val, lo, hi) as part:
            info = part.info()
            assert (info["row0"], info["M_local"], info["nz"]) == (lo, hi - lo, nnz_part[p])
            part.set_x(x)
            sp.lib().spmv_hip_memset(part.y_ptr, 0xFF, M * 8)
            part.run(sp.CSR_AUTO)
            got = part.get_y()
            assert np.all(np.isnan(got[:lo])) and np.all(np.isnan(got[hi:]))   # only its own rows
            y[lo:hi] = got[lo:hi]
    assert_parity(y, y_ref, row_ptr, col, val, x, what="nlpkkt-size 8 row blocks")
    assert np.max(np.abs(y - y_whole)) <= 1e-12 * np.max(np.abs(y_ref))
    # HLL: the reference's hack partitioner for 8 ranks
    r, c = _coo(row_ptr, col)
    hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, M, r, c, val))
    hb = sp.partition_hacks(hll, 8)
    rb = sp.hack_bounds_to_rows(hb, M)
    assert hb[0] == 0 and hb[-1] == hll.num_blocks and np.all(np.diff(hb) > 0)
    yh = np.full(M, np.nan)
    for p in range(8):
        lo, hi = int(rb[p]), int(rb[p + 1])
        with sp.HllDevice(hll, int(hb[p]), int(hb[p + 1])) as part:
            part.set_x(x)
            sp.lib().spmv_hip_memset(part.y_ptr, 0xFF, M * 8)
            part.run(sp.HLL_AUTO)
            got = part.get_y()
            assert np.all(np.isnan(got[:lo])) and np.all(np.isnan(got[hi:]))
            yh[lo:hi] = got[lo:hi]
    assert_parity(yh, y_ref, row_ptr, col, val, x, what="nlpkkt-size 8 hack ranges")


def test_c5_powerlaw_fp32_full_size_properties(gpu, oracle):
    n, row_ptr, col, val = synth.powerlaw()
    assert n == 1 << 24 and row_ptr[-1] > 2.5e8 and val.dtype == np.float32
    rng = np.random.default_rng(23)
    x1 = rng.uniform(-1, 1, n).astype(np.float32)
    x2 = rng.uniform(-1, 1, n).astype(np.float32)

    def sample_check(y, x, what):
        for lo in (0, n // 2 - 3000, n - 6000):
            hi = lo + 6000
            e0, e1 = row_ptr[lo], row_ptr[hi]
            rp = (row_ptr[lo:hi + 1] - e0).astype(np.int32)
            ref = oracle.csr_f32_accum64(rp, col[e0:e1], val[e0:e1], x)
            scale = max(np.max(np.abs(ref)), 1e-30)
            assert np.max(np.abs(y[lo:hi].astype(np.float64) - ref)) <= FP32_NORMWISE_RTOL * scale, f"{what} rows {lo}..{hi}"
        # the longest rows (the split-row kernels' share), wherever they are
        lens = np.diff(row_ptr)
        for r in np.argsort(lens)[-3:]:
            e0, e1 = row_ptr[r], row_ptr[r + 1]
            ref = float(np.dot(val[e0:e1].astype(np.float64), x[col[e0:e1]].astype(np.float64)))
            bound = FP32_NORMWISE_RTOL * float(np.sum(np.abs(val[e0:e1].astype(np.float64) * x[col[e0:e1]])))
            assert abs(float(y[r]) - ref) <= bound, f"{what} long row {r} ({lens[r]} entries)"

    with sp.CsrDevice(n, n, row_ptr, col, val) as dev:
        info = dev.info()
        assert info["stream_kernel"] == 3 and info["tile_entries"] < info["nz"]       # csr_tile ...
        # ... + the long rows' tiles + (round 3) the middle tier: rows of 128 < entries <= 1024 in LDS-tall blocks
        assert info["tile_entries"] + info["tile_long_entries"] + info["tile_mid_entries"] == info["nz"]
        assert info["tile_mid_entries"] > 3e7 and info["tile_mid_rows"] > 1e5 and info["tile_mid_items"] > 0
        assert info["tile_long_rows"] > 1000 and info["tile_split_rows"] == 0
        y1 = dev.spmv(x1, sp.CSR_AUTO)
        sample_check(y1, x1, "powerlaw full size")
        assert dev.spmv(x1, sp.CSR_AUTO).tobytes() == y1.tobytes()      # no atomics: same bits every launch
        y2 = dev.spmv(x2, sp.CSR_AUTO)
        y12 = dev.spmv((2.0 * x1 - 3.0 * x2).astype(np.float32), sp.CSR_AUTO)
        scale = float(np.max(np.abs(y1)) + np.max(np.abs(y2)))
        assert np.max(np.abs(y12.astype(np.float64) - (2.0 * y1.astype(np.float64) - 3.0 * y2))) <= 2e-5 * scale
    # the 8-way row partition of config 5: nnz-balanced in spite of the skew; two of its blocks on the GPU
    bounds = sp.partition_rows(row_ptr, 8)
    nnz_part = np.diff(row_ptr[bounds].astype(np.int64))
    assert bounds[-1] == n and nnz_part.max() <= 1.05 * nnz_part.mean()
    for p in (0, 7):
        lo, hi = int(bounds[p]), int(bounds[p + 1])
        e0, e1 = row_ptr[lo], row_ptr[hi]
        from sparsematrixvectormultiplication_amd.distributed import local_row_ptr
        with sp.CsrDevice(n, n, local_row_ptr(row_ptr, lo, hi), col[e0:e1], val[e0:e1], lo, hi) as part:
            got = part.spmv(x1, sp.CSR_AUTO)
            assert got[lo:hi].tobytes() == y1[lo:hi].tobytes() or \
                np.max(np.abs(got[lo:hi].astype(np.float64) - y1[lo:hi])) <= 1e-5 * float(np.max(np.abs(y1)))


@pytest.mark.parametrize("name", ["cant.mtx", "nlpkkt120.mtx"])
def test_real_suitesparse_file_if_present(gpu, oracle, name):
    """The real files of configs 2-4, when $SPMV_MTX_DIR (or $SPMV_MTX) provides them."""
    path = os.path.join(os.environ.get("SPMV_MTX_DIR", ""), name)
    if not os.path.isfile(path):
        pytest.skip(f"{name} is not available offline (set SPMV_MTX_DIR to the directory that holds it)")
    csr = sp.convert_in_csr(sp.read_matrix_market(path))
    x = np.ones(csr.N)
    y_ref = oracle.csr_serial(csr.row_ptr, csr.col_idx, csr.values, x)
    with sp.CsrDevice.from_host(csr) as dev:
        assert_parity(dev.spmv(x, sp.CSR_AUTO), y_ref, csr.row_ptr, csr.col_idx, csr.values, x, what=name)
        with sp.HllDevice.from_csr_device(dev) as h:
            assert_parity(h.spmv(x, sp.HLL_AUTO), y_ref, csr.row_ptr, csr.col_idx, csr.values, x, what=name + " hll")
