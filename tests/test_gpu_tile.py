"""GPU parity of csr_tile (2-D tiles: row-block accumulators in LDS x column passes, dense passes staged in
LDS, split-row kernels with stripe-ordered pieces for the longest rows) against the oracle, through the C-ABI.
fp64 gate 1e-10 (norm-wise and row-wise, tests/_util.py); fp32 1e-5 norm-wise against the fp64-accumulated
oracle loop (no reference counterpart for fp32: pinned by this repo's oracle only)."""
import ctypes as C

import numpy as np
import pytest

import sparsematrixvectormultiplication_amd as sp
from _util import FP32_NORMWISE_RTOL, assert_parity
from conftest import GOLDEN_CASES, golden_path, load_golden
from sparsematrixvectormultiplication_amd.device import set_tuning

pytestmark = pytest.mark.gpu


def scattered(rng, M, N, mean, sigma=None, dtype=np.float64, lens=None):
    lens = rng.poisson(mean, M).astype(np.int64) if lens is None else np.asarray(lens, dtype=np.int64)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(M), lens)
    if sigma is None:
        col = rng.integers(0, N, rp[-1])
    else:
        col = np.clip(rows * (N - 1) // max(M - 1, 1) + np.rint(rng.normal(0, sigma, rp[-1])).astype(np.int64), 0, N - 1)
    order = np.lexsort((col, rows))
    return rp, col[order].astype(np.int32), rng.uniform(-1, 1, rp[-1]).astype(dtype)


class tuned:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        for k, v in self.kv.items():
            set_tuning(k, v)

    def __exit__(self, *exc):
        defaults = {"stream_tile": -1, "tile_rows": 0, "tile_lmax": 1536, "tile_density": 4, "stream_kind": -1,
                    "tile_balance": 1, "tile_long": 1, "tile_pack": 1, "stream_local": 1, "tile_places": 0,
                    "tile_streams": 1, "tile_fit": 1, "tile_plan_on_device": 1, "place_tries": 12, "tile_min_pass": 256, "tile_mid": 1,
                    "tile_expand": -1}
        for k in self.kv:
            set_tuning(k, defaults[k])


def check(dev, x, y_ref, rp, col, val, dtype, what):
    item = np.dtype(dtype).itemsize
    first = None
    for rep in range(3):
        sp.lib().spmv_hip_memset(dev.y_ptr, 0xFF, len(y_ref) * item)  # every row must be written
        dev.set_x(x)
        dev.run(sp.CSR_STREAM)
        y = dev.get_y()
        if dtype == np.float64:
            assert_parity(y, y_ref, rp, col, val, x, what=f"{what} rep={rep}")
        else:
            assert np.all(np.isfinite(y))
            err = np.max(np.abs(y.astype(np.float64) - y_ref)) / max(np.max(np.abs(y_ref)), 1e-300)
            assert err <= FP32_NORMWISE_RTOL, f"{what}: {err:.3e}"
        first = y if first is None else first
        assert y.tobytes() == first.tobytes(), f"{what}: result changed between launches"
    return first


def reference(oracle, rp, col, val, x, dtype):
    return oracle.csr_serial(rp, col, val, x) if dtype == np.float64 else oracle.csr_f32_accum64(rp, col, val, x)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("rows", [256, 1024])
def test_tile_kernel_gather_passes_uniform_columns(gpu, oracle, dtype, rows):
    rng = np.random.default_rng(5)
    M, N = 7001, 2_000_003
    rp, col, val = scattered(rng, M, N, 18, dtype=dtype)
    x = rng.uniform(-1, 1, N).astype(dtype)
    y_ref = reference(oracle, rp, col, val, x, dtype)
    with tuned(stream_tile=1, tile_rows=rows):
        with sp.CsrDevice(M, N, rp, col, val) as dev:
            info = dev.info()
            assert info["stream_kernel"] == 3 and info["tile_blocks"] >= (M + rows - 1) // rows
            assert info["tile_entries"] == rp[-1] and info["tile_staged_entries"] == 0
            if rows == 1024:
                assert info["tile_passes"] > info["tile_blocks"]   # several column passes per block
            y = check(dev, x, y_ref, rp, col, val, dtype, f"uniform rows={rows}")
            # the gather kernel on the same handle agrees within the gate too
            with tuned(stream_kind=0):
                y0 = dev.spmv(x, sp.CSR_STREAM)
            scale = np.max(np.abs(y_ref))
            assert np.max(np.abs(y.astype(np.float64) - y0)) <= (1e-10 if dtype == np.float64 else 1e-5) * scale


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("N", [2_000_003, 8192 * 5, 4097])
def test_tile_kernel_expanded_x_gives_the_gather_passes_bits(gpu, oracle, dtype, N):
    """Round 3: a plan with gather passes runs on an EXPANDED x -- tile_expand writes every entry's x value into its pass's
    segment of a second vector x' (walking the entries by 32 KiB slice of x, the slice in LDS), and the packed kernel
    runs over x' with every pass's segment as its window.  The same products in the same order: the bits of the gather
    passes, which the same handle runs when "tile_expand" is 0 at launch.  N: a last slice of 3 values, whole slices,
    one short slice."""
    rng = np.random.default_rng(77)
    M = 9001
    lens = rng.poisson(14, M)
    lens[::997] = 3000                                    # a few long rows: runs across lanes, wavefronts and passes
    lens[5] = 0
    rp, col, val = scattered(rng, M, N, 14, dtype=dtype, lens=lens)
    x = rng.uniform(-1, 1, N).astype(dtype)
    y_ref = reference(oracle, rp, col, val, x, dtype)
    with tuned(stream_tile=1, tile_rows=1024, tile_expand=1, tile_lmax=4096, tile_density=0, tile_pack=0, stream_local=0):
        with sp.CsrDevice(M, N, rp, col, val) as dev:
            info = dev.info()
            assert info["stream_kernel"] == 3 and info["tile_staged_entries"] == 0
            assert info["tile_expanded_entries"] >= info["tile_entries"] == rp[-1]
            y = check(dev, x, y_ref, rp, col, val, dtype, f"expanded N={N}")
            set_tuning("tile_expand", 0)                  # the same handle, the gather passes
            y_gather = check(dev, x, y_ref, rp, col, val, dtype, f"gather N={N}")
            assert y.tobytes() == y_gather.tobytes()
            set_tuning("tile_expand", 1)
            # a foreign, misaligned x: the expansion copies 16-byte pieces -- such a launch takes the gather passes
            L, item = sp.lib(), x.itemsize
            buf, ybuf = C.c_void_p(), C.c_void_p()
            assert L.spmv_hip_malloc(C.byref(buf), (N + 64) * item) == 0 and L.spmv_hip_malloc(C.byref(ybuf), M * item) == 0
            try:
                for shift in (item, 0):
                    xp = C.c_void_p(buf.value + shift)
                    assert L.spmv_hip_memcpy_h2d(xp, x.ctypes.data_as(C.c_void_p), N * item) == 0
                    assert L.spmv_hip_memset(ybuf, 0xFF, M * item) == 0
                    assert L.spmv_hip_csr_run_on(dev.h, sp.CSR_STREAM, xp, ybuf, None) == 0
                    y2 = np.empty(M, dtype)
                    assert L.spmv_hip_memcpy_d2h(y2.ctypes.data_as(C.c_void_p), ybuf, M * item) == 0
                    assert y2.tobytes() == y.tobytes(), f"run_on shift={shift}"
            finally:
                L.spmv_hip_free(buf)
                L.spmv_hip_free(ybuf)
    # auto: a small plan is not expanded
    with tuned(stream_tile=1, tile_rows=1024):
        with sp.CsrDevice(M, N, rp, col, val) as dev:
            assert dev.info()["tile_expanded_entries"] == 0


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("pack", [1, 0])
@pytest.mark.parametrize("mean,sigma", [(3, 1500), (30, 9000), (12, 200)])
def test_tile_kernel_staged_passes_banded_columns(gpu, oracle, dtype, mean, sigma, pack):
    rng = np.random.default_rng(mean)
    M = N = 40_000
    rp, col, val = scattered(rng, M, N, mean, sigma=sigma, dtype=dtype)
    x = rng.uniform(-1, 1, N).astype(dtype)
    y_ref = reference(oracle, rp, col, val, x, dtype)
    # (the narrowest of these bands would get an x-window plan: switched off, the tile kernel is what is under test)
    with tuned(stream_tile=1, tile_rows=2048, tile_pack=pack, stream_local=0):
        with sp.CsrDevice(M, N, rp, col, val) as dev:
            info = dev.info()
            assert info["local_blocks"] == 0
            assert info["stream_kernel"] == 3 and info["tile_staged_entries"] > 0.5 * info["tile_entries"]
            check(dev, x, y_ref, rp, col, val, dtype, f"band mean={mean} sigma={sigma}")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("far,N", [(0.0, 60_000), (0.004, 60_000), (0.3, 3_000_000)])
def test_tile_kernel_packed_plan_outliers_and_fallback(gpu, oracle, dtype, far, N):
    """Upload's choice between the two kinds of plan.  A band (with a few entries anywhere in a matrix of modest
    width) gets the PACKED plan -- every pass cut at the window and staged, the kernel instantiation without gather
    code -- which shows as staged entries == entries; with a third of the entries anywhere in a wide matrix every
    stray entry would cost a window of its own, and the plan keeps gather passes.  Both against the oracle, x at
    every alignment the staging loads can meet."""
    rng = np.random.default_rng(int(far * 1000) + 5)
    M = 60_000
    rp, col, val = scattered(rng, M, N, 9, sigma=2500, dtype=dtype)
    stray = rng.random(rp[-1]) < far
    col = col.copy()
    col[stray] = rng.integers(0, N, int(stray.sum()))
    rows = np.repeat(np.arange(M), np.diff(rp))
    order = np.lexsort((col, rows))
    col, val = col[order], val[order]
    x = rng.uniform(-1, 1, N).astype(dtype)
    y_ref = reference(oracle, rp, col, val, x, dtype)
    item = np.dtype(dtype).itemsize
    L = sp.lib()
    with tuned(stream_tile=1, tile_rows=2048):
        with sp.CsrDevice(M, N, rp, col, val) as dev:
            info = dev.info()
            assert info["stream_kernel"] == 3 and info["tile_entries"] == rp[-1]
            # (a packed plan stages every entry of its tiles; what sat in windows too sparse for a pass is its remainder)
            packed = info["tile_staged_entries"] + info["tile_remainder_entries"] == info["tile_entries"]
            assert packed == (far < 0.1), info
            if packed:
                assert info["tile_staged_cols"] > 0
            if far == 0.004:  # the stray entries sit in windows too sparse for a pass: the remainder kernel adds them
                assert 0 < info["tile_remainder_entries"] <= 0.04 * rp[-1], info["tile_remainder_entries"]
            check(dev, x, y_ref, rp, col, val, dtype, f"packed={packed} far={far}")
            buf, ybuf = C.c_void_p(), C.c_void_p()
            assert L.spmv_hip_malloc(C.byref(buf), (N + 64) * item) == 0 and L.spmv_hip_malloc(C.byref(ybuf), M * item) == 0
            try:
                for shift in (item, 3 * item, 16):
                    xp = C.c_void_p(buf.value + shift)
                    assert L.spmv_hip_memcpy_h2d(xp, x.ctypes.data_as(C.c_void_p), N * item) == 0
                    assert L.spmv_hip_memset(ybuf, 0xFF, M * item) == 0
                    assert L.spmv_hip_csr_run_on(dev.h, sp.CSR_STREAM, xp, ybuf, None) == 0
                    y = np.empty(M, dtype=dtype)
                    assert L.spmv_hip_memcpy_d2h(y.ctypes.data_as(C.c_void_p), ybuf, M * item) == 0
                    err = np.max(np.abs(y.astype(np.float64) - y_ref)) / np.max(np.abs(y_ref))
                    assert err <= (1e-10 if dtype == np.float64 else FP32_NORMWISE_RTOL), f"shift={shift}: {err:.3e}"
            finally:
                L.spmv_hip_free(buf)
                L.spmv_hip_free(ybuf)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tile_kernel_skewed_rows_sub_runs_and_split_rows(gpu, oracle, dtype):
    """Power-law row lengths: runs longer than one lane's share (sub-runs + multi heads), rows beyond the tile
    limit (split-row kernels, pieces cut at column stripes), empty rows, a last block that is not full."""
    rng = np.random.default_rng(9)
    M, N = 6500, 3_000_000
    lens = np.minimum((1.08 / rng.random(M)).astype(np.int64), 60000)
    lens[rng.random(M) < 0.1] = 0
    lens[[11, 3000, 6499]] = [50000, 5000, 900]
    rp, col, val = scattered(rng, M, N, 0, dtype=dtype, lens=lens)
    x = rng.uniform(-1, 1, N).astype(dtype)
    y_ref = reference(oracle, rp, col, val, x, dtype)
    for lmax, balance, long_plan in ((16384, 1, 0), (700, 1, 0), (700, 0, 2), (40, 1, 2), (16384, 1, 2)):
        with tuned(stream_tile=1, tile_rows=512, tile_lmax=lmax, tile_balance=balance, tile_long=long_plan):
            with sp.CsrDevice(M, N, rp, col, val) as dev:
                info = dev.info()
                assert info["stream_kernel"] == 3 and info["tile_entries"] == int(lens[lens <= lmax].sum())
                if long_plan:    # the rows beyond the limit: their own tiles (work items + slabs), nothing left over
                    assert info["tile_long_rows"] == int((lens > lmax).sum()) > 0 and info["tile_split_rows"] == 0
                    assert info["tile_long_entries"] == int(lens[lens > lmax].sum()) and info["tile_long_items"] >= 1
                else:            # ... or the split-row kernels
                    assert info["tile_split_rows"] == int((lens > lmax).sum()) > 0 and info["tile_long_rows"] == 0
                check(dev, x, y_ref, rp, col, val, dtype, f"skewed lmax={lmax} balance={balance} long={long_plan}")


DIGEST_ARRAYS = ("tcol", "tkey", "tval", "pass", "stream_pass", "block_row", "stream_block", "sblock_rows", "rem_row", "rem_ptr",
                 "rem_col", "rem_val", "lt.tcol", "lt.tkey", "lt.tval", "lt.pass", "lt.block_row", "lt.block_pass", "lt.work",
                 "lt.item_first", "lt.row_map", "lt.block_of_row", "mt.tcol", "mt.tkey", "mt.tval", "mt.pass", "mt.block_row",
                 "mt.block_pass", "mt.work", "mt.item_first", "mt.row_map", "mt.block_of_row")


def digests_by_builder(make_handle):
    """the handle's tile-plan digest with the plan built by host threads and by the device kernels"""
    out = []
    for on_device in (0, 1):
        with tuned(tile_plan_on_device=on_device):
            with make_handle() as dev:
                out.append((dev.tile_digest(), dev.info()))
    return out


def assert_same_plan(host, device, what):
    (dh, ih), (dd, idv) = host, device
    for name, a, b in zip(DIGEST_ARRAYS, dh, dd):
        assert a == b, f"{what}: array {name} differs between the host-built and the device-built plan: {a} vs {b}"
    for key in ("tile_blocks", "tile_passes", "tile_entries", "tile_staged_entries", "tile_staged_cols", "tile_remainder_entries",
                "tile_long_rows", "tile_long_items", "tile_long_entries", "tile_mid_rows", "tile_mid_items", "tile_mid_entries",
                "tile_split_rows", "stream_bytes"):
        assert ih[key] == idv[key], (what, key, ih[key], idv[key])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_tile_plan_built_on_the_device_is_the_host_plan(gpu, oracle, dtype):
    """the csr_tile plan built by kernels from the CSR arrays in HBM (tile_plan_device.hpp: keys,
    segmented sort, greedy cuts per row block, per-pass re-sort, remainder) is BYTE FOR BYTE the plan tile_plan.hpp
    builds on host threads -- every array of both plans (ordinary tiles and the long rows' own), on scattered columns
    (gather passes), a band (packed plan), a band with stray entries (packed plan + remainder), skewed rows (long-row
    plan, split rows), empty rows and empty blocks, blocks of a single pass; and the device-built plan runs right."""
    rng = np.random.default_rng(71)
    cases = []
    M, N = 9001, 2_000_003
    rp, col, val = scattered(rng, M, N, 18, dtype=dtype)
    cases.append(("uniform", M, N, rp, col, val, dict(stream_tile=1, tile_rows=1024, stream_local=0)))
    M = N = 60_000
    rp, col, val = scattered(rng, M, N, 9, sigma=2500, dtype=dtype)
    cases.append(("band", M, N, rp, col, val, dict(stream_tile=1, tile_rows=2048, stream_local=0)))
    stray = rng.random(rp[-1]) < 0.004
    col2 = col.copy()
    col2[stray] = rng.integers(0, N, int(stray.sum()))
    rows = np.repeat(np.arange(M), np.diff(rp))
    order = np.lexsort((col2, rows))
    cases.append(("band + stray entries", M, N, rp, col2[order], val[order], dict(stream_tile=1, tile_rows=2048, stream_local=0)))
    M, N = 6500, 3_000_000
    lens = np.minimum((1.08 / rng.random(M)).astype(np.int64), 60000)
    lens[rng.random(M) < 0.1] = 0
    lens[1000:1700] = 0                      # a whole row block without entries
    lens[[11, 3000, 6499]] = [50000, 5000, 900]
    rp, col, val = scattered(rng, M, N, 0, dtype=dtype, lens=lens)
    cases.append(("skewed, long-row plan", M, N, rp, col, val, dict(stream_tile=1, tile_rows=512, tile_lmax=700, tile_long=2)))
    cases.append(("skewed, split rows", M, N, rp, col, val, dict(stream_tile=1, tile_rows=512, tile_lmax=700, tile_long=0)))
    cases.append(("skewed, auto block height", M, N, rp, col, val, dict(stream_tile=1, tile_lmax=40, tile_long=2, tile_places=64)))
    M, N = 3000, 50_000
    rp, col, val = scattered(rng, M, N, 3, sigma=300, dtype=dtype)   # blocks that hold a single pass in CSR order
    cases.append(("short rows, one pass per block", M, N, rp, col, val, dict(stream_tile=1, tile_rows=256, stream_local=0)))
    for what, M, N, rp, col, val, knobs in cases:
        with tuned(**knobs):
            host, device = digests_by_builder(lambda: sp.CsrDevice(M, N, rp, col, val))
            assert host[1]["stream_kernel"] == 3, what
            assert_same_plan(host, device, what)
            x = rng.uniform(-1, 1, N).astype(dtype)
            with tuned(tile_plan_on_device=1):
                with sp.CsrDevice(M, N, rp, col, val) as dev:
                    check(dev, x, reference(oracle, rp, col, val, x, dtype), rp, col, val, dtype, what + " (device-built plan)")


def test_hll_tile_plan_built_on_the_device_is_the_host_plan(gpu, oracle):
    """The same for the tile plan over an HLL slab's rows (padding slots included): host-uploaded slab and the slab
    built on the device from a resident CSR matrix."""
    from _util import coo_from_csr
    rng = np.random.default_rng(29)
    M, N = 9001, 1_500_000
    rp, col, val = scattered(rng, M, N, 16)
    r, c, v = coo_from_csr(rp, col, val)
    hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, N, r, c, v))
    with tuned(stream_tile=1, tile_rows=1024, stream_local=0):
        host, device = digests_by_builder(lambda: sp.HllDevice(hll))
        assert host[1]["stream_kernel"] == 2
        assert_same_plan(host, device, "hll slab")
        with sp.CsrDevice(M, N, rp, col, val) as cdev:
            built = digests_by_builder(lambda: sp.HllDevice.from_csr_device(cdev))
        assert built[1][1]["stream_kernel"] == 2
        assert_same_plan(host, built[1], "hll slab built on the device")


def test_tile_kernel_row_block_handle_and_foreign_x(gpu, oracle):
    """A row block keeps global columns and writes its rows of the full y; run_on with a 16-byte aligned x
    stages, with a misaligned x it gathers (same result within the gate)."""
    rng = np.random.default_rng(3)
    M = N = 30_000
    rp, col, val = scattered(rng, M, N, 14, sigma=2500)
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_serial(rp, col, val, x)
    L = sp.lib()
    with tuned(stream_tile=1, tile_rows=1024):
        with sp.CsrDevice(M, N, rp, col, val, row0=5000, row1=27001) as part:
            assert part.info()["stream_kernel"] == 3
            y = part.spmv(x, sp.CSR_STREAM)
            assert_parity(y[5000:27001], y_ref[5000:27001], rp[5000:27002] - rp[5000], col[rp[5000]:rp[27001]],
                          val[rp[5000]:rp[27001]], x, what="tile row block")
        with sp.CsrDevice(M, N, rp, col, val) as dev:
            assert dev.info()["tile_staged_entries"] > 0
            buf, ybuf = C.c_void_p(), C.c_void_p()
            assert L.spmv_hip_malloc(C.byref(buf), (N + 64) * 8) == 0 and L.spmv_hip_malloc(C.byref(ybuf), M * 8) == 0
            try:
                for shift in (0, 8, 16):
                    xp = C.c_void_p(buf.value + shift)
                    assert L.spmv_hip_memcpy_h2d(xp, x.ctypes.data_as(C.c_void_p), N * 8) == 0
                    assert L.spmv_hip_memset(ybuf, 0xFF, M * 8) == 0
                    assert L.spmv_hip_csr_run_on(dev.h, sp.CSR_STREAM, xp, ybuf, None) == 0
                    y = np.empty(M)
                    assert L.spmv_hip_memcpy_d2h(y.ctypes.data_as(C.c_void_p), ybuf, M * 8) == 0
                    assert_parity(y, y_ref, rp, col, val, x, what=f"tile run_on shift={shift}")
            finally:
                L.spmv_hip_free(buf)
                L.spmv_hip_free(ybuf)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_tile_kernel_on_reference_golden(gpu, name):
    """The reference's own fixtures through the tile kernel (forced; tiny matrices: one block)."""
    g = load_golden(name)
    csr = sp.convert_in_csr(sp.read_matrix_market(golden_path(name)))
    with tuned(stream_tile=1, tile_rows=256, tile_lmax=100):
        with sp.CsrDevice.from_host(csr) as dev:
            if csr.nz and not dev.info()["local_blocks"]:
                assert dev.info()["stream_kernel"] == 3
            for x, key in ((np.ones(csr.N), "y_ones"), (g["x_rand"], "y_rand")):
                assert_parity(dev.spmv(x, sp.CSR_STREAM), g[key], csr.row_ptr, csr.col_idx, csr.values, x,
                              what=f"{name}/{key}/tile")


@pytest.mark.parametrize("mean,sigma", [(16, None), (12, 6000)])
def test_hll_slab_rows_through_the_tile_kernel(gpu, oracle, mean, sigma):
    """An HLL slab whose columns are too scattered for the x-window plan gets the tile plan over its rows (padding
    slots included): host-built slab and device-built slab, whole matrix and a hack range, against the oracle and
    bit-reproducible; the gather kernel hll_lds on the same handle agrees within the gate."""
    from _util import coo_from_csr
    rng = np.random.default_rng(17 + mean)
    M, N = 9001, (1_500_000 if sigma is None else 9001)
    rp, col, val = scattered(rng, M, N, mean, sigma=sigma)
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_serial(rp, col, val, x)
    r, c, v = coo_from_csr(rp, col, val)
    hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, N, r, c, v))
    with tuned(stream_tile=1, tile_rows=1024, stream_local=0):  # (no x-window plan: the tile kernel is under test)
        with sp.HllDevice(hll) as dev:
            info = dev.info()
            assert info["local_blocks"] == 0
            assert info["stream_kernel"] == 2 and info["tile_entries"] + info["tile_long_entries"] == info["slots"]
            first = None
            for rep in range(3):
                sp.lib().spmv_hip_memset(dev.y_ptr, 0xFF, M * 8)
                y = dev.spmv(x, sp.HLL_LDS)
                assert_parity(y, y_ref, rp, col, val, x, what=f"hll tiles rep={rep}")
                first = y if first is None else first
                assert y.tobytes() == first.tobytes()
            with tuned(stream_kind=0):
                assert_parity(dev.spmv(x, sp.HLL_LDS), y_ref, rp, col, val, x, what="hll_lds on the same handle")
            for vname, variant in sorted(sp.HLL_VARIANTS.items()):
                assert_parity(dev.spmv(x, variant), y_ref, rp, col, val, x, what=f"hll {vname}")
        # a hack range keeps global rows; the slab built on the device gets the plan as well
        hb = sp.partition_hacks(hll, 3)
        rb = sp.hack_bounds_to_rows(hb, M)
        with sp.HllDevice(hll, int(hb[1]), int(hb[2])) as part:
            assert part.info()["stream_kernel"] == 2
            y = part.spmv(x, sp.HLL_LDS)
            lo, hi = int(rb[1]), int(rb[2])
            assert_parity(y[lo:hi], y_ref[lo:hi], rp[lo:hi + 1] - rp[lo], col[rp[lo]:rp[hi]], val[rp[lo]:rp[hi]], x,
                          what="hll tiles, hack range")
        with sp.CsrDevice(M, N, rp, col, val) as cdev, sp.HllDevice.from_csr_device(cdev) as built:
            assert built.info()["stream_kernel"] == 2
            assert_parity(built.spmv(x, sp.HLL_AUTO), y_ref, rp, col, val, x, what="hll tiles, device-built slab")


def test_tile_kernel_inside_a_graph_replay_and_power_iteration(gpu, oracle):
    """The tile kernel captured into a hipGraph (spmv_hip_csr_time_graph) and inside the on-device power iteration:
    launches must not need anything that a capture forbids; the result stays the oracle's."""
    rng = np.random.default_rng(41)
    M = N = 20_000
    rp, col, val = scattered(rng, M, N, 10, sigma=3000)
    x = rng.uniform(0.5, 1.0, N)
    y_ref = oracle.csr_serial(rp, col, val, x)
    with tuned(stream_tile=1, tile_rows=1024):
        with sp.CsrDevice(M, N, rp, col, val) as dev:
            assert dev.info()["stream_kernel"] == 3
            dev.set_x(x)
            assert dev.time_graph(sp.CSR_STREAM, 5, 3) > 0
            assert_parity(dev.get_y(), y_ref, rp, col, val, x, what="tile kernel replayed from a graph")
            dev.set_x(x)
            lam, _ = dev.power_iterate(3, use_graph=True)
            xs = x.copy()
            for _ in range(3):
                ys = oracle.csr_serial(rp, col, val, xs)
                lam_ref = float(np.linalg.norm(ys))
                xs = ys / lam_ref
            assert abs(lam - lam_ref) <= 1e-10 * lam_ref


def test_tile_kernel_random_shapes(gpu, oracle):
    """Seeded sweep over shapes the fixed cases do not hit: rows per block above and below M, N that is not a
    multiple of 4 (the last window piece ends inside a 16-byte piece), bands at the matrix edges, stray entries,
    empty rows, rows longer than a pass, both kinds of plan, fp64 and fp32 -- every case against the oracle, with y
    poisoned first and the launch repeated (bit-reproducible)."""
    import os
    rng = np.random.default_rng(int(os.environ.get("TILE_FUZZ_SEED", "20260")))  # (other seeds / more cases: by hand)
    packed_seen = plain_seen = expanded_seen = 0
    for case in range(int(os.environ.get("TILE_FUZZ_CASES", "64"))):
        dtype = np.float64 if case % 3 else np.float32
        M = int(rng.integers(1, 30_000))
        N = int(rng.integers(1, 200_000)) if case % 4 else M
        mean = float(rng.choice([0.5, 2, 3, 9, 30, 120]))
        if M * mean > 600_000:
            mean = 600_000 / M
        sigma = None if case % 5 == 0 else float(rng.choice([5, 300, 3000, 40_000]))
        rp, col, val = scattered(rng, M, N, mean, sigma=sigma, dtype=dtype)
        if rp[-1] and case % 7 == 3:  # a few stray entries anywhere
            stray = rng.random(rp[-1]) < 0.01
            col = col.copy()
            col[stray] = rng.integers(0, N, int(stray.sum()))
            rows = np.repeat(np.arange(M), np.diff(rp))
            order = np.lexsort((col, rows))
            col, val = col[order], val[order]
        x = rng.uniform(-1, 1, N).astype(dtype)
        y_ref = reference(oracle, rp, col, val, x, dtype)
        rows_per_block = int(rng.choice([256, 512, 2048, 4096]))
        # (tile_places: streams of many blocks on these small matrices -- the chip's own 512 / 256 places would give
        # every block its own workgroup; tile_rows 0 with few places exercises the fitted block count as well)
        places = int(rng.choice([0, 8, 16, 64]))
        if case % 8 == 7:
            rows_per_block = 0
        # (tile_expand 1: plans with gather passes run on an expanded x whatever their size and type -- mixed plans with
        # staged passes, empty blocks and single-entry passes included)
        with tuned(stream_tile=1, stream_local=0, tile_rows=rows_per_block, tile_pack=int(case % 6 != 5),
                   tile_lmax=int(rng.choice([64, 1024])), tile_places=places, tile_streams=int(case % 11 != 10),
                   tile_expand=int(case % 2)):
            with sp.CsrDevice(M, N, rp, col, val) as dev:
                info = dev.info()
                if info["stream_kernel"] != 3:
                    continue  # (an empty matrix gets no plan)
                expanded_seen += info["tile_expanded_entries"] > 0
                if info["tile_entries"] and info["tile_staged_entries"] + info["tile_remainder_entries"] == info["tile_entries"]:
                    packed_seen += 1
                else:
                    plain_seen += 1
                check(dev, x, y_ref, rp, col, val, dtype,
                      f"case {case}: M={M} N={N} mean={mean} sigma={sigma} rows={rows_per_block} places={places} "
                      f"{np.dtype(dtype).name}")
    assert packed_seen >= 8 and plain_seen >= 8 and expanded_seen >= 4, (packed_seen, plain_seen, expanded_seen)
