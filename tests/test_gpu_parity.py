"""GPU parity: the HIP kernels, called through the C-ABI, against the reference's
golden vectors and against the oracle on seeded inputs.  fp64 gate: 1e-10 (see _util)."""
import numpy as np
import pytest

import sparsematrixvectormultiplication_amd as sp
from _util import FP32_NORMWISE_RTOL, assert_parity, coo_from_csr, random_csr
from conftest import GOLDEN_CASES, golden_path, load_golden

pytestmark = pytest.mark.gpu

CSR_V = sorted(sp.CSR_VARIANTS.items())
HLL_V = sorted(sp.HLL_VARIANTS.items())


# ------------------------------------------------------------------ golden
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_csr_kernels_match_reference_golden(gpu, name):
    """.mtx -> C parser -> C builder -> HIP kernel, vs y computed by the compiled reference."""
    g = load_golden(name)
    csr = sp.convert_in_csr(sp.read_matrix_market(golden_path(name)))
    with sp.CsrDevice.from_host(csr) as dev:
        assert dev.info()["nz"] == int(g["nz"])
        for x, key in ((np.ones(csr.N), "y_ones"), (g["x_rand"], "y_rand")):
            for vname, variant in CSR_V + [("auto", sp.CSR_AUTO)]:
                y = dev.spmv(x, variant)
                assert_parity(y, g[key], csr.row_ptr, csr.col_idx, csr.values, x,
                              what=f"{name}/{key}/csr-{vname}")


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_hll_kernels_match_reference_golden(gpu, name):
    g = load_golden(name)
    pre = sp.read_matrix_market(golden_path(name))
    csr = sp.convert_in_csr(pre)
    hll = sp.convert_to_hll(pre)
    with sp.HllDevice(hll) as dev:
        info = dev.info()
        assert info["hacks"] == hll.num_blocks and info["slots"] == hll.slots
        for x, key in ((np.ones(csr.N), "y_ones"), (g["x_rand"], "y_rand")):
            for vname, variant in HLL_V + [("auto", sp.HLL_AUTO)]:
                y = dev.spmv(x, variant)
                # every HLL result is compared with the serial CSR result, as the
                # reference's drivers do (main_cuda.cu:586-594)
                assert_parity(y, g[key], csr.row_ptr, csr.col_idx, csr.values, x,
                              what=f"{name}/{key}/hll-{vname}")


# ------------------------------------------------------- seeded vs oracle
SHAPES = [
    # M, N, mean row, max row, empty fraction
    (1, 1, 1, 1, 0.0),
    (63, 70, 3, 8, 0.3),          # M not a multiple of 32/64, short rows
    (64, 64, 20, 40, 0.0),
    (1000, 1000, 27, 60, 0.05),   # nlpkkt-like row lengths
    (777, 2000, 64, 130, 0.0),    # cant-like row lengths, rectangular
    (5000, 5000, 2, 5, 0.5),      # roadNet-like: mostly empty / tiny rows
    (300, 9000, 700, 2000, 0.0),  # rows near the stream kernel's stage size
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s[:3])))
def test_csr_kernels_match_oracle_seeded(gpu, oracle, shape):
    M, N, mean, mx, empty = shape
    rng = np.random.default_rng(hash(shape) % (2 ** 32))
    row_ptr, col, val = random_csr(rng, M, N, mean, mx, empty)
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_serial(row_ptr, col, val, x)
    with sp.CsrDevice(M, N, row_ptr, col, val) as dev:
        for vname, variant in CSR_V:
            assert_parity(dev.spmv(x, variant), y_ref, row_ptr, col, val, x, what=f"csr-{vname}")


@pytest.mark.parametrize("mean", [1, 3, 7, 9, 12, 15, 17, 24, 40, 100, 300])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_stream_kernel_block_shapes_repeatable(gpu, oracle, mean, dtype):
    """The stream kernels pick their lane mapping from the rows a workgroup holds
    (> 256, 128..255, < 128 rows per 2048 / 4096 staged entries).  Sweep the mean row
    length across all regimes, both stage sizes and both second halves, launch
    repeatedly into a poisoned y: every row must be written, with the same value
    each time (regression: idle lanes once stored zeros over live rows)."""
    from sparsematrixvectormultiplication_amd.device import set_tuning
    rng = np.random.default_rng(1000 + mean)
    M, N = 6000, 7000
    row_ptr, col, val = random_csr(rng, M, N, mean, 4 * mean + 4, 0.03, dtype=dtype)
    x = rng.uniform(-1, 1, N).astype(dtype)
    if dtype == np.float64:
        y_ref = oracle.csr_serial(row_ptr, col, val, x)
    else:
        y_ref = oracle.csr_f32_accum64(row_ptr, col, val, x)
    item = np.dtype(dtype).itemsize
    try:
        # (the row-walk / pipe / ring variants live behind `make EXPERIMENTAL=1` since round 2)
        for cap, walk, blk in ((2048, 0, 256), (4096, 0, 256), (4096, 0, 512), (8192, 0, 512), (8192, 0, 1024),
                               (1024, 0, 256), (3072, 0, 256)):
            set_tuning("stream_cap", cap)
            with sp.CsrDevice(M, N, row_ptr, col, val) as dev:
                dev.set_x(x)
                if True:
                    set_tuning("stream_kind", walk)
                    set_tuning("stream_block", blk if blk in (256, 512, 1024) else 256)
                    set_tuning("pipe_wgs_per_cu", 2 if blk == 257 else (1 if walk == 4 else 5))
                    first = None
                    for rep in range(4):
                        sp.lib().spmv_hip_memset(dev.y_ptr, 0xFF, M * item)  # NaN pattern
                        dev.run(sp.CSR_STREAM)
                        y = dev.get_y()
                        if dtype == np.float64:
                            assert_parity(y, y_ref, row_ptr, col, val, x,
                                          what=f"cap={cap} kind={walk} block={blk} rep={rep}")
                        else:
                            err = np.max(np.abs(y.astype(np.float64) - y_ref)) / np.max(np.abs(y_ref))
                            assert err <= FP32_NORMWISE_RTOL, f"cap={cap} kind={walk}: {err:.3e}"
                        if first is None:
                            first = y
                        assert y.tobytes() == first.tobytes(), "result changed between launches"
    finally:
        set_tuning("stream_cap", 0)
        set_tuning("stream_kind", -1)
        set_tuning("stream_block", 256)
        set_tuning("pipe_wgs_per_cu", 5)


def test_csr_long_rows_are_split_and_summed(gpu, oracle):
    """Rows longer than one workgroup's stage (2048) and longer than one piece (8192)."""
    rng = np.random.default_rng(99)
    M, N = 40, 50000
    lens = np.array([3, 0, 2047, 2048, 2049, 5, 8192, 8193, 30000, 1, 0, 4096] + [7] * 28)
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(N, n, replace=False)) for n in lens]).astype(np.int32)
    val = rng.uniform(-1, 1, row_ptr[-1])
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_serial(row_ptr, col, val, x)
    from sparsematrixvectormultiplication_amd.device import set_tuning
    try:
        for cap, expect_long in ((2048, 7), (4096, 4), (8192, 3)):
            set_tuning("stream_cap", cap)
            with sp.CsrDevice(M, N, row_ptr, col, val) as dev:
                assert dev.info()["long_rows"] == expect_long
                for vname, variant in CSR_V:
                    assert_parity(dev.spmv(x, variant), y_ref, row_ptr, col, val, x,
                                  what=f"long-{vname}-cap{cap}")
                # repeated launches reuse the partial-sum scratch
                dev.run(sp.CSR_STREAM)
                assert_parity(dev.get_y(), y_ref, row_ptr, col, val, x, what="long-rerun")
    finally:
        set_tuning("stream_cap", 0)


def test_csr_degenerate_shapes(gpu, oracle):
    # no rows
    with sp.CsrDevice(0, 5, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0)) as dev:
        for _, variant in CSR_V:
            assert dev.spmv(np.ones(5), variant).shape == (0,)
    # rows but no nonzeros: y must be overwritten with zeros, not left stale
    with sp.CsrDevice(100, 3, np.zeros(101, np.int32), np.zeros(0, np.int32), np.zeros(0)) as dev:
        for _, variant in CSR_V:
            dev.set_x(np.ones(3))
            sp.lib().spmv_hip_memset(dev.y_ptr, 0xFF, 100 * 8)
            dev.run(variant)
            assert np.array_equal(dev.get_y(), np.zeros(100))
    # one dense row
    rng = np.random.default_rng(1)
    N = 3000
    row_ptr = np.array([0, N], np.int32)
    col = np.arange(N, dtype=np.int32)
    val = rng.uniform(-1, 1, N)
    x = rng.uniform(-1, 1, N)
    with sp.CsrDevice(1, N, row_ptr, col, val) as dev:
        for vname, variant in CSR_V:
            assert_parity(dev.spmv(x, variant), oracle.csr_serial(row_ptr, col, val, x), row_ptr,
                          col, val, x, what=f"dense-row-{vname}")


def test_hll_matches_oracle_seeded(gpu, oracle):
    rng = np.random.default_rng(21)
    for M, N, mean, mx, empty in [(95, 120, 4, 9, 0.2), (640, 640, 30, 70, 0.0),
                                  (70, 6000, 100, 1500, 0.0),   # hacks larger than the LDS stage
                                  (33, 20000, 5, 6000, 0.0)]:   # rows larger than the LDS stage
        row_ptr, col, val = random_csr(rng, M, N, mean, mx, empty)
        if mx >= 6000:  # force one very long row
            lens = np.diff(row_ptr)
            lens[17] = 6000
            row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            col = np.concatenate([np.sort(rng.choice(N, n, replace=False)) for n in lens]).astype(np.int32)
            val = rng.uniform(-1, 1, row_ptr[-1])
        r, c, v = coo_from_csr(row_ptr, col, val, rng)
        pre = sp.PreMatrix.from_arrays(M, N, r, c, v)
        hll = sp.convert_to_hll(pre)
        x = rng.uniform(-1, 1, N)
        y_ref = oracle.csr_serial(row_ptr, col, val, x)
        assert oracle.hll_serial(hll, x).tobytes() == y_ref.tobytes()  # K5 == K1 (same order)
        with sp.HllDevice(hll) as dev:
            for vname, variant in HLL_V:
                assert_parity(dev.spmv(x, variant), y_ref, row_ptr, col, val, x,
                              what=f"hll-{vname}-{M}x{N}")


# ------------------------------------------------------------------ fp32
def test_csr_fp32_matches_f64_accumulated_oracle(gpu, oracle):
    """Config 5's dtype: no reference counterpart exists (the reference is fp64 only);
    checked norm-wise against K1's algorithm on the fp32-rounded data with a double
    accumulator.  Parity for this dtype is therefore pinned by this repo's oracle only."""
    rng = np.random.default_rng(5)
    M = N = 4000
    row_ptr, col, val = random_csr(rng, M, N, 16, 400, 0.02, dtype=np.float32)
    x = rng.uniform(-1, 1, N).astype(np.float32)
    y_ref = oracle.csr_f32_accum64(row_ptr, col, val, x)
    with sp.CsrDevice(M, N, row_ptr, col, val) as dev:
        assert dev.info()["value_bytes"] == 4
        for vname, variant in CSR_V:
            y = dev.spmv(x, variant).astype(np.float64)
            err = np.max(np.abs(y - y_ref)) / np.max(np.abs(y_ref))
            assert err <= FP32_NORMWISE_RTOL, f"fp32 {vname}: {err:.3e}"


# ----------------------------------------------------- API / protocol
def test_row_blocks_reassemble_the_full_product(gpu, oracle):
    """Multi-GPU building block on one GPU: each row block writes its own slice of y."""
    rng = np.random.default_rng(8)
    M, N = 2500, 2500
    row_ptr, col, val = random_csr(rng, M, N, 20, 50, 0.05)
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_serial(row_ptr, col, val, x)
    bounds = sp.partition_rows(row_ptr, 4)
    y = np.zeros(M)
    for p in range(4):
        with sp.CsrDevice(M, N, row_ptr, col, val, int(bounds[p]), int(bounds[p + 1])) as dev:
            info = dev.info()
            assert (info["row0"], info["M_local"]) == (bounds[p], bounds[p + 1] - bounds[p])
            part = dev.spmv(x, sp.CSR_STREAM)
            lo, hi = bounds[p], bounds[p + 1]
            assert np.all(part[:lo] == 0) and np.all(part[hi:] == 0)  # only its own rows
            y[lo:hi] = part[lo:hi]
    assert_parity(y, y_ref, row_ptr, col, val, x, what="row blocks")


def test_run_on_caller_buffers_and_timing(gpu, oracle):
    import ctypes as C
    rng = np.random.default_rng(13)
    M, N = 3000, 3100
    row_ptr, col, val = random_csr(rng, M, N, 25, 60, 0.0)
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_serial(row_ptr, col, val, x)
    lib = sp.lib()
    dx, dy = C.c_void_p(), C.c_void_p()
    assert lib.spmv_hip_malloc(C.byref(dx), N * 8) == 0 and lib.spmv_hip_malloc(C.byref(dy), M * 8) == 0
    assert lib.spmv_hip_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), N * 8) == 0
    with sp.CsrDevice(M, N, row_ptr, col, val) as dev:
        dev.run_on(dx.value, dy.value, sp.CSR_STREAM)
        y = np.empty(M)
        assert lib.spmv_hip_memcpy_d2h(y.ctypes.data_as(C.c_void_p), dy, M * 8) == 0
        assert_parity(y, y_ref, row_ptr, col, val, x, what="run_on")
        dev.set_x(x)
        ms = dev.time(sp.CSR_STREAM, warmup=2, iters=6)
        assert ms.shape == (6,) and np.all(ms > 0) and np.all(ms < 1e3)
        assert_parity(dev.get_y(), y_ref, row_ptr, col, val, x, what="after time()")
    lib.spmv_hip_free(dx)
    lib.spmv_hip_free(dy)
    sp.flush_cache(64 << 20)


def test_single_rank_communicator(gpu):
    """RCCL path with world size 1 (all a one-GPU box can exercise): id, init, in-place
    all-gatherv leaves y untouched, destroy."""
    from sparsematrixvectormultiplication_amd.distributed import NativeComm
    comm = NativeComm(0, 1, lambda ident: ident)
    rng = np.random.default_rng(3)
    row_ptr, col, val = random_csr(rng, 500, 500, 9, 20, 0.0)
    x = rng.uniform(-1, 1, 500)
    with sp.CsrDevice(500, 500, row_ptr, col, val) as dev:
        y = dev.spmv(x)
        comm.allgatherv(dev.y_ptr, np.array([0, 500], np.int32), 8)
        assert np.array_equal(dev.get_y(), y)
        # both implementations of the all-gatherv, timed and compared on this "node" of one rank
        from sparsematrixvectormultiplication_amd.device import set_tuning
        mode, ms_b, ms_g = comm.autotune(dev.y_ptr, np.array([0, 500], np.int32), 8, 3)
        assert mode in (0, 1) and ms_b > 0 and ms_g > 0       # ms_g < 0 would mean: results differed
        assert np.array_equal(dev.get_y(), y)
        for forced in (1, 0):
            set_tuning("gather_mode", forced)
            comm.allgatherv(dev.y_ptr, np.array([0, 500], np.int32), 8)
            assert np.array_equal(dev.get_y(), y)
    comm.close()


@pytest.mark.parametrize("vb", [8, 4])
def test_padded_allgather_scatter_places_every_slice(gpu, vb):
    """The scatter kernel behind all-gatherv mode 1, without a communicator: a staging buffer laid
    out as RCCL's all-gather leaves it (slice p at p * widest) goes to the right rows of y for
    unequal, empty and single-row slices; the skipped rank's rows stay as they were."""
    import ctypes as C
    L = sp.lib()
    dt = np.float64 if vb == 8 else np.float32
    bounds = np.array([0, 700, 700, 701, 1500, 4096, 4100], np.int32)   # 6 "ranks"
    ranks, M = len(bounds) - 1, int(bounds[-1])
    widest = int(np.max(np.diff(bounds)))
    rng = np.random.default_rng(vb)
    stage = rng.uniform(-1, 1, ranks * widest).astype(dt)
    y0 = rng.uniform(-1, 1, M).astype(dt)
    d_stage, d_y = C.c_void_p(), C.c_void_p()
    assert L.spmv_hip_malloc(C.byref(d_stage), stage.nbytes) == 0 and L.spmv_hip_malloc(C.byref(d_y), y0.nbytes) == 0
    try:
        for skip in (-1, 0, 3, 5):
            assert L.spmv_hip_memcpy_h2d(d_stage, stage.ctypes.data_as(C.c_void_p), stage.nbytes) == 0
            assert L.spmv_hip_memcpy_h2d(d_y, y0.ctypes.data_as(C.c_void_p), y0.nbytes) == 0
            assert L.spmv_hip_comm_scatter_staged(d_stage, d_y, bounds.ctypes.data_as(sp._native.c_int_p), ranks,
                                                  skip, vb, None) == 0
            sp.hip_sync()
            y = np.empty(M, dt)
            assert L.spmv_hip_memcpy_d2h(y.ctypes.data_as(C.c_void_p), d_y, y.nbytes) == 0
            want = y0.copy()
            for p in range(ranks):
                if p != skip:
                    want[bounds[p]:bounds[p + 1]] = stage[p * widest:p * widest + bounds[p + 1] - bounds[p]]
            assert y.tobytes() == want.tobytes(), f"skip={skip}"
    finally:
        L.spmv_hip_free(d_stage)
        L.spmv_hip_free(d_y)


# ------------------------------------------ full-size, size-independent
def test_full_size_properties_nlpkkt_like(gpu, oracle):
    """BASELINE config 4's matrix shape at full size (3.5 M rows, ~98 M nnz): kernels agree
    with each other and with the oracle on a row sample; linearity A(ax+by) = aAx + bAy."""
    from sparsematrixvectormultiplication_amd import synth
    M, row_ptr, col, val = synth.kkt_like()
    rng = np.random.default_rng(4)
    x1, x2 = rng.uniform(-1, 1, M), rng.uniform(-1, 1, M)
    with sp.CsrDevice(M, M, row_ptr, col, val) as dev:
        y1 = dev.spmv(x1, sp.CSR_STREAM)
        y2 = dev.spmv(x2, sp.CSR_STREAM)
        y12 = dev.spmv(2.0 * x1 - 3.0 * x2, sp.CSR_STREAM)
        scale = np.max(np.abs(y1)) + np.max(np.abs(y2))
        assert np.max(np.abs(y12 - (2.0 * y1 - 3.0 * y2))) <= 1e-12 * scale * 8
        for variant in (sp.CSR_SUBWAVE, sp.CSR_WAVE_ROW):
            assert np.max(np.abs(dev.spmv(x1, variant) - y1)) <= 1e-12 * scale
        # symmetric matrix: x2 . (A x1) == x1 . (A x2)
        assert abs(np.dot(x2, y1) - np.dot(x1, y2)) <= 1e-9 * (np.linalg.norm(x1) * np.linalg.norm(y2))
        # oracle on a contiguous sample of rows from three places
        for lo in (0, M // 2 - 5000, M - 10000):
            hi = lo + 10000
            e0, e1 = row_ptr[lo], row_ptr[hi]
            rp = (row_ptr[lo:hi + 1] - e0).astype(np.int32)
            ref = oracle.csr_serial(rp, col[e0:e1], val[e0:e1], x1)
            assert_parity(y1[lo:hi], ref, rp, col[e0:e1], val[e0:e1], x1, what=f"rows {lo}..{hi}")


def test_powerlaw_fp32_with_many_long_rows(gpu, oracle):
    """BASELINE config 5's matrix family at 1/64 size (2^18 rows, ~4 M nnz, fp32): a large
    share of the entries sits in rows far longer than the LDS stage, so this exercises the
    piece / finish kernels at scale.  No reference counterpart (fp32): norm-wise 1e-5 against
    the fp64-accumulated oracle."""
    from sparsematrixvectormultiplication_amd import synth
    n, row_ptr, col, val = synth.powerlaw(1 << 18, 1 << 16, 5)
    assert val.dtype == np.float32 and np.diff(row_ptr).max() > 8192
    rng = np.random.default_rng(6)
    x = rng.uniform(-1, 1, n).astype(np.float32)
    y_ref = oracle.csr_f32_accum64(row_ptr, col, val, x)
    scale = np.max(np.abs(y_ref))
    with sp.CsrDevice(n, n, row_ptr, col, val) as dev:
        info = dev.info()
        assert info["long_rows"] > 0
        for vname, variant in (("stream", sp.CSR_STREAM), ("subwave", sp.CSR_SUBWAVE)):
            y = dev.spmv(x, variant).astype(np.float64)
            assert np.max(np.abs(y - y_ref)) <= FP32_NORMWISE_RTOL * scale, vname
        again = dev.spmv(x, sp.CSR_STREAM)
        assert again.tobytes() == dev.spmv(x, sp.CSR_STREAM).tobytes()  # no atomics: reproducible


def test_full_size_hll_equals_csr_fem_like(gpu, oracle):
    """BASELINE config 3's kernel above the Infinity Cache: the FEM-shaped generator at
    1.23 M rows / 96 M nnz built into HLL by the C host layer; the LDS kernel must agree with
    the CSR kernel (and with the oracle on a row sample), for x = 1 and a random x."""
    from sparsematrixvectormultiplication_amd import synth
    M, row_ptr, col, val = synth.fem_like((40, 40, 257), 1)
    rows = np.repeat(np.arange(M, dtype=np.int32), np.diff(row_ptr))
    hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, M, rows, col, val))
    assert hll.num_blocks == (M + 31) // 32
    rng = np.random.default_rng(12)
    with sp.HllDevice(hll) as hdev, sp.CsrDevice(M, M, row_ptr, col, val) as cdev:
        for x in (np.ones(M), rng.uniform(-1, 1, M)):
            y_csr = cdev.spmv(x, sp.CSR_STREAM)
            for variant in (sp.HLL_LDS, sp.HLL_SUBWAVE):
                y_hll = hdev.spmv(x, variant)
                scale = np.max(np.abs(y_csr))
                assert np.max(np.abs(y_hll - y_csr)) <= 1e-12 * scale
            lo, hi = M // 3, M // 3 + 5000
            e0, e1 = row_ptr[lo], row_ptr[hi]
            rp = (row_ptr[lo:hi + 1] - e0).astype(np.int32)
            ref = oracle.csr_serial(rp, col[e0:e1], val[e0:e1], x)
            assert_parity(y_hll[lo:hi], ref, rp, col[e0:e1], val[e0:e1], x, what="hll rows sample")
        # N1 at full size: the slab built on the GPU from the resident CSR is the host builder's
        with sp.HllDevice.from_csr_device(cdev) as built:
            assert built.info()["slots"] == hll.slots
            a, b = built.download(), hdev.download()
            for got, want in zip(a, b):
                assert got.tobytes() == want.tobytes()
            assert built.spmv(x, sp.HLL_LDS).tobytes() == hdev.spmv(x, sp.HLL_LDS).tobytes()


# ------------------------------------------- HLL built on the device (N1)
def _host_slab(hll):
    """The flat slab spmv_hip_hll_upload packs: hacks back to back, each on an even slot."""
    off, ja, as_ = [0], [], []
    for b in range(hll.num_blocks):
        rows, mz, j, a = hll.block(b)
        s = rows * mz
        pad = s & 1
        ja.append(np.concatenate([j, np.zeros(pad, np.int32)]))
        as_.append(np.concatenate([a, np.zeros(pad)]))
        off.append(off[-1] + s + pad)
    cat = lambda parts, dt: np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)
    return np.array(off, np.int64), cat(ja, np.int32), cat(as_, np.float64)


def _check_device_built_hll(pre, csr, x, y_ref, what):
    hll = sp.convert_to_hll(pre)
    off_h, ja_h, as_h = _host_slab(hll)
    with sp.CsrDevice.from_host(csr) as cdev, sp.HllDevice.from_csr_device(cdev) as hdev:
        info = hdev.info()
        assert info["hacks"] == hll.num_blocks and info["slots"] == hll.slots, what
        off, mz, ja, as_ = hdev.download()
        assert np.array_equal(mz, hll.maxnz), what
        assert np.array_equal(off, off_h), what
        assert ja.tobytes() == ja_h.tobytes(), f"{what}: JA differs from convert_to_hll"
        assert as_.tobytes() == as_h.tobytes(), f"{what}: AS differs from convert_to_hll"
        with sp.HllDevice(hll) as href:     # same slab -> same bits out of the same kernel
            for vname, variant in HLL_V:
                y = hdev.spmv(x, variant)
                assert y.tobytes() == href.spmv(x, variant).tobytes(), f"{what}/{vname}"
                assert_parity(y, y_ref, csr.row_ptr, csr.col_idx, csr.values, x, what=f"{what}/{vname}")


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_hll_built_on_device_equals_host_builder_golden(gpu, name):
    g = load_golden(name)
    pre = sp.read_matrix_market(golden_path(name))
    csr = sp.convert_in_csr(pre)
    lens = np.diff(csr.row_ptr)
    dup = any(len(np.unique(csr.col_idx[s:s + n])) != n for s, n in zip(csr.row_ptr[:-1], lens))
    if dup:
        # a repeated column: CSR keeps the reference quicksort's tie order, HLL the stable
        # file order (src/hll_matrix.c:14-21), so only the result is comparable
        with sp.CsrDevice.from_host(csr) as cdev, sp.HllDevice.from_csr_device(cdev) as hdev:
            assert_parity(hdev.spmv(g["x_rand"]), g["y_rand"], csr.row_ptr, csr.col_idx, csr.values,
                          g["x_rand"], what=f"{name}/device-hll")
        return
    _check_device_built_hll(pre, csr, g["x_rand"], g["y_rand"], name)


def test_hll_built_on_device_seeded(gpu, oracle):
    rng = np.random.default_rng(77)
    for M, N, mean, mx, empty in [(1, 5, 2, 2, 0.0), (31, 40, 3, 7, 0.5), (32, 32, 5, 9, 0.0),
                                  (33, 64, 6, 12, 0.1), (1000, 900, 27, 90, 0.05),
                                  (70, 6000, 100, 1500, 0.0), (4097, 4097, 1, 3, 0.6)]:
        row_ptr, col, val = random_csr(rng, M, N, mean, mx, empty)
        r, c, v = coo_from_csr(row_ptr, col, val, rng)
        pre = sp.PreMatrix.from_arrays(M, N, r, c, v)
        csr = sp.convert_in_csr(pre)
        x = rng.uniform(-1, 1, N)
        _check_device_built_hll(pre, csr, x, oracle.csr_serial(row_ptr, col, val, x), f"{M}x{N}")


def test_hll_from_csr_rejects_row_blocks_and_fp32(gpu):
    rng = np.random.default_rng(5)
    row_ptr, col, val = random_csr(rng, 200, 200, 5, 9, 0.0)
    with sp.CsrDevice(200, 200, row_ptr, col, val, row0=50, row1=150) as part:
        with pytest.raises(RuntimeError, match="whole fp64"):   # not cut on hack boundaries
            sp.HllDevice.from_csr_device(part)
    with sp.CsrDevice(200, 200, row_ptr, col, val.astype(np.float32)) as f32:
        with pytest.raises(RuntimeError, match="whole fp64"):
            sp.HllDevice.from_csr_device(f32)


# --------------------------------- stream kernel with the x window in LDS
X_WINDOW_SHAPES = [(3, 40, 0.4, 0.0), (9, 60, 0.05, 0.0), (27, 200, 0.0, 0.3), (64, 300, 0.0, 0.0), (300, 900, 0.0, 0.0),
                   (1500, 1900, 0.0, 0.0)]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
# (rows of ~1500 entries are all longer than a 1024-entry stage -- split-row kernels only, no x-window blocks: that
# combination is left out rather than skipped)
@pytest.mark.parametrize("lcap,mean,band,empty,far", [(lcap, *shape) for lcap in (1024, 2048) for shape in X_WINDOW_SHAPES
                                                      if shape[0] < lcap - 3])
def test_x_window_stream_kernel_matches_oracle_and_gather_kernel(gpu, oracle, dtype, lcap, mean, band, empty, far):
    """csr_stream_local (x lines staged in LDS, 16-bit local columns) on banded matrices over
    all block regimes: many tiny rows (row cap), ~75 rows per block, a few long rows per block,
    blocks cut by the line limit, partial last blocks.  Checked against the oracle and, bit
    for bit, against csr_stream (same products, same summation order), launched repeatedly
    into a poisoned y."""
    from sparsematrixvectormultiplication_amd.device import set_tuning
    from _util import banded_csr
    rng = np.random.default_rng(4000 + mean)
    M, N = 5003, 5600
    row_ptr, col, val = banded_csr(rng, M, N, mean, band, empty, dtype=dtype, far_frac=far)
    x = rng.uniform(-1, 1, N).astype(dtype)
    y_ref = oracle.csr_serial(row_ptr, col, val, x) if dtype == np.float64 else \
        oracle.csr_f32_accum64(row_ptr, col, val, x)
    item = np.dtype(dtype).itemsize
    try:
        set_tuning("local_cap", lcap)
        with sp.CsrDevice(M, N, row_ptr, col, val) as dev:
            info = dev.info()
            assert info["local_blocks"] > 0, "banded matrix should get an x-window plan"
            assert 32 <= info["local_stage_lines"] <= 256 and info["local_stage_lines"] % 32 == 0
            assert info["stream_bytes"] < info["algo_bytes"]
            dev.set_x(x)
            set_tuning("stream_kind", 0)
            dev.run(sp.CSR_STREAM)
            y_gather = dev.get_y()
            for kind in (5, -1):
                set_tuning("stream_kind", kind)
                for rep in range(3):
                    sp.lib().spmv_hip_memset(dev.y_ptr, 0xFF, M * item)
                    dev.run(sp.CSR_STREAM)
                    y = dev.get_y()
                    assert y.tobytes() == y_gather.tobytes(), f"kind={kind} rep={rep} differs from csr_stream"
            if dtype == np.float64:
                assert_parity(y, y_ref, row_ptr, col, val, x, what=f"x-window mean={mean}")
            else:
                err = np.max(np.abs(y.astype(np.float64) - y_ref)) / np.max(np.abs(y_ref))
                assert err <= FP32_NORMWISE_RTOL
    finally:
        set_tuning("local_cap", 0)
        set_tuning("stream_kind", -1)


def test_x_window_plan_is_refused_for_scattered_columns(gpu, oracle):
    """Uniformly random columns: a block of 2048 entries touches far more than 256 lines, the
    plan is not built and AUTO stays on the gather kernel; a row that alone needs more lines
    than a block may list does the same."""
    rng = np.random.default_rng(8)
    row_ptr, col, val = random_csr(rng, 3000, 40000, 30, 60, 0.0)
    x = rng.uniform(-1, 1, 40000)
    with sp.CsrDevice(3000, 40000, row_ptr, col, val) as dev:
        assert dev.info()["local_blocks"] == 0 and dev.info()["stream_bytes"] == 0
        assert_parity(dev.spmv(x, sp.CSR_STREAM), oracle.csr_serial(row_ptr, col, val, x), row_ptr, col, val, x)
    lens = np.full(50, 20)
    lens[7] = 1000   # 1000 columns spread over 50 000: ~1000 lines in one row
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(50000, n, replace=False)) if n == 1000 else
                          np.arange(n) + 10 for n in lens]).astype(np.int32)
    val = rng.uniform(-1, 1, row_ptr[-1])
    x = rng.uniform(-1, 1, 50000)
    with sp.CsrDevice(50, 50000, row_ptr, col, val) as dev:
        assert dev.info()["local_blocks"] == 0
        assert_parity(dev.spmv(x, sp.CSR_STREAM), oracle.csr_serial(row_ptr, col, val, x), row_ptr, col, val, x)


def test_x_window_plan_keeps_going_past_a_few_scattered_rows(gpu, oracle):
    """A banded matrix with a handful of rows whose columns are spread over all of x (each
    needs more lines than a block may list): those rows are handed to the split-row kernels,
    the rest keeps the x-window plan; fp64 and fp32.  (From 2^20 entries on: a smaller matrix keeps
    one launch of its gather kernel instead of paying two more for the split rows.)"""
    from _util import banded_csr
    rng = np.random.default_rng(57)
    M, N = 36000, 40000
    for dtype in (np.float64, np.float32):
        row_ptr, col, val = banded_csr(rng, M, N, 30, 200, 0.02, dtype=dtype)
        lens = np.diff(row_ptr).astype(np.int64)
        wild = [0, 777, 778, 3000, M - 1]
        cols, vals = [], []
        for r in range(M):
            if r in wild:
                n = 600 if r != 778 else 300
                cols.append(np.sort(rng.choice(N, n, replace=False)).astype(np.int32))
                vals.append(rng.uniform(-1, 1, n).astype(dtype))
                lens[r] = n
            else:
                cols.append(col[row_ptr[r]:row_ptr[r + 1]]); vals.append(val[row_ptr[r]:row_ptr[r + 1]])
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        c2, v2 = np.concatenate(cols).astype(np.int32), np.concatenate(vals).astype(dtype)
        x = rng.uniform(-1, 1, N).astype(dtype)
        assert rp[-1] >= 1 << 20
        with sp.CsrDevice(M, N, rp, c2, v2) as dev:
            info = dev.info()
            assert info["local_blocks"] > 0 and info["long_rows"] == len(wild)
            for vname, variant in CSR_V:
                y = dev.spmv(x, variant)
                if dtype == np.float64:
                    assert_parity(y, oracle.csr_serial(rp, c2, v2, x), rp, c2, v2, x, what=f"scattered rows {vname}")
                else:
                    ref = oracle.csr_f32_accum64(rp, c2, v2, x)
                    assert np.max(np.abs(y.astype(np.float64) - ref)) / np.max(np.abs(ref)) <= FP32_NORMWISE_RTOL


def test_auto_picks_lane_groups_for_mid_size_scattered_matrices_in_both_formats(gpu, oracle):
    """A mid-size matrix with scattered columns gets neither plan; AUTO then resolves to the lane-group kernel for CSR
    (round 2) and for HLL (round 3: hll_lds was 12-27 % slower on every such stand-in of the reference's list) --
    unless the rows are skewed, where lane groups of a fixed width lose."""
    from _util import coo_from_csr
    rng = np.random.default_rng(91)
    M = N = 40_000
    lens0 = np.minimum(rng.poisson(9, M), 14).astype(np.int64)
    rows0 = np.repeat(np.arange(M), lens0)
    cols0 = rng.integers(0, N, int(lens0.sum()))
    order0 = np.lexsort((cols0, rows0))
    rp = np.concatenate([[0], np.cumsum(lens0)]).astype(np.int32)
    col, val = cols0[order0].astype(np.int32), rng.uniform(-1, 1, int(lens0.sum()))
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_serial(rp, col, val, x)
    with sp.CsrDevice(M, N, rp, col, val) as dev:
        info = dev.info()
        assert info["local_blocks"] == 0 and info["tile_blocks"] == 0 and info["auto_variant"] == sp.CSR_SUBWAVE
        assert_parity(dev.spmv(x, sp.CSR_AUTO), y_ref, rp, col, val, x, what="csr auto = subwave")
        with sp.HllDevice.from_csr_device(dev) as h:
            hi = h.info()
            assert hi["local_blocks"] == 0 and hi["auto_variant"] == sp.HLL_SUBWAVE
            assert_parity(h.spmv(x, sp.HLL_AUTO), y_ref, rp, col, val, x, what="hll auto = subwave")
    # one hack with rows twenty times the mean: hll_lds keeps the slab
    lens = np.diff(rp).astype(np.int64)
    lens[5000:5032] = 400
    rows = np.repeat(np.arange(M), lens)
    cols2 = rng.integers(0, N, int(lens.sum()))
    order = np.lexsort((cols2, rows))
    rp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    c2, v2 = cols2[order].astype(np.int32), rng.uniform(-1, 1, int(lens.sum()))
    r, c, v = coo_from_csr(rp2, c2, v2)
    with sp.HllDevice(sp.convert_to_hll(sp.PreMatrix.from_arrays(M, N, r, c, v))) as h:
        assert h.info()["auto_variant"] == sp.HLL_LDS
        assert_parity(h.spmv(x, sp.HLL_AUTO), oracle.csr_serial(rp2, c2, v2, x), rp2, c2, v2, x, what="hll auto = lds (skewed hack)")


def test_small_matrix_with_scattered_rows_keeps_one_launch(gpu, oracle):
    """The same shape below 2^20 entries: no x-window plan with split rows (two more launches would cost more than the
    whole product); the gather kernel takes everything, and the blocks around the long rows hold few rows."""
    from _util import banded_csr
    rng = np.random.default_rng(58)
    M, N = 3000, 40000
    row_ptr, col, val = banded_csr(rng, M, N, 30, 200, 0.02)
    lens = np.diff(row_ptr).astype(np.int64)
    cols, vals = [], []
    for r in range(M):
        if r in (3, 1500, M - 2):
            cols.append(np.sort(rng.choice(N, 600, replace=False)).astype(np.int32))
            vals.append(rng.uniform(-1, 1, 600))
            lens[r] = 600
        else:
            cols.append(col[row_ptr[r]:row_ptr[r + 1]]); vals.append(val[row_ptr[r]:row_ptr[r + 1]])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    c2, v2 = np.concatenate(cols).astype(np.int32), np.concatenate(vals)
    x = rng.uniform(-1, 1, N)
    with sp.CsrDevice(M, N, rp, c2, v2) as dev:
        info = dev.info()
        assert info["local_blocks"] == 0 and info["long_rows"] == 0
        for vname, variant in CSR_V:
            assert_parity(dev.spmv(x, variant), oracle.csr_serial(rp, c2, v2, x), rp, c2, v2, x, what=f"small scattered {vname}")


def test_x_window_kernel_with_long_rows_row_blocks_and_foreign_x(gpu, oracle):
    """Long rows go to the split-row kernels beside the x-window blocks; a row block keeps
    global columns; run_on with a 128-byte aligned x uses the x-window kernel, a misaligned x
    silently takes the gather kernel (whole-line reads need the alignment)."""
    from _util import banded_csr
    rng = np.random.default_rng(31)
    M, N = 3000, 3000
    row_ptr, col, val = banded_csr(rng, M, N, 40, 150)
    lens = np.diff(row_ptr).astype(np.int64)
    lens[1000] = 2500                     # longer than the 2048-entry stage
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols, vals = [], []
    for r in range(M):
        if r == 1000:
            c = np.sort(rng.choice(N, 2500, replace=False)).astype(np.int32)
            cols.append(c); vals.append(rng.uniform(-1, 1, 2500))
        else:
            cols.append(col[row_ptr[r]:row_ptr[r + 1]]); vals.append(val[row_ptr[r]:row_ptr[r + 1]])
    col, val, row_ptr = np.concatenate(cols).astype(np.int32), np.concatenate(vals), rp
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_serial(row_ptr, col, val, x)
    with sp.CsrDevice(M, N, row_ptr, col, val) as dev:
        assert dev.info()["local_blocks"] > 0 and dev.info()["long_rows"] == 1
        assert_parity(dev.spmv(x, sp.CSR_STREAM), y_ref, row_ptr, col, val, x, what="x-window + long row")
        # foreign x / y buffers: aligned, then shifted by one element
        L = sp.lib()
        import ctypes as C
        buf = C.c_void_p()
        assert L.spmv_hip_malloc(C.byref(buf), (N + 32) * 8) == 0
        ybuf = C.c_void_p()
        assert L.spmv_hip_malloc(C.byref(ybuf), M * 8) == 0
        try:
            for shift in (0, 8):
                xp = C.c_void_p(buf.value + shift)
                assert L.spmv_hip_memcpy_h2d(xp, x.ctypes.data_as(C.c_void_p), N * 8) == 0
                assert L.spmv_hip_memset(ybuf, 0xFF, M * 8) == 0
                assert L.spmv_hip_csr_run_on(dev.h, sp.CSR_STREAM, xp, ybuf, None) == 0
                y = np.empty(M)
                assert L.spmv_hip_memcpy_d2h(y.ctypes.data_as(C.c_void_p), ybuf, M * 8) == 0
                assert_parity(y, y_ref, row_ptr, col, val, x, what=f"run_on shift={shift}")
        finally:
            L.spmv_hip_free(buf)
            L.spmv_hip_free(ybuf)
    with sp.CsrDevice(M, N, row_ptr, col, val, row0=700, row1=2100) as part:
        assert part.info()["local_blocks"] > 0
        y = part.spmv(x, sp.CSR_STREAM)
        assert_parity(y[700:2100], y_ref[700:2100], row_ptr[700:2101] - row_ptr[700],
                      col[row_ptr[700]:row_ptr[2100]], val[row_ptr[700]:row_ptr[2100]], x, what="row block")


@pytest.mark.parametrize("mean,band,empty,far", [(3, 40, 0.4, 0.0), (27, 200, 0.0, 0.3), (64, 300, 0.02, 0.0),
                                                 (300, 900, 0.0, 0.0)])
def test_hll_x_window_kernel_matches_oracle(gpu, oracle, mean, band, empty, far):
    """hll_lds_local (x lines staged in LDS, 16-bit local JA) on banded matrices: windows cut
    by slots, by the line limit and by the row cap; padding slots; M not a multiple of 32."""
    from sparsematrixvectormultiplication_amd.device import set_tuning
    from _util import banded_csr
    rng = np.random.default_rng(6000 + mean)
    M, N = 4099, 4500
    row_ptr, col, val = banded_csr(rng, M, N, mean, band, empty, far_frac=far)
    r, c, v = coo_from_csr(row_ptr, col, val, rng)
    hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, N, r, c, v))
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_serial(row_ptr, col, val, x)
    try:
        with sp.HllDevice(hll) as dev:
            info = dev.info()
            assert info["local_blocks"] > 0 and 0 < info["stream_bytes"] < info["algo_bytes"]
            dev.set_x(x)
            set_tuning("stream_kind", 0)
            dev.run(sp.HLL_LDS)
            y_gather = dev.get_y()
            assert_parity(y_gather, y_ref, row_ptr, col, val, x, what="hll_lds")
            set_tuning("stream_kind", -1)
            first = None
            for rep in range(3):
                sp.lib().spmv_hip_memset(dev.y_ptr, 0xFF, M * 8)
                dev.run(sp.HLL_LDS)
                y = dev.get_y()
                assert_parity(y, y_ref, row_ptr, col, val, x, what=f"hll_lds_local mean={mean} rep={rep}")
                first = y if first is None else first
                assert y.tobytes() == first.tobytes()
        # the slab built on the device gets the same plan and the same bits
        with sp.CsrDevice(M, N, row_ptr, col, val) as cdev, sp.HllDevice.from_csr_device(cdev) as built:
            assert built.info()["local_blocks"] == info["local_blocks"]
            assert built.spmv(x, sp.HLL_LDS).tobytes() == first.tobytes()
    finally:
        set_tuning("stream_kind", -1)


# ------------------------------------------- HLL hack ranges (multi-GPU shares)
def test_hll_hack_range_handles_cover_the_matrix(gpu, oracle):
    """HLL splits on hack boundaries.  Three handles holding the K8 partition's
    hack ranges each write only their rows of a full-length y; together they give the
    whole product.  The same ranges built on the device from 32-aligned CSR row blocks give
    the same bits; a row block that is not cut on a hack boundary is refused."""
    from _util import banded_csr
    rng = np.random.default_rng(91)
    for (M, N, gen) in ((1000, 1100, "banded"), (333, 2000, "random")):
        if gen == "banded":
            row_ptr, col, val = banded_csr(rng, M, N, 20, 120, 0.05)
        else:
            row_ptr, col, val = random_csr(rng, M, N, 15, 60, 0.1)
        r, c, v = coo_from_csr(row_ptr, col, val, rng)
        hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, N, r, c, v))
        x = rng.uniform(-1, 1, N)
        y_ref = oracle.csr_serial(row_ptr, col, val, x)
        hb = sp.partition_hacks(hll, 3)
        rb = sp.hack_bounds_to_rows(hb, M)
        assert hb[0] == 0 and hb[-1] == hll.num_blocks
        y_all = np.full(M, np.nan)
        for p in range(3):
            with sp.HllDevice(hll, int(hb[p]), int(hb[p + 1])) as part:
                info = part.info()
                assert (info["row0"], info["M_local"], info["M_total"]) == (rb[p], rb[p + 1] - rb[p], M)
                for vname, variant in HLL_V:
                    part.set_x(x)
                    sp.lib().spmv_hip_memset(part.y_ptr, 0xFF, M * 8)
                    part.run(variant)
                    y = part.get_y()
                    lo, hi = rb[p], rb[p + 1]
                    assert np.all(np.isnan(y[:lo])) and np.all(np.isnan(y[hi:])), f"{vname}: wrote outside its rows"
                    assert_parity(y[lo:hi], y_ref[lo:hi], row_ptr[lo:hi + 1] - row_ptr[lo],
                                  col[row_ptr[lo]:row_ptr[hi]], val[row_ptr[lo]:row_ptr[hi]], x,
                                  what=f"hack range {p} {vname}")
                y_all[lo:hi] = y[lo:hi]
                # same range from the resident CSR row block
                if hi > lo:
                    with sp.CsrDevice(M, N, row_ptr, col, val, row0=int(lo), row1=int(hi)) as cpart, \
                            sp.HllDevice.from_csr_device(cpart) as built:
                        assert built.info()["row0"] == lo and built.info()["slots"] == info["slots"]
                        built.set_x(x)
                        built.run(sp.HLL_LDS)
                        part.run(sp.HLL_LDS)
                        assert built.get_y()[lo:hi].tobytes() == part.get_y()[lo:hi].tobytes()
        assert not np.any(np.isnan(y_all))
    with sp.CsrDevice(M, N, row_ptr, col, val, row0=5, row1=200) as odd:
        with pytest.raises(RuntimeError, match="hack boundaries"):
            sp.HllDevice.from_csr_device(odd)


def test_graph_replayed_timing_leaves_the_result_intact(gpu, oracle):
    """spmv_hip_*_time_graph: a captured batch of launches replayed from a hipGraph; y after the
    replays is the same SpMV result, the per-SpMV time is positive and not above the
    event-timed launch loop by much (it removes host work, it cannot add kernel time)."""
    from _util import banded_csr
    rng = np.random.default_rng(123)
    M = N = 20000
    row_ptr, col, val = banded_csr(rng, M, N, 40, 200)
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_serial(row_ptr, col, val, x)
    with sp.CsrDevice(M, N, row_ptr, col, val) as dev:
        dev.set_x(x)
        t_graph = dev.time_graph(sp.CSR_STREAM, 10, 5)
        assert_parity(dev.get_y(), y_ref, row_ptr, col, val, x, what="after graph replay")
        t_loop = float(dev.time(sp.CSR_STREAM, 3, 20, zero_y=False).mean())
        assert 0 < t_graph < 5 * t_loop + 0.05
        with sp.HllDevice.from_csr_device(dev) as h:
            h.set_x(x)
            assert h.time_graph(sp.HLL_LDS, 10, 5) > 0
            assert_parity(h.get_y(), y_ref, row_ptr, col, val, x, what="hll after graph replay")


# ------------------------------------------------ iterated SpMV (N4)
def test_power_iteration_matches_the_oracle_loop(gpu, oracle):
    """spmv_hip_csr_power_iterate: x <- A x / ||A x||_2 repeated on the device (y fed back into x,
    norm by a fixed-order device reduction).  Against the same loop done with the oracle's serial
    kernel and numpy's norm; graph-captured and plain launches give the same bits; fp32 too."""
    from _util import banded_csr
    rng = np.random.default_rng(2024)
    n = 3000
    row_ptr, col, val = banded_csr(rng, n, n, 12, 40)
    x0 = rng.uniform(0.5, 1.0, n)
    iters = 6
    x_ref = x0.copy()
    for _ in range(iters):
        y_ref = oracle.csr_serial(row_ptr, col, val, x_ref)
        lam_ref = np.linalg.norm(y_ref)
        x_ref = y_ref / lam_ref
    with sp.CsrDevice(n, n, row_ptr, col, val) as dev:
        results = []
        for graph in (True, False):
            dev.set_x(x0)
            lam, ms = dev.power_iterate(iters, sp.CSR_STREAM, use_graph=graph)
            x, y = dev.get_x(), dev.get_y()
            assert ms > 0 and abs(lam - lam_ref) <= 1e-11 * lam_ref
            assert np.max(np.abs(x - x_ref)) <= 1e-10 * np.max(np.abs(x_ref))
            assert np.max(np.abs(y - y_ref)) <= 1e-10 * np.max(np.abs(y_ref))
            assert abs(np.linalg.norm(x) - 1.0) <= 1e-13
            results.append((lam, x.tobytes(), y.tobytes()))
        assert results[0] == results[1], "graph replay and plain launches differ"
        for variant in (sp.CSR_SUBWAVE, sp.CSR_WAVE_ROW):     # any kernel can drive the loop
            dev.set_x(x0)
            lam, _ = dev.power_iterate(iters, variant)
            assert abs(lam - lam_ref) <= 1e-11 * lam_ref
    with sp.CsrDevice(n, n, row_ptr, col, val.astype(np.float32)) as dev32:
        dev32.set_x(x0.astype(np.float32))
        lam32, _ = dev32.power_iterate(iters)
        assert abs(lam32 - lam_ref) <= 1e-4 * lam_ref
    rp2, c2, v2 = random_csr(rng, 50, 60, 4, 8, 0.0)
    with sp.CsrDevice(50, 60, rp2, c2, v2) as rect:
        with pytest.raises(RuntimeError, match="square"):
            rect.power_iterate(2)


def spd_banded(rng, n, per_row, band):
    """symmetric, strictly diagonally dominant (hence positive definite) banded matrix as CSR"""
    import scipy.sparse as sps
    r = np.repeat(np.arange(n), per_row)
    c = np.clip(r + rng.integers(-band, band + 1, len(r)), 0, n - 1)
    b = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))
    a = b + b.T
    a = a + sps.diags(np.asarray(abs(a).sum(axis=1)).ravel() + 1.0)
    a = a.tocsr()
    a.sum_duplicates()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32), np.ascontiguousarray(a.data)


def cg_with(spmv, b, iters):
    """the textbook loop spmv_hip_csr_cg runs, with a given product; returns (x, r.r history)"""
    x = np.zeros_like(b)
    r = b.copy()
    p = b.copy()
    rs = float(r @ r)
    hist = [rs]
    for _ in range(iters):
        q = spmv(p)
        alpha = rs / float(p @ q)
        x += alpha * p
        r -= alpha * q
        rs_new = float(r @ r)
        p = r + (rs_new / rs) * p
        rs = rs_new
        hist.append(rs)
    return x, np.array(hist)


def test_conjugate_gradients_match_the_oracle_loop(gpu, oracle):
    """spmv_hip_csr_cg (N4's second skeleton): plain CG from x0 = 0 on the device -- p is the handle's x, q = A p its
    y, fixed-order dot products, scalars never leave the device -- against the same loop with the oracle's serial
    kernel; every kernel variant can drive it; bit-reproducible; with a single-rank communicator the all-gatherv
    and the halo exchange give the plain loop's bits; fp32 within its own gate."""
    from sparsematrixvectormultiplication_amd.distributed import NativeComm
    rng = np.random.default_rng(808)
    n = 6000
    row_ptr, col, val = spd_banded(rng, n, 7, 60)
    x_true = rng.uniform(-1, 1, n)
    b = oracle.csr_serial(row_ptr, col, val, x_true)
    iters = 25
    x_ref, hist_ref = cg_with(lambda v: oracle.csr_serial(row_ptr, col, val, v), b, iters)
    assert hist_ref[-1] < 1e-6 * hist_ref[0]          # the reference loop itself converges on this matrix
    assert np.max(np.abs(x_ref - x_true)) <= 1e-3 * np.max(np.abs(x_true))
    with sp.CsrDevice(n, n, row_ptr, col, val) as dev:
        # (dot products are summed in another order than numpy's: the difference is rounding, and a Krylov recurrence
        # carries it along -- 1e-10 after a few steps, 1e-8 after 25)
        x5, hist5, _ = dev.cg(b, 5)
        x_ref5, hist_ref5 = cg_with(lambda v: oracle.csr_serial(row_ptr, col, val, v), b, 5)
        assert np.max(np.abs(x5 - x_ref5)) <= 1e-10 * np.max(np.abs(x_ref5))
        assert np.all(np.abs(hist5 - hist_ref5) <= 1e-10 * hist_ref5[0])
        x, hist, ms = dev.cg(b, iters)
        assert ms > 0 and hist[0] == pytest.approx(hist_ref[0], rel=1e-13)
        assert np.max(np.abs(x - x_ref)) <= 1e-7 * np.max(np.abs(x_ref))
        assert np.all(np.abs(hist - hist_ref) <= 1e-8 * hist_ref[0] + 1e-4 * hist_ref)
        assert np.max(np.abs(x - x_true)) <= 1e-3 * np.max(np.abs(x_true))   # and towards the solution
        resid = b - oracle.csr_serial(row_ptr, col, val, x)
        assert float(resid @ resid) <= 4.0 * hist[-1] + 1e-20 * hist[0]     # the recurrence's residual is the true one
        x2, hist2, _ = dev.cg(b, iters)
        assert x2.tobytes() == x.tobytes() and hist2.tobytes() == hist.tobytes()
        for variant in (sp.CSR_SUBWAVE, sp.CSR_WAVE_ROW, sp.CSR_THREAD_ROW):
            xv, _, _ = dev.cg(b, iters, variant)
            assert np.max(np.abs(xv - x_ref)) <= 1e-7 * np.max(np.abs(x_ref))
        comm = NativeComm(0, 1, lambda ident: ident)
        try:
            bounds = np.array([0, n], np.int32)
            xg, hg, _ = dev.cg(b, iters, bounds=bounds)                 # all-gatherv of p, all-gather of the dot products
            assert xg.tobytes() == x.tobytes() and hg.tobytes() == hist.tobytes()
            comm.halo_setup(dev, bounds)
            xh, hh, _ = dev.cg(b, iters, bounds=bounds, use_halo=True)
            assert xh.tobytes() == x.tobytes() and hh.tobytes() == hist.tobytes()
        finally:
            comm.close()
    with sp.CsrDevice(n, n, row_ptr, col, val.astype(np.float32)) as dev32:
        x32, hist32, _ = dev32.cg(b.astype(np.float32), 6)
        x_ref6, _ = cg_with(lambda v: oracle.csr_serial(row_ptr, col, val, v), b, 6)
        assert np.max(np.abs(x32 - x_ref6)) <= 1e-4 * np.max(np.abs(x_ref6))
    rp2, c2, v2 = random_csr(rng, 50, 60, 4, 8, 0.0)
    with sp.CsrDevice(50, 60, rp2, c2, v2) as rect:
        with pytest.raises(RuntimeError, match="square"):
            rect.cg(np.ones(50), 2)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_plan_built_on_the_device_equals_the_host_plan(gpu, oracle, dtype):
    """plan_count / plan_fill (LDS sort + unique per block) against csr_build_local / hll_build_local: the
    same blocks, the same number of listed lines, the same bits out of the kernels; a matrix whose
    blocks exceed the line limit falls back to the host builder either way."""
    from sparsematrixvectormultiplication_amd.device import set_tuning
    from _util import banded_csr
    rng = np.random.default_rng(77)
    M, N = 9001, 9300
    for mean, band, empty in ((4, 50, 0.3), (30, 180, 0.0), (120, 900, 0.0)):
        row_ptr, col, val = banded_csr(rng, M, N, mean, band, empty, dtype=dtype)
        x = rng.uniform(-1, 1, N).astype(dtype)
        got = {}
        try:
            for where in (1, 0):
                set_tuning("plan_on_device", where)
                with sp.CsrDevice(M, N, row_ptr, col, val) as dev:
                    info = dev.info()
                    y = dev.spmv(x, sp.CSR_STREAM)
                    entry = [info["local_blocks"], info["local_lines"], info["local_stage_lines"], y.tobytes()]
                    if dtype == np.float64:
                        with sp.HllDevice.from_csr_device(dev) as h:
                            hi = h.info()
                            entry += [hi["local_blocks"], hi["local_stage_lines"], h.spmv(x, sp.HLL_LDS).tobytes()]
                    got[where] = entry
        finally:
            set_tuning("plan_on_device", 1)
        assert got[1][0] > 0, "banded matrix should get a plan"
        assert got[1] == got[0], f"device-built plan differs from the host-built one (mean={mean})"
    # blocks beyond the line limit: the device pass reports it, the host builder takes over (and cuts by lines)
    row_ptr, col, val = banded_csr(rng, 3000, 40000, 30, 1500)
    x = rng.uniform(-1, 1, 40000)
    with sp.CsrDevice(3000, 40000, row_ptr, col, val) as dev:
        assert_parity(dev.spmv(x, sp.CSR_STREAM), oracle.csr_serial(row_ptr, col, val, x), row_ptr, col, val, x)


# ------------------------------------------- CSR built on the device from COO (N1)
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_csr_from_coo_on_device_golden(gpu, name):
    """spmv_hip_csr_from_coo against the host builder on the golden matrices: the same CSR arrays bit for
    bit where no (row, column) repeats; with repeats the same row_ptr / columns and the same multiset of
    values per run (the device sort is stable, the reference's quicksort is not); SpMV parity either way."""
    g = load_golden(name)
    pre = sp.read_matrix_market(golden_path(name))
    csr = sp.convert_in_csr(pre)
    with sp.CsrDevice.from_coo(pre.M, pre.N, pre.I, pre.J, pre.val) as dev:
        rp, col, val = dev.download()
        np.testing.assert_array_equal(rp, csr.row_ptr)
        np.testing.assert_array_equal(col, csr.col_idx)
        if name == "dup_entries":
            np.testing.assert_array_equal(np.sort(val), np.sort(csr.values))
        else:
            assert val.tobytes() == csr.values.tobytes()
        for x, key in ((np.ones(csr.N), "y_ones"), (g["x_rand"], "y_rand")):
            for vname, variant in CSR_V:
                assert_parity(dev.spmv(x, variant), g[key], csr.row_ptr, csr.col_idx, csr.values, x,
                              what=f"{name}/from_coo/{vname}")


def test_csr_from_coo_on_device_large_shuffled(gpu, oracle):
    """3 M shuffled triplets with empty rows, a heavy row and rows longer than the stage: CSR arrays equal to
    convert_in_csr's bit for bit, the handle gets its x-window plan, and the device-built HLL follows."""
    from _util import banded_csr
    rng = np.random.default_rng(99)
    M, N = 70001, 70500
    row_ptr, col, val = banded_csr(rng, M, N, 40, 300, 0.05)
    rows = np.repeat(np.arange(M, dtype=np.int32), np.diff(row_ptr))
    perm = rng.permutation(len(col))
    I, J, V = rows[perm], col[perm], val[perm]
    with sp.CsrDevice.from_coo(M, N, I, J, V) as dev:
        rp, c2, v2 = dev.download()
        np.testing.assert_array_equal(rp, row_ptr)
        np.testing.assert_array_equal(c2, col)
        assert v2.tobytes() == val.tobytes()
        assert dev.info()["local_blocks"] > 0
        x = rng.uniform(-1, 1, N)
        y_ref = oracle.csr_serial(row_ptr, col, val, x)
        assert_parity(dev.spmv(x), y_ref, row_ptr, col, val, x, what="from_coo large")
        with sp.HllDevice.from_csr_device(dev) as h:
            assert_parity(h.spmv(x), y_ref, row_ptr, col, val, x, what="from_coo -> hll")
    # degenerate and bad inputs
    with sp.CsrDevice.from_coo(5, 7, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0)) as empty:
        assert np.array_equal(empty.download()[0], np.zeros(6, np.int32))
        assert np.array_equal(empty.spmv(np.ones(7)), np.zeros(5))
    with pytest.raises(RuntimeError, match="outside"):
        sp.CsrDevice.from_coo(5, 7, np.array([1, 5], np.int32), np.array([0, 0], np.int32), np.ones(2))
    # scattered columns: no plan on the device, the host builder is asked (columns copied back) and refuses too
    Is = rng.integers(0, 3000, 90000).astype(np.int32)
    Js = rng.integers(0, 40000, 90000).astype(np.int32)
    Vs = rng.uniform(-1, 1, 90000)
    with sp.CsrDevice.from_coo(3000, 40000, Is, Js, Vs) as sc:
        assert sc.info()["local_blocks"] == 0
        ref = sp.convert_in_csr(sp.PreMatrix.from_arrays(3000, 40000, Is, Js, Vs))
        x = rng.uniform(-1, 1, 40000)
        assert_parity(sc.spmv(x), oracle.csr_serial(ref.row_ptr, ref.col_idx, ref.values, x), ref.row_ptr,
                      ref.col_idx, ref.values, x, what="from_coo scattered")


# ------------------------------------------------ halo exchange (N4, second half)
def test_needed_ranges_cover_what_the_rows_touch(gpu):
    from sparsematrixvectormultiplication_amd.distributed import needed_ranges
    from _util import banded_csr
    rng = np.random.default_rng(61)
    M = N = 20000
    row_ptr, col, val = banded_csr(rng, M, N, 25, 300, 0.02, far_frac=0.2)
    for r0, r1 in ((0, M), (5000, 9000), (19000, M)):
        with sp.CsrDevice(M, N, row_ptr, col, val, row0=r0, row1=r1) as dev:
            assert dev.info()["local_blocks"] > 0
            for cap in (32, 4, 1):
                rs = needed_ranges(dev, cap)
                assert 1 <= len(rs) <= cap and all(a < b for a, b in rs)
                assert all(rs[k][1] < rs[k + 1][0] for k in range(len(rs) - 1))
                touched = np.unique(col[row_ptr[r0]:row_ptr[r1]])
                inside = np.zeros(N, bool)
                for a, b in rs:
                    inside[a:b] = True
                assert inside[touched].all(), "a touched column lies outside the reported ranges"
                if cap == 32:   # and not wildly more than needed: whole lines of touched columns plus closed gaps
                    assert inside.sum() <= 16 * len(np.unique(touched // 16)) + 16 * 2000
    row_ptr, col, val = random_csr(rng, 500, 30000, 20, 40, 0.0)      # no plan: everything
    with sp.CsrDevice(500, 30000, row_ptr, col, val) as dev:
        assert needed_ranges(dev) == [(0, 30000)]


def test_power_iteration_with_halo_exchange_emulated_ranks(gpu, oracle):
    """Three 'ranks' in one process on one GPU: each holds its row block, asks the library what it needs of x
    (spmv_hip_csr_needed_ranges) and who sends what to whom (spmv_hip_halo_plan); the transport is emulated with
    host copies.  Everything a rank does not own and did not receive is NaN, so a missing halo entry poisons the
    result.  Against the single-handle power iteration."""
    import ctypes as C
    from sparsematrixvectormultiplication_amd.distributed import halo_plan, needed_ranges
    from _util import banded_csr
    rng = np.random.default_rng(62)
    n = 9000
    row_ptr, col, val = banded_csr(rng, n, n, 18, 120)
    x0 = rng.uniform(0.5, 1.0, n)
    iters = 5
    with sp.CsrDevice(n, n, row_ptr, col, val) as whole:
        whole.set_x(x0)
        lam_ref, _ = whole.power_iterate(iters)
        x_ref = whole.get_x()
    bounds = sp.partition_rows(row_ptr, 3)
    devs = [sp.CsrDevice(n, n, row_ptr, col, val, row0=int(bounds[r]), row1=int(bounds[r + 1])) for r in range(3)]
    try:
        needs = [needed_ranges(d) for d in devs]
        plans = [halo_plan(r, bounds, needs) for r in range(3)]
        assert sum(hi - lo for _, lo, hi in plans[1][1]) < n // 2, "banded matrix: the halo is a small part of x"
        xs = []
        for r in range(3):                                   # own range + halo of x0, NaN elsewhere
            x = np.full(n, np.nan)
            x[bounds[r]:bounds[r + 1]] = x0[bounds[r]:bounds[r + 1]]
            for q, lo, hi in plans[r][1]:
                x[lo:hi] = x0[lo:hi]
            xs.append(x)
        lam = 0.0
        for _ in range(iters):
            ys = []
            for r, d in enumerate(devs):
                d.set_x(xs[r])
                d.run(sp.CSR_STREAM)
                ys.append(d.get_y()[bounds[r]:bounds[r + 1]])
                assert not np.any(np.isnan(ys[-1])), "a row read an x entry it was not given"
            lam = float(np.sqrt(sum(np.sum(y * y) for y in ys)))       # partial sums + "all-reduce"
            own = [y / lam for y in ys]
            xs = []
            for r in range(3):
                x = np.full(n, np.nan)
                x[bounds[r]:bounds[r + 1]] = own[r]
                for q, lo, hi in plans[r][1]:                          # what q sends to r
                    x[lo:hi] = own[q][lo - bounds[q]:hi - bounds[q]]
                xs.append(x)
        assert abs(lam - lam_ref) <= 1e-12 * lam_ref
        for r in range(3):
            seg = slice(bounds[r], bounds[r + 1])
            assert np.max(np.abs(xs[r][seg] - x_ref[seg])) <= 1e-12 * np.max(np.abs(x_ref))
    finally:
        for d in devs:
            d.close()


def test_interior_and_boundary_blocks_emulated_ranks(gpu, oracle):
    """SURVEY 8(f) N4, overlap: a rank's x-window blocks split into interior blocks (own range of x only) and
    boundary blocks.  Three emulated ranks: the interior part runs while the halo of x is still NaN, the boundary
    part after it has 'arrived'; the two parts together must give the bits of the one-launch product, and no NaN
    may survive (an interior block that read a halo entry would leave one).  Also the shares."""
    from sparsematrixvectormultiplication_amd.distributed import halo_plan, needed_ranges
    from _util import banded_csr
    rng = np.random.default_rng(64)
    n = 30000
    row_ptr, col, val = banded_csr(rng, n, n, 22, 150)
    lens = np.diff(row_ptr)
    x0 = rng.uniform(0.5, 1.0, n)
    bounds = sp.partition_rows(row_ptr, 3)
    devs = [sp.CsrDevice(n, n, row_ptr, col, val, row0=int(bounds[r]), row1=int(bounds[r + 1])) for r in range(3)]
    try:
        needs = [needed_ranges(d) for d in devs]
        plans = [halo_plan(r, bounds, needs) for r in range(3)]
        for r, d in enumerate(devs):
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            counts = d.split_interior()
            info = d.info()
            assert counts["interior_blocks"] + counts["boundary_blocks"] == info["local_blocks"] > 0
            assert counts["interior_entries"] + counts["boundary_entries"] == info["nz"]
            assert counts["interior_blocks"] > 0.7 * info["local_blocks"]     # a band of +-150 in ~10 000 rows
            assert counts["boundary_blocks"] > 0                              # ... but the ends do reach out
            x_full = np.full(n, np.nan)
            x_full[lo:hi] = x0[lo:hi]
            for q, a, b in plans[r][1]:
                x_full[a:b] = x0[a:b]
            d.set_x(x_full)
            d.run(sp.CSR_STREAM)
            y_one = d.get_y()[lo:hi].copy()
            assert not np.any(np.isnan(y_one))
            x_own = np.full(n, np.nan)
            x_own[lo:hi] = x0[lo:hi]                                          # the halo has not arrived
            d.set_x(x_own)
            sp.lib().spmv_hip_memset(d.y_ptr, 0xFF, n * 8)
            d.run_part(0)
            y_mid = d.get_y()[lo:hi].copy()
            done = ~np.isnan(y_mid)                                           # rows of interior blocks
            assert done.sum() > 0.7 * (hi - lo) and np.array_equal(y_mid[done], y_one[done])
            d.set_x(x_full)                                                   # ... now it has
            d.run_part(1)
            y_two = d.get_y()[lo:hi]
            assert y_two.tobytes() == y_one.tobytes()
        # a handle without an x-window plan has no interior: everything runs as part 1
        rp2, c2, v2 = random_csr(rng, 2000, 2000, 30, 60, 0.0)
        with sp.CsrDevice(2000, 2000, rp2, c2, v2) as scattered:
            if not scattered.info()["local_blocks"]:
                assert scattered.split_interior()["interior_blocks"] == 0
                xs = rng.uniform(-1, 1, 2000)
                y_ref = scattered.spmv(xs, sp.CSR_STREAM)
                sp.lib().spmv_hip_memset(scattered.y_ptr, 0xFF, 2000 * 8)
                scattered.run_part(0)
                scattered.run_part(1)
                assert scattered.get_y().tobytes() == y_ref.tobytes()
    finally:
        for d in devs:
            d.close()


def test_column_split_emulated_ranks(gpu, oracle):
    """N4 overlap below block granularity (round 3): on a KKT-coupled cut every block of a rank also lists lines of the
    coupling block, which another rank owns -- no interior BLOCKS -- but about half of a row's entries have their column
    in the rank's own range.  spmv_hip_csr_split_columns splits the handle by column; part 0 must read nothing of x
    outside the own range (the halo is NaN while it runs), part 0 then part 1 = the product of the oracle within 1e-10,
    and twice the same bits."""
    from sparsematrixvectormultiplication_amd import synth
    rng = np.random.default_rng(65)
    M, row_ptr, col, val = synth.kkt_like((24, 24, 25), 5)
    x0 = rng.uniform(0.5, 1.0, M)
    ref = oracle.csr_serial(row_ptr, col, val, x0)
    bounds = sp.partition_rows(row_ptr, 4)
    shares = []
    for r in range(4):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        with sp.CsrDevice(M, M, row_ptr, col, val, row0=lo, row1=hi) as d:
            blocks = d.split_interior()
            counts = d.split_columns(lo, hi)
            sl = slice(int(row_ptr[lo]), int(row_ptr[hi]))
            inside = int(np.count_nonzero((col[sl] >= lo) & (col[sl] < hi)))
            assert counts == {"own_entries": inside, "halo_entries": (sl.stop - sl.start) - inside}
            shares.append((blocks["interior_entries"] / (sl.stop - sl.start), inside / (sl.stop - sl.start)))
            x_own = np.full(M, np.nan)
            x_own[lo:hi] = x0[lo:hi]                                          # the halo has not arrived
            d.set_x(x_own)
            sp.lib().spmv_hip_memset(d.y_ptr, 0xFF, M * 8)
            d.run_split(0)
            y_mid = d.get_y()[lo:hi].copy()
            assert not np.any(np.isnan(y_mid))                                # part 0 wrote every row and read no halo
            own_only = oracle.csr_serial(row_ptr, np.where((col >= lo) & (col < hi), col, 0).astype(np.int32),
                                         np.where((col >= lo) & (col < hi), val, 0.0), x0)[lo:hi]
            assert np.max(np.abs(y_mid - own_only) / np.maximum(np.abs(own_only), 1e-300)) <= 1e-10
            d.set_x(x0)                                                       # ... now it has
            d.run_split(1)
            y_two = d.get_y()[lo:hi].copy()
            assert np.max(np.abs(y_two - ref[lo:hi]) / np.maximum(np.abs(ref[lo:hi]), 1e-300)) <= 1e-10
            d.run_split(0)
            d.run_split(1)
            assert d.get_y()[lo:hi].tobytes() == y_two.tobytes()              # a fixed order of the two partial sums
            # the handle's one-launch product is untouched by the split
            assert np.max(np.abs(d.spmv(x0, sp.CSR_STREAM)[lo:hi] - ref[lo:hi]) / np.abs(ref[lo:hi])) <= 1e-10
    # the KKT cut: (almost) no interior blocks on the ranks that hold grid rows, but a large share of interior ENTRIES
    assert max(b for b, _ in shares) < 0.2 and min(c for _, c in shares) > 0.35, shares
    # a rank that owns everything has no halo entries: no sub-handles, run_split(0) is the whole product
    rp2, c2, v2 = random_csr(rng, 3000, 3000, 5, 40, 0.0)
    with sp.CsrDevice(3000, 3000, rp2, c2, v2) as whole:
        assert whole.split_columns(0, 3000)["halo_entries"] == 0
        xs = rng.uniform(-1, 1, 3000)
        y_ref = whole.spmv(xs, sp.CSR_AUTO)
        sp.lib().spmv_hip_memset(whole.y_ptr, 0xFF, 3000 * 8)
        whole.run_split(0)
        whole.run_split(1)
        assert whole.get_y().tobytes() == y_ref.tobytes()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pattern_plan_rebuilds_the_slots_bit_for_bit(gpu, oracle, dtype):
    """Round 3: where most rows of an x-window block are their predecessor shifted by a constant (stencils), the kernel
    does not read the 16-bit slot of every entry: upload keeps one pattern table per block and 4 bytes per row, the
    kernel rebuilds the block's slots from them in LDS.  The same slots, hence the same products in the same order: the
    bits of the kernel that reads the slot stream ("local_patterns" 0 at launch: the same handle).  Cases: the KKT-shaped
    stencil (auto: plan kept), a band with random columns (auto: not kept; forced: every row its own pattern), blocks of
    very short rows (several row passes per block), empty rows, a row block of a bigger matrix."""
    from sparsematrixvectormultiplication_amd import synth
    from sparsematrixvectormultiplication_amd.device import set_tuning
    from _util import banded_csr
    rng = np.random.default_rng(91)

    def run(M, N, rp, col, val, forced, expect_plan, row0=0, row1=None):
        x = rng.uniform(-1, 1, N).astype(dtype)
        ref = (oracle.csr_serial if dtype == np.float64 else oracle.csr_f32_accum64)(rp, col, val, x)
        set_tuning("local_patterns", 1 if forced else -1)
        try:
            with sp.CsrDevice(M, N, rp, col, val, row0=row0, row1=M if row1 is None else row1) as dev:
                info = dev.info()
                assert info["local_blocks"] > 0
                if expect_plan is not None:
                    assert (info["pattern_slots"] > 0) == expect_plan, info["pattern_slots"]
                lo, hi = row0, M if row1 is None else row1
                sp.lib().spmv_hip_memset(dev.y_ptr, 0xFF, M * x.itemsize)
                y = dev.spmv(x, sp.CSR_STREAM)[lo:hi].copy()
                tol = 1e-10 if dtype == np.float64 else 1e-5
                assert np.max(np.abs(y.astype(np.float64) - ref[lo:hi])) <= tol * max(np.max(np.abs(ref)), 1e-300)
                set_tuning("local_patterns", 0)                   # the same handle through the slot stream
                y0 = dev.spmv(x, sp.CSR_STREAM)[lo:hi].copy()
                assert y.tobytes() == y0.tobytes()
                return info
        finally:
            set_tuning("local_patterns", -1)

    M, rp, col, val = synth.kkt_like((24, 24, 25), 5)
    val = val.astype(dtype)
    info = run(M, M, rp, col, val, True, True)
    assert info["pattern_slots"] * 4 <= rp[-1]                    # a stencil: the tables hold a fraction of the slots
    run(M, M, rp, col, val, True, True, row0=M // 3, row1=2 * M // 3)
    run(M, M, rp, col, val, False, False)                         # auto: a matrix of this size lives in the cache
    if dtype == np.float64:                                       # auto: built (streamed, 28 per row), then kept or not by
        Mb, rpb, colb, valb = synth.kkt_like((64, 64, 66), 5)     # upload's own measurement -- either way the same bits
        assert rpb[-1] * 10 > (128 << 20)
        info = run(Mb, Mb, rpb, colb, valb, False, None)
        assert info["pattern_with_us"] > 0 and info["pattern_without_us"] > 0
        assert (info["pattern_slots"] > 0) == (info["pattern_with_us"] <= 0.98 * info["pattern_without_us"])
    rp2, col2, val2 = banded_csr(rng, 20000, 20000, 22, 150)       # random columns inside a band: no two rows alike
    run(20000, 20000, rp2, col2, val2.astype(dtype), False, False)
    run(20000, 20000, rp2, col2, val2.astype(dtype), True, True)
    # very short rows (hundreds of rows per block: several row passes), every 7th row empty, a diagonal band
    n = 60000
    lens = np.where(np.arange(n) % 7 == 3, 0, 3)
    rp3 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(n), lens)
    col3 = np.clip(rows + np.tile([-1, 0, 1], n)[:rp3[-1]], 0, n - 1).astype(np.int32)
    order = np.lexsort((col3, rows))
    val3 = rng.uniform(-1, 1, rp3[-1]).astype(dtype)
    run(n, n, rp3, col3[order], val3, True, True)


def test_hll_pattern_plan_rebuilds_the_slots_bit_for_bit(gpu, oracle):
    """The HLL twin (hll_lds_local<.., PAT>): in a slab every row of a hack has the hack's length, padding included, so
    the windows of a stencil's slab are rows shifted by a constant as well.  Forced plan against the same handle through
    the 16-bit slot stream: the same bits; both builders of the slab (host, from a resident CSR handle); hacks of
    different lengths, empty rows, a hack range of a bigger matrix."""
    from sparsematrixvectormultiplication_amd import synth
    from sparsematrixvectormultiplication_amd.device import set_tuning
    rng = np.random.default_rng(93)
    M, rp, col, val = synth.kkt_like((24, 24, 25), 5)
    x = rng.uniform(-1, 1, M)
    ref = oracle.csr_serial(rp, col, val, x)

    def check(dev, lo=0, hi=M):
        info = dev.info()
        assert info["local_blocks"] > 0 and 0 < info["pattern_slots"] * 2 <= info["slots"]
        sp.lib().spmv_hip_memset(dev.y_ptr, 0xFF, M * 8)
        y = dev.spmv(x, sp.HLL_LDS)[lo:hi].copy()
        assert np.max(np.abs(y - ref[lo:hi])) <= 1e-10 * np.max(np.abs(ref))
        set_tuning("local_patterns", 0)                           # the same handle through the slot stream
        y0 = dev.spmv(x, sp.HLL_LDS)[lo:hi].copy()
        set_tuning("local_patterns", 1)
        assert y.tobytes() == y0.tobytes()

    set_tuning("local_patterns", 1)
    try:
        hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, M, np.repeat(np.arange(M, dtype=np.int32), np.diff(rp)), col, val))
        with sp.HllDevice(hll) as dev:
            check(dev)
        hacks = (M + 31) // 32
        h0, h1 = hacks // 3, 2 * hacks // 3
        with sp.HllDevice(hll, h0, h1) as part:
            check(part, 32 * h0, min(M, 32 * h1))
        with sp.CsrDevice(M, M, rp, col, val) as cdev:
            with sp.HllDevice.from_csr_device(cdev) as dev2:
                check(dev2)
        # rows of different lengths inside a hack (padding becomes part of the patterns), every 9th row empty
        n = 30000
        lens = np.where(np.arange(n) % 9 == 4, 0, 3 + (np.arange(n) % 5))
        rp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rows = np.repeat(np.arange(n), lens)
        within = np.concatenate([np.arange(k) for k in lens]) if lens.sum() else np.zeros(0, int)
        col2 = np.clip(rows + 2 * within - 3, 0, n - 1).astype(np.int32)
        order = np.lexsort((col2, rows))
        col2 = col2[order]
        keep = np.ones(len(col2), bool)
        keep[1:] = (rows[order][1:] != rows[order][:-1]) | (col2[1:] != col2[:-1])
        rows2, col2 = rows[order][keep], col2[keep]
        rp2 = np.zeros(n + 1, np.int64)
        np.add.at(rp2, rows2 + 1, 1)
        rp2 = np.cumsum(rp2).astype(np.int32)
        val2 = rng.uniform(-1, 1, len(col2))
        x2 = rng.uniform(-1, 1, n)
        ref2 = oracle.csr_serial(rp2, col2, val2, x2)
        with sp.HllDevice(sp.convert_to_hll(sp.PreMatrix.from_arrays(n, n, rows2.astype(np.int32), col2, val2))) as dev3:
            assert dev3.info()["pattern_slots"] > 0
            y = dev3.spmv(x2, sp.HLL_LDS)
            assert np.max(np.abs(y - ref2)) <= 1e-10 * np.max(np.abs(ref2))
            set_tuning("local_patterns", 0)
            assert dev3.spmv(x2, sp.HLL_LDS).tobytes() == y.tobytes()
    finally:
        set_tuning("local_patterns", -1)


def test_pattern_plan_fuzz(gpu, oracle):
    """tests/fuzz_patterns.py with a few dozen cases: random shapes through the forced pattern plan, bit for bit against
    the slot stream and against the oracle within the gate."""
    import fuzz_patterns
    assert fuzz_patterns.run(40, 7, oracle) >= 20


def test_power_iteration_halo_single_rank_communicator(gpu):
    """The RCCL side at world size 1: setup (all-gather of the needs record), an exchange with no peers, the
    all-reduce of the norm; the halo loop then gives the plain loop's result bit for bit."""
    from sparsematrixvectormultiplication_amd.distributed import NativeComm
    from _util import banded_csr
    rng = np.random.default_rng(63)
    n = 4000
    row_ptr, col, val = banded_csr(rng, n, n, 15, 80)
    x0 = rng.uniform(0.5, 1.0, n)
    with sp.CsrDevice(n, n, row_ptr, col, val) as dev:
        dev.set_x(x0)
        lam_plain, _ = dev.power_iterate(4, use_graph=False)
        x_plain = dev.get_x()
        comm = NativeComm(0, 1, lambda ident: ident)
        try:
            info = comm.halo_setup(dev, np.array([0, n], np.int32))
            assert info == {"send_values": 0, "recv_values": 0, "peers": 0}
            comm.halo_exchange(dev.x_ptr, 8)
            dev.set_x(x0)
            lam_halo, ms = dev.power_iterate_halo(4)       # exchange on the second stream beside the interior blocks
            assert ms > 0 and lam_halo == lam_plain and dev.get_x().tobytes() == x_plain.tobytes()
            from sparsematrixvectormultiplication_amd.device import set_tuning
            set_tuning("halo_overlap", 0)                  # strictly in order: the same bits
            try:
                dev.set_x(x0)
                lam_serial, _ = dev.power_iterate_halo(4)
                assert lam_serial == lam_plain and dev.get_x().tobytes() == x_plain.tobytes()
            finally:
                set_tuning("halo_overlap", 1)
            # the loop over a column-split handle (what halo setup makes on a KKT-like cut at world size > 1; here made
            # by hand, a world of one has no halo): two launches per product, overlapped or in order the same bits, and
            # the plain loop's result within the tolerance (a row's two partial sums are added in another order)
            assert dev.split_columns(n // 4, 3 * n // 4)["halo_entries"] > 0
            dev.set_x(x0)
            lam_split, _ = dev.power_iterate_halo(4)
            x_split = dev.get_x()
            assert abs(lam_split - lam_plain) <= 1e-10 * abs(lam_plain)
            assert np.max(np.abs(x_split - x_plain) / np.maximum(np.abs(x_plain), 1e-300)) <= 1e-10
            set_tuning("halo_overlap", 0)
            try:
                dev.set_x(x0)
                lam_split_serial, _ = dev.power_iterate_halo(4)
                assert lam_split_serial == lam_split and dev.get_x().tobytes() == x_split.tobytes()
            finally:
                set_tuning("halo_overlap", 1)
        finally:
            comm.close()
