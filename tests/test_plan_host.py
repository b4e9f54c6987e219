"""The upload-time preprocessing (workgroup blocks, split rows, x-window plan) checked on the
host, without a GPU: spmv_hip_csr_plan_check rebuilds what upload builds and verifies every
invariant the kernels rely on (tests/ run it over the shapes the GPU parity tests use)."""
import numpy as np
import pytest

import sparsematrixvectormultiplication_amd as sp
from _util import banded_csr, random_csr
from conftest import GOLDEN_CASES, golden_path


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_plan_invariants_on_golden_matrices(name):
    csr = sp.convert_in_csr(sp.read_matrix_market(golden_path(name)))
    for vb in (8, 4):
        st = sp.csr_plan_check(csr.M, csr.N, csr.row_ptr, csr.col_idx, vb)
        assert st["gather_blocks"] + st["long_rows"] > 0 or csr.M == 0


@pytest.mark.parametrize("mean,band,empty,far", [(3, 40, 0.4, 0.0), (27, 200, 0.0, 0.3), (300, 900, 0.0, 0.0),
                                                 (1500, 1900, 0.0, 0.0)])
def test_plan_invariants_on_banded_matrices(mean, band, empty, far):
    rng = np.random.default_rng(mean)
    row_ptr, col, _ = banded_csr(rng, 4001, 5603, mean, band, empty, far_frac=far)
    for vb in (8, 4):
        st = sp.csr_plan_check(4001, 5603, row_ptr, col, vb)
        assert st["local_blocks"] > 0 and st["widest_lines"] <= 256 and st["split_rows"] == 0
        assert st["lines"] >= st["local_blocks"]


def test_plan_is_refused_or_partial_where_it_should_be():
    rng = np.random.default_rng(5)
    row_ptr, col, _ = random_csr(rng, 3000, 40000, 30, 60, 0.0)       # scattered: no plan
    assert sp.csr_plan_check(3000, 40000, row_ptr, col)["local_blocks"] == 0
    # banded + a few wild rows: plan kept, wild rows split -- from 2^20 entries on (the split-row kernels are two more
    # launches, which a smaller matrix cannot afford: it keeps its gather kernel and one launch)
    def with_wild_rows(M, N):
        row_ptr, col, _ = banded_csr(rng, M, N, 30, 200)
        lens = np.diff(row_ptr).astype(np.int64)
        cols = [col[row_ptr[r]:row_ptr[r + 1]] for r in range(M)]
        for r in (5, M // 2, M - 1):
            cols[r] = np.sort(rng.choice(N, 500, replace=False)).astype(np.int32)
            lens[r] = 500
        return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), cols
    rp_small, cols_small = with_wild_rows(3000, 40000)
    st_small = sp.csr_plan_check(3000, 40000, rp_small, np.concatenate(cols_small))
    assert st_small["local_blocks"] == 0 and st_small["split_rows"] == 0 and st_small["long_rows"] == 0
    M_big = 40000
    rp, cols = with_wild_rows(M_big, 40000)
    assert rp[-1] >= 1 << 20
    st = sp.csr_plan_check(M_big, 40000, rp, np.concatenate(cols))
    assert st["local_blocks"] > 0 and st["split_rows"] == 3 and st["long_rows"] == 3
    # unsorted and repeated columns inside rows are fine
    shuffled = np.concatenate([rng.permutation(c) for c in cols])
    st2 = sp.csr_plan_check(M_big, 40000, rp, shuffled)
    assert st2["local_blocks"] == st["local_blocks"] and st2["lines"] == st["lines"]
    with pytest.raises(ValueError, match="outside"):
        sp.csr_plan_check(2, 2, np.array([0, 1, 2], np.int32), np.array([0, 7], np.int32))


def test_long_rows_among_short_ones_get_blocks_of_few_rows():
    """Circuit-shaped matrix (adder_dcop_32-size: short rows plus a few rows and columns that touch a quarter of the
    matrix): the row-sum phase gives every row of a block the same number of lanes, so upload closes blocks around the
    long rows (plan_check verifies that no lane adds up more than 64 entries unless its row is alone); the matrix
    keeps its x-window plan and one launch -- no split rows."""
    import scipy.sparse as sps
    rng = np.random.default_rng(32)
    n = 1813
    r = np.repeat(np.arange(n), 4)
    c = np.clip(r + rng.integers(-50, 51, len(r)), 0, n - 1)
    a = sps.csr_matrix((np.ones(len(r)), (r, c)), shape=(n, n)).tolil()
    for k in rng.integers(0, n, 4):
        cols = rng.integers(0, n, 450)
        a[k, cols] = 1.0
        a[cols, k] = 1.0
    a = a.tocsr()
    a.sort_indices()
    st = sp.csr_plan_check(n, n, a.indptr.astype(np.int32), a.indices.astype(np.int32))
    assert st["local_blocks"] > 0 and st["split_rows"] == 0 and st["long_rows"] == 0
    assert st["local_blocks"] == st["gather_blocks"]
    assert st["local_blocks"] >= a.nnz // 2048 + 1 + 4      # the four long rows made blocks of their own
    # the same rows in an HLL slab: windows on the border between hacks of short and of long rows
    from _util import coo_from_csr
    rr, cc, vv = coo_from_csr(a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data, rng)
    assert sp.hll_plan_check(sp.convert_to_hll(sp.PreMatrix.from_arrays(n, n, rr, cc, vv)))["gather_windows"] > 0


def test_plan_on_the_cant_like_matrix():
    from sparsematrixvectormultiplication_amd import synth
    M, row_ptr, col, _ = synth.fem_like(synth.FEM_GRID, 1)
    st = sp.csr_plan_check(M, M, row_ptr, col)
    assert st["local_blocks"] == st["gather_blocks"] and st["widest_lines"] <= 64 and st["long_rows"] == 0


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_hll_plan_invariants_on_golden_matrices(name):
    hll = sp.convert_to_hll(sp.read_matrix_market(golden_path(name)))
    st = sp.hll_plan_check(hll)
    assert st["gather_windows"] >= (1 if hll.M else 0)


@pytest.mark.parametrize("mean,band,empty,far", [(3, 40, 0.4, 0.0), (27, 200, 0.0, 0.3), (300, 900, 0.0, 0.0)])
def test_hll_plan_invariants_on_banded_matrices(mean, band, empty, far):
    from _util import coo_from_csr
    rng = np.random.default_rng(100 + mean)
    M, N = 2051, 2600
    row_ptr, col, val = banded_csr(rng, M, N, mean, band, empty, far_frac=far)
    r, c, v = coo_from_csr(row_ptr, col, val, rng)
    hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, N, r, c, v))
    st = sp.hll_plan_check(hll)
    assert st["local_windows"] > 0 and st["widest_lines"] <= 256
    # scattered columns: no plan, the check of the plain windows still passes
    row_ptr, col, val = random_csr(rng, 500, 30000, 40, 80, 0.0)
    r, c, v = coo_from_csr(row_ptr, col, val, rng)
    assert sp.hll_plan_check(sp.convert_to_hll(sp.PreMatrix.from_arrays(500, 30000, r, c, v)))["local_windows"] == 0


def test_halo_plan_is_consistent_between_ranks():
    """spmv_hip_halo_plan (pure host logic): what rank r sends to p is exactly what p receives from r, segment
    by segment and in the same order; every needed entry outside a rank's own range is received exactly once;
    nothing is exchanged for entries a rank owns itself."""
    from sparsematrixvectormultiplication_amd.distributed import halo_plan
    rng = np.random.default_rng(9)
    for ranks in (1, 2, 3, 8):
        n = 10000
        cuts = np.sort(rng.choice(np.arange(1, n), ranks - 1, replace=False)) if ranks > 1 else np.array([], int)
        bounds = np.concatenate([[0], cuts, [n]]).astype(np.int32)
        if ranks >= 3:
            bounds[2] = bounds[1]          # a rank that owns nothing
        needs = []
        for r in range(ranks):
            k = int(rng.integers(0, 6))
            pts = np.sort(rng.choice(n + 1, 2 * k, replace=False))
            needs.append([(int(pts[2 * j]), int(pts[2 * j + 1])) for j in range(k)])
        plans = [halo_plan(r, bounds, needs) for r in range(ranks)]
        for r in range(ranks):
            send, recv = plans[r]
            for p in range(ranks):
                to_p = [(lo, hi) for q, lo, hi in send if q == p]
                from_r = [(lo, hi) for q, lo, hi in plans[p][1] if q == r]
                assert to_p == from_r, (ranks, r, p)
                assert all(bounds[r] <= lo < hi <= bounds[r + 1] for lo, hi in to_p)
            got = np.zeros(n, int)
            for q, lo, hi in recv:
                assert q != r and bounds[q] <= lo < hi <= bounds[q + 1]
                got[lo:hi] += 1
            want = np.zeros(n, int)
            for lo, hi in needs[r]:
                want[lo:hi] = 1
            want[bounds[r]:bounds[r + 1]] = 0
            assert np.array_equal(got, want), (ranks, r)


# ---------------------------------------------------------------- csr_tile plan (2-D tiles)
def _scattered(rng, M, N, mean, sigma=None):
    """rows of ~mean entries, columns uniform over N (sigma None) or gaussian around the diagonal"""
    lens = rng.poisson(mean, M).astype(np.int64)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(M), lens)
    if sigma is None:
        col = rng.integers(0, N, rp[-1])
    else:
        col = np.clip(rows * (N - 1) // max(M - 1, 1) + np.rint(rng.normal(0, sigma, rp[-1])).astype(np.int64), 0, N - 1)
    order = np.lexsort((col, rows))     # ascending columns inside each row, repeats allowed
    return rp, col[order].astype(np.int32)


@pytest.mark.parametrize("vb", [8, 4])
@pytest.mark.parametrize("rows_per_block,chunk", [(256, 2048), (2048, 2048), (3072, 2048)])
def test_tile_plan_on_scattered_banded_and_skewed_matrices(vb, rows_per_block, chunk):
    import functools
    real = sp.csr_tile_plan_check
    sp_check = functools.partial(real, chunk=chunk)
    rng = np.random.default_rng(77 + vb)
    # uniformly random columns: every block needs several gather passes
    rp, col = _scattered(rng, 9000, 3_000_000, 20)
    st = sp_check(9000, 3_000_000, rp, col, vb, rows_per_block)
    assert st["entries"] == rp[-1] and st["split_rows"] == 0 and st["passes"] >= st["blocks"]
    # the packed plan of the same matrix (the check builds both kinds): every pass cut at the window and staged, which
    # on scattered columns means many more passes -- what upload's fallback rule looks at
    # -- or, since windows with a handful of entries go to the remainder (held, not staged), almost no passes at all:
    # what upload's fallback rule looks at
    assert st["packed_entries"] == rp[-1] and st["packed_staged_entries"] < 0.5 * rp[-1]
    assert 0 <= st["packed_max_window"] <= 32768 // vb
    if rows_per_block == 2048:
        assert st["passes"] > 2 * st["blocks"] and st["staged_entries"] == 0
    # a band: passes get staged in LDS
    rp, col = _scattered(rng, 9000, 9000, 12, sigma=300)
    st = sp_check(9000, 9000, rp, col, vb, rows_per_block)
    assert st["staged_entries"] > 0.9 * st["entries"] and 0 < st["max_window"] <= 32768 // vb
    assert st["packed_entries"] == st["entries"] and 0.95 * st["entries"] <= st["packed_staged_entries"] <= st["entries"]
    assert st["packed_passes"] < 2 * st["passes"] + 8
    # skewed row lengths: runs longer than a lane's share (sub-runs), rows beyond the limit (split)
    lens = np.minimum((1.08 / rng.random(5000)).astype(np.int64), 40000)
    lens[[7, 4100]] = [30000, 70]
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.integers(0, 600_000, n)) for n in lens]).astype(np.int32)
    st = sp_check(5000, 600_000, rp, col, vb, rows_per_block, lmax=5000)
    assert st["split_rows"] == int((lens > 5000).sum()) >= 1
    assert st["entries"] == int(lens[lens <= 5000].sum())
    # every row beyond a tiny limit is split; empty matrix; one row
    st = sp_check(5000, 600_000, rp, col, vb, rows_per_block, lmax=1)
    assert st["entries"] == int((lens == 1).sum())
    # equal-work blocks: more of them than ceil(M / rows), never fewer; equal-height blocks on request
    stb = sp_check(5000, 600_000, rp, col, vb, rows_per_block, lmax=5000)
    stn = sp_check(5000, 600_000, rp, col, vb, rows_per_block, lmax=5000, balance=False)
    assert stn["blocks"] == -(-5000 // rows_per_block) <= stb["blocks"] and stb["entries"] == stn["entries"]
    assert sp_check(0, 5, np.zeros(1, np.int32), np.zeros(0, np.int32), vb, rows_per_block)["blocks"] == 0
    st = sp_check(1, 5, np.array([0, 3], np.int32), np.array([4, 0, 4], np.int32), vb, rows_per_block)
    assert st["blocks"] == 1 and st["entries"] == 3


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_tile_plan_on_golden_matrices(name):
    csr = sp.convert_in_csr(sp.read_matrix_market(golden_path(name)))
    st = sp.csr_tile_plan_check(csr.M, csr.N, csr.row_ptr, csr.col_idx, 8, 256, lmax=64)
    assert st["entries"] + 0 <= csr.nz


def test_tile_auto_plan_decisions():
    """What upload decides about the csr_tile plan (spmv_hip_csr_tile_auto_plan, host only): a band gets the packed
    plan, uniformly scattered columns the plan with gather passes and tall blocks -- but only from 800 000 columns on --
    and the number of row blocks is fitted to whole rounds of the workgroup places (here 16 places, so that a small
    matrix shows it): at most k x places blocks with the last round at least 90 % full, every block within the
    tallest the LDS takes, every block in exactly one stream."""
    from sparsematrixvectormultiplication_amd.device import set_tuning
    rng = np.random.default_rng(5)
    try:
        set_tuning("tile_places", 16)
        # a band, forced (the auto rule wants 2^20 rows)
        set_tuning("stream_tile", 1)
        M = N = 300_000
        rp, col = _scattered(rng, M, N, 5, sigma=900)
        st = sp.csr_tile_auto_plan(M, N, rp, col, 8)
        assert st["tiles"] and st["packed"] and not st["scattered"] and st["entries"] == rp[-1]
        assert st["tallest_block"] <= st["rows_per_block"] == 5888 and st["streams"] == 16
        rounds = -(-st["blocks"] // 16)
        assert st["blocks"] <= rounds * 16 and st["blocks"] >= 0.9 * rounds * 16, st
        st32 = sp.csr_tile_auto_plan(M, N, rp, col, 4)
        assert st32["packed"] and st32["rows_per_block"] == 12032
        # the same without the fitting: blocks of 4096 fp64 rows
        set_tuning("tile_fit", 0)
        plain = sp.csr_tile_auto_plan(M, N, rp, col, 8)
        assert plain["rows_per_block"] == 4096 and plain["blocks"] >= M // 4096
        set_tuning("tile_fit", 1)
        # one workgroup per block
        set_tuning("tile_streams", 0)
        assert sp.csr_tile_auto_plan(M, N, rp, col, 8)["streams"] >= st["blocks"]
        set_tuning("tile_streams", 1)
        # scattered columns: gather passes, one workgroup per CU geometry, blocks as tall as the LDS takes
        M, N = 200_000, 2_000_000
        rp, col = _scattered(rng, M, N, 6)
        sc = sp.csr_tile_auto_plan(M, N, rp, col, 8)
        assert sc["tiles"] and not sc["packed"] and sc["scattered"] and sc["tallest_block"] <= sc["rows_per_block"] <= 16128
        rounds = -(-sc["blocks"] // 16)
        assert sc["blocks"] <= rounds * 16
        # auto: scattered columns get tiles from 800 000 rows and columns on ...
        set_tuning("stream_tile", -1)
        assert sp.csr_tile_auto_plan(M, N, rp, col, 8)["tiles"] == 0      # 200 000 rows
        M = 900_000
        rp2, col2 = _scattered(rng, M, 900_000, 3)
        big = sp.csr_tile_auto_plan(M, 900_000, rp2, col2, 8)
        assert big["tiles"] == 1 and not big["packed"]
        rp2, col2 = _scattered(rng, M, 600_000, 3)
        assert sp.csr_tile_auto_plan(M, 600_000, rp2, col2, 8)["tiles"] == 0
        # ... a band of dense rows already from 4 Mi entries on: packed, one thin block per place
        M = N = 80_000
        rp3, col3 = _scattered(rng, M, N, 60, sigma=700)
        mid = sp.csr_tile_auto_plan(M, N, rp3, col3, 8)
        assert mid["tiles"] == 1 and mid["packed"] and 14 <= mid["blocks"] <= 16 and mid["tallest_block"] < 0.1 * M, mid

    finally:
        for k, v in (("tile_places", 0), ("stream_tile", -1), ("tile_fit", 1), ("tile_streams", 1)):
            set_tuning(k, v)
