"""Host-side mirror of the reference's C API (same names, same argument meaning,
same 0 / -1 error behaviour turned into exceptions) over libspmv_amd.so.

Everything here is plumbing around the hot path: Matrix Market ingest
(reference src/matrix_parser.c), COO->CSR / COO->HLL builders
(src/csr_matrix.c:63-126, src/hll_matrix.c:37-257), the nnz-balanced
partitioners (src/csr_matrix.c:167-266, src/hll_matrix.c:410-540) and the
difference metrics (src/performance_calculate.c:116-178,
cuda_src/performance_calculate.cu:103-148).  The arithmetic is done by the C
library; Python only owns the buffers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat

HACK_SIZE = 32
ITERATION_SKIP = 5


def _ip(a):
    return a.ctypes.data_as(nat.c_int_p)


def _dp(a):
    return a.ctypes.data_as(nat.c_double_p)


def _view(ptr, n, dtype):
    if n <= 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,))


class PreMatrix:
    """COO triplets (reference PreMatrix, libs/matrix_parser.h:6-14)."""

    def __init__(self, c_struct, owned_by_c, keep=()):
        self.c = c_struct
        self._owned_by_c = owned_by_c
        self._keep = keep

    M = property(lambda s: s.c.M)
    N = property(lambda s: s.c.N)
    nz = property(lambda s: s.c.nz)
    I = property(lambda s: _view(s.c.I, s.c.nz, np.int32))
    J = property(lambda s: _view(s.c.J, s.c.nz, np.int32))
    val = property(lambda s: _view(s.c.val, s.c.nz, np.float64))
    type = property(lambda s: bytes(s.c.type))

    @classmethod
    def from_arrays(cls, M, N, I, J, val, typecode=b"MCRG"):
        I = np.ascontiguousarray(I, dtype=np.int32)
        J = np.ascontiguousarray(J, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        if not (len(I) == len(J) == len(val)):
            raise ValueError("I, J, val must have the same length")
        s = nat.PreMatrix()
        s.M, s.N, s.nz = int(M), int(N), len(I)
        s.I, s.J, s.val = _ip(I), _ip(J), _dp(val)
        s.type = typecode
        return cls(s, owned_by_c=False, keep=(I, J, val))

    def close(self):
        if self._owned_by_c and self.c is not None:
            nat.lib().free_pre_matrix(C.byref(self.c))
        self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_matrix_market(filename) -> PreMatrix:
    """reference: read_matrix_market(filename, PreMatrix*) -> 0 / -1."""
    s = nat.PreMatrix()
    nat.lib().init_pre_matrix(C.byref(s))
    if nat.lib().read_matrix_market(str(filename).encode(), C.byref(s)) != 0:
        raise ValueError(f"read_matrix_market({filename!r}) failed (-1)")
    return PreMatrix(s, owned_by_c=True)


class CsrHost:
    """Host CSR (reference CSRMatrix, libs/csr_matrix.h:8-16)."""

    def __init__(self, c_struct, owned_by_c, keep=()):
        self.c = c_struct
        self._owned_by_c = owned_by_c
        self._keep = keep

    M = property(lambda s: s.c.M)
    N = property(lambda s: s.c.N)
    nz = property(lambda s: s.c.nz)
    row_ptr = property(lambda s: _view(s.c.row_ptr, s.c.M + 1, np.int32))
    col_idx = property(lambda s: _view(s.c.col_idx, s.c.nz, np.int32))
    values = property(lambda s: _view(s.c.values, s.c.nz, np.float64))

    @classmethod
    def from_arrays(cls, M, N, row_ptr, col_idx, values):
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float64)
        s = nat.CSRMatrix()
        s.M, s.N, s.nz = int(M), int(N), len(col_idx)
        s.row_ptr, s.col_idx, s.values = _ip(row_ptr), _ip(col_idx), _dp(values)
        s.type = b"MCRG"
        return cls(s, owned_by_c=False, keep=(row_ptr, col_idx, values))

    def close(self):
        if self._owned_by_c and self.c is not None:
            nat.lib().free_csr_matrix(C.byref(self.c))
        self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def convert_in_csr(pre: PreMatrix, matrix_name: str = "") -> CsrHost:
    """reference: convert_in_csr(pre, csr, name) -> 0 / -1 (src/csr_matrix.c:63-126)."""
    s = nat.CSRMatrix()
    if nat.lib().convert_in_csr(C.byref(pre.c), C.byref(s), matrix_name.encode()) != 0:
        raise MemoryError("convert_in_csr failed (-1)")
    return CsrHost(s, owned_by_c=True)


def save_csr_binary(csr: CsrHost, path: str, source_mtx: str = None) -> None:
    """Binary sidecar of a built CSR matrix (include/csr_cache.h)."""
    src = source_mtx.encode() if source_mtx else None
    if nat.lib().save_csr_binary(C.byref(csr.c), str(path).encode(), src) != 0:
        raise OSError(f"save_csr_binary({path}) failed (-1)")


def load_csr_binary(path: str, source_mtx: str = None) -> CsrHost:
    s = nat.CSRMatrix()
    src = source_mtx.encode() if source_mtx else None
    if nat.lib().load_csr_binary(str(path).encode(), C.byref(s), src) != 0:
        raise ValueError(f"load_csr_binary({path}): missing, stale or damaged sidecar (-1)")
    return CsrHost(s, owned_by_c=True)


def load_csr_cached(mtx_path: str):
    """(CsrHost, from_cache): "<mtx>.csrbin" when valid and fresh, else parse + build + write it."""
    s = nat.CSRMatrix()
    hit = C.c_int(0)
    if nat.lib().load_csr_cached(str(mtx_path).encode(), C.byref(s), C.byref(hit)) != 0:
        raise ValueError(f"load_csr_cached({mtx_path}) failed (-1)")
    return CsrHost(s, owned_by_c=True), bool(hit.value)


class HllHost:
    """Host HLL (reference HLLMatrix / ELLPACKBlock, libs/hll_matrix.h:15-27)."""

    def __init__(self, c_struct, total_rows, n_cols):
        self.c = c_struct
        self.M = total_rows
        self.N = n_cols

    num_blocks = property(lambda s: s.c.num_blocks)

    def block(self, b):
        blk = self.c.blocks[b]
        n = blk.M * blk.MAXNZ
        return blk.M, blk.MAXNZ, _view(blk.JA, n, np.int32), _view(blk.AS, n, np.float64)

    @property
    def maxnz(self):
        return np.array([self.c.blocks[b].MAXNZ for b in range(self.c.num_blocks)], dtype=np.int32)

    @property
    def slots(self):
        return int(sum(self.c.blocks[b].M * self.c.blocks[b].MAXNZ
                       for b in range(self.c.num_blocks)))

    def close(self):
        if self.c is not None:
            nat.lib().free_hll_matrix(C.byref(self.c))
        self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def convert_to_hll(pre: PreMatrix) -> HllHost:
    """reference: convert_to_hll(pre, hll) -> 0 / -1 (src/hll_matrix.c:37-257)."""
    s = nat.HLLMatrix()
    if nat.lib().convert_to_hll(C.byref(pre.c), C.byref(s)) != 0:
        raise ValueError("convert_to_hll failed (-1)")
    return HllHost(s, pre.M, pre.N)


def _take_and_free(ptr, n):
    out = np.array(np.ctypeslib.as_array(ptr, shape=(n,)), dtype=np.int32) if n > 0 else \
        np.zeros(0, np.int32)
    if ptr:
        _libc_free(ptr)
    return out


def _libc_free(ptr):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(C.cast(ptr, C.c_void_p))


def prepare_thread_distribution(row_ptr, num_threads, total_nnz=None):
    """reference: prepare_thread_distribution (src/csr_matrix.c:167-266).

    Returns (starts, ends) of the non-empty chunks; end is exclusive.
    """
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    M = len(row_ptr) - 1
    if total_nnz is None:
        total_nnz = int(row_ptr[-1] - row_ptr[0]) if M > 0 else 0
    s, e = nat.c_int_p(), nat.c_int_p()
    n = nat.lib().prepare_thread_distribution(M, _ip(row_ptr), int(num_threads), int(total_nnz),
                                              C.byref(s), C.byref(e))
    return _take_and_free(s, n), _take_and_free(e, n)


def prepare_thread_distribution_hll(hll: HllHost, num_threads):
    """reference: prepare_thread_distribution_hll (src/hll_matrix.c:410-540)."""
    s, e = nat.c_int_p(), nat.c_int_p()
    n = nat.lib().prepare_thread_distribution_hll(C.byref(hll.c), int(num_threads), C.byref(s),
                                                  C.byref(e))
    return _take_and_free(s, n), _take_and_free(e, n)


def partition_rows(row_ptr, parts):
    """Row bounds [0 = b0 <= b1 <= ... <= b_parts = M] for `parts` GPUs."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    bounds = np.zeros(parts + 1, dtype=np.int32)
    if nat.lib().spmv_hip_partition_rows(len(row_ptr) - 1, _ip(row_ptr), int(parts),
                                         _ip(bounds)) != 0:
        raise ValueError(nat.lib().spmv_hip_last_error().decode())
    return bounds


def csr_plan_check(M, N, row_ptr, col_idx, value_bytes=8):
    """Host-only self-check of the upload-time plan (spmv_hip_csr_plan_check); returns its stats."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    stats = np.zeros(6, dtype=np.int32)
    if nat.lib().spmv_hip_csr_plan_check(int(M), int(N), _ip(row_ptr), _ip(col_idx), int(value_bytes),
                                         _ip(stats)) != 0:
        raise ValueError(nat.lib().spmv_hip_last_error().decode())
    return dict(zip(("gather_blocks", "local_blocks", "lines", "widest_lines", "long_rows", "split_rows"),
                    (int(v) for v in stats)))


def csr_tile_plan_check(M, N, row_ptr, col_idx, value_bytes=8, rows_per_block=2048, lmax=1024, density=16, chunk=2048,
                        balance=True):
    """Host-only self-check of the csr_tile plan (spmv_hip_csr_tile_plan_check); returns its stats."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    stats = np.zeros(12, dtype=np.int64)  # the plan with gather passes, then the packed one (every pass staged)
    if nat.lib().spmv_hip_csr_tile_plan_check(int(M), int(N), _ip(row_ptr), _ip(col_idx), int(value_bytes),
                                              int(rows_per_block), int(lmax), int(density), int(chunk),
                                              int(bool(balance)), stats.ctypes.data_as(C.POINTER(C.c_longlong))) != 0:
        raise ValueError(nat.lib().spmv_hip_last_error().decode())
    names = ("blocks", "passes", "entries", "staged_entries", "split_rows", "max_window")
    return dict(zip(names + tuple("packed_" + n for n in names), (int(v) for v in stats)))


def csr_tile_auto_plan(M, N, row_ptr, col_idx, value_bytes=8):
    """What upload would decide about the csr_tile plan of this structure under the current tunings (host only)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    stats = np.zeros(10, dtype=np.int64)
    if nat.lib().spmv_hip_csr_tile_auto_plan(int(M), int(N), _ip(row_ptr), _ip(col_idx), int(value_bytes),
                                             stats.ctypes.data_as(C.POINTER(C.c_longlong))) != 0:
        raise ValueError(nat.lib().spmv_hip_last_error().decode())
    names = ("tiles", "packed", "scattered", "rows_per_block", "blocks", "streams", "passes", "tallest_block",
             "long_items", "entries")
    return dict(zip(names, (int(v) for v in stats)))


def hll_plan_check(hll: "HllHost"):
    """Host-only self-check of the HLL upload-time plan (spmv_hip_hll_plan_check); returns its stats."""
    stats = np.zeros(4, dtype=np.int32)
    if nat.lib().spmv_hip_hll_plan_check(C.byref(hll.c), int(hll.M), int(hll.N), _ip(stats)) != 0:
        raise ValueError(nat.lib().spmv_hip_last_error().decode())
    return dict(zip(("gather_windows", "local_windows", "lines", "widest_lines"), (int(v) for v in stats)))


def partition_hacks(hll: "HllHost", parts):
    """Hack bounds [0 = b0 <= ... <= b_parts = num_blocks] for `parts` GPUs (reference K8 greedy)."""
    bounds = np.zeros(parts + 1, dtype=np.int32)
    if nat.lib().spmv_hip_partition_hacks(C.byref(hll.c), int(parts), _ip(bounds)) != 0:
        raise ValueError(nat.lib().spmv_hip_last_error().decode())
    return bounds


def hack_bounds_to_rows(hack_bounds, M):
    """Row bounds of a hack partition: 32 x the hack bounds, clipped to M."""
    return np.minimum(np.asarray(hack_bounds, dtype=np.int64) * HACK_SIZE, M).astype(np.int32)


def compute_difference_metrics(ref, res, abs_tol=1e-5, rel_tol=1e-4):
    """CPU-build measure; the reference calls it with (1e-5, 1e-4) (main.c:145)."""
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.float64)
    return nat.lib().computeDifferenceMetrics(_dp(ref), _dp(res), len(ref), abs_tol, rel_tol, False)


def compute_difference_metrics_gpu(ref, res, rel_tol=1e-4):
    """GPU-build measure (cuda_src/performance_calculate.cu:103-148)."""
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.float64)
    return nat.lib().computeDifferenceMetricsGpu(_dp(ref), _dp(res), len(ref), rel_tol, False)


def calculate_flops(nz, seconds):
    return nat.lib().calculate_flops(int(nz), float(seconds))


def init_vector_at_one(n):
    v = np.empty(n, dtype=np.float64)
    nat.lib().init_vector_at_one(_dp(v), n)
    return v
