"""ctypes access to the CHECKER libraries -- test infrastructure only.

  liboracle_spmv.so    this repo's CPU restatement (oracle/cpu_spmv.c)
  _ref/libspmv_ref.so  the reference itself, compiled from /root/reference by
                       oracle/Makefile (absent when it was never built)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from sparsematrixvectormultiplication_amd import _native as nat

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_PATH = os.path.join(_HERE, "liboracle_spmv.so")
REF_PATH = os.path.join(_HERE, "_ref", "libspmv_ref.so")

_ip, _dp = nat.c_int_p, nat.c_double_p


def build():
    """Compile the oracle (and the reference, where /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class Oracle:
    """K1-K7 restated (see cpu_spmv.c for the reference line of each)."""

    def __init__(self, path=ORACLE_PATH):
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.csr_matrix_vector_mult.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp]
        L.csr_matrix_vector_mult.restype = None
        for name in ("spvm_csr_parallel", "spvm_csr_parallel_simd"):
            fn = getattr(L, name)
            fn.argtypes = [_ip, _ip, _dp, _dp, _dp, C.c_int, _ip, _ip]
            fn.restype = None
        L.spmv_hll_serial.argtypes = [C.c_int, C.POINTER(nat.ELLPACKBlock), _dp, _dp]
        L.spmv_hll_serial.restype = None
        for name in ("spmv_hll", "spmv_hll_simd"):
            fn = getattr(L, name)
            fn.argtypes = [C.POINTER(nat.ELLPACKBlock), _dp, _dp, C.c_int, _ip, _ip]
            fn.restype = None
        self.L = L
        self.extra = hasattr(L, "oracle_csr_f32_accum64")
        if self.extra:
            L.oracle_csr_f32_accum64.argtypes = [C.c_int, _ip, _ip, nat.c_float_p, nat.c_float_p, _dp]
            L.oracle_csr_f32_accum64.restype = None
            L.oracle_csr_row_abs_sums.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp]
            L.oracle_csr_row_abs_sums.restype = None
            L.oracle_max_threads.restype = C.c_int

    # K1 -- the oracle proper
    def csr_serial(self, row_ptr, col_idx, values, x):
        row_ptr, col_idx = _arr(row_ptr, np.int32), _arr(col_idx, np.int32)
        values, x = _arr(values, np.float64), _arr(x, np.float64)
        y = np.zeros(len(row_ptr) - 1, dtype=np.float64)  # K1 accumulates into zeroed y
        self.L.csr_matrix_vector_mult(len(y), row_ptr.ctypes.data_as(_ip),
                                      col_idx.ctypes.data_as(_ip), values.ctypes.data_as(_dp),
                                      x.ctypes.data_as(_dp), y.ctypes.data_as(_dp))
        return y

    def csr_parallel(self, row_ptr, col_idx, values, x, starts, ends, simd=False, y=None):
        row_ptr, col_idx = _arr(row_ptr, np.int32), _arr(col_idx, np.int32)
        values, x = _arr(values, np.float64), _arr(x, np.float64)
        starts, ends = _arr(starts, np.int32), _arr(ends, np.int32)
        if y is None:
            y = np.zeros(len(row_ptr) - 1, dtype=np.float64)
        fn = self.L.spvm_csr_parallel_simd if simd else self.L.spvm_csr_parallel
        fn(row_ptr.ctypes.data_as(_ip), col_idx.ctypes.data_as(_ip), values.ctypes.data_as(_dp),
           x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), len(starts), starts.ctypes.data_as(_ip),
           ends.ctypes.data_as(_ip))
        return y

    def hll_serial(self, hll, x):
        """hll: sparsematrixvectormultiplication_amd.HllHost (same struct layout)."""
        x = _arr(x, np.float64)
        y = np.zeros(hll.num_blocks * 32, dtype=np.float64)  # reference allocs num_blocks*32
        self.L.spmv_hll_serial(hll.num_blocks, hll.c.blocks, x.ctypes.data_as(_dp),
                               y.ctypes.data_as(_dp))
        return y[:hll.M]

    def hll_parallel(self, hll, x, starts, ends, simd=False):
        x = _arr(x, np.float64)
        starts, ends = _arr(starts, np.int32), _arr(ends, np.int32)
        y = np.zeros(hll.num_blocks * 32, dtype=np.float64)
        fn = self.L.spmv_hll_simd if simd else self.L.spmv_hll
        fn(hll.c.blocks, x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), len(starts),
           starts.ctypes.data_as(_ip), ends.ctypes.data_as(_ip))
        return y[:hll.M]

    def csr_f32_accum64(self, row_ptr, col_idx, values_f32, x_f32):
        row_ptr, col_idx = _arr(row_ptr, np.int32), _arr(col_idx, np.int32)
        values_f32, x_f32 = _arr(values_f32, np.float32), _arr(x_f32, np.float32)
        y = np.zeros(len(row_ptr) - 1, dtype=np.float64)
        self.L.oracle_csr_f32_accum64(len(y), row_ptr.ctypes.data_as(_ip),
                                      col_idx.ctypes.data_as(_ip),
                                      values_f32.ctypes.data_as(nat.c_float_p),
                                      x_f32.ctypes.data_as(nat.c_float_p), y.ctypes.data_as(_dp))
        return y

    def row_abs_sums(self, row_ptr, col_idx, values, x):
        row_ptr, col_idx = _arr(row_ptr, np.int32), _arr(col_idx, np.int32)
        values, x = _arr(values, np.float64), _arr(x, np.float64)
        out = np.zeros(len(row_ptr) - 1, dtype=np.float64)
        self.L.oracle_csr_row_abs_sums(len(out), row_ptr.ctypes.data_as(_ip),
                                       col_idx.ctypes.data_as(_ip), values.ctypes.data_as(_dp),
                                       x.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
        return out

    def max_threads(self):
        return self.L.oracle_max_threads()


def have_reference() -> bool:
    return os.path.exists(REF_PATH)


class Reference:
    """The compiled reference (oracle/_ref/libspmv_ref.so): its own parser,
    builders, partitioners, kernels and metrics, called through the struct
    layouts of its own headers (identical to ours by construction)."""

    def __init__(self):
        if not have_reference():
            raise FileNotFoundError(REF_PATH)
        L = C.CDLL(REF_PATH)
        P, Cs, H = nat.PreMatrix, nat.CSRMatrix, nat.HLLMatrix
        L.read_matrix_market.argtypes = [C.c_char_p, C.POINTER(P)]
        L.convert_in_csr.argtypes = [C.POINTER(P), C.POINTER(Cs), C.c_char_p]
        L.convert_to_hll.argtypes = [C.POINTER(P), C.POINTER(H)]
        L.csr_matrix_vector_mult.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp]
        L.csr_matrix_vector_mult.restype = None
        L.spmv_hll_serial.argtypes = [C.c_int, C.POINTER(nat.ELLPACKBlock), _dp, _dp]
        L.spmv_hll_serial.restype = None
        for name in ("spvm_csr_parallel", "spvm_csr_parallel_simd"):
            getattr(L, name).argtypes = [_ip, _ip, _dp, _dp, _dp, C.c_int, _ip, _ip]
            getattr(L, name).restype = None
        for name in ("spmv_hll", "spmv_hll_simd"):
            getattr(L, name).argtypes = [C.POINTER(nat.ELLPACKBlock), _dp, _dp, C.c_int, _ip, _ip]
            getattr(L, name).restype = None
        L.prepare_thread_distribution.argtypes = [C.c_int, _ip, C.c_int, C.c_longlong,
                                                  C.POINTER(_ip), C.POINTER(_ip)]
        L.prepare_thread_distribution_hll.argtypes = [C.POINTER(H), C.c_int, C.POINTER(_ip),
                                                      C.POINTER(_ip)]
        L.computeDifferenceMetrics.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_double, C.c_bool]
        L.computeDifferenceMetrics.restype = nat.DiffMetrics
        L.calculate_flops.argtypes = [C.c_int, C.c_double]
        L.calculate_flops.restype = C.c_double
        L.sort_row.argtypes = [_ip, _dp, C.c_size_t, C.c_size_t]
        L.sort_row.restype = None
        self.L = L

    def load(self, path):
        """(PreMatrix, CSRMatrix, HLLMatrix) C structs built by the reference."""
        pre, csr, hll = nat.PreMatrix(), nat.CSRMatrix(), nat.HLLMatrix()
        if self.L.read_matrix_market(str(path).encode(), C.byref(pre)) != 0:
            raise ValueError(f"reference failed to read {path}")
        if self.L.convert_in_csr(C.byref(pre), C.byref(csr), b"") != 0:
            raise ValueError("reference convert_in_csr failed")
        if self.L.convert_to_hll(C.byref(pre), C.byref(hll)) != 0:
            raise ValueError("reference convert_to_hll failed")
        return pre, csr, hll

    def csr_serial(self, csr, x):
        x = _arr(x, np.float64)
        y = np.zeros(csr.M, dtype=np.float64)
        self.L.csr_matrix_vector_mult(csr.M, csr.row_ptr, csr.col_idx, csr.values,
                                      x.ctypes.data_as(_dp), y.ctypes.data_as(_dp))
        return y

    def hll_serial(self, hll, M, x):
        x = _arr(x, np.float64)
        y = np.zeros(hll.num_blocks * 32, dtype=np.float64)
        self.L.spmv_hll_serial(hll.num_blocks, hll.blocks, x.ctypes.data_as(_dp),
                               y.ctypes.data_as(_dp))
        return y[:M]
