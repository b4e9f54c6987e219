"""ctypes binding of libspmv_amd.so (C host layer + HIP kernels + C-ABI).

The structures mirror include/*.h field for field (which in turn keep the
reference's layouts: libs/matrix_parser.h:6-14, libs/csr_matrix.h:8-16,
libs/hll_matrix.h:15-27, libs/performance_calculate.h:33-37).

There is no Python or CPU fallback: if the shared library is missing, loading
raises, and every GPU entry point raises SpmvHipError when the C-ABI returns -1.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspmv_amd.so")

c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)


class PreMatrix(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("nz", C.c_int), ("I", c_int_p), ("J", c_int_p),
                ("val", c_double_p), ("type", C.c_char * 4)]


class CSRMatrix(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("nz", C.c_int), ("row_ptr", c_int_p),
                ("col_idx", c_int_p), ("values", c_double_p), ("type", C.c_char * 4)]


class ELLPACKBlock(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("MAXNZ", C.c_int), ("JA", c_int_p),
                ("AS", c_double_p)]


class HLLMatrix(C.Structure):
    _fields_ = [("num_blocks", C.c_int), ("blocks", C.POINTER(ELLPACKBlock))]


class DiffMetrics(C.Structure):
    _fields_ = [("mean_abs_err", C.c_double), ("mean_rel_err", C.c_double),
                ("significant_diffs", C.c_int)]


class DevInfo(C.Structure):
    _fields_ = [("M_local", C.c_int), ("M_total", C.c_int), ("N", C.c_int), ("row0", C.c_int),
                ("nz", C.c_longlong), ("value_bytes", C.c_int), ("auto_variant", C.c_int),
                ("lanes_per_row", C.c_int), ("stream_blocks", C.c_int), ("long_rows", C.c_int),
                ("slots", C.c_longlong), ("hacks", C.c_int), ("algo_bytes", C.c_longlong),
                ("device_bytes", C.c_longlong), ("local_blocks", C.c_int), ("local_stage_lines", C.c_int),
                ("local_lines", C.c_longlong), ("stream_bytes", C.c_longlong), ("stream_kernel", C.c_int),
                ("tile_blocks", C.c_int), ("tile_passes", C.c_int), ("tile_split_rows", C.c_int),
                ("tile_entries", C.c_longlong), ("tile_staged_entries", C.c_longlong),
                ("tile_long_rows", C.c_int), ("tile_long_items", C.c_int), ("tile_long_entries", C.c_longlong),
                ("tile_staged_cols", C.c_longlong), ("tile_remainder_entries", C.c_longlong),
                ("tile_mid_rows", C.c_int), ("tile_mid_items", C.c_int), ("tile_mid_entries", C.c_longlong),
                ("place_tries", C.c_int), ("place_first_us", C.c_float), ("place_best_us", C.c_float),
                ("val_address", C.c_ulonglong), ("tile_expanded_entries", C.c_longlong),
                ("pattern_slots", C.c_longlong), ("pattern_with_us", C.c_float), ("pattern_without_us", C.c_float)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


# every symbol include/*.h declares and the product library defines
_PROTOTYPES = {
    # mmio.h
    "mm_read_banner": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mm_read_mtx_crd_size": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_int_p]),
    "mm_write_banner": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mm_write_mtx_crd_size": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mm_is_valid": (C.c_int, [C.c_char_p]),
    "mm_typecode_to_str": (C.c_void_p, [C.c_char_p]),
    # matrix_parser.h
    "init_pre_matrix": (None, [C.POINTER(PreMatrix)]),
    "free_pre_matrix": (None, [C.POINTER(PreMatrix)]),
    "read_matrix_market": (C.c_int, [C.c_char_p, C.POINTER(PreMatrix)]),
    "print_pre_matrix": (None, [C.POINTER(PreMatrix), C.c_bool]),
    # csr_matrix.h (product-side symbols)
    "init_csr_matrix": (None, [C.POINTER(CSRMatrix)]),
    "free_csr_matrix": (None, [C.POINTER(CSRMatrix)]),
    "convert_in_csr": (C.c_int, [C.POINTER(PreMatrix), C.POINTER(CSRMatrix), C.c_char_p]),
    "save_csr_binary": (C.c_int, [C.POINTER(CSRMatrix), C.c_char_p, C.c_char_p]),
    "load_csr_binary": (C.c_int, [C.c_char_p, C.POINTER(CSRMatrix), C.c_char_p]),
    "load_csr_cached": (C.c_int, [C.c_char_p, C.POINTER(CSRMatrix), C.POINTER(C.c_int)]),
    "print_csr_matrix": (None, [C.POINTER(CSRMatrix)]),
    "write_memory_stats_to_csv": (None, [C.c_char_p, C.c_int, C.c_size_t]),
    "prepare_thread_distribution": (C.c_int, [C.c_int, c_int_p, C.c_int, C.c_longlong,
                                              C.POINTER(c_int_p), C.POINTER(c_int_p)]),
    # hll_matrix.h (product-side symbols)
    "init_hll_matrix": (None, [C.POINTER(HLLMatrix)]),
    "convert_to_hll": (C.c_int, [C.POINTER(PreMatrix), C.POINTER(HLLMatrix)]),
    "free_hll_matrix": (None, [C.POINTER(HLLMatrix)]),
    "printHLLMatrix": (None, [C.POINTER(HLLMatrix)]),
    "prepare_thread_distribution_hll": (C.c_int, [C.POINTER(HLLMatrix), C.c_int,
                                                  C.POINTER(c_int_p), C.POINTER(c_int_p)]),
    # performance_calculate.h
    "computeDifferenceMetrics": (DiffMetrics, [c_double_p, c_double_p, C.c_int, C.c_double,
                                               C.c_double, C.c_bool]),
    "computeDifferenceMetricsGpu": (DiffMetrics, [c_double_p, c_double_p, C.c_int, C.c_double,
                                                  C.c_bool]),
    "initialize_metrics": (None, []),
    "cleanup_metrics": (None, []),
    "get_metric_value": (C.c_double, [C.c_int]),
    "get_relative_error": (C.c_double, [C.c_int]),
    "get_absolute_error": (C.c_double, [C.c_int]),
    "update_medium_metric": (None, [C.c_int, C.c_double]),
    "reset_medium_time_metrics": (None, []),
    "computeAverageErrors": (DiffMetrics, [C.c_int]),
    "accumulateErrors": (None, [C.POINTER(DiffMetrics), C.c_int]),
    "calculate_flops": (C.c_double, [C.c_int, C.c_double]),
    "print_flops": (None, [C.c_double]),
    "get_metric_stddev": (C.c_double, [C.c_int]),
    "get_metric_variance": (C.c_double, [C.c_int]),
    "get_metric_min": (C.c_double, [C.c_int]),
    "get_metric_median": (C.c_double, [C.c_int]),
    # utility.h
    "init_vector_at_one": (None, [c_double_p, C.c_int]),
    "write_results_to_csv": (None, None),
    "write_results_to_csv_gpu": (None, None),
    "write_block_result_to_csv": (None, [C.c_char_p, C.c_int] + [C.c_int] * 6 + [C.c_char_p]),
    "swap": (None, [c_int_p, c_int_p]),
    "swap_double": (None, [c_double_p, c_double_p]),
    "partition": (C.c_size_t, [c_int_p, c_double_p, C.c_size_t, C.c_size_t]),
    "sort_row": (None, [c_int_p, c_double_p, C.c_size_t, C.c_size_t]),
    "clear_cache": (None, [C.c_size_t]),
    "create_directory": (None, [C.c_char_p]),
    "process_matrix_file": (C.c_int, [C.c_char_p, C.POINTER(PreMatrix)]),
    # synth_matrix.h
    "synth_kkt_rows": (C.c_int, [C.c_int] * 3),
    "synth_kkt_row_ptr": (C.c_int, [C.c_int] * 3 + [c_int_p]),
    "synth_kkt_fill": (C.c_int, [C.c_int] * 3 + [C.c_ulonglong, C.c_int, C.c_int, c_int_p, c_int_p,
                                                 c_double_p]),
    "synth_fem_rows": (C.c_int, [C.c_int] * 3),
    "synth_fem_row_ptr": (C.c_int, [C.c_int] * 3 + [c_int_p]),
    "synth_fem_fill": (C.c_int, [C.c_int] * 3 + [C.c_ulonglong, C.c_int, C.c_int, c_int_p, c_int_p,
                                                 c_double_p]),
    "synth_powerlaw_row_ptr": (C.c_int, [C.c_int, C.c_int, C.c_ulonglong, c_int_p]),
    "synth_powerlaw_fill": (C.c_int, [C.c_int, C.c_int, C.c_ulonglong, C.c_int, C.c_int, c_int_p,
                                      c_int_p, c_float_p]),
    # spmv_hip.h
    "spmv_hip_device_count": (C.c_int, []),
    "spmv_hip_init": (C.c_int, [C.c_int]),
    "spmv_hip_shutdown": (C.c_int, []),
    "spmv_hip_sync": (C.c_int, []),
    "spmv_hip_stream": (C.c_void_p, []),
    "spmv_hip_last_error": (C.c_char_p, []),
    "spmv_hip_device_name": (C.c_int, [C.c_char_p, C.c_size_t, c_int_p, C.POINTER(C.c_longlong)]),
    "spmv_hip_flush_cache": (C.c_int, [C.c_size_t]),
    "spmv_hip_device_state": (C.c_int, [C.c_char_p, C.c_size_t]),
    "spmv_hip_stream_probe": (C.c_int, [C.c_size_t, C.c_int, C.c_int, c_float_p, c_float_p]),
    "spmv_hip_stream_probe_at": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, c_float_p, c_float_p]),
    "spmv_hip_gather_probe": (C.c_int, [C.c_int, C.c_size_t, C.c_int, c_double_p]),
    "spmv_hip_csr_tile_digest": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "spmv_hip_hll_tile_digest": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "spmv_hip_csr_addresses": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "spmv_hip_csr_stamp_blocks": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]),
    "spmv_hip_csr_relocate": (C.c_int, [C.c_void_p, C.c_int, C.c_ulonglong, C.c_ulonglong]),
    "spmv_hip_csr_relocate_vmm": (C.c_int, [C.c_void_p, C.c_int, C.c_ulonglong, C.c_ulonglong]),
    "spmv_hip_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "spmv_hip_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "spmv_hip_free": (C.c_int, [C.c_void_p]),
    "spmv_hip_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "spmv_hip_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "spmv_hip_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "spmv_hip_csr_upload": (C.c_int, [C.c_int, C.c_int, c_int_p, c_int_p, c_double_p, C.c_int,
                                      C.c_int, C.POINTER(C.c_void_p)]),
    "spmv_hip_csr_upload_f32": (C.c_int, [C.c_int, C.c_int, c_int_p, c_int_p, c_float_p, C.c_int,
                                          C.c_int, C.POINTER(C.c_void_p)]),
    "spmv_hip_csr_upload_matrix": (C.c_int, [C.POINTER(CSRMatrix), C.POINTER(C.c_void_p)]),
    "spmv_hip_csr_free": (None, [C.c_void_p]),
    "spmv_hip_csr_info": (C.c_int, [C.c_void_p, C.POINTER(DevInfo)]),
    "spmv_hip_csr_set_x": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spmv_hip_csr_run": (C.c_int, [C.c_void_p, C.c_int]),
    "spmv_hip_csr_get_y": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spmv_hip_csr_x_ptr": (C.c_void_p, [C.c_void_p]),
    "spmv_hip_csr_y_ptr": (C.c_void_p, [C.c_void_p]),
    "spmv_hip_csr_run_on": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spmv_hip_csr_time": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p]),
    "spmv_hip_csr_step_time": (C.c_int, [C.c_void_p, C.c_int, c_int_p, C.c_int, C.c_int, c_float_p,
                                         c_float_p]),
    "spmv_hip_hll_upload": (C.c_int, [C.POINTER(HLLMatrix), C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]),
    "spmv_hip_csr_time_graph": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_float_p]),
    "spmv_hip_hll_time_graph": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_float_p]),
    "spmv_hip_hll_x_ptr": (C.c_void_p, [C.c_void_p]),
    "spmv_hip_hll_y_ptr": (C.c_void_p, [C.c_void_p]),
    "spmv_hip_hll_from_csr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "spmv_hip_hll_download": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), c_int_p, c_int_p, c_double_p]),
    "spmv_hip_hll_free": (None, [C.c_void_p]),
    "spmv_hip_hll_info": (C.c_int, [C.c_void_p, C.POINTER(DevInfo)]),
    "spmv_hip_hll_set_x": (C.c_int, [C.c_void_p, c_double_p]),
    "spmv_hip_hll_run": (C.c_int, [C.c_void_p, C.c_int]),
    "spmv_hip_hll_get_y": (C.c_int, [C.c_void_p, c_double_p]),
    "spmv_hip_hll_run_on": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spmv_hip_hll_time": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p]),
    "spmv_hip_partition_rows": (C.c_int, [C.c_int, c_int_p, C.c_int, c_int_p]),
    "spmv_hip_comm_scatter_staged": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p]),
    "spmv_hip_comm_autotune": (C.c_int, [C.c_void_p, c_int_p, C.c_int, C.c_int, c_int_p, c_float_p]),
    "spmv_hip_hll_plan_check": (C.c_int, [C.POINTER(HLLMatrix), C.c_int, C.c_int, c_int_p]),
    "spmv_hip_csr_from_coo": (C.c_int, [C.c_int, C.c_int, C.c_longlong, c_int_p, c_int_p, c_double_p,
                                        C.POINTER(C.c_void_p)]),
    "spmv_hip_csr_download": (C.c_int, [C.c_void_p, c_int_p, c_int_p, C.c_void_p]),
    "spmv_hip_csr_plan_check": (C.c_int, [C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_int_p]),
    "spmv_hip_csr_tile_plan_check": (C.c_int, [C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.POINTER(C.c_longlong)]),
    "spmv_hip_csr_tile_auto_plan": (C.c_int, [C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, C.POINTER(C.c_longlong)]),
    "spmv_hip_csr_power_iterate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_int_p, C.c_int, c_double_p,
                                             c_float_p]),
    "spmv_hip_csr_cg": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_int_p, C.c_int, C.c_void_p, C.c_void_p, c_double_p,
                                  c_float_p]),
    "spmv_hip_csr_needed_ranges": (C.c_int, [C.c_void_p, C.c_int, c_int_p, c_int_p]),
    "spmv_hip_csr_split_interior": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    "spmv_hip_csr_split_columns": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_longlong)]),
    "spmv_hip_csr_run_split": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spmv_hip_csr_run_part": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spmv_hip_halo_plan": (C.c_int, [C.c_int, C.c_int, c_int_p, c_int_p, c_int_p, C.c_int, C.c_int, c_int_p, c_int_p,
                                     c_int_p, c_int_p]),
    "spmv_hip_comm_halo_setup": (C.c_int, [C.c_void_p, c_int_p]),
    "spmv_hip_comm_halo_exchange": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "spmv_hip_comm_halo_info": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), c_int_p]),
    "spmv_hip_csr_power_iterate_halo": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, c_float_p]),
    "spmv_hip_partition_hacks": (C.c_int, [C.POINTER(HLLMatrix), C.c_int, c_int_p]),
    "spmv_hip_hll_upload_part": (C.c_int, [C.POINTER(HLLMatrix), C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_void_p)]),
    "spmv_hip_hll_step_time": (C.c_int, [C.c_void_p, C.c_int, c_int_p, C.c_int, C.c_int, c_float_p,
                                         c_float_p]),
    "spmv_hip_comm_get_id": (C.c_int, [C.c_void_p]),
    "spmv_hip_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "spmv_hip_comm_destroy": (C.c_int, []),
    "spmv_hip_comm_info": (C.c_int, [c_int_p, c_int_p]),
    "spmv_hip_comm_allgatherv": (C.c_int, [C.c_void_p, c_int_p, C.c_int, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(sorted(_PROTOTYPES))
COMM_ID_BYTES = 128

_lib = None


def lib() -> C.CDLL:
    """Load libspmv_amd.so once.  Raises OSError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C sparsematrixvectormultiplication_amd/csrc`. "
                "There is no Python/CPU fallback for the SpMV path.")
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError = header/library drift, fail loudly
            fn.restype = restype
            if argtypes is not None:
                fn.argtypes = argtypes
        _lib = handle
    return _lib
