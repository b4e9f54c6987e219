"""MI355X-native SpMV engine (CSR + hacked-ELLPACK) -- Python host mirror.

The product is libspmv_amd.so: a plain-C host layer behind the reference's
header surface (include/csr_matrix.h, hll_matrix.h, performance_calculate.h,
matrix_parser.h, utility.h) plus hand-written HIP kernels for gfx950 behind the
C-ABI of include/spmv_hip.h.  This package is a ctypes mirror of that API for
tests and bench.py; it contains no arithmetic of its own and no CPU fallback.
"""
from ._native import EXPORTED_SYMBOLS, LIB_PATH, lib  # noqa: F401
from .host import (HACK_SIZE, ITERATION_SKIP, CsrHost, HllHost, PreMatrix,  # noqa: F401
                   calculate_flops, compute_difference_metrics, compute_difference_metrics_gpu,
                   convert_in_csr, convert_to_hll, init_vector_at_one, partition_rows,
                   prepare_thread_distribution, prepare_thread_distribution_hll,
                   read_matrix_market, csr_plan_check, csr_tile_plan_check, csr_tile_auto_plan, hll_plan_check, partition_hacks, hack_bounds_to_rows, save_csr_binary, load_csr_binary, load_csr_cached)
from .device import (CSR_AUTO, CSR_STREAM, CSR_SUBWAVE, CSR_THREAD_ROW, CSR_VARIANTS,  # noqa: F401
                     CSR_WAVE_ROW, HLL_AUTO, HLL_LDS, HLL_SUBWAVE, HLL_THREAD_ROW, HLL_VARIANTS,
                     CsrDevice, HllDevice, SpmvHipError, device_count, device_name, flush_cache,
                     hip_init, hip_stream, hip_sync, box_state, stream_probe, stream_probe_at, gather_probe, set_tuning)
