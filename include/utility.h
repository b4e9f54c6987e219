/*
 * utility.h -- small host helpers shared by the drivers.
 *
 * Kept API surface of the reference's libs/utility.h:7-31 plus the two CSV
 * writers of its CUDA twin (cuda_libs/utility.cuh:20-40), renamed with a
 * _gpu suffix because C cannot overload write_results_to_csv.
 * Implementation: csrc/host/utility.c.
 *
 * Deliberate differences from the reference:
 *   - create_directory() creates the directory if missing and NEVER deletes
 *     what is in it (the reference wipes every earlier result file,
 *     src/utility.c:200-209);
 *   - clear_cache() is sized by the caller; the GPU-side flush lives in
 *     spmv_hip_flush_cache() (include/spmv_hip.h).
 */
#ifndef SPMV_AMD_UTILITY_H
#define SPMV_AMD_UTILITY_H

#include <stddef.h>

#include "matrix_parser.h"
#include "performance_calculate.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ITERATION_SKIP 5
#define FREE_CHECK(ptr)       \
    do {                      \
        if ((ptr) != NULL) {  \
            free(ptr);        \
            (ptr) = NULL;     \
        }                     \
    } while (0)

/* x := 1.0 -- the reference's only input vector (src/utility.c:18-22). */
void init_vector_at_one(double *v, const int size);

/* CPU-build result row (schema of src/utility.c:95-138, unchanged). */
void write_results_to_csv(
    const char *matrix_name, const int num_rows, const int num_cols, const int nz,
    const int num_threads, const double time_serial, const double time_serial_hll,
    const double time_parallel, const double time_parallel_simd, const double time_parallel_hll,
    const double time_parallel_hll_simd, DiffMetrics error_csr, DiffMetrics error_hll,
    DiffMetrics error_csr_simd, DiffMetrics error_hll_simd, const double speedup_parallel,
    const double speedup_simd, const double speedup_hll, const double speedup_hll_simd,
    const double efficiency_parallel, const double efficiency_simd, const double efficiency_hll,
    const double efficiency_hll_simd, const double flops_serial,
    const double avg_flops_hll_serial, const double flops_parallel,
    const double flops_parallel_simd, const double flops_parallel_hll,
    const double flops_parallel_hll_simd, const char *output_file);

/* GPU-build result row (schema of cuda_src/utility.cu:94-136, unchanged;
 * "row" = thread-per-row, "warp" = wavefront-per-row on this hardware). */
void write_results_to_csv_gpu(
    const char *matrix_name, const int num_rows, const int num_cols, const int nz,
    const double time_serial, const double time_serial_hll, const double time_row_csr,
    const double time_warp_csr, const double time_warp_csr_shared,
    const double time_warp_shared_hll, const double time_row_hll, const double time_warp_hll,
    const double flops_serial, const double avg_flops_hll_serial, const double flops_row_csr,
    const double flops_warp_csr, const double flops_row_hll, const double flops_warp_hll,
    const double flops_warp_csr_shared, const double flops_warp_shared_hll,
    DiffMetrics mediumCsrParallel, DiffMetrics mediumCsrWarp, DiffMetrics mediumCsrWarpShared,
    DiffMetrics mediumHllNaive, DiffMetrics mediumHllWarp, DiffMetrics mediumHllWarpShared,
    const char *output_file);

/* GPU-build launch-shape row (schema of cuda_src/utility.cu:236-261). */
void write_block_result_to_csv(const char *matrix_name, const int nz, int block_size_csr_row,
                               int block_size_csr_warp, int block_size_csr_shared,
                               int block_size_hll_row, int block_size_hll_warp,
                               int block_size_hll_shared, const char *output_file);

/* paired quicksort of (col_idx, values)[low..high], both bounds inclusive
 * (reference: src/utility.c:38-91) */
void swap(int *a, int *b);
void swap_double(double *a, double *b);
size_t partition(int *col_idx, double *values, size_t low, size_t high);
void sort_row(int *col_idx, double *values, size_t low, size_t high);

void clear_cache(size_t clear_size_mb);
void create_directory(const char *path);
int process_matrix_file(const char *filepath, PreMatrix *pre_mat);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_AMD_UTILITY_H */
