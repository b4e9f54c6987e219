/*
 * csr_matrix.h -- CSR container, COO->CSR builder, nnz-balanced row
 * partitioner, and the CPU SpMV entry points.
 *
 * Kept API surface of the reference's libs/csr_matrix.h:8-33 (identical
 * struct layout and signatures).  Where each symbol lives in this repo:
 *
 *   init/free/convert_in_csr/print, prepare_thread_distribution,
 *   write_memory_stats_to_csv      -> csrc/host/csr_matrix.c   (product, C)
 *   csr_matrix_vector_mult (K1, the ORACLE), spvm_csr_parallel (K2),
 *   spvm_csr_parallel_simd (K3)    -> oracle/cpu_spmv.c        (checker and
 *                                     CPU baseline only; never linked into
 *                                     the product library)
 *   GPU SpMV                       -> include/spmv_hip.h       (product, HIP)
 *
 * The product GPU path has no CPU fallback: the three CPU kernels are test
 * infrastructure (oracle + reported CPU baseline), see DESIGN.md.
 */
#ifndef SPMV_AMD_CSR_MATRIX_H
#define SPMV_AMD_CSR_MATRIX_H

#include <stddef.h>

#include "matrix_parser.h"
#include "mmio.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 0-based CSR; column indices ascending inside each row. */
typedef struct {
    int M;            /* rows */
    int N;            /* columns */
    int nz;           /* stored entries */
    int *row_ptr;     /* [M+1] */
    int *col_idx;     /* [nz]  */
    double *values;   /* [nz]  */
    MM_typecode type; /* typecode of the source file */
} CSRMatrix;

void init_csr_matrix(CSRMatrix *mat);
void free_csr_matrix(CSRMatrix *mat);
/* COO -> CSR (reference: src/csr_matrix.c:63-126). 0 / -1. matrix_name is
 * accepted for signature parity and unused, as in the reference. */
int convert_in_csr(const PreMatrix *pre, CSRMatrix *csr, const char *matrix_name);
void print_csr_matrix(const CSRMatrix *mat);
void write_memory_stats_to_csv(const char *matrix_name, int nz, size_t total_memory_bytes);

/* K1 -- serial CSR SpMV, ACCUMULATES into caller-zeroed y
 * (reference: src/csr_matrix.c:130-139).  Defined in oracle/cpu_spmv.c. */
void csr_matrix_vector_mult(int num_row, const int *row_ptr, const int *col_idx,
                            const double *values, const double *x, double *y);

/* K2 -- OpenMP CSR SpMV over per-thread row ranges, overwrites y
 * (reference: src/csr_matrix.c:294-313).  Defined in oracle/cpu_spmv.c. */
void spvm_csr_parallel(const int *row_ptr, const int *col_idx, const double *values,
                       const double *x, double *y, int num_threads,
                       const int *thread_row_start, const int *thread_row_end);

/* K4 -- greedy contiguous nnz-balanced row split (reference:
 * src/csr_matrix.c:167-266).  Returns the number of non-empty chunks (0 on
 * failure) and mallocs the two arrays for the caller to free; end is
 * exclusive.  Also the template for the multi-GPU row split. */
int prepare_thread_distribution(const int num_row, const int *row_ptr, int num_threads,
                                const long long total_nnz, int **thread_row_start,
                                int **thread_row_end);

/* K3 -- K2 with an `omp simd` reduction inner loop
 * (reference: src/csr_matrix.c:269-292).  Defined in oracle/cpu_spmv.c. */
void spvm_csr_parallel_simd(const int *row_ptr, const int *col_idx, const double *values,
                            const double *x, double *y, int num_threads,
                            const int *thread_row_start, const int *thread_row_end);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_AMD_CSR_MATRIX_H */
