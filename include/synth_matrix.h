/*
 * synth_matrix.h -- seeded, shape-matched stand-ins for the benchmark
 * matrices that cannot be shipped (no SuiteSparse files and no network in the
 * build / GPU boxes).  Bench/test workload tooling: the
 * reference has only a tiny Python generator (src/matrix_generator.py) that is
 * not on the hot path; nothing here mirrors it.
 *
 * Every generator is stateless (entry values are a hash of seed, row, column),
 * so any row range can be produced independently -- each GPU rank builds only
 * its own row block -- and symmetric positions carry equal values.
 * Columns are ascending inside each row, as convert_in_csr leaves them.
 *
 * Usage: <kind>_row_ptr() fills row_ptr[0..M] for the whole matrix (cheap),
 * then <kind>_fill() writes col_idx / values for rows [row0, row1) at
 * positions row_ptr[r] - row_ptr[row0].
 */
#ifndef SPMV_AMD_SYNTH_MATRIX_H
#define SPMV_AMD_SYNTH_MATRIX_H

#ifdef __cplusplus
extern "C" {
#endif

/* nlpkkt-like: symmetric KKT-shaped [H B^T; B D] over an nx*ny*nz grid;
 * H, D = 13-point stencils, B = 15-point stencil => M = N = 2*nx*ny*nz, at most
 * 28 entries per row (interior rows exactly 28, mean ~26.9 at 120x120x123
 * which gives M = 3 542 400 like SuiteSparse nlpkkt120). */
int synth_kkt_rows(int nx, int ny, int nz);
int synth_kkt_row_ptr(int nx, int ny, int nz, int *row_ptr);
int synth_kkt_fill(int nx, int ny, int nz, unsigned long long seed, int row0, int row1,
                   const int *row_ptr, int *col_idx, double *values);

/* cant-like: symmetric FEM-shaped matrix, 3 unknowns per node of a gx*gy*gz
 * node grid, 27-point node stencil => M = N = 3*gx*gy*gz, at most 81 entries
 * per row (9 x 9 x 257 gives M = 62 451 like SuiteSparse cant). */
int synth_fem_rows(int gx, int gy, int gz);
int synth_fem_row_ptr(int gx, int gy, int gz, int *row_ptr);
int synth_fem_fill(int gx, int gy, int gz, unsigned long long seed, int row0, int row1,
                   const int *row_ptr, int *col_idx, double *values);

/* power-law: M = N = n, row degrees ~ 1.08/u clipped to [1, max_degree]
 * (Zipf-like, alpha = 2), columns half preferential (density ~ 1/sqrt(c)),
 * half uniform; a column may repeat inside a row (legal CSR).  fp32 values. */
int synth_powerlaw_row_ptr(int n, int max_degree, unsigned long long seed, int *row_ptr);
int synth_powerlaw_fill(int n, int max_degree, unsigned long long seed, int row0, int row1,
                        const int *row_ptr, int *col_idx, float *values);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_AMD_SYNTH_MATRIX_H */
