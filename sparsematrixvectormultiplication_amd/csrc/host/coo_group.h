/*
 * coo_group.h -- internal to the host layer: group COO triplets by row.
 *
 * Both format builders start the same way (reference: src/csr_matrix.c:85-112 histogram ->
 * scan -> scatter, src/hll_matrix.c:60-118 the same with per-row mallocs): entries bucketed by
 * row, FILE ORDER kept inside a row (the tie rules of both builders are defined on that order).
 * The reference does it with one serial pass of random writes; here it is a two-level counting
 * sort that all threads take part in and whose second level works on cache-sized row ranges.
 */
#ifndef SPMV_AMD_COO_GROUP_H
#define SPMV_AMD_COO_GROUP_H

#include <stddef.h>

/* row_off[M + 1] (row_off[r] = first slot of row r, row_off[M] = nz), cols[nz], vals[nz]; all
 * caller-allocated.  N >= 0: every index is checked against [0, M) x [0, N) and the first offending
 * pair is reported through *bad_row / *bad_col with return value -2.  -1: out of memory. */
int coo_group_by_row(int M, int N, size_t nz, const int *I, const int *J, const double *val, int *row_off,
                     int *cols, double *vals, int *bad_row, int *bad_col);

#endif
