/*
 * matrix_parser.h -- Matrix Market coordinate file -> COO triplets.
 *
 * Kept API surface of the reference's libs/matrix_parser.h:6-19 (struct
 * layout, names, argument order, 0 / -1 return convention).  Implementation:
 * csrc/host/matrix_parser.c (new code, same semantics as the reference's
 * src/matrix_parser.c:25-150).
 */
#ifndef SPMV_AMD_MATRIX_PARSER_H
#define SPMV_AMD_MATRIX_PARSER_H

#include <stdbool.h>

#include "mmio.h"

#ifdef __cplusplus
extern "C" {
#endif

/* COO matrix as read from file: 0-based, symmetric files already expanded,
 * entries in file order (a mirrored entry directly follows its original). */
typedef struct {
    int M;            /* rows */
    int N;            /* columns */
    int nz;           /* stored entries after symmetric expansion */
    int *I;           /* row index of each entry */
    int *J;           /* column index of each entry */
    double *val;      /* value of each entry (1.0 for pattern files) */
    MM_typecode type; /* Matrix Market typecode of the source file */
} PreMatrix;

void init_pre_matrix(PreMatrix *mat);
void free_pre_matrix(PreMatrix *mat);
/* 0 on success, -1 on any failure (message on stdout, as the reference). */
int read_matrix_market(const char *filename, PreMatrix *mat);
void print_pre_matrix(PreMatrix *mat, const bool full_print);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_AMD_MATRIX_PARSER_H */
