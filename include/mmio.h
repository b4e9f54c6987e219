/*
 * mmio.h -- minimal Matrix Market banner/size reader.
 *
 * Own restatement of the subset of the NIST "Matrix Market I/O library for
 * ANSI C" interface that the SpMV path needs (the reference vendors the NIST
 * library as libs/mmio.{h,c}; reference use sites: src/matrix_parser.c:33-54).
 * Only the interface (names, typecode letters, error codes) is shared with
 * the NIST library; the implementation in csrc/host/mmio.c is new.
 *
 * MM_typecode is a 4-character code:
 *   [0] object   'M' matrix
 *   [1] format   'C' coordinate | 'A' array
 *   [2] field    'R' real | 'C' complex | 'P' pattern | 'I' integer
 *   [3] symmetry 'G' general | 'S' symmetric | 'H' hermitian | 'K' skew
 */
#ifndef SPMV_AMD_MMIO_H
#define SPMV_AMD_MMIO_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_MAX_LINE_LENGTH 1025
#define MM_MAX_TOKEN_LENGTH 64
#define MatrixMarketBanner "%%MatrixMarket"

typedef char MM_typecode[4];

/* error codes (same numeric values as the NIST library) */
#define MM_COULD_NOT_READ_FILE 11
#define MM_PREMATURE_EOF 12
#define MM_NOT_MTX 13
#define MM_NO_HEADER 14
#define MM_UNSUPPORTED_TYPE 15
#define MM_LINE_TOO_LONG 16
#define MM_COULD_NOT_WRITE_FILE 17

/* queries */
#define mm_is_matrix(t) ((t)[0] == 'M')
#define mm_is_sparse(t) ((t)[1] == 'C')
#define mm_is_coordinate(t) ((t)[1] == 'C')
#define mm_is_dense(t) ((t)[1] == 'A')
#define mm_is_array(t) ((t)[1] == 'A')
#define mm_is_complex(t) ((t)[2] == 'C')
#define mm_is_real(t) ((t)[2] == 'R')
#define mm_is_pattern(t) ((t)[2] == 'P')
#define mm_is_integer(t) ((t)[2] == 'I')
#define mm_is_symmetric(t) ((t)[3] == 'S')
#define mm_is_general(t) ((t)[3] == 'G')
#define mm_is_skew(t) ((t)[3] == 'K')
#define mm_is_hermitian(t) ((t)[3] == 'H')

/* setters take a pointer to the typecode, as in the NIST interface */
#define mm_set_matrix(t) ((*(t))[0] = 'M')
#define mm_set_coordinate(t) ((*(t))[1] = 'C')
#define mm_set_sparse(t) ((*(t))[1] = 'C')
#define mm_set_array(t) ((*(t))[1] = 'A')
#define mm_set_dense(t) ((*(t))[1] = 'A')
#define mm_set_complex(t) ((*(t))[2] = 'C')
#define mm_set_real(t) ((*(t))[2] = 'R')
#define mm_set_pattern(t) ((*(t))[2] = 'P')
#define mm_set_integer(t) ((*(t))[2] = 'I')
#define mm_set_symmetric(t) ((*(t))[3] = 'S')
#define mm_set_general(t) ((*(t))[3] = 'G')
#define mm_set_skew(t) ((*(t))[3] = 'K')
#define mm_set_hermitian(t) ((*(t))[3] = 'H')
#define mm_clear_typecode(t) \
    ((*(t))[0] = (*(t))[1] = (*(t))[2] = ' ', (*(t))[3] = 'G')
#define mm_initialize_typecode(t) mm_clear_typecode(t)

int mm_is_valid(MM_typecode matcode);
int mm_read_banner(FILE *f, MM_typecode *matcode);
int mm_read_mtx_crd_size(FILE *f, int *M, int *N, int *nz);
int mm_write_banner(FILE *f, MM_typecode matcode);
int mm_write_mtx_crd_size(FILE *f, int M, int N, int nz);
/* returns a malloc'd string the caller frees (NIST convention) */
char *mm_typecode_to_str(MM_typecode matcode);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_AMD_MMIO_H */
