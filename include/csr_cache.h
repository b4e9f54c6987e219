/*
 * csr_cache.h -- binary sidecar of a built CSR matrix.
 *
 * The reference re-parses the Matrix Market text and rebuilds CSR on every run
 * (src/matrix_parser.c:25-150 -> src/csr_matrix.c:63-126); for nlpkkt120 that is
 * ~50 M text lines before the first SpMV.  The sidecar stores the finished
 * CSRMatrix (libs/csr_matrix.h:8-16 layout: int row_ptr / col_idx, double
 * values, MM typecode) next to the .mtx so later runs read three flat arrays.
 *
 * All functions: 0 on success, -1 on failure (message on stdout, as the rest of
 * the host layer).  A sidecar is only trusted when its header, its recorded
 * source-file size/mtime and the checksums of all three arrays match and the
 * structure is a valid CSR matrix; anything else is treated as "no cache".
 */
#ifndef SPMV_AMD_CSR_CACHE_H
#define SPMV_AMD_CSR_CACHE_H

#include "csr_matrix.h"

#ifdef __cplusplus
extern "C" {
#endif

/* write csr to path; source_mtx may be NULL (then no staleness stamp is recorded) */
int save_csr_binary(const CSRMatrix *csr, const char *path, const char *source_mtx);
/* read path into csr (caller frees with free_csr_matrix); if source_mtx is not NULL the
 * sidecar must carry that file's current size and mtime */
int load_csr_binary(const char *path, CSRMatrix *csr, const char *source_mtx);
/* "<mtx_path>.csrbin" if it is valid and fresh, else parse + convert_in_csr + (try to) write it.
 * *from_cache (may be NULL) tells which of the two happened. */
int load_csr_cached(const char *mtx_path, CSRMatrix *csr, int *from_cache);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_AMD_CSR_CACHE_H */
