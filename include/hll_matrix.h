/*
 * hll_matrix.h -- HLL (hacked ELLPACK, 32-row hacks) container, COO->HLL
 * builder, hack partitioner, and the CPU HLL SpMV entry points.
 *
 * Kept API surface of the reference's libs/hll_matrix.h:12-40 (identical
 * struct layout and signatures).  Where each symbol lives:
 *
 *   init/convert_to_hll/free/printHLLMatrix,
 *   prepare_thread_distribution_hll  -> csrc/host/hll_matrix.c (product, C)
 *   spmv_hll_serial (K5), spmv_hll (K6), spmv_hll_simd (K7)
 *                                    -> oracle/cpu_spmv.c      (checker and
 *                                       CPU baseline only)
 *   GPU SpMV                         -> include/spmv_hip.h     (product, HIP)
 *
 * save_hll_memory_stats is declared by the reference (hll_matrix.h:40) and
 * defined nowhere; it stays declared-only here too.
 */
#ifndef SPMV_AMD_HLL_MATRIX_H
#define SPMV_AMD_HLL_MATRIX_H

#include <stddef.h>

#include "matrix_parser.h"
#include "mmio.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HACK_SIZE 32

/* One hack: M rows (32, or M mod 32 for the last), each padded to MAXNZ
 * slots, ROW-MAJOR: slot (i, j) is at i * MAXNZ + j.  Padding slots carry
 * AS = 0.0 and JA = the row's last valid column (0 for an empty row), so a
 * kernel may run j over [0, MAXNZ) without a bounds test.  MAXNZ == 0 gives
 * JA == AS == NULL.  (reference: src/hll_matrix.c:81-126,216-247) */
typedef struct {
    int M;
    int N;
    int MAXNZ;
    int *JA;
    double *AS;
} ELLPACKBlock;

typedef struct {
    int num_blocks;
    ELLPACKBlock *blocks;
} HLLMatrix;

void init_hll_matrix(HLLMatrix *hll);
int convert_to_hll(const PreMatrix *pre, HLLMatrix *hll);
void free_hll_matrix(HLLMatrix *hll);
void printHLLMatrix(HLLMatrix *hll);

/* K5 (reference: src/hll_matrix.c:286-308).  Defined in oracle/cpu_spmv.c. */
void spmv_hll_serial(int num_blocks, const ELLPACKBlock *blocks, const double *x, double *y);

/* K8 -- greedy contiguous split over hacks, weight = padded slots
 * (reference: src/hll_matrix.c:410-540). */
int prepare_thread_distribution_hll(const HLLMatrix *matrix, int num_threads,
                                    int **thread_block_start, int **thread_block_end);

/* K6 / K7 (reference: src/hll_matrix.c:376-408 / 339-374).
 * Defined in oracle/cpu_spmv.c. */
void spmv_hll(const ELLPACKBlock *blocks, const double *x, double *y, int num_threads,
              int const *thread_block_start, int const *thread_block_end);
void spmv_hll_simd(const ELLPACKBlock *blocks, const double *x, double *y, int num_threads,
                   int const *thread_block_start, int const *thread_block_end);

int save_hll_memory_stats(const HLLMatrix *hll, const char *matrix_name, const char *filename);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_AMD_HLL_MATRIX_H */
