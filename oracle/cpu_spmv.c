/*
 * oracle/cpu_spmv.c -- CPU restatement of the reference's SpMV hot path.
 *
 * *** TEST INFRASTRUCTURE, NOT PRODUCT. ***
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libspmv_amd.so) never links, loads or
 * calls anything in oracle/; its GPU entry points fail with -1 when the HIP
 * device is unavailable instead of computing on the CPU.
 *
 * Parity pinning: this file is checked bit-for-bit (a) against the reference
 * itself compiled from /root/reference into oracle/_ref/ (tests/test_oracle_
 * vs_ref.py, runs wherever /root/reference exists) and (b) against the golden
 * vectors under tests/golden/ that tests/golden/make_golden.py produced by
 * running that compiled reference (runs everywhere, GPU box included), which
 * include the reference's own bundled matrix matrix_generated/
 * general_matrix.mtx.  Build with -O2 -ffp-contract=off (no FMA fusion, like
 * the reference's gcc -O2 x86-64 build) -- see oracle/Makefile.
 *
 * Each function names the reference lines it restates.
 */
#include <omp.h>
#include <stddef.h>

#include "csr_matrix.h"
#include "hll_matrix.h"

/* K1 -- src/csr_matrix.c:130-139.  Row by row, entries in stored (ascending
 * column) order, each product added straight into y[i]: y must arrive zeroed
 * and is ACCUMULATED into.  This is the function every other kernel in the
 * reference (and in this repo) is compared against. */
void csr_matrix_vector_mult(const int num_row, const int *row_ptr, const int *col_idx,
                            const double *values, const double *x, double *y) {
    for (int r = 0; r < num_row; ++r) {
        const int stop = row_ptr[r + 1];
        for (int e = row_ptr[r]; e < stop; ++e) y[r] += values[e] * x[col_idx[e]];
    }
}

/* K2 -- src/csr_matrix.c:294-313.  One OpenMP thread per precomputed row
 * range, private running sum per row, y overwritten.  The range is looked up
 * by omp_get_thread_num(), so the caller must make sure the runtime really
 * grants num_threads threads (OMP_DYNAMIC=false). */
void spvm_csr_parallel(const int *row_ptr, const int *col_idx, const double *values,
                       const double *x, double *y, int num_threads, const int *thread_row_start,
                       const int *thread_row_end) {
#pragma omp parallel num_threads(num_threads)
    {
        const int me = omp_get_thread_num();
        const int last = thread_row_end[me];
        for (int r = thread_row_start[me]; r < last; ++r) {
            double acc = 0.0;
            const int stop = row_ptr[r + 1];
            for (int e = row_ptr[r]; e < stop; ++e) acc += values[e] * x[col_idx[e]];
            y[r] = acc;
        }
    }
}

/* K3 -- src/csr_matrix.c:269-292.  K2 with the inner loop marked
 * `omp simd reduction(+)`, which lets the compiler reassociate the sum. */
void spvm_csr_parallel_simd(const int *row_ptr, const int *col_idx, const double *values,
                            const double *x, double *y, int num_threads,
                            const int *thread_row_start, const int *thread_row_end) {
#pragma omp parallel num_threads(num_threads)
    {
        const int me = omp_get_thread_num();
        const int last = thread_row_end[me];
        for (int r = thread_row_start[me]; r < last; ++r) {
            double acc = 0.0;
            const int begin = row_ptr[r], stop = row_ptr[r + 1];
#pragma omp simd reduction(+ : acc)
            for (int e = begin; e < stop; ++e) acc += values[e] * x[col_idx[e]];
            y[r] = acc;
        }
    }
}

/* one hack: rows r of the slab, slots 0..MAXNZ-1 of each row INCLUDING the
 * zero-valued padding, row-major */
static inline void hack_rows(const ELLPACKBlock *blk, const double *x, double *y_hack) {
    const int width = blk->MAXNZ;
    for (int r = 0; r < blk->M; ++r) {
        double acc = 0.0;
        for (int s = 0; s < width; ++s) {
            const size_t at = (size_t)r * width + s; /* reference uses int: r*MAXNZ+s */
            acc += blk->AS[at] * x[blk->JA[at]];
        }
        y_hack[r] = acc;
    }
}

/* K5 -- src/hll_matrix.c:286-308.  Hack b writes y[32*b .. 32*b + M_b). */
void spmv_hll_serial(const int num_blocks, const ELLPACKBlock *blocks, const double *x, double *y) {
    for (int b = 0; b < num_blocks; ++b) hack_rows(&blocks[b], x, y + (size_t)b * HACK_SIZE);
}

/* K6 -- src/hll_matrix.c:376-408.  K5 over a per-thread hack range. */
void spmv_hll(const ELLPACKBlock *blocks, const double *x, double *y, int num_threads,
              int const *thread_block_start, int const *thread_block_end) {
#pragma omp parallel num_threads(num_threads)
    {
        const int me = omp_get_thread_num();
        const int last = thread_block_end[me];
        for (int b = thread_block_start[me]; b < last; ++b)
            hack_rows(&blocks[b], x, y + (size_t)b * HACK_SIZE);
    }
}

/* K7 -- src/hll_matrix.c:339-374.  K6 with an `omp simd reduction` inner loop. */
void spmv_hll_simd(const ELLPACKBlock *blocks, const double *x, double *y, int num_threads,
                   int const *thread_block_start, int const *thread_block_end) {
#pragma omp parallel num_threads(num_threads)
    {
        const int me = omp_get_thread_num();
        const int last = thread_block_end[me];
        for (int b = thread_block_start[me]; b < last; ++b) {
            const ELLPACKBlock *blk = &blocks[b];
            const int width = blk->MAXNZ;
            const int *ja = blk->JA;
            const double *as = blk->AS;
            double *y_hack = y + (size_t)b * HACK_SIZE;
            for (int r = 0; r < blk->M; ++r) {
                double acc = 0.0;
#pragma omp simd reduction(+ : acc)
                for (int s = 0; s < width; ++s) {
                    const int at = r * width + s;
                    acc += as[at] * x[ja[at]];
                }
                y_hack[r] = acc;
            }
        }
    }
}

/* ---- fp32 data, fp64 accumulation ---------------------------------------
 * The reference has no single-precision path.  BASELINE.json config 5 (fp32
 * CSR) is checked against K1's algorithm run on the fp32-rounded data with a
 * double accumulator: same loop, same order, `float` loads widened first. */
void oracle_csr_f32_accum64(const int num_row, const int *row_ptr, const int *col_idx,
                            const float *values, const float *x, double *y) {
    for (int r = 0; r < num_row; ++r) {
        const int stop = row_ptr[r + 1];
        for (int e = row_ptr[r]; e < stop; ++e) y[r] += (double)values[e] * (double)x[col_idx[e]];
    }
}

/* per-row sum of |a_ij * x_j|: the scale that bounds how far a re-ordered
 * summation of the same row may legitimately drift (tests use it for the
 * 1e-10 parity gate on rows whose exact sum cancels to ~0) */
void oracle_csr_row_abs_sums(const int num_row, const int *row_ptr, const int *col_idx,
                             const double *values, const double *x, double *out) {
    for (int r = 0; r < num_row; ++r) {
        double acc = 0.0;
        const int stop = row_ptr[r + 1];
        for (int e = row_ptr[r]; e < stop; ++e) {
            const double p = values[e] * x[col_idx[e]];
            acc += p < 0 ? -p : p;
        }
        out[r] = acc;
    }
}

int oracle_max_threads(void) { return omp_get_max_threads(); }
