/*
 * utility.c -- host helpers behind include/utility.h.
 *
 * New implementations of the helpers the reference keeps in src/utility.c
 * (and the two CSV writers of cuda_src/utility.cu).  File formats written
 * here are byte-compatible with the reference's (same header line, same
 * column order, "%.15f" fields) so downstream scripts keep working.
 */
#include "utility.h"

#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

void init_vector_at_one(double *v, const int size) {
    for (int i = 0; i < size; i++) v[i] = 1.0;
}

void swap(int *a, int *b) {
    const int t = *a;
    *a = *b;
    *b = t;
}

void swap_double(double *a, double *b) {
    const double t = *a;
    *a = *b;
    *b = t;
}

/*
 * Lomuto partition around the LAST element, "<=" goes left.  This is the
 * partition scheme of the reference's paired quicksort (src/utility.c:38-53);
 * it is restated exactly because, for rows that hold the same column twice,
 * the order in which the equal keys come out (and therefore the order in
 * which their values are summed) is a property of this scheme.
 */
size_t partition(int *col_idx, double *values, const size_t low, const size_t high) {
    const int pivot = col_idx[high];
    size_t store = low;
    for (size_t j = low; j < high; ++j) {
        if (col_idx[j] <= pivot) {
            swap(&col_idx[store], &col_idx[j]);
            swap_double(&values[store], &values[j]);
            ++store;
        }
    }
    swap(&col_idx[store], &col_idx[high]);
    swap_double(&values[store], &values[high]);
    return store;
}

/*
 * Paired quicksort of (col_idx, values)[low..high], bounds inclusive.
 * Always iterative, smaller half first, so the explicit stack stays below
 * 2*log2(n) entries.  Sub-ranges are disjoint, so the visiting order does not
 * change the resulting permutation: output == the reference's sort_row
 * (src/utility.c:58-91) for every input, duplicates included.
 */
void sort_row(int *col_idx, double *values, size_t low, size_t high) {
    if (low >= high) return;
    size_t stack[2 * 64];
    int top = 0;
    stack[top++] = low;
    stack[top++] = high;
    while (top > 0) {
        const size_t hi = stack[--top];
        const size_t lo = stack[--top];
        if (lo >= hi) continue;
        const size_t p = partition(col_idx, values, lo, hi);
        const size_t left_n = p - lo;  /* [lo, p-1] */
        const size_t right_n = hi - p; /* [p+1, hi] */
        /* push the larger range first so the smaller one is handled next */
        if (left_n > right_n) {
            if (left_n > 1) { stack[top++] = lo; stack[top++] = p - 1; }
            if (right_n > 1) { stack[top++] = p + 1; stack[top++] = hi; }
        } else {
            if (right_n > 1) { stack[top++] = p + 1; stack[top++] = hi; }
            if (left_n > 1) { stack[top++] = lo; stack[top++] = p - 1; }
        }
    }
}

/* open for append; *is_new tells the caller to emit the header line */
static FILE *open_csv(const char *path, int *is_new) {
    FILE *fp = fopen(path, "a+");
    if (!fp) {
        printf("Errore nell'apertura del file %s\n", path);
        return NULL;
    }
    fseek(fp, 0, SEEK_END);
    *is_new = ftell(fp) == 0;
    return fp;
}

static void put_doubles(FILE *fp, int n, ...) {
    va_list ap;
    va_start(ap, n);
    for (int i = 0; i < n; ++i) fprintf(fp, ",%.15f", va_arg(ap, double));
    va_end(ap);
}

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
    const double flops_parallel_hll_simd, const char *output_file) {
    int is_new = 0;
    FILE *fp = open_csv(output_file, &is_new);
    if (!fp) return;
    if (is_new)
        fputs("matrix_name,rows,cols,nonzeros,num_threads,"
              "time_serial,time_serial_hll,time_parallel,time_parallel_simd,time_parallel_hll,"
              "time_parallel_hll_simd,"
              "error_csr_relative,error_csr_absolute,error_hll_relative,error_hll_absolute,"
              "error_csr_simd_relative,error_csr_simd_absolute,"
              "error_hll_simd_relative,error_hll_simd_absolute,"
              "flops_serial,flops_serial_hll,flops_parallel,flops_parallel_simd,"
              "flops_parallel_hll,flops_parallel_hll_simd,"
              "speedup_parallel,speedup_simd,speedup_hll,speedup_hll_simd,"
              "efficiency_parallel,efficiency_simd,efficiency_hll,efficiency_hll_simd\n",
              fp);
    fprintf(fp, "%s,%d,%d,%d,%d", matrix_name, num_rows, num_cols, nz, num_threads);
    put_doubles(fp, 6, time_serial, time_serial_hll, time_parallel, time_parallel_simd,
                time_parallel_hll, time_parallel_hll_simd);
    put_doubles(fp, 8, error_csr.mean_rel_err, error_csr.mean_abs_err, error_hll.mean_rel_err,
                error_hll.mean_abs_err, error_csr_simd.mean_rel_err, error_csr_simd.mean_abs_err,
                error_hll_simd.mean_rel_err, error_hll_simd.mean_abs_err);
    put_doubles(fp, 6, flops_serial, avg_flops_hll_serial, flops_parallel, flops_parallel_simd,
                flops_parallel_hll, flops_parallel_hll_simd);
    put_doubles(fp, 8, speedup_parallel, speedup_simd, speedup_hll, speedup_hll_simd,
                efficiency_parallel, efficiency_simd, efficiency_hll, efficiency_hll_simd);
    fputc('\n', fp);
    fclose(fp);
}

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
    const char *output_file) {
    int is_new = 0;
    FILE *fp = open_csv(output_file, &is_new);
    if (!fp) return;
    if (is_new)
        fputs("matrix_name,rows,cols,nonzeros,"
              "time_serial,time_serial_hll,time_row_csr,time_warp_csr,time_warp_shared_csr,"
              "time_row_hll,time_warp_hll,time_warp_shared_hll,"
              "flops_serial,avg_flops_hll_serial,flops_row_csr,flops_warp_csr,"
              "flops_warp_csr_shared,flops_row_hll,flops_warp_hll,flops_warp_shared_hll,"
              "relative_error_row_csr,absolute_error_row_csr,"
              "relative_error_warp_csr,absolute_error_warp_csr,"
              "relative_error_warp_shared_csr,absolute_error_warp_shared_csr,"
              "relative_error_row_hll,absolute_error_row_hll,"
              "relative_error_warp_hll,absolute_error_warp_hll,"
              "relative_error_warp_shared_hll,absolute_error_warp_shared_hll\n",
              fp);
    fprintf(fp, "%s,%d,%d,%d", matrix_name, num_rows, num_cols, nz);
    put_doubles(fp, 8, time_serial, time_serial_hll, time_row_csr, time_warp_csr,
                time_warp_csr_shared, time_row_hll, time_warp_hll, time_warp_shared_hll);
    put_doubles(fp, 8, flops_serial, avg_flops_hll_serial, flops_row_csr, flops_warp_csr,
                flops_warp_csr_shared, flops_row_hll, flops_warp_hll, flops_warp_shared_hll);
    put_doubles(fp, 12, mediumCsrParallel.mean_rel_err, mediumCsrParallel.mean_abs_err,
                mediumCsrWarp.mean_rel_err, mediumCsrWarp.mean_abs_err,
                mediumCsrWarpShared.mean_rel_err, mediumCsrWarpShared.mean_abs_err,
                mediumHllNaive.mean_rel_err, mediumHllNaive.mean_abs_err,
                mediumHllWarp.mean_rel_err, mediumHllWarp.mean_abs_err,
                mediumHllWarpShared.mean_rel_err, mediumHllWarpShared.mean_abs_err);
    fputc('\n', fp);
    fclose(fp);
}

void write_block_result_to_csv(const char *matrix_name, const int nz, int block_size_csr_row,
                               int block_size_csr_warp, int block_size_csr_shared,
                               int block_size_hll_row, int block_size_hll_warp,
                               int block_size_hll_shared, const char *output_file) {
    int is_new = 0;
    FILE *fp = open_csv(output_file, &is_new);
    if (!fp) return;
    if (is_new)
        fputs("matrix_name,nonzeros,block_size_csr_row,block_size_csr_warp,block_size_csr_shared,"
              "block_size_hll_row,block_size_hll_warp,block_size_hll_shared\n",
              fp);
    fprintf(fp, "%s,%d,%d,%d,%d,%d,%d,%d\n", matrix_name, nz, block_size_csr_row,
            block_size_csr_warp, block_size_csr_shared, block_size_hll_row, block_size_hll_warp,
            block_size_hll_shared);
    fclose(fp);
}

/* touch one byte per 64-byte line of a scratch buffer to evict CPU caches
 * between tests (reference: src/utility.c:141-159) */
void clear_cache(size_t clear_size_mb) {
    const size_t bytes = clear_size_mb << 20;
    volatile char *scratch = (volatile char *)malloc(bytes ? bytes : 1);
    if (!scratch) {
        perror("Failed to allocate cache clearing buffer");
        return;
    }
    for (size_t i = 0; i < bytes; i += 64) scratch[i] = (char)(i & 0xff);
    free((void *)scratch);
}

/* Create the directory if it does not exist.  Unlike the reference
 * (src/utility.c:200-209) an existing directory is left untouched. */
void create_directory(const char *path) {
    if (mkdir(path, 0777) == -1 && errno != EEXIST) {
        perror("Errore nella creazione della directory");
        exit(EXIT_FAILURE);
    }
}

int process_matrix_file(const char *filepath, PreMatrix *pre_mat) {
    init_pre_matrix(pre_mat);
    printf("\n===========================================\n");
    printf("Elaborazione matrice: %s\n", filepath);
    printf("===========================================\n");
    if (read_matrix_market(filepath, pre_mat) != 0) {
        printf("Errore nella lettura della matrice\n");
        return -1;
    }
    return 0;
}
