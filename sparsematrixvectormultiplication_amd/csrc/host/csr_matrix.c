/*
 * csr_matrix.c -- CSR container, COO->CSR builder and the nnz-balanced
 * contiguous row partitioner.
 *
 * Product host code (plain C) behind include/csr_matrix.h.  Reference
 * behaviour being matched: src/csr_matrix.c:11-25 (init/free), :63-126
 * (convert_in_csr), :167-266 (prepare_thread_distribution).  The CPU SpMV
 * kernels of that file are NOT here: they are the oracle / CPU baseline and
 * live in oracle/cpu_spmv.c.
 */
#include "csr_matrix.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "coo_group.h"
#include "utility.h"

void init_csr_matrix(CSRMatrix *mat) {
    mat->M = 0;
    mat->N = 0;
    mat->nz = 0;
    mat->row_ptr = NULL;
    mat->col_idx = NULL;
    mat->values = NULL;
}

void free_csr_matrix(CSRMatrix *mat) {
    FREE_CHECK(mat->row_ptr);
    FREE_CHECK(mat->col_idx);
    FREE_CHECK(mat->values);
    init_csr_matrix(mat);
}

void write_memory_stats_to_csv(const char *matrix_name, int nz, size_t total_memory_bytes) {
    const char *path = "../result/matrix_memory_stats_csr.csv";
    FILE *probe = fopen(path, "r");
    const int exists = probe != NULL;
    if (probe) fclose(probe);
    FILE *fp = fopen(path, "a");
    if (!fp) {
        printf("Errore nell'apertura del file CSV per le statistiche di memoria\n");
        return;
    }
    if (!exists) fprintf(fp, "Matrix Name,Non-Zero Elements,Memory Size (MB)\n");
    fprintf(fp, "%s,%d,%.4f\n", matrix_name, nz, (double)total_memory_bytes / (1024.0 * 1024.0));
    fclose(fp);
}

/* ---- per-row ordering ---------------------------------------------------
 *
 * The reference sorts every row with its paired Lomuto quicksort
 * (src/csr_matrix.c:115-123 -> src/utility.c:58-91).  For a row whose column
 * indices are all distinct the sorted row is unique, so ANY correct sort
 * reproduces the reference bit for bit; we use cheap ones.  Only a row that
 * holds the same column more than once depends on the sort's tie behaviour;
 * those rows are re-sorted from their original order with sort_row(), the
 * exact restatement of the reference's scheme.  This keeps the O(n^2) cost of
 * last-element-pivot quicksort on pre-sorted input away from the common case
 * (SuiteSparse files are mostly already ordered).
 */
typedef struct {
    int col;
    double val;
} ColVal;

static int cmp_colval(const void *a, const void *b) {
    const int ca = ((const ColVal *)a)->col, cb = ((const ColVal *)b)->col;
    return (ca > cb) - (ca < cb);
}

/* returns 1 if the row now holds a repeated column */
static int order_row_fast(int *col, double *val, int n, ColVal *scratch) {
    int sorted = 1, strict = 1;
    for (int k = 1; k < n; ++k) {
        if (col[k] < col[k - 1]) { sorted = 0; break; }
        if (col[k] == col[k - 1]) strict = 0;
    }
    if (sorted) return !strict;
    if (n <= 64) {
        for (int k = 1; k < n; ++k) {
            const int c = col[k];
            const double v = val[k];
            int p = k - 1;
            while (p >= 0 && col[p] > c) {
                col[p + 1] = col[p];
                val[p + 1] = val[p];
                --p;
            }
            col[p + 1] = c;
            val[p + 1] = v;
        }
    } else {
        for (int k = 0; k < n; ++k) {
            scratch[k].col = col[k];
            scratch[k].val = val[k];
        }
        qsort(scratch, (size_t)n, sizeof(ColVal), cmp_colval);
        for (int k = 0; k < n; ++k) {
            col[k] = scratch[k].col;
            val[k] = scratch[k].val;
        }
    }
    for (int k = 1; k < n; ++k)
        if (col[k] == col[k - 1]) return 1;
    return 0;
}

int convert_in_csr(const PreMatrix *pre, CSRMatrix *csr, const char *matrix_name) {
    (void)matrix_name;
    init_csr_matrix(csr);
    csr->M = pre->M;
    csr->N = pre->N;
    csr->nz = pre->nz;
    memcpy(csr->type, pre->type, sizeof(MM_typecode));

    const size_t M = (size_t)pre->M, nz = (size_t)pre->nz;
    csr->row_ptr = (int *)calloc(M + 1, sizeof(int));
    csr->col_idx = (int *)malloc((nz ? nz : 1) * sizeof(int));
    csr->values = (double *)malloc((nz ? nz : 1) * sizeof(double));
    if (!csr->row_ptr || !csr->col_idx || !csr->values) {
        printf("Errore di allocazione memoria nella conversione CSR\n");
        free_csr_matrix(csr);
        return -1;
    }

    /* histogram -> exclusive scan -> scatter in file order (all threads: coo_group.c); like the
     * reference, indices are trusted here (read_matrix_market has checked them) */
    int bad_r = 0, bad_c = 0;
    if (coo_group_by_row(pre->M, -1, nz, pre->I, pre->J, pre->val, csr->row_ptr, csr->col_idx, csr->values,
                         &bad_r, &bad_c) != 0) {
        printf("Errore di allocazione memoria nella conversione CSR\n");
        free_csr_matrix(csr);
        return -1;
    }

    /* ascending columns inside each row */
    int longest = 0;
    for (size_t r = 0; r < M; ++r) {
        const int len = csr->row_ptr[r + 1] - csr->row_ptr[r];
        if (len > longest) longest = len;
    }
    /* rows are independent: order them with all threads, each with its own scratch */
    int bad_alloc = 0;
#pragma omp parallel
    {
        ColVal *scratch = (ColVal *)malloc((size_t)(longest ? longest : 1) * sizeof(ColVal));
        int *keep_c = (int *)malloc((size_t)(longest ? longest : 1) * sizeof(int));
        double *keep_v = (double *)malloc((size_t)(longest ? longest : 1) * sizeof(double));
        if (!scratch || !keep_c || !keep_v) {
#pragma omp atomic write
            bad_alloc = 1;
        }
        /* all threads enter the worksharing loop or none does */
#pragma omp barrier
        int any_bad;
#pragma omp atomic read
        any_bad = bad_alloc;
        if (!any_bad) {
#pragma omp for schedule(dynamic, 2048)
            for (long long r = 0; r < (long long)M; ++r) {
                const int s = csr->row_ptr[r], len = csr->row_ptr[r + 1] - s;
                if (len < 2) continue;
                memcpy(keep_c, csr->col_idx + s, (size_t)len * sizeof(int));
                memcpy(keep_v, csr->values + s, (size_t)len * sizeof(double));
                if (order_row_fast(csr->col_idx + s, csr->values + s, len, scratch)) {
                    /* repeated column: tie order is defined by the reference's scheme */
                    memcpy(csr->col_idx + s, keep_c, (size_t)len * sizeof(int));
                    memcpy(csr->values + s, keep_v, (size_t)len * sizeof(double));
                    sort_row(csr->col_idx + s, csr->values + s, 0, (size_t)len - 1);
                }
            }
        }
        free(scratch);
        free(keep_c);
        free(keep_v);
    }
    if (bad_alloc) {
        printf("Errore di allocazione memoria nella conversione CSR\n");
        free_csr_matrix(csr);
        return -1;
    }
    return 0;
}

void print_csr_matrix(const CSRMatrix *mat) {
    MM_typecode tc;
    memcpy(tc, mat->type, sizeof tc);
    char *ts = mm_typecode_to_str(tc);
    printf("Dimensioni matrice: %d x %d\n", mat->M, mat->N);
    printf("Numero di elementi non-zero: %d\n", mat->nz);
    printf("Matrix type: %s\n", ts ? ts : "?");
    free(ts);
    if (mat->M > 30 || mat->N > 30) return;
    printf("row_ptr: ");
    for (int i = 0; i <= mat->M; i++) printf("%d ", mat->row_ptr[i]);
    printf("\ncol_idx: ");
    for (int i = 0; i < mat->nz; i++) printf("%d ", mat->col_idx[i]);
    printf("\nvalues: ");
    for (int i = 0; i < mat->nz; i++) printf("%f ", mat->values[i]);
    printf("\n");
}

/*
 * Greedy contiguous split of the rows into at most num_threads chunks of
 * about ceil(total_nnz / num_threads) nonzeros each (reference:
 * src/csr_matrix.c:167-266).  Walk the rows; a chunk is closed right after
 * the row that brings its running count to the target, except that the last
 * chunk always runs to the end; chunks that received no nonzero are dropped.
 * The same routine splits rows across GPUs (spmv_hip_partition_rows).
 *
 * The reference prints a per-thread table on every call; here that only
 * happens when SPMV_VERBOSE is set in the environment.
 */
int prepare_thread_distribution(const int num_row, const int *row_ptr, int num_threads,
                                const long long total_nnz, int **thread_row_start,
                                int **thread_row_end) {
    if (num_row <= 0 || num_threads <= 0) return 0;
    if (num_threads > num_row) num_threads = num_row;

    int *start = (int *)malloc((size_t)num_threads * sizeof(int));
    int *end = (int *)malloc((size_t)num_threads * sizeof(int));
    long long *load = (long long *)calloc((size_t)num_threads, sizeof(long long));
    if (!start || !end || !load) {
        free(start);
        free(end);
        free(load);
        *thread_row_start = NULL;
        *thread_row_end = NULL;
        return 0;
    }
    for (int t = 0; t < num_threads; ++t) start[t] = end[t] = -1;

    const long long target = (total_nnz + num_threads - 1) / num_threads;
    int t = 0;
    long long running = 0;
    for (int r = 0; r < num_row; ++r) {
        if (start[t] < 0) start[t] = r;
        const int len = row_ptr[r + 1] - row_ptr[r];
        running += len;
        load[t] += len;
        if (running >= target && t < num_threads - 1) {
            end[t] = r + 1;
            ++t;
            running = 0;
        }
    }
    if (t < num_threads) end[t] = num_row;

    int kept = 0;
    for (int k = 0; k < num_threads; ++k) {
        if (start[k] < 0 || end[k] < 0 || load[k] <= 0) continue;
        start[kept] = start[k];
        end[kept] = end[k];
        load[kept] = load[k];
        ++kept;
    }

    if (getenv("SPMV_VERBOSE")) {
        printf("\n--- Dettagli distribuzione thread ---\n");
        printf("Thread attivi: %d (su %d richiesti inizialmente)\n", kept, num_threads);
        for (int k = 0; k < kept; ++k)
            printf("Thread %d: %d righe (da %d a %d), %lld nnz (%.2f%% del totale)\n", k,
                   end[k] - start[k], start[k], end[k] - 1, load[k],
                   total_nnz ? (double)load[k] * 100.0 / (double)total_nnz : 0.0);
        printf("--- Fine dettagli distribuzione ---\n\n");
    }
    free(load);
    *thread_row_start = start;
    *thread_row_end = end;
    return kept;
}
