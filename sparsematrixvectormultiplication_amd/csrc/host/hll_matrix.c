/*
 * hll_matrix.c -- HLL (hacked ELLPACK) container, COO->HLL builder and the
 * slot-balanced contiguous hack partitioner.
 *
 * Product host code (plain C) behind include/hll_matrix.h.  Reference
 * behaviour being matched: src/hll_matrix.c:11-14 (init), :37-257
 * (convert_to_hll), :260-281 (free), :410-540
 * (prepare_thread_distribution_hll).  The CPU HLL SpMV kernels are the
 * oracle / CPU baseline and live in oracle/cpu_spmv.c.
 *
 * The builder goes COO -> (row-bucketed, column-ordered pairs) -> hacks with
 * three flat scratch arrays, instead of the reference's one malloc per row.
 */
#define _POSIX_C_SOURCE 200809L
#include "hll_matrix.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <time.h>

#include "coo_group.h"
#include "utility.h"

void init_hll_matrix(HLLMatrix *hll) {
    hll->num_blocks = 0;
    hll->blocks = NULL;
}

void free_hll_matrix(HLLMatrix *hll) {
    if (!hll) return;
    if (hll->blocks) {
        for (int b = 0; b < hll->num_blocks; ++b) {
            FREE_CHECK(hll->blocks[b].JA);
            FREE_CHECK(hll->blocks[b].AS);
        }
        FREE_CHECK(hll->blocks);
    }
    hll->num_blocks = 0;
}

/* Stable ordering of one row's (col, val) pairs by column: the reference
 * calls qsort() with a column-only comparator (src/hll_matrix.c:204-213),
 * which in glibc is a stable merge sort, so equal columns keep file order. */
static void order_pairs_stable(int *col, double *val, int n, int *tc, double *tv) {
    int sorted = 1;
    for (int k = 1; k < n; ++k)
        if (col[k] < col[k - 1]) { sorted = 0; break; }
    if (sorted) return;
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
        return;
    }
    /* bottom-up merge sort, ping-pong between (col,val) and (tc,tv) */
    int *sc = col, *dc = tc;
    double *sv = val, *dv = tv;
    for (int width = 1; width < n; width *= 2) {
        for (int lo = 0; lo < n; lo += 2 * width) {
            int mid = lo + width < n ? lo + width : n;
            int hi = lo + 2 * width < n ? lo + 2 * width : n;
            int a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                if (sc[b] < sc[a]) { dc[o] = sc[b]; dv[o++] = sv[b++]; }
                else { dc[o] = sc[a]; dv[o++] = sv[a++]; }
            }
            while (a < mid) { dc[o] = sc[a]; dv[o++] = sv[a++]; }
            while (b < hi) { dc[o] = sc[b]; dv[o++] = sv[b++]; }
        }
        int *xc = sc; sc = dc; dc = xc;
        double *xv = sv; sv = dv; dv = xv;
    }
    if (sc != col) {
        memcpy(col, sc, (size_t)n * sizeof(int));
        memcpy(val, sv, (size_t)n * sizeof(double));
    }
}

static double hll_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int convert_to_hll(const PreMatrix *pre, HLLMatrix *hll) {
    const int verbose = getenv("SPMV_VERBOSE") != NULL;
    const double t_start = hll_now();
    if (!hll) {
        printf("Errore: Parametri non validi\n");
        return -1;
    }
    init_hll_matrix(hll);
    if (!pre) {
        printf("Errore: Parametri non validi\n");
        return -1;
    }

    const int M = pre->M;
    const int num_blocks = (M + HACK_SIZE - 1) / HACK_SIZE;
    const size_t nz = (size_t)pre->nz;

    int *row_off = (int *)calloc((size_t)M + 1, sizeof(int));
    int *cols = (int *)malloc((nz ? nz : 1) * sizeof(int));
    double *vals = (double *)malloc((nz ? nz : 1) * sizeof(double));
    hll->blocks = (ELLPACKBlock *)calloc((size_t)(num_blocks ? num_blocks : 1), sizeof(ELLPACKBlock));
    hll->num_blocks = num_blocks;
    int *tc = NULL;
    double *tv = NULL;
    if (!row_off || !cols || !vals || !hll->blocks) {
        printf("Errore di allocazione memoria per i blocchi HLL\n");
        goto fail;
    }

    /* bucket the entries by row, keeping file order inside a row (all threads: coo_group.c) */
    {
        int bad_r = 0, bad_c = 0;
        const int g = coo_group_by_row(M, pre->N, nz, pre->I, pre->J, pre->val, row_off, cols, vals, &bad_r, &bad_c);
        if (g == -2) {
            printf("ERRORE: Indice non valido: riga=%d, colonna=%d\n", bad_r, bad_c);
            goto fail;
        }
        if (g != 0) {
            printf("Errore di allocazione memoria per i blocchi HLL\n");
            goto fail;
        }
    }
    int longest = 0;
    for (int r = 0; r < M; ++r)
        if (row_off[r + 1] - row_off[r] > longest) longest = row_off[r + 1] - row_off[r];
    const double t_grouped = hll_now();
    /* rows are independent: order them with all threads (per-thread merge scratch) */
    {
        int bad_alloc = 0;
#pragma omp parallel
        {
            int *my_c = (int *)malloc((size_t)(longest ? longest : 1) * sizeof(int));
            double *my_v = (double *)malloc((size_t)(longest ? longest : 1) * sizeof(double));
            if (!my_c || !my_v) {
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
                for (int r = 0; r < M; ++r) {
                    const int len = row_off[r + 1] - row_off[r];
                    if (len > 1) order_pairs_stable(cols + row_off[r], vals + row_off[r], len, my_c, my_v);
                }
            }
            free(my_c);
            free(my_v);
        }
        if (bad_alloc) {
            printf("Errore di allocazione memoria per sorted_elements\n");
            goto fail;
        }
    }

    const double t_ordered = hll_now();
    /* one ELLPACK slab per hack, row-major, padded; hacks are independent */
    int hack_failed = 0;
#pragma omp parallel for schedule(dynamic, 256)
    for (int b = 0; b < num_blocks; ++b) {
        const int r0 = b * HACK_SIZE;
        const int r1 = (b == num_blocks - 1) ? M : r0 + HACK_SIZE;
        const int rows = r1 - r0;
        int maxnz = 0;
        for (int r = r0; r < r1; ++r) {
            const int len = row_off[r + 1] - row_off[r];
            if (len > maxnz) maxnz = len;
        }
        ELLPACKBlock *blk = &hll->blocks[b];
        blk->M = rows;
        blk->N = pre->N;
        blk->MAXNZ = maxnz;
        blk->JA = NULL;
        blk->AS = NULL;
        if (maxnz == 0) continue;
        const size_t slots = (size_t)rows * (size_t)maxnz;
        blk->JA = (int *)malloc(slots * sizeof(int));
        blk->AS = (double *)malloc(slots * sizeof(double));
        if (!blk->JA || !blk->AS) {
#pragma omp atomic write
            hack_failed = 1;
            continue;
        }
        for (int r = r0; r < r1; ++r) {
            const int len = row_off[r + 1] - row_off[r];
            int *ja = blk->JA + (size_t)(r - r0) * maxnz;
            double *as = blk->AS + (size_t)(r - r0) * maxnz;
            memcpy(ja, cols + row_off[r], (size_t)len * sizeof(int));
            memcpy(as, vals + row_off[r], (size_t)len * sizeof(double));
            /* padding: value 0, column = the row's last real column (0 if none) */
            const int pad_col = len ? ja[len - 1] : 0;
            for (int j = len; j < maxnz; ++j) {
                ja[j] = pad_col;
                as[j] = 0.0;
            }
        }
    }
    if (hack_failed) {
        printf("ERRORE: Allocazione fallita per un blocco HLL\n");
        goto fail;
    }
    if (verbose)
        printf("convert_to_hll: group by row %.3f s, order rows %.3f s, fill hacks %.3f s\n", t_grouped - t_start,
               t_ordered - t_grouped, hll_now() - t_ordered);
    free(row_off);
    free(cols);
    free(vals);
    free(tc);
    free(tv);
    return 0;

fail:
    free(row_off);
    free(cols);
    free(vals);
    free(tc);
    free(tv);
    free_hll_matrix(hll);
    return -1;
}

void printHLLMatrix(HLLMatrix *hll) {
    printf("HLL Matrix con %d blocchi:\n", hll->num_blocks);
    for (int b = 0; b < hll->num_blocks; ++b) {
        const ELLPACKBlock *blk = &hll->blocks[b];
        printf("\nBlocco %d (%d righe, %d colonne, MAXNZ=%d):\n", b, blk->M, blk->N, blk->MAXNZ);
        for (int i = 0; i < blk->M; ++i) {
            printf("Riga %d: ", i);
            for (int j = 0; j < blk->MAXNZ; ++j) {
                const size_t idx = (size_t)i * blk->MAXNZ + j;
                printf("(%d, %.6f) ", blk->JA[idx], blk->AS[idx]);
            }
            printf("\n");
        }
    }
}

/*
 * Greedy contiguous split of the hacks into at most num_threads chunks
 * (reference: src/hll_matrix.c:410-540).  A hack's weight is the number of
 * its slots whose JA lies in [0, N) -- padding slots carry a valid column,
 * so in practice that is every slot, rows * MAXNZ.  Same closing rule as
 * prepare_thread_distribution.  Chunk ends fall on hack boundaries, i.e. on
 * multiples of 32 rows, which is what the multi-GPU HLL split needs.
 */
int prepare_thread_distribution_hll(const HLLMatrix *matrix, int num_threads,
                                    int **thread_block_start, int **thread_block_end) {
    if (!matrix || num_threads <= 0 || !thread_block_start || !thread_block_end) {
        printf("Errore: parametri non validi in prepare_thread_distribution_hll\n");
        return 0;
    }
    const int nb = matrix->num_blocks;
    if (num_threads > nb) num_threads = nb;
    if (num_threads <= 0) {
        *thread_block_start = NULL;
        *thread_block_end = NULL;
        return 0;
    }

    int *start = (int *)malloc((size_t)num_threads * sizeof(int));
    int *end = (int *)malloc((size_t)num_threads * sizeof(int));
    long long *load = (long long *)calloc((size_t)num_threads, sizeof(long long));
    long long *weight = (long long *)malloc((size_t)nb * sizeof(long long));
    if (!start || !end || !load || !weight) {
        free(start);
        free(end);
        free(load);
        free(weight);
        printf("Errore: allocazione memoria fallita\n");
        *thread_block_start = NULL;
        *thread_block_end = NULL;
        return 0;
    }
    for (int t = 0; t < num_threads; ++t) start[t] = end[t] = -1;

    long long total = 0;
    for (int b = 0; b < nb; ++b) {
        const ELLPACKBlock *blk = &matrix->blocks[b];
        long long w = 0;
        if (blk->JA) {
            const size_t slots = (size_t)blk->M * (size_t)blk->MAXNZ;
            for (size_t s = 0; s < slots; ++s) w += (blk->JA[s] >= 0 && blk->JA[s] < blk->N);
        }
        weight[b] = w;
        total += w;
    }

    const long long target = (total + num_threads - 1) / num_threads;
    int t = 0;
    long long running = 0;
    for (int b = 0; b < nb; ++b) {
        if (start[t] < 0) start[t] = b;
        running += weight[b];
        load[t] += weight[b];
        if (running >= target && t < num_threads - 1) {
            end[t] = b + 1;
            ++t;
            running = 0;
        }
    }
    if (t < num_threads) end[t] = nb;

    int kept = 0;
    for (int k = 0; k < num_threads; ++k) {
        if (start[k] < 0 || end[k] < 0 || load[k] <= 0) continue;
        start[kept] = start[k];
        end[kept] = end[k];
        load[kept] = load[k];
        ++kept;
    }
    if (getenv("SPMV_VERBOSE")) {
        printf("\n--- Dettagli distribuzione thread per HLL ---\n");
        printf("Thread attivi: %d (su %d richiesti inizialmente)\n", kept, num_threads);
        for (int k = 0; k < kept; ++k)
            printf("Thread %d: %d blocchi (da %d a %d), %lld nnz (%.2f%% del totale)\n", k,
                   end[k] - start[k], start[k], end[k] - 1, load[k],
                   total ? (double)load[k] * 100.0 / (double)total : 0.0);
        printf("--- Fine dettagli distribuzione HLL ---\n\n");
    }
    free(load);
    free(weight);
    *thread_block_start = start;
    *thread_block_end = end;
    return kept;
}
