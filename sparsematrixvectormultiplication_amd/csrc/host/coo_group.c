/* coo_group.c -- see coo_group.h. */
#include "coo_group.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int group_serial(int M, int N, size_t nz, const int *I, const int *J, const double *val, int *row_off,
                        int *cols, double *vals, int *bad_row, int *bad_col) {
    memset(row_off, 0, ((size_t)M + 1) * sizeof(int));
    for (size_t e = 0; e < nz; ++e) {
        const int r = I[e], c = J[e];
        if (r < 0 || r >= M || (N >= 0 && (c < 0 || c >= N))) {
            *bad_row = r;
            *bad_col = c;
            return -2;
        }
        row_off[r + 1]++;
    }
    for (int r = 0; r < M; ++r) row_off[r + 1] += row_off[r];
    int *cursor = (int *)malloc(((size_t)M ? (size_t)M : 1) * sizeof(int));
    if (!cursor) return -1;
    memcpy(cursor, row_off, (size_t)M * sizeof(int));
    for (size_t e = 0; e < nz; ++e) {
        const int dst = cursor[I[e]]++;
        cols[dst] = J[e];
        vals[dst] = val[e];
    }
    free(cursor);
    return 0;
}

int coo_group_by_row(int M, int N, size_t nz, const int *I, const int *J, const double *val, int *row_off,
                     int *cols, double *vals, int *bad_row, int *bad_col) {
    int threads = 1;
#ifdef _OPENMP
    threads = omp_get_max_threads();
#endif
    if (threads < 2 || nz < ((size_t)1 << 21) || M < 4096)
        return group_serial(M, N, nz, I, J, val, row_off, cols, vals, bad_row, bad_col);

    /* level 1: row ranges of 2^shift rows, at most 4096 of them; a range's entries are later
     * handled by one thread with its per-row counters resident in cache */
    int shift = 10;
    while (((size_t)M >> shift) + 1 > 4096) ++shift;
    const int B = (int)(((size_t)M + ((size_t)1 << shift) - 1) >> shift);
    size_t *cnt = (size_t *)calloc((size_t)threads * B, sizeof(size_t));
    size_t *bucket_begin = (size_t *)malloc(((size_t)B + 1) * sizeof(size_t));
    int *tI = (int *)malloc(nz * sizeof(int));
    int *tJ = (int *)malloc(nz * sizeof(int));
    double *tV = (double *)malloc(nz * sizeof(double));
    int rc = 0, first_bad_r = 0, first_bad_c = 0;
    size_t first_bad_at = (size_t)-1;
    if (!cnt || !bucket_begin || !tI || !tJ || !tV) {
        rc = -1;
        goto done;
    }
#pragma omp parallel num_threads(threads)
    {
        int t = 0, T = 1;
#ifdef _OPENMP
        t = omp_get_thread_num();
        T = omp_get_num_threads();
#endif
        /* contiguous chunk of the file per thread: together with the (bucket, thread) order of the
         * offsets below this keeps file order inside every bucket, hence inside every row */
        const size_t lo = nz * (size_t)t / (size_t)T, hi = nz * ((size_t)t + 1) / (size_t)T;
        size_t *mine = cnt + (size_t)t * B;
        for (size_t e = lo; e < hi; ++e) {
            const int r = I[e], c = J[e];
            if (r < 0 || r >= M || (N >= 0 && (c < 0 || c >= N))) {
#pragma omp critical
                if (e < first_bad_at) {
                    first_bad_at = e;
                    first_bad_r = r;
                    first_bad_c = c;
                }
                break;
            }
            mine[r >> shift]++;
        }
#pragma omp barrier
#pragma omp single
        {
            if (first_bad_at == (size_t)-1) {
                size_t run = 0;
                for (int b = 0; b < B; ++b) {
                    bucket_begin[b] = run;
                    for (int k = 0; k < T; ++k) {
                        const size_t n = cnt[(size_t)k * B + b];
                        cnt[(size_t)k * B + b] = run; /* becomes thread k's write cursor in bucket b */
                        run += n;
                    }
                }
                bucket_begin[B] = run;
            }
        } /* implicit barrier */
        if (first_bad_at == (size_t)-1) {
            for (size_t e = lo; e < hi; ++e) {
                const size_t dst = mine[I[e] >> shift]++;
                tI[dst] = I[e];
                tJ[dst] = J[e];
                tV[dst] = val[e];
            }
#pragma omp barrier
            /* level 2: one row range at a time */
            int *cursor = (int *)malloc(((size_t)1 << shift) * sizeof(int));
            if (!cursor) {
#pragma omp atomic write
                rc = -1;
            }
            /* every thread of the team must meet the worksharing loop or none: agree on the
             * allocation outcome first (first_bad_at is the same for all threads, so the barriers
             * inside this branch are met by all of them too) */
#pragma omp barrier
            int all_ok;
#pragma omp atomic read
            all_ok = rc;
            if (all_ok == 0) {
#pragma omp for schedule(dynamic, 4)
                for (int b = 0; b < B; ++b) {
                    const int r0 = b << shift;
                    const int rows = (M - r0) < (1 << shift) ? (M - r0) : (1 << shift);
                    const size_t e0 = bucket_begin[b], e1 = bucket_begin[b + 1];
                    memset(cursor, 0, (size_t)rows * sizeof(int));
                    for (size_t e = e0; e < e1; ++e) cursor[tI[e] - r0]++;
                    size_t run = e0;
                    for (int k = 0; k < rows; ++k) {
                        const int n = cursor[k];
                        row_off[r0 + k] = (int)run;
                        cursor[k] = (int)run;
                        run += (size_t)n;
                    }
                    for (size_t e = e0; e < e1; ++e) {
                        const int dst = cursor[tI[e] - r0]++;
                        cols[dst] = tJ[e];
                        vals[dst] = tV[e];
                    }
                }
            }
            free(cursor);
        }
    }
    if (first_bad_at != (size_t)-1) {
        *bad_row = first_bad_r;
        *bad_col = first_bad_c;
        rc = -2;
    } else if (rc == 0) {
        row_off[M] = (int)nz;
    }
done:
    free(cnt);
    free(bucket_begin);
    free(tI);
    free(tJ);
    free(tV);
    return rc;
}
