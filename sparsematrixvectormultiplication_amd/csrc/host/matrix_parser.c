/*
 * matrix_parser.c -- Matrix Market coordinate file -> PreMatrix (COO).
 *
 * Same observable semantics as the reference's read_matrix_market
 * (src/matrix_parser.c:25-150):
 *   - only `matrix coordinate` files are accepted;
 *   - entries are a whitespace-separated token stream after the size line
 *     ("i j v", or "i j" for pattern files whose value becomes 1.0);
 *   - indices are converted 1-based -> 0-based and range-checked;
 *   - for `symmetric` files every off-diagonal entry is followed directly by
 *     its mirror (j, i, v); skew-symmetric / hermitian files are NOT expanded
 *     (the reference tests mm_is_symmetric only);
 *   - 0 on success, -1 on any failure.
 *
 * The implementation is different: the file is slurped once and tokenised
 * with strtol/strtod (the conversion routines scanf itself uses, so values
 * are bit-identical) instead of one fscanf call per entry; files above 1 MiB
 * are tokenised by all OpenMP threads (see parse_body_parallel).
 */
#include "matrix_parser.h"

#include <ctype.h>
#include <errno.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void init_pre_matrix(PreMatrix *mat) {
    mat->M = 0;
    mat->N = 0;
    mat->nz = 0;
    mat->I = NULL;
    mat->J = NULL;
    mat->val = NULL;
}

void free_pre_matrix(PreMatrix *mat) {
    free(mat->I);
    free(mat->J);
    free(mat->val);
    init_pre_matrix(mat);
}

/* rest of the stream after the size line, NUL-terminated */
static char *slurp_rest(FILE *f, size_t *len_out) {
    long here = ftell(f);
    if (here < 0 || fseek(f, 0, SEEK_END) != 0) return NULL;
    long end = ftell(f);
    if (end < here || fseek(f, here, SEEK_SET) != 0) return NULL;
    size_t len = (size_t)(end - here);
    char *buf = (char *)malloc(len + 1);
    if (!buf) return NULL;
    size_t got = fread(buf, 1, len, f);
    buf[got] = '\0';
    *len_out = got;
    return buf;
}

static inline const char *skip_ws(const char *p) {
    while (*p && isspace((unsigned char)*p)) ++p;
    return p;
}

/* parse one decimal int the way scanf("%d") would: optional sign, digits */
static inline int next_int(const char **pp, int *out) {
    const char *p = skip_ws(*pp);
    char *end;
    errno = 0;
    long v = strtol(p, &end, 10);
    if (end == p) return 0;
    *out = (int)v;
    *pp = end;
    return 1;
}

static inline int next_double(const char **pp, double *out) {
    const char *p = skip_ws(*pp);
    char *end;
    double v = strtod(p, &end);
    if (end == p) return 0;
    *out = v;
    *pp = end;
    return 1;
}

/* ---- parallel body parser ---------------------------------
 * The entry list is a whitespace-separated token stream; entry e owns tokens
 * [k*e, k*e + k) with k = 2 (pattern) or 3.  Pass 1 counts the tokens of T
 * newline-free-cut chunks in parallel, a prefix sum gives every chunk its first
 * global token index, pass 2 converts tokens with strtol / strtod (the routines
 * scanf uses, so values are bit-identical to the reference's fscanf) straight
 * into raw per-entry arrays.  For symmetric files a second prefix sum over the
 * off-diagonal counts places each entry and its mirror exactly where the serial
 * reader would.  Output is identical to the serial path for any thread count.
 */
typedef struct {
    int ok;        /* 0 = a token failed to convert */
    int bad_entry; /* first offending entry (for the message) */
} ParseStatus;

static inline int is_ws(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

static size_t count_tokens(const char *p, const char *end) {
    size_t n = 0;
    while (p < end) {
        while (p < end && is_ws(*p)) ++p;
        if (p >= end) break;
        ++n;
        while (p < end && !is_ws(*p)) ++p;
    }
    return n;
}

static int parse_body_parallel(const char *text, size_t len, int file_nz, int k, int M, int N,
                               int *ri, int *rj, double *rv) {
    const int T = omp_get_max_threads();
    size_t *cut = (size_t *)malloc(((size_t)T + 1) * sizeof(size_t));
    size_t *first_tok = (size_t *)malloc(((size_t)T + 1) * sizeof(size_t));
    if (!cut || !first_tok) {
        free(cut);
        free(first_tok);
        return -1;
    }
    /* chunk boundaries moved forward to the next whitespace so no token is split */
    for (int t = 0; t <= T; ++t) {
        size_t c = len / (size_t)T * (size_t)t;
        if (t == T) c = len;
        while (c < len && c > 0 && !is_ws(text[c])) ++c;
        cut[t] = c;
    }
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < T; ++t) first_tok[t + 1] = count_tokens(text + cut[t], text + cut[t + 1]);
    first_tok[0] = 0;
    for (int t = 0; t < T; ++t) first_tok[t + 1] += first_tok[t];
    const size_t need = (size_t)k * (size_t)file_nz;
    int rc = 0;
    if (first_tok[T] < need) {
        const size_t got = first_tok[T];
        printf("Errore di lettura alla riga %zu: letti %zu valori invece di %d\n", got / (size_t)k + 1,
               got % (size_t)k, k);
        rc = -1;
    }
    int failed = 0;
    if (!rc) {
#pragma omp parallel for schedule(static, 1) reduction(| : failed)
        for (int t = 0; t < T; ++t) {
            const char *p = text + cut[t], *end = text + cut[t + 1];
            size_t tok = first_tok[t];
            while (p < end && tok < need && !failed) {
                while (p < end && is_ws(*p)) ++p;
                if (p >= end) break;
                const size_t e = tok / (size_t)k;
                const int field = (int)(tok % (size_t)k);
                char *stop;
                if (field < 2) {
                    const long v = strtol(p, &stop, 10);
                    if (stop == p || (stop < text + len && !is_ws(*stop) && *stop != '\0')) failed = 1;
                    (field == 0 ? ri : rj)[e] = (int)v - 1; /* 1-based -> 0-based */
                } else {
                    const double v = strtod(p, &stop);
                    if (stop == p) failed = 1;
                    rv[e] = v;
                }
                p = stop > p ? stop : p + 1;
                while (p < end && !is_ws(*p)) ++p; /* rest of an over-long token */
                ++tok;
            }
        }
        if (failed) {
            printf("Errore di lettura: valore non numerico nel corpo del file\n");
            rc = -1;
        }
    }
    if (!rc) {
        long long bad = -1;
#pragma omp parallel for schedule(static) reduction(max : bad)
        for (int e = 0; e < file_nz; ++e)
            if (ri[e] < 0 || ri[e] >= M || rj[e] < 0 || rj[e] >= N)
                if (bad < 0 || e < bad) bad = e; /* any one is enough to fail */
        if (bad >= 0) {
            printf("Errore: Indice fuori range (%d,%d) per matrice %dx%d\n", ri[bad] + 1, rj[bad] + 1, M, N);
            rc = -1;
        }
    }
    free(cut);
    free(first_tok);
    return rc;
}

/* serial twin of the above (small files; also the specification of the order) */
static int parse_body_serial(const char *text, int file_nz, int pattern, int M, int N, int *ri,
                             int *rj, double *rv) {
    const char *p = text;
    for (int e = 0; e < file_nz; ++e) {
        int i, j;
        double v = 1.0;
        int got = next_int(&p, &i);
        if (got) got += next_int(&p, &j);
        if (got == 2 && !pattern) got += next_double(&p, &v);
        if (got != (pattern ? 2 : 3)) {
            printf("Errore di lettura alla riga %d: letti %d valori invece di %d\n", e + 1, got,
                   pattern ? 2 : 3);
            return -1;
        }
        --i;
        --j;
        if (i < 0 || i >= M || j < 0 || j >= N) {
            printf("Errore: Indice fuori range (%d,%d) per matrice %dx%d\n", i + 1, j + 1, M, N);
            return -1;
        }
        ri[e] = i;
        rj[e] = j;
        rv[e] = v;
    }
    return 0;
}

#define PARALLEL_PARSE_MIN_BYTES (1u << 20)

int read_matrix_market(const char *filename, PreMatrix *mat) {
    FILE *f = fopen(filename, "r");
    if (!f) {
        printf("Errore nell'apertura del file\n");
        return -1;
    }
    if (mm_read_banner(f, &mat->type) != 0) {
        printf("Formato Matrix Market non riconosciuto.\n");
        fclose(f);
        return -1;
    }
    if (!mm_is_matrix(mat->type) || !mm_is_sparse(mat->type)) {
        printf("Sono supportare solo matrici sparse.\n");
        fclose(f);
        return -1;
    }
    int file_nz = 0;
    if (mm_read_mtx_crd_size(f, &mat->M, &mat->N, &file_nz) != 0 || file_nz < 0) {
        fclose(f);
        return -1;
    }

    const int symmetric = mm_is_symmetric(mat->type);
    const int pattern = mm_is_pattern(mat->type);
    const size_t cap = (size_t)file_nz * (symmetric ? 2u : 1u);
    if (cap > 0x7fffffffu) { /* nz is an int in the kept struct */
        printf("Errore nell'allocazione della memoria\n");
        fclose(f);
        return -1;
    }

    const int verbose = getenv("SPMV_VERBOSE") != NULL;
    double t_phase = omp_get_wtime();
    size_t len = 0;
    char *text = slurp_rest(f, &len);
    fclose(f);
    if (verbose) printf("[parser] read %zu bytes: %.3f s\n", len, omp_get_wtime() - t_phase);
    t_phase = omp_get_wtime();
    const size_t raw_n = file_nz ? (size_t)file_nz : 1;
    int *ri = (int *)malloc(raw_n * sizeof(int));
    int *rj = (int *)malloc(raw_n * sizeof(int));
    double *rv = (double *)malloc(raw_n * sizeof(double));
    int *I = NULL, *J = NULL;
    double *V = NULL;
    if (!text || !ri || !rj || !rv) {
        printf("Errore nell'allocazione della memoria\n");
        goto fail;
    }

    int rc;
    const char *force = getenv("SPMV_PARSE_THREADS"); /* "1" forces the serial path (tests) */
    if (len >= PARALLEL_PARSE_MIN_BYTES && !(force && atoi(force) == 1) && omp_get_max_threads() > 1) {
        if (pattern)
            for (int e = 0; e < file_nz; ++e) rv[e] = 1.0;
        rc = parse_body_parallel(text, len, file_nz, pattern ? 2 : 3, mat->M, mat->N, ri, rj, rv);
    } else {
        rc = parse_body_serial(text, file_nz, pattern, mat->M, mat->N, ri, rj, rv);
    }
    free(text);
    text = NULL;
    if (verbose) printf("[parser] tokenise + convert (%d threads): %.3f s\n", omp_get_max_threads(), omp_get_wtime() - t_phase);
    t_phase = omp_get_wtime();
    if (rc != 0) goto fail;

    if (!symmetric) { /* raw arrays are the result */
        mat->nz = file_nz;
        mat->I = ri;
        mat->J = rj;
        mat->val = rv;
        return 0;
    }

    /* symmetric: every off-diagonal entry is followed directly by its mirror */
    {
        const int T = omp_get_max_threads();
        size_t *start = (size_t *)calloc((size_t)T + 1, sizeof(size_t));
        size_t expanded = 0; /* written by the team itself: the runtime may deliver fewer than T threads */
        if (!start) goto fail;
#pragma omp parallel num_threads(T)
        {
            const int t = omp_get_thread_num(), nt = omp_get_num_threads();
            const int e0 = (int)((long long)file_nz * t / nt), e1 = (int)((long long)file_nz * (t + 1) / nt);
            size_t n = 0;
            for (int e = e0; e < e1; ++e) n += 1 + (ri[e] != rj[e]);
            start[t + 1] = n;
#pragma omp barrier
#pragma omp single
            {
                for (int q = 0; q < nt; ++q) start[q + 1] += start[q];
                expanded = start[nt];
                const size_t total = expanded ? expanded : 1;
                I = (int *)malloc(total * sizeof(int));
                J = (int *)malloc(total * sizeof(int));
                V = (double *)malloc(total * sizeof(double));
            }
            if (I && J && V) {
                size_t at = start[t];
                for (int e = e0; e < e1; ++e) {
                    I[at] = ri[e];
                    J[at] = rj[e];
                    V[at] = rv[e];
                    ++at;
                    if (ri[e] != rj[e]) {
                        I[at] = rj[e];
                        J[at] = ri[e];
                        V[at] = rv[e];
                        ++at;
                    }
                }
            }
        }
        const size_t total = expanded;
        free(start);
        if (!I || !J || !V) {
            printf("Errore di allocazione memoria\n");
            goto fail;
        }
        free(ri);
        free(rj);
        free(rv);
        if (verbose) printf("[parser] symmetric expansion: %.3f s\n", omp_get_wtime() - t_phase);
        mat->nz = (int)total;
        mat->I = I;
        mat->J = J;
        mat->val = V;
        return 0;
    }

fail:
    free(text);
    free(ri);
    free(rj);
    free(rv);
    free(I);
    free(J);
    free(V);
    return -1;
}

void print_pre_matrix(PreMatrix *mat, bool const full_print) {
    char *ts = mm_typecode_to_str(mat->type);
    printf("Dimensioni matrice: %d x %d\n", mat->M, mat->N);
    printf("Numero di elementi non-zero: %d\n", mat->nz);
    printf("Matrix type: %s\n", ts ? ts : "?");
    free(ts);
    if (mat->M > 30 || !full_print) return;
    printf("Indice righe (I): ");
    for (int i = 0; i < mat->nz; i++) printf("%d ", mat->I[i]);
    printf("\nIndice colonne (J): ");
    for (int i = 0; i < mat->nz; i++) printf("%d ", mat->J[i]);
    printf("\nValori: ");
    for (int i = 0; i < mat->nz; i++) printf("%f", mat->val[i]);
    printf("\n");
}
