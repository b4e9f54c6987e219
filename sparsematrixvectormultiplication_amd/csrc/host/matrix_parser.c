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
 * are bit-identical) instead of one fscanf call per entry.
 */
#include "matrix_parser.h"

#include <ctype.h>
#include <errno.h>
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

    size_t len = 0;
    char *text = slurp_rest(f, &len);
    fclose(f);
    int *I = (int *)malloc((cap ? cap : 1) * sizeof(int));
    int *J = (int *)malloc((cap ? cap : 1) * sizeof(int));
    double *V = (double *)malloc((cap ? cap : 1) * sizeof(double));
    if (!text || !I || !J || !V) {
        printf("Errore nell'allocazione della memoria\n");
        goto fail;
    }

    size_t n = 0;
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
            goto fail;
        }
        --i;
        --j;
        if (i < 0 || i >= mat->M || j < 0 || j >= mat->N) {
            printf("Errore: Indice fuori range (%d,%d) per matrice %dx%d\n", i + 1, j + 1, mat->M,
                   mat->N);
            goto fail;
        }
        I[n] = i;
        J[n] = j;
        V[n] = v;
        ++n;
        if (symmetric && i != j) {
            I[n] = j;
            J[n] = i;
            V[n] = v;
            ++n;
        }
    }
    free(text);

    mat->nz = (int)n;
    /* exact-size arrays, as the reference hands out */
    if (n < cap && n > 0) {
        int *I2 = (int *)realloc(I, n * sizeof(int));
        int *J2 = (int *)realloc(J, n * sizeof(int));
        double *V2 = (double *)realloc(V, n * sizeof(double));
        I = I2 ? I2 : I;
        J = J2 ? J2 : J;
        V = V2 ? V2 : V;
    }
    mat->I = I;
    mat->J = J;
    mat->val = V;
    return 0;

fail:
    free(text);
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
