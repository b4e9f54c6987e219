/*
 * mmio.c -- Matrix Market banner / size-line reader (own implementation of
 * the NIST mmio interface subset declared in include/mmio.h).
 *
 * Behaviour follows what the reference's parser relies on
 * (src/matrix_parser.c:33-54 calling libs/mmio.c:96-217):
 *   - the banner line carries five blank-separated tokens; all but the first
 *     are matched case-insensitively;
 *   - '%' lines after the banner are comments; blank lines before the size
 *     line are skipped;
 *   - the size line is "M N nz".
 */
#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#include "mmio.h"

struct mm_word {
    const char *word;
    char code;
};

static const struct mm_word k_format[] = {{"coordinate", 'C'}, {"array", 'A'}, {NULL, 0}};
static const struct mm_word k_field[] = {
    {"real", 'R'}, {"complex", 'C'}, {"pattern", 'P'}, {"integer", 'I'}, {NULL, 0}};
static const struct mm_word k_symm[] = {
    {"general", 'G'}, {"symmetric", 'S'}, {"hermitian", 'H'}, {"skew-symmetric", 'K'}, {NULL, 0}};

static char mm_lookup(const struct mm_word *table, char *token) {
    for (char *p = token; *p; ++p) *p = (char)tolower((unsigned char)*p);
    for (; table->word; ++table)
        if (strcmp(table->word, token) == 0) return table->code;
    return 0;
}

static const char *mm_reverse(const struct mm_word *table, char code) {
    for (; table->word; ++table)
        if (table->code == code) return table->word;
    return NULL;
}

int mm_is_valid(MM_typecode matcode) {
    if (!mm_is_matrix(matcode)) return 0;
    if (mm_is_dense(matcode) && mm_is_pattern(matcode)) return 0;
    if (mm_is_real(matcode) && mm_is_hermitian(matcode)) return 0;
    if (mm_is_pattern(matcode) && (mm_is_hermitian(matcode) || mm_is_skew(matcode))) return 0;
    return 1;
}

int mm_read_banner(FILE *f, MM_typecode *matcode) {
    char line[MM_MAX_LINE_LENGTH];
    char tok[5][MM_MAX_TOKEN_LENGTH];

    mm_clear_typecode(matcode);
    if (!fgets(line, sizeof line, f)) return MM_PREMATURE_EOF;
    if (sscanf(line, "%63s %63s %63s %63s %63s", tok[0], tok[1], tok[2], tok[3], tok[4]) != 5)
        return MM_PREMATURE_EOF;
    if (strncmp(tok[0], MatrixMarketBanner, strlen(MatrixMarketBanner)) != 0) return MM_NO_HEADER;

    for (char *p = tok[1]; *p; ++p) *p = (char)tolower((unsigned char)*p);
    if (strcmp(tok[1], "matrix") != 0) return MM_UNSUPPORTED_TYPE;
    (*matcode)[0] = 'M';

    char c;
    if (!(c = mm_lookup(k_format, tok[2]))) return MM_UNSUPPORTED_TYPE;
    (*matcode)[1] = c;
    if (!(c = mm_lookup(k_field, tok[3]))) return MM_UNSUPPORTED_TYPE;
    (*matcode)[2] = c;
    if (!(c = mm_lookup(k_symm, tok[4]))) return MM_UNSUPPORTED_TYPE;
    (*matcode)[3] = c;
    return 0;
}

int mm_read_mtx_crd_size(FILE *f, int *M, int *N, int *nz) {
    char line[MM_MAX_LINE_LENGTH];
    *M = *N = *nz = 0;
    /* skip comment lines */
    do {
        if (!fgets(line, sizeof line, f)) return MM_PREMATURE_EOF;
    } while (line[0] == '%');
    /* the first non-comment line is either the size line or blank */
    if (sscanf(line, "%d %d %d", M, N, nz) == 3) return 0;
    for (;;) {
        int got = fscanf(f, "%d %d %d", M, N, nz);
        if (got == EOF) return MM_PREMATURE_EOF;
        if (got == 3) return 0;
    }
}

int mm_write_banner(FILE *f, MM_typecode matcode) {
    char *s = mm_typecode_to_str(matcode);
    if (!s) return MM_COULD_NOT_WRITE_FILE;
    int n = fprintf(f, "%s %s\n", MatrixMarketBanner, s);
    free(s);
    return n < 0 ? MM_COULD_NOT_WRITE_FILE : 0;
}

int mm_write_mtx_crd_size(FILE *f, int M, int N, int nz) {
    return fprintf(f, "%d %d %d\n", M, N, nz) < 0 ? MM_COULD_NOT_WRITE_FILE : 0;
}

char *mm_typecode_to_str(MM_typecode matcode) {
    if (!mm_is_matrix(matcode)) return NULL;
    const char *fmt = mm_reverse(k_format, matcode[1]);
    const char *fld = mm_reverse(k_field, matcode[2]);
    const char *sym = mm_reverse(k_symm, matcode[3]);
    if (!fmt || !fld || !sym) return NULL;
    char buf[MM_MAX_LINE_LENGTH];
    snprintf(buf, sizeof buf, "matrix %s %s %s", fmt, fld, sym);
    char *out = (char *)malloc(strlen(buf) + 1);
    if (out) strcpy(out, buf);
    return out;
}
