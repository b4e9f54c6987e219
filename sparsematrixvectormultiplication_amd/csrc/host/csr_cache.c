/*
 * csr_cache.c -- binary sidecar of a built CSR matrix (include/csr_cache.h).
 *
 * File layout (little endian, the only byte order this code runs on; a tag in the
 * header rejects the other one):
 *   header  (struct below, 96 bytes)
 *   row_ptr (M + 1) x int32
 *   col_idx nz x int32
 *   values  nz x float64
 * Checksums are 64-bit FNV-1a folded over 8-byte words in 8 independent lanes, so
 * verification runs at memory speed rather than a byte at a time.
 */
#define _POSIX_C_SOURCE 200809L
#include "csr_cache.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "matrix_parser.h"
#include "utility.h"

#define CACHE_MAGIC "SPMVCSR1"
#define CACHE_ENDIAN_TAG 0x0102030405060708ull

typedef struct {
    char magic[8];
    uint64_t endian_tag;
    int32_t M, N, nz;
    char type[4];
    int64_t source_size;  /* -1: no stamp */
    int64_t source_mtime_s;
    int64_t source_mtime_ns;
    uint64_t sum_row_ptr, sum_col_idx, sum_values;
    uint64_t reserved[2];
} CacheHeader;

static uint64_t checksum(const void *data, size_t bytes) {
    const uint64_t prime = 0x100000001b3ull;
    uint64_t lane[8];
    for (int k = 0; k < 8; ++k) lane[k] = 0xcbf29ce484222325ull + (uint64_t)k;
    const unsigned char *p = (const unsigned char *)data;
    const size_t words = bytes / 8, body = words / 8 * 8;
    for (size_t w = 0; w < body; w += 8) {
        uint64_t v[8];
        memcpy(v, p + w * 8, 64);
        for (int k = 0; k < 8; ++k) lane[k] = (lane[k] ^ v[k]) * prime;
    }
    uint64_t h = 0xcbf29ce484222325ull;
    for (int k = 0; k < 8; ++k) h = (h ^ lane[k]) * prime;
    for (size_t b = body * 8; b < bytes; ++b) h = (h ^ p[b]) * prime;
    return (h ^ (uint64_t)bytes) * prime;
}

static int stamp_of(const char *path, int64_t *size, int64_t *sec, int64_t *nsec) {
    struct stat st;
    if (!path || stat(path, &st) != 0) return -1;
    *size = (int64_t)st.st_size;
    *sec = (int64_t)st.st_mtim.tv_sec;
    *nsec = (int64_t)st.st_mtim.tv_nsec;
    return 0;
}

int save_csr_binary(const CSRMatrix *csr, const char *path, const char *source_mtx) {
    if (!csr || !path || csr->M < 0 || csr->nz < 0 || !csr->row_ptr) {
        printf("Errore: matrice CSR non valida per il salvataggio binario\n");
        return -1;
    }
    CacheHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, CACHE_MAGIC, 8);
    h.endian_tag = CACHE_ENDIAN_TAG;
    h.M = csr->M;
    h.N = csr->N;
    h.nz = csr->nz;
    memcpy(h.type, csr->type, 4);
    h.source_size = -1;
    if (source_mtx && stamp_of(source_mtx, &h.source_size, &h.source_mtime_s, &h.source_mtime_ns) != 0)
        h.source_size = -1;
    const size_t nz = (size_t)csr->nz, rows = (size_t)csr->M + 1;
    h.sum_row_ptr = checksum(csr->row_ptr, rows * sizeof(int));
    h.sum_col_idx = checksum(csr->col_idx, nz * sizeof(int));
    h.sum_values = checksum(csr->values, nz * sizeof(double));

    /* write to a temporary name and rename: a reader never sees a half-written sidecar */
    const size_t len = strlen(path);
    char *tmp = (char *)malloc(len + 16);
    if (!tmp) return -1;
    snprintf(tmp, len + 16, "%s.tmp%d", path, (int)(h.sum_values & 0xffff));
    FILE *f = fopen(tmp, "wb");
    if (!f) {
        printf("Errore nell'apertura del file %s in scrittura\n", tmp);
        free(tmp);
        return -1;
    }
    int ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(csr->row_ptr, sizeof(int), rows, f) == rows &&
             (nz == 0 || (fwrite(csr->col_idx, sizeof(int), nz, f) == nz &&
                          fwrite(csr->values, sizeof(double), nz, f) == nz));
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp, path) == 0;
    if (!ok) {
        printf("Errore nella scrittura del file %s\n", path);
        remove(tmp);
    }
    free(tmp);
    return ok ? 0 : -1;
}

int load_csr_binary(const char *path, CSRMatrix *csr, const char *source_mtx) {
    if (!path || !csr) return -1;
    init_csr_matrix(csr);
    FILE *f = fopen(path, "rb");
    if (!f) return -1; /* no sidecar: not worth a message */
    CacheHeader h;
    int rc = -1;
    const char *why = "intestazione non valida";
    do {
        if (fread(&h, sizeof h, 1, f) != 1) break;
        if (memcmp(h.magic, CACHE_MAGIC, 8) != 0 || h.endian_tag != CACHE_ENDIAN_TAG) break;
        if (h.M < 0 || h.N < 0 || h.nz < 0) break;
        if (source_mtx) {
            int64_t size, sec, nsec;
            why = "il file sorgente e' cambiato";
            if (stamp_of(source_mtx, &size, &sec, &nsec) != 0) break;
            if (h.source_size != size || h.source_mtime_s != sec || h.source_mtime_ns != nsec) break;
        }
        const size_t nz = (size_t)h.nz, rows = (size_t)h.M + 1;
        struct stat st;
        why = "dimensione del file non coerente";
        if (fstat(fileno(f), &st) != 0 ||
            (uint64_t)st.st_size != sizeof h + rows * sizeof(int) + nz * (sizeof(int) + sizeof(double)))
            break;
        why = "memoria insufficiente";
        csr->row_ptr = (int *)malloc(rows * sizeof(int));
        csr->col_idx = (int *)malloc((nz ? nz : 1) * sizeof(int));
        csr->values = (double *)malloc((nz ? nz : 1) * sizeof(double));
        if (!csr->row_ptr || !csr->col_idx || !csr->values) break;
        why = "lettura incompleta";
        if (fread(csr->row_ptr, sizeof(int), rows, f) != rows) break;
        if (nz && (fread(csr->col_idx, sizeof(int), nz, f) != nz || fread(csr->values, sizeof(double), nz, f) != nz))
            break;
        why = "checksum errato";
        if (checksum(csr->row_ptr, rows * sizeof(int)) != h.sum_row_ptr ||
            checksum(csr->col_idx, nz * sizeof(int)) != h.sum_col_idx ||
            checksum(csr->values, nz * sizeof(double)) != h.sum_values)
            break;
        /* a valid CSR structure, whatever the checksums say */
        why = "struttura CSR non valida";
        int bad = csr->row_ptr[0] != 0 || csr->row_ptr[h.M] != h.nz;
        for (size_t r = 0; r + 1 < rows && !bad; ++r) bad = csr->row_ptr[r + 1] < csr->row_ptr[r];
        for (size_t e = 0; e < nz && !bad; ++e) bad = (unsigned)csr->col_idx[e] >= (unsigned)h.N;
        if (bad) break;
        csr->M = h.M;
        csr->N = h.N;
        csr->nz = h.nz;
        memcpy(csr->type, h.type, 4);
        rc = 0;
    } while (0);
    fclose(f);
    if (rc != 0) {
        printf("Cache CSR %s ignorata: %s\n", path, why);
        free_csr_matrix(csr);
    }
    return rc;
}

int load_csr_cached(const char *mtx_path, CSRMatrix *csr, int *from_cache) {
    if (from_cache) *from_cache = 0;
    if (!mtx_path || !csr) return -1;
    const size_t len = strlen(mtx_path);
    char *side = (char *)malloc(len + 8);
    if (!side) return -1;
    snprintf(side, len + 8, "%s.csrbin", mtx_path);
    struct stat st;
    if (stat(side, &st) == 0 && load_csr_binary(side, csr, mtx_path) == 0) {
        if (from_cache) *from_cache = 1;
        free(side);
        return 0;
    }
    PreMatrix pre;
    if (read_matrix_market(mtx_path, &pre) != 0) {
        free(side);
        return -1;
    }
    const int rc = convert_in_csr(&pre, csr, mtx_path);
    free_pre_matrix(&pre);
    if (rc == 0 && save_csr_binary(csr, side, mtx_path) != 0) {
        /* a read-only matrix directory is not an error: the run simply stays uncached */
    }
    free(side);
    return rc;
}
