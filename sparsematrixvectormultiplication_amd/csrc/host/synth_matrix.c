/*
 * synth_matrix.c -- seeded stand-in matrices (see include/synth_matrix.h).
 * Workload tooling for bench.py and the tests; no reference counterpart.
 */
#include "synth_matrix.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* splitmix64 finaliser */
static inline uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

/* uniform in (-1, 1), the same for (i, j) and (j, i) */
static inline double sym_value(uint64_t seed, int i, int j) {
    const uint64_t lo = (uint64_t)(i < j ? i : j), hi = (uint64_t)(i < j ? j : i);
    const uint64_t h = mix64(seed ^ mix64((hi << 32) | lo));
    return ((double)(h >> 11) + 0.5) * (2.0 / 9007199254740992.0) - 1.0;
}

/* ---- grid stencils --------------------------------------------------- */
typedef struct { int di, dj, dk; } Off;

static int cmp_off(const void *a, const void *b) {
    const Off *p = (const Off *)a, *q = (const Off *)b;
    if (p->dk != q->dk) return p->dk - q->dk;
    if (p->dj != q->dj) return p->dj - q->dj;
    return p->di - q->di;
}

/* 13 points: centre, +-1 and +-2 along each axis */
static int stencil13(Off *o) {
    int n = 0;
    o[n++] = (Off){0, 0, 0};
    for (int s = 1; s <= 2; ++s) {
        o[n++] = (Off){s, 0, 0};  o[n++] = (Off){-s, 0, 0};
        o[n++] = (Off){0, s, 0};  o[n++] = (Off){0, -s, 0};
        o[n++] = (Off){0, 0, s};  o[n++] = (Off){0, 0, -s};
    }
    qsort(o, (size_t)n, sizeof(Off), cmp_off);
    return n;
}

/* 15 points: centre, the 6 face neighbours, the 8 corners */
static int stencil15(Off *o) {
    int n = 0;
    o[n++] = (Off){0, 0, 0};
    o[n++] = (Off){1, 0, 0};  o[n++] = (Off){-1, 0, 0};
    o[n++] = (Off){0, 1, 0};  o[n++] = (Off){0, -1, 0};
    o[n++] = (Off){0, 0, 1};  o[n++] = (Off){0, 0, -1};
    for (int a = -1; a <= 1; a += 2)
        for (int b = -1; b <= 1; b += 2)
            for (int c = -1; c <= 1; c += 2) o[n++] = (Off){a, b, c};
    qsort(o, (size_t)n, sizeof(Off), cmp_off);
    return n;
}

/* all 27 neighbours */
static int stencil27(Off *o) {
    int n = 0;
    for (int c = -1; c <= 1; ++c)
        for (int b = -1; b <= 1; ++b)
            for (int a = -1; a <= 1; ++a) o[n++] = (Off){a, b, c};
    return n; /* already in ascending linear order */
}

static inline int inside(int i, int j, int k, const Off *o, int nx, int ny, int nz) {
    const int a = i + o->di, b = j + o->dj, c = k + o->dk;
    return a >= 0 && a < nx && b >= 0 && b < ny && c >= 0 && c < nz;
}

static inline int count_inside(int i, int j, int k, const Off *o, int n, int nx, int ny, int nz) {
    int c = 0;
    for (int s = 0; s < n; ++s) c += inside(i, j, k, &o[s], nx, ny, nz);
    return c;
}

/* ---- nlpkkt-like ------------------------------------------------------ */
int synth_kkt_rows(int nx, int ny, int nz) {
    const long long n1 = (long long)nx * ny * nz;
    return (nx < 5 || ny < 5 || nz < 5 || 2 * n1 > 0x3fffffff) ? -1 : (int)(2 * n1);
}

int synth_kkt_row_ptr(int nx, int ny, int nz, int *row_ptr) {
    const int M = synth_kkt_rows(nx, ny, nz);
    if (M < 0 || !row_ptr) return -1;
    const int n1 = M / 2;
    Off s13[13], s15[15];
    stencil13(s13);
    stencil15(s15);
    int *len = row_ptr + 1;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < n1; ++p) {
        const int i = p % nx, j = (p / nx) % ny, k = p / (nx * ny);
        const int c = count_inside(i, j, k, s13, 13, nx, ny, nz) +
                      count_inside(i, j, k, s15, 15, nx, ny, nz);
        len[p] = c;       /* [H  B^T] row */
        len[n1 + p] = c;  /* [B  D  ] row */
    }
    row_ptr[0] = 0;
    long long run = 0;
    for (int r = 0; r < M; ++r) {
        run += len[r];
        if (run > 0x7fffffff) return -1;
        row_ptr[r + 1] = (int)run;
    }
    return 0;
}

int synth_kkt_fill(int nx, int ny, int nz, unsigned long long seed, int row0, int row1,
                   const int *row_ptr, int *col_idx, double *values) {
    const int M = synth_kkt_rows(nx, ny, nz);
    if (M < 0 || row0 < 0 || row1 < row0 || row1 > M) return -1;
    const int n1 = M / 2;
    Off s13[13], s15[15];
    stencil13(s13);
    stencil15(s15);
    const int e0 = row_ptr[row0];
#pragma omp parallel for schedule(static)
    for (int r = row0; r < row1; ++r) {
        const int top = r < n1;
        const int p = top ? r : r - n1;
        const int i = p % nx, j = (p / nx) % ny, k = p / (nx * ny);
        int at = row_ptr[r] - e0;
        /* left block (columns < n1): H for top rows, B for bottom rows */
        const Off *left = top ? s13 : s15;
        const int nl = top ? 13 : 15;
        for (int s = 0; s < nl; ++s) {
            if (!inside(i, j, k, &left[s], nx, ny, nz)) continue;
            const int c = p + left[s].di + nx * (left[s].dj + ny * left[s].dk);
            col_idx[at] = c;
            values[at++] = sym_value(seed, r, c);
        }
        /* right block (columns >= n1): B^T for top rows, D for bottom rows */
        const Off *right = top ? s15 : s13;
        const int nr = top ? 15 : 13;
        for (int s = 0; s < nr; ++s) {
            if (!inside(i, j, k, &right[s], nx, ny, nz)) continue;
            const int c = n1 + p + right[s].di + nx * (right[s].dj + ny * right[s].dk);
            col_idx[at] = c;
            values[at++] = sym_value(seed, r, c);
        }
    }
    return 0;
}

/* ---- cant-like -------------------------------------------------------- */
int synth_fem_rows(int gx, int gy, int gz) {
    const long long nodes = (long long)gx * gy * gz;
    return (gx < 3 || gy < 3 || gz < 3 || 3 * nodes > 0x3fffffff) ? -1 : (int)(3 * nodes);
}

int synth_fem_row_ptr(int gx, int gy, int gz, int *row_ptr) {
    const int M = synth_fem_rows(gx, gy, gz);
    if (M < 0 || !row_ptr) return -1;
    Off s27[27];
    stencil27(s27);
    row_ptr[0] = 0;
    long long run = 0;
    for (int q = 0; q < M / 3; ++q) {
        const int i = q % gx, j = (q / gx) % gy, k = q / (gx * gy);
        const int c = 3 * count_inside(i, j, k, s27, 27, gx, gy, gz);
        for (int d = 0; d < 3; ++d) {
            run += c;
            if (run > 0x7fffffff) return -1;
            row_ptr[3 * q + d + 1] = (int)run;
        }
    }
    return 0;
}

int synth_fem_fill(int gx, int gy, int gz, unsigned long long seed, int row0, int row1,
                   const int *row_ptr, int *col_idx, double *values) {
    const int M = synth_fem_rows(gx, gy, gz);
    if (M < 0 || row0 < 0 || row1 < row0 || row1 > M) return -1;
    Off s27[27];
    stencil27(s27);
    const int e0 = row_ptr[row0];
#pragma omp parallel for schedule(static)
    for (int r = row0; r < row1; ++r) {
        const int q = r / 3;
        const int i = q % gx, j = (q / gx) % gy, k = q / (gx * gy);
        int at = row_ptr[r] - e0;
        for (int s = 0; s < 27; ++s) {
            if (!inside(i, j, k, &s27[s], gx, gy, gz)) continue;
            const int qn = q + s27[s].di + gx * (s27[s].dj + gy * s27[s].dk);
            for (int d = 0; d < 3; ++d) {
                const int c = 3 * qn + d;
                col_idx[at] = c;
                values[at++] = sym_value(seed, r, c);
            }
        }
    }
    return 0;
}

/* ---- power-law -------------------------------------------------------- */
static inline double unit(uint64_t h) { return ((double)(h >> 11) + 0.5) / 9007199254740992.0; }

static inline int pl_degree(int n, int max_degree, uint64_t seed, int r) {
    const double u = unit(mix64(seed ^ mix64(0xD1CEULL + (uint64_t)r)));
    double d = 1.08 / u;
    if (d > (double)max_degree) d = (double)max_degree;
    if (d > (double)n) d = (double)n;
    return d < 1.0 ? 1 : (int)d;
}

static inline int pl_column(int n, uint64_t seed, int r, int s) {
    const uint64_t h = mix64(seed ^ mix64(((uint64_t)r << 32) ^ (uint64_t)s ^ 0xC01ULL));
    const double u = unit(mix64(h));
    /* even draws: preferential (u^2 concentrates on small indices), odd: uniform */
    const double pos = (h & 1) ? u : u * u;
    int c = (int)(pos * (double)n);
    return c >= n ? n - 1 : c;
}

static int cmp_int(const void *a, const void *b) {
    const int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* sorted columns of row r (a column may repeat: that is legal CSR and the SpMV
 * kernels sum both entries); the row length is simply the drawn degree, so
 * row_ptr needs no column generation */
static void pl_row(int n, uint64_t seed, int r, int want, int *buf) {
    for (int s = 0; s < want; ++s) buf[s] = pl_column(n, seed, r, s);
    if (want > 1) qsort(buf, (size_t)want, sizeof(int), cmp_int);
}

int synth_powerlaw_row_ptr(int n, int max_degree, unsigned long long seed, int *row_ptr) {
    if (n <= 0 || max_degree <= 0 || !row_ptr) return -1;
    int *len = row_ptr + 1;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) len[r] = pl_degree(n, max_degree, seed, r);
    row_ptr[0] = 0;
    long long run = 0;
    for (int r = 0; r < n; ++r) {
        run += len[r];
        if (run > 0x7fffffff) return -1;
        row_ptr[r + 1] = (int)run;
    }
    return 0;
}

int synth_powerlaw_fill(int n, int max_degree, unsigned long long seed, int row0, int row1,
                        const int *row_ptr, int *col_idx, float *values) {
    if (n <= 0 || row0 < 0 || row1 < row0 || row1 > n) return -1;
    (void)max_degree;
    const int e0 = row_ptr[row0];
#pragma omp parallel for schedule(dynamic, 1024)
    for (int r = row0; r < row1; ++r) {
        const int at = row_ptr[r] - e0;
        const int want = row_ptr[r + 1] - row_ptr[r];
        pl_row(n, seed, r, want, col_idx + at);  /* sort in place in the output */
        for (int s = 0; s < want; ++s)
            values[at + s] = (float)sym_value(seed ^ (uint64_t)s, r, col_idx[at + s]);
    }
    return 0;
}
