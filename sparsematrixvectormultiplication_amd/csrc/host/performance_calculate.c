/*
 * performance_calculate.c -- running timing statistics, FLOPS and the two
 * result-difference measures behind include/performance_calculate.h.
 *
 * Reference semantics being matched: src/performance_calculate.c:13-178 for
 * the CPU build and cuda_src/performance_calculate.cu:16-148 for the GPU
 * build (the twin's computeDifferenceMetrics is exposed here as
 * computeDifferenceMetricsGpu).  Like the reference, the statistics are one
 * process-wide table and are not thread-safe.
 */
#include "performance_calculate.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "utility.h"

static MetricStats g_metrics[NUM_METRICS];

static void zero_stats(MetricStats *m) {
    m->sum = 0.0;
    m->min = DBL_MAX;
    m->max = 0.0;
    m->count = 0;
    m->relative_error = 0.0;
    m->absolute_error = 0.0;
}

void initialize_metrics(void) {
    for (int i = 0; i < NUM_METRICS; ++i) {
        zero_stats(&g_metrics[i]);
        g_metrics[i].capacity = INITIAL_CAPACITY;
        g_metrics[i].values = (double *)malloc(INITIAL_CAPACITY * sizeof(double));
        if (!g_metrics[i].values) {
            fprintf(stderr, "Failed to allocate memory for metrics\n");
            exit(EXIT_FAILURE);
        }
    }
}

void cleanup_metrics(void) {
    for (int i = 0; i < NUM_METRICS; ++i) {
        free(g_metrics[i].values);
        g_metrics[i].values = NULL;
        g_metrics[i].capacity = 0;
    }
}

void reset_medium_time_metrics(void) {
    for (int i = 0; i < NUM_METRICS; ++i) zero_stats(&g_metrics[i]);
}

/* mean of the timed samples (reference: src/performance_calculate.c:36-39) */
double get_metric_value(const MediumPerformanceMetric type) {
    const MetricStats *m = &g_metrics[type];
    return m->count ? m->sum / m->count : 0.0;
}

double get_relative_error(const MediumPerformanceMetric type) {
    return g_metrics[type].count ? g_metrics[type].relative_error : 0.0;
}

double get_absolute_error(const MediumPerformanceMetric type) {
    return g_metrics[type].count ? g_metrics[type].absolute_error : 0.0;
}

void update_medium_metric(const MediumPerformanceMetric type, const double value) {
    MetricStats *m = &g_metrics[type];
    if (!m->values || m->count >= m->capacity) {
        const int cap = m->capacity > 0 ? m->capacity * 2 : INITIAL_CAPACITY;
        double *grown = (double *)realloc(m->values, (size_t)cap * sizeof(double));
        if (!grown) {
            fprintf(stderr, "Failed to reallocate memory for metrics\n");
            exit(EXIT_FAILURE);
        }
        m->values = grown;
        m->capacity = cap;
    }
    m->values[m->count++] = value;
    m->sum += value;
    if (value < m->min) m->min = value;
    if (value > m->max) m->max = value;
}

/* errors are summed over EVERY iteration, warm-up included ... */
void accumulateErrors(const DiffMetrics *iteration_metrics, const MediumPerformanceMetric type) {
    g_metrics[type].absolute_error += iteration_metrics->mean_abs_err;
    g_metrics[type].relative_error += iteration_metrics->mean_rel_err;
}

/* ... and averaged over timed + ITERATION_SKIP iterations
 * (reference: src/performance_calculate.c:58-67) */
DiffMetrics computeAverageErrors(const MediumPerformanceMetric type) {
    DiffMetrics avg = {0.0, 0.0, 0};
    const int n = g_metrics[type].count;
    if (n > 0) {
        avg.mean_abs_err = get_absolute_error(type) / (n + ITERATION_SKIP);
        avg.mean_rel_err = get_relative_error(type) / (n + ITERATION_SKIP);
    }
    return avg;
}

/* 2 flops per stored nonzero (reference: src/performance_calculate.c:98-101);
 * the CSR nz is used for HLL runs as well (main.c:150, main_cuda.cu:594). */
double calculate_flops(const int nz, const double time) { return 2.0 * nz / time; }

void print_flops(double flops) {
    static const char *unit[] = {"FLOPS", "KFLOPS", "MFLOPS", "GFLOPS", "TFLOPS", "PFLOPS", "EFLOPS"};
    int u = 0;
    while (flops >= 1000.0 && u < 6) {
        flops /= 1000.0;
        ++u;
    }
    printf("%.3f %s\n", flops, unit[u]);
}

double get_metric_variance(const MediumPerformanceMetric type) {
    const MetricStats *m = &g_metrics[type];
    if (m->count < 2) return 0.0;
    const double mean = m->sum / m->count;
    double acc = 0.0;
    for (int i = 0; i < m->count; ++i) acc += (m->values[i] - mean) * (m->values[i] - mean);
    return acc / (m->count - 1);
}

double get_metric_stddev(const MediumPerformanceMetric type) { return sqrt(get_metric_variance(type)); }

double get_metric_min(const MediumPerformanceMetric type) {
    return g_metrics[type].count ? g_metrics[type].min : 0.0;
}

static int cmp_double(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

double get_metric_median(const MediumPerformanceMetric type) {
    const MetricStats *m = &g_metrics[type];
    if (!m->count) return 0.0;
    double *tmp = (double *)malloc((size_t)m->count * sizeof(double));
    if (!tmp) return 0.0;
    memcpy(tmp, m->values, (size_t)m->count * sizeof(double));
    qsort(tmp, (size_t)m->count, sizeof(double), cmp_double);
    const double med = (m->count & 1) ? tmp[m->count / 2]
                                      : 0.5 * (tmp[m->count / 2 - 1] + tmp[m->count / 2]);
    free(tmp);
    return med;
}

struct DifferenceMetrics computeDifferenceMetrics(const double *ref, const double *res, int n,
                                                  double abs_tol, double rel_tol,
                                                  bool print_summary) {
    struct DifferenceMetrics out = {0.0, 0.0, 0};
    if (n <= 0) {
        if (print_summary) {
            printf("--- Comparison Summary ---\n");
            printf("Vector size : 0\n");
            printf("Result : PASS (empty vectors)\n");
            printf("----------------------------\n");
        }
        return out;
    }
    double rel_sum = 0.0;
    int hits = 0;
    for (int i = 0; i < n; ++i) {
        const double d = fabs(ref[i] - res[i]);
        if (!(d > abs_tol)) continue; /* below the absolute floor: not a difference */
        const double scale = fmax(fmax(fabs(ref[i]), fabs(res[i])), rel_tol);
        const double rel = d / scale;
        if (rel > rel_tol) {
            rel_sum += rel;
            ++hits;
        }
    }
    out.mean_abs_err = 0.0; /* unused in this formulation, as in the reference */
    out.mean_rel_err = hits ? rel_sum / hits : 0.0;
    out.significant_diffs = hits;
    if (print_summary) {
        printf("--- Comparison Summary ---\n");
        printf("Vector size : %d\n", n);
        printf("Significant differences : %d\n", hits);
        printf("Mean Significant Relative Error : %.10e\n", out.mean_rel_err);
        printf("----------------------------\n");
    }
    return out;
}

struct DifferenceMetrics computeDifferenceMetricsGpu(const double *ref, const double *res, int n,
                                                     double rel_tol, bool print_summary) {
    struct DifferenceMetrics out = {0.0, 0.0, 0};
    if (n <= 0) {
        if (print_summary) {
            printf("--- Comparison Summary ---\n");
            printf("Vector size           : 0\n");
            printf("Result                : PASS (empty vectors)\n");
            printf("---------------------------\n");
        }
        return out;
    }
    double abs_sum = 0.0, rel_sum = 0.0;
    for (int i = 0; i < n; ++i) {
        const double d = fabs(ref[i] - res[i]);
        const double scale = fmax(fmax(fabs(ref[i]), fabs(res[i])), rel_tol);
        abs_sum += d;
        rel_sum += d / scale;
    }
    out.mean_abs_err = abs_sum / n;
    out.mean_rel_err = rel_sum / n;
    if (print_summary) {
        printf("--- Comparison Summary ---\n");
        printf("Vector size           : %d\n", n);
        printf("Mean Absolute Error   : %.10e\n", out.mean_abs_err);
        printf("Mean Relative Error   : %.10e\n", out.mean_rel_err);
        printf("---------------------------\n");
    }
    return out;
}
