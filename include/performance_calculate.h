/*
 * performance_calculate.h -- timing / FLOPS / result-difference metrics.
 *
 * Kept API surface of the reference's libs/performance_calculate.h:10-67
 * (CPU build) merged with its CUDA twin cuda_libs/performance_calculate.cuh
 * (GPU build).  The two reference headers declare the same names with
 * different enumerators and a different computeDifferenceMetrics; C has no
 * overloading, so here:
 *   - MediumPerformanceMetric carries the enumerators of BOTH twins (names
 *     are what callers use; numeric values were never part of the contract);
 *   - computeDifferenceMetrics          = the CPU build's 6-argument form
 *     (src/performance_calculate.c:116-178);
 *   - computeDifferenceMetricsGpu       = the CUDA build's mean-abs/mean-rel
 *     form (cuda_src/performance_calculate.cu:103-148), i.e. the numbers the
 *     reference's GPU CSV reports.
 * Implementation: csrc/host/performance_calculate.c.
 */
#ifndef SPMV_AMD_PERFORMANCE_CALCULATE_H
#define SPMV_AMD_PERFORMANCE_CALCULATE_H

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INITIAL_CAPACITY 100

typedef struct {
    double sum;            /* sum of timed samples (mean = sum / count) */
    double min;            /* smallest sample */
    double max;            /* largest sample */
    double *values;        /* every sample, for median / spread */
    double relative_error; /* accumulated per-iteration mean relative error */
    double absolute_error; /* accumulated per-iteration mean absolute error */
    int count;             /* number of timed samples */
    int capacity;          /* allocated length of values[] */
} MetricStats;

typedef enum {
    /* CPU build (libs/performance_calculate.h:23-31) */
    SERIAL_TIME,
    PARALLEL_CSR_TIME,
    PARALLEL_SIMD_CSR_TIME,
    PARALLEL_HLL_TIME,
    PARALLEL_HLL_SIMD_TIME,
    SERIAL_HLL_TIME,
    /* GPU build (cuda_libs/performance_calculate.cuh:19-29) */
    ROW_CSR_TIME,
    WARP_CSR_TIME,
    ROW_HLL_TIME,
    WARP_HLL_TIME,
    WARP_SHARED_MEMORY_CSR_TIME,
    WARP_SHARED_MEMORY_HLL_TIME,
    /* MI355X kernels that have no reference counterpart */
    STREAM_CSR_TIME,
    ALLGATHER_TIME,
    NUM_METRICS,
} MediumPerformanceMetric;

typedef struct DifferenceMetrics {
    double mean_abs_err;
    double mean_rel_err;
    int significant_diffs;
} DiffMetrics;

typedef struct performance_metrics {
    double time;
    double flops;
    double speedup;
    double efficiency;
} PerformanceMetrics;

/* CPU-build semantics: an element is a "significant difference" iff
 * |d| > abs_tol and |d| / max(|ref|, |res|, rel_tol) > rel_tol; returns the
 * count and the mean relative error of the significant ones; mean_abs_err is
 * always 0 (src/performance_calculate.c:116-178). */
struct DifferenceMetrics computeDifferenceMetrics(const double *ref, const double *res, int n,
                                                  double abs_tol, double rel_tol,
                                                  bool print_summary);

/* GPU-build semantics: mean |d| and mean |d| / max(|ref|, |res|, rel_tol)
 * over ALL elements; significant_diffs = 0
 * (cuda_src/performance_calculate.cu:103-148; its default rel_tol is 1e-4). */
struct DifferenceMetrics computeDifferenceMetricsGpu(const double *ref, const double *res, int n,
                                                     double rel_tol, bool print_summary);

void initialize_metrics(void);
void cleanup_metrics(void);
double get_metric_value(MediumPerformanceMetric type);
double get_relative_error(MediumPerformanceMetric type);
double get_absolute_error(MediumPerformanceMetric type);
void update_medium_metric(MediumPerformanceMetric type, double value);
void reset_medium_time_metrics(void);
DiffMetrics computeAverageErrors(const MediumPerformanceMetric type);
void accumulateErrors(const DiffMetrics *iteration_metrics, const MediumPerformanceMetric type);
double calculate_flops(int nz, double time);
void print_flops(double flops);

/* Declared by the reference (performance_calculate.h:58-62) and defined
 * nowhere in it; defined here because the values[] array makes them cheap. */
double get_metric_stddev(MediumPerformanceMetric type);
double get_metric_variance(MediumPerformanceMetric type);
/* additions: min / median of the timed samples */
double get_metric_min(MediumPerformanceMetric type);
double get_metric_median(MediumPerformanceMetric type);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_AMD_PERFORMANCE_CALCULATE_H */
