/*
 * spmv_bench.c -- C benchmark driver over libspmv_amd.so.
 *
 * Plays the role of the reference's main_cuda.cu (its protocol, its CSV), written
 * from scratch against include/spmv_hip.h:
 *   for every .mtx in a directory (or one file): parse -> CSR + HLL -> x = 1 ->
 *   every GPU kernel 5 warm-up + 95 timed runs (ITERATION_SKIP, main_cuda.cu:17,95),
 *   kernel-only HIP-event time, y copied back and compared after EVERY run
 *   (main_cuda.cu:183-187), mean time / 2*nnz/t FLOPS / mean errors to
 *     <out>/spmv_results_hip.csv            (the reference's GPU schema, unchanged)
 *     <out>/spmv_results_hip_roofline.csv   (algorithmic GB/s, % of 8 TB/s)
 *     <out>/spmv_results_hip_block_dim.csv  (the reference's launch-shape schema, main_cuda.cu:728)
 *     <out>/spmv_results_hip_launch_shape.csv (lanes per row, stage, workgroups, kernel chosen)
 *
 * The comparison vector is the reference's serial CSR result.  That kernel is the
 * ORACLE (oracle/cpu_spmv.c) and is deliberately not part of libspmv_amd.so, so
 * the driver takes it from a checker library given with --oracle <liboracle_spmv.so>
 * (dlopen); without it the GPU thread-per-row result stands in and the CPU columns
 * are written as 0.  There is no CPU fallback for the GPU columns.
 *
 *   spmv_bench [--oracle lib.so] [--out dir] [--iters 95] <file.mtx | directory>
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#include "csr_matrix.h"
#include "hll_matrix.h"
#include "csr_cache.h"
#include "matrix_parser.h"
#include "performance_calculate.h"
#include "spmv_hip.h"
#include "utility.h"

typedef void (*serial_csr_fn)(int, const int *, const int *, const double *, const double *, double *);
typedef void (*serial_hll_fn)(int, const ELLPACKBlock *, const double *, double *);

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    double time, flops;
    DiffMetrics err;
} Result;

/* one kernel, reference protocol: iterations 1..iters+SKIP-1, the first SKIP untimed */
static int run_gpu(int is_hll, void *dev, int variant, MediumPerformanceMetric slot, int iters,
                   const double *y_ref, double *y_gpu, int M, int nz, Result *out) {
    reset_medium_time_metrics();
    for (int i = 1; i < iters + ITERATION_SKIP; ++i) {
        float ms = 0.f;
        int rc = is_hll ? spmv_hip_hll_time((spmv_hll_dev *)dev, variant, 0, 1, 1, &ms)
                        : spmv_hip_csr_time((spmv_csr_dev *)dev, variant, 0, 1, 1, &ms);
        if (!rc) rc = is_hll ? spmv_hip_hll_get_y((spmv_hll_dev *)dev, y_gpu)
                             : spmv_hip_csr_get_y((spmv_csr_dev *)dev, y_gpu);
        if (rc) {
            fprintf(stderr, "GPU run failed: %s\n", spmv_hip_last_error());
            return -1;
        }
        DiffMetrics d = computeDifferenceMetricsGpu(y_ref, y_gpu, M, 1e-4, false);
        accumulateErrors(&d, slot);
        if (i > ITERATION_SKIP) update_medium_metric(slot, (double)ms / 1000.0);
    }
    out->time = get_metric_value(slot);
    out->flops = calculate_flops(nz, out->time);
    out->err = computeAverageErrors(slot);
    return 0;
}

/* --hll-on-device: HLL is built on the GPU from the resident CSR (spmv_hip_hll_from_csr);
 * the host builder then only runs when the checker's serial HLL pass needs its blocks */
static int g_hll_on_device = 0;
/* --cache: with --hll-on-device and no --oracle, CSR comes from "<file>.csrbin" when fresh */
static int g_cache = 0;
/* --csr-on-device: with --hll-on-device and no --oracle, the parsed triplets go straight to the GPU
 * (spmv_hip_csr_from_coo): no convert_in_csr on the host */
static int g_csr_on_device = 0;

static int bench_matrix(const char *path, const char *name, const char *out_dir, int iters,
                        serial_csr_fn serial_csr, serial_hll_fn serial_hll) {
    PreMatrix pre;
    CSRMatrix csr;
    HLLMatrix hll;
    const int host_hll = !g_hll_on_device || serial_hll != NULL;
    memset(&hll, 0, sizeof hll);
    init_pre_matrix(&pre);
    const int device_csr = g_csr_on_device && !host_hll && serial_csr == NULL;
    init_csr_matrix(&csr);
    if (device_csr) {
        if (process_matrix_file(path, &pre) != 0) return -1;
        csr.M = pre.M; /* only the sizes are used below; the arrays stay on the device */
        csr.N = pre.N;
        csr.nz = pre.nz;
    } else if (g_cache && !host_hll) {
        /* nothing downstream needs the COO triplets: take the built CSR from its sidecar */
        int hit = 0;
        if (load_csr_cached(path, &csr, &hit) != 0) return -1;
        printf("%s: CSR %s\n", name, hit ? "read from the .csrbin sidecar" : "parsed and built (sidecar written)");
    } else {
        if (process_matrix_file(path, &pre) != 0) return -1;
        if (convert_in_csr(&pre, &csr, name) != 0) { free_pre_matrix(&pre); return -1; }
    }
    if (host_hll && convert_to_hll(&pre, &hll) != 0) { free_csr_matrix(&csr); free_pre_matrix(&pre); return -1; }
    const int M = csr.M, N = csr.N, nz = csr.nz;
    const size_t padded_rows = (size_t)((M + HACK_SIZE - 1) / HACK_SIZE) * HACK_SIZE + 1;
    double *x = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    double *y_ref = (double *)calloc(padded_rows, sizeof(double));
    double *y_gpu = (double *)calloc((size_t)(M > 0 ? M : 1), sizeof(double));
    double *y_hll = (double *)calloc(padded_rows, sizeof(double));
    init_vector_at_one(x, N);

    spmv_csr_dev *dcsr = NULL;
    spmv_hll_dev *dhll = NULL;
    int rc = (device_csr ? spmv_hip_csr_from_coo(pre.M, pre.N, pre.nz, pre.I, pre.J, pre.val, &dcsr)
                         : spmv_hip_csr_upload_matrix(&csr, &dcsr)) ||
             spmv_hip_csr_set_x(dcsr, x) ||
             (g_hll_on_device ? spmv_hip_hll_from_csr(dcsr, &dhll) : spmv_hip_hll_upload(&hll, M, N, &dhll)) ||
             spmv_hip_hll_set_x(dhll, x);
    if (rc) {
        fprintf(stderr, "%s: GPU setup failed: %s\n", name, spmv_hip_last_error());
        goto done;
    }

    /* serial CPU runs (the checker library, if given) */
    Result serial = {0}, serial_h = {0};
    if (serial_csr) {
        reset_medium_time_metrics();
        for (int i = 1; i < iters + ITERATION_SKIP; ++i) {
            memset(y_ref, 0, sizeof(double) * (size_t)M);
            const double t = now_s();
            serial_csr(M, csr.row_ptr, csr.col_idx, csr.values, x, y_ref);
            if (i > ITERATION_SKIP) update_medium_metric(SERIAL_TIME, now_s() - t);
        }
        serial.time = get_metric_value(SERIAL_TIME);
        serial.flops = calculate_flops(nz, serial.time);
        reset_medium_time_metrics();
        for (int i = 1; i < iters + ITERATION_SKIP; ++i) {
            const double t = now_s();
            serial_hll(hll.num_blocks, hll.blocks, x, y_hll);
            if (i > ITERATION_SKIP) update_medium_metric(SERIAL_HLL_TIME, now_s() - t);
        }
        serial_h.time = get_metric_value(SERIAL_HLL_TIME);
        serial_h.flops = calculate_flops(nz, serial_h.time);
    } else {
        if (spmv_hip_csr_run(dcsr, SPMV_CSR_THREAD_ROW) || spmv_hip_csr_get_y(dcsr, y_ref)) { rc = -1; goto done; }
    }

    Result r_row = {0}, r_wave = {0}, r_sub = {0}, r_stream = {0}, h_row = {0}, h_sub = {0}, h_lds = {0};
    rc = run_gpu(0, dcsr, SPMV_CSR_THREAD_ROW, ROW_CSR_TIME, iters, y_ref, y_gpu, M, nz, &r_row) ||
         run_gpu(0, dcsr, SPMV_CSR_WAVE_ROW, WARP_CSR_TIME, iters, y_ref, y_gpu, M, nz, &r_wave) ||
         run_gpu(0, dcsr, SPMV_CSR_SUBWAVE, WARP_SHARED_MEMORY_CSR_TIME, iters, y_ref, y_gpu, M, nz, &r_sub) ||
         run_gpu(0, dcsr, SPMV_CSR_STREAM, STREAM_CSR_TIME, iters, y_ref, y_gpu, M, nz, &r_stream) ||
         run_gpu(1, dhll, SPMV_HLL_THREAD_ROW, ROW_HLL_TIME, iters, y_ref, y_gpu, M, nz, &h_row) ||
         run_gpu(1, dhll, SPMV_HLL_SUBWAVE, WARP_HLL_TIME, iters, y_ref, y_gpu, M, nz, &h_sub) ||
         run_gpu(1, dhll, SPMV_HLL_LDS, WARP_SHARED_MEMORY_HLL_TIME, iters, y_ref, y_gpu, M, nz, &h_lds);
    if (rc) goto done;

    char file[1024];
    snprintf(file, sizeof file, "%s/spmv_results_hip.csv", out_dir);
    /* reference column meaning: row = lane per row, warp = wavefront (64 lanes) per row,
     * warp_shared = the third kernel of each format (here: sub-wavefront CSR / LDS-staged HLL) */
    write_results_to_csv_gpu(name, M, N, nz, serial.time, serial_h.time, r_row.time, r_wave.time,
                             r_sub.time, h_lds.time, h_row.time, h_sub.time, serial.flops,
                             serial_h.flops, r_row.flops, r_wave.flops, h_row.flops, h_sub.flops,
                             r_sub.flops, h_lds.flops, r_row.err, r_wave.err, r_sub.err, h_row.err,
                             h_sub.err, h_lds.err, file);

    spmv_dev_info ci, hi;
    spmv_hip_csr_info(dcsr, &ci);
    spmv_hip_hll_info(dhll, &hi);
    snprintf(file, sizeof file, "%s/spmv_results_hip_roofline.csv", out_dir);
    FILE *fp = fopen(file, "a+");
    if (fp) {
        fseek(fp, 0, SEEK_END);
        if (ftell(fp) == 0)
            fputs("matrix_name,rows,cols,nonzeros,hll_slots,csr_algo_bytes,hll_algo_bytes,"
                  "time_stream_csr,gflops_stream_csr,gbps_stream_csr,pct_8TBs_stream_csr,"
                  "rel_err_stream_csr,time_lds_hll,gflops_lds_hll,gbps_lds_hll,pct_8TBs_lds_hll,"
                  "kernel_stream_csr,csr_format_bytes,kernel_lds_hll,hll_format_bytes\n", fp);
        const double gb_c = (double)ci.algo_bytes / r_stream.time / 1e9;
        const double gb_h = (double)hi.algo_bytes / h_lds.time / 1e9;
        /* which kernel the fast path resolved to, and the bytes its own arrays amount to */
        fprintf(fp, "%s,%d,%d,%d,%lld,%lld,%lld,%.9f,%.3f,%.1f,%.2f,%.3e,%.9f,%.3f,%.1f,%.2f,%s,%lld,%s,%lld\n", name,
                M, N, nz, hi.slots, ci.algo_bytes, hi.algo_bytes, r_stream.time,
                r_stream.flops / 1e9, gb_c, gb_c / 80.0, r_stream.err.mean_rel_err, h_lds.time,
                h_lds.flops / 1e9, gb_h, gb_h / 80.0,
                ci.stream_kernel == 1 ? "csr_stream_local" : (ci.stream_kernel == 2 ? "csr_stream_short" : (ci.stream_kernel == 3 ? "csr_tile" : "csr_stream")),
                ci.stream_bytes > 0 ? ci.stream_bytes : ci.algo_bytes,
                hi.stream_kernel == 1 ? "hll_lds_local" : (hi.stream_kernel == 2 ? "csr_tile(hll)" : "hll_lds"),
                hi.stream_bytes > 0 ? hi.stream_bytes : hi.algo_bytes);
        fclose(fp);
    }
    /* launch shapes (reference: write_block_result_to_csv, cuda_src/utility.cu:236-261, called at
     * main_cuda.cu:728 with the occupancy API's threads per block).  The row / wave / sub-wave kernels and
     * the x-window and gather stream kernels run 256-thread workgroups (4 wavefronts); csr_tile -- what the
     * "shared" slot (STREAM / LDS) resolves to for matrices without an x-window plan -- runs 512 (8
     * wavefronts).  What really differs per matrix -- lanes per row, stage, workgroups, kernel the fast path
     * resolved to -- goes to a second file. */
    snprintf(file, sizeof file, "%s/spmv_results_hip_block_dim.csv", out_dir);
    write_block_result_to_csv(name, nz, 256, 256, ci.stream_kernel == 3 ? 512 : 256, 256, 256,
                              hi.stream_kernel == 2 ? 512 : 256, file);
    snprintf(file, sizeof file, "%s/spmv_results_hip_launch_shape.csv", out_dir);
    fp = fopen(file, "a+");
    if (fp) {
        fseek(fp, 0, SEEK_END);
        if (ftell(fp) == 0)
            fputs("matrix_name,nonzeros,threads_per_workgroup_stream_csr,csr_lanes_per_row_subwave,csr_stream_kernel,"
                  "csr_stream_workgroups,csr_xwindow_workgroups,csr_xwindow_stage_lines,csr_split_rows,"
                  "hll_lanes_per_row_subwave,hll_lds_kernel,hll_lds_workgroups,hll_xwindow_workgroups,"
                  "hll_xwindow_stage_lines\n", fp);
        fprintf(fp, "%s,%d,%d,%d,%s,%d,%d,%d,%d,%d,%s,%d,%d,%d\n", name, nz, ci.stream_kernel == 3 ? 512 : 256, ci.lanes_per_row,
                ci.stream_kernel == 1 ? "csr_stream_local" : (ci.stream_kernel == 2 ? "csr_stream_short" : (ci.stream_kernel == 3 ? "csr_tile" : "csr_stream")),
                ci.stream_blocks, ci.local_blocks, ci.local_stage_lines, ci.long_rows, hi.lanes_per_row,
                hi.stream_kernel == 1 ? "hll_lds_local" : (hi.stream_kernel == 2 ? "csr_tile(hll)" : "hll_lds"), hi.stream_blocks, hi.local_blocks,
                hi.local_stage_lines);
        fclose(fp);
    }
    printf("%-28s M=%d nnz=%d  csr: row %.1f wave %.1f sub %.1f stream %.1f us | hll: row %.1f sub %.1f lds %.1f us"
           " | stream %.1f GFLOP/s, rel err %.2e\n",
           name, M, nz, r_row.time * 1e6, r_wave.time * 1e6, r_sub.time * 1e6, r_stream.time * 1e6,
           h_row.time * 1e6, h_sub.time * 1e6, h_lds.time * 1e6, r_stream.flops / 1e9,
           r_stream.err.mean_rel_err);
done:
    spmv_hip_csr_free(dcsr);
    spmv_hip_hll_free(dhll);
    free(x); free(y_ref); free(y_gpu); free(y_hll);
    if (host_hll) free_hll_matrix(&hll);
    free_csr_matrix(&csr);
    free_pre_matrix(&pre);
    return rc ? -1 : 0;
}

int main(int argc, char **argv) {
    const char *oracle = NULL, *out_dir = "result", *target = NULL;
    int iters = 95;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--oracle") && i + 1 < argc) oracle = argv[++i];
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out_dir = argv[++i];
        else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--hll-on-device")) g_hll_on_device = 1;
        else if (!strcmp(argv[i], "--cache")) g_cache = 1;
        else if (!strcmp(argv[i], "--csr-on-device")) g_csr_on_device = 1;
        else target = argv[i];
    }
    if (!target) {
        fprintf(stderr, "usage: %s [--oracle liboracle_spmv.so] [--out dir] [--iters n] [--hll-on-device] [--cache] [--csr-on-device] <file.mtx|dir>\n", argv[0]);
        return 2;
    }
    serial_csr_fn serial_csr = NULL;
    serial_hll_fn serial_hll = NULL;
    if (oracle) {
        void *h = dlopen(oracle, RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "cannot load checker %s: %s\n", oracle, dlerror()); return 2; }
        serial_csr = (serial_csr_fn)dlsym(h, "csr_matrix_vector_mult");
        serial_hll = (serial_hll_fn)dlsym(h, "spmv_hll_serial");
        if (!serial_csr || !serial_hll) { fprintf(stderr, "checker lacks the serial kernels\n"); return 2; }
    }
    if (spmv_hip_init(0) != 0) {
        fprintf(stderr, "no usable HIP device: %s\n", spmv_hip_last_error());
        return 1; /* no CPU fallback */
    }
    create_directory(out_dir); /* never wipes earlier results */
    initialize_metrics();
    int failures = 0;
    struct stat st;
    if (stat(target, &st) == 0 && S_ISDIR(st.st_mode)) {
        struct dirent **list;
        const int n = scandir(target, &list, NULL, alphasort);
        for (int i = 0; i < n; ++i) {
            const char *nm = list[i]->d_name;
            const size_t len = strlen(nm);
            if (nm[0] != '.' && len > 4 && !strcmp(nm + len - 4, ".mtx")) {
                char path[2048];
                snprintf(path, sizeof path, "%s/%s", target, nm);
                failures += bench_matrix(path, nm, out_dir, iters, serial_csr, serial_hll) != 0;
            }
            free(list[i]);
        }
        if (n >= 0) free(list);
    } else {
        const char *slash = strrchr(target, '/');
        failures += bench_matrix(target, slash ? slash + 1 : target, out_dir, iters, serial_csr, serial_hll) != 0;
    }
    cleanup_metrics();
    spmv_hip_shutdown();
    return failures ? 1 : 0;
}
