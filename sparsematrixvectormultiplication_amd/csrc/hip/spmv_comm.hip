// spmv_comm.hip -- the multi-GPU side of the C-ABI: row / hack partition bounds (the reference's
// greedy partitioners with num_threads = GPUs), the RCCL communicator, the all-gatherv of y in
// its two implementations with the on-node autotune, and the timed step (kernel + exchange).
#include "spmv_internal.hpp"

// ------------------------------------------------------------- multi-GPU
// Hack ranges for `parts` ranks by the reference's HLL partitioner (K8: greedy over hacks,
// weight = padded slots; src/hll_matrix.c:410-540).  bounds[p] .. bounds[p + 1] are HACK indices.
extern "C" int spmv_hip_partition_hacks(const HLLMatrix *hll, int parts, int *bounds) {
    if (!hll || parts <= 0 || !bounds || hll->num_blocks < 0) return fail("partition_hacks: bad arguments");
    const int H = hll->num_blocks;
    for (int p = 0; p <= parts; ++p) bounds[p] = H;
    bounds[0] = 0;
    if (H == 0) return 0;
    int *start = nullptr, *end = nullptr;
    const int got = prepare_thread_distribution_hll(hll, parts, &start, &end);
    for (int p = 0; p < got; ++p) bounds[p + 1] = (p == got - 1) ? H : start[p + 1];
    for (int p = got + 1; p <= parts; ++p) bounds[p] = H;
    if (got == 0) bounds[1] = H;
    free(start);
    free(end);
    return 0;
}

extern "C" int spmv_hip_partition_rows(int M, const int *row_ptr, int parts, int *bounds) {
    if (M < 0 || parts <= 0 || !bounds || (M > 0 && !row_ptr)) return fail("partition_rows: bad arguments");
    for (int p = 0; p <= parts; ++p) bounds[p] = M;
    bounds[0] = 0;
    if (M == 0) return 0;
    int *start = nullptr, *end = nullptr;
    const long long total = (long long)row_ptr[M] - row_ptr[0];
    const int got = prepare_thread_distribution(M, row_ptr, parts, total, &start, &end);
    // chunks are contiguous and ordered; rows of trailing empty chunks (if any)
    // and rows skipped by dropped zero-nnz chunks go to their left neighbour
    for (int p = 0; p < got; ++p) bounds[p + 1] = (p == got - 1) ? M : start[p + 1];
    for (int p = got + 1; p <= parts; ++p) bounds[p] = M;
    if (got == 0) bounds[1] = M;  // matrix without nonzeros: everything to part 0
    free(start);
    free(end);
    return 0;
}

extern "C" int spmv_hip_comm_get_id(void *id_bytes) {
    if (!id_bytes) return fail("comm_get_id: NULL buffer");
    static_assert(sizeof(ncclUniqueId) <= SPMV_COMM_ID_BYTES, "id buffer too small");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memset(id_bytes, 0, SPMV_COMM_ID_BYTES);
    memcpy(id_bytes, &id, sizeof id);
    return 0;
}

extern "C" int spmv_hip_comm_init(const void *id_bytes, int rank, int nranks) {
    if (need_device()) return -1;
    if (!id_bytes || rank < 0 || rank >= nranks) return fail("comm_init: bad arguments");
    if (g_comm) return fail("comm_init: communicator already exists");
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    NCCL_TRY(ncclCommInitRank(&g_comm, nranks, id, rank));
    g_comm_rank = rank;
    g_comm_size = nranks;
    return 0;
}

extern "C" int spmv_hip_comm_destroy(void) {
    if (g_comm) {
        NCCL_TRY(ncclCommDestroy(g_comm));
        g_comm = nullptr;
    }
    g_comm_rank = 0;
    g_comm_size = 1;
    return 0;
}

namespace {

constexpr int kMaxRanks = 64;
struct gather_bounds {
    int b[kMaxRanks + 1];
};

// staging -> y for every slice but `skip`: slice p holds bounds[p+1] - bounds[p] values at
// staging + p * max_rows values.  Words of 4 bytes (values are 4 or 8 bytes, offsets multiples of 4).
__global__ __launch_bounds__(kBlock) void scatter_staged(const unsigned *__restrict__ stage, unsigned *__restrict__ y,
                                                         gather_bounds bounds, int skip, long long max_rows,
                                                         int words_per_value) {
    const int p = blockIdx.y;
    if (p == skip) return;
    const long long words = (long long)(bounds.b[p + 1] - bounds.b[p]) * words_per_value;
    const unsigned *src = stage + (long long)p * max_rows * words_per_value;
    unsigned *dst = y + (long long)bounds.b[p] * words_per_value;
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < words; k += (long long)gridDim.x * kBlock)
        dst[k] = src[k];
}

__global__ __launch_bounds__(kBlock) void count_word_mismatches(const unsigned *__restrict__ a,
                                                                const unsigned *__restrict__ b, long long words,
                                                                unsigned long long *__restrict__ out) {
    unsigned long long bad = 0;
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < words; k += (long long)gridDim.x * kBlock)
        bad += a[k] != b[k];
    if (bad) atomicAdd(out, bad);
}

void *g_stage = nullptr;    // padded all-gather staging: ranks x max_rows values
size_t g_stage_bytes = 0;

int ensure_stage(size_t bytes) {
    if (bytes <= g_stage_bytes) return 0;
    if (g_stage) (void)hipFree(g_stage);
    g_stage = nullptr;
    g_stage_bytes = 0;
    HIP_TRY(hipMalloc(&g_stage, bytes));
    g_stage_bytes = bytes;
    return 0;
}

long long widest_slice(const int *bounds, int ranks) {
    long long w = 0;
    for (int r = 0; r < ranks; ++r) w = std::max<long long>(w, bounds[r + 1] - bounds[r]);
    return w;
}

int launch_scatter(const void *stage, void *d_y, const int *bounds, int ranks, int skip, long long max_rows,
                   int value_bytes, hipStream_t s) {
    gather_bounds gb;
    for (int r = 0; r <= ranks; ++r) gb.b[r] = bounds[r];
    const long long words = max_rows * (value_bytes / 4);
    const int gx = (int)std::max<long long>(1, std::min<long long>(1024, (words + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(scatter_staged, dim3(gx, ranks), dim3(kBlock), 0, s, (const unsigned *)stage, (unsigned *)d_y,
                       gb, skip, max_rows, value_bytes / 4);
    HIP_TRY(hipGetLastError());
    return 0;
}

int allgatherv_mode(void *d_y, const int *bounds, int value_bytes, hipStream_t s, int mode) {
    const ncclDataType_t dt = value_bytes == 8 ? ncclDouble : ncclFloat;
    if (mode == 0) {
        // RCCL has no all-gather-v: one broadcast per owner, fused into one group
        // so the 7 peer copies of every slice go out over distinct xGMI links at once
        NCCL_TRY(ncclGroupStart());
        for (int r = 0; r < g_comm_size; ++r) {
            const size_t count = (size_t)(bounds[r + 1] - bounds[r]);
            if (!count) continue;
            char *slice = (char *)d_y + (size_t)bounds[r] * value_bytes;
            NCCL_TRY(ncclBroadcast(slice, slice, count, dt, r, g_comm, s));
        }
        NCCL_TRY(ncclGroupEnd());
        return 0;
    }
    // every slice padded to the widest one: a single in-place ncclAllGather over a staging
    // buffer (RCCL's best-tuned collective on the xGMI mesh), then one kernel puts the peers'
    // slices where they belong in y
    const long long max_rows = widest_slice(bounds, g_comm_size);
    if (max_rows == 0) return 0;
    if (ensure_stage((size_t)g_comm_size * (size_t)max_rows * value_bytes)) return -1;
    char *mine = (char *)g_stage + (size_t)g_comm_rank * (size_t)max_rows * value_bytes;
    const size_t own = (size_t)(bounds[g_comm_rank + 1] - bounds[g_comm_rank]) * value_bytes;
    if (own)
        HIP_TRY(hipMemcpyAsync(mine, (char *)d_y + (size_t)bounds[g_comm_rank] * value_bytes, own,
                               hipMemcpyDeviceToDevice, s));
    NCCL_TRY(ncclAllGather(mine, g_stage, (size_t)max_rows, dt, g_comm, s));
    return launch_scatter(g_stage, d_y, bounds, g_comm_size, g_comm_rank, max_rows, value_bytes, s);
}

}  // namespace

extern "C" int spmv_hip_comm_allgatherv(void *d_y, const int *bounds, int value_bytes, void *stream) {
    if (need_device()) return -1;
    if (!g_comm) return fail("comm_allgatherv: no communicator (call spmv_hip_comm_init)");
    if (!d_y || !bounds) return fail("comm_allgatherv: NULL argument");
    if (value_bytes != 8 && value_bytes != 4) return fail("comm_allgatherv: value_bytes must be 4 or 8");
    if (g_comm_size > kMaxRanks) return fail("comm_allgatherv: more than %d ranks", kMaxRanks);
    return allgatherv_mode(d_y, bounds, value_bytes, stream ? (hipStream_t)stream : g_stream, g_gather_mode);
}

// The scatter half of mode 1 on its own (tests; hosts that gather with their own transport).
extern "C" int spmv_hip_comm_scatter_staged(const void *d_stage, void *d_y, const int *bounds, int ranks, int skip_rank,
                                            int value_bytes, void *stream) {
    if (need_device()) return -1;
    if (!d_stage || !d_y || !bounds || ranks <= 0 || ranks > kMaxRanks || (value_bytes != 4 && value_bytes != 8))
        return fail("comm_scatter_staged: bad arguments");
    return launch_scatter(d_stage, d_y, bounds, ranks, skip_rank, widest_slice(bounds, ranks), value_bytes,
                          stream ? (hipStream_t)stream : g_stream);
}

// Time both ways of doing the all-gatherv on THIS node (mean of `iters` after 2 warm-ups, maximum
// over ranks), check that the second reproduces the first bit for bit, and keep the faster one.
// Collective: every rank must call it with the same bounds.  y must hold a gathered vector already.
extern "C" int spmv_hip_comm_autotune(void *d_y, const int *bounds, int value_bytes, int iters, int *mode_out,
                                      float *ms_modes) {
    if (need_device()) return -1;
    if (!g_comm) return fail("comm_autotune: no communicator");
    if (!d_y || !bounds || iters <= 0) return fail("comm_autotune: bad arguments");
    if (g_comm_size > kMaxRanks) return fail("comm_autotune: more than %d ranks", kMaxRanks);
    const size_t bytes = (size_t)bounds[g_comm_size] * value_bytes;
    void *copy = nullptr;
    float *d_ms = nullptr;
    unsigned long long *d_bad = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    float ms[2] = {0, 0};
    unsigned long long bad = 0;
    do {
        hipError_t e = hipMalloc(&copy, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) e = hipMalloc((void **)&d_ms, 2 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&d_bad, sizeof *d_bad);
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e != hipSuccess) { rc = fail("comm_autotune: setup failed: %s", hipGetErrorString(e)); break; }
        for (int mode = 0; mode < 2 && !rc; ++mode) {
            for (int i = 0; i < 2 && !rc; ++i) rc = allgatherv_mode(d_y, bounds, value_bytes, g_stream, mode);
            if (rc) break;
            if (mode == 0) {  // the reference result
                e = hipMemcpyAsync(copy, d_y, bytes, hipMemcpyDeviceToDevice, g_stream);
            } else {          // must be the same words
                e = hipMemsetAsync(d_bad, 0, sizeof *d_bad, g_stream);
                if (e == hipSuccess && bytes)
                    hipLaunchKernelGGL(count_word_mismatches, dim3(512), dim3(kBlock), 0, g_stream,
                                       (const unsigned *)copy, (const unsigned *)d_y, (long long)(bytes / 4), d_bad);
            }
            if (e == hipSuccess) e = hipEventRecord(e0, g_stream);
            for (int i = 0; i < iters && !rc && e == hipSuccess; ++i)
                rc = allgatherv_mode(d_y, bounds, value_bytes, g_stream, mode);
            if (e == hipSuccess) e = hipEventRecord(e1, g_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms[mode], e0, e1);
            if (e != hipSuccess) rc = fail("comm_autotune: timing failed: %s", hipGetErrorString(e));
            ms[mode] /= (float)iters;
        }
        if (rc) break;
        // agree across ranks: slowest rank's time per mode, total mismatches
        e = hipMemcpy(d_ms, ms, sizeof ms, hipMemcpyHostToDevice);
        if (e != hipSuccess) { rc = fail("comm_autotune: copy failed: %s", hipGetErrorString(e)); break; }
        NCCL_TRY(ncclAllReduce(d_ms, d_ms, 2, ncclFloat, ncclMax, g_comm, g_stream));
        NCCL_TRY(ncclAllReduce(d_bad, d_bad, 1, ncclUint64, ncclSum, g_comm, g_stream));
        e = hipStreamSynchronize(g_stream);
        if (e == hipSuccess) e = hipMemcpy(ms, d_ms, sizeof ms, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail("comm_autotune: reduce failed: %s", hipGetErrorString(e)); break; }
        g_gather_mode = (bad == 0 && ms[1] < ms[0]) ? 1 : 0;
    } while (0);
    (void)hipFree(copy);
    (void)hipFree(d_ms);
    (void)hipFree(d_bad);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (mode_out) *mode_out = g_gather_mode;
    if (ms_modes) {
        ms_modes[0] = ms[0];
        ms_modes[1] = bad ? -1.0f : ms[1];  // negative: mode 1 did not reproduce mode 0 and is not used
    }
    return 0;
}

extern "C" int spmv_hip_csr_step_time(spmv_csr_dev *m, int variant, const int *bounds, int warmup,
                                      int iters, float *ms_kernel, float *ms_exchange) {
    if (need_device()) return -1;
    if (!m || !bounds) return fail("csr_step_time: NULL argument");
    return step_loop(m->y, m->value_bytes, bounds, warmup, iters, ms_kernel, ms_exchange,
                     [&] { return csr_launch_any(m, variant, m->x, m->y, g_stream); });
}

// HLL twin: bounds are ROW bounds (32 x the hack bounds of spmv_hip_partition_hacks, the last one M)
extern "C" int spmv_hip_hll_step_time(spmv_hll_dev *m, int variant, const int *bounds, int warmup,
                                      int iters, float *ms_kernel, float *ms_exchange) {
    if (need_device()) return -1;
    if (!m || !bounds) return fail("hll_step_time: NULL argument");
    return step_loop(m->y, 8, bounds, warmup, iters, ms_kernel, ms_exchange,
                     [&] { return hll_launch(m, variant, m->x, m->y, g_stream); });
}


// ------------------------------------------------------------- iterated SpMV
// SURVEY.md 8(f) N4.  The reference multiplies by a fixed x = 1 a hundred times; an iterated
// method feeds y back into x, which is what makes the all-gatherv a real exchange step.  Power
// iteration as the skeleton: y = A x on this rank's rows, all-gatherv(y), x = y / ||y||_2.  The
// norm is computed by every rank over the whole gathered y with a fixed two-stage reduction
// (deterministic, and identical on all ranks: no extra collective), everything stays on the
// library stream with no host synchronisation inside the loop, so on one GPU the whole loop is
// captured into a hipGraph.
namespace {

constexpr int kNormBlocks = 512;

template <typename T>
__global__ __launch_bounds__(kBlock) void norm2_partial(const T *__restrict__ y, long long n, double *__restrict__ part) {
    __shared__ double wave_sum[kBlock / 64];
    double acc = 0;
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += (long long)gridDim.x * kBlock) {
        const double v = (double)y[k];
        acc += v * v;
    }
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = wave_sum[0];
        for (int w = 1; w < kBlock / 64; ++w) s += wave_sum[w];
        part[blockIdx.x] = s;
    }
}

// one workgroup: partial sums in fixed order -> norm[0] = ||y||_2, norm[1] = 1 / ||y||_2 (0 if y = 0)
__global__ __launch_bounds__(kBlock) void norm2_finish(const double *__restrict__ part, int nparts, double *__restrict__ norm) {
    __shared__ double wave_sum[kBlock / 64];
    double acc = 0;
    for (int k = threadIdx.x; k < nparts; k += kBlock) acc += part[k];
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = wave_sum[0];
        for (int w = 1; w < kBlock / 64; ++w) s += wave_sum[w];
        const double nrm = sqrt(s);
        norm[0] = nrm;
        norm[1] = nrm > 0 ? 1.0 / nrm : 0.0;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void scale_into(const T *__restrict__ y, long long n, const double *__restrict__ norm,
                                                     T *__restrict__ x) {
    const double inv = norm[1];
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += (long long)gridDim.x * kBlock)
        x[k] = (T)((double)y[k] * inv);
}

template <typename T>
int power_iterations(spmv_csr_dev *m, int variant, int iters, const int *bounds, double *d_part, double *d_norm) {
    const long long n = m->M_total;
    const int grid = (int)std::max<long long>(1, std::min<long long>(kNormBlocks, (n + kBlock - 1) / kBlock));
    for (int i = 0; i < iters; ++i) {
        if (csr_launch_any(m, variant, m->x, m->y, g_stream)) return -1;
        if (g_comm && bounds && spmv_hip_comm_allgatherv(m->y, bounds, m->value_bytes, g_stream)) return -1;
        hipLaunchKernelGGL((norm2_partial<T>), dim3(grid), dim3(kBlock), 0, g_stream, (const T *)m->y, n, d_part);
        hipLaunchKernelGGL(norm2_finish, dim3(1), dim3(kBlock), 0, g_stream, d_part, grid, d_norm);
        hipLaunchKernelGGL((scale_into<T>), dim3(grid), dim3(kBlock), 0, g_stream, (const T *)m->y, n, d_norm, (T *)m->x);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

// `iters` steps of x <- A x / ||A x||_2 starting from the handle's current x (square matrices;
// every rank holds its row block, bounds = the row partition when a communicator exists, else NULL).
// On return x is the normalised iterate, y the last A x, *lambda the last ||A x||_2 (the dominant
// |eigenvalue| estimate), *ms_total the device time of the loop.  With one GPU and use_graph != 0
// the loop is captured once and replayed as one hipGraph launch.
extern "C" int spmv_hip_csr_power_iterate(spmv_csr_dev *m, int variant, int iters, const int *bounds, int use_graph,
                                          double *lambda, float *ms_total) {
    if (need_device()) return -1;
    if (!m || iters <= 0) return fail("power_iterate: bad arguments");
    if (m->M_total != m->N) return fail("power_iterate: needs a square matrix (%d x %d)", m->M_total, m->N);
    if (g_comm && !bounds) return fail("power_iterate: a communicator exists, the row bounds are required");
    double *d_part = nullptr, *d_norm = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int rc = 0;
    do {
        hipError_t e = hipMalloc((void **)&d_part, kNormBlocks * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&d_norm, 2 * sizeof(double));
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e != hipSuccess) { rc = fail("power_iterate: setup failed: %s", hipGetErrorString(e)); break; }
        auto loop = [&]() {
            return m->value_bytes == 8 ? power_iterations<double>(m, variant, iters, bounds, d_part, d_norm)
                                       : power_iterations<float>(m, variant, iters, bounds, d_part, d_norm);
        };
        if (use_graph && !g_comm) {
            e = hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal);
            if (e != hipSuccess) { rc = fail("power_iterate: capture failed: %s", hipGetErrorString(e)); break; }
            rc = loop();
            e = hipStreamEndCapture(g_stream, &graph);
            if (!rc && e != hipSuccess) rc = fail("power_iterate: capture failed: %s", hipGetErrorString(e));
            if (!rc && (e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)) != hipSuccess)
                rc = fail("power_iterate: hipGraphInstantiate failed: %s", hipGetErrorString(e));
            if (rc) break;
            e = hipEventRecord(e0, g_stream);
            if (e == hipSuccess) e = hipGraphLaunch(exec, g_stream);
        } else {
            e = hipEventRecord(e0, g_stream);
            if (e == hipSuccess) rc = loop();
            if (rc) break;
        }
        if (e == hipSuccess) e = hipEventRecord(e1, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        double nrm[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpy(nrm, d_norm, sizeof nrm, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail("power_iterate: run failed: %s", hipGetErrorString(e)); break; }
        if (lambda) *lambda = nrm[0];
        if (ms_total) *ms_total = ms;
    } while (0);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_part);
    (void)hipFree(d_norm);
    return rc;
}
