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

// what RCCL itself says about the communicator (not what the caller asked for): bench.py reports it
extern "C" int spmv_hip_comm_info(int *rank, int *nranks) {
    if (!g_comm) return fail("comm_info: no communicator");
    int r = -1, n = -1;
    NCCL_TRY(ncclCommUserRank(g_comm, &r));
    NCCL_TRY(ncclCommCount(g_comm, &n));
    if (rank) *rank = r;
    if (nranks) *nranks = n;
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
        // so the 7 peer copies of every slice go out over distinct xGMI links at once.
        // An error inside the group still closes it: a group left open would swallow every later
        // RCCL call of the process.
        NCCL_TRY(ncclGroupStart());
        ncclResult_t bad = ncclSuccess;
        int bad_rank = -1;
        for (int r = 0; r < g_comm_size && bad == ncclSuccess; ++r) {
            const size_t count = (size_t)(bounds[r + 1] - bounds[r]);
            if (!count) continue;
            char *slice = (char *)d_y + (size_t)bounds[r] * value_bytes;
            bad = ncclBroadcast(slice, slice, count, dt, r, g_comm, s);
            if (bad != ncclSuccess) bad_rank = r;
        }
        const ncclResult_t closed = ncclGroupEnd();
        if (bad != ncclSuccess)
            return fail("comm_allgatherv: ncclBroadcast(root %d) failed: %s", bad_rank, ncclGetErrorString(bad));
        if (closed != ncclSuccess) return fail("comm_allgatherv: ncclGroupEnd failed: %s", ncclGetErrorString(closed));
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

namespace {

// words [lo, hi) of y <- 0xFFFFFFFF (a NaN pattern for fp32 and fp64 alike)
__global__ __launch_bounds__(kBlock) void poison_words(unsigned *__restrict__ y, long long lo, long long hi) {
    for (long long k = lo + (long long)blockIdx.x * kBlock + threadIdx.x; k < hi; k += (long long)gridDim.x * kBlock)
        y[k] = 0xFFFFFFFFu;
}

// every slice of y this rank does NOT own is overwritten with the poison pattern, so that a gather
// which delivers nothing (or only part) cannot pass for one that did
int poison_peer_slices(void *d_y, const int *bounds, int value_bytes, hipStream_t s) {
    const long long w = value_bytes / 4;
    const long long own_lo = bounds[g_comm_rank] * w, own_hi = bounds[g_comm_rank + 1] * w, end = bounds[g_comm_size] * w;
    if (own_lo > 0) hipLaunchKernelGGL(poison_words, dim3(512), dim3(kBlock), 0, s, (unsigned *)d_y, 0LL, own_lo);
    if (own_hi < end) hipLaunchKernelGGL(poison_words, dim3(512), dim3(kBlock), 0, s, (unsigned *)d_y, own_hi, end);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

// Time both ways of doing the all-gatherv on THIS node (mean of `iters` after 2 warm-ups, maximum
// over ranks) and keep the faster one that is CORRECT.  Correctness of each mode is established from
// a poisoned vector: the gathered y the caller hands in is the reference copy; before a mode's check
// run every slice this rank does not own is overwritten with 0xFF bytes, the mode runs once, and y
// must equal the reference copy word for word (a mode that delivers nothing leaves poison behind).
// Mode 0 failing that check is an error; mode 1 failing it is rejected (ms_modes[1] < 0).
// Collective: every rank must call it with the same bounds.  y must hold a gathered vector already.
//
// No rank may skip a collective the others enter: everything that can fail on one rank alone (allocations,
// the staging buffer of mode 1) happens BEFORE the first collective, and the ranks then agree -- one
// all-reduce of a failure flag -- whether to go on; the same agreement closes every mode, so a rank
// whose launches failed takes all ranks out together with an error instead of leaving them in a
// collective it never joins.
namespace {

// Sum over the ranks of "this rank failed".  >0: some rank failed (all ranks see the same number);
// -1: the agreement itself failed (then nothing more can be said to the peers).
long long agree_failures(unsigned long long *d_flag, bool failed_here) {
    const unsigned long long mine = failed_here ? 1 : 0;
    unsigned long long all = 0;
    hipError_t e = hipMemcpyAsync(d_flag, &mine, sizeof mine, hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);  // `mine` is pageable: the copy is done here
    if (e != hipSuccess) return fail("comm_autotune: agreement copy failed: %s", hipGetErrorString(e));
    const ncclResult_t n = ncclAllReduce(d_flag, d_flag, 1, ncclUint64, ncclSum, g_comm, g_stream);
    if (n != ncclSuccess) return fail("comm_autotune: agreement all-reduce failed: %s", ncclGetErrorString(n));
    e = hipStreamSynchronize(g_stream);
    if (e == hipSuccess) e = hipMemcpy(&all, d_flag, sizeof all, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail("comm_autotune: agreement read-back failed: %s", hipGetErrorString(e));
    return (long long)all;
}

}  // namespace

extern "C" int spmv_hip_comm_autotune(void *d_y, const int *bounds, int value_bytes, int iters, int *mode_out,
                                      float *ms_modes) {
    if (need_device()) return -1;
    if (!g_comm) return fail("comm_autotune: no communicator");
    if (!d_y || !bounds || iters <= 0) return fail("comm_autotune: bad arguments");
    if (value_bytes != 8 && value_bytes != 4) return fail("comm_autotune: value_bytes must be 4 or 8");
    if (g_comm_size > kMaxRanks) return fail("comm_autotune: more than %d ranks", kMaxRanks);
    const size_t bytes = (size_t)bounds[g_comm_size] * value_bytes;
    void *copy = nullptr;
    float *d_ms = nullptr;
    unsigned long long *d_bad = nullptr;  // [0..1]: mismatching words per mode, [2]: the agreement flag
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    float ms[2] = {0, 0};
    unsigned long long bad[2] = {0, 0};
    // the agreement word first: without it this rank cannot even tell the others that it failed
    if (hipMalloc((void **)&d_bad, 3 * sizeof *d_bad) != hipSuccess)
        return fail("comm_autotune: cannot allocate the agreement word (no collective was entered)");
    do {
        // ---- everything that can fail on this rank alone, before any collective
        hipError_t e = hipMalloc(&copy, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) e = hipMalloc((void **)&d_ms, 2 * sizeof(float));
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 3 * sizeof *d_bad, g_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(copy, d_y, bytes, hipMemcpyDeviceToDevice, g_stream);  // the reference
        int local = 0;
        if (e != hipSuccess) local = fail("comm_autotune: setup failed: %s", hipGetErrorString(e));
        if (!local) local = ensure_stage((size_t)g_comm_size * (size_t)std::max<long long>(1, widest_slice(bounds, g_comm_size)) * value_bytes);
        std::string first_error = local ? spmv_hip_last_error() : "";
        long long failed = agree_failures(d_bad + 2, local != 0);
        if (failed < 0) { rc = -1; break; }
        if (failed > 0) {
            rc = local ? fail("%s", first_error.c_str())
                       : fail("comm_autotune: setup failed on %lld other rank(s); no mode was run", failed);
            break;
        }
        for (int mode = 0; mode < 2 && !rc; ++mode) {
            local = 0;
            do {
                // correctness from a poisoned vector
                local = poison_peer_slices(d_y, bounds, value_bytes, g_stream);
                if (!local) local = allgatherv_mode(d_y, bounds, value_bytes, g_stream, mode);
                if (local) break;
                if (bytes)
                    hipLaunchKernelGGL(count_word_mismatches, dim3(512), dim3(kBlock), 0, g_stream, (const unsigned *)copy,
                                       (const unsigned *)d_y, (long long)(bytes / 4), d_bad + mode);
                // whatever the mode did, timing (and the caller afterwards) works on the good vector
                e = hipMemcpyAsync(d_y, copy, bytes, hipMemcpyDeviceToDevice, g_stream);
                for (int i = 0; i < 2 && !local; ++i) local = allgatherv_mode(d_y, bounds, value_bytes, g_stream, mode);
                if (local) break;
                if (e == hipSuccess) e = hipEventRecord(e0, g_stream);
                for (int i = 0; i < iters && !local; ++i) local = allgatherv_mode(d_y, bounds, value_bytes, g_stream, mode);
                if (local) break;
                if (e == hipSuccess) e = hipEventRecord(e1, g_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
                if (e == hipSuccess) e = hipEventElapsedTime(&ms[mode], e0, e1);
                if (e != hipSuccess) local = fail("comm_autotune: timing failed: %s", hipGetErrorString(e));
                ms[mode] /= (float)iters;
            } while (0);
            // every rank closes the mode with the same agreement, whatever happened to it
            first_error = local ? spmv_hip_last_error() : "";
            failed = agree_failures(d_bad + 2, local != 0);
            if (failed < 0) rc = -1;
            else if (failed > 0)
                rc = local ? fail("%s", first_error.c_str())
                           : fail("comm_autotune: mode %d failed on %lld other rank(s)", mode, failed);
        }
        if (rc) break;
        // leave y as it was handed in (a rejected mode may have run last)
        e = hipMemcpyAsync(d_y, copy, bytes, hipMemcpyDeviceToDevice, g_stream);
        // agree across ranks: slowest rank's time per mode, total mismatches per mode
        if (e == hipSuccess) e = hipMemcpyAsync(d_ms, ms, sizeof ms, hipMemcpyHostToDevice, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        // (a failure here is local; the reductions below are still entered so that no peer waits alone)
        const bool copy_failed = e != hipSuccess;
        ncclResult_t n = ncclAllReduce(d_ms, d_ms, 2, ncclFloat, ncclMax, g_comm, g_stream);
        if (n == ncclSuccess) n = ncclAllReduce(d_bad, d_bad, 2, ncclUint64, ncclSum, g_comm, g_stream);
        if (n != ncclSuccess) { rc = fail("comm_autotune: ncclAllReduce failed: %s", ncclGetErrorString(n)); break; }
        if (copy_failed) { rc = fail("comm_autotune: copy failed: %s", hipGetErrorString(e)); break; }
        e = hipStreamSynchronize(g_stream);
        if (e == hipSuccess) e = hipMemcpy(ms, d_ms, sizeof ms, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail("comm_autotune: reduce failed: %s", hipGetErrorString(e)); break; }
        if (bad[0]) {
            rc = fail("comm_autotune: the grouped-broadcast all-gatherv left %llu wrong words in a poisoned y", bad[0]);
            break;
        }
        g_gather_mode = (bad[1] == 0 && ms[1] < ms[0]) ? 1 : 0;
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
        ms_modes[1] = bad[1] ? -1.0f : ms[1];  // negative: mode 1 did not reproduce the gathered vector and is not used
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

// ------------------------------------------------------------- halo exchange
// SURVEY.md 8(f) N4, second half.  In an iterated method a rank does not need the whole of x, only the
// entries its rows' columns touch; on banded matrices that is its own range plus a halo (nlpkkt-like
// matrix on 8 ranks: 14-15 % of x comes from other ranks, against the 87.5 % an all-gather delivers).
// Three pieces: what a handle needs (from its x-window plan), who sends what to whom (pure host
// logic, identical on every rank), and the exchange itself (one group of ncclSend / ncclRecv on
// contiguous segments of the vector: no packing, the segments are in place on both sides).
namespace {

struct halo_segment {
    int peer, lo, hi;  // vector entries [lo, hi) to / from `peer`
};
std::vector<halo_segment> g_halo_send, g_halo_recv;
bool g_halo_ready = false;


// merge sorted, possibly touching ranges; then close the smallest gaps until at most max_ranges remain
void squeeze_ranges(std::vector<std::pair<int, int>> &r, int max_ranges) {
    std::vector<std::pair<int, int>> out;
    for (const auto &p : r) {
        if (!out.empty() && p.first <= out.back().second) out.back().second = std::max(out.back().second, p.second);
        else out.push_back(p);
    }
    while ((int)out.size() > std::max(1, max_ranges)) {
        size_t best = 1;
        for (size_t k = 2; k < out.size(); ++k)
            if (out[k].first - out[k - 1].second < out[best].first - out[best - 1].second) best = k;
        out[best - 1].second = out[best].second;
        out.erase(out.begin() + (long)best);
    }
    r.swap(out);
}

}  // namespace

// The entries of x the handle's rows touch, as at most max_ranges ascending ranges [lo, hi) (128-byte line
// granularity; small gaps are closed when there are more).  From the x-window plan's line lists; a handle
// without a plan reports the whole vector.
static int spmv_hip_csr_needed_ranges_body(const spmv_csr_dev *m, int max_ranges, int *ranges, int *count) {
    if (need_device()) return -1;
    if (!m || !ranges || !count || max_ranges < 1) return fail("csr_needed_ranges: bad arguments");
    std::vector<std::pair<int, int>> r;
    if (m->local_blocks <= 0 || m->num_long > 0) {
        r.emplace_back(0, m->N);  // no plan, or rows outside it: everything
    } else {
        std::vector<int> lines((size_t)m->local_lines);
        HIP_TRY(hipStreamSynchronize(g_stream));
        if (!lines.empty())
            HIP_TRY(hipMemcpy(lines.data(), m->lines, lines.size() * sizeof(int), hipMemcpyDeviceToHost));
        std::sort(lines.begin(), lines.end());
        lines.erase(std::unique(lines.begin(), lines.end()), lines.end());
        const int per_line = kLineBytes / m->value_bytes;
        for (int l : lines) {
            const int lo = l * per_line, hi = (int)std::min<long long>((long long)(l + 1) * per_line, m->N);
            if (!r.empty() && r.back().second == lo) r.back().second = hi;
            else r.emplace_back(lo, hi);
        }
        squeeze_ranges(r, max_ranges);
    }
    *count = (int)r.size();
    for (size_t k = 0; k < r.size(); ++k) {
        ranges[2 * k] = r[k].first;
        ranges[2 * k + 1] = r[k].second;
    }
    return 0;
}

extern "C" int spmv_hip_csr_needed_ranges(const spmv_csr_dev *m, int max_ranges, int *ranges, int *count) {
    return guarded("csr_needed_ranges", [&] { return spmv_hip_csr_needed_ranges_body(m, max_ranges, ranges, count); });
}

// Pure host logic, the same on every rank: rank q owns entries [bounds[q], bounds[q + 1]); rank p needs
// counts[p] ranges at ranges + p * 2 * stride.  Out (triples peer, lo, hi; ascending per peer): what `rank`
// sends (its own entries that a peer needs) and what it receives (entries it needs that a peer owns).
extern "C" int spmv_hip_halo_plan(int ranks, int rank, const int *bounds, const int *counts, const int *ranges,
                                  int stride, int max_segments, int *send, int *nsend, int *recv, int *nrecv) {
    if (ranks < 1 || rank < 0 || rank >= ranks || !bounds || !counts || !ranges || !send || !nsend || !recv || !nrecv)
        return fail("halo_plan: bad arguments");
    int ns = 0, nr = 0;
    auto clip = [&](int owner, int lo, int hi, int &a, int &b) {
        a = std::max(lo, bounds[owner]);
        b = std::min(hi, bounds[owner + 1]);
        return a < b;
    };
    for (int p = 0; p < ranks; ++p) {
        if (p == rank) continue;
        for (int k = 0; k < counts[p]; ++k) {  // what p needs of mine
            int a, b;
            if (!clip(rank, ranges[(p * stride + k) * 2], ranges[(p * stride + k) * 2 + 1], a, b)) continue;
            if (ns >= max_segments) return fail("halo_plan: more than %d segments to send", max_segments);
            send[3 * ns] = p;
            send[3 * ns + 1] = a;
            send[3 * ns + 2] = b;
            ++ns;
        }
    }
    for (int q = 0; q < ranks; ++q) {
        if (q == rank) continue;
        for (int k = 0; k < counts[rank]; ++k) {  // what I need of q's
            int a, b;
            if (!clip(q, ranges[(rank * stride + k) * 2], ranges[(rank * stride + k) * 2 + 1], a, b)) continue;
            if (nr >= max_segments) return fail("halo_plan: more than %d segments to receive", max_segments);
            recv[3 * nr] = q;
            recv[3 * nr + 1] = a;
            recv[3 * nr + 2] = b;
            ++nr;
        }
    }
    *nsend = ns;
    *nrecv = nr;
    return 0;
}

namespace {
constexpr int kHaloRanges = 32;      // ranges a rank publishes
constexpr int kHaloSegments = 4096;  // segments a rank sends / receives
}  // namespace

// Collective: every rank publishes what its handle needs (all-gather of a small fixed-size record), derives
// its send / receive segments with spmv_hip_halo_plan, and keeps them for spmv_hip_comm_halo_exchange.
static int spmv_hip_comm_halo_setup_body(spmv_csr_dev *m, const int *bounds) {
    if (need_device()) return -1;
    if (!g_comm) return fail("comm_halo_setup: no communicator");
    if (!m || !bounds) return fail("comm_halo_setup: NULL argument");
    // the row bounds double as the ownership of x: only meaningful for square matrices
    if (m->M_total != m->N) return fail("comm_halo_setup: needs a square matrix (%d x %d)", m->M_total, m->N);
    g_halo_ready = false;
    constexpr int kRecord = 1 + 2 * kHaloRanges;
    std::vector<int> mine(kRecord, 0), all((size_t)kRecord * g_comm_size, 0);
    if (spmv_hip_csr_needed_ranges(m, kHaloRanges, mine.data() + 1, &mine[0])) return -1;
    int *d_all = nullptr;
    HIP_TRY(hipMalloc((void **)&d_all, all.size() * sizeof(int)));
    int rc = 0;
    do {
        hipError_t e = hipMemcpy(d_all + (size_t)g_comm_rank * kRecord, mine.data(), kRecord * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) { rc = fail("comm_halo_setup: copy failed: %s", hipGetErrorString(e)); break; }
        ncclResult_t n = ncclAllGather(d_all + (size_t)g_comm_rank * kRecord, d_all, kRecord, ncclInt, g_comm, g_stream);
        if (n != ncclSuccess) { rc = fail("comm_halo_setup: ncclAllGather failed: %s", ncclGetErrorString(n)); break; }
        e = hipStreamSynchronize(g_stream);
        if (e == hipSuccess) e = hipMemcpy(all.data(), d_all, all.size() * sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail("comm_halo_setup: gather failed: %s", hipGetErrorString(e)); break; }
        std::vector<int> counts((size_t)g_comm_size), ranges((size_t)g_comm_size * 2 * kHaloRanges);
        for (int p = 0; p < g_comm_size; ++p) {
            counts[p] = all[(size_t)p * kRecord];
            memcpy(&ranges[(size_t)p * 2 * kHaloRanges], &all[(size_t)p * kRecord + 1], 2 * kHaloRanges * sizeof(int));
        }
        std::vector<int> send(3 * kHaloSegments), recv(3 * kHaloSegments);
        int ns = 0, nr = 0;
        rc = spmv_hip_halo_plan(g_comm_size, g_comm_rank, bounds, counts.data(), ranges.data(), kHaloRanges,
                                kHaloSegments, send.data(), &ns, recv.data(), &nr);
        if (rc) break;
        g_halo_send.assign((size_t)ns, halo_segment{});
        g_halo_recv.assign((size_t)nr, halo_segment{});
        for (int k = 0; k < ns; ++k) g_halo_send[k] = halo_segment{send[3 * k], send[3 * k + 1], send[3 * k + 2]};
        for (int k = 0; k < nr; ++k) g_halo_recv[k] = halo_segment{recv[3 * k], recv[3 * k + 1], recv[3 * k + 2]};
        g_halo_ready = true;
    } while (0);
    (void)hipFree(d_all);
    // which of the handle's blocks can run while the halo is travelling
    long long by_blocks[4] = {0, 0, 0, 0};  // interior / boundary blocks, interior / boundary entries
    if (!rc) rc = spmv_hip_csr_split_interior(m, by_blocks);
    // ... and, below block granularity (round 3), which ENTRIES: on a KKT-coupled cut every block also lists lines of the
    // coupling block, which another rank owns (0 % interior blocks), but 13 of a row's 28 entries have their column in
    // the rank's own range -- the handle split by column gives those their own launch ("halo_split" 0: blocks only)
    // Only where the blocks leave most of the product waiting: the column split costs a second pass over the rows
    // (fem-large cut 8 ways: blocks 20.0 + 4.7 us with 94 % interior, columns 20.2 + 7.4; the KKT cut: blocks 0 + 24,
    // columns 14.3 + 18.5 -- profiles/r3_halo_shares.md)
    const bool blocks_do = by_blocks[2] >= by_blocks[3];  // at least half of the entries in interior blocks
    if (!rc && g_halo_split && g_comm_size > 1 && !blocks_do && !m->tiles_only && m->col && m->val)
        rc = spmv_hip_csr_split_columns(m, bounds[g_comm_rank], bounds[g_comm_rank + 1], nullptr);
    return rc;
}

extern "C" int spmv_hip_comm_halo_setup(spmv_csr_dev *m, const int *bounds) {
    return guarded("comm_halo_setup", [&] { return spmv_hip_comm_halo_setup_body(m, bounds); });
}

// values this rank sends / receives per exchange, and the number of peers it talks to
extern "C" int spmv_hip_comm_halo_info(long long *send_values, long long *recv_values, int *peers) {
    if (!g_halo_ready) return fail("comm_halo_info: spmv_hip_comm_halo_setup has not been called");
    long long s = 0, r = 0;
    std::vector<int> seen;
    for (const auto &g : g_halo_send) { s += g.hi - g.lo; seen.push_back(g.peer); }
    for (const auto &g : g_halo_recv) { r += g.hi - g.lo; seen.push_back(g.peer); }
    std::sort(seen.begin(), seen.end());
    seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
    if (send_values) *send_values = s;
    if (recv_values) *recv_values = r;
    if (peers) *peers = (int)seen.size();
    return 0;
}

// The exchange: every segment is contiguous in the vector on both sides, so it is one group of sends and
// receives straight out of / into d_vec (a full-length vector; each rank's own range holds its values).
extern "C" int spmv_hip_comm_halo_exchange(void *d_vec, int value_bytes, void *stream) {
    if (need_device()) return -1;
    if (!g_comm || !g_halo_ready) return fail("comm_halo_exchange: no communicator / no halo plan");
    if (!d_vec || (value_bytes != 4 && value_bytes != 8)) return fail("comm_halo_exchange: bad arguments");
    hipStream_t s = stream ? (hipStream_t)stream : g_stream;
    const ncclDataType_t dt = value_bytes == 8 ? ncclDouble : ncclFloat;
    if (g_halo_send.empty() && g_halo_recv.empty()) return 0;
    NCCL_TRY(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;  // an error inside the group still closes it (see allgatherv_mode)
    for (size_t k = 0; k < g_halo_send.size() && bad == ncclSuccess; ++k) {
        const auto &g = g_halo_send[k];
        bad = ncclSend((char *)d_vec + (size_t)g.lo * value_bytes, (size_t)(g.hi - g.lo), dt, g.peer, g_comm, s);
    }
    for (size_t k = 0; k < g_halo_recv.size() && bad == ncclSuccess; ++k) {
        const auto &g = g_halo_recv[k];
        bad = ncclRecv((char *)d_vec + (size_t)g.lo * value_bytes, (size_t)(g.hi - g.lo), dt, g.peer, g_comm, s);
    }
    const ncclResult_t closed = ncclGroupEnd();
    if (bad != ncclSuccess) return fail("comm_halo_exchange: ncclSend / ncclRecv failed: %s", ncclGetErrorString(bad));
    if (closed != ncclSuccess) return fail("comm_halo_exchange: ncclGroupEnd failed: %s", ncclGetErrorString(closed));
    return 0;
}

namespace {

// own rows only: sum of squares -> part[], then one workgroup folds them into sum[0]
__global__ __launch_bounds__(kBlock) void fold_partials(const double *__restrict__ part, int nparts, double *__restrict__ sum) {
    __shared__ double wave_sum[kBlock / 64];
    double acc = 0;
    for (int k = threadIdx.x; k < nparts; k += kBlock) acc += part[k];
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = wave_sum[0];
        for (int w = 1; w < kBlock / 64; ++w) s += wave_sum[w];
        sum[0] = s;
    }
}

__global__ void norm_from_sum(const double *__restrict__ sum, double *__restrict__ norm) {
    const double nrm = sqrt(sum[0]);
    norm[0] = nrm;
    norm[1] = nrm > 0 ? 1.0 / nrm : 0.0;
}

// One step: y_own = A x on this rank's rows, partial norm, all-reduce, x_own = y_own / norm, halo of x to the
// neighbours.  With a communicator and an interior / boundary split of the handle's blocks
// (spmv_hip_csr_split_interior) the exchange runs on the second stream BESIDE the interior blocks of the next
// product -- they read this rank's own range of x only -- and the boundary blocks wait for it:
//     stream 1:  ... scale x_own | E1 | interior blocks ........ | wait E2 | boundary blocks, norm, all-reduce, scale ...
//     stream 2:                  wait E1 | halo send / recv | E2
// The same kernels on the same blocks with the same x as the serial order: identical bits.
template <typename T>
int power_iterations_halo(spmv_csr_dev *m, int variant, int iters, double *d_part, double *d_sum, double *d_norm,
                          hipEvent_t scaled, hipEvent_t arrived) {
    const long long n = m->M_local;
    const int grid = (int)std::max<long long>(1, std::min<long long>(kNormBlocks, (n + kBlock - 1) / kBlock));
    T *y_own = (T *)m->y + m->row0, *x_own = (T *)m->x + m->row0;
    const bool fast_path = variant == SPMV_CSR_AUTO || variant == SPMV_CSR_STREAM;
    // the column split (own_part / halo_part) replaces the block split where halo setup made one: the product is then
    // ALWAYS the two launches, overlapped or not -- the same bits either way
    const bool col_split = g_comm && m->own_part && m->halo_part && fast_path;
    const bool overlap = g_comm && (m->have_split || col_split) && g_halo_overlap && fast_path;
    bool in_flight = false;  // a halo exchange of the current x is under way on the second stream
    for (int i = 0; i < iters; ++i) {
        if (overlap || col_split) {
            // interior: own x only
            if (col_split ? csr_launch_split(m, 0, m->x, m->y, g_stream) : csr_launch_part(m, 0, m->x, m->y, g_stream)) return -1;
            if (in_flight) HIP_TRY(hipStreamWaitEvent(g_stream, arrived, 0));
            in_flight = false;
            // the rest: boundary blocks + split rows / the entries whose columns other ranks own
            if (col_split ? csr_launch_split(m, 1, m->x, m->y, g_stream) : csr_launch_part(m, 1, m->x, m->y, g_stream)) return -1;
        } else {
            if (csr_launch_any(m, variant, m->x, m->y, g_stream)) return -1;
        }
        hipLaunchKernelGGL((norm2_partial<T>), dim3(grid), dim3(kBlock), 0, g_stream, (const T *)y_own, n, d_part);
        hipLaunchKernelGGL(fold_partials, dim3(1), dim3(kBlock), 0, g_stream, d_part, grid, d_sum);
        if (g_comm) {
            const ncclResult_t nr = ncclAllReduce(d_sum, d_sum, 1, ncclDouble, ncclSum, g_comm, g_stream);
            if (nr != ncclSuccess) return fail("power_iterate_halo: ncclAllReduce failed: %s", ncclGetErrorString(nr));
        }
        hipLaunchKernelGGL(norm_from_sum, dim3(1), dim3(1), 0, g_stream, d_sum, d_norm);
        hipLaunchKernelGGL((scale_into<T>), dim3(grid), dim3(kBlock), 0, g_stream, (const T *)y_own, n, d_norm, x_own);
        if (g_comm) {
            if (overlap && i + 1 < iters) {
                HIP_TRY(hipEventRecord(scaled, g_stream));
                HIP_TRY(hipStreamWaitEvent(g_stream2, scaled, 0));
                if (spmv_hip_comm_halo_exchange(m->x, m->value_bytes, g_stream2)) return -1;
                HIP_TRY(hipEventRecord(arrived, g_stream2));
                in_flight = true;
            } else if (spmv_hip_comm_halo_exchange(m->x, m->value_bytes, g_stream)) {
                return -1;
            }
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

// spmv_hip_csr_power_iterate with the halo exchange instead of the all-gatherv: every rank keeps only its
// own rows of y; the norm is one all-reduce of the ranks' partial sums of squares, each rank scales its
// own range of x and the halo segments of x travel (after spmv_hip_comm_halo_setup).  y holds only this
// rank's rows of the last A x on return, x its own range and its halo.
extern "C" int spmv_hip_csr_power_iterate_halo(spmv_csr_dev *m, int variant, int iters, double *lambda, float *ms_total) {
    if (need_device()) return -1;
    if (!m || iters <= 0) return fail("power_iterate_halo: bad arguments");
    if (m->M_total != m->N) return fail("power_iterate_halo: needs a square matrix (%d x %d)", m->M_total, m->N);
    if (g_comm && !g_halo_ready) return fail("power_iterate_halo: call spmv_hip_comm_halo_setup first");
    double *d_part = nullptr, *d_sum = nullptr, *d_norm = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, scaled = nullptr, arrived = nullptr;
    int rc = 0;
    do {
        hipError_t e = hipMalloc((void **)&d_part, kNormBlocks * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&d_sum, sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&d_norm, 2 * sizeof(double));
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&scaled, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&arrived, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(e0, g_stream);
        if (e != hipSuccess) { rc = fail("power_iterate_halo: setup failed: %s", hipGetErrorString(e)); break; }
        rc = m->value_bytes == 8 ? power_iterations_halo<double>(m, variant, iters, d_part, d_sum, d_norm, scaled, arrived)
                                 : power_iterations_halo<float>(m, variant, iters, d_part, d_sum, d_norm, scaled, arrived);
        if (rc) {
            (void)hipStreamSynchronize(g_stream2);  // nothing of the loop may outlive its events
            (void)hipStreamSynchronize(g_stream);
            break;
        }
        e = hipEventRecord(e1, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        double nrm[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpy(nrm, d_norm, sizeof nrm, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail("power_iterate_halo: run failed: %s", hipGetErrorString(e)); break; }
        if (lambda) *lambda = nrm[0];
        if (ms_total) *ms_total = ms;
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (scaled) (void)hipEventDestroy(scaled);
    if (arrived) (void)hipEventDestroy(arrived);
    (void)hipFree(d_part);
    (void)hipFree(d_sum);
    (void)hipFree(d_norm);
    return rc;
}

// ------------------------------------------------------------- conjugate gradients
// SURVEY.md 8(f) N4, the second iterated skeleton (the reference multiplies by a fixed x; a Krylov method is what
// an SpMV engine is for).  Plain CG for a symmetric positive definite A, x0 = 0:
//     r = b, p = b, rs = r.r;   repeat:  q = A p;  alpha = rs / p.q;  x += alpha p;  r -= alpha q;
//                                        rs' = r.r;  beta = rs' / rs;  p = r + beta p;  rs = rs'
// p is the handle's x (the SpMV input, full length on every rank), q its y (this rank's rows).  Every rank keeps
// its own rows of x, r; the SpMV's exchange is the same as in the power iteration -- all-gatherv of p, or the
// halo exchange when spmv_hip_comm_halo_setup has run and use_halo is set.  Dot products are fixed-order device
// reductions (grid-stride partial sums per workgroup, folded by one workgroup); across ranks the partial sums are
// ALL-GATHERED and added in rank order by every rank, so all ranks hold the same bits whatever reduction tree the
// collective library would pick for an all-reduce.  Scalars stay on the device: no host synchronisation in the loop.
namespace {

constexpr int kCgRs = 0, kCgPq = 1, kCgRsNew = 2, kCgAlpha = 3, kCgBeta = 4, kCgLocal = 5, kCgScalars = 8;

template <typename T>
__global__ __launch_bounds__(kBlock) void dot_partial(const T *__restrict__ a, const T *__restrict__ b, long long n,
                                                      double *__restrict__ part) {
    __shared__ double wave_sum[kBlock / 64];
    double acc = 0;
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += (long long)gridDim.x * kBlock)
        acc += (double)a[k] * (double)b[k];
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = wave_sum[0];
        for (int w = 1; w < kBlock / 64; ++w) s += wave_sum[w];
        part[blockIdx.x] = s;
    }
}

// x += alpha p, r -= alpha q on this rank's rows, and the workgroup's partial of the new r.r
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_update_x_r(long long n, const double *__restrict__ s, const T *__restrict__ p,
                                                        const T *__restrict__ q, T *__restrict__ x, T *__restrict__ r,
                                                        double *__restrict__ part) {
    __shared__ double wave_sum[kBlock / 64];
    const double alpha = s[kCgAlpha];
    double acc = 0;
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += (long long)gridDim.x * kBlock) {
        x[k] = (T)((double)x[k] + alpha * (double)p[k]);
        const T rk = (T)((double)r[k] - alpha * (double)q[k]);
        r[k] = rk;
        acc += (double)rk * (double)rk;
    }
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = wave_sum[0];
        for (int w = 1; w < kBlock / 64; ++w) t += wave_sum[w];
        part[blockIdx.x] = t;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void cg_update_p(long long n, const double *__restrict__ s, const T *__restrict__ r,
                                                      T *__restrict__ p) {
    const double beta = s[kCgBeta];
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += (long long)gridDim.x * kBlock)
        p[k] = (T)((double)r[k] + beta * (double)p[k]);
}

// the ranks' partial sums in rank order -> s[slot]
__global__ void cg_rank_sum(const double *__restrict__ gathered, int ranks, double *__restrict__ s, int slot) {
    double t = 0;
    for (int k = 0; k < ranks; ++k) t += gathered[k];
    s[slot] = t;
}
__global__ void cg_set_alpha(double *__restrict__ s) { s[kCgAlpha] = s[kCgPq] != 0.0 ? s[kCgRs] / s[kCgPq] : 0.0; }
__global__ void cg_set_beta(double *__restrict__ s, double *__restrict__ hist, int k) {
    s[kCgBeta] = s[kCgRs] != 0.0 ? s[kCgRsNew] / s[kCgRs] : 0.0;
    s[kCgRs] = s[kCgRsNew];
    if (hist) hist[k] = s[kCgRsNew];
}
__global__ void cg_record(const double *__restrict__ s, double *__restrict__ hist) { hist[0] = s[kCgRs]; }

// part[0 .. grid) of this rank -> the global sum in s[slot] on every rank
int cg_reduce(double *d_part, int grid, double *d_s, double *d_gath, int slot) {
    if (!g_comm) {
        hipLaunchKernelGGL(fold_partials, dim3(1), dim3(kBlock), 0, g_stream, d_part, grid, d_s + slot);
        return 0;
    }
    hipLaunchKernelGGL(fold_partials, dim3(1), dim3(kBlock), 0, g_stream, d_part, grid, d_s + kCgLocal);
    const ncclResult_t n = ncclAllGather(d_s + kCgLocal, d_gath, 1, ncclDouble, g_comm, g_stream);
    if (n != ncclSuccess) return fail("csr_cg: ncclAllGather failed: %s", ncclGetErrorString(n));
    hipLaunchKernelGGL(cg_rank_sum, dim3(1), dim3(1), 0, g_stream, d_gath, g_comm_size, d_s, slot);
    return 0;
}

int cg_exchange_p(spmv_csr_dev *m, const int *bounds, int use_halo) {
    if (!g_comm) return 0;
    if (use_halo) return spmv_hip_comm_halo_exchange(m->x, m->value_bytes, g_stream);
    return spmv_hip_comm_allgatherv(m->x, bounds, m->value_bytes, g_stream);
}

template <typename T>
int cg_run(spmv_csr_dev *m, int variant, int iters, const int *bounds, int use_halo, T *d_xs, T *d_r, double *d_s,
           double *d_part, double *d_gath, double *d_hist) {
    const long long n = m->M_local;
    const int grid = (int)std::max<long long>(1, std::min<long long>(kNormBlocks, (n + kBlock - 1) / kBlock));
    T *p_own = (T *)m->x + m->row0, *q_own = (T *)m->y + m->row0, *x_own = d_xs + m->row0;
    // rs = r.r with r = b (already in d_r and in p's own range); every rank gets the whole p
    hipLaunchKernelGGL((dot_partial<T>), dim3(grid), dim3(kBlock), 0, g_stream, (const T *)d_r, (const T *)d_r, n, d_part);
    if (cg_reduce(d_part, grid, d_s, d_gath, kCgRs)) return -1;
    hipLaunchKernelGGL(cg_record, dim3(1), dim3(1), 0, g_stream, d_s, d_hist);
    if (cg_exchange_p(m, bounds, use_halo)) return -1;
    for (int k = 0; k < iters; ++k) {
        if (csr_launch_any(m, variant, m->x, m->y, g_stream)) return -1;  // q = A p on this rank's rows
        hipLaunchKernelGGL((dot_partial<T>), dim3(grid), dim3(kBlock), 0, g_stream, (const T *)p_own, (const T *)q_own, n, d_part);
        if (cg_reduce(d_part, grid, d_s, d_gath, kCgPq)) return -1;
        hipLaunchKernelGGL(cg_set_alpha, dim3(1), dim3(1), 0, g_stream, d_s);
        hipLaunchKernelGGL((cg_update_x_r<T>), dim3(grid), dim3(kBlock), 0, g_stream, n, d_s, (const T *)p_own,
                           (const T *)q_own, x_own, d_r, d_part);
        if (cg_reduce(d_part, grid, d_s, d_gath, kCgRsNew)) return -1;
        hipLaunchKernelGGL(cg_set_beta, dim3(1), dim3(1), 0, g_stream, d_s, d_hist, k + 1);
        hipLaunchKernelGGL((cg_update_p<T>), dim3(grid), dim3(kBlock), 0, g_stream, n, d_s, (const T *)d_r, p_own);
        if (cg_exchange_p(m, bounds, use_halo)) return -1;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T>
int cg_body(spmv_csr_dev *m, int variant, int iters, const int *bounds, int use_halo, const void *b_host, void *x_host,
            double *rr_hist, float *ms_total) {
    const size_t n_all = (size_t)m->M_total, n_own = (size_t)m->M_local;
    T *d_xs = nullptr, *d_r = nullptr;
    double *d_s = nullptr, *d_part = nullptr, *d_gath = nullptr, *d_hist = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    do {
        hipError_t e = hipMalloc((void **)&d_xs, std::max<size_t>(n_all, 1) * sizeof(T));
        if (e == hipSuccess) e = hipMalloc((void **)&d_r, std::max<size_t>(n_own, 1) * sizeof(T));
        if (e == hipSuccess) e = hipMalloc((void **)&d_s, kCgScalars * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&d_part, kNormBlocks * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&d_gath, kMaxRanks * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&d_hist, ((size_t)iters + 1) * sizeof(double));
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) e = hipMemsetAsync(d_xs, 0, std::max<size_t>(n_all, 1) * sizeof(T), g_stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_s, 0, kCgScalars * sizeof(double), g_stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_hist, 0, ((size_t)iters + 1) * sizeof(double), g_stream);
        // r = b on this rank's rows; p = b: the own range of the handle's x (the rest arrives by the exchange)
        if (e == hipSuccess && n_own)
            e = hipMemcpyAsync(d_r, (const T *)b_host + m->row0, n_own * sizeof(T), hipMemcpyHostToDevice, g_stream);
        if (e == hipSuccess) e = hipMemsetAsync(m->x, 0, (size_t)m->N * sizeof(T), g_stream);
        if (e == hipSuccess && n_own)
            e = hipMemcpyAsync((T *)m->x + m->row0, d_r, n_own * sizeof(T), hipMemcpyDeviceToDevice, g_stream);
        if (e == hipSuccess) e = hipEventRecord(e0, g_stream);
        if (e != hipSuccess) { rc = fail("csr_cg: setup failed: %s", hipGetErrorString(e)); break; }
        rc = cg_run<T>(m, variant, iters, bounds, use_halo, d_xs, d_r, d_s, d_part, d_gath, d_hist);
        if (rc) {
            (void)hipStreamSynchronize(g_stream);
            break;
        }
        e = hipEventRecord(e1, g_stream);
        // the solution: every rank holds its rows; with a communicator all rows everywhere
        if (e == hipSuccess && g_comm && x_host) {
            if (spmv_hip_comm_allgatherv(d_xs, bounds, m->value_bytes, g_stream)) { rc = -1; break; }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && x_host) e = hipMemcpy(x_host, d_xs, n_all * sizeof(T), hipMemcpyDeviceToHost);
        if (e == hipSuccess && rr_hist) e = hipMemcpy(rr_hist, d_hist, ((size_t)iters + 1) * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail("csr_cg: run failed: %s", hipGetErrorString(e)); break; }
        if (ms_total) *ms_total = ms;
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_xs);
    (void)hipFree(d_r);
    (void)hipFree(d_s);
    (void)hipFree(d_part);
    (void)hipFree(d_gath);
    (void)hipFree(d_hist);
    return rc;
}

}  // namespace

extern "C" int spmv_hip_csr_cg(spmv_csr_dev *m, int variant, int iters, const int *bounds, int use_halo,
                               const void *b_host, void *x_host, double *rr_hist, float *ms_total) {
    if (need_device()) return -1;
    if (!m || iters < 0 || !b_host) return fail("csr_cg: bad arguments");
    if (m->M_total != m->N) return fail("csr_cg: needs a square matrix (%d x %d)", m->M_total, m->N);
    if (g_comm && !bounds) return fail("csr_cg: a communicator exists, the row bounds are required");
    if (g_comm && use_halo && !g_halo_ready) return fail("csr_cg: call spmv_hip_comm_halo_setup first");
    if (g_comm_size > kMaxRanks) return fail("csr_cg: more than %d ranks", kMaxRanks);
    return guarded("csr_cg", [&] {
        return m->value_bytes == 8 ? cg_body<double>(m, variant, iters, bounds, use_halo, b_host, x_host, rr_hist, ms_total)
                                   : cg_body<float>(m, variant, iters, bounds, use_halo, b_host, x_host, rr_hist, ms_total);
    });
}
