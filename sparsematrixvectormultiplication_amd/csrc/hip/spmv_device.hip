// spmv_device.hip -- library state, device bring-up, tuning knobs and the raw memory helpers of
// include/spmv_hip.h.  Stands where the reference's CUDA driver selects and queries the device
// (/root/reference/main_cuda.cu:100-133) and flushes caches between tests (cuda_src/utility.cu:148-175).
// Nothing here computes on the host: if HIP is unusable every entry point returns -1 with a message.
#include "spmv_internal.hpp"

namespace {

// Touch `n` 16-byte words so that L2 and the Infinity Cache are refilled with
// scratch data (answers clear_cache_kernel, cuda_src/utility.cu:140-145).
__global__ __launch_bounds__(kBlock) void flush_kernel(uint4 *__restrict__ buf, size_t n) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (; i < n; i += stride) {
        uint4 w = buf[i];
        w.x += 1u;
        buf[i] = w;
    }
}

// Read-only stream: every lane sums 16-byte non-temporal loads, one value per workgroup leaves the chip.  DEPTH loads in
// flight per lane (4: the probe of rounds 2-3, which the issue side bounds at ~5.7 TB/s; 16: enough to lean on HBM itself).
template <int DEPTH>
__global__ __launch_bounds__(kBlock) void stream_probe_kernel(const uint4 *__restrict__ buf, size_t n,
                                                              unsigned *__restrict__ sink) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u *p = reinterpret_cast<const v4u *>(buf);
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kBlock;
    unsigned acc = 0;
    for (; i + (DEPTH - 1) * stride < n; i += DEPTH * stride) {
        v4u v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) v[k] = __builtin_nontemporal_load(p + i + k * stride);
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) acc += v[k][k & 3];
    }
    for (; i < n; i += stride) acc += __builtin_nontemporal_load(p + i).x;
    if (acc == 0x9e3779b9u) sink[blockIdx.x] = acc;  // never true on the zeroed buffer: keeps the loads alive
}

// What a gathered value costs when every lane of a wave-instruction hits a different line that lives in L2: each
// wavefront issues 8 independent gathers per trip from a table of `mask + 1` elements (the pattern keeps its shape
// and moves over the table).  tools/ubench_gather.hip is the stand-alone original of this probe.
template <typename T>
__global__ __launch_bounds__(kBlock) void gather_probe_kernel(const T *__restrict__ x, const unsigned *__restrict__ idx,
                                                              T *__restrict__ out, int iters, unsigned mask) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    unsigned rot = wave * 977u;
    T acc = 0;
    unsigned my[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) my[j] = idx[j * 64 + lane];
    for (int it = 0; it < iters; ++it) {
        T v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = ((my[j] + rot) & mask) * (unsigned)sizeof(T);
            v[j] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(x) + off);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
        rot += 4099u * 16u;
    }
    if (acc == T(12345.678)) out[0] = acc;  // never true on the zeroed table: keeps the loads alive
}

}  // namespace

// ------------------------------------------------------------------ state
static thread_local char g_error[512] = "";
int g_device = -1;
hipStream_t g_stream = nullptr;
hipStream_t g_stream2 = nullptr;
static void *g_flush_buf = nullptr;
static size_t g_flush_bytes = 0;
ncclComm_t g_comm = nullptr;
int g_comm_rank = 0, g_comm_size = 1;

int g_stream_cap = 0;
int g_stream_block = 256;
int g_stream_nt = 1;
int g_local_nt = -1;
int g_stream_xcd = 0;
int g_gather_mode = 0;
int g_local_cap = 0;
int g_stream_local = 1;
int g_plan_on_device = 1;
int g_stream_kind = -1;
int g_stream_tile = -1;
int g_tile_rows = 0;
int g_tile_lmax = 1536;  // (1024 until round 3: config 5 1016 us at 1024, 1000 at 1536, 1001 at 2048; with the expanded short rows 942 / 927 / 930)
int g_tile_density = 4;
int g_tile_plan_on_device = 1;
int g_place_tries = 12;
int g_tile_mid = 1;
int g_tile_mid_lo = 0;  // 0: auto (48 entries for fp32, 128 for fp64)
int g_tile_gather_ahead = 0;  // measured: no gain (profiles/r3_ab_gather_ahead.txt)
int g_tile_expand = -1;       // plans with gather passes run on an expanded x (tile_expand + csr_tile<.., XE>): auto
int g_tile_probe = 0;
int g_skew_rows = 1;
int g_tile_fit = 1;
int g_tile_streams = 1;
int g_tile_places = 0;
int g_tile_min_pass = 256;
int g_local_patterns = -1;
int g_tile_mid_items = 0;
int g_tile_items = 1008;  // two rounds of the 512 places: 1.222 ms on the power-law matrix against 1.248 with 4096, 1.231 with 504
int g_tile_pack = 1;
int g_tile_long = 1;
int g_halo_overlap = 1;
int g_halo_split = 1;
int g_tile_balance = 1;
int g_pipe_wgs_per_cu = 5;
int g_num_cus = 256;
int g_probe_mask = 1023;
static int g_probe_depth = 4;  // "probe_depth" tuning knob: loads in flight per lane of the stream probe (4 | 16)

int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return -1;
}

int need_device() {
    if (g_device < 0) return fail("spmv_hip_init() has not been called (or failed): no HIP device");
    return 0;
}

// ------------------------------------------------------------------ device
extern "C" int spmv_hip_set_tuning(const char *key, int value);
extern "C" int spmv_hip_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        fail("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return -1;
    }
    return n;
}

extern "C" int spmv_hip_init(int device) {
    int n = spmv_hip_device_count();
    if (n <= 0) return n < 0 ? -1 : fail("no HIP device visible");
    if (device < 0 || device >= n) return fail("device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    if (g_stream && g_device != device) {
        (void)hipStreamDestroy(g_stream);
        g_stream = nullptr;
        if (g_stream2) (void)hipStreamDestroy(g_stream2);
        g_stream2 = nullptr;
    }
    if (!g_stream) HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    if (!g_stream2) HIP_TRY(hipStreamCreateWithFlags(&g_stream2, hipStreamNonBlocking));
    // SPMV_TUNING="key=value,key=value": same keys as spmv_hip_set_tuning (profiling aid)
    if (const char *env = getenv("SPMV_TUNING")) {
        std::string all(env);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(pos, end - pos);
            const size_t eq = item.find('=');
            if (eq != std::string::npos &&
                spmv_hip_set_tuning(item.substr(0, eq).c_str(), atoi(item.c_str() + eq + 1)) != 0)
                return -1;
            pos = end + 1;
        }
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
        g_num_cus = cus;
    g_device = device;
    return 0;
}

extern "C" int spmv_hip_shutdown(void) {
    if (g_comm) {
        (void)ncclCommDestroy(g_comm);
        g_comm = nullptr;
    }
    if (g_flush_buf) {
        (void)hipFree(g_flush_buf);
        g_flush_buf = nullptr;
        g_flush_bytes = 0;
    }
    if (g_stream2) {
        (void)hipStreamDestroy(g_stream2);
        g_stream2 = nullptr;
    }
    if (g_stream) {
        (void)hipStreamDestroy(g_stream);
        g_stream = nullptr;
    }
    g_device = -1;
    return 0;
}

extern "C" int spmv_hip_set_tuning(const char *key, int value) {
    if (!key) return fail("set_tuning: NULL key");
    if (!strcmp(key, "stream_cap")) {
        if (value != 0 && value != 1024 && value != 2048 && value != 3072 && value != 4096 && value != 8192)
            return fail("set_tuning: stream_cap must be 0 (auto), 1024, 2048, 3072, 4096 or 8192");
        g_stream_cap = value;
    } else if (!strcmp(key, "stream_block")) {
        if (value != 256 && value != 512 && value != 1024)
            return fail("set_tuning: stream_block must be 256, 512 or 1024");
        g_stream_block = value;
    } else if (!strcmp(key, "stream_nt")) {
        g_stream_nt = value != 0;
    } else if (!strcmp(key, "stream_xcd")) {
        if (value < -1) return fail("set_tuning: stream_xcd must be -1, 0 or a positive run length");
        g_stream_xcd = value;
    } else if (!strcmp(key, "probe_mask")) {
        g_probe_mask = value;
    } else if (!strcmp(key, "stream_kind")) {
        if ((value < -1 || value > 6) && (value < 10 || value > 17))
            return fail("set_tuning: stream_kind must be -1..6 (or 10..17 for the ablation probes)");
#ifndef SPMV_EXPERIMENTAL
        if (value != -1 && value != 0 && value != 5 && value != 6)
            return fail("set_tuning: stream_kind %d is an experimental kernel; this library was built without "
                        "EXPERIMENTAL=1 (make -C csrc EXPERIMENTAL=1)", value);
#endif
        g_stream_kind = value;
    } else if (!strcmp(key, "stream_tile")) {
        if (value < -1 || value > 1) return fail("set_tuning: stream_tile must be -1 (auto), 0 or 1");
        g_stream_tile = value;
    } else if (!strcmp(key, "tile_lmax")) {
        if (value < 1 || value > 65536) return fail("set_tuning: tile_lmax must be 1..65536");
        g_tile_lmax = value;
    } else if (!strcmp(key, "tile_rows")) {
        if (value != 0 && (value < 256 || value > kTileRowsMax || (value & 255)))
            return fail("set_tuning: tile_rows must be 0 (auto) or a multiple of 256 in 256..%d", kTileRowsMax);
        g_tile_rows = value;
    } else if (!strcmp(key, "halo_split")) {
        g_halo_split = value != 0;  // read by spmv_hip_comm_halo_setup
    } else if (!strcmp(key, "halo_overlap")) {
        g_halo_overlap = value != 0;
    } else if (!strcmp(key, "tile_pack")) {
        g_tile_pack = value != 0;
    } else if (!strcmp(key, "tile_long")) {
        if (value < 0 || value > 2) return fail("set_tuning: tile_long must be 0, 1 (rows with >= 2^20 entries together) or 2 (always)");
        g_tile_long = value;
    } else if (!strcmp(key, "tile_balance")) {
        g_tile_balance = value != 0;
    } else if (!strcmp(key, "tile_min_pass")) {
        if (value < 0 || value > 2048) return fail("set_tuning: tile_min_pass must be 0 (no remainder) .. 2048");
        g_tile_min_pass = value;
    } else if (!strcmp(key, "tile_places")) {
        if (value < 0 || (value & 7) || value > 4096) return fail("set_tuning: tile_places must be 0 (the chip's) or a multiple of 8 up to 4096");
        g_tile_places = value;
    } else if (!strcmp(key, "tile_items")) {
        if (value < 8 || value > 65536) return fail("set_tuning: tile_items must be 8..65536");
        g_tile_items = value;
    } else if (!strcmp(key, "local_patterns")) {
        g_local_patterns = value < 0 ? -1 : value != 0;  // read at upload (the plan) and at launch (0: the slot stream)
    } else if (!strcmp(key, "tile_mid_items")) {
        if (value < 0 || value > 65536) return fail("set_tuning: tile_mid_items must be 0 (auto: three rounds of the CUs) .. 65536");
        g_tile_mid_items = value;
    } else if (!strcmp(key, "tile_streams")) {
        g_tile_streams = value != 0;
    } else if (!strcmp(key, "tile_fit")) {
        g_tile_fit = value != 0;
    } else if (!strcmp(key, "skew_rows")) {
        g_skew_rows = value != 0;
    } else if (!strcmp(key, "tile_probe")) {
        g_tile_probe = value & 15;
    } else if (!strcmp(key, "probe_depth")) {
        if (value != 4 && value != 16) return fail("set_tuning: probe_depth must be 4 or 16");
        g_probe_depth = value;
    } else if (!strcmp(key, "tile_gather_ahead")) {
        g_tile_gather_ahead = value != 0;  // read at launch
    } else if (!strcmp(key, "tile_expand")) {
        g_tile_expand = value < 0 ? -1 : value != 0;  // read at upload (the plan) and at launch (0: the gather passes)
    } else if (!strcmp(key, "tile_mid_lo")) {
        if (value != 0 && (value < 8 || value > 1024)) return fail("set_tuning: tile_mid_lo must be 0 (auto) or 8..1024");
        g_tile_mid_lo = value;  // takes effect at the next upload
    } else if (!strcmp(key, "tile_mid")) {
        g_tile_mid = value != 0;  // takes effect at the next upload
    } else if (!strcmp(key, "place_tries")) {
        if (value < 0 || value > 16) return fail("set_tuning: place_tries must be 0..16");
        g_place_tries = value;  // takes effect at the next upload
    } else if (!strcmp(key, "tile_plan_on_device")) {
        g_tile_plan_on_device = value != 0;  // takes effect at the next upload
    } else if (!strcmp(key, "tile_density")) {
        if (value < 0 || value > 4096) return fail("set_tuning: tile_density must be 0 (never stage) .. 4096");
        g_tile_density = value;
    } else if (!strcmp(key, "gather_mode")) {
        if (value != 0 && value != 1) return fail("set_tuning: gather_mode must be 0 (broadcasts) or 1 (padded all-gather)");
        g_gather_mode = value;
    } else if (!strcmp(key, "local_nt")) {
        if (value < -1 || value > 1) return fail("set_tuning: local_nt must be -1 (auto), 0 or 1");
        g_local_nt = value;
    } else if (!strcmp(key, "local_cap")) {
        if (value != 0 && value != 1024 && value != 2048 && value != 3072)
            return fail("set_tuning: local_cap must be 0, 1024, 2048 or 3072");
        g_local_cap = value;  // takes effect at the next upload
    } else if (!strcmp(key, "plan_on_device")) {
        g_plan_on_device = value != 0;  // takes effect at the next upload
    } else if (!strcmp(key, "stream_local")) {
        g_stream_local = value != 0;  // takes effect at the next upload
    } else if (!strcmp(key, "pipe_wgs_per_cu")) {
        if (value < 1 || value > 8) return fail("set_tuning: pipe_wgs_per_cu must be 1..8");
        g_pipe_wgs_per_cu = value;
    } else {
        return fail("set_tuning: unknown key '%s'", key);
    }
    return 0;
}

extern "C" int spmv_hip_sync(void) {
    if (need_device()) return -1;
    HIP_TRY(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" void *spmv_hip_stream(void) { return (void *)g_stream; }

extern "C" const char *spmv_hip_last_error(void) { return g_error; }

extern "C" int spmv_hip_device_name(char *buf, size_t len, int *compute_units, long long *hbm_bytes) {
    if (need_device()) return -1;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g_device));
    if (buf && len) snprintf(buf, len, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (long long)prop.totalGlobalMem;
    return 0;
}

extern "C" int spmv_hip_flush_cache(size_t bytes) {
    if (need_device()) return -1;
    if (bytes < 16) bytes = 16;
    if (bytes > g_flush_bytes) {
        if (g_flush_buf) HIP_TRY(hipFree(g_flush_buf));
        g_flush_buf = nullptr;
        g_flush_bytes = 0;
        HIP_TRY(hipMalloc(&g_flush_buf, bytes));
        HIP_TRY(hipMemset(g_flush_buf, 0, bytes));
        g_flush_bytes = bytes;
    }
    hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(kBlock), 0, g_stream, (uint4 *)g_flush_buf,
                       bytes / 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int spmv_hip_device_state(char *buf, size_t len) {
    if (need_device()) return -1;
    if (!buf || !len) return fail("device_state: NULL buffer");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g_device));
    char pci[32] = "";
    (void)hipDeviceGetPCIBusId(pci, (int)sizeof pci, g_device);
    int sclk = 0, mclk = 0, bus = 0, l2 = 0;
    (void)hipDeviceGetAttribute(&sclk, hipDeviceAttributeClockRate, g_device);
    (void)hipDeviceGetAttribute(&mclk, hipDeviceAttributeMemoryClockRate, g_device);
    (void)hipDeviceGetAttribute(&bus, hipDeviceAttributeMemoryBusWidth, g_device);
    (void)hipDeviceGetAttribute(&l2, hipDeviceAttributeL2CacheSize, g_device);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    snprintf(buf, len, "pci=%s;arch=%s;cus=%d;xcds=%d;sclk_khz=%d;mclk_khz=%d;mem_bus_bits=%d;l2_bytes=%d;"
                       "hbm_bytes=%zu;hbm_free_bytes=%zu",
             pci, prop.gcnArchName, prop.multiProcessorCount, prop.multiProcessorCount / 32, sclk, mclk, bus, l2,
             total_b, free_b);
    return 0;
}

static int stream_probe_run(const void *buf, size_t bytes, int warmup, int iters, float *ms_mean, float *ms_min) {
    unsigned *sink = nullptr;
    const int grid = g_num_cus * 8;
    HIP_TRY(hipMalloc((void **)&sink, (size_t)grid * sizeof(unsigned)));
    std::vector<float> ms((size_t)iters, 0.f);
    const int rc = time_loop(warmup, iters, ms.data(),
                             [&]() {
                                 if (g_probe_depth >= 16)
                                     hipLaunchKernelGGL(stream_probe_kernel<16>, dim3(grid), dim3(kBlock), 0, g_stream,
                                                        (const uint4 *)buf, bytes / 16, sink);
                                 else
                                     hipLaunchKernelGGL(stream_probe_kernel<4>, dim3(grid), dim3(kBlock), 0, g_stream,
                                                        (const uint4 *)buf, bytes / 16, sink);
                                 hipError_t e = hipGetLastError();
                                 return e == hipSuccess ? 0 : fail("stream_probe launch: %s", hipGetErrorString(e));
                             },
                             []() { return 0; });
    (void)hipFree(sink);
    if (rc) return rc;
    double sum = 0;
    float mn = ms[0];
    for (float v : ms) {
        sum += v;
        mn = std::min(mn, v);
    }
    if (ms_mean) *ms_mean = (float)(sum / iters);
    if (ms_min) *ms_min = mn;
    return 0;
}

extern "C" int spmv_hip_stream_probe(size_t bytes, int warmup, int iters, float *ms_mean, float *ms_min) {
    if (need_device()) return -1;
    if (iters <= 0 || iters > 1000 || warmup < 0) return fail("stream_probe: iters must be 1..1000");
    if (bytes < (1u << 20)) bytes = 1u << 20;
    if (bytes > g_flush_bytes) {
        if (g_flush_buf) HIP_TRY(hipFree(g_flush_buf));
        g_flush_buf = nullptr;
        g_flush_bytes = 0;
        HIP_TRY(hipMalloc(&g_flush_buf, bytes));
        HIP_TRY(hipMemset(g_flush_buf, 0, bytes));
        g_flush_bytes = bytes;
    }
    return stream_probe_run(g_flush_buf, bytes, warmup, iters, ms_mean, ms_min);
}

// the same probe over memory the caller names (16-byte aligned): what THIS allocation gives a pure stream
extern "C" int spmv_hip_stream_probe_at(const void *dptr, size_t bytes, int warmup, int iters, float *ms_mean, float *ms_min) {
    if (need_device()) return -1;
    if (!dptr || ((uintptr_t)dptr & 15) || bytes < 16 || iters <= 0 || iters > 1000 || warmup < 0)
        return fail("stream_probe_at: bad arguments");
    return stream_probe_run(dptr, bytes, warmup, iters, ms_mean, ms_min);
}

extern "C" int spmv_hip_gather_probe(int value_bytes, size_t table_bytes, int waves_per_cu, double *values_per_s) {
    if (need_device()) return -1;
    if ((value_bytes != 4 && value_bytes != 8) || !values_per_s || waves_per_cu < 4 || waves_per_cu > 32 || (waves_per_cu & 3))
        return fail("gather_probe: value_bytes 4 | 8, waves_per_cu a multiple of 4 in 4..32");
    size_t elems = 1;
    while (elems * 2 * (size_t)value_bytes <= table_bytes) elems *= 2;  // a power of two of elements
    if (elems < 4096 || elems * (size_t)value_bytes > ((size_t)1 << 31)) return fail("gather_probe: table of 32 KiB .. 2 GiB");
    void *table = nullptr, *out = nullptr;
    unsigned *idx = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    do {
        // 64 different random lines per wave-instruction (a fixed xorshift pattern)
        std::vector<unsigned> pattern(8 * 64);
        unsigned state = 2463534242u;
        for (auto &v : pattern) {
            state ^= state << 13;
            state ^= state >> 17;
            state ^= state << 5;
            v = state;
        }
        hipError_t e = hipMalloc(&table, elems * (size_t)value_bytes);
        if (e == hipSuccess) e = hipMalloc((void **)&idx, pattern.size() * sizeof(unsigned));
        if (e == hipSuccess) e = hipMalloc(&out, 64);
        if (e == hipSuccess) e = hipMemsetAsync(table, 0, elems * (size_t)value_bytes, g_stream);
        if (e == hipSuccess) e = hipMemcpy(idx, pattern.data(), pattern.size() * sizeof(unsigned), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e != hipSuccess) { rc = fail("gather_probe: setup failed: %s", hipGetErrorString(e)); break; }
        const int iters = 1000, blocks = g_num_cus * waves_per_cu / 4;
        const unsigned mask = (unsigned)elems - 1;
        auto launch = [&](int n) {
            if (value_bytes == 8)
                hipLaunchKernelGGL((gather_probe_kernel<double>), dim3(blocks), dim3(kBlock), 0, g_stream, (const double *)table, idx, (double *)out, n, mask);
            else
                hipLaunchKernelGGL((gather_probe_kernel<float>), dim3(blocks), dim3(kBlock), 0, g_stream, (const float *)table, idx, (float *)out, n, mask);
        };
        launch(50);
        e = hipEventRecord(e0, g_stream);
        launch(iters);
        if (e == hipSuccess) e = hipEventRecord(e1, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess || ms <= 0) { rc = fail("gather_probe: run failed: %s", hipGetErrorString(e)); break; }
        *values_per_s = (double)iters * 8.0 * 64.0 * (double)waves_per_cu * (double)g_num_cus / (ms * 1e-3);
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(table);
    (void)hipFree(idx);
    (void)hipFree(out);
    return rc;
}

extern "C" int spmv_hip_malloc(void **dptr, size_t bytes) {
    if (need_device()) return -1;
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
    return 0;
}
extern "C" int spmv_hip_free(void *dptr) {
    if (dptr) HIP_TRY(hipFree(dptr));
    return 0;
}
extern "C" int spmv_hip_memcpy_h2d(void *dptr, const void *hptr, size_t bytes) {
    if (need_device()) return -1;
    HIP_TRY(hipMemcpy(dptr, hptr, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int spmv_hip_memcpy_d2h(void *hptr, const void *dptr, size_t bytes) {
    if (need_device()) return -1;
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(hptr, dptr, bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int spmv_hip_memset(void *dptr, int byte, size_t bytes) {
    if (need_device()) return -1;
    HIP_TRY(hipMemsetAsync(dptr, byte, bytes, g_stream));
    return 0;
}

