// spmv_hip.hip -- the C-ABI of include/spmv_hip.h: device memory management,
// upload-time preprocessing, kernel launchers, timing and the RCCL exchange.
//
// This translation unit stands where the reference's CUDA driver touches the
// device (/root/reference/main_cuda.cu:135-145, :149-166, :212-238, :285-317,
// :369-455, :545-568, :613-637, :682-744).  Nothing here computes on the host:
// if HIP is unusable every entry point returns -1 with a message.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "csr_kernels.hpp"
#include "hll_kernels.hpp"
#include "spmv_hip.h"

using namespace spmv;

// ------------------------------------------------------------------ state
namespace {

thread_local char g_error[512] = "";
int g_device = -1;
hipStream_t g_stream = nullptr;
void *g_flush_buf = nullptr;
size_t g_flush_bytes = 0;
ncclComm_t g_comm = nullptr;
int g_comm_rank = 0, g_comm_size = 1;

// kernel tuning knobs (spmv_hip_set_tuning); defaults are the measured best
int g_stream_cap = 0;     // nnz staged per stream workgroup (fixed at upload); 0 = by matrix size
int g_stream_block = 256; // threads per csr_stream workgroup
int g_stream_nt = 1;      // non-temporal loads for col/val in the gather stream kernels
int g_local_nt = -1;      // same for the x-window kernel: -1 = auto (off while the matrix fits the Infinity Cache)
int g_stream_xcd = 0;     // blocks per XCD run (xcd_chunked); 0 = dispatch order, -1 = one contiguous eighth per XCD
int g_gather_mode = 0;     // all-gatherv: 0 = one ncclBroadcast per owner in a group, 1 = padded ncclAllGather + scatter
int g_local_cap = 0;       // stage of the x-window plan: 0 = auto, 1024 or 2048
int g_stream_local = 1;    // build the x-window plan at upload (csr_stream_local) when it pays
int g_stream_kind = -1;    // -1 = auto (csr_stream_local when the matrix has a plan, else csr_stream), 5 = local,
                           // 0 = csr_stream (products), 1 = csr_stream_rows (row walk), 2 = csr_stream_pipe
int g_pipe_wgs_per_cu = 5; // resident workgroups per CU the persistent grid is sized for
int g_num_cus = 256;
int g_probe_mask = 1023;   // csr_probe: table size - 1 (entries) of the folded gather

int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return -1;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t err__ = (expr);                                                        \
        if (err__ != hipSuccess)                                                          \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), __FILE__, \
                        __LINE__);                                                        \
    } while (0)

#define NCCL_TRY(expr)                                                                     \
    do {                                                                                   \
        ncclResult_t err__ = (expr);                                                       \
        if (err__ != ncclSuccess)                                                          \
            return fail("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(err__), __FILE__, \
                        __LINE__);                                                         \
    } while (0)

int need_device() {
    if (g_device < 0) return fail("spmv_hip_init() has not been called (or failed): no HIP device");
    return 0;
}

constexpr int kPad = 8192 + 64;  // zero entries behind col/val: the stream / LDS kernels stage
                                   // whole units without bounds tests (>= kStreamCapMax, kHllCap)

template <typename T>
int upload_array(T **dptr, const T *host, size_t count, size_t pad) {
    HIP_TRY(hipMalloc((void **)dptr, (count + pad) * sizeof(T)));
    if (count) HIP_TRY(hipMemcpy(*dptr, host, count * sizeof(T), hipMemcpyHostToDevice));
    if (pad) HIP_TRY(hipMemset(*dptr + count, 0, pad * sizeof(T)));
    return 0;
}

int pow2_floor(int v) {
    int p = 1;
    while (p * 2 <= v) p *= 2;
    return p;
}

}  // namespace

// ---------------------------------------------------------------- handles
struct spmv_csr_dev {
    int value_bytes = 8;
    int M_local = 0, M_total = 0, N = 0, row0 = 0;
    long long nz = 0;
    int *row_ptr = nullptr;  // [M_local + 1], rebased to 0
    int *col = nullptr;
    void *val = nullptr;
    void *x = nullptr;  // [N]
    void *y = nullptr;  // [M_total]
    // stream kernel
    int4 *desc = nullptr;
    int num_blocks = 0;
    int4 *long_rows = nullptr;
    int num_long = 0;
    int4 *pieces = nullptr;
    void *partial = nullptr;
    int num_partial = 0;
    int stream_cap = 2048;
    bool ring_ok = false;  // blocks respect the ring kernel's row limit
    // stream kernel with the x window in LDS (csr_stream_local): own blocks, 16-bit local columns
    int4 *ldesc4 = nullptr;           // [local_blocks] like desc
    int2 *ldesc = nullptr;            // [local_blocks] {first line in `lines`, line count}
    int *lines = nullptr;             // x line ids, block after block, ascending inside a block
    unsigned short *lcol = nullptr;   // [nz + pad] slot of each entry in its block's staged lines
    int local_blocks = 0;             // 0: no plan (not profitable / not possible)
    int local_stage_lines = 0;        // LDS stage: most lines any block lists, in steps of 32
    int local_cap = 2048;
    long long local_lines = 0;
    // heuristics
    int lanes_per_row = 16;
    int auto_variant = SPMV_CSR_STREAM;
    int max_row = 0;
    size_t device_bytes = 0;
};

struct spmv_hll_dev {
    int M = 0, N = 0, hacks = 0;  // rows / hacks HELD by this handle
    int M_total = 0, row0 = 0;    // rows of the whole matrix (length of y), first global row (multiple of 32)
    long long nz_hint = 0;
    long long slots = 0;
    long long *hack_off = nullptr;  // [hacks + 1]
    int *maxnz = nullptr;           // [hacks]
    int *JA = nullptr;
    double *AS = nullptr;
    int4 *hdesc = nullptr;  // [num_blocks] {first row, rows, first slot lo, first slot hi}
    int num_blocks = 0;
    int stage_slots = kHllCap;  // LDS stage of hll_lds: the largest workgroup, <= kHllCap
    // hll_lds_local (x window in LDS): own windows, 16-bit local JA
    int4 *ldesc4 = nullptr;
    int4 *ldesc = nullptr;  // {first line, lines, slots of the window from its even base, 0}
    int *lines = nullptr;
    unsigned short *lja = nullptr;
    int local_blocks = 0, local_stage_lines = 0;
    long long local_lines = 0;
    double *x = nullptr;
    double *y = nullptr;
    int lanes_per_row = 8;
    int auto_variant = SPMV_HLL_LDS;
    size_t device_bytes = 0;
};

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
    }
    if (!g_stream) HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
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
        if (value != 0 && value != 1024 && value != 2048 && value != 4096 && value != 8192)
            return fail("set_tuning: stream_cap must be 0 (auto), 1024, 2048, 4096 or 8192");
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
        if ((value < -1 || value > 5) && (value < 10 || value > 17))
            return fail("set_tuning: stream_kind must be -1..5 (or 10..17 for the ablation probes)");
        g_stream_kind = value;
    } else if (!strcmp(key, "gather_mode")) {
        if (value != 0 && value != 1) return fail("set_tuning: gather_mode must be 0 (broadcasts) or 1 (padded all-gather)");
        g_gather_mode = value;
    } else if (!strcmp(key, "local_nt")) {
        if (value < -1 || value > 1) return fail("set_tuning: local_nt must be -1 (auto), 0 or 1");
        g_local_nt = value;
    } else if (!strcmp(key, "local_cap")) {
        if (value != 0 && value != 1024 && value != 2048) return fail("set_tuning: local_cap must be 0, 1024 or 2048");
        g_local_cap = value;  // takes effect at the next upload
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

// ----------------------------------------------------------- CSR: upload
namespace {

// Cut rows [0, M) (row_ptr rebased to 0) into workgroup-sized blocks for the
// stream kernels: desc = {first row, first entry, rows, end entry}, each block's entries
// (counted from the even entry at or below its first) fit `cap`.  A row that
// cannot be staged is cut into pieces {row, first entry, end entry, slot} whose
// partial sums csr_long_finish adds up per long row {row, first slot, pieces, 0}.
// `split` (optional, one byte per row) marks rows that go to the split-row kernels whatever their
// length: rows the x-window plan cannot take (more x lines than a block may list).
void csr_build_blocks(int M, const int *rp, int cap, int rows_cap, std::vector<int4> &desc,
                      std::vector<int4> &pieces, std::vector<int4> &long_rows,
                      const std::vector<unsigned char> *split = nullptr) {
    desc.clear();
    pieces.clear();
    long_rows.clear();
    auto is_long = [&](int row) { return rp[row + 1] - rp[row] > cap - 3 || (split && (*split)[row]); };
    int r = 0;
    while (r < M) {
        const int n0 = rp[r];
        const int base = n0 & kBaseMask;
        if (is_long(r)) {
            const int first_slot = (int)pieces.size();
            for (int p = n0; p < rp[r + 1]; p += kLongPiece)
                pieces.push_back(int4{r, p, std::min(p + kLongPiece, rp[r + 1]), (int)pieces.size()});
            long_rows.push_back(int4{r, first_slot, (int)pieces.size() - first_slot, 0});
            ++r;
            continue;
        }
        int r1 = r;
        while (r1 < M && r1 - r < rows_cap && rp[r1 + 1] - base <= cap && !is_long(r1)) ++r1;
        desc.push_back(int4{r, n0, r1 - r, rp[r1]});
        r = r1;
    }
}

// Blocks for csr_stream_local: csr_build_blocks' cut with one more limit, the number of
// distinct x lines (1 << line_shift elements each) a block touches.  Fills, per block, the
// ascending list of those lines and, per entry, its 16-bit slot (rank of its line in the
// list * elements per line + column % elements per line).  Returns false when some row alone
// needs more than lines_max lines, or when the line limit (rather than cap) decides so many
// cuts that the blocks would run mostly empty: the caller then keeps the gather kernel.
struct LocalPlan {
    std::vector<unsigned char> split;  // CSR: rows handed to the split-row kernels (too many x lines)
    std::vector<int4> desc;
    std::vector<int4> hll_ldesc;  // HLL: {first line, lines, slots from the even base, 0}
    std::vector<int2> ldesc;
    std::vector<int> lines;
    std::vector<unsigned short> lcol;
    int stage_lines = 0;
};

bool csr_build_local(int M, int N, const int *rp, const int *col, long long nz, int cap, int rows_cap,
                     int line_shift, int lines_max, const std::vector<int4> &baseline, LocalPlan &plan) {
    const int total_lines = (int)(((long long)N + (1 << line_shift) - 1) >> line_shift);
    const int line_mask = (1 << line_shift) - 1;
    std::vector<int> stamp((size_t)total_lines + 1, -1), rank((size_t)total_lines + 1, 0), cur;
    plan.lcol.assign((size_t)nz + kPad, 0);
    plan.split.assign((size_t)M, 0);
    plan.desc.clear();
    plan.ldesc.clear();
    plan.lines.clear();
    int widest = 0;
    long long split_entries = 0;
    int r = 0;
    while (r < M) {
        const int n0 = rp[r];
        const int base = n0 & kBaseMask;
        if (rp[r + 1] - n0 > cap - 3) {  // long row: csr_long_pieces, as in csr_build_blocks
            ++r;
            continue;
        }
        const int blk = (int)plan.desc.size();
        cur.clear();
        int r1 = r;
        while (r1 < M && r1 - r < rows_cap && rp[r1 + 1] - base <= cap && rp[r1 + 1] - rp[r1] <= cap - 3) {
            const size_t before = cur.size();
            for (int e = rp[r1]; e < rp[r1 + 1]; ++e) {
                const int l = col[e] >> line_shift;
                if (stamp[l] != blk) {
                    stamp[l] = blk;
                    cur.push_back(l);
                }
            }
            if ((int)cur.size() > lines_max) {  // this row does not fit any more: take it back
                for (size_t k = before; k < cur.size(); ++k) stamp[cur[k]] = -1;
                cur.resize(before);
                break;
            }
            ++r1;
        }
        if (r1 == r) {
            // one row alone touches more lines than a block may list: it goes to the split-row
            // (gather) kernels like a long row; a matrix made of such rows keeps the gather kernel
            plan.split[r] = 1;
            split_entries += rp[r + 1] - n0;
            if (split_entries * 20 > nz) return false;
            ++r;
            continue;
        }
        if (cur.empty()) cur.push_back(0);  // only empty rows: the kernel still stages one line
        std::sort(cur.begin(), cur.end());
        for (size_t k = 0; k < cur.size(); ++k) rank[cur[k]] = (int)k;
        for (int e = rp[r]; e < rp[r1]; ++e)
            plan.lcol[e] = (unsigned short)((rank[col[e] >> line_shift] << line_shift) | (col[e] & line_mask));
        plan.desc.push_back(int4{r, n0, r1 - r, rp[r1]});
        plan.ldesc.push_back(int2{(int)plan.lines.size(), (int)cur.size()});
        plan.lines.insert(plan.lines.end(), cur.begin(), cur.end());
        widest = std::max(widest, (int)cur.size());
        r = r1;
        // the line limit is cutting blocks well short of what cap alone allows: give up early
        if ((plan.desc.size() & 1023) == 0) {
            const size_t plain = std::lower_bound(baseline.begin(), baseline.end(), r,
                                                  [](const int4 &d, int row) { return d.x < row; }) -
                                 baseline.begin();
            if (plan.desc.size() > plain + plain / 5 + 16) return false;
        }
    }
    if (plan.desc.size() > baseline.size() + baseline.size() / 5 + 1) return false;
    // the kernel stages whole passes of kLocalLineQuantum lines and reads the list unconditionally
    plan.stage_lines = std::max(kLocalLineQuantum,
                                (widest + kLocalLineQuantum - 1) / kLocalLineQuantum * kLocalLineQuantum);
    plan.lines.insert(plan.lines.end(), (size_t)kLocalLinesMax, 0);
    return true;
}

template <typename T>
int csr_upload_impl(int M, int N, const int *row_ptr, const int *col_idx, const T *values, int row0,
                    int row1, spmv_csr_dev **out) {
    if (need_device()) return -1;
    if (!out) return fail("csr_upload: out is NULL");
    *out = nullptr;
    if (M < 0 || N < 0 || !row_ptr) return fail("csr_upload: bad arguments");
    if (row0 < 0 || row1 < row0 || row1 > M) return fail("csr_upload: bad row range [%d, %d) of %d", row0, row1, M);
    const int Ml = row1 - row0;
    const int e0 = row_ptr[row0], e1 = row_ptr[row1];
    const long long nz = (long long)e1 - e0;
    if (nz < 0) return fail("csr_upload: row_ptr is not monotone");
    if (nz > 0 && (!col_idx || !values)) return fail("csr_upload: col_idx / values are NULL");
    if ((unsigned long long)N * sizeof(T) >= (1ull << 32))
        return fail("csr_upload: N = %d exceeds the 32-bit gather offset range of the kernels", N);
    // a column index outside [0, N) would make the kernels gather out of bounds
    for (int e = e0; e < e1; ++e)
        if ((unsigned)col_idx[e] >= (unsigned)N)
            return fail("csr_upload: column index %d at entry %d is outside [0, %d)", col_idx[e], e, N);

    spmv_csr_dev *m = new (std::nothrow) spmv_csr_dev();
    if (!m) return fail("csr_upload: out of host memory");
    m->value_bytes = (int)sizeof(T);
    m->M_local = Ml;
    m->M_total = M;
    m->N = N;
    m->row0 = row0;
    m->nz = nz;

    std::vector<int> rp((size_t)Ml + 1);
    int max_row = 0;
    for (int r = 0; r <= Ml; ++r) rp[r] = row_ptr[row0 + r] - e0;
    for (int r = 0; r < Ml; ++r) {
        if (rp[r + 1] < rp[r]) {
            delete m;
            return fail("csr_upload: row_ptr decreases at row %d", row0 + r);
        }
        max_row = std::max(max_row, rp[r + 1] - rp[r]);
    }
    m->max_row = max_row;

    std::vector<int4> desc, pieces, long_rows;
    // The x-window kernel first (csr_stream_local): own blocks at a 2048-entry stage.  One stage
    // size per handle, so that its blocks, the gather kernel's and the split long rows agree on
    // which rows are long: a matrix that gets a plan runs everything at 2048.  An explicit
    // stream_cap other than 2048 asks for the gather kernel's configuration and skips the plan.
    LocalPlan local;
    bool have_local = false;
    constexpr int line_shift = sizeof(T) == 8 ? 4 : 5;  // 128-byte lines
    // (1024-entry blocks were tried for small matrices: cant-like 13.2 us against 11.7 us at 2048)
    const int lcap = g_local_cap ? g_local_cap : 2048;
    if (g_stream_local && nz > 0 && (g_stream_cap == 0 || g_stream_cap == lcap)) {
        csr_build_blocks(Ml, rp.data(), lcap, kStreamRowsCap, desc, pieces, long_rows);
        have_local = csr_build_local(Ml, N, rp.data(), col_idx + e0, nz, lcap, kStreamRowsCap, line_shift,
                                     kLocalLinesMax, desc, local);
    }
    // else: larger stages amortise per-workgroup latency on big matrices; small ones need
    // enough workgroups to fill 256 CUs (measured: cant-like 2048, nlpkkt-like 4096)
    m->stream_cap = have_local ? lcap : (g_stream_cap ? g_stream_cap : (nz >= (16LL << 20) ? 4096 : 2048));
    m->local_cap = lcap;
    // the ring kernel stages at most kRingRows - 1 rows per block; only worth it when such
    // blocks are still (nearly) full, i.e. rows are not tiny
    m->ring_ok = m->stream_cap == kRingCap && Ml > 0 && (double)nz / Ml >= 1.25 * kRingCap / (kRingRows - 1);
    csr_build_blocks(Ml, rp.data(), m->stream_cap, m->ring_ok ? kRingRows - 1 : kStreamRowsCap, desc,
                     pieces, long_rows, have_local ? &local.split : nullptr);
    m->num_blocks = (int)desc.size();
    m->num_long = (int)long_rows.size();
    m->num_partial = (int)pieces.size();
    const int num_partial = m->num_partial;

    int rc = 0;
    rc |= upload_array(&m->row_ptr, rp.data(), rp.size(), (size_t)kRingRows + 64);
    if (!rc && have_local) {
        rc |= upload_array(&m->ldesc4, local.desc.data(), local.desc.size(), 1);
        if (!rc) rc |= upload_array(&m->ldesc, local.ldesc.data(), local.ldesc.size(), 1);
        if (!rc) rc |= upload_array(&m->lines, local.lines.data(), local.lines.size(), 0);
        if (!rc) rc |= upload_array(&m->lcol, local.lcol.data(), local.lcol.size(), 0);
        if (!rc) {
            m->local_blocks = (int)local.desc.size();
            m->local_stage_lines = local.stage_lines;
            m->local_lines = (long long)local.lines.size() - kLocalLinesMax;
        }
    }
    if (!rc) rc |= upload_array(&m->col, col_idx ? col_idx + e0 : nullptr, (size_t)nz, kPad);
    if (!rc) rc |= upload_array((T **)&m->val, values ? values + e0 : nullptr, (size_t)nz, kPad);
    if (!rc) rc |= upload_array(&m->desc, desc.data(), desc.size(), 1);
    if (!rc && m->num_long) rc |= upload_array(&m->long_rows, long_rows.data(), long_rows.size(), 0);
    if (!rc && num_partial) rc |= upload_array(&m->pieces, pieces.data(), pieces.size(), 0);
    if (!rc && num_partial) {
        hipError_t e = hipMalloc(&m->partial, (size_t)num_partial * sizeof(T));
        if (e != hipSuccess) rc = fail("hipMalloc(partial) failed: %s", hipGetErrorString(e));
    }
    if (!rc) {
        // x is read in whole 128-byte lines by csr_stream_local: room for the tail of the last one
        const size_t x_bytes = std::max<size_t>((size_t)N, 1) * sizeof(T) + kLineBytes;
        hipError_t e = hipMalloc(&m->x, x_bytes);
        if (e == hipSuccess) e = hipMalloc(&m->y, std::max<size_t>((size_t)M, 1) * sizeof(T));
        if (e == hipSuccess) e = hipMemset(m->x, 0, x_bytes);
        if (e == hipSuccess) e = hipMemset(m->y, 0, std::max<size_t>((size_t)M, 1) * sizeof(T));
        if (e != hipSuccess) rc = fail("hipMalloc(x/y) failed: %s", hipGetErrorString(e));
    }
    if (rc) {
        spmv_hip_csr_free(m);
        return -1;
    }
    m->device_bytes = rp.size() * 4 + ((size_t)nz + kPad) * (4 + sizeof(T)) + desc.size() * 16 +
                      long_rows.size() * 16 + pieces.size() * 16 + (size_t)num_partial * sizeof(T) +
                      ((size_t)N + (size_t)M) * sizeof(T);
    if (have_local)
        m->device_bytes += local.desc.size() * 24 + local.lines.size() * 4 + local.lcol.size() * 2;

    // lanes per row for the SUBWAVE kernel: about half the mean row length,
    // rounded to a power of two, so that a typical row takes 1-2 passes
    const double mean = Ml ? (double)nz / Ml : 0.0;
    int lanes = pow2_floor(std::max(2, (int)(mean / 2.0 + 0.5)));
    m->lanes_per_row = std::min(32, std::max(2, lanes));
    m->auto_variant = SPMV_CSR_STREAM;
    *out = m;
    return 0;
}

}  // namespace

extern "C" int spmv_hip_csr_upload(int M, int N, const int *row_ptr, const int *col_idx,
                                   const double *values, int row0, int row1, spmv_csr_dev **out) {
    return csr_upload_impl<double>(M, N, row_ptr, col_idx, values, row0, row1, out);
}

extern "C" int spmv_hip_csr_upload_f32(int M, int N, const int *row_ptr, const int *col_idx,
                                       const float *values, int row0, int row1, spmv_csr_dev **out) {
    return csr_upload_impl<float>(M, N, row_ptr, col_idx, values, row0, row1, out);
}

extern "C" int spmv_hip_csr_upload_matrix(const CSRMatrix *csr, spmv_csr_dev **out) {
    if (!csr) return fail("csr_upload_matrix: csr is NULL");
    return spmv_hip_csr_upload(csr->M, csr->N, csr->row_ptr, csr->col_idx, csr->values, 0, csr->M, out);
}

extern "C" void spmv_hip_csr_free(spmv_csr_dev *m) {
    if (!m) return;
    (void)hipFree(m->row_ptr);
    (void)hipFree(m->col);
    (void)hipFree(m->val);
    (void)hipFree(m->desc);
    (void)hipFree(m->ldesc4);
    (void)hipFree(m->ldesc);
    (void)hipFree(m->lines);
    (void)hipFree(m->lcol);
    (void)hipFree(m->long_rows);
    (void)hipFree(m->pieces);
    (void)hipFree(m->partial);
    (void)hipFree(m->x);
    (void)hipFree(m->y);
    delete m;
}

extern "C" int spmv_hip_csr_info(const spmv_csr_dev *m, spmv_dev_info *out) {
    if (!m || !out) return fail("csr_info: NULL argument");
    memset(out, 0, sizeof *out);
    out->M_local = m->M_local;
    out->M_total = m->M_total;
    out->N = m->N;
    out->row0 = m->row0;
    out->nz = m->nz;
    out->value_bytes = m->value_bytes;
    out->auto_variant = m->auto_variant;
    out->lanes_per_row = m->lanes_per_row;
    out->stream_blocks = m->num_blocks;
    out->long_rows = m->num_long;
    const long long vb = m->value_bytes;
    // nnz (val + 4) + 4 (M + 1) + val M [y] + val N [x]
    out->algo_bytes = m->nz * (vb + 4) + 4LL * (m->M_local + 1) + vb * m->M_local + vb * m->N;
    out->device_bytes = (long long)m->device_bytes;
    out->local_blocks = m->local_blocks;
    out->local_stage_lines = m->local_stage_lines;
    out->local_lines = m->local_lines;
    if (m->local_blocks > 0)
        out->stream_bytes = m->nz * (vb + 2) + 4 * m->local_lines + 24LL * m->local_blocks +
                            4LL * (m->M_local + 1) + vb * m->M_local + vb * m->N;
    return 0;
}

extern "C" int spmv_hip_csr_set_x(spmv_csr_dev *m, const void *x_host) {
    if (need_device()) return -1;
    if (!m || !x_host) return fail("csr_set_x: NULL argument");
    HIP_TRY(hipMemcpyAsync(m->x, x_host, (size_t)m->N * m->value_bytes, hipMemcpyHostToDevice, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int spmv_hip_csr_get_y(spmv_csr_dev *m, void *y_host) {
    if (need_device()) return -1;
    if (!m || !y_host) return fail("csr_get_y: NULL argument");
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(y_host, m->y, (size_t)m->M_total * m->value_bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void *spmv_hip_csr_x_ptr(spmv_csr_dev *m) { return m ? m->x : nullptr; }
extern "C" void *spmv_hip_csr_y_ptr(spmv_csr_dev *m) { return m ? m->y : nullptr; }

// ----------------------------------------------------------- CSR: launch
namespace {

template <typename T, int L>
void launch_vector(const spmv_csr_dev *m, const T *x, T *y, hipStream_t s) {
    constexpr int rows = kBlock / L;
    const int grid = (m->M_local + rows - 1) / rows;
    hipLaunchKernelGGL((csr_vector<T, L, 1, false>), dim3(grid), dim3(kBlock), 0, s, m->M_local,
                       m->row_ptr, m->col, (const T *)m->val, x, y);
}

template <typename T>
int csr_launch(const spmv_csr_dev *m, int variant, const T *x, T *y_full, hipStream_t s) {
    if (m->M_local == 0) return 0;
    T *y = y_full + m->row0;
    if (variant == SPMV_CSR_AUTO) variant = m->auto_variant;
    switch (variant) {
        case SPMV_CSR_THREAD_ROW: {
            const int grid = (m->M_local + kBlock - 1) / kBlock;
            hipLaunchKernelGGL((csr_thread_row<T>), dim3(grid), dim3(kBlock), 0, s, m->M_local,
                               m->row_ptr, m->col, (const T *)m->val, x, y);
            break;
        }
        case SPMV_CSR_WAVE_ROW: {
            constexpr int rows = kBlock / 64;
            const int grid = (m->M_local + rows - 1) / rows;
            hipLaunchKernelGGL((csr_vector<T, 64, 2, true>), dim3(grid), dim3(kBlock), 0, s,
                               m->M_local, m->row_ptr, m->col, (const T *)m->val, x, y);
            break;
        }
        case SPMV_CSR_SUBWAVE:
            switch (m->lanes_per_row) {
                case 2: launch_vector<T, 2>(m, x, y, s); break;
                case 4: launch_vector<T, 4>(m, x, y, s); break;
                case 8: launch_vector<T, 8>(m, x, y, s); break;
                case 16: launch_vector<T, 16>(m, x, y, s); break;
                default: launch_vector<T, 32>(m, x, y, s); break;
            }
            break;
        case SPMV_CSR_STREAM: {
            if (m->num_blocks > 0) {
                const int per_xcd = (m->num_blocks + 7) / 8;
#define SPMV_ARGS m->desc, m->row_ptr, m->col, (const T *)m->val, x, y
#define SPMV_LAUNCH_PROD(NT, CAP, BLOCK)                                                          \
    hipLaunchKernelGGL((csr_stream<T, NT, CAP, BLOCK>), dim3(grid_blocks), dim3(BLOCK), 0, s,      \
                       m->num_blocks, chunk, SPMV_ARGS)
#define SPMV_LAUNCH_FLAGS(MACRO, ...)              \
    do {                                           \
        if (g_stream_nt) MACRO(true, __VA_ARGS__); \
        else MACRO(false, __VA_ARGS__);            \
    } while (0)
                // blocks per XCD run; a dummy empty block is harmless for the persistent kernels
                const int chunk = g_stream_xcd < 0 ? per_xcd : g_stream_xcd;
                const int grid_blocks = chunk > 0 ? (m->num_blocks + 8 * chunk - 1) / (8 * chunk) * (8 * chunk)
                                                  : m->num_blocks;
                const int cap = m->stream_cap, blk = g_stream_block;
                // the x-window kernel reads whole aligned lines of x
                const bool local = (g_stream_kind == -1 || g_stream_kind == 5) && m->local_blocks > 0 &&
                                   ((uintptr_t)x & (kLineBytes - 1)) == 0;
                if (local) {
                    // runs of 16 neighbouring blocks per XCD: each L2 keeps its own window of x lines
                    // (measured flat from 8 to 128 on three matrices); stream_xcd overrides
                    const int lchunk = g_stream_xcd < 0 ? (m->local_blocks + 7) / 8 : (g_stream_xcd ? g_stream_xcd : 16);
                    const int lgrid = lchunk > 0 ? (m->local_blocks + 8 * lchunk - 1) / (8 * lchunk) * (8 * lchunk)
                                                 : m->local_blocks;
                    const size_t lds = std::max((size_t)m->local_cap * sizeof(T), (size_t)m->local_stage_lines * kLineBytes);
#define SPMV_LOCAL(NT, CAP)                                                                                   \
    hipLaunchKernelGGL((csr_stream_local<T, NT, CAP>), dim3(lgrid), dim3(kBlock), lds, s, m->local_blocks, lchunk, \
                       m->ldesc4, m->ldesc, m->lines, m->row_ptr, m->lcol, (const T *)m->val, x, y)
                    // streamed-once hint only when the matrix cannot live in the 256 MiB Infinity Cache anyway
                    // (cant-like, 53 MB: 10.9 us without it, 11.7 us with; fem-large: 160 vs 151 us)
                    const bool lnt = g_local_nt < 0 ? m->nz * (long long)(sizeof(T) + 2) > (128LL << 20) : g_local_nt != 0;
                    if (m->local_cap == 1024) { if (lnt) SPMV_LOCAL(true, 1024); else SPMV_LOCAL(false, 1024); }
                    else { if (lnt) SPMV_LOCAL(true, 2048); else SPMV_LOCAL(false, 2048); }
#undef SPMV_LOCAL
                } else if (g_stream_kind == 4 && m->ring_ok) {
                    // loader / consumer ring: one persistent 512-thread workgroup per CU
                    const int wgs = std::max(1, std::min(g_num_cus * g_pipe_wgs_per_cu, m->num_blocks));
                    if (g_pipe_wgs_per_cu >= 2) {
                        if (g_stream_nt) hipLaunchKernelGGL((csr_stream_ring<T, true, 3, 2>), dim3(wgs), dim3(kRingBlock), 0, s, m->num_blocks, SPMV_ARGS);
                        else hipLaunchKernelGGL((csr_stream_ring<T, false, 3, 2>), dim3(wgs), dim3(kRingBlock), 0, s, m->num_blocks, SPMV_ARGS);
                    } else {
                        if (g_stream_nt) hipLaunchKernelGGL((csr_stream_ring<T, true, 4, 3>), dim3(wgs), dim3(kRingBlock), 0, s, m->num_blocks, SPMV_ARGS);
                        else hipLaunchKernelGGL((csr_stream_ring<T, false, 4, 3>), dim3(wgs), dim3(kRingBlock), 0, s, m->num_blocks, SPMV_ARGS);
                    }
                } else if (g_stream_kind >= 10 && g_stream_kind <= 17 && (cap == 2048 || cap == 4096)) {
                    // ablation probes (measurement only; y is not A x)
#define SPMV_PROBE(CAP, MODE) hipLaunchKernelGGL((csr_probe<T, true, CAP, MODE>), dim3(grid_blocks), dim3(kBlock), 0, s, m->num_blocks, chunk, g_probe_mask, SPMV_ARGS)
                    const int mode = g_stream_kind - 10;
                    if (cap == 2048) { if (mode == 0) SPMV_PROBE(2048, 0); else if (mode == 1) SPMV_PROBE(2048, 1); else if (mode == 2) SPMV_PROBE(2048, 2); else if (mode == 3) SPMV_PROBE(2048, 3); else if (mode == 5) SPMV_PROBE(2048, 5); else SPMV_PROBE(2048, 7); }
                    else { if (mode == 0) SPMV_PROBE(4096, 0); else if (mode == 1) SPMV_PROBE(4096, 1); else if (mode == 2) SPMV_PROBE(4096, 2); else if (mode == 3) SPMV_PROBE(4096, 3); else if (mode == 5) SPMV_PROBE(4096, 5); else SPMV_PROBE(4096, 7); }
#undef SPMV_PROBE
                } else if (g_stream_kind == 2 && cap <= 4096) {
                    // persistent grid: what is resident at once (at least two blocks each)
                    int wgs = std::max(8, std::min(g_num_cus * g_pipe_wgs_per_cu, (m->num_blocks + 1) / 2) / 8 * 8);
                    if (cap == 2048) {
                        if (g_stream_nt) hipLaunchKernelGGL((csr_stream_pipe<T, true, 2048>), dim3(wgs), dim3(kBlock), 0, s, m->num_blocks, chunk, SPMV_ARGS);
                        else hipLaunchKernelGGL((csr_stream_pipe<T, false, 2048>), dim3(wgs), dim3(kBlock), 0, s, m->num_blocks, chunk, SPMV_ARGS);
                    } else {
                        if (g_stream_nt) hipLaunchKernelGGL((csr_stream_pipe<T, true, 4096>), dim3(wgs), dim3(kBlock), 0, s, m->num_blocks, chunk, SPMV_ARGS);
                        else hipLaunchKernelGGL((csr_stream_pipe<T, false, 4096>), dim3(wgs), dim3(kBlock), 0, s, m->num_blocks, chunk, SPMV_ARGS);
                    }
                } else if ((g_stream_kind == 1 || g_stream_kind == 3) && cap <= 4096) {
                    // kind 1: one block per workgroup; kind 3: persistent grid-stride
                    const bool persist = g_stream_kind == 3;
                    const int wgs = persist ? std::max(8, std::min(g_num_cus * g_pipe_wgs_per_cu, m->num_blocks) / 8 * 8)
                                            : grid_blocks;
#define SPMV_WALK(NT, CAP, P) hipLaunchKernelGGL((csr_stream_walk<T, NT, CAP, P>), dim3(wgs), dim3(kBlock), 0, s, m->num_blocks, chunk, SPMV_ARGS)
                    if (cap == 2048) {
                        if (persist) { if (g_stream_nt) SPMV_WALK(true, 2048, true); else SPMV_WALK(false, 2048, true); }
                        else { if (g_stream_nt) SPMV_WALK(true, 2048, false); else SPMV_WALK(false, 2048, false); }
                    } else {
                        if (persist) { if (g_stream_nt) SPMV_WALK(true, 4096, true); else SPMV_WALK(false, 4096, true); }
                        else { if (g_stream_nt) SPMV_WALK(true, 4096, false); else SPMV_WALK(false, 4096, false); }
                    }
#undef SPMV_WALK
                } else if (cap == 1024) {
                    SPMV_LAUNCH_FLAGS(SPMV_LAUNCH_PROD, 1024, 256);
                } else if (cap == 2048) {
                    SPMV_LAUNCH_FLAGS(SPMV_LAUNCH_PROD, 2048, 256);
                } else if (cap == 4096 && blk == 512) {
                    SPMV_LAUNCH_FLAGS(SPMV_LAUNCH_PROD, 4096, 512);
                } else if (cap == 4096) {
                    SPMV_LAUNCH_FLAGS(SPMV_LAUNCH_PROD, 4096, 256);
                } else if (blk == 1024) {
                    SPMV_LAUNCH_FLAGS(SPMV_LAUNCH_PROD, 8192, 1024);
                } else {
                    SPMV_LAUNCH_FLAGS(SPMV_LAUNCH_PROD, 8192, 512);
                }
#undef SPMV_LAUNCH_FLAGS
#undef SPMV_LAUNCH_PROD
#undef SPMV_ARGS
            }
            if (m->num_long) {
                hipLaunchKernelGGL((csr_long_pieces<T, true>), dim3(m->num_partial), dim3(kBlock), 0, s,
                                   m->num_partial, m->pieces, m->col, (const T *)m->val, x,
                                   (T *)m->partial);
                hipLaunchKernelGGL((csr_long_finish<T>), dim3(m->num_long), dim3(64), 0, s,
                                   m->num_long, m->long_rows, (const T *)m->partial, y);
            }
            break;
        }
        default:
            return fail("unknown CSR variant %d", variant);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int csr_launch_any(const spmv_csr_dev *m, int variant, const void *x, void *y, hipStream_t s) {
    if (m->value_bytes == 8) return csr_launch<double>(m, variant, (const double *)x, (double *)y, s);
    return csr_launch<float>(m, variant, (const float *)x, (float *)y, s);
}

}  // namespace

extern "C" int spmv_hip_csr_run(spmv_csr_dev *m, int variant) {
    if (need_device()) return -1;
    if (!m) return fail("csr_run: NULL handle");
    return csr_launch_any(m, variant, m->x, m->y, g_stream);
}

extern "C" int spmv_hip_csr_run_on(spmv_csr_dev *m, int variant, const void *d_x, void *d_y, void *stream) {
    if (need_device()) return -1;
    if (!m || !d_x || !d_y) return fail("csr_run_on: NULL argument");
    return csr_launch_any(m, variant, d_x, d_y, stream ? (hipStream_t)stream : g_stream);
}

namespace {

// events around each launch on the stream the kernel runs on
template <typename Launch, typename Zero>
int time_loop(int warmup, int iters, float *ms_each, Launch launch, Zero zero_y) {
    // zero_y() is a no-op when the caller did not ask for the reference's memset
    if (iters <= 0 || !ms_each) return fail("time: iters must be > 0 and ms_each non-NULL");
    std::vector<hipEvent_t> ev((size_t)iters * 2);
    for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
    int rc = 0;
    for (int i = 0; i < warmup && !rc; ++i) {
        rc = zero_y();
        if (!rc) rc = launch();
    }
    for (int i = 0; i < iters && !rc; ++i) {
        rc = zero_y();
        if (rc) break;
        HIP_TRY(hipEventRecord(ev[2 * i], g_stream));
        rc = launch();
        HIP_TRY(hipEventRecord(ev[2 * i + 1], g_stream));
    }
    if (!rc) {
        HIP_TRY(hipStreamSynchronize(g_stream));
        for (int i = 0; i < iters; ++i) HIP_TRY(hipEventElapsedTime(&ms_each[i], ev[2 * i], ev[2 * i + 1]));
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

}  // namespace

extern "C" int spmv_hip_csr_time(spmv_csr_dev *m, int variant, int warmup, int iters, int zero_y,
                                 float *ms_each) {
    if (need_device()) return -1;
    if (!m) return fail("csr_time: NULL handle");
    return time_loop(
        warmup, iters, ms_each, [&] { return csr_launch_any(m, variant, m->x, m->y, g_stream); },
        [&]() -> int {
            if (zero_y) HIP_TRY(hipMemsetAsync(m->y, 0, (size_t)m->M_total * m->value_bytes, g_stream));
            return 0;
        });
}

namespace {

// `iters` back-to-back launches captured once into a hipGraph and replayed `replays` times:
// what a launch-bound loop (small matrices: an 11 us kernel against ~6 us of per-launch host
// work) costs per SpMV when the host is out of the way.  ms_per_iter = mean over the replays.
template <typename Launch>
int graph_loop(int iters, int replays, float *ms_per_iter, Launch launch) {
    if (iters <= 0 || replays <= 0 || !ms_per_iter) return fail("time_graph: bad arguments");
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    hipError_t e = hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return fail("hipStreamBeginCapture failed: %s", hipGetErrorString(e));
    for (int i = 0; i < iters && !rc; ++i) rc = launch();
    e = hipStreamEndCapture(g_stream, &graph);
    if (!rc && e != hipSuccess) rc = fail("hipStreamEndCapture failed: %s", hipGetErrorString(e));
    if (!rc) {
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) rc = fail("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    if (!rc) {
        e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) e = hipGraphLaunch(exec, g_stream);  // warm-up replay
        if (e == hipSuccess) e = hipEventRecord(e0, g_stream);
        for (int r = 0; r < replays && e == hipSuccess; ++r) e = hipGraphLaunch(exec, g_stream);
        if (e == hipSuccess) e = hipEventRecord(e1, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) rc = fail("graph replay failed: %s", hipGetErrorString(e));
        else *ms_per_iter = ms / ((float)replays * (float)iters);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
}

}  // namespace

extern "C" int spmv_hip_csr_time_graph(spmv_csr_dev *m, int variant, int iters, int replays, float *ms_per_iter) {
    if (need_device()) return -1;
    if (!m) return fail("csr_time_graph: NULL handle");
    return graph_loop(iters, replays, ms_per_iter, [&] { return csr_launch_any(m, variant, m->x, m->y, g_stream); });
}

// ----------------------------------------------------------------- HLL
namespace {

// Cut the rows of the flat slab into workgroup windows for hll_lds: consecutive rows whose
// slots, counted from the even slot at or below the first row's start, fit `cap` (at most
// kStreamRowsCap rows).  A row that alone does not fit gets a window of its own.  Returns
// the widest window (in slots, from its even base).
long long hll_build_blocks(int M, int hacks, const long long *off, const int *mz, int cap,
                           std::vector<int4> &desc) {
    desc.clear();
    (void)hacks;
    long long widest = 0;
    auto start_of = [&](int r) { return off[r / kHack] + (long long)(r % kHack) * mz[r / kHack]; };
    int r = 0;
    while (r < M) {
        const long long s0 = start_of(r);
        const long long base = s0 & ~1LL;
        int r1 = r + 1;  // the first row is always taken (even if it alone exceeds cap)
        while (r1 < M && r1 - r < kStreamRowsCap && start_of(r1) + mz[r1 / kHack] - base <= cap) ++r1;
        const long long span = start_of(r1 - 1) + mz[(r1 - 1) / kHack] - base;
        if (r1 - r > 1 || span <= cap) widest = std::max(widest, span);
        desc.push_back(int4{r, r1 - r, (int)(s0 & 0xffffffffLL), (int)(s0 >> 32)});
        r = r1;
    }
    return widest;
}

}  // namespace

namespace {

// Windows for hll_lds_local: hll_build_blocks' cut at `cap` slots with the line limit on top
// (see csr_build_local).  ja is the flat host slab.  false: keep the gather kernel.
bool hll_build_local(int M, int N, const long long *off, const int *mz, const int *ja, long long slots_padded,
                     int cap, int lines_max, const std::vector<int4> &baseline, LocalPlan &plan) {
    constexpr int line_shift = 4;  // fp64: 16 per 128-byte line
    const int total_lines = (int)(((long long)N + 15) >> line_shift);
    std::vector<int> stamp((size_t)total_lines + 1, -1), rank((size_t)total_lines + 1, 0), cur;
    plan.lcol.assign((size_t)slots_padded + kPad, 0);
    plan.desc.clear();
    plan.hll_ldesc.clear();
    plan.lines.clear();
    int widest = 0;
    auto start_of = [&](int r) { return off[r / kHack] + (long long)(r % kHack) * mz[r / kHack]; };
    int r = 0;
    while (r < M) {
        const long long s0 = start_of(r);
        const long long base = s0 & ~1LL;
        const int blk = (int)plan.desc.size();
        cur.clear();
        int r1 = r;
        while (r1 < M && r1 - r < kStreamRowsCap && start_of(r1) + mz[r1 / kHack] - base <= cap) {
            const size_t before = cur.size();
            const long long a = start_of(r1);
            for (long long k = a; k < a + mz[r1 / kHack]; ++k) {
                const int l = ja[k] >> line_shift;
                if (stamp[l] != blk) {
                    stamp[l] = blk;
                    cur.push_back(l);
                }
            }
            if ((int)cur.size() > lines_max) {
                for (size_t k = before; k < cur.size(); ++k) stamp[cur[k]] = -1;
                cur.resize(before);
                break;
            }
            ++r1;
        }
        if (r1 == r) return false;  // a row that alone exceeds the stage or the line limit
        if (cur.empty()) cur.push_back(0);  // rows without slots: the kernel still stages one line
        std::sort(cur.begin(), cur.end());
        for (size_t k = 0; k < cur.size(); ++k) rank[cur[k]] = (int)k;
        for (long long k = s0; k < start_of(r1 - 1) + mz[(r1 - 1) / kHack]; ++k)
            plan.lcol[k] = (unsigned short)((rank[ja[k] >> line_shift] << line_shift) | (ja[k] & 15));
        plan.desc.push_back(int4{r, r1 - r, (int)(s0 & 0xffffffffLL), (int)(s0 >> 32)});
        plan.hll_ldesc.push_back(int4{(int)plan.lines.size(), (int)cur.size(),
                                      (int)(start_of(r1 - 1) + mz[(r1 - 1) / kHack] - base), 0});
        plan.lines.insert(plan.lines.end(), cur.begin(), cur.end());
        widest = std::max(widest, (int)cur.size());
        r = r1;
        if ((plan.desc.size() & 1023) == 0) {
            const size_t plain = std::lower_bound(baseline.begin(), baseline.end(), r,
                                                  [](const int4 &d, int row) { return d.x < row; }) -
                                 baseline.begin();
            if (plan.desc.size() > plain + plain / 5 + 16) return false;
        }
    }
    if (plan.desc.size() > baseline.size() + baseline.size() / 5 + 1) return false;
    plan.stage_lines = std::max(kLocalLineQuantum,
                                (widest + kLocalLineQuantum - 1) / kLocalLineQuantum * kLocalLineQuantum);
    plan.lines.insert(plan.lines.end(), (size_t)kLocalLinesMax, 0);
    return true;
}

// offsets of the hacks in the flat slab: every hack starts on an even slot
long long hll_offsets(int total_rows, const std::vector<int> &mz, std::vector<long long> &off,
                      long long &true_slots) {
    const int H = (int)mz.size();
    off.assign((size_t)H + 1, 0);
    true_slots = 0;
    for (int h = 0; h < H; ++h) {
        const int rows = (h == H - 1) ? total_rows - h * kHack : kHack;
        const long long s = (long long)rows * mz[h];
        true_slots += s;
        off[h + 1] = off[h] + ((s + 1) & ~1LL);
    }
    return off[H];
}

// workgroup windows, small arrays and vectors of a handle whose JA / AS are already on the device
int hll_finish_handle(spmv_hll_dev *m, int total_rows, int N, const std::vector<long long> &off,
                      const std::vector<int> &mz, long long true_slots, bool upload_maxnz, const int *ja_host,
                      int matrix_rows = -1, int row0 = 0) {
    m->M_total = matrix_rows < 0 ? total_rows : matrix_rows;
    m->row0 = row0;
    const int H = (int)mz.size();
    // like the CSR stream kernel: larger stages for matrices that have plenty of work
    const int cap = true_slots >= (16LL << 20) ? kHllCap : kHllCap / 2;
    std::vector<int4> hdesc;
    const long long widest =
        std::max<long long>(2 * kStreamUnit, hll_build_blocks(total_rows, H, off.data(), mz.data(), cap, hdesc));
    m->M = total_rows;
    m->N = N;
    m->hacks = H;
    m->slots = true_slots;
    m->num_blocks = (int)hdesc.size();
    m->stage_slots = (int)std::min<long long>(kHllCap, (widest + kStreamUnit - 1) / kStreamUnit * kStreamUnit);
    int rc = 0;
    if (!m->hack_off) rc |= upload_array(&m->hack_off, off.data(), off.size(), 0);
    if (!rc && upload_maxnz) rc |= upload_array(&m->maxnz, mz.data(), mz.size(), 1);
    if (!rc) rc |= upload_array(&m->hdesc, hdesc.data(), hdesc.size(), 1);
    if (!rc) {
        const size_t x_bytes = std::max<size_t>((size_t)N, 1) * sizeof(double) + kLineBytes;  // whole-line reads
        hipError_t e = hipMalloc((void **)&m->x, x_bytes);
        if (e == hipSuccess) e = hipMalloc((void **)&m->y, std::max<size_t>((size_t)m->M_total, 1) * sizeof(double));
        if (e == hipSuccess) e = hipMemset(m->x, 0, x_bytes);
        if (e == hipSuccess) e = hipMemset(m->y, 0, std::max<size_t>((size_t)m->M_total, 1) * sizeof(double));
        if (e != hipSuccess) rc = fail("hipMalloc(x/y) failed: %s", hipGetErrorString(e));
    }
    m->device_bytes = off.size() * 8 + mz.size() * 4 + ((size_t)off[H] + kPad) * 12 + hdesc.size() * 16 +
                      ((size_t)N + (size_t)total_rows) * 8;
    // the x-window kernel: windows of 2048 slots, 16-bit local JA (needs the slab on the host)
    if (!rc && ja_host && g_stream_local && true_slots > 0) {
        std::vector<int4> plain;
        LocalPlan local;
        hll_build_blocks(total_rows, H, off.data(), mz.data(), 2048, plain);
        if (hll_build_local(total_rows, N, off.data(), mz.data(), ja_host, off[H], 2048, kLocalLinesMax, plain, local)) {
            rc |= upload_array(&m->ldesc4, local.desc.data(), local.desc.size(), 1);
            if (!rc) rc |= upload_array(&m->ldesc, local.hll_ldesc.data(), local.hll_ldesc.size(), 1);
            if (!rc) rc |= upload_array(&m->lines, local.lines.data(), local.lines.size(), 0);
            if (!rc) rc |= upload_array(&m->lja, local.lcol.data(), local.lcol.size(), 0);
            if (!rc) {
                m->local_blocks = (int)local.desc.size();
                m->local_stage_lines = local.stage_lines;
                m->local_lines = (long long)local.lines.size() - kLocalLinesMax;
                m->device_bytes += local.desc.size() * 32 + local.lines.size() * 4 + local.lcol.size() * 2;
            }
        }
    }
    const double mean = total_rows ? (double)true_slots / total_rows : 0.0;
    m->lanes_per_row = std::min(32, std::max(2, pow2_floor(std::max(2, (int)(mean / 2.0 + 0.5)))));
    return rc;
}

}  // namespace

// Hacks [hack0, hack1) of the matrix, i.e. rows [32 hack0, min(32 hack1, total_rows)): one
// rank's share under the reference's hack partitioner (prepare_thread_distribution_hll,
// src/hll_matrix.c:410-540); y stays full length, the kernels write this handle's rows.
extern "C" int spmv_hip_hll_upload_part(const HLLMatrix *hll, int total_rows, int N, int hack0, int hack1,
                                        spmv_hll_dev **out) {
    if (need_device()) return -1;
    if (!hll || !out) return fail("hll_upload: NULL argument");
    *out = nullptr;
    if ((unsigned long long)N * 8 >= (1ull << 32))
        return fail("hll_upload: N = %d exceeds the 32-bit gather offset range of the kernels", N);
    const int Hall = hll->num_blocks;
    if (Hall != (total_rows + kHack - 1) / kHack)
        return fail("hll_upload: %d hacks do not match %d rows", Hall, total_rows);
    if (hack0 < 0 || hack1 < hack0 || hack1 > Hall)
        return fail("hll_upload: bad hack range [%d, %d) of %d", hack0, hack1, Hall);
    const int H = hack1 - hack0;
    const int row0 = hack0 * kHack;
    const int rows = std::min(hack1 * kHack, total_rows) - std::min(row0, total_rows);

    std::vector<int> mz((size_t)H, 0);
    for (int h = 0; h < H; ++h) {
        const ELLPACKBlock *b = &hll->blocks[hack0 + h];
        const int expect = (hack0 + h == Hall - 1) ? total_rows - (hack0 + h) * kHack : kHack;
        if (b->M != expect) return fail("hll_upload: hack %d holds %d rows, expected %d", hack0 + h, b->M, expect);
        if (b->MAXNZ < 0 || (b->MAXNZ > 0 && (!b->JA || !b->AS)))
            return fail("hll_upload: hack %d is malformed", hack0 + h);
        mz[h] = b->MAXNZ;
        const long long s = (long long)b->M * b->MAXNZ;
        for (long long k = 0; k < s; ++k)
            if ((unsigned)b->JA[k] >= (unsigned)N)
                return fail("hll_upload: column index %d in hack %d is outside [0, %d)", b->JA[k], hack0 + h, N);
    }
    std::vector<long long> off;
    long long true_slots = 0;
    const long long S = hll_offsets(rows, mz, off, true_slots);
    if (S > (1LL << 40)) return fail("hll_upload: %lld padded slots is unreasonable", S);

    // pack every hack into one flat pair of host arrays, then two copies
    std::vector<int> ja((size_t)S + kPad, 0);
    std::vector<double> as((size_t)S + kPad, 0.0);
    for (int h = 0; h < H; ++h) {
        const ELLPACKBlock *b = &hll->blocks[hack0 + h];
        const size_t s = (size_t)b->M * b->MAXNZ;
        if (!s) continue;
        memcpy(&ja[(size_t)off[h]], b->JA, s * sizeof(int));
        memcpy(&as[(size_t)off[h]], b->AS, s * sizeof(double));
    }
    spmv_hll_dev *m = new (std::nothrow) spmv_hll_dev();
    if (!m) return fail("hll_upload: out of host memory");
    int rc = upload_array(&m->JA, ja.data(), ja.size(), 0);
    if (!rc) rc |= upload_array(&m->AS, as.data(), as.size(), 0);
    if (!rc) rc |= hll_finish_handle(m, rows, N, off, mz, true_slots, true, ja.data(), total_rows, row0);
    if (rc) {
        spmv_hip_hll_free(m);
        return -1;
    }
    *out = m;
    return 0;
}

extern "C" int spmv_hip_hll_upload(const HLLMatrix *hll, int total_rows, int N, spmv_hll_dev **out) {
    if (!hll) return fail("hll_upload: NULL argument");
    return spmv_hip_hll_upload_part(hll, total_rows, N, 0, hll->num_blocks, out);
}

// HLL built on the device from a resident CSR matrix (whole matrix, fp64).
extern "C" int spmv_hip_hll_from_csr(const spmv_csr_dev *csr, spmv_hll_dev **out) {
    if (need_device()) return -1;
    if (!csr || !out) return fail("hll_from_csr: NULL argument");
    *out = nullptr;
    // a row block works when it starts on a hack boundary and ends on one (or at the last row)
    if (csr->value_bytes != 8 || csr->row0 % kHack != 0 ||
        ((csr->row0 + csr->M_local) % kHack != 0 && csr->row0 + csr->M_local != csr->M_total))
        return fail("hll_from_csr: needs a whole fp64 CSR matrix (or a row block cut on hack boundaries)");
    const int M = csr->M_local, N = csr->N, H = (M + kHack - 1) / kHack;
    spmv_hll_dev *m = new (std::nothrow) spmv_hll_dev();
    if (!m) return fail("hll_from_csr: out of host memory");
    std::vector<int> mz((size_t)H, 0);
    std::vector<long long> off;
    long long true_slots = 0;
    int rc = 0;
    do {
        hipError_t e = hipMalloc((void **)&m->maxnz, ((size_t)H + 1) * sizeof(int));
        if (e != hipSuccess) { rc = fail("hipMalloc(maxnz) failed: %s", hipGetErrorString(e)); break; }
        if (H > 0) {
            hipLaunchKernelGGL(hll_hack_maxnz, dim3((H + kBlock - 1) / kBlock), dim3(kBlock), 0, g_stream, M, H,
                               csr->row_ptr, m->maxnz);
            e = hipMemcpyAsync(mz.data(), m->maxnz, (size_t)H * sizeof(int), hipMemcpyDeviceToHost, g_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
            if (e != hipSuccess) { rc = fail("hll_from_csr: maxnz pass failed: %s", hipGetErrorString(e)); break; }
        }
        const long long S = hll_offsets(M, mz, off, true_slots);  // H-sized scan on the host
        if (S > (1LL << 40)) { rc = fail("hll_from_csr: %lld padded slots is unreasonable", S); break; }
        e = hipMalloc((void **)&m->JA, ((size_t)S + kPad) * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&m->AS, ((size_t)S + kPad) * sizeof(double));
        if (e == hipSuccess) e = hipMemsetAsync(m->JA, 0, ((size_t)S + kPad) * sizeof(int), g_stream);
        if (e == hipSuccess) e = hipMemsetAsync(m->AS, 0, ((size_t)S + kPad) * sizeof(double), g_stream);
        if (e != hipSuccess) { rc = fail("hll_from_csr: slab allocation failed: %s", hipGetErrorString(e)); break; }
        rc = upload_array(&m->hack_off, off.data(), off.size(), 0);
        if (rc) break;
        if (M > 0) {
            hipLaunchKernelGGL((hll_fill_from_csr<double>), dim3((M + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock),
                               0, g_stream, M, csr->row_ptr, csr->col, (const double *)csr->val, m->hack_off,
                               m->maxnz, m->JA, m->AS);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
            if (e != hipSuccess) { rc = fail("hll_from_csr: fill failed: %s", hipGetErrorString(e)); break; }
        }
        // the x-window plan is built on the host from the finished JA (one D2H copy of 4 bytes per slot)
        std::vector<int> ja_host;
        if (g_stream_local && S > 0) {
            ja_host.resize((size_t)S);
            e = hipMemcpy(ja_host.data(), m->JA, (size_t)S * sizeof(int), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { rc = fail("hll_from_csr: JA download failed: %s", hipGetErrorString(e)); break; }
        }
        rc = hll_finish_handle(m, M, N, off, mz, true_slots, false, ja_host.empty() ? nullptr : ja_host.data(),
                               csr->M_total, csr->row0);
    } while (0);
    if (rc) {
        spmv_hip_hll_free(m);
        return -1;
    }
    *out = m;
    return 0;
}

// flat slab back to the host (tests; hosts that want the HLL arrays): hack_off[hacks + 1],
// maxnz[hacks], JA / AS [hack_off[hacks]]; any pointer may be NULL
extern "C" int spmv_hip_hll_download(const spmv_hll_dev *m, long long *hack_off, int *maxnz, int *JA, double *AS) {
    if (need_device()) return -1;
    if (!m) return fail("hll_download: NULL handle");
    HIP_TRY(hipStreamSynchronize(g_stream));
    std::vector<long long> off((size_t)m->hacks + 1);
    HIP_TRY(hipMemcpy(off.data(), m->hack_off, off.size() * sizeof(long long), hipMemcpyDeviceToHost));
    if (hack_off) memcpy(hack_off, off.data(), off.size() * sizeof(long long));
    if (maxnz && m->hacks) HIP_TRY(hipMemcpy(maxnz, m->maxnz, (size_t)m->hacks * sizeof(int), hipMemcpyDeviceToHost));
    const size_t S = (size_t)off[m->hacks];
    if (JA && S) HIP_TRY(hipMemcpy(JA, m->JA, S * sizeof(int), hipMemcpyDeviceToHost));
    if (AS && S) HIP_TRY(hipMemcpy(AS, m->AS, S * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void spmv_hip_hll_free(spmv_hll_dev *m) {
    if (!m) return;
    (void)hipFree(m->hack_off);
    (void)hipFree(m->maxnz);
    (void)hipFree(m->JA);
    (void)hipFree(m->AS);
    (void)hipFree(m->hdesc);
    (void)hipFree(m->ldesc4);
    (void)hipFree(m->ldesc);
    (void)hipFree(m->lines);
    (void)hipFree(m->lja);
    (void)hipFree(m->x);
    (void)hipFree(m->y);
    delete m;
}

extern "C" int spmv_hip_hll_info(const spmv_hll_dev *m, spmv_dev_info *out) {
    if (!m || !out) return fail("hll_info: NULL argument");
    memset(out, 0, sizeof *out);
    out->M_local = m->M;
    out->M_total = m->M_total;
    out->row0 = m->row0;
    out->N = m->N;
    out->value_bytes = 8;
    out->auto_variant = m->auto_variant;
    out->lanes_per_row = m->lanes_per_row;
    out->stream_blocks = m->num_blocks;
    out->slots = m->slots;
    out->hacks = m->hacks;
    // S (val + 4) + 12 H + val (M + N)
    out->algo_bytes = m->slots * 12 + 12LL * m->hacks + 8LL * ((long long)m->M + m->N);
    out->device_bytes = (long long)m->device_bytes;
    out->local_blocks = m->local_blocks;
    out->local_stage_lines = m->local_stage_lines;
    out->local_lines = m->local_lines;
    if (m->local_blocks > 0)
        out->stream_bytes = m->slots * 10 + 4 * m->local_lines + 32LL * m->local_blocks + 12LL * m->hacks +
                            8LL * ((long long)m->M + m->N);
    return 0;
}

extern "C" void *spmv_hip_hll_x_ptr(spmv_hll_dev *m) { return m ? m->x : nullptr; }
extern "C" void *spmv_hip_hll_y_ptr(spmv_hll_dev *m) { return m ? m->y : nullptr; }

extern "C" int spmv_hip_hll_set_x(spmv_hll_dev *m, const double *x_host) {
    if (need_device()) return -1;
    if (!m || !x_host) return fail("hll_set_x: NULL argument");
    HIP_TRY(hipMemcpyAsync(m->x, x_host, (size_t)m->N * 8, hipMemcpyHostToDevice, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int spmv_hip_hll_get_y(spmv_hll_dev *m, double *y_host) {
    if (need_device()) return -1;
    if (!m || !y_host) return fail("hll_get_y: NULL argument");
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(y_host, m->y, (size_t)m->M_total * 8, hipMemcpyDeviceToHost));  // whole y, as for CSR
    return 0;
}

namespace {

template <int L>
void launch_hll_vector(const spmv_hll_dev *m, const double *x, double *y, hipStream_t s) {
    constexpr int rows = kBlock / L;
    hipLaunchKernelGGL((hll_vector<double, L>), dim3((m->M + rows - 1) / rows), dim3(kBlock), 0, s,
                       m->M, m->hack_off, m->maxnz, m->JA, m->AS, x, y);
}

int hll_launch(const spmv_hll_dev *m, int variant, const double *x, double *y_full, hipStream_t s) {
    if (m->M == 0) return 0;
    double *y = y_full + m->row0;  // the kernels number this handle's rows from 0
    if (variant == SPMV_HLL_AUTO) variant = m->auto_variant;
    switch (variant) {
        case SPMV_HLL_THREAD_ROW:
            hipLaunchKernelGGL((hll_thread_row<double>), dim3((m->M + kBlock - 1) / kBlock),
                               dim3(kBlock), 0, s, m->M, m->hack_off, m->maxnz, m->JA, m->AS, x, y);
            break;
        case SPMV_HLL_SUBWAVE:
            switch (m->lanes_per_row) {
                case 2: launch_hll_vector<2>(m, x, y, s); break;
                case 4: launch_hll_vector<4>(m, x, y, s); break;
                case 8: launch_hll_vector<8>(m, x, y, s); break;
                case 16: launch_hll_vector<16>(m, x, y, s); break;
                default: launch_hll_vector<32>(m, x, y, s); break;
            }
            break;
        case SPMV_HLL_LDS: {
            if ((g_stream_kind == -1 || g_stream_kind == 5) && m->local_blocks > 0 &&
                ((uintptr_t)x & (kLineBytes - 1)) == 0) {
                const int lchunk = g_stream_xcd < 0 ? (m->local_blocks + 7) / 8 : (g_stream_xcd ? g_stream_xcd : 16);
                const int lgrid = (m->local_blocks + 8 * lchunk - 1) / (8 * lchunk) * (8 * lchunk);
                const size_t llds = std::max((size_t)2048 * sizeof(double), (size_t)m->local_stage_lines * kLineBytes);
                const bool lnt = g_local_nt < 0 ? m->slots * 10 > (128LL << 20) : g_local_nt != 0;
                if (lnt)
                    hipLaunchKernelGGL((hll_lds_local<double, true, 2048>), dim3(lgrid), dim3(kBlock), llds, s,
                                       m->local_blocks, lchunk, m->ldesc4, m->ldesc, m->lines, m->hack_off, m->maxnz,
                                       m->lja, m->AS, x, y);
                else
                    hipLaunchKernelGGL((hll_lds_local<double, false, 2048>), dim3(lgrid), dim3(kBlock), llds, s,
                                       m->local_blocks, lchunk, m->ldesc4, m->ldesc, m->lines, m->hack_off, m->maxnz,
                                       m->lja, m->AS, x, y);
                break;
            }
            const size_t lds = 32 + ((size_t)m->stage_slots + 2) * sizeof(double);
#define SPMV_HLL_LDS_LAUNCH(MAXU)                                                                  \
    hipLaunchKernelGGL((hll_lds<double, true, MAXU>), dim3(m->num_blocks), dim3(kBlock), lds, s,    \
                       m->stage_slots, m->hdesc, m->hack_off, m->maxnz, m->JA, m->AS, x, y)
            const int units = m->stage_slots / kStreamUnit;
            if (units <= 2) SPMV_HLL_LDS_LAUNCH(2);
            else if (units <= 4) SPMV_HLL_LDS_LAUNCH(4);
            else if (units <= 6) SPMV_HLL_LDS_LAUNCH(6);
            else SPMV_HLL_LDS_LAUNCH(8);
#undef SPMV_HLL_LDS_LAUNCH
            break;
        }
        default:
            return fail("unknown HLL variant %d", variant);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" int spmv_hip_hll_run(spmv_hll_dev *m, int variant) {
    if (need_device()) return -1;
    if (!m) return fail("hll_run: NULL handle");
    return hll_launch(m, variant, m->x, m->y, g_stream);
}

extern "C" int spmv_hip_hll_run_on(spmv_hll_dev *m, int variant, const void *d_x, void *d_y, void *stream) {
    if (need_device()) return -1;
    if (!m || !d_x || !d_y) return fail("hll_run_on: NULL argument");
    return hll_launch(m, variant, (const double *)d_x, (double *)d_y, stream ? (hipStream_t)stream : g_stream);
}

extern "C" int spmv_hip_hll_time(spmv_hll_dev *m, int variant, int warmup, int iters, int zero_y,
                                 float *ms_each) {
    if (need_device()) return -1;
    if (!m) return fail("hll_time: NULL handle");
    return time_loop(
        warmup, iters, ms_each, [&] { return hll_launch(m, variant, m->x, m->y, g_stream); },
        [&]() -> int {
            if (zero_y) HIP_TRY(hipMemsetAsync(m->y, 0, (size_t)m->M_total * 8, g_stream));
            return 0;
        });
}

extern "C" int spmv_hip_hll_time_graph(spmv_hll_dev *m, int variant, int iters, int replays, float *ms_per_iter) {
    if (need_device()) return -1;
    if (!m) return fail("hll_time_graph: NULL handle");
    return graph_loop(iters, replays, ms_per_iter, [&] { return hll_launch(m, variant, m->x, m->y, g_stream); });
}

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

namespace {

// one step = this rank's kernel, then the all-gatherv of y (when a communicator exists)
template <typename Launch>
int step_loop(void *y, int value_bytes, const int *bounds, int warmup, int iters, float *ms_kernel,
              float *ms_exchange, Launch launch) {
    if (iters <= 0) return fail("step_time: iters must be > 0");
    std::vector<hipEvent_t> ev((size_t)iters * 3);
    for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
    int rc = 0;
    for (int i = -warmup; i < iters && !rc; ++i) {
        if (i >= 0) HIP_TRY(hipEventRecord(ev[3 * i], g_stream));
        rc = launch();
        if (i >= 0) HIP_TRY(hipEventRecord(ev[3 * i + 1], g_stream));
        if (!rc && g_comm) rc = spmv_hip_comm_allgatherv(y, bounds, value_bytes, g_stream);
        if (i >= 0) HIP_TRY(hipEventRecord(ev[3 * i + 2], g_stream));
    }
    if (!rc) {
        HIP_TRY(hipStreamSynchronize(g_stream));
        for (int i = 0; i < iters; ++i) {
            float a = 0, b = 0;
            HIP_TRY(hipEventElapsedTime(&a, ev[3 * i], ev[3 * i + 1]));
            HIP_TRY(hipEventElapsedTime(&b, ev[3 * i + 1], ev[3 * i + 2]));
            if (ms_kernel) ms_kernel[i] = a;
            if (ms_exchange) ms_exchange[i] = b;
        }
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

}  // namespace

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
