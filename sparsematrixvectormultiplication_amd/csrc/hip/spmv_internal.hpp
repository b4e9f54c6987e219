// spmv_internal.hpp -- what the translation units behind include/spmv_hip.h share: error
// reporting, library state, the two device handles, upload helpers and the timing loops.
// Not installed; nothing here is part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#include <string>
#include <vector>

#include "csr_kernels.hpp"
#ifdef SPMV_EXPERIMENTAL  // make EXPERIMENTAL=1: the stream variants that lost their A/B + the ablation probes
#include "csr_kernels_experimental.hpp"
#endif
#include "hll_kernels.hpp"
#include "tile_kernels.hpp"
#include "spmv_hip.h"

using namespace spmv;

// ------------------------------------------------------------------ state (spmv_device.hip)
extern int g_device;
extern hipStream_t g_stream;
extern hipStream_t g_stream2;  // second stream: the halo exchange that runs beside the interior blocks
extern ncclComm_t g_comm;
extern int g_comm_rank, g_comm_size;

// kernel tuning knobs (spmv_hip_set_tuning); defaults are the measured best
extern int g_stream_cap;      // nnz staged per stream workgroup (fixed at upload); 0 = by matrix size
extern int g_stream_block;    // threads per csr_stream workgroup
extern int g_stream_nt;       // non-temporal loads for col/val in the gather stream kernels
extern int g_local_nt;        // same for the x-window kernels: -1 = auto (off while the matrix fits the Infinity Cache)
extern int g_stream_xcd;      // blocks per XCD run (xcd_chunked); 0 = default, -1 = one contiguous eighth per XCD
extern int g_halo_split;      // halo setup splits the handle by columns (1) or by blocks only (0)
extern int g_halo_overlap;    // power iteration with halo: exchange beside the interior blocks (1) or strictly in order (0)
extern int g_gather_mode;     // all-gatherv: 0 = one ncclBroadcast per owner in a group, 1 = padded ncclAllGather + scatter
extern int g_local_cap;       // stage of the x-window plan: 0 = auto, 1024 or 2048
extern int g_stream_local;    // build the x-window plan at upload when it pays
extern int g_plan_on_device;  // ... with the device kernels where they apply (0: always on the host)
extern int g_stream_kind;     // -1 = auto (x-window kernel when the matrix has a plan, else csr_stream), 5 = x-window,
                              // 0 = csr_stream, 1 = row walk, 2 = pipe, 3 = persistent walk, 4 = ring, 10..17 = probes
extern int g_pipe_wgs_per_cu; // resident workgroups per CU the persistent grids are sized for
extern int g_stream_tile;     // csr_tile plan at upload: -1 = auto (no x-window plan, enough rows), 0 = never, 1 = whenever no x-window plan
extern int g_tile_rows;       // rows per block: 0 = auto, else a power of two in 256..8192
extern int g_tile_lmax;       // rows longer than this stay with the split-row kernels
constexpr long long kTileMidEntries = 4LL << 20;  // (auto) ... or, for a band of dense rows, entries from which it gets one (packed plans only)
constexpr long long kSplitMinEntries = 1LL << 20;  // entries from which rows may be handed to the split-row kernels (two more launches)
constexpr long long kTileMinRows = 800000;  // (auto) rows from which a handle without an x-window plan gets a tile plan
extern int g_tile_min_pass;   // windows of a packed plan with fewer entries than this (and sparser than 1 per 16 columns) go to the remainder; 0: none
extern int g_tile_places;     // 0: the chip's (2 or 1 workgroups per CU) | the number of workgroup places the streams / the block count are made for
extern int g_tile_mid_items;  // work items of the middle tier (0: three rounds of the CUs)
extern int g_tile_items;      // work items the long rows' passes are dealt out to (about)
extern int g_tile_streams;    // 1: one csr_tile workgroup per place of the chip walks several row blocks back to back
extern int g_tile_fit;        // 1: (auto rows) the number of row blocks is fitted to whole rounds of the chip's workgroup places
extern int g_skew_rows;       // 1: gather-kernel handles hand rows far longer than the average to the split-row kernels
extern int g_tile_probe;      // measurement only: bit 0 loads, staging and barriers only, bit 1 no gathers, bit 2 no run sums (y is then wrong), bit 3 one workgroup per CU
extern int g_tile_balance;    // 1: row blocks of about equal entry counts (keeps the workgroups in step), 0: equal row counts
extern int g_tile_long;       // 1: the rows beyond the tile limit get a tile plan of their own (compacted rows, work items, slabs)
extern int g_tile_pack;       // 1: passes that can be staged store head | row | column offset in one 32-bit word (no key read)
extern int g_tile_density;    // a pass is staged when it holds at least one entry per this many columns of its window
extern int g_tile_mid_lo;     // a scattered matrix's rows longer than this (up to tile_lmax) form the middle tier ("tile_mid_lo")
extern int g_local_patterns;  // x-window plans: -1 auto (a pattern plan where it holds a quarter of the slots at most), 0 never, 1 always ("local_patterns")
extern int g_tile_expand;     // plans with gather passes: -1 auto, 0 never, 1 always: x expanded into entry order ahead of csr_tile ("tile_expand")
extern int g_tile_gather_ahead;  // 1: plans with gather passes run the csr_tile instantiation that gathers one pass early
extern int g_tile_mid;        // 1: scattered plans get that tier (when it holds >= 2^22 entries), 0: never
extern int g_place_tries;     // other placements of the value array upload tries for large handles (0: none)
extern int g_tile_plan_on_device;  // 1: the csr_tile plan is built by kernels (tile_plan_device.hpp), 0: by host threads (tile_plan.hpp)
extern int g_num_cus;
extern int g_probe_mask;      // csr_probe: table size - 1 (entries) of the folded gather

int fail(const char *fmt, ...);  // records the message for spmv_hip_last_error(), returns -1
int need_device();               // 0, or -1 when spmv_hip_init() has not succeeded

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

// The C-ABI never lets a C++ exception cross into the caller: host-side packing uses std::vector,
// whose allocation failure (a matrix too large for host memory) becomes -1 + message like any other.
template <typename F>
int guarded(const char *what, F body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail("%s: out of host memory", what);
    } catch (const std::exception &e) {
        return fail("%s: %s", what, e.what());
    }
}

constexpr int kRowPtrPad = 384;  // entries behind row_ptr (kernels that stage whole row_ptr segments)
constexpr int kPad = 8192 + 64;  // zero entries behind col/val: the stream / LDS kernels stage
                                   // whole units without bounds tests (>= kStreamCapMax, kHllCap)

template <typename T>
int upload_array(T **dptr, const T *host, size_t count, size_t pad) {
    HIP_TRY(hipMalloc((void **)dptr, (count + pad) * sizeof(T)));
    if (count) HIP_TRY(hipMemcpy(*dptr, host, count * sizeof(T), hipMemcpyHostToDevice));
    if (pad) HIP_TRY(hipMemset(*dptr + count, 0, pad * sizeof(T)));
    return 0;
}

// SPMV_TRACE_UPLOAD=1: where an upload spends its time, phase by phase, on stderr (tools/time_upload.py).
// mark(name) closes the phase that began at the previous mark (the device is synchronised first, so device work is
// charged to the phase that launched it).
struct UploadTrace {
    bool on;
    double t_last;
    const char *who;
    static double now() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    explicit UploadTrace(const char *name) : on(getenv("SPMV_TRACE_UPLOAD") != nullptr), t_last(0), who(name) {
        if (on) t_last = now();
    }
    void mark(const char *phase) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const double t = now();
        fprintf(stderr, "[upload trace] %-18s %-34s %8.1f ms\n", who, phase, (t - t_last) * 1e3);
        t_last = t;
    }
};

inline int pow2_floor(int v) {
    int p = 1;
    while (p * 2 <= v) p *= 2;
    return p;
}

// The row-sum phase of the stream / LDS kernels gives every row of a block the same number of lanes, sized by the
// NUMBER of rows in the block (lanes_for_rows in csr_kernels.hpp; mirrored here for the host).  One row of hundreds of
// entries in a block of hundreds of short rows is then summed by a single lane while the other 255 wait (circuit
// matrices: adder_dcop_32-size stand-in 11.5 us where one launch costs 6.3).  Upload therefore closes a block before a
// row -- or before more rows join a long one -- whenever some lane would have to add up more than kSkewPerLane entries:
// the long row ends up in a block of few rows and gets 8..64 lanes.  Uniform matrices never get there (rows x length
// <= stage, so length / lanes <= 32).
constexpr int kSkewPerLane = 64;
inline int host_lanes_for_rows(int nrows) {
    if (nrows > kBlock / 2 || nrows <= 0) return 1;
    return std::min(64, pow2_floor(kBlock / nrows));
}
inline bool skew_cut(int nrows, int maxlen, int len) {
    if (nrows <= 0) return false;  // a block always takes its first row
    return std::max(maxlen, len) / host_lanes_for_rows(nrows + 1) > kSkewPerLane;
}

// ---------------------------------------------------------------- handles
namespace spmv {
struct TileExpansion;  // tile_plan_device.hpp
}
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
    // the pattern plan (round 3; csr_kernels.hpp, PAT): where most rows of the blocks are their predecessor shifted by a
    // constant, the kernel rebuilds the slots from a table per block and 4 bytes per row instead of reading lcol
    unsigned short *ptab = nullptr;   // [pat_slots + pad] the blocks' pattern tables
    unsigned *rinfo = nullptr;        // [M_local] start of the row's pattern in its block's table | shift << 16
    int2 *pdesc = nullptr;            // [local_blocks] {first element in ptab (even), elements}
    long long pat_slots = 0;          // elements of all tables
    int pat_max = 0;                  // the largest table (elements)
    float pat_with_us = 0, pat_without_us = 0;  // (auto) the kernel with / without the plan, timed at upload (csr_tune_patterns)
    int local_blocks = 0;             // 0: no plan (not profitable / not possible)
    int local_stage_lines = 0;        // LDS stage: most lines any block lists, in steps of 32
    int local_cap = 2048;
    long long local_lines = 0;
    // N4 overlap: x-window blocks whose listed x lines all lie in this handle's own range of x (interior) / the rest
    int *interior_ids = nullptr, *boundary_ids = nullptr;
    int num_interior = 0, num_boundary = 0;
    bool have_split = false;
    // N4 overlap below block granularity (round 3): the handle's entries split by COLUMN -- own_part holds those whose
    // column lies in the rank's own range of x (it can run before the halo has arrived), halo_part the rest; two
    // sub-handles over the same rows, launched on this handle's vectors (spmv_hip_csr_split_columns)
    spmv_csr_dev *own_part = nullptr, *halo_part = nullptr;
    long long own_entries = 0, halo_entries = 0;
    bool tiles_only = false;  // a handle made of tile plans alone (the tile side of an HLL handle): STREAM only
    // csr_tile (2-D tiles: row-block accumulators in LDS x column passes), for matrices without an x-window plan
    int tile_blocks = 0;              // 0: no tiles
    int tile_rows = 0;                // rows per block
    int tile_chunk = 0;               // entries per pass at most (2048)
    bool tile_packed = false;         // stageable passes hold packed column words (kernel instantiation with the decode)
    int tile_lds_min = 0;             // LDS bytes to ask for at least (scattered matrices: one workgroup per CU)
    int tile_passes = 0;
    int tile_max_win = 0;             // widest staged window (columns)
    long long tile_entries = 0, tile_staged = 0, tile_staged_cols = 0, tile_padded = 0;
    int *tile_block_row = nullptr;    // [tile_blocks + 1] first row of every block
    int *tile_block_pass = nullptr;   // [tile_blocks + 1]
    int4 *tile_pass = nullptr;        // [tile_passes] in stream order
    int tile_rem_rows = 0;            // rows with remainder entries (windows too sparse for a pass), tile_remainder
    long long tile_rem_entries = 0;
    int *tile_rem_row = nullptr, *tile_rem_ptr = nullptr, *tile_rem_col = nullptr;  // [rows], [rows + 1], [entries]
    void *tile_rem_val = nullptr;
    int tile_streams = 0;             // workgroups of the csr_tile launch: each walks the blocks of one stream
    int *tile_stream_block = nullptr; // [tile_streams + 1] first block (in tile_sblock_rows) of every stream
    int2 *tile_sblock_rows = nullptr; // [tile_blocks] {first row, rows} in stream order
    int *tcol = nullptr;              // [tile_padded + kTileChunkMax]
    unsigned short *tkey = nullptr;
    void *tval = nullptr;
    // the expansion of x for a plan with gather passes (round 3; tile_kernels.hpp, XE): x value per tile entry, written
    // by tile_expand ahead of every csr_tile launch of this handle, and what tile_expand walks
    void *xe = nullptr;               // [tile_padded + kTileChunkMax]
    spmv::TileExpansion *expansion = nullptr;
    // long-row tile plan: the rows beyond the tile limit, compacted into their own row blocks; a block's passes
    // are dealt out to several workgroups (work items), each leaves its accumulators in a slab
    struct long_tiles {
        int blocks = 0, rows = 0, rows_per_block = 0, passes = 0, items = 0, max_win = 0;
        long long entries = 0, padded = 0, staged = 0, staged_cols = 0;
        bool packed = false;
        int *block_row = nullptr, *block_pass = nullptr, *block_of_row = nullptr, *item_first = nullptr, *row_map = nullptr;
        int4 *pass = nullptr, *work = nullptr;
        int *tcol = nullptr;
        unsigned short *tkey = nullptr;
        void *tval = nullptr, *slab = nullptr;
    } lt, mt;  // lt: rows beyond tile_lmax (2048 compacted rows per block); mt (round 3): the MIDDLE tier of a scattered
               // matrix, rows of tile_mid_lo < entries <= tile_lmax in blocks as tall as the LDS takes, so that their
               // column ranges hold enough entries to be staged as well
    int4 *tile_long_rows = nullptr;   // rows beyond the tile limit {row, first slot, pieces, 0} ...
    int4 *tile_pieces = nullptr;      // ... and their pieces, cut at column stripes, stripe by stripe
    int tile_num_long = 0, tile_num_pieces = 0;
    // heuristics
    int lanes_per_row = 16;
    int auto_variant = SPMV_CSR_STREAM;
    int max_row = 0;
    size_t device_bytes = 0;
    int place_tries = 0;  // placement tuning at upload (csr_tune_placement): placements timed, first / kept time
    float place_first_us = 0, place_best_us = 0;
    // arrays moved to a chosen address (spmv_hip_csr_relocate): the field points INTO `raw`, which is what gets freed
    // (vmm: the memory came from hipMemCreate + hipMemAddressReserve + hipMemMap, `raw` is the reserved range)
    struct relocated { void **field; void *raw; size_t size; bool vmm = false; hipMemGenericAllocationHandle_t phys = {}; size_t mapped = 0; };
    std::vector<relocated> relocs;
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
    unsigned *row_seg = nullptr;   // [M] a row's (first slot in its window | slots << 16)
    // the pattern plan of the windows (round 3; see spmv_csr_dev): hll_lds_local<.., PAT> does not read lja
    unsigned short *ptab = nullptr;
    unsigned *rinfo = nullptr;
    int2 *pdesc = nullptr;
    long long pat_slots = 0;
    float pat_with_us = 0, pat_without_us = 0;
    spmv_csr_dev *tiles = nullptr; // csr_tile over the slab's rows (padding slots included), when the slab gets no x-window plan
    int local_blocks = 0, local_stage_lines = 0;
    long long local_lines = 0;
    double *x = nullptr;
    double *y = nullptr;
    int lanes_per_row = 8;
    int auto_variant = SPMV_HLL_LDS;
    size_t device_bytes = 0;
    int place_tries = 0;  // placement tuning at upload (hll_tune_placement)
    float place_first_us = 0, place_best_us = 0;
};

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

// spmv_csr.hip: a handle around device arrays that already hold the matrix (ownership passes to the handle
// on success only); col / val carry kPad zeroed entries behind the last one
int csr_adopt_f64(int M, int N, const int *row_ptr_host, int *d_col, double *d_val, spmv_csr_dev **out);
// spmv_csr.hip: tile plans alone for rows given as (first entry, length) over host arrays (an HLL slab's rows);
// *out = NULL when the rows get no plan
// (col / val: host copies of the rows' arrays, or NULL when d_col / d_val -- the same arrays on the device -- are given
// and the plan is built there)
int csr_tiles_from_rows_f64(int M_local, int M_total, int row0, int N, const int *row_begin, const int *row_len,
                            long long entries, const int *col, const double *val, spmv_csr_dev **out,
                            const int *d_col = nullptr, const double *d_val = nullptr);

int csr_tile_digest(const spmv_csr_dev *m, unsigned long long *out);  // spmv_csr.hip: see spmv_hip_csr_tile_digest
// launchers the timing / exchange code calls across translation units
int csr_launch_any(const spmv_csr_dev *m, int variant, const void *x, void *y, hipStream_t s);
// part 0: the interior x-window blocks only; part 1: everything else (boundary blocks, split rows)
int csr_launch_part(const spmv_csr_dev *m, int part, const void *x, void *y, hipStream_t s);
// column split: part 0: y = A_own x (own range of x only); part 1: y += A_halo x (after the halo has arrived)
int csr_launch_split(const spmv_csr_dev *m, int part, const void *x, void *y, hipStream_t s);
int hll_launch(const spmv_hll_dev *m, int variant, const double *x, double *y_full, hipStream_t s);

// ------------------------------------------------------------------ timing loops
// events around each launch on the stream the kernel runs on
template <typename Launch, typename Zero>
int time_loop(int warmup, int iters, float *ms_each, Launch launch, Zero zero_y) {
    // zero_y() is a no-op when the caller did not ask for the reference's memset
    if (iters <= 0 || !ms_each) return fail("time: iters must be > 0 and ms_each non-NULL");
    // the events are kept from call to call (per device): creating and destroying 2 x iters of them costs ~0.2 ms, which a
    // caller that brackets 20 launches with a wall clock -- bench.py on the driver's command -- sees as 10 us per step
    static std::vector<hipEvent_t> pool;
    static int pool_device = -1;
    int rc = 0;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) rc = fail("time: %s failed: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (pool_device != g_device) {  // (events of another device's context: let them go with it)
        pool.clear();
        pool_device = g_device;
    }
    while (pool.size() < (size_t)iters * 2) {
        hipEvent_t e = nullptr;
        if (!hip_ok(hipEventCreate(&e), "hipEventCreate")) return rc;
        pool.push_back(e);
    }
    hipEvent_t *ev = pool.data();
    for (int i = 0; i < warmup && !rc; ++i) {
        rc = zero_y();
        if (!rc) rc = launch();
    }
    for (int i = 0; i < iters && !rc; ++i) {
        rc = zero_y();
        if (rc) break;
        if (!hip_ok(hipEventRecord(ev[2 * i], g_stream), "hipEventRecord")) break;
        rc = launch();
        if (rc) break;
        if (!hip_ok(hipEventRecord(ev[2 * i + 1], g_stream), "hipEventRecord")) break;
    }
    if (!rc && hip_ok(hipStreamSynchronize(g_stream), "hipStreamSynchronize"))
        for (int i = 0; i < iters; ++i)
            if (!hip_ok(hipEventElapsedTime(&ms_each[i], ev[2 * i], ev[2 * i + 1]), "hipEventElapsedTime")) break;
    return rc;
}

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

// one step = this rank's kernel, then the all-gatherv of y (when a communicator exists)
template <typename Launch>
int step_loop(void *y, int value_bytes, const int *bounds, int warmup, int iters, float *ms_kernel,
              float *ms_exchange, Launch launch) {
    if (iters <= 0) return fail("step_time: iters must be > 0");
    // (events kept from call to call, per device: see time_loop)
    static std::vector<hipEvent_t> pool;
    static int pool_device = -1;
    int rc = 0;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) rc = fail("step_time: %s failed: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (pool_device != g_device) {
        pool.clear();
        pool_device = g_device;
    }
    while (pool.size() < (size_t)iters * 3) {
        hipEvent_t e = nullptr;
        if (!hip_ok(hipEventCreate(&e), "hipEventCreate")) return rc;
        pool.push_back(e);
    }
    hipEvent_t *ev = pool.data();
    for (int i = -warmup; i < iters && !rc; ++i) {
        if (i >= 0 && !hip_ok(hipEventRecord(ev[3 * i], g_stream), "hipEventRecord")) break;
        rc = launch();
        if (rc) break;
        if (i >= 0 && !hip_ok(hipEventRecord(ev[3 * i + 1], g_stream), "hipEventRecord")) break;
        if (g_comm) rc = spmv_hip_comm_allgatherv(y, bounds, value_bytes, g_stream);
        if (rc) break;
        if (i >= 0 && !hip_ok(hipEventRecord(ev[3 * i + 2], g_stream), "hipEventRecord")) break;
    }
    if (!rc && hip_ok(hipStreamSynchronize(g_stream), "hipStreamSynchronize")) {
        for (int i = 0; i < iters && !rc; ++i) {
            float a = 0, b = 0;
            if (!hip_ok(hipEventElapsedTime(&a, ev[3 * i], ev[3 * i + 1]), "hipEventElapsedTime")) break;
            if (!hip_ok(hipEventElapsedTime(&b, ev[3 * i + 1], ev[3 * i + 2]), "hipEventElapsedTime")) break;
            if (ms_kernel) ms_kernel[i] = a;
            if (ms_exchange) ms_exchange[i] = b;
        }
    }
    return rc;
}
