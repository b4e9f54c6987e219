/*
 * spmv_hip.h -- C-ABI of the MI355X (gfx950) SpMV device layer.
 *
 * This is the one NEW seam the build adds to the reference's header surface.
 * It sits where the reference's CUDA driver talks to the device
 * (/root/reference/main_cuda.cu): device allocation + upload (:135-145 CSR,
 * :369-402 HLL), the six kernel launch sites (:166, :238, :317 CSR; :454,
 * :568, :637 HLL), the per-iteration result copy-back (e.g. :183) and the
 * frees (:682-685, :731-744).  The shape is the one that driver dictates:
 * upload once -> run many -> fetch y.
 *
 * Conventions (the reference's own, libs/csr_matrix.h / src/csr_matrix.c:74-78):
 *   - plain C: pointers and sizes only, no C++/torch types;
 *   - every call returns 0 on success and -1 on failure; nothing in the
 *     library calls exit(); the message of the last failure is available from
 *     spmv_hip_last_error();
 *   - host arrays are borrowed for the duration of the call only; device
 *     handles are opaque and owned by the library;
 *   - one host thread, one device per process (multi-GPU = one process per
 *     GPU; see spmv_hip_comm_* below);
 *   - there is NO CPU fallback: without a usable HIP device every compute
 *     entry point fails with -1.
 *
 * Reference-side binding: see INTEGRATION.md.
 */
#ifndef SPMV_AMD_SPMV_HIP_H
#define SPMV_AMD_SPMV_HIP_H

#include <stddef.h>

#include "csr_matrix.h"
#include "hll_matrix.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spmv_csr_dev spmv_csr_dev; /* a CSR matrix (or a row block of one) resident in HBM */
typedef struct spmv_hll_dev spmv_hll_dev; /* an HLL matrix resident in HBM as one flat slab */

/* CSR kernel selection.  1..3 answer the reference's three CUDA CSR kernels
 * (cuda_src/csr_matrix_cuda.cu:122-241); 4 has no reference counterpart. */
enum {
    SPMV_CSR_AUTO = 0,       /* pick from the matrix' row-length statistics */
    SPMV_CSR_THREAD_ROW = 1, /* one lane per row            (replaces spmv_csr_naive_kernel) */
    SPMV_CSR_WAVE_ROW = 2,   /* one 64-lane wavefront per row, 2-wide vector loads
                                (replaces spmv_csr_warp_kernel) */
    SPMV_CSR_SUBWAVE = 3,    /* 2..32 lanes per row, picked from mean nnz/row
                                (replaces spmv_csr_warp_shared_memory_kernel's slot:
                                the x-cache idea is dropped) */
    SPMV_CSR_STREAM = 4      /* nnz-balanced row blocks streamed through LDS: csr_stream_local (x lines of
                                the block staged in LDS, 16-bit local columns) when upload found an x-window
                                plan for the matrix; csr_tile (row-block accumulators in LDS, column passes
                                that keep the gathered band of x in L2, dense passes staged in LDS) for large
                                matrices with scattered columns; else csr_stream (gathers); what AUTO resolves to */
};

/* HLL kernel selection.  1..3 answer cuda_src/hll_matrix.cu:346-479. */
enum {
    SPMV_HLL_AUTO = 0,
    SPMV_HLL_THREAD_ROW = 1, /* one lane per row over the row-major slab (spmv_hll_naive_kernel) */
    SPMV_HLL_SUBWAVE = 2,    /* a lane group per row                     (spmv_hll_warp_kernel) */
    SPMV_HLL_LDS = 3         /* row-aligned slab windows staged through LDS (spmv_hll_warp_shared_kernel_v1's
                                slot): hll_lds_local with an x-window plan; for a large slab with scattered columns
                                the 2-D tile kernel over the slab's rows; else hll_lds; what AUTO resolves to */
};

typedef struct {
    int M_local;           /* rows held by this handle                          */
    int M_total;           /* rows of the whole matrix (length of y)            */
    int N;                 /* columns (length of x)                             */
    int row0;              /* first global row of this handle's block           */
    long long nz;          /* stored entries held by this handle                */
    int value_bytes;       /* 8 = fp64, 4 = fp32                                */
    int auto_variant;      /* what SPMV_*_AUTO resolves to                      */
    int lanes_per_row;     /* SUBWAVE group width chosen at upload              */
    int stream_blocks;     /* workgroups of the STREAM / LDS kernel             */
    int long_rows;         /* rows split over several workgroups                */
    long long slots;       /* HLL only: padded slots S                          */
    int hacks;             /* HLL only: number of hacks                         */
    long long algo_bytes;  /* algorithmic HBM bytes of one SpMV  */
    long long device_bytes;/* HBM held by the handle                            */
    int local_blocks;      /* CSR: workgroups of the x-window stream kernel, 0 = matrix has no plan */
    int local_stage_lines; /* CSR: x lines (128 B) its widest block stages in LDS */
    long long local_lines; /* CSR: x lines listed over all blocks                 */
    long long stream_bytes;/* CSR: bytes the x-window kernel really streams from HBM:
                              nz (val + 2) + 4 lines + 24 blocks + 4 (M + 1) + val (M + N);
                              0 without a plan (then algo_bytes is what moves)    */
    int stream_kernel;     /* which kernel STREAM / LDS (and AUTO) launches with default tuning:
                              CSR 0 csr_stream, 1 csr_stream_local, 2 csr_stream_short, 3 csr_tile;
                              HLL 0 hll_lds, 1 hll_lds_local, 2 csr_tile over the slab's rows (padding slots included) */
    int tile_blocks;       /* CSR: row blocks (workgroups) of csr_tile, 0 = no tile plan */
    int tile_passes;       /* CSR: column passes over all blocks */
    int tile_split_rows;   /* CSR: rows beyond the tile limit (split-row kernels, stripe-ordered pieces) */
    long long tile_entries;        /* CSR: entries held by the tiles */
    long long tile_staged_entries; /* CSR: ... of which in passes whose x slice is staged in LDS */
    int tile_long_rows;            /* CSR: rows beyond the tile limit that got the long-row tile plan (compacted rows,
                                      a block's passes dealt out to many workgroups, slabs added in order) */
    int tile_long_items;           /* CSR: workgroups of that plan */
    long long tile_long_entries;   /* CSR: entries it holds */
    long long tile_staged_cols;    /* CSR: x values the staged passes copy to LDS per SpMV (all tile plans): traffic served by L2 */
    long long tile_remainder_entries; /* CSR: ... of tile_entries, in windows too sparse for a pass: added to y by tile_remainder behind the tiles */
    int tile_mid_rows;             /* CSR: rows of the MIDDLE tier of a scattered plan (128 < entries <= tile_lmax, compacted into
                                      blocks as tall as the LDS takes, packed plan, work items + slabs like the long rows) */
    int tile_mid_items;            /* CSR: workgroups of that plan */
    long long tile_mid_entries;    /* CSR: entries it holds */
    int place_tries;       /* placements of the value array that upload timed (0: not tuned -- small handle, or "place_tries" 0) */
    float place_first_us;  /* kernel time at the placement hipMalloc gave first */
    float place_best_us;   /* ... at the placement the handle kept */
    unsigned long long val_address; /* where the value array lies now (placement record of a bench line) */
    long long tile_expanded_entries; /* CSR: entry slots of a tile plan with gather passes that run on an expanded x (tile_expand
                                        writes every entry's x value in entry order, csr_tile streams it: "tile_expand") */
    long long pattern_slots; /* CSR: slots held by the pattern tables of an x-window plan (0: the kernel reads the 16-bit slot of
                                every entry) -- where most rows of a block are their predecessor shifted by a constant the
                                kernel rebuilds the slots from one table per block and 4 bytes per row ("local_patterns") */
    float pattern_with_us;    /* (auto) what upload measured for its kernel with the pattern plan ... */
    float pattern_without_us; /* ... and without (0 / 0: no plan was built); the plan stays where it is at least 2 % faster */
} spmv_dev_info;

/* ---- device ------------------------------------------------------------ */
int spmv_hip_device_count(void);               /* -1 when HIP itself is unusable */
int spmv_hip_init(int device);                 /* select device, create the library stream */
int spmv_hip_shutdown(void);
int spmv_hip_sync(void);                       /* wait for the library stream */
void *spmv_hip_stream(void);                   /* the hipStream_t kernels are launched on */
const char *spmv_hip_last_error(void);
int spmv_hip_device_name(char *buf, size_t len, int *compute_units, long long *hbm_bytes);
/* Evict L2 + Infinity Cache by streaming through a scratch buffer of `bytes`
 * (answers clear_gpu_cache / clear_cache_kernel, cuda_src/utility.cu:140-175;
 * the reference's 64 MiB is far below MI355X's 256 MiB Infinity Cache). */
int spmv_hip_flush_cache(size_t bytes);
/* Box state for bench records (round 3: the same code runs 180 or 200 us on the headline matrix depending on the box):
 * "key=value;..." with pci (bus id, to find the card under /sys/class/drm), arch, cus, sclk_khz / mclk_khz (HIP clock
 * attributes), mem_bus_bits, l2_bytes, hbm_bytes, xcds (from the chip's CU count: 32 per XCD).  Needs spmv_hip_init. */
int spmv_hip_device_state(char *buf, size_t len);
/* Read-only streaming probe: `iters` launches that each read `bytes` (a scratch buffer the library keeps) with 16-byte
 * non-temporal loads, 2048-thread-blocks grid-stride; mean / min event time per launch in ms.  What this box's HBM
 * gives a pure stream right now: the yardstick beside every kernel time in a bench line. */
int spmv_hip_stream_probe(size_t bytes, int warmup, int iters, float *ms_mean, float *ms_min);
/* the same read-only stream over device memory the caller names (16-byte aligned) */
int spmv_hip_stream_probe_at(const void *dptr, size_t bytes, int warmup, int iters, float *ms_mean, float *ms_min);
/* Gather probe: values per second the chip delivers when every lane of every gather wave-instruction reads a different
 * random line of a table of `table_bytes` (rounded down to a power of two of elements; <= 2 MiB: L2 resident on every
 * XCD), `waves_per_cu` wavefronts per CU with 8 independent gathers in flight each.  The ceiling of a gather-bound
 * kernel (csr_tile's gather passes, the gather stream kernels on scattered columns), as the stream probe is the
 * ceiling of a streaming one. */
int spmv_hip_gather_probe(int value_bytes, size_t table_bytes, int waves_per_cu, double *values_per_s);
/* Kernel tuning knobs, for A/B measurements (defaults are the measured best; also settable through the
 * environment, SPMV_TUNING="key=value,...", read by spmv_hip_init):
 *   read at upload
 *     "stream_cap"    0 (auto) | 1024 | 2048 | 3072 | 4096 | 8192 entries staged per workgroup of the gather stream
 *                     kernel; a value other than the x-window stage skips the x-window plan
 *     "stream_local"  1 | 0   build the x-window plan (16-bit local columns + line lists) when the matrix allows
 *     "plan_on_device" 1 | 0  build that plan with the device kernels where they apply (every block within the
 *                     line limit), else / otherwise on the host
 *     "local_cap"     0 (auto = 2048) | 1024 | 2048 | 3072   stage of the x-window kernels (3072: +0.5..3 % on the
 *                     nlpkkt-like matrix depending on the box, -8 % on the cant-like one)
 *     "skew_rows"     1 | 0   handles that run the gather kernels give rows longer than max(128, 16 x the average
 *                     row) to the split-row kernels (one workgroup per row piece) instead of leaving each to one lane
 *     "stream_tile"   -1 (auto) | 0 | 1   build the 2-D tile plan (csr_tile) when the matrix gets no x-window plan;
 *                     "tile_rows" 0 (auto: 32 KiB of accumulators for banded matrices, as many rows as the LDS takes for
 *                     scattered ones: 16128 fp64 / 32512 fp32) | a multiple of 256 in 256..32768 rows per block; "tile_lmax" (1536) longest row kept in
 *                     the ordinary tiles; "tile_density" (4) columns per entry up to which a pass is staged in LDS;
 *                     "tile_balance" 1 | 0 row blocks of equal entry / row counts; "tile_long" 1 | 0 | 2 a tile plan of
 *                     their own for the rows beyond tile_lmax (0: split-row kernels, 2: however few they are);
 *                     "tile_fit" 1 | 0 (with tile_rows 0) the number of row blocks is fitted to whole rounds of the
 *                     workgroups the chip holds at once (512 banded / 256 scattered), blocks up to the tallest the LDS takes;
 *                     "tile_streams" 1 | 0 one csr_tile workgroup per place of the chip walks several row blocks back to
 *                     back (0: one workgroup per block); "tile_places" 0 (the chip's: 2 or 1 per CU) | a multiple of 8: how
 *                     many workgroups the streams and the block count are made for (tests); "tile_items" (1008) work items the long rows' passes are dealt out to;
 *                     "tile_min_pass" (256) a packed plan's windows with fewer entries than this, and fewer than one per
 *                     16 columns, go to the remainder kernel instead of being a pass (0: no remainder);
 *                     "local_patterns" -1 | 0 | 1 (read at upload and at launch) x-window plans: the kernel rebuilds a block's
 *                         slots from a pattern table instead of reading them -- auto: built for streamed matrices of at least
 *                         12 entries per row whose tables hold at most a quarter of the slots, then kept only if upload
 *                         measures its kernel at least 2 % faster with it on this handle; 0 never; 1 always
 *                     "tile_mid_items" (0 = three rounds of the CUs) work items of the middle tier
 *                     "tile_expand" -1 | 0 | 1 (read at upload and at launch) a tile plan with gather passes runs on an
 *                         expanded x -- auto: from 2^22 entries on when under a tenth of them are staged, fp32 always, fp64
 *                         when x is beyond 100 MB; 0 never; 1 always
 *                     "tile_gather_ahead" 0 | 1 (read at launch) plans with gather passes send a pass's gathers out one pass
 *                     early (twice as many in flight per CU) -- measured to buy nothing (1113 vs 1108 us on config 5, 514.9 vs
 *                     514.8 on uniformly random columns: profiles/r3_ab_gather_ahead.txt), hence off;
 *                     "tile_mid" 1 | 0 a scattered plan gives its rows of more than "tile_mid_lo" entries (0 = auto: 48 for fp32,
 *                     128 for fp64; up to tile_lmax) a tier of their own -- compacted, blocks as tall as the LDS takes, every
 *                     pass staged -- when they hold >= 2^22 entries;
 *                     "tile_plan_on_device" 1 | 0 the plan is built by kernels from the CSR arrays in HBM (round 3) or by host
 *                     threads; the two builders give the same bytes;
 *                     "tile_pack" 1 | 0 banded matrices get the PACKED plan (every pass cut at the 32 KiB window and
 *                     staged, keys in the column words, kernel instantiation without gather code) unless its passes
 *                     would average fewer than 256 entries; 0: always the plan with gather passes
 *   read at launch
 *     "stream_kind"   -1 (auto: x-window kernel when the handle has a plan, csr_tile when it has tiles, else
 *                     csr_stream) | 5 x-window | 6 csr_tile |
 *                     0 csr_stream; only in a `make EXPERIMENTAL=1` build: 1 row walk | 2 persistent pipe |
 *                     3 persistent row walk | 4 loader/consumer ring | 10..17 ablation probes (measurement only)
 *     "place_tries"   (12) read at upload: how many other placements of the value array a handle that streams >= 128 MiB of
 *                     values tries (a fresh allocation each, the kernel timed 2 + 6 launches on it), keeping the fastest.
 *                     Round 3 found the x-window kernel's time on the headline matrix to depend on WHERE the values lie:
 *                     the same matrix runs in 182-187 or in 199-205 us, deterministically per address
 *                     (profiles/r3_placement_*.txt); 0 keeps what hipMalloc gave first
 *     "stream_nt" 0/1, "local_nt" -1 (auto) / 0 / 1   non-temporal hint on the streamed arrays
 *     "stream_xcd"    blocks per XCD run: 0 default (16 for the x-window kernels, dispatch order otherwise),
 *                     -1 one contiguous eighth per XCD, n > 0 runs of n
 *     "stream_block"  256 | 512 | 1024 threads (csr_stream at 4096 / 8192), "pipe_wgs_per_cu" 1..8,
 *     "probe_mask"    table size - 1 of the folded gather probe
 *     "gather_mode"   all-gatherv: 0 grouped broadcasts | 1 padded all-gather + scatter (see spmv_hip_comm_autotune) */
int spmv_hip_set_tuning(const char *key, int value);

/* raw device buffers, for callers that keep x / y on the device themselves */
int spmv_hip_malloc(void **dptr, size_t bytes);
int spmv_hip_free(void *dptr);
int spmv_hip_memcpy_h2d(void *dptr, const void *hptr, size_t bytes);
int spmv_hip_memcpy_d2h(void *hptr, const void *dptr, size_t bytes);
int spmv_hip_memset(void *dptr, int byte, size_t bytes);

/* ---- CSR --------------------------------------------------------------- */
/* Upload rows [row0, row1) of a host CSR matrix (row_ptr has M+1 entries and
 * is NOT rebased by the caller).  row0 = 0, row1 = M uploads everything.
 * x and y buffers of full length (N, M) are allocated with the handle. */
int spmv_hip_csr_upload(int M, int N, const int *row_ptr, const int *col_idx,
                        const double *values, int row0, int row1, spmv_csr_dev **out);
int spmv_hip_csr_upload_f32(int M, int N, const int *row_ptr, const int *col_idx,
                            const float *values, int row0, int row1, spmv_csr_dev **out);
/* Host-only self-check of what upload precomputes (workgroup blocks, split rows, the x-window plan: 16-bit
 * local columns + per-block line lists) for a CSR structure; needs no device.  0 when every invariant holds;
 * stats[6] (optional): gather blocks, x-window blocks (0 = no plan), listed lines, widest block's lines, long
 * rows, rows handed to the split-row kernels because they alone touch too many lines. */
int spmv_hip_csr_plan_check(int M, int N, const int *row_ptr, const int *col_idx, int value_bytes, int *stats);
/* The same for the csr_tile plan (row blocks x column passes, see spmv_dev_info.tile_*): builds it as upload
 * would with the given parameters (rows per block: multiple of 256 in 256..32768; lmax: longest row kept in the
 * tiles; density: columns per entry up to which a pass is staged; chunk: 2048 entries per pass; balance: 1 = row blocks of about equal entry counts) and replays the kernel's bookkeeping with
 * integer checksums; needs no device.  Both kinds of plan are built and checked: the one with gather passes, then
 * the PACKED one (every pass cut at the window and staged, column words carry the keys: what upload builds for
 * banded matrices, run by the kernel instantiation without gather code).  stats[12] (optional), six per kind in
 * that order: row blocks, passes, entries in tiles, entries in staged passes, rows left to the split-row
 * kernels, widest staged window (columns). */
int spmv_hip_csr_tile_plan_check(int M, int N, const int *row_ptr, const int *col_idx, int value_bytes,
                                 int rows_per_block, int lmax, int density, int chunk, int balance, long long *stats);
/* What upload WOULD decide for a CSR structure under the current tunings (host only, values taken as 1): stats[10] =
 * a tile plan is built (0: the gather kernels keep the matrix), packed plan, scattered geometry (one workgroup per
 * CU), rows per block the kernel is launched for, row blocks, streams (workgroups), passes, rows of the tallest
 * block, work items of the long rows' plan, entries in the ordinary tiles. */
int spmv_hip_csr_tile_auto_plan(int M, int N, const int *row_ptr, const int *col_idx, int value_bytes, long long *stats);
/* Digests of the arrays of a handle's csr_tile plans (the tests' way of saying "the plan built on the device is the plan
 * the host builder makes"): digest[2 k] = elements, digest[2 k + 1] = a hash of the bytes of array k, 32 arrays (entry
 * arrays, pass descriptors, stream tables, remainder, the long rows' plan, the middle tier); digest has 64 entries. */
int spmv_hip_csr_tile_digest(const spmv_csr_dev *m, unsigned long long *digest);
int spmv_hip_hll_tile_digest(const spmv_hll_dev *m, unsigned long long *digest);
/* COO triplets (0-based, any order) -> a CSR handle, built ON THE DEVICE (upload of the
 * triplets, one stable radix sort by (row, column), row pointers and the x-window plan by kernels).  Same
 * matrix as convert_in_csr + spmv_hip_csr_upload_matrix; entries that repeat one (row, column) keep file
 * order here (the reference's quicksort leaves them in its own order), which only reorders equal-column
 * terms of a row's sum. */
int spmv_hip_csr_from_coo(int M, int N, long long nz, const int *I, const int *J, const double *val,
                          spmv_csr_dev **out);
/* the handle's CSR arrays back to the host: row_ptr[M_local + 1] rebased to 0, col[nz], val[nz] (handle's
 * dtype); any pointer may be NULL */
int spmv_hip_csr_download(const spmv_csr_dev *m, int *row_ptr, int *col, void *val);
/* convenience over the kept struct */
int spmv_hip_csr_upload_matrix(const CSRMatrix *csr, spmv_csr_dev **out);
void spmv_hip_csr_free(spmv_csr_dev *m);
int spmv_hip_csr_info(const spmv_csr_dev *m, spmv_dev_info *out);
/* device addresses of the handle's arrays (placement studies): out[8] = row_ptr, col, val, x, y, lcol, lines, ldesc4
 * (0 where the handle has none) */
int spmv_hip_csr_addresses(const spmv_csr_dev *m, unsigned long long *out);
/* Measurement only: one launch of the x-window kernel (fp64, 2048-entry stage) with per-workgroup time stamps behind
 * `warm` ordinary launches: stamps[3 * b + {0, 1, 2}] = start, end (ticks of the constant 100 MHz clock), dispatch id << 8
 * | XCD of block b; stamps has 3 * local_blocks entries. */
int spmv_hip_csr_stamp_blocks(spmv_csr_dev *m, int warm, unsigned long long *stamps);
/* Move one array of the handle (same numbering) to an address of the form (multiple of `align`) + offset; align a
 * power of two >= 256, offset a multiple of 256 below it.  Round 3 found the x-window kernel's time on the headline
 * matrix to depend on where its arrays lie (profiles/r3_placement_*.txt); this is the tool that study used. */
int spmv_hip_csr_relocate(spmv_csr_dev *m, int which, unsigned long long align, unsigned long long offset);
/* the same with memory from HIP's virtual-memory API (hipMemCreate / hipMemAddressReserve / hipMemMap): the virtual
 * address is aligned to `align` exactly as asked, whatever hipMalloc would have chosen */
int spmv_hip_csr_relocate_vmm(spmv_csr_dev *m, int which, unsigned long long align, unsigned long long offset);

/* library-owned vectors: host -> x, run, y -> host (y has M_total entries;
 * this handle writes rows [row0, row0 + M_local) of it) */
int spmv_hip_csr_set_x(spmv_csr_dev *m, const void *x_host);   /* N values of the handle's dtype */
int spmv_hip_csr_run(spmv_csr_dev *m, int variant);             /* asynchronous on the library stream */
int spmv_hip_csr_get_y(spmv_csr_dev *m, void *y_host);          /* syncs, copies M_total values */
void *spmv_hip_csr_x_ptr(spmv_csr_dev *m);                      /* device pointers of those vectors */
void *spmv_hip_csr_y_ptr(spmv_csr_dev *m);

/* caller-owned device vectors (d_y points at element 0 of the FULL y) and
 * caller's stream (NULL = library stream) */
int spmv_hip_csr_run_on(spmv_csr_dev *m, int variant, const void *d_x, void *d_y, void *stream);

/* The reference's timing protocol (main_cuda.cu:159-200): per iteration
 * [zero y if zero_y], record an event, launch, record an event; `warmup`
 * untimed iterations first.  ms_each receives `iters` kernel durations in
 * milliseconds.  The kernels overwrite every row of y, so zero_y only matters
 * for protocol fidelity (the memset sits outside the event pair either way). */
int spmv_hip_csr_time(spmv_csr_dev *m, int variant, int warmup, int iters, int zero_y,
                      float *ms_each);

/* ---- HLL --------------------------------------------------------------- */
/* total_rows = the matrix' M (the last hack may hold fewer than 32 rows). */
int spmv_hip_hll_upload(const HLLMatrix *hll, int total_rows, int N, spmv_hll_dev **out);
/* Host-only self-check of what HLL upload precomputes (slab offsets, workgroup windows, x-window plan); needs no
 * device.  stats[4] (optional): gather windows, x-window windows (0 = no plan), listed lines, widest window's lines. */
int spmv_hip_hll_plan_check(const HLLMatrix *hll, int total_rows, int N, int *stats);
/* One rank's share: hacks [hack0, hack1) = rows [32 hack0, min(32 hack1, total_rows)).  y keeps the
 * full length; the kernels write this handle's rows (HLL is split on hack boundaries). */
int spmv_hip_hll_upload_part(const HLLMatrix *hll, int total_rows, int N, int hack0, int hack1,
                             spmv_hll_dev **out);
/* `iters` launches captured once into a hipGraph and replayed `replays` times (after one warm-up
 * replay): mean time per SpMV with the per-launch host work out of the way -- the number that
 * matters for launch-bound matrices (cant: ~11 us kernel).  Also for HLL below. */
int spmv_hip_csr_time_graph(spmv_csr_dev *m, int variant, int iters, int replays, float *ms_per_iter);
int spmv_hip_hll_time_graph(spmv_hll_dev *m, int variant, int iters, int replays, float *ms_per_iter);

/* build the HLL slab ON THE DEVICE from a resident whole fp64 CSR matrix
 * (per-hack maximum, H-sized offset scan on the host, fill kernel); same slab as
 * convert_to_hll + spmv_hip_hll_upload give when no column repeats inside a row. */
int spmv_hip_hll_from_csr(const spmv_csr_dev *csr, spmv_hll_dev **out);
/* flat slab back to the host: hack_off[hacks + 1] (slot offsets, each hack starts on an even
 * slot), maxnz[hacks], JA / AS [hack_off[hacks]]; any pointer may be NULL */
void *spmv_hip_hll_x_ptr(spmv_hll_dev *m); /* device pointers of the handle's x [N] and y [M] */
void *spmv_hip_hll_y_ptr(spmv_hll_dev *m);
int spmv_hip_hll_download(const spmv_hll_dev *m, long long *hack_off, int *maxnz, int *JA, double *AS);
void spmv_hip_hll_free(spmv_hll_dev *m);
int spmv_hip_hll_info(const spmv_hll_dev *m, spmv_dev_info *out);
int spmv_hip_hll_set_x(spmv_hll_dev *m, const double *x_host);
int spmv_hip_hll_run(spmv_hll_dev *m, int variant);
int spmv_hip_hll_get_y(spmv_hll_dev *m, double *y_host);
int spmv_hip_hll_run_on(spmv_hll_dev *m, int variant, const void *d_x, void *d_y, void *stream);
int spmv_hip_hll_time(spmv_hll_dev *m, int variant, int warmup, int iters, int zero_y,
                      float *ms_each);

/* ---- multi-GPU: one process per GPU, rows split by nnz ------------------ */
/* Contiguous nnz-balanced row split for `parts` GPUs: the reference's greedy
 * (prepare_thread_distribution, src/csr_matrix.c:167-266) with fixed-size
 * output: bounds[0..parts] with bounds[0] = 0, bounds[parts] = M; a part may
 * be empty (bounds[p] == bounds[p+1]).  Pure host code, no device needed. */
int spmv_hip_partition_rows(int M, const int *row_ptr, int parts, int *bounds);
/* The same for HLL with the reference's hack partitioner (prepare_thread_distribution_hll,
 * src/hll_matrix.c:410-540, weight = padded slots): bounds[parts + 1] are HACK indices. */
int spmv_hip_partition_hacks(const HLLMatrix *hll, int parts, int *bounds);

/* RCCL communicator over the GPUs of one node.  Rank 0 creates the id
 * (SPMV_COMM_ID_BYTES opaque bytes) and hands it to the other processes by
 * any means (the Python host uses torch.distributed's store). */
#define SPMV_COMM_ID_BYTES 128
int spmv_hip_comm_get_id(void *id_bytes);
int spmv_hip_comm_init(const void *id_bytes, int rank, int nranks);
int spmv_hip_comm_destroy(void);
/* what RCCL reports for the communicator (ncclCommUserRank / ncclCommCount) */
int spmv_hip_comm_info(int *rank, int *nranks);
/* In-place all-gatherv of y over xGMI: rank r contributes
 * d_y[bounds[r] .. bounds[r+1]) and receives everybody else's rows, as one
 * grouped set of ncclBroadcast calls on `stream` (NULL = library stream).
 * value_bytes is 8 (fp64) or 4 (fp32). */
int spmv_hip_comm_allgatherv(void *d_y, const int *bounds, int value_bytes, void *stream);
/* The scatter half of the padded all-gather on its own: slice p of `d_stage` (at p * widest slice
 * values) -> rows [bounds[p], bounds[p+1]) of d_y, for every p but skip_rank (-1: all). */
int spmv_hip_comm_scatter_staged(const void *d_stage, void *d_y, const int *bounds, int ranks, int skip_rank,
                                 int value_bytes, void *stream);
/* Collective.  Times the two implementations of the all-gatherv on this node -- (0) one
 * ncclBroadcast per owner inside a group, in place; (1) a single ncclAllGather of slices padded to the
 * widest one into a staging buffer + one scatter kernel -- takes the maximum over ranks, checks that (1)
 * reproduces (0) bit for bit, and makes the faster one the mode spmv_hip_comm_allgatherv uses from
 * then on (also settable: spmv_hip_set_tuning("gather_mode", 0 | 1)).  d_y must already hold a
 * gathered vector.  ms_modes[2] (optional) receives the two times, ms_modes[1] < 0 if (1) was rejected. */
int spmv_hip_comm_autotune(void *d_y, const int *bounds, int value_bytes, int iters, int *mode_out,
                           float *ms_modes);
/* One multi-GPU step, timed: SpMV on this rank's rows then the all-gatherv of
 * the library-owned y, both on the library stream, events around each part.
 * ms_kernel / ms_exchange receive `iters` values (either may be NULL). */
int spmv_hip_csr_step_time(spmv_csr_dev *m, int variant, const int *bounds, int warmup, int iters,
                           float *ms_kernel, float *ms_exchange);
/* HLL twin; bounds are ROW bounds (32 x the hack bounds of spmv_hip_partition_hacks, last = M) */
int spmv_hip_hll_step_time(spmv_hll_dev *m, int variant, const int *bounds, int warmup, int iters,
                           float *ms_kernel, float *ms_exchange);

/* SURVEY 8(f) N4 -- iterated SpMV (power iteration as the skeleton): `iters` steps of
 * x <- A x / ||A x||_2 from the handle's current x; y = A x on this rank's rows, all-gatherv(y) when a
 * communicator exists (bounds = the row partition, else NULL), the 2-norm by a fixed-order device
 * reduction that every rank repeats over the gathered y (same bits everywhere, no extra collective).
 * No host synchronisation inside the loop; with one GPU and use_graph != 0 the loop is one hipGraph.
 * Out: x normalised iterate, y last A x, *lambda last ||A x||_2, *ms_total device time of the loop. */
int spmv_hip_csr_power_iterate(spmv_csr_dev *m, int variant, int iters, const int *bounds, int use_graph,
                               double *lambda, float *ms_total);

/* N4, second half -- the halo exchange of an iterated method: a rank needs only the entries of x its rows'
 * columns touch (its own range plus a halo on banded matrices), not the whole all-gathered vector.
 *   spmv_hip_csr_needed_ranges   those entries as at most max_ranges ascending ranges [lo, hi) (ranges[2 * max]),
 *                                from the handle's x-window plan; the whole vector when it has none
 *   spmv_hip_halo_plan           pure host logic, identical on every rank: from every rank's ranges (counts[ranks],
 *                                ranges[ranks * 2 * stride]) and the ownership bounds, the (peer, lo, hi) triples
 *                                this rank sends and receives (send / recv [3 * max_segments])
 *   spmv_hip_comm_halo_setup     collective: all-gathers the ranks' needs, runs the plan, keeps the segments
 *   spmv_hip_comm_halo_exchange  one group of ncclSend / ncclRecv on those segments, in place in d_vec
 *   spmv_hip_comm_halo_info      values sent / received per exchange, peers talked to
 *   spmv_hip_csr_power_iterate_halo   the power iteration with that exchange: partial norms + one all-reduce,
 *                                every rank scales its own range of x, halo segments of x travel */
/* N4, overlap of the exchange with the product.  A rank's x-window blocks are split into INTERIOR blocks (every
 * x line they list lies in the rank's own range [row0, row0 + M_local) of x: they can run before the halo has
 * arrived) and BOUNDARY blocks (the rest, plus rows outside the plan).
 *   spmv_hip_csr_split_interior   computes the split from the handle's plan (spmv_hip_comm_halo_setup calls it);
 *                                 counts[4] (optional): interior blocks, boundary blocks, entries in interior
 *                                 blocks, entries elsewhere.  A handle without an x-window plan has no interior.
 *   spmv_hip_csr_run_part         part 0 = interior blocks only, part 1 = everything else; 0 then 1 = one
 *                                 spmv_hip_csr_run_on(STREAM), bit for bit (d_x / d_y / stream NULL = the handle's)
 *   spmv_hip_csr_power_iterate_halo   uses it when a communicator exists: the halo exchange runs on a second
 *                                 stream beside the interior blocks, the boundary blocks wait for its event
 *                                 ("halo_overlap" tuning knob 1 | 0) */
/* N4, the second skeleton: `iters` steps of plain conjugate gradients for a symmetric positive definite A, from
 * x0 = 0: p is the handle's x (gathered / halo-exchanged every step exactly as in the power iteration: bounds = the
 * row partition when a communicator exists; use_halo != 0 after spmv_hip_comm_halo_setup), q = A p its y, every rank
 * keeps its rows of x and r.  Dot products are fixed-order device reductions; across ranks the partial sums are
 * all-gathered and added in rank order by every rank (same bits everywhere).  No host synchronisation in the loop.
 * b_host: the right-hand side, M_total values of the handle's dtype (a rank reads its own rows).  Out: x_host
 * (optional) the iterate after `iters` steps, M_total values; rr_hist (optional) [iters + 1] the squared residual
 * norm r.r before the first step and after every step; *ms_total device time of the loop. */
int spmv_hip_csr_cg(spmv_csr_dev *m, int variant, int iters, const int *bounds, int use_halo, const void *b_host,
                    void *x_host, double *rr_hist, float *ms_total);
int spmv_hip_csr_split_interior(spmv_csr_dev *m, long long *counts);
/* N4 overlap below block granularity (round 3).  On a KKT-coupled cut every block also lists lines of the coupling block,
 * which another rank owns: no interior BLOCKS -- but 13 of a row's 28 entries have their column in the rank's own range.
 *   spmv_hip_csr_split_columns   splits the handle's entries by column into two sub-handles over the same rows: [col_lo,
 *                                col_hi) = the rank's own range of x; counts[2] (optional): entries inside / outside
 *                                (spmv_hip_comm_halo_setup calls it with the rank's bounds; "halo_split" 0: not)
 *   spmv_hip_csr_run_split       part 0: y = A_own x (reads nothing of x outside the range: it can run while the halo
 *                                travels); part 1: y += A_halo x.  0 then 1 = the handle's product up to the order in
 *                                which a row's two partial sums are added (a fixed order: reproducible)
 * spmv_hip_csr_power_iterate_halo uses the column split where halo setup made one. */
int spmv_hip_csr_split_columns(spmv_csr_dev *m, int col_lo, int col_hi, long long *counts);
int spmv_hip_csr_run_split(spmv_csr_dev *m, int part, const void *d_x, void *d_y, void *stream);
int spmv_hip_csr_run_part(spmv_csr_dev *m, int part, const void *d_x, void *d_y, void *stream);
int spmv_hip_csr_needed_ranges(const spmv_csr_dev *m, int max_ranges, int *ranges, int *count);
int spmv_hip_halo_plan(int ranks, int rank, const int *bounds, const int *counts, const int *ranges, int stride,
                       int max_segments, int *send, int *nsend, int *recv, int *nrecv);
int spmv_hip_comm_halo_setup(spmv_csr_dev *m, const int *bounds);
int spmv_hip_comm_halo_exchange(void *d_vec, int value_bytes, void *stream);
int spmv_hip_comm_halo_info(long long *send_values, long long *recv_values, int *peers);
int spmv_hip_csr_power_iterate_halo(spmv_csr_dev *m, int variant, int iters, double *lambda, float *ms_total);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_AMD_SPMV_HIP_H */
