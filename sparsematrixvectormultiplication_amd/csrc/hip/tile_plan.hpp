// tile_plan.hpp -- upload-time builder of the csr_tile format (host code, no device needed: the
// plan check of tests/ runs it on the CPU).  See tile_kernels.hpp for what the kernel does with it.
//
//   rows        cut into blocks of consecutive rows: at most rows_per_block of them (the accumulators' room) and
//               about equally many entries each (`balance`: a block closes once it holds the mean number of
//               entries of a full-height block) -- workgroups that take equally long stay in step on their
//               way up the columns, which is what keeps the band of x they gather from inside L2
//   rows        are given as (first entry, length) pairs, so that a plan can also be built over a COMPACTED set of
//               rows (the long rows of a matrix, see spmv_csr.hip); pos_bits = bits of a row's length, the rest of 32
//               bits numbers the rows of a block (17 / 15 for ordinary plans, 21 / 11 for long-row plans)
//   long rows   (more than lmax < 2^pos_bits entries) are left out: `split` marks them
//   per block   its entries ordered by column are cut greedily into passes: a pass takes entries while
//               it has fewer than kTileChunkMax and -- as long as that keeps it dense enough to be worth
//               staging -- while its column range fits the LDS window; inside a pass entries are
//               ordered by (row, column) and the first entry of every row carries the head flag
//   staging     a pass is staged (its slice of x copied to LDS) when its column range fits the window
//               and holds at least one entry per `density` columns
//   pack        EVERY pass is cut at the window and staged, however few entries that leaves it (the sparse tails
//               of a band: a handful of passes per block with a few entries each), and its entries are packed:
//               the kernel for such a plan has no gather and no key array, and sends out the same loads in every pass
#pragma once
#include <hip/hip_runtime.h>  // int4

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <exception>
#include <thread>
#include <vector>

#include "tile_kernels.hpp"

namespace spmv {

template <typename T>
struct TilePlan {
    int rows_per_block = 0;
    int chunk = 0;                   // entries per pass at most: 2048 or 4096 (the kernel's template parameter)
    int win_cols = 0;                // widest window a pass may stage
    int num_blocks = 0;
    int max_win = 0;                 // widest staged window (columns, multiple of 4)
    long long entries = 0;           // entries held by the tiles (without padding)
    long long staged_entries = 0;    // ... of which in staged passes
    long long staged_cols = 0;       // sum of staged windows (x values copied to LDS per SpMV)
    std::vector<int> block_row;      // [num_blocks + 1] first row of every block (blocks hold <= rows_per_block rows)
    std::vector<int> block_pass;     // [num_blocks + 1]
    std::vector<int4> pass_desc;     // {first entry, entries, window base, window columns (0 = gather)}
    std::vector<int> tcol;           // [padded entries + kTileChunkMax]
    std::vector<unsigned short> tkey;  // (a packed plan: padding only)
    std::vector<T> tval;
    std::vector<unsigned char> split;  // [M] 1: row is not in the tiles (longer than lmax)
    // remainder (packed plans built with min_pass > 0): the entries of windows too sparse to be worth a pass -- the far
    // tails of a band, stray entries -- by (row, column); added to y behind the tiles by tile_remainder
    std::vector<int> rem_row, rem_col;  // row in the plan's row space, column
    std::vector<T> rem_val;
    // streams (tile_make_streams): what ONE workgroup walks -- the passes of its blocks back to back, so that the
    // loads of a block's first passes go out while the block before it is still being summed
    int num_streams = 0;
    std::vector<int4> spass;           // pass descriptors in stream order, kTilePassLast on a block's last pass
    std::vector<int> stream_pass;      // [num_streams + 1] first pass (in spass) of every stream
    std::vector<int> stream_block;     // [num_streams + 1] first block (in sblock_rows) of every stream
    std::vector<int2> sblock_rows;     // {first row, rows} of the blocks in stream order
};

namespace tile_detail {

// body(0) .. body(n - 1) on n threads.  Nothing a body throws (std::bad_alloc at these sizes: a plan holds two or
// three host copies of a 100-260 M-entry matrix) may leave its thread -- that would be std::terminate for the whole
// process, past the C-ABI's guarded() -- so every body runs inside a catch, every started thread is joined whatever
// happens (also when starting a later one fails), and the first exception is rethrown on the caller's thread.
template <typename F>
void run_threads(int n, F body) {
    std::vector<std::exception_ptr> errors((size_t)std::max(n, 0));
    struct Joiner {
        std::vector<std::thread> pool;
        ~Joiner() {
            for (auto &th : pool)
                if (th.joinable()) th.join();
        }
    } joiner;
    joiner.pool.reserve((size_t)std::max(n, 0));
    std::exception_ptr start_error;
    for (int th = 0; th < n; ++th) {
        try {
            joiner.pool.emplace_back([&errors, &body, th] {
                try {
                    body(th);
                } catch (...) {
                    errors[(size_t)th] = std::current_exception();
                }
            });
        } catch (...) {  // the thread could not be started: its share runs here, after the others have been joined
            start_error = std::current_exception();
            break;
        }
    }
    const int started = (int)joiner.pool.size();
    for (auto &th : joiner.pool) th.join();
    for (int th = started; th < n; ++th) body(th);  // (throws straight to the caller)
    (void)start_error;
    for (auto &e : errors)
        if (e) std::rethrow_exception(e);
}

template <typename T>
struct Part {  // what one builder thread produced for its range of blocks
    std::vector<int> passes_per_block;
    std::vector<int4> pass_desc;  // first entry relative to the part
    std::vector<int> tcol;
    std::vector<unsigned short> tkey;
    std::vector<T> tval;
    long long entries = 0, staged_entries = 0, staged_cols = 0;
    int max_win = 0;
    bool failed = false;
    std::vector<int> rem_row, rem_col;
    std::vector<T> rem_val;
};

// std::sort of 64-bit keys whose top 32 bits are < key_top, with several threads: bucket by the leading bits
// (two passes), sort the buckets independently.  Used for the few, very large blocks of a long-row plan.
inline void sort_keys(std::vector<uint64_t> &keys, uint32_t key_top, int threads) {
    const size_t n = keys.size();
    if (threads <= 1 || n < (size_t)1 << 22) {
        std::sort(keys.begin(), keys.end());
        return;
    }
    constexpr int kBuckets = 1024;
    int shift = 0;
    while (((uint64_t)key_top >> shift) >= (uint64_t)kBuckets) ++shift;  // bucket = (key >> 32) >> shift < kBuckets
    std::vector<size_t> start((size_t)kBuckets + 1, 0);
    for (uint64_t k : keys) ++start[(size_t)((k >> 32) >> shift) + 1];
    for (int b = 0; b < kBuckets; ++b) start[(size_t)b + 1] += start[(size_t)b];
    std::vector<uint64_t> tmp(n);
    {
        std::vector<size_t> at(start.begin(), start.end() - 1);
        for (uint64_t k : keys) tmp[at[(size_t)((k >> 32) >> shift)]++] = k;
    }
    run_threads(threads, [&](int th) {
        for (int b = th; b < kBuckets; b += threads) std::sort(tmp.begin() + (long)start[(size_t)b], tmp.begin() + (long)start[(size_t)b + 1]);
    });
    keys.swap(tmp);
}

template <typename T>
void build_range(int b0, int b1, const int *block_row, const int *row_begin, const int *row_len, const int *col,
                 const T *val, int lmax, int pos_bits, int chunk, int win_cols, int density, int inner_threads,
                 uint32_t col_top, bool pack, int min_pass, Part<T> &out) {
    // column << 32 | local row << pos_bits | position inside the row (rows <= lmax < 2^pos_bits, local rows < 2^(32 - pos_bits))
    std::vector<uint64_t> keyed;
    const uint32_t pos_mask = (1u << pos_bits) - 1;
    std::vector<uint64_t> pass;
    struct Stray {
        int row, col;
        T val;
    };
    std::vector<Stray> rem;
    for (int b = b0; b < b1; ++b) {
        rem.clear();
        const int r0 = block_row[b], r1 = block_row[b + 1];
        keyed.clear();
        for (int r = r0; r < r1; ++r) {
            const int len = row_len[r];
            if (len > lmax) continue;
            for (int k = 0; k < len; ++k)
                keyed.push_back(((uint64_t)(unsigned)col[row_begin[r] + k] << 32) | ((uint64_t)(r - r0) << pos_bits) | (uint64_t)k);
        }
        const size_t n = keyed.size();
        // one pass in CSR order when everything fits: no sort needed to cut, (row, column) is the input order
        bool sorted_by_col = false;
        bool one_pass = n <= (size_t)chunk;
        if (one_pass && pack && n > 0) {  // ... and, in a plan that stages every pass, fits one window
            uint32_t cmin = 0xffffffffu, cmax = 0;
            for (uint64_t k : keyed) {
                cmin = std::min(cmin, (uint32_t)(k >> 32));
                cmax = std::max(cmax, (uint32_t)(k >> 32));
            }
            one_pass = (long long)cmax - (long long)(cmin & ~3u) < win_cols;
        }
        if (!one_pass) {
            sort_keys(keyed, col_top, inner_threads);
            sorted_by_col = true;
        }
        int passes = 0;
        size_t i = 0;
        while (i < n) {
            size_t j;
            if (!sorted_by_col) {
                j = n;
            } else {
                // window-limited cut first: how many entries fall into [base, base + win_cols)?
                const long long base = (long long)(keyed[i] >> 32) & ~3LL;
                size_t w = i;
                const size_t cap = std::min(n, i + (size_t)chunk);
                while (w < cap && (long long)(keyed[w] >> 32) < base + win_cols) ++w;
                const long long span = (long long)(keyed[w - 1] >> 32) - base + 1;
                if (pack && (int)(w - i) < min_pass && (long long)(w - i) * 16 < span) {
                    // a window with a handful of entries far apart (the far tail of a band, stray entries): not worth
                    // a pass -- a slice of x, two barriers -- of its own: to the remainder
                    for (size_t k = i; k < w; ++k) {
                        const int lrow = (int)((uint32_t)keyed[k] >> pos_bits), pos = (int)((uint32_t)keyed[k] & pos_mask);
                        const int e = row_begin[r0 + lrow] + pos;
                        rem.push_back(Stray{r0 + lrow, col[e], val[e]});
                    }
                    i = w;
                    continue;
                }
                if (pack || (long long)(w - i) * density >= span) j = w;  // dense enough (pack: always): a staged pass
                else j = cap;                                             // sparse here: a full gather pass
            }
            pass.assign(keyed.begin() + (long)i, keyed.begin() + (long)j);
            int cmin = 0x7fffffff, cmax = 0;
            for (uint64_t k : pass) {
                const int c = (int)(k >> 32);
                cmin = std::min(cmin, c);
                cmax = std::max(cmax, c);
            }
            // order inside the pass: (row, position in the row) = (row, column), duplicates in file order
            if (sorted_by_col)
                std::sort(pass.begin(), pass.end(),
                          [](uint64_t a, uint64_t b) { return (uint32_t)a < (uint32_t)b; });
            const int count = (int)pass.size();
            // the staged slice: whole 16-byte pieces from a 16-byte aligned start, never past the piece that holds
            // x[N - 1] (a caller's x need not have anything allocated behind it)
            constexpr int kPer = 16 / (int)sizeof(T);
            const int wbase = cmin & ~3;
            int wlen = ((cmax - wbase + 1) + 3) & ~3;
            wlen = std::min(wlen, ((int)col_top - wbase + kPer - 1) / kPer * kPer);
            const bool staged = wlen <= win_cols && wlen <= (int)kTilePackColMask + 1 && (pack || (long long)count * density >= wlen);
            if (pack && !staged) {  // (cannot happen: every cut above is window-limited)
                out.failed = true;
                return;
            }
            const int e_first = (int)out.tcol.size();
            int prev_row = -1;
            for (uint64_t k : pass) {
                const int lrow = (int)((uint32_t)k >> pos_bits), pos = (int)((uint32_t)k & pos_mask);
                const int e = row_begin[r0 + lrow] + pos;
                const bool head = lrow != prev_row;
                if (staged && pack)  // packed: head << 31 | local row << 14 | column - window base (the key array is not read)
                    out.tcol.push_back((int)(((unsigned)head << 31) | ((unsigned)lrow << kTilePackShift) | (unsigned)(col[e] - wbase)));
                else
                    out.tcol.push_back(col[e]);
                out.tval.push_back(val[e]);
                if (!pack) out.tkey.push_back((unsigned short)(lrow | (head ? kTileHead : 0)));
                prev_row = lrow;
            }
            while (out.tcol.size() & 3) {  // the next pass starts on a multiple of 4
                out.tcol.push_back(wbase);
                out.tval.push_back(T(0));
                if (!pack) out.tkey.push_back(0);
            }
            out.pass_desc.push_back(int4{e_first, count, wbase, staged ? (wlen | (pack ? kTilePassPacked : 0)) : 0});
            out.entries += count;
            if (staged) {
                out.staged_entries += count;
                out.staged_cols += wlen;
                out.max_win = std::max(out.max_win, wlen);
            }
            ++passes;
            i = j;
        }
        if (!rem.empty()) {  // by (row, column): the remainder kernel adds a row's entries in that order
            std::stable_sort(rem.begin(), rem.end(), [](const Stray &a, const Stray &b) { return a.row != b.row ? a.row < b.row : a.col < b.col; });
            for (const Stray &st : rem) {
                out.rem_row.push_back(st.row);
                out.rem_col.push_back(st.col);
                out.rem_val.push_back(st.val);
            }
        }
        if (passes == 0) {  // a block without entries (empty rows, or only rows beyond the limit): one pass of none, so
            out.pass_desc.push_back(int4{(int)out.tcol.size(), 0, 0, pack ? ((16 / (int)sizeof(T)) | kTilePassPacked) : 0});  // that whoever
            passes = 1;                                                                           // walks it writes its zeros
        }
        out.passes_per_block.push_back(passes);
    }
}

}  // namespace tile_detail

// Row blocks: consecutive rows, at most rows_per_block of them, closed once they hold `target` entries (rows longer than
// lmax hold none: they are not in the tiles).  Returns the first row of every block and M behind the last.
inline std::vector<int> tile_cut_rows(int M, const int *row_len, int lmax, int rows_per_block, long long target) {
    std::vector<int> block_row(1, 0);
    long long held = 0;
    for (int r = 0; r < M; ++r) {
        const int len = row_len[r] > lmax ? 0 : row_len[r];
        if (r - block_row.back() == rows_per_block || (held >= target && r > block_row.back())) {
            block_row.push_back(r);
            held = 0;
        }
        held += len;
    }
    if (M > 0) block_row.push_back(M);
    return block_row;
}

// false: the tiles would not hold the matrix (entry offsets beyond 32 bits).  target_entries: entries at which a block
// closes (0: balance ? the mean of a full-height block : never -- blocks of rows_per_block rows)
template <typename T>
bool tile_build(int M, int N, const int *row_begin, const int *row_len, const int *col, const T *val, int rows_per_block,
                int lmax, int density, int chunk, bool balance, int pos_bits, TilePlan<T> &plan, bool pack = true,
                long long target_entries = 0, int min_pass = 0) {
    // the window a pass may stage: kTileTrips trips of the workgroup = 32 KiB, which with a 2048-entry chunk and
    // 2048 fp64 accumulators lets two workgroups share a CU's LDS, and with 8192 of them still fits one
    const int win_cols = kTileTrips * kTileTripBytes / (int)sizeof(T);
    plan = TilePlan<T>();
    plan.rows_per_block = rows_per_block;
    plan.chunk = chunk;
    plan.win_cols = win_cols;
    plan.split.assign((size_t)M, 0);
    long long in_tiles = 0;
    for (int r = 0; r < M; ++r) {
        plan.split[r] = row_len[r] > lmax;
        if (!plan.split[r]) in_tiles += row_len[r];
    }
    // block boundaries: the row cap, and (balance) the mean entry count of a full-height block
    const long long full_blocks = std::max(1, (M + rows_per_block - 1) / rows_per_block);
    const long long target = target_entries > 0 ? std::max<long long>(chunk, target_entries)
                             : balance          ? std::max<long long>(chunk, (in_tiles + full_blocks - 1) / full_blocks)
                                                : (1LL << 62);
    plan.block_row = tile_cut_rows(M, row_len, lmax, rows_per_block, target);
    plan.num_blocks = (int)plan.block_row.size() - 1;
    const int B = plan.num_blocks;
    int threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    const int hw_threads = threads;
    threads = std::max(1, std::min(threads, B >= 64 ? B / 4 : B));
    const int inner_threads = std::max(1, hw_threads / threads);  // few, large blocks: threads inside the sort instead
    std::vector<tile_detail::Part<T>> parts((size_t)threads);
    // blocks are dealt out in contiguous ranges balanced by entries
    std::vector<int> cut((size_t)threads + 1, B);
    cut[0] = 0;
    {
        long long total = 0, run = 0;
        for (int r = 0; r < M; ++r) total += row_len[r];
        int th = 1, r = 0;
        for (int b = 0; b < B && th < threads; ++b) {
            for (; r < plan.block_row[(size_t)b + 1]; ++r) run += row_len[r];
            if (run * threads >= total * th) cut[th++] = b + 1;
        }
    }
    tile_detail::run_threads(threads, [&](int th) {
        tile_detail::build_range<T>(cut[th], cut[th + 1], plan.block_row.data(), row_begin, row_len, col, val, lmax,
                                    pos_bits, chunk, win_cols, density, inner_threads, (uint32_t)std::max(N, 1), pack, pack ? min_pass : 0, parts[th]);
    });
    size_t total_entries = 0, total_passes = 0;
    for (const auto &p : parts)
        if (p.failed) return false;
    for (const auto &p : parts) {
        total_entries += p.tcol.size();
        total_passes += p.pass_desc.size();
    }
    if (total_entries + kTileChunkMax >= 0x7fffffffull) return false;
    plan.tcol.reserve(total_entries + kTileChunkMax);
    if (!pack) plan.tkey.reserve(total_entries + kTileChunkMax);
    plan.tval.reserve(total_entries + kTileChunkMax);
    plan.pass_desc.reserve(total_passes);
    plan.block_pass.assign(1, 0);
    for (auto &p : parts) {
        const int base = (int)plan.tcol.size();
        for (int4 d : p.pass_desc) {
            d.x += base;
            plan.pass_desc.push_back(d);
        }
        for (int n : p.passes_per_block) plan.block_pass.push_back(plan.block_pass.back() + n);
        plan.tcol.insert(plan.tcol.end(), p.tcol.begin(), p.tcol.end());
        plan.tkey.insert(plan.tkey.end(), p.tkey.begin(), p.tkey.end());
        plan.tval.insert(plan.tval.end(), p.tval.begin(), p.tval.end());
        plan.entries += p.entries;
        plan.staged_entries += p.staged_entries;
        plan.staged_cols += p.staged_cols;
        plan.max_win = std::max(plan.max_win, p.max_win);
        plan.rem_row.insert(plan.rem_row.end(), p.rem_row.begin(), p.rem_row.end());
        plan.rem_col.insert(plan.rem_col.end(), p.rem_col.begin(), p.rem_col.end());
        plan.rem_val.insert(plan.rem_val.end(), p.rem_val.begin(), p.rem_val.end());
        p = tile_detail::Part<T>();  // a part's arrays are released as soon as they are merged: one copy fewer at the peak
    }
    // the kernel loads whole units past a pass's end
    plan.tcol.insert(plan.tcol.end(), (size_t)kTileChunkMax, 0);
    plan.tkey.insert(plan.tkey.end(), (size_t)kTileChunkMax, 0);
    plan.tval.insert(plan.tval.end(), (size_t)kTileChunkMax, T(0));
    return true;
}

// Streams over a built plan: the chip holds `places` workgroups at once; with more blocks than that, workgroup s of XCD x
// (blockIdx = x + 8 s) walks blocks s, s + W, s + 2 W, ... of that XCD's contiguous eighth of the blocks (W = places / 8
// workgroups per XCD), i.e. in every 'round' the workgroups of an XCD still sit on neighbouring blocks, which keeps the
// band of x they gather from together inside that XCD's L2 (tile_kernels.hpp).  Only the descriptors are re-ordered; the
// entries stay where they are.
template <typename T>
void tile_make_streams(TilePlan<T> &plan, int places) {
    const int B = plan.num_blocks;
    places = std::max(8, places / 8 * 8);
    std::vector<std::vector<int>> streams;
    if (B <= places) {
        // (workgroup ids go round-robin over the XCDs: stream x + 8 j = block j of XCD x's eighth; with a block count
        // that is no multiple of 8 a few streams stay empty)
        const int per_xcd = (B + 7) / 8;
        streams.assign((size_t)(per_xcd * 8), {});
        for (int b = 0; b < B; ++b) streams[(size_t)((b / per_xcd) + 8 * (b % per_xcd))].push_back(b);
    } else {
        const int per_xcd = (B + 7) / 8, W = places / 8;
        streams.assign((size_t)places, {});
        for (int b = 0; b < B; ++b) {
            const int x = b / per_xcd, local = b % per_xcd;
            streams[(size_t)(x + 8 * (local % W))].push_back(b);
        }
    }
    plan.num_streams = (int)streams.size();
    plan.spass.clear();
    plan.sblock_rows.clear();
    plan.stream_pass.assign(1, 0);
    plan.stream_block.assign(1, 0);
    for (const auto &st : streams) {
        for (int b : st) {
            for (int p = plan.block_pass[(size_t)b]; p < plan.block_pass[(size_t)b + 1]; ++p) {
                int4 d = plan.pass_desc[(size_t)p];
                if (p + 1 == plan.block_pass[(size_t)b + 1]) d.w |= kTilePassLast;
                plan.spass.push_back(d);
            }
            plan.sblock_rows.push_back(int2{plan.block_row[(size_t)b], plan.block_row[(size_t)b + 1] - plan.block_row[(size_t)b]});
        }
        plan.stream_pass.push_back((int)plan.spass.size());
        plan.stream_block.push_back((int)plan.sblock_rows.size());
    }
}

}  // namespace spmv
