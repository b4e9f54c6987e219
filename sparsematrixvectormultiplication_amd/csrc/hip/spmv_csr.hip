// spmv_csr.hip -- CSR side of the C-ABI: upload-time preprocessing (workgroup blocks, the
// x-window plan, split long rows), kernel launchers, timing.  Replaces the device code of the
// reference's per-matrix CSR section (/root/reference/main_cuda.cu:135-145, :149-166, :212-238,
// :285-317, :682-685).
#include "spmv_internal.hpp"

#include "plan_kernels.hpp"
#include <chrono>

#include "tile_plan.hpp"
#include "tile_plan_device.hpp"

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
        int r1 = r, maxlen = 0;
        while (r1 < M && r1 - r < rows_cap && rp[r1 + 1] - base <= cap && !is_long(r1) &&
               !skew_cut(r1 - r, maxlen, rp[r1 + 1] - rp[r1])) {
            maxlen = std::max(maxlen, rp[r1 + 1] - rp[r1]);
            ++r1;
        }
        desc.push_back(int4{r, n0, r1 - r, rp[r1]});
        r = r1;
    }
}


// Pieces of the rows marked in `split` for a tiled handle: cut where the column crosses a stripe of
// `stripe_cols` columns (and every kLongPiece entries), slots numbered row by row (csr_long_finish adds a
// row's slots in that order), but the pieces themselves ordered stripe by stripe: the launch then sweeps
// x one L2-sized stripe at a time instead of gathering from all of it at once.
void build_striped_pieces(int M, const int *rp, const int *col, const std::vector<unsigned char> &split,
                          int stripe_cols, std::vector<int4> &pieces, std::vector<int4> &long_rows) {
    struct Tagged {
        int stripe;
        int4 d;
    };
    std::vector<Tagged> tmp;
    long_rows.clear();
    for (int r = 0; r < M; ++r) {
        if (!split[r]) continue;
        const int first_slot = (int)tmp.size();
        int e = rp[r];
        const int end = rp[r + 1];
        while (e < end) {
            const int s = col[e] / stripe_cols;
            int f = e + 1;
            while (f < end && f - e < kLongPiece && col[f] / stripe_cols == s) ++f;
            tmp.push_back(Tagged{s, int4{r, e, f, (int)tmp.size()}});
            e = f;
        }
        long_rows.push_back(int4{r, first_slot, (int)tmp.size() - first_slot, 0});
    }
    std::stable_sort(tmp.begin(), tmp.end(), [](const Tagged &a, const Tagged &b) { return a.stripe < b.stripe; });
    pieces.resize(tmp.size());
    for (size_t k = 0; k < tmp.size(); ++k) pieces[k] = tmp[k].d;
}

// Long-row tile plan.  In: `split` marks the rows the ordinary tiles left out.  Those of them shorter than 2^21
// entries are compacted (rows[v] = the v-th such row) and given a tile plan of their own with pos_bits = 21;
// `split` keeps only what this plan does not take.  A block's passes are cut into work items of about equal pass
// counts, a few thousand in all.  false: nothing to do (fewer than 2^20 entries in such rows).
// build(rows, begin, len, pack, plan): tile_build over the compacted rows with pos_bits = 21 -- on the host or on the device
// (len_lo, len_hi]: which of the marked rows this plan takes; kPosBits / kRowsPerBlock: 21 / 2048 for the long rows
// (up to 2^21 - 1 entries each), 17 / as tall as the LDS takes for the middle tier of a scattered matrix
template <typename T, typename Build>
bool build_long_tiles(int M, int N, const int *rp, const int *row_len, int chunk,
                      std::vector<unsigned char> &split, TilePlan<T> &plan, std::vector<int> &rows,
                      std::vector<int4> &work, std::vector<int> &item_first, bool &packed, Build build,
                      int len_lo = 0, int len_hi = (1 << 21) - 1, int kPosBits = 21, int kRowsPerBlock = 2048, int items = 0) {
    if (items <= 0) items = g_tile_items;
    rows.clear();
    long long entries = 0;
    for (int r = 0; r < M; ++r)
        if (split[(size_t)r] && row_len[r] > len_lo && row_len[r] <= len_hi) {
            rows.push_back(r);
            entries += row_len[r];
        }
    if (rows.empty() || (entries < (1LL << 20) && g_tile_long < 2)) return false;  // (2: whatever the size, tests)
    std::vector<int> vbegin(rows.size()), vlen(rows.size());
    for (size_t v = 0; v < rows.size(); ++v) {
        vbegin[v] = rp[rows[v]];
        vlen[v] = row_len[rows[v]];
    }
    // packed (every pass staged) unless that leaves passes of a few entries each
    packed = g_tile_pack != 0;
    if (!build((int)rows.size(), vbegin.data(), vlen.data(), kRowsPerBlock, len_hi, kPosBits, packed, plan)) return false;
    if (packed && plan.entries < (long long)plan.pass_desc.size() * (chunk / 8)) {
        packed = false;
        if (!build((int)rows.size(), vbegin.data(), vlen.data(), kRowsPerBlock, len_hi, kPosBits, false, plan)) return false;
    }
    for (int r : rows) split[(size_t)r] = 0;
    // work items: ~tile_items (1008: two rounds of the 512 places) of them over all blocks, at least 4 passes each
    const long long passes = (long long)plan.pass_desc.size();
    const int per_item = (int)std::max<long long>(4, (passes + items - 1) / items);
    work.clear();
    item_first.assign(1, 0);
    for (int b = 0; b < plan.num_blocks; ++b) {
        for (int p = plan.block_pass[(size_t)b]; p < plan.block_pass[(size_t)b + 1]; p += per_item)
            work.push_back(int4{b, p, std::min(p + per_item, plan.block_pass[(size_t)b + 1]), (int)work.size()});
        item_first.push_back((int)work.size());
    }
    return true;
}

// Everything the tile plans of one handle consist of on the host, between planning and upload.
template <typename T>
struct TileBuild {
    TilePlan<T> tiles, ltiles;
    bool have_tiles = false, have_long_tiles = false, scattered = false;
    bool packed = false;     // the ordinary tiles are a packed plan: every pass staged (never for scattered matrices)
    bool lt_packed = false;  // ... the long rows' tiles
    std::vector<int4> tile_pieces, tile_long, lt_work;
    std::vector<int> lt_rows, lt_item_first;
    // the middle tier of a scattered matrix (rows of g_tile_mid_lo < entries <= tile_lmax): the same kind of plan
    TilePlan<T> mtiles;
    bool have_mid_tiles = false, mt_packed = false;
    std::vector<int4> mt_work;
    std::vector<int> mt_rows, mt_item_first;
    std::shared_ptr<TileDevArrays<T>> mtiles_dev;
    // plans built on the device (tile_plan_device.hpp) keep their entry arrays there: the handle adopts them
    std::shared_ptr<TileDevArrays<T>> tiles_dev, ltiles_dev;
    std::string dev_error;  // a device build that failed with a HIP error (the host builder took over)
};

// The tile plans for rows given as (first entry, length) pairs over host arrays col / val -- a CSR matrix's rows, or
// the rows of an HLL slab (spmv_hll.hip).  rp: the CSR row pointer when the rows ARE a CSR matrix's (then rows that
// neither plan takes go to stripe-cut split-row pieces), nullptr otherwise (then such rows veto the plan).
//
// Rows per block (auto): two regimes, told apart on a sample of the rows.
//  * banded (most entries in passes that can be staged): 32 KiB of accumulators (4096 fp64 / 8192 fp32 rows), so
//    that two workgroups with their 32 KiB x slices share a CU (road-like, wide band: measured best of 2048 / 4096 /
//    8192);
//  * scattered (gather passes): what matters is that a pass spans little of x (the band all blocks gather from
//    together must fit L2), that the blocks run in few rounds (a new round starts again at column 0) and that not
//    too many of them are on the way at once: the tallest blocks (up to 16384 rows) that still leave ~2 per CU, ONE
//    workgroup per CU.  Power-law matrix (fp32, 2^24 rows): 2.90 ms at 8192 rows, 1.97 ms at 16384 with one
//    workgroup per CU, 2.39 ms with two.
constexpr int kTileGatherMinCols = 800000;  // (auto) columns from which a tile plan with gather passes is built

// din (optional): the same rows on the device (row_begin / row_len / col / val there): the plans are then built by
// tile_build_device -- byte for byte what tile_build makes of the host arrays, which remain the fallback when a
// device build fails with a HIP error (hcol / hval may be NULL when there is a din: then such a failure vetoes the plan).
template <typename T>
void tile_plan_all(int Ml, int N, const int *row_begin, const int *row_len, const int *rp, long long nz, const int *hcol,
                   const T *hval, TileBuild<T> &tb, const TileDevInput<T> *din = nullptr) {
    const int chunk = 2048;
    UploadTrace trace("tile_plan_all");
    bool dev_ok = din != nullptr;
    std::vector<void *> dev_tmp;  // device copies of compacted row lists (long rows' plan), freed on return
    struct FreeAll {
        std::vector<void *> &v;
        ~FreeAll() {
            for (void *p : v) (void)hipFree(p);
        }
    } free_all{dev_tmp};
    // one tile_build, wherever: rows [s0, s0 + rows) of the handle's row arrays (begin_h / len_h point at row s0), or --
    // own_rows -- a row list of its own that exists on the host only
    auto build_at = [&](int rows, const int *begin_h, const int *len_h, const int *begin_d, const int *len_d, int rpb, int lmax,
                        int pos_bits, bool pack, long long target, int min_pass, TilePlan<T> &plan,
                        std::shared_ptr<TileDevArrays<T>> &arrays) -> bool {
        arrays.reset();
        if (dev_ok) {
            TileDevInput<T> in = *din;
            in.row_begin = begin_d;
            in.row_len = len_d;
            const int rc = tile_build_device<T>(rows, N, in, len_h, rpb, lmax, g_tile_density, chunk, g_tile_balance != 0 || pos_bits != 17,
                                                pos_bits, plan, arrays, pack, target, min_pass, tb.dev_error);
            if (rc >= 0) return rc == 1;
            dev_ok = false;  // a HIP error: everything from here on on the host
            arrays.reset();
        }
        if (!hcol || !hval) return false;
        return tile_build<T>(rows, N, begin_h, len_h, hcol, hval, rpb, lmax, g_tile_density, chunk, g_tile_balance != 0 || pos_bits != 17,
                             pos_bits, plan, pack, target, min_pass);
    };
    std::shared_ptr<TileDevArrays<T>> probe_arrays;
    int lmax_eff = g_tile_lmax;  // longest row of the ordinary tiles (g_tile_mid_lo once a middle tier is decided on)
    auto build_rows = [&](int rows, int s0, int rpb, bool pack, long long target, int min_pass, TilePlan<T> &plan,
                          std::shared_ptr<TileDevArrays<T>> &arrays) {
        return build_at(rows, row_begin + s0, row_len + s0, din ? din->row_begin + s0 : nullptr, din ? din->row_len + s0 : nullptr, rpb,
                        lmax_eff, 17, pack, target, min_pass, plan, arrays);
    };
    int rb = g_tile_rows;
    // Which kernel: a banded matrix is built PACKED -- every pass cut at the window and staged, the sparse tails too
    // (tile_plan.hpp) -- for the kernel instantiation without gather code, unless that leaves passes of a few entries
    // each (entries far from the band: one window, i.e. one pass, per stray entry); anything else keeps gather passes.
    const int banded_rows = 32768 / (int)sizeof(T);
    // the tallest blocks a CU's LDS takes (tile_kernels.hpp): the packed kernel is built for the banded limit
    constexpr int banded_rows_max = tile_banded_rows_max<T>();
    constexpr int scattered_rows_max = tile_scattered_rows_max<T>();
    bool want_pack = g_tile_pack != 0;
    // (... or slices several times the size of the entries they serve: 32 KiB of x out of L2 for a few hundred entries
    // costs as much as gathering them, measured on 30 uniformly random columns per row of a 1 M-column matrix)
    auto pack_pays = [&](const TilePlan<T> &p) {
        return p.entries >= (long long)p.pass_desc.size() * (chunk / 8) &&
               p.staged_cols * (long long)sizeof(T) <= 2 * p.entries * (long long)(4 + sizeof(T));
    };
    {
        // a slice of the matrix from its middle (rows keep their global columns)
        const int srows = rb ? rb : banded_rows;
        const int sample = std::min(Ml, 8 * srows), s0 = (Ml - sample) / 2;
        TilePlan<T> probe;
        const bool ok = build_rows(sample, s0, srows, false, 0, 0, probe, probe_arrays);
        const bool banded = ok && probe.staged_entries * 2 >= probe.entries;
        if (!rb) {
            if (banded) {
                rb = banded_rows;
            } else {
                rb = 16384;
                while (rb > 2048 && (long long)Ml < 448LL * rb) rb >>= 1;  // at least ~1.75 blocks per CU
                rb = std::min(rb, scattered_rows_max);
                tb.scattered = true;
            }
        }
        want_pack = want_pack && banded && rb <= banded_rows_max;
        if (want_pack) {  // the same slice as a packed plan
            const bool pok = build_rows(sample, s0, srows, true, 0, 0, probe, probe_arrays);
            want_pack = pok && pack_pays(probe);
        }
        probe_arrays.reset();
    }
    trace.mark("sample probes");
    tb.packed = want_pack && !tb.scattered;
    // The middle tier (round 3).  In a scattered plan every entry is a gathered value, and the gathers are what the
    // kernel waits for (config 5: 0.91 of 1.19 ms for the rows of up to 1024 entries).  Rows of more than mid_lo
    // entries, compacted into blocks as tall as the LDS takes (16384 fp32 rows), put enough entries into every
    // 32 KiB column range for the PACKED kernel: their x look-ups move to LDS like the long rows' -- when there are
    // enough of them (2^22 entries) for the extra launch and its slabs to pay.
    bool want_mid = false;
    // where the tier starts: the taller the blocks (fp32: 32512 rows, fp64: 16128), the shorter the rows that still fill
    // a column range -- config 5 (fp32): 1051 us from 128 entries on, 1031 from 96, 1023 from 64, 1014 from 48, 1046 from
    // 32 (profiles/r3_sweep_mid_lo.txt)
    const int mid_lo = g_tile_mid_lo > 0 ? g_tile_mid_lo : sizeof(T) == 4 ? 48 : 128;
    if (tb.scattered && g_tile_mid && g_tile_long && !g_tile_rows && g_tile_lmax > mid_lo && g_tile_pack) {
        long long mid_entries = 0;
        for (int r = 0; r < Ml; ++r)
            if (row_len[r] > mid_lo && row_len[r] <= g_tile_lmax) mid_entries += row_len[r];
        want_mid = mid_entries >= (4LL << 20);
        if (want_mid) lmax_eff = mid_lo;
    }
    // Mid-size matrices (fewer rows than kTileMinRows, but entries for four full passes on every place): only a
    // band of dense rows pays -- the packed plan, with blocks thin enough to give every place one (their slices stay
    // small next to their entries: 33 per row, 503 625 rows: 48.6 us in 493 blocks of 1024 rows against 68.3 us for the
    // gather kernel and 59.5 us in 247 blocks of 2048; 50 per row, 161 070 rows: 30.8 against 39.9 us).
    const bool mid = g_stream_tile < 0 && (long long)Ml < kTileMinRows;
    if (mid && !tb.packed) return;
    // (auto) tile plans pay from about 800 000 rows / columns on (kTileMinRows, kTileGatherMinCols): 5 uniformly random
    // columns per row tie with the gather kernel at 750 000 rows (37.4 vs 37.9 us) and lose at 500 000 (28 vs 23 us); at
    // 1.09 M rows 3 per row win 37 vs 44 us, power-law fp32 74 vs 107 us, a road-like band 24 vs 29 us.  (Before the block
    // count was fitted to rounds the break-even was 1.5 M columns: 2 x 256 short blocks on 256 places.)
    if (g_stream_tile < 0 && !tb.packed && N < kTileGatherMinCols) return;
    // How many blocks: equal-work blocks finish together, so the kernel runs in ROUNDS of as many blocks as the chip
    // holds at once (2 per CU banded, 1 scattered) and a last round of a few blocks costs a whole one -- 530 blocks on
    // 512 places took 2 x 215 us on a dense band, 1040 on 256 five rounds instead of four on the power-law matrix.
    // So (auto): the smallest number of rounds k for which blocks of up to the tallest height the LDS takes, closed
    // at in_tiles / (98.5 % of k rounds' places) entries, come out as at most k rounds' worth.
    long long target = 0;
    if (!g_tile_rows && g_tile_balance && g_tile_fit) {
        const int places = g_tile_places ? g_tile_places : (tb.scattered ? 1 : 2) * g_num_cus;
        const int rows_max = tb.scattered ? scattered_rows_max : banded_rows_max;
        long long in_tiles = 0;
        for (int r = 0; r < Ml; ++r)
            if (row_len[r] <= lmax_eff) in_tiles += row_len[r];
        const long long full = std::max(1, (Ml + rb - 1) / rb);
        const int b0 = (int)tile_cut_rows(Ml, row_len, lmax_eff, rb, std::max<long long>(chunk, (in_tiles + full - 1) / full)).size() - 1;
        // (a matrix whose blocks fill less than 90 % of one round keeps them: thinner blocks would each read more of x)
        const int kmax = mid ? 1 : b0 * 10LL > places * 9LL ? (b0 + places - 1) / places : 0;
        for (int k = 1; k <= kmax; ++k) {
            const long long want = (long long)k * places * 197 / 200;
            const long long t = std::max<long long>(chunk, (in_tiles + want - 1) / want);
            const int b = (int)tile_cut_rows(Ml, row_len, lmax_eff, rows_max, t).size() - 1;
            if (b <= k * places) {
                rb = rows_max;
                target = t;
                break;
            }
        }
    }
    trace.mark("block count fit");
    tb.have_tiles = build_rows(Ml, 0, rb, tb.packed, target,
                               // (its extra launch costs ~2 us: not for matrices whose whole product takes 25)
                               nz >= (16LL << 20) || g_stream_tile == 1 ? g_tile_min_pass : 0, tb.tiles, tb.tiles_dev);
    // (the remainder is for a few per cent of far-out entries: more than 4 % and the plan is rebuilt without one)
    if (tb.have_tiles && tb.packed && (long long)tb.tiles.rem_row.size() * 25 > tb.tiles.entries)
        tb.have_tiles = build_rows(Ml, 0, rb, tb.packed, target, 0, tb.tiles, tb.tiles_dev);
    // (the whole matrix may differ from the sample)
    if (tb.have_tiles && tb.packed && !pack_pays(tb.tiles)) {
        tb.packed = false;
        if (g_stream_tile < 0 && N < kTileGatherMinCols) {
            tb.have_tiles = false;
            return;
        }
        tb.have_tiles = build_rows(Ml, 0, rb, false, target, 0, tb.tiles, tb.tiles_dev);
    }
    // (auto) a matrix made mostly of rows beyond the tile limit gains nothing without the long rows' plan
    if (tb.have_tiles && g_stream_tile < 0 && !g_tile_long && tb.tiles.entries * 2 < nz) tb.have_tiles = false;
    if (!tb.have_tiles) return;
    trace.mark("ordinary tiles");
    // one workgroup per place of the chip walks several blocks back to back (tile_streams 0: one workgroup per block)
    // (tile_places: tests walk long streams on small matrices)
    tile_make_streams(tb.tiles, !g_tile_streams ? 0x3fffffff : g_tile_places ? g_tile_places : (tb.scattered ? 1 : 2) * g_num_cus);
    // The rows beyond the tile limit, compacted: their own row blocks (<= 2048 of them each), same passes -- so many
    // entries per column range that every pass is staged: the long rows' x lookups happen in LDS at the HBM streaming
    // rate instead of going through the gather path -- and a block's passes dealt out to many workgroups.
    std::vector<unsigned char> leftover = tb.tiles.split;
    if (g_tile_long)
        tb.have_long_tiles = build_long_tiles<T>(
            Ml, N, row_begin, row_len, chunk, leftover, tb.ltiles, tb.lt_rows, tb.lt_work, tb.lt_item_first, tb.lt_packed,
            [&](int rows, const int *begin_h, const int *len_h, int rpb, int lmax, int pos_bits, bool pack, TilePlan<T> &plan) {
                int *d_begin = nullptr, *d_len = nullptr;
                if (dev_ok) {  // the compacted row list exists on the host only: a device copy for this build
                    if (upload_array(&d_begin, begin_h, (size_t)rows, 1) == 0) dev_tmp.push_back(d_begin);
                    else dev_ok = false;
                    if (dev_ok && upload_array(&d_len, len_h, (size_t)rows, 1) == 0) dev_tmp.push_back(d_len);
                    else dev_ok = false;
                }
                return build_at(rows, begin_h, len_h, d_begin, d_len, rpb, lmax, pos_bits, pack, 0, 0, plan, tb.ltiles_dev);
            },
            want_mid ? g_tile_lmax : 0);  // (with a middle tier the long rows' plan starts behind it)
    // the middle tier: what the ordinary tiles left out up to tile_lmax (the long rows' plan has taken the rest)
    if (want_mid && tb.have_tiles)
        tb.have_mid_tiles = build_long_tiles<T>(
            Ml, N, row_begin, row_len, chunk, leftover, tb.mtiles, tb.mt_rows, tb.mt_work, tb.mt_item_first, tb.mt_packed,
            [&](int rows, const int *begin_h, const int *len_h, int rpb, int lmax, int pos_bits, bool pack, TilePlan<T> &plan) {
                if (!pack) return false;  // (a middle tier with gather passes would be the ordinary tiles again, plus slabs)
                int *d_begin = nullptr, *d_len = nullptr;
                if (dev_ok) {
                    if (upload_array(&d_begin, begin_h, (size_t)rows, 1) == 0) dev_tmp.push_back(d_begin);
                    else dev_ok = false;
                    if (dev_ok && upload_array(&d_len, len_h, (size_t)rows, 1) == 0) dev_tmp.push_back(d_len);
                    else dev_ok = false;
                }
                return build_at(rows, begin_h, len_h, d_begin, d_len, rpb, lmax, pos_bits, pack, 0, 0, plan, tb.mtiles_dev);
            },
            // (its blocks fill a CU's LDS: one workgroup per CU, so whole rounds of the CUs -- three of them: config 5
            // 1008 / 954 / 898 / 918 / 943 / 996 us at 256 / 512 / 768 / 1008 / 1280 / 2016 items, profiles/r3_sweep_mid_items.txt)
            mid_lo, g_tile_lmax, 17, scattered_rows_max, g_tile_mid_items ? g_tile_mid_items : 3 * std::max(g_num_cus, 1));
    if (want_mid && tb.have_tiles && (!tb.have_mid_tiles || !tb.mt_packed)) {
        // the tier did not come about (its passes would average fewer than 256 entries, or its build failed): the plan
        // without one, from the start -- the ordinary tiles then take the rows up to tile_lmax again
        const int keep = g_tile_mid;
        g_tile_mid = 0;
        tb = TileBuild<T>();
        tile_plan_all<T>(Ml, N, row_begin, row_len, rp, nz, hcol, hval, tb, din);
        g_tile_mid = keep;
        return;
    }
    trace.mark("streams, long rows' tiles");
    bool any_left = false;
    for (unsigned char f : leftover) any_left |= f != 0;
    if (!any_left) return;
    if (!rp) {  // no CSR arrays on the device to fall back on
        tb.have_tiles = tb.have_long_tiles = false;
        return;
    }
    // whatever is left (rows of 2^21 entries and more, or all long rows with tile_long = 0) stays with the split-row
    // kernels, pieces cut at stripes of 1 MiB of x: a quarter of an XCD's L2
    const int stripe_cols = (1 << 20) / (int)sizeof(T);
    build_striped_pieces(Ml, rp, hcol, leftover, stripe_cols, tb.tile_pieces, tb.tile_long);
}

// csr_tile asks for up to a CU's whole LDS as dynamic shared memory; HIP wants that allowed per kernel beyond
// 64 KiB.  Done once per value type, at upload (not at launch: a launch may sit inside a graph capture).
template <typename T>
int tile_allow_lds() {
    // the attribute belongs to (function, device): spmv_hip_init may have switched devices since the last upload
    static int done_for_device = -1;
    const bool done = done_for_device == g_device;
    if (done) return 0;
    constexpr int kXpTrips = 2048 * (int)sizeof(T) / kTileTripBytes;  // the expanded plans' instantiation
    const void *fns[8] = {(const void *)csr_tile<T, false, 2048, kXpTrips, true>, (const void *)csr_tile<T, true, 2048, kXpTrips, true>,
                          (const void *)csr_tile<T, false, 2048, kTileTrips, false>, (const void *)csr_tile<T, true, 2048, kTileTrips, false>,
                          (const void *)csr_tile<T, false, 2048, kTileTrips, true>, (const void *)csr_tile<T, true, 2048, kTileTrips, true>,
                          (const void *)csr_tile<T, false, 2048, kTileTrips, false, true>,
                          (const void *)csr_tile<T, true, 2048, kTileTrips, false, true>};
    for (const void *fn : fns) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done_for_device = g_device;
    return 0;
}

// the plans' arrays -> the handle
template <typename T>
int tile_upload_all(spmv_csr_dev *m, const TileBuild<T> &tb) {
    int rc = 0;
    // the kernels need more than 64 KiB of dynamic LDS: where the device will not allow that the handle simply gets
    // no tiles and keeps its gather kernels (return 1; the message stays in spmv_hip_last_error)
    if ((tb.have_tiles || tb.have_long_tiles || tb.have_mid_tiles) && tile_allow_lds<T>()) return 1;
    if (tb.have_tiles) {
        const TilePlan<T> &tiles = tb.tiles;
        // (the kernel walks STREAMS: descriptors in stream order, the passes / blocks of every stream, the blocks' rows)
        rc |= upload_array(&m->tile_block_pass, tiles.stream_pass.data(), tiles.stream_pass.size(), 1);
        if (!rc) rc |= upload_array(&m->tile_block_row, tiles.block_row.data(), tiles.block_row.size(), 1);
        if (!rc) rc |= upload_array(&m->tile_pass, tiles.spass.data(), tiles.spass.size(), 1);
        TileDevArrays<T> *dev = tb.tiles_dev.get();  // built on the device: the entry arrays are there already
        if (!rc && !tiles.rem_row.empty()) {  // remainder: rows that have any, their entry ranges
            std::vector<int> rrow, rptr;
            for (size_t k = 0; k < tiles.rem_row.size(); ++k) {
                if (rrow.empty() || rrow.back() != tiles.rem_row[k]) {
                    rrow.push_back(tiles.rem_row[k]);
                    rptr.push_back((int)k);
                }
            }
            rptr.push_back((int)tiles.rem_row.size());
            rc |= upload_array(&m->tile_rem_row, rrow.data(), rrow.size(), 0);
            if (!rc) rc |= upload_array(&m->tile_rem_ptr, rptr.data(), rptr.size(), 0);
            if (!rc && dev) {
                m->tile_rem_col = dev->rem_col;
                m->tile_rem_val = dev->rem_val;
                dev->rem_col = nullptr;
                dev->rem_val = nullptr;
            }
            if (!rc && !dev) rc |= upload_array(&m->tile_rem_col, tiles.rem_col.data(), tiles.rem_col.size(), 0);
            if (!rc && !dev) rc |= upload_array((T **)&m->tile_rem_val, tiles.rem_val.data(), tiles.rem_val.size(), 0);
            if (!rc) {
                m->tile_rem_rows = (int)rrow.size();
                m->tile_rem_entries = (long long)tiles.rem_row.size();
                m->device_bytes += rrow.size() * 8 + tiles.rem_row.size() * (4 + sizeof(T));
            }
        }
        if (!rc) rc |= upload_array(&m->tile_stream_block, tiles.stream_block.data(), tiles.stream_block.size(), 1);
        if (!rc) rc |= upload_array(&m->tile_sblock_rows, tiles.sblock_rows.data(), tiles.sblock_rows.size(), 1);
        const size_t tcol_count = dev ? dev->tcol_count : tiles.tcol.size(), tkey_count = dev ? dev->tkey_count : tiles.tkey.size();
        if (!rc && dev) {
            m->tcol = dev->tcol;
            m->tkey = dev->tkey;
            m->tval = dev->tval;
            dev->tcol = nullptr;
            dev->tkey = nullptr;
            dev->tval = nullptr;
        }
        if (!rc && !dev) rc |= upload_array(&m->tcol, tiles.tcol.data(), tiles.tcol.size(), 0);
        if (!rc && !dev) rc |= upload_array(&m->tkey, tiles.tkey.data(), tiles.tkey.size(), 0);
        if (!rc && !dev) rc |= upload_array((T **)&m->tval, tiles.tval.data(), tiles.tval.size(), 0);
        if (!rc && !tb.tile_long.empty()) rc |= upload_array(&m->tile_long_rows, tb.tile_long.data(), tb.tile_long.size(), 0);
        if (!rc && !tb.tile_pieces.empty()) rc |= upload_array(&m->tile_pieces, tb.tile_pieces.data(), tb.tile_pieces.size(), 0);
        if (!rc) {
            m->tile_blocks = tiles.num_blocks;
            m->tile_streams = tiles.num_streams;
            m->tile_rows = tiles.rows_per_block;
            m->tile_chunk = tiles.chunk;
            m->tile_packed = tb.packed;
            m->tile_lds_min = tb.scattered ? 84 * 1024 : 0;  // more than half a CU's LDS: one workgroup per CU
            m->tile_passes = (int)tiles.pass_desc.size();
            m->tile_max_win = tiles.max_win;
            m->tile_entries = tiles.entries;
            m->tile_staged = tiles.staged_entries;
            m->tile_staged_cols = tiles.staged_cols;
            m->tile_padded = (long long)tcol_count - kTileChunkMax;
            m->tile_num_long = (int)tb.tile_long.size();
            m->tile_num_pieces = (int)tb.tile_pieces.size();
            m->device_bytes += tcol_count * (4 + sizeof(T)) + tkey_count * 2 + tiles.pass_desc.size() * 16 + tiles.block_pass.size() * 4 +
                               (tb.tile_pieces.size() + tb.tile_long.size()) * 16;
        }
    }
    // a compacted tier (the long rows' plan, the middle tier): block tables, work items, entry arrays, slabs
    auto upload_tier = [&](const TilePlan<T> &ltiles, TileDevArrays<T> *ldev, const std::vector<int> &rows,
                           const std::vector<int4> &work, const std::vector<int> &item_first, bool packed,
                           spmv_csr_dev::long_tiles &L) {
        std::vector<int> block_of_row(rows.size());
        for (int b = 0; b < ltiles.num_blocks; ++b)
            for (int v = ltiles.block_row[(size_t)b]; v < ltiles.block_row[(size_t)b + 1]; ++v) block_of_row[(size_t)v] = b;
        rc |= upload_array(&L.block_row, ltiles.block_row.data(), ltiles.block_row.size(), 1);
        if (!rc) rc |= upload_array(&L.block_pass, ltiles.block_pass.data(), ltiles.block_pass.size(), 1);
        if (!rc) rc |= upload_array(&L.block_of_row, block_of_row.data(), block_of_row.size(), 1);
        if (!rc) rc |= upload_array(&L.item_first, item_first.data(), item_first.size(), 1);
        if (!rc) rc |= upload_array(&L.row_map, rows.data(), rows.size(), 1);
        if (!rc) rc |= upload_array(&L.pass, ltiles.pass_desc.data(), ltiles.pass_desc.size(), 1);
        if (!rc) rc |= upload_array(&L.work, work.data(), work.size(), 1);
        const size_t ltcol_count = ldev ? ldev->tcol_count : ltiles.tcol.size(), ltkey_count = ldev ? ldev->tkey_count : ltiles.tkey.size();
        if (!rc && ldev) {
            L.tcol = ldev->tcol;
            L.tkey = ldev->tkey;
            L.tval = ldev->tval;
            ldev->tcol = nullptr;
            ldev->tkey = nullptr;
            ldev->tval = nullptr;
        }
        if (!rc && !ldev) rc |= upload_array(&L.tcol, ltiles.tcol.data(), ltiles.tcol.size(), 0);
        if (!rc && !ldev) rc |= upload_array(&L.tkey, ltiles.tkey.data(), ltiles.tkey.size(), 0);
        if (!rc && !ldev) rc |= upload_array((T **)&L.tval, ltiles.tval.data(), ltiles.tval.size(), 0);
        if (!rc) {
            const size_t slab_bytes = std::max<size_t>(1, work.size()) * (size_t)ltiles.rows_per_block * sizeof(T);
            hipError_t e = hipMalloc(&L.slab, slab_bytes);
            if (e != hipSuccess) rc = fail("hipMalloc(slabs) failed: %s", hipGetErrorString(e));
            m->device_bytes += slab_bytes;
        }
        if (!rc) {
            L.blocks = ltiles.num_blocks;
            L.rows = (int)rows.size();
            L.rows_per_block = ltiles.rows_per_block;
            L.passes = (int)ltiles.pass_desc.size();
            L.items = (int)work.size();
            L.max_win = ltiles.max_win;
            L.entries = ltiles.entries;
            L.padded = (long long)ltcol_count - kTileChunkMax;
            L.staged = ltiles.staged_entries;
            L.staged_cols = ltiles.staged_cols;
            L.packed = packed;
            m->device_bytes += ltcol_count * (4 + sizeof(T)) + ltkey_count * 2 + ltiles.pass_desc.size() * 16 + work.size() * 16 +
                               rows.size() * 8;
        }
    };
    // a plan whose passes gather (scattered columns: next to none of them dense enough to stage) runs on an expanded x:
    // tile_expand walks the entries slice by slice of x and writes every entry's value into its pass's segment of x',
    // which csr_tile<.., PACK> stages as that pass's window (tile_kernels.hpp).  The same products in the same order:
    // the same bits as the gather passes.
    // auto: plans from 2^22 entries on with fewer than a tenth of them in staged passes -- fp32 always (uniformly random
    // columns, 2 M x 10 ... 16 M x 8: -17 ... -37 %; config 5's short rows 491 -> 141 + 180 us), fp64 when x is beyond
    // 100 MB: an fp64 gather brings twice the bytes per line and the expansion moves twice the bytes per entry -- 4 M x
    // 20: 518 -> 539 us, 12 M x 4: 462 -> 454, 16 M x 8: 1095 -> 939 (profiles/r3_ab_expand.txt)
    // ("tile_expand" 0: never, 1: always)
    const bool expand_pays = sizeof(T) == 4 || (long long)m->N * (long long)sizeof(T) >= (100LL << 20);
    if (!rc && tb.have_tiles && !tb.packed && m->tcol && m->tile_padded > 0 &&
        (g_tile_expand > 0 || (g_tile_expand < 0 && expand_pays && m->tile_entries >= (1 << 22) &&
                               m->tile_staged * 10 <= m->tile_entries))) {
        const size_t slots = (size_t)m->tile_padded;
        m->expansion = new (std::nothrow) TileExpansion();
        if (!m->expansion) return fail("csr_upload: out of host memory");
        std::string err;
        if (tile_build_expansion<T>(m->N, m->tcol, m->tkey, slots, m->tile_pass, (int)tb.tiles.spass.size(), g_stream, *m->expansion, err) < 0)
            return fail("%s", err.c_str());
        const size_t xe_bytes = (slots + kTileChunkMax) * sizeof(T);
        hipError_t e = hipMalloc(&m->xe, xe_bytes);
        if (e == hipSuccess) e = hipMemsetAsync(m->xe, 0, xe_bytes, g_stream);
        if (e != hipSuccess) return fail("hipMalloc(expanded x) failed: %s", hipGetErrorString(e));
        m->device_bytes += xe_bytes + (slots + 64) * 2 + (slots + kTileChunkMax) * 4 + tb.tiles.spass.size() * 16 +
                           (size_t)m->expansion->chunks * 24 + (m->expansion->runs + m->expansion->groups) * 4;
    }
    if (!rc && tb.have_long_tiles) upload_tier(tb.ltiles, tb.ltiles_dev.get(), tb.lt_rows, tb.lt_work, tb.lt_item_first, tb.lt_packed, m->lt);
    if (!rc && tb.have_mid_tiles) upload_tier(tb.mtiles, tb.mtiles_dev.get(), tb.mt_rows, tb.mt_work, tb.mt_item_first, tb.mt_packed, m->mt);
    return rc;
}

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
        int r1 = r, maxlen = 0;
        while (r1 < M && r1 - r < rows_cap && rp[r1 + 1] - base <= cap && rp[r1 + 1] - rp[r1] <= cap - 3 &&
               !skew_cut(r1 - r, maxlen, rp[r1 + 1] - rp[r1])) {
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
            maxlen = std::max(maxlen, rp[r1 + 1] - rp[r1]);
            ++r1;
        }
        if (r1 == r) {
            // one row alone touches more lines than a block may list: it goes to the split-row
            // (gather) kernels like a long row; a matrix made of such rows keeps the gather kernel
            // ... and a SMALL matrix with such a row keeps it too: the split-row kernels are two more launches
            // (~5 us), more than the whole product of a matrix below ~1 M entries (adder_dcop_32-size stand-in:
            // 11.5 us with the plan + split rows against 6.3 us for one launch, profiles/r2_reference_list_stand_ins.md)
            if (nz < kSplitMinEntries) return false;
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

// The x-window plan of the blocks `desc` (cut by entries and rows only) built on the device from the
// uploaded column indices (plan_kernels.hpp).  1: the handle carries the plan; 0: some block lists
// more than kLocalLinesMax lines, the host builder (which may cut blocks by lines or split rows)
// has to decide; 2: so many blocks list so many lines that the host builder would refuse too; -1: HIP error.
template <int SHIFT>
int csr_plan_on_device(spmv_csr_dev *m, const std::vector<int4> &desc, long long nz) {
    const int W = (int)desc.size();
    if (W == 0) return 0;
    std::vector<long long> seg_begin((size_t)W);
    std::vector<int> seg_len((size_t)W);
    for (int w = 0; w < W; ++w) {
        seg_begin[w] = desc[w].y;
        seg_len[w] = desc[w].w - desc[w].y;
        if (seg_len[w] > kPlanCap) return 0;
    }
    long long *d_begin = nullptr;
    int *d_len = nullptr, *d_n = nullptr, *d_off = nullptr;
    int result = -1;
    do {
        if (upload_array(&d_begin, seg_begin.data(), seg_begin.size(), 0)) break;
        if (upload_array(&d_len, seg_len.data(), seg_len.size(), 0)) break;
        hipError_t e = hipMalloc((void **)&d_n, (size_t)W * sizeof(int));
        if (e != hipSuccess) { fail("csr plan: hipMalloc failed: %s", hipGetErrorString(e)); break; }
        hipLaunchKernelGGL((plan_count<SHIFT>), dim3(W), dim3(kBlock), 0, g_stream, W, d_begin, d_len, m->col, d_n);
        std::vector<int> nl((size_t)W);
        e = hipMemcpyAsync(nl.data(), d_n, (size_t)W * sizeof(int), hipMemcpyDeviceToHost, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        if (e != hipSuccess) { fail("csr plan: count pass failed: %s", hipGetErrorString(e)); break; }
        std::vector<int> line_off((size_t)W);
        std::vector<int2> ldesc((size_t)W);
        long long total = 0;
        int widest = 0;
        bool fits = true;
        for (int w = 0; w < W && fits; ++w) {
            const int n = std::max(nl[w], 1);  // a block of empty rows still stages one line
            fits = nl[w] <= kLocalLinesMax && total + n < (1LL << 31);
            line_off[w] = (int)total;
            ldesc[w] = int2{(int)total, n};
            total += n;
            widest = std::max(widest, n);
        }
        if (!fits) {
            // Some block lists too many lines.  The host builder may still find a plan (blocks cut by lines, a few rows
            // split) -- unless the matrix is plainly scattered: a block with n lines needs at least n / 256 line-limited
            // blocks, a line-limited block can overlap two of ours, and the host refuses a plan with more than 1.2 x
            // our block count.  2 = refused here (the host builder would spend ~0.2 s at 2.6e8 entries to say the same).
            long long need = 0;
            for (int w = 0; w < W; ++w) need += std::max(1, (nl[w] + kLocalLinesMax - 1) / kLocalLinesMax);
            result = (need + 1) / 2 > (long long)W + W / 5 + 1 ? 2 : 0;
            break;
        }
        if (upload_array(&d_off, line_off.data(), line_off.size(), 0)) break;
        e = hipMalloc((void **)&m->lines, ((size_t)total + kLocalLinesMax) * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&m->lcol, ((size_t)nz + kPad) * sizeof(unsigned short));
        if (e == hipSuccess) e = hipMemsetAsync(m->lines, 0, ((size_t)total + kLocalLinesMax) * sizeof(int), g_stream);
        if (e == hipSuccess) e = hipMemsetAsync(m->lcol, 0, ((size_t)nz + kPad) * sizeof(unsigned short), g_stream);
        if (e != hipSuccess) { fail("csr plan: allocation failed: %s", hipGetErrorString(e)); break; }
        hipLaunchKernelGGL((plan_fill<SHIFT>), dim3(W), dim3(kBlock), 0, g_stream, W, d_begin, d_len, m->col, d_off,
                           m->lines, m->lcol);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        if (e != hipSuccess) { fail("csr plan: fill pass failed: %s", hipGetErrorString(e)); break; }
        if (upload_array(&m->ldesc4, desc.data(), desc.size(), 1)) break;
        if (upload_array(&m->ldesc, ldesc.data(), ldesc.size(), 1)) break;
        m->local_blocks = W;
        m->local_lines = total;
        m->local_stage_lines = std::max(kLocalLineQuantum,
                                        (widest + kLocalLineQuantum - 1) / kLocalLineQuantum * kLocalLineQuantum);
        result = 1;
    } while (0);
    (void)hipFree(d_begin);
    (void)hipFree(d_len);
    (void)hipFree(d_n);
    (void)hipFree(d_off);
    if (result != 1) {
        (void)hipFree(m->lines);
        (void)hipFree(m->lcol);
        (void)hipFree(m->ldesc4);
        (void)hipFree(m->ldesc);
        m->lines = nullptr;
        m->lcol = nullptr;
        m->ldesc4 = nullptr;
        m->ldesc = nullptr;
        m->local_blocks = 0;
    }
    return result;
}

// Where the value array lies decides -- for as long as the allocation lives, by a mechanism the counters at hand do not
// name (profiles/r3_placement_*.txt: not the XCD mapping, not the TLB, not one slow XCD; every block of one HALF of
// the matrix is a little slower) -- whether the x-window kernel runs the headline matrix in 182-187 or in 199-205 us.
// So a handle that streams enough values for it to matter times its own kernel on a few placements and keeps the best:
// up to g_place_tries fresh allocations of the value array (earlier candidates stay allocated meanwhile, so every one
// is a different place), 2 + 6 launches each.  ~2 ms per candidate at 100 M entries; upload itself takes 50-100.
// Three levels exist -- both halves of the array fast (180-182 us on the headline matrix), one (186-194), none
// (199-205); fresh allocations land on them roughly 2 : 5 : 5 (profiles/r3_placement_*.txt) -- so the search goes on
// until a candidate is 8.5 % faster than the slowest seen (= the top level reached) or the tries are used up.
template <typename T>
int csr_tune_placement(spmv_csr_dev *m) {
    const size_t bytes = ((size_t)m->nz + kPad) * sizeof(T);
    if (g_place_tries <= 0 || (size_t)m->nz * sizeof(T) < ((size_t)128 << 20) || m->tiles_only || !m->val) return 0;
    if (m->local_blocks == 0 && m->tile_blocks > 0) return 0;  // csr_tile streams its own re-ordered copy, not m->val
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return 0;
    }
    auto measure = [&](float &us) {
        for (int i = 0; i < 2; ++i)
            if (csr_launch_any(m, SPMV_CSR_AUTO, m->x, m->y, g_stream)) return -1;
        hipError_t e = hipEventRecord(e0, g_stream);
        for (int i = 0; i < 6 && e == hipSuccess; ++i)
            if (csr_launch_any(m, SPMV_CSR_AUTO, m->x, m->y, g_stream)) return -1;
        if (e == hipSuccess) e = hipEventRecord(e1, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) return fail("placement tuning: timing failed: %s", hipGetErrorString(e));
        us = ms * 1e3f / 6.0f;
        return 0;
    };
    int rc = 0;
    void *first = m->val, *best = m->val;
    float best_us = 0;
    std::vector<void *> others;
    rc = measure(best_us);
    m->place_first_us = best_us;
    m->place_tries = 1;
    float worst_us = best_us;
    for (int t = 0; t < g_place_tries && !rc; ++t) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) break;  // (out of memory for another copy: keep what we have)
        others.push_back(p);
        if (hipMemcpy(p, first, bytes, hipMemcpyDeviceToDevice) != hipSuccess) break;
        m->val = p;
        float us = 0;
        rc = measure(us);
        if (rc) break;
        ++m->place_tries;
        if (us < best_us * 0.985f) {  // (1.5 %: above the run-to-run noise of 6 launches)
            best = p;
            best_us = us;
        }
        worst_us = std::max(worst_us, us);
        if (best_us < worst_us * 0.915f) break;  // both halves of the array at their fast level (see above): nothing better to find
    }
    m->val = best;
    m->place_best_us = best_us;
    if (best != first) (void)hipFree(first);
    for (void *p : others)
        if (p != best) (void)hipFree(p);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// The pattern plan of a handle's x-window plan (csr_kernels.hpp, PAT; plan_kernels.hpp): built on the device from the
// plan's own arrays, whichever builder made them.  auto: kept when the tables hold at most a quarter of the slots.
// Where it pays (same handle, same placement, the two instantiations alternately: profiles/r3_ab_patterns.txt): the
// nlpkkt-like matrix 190 -> 178 us (bench.py over 4 uploads each; 204 -> 185 on a slow placement), a 27-point stencil
// 197 -> 190; neutral on the FEM-shaped matrix (75 per row: 155 / 157) and on nlpkkt80-size (59 / 60); a loss where the
// rows are short -- several rows per lane group, their patterns fetched pass after pass: 7-point 271 -> 285, 5-point
// 186 -> 210 --, in fp32 (124 -> 133: the same LDS work for half the bytes) and for matrices that live in the Infinity
// Cache (cant-size 10.8 -> 12.4).  The gain goes with the placement of the arrays: 1-2 % on a fast one, 9 % on a slow one.
// (nlpkkt80-size, 29 M entries: 56.3 -> 58.4.)  And even where it gains the gain goes with the placement of the arrays
// (27-point stencil: 204 -> 195 on one box, 177 -> 183 on another).  Hence auto: streamed matrices (the `nt` threshold)
// of at least 12 entries per row whose tables hold at most a quarter of the slots get a plan BUILT, and upload then
// times its own kernel with and without it and keeps the plan only if it is at least 2 % faster on this handle
// (csr_tune_patterns, beside the placement search).
int csr_build_patterns(spmv_csr_dev *m) {
    if (g_local_patterns == 0 || m->local_blocks <= 0 || !m->lcol || !m->ldesc4 || !m->row_ptr || m->M_local <= 0) return 0;
    if (g_local_patterns < 0 && (m->nz * (m->value_bytes + 2LL) <= (128LL << 20) || m->nz < 12LL * m->M_local)) return 0;
    UploadTrace trace("csr_build_patterns");
    const int B = m->local_blocks;
    int *rowflag = nullptr, *pcount = nullptr;
    long long *pbase = nullptr;
    auto drop_tmp = [&] {
        (void)hipFree(rowflag);
        (void)hipFree(pcount);
        (void)hipFree(pbase);
    };
    hipError_t e = hipMalloc((void **)&rowflag, (size_t)m->M_local * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&pcount, (size_t)B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&pbase, (size_t)B * sizeof(long long));
    if (e == hipSuccess) e = hipMemsetAsync(rowflag, 0, (size_t)m->M_local * sizeof(int), g_stream);
    if (e != hipSuccess) {
        drop_tmp();
        return fail("pattern plan: allocation failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL((pat_mark<256>), dim3(B), dim3(256), 0, g_stream, B, m->ldesc4, m->row_ptr, m->lcol, rowflag, pcount);
    std::vector<int> h_count((size_t)B);
    e = hipMemcpyAsync(h_count.data(), pcount, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    if (e != hipSuccess) {
        drop_tmp();
        return fail("pattern plan: marking the rows failed: %s", hipGetErrorString(e));
    }
    std::vector<long long> h_base((size_t)B);
    long long total = 0;
    int widest = 0;
    for (int b = 0; b < B; ++b) {
        h_base[(size_t)b] = total;
        total += h_count[(size_t)b];
        widest = std::max(widest, h_count[(size_t)b]);
    }
    trace.mark("rows marked");
    // (auto) a plan whose tables hold more than a quarter of the slots keeps reading the slot stream
    if ((g_local_patterns < 0 && total * 4 > m->nz) || total > 0x7ffffff0LL) {
        drop_tmp();
        return 0;
    }
    e = hipMalloc((void **)&m->ptab, ((size_t)total + 1024) * sizeof(unsigned short));
    if (e == hipSuccess) e = hipMalloc((void **)&m->rinfo, (size_t)m->M_local * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void **)&m->pdesc, (size_t)B * sizeof(int2));
    if (e == hipSuccess) e = hipMemsetAsync(m->ptab, 0, ((size_t)total + 1024) * sizeof(unsigned short), g_stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->rinfo, 0, (size_t)m->M_local * sizeof(unsigned), g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pbase, h_base.data(), (size_t)B * sizeof(long long), hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((pat_fill<256>), dim3(B), dim3(256), 0, g_stream, B, m->ldesc4, m->row_ptr, m->lcol, rowflag, pbase, m->rinfo,
                           m->ptab, m->pdesc);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    drop_tmp();
    if (e != hipSuccess) {
        (void)hipFree(m->ptab);
        (void)hipFree(m->rinfo);
        (void)hipFree(m->pdesc);
        m->ptab = nullptr;
        m->rinfo = nullptr;
        m->pdesc = nullptr;
        return fail("pattern plan: building the tables failed: %s", hipGetErrorString(e));
    }
    m->pat_slots = total;
    m->pat_max = widest;
    m->device_bytes += ((size_t)total + 1024) * 2 + (size_t)m->M_local * 4 + (size_t)B * 8;
    trace.mark("tables");
    return 0;
}

// (auto) the handle's kernel with and without its pattern plan, alternately, two rounds of 2 + 6 launches each: the plan
// stays if it is at least 2 % faster here.  Never a reason to lose the handle.
int csr_tune_patterns(spmv_csr_dev *m) {
    if (g_local_patterns >= 0 || !m->ptab || !m->x || !m->y) return 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return 0;
    }
    auto measure = [&](int patterns, float &us) {
        const int keep = g_local_patterns;
        g_local_patterns = patterns;
        int rc = 0;
        for (int i = 0; i < 2 && !rc; ++i) rc = csr_launch_any(m, SPMV_CSR_AUTO, m->x, m->y, g_stream);
        hipError_t e = rc ? hipErrorUnknown : hipEventRecord(e0, g_stream);
        for (int i = 0; i < 6 && e == hipSuccess && !rc; ++i) rc = csr_launch_any(m, SPMV_CSR_AUTO, m->x, m->y, g_stream);
        if (e == hipSuccess && !rc) e = hipEventRecord(e1, g_stream);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(g_stream);
        float ms = 0;
        if (e == hipSuccess && !rc) e = hipEventElapsedTime(&ms, e0, e1);
        g_local_patterns = keep;
        us = ms * 1e3f / 6.0f;
        return (e == hipSuccess && !rc) ? 0 : -1;
    };
    float with_us = 0, without_us = 0;
    bool ok = true;
    for (int round = 0; round < 2 && ok; ++round) {
        float a = 0, b = 0;
        ok = measure(1, a) == 0 && measure(0, b) == 0;
        with_us = round ? std::min(with_us, a) : a;
        without_us = round ? std::min(without_us, b) : b;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    m->pat_with_us = with_us;
    m->pat_without_us = without_us;
    if (!ok || with_us > 0.98f * without_us) {  // not faster here: the slot stream stays
        (void)hipFree(m->ptab);
        (void)hipFree(m->rinfo);
        (void)hipFree(m->pdesc);
        m->ptab = nullptr;
        m->rinfo = nullptr;
        m->pdesc = nullptr;
        m->device_bytes -= std::min(m->device_bytes, ((size_t)m->pat_slots + 1024) * 2 + (size_t)m->M_local * 4 + (size_t)m->local_blocks * 8);
        m->pat_slots = 0;
    }
    return 0;
}

template <typename T>
int csr_upload_impl(int M, int N, const int *row_ptr, const int *col_idx, const T *values, int row0,
                    int row1, spmv_csr_dev **out, int *adopt_col = nullptr, T *adopt_val = nullptr) {
    // adopt_col / adopt_val: device arrays of row_ptr[row1] - row_ptr[row0] (+ kPad zeroed) entries that the
    // handle takes over instead of uploading col_idx / values (which are then NULL); columns already checked
    if (need_device()) return -1;
    if (!out) return fail("csr_upload: out is NULL");
    *out = nullptr;
    if (M < 0 || N < 0 || !row_ptr) return fail("csr_upload: bad arguments");
    if (row0 < 0 || row1 < row0 || row1 > M) return fail("csr_upload: bad row range [%d, %d) of %d", row0, row1, M);
    const int Ml = row1 - row0;
    const int e0 = row_ptr[row0], e1 = row_ptr[row1];
    const long long nz = (long long)e1 - e0;
    if (nz < 0) return fail("csr_upload: row_ptr is not monotone");
    if (nz > 0x7fffffffLL - kPad) return fail("csr_upload: %lld entries exceed the 32-bit entry index of the kernels", nz);
    if (nz > 0 && !adopt_col && (!col_idx || !values)) return fail("csr_upload: col_idx / values are NULL");
    UploadTrace trace("csr_upload");
    if ((unsigned long long)N * sizeof(T) >= (1ull << 32))
        return fail("csr_upload: N = %d exceeds the 32-bit gather offset range of the kernels", N);
    // a column index outside [0, N) would make the kernels gather out of bounds
    if (col_idx && e1 > e0) {  // (a few threads: one pass over 2.6e8 indices is 0.15 s of a 1 s upload on one core)
        const int threads = (long long)e1 - e0 >= (1 << 22) ? (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1;
        std::vector<long long> first_bad((size_t)threads, -1);
        tile_detail::run_threads(threads, [&](int th) {
            const long long lo = e0 + ((long long)e1 - e0) * th / threads, hi = e0 + ((long long)e1 - e0) * (th + 1) / threads;
            for (long long e = lo; e < hi; ++e)
                if ((unsigned)col_idx[e] >= (unsigned)N) {
                    first_bad[(size_t)th] = e;
                    break;
                }
        });
        for (long long e : first_bad)
            if (e >= 0) return fail("csr_upload: column index %d at entry %lld is outside [0, %d)", col_idx[e], e, N);
    }

    trace.mark("column check");
    spmv_csr_dev *m = new (std::nothrow) spmv_csr_dev();
    if (!m) return fail("csr_upload: out of host memory");
    // on failure adopted arrays go back to the caller untouched
    auto drop = [&](spmv_csr_dev *h) {
        if (adopt_col) h->col = nullptr;
        if (adopt_val) h->val = nullptr;
        spmv_hip_csr_free(h);
    };
    m->value_bytes = (int)sizeof(T);
    m->M_local = Ml;
    m->M_total = M;
    m->N = N;
    m->row0 = row0;
    m->nz = nz;
    m->col = adopt_col;
    m->val = adopt_val;

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
    bool on_device = false;
    if (g_stream_local && nz > 0 && (g_stream_cap == 0 || g_stream_cap == lcap)) {
        csr_build_blocks(Ml, rp.data(), lcap, kStreamRowsCap, desc, pieces, long_rows);
        // on the device when no block lists more than 256 lines (the common case for matrices that get
        // a plan at all: then the line limit would not have moved a block boundary on the host either);
        // otherwise the host builder decides (line-limited blocks, split rows, or no plan)
        int dev = 0;
        if (g_plan_on_device && lcap == kPlanCap) {
            if (!m->col && upload_array(&m->col, col_idx + e0, (size_t)nz, kPad)) {
                drop(m);
                return -1;
            }
            dev = csr_plan_on_device<line_shift>(m, desc, nz);
            if (dev < 0) {
                drop(m);
                return -1;
            }
        }
        on_device = dev == 1;
        const bool refused = dev == 2;  // plainly scattered columns: no plan, and no need to ask the host builder
        std::vector<int> col_back;  // adopted arrays live on the device only: the host builder needs a copy
        if (!on_device && !refused && !col_idx) {
            col_back.resize((size_t)nz);
            if (hipMemcpy(col_back.data(), m->col, (size_t)nz * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
                drop(m);
                return fail("csr_upload: copying the columns back for the host plan failed");
            }
        }
        have_local = on_device ||
                     (!refused &&
                      csr_build_local(Ml, N, rp.data(), col_idx ? col_idx + e0 : col_back.data(), nz, lcap,
                                      kStreamRowsCap, line_shift, kLocalLinesMax, desc, local));
        if (on_device) local.split.assign((size_t)Ml, 0);
    }
    trace.mark("x-window plan (or its refusal)");
    // No x-window plan: columns too scattered for 256 lines per block.  Then the 2-D tiles (csr_tile):
    // row-block accumulators in LDS, the block's entries re-ordered into column passes so that all
    // workgroups sweep x together (L2-resident band), dense passes staged in LDS.  Needs enough row
    // blocks to fill the chip; rows longer than tile_lmax stay with the split-row kernels, their
    // pieces cut at column stripes.
    TileBuild<T> tb;
    if (!have_local && nz > 0 && g_stream_tile != 0 && g_stream_cap == 0 &&
        (g_stream_tile == 1 || (long long)Ml >= kTileMinRows || nz >= kTileMidEntries)) {
        std::vector<int> col_copy;
        std::vector<T> val_copy;
        const int *hcol = col_idx ? col_idx + e0 : nullptr;
        const T *hval = values ? values + e0 : nullptr;
        if (!hcol) {  // adopted arrays live on the device only
            col_copy.resize((size_t)nz);
            val_copy.resize((size_t)nz);
            if (hipMemcpy(col_copy.data(), m->col, (size_t)nz * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(val_copy.data(), m->val, (size_t)nz * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) {
                drop(m);
                return fail("csr_upload: copying the matrix back for the tile plan failed");
            }
            hcol = col_copy.data();
            hval = val_copy.data();
        }
        std::vector<int> row_len((size_t)Ml);
        for (int r = 0; r < Ml; ++r) row_len[(size_t)r] = rp[(size_t)r + 1] - rp[(size_t)r];
        // the plan is built on the device from the CSR arrays there (tile_plan_device.hpp; "tile_plan_on_device" 0: on
        // host threads, tile_plan.hpp -- the two give the same bytes): the matrix goes up first
        TileDevInput<T> din;
        int *d_row_len = nullptr;
        bool on_dev = g_tile_plan_on_device != 0;
        if (on_dev) {
            int urc = 0;
            if (!m->col) urc |= upload_array(&m->col, col_idx ? col_idx + e0 : nullptr, (size_t)nz, kPad);
            if (!urc && !m->val) urc |= upload_array((T **)&m->val, values ? values + e0 : nullptr, (size_t)nz, kPad);
            if (!urc && !m->row_ptr) urc |= upload_array(&m->row_ptr, rp.data(), rp.size(), (size_t)kRowPtrPad);
            if (!urc) urc |= upload_array(&d_row_len, row_len.data(), row_len.size(), 1);
            if (urc) {
                (void)hipFree(d_row_len);
                drop(m);
                return -1;
            }
            din.row_begin = m->row_ptr;
            din.row_len = d_row_len;
            din.col = m->col;
            din.val = (const T *)m->val;
            din.stream = g_stream;
        }
        trace.mark("matrix to the device");
        tile_plan_all<T>(Ml, N, rp.data(), row_len.data(), rp.data(), nz, hcol, hval, tb, on_dev ? &din : nullptr);
        (void)hipFree(d_row_len);
        trace.mark("tile plans");
    }
    // else: larger stages amortise per-workgroup latency on big matrices; small ones need
    // enough workgroups to fill 256 CUs (measured: cant-like 2048, nlpkkt-like 4096)
    m->stream_cap = have_local ? lcap : (g_stream_cap ? g_stream_cap : (nz >= (16LL << 20) ? 4096 : 2048));
    m->local_cap = lcap;
    // the ring kernel stages at most kRingRows - 1 rows per block; only worth it when such
    // blocks are still (nearly) full, i.e. rows are not tiny
    int rows_cap = kStreamRowsCap;
#ifdef SPMV_EXPERIMENTAL
    static_assert(kRingRows + 64 <= kRowPtrPad, "row_ptr padding covers the ring kernel's staged segment");
    m->ring_ok = m->stream_cap == kRingCap && Ml > 0 && (double)nz / Ml >= 1.25 * kRingCap / (kRingRows - 1);
    if (m->ring_ok) rows_cap = kRingRows - 1;
#endif
    // Skewed rows (gather kernels): a block's rows are summed by lane groups sized for the NUMBER of rows in it, so in
    // a block of a few hundred short rows one row of a thousand entries is summed by a single lane while everybody else
    // waits (webbase-like stand-in, 1 M rows, 3.3 entries per row on average, 2 262 in the longest: 73 us; with its rows
    // beyond the limit below handed to the split-row kernels, one workgroup each: see profiles/r2_reference_list_stand_ins.md).
    // Rows longer than max(128, 16 x the average row) count as long there -- from 2^20 entries on: the two extra
    // launches of the split-row kernels cost ~5 us, more than a small matrix's whole product.
    std::vector<unsigned char> skew_split;
    if (!have_local && g_skew_rows && Ml > 0 && nz >= (1LL << 20)) {
        const long long limit = std::max<long long>(128, 16 * (nz / Ml));
        if (limit < m->stream_cap - 3) {
            bool any = false;
            skew_split.assign((size_t)Ml, 0);
            for (int r = 0; r < Ml; ++r)
                if (rp[(size_t)r + 1] - rp[(size_t)r] > limit) skew_split[(size_t)r] = 1, any = true;
            if (!any) skew_split.clear();
        }
    }
    csr_build_blocks(Ml, rp.data(), m->stream_cap, rows_cap, desc, pieces, long_rows,
                     have_local ? &local.split : skew_split.empty() ? nullptr : &skew_split);
    m->num_blocks = (int)desc.size();
    m->num_long = (int)long_rows.size();
    m->num_partial = (int)pieces.size();
    const int num_partial = m->num_partial;

    int rc = 0;
    if (!m->row_ptr) rc |= upload_array(&m->row_ptr, rp.data(), rp.size(), (size_t)kRowPtrPad);
    if (!rc && have_local && !on_device) {
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
    if (!rc && m->local_blocks > 0) rc |= csr_build_patterns(m);
    if (!rc && !m->col) rc |= upload_array(&m->col, col_idx ? col_idx + e0 : nullptr, (size_t)nz, kPad);
    if (!rc && !m->val) rc |= upload_array((T **)&m->val, values ? values + e0 : nullptr, (size_t)nz, kPad);
    if (!rc) rc |= upload_array(&m->desc, desc.data(), desc.size(), 1);
    if (!rc && m->num_long) rc |= upload_array(&m->long_rows, long_rows.data(), long_rows.size(), 0);
    if (!rc && num_partial) rc |= upload_array(&m->pieces, pieces.data(), pieces.size(), 0);
    const size_t bytes_before_tiles = m->device_bytes;
    if (!rc && tile_upload_all<T>(m, tb) < 0) rc = -1;  // (1 = the device refused the LDS size: no tiles, not an error)
    const size_t tile_bytes = m->device_bytes - bytes_before_tiles;  // (device_bytes is recomputed below)
    const int partial_slots = std::max(num_partial, (int)tb.tile_pieces.size());
    if (!rc && partial_slots) {
        hipError_t e = hipMalloc(&m->partial, (size_t)partial_slots * sizeof(T));
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
        drop(m);
        return -1;
    }
    m->device_bytes = rp.size() * 4 + ((size_t)nz + kPad) * (4 + sizeof(T)) + desc.size() * 16 +
                      long_rows.size() * 16 + pieces.size() * 16 + (size_t)num_partial * sizeof(T) +
                      ((size_t)N + (size_t)M) * sizeof(T);
    if (have_local)
        m->device_bytes += (size_t)m->local_blocks * 24 + ((size_t)m->local_lines + kLocalLinesMax) * 4 +
                           ((size_t)nz + kPad) * 2;
    m->device_bytes += tile_bytes;

    // lanes per row for the SUBWAVE kernel: about half the mean row length,
    // rounded to a power of two, so that a typical row takes 1-2 passes
    const double mean = Ml ? (double)nz / Ml : 0.0;
    int lanes = pow2_floor(std::max(2, (int)(mean / 2.0 + 0.5)));
    m->lanes_per_row = std::min(32, std::max(2, lanes));
    m->auto_variant = SPMV_CSR_STREAM;
    // Mid-size matrices that get neither plan (scattered columns, below the tile plans' size): the lane-group kernel
    // beats the gather stream kernels by 3-10 % on every such stand-in of the reference's list (cop20k_A-size 19.2 vs
    // 21.2 us, PR02R-size 37 vs 41, amazon0302-size 12.1 vs 13.5; profiles/r2_reference_list_stand_ins.md) -- unless rows
    // are skewed (its lanes per row are fixed), and not on large ones (road-like 12 M rows: 282 vs 255 us).
    if (!have_local && !tb.have_tiles && nz < (20LL << 20) && max_row <= std::max(64.0, 8.0 * mean))
        m->auto_variant = SPMV_CSR_SUBWAVE;
    trace.mark("blocks, remaining uploads, vectors");
    // Both searches below compare launch times: they begin in the card's steady state -- after an idle stretch (this
    // upload) the same launch costs 178, then 208, then, from about the 60th on, 175 us (profiles/r3_launch_time_series.txt),
    // a drift as large as what the searches look for.  ~15 ms of the handle's own kernel first.
    if ((m->ptab && g_local_patterns < 0) || (g_place_tries > 0 && (size_t)m->nz * sizeof(T) >= ((size_t)128 << 20) && m->val && !m->tiles_only)) {
        const auto t_settle = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_settle).count() < 15.0) {
            bool bad = false;
            for (int i = 0; i < 16 && !bad; ++i) bad = csr_launch_any(m, SPMV_CSR_AUTO, m->x, m->y, g_stream) != 0;
            if (bad || hipStreamSynchronize(g_stream) != hipSuccess) break;
        }
    }
    (void)csr_tune_patterns(m);      // (auto) the pattern plan stays only where it measures faster on this handle
    (void)csr_tune_placement<T>(m);  // (never a reason to lose the handle: whatever went wrong in there, it holds a valid array)
    trace.mark("placement tuning");
    *out = m;
    return 0;
}

}  // namespace

// Host-only self-check of the upload-time preprocessing (no device needed): builds the workgroup
// blocks and the x-window plan for the given CSR structure exactly as upload does and verifies
// their invariants -- blocks and split rows partition the rows, every block fits the stage and
// lists at most kLocalLinesMax strictly ascending lines, every entry's 16-bit slot leads back to
// its column.  stats (optional, 6 ints): gather blocks, x-window blocks (0 = no plan), listed
// lines, widest block's lines, long rows, rows handed over because of the line limit.
static int spmv_hip_csr_plan_check_body(int M, int N, const int *row_ptr, const int *col_idx, int value_bytes,
                                       int *stats) {
    if (M < 0 || N < 0 || !row_ptr || (value_bytes != 4 && value_bytes != 8)) return fail("plan_check: bad arguments");
    const long long nz = row_ptr[M];
    if (nz > 0 && !col_idx) return fail("plan_check: col_idx is NULL");
    for (int r = 0; r < M; ++r)
        if (row_ptr[r + 1] < row_ptr[r]) return fail("plan_check: row_ptr decreases at row %d", r);
    for (long long e = 0; e < nz; ++e)
        if ((unsigned)col_idx[e] >= (unsigned)N) return fail("plan_check: column %d outside [0, %d)", col_idx[e], N);
    const int cap = 2048, shift = value_bytes == 8 ? 4 : 5, mask = (1 << shift) - 1;
    std::vector<int4> desc, pieces, long_rows;
    LocalPlan plan;
    csr_build_blocks(M, row_ptr, cap, kStreamRowsCap, desc, pieces, long_rows);
    const bool have = nz > 0 && csr_build_local(M, N, row_ptr, col_idx, nz, cap, kStreamRowsCap, shift,
                                                kLocalLinesMax, desc, plan);
    csr_build_blocks(M, row_ptr, cap, kStreamRowsCap, desc, pieces, long_rows, have ? &plan.split : nullptr);
    // gather blocks + long rows partition the rows
    std::vector<unsigned char> seen((size_t)M, 0);
    auto claim = [&](int r0, int n, const char *what) {
        for (int r = r0; r < r0 + n; ++r) {
            if (r < 0 || r >= M || seen[r]) return fail("plan_check: row %d claimed twice or out of range (%s)", r, what);
            seen[r] = 1;
        }
        return 0;
    };
    // no lane of the row-sum phase adds up more than kSkewPerLane entries unless its row is alone in the block
    auto lane_load_ok = [&](const int4 &d) {
        int longest = 0;
        for (int r = d.x; r < d.x + d.z; ++r) longest = std::max(longest, row_ptr[r + 1] - row_ptr[r]);
        return d.z == 1 || longest / host_lanes_for_rows(d.z) <= kSkewPerLane;
    };
    for (const int4 &d : desc) {
        if (claim(d.x, d.z, "gather block")) return -1;
        if (d.z <= 0 || d.z > kStreamRowsCap || d.y != row_ptr[d.x] || d.w != row_ptr[d.x + d.z] ||
            d.w - (d.y & kBaseMask) > cap)
            return fail("plan_check: gather block at row %d is malformed", d.x);
        if (!lane_load_ok(d)) return fail("plan_check: gather block at row %d leaves a long row to too few lanes", d.x);
    }
    int split_rows = 0;
    for (const int4 &l : long_rows) {
        if (claim(l.x, 1, "long row")) return -1;
        long long covered = 0;
        for (int k = 0; k < l.z; ++k) {
            const int4 &pc = pieces[(size_t)l.y + k];
            if (pc.x != l.x || pc.z - pc.y <= 0 || pc.z - pc.y > kLongPiece) return fail("plan_check: bad piece of row %d", l.x);
            covered += pc.z - pc.y;
        }
        if (covered != row_ptr[l.x + 1] - row_ptr[l.x]) return fail("plan_check: pieces of row %d do not cover it", l.x);
        const bool by_length = row_ptr[l.x + 1] - row_ptr[l.x] > cap - 3;
        if (!by_length && !(have && plan.split[l.x])) return fail("plan_check: row %d is split without a reason", l.x);
        split_rows += !by_length;
    }
    for (int r = 0; r < M; ++r)
        if (!seen[r]) return fail("plan_check: row %d belongs to no block", r);
    int widest = 0;
    if (have) {
        std::fill(seen.begin(), seen.end(), 0);
        if (plan.desc.size() != plan.ldesc.size()) return fail("plan_check: descriptor arrays differ in length");
        for (size_t b = 0; b < plan.desc.size(); ++b) {
            const int4 &d = plan.desc[b];
            const int2 &ld = plan.ldesc[b];
            if (claim(d.x, d.z, "x-window block")) return -1;
            if (d.z <= 0 || d.z > kStreamRowsCap || d.y != row_ptr[d.x] || d.w != row_ptr[d.x + d.z] ||
                d.w - (d.y & kBaseMask) > cap)
                return fail("plan_check: x-window block at row %d is malformed", d.x);
            if (!lane_load_ok(d)) return fail("plan_check: x-window block at row %d leaves a long row to too few lanes", d.x);
            if (ld.y < 1 || ld.y > kLocalLinesMax || ld.x < 0 || (size_t)ld.x + ld.y > plan.lines.size())
                return fail("plan_check: line list of block %zu is malformed", b);
            for (int k = 1; k < ld.y; ++k)
                if (plan.lines[ld.x + k] <= plan.lines[ld.x + k - 1]) return fail("plan_check: lines of block %zu not ascending", b);
            for (int e = d.y; e < d.w; ++e) {
                const int slot = plan.lcol[e], rank = slot >> shift;
                if (rank >= ld.y || plan.lines[ld.x + rank] != (col_idx[e] >> shift) || (slot & mask) != (col_idx[e] & mask))
                    return fail("plan_check: entry %d (column %d) has slot %d, which is not its column", e, col_idx[e], slot);
            }
            widest = std::max(widest, ld.y);
        }
        // every row is in an x-window block, long, or split because of its lines
        for (int r = 0; r < M; ++r) {
            const bool by_length = row_ptr[r + 1] - row_ptr[r] > cap - 3;
            if (!seen[r] && !by_length && !plan.split[r]) return fail("plan_check: row %d is in no x-window block", r);
            if (seen[r] && (by_length || plan.split[r])) return fail("plan_check: row %d is both in a block and split", r);
        }
        if (plan.stage_lines < widest || plan.stage_lines % kLocalLineQuantum) return fail("plan_check: bad stage size");
    }
    if (stats) {
        stats[0] = (int)desc.size();
        stats[1] = have ? (int)plan.desc.size() : 0;
        stats[2] = have ? (int)plan.lines.size() - kLocalLinesMax : 0;
        stats[3] = widest;
        stats[4] = (int)long_rows.size();
        stats[5] = split_rows;
    }
    return 0;
}

extern "C" int spmv_hip_csr_plan_check(int M, int N, const int *row_ptr, const int *col_idx, int value_bytes,
                                       int *stats) {
    return guarded("csr_plan_check", [&] { return spmv_hip_csr_plan_check_body(M, N, row_ptr, col_idx, value_bytes, stats); });
}

// Host-only self-check of the csr_tile plan (no device needed): builds the tiles and the stripe-cut pieces
// exactly as upload does and (1) verifies the structural invariants the kernel relies on (aligned passes
// within the chunk, ascending rows with a head flag on every row's first entry, staged columns inside the
// window, passes walking the columns upwards), (2) adds an integer checksum per entry into its row's
// accumulator and compares every row with the checksum of its entries taken straight from the CSR arrays.  stats (optional, 6 values per
// plan kind): blocks, passes, entries in tiles, entries in staged passes, rows left to the split-row kernels, widest window.
template <typename T>
static int tile_plan_check(int M, int N, const int *rp, const int *col, int rows_per_block, int lmax, int density,
                           int chunk, int balance, bool pack, long long *stats) {
    const long long nz = rp[M];
    std::vector<T> val((size_t)nz);
    for (long long e = 0; e < nz; ++e) val[(size_t)e] = (T)(1 + e % 7);
    TilePlan<T> plan;
    std::vector<int> row_len((size_t)M);
    for (int r = 0; r < M; ++r) row_len[(size_t)r] = rp[r + 1] - rp[r];
    if (!tile_build<T>(M, N, rp, row_len.data(), col, val.data(), rows_per_block, lmax, density, chunk, balance != 0, 17, plan, pack, 0, pack ? 256 : 0))
        return fail("tile_plan_check: the plan does not fit 32-bit entry offsets");
    // the remainder (packed plans: entries of windows too sparse for a pass): by (row, column), rows of the tiles only
    std::vector<unsigned long long> rem_sum((size_t)M, 0ull);
    if (plan.rem_row.size() != plan.rem_col.size() || plan.rem_row.size() != plan.rem_val.size() || (!pack && !plan.rem_row.empty()))
        return fail("tile_plan_check: remainder arrays disagree");
    for (size_t k = 0; k < plan.rem_row.size(); ++k) {
        const int r = plan.rem_row[k], c = plan.rem_col[k];
        if ((unsigned)r >= (unsigned)M || (unsigned)c >= (unsigned)N || plan.split[(size_t)r])
            return fail("tile_plan_check: remainder entry %zu is out of range", k);
        if (k > 0 && (plan.rem_row[k - 1] > r || (plan.rem_row[k - 1] == r && plan.rem_col[k - 1] > c)))
            return fail("tile_plan_check: the remainder is not ordered by (row, column)");
        rem_sum[(size_t)r] += (unsigned long long)(c + 1) * 0x9E3779B97F4A7C15ull + (unsigned long long)(double)plan.rem_val[k];
    }
    const int win_cols = plan.win_cols;
    auto h = [](long long c, double v) { return (unsigned long long)(c + 1) * 0x9E3779B97F4A7C15ull + (unsigned long long)v; };
    if ((int)plan.block_pass.size() != plan.num_blocks + 1 || plan.block_pass.back() != (int)plan.pass_desc.size() ||
        (int)plan.block_row.size() != plan.num_blocks + 1 || plan.block_row[0] != 0 || plan.block_row.back() != M)
        return fail("tile_plan_check: block / pass tables disagree");
    long long seen_entries = 0;
    std::vector<unsigned long long> acc((size_t)rows_per_block);
    for (int b = 0; b < plan.num_blocks; ++b) {
        std::fill(acc.begin(), acc.end(), 0ull);
        const int r0 = plan.block_row[(size_t)b], nrows = plan.block_row[(size_t)b + 1] - r0;
        if (nrows <= 0 || nrows > rows_per_block) return fail("tile_plan_check: block %d holds %d rows", b, nrows);
        long long last_max_col = -1;
        for (int p = plan.block_pass[b]; p < plan.block_pass[b + 1]; ++p) {
            const int4 d = plan.pass_desc[p];
            const int count = d.y, wbase = d.z, wlen = d.w & kTileWlenMask;
            const bool packed = (d.w & kTilePassPacked) != 0;
            if (packed && !wlen) return fail("tile_plan_check: pass %d is packed without a window", p);
            // a packed plan is for the kernel without gather code: every pass staged and packed, no key array
            if (packed != pack) return fail("tile_plan_check: pass %d of a %s plan is %s", p, pack ? "packed" : "plain", packed ? "packed" : "not");
            constexpr int kPer = 16 / (int)sizeof(T);
            // (a block without entries carries one pass of none)
            const bool none = count == 0 && plan.block_pass[b + 1] - plan.block_pass[b] == 1;
            if ((count <= 0 && !none) || count > chunk || (d.x & 3) || (wbase & 3) || (wlen % kPer) || wlen < 0)
                return fail("tile_plan_check: pass %d is malformed", p);
            if ((size_t)d.x + (size_t)count > plan.tcol.size() - kTileChunkMax) return fail("tile_plan_check: pass %d leaves the arrays", p);
            if (wlen && (wlen > win_cols || wbase + wlen > (N + kPer - 1) / kPer * kPer))
                return fail("tile_plan_check: window of pass %d is too wide or leaves x", p);
            int prev_row = -1;
            long long cmin = 1LL << 40, cmax = -1;
            for (int i = 0; i < count; ++i) {
                // a pass that can be staged holds packed words; decode as the kernel does
                const unsigned word = (unsigned)plan.tcol[(size_t)d.x + i];
                const int c = packed ? wbase + (int)(word & kTilePackColMask) : (int)word;
                const unsigned key = packed ? ((word >> 16) & (unsigned)kTileHead) | ((word >> kTilePackShift) & (unsigned)kTileRowMask)
                                            : plan.tkey[(size_t)d.x + i];
                const int lrow = (int)(key & kTileRowMask);
                if ((unsigned)c >= (unsigned)N || lrow >= nrows) return fail("tile_plan_check: entry %d of pass %d is out of range", i, p);
                if (wlen && (c < wbase || c >= wbase + wlen)) return fail("tile_plan_check: staged pass %d misses column %d", p, c);
                if (!wlen && c < wbase) return fail("tile_plan_check: pass %d has a column below its base", p);
                const bool head = (key & kTileHead) != 0;
                if (head != (lrow != prev_row)) return fail("tile_plan_check: head flag of entry %d in pass %d is wrong", i, p);
                if (lrow < prev_row) return fail("tile_plan_check: rows of pass %d are not ascending", p);
                if (key & ~(unsigned)(kTileHead | kTileRowMask)) return fail("tile_plan_check: stray key bits in pass %d", p);
                prev_row = lrow;
                cmin = std::min<long long>(cmin, c);
                cmax = std::max<long long>(cmax, c);
                // what the kernel does with the entry: every run goes to its row's accumulator exactly once
                acc[(size_t)lrow] += h(c, (double)plan.tval[(size_t)d.x + i]);
            }
            // passes of a block walk the columns upwards (that is what keeps the gathered band in L2)
            if (cmin < last_max_col) return fail("tile_plan_check: passes of block %d do not ascend in columns", b);
            last_max_col = cmax;
            seen_entries += count;
        }
        for (int i = 0; i < nrows; ++i) {
            unsigned long long want = 0;
            if (!plan.split[(size_t)r0 + i])
                for (int e = rp[r0 + i]; e < rp[r0 + i + 1]; ++e) want += h(col[e], (double)val[(size_t)e]);
            if (acc[(size_t)i] + rem_sum[(size_t)r0 + i] != want) return fail("tile_plan_check: row %d does not add up through the tiles", r0 + i);
        }
    }
    if (seen_entries != plan.entries) return fail("tile_plan_check: entry count disagrees");
    // streams (what a workgroup walks): every block exactly once, its passes in order, the flag on its last pass
    for (int places : {8, 64, 512, 1 << 30}) {
        tile_make_streams(plan, places);
        if (plan.spass.size() != plan.pass_desc.size() || (int)plan.sblock_rows.size() != plan.num_blocks ||
            (int)plan.stream_pass.size() != plan.num_streams + 1 || (int)plan.stream_block.size() != plan.num_streams + 1 ||
            (plan.num_streams & 7) || (plan.num_blocks > places && plan.num_streams > std::max(8, places / 8 * 8)))
            return fail("tile_plan_check: stream tables disagree (%d places)", places);
        std::vector<unsigned char> seen_block((size_t)plan.num_blocks, 0);
        for (int st = 0; st < plan.num_streams; ++st) {
            int p = plan.stream_pass[(size_t)st];
            for (int k = plan.stream_block[(size_t)st]; k < plan.stream_block[(size_t)st + 1]; ++k) {
                const int2 br = plan.sblock_rows[(size_t)k];
                const int b = (int)(std::upper_bound(plan.block_row.begin(), plan.block_row.end() - 1, br.x) - plan.block_row.begin()) - 1;
                if (b < 0 || b >= plan.num_blocks || plan.block_row[(size_t)b] != br.x || plan.block_row[(size_t)b + 1] - br.x != br.y ||
                    seen_block[(size_t)b]++)
                    return fail("tile_plan_check: stream %d lists a block twice or with the wrong rows", st);
                for (int q = plan.block_pass[(size_t)b]; q < plan.block_pass[(size_t)b + 1]; ++q, ++p) {
                    const int4 a = plan.pass_desc[(size_t)q], c = plan.spass[(size_t)p];
                    const bool last = q + 1 == plan.block_pass[(size_t)b + 1];
                    if (a.x != c.x || a.y != c.y || a.z != c.z || (a.w | (last ? kTilePassLast : 0)) != c.w)
                        return fail("tile_plan_check: stream %d does not repeat block %d's passes", st, b);
                }
            }
            if (p != plan.stream_pass[(size_t)st + 1]) return fail("tile_plan_check: stream %d's passes and blocks disagree", st);
        }
        for (unsigned char c : seen_block)
            if (c != 1) return fail("tile_plan_check: a block is in no stream");
    }
    // rows beyond the limit: stripe-cut pieces cover each exactly once, slots row by row
    std::vector<int4> pieces, long_rows;
    const int stripe_cols = (1 << 20) / (int)sizeof(T);
    build_striped_pieces(M, rp, col, plan.split, stripe_cols, pieces, long_rows);
    std::vector<const int4 *> by_slot(pieces.size(), nullptr);
    long long split_entries = 0;
    int last_stripe = -1;
    for (const int4 &pc : pieces) {
        if (pc.w < 0 || (size_t)pc.w >= pieces.size() || by_slot[(size_t)pc.w]) return fail("tile_plan_check: piece slots are not a permutation");
        by_slot[(size_t)pc.w] = &pc;
        if (pc.z <= pc.y || pc.z - pc.y > kLongPiece) return fail("tile_plan_check: bad piece of row %d", pc.x);
        const int stripe = col[pc.y] / stripe_cols;
        if (stripe < last_stripe) return fail("tile_plan_check: pieces are not ordered by stripe");
        last_stripe = stripe;
    }
    int n_split = 0;
    for (const int4 &l : long_rows) {
        if (!plan.split[(size_t)l.x]) return fail("tile_plan_check: row %d is split without being long", l.x);
        int at = rp[l.x];
        for (int k = 0; k < l.z; ++k) {
            const int4 *pc = by_slot[(size_t)l.y + k];
            if (!pc || pc->x != l.x || pc->y != at) return fail("tile_plan_check: pieces of row %d are not contiguous in slot order", l.x);
            at = pc->z;
        }
        if (at != rp[l.x + 1]) return fail("tile_plan_check: pieces of row %d do not cover it", l.x);
        split_entries += rp[l.x + 1] - rp[l.x];
        ++n_split;
    }
    for (int r = 0; r < M; ++r) n_split -= plan.split[(size_t)r];
    if (n_split != 0) return fail("tile_plan_check: split rows and piece lists disagree");
    if (split_entries + plan.entries + (long long)plan.rem_row.size() != nz) return fail("tile_plan_check: tiles + remainder + split rows do not hold every entry");
    if (stats) {
        stats[0] = plan.num_blocks;
        stats[1] = (long long)plan.pass_desc.size();
        stats[2] = plan.entries + (long long)plan.rem_row.size();  // (packed: the remainder counts as held, not as staged)
        stats[3] = plan.staged_entries;
        stats[4] = (long long)long_rows.size();
        stats[5] = plan.max_win;
    }
    return 0;
}

extern "C" int spmv_hip_csr_tile_plan_check(int M, int N, const int *row_ptr, const int *col_idx, int value_bytes,
                                            int rows_per_block, int lmax, int density, int chunk, int balance,
                                            long long *stats) {
    if (M < 0 || N < 0 || !row_ptr || (value_bytes != 4 && value_bytes != 8)) return fail("tile_plan_check: bad arguments");
    if (rows_per_block < 256 || rows_per_block > kTileRowsMax || (rows_per_block & 255) || lmax < 1 ||
        lmax > 65536 || density < 1 || chunk != 2048)
        return fail("tile_plan_check: bad plan parameters");
    const long long nz = row_ptr[M];
    if (nz > 0 && !col_idx) return fail("tile_plan_check: col_idx is NULL");
    for (int r = 0; r < M; ++r)
        if (row_ptr[r + 1] < row_ptr[r]) return fail("tile_plan_check: row_ptr decreases at row %d", r);
    for (long long e = 0; e < nz; ++e)
        if ((unsigned)col_idx[e] >= (unsigned)N) return fail("tile_plan_check: column %d outside [0, %d)", col_idx[e], N);
    // both kinds of plan: with gather passes (the stats are this one's) and packed (every pass staged)
    return guarded("tile_plan_check", [&] {
        for (int pack = 0; pack < 2; ++pack) {
            long long *st = pack ? (stats ? stats + 6 : nullptr) : stats;
            const int rc = value_bytes == 8 ? tile_plan_check<double>(M, N, row_ptr, col_idx, rows_per_block, lmax, density, chunk, balance, pack != 0, st)
                                            : tile_plan_check<float>(M, N, row_ptr, col_idx, rows_per_block, lmax, density, chunk, balance, pack != 0, st);
            if (rc) return rc;
        }
        return 0;
    });
}

// What upload would decide for this structure (host only, current tunings): tile_plan_all's choices.
template <typename T>
static int tile_auto_plan(int M, int N, const int *rp, const int *col, long long *stats) {
    const long long nz = rp[M];
    std::vector<T> val((size_t)nz, T(1));
    std::vector<int> row_len((size_t)M);
    for (int r = 0; r < M; ++r) row_len[(size_t)r] = rp[r + 1] - rp[r];
    TileBuild<T> tb;
    tile_plan_all<T>(M, N, rp, row_len.data(), rp, nz, col, val.data(), tb);
    for (int k = 0; k < 10; ++k) stats[k] = 0;
    stats[0] = tb.have_tiles;
    if (!tb.have_tiles) return 0;
    int tallest = 0;
    for (int b = 0; b < tb.tiles.num_blocks; ++b)
        tallest = std::max(tallest, tb.tiles.block_row[(size_t)b + 1] - tb.tiles.block_row[(size_t)b]);
    stats[1] = tb.packed;
    stats[2] = tb.scattered;
    stats[3] = tb.tiles.rows_per_block;
    stats[4] = tb.tiles.num_blocks;
    stats[5] = tb.tiles.num_streams;
    stats[6] = (long long)tb.tiles.pass_desc.size();
    stats[7] = tallest;
    stats[8] = tb.have_long_tiles ? (long long)tb.lt_work.size() : 0;
    stats[9] = tb.tiles.entries;
    return 0;
}

extern "C" int spmv_hip_csr_tile_auto_plan(int M, int N, const int *row_ptr, const int *col_idx, int value_bytes,
                                           long long *stats) {
    if (M < 0 || N < 0 || !row_ptr || !stats || (value_bytes != 4 && value_bytes != 8)) return fail("tile_auto_plan: bad arguments");
    const long long nz = row_ptr[M];
    if (nz > 0 && !col_idx) return fail("tile_auto_plan: col_idx is NULL");
    for (int r = 0; r < M; ++r)
        if (row_ptr[r + 1] < row_ptr[r]) return fail("tile_auto_plan: row_ptr decreases at row %d", r);
    for (long long e = 0; e < nz; ++e)
        if ((unsigned)col_idx[e] >= (unsigned)N) return fail("tile_auto_plan: column %d outside [0, %d)", col_idx[e], N);
    return guarded("tile_auto_plan", [&] {
        return value_bytes == 8 ? tile_auto_plan<double>(M, N, row_ptr, col_idx, stats) : tile_auto_plan<float>(M, N, row_ptr, col_idx, stats);
    });
}

// a whole fp64 matrix whose col / val already sit on the device (spmv_coo.hip)
int csr_adopt_f64(int M, int N, const int *row_ptr_host, int *d_col, double *d_val, spmv_csr_dev **out) {
    return guarded("csr_adopt", [&] { return csr_upload_impl<double>(M, N, row_ptr_host, nullptr, nullptr, 0, M, out, d_col, d_val); });
}

// A handle that carries NOTHING but tile plans, for rows given as (first entry, length) pairs over host arrays:
// how an HLL slab whose columns are too scattered for the x-window plan gets csr_tile (spmv_hll.hip).  The rows
// are rows [row0, row0 + M_local) of a matrix with M_total rows; launched with the caller's x and full-length y.
// *out stays NULL (return 0) when the rows get no plan.
int csr_tiles_from_rows_f64(int M_local, int M_total, int row0, int N, const int *row_begin, const int *row_len,
                            long long entries, const int *col, const double *val, spmv_csr_dev **out,
                            const int *d_col, const double *d_val) {
    *out = nullptr;
    if (M_local <= 0 || entries <= 0 || g_stream_tile == 0) return 0;
    if (g_stream_tile < 0 && (long long)M_local < kTileMinRows && entries < kTileMidEntries) return 0;
    return guarded("hll tile plan", [&] {
        TileBuild<double> tb;
        // d_col / d_val: the same slab on the device -- the plan is then built there (the rows' begin / length lists go up)
        TileDevInput<double> din;
        int *d_begin = nullptr, *d_len = nullptr;
        bool on_dev = g_tile_plan_on_device != 0 && d_col && d_val;
        if (on_dev && (upload_array(&d_begin, row_begin, (size_t)M_local, 1) || upload_array(&d_len, row_len, (size_t)M_local, 1)))
            on_dev = false;
        if (on_dev) {
            din.row_begin = d_begin;
            din.row_len = d_len;
            din.col = d_col;
            din.val = d_val;
            din.stream = g_stream;
        }
        tile_plan_all<double>(M_local, N, row_begin, row_len, nullptr, entries, col, val, tb, on_dev ? &din : nullptr);
        (void)hipFree(d_begin);
        (void)hipFree(d_len);
        if (!tb.have_tiles) return 0;
        spmv_csr_dev *m = new (std::nothrow) spmv_csr_dev();
        if (!m) return fail("hll tile plan: out of host memory");
        m->value_bytes = 8;
        m->M_local = M_local;
        m->M_total = M_total;
        m->N = N;
        m->row0 = row0;
        m->nz = entries;
        m->tiles_only = true;
        m->auto_variant = SPMV_CSR_STREAM;
        const int trc = tile_upload_all<double>(m, tb);
        if (trc) {  // 1: no tiles on this device (the slab keeps hll_lds), -1: a real failure
            spmv_hip_csr_free(m);
            return trc < 0 ? -1 : 0;
        }
        *out = m;
        return 0;
    });
}

static void release_relocated(spmv_csr_dev::relocated &r) {
    if (!r.raw) return;
    if (r.vmm) {
        (void)hipMemUnmap(r.raw, r.mapped);
        (void)hipMemAddressFree(r.raw, r.mapped);
        (void)hipMemRelease(r.phys);
    } else {
        (void)hipFree(r.raw);
    }
    r.raw = nullptr;
}

extern "C" int spmv_hip_csr_upload(int M, int N, const int *row_ptr, const int *col_idx,
                                   const double *values, int row0, int row1, spmv_csr_dev **out) {
    return guarded("csr_upload", [&] { return csr_upload_impl<double>(M, N, row_ptr, col_idx, values, row0, row1, out); });
}

extern "C" int spmv_hip_csr_upload_f32(int M, int N, const int *row_ptr, const int *col_idx,
                                       const float *values, int row0, int row1, spmv_csr_dev **out) {
    return guarded("csr_upload_f32", [&] { return csr_upload_impl<float>(M, N, row_ptr, col_idx, values, row0, row1, out); });
}

extern "C" int spmv_hip_csr_upload_matrix(const CSRMatrix *csr, spmv_csr_dev **out) {
    if (!csr) return fail("csr_upload_matrix: csr is NULL");
    return spmv_hip_csr_upload(csr->M, csr->N, csr->row_ptr, csr->col_idx, csr->values, 0, csr->M, out);
}

extern "C" void spmv_hip_csr_free(spmv_csr_dev *m) {
    if (!m) return;
    spmv_hip_csr_free(m->own_part);
    spmv_hip_csr_free(m->halo_part);
    for (auto &r : m->relocs) {  // relocated arrays: the field points into r.raw
        *r.field = nullptr;
        release_relocated(r);
    }
    (void)hipFree(m->row_ptr);
    (void)hipFree(m->col);
    (void)hipFree(m->val);
    (void)hipFree(m->desc);
    (void)hipFree(m->ldesc4);
    (void)hipFree(m->ldesc);
    (void)hipFree(m->lines);
    (void)hipFree(m->lcol);
    (void)hipFree(m->ptab);
    (void)hipFree(m->rinfo);
    (void)hipFree(m->pdesc);
    for (spmv_csr_dev::long_tiles *tier : {&m->mt}) {
        (void)hipFree(tier->block_row);
        (void)hipFree(tier->block_pass);
        (void)hipFree(tier->block_of_row);
        (void)hipFree(tier->item_first);
        (void)hipFree(tier->row_map);
        (void)hipFree(tier->pass);
        (void)hipFree(tier->work);
        (void)hipFree(tier->tcol);
        (void)hipFree(tier->tkey);
        (void)hipFree(tier->tval);
        (void)hipFree(tier->slab);
    }
    (void)hipFree(m->lt.block_row);
    (void)hipFree(m->lt.block_pass);
    (void)hipFree(m->lt.block_of_row);
    (void)hipFree(m->lt.item_first);
    (void)hipFree(m->lt.row_map);
    (void)hipFree(m->lt.pass);
    (void)hipFree(m->lt.work);
    (void)hipFree(m->lt.tcol);
    (void)hipFree(m->lt.tkey);
    (void)hipFree(m->lt.tval);
    (void)hipFree(m->lt.slab);
    (void)hipFree(m->interior_ids);
    (void)hipFree(m->boundary_ids);
    (void)hipFree(m->tile_block_pass);
    (void)hipFree(m->tile_block_row);
    (void)hipFree(m->tile_pass);
    (void)hipFree(m->tile_rem_row);
    (void)hipFree(m->tile_rem_ptr);
    (void)hipFree(m->tile_rem_col);
    (void)hipFree(m->tile_rem_val);
    (void)hipFree(m->tile_stream_block);
    (void)hipFree(m->tile_sblock_rows);
    (void)hipFree(m->tcol);
    (void)hipFree(m->tkey);
    (void)hipFree(m->tval);
    (void)hipFree(m->xe);
    delete m->expansion;
    (void)hipFree(m->tile_long_rows);
    (void)hipFree(m->tile_pieces);
    (void)hipFree(m->long_rows);
    (void)hipFree(m->pieces);
    (void)hipFree(m->partial);
    (void)hipFree(m->x);
    (void)hipFree(m->y);
    delete m;
}

// Placement study / placement rule: move one array of the handle to an address of the form
// (multiple of `align`) + offset.  which: 0 row_ptr, 1 col, 2 val, 3 x, 4 y, 5 lcol, 6 lines, 7 ldesc4.
// vmm != 0: the new home is built with the virtual-memory API instead of hipMalloc -- physical memory from
// hipMemCreate (one handle, size rounded to the recommended granularity), a virtual range reserved with the asked
// alignment, mapped and made accessible: VA alignment is then ours to choose, the physical side is whatever the
// driver's allocator gives a request of that size.
static int relocate_impl(spmv_csr_dev *m, int which, unsigned long long align, unsigned long long offset, int vmm) {
    if (need_device()) return -1;
    if (!m || which < 0 || which > 7) return fail("csr_relocate: bad arguments");
    if (align < 256 || (align & (align - 1)) || (offset & 255) || offset >= align)
        return fail("csr_relocate: align must be a power of two >= 256, offset a multiple of 256 below it");
    void **fields[8] = {(void **)&m->row_ptr, (void **)&m->col, &m->val, &m->x, &m->y, (void **)&m->lcol,
                        (void **)&m->lines, (void **)&m->ldesc4};
    void **field = fields[which];
    if (!*field) return 0;  // the handle has no such array
    HIP_TRY(hipStreamSynchronize(g_stream));
    size_t size = 0;
    size_t at = m->relocs.size();
    for (size_t k = 0; k < m->relocs.size(); ++k)
        if (m->relocs[k].field == field) at = k;
    if (at < m->relocs.size()) size = m->relocs[at].size;
    else HIP_TRY(hipMemPtrGetInfo(*field, &size));
    spmv_csr_dev::relocated fresh;
    fresh.field = field;
    fresh.size = size;
    char *p = nullptr;
    if (!vmm) {
        HIP_TRY(hipMalloc(&fresh.raw, size + align + offset));
        p = (char *)(((uintptr_t)fresh.raw + align - 1) & ~(uintptr_t)(align - 1)) + offset;
    } else {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = g_device;
        size_t gran = 0;
        HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        if (gran == 0) gran = 2u << 20;
        fresh.vmm = true;
        fresh.mapped = (size + offset + gran - 1) / gran * gran;
        HIP_TRY(hipMemCreate(&fresh.phys, fresh.mapped, &prop, 0));
        hipError_t e = hipMemAddressReserve(&fresh.raw, fresh.mapped, align, nullptr, 0);
        if (e == hipSuccess) {
            e = hipMemMap(fresh.raw, fresh.mapped, 0, fresh.phys, 0);
            if (e == hipSuccess) {
                hipMemAccessDesc acc = {};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                e = hipMemSetAccess(fresh.raw, fresh.mapped, &acc, 1);
                if (e != hipSuccess) (void)hipMemUnmap(fresh.raw, fresh.mapped);
            }
            if (e != hipSuccess) (void)hipMemAddressFree(fresh.raw, fresh.mapped);
        }
        if (e != hipSuccess) {
            (void)hipMemRelease(fresh.phys);
            return fail("csr_relocate: virtual-memory allocation failed: %s", hipGetErrorString(e));
        }
        if ((uintptr_t)fresh.raw & (align - 1)) {
            release_relocated(fresh);
            return fail("csr_relocate: the reserved range is not aligned to %llu", align);
        }
        p = (char *)fresh.raw + offset;
    }
    hipError_t e = hipMemcpy(p, *field, size, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) {
        release_relocated(fresh);
        return fail("csr_relocate: copy failed: %s", hipGetErrorString(e));
    }
    if (at < m->relocs.size()) release_relocated(m->relocs[at]);
    else (void)hipFree(*field);
    *field = p;
    if (at < m->relocs.size()) m->relocs[at] = fresh;
    else m->relocs.push_back(fresh);
    return 0;
}

extern "C" int spmv_hip_csr_relocate(spmv_csr_dev *m, int which, unsigned long long align, unsigned long long offset) {
    return guarded("csr_relocate", [&] { return relocate_impl(m, which, align, offset, 0); });
}
extern "C" int spmv_hip_csr_relocate_vmm(spmv_csr_dev *m, int which, unsigned long long align, unsigned long long offset) {
    return guarded("csr_relocate_vmm", [&] { return relocate_impl(m, which, align, offset, 1); });
}

// Measurement only: one launch of the x-window kernel's STAMP instantiation (same code + three stores per workgroup)
// behind `warm` ordinary launches; stamps[3 * b] = {start, end (ticks of the constant 100 MHz clock), dispatch id << 8 |
// XCD} of block b.  Shows WHERE a slow launch loses its time: every block a little, or one XCD's share as a tail.
extern "C" int spmv_hip_csr_stamp_blocks(spmv_csr_dev *m, int warm, unsigned long long *stamps_host) {
    if (need_device()) return -1;
    if (!m || !stamps_host) return fail("csr_stamp_blocks: NULL argument");
    if (m->local_blocks <= 0 || m->value_bytes != 8 || m->local_cap != 2048)
        return fail("csr_stamp_blocks: needs an fp64 handle with an x-window plan at the 2048-entry stage");
    unsigned long long *d = nullptr;
    const size_t bytes = (size_t)m->local_blocks * 3 * sizeof(unsigned long long);
    HIP_TRY(hipMalloc((void **)&d, bytes));
    int rc = 0;
    // (warm >= 1000, measurement only: the stamped launch does not read the slot stream -- the time the kernel would take
    // if the slots came from a per-block pattern; y is then wrong)
    const int probe = warm >= 1000 ? 1 : 0;
    warm %= 1000;
    for (int i = 0; i < warm && !rc; ++i) rc = csr_launch_any(m, SPMV_CSR_STREAM, m->x, m->y, g_stream);
    if (!rc) {
        const int lcount = m->local_blocks;
        const int lchunk = g_stream_xcd < 0 ? (lcount + 7) / 8 : (g_stream_xcd ? g_stream_xcd : 16);
        const int lgrid = (lcount + 8 * lchunk - 1) / (8 * lchunk) * (8 * lchunk);
        const size_t lds = std::max((size_t)m->local_cap * sizeof(double), (size_t)m->local_stage_lines * kLineBytes);
        const bool lnt = g_local_nt < 0 ? m->nz * 10LL > (128LL << 20) : g_local_nt != 0;
        double *y = (double *)m->y + m->row0;
        if (lnt)
            hipLaunchKernelGGL((csr_stream_local<double, true, 2048, true>), dim3(lgrid), dim3(kBlock), lds, g_stream, lcount, lchunk,
                               (const int *)nullptr, m->ldesc4, m->ldesc, m->lines, m->row_ptr, m->lcol, (const double *)m->val,
                               (const double *)m->x, y, d, probe);
        else
            hipLaunchKernelGGL((csr_stream_local<double, false, 2048, true>), dim3(lgrid), dim3(kBlock), lds, g_stream, lcount, lchunk,
                               (const int *)nullptr, m->ldesc4, m->ldesc, m->lines, m->row_ptr, m->lcol, (const double *)m->val,
                               (const double *)m->x, y, d, probe);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        if (e == hipSuccess) e = hipMemcpy(stamps_host, d, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail("csr_stamp_blocks: %s", hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}

// Digest of every array of the handle's tile plans (tests: a plan built on the device against one built on the host):
// out[2 k] = elements, out[2 k + 1] = FNV-1a of the bytes of array k, in the order tcol, tkey, tval, pass descriptors
// (stream order), stream_pass, block_row, stream_block, sblock_rows, rem_row, rem_ptr, rem_col, rem_val, then the long
// rows' plan: tcol, tkey, tval, pass, block_row, block_pass, work, item_first, row_map, block_of_row, and the same ten
// for the middle tier.  32 arrays.
int csr_tile_digest(const spmv_csr_dev *m, unsigned long long *out) {
    if (need_device()) return -1;
    if (!m || !out) return fail("csr_tile_digest: NULL argument");
    HIP_TRY(hipStreamSynchronize(g_stream));
    const size_t vb = (size_t)m->value_bytes;
    struct Arr { const void *p; size_t count, elem; };
    const size_t tpad = m->tile_blocks > 0 ? (size_t)m->tile_padded + kTileChunkMax : 0;
    std::vector<Arr> arrs = {
        {m->tcol, tpad, 4}, {m->tkey, m->tile_packed ? (tpad ? (size_t)kTileChunkMax : 0) : tpad, 2}, {m->tval, tpad, vb},
        {m->tile_pass, (size_t)m->tile_passes, 16}, {m->tile_block_pass, m->tile_blocks > 0 ? (size_t)m->tile_streams + 1 : 0, 4},
        {m->tile_block_row, m->tile_blocks > 0 ? (size_t)m->tile_blocks + 1 : 0, 4},
        {m->tile_stream_block, m->tile_blocks > 0 ? (size_t)m->tile_streams + 1 : 0, 4}, {m->tile_sblock_rows, (size_t)m->tile_blocks, 8},
        {m->tile_rem_row, (size_t)m->tile_rem_rows, 4}, {m->tile_rem_ptr, m->tile_rem_rows > 0 ? (size_t)m->tile_rem_rows + 1 : 0, 4},
        {m->tile_rem_col, (size_t)m->tile_rem_entries, 4}, {m->tile_rem_val, (size_t)m->tile_rem_entries, vb}};
    for (const spmv_csr_dev::long_tiles *tier : {&m->lt, &m->mt}) {
        const auto &L = *tier;
        const size_t lpad = L.items > 0 ? (size_t)L.padded + kTileChunkMax : 0;
        const Arr more[10] = {{L.tcol, lpad, 4}, {L.tkey, L.packed ? (lpad ? (size_t)kTileChunkMax : 0) : lpad, 2}, {L.tval, lpad, vb},
                              {L.pass, (size_t)L.passes, 16}, {L.block_row, L.items > 0 ? (size_t)L.blocks + 1 : 0, 4},
                              {L.block_pass, L.items > 0 ? (size_t)L.blocks + 1 : 0, 4}, {L.work, (size_t)L.items, 16},
                              {L.item_first, L.items > 0 ? (size_t)L.blocks + 1 : 0, 4}, {L.row_map, (size_t)L.rows, 4},
                              {L.block_of_row, (size_t)L.rows, 4}};
        arrs.insert(arrs.end(), more, more + 10);
    }
    std::vector<unsigned char> buf;
    for (int k = 0; k < 32; ++k) {
        const size_t bytes = arrs[k].p ? arrs[k].count * arrs[k].elem : 0;
        unsigned long long h = 1469598103934665603ull;
        if (bytes) {
            buf.resize(bytes);
            HIP_TRY(hipMemcpy(buf.data(), arrs[k].p, bytes, hipMemcpyDeviceToHost));
            // eight interleaved lanes of FNV-1a (one serial chain would take seconds per GB), folded at the end
            unsigned long long lane[8];
            for (int j = 0; j < 8; ++j) lane[j] = 1469598103934665603ull + (unsigned long long)j;
            size_t i = 0;
            for (; i + 8 <= bytes; i += 8)
                for (int j = 0; j < 8; ++j) lane[j] = (lane[j] ^ buf[i + j]) * 1099511628211ull;
            for (; i < bytes; ++i) lane[0] = (lane[0] ^ buf[i]) * 1099511628211ull;
            for (int j = 0; j < 8; ++j) h = (h ^ lane[j]) * 1099511628211ull;
        }
        out[2 * k] = arrs[k].p ? arrs[k].count : 0;
        out[2 * k + 1] = h;
    }
    return 0;
}

extern "C" int spmv_hip_csr_tile_digest(const spmv_csr_dev *m, unsigned long long *out) {
    return guarded("csr_tile_digest", [&] { return csr_tile_digest(m, out); });
}

extern "C" int spmv_hip_csr_addresses(const spmv_csr_dev *m, unsigned long long *out) {
    if (!m || !out) return fail("csr_addresses: NULL argument");
    const void *p[8] = {m->row_ptr, m->col, m->val, m->x, m->y, m->lcol, m->lines, m->ldesc4};
    for (int i = 0; i < 8; ++i) out[i] = (unsigned long long)(uintptr_t)p[i];
    return 0;
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
    out->tile_blocks = m->tile_blocks;
    out->tile_passes = m->tile_passes;
    out->tile_entries = m->tile_entries + m->tile_rem_entries;
    out->tile_remainder_entries = m->tile_rem_entries;
    out->tile_staged_entries = m->tile_staged;
    out->tile_split_rows = m->tile_num_long;
    out->tile_long_rows = m->lt.rows;
    out->tile_long_entries = m->lt.entries;
    out->tile_staged_cols = m->tile_staged_cols + m->lt.staged_cols + m->mt.staged_cols;
    out->tile_long_items = m->lt.items;
    out->tile_mid_rows = m->mt.rows;
    out->tile_mid_entries = m->mt.entries;
    out->tile_mid_items = m->mt.items;
    out->place_tries = m->place_tries;
    out->place_first_us = m->place_first_us;
    out->place_best_us = m->place_best_us;
    out->val_address = (unsigned long long)(uintptr_t)m->val;
    out->tile_expanded_entries = m->xe && m->expansion ? (long long)m->expansion->entries : 0;
    out->pattern_slots = m->ptab ? m->pat_slots : 0;
    out->pattern_with_us = m->pat_with_us;
    out->pattern_without_us = m->pat_without_us;
    out->stream_kernel = m->local_blocks > 0 ? 1 : m->tile_blocks > 0 ? 3
                         : ((m->stream_cap == 4096 || m->stream_cap == 2048) && m->M_local > 0 &&
                            m->nz < (long long)m->M_local * (m->stream_cap / kBlock)) ? 2 : 0;
    if (m->local_blocks > 0)  // (a pattern plan: the tables and 4 bytes per row instead of 2 bytes per entry)
        out->stream_bytes = m->nz * vb + (m->ptab ? 2 * m->pat_slots + 4LL * m->M_local + 8LL * m->local_blocks : 2 * m->nz) +
                            4 * m->local_lines + 24LL * m->local_blocks + 4LL * (m->M_local + 1) + vb * m->M_local + vb * m->N;
    else if (m->tile_blocks > 0) {  // tiles: 4-byte column + 2-byte key + value per (padded) entry; rows beyond the limit as CSR
        out->stream_bytes = m->tile_padded * (vb + 6) - (m->tile_packed ? 2 : 0) * m->tile_staged + 16LL * m->tile_passes +
                            4LL * m->tile_blocks +
                            std::max<long long>(0, m->nz - m->tile_entries - m->tile_rem_entries - m->lt.entries - m->mt.entries) * (vb + 4) +
                            m->tile_rem_entries * (vb + 4) + 16LL * m->tile_num_pieces + vb * m->M_local + vb * m->N;
        // an expanded plan: tile_expand reads 2 bytes per entry (+ 4 per run and per 64 entries) and writes its x value,
        // csr_tile reads that value as its window and a packed column word where it read column and key; every chunk
        // copies its 32 KiB slice of x (out of L2 mostly)
        if (m->xe && m->expansion)
            out->stream_bytes += (long long)m->expansion->entries * (2 * vb) + 24LL * m->expansion->chunks +
                                 4LL * (long long)(m->expansion->runs + m->expansion->groups);
        for (const spmv_csr_dev::long_tiles *tier : {&m->lt, &m->mt})  // entries, descriptors, slabs written and read
            out->stream_bytes += tier->padded * (vb + 6) - (tier->packed ? 2 : 0) * tier->staged + 16LL * tier->passes +
                                 2 * vb * (long long)tier->items * tier->rows_per_block;
    }
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
int csr_launch(const spmv_csr_dev *m, int variant, const T *x, T *y_full, hipStream_t s, int part = -1) {
    if (m->M_local == 0) return 0;
    T *y = y_full + m->row0;
    if (variant == SPMV_CSR_AUTO) variant = m->auto_variant;
    if (m->tiles_only && variant != SPMV_CSR_STREAM) return fail("csr_launch: a tiles-only handle runs the tile kernel only");
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
            if (m->num_blocks > 0 || m->tiles_only) {
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
                // a PACKED plan copies its x slices as 16-byte pieces and has no gather code to fall back on: with an x
                // that is not 16-byte aligned the clamp of the last piece no longer keeps the loads inside x, so such a
                // launch goes to the gather kernels (a tiles-only handle has none: error)
                const bool x_ok_for_tiles = !m->tile_packed || ((uintptr_t)x & 15) == 0;
                const bool tiled = !local && m->tile_blocks > 0 && x_ok_for_tiles &&
                                   (g_stream_kind == -1 || g_stream_kind == 6 || m->tiles_only);
                if (m->tiles_only && !tiled)
                    return fail(x_ok_for_tiles ? "csr_launch: a tiles-only handle has nothing else to run"
                                               : "csr_launch: a packed tile plan needs a 16-byte aligned x");
                if (tiled) {
                    // staging copies 16-byte pieces of x: plans with gather passes fall back on gathering everything when
                    // x is not 16-byte aligned, packed plans (no gather code) load the pieces unaligned
                    const int stage_ok = ((uintptr_t)x & 15) == 0;
                    // (probe bit 3, measurement only: one workgroup per CU)
                    const size_t lds = std::max((size_t)std::max(m->tile_lds_min, g_tile_probe & 8 ? 84 * 1024 : 0),
                                                (size_t)kTileSlotBytes + (size_t)m->tile_rows * sizeof(T) +
                                                    (m->tile_packed ? (size_t)kTileTrips * kTileTripBytes  // (the packed kernel stores every trip's piece)
                                                                    : stage_ok ? (size_t)m->tile_max_win * sizeof(T) : 0));
                    const bool tnt = m->nz * (long long)(sizeof(T) + 6) > (128LL << 20);
                    const int which = tnt ? 1 : 0;
                    // (more than 64 KiB of dynamic LDS: allowed for these kernels once, at upload -- tile_allow_lds)
#define SPMV_TILE(NT, PACK)                                                                                            \
    hipLaunchKernelGGL((csr_tile<T, NT, 2048, kTileTrips, PACK>), dim3((m->tile_streams + 7) / 8 * 8), dim3(kTileBlock), lds, s, \
                       m->tile_streams, m->tile_rows, stage_ok, g_tile_probe, (const int4 *)nullptr, (T *)nullptr,       \
                       m->tile_block_row, m->tile_block_pass, m->tile_pass, m->tcol, m->tkey, (const T *)m->tval,        \
                       m->tile_stream_block, m->tile_sblock_rows, x, y)
#define SPMV_TILE_GA(NT)                                                                                               \
    hipLaunchKernelGGL((csr_tile<T, NT, 2048, kTileTrips, false, true>), dim3((m->tile_streams + 7) / 8 * 8), dim3(kTileBlock), lds, s, \
                       m->tile_streams, m->tile_rows, stage_ok, g_tile_probe, (const int4 *)nullptr, (T *)nullptr,       \
                       m->tile_block_row, m->tile_block_pass, m->tile_pass, m->tcol, m->tkey, (const T *)m->tval,        \
                       m->tile_stream_block, m->tile_sblock_rows, x, y)
                    // an expanded plan: x into the passes' segments first, then the packed kernel over x' (16-byte pieces of
                    // x: aligned x only)
                    const bool expanded = m->xe && m->expansion && m->expansion->chunks > 0 && stage_ok && g_tile_expand != 0;
                    if (expanded) {
                        hipLaunchKernelGGL((tile_expand<T>), dim3(m->expansion->chunks), dim3(kExpandBlock), 0, s, m->N, g_tile_probe,
                                           m->expansion->chunk, m->expansion->chunk_runs, m->expansion->lcol, m->expansion->group_run, m->expansion->delta, x, (T *)m->xe);
                        // (a pass's window is its own segment: 2048 values at most, one staging trip in fp32, two in fp64)
                        constexpr int kXpTrips = 2048 * (int)sizeof(T) / kTileTripBytes;
                        const size_t xlds = std::max((size_t)m->tile_lds_min, (size_t)kTileSlotBytes + (size_t)m->tile_rows * sizeof(T) +
                                                                                  (size_t)kXpTrips * kTileTripBytes);
#define SPMV_TILE_XP(NT)                                                                                               \
    hipLaunchKernelGGL((csr_tile<T, NT, 2048, kXpTrips, true>), dim3((m->tile_streams + 7) / 8 * 8), dim3(kTileBlock), xlds, s, \
                       m->tile_streams, m->tile_rows, 1, g_tile_probe, (const int4 *)nullptr, (T *)nullptr,              \
                       m->tile_block_row, m->tile_block_pass, m->expansion->pass, m->expansion->words,                   \
                       (const unsigned short *)nullptr, (const T *)m->tval, m->tile_stream_block, m->tile_sblock_rows,   \
                       (const T *)m->xe, y)
                        if (which) SPMV_TILE_XP(true); else SPMV_TILE_XP(false);
#undef SPMV_TILE_XP
                    }
                    else if (m->tile_packed) { if (which) SPMV_TILE(true, true); else SPMV_TILE(false, true); }
                    else if (g_tile_gather_ahead) { if (which) SPMV_TILE_GA(true); else SPMV_TILE_GA(false); }  // gathers one pass early
                    else { if (which) SPMV_TILE(true, false); else SPMV_TILE(false, false); }
#undef SPMV_TILE_GA
#undef SPMV_TILE
                    if (m->tile_rem_rows > 0)  // what the packed plan left out: added behind the tiles
                        hipLaunchKernelGGL((tile_remainder<T>), dim3((m->tile_rem_rows + 255) / 256), dim3(256), 0, s, m->tile_rem_rows,
                                           m->tile_rem_row, m->tile_rem_ptr, m->tile_rem_col, (const T *)m->tile_rem_val, x, y);
                    // the compacted tiers -- the long rows' own tiles, the middle tier of a scattered matrix: work items ->
                    // slabs -> y (after the ordinary tiles wrote 0 there)
                    for (const spmv_csr_dev::long_tiles *tier : {&m->lt, &m->mt}) {
                        const auto &L = *tier;
                        if (L.items <= 0) continue;
                        const size_t llds = (size_t)kTileSlotBytes + (size_t)L.rows_per_block * sizeof(T) +
                                            (L.packed ? (size_t)kTileTrips * kTileTripBytes : stage_ok ? (size_t)L.max_win * sizeof(T) : 0);
#define SPMV_LTILE(NT, PACK)                                                                                           \
    hipLaunchKernelGGL((csr_tile<T, NT, 2048, kTileTrips, PACK>), dim3((L.items + 7) / 8 * 8), dim3(kTileBlock), llds, s, L.items, \
                       L.rows_per_block, stage_ok, g_tile_probe, (const int4 *)L.work, (T *)L.slab, L.block_row,        \
                       L.block_pass, L.pass, L.tcol, L.tkey, (const T *)L.tval, (const int *)nullptr, (const int2 *)nullptr, x, y)
                        if (L.packed) { if (which) SPMV_LTILE(true, true); else SPMV_LTILE(false, true); }
                        else { if (which) SPMV_LTILE(true, false); else SPMV_LTILE(false, false); }
#undef SPMV_LTILE
                        hipLaunchKernelGGL((tile_slab_finish<T>), dim3((L.rows + kFinishRows - 1) / kFinishRows),
                                           dim3(kFinishRows * kFinishGroups), 0, s, L.rows,
                                           L.rows_per_block, L.block_row, L.block_of_row, L.item_first, L.row_map,
                                           (const T *)L.slab, y);
                    }
                    if (m->tile_num_long) {  // rows beyond the tile limit: stripe-ordered pieces, slots added row by row
                        hipLaunchKernelGGL((csr_long_pieces<T, true>), dim3(m->tile_num_pieces), dim3(kBlock), 0, s,
                                           m->tile_num_pieces, m->tile_pieces, m->col, (const T *)m->val, x, (T *)m->partial);
                        hipLaunchKernelGGL((csr_long_finish<T>), dim3(m->tile_num_long), dim3(64), 0, s, m->tile_num_long,
                                           m->tile_long_rows, (const T *)m->partial, y);
                    }
                    HIP_TRY(hipGetLastError());
                    return 0;
                }
                if (local) {
                    // runs of 16 neighbouring blocks per XCD: each L2 keeps its own window of x lines
                    // (measured flat from 8 to 128 on three matrices); stream_xcd overrides
                    // part < 0: all blocks; 0 / 1: the interior / boundary sub-list (csr_launch_part)
                    const int lcount = part < 0 ? m->local_blocks : part == 0 ? m->num_interior : m->num_boundary;
                    const int *lids = part < 0 ? nullptr : part == 0 ? m->interior_ids : m->boundary_ids;
                    const int lchunk = g_stream_xcd < 0 ? (lcount + 7) / 8 : (g_stream_xcd ? g_stream_xcd : 16);
                    const int lgrid = lchunk > 0 ? (lcount + 8 * lchunk - 1) / (8 * lchunk) * (8 * lchunk) : lcount;
                    if (lcount > 0) {
                    const size_t lds = std::max((size_t)m->local_cap * sizeof(T), (size_t)m->local_stage_lines * kLineBytes);
                    // a pattern plan: the slots are rebuilt in LDS (behind the stage) from the block's pattern table, not read
                    // entry by entry
                    const bool patterns = m->ptab && m->rinfo && m->pdesc && g_local_patterns != 0;
                    const size_t pat_lds = lds + ((size_t)m->local_cap + 8) * sizeof(unsigned short);
#define SPMV_LOCAL(NT, CAP)                                                                                   \
    do {                                                                                                      \
        if (patterns)                                                                                         \
            hipLaunchKernelGGL((csr_stream_local<T, NT, CAP, false, true>), dim3(lgrid), dim3(kBlock), pat_lds, s, lcount, lchunk, \
                               lids, m->ldesc4, m->ldesc, m->lines, m->row_ptr, m->lcol, (const T *)m->val, x, y, \
                               (unsigned long long *)nullptr, 0, m->pdesc, m->rinfo, m->ptab, (int)lds);       \
        else                                                                                                  \
            hipLaunchKernelGGL((csr_stream_local<T, NT, CAP>), dim3(lgrid), dim3(kBlock), lds, s, lcount, lchunk, lids, \
                               m->ldesc4, m->ldesc, m->lines, m->row_ptr, m->lcol, (const T *)m->val, x, y);   \
    } while (0)
                    // streamed-once hint only when the matrix cannot live in the 256 MiB Infinity Cache anyway
                    // (cant-like, 53 MB: 10.9 us without it, 11.7 us with; fem-large: 160 vs 151 us)
                    const bool lnt = g_local_nt < 0 ? m->nz * (long long)(sizeof(T) + 2) > (128LL << 20) : g_local_nt != 0;
                    if (m->local_cap == 1024) { if (lnt) SPMV_LOCAL(true, 1024); else SPMV_LOCAL(false, 1024); }
                    else if (m->local_cap == 3072) { if (lnt) SPMV_LOCAL(true, 3072); else SPMV_LOCAL(false, 3072); }
                    else { if (lnt) SPMV_LOCAL(true, 2048); else SPMV_LOCAL(false, 2048); }
                    }
#undef SPMV_LOCAL
#ifdef SPMV_EXPERIMENTAL
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
#endif
                } else if (cap == 1024) {
                    SPMV_LAUNCH_FLAGS(SPMV_LAUNCH_PROD, 1024, 256);
                } else if (cap == 3072) {
                    SPMV_LAUNCH_FLAGS(SPMV_LAUNCH_PROD, 3072, 256);
                } else if (cap == 2048 && m->nz < (long long)m->M_local * (2048 / kBlock)) {
                    if (g_stream_nt) hipLaunchKernelGGL((csr_stream_short<T, true, 2048, 256>), dim3(grid_blocks), dim3(256), 0, s, m->num_blocks, chunk, SPMV_ARGS);
                    else hipLaunchKernelGGL((csr_stream_short<T, false, 2048, 256>), dim3(grid_blocks), dim3(256), 0, s, m->num_blocks, chunk, SPMV_ARGS);
                } else if (cap == 2048) {
                    SPMV_LAUNCH_FLAGS(SPMV_LAUNCH_PROD, 2048, 256);
                } else if (cap == 4096 && blk == 512) {
                    SPMV_LAUNCH_FLAGS(SPMV_LAUNCH_PROD, 4096, 512);
                } else if (cap == 4096 && m->nz < (long long)m->M_local * (4096 / kBlock)) {
                    // short rows: blocks of up to 1024 rows, row extents of all passes loaded up front
                    if (g_stream_nt) hipLaunchKernelGGL((csr_stream_short<T, true, 4096, 256>), dim3(grid_blocks), dim3(256), 0, s, m->num_blocks, chunk, SPMV_ARGS);
                    else hipLaunchKernelGGL((csr_stream_short<T, false, 4096, 256>), dim3(grid_blocks), dim3(256), 0, s, m->num_blocks, chunk, SPMV_ARGS);
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
            if (m->num_long && part != 0) {
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

}  // namespace

int csr_launch_any(const spmv_csr_dev *m, int variant, const void *x, void *y, hipStream_t s) {
    if (m->value_bytes == 8) return csr_launch<double>(m, variant, (const double *)x, (double *)y, s);
    return csr_launch<float>(m, variant, (const float *)x, (float *)y, s);
}

// N4 overlap: the interior x-window blocks (part 0: rows whose x lines all lie in the handle's own range, so
// they can run while the halo of x is still travelling) / everything else (part 1).  Together = one
// csr_launch_any(STREAM): the same kernels on the same blocks, hence the same bits.  A handle without the
// split (no x-window plan, foreign misaligned x) runs everything as part 1.
int csr_launch_part(const spmv_csr_dev *m, int part, const void *x, void *y, hipStream_t s) {
    const bool split_ok = m->have_split && m->local_blocks > 0 && ((uintptr_t)x & (kLineBytes - 1)) == 0 &&
                          (g_stream_kind == -1 || g_stream_kind == 5);
    if (!split_ok) return part == 0 ? 0 : csr_launch_any(m, SPMV_CSR_STREAM, x, y, s);
    if (m->value_bytes == 8) return csr_launch<double>(m, SPMV_CSR_STREAM, (const double *)x, (double *)y, s, part);
    return csr_launch<float>(m, SPMV_CSR_STREAM, (const float *)x, (float *)y, s, part);
}

// Which x-window blocks of the handle are interior: every x line they list lies inside the handle's own rows'
// range of x (square matrix, x owned like y: entries [row0, row0 + M_local)).  Computed from the plan's line
// lists (ascending inside a block: first and last line decide).  counts (optional, 4 values): interior blocks,
// boundary blocks, entries in interior blocks, entries in boundary blocks + split rows.
static int csr_split_interior_body(spmv_csr_dev *m, long long *counts) {
    if (need_device()) return -1;
    if (!m) return fail("csr_split_interior: NULL handle");
    if (m->M_total != m->N) return fail("csr_split_interior: needs a square matrix (%d x %d)", m->M_total, m->N);
    (void)hipFree(m->interior_ids);
    (void)hipFree(m->boundary_ids);
    m->interior_ids = m->boundary_ids = nullptr;
    m->num_interior = m->num_boundary = 0;
    m->have_split = false;
    long long e_in = 0, e_out = m->nz;
    if (m->local_blocks > 0) {
        const int B = m->local_blocks;
        std::vector<int2> ld((size_t)B);
        std::vector<int4> d4((size_t)B);
        std::vector<int> lines((size_t)m->local_lines);
        HIP_TRY(hipStreamSynchronize(g_stream));
        HIP_TRY(hipMemcpy(ld.data(), m->ldesc, ld.size() * sizeof(int2), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(d4.data(), m->ldesc4, d4.size() * sizeof(int4), hipMemcpyDeviceToHost));
        if (!lines.empty()) HIP_TRY(hipMemcpy(lines.data(), m->lines, lines.size() * sizeof(int), hipMemcpyDeviceToHost));
        const int per_line = kLineBytes / m->value_bytes;
        const long long own_lo = m->row0, own_hi = (long long)m->row0 + m->M_local;  // [lo, hi) of x
        std::vector<int> in_ids, out_ids;
        for (int b = 0; b < B; ++b) {
            const long long first = (long long)lines[(size_t)ld[b].x] * per_line;
            const long long last = ((long long)lines[(size_t)ld[b].x + ld[b].y - 1] + 1) * per_line;  // exclusive
            // (a block of empty rows lists line 0 only to have something to stage: it reads nothing)
            const bool empty = d4[b].w == d4[b].y;
            if (empty || (first >= own_lo && std::min<long long>(last, m->N) <= own_hi)) {
                in_ids.push_back(b);
                e_in += d4[b].w - d4[b].y;
            } else {
                out_ids.push_back(b);
            }
        }
        e_out = m->nz - e_in;
        if (upload_array(&m->interior_ids, in_ids.data(), in_ids.size(), 1)) return -1;
        if (upload_array(&m->boundary_ids, out_ids.data(), out_ids.size(), 1)) return -1;
        m->num_interior = (int)in_ids.size();
        m->num_boundary = (int)out_ids.size();
        m->have_split = true;
    }
    if (counts) {
        counts[0] = m->num_interior;
        counts[1] = m->num_boundary;
        counts[2] = e_in;
        counts[3] = e_out;
    }
    return 0;
}

extern "C" int spmv_hip_csr_split_interior(spmv_csr_dev *m, long long *counts) {
    return guarded("csr_split_interior", [&] { return csr_split_interior_body(m, counts); });
}

// ---- N4 overlap below block granularity: the column split
namespace {

template <typename T>
__global__ __launch_bounds__(kBlock) void add_rows(long long n, const T *__restrict__ t, T *__restrict__ y) {
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < n; k += (long long)gridDim.x * kBlock) y[k] += t[k];
}

template <typename T>
int csr_split_columns_impl(spmv_csr_dev *m, int col_lo, int col_hi, long long *counts) {
    spmv_hip_csr_free(m->own_part);
    spmv_hip_csr_free(m->halo_part);
    m->own_part = m->halo_part = nullptr;
    m->own_entries = m->halo_entries = 0;
    const int Ml = m->M_local;
    const size_t nz = (size_t)m->nz;
    std::vector<int> rp((size_t)Ml + 1), col(nz);
    std::vector<T> val(nz);
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(rp.data(), m->row_ptr, rp.size() * sizeof(int), hipMemcpyDeviceToHost));
    if (nz) HIP_TRY(hipMemcpy(col.data(), m->col, nz * sizeof(int), hipMemcpyDeviceToHost));
    if (nz) HIP_TRY(hipMemcpy(val.data(), m->val, nz * sizeof(T), hipMemcpyDeviceToHost));
    // two CSR matrices over the same rows: entries keep their order inside a row
    std::vector<int> rp_own((size_t)m->M_total + 1, 0), rp_halo((size_t)m->M_total + 1, 0), c_own, c_halo;
    std::vector<T> v_own, v_halo;
    c_own.reserve(nz);
    v_own.reserve(nz);
    for (int r = 0; r < Ml; ++r) {
        for (int e = rp[(size_t)r]; e < rp[(size_t)r + 1]; ++e) {
            if (col[(size_t)e] >= col_lo && col[(size_t)e] < col_hi) {
                c_own.push_back(col[(size_t)e]);
                v_own.push_back(val[(size_t)e]);
            } else {
                c_halo.push_back(col[(size_t)e]);
                v_halo.push_back(val[(size_t)e]);
            }
        }
        rp_own[(size_t)m->row0 + r + 1] = (int)c_own.size();
        rp_halo[(size_t)m->row0 + r + 1] = (int)c_halo.size();
    }
    for (int r = m->row0 + Ml; r < m->M_total; ++r) {  // rows behind the handle's block: empty
        rp_own[(size_t)r + 1] = (int)c_own.size();
        rp_halo[(size_t)r + 1] = (int)c_halo.size();
    }
    m->own_entries = (long long)c_own.size();
    m->halo_entries = (long long)c_halo.size();
    if (counts) {
        counts[0] = m->own_entries;
        counts[1] = m->halo_entries;
    }
    if (c_halo.empty()) return 0;  // nothing comes from other ranks: the handle itself is the interior (same bits as ever)
    // (the sub-handles are ordinary handles: plans, block cuts, AUTO -- but no placement search of their own)
    const int keep = g_place_tries;
    g_place_tries = 0;
    int rc = csr_upload_impl<T>(m->M_total, m->N, rp_own.data(), c_own.data(), v_own.data(), m->row0, m->row0 + Ml, &m->own_part);
    if (!rc) rc = csr_upload_impl<T>(m->M_total, m->N, rp_halo.data(), c_halo.data(), v_halo.data(), m->row0, m->row0 + Ml, &m->halo_part);
    g_place_tries = keep;
    if (rc) {
        spmv_hip_csr_free(m->own_part);
        spmv_hip_csr_free(m->halo_part);
        m->own_part = m->halo_part = nullptr;
        return -1;
    }
    return 0;
}

}  // namespace

// Split the handle's entries by column: [col_lo, col_hi) = the rank's own range of x.  counts[2] (optional): entries
// whose column lies inside / outside.  Afterwards spmv_hip_csr_run_split(part 0) computes y = A_own x -- it reads
// nothing of x outside [col_lo, col_hi) -- and part 1 adds A_halo x; 0 then 1 = the handle's product up to the
// order in which a row's two partial sums are added.  A handle without outside entries keeps no sub-handles.
extern "C" int spmv_hip_csr_split_columns(spmv_csr_dev *m, int col_lo, int col_hi, long long *counts) {
    if (need_device()) return -1;
    if (!m || col_lo < 0 || col_hi < col_lo || col_hi > m->N) return fail("csr_split_columns: bad arguments");
    if (m->tiles_only || !m->col || !m->val) return fail("csr_split_columns: the handle does not hold its CSR arrays");
    return guarded("csr_split_columns", [&] {
        return m->value_bytes == 8 ? csr_split_columns_impl<double>(m, col_lo, col_hi, counts)
                                   : csr_split_columns_impl<float>(m, col_lo, col_hi, counts);
    });
}

int csr_launch_split(const spmv_csr_dev *m, int part, const void *x, void *y, hipStream_t s) {
    if (!m->own_part || !m->halo_part) {  // no split (nothing comes from outside): everything is part 0
        return part == 0 ? csr_launch_any(m, SPMV_CSR_AUTO, x, y, s) : 0;
    }
    if (part == 0) return csr_launch_any(m->own_part, SPMV_CSR_AUTO, x, y, s);
    // y_own += A_halo x: the product into the sub-handle's own y, then one pass over this rank's rows
    if (csr_launch_any(m->halo_part, SPMV_CSR_AUTO, x, m->halo_part->y, s)) return -1;
    const long long n = m->M_local;
    const int grid = (int)std::max<long long>(1, std::min<long long>(2048, (n + kBlock - 1) / kBlock));
    if (m->value_bytes == 8)
        hipLaunchKernelGGL((add_rows<double>), dim3(grid), dim3(kBlock), 0, s, n, (const double *)m->halo_part->y + m->row0, (double *)y + m->row0);
    else
        hipLaunchKernelGGL((add_rows<float>), dim3(grid), dim3(kBlock), 0, s, n, (const float *)m->halo_part->y + m->row0, (float *)y + m->row0);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int spmv_hip_csr_run_split(spmv_csr_dev *m, int part, const void *d_x, void *d_y, void *stream) {
    if (need_device()) return -1;
    if (!m || (part != 0 && part != 1)) return fail("csr_run_split: bad arguments");
    return csr_launch_split(m, part, d_x ? d_x : m->x, d_y ? d_y : m->y, stream ? (hipStream_t)stream : g_stream);
}

extern "C" int spmv_hip_csr_run_part(spmv_csr_dev *m, int part, const void *d_x, void *d_y, void *stream) {
    if (need_device()) return -1;
    if (!m || (part != 0 && part != 1)) return fail("csr_run_part: bad arguments");
    return csr_launch_part(m, part, d_x ? d_x : m->x, d_y ? d_y : m->y, stream ? (hipStream_t)stream : g_stream);
}



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

extern "C" int spmv_hip_csr_time_graph(spmv_csr_dev *m, int variant, int iters, int replays, float *ms_per_iter) {
    if (need_device()) return -1;
    if (!m) return fail("csr_time_graph: NULL handle");
    return graph_loop(iters, replays, ms_per_iter, [&] { return csr_launch_any(m, variant, m->x, m->y, g_stream); });
}

