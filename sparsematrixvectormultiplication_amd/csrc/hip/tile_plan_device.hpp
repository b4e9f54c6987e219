// tile_plan_device.hpp -- the csr_tile plan built ON THE DEVICE (format builders on device).
//
// tile_build (tile_plan.hpp) on host threads costs 0.5-3 s for 36-262 M entries: it keys, sorts and re-orders every
// entry of the matrix, several host copies of it.  The same plan, byte for byte, from the CSR arrays that are already
// in HBM (the reference's one-off costs this replaces: src/csr_matrix.c:63-126, main_cuda.cu:381-400):
//
//   1. tile_fill_keys      every entry of a row that stays in the tiles -> column << 32 | local row << pos_bits | position
//                          (the host builder's key), block after block
//   2. rocPRIM             one segmented radix sort of those keys, one segment per row block: ascending columns,
//                          CSR order among equal columns (the keys are unique, so the result is the host's std::sort)
//   3. tile_cut_passes     one workgroup per row block walks its sorted keys and makes the host builder's greedy cuts --
//                          window-limited, density rule, remainder rule -- counting the lanes' share of each candidate
//                          range in parallel; run twice: count (passes, remainder entries, padded entries per block),
//                          then, with the prefix sums in between, fill (pass descriptors, where each pass's keys are)
//   4. tile_emit_pass      one workgroup per pass: the pass's <= 2048 keys sorted by (local row, position) in LDS
//                          (bitonic), head flags, packed column words / keys, values; padding to a multiple of four
//   5. remainder           the entries of windows too sparse for a pass: a second segmented sort by (row, column,
//                          position) and one kernel
//
// Only the small arrays (row blocks, per-block counts, pass descriptors, the remainder's rows) visit the host.
// tests/test_gpu_tile.py compares digests of every array of a handle built this way with one built by tile_plan.hpp.
#pragma once
#include <rocprim/rocprim.hpp>

#include <memory>

#include "tile_plan.hpp"

namespace spmv {

// the big arrays of a device-built plan; freed with the last owner -- a handle that adopts one takes the pointer and
// leaves nullptr behind
template <typename T>
struct TileDevArrays {
    int *tcol = nullptr;
    unsigned short *tkey = nullptr;
    T *tval = nullptr;
    size_t tcol_count = 0, tkey_count = 0;  // elements incl. the kTileChunkMax zeros behind the last pass
    int *rem_col = nullptr;
    T *rem_val = nullptr;
    size_t rem_count = 0;
    ~TileDevArrays() {
        (void)hipFree(tcol);
        (void)hipFree(tkey);
        (void)hipFree(tval);
        (void)hipFree(rem_col);
        (void)hipFree(rem_val);
    }
};

// where a device build reads the matrix from
template <typename T>
struct TileDevInput {
    const int *row_begin = nullptr, *row_len = nullptr;  // device, per row of the plan's row space
    const int *col = nullptr;                            // device
    const T *val = nullptr;
    hipStream_t stream = nullptr;
};

namespace tile_dev {

struct CutParams {
    int chunk, win_cols, density, min_pass, pos_bits, col_top, kper, pack;
};

constexpr int kCutBlock = 256;

// slice blockIdx.y of block blockIdx.x: one wavefront per row, lanes over the row's entries.
// PAIRS = false: keys[k] = column << 32 | local row << pos_bits | position (one segment per block for the segmented sort);
// PAIRS = true:  keys[k] = block << 32 | column, payload[k] = local row << pos_bits | position (one global stable sort)
template <bool PAIRS>
__global__ __launch_bounds__(256) void tile_fill_keys(int B, const int *__restrict__ block_row, const int *__restrict__ row_begin,
                                                      const int *__restrict__ row_len, const int *__restrict__ koff,
                                                      const int *__restrict__ col, int lmax, int pos_bits,
                                                      unsigned long long *__restrict__ keys, unsigned *__restrict__ payload) {
    const int b = blockIdx.x;
    if (b >= B) return;
    const int r0 = block_row[b], r1 = block_row[b + 1];
    const int slices = gridDim.y, per = (r1 - r0 + slices - 1) / slices;
    const int s0 = r0 + (int)blockIdx.y * per, s1 = min(r1, s0 + per);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = s0 + wave; r < s1; r += 4) {
        const int len = row_len[r];
        if (len > lmax) continue;
        const int beg = row_begin[r], o = koff[r];
        const unsigned hi = (unsigned)(r - r0) << pos_bits;
        for (int k = lane; k < len; k += 64) {
            const unsigned c = (unsigned)col[(size_t)beg + k];
            if (PAIRS) {
                keys[(size_t)o + k] = ((unsigned long long)(unsigned)b << 32) | c;
                payload[(size_t)o + k] = hi | (unsigned)k;
            } else {
                keys[(size_t)o + k] = ((unsigned long long)c << 32) | hi | (unsigned)k;
            }
        }
    }
}

// (block << 32 | column, payload) -> column << 32 | payload: what the cut and emit kernels read
__global__ __launch_bounds__(256) void tile_repack_keys(size_t n, const unsigned long long *__restrict__ bc,
                                                        const unsigned *__restrict__ payload, unsigned long long *__restrict__ keys) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256)
        keys[k] = (bc[k] << 32) | payload[k];
}

// sum of `v` over the workgroup, the same value in every thread (v is a per-wave count already)
__device__ __forceinline__ int cut_block_sum(int wave_count, int *red) {
    const int t = threadIdx.x;
    __syncthreads();  // red is free again
    if ((t & 63) == 0) red[t >> 6] = wave_count;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// The host builder's greedy cuts (tile_plan.hpp, build_range) over the sorted keys of block b.
// COUNT: n_pass / n_rem / ent_len [b].  FILL: pass_desc / pass_src at pass_off[b], remainder keys at rem_off[b].
template <bool FILL>
__global__ __launch_bounds__(kCutBlock) void tile_cut_passes(int B, const int *__restrict__ kb,
                                                             const unsigned long long *__restrict__ keys, CutParams P,
                                                             int *__restrict__ n_pass, int *__restrict__ n_rem,
                                                             int *__restrict__ ent_len, const int *__restrict__ pass_off,
                                                             const int *__restrict__ rem_off, const int *__restrict__ ent_off,
                                                             int4 *__restrict__ pass_desc, int2 *__restrict__ pass_src,
                                                             unsigned long long *__restrict__ rem_keys, int *__restrict__ failed,
                                                             int4 *__restrict__ rec = nullptr, const int *__restrict__ rec_off = nullptr,
                                                             int *__restrict__ n_rec = nullptr) {
    __shared__ int red[4];
    const int b = blockIdx.x;
    if (b >= B) return;
    const int t = threadIdx.x;
    const int key0 = kb[b], n = kb[b + 1] - key0;
    const unsigned long long *K = keys + key0;
    int passes = 0, rems = 0, ent = 0;  // the same in every thread
    // COUNT with rec: every step of the walk is written down -- a pass {i, j, entries before it, its ordinal}, a window
    // that goes to the remainder {i, w, remainder entries before it, -1} -- so that the descriptors can be filled in
    // without walking the block a second time (tile_fill_from_records); a block with more steps than its share of rec
    // holds reports that (n_rec[b] = -1) and the caller walks again
    int steps = 0;
    const int rec_base = (!FILL && rec) ? rec_off[b] : 0, rec_cap = (!FILL && rec) ? rec_off[b + 1] - rec_off[b] : 0;
    auto record = [&](int i, int j, int before, int ordinal) {
        if (!FILL && rec) {
            if (steps < rec_cap && t == 0) rec[rec_base + steps] = int4{i, j, before, ordinal};
            ++steps;
        }
    };
    const int pbase = FILL ? pass_off[b] : 0, rbase = FILL ? rem_off[b] : 0, ebase = FILL ? ent_off[b] : 0;
    auto col_of = [&](int idx) { return (int)(K[idx] >> 32); };
    auto emit = [&](int i, int j) {
        const int count = j - i, cmin = col_of(i), cmax = col_of(j - 1);
        const int wbase = cmin & ~3;
        int wlen = ((cmax - wbase + 1) + 3) & ~3;
        wlen = min(wlen, (P.col_top - wbase + P.kper - 1) / P.kper * P.kper);
        const bool staged = wlen <= P.win_cols && wlen <= (int)kTilePackColMask + 1 &&
                            (P.pack || (long long)count * P.density >= wlen);
        if (P.pack && !staged && t == 0) atomicOr(failed, 1);  // (cannot happen: every cut is window-limited)
        if (FILL && t == 0) {
            pass_desc[pbase + passes] = int4{ebase + ent, count, wbase, staged ? (wlen | (P.pack ? kTilePassPacked : 0)) : 0};
            pass_src[pbase + passes] = int2{key0 + i, b};
        }
        record(i, j, ent, passes);
        ++passes;
        ent += (count + 3) & ~3;
    };
    if (n > 0) {
        bool one_pass = n <= P.chunk;
        if (one_pass && P.pack) one_pass = (long long)col_of(n - 1) - (long long)(col_of(0) & ~3) < P.win_cols;
        if (one_pass) {
            emit(0, n);
        } else {
            int i = 0;
            while (i < n) {
                const long long base = (long long)col_of(i) & ~3LL;
                const int cap = min(n, i + P.chunk);
                const long long limit = base + P.win_cols;
                // sorted ascending: the entries below the window's end are a prefix of [i, cap)
                // (the candidate range's columns copied to LDS first -- one trip to memory per step instead of three
                // dependent ones -- was measured and is no faster: 54.7 against 49.9 ms for config 5's long rows; a step
                // costs its barriers, not its loads)
                int c = 0;
                for (int idx = i + t; idx < i + P.chunk; idx += kCutBlock) {  // (the same trip count for all lanes)
                    const bool in = idx < cap && (long long)col_of(min(idx, cap - 1)) < limit;
                    c += __popcll(__ballot(in));
                }
                c = cut_block_sum(c, red);
                const int w = i + c;
                const long long span = (long long)col_of(w - 1) - base + 1;
                if (P.pack && c < P.min_pass && (long long)c * 16 < span) {
                    // a window with a handful of entries far apart: to the remainder, keyed (local row, column, position)
                    if (FILL) {
                        const unsigned long long pos_mask = (1ull << P.pos_bits) - 1;
                        for (int idx = i + t; idx < w; idx += kCutBlock) {
                            const unsigned long long k = K[idx];
                            const unsigned long long lrow = (k & 0xffffffffull) >> P.pos_bits, pos = k & pos_mask;
                            rem_keys[(size_t)rbase + rems + (idx - i)] = (lrow << (32 + P.pos_bits)) | ((k >> 32) << P.pos_bits) | pos;
                        }
                    }
                    record(i, w, rems, -1);
                    rems += c;
                    i = w;
                    continue;
                }
                const int j = (P.pack || (long long)c * P.density >= span) ? w : cap;
                emit(i, j);
                i = j;
            }
        }
    }
    if (passes == 0) {  // a block without entries in passes: one pass of none, so that whoever walks it writes its zeros
        if (FILL && t == 0) {
            pass_desc[pbase] = int4{ebase + ent, 0, 0, P.pack ? (P.kper | kTilePassPacked) : 0};
            pass_src[pbase] = int2{key0, b};
        }
        passes = 1;
    }
    if (!FILL && t == 0) {
        n_pass[b] = passes;
        n_rem[b] = rems;
        ent_len[b] = ent;
        if (rec) n_rec[b] = steps <= rec_cap ? steps : -1;
    }
}

// What tile_cut_passes<true> writes, from the steps tile_cut_passes<false> wrote down: one workgroup per block, a
// thread per pass record; the remainder windows' keys copied by all threads, window after window.
__global__ __launch_bounds__(kCutBlock) void tile_fill_from_records(int B, const int *__restrict__ kb,
                                                                    const unsigned long long *__restrict__ keys, CutParams P,
                                                                    const int4 *__restrict__ rec, const int *__restrict__ rec_off,
                                                                    const int *__restrict__ n_rec, const int *__restrict__ pass_off,
                                                                    const int *__restrict__ rem_off, const int *__restrict__ ent_off,
                                                                    int4 *__restrict__ pass_desc, int2 *__restrict__ pass_src,
                                                                    unsigned long long *__restrict__ rem_keys) {
    const int b = blockIdx.x;
    if (b >= B) return;
    const int t = threadIdx.x;
    const int key0 = kb[b];
    const unsigned long long *K = keys + key0;
    const int pbase = pass_off[b], rbase = rem_off[b], ebase = ent_off[b];
    const int4 *R = rec + rec_off[b];
    const int steps = n_rec[b];
    bool any_pass = false;
    for (int r = t; r < steps; r += kCutBlock) {
        const int4 d = R[r];
        if (d.w < 0) continue;
        const int i = d.x, j = d.y, count = j - i, cmin = (int)(K[i] >> 32), cmax = (int)(K[j - 1] >> 32);
        const int wbase = cmin & ~3;
        int wlen = ((cmax - wbase + 1) + 3) & ~3;
        wlen = min(wlen, (P.col_top - wbase + P.kper - 1) / P.kper * P.kper);
        const bool staged = wlen <= P.win_cols && wlen <= (int)kTilePackColMask + 1 && (P.pack || (long long)count * P.density >= wlen);
        pass_desc[pbase + d.w] = int4{ebase + d.z, count, wbase, staged ? (wlen | (P.pack ? kTilePassPacked : 0)) : 0};
        pass_src[pbase + d.w] = int2{key0 + i, b};
        any_pass = true;
    }
    const unsigned long long pos_mask = (1ull << P.pos_bits) - 1;
    for (int r = 0; r < steps; ++r) {  // (the same records for every thread)
        const int4 d = R[r];
        if (d.w >= 0) continue;
        for (int idx = d.x + t; idx < d.y; idx += kCutBlock) {
            const unsigned long long k = K[idx];
            const unsigned long long lrow = (k & 0xffffffffull) >> P.pos_bits, pos = k & pos_mask;
            rem_keys[(size_t)rbase + d.z + (idx - d.x)] = (lrow << (32 + P.pos_bits)) | ((k >> 32) << P.pos_bits) | pos;
        }
    }
    // a block without entries in passes: one pass of none (ent_off[b + 1] - ent_off[b] = 0 then)
    if (__syncthreads_or(any_pass) == 0 && t == 0) {
        pass_desc[pbase] = int4{ebase, 0, 0, P.pack ? (P.kper | kTilePassPacked) : 0};
        pass_src[pbase] = int2{key0, b};
    }
}

// One workgroup per pass: keys -> (local row, position) order -> the plan's entry arrays.
template <typename T, bool PACK>
__global__ __launch_bounds__(256) void tile_emit_pass(int num_pass, const int4 *__restrict__ pass_desc,
                                                      const int2 *__restrict__ pass_src,
                                                      const unsigned long long *__restrict__ keys,
                                                      const int *__restrict__ block_row, const int *__restrict__ row_begin,
                                                      const int *__restrict__ col, const T *__restrict__ val, int pos_bits,
                                                      int *__restrict__ tcol, unsigned short *__restrict__ tkey,
                                                      T *__restrict__ tval) {
    __shared__ unsigned pl[2048];
    const int p = blockIdx.x;
    if (p >= num_pass) return;
    const int4 d = pass_desc[p];
    const int count = d.y;
    if (count == 0) return;
    const int2 src = pass_src[p];
    const int t = threadIdx.x;
    int size = 2;
    while (size < count) size <<= 1;  // <= 2048
    for (int k = t; k < size; k += 256) pl[k] = k < count ? (unsigned)keys[(size_t)src.x + k] : 0xffffffffu;
    for (int span = 2; span <= size; span <<= 1) {
        for (int stride = span >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int q = t; q < size / 2; q += 256) {
                const int i = ((q / stride) * stride << 1) + (q % stride);
                const int j = i + stride;
                const bool up = (i & span) == 0;
                const unsigned a = pl[i], c = pl[j];
                if ((a > c) == up) {
                    pl[i] = c;
                    pl[j] = a;
                }
            }
        }
    }
    __syncthreads();
    const int r0 = block_row[src.y];
    const unsigned pos_mask = (1u << pos_bits) - 1;
    const int padded = (count + 3) & ~3;
    for (int k = t; k < padded; k += 256) {
        if (k < count) {
            const unsigned w = pl[k];
            const unsigned lrow = w >> pos_bits, pos = w & pos_mask;
            const bool head = k == 0 || (pl[k - 1] >> pos_bits) != lrow;
            const size_t e = (size_t)row_begin[r0 + (int)lrow] + pos;
            const int c = col[e];
            if (PACK && d.w)
                tcol[(size_t)d.x + k] = (int)(((unsigned)head << 31) | (lrow << kTilePackShift) | (unsigned)(c - d.z));
            else
                tcol[(size_t)d.x + k] = c;
            tval[(size_t)d.x + k] = val[e];
            if (!PACK) tkey[(size_t)d.x + k] = (unsigned short)(lrow | (head ? (unsigned)kTileHead : 0u));
        } else {  // the next pass starts on a multiple of 4
            tcol[(size_t)d.x + k] = d.z;
            tval[(size_t)d.x + k] = T(0);
            if (!PACK) tkey[(size_t)d.x + k] = 0;
        }
    }
}

// sorted remainder keys of block b -> (row in the plan's row space, column, value)
template <typename T>
__global__ __launch_bounds__(256) void tile_emit_remainder(int B, const int *__restrict__ rem_off,
                                                           const unsigned long long *__restrict__ rem_keys,
                                                           const int *__restrict__ block_row, const int *__restrict__ row_begin,
                                                           const T *__restrict__ val, int pos_bits, int *__restrict__ rem_row,
                                                           int *__restrict__ rem_col, T *__restrict__ rem_val) {
    const int b = blockIdx.x;
    if (b >= B) return;
    const int r0 = block_row[b];
    const unsigned long long pos_mask = (1ull << pos_bits) - 1;
    for (int k = rem_off[b] + (int)threadIdx.x; k < rem_off[b + 1]; k += 256) {
        const unsigned long long key = rem_keys[k];
        const int lrow = (int)(key >> (32 + pos_bits)), c = (int)((key >> pos_bits) & 0xffffffffull), pos = (int)(key & pos_mask);
        const int r = r0 + lrow;
        rem_row[k] = r;
        rem_col[k] = c;
        rem_val[k] = val[(size_t)row_begin[r] + pos];
    }
}

inline unsigned bits_for(unsigned long long v) {  // bits needed to tell 0 .. v apart
    unsigned b = 1;
    while (b < 64 && (1ull << b) <= v) ++b;
    return b;
}

// frees on every path
struct Scratch {
    std::vector<void *> held;
    ~Scratch() {
        for (void *p : held) (void)hipFree(p);
    }
    template <typename U>
    hipError_t alloc(U **p, size_t count) {
        *p = nullptr;
        const hipError_t e = hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(U));
        if (e == hipSuccess) held.push_back(*p);
        return e;
    }
};

}  // namespace tile_dev

// tile_build (tile_plan.hpp) with the matrix read from the device.  h_row_len: the rows' lengths on the host (row
// blocks are cut there, as tile_build does).  1: plan built -- plan.dev holds the entry arrays, the small arrays are in
// the plan as usual (pass_desc, block_pass, rem_row, the sums); 0: the tiles would not hold the matrix; -1: HIP error
// (message through err).
template <typename T>
int tile_build_device(int M, int N, const TileDevInput<T> &in, const int *h_row_len, int rows_per_block, int lmax, int density,
                      int chunk, bool balance, int pos_bits, TilePlan<T> &plan, std::shared_ptr<TileDevArrays<T>> &dev, bool pack,
                      long long target_entries, int min_pass, std::string &err) {
    using namespace tile_dev;
    UploadTrace trace("tile_build_device");
    const int win_cols = kTileTrips * kTileTripBytes / (int)sizeof(T);
    plan = TilePlan<T>();
    dev.reset();
    plan.rows_per_block = rows_per_block;
    plan.chunk = chunk;
    plan.win_cols = win_cols;
    plan.split.assign((size_t)M, 0);
    long long in_tiles = 0;
    std::vector<int> koff((size_t)M + 1, 0);
    for (int r = 0; r < M; ++r) {
        plan.split[(size_t)r] = h_row_len[r] > lmax;
        if (!plan.split[(size_t)r]) in_tiles += h_row_len[r];
        if (in_tiles > 0x7ffffff0LL) return 0;
        koff[(size_t)r + 1] = (int)in_tiles;
    }
    const long long full_blocks = std::max(1, (M + rows_per_block - 1) / rows_per_block);
    const long long target = target_entries > 0 ? std::max<long long>(chunk, target_entries)
                             : balance          ? std::max<long long>(chunk, (in_tiles + full_blocks - 1) / full_blocks)
                                                : (1LL << 62);
    plan.block_row = tile_cut_rows(M, h_row_len, lmax, rows_per_block, target);
    plan.num_blocks = (int)plan.block_row.size() - 1;
    const int B = plan.num_blocks;
    if (B <= 0) {  // no rows: an empty plan, as tile_build leaves it
        plan.block_pass.assign(1, 0);
        dev = std::make_shared<TileDevArrays<T>>();
        return 1;
    }
    std::vector<int> kb((size_t)B + 1);
    for (int b = 0; b <= B; ++b) kb[(size_t)b] = koff[(size_t)plan.block_row[(size_t)b]];
    const size_t n_total = (size_t)in_tiles;
    hipStream_t s = in.stream;
    Scratch tmp;
    auto bad = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return false;
        err = std::string("tile plan on the device: ") + what + " failed: " + hipGetErrorString(e);
        return true;
    };
    int *d_block_row, *d_koff, *d_kb, *d_counts, *d_offs, *d_failed;
    unsigned long long *d_keys_a, *d_keys_b;
    hipError_t e = tmp.alloc(&d_block_row, (size_t)B + 1);
    if (e == hipSuccess) e = tmp.alloc(&d_koff, (size_t)M + 1);
    if (e == hipSuccess) e = tmp.alloc(&d_kb, (size_t)B + 1);
    if (e == hipSuccess) e = tmp.alloc(&d_counts, 3 * (size_t)B);
    if (e == hipSuccess) e = tmp.alloc(&d_offs, 3 * ((size_t)B + 1));
    if (e == hipSuccess) e = tmp.alloc(&d_failed, 1);
    if (e == hipSuccess) e = tmp.alloc(&d_keys_a, n_total);
    if (e == hipSuccess) e = tmp.alloc(&d_keys_b, n_total);
    if (e == hipSuccess) e = hipMemcpyAsync(d_block_row, plan.block_row.data(), ((size_t)B + 1) * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_koff, koff.data(), ((size_t)M + 1) * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_kb, kb.data(), ((size_t)B + 1) * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_failed, 0, sizeof(int), s);
    if (bad(e, "allocation")) return -1;
    trace.mark("row blocks, offsets, allocation");
    // 1. keys, block after block (few blocks: more slices each, so that the chip still has work)
    const int slices = B >= 2048 ? 1 : B >= 256 ? 8 : 64;
    // 2. sort.  Many blocks of moderate size: one segment per row block.  A FEW HUGE blocks (the long rows' plan: ten
    // blocks of 1.5e7 keys) are the segmented sort's worst case -- 0.49 s for 1.4e8 keys where the ordinary plan's
    // 1.2e8 keys in 1260 segments take 9 ms: those go through ONE stable radix sort of (block << 32 | column) with the
    // (local row, position) word as payload -- CSR order among equal (block, column), the same total order.
    unsigned long long *sorted = d_keys_a;
    const bool few_huge = n_total / (size_t)B >= ((size_t)1 << 20);
    if (few_huge) {
        unsigned *d_pay_a = nullptr, *d_pay_b = nullptr;
        e = tmp.alloc(&d_pay_a, n_total);
        if (e == hipSuccess) e = tmp.alloc(&d_pay_b, n_total);
        if (bad(e, "allocation")) return -1;
        hipLaunchKernelGGL((tile_fill_keys<true>), dim3(B, slices), dim3(256), 0, s, B, d_block_row, in.row_begin, in.row_len, d_koff,
                           in.col, lmax, pos_bits, d_keys_a, d_pay_a);
        trace.mark("keys");
        const unsigned end_bit = 32 + bits_for((unsigned long long)B);
        size_t tmp_bytes = 0;
        e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys_a, d_keys_b, d_pay_a, d_pay_b, n_total, 0u, end_bit, s);
        void *d_tmp = nullptr;
        if (e == hipSuccess) e = tmp.alloc((char **)&d_tmp, tmp_bytes);
        if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys_a, d_keys_b, d_pay_a, d_pay_b, n_total, 0u, end_bit, s);
        if (bad(e, "radix sort")) return -1;
        hipLaunchKernelGGL(tile_repack_keys, dim3(4096), dim3(256), 0, s, n_total, d_keys_b, d_pay_b, d_keys_a);
        sorted = d_keys_a;
    } else {
        hipLaunchKernelGGL((tile_fill_keys<false>), dim3(B, slices), dim3(256), 0, s, B, d_block_row, in.row_begin, in.row_len, d_koff,
                           in.col, lmax, pos_bits, d_keys_a, (unsigned *)nullptr);
        trace.mark("keys");
    }
    if (!few_huge && n_total > 0) {
        const unsigned end_bit = 32 + bits_for((unsigned long long)std::max(N, 1));
        size_t tmp_bytes = 0;
        e = rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, d_keys_a, d_keys_b, (unsigned)n_total, (unsigned)B, d_kb, d_kb + 1,
                                               0u, end_bit, s);
        void *d_tmp = nullptr;
        if (e == hipSuccess) e = tmp.alloc((char **)&d_tmp, tmp_bytes);
        if (e == hipSuccess)
            e = rocprim::segmented_radix_sort_keys(d_tmp, tmp_bytes, d_keys_a, d_keys_b, (unsigned)n_total, (unsigned)B, d_kb, d_kb + 1,
                                                   0u, end_bit, s);
        if (bad(e, "segmented sort")) return -1;
        sorted = d_keys_b;
    }
    trace.mark("segmented sort");
    // 3. cuts: count, prefix sums on the host (three ints per block), fill
    CutParams P{chunk, win_cols, density, pack ? min_pass : 0, pos_bits, std::max(N, 1), 16 / (int)sizeof(T), pack ? 1 : 0};
    // (the walk writes its steps down -- room for one step per 32 keys of a block and 64 more -- so that filling in the
    // descriptors needs no second walk: config 5's long rows 2 x 50 ms -> 50 ms)
    std::vector<int> rec_off((size_t)B + 1, 0);
    for (int b = 0; b < B; ++b) rec_off[(size_t)b + 1] = rec_off[(size_t)b] + (kb[(size_t)b + 1] - kb[(size_t)b]) / 32 + 64;
    int4 *d_rec = nullptr;
    int *d_rec_off = nullptr, *d_n_rec = nullptr;
    e = tmp.alloc(&d_rec, (size_t)rec_off[(size_t)B]);
    if (e == hipSuccess) e = tmp.alloc(&d_rec_off, (size_t)B + 1);
    if (e == hipSuccess) e = tmp.alloc(&d_n_rec, (size_t)B);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rec_off, rec_off.data(), rec_off.size() * sizeof(int), hipMemcpyHostToDevice, s);
    if (bad(e, "allocation")) return -1;
    hipLaunchKernelGGL((tile_cut_passes<false>), dim3(B), dim3(kCutBlock), 0, s, B, d_kb, sorted, P, d_counts, d_counts + B,
                       d_counts + 2 * (size_t)B, (const int *)nullptr, (const int *)nullptr, (const int *)nullptr, (int4 *)nullptr,
                       (int2 *)nullptr, (unsigned long long *)nullptr, d_failed, d_rec, (const int *)d_rec_off, d_n_rec);
    std::vector<int> counts(3 * (size_t)B), n_rec((size_t)B);
    int failed = 0;
    e = hipMemcpyAsync(counts.data(), d_counts, counts.size() * sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(n_rec.data(), d_n_rec, n_rec.size() * sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&failed, d_failed, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (bad(e, "counting the passes")) return -1;
    if (failed) return 0;
    trace.mark("cuts: count");
    std::vector<int> offs(3 * ((size_t)B + 1), 0);
    int *pass_off = offs.data(), *rem_off = pass_off + B + 1, *ent_off = rem_off + B + 1;
    long long total_ent = 0, total_rem = 0, total_pass = 0;
    for (int b = 0; b < B; ++b) {
        total_pass += counts[(size_t)b];
        total_rem += counts[(size_t)B + b];
        total_ent += counts[2 * (size_t)B + b];
        if (total_ent + kTileChunkMax >= 0x7fffffffLL || total_pass >= 0x7fffffffLL) return 0;
        pass_off[b + 1] = (int)total_pass;
        rem_off[b + 1] = (int)total_rem;
        ent_off[b + 1] = (int)total_ent;
    }
    auto arrays = std::make_shared<TileDevArrays<T>>();
    arrays->tcol_count = (size_t)total_ent + kTileChunkMax;
    arrays->tkey_count = pack ? (size_t)kTileChunkMax : arrays->tcol_count;
    arrays->rem_count = (size_t)total_rem;
    int4 *d_pass_desc;
    int2 *d_pass_src;
    unsigned long long *d_rem_a = nullptr, *d_rem_b = nullptr;
    int *d_rem_row = nullptr;
    e = tmp.alloc(&d_pass_desc, (size_t)total_pass);
    if (e == hipSuccess) e = tmp.alloc(&d_pass_src, (size_t)total_pass);
    if (e == hipSuccess) e = hipMalloc((void **)&arrays->tcol, arrays->tcol_count * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&arrays->tkey, arrays->tkey_count * sizeof(unsigned short));
    if (e == hipSuccess) e = hipMalloc((void **)&arrays->tval, arrays->tcol_count * sizeof(T));
    if (e == hipSuccess) e = hipMemsetAsync(arrays->tcol + total_ent, 0, (size_t)kTileChunkMax * sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(arrays->tkey + (pack ? 0 : total_ent), 0, (size_t)kTileChunkMax * sizeof(unsigned short), s);
    if (e == hipSuccess) e = hipMemsetAsync(arrays->tval + total_ent, 0, (size_t)kTileChunkMax * sizeof(T), s);
    if (e == hipSuccess && total_rem) {
        e = tmp.alloc(&d_rem_a, (size_t)total_rem);
        if (e == hipSuccess) e = tmp.alloc(&d_rem_b, (size_t)total_rem);
        if (e == hipSuccess) e = tmp.alloc(&d_rem_row, (size_t)total_rem);
        if (e == hipSuccess) e = hipMalloc((void **)&arrays->rem_col, (size_t)total_rem * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&arrays->rem_val, (size_t)total_rem * sizeof(T));
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_offs, offs.data(), offs.size() * sizeof(int), hipMemcpyHostToDevice, s);
    if (bad(e, "allocating the plan")) return -1;
    const int *d_pass_off = d_offs, *d_rem_off = d_offs + B + 1, *d_ent_off = d_offs + 2 * ((size_t)B + 1);
    bool all_recorded = true;
    for (int b = 0; b < B; ++b) all_recorded = all_recorded && n_rec[(size_t)b] >= 0;
    if (all_recorded)
        hipLaunchKernelGGL(tile_fill_from_records, dim3(B), dim3(kCutBlock), 0, s, B, d_kb, sorted, P, (const int4 *)d_rec,
                           (const int *)d_rec_off, (const int *)d_n_rec, d_pass_off, d_rem_off, d_ent_off, d_pass_desc, d_pass_src, d_rem_a);
    else  // (a block of passes of a few entries each: the second walk)
        hipLaunchKernelGGL((tile_cut_passes<true>), dim3(B), dim3(kCutBlock), 0, s, B, d_kb, sorted, P, (int *)nullptr, (int *)nullptr,
                           (int *)nullptr, d_pass_off, d_rem_off, d_ent_off, d_pass_desc, d_pass_src, d_rem_a, d_failed);
    trace.mark("cuts: fill");
    // 4. the passes' entries
    if (total_pass > 0) {
        if (pack)
            hipLaunchKernelGGL((tile_emit_pass<T, true>), dim3((unsigned)total_pass), dim3(256), 0, s, (int)total_pass, d_pass_desc,
                               d_pass_src, sorted, d_block_row, in.row_begin, in.col, in.val, pos_bits, arrays->tcol, arrays->tkey,
                               arrays->tval);
        else
            hipLaunchKernelGGL((tile_emit_pass<T, false>), dim3((unsigned)total_pass), dim3(256), 0, s, (int)total_pass, d_pass_desc,
                               d_pass_src, sorted, d_block_row, in.row_begin, in.col, in.val, pos_bits, arrays->tcol, arrays->tkey,
                               arrays->tval);
    }
    trace.mark("emit passes");
    // 5. the remainder: by (row, column, position) inside every block
    if (total_rem) {
        size_t tmp_bytes = 0;
        e = rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, d_rem_a, d_rem_b, (unsigned)total_rem, (unsigned)B, d_rem_off,
                                               d_rem_off + 1, 0u, 64u, s);
        void *d_tmp = nullptr;
        if (e == hipSuccess) e = tmp.alloc((char **)&d_tmp, tmp_bytes);
        if (e == hipSuccess)
            e = rocprim::segmented_radix_sort_keys(d_tmp, tmp_bytes, d_rem_a, d_rem_b, (unsigned)total_rem, (unsigned)B, d_rem_off,
                                                   d_rem_off + 1, 0u, 64u, s);
        if (bad(e, "sorting the remainder")) return -1;
        hipLaunchKernelGGL((tile_emit_remainder<T>), dim3(B), dim3(256), 0, s, B, d_rem_off, d_rem_b, d_block_row, in.row_begin, in.val,
                           pos_bits, d_rem_row, arrays->rem_col, arrays->rem_val);
        plan.rem_row.resize((size_t)total_rem);
        e = hipMemcpyAsync(plan.rem_row.data(), d_rem_row, (size_t)total_rem * sizeof(int), hipMemcpyDeviceToHost, s);
        if (bad(e, "remainder rows")) return -1;
    }
    plan.pass_desc.resize((size_t)total_pass);
    e = hipMemcpyAsync(plan.pass_desc.data(), d_pass_desc, (size_t)total_pass * sizeof(int4), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (bad(e, "building the plan")) return -1;
    trace.mark("remainder, descriptors back");
    plan.block_pass.assign(pass_off, pass_off + B + 1);
    for (const int4 &d : plan.pass_desc) {
        plan.entries += d.y;
        if (d.w && d.y) {  // (the pass of none of an empty block carries a window in a packed plan but holds nothing)
            const int wlen = d.w & kTileWlenMask;
            plan.staged_entries += d.y;
            plan.staged_cols += wlen;
            plan.max_win = std::max(plan.max_win, wlen);
        }
    }
    dev = arrays;
    return 1;
}

// ---- the expansion plan of a tile plan with gather passes (tile_expand + csr_tile<.., PACK> over x', tile_kernels.hpp)
namespace tile_dev {

// pass_of[e] = the pass (index in `pass`) whose region holds slot e: a pass owns [d.x, d.x + round_up4(d.y))
__global__ __launch_bounds__(256) void expand_pass_of(int passes, const int4 *__restrict__ pass, unsigned *__restrict__ pass_of) {
    const int p = blockIdx.x;
    if (p >= passes) return;
    const int4 d = pass[p];
    for (int i = threadIdx.x; i < ((d.y + 3) & ~3); i += 256) pass_of[d.x + i] = (unsigned)p;
}
// sort 1: key = first slot of the pass's region << slice_bits | slice of the entry's column (slots behind a pass's entries,
// and slots no pass owns: `slices`, one more than any slice -- behind the pass's entries); payload = the slot.  Regions are
// disjoint and cover every slot: a region occupies the same positions before and after the sort
__global__ __launch_bounds__(256) void expand_keys(size_t n, const int *__restrict__ tcol, const unsigned *__restrict__ pass_of,
                                                   const int4 *__restrict__ pass, int slice_shift, unsigned slices, unsigned slice_bits,
                                                   unsigned long long *__restrict__ keys, unsigned *__restrict__ slot) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const unsigned p = pass_of[e];
        unsigned lo = slices, region = (unsigned)e;  // (a slot no pass owns: a region of its own)
        if (p != 0xffffffffu) {
            const int4 d = pass[p];
            region = (unsigned)d.x;
            if ((long long)e < (long long)d.x + d.y) lo = min((unsigned)tcol[e] >> slice_shift, slices - 1);
        }
        keys[e] = ((unsigned long long)region << slice_bits) | lo;
        slot[e] = (unsigned)e;
    }
}
// after sort 1 position q holds slot order1[q]: the entry's place in x' is q (a pass's region is the same in both
// orders).  place[e] = q; the key of sort 2 = the slice (slots that are not entries: the last slice, expanded like
// entries, read by nobody), payload q
__global__ __launch_bounds__(256) void expand_places(size_t n, const unsigned long long *__restrict__ sorted, const unsigned *__restrict__ order1,
                                                     unsigned slices, unsigned slice_bits, unsigned *__restrict__ place,
                                                     unsigned *__restrict__ key2, unsigned *__restrict__ pos) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
        place[order1[q]] = (unsigned)q;
        key2[q] = min((unsigned)(sorted[q] & ((1ull << slice_bits) - 1)), slices - 1);
        pos[q] = (unsigned)q;
    }
}
// first[c] = the first sorted position whose slice is >= c (c = 0 .. slices)
__global__ __launch_bounds__(256) void expand_starts(size_t n, const unsigned *__restrict__ sorted, unsigned slices, unsigned *__restrict__ first) {
    const unsigned c = blockIdx.x * 256 + threadIdx.x;
    if (c > slices) return;
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (sorted[mid] < c) lo = mid + 1;
        else hi = mid;
    }
    first[c] = (unsigned)lo;
}
// lcol[k] = the column of the entry that lives at x'[dest[k]], inside its slice
__global__ __launch_bounds__(256) void expand_lcol(size_t n, const unsigned *__restrict__ slice_sorted, const unsigned *__restrict__ dest,
                                                   const unsigned *__restrict__ order1, const int *__restrict__ tcol, int slice_shift,
                                                   const unsigned *__restrict__ run_start, unsigned short *__restrict__ lcol) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const long long l = (long long)(unsigned)tcol[order1[dest[k]]] - ((long long)slice_sorted[k] << slice_shift);
        lcol[k] = (unsigned short)(max(0ll, min(l, (1ll << slice_shift) - 1)) | (run_start[k] ? kExpandRunStart : 0u));
    }
}
// a RUN: consecutive entries of the slice order whose places in x' are consecutive as well (and in one slice)
__global__ __launch_bounds__(256) void expand_run_starts(size_t n, const unsigned *__restrict__ slice_sorted, const unsigned *__restrict__ dest,
                                                         unsigned *__restrict__ run_start) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256)
        run_start[k] = k == 0 || dest[k] != dest[k - 1] + 1 || slice_sorted[k] != slice_sorted[k - 1];
}
// delta[r] = place - k for the entries of run r (runs_upto[k] = runs that have begun up to and including k)
__global__ __launch_bounds__(256) void expand_run_delta(size_t n, const unsigned *__restrict__ run_start, const unsigned *__restrict__ runs_upto,
                                                        const unsigned *__restrict__ dest, unsigned *__restrict__ delta) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256)
        if (run_start[k]) delta[runs_upto[k] - 1] = dest[k] - (unsigned)k;
}
// group_run[chunk.w + j] = the run of the chunk's entry 64 j; chunk_runs = {the chunk's first run, its runs}
__global__ __launch_bounds__(256) void expand_groups(int chunks, const int4 *__restrict__ chunk, const unsigned *__restrict__ runs_upto,
                                                     unsigned *__restrict__ group_run, int2 *__restrict__ chunk_runs) {
    if ((int)blockIdx.x >= chunks) return;
    const int4 c = chunk[blockIdx.x];
    for (int j = threadIdx.x; j < (c.z + 63) / 64; j += 256) group_run[c.w + j] = runs_upto[(size_t)c.y + 64 * (size_t)j] - 1;
    if (threadIdx.x == 0) {
        const unsigned r0 = runs_upto[(size_t)c.y] - 1, r1 = runs_upto[(size_t)c.y + (size_t)c.z - 1] - 1;
        chunk_runs[blockIdx.x] = make_int2((int)r0, (int)(r1 - r0 + 1));
    }
}
// the packed column word of slot e: head << 31 | local row << 14 | place in the pass's segment of x'
__global__ __launch_bounds__(256) void expand_words(size_t n, const unsigned short *__restrict__ tkey, const unsigned *__restrict__ place,
                                                    const unsigned *__restrict__ pass_of, const int4 *__restrict__ pass,
                                                    int *__restrict__ words) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const unsigned p = pass_of[e];
        unsigned w = 0;
        if (p != 0xffffffffu) {
            const unsigned key = tkey[e];
            w = ((key & (unsigned)kTileHead) << 16) | ((key & (unsigned)kTileRowMask) << kTilePackShift) |
                ((place[e] - (unsigned)pass[p].x) & kTilePackColMask);
        }
        words[e] = (int)w;
    }
}
// the packed plan's descriptors: the pass's own segment of x' as its window
__global__ __launch_bounds__(256) void expand_pass_desc(int passes, const int4 *__restrict__ pass, int4 *__restrict__ out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= passes) return;
    const int4 d = pass[p];
    out[p] = make_int4(d.x, d.y, d.x, max((d.y + 3) & ~3, 4) | kTilePassPacked | (d.w & kTilePassLast));
}

}  // namespace tile_dev

// what tile_expand walks and what csr_tile<.., PACK> runs on, freed with the owner
struct TileExpansion {
    unsigned short *lcol = nullptr;  // [slots + pad] column inside the slice | run start << 15, slice order
    unsigned *delta = nullptr;       // [runs] place in x' - position in slice order, per run
    unsigned *group_run = nullptr;   // [groups] the run of every 64th entry of a chunk
    int4 *chunk = nullptr;           // [chunks] {slice, first, entries, first group}
    int2 *chunk_runs = nullptr;      // [chunks] {first run, runs}
    size_t runs = 0, groups = 0;
    int *words = nullptr;            // [slots + kTileChunkMax] packed column words, the plan's entry order
    int4 *pass = nullptr;            // [passes] the plan's descriptors with the pass's segment of x' as window
    int chunks = 0;
    size_t entries = 0;
    ~TileExpansion() {
        (void)hipFree(lcol);
        (void)hipFree(delta);
        (void)hipFree(group_run);
        (void)hipFree(chunk_runs);
        (void)hipFree(chunk);
        (void)hipFree(words);
        (void)hipFree(pass);
    }
};

// 1: built; -1: HIP error (message through err).  tcol, tkey: the plan's arrays on the device, n slots; d_pass: its
// descriptors (device, any order), passes of them.
template <typename T>
int tile_build_expansion(int N, const int *tcol, const unsigned short *tkey, size_t n, const int4 *d_pass, int passes, hipStream_t s,
                         TileExpansion &ex, std::string &err) {
    using namespace tile_dev;
    UploadTrace trace("tile_build_expansion");
    constexpr int W = tile_slice_cols<T>();
    static_assert((W & (W - 1)) == 0, "slices are a power of two wide");
    const int slice_shift = (int)bits_for((unsigned long long)W) - 1;
    const unsigned slices = (unsigned)(((long long)std::max(N, 1) + W - 1) / W);
    Scratch tmp;
    auto bad = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return false;
        err = std::string("expansion plan: ") + what + " failed: " + hipGetErrorString(e);
        return true;
    };
    unsigned long long *keys_a, *keys_b;
    unsigned *slot_a, *order1, *pass_of, *place, *key2_a, *key2_b, *pos_a, *first, *dest, *run_start, *runs_upto;
    hipError_t e = tmp.alloc(&keys_a, n);
    if (e == hipSuccess) e = tmp.alloc(&keys_b, n);
    if (e == hipSuccess) e = tmp.alloc(&slot_a, n);
    if (e == hipSuccess) e = tmp.alloc(&order1, n);
    if (e == hipSuccess) e = tmp.alloc(&pass_of, n);
    if (e == hipSuccess) e = tmp.alloc(&place, n);
    if (e == hipSuccess) e = tmp.alloc(&key2_a, n);
    if (e == hipSuccess) e = tmp.alloc(&key2_b, n);
    if (e == hipSuccess) e = tmp.alloc(&pos_a, n);
    if (e == hipSuccess) e = tmp.alloc(&first, (size_t)slices + 1);
    if (e == hipSuccess) e = tmp.alloc(&dest, n);
    if (e == hipSuccess) e = tmp.alloc(&run_start, n);
    if (e == hipSuccess) e = tmp.alloc(&runs_upto, n);
    if (e == hipSuccess) e = hipMalloc((void **)&ex.lcol, (n + 64) * sizeof(unsigned short));
    if (e == hipSuccess) e = hipMalloc((void **)&ex.words, (n + kTileChunkMax) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&ex.pass, (size_t)std::max(passes, 1) * sizeof(int4));
    if (e == hipSuccess) e = hipMemsetAsync(pass_of, 0xff, n * sizeof(unsigned), s);
    if (e == hipSuccess) e = hipMemsetAsync(ex.words + n, 0, (size_t)kTileChunkMax * sizeof(int), s);
    if (bad(e, "allocation")) return -1;
    if (passes > 0) hipLaunchKernelGGL(expand_pass_of, dim3(passes), dim3(256), 0, s, passes, d_pass, pass_of);
    const unsigned slice_bits = bits_for((unsigned long long)slices);  // values 0 .. slices
    const unsigned key_bits = slice_bits + bits_for((unsigned long long)n);
    hipLaunchKernelGGL(expand_keys, dim3(4096), dim3(256), 0, s, n, tcol, pass_of, d_pass, slice_shift, slices, slice_bits, keys_a, slot_a);
    size_t tmp_bytes = 0;
    void *d_tmp = nullptr;
    e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a, keys_b, slot_a, order1, n, 0u, key_bits, s);
    if (e == hipSuccess) e = tmp.alloc((char **)&d_tmp, tmp_bytes);
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, keys_a, keys_b, slot_a, order1, n, 0u, key_bits, s);
    if (bad(e, "sort by (pass, slice)")) return -1;
    hipLaunchKernelGGL(expand_places, dim3(4096), dim3(256), 0, s, n, keys_b, order1, slices, slice_bits, place, key2_a, pos_a);
    size_t tmp2_bytes = 0;
    void *d_tmp2 = nullptr;
    const unsigned end_bit = bits_for((unsigned long long)slices);
    e = rocprim::radix_sort_pairs(nullptr, tmp2_bytes, key2_a, key2_b, pos_a, dest, n, 0u, end_bit, s);
    if (e == hipSuccess) e = tmp.alloc((char **)&d_tmp2, tmp2_bytes);
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp2, tmp2_bytes, key2_a, key2_b, pos_a, dest, n, 0u, end_bit, s);
    if (bad(e, "sort by slice")) return -1;
    hipLaunchKernelGGL(expand_starts, dim3((slices + 1 + 255) / 256), dim3(256), 0, s, n, key2_b, slices, first);
    hipLaunchKernelGGL(expand_run_starts, dim3(4096), dim3(256), 0, s, n, key2_b, dest, run_start);
    size_t tmp3_bytes = 0;
    void *d_tmp3 = nullptr;
    e = rocprim::inclusive_scan(nullptr, tmp3_bytes, run_start, runs_upto, n, rocprim::plus<unsigned>(), s);
    if (e == hipSuccess) e = tmp.alloc((char **)&d_tmp3, tmp3_bytes);
    if (e == hipSuccess) e = rocprim::inclusive_scan(d_tmp3, tmp3_bytes, run_start, runs_upto, n, rocprim::plus<unsigned>(), s);
    if (bad(e, "numbering the runs")) return -1;
    hipLaunchKernelGGL(expand_lcol, dim3(4096), dim3(256), 0, s, n, key2_b, dest, order1, tcol, slice_shift, run_start, ex.lcol);
    hipLaunchKernelGGL(expand_words, dim3(4096), dim3(256), 0, s, n, tkey, place, pass_of, d_pass, ex.words);
    if (passes > 0) hipLaunchKernelGGL(expand_pass_desc, dim3((passes + 255) / 256), dim3(256), 0, s, passes, d_pass, ex.pass);
    std::vector<unsigned> h_first((size_t)slices + 1);
    unsigned runs = 0;
    e = hipMemcpyAsync(h_first.data(), first, h_first.size() * sizeof(unsigned), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && n) e = hipMemcpyAsync(&runs, runs_upto + (n - 1), sizeof(unsigned), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (bad(e, "keys, sorts, words")) return -1;
    trace.mark("keys, two sorts, runs, slice columns, packed words");
    std::vector<int4> chunks;
    size_t groups = 0;
    for (unsigned c = 0; c < slices; ++c)
        for (unsigned k = h_first[c]; k < h_first[(size_t)c + 1]; k += (unsigned)kExpandChunk) {
            const unsigned cnt = std::min<unsigned>((unsigned)kExpandChunk, h_first[(size_t)c + 1] - k);
            chunks.push_back(make_int4((int)c, (int)k, (int)cnt, (int)groups));
            groups += (cnt + 63) / 64;
        }
    e = hipMalloc((void **)&ex.chunk, std::max<size_t>(1, chunks.size()) * sizeof(int4));
    if (e == hipSuccess) e = hipMalloc((void **)&ex.chunk_runs, std::max<size_t>(1, chunks.size()) * sizeof(int2));
    if (e == hipSuccess) e = hipMalloc((void **)&ex.delta, std::max<size_t>(1, runs) * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void **)&ex.group_run, std::max<size_t>(1, groups) * sizeof(unsigned));
    if (e == hipSuccess && !chunks.empty())
        e = hipMemcpyAsync(ex.chunk, chunks.data(), chunks.size() * sizeof(int4), hipMemcpyHostToDevice, s);
    if (bad(e, "chunks")) return -1;
    hipLaunchKernelGGL(expand_run_delta, dim3(4096), dim3(256), 0, s, n, run_start, runs_upto, dest, ex.delta);
    if (!chunks.empty())
        hipLaunchKernelGGL(expand_groups, dim3((unsigned)chunks.size()), dim3(256), 0, s, (int)chunks.size(), ex.chunk, runs_upto,
                           ex.group_run, ex.chunk_runs);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (bad(e, "runs, groups")) return -1;
    ex.chunks = (int)chunks.size();
    ex.entries = n;
    ex.runs = runs;
    ex.groups = groups;
    return 1;
}

}  // namespace spmv
