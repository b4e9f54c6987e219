// plan_kernels.hpp -- the x-window plan built on the device.
//
// The plan of a block (or HLL window) is the ascending list of the distinct 128-byte lines of x its
// column indices touch, plus, per entry, the 16-bit slot (rank of its line in that list * elements
// per line + column % elements per line).  On the host that is one pass with a stamp array
// (csr_build_local / hll_build_local); here a workgroup sorts the <= 2048 line ids of its segment in
// LDS (bitonic network), marks the heads of equal runs, scans them, and -- in the second kernel --
// writes the list and looks every entry's line up in it by binary search.  Two kernels because the
// lists are packed back to back: the host turns the counts into offsets in between (one int per block).
#pragma once
#include <hip/hip_runtime.h>

#include <climits>

#include "csr_kernels.hpp"

namespace spmv {

constexpr int kPlanCap = 2048;               // entries per segment (the x-window kernels' stage)
constexpr int kPlanPer = kPlanCap / kBlock;  // keys per thread

// keys[kPlanCap] <- line ids of the segment (INT_MAX padding), sorted ascending.  scan[kBlock + 1]
// <- exclusive scan of the number of run heads each thread owns (thread t owns sorted positions
// [t * kPlanPer, (t + 1) * kPlanPer)); returns the number of distinct lines.
template <int SHIFT>
__device__ __forceinline__ int sort_segment_lines(int *keys, int *scan, const int *__restrict__ idx, long long begin,
                                                  int len) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < kPlanPer; ++j) {
        const int k = t + j * kBlock;
        keys[k] = k < len ? (idx[begin + k] >> SHIFT) : INT_MAX;
    }
    for (int size = 2; size <= kPlanCap; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kPlanPer / 2; ++j) {
                const int p = t + j * kBlock;                       // pair number, 0 .. kPlanCap / 2 - 1
                const int i = ((p / stride) * stride << 1) + (p % stride);
                const int q = i + stride;
                const bool up = (i & size) == 0;
                const int a = keys[i], b = keys[q];
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[q] = a;
                }
            }
        }
    }
    __syncthreads();
    int heads = 0;
#pragma unroll
    for (int j = 0; j < kPlanPer; ++j) {
        const int pos = t * kPlanPer + j;
        const int k = keys[pos];
        heads += (k != INT_MAX) && (pos == 0 || k != keys[pos - 1]);
    }
    scan[t + 1] = heads;
    if (t == 0) scan[0] = 0;
    __syncthreads();
    for (int d = 1; d < kBlock; d <<= 1) {  // inclusive Hillis-Steele over scan[1 .. kBlock]
        const int v = (t + 1 > d) ? scan[t + 1 - d] : 0;
        __syncthreads();
        scan[t + 1] += v;
        __syncthreads();
    }
    return scan[kBlock];
}

template <int SHIFT>
__global__ __launch_bounds__(kBlock) void plan_count(int segments, const long long *__restrict__ seg_begin,
                                                     const int *__restrict__ seg_len, const int *__restrict__ idx,
                                                     int *__restrict__ nlines) {
    __shared__ int keys[kPlanCap];
    __shared__ int scan[kBlock + 1];
    const int b = blockIdx.x;
    if (b >= segments) return;
    const int n = sort_segment_lines<SHIFT>(keys, scan, idx, seg_begin[b], seg_len[b]);
    if (threadIdx.x == 0) nlines[b] = n;
}

// Needs every segment to list at most `max_lines` (<= kPlanCap / 8) lines: the host checks the counts first.
template <int SHIFT>
__global__ __launch_bounds__(kBlock) void plan_fill(int segments, const long long *__restrict__ seg_begin,
                                                    const int *__restrict__ seg_len, const int *__restrict__ idx,
                                                    const int *__restrict__ line_off, int *__restrict__ lines,
                                                    unsigned short *__restrict__ slot) {
    __shared__ int keys[kPlanCap];
    __shared__ int scan[kBlock + 1];
    __shared__ int uniq[kLocalLinesMax];
    const int b = blockIdx.x;
    if (b >= segments) return;
    const int t = threadIdx.x;
    const long long begin = seg_begin[b];
    const int len = seg_len[b];
    const int n = sort_segment_lines<SHIFT>(keys, scan, idx, begin, len);
    int rank = scan[t];
#pragma unroll
    for (int j = 0; j < kPlanPer; ++j) {
        const int pos = t * kPlanPer + j;
        const int k = keys[pos];
        if (k != INT_MAX && (pos == 0 || k != keys[pos - 1])) {
            if (rank < kLocalLinesMax) uniq[rank] = k;
            lines[line_off[b] + rank] = k;
            ++rank;
        }
    }
    __syncthreads();
    constexpr int mask = (1 << SHIFT) - 1;
    for (int k = t; k < len; k += kBlock) {
        const int c = idx[begin + k];
        const int line = c >> SHIFT;
        int lo = 0, hi = n - 1;  // the line is in the list: plain binary search
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (uniq[mid] < line) lo = mid + 1;
            else hi = mid;
        }
        slot[begin + k] = (unsigned short)((lo << SHIFT) | (c & mask));
    }
}

// ---- the pattern plan of an x-window plan (csr_stream_local<.., PAT>): one workgroup per block.
// pat_mark: rowflag[r] = 0x10000 | (delta & 0xffff) when row r's slots are the previous row's (same block, same length)
// plus the constant delta, else 0 (its slots go into the block's pattern table); pcount[b] = slots in that table (even).
// HLL = false: desc = {first row, first entry, rows, end entry}, a row's slots are slot[row_ptr[r] .. row_ptr[r + 1]);
// HLL = true (hll_lds_local's windows): desc = {first row, rows, first slot lo, hi}, a row's slots are
// slot[base + (row_seg[r] & 0xffff) .. + (row_seg[r] >> 16)) with base = the window's even base
template <bool HLL>
__device__ __forceinline__ void pat_block_rows(const int4 d, int &r0, int &nrows, long long &base) {
    r0 = d.x;
    nrows = HLL ? d.y : d.z;
    base = HLL ? ((((long long)d.w << 32) | (unsigned)d.z) & ~1LL) : 0;
}
template <bool HLL>
__device__ __forceinline__ void pat_row_slots(const int *__restrict__ row_ptr, const unsigned *__restrict__ row_seg, long long base,
                                              int r, long long &s, int &len) {
    if constexpr (HLL) {
        const unsigned seg = row_seg[r];
        s = base + (long long)(seg & 0xffffu);
        len = (int)(seg >> 16);
    } else {
        s = row_ptr[r];
        len = row_ptr[r + 1] - row_ptr[r];
    }
}
template <int BLOCK, bool HLL = false>
__global__ __launch_bounds__(BLOCK) void pat_mark(int blocks, const int4 *__restrict__ desc, const int *__restrict__ row_ptr,
                                                  const unsigned short *__restrict__ slot, int *__restrict__ rowflag,
                                                  int *__restrict__ pcount, const unsigned *__restrict__ row_seg = nullptr) {
    __shared__ int cnt;
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= blocks) return;
    int r0, nrows;
    long long base;
    pat_block_rows<HLL>(desc[b], r0, nrows, base);
    if (t == 0) cnt = 0;
    __syncthreads();
    for (int i = t; i < nrows; i += BLOCK) {
        long long s;
        int len;
        pat_row_slots<HLL>(row_ptr, row_seg, base, r0 + i, s, len);
        bool derived = false;
        int delta = 0;
        if (i > 0 && len > 0) {
            long long ps;
            int plen;
            pat_row_slots<HLL>(row_ptr, row_seg, base, r0 + i - 1, ps, plen);
            if (plen == len) {
                delta = (int)slot[s] - (int)slot[ps];
                derived = true;
                for (int j = 1; j < len; ++j)
                    if ((int)slot[s + j] - (int)slot[ps + j] != delta) {
                        derived = false;
                        break;
                    }
            }
        }
        rowflag[r0 + i] = derived ? (0x10000 | (delta & 0xffff)) : 0;
        if (!derived && len > 0) atomicAdd(&cnt, (len + 7) & ~7);  // a row's pattern: whole groups of 8 slots
    }
    __syncthreads();
    if (t == 0) pcount[b] = cnt;
}
// pat_fill: rinfo[r] = the group of 8 slots at which the row's pattern starts in the block's table | shift << 16 (a row that is its predecessor
// shifted shares the predecessor's pattern, its shift the sum of the deltas since); the table itself; pdesc[b] = {first
// element in ptab, elements}
template <int BLOCK, bool HLL = false>
__global__ __launch_bounds__(BLOCK) void pat_fill(int blocks, const int4 *__restrict__ desc, const int *__restrict__ row_ptr,
                                                  const unsigned short *__restrict__ slot, const int *__restrict__ rowflag,
                                                  const long long *__restrict__ pbase, unsigned *__restrict__ rinfo,
                                                  unsigned short *__restrict__ ptab, int2 *__restrict__ pdesc,
                                                  const unsigned *__restrict__ row_seg = nullptr) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= blocks) return;
    int r0, nrows;
    long long base;
    pat_block_rows<HLL>(desc[b], r0, nrows, base);
    const long long pb = pbase[b];
    if (t == 0) {
        int filled = 0, cur = 0, shift = 0;
        for (int i = 0; i < nrows; ++i) {
            const int f = rowflag[r0 + i];
            if (f & 0x10000) {
                shift += (int)(short)(f & 0xffff);
            } else {
                long long s;
                int len;
                pat_row_slots<HLL>(row_ptr, row_seg, base, r0 + i, s, len);
                cur = filled;  // (in groups of 8 slots)
                shift = 0;
                filled += (len + 7) >> 3;
            }
            rinfo[r0 + i] = (unsigned)cur | ((unsigned)(shift & 0xffff) << 16);
        }
        pdesc[b] = make_int2((int)pb, filled * 8);
    }
    __syncthreads();
    for (int i = t; i < nrows; i += BLOCK) {
        if (rowflag[r0 + i] & 0x10000) continue;
        long long s;
        int len;
        pat_row_slots<HLL>(row_ptr, row_seg, base, r0 + i, s, len);
        const long long o = pb + 8 * (long long)(rinfo[r0 + i] & 0xffffu);
        for (int j = 0; j < len; ++j) ptab[o + j] = slot[s + j];
    }
}

}  // namespace spmv
