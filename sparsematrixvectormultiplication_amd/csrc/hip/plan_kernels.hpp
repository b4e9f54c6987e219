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

}  // namespace spmv
