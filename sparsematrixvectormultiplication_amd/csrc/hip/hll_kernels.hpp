// hll_kernels.hpp -- HLL (hacked ELLPACK, 32-row hacks) SpMV for gfx950.
//
// Device layout ("flat slab"): the reference keeps one JA/AS allocation pair
// per hack plus an array of structs holding device pointers
// (main_cuda.cu:369-402; ~221 k cudaMallocs for an nlpkkt120-sized matrix).
// Here all hacks sit back to back in ONE JA array and ONE AS array, each hack
// still ROW-MAJOR exactly as the host struct stores it
// (slot (i, j) of hack h at hack_off[h] + i * maxnz[h] + j, libs/hll_matrix.h
// and src/hll_matrix.c:235), every hack start padded to an even slot so
// 8/16-byte loads stay aligned:
//     hack_off[H+1] (int64)   maxnz[H] (int32)   JA[S] (int32)   AS[S] (T)
//
// Kernels (replacing cuda_src/hll_matrix.cu:346-479):
//   hll_thread_row   one lane per row, walks its row of the row-major slab
//   hll_vector<L>    L lanes per row
//   hll_lds          a workgroup takes one or more whole hacks: the slab range
//                    is read linearly (perfectly coalesced 8/16-byte loads,
//                    whatever maxnz is), multiplied by the gathered x, staged
//                    in LDS, then each row is summed out of LDS by a lane
//                    group.  Row-major slabs are the worst case for a
//                    lane-per-row walk (stride maxnz between lanes); staging
//                    the slab turns them into a linear stream.
#pragma once
#include <hip/hip_runtime.h>

#include "csr_kernels.hpp"
#include "wave_ops.hpp"

namespace spmv {

constexpr int kHack = 32;         // HACK_SIZE, libs/hll_matrix.h:12
constexpr int kHllCap = 4096;     // slots staged per workgroup (32 KiB of fp64 products)

template <typename T>
__global__ __launch_bounds__(kBlock) void hll_thread_row(int M, const long long *__restrict__ hack_off,
                                                         const int *__restrict__ maxnz,
                                                         const int *__restrict__ JA,
                                                         const T *__restrict__ AS,
                                                         const T *__restrict__ x,
                                                         T *__restrict__ y) {
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= M) return;
    const int h = r / kHack, i = r % kHack;
    const int m = maxnz[h];
    const long long at = hack_off[h] + (long long)i * m;
    T acc = 0;
    for (int j = 0; j < m; ++j) acc += AS[at + j] * gather(x, JA[at + j]);
    y[r] = acc;
}

template <typename T, int L>
__global__ __launch_bounds__(kBlock) void hll_vector(int M, const long long *__restrict__ hack_off,
                                                     const int *__restrict__ maxnz,
                                                     const int *__restrict__ JA,
                                                     const T *__restrict__ AS,
                                                     const T *__restrict__ x, T *__restrict__ y) {
    constexpr int kRows = kBlock / L;
    const int r = blockIdx.x * kRows + threadIdx.x / L;
    const int lane = threadIdx.x % L;
    T acc = 0;
    if (r < M) {
        const int h = r / kHack, i = r % kHack;
        const int m = maxnz[h];
        const long long at = hack_off[h] + (long long)i * m;
        for (int j = lane; j < m; j += L) acc += AS[at + j] * gather(x, JA[at + j]);
    }
    acc = group_sum<L>(acc);
    if (lane == 0 && r < M) y[r] = acc;
}

// Stage NU units (NU * 512 slots starting at slot k0 of the range) as products:
// straight-line, unconditional loads; only the LDS stores are bounded by count.
// The last unit may overshoot the range by < 512 slots; JA / AS carry that much
// zero padding behind the slab and later hacks' slots are valid anyway.
template <typename T, bool NT, int NU>
__device__ __forceinline__ void stage_units(T *prod, const int *__restrict__ JA,
                                            const T *__restrict__ AS, const T *__restrict__ x,
                                            long long from, int k0, int count) {
    using V2 = typename vec2<T>::type;
    const int t = threadIdx.x;
    v2i c[NU];
    V2 v[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int k = k0 + u * kStreamUnit + 2 * t;
        c[u] = stream_load<NT>(reinterpret_cast<const v2i *>(JA + from + k));
        v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(AS + from + k));
    }
    T xv[2 * NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        xv[2 * u] = gather(x, c[u].x);
        xv[2 * u + 1] = gather(x, c[u].y);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int k = k0 + u * kStreamUnit + 2 * t;
        if (k < count) {
            V2 p;
            p.x = v[u].x * xv[2 * u];
            p.y = v[u].y * xv[2 * u + 1];
            *reinterpret_cast<V2 *>(&prod[k]) = p;
        }
    }
}

// Stage slots [from, from + count) of the flat slab as products in LDS.
// `from` is even; prod[k] receives slot from + k.  The number of units is
// wave-uniform, so the dispatch below is scalar branching, not predication.
template <typename T, bool NT>
__device__ __forceinline__ void stage_products(T *prod, const int *__restrict__ JA,
                                               const T *__restrict__ AS,
                                               const T *__restrict__ x, long long from,
                                               int count) {
    int units = (count + kStreamUnit - 1) / kStreamUnit;
    int k0 = 0;
    while (units >= 4) {
        stage_units<T, NT, 4>(prod, JA, AS, x, from, k0, count);
        k0 += 4 * kStreamUnit;
        units -= 4;
    }
    if (units == 3) stage_units<T, NT, 3>(prod, JA, AS, x, from, k0, count);
    else if (units == 2) stage_units<T, NT, 2>(prod, JA, AS, x, from, k0, count);
    else if (units == 1) stage_units<T, NT, 1>(prod, JA, AS, x, from, k0, count);
}

// Workgroup b owns hacks [hblk[b], hblk[b+1]).  Host packing (hll_build_blocks)
// keeps the slots of a multi-hack workgroup within kHllCap; a single hack
// larger than that is walked in row chunks (or, if even one row does not fit,
// row by row with register accumulation).
template <typename T, bool NT>
__global__ __launch_bounds__(kBlock) void hll_lds(int M, const int *__restrict__ hblk,
                                                  const long long *__restrict__ hack_off,
                                                  const int *__restrict__ maxnz,
                                                  const int *__restrict__ JA,
                                                  const T *__restrict__ AS,
                                                  const T *__restrict__ x, T *__restrict__ y) {
    __shared__ T prod[kHllCap + 2];
    __shared__ T wave_part[kBlock / 64];
    const int t = threadIdx.x;
    const int h0 = hblk[blockIdx.x], h1 = hblk[blockIdx.x + 1];
    const long long base = hack_off[h0];
    const long long span = hack_off[h1] - base;
    const int row_first = h0 * kHack;
    const int row_last = min(h1 * kHack, M);  // exclusive
    const int nrows = row_last - row_first;

    if (span <= kHllCap) {
        stage_products<T, NT>(prod, JA, AS, x, base, (int)span);
        __syncthreads();
        int lanes = 1;
        if (nrows <= kBlock / 2) {
            lanes = 1 << (31 - __clz(kBlock / (nrows > 0 ? nrows : 1)));
            if (lanes > 64) lanes = 64;
        }
        const int rows_per_pass = kBlock / lanes;
        const int my_row = t / lanes, my_lane = t % lanes;
        for (int first = 0; first < nrows; first += rows_per_pass) {
            const int q = first + my_row;
            T acc = 0;
            if (q < nrows) {
                const int h = h0 + q / kHack, i = q % kHack;
                const int m = maxnz[h];
                const int lo = (int)(hack_off[h] - base) + i * m;
                acc = lds_strided_sum(prod, lo, lo + m, my_lane, lanes);
            }
            acc = group_sum_rt(acc, lanes);
            if (my_lane == 0 && q < nrows) y[row_first + q] = acc;
        }
        return;
    }

    // one oversized hack
    const int m = maxnz[h0];
    if (m <= kHllCap) {
        const int rows_per_chunk = kHllCap / m;  // >= 1, < 32 here
        int lanes = 1 << (31 - __clz(kBlock / rows_per_chunk));
        if (lanes > 64) lanes = 64;
        const int my_row = t / lanes, my_lane = t % lanes;
        for (int i0 = 0; i0 < nrows; i0 += rows_per_chunk) {
            const int rows = min(rows_per_chunk, nrows - i0);
            const long long from = base + (long long)i0 * m;
            const long long from_even = from & ~1LL;
            const int shift = (int)(from - from_even);
            __syncthreads();  // previous chunk fully consumed
            stage_products<T, NT>(prod, JA, AS, x, from_even, rows * m + shift);
            __syncthreads();
            for (int first = 0; first < rows; first += kBlock / lanes) {
                const int q = first + my_row;
                T acc = 0;
                if (q < rows) {
                    const int lo = shift + q * m;
                    acc = lds_strided_sum(prod, lo, lo + m, my_lane, lanes);
                }
                acc = group_sum_rt(acc, lanes);
                if (my_lane == 0 && q < rows) y[row_first + i0 + q] = acc;
            }
        }
        return;
    }

    // rows longer than the stage: whole workgroup per row, registers only
    for (int i = 0; i < nrows; ++i) {
        const long long at = base + (long long)i * m;
        T acc = 0;
        for (int j = t; j < m; j += kBlock) acc += AS[at + j] * gather(x, JA[at + j]);
        acc = group_sum<64>(acc);
        __syncthreads();
        if ((t & 63) == 0) wave_part[t >> 6] = acc;
        __syncthreads();
        if (t == 0) {
            T s = wave_part[0];
            for (int w = 1; w < kBlock / 64; ++w) s += wave_part[w];
            y[row_first + i] = s;
        }
    }
}

}  // namespace spmv
