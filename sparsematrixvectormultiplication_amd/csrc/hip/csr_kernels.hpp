// csr_kernels.hpp -- CSR SpMV kernels for gfx950 (MI355X).
//
// y = A x for A in 0-based CSR (the reference's CSRMatrix,
// libs/csr_matrix.h:8-16).  These replace the reference's three CUDA kernels
// (cuda_src/csr_matrix_cuda.cu:122-241); none of them is a translation:
//
//   csr_thread_row   one lane per row                 (G1's job)
//   csr_vector<L,V>  L lanes per row, V-wide loads    (G2's job; L = 64, V = 2 is
//                    the "one wavefront per row, vectorised loads" kernel)
//   csr_stream       nnz-balanced row blocks streamed through LDS: every lane
//                    loads the same number of (col, val) pairs with fully
//                    coalesced 8/16-byte loads no matter how the rows are cut,
//                    multiplies by the gathered x, parks the products in LDS,
//                    and the rows are then summed out of LDS by lane groups.
//                    This is the kernel that approaches the HBM roofline on
//                    short-row matrices (nlpkkt: ~27 nnz/row).
//
// The path is a bandwidth-bound gather (0.17 flop/byte in fp64): no MFMA.
#pragma once
#include <hip/hip_runtime.h>

#include "wave_ops.hpp"

namespace spmv {

constexpr int kBlock = 256;          // threads per workgroup (4 wavefronts)
constexpr int kStreamCap = 2048;     // nnz staged per workgroup (16 KiB of fp64 products)
constexpr int kStreamUnit = 2 * kBlock;  // nnz one pass of the workgroup covers
constexpr int kStreamRowsCap = 1024; // rows per workgroup (bounds the empty-row case)
constexpr int kLongPiece = 8192;     // nnz per workgroup when one row is split
constexpr int kLongFlag = 0x40000000;

typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));
template <typename T> struct vec2;
template <> struct vec2<float> { using type = v2f; };
template <> struct vec2<double> { using type = v2d; };

// streamed-once data goes past the caches' retention (global_load ... nt) so
// that x keeps its place in L2 / Infinity Cache
template <bool NT, typename V>
__device__ __forceinline__ V stream_load(const V *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// Workgroup ids are dealt round-robin over the 8 XCDs; give each XCD one
// contiguous eighth of the work so that its private L2 sees one window of x
// instead of all eight XCDs caching the same window.
__device__ __forceinline__ int xcd_contiguous(int bid, int per_xcd) {
    return (bid & 7) * per_xcd + (bid >> 3);
}

// ---------------------------------------------------------------- thread/row
template <typename T>
__global__ __launch_bounds__(kBlock) void csr_thread_row(int M, const int *__restrict__ row_ptr,
                                                         const int *__restrict__ col,
                                                         const T *__restrict__ val,
                                                         const T *__restrict__ x,
                                                         T *__restrict__ y) {
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= M) return;
    T acc = 0;
    const int stop = row_ptr[r + 1];
    for (int e = row_ptr[r]; e < stop; ++e) acc += val[e] * x[col[e]];
    y[r] = acc;
}

// ------------------------------------------------------------ L lanes per row
// VEC = 2: every lane loads two neighbouring entries with one 8-byte (col) and
// one 16-byte (fp64 val) load; the row start is peeled down to an even entry
// so the wide loads stay naturally aligned.  The arrays carry >= 2 entries of
// zero padding behind the last nonzero, so the peel never leaves the buffer.
template <typename T, int L, int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void csr_vector(int M, const int *__restrict__ row_ptr,
                                                     const int *__restrict__ col,
                                                     const T *__restrict__ val,
                                                     const T *__restrict__ x, T *__restrict__ y) {
    using V2 = typename vec2<T>::type;
    constexpr int kRows = kBlock / L;
    const int r = blockIdx.x * kRows + threadIdx.x / L;
    const int lane = threadIdx.x % L;
    T acc = 0;
    if (r < M) {
        const int begin = row_ptr[r], stop = row_ptr[r + 1];
        if constexpr (VEC == 1) {
            for (int e = begin + lane; e < stop; e += L) acc += val[e] * x[col[e]];
        } else {
            for (int e = (begin & ~1) + 2 * lane; e < stop; e += 2 * L) {
                const v2i c = stream_load<NT>(reinterpret_cast<const v2i *>(col + e));
                const V2 v = stream_load<NT>(reinterpret_cast<const V2 *>(val + e));
                const T p0 = e >= begin ? v.x * x[c.x] : T(0);
                const T p1 = e + 1 < stop ? v.y * x[c.y] : T(0);
                acc += p0;
                acc += p1;
            }
        }
    }
    acc = group_sum<L>(acc);  // every lane of the wavefront takes part
    if (lane == 0 && r < M) y[r] = acc;
}

// -------------------------------------------------------------------- stream
// Workgroup b owns rows [desc[b].x, desc[b+1].x) whose entries start at
// desc[b].y; descriptors are built on the host at upload (csr_build_blocks)
// so that the entries of a workgroup fit the LDS stage.  A row longer than
// the stage is cut into pieces of kLongPiece entries (desc.x carries
// kLongFlag, desc.z the slot of the piece's partial sum); csr_long_finish
// then adds the pieces of each such row in slot order, so results do not
// depend on scheduling (no atomics).
template <typename T, bool NT, bool XCD>
__global__ __launch_bounds__(kBlock) void csr_stream(int num_blocks, int per_xcd,
                                                     const int4 *__restrict__ desc,
                                                     const int *__restrict__ row_ptr,
                                                     const int *__restrict__ col,
                                                     const T *__restrict__ val,
                                                     const T *__restrict__ x, T *__restrict__ y,
                                                     T *__restrict__ partial) {
    using V2 = typename vec2<T>::type;
    constexpr int kUnits = kStreamCap / kStreamUnit;
    __shared__ T prod[kStreamCap];
    __shared__ T wave_part[kBlock / 64];

    const int b = XCD ? xcd_contiguous(blockIdx.x, per_xcd) : (int)blockIdx.x;
    if (b >= num_blocks) return;  // whole workgroup leaves together
    const int t = threadIdx.x;
    const int4 d0 = desc[b];
    const int4 d1 = desc[b + 1];
    const int n0 = d0.y, n1 = d1.y;

    if (d0.x & kLongFlag) {
        // one piece of a long row: strided register accumulation, no staging
        T acc = 0;
        for (int e = (n0 & ~1) + 2 * t; e < n1; e += kStreamUnit) {
            const v2i c = stream_load<NT>(reinterpret_cast<const v2i *>(col + e));
            const V2 v = stream_load<NT>(reinterpret_cast<const V2 *>(val + e));
            const T p0 = e >= n0 ? v.x * x[c.x] : T(0);
            const T p1 = e + 1 < n1 ? v.y * x[c.y] : T(0);
            acc += p0;
            acc += p1;
        }
        acc = group_sum<64>(acc);
        if ((t & 63) == 0) wave_part[t >> 6] = acc;
        __syncthreads();
        if (t == 0) {
            T s = wave_part[0];
            for (int w = 1; w < kBlock / 64; ++w) s += wave_part[w];
            partial[d0.z] = s;
        }
        return;
    }

    const int r0 = d0.x, r1 = d1.x & ~kLongFlag;
    const int nrows = r1 - r0;
    const int base = n0 & ~1;

    // lanes per row for the LDS sum: the largest power of two that still
    // gives every row of the block its own lane group in one pass
    int lanes = 1;
    if (nrows <= kBlock / 2) {
        const int q = kBlock / (nrows > 0 ? nrows : 1);
        lanes = 1 << (31 - __clz(q));
        if (lanes > 64) lanes = 64;
    }
    const int rows_per_pass = kBlock / lanes;
    const int my_row = t / lanes, my_lane = t % lanes;

    // row extents of the first pass: issued before the stream so their latency
    // overlaps it
    int seg_lo = 0, seg_hi = 0;
    if (my_row < nrows) {
        seg_lo = row_ptr[r0 + my_row] - base;
        seg_hi = row_ptr[r0 + my_row + 1] - base;
    }

    // stage: all loads of the block are independent and issued back to back
    v2i c[kUnits];
    V2 v[kUnits];
#pragma unroll
    for (int u = 0; u < kUnits; ++u) {
        const int e = base + u * kStreamUnit + 2 * t;
        if (e < n1) {
            c[u] = stream_load<NT>(reinterpret_cast<const v2i *>(col + e));
            v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e));
        } else {
            c[u] = v2i{0, 0};
            v[u] = V2{0, 0};
        }
    }
#pragma unroll
    for (int u = 0; u < kUnits; ++u) {
        V2 p;
        p.x = v[u].x * x[c[u].x];
        p.y = v[u].y * x[c[u].y];
        *reinterpret_cast<V2 *>(&prod[u * kStreamUnit + 2 * t]) = p;
    }
    __syncthreads();

    // sum the rows out of LDS; all lanes stay in the loop (group_sum needs them)
    for (int first = 0; first < nrows; first += rows_per_pass) {
        const int row = first + my_row;
        if (first > 0) {
            seg_lo = seg_hi = 0;
            if (row < nrows) {
                seg_lo = row_ptr[r0 + row] - base;
                seg_hi = row_ptr[r0 + row + 1] - base;
            }
        }
        T acc = 0;
        for (int k = seg_lo + my_lane; k < seg_hi; k += lanes) acc += prod[k];
        acc = group_sum_rt(acc, lanes);
        if (my_lane == 0 && row < nrows) y[r0 + row] = acc;
    }
}

// One wavefront per long row: add its pieces in slot order.
template <typename T>
__global__ __launch_bounds__(64) void csr_long_finish(int count, const int4 *__restrict__ rows,
                                                      const T *__restrict__ partial,
                                                      T *__restrict__ y) {
    const int i = blockIdx.x;
    if (i >= count) return;
    const int4 d = rows[i];  // x = row, y = first slot, z = pieces
    T acc = 0;
    for (int k = threadIdx.x; k < d.z; k += 64) acc += partial[d.y + k];
    acc = group_sum<64>(acc);
    if (threadIdx.x == 0) y[d.x] = acc;
}

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

}  // namespace spmv
