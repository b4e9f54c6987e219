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
//                    The default when the matrix gets no x-window plan.
//   csr_stream_local the same blocks with the x lines a block touches copied into
//                    LDS by full-width loads and 16-bit local column slots instead
//                    of gathers: the default wherever upload finds a plan, and the
//                    kernel that reaches the HBM roofline on stencil matrices
//                    (nlpkkt-like: 83-89 % of 8 TB/s).
//   csr_long_pieces / csr_long_finish   rows longer than a block's stage.
// Ablations and the other stream variants: csr_kernels_experimental.hpp.
//
// The path is a bandwidth-bound gather (0.17 flop/byte in fp64): no MFMA.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wave_ops.hpp"

namespace spmv {

constexpr int kBlock = 256;          // threads per workgroup (4 wavefronts)
constexpr int kStreamCapMax = 8192;  // largest nnz stage per workgroup (64 KiB of fp64 products)
constexpr int kStreamUnit = 2 * kBlock;  // nnz one pass of the workgroup covers
constexpr int kStreamRowsCap = 1024; // rows per workgroup (bounds the empty-row case)
constexpr int kLongPiece = 8192;     // nnz per workgroup when one row is split
constexpr int kBaseMask = ~3;        // a block is staged from the multiple-of-4 entry at or below its first
                                     // (16-byte alignment of the column loads / LDS-DMA)

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

// x[c] with a wave-uniform base and a 32-bit byte offset: the compiler emits the
// `global_load v, v_off, s[base:base+1]` form (one address VGPR per lane instead of
// two and no 64-bit VALU address math).  The gathers are what keeps the texture
// addresser busy in these kernels, so their address payload matters.  Needs
// N * sizeof(T) < 4 GiB, which upload checks.
template <typename T>
__device__ __forceinline__ T gather(const T *__restrict__ x, int c) {
    const unsigned off = (unsigned)c * (unsigned)sizeof(T);
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(x) + off);
}

// Workgroup ids are dealt round-robin over the 8 XCDs, each with its own L2.  XCD x (= id & 7) takes
// runs of `chunk` consecutive blocks, the eight XCDs sit on eight neighbouring
// runs.  chunk = 0 keeps dispatch order (neighbouring blocks on different XCDs:
// every L2 sees the x window of ALL resident blocks); a chunk of a few hundred
// blocks gives each L2 only its own blocks' x window while the chip as a whole
// still streams through one moving window of the matrix (DRAM page locality).
__device__ __forceinline__ int xcd_chunked(int id, int chunk) {
    if (chunk <= 0) return id;
    const int xcd = id & 7, seq = id >> 3;
    return ((seq / chunk) * 8 + xcd) * chunk + seq % chunk;
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
    for (int e = row_ptr[r]; e < stop; ++e) acc += val[e] * gather(x, col[e]);
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
            for (int e = begin + lane; e < stop; e += L) acc += val[e] * gather(x, col[e]);
        } else {
            for (int e = (begin & ~1) + 2 * lane; e < stop; e += 2 * L) {
                const v2i c = stream_load<NT>(reinterpret_cast<const v2i *>(col + e));
                const V2 v = stream_load<NT>(reinterpret_cast<const V2 *>(val + e));
                const T p0 = e >= begin ? v.x * gather(x, c.x) : T(0);
                const T p1 = e + 1 < stop ? v.y * gather(x, c.y) : T(0);
                acc += p0;
                acc += p1;
            }
        }
    }
    acc = group_sum<L>(acc);  // every lane of the wavefront takes part
    if (lane == 0 && r < M) y[r] = acc;
}

// -------------------------------------------------------------------- stream
// Upload cuts the rows into workgroup-sized blocks (csr_build_blocks): block b
// is desc[b] = {first row, first entry, rows, end entry}; its entries fit the LDS stage
// (CAP).  A row longer than the stage is left out of the blocks and handled by
// csr_long_pieces / csr_long_finish (pieces of kLongPiece entries whose partial
// sums are added in slot order, so results do not depend on scheduling: no
// atomics).
//
// All stream kernels stage with unconditional loads: col/val carry >= CAP
// entries of zero padding behind the last nonzero; entries past a block's end
// belong to later rows and their products land in LDS slots no row of the block
// reads.

// Sum prod[lo + lane], prod[lo + lane + lanes], ... below hi with four
// independent accumulators so that four LDS reads are in flight per lane
// instead of a read -> wait -> add chain.
template <typename T>
__device__ __forceinline__ T lds_strided_sum(const T *prod, int lo, int hi, int lane, int lanes) {
    T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int k = lo + lane;
    for (; k + 3 * lanes < hi; k += 4 * lanes) {
        a0 += prod[k];
        a1 += prod[k + lanes];
        a2 += prod[k + 2 * lanes];
        a3 += prod[k + 3 * lanes];
    }
    for (; k < hi; k += lanes) a0 += prod[k];
    return (a0 + a1) + (a2 + a3);
}

// lanes per row for the LDS sum: the largest power of two that still gives
// every row of the block its own lane group in one pass
template <int BLOCK>
__device__ __forceinline__ int lanes_for_rows(int nrows) {
    if (nrows > BLOCK / 2 || nrows <= 0) return 1;
    const int l = 1 << (31 - __clz(BLOCK / nrows));
    return l > 64 ? 64 : l;
}

// Rows [r0, r0 + nrows) out of the staged products: lane groups of `lanes`.
template <typename T, int BLOCK>
__device__ __forceinline__ void sum_rows_from_lds(const T *prod, const int *__restrict__ row_ptr,
                                                  T *__restrict__ y, int r0, int nrows, int base,
                                                  int lanes, int seg_lo, int seg_hi) {
    const int t = threadIdx.x;
    const int rows_per_pass = BLOCK / lanes;
    const int my_row = t / lanes, my_lane = t % lanes;
    for (int first = 0; first < nrows; first += rows_per_pass) {  // all lanes stay in the loop
        const int row = first + my_row;
        if (first > 0) {
            seg_lo = seg_hi = 0;
            if (row < nrows) {
                seg_lo = row_ptr[r0 + row] - base;
                seg_hi = row_ptr[r0 + row + 1] - base;
            }
        }
        T acc = lds_strided_sum(prod, seg_lo, seg_hi, my_lane, lanes);
        acc = group_sum_rt(acc, lanes);
        if (my_lane == 0 && row < nrows) y[r0 + row] = acc;
    }
}

// Row extents of every pass of the row-sum phase, loaded at kernel start so that they ride along
// with the stream instead of costing a dependent global round trip per pass.  More than one pass
// only happens with one lane per row (a block of > BLOCK / 2 rows: matrices with very short rows).
// PRELOAD = false keeps only the first pass in registers (12 VGPRs fewer: the difference between 7
// and 6 resident workgroups per CU for the x-window kernel) and fetches later passes when it gets
// there; the launcher picks PRELOAD for handles whose mean row is shorter than stage / BLOCK.
template <int BLOCK, bool PRELOAD>
struct row_extents {
    static constexpr int kPasses = PRELOAD ? kStreamRowsCap / BLOCK : 1;
    int lo[kPasses], hi[kPasses];
};

template <int BLOCK, bool PRELOAD>
__device__ __forceinline__ row_extents<BLOCK, PRELOAD> load_row_extents(const int *__restrict__ row_ptr, int r0,
                                                                        int nrows, int lanes, int base) {
    row_extents<BLOCK, PRELOAD> e;
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < row_extents<BLOCK, PRELOAD>::kPasses; ++p) {
        const int row = p == 0 ? t / lanes : p * BLOCK + t;  // passes > 0 exist only for lanes == 1
        // raw row_ptr values: nothing is computed from them here, so nothing waits for them before
        // the stream loads have been issued (base is subtracted where they are used)
        e.lo[p] = e.hi[p] = base;
        if ((p == 0 || lanes == 1) && row < nrows) {
            e.lo[p] = row_ptr[r0 + row];
            e.hi[p] = row_ptr[r0 + row + 1];
        }
    }
    return e;
}

template <typename T, int BLOCK, bool PRELOAD>
__device__ __forceinline__ void sum_rows_from_lds(const T *prod, const int *__restrict__ row_ptr, T *__restrict__ y,
                                                  int r0, int nrows, int base, int lanes,
                                                  const row_extents<BLOCK, PRELOAD> &e) {
    if constexpr (PRELOAD) {
        const int t = threadIdx.x;
        const int my_lane = t % lanes;
#pragma unroll
        for (int p = 0; p < row_extents<BLOCK, PRELOAD>::kPasses; ++p) {
            const int row = p == 0 ? t / lanes : p * BLOCK + t;
            if (p > 0 && (lanes != 1 || p * BLOCK >= nrows)) break;  // wave-uniform
            T acc = lds_strided_sum(prod, e.lo[p] - base, e.hi[p] - base, my_lane, lanes);
            acc = group_sum_rt(acc, lanes);
            if (my_lane == 0 && row < nrows) y[r0 + row] = acc;
        }
    } else {
        sum_rows_from_lds<T, BLOCK>(prod, row_ptr, y, r0, nrows, base, lanes, e.lo[0] - base, e.hi[0] - base);
    }
}

// NU units of one block, straight-line: all (col, val) pair loads, then all gathers, then the
// products into LDS.  The unit count of a block is wave-uniform, so the caller's switch is scalar.
template <typename T, bool NT, int BLOCK, int NU>
__device__ __forceinline__ void stream_stage(T *prod, const int *__restrict__ col, const T *__restrict__ val,
                                             const T *__restrict__ x, int e_first) {
    using V2 = typename vec2<T>::type;
    constexpr int kUnit = 2 * BLOCK;
    const int t = threadIdx.x;
    v2i c[NU];
    V2 v[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        c[u] = stream_load<NT>(reinterpret_cast<const v2i *>(col + e_first + u * kUnit));
        v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e_first + u * kUnit));
    }
    T xv[2 * NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        xv[2 * u] = gather(x, c[u].x);
        xv[2 * u + 1] = gather(x, c[u].y);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        V2 p;
        p.x = v[u].x * xv[2 * u];
        p.y = v[u].y * xv[2 * u + 1];
        *reinterpret_cast<V2 *>(&prod[u * kUnit + 2 * t]) = p;
    }
}

// One workgroup per block: stage products, sum rows.
template <typename T, bool NT, int CAP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void csr_stream(int num_blocks, int xcd_chunk,
                                                    const int4 *__restrict__ desc,
                                                    const int *__restrict__ row_ptr,
                                                    const int *__restrict__ col,
                                                    const T *__restrict__ val,
                                                    const T *__restrict__ x, T *__restrict__ y) {
    using V2 = typename vec2<T>::type;
    constexpr int kUnit = 2 * BLOCK;  // entries one pass of the workgroup covers
    constexpr int kUnits = CAP / kUnit;
    __shared__ T prod[CAP];

    const int b = xcd_chunked(blockIdx.x, xcd_chunk);
    if (b >= num_blocks) return;  // whole workgroup leaves together
    const int t = threadIdx.x;
    const int4 d = desc[b];
    const int r0 = d.x, nrows = d.z;
    const int base = d.y & kBaseMask;

    // row extents of the first pass go out first so they are back early
    const int lanes = lanes_for_rows<BLOCK>(nrows);
    int seg_lo = 0, seg_hi = 0;
    if (t / lanes < nrows) {
        seg_lo = row_ptr[r0 + t / lanes];
        seg_hi = row_ptr[r0 + t / lanes + 1];
    }

    const int e_first = base + 2 * t;
    const int units = (d.w - base + kUnit - 1) / kUnit;  // wave-uniform
    if (units == kUnits) {
        // a full block: everything straight-line
        v2i c[kUnits];
        V2 v[kUnits];
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            c[u] = stream_load<NT>(reinterpret_cast<const v2i *>(col + e_first + u * kUnit));
            v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e_first + u * kUnit));
        }
        T xv[2 * kUnits];
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            xv[2 * u] = gather(x, c[u].x);
            xv[2 * u + 1] = gather(x, c[u].y);
        }
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            V2 p;
            p.x = v[u].x * xv[2 * u];
            p.y = v[u].y * xv[2 * u + 1];
            *reinterpret_cast<V2 *>(&prod[u * kUnit + 2 * t]) = p;
        }
    } else {
        // a block cut short by its row cap or by the end of the matrix: only the
        // units it really has (scalar loop, no wasted traffic)
        for (int u = 0; u < units; ++u) {
            const v2i c = stream_load<NT>(reinterpret_cast<const v2i *>(col + e_first + u * kUnit));
            const V2 v = stream_load<NT>(reinterpret_cast<const V2 *>(val + e_first + u * kUnit));
            V2 p;
            p.x = v.x * gather(x, c.x);
            p.y = v.y * gather(x, c.y);
            *reinterpret_cast<V2 *>(&prod[u * kUnit + 2 * t]) = p;
        }
    }
    __syncthreads();
    sum_rows_from_lds<T, BLOCK>(prod, row_ptr, y, r0, nrows, base, lanes, seg_lo - base, seg_hi - base);
}

// csr_stream for matrices with very short rows (mean row < stage / 256): most blocks are then cut by
// the row cap, not by the stage, so (1) every unit count gets straight-line staging instead of the
// one-unit-at-a-time loop csr_stream keeps for its rare short blocks, and (2) the row extents of all
// passes of the row-sum phase (up to 1024 rows, one lane each) are loaded up front.  Road-like
// 3-per-row matrix: 316 -> 236 us; kept apart from csr_stream because the extra code costs the full
// blocks of gather-bound matrices 4 % (profiles/r1b_ab_short_row_kernel.txt).
template <typename T, bool NT, int CAP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void csr_stream_short(int num_blocks, int xcd_chunk,
                                                    const int4 *__restrict__ desc,
                                                    const int *__restrict__ row_ptr,
                                                    const int *__restrict__ col,
                                                    const T *__restrict__ val,
                                                    const T *__restrict__ x, T *__restrict__ y) {
    constexpr int kUnit = 2 * BLOCK;  // entries one pass of the workgroup covers
    constexpr int kUnits = CAP / kUnit;
    static_assert(kUnits >= 1 && kUnits <= 8, "stage of 1..8 units");
    __shared__ T prod[CAP];

    const int b = xcd_chunked(blockIdx.x, xcd_chunk);
    if (b >= num_blocks) return;  // whole workgroup leaves together
    const int t = threadIdx.x;
    const int4 d = desc[b];
    const int r0 = d.x, nrows = d.z;
    const int base = d.y & kBaseMask;

    // row extents go out first so they are back early
    const int lanes = lanes_for_rows<BLOCK>(nrows);
    const row_extents<BLOCK, true> ext = load_row_extents<BLOCK, true>(row_ptr, r0, nrows, lanes, base);

    const int e_first = base + 2 * t;
    // only the units the block really has (a block cut short by its row cap -- matrices with very
    // short rows -- or by the end of the matrix streams no padding)
    const int units = (d.w - base + kUnit - 1) / kUnit;  // wave-uniform, 0..kUnits
    if (units == kUnits) {
        stream_stage<T, NT, BLOCK, kUnits>(prod, col, val, x, e_first);  // the common case first
    } else {
        switch (units) {
            case 1: if constexpr (kUnits > 1) stream_stage<T, NT, BLOCK, 1>(prod, col, val, x, e_first); break;
            case 2: if constexpr (kUnits > 2) stream_stage<T, NT, BLOCK, 2>(prod, col, val, x, e_first); break;
            case 3: if constexpr (kUnits > 3) stream_stage<T, NT, BLOCK, 3>(prod, col, val, x, e_first); break;
            case 4: if constexpr (kUnits > 4) stream_stage<T, NT, BLOCK, 4>(prod, col, val, x, e_first); break;
            case 5: if constexpr (kUnits > 5) stream_stage<T, NT, BLOCK, 5>(prod, col, val, x, e_first); break;
            case 6: if constexpr (kUnits > 6) stream_stage<T, NT, BLOCK, 6>(prod, col, val, x, e_first); break;
            case 7: if constexpr (kUnits > 7) stream_stage<T, NT, BLOCK, 7>(prod, col, val, x, e_first); break;
            default: break;  // a block of empty rows
        }
    }
    __syncthreads();
    sum_rows_from_lds<T, BLOCK, true>(prod, row_ptr, y, r0, nrows, base, lanes, ext);
}

// --------------------------------------------------------- stream, local x
// csr_stream pays for its gathers in the texture addresser: a 64-lane gather over a
// stencil-like row touches 16-28 different 128-byte lines of x and occupies the address
// path for ~60 cycles (profiles/r1_ubench_gather_cost.txt), which on the nlpkkt-like matrix
// adds up to about as much time as the HBM stream itself.  Yet the x lines one block needs
// are few: ~75 neighbouring stencil rows share their 28 column offsets, so 2048 entries
// touch only ~160 lines.  Upload therefore writes, per block, the sorted list of the x lines
// it touches and replaces each 32-bit column by a 16-bit slot in that list
// (rank * elements-per-line + column % elements-per-line).  The kernel copies the listed
// lines into LDS with full-width coalesced loads (8 lines per wave instruction instead of
// 16-28 lines for a quarter of the payload) and gathers from LDS.  HBM traffic drops too:
// 2 bytes of index per entry instead of 4, plus 4 bytes per listed line.
//
// x must be 128-byte aligned: whole lines are read, the tail of the last one may lie behind
// x[N - 1] (inside the same aligned line, hence inside the allocation's last page).
constexpr int kLineBytes = 128;
constexpr int kLocalLinesMax = 256;  // lines per block: rank fits the 16-bit slot for fp64 and fp32
constexpr int kLocalLineQuantum = kBlock / 8;  // lines one pass of the workgroup stages (16 B per lane)

template <typename T, bool NT, int CAP, int ROUNDS>
__device__ __forceinline__ void local_stage_full(T *stage, const int *__restrict__ my_lines, int last_line,
                                                 const unsigned short *__restrict__ lcol,
                                                 const T *__restrict__ val, const T *__restrict__ x,
                                                 int e_first, bool no_slots = false) {
    using V2 = typename vec2<T>::type;
    constexpr int kUnit = 2 * kBlock, kUnits = CAP / kUnit;
    const int t = threadIdx.x;
    // the line list first: the x loads depend on it, and loads come back in issue order
    int line[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        // lanes behind the end of the list repeat its last line: one more hit on a line that is
        // being fetched anyway instead of up to 31 extra lines
        const int at = k * kLocalLineQuantum + (t >> 3);
        line[k] = stream_load<NT>(my_lines + (k == ROUNDS - 1 ? min(at, last_line) : at));
    }
    unsigned c[kUnits];
    V2 v[kUnits];
#pragma unroll
    for (int u = 0; u < kUnits; ++u) {
        // (no_slots, measurement only -- the STAMP instantiation's probe: the slot stream is not read at all; what the
        // kernel would cost if a block's slots came from a per-block pattern instead of a 16-bit word per entry)
        c[u] = no_slots ? (unsigned)(t & 15) * 0x10001u : stream_load<NT>(reinterpret_cast<const unsigned *>(lcol + e_first + u * kUnit));
        v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e_first + u * kUnit));
    }
    // keep the HBM stream ahead of the wait for the list (the scheduler would otherwise sink
    // the value loads behind the x loads, i.e. behind one full memory latency)
    __builtin_amdgcn_sched_barrier(0);
    uint4 xl[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const unsigned off = (unsigned)line[k] * (unsigned)kLineBytes + (unsigned)(t & 7) * 16u;
        xl[k] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(x) + off);
    }
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k)
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(stage) + (k * kBlock + t) * 16) = xl[k];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kUnits; ++u) {
        v[u].x *= stage[c[u] & 0xffffu];
        v[u].y *= stage[c[u] >> 16];
    }
    __syncthreads();  // the products take the place of the staged lines
#pragma unroll
    for (int u = 0; u < kUnits; ++u) *reinterpret_cast<V2 *>(&stage[u * kUnit + 2 * t]) = v[u];
}

// ---- slot PATTERNS (round 3).  The kernel is bound by its loads in flight, not by its bytes (12-bit slots: 4.6 % fewer
// bytes, no time gained; the slot stream not read at all: 181 -> 145 us on the nlpkkt-like matrix).  And the slot stream
// is redundant where the matrix is a stencil: consecutive rows of a block have the SAME slots shifted by a constant (92 %
// of the nlpkkt-like matrix's entries, 97 % of the FEM-shaped one's, lie in such rows).  A pattern plan stores a row's
// slots only where they are not the previous row's plus a constant (the block's pattern table, ptab), and per row where
// its pattern starts in that table and its shift: 4 bytes per row instead of 2 per entry.  Every lane group fetches its
// row's pattern (16-byte groups of 8 slots; neighbouring rows share them: cache hits) and writes the row's slots --
// pattern + shift -- into an LDS array, from which the products read their slot pairs instead of from memory: the same
// slots, the same products, the same bits.
struct pat_ctx {
    unsigned short *slots;          // LDS [CAP + 8]: the block's slots, entry order, from `base` on
    const uint4 *ptab8;             // the block's pattern table in memory, groups of 8 slots (a row's pattern starts at a group)
    const unsigned *rinfo;          // per row: first group of its pattern in the block's table | shift << 16
    const int *row_ptr;             // CSR: the rows' extents ...
    const unsigned *row_seg;        // ... HLL (hll_lds_local): first slot in the window | slots << 16 per row (base = 0 then)
    int r0, nrows, base, first, end, lanes, seg_lo, seg_hi;  // seg_lo / seg_hi: the first row's extent, RAW (CSR: two row_ptr
                                    // values; HLL: its row_seg word in seg_lo) -- decoded where it is used, behind the stream
    unsigned ri;                    // rinfo of this lane group's first row
};
// a row's first slot (as the slots array counts: from `base`) and its length, out of the raw extent
template <bool HLL>
__device__ __forceinline__ void pat_row_extent(const pat_ctx &c, int raw_lo, int raw_hi, int &lo, int &len) {
    if constexpr (HLL) {
        lo = (int)((unsigned)raw_lo & 0xffffu);
        len = (int)((unsigned)raw_lo >> 16);
    } else {
        lo = raw_lo - c.base;
        len = raw_hi - raw_lo;
    }
}

// slots[at .. at + 8) (those below len) = the pattern group + shift.  Written as PAIRS wherever a pair lies on a 4-byte
// boundary of the LDS array (a row starts at an even or an odd entry of its block): four or five LDS writes per group
// instead of eight -- these writes are most of what a pattern plan costs.
typedef unsigned short pat_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pat_shifted(unsigned pair, unsigned shift2) {  // both halves + shift (v_pk_add_u16)
    const pat_u16x2 a = __builtin_bit_cast(pat_u16x2, pair), b = __builtin_bit_cast(pat_u16x2, shift2);
    return __builtin_bit_cast(unsigned, (pat_u16x2)(a + b));
}
__device__ __forceinline__ void pat_write_group(unsigned short *slots, int at, int j0, int len, const uint4 p, int shift) {
    const unsigned shift2 = ((unsigned)shift & 0xffffu) * 0x10001u;
    const unsigned w[4] = {pat_shifted(p.x, shift2), pat_shifted(p.y, shift2), pat_shifted(p.z, shift2), pat_shifted(p.w, shift2)};
    const int left = len - j0;  // slots of this group that exist (>= 1)
    if ((at & 1) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (2 * k + 1 < left) *reinterpret_cast<unsigned *>(slots + at + 2 * k) = w[k];
            else if (2 * k < left) slots[at + 2 * k] = (unsigned short)w[k];
        }
    } else {
        slots[at] = (unsigned short)w[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // the pairs (1, 2), (3, 4), (5, 6) of the group
            const unsigned pair = (w[k] >> 16) | (w[k + 1] << 16);
            if (2 * k + 2 < left) *reinterpret_cast<unsigned *>(slots + at + 2 * k + 1) = pair;
            else if (2 * k + 1 < left) slots[at + 2 * k + 1] = (unsigned short)pair;
        }
        if (7 < left) slots[at + 7] = (unsigned short)(w[3] >> 16);
    }
}

// the rows of the block's FIRST pass of lane groups: up to two pattern groups per lane are loaded here (the caller lets
// them queue behind the value stream) ...
template <bool HLL = false>
__device__ __forceinline__ void pat_load_first(const pat_ctx c, uint4 (&p)[2]) {
    int lo, len;
    pat_row_extent<HLL>(c, c.seg_lo, c.seg_hi, lo, len);
    const int my_lane = threadIdx.x % c.lanes, off8 = (int)(c.ri & 0xffffu);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int g = my_lane + k * c.lanes;
        p[k] = c.ptab8[off8 + (g * 8 < len ? g : 0)];
    }
}
// ... and written here, with whatever is left: longer rows, further passes of rows (blocks of very short rows), and zeros
// for the halves of the lanes' pairs that lie outside the block (in front of its first entry, behind its last)
template <int BLOCK, int CAP, bool HLL = false>
__device__ __forceinline__ void pat_expand_slots(const pat_ctx c, const uint4 (&p)[2]) {
    const int t = threadIdx.x;
    const int my_row = t / c.lanes, my_lane = t % c.lanes, rows_per_pass = BLOCK / c.lanes;
    int raw_lo = c.seg_lo, raw_hi = c.seg_hi;
    unsigned ri = c.ri;
    for (int first = 0; first < c.nrows; first += rows_per_pass) {  // all lanes stay in the loop
        const int row = first + my_row;
        if (first > 0) {
            raw_lo = raw_hi = 0;
            ri = 0;
            if (row < c.nrows) {
                if constexpr (HLL) {
                    raw_lo = (int)c.row_seg[c.r0 + row];
                } else {
                    raw_lo = c.row_ptr[c.r0 + row];
                    raw_hi = c.row_ptr[c.r0 + row + 1];
                }
                ri = c.rinfo[c.r0 + row];
            }
        }
        int lo, len;
        pat_row_extent<HLL>(c, raw_lo, raw_hi, lo, len);
        if (first > 0 && row >= c.nrows) len = 0;
        const int off8 = (int)(ri & 0xffffu), shift = (int)(short)(ri >> 16);
        int g = my_lane;
        if (first == 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k, g += c.lanes)
                if (g * 8 < len) pat_write_group(c.slots, lo + g * 8, g * 8, len, p[k], shift);
        }
        for (; g * 8 < len; g += c.lanes) pat_write_group(c.slots, lo + g * 8, g * 8, len, c.ptab8[off8 + g], shift);
    }
    constexpr int kUnit = 2 * BLOCK;
#pragma unroll
    for (int u = 0; u < CAP / kUnit; ++u) {
        const int e = c.base + u * kUnit + 2 * t;
        if (e < c.first || e >= c.end) c.slots[e - c.base] = 0;
        if (e + 1 < c.first || e + 1 >= c.end) c.slots[e + 1 - c.base] = 0;
    }
}

template <typename T, bool NT, int CAP, int ROUNDS, bool HLL = false>
__device__ __forceinline__ void local_stage_full_pat(T *stage, const int *__restrict__ my_lines, int last_line,
                                                     const T *__restrict__ val, const T *__restrict__ x, int e_first,
                                                     const pat_ctx pc) {
    using V2 = typename vec2<T>::type;
    constexpr int kUnit = 2 * kBlock, kUnits = CAP / kUnit;
    const int t = threadIdx.x;
    int line[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int at = k * kLocalLineQuantum + (t >> 3);
        line[k] = stream_load<NT>(my_lines + (k == ROUNDS - 1 ? min(at, last_line) : at));
    }
    V2 v[kUnits];
#pragma unroll
    for (int u = 0; u < kUnits; ++u) v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e_first + u * kUnit));
    __builtin_amdgcn_sched_barrier(0);
    // behind the stream: the rows' pattern groups (their addresses came with the row extents, ahead of the stream),
    // then the x lines
    uint4 pg[2];
    pat_load_first<HLL>(pc, pg);
    uint4 xl[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const unsigned off = (unsigned)line[k] * (unsigned)kLineBytes + (unsigned)(t & 7) * 16u;
        xl[k] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(x) + off);
    }
    __builtin_amdgcn_sched_barrier(0);
    pat_expand_slots<kBlock, CAP, HLL>(pc, pg);
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k)
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(stage) + (k * kBlock + t) * 16) = xl[k];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kUnits; ++u) {
        const unsigned c = *reinterpret_cast<const unsigned *>(pc.slots + u * kUnit + 2 * t);
        v[u].x *= stage[c & 0xffffu];
        v[u].y *= stage[c >> 16];
    }
    __syncthreads();  // the products take the place of the staged lines
#pragma unroll
    for (int u = 0; u < kUnits; ++u) *reinterpret_cast<V2 *>(&stage[u * kUnit + 2 * t]) = v[u];
}

// STAMP (measurement only, spmv_hip_csr_stamp_blocks): every workgroup leaves {start, end} of the constant 100 MHz
// clock and the XCD it ran on in stamps[3 * block ..]; the product instantiation (STAMP = false) has no trace of it.
// PAT: a pattern plan (above) -- lcol is not read; pdesc[b] = {first element of block b's pattern table in ptab (a
// multiple of 8), elements in it}, rinfo per row, stage_bytes = where the LDS array of the slots begins.
template <typename T, bool NT, int CAP, bool STAMP = false, bool PAT = false>
__global__ __launch_bounds__(kBlock) void csr_stream_local(int num_blocks, int xcd_chunk,
                                                           const int *__restrict__ ids,
                                                           const int4 *__restrict__ desc,
                                                           const int2 *__restrict__ ldesc,
                                                           const int *__restrict__ lines,
                                                           const int *__restrict__ row_ptr,
                                                           const unsigned short *__restrict__ lcol,
                                                           const T *__restrict__ val,
                                                           const T *__restrict__ x, T *__restrict__ y,
                                                           unsigned long long *__restrict__ stamps = nullptr, int probe = 0,
                                                           const int2 *__restrict__ pdesc = nullptr,
                                                           const unsigned *__restrict__ rinfo = nullptr,
                                                           const unsigned short *__restrict__ ptab = nullptr, int stage_bytes = 0) {
        using V2 = typename vec2<T>::type;
    const bool no_slots = STAMP && (probe & 1);  // (measurement only; constant false in the product instantiation)
    unsigned long long t_start = 0;
    if constexpr (STAMP) t_start = __builtin_amdgcn_s_memrealtime();
    constexpr int kUnit = 2 * kBlock, kUnits = CAP / kUnit;
    // one LDS stage, used twice: first the x lines of the block, then (after every lane has
    // gathered its x values into registers) the CAP products.  max(CAP values, staged lines).
    extern __shared__ __attribute__((aligned(16))) unsigned char local_smem[];
    T *stage = reinterpret_cast<T *>(local_smem);

    const int at = xcd_chunked(blockIdx.x, xcd_chunk);
    if (at >= num_blocks) return;
    // ids (optional): a sub-list of the blocks -- the interior or the boundary blocks of a rank (N4 overlap)
    const int b = ids ? ids[at] : at;
    const int t = threadIdx.x;
    const int4 d = desc[b];
    const int2 ld = ldesc[b];
    const int r0 = d.x, nrows = d.z;
    const int base = d.y & kBaseMask;

    const int lanes = lanes_for_rows<kBlock>(nrows);
    // (only the first pass's row extents are loaded up front: keeping all passes in registers, as
    // csr_stream_short does, costs 12 VGPRs = one resident workgroup per CU here)
    int seg_lo = 0, seg_hi = 0;
    if (t / lanes < nrows) {
        seg_lo = row_ptr[r0 + t / lanes];
        seg_hi = row_ptr[r0 + t / lanes + 1];
    }

    const int e_first = base + 2 * t;
    const int units = (d.w - base + kUnit - 1) / kUnit;                       // wave-uniform
    const int rounds = (ld.y + kLocalLineQuantum - 1) / kLocalLineQuantum;    // wave-uniform, 1..8
    const int *my_lines = lines + ld.x;
    if constexpr (PAT) {
        const int2 pd = pdesc[b];
        pat_ctx pc;
        pc.slots = reinterpret_cast<unsigned short *>(local_smem + stage_bytes);
        pc.ptab8 = reinterpret_cast<const uint4 *>(ptab + pd.x);
        pc.rinfo = rinfo;
        pc.row_ptr = row_ptr;
        pc.row_seg = nullptr;
        pc.r0 = r0;
        pc.nrows = nrows;
        pc.base = base;
        pc.first = d.y;
        pc.end = d.w;
        pc.lanes = lanes;
        pc.seg_lo = seg_lo;
        pc.seg_hi = seg_hi;
        pc.ri = t / lanes < nrows ? rinfo[r0 + t / lanes] : 0u;
        if (units == kUnits) {
            switch (rounds) {
                case 1: local_stage_full_pat<T, NT, CAP, 1>(stage, my_lines, ld.y - 1, val, x, e_first, pc); break;
                case 2: local_stage_full_pat<T, NT, CAP, 2>(stage, my_lines, ld.y - 1, val, x, e_first, pc); break;
                case 3: local_stage_full_pat<T, NT, CAP, 3>(stage, my_lines, ld.y - 1, val, x, e_first, pc); break;
                case 4: local_stage_full_pat<T, NT, CAP, 4>(stage, my_lines, ld.y - 1, val, x, e_first, pc); break;
                case 5: local_stage_full_pat<T, NT, CAP, 5>(stage, my_lines, ld.y - 1, val, x, e_first, pc); break;
                case 6: local_stage_full_pat<T, NT, CAP, 6>(stage, my_lines, ld.y - 1, val, x, e_first, pc); break;
                case 7: local_stage_full_pat<T, NT, CAP, 7>(stage, my_lines, ld.y - 1, val, x, e_first, pc); break;
                default: local_stage_full_pat<T, NT, CAP, 8>(stage, my_lines, ld.y - 1, val, x, e_first, pc); break;
            }
        } else {
            // a block cut short: plain loops
            for (int k = 0; k < rounds; ++k) {
                const int line = my_lines[min(k * kLocalLineQuantum + (t >> 3), ld.y - 1)];
                const unsigned off = (unsigned)line * (unsigned)kLineBytes + (unsigned)(t & 7) * 16u;
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(stage) + (k * kBlock + t) * 16) =
                    *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(x) + off);
            }
            uint4 pg[2];
            pat_load_first(pc, pg);
            pat_expand_slots<kBlock, CAP>(pc, pg);
            __syncthreads();
            V2 p[kUnits];
#pragma unroll
            for (int u = 0; u < kUnits; ++u) {
                if (u < units) {
                    const unsigned c = *reinterpret_cast<const unsigned *>(pc.slots + u * kUnit + 2 * t);
                    p[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e_first + u * kUnit));
                    p[u].x *= stage[c & 0xffffu];
                    p[u].y *= stage[c >> 16];
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kUnits; ++u)
                if (u < units) *reinterpret_cast<V2 *>(&stage[u * kUnit + 2 * t]) = p[u];
        }
    } else if (units == kUnits) {
        switch (rounds) {
            case 1: local_stage_full<T, NT, CAP, 1>(stage, my_lines, ld.y - 1, lcol, val, x, e_first, no_slots); break;
            case 2: local_stage_full<T, NT, CAP, 2>(stage, my_lines, ld.y - 1, lcol, val, x, e_first, no_slots); break;
            case 3: local_stage_full<T, NT, CAP, 3>(stage, my_lines, ld.y - 1, lcol, val, x, e_first, no_slots); break;
            case 4: local_stage_full<T, NT, CAP, 4>(stage, my_lines, ld.y - 1, lcol, val, x, e_first, no_slots); break;
            case 5: local_stage_full<T, NT, CAP, 5>(stage, my_lines, ld.y - 1, lcol, val, x, e_first, no_slots); break;
            case 6: local_stage_full<T, NT, CAP, 6>(stage, my_lines, ld.y - 1, lcol, val, x, e_first, no_slots); break;
            case 7: local_stage_full<T, NT, CAP, 7>(stage, my_lines, ld.y - 1, lcol, val, x, e_first, no_slots); break;
            default: local_stage_full<T, NT, CAP, 8>(stage, my_lines, ld.y - 1, lcol, val, x, e_first, no_slots); break;
        }
    } else {
        // a block cut short (row cap, line cap, end of the matrix): plain loops
        for (int k = 0; k < rounds; ++k) {
            const int line = my_lines[min(k * kLocalLineQuantum + (t >> 3), ld.y - 1)];
            const unsigned off = (unsigned)line * (unsigned)kLineBytes + (unsigned)(t & 7) * 16u;
            *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(stage) + (k * kBlock + t) * 16) =
                *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(x) + off);
        }
        __syncthreads();
        V2 p[kUnits];
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            if (u < units) {
                const unsigned c = stream_load<NT>(reinterpret_cast<const unsigned *>(lcol + e_first + u * kUnit));
                p[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e_first + u * kUnit));
                p[u].x *= stage[c & 0xffffu];
                p[u].y *= stage[c >> 16];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kUnits; ++u)
            if (u < units) *reinterpret_cast<V2 *>(&stage[u * kUnit + 2 * t]) = p[u];
    }
    __syncthreads();
    sum_rows_from_lds<T, kBlock>(stage, row_ptr, y, r0, nrows, base, lanes, seg_lo - base, seg_hi - base);
    if constexpr (STAMP) {
        __syncthreads();
        if (threadIdx.x == 0) {
            stamps[3 * (size_t)b] = t_start;
            stamps[3 * (size_t)b + 1] = __builtin_amdgcn_s_memrealtime();
            // HW_REG_XCC_ID (20), bits 3:0: the XCD this workgroup ran on; next to it the dispatch id
            stamps[3 * (size_t)b + 2] = ((unsigned long long)blockIdx.x << 8) | (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xf);
        }
    }
}

// ----------------------------------------------------------------- long rows
// piece = {row, first entry, end entry, slot}: one workgroup accumulates the
// piece in registers (no staging) and stores its partial sum.
template <typename T, bool NT>
__global__ __launch_bounds__(kBlock) void csr_long_pieces(int count, const int4 *__restrict__ pieces,
                                                          const int *__restrict__ col,
                                                          const T *__restrict__ val,
                                                          const T *__restrict__ x,
                                                          T *__restrict__ partial) {
    using V2 = typename vec2<T>::type;
    constexpr int kUnits = 8;  // 4096 entries = 16 loads per lane in flight per trip
    __shared__ T wave_part[kBlock / 64];
    if ((int)blockIdx.x >= count) return;
    const int t = threadIdx.x;
    const int4 d = pieces[blockIdx.x];
    const int n0 = d.y, n1 = d.z;
    T a0 = 0, a1 = 0;
    int e0 = (n0 & ~1) + 2 * t;
    // whole trips: every entry of the trip lies inside [n0, n1) except possibly the
    // very first one of the piece (odd n0), which is masked
    for (; e0 - 2 * t + kUnits * kStreamUnit <= n1; e0 += kUnits * kStreamUnit) {
        v2i c[kUnits];
        V2 v[kUnits];
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            c[u] = stream_load<NT>(reinterpret_cast<const v2i *>(col + e0 + u * kStreamUnit));
            v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e0 + u * kStreamUnit));
        }
        T xv[2 * kUnits];
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            xv[2 * u] = gather(x, c[u].x);
            xv[2 * u + 1] = gather(x, c[u].y);
        }
        if (e0 < n0) v[0].x = T(0);  // only lane 0 of the first trip of an odd-start piece
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            a0 += v[u].x * xv[2 * u];
            a1 += v[u].y * xv[2 * u + 1];
        }
    }
    // remainder: bounded per entry
    for (; e0 < n1; e0 += kStreamUnit) {
        const v2i c = stream_load<NT>(reinterpret_cast<const v2i *>(col + e0));
        const V2 v = stream_load<NT>(reinterpret_cast<const V2 *>(val + e0));
        if (e0 >= n0) a0 += v.x * gather(x, c.x);
        if (e0 + 1 < n1) a1 += v.y * gather(x, c.y);
    }
    T acc = group_sum<64>(a0 + a1);
    if ((t & 63) == 0) wave_part[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        T s = wave_part[0];
        for (int w = 1; w < kBlock / 64; ++w) s += wave_part[w];
        partial[d.w] = s;
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


}  // namespace spmv
