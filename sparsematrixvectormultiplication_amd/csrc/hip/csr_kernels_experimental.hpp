// csr_kernels_experimental.hpp -- the stream-kernel variants that were built to find out why the
// gather kernel (csr_stream) stays ~35 % above the pure stream on stencil matrices, kept as the
// evidence behind DESIGN.md section 4 and selectable with spmv_hip_set_tuning("stream_kind", ...):
//   csr_probe        (kinds 10..17) ablation of csr_stream, measurement only (y is NOT A x)
//   csr_stream_pipe  (2)  persistent, two register stages
//   csr_stream_ring  (4)  loader wavefront + consumer wavefronts over an LDS ring (LDS-DMA)
//   csr_stream_walk  (1, 3) raw entries staged in LDS, lane = row
// None of them is a default; the product kernels are in csr_kernels.hpp.
#pragma once
#include "csr_kernels.hpp"

namespace spmv {

// ------------------------------------------------------------------- probe
// Ablation of csr_stream (measurement aid, results are NOT y = A x): what does each
// phase cost?  MODE bit 0: gather x (else x = 1), bit 2: gather from x[c & table_mask], bit 1: LDS stage + row sums (else
// lanes keep their products and one value per lane-pair is stored).
template <typename T, bool NT, int CAP, int MODE>
__global__ __launch_bounds__(kBlock) void csr_probe(int num_blocks, int xcd_chunk, int table_mask,
                                                    const int4 *__restrict__ desc,
                                                    const int *__restrict__ row_ptr,
                                                    const int *__restrict__ col,
                                                    const T *__restrict__ val,
                                                    const T *__restrict__ x, T *__restrict__ y) {
    using V2 = typename vec2<T>::type;
    constexpr int kUnits = CAP / kStreamUnit;
    __shared__ T prod[CAP];
    const int b = xcd_chunked(blockIdx.x, xcd_chunk);
    if (b >= num_blocks) return;
    const int t = threadIdx.x;
    const int4 d = desc[b];
    const int r0 = d.x, nrows = d.z;
    const int base = d.y & kBaseMask;
    const int lanes = lanes_for_rows<kBlock>(nrows);
    int seg_lo = 0, seg_hi = 0;
    if ((MODE & 2) && t / lanes < nrows) {
        seg_lo = row_ptr[r0 + t / lanes];
        seg_hi = row_ptr[r0 + t / lanes + 1];
    }
    v2i c[kUnits];
    V2 v[kUnits];
    const int e_first = base + 2 * t;
#pragma unroll
    for (int u = 0; u < kUnits; ++u) {
        c[u] = stream_load<NT>(reinterpret_cast<const v2i *>(col + e_first + u * kStreamUnit));
        v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e_first + u * kStreamUnit));
    }
    T xv[2 * kUnits];
#pragma unroll
    for (int u = 0; u < kUnits; ++u) {
        if (MODE & 4) {  // same instructions, indices folded into an 8 KiB (L1-resident) table
            xv[2 * u] = gather(x, c[u].x & table_mask);
            xv[2 * u + 1] = gather(x, c[u].y & table_mask);
        } else if (MODE & 1) {
            xv[2 * u] = gather(x, c[u].x);
            xv[2 * u + 1] = gather(x, c[u].y);
        } else {
            xv[2 * u] = T(c[u].x & 1);  // keeps the column loads alive
            xv[2 * u + 1] = T(c[u].y & 1);
        }
    }
    if (MODE & 2) {
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            V2 p;
            p.x = v[u].x * xv[2 * u];
            p.y = v[u].y * xv[2 * u + 1];
            *reinterpret_cast<V2 *>(&prod[u * kStreamUnit + 2 * t]) = p;
        }
        __syncthreads();
        sum_rows_from_lds<T, kBlock>(prod, row_ptr, y, r0, nrows, base, lanes, seg_lo - base, seg_hi - base);
    } else {
        T acc = 0;
#pragma unroll
        for (int u = 0; u < kUnits; ++u) acc += v[u].x * xv[2 * u] + v[u].y * xv[2 * u + 1];
        acc = group_sum<64>(acc);  // every lane's loads feed a stored value
        if ((t & 63) == 0 && (t >> 6) < nrows) y[r0 + (t >> 6)] = acc;
    }
}

// ------------------------------------------------------- stream, persistent
// csr_stream spends its life in three dependent waits (HBM stream -> L2/MALL
// gather -> LDS row sums) and only the first of them has HBM requests in
// flight, so with 16-32 waves per CU the bytes in flight hover around what
// Little's law asks for at ~6 TB/s.  Here a workgroup owns a contiguous run of
// blocks and keeps TWO register stages: while block b is gathered, multiplied
// and summed, the (col, val) stream of its next block is already in flight, so
// every resident wave always has its share of HBM requests outstanding.  The
// grid is sized to what is resident at once (persistent, grid-stride over the
// blocks) and there is no inter-workgroup communication (nothing to deadlock
// on).
template <typename T, bool NT, int CAP>
__global__ __launch_bounds__(kBlock) void csr_stream_pipe(int num_blocks, int xcd_chunk,
                                                          const int4 *__restrict__ desc,
                                                          const int *__restrict__ row_ptr,
                                                          const int *__restrict__ col,
                                                          const T *__restrict__ val,
                                                          const T *__restrict__ x,
                                                          T *__restrict__ y) {
    using V2 = typename vec2<T>::type;
    constexpr int kUnit = 2 * kBlock;
    constexpr int kUnits = CAP / kUnit;
    __shared__ T prod[CAP];
    const int t = threadIdx.x;

    // Workgroup w takes blocks w, w + G, w + 2G, ... (G = gridDim.x): at any
    // moment the resident workgroups sit on one moving window of the matrix,
    // which keeps DRAM pages open; giving each workgroup its own contiguous
    // run (thousands of independent streams) measured 20 % slower.
    // (with xcd_chunk > 0 the linear index is remapped so that an XCD's
    // workgroups share runs of blocks, see xcd_chunked)
    const int G = gridDim.x;
    int b = blockIdx.x;
    const int total = xcd_chunk > 0 ? (num_blocks + 8 * xcd_chunk - 1) / (8 * xcd_chunk) * (8 * xcd_chunk) : num_blocks;
    if (b >= total) return;
    auto block_of = [&](int linear) {  // descriptor of a linear index; past the end: an empty block
        const int real = xcd_chunked(linear, xcd_chunk);
        return real < num_blocks ? desc[real] : int4{0, 0, 0, 0};
    };

    auto issue = [&](v2i(&c)[kUnits], V2(&v)[kUnits], int first_entry) {
        const int e_first = (first_entry & kBaseMask) + 2 * t;
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            c[u] = stream_load<NT>(reinterpret_cast<const v2i *>(col + e_first + u * kUnit));
            v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e_first + u * kUnit));
        }
    };
    // consume one staged block; `prefetch` issues the next block's stream right
    // after this block's gathers so both are in flight together
    auto step = [&](const int4 d, v2i(&c)[kUnits], V2(&v)[kUnits], auto prefetch) {
        const int r0 = d.x, nrows = d.z;
        const int base = d.y & kBaseMask;
        const int lanes = lanes_for_rows<kBlock>(nrows);
        int seg_lo = 0, seg_hi = 0;
        if (t / lanes < nrows) {
            seg_lo = row_ptr[r0 + t / lanes];
            seg_hi = row_ptr[r0 + t / lanes + 1];
        }
        T xv[2 * kUnits];
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            xv[2 * u] = gather(x, c[u].x);
            xv[2 * u + 1] = gather(x, c[u].y);
        }
        prefetch();
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            V2 p;
            p.x = v[u].x * xv[2 * u];
            p.y = v[u].y * xv[2 * u + 1];
            *reinterpret_cast<V2 *>(&prod[u * kUnit + 2 * t]) = p;
        }
        __syncthreads();
        sum_rows_from_lds<T, kBlock>(prod, row_ptr, y, r0, nrows, base, lanes, seg_lo - base,
                                     seg_hi - base);
        __syncthreads();  // prod is rewritten by the next step
    };

    // Steps that prefetch do so unconditionally (a prefetch under a branch would
    // force the compiler to wait for ALL outstanding loads at the join, which
    // drains the very loads that are meant to stay in flight), so the last one
    // or two blocks of the run are peeled.
    v2i cA[kUnits], cB[kUnits];
    V2 vA[kUnits], vB[kUnits];
    int left = (total - 1 - b) / G + 1;
    int4 d = block_of(b);
    issue(cA, vA, d.y);
    while (left >= 3) {
        const int4 d1 = block_of(b + G);
        step(d, cA, vA, [&] { issue(cB, vB, d1.y); });
        const int4 d2 = block_of(b + 2 * G);
        step(d1, cB, vB, [&] { issue(cA, vA, d2.y); });
        d = d2;
        b += 2 * G;
        left -= 2;
    }
    if (left == 2) {
        const int4 d1 = block_of(b + G);
        step(d, cA, vA, [&] { issue(cB, vB, d1.y); });
        step(d1, cB, vB, [] {});
    } else {
        step(d, cA, vA, [] {});
    }
}

// ------------------------------------------------------------- stream, ring
// Loads return IN ORDER per wavefront, so in every kernel above a wave's gathers queue
// behind its own HBM stream (and a prefetched stream in front of the gathers makes them
// wait a full HBM latency).  Here the two kinds of traffic live in different waves:
//
//   wave 0      LOADER: streams each block's (col, val) and its row_ptr segment straight
//               into an LDS ring with LDS-DMA (global_load_lds: no VGPRs, lane-linear
//               image), three blocks ahead, counted vmcnt, one raw s_barrier per block;
//   waves 1-7   CONSUMERS: out of LDS only -- row extents, columns, values; their memory
//               queue holds nothing but x gathers and y stores.
//
// Consumer lane mapping: a wave takes tiles of RW neighbouring rows x SW slices
// (RW * SW = 64, SW from the block's mean row length so that a lane owns <= 8 entries);
// lane = slice * RW + row: neighbouring lanes are neighbouring rows at the same position
// (neighbouring x for stencil / FEM / banded matrices) and a row's partial sums are RW
// lanes apart, summed in registers (strided_sum) -- no LDS round trip, no extra barrier.
//
// One persistent workgroup per CU, grid-stride over the blocks.  Blocks for this kernel
// hold at most kRingRows rows (the staged row_ptr segment); rows longer than the stage go
// to the piece kernels as usual.
constexpr int kRingCap = 2048;
constexpr int kRingRows = 320;   // staged row_ptr entries per block: rows <= kRingRows - 1
constexpr int kRingBlock = 512;  // 1 loader + 7 consumer wavefronts

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <typename T>
struct ring_slot {
    T val[kRingCap];
    int col[kRingCap];
    int rp[kRingRows];
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {  // s_waitcnt vmcnt(N) only (gfx9 encoding)
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// SLOTS ring slots, AHEAD (= SLOTS - 1) blocks in flight ahead of the consumers:
// <4, 3> = 101 KiB of LDS, one workgroup per CU; <3, 2> = 76 KiB, two per CU (their
// consumer chains interleave).
template <typename T, bool NT, int SLOTS, int AHEAD>
__global__ __launch_bounds__(kRingBlock) void csr_stream_ring(int num_blocks,
                                                              const int4 *__restrict__ desc,
                                                              const int *__restrict__ row_ptr,
                                                              const int *__restrict__ col,
                                                              const T *__restrict__ val,
                                                              const T *__restrict__ x,
                                                              T *__restrict__ y) {
    static_assert(AHEAD == SLOTS - 1 && AHEAD >= 1 && AHEAD <= 3, "ring geometry");
    __shared__ ring_slot<T> ring[SLOTS];
    constexpr int kColDma = kRingCap * 4 / 1024;              // 1 KiB per wave-instruction
    constexpr int kValDma = kRingCap * (int)sizeof(T) / 1024;
    constexpr int kRpDma = kRingRows / 64;                    // 4-byte form: 256 B each
    constexpr int kDma = kColDma + kValDma + kRpDma;          // LDS-DMA instructions per block
    constexpr unsigned kAux = NT ? 2u : 0u;
    static_assert((AHEAD - 1) * kDma <= 63, "the blocks allowed to stay in flight must fit the vmcnt counter");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int G = gridDim.x;
    const int b0 = blockIdx.x;
    if (b0 >= num_blocks) return;  // the whole workgroup
    const int n = (num_blocks - 1 - b0) / G + 1;

    if (wave == 0) {
        // ------------------------------------------------------------ loader
        auto issue = [&](int k) {
            const int4 d = desc[b0 + k * G];
            ring_slot<T> &s = ring[k % SLOTS];
            const int base = d.y & kBaseMask;
            constexpr int kValPerLane = 16 / (int)sizeof(T);
#pragma unroll
            for (int i = 0; i < kColDma; ++i)
                __builtin_amdgcn_global_load_lds((glb_void *)(col + base + i * 256 + lane * 4),
                                                 (lds_void *)(s.col + i * 256), 16, 0, kAux);
#pragma unroll
            for (int i = 0; i < kValDma; ++i)
                __builtin_amdgcn_global_load_lds(
                    (glb_void *)(val + base + i * 64 * kValPerLane + lane * kValPerLane),
                    (lds_void *)(s.val + i * 64 * kValPerLane), 16, 0, kAux);
#pragma unroll
            for (int i = 0; i < kRpDma; ++i)
                __builtin_amdgcn_global_load_lds((glb_void *)(row_ptr + d.x + i * 64 + lane),
                                                 (lds_void *)(s.rp + i * 64), 4, 0, 0);
        };
        for (int k = 0; k < AHEAD && k < n; ++k) issue(k);
        for (int k = 0; k < n; ++k) {
            // block k has landed when at most the blocks issued after it are outstanding
            const int ahead = min(n - 1 - k, AHEAD - 1);
            if (AHEAD >= 3 && ahead >= 2) wait_vmcnt<(AHEAD >= 3 ? 2 : 0) * kDma>();
            else if (AHEAD >= 2 && ahead == 1) wait_vmcnt<(AHEAD >= 2 ? 1 : 0) * kDma>();
            else wait_vmcnt<0>();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();  // block k is ready; consumers are done with block k-1
            asm volatile("" ::: "memory");
            if (k + AHEAD < n) issue(k + AHEAD);  // into the slot of block k-1
        }
        return;
    }

    // ---------------------------------------------------------------- consumers
    const int cw = wave - 1;
    for (int k = 0; k < n; ++k) {
        const int4 d = desc[b0 + k * G];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const ring_slot<T> &s = ring[k % SLOTS];
        const int r0 = d.x, nrows = d.z;
        const int base = d.y & kBaseMask;
        const int mean = (d.w - d.y) / (nrows > 0 ? nrows : 1);
        int sw = 1;
        while (sw < 64 && sw * 8 < mean) sw <<= 1;  // slices per row: <= 8 entries per lane
        const int rw = 64 / sw;
        const int row_in = lane & (rw - 1), slice = lane / rw;
        const int tiles = (nrows + rw - 1) / rw;
        for (int tile = cw; tile < tiles; tile += kRingBlock / 64 - 1) {
            const int row = tile * rw + row_in;
            int lo = 0, hi = 0;
            if (row < nrows) {
                lo = s.rp[row] - base;
                hi = s.rp[row + 1] - base;
            }
            T a0 = 0, a1 = 0;
            for (int kk = lo + slice; __builtin_amdgcn_ballot_w64(kk < hi) != 0; kk += 8 * sw) {
                int cc[8];
                T xx[8], vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = kk + j * sw;
                    cc[j] = s.col[e < kRingCap ? e : kRingCap - 1];  // past the row: any staged column
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) xx[j] = gather(x, cc[j]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = kk + j * sw;
                    vv[j] = e < hi ? s.val[e] : T(0);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j & 1) a1 += vv[j] * xx[j];
                    else a0 += vv[j] * xx[j];
                }
            }
            const T acc = strided_sum(a0 + a1, rw);
            if (slice == 0 && row < nrows) y[r0 + row] = acc;
        }
    }
}

// ------------------------------------------------------------ stream, row walk
// Same blocks, different second half, and the default for fp64.
//
// Why: on gfx950 an 8-byte gather costs per DISTINCT CACHE LINE of the wave
// instruction, not per lane (tools/ubench_gather.hip: 64 lanes on 64 lines that
// hit L2 = 145 cycles, on 28 clusters = 60, 8 lanes per line = 36, contiguous =
// 17), and the coalesced (col, val) stream itself already takes ~0.4 cycles
// per nonzero of the same address/L1 pipeline.  At 70 % of the HBM roofline a CU
// has ~1.4 cycles per nonzero in total, so gathers must average well under one
// cycle per lane.  With one lane per ENTRY (csr_stream) the 64 lanes of a
// gather hold different offsets of the same few rows: far-apart addresses,
// ~1 cycle per lane, and the vector-memory pipeline (TA_BUSY 82-92 %), not HBM,
// sets the pace.  Here the raw (col, val) pairs are staged in LDS and lane i
// walks ROW i (S lanes interleaved per row when a block has fewer than 128
// rows): neighbouring lanes are neighbouring rows at the same position, whose
// columns are neighbours for stencil / FEM / banded matrices, so a gather
// touches a handful of lines.  For unstructured matrices it costs what
// csr_stream's gather costs.
//
// Structure: persistent, grid-stride over the blocks (DRAM page locality: all
// resident workgroups sit on one moving window of the matrix).  Per block:
// registers -> LDS, barrier, issue the NEXT block's stream into the freed
// registers, walk the rows (gathers in batches of 8/4/2/1 so several are in
// flight per lane), merge the S partial sums of each row through LDS, store y,
// barrier.  The HBM stream of block b+G is in flight during the walk of block b.
//
// LDS index k -> k + (k >> 5): one pad slot per 32 entries, so that rows whose
// length shares a factor with the bank count (28-entry rows: 4-way conflicts
// unpadded) spread over the banks.
__device__ __forceinline__ int lds_pad(int k) { return k + (k >> 5); }

template <typename T, bool NT, int CAP, bool PERSIST>
__global__ __launch_bounds__(kBlock) void csr_stream_walk(int num_blocks, int xcd_chunk,
                                                          const int4 *__restrict__ desc,
                                                          const int *__restrict__ row_ptr,
                                                          const int *__restrict__ col,
                                                          const T *__restrict__ val,
                                                          const T *__restrict__ x,
                                                          T *__restrict__ y) {
    using V2 = typename vec2<T>::type;
    constexpr int kUnits = CAP / kStreamUnit;
    constexpr int kLds = CAP + CAP / 32 + 2;
    __shared__ T lv[kLds];
    __shared__ int lc[kLds];
    __shared__ T part[kBlock];
    const int t = threadIdx.x;
    const int G = gridDim.x;
    int b = blockIdx.x;
    const int total = xcd_chunk > 0 ? (num_blocks + 8 * xcd_chunk - 1) / (8 * xcd_chunk) * (8 * xcd_chunk) : num_blocks;
    if (b >= total) return;
    auto block_of = [&](int linear) {
        const int real = xcd_chunked(linear, xcd_chunk);
        return real < num_blocks ? desc[real] : int4{0, 0, 0, 0};
    };

    v2i c[kUnits];
    V2 v[kUnits];
    auto issue = [&](int first_entry) {
        const int e_first = (first_entry & kBaseMask) + 2 * t;
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            c[u] = stream_load<NT>(reinterpret_cast<const v2i *>(col + e_first + u * kStreamUnit));
            v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(val + e_first + u * kStreamUnit));
        }
    };
    // N gathers in flight, then N fused multiply-adds
    auto batch = [&](auto n_tag, int k, int step, T &a0, T &a1) {
        constexpr int N = decltype(n_tag)::value;
        int cc[N];
        T xx[N], vv[N];
#pragma unroll
        for (int j = 0; j < N; ++j) cc[j] = lc[lds_pad(k + j * step)];
#pragma unroll
        for (int j = 0; j < N; ++j) xx[j] = gather(x, cc[j]);
#pragma unroll
        for (int j = 0; j < N; ++j) vv[j] = lv[lds_pad(k + j * step)];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j & 1) a1 += vv[j] * xx[j];
            else a0 += vv[j] * xx[j];
        }
    };
    auto walk = [&](int lo, int hi, int first, int step) -> T {
        T a0 = 0, a1 = 0;
        int k = lo + first;
        int rem = k < hi ? (hi - k + step - 1) / step : 0;  // entries this lane owns
        while (rem >= 8) {
            batch(std::integral_constant<int, 8>{}, k, step, a0, a1);
            k += 8 * step;
            rem -= 8;
        }
        if (rem & 4) {
            batch(std::integral_constant<int, 4>{}, k, step, a0, a1);
            k += 4 * step;
        }
        if (rem & 2) {
            batch(std::integral_constant<int, 2>{}, k, step, a0, a1);
            k += 2 * step;
        }
        if (rem & 1) batch(std::integral_constant<int, 1>{}, k, step, a0, a1);
        return a0 + a1;
    };
    // one block: its stream is in c/v on entry; `prefetch` refills c/v for a later block
    auto step_block = [&](const int4 d, auto prefetch) {
        const int r0 = d.x, nrows = d.z;
        const int base = d.y & kBaseMask;
        // lane -> (row, slice), rows fastest so that neighbouring lanes are neighbouring rows
        const int nr = nrows > 0 ? nrows : 1;  // a dummy block past the end has no rows
        const int slices = nrows < kBlock ? kBlock / nr : 1;
        const int slice = nrows < kBlock ? t / nr : 0;
        int row = nrows < kBlock ? t - slice * nrows : t;
        const bool live = slice < slices;
        int lo = 0, hi = 0;
        if (live && row < nrows) {
            lo = row_ptr[r0 + row];
            hi = row_ptr[r0 + row + 1];
        }
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            const int k = lds_pad(u * kStreamUnit + 2 * t);  // a pair never straddles a pad slot
            lc[k] = c[u].x;
            lc[k + 1] = c[u].y;
            lv[k] = v[u].x;
            lv[k + 1] = v[u].y;
        }
        __syncthreads();
        prefetch();
        lo -= base;
        hi -= base;
        for (int first = 0; first < nrows; first += kBlock) {  // one trip unless nrows > 256
            if (first > 0) {
                row = first + t;
                lo = hi = 0;
                if (row < nrows) {
                    lo = row_ptr[r0 + row] - base;
                    hi = row_ptr[r0 + row + 1] - base;
                }
            }
            const T acc = walk(lo, hi, slice, slices);
            if (slices == 1) {
                if (live && row < nrows) y[r0 + row] = acc;  // lanes past the last row own nothing
            } else {
                part[t] = acc;
                __syncthreads();
                if (t < nrows) {
                    T s = part[t];
                    for (int q = 1; q < slices; ++q) s += part[q * nrows + t];
                    y[r0 + t] = s;
                }
            }
        }
        __syncthreads();  // lc / lv / part are rewritten by the next block
    };

    if constexpr (!PERSIST) {
        // one block per workgroup: overlap comes from the other workgroups on the CU.
        // (Loads return in order per wave, so gathers issued behind a prefetched HBM
        // stream would wait for it; the persistent form only pays off when the walk
        // is short.)
        const int4 d = block_of(b);
        issue(d.y);
        step_block(d, [] {});
        return;
    }
    int left = (total - 1 - b) / G + 1;
    int4 d = block_of(b);
    issue(d.y);
    while (left >= 2) {  // prefetching steps are unconditional (see csr_stream_pipe)
        const int4 dn = block_of(b + G);
        step_block(d, [&] { issue(dn.y); });
        d = dn;
        b += G;
        --left;
    }
    step_block(d, [] {});
}

}  // namespace spmv
