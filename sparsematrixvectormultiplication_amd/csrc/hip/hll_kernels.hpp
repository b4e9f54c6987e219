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
template <typename T, bool NT, int MAXU>
__device__ __forceinline__ void stage_products(T *prod, const int *__restrict__ JA,
                                               const T *__restrict__ AS,
                                               const T *__restrict__ x, long long from,
                                               int count) {
    // all units of the range in ONE batch: every load of the block is in flight together
    // instead of one HBM latency per batch.  MAXU (2, 4, 6 or 8 units of 512 slots) is
    // picked at upload from the largest workgroup of the matrix, so small hacks do not pay
    // the registers of an 8-unit batch.
    const int units = (count + kStreamUnit - 1) / kStreamUnit;
    if (units <= MAXU) {
        switch (units) {
            case 1: stage_units<T, NT, 1>(prod, JA, AS, x, from, 0, count); break;
            case 2: stage_units<T, NT, 2>(prod, JA, AS, x, from, 0, count); break;
            case 3: if constexpr (MAXU >= 3) stage_units<T, NT, 3>(prod, JA, AS, x, from, 0, count); break;
            case 4: if constexpr (MAXU >= 4) stage_units<T, NT, 4>(prod, JA, AS, x, from, 0, count); break;
            case 5: if constexpr (MAXU >= 5) stage_units<T, NT, 5>(prod, JA, AS, x, from, 0, count); break;
            case 6: if constexpr (MAXU >= 6) stage_units<T, NT, 6>(prod, JA, AS, x, from, 0, count); break;
            case 7: if constexpr (MAXU >= 7) stage_units<T, NT, 7>(prod, JA, AS, x, from, 0, count); break;
            case 8: if constexpr (MAXU >= 8) stage_units<T, NT, 8>(prod, JA, AS, x, from, 0, count); break;
            default: break;
        }
        return;
    }
    // more than MAXU units (only the row-chunk path of an oversized hack gets here)
    int k0 = 0, left = units;
    while (left >= MAXU) {
        stage_units<T, NT, MAXU>(prod, JA, AS, x, from, k0, count);
        k0 += MAXU * kStreamUnit;
        left -= MAXU;
    }
    for (; left > 0; --left, k0 += kStreamUnit) stage_units<T, NT, 1>(prod, JA, AS, x, from, k0, count);
}

// Workgroup b owns the consecutive rows [desc[b].x, desc[b].x + desc[b].y) of the matrix,
// i.e. one contiguous window of the flat slab starting at slot (desc[b].w : desc[b].z)
// -- rows of neighbouring hacks, each with its own hack's maxnz.  Host packing
// (hll_build_blocks) fills the window up to the LDS stage, so workgroups are as full as
// the CSR kernel's whatever maxnz is; a row that alone exceeds the stage gets a workgroup
// of its own and is accumulated in registers.
//
// LDS is dynamic: `stage_slots` (+2) products, sized at upload to the largest workgroup
// (at most kHllCap), so matrices with short rows keep more workgroups resident.
template <typename T, bool NT, int MAXU>
__global__ __launch_bounds__(kBlock) void hll_lds(int stage_slots, const int4 *__restrict__ desc,
                                                  const long long *__restrict__ hack_off,
                                                  const int *__restrict__ maxnz,
                                                  const int *__restrict__ JA,
                                                  const T *__restrict__ AS,
                                                  const T *__restrict__ x, T *__restrict__ y) {
    extern __shared__ __align__(16) unsigned char hll_dyn_lds[];
    T *wave_part = reinterpret_cast<T *>(hll_dyn_lds);                   // [kBlock / 64]
    T *prod = reinterpret_cast<T *>(hll_dyn_lds) + 16 / sizeof(T) * 2;  // [stage_slots + 2]
    const int t = threadIdx.x;
    const int4 d = desc[blockIdx.x];
    const int row_first = d.x, nrows = d.y & 0xffff;
    const int window_slots = (int)((unsigned)d.y >> 16);  // slots from the even base to the window's end (0: one very long row)
    const long long first_slot = ((long long)d.w << 32) | (unsigned)d.z;
    const long long base = first_slot & ~1LL;  // 8/16-byte aligned stage loads

    // slot range of a row relative to base
    auto row_range = [&](int q, int &lo, int &m) {
        const int r = row_first + q;
        const int h = r / kHack;
        m = maxnz[h];
        lo = (int)(hack_off[h] + (long long)(r % kHack) * m - base);
    };

    if (nrows == 1 && (window_slots == 0 || window_slots > stage_slots)) {
        int lo1, m1;
        row_range(0, lo1, m1);
        if (lo1 + m1 > stage_slots) {
            // a single row longer than the stage: whole workgroup, registers only.  Trips of 8 units of pair loads
            // (16 loads per lane in flight, as csr_long_pieces does): a circuit-like slab -- a few rows that touch
            // everything, hence hacks of 32 rows x tens of thousands of slots -- spent its time in this loop when it
            // took one dependent load at a time (dc1-size stand-in: 60 us against 12 us for the same matrix as CSR)
            using V2 = typename vec2<T>::type;
            constexpr int kUnits = 8;
            const long long n0 = base + lo1, n1 = n0 + m1;
            T a0 = 0, a1 = 0;
            long long e0 = (n0 & ~1LL) + 2 * t;
            for (; e0 - 2 * t + kUnits * kStreamUnit <= n1; e0 += kUnits * kStreamUnit) {
                v2i c[kUnits];
                V2 v[kUnits];
#pragma unroll
                for (int u = 0; u < kUnits; ++u) {
                    c[u] = stream_load<NT>(reinterpret_cast<const v2i *>(JA + e0 + u * kStreamUnit));
                    v[u] = stream_load<NT>(reinterpret_cast<const V2 *>(AS + e0 + u * kStreamUnit));
                }
                T xv[2 * kUnits];
#pragma unroll
                for (int u = 0; u < kUnits; ++u) {
                    xv[2 * u] = gather(x, c[u].x);
                    xv[2 * u + 1] = gather(x, c[u].y);
                }
                if (e0 < n0) v[0].x = T(0);  // only lane 0 of the first trip of a row that starts on an odd slot
#pragma unroll
                for (int u = 0; u < kUnits; ++u) {
                    a0 += v[u].x * xv[2 * u];
                    a1 += v[u].y * xv[2 * u + 1];
                }
            }
            for (; e0 < n1; e0 += kStreamUnit) {  // remainder: bounded per slot
                const v2i c = stream_load<NT>(reinterpret_cast<const v2i *>(JA + e0));
                const V2 v = stream_load<NT>(reinterpret_cast<const V2 *>(AS + e0));
                if (e0 >= n0) a0 += v.x * gather(x, c.x);
                if (e0 + 1 < n1) a1 += v.y * gather(x, c.y);
            }
            T acc = group_sum<64>(a0 + a1);
            if ((t & 63) == 0) wave_part[t >> 6] = acc;
            __syncthreads();
            if (t == 0) {
                T s = wave_part[0];
                for (int w = 1; w < kBlock / 64; ++w) s += wave_part[w];
                y[row_first] = s;
            }
            return;
        }
    }

    // lane -> row of the first pass and its slot range, looked up BEFORE the stream so
    // the two small global loads ride along with it
    int lanes = 1;
    if (nrows <= kBlock / 2) {
        lanes = 1 << (31 - __clz(kBlock / (nrows > 0 ? nrows : 1)));
        if (lanes > 64) lanes = 64;
    }
    const int rows_per_pass = kBlock / lanes;
    const int my_row = t / lanes, my_lane = t % lanes;
    // the first pass's row extent: the RAW table values are loaded here, next to the stream, and turned into a slot range
    // behind the stage (computing it on the spot makes the compiler wait for the tables before the stream has gone out:
    // see hll_lds_local); how much to stage comes with the descriptor
    int m0 = 0;
    long long ho0 = 0;
    if (my_row < nrows) {
        const int h = (row_first + my_row) / kHack;
        m0 = maxnz[h];
        ho0 = hack_off[h];
    }
    stage_products<T, NT, MAXU>(prod, JA, AS, x, base, window_slots);
    __syncthreads();
    int lo = my_row < nrows ? (int)(ho0 + (long long)((row_first + my_row) % kHack) * m0 - base) : 0, m_row = m0;
    for (int first = 0; first < nrows; first += rows_per_pass) {
        const int q = first + my_row;
        if (first > 0) {
            lo = m_row = 0;
            if (q < nrows) row_range(q, lo, m_row);
        }
        T acc = lds_strided_sum(prod, lo, lo + m_row, my_lane, lanes);
        acc = group_sum_rt(acc, lanes);
        if (my_lane == 0 && q < nrows) y[row_first + q] = acc;
    }
}

// hll_lds with the x window in LDS: same windows idea, but the window's x lines are staged in
// LDS by full-width loads and JA is replaced by 16-bit slots into that stage -- the HLL twin of
// csr_stream_local (see csr_kernels.hpp for the reasoning and the line-list format).  Windows
// are cut at upload with both limits (kLocalCap slots, kLocalLinesMax lines); padding slots
// repeat the row's last column, so they add no lines.
// PAT (round 3): a pattern plan (csr_kernels.hpp, "slot PATTERNS") -- lja is not read; the slots of a window are rebuilt
// in LDS from its pattern table: in a slab every row of a hack has the hack's length, padding included (a padding slot
// repeats the row's last column: part of the pattern), so every row but a window's first can be its predecessor shifted.
template <typename T, bool NT, int CAP, bool PAT = false>
__global__ __launch_bounds__(kBlock) void hll_lds_local(int num_blocks, int xcd_chunk,
                                                        const int4 *__restrict__ desc,
                                                        const int4 *__restrict__ ldesc,
                                                        const int *__restrict__ lines,
                                                        const unsigned *__restrict__ row_seg,
                                                        const unsigned short *__restrict__ lja,
                                                        const T *__restrict__ AS,
                                                        const T *__restrict__ x, T *__restrict__ y,
                                                        const int2 *__restrict__ pdesc = nullptr,
                                                        const unsigned *__restrict__ rinfo = nullptr,
                                                        const unsigned short *__restrict__ ptab = nullptr, int stage_bytes = 0) {
        using V2 = typename vec2<T>::type;
    constexpr int kUnit = 2 * kBlock, kUnits = CAP / kUnit;
    extern __shared__ __align__(16) unsigned char hll_local_lds[];
    T *stage = reinterpret_cast<T *>(hll_local_lds);  // x lines first, then the products

    const int b = xcd_chunked(blockIdx.x, xcd_chunk);
    if (b >= num_blocks) return;
    const int t = threadIdx.x;
    const int4 d = desc[b];
    const int4 ld = ldesc[b];  // {first line, lines, slots of the window counted from its even base, -}
    const int row_first = d.x, nrows = d.y;
    const long long first_slot = ((long long)d.w << 32) | (unsigned)d.z;
    const long long base = first_slot & ~1LL;

    // a row's slots inside its window, precomputed at upload (hll_row_segments): one 4-byte load per lane
    // where the hack tables cost two loads and 64-bit arithmetic (nlpkkt-like: 208 -> see profiles/r2_hll_*)
    auto row_range = [&](int q, int &lo, int &m) {
        const unsigned seg = row_seg[row_first + q];
        lo = (int)(seg & 0xffffu);
        m = (int)(seg >> 16);
    };
    const int lanes = lanes_for_rows<kBlock>(nrows);
    const int rows_per_pass = kBlock / lanes;
    const int my_row = t / lanes, my_lane = t % lanes;
    // The first pass's row extent rides along with the stream: the RAW word is loaded here and decoded behind the stage.
    // (Decoding it on the spot -- as this kernel did until round 3 -- makes the compiler wait for it, vmcnt(0), BEFORE
    // the block's stream loads have gone out: one exposed memory latency per block, and the reason hll_lds_local ran
    // 7 % behind csr_stream_local on identical data: profiles/r3_hll_as_csr_*.txt, r3_isa_mix_*.txt.)
    unsigned seg0 = 0;
    if (my_row < nrows) seg0 = row_seg[row_first + my_row];
    // the slot count comes with the descriptor, so the stream does not wait for the hack table
    const int count = ld.z;
    const int units = (count + kUnit - 1) / kUnit;                            // wave-uniform
    const int rounds = (ld.y + kLocalLineQuantum - 1) / kLocalLineQuantum;    // wave-uniform
    const int *my_lines = lines + ld.x;
    const unsigned short *wj = lja + base;
    const T *wa = AS + base;
    if constexpr (PAT) {
        const int2 pd = pdesc[b];
        pat_ctx pc;
        pc.slots = reinterpret_cast<unsigned short *>(hll_local_lds + stage_bytes);
        pc.ptab8 = reinterpret_cast<const uint4 *>(ptab + pd.x);
        pc.rinfo = rinfo;
        pc.row_ptr = nullptr;
        pc.row_seg = row_seg;
        pc.r0 = row_first;
        pc.nrows = nrows;
        pc.base = 0;
        pc.first = (int)(first_slot - base);
        pc.end = count;
        pc.lanes = lanes;
        pc.seg_lo = (int)seg0;  // (raw: decoded behind the stream)
        pc.seg_hi = 0;
        pc.ri = my_row < nrows ? rinfo[row_first + my_row] : 0u;
        if (units == kUnits) {
            switch (rounds) {
                case 1: local_stage_full_pat<T, NT, CAP, 1, true>(stage, my_lines, ld.y - 1, wa, x, 2 * t, pc); break;
                case 2: local_stage_full_pat<T, NT, CAP, 2, true>(stage, my_lines, ld.y - 1, wa, x, 2 * t, pc); break;
                case 3: local_stage_full_pat<T, NT, CAP, 3, true>(stage, my_lines, ld.y - 1, wa, x, 2 * t, pc); break;
                case 4: local_stage_full_pat<T, NT, CAP, 4, true>(stage, my_lines, ld.y - 1, wa, x, 2 * t, pc); break;
                case 5: local_stage_full_pat<T, NT, CAP, 5, true>(stage, my_lines, ld.y - 1, wa, x, 2 * t, pc); break;
                case 6: local_stage_full_pat<T, NT, CAP, 6, true>(stage, my_lines, ld.y - 1, wa, x, 2 * t, pc); break;
                case 7: local_stage_full_pat<T, NT, CAP, 7, true>(stage, my_lines, ld.y - 1, wa, x, 2 * t, pc); break;
                default: local_stage_full_pat<T, NT, CAP, 8, true>(stage, my_lines, ld.y - 1, wa, x, 2 * t, pc); break;
            }
        } else {
            for (int k = 0; k < rounds; ++k) {
                const int line = my_lines[min(k * kLocalLineQuantum + (t >> 3), ld.y - 1)];
                const unsigned off = (unsigned)line * (unsigned)kLineBytes + (unsigned)(t & 7) * 16u;
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(stage) + (k * kBlock + t) * 16) =
                    *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(x) + off);
            }
            uint4 pg[2];
            pat_load_first<true>(pc, pg);
            pat_expand_slots<kBlock, CAP, true>(pc, pg);
            __syncthreads();
            V2 p[kUnits];
#pragma unroll
            for (int u = 0; u < kUnits; ++u) {
                if (u < units) {
                    const unsigned c = *reinterpret_cast<const unsigned *>(pc.slots + u * kUnit + 2 * t);
                    p[u] = stream_load<NT>(reinterpret_cast<const V2 *>(wa + 2 * t + u * kUnit));
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
            case 1: local_stage_full<T, NT, CAP, 1>(stage, my_lines, ld.y - 1, wj, wa, x, 2 * t); break;
            case 2: local_stage_full<T, NT, CAP, 2>(stage, my_lines, ld.y - 1, wj, wa, x, 2 * t); break;
            case 3: local_stage_full<T, NT, CAP, 3>(stage, my_lines, ld.y - 1, wj, wa, x, 2 * t); break;
            case 4: local_stage_full<T, NT, CAP, 4>(stage, my_lines, ld.y - 1, wj, wa, x, 2 * t); break;
            case 5: local_stage_full<T, NT, CAP, 5>(stage, my_lines, ld.y - 1, wj, wa, x, 2 * t); break;
            case 6: local_stage_full<T, NT, CAP, 6>(stage, my_lines, ld.y - 1, wj, wa, x, 2 * t); break;
            case 7: local_stage_full<T, NT, CAP, 7>(stage, my_lines, ld.y - 1, wj, wa, x, 2 * t); break;
            default: local_stage_full<T, NT, CAP, 8>(stage, my_lines, ld.y - 1, wj, wa, x, 2 * t); break;
        }
    } else {
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
                const unsigned c = stream_load<NT>(reinterpret_cast<const unsigned *>(wj + 2 * t + u * kUnit));
                p[u] = stream_load<NT>(reinterpret_cast<const V2 *>(wa + 2 * t + u * kUnit));
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
    int lo = (int)(seg0 & 0xffffu), m_row = (int)(seg0 >> 16);
    for (int first = 0; first < nrows; first += rows_per_pass) {
        const int q = first + my_row;
        if (first > 0) {
            lo = m_row = 0;
            if (q < nrows) row_range(q, lo, m_row);
        }
        T acc = lds_strided_sum(stage, lo, lo + m_row, my_lane, lanes);
        acc = group_sum_rt(acc, lanes);
        if (my_lane == 0 && q < nrows) y[row_first + q] = acc;
    }
}

// ------------------------------------------------------ CSR -> HLL on the device
// the reference builds HLL on the host with one qsort and two mallocs
// per row / hack (src/hll_matrix.c:37-257) and uploads hack by hack.  With the CSR matrix
// already resident (columns ascending inside each row, as convert_in_csr leaves them) the
// flat slab is two trivial kernels: the per-hack maximum row length, then a fill that
// copies each row and pads it exactly as the host builder does (value 0, column = the
// row's last real column, 0 for an empty row; src/hll_matrix.c:129-140,241-246).

// one wavefront per row: lanes stride over the row's maxnz slots
template <typename T>
__global__ __launch_bounds__(kBlock) void hll_fill_from_csr(int M, const int *__restrict__ row_ptr,
                                                            const int *__restrict__ col,
                                                            const T *__restrict__ val,
                                                            const long long *__restrict__ hack_off,
                                                            const int *__restrict__ maxnz,
                                                            int *__restrict__ JA, T *__restrict__ AS) {
    const int r = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
    if (r >= M) return;
    const int lane = threadIdx.x & 63;
    const int h = r / kHack;
    const int m = maxnz[h];
    const int begin = row_ptr[r], len = row_ptr[r + 1] - begin;
    const long long at = hack_off[h] + (long long)(r % kHack) * m;
    const int pad_col = len > 0 ? col[begin + len - 1] : 0;
    for (int j = lane; j < m; j += 64) {
        const bool real = j < len;
        JA[at + j] = real ? col[begin + j] : pad_col;
        AS[at + j] = real ? val[begin + j] : T(0);
    }
}

}  // namespace spmv
