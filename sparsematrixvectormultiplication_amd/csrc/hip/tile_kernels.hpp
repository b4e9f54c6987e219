// tile_kernels.hpp -- csr_tile: CSR SpMV for matrices whose columns are too scattered for the
// x-window plan of csr_stream_local (road-like graphs, wide random bands, uniformly random and
// power-law columns: the graph / circuit / economics matrices of the reference's own list,
// /root/reference/result/result_cuda.csv:2-31).
//
// What limits the gather kernels there ("gather wall"): a 64-lane gather costs the
// texture addresser per DISTINCT LINE (~145 cycles for 64 lines that hit L2), and once x outgrows an
// XCD's 4 MiB L2 every gathered value drags a whole line across the fabric (~660 cycles).  csr_tile
// restructures the product in two dimensions instead of one:
//
//   * rows are cut into ROW BLOCKS of consecutive rows (at most `rows_per_block`, about equally many
//     entries each, as many blocks as fill whole rounds of the workgroups the chip holds at once); a workgroup
//     keeps a block's y values as accumulators in LDS while it walks the block's passes, and walks several
//     blocks back to back (a STREAM: the next block's first loads are in flight while this one is summed);
//   * the block's entries are re-ordered at upload into PASSES = consecutive column ranges of at most 2048
//     entries, and inside a pass by (row, column).  All workgroups start at column 0 and sweep upwards
//     together -- equal work per block keeps them in step -- so at any moment the chip gathers from a
//     narrow band of x that stays in L2 (the L2-sized column stripes of a DCSR scheme, without per-stripe
//     row lists or partial sums in memory: the accumulators never leave LDS).  Measured on the power-law
//     matrix: 2.9 cycles per gathered value and CU when the blocks are in step and a pass spans <= 2 MB of
//     x, against 7-8 when they are not;
//   * a pass whose column range is narrow and dense enough is STAGED: its slice of x is copied into
//     LDS with full-width coalesced loads and the per-entry lookups become ds_reads (the x-window
//     idea with a dense window instead of a line list);
//     a banded matrix gets a PACKED plan: every pass is cut at the slice's width and staged (entries of
//     almost empty windows go to a remainder that tile_remainder adds behind the tiles), and the kernel
//     instantiation for it has no gather code and sends out the same loads in every pass;
//   * inside a pass an entry carries a key {head flag, local row} -- 16 bits beside a 32-bit column, or,
//     in a packed plan, packed with the column's offset in the slice into one 32-bit word --
//     and every lane holds FOUR CONSECUTIVE entries, so most of a row's run is added up in registers:
//     runs that begin and end inside a lane go straight to the accumulator; a run that crosses lanes is
//     finished by a right-to-left segmented scan over the lanes' leading partial sums (DPP row shifts and
//     v_readlane inside a wavefront, skipped where every lane holds a head; one LDS slot per wavefront
//     across them) and added by the lane that holds its head.  One owner per (row, pass), passes in order,
//     a fixed reduction tree: no atomics, the same bits on every launch;
//   * rows longer than the plan's limit (1024 entries) are compacted into row blocks of their own, whose
//     column ranges hold so many entries that every pass is staged; such a block's passes are dealt out to
//     many workgroups (work items), each leaves its accumulators in a slab, tile_slab_finish adds them.
//
// No reference counterpart (its CUDA kernels gather x per entry, cuda_src/csr_matrix_cuda.cu:122-241).
#pragma once
#include <hip/hip_runtime.h>

#include "csr_kernels.hpp"

namespace spmv {

typedef unsigned v4u __attribute__((ext_vector_type(4)));  // native vectors: stay in registers where uint4 (a struct) may not
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr int kTileBlock = 512;      // threads per workgroup (8 wavefronts)
constexpr int kTileWaves = kTileBlock / 64;
constexpr int kTileChunkMax = 4096;  // padding behind the entry arrays (the plan's chunk, entries per pass at most, is 2048)
constexpr int kTileRowsMax = 32768;  // rows per block at most (local row fits the key's 15 bits; fp64 blocks stop at what the LDS takes)
constexpr int kTileHead = 0x8000;    // key bit: first entry of its row in this pass
constexpr int kTileRowMask = 0x7fff;
// The passes of a PACKED plan (pass_desc.w bit 30; all of them or none) store their entries packed: the column word holds
// head << 31 | local row << 14 | (column - first staged column) and there is no key array --
// 4 + sizeof(T) bytes per entry, CSR's own, instead of 6 + sizeof(T).
constexpr int kTilePassPacked = 1 << 30;  // pass_desc.w bit: the pass's entries are packed
constexpr int kTilePassLast = 1 << 29;    // pass_desc.w bit (stream order only): the last pass of its block
constexpr int kTileWlenMask = (1 << 20) - 1;  // pass_desc.w: the staged columns
constexpr int kTilePackShift = 14;
constexpr unsigned kTilePackColMask = (1u << kTilePackShift) - 1;
constexpr int kTileTripBytes = kTileBlock * 16;  // x bytes one staging trip of the workgroup copies
constexpr int kTileTrips = 4;        // trips per pass: windows of up to 32 KiB
constexpr int kTileAhead = 3;        // passes whose entries are in flight beyond the one being worked on
// (a compact variant -- 24 KiB windows, 1 pass ahead, 79 VGPRs, three workgroups per CU -- was measured and is no
// faster: road-like 191 us at 3072 rows against 173 us for this one at 4096, profiles/r2_ab_csr_tile.txt)
constexpr int kTileSlotBytes = 256;  // LDS in front of the accumulators: one (sum, closed) slot per wavefront and quad
constexpr int kTileLdsBytes = 160 * 1024;  // a CU's LDS
// the tallest row blocks: two workgroups per CU (banded plans) / one (scattered), each with its wave slots, its
// accumulators and a full 32 KiB x slice
template <typename T>
constexpr int tile_banded_rows_max() {
    return (kTileLdsBytes / 2 - kTileSlotBytes - kTileTrips * kTileTripBytes) / (int)sizeof(T) / 256 * 256;
}
template <typename T>
constexpr int tile_scattered_rows_max() {
    constexpr int fit = (kTileLdsBytes - kTileSlotBytes - kTileTrips * kTileTripBytes) / (int)sizeof(T) / 256 * 256;
    return fit < kTileRowsMax ? fit : kTileRowsMax;
}

template <typename T> struct vec4v;  // four values of a lane's quad
template <> struct vec4v<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct vec4v<double> { typedef double type __attribute__((ext_vector_type(4))); };

// What a lane holds of one pass while earlier passes are still being worked on: QUADS groups of four
// consecutive (column, key, value) entries (and, separately, TRIPS 16-byte pieces of a pass's x slice).  Kept
// as separate arrays of native vectors (not a struct) so that every element lives in its own register
// whatever the optimiser makes of the rotation of the sets below.
#define SPMV_TILE_ENTRY_REGS(name) \
    v4i name##_c[kQuads];          \
    v2u name##_k[kQuads];          \
    V4 name##_v[kQuads]
#define SPMV_TILE_ENTRY_ARGS(name) name##_c, name##_k, name##_v

// Every load of a pass goes out unconditionally -- a full chunk of entries and TRIPS pieces of x, whatever
// the pass really holds (lanes behind the pass's end re-read its last quad, lanes behind a window's end its
// last 16 bytes: one line for all of them, no extra traffic) -- so that the NUMBER of loads in flight is a
// compile-time constant: vmcnt retires in order, and only with a known count can the wait for the current
// pass's gathers leave the next pass's loads in flight.
template <typename T, bool NT, int CH, bool PACK>
__device__ __forceinline__ void tile_issue_entries(v4i (&rc)[CH / (4 * kTileBlock)], v2u (&rk)[CH / (4 * kTileBlock)],
                                                   typename vec4v<T>::type (&rv)[CH / (4 * kTileBlock)], const int4 d,
                                                   const int *__restrict__ tcol, const unsigned short *__restrict__ tkey,
                                                   const T *__restrict__ tval) {
    using V4 = typename vec4v<T>::type;
    constexpr int kQuads = CH / (4 * kTileBlock);
    const int t = threadIdx.x;
    const int e_last = d.x + ((max(d.y, 1) - 1) & ~3);
#pragma unroll
    for (int u = 0; u < kQuads; ++u) {
        const int e = min(d.x + 4 * t + u * 4 * kTileBlock, e_last);
        rc[u] = stream_load<NT>(reinterpret_cast<const v4i *>(tcol + e));
        // (a packed plan's keys ride in its column words: the key array is not read at all)
        if constexpr (!PACK) rk[u] = stream_load<NT>(reinterpret_cast<const v2u *>(tkey + e));
        rv[u] = stream_load<NT>(reinterpret_cast<const V4 *>(tval + e));
    }
}

// the pass's slice of x (a gather pass: its first 16 bytes, again and again): 16 bytes per lane and trip
template <typename T, int TRIPS>
__device__ __forceinline__ void tile_issue_window(v4u (&rw)[TRIPS], const int4 d, int stage_ok, const T *__restrict__ x) {
    constexpr int kPer = 16 / (int)sizeof(T);
    const int t = threadIdx.x;
    const int wlen = stage_ok ? max(d.w & kTileWlenMask, kPer) : kPer;
    const char *src = reinterpret_cast<const char *>(x + d.z);
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
        const int j = min((k * kTileBlock + t) * kPer, wlen - kPer);
        rw[k] = *reinterpret_cast<const v4u *>(src + (size_t)j * sizeof(T));
    }
}

// value of lane + N inside the lane's row of 16 (DPP row_shl: a VALU move, no LDS crossbar); 0 past the row's end
template <int CTRL>
__device__ __forceinline__ double row_down(double v) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
template <int CTRL>
__device__ __forceinline__ float row_down(float v) {
    return __int_as_float(dpp_i32<CTRL>(__float_as_int(v)));
}
__device__ __forceinline__ double read_lane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float read_lane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Right-to-left segmented scan over the 64 lanes: on entry r = the lane's leading partial sum, h = 1 when the lane
// holds a head; on exit r(s) = lead(s) + lead(s + 1) + ... up to and including the first lane >= s that holds a head
// (or the wavefront's end), h(s) = whether there was one.  Rows of 16 lanes by DPP row shifts, the four rows stitched
// from the last to the first with v_readlane: ~100 cycles of dependent VALU / SALU work where six ds_bpermute steps
// (three crossbar trips each) were ~600.
template <typename T>
__device__ __forceinline__ void chain_scan(T &r, int &h, int lane) {
    const int lir = lane & 15;
#define SPMV_CHAIN_STEP(N, CTRL)              \
    {                                         \
        const T rn = row_down<CTRL>(r);       \
        const int hn = dpp_i32<CTRL>(h);      \
        if (lir + N < 16 && !h) {             \
            r += rn;                          \
            h = hn;                           \
        }                                     \
    }
    SPMV_CHAIN_STEP(1, 0x101)  // row_shl:1
    SPMV_CHAIN_STEP(2, 0x102)
    SPMV_CHAIN_STEP(4, 0x104)
    SPMV_CHAIN_STEP(8, 0x108)
#undef SPMV_CHAIN_STEP
#pragma unroll
    for (int q = 2; q >= 0; --q) {  // what a chain that leaves row q collects from row q + 1 on
        const T rn = read_lane(r, 16 * (q + 1));
        const int hn = __builtin_amdgcn_readlane(h, 16 * (q + 1));
        if ((lane >> 4) == q && !h) {
            r += rn;
            h = hn;
        }
    }
}

// One pass.  (cc, ck, cv) and cw hold its entries and its x slice; the loads this pass sends out, in this order:
// its own gathers, the x slice of the pass AFTER THE NEXT (back into cw, whose contents have just gone to LDS; the
// next pass's slice is in flight in the other register set), the entries of the pass kTileAhead passes further on
// (fc, fk, fv).  vmcnt retires in order: the wait for a slice at the top of a pass drains everything sent out
// before that slice, so what stays in flight across the wait is what was sent out after it -- with the slice one
// pass ahead that was ONE pass's entries per workgroup (24 KB; 4.1-4.4 TB/s for the bare stream of entries), with
// it two passes ahead it is two.  Past the block's last pass the loads repeat that pass.
// GA (gather ahead, round 3; plans with gather passes only; OFF by default: it did not pay, see the end of this comment): the gathers of a pass go out one pass EARLY -- at the top
// of the pass before it, from that pass's column words (nc, in registers since two passes), into the other of two
// register sets (xn) -- and this pass works on the set filled a pass ago (xg).  Without it a pass's gathers are sent
// and awaited inside the pass: 8 wavefronts x 4 values = 32 gather wave-instructions in flight per CU, against the 128
// the gather probe needs for its 265 G values / s; config 5's short rows ran at 128 G values / s.  All of them go
// out unconditionally (a staged pass gathers its window's first column, one line for everybody): the number of loads
// in flight stays a compile-time constant, which is what lets vmcnt leave the later ones in flight.
// Measured (profiles/r3_ab_gather_ahead.txt): the waits come out as designed (vmcnt up to 21 in the fp32 kernel) and the
// time does not move -- config 5 1113 -> 1108 us, uniformly random columns 514.9 -> 514.8: the gathers in flight per
// wavefront are not what bounds these passes.  Kept behind "tile_gather_ahead" for the record.
template <typename T, int Q>
__device__ __forceinline__ void tile_issue_gathers(T (&xn)[4 * Q], const v4i (&nc)[Q], const int4 dn, int stage_ok,
                                                   const T *__restrict__ x) {
    const int t = threadIdx.x;
    const bool gathers = !(stage_ok && (dn.w & kTileWlenMask));
#pragma unroll
    for (int u = 0; u < Q; ++u) {
        const int i = u * 4 * kTileBlock + 4 * t;
#pragma unroll
        for (int q = 0; q < 4; ++q) xn[4 * u + q] = gather(x, gathers && i + q < dn.y ? nc[u][q] : dn.z);
    }
}

template <typename T, bool NT, int CH, int TRIPS, bool PACK, bool GA = false>
__device__ __forceinline__ void tile_pass(v4i (&cc)[CH / (4 * kTileBlock)], v2u (&ck)[CH / (4 * kTileBlock)],
                                          typename vec4v<T>::type (&cv)[CH / (4 * kTileBlock)], v4u (&cw)[TRIPS],
                                          v4i (&fc)[CH / (4 * kTileBlock)], v2u (&fk)[CH / (4 * kTileBlock)],
                                          typename vec4v<T>::type (&fv)[CH / (4 * kTileBlock)],
                                          const int4 d, const int4 dw, const int4 de, bool live, int stage_ok,
                                          int probe, int &bi, const int2 *__restrict__ sblock_rows, int rows_per_block,
                                          T *__restrict__ y, T *acc, T *xs, T *wave_r, int *wave_h,
                                          const int *__restrict__ tcol, const unsigned short *__restrict__ tkey,
                                          const T *__restrict__ tval, const T *__restrict__ x,
                                          const v4i (&nc)[CH / (4 * kTileBlock)], const int4 dn,
                                          T (&xg)[GA ? CH / kTileBlock : 1], T (&xn)[GA ? CH / kTileBlock : 1]) {
    constexpr int kQuads = CH / (4 * kTileBlock);
    constexpr int kPer = 16 / (int)sizeof(T);
    // (the probe bits stay run-time tests on purpose: with them folded to constants this compiler's register
    // allocation tips over the 128-VGPR cap of two workgroups per CU -- 32-96 bytes of scratch, road 167 -> 197 us)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // d / dw / de: the descriptors of this pass, of the pass whose slice goes out now (two on) and of the pass whose
    // entries go out now (three on), fetched by the caller a pass early: three dependent scalar loads at the top of
    // every pass were on each wavefront's critical path.  live = false: a pass behind the block's last one (the loop
    // below always runs four), which repeats it with no entries
    const int count = live ? d.y : 0, wbase = d.z;
    // PACK: a plan all of whose passes are staged and packed (tile_plan.hpp cuts every pass at the window) -- no
    // gather, no key array, and the same loads in every pass whatever it holds
    const int wlen = PACK ? max(d.w & kTileWlenMask, kPer) : stage_ok ? (d.w & kTileWlenMask) : 0;
    if constexpr (PACK) {
        // every piece is stored, needed or not, each to its own place (the launch gives xs room for all TRIPS trips): a
        // store the compiler may skip leaves its load pending on that path, and the next write to those registers
        // then waits for it -- and for everything sent out before it
#pragma unroll
        for (int k = 0; k < TRIPS; ++k) *reinterpret_cast<v4u *>(xs + (k * kTileBlock + t) * kPer) = cw[k];
        // (the reload of cw below stays behind these stores: hoisted above them it lands in a third register set)
        __builtin_amdgcn_sched_barrier(0);
    } else if (wlen) {
#pragma unroll
        for (int k = 0; k < TRIPS; ++k) {
            const int j = min((k * kTileBlock + t) * kPer, wlen - kPer);  // (the same bytes to the same place)
            if (k * kTileBlock * kPer < wlen) *reinterpret_cast<v4u *>(xs + j) = cw[k];
        }
    }
    T xv[(PACK || GA) ? 1 : 4 * kQuads];
    if constexpr (GA) {
        tile_issue_gathers<T, kQuads>(xn, nc, dn, stage_ok, x);  // the NEXT pass's values; this pass's are in xg
    } else if constexpr (!PACK) {
        if (!wlen && !(probe & 2)) {  // gathers go out first (vmcnt retires in order): they are back when the barrier opens
#pragma unroll
            for (int u = 0; u < kQuads; ++u) {
                const int i = u * 4 * kTileBlock + 4 * t;
                // entries behind the pass's end belong to the next pass: masked out, never looked up
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[4 * u + q] = gather(x, i + q < count ? cc[u][q] : wbase);
            }
        }
    }
    // (the last passes re-issue the block's last one: the count in flight stays a constant)
    tile_issue_window<T, TRIPS>(cw, dw, PACK ? 1 : stage_ok, x);
    tile_issue_entries<T, NT, CH, PACK>(fc, fk, fv, de, tcol, tkey, tval);
    __syncthreads();  // xs is in place; everybody is done with the previous pass's wave slots
    if (probe & 1) {  // (measurement only: loads, staging and barriers, nothing else)
        __syncthreads();
        return;
    }
    // ---- a lane's quads: products, the runs that close inside the quad, the open ends
    T lead[kQuads], tail[kQuads];   // sum before the quad's first head (the whole quad without one) / from its last head on
    int tail_row[kQuads];           // local row of the run `tail` belongs to (-1: none)
    bool has_head[kQuads];
#pragma unroll
    for (int u = 0; u < kQuads; ++u) {
        const int i = u * 4 * kTileBlock + 4 * t;
        T cur = T(0);
        int row = -1;
        bool seen = false;
        lead[u] = T(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = i + q < count;
            const unsigned word = (unsigned)cc[u][q];
            // key = head flag | local row: out of the column word of a packed plan, else out of the key array
            unsigned key;
            T xq;
            if constexpr (PACK) {
                key = ((word >> 16) & (unsigned)kTileHead) | ((word >> kTilePackShift) & (unsigned)kTileRowMask);
                xq = xs[in ? (int)(word & kTilePackColMask) : 0];
            } else {
                key = (q & 1 ? ck[u][q >> 1] >> 16 : ck[u][q >> 1]) & 0xffffu;
                if (wlen) xq = xs[in ? (int)word - wbase : 0];
                else if constexpr (GA) xq = xg[4 * u + q];
                else xq = (probe & 2) ? T(1) : xv[4 * u + q];
            }
            const T pr = in ? cv[u][q] * xq : T(0);
            // an entry behind the pass's end closes whatever run is open and opens nothing
            if (!in || (key & kTileHead)) {
                if (!seen) lead[u] = cur;
                else if (row >= 0 && !(probe & 4)) acc[row] += cur;  // a run inside the quad: this lane owns its row
                seen = true;
                cur = pr;
                row = in ? (int)(key & kTileRowMask) : -1;
            } else {
                cur += pr;
            }
        }
        has_head[u] = seen;
        if (!seen) lead[u] = cur;
        tail[u] = seen ? cur : T(0);
        tail_row[u] = seen ? row : -1;
    }
    if (probe & 4) return;  // (measurement only: no cross-lane run sums)
    // ---- runs that cross lanes: right-to-left segmented scan of `lead` over the lanes of a wavefront,
    // R(s) = lead(s) + (s has a head ? 0 : R(s + 1)): what a run ending in or after lane s collects from s on
    T chain[kQuads];
    bool closed[kQuads];
#pragma unroll
    for (int u = 0; u < kQuads; ++u) {
        T r = lead[u];
        int h = has_head[u] ? 1 : 0;
        // short rows: every lane of the wavefront holds a head, every chain ends in the next lane -- no scan
        // short rows: where every lane of the wavefront holds a head every chain ends in the next lane -- no scan
        if (__ballot(h != 0) != ~0ull) chain_scan(r, h, lane);  // wave-uniform
        if (lane == 0) {  // what the previous wavefront's open run collects from this one, and whether it ends here
            wave_r[u * kTileWaves + wave] = r;
            wave_h[u * kTileWaves + wave] = h;
        }
        // what lane s's tail run collects: the chain starting at lane s + 1
        // (wave_shl:1 -- a DPP move across the whole wavefront, gfx9 only -- where __shfl_down would take three trips
        // through the LDS crossbar; lane 63 is overwritten below)
        chain[u] = row_down<0x130>(r);
        closed[u] = dpp_i32<0x130>(h) != 0;
        if (lane == 63) {
            chain[u] = T(0);
            closed[u] = false;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kQuads; ++u) {
        if (tail_row[u] >= 0) {
            T s = chain[u];
            if (!closed[u]) {  // the run goes on into the following wavefronts (and quads): slot order = entry order
                for (int w = u * kTileWaves + wave + 1; w < kQuads * kTileWaves; ++w) {
                    s += wave_r[w];
                    if (wave_h[w]) break;
                }
            }
            acc[tail_row[u]] += tail[u] + s;
        }
    }
    // no barrier here: the next pass stores xs (last read before the second barrier above) and rewrites the
    // wave slots only behind its own first barrier, which every lane reaches after this point
    if (sblock_rows && live && (d.w & kTilePassLast)) {  // wave-uniform
        // a stream's block ends here: its rows go out, the accumulators start again at 0 for the next block, whose
        // first passes' loads are already on their way (the next accumulator update is behind the next pass's barrier)
        // (the queue is drained here once per block -- vmcnt(0): the block's rows come by a vector load, the youngest in the
        // queue.  Fetching them a pass early by a scalar load removes the drain and buys nothing measurable -- road-like
        // 132.7 -> 132.3 us, and 2 % lost on the plans with gather passes: what is waited for are the next block's first
        // passes, which are needed next anyway.)
        __syncthreads();
        const int2 br = sblock_rows[bi++];
        for (int i = t; i < rows_per_block; i += kTileBlock) {
            if (i < br.y) y[br.x + i] = acc[i];
            acc[i] = T(0);
        }
    }
}

// One workgroup per row block; passes software-pipelined: while pass p is multiplied and summed, the x slices of
// passes p + 1, p + 2 and the entries of passes p + 1 .. p + 3 are already on their way into registers.
// pass = {first entry (multiple of 4), entries, first staged column (multiple of 4), staged columns (0: gather)}
// (second launch bound = wavefronts per SIMD: two resident workgroups per CU)
template <typename T, bool NT, int CH, int TRIPS, bool PACK, bool GA = false>
__global__ __launch_bounds__(kTileBlock, 4) void csr_tile(int num_blocks, int rows_per_block,
                                                                          int stage_ok, int probe,
                                                                          const int4 *__restrict__ work, T *__restrict__ slab,
                                                                          const int *__restrict__ block_row,
                                                                          const int *__restrict__ block_pass,
                                                                          const int4 *__restrict__ pass_desc,
                                                                          const int *__restrict__ tcol,
                                                                          const unsigned short *__restrict__ tkey,
                                                                          const T *__restrict__ tval,
                                                                          const int *__restrict__ stream_block,
                                                                          const int2 *__restrict__ sblock_rows,
                                                                          const T *__restrict__ x, T *__restrict__ y) {
    using V4 = typename vec4v<T>::type;
    constexpr int kQuads = CH / (4 * kTileBlock);  // groups of four consecutive entries a lane holds per pass
    // LDS: wave slots | acc[rows_per_block] | xs[widest staged window]
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
    T *wave_r = reinterpret_cast<T *>(tile_smem);
    int *wave_h = reinterpret_cast<int *>(wave_r + kQuads * kTileWaves);
    T *acc = reinterpret_cast<T *>(tile_smem + kTileSlotBytes);
    T *xs = acc + rows_per_block;

    // workgroup ids go round-robin over the 8 XCDs: give every XCD one contiguous eighth of the row blocks, in
    // order, so that neighbouring blocks -- whose x slices overlap -- find each other's lines in the same L2
    // (a stream plan -- work == nullptr -- has that mapping built in: stream x + 8 j is XCD x's, tile_make_streams)
    const int per_xcd = (num_blocks + 7) >> 3;
    const int id = work ? (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : (int)blockIdx.x;
    if ((work && (int)(blockIdx.x >> 3) >= per_xcd) || id >= num_blocks) return;
    const int t = threadIdx.x;
    // work (optional): the row blocks' passes dealt out to SEVERAL workgroups each -- {block, first pass, end pass,
    // slab}: a block of a few very long rows has far more passes than one workgroup should walk; every workgroup
    // then leaves its accumulators in its own slab and tile_slab_finish adds a row's slabs in order
    // Without work: a STREAM -- block_pass[id] .. block_pass[id + 1] are the passes (in pass_desc, stream order) of all
    // the blocks this workgroup walks, one after the other without letting the loads run dry in between; a block's last
    // pass carries kTilePassLast, behind it the accumulators go to y (rows from sblock_rows) and start again at 0.
    int p0, p1, bi = 0, nrows = 0;
    T *out = nullptr;
    if (work) {
        const int4 w = work[id];
        p0 = w.y;
        p1 = w.z;
        out = slab + (size_t)w.w * rows_per_block;
        nrows = block_row[w.x + 1] - block_row[w.x];  // <= rows_per_block
    } else {
        p0 = block_pass[id];
        p1 = block_pass[id + 1];
        bi = stream_block[id];
    }
    const int2 *flush_rows = work ? nullptr : sblock_rows;
    for (int i = t; i < rows_per_block; i += kTileBlock) acc[i] = T(0);
    if (p0 < p1) {
        // entries: four rotating register sets (the current pass + kTileAhead = 3 in flight), x slices: two
        SPMV_TILE_ENTRY_REGS(e0);
        SPMV_TILE_ENTRY_REGS(e1);
        v4u wa[TRIPS], wb[TRIPS];
        const int pl = p1 - 1;
        // (GA: two sets of gathered values, this pass's and the next one's; one dummy element otherwise)
        T xa[GA ? CH / kTileBlock : 1], xb[GA ? CH / kTileBlock : 1];
#define SPMV_TILE_PASS(cur, fill, wcur, D, DW, DE, P) SPMV_TILE_PASS_GA(cur, fill, wcur, D, DW, DE, P, cur, D, xa, xa)
#define SPMV_TILE_PASS_GA(cur, fill, wcur, D, DW, DE, P, nxt, DN, XG, XN)                                            \
    tile_pass<T, NT, CH, TRIPS, PACK, GA>(SPMV_TILE_ENTRY_ARGS(cur), wcur, SPMV_TILE_ENTRY_ARGS(fill), D, DW, DE, (P) <= pl, \
                                          stage_ok, probe, bi, flush_rows, rows_per_block, y, acc, xs, wave_r, wave_h, tcol, tkey, \
                                          tval, x, nxt##_c, DN, XG, XN)
        SPMV_TILE_ENTRY_REGS(e2);
        SPMV_TILE_ENTRY_REGS(e3);
        // descriptors of passes p .. p + 3 (clamped to the block's last pass), replaced one per pass
        int4 d0 = pass_desc[p0], d1 = pass_desc[min(p0 + 1, pl)], d2 = pass_desc[min(p0 + 2, pl)], d3 = pass_desc[min(p0 + 3, pl)];
        // (in the order the passes send them out: slice p, entries p + 1; slice p + 1, entries p + 2)
        // (the fences keep the compiler from sorting them by kind: what counts is their order in the queue)
        tile_issue_entries<T, NT, CH, PACK>(SPMV_TILE_ENTRY_ARGS(e0), d0, tcol, tkey, tval);
        __builtin_amdgcn_sched_barrier(0);
        tile_issue_window<T, TRIPS>(wa, d0, PACK ? 1 : stage_ok, x);
        __builtin_amdgcn_sched_barrier(0);
        tile_issue_entries<T, NT, CH, PACK>(SPMV_TILE_ENTRY_ARGS(e1), d1, tcol, tkey, tval);
        __builtin_amdgcn_sched_barrier(0);
        tile_issue_window<T, TRIPS>(wb, d1, PACK ? 1 : stage_ok, x);
        __builtin_amdgcn_sched_barrier(0);
        tile_issue_entries<T, NT, CH, PACK>(SPMV_TILE_ENTRY_ARGS(e2), d2, tcol, tkey, tval);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PACK) {
            // the loop always runs four passes per trip: a pass it may skip leaves the compiler's count of what is in
            // flight different on the two paths, and every wait behind the merge is then for the larger one.  The last
            // one to three passes follow the loop, where nothing merges back.
            int p = p0;
            for (; p + 4 <= p1; p += 4) {  // wave-uniform
                const int4 n0 = pass_desc[min(p + 4, pl)];
                SPMV_TILE_PASS(e0, e3, wa, d0, d2, d3, p);
                const int4 n1 = pass_desc[min(p + 5, pl)];
                SPMV_TILE_PASS(e1, e0, wb, d1, d3, n0, p + 1);
                const int4 n2 = pass_desc[min(p + 6, pl)];
                SPMV_TILE_PASS(e2, e1, wa, d2, n0, n1, p + 2);
                const int4 n3 = pass_desc[min(p + 7, pl)];
                SPMV_TILE_PASS(e3, e2, wb, d3, n1, n2, p + 3);
                d0 = n0;
                d1 = n1;
                d2 = n2;
                d3 = n3;
            }
            if (p < p1) {  // (what they send out repeats the last pass: d0 .. d3 are clamped to it)
                SPMV_TILE_PASS(e0, e3, wa, d0, d2, d3, p);
                if (p + 1 < p1) {
                    SPMV_TILE_PASS(e1, e0, wb, d1, d3, d3, p + 1);
                    if (p + 2 < p1) SPMV_TILE_PASS(e2, e1, wa, d2, d3, d3, p + 2);
                }
            }
        } else if constexpr (GA) {
            // gather ahead: pass p works on the values gathered during pass p - 1 and sends out those of pass p + 1
            // (the prologue sends out pass p0's: its entries are the oldest loads in the queue)
            tile_issue_gathers<T, kQuads>(xa, e0_c, d0, stage_ok, x);
            for (int p = p0; p < p1; p += 4) {  // wave-uniform
                const int4 n0 = pass_desc[min(p + 4, pl)];
                SPMV_TILE_PASS_GA(e0, e3, wa, d0, d2, d3, p, e1, d1, xa, xb);
                const int4 n1 = pass_desc[min(p + 5, pl)];
                if (p + 1 < p1) SPMV_TILE_PASS_GA(e1, e0, wb, d1, d3, n0, p + 1, e2, d2, xb, xa);
                const int4 n2 = pass_desc[min(p + 6, pl)];
                if (p + 2 < p1) SPMV_TILE_PASS_GA(e2, e1, wa, d2, n0, n1, p + 2, e3, d3, xa, xb);
                const int4 n3 = pass_desc[min(p + 7, pl)];
                if (p + 3 < p1) SPMV_TILE_PASS_GA(e3, e2, wb, d3, n1, n2, p + 3, e0, n0, xb, xa);
                d0 = n0;
                d1 = n1;
                d2 = n2;
                d3 = n3;
            }
        } else {
            // (plans with gather passes wait for their gathers in every pass anyway -- which drains the queue down to
            // what went out behind them -- and with the tail passes written out these instantiations spill)
            for (int p = p0; p < p1; p += 4) {  // wave-uniform
                const int4 n0 = pass_desc[min(p + 4, pl)];
                SPMV_TILE_PASS(e0, e3, wa, d0, d2, d3, p);
                const int4 n1 = pass_desc[min(p + 5, pl)];
                if (p + 1 < p1) SPMV_TILE_PASS(e1, e0, wb, d1, d3, n0, p + 1);
                const int4 n2 = pass_desc[min(p + 6, pl)];
                if (p + 2 < p1) SPMV_TILE_PASS(e2, e1, wa, d2, n0, n1, p + 2);
                const int4 n3 = pass_desc[min(p + 7, pl)];
                if (p + 3 < p1) SPMV_TILE_PASS(e3, e2, wb, d3, n1, n2, p + 3);
                d0 = n0;
                d1 = n1;
                d2 = n2;
                d3 = n3;
            }
        }
#undef SPMV_TILE_PASS
#undef SPMV_TILE_PASS_GA
    }
    if (out) {  // (a stream has written its rows block by block)
        __syncthreads();
        for (int i = t; i < nrows; i += kTileBlock) out[i] = acc[i];
    }
}

// The EXPANSION of x for a plan with gather passes (round 3).  A 64-lane gather of scattered columns costs the texture
// addresser 64 lines for 256 useful bytes, whether they hit L2 or not: config 5's short rows spent 315 of their 491 us
// there.  An expanded plan gathers nothing: every pass owns a SEGMENT of a second vector x' with one value per entry --
// the pass's x values, ordered by (32 KiB slice of x, row, column) -- and runs as a PACKED pass whose window is that
// segment (csr_tile<.., PACK>, unchanged: the column word holds the entry's place in the segment).  tile_expand fills
// x' ahead of every csr_tile launch, walking the plan's entries in SLICE order: a workgroup copies its slice of x to
// LDS with coalesced loads, looks the columns up there, and writes the values to their segments -- the entries of one
// (slice, pass) lie together in both orders, so the writes are runs, not single values.  The products and the order in
// which a row's are added are those of the gather passes: the same bits.
// Entry k (slice order) goes to x'[k + delta[r]], r = the RUN it belongs to -- consecutive entries whose places in x'
// are consecutive too (the entries of one (slice, pass), or several that happen to touch): bit 15 of lcol marks a run's
// first entry, group_run holds the run of every 64th entry of a chunk, and a wavefront numbers its lanes' runs with one
// ballot.  The chunk's offsets (its first kExpandRunsLds runs) and group numbers are copied to LDS beside the slice:
// 2 bytes read and one value written per entry: 141 us on config 5's 63 M short-row entries.  (A place per entry is 4
// bytes more: 189 us; the offsets fetched from memory per entry -- a second, dependent trip -- 252 us; with the places
// streamed in order, no offsets at all, 108 us.)
// chunk = {slice, first k, entries, first group}; chunk_runs = {first run, runs}.
constexpr int kExpandBlock = 512;
constexpr int kExpandChunk = 16384;
constexpr int kExpandRunsLds = 1024;
constexpr unsigned kExpandRunStart = 0x8000u;
template <typename T>
constexpr int tile_slice_cols() {
    return kTileTrips * kTileTripBytes / (int)sizeof(T);
}
template <typename T>
__global__ __launch_bounds__(kExpandBlock) void tile_expand(int N, int probe, const int4 *__restrict__ chunk, const int2 *__restrict__ chunk_runs,
                                                            const unsigned short *__restrict__ lcol, const unsigned *__restrict__ group_run,
                                                            const unsigned *__restrict__ delta, const T *__restrict__ x,
                                                            T *__restrict__ xe) {
    constexpr int W = tile_slice_cols<T>(), kPer = 16 / (int)sizeof(T);
    static_assert(W <= (int)kExpandRunStart, "a column inside its slice leaves bit 15 free");
    __shared__ __attribute__((aligned(16))) T xs[W];
    __shared__ unsigned dl[kExpandRunsLds];
    __shared__ unsigned gl[kExpandChunk / 64];
    const int4 c = chunk[blockIdx.x];
    const int2 cr = chunk_runs[blockIdx.x];
    const int t = threadIdx.x, lane = t & 63;
    const long long base = (long long)c.x * W;
    const int wlen = (int)min((long long)W, (long long)N - base);
    const int whole = wlen / kPer * kPer;
    // (x is 16-byte aligned -- the launch checks -- and a slice begins at a multiple of 32 KiB)
    const v4u *src = reinterpret_cast<const v4u *>(x + base);
    for (int j = t * kPer; j < whole; j += kExpandBlock * kPer) *reinterpret_cast<v4u *>(xs + j) = src[j / kPer];
    if (t < wlen - whole) xs[whole + t] = x[base + whole + t];
    const int n = c.z;
    for (int j = t; j < min(cr.y, kExpandRunsLds); j += kExpandBlock) dl[j] = delta[cr.x + j];
    for (int j = t; j < (n + 63) / 64; j += kExpandBlock) gl[j] = group_run[c.w + j] - (unsigned)cr.x;  // the run inside the chunk
    __syncthreads();
    const unsigned short *lc = lcol + c.y;
    const unsigned long long upto = (2ull << lane) - 1;  // lanes 0 .. lane
    constexpr int U = 8;  // loads of U trips in flight before the first store
    for (int i0 = 0; i0 < n; i0 += U * kExpandBlock) {  // wave-uniform
        unsigned raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kExpandBlock + t;
            raw[u] = i < n ? (unsigned)stream_load<true>(lc + i) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kExpandBlock + t;
            const bool in = i < n;
            // the runs that begin in lanes 1 .. lane come on top of the run the group's first entry belongs to
            const unsigned long long starts = __ballot(in && (raw[u] & kExpandRunStart)) & ~1ull;
            if (in) {
                const unsigned r = gl[i >> 6] + (unsigned)__popcll(starts & upto);
                // (probe, measurement only: bit 0 -- the values go out in slice order, as a stream)
                const unsigned dv = (probe & 1) ? 0u : r < (unsigned)kExpandRunsLds ? dl[r] : delta[(unsigned)cr.x + r];
                xe[(unsigned)(c.y + i) + dv] = xs[raw[u] & (kExpandRunStart - 1)];
            }
        }
    }
}

// y[row_map[v]] = the sum of the slabs of virtual row v's block (long-row plans), in a fixed order: the block's
// work items are cut into kFinishGroups contiguous groups, a thread adds one group's slabs in item order (loads
// independent of each other: several in flight), then the groups' sums are added in group order.
constexpr int kFinishRows = 64, kFinishGroups = 16;
template <typename T>
__global__ __launch_bounds__(kFinishRows *kFinishGroups) void tile_slab_finish(
    int rows, int rows_per_block, const int *__restrict__ block_row, const int *__restrict__ block_of_row,
    const int *__restrict__ item_first, const int *__restrict__ row_map, const T *__restrict__ slab, T *__restrict__ y) {
    __shared__ T part[kFinishGroups][kFinishRows];
    const int lane = threadIdx.x % kFinishRows, g = threadIdx.x / kFinishRows;
    const int v = blockIdx.x * kFinishRows + lane;
    T s = T(0);
    if (v < rows) {
        const int b = block_of_row[v];
        const int local = v - block_row[b];
        const int w0 = item_first[b], n = item_first[b + 1] - w0;
        const int lo = w0 + (int)((long long)n * g / kFinishGroups), hi = w0 + (int)((long long)n * (g + 1) / kFinishGroups);
        const T *col = slab + local;
#pragma unroll 4
        for (int w = lo; w < hi; ++w) s += col[(size_t)w * rows_per_block];
    }
    part[g][lane] = s;
    __syncthreads();
    if (g == 0 && v < rows) {
        T total = part[0][lane];
#pragma unroll
        for (int k = 1; k < kFinishGroups; ++k) total += part[k][lane];
        y[row_map[v]] = total;
    }
}
// The remainder of a packed plan: the entries of windows too sparse for a pass (tile_plan.hpp), one lane per row that has
// any -- its few products in (row, column) order added to what the tiles left in y.  One owner per row, launched behind
// the tiles on the same stream: the same bits every time.
template <typename T>
__global__ __launch_bounds__(256) void tile_remainder(int rows, const int *__restrict__ rrow, const int *__restrict__ rptr,
                                                      const int *__restrict__ rcol, const T *__restrict__ rval,
                                                      const T *__restrict__ x, T *__restrict__ y) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= rows) return;
    T s = T(0);
    for (int e = rptr[k]; e < rptr[k + 1]; ++e) s += rval[e] * x[rcol[e]];
    y[rrow[k]] += s;
}
#undef SPMV_TILE_ENTRY_REGS
#undef SPMV_TILE_ENTRY_ARGS

}  // namespace spmv
