// spmv_hll.hip -- HLL side of the C-ABI: the flat slab, its workgroup windows and x-window plan,
// the device builder from a resident CSR matrix, launchers, timing.  Replaces the per-hack
// allocations and launches of /root/reference/main_cuda.cu:369-455, :545-568, :613-637, :731-744.
#include <chrono>
#include "spmv_internal.hpp"

#include "plan_kernels.hpp"

namespace {

// first pass of the device builder (see hll_fill_from_csr in hll_kernels.hpp): per-hack maximum row length
__global__ __launch_bounds__(kBlock) void hll_hack_maxnz(int M, int hacks, const int *__restrict__ row_ptr,
                                                         int *__restrict__ maxnz) {
    const int h = blockIdx.x * kBlock + threadIdx.x;
    if (h >= hacks) return;
    const int r0 = h * kHack, r1 = min(r0 + kHack, M);
    int m = 0;
    for (int r = r0; r < r1; ++r) m = max(m, row_ptr[r + 1] - row_ptr[r]);
    maxnz[h] = m;
}

}  // namespace

// ----------------------------------------------------------------- HLL
namespace {

// Cut the rows of the flat slab into workgroup windows for hll_lds: consecutive rows whose
// slots, counted from the even slot at or below the first row's start, fit `cap` (at most
// kStreamRowsCap rows).  A row that alone does not fit gets a window of its own.  Returns
// the widest window (in slots, from its even base).
long long hll_build_blocks(int M, int hacks, const long long *off, const int *mz, int cap,
                           std::vector<int4> &desc) {
    desc.clear();
    (void)hacks;
    long long widest = 0;
    auto start_of = [&](int r) { return off[r / kHack] + (long long)(r % kHack) * mz[r / kHack]; };
    int r = 0;
    while (r < M) {
        const long long s0 = start_of(r);
        const long long base = s0 & ~1LL;
        int r1 = r + 1;  // the first row is always taken (even if it alone exceeds cap)
        int maxlen = mz[r / kHack];
        // (skew_cut: a window on the border between a hack of short rows and one of long rows -- see spmv_internal.hpp)
        while (r1 < M && r1 - r < kStreamRowsCap && start_of(r1) + mz[r1 / kHack] - base <= cap &&
               !skew_cut(r1 - r, maxlen, mz[r1 / kHack])) {
            maxlen = std::max(maxlen, mz[r1 / kHack]);
            ++r1;
        }
        const long long span = start_of(r1 - 1) + mz[(r1 - 1) / kHack] - base;
        if (r1 - r > 1 || span <= cap) widest = std::max(widest, span);
        desc.push_back(int4{r, r1 - r, (int)(s0 & 0xffffffffLL), (int)(s0 >> 32)});
        r = r1;
    }
    return widest;
}

}  // namespace

namespace {

// Windows for hll_lds_local: hll_build_blocks' cut at `cap` slots with the line limit on top
// (see csr_build_local).  ja is the flat host slab.  false: keep the gather kernel.
bool hll_build_local(int M, int N, const long long *off, const int *mz, const int *ja, long long slots_padded,
                     int cap, int lines_max, const std::vector<int4> &baseline, LocalPlan &plan) {
    constexpr int line_shift = 4;  // fp64: 16 per 128-byte line
    const int total_lines = (int)(((long long)N + 15) >> line_shift);
    std::vector<int> stamp((size_t)total_lines + 1, -1), rank((size_t)total_lines + 1, 0), cur;
    plan.lcol.assign((size_t)slots_padded + kPad, 0);
    plan.desc.clear();
    plan.hll_ldesc.clear();
    plan.lines.clear();
    int widest = 0;
    auto start_of = [&](int r) { return off[r / kHack] + (long long)(r % kHack) * mz[r / kHack]; };
    int r = 0;
    while (r < M) {
        const long long s0 = start_of(r);
        const long long base = s0 & ~1LL;
        const int blk = (int)plan.desc.size();
        cur.clear();
        int r1 = r, maxlen = 0;
        while (r1 < M && r1 - r < kStreamRowsCap && start_of(r1) + mz[r1 / kHack] - base <= cap &&
               !skew_cut(r1 - r, maxlen, mz[r1 / kHack])) {
            const size_t before = cur.size();
            const long long a = start_of(r1);
            for (long long k = a; k < a + mz[r1 / kHack]; ++k) {
                const int l = ja[k] >> line_shift;
                if (stamp[l] != blk) {
                    stamp[l] = blk;
                    cur.push_back(l);
                }
            }
            if ((int)cur.size() > lines_max) {
                for (size_t k = before; k < cur.size(); ++k) stamp[cur[k]] = -1;
                cur.resize(before);
                break;
            }
            maxlen = std::max(maxlen, mz[r1 / kHack]);
            ++r1;
        }
        if (r1 == r) return false;  // a row that alone exceeds the stage or the line limit
        if (cur.empty()) cur.push_back(0);  // rows without slots: the kernel still stages one line
        std::sort(cur.begin(), cur.end());
        for (size_t k = 0; k < cur.size(); ++k) rank[cur[k]] = (int)k;
        for (long long k = s0; k < start_of(r1 - 1) + mz[(r1 - 1) / kHack]; ++k)
            plan.lcol[k] = (unsigned short)((rank[ja[k] >> line_shift] << line_shift) | (ja[k] & 15));
        plan.desc.push_back(int4{r, r1 - r, (int)(s0 & 0xffffffffLL), (int)(s0 >> 32)});
        plan.hll_ldesc.push_back(int4{(int)plan.lines.size(), (int)cur.size(),
                                      (int)(start_of(r1 - 1) + mz[(r1 - 1) / kHack] - base), 0});
        plan.lines.insert(plan.lines.end(), cur.begin(), cur.end());
        widest = std::max(widest, (int)cur.size());
        r = r1;
        if ((plan.desc.size() & 1023) == 0) {
            const size_t plain = std::lower_bound(baseline.begin(), baseline.end(), r,
                                                  [](const int4 &d, int row) { return d.x < row; }) -
                                 baseline.begin();
            if (plan.desc.size() > plain + plain / 5 + 16) return false;
        }
    }
    if (plan.desc.size() > baseline.size() + baseline.size() / 5 + 1) return false;
    plan.stage_lines = std::max(kLocalLineQuantum,
                                (widest + kLocalLineQuantum - 1) / kLocalLineQuantum * kLocalLineQuantum);
    plan.lines.insert(plan.lines.end(), (size_t)kLocalLinesMax, 0);
    return true;
}

// offsets of the hacks in the flat slab: every hack starts on an even slot
long long hll_offsets(int total_rows, const std::vector<int> &mz, std::vector<long long> &off,
                      long long &true_slots) {
    const int H = (int)mz.size();
    off.assign((size_t)H + 1, 0);
    true_slots = 0;
    for (int h = 0; h < H; ++h) {
        const int rows = (h == H - 1) ? total_rows - h * kHack : kHack;
        const long long s = (long long)rows * mz[h];
        true_slots += s;
        off[h + 1] = off[h] + ((s + 1) & ~1LL);
    }
    return off[H];
}

// workgroup windows, small arrays and vectors of a handle whose JA / AS are already on the device
// row_seg[r] = (first slot of row r relative to its window's even base) | (slots of the row << 16), for the
// rows of the x-window plan's windows: what hll_lds_local's row-sum phase needs, in one word.
__global__ __launch_bounds__(kBlock) void hll_row_segments(int num_blocks, const int4 *__restrict__ desc,
                                                           const long long *__restrict__ hack_off,
                                                           const int *__restrict__ maxnz,
                                                           unsigned *__restrict__ row_seg) {
    const int b = blockIdx.x;
    if (b >= num_blocks) return;
    const int4 d = desc[b];
    const long long base = (((long long)d.w << 32) | (unsigned)d.z) & ~1LL;
    for (int q = threadIdx.x; q < d.y; q += kBlock) {
        const int r = d.x + q, h = r / kHack;
        const int m = maxnz[h];
        const long long lo = hack_off[h] + (long long)(r % kHack) * m - base;
        row_seg[r] = (unsigned)lo | ((unsigned)m << 16);
    }
}

// row_seg for the rows of the x-window plan's windows (hack tables and window descriptors are on the device)
int hll_fill_row_segments(spmv_hll_dev *m) {
    HIP_TRY(hipMalloc((void **)&m->row_seg, std::max<size_t>((size_t)m->M, 1) * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(m->row_seg, 0, std::max<size_t>((size_t)m->M, 1) * sizeof(unsigned), g_stream));
    hipLaunchKernelGGL(hll_row_segments, dim3(m->local_blocks), dim3(kBlock), 0, g_stream, m->local_blocks, m->ldesc4,
                       m->hack_off, m->maxnz, m->row_seg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g_stream));
    m->device_bytes += (size_t)m->M * sizeof(unsigned);
    return 0;
}

int hll_finish_handle(spmv_hll_dev *m, int total_rows, int N, const std::vector<long long> &off,
                      const std::vector<int> &mz, long long true_slots, bool upload_maxnz, const int *ja_host,
                      int matrix_rows = -1, int row0 = 0, const double *as_host = nullptr) {
    m->M_total = matrix_rows < 0 ? total_rows : matrix_rows;
    m->row0 = row0;
    const int H = (int)mz.size();
    // like the CSR stream kernel: larger stages for matrices that have plenty of work
    const int cap = true_slots >= (16LL << 20) ? kHllCap : kHllCap / 2;
    std::vector<int4> hdesc;
    const long long widest =
        std::max<long long>(2 * kStreamUnit, hll_build_blocks(total_rows, H, off.data(), mz.data(), cap, hdesc));
    m->M = total_rows;
    m->N = N;
    m->hacks = H;
    m->slots = true_slots;
    // what hll_lds gets: .y = rows | slots of the window (from its even base) << 16 -- the kernel then knows how much to
    // stage without first asking the hack tables (0 in the upper half: a single row of more than 65535 slots, which has
    // its own path in the kernel)
    for (int4 &d : hdesc) {
        const int last = d.x + d.y - 1;
        const long long s0 = (((long long)d.w << 32) | (unsigned)d.z) & ~1LL;
        const long long span = off[(size_t)(last / kHack)] + (long long)(last % kHack) * mz[(size_t)(last / kHack)] + mz[(size_t)(last / kHack)] - s0;
        d.y |= span <= 0xffff ? (int)span << 16 : 0;
    }
    m->num_blocks = (int)hdesc.size();
    m->stage_slots = (int)std::min<long long>(kHllCap, (widest + kStreamUnit - 1) / kStreamUnit * kStreamUnit);
    int rc = 0;
    if (!m->hack_off) rc |= upload_array(&m->hack_off, off.data(), off.size(), 0);
    if (!rc && upload_maxnz) rc |= upload_array(&m->maxnz, mz.data(), mz.size(), 1);
    if (!rc) rc |= upload_array(&m->hdesc, hdesc.data(), hdesc.size(), 1);
    if (!rc) {
        const size_t x_bytes = std::max<size_t>((size_t)N, 1) * sizeof(double) + kLineBytes;  // whole-line reads
        hipError_t e = hipMalloc((void **)&m->x, x_bytes);
        if (e == hipSuccess) e = hipMalloc((void **)&m->y, std::max<size_t>((size_t)m->M_total, 1) * sizeof(double));
        if (e == hipSuccess) e = hipMemset(m->x, 0, x_bytes);
        if (e == hipSuccess) e = hipMemset(m->y, 0, std::max<size_t>((size_t)m->M_total, 1) * sizeof(double));
        if (e != hipSuccess) rc = fail("hipMalloc(x/y) failed: %s", hipGetErrorString(e));
    }
    m->device_bytes = off.size() * 8 + mz.size() * 4 + ((size_t)off[H] + kPad) * 12 + hdesc.size() * 16 +
                      ((size_t)N + (size_t)total_rows) * 8;
    // the x-window kernel: windows of 2048 slots, 16-bit local JA (needs the slab on the host)
    if (!rc && ja_host && g_stream_local && true_slots > 0) {
        std::vector<int4> plain;
        LocalPlan local;
        hll_build_blocks(total_rows, H, off.data(), mz.data(), 2048, plain);
        if (hll_build_local(total_rows, N, off.data(), mz.data(), ja_host, off[H], 2048, kLocalLinesMax, plain, local)) {
            rc |= upload_array(&m->ldesc4, local.desc.data(), local.desc.size(), 1);
            if (!rc) rc |= upload_array(&m->ldesc, local.hll_ldesc.data(), local.hll_ldesc.size(), 1);
            if (!rc) rc |= upload_array(&m->lines, local.lines.data(), local.lines.size(), 0);
            if (!rc) rc |= upload_array(&m->lja, local.lcol.data(), local.lcol.size(), 0);
            if (!rc) {
                m->local_blocks = (int)local.desc.size();
                m->local_stage_lines = local.stage_lines;
                m->local_lines = (long long)local.lines.size() - kLocalLinesMax;
                m->device_bytes += local.desc.size() * 32 + local.lines.size() * 4 + local.lcol.size() * 2;
            }
        }
    }
    if (!rc && m->local_blocks > 0) rc = hll_fill_row_segments(m);
    // no x-window plan (columns too scattered): the 2-D tiles over the slab's rows, padding slots included -- the
    // rows of hack h are maxnz[h] slots each, starting at hack_off[h] + i * maxnz[h] (needs the slab on the host)
    // (auto: not for a skewed slab -- a hack whose rows are 16 times the mean is 32 rows of mostly padding, and tiles
    // over such rows lose to hll_lds: webbase-like stand-in, 1 M rows, 455 vs 159 us)
    bool skewed = false;
    if (g_stream_tile < 0 && total_rows > 0) {
        const long long mean = std::max<long long>(1, off[H] / total_rows);
        for (int h = 0; h < H && !skewed; ++h) skewed = mz[(size_t)h] > 16 * mean;
    }
    // (the plan is built on the device from the slab there; the host copies, where the caller has them, are the fallback)
    const bool slab_at_hand = (ja_host && as_host) || (g_tile_plan_on_device && m->JA && m->AS);
    if (!rc && !skewed && m->local_blocks == 0 && slab_at_hand && true_slots > 0 && off[H] < 0x7fffffffLL) {
        std::vector<int> row_begin((size_t)total_rows), row_len((size_t)total_rows);
        for (int r = 0; r < total_rows; ++r) {
            const int h = r / kHack;
            row_len[(size_t)r] = mz[(size_t)h];
            row_begin[(size_t)r] = (int)(off[(size_t)h] + (long long)(r % kHack) * mz[(size_t)h]);
        }
        rc = csr_tiles_from_rows_f64(total_rows, m->M_total, row0, N, row_begin.data(), row_len.data(), true_slots, ja_host,
                                     as_host, &m->tiles, m->JA, m->AS);
        if (!rc && m->tiles) m->device_bytes += m->tiles->device_bytes;
    }
    const double mean = total_rows ? (double)true_slots / total_rows : 0.0;
    m->lanes_per_row = std::min(32, std::max(2, pow2_floor(std::max(2, (int)(mean / 2.0 + 0.5)))));
    // Mid-size slabs that get neither plan (scattered columns, below the tile plans' size): the lane-group kernel beats
    // hll_lds by 10-27 % on every such stand-in of the reference's list (thermal1-size 6.7 vs 8.4 us, thermomech_TK-size
    // 7.2 vs 9.2, cop20k_A-size 19.1 vs 21.5, mac_econ-size 10.9 vs 12.2, amazon0302-size 12.0 vs 13.4;
    // profiles/r3_reference_list_stand_ins.md) -- unless hacks are skewed (its lanes per row are fixed); the rule CSR
    // upload has had since round 2
    int widest_hack = 0;
    for (int h = 0; h < H; ++h) widest_hack = std::max(widest_hack, mz[(size_t)h]);
    m->auto_variant = SPMV_HLL_LDS;
    if (m->local_blocks == 0 && !m->tiles && true_slots < (20LL << 20) && widest_hack <= std::max(64.0, 8.0 * mean))
        m->auto_variant = SPMV_HLL_SUBWAVE;
    return rc;
}

}  // namespace

// The HLL twins of csr_build_patterns / csr_tune_patterns (spmv_csr.hip): the windows' pattern plan, built on the device
// from the plan's own arrays for streamed slabs of at least 12 slots per row whose tables hold at most a quarter of the
// slots, and kept only where upload times its own kernel at least 2 % faster with it.
static int hll_build_patterns(spmv_hll_dev *m) {
    if (g_local_patterns == 0 || m->local_blocks <= 0 || !m->lja || !m->ldesc4 || !m->row_seg || m->M <= 0) return 0;
    if (g_local_patterns < 0 && (m->slots * 10LL <= (128LL << 20) || m->slots < 12LL * m->M)) return 0;
    const int B = m->local_blocks;
    int *rowflag = nullptr, *pcount = nullptr;
    long long *pbase = nullptr;
    auto drop_tmp = [&] {
        (void)hipFree(rowflag);
        (void)hipFree(pcount);
        (void)hipFree(pbase);
    };
    hipError_t e = hipMalloc((void **)&rowflag, (size_t)m->M * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&pcount, (size_t)B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&pbase, (size_t)B * sizeof(long long));
    if (e == hipSuccess) e = hipMemsetAsync(rowflag, 0, (size_t)m->M * sizeof(int), g_stream);
    if (e != hipSuccess) {
        drop_tmp();
        return fail("pattern plan: allocation failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL((pat_mark<256, true>), dim3(B), dim3(256), 0, g_stream, B, m->ldesc4, (const int *)nullptr, m->lja, rowflag,
                       pcount, m->row_seg);
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
    for (int b = 0; b < B; ++b) {
        h_base[(size_t)b] = total;
        total += h_count[(size_t)b];
    }
    if ((g_local_patterns < 0 && total * 4 > m->slots) || total > 0x7ffffff0LL) {
        drop_tmp();
        return 0;
    }
    e = hipMalloc((void **)&m->ptab, ((size_t)total + 1024) * sizeof(unsigned short));
    if (e == hipSuccess) e = hipMalloc((void **)&m->rinfo, (size_t)m->M * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void **)&m->pdesc, (size_t)B * sizeof(int2));
    if (e == hipSuccess) e = hipMemsetAsync(m->ptab, 0, ((size_t)total + 1024) * sizeof(unsigned short), g_stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->rinfo, 0, (size_t)m->M * sizeof(unsigned), g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pbase, h_base.data(), (size_t)B * sizeof(long long), hipMemcpyHostToDevice, g_stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((pat_fill<256, true>), dim3(B), dim3(256), 0, g_stream, B, m->ldesc4, (const int *)nullptr, m->lja, rowflag,
                           pbase, m->rinfo, m->ptab, m->pdesc, m->row_seg);
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
    m->device_bytes += ((size_t)total + 1024) * 2 + (size_t)m->M * 4 + (size_t)B * 8;
    return 0;
}

// ~15 ms of the handle's own kernel ahead of the two searches (see csr_upload_impl: after an idle stretch a launch's time
// drifts by as much as the searches look for)
static void hll_settle(spmv_hll_dev *m) {
    const bool will_search = (m->ptab && g_local_patterns < 0) ||
                             (g_place_tries > 0 && (size_t)m->slots * sizeof(double) >= ((size_t)128 << 20) && m->AS);
    if (!will_search || !m->x || !m->y) return;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < 15.0) {
        bool bad = false;
        for (int i = 0; i < 16 && !bad; ++i) bad = hll_launch(m, SPMV_HLL_AUTO, m->x, m->y, g_stream) != 0;
        if (bad || hipStreamSynchronize(g_stream) != hipSuccess) break;
    }
}

static void hll_tune_patterns(spmv_hll_dev *m) {
    if (g_local_patterns >= 0 || !m->ptab || !m->x || !m->y) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return;
    }
    auto measure = [&](int patterns, float &us) {
        const int keep = g_local_patterns;
        g_local_patterns = patterns;
        int rc = 0;
        for (int i = 0; i < 2 && !rc; ++i) rc = hll_launch(m, SPMV_HLL_AUTO, m->x, m->y, g_stream);
        hipError_t e = rc ? hipErrorUnknown : hipEventRecord(e0, g_stream);
        for (int i = 0; i < 6 && e == hipSuccess && !rc; ++i) rc = hll_launch(m, SPMV_HLL_AUTO, m->x, m->y, g_stream);
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
    if (!ok || with_us > 0.98f * without_us) {
        (void)hipFree(m->ptab);
        (void)hipFree(m->rinfo);
        (void)hipFree(m->pdesc);
        m->ptab = nullptr;
        m->rinfo = nullptr;
        m->pdesc = nullptr;
        m->device_bytes -= std::min(m->device_bytes, ((size_t)m->pat_slots + 1024) * 2 + (size_t)m->M * 4 + (size_t)m->local_blocks * 8);
        m->pat_slots = 0;
    }
}

// The HLL twin of csr_tune_placement (spmv_csr.hip): a slab whose AS array is large enough for its placement to
// matter times its own kernel on a few fresh allocations of AS and keeps the fastest.
static void hll_tune_placement(spmv_hll_dev *m) {
    size_t bytes = 0;
    if (g_place_tries <= 0 || (size_t)m->slots * sizeof(double) < ((size_t)128 << 20) || !m->AS) return;
    if (hipMemPtrGetInfo(m->AS, &bytes) != hipSuccess || bytes < (size_t)m->slots * sizeof(double)) return;
    if (m->local_blocks == 0 && m->tiles) return;  // csr_tile over the slab's rows streams its own re-ordered copy
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return;
    }
    auto measure = [&](float &us) {
        for (int i = 0; i < 2; ++i)
            if (hll_launch(m, SPMV_HLL_AUTO, m->x, m->y, g_stream)) return -1;
        hipError_t e = hipEventRecord(e0, g_stream);
        for (int i = 0; i < 6 && e == hipSuccess; ++i)
            if (hll_launch(m, SPMV_HLL_AUTO, m->x, m->y, g_stream)) return -1;
        if (e == hipSuccess) e = hipEventRecord(e1, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) return -1;
        us = ms * 1e3f / 6.0f;
        return 0;
    };
    double *first = m->AS, *best = m->AS;
    float best_us = 0;
    std::vector<double *> others;
    int rc = measure(best_us);
    m->place_first_us = best_us;
    m->place_tries = 1;
    float worst_us = best_us;
    for (int t = 0; t < g_place_tries && !rc; ++t) {
        double *p = nullptr;
        if (hipMalloc((void **)&p, bytes) != hipSuccess) break;
        others.push_back(p);
        if (hipMemcpy(p, first, bytes, hipMemcpyDeviceToDevice) != hipSuccess) break;
        m->AS = p;
        float us = 0;
        rc = measure(us);
        if (rc) break;
        ++m->place_tries;
        if (us < best_us * 0.985f) {
            best = p;
            best_us = us;
        }
        worst_us = std::max(worst_us, us);
        if (best_us < worst_us * 0.915f) break;  // both halves at their fast level: nothing better to find
    }
    m->AS = best;
    m->place_best_us = best_us;
    if (best != first) (void)hipFree(first);
    for (double *p : others)
        if (p != best) (void)hipFree(p);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

// Host-only self-check of what HLL upload precomputes (flat slab offsets, workgroup windows, the
// x-window plan); needs no device.  stats (optional, 4 ints): gather windows, x-window windows
// (0 = no plan), listed lines, widest window's lines.
static int spmv_hip_hll_plan_check_body(const HLLMatrix *hll, int total_rows, int N, int *stats) {
    if (!hll || total_rows < 0 || N < 0) return fail("hll_plan_check: bad arguments");
    const int H = hll->num_blocks;
    if (H != (total_rows + kHack - 1) / kHack) return fail("hll_plan_check: %d hacks do not match %d rows", H, total_rows);
    std::vector<int> mz((size_t)H, 0);
    for (int h = 0; h < H; ++h) {
        const ELLPACKBlock *b = &hll->blocks[h];
        if (b->MAXNZ < 0 || (b->MAXNZ > 0 && (!b->JA || !b->AS))) return fail("hll_plan_check: hack %d is malformed", h);
        if (b->M != ((h == H - 1) ? total_rows - h * kHack : kHack)) return fail("hll_plan_check: hack %d has %d rows", h, b->M);
        mz[h] = b->MAXNZ;
    }
    std::vector<long long> off;
    long long true_slots = 0;
    const long long S = hll_offsets(total_rows, mz, off, true_slots);
    std::vector<int> ja((size_t)S + kPad, 0);
    for (int h = 0; h < H; ++h) {
        const size_t s = (size_t)hll->blocks[h].M * mz[h];
        for (size_t k = 0; k < s; ++k) {
            if ((unsigned)hll->blocks[h].JA[k] >= (unsigned)N) return fail("hll_plan_check: column outside [0, %d) in hack %d", N, h);
            ja[(size_t)off[h] + k] = hll->blocks[h].JA[k];
        }
    }
    auto start_of = [&](int r) { return off[r / kHack] + (long long)(r % kHack) * mz[r / kHack]; };
    std::vector<int4> plain;
    LocalPlan plan;
    hll_build_blocks(total_rows, H, off.data(), mz.data(), 2048, plain);
    int next = 0;
    for (const int4 &d : plain) {  // windows tile the rows in order
        if (d.x != next || d.y <= 0) return fail("hll_plan_check: gather window at row %d out of order", d.x);
        next = d.x + d.y;
    }
    if (next != total_rows) return fail("hll_plan_check: gather windows end at row %d of %d", next, total_rows);
    const bool have = true_slots > 0 && hll_build_local(total_rows, N, off.data(), mz.data(), ja.data(), S, 2048,
                                                        kLocalLinesMax, plain, plan);
    int widest = 0;
    if (have) {
        next = 0;
        for (size_t b = 0; b < plan.desc.size(); ++b) {
            const int4 &d = plan.desc[b];
            const int4 &ld = plan.hll_ldesc[b];
            if (d.x != next || d.y <= 0 || d.y > kStreamRowsCap) return fail("hll_plan_check: x-window window %zu out of order", b);
            next = d.x + d.y;
            const long long s0 = ((long long)d.w << 32) | (unsigned)d.z, base = s0 & ~1LL;
            const long long end = start_of(d.x + d.y - 1) + mz[(d.x + d.y - 1) / kHack];
            if (s0 != start_of(d.x) || ld.z != (int)(end - base) || ld.z > 2048) return fail("hll_plan_check: window %zu has a wrong extent", b);
            if (ld.y < 1 || ld.y > kLocalLinesMax) return fail("hll_plan_check: window %zu lists %d lines", b, ld.y);
            for (int k = 1; k < ld.y; ++k)
                if (plan.lines[ld.x + k] <= plan.lines[ld.x + k - 1]) return fail("hll_plan_check: lines of window %zu not ascending", b);
            for (long long k = s0; k < end; ++k) {
                const int slot = plan.lcol[k], rank = slot >> 4;
                if (rank >= ld.y || plan.lines[ld.x + rank] != (ja[k] >> 4) || (slot & 15) != (ja[k] & 15))
                    return fail("hll_plan_check: slot %lld (column %d) maps to %d", k, ja[k], slot);
            }
            widest = std::max(widest, ld.y);
        }
        if (next != total_rows) return fail("hll_plan_check: x-window windows end at row %d of %d", next, total_rows);
    }
    if (stats) {
        stats[0] = (int)plain.size();
        stats[1] = have ? (int)plan.desc.size() : 0;
        stats[2] = have ? (int)plan.lines.size() - kLocalLinesMax : 0;
        stats[3] = widest;
    }
    return 0;
}

extern "C" int spmv_hip_hll_plan_check(const HLLMatrix *hll, int total_rows, int N, int *stats) {
    return guarded("hll_plan_check", [&] { return spmv_hip_hll_plan_check_body(hll, total_rows, N, stats); });
}

namespace {

// x-window plan of a device-resident slab, built by plan_count / plan_fill.  Returns 1 when the handle
// now carries the plan, 0 when some window lists more than kLocalLinesMax lines (caller falls back to
// the host builder), -1 on a HIP error.
int hll_plan_on_device(spmv_hll_dev *m, int total_rows, const std::vector<long long> &off, const std::vector<int> &mz) {
    const int H = (int)mz.size();
    std::vector<int4> win;
    hll_build_blocks(total_rows, H, off.data(), mz.data(), kPlanCap, win);
    const int W = (int)win.size();
    if (W == 0) return 0;
    auto start_of = [&](int r) { return off[r / kHack] + (long long)(r % kHack) * mz[r / kHack]; };
    std::vector<long long> seg_begin((size_t)W);
    std::vector<int> seg_len((size_t)W), count((size_t)W);
    for (int w = 0; w < W; ++w) {
        const int r0 = win[w].x, r1 = r0 + win[w].y - 1;
        const long long s0 = start_of(r0), end = start_of(r1) + mz[r1 / kHack];
        if (end - (s0 & ~1LL) > kPlanCap) return 0;  // a row that alone exceeds the stage: no plan
        seg_begin[w] = s0;
        seg_len[w] = (int)(end - s0);
        count[w] = (int)(end - (s0 & ~1LL));
    }
    long long *d_begin = nullptr;
    int *d_len = nullptr, *d_n = nullptr, *d_off = nullptr;
    int result = -1;
    do {
        if (upload_array(&d_begin, seg_begin.data(), seg_begin.size(), 0)) break;
        if (upload_array(&d_len, seg_len.data(), seg_len.size(), 0)) break;
        hipError_t e = hipMalloc((void **)&d_n, (size_t)W * sizeof(int));
        if (e != hipSuccess) { fail("hll plan: hipMalloc failed: %s", hipGetErrorString(e)); break; }
        hipLaunchKernelGGL((plan_count<4>), dim3(W), dim3(kBlock), 0, g_stream, W, d_begin, d_len, m->JA, d_n);
        std::vector<int> nl((size_t)W);
        e = hipMemcpyAsync(nl.data(), d_n, (size_t)W * sizeof(int), hipMemcpyDeviceToHost, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        if (e != hipSuccess) { fail("hll plan: count pass failed: %s", hipGetErrorString(e)); break; }
        std::vector<int> line_off((size_t)W);
        std::vector<int4> ldesc((size_t)W);
        long long total = 0;
        int widest = 0;
        bool fits = true;
        for (int w = 0; w < W && fits; ++w) {
            const int n = std::max(nl[w], 1);  // a window of empty rows still stages one line
            fits = nl[w] <= kLocalLinesMax && total + n < (1LL << 31);
            line_off[w] = (int)total;
            ldesc[w] = int4{(int)total, n, count[w], 0};
            total += n;
            widest = std::max(widest, n);
        }
        if (!fits) { result = 0; break; }
        if (upload_array(&d_off, line_off.data(), line_off.size(), 0)) break;
        const size_t S = (size_t)off[H];
        e = hipMalloc((void **)&m->lines, ((size_t)total + kLocalLinesMax) * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&m->lja, (S + kPad) * sizeof(unsigned short));
        if (e == hipSuccess) e = hipMemsetAsync(m->lines, 0, ((size_t)total + kLocalLinesMax) * sizeof(int), g_stream);
        if (e == hipSuccess) e = hipMemsetAsync(m->lja, 0, (S + kPad) * sizeof(unsigned short), g_stream);
        if (e != hipSuccess) { fail("hll plan: allocation failed: %s", hipGetErrorString(e)); break; }
        hipLaunchKernelGGL((plan_fill<4>), dim3(W), dim3(kBlock), 0, g_stream, W, d_begin, d_len, m->JA, d_off,
                           m->lines, m->lja);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        if (e != hipSuccess) { fail("hll plan: fill pass failed: %s", hipGetErrorString(e)); break; }
        if (upload_array(&m->ldesc4, win.data(), win.size(), 1)) break;
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
    if (result != 1) {  // leave no half-built plan behind
        (void)hipFree(m->lines);
        (void)hipFree(m->lja);
        (void)hipFree(m->ldesc4);
        (void)hipFree(m->ldesc);
        m->lines = nullptr;
        m->lja = nullptr;
        m->ldesc4 = nullptr;
        m->ldesc = nullptr;
        m->local_blocks = 0;
    }
    return result;
}

}  // namespace

// Hacks [hack0, hack1) of the matrix, i.e. rows [32 hack0, min(32 hack1, total_rows)): one
// rank's share under the reference's hack partitioner (prepare_thread_distribution_hll,
// src/hll_matrix.c:410-540); y stays full length, the kernels write this handle's rows.
static int spmv_hip_hll_upload_part_body(const HLLMatrix *hll, int total_rows, int N, int hack0, int hack1,
                                        spmv_hll_dev **out) {
    if (need_device()) return -1;
    if (!hll || !out) return fail("hll_upload: NULL argument");
    *out = nullptr;
    if ((unsigned long long)N * 8 >= (1ull << 32))
        return fail("hll_upload: N = %d exceeds the 32-bit gather offset range of the kernels", N);
    const int Hall = hll->num_blocks;
    if (Hall != (total_rows + kHack - 1) / kHack)
        return fail("hll_upload: %d hacks do not match %d rows", Hall, total_rows);
    if (hack0 < 0 || hack1 < hack0 || hack1 > Hall)
        return fail("hll_upload: bad hack range [%d, %d) of %d", hack0, hack1, Hall);
    const int H = hack1 - hack0;
    const int row0 = hack0 * kHack;
    const int rows = std::min(hack1 * kHack, total_rows) - std::min(row0, total_rows);

    std::vector<int> mz((size_t)H, 0);
    for (int h = 0; h < H; ++h) {
        const ELLPACKBlock *b = &hll->blocks[hack0 + h];
        const int expect = (hack0 + h == Hall - 1) ? total_rows - (hack0 + h) * kHack : kHack;
        if (b->M != expect) return fail("hll_upload: hack %d holds %d rows, expected %d", hack0 + h, b->M, expect);
        if (b->MAXNZ < 0 || (b->MAXNZ > 0 && (!b->JA || !b->AS)))
            return fail("hll_upload: hack %d is malformed", hack0 + h);
        mz[h] = b->MAXNZ;
        const long long s = (long long)b->M * b->MAXNZ;
        for (long long k = 0; k < s; ++k)
            if ((unsigned)b->JA[k] >= (unsigned)N)
                return fail("hll_upload: column index %d in hack %d is outside [0, %d)", b->JA[k], hack0 + h, N);
    }
    std::vector<long long> off;
    long long true_slots = 0;
    const long long S = hll_offsets(rows, mz, off, true_slots);
    if (S > (1LL << 40)) return fail("hll_upload: %lld padded slots is unreasonable", S);

    // pack every hack into one flat pair of host arrays, then two copies
    std::vector<int> ja((size_t)S + kPad, 0);
    std::vector<double> as((size_t)S + kPad, 0.0);
    for (int h = 0; h < H; ++h) {
        const ELLPACKBlock *b = &hll->blocks[hack0 + h];
        const size_t s = (size_t)b->M * b->MAXNZ;
        if (!s) continue;
        memcpy(&ja[(size_t)off[h]], b->JA, s * sizeof(int));
        memcpy(&as[(size_t)off[h]], b->AS, s * sizeof(double));
    }
    spmv_hll_dev *m = new (std::nothrow) spmv_hll_dev();
    if (!m) return fail("hll_upload: out of host memory");
    int rc = upload_array(&m->JA, ja.data(), ja.size(), 0);
    if (!rc) rc |= upload_array(&m->AS, as.data(), as.size(), 0);
    int planned = 0;  // the x-window plan on the device where it applies, else by the host builder from `ja`
    if (!rc && g_stream_local && g_plan_on_device && true_slots > 0) {
        planned = hll_plan_on_device(m, rows, off, mz);
        if (planned < 0) rc = -1;
    }
    if (!rc) rc |= hll_finish_handle(m, rows, N, off, mz, true_slots, true, planned ? nullptr : ja.data(), total_rows, row0,
                                     as.data());
    if (!rc && planned)
        m->device_bytes += (size_t)m->local_blocks * 32 + ((size_t)m->local_lines + kLocalLinesMax) * 4 +
                           ((size_t)S + kPad) * 2;
    if (rc) {
        spmv_hip_hll_free(m);
        return -1;
    }
    (void)hll_build_patterns(m);
    hll_settle(m);  // (the searches below compare launch times: the card's steady state first)
    hll_tune_patterns(m);  // (never a reason to lose the handle)
    hll_tune_placement(m);
    *out = m;
    return 0;
}

extern "C" int spmv_hip_hll_upload_part(const HLLMatrix *hll, int total_rows, int N, int hack0, int hack1,
                                        spmv_hll_dev **out) {
    return guarded("hll_upload", [&] { return spmv_hip_hll_upload_part_body(hll, total_rows, N, hack0, hack1, out); });
}

extern "C" int spmv_hip_hll_upload(const HLLMatrix *hll, int total_rows, int N, spmv_hll_dev **out) {
    if (!hll) return fail("hll_upload: NULL argument");
    return spmv_hip_hll_upload_part(hll, total_rows, N, 0, hll->num_blocks, out);
}

// HLL built on the device from a resident CSR matrix (whole matrix, fp64).
static int spmv_hip_hll_from_csr_body(const spmv_csr_dev *csr, spmv_hll_dev **out) {
    if (need_device()) return -1;
    if (!csr || !out) return fail("hll_from_csr: NULL argument");
    *out = nullptr;
    // a row block works when it starts on a hack boundary and ends on one (or at the last row)
    if (csr->value_bytes != 8 || csr->row0 % kHack != 0 ||
        ((csr->row0 + csr->M_local) % kHack != 0 && csr->row0 + csr->M_local != csr->M_total))
        return fail("hll_from_csr: needs a whole fp64 CSR matrix (or a row block cut on hack boundaries)");
    const int M = csr->M_local, N = csr->N, H = (M + kHack - 1) / kHack;
    spmv_hll_dev *m = new (std::nothrow) spmv_hll_dev();
    if (!m) return fail("hll_from_csr: out of host memory");
    std::vector<int> mz((size_t)H, 0);
    std::vector<long long> off;
    long long true_slots = 0;
    int rc = 0;
    do {
        hipError_t e = hipMalloc((void **)&m->maxnz, ((size_t)H + 1) * sizeof(int));
        if (e != hipSuccess) { rc = fail("hipMalloc(maxnz) failed: %s", hipGetErrorString(e)); break; }
        if (H > 0) {
            hipLaunchKernelGGL(hll_hack_maxnz, dim3((H + kBlock - 1) / kBlock), dim3(kBlock), 0, g_stream, M, H,
                               csr->row_ptr, m->maxnz);
            e = hipMemcpyAsync(mz.data(), m->maxnz, (size_t)H * sizeof(int), hipMemcpyDeviceToHost, g_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
            if (e != hipSuccess) { rc = fail("hll_from_csr: maxnz pass failed: %s", hipGetErrorString(e)); break; }
        }
        const long long S = hll_offsets(M, mz, off, true_slots);  // H-sized scan on the host
        if (S > (1LL << 40)) { rc = fail("hll_from_csr: %lld padded slots is unreasonable", S); break; }
        e = hipMalloc((void **)&m->JA, ((size_t)S + kPad) * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&m->AS, ((size_t)S + kPad) * sizeof(double));
        if (e == hipSuccess) e = hipMemsetAsync(m->JA, 0, ((size_t)S + kPad) * sizeof(int), g_stream);
        if (e == hipSuccess) e = hipMemsetAsync(m->AS, 0, ((size_t)S + kPad) * sizeof(double), g_stream);
        if (e != hipSuccess) { rc = fail("hll_from_csr: slab allocation failed: %s", hipGetErrorString(e)); break; }
        rc = upload_array(&m->hack_off, off.data(), off.size(), 0);
        if (rc) break;
        if (M > 0) {
            hipLaunchKernelGGL((hll_fill_from_csr<double>), dim3((M + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock),
                               0, g_stream, M, csr->row_ptr, csr->col, (const double *)csr->val, m->hack_off,
                               m->maxnz, m->JA, m->AS);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
            if (e != hipSuccess) { rc = fail("hll_from_csr: fill failed: %s", hipGetErrorString(e)); break; }
        }
        // the x-window plan: on the device when every 2048-slot window lists at most 256 lines (then the
        // line limit would not have moved any window boundary on the host either); otherwise the host
        // builder decides (line-limited windows or no plan), from one D2H copy of JA
        int planned = 0;
        if (g_stream_local && g_plan_on_device && S > 0) {
            planned = hll_plan_on_device(m, M, off, mz);
            if (planned < 0) { rc = -1; break; }
        }
        std::vector<int> ja_host;
        std::vector<double> as_host;
        if (g_stream_local && S > 0 && !planned) {
            ja_host.resize((size_t)S);
            e = hipMemcpy(ja_host.data(), m->JA, (size_t)S * sizeof(int), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { rc = fail("hll_from_csr: JA download failed: %s", hipGetErrorString(e)); break; }
            // (the values too when the tile plan of a slab without an x-window plan is to be built on the HOST:
            // "tile_plan_on_device" 0; the device builder reads the slab where it is)
            if (!g_tile_plan_on_device && g_stream_tile != 0 &&
                (g_stream_tile == 1 || (long long)M >= kTileMinRows || S >= kTileMidEntries)) {
                as_host.resize((size_t)S);
                e = hipMemcpy(as_host.data(), m->AS, (size_t)S * sizeof(double), hipMemcpyDeviceToHost);
                if (e != hipSuccess) { rc = fail("hll_from_csr: AS download failed: %s", hipGetErrorString(e)); break; }
            }
        }
        rc = hll_finish_handle(m, M, N, off, mz, true_slots, false, ja_host.empty() ? nullptr : ja_host.data(),
                               csr->M_total, csr->row0, as_host.empty() ? nullptr : as_host.data());
        if (!rc && planned)
            m->device_bytes += (size_t)m->local_blocks * 32 + ((size_t)m->local_lines + kLocalLinesMax) * 4 +
                               ((size_t)S + kPad) * 2;
    } while (0);
    if (rc) {
        spmv_hip_hll_free(m);
        return -1;
    }
    (void)hll_build_patterns(m);
    hll_settle(m);  // (the searches below compare launch times: the card's steady state first)
    hll_tune_patterns(m);  // (never a reason to lose the handle)
    hll_tune_placement(m);
    *out = m;
    return 0;
}

extern "C" int spmv_hip_hll_from_csr(const spmv_csr_dev *csr, spmv_hll_dev **out) {
    return guarded("hll_from_csr", [&] { return spmv_hip_hll_from_csr_body(csr, out); });
}

// flat slab back to the host (tests; hosts that want the HLL arrays): hack_off[hacks + 1],
// maxnz[hacks], JA / AS [hack_off[hacks]]; any pointer may be NULL
extern "C" int spmv_hip_hll_download(const spmv_hll_dev *m, long long *hack_off, int *maxnz, int *JA, double *AS) {
    if (need_device()) return -1;
    if (!m) return fail("hll_download: NULL handle");
    HIP_TRY(hipStreamSynchronize(g_stream));
    std::vector<long long> off((size_t)m->hacks + 1);
    HIP_TRY(hipMemcpy(off.data(), m->hack_off, off.size() * sizeof(long long), hipMemcpyDeviceToHost));
    if (hack_off) memcpy(hack_off, off.data(), off.size() * sizeof(long long));
    if (maxnz && m->hacks) HIP_TRY(hipMemcpy(maxnz, m->maxnz, (size_t)m->hacks * sizeof(int), hipMemcpyDeviceToHost));
    const size_t S = (size_t)off[m->hacks];
    if (JA && S) HIP_TRY(hipMemcpy(JA, m->JA, S * sizeof(int), hipMemcpyDeviceToHost));
    if (AS && S) HIP_TRY(hipMemcpy(AS, m->AS, S * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void spmv_hip_hll_free(spmv_hll_dev *m) {
    if (!m) return;
    (void)hipFree(m->hack_off);
    (void)hipFree(m->maxnz);
    (void)hipFree(m->JA);
    (void)hipFree(m->AS);
    (void)hipFree(m->hdesc);
    (void)hipFree(m->ldesc4);
    (void)hipFree(m->ldesc);
    (void)hipFree(m->lines);
    (void)hipFree(m->lja);
    (void)hipFree(m->row_seg);
    (void)hipFree(m->ptab);
    (void)hipFree(m->rinfo);
    (void)hipFree(m->pdesc);
    spmv_hip_csr_free(m->tiles);
    (void)hipFree(m->x);
    (void)hipFree(m->y);
    delete m;
}

// the digest of the tile plan over the slab's rows (tests); all zeros without one
extern "C" int spmv_hip_hll_tile_digest(const spmv_hll_dev *m, unsigned long long *out) {
    if (!m || !out) return fail("hll_tile_digest: NULL argument");
    if (!m->tiles) {
        memset(out, 0, 64 * sizeof(unsigned long long));
        return 0;
    }
    return guarded("hll_tile_digest", [&] { return csr_tile_digest(m->tiles, out); });
}

extern "C" int spmv_hip_hll_info(const spmv_hll_dev *m, spmv_dev_info *out) {
    if (!m || !out) return fail("hll_info: NULL argument");
    memset(out, 0, sizeof *out);
    out->M_local = m->M;
    out->M_total = m->M_total;
    out->row0 = m->row0;
    out->N = m->N;
    out->value_bytes = 8;
    out->auto_variant = m->auto_variant;
    out->lanes_per_row = m->lanes_per_row;
    out->stream_blocks = m->num_blocks;
    out->slots = m->slots;
    out->hacks = m->hacks;
    // S (val + 4) + 12 H + val (M + N)
    out->algo_bytes = m->slots * 12 + 12LL * m->hacks + 8LL * ((long long)m->M + m->N);
    out->device_bytes = (long long)m->device_bytes;
    out->local_blocks = m->local_blocks;
    out->place_tries = m->place_tries;
    out->place_first_us = m->place_first_us;
    out->place_best_us = m->place_best_us;
    out->val_address = (unsigned long long)(uintptr_t)m->AS;
    out->pattern_slots = m->ptab ? m->pat_slots : 0;
    out->pattern_with_us = m->pat_with_us;
    out->pattern_without_us = m->pat_without_us;
    out->stream_kernel = m->local_blocks > 0 ? 1 : m->tiles ? 2 : 0;
    if (m->tiles) {
        spmv_dev_info t;
        if (spmv_hip_csr_info(m->tiles, &t) == 0) {
            out->tile_blocks = t.tile_blocks;
            out->tile_passes = t.tile_passes;
            out->tile_entries = t.tile_entries;
            out->tile_staged_entries = t.tile_staged_entries;
            out->tile_long_rows = t.tile_long_rows;
            out->tile_long_items = t.tile_long_items;
            out->tile_long_entries = t.tile_long_entries;
            out->tile_staged_cols = t.tile_staged_cols;
            out->tile_remainder_entries = t.tile_remainder_entries;
            out->stream_bytes = t.stream_bytes + 12LL * m->hacks;
        }
    }
    out->local_stage_lines = m->local_stage_lines;
    out->local_lines = m->local_lines;
    if (m->local_blocks > 0)  // (a pattern plan: the tables and 4 bytes per row instead of 2 bytes per slot)
        out->stream_bytes = m->slots * 8 + (m->ptab ? 2 * m->pat_slots + 4LL * m->M + 8LL * m->local_blocks : 2 * m->slots) +
                            4 * m->local_lines + 32LL * m->local_blocks + 12LL * m->hacks + 8LL * ((long long)m->M + m->N);
    return 0;
}

extern "C" void *spmv_hip_hll_x_ptr(spmv_hll_dev *m) { return m ? m->x : nullptr; }
extern "C" void *spmv_hip_hll_y_ptr(spmv_hll_dev *m) { return m ? m->y : nullptr; }

extern "C" int spmv_hip_hll_set_x(spmv_hll_dev *m, const double *x_host) {
    if (need_device()) return -1;
    if (!m || !x_host) return fail("hll_set_x: NULL argument");
    HIP_TRY(hipMemcpyAsync(m->x, x_host, (size_t)m->N * 8, hipMemcpyHostToDevice, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return 0;
}

extern "C" int spmv_hip_hll_get_y(spmv_hll_dev *m, double *y_host) {
    if (need_device()) return -1;
    if (!m || !y_host) return fail("hll_get_y: NULL argument");
    HIP_TRY(hipStreamSynchronize(g_stream));
    HIP_TRY(hipMemcpy(y_host, m->y, (size_t)m->M_total * 8, hipMemcpyDeviceToHost));  // whole y, as for CSR
    return 0;
}

namespace {

template <int L>
void launch_hll_vector(const spmv_hll_dev *m, const double *x, double *y, hipStream_t s) {
    constexpr int rows = kBlock / L;
    hipLaunchKernelGGL((hll_vector<double, L>), dim3((m->M + rows - 1) / rows), dim3(kBlock), 0, s,
                       m->M, m->hack_off, m->maxnz, m->JA, m->AS, x, y);
}

}  // namespace

int hll_launch(const spmv_hll_dev *m, int variant, const double *x, double *y_full, hipStream_t s) {
    if (m->M == 0) return 0;
    double *y = y_full + m->row0;  // the kernels number this handle's rows from 0
    if (variant == SPMV_HLL_AUTO) variant = m->auto_variant;
    switch (variant) {
        case SPMV_HLL_THREAD_ROW:
            hipLaunchKernelGGL((hll_thread_row<double>), dim3((m->M + kBlock - 1) / kBlock),
                               dim3(kBlock), 0, s, m->M, m->hack_off, m->maxnz, m->JA, m->AS, x, y);
            break;
        case SPMV_HLL_SUBWAVE:
            switch (m->lanes_per_row) {
                case 2: launch_hll_vector<2>(m, x, y, s); break;
                case 4: launch_hll_vector<4>(m, x, y, s); break;
                case 8: launch_hll_vector<8>(m, x, y, s); break;
                case 16: launch_hll_vector<16>(m, x, y, s); break;
                default: launch_hll_vector<32>(m, x, y, s); break;
            }
            break;
        case SPMV_HLL_LDS: {
            if ((g_stream_kind == -1 || g_stream_kind == 5) && m->local_blocks > 0 &&
                ((uintptr_t)x & (kLineBytes - 1)) == 0) {
                const int lchunk = g_stream_xcd < 0 ? (m->local_blocks + 7) / 8 : (g_stream_xcd ? g_stream_xcd : 16);
                const int lgrid = (m->local_blocks + 8 * lchunk - 1) / (8 * lchunk) * (8 * lchunk);
                const size_t llds = std::max((size_t)2048 * sizeof(double), (size_t)m->local_stage_lines * kLineBytes);
                const bool lnt = g_local_nt < 0 ? m->slots * 10 > (128LL << 20) : g_local_nt != 0;
                // a pattern plan: the windows' slots are rebuilt in LDS (behind the stage), lja is not read
                const bool patterns = m->ptab && m->rinfo && m->pdesc && g_local_patterns != 0;
                const size_t pat_lds = llds + ((size_t)2048 + 8) * sizeof(unsigned short);
                if (patterns && lnt)
                    hipLaunchKernelGGL((hll_lds_local<double, true, 2048, true>), dim3(lgrid), dim3(kBlock), pat_lds, s,
                                       m->local_blocks, lchunk, m->ldesc4, m->ldesc, m->lines, m->row_seg,
                                       m->lja, m->AS, x, y, m->pdesc, m->rinfo, m->ptab, (int)llds);
                else if (patterns)
                    hipLaunchKernelGGL((hll_lds_local<double, false, 2048, true>), dim3(lgrid), dim3(kBlock), pat_lds, s,
                                       m->local_blocks, lchunk, m->ldesc4, m->ldesc, m->lines, m->row_seg,
                                       m->lja, m->AS, x, y, m->pdesc, m->rinfo, m->ptab, (int)llds);
                else if (lnt)
                    hipLaunchKernelGGL((hll_lds_local<double, true, 2048>), dim3(lgrid), dim3(kBlock), llds, s,
                                       m->local_blocks, lchunk, m->ldesc4, m->ldesc, m->lines, m->row_seg,
                                       m->lja, m->AS, x, y);
                else
                    hipLaunchKernelGGL((hll_lds_local<double, false, 2048>), dim3(lgrid), dim3(kBlock), llds, s,
                                       m->local_blocks, lchunk, m->ldesc4, m->ldesc, m->lines, m->row_seg,
                                       m->lja, m->AS, x, y);
                break;
            }
            // csr_tile over the slab's rows (a packed plan copies 16-byte pieces of x and has no gather code: a foreign x
            // that is not 16-byte aligned goes to hll_lds below instead of being read past its end)
            if (m->tiles && (g_stream_kind == -1 || g_stream_kind == 6) && (!m->tiles->tile_packed || ((uintptr_t)x & 15) == 0))
                return csr_launch_any(m->tiles, SPMV_CSR_STREAM, x, y_full, s);
            const size_t lds = 32 + ((size_t)m->stage_slots + 2) * sizeof(double);
#define SPMV_HLL_LDS_LAUNCH(MAXU)                                                                  \
    hipLaunchKernelGGL((hll_lds<double, true, MAXU>), dim3(m->num_blocks), dim3(kBlock), lds, s,    \
                       m->stage_slots, m->hdesc, m->hack_off, m->maxnz, m->JA, m->AS, x, y)
            const int units = m->stage_slots / kStreamUnit;
            if (units <= 2) SPMV_HLL_LDS_LAUNCH(2);
            else if (units <= 4) SPMV_HLL_LDS_LAUNCH(4);
            else if (units <= 6) SPMV_HLL_LDS_LAUNCH(6);
            else SPMV_HLL_LDS_LAUNCH(8);
#undef SPMV_HLL_LDS_LAUNCH
            break;
        }
        default:
            return fail("unknown HLL variant %d", variant);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}



extern "C" int spmv_hip_hll_run(spmv_hll_dev *m, int variant) {
    if (need_device()) return -1;
    if (!m) return fail("hll_run: NULL handle");
    return hll_launch(m, variant, m->x, m->y, g_stream);
}

extern "C" int spmv_hip_hll_run_on(spmv_hll_dev *m, int variant, const void *d_x, void *d_y, void *stream) {
    if (need_device()) return -1;
    if (!m || !d_x || !d_y) return fail("hll_run_on: NULL argument");
    return hll_launch(m, variant, (const double *)d_x, (double *)d_y, stream ? (hipStream_t)stream : g_stream);
}

extern "C" int spmv_hip_hll_time(spmv_hll_dev *m, int variant, int warmup, int iters, int zero_y,
                                 float *ms_each) {
    if (need_device()) return -1;
    if (!m) return fail("hll_time: NULL handle");
    return time_loop(
        warmup, iters, ms_each, [&] { return hll_launch(m, variant, m->x, m->y, g_stream); },
        [&]() -> int {
            if (zero_y) HIP_TRY(hipMemsetAsync(m->y, 0, (size_t)m->M_total * 8, g_stream));
            return 0;
        });
}

extern "C" int spmv_hip_hll_time_graph(spmv_hll_dev *m, int variant, int iters, int replays, float *ms_per_iter) {
    if (need_device()) return -1;
    if (!m) return fail("hll_time_graph: NULL handle");
    return graph_loop(iters, replays, ms_per_iter, [&] { return hll_launch(m, variant, m->x, m->y, g_stream); });
}

