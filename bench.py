#!/usr/bin/env python3
"""bench.py -- SpMV throughput on MI355X, the reference's protocol, one JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload nlpkkt|cant|cant_hll|powerlaw]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one SpMV y = A x over the whole matrix (x = 1, the reference's input,
src/utility.c:18-22; inputs resident in HBM before the timed region).  For N > 1
the rows are split into N contiguous nnz-balanced blocks (one process per GPU, the
reference's own greedy, src/csr_matrix.c:167-266) and a step ends with the RCCL
all-gatherv of y over xGMI, so the same total work is divided: strong scaling.

Default workload = the configuration the 70 %-of-roofline target is quoted on:
BASELINE.json configs[3] "nlpkkt120 fp64 CSR" run on however many GPUs were asked
for.  SuiteSparse files are not available offline, so unless --mtx points at the
real file a seeded, shape-matched stand-in is generated (M = 3 542 400 like
nlpkkt120, ~98 M nnz, <= 28 per row; include/synth_matrix.h) and labelled as such.
configs[1]/[2] (cant CSR / HLL: 49 MB, Infinity-Cache resident, ~6 us at the
roofline, i.e. launch-bound) are reported in the same line under "also".

Timing: W untimed steps, then exactly K steps between (barrier +
torch.cuda.synchronize()) pairs, max over ranks.  Every timed launch is also
bracketed by HIP events on the library's own stream (C side); roofline.achieved
is algorithmic bytes / mean event time of the dominant kernel.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_DYNAMIC", "false")
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _host_cores():
    """Cores this process may use, read BEFORE any OpenMP runtime starts (with
    OMP_PROC_BIND the runtime pins the main thread, after which the affinity mask of
    the process reads as one core) and capped by the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


HOST_CORES = _host_cores()

import numpy as np  # noqa: E402

KERNEL_SOURCES = {  # the file a kernel's code lives in: a traffic measurement belongs to one revision of it
    "csr": os.path.join(ROOT, "sparsematrixvectormultiplication_amd", "csrc", "hip", "csr_kernels.hpp"),
    "tile": os.path.join(ROOT, "sparsematrixvectormultiplication_amd", "csrc", "hip", "tile_kernels.hpp"),
    "hll": os.path.join(ROOT, "sparsematrixvectormultiplication_amd", "csrc", "hip", "hll_kernels.hpp"),
}


def kernel_source_sha(kernel):
    """sha of the kernel's source with // comments and white space removed: an edited comment does not unstamp a
    measurement, any change of code does."""
    import hashlib
    import re
    path = KERNEL_SOURCES["hll" if kernel.startswith("hll") else "tile" if kernel == "csr_tile" else "csr"]
    try:
        text = open(path, encoding="utf-8", errors="replace").read()
    except OSError:
        return None
    code = re.sub(r"\s+", "", re.sub(r"//[^\n]*", "", text))
    return hashlib.sha256(code.encode()).hexdigest()[:16]


def measured_traffic(kernel, workload, format_bytes, blocks, table_path=None):
    """(bytes, source) -- HBM bytes per launch of `kernel` on `workload` from the committed rocprofv3 PMC
    passes (profiles/traffic.json, written by tools/prof_traffic.py: FETCH_SIZE doubled per the gfx950
    correction in MI355X_MICROARCH.md + WRITE_SIZE, separate --pmc passes), or (None, reason).  The table
    is NOT a measurement of this run: an entry is only handed out when it was taken on the same kernel
    source revision (sha of the kernels header), the same bytes of the kernel's own format and the same
    number of workgroups; anything else -- a rebuilt, retuned or re-planned kernel -- gets null."""
    try:
        table = json.load(open(table_path or os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None, "no profiles/traffic.json"
    e = table.get(f"{kernel}|{workload}")
    if not isinstance(e, dict):
        return None, "no PMC pass recorded for this kernel on this workload"
    if e.get("kernel_src_sha") != kernel_source_sha(kernel):
        return None, "kernel source changed since the PMC pass (re-run tools/prof.sh + tools/prof_traffic.py)"
    if int(e.get("format_bytes", -1)) != int(format_bytes) or int(e.get("blocks", -1)) != int(blocks):
        return None, "format bytes / workgroups differ from the profiled run"
    return int(e["bytes"]), (f"profiles/traffic.json <- {e.get('source', '?')} (rocprofv3 --pmc FETCH_SIZE x2 + "
                             "WRITE_SIZE, separate passes; not measured in this run)")


HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy rate ~6300


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=95)      # reference: 95 timed ...
    p.add_argument("--warmup", type=int, default=5)      # ... after ITERATION_SKIP = 5
    p.add_argument("--settle-ms", type=float, default=60.0,
                   help="milliseconds of untimed launches of the product's own kernel right before the warm-up steps "
                        "(0: none).  After an idle stretch -- upload, the box record -- the card goes through a transient of "
                        "a few milliseconds in which the same launch takes 178, then 203, then 185 us "
                        "(profiles/r3_launch_time_series.txt); the W warm-up steps the driver asks for are shorter than that")
    p.add_argument("--workload", default="nlpkkt", choices=["nlpkkt", "cant", "cant_hll", "powerlaw"])
    p.add_argument("--variant", default="auto")
    p.add_argument("--mtx", default=os.environ.get("SPMV_MTX"))
    p.add_argument("--grid", default=None, help="nx,ny,nz of the stand-in generator")
    p.add_argument("--powerlaw-n", type=int, default=1 << 24)
    p.add_argument("--exchange", default="rccl", choices=["rccl", "torch", "gloo-host"],
                   help="gloo-host: debug transport through host memory (several ranks may share one GPU)")
    p.add_argument("--check", action="store_true", help="every rank checks the gathered y against the oracle")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-cpu-sweep", action="store_true", help="skip the reference's thread sweep (main.c:18)")
    p.add_argument("--no-also", action="store_true", help="skip the cant CSR/HLL side measurements")
    p.add_argument("--cpu-iters", type=int, default=0, help="0 = size for ~10 s")
    p.add_argument("--dist-timeout", type=int, default=int(os.environ.get("SPMV_DIST_TIMEOUT", "300")),
                   help="seconds a rank waits in a torch.distributed collective before it gives up (N > 1)")
    p.add_argument("--fail-rank", type=int, default=-1,
                   help="TEST ONLY: this rank raises before the timed region (the job must end non-zero, not hang)")
    p.add_argument("--no-box-state", action="store_true", help="skip the sysfs / stream-probe record of the box")
    return p.parse_args()


# ----------------------------------------------------------------- workloads
def load_workload(args, sp, synth, rank, world):
    """Returns dict(M, N, row_ptr (full), col, val (this rank's rows), bounds, name, data)."""
    wl = args.workload
    if args.mtx:
        pre = sp.read_matrix_market(args.mtx)
        csr = sp.convert_in_csr(pre)
        M, N = csr.M, csr.N
        row_ptr = np.array(csr.row_ptr)
        bounds = sp.partition_rows(row_ptr, world)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        e0, e1 = row_ptr[r0], row_ptr[r1]
        out = dict(M=M, N=N, row_ptr=row_ptr, col=np.array(csr.col_idx[e0:e1]),
                   val=np.array(csr.values[e0:e1]), bounds=bounds,
                   name=os.path.basename(args.mtx), data=f"file:{os.path.basename(args.mtx)}")
        if wl == "cant_hll":
            out["hll"] = sp.convert_to_hll(pre)
        return out
    if wl == "nlpkkt":
        grid = tuple(int(v) for v in args.grid.split(",")) if args.grid else synth.KKT_GRID
        M = sp.lib().synth_kkt_rows(*grid)
        row_ptr = np.zeros(M + 1, np.int32)
        assert sp.lib().synth_kkt_row_ptr(*grid, row_ptr.ctypes.data_as(sp._native.c_int_p)) == 0
        bounds = sp.partition_rows(row_ptr, world)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        _, _, col, val = synth.kkt_like(grid, 2, r0, r1)
        return dict(M=M, N=M, row_ptr=row_ptr, col=col, val=val, bounds=bounds,
                    name=f"nlpkkt120-like stand-in {grid[0]}x{grid[1]}x{grid[2]}",
                    data="synthetic (seeded KKT-shaped stencil matrix, seed 2)")
    if wl in ("cant", "cant_hll"):
        grid = tuple(int(v) for v in args.grid.split(",")) if args.grid else synth.FEM_GRID
        M, row_ptr, col_all, val_all = synth.fem_like(grid, 1)
        bounds = sp.partition_rows(row_ptr, world)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        e0, e1 = row_ptr[r0], row_ptr[r1]
        out = dict(M=M, N=M, row_ptr=row_ptr, col=col_all[e0:e1], val=val_all[e0:e1], bounds=bounds,
                   name=f"cant-like stand-in {grid[0]}x{grid[1]}x{grid[2]}x3",
                   data="synthetic (seeded FEM-shaped stencil matrix, seed 1)")
        if wl == "cant_hll":
            from _bench_util import coo_of
            r, c = coo_of(row_ptr, col_all)
            out["hll"] = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, M, r, c, val_all))
        return out
    if wl == "powerlaw":
        n = args.powerlaw_n
        L = sp.lib()
        row_ptr = np.zeros(n + 1, np.int32)
        assert L.synth_powerlaw_row_ptr(n, 1 << 20, 5, row_ptr.ctypes.data_as(sp._native.c_int_p)) == 0
        bounds = sp.partition_rows(row_ptr, world)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        _, _, col, val = synth.powerlaw(n, 1 << 20, 5, r0, r1)
        return dict(M=n, N=n, row_ptr=row_ptr, col=col, val=val, bounds=bounds,
                    name=f"power-law {n}x{n} fp32", data="synthetic (seeded power-law, seed 5)")
    raise ValueError(wl)


# ----------------------------------------------------------------- CPU baseline
def settle(dev, variant, ms=40.0):
    """Untimed launches for `ms` milliseconds: the card's transient after an idle stretch (an upload) takes a few
    milliseconds in which one and the same launch costs 178, then 208, then 175 us (profiles/r3_launch_time_series.txt)."""
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        dev.time(variant, 0, 20, zero_y=False)


def cpu_baseline(wl, cpu_iters):
    """The reference's OpenMP CSR kernel (spvm_csr_parallel, src/csr_matrix.c:294-313) on the
    host cores of this box: the compiled reference itself when oracle/_ref travelled here
    (kind "reference"), else this repo's restatement (kind "port")."""
    import ctypes as C
    from oracle.oracle import Oracle, Reference, have_reference
    import sparsematrixvectormultiplication_amd as sp

    row_ptr, col, val = wl["row_ptr"], wl["col_full"], wl["val_full"]
    M, nnz = wl["M"], int(row_ptr[-1])
    x = np.ones(wl["N"])
    threads = min(HOST_CORES, M)
    starts, ends = sp.prepare_thread_distribution(row_ptr, threads, nnz)
    threads = len(starts)
    val64 = np.ascontiguousarray(val, dtype=np.float64)
    y = np.zeros(M)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    if have_reference():
        kind, fn = "reference", Reference().L.spvm_csr_parallel
    else:
        kind, fn = "port", Oracle().L.spvm_csr_parallel
    args = (row_ptr.ctypes.data_as(ip), col.ctypes.data_as(ip), val64.ctypes.data_as(dp),
            x.ctypes.data_as(dp), y.ctypes.data_as(dp), threads, starts.ctypes.data_as(ip),
            ends.ctypes.data_as(ip))
    t = time.perf_counter()
    fn(*args)
    first = time.perf_counter() - t
    iters = cpu_iters or int(max(5, min(95, 10.0 / max(first, 1e-4))))
    for _ in range(min(5, iters)):  # the reference's warm-up
        fn(*args)
    samples = []
    for _ in range(iters):
        t = time.perf_counter()
        fn(*args)
        samples.append(time.perf_counter() - t)
    mean = float(np.mean(samples))
    # the oracle kernel itself (K1, one core), three runs
    serial_fn = (Reference() if kind == "reference" else Oracle()).L.csr_matrix_vector_mult
    ys = np.zeros(M)
    serial = []
    for _ in range(3):
        ys[:] = 0.0
        t = time.perf_counter()
        serial_fn(M, args[0], args[1], args[2], args[3], ys.ctypes.data_as(dp))
        serial.append(time.perf_counter() - t)
    return {"value": round(2.0 * nnz / mean / 1e9, 3), "unit": "GFLOP/s", "cores": threads,
            "kind": kind, "kernel": "spvm_csr_parallel (OpenMP, nnz-balanced row ranges)",
            "ms_per_step": round(mean * 1e3, 4),
            "serial_csr_gflops_1core": round(2.0 * nnz / min(serial) / 1e9, 3),
            "sample": f"{iters} timed SpMVs over the whole {wl['name']} matrix "
                      f"({nnz} nnz) after 5 warm-ups, x = 1"}, y


REFERENCE_THREAD_SWEEP = (2, 4, 8, 16, 32, 40)  # main.c:18


def _time_calls(fn, args, warm, iters):
    for _ in range(warm):
        fn(*args)
    samples = []
    for _ in range(iters):
        t = time.perf_counter()
        fn(*args)
        samples.append(time.perf_counter() - t)
    return float(np.mean(samples))


def cpu_thread_sweep(wl, hll=None, iters=8):
    """Optional extra fields of cpu_baseline: the reference's thread sweep (main.c:18,172: T in
    {2,4,8,16,32,40}, here capped at the box's cores -- more threads than cores only measures the
    scheduler) over its OpenMP kernels K2 spvm_csr_parallel / K3 spvm_csr_parallel_simd and, when a
    host HLL is at hand, K6 spmv_hll / K7 spmv_hll_simd, each over the reference's own partition for
    that T.  GFLOP/s with the CSR nnz (M1).  2 warm-ups + `iters` timed runs per cell."""
    import ctypes as C
    from oracle.oracle import Oracle, Reference, have_reference
    import sparsematrixvectormultiplication_amd as sp

    lib = Reference().L if have_reference() else Oracle().L
    row_ptr, col, val = wl["row_ptr"], wl["col_full"], wl["val_full"]
    M, nnz = wl["M"], int(row_ptr[-1])
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    val64 = np.ascontiguousarray(val, dtype=np.float64)
    x = np.ones(wl["N"])
    y = np.zeros(max(M, (hll.num_blocks * 32) if hll is not None else 0))
    rows = []
    for T in REFERENCE_THREAD_SWEEP:
        if T > HOST_CORES or T > M:   # main.c:177 skips T > M
            continue
        cell = {"threads": T}
        starts, ends = sp.prepare_thread_distribution(row_ptr, T, nnz)
        a = (row_ptr.ctypes.data_as(ip), col.ctypes.data_as(ip), val64.ctypes.data_as(dp), x.ctypes.data_as(dp),
             y.ctypes.data_as(dp), len(starts), starts.ctypes.data_as(ip), ends.ctypes.data_as(ip))
        cell["csr"] = round(2.0 * nnz / _time_calls(lib.spvm_csr_parallel, a, 2, iters) / 1e9, 3)
        cell["csr_simd"] = round(2.0 * nnz / _time_calls(lib.spvm_csr_parallel_simd, a, 2, iters) / 1e9, 3)
        if hll is not None and T <= hll.num_blocks:
            hs, he = sp.prepare_thread_distribution_hll(hll, T)
            b = (hll.c.blocks, x.ctypes.data_as(dp), y.ctypes.data_as(dp), len(hs), hs.ctypes.data_as(ip),
                 he.ctypes.data_as(ip))
            cell["hll"] = round(2.0 * nnz / _time_calls(lib.spmv_hll, b, 2, iters) / 1e9, 3)
            cell["hll_simd"] = round(2.0 * nnz / _time_calls(lib.spmv_hll_simd, b, 2, iters) / 1e9, 3)
        rows.append(cell)
    return {"unit": "GFLOP/s", "kind": "reference" if have_reference() else "port",
            "kernels": "K2 spvm_csr_parallel, K3 spvm_csr_parallel_simd" +
                       (", K6 spmv_hll, K7 spmv_hll_simd" if hll is not None else ""),
            "threads_of_the_reference": list(REFERENCE_THREAD_SWEEP), "host_cores": HOST_CORES,
            "timed_runs_per_cell": iters, "rows": rows}


def cpu_baseline_hll(wl, cpu_iters):
    """The reference's OpenMP HLL kernel (spmv_hll, src/hll_matrix.c:376-408) over its own hack
    partition (prepare_thread_distribution_hll) on the host cores of this box; the compiled
    reference when oracle/_ref is here (kind "reference"), else the restatement (kind "port")."""
    import ctypes as C
    from oracle.oracle import Oracle, Reference, have_reference
    import sparsematrixvectormultiplication_amd as sp

    hll, M, nnz = wl["hll"], wl["M"], int(wl["row_ptr"][-1])
    x = np.ones(wl["N"])
    starts, ends = sp.prepare_thread_distribution_hll(hll, min(HOST_CORES, max(1, hll.num_blocks)))
    threads = len(starts)
    y = np.zeros(hll.num_blocks * 32)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    lib = Reference().L if have_reference() else Oracle().L
    kind = "reference" if have_reference() else "port"
    args = (hll.c.blocks, x.ctypes.data_as(dp), y.ctypes.data_as(dp), threads, starts.ctypes.data_as(ip),
            ends.ctypes.data_as(ip))
    t = time.perf_counter()
    lib.spmv_hll(*args)
    first = time.perf_counter() - t
    iters = cpu_iters or int(max(5, min(95, 10.0 / max(first, 1e-4))))
    for _ in range(min(5, iters)):
        lib.spmv_hll(*args)
    samples = []
    for _ in range(iters):
        t = time.perf_counter()
        lib.spmv_hll(*args)
        samples.append(time.perf_counter() - t)
    mean = float(np.mean(samples))
    ys = np.zeros(hll.num_blocks * 32)
    serial = []
    for _ in range(3):
        t = time.perf_counter()
        lib.spmv_hll_serial(hll.num_blocks, hll.c.blocks, x.ctypes.data_as(dp), ys.ctypes.data_as(dp))
        serial.append(time.perf_counter() - t)
    return {"value": round(2.0 * nnz / mean / 1e9, 3), "unit": "GFLOP/s", "cores": threads, "kind": kind,
            "kernel": "spmv_hll (OpenMP over the reference's hack partition)", "ms_per_step": round(mean * 1e3, 4),
            "serial_hll_gflops_1core": round(2.0 * nnz / min(serial) / 1e9, 3),
            "sample": f"{iters} timed HLL SpMVs over the whole {wl['name']} matrix ({hll.slots} slots, "
                      f"flops counted with the CSR nnz {nnz}) after 5 warm-ups, x = 1"}, y[:M]


# ----------------------------------------------------------------- side measurements
def side_measurement(sp, synth, which, steps, warmup, cpu_sweep=False):
    from sparsematrixvectormultiplication_amd.device import CSR_STREAM_KERNELS, HLL_LDS_KERNELS
    """cant-like CSR / HLL on this GPU (BASELINE configs[1], [2]) and the same FEM-shaped
    generator scaled past the Infinity Cache; kernel-only event times."""
    if which == "fem_large_csr":
        grid = (40, 40, 257)
        M, row_ptr, col, val = synth.fem_like(grid, 1)
        nnz = int(row_ptr[-1])
        with sp.CsrDevice(M, M, row_ptr, col, val) as dev:
            dev.set_x(np.ones(M))
            info = dev.info()
            settle(dev, sp.CSR_AUTO)
            ms = dev.time(sp.CSR_AUTO, warmup, steps, zero_y=True)
            # the same matrix as HLL, slab built on the GPU from the resident CSR
            with sp.HllDevice.from_csr_device(dev) as hdev:
                hdev.set_x(np.ones(M))
                hinfo = hdev.info()
                settle(hdev, sp.HLL_LDS)
                hms = hdev.time(sp.HLL_LDS, warmup, steps, zero_y=True)
            hll = {"kernel": HLL_LDS_KERNELS[hinfo["stream_kernel"]], "slots": hinfo["slots"],
                   "algo_bytes": hinfo["algo_bytes"], "format_bytes": hinfo["stream_bytes"] or hinfo["algo_bytes"],
                   "gflops": round(2.0 * nnz / (hms.mean() * 1e-3) / 1e9, 1),
                   "gbps": round(hinfo["algo_bytes"] / (hms.mean() * 1e-3) / 1e9, 1),
                   "pct_of_8TBs": round(hinfo["algo_bytes"] / (hms.mean() * 1e-3) / 1e9 / 80.0, 2),
                   "us": round(float(hms.mean()) * 1e3, 2)}
        return {"hll_hack32": hll,
                "workload": "cant-like generator scaled to %dx%dx%dx3 (M=%d, nnz=%d, %.2f GB) fp64 CSR" %
                (*grid, M, nnz, info["algo_bytes"] / 1e9), "algo_bytes": info["algo_bytes"],
                "auto": {"gflops": round(2.0 * nnz / (ms.mean() * 1e-3) / 1e9, 1),
                         "gbps": round(info["algo_bytes"] / (ms.mean() * 1e-3) / 1e9, 1),
                         "pct_of_8TBs": round(info["algo_bytes"] / (ms.mean() * 1e-3) / 1e9 / 80.0, 2),
                         "kernel": CSR_STREAM_KERNELS[info["stream_kernel"]],
                         "format_bytes": info["stream_bytes"] or info["algo_bytes"],
                         "us": round(float(ms.mean()) * 1e3, 2)}}
    if which in ("road_like", "wide_band", "dense_band"):
        # two of tools/matrix_zoo.py's classes without an x-window plan (the reference's own list is full of such
        # graph / circuit matrices, result/result_cuda.csv:2-31): what the 2-D tile kernel makes of them
        import scipy.sparse as sps
        # (dense_band: rows and entries per row of the reference's largest matrix, Cube_Coup_dt0, result_cuda.csv:3)
        n, per_row, sigma = {"road_like": (12_000_000, 3, 2000.0), "wide_band": (2_000_000, 30, 20000.0),
                             "dense_band": (2_164_760, 59, 2500.0)}[which]
        rng = np.random.default_rng(2026)
        r = np.repeat(np.arange(n, dtype=np.int64), per_row)
        c = np.clip(r + np.rint(rng.normal(0, sigma, len(r))).astype(np.int64), 0, n - 1)
        a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))
        a.sum_duplicates()
        a.sort_indices()
        from sparsematrixvectormultiplication_amd.device import CSR_STREAM_KERNELS
        with sp.CsrDevice(n, n, a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data) as dev:
            x = rng.uniform(-1, 1, n)
            y = dev.spmv(x, sp.CSR_AUTO)
            lo = n // 3
            err = float(np.max(np.abs(y[lo:lo + 50000] - (a[lo:lo + 50000] @ x))) / max(np.max(np.abs(y)), 1e-300))
            dev.set_x(np.ones(n))
            info = dev.info()
            settle(dev, sp.CSR_AUTO)
            ms = dev.time(sp.CSR_AUTO, warmup, max(5, steps // 4), zero_y=True)
        return {"workload": f"{which.replace('_', ' ')}: {per_row} per row, columns N(row, {sigma:.0f}), n = {n} fp64 CSR",
                "rows": n, "nnz": int(a.nnz), "algo_bytes": info["algo_bytes"],
                "auto": {"kernel": CSR_STREAM_KERNELS[info["stream_kernel"]], "us": round(float(ms.mean()) * 1e3, 1),
                         "gflops": round(2.0 * a.nnz / (ms.mean() * 1e-3) / 1e9, 1),
                         "gbps": round(info["algo_bytes"] / (ms.mean() * 1e-3) / 1e9, 1),
                         "pct_of_8TBs": round(info["algo_bytes"] / (ms.mean() * 1e-3) / 1e9 / 80.0, 2),
                         "format_bytes": info["stream_bytes"],
                         "staged_share": round(info["tile_staged_entries"] / max(1, info["tile_entries"]), 3),
                         "max_diff_vs_scipy_on_50000_rows_over_max_y": err}}
    if which == "powerlaw_f32":
        # BASELINE configs[4] at full size on ONE GPU (the 8-GPU run is the driver's): 2^24 rows, 2.6e8 nnz, fp32
        n, row_ptr, col, val = synth.powerlaw()
        nnz = int(row_ptr[-1])
        with sp.CsrDevice(n, n, row_ptr, col, val) as dev:
            dev.set_x(np.ones(n, dtype=np.float32))
            info = dev.info()
            settle(dev, sp.CSR_AUTO)
            ms = dev.time(sp.CSR_AUTO, warmup, max(5, steps // 4), zero_y=True)
        from sparsematrixvectormultiplication_amd.device import CSR_STREAM_KERNELS
        t = float(ms.mean()) * 1e-3
        kernel = CSR_STREAM_KERNELS[info["stream_kernel"]]
        # Two ceilings for this matrix.  By HBM bytes it sits far below 8 TB/s -- because what bounds it is not the
        # stream but the values it GATHERS (every entry of a pass that is not staged in LDS costs one request for a
        # line out of L2).  The second roofline prices exactly that: gathered values per second against what this
        # chip delivers for 64-different-lines gathers from an L2-resident table (spmv_hip_gather_probe, measured in
        # this run; tools/ubench_gather.hip is its stand-alone original).
        gathered = int(info["tile_entries"] - info["tile_staged_entries"] - info["tile_remainder_entries"]) \
            if kernel == "csr_tile" else int(info["nz"])
        traffic, traffic_source = measured_traffic(kernel, "power-law %dx%d fp32" % (n, n), info["stream_bytes"],
                                                   info.get("tile_blocks", 0) or info["stream_blocks"])
        rooflines = [{"bound": "hbm", "achieved": round(info["algo_bytes"] / t / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                      "frac": round(info["algo_bytes"] / t / 1e9 / HBM_PEAK_GBPS, 4), "traffic": traffic,
                      "traffic_source": traffic_source, "algorithmic_bytes_per_launch": int(info["algo_bytes"])}]
        expanded = int(info.get("tile_expanded_entries", 0))
        if expanded and kernel == "csr_tile":
            # (round 3) the short rows' passes no longer gather: tile_expand writes every entry's x value into its pass's
            # segment of x' and the packed kernel stages that segment -- what the product moves is its format's bytes
            gathered = 0
            rooflines.append({"bound": "hbm", "basis": "format bytes", "achieved": round(info["stream_bytes"] / t / 1e9, 1),
                              "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(info["stream_bytes"] / t / 1e9 / HBM_PEAK_GBPS, 4),
                              "format_bytes_per_launch": int(info["stream_bytes"]), "expanded_entry_slots": expanded,
                              "note": "no gather passes left (expanded plan): every tier streams entries and x slices / "
                                      "segments; format bytes = entries, descriptors, slabs, x' written and read, x and y once "
                                      "(x slices staged out of L2 not counted)"})
        try:
            if expanded and kernel == "csr_tile":
                raise StopIteration
            peak = sp.gather_probe(4, 2 << 20, 16)
            rooflines.append({"bound": "l2_gather", "achieved": round(gathered / t / 1e9, 2), "peak": round(peak / 1e9, 2),
                              "unit": "G gathered values/s", "frac": round(gathered / t / peak, 4),
                              "gathered_values_per_launch": gathered,
                              "note": "achieved = values the product gathers (entries of passes not staged in LDS) / time of the "
                                      "WHOLE product, staged passes and the long rows' launch included; peak = 64-different-"
                                      "lines gathers from a 2 MiB table, 16 wavefronts per CU x 8 in flight, measured in this run"})
        except StopIteration:
            pass
        except Exception as exc:
            rooflines.append({"bound": "l2_gather", "error": str(exc)})
        return {"workload": "power-law 2^24 x 2^24 fp32 CSR (config 5 on one GPU; no reference counterpart for fp32)",
                "rows": n, "nnz": nnz, "algo_bytes": info["algo_bytes"], "rooflines": rooflines,
                "auto": {"kernel": kernel, "ms": round(float(ms.mean()), 4),
                         "gflops": round(2.0 * nnz / (ms.mean() * 1e-3) / 1e9, 1),
                         "gbps": round(info["algo_bytes"] / (ms.mean() * 1e-3) / 1e9, 1),
                         "pct_of_8TBs": round(info["algo_bytes"] / (ms.mean() * 1e-3) / 1e9 / 80.0, 2),
                         "rows_in_split_row_kernels": info["tile_split_rows"],
                         "entries_in_tiles": info["tile_entries"], "entries_in_staged_passes": info["tile_staged_entries"],
                         "entry_slots_on_expanded_x": expanded, "entries_in_middle_tier": info.get("tile_mid_entries", 0),
                         "entries_in_long_rows_tier": info.get("tile_long_entries", 0),
                         "format_bytes": info["stream_bytes"]}}
    M, row_ptr, col, val = synth.fem_like()
    nnz = int(row_ptr[-1])
    x = np.ones(M)
    if which == "cant_csr":
        out = {}
        with sp.CsrDevice(M, M, row_ptr, col, val) as dev:
            dev.set_x(x)
            info = dev.info()
            for name, variant in (("wave_row", sp.CSR_WAVE_ROW), ("stream", sp.CSR_STREAM)):
                settle(dev, variant)
                ms = dev.time(variant, warmup, steps, zero_y=True)
                out[name] = {"gflops": round(2.0 * nnz / (ms.mean() * 1e-3) / 1e9, 1),
                             "gbps": round(info["algo_bytes"] / (ms.mean() * 1e-3) / 1e9, 1),
                             "us": round(float(ms.mean()) * 1e3, 2),
                             "us_median": round(float(np.median(ms)) * 1e3, 2)}
            # the launch-bound case: 20 launches replayed from one hipGraph, wall time per SpMV
            out["stream"]["us_per_spmv_graph_replay"] = round(dev.time_graph(sp.CSR_STREAM, 20, 10) * 1e3, 2)
        out["stream"]["kernel"] = CSR_STREAM_KERNELS[info["stream_kernel"]]
        return {"workload": "cant-like fp64 CSR (M=62451, nnz=%d; Infinity-Cache resident)" % nnz,
                "algo_bytes": info["algo_bytes"], **out}
    from _bench_util import coo_of
    r, c = coo_of(row_ptr, col)
    hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, M, r, c, val))
    with sp.HllDevice(hll) as dev:
        dev.set_x(x)
        info = dev.info()
        settle(dev, sp.HLL_LDS)
        ms = dev.time(sp.HLL_LDS, warmup, steps, zero_y=True)
    cpu = None
    if cpu_sweep:
        try:
            cpu = cpu_thread_sweep(dict(row_ptr=row_ptr, col_full=col, val_full=val, M=M, N=M), hll)
        except Exception as exc:
            cpu = {"error": str(exc)}
    return {"workload": "cant-like fp64 HLL hack=32 (slots=%d; flops counted with the CSR nnz)" %
            info["slots"], "algo_bytes": info["algo_bytes"], "cpu_thread_sweep_csr_and_hll": cpu,
            "lds": {"gflops": round(2.0 * nnz / (ms.mean() * 1e-3) / 1e9, 1),
                    "gbps": round(info["algo_bytes"] / (ms.mean() * 1e-3) / 1e9, 1),
                    "us": round(float(ms.mean()) * 1e3, 2)}}


# ----------------------------------------------------------------- launcher
def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N bench.py <same arguments>` as a CHILD process (never an exec: this process must
    stay clear of the GPU and simply waits), one rank per GPU, rendezvous on 127.0.0.1.  The ranks'
    stderr passes through; rank 0's JSON line is the only thing printed on stdout.  Returns the child's
    exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log(f"[bench] starting {n} ranks: {' '.join(cmd)}")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, cwd=ROOT, env=dict(os.environ))
    last_json = None
    for line in proc.stdout:
        if line.lstrip().startswith("{"):
            last_json = line.strip()
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if last_json is not None:
        print(last_json, flush=True)
    elif rc == 0:
        log("[bench] the ranks exited without printing a result line")
        rc = 1
    return rc


# ----------------------------------------------------------------- main
def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (before anything here has
        # touched the GPU) and relay rank 0's JSON line
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        args.gpus = world

    import torch  # first, so that one HIP runtime serves torch and libspmv_amd.so alike
    import torch.distributed as dist
    import sparsematrixvectormultiplication_amd as sp
    from sparsematrixvectormultiplication_amd import synth
    from sparsematrixvectormultiplication_amd.distributed import (NativeComm, allgatherv_rows_torch,
                                                                 local_row_ptr)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to measure)")
    # one rank per GPU; if the launcher narrowed the visible devices per rank, ordinals restart at 0
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    sp.hip_init(local_rank)
    dev_name, cus, _ = sp.device_name()
    if world > 1:
        import datetime
        # a dead or stuck rank must end the job, not stall the unattended run: every torch.distributed
        # collective below gives up after --dist-timeout seconds (gloo: raises; nccl: the watchdog aborts)
        limit = datetime.timedelta(seconds=max(30, args.dist_timeout))
        if args.exchange == "gloo-host":
            dist.init_process_group("gloo", timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=limit)

    def barrier_sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        sp.hip_sync()

    t0 = time.time()
    wl = load_workload(args, sp, synth, rank, world)
    M, N, bounds = wl["M"], wl["N"], wl["bounds"]
    nnz_total = int(wl["row_ptr"][-1])
    log(f"[rank {rank}] {wl['name']}: M={M} nnz={nnz_total} built in {time.time() - t0:.1f}s on {dev_name}")

    hll_mode = args.workload == "cant_hll"
    if hll_mode:
        # HLL splits on hack boundaries with the reference's hack partitioner
        hack_bounds = sp.partition_hacks(wl["hll"], world)
        bounds = sp.hack_bounds_to_rows(hack_bounds, M)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    if hll_mode:
        dev = sp.HllDevice(wl["hll"], int(hack_bounds[rank]), int(hack_bounds[rank + 1]))
        variant = sp.HLL_AUTO if args.variant == "auto" else sp.HLL_VARIANTS[args.variant]
        vb = 8
    else:
        dev = sp.CsrDevice(M, N, local_row_ptr(wl["row_ptr"], r0, r1), wl["col"], wl["val"], r0, r1)
        variant = sp.CSR_AUTO if args.variant == "auto" else sp.CSR_VARIANTS[args.variant]
        vb = 4 if wl["val"].dtype == np.float32 else 8
    info = dev.info()
    x = np.ones(N, dtype=np.float32 if vb == 4 else np.float64)
    dev.set_x(x)
    if args.fail_rank == rank:
        raise RuntimeError("--fail-rank: this rank fails before the timed region (test of the job's exit path)")

    # the exchange step (N > 1): RCCL all-gatherv of y
    comm, exchange, gather_mode, rccl_ranks, autotune_ms = None, "none", 0, None, None
    if world > 1:
        exchange = args.exchange
        if exchange == "rccl":
            ok = 1
            try:
                def share(ident):
                    box = [ident]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
                comm = NativeComm(rank, world, share)
                dev.step_time(bounds, variant, 0, 1)      # first exchange: fail here, not in the timed region
                # which all-gatherv is faster on THIS node (grouped broadcasts vs padded all-gather)
                rccl_ranks = comm.rccl_ranks()[1]
                gather_mode, ms_b, ms_g = comm.autotune(dev.y_ptr, bounds, vb, 10)
                autotune_ms = {"grouped_broadcasts": round(ms_b, 5),
                               "padded_allgather_scatter": None if ms_g < 0 else round(ms_g, 5)}
                log(f"[rank {rank}] all-gatherv: grouped broadcasts {ms_b * 1e3:.1f} us, padded all-gather "
                    f"{'rejected' if ms_g < 0 else f'{ms_g * 1e3:.1f} us'} -> mode {gather_mode}")
            except Exception as exc:  # both transports are RCCL; fall back to torch's
                log(f"[rank {rank}] native RCCL communicator unavailable ({exc})")
                ok = 0
            # every rank must take the same transport
            flag = torch.tensor([ok], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if comm is not None:
                    comm.close()
                    comm = None
                exchange = "torch"
                log(f"[rank {rank}] using torch.distributed for the all-gatherv")
        if exchange == "gloo-host":
            y_host = torch.zeros(M, dtype=torch.float32 if vb == 4 else torch.float64)
        if exchange == "torch":
            rccl_ranks = dist.get_world_size()  # torch's communicator (backend nccl = RCCL)
            y_t = torch.zeros(M, dtype=torch.float32 if vb == 4 else torch.float64, device="cuda")
            x_t = torch.ones(N, dtype=y_t.dtype, device="cuda")
            stream = torch.cuda.current_stream().cuda_stream

    # ---- timed region
    K, W = args.steps, args.warmup
    ms_kernel = ms_xchg = None
    settle_launches = 0
    if args.settle_ms > 0:  # (untimed, not steps: the card's own ramp after idling; every rank does the same)
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            dev.time(variant, 0, 20, zero_y=False)
            settle_launches += 20
    if world == 1:
        dev.time(variant, W, 1, zero_y=False) if W else None
        barrier_sync()
        t = time.perf_counter()
        ms_kernel = dev.time(variant, 0, K, zero_y=False)  # K launches, HIP events around each
        barrier_sync()
        wall = time.perf_counter() - t
    elif exchange == "rccl":
        dev.step_time(bounds, variant, W, 1)
        barrier_sync()
        t = time.perf_counter()
        ms_kernel, ms_xchg = dev.step_time(bounds, variant, 0, K)
        barrier_sync()
        wall = time.perf_counter() - t
    elif exchange == "gloo-host":
        import ctypes as C

        def host_step():
            t_k = dev.time(variant, 0, 1, zero_y=False)
            y_host.copy_(torch.from_numpy(dev.get_y()))
            allgatherv_rows_torch(y_host, bounds)
            sp.lib().spmv_hip_memcpy_h2d(C.c_void_p(dev.y_ptr), C.c_void_p(y_host.data_ptr()), M * vb)
            return t_k[0]
        for _ in range(W):
            host_step()
        barrier_sync()
        t = time.perf_counter()
        ms_kernel = np.array([host_step() for _ in range(K)])
        barrier_sync()
        wall = time.perf_counter() - t
        ms_xchg = np.zeros(K)
    else:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * K)]

        def step(i=None):
            if i is not None:
                ev[3 * i].record()
            dev.run_on(x_t.data_ptr(), y_t.data_ptr(), variant, stream)
            if i is not None:
                ev[3 * i + 1].record()
            allgatherv_rows_torch(y_t, bounds)
            if i is not None:
                ev[3 * i + 2].record()
        for _ in range(W):
            step()
        barrier_sync()
        t = time.perf_counter()
        for i in range(K):
            step(i)
        barrier_sync()
        wall = time.perf_counter() - t
        ms_kernel = np.array([ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(K)])
        ms_xchg = np.array([ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(K)])

    stats = torch.tensor([wall, float(np.mean(ms_kernel)), float(np.mean(ms_xchg)) if ms_xchg is not None else 0.0,
                          float(info["algo_bytes"]), float(info["slots"] if hll_mode else info["nz"]),
                          float(info.get("stream_bytes", 0))], dtype=torch.float64)
    if world > 1:
        if exchange != "gloo-host":
            stats = stats.cuda()
        gathered = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(gathered, stats)
        per_rank = torch.stack(gathered).cpu().numpy()
    else:
        per_rank = stats.numpy()[None, :]
    wall_max = float(per_rank[:, 0].max())

    # ---- parity spot check on this rank's rows (checker: the oracle)
    y_gpu = dev.get_y() if exchange != "torch" else y_t.cpu().numpy()
    if args.check:
        # every rank: the gathered y against the serial oracle on the WHOLE matrix
        from oracle.oracle import Oracle
        if args.workload == "nlpkkt" and not args.mtx:
            grid = tuple(int(v) for v in args.grid.split(",")) if args.grid else synth.KKT_GRID
            _, rp_all, col_all, val_all = synth.kkt_like(grid, 2)
        elif args.workload in ("cant", "cant_hll") and not args.mtx:
            grid = tuple(int(v) for v in args.grid.split(",")) if args.grid else synth.FEM_GRID
            _, rp_all, col_all, val_all = synth.fem_like(grid, 1)
        elif args.workload == "powerlaw" and not args.mtx:
            _, rp_all, col_all, val_all = synth.powerlaw(args.powerlaw_n, 1 << 20, 5)
        else:
            raise SystemExit("--check is implemented for the synthetic nlpkkt / cant / powerlaw workloads")
        if vb == 4:
            # fp32 has no reference counterpart: K1's loop on the fp32 data with a double accumulator, norm-wise 1e-5
            y_ref = Oracle().csr_f32_accum64(rp_all, col_all, val_all, np.ones(N, dtype=np.float32))
            gate = 1e-5
        else:
            y_ref = Oracle().csr_serial(rp_all, col_all, val_all, np.ones(N))
            gate = 1e-10
        err = float(np.max(np.abs(y_gpu - y_ref)) / max(np.max(np.abs(y_ref)), 1e-300))
        log(f"[rank {rank}] check: max|y - y_ref| / max|y_ref| = {err:.3e} over {M} rows (rows {r0}..{r1} computed here)")
        if not err <= gate:
            raise SystemExit(f"[rank {rank}] gathered y differs from the oracle: {err:.3e} (gate {gate:g})")
    result = None
    parity_failed = False
    if rank == 0:
        ms_step = wall_max / K * 1e3
        gflops = 2.0 * nnz_total / (wall_max / K) / 1e9
        algo_total = float(per_rank[:, 3].sum())
        # dominant kernel = this rank's SpMV kernel; slowest rank sets the pace
        slow = int(np.argmax(per_rank[:, 1]))
        k_ms = float(per_rank[slow, 1])
        achieved = per_rank[slow, 3] / (k_ms * 1e-3) / 1e9
        # the STREAM variant runs csr_stream_local (x lines staged in LDS, 16-bit local columns)
        # when upload found a plan for the matrix, else csr_stream (gathers)
        from sparsematrixvectormultiplication_amd.device import CSR_STREAM_KERNELS, HLL_LDS_KERNELS
        stream_name = CSR_STREAM_KERNELS[info["stream_kernel"]] if not hll_mode else None  # incl. csr_tile
        kernel_name = (HLL_LDS_KERNELS[info["stream_kernel"]] if hll_mode else
                       {0: stream_name, 1: "csr_thread_row", 2: "csr_vector<64,2>", 3: "csr_vector<L,1>",
                        4: stream_name}[variant if variant else info["auto_variant"]])
        moved = float(per_rank[slow, 5]) if per_rank[slow, 5] > 0 else float(per_rank[slow, 3])
        traffic, traffic_source = (measured_traffic(kernel_name, wl["name"], moved,
                                                    info["local_blocks"] or info.get("tile_blocks", 0) or info["stream_blocks"])
                                   if world == 1 else (None, "PMC passes are taken at N = 1 only"))
        result = {
            "metric": "SpMV GFLOP/s (2*nnz flops / step time); achieved HBM GB/s and % of 8 TB/s alongside",
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(ms_step, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if vb == 4 else "f64", "data": wl["data"],
            "hbm_gbps": round(algo_total / (wall_max / K) / 1e9, 1),
            "hbm_pct_of_8TBs": round(algo_total / (wall_max / K) / 1e9 / HBM_PEAK_GBPS * 100, 2),
            "config": {"workload": wl["name"] + (" HLL hack=32" if hll_mode else " CSR"),
                       "rows": M, "cols": N, "nnz": nnz_total, "x": "ones",
                       "workload_key": wl["name"],
                       "kernel": kernel_name,
                       "workgroups": int(info["local_blocks"] or info.get("tile_blocks", 0) or info["stream_blocks"]),
                       "parallelism": f"row-block x{world}" if world > 1 else "1 GPU",
                       "exchange": {"none": "none", "rccl": "RCCL all-gatherv(y), C-ABI communicator, " +
                                    ("one padded ncclAllGather + scatter" if gather_mode == 1 else
                                     "one ncclBroadcast per owner in a group") + " (picked by timing both)",
                                    "torch": "RCCL all-gatherv(y) via torch.distributed",
                                    "gloo-host": "DEBUG: all-gatherv(y) through host memory (gloo)"}[exchange],
                       "nnz_imbalance_max_over_mean": round(float(per_rank[:, 4].max() / per_rank[:, 4].mean()), 4),
                       "device": dev_name,
                       # where the value array lies decides which of two speeds the x-window kernel runs at (DESIGN.md,
                       # profiles/r3_placement_*.txt): what upload's placement tuning tried and kept on rank 0
                       "placement": {"val_address": hex(int(info.get("val_address", 0))),
                                     "placements_timed_at_upload": int(info.get("place_tries", 0)),
                                     "kernel_us_at_first_placement": round(float(info.get("place_first_us", 0.0)), 1),
                                     "kernel_us_at_kept_placement": round(float(info.get("place_best_us", 0.0)), 1)},
                       # the x-window kernel's pattern plan (slots rebuilt from a table per block instead of read per
                       # entry): built where the structure allows, kept where upload measured it faster on this handle
                       # untimed launches right before the W warm-up steps (--settle-ms): the card's transient after idling
                       "settle_launches_before_warmup": settle_launches,
                       "pattern_plan": {"slots_in_tables": int(info.get("pattern_slots", 0)),
                                        "kernel_us_with": round(float(info.get("pattern_with_us", 0.0)), 1),
                                        "kernel_us_without": round(float(info.get("pattern_without_us", 0.0)), 1)}},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": int(per_rank[slow, 3]),
                         # what the kernel's own format streams from HBM (2-byte local columns + line
                         # lists instead of 4-byte columns for csr_stream_local); `achieved` above is
                         # by the CSR formula of SURVEY 8(d), this is the same time priced by these bytes
                         "format_bytes_per_launch": int(moved),
                         "achieved_by_format_bytes": round(moved / (k_ms * 1e-3) / 1e9, 1),
                         "frac_by_format_bytes": round(moved / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "kernel_ms_mean": round(k_ms, 5),
                         "kernel_ms_min": round(float(np.min(ms_kernel)), 5) if slow == 0 else None},
        }
        if not args.no_box_state:
            # what distinguishes this box from the next one of the pool (the same code runs the headline kernel in
            # 180-187 or 199-204 us depending on it): HIP attributes, sysfs state of the card, a read-only stream probe
            try:
                box = sp.box_state()
                result["box"] = box
                sp_ms = box.get("stream_probe", {}).get("ms_mean")
                if sp_ms:
                    result["roofline"]["stream_probe_gbps"] = box["stream_probe"]["gbps_mean"]
                    result["roofline"]["format_rate_over_stream_probe"] = round(
                        (moved / (k_ms * 1e-3)) / (box["stream_probe"]["bytes"] / (sp_ms * 1e-3)), 4)
            except Exception as exc:  # a record, never a reason to lose the line
                result["box"] = {"error": str(exc)}
        if world > 1:
            result["per_step_ms"] = {"kernel_max_over_ranks": round(float(per_rank[:, 1].max()), 5),
                                     "allgatherv_max_over_ranks": round(float(per_rank[:, 2].max()), 5)}
            result["config"]["rccl_ranks"] = rccl_ranks          # what ncclCommCount reports (None: gloo-host)
            result["config"]["allgatherv_autotune_ms"] = autotune_ms
            result["config"]["rows_per_rank"] = [int(bounds[r + 1] - bounds[r]) for r in range(world)]

    # CPU baseline + oracle check: rank 0, N = 1 only (bounded, ~10 s)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and vb == 8:
        wl["col_full"], wl["val_full"] = wl["col"], wl["val"]
        cb, y_cpu = cpu_baseline_hll(wl, args.cpu_iters) if hll_mode else cpu_baseline(wl, args.cpu_iters)
        if not args.no_cpu_sweep:
            try:
                cb["thread_sweep"] = cpu_thread_sweep(wl, wl.get("hll"))
            except Exception as exc:  # optional fields must never lose the headline line
                cb["thread_sweep"] = {"error": str(exc)}
        result["cpu_baseline"] = cb
        scale = max(float(np.max(np.abs(y_cpu))), 1e-300)
        diff = float(np.max(np.abs(y_gpu - y_cpu)) / scale)
        result["parity_vs_cpu_reference"] = {"max_abs_diff_over_max_abs": diff, "gate": 1e-10,
                                             "parity_ok": bool(diff <= 1e-10)}
        if not diff <= 1e-10:  # a wrong-result kernel publishes no throughput
            log(f"[bench] PARITY FAILURE: GPU y differs from the CPU reference by {diff:.3e} (gate 1e-10)")
            result["value"] = None
            result["invalid"] = "parity gate failed"
            parity_failed = True
    if rank == 0 and world == 1 and not args.no_also and args.workload == "nlpkkt":
        dev.close()
        result["also"] = []
        for which in ("cant_csr", "cant_hll", "fem_large_csr", "powerlaw_f32", "road_like", "wide_band", "dense_band"):
            try:  # side numbers must never lose the headline line, nor each other
                result["also"].append(side_measurement(sp, synth, which, K, W,
                                                       cpu_sweep=which == "cant_hll" and not (args.no_cpu_baseline or args.no_cpu_sweep)))
            except Exception as exc:
                result["also"].append({"workload": which, "error": str(exc)})
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result, default=lambda o: o.item() if hasattr(o, "item") else str(o)), flush=True)
    if parity_failed:
        raise SystemExit(3)


if __name__ == "__main__":
    try:
        main()
    except BaseException as exc:  # noqa: BLE001
        code = exc.code if isinstance(exc, SystemExit) else 1
        if code in (None, 0):
            raise
        if not isinstance(exc, SystemExit):
            import traceback
            traceback.print_exc()
        elif not isinstance(code, int):
            log(code)
        sys.stdout.flush()
        sys.stderr.flush()
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            # a failed rank of a multi-rank job leaves at once: interpreter shutdown would otherwise try to tear
            # down process groups whose peers are still inside a collective; the launcher sees the non-zero code,
            # stops the other ranks and the whole job ends non-zero
            os._exit(code if isinstance(code, int) else 1)
        raise
