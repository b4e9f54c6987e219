#!/usr/bin/env python3
"""A/B of the csr_tile plan against the gather kernels on the matrix classes that get no x-window plan
(tools/matrix_zoo.py's road-like / wide band / uniform cases and BASELINE config 5's power-law matrix).
usage: time_tile.py [case ...]   cases: road band uniform powerlaw   env TILE_ROWS="1024,2048,4096" """
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402
from sparsematrixvectormultiplication_amd.device import set_tuning  # noqa: E402

rng = np.random.default_rng(2026)


def banded_random(n, per_row, sigma):
    r = np.repeat(np.arange(n, dtype=np.int64), per_row)
    c = np.clip(r + np.rint(rng.normal(0, sigma, len(r))).astype(np.int64), 0, n - 1)
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    return n, a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data


def uniform_random(n, per_row):
    r = np.repeat(np.arange(n, dtype=np.int64), per_row)
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, rng.integers(0, n, len(r)))), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    return n, a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data


CASES = {
    "road": lambda: banded_random(12_000_000, 3, 2000.0),
    "band": lambda: banded_random(2_000_000, 30, 20000.0),
    "uniform": lambda: uniform_random(4_000_000, 20),
    "powerlaw": lambda: synth.powerlaw(1 << 24, 1 << 20, 5),
    # the sizes of the reference's own graph matrices (roadNet-PA, webbase-1M: ~1 M rows, ~3 M entries; they fit the
    # Infinity Cache): does the auto rule (csr_tile from 2^20 rows on) still pay there?
    "road1m": lambda: banded_random(1_090_000, 3, 2000.0),
    "uniform1m": lambda: uniform_random(1_090_000, 3),
    "web1m": lambda: synth.powerlaw(1 << 20, 1 << 12, 7),
}
want = sys.argv[1:] or ["road", "band", "uniform", "powerlaw"]
for w in want:  # uniform:<rows>:<per row>, band:<rows>:<per row>:<sigma>, web:<log2 rows>:<longest row>
    f = w.split(":")
    if len(f) > 1 and f[0] == "uniform":
        CASES[w] = lambda f=f: uniform_random(int(f[1]), int(f[2]))
    elif len(f) > 1 and f[0] == "band":
        CASES[w] = lambda f=f: banded_random(int(f[1]), int(f[2]), float(f[3]))
    elif len(f) > 1 and f[0] == "web":
        CASES[w] = lambda f=f: synth.powerlaw(1 << int(f[1]), int(f[2]), 7)
tile_rows = [int(v) for v in os.environ.get("TILE_ROWS", "2048").split(",")]
dens = [int(v) for v in os.environ.get("TILE_DENSITY", "4").split(",")]
chunks = [2048]
sp.hip_init(0)
for name in want:
    t = time.perf_counter()
    M, rp, col, val = CASES[name]()
    nnz = int(rp[-1])
    if os.environ.get("TILE_F32"):
        val = val.astype(np.float32)
    x = np.ones(M, dtype=val.dtype)
    print(f"== {name}: M={M} nnz={nnz} dtype={val.dtype} built in {time.perf_counter() - t:.1f}s", flush=True)
    set_tuning("stream_tile", 0)
    with sp.CsrDevice(M, M, rp, col, val) as dev:
        info = dev.info()
        dev.set_x(x)
        y0 = dev.spmv(x, sp.CSR_STREAM)
        ms = dev.time(sp.CSR_STREAM, 3, 20, zero_y=False)
        print(f"   gather ({sp.device.CSR_STREAM_KERNELS[info['stream_kernel']]}): {ms.mean() * 1e3:8.1f} us  "
              f"{info['algo_bytes'] / ms.mean() / 1e6:7.0f} GB/s  {info['algo_bytes'] / ms.mean() / 1e6 / 80:5.1f} % of 8 TB/s",
              flush=True)
    packs = [int(v) for v in os.environ.get("TILE_PACK", "1").split(",")]
    for tr, dn, ch, dn_pack in [(a, b, c, d) for c in chunks for a in tile_rows for b in dens for d in packs]:
        if True:
            set_tuning("tile_balance", int(os.environ.get("TILE_BALANCE", "1")))
            set_tuning("tile_long", int(os.environ.get("TILE_LONG", "1")))
            set_tuning("tile_fit", int(os.environ.get("TILE_FIT", "1")))
            set_tuning("tile_streams", int(os.environ.get("TILE_STREAMS", "1")))
            set_tuning("tile_items", int(os.environ.get("TILE_ITEMS", "1008")))
            set_tuning("tile_min_pass", int(os.environ.get("TILE_MIN_PASS", "256")))
            set_tuning("tile_pack", dn_pack)
            set_tuning("tile_lmax", int(os.environ.get("TILE_LMAX", "1536")))
            set_tuning("stream_tile", 1)
            set_tuning("tile_rows", tr)
            set_tuning("tile_density", dn)
            t = time.perf_counter()
            with sp.CsrDevice(M, M, rp, col, val) as dev:
                up = time.perf_counter() - t
                info = dev.info()
                y1 = dev.spmv(x, sp.CSR_STREAM)
                err = float(np.max(np.abs(y1.astype(np.float64) - y0)) / max(np.max(np.abs(y0)), 1e-300))
                ms = dev.time(sp.CSR_STREAM, 3, 20, zero_y=False)
                for probe in [int(v) for v in os.environ.get("TILE_PROBE", "").split(",") if v]:  # EXPERIMENTAL=1 builds only
                    set_tuning("tile_probe", probe)
                    pm = dev.time(sp.CSR_STREAM, 2, 10, zero_y=False)
                    print(f"      probe {probe} (1 loads, staging, barriers only -- expanded plans: + x' written in slice order; 2 no gathers; 4 no run sums; 8 one workgroup per CU): {pm.mean() * 1e3:8.1f} us", flush=True)
                    set_tuning("tile_probe", 0)
                print(f"   tile pack={dn_pack} rows={tr:5d} density={dn:3d}: {ms.mean() * 1e3:8.1f} us  {info['algo_bytes'] / ms.mean() / 1e6:7.0f} GB/s  "
                      f"{info['algo_bytes'] / ms.mean() / 1e6 / 80:5.1f} %  blocks={info['tile_blocks']} passes={info['tile_passes']} "
                      f"staged={info['tile_staged_entries'] / max(1, info['tile_entries']):.2f} window_MB={info['tile_staged_cols'] * val.itemsize / 1e6:.0f} split_rows={info['tile_split_rows']} "
                      f"in_tiles={info['tile_entries'] / nnz:.2f} long_rows={info['tile_long_rows']} long_items={info['tile_long_items']} in_long={info['tile_long_entries'] / nnz:.2f} remainder={info['tile_remainder_entries'] / nnz:.4f} format_bytes={info['stream_bytes']} upload={up:.1f}s "
                      f"diff_vs_gather={err:.1e}", flush=True)
    set_tuning("stream_tile", -1)
    set_tuning("tile_rows", 0)
    set_tuning("tile_density", 4)
