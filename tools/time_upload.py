"""Upload cost of the CSR / HLL handles with and without the x-window plan (host-side work)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth
from sparsematrixvectormultiplication_amd.device import set_tuning

sp.hip_init(0)
for name, (M, row_ptr, col, val) in (("cant-like", synth.fem_like(synth.FEM_GRID, 1)),
                                     ("nlpkkt-like", synth.kkt_like(synth.KKT_GRID, 1))):
    for local in (0, 1, 1):
        set_tuning("stream_local", local)
        t = time.perf_counter()
        d = sp.CsrDevice(M, M, row_ptr, col, val)
        sp.hip_sync()
        dt = time.perf_counter() - t
        print(f"{name:12s} CSR upload, x-window plan {'on ' if local else 'off'}: {dt:.3f} s "
              f"(blocks {d.info()['local_blocks']})", flush=True)
        if local:
            t = time.perf_counter()
            h = sp.HllDevice.from_csr_device(d)
            sp.hip_sync()
            print(f"{name:12s} HLL from resident CSR incl. plan: {time.perf_counter() - t:.3f} s", flush=True)
            h.close()
        d.close()

# COO -> CSR: host builder + upload against the device builder (triplets shuffled, as from an unordered file)
rng = np.random.default_rng(0)
for name, (M, row_ptr, col, val) in (("nlpkkt-like", synth.kkt_like(synth.KKT_GRID, 1)),):
    rows = np.repeat(np.arange(M, dtype=np.int32), np.diff(row_ptr))
    for label, perm in (("row-sorted", None), ("shuffled", rng.permutation(len(col)))):
        I, J, V = (rows, col, val) if perm is None else (rows[perm], col[perm], val[perm])
        t = time.perf_counter()
        csr = sp.convert_in_csr(sp.PreMatrix.from_arrays(M, M, I, J, V))
        t1 = time.perf_counter()
        d = sp.CsrDevice.from_host(csr)
        sp.hip_sync()
        t2 = time.perf_counter()
        d.close()
        best = 1e9
        for _ in range(2):
            t3 = time.perf_counter()
            d = sp.CsrDevice.from_coo(M, M, I, J, V)
            sp.hip_sync()
            best = min(best, time.perf_counter() - t3)
            d.close()
        print(f"{name:12s} COO ({label}) -> device CSR: host convert_in_csr {t1 - t:.3f} s + upload {t2 - t1:.3f} s; "
              f"spmv_hip_csr_from_coo {best:.3f} s", flush=True)


# csr_tile plans: built on host threads (tile_plan.hpp) against built on the device (tile_plan_device.hpp, round 3)
import scipy.sparse as sps


def road_like(n=12_000_000, per_row=3, sigma=2000.0):
    r = np.repeat(np.arange(n, dtype=np.int64), per_row)
    c = np.clip(r + np.rint(rng.normal(0, sigma, len(r))).astype(np.int64), 0, n - 1)
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    return n, a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data


for name, make in (("power-law 2^24 fp32", lambda: synth.powerlaw()), ("road-like 12 M rows fp64", road_like)):
    n, rp_, col_, val_ = make()
    set_tuning("place_tries", 0)
    for on_device in (0, 1, 1):
        set_tuning("tile_plan_on_device", on_device)
        t = time.perf_counter()
        d = sp.CsrDevice(n, n, rp_, col_, val_)
        sp.hip_sync()
        dt = time.perf_counter() - t
        info = d.info()
        print(f"{name:26s} upload incl. csr_tile plan, built on the {'device' if on_device else 'host  '}: {dt:.3f} s "
              f"(blocks {info['tile_blocks']}, passes {info['tile_passes']}, long-row items {info['tile_long_items']}, "
              f"digest {hash(tuple(d.tile_digest())) & 0xffffffff:08x})", flush=True)
        d.close()
    set_tuning("tile_plan_on_device", 1)
    set_tuning("place_tries", 12)
    del rp_, col_, val_
