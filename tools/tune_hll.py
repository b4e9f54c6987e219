#!/usr/bin/env python3
"""HLL kernels on the FEM-shaped generator (cant-like or scaled past the Infinity Cache)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402

from sparsematrixvectormultiplication_amd.device import set_tuning  # noqa: E402
which = sys.argv[1] if len(sys.argv) > 1 else "cant"
sp.hip_init(0)
if which == "kkt":
    M, row_ptr, col, val = synth.kkt_like()
else:
    M, row_ptr, col, val = synth.fem_like((40, 40, 257) if which == "big" else synth.FEM_GRID, 1)
rows = np.repeat(np.arange(M, dtype=np.int32), np.diff(row_ptr))
t = time.perf_counter()
hll = sp.convert_to_hll(sp.PreMatrix.from_arrays(M, M, rows, col, val))
print(f"M={M} nnz={row_ptr[-1]} convert_to_hll {time.perf_counter() - t:.2f} s")
x = np.ones(M)
with sp.HllDevice(hll) as dev, sp.CsrDevice(M, M, row_ptr, col, val) as cdev:
    dev.set_x(x)
    cdev.set_x(x)
    info = dev.info()
    print(f"slots={info['slots']} (padding {info['slots'] / row_ptr[-1] - 1:.1%}) algo_bytes={info['algo_bytes']} "
          f"windows={info['stream_blocks']} x-window plan: blocks={info['local_blocks']} stage_lines={info['local_stage_lines']} "
          f"lines={info['local_lines']} format_bytes={info['stream_bytes']}")
    y_ref = cdev.spmv(x, sp.CSR_STREAM)
    arms = [(name, v, -1) for name, v in sorted(sp.HLL_VARIANTS.items())] + [("lds (gather kernel)", sp.HLL_LDS, 0)]
    for name, v, kind in arms:
        set_tuning("stream_kind", kind)
        ms = dev.time(v, 3, 30)
        err = np.max(np.abs(dev.get_y() - y_ref)) / np.max(np.abs(y_ref))
        print(f"hll {name:20s} {ms.mean() * 1e3:8.1f} us  {info['algo_bytes'] / ms.mean() / 1e6:7.0f} GB/s on HLL bytes "
              f"({info['algo_bytes'] / ms.mean() / 1e6 / 80:.1f} % of 8 TB/s)  {2 * row_ptr[-1] / ms.mean() / 1e6:7.0f} GFLOP/s  err {err:.1e}")
    ms = cdev.time(sp.CSR_STREAM, 3, 30)
    print(f"csr stream     {ms.mean() * 1e3:8.1f} us")
