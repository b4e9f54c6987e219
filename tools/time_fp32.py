#!/usr/bin/env python3
"""The fp32 path on the banded BASELINE shapes (config 5 is fp32; its own matrix is gather-bound, so this shows
what the fp32 x-window kernel does when a plan exists): nlpkkt-like and fem-large with values cast to fp32."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth
from sparsematrixvectormultiplication_amd.device import CSR_STREAM_KERNELS

sp.hip_init(0)
for name, (M, row_ptr, col, val) in (("nlpkkt-like", synth.kkt_like()), ("fem-large", synth.fem_like((40, 40, 257), 1))):
    for dtype in (np.float64, np.float32):
        with sp.CsrDevice(M, M, row_ptr, col, val.astype(dtype)) as dev:
            dev.set_x(np.ones(M, dtype))
            info = dev.info()
            ms = dev.time(sp.CSR_AUTO, 3, 30, zero_y=False)
            y = dev.get_y().astype(np.float64)
        gb = info["algo_bytes"] / ms.mean() / 1e6
        print(f"{name:12s} {np.dtype(dtype).name}: {CSR_STREAM_KERNELS[info['stream_kernel']]:17s} {ms.mean() * 1e3:7.1f} us  "
              f"{2 * info['nz'] / ms.mean() / 1e6:7.0f} GFLOP/s  {gb:6.0f} GB/s algorithmic = {gb / 80:.1f} % of 8 TB/s "
              f"(format bytes {info['stream_bytes'] or info['algo_bytes']})  |y|_max {np.max(np.abs(y)):.6g}", flush=True)
