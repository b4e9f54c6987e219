#!/usr/bin/env python3
"""Experiment behind DESIGN "Next" item 3: would column stripes of <= 4 MiB of x pay on the fp32
power-law matrix (BASELINE config 5)?  Cuts the matrix into S column stripes on the host (all rows,
columns of one stripe), uploads each as an ordinary CSR handle and times the stream kernel on it;
the sum over the stripes is what a striped SpMV would cost before the y accumulation is added."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sp.hip_init(0)
t = time.perf_counter()
_, row_ptr, col, val = synth.powerlaw(n)
nnz = int(row_ptr[-1])
print(f"power-law n={n} nnz={nnz} generated in {time.perf_counter() - t:.1f} s", flush=True)
with sp.CsrDevice(n, n, row_ptr, col, val) as whole:
    whole.set_x(np.ones(n, np.float32))
    ms = whole.time(sp.CSR_STREAM, 2, 5, zero_y=False)
    print(f"whole matrix: {ms.mean():.3f} ms  long_rows={whole.info()['long_rows']}", flush=True)
rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(row_ptr))
width = (n + S - 1) // S
total = 0.0
for s in range(S):
    lo, hi = s * width, min(n, (s + 1) * width)
    keep = (col >= lo) & (col < hi)
    cnt = np.bincount(rows[keep], minlength=n)
    rp = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(cnt, out=rp[1:])
    with sp.CsrDevice(n, n, rp, col[keep], val[keep]) as d:
        d.set_x(np.ones(n, np.float32))
        ms = d.time(sp.CSR_STREAM, 2, 5, zero_y=False)
        info = d.info()
    total += float(ms.mean())
    print(f"stripe {s:2d} cols [{lo}, {hi}): nnz={int(keep.sum()):9d} long_rows={info['long_rows']:4d} "
          f"blocks={info['stream_blocks']:6d} plan={info['local_blocks']:6d}: {ms.mean() * 1e3:8.1f} us", flush=True)
print(f"sum over {S} stripes: {total:.3f} ms (+ y accumulation)")
