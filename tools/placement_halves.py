"""Round 3: does a read-only stream that leans harder on HBM (16 loads of 16 bytes in flight per lane instead of 4) see
what the headline kernel sees -- that one half of a value array can be a 'fast' region and the other a 'slow' one?
Per placement of `val`: kernel time; from one stamped launch the time of the matrix's first and second half; the
deep probe's rate over the first and over the second half of val.
Usage (GPU box): python tools/placement_halves.py > gpurun_out/placement_halves.txt"""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

MB = 1 << 20
sp.hip_init(0)
sp.set_tuning("place_tries", 0)
print("box_state", json.dumps(sp.box_state()), flush=True)
M, rp, col, val = synth.kkt_like(synth.KKT_GRID, 2)
d = sp.CsrDevice(M, M, rp, col, val)
d.set_x(np.ones(M))
nz = d.info()["nz"]
half = nz // 2 * 8 // 4096 * 4096


def rate(ptr, nbytes, depth):
    sp.set_tuning("probe_depth", depth)
    mean, mn = sp.stream_probe_at(ptr, nbytes, 2, 8)
    sp.set_tuning("probe_depth", 4)
    return nbytes / (mean * 1e-3) / 1e9


for k in range(10):
    if k:
        d.relocate("val", 64 * MB, (k % 4) * 2 * MB)
    us = float(d.time(sp.CSR_STREAM, 4, 20, zero_y=False).mean() * 1e3)
    s0, s1, disp, xcd = d.stamp_blocks(3)
    t0, n = s0.min(), len(s0)
    first = (s1[:n // 2].max() - t0) / 100.0
    total = (s1.max() - t0) / 100.0
    a = d.addresses()["val"]
    print(f"placement {k}: kernel {us:6.1f} us (first half {first:5.1f}, second half {total - first:5.1f})   probe depth 4: "
          f"{rate(a, half, 4):5.0f} / {rate(a + half, half, 4):5.0f} GB/s   depth 16: {rate(a, half, 16):5.0f} / "
          f"{rate(a + half, half, 16):5.0f} GB/s (first / second half of val)   val@{a:#x}", flush=True)
d.close()
