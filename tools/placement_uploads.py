"""Round 3: how does the headline kernel's time vary over FRESH uploads of the same matrix in one process (the way a
caller would meet the placement effect), and which half of the matrix is slow when it is?
Per upload: kernel time (4 + 20 launches) and, from one stamped launch, the time the first half of the blocks took and
the time the second half took.  Usage (GPU box): python tools/placement_uploads.py > gpurun_out/placement_uploads.txt"""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

sp.hip_init(0)
print("box_state", json.dumps(sp.box_state()), flush=True)
M, rp, col, val = synth.kkt_like(synth.KKT_GRID, 2)
x = np.ones(M)


def halves(d):
    s0, s1, disp, xcd = d.stamp_blocks(3)
    t0 = s0.min()
    n = len(s0)
    first_done = (s1[:n // 2].max() - t0) / 100.0
    total = (s1.max() - t0) / 100.0
    return first_done, total - first_done, total


held = []
for k in range(12):
    d = sp.CsrDevice(M, M, rp, col, val)
    d.set_x(x)
    us = float(d.time(sp.CSR_STREAM, 4, 20, zero_y=False).mean() * 1e3)
    h = halves(d)
    a = d.addresses()
    print(f"upload {k:2d}: {us:6.1f} us   first half {h[0]:6.1f} us, second half {h[1]:6.1f} us (stamped launch {h[2]:6.1f})   "
          f"val@{a['val']:#x} lcol@{a['lcol']:#x}", flush=True)
    if k % 3 == 2:      # every third upload stays alive: the next ones land elsewhere
        held.append(d)
    else:
        d.close()
for d in held:
    d.close()
