#!/usr/bin/env python3
"""What the halo exchange moves against the all-gather, from the library's own answers (needed ranges of every
rank's handle, spmv_hip_halo_plan): nlpkkt-like and fem-large split 2 / 4 / 8 ways by the reference's greedy."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth
from sparsematrixvectormultiplication_amd.distributed import halo_plan, needed_ranges

sp.hip_init(0)
for name, (M, row_ptr, col, val) in (("nlpkkt-like", synth.kkt_like()), ("fem-large", synth.fem_like((40, 40, 257), 1))):
    for G in (2, 4, 8):
        bounds = sp.partition_rows(row_ptr, G)
        needs = []
        for r in range(G):
            r0, r1 = int(bounds[r]), int(bounds[r + 1])
            e0, e1 = int(row_ptr[r0]), int(row_ptr[r1])
            rp = np.zeros(M + 1, dtype=np.int32)      # what bench.py hands over: full row_ptr shape, local entries
            rp[r0:r1 + 1] = row_ptr[r0:r1 + 1] - e0
            rp[r1 + 1:] = e1 - e0
            with sp.CsrDevice(M, M, rp, col[e0:e1], val[e0:e1], row0=r0, row1=r1) as d:
                needs.append(needed_ranges(d))
        recv = [sum(hi - lo for _, lo, hi in halo_plan(r, bounds, needs)[1]) for r in range(G)]
        peers = [len({q for q, _, _ in halo_plan(r, bounds, needs)[1]}) for r in range(G)]
        gather = [M - int(bounds[r + 1] - bounds[r]) for r in range(G)]
        print(f"{name:12s} {G} ranks: halo receives per rank {min(recv) / M:.3f}..{max(recv) / M:.3f} of x "
              f"(from {min(peers)}..{max(peers)} peers, <= {max(len(n) for n in needs)} ranges each) against "
              f"{min(gather) / M:.3f}..{max(gather) / M:.3f} for the all-gather: "
              f"{sum(gather) / max(1, sum(recv)):.1f}x less to move", flush=True)
