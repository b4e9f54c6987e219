#!/usr/bin/env python3
"""N4 overlap: interior / boundary shares of every rank's blocks for the nlpkkt120-like matrix cut 8 ways (and
the FEM-shaped one), on one GPU: how much of a rank's product can run while the halo of x is still travelling."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402

sp.hip_init(0)
for name, gen in (("nlpkkt120-like 120x120x123", lambda: synth.kkt_like()),
                  ("fem-large 40x40x257x3", lambda: synth.fem_like((40, 40, 257), 1))):
    M, rp, col, val = gen()
    bounds = sp.partition_rows(rp, 8)
    print(f"== {name}: M={M} nnz={int(rp[-1])}, 8 nnz-balanced row blocks")
    print("| rank | rows | x-window blocks | interior blocks | boundary blocks | interior entries | boundary entries | interior share | kernel interior us | boundary us | whole us | column split: own entries | halo entries | own share | own us | halo us |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in range(8):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        with sp.CsrDevice(M, M, rp, col, val, lo, hi) as dev:
            c = dev.split_interior()
            dev.set_x(np.ones(M))
            L = sp.lib()
            import ctypes as C
            import time
            def timed(fn, n=20):
                for _ in range(3):
                    fn()
                sp.hip_sync()
                t = time.perf_counter()
                for _ in range(n):
                    fn()
                sp.hip_sync()
                return (time.perf_counter() - t) / n * 1e6
            t_in = timed(lambda: dev.run_part(0))
            t_out = timed(lambda: dev.run_part(1))
            t_all = timed(lambda: dev.run(sp.CSR_STREAM))
            share = c["interior_entries"] / max(1, c["interior_entries"] + c["boundary_entries"])
            # below block granularity (round 3): the handle split by column
            cs = dev.split_columns(lo, hi)
            t_own = timed(lambda: dev.run_split(0))
            t_halo = timed(lambda: dev.run_split(1))
            cshare = cs["own_entries"] / max(1, cs["own_entries"] + cs["halo_entries"])
            print(f"| {r} | {hi - lo} | {dev.info()['local_blocks']} | {c['interior_blocks']} | {c['boundary_blocks']} | "
                  f"{c['interior_entries']} | {c['boundary_entries']} | {share:.3f} | {t_in:.1f} | {t_out:.1f} | {t_all:.1f} | "
                  f"{cs['own_entries']} | {cs['halo_entries']} | {cshare:.3f} | {t_own:.1f} | {t_halo:.1f} |", flush=True)
