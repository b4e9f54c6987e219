"""Round 3: which address bits of which array decide whether csr_stream_local runs the nlpkkt-like matrix in ~190 or
~204 us (tools/placement_probe.py showed: same box, same process, same matrix -- the time goes with the ALLOCATION).
Arrays are moved to chosen addresses with spmv_hip_csr_relocate and the kernel is re-timed.
Usage (GPU box): python tools/placement_sweep.py > gpurun_out/placement_sweep.txt
"""
import itertools
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

MB = 1 << 20
sp.hip_init(0)
print("box_state", json.dumps(sp.box_state()), flush=True)


def t(dev, iters=25):
    ms = dev.time(sp.CSR_STREAM, 5, iters, zero_y=False)
    return float(ms.mean() * 1e3), float(ms.min() * 1e3)


def show(dev, label):
    mean, mn = t(dev)
    a = dev.addresses()
    bits = " ".join(f"{k}:{(a[k] >> 21) & 0xff:02x}" for k in ("val", "lcol", "x", "y", "lines", "row_ptr"))
    print(f"{label:46s} mean {mean:7.1f} min {mn:7.1f} us   address bits 21..28  {bits}", flush=True)
    return mean


def study(name, M, rp, col, val):
    d = sp.CsrDevice(M, M, rp, col, val)
    d.set_x(np.ones(M))
    print(f"==== {name}: blocks {d.info()['local_blocks']}", flush=True)
    show(d, "as uploaded")
    BIG = 64 * MB
    for w in ("row_ptr", "lines", "ldesc4", "y", "x", "lcol", "val"):
        d.relocate(w, BIG, 0)
    show(d, "every array at a multiple of 64 MiB")
    # ---- 1. bit 21 of val / lcol / x / y
    res = {}
    for pv, pl, px, py in itertools.product((0, 1), repeat=4):
        d.relocate("val", BIG, pv * 2 * MB)
        d.relocate("lcol", BIG, pl * 2 * MB)
        d.relocate("x", BIG, px * 2 * MB)
        d.relocate("y", BIG, py * 2 * MB)
        res[(pv, pl, px, py)] = show(d, f"bit21 val {pv} lcol {pl} x {px} y {py}")
    for k, nm in enumerate(("val", "lcol", "x", "y")):
        m0 = np.mean([v for key, v in res.items() if key[k] == 0])
        m1 = np.mean([v for key, v in res.items() if key[k] == 1])
        print(f"   marginal bit 21 of {nm}: 0 -> {m0:.1f} us, 1 -> {m1:.1f} us", flush=True)
    best = min(res, key=res.get)
    print("   best", best, res[best], flush=True)
    d.relocate("lcol", BIG, best[1] * 2 * MB)
    d.relocate("x", BIG, best[2] * 2 * MB)
    d.relocate("y", BIG, best[3] * 2 * MB)
    # ---- 2. which bits of val's base matter: 64 MiB multiple + k * 2 MiB
    for k in range(0, 32):
        d.relocate("val", BIG, k * 2 * MB)
        show(d, f"val at 64 MiB * n + {2 * k} MiB")
    # ---- 3. finer than 2 MiB
    for off in (0, 4096, 65536, 256 << 10, 512 << 10, MB, MB + (512 << 10), 2 * MB, 2 * MB + 4096, 2 * MB + (512 << 10), 3 * MB):
        d.relocate("val", BIG, off)
        show(d, f"val at 64 MiB * n + {off} B")
    d.relocate("val", BIG, best[0] * 2 * MB)
    # ---- 4. the small arrays
    for w in ("lines", "row_ptr", "ldesc4"):
        for off in (0, 2 * MB, 4 * MB, 6 * MB):
            d.relocate(w, BIG, off)
            show(d, f"{w} at 64 MiB * n + {off >> 20} MiB")
        d.relocate(w, BIG, 0)
    # ---- 5. lcol
    for k in (0, 1, 2, 3, 4, 5, 8, 9, 16, 17):
        d.relocate("lcol", BIG, k * 2 * MB)
        show(d, f"lcol at 64 MiB * n + {2 * k} MiB")
    d.close()


M, rp, col, val = synth.kkt_like(synth.KKT_GRID, 2)
study("nlpkkt-like", M, rp, col, val)
del rp, col, val
if "--quick" not in sys.argv:
    M, rp, col, val = synth.fem_like((40, 40, 257), 1)
    study("fem-large", M, rp, col, val)
