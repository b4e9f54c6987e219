"""Round 3: does the x-window kernel's placement sensitivity go away when the arrays' VIRTUAL alignment is chosen
(HIP virtual-memory API) instead of left to hipMalloc?  Every line: one fresh allocation of the named array(s), then
5 + 25 timed launches.  Usage (GPU box): python tools/placement_vmm.py > gpurun_out/placement_vmm.txt"""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

MB, GB = 1 << 20, 1 << 30
sp.hip_init(0)
print("box_state", json.dumps(sp.box_state()), flush=True)


def show(dev, label):
    ms = dev.time(sp.CSR_STREAM, 5, 25, zero_y=False)
    a = dev.addresses()
    print(f"{label:60s} mean {ms.mean() * 1e3:7.1f} min {ms.min() * 1e3:7.1f} us   val@{a['val']:#x} lcol@{a['lcol']:#x} "
          f"x@{a['x']:#x} y@{a['y']:#x}", flush=True)
    return float(ms.mean() * 1e3)


def study(name, M, rp, col, val):
    d = sp.CsrDevice(M, M, rp, col, val)
    d.set_x(np.ones(M))
    print(f"==== {name}: blocks {d.info()['local_blocks']}", flush=True)
    show(d, "as uploaded (hipMalloc)")
    for align in (2 * MB, 4 * MB, 32 * MB, GB):
        for rep in range(3):
            d.relocate("val", align, 0, vmm=True)
            show(d, f"val: VMM, VA aligned to {align >> 20} MiB, try {rep}")
    for rep in range(4):
        for w in ("val", "lcol", "x", "y", "lines", "row_ptr", "ldesc4"):
            d.relocate(w, GB, 0, vmm=True)
        show(d, f"all arrays: VMM, VA aligned to 1 GiB, try {rep}")
    for rep in range(4):
        for w in ("val", "lcol", "x", "y", "lines", "row_ptr", "ldesc4"):
            d.relocate(w, 2 * MB, 0, vmm=True)
        show(d, f"all arrays: VMM, VA aligned to 2 MiB, try {rep}")
    for rep in range(4):
        for w in ("val", "lcol", "x", "y", "lines", "row_ptr", "ldesc4"):
            d.relocate(w, 2 * MB, 0, vmm=False)
        show(d, f"all arrays: hipMalloc again, try {rep}")
    d.close()


M, rp, col, val = synth.kkt_like(synth.KKT_GRID, 2)
study("nlpkkt-like", M, rp, col, val)
del rp, col, val
M, rp, col, val = synth.fem_like((40, 40, 257), 1)
study("fem-large", M, rp, col, val)
