"""Round 3: is it the PHYSICAL place of an allocation that makes it fast or slow?
 (1) 1 GiB buffers allocated one after the other and kept: read-only stream rate of each -> bandwidth by allocation order;
 (2) the nlpkkt-like handle: `val` relocated 10 times; each time the stream rate over val itself, over lcol, and the kernel.
Usage (GPU box): python tools/placement_ballast.py > gpurun_out/placement_ballast.txt"""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

GB, MB = 1 << 30, 1 << 20
sp.hip_init(0)
print("box_state", json.dumps(sp.box_state(0)), flush=True)
lib = sp.lib()

# ---- (2) first, on a fresh heap
M, rp, col, val = synth.kkt_like(synth.KKT_GRID, 2)
d = sp.CsrDevice(M, M, rp, col, val)
d.set_x(np.ones(M))
nz = d.info()["nz"]


def kernel_us():
    return float(d.time(sp.CSR_STREAM, 4, 20, zero_y=False).mean() * 1e3)


def rate(ptr, nbytes):
    mean, mn = sp.stream_probe_at(ptr, nbytes // 16 * 16, 2, 8)
    return nbytes / (mean * 1e-3) / 1e9


a = d.addresses()
print(f"as uploaded: kernel {kernel_us():6.1f} us   stream over val {rate(a['val'], nz * 8):6.0f} GB/s, over lcol "
      f"{rate(a['lcol'], nz * 2):6.0f} GB/s", flush=True)
for k in range(10):
    d.relocate("val", 64 * MB, k * 2 * MB)
    a = d.addresses()
    print(f"val placement {k}: kernel {kernel_us():6.1f} us   stream over val {rate(a['val'], nz * 8):6.0f} GB/s, over lcol "
          f"{rate(a['lcol'], nz * 2):6.0f} GB/s   val@{a['val']:#x}", flush=True)
for k in range(6):
    d.relocate("lcol", 64 * MB, k * 2 * MB)
    a = d.addresses()
    print(f"lcol placement {k}: kernel {kernel_us():6.1f} us   stream over val {rate(a['val'], nz * 8):6.0f} GB/s, over lcol "
          f"{rate(a['lcol'], nz * 2):6.0f} GB/s   lcol@{a['lcol']:#x}", flush=True)
d.close()

# ---- (1) bandwidth by allocation order: 96 x 1 GiB, all kept
held = []
rates = []
for i in range(96):
    p = C.c_void_p()
    if lib.spmv_hip_malloc(C.byref(p), GB) != 0:
        break
    lib.spmv_hip_memset(p, 0, GB)
    held.append(p)
    rates.append(rate(p.value, GB))
    if i % 8 == 7:
        print(f"GiB {i - 7:3d}..{i:3d}: " + " ".join(f"{r:5.0f}" for r in rates[-8:]) + f"   @{held[i - 7].value:#x}", flush=True)
print(f"1 GiB buffers: min {min(rates):.0f} max {max(rates):.0f} mean {np.mean(rates):.0f} GB/s", flush=True)
# a second pass over the same buffers (is a buffer's rate stable?)
again = [rate(p.value, GB) for p in held[:16]]
print("first 16 again: " + " ".join(f"{r:5.0f}" for r in again), flush=True)
for p in held:
    lib.spmv_hip_free(p)
