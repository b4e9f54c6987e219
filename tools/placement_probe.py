"""Round 3, verdict item 1: why does csr_stream_local run the nlpkkt-like matrix in 180-187 us on some boxes and in
199-204 us on others, with identical code, while the FEM-shaped matrix is fast on both?

Everything that could differ between two runs of the same binary is varied inside ONE process on ONE box:
  * the box itself is recorded (sp.box_state(): HIP attributes, sysfs partition modes / clocks / power cap, a
    read-only stream probe);
  * the same matrix is uploaded several times (other virtual / physical addresses each time, earlier handles alive
    or freed, filler allocations in between);
  * the workgroup -> XCD mapping is swept (stream_xcd: runs of n blocks per XCD, -1 = one contiguous eighth);
  * the far KKT block is removed by a symmetric permutation (unknown p and its multiplier n1 + p become
    neighbours 2p, 2p + 1): same entries, same rows per block, no far x lines.
Usage (GPU box): python tools/placement_probe.py [--quick] > gpurun_out/placement.txt
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

quick = "--quick" in sys.argv
sp.hip_init(0)
print("box_state", json.dumps(sp.box_state()), flush=True)


def timed(dev, label, iters=40):
    ms = dev.time(sp.CSR_STREAM, 5, iters, zero_y=False)
    a = dev.addresses()
    print(f"{label:58s} mean {ms.mean() * 1e3:7.1f} us  min {ms.min() * 1e3:7.1f}  med {np.median(ms) * 1e3:7.1f}  "
          f"val@{a['val']:#x} lcol@{a['lcol']:#x} x@{a['x']:#x} y@{a['y']:#x}", flush=True)
    return float(ms.mean())


M, rp, col, val = synth.kkt_like(synth.KKT_GRID, 2)
x = np.ones(M)


def upload(rp_, col_, val_):
    d = sp.CsrDevice(M, M, rp_, col_, val_)
    d.set_x(x)
    return d


A = upload(rp, col, val)
info = A.info()
print(f"nlpkkt-like: blocks {info['local_blocks']} lines {info['local_lines']} algo {info['algo_bytes']} "
      f"stream {info['stream_bytes']}", flush=True)
timed(A, "A  first upload")
timed(A, "A  again")
print("stream probe 1 GiB:", sp.stream_probe(1 << 30, 2, 10), flush=True)

# ---- XCD mapping
for xcd in (1, 4, 16, 64, 256, 1024, -1, 0):
    sp.set_tuning("stream_xcd", xcd)
    timed(A, f"A  stream_xcd {xcd}")
sp.set_tuning("stream_xcd", 0)
for nt in (0, 1, -1):
    sp.set_tuning("local_nt", nt)
    timed(A, f"A  local_nt {nt}")
sp.set_tuning("local_nt", -1)

# ---- placement: more uploads of the same matrix
lib = sp.lib()
import ctypes as C
B = upload(rp, col, val)
timed(B, "B  second upload, A alive")
timed(A, "A  with B alive")
A.close()
Cd = upload(rp, col, val)
timed(Cd, "C  third upload, A freed (may reuse A's memory)")
fill = []
for size in ((2 << 20) + (64 << 10), (1 << 30) + 4096, (37 << 20) + 12288):
    p = C.c_void_p()
    assert lib.spmv_hip_malloc(C.byref(p), size) == 0
    fill.append(p)
    D = upload(rp, col, val)
    timed(D, f"D  upload behind a filler of {size} B")
    D.close()
for p in fill:
    lib.spmv_hip_free(p)
timed(B, "B  at the end")
B.close()
Cd.close()

if not quick:
    # ---- no far block: p -> 2p, n1 + p -> 2p + 1 (symmetric permutation)
    import scipy.sparse as sps
    t0 = time.time()
    n1 = M // 2
    perm = np.empty(M, dtype=np.int64)
    perm[:n1] = 2 * np.arange(n1)
    perm[n1:] = 2 * np.arange(n1) + 1
    a = sps.csr_matrix((val, col, rp), shape=(M, M))
    coo = a.tocoo()
    b = sps.csr_matrix((coo.data, (perm[coo.row], perm[coo.col])), shape=(M, M))
    b.sort_indices()
    del a, coo
    print(f"interleaved ordering built in {time.time() - t0:.1f} s, nnz {b.nnz}", flush=True)
    E = upload(b.indptr.astype(np.int32), b.indices.astype(np.int32), b.data)
    ie = E.info()
    print(f"interleaved: blocks {ie['local_blocks']} lines {ie['local_lines']} stream {ie['stream_bytes']}", flush=True)
    for xcd in (0, -1, 64):
        sp.set_tuning("stream_xcd", xcd)
        timed(E, f"E  interleaved KKT ordering, stream_xcd {xcd}")
    sp.set_tuning("stream_xcd", 0)
    E.close()
    del b
    # ---- the FEM-shaped matrix on the same box
    Mf, rpf, colf, valf = synth.fem_like((40, 40, 257), 1)
    F = sp.CsrDevice(Mf, Mf, rpf, colf, valf)
    F.set_x(np.ones(Mf))
    for xcd in (0, -1):
        sp.set_tuning("stream_xcd", xcd)
        timed(F, f"F  fem-large, stream_xcd {xcd}")
    sp.set_tuning("stream_xcd", 0)
    F.close()
print("stream probe 1 GiB:", sp.stream_probe(1 << 30, 2, 10), flush=True)
