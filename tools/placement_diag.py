"""Round 3: where does a slow placement of csr_stream_local lose its time?
 (1) for several placements of `val` (spmv_hip_csr_relocate): the kernel time under different XCD run lengths;
 (2) for the fastest and the slowest placement seen: per-workgroup time stamps of one launch (spmv_hip_csr_stamp_blocks)
     -> per XCD: when its last block ended, mean block lifetime; and the chip's progress over time.
Usage (GPU box): python tools/placement_diag.py > gpurun_out/placement_diag.txt"""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

MB = 1 << 20
sp.hip_init(0)
print("box_state", json.dumps(sp.box_state()), flush=True)
M, rp, col, val = synth.kkt_like(synth.KKT_GRID, 2)
d = sp.CsrDevice(M, M, rp, col, val)
d.set_x(np.ones(M))


def t(iters=20):
    ms = d.time(sp.CSR_STREAM, 4, iters, zero_y=False)
    return float(ms.mean() * 1e3)


def stamps(label):
    s0, s1, disp, xcd = d.stamp_blocks(3)
    t0 = s0.min()
    total = (s1.max() - t0) / 100.0
    print(f"  [{label}] stamped launch: {total:.1f} us from the first start to the last end; block lifetime mean "
          f"{(s1 - s0).mean() / 100.0:.2f} us, p95 {np.percentile(s1 - s0, 95) / 100.0:.2f} us", flush=True)
    for k in range(8):
        sel = xcd == k
        if not sel.any():
            continue
        print(f"    XCD {k}: {int(sel.sum()):6d} blocks, dispatch ids %8 = {sorted(set((disp[sel] % 8).tolist()))}, first start "
              f"{(s0[sel].min() - t0) / 100.0:6.1f} us, last end {(s1[sel].max() - t0) / 100.0:6.1f} us, mean lifetime "
              f"{(s1[sel] - s0[sel]).mean() / 100.0:5.2f} us", flush=True)
    # progress: blocks finished per 20 us slice, whole chip
    edges = np.arange(0, total + 20, 20) * 100 + t0
    done = np.histogram(s1, bins=edges)[0]
    print("    blocks finished per 20 us:", " ".join(str(int(v)) for v in done), flush=True)
    # lifetime by position in the matrix (tenths)
    n = len(s0)
    life = (s1 - s0) / 100.0
    print("    mean lifetime by tenth of the matrix:", " ".join(f"{life[i * n // 10:(i + 1) * n // 10].mean():.2f}" for i in range(10)),
          flush=True)


print("as uploaded:", f"{t():.1f} us", flush=True)
stamps("as uploaded")
seen = []
for k in range(12):
    d.relocate("val", 64 * MB, k * 2 * MB)
    line = [f"val placement {k:2d}: default {t():6.1f}"]
    base = t()
    for xcd in (8, 13, 24, 32, 64, 100, 256):
        sp.set_tuning("stream_xcd", xcd)
        line.append(f"xcd{xcd} {t(12):6.1f}")
    sp.set_tuning("stream_xcd", 0)
    print("  ".join(line), flush=True)
    seen.append((base, k))
    if k in (0, 3, 7, 11):
        stamps(f"placement {k}, {base:.1f} us")
print("fastest / slowest placement:", min(seen), max(seen), flush=True)
d.close()
