#!/usr/bin/env python3
"""A/B the CSR kernels and the stream kernel's knobs in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24).  Usage: python tools/tune_csr.py [nlpkkt|cant] [rounds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402
from sparsematrixvectormultiplication_amd.device import set_tuning  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "nlpkkt"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sp.hip_init(0)
if which == "nlpkkt":
    M, row_ptr, col, val = synth.kkt_like()
elif which == "fembig":
    M, row_ptr, col, val = synth.fem_like((40, 40, 257), 1)   # 1.23 M rows, ~92 M nnz, ~75 per row
else:
    M, row_ptr, col, val = synth.fem_like()
x = np.ones(M)
devs = {}
for cap in (2048, 4096, 8192):
    set_tuning("stream_cap", cap)
    d = sp.CsrDevice(M, M, row_ptr, col, val)
    d.set_x(x)
    devs[cap] = d
set_tuning("stream_cap", 0)
for lc in (1024, 2048, 3072):
    set_tuning("local_cap", lc)
    d = sp.CsrDevice(M, M, row_ptr, col, val)
    d.set_x(x)
    devs[f"local{lc}"] = d
set_tuning("local_cap", 0)
info = devs["local2048"].info()
algo = info["algo_bytes"]
print(f"{which}: M={M} nnz={info['nz']} algo_bytes={algo} lanes_per_row={info['lanes_per_row']} "
      f"local_blocks={info['local_blocks']} (prod {info['stream_blocks']}) stage_lines={info['local_stage_lines']} "
      f"lines={info['local_lines']} stream_bytes={info['stream_bytes']}")
arms = [(f"prod cap={cap} block={blk}", dict(stream_kind=0, stream_block=blk, stream_nt=1, stream_xcd=0), cap, sp.CSR_STREAM)
        for cap, blk in ((2048, 256), (4096, 256), (4096, 512))]
arms += [("x-window (local) cap=2048", dict(stream_kind=5, local_nt=1, stream_xcd=0), "local2048", sp.CSR_STREAM),
         ("x-window (local) cap=1024", dict(stream_kind=5, local_nt=1, stream_xcd=0), "local1024", sp.CSR_STREAM),
         ("x-window (local) cap=3072", dict(stream_kind=5, local_nt=1, stream_xcd=0), "local3072", sp.CSR_STREAM),
         ("x-window (local) cap=3072 nt=0", dict(stream_kind=5, local_nt=0, stream_xcd=0), "local3072", sp.CSR_STREAM),
         ("x-window (local) cap=2048 nt=0", dict(stream_kind=5, local_nt=0, stream_xcd=0), "local2048", sp.CSR_STREAM),
         ("x-window (local) cap=1024 nt=0", dict(stream_kind=5, local_nt=0, stream_xcd=0), "local1024", sp.CSR_STREAM),
         ("x-window (local) cap=2048 xcd=-1", dict(stream_kind=5, local_nt=1, stream_xcd=-1), "local2048", sp.CSR_STREAM),
         ] + [(f"x-window (local) cap=2048 xcd={c}", dict(stream_kind=5, local_nt=1, stream_xcd=c), "local2048", sp.CSR_STREAM)
              for c in (8, 64)]
if os.environ.get("TUNE_ALL"):
    arms += [(f"walk cap={cap}", dict(stream_kind=1, stream_nt=1), cap, sp.CSR_STREAM) for cap in (2048, 4096)]
    arms += [(f"RING cap=2048 wgs/cu={w} nt={nt}", dict(stream_kind=4, stream_nt=nt, pipe_wgs_per_cu=w), 2048, sp.CSR_STREAM)
             for w in (1, 2) for nt in (1, 0)]
    arms += [(f"pipe cap={cap} wgs/cu={w} nt={nt}", dict(stream_kind=2, stream_nt=nt, pipe_wgs_per_cu=w), cap, sp.CSR_STREAM)
             for cap in (2048, 4096) for w in (4,) for nt in (1,)]
arms += [(f"PROBE cap=4096 " + name, dict(stream_kind=10 + mode), 4096, sp.CSR_STREAM)
         for mode, name in ((0, "stream only"), (3, "all (= prod)"))]
arms += [("subwave", {}, 2048, sp.CSR_SUBWAVE), ("wave_row", {}, 2048, sp.CSR_WAVE_ROW)]
res = {a[0]: [] for a in arms}
for r in range(rounds):
    for name, knobs, cap, variant in arms:
        for k, v in knobs.items():
            set_tuning(k, v)
        ms = devs[cap].time(variant, warmup=2, iters=20, zero_y=False)
        res[name].append(ms)
t0 = __import__("time").perf_counter(); sp.flush_cache(1 << 30); sp.flush_cache(1 << 30)
t1 = __import__("time").perf_counter(); sp.flush_cache(1 << 30); t2 = __import__("time").perf_counter()
print(f"read+write sweep of 1 GiB: {2 * (1 << 30) / (t2 - t1) / 1e9:.0f} GB/s (host-timed, incl. launch+sync)")
print(f"{'arm':34s} {'mean us':>9s} {'min us':>9s} {'GB/s(mean)':>11s} {'% of 8TB/s':>10s}")
for name, *_ in arms:
    ms = np.concatenate(res[name])
    print(f"{name:34s} {ms.mean()*1e3:9.1f} {ms.min()*1e3:9.1f} {algo/ms.mean()/1e6:11.1f} "
          f"{algo/ms.mean()/1e6/80:10.1f}")
