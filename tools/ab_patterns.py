#!/usr/bin/env python3
"""A/B of the pattern plan of csr_stream_local on ONE handle -- same arrays, same placement of the value array:
"local_patterns" is read at launch, so the same upload runs both instantiations alternately."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402
from sparsematrixvectormultiplication_amd.device import set_tuning  # noqa: E402

import scipy.sparse as sps  # noqa: E402

rng = np.random.default_rng(2026)


def stencil(dims, offsets):
    n = int(np.prod(dims))
    idx = np.arange(n, dtype=np.int64)
    coords = np.unravel_index(idx, dims)
    rows, cols = [], []
    for off in offsets:
        ok = np.ones(n, bool)
        lin = idx.copy()
        stride = 1
        for d in range(len(dims) - 1, -1, -1):
            c = coords[d] + off[d]
            ok &= (c >= 0) & (c < dims[d])
            lin += off[d] * stride
            stride *= dims[d]
        rows.append(idx[ok])
        cols.append(lin[ok])
    r, c = np.concatenate(rows), np.concatenate(cols)
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n))
    a.sort_indices()
    return n, a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data


off5 = [(0, 0), (0, 1), (0, -1), (1, 0), (-1, 0)]
off7 = [(0, 0, 0)] + [tuple(s * (1 if k == d else 0) for k in range(3)) for d in range(3) for s in (1, -1)]
off27 = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]

sp.hip_init(0)
set_tuning("local_patterns", int(os.environ.get("PATTERNS", "1")))
cases = [("nlpkkt120-like 120x120x123", lambda: synth.kkt_like(), np.float64),
         ("fem-large 40x40x257x3", lambda: synth.fem_like((40, 40, 257), 1), np.float64),
         ("cant-like", lambda: synth.fem_like(synth.FEM_GRID, 1), np.float64),
         ("nlpkkt120-like fp32", lambda: synth.kkt_like(), np.float32),
         ("nlpkkt80-like 80x80x83", lambda: synth.kkt_like((80, 80, 83), 2), np.float64),
         ("2-D 5-point stencil 4000 x 4000", lambda: stencil((4000, 4000), off5), np.float64),
         ("3-D 7-point stencil 256^3", lambda: stencil((256, 256, 256), off7), np.float64),
         ("3-D 27-point stencil 160^3", lambda: stencil((160, 160, 160), off27), np.float64)]
want = sys.argv[1:]
if want:
    cases = [c for c in cases if any(w in c[0] for w in want)]
for name, gen, dtype in cases:
    M, rp, col, val = gen()
    val = val.astype(dtype)
    with sp.CsrDevice(M, M, rp, col, val) as dev:
        dev.set_x(np.ones(M, dtype))
        info = dev.info()
        rows = {0: [], 1: []}
        ys = {}
        for rnd in range(5):
            for p in (0, 1):
                set_tuning("local_patterns", p)
                ms = dev.time(sp.CSR_STREAM, 3, 40, zero_y=False)
                rows[p].append(float(ms.mean()) * 1e3)
                ys[p] = dev.get_y().copy()
        set_tuning("local_patterns", int(os.environ.get("PATTERNS", "1")))
        same = ys[0].tobytes() == ys[1].tobytes()
        print(f"{name}: nnz={int(rp[-1])} blocks {info['local_blocks']} stage lines {info['local_stage_lines']} pattern slots {info.get('pattern_slots', '?')} | slot stream "
              f"{' '.join(f'{v:.1f}' for v in rows[0])} (mean {np.mean(rows[0]):.1f}) | pattern plan "
              f"{' '.join(f'{v:.1f}' for v in rows[1])} (mean {np.mean(rows[1]):.1f}) us | same bits: {same}", flush=True)
