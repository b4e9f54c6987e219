#!/usr/bin/env python3
"""The nlpkkt120-like (kkt) or FEM-shaped (big) matrix through csr_stream_local and hll_lds_local in ONE process,
alternating (timing order matters on a card whose clocks move), a few launches each.  Also the thing to run under
rocprofv3 --pmc to see which counters differ between the two x-window kernels on the same matrix."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402

sp.hip_init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "kkt"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1
M, rp, col, val = synth.kkt_like() if which == "kkt" else synth.fem_like((40, 40, 257), 1)
with sp.CsrDevice(M, M, rp, col, val) as dev:
    dev.set_x(np.ones(M))
    with sp.HllDevice.from_csr_device(dev) as h:
        h.set_x(np.ones(M))
        # (round 3: where the value array lies moves either kernel by up to 8 % -- what upload's placement tuning saw)
        for name, info in (("csr", dev.info()), ("hll", h.info())):
            print(f"{name}: {info['place_tries']} placements timed at upload, first {info['place_first_us']:.1f} us, kept "
                  f"{info['place_best_us']:.1f} us, values at {info['val_address']:#x}", flush=True)
        for r in range(rounds):
            hms = h.time(sp.HLL_LDS, 2, 20, zero_y=False)
            ms = dev.time(sp.CSR_STREAM, 2, 20, zero_y=False)
            print(f"round {r}: hll {hms.mean() * 1e3:.1f} us (min {hms.min() * 1e3:.1f})   csr {ms.mean() * 1e3:.1f} us (min {ms.min() * 1e3:.1f})   "
                  f"slots={h.info()['slots']} nnz={dev.info()['nz']}", flush=True)
