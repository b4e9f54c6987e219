#!/usr/bin/env python3
"""What the x-window kernel would cost without its slot stream (the 16-bit word per entry): one stamped launch with the
slot loads replaced by a constant (measurement only, y is wrong) beside stamped ordinary launches, same handle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402

sp.hip_init(0)
for name, gen in (("nlpkkt120-like 120x120x123", lambda: synth.kkt_like()),
                  ("fem-large 40x40x257x3", lambda: synth.fem_like((40, 40, 257), 1))):
    M, rp, col, val = gen()
    with sp.CsrDevice(M, M, rp, col, val) as dev:
        dev.set_x(np.ones(M))
        out = {0: [], 1000: []}
        for rnd in range(5):
            for mode in (0, 1000):
                start, end, _, _ = dev.stamp_blocks(mode + 3)
                out[mode].append((end.max() - start.min()) * 0.01)
        ms = dev.time(sp.CSR_STREAM, 3, 40, zero_y=False)
        print(f"{name}: events {ms.mean() * 1e3:.1f} us | stamped launch, first start to last end: with the slot stream "
              f"{' '.join(f'{v:.1f}' for v in out[0])} us | without {' '.join(f'{v:.1f}' for v in out[1000])} us", flush=True)
