#!/usr/bin/env python3
"""The time of one launch of the headline kernel over batches of 20 launches with idle stretches in between: the card's
transient after idling (why bench.py settles for a few dozen milliseconds before the warm-up steps)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402

sp.hip_init(0)
M, rp, col, val = synth.kkt_like()
with sp.CsrDevice(M, M, rp, col, val) as dev:
    dev.set_x(np.ones(M))
    for rep, idle in enumerate((0.0, 0.0, 0.5, 0.0, 2.0, 0.0)):
        time.sleep(idle)
        ms = dev.time(sp.CSR_AUTO, 5, 20, zero_y=False) * 1e3
        print(f"batch {rep} (after {idle:.1f} s idle, 5 warm-ups): mean {ms.mean():.1f} us | " + " ".join(f"{v:.0f}" for v in ms), flush=True)
    ms = dev.time(sp.CSR_AUTO, 5, 400, zero_y=False) * 1e3
    print("400 launches in a row, means of 20: " + " ".join(f"{ms[i:i + 20].mean():.0f}" for i in range(0, 400, 20)), flush=True)
