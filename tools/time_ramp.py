#!/usr/bin/env python3
"""Does the headline kernel time drift over a long run (clock ramp, thermal)?  Three times the bench protocol (5 + 95 launches) and
2000 launches in blocks of 200 on the nlpkkt120-like stand-in: 179.3-179.5 us throughout on the box it was run on."""
import os
import sys

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth
sp.hip_init(0)
M, rp, col, val = synth.kkt_like()
with sp.CsrDevice(M, M, rp, col, val) as dev:
    dev.set_x(np.ones(M))
    for rep in range(3):
        ms = dev.time(sp.CSR_AUTO, 5, 95, zero_y=True)
        print(f"rep {rep}: mean {ms.mean()*1e3:.1f} first10 {ms[:10].mean()*1e3:.1f} mid {ms[40:50].mean()*1e3:.1f} last10 {ms[-10:].mean()*1e3:.1f} min {ms.min()*1e3:.1f} max {ms.max()*1e3:.1f}")
    ms = dev.time(sp.CSR_AUTO, 5, 2000, zero_y=False)
    print("2000 iters:", " ".join(f"{ms[i:i+200].mean()*1e3:.1f}" for i in range(0, 2000, 200)))
