#!/usr/bin/env python3
"""A/B of the HLL x-window kernel's pattern plan on ONE handle (forced at upload; "local_patterns" is read at launch)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402
from sparsematrixvectormultiplication_amd.device import set_tuning  # noqa: E402

sp.hip_init(0)
for name, gen in (("nlpkkt120-like 120x120x123", lambda: synth.kkt_like()),
                  ("fem-large 40x40x257x3", lambda: synth.fem_like((40, 40, 257), 1))):
    M, rp, col, val = gen()
    set_tuning("local_patterns", 1)
    with sp.CsrDevice(M, M, rp, col, val) as cdev, sp.HllDevice.from_csr_device(cdev) as dev:
        dev.set_x(np.ones(M))
        info = dev.info()
        rows = {0: [], 1: []}
        ys = {}
        for rnd in range(5):
            for p in (0, 1):
                set_tuning("local_patterns", p)
                ms = dev.time(sp.HLL_LDS, 3, 40, zero_y=False)
                rows[p].append(float(ms.mean()) * 1e3)
                ys[p] = dev.get_y().copy()
        same = ys[0].tobytes() == ys[1].tobytes()
        print(f"{name} HLL: slots={info['slots']} pattern slots {info['pattern_slots']} | slot stream "
              f"{' '.join(f'{v:.1f}' for v in rows[0])} (mean {np.mean(rows[0]):.1f}) | pattern plan "
              f"{' '.join(f'{v:.1f}' for v in rows[1])} (mean {np.mean(rows[1]):.1f}) us | same bits: {same}", flush=True)
    set_tuning("local_patterns", -1)
    with sp.CsrDevice(M, M, rp, col, val) as cdev, sp.HllDevice.from_csr_device(cdev) as dev:
        i = dev.info()
        print(f"   auto: pattern slots {i['pattern_slots']}, upload timed {i['pattern_with_us']:.1f} us with / {i['pattern_without_us']:.1f} without", flush=True)
