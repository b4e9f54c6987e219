#!/usr/bin/env python3
"""The nlpkkt120-like matrix through csr_stream_local and hll_lds_local, a few launches each (to be run under
rocprofv3: which counters differ between the two x-window kernels on the same matrix?)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402

sp.hip_init(0)
M, rp, col, val = synth.kkt_like() if (len(sys.argv) < 2 or sys.argv[1] == "kkt") else synth.fem_like((40, 40, 257), 1)
with sp.CsrDevice(M, M, rp, col, val) as dev:
    dev.set_x(np.ones(M))
    ms = dev.time(sp.CSR_STREAM, 2, 8, zero_y=False)
    print(f"csr {ms.mean() * 1e3:.1f} us")
    with sp.HllDevice.from_csr_device(dev) as h:
        h.set_x(np.ones(M))
        hms = h.time(sp.HLL_LDS, 2, 8, zero_y=False)
        print(f"hll {hms.mean() * 1e3:.1f} us slots={h.info()['slots']} blocks={h.info()['local_blocks']}")
