"""Upload cost of the CSR / HLL handles with and without the x-window plan (host-side work)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth
from sparsematrixvectormultiplication_amd.device import set_tuning

sp.hip_init(0)
for name, (M, row_ptr, col, val) in (("cant-like", synth.fem_like(synth.FEM_GRID, 1)),
                                     ("nlpkkt-like", synth.kkt_like(synth.KKT_GRID, 1))):
    for local in (0, 1, 1):
        set_tuning("stream_local", local)
        t = time.perf_counter()
        d = sp.CsrDevice(M, M, row_ptr, col, val)
        sp.hip_sync()
        dt = time.perf_counter() - t
        print(f"{name:12s} CSR upload, x-window plan {'on ' if local else 'off'}: {dt:.3f} s "
              f"(blocks {d.info()['local_blocks']})", flush=True)
        if local:
            t = time.perf_counter()
            h = sp.HllDevice.from_csr_device(d)
            sp.hip_sync()
            print(f"{name:12s} HLL from resident CSR incl. plan: {time.perf_counter() - t:.3f} s", flush=True)
            h.close()
        d.close()
