"""Time the two ways to get an HLL matrix into HBM: host builder + upload
versus the device builder working from the resident CSR.  Prints one line per matrix."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

sp.hip_init(0)
for name, (M, row_ptr, col, val) in (("cant-like", synth.fem_like(synth.FEM_GRID, 1)),
                                     ("fem-large", synth.fem_like((40, 40, 257), 1)),
                                     ("nlpkkt-like", synth.kkt_like(synth.KKT_GRID, 1))):
    rows = np.repeat(np.arange(M, dtype=np.int32), np.diff(row_ptr))
    pre = sp.PreMatrix.from_arrays(M, M, rows, col, val)
    t0 = time.perf_counter()
    hll = sp.convert_to_hll(pre)
    t1 = time.perf_counter()
    hdev = sp.HllDevice(hll)
    sp.hip_sync()
    t2 = time.perf_counter()
    cdev = sp.CsrDevice(M, M, row_ptr, col, val)
    sp.hip_sync()
    best = 1e9
    for _ in range(3):
        t3 = time.perf_counter()
        built = sp.HllDevice.from_csr_device(cdev)
        sp.hip_sync()
        best = min(best, time.perf_counter() - t3)
        built.close()
    print(f"{name:12s} M={M} nnz={int(row_ptr[-1])} slots={hll.slots}: host convert_to_hll {t1 - t0:.3f} s "
          f"+ pack/upload {t2 - t1:.3f} s; device builder from resident CSR {best * 1e3:.2f} ms", flush=True)
    hdev.close(); cdev.close(); hll.close()
