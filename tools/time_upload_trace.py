"""Upload of the power-law matrix with SPMV_TRACE_UPLOAD=1: the library prints where the time goes (stderr)."""
import os
import sys
import time

os.environ["SPMV_TRACE_UPLOAD"] = "1"
sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

sp.hip_init(0)
n, rp, col, val = synth.powerlaw()
for rep in range(2):
    t = time.perf_counter()
    d = sp.CsrDevice(n, n, rp, col, val)
    sp.hip_sync()
    print(f"power-law upload {rep}: {time.perf_counter() - t:.3f} s", flush=True)
    d.close()
