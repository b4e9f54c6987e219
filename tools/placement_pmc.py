"""States of one handle that differ only in where `val` lies (spmv_hip_csr_relocate), 5 + 10 launches each, so that a
rocprofv3 --pmc pass over this program shows which counters move with the 190 / 204 us modes.
Prints 'STATE k mean_us'; tools/placement_pmc_report.py joins that with the pass's per-dispatch counters."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

MB = 1 << 20
sp.hip_init(0)
M, rp, col, val = synth.kkt_like(synth.KKT_GRID, 2)
d = sp.CsrDevice(M, M, rp, col, val)
d.set_x(np.ones(M))
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    d.relocate("val", 64 * MB, k * 2 * MB)
    ms = d.time(sp.CSR_STREAM, 5, 10, zero_y=False)
    print(f"STATE {k} {ms.mean() * 1e3:.1f}", flush=True)
d.close()
