"""Round 3, HLL against CSR: is hll_lds_local slower than csr_stream_local because of the KERNEL or because of the DATA
(padding slots, row-aligned windows of equal-length rows)?  The slab's slots -- padding included -- are uploaded as a
CSR matrix (row r = the maxnz[h] slots of row r) and run through csr_stream_local: same entries, same rows, the other
kernel.  Usage (GPU box): python tools/hll_as_csr.py [kkt|big]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd import synth

sp.hip_init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "big"
M, rp, col, val = synth.kkt_like() if which == "kkt" else synth.fem_like((40, 40, 257), 1)
x = np.ones(M)
with sp.CsrDevice(M, M, rp, col, val) as dev:
    dev.set_x(x)
    with sp.HllDevice.from_csr_device(dev) as h:
        h.set_x(x)
        off, mz, ja, as_ = h.download()
        hi, ci = h.info(), dev.info()
        # the slab as CSR: every row of hack k has maxnz[k] slots, row-major inside the hack
        rows_in = np.minimum(32, M - 32 * np.arange(len(mz)))
        lens = np.repeat(mz, rows_in).astype(np.int64)
        starts = np.repeat(off[:-1], rows_in) + (np.arange(M) % 32) * lens
        idx = np.repeat(starts, lens) + (np.arange(lens.sum()) - np.repeat(np.cumsum(lens) - lens, lens))
        rp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        with sp.CsrDevice(M, M, rp2, ja[idx].astype(np.int32), as_[idx]) as fake:
            fake.set_x(x)
            fi = fake.info()
            assert np.max(np.abs(fake.spmv(x) - dev.spmv(x))) <= 1e-9 * np.max(np.abs(dev.spmv(x)))
            fake.set_x(x)
            dev.set_x(x)
            print(f"{which}: csr blocks {ci['local_blocks']} lines {ci['local_lines']}   hll windows {hi['local_blocks']} lines {hi['local_lines']} "
                  f"slots {hi['slots']}   slab-as-csr blocks {fi['local_blocks']} lines {fi['local_lines']} entries {fi['nz']}", flush=True)
            for r in range(3):
                a = dev.time(sp.CSR_STREAM, 2, 20, zero_y=False).mean() * 1e3
                b = h.time(sp.HLL_LDS, 2, 20, zero_y=False).mean() * 1e3
                c = fake.time(sp.CSR_STREAM, 2, 20, zero_y=False).mean() * 1e3
                print(f"round {r}: csr_stream_local on the CSR matrix {a:6.1f} us | hll_lds_local on the slab {b:6.1f} us | "
                      f"csr_stream_local on the slab's slots as CSR {c:6.1f} us", flush=True)
