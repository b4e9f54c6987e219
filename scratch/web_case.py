import sys, numpy as np, scipy.sparse as sps
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import sparsematrixvectormultiplication_amd as sp
from sparsematrixvectormultiplication_amd.device import set_tuning
rng = np.random.default_rng(30)
def powerlaw(n, nnz, longest):
    lens = np.minimum((1.0 / rng.random(n)) ** 0.9, longest)
    lens = np.maximum(1, np.rint(lens * nnz / lens.sum())).astype(np.int64)
    r = np.repeat(np.arange(n, dtype=np.int64), lens)
    hot = np.minimum((n * rng.random(len(r)) ** 3).astype(np.int64), n - 1)
    c = np.where(rng.random(len(r)) < 0.5, hot, rng.integers(0, n, len(r)))
    a = sps.csr_matrix((rng.uniform(-1, 1, len(r)), (r, c)), shape=(n, n)); a.sum_duplicates(); a.sort_indices(); return a
a = powerlaw(1000005, 3105536, 4700)
lens = np.diff(a.indptr)
print("rows", a.shape[0], "nnz", a.nnz, "max row", lens.max(), "rows>1024", (lens > 1024).sum(), "rows>2045", (lens > 2045).sum(), "share in rows>1024", lens[lens > 1024].sum() / a.nnz)
sp.hip_init(0)
M = a.shape[0]
rp, col, val = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
x = np.ones(M)
for tag, kv in (("auto", {}), ("noskew", {"skew_rows": 0}), ("tile", {"stream_tile": 1}), ("cap1024", {"stream_cap": 1024})):
    for k, v in kv.items(): set_tuning(k, v)
    with sp.CsrDevice(M, M, rp, col, val) as dev:
        i = dev.info(); dev.set_x(x)
        ms = dev.time(sp.CSR_STREAM, 5, 50, zero_y=False)
        print(tag, sp.device.CSR_STREAM_KERNELS[i["stream_kernel"]], f"{ms.mean()*1e3:.1f} us (min {ms.min()*1e3:.1f})", {k: i[k] for k in ("num_blocks", "long_rows", "tile_blocks", "tile_long_rows", "tile_split_rows") if k in i})
    for k in kv: set_tuning(k, -1 if k == "stream_tile" else 1 if k == "skew_rows" else 0)
