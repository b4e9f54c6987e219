"""Row-partitioned SpMV over the GPUs of one node: one process per GPU.

The reference is single-GPU; its only partitioning logic is the OpenMP one
(prepare_thread_distribution, src/csr_matrix.c:167-266), which is reused here
unchanged as the row split: contiguous row blocks balanced by nnz.  Each rank
holds its row block (row_ptr rebased, global column indices), a full copy of
x, and writes rows [b_r, b_{r+1}) of a full-length y.  The single exchange
step is an in-place all-gatherv of y over xGMI (RCCL): one broadcast per
owner, issued as one group.  Row splitting needs no reduction.

Two transports for the exchange, both RCCL:
  "rccl"  -- the C-ABI's own communicator (spmv_hip_comm_*), on the library
             stream, so launch + exchange are enqueued back to back from C;
  "torch" -- torch.distributed (backend nccl = RCCL on ROCm; gloo on CPU for
             the world_size-2 tests), broadcasting views of one y tensor.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat
from .device import CSR_AUTO, CsrDevice, SpmvHipError, _check
from .host import partition_rows


def local_rows(bounds, rank):
    """[row0, row1) owned by `rank`."""
    return int(bounds[rank]), int(bounds[rank + 1])


def slice_csr(row_ptr, col_idx, values, row0, row1):
    """Rows [row0, row1) of a host CSR as (rebased row_ptr, col, val) views."""
    e0, e1 = int(row_ptr[row0]), int(row_ptr[row1])
    return (row_ptr[row0:row1 + 1] - row_ptr[row0]).astype(np.int32), col_idx[e0:e1], values[e0:e1]


def local_row_ptr(row_ptr_full, row0, row1):
    """A full-length (M + 1) row_ptr describing ONLY rows [row0, row1), with entry
    row_ptr_full[row0] renumbered to 0.  Handing this plus the rank's own col / val
    arrays to spmv_hip_csr_upload(..., row0, row1) uploads just that block: the C-ABI
    reads col_idx + row_ptr[row0] = the first local entry."""
    rp = np.ascontiguousarray(row_ptr_full, dtype=np.int32)
    M = len(rp) - 1
    e0 = int(rp[row0])
    return np.concatenate([np.zeros(row0, np.int32), rp[row0:row1 + 1] - e0,
                           np.full(M - row1, rp[row1] - e0, np.int32)]).astype(np.int32)


def allgatherv_rows_torch(y_full, bounds, group=None):
    """In-place all-gatherv of a full-length y tensor with torch.distributed.

    Rank r owns y_full[bounds[r]:bounds[r+1]].  Implemented as one broadcast
    per non-empty owner (RCCL/gloo have no all-gather-v); with the nccl backend
    the broadcasts are coalesced into a single group launch.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    owners = [r for r in range(world) if bounds[r + 1] > bounds[r]]
    if dist.get_backend(group) == "nccl" and hasattr(dist, "_coalescing_manager"):
        try:
            with dist._coalescing_manager(group=group, device=y_full.device, async_ops=False):
                for r in owners:
                    dist.broadcast(y_full[int(bounds[r]):int(bounds[r + 1])], src=r, group=group)
            return
        except (TypeError, RuntimeError):
            pass  # older/newer signature: fall through to plain broadcasts
    for r in owners:
        dist.broadcast(y_full[int(bounds[r]):int(bounds[r + 1])], src=r, group=group)


HALO_MAX_RANGES = 32


def needed_ranges(dev, max_ranges=HALO_MAX_RANGES):
    """[(lo, hi), ...]: the entries of x the handle's rows touch (spmv_hip_csr_needed_ranges)."""
    buf = np.zeros(2 * max_ranges, dtype=np.int32)
    n = C.c_int(0)
    _check(nat.lib().spmv_hip_csr_needed_ranges(dev.h, int(max_ranges), buf.ctypes.data_as(nat.c_int_p), C.byref(n)),
           "spmv_hip_csr_needed_ranges")
    return [(int(buf[2 * k]), int(buf[2 * k + 1])) for k in range(n.value)]


def halo_plan(rank, bounds, all_ranges, max_segments=4096):
    """Who sends what to whom (spmv_hip_halo_plan): all_ranges[p] = rank p's needed ranges.  Returns
    (send, recv), lists of (peer, lo, hi)."""
    ranks = len(all_ranges)
    stride = max(1, max(len(r) for r in all_ranges))
    counts = np.array([len(r) for r in all_ranges], dtype=np.int32)
    flat = np.zeros(ranks * 2 * stride, dtype=np.int32)
    for p, rs in enumerate(all_ranges):
        for k, (lo, hi) in enumerate(rs):
            flat[(p * stride + k) * 2], flat[(p * stride + k) * 2 + 1] = lo, hi
    b = np.ascontiguousarray(bounds, dtype=np.int32)
    send, recv = np.zeros(3 * max_segments, np.int32), np.zeros(3 * max_segments, np.int32)
    ns, nr = C.c_int(0), C.c_int(0)
    if nat.lib().spmv_hip_halo_plan(ranks, int(rank), b.ctypes.data_as(nat.c_int_p), counts.ctypes.data_as(nat.c_int_p),
                                    flat.ctypes.data_as(nat.c_int_p), stride, max_segments,
                                    send.ctypes.data_as(nat.c_int_p), C.byref(ns), recv.ctypes.data_as(nat.c_int_p),
                                    C.byref(nr)) != 0:
        raise SpmvHipError(nat.lib().spmv_hip_last_error().decode())
    trip = lambda a, n: [(int(a[3 * k]), int(a[3 * k + 1]), int(a[3 * k + 2])) for k in range(n)]
    return trip(send, ns.value), trip(recv, nr.value)


class NativeComm:
    """RCCL communicator owned by libspmv_amd.so (spmv_hip_comm_*)."""

    def __init__(self, rank, world, exchange_id):
        """exchange_id(bytes_or_None) -> bytes: hands rank 0's id to every rank."""
        buf = C.create_string_buffer(nat.COMM_ID_BYTES)
        if rank == 0:
            _check(nat.lib().spmv_hip_comm_get_id(buf), "spmv_hip_comm_get_id")
        ident = exchange_id(bytes(buf.raw) if rank == 0 else None)
        if len(ident) != nat.COMM_ID_BYTES:
            raise SpmvHipError("communicator id has the wrong length")
        _check(nat.lib().spmv_hip_comm_init(ident, int(rank), int(world)), "spmv_hip_comm_init")
        self.rank, self.world = rank, world

    def allgatherv(self, d_y: int, bounds, value_bytes=8, stream: int = 0):
        b = np.ascontiguousarray(bounds, dtype=np.int32)
        _check(nat.lib().spmv_hip_comm_allgatherv(C.c_void_p(d_y), b.ctypes.data_as(nat.c_int_p),
                                                  int(value_bytes), C.c_void_p(stream)),
               "spmv_hip_comm_allgatherv")

    def autotune(self, d_y: int, bounds, value_bytes=8, iters=10):
        """Collective: time both all-gatherv implementations on this node, keep the faster one that
        reproduces the other bit for bit.  Returns (mode, ms_broadcasts, ms_padded_allgather)."""
        b = np.ascontiguousarray(bounds, dtype=np.int32)
        mode = C.c_int(0)
        ms = (C.c_float * 2)()
        _check(nat.lib().spmv_hip_comm_autotune(C.c_void_p(d_y), b.ctypes.data_as(nat.c_int_p), int(value_bytes),
                                                int(iters), C.byref(mode), ms), "spmv_hip_comm_autotune")
        return int(mode.value), float(ms[0]), float(ms[1])

    def halo_setup(self, dev, bounds):
        """Collective: publish what dev needs of x, derive this rank's send / receive segments."""
        b = np.ascontiguousarray(bounds, dtype=np.int32)
        _check(nat.lib().spmv_hip_comm_halo_setup(dev.h, b.ctypes.data_as(nat.c_int_p)), "spmv_hip_comm_halo_setup")
        s, r, p = C.c_longlong(0), C.c_longlong(0), C.c_int(0)
        _check(nat.lib().spmv_hip_comm_halo_info(C.byref(s), C.byref(r), C.byref(p)), "spmv_hip_comm_halo_info")
        return {"send_values": int(s.value), "recv_values": int(r.value), "peers": int(p.value)}

    def halo_exchange(self, d_vec: int, value_bytes=8, stream: int = 0):
        _check(nat.lib().spmv_hip_comm_halo_exchange(C.c_void_p(d_vec), int(value_bytes), C.c_void_p(stream)),
               "spmv_hip_comm_halo_exchange")

    def rccl_ranks(self):
        """(rank, ranks) as RCCL itself reports them for the communicator."""
        r, n = C.c_int(-1), C.c_int(-1)
        _check(nat.lib().spmv_hip_comm_info(C.byref(r), C.byref(n)), "spmv_hip_comm_info")
        return int(r.value), int(n.value)

    def close(self):
        nat.lib().spmv_hip_comm_destroy()


class RowPartitionedCsr:
    """This rank's row block of a CSR matrix on its GPU + the y exchange."""

    def __init__(self, M, N, row_ptr_full, local_col, local_val, bounds, rank, comm=None):
        self.M, self.N = int(M), int(N)
        self.bounds = np.ascontiguousarray(bounds, dtype=np.int32)
        self.rank = rank
        self.row0, self.row1 = local_rows(self.bounds, rank)
        # upload only this rank's block: hand the C-ABI a CSR whose row_ptr is
        # the full one but whose col/val pointers are shifted so that entry
        # row_ptr[row0] is element 0 of the local arrays
        local_rp = local_row_ptr(row_ptr_full, self.row0, self.row1)
        self.dev = CsrDevice(self.M, self.N, local_rp, local_col, local_val, self.row0, self.row1)
        self.comm = comm
        self.value_bytes = 4 if np.dtype(self.dev.dtype) == np.float32 else 8

    @classmethod
    def balanced_bounds(cls, row_ptr_full, world):
        return partition_rows(row_ptr_full, world)

    def step(self, variant=CSR_AUTO):
        """y = A x on this rank's rows, then all-gatherv(y) (asynchronous)."""
        self.dev.run(variant)
        if self.comm is not None:
            self.comm.allgatherv(self.dev.y_ptr, self.bounds, self.value_bytes)

    def close(self):
        self.dev.close()
