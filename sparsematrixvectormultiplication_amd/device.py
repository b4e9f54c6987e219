"""GPU side of the path: thin objects over the C-ABI of include/spmv_hip.h.

Shape dictated by the reference's CUDA driver (main_cuda.cu): upload once
(:135-145, :369-402) -> run many (:166, :238, :317, :454, :568, :637) ->
fetch y (:183).  No CPU fallback anywhere: a failed C call raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat
from .host import CsrHost, HllHost

CSR_AUTO, CSR_THREAD_ROW, CSR_WAVE_ROW, CSR_SUBWAVE, CSR_STREAM = 0, 1, 2, 3, 4
HLL_AUTO, HLL_THREAD_ROW, HLL_SUBWAVE, HLL_LDS = 0, 1, 2, 3
CSR_STREAM_KERNELS = ("csr_stream", "csr_stream_local", "csr_stream_short", "csr_tile")
HLL_LDS_KERNELS = ("hll_lds", "hll_lds_local", "csr_tile (HLL slab rows)")
CSR_VARIANTS = {"thread_row": CSR_THREAD_ROW, "wave_row": CSR_WAVE_ROW, "subwave": CSR_SUBWAVE,
                "stream": CSR_STREAM}
HLL_VARIANTS = {"thread_row": HLL_THREAD_ROW, "subwave": HLL_SUBWAVE, "lds": HLL_LDS}


class SpmvHipError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        msg = nat.lib().spmv_hip_last_error()
        raise SpmvHipError(f"{what}: {msg.decode(errors='replace') if msg else 'failed (-1)'}")


def device_count() -> int:
    return nat.lib().spmv_hip_device_count()


def hip_init(device: int = 0) -> None:
    _check(nat.lib().spmv_hip_init(int(device)), "spmv_hip_init")


def hip_sync() -> None:
    _check(nat.lib().spmv_hip_sync(), "spmv_hip_sync")


def hip_stream() -> int:
    return nat.lib().spmv_hip_stream() or 0


def device_name():
    buf = C.create_string_buffer(256)
    cus, mem = C.c_int(), C.c_longlong()
    _check(nat.lib().spmv_hip_device_name(buf, 256, C.byref(cus), C.byref(mem)), "device_name")
    return buf.value.decode(), cus.value, mem.value


def flush_cache(nbytes: int = 1 << 30) -> None:
    """Refill L2 + the 256 MiB Infinity Cache with scratch data."""
    _check(nat.lib().spmv_hip_flush_cache(int(nbytes)), "spmv_hip_flush_cache")


def stream_probe(nbytes: int = 1 << 30, warmup: int = 3, iters: int = 10):
    """(mean, min) ms of a read-only 16-byte-per-lane stream over `nbytes` of HBM (spmv_hip_stream_probe)."""
    mean, mn = C.c_float(0), C.c_float(0)
    _check(nat.lib().spmv_hip_stream_probe(int(nbytes), int(warmup), int(iters), C.byref(mean), C.byref(mn)),
           "spmv_hip_stream_probe")
    return float(mean.value), float(mn.value)


def stream_probe_at(dptr: int, nbytes: int, warmup: int = 2, iters: int = 8):
    """(mean, min) ms of the read-only stream over [dptr, dptr + nbytes) (spmv_hip_stream_probe_at)."""
    mean, mn = C.c_float(0), C.c_float(0)
    _check(nat.lib().spmv_hip_stream_probe_at(C.c_void_p(dptr), int(nbytes), int(warmup), int(iters), C.byref(mean),
                                              C.byref(mn)), "spmv_hip_stream_probe_at")
    return float(mean.value), float(mn.value)


def gather_probe(value_bytes: int = 4, table_bytes: int = 2 << 20, waves_per_cu: int = 16) -> float:
    """values / s of 64-different-lines gathers from an L2-resident table (spmv_hip_gather_probe)."""
    out = C.c_double(0)
    _check(nat.lib().spmv_hip_gather_probe(int(value_bytes), int(table_bytes), int(waves_per_cu), C.byref(out)),
           "spmv_hip_gather_probe")
    return float(out.value)


def _read(path, limit=400):
    try:
        with open(path) as fh:
            return fh.read(limit).strip()
    except OSError:
        return None


def box_state(probe_bytes: int = 1 << 30) -> dict:
    """What distinguishes one GPU box of the pool from another, for bench records: the HIP attributes
    (spmv_hip_device_state), the card's sysfs state (partition modes, DPM clock tables with the active level,
    power cap, VBIOS) found through its PCI bus id, and the time of a read-only stream over 1 GiB."""
    import glob
    import os
    buf = C.create_string_buffer(512)
    _check(nat.lib().spmv_hip_device_state(buf, 512), "spmv_hip_device_state")
    out = dict(item.split("=", 1) for item in buf.value.decode().split(";") if "=" in item)
    for k in ("cus", "xcds", "sclk_khz", "mclk_khz", "mem_bus_bits", "l2_bytes", "hbm_bytes", "hbm_free_bytes"):
        if k in out:
            out[k] = int(out[k])
    pci = out.get("pci", "").lower()
    card = None
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if pci and os.path.basename(os.path.realpath(dev)).lower() == pci:
            card = dev
            break
    sysfs = {}
    if card:
        for name in ("current_compute_partition", "current_memory_partition", "pp_dpm_sclk", "pp_dpm_mclk",
                     "pp_dpm_fclk", "pp_dpm_socclk", "power_dpm_force_performance_level", "vbios_version",
                     "mem_info_vram_used", "unique_id"):
            v = _read(os.path.join(card, name))
            if v is not None:
                # DPM tables: keep the active level ("*") only
                if name.startswith("pp_dpm_"):
                    act = [ln for ln in v.splitlines() if ln.rstrip().endswith("*")]
                    v = {"active": act[0].rstrip(" *") if act else None, "levels": len(v.splitlines())}
                sysfs[name] = v
        for hw in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*"))):
            for name in ("power1_cap", "power1_average", "power1_input", "temp1_input", "freq1_input", "freq2_input"):
                v = _read(os.path.join(hw, name))
                if v is not None:
                    sysfs[name] = v
    out["sysfs_card"] = card
    out["sysfs"] = sysfs
    if probe_bytes:
        mean, mn = stream_probe(probe_bytes, 3, 10)
        out["stream_probe"] = {"bytes": int(probe_bytes), "ms_mean": round(mean, 5), "ms_min": round(mn, 5),
                               "gbps_mean": round(probe_bytes / (mean * 1e-3) / 1e9, 1)}
    return out


def set_tuning(key: str, value: int) -> None:
    """A/B knobs of the stream kernel: stream_cap (at upload), stream_nt, stream_xcd."""
    _check(nat.lib().spmv_hip_set_tuning(key.encode(), int(value)), "spmv_hip_set_tuning")


class _Handle:
    _free = None

    def __init__(self):
        self.h = C.c_void_p()

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            getattr(nat.lib(), self._free)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class CsrDevice(_Handle):
    """A CSR matrix, or rows [row0, row1) of one, resident in HBM."""

    _free = "spmv_hip_csr_free"

    def __init__(self, M, N, row_ptr, col_idx, values, row0=0, row1=None):
        super().__init__()
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        if values.dtype == np.float32:
            values = np.ascontiguousarray(values)
            fn, vp, self.dtype = nat.lib().spmv_hip_csr_upload_f32, nat.c_float_p, np.float32
        else:
            values = np.ascontiguousarray(values, dtype=np.float64)
            fn, vp, self.dtype = nat.lib().spmv_hip_csr_upload, nat.c_double_p, np.float64
        if len(row_ptr) != M + 1:
            raise ValueError("row_ptr must have M + 1 entries")
        row1 = M if row1 is None else row1
        _check(fn(int(M), int(N), row_ptr.ctypes.data_as(nat.c_int_p),
                  col_idx.ctypes.data_as(nat.c_int_p), values.ctypes.data_as(vp), int(row0),
                  int(row1), C.byref(self.h)), "spmv_hip_csr_upload")
        self.M, self.N = int(M), int(N)

    @classmethod
    def from_coo(cls, M, N, I, J, val):
        """CSR built on the device from COO triplets (spmv_hip_csr_from_coo); fp64."""
        I = np.ascontiguousarray(I, dtype=np.int32)
        J = np.ascontiguousarray(J, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        self = cls.__new__(cls)
        _Handle.__init__(self)
        _check(nat.lib().spmv_hip_csr_from_coo(int(M), int(N), len(I), I.ctypes.data_as(nat.c_int_p),
                                               J.ctypes.data_as(nat.c_int_p), val.ctypes.data_as(nat.c_double_p),
                                               C.byref(self.h)), "spmv_hip_csr_from_coo")
        self.M, self.N, self.dtype = int(M), int(N), np.float64
        return self

    def download(self):
        """(row_ptr, col_idx, values) of the handle's rows, from the device."""
        info = self.info()
        rp = np.zeros(info["M_local"] + 1, dtype=np.int32)
        col = np.zeros(max(info["nz"], 1), dtype=np.int32)
        val = np.zeros(max(info["nz"], 1), dtype=self.dtype)
        _check(nat.lib().spmv_hip_csr_download(self.h, rp.ctypes.data_as(nat.c_int_p), col.ctypes.data_as(nat.c_int_p),
                                               val.ctypes.data_as(C.c_void_p)), "spmv_hip_csr_download")
        return rp, col[:info["nz"]], val[:info["nz"]]

    @classmethod
    def from_host(cls, csr: CsrHost, row0=0, row1=None):
        return cls(csr.M, csr.N, csr.row_ptr, csr.col_idx, csr.values, row0, row1)

    def info(self) -> dict:
        out = nat.DevInfo()
        _check(nat.lib().spmv_hip_csr_info(self.h, C.byref(out)), "spmv_hip_csr_info")
        return out.as_dict()

    def addresses(self) -> dict:
        out = (C.c_ulonglong * 8)()
        _check(nat.lib().spmv_hip_csr_addresses(self.h, out), "spmv_hip_csr_addresses")
        return dict(zip(("row_ptr", "col", "val", "x", "y", "lcol", "lines", "ldesc4"), (int(v) for v in out)))

    ARRAYS = ("row_ptr", "col", "val", "x", "y", "lcol", "lines", "ldesc4")

    def tile_digest(self):
        """(elements, hash) of each of the 32 arrays of the handle's tile plans (spmv_hip_csr_tile_digest)."""
        out = (C.c_ulonglong * 64)()
        _check(nat.lib().spmv_hip_csr_tile_digest(self.h, out), "spmv_hip_csr_tile_digest")
        return [(int(out[2 * k]), int(out[2 * k + 1])) for k in range(32)]

    def stamp_blocks(self, warm: int = 3):
        """(start, end, dispatch id, xcd) per x-window block of one stamped launch; times in ticks of 10 ns."""
        n = self.info()["local_blocks"]
        buf = np.zeros(3 * n, dtype=np.uint64)
        _check(nat.lib().spmv_hip_csr_stamp_blocks(self.h, int(warm), buf.ctypes.data_as(C.POINTER(C.c_ulonglong))),
               "spmv_hip_csr_stamp_blocks")
        buf = buf.reshape(n, 3)
        return buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64), (buf[:, 2] >> np.uint64(8)).astype(np.int64), \
            (buf[:, 2] & np.uint64(0xf)).astype(np.int64)

    def relocate(self, which: str, align: int, offset: int, vmm: bool = False):
        fn = nat.lib().spmv_hip_csr_relocate_vmm if vmm else nat.lib().spmv_hip_csr_relocate
        _check(fn(self.h, self.ARRAYS.index(which), int(align), int(offset)), "spmv_hip_csr_relocate")

    def set_x(self, x):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        if len(x) != self.N:
            raise ValueError(f"x has {len(x)} entries, matrix has {self.N} columns")
        _check(nat.lib().spmv_hip_csr_set_x(self.h, x.ctypes.data_as(C.c_void_p)), "csr_set_x")

    def run(self, variant=CSR_AUTO):
        _check(nat.lib().spmv_hip_csr_run(self.h, int(variant)), "spmv_hip_csr_run")

    def get_y(self):
        y = np.empty(self.M, dtype=self.dtype)
        _check(nat.lib().spmv_hip_csr_get_y(self.h, y.ctypes.data_as(C.c_void_p)), "csr_get_y")
        return y

    def spmv(self, x, variant=CSR_AUTO):
        self.set_x(x)
        self.run(variant)
        return self.get_y()

    def run_on(self, d_x: int, d_y: int, variant=CSR_AUTO, stream: int = 0):
        _check(nat.lib().spmv_hip_csr_run_on(self.h, int(variant), C.c_void_p(d_x),
                                             C.c_void_p(d_y), C.c_void_p(stream)), "csr_run_on")

    x_ptr = property(lambda s: nat.lib().spmv_hip_csr_x_ptr(s.h) or 0)
    y_ptr = property(lambda s: nat.lib().spmv_hip_csr_y_ptr(s.h) or 0)

    def time(self, variant=CSR_AUTO, warmup=5, iters=95, zero_y=True):
        """Per-launch kernel milliseconds, reference protocol (main_cuda.cu:159-200)."""
        ms = np.zeros(iters, dtype=np.float32)
        _check(nat.lib().spmv_hip_csr_time(self.h, int(variant), int(warmup), int(iters),
                                           int(bool(zero_y)), ms.ctypes.data_as(nat.c_float_p)),
               "csr_time")
        return ms

    def time_graph(self, variant=CSR_AUTO, iters=20, replays=10) -> float:
        """ms per SpMV when `iters` launches are replayed from one hipGraph."""
        ms = C.c_float(0)
        _check(nat.lib().spmv_hip_csr_time_graph(self.h, int(variant), int(iters), int(replays), C.byref(ms)),
               "csr_time_graph")
        return float(ms.value)

    def power_iterate(self, iters, variant=CSR_AUTO, bounds=None, use_graph=True):
        """iters steps of x <- A x / ||A x||_2 on the device; returns (lambda, ms_total)."""
        lam, ms = C.c_double(0), C.c_float(0)
        b = None if bounds is None else np.ascontiguousarray(bounds, dtype=np.int32)
        _check(nat.lib().spmv_hip_csr_power_iterate(self.h, int(variant), int(iters),
                                                    None if b is None else b.ctypes.data_as(nat.c_int_p),
                                                    int(bool(use_graph)), C.byref(lam), C.byref(ms)),
               "csr_power_iterate")
        return float(lam.value), float(ms.value)

    def cg(self, b, iters, variant=CSR_AUTO, bounds=None, use_halo=False):
        """iters steps of conjugate gradients from x0 = 0 (spmv_hip_csr_cg); returns (x, r.r history, ms)."""
        b = np.ascontiguousarray(b, dtype=self.dtype)
        if len(b) != self.M:
            raise ValueError("b must have M entries")
        x = np.zeros(self.M, dtype=self.dtype)
        hist = np.zeros(iters + 1)
        ms = C.c_float(0)
        bb = None if bounds is None else np.ascontiguousarray(bounds, dtype=np.int32)
        _check(nat.lib().spmv_hip_csr_cg(self.h, int(variant), int(iters),
                                         None if bb is None else bb.ctypes.data_as(nat.c_int_p), int(bool(use_halo)),
                                         b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                                         hist.ctypes.data_as(nat.c_double_p), C.byref(ms)), "spmv_hip_csr_cg")
        return x, hist, float(ms.value)

    def split_interior(self) -> dict:
        """Split the x-window blocks into interior (own range of x only) and boundary ones
        (spmv_hip_csr_split_interior); returns the block and entry counts."""
        counts = (C.c_longlong * 4)()
        _check(nat.lib().spmv_hip_csr_split_interior(self.h, counts), "spmv_hip_csr_split_interior")
        return dict(zip(("interior_blocks", "boundary_blocks", "interior_entries", "boundary_entries"),
                        (int(v) for v in counts)))

    def split_columns(self, col_lo: int, col_hi: int) -> dict:
        """Split the handle's entries by column ([col_lo, col_hi) = the rank's own range of x) into two sub-handles
        (spmv_hip_csr_split_columns); returns the entry counts inside / outside the range."""
        counts = (C.c_longlong * 2)()
        _check(nat.lib().spmv_hip_csr_split_columns(self.h, int(col_lo), int(col_hi), counts), "spmv_hip_csr_split_columns")
        return {"own_entries": int(counts[0]), "halo_entries": int(counts[1])}

    def run_split(self, part: int):
        """part 0: y = A_own x (own range of x only); part 1: y += A_halo x (asynchronous on the library stream)."""
        _check(nat.lib().spmv_hip_csr_run_split(self.h, int(part), None, None, None), "spmv_hip_csr_run_split")

    def run_part(self, part: int):
        """part 0: interior blocks only; part 1: the rest (asynchronous on the library stream)."""
        _check(nat.lib().spmv_hip_csr_run_part(self.h, int(part), None, None, None), "spmv_hip_csr_run_part")

    def power_iterate_halo(self, iters, variant=CSR_AUTO):
        """power_iterate with the halo exchange (NativeComm.halo_setup first when a communicator exists)."""
        lam, ms = C.c_double(0), C.c_float(0)
        _check(nat.lib().spmv_hip_csr_power_iterate_halo(self.h, int(variant), int(iters), C.byref(lam), C.byref(ms)),
               "csr_power_iterate_halo")
        return float(lam.value), float(ms.value)

    def get_x(self):
        x = np.empty(self.N, dtype=self.dtype)
        _check(nat.lib().spmv_hip_memcpy_d2h(x.ctypes.data_as(C.c_void_p), C.c_void_p(self.x_ptr), x.nbytes),
               "memcpy_d2h")
        return x

    def step_time(self, bounds, variant=CSR_AUTO, warmup=5, iters=95):
        """Multi-GPU step (SpMV + all-gatherv of y): per-step kernel and exchange ms."""
        b = np.ascontiguousarray(bounds, dtype=np.int32)
        mk, mx = np.zeros(iters, np.float32), np.zeros(iters, np.float32)
        _check(nat.lib().spmv_hip_csr_step_time(self.h, int(variant), b.ctypes.data_as(nat.c_int_p),
                                                int(warmup), int(iters),
                                                mk.ctypes.data_as(nat.c_float_p),
                                                mx.ctypes.data_as(nat.c_float_p)), "csr_step_time")
        return mk, mx


class HllDevice(_Handle):
    """An HLL matrix resident in HBM as one flat slab."""

    _free = "spmv_hip_hll_free"

    def __init__(self, hll: HllHost = None, hack0=0, hack1=None):
        """The whole matrix, or hacks [hack0, hack1) of it (one rank's share)."""
        super().__init__()
        if hll is not None:
            hack1 = hll.num_blocks if hack1 is None else hack1
            _check(nat.lib().spmv_hip_hll_upload_part(C.byref(hll.c), int(hll.M), int(hll.N), int(hack0),
                                                      int(hack1), C.byref(self.h)), "spmv_hip_hll_upload_part")
            self.M, self.N = hll.M, hll.N

    def time_graph(self, variant=HLL_AUTO, iters=20, replays=10) -> float:
        ms = C.c_float(0)
        _check(nat.lib().spmv_hip_hll_time_graph(self.h, int(variant), int(iters), int(replays), C.byref(ms)),
               "hll_time_graph")
        return float(ms.value)

    def step_time(self, row_bounds, variant=HLL_AUTO, warmup=5, iters=95):
        """Multi-GPU step (SpMV on this rank's hacks + all-gatherv of y): kernel and exchange ms."""
        b = np.ascontiguousarray(row_bounds, dtype=np.int32)
        mk, mx = np.zeros(iters, np.float32), np.zeros(iters, np.float32)
        _check(nat.lib().spmv_hip_hll_step_time(self.h, int(variant), b.ctypes.data_as(nat.c_int_p),
                                                int(warmup), int(iters), mk.ctypes.data_as(nat.c_float_p),
                                                mx.ctypes.data_as(nat.c_float_p)), "hll_step_time")
        return mk, mx

    @classmethod
    def from_csr_device(cls, csr: "CsrDevice"):
        """HLL built on the GPU from a resident CSR matrix (spmv_hip_hll_from_csr)."""
        self = cls()
        _check(nat.lib().spmv_hip_hll_from_csr(csr.h, C.byref(self.h)), "spmv_hip_hll_from_csr")
        self.M, self.N = csr.info()["M_total"], csr.N
        return self

    def run_on(self, d_x: int, d_y: int, variant=HLL_AUTO, stream: int = 0):
        _check(nat.lib().spmv_hip_hll_run_on(self.h, int(variant), C.c_void_p(d_x), C.c_void_p(d_y),
                                             C.c_void_p(stream)), "hll_run_on")

    x_ptr = property(lambda s: nat.lib().spmv_hip_hll_x_ptr(s.h) or 0)
    y_ptr = property(lambda s: nat.lib().spmv_hip_hll_y_ptr(s.h) or 0)

    def tile_digest(self):
        out = (C.c_ulonglong * 64)()
        _check(nat.lib().spmv_hip_hll_tile_digest(self.h, out), "spmv_hip_hll_tile_digest")
        return [(int(out[2 * k]), int(out[2 * k + 1])) for k in range(32)]

    def download(self):
        """(hack_off, maxnz, JA, AS) of the flat device slab."""
        info = self.info()
        H = info["hacks"]
        off = np.zeros(H + 1, dtype=np.int64)
        mz = np.zeros(max(H, 1), dtype=np.int32)
        _check(nat.lib().spmv_hip_hll_download(self.h, off.ctypes.data_as(C.POINTER(C.c_longlong)),
                                               mz.ctypes.data_as(nat.c_int_p), None, None), "hll_download")
        S = int(off[H])
        ja, as_ = np.zeros(max(S, 1), np.int32), np.zeros(max(S, 1), np.float64)
        _check(nat.lib().spmv_hip_hll_download(self.h, None, None, ja.ctypes.data_as(nat.c_int_p),
                                               as_.ctypes.data_as(nat.c_double_p)), "hll_download")
        return off, mz[:H], ja[:S], as_[:S]

    def info(self) -> dict:
        out = nat.DevInfo()
        _check(nat.lib().spmv_hip_hll_info(self.h, C.byref(out)), "spmv_hip_hll_info")
        return out.as_dict()

    def set_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if len(x) != self.N:
            raise ValueError(f"x has {len(x)} entries, matrix has {self.N} columns")
        _check(nat.lib().spmv_hip_hll_set_x(self.h, x.ctypes.data_as(nat.c_double_p)), "hll_set_x")

    def run(self, variant=HLL_AUTO):
        _check(nat.lib().spmv_hip_hll_run(self.h, int(variant)), "spmv_hip_hll_run")

    def get_y(self):
        y = np.empty(self.M, dtype=np.float64)
        _check(nat.lib().spmv_hip_hll_get_y(self.h, y.ctypes.data_as(nat.c_double_p)), "hll_get_y")
        return y

    def spmv(self, x, variant=HLL_AUTO):
        self.set_x(x)
        self.run(variant)
        return self.get_y()

    def time(self, variant=HLL_AUTO, warmup=5, iters=95, zero_y=True):
        ms = np.zeros(iters, dtype=np.float32)
        _check(nat.lib().spmv_hip_hll_time(self.h, int(variant), int(warmup), int(iters),
                                           int(bool(zero_y)), ms.ctypes.data_as(nat.c_float_p)),
               "hll_time")
        return ms
