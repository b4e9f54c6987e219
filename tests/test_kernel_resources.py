"""Register budgets of the hot kernels, checked on the code hipcc generates for gfx950 (no GPU needed).

csr_tile sits at ~126 VGPRs under a cap of 128 (two 512-thread workgroups per CU); small source changes have tipped
its register allocation into scratch before (private_segment_fixed_size > 0: 167 -> 197 us on the road-like matrix).
A spill there, or in the x-window kernels, is a performance bug that no parity test sees."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
SRC = os.path.join(ROOT, "sparsematrixvectormultiplication_amd", "csrc", "hip")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hot_kernels_do_not_spill_and_keep_their_occupancy():
    tmp = tempfile.mkdtemp(prefix="spmv_regs_")
    try:
        proc = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                               "-I" + SRC, "-c", os.path.join(SRC, "spmv_csr.hip"), "-o", os.path.join(tmp, "o.o"),
                               "-save-temps=obj"], capture_output=True, text=True, timeout=600, cwd=tmp)
        assert proc.returncode == 0, proc.stderr[-2000:]
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
        assert asm, os.listdir(tmp)
        text = open(os.path.join(tmp, asm[0])).read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # the kernel descriptors' metadata: name, scratch bytes, VGPRs
    kernels = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
        kernels[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    # csr_tile<T, NT, 2048, 4, PACK, GA>: the GA (gather ahead) instantiations are an experiment that did not pay and is off
    # by default (profiles/r3_ab_gather_ahead.txt); the product's eight are the ones held to the budget
    all_tile = {k: v for k, v in kernels.items() if "csr_tile" in k}
    # (Li1 / Li2: the packed kernel with one or two staging trips, for the expanded plans)
    tile = {k: v for k, v in all_tile.items() if re.search(r"csr_tileI[df]Lb[01]ELi2048ELi[124]ELb[01]ELb0EE", k)}
    # (the measurement-only STAMP instantiations of csr_stream_local are the ...ELb1EE ones)
    local = {k: v for k, v in kernels.items() if "csr_stream_local" in k and not k.split("csr_stream_localI")[1].startswith(
        ("dLb0ELi2048ELb1", "dLb1ELi2048ELb1"))}
    assert len(tile) == 12 and len(all_tile) == 16 and len(local) >= 6, sorted(kernels)  # {fp64, fp32} x {nt} x {packed}; stages x {nt} x dtypes
    expand = {k: v for k, v in kernels.items() if "tile_expand" in k}  # x into the passes' segments (expanded plans)
    assert len(expand) == 2, sorted(kernels)
    for name, (scratch, vgprs) in {**tile, **local, **expand}.items():
        assert scratch == 0, f"{name} spills {scratch} bytes of scratch ({vgprs} VGPRs)"
    for name, (_, vgprs) in tile.items():
        assert vgprs <= 128, f"{name}: {vgprs} VGPRs: two workgroups per CU no longer fit"
    for name, (_, vgprs) in local.items():
        if "Li2048" in name:
            assert vgprs <= 72, f"{name}: {vgprs} VGPRs: seven workgroups per CU no longer fit"
