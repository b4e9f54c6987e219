"""The C-ABI library loads without a GPU and exports every symbol the headers declare."""
import ctypes
import os
import re

import sparsematrixvectormultiplication_amd as sp
from conftest import ROOT

# declared in the kept headers but implemented by the CHECKER (oracle/cpu_spmv.c),
# never by the product library -- the GPU path has no CPU fallback
ORACLE_ONLY = {"csr_matrix_vector_mult", "spvm_csr_parallel", "spvm_csr_parallel_simd",
               "spmv_hll_serial", "spmv_hll", "spmv_hll_simd"}
# declared by the reference and defined nowhere in it either
DECLARED_ONLY = {"save_hll_memory_stats"}


def header_functions():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        text = open(os.path.join(inc, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"^\s*#.*?(?<!\\)$", "", text, flags=re.M)
        for m in re.finditer(r"\b([A-Za-z_]\w*)\s*\([^;{}()]*(?:\([^()]*\)[^;{}()]*)*\)\s*;", text):
            names.add(m.group(1))
    return names - {"defined", "sizeof"}


def test_library_loads_without_gpu():
    lib = sp.lib()
    assert isinstance(lib, ctypes.CDLL)
    assert os.path.dirname(sp.LIB_PATH).endswith("sparsematrixvectormultiplication_amd")


def test_every_declared_symbol_is_exported():
    lib = sp.lib()
    declared = header_functions()
    assert {"spmv_hip_csr_upload", "convert_in_csr", "convert_to_hll",
            "computeDifferenceMetrics"} <= declared, "header parser lost functions"
    missing = []
    for name in sorted(declared - ORACLE_ONLY - DECLARED_ONLY):
        try:
            getattr(lib, name)
        except AttributeError:
            missing.append(name)
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    # and the binding table is in step with the headers
    assert set(sp.EXPORTED_SYMBOLS) <= declared, set(sp.EXPORTED_SYMBOLS) - declared


def test_product_does_not_export_cpu_spmv():
    """No CPU SpMV hides in the product: those symbols exist only in oracle/."""
    lib = sp.lib()
    for name in ORACLE_ONLY:
        assert not hasattr(lib, name), f"product library must not define {name}"


def test_gpu_entry_points_fail_loudly_without_device():
    import numpy as np
    import pytest
    if sp.device_count() > 0:
        pytest.skip("a HIP device is present; the no-device behaviour is checked on CPU hosts")
    with pytest.raises(sp.SpmvHipError):
        sp.hip_init(0)
    rp = np.array([0, 1], dtype=np.int32)
    with pytest.raises(sp.SpmvHipError):
        sp.CsrDevice(1, 1, rp, np.array([0], np.int32), np.array([1.0]))


def test_only_the_declared_c_symbols_leave_the_library():
    """libspmv_amd.map: the dynamic symbol table holds the functions include/*.h declare and nothing else --
    no C++ helpers (_Z...), no tuning globals."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", sp.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    declared = header_functions() - ORACLE_ONLY - DECLARED_ONLY - {"free"}
    assert not [s for s in exported if s.startswith("_Z")], "C++ symbols exported"
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    listed = set(re.findall(r"^\s+(\w+);", open(os.path.join(ROOT, "sparsematrixvectormultiplication_amd", "csrc",
                                                               "libspmv_amd.map")).read(), flags=re.M))
    assert listed == declared, (sorted(listed - declared), sorted(declared - listed))
