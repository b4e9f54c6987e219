import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["general_matrix", "sym_empty_rows", "pattern_rect", "long_row_int", "dup_entries",
                "sym_pattern", "banded_scaled", "one_by_one", "no_entries"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The product and checker libraries are built by __graft_entry__.build();
    build them on demand so a bare `pytest` works in a fresh checkout."""
    import __graft_entry__ as entry
    entry.build_if_missing()


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def gpu():
    """Initialise the HIP device once; fail loudly (never skip) if it is unusable."""
    import sparsematrixvectormultiplication_amd as sp
    sp.hip_init(0)
    return sp


def load_golden(name):
    """np.load with its default allow_pickle=False."""
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_path(name):
    return os.path.join(GOLDEN, name + ".mtx")
