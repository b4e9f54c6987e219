"""Small helpers shared by bench.py (kept out of the product package)."""
import numpy as np


def coo_of(row_ptr, col):
    """Row / column index arrays (file order = CSR order) of a CSR pattern."""
    rows = np.repeat(np.arange(len(row_ptr) - 1, dtype=np.int32), np.diff(row_ptr))
    return rows, np.asarray(col, dtype=np.int32)
