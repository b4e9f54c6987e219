#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    make -C oracle            # compiles /root/reference -> oracle/_ref/libspmv_ref.so
    python tests/golden/make_golden.py

For every case it writes <case>.mtx (the input; general_matrix.mtx is the data
file the reference ships as matrix_generated/general_matrix.mtx, the others
are produced here from a seeded numpy generator) and <case>.npz holding what the
compiled reference computed from that file:

  pre_I/J/val            read_matrix_market            (src/matrix_parser.c:25)
  row_ptr/col_idx/values convert_in_csr                (src/csr_matrix.c:63)
  hll_rows/maxnz/JA/AS   convert_to_hll, hacks flattened in order (src/hll_matrix.c:37)
  y_ones, y_rand         csr_matrix_vector_mult, x = 1 and x = x_rand (src/csr_matrix.c:130)
  yh_ones, yh_rand       spmv_hll_serial                (src/hll_matrix.c:286)
  part_T<k>_{s,e}        prepare_thread_distribution     (src/csr_matrix.c:167)
  hpart_T<k>_{s,e}       prepare_thread_distribution_hll (src/hll_matrix.c:410)

Only data is stored -- no reference source text.
"""
import ctypes as C
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.oracle import Reference  # noqa: E402
from sparsematrixvectormultiplication_amd import _native as nat  # noqa: E402

REF_MTX = "/root/reference/matrix_generated/general_matrix.mtx"


def write_mtx(path, M, N, rows, cols, vals, field="real", symmetry="general", comment=None):
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate {field} {symmetry}\n")
        f.write(f"% {comment or 'seeded test matrix for the SpMV golden vectors'}\n")
        f.write(f"{M} {N} {len(rows)}\n")
        for k in range(len(rows)):
            if field == "pattern":
                f.write(f"{rows[k] + 1} {cols[k] + 1}\n")
            elif field == "integer":
                f.write(f"{rows[k] + 1} {cols[k] + 1} {int(vals[k])}\n")
            else:
                f.write(f"{rows[k] + 1} {cols[k] + 1} {float(vals[k])!r}\n")


def random_coo(rng, M, N, nnz, lower_only=False, empty_rows=()):
    seen, rows, cols = set(), [], []
    while len(rows) < nnz:
        i, j = int(rng.integers(M)), int(rng.integers(N))
        if lower_only and j > i:
            i, j = j, i
        if i in empty_rows or (lower_only and j in empty_rows) or (i, j) in seen:
            continue
        seen.add((i, j))
        rows.append(i)
        cols.append(j)
    return np.array(rows), np.array(cols), rng.uniform(-1, 1, nnz)


def make_inputs():
    rng = np.random.default_rng(20250704)
    shutil.copyfile(REF_MTX, os.path.join(HERE, "general_matrix.mtx"))
    cases = ["general_matrix"]

    # symmetric real, M = 97 (not a multiple of 32), several empty rows
    r, c, v = random_coo(rng, 97, 97, 420, lower_only=True, empty_rows={0, 31, 32, 64, 96})
    write_mtx(os.path.join(HERE, "sym_empty_rows.mtx"), 97, 97, r, c, v, symmetry="symmetric")
    cases.append("sym_empty_rows")

    # rectangular pattern matrix
    r, c, v = random_coo(rng, 70, 45, 300)
    write_mtx(os.path.join(HERE, "pattern_rect.mtx"), 70, 45, r, c, v, field="pattern")
    cases.append("pattern_rect")

    # one dense row among short ones, entries in shuffled order, integer field
    r, c, v = random_coo(rng, 200, 200, 500)
    keep = r != 77
    r, c = r[keep], c[keep]
    r = np.concatenate([r, np.full(200, 77)])
    c = np.concatenate([c, rng.permutation(200)])
    v = rng.integers(-9, 10, len(r)).astype(float)
    order = rng.permutation(len(r))
    write_mtx(os.path.join(HERE, "long_row_int.mtx"), 200, 200, r[order], c[order], v[order],
              field="integer")
    cases.append("long_row_int")

    # the same (i, j) stored several times: exercises the sort's tie order
    r, c, v = random_coo(rng, 40, 40, 150)
    dup = rng.integers(0, 150, 60)
    r = np.concatenate([r, r[dup], r[dup[:20]]])
    c = np.concatenate([c, c[dup], c[dup[:20]]])
    v = np.concatenate([v, rng.uniform(-1, 1, 80)])
    order = rng.permutation(len(r))
    write_mtx(os.path.join(HERE, "dup_entries.mtx"), 40, 40, r[order], c[order], v[order])
    cases.append("dup_entries")

    # symmetric pattern, M a multiple of 32, with diagonal entries
    r, c, v = random_coo(rng, 64, 64, 260, lower_only=True)
    write_mtx(os.path.join(HERE, "sym_pattern.mtx"), 64, 64, r, c, v, field="pattern",
              symmetry="symmetric")
    cases.append("sym_pattern")

    # banded matrix with wide value range (cancellation inside rows)
    rows, cols, vals = [], [], []
    for i in range(130):
        for d in (-17, -3, -1, 0, 1, 3, 17):
            j = i + d
            if 0 <= j < 130:
                rows.append(i)
                cols.append(j)
                vals.append(float(rng.uniform(-1, 1) * 10.0 ** rng.integers(-8, 9)))
    write_mtx(os.path.join(HERE, "banded_scaled.mtx"), 130, 130, np.array(rows), np.array(cols),
              np.array(vals))
    cases.append("banded_scaled")

    # degenerate shapes
    write_mtx(os.path.join(HERE, "one_by_one.mtx"), 1, 1, np.array([0]), np.array([0]),
              np.array([-2.5]))
    cases.append("one_by_one")
    write_mtx(os.path.join(HERE, "no_entries.mtx"), 5, 7, np.array([], int), np.array([], int),
              np.array([]))
    cases.append("no_entries")
    return cases


def view(ptr, n, dtype):
    return np.array(np.ctypeslib.as_array(ptr, shape=(n,)), dtype=dtype) if n > 0 and ptr else \
        np.zeros(0, dtype)


def take(ptr, n):
    out = view(ptr, n, np.int32)
    return out


def main():
    ref = Reference()
    cases = make_inputs()
    rng = np.random.default_rng(7)
    for name in cases:
        path = os.path.join(HERE, name + ".mtx")
        pre, csr, hll = ref.load(path)
        M, N, nz = csr.M, csr.N, csr.nz
        out = dict(M=M, N=N, nz=nz, typecode=np.frombuffer(bytes(pre.type), dtype=np.uint8),
                   pre_I=view(pre.I, nz, np.int32), pre_J=view(pre.J, nz, np.int32),
                   pre_val=view(pre.val, nz, np.float64),
                   row_ptr=view(csr.row_ptr, M + 1, np.int32),
                   col_idx=view(csr.col_idx, nz, np.int32),
                   values=view(csr.values, nz, np.float64))
        rows, maxnz, ja, as_ = [], [], [], []
        for b in range(hll.num_blocks):
            blk = hll.blocks[b]
            rows.append(blk.M)
            maxnz.append(blk.MAXNZ)
            ja.append(view(blk.JA, blk.M * blk.MAXNZ, np.int32))
            as_.append(view(blk.AS, blk.M * blk.MAXNZ, np.float64))
        out.update(hll_rows=np.array(rows, np.int32), hll_maxnz=np.array(maxnz, np.int32),
                   hll_JA=np.concatenate(ja) if ja else np.zeros(0, np.int32),
                   hll_AS=np.concatenate(as_) if as_ else np.zeros(0, np.float64))
        x_ones = np.ones(N)
        x_rand = rng.uniform(-1, 1, N)
        out.update(x_rand=x_rand, y_ones=ref.csr_serial(csr, x_ones),
                   y_rand=ref.csr_serial(csr, x_rand), yh_ones=ref.hll_serial(hll, M, x_ones),
                   yh_rand=ref.hll_serial(hll, M, x_rand))
        for T in (2, 3, 8):
            s, e = nat.c_int_p(), nat.c_int_p()
            n = ref.L.prepare_thread_distribution(M, csr.row_ptr, T, nz, C.byref(s), C.byref(e)) \
                if M > 0 else 0
            out[f"part_T{T}_s"], out[f"part_T{T}_e"] = take(s, n), take(e, n)
            s, e = nat.c_int_p(), nat.c_int_p()
            n = ref.L.prepare_thread_distribution_hll(C.byref(hll), T, C.byref(s), C.byref(e))
            out[f"hpart_T{T}_s"], out[f"hpart_T{T}_e"] = take(s, n), take(e, n)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(f"{name:18s} M={M:4d} N={N:4d} nz={nz:5d} hacks={hll.num_blocks}")


if __name__ == "__main__":
    main()
