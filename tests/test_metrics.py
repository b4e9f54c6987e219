"""performance_calculate.h semantics (M1-M4 of SURVEY.md 8a)."""
import ctypes as C

import numpy as np
import pytest

import sparsematrixvectormultiplication_amd as sp
from oracle.oracle import Reference, have_reference


def test_calculate_flops_is_two_per_nonzero():
    assert sp.calculate_flops(4007383, 0.000179) == 2.0 * 4007383 / 0.000179


def test_difference_metrics_cpu_form():
    ref = np.array([1.0, 2.0, 0.0, 1e-9, 100.0])
    res = np.array([1.0, 2.1, 2e-5, 5e-9, 100.0 + 2e-5])
    m = sp.compute_difference_metrics(ref, res, 1e-5, 1e-4)
    # element 1: |d| = 0.1 -> rel 0.1/2.1; element 2: |d| = 2e-5 > 1e-5, rel = 2e-5/1e-4 = 0.2;
    # element 3: below abs_tol; element 4: |d| = 2e-5 > abs_tol but rel 2e-7 <= rel_tol
    assert m.significant_diffs == 2
    assert m.mean_abs_err == 0.0
    assert m.mean_rel_err == pytest.approx((0.1 / 2.1 + 0.2) / 2, rel=1e-12)
    assert sp.compute_difference_metrics(ref, ref).significant_diffs == 0
    assert sp.compute_difference_metrics(ref[:0], ref[:0]).significant_diffs == 0


def test_difference_metrics_gpu_form():
    ref = np.array([1.0, -2.0, 0.0])
    res = np.array([1.5, -2.0, 1e-6])
    m = sp.compute_difference_metrics_gpu(ref, res, 1e-4)
    assert m.mean_abs_err == pytest.approx((0.5 + 0 + 1e-6) / 3, rel=1e-12)
    assert m.mean_rel_err == pytest.approx((0.5 / 1.5 + 0 + 1e-6 / 1e-4) / 3, rel=1e-12)


@pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built")
def test_difference_metrics_vs_compiled_reference():
    ref = Reference()
    rng = np.random.default_rng(2)
    a = rng.uniform(-1, 1, 5000)
    b = a + rng.uniform(-1, 1, 5000) * 10.0 ** rng.integers(-12, -1, 5000)
    dp = C.POINTER(C.c_double)
    theirs = ref.L.computeDifferenceMetrics(a.ctypes.data_as(dp), b.ctypes.data_as(dp), 5000, 1e-5,
                                            1e-4, False)
    mine = sp.compute_difference_metrics(a, b, 1e-5, 1e-4)
    assert (mine.significant_diffs, mine.mean_rel_err, mine.mean_abs_err) == \
        (theirs.significant_diffs, theirs.mean_rel_err, theirs.mean_abs_err)
    assert ref.L.calculate_flops(1234567, 0.0321) == sp.calculate_flops(1234567, 0.0321)


def test_running_metrics_protocol():
    """mean over timed samples; errors averaged over count + ITERATION_SKIP
    (src/performance_calculate.c:36-39,53-67)."""
    from sparsematrixvectormultiplication_amd import _native as nat
    lib = sp.lib()
    lib.initialize_metrics()
    lib.reset_medium_time_metrics()
    WARP_CSR_TIME = 7
    samples = [0.5, 0.25, 1.0, 0.75]
    for s in samples:
        lib.update_medium_metric(WARP_CSR_TIME, s)
    for _ in range(len(samples) + sp.ITERATION_SKIP):  # errors are accumulated every iteration
        d = nat.DiffMetrics(2.0, 4.0, 0)
        lib.accumulateErrors(C.byref(d), WARP_CSR_TIME)
    assert lib.get_metric_value(WARP_CSR_TIME) == np.mean(samples)
    avg = lib.computeAverageErrors(WARP_CSR_TIME)
    assert avg.mean_abs_err == 2.0 and avg.mean_rel_err == 4.0
    assert lib.get_metric_min(WARP_CSR_TIME) == 0.25
    assert lib.get_metric_median(WARP_CSR_TIME) == 0.625
    assert lib.get_metric_value(0) == 0.0  # untouched metric
    lib.reset_medium_time_metrics()
    assert lib.get_metric_value(WARP_CSR_TIME) == 0.0
    lib.cleanup_metrics()
