"""Known answers the REFERENCE holds, reproduced through this build's evaluation path: the maximum cross-range of the
shuttle re-entry problem, objective -0.5958800738629952 +- 1e-2 and, with the heating-rate bound added, -0.534620087611498
+- 1e-2, for LGL3 / LGL5 / LGL7 / Trapezoidal x {HighestOrderSpline, BlockConstant}
(/root/reference/asset_asrl/test/test_FullProblems/test_Reentry.py:116-127,184-224; 64 segments as in :112,176; SURVEY.md
section 8c item vi).  The solver loop is the small harness of tests/kkt_harness.py (PSIOPT is out of
scope); what is under test is everything that feeds it: the phase layout and index tables, the defect / mesh-spacing /
control-spline values, their Jacobians and the Lagrangian Hessian, and the sparse KKT assembly --

* on the CPU from the oracle (oracle/fullnlp.cpp): this is what pins the oracle to a reference-held number;
* on the GPU (-m gpu) from the device kernels through the C ABI and the C++ host shim's KktAssembly."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kkt_harness as kh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_OBJECTIVE, REFERENCE_TOLERANCE = -0.5958800738629952, 0.01     # test_Reentry.py:116-117
REFERENCE_OBJECTIVE_HEATING = -0.534620087611498                          # test_Reentry.py:119-120: with the heating-rate bound
CASES = [(m, c) for m in ("LGL3", "LGL5", "LGL7", "Trapezoidal") for c in ("HighestOrderSpline", "BlockConstant")]
# what this build's own 64-segment solutions give (recorded from the oracle run; the spread is discretisation error)
OWN = {("LGL3", "HighestOrderSpline"): -0.59587993, ("LGL3", "BlockConstant"): -0.59586140,
       ("LGL5", "HighestOrderSpline"): -0.59588007, ("LGL5", "BlockConstant"): -0.59586131,
       ("LGL7", "HighestOrderSpline"): -0.59588007, ("LGL7", "BlockConstant"): -0.59586131,
       ("Trapezoidal", "HighestOrderSpline"): -0.59581488, ("Trapezoidal", "BlockConstant"): -0.59587127}


# the second known answer of the same test: the heating-rate bound q(h, v, alpha) <= Qlimit at every state added
# (addUpperFuncBound("Path", QFunc(), [0, 2, 6], Qlimit, 1 / Qlimit), test_Reentry.py:99-109,206-222) -- a user path function
# through the plain-function kernels, as an inequality.  64-segment values of this build:
OWN_HEATING = {("LGL3", "HighestOrderSpline"): -0.53472810, ("LGL3", "BlockConstant"): -0.53382557,
               ("LGL5", "HighestOrderSpline"): -0.53454201, ("LGL5", "BlockConstant"): -0.53356540,
               ("LGL7", "HighestOrderSpline"): -0.53456086, ("LGL7", "BlockConstant"): -0.53347575,
               ("Trapezoidal", "HighestOrderSpline"): -0.53402632, ("Trapezoidal", "BlockConstant"): -0.53395162}


def _check(prob, x, lam, info, key, ref=REFERENCE_OBJECTIVE, own=OWN, tol_own=2e-6):
    assert info["feasible"] and info["converged"], info
    assert abs(info["objective"] - ref) < REFERENCE_TOLERANCE                            # the reference's own assertion
    assert abs(info["objective"] - own[key]) < tol_own, (info["objective"], own[key])    # and the solution is THE solution
    assert np.all(x >= prob["lb"] - 1e-9) and np.all(x <= prob["ub"] + 1e-9)


@pytest.mark.parametrize("mode,control", CASES)
def test_oracle_reproduces_the_reference_objective(oracle, mode, control):
    prob = kh.reentry_problem(mode, control, 64)
    prov = kh.OracleProvider(oracle, prob)
    x, lam, info = kh.solve_reentry(prov, prob)
    _check(prob, x, lam, info, (mode, control))
    assert np.abs(prov.con(x)).max() < 1e-7
    # second stage (test_Reentry.py:206-222): the heating-rate bound added, optimised again from the solution
    prob2 = kh.reentry_problem(mode, control, 64, heating=True)
    prov2 = kh.OracleProvider(oracle, prob2)
    x2, lam2, info2 = kh.solve_reentry(prov2, prob2, x0=x)
    _check(prob2, x2, lam2, info2, (mode, control), REFERENCE_OBJECTIVE_HEATING, OWN_HEATING, 5e-6)
    g = prov2.con(x2)[prob2["slack_rows"]]
    assert g.max() < 1e-7 and g.max() > -1e-3                  # the bound holds, and it is active somewhere


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    pk = os.path.join(ROOT, "asset_asrl_amd")
    so = str(tmp_path_factory.mktemp("shim") / "shim_driver.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_shim_driver.cpp"),
                           "-o", so, "-L" + pk, "-lasset_host", "-lasset_hip", "-Wl,-rpath," + pk])
    return C.CDLL(so)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,control", CASES)
def test_device_path_reproduces_the_reference_objective(oracle, shim, mode, control):
    prob = kh.reentry_problem(mode, control, 64)
    prov = kh.DeviceProvider(shim, prob)
    try:
        x, lam, info = kh.solve_reentry(prov, prob)
        _check(prob, x, lam, info, (mode, control))
        # at the solution the device and the oracle agree on the residuals and on the KKT blocks the last step used
        ref = kh.OracleProvider(oracle, prob)
        c_d, g_d, W_d, J_d = prov.kkt(x, lam)
        c_o, g_o, W_o, J_o = ref.kkt(x, lam)
        assert np.abs(c_d - c_o).max() < 1e-10 * max(1.0, np.abs(x).max())
        assert np.abs(g_d - g_o).max() < 1e-8 * max(1.0, np.abs(g_o).max())
        assert abs(W_d - W_o).max() < 1e-8 * max(1.0, abs(W_o).max()) and abs(J_d - J_o).max() < 1e-8 * max(1.0, abs(J_o).max())
    finally:
        prov.close()
    # second stage: the heating-rate bound (a user path function on the device), from the solution
    prob2 = kh.reentry_problem(mode, control, 64, heating=True)
    prov2 = kh.DeviceProvider(shim, prob2)
    try:
        x2, lam2, info2 = kh.solve_reentry(prov2, prob2, x0=x)
        _check(prob2, x2, lam2, info2, (mode, control), REFERENCE_OBJECTIVE_HEATING, OWN_HEATING, 5e-6)
    finally:
        prov2.close()
