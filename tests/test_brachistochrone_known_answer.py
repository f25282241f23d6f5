"""BASELINE.json configs[0] -- the reference's own CPU-runnable case, /root/reference/examples/Brachistochrone.py:15-67 (a bead from (0, 10) at
rest to (10, 5) under g = 9.81, theta in [-0.1, 2], minimum time on 32 LGL3 segments) -- solved through this build's evaluation path and
compared with the ANALYTIC answer: the cycloid's descent time 1.8012954830137 s.  The example holds no number; the cycloid does.

LGL5 and LGL7 on the example's 32 segments reproduce it to 1e-13 (LGL3: 1e-8; BlockConstant control and Trapezoidal: 2e-4, the error of a
piece-wise constant / second-order control on that mesh): defects, mesh spacing, control splines, their Jacobians and the Lagrangian
Hessian of the oracle -- and of the device kernels (-m gpu) -- reach the exact optimum of a problem with a closed-form solution."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kkt_harness as kh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(m, c) for m in ("LGL3", "LGL5", "LGL7", "Trapezoidal") for c in ("HighestOrderSpline", "BlockConstant")]


def _tolerance(mode, control):
    if control == "BlockConstant" or mode == "Trapezoidal":
        return 3e-4
    return 2e-8 if mode == "LGL3" else 2e-12


def _check(x, info, prob, mode, control):
    exact = kh.brachistochrone_exact()
    assert abs(exact - 1.8012954830137) < 1e-12
    assert info["converged"] and info["feasible"], info
    assert 0.0 <= info["objective"] - exact < _tolerance(mode, control) or abs(info["objective"] - exact) < 2e-12
    ix = prob["ix"]
    theta = np.array([x[ix.getXTUVarLoc(4, k)] for k in range(ix.numStates)])
    assert np.all(np.diff(theta) > -1e-9) and theta[0] < 0.05 and -0.1 < theta.min() and theta.max() < 2.0   # the cycloid's angle grows in time, inside the bounds


@pytest.mark.parametrize("mode,control", CASES)
def test_oracle_reaches_the_cycloid(oracle, mode, control):
    prob = kh.brachistochrone_problem(mode, control, 32)
    prov = kh.OracleProvider(oracle, prob)
    x, lam, info = kh.solve_optimize_only(prov, prob, step_cap=np.inf, tol=1e-10)
    _check(x, info, prob, mode, control)
    assert np.abs(prov.con(x)).max() < 1e-9


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    pk = os.path.join(ROOT, "asset_asrl_amd")
    so = str(tmp_path_factory.mktemp("shim") / "shim_driver.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_shim_driver.cpp"),
                           "-o", so, "-L" + pk, "-lasset_host", "-lasset_hip", "-Wl,-rpath," + pk])
    return C.CDLL(so)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,control", [("LGL3", "HighestOrderSpline"), ("LGL7", "HighestOrderSpline"), ("LGL5", "BlockConstant")])
def test_device_path_reaches_the_cycloid(shim, mode, control):
    prob = kh.brachistochrone_problem(mode, control, 32)
    prov = kh.DeviceProvider(shim, prob)
    try:
        x, lam, info = kh.solve_optimize_only(prov, prob, step_cap=np.inf, tol=1e-10)
        _check(x, info, prob, mode, control)
    finally:
        prov.close()
