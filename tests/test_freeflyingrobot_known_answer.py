"""A third known answer the REFERENCE holds, reproduced through this build's evaluation path: the free-flying robot of
/root/reference/asset_asrl/test/test_FullProblems/test_FreeFlyingRobot.py:14-100 -- four thrusters in [0, 1], minimum integral of
their sum (a bang-bang solution), objective 7.9115 +- 0.01 for LGL3 / LGL5 / LGL7 / Trapezoidal x {HighestOrderSpline, BlockConstant}
at 256 segments (:47-49, :93-100).  A (6, 4, 0) ODE that is none of the BASELINE workloads, a linear integral objective, 1 028
bounded controls most of which end on a bound; everything nonlinear comes out of the assembly under test (tests/kkt_harness.py) --

* on the CPU from the oracle (oracle/fullnlp.cpp): all eight cases;
* on the GPU (-m gpu) from the device kernels through the C ABI and the C++ host shim's KktAssembly: three cases."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kkt_harness as kh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_OBJECTIVE, REFERENCE_TOLERANCE = 7.9115, 0.01                  # test_FreeFlyingRobot.py:47-48
CASES = [(m, c) for m in ("LGL3", "LGL5", "LGL7", "Trapezoidal") for c in ("HighestOrderSpline", "BlockConstant")]
# this build's own 256-segment solutions (recorded from the oracle run).  The three BlockConstant LGL cases agree to 1e-9: with
# the thrusters constant over a segment the switching structure is the same and the states are integrated to rounding
OWN = {("LGL3", "HighestOrderSpline"): 7.91218920, ("LGL3", "BlockConstant"): 7.91051531,
       ("LGL5", "HighestOrderSpline"): 7.91159263, ("LGL5", "BlockConstant"): 7.91051531,
       ("LGL7", "HighestOrderSpline"): 7.91253017, ("LGL7", "BlockConstant"): 7.91051531,
       ("Trapezoidal", "HighestOrderSpline"): 7.91209567, ("Trapezoidal", "BlockConstant"): 7.91076321}
DEVICE_CASES = [("LGL3", "BlockConstant"), ("LGL7", "HighestOrderSpline"), ("Trapezoidal", "HighestOrderSpline")]


def _check(prob, x, info, key):
    assert info["feasible"] and info["converged"], info
    assert abs(info["objective"] - REFERENCE_OBJECTIVE) < REFERENCE_TOLERANCE              # the reference's own assertion
    assert abs(info["objective"] - OWN[key]) < 5e-6, (info["objective"], OWN[key])         # and the solution is THE solution
    assert np.all(x >= prob["lb"] - 1e-9) and np.all(x <= prob["ub"] + 1e-9)


@pytest.mark.parametrize("mode,control", CASES)
def test_oracle_reproduces_the_reference_objective(oracle, mode, control):
    prob = kh.freeflyingrobot_problem(mode, control, 256)
    prov = kh.OracleProvider(oracle, prob)
    x, lam, info = kh.solve_optimize_only(prov, prob, step_cap=np.inf)
    _check(prob, x, info, (mode, control))
    assert np.abs(prov.con(x)).max() < 1e-7
    # bang-bang: nearly every thruster value sits on one of its bounds
    free = (prob["ub"] - prob["lb"] == 1.0)
    on_bound = np.minimum(x[free] - 0.0, 1.0 - x[free]) < 1e-3
    assert on_bound.mean() > 0.9


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    pk = os.path.join(ROOT, "asset_asrl_amd")
    so = str(tmp_path_factory.mktemp("shim") / "shim_driver.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_shim_driver.cpp"),
                           "-o", so, "-L" + pk, "-lasset_host", "-lasset_hip", "-Wl,-rpath," + pk])
    return C.CDLL(so)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,control", DEVICE_CASES)
def test_device_path_reproduces_the_reference_objective(oracle, shim, mode, control):
    prob = kh.freeflyingrobot_problem(mode, control, 256)
    prov = kh.DeviceProvider(shim, prob)
    try:
        x, lam, info = kh.solve_optimize_only(prov, prob, step_cap=np.inf)
        _check(prob, x, info, (mode, control))
        ref = kh.OracleProvider(oracle, prob)
        c_d, g_d, W_d, J_d = prov.kkt(x, lam)
        c_o, g_o, W_o, J_o = ref.kkt(x, lam)
        assert np.abs(c_d - c_o).max() < 1e-10 * max(1.0, np.abs(x).max())
        assert np.abs(g_d - g_o).max() < 1e-8 * max(1.0, np.abs(g_o).max())
        assert np.abs(prov.objective_gradient() - ref.objective_gradient()).max() < 1e-8 * max(1.0, np.abs(ref.objective_gradient()).max())
        assert abs(prov.objective(x) - ref.objective(x)) < 1e-10 * abs(ref.objective(x))
        assert abs(W_d - W_o).max() < 1e-8 * max(1.0, abs(W_o).max()) and abs(J_d - J_o).max() < 1e-8 * max(1.0, abs(J_o).max())
    finally:
        prov.close()
