"""The last known answer the REFERENCE holds for a problem of this path, reproduced through this build's evaluation path: the
Delta III ascent of /root/reference/asset_asrl/test/test_FullProblems/test_Delta3Launch.py:14-300 -- FOUR phases (the burns between
the jettison events), each with its own ODE object (7 states, 3 controls; thrust along u / |u|, drag in a rotating exponential
atmosphere), stage masses fixed at the front of each phase, the phases linked in position, velocity, time and thrust direction,
|u| in [0.5, 1.5] and |r| >= 0.999999 Re at every state (796 nonlinear inequalities), five orbit-insertion conditions at the end;
maximise the final mass.  Reference: 7529.749892668763 kg +- 1.0 for LGL3 / LGL5 / LGL7 (200 points) and Trapezoidal (500) x
{HighestOrderSpline, BlockConstant} (:161-162, :279-292).

The harness (tests/kkt_harness.py; not PSIOPT) solves LGL3 from the reference's straight-line initial guess and starts LGL5 / LGL7
from that solution on the same mesh (from the straight line its loop does not converge for them within 400 iterations -- a limit
of the harness, not of the assembly).  HighestOrderSpline: 7529.74911 (LGL3), 7529.74868 (LGL5), 7529.74853 (LGL7) kg -- 0.8-1.4 g
from the reference's thirteen digits (1e-7 relative); BlockConstant: 7529.264 kg in all three (0.49 kg below: the thrust
direction is constant over a segment), Trapezoidal x 500: 7530.248 kg.

* on the CPU from the oracle (oracle/fullnlp.cpp): seven of the eight cases (Trapezoidal x BlockConstant needs a 500-point LGL3
  start that the loop does not converge);
* on the GPU (-m gpu) from the device kernels through the C ABI and the C++ host shim's KktAssembly: LGL3 x HighestOrderSpline --
  four run-time compiled ODEs, eleven functions, 2 189 variables."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kkt_harness as kh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_OBJECTIVE, REFERENCE_TOLERANCE = 7529.749892668763, 1.0        # test_Delta3Launch.py:161, :279 (kg)
OWN = {("LGL3", "HighestOrderSpline"): 7529.74911, ("LGL5", "HighestOrderSpline"): 7529.74868, ("LGL7", "HighestOrderSpline"): 7529.74853,
       ("LGL3", "BlockConstant"): 7529.26406, ("LGL5", "BlockConstant"): 7529.26418, ("LGL7", "BlockConstant"): 7529.26367,
       ("Trapezoidal", "HighestOrderSpline"): 7530.24832}
SOLVER = dict(step_cap=1.0, relative_push=True, feasibility_first=True, mu=1e-2)


def _check(prob, x, info, key):
    mass = -info["objective"] * prob["objective_scale"]
    assert info["feasible"] and info["converged"], info
    assert abs(mass - REFERENCE_OBJECTIVE) < REFERENCE_TOLERANCE                               # the reference's own assertion
    assert abs(mass - OWN[key]) < 5e-3, (mass, OWN[key])                                       # and the solution is THE solution
    if key[1] == "HighestOrderSpline" and key[0] != "Trapezoidal":
        assert abs(mass - REFERENCE_OBJECTIVE) < 5e-3                                           # 7e-7 relative
    assert np.all(x >= prob["lb"] - 1e-9) and np.all(x <= prob["ub"] + 1e-9)


def _constraints_hold(prov, prob, x):
    c = prov.con(x)
    rows, (slo, shi) = prob["slack_rows"], prob["slack_bounds"]
    assert np.abs(np.delete(c, rows)).max() < 1e-7            # defects, spacing, control splines of four phases; the orbit conditions
    assert np.all(-c[rows] >= slo - 1e-7) and np.all(-c[rows] <= shi + 1e-7)   # -g(x) = s inside its bounds: the norm bounds hold
    A, b = prob["linear_rows"]
    assert np.abs(A @ x - b).max() < 1e-9                      # the links


@pytest.mark.parametrize("control", ["HighestOrderSpline", "BlockConstant"])
def test_oracle_reproduces_the_reference_objective(oracle, control):
    p3 = kh.delta3_problem("LGL3", control, 200)
    prov = kh.OracleProvider(oracle, p3)
    x3, _, info = kh.solve_linked(prov, p3, **SOLVER)
    _check(p3, x3, info, ("LGL3", control))
    _constraints_hold(prov, p3, x3)
    for mode in ("LGL5", "LGL7"):
        prob = kh.delta3_problem(mode, control, 200, warm=(p3, x3))
        prov = kh.OracleProvider(oracle, prob)
        x, _, info = kh.solve_linked(prov, prob, **dict(SOLVER, mu=1e-4))
        _check(prob, x, info, (mode, control))
        _constraints_hold(prov, prob, x)


def test_oracle_reproduces_the_reference_objective_trapezoidal(oracle):
    prob = kh.delta3_problem("Trapezoidal", "HighestOrderSpline", 500)
    prov = kh.OracleProvider(oracle, prob)
    x, _, info = kh.solve_linked(prov, prob, **SOLVER)
    _check(prob, x, info, ("Trapezoidal", "HighestOrderSpline"))
    _constraints_hold(prov, prob, x)


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    pk = os.path.join(ROOT, "asset_asrl_amd")
    so = str(tmp_path_factory.mktemp("shim") / "shim_driver.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_shim_driver.cpp"),
                           "-o", so, "-L" + pk, "-lasset_host", "-lasset_hip", "-Wl,-rpath," + pk])
    return C.CDLL(so)


@pytest.mark.gpu
def test_device_path_reproduces_the_reference_objective(oracle, shim):
    prob = kh.delta3_problem("LGL3", "HighestOrderSpline", 200)
    prov = kh.DeviceProvider(shim, prob)
    try:
        x, lam, info = kh.solve_linked(prov, prob, **SOLVER)
        _check(prob, x, info, ("LGL3", "HighestOrderSpline"))
        # at the solution the device and the oracle agree on the residuals and on the KKT blocks of all four phases
        ref = kh.OracleProvider(oracle, prob)
        lam_in = lam[:prov.m]
        c_d, g_d, W_d, J_d = prov.kkt(x, lam_in)
        c_o, g_o, W_o, J_o = ref.kkt(x, lam_in)
        assert np.abs(c_d - c_o).max() < 1e-10 * max(1.0, np.abs(x).max())
        assert np.abs(g_d - g_o).max() < 1e-8 * max(1.0, np.abs(g_o).max())
        assert abs(W_d - W_o).max() < 1e-8 * max(1.0, abs(W_o).max()) and abs(J_d - J_o).max() < 1e-8 * max(1.0, abs(J_o).max())
    finally:
        prov.close()
