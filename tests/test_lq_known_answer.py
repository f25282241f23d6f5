"""The known answer of the reference's SCALING test, reproduced through this build's evaluation path -- with the running cost once as
an integral objective and once as an INTEGRAL PARAMETER FUNCTION (round 5's Phase.addIntegralParamFunction: the accumulation -p and
the segment quadratures on one shared constraint row, ODEPhaseBase.cpp:835-889): /root/reference/asset_asrl/test/test_AutoScaling/
test_ObjScaling.py:11-141 -- x' = x / 2 + u, x(0) = 1, minimise pi * int_0^1 (u^2 + x u + 1.25 x^2) dt + e * x(1); the reference asserts the
final state 0.3185865574270634 +- 1e-3 for every formulation (:36-37, :218-236).  The problem is linear-quadratic, so the exact
answer is available independently: Pontryagin's two-point boundary-value problem (below, scipy) gives x(1) = 0.318567325 -- the
reference's recorded value is 1.9e-5 from it, inside its own tolerance.  (Its adaptive mesh and unit scaling are solver-side; here the
mesh is fixed and fine.)

* on the CPU from the oracle (oracle/fullnlp.cpp): both formulations, LGL3 / LGL5 / LGL7 / Trapezoidal, both control modes;
* on the GPU (-m gpu) from the device kernels through the C ABI and the C++ host shim's KktAssembly: both formulations."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kkt_harness as kh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_FINAL_STATE, REFERENCE_TOLERANCE = 0.3185865574270634, 1e-3     # test_ObjScaling.py:36-37
CASES = [("LGL3", "HighestOrderSpline", 64), ("LGL5", "HighestOrderSpline", 32), ("LGL7", "HighestOrderSpline", 20),
         ("LGL5", "BlockConstant", 64), ("LGL7", "BlockConstant", 64), ("Trapezoidal", "HighestOrderSpline", 256)]


def exact_final_state():
    """u = -(pi x + lam) / (2 pi);  lam' = -(pi (u + 2.5 x) + lam / 2);  x(0) = 1, lam(1) = e."""
    from scipy.integrate import solve_bvp

    def rhs(t, y):
        u = -(np.pi * y[0] + y[1]) / (2 * np.pi)
        return np.vstack([0.5 * y[0] + u, -(np.pi * (u + 2.5 * y[0]) + 0.5 * y[1])])
    t = np.linspace(0, 1, 200)
    sol = solve_bvp(rhs, lambda ya, yb: np.array([ya[0] - 1.0, yb[1] - np.e]), t, np.ones((2, t.size)), tol=1e-12, max_nodes=100000)
    assert sol.status == 0
    return float(sol.y[0, -1])


@pytest.mark.parametrize("mode,control,nseg", CASES)
def test_oracle_reproduces_the_reference_final_state(oracle, mode, control, nseg):
    exact = exact_final_state()
    assert abs(exact - 0.318567325) < 1e-8 and abs(exact - REFERENCE_FINAL_STATE) < REFERENCE_TOLERANCE
    out = []
    for integral_param in (False, True):
        prob = kh.lq_problem(mode, control, nseg, integral_param=integral_param)
        prov = kh.OracleProvider(oracle, prob)
        x, lam, info = kh.solve_optimize_only(prov, prob, step_cap=np.inf, tol=1e-10)
        assert info["converged"] and info["feasible"], info
        xf = x[prob["final_state"]]
        assert abs(xf - REFERENCE_FINAL_STATE) < REFERENCE_TOLERANCE          # the reference's own assertion
        assert abs(xf - exact) < (5e-6 if mode in ("LGL3", "Trapezoidal") or control == "BlockConstant" else 1e-8)
        if integral_param:       # the static parameter IS the integral: the shared row holds
            ix = prob["ix"]
            p = x[ix.var_offset + ix.StaticParamLoc0]
            assert abs(np.pi * p + np.e * xf - info["objective"]) < 1e-12 and np.abs(prov.con(x)).max() < 1e-9
        out.append((xf, info["objective"]))
    # the two formulations are the same problem: same optimum, same cost
    assert abs(out[0][0] - out[1][0]) < 1e-8 and abs(out[0][1] - out[1][1]) < 1e-8


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    pk = os.path.join(ROOT, "asset_asrl_amd")
    so = str(tmp_path_factory.mktemp("shim") / "shim_driver.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_shim_driver.cpp"),
                           "-o", so, "-L" + pk, "-lasset_host", "-lasset_hip", "-Wl,-rpath," + pk])
    return C.CDLL(so)


@pytest.mark.gpu
@pytest.mark.parametrize("integral_param", [False, True])
def test_device_path_reproduces_the_reference_final_state(oracle, shim, integral_param):
    prob = kh.lq_problem("LGL7", "HighestOrderSpline", 20, integral_param=integral_param)
    prov = kh.DeviceProvider(shim, prob)
    try:
        x, lam, info = kh.solve_optimize_only(prov, prob, step_cap=np.inf, tol=1e-10)
        assert info["converged"] and info["feasible"], info
        xf = x[prob["final_state"]]
        assert abs(xf - REFERENCE_FINAL_STATE) < REFERENCE_TOLERANCE and abs(xf - 0.318567325) < 1e-8
        ref = kh.OracleProvider(oracle, prob)
        c_d, g_d, W_d, J_d = prov.kkt(x, lam)
        c_o, g_o, W_o, J_o = ref.kkt(x, lam)
        assert np.abs(c_d - c_o).max() < 1e-10 and np.abs(g_d - g_o).max() < 1e-8 * max(1.0, np.abs(g_o).max())
        assert abs(W_d - W_o).max() < 1e-8 * max(1.0, abs(W_o).max()) and abs(J_d - J_o).max() < 1e-8 * max(1.0, abs(J_o).max())
    finally:
        prov.close()
