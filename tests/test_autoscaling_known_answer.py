"""AutoScaling end to end, on the problem the reference tests it with: /root/reference/asset_asrl/test/
test_AutoScaling/test_Reentry.py:14-226 -- the shuttle re-entry stated in FEET AND SECONDS, the phase told the units
(setUnits(h = 1e5 ft, v = 1e5 ft / min, t = 1 min), setAutoScaling(True)) and asked for the same two answers as the unscaled
test: cross-range 0.5958800738629952 and, with the heating-rate bound, 0.534620087611498, +- 1e-2 (:123-127).

With AutoScaling the phase evaluates IOScaled(dimensional ODE) and wraps user functions the same way (Phase.setUnits,
Phase._scaled_func; ODEPhase.h:87-109, :293-326).  That composition is mathematically the non-dimensional `reentry` the oracle
holds -- an independent statement of the same dynamics -- so the oracle's non-dimensional problem is the checker here:

* CPU: the scaled functions against the oracle's non-dimensional ones at random points (host evaluation of the expressions), and
  the scaled problem's start, bounds and tables against the non-dimensional problem's;
* GPU (-m gpu): the dimensional problem through the device path (run-time compiled scaled ODE, scaled path function, C ABI,
  KktAssembly) reaches both objectives, and at the solutions agrees with the oracle's non-dimensional assembly block for block."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kkt_harness as kh
from test_reentry_known_answer import OWN, OWN_HEATING, REFERENCE_OBJECTIVE, REFERENCE_OBJECTIVE_HEATING, REFERENCE_TOLERANCE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_value(ode, y, nout):
    dp = C.POINTER(C.c_double)
    out = np.zeros(nout)
    C.CFUNCTYPE(None, dp, dp, C.c_void_p)(ode.f)(np.ascontiguousarray(y).ctypes.data_as(dp), out.ctypes.data_as(dp), ode.ctx)
    return out


def test_scaled_dimensional_functions_equal_the_nondimensional_ones(oracle):
    prob = kh.reentry_problem("LGL7", "HighestOrderSpline", 8, heating=True, autoscaled=True)
    ph = prob["phase"]
    assert ph.AutoScaling and ph._active_ode().ode_name == "reentry_feet_scaled"
    F, Q = ph._active_ode().vf(), [e[2] for e in prob["entries"] if e[1] == "eq0"][0]
    ode, heat = oracle.get_ode("reentry", 0), oracle.get_ode("reentry_heating", 0)
    rng = np.random.default_rng(5)
    for _ in range(40):      # scaled variables: h / 1e5 ft, theta, v / (1e5 ft / min), gamma, psi, t / min, alpha, beta
        y = np.array([rng.uniform(0.8, 2.6), rng.uniform(-0.5, 0.5), rng.uniform(3, 15), rng.uniform(-0.1, 0.05), rng.uniform(0, 1.6),
                      rng.uniform(0, 30), rng.uniform(0, 0.4), rng.uniform(-1.2, 0.01)])
        ref = _oracle_value(ode, y, 5)
        assert np.abs(F.compute(y) - ref).max() < 1e-13 * max(1.0, np.abs(ref).max())
        q = np.array([y[0], y[2], y[6]])
        assert abs(Q.compute(q)[0] - _oracle_value(heat, q, 1)[0]) < 1e-12 * max(1.0, abs(_oracle_value(heat, q, 1)[0]))
    # the scaled problem IS the non-dimensional problem: start, bounds, cost, tables
    nd = kh.reentry_problem("LGL7", "HighestOrderSpline", 8, heating=True)
    assert np.abs(prob["x0"] - nd["x0"]).max() < 1e-13
    for k in ("lb", "ub", "cost"):
        np.testing.assert_array_equal(np.isfinite(prob[k]), np.isfinite(nd[k]))
        fin = np.isfinite(nd[k])
        assert np.abs(prob[k][fin] - nd[k][fin]).max() < 1e-13
    np.testing.assert_array_equal(prob["V"], nd["V"])
    np.testing.assert_array_equal(prob["slack_rows"], nd["slack_rows"])


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    pk = os.path.join(ROOT, "asset_asrl_amd")
    so = str(tmp_path_factory.mktemp("shim") / "shim_driver.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_shim_driver.cpp"),
                           "-o", so, "-L" + pk, "-lasset_host", "-lasset_hip", "-Wl,-rpath," + pk])
    return C.CDLL(so)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,control", [("LGL7", "HighestOrderSpline"), ("LGL3", "BlockConstant")])
def test_device_path_solves_the_dimensional_problem_with_autoscaling(oracle, shim, mode, control):
    xs = None
    for heating, ref, own, tol in ((False, REFERENCE_OBJECTIVE, OWN, 2e-6), (True, REFERENCE_OBJECTIVE_HEATING, OWN_HEATING, 5e-6)):
        prob = kh.reentry_problem(mode, control, 64, heating=heating, autoscaled=True)
        prov = kh.DeviceProvider(shim, prob)
        try:
            x, lam, info = kh.solve_reentry(prov, prob, x0=xs)
            assert info["feasible"] and info["converged"], info
            assert abs(info["objective"] - ref) < REFERENCE_TOLERANCE                    # the reference's own assertion
            assert abs(info["objective"] - own[(mode, control)]) < tol                     # the unscaled problem's solution
            # the oracle's NON-DIMENSIONAL assembly at the same point
            chk = kh.OracleProvider(oracle, kh.reentry_problem(mode, control, 64, heating=heating))
            lam_in = lam[:prov.m]
            c_d, g_d, W_d, J_d = prov.kkt(x, lam_in)
            c_o, g_o, W_o, J_o = chk.kkt(x, lam_in)
            assert np.abs(c_d - c_o).max() < 1e-10 * max(1.0, np.abs(x).max())
            assert np.abs(g_d - g_o).max() < 1e-8 * max(1.0, np.abs(g_o).max())
            assert abs(W_d - W_o).max() < 1e-8 * max(1.0, abs(W_o).max()) and abs(J_d - J_o).max() < 1e-8 * max(1.0, abs(J_o).max())
        finally:
            prov.close()
        xs = x
