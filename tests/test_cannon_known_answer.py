"""A fourth known answer the REFERENCE holds, and its first MULTI-PHASE one, reproduced through this build's evaluation path: the
cannon ball of /root/reference/asset_asrl/test/test_FullProblems/test_MultiPhaseCannon.py:43-175 -- maximum range for a bounded
muzzle energy, drag in an exponential atmosphere, the ball's radius free; TWO phases of one ODE (ascent to the apex, descent to
the ground) joined by a forward link of the state and a direct link of the radius.  Reference: range -3280.2039356471037 m +- 1.0
at 16 segments per phase (LGL3 / LGL5 / LGL7) and 128 (Trapezoidal) (:83-87, :166-167).

What it exercises that the other known answers do not: an ODE PARAMETER (the radius: p = 1, the parameter columns of the defect
blocks and their node-summed Hessian entries), a second phase whose variables and rows start behind the first's
(Phase.layout(Vstart, Estart) = transcribe_phase's offsets), a user function of a state AND a parameter over the "Front" region
as an inequality.  The links are linear and live in the harness (tests/kkt_harness.py: LinearRows), like the boundary values.

* on the CPU from the oracle (oracle/fullnlp.cpp): all four transcriptions -- the LGL5 and LGL7 solutions are 4e-4 m from the
  reference's thirteen digits (1e-7 relative), LGL3 8e-3 m, Trapezoidal 0.29 m: the reference's number is its high-order value;
* on the GPU (-m gpu) from the device kernels through the C ABI and the C++ host shim's KktAssembly: two transcriptions."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kkt_harness as kh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_OBJECTIVE, REFERENCE_TOLERANCE = -3280.2039356471037, 1.0      # test_MultiPhaseCannon.py:84-85 (metres)
NSEG = {"LGL3": 16, "LGL5": 16, "LGL7": 16, "Trapezoidal": 128}          # :167
OWN = {"LGL3": -3280.21210, "LGL5": -3280.20354, "LGL7": -3280.20358, "Trapezoidal": -3279.90984}   # this build (oracle run)
SOLVER = dict(step_cap=5.0, relative_push=True, feasibility_first=True, mu=1e-3, tol=1e-10)


def _check(prob, x, info, mode):
    rng = info["objective"] * prob["objective_scale"]
    assert info["feasible"] and info["converged"], info
    assert abs(rng - REFERENCE_OBJECTIVE) < REFERENCE_TOLERANCE                               # the reference's own assertion
    assert abs(rng - OWN[mode]) < 2e-3, (rng, OWN[mode])                                      # and the solution is THE solution
    if mode in ("LGL5", "LGL7"):
        assert abs(rng - REFERENCE_OBJECTIVE) < 2e-3                                           # 6e-7 relative
    assert np.all(x >= prob["lb"] - 1e-9) and np.all(x <= prob["ub"] + 1e-9)


@pytest.mark.parametrize("mode", list(NSEG))
def test_oracle_reproduces_the_reference_objective(oracle, mode):
    prob = kh.cannon_problem(mode, NSEG[mode])
    prov = kh.OracleProvider(oracle, prob)
    x, lam, info = kh.solve_linked(prov, prob, **SOLVER)
    _check(prob, x, info, mode)
    c = prov.con(x)
    rows = prob["slack_rows"]
    assert np.abs(np.delete(c, rows)).max() < 1e-7            # defects and mesh spacing of both phases
    assert -1e-5 < c[rows].max() < 1e-7                        # the muzzle-energy bound holds, and it is active
    A, b = prob["linear_rows"]
    assert np.abs(A @ x - b).max() < 1e-9                      # the links: the descent starts where the ascent ends, one radius
    ia = prob["parts"][0]["ix"]
    assert abs(x[ia.getXTUVarLoc(1, ia.numStates - 1)]) < 1e-12   # the ascent ends at the apex


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    pk = os.path.join(ROOT, "asset_asrl_amd")
    so = str(tmp_path_factory.mktemp("shim") / "shim_driver.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_shim_driver.cpp"),
                           "-o", so, "-L" + pk, "-lasset_host", "-lasset_hip", "-Wl,-rpath," + pk])
    return C.CDLL(so)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["LGL5", "Trapezoidal"])
def test_device_path_reproduces_the_reference_objective(oracle, shim, mode):
    prob = kh.cannon_problem(mode, NSEG[mode])
    prov = kh.DeviceProvider(shim, prob)
    try:
        x, lam, info = kh.solve_linked(prov, prob, **SOLVER)
        _check(prob, x, info, mode)
        # at the solution the device and the oracle agree on the residuals and on the KKT blocks of BOTH phases
        ref = kh.OracleProvider(oracle, prob)
        lam_in = lam[:prov.m]
        c_d, g_d, W_d, J_d = prov.kkt(x, lam_in)
        c_o, g_o, W_o, J_o = ref.kkt(x, lam_in)
        assert np.abs(c_d - c_o).max() < 1e-10 * max(1.0, np.abs(x).max())
        assert np.abs(g_d - g_o).max() < 1e-8 * max(1.0, np.abs(g_o).max())
        assert abs(W_d - W_o).max() < 1e-8 * max(1.0, abs(W_o).max()) and abs(J_d - J_o).max() < 1e-8 * max(1.0, abs(J_o).max())
    finally:
        prov.close()
