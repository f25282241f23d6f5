"""CPU tests: the oracle against the golden vectors, the reference's own self-consistency recipe and
known answers.  (the reference pins this path only through J^T L == adjgrad to
1e-12 and finite-difference agreement to 1e-4 -- asset_asrl/test/test_VectorFunctions/__init__.py:40-67.)"""
import glob
import os

import numpy as np
import pytest

from helpers import Workload, rel_err

GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if os.path.basename(p) != "pathfuncs.npz")   # (the defect vectors; pathfuncs.npz: test_pathfuncs_oracle.py)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_matches_golden(oracle, path):
    g = np.load(path)
    ode = oracle.get_ode(str(g["ode"]), 0)
    mode, blocked = oracle.MODES[str(g["mode"])], bool(g["blocked"])
    for s in range(g["x"].shape[0]):
        fx, jx, gx, hx = oracle.defect_all(ode, mode, g["x"][s], g["lam"][s], blocked)
        scale = max(1.0, np.abs(g["x"][s]).max())
        assert np.abs(fx - g["fx"][s]).max() / scale < 1e-13        # residuals: parity bar is 1e-10
        assert rel_err(jx, g["jx"][s]) < 1e-12                       # derivatives: parity bar is 1e-8
        assert rel_err(gx, g["gx"][s]) < 1e-12
        assert rel_err(hx, g["hx"][s]) < 1e-12
        np.testing.assert_array_equal(fx, oracle.defect_compute(ode, mode, g["x"][s], blocked))
        f2, j2 = oracle.defect_jacobian(ode, mode, g["x"][s], blocked)
        np.testing.assert_array_equal(fx, f2)
        np.testing.assert_array_equal(jx, j2)


CASES = [("brachistochrone", "LGL3", False), ("reentry", "LGL7", False), ("reentry", "Trapezoidal", False),
         ("twobody_lt", "LGL5", True), ("betts_lowthrust", "LGL5", False), ("betts_lowthrust", "LGL3", True)]


@pytest.mark.parametrize("ode,mode,blocked", CASES)
def test_reference_self_consistency_recipe(oracle, ode, mode, blocked):
    w = Workload(ode, mode, 6, blocked)
    o = oracle.get_ode(ode, 0)
    m = oracle.MODES[mode]
    x = w.X[w.vindex[3]]
    lam = w.L[w.cindex[3]] / 100.0
    fx, jx, gx, hx = oracle.defect_all(o, m, x, lam, w.blocked)
    assert np.abs(jx.T @ lam - gx).max() < 1e-12 * max(1.0, np.abs(gx).max())   # "Adjoint gradients do not match"
    assert np.abs(hx - hx.T).max() < 1e-13 * max(1.0, np.abs(hx).max())
    eps = 1e-6
    jfd = np.zeros_like(jx)
    hfd = np.zeros_like(hx)
    for i in range(x.size):
        e = np.zeros(x.size)
        e[i] = eps
        jfd[:, i] = (oracle.defect_compute(o, m, x + e, w.blocked) - oracle.defect_compute(o, m, x - e, w.blocked)) / (2 * eps)
        hfd[:, i] = (oracle.defect_jacobian(o, m, x + e, w.blocked)[1].T @ lam
                     - oracle.defect_jacobian(o, m, x - e, w.blocked)[1].T @ lam) / (2 * eps)
    assert np.abs(jx - jfd).max() < 1e-4 * max(1.0, np.abs(jx).max())
    assert np.abs(hx - 0.5 * (hfd + hfd.T)).max() < 1e-4 * max(1.0, np.abs(hx).max())


def test_coefficient_identities(oracle):
    for cs in (2, 3, 4):
        A, U, C = (oracle.lgl_table(cs, k) for k in "AUC")
        assert np.abs(A.sum(1) - 1).max() < 5e-15
        assert np.abs(U.sum(1) - 1).max() < 5e-15
        assert np.abs(C.sum(1)).max() < 5e-15


@pytest.mark.parametrize("cs,order", [(2, 3), (3, 5), (4, 7)])
def test_polynomial_exactness(oracle, cs, order):
    """Defect of x(t)=t^k, xdot=k t^(k-1) vanishes for k<=order: use the value formula with
    f supplied by the polynomial, i.e. check sum_j(C x_j + h D f_j) + h E f(tau_i) through the tables."""
    s, A, B, C, D, E = (oracle.lgl_table(cs, k) for k in "sABCDE")
    tc = oracle.lgl_table(cs, "tc")
    t0, h = 0.3, 0.8
    for k in range(1, order + 1):
        x = (t0 + h * tc) ** k
        f = k * (t0 + h * tc) ** (k - 1)
        for i in range(cs - 1):
            xi = A[i] @ x + h * (B[i] @ f)
            ti = t0 + h * s[i]
            assert abs(xi - ti ** k) < 2e-14          # interpolation is exact
            d = C[i] @ x + h * (D[i] @ f) + h * E[i] * (k * ti ** (k - 1))
            assert abs(d) < 2e-14
    x = (t0 + h * tc) ** (order + 2)
    f = (order + 2) * (t0 + h * tc) ** (order + 1)
    d = C[0] @ x + h * (D[0] @ f) + h * E[0] * ((order + 2) * (t0 + h * s[0]) ** (order + 1))
    assert abs(d) > 1e-7


def test_generated_provider_matches_ad2(oracle):
    rng = np.random.default_rng(5)
    for ode, mode in [("reentry", "LGL7"), ("betts_lowthrust", "LGL5"), ("twobody_lt", "LGL5"),
                      ("brachistochrone", "LGL3"), ("synthetic32", "LGL3")]:
        try:
            o1 = oracle.get_ode(ode, 1)
        except KeyError:
            pytest.skip("oracle/gen/odes_gen.c not linked")
        o0 = oracle.get_ode(ode, 0)
        w = Workload(ode, mode, 4)
        x, lam = w.X[w.vindex[1]], rng.uniform(-1, 1, w.OR)
        a = oracle.defect_all(o0, oracle.MODES[mode], x, lam)
        b = oracle.defect_all(o1, oracle.MODES[mode], x, lam)
        for u, v in zip(a, b):
            assert rel_err(u, v) < 1e-12


@pytest.mark.parametrize("name,mode,nseg,blocked", [("reentry", "LGL7", 203, False), ("twobody_lt", "LGL5", 101, True),
                                                   ("betts_lowthrust", "LGL5", 50, False),
                                                   ("brachistochrone", "LGL3", 42, False)])
def test_four_segment_batches_match_the_scalar_loop(oracle, name, mode, nseg, blocked):
    """bench.py's cpu_baseline leg runs the oracle four segments at a time (oracle/batch4.h, the reference's SuperScalar
    loop, DenseFunctionBase.h:1318-1380): same blocks, same scattered KKT values as one segment at a time -- including a
    remainder that is not a multiple of four, split over three threads."""
    from helpers import Workload
    w = Workload(name, mode, nseg, blocked)
    ode = oracle.get_ode(name, 1)                      # generated analytic derivatives: the provider that has four-wide bodies
    out = []
    for b4 in (False, True):
        nlp = oracle.Nlp(ode, oracle.MODES[mode], blocked, w.vindex, w.cindex, w.n_primal, w.n_equal, 3)
        if b4:
            assert nlp.set_batch4(True)
        out.append(list(nlp.eval(oracle.JAC_ADJGRAD_HESS, w.X, w.L)) + list(nlp.eval_blocks(oracle.JAC_ADJGRAD_HESS, w.X, w.L)))
    for a, b in zip(*out):
        np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(a).max()))
    # the AD2 provider has no four-wide twin: the switch reports it and the scalar loop stays
    nlp = oracle.Nlp(oracle.get_ode(name, 0), oracle.MODES[mode], blocked, w.vindex, w.cindex, w.n_primal, w.n_equal, 1)
    assert not nlp.set_batch4(True)
