"""GPU tests of the Python-visible Phase surface: ``ode.phase(...)``, ``get_defect().computeall`` and the reference's
own self-consistency recipe applied to the device results."""
import numpy as np
import pytest

from asset_asrl_amd import ode as odelib
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _brach_traj():
    g, theta0, tf = 9.81, 1.0, 1.0
    rows = []
    for t in np.linspace(0, tf, 100):        # examples/Brachistochrone.py:52-60
        rows.append([10 * t / tf, 10 - 5 * t / tf, g * t * np.cos(theta0) + 0.1, t, theta0])
    return rows


def test_phase_get_defect_computeall(oracle):
    phase = odelib.Brachistochrone(9.81).phase("LGL3", _brach_traj(), 40)     # BASELINE config 0
    F = phase.get_defect()
    assert (F.IRows(), F.ORows()) == (10, 3)
    X = phase.solver_input()
    ev = phase.evaluator
    x = X[ev.vindex[7]]
    lam = np.array([0.3, -1.2, 2.0])
    fx, jx, gx, hx = F.computeall(x, lam)
    rfx, rjx, rgx, rhx = oracle.defect_all(oracle.get_ode("brachistochrone", 0), oracle.LGL3, x, lam)
    assert np.abs(fx - rfx).max() < 1e-10 and rel_err(jx, rjx) < 1e-8 and rel_err(gx, rgx) < 1e-8 and rel_err(hx, rhx) < 1e-8
    # reference recipe (test_VectorFunctions/__init__.py:40-67)
    assert np.abs(jx.T @ lam - gx).max() < 1e-12 * max(1, np.abs(gx).max())
    np.testing.assert_allclose(F.compute(x), fx, atol=1e-14)
    np.testing.assert_allclose(F.jacobian(x), jx, atol=1e-13)
    np.testing.assert_allclose(F.adjointgradient(x, lam), gx, atol=1e-12)
    np.testing.assert_allclose(F.adjointhessian(x, lam), hx, atol=1e-12)
    eps = 1e-6
    jfd = np.array([(F.compute(x + eps * e) - F.compute(x - eps * e)) / (2 * eps) for e in np.eye(10)]).T
    assert np.abs(jfd - jx).max() < 1e-4
    with pytest.raises(ValueError):
        F.compute(np.zeros(9))
    res = phase.test_threads(1, 8, 5, verbose=False)
    assert res["segments"] == 40 and res["evalKKT_ms"] > 0


def test_phase_modes_and_errors():
    ode = odelib.TwoBody(1.0, 0.01)
    traj = np.zeros((50, 10))
    traj[:, 0] = 1.0
    traj[:, 4] = 1.0
    traj[:, 6] = np.linspace(0, 3, 50)
    traj[:, 7:] = 0.01
    ph = ode.phase("LGL5")
    ph.setTraj(traj, 75)
    ph.setControlMode("BlockConstant")
    ph.transcribe()
    assert ph.evaluator.IR == 24 and ph.evaluator.OR == 12          # SURVEY section 8 config 4 sizes
    assert ph.solver_input().size == 151 * 7 + 75 * 3
    with pytest.raises(ValueError):
        ph.switchTranscriptionMode("LGL9")
    with pytest.raises(ValueError):
        ph.setControlMode("Bogus")
    bad = traj.copy()
    bad[3, 2] = np.nan
    with pytest.raises(ValueError):
        ph.setTraj(bad, 10)


RECIPE_CASES = [("brachistochrone", "LGL3", False), ("reentry", "LGL7", False), ("reentry", "Trapezoidal", False),
                ("twobody_lt", "LGL5", True), ("betts_lowthrust", "LGL5", False), ("betts_lowthrust", "LGL3", True),
                ("synthetic32", "LGL7", False)]


@pytest.mark.parametrize("ode,mode,blocked", RECIPE_CASES)
def test_reference_self_consistency_recipe_on_the_device(ode, mode, blocked):
    """The reference's own derivative test (asset_asrl/test/test_VectorFunctions/__init__.py:40-67: ``jx^T L == gx`` to 1e-12,
    analytic Jacobian against central differences of ``compute`` to 1e-4, analytic Hessian against the symmetrised differences of
    ``adjointgradient`` to 1e-4) -- which its suite never applies to the defects -- applied to the DEVICE's defect function
    (``phase.get_defect()``: the resident, unit and row kernels behind one segment) for every kernel family.  No oracle involved."""
    from asset_asrl_amd.phase import DefectFunction
    from helpers import Workload
    w = Workload(ode, mode, 6, blocked)
    F = DefectFunction(ode, mode, w.blocked)
    x = w.X[w.vindex[3]]
    lam = w.L[w.cindex[3]] / 100.0
    fx, jx, gx, hx = F.computeall(x, lam)
    assert np.abs(jx.T @ lam - gx).max() < 1e-12 * max(1.0, np.abs(gx).max())     # "Adjoint gradients do not match"
    assert np.abs(hx - hx.T).max() < 1e-13 * max(1.0, np.abs(hx).max())
    eps = 1e-6
    jfd, hfd = np.zeros_like(jx), np.zeros_like(hx)
    for i in range(x.size):
        e = np.zeros(x.size)
        e[i] = eps
        jfd[:, i] = (F.compute(x + e) - F.compute(x - e)) / (2 * eps)
        hfd[:, i] = (F.adjointgradient(x + e, lam) - F.adjointgradient(x - e, lam)) / (2 * eps)
    assert np.abs(jx - jfd).max() < 1e-4 * max(1.0, np.abs(jx).max())
    assert np.abs(hx - 0.5 * (hfd + hfd.T)).max() < 1e-4 * max(1.0, np.abs(hx).max())
