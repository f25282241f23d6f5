"""De Boor mesh-error estimate: the oracle's restatement of
ODEPhase<DODE>::get_meshinfo_deboor against a known answer (CPU), and the device estimator against the oracle (GPU)."""
import numpy as np
import pytest

from asset_asrl_amd import synth
from helpers import make_vanderpol


def test_oracle_error_scaling_known_answer(oracle):
    """Known answer for the oracle: on an exact solution the estimated local error falls like h^(Order+1).  The forced
    Van der Pol oscillator with mu = 0, u = 0 is x0' = x1, x1' = -x0 with the solution (sin t, cos t); halving the
    mesh must divide the largest estimate by 2^(Order+1)."""
    ode = oracle.get_ode("vanderpol", 0)

    def errors(mode, nseg):
        cs = synth.MODE_CS[mode]
        K = cs - 1
        tc = synth._TC[cs]
        edges = np.linspace(0.0, 3.0, nseg + 1)
        t = np.concatenate([edges[:-1, None] + np.asarray(tc[:K])[None, :] * np.diff(edges)[:, None]]).ravel()
        t = np.append(t, 3.0)
        traj = np.column_stack([np.sin(t), np.cos(t), t, np.zeros_like(t), np.zeros_like(t)])   # [x0,x1,t,u,mu=0]
        _, err, _ = oracle.mesh_error_deboor(ode, oracle.MODES[mode], traj)
        return np.abs(err).max()

    for mode, order in (("LGL3", 3), ("LGL5", 5), ("Trapezoidal", 2)):
        e1, e2 = errors(mode, 40), errors(mode, 80)
        rate = np.log2(e1 / e2)
        assert abs(rate - (order + 1)) < 0.35, (mode, e1, e2, rate)      # local error O(h^(Order+1))


@pytest.mark.gpu
@pytest.mark.parametrize("ode,mode,nseg,blocked", [("reentry", "LGL7", 257, False), ("reentry", "LGL3", 64, False),
                                                   ("twobody_lt", "LGL5", 75, True), ("betts_lowthrust", "LGL5", 33, False),
                                                   ("brachistochrone", "Trapezoidal", 40, False),
                                                   ("twobody_lt", "Trapezoidal", 21, True)])
def test_device_estimator_matches_oracle(oracle, ode, mode, nseg, blocked):
    from asset_asrl_amd import mesh
    traj = synth.make_traj(ode, mode, nseg)
    tsnd, err, dist, emax, dmax = mesh.mesh_error_deboor(ode, mode, traj, blocked)
    rt, rerr, rdist = oracle.mesh_error_deboor(oracle.get_ode(ode, 0), oracle.MODES[mode], traj, blocked)
    np.testing.assert_allclose(tsnd, rt, rtol=0, atol=1e-14)
    # y_i divides O(1) node data by h^Order (1e13 for LGL7 at h = 0.04): compare relative to each column's size
    for got, ref in ((err, rerr), (dist, rdist)):
        assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()
    np.testing.assert_allclose(emax, np.abs(rerr).max(axis=0), rtol=1e-9)
    np.testing.assert_allclose(dmax, np.abs(rdist).max(axis=0), rtol=1e-9)


@pytest.mark.gpu
def test_phase_mesh_info_with_user_ode(oracle):
    ode = make_vanderpol()
    traj = synth.make_traj("vanderpol", "LGL5", 30, sizes=(2, 1, 1))
    ph = ode.phase("LGL5", traj, 30)
    tsnd, err, dist = ph.get_meshinfo_deboor()
    rt, rerr, rdist = oracle.mesh_error_deboor(oracle.get_ode("vanderpol", 0), oracle.MODES["LGL5"], ph.ActiveTraj)
    assert np.abs(err - rerr).max() <= 1e-9 * np.abs(rerr).max()
    t2, bins, error = ph.getMeshInfo(False, 12)
    assert bins.shape == (13,) and bins[0] == 0.0 and bins[-1] == 1.0 and np.all(np.diff(bins) > 0)
    np.testing.assert_allclose(error, np.abs(rerr).max(axis=0), rtol=1e-9)
