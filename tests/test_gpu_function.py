"""Plain vector functions batched over applications: a nonlinear path constraint written in
the DSL, compiled at run time, against the oracle's AD2 derivatives of the same function pushed through the oracle's
NLP restatement (blocks, scattered CSR values, RHS vectors); and the reference's LGL mesh-spacing relation against
its closed-form Jacobian (MeshSpacingConstraints.h:128-142)."""
import numpy as np
import pytest

from asset_asrl_amd import vf
from asset_asrl_amd.evaluator import CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, unpack_kkt_block
from asset_asrl_amd.pathfuncs import FunctionEvaluator, LGLMeshSpacing
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _pathcon():
    a = vf.Arguments(6)
    x0, x1, x2, t, u0, u1 = a.tolist()
    return vf.stack([x0 * x0 + x1 * u0 - vf.sin(x2), u0 * u0 + u1 * u1 - 1.0 + t * x0 * vf.exp(-1.0 * x1)])


def _tables(napp, ir, orr, n_primal, n_equal, seed):
    """Index tables with shared variables: consecutive applications overlap in half of their inputs."""
    rng = np.random.default_rng(seed)
    step = ir // 2
    vindex = (np.arange(napp)[:, None] * step + np.arange(ir)[None, :] + 3).astype(np.int32)
    cindex = (np.arange(napp)[:, None] * orr + np.arange(orr)[None, :] + 1).astype(np.int32)
    assert vindex.max() < n_primal and cindex.max() < n_equal
    return vindex, cindex, rng.uniform(-1, 1, n_primal), 10.0 * rng.uniform(-1, 1, n_equal)


def test_path_constraint_matches_oracle_nlp(oracle):
    napp, ir, orr = 211, 6, 2
    n_primal, n_equal = 3 * napp + 20, 2 * napp + 5
    vindex, cindex, X, L = _tables(napp, ir, orr, n_primal, n_equal, 11)
    ev = FunctionEvaluator(_pathcon(), "pathcon", vindex, cindex, n_primal, n_equal)
    assert (ev.IR, ev.OR, ev.NKKT) == (6, 2, 21 + 12)
    nlp = oracle.Nlp(oracle.get_ode("pathcon", 0), oracle.MODES["Function"], False, vindex, cindex, n_primal, n_equal, 2)
    for what in (JAC_ADJGRAD_HESS, CON, CON_ADJGRAD, JAC, JAC_ADJGRAD):
        rfx, ragx, rkkt = nlp.eval_blocks(what, X, L)
        fx, agx, kkt = ev.eval(what, X, L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
        assert np.abs(fx - rfx).max() < 1e-10
        if agx is not None:
            assert rel_err(agx, ragx) < 1e-8
        if kkt is not None:
            assert rel_err(kkt, rkkt) < 1e-8
    # on-device assembly into the solver's value array (shared variables -> shared locations)
    locs = nlp.kkt_locations()[:nlp.num_user_kkt].reshape(napp, ev.NKKT)
    assert np.unique(locs).size < locs.size
    ev.set_kkt_map(locs, nlp.nnz)
    for what in (JAC_ADJGRAD_HESS, JAC_ADJGRAD):
        vals = np.zeros(nlp.nnz)
        ev.eval_assembled(what, X, L, vals)
        assert rel_err(vals, nlp.eval(what, X, L)[2]) < 1e-8
    ev.close()


def test_function_on_tabulated_data_matches_oracle_nlp(oracle):
    """A plain function built on three ``vf.InterpTable1D`` tables, batched over applications that share variables (the reference's
    InterpFunction1D inside a path function, CommonFunctions/InterpTable1D.h:322-401) -- against the oracle's NLP restatement over
    the same right-hand side (oracle/odes.h: tabulated)."""
    from helpers import make_tabulated
    napp, ir, orr = 300, 4, 2
    n_primal, n_equal = 2 * napp + 20, 2 * napp + 5
    vindex, cindex, X, L = _tables(napp, ir, orr, n_primal, n_equal, 12)
    X *= 2.4                                                               # past both ends of every table
    ev = FunctionEvaluator(make_tabulated().vf(), "tabulated_fn", vindex, cindex, n_primal, n_equal)
    nlp = oracle.Nlp(oracle.get_ode("tabulated", 0), oracle.MODES["Function"], False, vindex, cindex, n_primal, n_equal, 2)
    for what in (JAC_ADJGRAD_HESS, CON, CON_ADJGRAD, JAC, JAC_ADJGRAD):
        rfx, ragx, rkkt = nlp.eval_blocks(what, X, L)
        fx, agx, kkt = ev.eval(what, X, L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
        assert np.abs(fx - rfx).max() < 1e-10
        if agx is not None:
            assert rel_err(agx, ragx) < 1e-8
        if kkt is not None:
            assert rel_err(kkt, rkkt) < 1e-8
    ev.close()


def test_lgl_mesh_spacing_closed_form():
    cs, napp = 4, 100
    rng = np.random.default_rng(2)
    t0 = np.sort(rng.uniform(0, 50, napp))
    h = rng.uniform(0.5, 2.0, napp)
    frac = np.sort(rng.uniform(0.05, 0.95, (napp, 2)), axis=1)
    T = np.column_stack([t0, t0 + frac[:, 0] * h, t0 + frac[:, 1] * h, t0 + h])       # node times per application
    X = T.ravel()
    vindex = np.arange(napp * cs, dtype=np.int32).reshape(napp, cs)
    cindex = np.arange(napp * (cs - 2), dtype=np.int32).reshape(napp, cs - 2)
    L = rng.uniform(-1, 1, napp * (cs - 2))
    ev = FunctionEvaluator(LGLMeshSpacing(cs), "lglmeshspacing4", vindex, cindex, X.size, L.size)
    fx, agx, kkt = ev.eval(JAC_ADJGRAD_HESS, X, L)
    tc = np.array([0.0, 2.65575603264643e-1, 7.34424396735357e-1, 1.0])
    for V in (0, 17, 99):
        x = T[V]
        hh = x[3] - x[0]
        assert np.allclose(fx[V], tc[1:3] - (x[1:3] - x[0]) / hh, atol=1e-14)
        J = np.zeros((2, 4))
        for i in range(2):                                   # MeshSpacingConstraints.h:136-141
            J[i, i + 1] = -1.0 / hh
            J[i, 0] = 1.0 / hh - (x[1 + i] - x[0]) / hh ** 2
            J[i, 3] = (x[1 + i] - x[0]) / hh ** 2
        H, Jd = unpack_kkt_block(kkt[V], 4, 2)
        assert rel_err(Jd, J) < 1e-12
        assert rel_err(agx[V], J.T @ L[2 * V:2 * V + 2]) < 1e-12
        eps = 1e-6                                           # adjoint Hessian against central differences of J^T lam
        Hfd = np.zeros((4, 4))
        for c in range(4):
            for sgn in (1, -1):
                xp = x.copy()
                xp[c] += sgn * eps
                hp = xp[3] - xp[0]
                Jp = np.zeros((2, 4))
                for i in range(2):
                    Jp[i, i + 1] = -1.0 / hp
                    Jp[i, 0] = 1.0 / hp - (xp[1 + i] - xp[0]) / hp ** 2
                    Jp[i, 3] = (xp[1 + i] - xp[0]) / hp ** 2
                Hfd[:, c] += sgn * (Jp.T @ L[2 * V:2 * V + 2]) / (2 * eps)
        assert rel_err(H, Hfd) < 1e-6
    ev.close()


def test_phase_add_constraints_evaluate_over_their_regions(oracle):
    """Phase.addEqualCon / addInequalCon (ODEPhaseBase.h; region tables of PhaseIndexer.cpp:132-360): a path equality
    at every state and a pair-wise inequality, evaluated on the device through the phase, against the oracle's NLP
    restatement running the same index tables."""
    from asset_asrl_amd.ode import ShuttleReentry
    from helpers import Workload
    w = Workload("reentry", "LGL5", 23)
    ph = ShuttleReentry().phase("LGL5", w.traj, 23)
    ph.setControlMode("NoSpline")                                             # (only the user functions beside the defects:
    ph.EnableMeshSpacing = False                                              #  the phase's own ones: test_gpu_phase_functions.py)
    k_eq = ph.addEqualCon("Path", _pathcon(), [0, 1, 2, 5, 6, 7])            # x0, x1, x2, t, u0, u1 of every state
    a = vf.Arguments(4)
    k_iq = ph.addInequalCon("PairWisePath", vf.stack([a[0] * a[2] - a[1] * a[3] - 0.5]), [3, 4])
    assert (k_eq, k_iq) == (0, 0)
    ph.transcribe()
    ix = ph._indexer
    S = ix.numStates
    (eq,), (iq,) = ph.equality_evaluators, ph.inequality_evaluators
    assert eq.nseg == S and iq.nseg == S - 1
    assert ph.numPhaseEqCons == ix.numPhaseEqCons + 2 * S and ph.numPhaseIqCons == S - 1
    assert ph.evaluator.n_equal == ph.numPhaseEqCons                      # one multiplier vector for all equalities
    X = ph.solver_input()
    rng = np.random.default_rng(5)
    L = 10.0 * rng.uniform(-1, 1, ph.numPhaseEqCons)
    V, Cx, _ = ix.make_Vindex_Cindex("Path", [0, 1, 2, 5, 6, 7], orows=2)
    nlp = oracle.Nlp(oracle.get_ode("pathcon", 0), oracle.MODES["Function"], False, V, Cx, X.size, L.size, 2)
    for what in (JAC_ADJGRAD_HESS, CON):
        rfx, ragx, rkkt = nlp.eval_blocks(what, X, L)
        fx, agx, kkt = eq.eval(what, X, L if what == JAC_ADJGRAD_HESS else None)
        assert np.abs(fx - rfx).max() < 1e-10
        if kkt is not None:
            assert rel_err(agx, ragx) < 1e-8 and rel_err(kkt, rkkt) < 1e-8
    # the inequality: closed form
    Li = rng.uniform(-1, 1, ph.numPhaseIqCons)
    fx, agx, kkt = iq.eval(JAC_ADJGRAD_HESS, X, Li)
    xs = X[: S * 8].reshape(S, 8)
    np.testing.assert_allclose(fx[:, 0], xs[:-1, 3] * xs[1:, 3] - xs[:-1, 4] * xs[1:, 4] - 0.5, atol=1e-13)
    H, J = unpack_kkt_block(kkt[4], 4, 1)
    np.testing.assert_allclose(J[0], [xs[5, 3], -xs[5, 4], xs[4, 3], -xs[4, 4]], atol=1e-13)
    np.testing.assert_allclose(H, Li[4] * np.array([[0, 0, 1, 0], [0, 0, 0, -1], [1, 0, 0, 0], [0, -1, 0, 0.0]]), atol=1e-13)
    # the defects still evaluate with the longer multiplier vector
    fxd, _, _ = ph.evaluator.eval(CON, X)
    assert fxd.shape == (23, ph.evaluator.OR)


def test_segment_quadrature_and_control_spline_on_the_device():
    """LGLIntegral / LGLControlSpline (LGLIntegrals.h:9-52, LGLControlSplines.h:64-108) batched over segments on the
    device: known answers -- the LGL7 quadrature is exact for the cubic integrand of a polynomial trajectory, and the
    spline relation vanishes on a control that is one cubic across both segments -- and value, Jacobian, adjoint gradient
    and adjoint Hessian of every application against the oracle's restatement (oracle/pathfuncs.cpp, itself pinned by
    the 50-digit vectors of tests/golden/pathfuncs.npz)."""
    from asset_asrl_amd.pathfuncs import LGLControlSpline, LGLIntegral
    tc = np.array([0.0, 2.65575603264643e-1, 7.34424396735357e-1, 1.0])
    nseg = 150
    rng = np.random.default_rng(9)
    t0 = np.sort(rng.uniform(0, 20, nseg))
    h = rng.uniform(0.3, 2.0, nseg)
    # --- quadrature: states (t^3, t), integrand x1^2 + x0 = t^2 + t^3
    g = vf.Arguments(2)
    F = LGLIntegral(g.coeff(1) * g.coeff(1) + g.coeff(0), 4, 2)
    T = t0[:, None] + tc[None, :] * h[:, None]
    Z = np.stack([T ** 3, T, T], axis=2).reshape(nseg, 12)                   # [x0, x1, t] at the four nodes
    X = Z.ravel()
    vindex = np.arange(nseg * 12, dtype=np.int32).reshape(nseg, 12)
    cindex = np.arange(nseg, dtype=np.int32).reshape(nseg, 1)
    L = rng.uniform(0.5, 1.5, nseg)
    ev = FunctionEvaluator(F, "lglintegral_test", vindex, cindex, X.size, L.size)
    fx, agx, kkt = ev.eval(JAC_ADJGRAD_HESS, X, L)
    prim = lambda t: t ** 3 / 3 + t ** 4 / 4
    np.testing.assert_allclose(fx[:, 0], prim(t0 + h) - prim(t0), rtol=1e-12, atol=1e-12)
    from oracle import bindings as ob                                         # every application against the oracle's
    quad2 = ob.get_ode("integrand_quad2", 0)                                  # restatement of LGLIntegrals.h:18-52
    for V in range(nseg):
        rfx, rjx, rgx, rhx = ob.lgl_integral_all(quad2, 4, 2, 0, Z[V], L[V:V + 1])
        Hd, Jd = unpack_kkt_block(kkt[V], 12, 1)
        assert abs(fx[V, 0] - rfx[0]) < 1e-10 * max(1.0, np.abs(Z).max())
        assert rel_err(Jd, rjx) < 1e-8 and rel_err(agx[V], rgx) < 1e-8 and rel_err(Hd, rhx) < 1e-8
    H, J = unpack_kkt_block(kkt[3], 12, 1)
    np.testing.assert_allclose(J[0] * L[3], agx[3], rtol=1e-12, atol=1e-12)   # one output: J^T lam = lam * J
    assert np.abs(H - H.T).max() == 0.0
    ev.close()
    # --- control spline over pairs of segments: [t, u0, u1] at the 7 nodes of two adjacent segments
    S = LGLControlSpline(4, 2)
    npair = 60
    ta = np.sort(rng.uniform(0, 10, npair))
    h0, h1 = rng.uniform(0.5, 1.5, npair), rng.uniform(0.5, 1.5, npair)
    ts = np.concatenate([ta[:, None] + tc[None, :] * h0[:, None], (ta + h0)[:, None] + tc[None, 1:] * h1[:, None]], axis=1)
    Zs = np.stack([ts, 1 + 2 * ts - 0.5 * ts ** 2 + 0.3 * ts ** 3, ts ** 3 - ts], axis=2).reshape(npair, 21)
    vind = np.arange(npair * 21, dtype=np.int32).reshape(npair, 21)
    cind = np.arange(npair * 4, dtype=np.int32).reshape(npair, 4)
    ev = FunctionEvaluator(S, "lglcontrolspline4", vind, cind, Zs.size, npair * 4)
    fx = ev.eval(CON, Zs.ravel())[0]
    assert np.abs(fx).max() < 1e-9 * max(1.0, np.abs(Zs).max() ** 3)         # 15-digit weight literals
    Zr = Zs + rng.uniform(-0.05, 0.05, Zs.shape) * (np.arange(21) % 3 != 0)   # perturbed controls, same node times
    Ls = rng.uniform(-2, 2, npair * 4)
    fx, agx, kkt = ev.eval(JAC_ADJGRAD_HESS, Zr.ravel(), Ls)                  # against LGLControlSplines.h:92-309 restated
    for V in range(npair):
        rfx, rjx, rgx, rhx = ob.control_spline_all(4, 2, Zr[V], Ls[4 * V:4 * V + 4])
        Hd, Jd = unpack_kkt_block(kkt[V], 21, 4)
        assert np.abs(fx[V] - rfx).max() < 1e-10 * max(1.0, np.abs(rfx).max())
        assert rel_err(Jd, rjx) < 1e-8 and rel_err(agx[V], rgx) < 1e-8 and rel_err(Hd, rhx) < 1e-8
    Zbad = Zs.copy()
    Zbad[:, -1] += 0.1                                                        # a kink in the last node's control
    assert np.abs(ev.eval(CON, Zbad.ravel())[0]).max(axis=1).min() > 1e-2
    ev.close()


def test_mesh_spacing_and_parametrised_integral_match_the_oracle():
    """LGLMeshSpacing<3,4>, SingleMeshSpacing (MeshSpacingConstraints.h:8-193) and an LGL5 / LGL3 quadrature whose integrand
    takes a phase parameter, batched on the device, against the oracle's closed forms application by application."""
    from oracle import bindings as ob
    from asset_asrl_amd.pathfuncs import LGLIntegral, LGLMeshSpacing, SingleMeshSpacing
    rng = np.random.default_rng(31)
    napp = 90

    def run(F, name, Z, orr, ref):
        irr = Z.shape[1]
        vindex = np.arange(napp * irr, dtype=np.int32).reshape(napp, irr)
        cindex = np.arange(napp * orr, dtype=np.int32).reshape(napp, orr)
        L = rng.uniform(-2, 2, napp * orr)
        ev = FunctionEvaluator(F, name, vindex, cindex, Z.size, L.size)
        fx, agx, kkt = ev.eval(JAC_ADJGRAD_HESS, Z.ravel(), L)
        for V in range(napp):
            rfx, rjx, rgx, rhx = ref(Z[V], L[orr * V:orr * V + orr])
            Hd, Jd = unpack_kkt_block(kkt[V], irr, orr)
            assert np.abs(fx[V] - rfx).max() < 1e-10 * max(1.0, np.abs(Z).max())
            assert rel_err(Jd, rjx) < 1e-8 and rel_err(agx[V], rgx) < 1e-8 and rel_err(Hd, rhx, floor=1e-12) < 1e-8
        ev.close()

    def times(cs):
        tc = synth_tc(cs)
        t = rng.uniform(0, 5, (napp, 1)) + tc[None, :] * rng.uniform(0.4, 2.0, (napp, 1))
        return t + rng.uniform(-0.02, 0.02, t.shape) * (np.arange(cs) > 0)

    for cs in (3, 4):
        run(LGLMeshSpacing(cs), f"lglmeshspacing{cs}", times(cs), cs - 2, lambda x, l, cs=cs: ob.lgl_mesh_spacing_all(cs, x, l))
    s = float(synth_tc(4)[1])
    run(SingleMeshSpacing(s, 2.5), "singlemeshspacing_t", np.sort(rng.uniform(0, 3, (napp, 3)), axis=1), 1,
        lambda x, l: ob.single_mesh_spacing_all(s, x, l, scale=2.5))
    g = vf.Arguments(4)
    integ = g.coeff(3) * g.coeff(0) * g.coeff(0) + vf.sin(g.coeff(1)) * g.coeff(2) \
        + vf.exp(-1.0 * (g.coeff(0) * g.coeff(2))) / (1.0 + g.coeff(3) * g.coeff(3))
    powp = ob.get_ode("integrand_powp", 0)
    for cs in (3, 2):
        Z = np.concatenate([np.concatenate([rng.uniform(-1, 1, (napp, cs, 3)), times(cs)[:, :, None]], axis=2).reshape(napp, -1),
                            rng.uniform(0.5, 1.5, (napp, 1))], axis=1)
        run(LGLIntegral(integ, cs, 3, 1), f"lglintegral{cs}_powp", Z, 1,
            lambda x, l, cs=cs: ob.lgl_integral_all(powp, cs, 3, 1, x, l))


def synth_tc(cs):
    from asset_asrl_amd import synth
    return synth._TC[cs]
