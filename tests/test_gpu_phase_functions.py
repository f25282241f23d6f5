"""What ``Phase.transcribe()`` registers beside the defects, as the reference's own transcribe does
(ODEPhaseBase.cpp:962-1061, 743-889, 1371-1375): the mesh-spacing equalities (LGLMeshSpacing over every defect,
SingleMeshSpacing at every inner nodal state), the control-spline equalities of the spline control modes, integral
objectives -- each one device evaluator, checked application by application against the oracle's restatements, and the
row numbering / index tables against the registration order of the reference."""
import numpy as np
import pytest

from asset_asrl_amd import vf
from asset_asrl_amd.evaluator import JAC_ADJGRAD_HESS, unpack_kkt_block
from asset_asrl_amd.ode import ShuttleReentry
from helpers import Workload, rel_err

pytestmark = pytest.mark.gpu


def _check(ev, X, L, ref, scale=1.0):
    fx, agx, kkt = ev.eval(JAC_ADJGRAD_HESS, X, L)
    for V in range(ev.nseg):
        rfx, rjx, rgx, rhx = ref(V, X[ev.vindex[V]], L[ev.cindex[V]])
        Hd, Jd = unpack_kkt_block(kkt[V], ev.IR, ev.OR)
        assert np.abs(fx[V] - rfx).max() < 1e-10 * max(1.0, scale)
        assert rel_err(Jd, rjx) < 1e-8 and rel_err(agx[V], rgx) < 1e-8 and rel_err(Hd, rhx, floor=1e-9) < 1e-8


@pytest.mark.parametrize("mode,control,order", [("LGL7", "HighestOrderSpline", 2), ("LGL7", "FirstOrderSpline", 1),
                                                ("LGL5", "HighestOrderSpline", 1)])
def test_transcribe_registers_spacing_and_spline_equalities(oracle, mode, control, order):
    nseg = 17
    w = Workload("reentry", mode, nseg)
    ph = ShuttleReentry().phase(mode, w.traj, nseg)
    ph.setControlMode(control)
    ph.transcribe()
    ix, cs, D = ph._indexer, w.cs, nseg
    S, xtu, tcol = ix.numStates, ix.XtUVars(), ph.ode.TVar()
    evs = ph.phase_function_evaluators
    assert set(evs) == {"mesh_spacing", "nodal_spacing", "control_spline"}
    ms, ns, sp = evs["mesh_spacing"], evs["nodal_spacing"], evs["control_spline"]
    # rows: defects, then LGLMeshSpacing, SingleMeshSpacing, LGLControlSpline (transcribe_phase order)
    r0 = ix.numPhaseEqCons
    np.testing.assert_array_equal(ms.cindex.ravel(), r0 + np.arange(D * (cs - 2)))
    r1 = r0 + D * (cs - 2)
    np.testing.assert_array_equal(ns.cindex.ravel(), r1 + np.arange(D - 1))
    r2 = r1 + D - 1
    np.testing.assert_array_equal(sp.cindex.ravel(), r2 + np.arange((D - 1) * 2 * order))
    assert ph.numPhaseEqCons == r2 + (D - 1) * 2 * order == ph.evaluator.n_equal
    # tables: the node times of a defect / first, nodal, last time / [t, u0, u1] of the 2cs-1 states of a defect pair
    t_of = lambda k: k * xtu + tcol
    np.testing.assert_array_equal(ms.vindex, [[t_of(i * (cs - 1) + j) for j in range(cs)] for i in range(D)])
    np.testing.assert_array_equal(ns.vindex, [[t_of(0), t_of(i * (cs - 1)), t_of(S - 1)] for i in range(1, D)])
    np.testing.assert_array_equal(sp.vindex, [[(i * (cs - 1) + j) * xtu + v for j in range(2 * cs - 1) for v in (tcol, tcol + 1, tcol + 2)]
                                              for i in range(D - 1)])
    X = ph.solver_input()
    rng = np.random.default_rng(3)
    X = X + rng.uniform(-1e-3, 1e-3, X.size)                                   # off the exact spacing: non-zero residuals
    L = rng.uniform(-2, 2, ph.numPhaseEqCons)
    _check(ms, X, L, lambda V, x, l: oracle.lgl_mesh_spacing_all(cs, x, l))
    _check(ns, X, L, lambda V, x, l: oracle.single_mesh_spacing_all((V + 1) / D, x, l), scale=np.abs(X).max())
    _check(sp, X, L, lambda V, x, l: oracle.control_spline_all(cs, 2, x, l, order=order), scale=np.abs(X).max() ** 2)
    # the defects still evaluate with the longer multiplier vector
    assert ph.evaluator.eval(JAC_ADJGRAD_HESS, X, L)[0].shape == (nseg, ph.evaluator.OR)


def test_no_spline_equalities_without_spline_control(oracle):
    w = Workload("reentry", "LGL7", 5)
    for control in ("NoSpline", "BlockConstant"):
        ph = ShuttleReentry().phase("LGL7", w.traj, 5)
        ph.setControlMode(control)
        ph.transcribe()
        assert set(ph.phase_function_evaluators) == {"mesh_spacing", "nodal_spacing"}
    ph = ShuttleReentry().phase("LGL3", Workload("reentry", "LGL3", 5).traj, 5)
    ph.transcribe()
    assert set(ph.phase_function_evaluators) == {"nodal_spacing"}              # LGL3: no inner cardinal node, no spline


def test_integral_objective_evaluates_the_segment_quadrature(oracle):
    """addIntegralObjective (ODEPhaseBase.cpp:743-889): LGLIntegral of the integrand over every defect; the evaluator's
    multiplier vector is [ObjScale], its adjoint gradient the scaled gradient, its Hessian block the scaled Hessian."""
    nseg = 19
    w = Workload("reentry", "LGL7", nseg)
    ph = ShuttleReentry().phase("LGL7", w.traj, nseg)
    g = vf.Arguments(2)
    assert ph.addIntegralObjective(g.coeff(1) * g.coeff(1) + g.coeff(0), [2, 0]) == 0      # inputs (v, h): h^2 + v
    ph.transcribe()
    (ob_ev,) = ph.objective_evaluators
    ix, cs, xtu, tcol = ph._indexer, 4, ph._indexer.XtUVars(), ph.ode.TVar()
    np.testing.assert_array_equal(ob_ev.vindex, [[(i * 3 + j) * xtu + v for j in range(cs) for v in (2, 0, tcol)] for i in range(nseg)])
    X = ph.solver_input()
    scale = np.array([0.37])
    quad2 = oracle.get_ode("integrand_quad2", 0)
    _check(ob_ev, X, scale, lambda V, x, l: oracle.lgl_integral_all(quad2, 4, 2, 0, x, l), scale=np.abs(X).max() ** 2)
    # the objective value: sum over the defects of h * sum_j w_j (h_j^2 + v_j)  [inputs (v, h): y1^2 + y0], w = Reduced_Integral_Weights of the header
    import json
    import os
    wts = np.array(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lgl_tables.json")))["tables"]["4"]["Reduced_Integral_Weights"])
    fx = ob_ev.eval(0, X)[0]
    T = ph.ActiveTraj
    expect = sum((T[3 * i + 3, tcol] - T[3 * i, tcol]) * np.dot(wts, T[3 * i:3 * i + 4, 0] ** 2 + T[3 * i:3 * i + 4, 2]) for i in range(nseg))
    assert abs(fx.sum() - expect) < 1e-11 * abs(expect), (fx.sum(), expect)


def test_function_bundle_gives_the_blocks_of_the_separate_launches():
    """Phase.function_bundle(): the mesh-spacing, nodal-spacing and control-spline equalities, a user path equality, a
    pair-wise inequality and an integral objective in ONE launch -- bit for bit what one launch per function returns, for
    the three block kinds and the value-only kind."""
    import torch
    from asset_asrl_amd.evaluator import CON, JAC_ADJGRAD, JAC_ADJGRAD_HESS
    nseg = 37
    w = Workload("reentry", "LGL7", nseg)
    ph = ShuttleReentry().phase("LGL7", w.traj, nseg)
    a = vf.Arguments(6)
    x0, x1, x2, t, u0, u1 = a.tolist()
    ph.addEqualCon("Path", vf.stack([x0 * x0 + x1 * u0 - vf.sin(x2), u0 * u0 + u1 * u1 - 1.0 + t * x0 * vf.exp(-1.0 * x1)]),
                   [0, 1, 2, 5, 6, 7])
    b = vf.Arguments(4)
    ph.addInequalCon("PairWisePath", vf.stack([b[0] * b[2] - b[1] * b[3] - 0.5]), [0, 1])
    g = vf.Arguments(2)
    ph.addIntegralObjective(g.coeff(1) * g.coeff(1) + g.coeff(0), [2, 0])
    ph.transcribe()
    bundle, members = ph.function_bundle()
    assert [k for k, _ in members] == ["equality"] * 4 + ["inequality", "objective"]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    X = torch.from_numpy(ph.solver_input()).to(dev)
    LE = torch.from_numpy(rng.uniform(-1, 1, ph.numPhaseEqCons)).to(dev)
    LI = torch.from_numpy(rng.uniform(-1, 1, max(1, ph.numPhaseIqCons))).to(dev)
    LO = torch.tensor([0.75], dtype=torch.float64, device=dev)
    Ls = [{"equality": LE, "inequality": LI, "objective": LO}[k] for k, _ in members]

    def outs():
        return ([torch.full((e.nseg * e.OR,), np.nan, dtype=torch.float64, device=dev) for _, e in members],
                [torch.full((e.nseg * e.IR,), np.nan, dtype=torch.float64, device=dev) for _, e in members],
                [torch.full((e.nseg * e.KSTRIDE,), np.nan, dtype=torch.float64, device=dev) for _, e in members])
    for what in (JAC_ADJGRAD_HESS, JAC_ADJGRAD, CON):
        fb, gb, kb = outs()
        fs, gs, ks = outs()
        torch.cuda.synchronize()                               # (torch fills on its stream, the evaluators run on their own)
        bundle.eval_device(what, X, Ls, fb, gb if what != CON else [None] * len(members), kb if what != CON else [None] * len(members))
        for (_, e), l, f, g_, k in zip(members, Ls, fs, gs, ks):
            e.eval_device(what, X, l if what != CON else None, f, g_ if what != CON else None, k if what != CON else None)
        torch.cuda.synchronize()
        for one, two in zip(fb + (gb + kb if what != CON else []), fs + (gs + ks if what != CON else [])):
            assert torch.equal(one, two) and not torch.isnan(one).any()
    bundle.close()


def test_integral_param_function_shares_one_row_between_accumulation_and_quadrature(oracle):
    """addIntegralParamFunction (ODEPhaseBase.cpp:835-889 / PhaseIndexer::addAccumulation, PhaseIndexer.cpp:41-76): the
    accumulation -scale * p over the Params region claims a row; every application of the segment quadrature carries that row.
    Summed over the row: int (h^2 + v) dt - scale * p."""
    nseg = 13
    w = Workload("reentry", "LGL7", nseg)
    ph = ShuttleReentry().phase("LGL7", w.traj, nseg)
    ph.setStaticParams([0.25, -1.5])
    g = vf.Arguments(2)
    assert ph.addIntegralParamFunction(g.coeff(1) * g.coeff(1) + g.coeff(0), [2, 0], accum_param=1, scale=2.0) == 0
    ph.transcribe()
    ((acc, quad),) = ph.integral_param_evaluators
    ix, cs, xtu, tcol = ph._indexer, 4, ph._indexer.XtUVars(), ph.ode.TVar()
    row = int(acc.cindex[0, 0])
    assert acc.vindex.tolist() == [[ix.StaticParamLoc0 + 1]] and acc.cindex.shape == (1, 1)
    np.testing.assert_array_equal(quad.cindex, np.full((nseg, 1), row))
    np.testing.assert_array_equal(quad.vindex, [[(i * 3 + j) * xtu + v for j in range(cs) for v in (2, 0, tcol)] for i in range(nseg)])
    # rows: defects, mesh spacing, nodal spacing, control spline, then the ONE row of the pair; nothing after it here
    assert ph.numPhaseEqCons == row + 1 == ph.evaluator.n_equal
    X = ph.solver_input()
    assert X[ix.StaticParamLoc0 + 1] == -1.5
    rng = np.random.default_rng(11)
    L = rng.uniform(-2, 2, ph.numPhaseEqCons)
    quad2 = oracle.get_ode("integrand_quad2", 0)
    _check(quad, X, L, lambda V, x, l: oracle.lgl_integral_all(quad2, 4, 2, 0, x, l), scale=np.abs(X).max() ** 2)
    fa, ga, ka = acc.eval(JAC_ADJGRAD_HESS, X, L)
    assert fa.shape == (1, 1) and fa[0, 0] == -2.0 * -1.5 and ga[0, 0] == -2.0 * L[row]
    # the row's value: the quadrature of every defect plus the accumulation
    import json
    import os
    wts = np.array(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lgl_tables.json")))["tables"]["4"]["Reduced_Integral_Weights"])
    T = ph.ActiveTraj
    integral = sum((T[3 * i + 3, tcol] - T[3 * i, tcol]) * np.dot(wts, T[3 * i:3 * i + 4, 0] ** 2 + T[3 * i:3 * i + 4, 2]) for i in range(nseg))
    total = quad.eval(0, X)[0].sum() + fa[0, 0]
    assert abs(total - (integral + 3.0)) < 1e-11 * max(1.0, abs(integral))
