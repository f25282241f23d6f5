"""Region index tables of PhaseIndexer.make_Vindex_Cindex (PhaseIndexer.cpp:132-360) on small hand-checkable phases."""
import numpy as np
import pytest

from asset_asrl_amd.indexing import PhaseIndexer


def _ix(blocked=False, xv=2, uv=1, pv=1, spv=1, cs=3, nd=3):
    ix = PhaseIndexer(xv, uv, pv, spv)
    ix.set_dimensions(cs, nd, blocked)
    ix.begin_indexing(5, 2)
    return ix


def test_path_and_boundary_regions_non_blocked():
    ix = _ix()                                  # q = 4, S = 7 states, P at 5 + 28, SP at 5 + 29
    q, S, o = 4, 7, 5
    V, Cx, nxt = ix.make_Vindex_Cindex("Path", [0, 3], [0], [0], orows=2)
    assert V.shape == (S, 4) and Cx.shape == (S, 2)
    np.testing.assert_array_equal(V[:, 0], o + q * np.arange(S))
    np.testing.assert_array_equal(V[:, 1], o + q * np.arange(S) + 3)
    assert (V[:, 2] == o + S * q).all() and (V[:, 3] == o + S * q + 1).all()
    c0 = 2 + ix.numPhaseEqCons                  # rows follow the defects' (PhaseIndexer.cpp:176-187)
    np.testing.assert_array_equal(Cx.ravel(), c0 + np.arange(2 * S))
    assert nxt == c0 + 2 * S
    V, _, _ = ix.make_Vindex_Cindex("Front", [0, 1, 2])
    np.testing.assert_array_equal(V, [[o, o + 1, o + 2]])
    V, _, _ = ix.make_Vindex_Cindex("Back", [2])
    np.testing.assert_array_equal(V, [[o + (S - 1) * q + 2]])
    V, _, _ = ix.make_Vindex_Cindex("FrontandBack", [2], [0])
    np.testing.assert_array_equal(V, [[o + 2, o + (S - 1) * q + 2, o + S * q]])
    V, _, _ = ix.make_Vindex_Cindex("BackandFront", [2])
    np.testing.assert_array_equal(V, [[o + (S - 1) * q + 2, o + 2]])
    V, _, _ = ix.make_Vindex_Cindex("InnerPath", [1])
    np.testing.assert_array_equal(V[:, 0], o + q * np.arange(1, S - 1) + 1)
    V, _, _ = ix.make_Vindex_Cindex("NodalPath", [1])          # every (CS-1)-th state: the defect boundaries
    np.testing.assert_array_equal(V[:, 0], o + q * np.array([0, 2, 4, 6]) + 1)
    V, _, _ = ix.make_Vindex_Cindex("PairWisePath", [0])
    np.testing.assert_array_equal(V, np.column_stack([o + q * np.arange(S - 1), o + q * np.arange(1, S)]))
    V, _, _ = ix.make_Vindex_Cindex("FrontNodalBackPath", [0])
    np.testing.assert_array_equal(V, [[o, o + q * 2, o + q * 6], [o, o + q * 4, o + q * 6]])
    V, Cx, _ = ix.make_Vindex_Cindex("Params", [], [0], [0], orows=1, next_cloc=40)
    np.testing.assert_array_equal(V, [[o + S * q, o + S * q + 1]])
    np.testing.assert_array_equal(Cx, [[40]])


def test_control_only_functions_in_a_block_constant_phase_apply_once_per_defect():
    ix = _ix(blocked=True)                      # X = [7 states x (x0,x1,t)] [3 x u] [P] [SP]
    S, xt, o = 7, 3, 5
    V, Cx, _ = ix.make_Vindex_Cindex("Path", [3], orows=1)     # variable 3 = the control
    np.testing.assert_array_equal(V[:, 0], o + S * xt + np.arange(3))
    assert Cx.shape == (3, 1)
    V, _, _ = ix.make_Vindex_Cindex("PairWisePath", [3])
    np.testing.assert_array_equal(V, [[o + S * xt, o + S * xt + 1], [o + S * xt + 1, o + S * xt + 2]])
    V, _, _ = ix.make_Vindex_Cindex("Path", [0, 3])            # a state too: every state, control of its defect
    assert V.shape == (S, 2)
    np.testing.assert_array_equal(V[:, 1], o + S * xt + np.array([0, 0, 1, 1, 2, 2, 2]))


def test_errors():
    ix = _ix()
    with pytest.raises(ValueError):
        ix.make_Vindex_Cindex("Sideways", [0])
    with pytest.raises(ValueError):
        ix.make_Vindex_Cindex("Path", [9])


def test_integral_param_function_tables_share_the_accumulation_row():
    """Host-only (Phase.layout): PhaseIndexer::addAccumulation (PhaseIndexer.cpp:41-76) -- the accumulation function over the
    Params region claims one equality row, and every application of the integrand's quadrature names that same row."""
    import numpy as np
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ShuttleReentry
    from helpers import Workload
    nseg = 6
    w = Workload("reentry", "LGL5", nseg)
    ph = ShuttleReentry().phase("LGL5", w.traj, nseg)
    ph.setStaticParams([1.0, 2.0, 3.0])
    g = vf.Arguments(1)
    ph.addIntegralParamFunction(g.coeff(0) * g.coeff(0), [3], accum_param=2)
    a = vf.Arguments(2)
    ph.addEqualCon("Front", vf.stack([a[0] - a[1]]), [0, 1])
    ix, (V, C), entries, neq, niq = ph.layout(Vstart=10, Estart=4)
    tags = [e[1] for e in entries]
    i_acc, i_int = tags.index("ipf0_acc"), tags.index("ipf0_int")
    assert i_int == i_acc + 1 and tags.index("eq0") > i_int          # transcribe_integrals precedes the user functions
    Va, Ca = entries[i_acc][4], entries[i_acc][5]
    Vi, Ci = entries[i_int][4], entries[i_int][5]
    assert Va.tolist() == [[10 + ix.StaticParamLoc0 + 2]] and Ca.shape == (1, 1)
    assert Ci.shape == (nseg, 1) and (Ci == Ca[0, 0]).all()
    assert Vi.shape == (nseg, 3 * 2)                                  # (gamma, t) at the three states of every defect
    rows_before = Ca[0, 0] - 4
    assert neq == rows_before + 1 + 1                                 # the pair's one row, then the Front equality's
    assert entries[tags.index("eq0")][5][0, 0] == Ca[0, 0] + 1
    X = ph.solver_input() if False else ix.makeSolverInput(ph.ActiveTraj, ph.ActiveStaticParams)
    assert X.size == ix.numPhaseVars and np.array_equal(X[ix.StaticParamLoc0:], [1.0, 2.0, 3.0])


def test_integral_functions_under_autoscaling_constrain_the_same_integral():
    """AutoScaling and the integral functions (ODEPhaseBase.cpp:796-803, 846-861): with the solver's variables in scaled units the
    integrand is IOScaled over its input units with its output scale, and the accumulation of an integral parameter function is
    ``-AccScale * p_scaled``, AccScale = SPUnits[p] * output_scale / t_unit -- so the row the solver sees is
    ``(output_scale / t_unit) * (int f dt - scale * p)`` of the unscaled problem, and an integral objective ``(output_scale / t_unit) *
    int f dt``.  Host-only: the phase's function tables evaluated with the expression graph (no device)."""
    import numpy as np
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ShuttleReentry
    from helpers import Workload
    nseg = 5
    w = Workload("reentry", "LGL7", nseg)
    units = np.array([2.0, 0.5, 3.0, 1.5, 0.8, 4.0, 1.25, 2.5])
    sp, spu = np.array([0.3, -1.2]), np.array([4.0, 0.5])
    oscale_p, oscale_o, scale = 3.0, 0.7, 2.0

    def build(auto):
        ph = ShuttleReentry().phase("LGL7", w.traj, nseg)
        ph.setStaticParams(sp, units=spu)
        g = vf.Arguments(3)                                    # (state variables 2 and 0, static parameter 1)
        ph.addIntegralParamFunction(g[1] * g[1] + g[0] * g[2], [2, 0], SPVars=[1], accum_param=0, scale=scale, output_scale=oscale_p)
        g2 = vf.Arguments(3)
        ph.addIntegralObjective(vf.sin(g2[0]) * g2[1] + g2[2] * g2[2], [2, 0], SPVars=[1], output_scale=oscale_o)
        if auto:
            ph.setUnits(units)
            ph.setAutoScaling(True)
        ix, _, entries, _, _ = ph.layout()
        X = ix.makeSolverInput(ph.ActiveTraj / units if auto else ph.ActiveTraj, sp / spu if auto else sp)
        val = {}
        for kind, tag, F, name, V, Cx, consts in entries:
            if tag.startswith(("ipf", "obj")):
                val[tag] = sum(float(F.compute(X[V[a]])[0]) for a in range(V.shape[0]))
        return val
    plain, scaled = build(False), build(True)
    t_unit = units[5]
    row_plain = plain["ipf0_acc"] + plain["ipf0_int"]
    row_scaled = scaled["ipf0_acc"] + scaled["ipf0_int"]
    assert abs(plain["ipf0_acc"] + scale * sp[0]) < 1e-15 and abs(plain["ipf0_int"]) > 1e-3
    assert abs(row_scaled - oscale_p / t_unit * row_plain) <= 1e-13 * max(1.0, abs(row_plain))
    assert abs(scaled["obj0"] - oscale_o / t_unit * plain["obj0"]) <= 1e-13 * max(1.0, abs(plain["obj0"]))
