"""The OptimalControlProblem mirror (asset_asrl_amd/ocp.py): phases placed one after the other in the solver vector, their
equality and inequality rows numbered as the reference's transcribe_phases does
(/root/reference/src/OptimalControl/OptimalControlProblem.cpp:115-155: Vstart / Estart / Istart are the sums over the phases
before).  Host only."""
import numpy as np
import pytest

from asset_asrl_amd import vf
from asset_asrl_amd.ocp import OptimalControlProblem
from asset_asrl_amd.ode import ShuttleReentry, TwoBody
from helpers import Workload


def _problem():
    ocp = OptimalControlProblem()
    p0 = ShuttleReentry().phase("LGL7", Workload("reentry", "LGL7", 5).traj, 5)          # spline controls: spacing + spline rows
    b = vf.Arguments(4)
    p0.addInequalCon("PairWisePath", vf.stack([b[0] * b[2] - b[1] * b[3] - 0.5]), [0, 1])
    p1 = TwoBody().phase("LGL5", Workload("twobody_lt", "LGL5", 4, True).traj, 4)
    p1.setControlMode("BlockConstant")
    p2 = ShuttleReentry().phase("LGL3", Workload("reentry", "LGL3", 3).traj, 3)
    p2.addInequalCon("Path", vf.stack([vf.Arguments(2)[0] - vf.Arguments(2)[1]]), [0, 2])
    assert ocp.addPhases([p0, p1, p2]) == [0, 1, 2] and ocp.getPhaseNum(p1) == 1 and ocp.Phase(2) is p2
    with pytest.raises(ValueError):
        ocp.addPhase(p1)
    return ocp


def test_phases_are_laid_out_one_after_the_other():
    ocp = _problem()
    lays = ocp.transcribe_phases()
    v = e = i = 0
    for ph, (ix, (V, Cx), entries, neq, niq) in zip(ocp.phases, lays):
        alone = ph.layout()                                              # the same phase on its own, at offset 0
        assert (ix.var_offset, ix.con_offset) == (v, e)
        np.testing.assert_array_equal(V, alone[1][0] + v)
        np.testing.assert_array_equal(Cx, alone[1][1] + e)
        assert V.min() >= v and V.max() < v + ix.numPhaseVars and Cx.min() == e
        for ent, ent0 in zip(entries, alone[2]):
            np.testing.assert_array_equal(ent[4], ent0[4] + v)
            off = {"inequality": i, "objective": 0}.get(ent[0], e)       # equality rows at Estart, inequality rows at Istart
            np.testing.assert_array_equal(ent[5], ent0[5] + off)
        assert (neq, niq) == (alone[3], alone[4])
        v, e, i = v + ix.numPhaseVars, e + neq, i + niq
    assert (ocp.n_primal, ocp.n_equal, ocp.n_inequal) == (v, e, i)
    assert ocp.numPhaseIqCons == [15, 0, 4]                              # 16 states -> 15 pairs; the 4 states of the LGL3 phase
    # every equality row of the problem is claimed exactly once
    rows = np.concatenate([lay[1][1].ravel() for lay in lays] +
                          [ent[5].ravel() for lay in lays for ent in lay[2] if ent[0] in ("auto", "equality")])
    np.testing.assert_array_equal(np.sort(rows), np.arange(ocp.n_equal))
    X = ocp.solver_input()
    for ph, lay in zip(ocp.phases, lays):
        ix = lay[0]
        np.testing.assert_array_equal(X[ix.var_offset:ix.var_offset + ix.numPhaseVars], ph.layout()[0].makeSolverInput(ph.ActiveTraj))


def test_phase_sharded_evaluator_needs_like_phases():
    with pytest.raises(ValueError):
        _problem().phase_sharded_evaluator(rank=0, world=1, evaluator_factory=lambda *a: None)
