"""CPU tests of the layout contract: phase indexer, block slot order, CSR assembly, thread-split invariance."""
import numpy as np
import pytest

from asset_asrl_amd.evaluator import unpack_kkt_block
from asset_asrl_amd.indexing import PhaseIndexer, thread_split
from helpers import Workload, rel_err


def test_indexer_reference_test_case(oracle):
    """(Xv,Uv,Pv,SPv)=(6,3,1,2), CS=3, 5 defects -- the case hard-coded in PhaseIndexer::Test
    (/root/reference/src/OptimalControl/PhaseIndexer.cpp:560-618)."""
    ix = PhaseIndexer(6, 3, 1, 2)
    ix.set_dimensions(3, 5, False)
    assert ix.numStates == 11 and ix.numPhaseVars == 11 * 10 + 1 + 2
    V, Cx = ix.make_defect_Vindex_Cindex()
    assert V.shape == (5, 31) and Cx.shape == (5, 12)
    # DefectPath column i = states 2i,2i+1,2i+2 (10 vars each) then the ODE parameter at 110
    for i in range(5):
        np.testing.assert_array_equal(V[i, :30], np.arange(20 * i, 20 * i + 30))
        assert V[i, 30] == 110
    np.testing.assert_array_equal(Cx.ravel(), np.arange(60))
    Vo, Co = oracle.phase_defect_index(6, 3, 1, 2, 3, 5, False)
    np.testing.assert_array_equal(V, Vo)
    np.testing.assert_array_equal(Cx, Co)
    # blocked: nodes carry (x,t) only, one control vector per defect after the nodes
    ix.set_dimensions(3, 5, True)
    assert ix.numPhaseVars == 11 * 7 + 5 * 3 + 1 + 2
    V, Cx = ix.make_defect_Vindex_Cindex()
    Vo, Co = oracle.phase_defect_index(6, 3, 1, 2, 3, 5, True, 0, 0)
    np.testing.assert_array_equal(V, Vo)
    for i in range(5):
        np.testing.assert_array_equal(V[i, :21], np.arange(14 * i, 14 * i + 21))
        np.testing.assert_array_equal(V[i, 21:24], 77 + 3 * i + np.arange(3))
        assert V[i, 24] == 92
    assert oracle.phase_num_vars(6, 3, 1, 2, 3, 5, True) == ix.numPhaseVars


def test_indexer_offsets_and_roundtrip(oracle):
    for blocked in (False, True):
        ix = PhaseIndexer(5, 2, 1, 0)
        ix.set_dimensions(4, 7, blocked)
        ix.begin_indexing(13, 29)
        V, Cx = ix.make_defect_Vindex_Cindex()
        Vo, Co = oracle.phase_defect_index(5, 2, 1, 0, 4, 7, blocked, 13, 29)
        np.testing.assert_array_equal(V, Vo)
        np.testing.assert_array_equal(Cx, Co)
        rng = np.random.default_rng(0)
        traj = rng.normal(size=(ix.numStates, 9))
        traj[:, -1] = 0.7
        if blocked:  # one control per segment: make the trajectory block constant
            for s in range(ix.numStates):
                traj[s, 6:8] = traj[min(s // 3, 6) * 3, 6:8]
        ix.begin_indexing(0, 0)
        X = ix.makeSolverInput(traj)
        back, _ = ix.collectSolverOutput(X)
        np.testing.assert_allclose(back, traj)


def test_thread_split_rule():
    assert thread_split(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert thread_split(3, 8) == [(0, 1), (1, 1), (2, 1)]


def test_kkt_slot_order_and_assembly(oracle):
    """2-segment toy: block order = for i<IR {H(j,i) j>=i ; J(j,i)}; scatter puts H at (VLoc j, VLoc i)
    upper-transposed and J at (VLoc i, Primal + CLoc j) (DenseFunctionBase.h:1112-1123, NonLinearProgram.cpp:282-307)."""
    w = Workload("brachistochrone", "LGL3", 2)
    nlp = w.oracle_nlp(oracle, threads=1)
    fxb, agxb, kkt = nlp.eval_blocks(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    o = oracle.get_ode("brachistochrone", 0)
    outer, inner = nlp.csr()
    dense = np.zeros((nlp.kkt_dim, nlp.kkt_dim))
    for V in range(2):
        fx, jx, gx, hx = oracle.defect_all(o, oracle.LGL3, w.X[w.vindex[V]], w.L[w.cindex[V]])
        H, J = unpack_kkt_block(kkt[V], w.IR, w.OR)
        np.testing.assert_array_equal(H, np.tril(hx) + np.tril(hx, -1).T)
        np.testing.assert_array_equal(J, jx)
        np.testing.assert_array_equal(fxb[V], fx)
        np.testing.assert_array_equal(agxb[V], gx)
        for a in range(w.IR):
            for b in range(a + 1):
                r, c = sorted((w.vindex[V, a], w.vindex[V, b]))
                dense[r, c] += hx[a, b]
            for k in range(w.OR):
                dense[w.vindex[V, a], w.n_primal + w.cindex[V, k]] += jx[k, a]
    FXE, AGX, vals = nlp.eval(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    got = np.zeros_like(dense)
    for r in range(nlp.kkt_dim):
        for k in range(outer[r], outer[r + 1]):
            assert inner[k] >= r                       # upper triangular CSR
            got[r, inner[k]] = vals[k]
    np.testing.assert_allclose(got, dense, rtol=0, atol=1e-12 * np.abs(dense).max())
    agx_ref = np.zeros(w.n_primal)
    np.add.at(agx_ref, w.vindex.ravel(), agxb.ravel())
    np.testing.assert_allclose(AGX, agx_ref, atol=1e-12 * np.abs(agx_ref).max())
    np.testing.assert_array_equal(FXE[w.cindex.ravel()], fxb.ravel())


@pytest.mark.parametrize("ode,mode,blocked", [("reentry", "LGL7", False), ("twobody_lt", "LGL5", True),
                                               ("betts_lowthrust", "LGL5", False)])
def test_thread_count_invariance(oracle, ode, mode, blocked):
    """NLPTest's i-thread vs j-thread comparison (NonLinearProgram.cpp:754-784), asserted."""
    w = Workload(ode, mode, 23, blocked)
    a = w.oracle_nlp(oracle, threads=1).eval(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    b = w.oracle_nlp(oracle, threads=5).eval(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    for u, v in zip(a, b):
        assert rel_err(u, v) < 1e-13
    for what in (oracle.CON, oracle.CON_ADJGRAD, oracle.JAC, oracle.JAC_ADJGRAD):
        a = w.oracle_nlp(oracle, threads=1).eval(what, w.X, w.L)
        b = w.oracle_nlp(oracle, threads=3).eval(what, w.X, w.L)
        for u, v in zip(a, b):
            if u is not None:
                assert rel_err(u, v) < 1e-13
