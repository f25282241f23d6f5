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
