"""EnableHessianSparsity of the Trapezoidal defects (TrapezoidalDefects.h:39-141), in the oracle's restatement of the NLP: the
mask claims fewer KKT slots and loses nothing -- the assembled matrix is the unmasked one, the dropped entries were exact zeros."""
import numpy as np
import pytest

from helpers import Workload


def _dense(nlp, vals):
    outer, inner = nlp.csr()
    M = np.zeros((nlp.kkt_dim, nlp.kkt_dim))
    for r in range(nlp.kkt_dim):
        M[r, inner[outer[r]:outer[r + 1]]] = vals[outer[r]:outer[r + 1]]
    return M


@pytest.mark.parametrize("ode,blocked", [("reentry", False), ("twobody_lt", True), ("betts_lowthrust", False), ("brachistochrone", True)])
def test_mask_drops_only_structural_zeros(oracle, ode, blocked):
    w = Workload(ode, "Trapezoidal", 7, blocked, var_offset=1, con_offset=2, extra_vars=2)
    full = w.oracle_nlp(oracle, threads=1)
    masked = w.oracle_nlp(oracle, threads=2, hessian_sparsity=True)
    q = w.IR // 2 if not (w.IR % 2) and not blocked else None
    assert masked.num_user_kkt < full.num_user_kkt and masked.nnz <= full.nnz
    _, _, vf_ = full.eval(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    fx, agx, vm = masked.eval(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    np.testing.assert_array_equal(_dense(full, vf_), _dense(masked, vm))
    rfx, ragx, _ = full.eval(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    np.testing.assert_array_equal(fx, rfx)
    np.testing.assert_array_equal(agx, ragx)
    # the LGL transcriptions have no such mask: the switch changes nothing there
    wl = Workload(ode, "LGL3", 5, blocked)
    assert wl.oracle_nlp(oracle, hessian_sparsity=True).num_user_kkt == wl.oracle_nlp(oracle).num_user_kkt


def test_python_mask_and_locations_match_the_oracle(oracle):
    from asset_asrl_amd.indexing import kkt_slot_locations, trapezoidal_hessian_mask
    from asset_asrl_amd.ode import TwoBody
    w = Workload("twobody_lt", "Trapezoidal", 6, True)
    m = trapezoidal_hessian_mask(6, 3, 0, True)
    locs, nnz = kkt_slot_locations(w.vindex, w.cindex, w.n_primal, hess_mask=m)
    nlp = w.oracle_nlp(oracle, hessian_sparsity=True)
    # the oracle's NLP also holds its own diagonal entries (primal diagonal, equality pivots): compare the claimed slots' coordinates
    rows, cols = nlp.kkt_coords()
    kept = locs[locs >= 0]
    assert kept.size == nlp.num_user_kkt and (locs < 0).sum() == 6 * int((~m)[np.tril_indices(m.shape[0])].sum())
    ph = TwoBody().phase("Trapezoidal", w.traj, 6)
    ph.setControlMode("BlockConstant")
    assert ph.hessian_mask() is None
    ph.EnableHessianSparsity = True
    np.testing.assert_array_equal(ph.hessian_mask(), m)
