"""The layout of the KKT blocks is the handle's own and it exports it (include/asset_hip.h: asset_hip_defect_kkt_layout; the
reference: getKKTSpace is a method of the plug-in, SolverInterfaceSpecs.h:41-92 -- the solver maps every (row, col) it is told to
a matrix location whatever the order, NonLinearProgram.cpp:282-330).  Narrow transcriptions write the Jacobian column-major, then
the packed lower triangle of H, each region a whole number of 128-byte lines; plain functions and wide shapes the reference's
order.  Here: the exported table against the host mirror, the raw blocks against the oracle entry by entry THROUGH the table,
padding never written, and a host scatter of the blocks in the handle's order against the scatter in the reference's order, bit
for bit."""
import numpy as np
import pytest

from asset_asrl_amd import _lib
from asset_asrl_amd.evaluator import JAC, JAC_ADJGRAD_HESS, KEEP_HESSIAN_SLOTS, DefectEvaluator, kkt_layout_table, reference_slot_order
from asset_asrl_amd.indexing import kkt_slot_locations
from helpers import Workload, rel_err

pytestmark = pytest.mark.gpu

CASES = [("reentry", "LGL7", False, 131), ("reentry", "LGL7", True, 67), ("reentry", "LGL5", False, 200), ("reentry", "LGL3", False, 33),
         ("twobody_lt", "LGL5", True, 257), ("twobody_lt", "LGL7", False, 90), ("brachistochrone", "LGL3", False, 40),
         ("betts_lowthrust", "LGL5", False, 60), ("reentry", "Trapezoidal", False, 100), ("synthetic32", "Trapezoidal", False, 7),
         ("synthetic32", "LGL7", False, 5)]


@pytest.mark.parametrize("ode,mode,blocked,nseg", CASES)
def test_exported_layout_and_raw_blocks(oracle, ode, mode, blocked, nseg):
    import torch
    w = Workload(ode, mode, nseg, blocked)
    ev = DefectEvaluator(ode, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    IR, OR = ev.IR, ev.OR
    # the handle's table = the table of the compiled (ode, mode, blocked) = the host mirror of the kernels' arithmetic
    kl, nk, stride, rows, cols = _lib.kkt_layout(ode, _lib.MODES[mode], w.blocked)
    assert (kl, nk, stride) == (ev.kkt_layout, ev.NKKT, ev.KSTRIDE)
    np.testing.assert_array_equal(rows, ev.kkt_rows)
    np.testing.assert_array_equal(cols, ev.kkt_cols)
    st2, r2, c2 = kkt_layout_table(IR, OR, kl)
    assert st2 == stride and np.array_equal(r2, rows) and np.array_equal(c2, cols)
    assert kl == (0 if IR >= 64 and mode != "Trapezoidal" else 1)
    if kl == 1:        # both regions start on a 128-byte line, and so does every block
        assert stride % 16 == 0 and ((OR * IR + 15) // 16 * 16) % 16 == 0
        assert np.all(rows[:OR * IR] >= IR) and np.all(rows[(OR * IR + 15) // 16 * 16:][rows[(OR * IR + 15) // 16 * 16:] >= 0] < IR)
    real = rows >= 0
    assert int(real.sum()) == ev.NKKT
    # raw blocks on the device, the array pre-filled with a sentinel: every entry through the table against the oracle, padding untouched
    dev = torch.device("cuda:0")
    X, L = torch.from_numpy(w.X).to(dev), torch.from_numpy(w.L).to(dev)
    fx = torch.empty(nseg * OR, dtype=torch.float64, device=dev)
    agx = torch.empty(nseg * IR, dtype=torch.float64, device=dev)
    SENT = -7.25e33
    kkt = torch.full((nseg * stride,), SENT, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ev.eval_device(JAC_ADJGRAD_HESS, X, L, fx, agx, kkt)
    torch.cuda.synchronize()
    blocks = kkt.cpu().numpy().reshape(nseg, stride)
    assert np.all(blocks[:, ~real] == SENT) and not np.any(blocks[:, real] == SENT)
    nlp = w.oracle_nlp(oracle)
    rfx, ragx, rkkt = nlp.eval_blocks(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    o = oracle.get_ode(ode, 0)
    for V in range(0, nseg, max(1, nseg // 7)):
        _, rjx, _, rhx = oracle.defect_all(o, oracle.MODES[mode], w.X[w.vindex[V]], w.L[w.cindex[V]], blocked=w.blocked)
        scale_j, scale_h = max(1.0, np.abs(rjx).max()), max(1.0, np.abs(rhx).max())
        for k in np.nonzero(real)[0]:
            r, c = int(rows[k]), int(cols[k])
            if r < IR:
                assert r >= c and abs(blocks[V, k] - rhx[r, c]) <= 1e-8 * scale_h
            else:
                assert abs(blocks[V, k] - rjx[r - IR, c]) <= 1e-8 * scale_j
    # the canonical view (what DefectEvaluator.eval returns) is the oracle's block array
    assert rel_err(ev.kkt_to_reference(blocks), rkkt) < 1e-8
    # the Jacobian kinds: Hessian slots zero, or left alone with KEEP_HESSIAN_SLOTS -- in the handle's layout too
    kkt.fill_(SENT)
    torch.cuda.synchronize()
    ev.eval_device(JAC, X, None, fx, None, kkt)
    torch.cuda.synchronize()
    b1 = kkt.cpu().numpy().reshape(nseg, stride)
    hslots, jslots = real & (rows < IR), real & (rows >= IR)
    assert np.all(b1[:, hslots] == 0.0) and np.all(b1[:, ~real] == SENT)
    # (the Jacobian kinds are kernels of their own -- two ODE phases, other sums: equal to rounding, not bit for bit)
    assert rel_err(b1[:, jslots], blocks[:, jslots]) < 1e-12
    kkt.fill_(SENT)
    torch.cuda.synchronize()
    ev.eval_device(JAC | KEEP_HESSIAN_SLOTS, X, None, fx, None, kkt)
    torch.cuda.synchronize()
    b2 = kkt.cpu().numpy().reshape(nseg, stride)
    np.testing.assert_array_equal(b2[:, jslots], b1[:, jslots])
    assert np.all((b2[:, hslots] == SENT) | (b2[:, hslots] == 0.0)) and np.all(b2[:, ~real] == SENT)
    ev.close()


@pytest.mark.parametrize("ode,mode,nseg", [("reentry", "LGL7", 500), ("reentry", "LGL5", 333), ("brachistochrone", "LGL7", 77)])
def test_scatter_in_the_handles_order_is_bitwise_the_scatter_in_the_reference_order(ode, mode, nseg):
    """A phase without parameters: every matrix location receives at most two contributions (the node two segments share), and
    a + b = b + a -- so KKTFillAll over the blocks in the order the device writes them (the order BatchedDefectConstraint's
    getKKTSpace tells the solver) fills the CSR value array bit for bit as the fill over the reference's slot order does."""
    w = Workload(ode, mode, nseg)
    ev = DefectEvaluator(ode, mode, False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    locs, nnz = kkt_slot_locations(w.vindex, w.cindex, w.n_primal)          # canonical numbering: [nseg, NKKT]
    _, _, native = ev.eval(JAC_ADJGRAD_HESS, w.X, w.L, native=True)
    assert native.shape == (nseg, ev.KSTRIDE)
    canonical = ev.kkt_to_reference(native)
    v_ref = np.zeros(nnz)
    np.add.at(v_ref, locs.ravel(), canonical.ravel())                       # segment by segment, reference slot order
    # the same locations listed in the handle's order (what analyzeSparsity would hand back for the handle's getKKTSpace)
    rr, cc = reference_slot_order(ev.IR, ev.OR)
    where = {(int(r), int(c)): k for k, (r, c) in enumerate(zip(rr, cc))}
    real = np.nonzero(ev.kkt_rows >= 0)[0]
    canon_of = np.asarray([where[(int(ev.kkt_rows[k]), int(ev.kkt_cols[k]))] for k in real])
    v_nat = np.zeros(nnz)
    np.add.at(v_nat, locs[:, canon_of].ravel(), native[:, real].ravel())
    np.testing.assert_array_equal(v_nat, v_ref)
    assert np.abs(v_ref).max() > 0
    ev.close()
