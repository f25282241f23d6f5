"""Run-time compiled ODEs of UNUSUAL dimensions through every narrow form of the dense part, against the oracle: one smooth
right-hand side defined for any (states, controls, parameters) (helpers.make_shape; oracle/odes.h: shape_nmp, differentiated by AD2).
The BASELINE workloads and the other user ODEs of the suite have 2-7 or 12+ states; these have ONE state, no controls, node strides
that are and are not multiples of four (the rule that picks the row-wise form of csrc/defect_rowdpp.h over the tile form of
csrc/defect_resident.h), parameters with and without controls, two row groups of defect rows, N + 1 = 16 (the last shape of the
resident kernel) and N + 1 = 17 (the first of the fallback, csrc/defect_kernels.h) -- at a ragged small mesh and at one that takes
the looped kernels (there the two kinds that form blocks); all five evaluation kinds, and on-device assembly."""
import numpy as np
import pytest

from asset_asrl_amd import jit
from asset_asrl_amd.evaluator import CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, DefectEvaluator
from helpers import Workload, make_shape, rel_err
from test_gpu_parity import _check_blocks

pytestmark = pytest.mark.gpu
CASES = [(1, 0, 0, "LGL7", False), (1, 0, 0, "LGL3", False), (1, 1, 0, "LGL5", False), (1, 1, 0, "LGL7", True),
         (2, 1, 0, "Trapezoidal", True), (2, 1, 0, "LGL5", True), (3, 0, 1, "LGL5", False), (3, 0, 1, "LGL7", False),
         (4, 4, 0, "LGL7", False), (4, 4, 0, "LGL7", True), (5, 3, 2, "LGL5", False), (5, 3, 2, "LGL3", True),
         (6, 0, 0, "LGL7", False), (8, 3, 1, "LGL7", False), (8, 3, 1, "LGL3", False), (10, 4, 0, "LGL3", False),
         (11, 4, 0, "LGL3", False), (11, 4, 0, "LGL5", False)]


@pytest.mark.parametrize("n,m,p,mode,blocked", CASES)
def test_shape_matches_oracle(oracle, n, m, p, mode, blocked):
    name = jit.ensure_kernel(make_shape(n, m, p), mode, blocked)
    ode = oracle.get_ode(f"shape_{n}_{m}_{p}", 0)
    cs = {"Trapezoidal": 2, "LGL3": 2, "LGL5": 3, "LGL7": 4}[mode]
    big = 21011 if cs * (n + 1 + m) + p <= 30 else 9001        # (the looped kernels either way; the oracle's AD2 pass is what takes the time)
    for nseg in (43, big):
        w = Workload(f"shape_{n}_{m}_{p}", mode, nseg, blocked, sizes=(n, m, p), var_offset=2, con_offset=1, extra_vars=3)
        nlp = oracle.Nlp(ode, oracle.MODES[mode], w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal, 8)
        ev = DefectEvaluator(name, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
        for what in ((JAC_ADJGRAD_HESS, CON, CON_ADJGRAD, JAC, JAC_ADJGRAD) if nseg < 100 else (JAC_ADJGRAD_HESS, JAC)):
            ref = nlp.eval_blocks(what, w.X, w.L)
            got = ev.eval(what, w.X, w.L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
            _check_blocks(got, ref, w, what)
        if nseg < 100:      # on-device assembly through the same kernels (shared boundary nodes, parameters, block controls)
            locs = nlp.kkt_locations()[:nlp.num_user_kkt].reshape(w.nseg, ev.NKKT)
            ev.set_kkt_map(locs, nlp.nnz)
            vals = np.zeros(nlp.nnz)
            ev.eval_assembled(JAC_ADJGRAD_HESS, w.X, w.L, vals)
            assert rel_err(vals, nlp.eval(JAC_ADJGRAD_HESS, w.X, w.L)[2]) < 1e-8
        ev.close()
