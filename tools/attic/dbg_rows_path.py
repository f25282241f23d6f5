import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from helpers import Workload, make_coupled
from asset_asrl_amd import jit
from asset_asrl_amd.evaluator import DefectEvaluator
n, mode, blocked = int(sys.argv[1]), sys.argv[2], sys.argv[3] == "1"
name = jit.ensure_kernel(make_coupled(n), mode, blocked)
w = Workload(f"coupled{n}", mode, 29, blocked, sizes=(n, 3, 2), var_offset=2, con_offset=1, extra_vars=3)
ev = DefectEvaluator(name, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
fx, agx, kkt = ev.eval(4, w.X, w.L)
np.save(sys.argv[4], kkt)
print("IR", ev.IR, "sum", float(np.abs(kkt).sum()))
