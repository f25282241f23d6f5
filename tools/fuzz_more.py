"""The randomised-shape test of tests/test_gpu_fuzz.py over more seeds than the suite's fixed 24 cases (a one-off after kernel changes):
  python tools/fuzz_more.py [cases per seed] [seed ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import Workload
from test_gpu_fuzz import _cases
from test_gpu_parity import _check_blocks
from asset_asrl_amd import _lib
from asset_asrl_amd.evaluator import CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, DefectEvaluator
from oracle import bindings as ob
ob.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seeds = [int(v) for v in sys.argv[2:]] or [1, 2, 3, 4, 5]
ran = bad = 0
for seed in seeds:
    for ode, mode, blocked, nseg, voff, coff, extra in _cases(n, seed):
        if not _lib.has_kernel(ode, _lib.MODES[mode], blocked):
            continue
        w = Workload(ode, mode, nseg, blocked, seed=nseg + voff + seed, var_offset=voff, con_offset=coff, extra_vars=extra)
        nlp = w.oracle_nlp(ob, threads=8)
        ev = DefectEvaluator(ode, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
        for what in (JAC_ADJGRAD_HESS, JAC_ADJGRAD, JAC, CON_ADJGRAD, CON):
            ref = nlp.eval_blocks(what, w.X, w.L)
            got = ev.eval(what, w.X, w.L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
            try:
                _check_blocks(got, ref, w, what)
            except AssertionError as exc:
                bad += 1
                print("MISMATCH", seed, ode, mode, blocked, nseg, voff, coff, extra, what, str(exc)[:200])
        ev.close()
        ran += 1
print(f"{ran} random shapes x 5 kinds, {bad} mismatches")
