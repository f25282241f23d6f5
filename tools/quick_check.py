"""Parity of one (ODE, transcription) against the oracle for an experimental library: every evaluation kind, a few mesh sizes.

  ASSET_HIP_LIB=exp_build/v_x/lib.so python tools/quick_check.py reentry LGL7 [blocked] [sizes ...]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import Workload, rel_err
from asset_asrl_amd.evaluator import DefectEvaluator
from oracle import bindings as ob
ob.build()
ode, mode = sys.argv[1], sys.argv[2]
blocked = len(sys.argv) > 3 and sys.argv[3] == "1"
sizes = [int(v) for v in sys.argv[4:]] or [1, 2, 3, 7, 64, 257, 2049, 10000, 12345, 30011]
worst = 0.0
for nseg in sizes:
    w = Workload(ode, mode, nseg, blocked)
    ev = DefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    nlp = w.oracle_nlp(ob, threads=8)
    for what in (4, 0, 1, 2, 3):
        L = w.L if what in (1, 3, 4) else None
        got, ref = ev.eval(what, w.X, L), nlp.eval_blocks(what, w.X, w.L)
        e = [np.abs(got[0] - ref[0]).max() / max(1.0, np.abs(w.X).max())]
        if what in (1, 3, 4): e.append(rel_err(got[1], ref[1]))
        if what >= 2: e.append(rel_err(got[2], ref[2]))
        worst = max(worst, max(e))
        if max(e) > 1e-9 or not np.isfinite(max(e)):
            bad = np.nonzero(np.abs(got[0] - ref[0]).max(axis=1) > 1e-9)[0]
            kb = np.nonzero(np.abs(got[2] - ref[2]).max(axis=1) > 1e-7 * max(1.0, np.abs(ref[2]).max()))[0] if what >= 2 else []
            print(f"MISMATCH {ode} {mode} x{nseg} kind {what}: {e}; segments with wrong values {len(bad)}: {bad[:12]} ... {bad[-4:]}; wrong blocks {len(kb)}: {kb[:12]}")
    ev.close()
print(f"{os.environ.get('ASSET_HIP_LIB', 'default')} {ode} {mode} blocked={blocked} sizes {sizes}: worst error {worst:.2e}", "OK" if worst < 1e-9 else "FAILED")
