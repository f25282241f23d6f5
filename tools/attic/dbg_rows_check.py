"""Which part of the blocks of the 32-state LGL7 shape differs from the tile kernel (ASSET_HIP_NO_ROWS=1 reference file)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import DefectEvaluator
w = Workload("synthetic32", "LGL7", 600, False)
ev = DefectEvaluator("synthetic32", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
fx, agx, kkt = ev.eval(4, w.X, w.L)
out = sys.argv[1]
if not os.path.exists(out):
    np.savez(out, fx=fx, agx=agx, kkt=kkt); print("saved"); sys.exit(0)
ref = np.load(out)
IR, OR = ev.IR, ev.OR
def rel(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
print("fx", rel(fx, ref["fx"]), "agx", rel(agx, ref["agx"]), "kkt", rel(kkt, ref["kkt"]))
k = np.asarray(kkt).reshape(600, -1); kr = ref["kkt"].reshape(600, -1)
d = np.abs(k - kr).max(axis=0)
pos = 0; bad = []
for c in range(IR):
    nh = IR - c
    eh = d[pos:pos + nh]; ej = d[pos + nh:pos + nh + OR]
    if eh.max() > 1e-9 * np.abs(kr).max(): bad.append(("H", c, int(np.argmax(eh)) + c, float(eh.max())))
    if ej.max() > 1e-9 * np.abs(kr).max(): bad.append(("J", c, int(np.argmax(ej)), float(ej.max())))
    pos += nh + OR
print(len(bad), bad[:12])
da = np.abs(np.asarray(agx).reshape(600, -1) - ref["agx"].reshape(600, -1)).max(axis=0)
print("agx bad cols", np.nonzero(da > 1e-9 * np.abs(ref["agx"]).max())[0][:20])
segerr = np.abs(k - kr).max(axis=1); print("segments with errors", np.nonzero(segerr > 1e-9 * np.abs(kr).max())[0][:20], (segerr > 1e-9 * np.abs(kr).max()).sum())
