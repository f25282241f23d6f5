import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import Workload
from asset_asrl_amd.evaluator import DefectEvaluator
from oracle import bindings as ob
ob.build()
ode, mode, blocked, nseg = sys.argv[1], sys.argv[2], sys.argv[3] == "1", int(sys.argv[4])
w = Workload(ode, mode, nseg, blocked)
ev = DefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
got, ref = ev.eval(4, w.X, w.L), w.oracle_nlp(ob, threads=8).eval_blocks(4, w.X, w.L)
badf = np.abs(got[0] - ref[0]).max(axis=1) > 1e-9
badg = np.abs(got[1] - ref[1]).max(axis=1) > 1e-7 * np.abs(ref[1]).max()
badk = np.abs(got[2] - ref[2]).max(axis=1) > 1e-7 * np.abs(ref[2]).max()
nsh = 2048; per, rem = nseg // nsh, nseg % nsh
for sh in list(range(6)) + [1000, 1001, 2046, 2047]:
    f = sh * per + min(sh, rem); c = per + (1 if sh < rem else 0)
    print(f"share {sh} (wave {sh & 1} of wg {sh >> 1}) segs {f}..{f + c - 1}: fx " + "".join("x" if b else "." for b in badf[f:f + c])
          + " agx " + "".join("x" if b else "." for b in badg[f:f + c]) + " kkt " + "".join("x" if b else "." for b in badk[f:f + c]))
# which entries of fx / which block slots are wrong in the first bad segment
s = int(np.nonzero(badf | badk)[0][0])
print("first bad segment", s, "fx diff", np.round(got[0][s] - ref[0][s], 4))
d = np.abs(got[2][s] - ref[2][s]) > 1e-7 * np.abs(ref[2]).max()
print("bad kkt slots", int(d.sum()), "of", d.size, np.nonzero(d)[0][:40])
