"""Phase stamps (s_memtime) of the four waves of one workgroup of the row-wise wide dense stage, segment 2 of workgroup 7.

  python tools/build_one.py tu_synthetic32_lgl4_0 build_dbg/rt/lib.so -DASSET_TIMING
  ASSET_HIP_LIB=build_dbg/rt/lib.so python tools/dbg_time_rows.py
"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import DefectEvaluator
nseg = 12500
w = Workload("synthetic32", "LGL7", nseg, False)
ev = DefectEvaluator("synthetic32", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
fx, agx, kkt = ev.eval(4, w.X, w.L)
G = 256; per = nseg // G; rem = nseg % G
first = 7 * per + min(7, rem)
row = np.asarray(fx).reshape(nseg, -1)[first + 2]
names = ["wait B0", "S0", "wait B1", "prep", "wait B2", "H columns", "J columns"]
for wv in range(4):
    print(f"wave {wv}: " + " | ".join(f"{names[t]} {row[wv * 8 + t]:8.0f}" for t in range(7)))
