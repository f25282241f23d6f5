"""Prints the clock64() stamps a -DASSET_TIMING build of the one-launch kernel of heavy ODEs (defect_ures.h) leaves in the KKT block of
workgroup 7's last group: per wave, cycles since the wave's start.

  python tools/build_one.py tu_betts_lowthrust_lgl3_0 build_dbg/libuT.so -DASSET_TIMING
  ASSET_HIP_LIB=build_dbg/libuT.so python tools/dbg_time_ures.py 1000 betts_lowthrust LGL5
"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import DefectEvaluator
nseg = int(sys.argv[1]); ode, mode = sys.argv[2], sys.argv[3]
w = Workload(ode, mode, nseg, False)
ev = DefectEvaluator(ode, mode, False, w.vindex, w.cindex, w.n_primal, w.n_equal)
for rep in range(3):
    fx, agx, kkt = ev.eval(4, w.X, w.L)
cus = 256
gmax = int(os.environ.get("GMAX", 64 // {"LGL3": 2, "LGL5": 3, "LGL7": 4}[mode]))
rounds = -(-nseg // (cus * gmax)); G = min(gmax, max(1, -(-nseg // (cus * rounds)))); ngrp = -(-nseg // G)
grp = 7
while grp + min(ngrp, cus) < ngrp: grp += min(ngrp, cus)
row = kkt.reshape(nseg, -1)[grp * G]
names = ["start", "P0 issued", "barrier", "P1 f_j", "P2 interior unit", "barrier", "P3 cardinal unit", "barrier", "dense passes"]
print("G", G, "groups", ngrp, "group read", grp)
for wv in range(8):
    n = int(row[wv * 16 + 15]); d = row[wv * 16: wv * 16 + n].astype(np.int64)
    print("wave", wv, " ".join(f"{v:7d}" for v in d))
print("stamps:", ", ".join(names), "(waves without a unit skip P1's stamp)")
