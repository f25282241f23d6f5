"""Prints the clock64() deltas a -DASSET_TIMING build of the resident kernel leaves in FX (workgroup 7).

  python tools/build_one.py tu_reentry_lgl4_0 build_dbg/libdbgT.so -DASSET_TIMING
  ASSET_HIP_LIB=build_dbg/libdbgT.so python tools/dbg_time_res.py
"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import *
nseg=int(sys.argv[1]) if len(sys.argv)>1 else 10000
ode, mode, blocked = (sys.argv[2], sys.argv[3], sys.argv[4] == "1") if len(sys.argv) > 4 else ("reentry", "LGL7", False)
w=Workload(ode,mode,nseg,blocked)
ev=DefectEvaluator(ode,mode,blocked,w.vindex,w.cindex,w.n_primal,w.n_equal)
for rep in range(3):
    fx,agx,kkt=ev.eval(4,w.X,w.L)
G=min(int(os.environ.get('SHARES','2048')),nseg); per=nseg//G; rem=nseg%G
sh=int(os.environ.get('SHARE', 14 if os.environ.get('PAIR','1')=='1' else 7))   # (pair form: workgroup 7's wave 0 is share 14)
first=sh*per+min(sh,rem)
OR=ev.OR
d=fx.ravel()[first*OR:first*OR+13].astype(int)
names=["P0 gather, tables","P1 cardinal f_save","P2 interior fjgh","P3 cardinal fjgh_load","record loads landed","segment 0","seg 1: R1 rows, CL/WL","seg 1: R2 DI fragments","seg 1: R3 J product + stores","seg 1: R4 H tile columns + stores","remaining segments"]
print('segments per wave', per, 'total cycles', d[:len(names)].sum())
for n,v in zip(names,d): print(f"  {n:34s} {v:7d} cycles")
