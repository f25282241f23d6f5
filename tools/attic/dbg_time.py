"""Prints the clock64() deltas a -DASSET_TIMING build leaves in FX (workgroup 7).

  python tools/build_one.py tu_reentry_lgl4_0 build_dbg/libdbgT.so -DASSET_TIMING
  ASSET_HIP_LIB=build_dbg/libdbgT.so python tools/dbg_time.py 2048                      # dense stage (its grid)
  ASSET_HIP_TUNING=1 ASSET_HIP_SKIP_DENSE=1 ASSET_HIP_LIB=build_dbg/libdbgT.so python tools/dbg_time.py 1024   # ODE stage
"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import *
w=Workload("reentry","LGL7",10000,False)
ev=DefectEvaluator("reentry","LGL7",False,w.vindex,w.cindex,w.n_primal,w.n_equal)
for rep in range(3):
    fx,agx,kkt=ev.eval(4,w.X,w.L)
G=int(sys.argv[1]) if len(sys.argv)>1 else 2048; per=10000//G; rem=10000%G
first=7*per+min(7,rem)
d=fx.ravel()[first*15:first*15+23].astype(int)
if os.environ.get("ASSET_DBG_FUSED"):      # fused single launch (the default for this workload): ODE stage stamps first
    names=["P0 gather -> mirror","P1 cardinal f_save (+ tables)","P2 interior fjgh","copy-out of P2 + P3 cardinal fjgh",
           "copy-out of P3 and mirror issued","lane record loads issued","slot stores landed (vmcnt 0)","to the segment loop",
           "constant tiles","segment 0 (+ slot of segment 1)","D1 DI/DC tiles","time columns, FX","D2+D3 fragments, M product",
           "rank-2 rows","D4 H/J products","D5+D6 adjoint gradient, stores","remaining segments"]
elif os.environ.get("ASSET_HIP_SKIP_DENSE"):
    names=["P0 gather","P1 cardinal f_save","P2 interior fjgh + copy-out","P3 ..."]
else:
    names=["group start","constant tiles","segment 0 (+ slot of segment 1)","D1 DI/DC tiles","time columns, FX","D2+D3 fragments, M product",
           "rank-2 rows","D4 H/J products","D5+D6 adjoint gradient, stores","remaining segments"]
print('segments per workgroup', per)
for n,v in zip(names,d): print(f"  {n:34s} {v:7d} cycles")
