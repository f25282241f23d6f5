"""Prints the clock64() deltas a -DASSET_TIMING build of the wide dense kernel (csrc/defect_wide.h) leaves in FX
(workgroup 7, its second segment, one row per wave).

  python tools/build_one.py tu_synthetic32_lgl4_0 build_dbg/libwT.so -DASSET_TIMING
  ASSET_HIP_LIB=build_dbg/libwT.so python tools/dbg_time_wide.py [nseg]
"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import *
nseg=int(sys.argv[1]) if len(sys.argv)>1 else 2560
w=Workload("synthetic32","LGL7",nseg,False)
ev=DefectEvaluator("synthetic32","LGL7",False,w.vindex,w.cindex,w.n_primal,w.n_equal)
for rep in range(3):
    fx,agx,kkt=ev.eval(4,w.X,w.L)
G=256; per=nseg//G; rem=nseg%G
first=7*per+min(7,rem)
names=["slot","DI state rows","time columns","HI, rank-2 row","adjoint gradient","tiles of this wave","wait for the other waves"]
d=fx.ravel()[first*96:first*96+48].astype(int).reshape(4,12)
print('segments per workgroup', per)
print(f"  {'':28s}" + "".join(f"  wave {k}" for k in range(4)))
for t,nm in enumerate(names): print(f"  {nm:28s}" + "".join(f"{d[k,t]:8d}" for k in range(4)))
for t,nm in ((8,"  in H tile rows"),(9,"  in J tile chunks"),(10,"  units taken")): print(f"  {nm:28s}" + "".join(f"{d[k,t]:8d}" for k in range(4)))
print(f"  {'total':28s}" + "".join(f"{d[k,:len(names)].sum():8d}" for k in range(4)))
