"""Raw clock64() deltas of a -DASSET_TIMING build (workgroup 7, wave 0), as many as the kernel left: python tools/dbg_time_raw.py nseg ode mode blocked [share]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import *
nseg=int(sys.argv[1]); ode, mode, blocked = sys.argv[2], sys.argv[3], sys.argv[4] == "1"
w=Workload(ode,mode,nseg,blocked)
ev=DefectEvaluator(ode,mode,blocked,w.vindex,w.cindex,w.n_primal,w.n_equal)
for rep in range(3): fx,agx,kkt=ev.eval(4,w.X,w.L)
f=fx.ravel(); OR=ev.OR
# the stamps sit at the first segment of workgroup 7's share: find rows whose values look like cycle counts (integers > 50)
for s in range(nseg):
    d=f[s*OR:s*OR+23]
    if len(d) >= 6 and np.all(d[:6]==np.round(d[:6])) and np.all(np.abs(d[:6])>30) and np.abs(d[:6]).max()>1000:
        print('segment', s, 'deltas', [int(x) for x in d if x==round(x) and abs(x) > 0][:23], 'sum', int(sum(x for x in d if x==round(x)))); break
