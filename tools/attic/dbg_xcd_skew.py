import sys, os, numpy as np
ROOT='/root/repo'
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import *
nseg=10000
w=Workload("reentry","LGL7",nseg,False)
ev=DefectEvaluator("reentry","LGL7",False,w.vindex,w.cindex,w.n_primal,w.n_equal)
G=2048; per=nseg//G; rem=nseg%G
first=np.array([s*per+min(s,rem) for s in range(G)])
for rep in range(8):
    fx,agx,kkt=ev.eval(4,w.X,w.L)
    fx=fx.reshape(nseg,-1)
    t0=fx[first,0]; t1=fx[first,1]; xcc=fx[first,3].astype(np.int64)&15
    base=t0.min(); st=(t0-base)/100.0; en=(t1-base)/100.0
    print(rep, " ".join(f"x{x}:{np.median(st[xcc==x]):.2f}/{en[xcc==x].max():.1f}" for x in range(8)))
