"""Times the on-device assembled evaluation (ODE stage + dense stage + KKT scatter) against the block evaluation.
usage: python tests/timing_assembled.py [ode mode nseg blocked]"""
import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives under tests/: it checks against the oracle)
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload, rel_err
from asset_asrl_amd.evaluator import DefectEvaluator, JAC_ADJGRAD_HESS
from oracle import bindings as ob
ode=sys.argv[1] if len(sys.argv)>1 else "reentry"; mode=sys.argv[2] if len(sys.argv)>2 else "LGL7"
nseg=int(sys.argv[3]) if len(sys.argv)>3 else 10000; blocked=bool(int(sys.argv[4])) if len(sys.argv)>4 else False
w=Workload(ode,mode,nseg,blocked)
nlp=w.oracle_nlp(ob,threads=8)
ev=DefectEvaluator(ode,mode,w.blocked,w.vindex,w.cindex,w.n_primal,w.n_equal)
locs=nlp.kkt_locations()[:nlp.num_user_kkt].reshape(nseg,ev.NKKT)
ev.set_kkt_map(locs,nlp.nnz)
dev=torch.device("cuda:0")
X=torch.from_numpy(w.X).to(dev); L=torch.from_numpy(w.L).to(dev)
fx=torch.zeros(nseg*ev.OR,dtype=torch.float64,device=dev); agx=torch.zeros(nseg*ev.IR,dtype=torch.float64,device=dev)
kkt=torch.zeros(nseg*ev.KSTRIDE,dtype=torch.float64,device=dev); vals=torch.zeros(nlp.nnz,dtype=torch.float64,device=dev)
st=torch.cuda.Stream()   # a real stream: a null handle would select the evaluator's own
def timeit(fn,n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n): fn()
    e1.record(st); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
tb=timeit(lambda: ev.eval_device(JAC_ADJGRAD_HESS,X,L,fx,agx,kkt,st))
ta=timeit(lambda: ev.eval_assembled_device(JAC_ADJGRAD_HESS,X,L,fx,agx,vals,st))
vals.zero_(); torch.cuda.synchronize()   # (the clear runs on torch's stream, the evaluation on st)
ev.eval_assembled_device(JAC_ADJGRAD_HESS,X,L,fx,agx,vals,st); torch.cuda.synchronize()
_,_,ref=nlp.eval(JAC_ADJGRAD_HESS,w.X,w.L)
print(f"{ode} {mode} nseg={nseg}: blocks {tb:.1f} us, assembled {ta:.1f} us; nnz {nlp.nnz} vs slots {nseg*ev.NKKT}; err {rel_err(vals.cpu().numpy(),ref):.1e}")
