"""Times one workload's evaluation on the GPU and checks it against the oracle.
usage: [ASSET_HIP_LIB=..] [ASSET_HIP_TUNING=1 ASSET_HIP_SKIP_DENSE=1] python tests/timing_blocks.py [ode mode nseg blocked what]"""
import sys, os, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives under tests/: it checks against the oracle)
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload, rel_err
from asset_asrl_amd.evaluator import DefectEvaluator
ode=sys.argv[1] if len(sys.argv)>1 else "reentry"; mode=sys.argv[2] if len(sys.argv)>2 else "LGL7"
nseg=int(sys.argv[3]) if len(sys.argv)>3 else 10000; blocked=bool(int(sys.argv[4])) if len(sys.argv)>4 else False
what=int(sys.argv[5]) if len(sys.argv)>5 else 4
w=Workload(ode,mode,nseg,blocked)
ev=DefectEvaluator(ode,mode,blocked,w.vindex,w.cindex,w.n_primal,w.n_equal)
IR,OR,NK=ev.IR,ev.OR,ev.KSTRIDE   # (blocks in the handle's layout)
dev=torch.device("cuda:0")
X=torch.from_numpy(w.X).to(dev); L=torch.from_numpy(w.L).to(dev)
fx=torch.zeros(nseg*OR,dtype=torch.float64,device=dev); agx=torch.zeros(nseg*IR,dtype=torch.float64,device=dev)
kkt=torch.zeros(nseg*int(os.environ.get('ASSET_FAKE_KKT',NK)),dtype=torch.float64,device=dev)
if os.environ.get('NOKKT'): kkt=None
ms=min(ev.time_device(what,X,L,fx,agx,kkt,10,200) for _ in range(3))
msg=f"{ode} {mode} nseg={nseg} what={what}: {ms*1e3:.2f} us"
if not os.environ.get("ASSET_HIP_SKIP_DENSE") and not os.environ.get("NOCHECK"):
    sys.path.insert(0,os.path.join(ROOT))
    from oracle import bindings as ob
    ws=Workload(ode,mode,min(nseg,300),blocked)
    evs=DefectEvaluator(ode,mode,blocked,ws.vindex,ws.cindex,ws.n_primal,ws.n_equal)
    got=evs.eval(what,ws.X,ws.L); ref=ws.oracle_nlp(ob,threads=4).eval_blocks(what,ws.X,ws.L)
    msg+="  err "+" ".join(f"{rel_err(g,r):.1e}" for g,r in zip(got,ref) if g is not None and r is not None)
print(msg)
