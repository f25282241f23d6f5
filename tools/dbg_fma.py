import sys, os, numpy as np
np.set_printoptions(linewidth=200, precision=3)
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload, rel_err
from asset_asrl_amd.evaluator import *
from oracle import bindings as ob
cases=[tuple(c.split(',')) for c in sys.argv[1:]] or [("reentry","LGL7","67","0")]
for ode,mode,nseg,blocked in cases:
    nseg=int(nseg); blocked=bool(int(blocked))
    w=Workload(ode,mode,nseg,blocked,var_offset=3,con_offset=2,extra_vars=4)
    nlp=w.oracle_nlp(ob)
    ev=DefectEvaluator(ode,mode,w.blocked,w.vindex,w.cindex,w.n_primal,w.n_equal)
    for what in (4,0,1,2,3,4):
        ref=nlp.eval_blocks(what,w.X,w.L)
        got=ev.eval(what,w.X,w.L if what in (1,3,4) else None)
        e=[np.abs(got[0]-ref[0]).max()]
        e.append(None if got[1] is None else rel_err(got[1],ref[1]))
        e.append(None if got[2] is None else rel_err(got[2],ref[2]))
        print(ode,mode,nseg,blocked,'what',what,e)
        if got[2] is not None and e[2]>1e-9:
            H,J=unpack_kkt_block(got[2][0],w.IR,w.OR); Hr,Jr=unpack_kkt_block(ref[2][0],w.IR,w.OR)
            print(' J err rows', np.abs(J-Jr).max(axis=1)); print(' J err cols', np.abs(J-Jr).max(axis=0))
            print(' H err cols', np.abs(H-Hr).max(axis=0)); print(' H err rows', np.abs(H-Hr).max(axis=1))
        if got[1] is not None and e[1]>1e-9: print(' agx err', np.abs(got[1][0]-ref[1][0]))
