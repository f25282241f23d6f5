import sys, os, numpy as np
np.set_printoptions(linewidth=200, precision=6)
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import *
from asset_asrl_amd import ode as odelib
from asset_asrl_amd.vf.ir import evaluate
from oracle import bindings as ob
w=Workload("reentry","LGL7",8,False)
ev=DefectEvaluator("reentry","LGL7",False,w.vindex,w.cindex,w.n_primal,w.n_equal)
d=odelib.ShuttleReentry().derivatives()
A,B,U,s=[ob.lgl_table(4,k) for k in "ABUs"]
for what in (4,3,0):
    got=ev.eval(what,w.X,w.L if what in (1,3,4) else None)[0]
    for seg in (0,5):
        z=w.X[w.vindex[seg]]; q=8; n=5
        card=[z[j*q:(j+1)*q] for j in range(4)]
        f=[np.array(evaluate(d.f,c)) for c in card]
        h=card[3][5]-card[0][5]
        ref=[]
        for i in range(3):
            y=np.zeros(8)
            y[:5]=sum(A[i][j]*card[j][:5]+B[i][j]*h*f[j] for j in range(4))
            y[5]=card[0][5]+h*s[i]
            y[6:]=sum(U[i][j]*card[j][6:] for j in range(4))
            ref+=list(evaluate(d.f,y))
        print('what',what,'seg',seg,'If err',np.abs(got[seg]-np.array(ref)))
