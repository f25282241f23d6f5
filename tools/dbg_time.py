"""Prints the clock64() deltas a -DASSET_TIMING build leaves in FX (workgroup 7).  usage: dbg_time.py <grid>"""
import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from helpers import Workload
from asset_asrl_amd.evaluator import *
w=Workload("reentry","LGL7",10000,False)
ev=DefectEvaluator("reentry","LGL7",False,w.vindex,w.cindex,w.n_primal,w.n_equal)
for rep in range(3):
    fx,agx,kkt=ev.eval(4,w.X,w.L)
G=int(sys.argv[1]) if len(sys.argv)>1 else 1536; per=10000//G; rem=10000%G
first=7*per+min(7,rem)
print('segs per wg', per, 'deltas:', fx.ravel()[first*15:first*15+23].astype(int))
