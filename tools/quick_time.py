"""Kernel time of one evaluation kind on a synthetic phase (HIP events on the handle's stream): experiments.

  ASSET_HIP_LIB=build_dbg/libexp.so python tools/quick_time.py reentry LGL7 10000 [blocked] [kind]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import Workload
from asset_asrl_amd.evaluator import DefectEvaluator

ode, mode, nseg = sys.argv[1], sys.argv[2], int(sys.argv[3])
blocked = len(sys.argv) > 4 and sys.argv[4] == "1"
kind = int(sys.argv[5]) if len(sys.argv) > 5 else 4
w = Workload(ode, mode, nseg, blocked)
ev = DefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
dev = torch.device("cuda:0")
X, L = torch.from_numpy(w.X).to(dev), torch.from_numpy(w.L).to(dev)
fx = torch.empty(nseg * ev.OR, dtype=torch.float64, device=dev)
agx = torch.empty(nseg * ev.IR, dtype=torch.float64, device=dev)
kkt = torch.empty(nseg * ev.KSTRIDE, dtype=torch.float64, device=dev)
ts = []
for rep in range(int(os.environ.get('QT_REPS', '6'))):
    k = kind & 0xFF
    ts.append(ev.time_device(kind, X, L if k in (1, 3, 4) else None, fx, agx if k in (1, 3, 4) else None,
                             kkt if k >= 2 else None, warmup=int(os.environ.get('QT_WARMUP', '5')), iters=int(os.environ.get('QT_ITERS', '200'))))
print(f"{os.environ.get('ASSET_HIP_LIB', 'default')} {ode} {mode} x{nseg} kind {kind}: " + " ".join(f"{1e3 * t:.2f}" for t in ts) + " us")
