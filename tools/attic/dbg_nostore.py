import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import Workload
from asset_asrl_amd.evaluator import DefectEvaluator
ode, mode, n = (sys.argv[1], sys.argv[2], int(sys.argv[3])) if len(sys.argv) > 3 else ("reentry", "LGL7", 10000)
w = Workload(ode, mode, n, False)
ev = DefectEvaluator(ode, mode, False, w.vindex, w.cindex, w.n_primal, w.n_equal)
dev = torch.device("cuda:0")
X, L = torch.from_numpy(w.X).to(dev), torch.from_numpy(w.L).to(dev)
fx = torch.empty(n * ev.OR, dtype=torch.float64, device=dev)
agx = torch.empty(n * ev.IR, dtype=torch.float64, device=dev)
kkt = torch.empty(n * ev.KSTRIDE, dtype=torch.float64, device=dev)
for name, k in (("with block stores", kkt), ("no block stores", None)):
    ts = [ev.time_device(4, X, L, fx, agx, k, warmup=5, iters=200) for _ in range(5)]
    print(name, " ".join(f"{1e3*t:.2f}" for t in ts))
