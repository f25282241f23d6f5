"""One-off (round 6): where the fixed host overhead of a 20-step timed region goes (t0 -> enqueue done -> last event seen -> fence done)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import Workload
from asset_asrl_amd.evaluator import JAC_ADJGRAD_HESS, DefectEvaluator
w = Workload("reentry", "LGL7", 10000)
ev = DefectEvaluator("reentry", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
dev = torch.device("cuda:0")
X, L = torch.from_numpy(w.X).to(dev), torch.from_numpy(w.L).to(dev)
fx = torch.empty(10000 * ev.OR, dtype=torch.float64, device=dev); agx = torch.empty(10000 * ev.IR, dtype=torch.float64, device=dev)
kkt = torch.empty(10000 * ev.KSTRIDE, dtype=torch.float64, device=dev)
stream = torch.cuda.Stream(device=dev)
step = ev.bind_device(JAC_ADJGRAD_HESS, X, L, fx, agx, kkt, stream)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(stream):
    for _ in range(300): step()
    for rep in range(6):
        for _ in range(5): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); a.record(stream)
        for _ in range(20): step()
        t1 = time.perf_counter(); b.record(stream)
        while not b.query(): pass
        t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"enqueue {1e6*(t1-t0):.0f} us, until last event seen {1e6*(t2-t0):.0f}, fence {1e6*(t3-t2):.0f}, events {1e3*a.elapsed_time(b):.0f} us")
