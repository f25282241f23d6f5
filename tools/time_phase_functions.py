"""Device time of everything a 10 000-segment Reentry-LGL7 phase hands the solver beside the defects:
the mesh-spacing, nodal-spacing and control-spline equalities the phase registers itself, a user path equality at every
state and an integral objective -- each one batched evaluator (csrc/func_kernels.h, one thread per application).

  python tools/time_phase_functions.py            # on the GPU box
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import Workload  # noqa: E402

from asset_asrl_amd import vf  # noqa: E402
from asset_asrl_amd.evaluator import JAC_ADJGRAD_HESS  # noqa: E402
from asset_asrl_amd.ode import ShuttleReentry  # noqa: E402


def main():
    nseg = 10000
    w = Workload("reentry", "LGL7", nseg)
    ph = ShuttleReentry().phase("LGL7", w.traj, nseg)
    a = vf.Arguments(6)
    x0, x1, x2, t, u0, u1 = a.tolist()
    ph.addEqualCon("Path", vf.stack([x0 * x0 + x1 * u0 - vf.sin(x2), u0 * u0 + u1 * u1 - 1.0 + t * x0 * vf.exp(-1.0 * x1)]),
                   [0, 1, 2, 5, 6, 7])
    g = vf.Arguments(2)
    ph.addIntegralObjective(g.coeff(1) * g.coeff(1) + g.coeff(0), [2, 0])
    ph.transcribe()
    dev = torch.device("cuda:0")
    X = torch.from_numpy(ph.solver_input()).to(dev)
    L = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, ph.numPhaseEqCons)).to(dev)
    Lo = torch.tensor([0.5], dtype=torch.float64, device=dev)
    evs = [("defects", ph.evaluator, L)] + [(k, e, L) for k, e in ph.phase_function_evaluators.items()] + \
          [("user path equality (2 rows x every state)", ph.equality_evaluators[0], L), ("integral objective", ph.objective_evaluators[0], Lo)]
    out = {}
    for name, ev, lam in evs:
        fx = torch.empty(ev.nseg * ev.OR, dtype=torch.float64, device=dev)
        agx = torch.empty(ev.nseg * ev.IR, dtype=torch.float64, device=dev)
        kkt = torch.empty(ev.nseg * ev.KSTRIDE, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ms = ev.time_device(JAC_ADJGRAD_HESS, X, lam, fx, agx, kkt, warmup=5, iters=100)
        out[name] = {"applications": ev.nseg, "IR": ev.IR, "OR": ev.OR, "us": round(ms * 1e3, 2),
                     "block_MB": round(ev.nseg * (ev.NKKT + ev.IR + ev.OR) * 8 / 1e6, 2)}
    # the same functions as ONE launch (Phase.function_bundle)
    bundle, members = ph.function_bundle()
    Ls = [Lo if kind == "objective" else L for kind, _ in members]
    fxs = [torch.empty(e.nseg * e.OR, dtype=torch.float64, device=dev) for _, e in members]
    agxs = [torch.empty(e.nseg * e.IR, dtype=torch.float64, device=dev) for _, e in members]
    kkts = [torch.empty(e.nseg * e.KSTRIDE, dtype=torch.float64, device=dev) for _, e in members]
    st = torch.cuda.Stream()
    call = bundle.bind_device(JAC_ADJGRAD_HESS, X, Ls, fxs, agxs, kkts, st)
    with torch.cuda.stream(st):
        for _ in range(20):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(200):
            call()
        e1.record(st)
        e1.synchronize()
    out["all functions beside the defects, one bundled launch"] = {"functions": len(members), "us": round(e0.elapsed_time(e1) / 200 * 1e3, 2)}
    seq = [e.bind_device(JAC_ADJGRAD_HESS, X, l, f, g, k, st) for (_, e), l, f, g, k in zip(members, Ls, fxs, agxs, kkts)]
    with torch.cuda.stream(st):
        e0.record(st)
        for _ in range(200):
            for c in seq:
                c()
        e1.record(st)
        e1.synchronize()
    out["the same, one launch per function"] = {"us": round(e0.elapsed_time(e1) / 200 * 1e3, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
