#!/usr/bin/env python3
"""Headline benchmark: NLP func+Jacobian+Hessian evaluation throughput (mesh segments/s).

One "step" = one evalKKT-equivalent pass of the defect constraint over every segment of a synthetic phase
(mode JAC_ADJGRAD_HESS: value, adjoint gradient, dense Jacobian and lower-triangular adjoint Hessian blocks),
inputs (X, L, index tables) already resident in HBM.  Default workload: the north-star's 10 000-segment LGL7
phase with the Shuttle Reentry ODE (BASELINE.json configs[2] dynamics at the target's size).  With N GPUs each
rank evaluates its own 10 000-segment shard (segments are independent: no data-path collective; weak scaling).

Prints ONE JSON line (rank 0) carrying `roofline` (HBM, algorithmic bytes / HIP-event kernel time) and, at N=1,
`cpu_baseline` (the oracle's multi-threaded C++ restatement of the reference's evalKKT on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (ode, mode, segments per GPU, blocked)
    "reentry_lgl7_10k": ("reentry", "LGL7", 10000, False),
    "reentry_lgl7_5k": ("reentry", "LGL7", 5000, False),
    "betts_lgl5_1k": ("betts_lowthrust", "LGL5", 1000, False),
    "twobody_lgl5_blocked_10k": ("twobody_lt", "LGL5", 10000, True),
    "brachistochrone_lgl3_40": ("brachistochrone", "LGL3", 40, False),
    "reentry_lgl7_100k": ("reentry", "LGL7", 100000, False),
    "synthetic32_lgl7_100k": ("synthetic32", "LGL7", 100000, False),     # BASELINE configs[4] on one GPU
    "synthetic32_lgl7_12500": ("synthetic32", "LGL7", 12500, False),     # its per-GPU share on 8 GPUs
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_segment(IR, OR):
    """reads z, lam; writes fx, adjgrad, H lower triangle, J."""
    return 8 * ((IR + OR) + (OR + IR + IR * (IR + 1) // 2 + OR * IR))


def cpu_baseline(w, budget_s=15.0):
    """Oracle evalKKT-equivalent (NLPTest protocol: zero CSR values, eval, scatter) on the host cores."""
    import numpy as np

    from oracle import bindings as ob
    ob.build()
    threads = min(16, os.cpu_count() or 1)         # reference default: min(16, hw threads)
    try:
        ode = ob.get_ode(w.ode, 1)
        kind_note = "generated analytic ODE derivatives"
    except KeyError:
        ode = ob.get_ode(w.ode, 0)
        kind_note = "AD2 ODE derivatives"
    nlp = ob.Nlp(ode, ob.MODES[w.mode], w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal, threads)
    import ctypes as C
    X = np.ascontiguousarray(w.X)
    L = np.ascontiguousarray(w.L)
    FXE, AGX, vals = np.zeros(w.n_equal), np.zeros(w.n_primal), np.zeros(nlp.nnz)
    dp = C.POINTER(C.c_double)
    args = [a.ctypes.data_as(dp) for a in (X, L, FXE, AGX, vals)]

    def one():
        vals.fill(0.0)
        rc = ob.lib().oracle_nlp_eval(nlp.h, ob.JAC_ADJGRAD_HESS, *args)
        assert rc == 0
    for _ in range(2):
        one()
    t0 = time.perf_counter()
    one()
    t1 = time.perf_counter() - t0
    reps = int(max(3, min(200, budget_s / max(t1, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    dt = (time.perf_counter() - t0) / reps
    return {"value": w.nseg / dt, "unit": "segments/s", "cores": threads, "kind": "port",
            "sample": f"{reps} evalKKT-equivalents of the same {w.nseg}-segment phase "
                      f"({kind_note}, std::thread ByApplication split, CSR scatter); {dt * 1e3:.3f} ms each",
            "ms_per_eval": dt * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="reentry_lgl7_10k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs a torch.distributed launch with WORLD_SIZE={a.gpus} (got {world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("ASSET_BENCH_FORCE_DIST"):   # (the switch exercises the RCCL path with one rank)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from helpers import Workload

    from asset_asrl_amd.evaluator import JAC_ADJGRAD_HESS, DefectEvaluator

    ode, mode, nseg, blocked = WORKLOADS[a.workload]
    w = Workload(ode, mode, nseg, blocked, seed=20260723 + rank)
    ev = DefectEvaluator(ode, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal, device=local_rank)
    dev = torch.device("cuda", local_rank)
    X = torch.from_numpy(w.X).to(dev)
    L = torch.from_numpy(w.L).to(dev)
    fx = torch.empty(nseg * ev.OR, dtype=torch.float64, device=dev)
    agx = torch.empty(nseg * ev.IR, dtype=torch.float64, device=dev)
    kkt = torch.empty(nseg * ev.NKKT, dtype=torch.float64, device=dev)
    stream = torch.cuda.Stream(device=dev)   # a stream of its own: the legacy default stream adds ~3 us of implicit synchronisation per launch

    step = ev.bind_device(JAC_ADJGRAD_HESS, X, L, fx, agx, kkt, stream)   # (arguments converted once: ~1 us of host time per step)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # kernel-only duration on the handle's own stream, HIP events around the launches
    ms_kernel = ev.time_device(JAC_ADJGRAD_HESS, X, L, fx, agx, kkt, warmup=5, iters=max(20, min(a.steps, 200)))
    bseg = algorithmic_bytes_per_segment(ev.IR, ev.OR)
    achieved = nseg * bseg / (ms_kernel * 1e-3) / 1e9

    traffic = None
    prof = os.path.join(ROOT, "profiles", f"r1_{a.workload}_pmc.json")
    if os.path.exists(prof):  # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
        try:
            traffic = json.load(open(prof))["hbm"]["bytes_per_launch"]
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "NLP func+Jacobian+Hessian eval throughput (mesh segments/s)",
            "value": world * nseg * a.steps / dt,
            "unit": "segments/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{ode} ODE, {mode}, {nseg} segments per GPU"
                                   f"{', BlockConstant control' if w.blocked else ''}; evalKKT-equivalent "
                                   "(value + adjoint gradient + Jacobian + adjoint-Hessian blocks), inputs resident in HBM",
                       "name": a.workload, "IR": ev.IR, "OR": ev.OR, "kkt_slots_per_segment": ev.NKKT,
                       "congruence": "mfma_f64_16x16x4",
                       "sharding": f"{world} x {nseg} independent segments, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("lgl_defect_kernel (ODE stage) + lgl_wide_dense_kernel" if ev.IR > 100 else "lgl_defect_kernel")
                                   + ": one evaluation = ODE-stage launch + dense-stage launch (both timed)",
                         "kernel_ms": ms_kernel,
                         "algorithmic_bytes_per_segment": bseg},
        }
        if world == 1 and not a.no_cpu_baseline:
            # bounded sample: the oracle's CSR scatter needs 12 B per KKT slot on the host -- cap it at 2e8 slots
            cap = max(1, int(2e8) // ev.NKKT)
            wc = w if nseg <= cap else Workload(ode, mode, cap, blocked, seed=20260723)
            out["cpu_baseline"] = cpu_baseline(wc)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
