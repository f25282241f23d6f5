#!/usr/bin/env python3
"""Headline benchmark: NLP func+Jacobian+Hessian evaluation throughput (mesh segments/s).

One "step" = one evalKKT-equivalent pass of the defect constraint over every segment of ONE synthetic phase
(mode JAC_ADJGRAD_HESS: value, adjoint gradient, dense Jacobian and lower-triangular adjoint Hessian blocks),
inputs (X, L, index tables) already resident in HBM.  Default workload: the north-star's 10 000-segment LGL7
phase with the Shuttle Reentry ODE (BASELINE.json configs[2] dynamics at the target's size).

N = 1: one GPU evaluates the whole phase.
N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling of the same phase -- the segments are
sharded by the reference's ByApplication rule (IndexingData.h:117-146), every rank evaluates its shard into its own
HBM buffer, and the per-shard FX / AGX / KKT blocks are gathered to rank 0 with ONE device-to-device RCCL gather
(xGMI) inside the timed region, each step -- the exchange a host KKT system on rank 0 needs every PSIOPT iteration.
`value` counts the whole phase's segments per second WITH the exchange; the line also carries the rate without it,
the exchange time alone and every rank's kernel time.  Workload `multispacecraft_8x1250` (BASELINE.json configs[3])
deals whole phases to the ranks instead.

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
    # name: (ode, mode, segments of the phase, blocked)
    "reentry_lgl7_10k": ("reentry", "LGL7", 10000, False),                # north-star target size
    "reentry_lgl7_5k": ("reentry", "LGL7", 5000, False),                  # BASELINE.json configs[2] (initial mesh)
    "betts_lgl5_1k": ("betts_lowthrust", "LGL5", 1000, False),            # BASELINE.json configs[1]
    "twobody_lgl5_blocked_10k": ("twobody_lt", "LGL5", 10000, True),      # configs[3] dynamics as one phase
    "twobody_lgl5_blocked_100k": ("twobody_lt", "LGL5", 100000, True),    # ... on a looped mesh (a light right-hand side: single-wave workgroups)
    "brachistochrone_lgl3_40": ("brachistochrone", "LGL3", 40, False),    # configs[0]
    "twobody_lgl7_10k": ("twobody_lt", "LGL7", 10000, False),            # mid-width shapes (32 < IR < 64): IR = 40
    "betts_lgl7_5k": ("betts_lowthrust", "LGL7", 5000, False),            #                                   IR = 45
    "reentry_lgl7_100k": ("reentry", "LGL7", 100000, False),
    "reentry_lgl7_1m": ("reentry", "LGL7", 1000000, False),               # HBM-resident (8.8 GB of blocks)
    "synthetic32_lgl7_100k": ("synthetic32", "LGL7", 100000, False),      # BASELINE.json configs[4]
    "synthetic32_lgl7_12500": ("synthetic32", "LGL7", 12500, False),      # its per-GPU share on 8 GPUs
    "multispacecraft_8x1250": ("twobody_lt", "LGL5", 1250, True),         # configs[3]: 8 linked phases, dealt to the ranks
    "reentry_trap_10k": ("reentry", "Trapezoidal", 10000, False),         # north_star names "LGL/Trapezoidal": the two-node scheme
    "twobody_trap_blocked_10k": ("twobody_lt", "Trapezoidal", 10000, True),
}
MULTI_PHASE = {"multispacecraft_8x1250": 8}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
FP64_PEAK_GFLOPS = 78600.0  # MI355X FP64 vector = FP64 matrix: 16 FMA lanes per cycle and SIMD (half the 157.3 TFLOP/s FP32 vector rate of MI355X_MICROARCH.md)


def algorithmic_bytes_per_segment(IR, OR):
    """reads z, lam; writes fx, adjgrad, H lower triangle, J."""
    return 8 * ((IR + OR) + (OR + IR + IR * (IR + 1) // 2 + OR * IR))


def physical_cores() -> int:
    """Distinct cores behind the hardware threads this process may run on (SMT siblings counted once)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    seen = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                seen.add(f.read().strip())
        except OSError:
            seen.add(str(c))
    return max(1, len(seen))


def cpu_baseline(w, budget_s=14.0):
    """Oracle evalKKT-equivalent (NLPTest protocol: zero CSR values, eval, scatter) on the host cores, built the way
    the reference builds itself (-O2 -march=native -ffast-math), evaluation threads kept alive between evaluations as the
    reference's pool is (oracle/nlp.cpp: WorkerPool), at the reference's default thread count min(16, hw threads), at
    every physical core and at every hardware thread; the zero-fill of the CSR value array (PSIOPT.cpp:107, one thread)
    and the evaluation are timed separately and both reported."""
    import numpy as np

    from oracle import bindings as ob
    ob.use_native()
    hw, phys = os.cpu_count() or 1, physical_cores()
    try:
        ode = ob.get_ode(w.ode, 1)
        kind_note = "generated analytic ODE derivatives"
    except KeyError:
        ode = ob.get_ode(w.ode, 0)
        kind_note = "AD2 ODE derivatives"
    import ctypes as C
    X = np.ascontiguousarray(w.X)
    L = np.ascontiguousarray(w.L)
    dp = C.POINTER(C.c_double)
    runs = {}
    configs = [(t, b) for t in sorted({min(16, hw), min(64, phys), phys, hw}) for b in (False, True)]
    for threads, batch4 in configs:
        nlp = ob.Nlp(ode, ob.MODES[w.mode], w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal, threads)
        if batch4 and not nlp.set_batch4(True):            # four segments per pass (the reference's SuperScalar loop)
            continue
        FXE, AGX, vals = np.zeros(w.n_equal), np.zeros(w.n_primal), np.zeros(nlp.nnz)
        args = [a.ctypes.data_as(dp) for a in (X, L, FXE, AGX, vals)]
        t_zero = t_eval = 0.0

        def one():
            nonlocal t_zero, t_eval
            t0 = time.perf_counter()
            vals.fill(0.0)
            t1 = time.perf_counter()
            rc = ob.lib().oracle_nlp_eval(nlp.h, ob.JAC_ADJGRAD_HESS, *args)
            t_eval += time.perf_counter() - t1
            t_zero += t1 - t0
            assert rc == 0
        for _ in range(3):
            one()
        t0 = time.perf_counter()
        one()
        t1 = time.perf_counter() - t0
        reps = int(max(20, min(400, budget_s / len(configs) / max(t1, 1e-6))))
        t_zero = t_eval = 0.0
        t0 = time.perf_counter()
        for _ in range(reps):
            one()
        dt = (time.perf_counter() - t0) / reps
        runs[(threads, batch4)] = {"threads": threads, "segments_per_pass": 4 if batch4 else 1,
                                   "segments_per_s": w.nseg / dt, "ms_per_eval": dt * 1e3,
                                   "ms_zero_fill": t_zero / reps * 1e3, "ms_eval_only": t_eval / reps * 1e3, "reps": reps}
        del nlp
    for r in runs.values():
        r["segments_per_s_eval_only"] = w.nseg / (r["ms_eval_only"] * 1e-3)
    # `value` is the evaluation ALONE, as the reference's own NLPTest times it (NonLinearProgram.cpp:741-752: the timer starts
    # after the KKT values are zeroed); PSIOPT's convention -- the single-threaded zero-fill of the value array inside the
    # timed call (PSIOPT.cpp:107) -- is reported beside it as `with_fill`.
    best = max(runs.values(), key=lambda r: r["segments_per_s_eval_only"])
    return {"value": best["segments_per_s_eval_only"], "unit": "segments/s", "cores": best["threads"], "kind": "port",
            "eval_only": {"segments_per_s": best["segments_per_s_eval_only"], "ms": best["ms_eval_only"],
                          "convention": "NLPTest: timer around evalKKT, value array zeroed before it starts"},
            "with_fill": {"segments_per_s": best["segments_per_s"], "ms": best["ms_per_eval"], "ms_zero_fill": best["ms_zero_fill"],
                          "convention": "PSIOPT: one thread zero-fills the CSR value array, then evalKKT"},
            "cpu_model": ob.cpu_model(), "hw_threads": hw, "physical_cores": phys, "build": "g++ -O2 -march=native -ffast-math",
            "runs": list(runs.values()),
            "sample": f"evalKKT-equivalents of the same {w.nseg}-segment phase ({kind_note}, persistent worker pool, "
                      f"ByApplication split, CSR scatter; value = the evaluation alone, the zero-fill of the value array timed "
                      f"separately (with_fill); one segment per pass and four per pass in AVX registers, the reference's "
                      f"SuperScalar loop); best of (threads, segments per pass) "
                      f"{sorted((r['threads'], r['segments_per_pass']) for r in runs.values())}: "
                      f"{best['threads']} threads x {best['segments_per_pass']}, {best['ms_eval_only']:.3f} ms each "
                      f"({best['ms_per_eval']:.3f} ms with the fill)",
            "ms_per_eval": best["ms_eval_only"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="reentry_lgl7_10k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inprocess-shards", type=int, default=0,
                    help="N >= 1: also time the constraint as N device handles in THIS process (the C ABI's asset_hip_defect_create_sharded, "
                         "shard k on device k mod the visible devices): host-visible rates of the blocks and of the assembled values")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started plainly with --gpus N: launch the ranks ourselves -- fresh child processes of a parent that has not touched the
        # GPU (nothing above imports torch or the HIP library) -- and relay rank 0's JSON line.  Never an exec.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        js = [ln for ln in lines if ln.lstrip().startswith("{")]
        for ln in lines:
            if not js or ln is not js[-1]:
                print(ln, file=sys.stderr)
        if js:
            print(js[-1], flush=True)              # the JSON line last
        raise SystemExit(r.returncode if (r.returncode or js) else 1)
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} under a torch.distributed launch needs WORLD_SIZE={a.gpus} (got {world})")

    import numpy as np
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    backend = os.environ.get("ASSET_BENCH_BACKEND", "nccl")     # "gloo": functional runs of N ranks that share devices
    if backend != "nccl":                                        # (RCCL refuses two ranks on one GPU); never for a reported number
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or bool(os.environ.get("ASSET_BENCH_FORCE_DIST"))   # (the switch exercises the RCCL path with one rank)
    if use_dist:
        import torch.distributed as dist
        if world == 1:                                   # forced single-rank run of the RCCL path
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    # CPU-side barriers of the host-visible exchange (no device work to wait for: a gloo group next to the RCCL one)
    gloo_group = None
    if dist is not None:
        try:
            gloo_group = dist.new_group(backend="gloo") if backend == "nccl" else None
        except Exception as exc:                           # (no usable interface for gloo: barrier on the RCCL group instead)
            print(f"[bench] gloo group unavailable ({exc}); host-visible barriers use the RCCL group", file=sys.stderr)

    from helpers import Workload

    from asset_asrl_amd.distributed import PhaseShardedEvaluator, ShardedDefectEvaluator
    from asset_asrl_amd.evaluator import (CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, KEEP_HESSIAN_SLOTS,
                                          DefectEvaluator)

    ode, mode, nseg, blocked = WORKLOADS[a.workload]
    nphases = MULTI_PHASE.get(a.workload, 1)
    # every rank builds the same phase (same seed): the solver vectors X and L are replicated, as a host solver would
    # broadcast them each iteration (8 * (n_primal + n_equal) bytes: 1.8 MB for the default workload)
    if nphases == 1:
        w = Workload(ode, mode, nseg, blocked)
        n_primal, n_equal, Xh, Lh = w.n_primal, w.n_equal, w.X, w.L
        total_segments = nseg
    else:
        # configs[3]: linked phases in ONE solver vector, placed by the OptimalControlProblem mirror
        # (asset_asrl_amd/ocp.py; OptimalControlProblem.cpp:115-155): defects only, no mesh-spacing rows
        from asset_asrl_amd.ocp import OptimalControlProblem
        from asset_asrl_amd.ode import ODE_LIBRARY
        ocp, ws = OptimalControlProblem(), []
        for k in range(nphases):
            wk = Workload(ode, mode, nseg, blocked, seed=100 + k)
            ph = ODE_LIBRARY[ode]().phase(mode, wk.traj, nseg)
            ph.setControlMode("BlockConstant" if blocked else "NoSpline")
            ph.EnableMeshSpacing = False
            ocp.addPhase(ph)
            ws.append(wk)
        phase_tables = ocp.defect_tables()
        n_primal, n_equal = ocp.n_primal, ocp.n_equal
        Xh = ocp.solver_input()
        Lh = np.concatenate([wk.L[:wk.indexer.numPhaseEqCons] for wk in ws])
        w = ws[0]
        total_segments = nphases * nseg
    blocked = w.blocked
    X = torch.from_numpy(Xh).to(dev)
    L = torch.from_numpy(Lh).to(dev)
    stream = torch.cuda.Stream(device=dev)   # a stream of its own: the legacy default stream adds ~3 us of implicit synchronisation per launch

    exchange = None
    if use_dist or nphases > 1:
        if nphases == 1:
            sh = ShardedDefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, n_primal, n_equal, rank=rank, world=world,
                                        device=local_rank)
            local_segments = sh.count
            evs = [sh.ev] if sh.ev is not None else []
        else:
            sh = PhaseShardedEvaluator(ode, mode, blocked, phase_tables, n_primal, n_equal,
                                       rank=rank, world=world, device=local_rank)
            local_segments = len(sh.mine) * nseg
            evs = [sh.ev] if sh.ev is not None else []
        sh.alloc_device(dev, always_exchange=use_dist)
        IR, OR, NKKT, KSTRIDE = sh.IR, sh.OR, sh.NKKT, sh.KSTRIDE

        def evaluate():
            sh.eval_device(JAC_ADJGRAD_HESS, X, L, stream)

        def exchange():
            sh.gather_device()

        def step():
            evaluate()
            exchange()
    else:
        ev = DefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, n_primal, n_equal, device=local_rank)
        evs, local_segments = [ev], nseg
        IR, OR, NKKT, KSTRIDE = ev.IR, ev.OR, ev.NKKT, ev.KSTRIDE    # (KSTRIDE: doubles per block in the handle's layout, >= NKKT)
        fx = torch.empty(nseg * OR, dtype=torch.float64, device=dev)
        agx = torch.empty(nseg * IR, dtype=torch.float64, device=dev)
        kkt = torch.empty(nseg * KSTRIDE, dtype=torch.float64, device=dev)
        step = evaluate = ev.bind_device(JAC_ADJGRAD_HESS, X, L, fx, agx, kkt, stream)   # (arguments converted once)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    last_event_ms = [0.0]

    def timed(fn, steps, warmup):
        with torch.cuda.stream(stream):
            for _ in range(warmup):
                fn()
            fence()
            t0 = time.perf_counter()
            ev_a.record(stream)              # HIP events on the stream the kernels are launched on, around the same K steps
            for _ in range(steps):
                fn()
            ev_b.record(stream)
            while not ev_b.query():          # (poll for the last step: a blocking wait adds the scheduler's wake-up to the host clock -- 40-140 us,
                pass                         #  2-7 us per step of the driver's 20-step run; the fence below then returns at once)
            fence()
            dt = time.perf_counter() - t0
            last_event_ms[0] = ev_a.elapsed_time(ev_b)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # Kernel-only duration of this rank's share on the handle's own stream, HIP events around the launches -- measured
    # FIRST and until it settles: the part clocks up under load (the same kernel: 49.8 us over the first 100 launches after
    # an idle second, 45.2 after 300, 42.6 after 900), so the rounds below are repeated until two agree within 1 %.  The
    # timed region of the contract follows immediately, on a device that is already at its working clocks.
    ms_kernel, kernel_rounds, ms_con, ms_rhs, ms_soe, ms_aug, ms_soe_keep, ms_aug_keep = 0.0, [], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0
    if evs:
        e0 = evs[0]
        n0 = e0.nseg
        kfx = torch.empty(n0 * OR, dtype=torch.float64, device=dev)
        kagx = torch.empty(n0 * IR, dtype=torch.float64, device=dev)
        kkkt = torch.empty(n0 * KSTRIDE, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        iters = 200 if n0 * (OR + IR + NKKT) * 8 < (1 << 30) else 20
        for _ in range(8):
            kernel_rounds.append(e0.time_device(JAC_ADJGRAD_HESS, X, L, kfx, kagx, kkkt, warmup=5, iters=iters))
            if len(kernel_rounds) >= 2 and abs(kernel_rounds[-1] - kernel_rounds[-2]) <= 0.01 * kernel_rounds[-1]:
                break
        ms_kernel = kernel_rounds[-1]
        # secondary number: the value-only kind, evalOCC / CON -- what a line search calls
        ms_con = e0.time_device(CON, X, None, kfx, None, None, warmup=5, iters=iters)
        ms_rhs = e0.time_device(CON_ADJGRAD, X, L, kfx, kagx, None, warmup=5, iters=iters)   # evalRHS: value + J^T lam
        ms_soe = e0.time_device(JAC, X, None, kfx, None, kkkt, warmup=5, iters=iters)         # evalSOE: value + Jacobian blocks
        ms_aug = e0.time_device(JAC_ADJGRAD, X, L, kfx, kagx, kkkt, warmup=5, iters=iters)    # evalAUG
        # the same two with the Hessian slots of the blocks left untouched (KKTFillJac never reads them)
        ms_soe_keep = e0.time_device(JAC | KEEP_HESSIAN_SLOTS, X, None, kfx, None, kkkt, warmup=5, iters=iters)
        ms_aug_keep = e0.time_device(JAC_ADJGRAD | KEEP_HESSIAN_SLOTS, X, L, kfx, kagx, kkkt, warmup=5, iters=iters)
        del kfx, kagx, kkkt

    dt = timed(step, a.steps, a.warmup)                       # THE measurement: K steps, exchange included when N > 1
    ms_timed_region = last_event_ms[0] / a.steps              # average launch-to-launch duration over the timed region
    extra = {}
    if exchange is not None and use_dist:
        k2 = max(10, min(a.steps, 200))
        extra["ms_per_step_without_exchange"] = timed(evaluate, k2, 5) / k2 * 1e3
        extra["value_without_exchange"] = total_segments / (extra["ms_per_step_without_exchange"] * 1e-3)
        extra["exchange_ms"] = timed(exchange, k2, 5) / k2 * 1e3
        extra["exchange_bytes_into_root"] = 8 * max(world - 1, 1) * (sh.slot_doubles if nphases == 1 else sh._flat_doubles(sh.per_rank))

    # host-visible rate (NOT `value`): the blocks in host memory, where the reference's solver keeps its KKT system.  Every
    # rank copies its shard over its own PCIe link into its range of one page-locked host buffer all ranks map
    # (asset_asrl_amd/distributed.py: HostSharedBlocks); the step ends at a barrier.  One GPU: one link.
    host_visible = None
    if nphases == 1 and nseg * (OR + IR + NKKT) * 8 <= 4 << 30:
        k3 = max(5, min(a.steps, 50))
        ok = 1
        if use_dist:
            try:
                sh.alloc_host_shared(barrier_group=gloo_group)
            except Exception as exc:                      # (e.g. no /dev/shm, page-locking refused): report, do not lose the run
                ok, host_visible = 0, {"error": f"shared host buffer unavailable: {exc}"}
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # every rank takes the same branch
            ok = int(flag.item())

            def host_step():
                evaluate()
                sh.push_host()
                sh.wait_host(stream)
            bytes_rank = sh.slot_doubles * 8
        else:
            hfx, hagx, hkkt = (torch.empty(t.numel(), dtype=torch.float64, pin_memory=True) for t in (fx, agx, kkt))

            def host_step():
                evaluate()
                hfx.copy_(fx, non_blocking=True)
                hagx.copy_(agx, non_blocking=True)
                hkkt.copy_(kkt, non_blocking=True)
                stream.synchronize()
            bytes_rank = (fx.numel() + agx.numel() + kkt.numel()) * 8
        if ok:
            th = timed(host_step, k3, 2) / k3
            host_visible = {"ms_per_step": th * 1e3, "segments_per_s": total_segments / th, "bytes_per_rank": bytes_rank,
                            "path": "evaluation, then every rank's FX/AGX/KKT blocks device-to-host over its own PCIe link into "
                                    "one page-locked host buffer shared by the ranks; barrier"}
            if use_dist:
                sh._host.close()
        elif host_visible is None:
            host_visible = {"error": "shared host buffer unavailable on another rank"}

    # the same with the KKT entries ASSEMBLED on the device(s) (SURVEY section 8 rows f-1 x e): every rank sums its shard into a
    # compact value array and pushes it into ONE shared page-locked value array -- nnz values instead of nseg * NKKT block
    # slots, no host-side scatter.  N = 1: asset_hip_defect_eval_assembled_zeroed's path on device pointers + one copy.
    host_visible_assembled = None
    if nphases == 1 and host_visible is not None and "error" not in host_visible and nseg * NKKT <= 60_000_000:
        from asset_asrl_amd.indexing import kkt_slot_locations
        locs, nnz = kkt_slot_locations(w.vindex, w.cindex, n_primal)
        k3 = max(5, min(a.steps, 50))
        if use_dist:
            sh.set_kkt_map(locs, nnz).alloc_assembled(dev, barrier_group=gloo_group)

            def asm_step():
                sh.eval_assembled_device(JAC_ADJGRAD_HESS, X, L, stream)
                sh.push_assembled()
                sh.wait_assembled(stream)
            bytes_rank = (sh._asm_locs[rank].size + sh._fa.numel()) * 8
        else:
            ev.set_kkt_map(locs, nnz)
            dvals = torch.zeros(nnz, dtype=torch.float64, device=dev)
            hvals = torch.empty(nnz, dtype=torch.float64, pin_memory=True)

            def asm_step():
                dvals.zero_()
                ev.eval_assembled_device(JAC_ADJGRAD_HESS, X, L, fx, agx, dvals, stream)
                hvals.copy_(dvals, non_blocking=True)
                hfx.copy_(fx, non_blocking=True)
                hagx.copy_(agx, non_blocking=True)
                stream.synchronize()
            bytes_rank = (nnz + fx.numel() + agx.numel()) * 8
        th = timed(asm_step, k3, 2) / k3
        host_visible_assembled = {"ms_per_step": th * 1e3, "segments_per_s": total_segments / th, "bytes_per_rank": bytes_rank,
                                  "kkt_values": nnz, "block_slots": nseg * NKKT,
                                  "path": "every rank assembles its shard's KKT entries on its device and pushes the values over its own "
                                          "PCIe link into one shared page-locked value array (+ FX / AGX blocks); the root adds the few "
                                          "entries shard boundaries share; barrier"}
        if use_dist:
            sh._hostv.close()

    # the constraint as N device handles in ONE process, through the C ABI alone (asset_hip_defect_create_sharded: the reference's
    # thread_split with a device per chunk; no RCCL, no second process): host-visible rates, outputs page-locked once (what a solver's
    # RHS / KKT arrays are) and -- for contrast -- fresh pageable outputs every call (one host thread per shard inside the library)
    inprocess = None
    if a.inprocess_shards > 0 and world == 1 and nphases == 1 and nseg * (OR + IR + KSTRIDE) * 8 <= 4 << 30:
        from asset_asrl_amd.evaluator import ShardedDefectEvaluator as InProcessShards
        from asset_asrl_amd.indexing import kkt_slot_locations
        ndev = torch.cuda.device_count()
        devices = [k % ndev for k in range(a.inprocess_shards)]
        k4 = max(5, min(a.steps, 30))

        def wall(fn, n=k4):
            fn(); fn()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            return (time.perf_counter() - t0) / n
        inprocess = {"shards": a.inprocess_shards, "devices": devices,
                     "path": "asset_hip_sharded_eval / _eval_assembled: X, L to every shard's device, all shards enqueued, every shard's slice of the "
                             "blocks (or its range of the assembled values) over its device's PCIe link into the caller's host arrays; wall clock per call"}
        for nsh in sorted({1, a.inprocess_shards}):
            se = InProcessShards(ode, mode, blocked, w.vindex, w.cindex, n_primal, n_equal, devices[:nsh])
            t_pageable = wall(lambda: se.eval(JAC_ADJGRAD_HESS, Xh, Lh, native=True), max(3, k4 // 3))
            se.pin_outputs()
            t_pinned = wall(lambda: se.eval(JAC_ADJGRAD_HESS, Xh, Lh))
            entry = {"blocks_pinned_ms": t_pinned * 1e3, "blocks_pinned_segments_per_s": nseg / t_pinned,
                     "blocks_pageable_ms": t_pageable * 1e3, "bytes": nseg * (OR + IR + KSTRIDE) * 8}
            if nseg * NKKT <= 60_000_000:
                locs_s, nnz_s = kkt_slot_locations(w.vindex, w.cindex, n_primal)
                se.set_kkt_map(locs_s, nnz_s)
                hv = torch.zeros(nnz_s, dtype=torch.float64, pin_memory=True).numpy()

                def asm():       # (accumulates, as the C ABI says: the solver's own zero-fill of its value array is not this function's time)
                    se.eval_assembled(JAC_ADJGRAD_HESS, Xh, Lh, hv)
                t_asm = wall(asm)
                entry.update({"assembled_ms": t_asm * 1e3, "assembled_segments_per_s": nseg / t_asm, "kkt_values": nnz_s})
            inprocess[f"{nsh}_shard{'s' if nsh > 1 else ''}"] = entry
            se.close()

    bseg = algorithmic_bytes_per_segment(IR, OR)
    # roofline: the launch duration over the TIMED REGION (HIP events around the K steps, N = 1: one evaluation per step,
    # nothing else on the stream); with an exchange in the step (N > 1) the kernel's own settled HIP-event figure
    ms_roof = ms_timed_region if (world == 1 and exchange is None) else ms_kernel
    achieved = local_segments * bseg / (ms_roof * 1e-3) / 1e9 if ms_roof > 0 else 0.0
    per_rank_ms, per_rank_segments = [ms_kernel], [local_segments]
    if dist is not None:
        t = torch.tensor([ms_kernel, float(local_segments)], dtype=torch.float64, device=dev)
        allk = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allk, t)
        per_rank_ms = [float(x[0].item()) for x in allk]
        per_rank_segments = [int(x[1].item()) for x in allk]
    # every rank's own kernel against ITS roofline: algorithmic bytes of its share / its settled HIP-event kernel time
    per_rank_frac = [(n * bseg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else 0.0 for n, ms in zip(per_rank_segments, per_rank_ms)]

    # FP64 side of the roofline (SURVEY section 8d "MFMA/FP64-vector bound only checked for config 5"): useful flops of the node-wise
    # sparse form (asset_asrl_amd/flopcount.py) against the FP64 vector peak, the arithmetic intensity against the ridge, and -- from the
    # committed counter profile -- the flops the kernels ISSUE if every vector instruction were a 64-lane FMA (an upper bound)
    fp64 = None
    if mode != "Trapezoidal":
        from asset_asrl_amd import synth
        from asset_asrl_amd.flopcount import sparse_useful_flops
        from asset_asrl_amd.ode import ODE_LIBRARY
        fl = sparse_useful_flops(ODE_LIBRARY[ode]().derivatives(), synth.MODE_CS[mode], blocked)
        ach = local_segments * fl["total"] / (ms_roof * 1e-3) / 1e9 if ms_roof > 0 else 0.0
        fp64 = {"bound": "fp64_valu", "peak": FP64_PEAK_GFLOPS, "unit": "GFLOP/s", "achieved": ach, "frac": ach / FP64_PEAK_GFLOPS,
                "useful_flops_per_segment": fl["total"], "useful_flops_algebra": fl["algebra"], "useful_flops_ode": fl["ode"],
                "flop_per_algorithmic_byte": fl["total"] / bseg, "ridge_flop_per_byte": FP64_PEAK_GFLOPS / HBM_PEAK_GBS,
                "dense_reference_flops_per_segment": fl["dense_survey"],
                "note": "useful = FMAs (x 2) of the node-wise sparse form of LGLDefects.h:414-512 for the entries a block holds, with the ODE's "
                        "structural sparsity, + (CS + K) ODE bodies; dense_reference = SURVEY 8(d)'s formula for the reference's dense products"}

    traffic, traffic_src = None, None
    prof = next((q for q in (os.path.join(ROOT, "profiles", f"r{r}_{a.workload}_pmc.json") for r in (6, 5, 4, 3, 2)) if os.path.exists(q)), "")
    if world == 1 and prof:   # NOT measured by this run: HBM bytes per evaluation from the committed rocprofv3 --pmc passes
        try:
            pj = json.load(open(prof))
            traffic = pj["hbm"]["bytes_per_launch"]
            traffic_src = f"committed profile profiles/{os.path.basename(prof)} (separate rocprofv3 --pmc passes of this command)"
            if fp64 is not None:
                valu = sum(v.get("SQ_INSTS_VALU", 0.0) for k, v in pj.get("sq_counters_per_dispatch", {}).items() if k != "secondary")
                if valu > 0:
                    fp64["issued_flops_per_segment_upper_bound"] = valu * 128.0 / nseg
                    fp64["issued_source"] = (f"SQ_INSTS_VALU of the evaluation's kernels x 64 lanes x 2 (profiles/{os.path.basename(prof)}): what the "
                                             "vector ALU would do if every instruction were a full-wave FMA")
        except Exception:
            traffic = None

    if rank == 0:
        phases_note = f"{nphases} linked phases x {nseg} segments" if nphases > 1 else f"{nseg} segments"
        if world == 1:
            sharding = "one GPU evaluates the whole phase"
        elif nphases > 1:
            sharding = (f"{nphases} phases dealt round-robin to {world} GPUs; per step one device-to-device RCCL gather of "
                        "every rank's FX/AGX/KKT blocks to rank 0")
        else:
            sharding = (f"one phase, contiguous segment ranges on {world} GPUs (ByApplication rule); per step one "
                        "device-to-device RCCL gather of every shard's FX/AGX/KKT blocks to rank 0")
        out = {
            "metric": "NLP func+Jacobian+Hessian eval throughput (mesh segments/s)",
            "value": total_segments * a.steps / dt,
            "unit": "segments/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{ode} ODE, {mode}, {phases_note}"
                                   f"{', BlockConstant control' if blocked else ''}; evalKKT-equivalent "
                                   "(value + adjoint gradient + Jacobian + adjoint-Hessian blocks), inputs resident in HBM",
                       "name": a.workload, "IR": IR, "OR": OR, "kkt_slots_per_segment": NKKT, "kkt_block_stride": KSTRIDE,
                       "total_segments": total_segments, "sharding": sharding,
                       # which of the line's rates the strong-scaling target of BASELINE.json (>= 6x at 8 GPUs) is claimed on
                       "scaling_claim": ("none at this size: a 10 000-segment phase leaves 1 250 segments per GPU, one latency-bound "
                                         "launch (~17 us floor against 26 us on one GPU) -- `value_without_exchange` can reach ~2.4x, and "
                                         "`value` carries the gather of 84 MB of blocks into one GPU on top; the >= 6x target is claimed on "
                                         "`value_without_exchange` (and `host_visible_assembled`, every rank over its own PCIe link) of the "
                                         "large meshes: --workload reentry_lgl7_1m / synthetic32_lgl7_100k (tools/bench_scale.sh)"
                                         if total_segments < 50000 else
                                         "`value_without_exchange`: every GPU evaluates its contiguous range of the segments, nothing crosses "
                                         "xGMI inside the evaluation; `value` adds the device-to-device gather of all blocks into rank 0, which a "
                                         "host-resident solver does not need (`host_visible_assembled`: every rank pushes its assembled values "
                                         "over its own PCIe link)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": ("lgl_defect_kernel (ODE stage) + lgl_rows_kernel (dense stage by output rows; LGL3: lgl_wide_dense_kernel)" if IR >= 64 else
                                    "lgl_resident_kernel (one launch, ODE results resident in LDS; dense part as matrix-instruction tiles, or by "
                                    "output rows with DPP-broadcast operands for the shapes the tiles would pad: defect_rowdpp.h) for the narrow LGL shapes; otherwise "
                                    "lgl_defect_kernel (fused launch, or ODE-stage + dense-stage launches; ODE stage in units for heavy ODEs)")
                                   + "; rank 0's share, all launches of one evaluation timed",
                         "launch_ms": ms_roof, "launch_ms_source": ("HIP events around the K timed steps / K" if ms_roof is ms_timed_region
                                                                     else "settled HIP-event rounds (the step also holds the exchange)"),
                         "kernel_ms": ms_kernel, "kernel_ms_rounds": kernel_rounds, "segments_in_kernel": local_segments,
                         "algorithmic_bytes_per_segment": bseg, "fp64": fp64},
            "per_rank_kernel_ms": per_rank_ms, "per_rank_segments": per_rank_segments, "per_rank_roofline_frac": per_rank_frac,
            "secondary": {"evalOCC (CON: defect values only), rank 0's share, kernel": {
                "ms": ms_con, "segments_per_s": local_segments / (ms_con * 1e-3) if ms_con > 0 else 0.0,
                "algorithmic_bytes_per_segment": 8 * (IR + OR) + 8 * OR},
                "evalRHS (CON_ADJGRAD: values + adjoint gradient, no Jacobian formed), rank 0's share, kernel": {
                "ms": ms_rhs, "segments_per_s": local_segments / (ms_rhs * 1e-3) if ms_rhs > 0 else 0.0,
                "algorithmic_bytes_per_segment": 8 * (IR + OR) + 8 * (OR + IR)},
                "evalSOE (JAC: values + Jacobian blocks, Hessian slots zero), kernel ms": ms_soe,
                "evalAUG (JAC_ADJGRAD), kernel ms": ms_aug,
                "evalSOE with the Hessian slots left untouched (JAC | KEEP_HESSIAN_SLOTS), kernel ms": ms_soe_keep,
                "evalAUG with the Hessian slots left untouched (JAC_ADJGRAD | KEEP_HESSIAN_SLOTS), kernel ms": ms_aug_keep},
        }
        out.update(extra)
        if host_visible is not None:
            out["host_visible"] = host_visible
        if host_visible_assembled is not None:
            out["host_visible_assembled"] = host_visible_assembled
        if inprocess is not None:
            out["inprocess_shards"] = inprocess
        if world == 1 and not a.no_cpu_baseline:
            # bounded sample: the oracle's CSR scatter needs 12 B per KKT slot on the host -- cap it at 2e8 slots
            cap = max(1, int(2e8) // NKKT)
            wc = w if nseg <= cap else Workload(ode, mode, cap, blocked)
            out["cpu_baseline"] = cpu_baseline(wc)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        line = json.dumps(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line must be the LAST line of stdout: the RCCL loader printf()s lines of its own, which sit in the C
        # library's buffer (stdout is a pipe) until the process exits unless they are flushed first
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == "__main__":
    main()
