"""GPU parity tests proper: the HIP path through the C ABI against the oracle on the same seeded inputs, and
against the committed golden vectors.  Tolerances are the north-star's: 1e-10 relative on residuals (relative
to the magnitude of the terms), 1e-8 relative on derivatives."""
import glob
import os

import numpy as np
import pytest

from asset_asrl_amd import _lib
from asset_asrl_amd.evaluator import (CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, DefectEvaluator,
                                      unpack_kkt_block)
from helpers import Workload, rel_err

pytestmark = pytest.mark.gpu

TOL_RES, TOL_DER = 1e-10, 1e-8


def _check_blocks(got, ref, w, what):
    fx, agx, kkt = got
    rfx, ragx, rkkt = ref
    scale = max(1.0, float(np.abs(w.X).max()))
    assert np.abs(fx - rfx).max() / scale < TOL_RES
    if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS):
        assert rel_err(agx, ragx) < TOL_DER
    if what >= JAC:
        assert kkt.shape == rkkt.shape
        assert rel_err(kkt, rkkt) < TOL_DER
        # per-entry check relative to the block's own scale (catches a wrong small entry hidden by a big one)
        bs = np.maximum(1.0, np.abs(rkkt).max(axis=1, keepdims=True))
        assert (np.abs(kkt - rkkt) / bs).max() < TOL_DER


CONFIGS = [
    ("brachistochrone", "LGL3", 40, False),      # BASELINE config 0
    ("betts_lowthrust", "LGL5", 100, False),     # config 1 shape (1000 segments in test_gpu_full_size)
    ("reentry", "LGL7", 257, False),             # config 2 shape, ragged vs the group size
    ("twobody_lt", "LGL5", 75, True),            # config 3: BlockConstant
    ("reentry", "LGL3", 64, False),
    ("reentry", "LGL5", 31, False),
    ("twobody_lt", "LGL7", 33, False),
    ("betts_lowthrust", "LGL3", 17, True),
    ("brachistochrone", "LGL7", 5, True),
    ("synthetic32", "LGL3", 9, False),
    ("reentry", "Trapezoidal", 130, False),
    ("twobody_lt", "Trapezoidal", 33, True),
    ("betts_lowthrust", "Trapezoidal", 21, False),
    ("synthetic32", "Trapezoidal", 5, False),
    ("synthetic32", "LGL5", 7, False),
    ("synthetic32", "LGL7", 6, False),           # config 4 shape: the wide-shape dense path (Dims::WIDE)
]


def _all_kernels():
    out = []
    for ode in ("brachistochrone", "reentry", "twobody_lt", "betts_lowthrust", "synthetic32"):
        for mode in ("Trapezoidal", "LGL3", "LGL5", "LGL7"):
            for blocked in (False, True):
                out.append((ode, mode, blocked))
    return out


@pytest.mark.parametrize("ode,mode,blocked", _all_kernels())
def test_every_compiled_instantiation(oracle, ode, mode, blocked):
    """One small full evaluation per (ODE, transcription, control mode) kernel instantiation."""
    if not _lib.has_kernel(ode, _lib.MODES[mode], blocked):
        pytest.skip("not instantiated")
    w = Workload(ode, mode, 19, blocked, seed=7)
    ev = DefectEvaluator(ode, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    _check_blocks(ev.eval(JAC_ADJGRAD_HESS, w.X, w.L), w.oracle_nlp(oracle, threads=2).eval_blocks(oracle.JAC_ADJGRAD_HESS, w.X, w.L),
                  w, JAC_ADJGRAD_HESS)
    ev.close()


@pytest.mark.parametrize("ode,mode,nseg,blocked", CONFIGS)
def test_all_evaluation_kinds_match_oracle(oracle, ode, mode, nseg, blocked):
    w = Workload(ode, mode, nseg, blocked, var_offset=3, con_offset=2, extra_vars=4)
    nlp = w.oracle_nlp(oracle, threads=4)
    ev = DefectEvaluator(ode, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    assert (ev.IR, ev.OR, ev.NKKT) == (w.IR, w.OR, w.NKKT)
    for what in (JAC_ADJGRAD_HESS, CON, CON_ADJGRAD, JAC, JAC_ADJGRAD):
        ref = nlp.eval_blocks(what, w.X, w.L)
        got = ev.eval(what, w.X, w.L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
        if what in (JAC, JAC_ADJGRAD):   # Hessian slots are written as zero in Jacobian-only kinds
            H, _ = unpack_kkt_block(got[2][0], w.IR, w.OR)
            assert np.abs(H).max() == 0.0
        _check_blocks(got, ref, w, what)
    ev.close()


@pytest.mark.parametrize("ode,mode,nseg,blocked", [
    ("reentry", "LGL7", 1000, False),            # fused launch, tiles held until the stores
    ("reentry", "LGL7", 14400, False),           # ODE-stage + dense-stage launches
    ("twobody_lt", "LGL7", 300, False),          # tiles stored as they complete
    ("reentry", "Trapezoidal", 500, False),
    ("synthetic32", "LGL7", 9, False)])          # four-wave dense kernel
def test_jacobian_kinds_can_leave_the_hessian_slots_untouched(ode, mode, nseg, blocked):
    """KEEP_HESSIAN_SLOTS (include/asset_hip.h): the reference's Jacobian-only fill never reads the Hessian slots of a block
    (DenseFunctionBase.h:1468-1523 KKTFillJac), so evalSOE / evalAUG need not have them written.  Device against device,
    bit for bit: the Jacobian slots are those of the plain kind, the Hessian slots still hold what the evaluation before
    left there (the handle's block buffer persists between host-pointer calls)."""
    from asset_asrl_amd.evaluator import KEEP_HESSIAN_SLOTS
    w = Workload(ode, mode, nseg, blocked)
    ev = DefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    full = [a.copy() for a in ev.eval(JAC_ADJGRAD_HESS, w.X, w.L)]
    hmask = np.zeros(w.NKKT, dtype=bool)
    k = 0
    for i in range(w.IR):
        hmask[k:k + w.IR - i] = True
        k += (w.IR - i) + w.OR
    for what in (JAC, JAC_ADJGRAD):
        L = w.L if what == JAC_ADJGRAD else None
        plain = [None if a is None else a.copy() for a in ev.eval(what, w.X, L)]
        assert np.all(plain[2][:, hmask] == 0.0)
        ev.eval(JAC_ADJGRAD_HESS, w.X, w.L)                                   # refill the buffer with Hessians ...
        kept = ev.eval(what | KEEP_HESSIAN_SLOTS, w.X, L)                      # ... which this evaluation leaves alone
        assert np.array_equal(kept[2][:, hmask], full[2][:, hmask]) and np.abs(full[2][:, hmask]).max() > 0
        assert np.array_equal(kept[2][:, ~hmask], plain[2][:, ~hmask]) and np.array_equal(kept[0], plain[0])
        if what == JAC_ADJGRAD:
            assert np.array_equal(kept[1], plain[1])
    with pytest.raises(Exception):
        ev.eval(CON | KEEP_HESSIAN_SLOTS, w.X)
    ev.close()


@pytest.mark.parametrize("nseg", [1, 2, 3, 4, 5, 63, 64, 65])
def test_ragged_segment_counts(oracle, nseg):
    w = Workload("reentry", "LGL7", nseg)
    ev = DefectEvaluator("reentry", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    _check_blocks(ev.eval(JAC_ADJGRAD_HESS, w.X, w.L), w.oracle_nlp(oracle).eval_blocks(oracle.JAC_ADJGRAD_HESS, w.X, w.L),
                  w, JAC_ADJGRAD_HESS)


@pytest.mark.parametrize("ode,mode,blocked,nseg", [
    # Reentry-LGL7 on 256 CUs x 8 workgroups: fused up to 7 segments per workgroup (14 336), two launches beyond; the
    # two-wave form takes shares of exactly 4 (GF2 = 8)
    ("reentry", "LGL7", False, 2047), ("reentry", "LGL7", False, 2049), ("reentry", "LGL7", False, 6145),
    ("reentry", "LGL7", False, 8192), ("reentry", "LGL7", False, 8193), ("reentry", "LGL7", False, 14336),
    ("reentry", "LGL7", False, 14337),
    # TwoBody-LGL5-BlockConstant: two-wave shares of 4..10 (GF2 = 20), one-wave up to GF = 15, then two launches
    ("twobody_lt", "LGL5", True, 6143), ("twobody_lt", "LGL5", True, 8193), ("twobody_lt", "LGL5", True, 20481),
    ("twobody_lt", "LGL5", True, 30719), ("twobody_lt", "LGL5", True, 30721),
    # Betts (ODE stage in units): fewer segments than one group, a ragged last group
    ("betts_lowthrust", "LGL5", False, 3), ("betts_lowthrust", "LGL5", False, 1031),
    # ... whose dense part takes its slots one segment ahead (global -> LDS): one segment per wave (nothing ahead), two, groups of
    # 3 + 2 (slot 0 of the second group asked for by the last segment of the first), three groups
    ("betts_lowthrust", "LGL7", False, 1000), ("betts_lowthrust", "LGL7", False, 1031), ("betts_lowthrust", "LGL7", False, 4500),
    ("betts_lowthrust", "LGL7", False, 9001),
    # (round 6) shapes with both forms of the dense part (registry.h: alt, lpair): tiles below two and a half segments per workgroup, rows from there to the
    # end of the one-group kernel, the looped pair kernel (rows) on every looped mesh
    ("reentry", "LGL7", False, 2559), ("reentry", "LGL7", False, 2560), ("reentry", "LGL7", False, 6143), ("reentry", "LGL7", False, 6144),
    ("reentry", "LGL7", False, 10240), ("reentry", "LGL7", False, 10241), ("reentry", "LGL7", False, 15000),
    ("reentry", "LGL7", False, 30720), ("reentry", "LGL7", False, 30721), ("reentry", "LGL3", False, 6200), ("reentry", "LGL5", False, 43100),
    # light right-hand sides keep single-wave workgroups on looped meshes (ResDims::LOOP_PAIR)
    ("twobody_lt", "LGL5", True, 60003), ("brachistochrone", "LGL7", False, 40001)])
def test_launch_form_boundaries(oracle, ode, mode, blocked, nseg):
    """Mesh sizes on either side of every switch of the launcher (csrc/registry.h: launch_lgl_table): one-wave fused,
    two-wave fused, ODE stage + dense stage, units -- every block against the oracle."""
    w = Workload(ode, mode, nseg, blocked)
    ev = DefectEvaluator(ode, mode, blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    ref = w.oracle_nlp(oracle, threads=8).eval_blocks(oracle.JAC_ADJGRAD_HESS, w.X, w.L)
    _check_blocks(ev.eval(JAC_ADJGRAD_HESS, w.X, w.L), ref, w, JAC_ADJGRAD_HESS)
    ev.close()


@pytest.mark.parametrize("ode,mode,blocked,nseg", [("reentry", "LGL7", False, 1), ("reentry", "LGL7", False, 17),
                                                   ("reentry", "LGL7", False, 4099), ("twobody_lt", "LGL5", True, 43),
                                                   ("betts_lowthrust", "LGL5", True, 22), ("betts_lowthrust", "Trapezoidal", False, 33),
                                                   ("synthetic32", "LGL7", False, 20), ("brachistochrone", "LGL3", False, 65)])
def test_value_and_adjoint_gradient_kernel_on_ragged_groups(oracle, ode, mode, blocked, nseg):
    """evalOCC / evalRHS go through the vector-Jacobian kernel (csrc/defect_adjgrad.h: no Jacobian is formed): group sizes
    that leave a ragged last workgroup, controls as parameters, a phase parameter, the Trapezoidal form, a wide ODE."""
    w = Workload(ode, mode, nseg, blocked, var_offset=2, con_offset=1, extra_vars=3)
    nlp = w.oracle_nlp(oracle, threads=4)
    ev = DefectEvaluator(ode, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    for what in (CON, CON_ADJGRAD):
        _check_blocks(ev.eval(what, w.X, w.L if what == CON_ADJGRAD else None), nlp.eval_blocks(what, w.X, w.L), w, what)
    ev.close()


@pytest.mark.parametrize("mode,nseg", [("LGL7", 1), ("LGL7", 2), ("LGL7", 3), ("LGL7", 257), ("LGL5", 1), ("LGL5", 511), ("LGL3", 513)])
def test_ragged_segment_counts_wide_shapes(oracle, mode, nseg):
    """The four-wave dense kernel (csrc/defect_wide.h): fewer segments than workgroups, one segment, counts that leave
    the workgroups with unequal shares."""
    w = Workload("synthetic32", mode, nseg)
    ev = DefectEvaluator("synthetic32", mode, False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    _check_blocks(ev.eval(JAC_ADJGRAD_HESS, w.X, w.L), w.oracle_nlp(oracle, threads=8).eval_blocks(oracle.JAC_ADJGRAD_HESS, w.X, w.L),
                  w, JAC_ADJGRAD_HESS)
    ev.close()


GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if os.path.basename(p) != "pathfuncs.npz")   # (the defect vectors; pathfuncs.npz: test_pathfuncs_oracle.py)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_hip_matches_golden(path):
    g = np.load(path)
    ode, mode, blocked = str(g["ode"]), str(g["mode"]), bool(g["blocked"])
    if not _lib.has_kernel(ode, _lib.MODES[mode], blocked):
        pytest.skip("no device kernel for this size yet")
    ns, IR = g["x"].shape
    OR = g["lam"].shape[1]
    X, L = g["x"].ravel(), g["lam"].ravel()
    V = np.arange(ns * IR, dtype=np.int32).reshape(ns, IR)
    Cx = np.arange(ns * OR, dtype=np.int32).reshape(ns, OR)
    ev = DefectEvaluator(ode, mode, blocked, V, Cx, X.size, L.size)
    fx, agx, kkt = ev.eval(JAC_ADJGRAD_HESS, X, L)
    for s in range(ns):
        H, J = unpack_kkt_block(kkt[s], IR, OR)
        assert np.abs(fx[s] - g["fx"][s]).max() / max(1.0, np.abs(g["x"][s]).max()) < TOL_RES
        assert rel_err(J, g["jx"][s]) < TOL_DER
        assert rel_err(agx[s], g["gx"][s]) < TOL_DER
        assert rel_err(H, g["hx"][s]) < TOL_DER


def test_gpu_full_size_properties(oracle):
    """BASELINE sizes: 10k-segment LGL7 phase.  Oracle on a strided sample + size-independent properties:
    J^T lam == adjoint gradient for every segment, shifting all node times leaves the autonomous reentry
    defects unchanged, and the evaluation is idempotent (bitwise repeatable)."""
    w = Workload("reentry", "LGL7", 10000)
    ev = DefectEvaluator("reentry", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    fx, agx, kkt = ev.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    fx2, agx2, kkt2 = ev.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    assert np.array_equal(fx, fx2) and np.array_equal(agx, agx2) and np.array_equal(kkt, kkt2)
    o = oracle.get_ode("reentry", 0)
    for V in range(0, 10000, 397):
        rfx, rjx, rgx, rhx = oracle.defect_all(o, oracle.LGL7, w.X[w.vindex[V]], w.L[w.cindex[V]])
        H, J = unpack_kkt_block(kkt[V], w.IR, w.OR)
        assert np.abs(fx[V] - rfx).max() / max(1.0, np.abs(w.X).max()) < TOL_RES
        assert rel_err(J, rjx) < TOL_DER and rel_err(H, rhx) < TOL_DER and rel_err(agx[V], rgx) < TOL_DER
    # J^T lam identity on every segment (reference recipe, 1e-12 -> here relative 1e-10 over 10k segments)
    lam = w.L[w.cindex]                                    # [nseg, OR]
    k = 0
    JT = np.zeros((w.nseg, w.IR, w.OR))
    for i in range(w.IR):
        k += w.IR - i
        JT[:, i, :] = kkt[:, k:k + w.OR]
        k += w.OR
    g2 = np.einsum("sio,so->si", JT, lam)
    assert rel_err(agx, g2) < 1e-12
    # time-shift invariance (reentry has no explicit time dependence)
    X2 = w.X.copy()
    tidx = w.vindex[:, [w.indexer.xv + j * w.indexer.XtUVars() for j in range(w.cs)]]
    X2[np.unique(tidx)] += 0.25
    fx3, _, _ = ev.eval(CON, X2)
    assert np.abs(fx3 - fx).max() < 1e-9


@pytest.mark.parametrize("ode,mode,nseg,blocked", [
    ("betts_lowthrust", "LGL5", 1000, False),        # BASELINE configs[1]
    ("reentry", "LGL7", 5000, False),                # BASELINE configs[2] (initial mesh)
])
def test_baseline_configs_at_full_size_match_the_oracle_everywhere(oracle, ode, mode, nseg, blocked):
    """Every block of every segment against the oracle's NLP restatement at the BASELINE sizes."""
    w = Workload(ode, mode, nseg, blocked)
    ev = DefectEvaluator(ode, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    nlp = w.oracle_nlp(oracle, threads=8)
    for what in (JAC_ADJGRAD_HESS, CON):
        _check_blocks(ev.eval(what, w.X, w.L if what == JAC_ADJGRAD_HESS else None), nlp.eval_blocks(what, w.X, w.L), w, what)
    ev.close()


def test_baseline_multi_phase_config_eight_linked_phases(oracle):
    """BASELINE configs[3]: eight TwoBody LGL5 phases with BlockConstant control in one solver vector (their variables
    and constraint rows one after the other, as OptimalControlProblem lays linked phases out): one evaluator per phase,
    each against the oracle on its own slice."""
    from asset_asrl_amd.ocp import OptimalControlProblem
    from asset_asrl_amd.ode import TwoBody
    nseg, phases = 400, 8
    ocp, ws = OptimalControlProblem(), []
    for k in range(phases):                                # placed by the OptimalControlProblem mirror (ocp.py)
        wk = Workload("twobody_lt", "LGL5", nseg, True, seed=100 + k)
        ph = TwoBody().phase("LGL5", wk.traj, nseg)
        ph.setControlMode("BlockConstant")
        ph.EnableMeshSpacing = False
        ocp.addPhase(ph)
        ws.append(wk)
    n_primal, n_equal = ocp.n_primal, ocp.n_equal
    X = ocp.solver_input()
    L = np.concatenate([wk.L for wk in ws])
    for wk, (V, Cx) in zip(ws, ocp.defect_tables()):
        wk.vindex, wk.cindex = V, Cx
    for w in ws:
        ev = DefectEvaluator("twobody_lt", "LGL5", True, w.vindex, w.cindex, n_primal, n_equal)
        nlp = oracle.Nlp(oracle.get_ode("twobody_lt", 0), oracle.MODES["LGL5"], True, w.vindex, w.cindex, n_primal, n_equal, 4)
        got = ev.eval(JAC_ADJGRAD_HESS, X, L)
        ref = nlp.eval_blocks(JAC_ADJGRAD_HESS, X, L)
        fx, agx, kkt = got
        assert np.abs(fx - ref[0]).max() / max(1.0, np.abs(X).max()) < TOL_RES
        assert rel_err(agx, ref[1]) < TOL_DER and rel_err(kkt, ref[2]) < TOL_DER
        ev.close()


def test_gpu_full_size_wide_shape_properties(oracle):
    """BASELINE config 4 at full size: 100 000 segments of the 32-state ODE in LGL7 (17.5 GB of blocks, kept on the
    device).  Oracle on a strided sample, J^T lam == adjoint gradient for every segment, symmetric use of the
    Hessian slots is implied by the sample; the evaluation is bitwise repeatable."""
    import torch
    nseg = 100000
    w = Workload("synthetic32", "LGL7", nseg)
    ev = DefectEvaluator("synthetic32", "LGL7", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    dev = torch.device("cuda:0")
    X, L = torch.from_numpy(w.X).to(dev), torch.from_numpy(w.L).to(dev)
    fx = torch.zeros(nseg, w.OR, dtype=torch.float64, device=dev)
    agx = torch.zeros(nseg, w.IR, dtype=torch.float64, device=dev)
    kkt = torch.zeros(nseg, ev.KSTRIDE, dtype=torch.float64, device=dev)
    assert ev.kkt_layout == 0 and ev.KSTRIDE == w.NKKT            # (a wide shape: the reference's slot order, which the J^T lam check below walks)
    torch.cuda.synchronize()                              # (torch fills on its stream, the evaluator runs on its own)
    ev.eval_device(JAC_ADJGRAD_HESS, X, L, fx, agx, kkt)
    torch.cuda.synchronize()
    ref = (fx.clone(), agx.clone(), kkt[::1000].clone())
    torch.cuda.synchronize()
    ev.eval_device(JAC_ADJGRAD_HESS, X, L, fx, agx, kkt)
    torch.cuda.synchronize()
    assert torch.equal(fx, ref[0]) and torch.equal(agx, ref[1]) and torch.equal(kkt[::1000], ref[2])
    # oracle on a strided sample
    o = oracle.get_ode("synthetic32", 0)
    for V in range(0, nseg, 9973):
        rfx, rjx, rgx, rhx = oracle.defect_all(o, oracle.LGL7, w.X[w.vindex[V]], w.L[w.cindex[V]])
        H, J = unpack_kkt_block(kkt[V].cpu().numpy(), w.IR, w.OR)
        assert np.abs(fx[V].cpu().numpy() - rfx).max() / max(1.0, np.abs(w.X).max()) < TOL_RES
        assert rel_err(J, rjx) < TOL_DER and rel_err(H, rhx) < TOL_DER and rel_err(agx[V].cpu().numpy(), rgx) < TOL_DER
    # J^T lam identity on every segment, in chunks on the device
    jslot = np.zeros((w.IR, w.OR), dtype=np.int64)        # slot of J(o, i) in a block: column i's run, after its H entries
    k = 0
    for i in range(w.IR):
        k += w.IR - i
        jslot[i] = np.arange(k, k + w.OR)
        k += w.OR
    jslot_d = torch.from_numpy(jslot.ravel()).to(dev)
    lam = L[torch.from_numpy(w.cindex.astype(np.int64)).to(dev)]      # [nseg, OR]
    worst = 0.0
    for s0 in range(0, nseg, 10000):
        JT = kkt[s0:s0 + 10000].index_select(1, jslot_d).view(-1, w.IR, w.OR)
        g2 = torch.einsum("sio,so->si", JT, lam[s0:s0 + 10000])
        worst = max(worst, float((g2 - agx[s0:s0 + 10000]).abs().max() / agx[s0:s0 + 10000].abs().max()))
    assert worst < 1e-12
    ev.close()


@pytest.mark.parametrize("ode,mode,blocked", [("reentry", "LGL7", False), ("twobody_lt", "LGL5", True), ("betts_lowthrust", "Trapezoidal", False)])
def test_renumbered_variables_give_the_same_blocks(ode, mode, blocked):
    """The kernels read the solver vectors only through Vindex / Cindex (ComputableBase.h:351-378): renumbering the
    variables and multipliers by a random permutation -- tables that are no longer those of a phase -- must not change
    a single bit of the blocks."""
    w = Workload(ode, mode, 77, blocked, var_offset=3, con_offset=2, extra_vars=4)
    ev = DefectEvaluator(ode, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    ref = [a.copy() for a in ev.eval(JAC_ADJGRAD_HESS, w.X, w.L)]
    ref0 = ev.eval(CON, w.X)[0].copy()
    ev.close()
    rng = np.random.default_rng(3)
    pv, pc = rng.permutation(w.n_primal), rng.permutation(w.n_equal)     # new position of old variable i: pv[i]
    X2, L2 = np.empty_like(w.X), np.empty_like(w.L)
    X2[pv], L2[pc] = w.X, w.L
    ev2 = DefectEvaluator(ode, mode, w.blocked, pv[w.vindex].astype(np.int32), pc[w.cindex].astype(np.int32), w.n_primal, w.n_equal)
    got = ev2.eval(JAC_ADJGRAD_HESS, X2, L2)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ev2.eval(CON, X2)[0], ref0)
    ev2.close()


def test_pinned_outputs_give_the_same_blocks(oracle):
    w = Workload("reentry", "LGL5", 300)
    ev = DefectEvaluator("reentry", "LGL5", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    ref = [None if a is None else a.copy() for a in ev.eval(JAC_ADJGRAD_HESS, w.X, w.L)]
    ev.pin_outputs()
    got = ev.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    assert got[2].shape == (w.nseg, ev.KSTRIDE)         # pinned outputs: the blocks as the C ABI returns them, in the handle's layout
    for a, b in zip((got[0], got[1], ev.kkt_to_reference(got[2])), ref):
        np.testing.assert_array_equal(a, b)
    assert ev.eval(CON, w.X)[0] is got[0]               # the same page-locked arrays are returned every time
    ev.close()
