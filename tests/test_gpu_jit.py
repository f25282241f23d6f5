"""A user-defined ODE compiled at run time, checked on the GPU against the oracle's independent AD2 derivatives of
the same right-hand side (oracle/odes.h ``vanderpol``): every transcription, both control modes, all five
evaluation kinds, and through the Phase API."""
import numpy as np
import pytest

from asset_asrl_amd import jit
from asset_asrl_amd.evaluator import CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, DefectEvaluator
from helpers import Workload, make_coupled, make_coupled12, make_driven, make_vanderpol, rel_err
from test_gpu_parity import _check_blocks

pytestmark = pytest.mark.gpu
SIZES = (2, 1, 1)


@pytest.mark.parametrize("mode", ["Trapezoidal", "LGL3", "LGL5", "LGL7"])
@pytest.mark.parametrize("blocked", [False, True])
def test_jit_ode_matches_oracle(oracle, mode, blocked):
    name = jit.ensure_kernel(make_vanderpol(), mode, blocked)
    w = Workload("vanderpol", mode, 37, blocked, sizes=SIZES, var_offset=2, con_offset=1, extra_vars=3)
    nlp = oracle.Nlp(oracle.get_ode("vanderpol", 0), oracle.MODES[mode], w.blocked, w.vindex, w.cindex, w.n_primal,
                     w.n_equal, 2)
    ev = DefectEvaluator(name, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    for what in (JAC_ADJGRAD_HESS, CON, CON_ADJGRAD, JAC, JAC_ADJGRAD):
        ref = nlp.eval_blocks(what, w.X, w.L)
        got = ev.eval(what, w.X, w.L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
        _check_blocks(got, ref, w, what)
    ev.close()


@pytest.mark.parametrize("n,mode,blocked", [(12, "LGL7", False), (12, "LGL7", True), (12, "LGL5", False),
                                            (16, "LGL7", True), (16, "LGL5", False)])
def test_wide_user_ode_with_controls_and_parameters(oracle, n, mode, blocked):
    """(12, 3, 2) in LGL7: IR = 66 -> the four-wave dense kernel with control-interpolation rows and parameter columns
    (csrc/defect_wide.h); its BlockConstant and LGL5 forms take the single-wave kernel.  (16, 3, 2): the BlockConstant
    LGL7 form (IR = 73, five parameter columns) and the LGL5 form (IR = 62, narrow) of a second size."""
    from asset_asrl_amd.evaluator import JAC_ADJGRAD_HESS as KIND
    name = jit.ensure_kernel(make_coupled(n), mode, blocked)
    w = Workload(f"coupled{n}", mode, 29, blocked, sizes=(n, 3, 2), var_offset=2, con_offset=1, extra_vars=3)
    nlp = oracle.Nlp(oracle.get_ode(f"coupled{n}", 0), oracle.MODES[mode], w.blocked, w.vindex, w.cindex, w.n_primal,
                     w.n_equal, 2)
    ev = DefectEvaluator(name, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    assert ev.IR == (4 if mode == "LGL7" else 3) * (n + 1 if blocked else n + 4) + (5 if blocked else 2)
    for what in (KIND, CON, CON_ADJGRAD, JAC, JAC_ADJGRAD):
        ref = nlp.eval_blocks(what, w.X, w.L)
        got = ev.eval(what, w.X, w.L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
        _check_blocks(got, ref, w, what)
    # on-device assembly through the same kernels (shared boundary nodes and the two phase parameters)
    locs = nlp.kkt_locations()[:nlp.num_user_kkt].reshape(w.nseg, ev.NKKT)
    ev.set_kkt_map(locs, nlp.nnz)
    vals = np.zeros(nlp.nnz)
    ev.eval_assembled(KIND, w.X, w.L, vals)
    assert rel_err(vals, nlp.eval(KIND, w.X, w.L)[2]) < 1e-8
    ev.close()


@pytest.mark.parametrize("n,mode", [(14, "LGL7"), (20, "LGL5"), (20, "LGL7"), (14, "LGL5"), (14, "LGL3")])
def test_wide_user_ode_with_controls_and_no_parameters(oracle, n, mode):
    """(14, 3, 0) in LGL7 (IR = 72), (20, 3, 0) in LGL5 (IR = 72) and LGL7 (IR = 96): wide shapes without segment parameters
    -- the row-wise dense stage (csrc/defect_rows.h) with control-interpolation rows, which the 32-state BASELINE ODE does not
    have; every evaluation kind, and the assembled kind through the tile kernel.  (14, 3, 0) in LGL5 / LGL3: the narrow
    neighbours of the same right-hand side."""
    name = jit.ensure_kernel(make_driven(n), mode, False)
    w = Workload(f"driven{n}", mode, 31, False, sizes=(n, 3, 0), var_offset=2, con_offset=1, extra_vars=3)
    nlp = oracle.Nlp(oracle.get_ode(f"driven{n}", 0), oracle.MODES[mode], False, w.vindex, w.cindex, w.n_primal, w.n_equal, 2)
    ev = DefectEvaluator(name, mode, False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    assert ev.IR == {"LGL3": 2, "LGL5": 3, "LGL7": 4}[mode] * (n + 4)
    for what in (JAC_ADJGRAD_HESS, CON, CON_ADJGRAD, JAC, JAC_ADJGRAD):
        ref = nlp.eval_blocks(what, w.X, w.L)
        got = ev.eval(what, w.X, w.L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
        _check_blocks(got, ref, w, what)
    locs = nlp.kkt_locations()[:nlp.num_user_kkt].reshape(w.nseg, ev.NKKT)
    ev.set_kkt_map(locs, nlp.nnz)
    vals = np.zeros(nlp.nnz)
    ev.eval_assembled(JAC_ADJGRAD_HESS, w.X, w.L, vals)
    assert rel_err(vals, nlp.eval(JAC_ADJGRAD_HESS, w.X, w.L)[2]) < 1e-8
    ev.close()


def test_phase_api_with_user_ode(oracle):
    ode = make_vanderpol()
    w = Workload("vanderpol", "LGL5", 12, sizes=SIZES)
    ph = ode.phase("LGL5", w.traj, 12)
    d = ph.get_defect()
    assert (d.IRows(), d.ORows()) == (3 * 4 + 1, 2 * 2)
    rng = np.random.default_rng(5)
    nodes = [np.concatenate([rng.uniform(-1, 1, 2), [t], rng.uniform(-1, 1, 1)]) for t in (0.0, 0.4, 1.0)]
    x = np.concatenate(nodes + [[0.9]])           # z = [x, t, u] at the three cardinal nodes, then the parameter
    lam = rng.uniform(-1, 1, 4)
    fx, jx, gx, hx = d.computeall(x, lam)
    rfx, rjx, rgx, rhx = oracle.defect_all(oracle.get_ode("vanderpol", 0), oracle.MODES["LGL5"], x, lam)
    assert np.abs(fx - rfx).max() < 1e-10
    assert rel_err(jx, rjx) < 1e-8 and rel_err(gx, rgx) < 1e-8 and rel_err(hx, rhx) < 1e-8
    res = ph.test_threads(1, 1, 2, verbose=False)
    assert res["segments"] == 12


def test_autoscaling_phase_evaluates_the_scaled_dynamics(oracle):
    """setUnits / setAutoScaling (ODEPhase.h:87-109, 293-326): the defect of IOScaled(ode, units, t-unit / x-units) at
    z / units equals the unscaled defect divided by the state units, its Jacobian is diag(1/ux) J diag(uz)."""
    from asset_asrl_amd.ode import ShuttleReentry
    w = Workload("reentry", "LGL5", 9)
    units = np.array([2.0, 0.5, 3.0, 1.5, 0.8, 4.0, 1.25, 2.5])
    ph = ShuttleReentry().phase("LGL5", w.traj, 9)
    ph.setUnits(units)
    ph.setAutoScaling(True)
    d = ph.get_defect()
    rng = np.random.default_rng(3)
    zi = w.X[w.vindex[4]]                                 # one segment's variables, unscaled
    lam = rng.uniform(-1, 1, d.ORows())
    uz = np.tile(units, 3)                                # [x,t,u] units at the three cardinal nodes (no parameters)
    ux = np.tile(units[:5], 2)                            # defect rows: two interior points x five states
    fx, jx, gx, hx = d.computeall(zi / uz, lam)
    rfx, rjx, rgx, rhx = oracle.defect_all(oracle.get_ode("reentry", 0), oracle.MODES["LGL5"], zi, lam / ux)
    assert np.abs(fx - rfx / ux).max() < 1e-10
    assert rel_err(jx, rjx * uz[None, :] / ux[:, None]) < 1e-8
    assert rel_err(gx, rgx * uz) < 1e-8
    assert rel_err(hx, rhx * uz[None, :] * uz[:, None]) < 1e-8
    assert np.allclose(ph.solver_input()[:8], w.traj[0] / units)       # the NLP variables are in scaled units


def test_cold_run_time_compilation_of_a_never_seen_ode(oracle):
    """The run-time compiler itself, cold, on the GPU box: an ODE whose generated code no cache can hold -- the forced
    Van der Pol oscillator with exp(-t/10) written as exp(-(1/10 + c) t) * exp(c t), c drawn from os.urandom for this test
    run (the same real function to rounding, a different expression graph, hence a different content hash) -- goes through
    code generation, in-process compilation (hiprtc), module load and registration here, and its blocks match the oracle's Van der Pol."""
    import os
    import shutil
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase
    c = 1e-3 * (1 + int.from_bytes(os.urandom(4), "little") / 2 ** 32)
    a = ODEArguments(2, 1, 1)
    x0, x1 = a.XVec().tolist()
    u, mu, t = a.UVar(0), a.PVar(0), a.TVar()
    ode = ODEBase(vf.stack([x1, mu * (1.0 - x0 * x0) * x1 - x0 + u * (vf.exp(-(0.1 + c) * t) * vf.exp(c * t))]), 2, 1, 1,
                  name="vanderpol_salted")
    name = jit.device_name(ode)
    wd = os.path.join(jit.JIT_DIR, name)
    assert not os.path.exists(wd), "the salted ODE must not be in the plugin cache"
    try:
        os.environ.pop("ASSET_HIP_JIT", None)                            # the default route: hiprtc, inside this process
        assert jit.ensure_kernel(ode, "LGL5", False) == name             # compiles here, now
        made = os.listdir(wd)
        assert any(f.startswith("module_lgl5_0_") and f.endswith(".rtc") for f in made)
        assert not any(f.endswith(".so") or f.endswith(".hip") for f in made)   # no compiler driver, no host code
        w = Workload("vanderpol", "LGL5", 21, False, sizes=SIZES)
        nlp = oracle.Nlp(oracle.get_ode("vanderpol", 0), oracle.MODES["LGL5"], False, w.vindex, w.cindex, w.n_primal, w.n_equal, 2)
        ev = DefectEvaluator(name, "LGL5", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
        _check_blocks(ev.eval(JAC_ADJGRAD_HESS, w.X, w.L), nlp.eval_blocks(JAC_ADJGRAD_HESS, w.X, w.L), w, JAC_ADJGRAD_HESS)
        ev.close()
    finally:
        shutil.rmtree(wd, ignore_errors=True)                             # (the module stays loaded; its files need not stay)


def test_run_time_module_follows_the_handle_to_another_device(oracle):
    """A module compiled at run time is loaded on every device it is used on (registry.h: RtcModule): a handle created with
    device = 1 while device 0 is current launches the module's kernels on device 1.  Needs two GPUs."""
    from asset_asrl_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("one GPU visible")
    ode = make_vanderpol()
    ph = ode.phase("LGL5", Workload("vanderpol", "LGL5", 9, sizes=(2, 1, 1)).traj, 9)
    ph.device = 1
    ev = ph.evaluator
    w = Workload("vanderpol", "LGL5", 9, sizes=(2, 1, 1))
    X, L = ph.solver_input(), np.linspace(-1.0, 1.0, ev.n_equal)
    got = ev.eval(JAC_ADJGRAD_HESS, X, L)
    ph.device = 0
    ph._ev = None
    same = ph.evaluator.eval(JAC_ADJGRAD_HESS, X, L)
    for a, b in zip(got, same):
        np.testing.assert_array_equal(a, b)


def test_cached_code_object_loads_as_a_second_module_and_both_launch_from_two_threads(oracle):
    """What the two-GPU test above checks needs a second device; what it rests on does not: a run-time module is an object of its
    own (registry.h: RtcModule -- code object, lowered names, one lazily loaded hipModule_t per device), several of them live in
    one process, and their first launches may come from different threads at once (the per-device table is
    sized once and the function handle copied out under the lock).  Here the cached code object of the Van der Pol ODE is
    registered a second time under another name -- a distinct RtcModule, loaded from the cache file without a compiler -- and
    the two are evaluated concurrently from two threads: same bits, and the oracle's blocks."""
    import ctypes as C
    import glob
    import os
    import threading
    from asset_asrl_amd import _lib
    ode = make_vanderpol()
    name = jit.ensure_kernel(ode, "LGL5", False)
    caches = sorted(glob.glob(os.path.join(jit.JIT_DIR, name, "module_lgl5_0_*.rtc")), key=os.path.getmtime)
    assert caches, "the module cache of the Van der Pol ODE"
    alias = (name + "_again").encode()
    rc = _lib.lib().asset_hip_jit_plugin(alias, None, b"unused", 1, _lib.MODES["LGL5"], 0, 0, None, 0, caches[-1].encode())
    assert rc == 0, _lib.lib().asset_hip_last_error()
    assert _lib.has_kernel(alias.decode(), _lib.MODES["LGL5"], False)
    w = Workload("vanderpol", "LGL5", 2049, False, sizes=SIZES)
    ref = oracle.Nlp(oracle.get_ode("vanderpol", 0), oracle.MODES["LGL5"], False, w.vindex, w.cindex, w.n_primal, w.n_equal,
                     4).eval_blocks(JAC_ADJGRAD_HESS, w.X, w.L)
    out, errs = {}, []

    def run(nm):                       # (handle creation and the first launch of a module happen inside the thread)
        try:
            ev = DefectEvaluator(nm, "LGL5", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
            res = None
            for _ in range(5):
                res = [a.copy() for a in ev.eval(JAC_ADJGRAD_HESS, w.X, w.L)]
            out[nm] = res
            ev.close()
        except Exception as exc:       # noqa: BLE001
            errs.append(exc)
    ts = [threading.Thread(target=run, args=(nm,)) for nm in (name, alias.decode())]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for a, b in zip(out[name], out[alias.decode()]):
        np.testing.assert_array_equal(a, b)
    _check_blocks(out[name], ref, w, JAC_ADJGRAD_HESS)


@pytest.mark.parametrize("mode,blocked", [("LGL5", False), ("LGL7", True)])
def test_composed_user_ode_block_form_against_flat_form(mode, blocked):
    """A user ODE written as a composition ``outer(inner(y))`` (helpers.make_nested_orbit): its derivatives are formed by the
    block-wise chain rule across the composition's cuts (vf/codegen.py; the reference: NestedFunction.h:140-270) -- another body
    of device code than the flattened expression gives, in fewer operations.  Both are compiled and run through the same kernels:
    values, adjoint gradient and KKT blocks agree to rounding (the flattened form is what every other test holds to the oracle)."""
    from helpers import make_nested_orbit
    blk, flat = make_nested_orbit(), make_nested_orbit(flat=True)
    assert blk.derivatives().chain_rule["form"].startswith("block") and flat.derivatives().chain_rule["form"] == "flat"
    assert blk.derivatives().stats()["ops_fjgh"] < flat.derivatives().stats()["ops_fjgh"]
    nb, nf = jit.ensure_kernel(blk, mode, blocked), jit.ensure_kernel(flat, mode, blocked)
    assert nb != nf                                                     # two modules (the name hashes the generated body)
    w = Workload("nested_orbit", mode, 61, blocked, sizes=(4, 1, 1))
    stride, S = (5 if w.blocked else 6), w.indexer.numStates
    rng = np.random.default_rng(11)
    w.X[0:S * stride:stride] = rng.uniform(0.6, 1.4, S)                # rho > 0
    w.X[2:S * stride:stride] = rng.uniform(0.2, 0.9, S)                # z away from 0: |R| > 0
    eb = DefectEvaluator(nb, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    ef = DefectEvaluator(nf, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    for what in (JAC_ADJGRAD_HESS, JAC_ADJGRAD, CON):
        L = w.L if what != CON else None
        a, b = eb.eval(what, w.X, L), ef.eval(what, w.X, L)
        for x, y in zip(a, b):
            assert (x is None) == (y is None)
            if x is not None:
                assert np.all(np.isfinite(y)) and rel_err(x, y) < 1e-12
    # J^T lam == adjoint gradient on the block-form module (the reference's own consistency recipe)
    fx, agx, kkt = eb.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    from asset_asrl_amd.evaluator import unpack_kkt_block
    for V in (0, 17, 60):
        H, J = unpack_kkt_block(kkt[V], eb.IR, eb.OR)
        assert rel_err(J.T @ w.L[w.cindex[V]], agx[V]) < 1e-12 and rel_err(H, H.T) == 0.0
    eb.close()
    ef.close()


@pytest.mark.parametrize("mode,blocked", [("LGL5", False), ("LGL7", True), ("Trapezoidal", False)])
def test_user_ode_with_conditionals_matches_oracle(oracle, mode, blocked):
    """``vf.ifelse`` / ``vf.abs`` / ``vf.sign`` on the device (round 6; the reference's IfElseFunction, ConditionalStatement,
    SignFunction: CommonFunctions/Conditional.h:19-260): a switched oscillator whose lanes take different branches, run-time
    compiled, against the oracle's AD2 derivatives of the same right-hand side written with C++ branches (oracle/odes.h: switched)."""
    from helpers import make_switched
    name = jit.ensure_kernel(make_switched(), mode, blocked)
    w = Workload("switched", mode, 83, blocked, sizes=(2, 1, 0), var_offset=1, con_offset=2, extra_vars=2)
    nlp = oracle.Nlp(oracle.get_ode("switched", 0), oracle.MODES[mode], w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal, 2)
    ev = DefectEvaluator(name, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    for what in (JAC_ADJGRAD_HESS, CON, CON_ADJGRAD, JAC, JAC_ADJGRAD):
        ref = nlp.eval_blocks(what, w.X, w.L)
        got = ev.eval(what, w.X, w.L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
        _check_blocks(got, ref, w, what)
    ev.close()


@pytest.mark.parametrize("mode,blocked", [("LGL5", False), ("LGL7", True), ("Trapezoidal", False), ("LGL3", False)])
def test_user_ode_on_tabulated_data_matches_oracle(oracle, mode, blocked):
    """``vf.InterpTable1D`` on the device (round 6; the reference's InterpTable1D / InterpFunction1D, CommonFunctions/InterpTable1D.h):
    a cubic table with uneven abscissae over a state (bisection per lane), a linear one over the time, a two-valued cubic one over the
    other state -- constant arrays of the run-time compiled module -- against the oracle's AD2 derivatives of the same right-hand side
    over its restatement of the reference's table formulas (oracle/odes.h: tabulated, oracle/interp_table.h).  The synthetic states
    reach past both ends of the tables: the clamped element extrapolates, as in the reference."""
    from helpers import make_tabulated
    name = jit.ensure_kernel(make_tabulated(), mode, blocked)
    w = Workload("tabulated", mode, 83, blocked, sizes=(2, 1, 0), var_offset=1, con_offset=2, extra_vars=2)
    ix = w.indexer
    S, xtu = ix.numStates, (ix.XtVars() if w.blocked else ix.XtUVars())    # (BlockConstant: the controls follow the states as a block)
    st = w.X[1:1 + S * xtu].reshape(S, xtu)
    st[:, 0] = np.random.default_rng(4).uniform(-1.8, 2.6, S)              # altitude: across (and beyond) the uneven table
    st[:, 1] *= 2.2                                                        # speed: beyond the two-valued table at both ends
    nlp = oracle.Nlp(oracle.get_ode("tabulated", 0), oracle.MODES[mode], w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal, 2)
    ev = DefectEvaluator(name, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    for what in (JAC_ADJGRAD_HESS, CON, CON_ADJGRAD, JAC, JAC_ADJGRAD):
        ref = nlp.eval_blocks(what, w.X, w.L)
        got = ev.eval(what, w.X, w.L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
        _check_blocks(got, ref, w, what)
    ev.close()
