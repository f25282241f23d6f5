"""On-device KKT assembly: blocks scattered into the solver's CSR value array on the GPU,
checked against the oracle's restatement of NonLinearProgram::evalKKT / evalSOE (oracle/nlp.cpp), which performs
the reference's indexed += on the host.  The sparsity analysis (CSR structure, KKTLocations) comes from the oracle,
standing in for the host solver that owns it."""
import numpy as np
import pytest
import torch

from asset_asrl_amd import _lib
from asset_asrl_amd.evaluator import JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, DefectEvaluator
from helpers import Workload, rel_err

pytestmark = pytest.mark.gpu

CASES = [
    ("reentry", "LGL7", 257, False),            # shared boundary nodes between adjacent segments
    ("twobody_lt", "LGL5", 75, True),           # BlockConstant: per-segment control parameters
    ("betts_lowthrust", "LGL5", 40, False),     # a phase parameter shared by every segment (many-way clash)
    ("brachistochrone", "Trapezoidal", 33, False),
    ("synthetic32", "LGL7", 5, False),          # wide shape: map entries read at the stores
]


def _setup(oracle, ode, mode, nseg, blocked, accumulate=False):
    w = Workload(ode, mode, nseg, blocked, var_offset=3, con_offset=2, extra_vars=4)
    nlp = w.oracle_nlp(oracle, threads=2)
    ev = DefectEvaluator(ode, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    locs = nlp.kkt_locations()[:nlp.num_user_kkt].reshape(w.nseg, ev.NKKT)
    ev.set_kkt_map(locs, nlp.nnz, accumulate)
    return w, nlp, ev, locs


@pytest.mark.parametrize("ode,mode,nseg,blocked", CASES)
def test_assembled_values_match_host_scatter(oracle, ode, mode, nseg, blocked):
    w, nlp, ev, locs = _setup(oracle, ode, mode, nseg, blocked)
    assert np.unique(locs).size < locs.size                    # the case does exercise shared locations
    for what in (JAC_ADJGRAD_HESS, JAC_ADJGRAD, JAC):
        _, _, ref = nlp.eval(what, w.X, w.L)
        base = np.random.default_rng(what).uniform(-1, 1, nlp.nnz)   # values other functions already added
        vals = base.copy()
        fx, agx = ev.eval_assembled(what, w.X, w.L if what != JAC else None, vals)
        assert rel_err(vals - base, ref) < 1e-8
        touched = np.zeros(nlp.nnz, bool)
        touched[locs.ravel()] = True
        np.testing.assert_array_equal(vals[~touched], base[~touched])     # nothing outside the constraint's slots moves
        rfx, ragx, _ = nlp.eval_blocks(what, w.X, w.L)
        assert np.abs(fx - rfx).max() < 1e-10 * max(1.0, np.abs(w.X).max())
        if agx is not None:
            assert rel_err(agx, ragx) < 1e-8
    # first function into a freshly zeroed array: its range is overwritten by one copy, nothing else is touched
    _, _, ref = nlp.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    vals = np.zeros(nlp.nnz)
    vals[:locs.min()] = 7.0                                       # (outside the constraint's range)
    ev.eval_assembled(JAC_ADJGRAD_HESS, w.X, w.L, vals, target_zeroed=True)
    assert rel_err(vals[locs.min():], ref[locs.min():]) < 1e-8 and np.all(vals[:locs.min()] == 7.0)
    ev.close()


@pytest.mark.parametrize("mode", ["LGL7", "Trapezoidal"])
def test_assembled_device_pointers(oracle, mode):
    """Default map: the device array holds zeros at the constraint's locations and receives its contributions."""
    w, nlp, ev, _ = _setup(oracle, "reentry", mode, 300, False)
    dev = torch.device("cuda:0")
    X, L = torch.from_numpy(w.X).to(dev), torch.from_numpy(w.L).to(dev)
    fx = torch.zeros(w.nseg * ev.OR, dtype=torch.float64, device=dev)
    agx = torch.zeros(w.nseg * ev.IR, dtype=torch.float64, device=dev)
    vals = torch.zeros(nlp.nnz, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()                                   # (torch fills on its stream, the evaluator runs on its own)
    ev.eval_assembled_device(JAC_ADJGRAD_HESS, X, L, fx, agx, vals)
    torch.cuda.synchronize()
    _, _, ref = nlp.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    assert rel_err(vals.cpu().numpy(), ref) < 1e-8
    ev.close()


@pytest.mark.parametrize("ode,nseg", [("reentry", 300), ("synthetic32", 7)])
def test_assembled_device_pointers_accumulate(oracle, ode, nseg):
    w, nlp, ev, _ = _setup(oracle, ode, "LGL7", nseg, False, accumulate=True)
    dev = torch.device("cuda:0")
    X, L = torch.from_numpy(w.X).to(dev), torch.from_numpy(w.L).to(dev)
    fx = torch.zeros(w.nseg * ev.OR, dtype=torch.float64, device=dev)
    agx = torch.zeros(w.nseg * ev.IR, dtype=torch.float64, device=dev)
    vals = torch.zeros(nlp.nnz, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()                                   # (torch fills on its stream, the evaluator runs on its own)
    for _ in range(2):                                         # two evaluations into the same array: it accumulates
        ev.eval_assembled_device(JAC_ADJGRAD_HESS, X, L, fx, agx, vals)
    torch.cuda.synchronize()
    _, _, ref = nlp.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    assert rel_err(vals.cpu().numpy(), 2.0 * ref) < 1e-8
    ev.close()


def test_assembly_argument_errors(oracle):
    w = Workload("brachistochrone", "LGL3", 8)
    ev = DefectEvaluator("brachistochrone", "LGL3", False, w.vindex, w.cindex, w.n_primal, w.n_equal)
    with pytest.raises(ValueError):
        ev.eval_assembled(JAC_ADJGRAD_HESS, w.X, w.L, np.zeros(10))
    ev._nvalues = 10
    with pytest.raises(_lib.AssetHipError, match="no kkt map"):
        ev.eval_assembled(JAC_ADJGRAD_HESS, w.X, w.L, np.zeros(10))
    bad = np.zeros((w.nseg, ev.NKKT), dtype=np.int32)
    bad[3, 5] = 99
    with pytest.raises(_lib.AssetHipError, match="outside"):
        ev.set_kkt_map(bad, 50)
    ev.close()


@pytest.mark.parametrize("ode,mode,nseg,blocked", [
    ("betts_lowthrust", "LGL5", 1000, False),        # BASELINE configs[1]: the phase parameter couples all 1 000 segments
    ("twobody_lt", "LGL5", 10000, True),             # BASELINE configs[3] dynamics, BlockConstant control, one phase's size
])
def test_full_size_assembly_and_rhs_on_the_device_are_exact_and_repeatable(oracle, ode, mode, nseg, blocked):
    """asset_hip_defect_eval_kkt_device at BASELINE sizes: KKT values, FXE and AGX assembled on the device against the
    oracle's evalKKT restatement, and bit-for-bit the same in two runs -- the many-way sums (parameter-parameter Hessian
    entries, the parameter's adjoint gradient) are staged / gathered in a fixed order, not added atomically."""
    w, nlp, ev, locs = _setup(oracle, ode, mode, nseg, blocked)
    counts = np.bincount(locs.ravel())
    if ode == "betts_lowthrust":
        assert counts.max() == nseg                          # the many-way location exists (H of the parameter with itself)
    dev = torch.device("cuda:0")
    X, L = torch.from_numpy(w.X).to(dev), torch.from_numpy(w.L).to(dev)
    runs = []
    for _ in range(2):
        FXE = torch.zeros(w.n_equal, dtype=torch.float64, device=dev)
        AGX = torch.zeros(w.n_primal, dtype=torch.float64, device=dev)
        vals = torch.zeros(nlp.nnz, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ev.eval_kkt_device(JAC_ADJGRAD_HESS, X, L, FXE, AGX, vals)
        torch.cuda.synchronize()
        runs.append((FXE.cpu().numpy(), AGX.cpu().numpy(), vals.cpu().numpy()))
    for a, b in zip(*runs):
        np.testing.assert_array_equal(a, b)                  # bitwise repeatable
    rFXE, rAGX, rvals = nlp.eval(JAC_ADJGRAD_HESS, w.X, w.L)
    FXE, AGX, vals = runs[0]
    assert np.abs(FXE - rFXE).max() < 1e-10 * max(1.0, np.abs(w.X).max())
    assert rel_err(AGX, rAGX) < 1e-8 and rel_err(vals, rvals) < 1e-8
    # value-only and gradient kinds through the same entry point
    FXE = torch.zeros(w.n_equal, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()                                   # (the fill runs on torch's stream, the evaluation on the handle's: without
    ev.eval_kkt_device(0, X, None, FXE, None, None)            #  this the fill can land AFTER the kernel's result -- seen once, round 6)
    torch.cuda.synchronize()
    assert np.abs(FXE.cpu().numpy() - rFXE).max() < 1e-10 * max(1.0, np.abs(w.X).max())
    ev.close()
