"""Randomised shapes: (ODE, transcription, control mode, segment count, index offsets) drawn from a fixed seed, every
evaluation kind against the oracle.  Catches what the hand-picked cases miss: segment counts that leave workgroups
empty or with one segment, counts just above a multiple of the group size, large variable offsets."""
import numpy as np
import pytest

from asset_asrl_amd import _lib
from asset_asrl_amd.evaluator import CON, CON_ADJGRAD, JAC, JAC_ADJGRAD, JAC_ADJGRAD_HESS, DefectEvaluator
from helpers import Workload
from test_gpu_parity import _check_blocks

pytestmark = pytest.mark.gpu


def _cases(n, seed=20261002):
    rng = np.random.default_rng(seed)
    odes = ["brachistochrone", "reentry", "twobody_lt", "betts_lowthrust"]
    modes = ["Trapezoidal", "LGL3", "LGL5", "LGL7"]
    out = []
    for _ in range(n):
        nseg = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 63, 255, 256, 257, 1023, 1025, 2047, 2049, 4100, 6151]))
        out.append((str(rng.choice(odes)), str(rng.choice(modes)), bool(rng.integers(2)), nseg,
                    int(rng.integers(0, 50)), int(rng.integers(0, 20)), int(rng.integers(0, 9))))
    return out


@pytest.mark.parametrize("ode,mode,blocked,nseg,voff,coff,extra", _cases(24))
def test_random_shape(oracle, ode, mode, blocked, nseg, voff, coff, extra):
    if not _lib.has_kernel(ode, _lib.MODES[mode], blocked and ode != "synthetic32"):
        pytest.skip("not instantiated")
    w = Workload(ode, mode, nseg, blocked, seed=nseg + voff, var_offset=voff, con_offset=coff, extra_vars=extra)
    nlp = w.oracle_nlp(oracle, threads=8)
    ev = DefectEvaluator(ode, mode, w.blocked, w.vindex, w.cindex, w.n_primal, w.n_equal)
    for what in (JAC_ADJGRAD_HESS, JAC_ADJGRAD, CON):
        ref = nlp.eval_blocks(what, w.X, w.L)
        got = ev.eval(what, w.X, w.L if what in (CON_ADJGRAD, JAC_ADJGRAD, JAC_ADJGRAD_HESS) else None)
        _check_blocks(got, ref, w, what)
    ev.close()
