"""Shared workload construction for the parity tests (oracle side lives in oracle/, product in asset_asrl_amd/)."""
from __future__ import annotations

import numpy as np

from asset_asrl_amd import synth
from asset_asrl_amd.indexing import PhaseIndexer


class Workload:
    """A synthetic phase: solver vector X, multipliers L and the defect index tables."""

    def __init__(self, ode: str, mode: str, nseg: int, blocked: bool = False, seed: int = synth.SEED,
                 var_offset: int = 0, con_offset: int = 0, extra_vars: int = 0, sizes=None):
        xv, uv, pv = sizes if sizes is not None else synth.ODE_SIZES[ode]
        self.ode, self.mode, self.nseg, self.blocked = ode, mode, nseg, bool(blocked) and uv > 0
        self.cs = synth.MODE_CS[mode]
        self.traj = synth.make_traj(ode, mode, nseg, seed=seed, sizes=sizes)
        ix = PhaseIndexer(xv, uv, pv, 0)
        ix.set_dimensions(self.cs, nseg, self.blocked)
        ix.begin_indexing(var_offset, con_offset)
        self.indexer = ix
        self.vindex, self.cindex = ix.make_defect_Vindex_Cindex()
        self.IR, self.OR = ix.defect_sizes()
        self.n_primal = var_offset + ix.numPhaseVars + extra_vars
        self.n_equal = con_offset + ix.numPhaseEqCons
        rng = np.random.default_rng(seed + 7)
        self.X = rng.uniform(-1, 1, self.n_primal)
        self.X[var_offset:var_offset + ix.numPhaseVars] = ix.makeSolverInput(self.traj)
        self.L = synth.make_multipliers(self.n_equal, seed=seed + 1)
        self.NKKT = self.IR * (self.IR + 1) // 2 + self.OR * self.IR

    def oracle_nlp(self, ob, threads: int = 1, provider: int = 0):
        return ob.Nlp(ob.get_ode(self.ode, provider), ob.MODES[self.mode], self.blocked, self.vindex, self.cindex,
                      self.n_primal, self.n_equal, threads)


def rel_err(a, b, floor=1.0):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(floor, float(np.max(np.abs(b)))))


def make_vanderpol():
    """A user-defined ODE (not compiled into libasset_hip.so): forced Van der Pol oscillator with one control and one
    parameter, x0' = x1, x1' = mu (1 - x0^2) x1 - x0 + u exp(-t/10).  The oracle holds the same right-hand side as
    ``vanderpol`` (oracle/odes.h), differentiated independently by AD2."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase

    class VanDerPol(ODEBase):
        def __init__(self):
            a = ODEArguments(2, 1, 1)
            x0, x1 = a.XVec().tolist()
            u, mu, t = a.UVar(0), a.PVar(0), a.TVar()
            super().__init__(vf.stack([x1, mu * (1.0 - x0 * x0) * x1 - x0 + u * vf.exp(-0.1 * t)]), 2, 1, 1,
                             name="vanderpol")

    return VanDerPol()


def make_coupled(n: int = 12):
    """A wider user-defined ODE, (n, 3, 2): x_k' = -x_k/2 + sin(x_{k+1}) x_{k+5} u_{k mod 3} + p0 cos t + p1 x_k x_{k+7}
    (indices mod n).  n = 12: in LGL7 a segment has IR = 66 inputs, so its dense stage is the four-wave kernel -- with
    control rows and parameter columns, which the 32-state BASELINE ODE does not have; n = 16: the BlockConstant form
    (IR = 73) is wide too, with five parameter columns.  The oracle holds the same right-hand sides as ``coupled12`` /
    ``coupled16`` (oracle/odes.h)."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase

    class Coupled(ODEBase):
        def __init__(self):
            a = ODEArguments(n, 3, 2)
            x = a.XVec().tolist()
            t, p0, p1 = a.TVar(), a.PVar(0), a.PVar(1)
            u = [a.UVar(k) for k in range(3)]
            rhs = [-0.5 * x[k] + vf.sin(x[(k + 1) % n]) * x[(k + 5) % n] * u[k % 3] + p0 * vf.cos(t) + p1 * x[k] * x[(k + 7) % n]
                   for k in range(n)]
            super().__init__(vf.stack(rhs), n, 3, 2, name=f"coupled{n}")

    return Coupled()


def make_coupled12():
    return make_coupled(12)
