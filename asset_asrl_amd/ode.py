"""ODE objects: ``ODEArguments`` / ``ODEBase`` surface, the synthetic 32-state ODE and the registry of library ODEs.

``ODEArguments(Xv,Uv,Pv)`` and ``ODEBase(odefunc,Xv,Uv,Pv)`` keep the reference's names and
index conventions (input ``[x(Xv), t, u(Uv), p(Pv)]``; /root/reference/src/OptimalControl/
ODEArguments.h:22-40, ODESizes.h:7-118, asset_asrl/OptimalControl/ODEBaseClass.py:7-45).

The dynamics of the BASELINE.json configurations (the reference's example scripts restated in this DSL) live in
``workloads.py`` and are re-exported here under their old names; ``Synthetic32`` (defined by this
build, no reference counterpart) stays here.
"""
from __future__ import annotations

import numpy as np

from . import vf
from .vf.codegen import OdeDerivatives, differentiate
from .vf.functions import VectorFunction


class ODEArguments(VectorFunction):
    def __init__(self, Xv: int, Uv: int = 0, Pv: int = 0):
        n = Xv + 1 + Uv + Pv
        super().__init__(n, vf.Arguments(n).outs)
        self._xv, self._uv, self._pv = Xv, Uv, Pv

    def XVars(self):
        return self._xv

    def UVars(self):
        return self._uv

    def PVars(self):
        return self._pv

    def XVec(self):
        return self.segment(0, self._xv)

    def XVar(self, i):
        return self.XVec().coeff(i)

    def XtVec(self):
        return self.segment(0, self._xv + 1)

    def TVar(self):
        return self.coeff(self._xv)

    def UVec(self):
        return self.segment(self._xv + 1, self._uv)

    def UVar(self, i):
        return self.UVec().coeff(i)

    def PVec(self):
        return self.segment(self._xv + 1 + self._uv, self._pv)

    def PVar(self, i):
        return self.PVec().coeff(i)


class ODEBase:
    """Holds an ODE right-hand side and its (x,u,p) sizes; entry point to ``phase(...)``."""

    def __init__(self, odefunc: VectorFunction, Xvars: int, Uvars=None, Pvars=None, name: str | None = None):
        self.func = odefunc
        self._xv = int(Xvars)
        self._uv = int(Uvars or 0)
        self._pv = int(Pvars or 0)
        nin = self._xv + 1 + self._uv + self._pv
        if odefunc.IRows() != nin:
            raise ValueError(f"ODE function input size {odefunc.IRows()} does not match Xvars+1+Uvars+Pvars = {nin}")
        if odefunc.ORows() != self._xv:
            raise ValueError(f"ODE function output size {odefunc.ORows()} does not match Xvars = {self._xv}")
        self.ode_name = name or type(self).__name__.lower()
        self._derivs: OdeDerivatives | None = None

    # ---- reference size names ------------------------------------------------------
    def XVars(self):
        return self._xv

    def UVars(self):
        return self._uv

    def PVars(self):
        return self._pv

    def TVar(self):
        return self._xv

    def XtVars(self):
        return self._xv + 1

    def XtUVars(self):
        return self._xv + 1 + self._uv

    def XtUPVars(self):
        return self._xv + 1 + self._uv + self._pv

    def vf(self) -> VectorFunction:
        return self.func

    def derivatives(self) -> OdeDerivatives:
        if self._derivs is None:
            self._derivs = differentiate(self.ode_name, self.func, self._xv, self._uv, self._pv)
        return self._derivs

    def phase(self, mode, traj=None, nsegs=None):
        from .phase import Phase
        ph = Phase(self, mode)
        if traj is not None:
            ph.setTraj(traj, nsegs if nsegs is not None else max(len(traj) - 1, 1))
        return ph


# =========================================================================== library

class Synthetic32(ODEBase):
    """xdot_k = -a_k x_k + b_k sin(x_{k+1}) x_{k+5} + c_k cos t  (indices mod 32); a,b,c ~ U(0.5,1.5), seed 32."""

    def __init__(self, n: int = 32, seed: int = 32):
        rng = np.random.default_rng(seed)
        a, b, c = (rng.uniform(0.5, 1.5, n) for _ in range(3))
        args = ODEArguments(n)
        x = args.XVec().tolist()
        t = args.TVar()
        outs = [(-a[k]) * x[k] + b[k] * vf.sin(x[(k + 1) % n]) * x[(k + 5) % n] + c[k] * vf.cos(t)
                for k in range(n)]
        self.coeffs = (a, b, c)
        super().__init__(vf.stack(outs), n, name=f"synthetic{n}")


def __getattr__(name):
    """The workload definitions (``workloads.py``) and the registry of library ODEs under their old names here."""
    if name in ("Brachistochrone", "LTModel", "ShuttleReentry", "TwoBody", "ODE_LIBRARY"):
        from . import workloads
        return getattr(workloads, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
