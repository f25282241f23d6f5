"""Mesh-error estimation of a phase trajectory on the device.

``mesh_error_deboor`` is the de Boor estimator of the reference (``ODEPhase<DODE>::get_meshinfo_deboor``,
/root/reference/src/OptimalControl/ODEPhase.h:442-585); ``mesh_info`` adds what ``ODEPhaseBase::getMeshInfo`` does
with it on the host (ODEPhaseBase.h:1355-1399): per-block infinity norms, the cumulative node-density integral and
the equidistributed bin edges for ``n`` new segments."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

_dp = C.POINTER(C.c_double)


def mesh_error_deboor(ode_name: str, mode: str, traj, blocked: bool = False, device: int = 0):
    """Returns (tsnd[nb+1], mesh_errors[XV, nb+1], mesh_dist[XV, nb+1], error[nb+1], dist[nb+1])."""
    T = np.ascontiguousarray(traj, dtype=np.float64)
    xv, uv, pv = _lib.ode_sizes(ode_name)
    if T.ndim != 2 or T.shape[1] != xv + 1 + uv + pv:
        raise ValueError(f"trajectory rows must have {xv + 1 + uv + pv} columns [x,t,u,p]")
    cs = 2 if mode == "Trapezoidal" else _lib.MODES[mode]
    nb = (T.shape[0] - 1) // (cs - 1)
    tsnd = np.empty(nb + 1)
    err, dist = np.empty((nb + 1, xv)), np.empty((nb + 1, xv))
    emax, dmax = np.empty(nb + 1), np.empty(nb + 1)
    p = lambda a: a.ctypes.data_as(_dp)
    _lib.check(_lib.lib().asset_hip_mesh_error_deboor(ode_name.encode(), _lib.MODES[mode], int(blocked), p(T), T.shape[0],
                                                      p(tsnd), p(err), p(dist), p(emax), p(dmax), int(device)),
               "asset_hip_mesh_error_deboor")
    return tsnd, err.T.copy(), dist.T.copy(), emax, dmax


def bins_from_density(tsnd, error, dist, n: int):
    """(tsnd, bins, error): cumulative node-density integral and the equidistributed edges of ``n`` new segments
    (ODEPhaseBase.h:1372-1398)."""
    distint = np.zeros_like(dist)
    distint[1:] = np.cumsum(dist[:-1] * np.diff(tsnd))
    distint /= distint[-1]
    bins = np.linspace(0.0, 1.0, n + 1)
    elem = 0
    for i in range(1, n):
        di = i / n
        elem = int(np.searchsorted(distint[elem:], di, side="right")) + elem - 1     # std::upper_bound from `elem`
        t0, t1, d0, d1 = tsnd[elem], tsnd[elem + 1], distint[elem], distint[elem + 1]
        bins[i] = (di - d0) / ((d1 - d0) / (t1 - t0)) + t0
    return tsnd, bins, error


def mesh_info(ode_name: str, mode: str, traj, n: int, blocked: bool = False, device: int = 0):
    """(tsnd, bins, error) -- ODEPhaseBase::getMeshInfo(False, n)."""
    tsnd, _, _, error, dist = mesh_error_deboor(ode_name, mode, traj, blocked, device)
    return bins_from_density(tsnd, error, dist, n)


class MeshIterateInfo:
    """One iterate of the adaptive mesh loop (MeshIterateInfo.h:6-86): the estimate on the current mesh, its summary numbers and the
    cumulative error-density integral the next mesh's edges are read from."""

    def __init__(self, numsegs: int, tol: float, times, error, distribution):
        self.numsegs = self.up_numsegs = int(numsegs)
        self.tol, self.converged, self.global_error = float(tol), False, -1.0
        self.times, self.error, self.distribution = (np.asarray(v, dtype=float).copy() for v in (times, error, distribution))
        hs = np.diff(self.times)
        self.max_error = float(self.error.max())
        self.avg_error = float((self.error[:-1] * hs).sum())
        self.gmean_error = float(np.exp((np.log(self.max_error) + np.log(self.avg_error)) / 2.0))
        self.distintegral = np.zeros_like(self.times)
        self.distintegral[1:] = np.cumsum(self.distribution[:-1] * hs)
        self.distintegral /= self.distintegral[-1]

    def calc_bins(self, nbins: int):
        return bins_from_density(self.times, self.error, self.distribution, int(nbins))[1]
