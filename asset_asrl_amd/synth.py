"""Synthetic random-state meshes for the BASELINE.json configs.

One definition of "the workload" shared by bench.py, the parity tests and the golden-vector
generator: node times strictly increasing with the LGL cardinal spacing inside each segment,
states/controls uniform in ranges that keep every config ODE well defined, multipliers
``100*U(-1,1)`` (the reference's own NLPTest recipe, /root/reference/src/Solvers/
NonLinearProgram.cpp:708-710).  Pure numpy; nothing here runs in the timed region.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

SEED = 20260723

# cardinal spacings (fraction of a segment) -- values restated from LGLCoeffs.h CardinalSpacings
_TC = {
    2: np.array([0.0, 1.0]),
    3: np.array([0.0, 0.5, 1.0]),
    4: np.array([0.0, 2.65575603264643e-1, 7.34424396735357e-1, 1.0]),
}

MODE_CS = {"Trapezoidal": 2, "LGL3": 2, "LGL5": 3, "LGL7": 4}

# name -> (xv, uv, pv)
ODE_SIZES: Dict[str, Tuple[int, int, int]] = {
    "brachistochrone": (3, 1, 0),
    "reentry": (5, 2, 0),
    "twobody_lt": (6, 3, 0),
    "betts_lowthrust": (7, 3, 1),
    "synthetic32": (32, 0, 0),
}


def _states(name: str, rng: np.random.Generator, nnodes: int, sizes=None):
    """Returns (X[nnodes,xv], U[nnodes,uv], P[pv]) in well-conditioned ranges.  ``sizes`` = (xv, uv, pv) describes
    an ODE outside the config table (user-defined): states/controls ~ U(-1,1), parameters ~ U(0.5,1.5)."""
    xv, uv, pv = sizes if sizes is not None else ODE_SIZES[name]
    u = rng.uniform
    if name == "brachistochrone":
        X = np.column_stack([u(0, 10, nnodes), u(0, 10, nnodes), u(0.5, 10, nnodes)])
        U = u(0.1, 1.4, (nnodes, 1))
        P = np.zeros(0)
    elif name == "reentry":
        X = np.column_stack([u(0.8, 2.6, nnodes), u(-0.5, 0.5, nnodes), u(1.5, 15.0, nnodes),
                             u(-0.1, 0.1, nnodes), u(0.2, 1.5, nnodes)])
        U = np.column_stack([u(0.1, 0.5, nnodes), u(-1.0, 0.5, nnodes)])
        P = np.zeros(0)
    elif name == "twobody_lt":
        d = rng.normal(size=(nnodes, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        X = np.column_stack([d * u(0.8, 1.2, (nnodes, 1)), u(-1, 1, (nnodes, 3))])
        U = u(-1, 1, (nnodes, 3))
        P = np.zeros(0)
    elif name == "betts_lowthrust":
        X = np.column_stack([u(1.0, 2.0, nnodes), u(-0.2, 0.2, nnodes), u(-0.2, 0.2, nnodes),
                             u(-0.5, 0.5, nnodes), u(-0.5, 0.5, nnodes), u(0.0, 6.0, nnodes),
                             u(0.3, 1.0, nnodes)])
        d = rng.normal(size=(nnodes, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        U = d * u(0.7, 1.3, (nnodes, 1))
        P = u(-50.0, 0.0, 1)
    elif name == "synthetic32":
        X = u(-1, 1, (nnodes, 32))
        U = np.zeros((nnodes, 0))
        P = np.zeros(0)
    elif sizes is not None:
        X, U, P = u(-1, 1, (nnodes, xv)), u(-1, 1, (nnodes, uv)), u(0.5, 1.5, pv)
    else:
        raise KeyError(name)
    assert X.shape[1] == xv and U.shape[1] == uv and P.size == pv
    return X, U, P


def make_traj(name: str, mode: str, nseg: int, seed: int = SEED, T: float = 10.0, sizes=None) -> np.ndarray:
    """Trajectory ``[nnodes, xv+1+uv+pv]`` in the reference's ``setTraj`` row layout [x,t,u,p]."""
    cs = MODE_CS[mode]
    K = cs - 1
    nnodes = K * nseg + 1
    rng = np.random.default_rng(seed)
    X, U, P = _states(name, rng, nnodes, sizes)
    h0 = T / nseg
    t = np.empty(nnodes)
    for j in range(K):
        t[j:nnodes - 1:K] = (np.arange(nseg) + _TC[cs][j]) * h0
    t[-1] = T
    Pm = np.tile(P, (nnodes, 1))
    return np.column_stack([X, t, U, Pm])


def make_multipliers(nrows: int, seed: int = SEED + 1) -> np.ndarray:
    return 100.0 * np.random.default_rng(seed).uniform(-1.0, 1.0, nrows)
