"""Dump value / Jacobian / adjoint gradient / adjoint Hessian of every library ODE (asset_asrl_amd.workloads.ODE_LIBRARY) at
seeded points, by the host-side walk of the expression DAG (vf.ir.evaluate): the fixture tests/golden/dynamics/workload_dynamics.npz.

The fixture in the tree was written by this script at commit aa3db1b (round 4's definitions of the dynamics), BEFORE round 5
restated them from their mathematical models; tests/test_workloads_dynamics.py holds the restated definitions to it.

    python tools/dump_workload_dynamics.py tests/golden/dynamics/workload_dynamics.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from asset_asrl_amd.vf.ir import GRAPH as G            # noqa: E402
from asset_asrl_amd.vf.ir import evaluate               # noqa: E402
from asset_asrl_amd.workloads import ODE_LIBRARY        # noqa: E402

NPTS = 24


def points(name, nin, rng):
    """Points inside each model's domain of definition (positive radius / speed / semi-latus rectum / weight ...)."""
    y = rng.uniform(-1.0, 1.0, (NPTS, nin))
    if name == "reentry":                  # h, theta, v, gamma, psi | t | alpha, beta
        y[:, 0] = rng.uniform(0.3, 2.6, NPTS)
        y[:, 1] = rng.uniform(-1.2, 1.2, NPTS)
        y[:, 2] = rng.uniform(0.5, 16.0, NPTS)
        y[:, 3] = rng.uniform(-1.2, 1.2, NPTS)
    elif name == "twobody_lt":
        y[:, :3] += np.sign(y[:, :3]) * 0.3
    elif name == "betts_lowthrust":        # p f g h k L w | t | u(3) | tau
        y[:, 0] = rng.uniform(1.0, 2.5, NPTS)
        y[:, 1:5] = rng.uniform(-0.4, 0.4, (NPTS, 4))
        y[:, 5] = rng.uniform(0.0, 12.0, NPTS)
        y[:, 6] = rng.uniform(0.2, 1.0, NPTS)
        y[:, 8:11] += np.sign(y[:, 8:11]) * 0.2
        y[:, 11] = rng.uniform(-40.0, 0.0, NPTS)
    return y


def main(out):
    rng = np.random.default_rng(20261003)
    data = {}
    for name, cls in ODE_LIBRARY.items():
        ode = cls()
        d = ode.derivatives()
        N, n = d.nin, d.xv
        Y = points(name, N, rng)
        Lm = rng.uniform(-1.0, 1.0, (NPTS, n))
        roots = list(d.f) + [e for r in d.J for e in r] + list(d.g) + [d.H[i][j] for i in range(N) for j in range(i + 1)]
        F = np.empty((NPTS, n)); J = np.empty((NPTS, n, N)); g = np.empty((NPTS, N)); H = np.zeros((NPTS, N, N))
        for k in range(NPTS):
            v = np.array(evaluate(roots, Y[k], Lm[k]))
            F[k] = v[:n]; J[k] = v[n:n + n * N].reshape(n, N); g[k] = v[n + n * N:n + n * N + N]
            h = v[n + n * N + N:]
            e = 0
            for i in range(N):
                for j in range(i + 1):
                    H[k, i, j] = H[k, j, i] = h[e]; e += 1
        for key, a in (("y", Y), ("lam", Lm), ("f", F), ("J", J), ("g", g), ("H", H)):
            data[f"{name}/{key}"] = a
        print(name, d.stats())
    np.savez_compressed(out, **data)


if __name__ == "__main__":
    main(sys.argv[1])
