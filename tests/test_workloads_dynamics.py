"""The library ODEs (asset_asrl_amd/workloads.py) against the fixture tests/golden/dynamics/workload_dynamics.npz: value, Jacobian,
adjoint gradient and adjoint Hessian that round 4's definitions of the same dynamics produced at seeded points
(tools/dump_workload_dynamics.py, run at commit aa3db1b).  Round 5 restated every model from its equations of motion -- the
low-thrust model in closed form in the equinoctial elements, without the Cartesian detour -- so the comparison is to rounding,
not bitwise: 1e-13 of each block's scale (measured: 2e-16 ... 4e-15)."""
import os

import numpy as np
import pytest

from asset_asrl_amd.vf.ir import evaluate
from asset_asrl_amd.workloads import ODE_LIBRARY

FIX = os.path.join(os.path.dirname(__file__), "golden", "dynamics", "workload_dynamics.npz")
TOL = 1e-13


@pytest.mark.parametrize("name", sorted(ODE_LIBRARY))
def test_restated_dynamics_match_round4_definitions(name):
    z = np.load(FIX)
    Y, Lm = z[f"{name}/y"], z[f"{name}/lam"]
    d = ODE_LIBRARY[name]().derivatives()
    N, n = d.nin, d.xv
    roots = list(d.f) + [e for r in d.J for e in r] + list(d.g) + [d.H[i][j] for i in range(N) for j in range(i + 1)]
    worst = 0.0
    for k in range(Y.shape[0]):
        v = np.array(evaluate(roots, Y[k], Lm[k]))
        f, J, g = v[:n], v[n:n + n * N].reshape(n, N), v[n + n * N:n + n * N + N]
        H = np.zeros((N, N))
        H[np.tril_indices(N)] = v[n + n * N + N:]
        H = H + np.tril(H, -1).T
        for got, key in ((f, "f"), (J, "J"), (g, "g"), (H, "H")):
            ref = z[f"{name}/{key}"][k]
            err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
            worst = max(worst, err)
            assert err < TOL, (name, key, k, err)
    print(name, "worst relative difference", worst)
