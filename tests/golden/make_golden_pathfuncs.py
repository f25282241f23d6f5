"""Golden vectors for the other per-segment functions of a phase: mesh spacing, control spline, segment quadrature.

Only the VALUE formulas of the reference are restated here --
    SingleMeshSpacing   scale * (s (t_f - t_0) - (t_j - t_0))                       MeshSpacingConstraints.h:33-41
    LGLMeshSpacing      tc[i+1] - (t_{i+1} - t_0) / (t_{cs-1} - t_0)                  MeshSpacingConstraints.h:118-126
    LGLControlSpline    sum_i UOne[j][i] u_i / h0^(j+1) - UZero[j][i] u_{i+cs-1} / h1^(j+1)   LGLControlSplines.h:92-108
    LGLIntegral         (t_{cs-1} - t_0) sum_i w_i I([x_i, p])                          LGLIntegrals.h:18-52
-- evaluated in 50-digit mpmath arithmetic and differentiated exactly by the second-order forward AD of
make_golden.py; the weights come from tests/golden/lgl_tables.json (the reference header, parsed).  The oracle's closed
forms (oracle/pathfuncs.cpp) and the device are checked against tests/golden/pathfuncs.npz.

Usage:  python tests/golden/make_golden_pathfuncs.py
"""
from __future__ import annotations

import json
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import D2, MP  # noqa: E402

TAB = json.load(open(os.path.join(HERE, "lgl_tables.json")))["tables"]


def single_mesh_spacing(z, s, scale):
    return [((z[2] - z[0]) * s - (z[1] - z[0])) * scale]


def lgl_mesh_spacing(z, cs):
    tc = TAB[str(cs)]["CardinalSpacings"]
    h = z[cs - 1] - z[0]
    return [tc[i + 1] - (z[1 + i] - z[0]) / h for i in range(cs - 2)]


def control_spline(z, cs, usize, order):
    uone, uzero = TAB[str(cs)]["UOneSpline_Weights"], TAB[str(cs)]["UZeroSpline_Weights"]
    tu = usize + 1
    t = lambda i: z[i * tu]
    u = lambda i, k: z[i * tu + 1 + k]
    h0, h1 = t(cs - 1) - t(0), t(2 * cs - 2) - t(cs - 1)
    out = []
    for j in range(order):
        h0p, h1p = h0, h1
        for _ in range(j):
            h0p, h1p = h0p * h0, h1p * h1
        for k in range(usize):
            acc = 0
            for i in range(cs):
                acc = acc + (float(uone[j][i]) * u(i, k)) / h0p - (float(uzero[j][i]) * u(i + cs - 1, k)) / h1p
            out.append(acc)
    return out


def integrand_quad2(y, M):
    return y[1] * y[1] + y[0]


def integrand_powp(y, M):
    return y[3] * y[0] * y[0] + M.sin(y[1]) * y[2] + M.exp(-(y[0] * y[2])) / (1.0 + y[3] * y[3])


def lgl_integral(z, cs, xv, pv, integrand):
    w = TAB[str(cs)]["Reduced_Integral_Weights"]
    xtv = xv + 1
    h = z[(cs - 1) * xtv + xv] - z[xv]
    p = list(z[cs * xtv:])
    acc = 0
    for i in range(cs):
        acc = acc + float(w[i]) * integrand(list(z[i * xtv:i * xtv + xv]) + p, MP)
    return [acc * h]


def evaluate(fn, z, lam):
    n = len(z)
    zs = [D2.var(mp.mpf(float(v)), i, n) for i, v in enumerate(z)]
    d = fn(zs)
    fx = np.array([float(e.v) for e in d])
    jx = np.array([[float(e.g[i]) for i in range(n)] for e in d])
    g = sum((mp.mpf(float(lam[k])) * d[k].g for k in range(len(d))), np.full(n, mp.mpf(0), dtype=object))
    h = sum((mp.mpf(float(lam[k])) * d[k].h for k in range(len(d))), np.full((n, n), mp.mpf(0), dtype=object))
    return fx, jx, np.array([float(v) for v in g]), np.array([[float(v) for v in r] for r in h])


def nodes(rng, cs, two_segments=False):
    tc = np.array(TAB[str(cs)]["CardinalSpacings"])
    t0, h0 = rng.uniform(0, 5), rng.uniform(0.4, 2.0)
    t = t0 + tc * h0
    if two_segments:
        h1 = rng.uniform(0.4, 2.0)
        t = np.concatenate([t, t0 + h0 + tc[1:] * h1])
    return t + rng.uniform(-0.02, 0.02, t.size) * (np.arange(t.size) > 0)     # (not exactly on the LGL spacing)


def main():
    rng = np.random.default_rng(20260801)
    out = {}

    def add(name, fn, zgen, orr, nsample=3):
        xs, ls, res = [], [], []
        for _ in range(nsample):
            z = zgen()
            lam = rng.uniform(-2, 2, orr)
            xs.append(z), ls.append(lam), res.append(evaluate(fn, z, lam))
        out[name + "_x"], out[name + "_lam"] = np.array(xs), np.array(ls)
        for k, key in enumerate(("fx", "jx", "gx", "hx")):
            out[f"{name}_{key}"] = np.array([r[k] for r in res])
        print(name, out[name + "_hx"].shape, flush=True)

    s4 = TAB["4"]["CardinalSpacings"][1]
    add("single_mesh_spacing", lambda z: single_mesh_spacing(z, float(s4), 2.5), lambda: np.sort(rng.uniform(0, 3, 3)), 1)
    for cs in (3, 4):
        add(f"lgl_mesh_spacing{cs}", lambda z, cs=cs: lgl_mesh_spacing(z, cs), lambda cs=cs: nodes(rng, cs), cs - 2)
    for cs, usize in ((3, 2), (4, 2), (4, 1)):
        def zgen(cs=cs, usize=usize):
            t = nodes(rng, cs, True)
            return np.column_stack([t] + [rng.uniform(-1, 1, t.size) for _ in range(usize)]).ravel()
        add(f"control_spline{cs}_{usize}", lambda z, cs=cs, usize=usize: control_spline(z, cs, usize, cs - 2), zgen,
            usize * (cs - 2))
    for cs, xv, pv, name, fn in ((4, 2, 0, "quad2", integrand_quad2), (3, 3, 1, "powp", integrand_powp),
                                 (2, 3, 1, "powp", integrand_powp)):
        def zgen(cs=cs, xv=xv, pv=pv):
            t = nodes(rng, cs)
            return np.concatenate([np.column_stack([rng.uniform(-1, 1, (cs, xv)), t]).ravel(), rng.uniform(0.5, 1.5, pv)])
        add(f"lgl_integral{cs}_{name}", lambda z, cs=cs, xv=xv, pv=pv, fn=fn: lgl_integral(z, cs, xv, pv, fn), zgen, 1)
    np.savez_compressed(os.path.join(HERE, "pathfuncs.npz"), **out)
    print("wrote pathfuncs.npz")


if __name__ == "__main__":
    main()
