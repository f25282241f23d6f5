"""Generate the golden vectors under tests/golden/ (run once, here; commit the .npz files).

What is computed: the defect *value* formula of the reference --
    d_i = sum_j (C_ij x_j + h D_ij f_j) + h E_i f(x^_i, tau_i, u^_i, P),   x^_i = sum_j (A_ij x_j + h B_ij f_j)
(/root/reference/src/OptimalControl/LGLDefects.h:57-122) and, for Trapezoidal,
    d = -[(x_1 - x_0) - (h/2)(f_0 + f_1)]          (TrapezoidalDefects.h:146-184)
-- evaluated with 50-digit mpmath arithmetic, and differentiated *exactly* by second-order
forward AD carried out in the same 50-digit arithmetic.  Nothing of the reference's (or this
repo's) Jacobian / adjoint-Hessian algorithms is used, so the vectors pin
``computeall(x,l) = (fx, jx, gx=jx^T l, hx=sum_k l_k grad^2 d_k)`` independently.

The ODE right-hand sides are written here a third time (after asset_asrl_amd/ode.py and
oracle/odes.h), from the reference example scripts, over a generic scalar.  All model constants
are first formed in IEEE double exactly as the example scripts do and then promoted, so the golden
function is the same real-valued function the double codes implement.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import math
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from asset_asrl_amd import synth  # noqa: E402
import json  # noqa: E402

# The collocation weights come from the reference's own header, parsed by tests/golden/parse_lglcoeffs.py into
# tests/golden/lgl_tables.json -- NOT from the oracle (or the product): the vectors must not share their only copy of
# the coefficients with the code they check.
_REF_NAMES = {"s": "InteriorSpacings", "A": "Cardinal_XInterp_Weights", "B": "Cardinal_DXInterp_Weights",
              "U": "Cardinal_UPoly_Weights", "C": "Cardinal_XDef_Weights", "D": "Cardinal_DXDef_Weights",
              "E": "Interior_DXDef_Weights"}
_REF_TABLES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lgl_tables.json")))["tables"]


def ref_table(cs, which):
    return np.array(_REF_TABLES[str(cs)][_REF_NAMES[which]], dtype=float)


def synthetic32_coeffs(n: int = 32, seed: int = 32) -> np.ndarray:
    """a, b, c ~ U(0.5, 1.5), seed 32 -- the definition of the synthetic ODE, restated here."""
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.uniform(0.5, 1.5, n) for _ in range(3)])

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- AD scalar
class D2:
    """value + gradient + Hessian w.r.t. N inputs, all mpf (numpy object arrays)."""

    __slots__ = ("v", "g", "h")

    def __init__(self, v, g, h):
        self.v, self.g, self.h = v, g, h

    @staticmethod
    def const(c, n):
        return D2(mp.mpf(c), np.full(n, mp.mpf(0), dtype=object), np.full((n, n), mp.mpf(0), dtype=object))

    @staticmethod
    def var(val, i, n):
        d = D2.const(val, n)
        d.g[i] = mp.mpf(1)
        return d

    def _n(self):
        return self.g.shape[0]

    def _lift(self, o):
        return o if isinstance(o, D2) else D2.const(o, self._n())

    def _un(self, v, d1, d2):
        return D2(v, d1 * self.g, d1 * self.h + d2 * np.outer(self.g, self.g))

    def __add__(self, o):
        o = self._lift(o)
        return D2(self.v + o.v, self.g + o.g, self.h + o.h)

    __radd__ = __add__

    def __neg__(self):
        return D2(-self.v, -self.g, -self.h)

    def __sub__(self, o):
        return self + (-self._lift(o))

    def __rsub__(self, o):
        return self._lift(o) - self

    def __mul__(self, o):
        if not isinstance(o, D2):
            c = mp.mpf(o)
            return D2(self.v * c, self.g * c, self.h * c)
        og = np.outer(self.g, o.g)
        return D2(self.v * o.v, self.g * o.v + o.g * self.v, self.h * o.v + o.h * self.v + og + og.T)

    __rmul__ = __mul__

    def recip(self):
        r = 1 / self.v
        return self._un(r, -r * r, 2 * r * r * r)

    def __truediv__(self, o):
        if not isinstance(o, D2):
            return self * (1 / mp.mpf(o))
        return self * o.recip()

    def __rtruediv__(self, o):
        return self.recip() * o

    def __pow__(self, k):
        if isinstance(k, int):
            if k == 0:
                return D2.const(1, self._n())
            r = self
            for _ in range(abs(k) - 1):
                r = r * self
            return r if k > 0 else r.recip()
        k = mp.mpf(k)
        p = self.v ** k
        return self._un(p, k * p / self.v, k * (k - 1) * p / (self.v * self.v))


class MP:
    """math namespace for D2 / mpf"""

    @staticmethod
    def sin(a):
        s, c = mp.sin(a.v), mp.cos(a.v)
        return a._un(s, c, -s)

    @staticmethod
    def cos(a):
        s, c = mp.sin(a.v), mp.cos(a.v)
        return a._un(c, -s, -c)

    @staticmethod
    def tan(a):
        t = mp.tan(a.v)
        d = 1 + t * t
        return a._un(t, d, 2 * t * d)

    @staticmethod
    def exp(a):
        e = mp.exp(a.v)
        return a._un(e, e, e)

    @staticmethod
    def sqrt(a):
        r = mp.sqrt(a.v)
        return a._un(r, 1 / (2 * r), -1 / (4 * r * a.v))


# --------------------------------------------------------------------------- ODEs (generic scalar)
def ode_brachistochrone(y, M):
    g = 9.81
    v, theta = y[2], y[4]
    return [M.sin(theta) * v, -1.0 * M.cos(theta) * v, g * M.cos(theta)]


def ode_reentry(y, M):
    g0, W = 32.2, 203000
    Lstar, Tstar = 100000.0, 60.0
    Mstar = W / g0
    Rhostar = Mstar / (Lstar ** 3)
    Mustar = (Lstar ** 3) / (Tstar ** 2)
    Re = 20902900 / Lstar
    S = 2690.0 / (Lstar ** 2)
    m = (W / g0) / Mstar
    mu = (0.140765e17) / Mustar
    rho0 = .002378 / Rhostar
    h_ref = 23800 / Lstar
    a0, a1 = -.20704, .029244
    b0, b1, b2 = .07854, -.61592e-2, .621408e-3
    h, theta, v, gamma, psi = y[0:5]
    alpha, beta = y[6], y[7]
    alphadeg = (180.0 / np.pi) * alpha
    CL = a0 + a1 * alphadeg
    CD = b0 + b1 * alphadeg + b2 * (alphadeg ** 2)
    rho = rho0 * M.exp(-h / h_ref)
    r = h + Re
    L = 0.5 * CL * S * rho * (v ** 2)
    Dd = 0.5 * CD * S * rho * (v ** 2)
    g = mu / (r ** 2)
    sgam, cgam = M.sin(gamma), M.cos(gamma)
    sbet, cbet = M.sin(beta), M.cos(beta)
    spsi, cpsi = M.sin(psi), M.cos(psi)
    tantheta = M.tan(theta)
    return [v * sgam,
            (v / r) * cgam * cpsi,
            -Dd / m - g * sgam,
            (L / (m * v)) * cbet + cgam * (v / r - g / v),
            L * sbet / (m * v * cgam) + (v / r) * cgam * spsi * tantheta]


def ode_twobody_lt(y, M):
    P1mu, ltacc = 1.0, 0.01
    r2 = y[0] * y[0] + y[1] * y[1] + y[2] * y[2]
    rn = M.sqrt(r2)
    r3 = rn * rn * rn
    out = [y[3], y[4], y[5]]
    for i in range(3):
        out.append((y[i] / r3) * (-P1mu) + y[7 + i] * ltacc)
    return out


def _norm3(a, M):
    return M.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])


def _unit3(a, M):
    n = _norm3(a, M)
    return [a[0] / n, a[1] / n, a[2] / n]


def _cross3(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def ode_betts(y, M):
    g0, W, mu_e, Lstar = 32.174, 1, 1.407645794e16, 20925662.73
    Tstar = Lstar / np.sqrt(mu_e / Lstar)
    Mstar = W / g0
    Fstar = Mstar * Lstar / (Tstar ** 2)
    Astar = Lstar / (Tstar ** 2)
    Mustar = (Lstar ** 3) / (Tstar ** 2)
    Re = 20925662.73 / Lstar
    mu = float(mu_e / Mustar)
    T = 4.446618e-3 / Fstar
    Isp = 450 / Tstar
    gs = g0 / Astar
    J2, J3, J4 = 1082.639e-6, -2.565e-6, -1.608e-6
    p, f, g, h, k, L, ww = y[0:7]
    U = _unit3(y[8:11], M)
    tau = y[11]
    sinL, cosL = M.sin(L), M.cos(L)
    # MEECartFunc
    sqp = M.sqrt(mu / p)
    w = 1 + f * cosL + g * sinL
    s2 = 1 + h ** 2 + k ** 2
    a2 = h ** 2 - k ** 2
    r = p / w
    r_s2 = r / s2
    subs2 = 1.0 / s2
    R = [r_s2 * (cosL + a2 * cosL + 2. * h * k * sinL),
         r_s2 * (sinL - a2 * sinL + 2. * h * k * cosL),
         r_s2 * (2.0 * (h * sinL - k * cosL))]
    vs = -subs2 * sqp
    V = [vs * (sinL + a2 * sinL - 2. * h * k * cosL + g - 2. * f * h * k + a2 * g),
         vs * (-cosL + a2 * cosL + 2. * h * k * sinL - f + 2. * g * h * k + a2 * f),
         vs * (-2.0 * (h * cosL + k * sinL + f * h + g * k))]
    # ZonalGrav
    rn = _norm3(R, M)
    Ir = _unit3(R, M)
    IrN = Ir[2]
    In = _unit3([0.0 - Ir[0] * IrN, 0.0 - Ir[1] * IrN, 1.0 - Ir[2] * IrN], M)
    sphi = Ir[2]
    cphi = M.sqrt(1 - sphi ** 2)
    P2 = 0.5 * (3.0 * (sphi ** 2) - 1.0)
    P3 = 0.5 * (5.0 * (sphi ** 3) - 3 * sphi)
    P4 = (35 / 8) * (sphi ** 4) - (30 / 8) * (sphi ** 2) + 3 / 8
    D2_ = 3 * sphi
    D3 = 0.5 * (15.0 * (sphi ** 2) - 3.0)
    D4 = (35 / 2) * (sphi ** 3) - (30 / 4) * sphi
    Rr = Re / rn
    gn = (D2_ * J2 * (Rr ** 2) + D3 * J3 * (Rr ** 3) + D4 * J4 * (Rr ** 4)) * cphi
    gr = (3 * P2 * J2) * (Rr ** 2) + (4 * P3 * J3) * (Rr ** 3) + (5 * P4 * J4) * (Rr ** 4)
    gsc = -mu / (R[0] * R[0] + R[1] * R[1] + R[2] * R[2])
    Gc = [(gn * In[i] - gr * Ir[i]) * gsc for i in range(3)]
    Nhat = _unit3(_cross3(R, V), M)
    That = _unit3(_cross3(Nhat, R), M)
    accJ = [sum(B[i] * Gc[i] for i in range(3)) for B in (Ir, That, Nhat)]
    thr = gs * T * (1 + .01 * tau)
    ur, ut, un = [thr * U[i] / ww + accJ[i] for i in range(3)]
    # MEEDynamics2
    sq = M.sqrt(p) / np.sqrt(mu)
    hs = h * sinL - k * cosL
    hk = (s2 * un / w) / 2.0
    out = [2. * (p / w) * ut,
           ur * sinL + ((w + 1) * cosL + f) * (ut / w) - hs * (g * un / w),
           -ur * cosL + ((w + 1) * sinL + g) * (ut / w) + hs * (f * un / w),
           cosL * hk,
           sinL * hk,
           mu * (w / p) * (w / p) + (1.0 / w) * hs * un]
    out = [o * sq for o in out]
    out.append(-T * (1 + .01 * tau) / Isp)
    return out


def ode_synthetic32(y, M):
    n = 32
    abc = synthetic32_coeffs()
    a, b, c = abc[:n], abc[n:2 * n], abc[2 * n:]
    ct = M.cos(y[n])
    return [(-float(a[k])) * y[k] + float(b[k]) * M.sin(y[(k + 1) % n]) * y[(k + 5) % n] + float(c[k]) * ct
            for k in range(n)]


ODES = {"brachistochrone": ode_brachistochrone, "reentry": ode_reentry, "twobody_lt": ode_twobody_lt,
        "betts_lowthrust": ode_betts, "synthetic32": ode_synthetic32}


# --------------------------------------------------------------------------- defect value formula
def defect_value(name, mode, blocked, z, M):
    """z: list of scalars (IR).  Returns list of OR scalars."""
    xv, uv, pv = synth.ODE_SIZES[name]
    n = xv
    m, p = (0, uv + pv) if blocked else (uv, pv)
    q = n + 1 + m
    cs = synth.MODE_CS[mode]
    ode = ODES[name]
    P = z[cs * q:]
    card = [list(z[j * q:(j + 1) * q]) + list(P) for j in range(cs)]
    fj = [ode(c, M) for c in card]
    h = card[-1][n] - card[0][n]
    if mode == "Trapezoidal":
        return [-((card[1][k] - card[0][k]) - (h / 2.0) * (fj[0][k] + fj[1][k])) for k in range(n)]
    tab = {k: ref_table(cs, k) for k in "sABUCDE"}
    out = []
    for i in range(cs - 1):
        xi = []
        for k in range(n):
            acc = 0
            for j in range(cs):
                acc = acc + (float(tab["A"][i][j]) * card[j][k] + (float(tab["B"][i][j]) * h) * fj[j][k])
            xi.append(acc)
        ti = card[0][n] + h * float(tab["s"][i])
        ui = []
        for k in range(m):
            acc = 0
            for j in range(cs):
                acc = acc + float(tab["U"][i][j]) * card[j][n + 1 + k]
            ui.append(acc)
        fi = ode(xi + [ti] + ui + list(P), M)
        for k in range(n):
            acc = 0
            for j in range(cs):
                acc = acc + (float(tab["C"][i][j]) * card[j][k] + (float(tab["D"][i][j]) * h) * fj[j][k])
            out.append(acc + (h * float(tab["E"][i])) * fi[k])
    return out


def segment_input(name, mode, blocked, traj, seg):
    xv, uv, pv = synth.ODE_SIZES[name]
    cs = synth.MODE_CS[mode]
    K = cs - 1
    rows = traj[seg * K: seg * K + cs]
    if blocked:
        z = [rows[j, :xv + 1] for j in range(cs)] + [rows[0, xv + 1:]]
    else:
        z = [rows[j, :xv + 1 + uv] for j in range(cs)] + [rows[0, xv + 1 + uv:]]
    return np.concatenate(z)


def golden_case(name, mode, blocked, nseg_mesh, segs, seed):
    traj = synth.make_traj(name, mode, nseg_mesh, seed=seed)
    xs, ls, fxs, jxs, gxs, hxs = [], [], [], [], [], []
    for s in segs:
        z = segment_input(name, mode, blocked, traj, s)
        IR = z.size
        OR = (synth.MODE_CS[mode] - 1) * synth.ODE_SIZES[name][0]
        lam = synth.make_multipliers(OR, seed=seed + 100 + s)
        zs = [D2.var(mp.mpf(float(v)), i, IR) for i, v in enumerate(z)]
        d = defect_value(name, mode, blocked, zs, MP)
        assert len(d) == OR
        fx = np.array([float(e.v) for e in d])
        jx = np.array([[float(e.g[i]) for i in range(IR)] for e in d])
        gacc = sum((mp.mpf(float(lam[k])) * d[k].g for k in range(OR)), np.full(IR, mp.mpf(0), dtype=object))
        hacc = sum((mp.mpf(float(lam[k])) * d[k].h for k in range(OR)), np.full((IR, IR), mp.mpf(0), dtype=object))
        xs.append(z), ls.append(lam), fxs.append(fx), jxs.append(jx)
        gxs.append(np.array([float(v) for v in gacc]))
        hxs.append(np.array([[float(v) for v in row] for row in hacc]))
    return dict(x=np.array(xs), lam=np.array(ls), fx=np.array(fxs), jx=np.array(jxs), gx=np.array(gxs),
                hx=np.array(hxs))


CASES = [
    # (ode, mode, blocked, mesh segments, which segments, seed)
    ("brachistochrone", "LGL3", False, 40, [0, 17, 39], 101),
    ("reentry", "LGL3", False, 64, [0, 31, 63], 102),
    ("reentry", "LGL5", False, 64, [0, 31, 63], 103),
    ("reentry", "LGL7", False, 64, [0, 31, 63], 104),
    ("reentry", "Trapezoidal", False, 64, [0, 31, 63], 105),
    ("twobody_lt", "LGL5", False, 75, [0, 40, 74], 106),
    ("twobody_lt", "LGL5", True, 75, [0, 40, 74], 107),
    ("twobody_lt", "Trapezoidal", True, 75, [3], 108),
    ("betts_lowthrust", "LGL5", False, 100, [0, 50, 99], 109),
    ("betts_lowthrust", "LGL7", True, 100, [7], 110),
    ("synthetic32", "LGL7", False, 10, [4], 111),
]


def case_file(name, mode, blocked):
    return os.path.join(HERE, f"{name}_{mode}{'_blocked' if blocked else ''}.npz")


if __name__ == "__main__":
    only = sys.argv[1:] or None
    for (name, mode, blocked, nm, segs, seed) in CASES:
        if only and name not in only:
            continue
        out = golden_case(name, mode, blocked, nm, segs, seed)
        np.savez_compressed(case_file(name, mode, blocked), ode=name, mode=mode, blocked=blocked, seed=seed,
                            mesh_segments=nm, segments=np.array(segs), **out)
        print("wrote", os.path.basename(case_file(name, mode, blocked)), out["hx"].shape, flush=True)
