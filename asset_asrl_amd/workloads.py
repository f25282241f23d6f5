"""The dynamics of the BASELINE.json configurations, written in this package's DSL -- WORKLOAD DEFINITIONS, not product
code: they are what a user of the reference writes in a script, restated here because bench.py, the tests and the kernels
compiled into libasset_hip.so (build.py: one translation unit per (ODE, transcription, control mode)) need them by name.
The equations are the reference's example scripts':

* ``Brachistochrone``   examples/Brachistochrone.py:15-33
* ``LTModel``           examples/BettsLowThrust.py:212-400  (MEE + J2..J4 zonal gravity + thrust)
* ``ShuttleReentry``    examples/Reentry.py:30-97
* ``TwoBody``           examples/MultiSpacecraftOptimization.py:17-34

The DSL itself (``ODEArguments``, ``ODEBase``) and the synthetic 32-state ODE this build defines are in ``ode.py``.
"""
from __future__ import annotations

import numpy as np

from . import vf
from .ode import ODEArguments, ODEBase, Synthetic32

# =========================================================================== library

class Brachistochrone(ODEBase):
    def __init__(self, g: float = 9.81):
        XtU = ODEArguments(3, 1)
        x, y, v = XtU.XVec().tolist()
        theta = XtU.UVar(0)
        xdot = vf.sin(theta) * v
        ydot = -1.0 * vf.cos(theta) * v
        vdot = g * vf.cos(theta)
        super().__init__(vf.stack([xdot, ydot, vdot]), 3, 1, name="brachistochrone")


class ShuttleReentry(ODEBase):
    """Non-dimensionalised shuttle reentry dynamics (5 states, 2 controls)."""

    def __init__(self):
        g0 = 32.2
        W = 203000.0
        Lstar = 100000.0
        Tstar = 60.0
        Mstar = W / g0
        Vstar = Lstar / Tstar
        Mustar = (Lstar ** 3) / (Tstar ** 2)
        Rhostar = Mstar / (Lstar ** 3)
        Re = 20902900.0 / Lstar
        S = 2690.0 / (Lstar ** 2)
        m = (W / g0) / Mstar
        mu = 0.140765e17 / Mustar
        rho0 = 0.002378 / Rhostar
        h_ref = 23800.0 / Lstar
        a0, a1 = -0.20704, 0.029244
        b0, b1, b2 = 0.07854, -0.61592e-2, 0.621408e-3
        del Vstar

        XtU = ODEArguments(5, 2)
        h, theta, v, gamma, psi = XtU.XVec().tolist()
        alpha, beta = XtU.UVec().tolist()

        alphadeg = (180.0 / np.pi) * alpha
        CL = a0 + a1 * alphadeg
        CD = b0 + b1 * alphadeg + b2 * (alphadeg ** 2)
        rho = rho0 * vf.exp(-h / h_ref)
        r = h + Re

        L = 0.5 * CL * S * rho * (v ** 2)
        D = 0.5 * CD * S * rho * (v ** 2)
        g = mu / (r ** 2)

        sgam, cgam = vf.sin(gamma), vf.cos(gamma)
        sbet, cbet = vf.sin(beta), vf.cos(beta)
        spsi, cpsi = vf.sin(psi), vf.cos(psi)
        tantheta = vf.tan(theta)

        hdot = v * sgam
        thetadot = (v / r) * cgam * cpsi
        vdot = -D / m - g * sgam
        gammadot = (L / (m * v)) * cbet + cgam * (v / r - g / v)
        psidot = L * sbet / (m * v * cgam) + (v / r) * cgam * spsi * tantheta
        super().__init__(vf.stack([hdot, thetadot, vdot, gammadot, psidot]), 5, 2, name="reentry")


class TwoBody(ODEBase):
    def __init__(self, P1mu: float = 1.0, ltacc=0.01):
        Xvars, Uvars = 6, (3 if ltacc is not False else 0)
        args = ODEArguments(Xvars, Uvars)
        r = args.head3()
        v = args.segment3(3)
        g = r.normalized_power3() * (-P1mu)
        acc = g + args.tail3() * ltacc if Uvars else g
        super().__init__(vf.stack([v, acc]), Xvars, Uvars, name="twobody_lt" if Uvars else "twobody")


# ---- Betts low-thrust MEE model ----------------------------------------------------------------

def _betts_constants():
    g0 = 32.174
    W = 1.0
    mu_e = 1.407645794e16
    Lstar = 20925662.73
    Tstar = Lstar / np.sqrt(mu_e / Lstar)
    Mstar = W / g0
    Fstar = Mstar * Lstar / (Tstar ** 2)
    Astar = Lstar / (Tstar ** 2)
    Mustar = (Lstar ** 3) / (Tstar ** 2)
    return dict(
        Re=20925662.73 / Lstar, mu=mu_e / Mustar, Thrust=4.446618e-3 / Fstar, Isp=450.0 / Tstar,
        gs=g0 / Astar, J2=1082.639e-6, J3=-2.565e-6, J4=-1.608e-6)


def RTNBasisFunc():
    R, V = vf.Arguments(6).tolist([(0, 3), (3, 3)])
    Rhat = R.normalized()
    Nhat = R.cross(V).normalized()
    That = Nhat.cross(R).normalized()
    return vf.stack(Rhat, That, Nhat)


def MEECartFunc(mu):
    X = vf.Arguments(6)
    p, f, g, h, k, L = X.tolist()
    sinL, cosL = vf.sin(L), vf.cos(L)
    sqp = vf.sqrt(mu / p)
    w = 1 + f * cosL + g * sinL
    s2 = 1 + h ** 2 + k ** 2
    a2 = h ** 2 - k ** 2
    r = p / w
    r_s2 = r / s2
    subs2 = 1.0 / s2
    R = r_s2 * vf.stack([cosL + a2 * cosL + 2. * h * k * sinL,
                         sinL - a2 * sinL + 2. * h * k * cosL,
                         2.0 * (h * sinL - k * cosL)])
    V = -subs2 * sqp * vf.stack([sinL + a2 * sinL - 2. * h * k * cosL + g - 2. * f * h * k + a2 * g,
                                 -cosL + a2 * cosL + 2. * h * k * sinL - f + 2. * g * h * k + a2 * f,
                                 -2.0 * (h * cosL + k * sinL + f * h + g * k)])
    return vf.stack([R, V])


def ZonalGrav(mu, Re, J2, J3, J4):
    X = vf.Arguments(6)
    R, V = X.tolist([(0, 3), (3, 3)])
    r = R.norm()
    Ir = R.normalized()
    North = np.array([0, 0, 1.0])
    In = (North - Ir * (Ir.dot(North))).normalized()
    sphi = Ir[2]
    cphi = vf.sqrt(1 - sphi ** 2)
    P2 = 0.5 * (3.0 * (sphi ** 2) - 1.0)
    P3 = 0.5 * (5.0 * (sphi ** 3) - 3 * sphi)
    P4 = (35 / 8) * (sphi ** 4) - (30 / 8) * (sphi ** 2) + 3 / 8
    D2 = 3 * sphi
    D3 = 0.5 * (15.0 * (sphi ** 2) - 3.0)
    D4 = (35 / 2) * (sphi ** 3) - (30 / 4) * (sphi)
    Js, Ps, Ds = [J2, J3, J4], [P2, P3, P4], [D2, D3, D4]
    grs, gns = [], []
    for k in range(2, 5):
        gns.append(Ds[k - 2] * Js[k - 2] * ((Re / r) ** k))
        grs.append(((k + 1) * Ps[k - 2] * Js[k - 2]) * ((Re / r) ** k))
    gn = vf.sum(gns) * cphi
    gr = vf.sum(grs)
    Gcart = (gn * In - gr * Ir) * (-mu / R.squared_norm())
    M = vf.RowMatrix(RTNBasisFunc(), 3, 3)
    return M * Gcart


def MEEDynamics2(mu):
    X = vf.Arguments(9)
    p, f, g, h, k, L, ur, ut, un = X.tolist()
    sinL, cosL = vf.sin(L), vf.cos(L)
    w = 1. + f * cosL + g * sinL
    Xtmp = vf.stack(X, sinL, cosL, w)

    X2 = vf.Arguments(12)
    p, f, g, h, k, L, ur, ut, un, sinL, cosL, w = X2.tolist()
    hk = X2.segment2(3)
    sqp = vf.sqrt(p) / np.sqrt(mu)
    s2 = 1. + hk.squared_norm()
    pdot = 2. * (p / w) * ut
    fdot = vf.sum([ur * sinL, ((w + 1) * cosL + f) * (ut / w), -(h * sinL - k * cosL) * (g * un / w)])
    gdot = vf.sum([-ur * cosL, ((w + 1) * sinL + g) * (ut / w), (h * sinL - k * cosL) * (f * un / w)])
    hkdot = vf.stack([cosL, sinL]) * ((s2 * un / w) / 2.0)
    Ldot = mu * (w / p) * (w / p) + (1.0 / w) * (h * sinL - k * cosL) * un
    return (vf.stack([pdot, fdot, gdot, hkdot, Ldot]) * sqp)(Xtmp)


class LTModel(ODEBase):
    """Betts low-thrust orbit transfer: 6 MEEs + weight, RTN thrust direction, throttle parameter."""

    def __init__(self, mu=None, T=None, gs=None, Isp=None, Re=None, J2=None):
        c = _betts_constants()
        mu = c["mu"] if mu is None else mu
        T = c["Thrust"] if T is None else T
        gs = c["gs"] if gs is None else gs
        Isp = c["Isp"] if Isp is None else Isp
        Re = c["Re"] if Re is None else Re
        J2 = c["J2"] if J2 is None else J2

        XtUP = ODEArguments(7, 3, 1)
        MEEs = XtUP.XVec().head(6)
        ww = XtUP.XVar(6)
        U = XtUP.UVec().head3().normalized()
        tau = XtUP.PVar(0)
        wwdot = -T * (1 + .01 * tau) / (Isp)
        acc_T = gs * T * (1 + .01 * tau) * U / ww
        acc_J2 = ZonalGrav(mu, Re, J2, c["J3"], c["J4"])(MEECartFunc(mu))(MEEs)
        acc = acc_T + acc_J2
        Xdot = MEEDynamics2(mu).eval(vf.stack(MEEs, acc))
        super().__init__(vf.stack([Xdot, wwdot]), 7, 3, 1, name="betts_lowthrust")




# name -> class of every ODE compiled into libasset_hip.so (build.py)
ODE_LIBRARY = {
    "brachistochrone": Brachistochrone,
    "reentry": ShuttleReentry,
    "twobody_lt": TwoBody,
    "betts_lowthrust": LTModel,
    "synthetic32": Synthetic32,
}
