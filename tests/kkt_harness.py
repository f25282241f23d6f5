"""TEST HARNESS, not a product: a small primal-dual interior-point Newton loop that consumes the evalKKT / evalOCC
results of a sparse assembly (the oracle's FullNlp here, the product's KktAssembly on the GPU box) and solves the shuttle
re-entry problem of the reference's own full-problem test, so that the path can be checked against a number the REFERENCE
holds: /root/reference/asset_asrl/test/test_FullProblems/test_Reentry.py:116-127 (objective -0.5958800738629952 +- 1e-2 for
LGL3/5/7/Trapezoidal x {HighestOrderSpline, BlockConstant}; the problem set-up is :130-175).  PSIOPT itself stays out of
scope; the loop is deliberately minimal (filter line search, inertia-free curvature test, monotone barrier) -- what has grown since
round 3 is the list of PROBLEMS stated for it (every full-problem test of the reference with a known answer, below) and the thin
wrappers they need (slack rows, link rows, variable scaling), not the solver.

The linear parts of the problem (boundary values, variable bounds, the upper bound on the final time, the objective
-(theta_f - theta_0)) are handled here as fixed variables, bounds and a constant cost vector; every NONLINEAR function the
phase registers -- defects, mesh spacing, control splines -- comes from the assembly under test: constraint values, the
Jacobian (off-diagonal block of the upper-triangular KKT CSR) and the Lagrangian Hessian sum_k lam_k grad^2 c_k (its
primal block).

A second problem of the same kind (round 5): the cart-pole swing-up of test_FullProblems/test_CartPole.py:11-95, objective
58.83219229674185 +- 0.1 -- there the cost is NONLINEAR (the integral of u^2: an LGLIntegral objective the assembly evaluates), so
the loop also takes the objective's value and gradient from the provider (`objective`, `objective_gradient`); its Hessian is in the
primal block already (ObjScale = 1)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# ---------------------------------------------------------------------------------------------- the problem
def heating_bound():
    """(q(h, v, alpha) - Qlimit) / Qlimit, the function test_Reentry.py:99-109,206 adds with addUpperFuncBound("Path", QFunc(),
    [0, 2, 6], Qlimit, 1 / Qlimit), in the product's DSL (the oracle holds it as `reentry_heating`, oracle/odes.h)."""
    from asset_asrl_amd import vf
    g0, W, Lstar, Tstar = 32.2, 203000.0, 100000.0, 60.0
    Mstar, Vstar = W / g0, Lstar / Tstar
    Rhostar = Mstar / Lstar ** 3
    rho0, h_ref = 0.002378 / Rhostar, 23800.0 / Lstar
    c0, c1, c2, c3, Qlimit = 1.0672181, -0.19213774e-1, 0.21286289e-3, -0.10117e-5, 70.0
    h, v, alpha = vf.Arguments(3).tolist()
    alphadeg = (180.0 / np.pi) * alpha
    rhodim = (rho0 * Rhostar) * vf.exp(-1.0 * h / h_ref)
    qr = 17700.0 * vf.sqrt(rhodim) * ((0.0001 * (v * Vstar)) ** 3.07)
    qa = c0 + c1 * alphadeg + c2 * (alphadeg * alphadeg) + c3 * (alphadeg * alphadeg * alphadeg)
    return vf.stack([(qa * qr - Qlimit) * (1.0 / Qlimit)])


def reentry_dimensional_ode():
    """The shuttle re-entry dynamics in ENGLISH UNITS (feet, seconds, slugs), as the reference's AutoScaling test states them
    (asset_asrl/test/test_AutoScaling/test_Reentry.py:14-103) -- with Phase.setUnits(h = 1e5 ft, v = 1e5 ft / min, t = 1 min) and
    AutoScaling the phase evaluates IOScaled(this), which is mathematically the non-dimensional `reentry` of the library and of
    the oracle."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase
    g0, W, Re, S, mu, rho0, h_ref = 32.2, 203000.0, 20902900.0, 2690.0, 0.140765e17, 0.002378, 23800.0
    m = W / g0
    a0, a1, b0, b1, b2 = -0.20704, 0.029244, 0.07854, -0.61592e-2, 0.621408e-3

    class ShuttleReentryFeet(ODEBase):
        def __init__(self):
            a = ODEArguments(5, 2)
            h, theta, v, gamma, psi = a.XVec().tolist()
            alpha, beta = a.UVar(0), a.UVar(1)
            alphadeg = (180.0 / np.pi) * alpha
            CL = a0 + a1 * alphadeg
            CD = b0 + b1 * alphadeg + b2 * (alphadeg * alphadeg)
            rho = rho0 * vf.exp(-1.0 * h / h_ref)
            r = h + Re
            L = 0.5 * CL * S * rho * (v * v)
            D = 0.5 * CD * S * rho * (v * v)
            g = mu / (r * r)
            sgam, cgam = vf.sin(gamma), vf.cos(gamma)
            rates = [v * sgam, (v / r) * cgam * vf.cos(psi), -1.0 * D / m - g * sgam,
                     (L / (m * v)) * vf.cos(beta) + cgam * (v / r - g / v),
                     L * vf.sin(beta) / (m * v * cgam) + (v / r) * cgam * vf.sin(psi) * vf.tan(theta)]
            super().__init__(vf.stack(rates), 5, 2, 0, name="reentry_feet")

    return ShuttleReentryFeet()


def heating_bound_dimensional():
    """(q(h [ft], v [ft/s], alpha) - Qlimit) / Qlimit: QFunc of test_AutoScaling/test_Reentry.py:105-114 with the bound's scale."""
    from asset_asrl_amd import vf
    rho0, h_ref = 0.002378, 23800.0
    c0, c1, c2, c3, Qlimit = 1.0672181, -0.19213774e-1, 0.21286289e-3, -0.10117e-5, 70.0
    h, v, alpha = vf.Arguments(3).tolist()
    alphadeg = (180.0 / np.pi) * alpha
    qr = 17700.0 * vf.sqrt(rho0 * vf.exp(-1.0 * h / h_ref)) * ((0.0001 * v) ** 3.07)
    qa = c0 + c1 * alphadeg + c2 * (alphadeg * alphadeg) + c3 * (alphadeg * alphadeg * alphadeg)
    return vf.stack([(qa * qr - Qlimit) * (1.0 / Qlimit)])


def reentry_problem(mode: str, control: str, nseg: int = 64, heating: bool = False, autoscaled: bool = False, phase=None):
    """-> dict(phase, ix, x0, lb, ub, cost, V, Cx, entries, n_equal); constants of test_Reentry.py:14-47,130-175.  heating: the
    heating-rate bound at every state; registered with the assembly as an EQUALITY whose rows (`slack_rows`) the harness turns
    into  g(x) + s = 0, s >= 0  (SlackRows below)."""
    from asset_asrl_amd.ode import ShuttleReentry
    Lstar, Tstar = 100000.0, 60.0
    Vstar = Lstar / Tstar
    tmax, Re = 2500 / Tstar, 20902900 / Lstar
    tf = 2000 / Tstar
    ht0, htf, vt0, vtf = 260000 / Lstar, 80000 / Lstar, 25600 / Vstar, 2500 / Vstar
    thetaf = (vt0 * tf + 0.5 * (vtf - vt0) * tf) / Re
    g0, gf, psi0 = np.deg2rad(-1.0), np.deg2rad(-5.0), np.deg2rad(90.0)
    ts = np.linspace(0, tf, 200)
    s = ts / tf
    traj = np.column_stack([ht0 * (1 - s) + htf * s, thetaf * s, vt0 * (1 - s) + vtf * s, g0 * (1 - s) + gf * s,
                            np.full_like(s, psi0), ts, 0 * s, 0 * s])
    if phase is not None:                                        # (the adaptive mesh loop: the same phase on its new mesh)
        ph = phase
    elif autoscaled:   # test_AutoScaling/test_Reentry.py:166-175: the problem in feet and seconds, the phase scales it (the solver's
        units = np.array([Lstar, 1.0, Vstar, 1.0, 1.0, Tstar, 1.0, 1.0])   # variables, bounds and cost below are in scaled units)
        ph = reentry_dimensional_ode().phase(mode, traj * units, nseg)
        ph.setUnits(units)
        ph.setAutoScaling(True)
    else:
        ph = ShuttleReentry().phase(mode, traj, nseg)
    if phase is None:
        ph.setControlMode(control)
        if heating:
            ph.addEqualCon("Path", heating_bound_dimensional() if autoscaled else heating_bound(), [0, 2, 6])
    ix, (V, Cx), entries, n_equal, _ = ph.layout()
    x0 = ix.makeSolverInput(ph.ActiveTraj / ph.XtUPUnits if autoscaled else ph.ActiveTraj)
    n, S, D = x0.size, ix.numStates, ix.numDefects
    lb, ub, cost = np.full(n, -np.inf), np.full(n, np.inf), np.zeros(n)
    d89, d90, d1 = np.deg2rad(89.0), np.deg2rad(90.0), np.deg2rad(1.0)
    for k in range(S):                                           # addLUVarBounds("Path", [1, 3], -89deg, 89deg)
        for v in (1, 3):
            lb[ix.getXTUVarLoc(v, k)], ub[ix.getXTUVarLoc(v, k)] = -d89, d89
    where_u = [(0, d) for d in range(D)] if ix.BlockedControls else [(k, None) for k in range(S)]
    for k, d in where_u:                                         # controls: alpha in +-90deg, beta in [-90deg, 1deg]
        la, lbeta = ix.getXTUVarLoc(6, k, d), ix.getXTUVarLoc(7, k, d)
        lb[la], ub[la], lb[lbeta], ub[lbeta] = -d90, d90, -d90, d1
    for v in range(6):                                           # addBoundaryValue("Front", range(0, 6), TrajIG[0][0:6])
        lb[ix.getXTUVarLoc(v, 0)] = ub[ix.getXTUVarLoc(v, 0)] = traj[0, v]
    for v, val in ((0, htf), (2, vtf), (3, gf)):                 # addBoundaryValue("Back", [0, 2, 3], ...)
        lb[ix.getXTUVarLoc(v, S - 1)] = ub[ix.getXTUVarLoc(v, S - 1)] = val
    ub[ix.getXTUVarLoc(5, S - 1)] = tmax                         # addUpperDeltaTimeBound(tmax) with t_0 fixed at 0
    cost[ix.getXTUVarLoc(1, S - 1)] = -1.0                       # addDeltaVarObjective(1, -1.0); theta_0 is fixed at 0
    slack_rows = np.concatenate([e[5].ravel() for e in entries if e[0] == "equality"] + [np.zeros(0, dtype=np.int32)])
    return dict(phase=ph, ix=ix, x0=x0, lb=lb, ub=ub, cost=cost, V=V, Cx=Cx, entries=entries, n_equal=n_equal,
                slack_rows=slack_rows)


def brachistochrone_exact(x=10.0, drop=5.0, g=9.81):
    """The cycloid through the start: x = R (phi - sin phi), drop = R (1 - cos phi); descent time phi sqrt(R / g)."""
    from scipy.optimize import brentq
    phi = brentq(lambda p: (p - np.sin(p)) / (1 - np.cos(p)) - x / drop, 0.1, 2 * np.pi - 0.1, xtol=1e-15)
    return float(phi * np.sqrt(drop / (1 - np.cos(phi)) / g))


def brachistochrone_problem(mode: str, control: str, nseg: int = 32):
    """examples/Brachistochrone.py:36-67 (BASELINE.json configs[0], the reference's own CPU-runnable case): a bead from (0, 10) at rest
    to (10, 5) under g = 9.81, theta in [-0.1, 2], minimum time.  The library ODE `brachistochrone`; linear cost t_f."""
    from asset_asrl_amd.ode import Brachistochrone
    g, theta0, tf = 9.81, 1.0, 1.0
    traj = np.array([[10 * t / tf, 10 - 5 * t / tf, g * t * np.cos(theta0), t, theta0] for t in np.linspace(0, tf, 100)])
    ph = Brachistochrone(g).phase(mode, traj, nseg)
    ph.setControlMode(control)
    ix, (V, Cx), entries, n_equal, _ = ph.layout()
    x0 = ix.makeSolverInput(ph.ActiveTraj)
    n, S, D = x0.size, ix.numStates, ix.numDefects
    lb, ub, cost = np.full(n, -np.inf), np.full(n, np.inf), np.zeros(n)
    for k, dd in ([(0, dd) for dd in range(D)] if ix.BlockedControls else [(k, None) for k in range(S)]):
        lu = ix.getXTUVarLoc(4, k, dd)                           # addLUVarBound("Path", 4, -0.1, 2.00)
        lb[lu], ub[lu] = -0.1, 2.0
    for v, val in enumerate([0.0, 10.0, 0.0, 0.0]):              # addBoundaryValue("Front", range(0, 4), [x0, y0, v0, 0])
        lb[ix.getXTUVarLoc(v, 0)] = ub[ix.getXTUVarLoc(v, 0)] = val
    for v, val in ((0, 10.0), (1, 5.0)):                         # addBoundaryValue("Back", [0, 1], [xf, yf])
        lb[ix.getXTUVarLoc(v, S - 1)] = ub[ix.getXTUVarLoc(v, S - 1)] = val
    cost[ix.getXTUVarLoc(3, S - 1)] = 1.0                        # addDeltaTimeObjective(1.0); t_0 is fixed at 0
    return dict(phase=ph, ix=ix, x0=x0, lb=lb, ub=ub, cost=cost, V=V, Cx=Cx, entries=entries, n_equal=n_equal,
                slack_rows=np.zeros(0, dtype=np.int32), ode_name="brachistochrone", usize=1)


def cartpole_ode():
    """The reference test's dynamics (test_CartPole.py:11-32; l, m1, m2, g of :43-46) in the product's DSL; the oracle holds the
    same right-hand side as `cartpole` (oracle/odes.h), differentiated independently by AD2."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase
    l, m1, m2, g = 0.5, 1.0, 0.3, 9.81

    class CartPole(ODEBase):
        def __init__(self):
            a = ODEArguments(4, 1)
            q1, q2, q1d, q2d = a.XVec().tolist()
            u = a.UVar(0)
            s2, c2 = vf.sin(q2), vf.cos(q2)
            den = m1 + m2 * (1.0 - c2 * c2)
            q1dd = (l * m2 * s2 * (q2d * q2d) + u + m2 * g * c2 * s2) / den
            q2dd = -1.0 * (l * m2 * c2 * s2 * (q2d * q2d) + u * c2 + (m1 * g + m2 * g) * s2) / (l * den)
            super().__init__(vf.stack([q1d, q2d, q1dd, q2dd]), 4, 1, 0, name="cartpole")

    return CartPole()


def cartpole_problem(mode: str, control: str, nseg: int, phase=None):
    """test_CartPole.py:41-70: swing the pole up in tf = 2 while the cart moves d = 1, |u| <= 20, |q1| <= 2, minimise int u^2 dt.
    The integrand is taken over the node values (q1, u) -- u^2 with an unused first input: the oracle's function records have at
    least two inputs -- which is the same objective."""
    from asset_asrl_amd import vf
    umax, dmax, tf, d = 20.0, 2.0, 2.0, 1.0
    ts = np.linspace(0, tf, 100)
    traj = np.array([[d * t / tf, np.pi * t / tf, 0.0, 0.0, t, 0.0] for t in ts])
    ph = phase                                                   # (the adaptive mesh loop: the same phase on its new mesh)
    if ph is None:
        ph = cartpole_ode().phase(mode, traj, nseg)
        ph.setControlMode(control)
        a = vf.Arguments(2)
        ph.addIntegralObjective(a.coeff(1) * a.coeff(1), [0, 5])     # addIntegralObjective(Args(1)[0]**2, [5])
    ix, (V, Cx), entries, n_equal, _ = ph.layout()
    x0 = ix.makeSolverInput(ph.ActiveTraj)
    n, S, D = x0.size, ix.numStates, ix.numDefects
    lb, ub, cost = np.full(n, -np.inf), np.full(n, np.inf), np.zeros(n)
    for k in range(S):                                           # addLUVarBound("Path", 0, -dmax, dmax)
        lb[ix.getXTUVarLoc(0, k)], ub[ix.getXTUVarLoc(0, k)] = -dmax, dmax
    for k, dd in ([(0, dd) for dd in range(D)] if ix.BlockedControls else [(k, None) for k in range(S)]):
        lu = ix.getXTUVarLoc(5, k, dd)                           # addLUVarBound("Path", 5, -umax, umax)
        lb[lu], ub[lu] = -umax, umax
    for v, val in enumerate([0.0, 0.0, 0.0, 0.0, 0.0]):          # addBoundaryValue("Front", range(0, 5), ...)
        lb[ix.getXTUVarLoc(v, 0)] = ub[ix.getXTUVarLoc(v, 0)] = val
    for v, val in enumerate([d, np.pi, 0.0, 0.0, tf]):           # addBoundaryValue("Back", range(0, 5), ...)
        lb[ix.getXTUVarLoc(v, S - 1)] = ub[ix.getXTUVarLoc(v, S - 1)] = val
    return dict(phase=ph, ix=ix, x0=x0, lb=lb, ub=ub, cost=cost, V=V, Cx=Cx, entries=entries, n_equal=n_equal,
                slack_rows=np.zeros(0, dtype=np.int32), ode_name="cartpole", integrands={"obj0": ("integrand_usq", 2)},
                usize=1)


def freeflyingrobot_ode():
    """test_FreeFlyingRobot.py:14-35 (alpha = beta = 0.2) in the product's DSL; the oracle holds it as `freeflyingrobot`."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase
    alpha, beta = 0.2, 0.2

    class FreeFlyingRobot(ODEBase):
        def __init__(self):
            a = ODEArguments(6, 4)
            x, y, vx, vy, theta, omega = a.XVec().tolist()
            u = [a.UVar(k) for k in range(4)]
            vscale = u[0] - u[1] + u[2] - u[3]
            rates = [vx, vy, vf.cos(theta) * vscale, vf.sin(theta) * vscale, omega,
                     alpha * u[0] - alpha * u[1] - beta * u[2] + beta * u[3]]
            super().__init__(vf.stack(rates), 6, 4, 0, name="freeflyingrobot")

    return FreeFlyingRobot()


def freeflyingrobot_problem(mode: str, control: str, nseg: int = 256):
    """test_FreeFlyingRobot.py:50-76: from (-10, -10) at rest, heading pi/2, to the origin at rest, heading 0, in tf = 12; four
    thrusters in [0, 1]; minimise the integral of their sum (bang-bang)."""
    from asset_asrl_amd import vf
    tf = 12.0
    X0, XF = np.array([-10, -10, 0, 0, np.pi / 2, 0, 0.0]), np.array([0, 0, 0, 0, 0, 0, tf])
    ts = np.linspace(0, tf, 100)
    traj = np.array([np.concatenate([X0 + (t / tf) * (XF - X0), 0.5 * np.ones(4)]) for t in ts])
    ph = freeflyingrobot_ode().phase(mode, traj, nseg)
    ph.setControlMode(control)
    a = vf.Arguments(4)
    ph.addIntegralObjective(a.coeff(0) + a.coeff(1) + a.coeff(2) + a.coeff(3), [7, 8, 9, 10])
    ix, (V, Cx), entries, n_equal, _ = ph.layout()
    x0 = ix.makeSolverInput(ph.ActiveTraj)
    n, S, D = x0.size, ix.numStates, ix.numDefects
    lb, ub, cost = np.full(n, -np.inf), np.full(n, np.inf), np.zeros(n)
    for k, dd in ([(0, dd) for dd in range(D)] if ix.BlockedControls else [(k, None) for k in range(S)]):
        for v in range(7, 11):                                   # addLUVarBounds("Path", range(7, 11), 0.0, 1.0)
            lu = ix.getXTUVarLoc(v, k, dd)
            lb[lu], ub[lu] = 0.0, 1.0
    for v in range(7):                                           # addBoundaryValue("Front" / "Back", range(0, 7), ...)
        lb[ix.getXTUVarLoc(v, 0)] = ub[ix.getXTUVarLoc(v, 0)] = X0[v]
        lb[ix.getXTUVarLoc(v, S - 1)] = ub[ix.getXTUVarLoc(v, S - 1)] = XF[v]
    return dict(phase=ph, ix=ix, x0=x0, lb=lb, ub=ub, cost=cost, V=V, Cx=Cx, entries=entries, n_equal=n_equal,
                slack_rows=np.zeros(0, dtype=np.int32), ode_name="freeflyingrobot", integrands={"obj0": ("integrand_sum4", 4)},
                usize=4)


def cannon_constants():
    """test_MultiPhaseCannon.py:16-34"""
    g0, Lstar, Tstar, Mstar = 9.81, 1000.0, 60.0, 10.0
    Astar, Vstar, Rhostar = Lstar / Tstar ** 2, Lstar / Tstar, Mstar / Lstar ** 3
    Estar = Mstar * Vstar ** 2
    return dict(Lstar=Lstar, Tstar=Tstar, CD=0.5, RhoAir=1.225 / Rhostar, RhoIron=7870.0 / Rhostar, h_scale=8.44e3 / Lstar,
                E0=400000.0 / Estar, g=g0 / Astar)


def cannon_ode():
    """test_MultiPhaseCannon.py:43-72 in the product's DSL -- an ODE with a PARAMETER (the ball's radius); the oracle holds it as
    `cannon` (oracle/odes.h)."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase
    k = cannon_constants()

    class Cannon(ODEBase):
        def __init__(self):
            a = ODEArguments(4, 0, 1)
            v, gamma, h, r = a.XVec().tolist()
            rad = a.PVar(0)
            S = np.pi * (rad * rad)
            M = (4.0 / 3.0) * (np.pi * k["RhoIron"]) * (rad * rad * rad)
            rho = k["RhoAir"] * vf.exp(-1.0 * h / k["h_scale"])
            D = (0.5 * k["CD"]) * rho * (v * v) * S
            rates = [-1.0 * D / M - k["g"] * vf.sin(gamma), -1.0 * k["g"] * vf.cos(gamma) / v, v * vf.sin(gamma), v * vf.cos(gamma)]
            super().__init__(vf.stack(rates), 4, 0, 1, name="cannon")

    return Cannon()


def cannon_energy():
    """(E(v, rad) - E0) / 100 <= 0 at the muzzle (EFunc() * .01, test_MultiPhaseCannon.py:75-79,131); the oracle's `cannon_energy`."""
    from asset_asrl_amd import vf
    k = cannon_constants()
    v, rad = vf.Arguments(2).tolist()
    M = (4.0 / 3.0) * (np.pi * k["RhoIron"]) * (rad * rad * rad)
    return vf.stack([(0.5 * M * (v * v) - k["E0"]) * 0.01])


def _cannon_rhs(y):
    k = cannon_constants()
    v, gamma, h, r, t, rad = y
    M = (4.0 / 3.0) * np.pi * k["RhoIron"] * rad ** 3
    D = 0.5 * k["CD"] * k["RhoAir"] * np.exp(-h / k["h_scale"]) * v * v * np.pi * rad * rad
    return np.array([-D / M - k["g"] * np.sin(gamma), -k["g"] * np.cos(gamma) / v, v * np.sin(gamma), v * np.cos(gamma), 1.0, 0.0])


def _rk4_until(y0, tmax, stop, dt=1e-3):
    """the initial guess only (the reference integrates it with its adaptive integrator, :107-116): classical RK4 until `stop`"""
    out, y = [np.array(y0, dtype=float)], np.array(y0, dtype=float)
    while y[4] - y0[4] < tmax:
        k1 = _cannon_rhs(y); k2 = _cannon_rhs(y + 0.5 * dt * k1); k3 = _cannon_rhs(y + 0.5 * dt * k2); k4 = _cannon_rhs(y + dt * k3)
        y = y + (dt / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
        out.append(y.copy())
        if stop(y):
            break
    return np.array(out)


def cannon_problem(mode: str, nseg: int):
    """test_MultiPhaseCannon.py:89-150: TWO phases (ascent to the apex, descent to the ground) of the same ODE, linked -- the ball's
    radius is an ODE parameter of each phase, the two tied by a direct link; maximise the range for a bounded muzzle energy.
    -> a problem of two `parts` in one solver vector: the second phase's variables and equality rows start behind the first's
    (Phase.layout(Vstart, Estart): transcribe_phase(Vstart, Estart, ...), OptimalControlProblem.cpp:131-146).  The links
    (addForwardLinkEqualCon / addDirectLinkEqualCon: linear) live in the harness as `linear_rows`; the muzzle-energy bound is a
    user function of the assembly (slack row)."""
    k = cannon_constants()
    Lstar, Tstar = k["Lstar"], k["Tstar"]
    rad0, h0, r0, gamma0 = 0.1 / Lstar, 100.0 / Lstar, 0.0, np.deg2rad(45.0)
    m0 = (4.0 / 3.0) * np.pi * k["RhoIron"] * rad0 ** 3
    v0 = np.sqrt(2 * k["E0"] / m0) * 0.99
    asc = _rk4_until([v0, gamma0, h0, r0, 0.0, rad0], 60.0 / Tstar, lambda y: y[0] * np.sin(y[1]) < 0)
    des = _rk4_until(asc[-1], 30.0 / Tstar, lambda y: y[2] < 0)
    ode = cannon_ode()
    pa = ode.phase(mode, asc, nseg)
    pa.addEqualCon("Front", cannon_energy(), [0], [0])          # addInequalCon("Front", EFunc() * .01, [0], [0], []): a slack row
    pd = ode.phase(mode, des, nseg)
    ia, (Va, Ca), ea, na_eq, _ = pa.layout()
    xa = ia.makeSolverInput(pa.ActiveTraj)
    idd, (Vd, Cd), ed, nd_eq, _ = pd.layout(Vstart=xa.size, Estart=na_eq)
    xd = idd.makeSolverInput(pd.ActiveTraj)
    x0 = np.concatenate([xa, xd])
    n = x0.size
    lb, ub, cost = np.full(n, -np.inf), np.full(n, np.inf), np.zeros(n)
    Sa, Sd = ia.numStates, idd.numStates
    rad_a, rad_d = ia.var_offset + ia.ODEParamLoc0, idd.var_offset + idd.ODEParamLoc0
    lb[rad_a] = 0.01 / Lstar                                    # aphase.addLowerVarBound("ODEParams", 0, 0.01 / Lstar)
    lb[ia.getXTUVarLoc(1, 0)] = 0.0                             # aphase.addLowerVarBound("Front", 1, 0.0)
    for v, val in ((2, h0), (3, r0), (4, 0.0)):                 # aphase.addBoundaryValue("Front", [2, 3, 4], [h0, r0, 0])
        lb[ia.getXTUVarLoc(v, 0)] = ub[ia.getXTUVarLoc(v, 0)] = val
    lb[ia.getXTUVarLoc(1, Sa - 1)] = ub[ia.getXTUVarLoc(1, Sa - 1)] = 0.0      # aphase.addBoundaryValue("Back", [1], [0.0])
    lb[idd.getXTUVarLoc(2, Sd - 1)] = ub[idd.getXTUVarLoc(2, Sd - 1)] = 0.0    # dphase.addBoundaryValue("Back", [2], [0.0])
    cost[idd.getXTUVarLoc(3, Sd - 1)] = -1.0                    # dphase.addValueObjective("Back", 3, -1.0)
    rows = [(ia.getXTUVarLoc(v, Sa - 1), idd.getXTUVarLoc(v, 0)) for v in range(5)] + [(rad_a, rad_d)]   # the two links
    A = sp.csr_matrix((np.tile([1.0, -1.0], len(rows)), (np.repeat(np.arange(len(rows)), 2), np.ravel(rows))), shape=(len(rows), n))
    slack_rows = np.concatenate([e[5].ravel() for e in ea if e[0] == "equality"])
    parts = [dict(phase=pa, ix=ia, V=Va, Cx=Ca, entries=ea, ode_name="cannon", functions={"eq0": "cannon_energy"}),
             dict(phase=pd, ix=idd, V=Vd, Cx=Cd, entries=ed, ode_name="cannon", functions={})]
    var_scale = np.ones(n)
    var_scale[[rad_a, rad_d]] = rad0
    return dict(parts=parts, x0=x0, lb=lb, ub=ub, cost=cost, n_equal=na_eq + nd_eq, slack_rows=slack_rows,
                linear_rows=(A, np.zeros(len(rows))), objective_scale=Lstar, var_scale=var_scale)


class LinearRows:
    """Appends linear equality rows A x = b (the links between phases) to an assembly: multipliers behind the inner ones."""

    def __init__(self, inner, A, b):
        self.inner, self.A, self.b, self.n, self.mi = inner, sp.csr_matrix(A), np.asarray(b, dtype=float), inner.n, inner.m
        self.m = inner.m + self.A.shape[0]

    def kkt(self, x, lam):
        c, agx, W, J = self.inner.kkt(x, lam[:self.mi])
        return (np.concatenate([c, self.A @ x - self.b]), agx + self.A.T @ lam[self.mi:], W, sp.vstack([J, self.A], format="csr"))

    def con(self, x):
        return np.concatenate([self.inner.con(x), self.A @ x - self.b])


def delta3_constants():
    """test_Delta3Launch.py:16-99 (non-dimensional: Earth radius, first-stage burn time, lift-off mass)"""
    g0, Lstar, Tstar, Mstar = 9.80665, 6378145.0, 961.0, 301454.0
    Astar, Rhostar = Lstar / Tstar ** 2, Mstar / Lstar ** 3
    Mustar, Fstar = Lstar ** 3 / Tstar ** 2, Lstar / Tstar ** 2 * Mstar
    k = dict(Lstar=Lstar, Tstar=Tstar, Mstar=Mstar, Vstar=Lstar / Tstar, mu=3.986012e14 / Mustar, Re=1.0, We=7.29211585e-5 * Tstar,
             RhoAir=1.225 / Rhostar, h_scale=7200.0 / Lstar, g=g0 / Astar, CD=0.5, S=4 * np.pi / Lstar ** 2)
    TS, T1, T2 = 628500.0 / Fstar, 1083100.0 / Fstar, 110094.0 / Fstar
    IS, I1, I2 = 283.33364 / Tstar, 301.68 / Tstar, 467.21 / Tstar
    tS, t1, t2 = 75.2 / Tstar, 261.0 / Tstar, 700.0 / Tstar
    TMS, TM1, TM2, TMPay = 19290.0 / Mstar, 104380.0 / Mstar, 19300.0 / Mstar, 4164.0 / Mstar
    PMS, PM1, PM2 = 17010.0 / Mstar, 95550.0 / Mstar, 16820.0 / Mstar
    SMS, SM1 = TMS - PMS, TM1 - PM1
    k["thrust"] = [6 * TS + T1, 3 * TS + T1, T1, T2]
    k["mdot"] = [(6 * TS / IS + T1 / I1) / k["g"], (3 * TS / IS + T1 / I1) / k["g"], T1 / (k["g"] * I1), T2 / (k["g"] * I2)]
    k["tf"] = [tS, 2 * tS, t1, t1 + t2]
    m01 = 9 * TMS + TM1 + TM2 + TMPay
    mf1 = m01 - 6 * PMS - (tS / t1) * PM1
    m02 = mf1 - 6 * SMS
    mf2 = m02 - 3 * PMS - (tS / t1) * PM1
    m03 = mf2 - 3 * SMS
    mf3 = m03 - (1 - 2 * tS / t1) * PM1
    m04 = mf3 - SM1
    k["m0"], k["mf"] = [m01, m02, m03, m04], [mf1, mf2, mf3, m04 - PM2]
    return k


def delta3_ode(stage: int):
    """test_Delta3Launch.py:104-131 in the product's DSL: stage `stage`'s thrust and mass flow; the oracle's `delta3_<stage + 1>`."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase
    k = delta3_constants()
    T, mdot = k["thrust"][stage], k["mdot"][stage]

    class Delta3(ODEBase):
        def __init__(self):
            a = ODEArguments(7, 3)
            X = a.XVec()
            r, v, m = X.head3(), X.segment3(3), a.XVar(6)
            u = a.UVec().normalized()
            rho = k["RhoAir"] * vf.exp(-1.0 * (r.norm() - k["Re"]) / k["h_scale"])
            vr = v + r.cross(np.array([0.0, 0.0, k["We"]]))
            D = (-0.5 * k["CD"] * k["S"]) * rho * (vr * vr.norm())
            vdot = (-k["mu"]) * r.normalized_power3() + (T * u + D) / m
            super().__init__(vf.stack(v, vdot, 0.0 * m - mdot), 7, 3, 0, name=f"delta3_{stage + 1}")

    return Delta3()


def delta3_orbit():
    """The five insertion conditions of test_Delta3Launch.py:133-157 as smooth functions with the same zero set near the solution (no
    arccos, no branch): a - at, |e| - et, cos i - cos it, the node line's direction, the cosine of the argument of perigee; the
    oracle's `delta3_orbit`."""
    from asset_asrl_amd import vf
    k = delta3_constants()
    at, et, it, Ot, Wt = 24361140.0 / k["Lstar"], 0.7308, np.deg2rad(28.5), np.deg2rad(269.8), np.deg2rad(130.5)
    a = vf.Arguments(6)
    r, v = a.head3(), a.tail3()
    h = r.cross(v)
    e = v.cross(h) / k["mu"] - r.normalized()
    nx, ny = -1.0 * h[1], h[0]
    nn = (nx * nx + ny * ny).sqrt()
    eps = 0.5 * v.squared_norm() - k["mu"] / r.norm()
    return vf.stack([-0.5 * k["mu"] / eps - at, e.norm() - et, h[2] / h.norm() - np.cos(it),
                     (nx * np.sin(Ot) - ny * np.cos(Ot)) / nn, (nx * e[0] + ny * e[1]) / (nn * e.norm()) - np.cos(Wt)])


def _classic_to_cartesian(oe, mu):
    """(a, e, i, Omega, omega, M) -> [r, v] (the initial guess only: Astro.classic_to_cartesian, test_Delta3Launch.py:179)"""
    a, e, i, Om, w, M = oe
    E = M
    for _ in range(50):
        E -= (E - e * np.sin(E) - M) / (1 - e * np.cos(E))
    nu = 2 * np.arctan2(np.sqrt(1 + e) * np.sin(E / 2), np.sqrt(1 - e) * np.cos(E / 2))
    p = a * (1 - e * e)
    rp = p / (1 + e * np.cos(nu)) * np.array([np.cos(nu), np.sin(nu), 0.0])
    vp = np.sqrt(mu / p) * np.array([-np.sin(nu), e + np.cos(nu), 0.0])
    R3 = lambda t: np.array([[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1.0]])
    R1 = lambda t: np.array([[1.0, 0, 0], [0, np.cos(t), -np.sin(t)], [0, np.sin(t), np.cos(t)]])
    Q = R3(Om) @ R1(i) @ R3(w)
    return np.concatenate([Q @ rp, Q @ vp])


def delta3_problem(mode: str, control: str, npts: int, warm=None):
    """test_Delta3Launch.py:164-262: FOUR phases (the burns between the jettison events), each its own ODE object, linked in
    position, velocity, time and thrust direction; stage masses fixed at the front of each phase; maximise the mass at the end of
    the last; the target orbit's five conditions at its end; |u| in [0.5, 1.5] and |r| >= 0.999999 Re at every state."""
    from asset_asrl_amd import vf
    k = delta3_constants()
    at, et, Ot, Wt, istart = 24361140.0 / k["Lstar"], 0.7308, np.deg2rad(269.8), np.deg2rad(130.5), np.deg2rad(28.5)
    y0 = np.zeros(6)
    y0[:3] = np.array([np.cos(istart), 0.0, np.sin(istart)]) * k["Re"]
    y0[3:] = -np.cross(y0[:3], [0.0, 0.0, k["We"]])
    y0[3] += 0.0001 / k["Vstar"]
    yf = _classic_to_cartesian([at, et, istart, Ot, Wt, -0.05], k["mu"])
    tf, m0, mf = k["tf"], k["m0"], k["mf"]
    igs = [[], [], [], []]
    for t in np.linspace(0, tf[3], npts):
        ph = next((j for j in range(4) if t < tf[j]), None)
        if ph is None:
            continue
        X = np.zeros(11)
        X[:6] = y0 + (yf - y0) * (t / tf[3])
        t_lo = 0.0 if ph == 0 else tf[ph - 1]
        X[6] = m0[ph] + (mf[ph] - m0[ph]) * ((t - t_lo) / (tf[ph] - t_lo))
        X[7], X[8] = t, 1.0
        igs[ph].append(X)
    if warm is not None:       # (a converged solution of another transcription on the same mesh, per phase: (problem, x))
        wp, wx = warm
        igs = [part["ix"].collectSolverOutput(wx[part["ix"].var_offset:part["ix"].var_offset + part["ix"].numPhaseVars])[0]
               for part in wp["parts"]]
    a3 = vf.Arguments(3)
    nsegs = [len(g) - 1 for g in igs] if warm is None else [part["ix"].numDefects for part in warm[0]["parts"]]
    first = [np.array(g[0]) for g in igs]
    parts, xs, nvar, nrow = [], [], 0, 0
    for j in range(4):
        ph = delta3_ode(j).phase(mode, np.array(igs[j]), nsegs[j])
        ph.setControlMode(control)
        ph.addEqualCon("Path", vf.stack([a3.norm()]), [8, 9, 10])       # addLUNormBound("Path", [8, 9, 10], .5, 1.5): slack rows
        ph.addEqualCon("Path", vf.stack([a3.norm()]), [0, 1, 2])        # addLowerNormBound("Path", [0, 1, 2], Re * .999999)
        fmap = {"eq0": "norm3", "eq1": "norm3"}
        if j == 3:
            ph.addEqualCon("Back", delta3_orbit(), range(6))            # addEqualCon("Back", TargetOrbit(...), range(0, 6))
            fmap["eq2"] = "delta3_orbit"
        ix, (V, Cx), entries, neq, _ = ph.layout(Vstart=nvar, Estart=nrow)
        x = ix.makeSolverInput(ph.ActiveTraj)
        parts.append(dict(phase=ph, ix=ix, V=V, Cx=Cx, entries=entries, ode_name=f"delta3_{j + 1}", functions=fmap, usize=3))
        xs.append(x)
        nvar, nrow = nvar + x.size, nrow + neq
    x0 = np.concatenate(xs)
    n = x0.size
    lb, ub, cost = np.full(n, -np.inf), np.full(n, np.inf), np.zeros(n)
    fix = lambda loc, val: (lb.__setitem__(loc, val), ub.__setitem__(loc, val))
    ix1, ix4 = parts[0]["ix"], parts[3]["ix"]
    for v in range(8):                                                  # phase1.addBoundaryValue("Front", range(0, 8), IG1[0][0:8])
        fix(ix1.getXTUVarLoc(v, 0), first[0][v])
    for j in range(3):                                                  # addBoundaryValue("Back", [7], [tf_phase])
        fix(parts[j]["ix"].getXTUVarLoc(7, parts[j]["ix"].numStates - 1), tf[j])
    for j in range(1, 4):                                               # addBoundaryValue("Front", [6], [m0_phase])
        fix(parts[j]["ix"].getXTUVarLoc(6, 0), m0[j])
    ub[ix4.getXTUVarLoc(7, ix4.numStates - 1)] = tf[3]                  # phase4.addUpperVarBound("Back", 7, tf_phase4)
    cost[ix4.getXTUVarLoc(6, ix4.numStates - 1)] = -1.0                 # phase4.addValueObjective("Back", 6, -1.0)
    links = []                                                          # ocp.addForwardLinkEqualCon(phase1, phase4, [0..5, 7, 8, 9, 10])
    for j in range(3):
        ia, ib = parts[j]["ix"], parts[j + 1]["ix"]
        links += [(ia.getXTUVarLoc(v, ia.numStates - 1), ib.getXTUVarLoc(v, 0)) for v in (0, 1, 2, 3, 4, 5, 7, 8, 9, 10)]
    A = sp.csr_matrix((np.tile([1.0, -1.0], len(links)), (np.repeat(np.arange(len(links)), 2), np.ravel(links))), shape=(len(links), n))
    srows, slo, shi = [], [], []
    for part in parts:                                                  # g(x) + s = 0:  s in [-1.5, -0.5] / s <= -0.999999 Re
        for e in part["entries"]:
            if e[1] in ("eq0", "eq1"):
                r = e[5].ravel()
                srows.append(r)
                slo.append(np.full(r.size, -1.5 if e[1] == "eq0" else -np.inf))
                shi.append(np.full(r.size, -0.5 if e[1] == "eq0" else -0.999999 * k["Re"]))
    return dict(parts=parts, x0=x0, lb=lb, ub=ub, cost=cost, n_equal=nrow, slack_rows=np.concatenate(srows),
                slack_bounds=(np.concatenate(slo), np.concatenate(shi)), linear_rows=(A, np.zeros(len(links))),
                objective_scale=k["Mstar"])


def lq_ode():
    """x' = x / 2 + u (test_ObjScaling.py:11-21); the oracle's `lq1`."""
    from asset_asrl_amd import vf
    from asset_asrl_amd.ode import ODEArguments, ODEBase

    class LQ(ODEBase):
        def __init__(self):
            a = ODEArguments(1, 1)
            super().__init__(vf.stack([0.5 * a.XVar(0) + a.UVar(0)]), 1, 1, 0, name="lq1")

    return LQ()


def lq_problem(mode: str, control: str, nseg: int, integral_param: bool = False):
    """test_ObjScaling.py:44-141 on a fixed mesh: x(0) = 1, t in [0, 1], minimise pi * int (u^2 + x u + 1.25 x^2) dt + e * x(1).
    integral_param = False: the running cost as an integral objective (:77).  True: as an INTEGRAL PARAMETER FUNCTION (:126-128) --
    the static parameter p is made to equal the integral by one constraint row that the accumulation -p and the segment quadratures
    share (Phase.addIntegralParamFunction; ODEPhaseBase.cpp:835-889), and the cost is e * x(1) + pi * p, linear."""
    from asset_asrl_amd import vf
    traj = np.array([[1.0, t, 0.4] for t in np.linspace(0.0, 1.0, 100)])
    ph = lq_ode().phase(mode, traj, nseg)
    ph.setControlMode(control)
    a = vf.Arguments(2)
    run = a.coeff(1) * a.coeff(1) + a.coeff(0) * a.coeff(1) + 1.25 * (a.coeff(0) * a.coeff(0))
    if integral_param:
        ph.setStaticParams([0.0])
        ph.addIntegralParamFunction(run, [0, 2], accum_param=0, scale=1.0)
    else:
        ph.addIntegralObjective(run * np.pi, [0, 2])
    ix, (V, Cx), entries, n_equal, _ = ph.layout()
    x0 = ix.makeSolverInput(ph.ActiveTraj, ph.ActiveStaticParams)
    n, S = x0.size, ix.numStates
    lb, ub, cost = np.full(n, -np.inf), np.full(n, np.inf), np.zeros(n)
    lb[ix.getXTUVarLoc(0, 0)] = ub[ix.getXTUVarLoc(0, 0)] = 1.0          # addBoundaryValue("Front", [0, 1], [x0, t0])
    lb[ix.getXTUVarLoc(1, 0)] = ub[ix.getXTUVarLoc(1, 0)] = 0.0
    lb[ix.getXTUVarLoc(1, S - 1)] = ub[ix.getXTUVarLoc(1, S - 1)] = 1.0  # addBoundaryValue("Back", [1], [tf])
    cost[ix.getXTUVarLoc(0, S - 1)] = np.e                               # addValueObjective("Back", 0, vscale)
    if integral_param:
        cost[ix.var_offset + ix.StaticParamLoc0] = np.pi                 # addValueObjective("StaticParams", 0, iscale)
    return dict(phase=ph, ix=ix, x0=x0, lb=lb, ub=ub, cost=cost, V=V, Cx=Cx, entries=entries, n_equal=n_equal,
                slack_rows=np.zeros(0, dtype=np.int32), ode_name="lq1", integrands={"obj0": ("integrand_lq_pi", 2), "ipf0_int": ("integrand_lq", 2)},
                usize=1, final_state=ix.getXTUVarLoc(0, S - 1))


class ScaledVars:
    """The assembly in scaled variables x = s * xs (a harness-level diagonal scaling: the cannon ball's radius is 1e-4 in the
    problem's units, the other variables O(1) -- PSIOPT's own scaling is not part of this loop)."""

    def __init__(self, inner, scale):
        self.inner, self.s, self.n, self.m = inner, np.asarray(scale, dtype=float), inner.n, inner.m
        self.D = sp.diags(self.s)

    def kkt(self, xs, lam):
        c, agx, W, J = self.inner.kkt(self.s * xs, lam)
        return c, self.s * agx, (self.D @ W @ self.D).tocsr(), (J @ self.D).tocsr()

    def con(self, xs):
        return self.inner.con(self.s * xs)


class SlackRows:
    """Turns rows of an equality-only assembly into inequalities g(x) <= 0: variables [x ; s], rows g(x) + s = 0, s >= 0."""

    def __init__(self, inner, rows):
        self.inner, self.rows, self.n, self.m, self.ns = inner, np.asarray(rows), inner.n, inner.m, len(rows)
        self.E = sp.csr_matrix((np.ones(self.ns), (self.rows, np.arange(self.ns))), shape=(self.m, self.ns))

    def kkt(self, xe, lam):
        c, agx, W, J = self.inner.kkt(xe[:self.n], lam)
        c = c.copy()
        c[self.rows] += xe[self.n:]
        return (c, np.concatenate([agx, lam[self.rows]]), sp.block_diag([W, sp.csr_matrix((self.ns, self.ns))], format="csr"),
                sp.hstack([J, self.E], format="csr"))

    def con(self, xe):
        c = self.inner.con(xe[:self.n]).copy()
        c[self.rows] += xe[self.n:]
        return c


class CsrKkt:
    """Splits the value array of an upper-triangular row-major KKT CSR (NonLinearProgram.cpp:267-344) into the primal
    Hessian block and the equality Jacobian (rows at primal + CLoc)."""

    def __init__(self, outer, inner, n, m):
        self.n, self.m, self.outer, self.inner = n, m, np.asarray(outer), np.asarray(inner)

    def split(self, vals):
        n, m = self.n, self.m
        M = sp.csr_matrix((vals, self.inner, self.outer), shape=(len(self.outer) - 1,) * 2)
        U = M[:n, :n]
        W = U + sp.triu(U, 1).T
        return W.tocsr(), M[:n, n:n + m].T.tocsr()


class OracleProvider:
    """The oracle's restatement of NonLinearProgram (oracle/fullnlp.cpp) as the assembly: the CPU path.  A problem is one phase
    (the dict itself) or several `parts` sharing the solver vector."""

    def __init__(self, ob, prob):
        from asset_asrl_amd import synth
        n, m = prob["x0"].size, prob["n_equal"]
        nlp = ob.FullNlp(n, m, 0)
        affine = []
        for part in prob.get("parts", [prob]):
            ph, ix = part["phase"], part["ix"]
            nlp.add(1, ob.get_ode(part.get("ode_name", "reentry"), 0), ob.MODES[ph.TranscriptionMode], ix.BlockedControls, part["V"],
                    part["Cx"])
            cs = synth.MODE_CS[ph.TranscriptionMode]
            for kind, tag, F, name, V, Cx, consts in part["entries"]:
                if tag == "mesh_spacing":
                    nlp.add_mesh_spacing(1, cs, V, Cx)
                elif tag == "nodal_spacing":
                    nlp.add_single_mesh_spacing(1, consts.ravel(), V, Cx)
                elif tag == "control_spline":
                    nlp.add_control_spline(1, cs, part.get("usize", 2), V, Cx)
                elif kind == "objective":                        # an integral objective: the segment quadrature of an integrand
                    name, nx = part["integrands"][tag]
                    nlp.add_integral(0, ob.get_ode(name, 0), cs, nx, 0, V, Cx)
                    self.has_objective = True
                elif kind == "equality" and tag.endswith("_int"):    # an integral parameter function's quadratures: ONE shared row
                    name, nx = part["integrands"][tag]
                    nlp.add_integral(1, ob.get_ode(name, 0), cs, nx, 0, V, Cx)
                elif kind == "equality" and tag.endswith("_acc"):    # ... and its accumulation -scale * p on that row: linear, added to the
                    affine.append((int(Cx[0, 0]), int(V[0, 0]), float(F.compute(np.ones(1))[0])))   # oracle's values here (its function records have two inputs at least)
                elif kind == "equality":                         # a user function (the heating-rate bound, the muzzle-energy bound)
                    nlp.add(1, ob.get_ode(part.get("functions", {"eq0": "reentry_heating"})[tag], 0), ob.MODES["Function"], False, V, Cx)
                else:
                    raise ValueError(tag)
        nlp.analyze()
        nlp.set_solver_coeffs(np.zeros(nlp.num_solver_kkt))
        self.nlp, self.n, self.m = nlp, n, m
        self.csr = CsrKkt(*nlp.csr(), n, m)
        self.calls = 0
        self.B = sp.csr_matrix(([a[2] for a in affine], ([a[0] for a in affine], [a[1] for a in affine])), shape=(m, n)) if affine else None

    has_objective, pgx = False, None

    def kkt(self, x, lam):
        self.calls += 1
        _, self.pgx, agx, fxe, _, vals = self.nlp.eval(4, 1.0, x, lam, np.zeros(1))
        W, J = self.csr.split(vals)
        if self.B is not None:
            fxe, agx, J = fxe + self.B @ x, agx + self.B.T @ lam, (J + self.B).tocsr()
        return fxe, agx, W, J

    def con(self, x):
        c = self.nlp.eval(0, 1.0, x, np.zeros(self.m), np.zeros(1))[3]
        return c if self.B is None else c + self.B @ x

    def objective(self, x):
        return float(self.nlp.eval(0, 1.0, x, np.zeros(self.m), np.zeros(1))[0]) if self.has_objective else 0.0

    def objective_gradient(self):
        """of the last kkt() call"""
        return self.pgx if self.has_objective else 0.0


# ---------------------------------------------------------------------------------------------- the loop
def solve_ip(provider, x0, lb, ub, cost, tol=1e-7, maxit=400, mu=0.1, verbose=False, feasibility=False, step_cap=1.0,
             relative_push=False):
    """min cost.x  s.t. c(x) = 0, lb <= x <= ub (lb == ub: fixed).  -> (x, lam, info).  Primal-dual barrier Newton
    steps on the KKT system [W + Sigma + dw I, J^T; J, 0]: dw is raised until the step has positive curvature (the
    inertia-free test) and is no longer than step_cap; the step length is found by backtracking against a filter on
    (constraint violation, barrier objective).  feasibility = True is the 'solve' half of the reference's solve_optimize
    (test_Reentry.py:177): least-change Newton steps on c(x) = 0 inside the bounds (identity in place of the Lagrangian
    Hessian, no cost), stopping when the constraints hold."""
    if feasibility:
        cost, step_cap = np.zeros_like(cost), np.inf
    x = x0.copy()
    fixed = lb == ub
    x[fixed] = lb[fixed]
    free = np.flatnonzero(~fixed)
    hasl, hasu = np.isfinite(lb) & ~fixed, np.isfinite(ub) & ~fixed
    gap = np.where(np.isfinite(ub - lb), 1e-2 * (ub - lb), 1e-2)   # push the start into the interior
    if relative_push:   # a variable much smaller than 1e-2 that sits inside its bound stays where it is (the cannon ball's radius)
        gl_, gu_ = np.where(x > lb, np.minimum(gap, 0.5 * (x - lb)), gap), np.where(x < ub, np.minimum(gap, 0.5 * (ub - x)), gap)
    else:
        gl_, gu_ = gap, gap
    x = np.where(hasl, np.maximum(x, lb + gl_), x)
    x = np.where(hasu, np.minimum(x, ub - gu_), x)
    m, nf = provider.m, free.size
    lam = np.zeros(m)
    sl = lambda v: np.where(hasl, v - lb, 1.0)
    su = lambda v: np.where(hasu, ub - v, 1.0)
    zl, zu = np.where(hasl, mu / sl(x), 0.0), np.where(hasu, mu / su(x), 0.0)
    dw, filt = 0.0, []
    # a nonlinear part of the cost (an integral objective the assembly evaluates): value and gradient from the provider
    nonlin = (not feasibility) and getattr(provider, "has_objective", False)
    fobj = (lambda xx: provider.objective(xx)) if nonlin else (lambda xx: 0.0)
    pgrad = (lambda: provider.objective_gradient()) if nonlin else (lambda: 0.0)
    phi = lambda xx: cost @ xx + fobj(xx) - mu * (np.log(sl(xx))[hasl].sum() + np.log(su(xx))[hasu].sum())
    info = dict(iters=maxit, converged=False)
    for it in range(maxit):
        c, agx, W, J = provider.kkt(x, lam)
        pg = pgrad()
        Jf = J[:, free]
        if feasibility:
            if np.abs(c).max() < tol:
                info.update(iters=it, converged=True)
                break
            lam, agx, W = 0.0 * lam, 0.0 * agx, sp.identity(x.size, format="csr")
        elif it == 0:                                            # least-squares multiplier estimate at the start
            g0 = (cost + pg - np.where(hasl, mu / sl(x), 0.0) + np.where(hasu, mu / su(x), 0.0))[free]
            K0 = sp.bmat([[sp.identity(nf), Jf.T], [Jf, -1e-9 * sp.identity(m)]], format="csc")
            lam = spla.splu(K0).solve(np.concatenate([-g0, np.zeros(m)]))[nf:]
            c, agx, W, J = provider.kkt(x, lam)
            pg = pgrad()
        gl = cost + pg + agx - zl + zu
        err = lambda mu_: max(np.abs(gl[free]).max(), np.abs(c).max(), np.abs(sl(x) * zl - mu_)[hasl].max(initial=0.0),
                              np.abs(su(x) * zu - mu_)[hasu].max(initial=0.0))
        if not feasibility and err(0.0) < tol:
            info.update(iters=it, converged=True)
            break
        while err(mu) < 10.0 * mu and mu > tol / 10:
            mu, filt = max(tol / 10, min(0.2 * mu, mu ** 1.5)), []
        sig = zl / sl(x) + zu / su(x)
        gphi = cost + pg - np.where(hasl, mu / sl(x), 0.0) + np.where(hasu, mu / su(x), 0.0)
        Wf = W[free][:, free]
        rhs = -np.concatenate([(gphi + agx)[free], c])
        tau = max(0.99, 1.0 - mu)
        dw = dw / 3 if dw > 1e-6 else 0.0
        for _try in range(60):
            K = sp.bmat([[Wf + sp.diags(sig[free] + dw), Jf.T], [Jf, -1e-9 * sp.identity(m)]], format="csc")
            try:
                sol = spla.splu(K).solve(rhs)
            except RuntimeError:
                sol = np.full(nf + m, np.nan)
            if np.all(np.isfinite(sol)):
                dx = np.zeros_like(x)
                dx[free] = sol[:nf]
                curv = dx[free] @ (Wf @ dx[free]) + (sig + dw) @ (dx * dx)
                if curv >= 1e-8 * (dx @ dx) and np.abs(dx).max() <= step_cap:
                    break
            dw = max(1e-4, 4.0 * dw)
        dlam = sol[nf:]
        a = 1.0                                                  # fraction to the boundary
        neg, pos = hasl & (dx < 0), hasu & (dx > 0)
        if neg.any():
            a = min(a, (-tau * sl(x)[neg] / dx[neg]).min())
        if pos.any():
            a = min(a, (tau * su(x)[pos] / dx[pos]).min())
        th0, ph0, lin = np.abs(c).sum(), phi(x), gphi @ dx
        for _ls in range(40):                                    # backtracking against the filter
            xt = x + a * dx
            tht, pht = np.abs(provider.con(xt)).sum(), phi(xt)
            armijo = lin < 0 and th0 < 1e-4 * max(1.0, filt[0][0] if filt else th0) and pht <= ph0 + 1e-4 * a * lin
            if feasibility:
                if tht <= (1 - 1e-4 * a) * th0:                  # (monotone in the violation)
                    break
            elif armijo or all(tht <= (1 - 1e-5) * tf or pht <= pf - 1e-5 * tf for tf, pf in filt + [(th0, ph0)]):
                break
            a *= 0.5
        if not armijo:
            filt.append((th0, ph0))
        dzl = np.where(hasl, mu / sl(x) - zl - zl / sl(x) * dx, 0.0)
        dzu = np.where(hasu, mu / su(x) - zu + zu / su(x) * dx, 0.0)
        az = 1.0
        if (dzl < 0).any():
            az = min(az, (-tau * zl[dzl < 0] / dzl[dzl < 0]).min())
        if (dzu < 0).any():
            az = min(az, (-tau * zu[dzu < 0] / dzu[dzu < 0]).min())
        x, lam = xt, lam + a * dlam
        zl, zu = zl + az * dzl, zu + az * dzu
        zl = np.where(hasl, np.clip(zl, mu / (1e10 * sl(x)), 1e10 * mu / sl(x)), 0.0)
        zu = np.where(hasu, np.clip(zu, mu / (1e10 * su(x)), 1e10 * mu / su(x)), 0.0)
        if verbose:
            print(f"{it:4d} obj {cost @ x:+.8f} |c| {np.abs(c).max():.2e} |gl| {np.abs(gl[free]).max():.2e} mu {mu:.1e} "
                  f"a {a:.2e} dw {dw:.1e} |dx| {np.abs(dx).max():.2e} tries {_try} ls {_ls} filt {len(filt)}")
    info["objective"] = float(cost @ x + fobj(x))
    return x, lam, info


def solve_reentry(provider, prob, verbose=False, x0=None):
    """'solve' (feasibility) then 'optimize', as phase.solve_optimize() does in the reference's test.  -> (x, lam, info).
    With slack rows (the heating bound) the variables are [x ; s]; `x0` (a solution without the bound) warm-starts x."""
    rows = prob["slack_rows"]
    lb, ub, cost, xs = prob["lb"], prob["ub"], prob["cost"], (prob["x0"] if x0 is None else x0)
    if len(rows):
        g = provider.con(xs)[rows]
        provider = SlackRows(provider, rows)
        xs = np.concatenate([xs, np.maximum(-g, 1e-2)])
        lb, ub = np.concatenate([lb, np.zeros(len(rows))]), np.concatenate([ub, np.full(len(rows), np.inf)])
        cost = np.concatenate([cost, np.zeros(len(rows))])
    x, _, feas = solve_ip(provider, xs, lb, ub, cost, feasibility=True, mu=1e-6, verbose=verbose)
    x, lam, info = solve_ip(provider, x, lb, ub, cost, verbose=verbose)
    info["feasibility_iters"], info["feasible"] = feas["iters"], feas["converged"]
    return x[:prob["x0"].size], lam, info


def solve_linked(provider, prob, verbose=False, feasibility_first=False, **kw):
    """ocp.optimize() of a problem with slack rows and linear link rows (the two-phase cannon).  -> (x, lam, info)."""
    rows, (A, b) = prob["slack_rows"], prob["linear_rows"]
    lb, ub, cost, xs = prob["lb"], prob["ub"], prob["cost"], prob["x0"]
    n = xs.size
    sc = prob.get("var_scale")
    if sc is not None:
        provider = ScaledVars(provider, sc)
        lb, ub, cost, xs, A = lb / sc, ub / sc, cost * sc, xs / sc, sp.csr_matrix(A) @ sp.diags(sc)
    if len(rows):
        g = provider.con(xs)[rows]
        provider = SlackRows(provider, rows)
        slo, shi = prob.get("slack_bounds", (np.zeros(len(rows)), np.full(len(rows), np.inf)))
        s0 = np.maximum(-g, 1e-2) if "slack_bounds" not in prob else np.clip(-g, np.where(np.isfinite(slo), slo, -np.inf), shi)
        xs = np.concatenate([xs, s0])
        lb, ub = np.concatenate([lb, slo]), np.concatenate([ub, shi])
        cost = np.concatenate([cost, np.zeros(len(rows))])
        A = sp.hstack([A, sp.csr_matrix((A.shape[0], len(rows)))], format="csr")
    provider = LinearRows(provider, A, b)
    if feasibility_first:
        xs, _, _ = solve_ip(provider, xs, lb, ub, cost, feasibility=True, mu=1e-6, verbose=verbose, relative_push=kw.get("relative_push", False))
    x, lam, info = solve_ip(provider, xs, lb, ub, cost, verbose=verbose, **kw)
    info["feasible"] = bool(np.abs(provider.con(x)).max() < 1e-6)
    return (x[:n] if sc is None else sc * x[:n]), lam, info


def solve_optimize_only(provider, prob, verbose=False, **kw):
    """phase.optimize() from the initial guess, as test_CartPole.py:71 does (no feasibility stage).  -> (x, lam, info)."""
    x, lam, info = solve_ip(provider, prob["x0"], prob["lb"], prob["ub"], prob["cost"], verbose=verbose, **kw)
    info["feasible"] = bool(np.abs(provider.con(x)).max() < 1e-6)
    return x, lam, info


def solve_adaptive(make_provider, rebuild, prob, meshinfo, verbose=False, solver=None, **kw):
    """The reference's adaptive mesh loop (ODEPhaseBase.cpp:1639-1673 around checkMesh / updateMesh, :1443-1542) with this harness
    as the solver: optimise, estimate the error of the solution (`meshinfo(phase)` -> (tsnd, errors, dist): the device estimator or
    the oracle's), stop when it is below the phase's MeshTol, otherwise re-mesh by the error density and optimise again from the
    re-distributed solution.  `rebuild(phase)` -> the problem on the phase's new mesh.  -> (problem, x, lam, info)."""
    ph = prob["phase"]
    for it in range(ph.MaxMeshIters + 1):
        prov = make_provider(prob)
        try:
            x, lam, info = (solver or solve_optimize_only)(prov, prob, verbose=verbose, **kw)
        finally:
            if hasattr(prov, "close"):
                prov.close()
        ph.ActiveTraj = prob["ix"].collectSolverOutput(x)[0]
        done = ph.checkMesh(meshinfo=lambda: meshinfo(ph))
        if verbose:
            m = ph.MeshIters[-1]
            print(f"mesh iteration {it}: {m.numsegs} segments, max error {m.max_error:.3e}, objective {info['objective']:.8f}")
        if done or not info["converged"] or it == ph.MaxMeshIters:
            break
        ph.updateMesh()
        prob = rebuild(ph)
    info["mesh_iterations"], info["mesh_converged"], info["segments"] = len(ph.MeshIters), ph.MeshConverged, ph.numDefects
    return prob, x, lam, info


class DeviceProvider:
    """The product: every function evaluated on the GPU through the C ABI and assembled by the C++ host shim's
    KktAssembly (asset_asrl_amd/host/kkt_assembly.h) -- `shim` is tests/host_shim_driver.cpp compiled by the test."""

    def __init__(self, shim, prob):
        import ctypes as C
        from asset_asrl_amd import _lib, jit
        n, m = prob["x0"].size, prob["n_equal"]
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)

        class FnDesc(C.Structure):
            _fields_ = [("kind", C.c_int), ("name", C.c_char_p), ("mode", C.c_int), ("blocked", C.c_int), ("ir", C.c_int),
                        ("orr", C.c_int), ("nappl", C.c_int), ("vindex", ip), ("cindex", ip), ("consts", dp), ("nconst", C.c_int)]
        mode_ids = {"LGL3": _lib.LGL3, "LGL5": _lib.LGL5, "LGL7": _lib.LGL7, "Trapezoidal": _lib.TRAPEZOIDAL}
        fns = []                                                 # (kind 1: equality; 0: objective -- host/kkt_assembly.h)
        for part in prob.get("parts", [prob]):
            ph, ix = part["phase"], part["ix"]
            fns.append((jit.ensure_kernel(ph._active_ode(), ph.TranscriptionMode, ix.BlockedControls), mode_ids[ph.TranscriptionMode],
                        int(ix.BlockedControls), part["V"], part["Cx"], None, 1))
            fns += [(jit.ensure_function(F, name), _lib.FUNCTION, 0, V, Cx, consts, 0 if kind == "objective" else 1)
                    for kind, _, F, name, V, Cx, consts in part["entries"]]
        self.has_objective = any(f[6] == 0 for f in fns)
        self._keep, descs = [], (FnDesc * len(fns))()
        for k, (name, mode, blocked, V, Cx, consts, fkind) in enumerate(fns):
            v, c = np.ascontiguousarray(V, dtype=np.int32), np.ascontiguousarray(Cx, dtype=np.int32)
            cc = None if consts is None else np.ascontiguousarray(consts, dtype=float)
            self._keep += [v, c, cc]
            descs[k] = FnDesc(fkind, name.encode(), mode, blocked, v.shape[1], c.shape[1], v.shape[0], v.ctypes.data_as(ip),
                              c.ctypes.data_as(ip), None if cc is None else cc.ctypes.data_as(dp), 0 if cc is None else cc.shape[1])
        err = C.create_string_buffer(512)
        shim.fullnlp_create.restype = C.c_void_p
        self._h = C.c_void_p(shim.fullnlp_create(descs, len(fns), n, m, 0, err, 512))
        assert self._h, err.value
        sz = (C.c_int * 4)()
        shim.fullnlp_sizes(self._h, sz)
        self.nnz = sz[1]
        outer, inner, locs = np.zeros(sz[0] + 1, np.int32), np.zeros(sz[1], np.int32), np.zeros(sz[2], np.int32)
        shim.fullnlp_structure(self._h, outer.ctypes.data_as(ip), inner.ctypes.data_as(ip), locs.ctypes.data_as(ip))
        shim.fullnlp_set_solver_coeffs(self._h, np.zeros(sz[3]).ctypes.data_as(dp))
        self.shim, self.n, self.m, self.csr, self.calls, self._C = shim, n, m, CsrKkt(outer, inner, n, m), 0, C

    def _eval(self, level, x, lam):
        C, dp = self._C, self._C.POINTER(self._C.c_double)
        x, lam = np.ascontiguousarray(x, dtype=float), np.ascontiguousarray(lam, dtype=float)
        val, err = C.c_double(0.0), C.create_string_buffer(512)
        pgx, agx, fxe, fxi, vals = np.zeros(self.n), np.zeros(self.n), np.zeros(self.m), np.zeros(1), np.zeros(self.nnz)
        p = lambda a: a.ctypes.data_as(dp)
        rc = self.shim.fullnlp_eval(self._h, level, C.c_double(1.0), p(x), p(lam), p(fxi), C.byref(val), p(pgx), p(agx), p(fxe),
                                    p(fxi), p(vals), err, 512)
        assert rc == 0, err.value
        self._val, self._pgx = float(val.value), pgx
        return fxe, agx, vals

    def kkt(self, x, lam):
        self.calls += 1
        fxe, agx, vals = self._eval(4, x, lam)
        self.pgx = self._pgx
        return (fxe, agx) + self.csr.split(vals)

    def con(self, x):
        return self._eval(0, x, np.zeros(self.m))[0]

    def objective(self, x):
        self._eval(0, x, np.zeros(self.m))
        return self._val if self.has_objective else 0.0

    def objective_gradient(self):
        """of the last kkt() call"""
        return self.pgx if self.has_objective else 0.0

    def close(self):
        if self._h:
            self.shim.fullnlp_destroy(self._h)
            self._h = None
