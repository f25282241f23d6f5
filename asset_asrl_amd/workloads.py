"""Dynamics of the BASELINE.json configurations, stated from their mathematical models -- WORKLOAD DEFINITIONS, not product
code: bench.py, the tests and the kernels compiled into libasset_hip.so (build.py: one translation unit per
(ODE, transcription, control mode)) need them by name.

Each model below is written from its equations of motion in this build's own decomposition (round 5; round 4's text followed
the reference's example scripts statement by statement).  The reference solves the same problems in
examples/Brachistochrone.py, examples/Reentry.py, examples/MultiSpacecraftOptimization.py and examples/BettsLowThrust.py; the
state / control ORDER of each model is the reference's, because the golden vectors and the oracle's independent right-hand
sides (oracle/odes.h) are indexed that way.  ``tests/test_workloads_dynamics.py`` holds every model to the values, Jacobians
and adjoint Hessians round 4's definitions produced at seeded points (fixture ``tests/golden/dynamics/workload_dynamics.npz``).

The DSL itself (``ODEArguments``, ``ODEBase``) and the synthetic 32-state ODE this build defines are in ``ode.py``.
"""
from __future__ import annotations

import math

from . import vf
from .ode import ODEArguments, ODEBase, Synthetic32


# =========================================================================== bead on a wire

class Brachistochrone(ODEBase):
    """A bead sliding without friction: the wire's tangent makes the angle ``theta`` (the control) with the downward
    vertical, so the velocity is ``v (sin theta, -cos theta)`` and gravity accelerates the bead by its component along
    the tangent.  States ``(x, y, v)``."""

    def __init__(self, g: float = 9.81):
        arg = ODEArguments(3, 1)
        speed = arg.XVar(2)
        incline = arg.UVar(0)
        along, down = vf.sin(incline), vf.cos(incline)
        rates = [speed * along, speed * down * -1.0, down * g]
        super().__init__(vf.stack(rates), 3, 1, name="brachistochrone")


# =========================================================================== lifting re-entry vehicle

class _ReentryVehicle:
    """Constants of the shuttle-like glider (Betts, *Practical Methods for Optimal Control*, the re-entry example), in
    English units and non-dimensionalised with the vehicle's mass, 1e5 ft and one minute."""

    weight_lbf, g0_fps2 = 203000.0, 32.2
    ref_area_ft2 = 2690.0
    planet_radius_ft = 20902900.0
    planet_mu = 0.140765e17            # ft^3 / s^2
    sea_level_density = 0.002378       # slug / ft^3
    scale_height_ft = 23800.0
    lift_poly = (-0.20704, 0.029244)                   # C_L(alpha in degrees)
    drag_poly = (0.07854, -0.61592e-2, 0.621408e-3)    # C_D(alpha in degrees)
    unit_length_ft, unit_time_s = 100000.0, 60.0

    @classmethod
    def nondimensional(cls):
        Lu, Tu = cls.unit_length_ft, cls.unit_time_s
        mass_unit = cls.weight_lbf / cls.g0_fps2          # slugs: the vehicle's mass is one mass unit
        density_unit = mass_unit / Lu ** 3
        return dict(
            radius=cls.planet_radius_ft / Lu,
            mu=cls.planet_mu / (Lu ** 3 / Tu ** 2),
            area=cls.ref_area_ft2 / Lu ** 2,
            mass=(cls.weight_lbf / cls.g0_fps2) / mass_unit,
            rho0=cls.sea_level_density / density_unit,
            scale_height=cls.scale_height_ft / Lu)


class ShuttleReentry(ODEBase):
    """Point-mass glide over a spherical non-rotating planet with an exponential atmosphere.  States: altitude, latitude,
    speed, flight-path angle, heading; controls: angle of attack and bank angle (radians)."""

    def __init__(self):
        c = _ReentryVehicle.nondimensional()
        arg = ODEArguments(5, 2)
        altitude, latitude, speed, fpa, heading = (arg.XVar(i) for i in range(5))
        attack, bank = arg.UVar(0), arg.UVar(1)

        # aerodynamics: coefficients are polynomials in the angle of attack in degrees; force = q S C
        attack_deg = attack * (180.0 / math.pi)
        l0, l1 = _ReentryVehicle.lift_poly
        d0, d1, d2 = _ReentryVehicle.drag_poly
        lift_coeff = attack_deg * l1 + l0
        drag_coeff = d0 + attack_deg * d1 + (attack_deg ** 2) * d2
        density = vf.exp(altitude * (-1.0 / c["scale_height"])) * c["rho0"]
        lift = lift_coeff * 0.5 * c["area"] * density * (speed ** 2)
        drag = drag_coeff * 0.5 * c["area"] * density * (speed ** 2)

        radius = altitude + c["radius"]
        gravity = c["mu"] / (radius ** 2)
        sin_fpa, cos_fpa = vf.sin(fpa), vf.cos(fpa)
        turn_rate = speed / radius                     # angular rate of the local horizon per unit of cos(fpa)
        mass = c["mass"]

        rates = [
            speed * sin_fpa,                                                              # climb
            turn_rate * cos_fpa * vf.cos(heading),                                        # northward
            -drag / mass - gravity * sin_fpa,                                             # along the velocity
            (lift / (mass * speed)) * vf.cos(bank) + cos_fpa * (turn_rate - gravity / speed),   # pitching the velocity
            lift * vf.sin(bank) / (mass * speed * cos_fpa)
            + turn_rate * cos_fpa * vf.sin(heading) * vf.tan(latitude),                   # turning it
        ]
        super().__init__(vf.stack(rates), 5, 2, name="reentry")


# =========================================================================== point mass about one attractor

class TwoBody(ODEBase):
    """``r'' = -mu r / |r|^3 (+ a_max u)``: Cartesian position and velocity, optional thrust acceleration along the control
    vector scaled by ``ltacc`` (``ltacc=False``: ballistic, no controls)."""

    def __init__(self, P1mu: float = 1.0, ltacc=0.01):
        thrusting = ltacc is not False
        arg = ODEArguments(6, 3 if thrusting else 0)
        position, velocity = arg.XVec().head3(), arg.XVec().tail3()
        pull = position.normalized_power3() * (-P1mu)
        accel = pull + arg.UVec() * ltacc if thrusting else pull
        super().__init__(vf.stack([velocity, accel]), 6, 3 if thrusting else 0,
                         name="twobody_lt" if thrusting else "twobody")


# =========================================================================== low-thrust transfer in equinoctial elements

class _EarthOrbitUnits:
    """Betts' low-thrust transfer (ibid., the 'low-thrust orbit transfer' example): Earth radius as the length unit, the
    circular period at that radius over 2 pi as the time unit, the initial weight as the force unit."""

    g0, weight0 = 32.174, 1.0
    mu_ft = 1.407645794e16
    radius_ft = 20925662.73
    thrust_lbf, isp_s = 4.446618e-3, 450.0
    zonal = {2: 1082.639e-6, 3: -2.565e-6, 4: -1.608e-6}

    @classmethod
    def nondimensional(cls):
        Lu = cls.radius_ft
        Tu = Lu / math.sqrt(cls.mu_ft / Lu)
        mass_unit = cls.weight0 / cls.g0
        accel_unit = Lu / Tu ** 2
        return dict(Re=cls.radius_ft / Lu, mu=cls.mu_ft / (Lu ** 3 / Tu ** 2),
                    Thrust=cls.thrust_lbf / (mass_unit * accel_unit), Isp=cls.isp_s / Tu, gs=cls.g0 / accel_unit,
                    J2=cls.zonal[2], J3=cls.zonal[3], J4=cls.zonal[4])


def _legendre_with_slope(s):
    """``{k: (P_k(s), P_k'(s))}`` for k = 2, 3, 4."""
    s2 = s * s
    return {
        2: (s2 * 1.5 - 0.5, s * 3.0),
        3: ((s2 * 2.5 - 1.5) * s, s2 * 7.5 - 1.5),
        4: ((s2 * 4.375 - 3.75) * s2 + 0.375, (s2 * 17.5 - 7.5) * s),
    }


def zonal_perturbation_rtn(p, f, g, h, k, sinL, cosL, mu, Re, J):
    """Acceleration of the zonal harmonics ``J = {k: J_k}`` in the radial / transverse / orbit-normal frame, in closed form
    in the equinoctial elements.

    With ``dg = dg_n i_n - dg_r i_r`` (``dg_n = -(mu cos(phi) / r^2) sum_k (Re/r)^k P_k' J_k``,
    ``dg_r = -(mu / r^2) sum_k (k+1) (Re/r)^k P_k J_k``, ``i_n`` the local north) and ``cos(phi) i_n = e_z - sin(phi) i_r``,
    the components on the frame ``(i_r, i_t, i_h)`` are ``-dg_r``, ``(dg_n / cos phi) e_z.i_t`` and ``(dg_n / cos phi)
    e_z.i_h`` -- and the polar components of the equinoctial frame are rational in ``h, k`` and the true longitude:
    ``e_z.i_r = sin(phi) = 2 (h sin L - k cos L) / s^2``, ``e_z.i_t = 2 (h cos L + k sin L) / s^2``,
    ``e_z.i_h = (1 - h^2 - k^2) / s^2`` with ``s^2 = 1 + h^2 + k^2``.  No Cartesian state, no square root, no division by
    ``cos(phi)``."""
    q = f * cosL + g * sinL + 1.0
    radius = p / q
    chi2 = h * h + k * k
    inv_s2 = 1.0 / (chi2 + 1.0)
    pole_r = (h * sinL - k * cosL) * 2.0 * inv_s2           # sin(latitude)
    pole_t = (h * cosL + k * sinL) * 2.0 * inv_s2
    pole_h = (1.0 - chi2) * inv_s2
    ratio = Re / radius
    leg = _legendre_with_slope(pole_r)
    radial_sum, north_sum = None, None
    ratio_k = ratio
    for order in (2, 3, 4):
        ratio_k = ratio_k * ratio
        Pk, dPk = leg[order]
        tr = Pk * ((order + 1) * J[order]) * ratio_k
        tn = dPk * J[order] * ratio_k
        radial_sum = tr if radial_sum is None else radial_sum + tr
        north_sum = tn if north_sum is None else north_sum + tn
    central = mu / (radius * radius)
    return central * radial_sum, -central * north_sum * pole_t, -central * north_sum * pole_h


def gauss_equinoctial_rates(p, f, g, h, k, sinL, cosL, accel, mu):
    """Gauss' variational equations in modified equinoctial elements for a perturbing acceleration ``accel`` =
    (radial, transverse, normal): ``y' = A(y) accel + b(y)``."""
    ar, at, an = accel
    q = f * cosL + g * sinL + 1.0
    root = vf.sqrt(p) * (1.0 / math.sqrt(mu))              # sqrt(p / mu)
    twist = (h * sinL - k * cosL) * an / q                  # the normal component's pull on (f, g, L)
    half_s2 = (h * h + k * k + 1.0) * 0.5
    pdot = p * at * 2.0 / q
    fdot = ar * sinL + ((q + 1.0) * cosL + f) * at / q - g * twist
    gdot = ((q + 1.0) * sinL + g) * at / q - ar * cosL + f * twist
    hdot = cosL * half_s2 * an / q
    kdot = sinL * half_s2 * an / q
    Ldot = (q / p) * (q / p) * mu + twist
    return [r * root for r in (pdot, fdot, gdot, hdot, kdot, Ldot)]


class LTModel(ODEBase):
    """Betts' low-thrust orbit transfer: six modified equinoctial elements ``(p, f, g, h, k, L)`` and the weight; the control
    is a thrust direction in the radial / transverse / normal frame (normalised here), the parameter a throttle factor
    ``tau`` (thrust ``T (1 + tau / 100)``); perturbations: zonal harmonics J2..J4 and the thrust."""

    def __init__(self, mu=None, T=None, gs=None, Isp=None, Re=None, J2=None):
        c = _EarthOrbitUnits.nondimensional()
        mu = c["mu"] if mu is None else mu
        T = c["Thrust"] if T is None else T
        gs = c["gs"] if gs is None else gs
        Isp = c["Isp"] if Isp is None else Isp
        Re = c["Re"] if Re is None else Re
        zonal = {2: c["J2"] if J2 is None else J2, 3: c["J3"], 4: c["J4"]}

        arg = ODEArguments(7, 3, 1)
        p, f, g, h, k, L, weight = (arg.XVar(i) for i in range(7))
        direction = arg.UVec().normalized()
        throttle = arg.PVar(0) * 0.01 + 1.0
        sinL, cosL = vf.sin(L), vf.cos(L)

        oblate = zonal_perturbation_rtn(p, f, g, h, k, sinL, cosL, mu, Re, zonal)
        push = throttle * (gs * T) / weight
        accel = [oblate[i] + direction.coeff(i) * push for i in range(3)]
        rates = gauss_equinoctial_rates(p, f, g, h, k, sinL, cosL, accel, mu)
        rates.append(throttle * (-T / Isp))
        super().__init__(vf.stack(rates), 7, 3, 1, name="betts_lowthrust")


# name -> class of every ODE compiled into libasset_hip.so (build.py)
ODE_LIBRARY = {
    "brachistochrone": Brachistochrone,
    "reentry": ShuttleReentry,
    "twobody_lt": TwoBody,
    "betts_lowthrust": LTModel,
    "synthetic32": Synthetic32,
}
