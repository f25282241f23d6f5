// ORACLE (test infrastructure, never shipped or imported by the product path).
//
// Right-hand sides of the BASELINE.json config ODEs, written as templates over the scalar type so
// the same text yields values (S=double) and exact derivatives (S=AD2<N>).  Each function restates
// the dynamics of a reference example script (input layout y = [x, t, u, p]):
//   brachistochrone  /root/reference/examples/Brachistochrone.py:15-33
//   reentry          /root/reference/examples/Reentry.py:14-97
//   twobody_lt       /root/reference/examples/MultiSpacecraftOptimization.py:17-34
//   betts_lowthrust  /root/reference/examples/BettsLowThrust.py:22-48,212-400
//   synthetic32      SURVEY.md section 8(d) (defined by this build)
#pragma once
#include <cmath>
#include <cstdint>

#include "ad2.h"
#include "interp_table.h"

namespace oracle_odes {

using std::cos;
using std::exp;
using std::sin;
using std::sqrt;
using std::tan;

// ------------------------------------------------------------------ brachistochrone (3,1,0)
template <class S>
void brachistochrone(const S* y, S* f, const void*) {
  const double g = 9.81;
  const S& v = y[2];
  const S& theta = y[4];
  f[0] = sin(theta) * v;
  f[1] = -1.0 * cos(theta) * v;
  f[2] = g * cos(theta);
}

// ------------------------------------------------------------------ forced Van der Pol with parameter (2,1,1)
// Not a BASELINE config: the independent check for ODEs that reach the device through run-time compilation
// (tests/test_gpu_jit.py defines the same right-hand side in the product's expression DSL).
template <class S>
void vanderpol(const S* y, S* f, const void*) {
  const S &x0 = y[0], &x1 = y[1], &t = y[2], &u = y[3], &mu = y[4];
  f[0] = x1;
  f[1] = mu * (1.0 - x0 * x0) * x1 - x0 + u * exp(-0.1 * t);
}

// ------------------------------------------------------------------ a switched oscillator (2,1,0): conditionals, |.| and sign
// Not a BASELINE config: the independent check for the DSL's ifelse / abs / sign nodes on the device (round 6; the reference's
// IfElseFunction and ConditionalStatement, CommonFunctions/Conditional.h:19-260 -- value and derivatives of the branch the test picks).
// tests/helpers.py: make_switched defines the same right-hand side in the product's expression DSL.
static inline double valof(double a) { return a; }
template <int N> static inline double valof(const AD2<N>& a) { return a.v; }
static inline double absv(double a) { return std::fabs(a); }
template <int N> static inline AD2<N> absv(const AD2<N>& a) { return ad2_unary(a, std::fabs(a.v), (a.v > 0.0) - (a.v < 0.0), 0.0); }
template <class S>
void switched(const S* y, S* f, const void*) {
  const S &x0 = y[0], &x1 = y[1], &t = y[2], &u = y[3];
  const bool stiff = valof(x0) > 0.25 && valof(x1) >= 0.0;
  const S spring = stiff ? S(3.0 * x0 + 0.5 * x0 * x0) : S(x0 * 1.0);
  const double dir = (valof(x1) > 0.0) - (valof(x1) < 0.0);          // sign(x1): piecewise constant, no derivative
  f[0] = x1 + (valof(t) < 5.0 ? S(0.1 * sin(x0)) : S(0.0 * x0));
  f[1] = -1.0 * spring - 0.3 * absv(x1) * x1 - 0.05 * dir + u * cos(t);
}

// ------------------------------------------------------------------ a sounding rocket on tabulated data (2,1,0): InterpTable1D
// Not a BASELINE config: the independent check for tables in a device body (round 6; the reference's InterpTable1D / InterpFunction1D,
// CommonFunctions/InterpTable1D.h).  Density over the altitude from a cubic table with UNEVEN abscissae, thrust over time from a linear
// one, a two-valued cubic table over the speed.  tests/helpers.py: make_tabulated defines the same right-hand side with vf.InterpTable1D.
template <class S>
void tabulated(const S* y, S* f, const void*) {
  const S &x0 = y[0], &x1 = y[1], &t = y[2], &u = y[3];
  const S rho = tab_eval(tabulated_table(0), 0, x0);
  const S thr = tab_eval(tabulated_table(1), 0, t);
  const S w0 = tab_eval(tabulated_table(2), 0, x1), w1 = tab_eval(tabulated_table(2), 1, x1);
  f[0] = x1 + 0.1 * w0 * w1;
  f[1] = u * thr - 0.05 * rho * x1 * x1 - 1.0;
}

// ------------------------------------------------------------------ coupled oscillators (12,3,2): wide shapes with u and p
// Not a BASELINE config: in LGL7 its segment has IR = 66 inputs, which puts a run-time compiled ODE with controls and
// parameters through the four-wave dense kernel (tests/test_gpu_jit.py defines the same right-hand side in the DSL).
template <int n, class S>
void coupled_n(const S* y, S* f) {
  const S& t = y[n];
  const S* u = y + n + 1;
  const S &p0 = y[n + 4], &p1 = y[n + 5];
  for (int k = 0; k < n; k++)
    f[k] = -0.5 * y[k] + sin(y[(k + 1) % n]) * y[(k + 5) % n] * u[k % 3] + p0 * cos(t) + p1 * y[k] * y[(k + 7) % n];
}
template <class S>
void coupled12(const S* y, S* f, const void*) { coupled_n<12>(y, f); }
// (16,3,2): with BlockConstant control its LGL7 segment has IR = 73 -- the four-wave kernel with five parameter columns
template <class S>
void coupled16(const S* y, S* f, const void*) { coupled_n<16>(y, f); }

// ------------------------------------------------------------------ driven chains (n,3,0): wide shapes with controls, no parameters
// Not a BASELINE config: (14,3,0) in LGL7 has IR = 72 and (20,3,0) in LGL5 IR = 72 -- run-time compiled ODEs WITH control rows
// for the row-wise wide dense stage (csrc/defect_rows.h), which takes shapes without segment parameters
// (tests/test_gpu_jit.py defines the same right-hand side in the DSL).
template <int n, class S>
void driven_n(const S* y, S* f) {
  const S& t = y[n];
  const S* u = y + n + 1;
  for (int k = 0; k < n; k++)
    f[k] = -0.5 * y[k] + sin(y[(k + 1) % n]) * y[(k + 5) % n] * u[k % 3] + 0.3 * cos(t) * y[(k + 3) % n] + 0.1 * u[(k + 1) % 3] * u[(k + 1) % 3];
}
template <class S>
void driven14(const S* y, S* f, const void*) { driven_n<14>(y, f); }
template <class S>
void driven20(const S* y, S* f, const void*) { driven_n<20>(y, f); }

// ------------------------------------------------------------------ Delta III ascent (7,3,0), four stages
// The dynamics of the reference's four-phase full-problem test (asset_asrl/test/test_FullProblems/test_Delta3Launch.py:14-131;
// units and stage constants :16-99): y = [r(3), v(3), m, t, u(3)], thrust along u / |u|, drag in an exponential atmosphere that
// rotates with the Earth.  delta3_<k>: stage k's thrust and mass flow.  norm3: |a| of three inputs (the control-norm and radius
// bounds, :216-243) -- record (1, 1, 0).  delta3_orbit: the five insertion conditions on (r, v) at the end of the last phase,
// record (5, 0, 0): semi-major axis, eccentricity, inclination, node and perigee directions.  The reference writes the last three
// with arccos and branches (:133-157); here they are the smooth conditions with the same zero set near the solution --
// cos i, the node line's direction, the cosine of the argument of perigee -- so that no branch has to be differentiated.
// tests/kkt_harness.py defines all of them in the product's expression DSL.
namespace delta3_units {
constexpr double g0 = 9.80665, Lstar = 6378145.0, Tstar = 961.0, Mstar = 301454.0;
constexpr double Astar = Lstar / (Tstar * Tstar), Rhostar = Mstar / (Lstar * Lstar * Lstar);
constexpr double Mustar = (Lstar * Lstar * Lstar) / (Tstar * Tstar), Fstar = Astar * Mstar;
constexpr double mu = 3.986012e14 / Mustar, Re = 1.0, We = 7.29211585e-5 * Tstar;
constexpr double RhoAir = 1.225 / Rhostar, h_scale = 7200.0 / Lstar, g = g0 / Astar;
constexpr double CD = 0.5, Sref = 4.0 * M_PI / (Lstar * Lstar);
constexpr double TS = 628500.0 / Fstar, T1 = 1083100.0 / Fstar, T2 = 110094.0 / Fstar;
constexpr double IS = 283.33364 / Tstar, I1 = 301.68 / Tstar, I2 = 467.21 / Tstar;
constexpr double thrust[4] = {6 * TS + T1, 3 * TS + T1, T1, T2};
constexpr double mdot[4] = {(6 * TS / IS + T1 / I1) / g, (3 * TS / IS + T1 / I1) / g, T1 / (g * I1), T2 / (g * I2)};
}  // namespace delta3_units
template <int PH, class S>
void delta3_stage(const S* y, S* f) {
  using namespace delta3_units;
  const S *r = y, *v = y + 3, *u = y + 8;
  const S& m = y[6];
  const S rn = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  const S un = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  const S rho = RhoAir * exp(-1.0 * (rn - Re) / h_scale);
  // vr = v + r x [0, 0, We]
  const S vr[3] = {v[0] + r[1] * We, v[1] - r[0] * We, v[2]};
  const S vrn = sqrt(vr[0] * vr[0] + vr[1] * vr[1] + vr[2] * vr[2]);
  const S r3 = rn * rn * rn;
  for (int k = 0; k < 3; k++) {
    f[k] = v[k];
    const S D = (-0.5 * CD * Sref) * rho * (vr[k] * vrn);
    f[3 + k] = (-1.0 * mu) * r[k] / r3 + (thrust[PH] * (u[k] / un) + D) / m;
  }
  f[6] = 0.0 * m - mdot[PH];
}
template <class S> void delta3_1(const S* y, S* f, const void*) { delta3_stage<0>(y, f); }
template <class S> void delta3_2(const S* y, S* f, const void*) { delta3_stage<1>(y, f); }
template <class S> void delta3_3(const S* y, S* f, const void*) { delta3_stage<2>(y, f); }
template <class S> void delta3_4(const S* y, S* f, const void*) { delta3_stage<3>(y, f); }
template <class S>
void norm3(const S* y, S* f, const void*) { f[0] = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]); }
template <class S>
void delta3_orbit(const S* y, S* f, const void*) {
  using namespace delta3_units;
  const double at = 24361140.0 / Lstar, et = 0.7308, it = 28.5 * M_PI / 180.0, Ot = 269.8 * M_PI / 180.0, Wt = 130.5 * M_PI / 180.0;
  const S *r = y, *v = y + 3;
  const S h[3] = {r[1] * v[2] - r[2] * v[1], r[2] * v[0] - r[0] * v[2], r[0] * v[1] - r[1] * v[0]};
  const S rn = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  const S v2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const S hn = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
  // evec = v x h / mu - r / |r|
  const S e[3] = {(v[1] * h[2] - v[2] * h[1]) / mu - r[0] / rn, (v[2] * h[0] - v[0] * h[2]) / mu - r[1] / rn,
                  (v[0] * h[1] - v[1] * h[0]) / mu - r[2] / rn};
  const S en = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  const S nx = -1.0 * h[1], ny = h[0];                      // node line z x h
  const S nn = sqrt(nx * nx + ny * ny);
  f[0] = -0.5 * mu / (0.5 * v2 - mu / rn) - at;
  f[1] = en - et;
  f[2] = h[2] / hn - std::cos(it);
  f[3] = (nx * std::sin(Ot) - ny * std::cos(Ot)) / nn;
  f[4] = (nx * e[0] + ny * e[1]) / (nn * en) - std::cos(Wt);
}

// ------------------------------------------------------------------ a family of shapes (n, m, p) for the narrow kernels
// Not BASELINE configs: one smooth right-hand side for ANY (states, controls, parameters), so that run-time compiled ODEs of
// unusual dimensions -- one state, no controls, node strides that are / are not multiples of four, N + 1 = 16 and 17, two row
// groups of defect rows -- reach every narrow dense-part form (csrc/defect_resident.h tile form, csrc/defect_rowdpp.h row form,
// the fallback of csrc/defect_kernels.h).  tests/helpers.py: make_shape(n, m, p) defines the same right-hand side in the DSL.
//   x_k' = -x_k/2 + sin(x_{k+1}) x_{k+2} [u_{k mod m}] + 0.3 cos(t) x_{k+3} [+ 0.1 u_{(k+1) mod m}^2] [+ p_0 x_k x_{k+1} + p_{p-1} cos t]
template <int n, int m, int p, class S>
void shape_nmp(const S* y, S* f) {
  const S& t = y[n];
  const S* u = y + n + 1;
  const S* par = y + n + 1 + m;
  for (int k = 0; k < n; k++) {
    S v = sin(y[(k + 1) % n]) * y[(k + 2) % n];
    if (m > 0) v = v * u[k % (m > 0 ? m : 1)];
    v = v - 0.5 * y[k] + 0.3 * cos(t) * y[(k + 3) % n];
    if (m > 0) v = v + 0.1 * u[(k + 1) % (m > 0 ? m : 1)] * u[(k + 1) % (m > 0 ? m : 1)];
    if (p > 0) v = v + par[0] * y[k] * y[(k + 1) % n] + par[p > 0 ? p - 1 : 0] * cos(t);
    f[k] = v;
  }
}
#define ORACLE_SHAPE_ODE(n, m, p) \
  template <class S>              \
  void shape_##n##_##m##_##p(const S* y, S* f, const void*) { shape_nmp<n, m, p>(y, f); }
ORACLE_SHAPE_ODE(1, 0, 0)
ORACLE_SHAPE_ODE(1, 1, 0)
ORACLE_SHAPE_ODE(2, 1, 0)
ORACLE_SHAPE_ODE(3, 0, 1)
ORACLE_SHAPE_ODE(4, 4, 0)
ORACLE_SHAPE_ODE(5, 3, 2)
ORACLE_SHAPE_ODE(6, 0, 0)
ORACLE_SHAPE_ODE(8, 3, 1)
ORACLE_SHAPE_ODE(10, 4, 0)
ORACLE_SHAPE_ODE(11, 4, 0)

// ------------------------------------------------------------------ a nonlinear path constraint, 2 outputs of 6 inputs
// Not an ODE: the independent check for plain functions batched over applications (mode FUNCTION);
// tests/test_gpu_function.py defines the same function in the product's expression DSL.
template <class S>
void pathcon(const S* y, S* f, const void*) {
  const S &x0 = y[0], &x1 = y[1], &x2 = y[2], &t = y[3], &u0 = y[4], &u1 = y[5];
  f[0] = x0 * x0 + x1 * u0 - sin(x2);
  f[1] = u0 * u0 + u1 * u1 - 1.0 + t * x0 * exp(-x1);
}

// ------------------------------------------------------------------ the heating-rate bound of the shuttle re-entry problem
// (q(h, v, alpha) - Qlimit) / Qlimit with q = qa(alpha) qr(h, v), the function the reference's full-problem test adds with
// addUpperFuncBound("Path", QFunc(), [0, 2, 6], Qlimit, 1 / Qlimit)  (asset_asrl/test/test_FullProblems/test_Reentry.py:99-109,
// 206; constants :14-47).  Record (1, 1, 0): one output of three inputs (h, v, alpha).  tests/kkt_harness.py defines the same
// function in the product's expression DSL.
template <class S>
void reentry_heating(const S* y, S* f, const void*) {
  const double g0 = 32.2, W = 203000.0, Lstar = 100000.0, Tstar = 60.0, Mstar = W / g0;
  const double Vstar = Lstar / Tstar, Rhostar = Mstar / (Lstar * Lstar * Lstar);
  const double rho0 = 0.002378 / Rhostar, h_ref = 23800.0 / Lstar;
  const double c0 = 1.0672181, c1 = -0.19213774e-1, c2 = 0.21286289e-3, c3 = -0.10117e-5, Qlimit = 70.0;
  const S &h = y[0], &v = y[1], &alpha = y[2];
  const S alphadeg = (180.0 / M_PI) * alpha;
  const S rhodim = (rho0 * Rhostar) * exp(-1.0 * h / h_ref);
  const S vdim = v * Vstar;
  const S qr = 17700.0 * sqrt(rhodim) * pow(0.0001 * vdim, 3.07);
  const S qa = c0 + c1 * alphadeg + c2 * (alphadeg * alphadeg) + c3 * (alphadeg * alphadeg * alphadeg);
  f[0] = (qa * qr - Qlimit) * (1.0 / Qlimit);
}

// ------------------------------------------------------------------ cart-pole swing-up (4,1,0)
// The dynamics of the reference's second full-problem test with a known answer (asset_asrl/test/test_FullProblems/
// test_CartPole.py:11-32; l = 0.5, m1 = 1, m2 = 0.3, g = 9.81 as in :43-46): y = [q1, q2, q1', q2', t, u].
// tests/kkt_harness.py defines the same right-hand side in the product's expression DSL.
template <class S>
void cartpole(const S* y, S* f, const void*) {
  const double l = 0.5, m1 = 1.0, m2 = 0.3, g = 9.81;
  const S &q2 = y[1], &q1d = y[2], &q2d = y[3], &u = y[5];
  const S s2 = sin(q2), c2 = cos(q2);
  const S den = m1 + m2 * (1.0 - c2 * c2);
  f[0] = q1d;
  f[1] = q2d;
  f[2] = (l * m2 * s2 * (q2d * q2d) + u + m2 * g * c2 * s2) / den;
  f[3] = -1.0 * (l * m2 * c2 * s2 * (q2d * q2d) + u * c2 + (m1 * g + m2 * g) * s2) / (l * den);
}

// ------------------------------------------------------------------ free-flying robot (6,4,0)
// The dynamics of the reference's third full-problem test with a known answer (asset_asrl/test/test_FullProblems/
// test_FreeFlyingRobot.py:14-35; alpha = beta = 0.2 as in :52): y = [x, y, vx, vy, theta, omega, t, u0..u3].
// tests/kkt_harness.py defines the same right-hand side in the product's expression DSL.
template <class S>
void freeflyingrobot(const S* y, S* f, const void*) {
  const double alpha = 0.2, beta = 0.2;
  const S &theta = y[4], &omega = y[5];
  const S* u = y + 7;
  const S vscale = u[0] - u[1] + u[2] - u[3];
  f[0] = y[2];
  f[1] = y[3];
  f[2] = cos(theta) * vscale;
  f[3] = sin(theta) * vscale;
  f[4] = omega;
  f[5] = alpha * u[0] - alpha * u[1] - beta * u[2] + beta * u[3];
}

// ------------------------------------------------------------------ cannon ball with drag (4,0,1): an ODE PARAMETER (the ball's radius)
// The dynamics of the reference's two-phase full-problem test (asset_asrl/test/test_FullProblems/test_MultiPhaseCannon.py:16-72,
// non-dimensional units of :16-34): y = [v, gamma, h, r, t, rad].  cannon_energy: (E(v, rad) - E0) / 100, the muzzle-energy bound of
// :75-79,131 -- record (1, 0, 0): one output of the two inputs (v, rad).  tests/kkt_harness.py defines both in the product's DSL.
namespace cannon_units {
constexpr double g0 = 9.81, Lstar = 1000.0, Tstar = 60.0, Mstar = 10.0;
constexpr double Astar = Lstar / (Tstar * Tstar), Vstar = Lstar / Tstar, Rhostar = Mstar / (Lstar * Lstar * Lstar);
constexpr double Estar = Mstar * (Vstar * Vstar);
constexpr double CD = 0.5, RhoAir = 1.225 / Rhostar, RhoIron = 7870.0 / Rhostar, h_scale = 8.44e3 / Lstar, E0 = 400000.0 / Estar;
constexpr double g = g0 / Astar;
}  // namespace cannon_units
template <class S>
void cannon(const S* y, S* f, const void*) {
  using namespace cannon_units;
  const S &v = y[0], &gamma = y[1], &h = y[2], &rad = y[5];
  const S Sref = M_PI * (rad * rad);
  const S M = (4.0 / 3.0) * (M_PI * RhoIron) * (rad * rad * rad);
  const S rho = RhoAir * exp(-1.0 * h / h_scale);
  const S D = (0.5 * CD) * rho * (v * v) * Sref;
  f[0] = -1.0 * D / M - g * sin(gamma);
  f[1] = -1.0 * g * cos(gamma) / v;
  f[2] = v * sin(gamma);
  f[3] = v * cos(gamma);
}
template <class S>
void cannon_energy(const S* y, S* f, const void*) {
  using namespace cannon_units;
  const S &v = y[0], &rad = y[1];
  const S M = (4.0 / 3.0) * (M_PI * RhoIron) * (rad * rad * rad);
  f[0] = (0.5 * M * (v * v) - E0) * 0.01;
}

// ------------------------------------------------------------------ integrands (one output) for the segment quadrature
// quad2: I(x0, x1) = x1^2 + x0 (the integrand of tests/test_gpu_function.py);  record (xv, uv, pv) = (1, 0, 0): 2 inputs.
// powp: I(x0, x1, x2, p) = p x0^2 + sin(x1) x2 + exp(-x0 x2) / (1 + p^2): three node values and a phase parameter;
// record (1, 2, 0): 4 inputs.
// pairprod: c(a0, a1, b0, b1) = a0 b0 - a1 b1 - 0.5 -- the pair-wise path inequality of the function tests; record (1, 2, 0)
template <class S>
void pairprod(const S* y, S* f, const void*) { f[0] = y[0] * y[2] - y[1] * y[3] - 0.5; }
template <class S>
void integrand_quad2(const S* y, S* f, const void*) { f[0] = y[1] * y[1] + y[0]; }
// usq: I(x0, x1) = x1^2 -- the control effort u^2 of the cart-pole problem (test_CartPole.py:69), taken over the node values
// (q1, u): record (1, 0, 0), 2 inputs, the first unused
template <class S>
void integrand_usq(const S* y, S* f, const void*) { f[0] = y[1] * y[1]; }
// sum4: I(u0..u3) = u0 + u1 + u2 + u3 -- the thruster effort of the free-flying robot (test_FreeFlyingRobot.py:37-41,75);
// record (1, 2, 0): 4 inputs
template <class S>
void integrand_sum4(const S* y, S* f, const void*) { f[0] = y[0] + y[1] + y[2] + y[3]; }
// lq: I(x, u) = u^2 + x u + 1.25 x^2, the running cost of the reference's scaling test (asset_asrl/test/test_AutoScaling/
// test_ObjScaling.py:23-27); lq_pi: the same times pi (ODE.obj() * iscale, :77).  Records (1, 0, 0).  lq1: its ODE x' = x / 2 + u (:11-21).
template <class S>
void integrand_lq(const S* y, S* f, const void*) { f[0] = y[1] * y[1] + y[0] * y[1] + 1.25 * (y[0] * y[0]); }
template <class S>
void integrand_lq_pi(const S* y, S* f, const void*) { f[0] = M_PI * (y[1] * y[1] + y[0] * y[1] + 1.25 * (y[0] * y[0])); }
template <class S>
void lq1(const S* y, S* f, const void*) { f[0] = 0.5 * y[0] + y[2]; }
template <class S>
void integrand_powp(const S* y, S* f, const void*) {
  f[0] = y[3] * y[0] * y[0] + sin(y[1]) * y[2] + exp(-(y[0] * y[2])) / (1.0 + y[3] * y[3]);
}

// ------------------------------------------------------------------ shuttle reentry (5,2,0)
template <class S>
void reentry(const S* y, S* f, const void*) {
  const double g0 = 32.2, W = 203000.0;
  const double Lstar = 100000.0, Tstar = 60.0, Mstar = W / g0;
  const double Rhostar = Mstar / (Lstar * Lstar * Lstar);
  const double Mustar = (Lstar * Lstar * Lstar) / (Tstar * Tstar);
  const double Re = 20902900.0 / Lstar;
  const double Sref = 2690.0 / (Lstar * Lstar);
  const double m = (W / g0) / Mstar;
  const double mu = 0.140765e17 / Mustar;
  const double rho0 = 0.002378 / Rhostar;
  const double h_ref = 23800.0 / Lstar;
  const double a0 = -0.20704, a1 = 0.029244;
  const double b0 = 0.07854, b1 = -0.61592e-2, b2 = 0.621408e-3;

  const S &h = y[0], &theta = y[1], &v = y[2], &gamma = y[3], &psi = y[4];
  const S &alpha = y[6], &beta = y[7];

  S alphadeg = (180.0 / M_PI) * alpha;
  S CL = a0 + a1 * alphadeg;
  S CD = b0 + b1 * alphadeg + b2 * (alphadeg * alphadeg);
  S rho = rho0 * exp(-h / h_ref);
  S r = h + Re;
  S L = 0.5 * CL * Sref * rho * (v * v);
  S D = 0.5 * CD * Sref * rho * (v * v);
  S g = mu / (r * r);
  S sgam = sin(gamma), cgam = cos(gamma);
  S sbet = sin(beta), cbet = cos(beta);
  S spsi = sin(psi), cpsi = cos(psi);
  S tantheta = tan(theta);

  f[0] = v * sgam;
  f[1] = (v / r) * cgam * cpsi;
  f[2] = -D / m - g * sgam;
  f[3] = (L / (m * v)) * cbet + cgam * (v / r - g / v);
  f[4] = L * sbet / (m * v * cgam) + (v / r) * cgam * spsi * tantheta;
}

// ------------------------------------------------------------------ two-body + low-thrust (6,3,0)
template <class S>
void twobody_lt(const S* y, S* f, const void*) {
  const double P1mu = 1.0, ltacc = 0.01;
  S r2 = y[0] * y[0] + y[1] * y[1] + y[2] * y[2];
  S rn = sqrt(r2);
  S r3 = rn * rn * rn;
  for (int i = 0; i < 3; i++) {
    f[i] = y[3 + i];
    f[3 + i] = (y[i] / r3) * (-P1mu) + y[7 + i] * ltacc;
  }
}

// ------------------------------------------------------------------ Betts low thrust MEE (7,3,1)
struct BettsConsts {
  double Re, mu, Thrust, Isp, gs, J2, J3, J4;
  BettsConsts() {
    const double g0 = 32.174, W = 1.0, mu_e = 1.407645794e16, Lstar = 20925662.73;
    const double Tstar = Lstar / std::sqrt(mu_e / Lstar);
    const double Mstar = W / g0;
    const double Fstar = Mstar * Lstar / (Tstar * Tstar);
    const double Astar = Lstar / (Tstar * Tstar);
    const double Mustar = (Lstar * Lstar * Lstar) / (Tstar * Tstar);
    Re = 20925662.73 / Lstar;
    mu = mu_e / Mustar;
    Thrust = 4.446618e-3 / Fstar;
    Isp = 450.0 / Tstar;
    gs = g0 / Astar;
    J2 = 1082.639e-6;
    J3 = -2.565e-6;
    J4 = -1.608e-6;
  }
};

template <class S>
static inline S norm3(const S* a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
template <class S>
static inline void cross3(const S* a, const S* b, S* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
template <class S>
static inline void normalize3(const S* a, S* out) {
  S n = norm3(a);
  for (int i = 0; i < 3; i++) out[i] = a[i] / n;
}

template <class S>
void betts_lowthrust(const S* y, S* f, const void*) {
  static const BettsConsts c;
  const double mu = c.mu;
  const S &p = y[0], &ff = y[1], &g = y[2], &h = y[3], &k = y[4], &L = y[5], &ww = y[6];
  const S* Uraw = y + 8;
  const S& tau = y[11];

  // MEE -> Cartesian position/velocity (BettsLowThrust.py:212-238)
  S sinL = sin(L), cosL = cos(L);
  S sqmp = sqrt(mu / p);
  S w = 1.0 + ff * cosL + g * sinL;
  S s2 = 1.0 + h * h + k * k;
  S a2 = h * h - k * k;
  S r = p / w;
  S r_s2 = r / s2;
  S subs2 = 1.0 / s2;
  S R[3], V[3];
  R[0] = r_s2 * (cosL + a2 * cosL + 2.0 * h * k * sinL);
  R[1] = r_s2 * (sinL - a2 * sinL + 2.0 * h * k * cosL);
  R[2] = r_s2 * (2.0 * (h * sinL - k * cosL));
  S vs = -subs2 * sqmp;
  V[0] = vs * (sinL + a2 * sinL - 2.0 * h * k * cosL + g - 2.0 * ff * h * k + a2 * g);
  V[1] = vs * (-cosL + a2 * cosL + 2.0 * h * k * sinL - ff + 2.0 * g * h * k + a2 * ff);
  V[2] = vs * (-2.0 * (h * cosL + k * sinL + ff * h + g * k));

  // zonal gravity J2..J4 in RTN (BettsLowThrust.py:250-304)
  S rn = norm3(R);
  S Ir[3];
  normalize3(R, Ir);
  S IrN = Ir[2];  // Ir . North, North = (0,0,1)
  S Inraw[3] = {0.0 - Ir[0] * IrN, 0.0 - Ir[1] * IrN, 1.0 - Ir[2] * IrN};
  S In[3];
  normalize3(Inraw, In);
  S sphi = Ir[2];
  S cphi = sqrt(1.0 - sphi * sphi);
  S sphi2 = sphi * sphi, sphi3 = sphi2 * sphi, sphi4 = sphi2 * sphi2;
  S P2 = 0.5 * (3.0 * sphi2 - 1.0);
  S P3 = 0.5 * (5.0 * sphi3 - 3.0 * sphi);
  S P4 = (35.0 / 8.0) * sphi4 - (30.0 / 8.0) * sphi2 + 3.0 / 8.0;
  S D2 = 3.0 * sphi;
  S D3 = 0.5 * (15.0 * sphi2 - 3.0);
  S D4 = (35.0 / 2.0) * sphi3 - (30.0 / 4.0) * sphi;
  S Rr = c.Re / rn;
  S Rr2 = Rr * Rr, Rr3 = Rr2 * Rr, Rr4 = Rr2 * Rr2;
  S gn = (D2 * c.J2 * Rr2 + D3 * c.J3 * Rr3 + D4 * c.J4 * Rr4) * cphi;
  S gr = (3.0 * P2 * c.J2) * Rr2 + (4.0 * P3 * c.J3) * Rr3 + (5.0 * P4 * c.J4) * Rr4;
  S gsc = -mu / (R[0] * R[0] + R[1] * R[1] + R[2] * R[2]);
  S Gcart[3];
  for (int i = 0; i < 3; i++) Gcart[i] = (gn * In[i] - gr * Ir[i]) * gsc;
  // RTN basis rows: Rhat, That = (Nhat x R)^, Nhat = (R x V)^
  S Nraw[3], Nhat[3], Traw[3], That[3];
  cross3(R, V, Nraw);
  normalize3(Nraw, Nhat);
  cross3(Nhat, R, Traw);
  normalize3(Traw, That);
  S accJ[3];
  accJ[0] = Ir[0] * Gcart[0] + Ir[1] * Gcart[1] + Ir[2] * Gcart[2];
  accJ[1] = That[0] * Gcart[0] + That[1] * Gcart[1] + That[2] * Gcart[2];
  accJ[2] = Nhat[0] * Gcart[0] + Nhat[1] * Gcart[1] + Nhat[2] * Gcart[2];

  // thrust acceleration (BettsLowThrust.py:381-387)
  S Uhat[3];
  normalize3(Uraw, Uhat);
  S thr = c.gs * c.Thrust * (1.0 + 0.01 * tau);
  S ur = thr * Uhat[0] / ww + accJ[0];
  S ut = thr * Uhat[1] / ww + accJ[1];
  S un = thr * Uhat[2] / ww + accJ[2];

  // MEE dynamics (BettsLowThrust.py:337-363)
  S sqp = sqrt(p) / std::sqrt(mu);
  S hs = h * sinL - k * cosL;
  f[0] = (2.0 * (p / w) * ut) * sqp;
  f[1] = (ur * sinL + ((w + 1.0) * cosL + ff) * (ut / w) - hs * (g * un / w)) * sqp;
  f[2] = (-ur * cosL + ((w + 1.0) * sinL + g) * (ut / w) + hs * (ff * un / w)) * sqp;
  S hk = (s2 * un / w) / 2.0;
  f[3] = (cosL * hk) * sqp;
  f[4] = (sinL * hk) * sqp;
  f[5] = (mu * (w / p) * (w / p) + (1.0 / w) * hs * un) * sqp;
  f[6] = S(-c.Thrust) * (1.0 + 0.01 * tau) / c.Isp;
}

// ------------------------------------------------------------------ synthetic-32 (32,0,0)
// Coefficients a,b,c ~ U(0.5,1.5) are supplied by the caller (ctx -> 3*32 doubles) so that the
// python test harness and the product share one numpy.random.default_rng(32) draw.
template <class S>
void synthetic32(const S* y, S* f, const void* ctx) {
  const int n = 32;
  const double* a = static_cast<const double*>(ctx);
  const double* b = a + n;
  const double* c = b + n;
  const S& t = y[n];
  S ct = cos(t);
  for (int k = 0; k < n; k++) {
    f[k] = (-a[k]) * y[k] + b[k] * sin(y[(k + 1) % n]) * y[(k + 5) % n] + c[k] * ct;
  }
}

}  // namespace oracle_odes
