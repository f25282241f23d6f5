// ORACLE (test infrastructure): the other per-segment functions a phase registers beside the defects -- restated from
//   /root/reference/src/OptimalControl/MeshSpacingConstraints.h:8-98    SingleMeshSpacing
//   /root/reference/src/OptimalControl/MeshSpacingConstraints.h:101-193 LGLMeshSpacing<CSC>
//   /root/reference/src/OptimalControl/LGLControlSplines.h:64-315       LGLControlSpline<CSC,USZ,Order>
//   /root/reference/src/OptimalControl/LGLIntegrals.h:9-52              LGLIntegral (reduced quadrature of an integrand)
// value, Jacobian, adjoint gradient J^T lam and adjoint Hessian, in the closed forms the reference writes (the integral
// is an expression template there: its derivatives are the chain rule over the value definition, restated here with
// the integrand's own value / gradient / Hessian).  jx: orows x irows column-major; hx: irows x irows column-major.
#include <cmath>
#include <cstring>
#include <vector>

#include "lgl_coeffs.h"
#include "oracle.h"

namespace {
inline double& at(double* m, int ld, int r, int c) { return m[size_t(c) * ld + r]; }
void adjgrad_from_jac(int orr, int irr, const double* jx, const double* lam, double* agx) {
  for (int c = 0; c < irr; c++) {
    double s = 0.0;
    for (int r = 0; r < orr; r++) s += lam[r] * jx[size_t(c) * orr + r];
    agx[c] = s;
  }
}
}  // namespace

extern "C" {

// MeshSpacingConstraints.h:33-41 (value), :52-60 (jacobian), :88-96 (no second derivatives: linear)
int oracle_single_mesh_spacing_all(double cardinal_spacing, double scale, const double* x, const double* lam, double* fx,
                                   double* jx, double* agx, double* hx) {
  const double h = x[2] - x[0];
  fx[0] = (cardinal_spacing * h - (x[1] - x[0])) * scale;
  if (jx) {
    jx[0] = (1.0 - cardinal_spacing) * scale;
    jx[1] = -1.0 * scale;
    jx[2] = cardinal_spacing * scale;
    if (agx && lam) adjgrad_from_jac(1, 3, jx, lam, agx);
  }
  if (hx) std::memset(hx, 0, 9 * sizeof(double));
  return 0;
}

// MeshSpacingConstraints.h:118-126 (value), :136-146 (jacobian), :176-191 (adjoint Hessian)
int oracle_lgl_mesh_spacing_all(int cs, const double* x, const double* lam, double* fx, double* jx, double* agx,
                                double* hx) {
  if (cs < 3 || cs > 4) return -1;
  const double* tc = oracle_lgl_table(cs, "tc");
  const int orr = cs - 2, irr = cs;
  const double h = x[cs - 1] - x[0], h2 = h * h, h3 = h2 * h;
  if (jx) std::memset(jx, 0, sizeof(double) * orr * irr);
  if (hx) std::memset(hx, 0, sizeof(double) * irr * irr);
  for (int i = 0; i < orr; i++) {
    fx[i] = tc[i + 1] - (x[1 + i] - x[0]) / h;
    if (jx) {
      at(jx, orr, i, i + 1) = -1.0 / h;
      at(jx, orr, i, 0) = 1.0 / h - (x[1 + i] - x[0]) / h2;
      at(jx, orr, i, cs - 1) = (x[1 + i] - x[0]) / h2;
    }
    if (hx && lam) {
      const double l = lam[i], d = x[1 + i] - x[0];
      at(hx, irr, 0, i + 1) += -l * 1.0 / h2;
      at(hx, irr, i + 1, 0) += -l * 1.0 / h2;
      at(hx, irr, cs - 1, i + 1) += l * 1.0 / h2;
      at(hx, irr, i + 1, cs - 1) += l * 1.0 / h2;
      at(hx, irr, 0, 0) += l * 2.0 / h2 - l * 2.0 * d / h3;
      at(hx, irr, cs - 1, cs - 1) += -l * 2.0 * d / h3;
      at(hx, irr, 0, cs - 1) += l * (2.0 * d / h3 - 1.0 / h2);
      at(hx, irr, cs - 1, 0) += l * (2.0 * d / h3 - 1.0 / h2);
    }
  }
  if (jx && agx && lam) adjgrad_from_jac(orr, irr, jx, lam, agx);
  return 0;
}

// LGLControlSplines.h:92-108 (value), :130-162 (jacobian), :199-309 (adjoint Hessian).  Input: 2*cs-1 blocks [t, u(usize)].
int oracle_control_spline_all(int cs, int usize, int order, const double* x, const double* lam, double* fx, double* jx,
                              double* agx, double* hx) {
  if (cs < 3 || cs > 4 || usize < 1) return -1;
  if (order <= 0) order = cs - 2;
  const double(*uone)[4] = oracle_uone_spline_weights(cs);
  const double(*uzero)[4] = oracle_uzero_spline_weights(cs);
  const int tu = usize + 1, tunum = 2 * cs - 1, irr = tunum * tu, orr = usize * order;
  auto T = [&](int i) { return x[i * tu]; };
  auto U = [&](int i, int k) { return x[i * tu + 1 + k]; };
  const double h0 = T(cs - 1) - T(0), h1 = T(tunum - 1) - T(cs - 1);
  std::memset(fx, 0, sizeof(double) * orr);
  if (jx) std::memset(jx, 0, sizeof(double) * orr * irr);
  if (hx) std::memset(hx, 0, sizeof(double) * irr * irr);
  const int c0 = 0, cm = (cs - 1) * tu, cf = (2 * cs - 2) * tu;
  for (int j = 0; j < order; j++) {
    const double h0pow = 1.0 / std::pow(h0, double(j + 1)), h1pow = 1.0 / std::pow(h1, double(j + 1));
    const double hdt = double(j + 1), h2dt = double(j + 2);
    double OTH = 0.0, ZTH = 0.0;
    for (int i = 0; i < cs; i++) {
      const double wo = uone[j][i], wz = uzero[j][i];
      for (int k = 0; k < usize; k++) {
        const int row = j * usize + k;
        fx[row] += (wo * h0pow) * U(i, k) - (wz * h1pow) * U(i + cs - 1, k);
        if (jx) {
          at(jx, orr, row, c0) += (wo * hdt * h0pow / h0) * U(i, k);
          at(jx, orr, row, cm) += -(wo * hdt * h0pow / h0) * U(i, k) - (wz * hdt * h1pow / h1) * U(i + cs - 1, k);
          at(jx, orr, row, cf) += (wz * hdt * h1pow / h1) * U(i + cs - 1, k);
          at(jx, orr, row, i * tu + 1 + k) += wo * h0pow;
          at(jx, orr, row, (i + cs - 1) * tu + 1 + k) += -wz * h1pow;
        }
        if (hx && lam) {
          const double l = lam[row];
          OTH += (wo * hdt * h2dt * h0pow / (h0 * h0)) * U(i, k) * l;
          ZTH += (wz * hdt * h2dt * h1pow / (h1 * h1)) * U(i + cs - 1, k) * l;
          const double a = l * (wo * hdt * h0pow / h0), b = l * (wz * hdt * h1pow / h1);
          const int cu0 = i * tu + 1 + k, cu1 = (i + cs - 1) * tu + 1 + k;
          at(hx, irr, c0, cu0) += a;
          at(hx, irr, cm, cu0) -= a;
          at(hx, irr, cu0, c0) += a;
          at(hx, irr, cu0, cm) -= a;
          at(hx, irr, cm, cu1) -= b;
          at(hx, irr, cf, cu1) += b;
          at(hx, irr, cu1, cm) -= b;
          at(hx, irr, cu1, cf) += b;
        }
      }
    }
    if (hx && lam) {
      at(hx, irr, c0, c0) += OTH;
      at(hx, irr, cm, cm) += OTH;
      at(hx, irr, c0, cm) += -OTH;
      at(hx, irr, cm, c0) += -OTH;
      at(hx, irr, cm, cm) -= ZTH;
      at(hx, irr, cf, cf) -= ZTH;
      at(hx, irr, cf, cm) += ZTH;
      at(hx, irr, cm, cf) += ZTH;
    }
  }
  if (jx && agx && lam) adjgrad_from_jac(orr, irr, jx, lam, agx);
  return 0;
}

// LGLIntegrals.h:18-52: inputs [x_0(xv), t_0, ..., x_{cs-1}(xv), t_{cs-1}, p(pv)]; value h * sum_i w_i I([x_i, p]) with
// w = Reduced_Integral_Weights and h = t_{cs-1} - t_0.  `integrand`: one output of xv + pv inputs (its oracle_ode
// record has xv = 1 output and nin = 1 + 1 + uv + pv = xv_arg + pv_arg inputs).
int oracle_lgl_integral_all(const oracle_ode* integrand, int cs, int xv, int pv, const double* x, const double* lam,
                            double* fx, double* jx, double* agx, double* hx) {
  if (cs < 2 || cs > 4 || !integrand) return -1;
  const int nin = integrand->xv + 1 + integrand->uv + integrand->pv;
  if (integrand->xv != 1 || nin != xv + pv) return -2;
  const double* w = oracle_reduced_integral_weights(cs);
  const int xtv = xv + 1, irr = cs * xtv + pv, t0c = xv, tfc = (cs - 1) * xtv + xv, p0 = cs * xtv;
  const double h = x[tfc] - x[t0c];
  if (jx) std::memset(jx, 0, sizeof(double) * irr);
  if (hx) std::memset(hx, 0, sizeof(double) * irr * irr);
  const double one = 1.0;
  double sum = 0.0;
  std::vector<double> y(nin), J(nin), g(nin), H(size_t(nin) * nin);
  auto col = [&](int i, int a) { return a < xv ? i * xtv + a : p0 + (a - xv); };   // where integrand input a of node i lives
  const double l = lam ? lam[0] : 0.0;
  for (int i = 0; i < cs; i++) {
    for (int a = 0; a < xv; a++) y[a] = x[i * xtv + a];
    for (int a = 0; a < pv; a++) y[xv + a] = x[p0 + a];
    double fi = 0.0;
    integrand->fjgh(y.data(), &one, &fi, J.data(), g.data(), H.data(), integrand->ctx);
    sum += w[i] * fi;
    if (jx)
      for (int a = 0; a < nin; a++) jx[col(i, a)] += h * w[i] * J[a];
    if (hx && lam)
      for (int a = 0; a < nin; a++) {
        for (int b = 0; b < nin; b++) at(hx, irr, col(i, a), col(i, b)) += l * h * w[i] * H[size_t(a) * nin + b];
        const double d = l * w[i] * J[a];                  // d2 / d(input a) d(t_f) = + w_i dI/da ; d(t_0): minus
        at(hx, irr, col(i, a), tfc) += d;
        at(hx, irr, tfc, col(i, a)) += d;
        at(hx, irr, col(i, a), t0c) -= d;
        at(hx, irr, t0c, col(i, a)) -= d;
      }
  }
  fx[0] = h * sum;
  if (jx) {
    jx[tfc] += sum;
    jx[t0c] -= sum;
    if (agx && lam) adjgrad_from_jac(1, irr, jx, lam, agx);
  }
  return 0;
}

const double* oracle_aux_table(int cs, const char* which) {
  if (!which) return nullptr;
  if (!std::strcmp(which, "reduced_integral")) return oracle_reduced_integral_weights(cs);
  if (cs != 3 && cs != 4) return nullptr;
  if (!std::strcmp(which, "uzero0")) return oracle_uzero_spline_weights(cs)[0];
  if (!std::strcmp(which, "uzero1")) return oracle_uzero_spline_weights(cs)[1];
  if (!std::strcmp(which, "uone0")) return oracle_uone_spline_weights(cs)[0];
  if (!std::strcmp(which, "uone1")) return oracle_uone_spline_weights(cs)[1];
  return nullptr;
}

}  // extern "C"
