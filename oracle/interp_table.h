// ORACLE (test infrastructure, never shipped or imported by the product path).
//
// Tabulated data interpolated along one axis -- a restatement of the reference's InterpTable1D
// (/root/reference/src/VectorFunctions/CommonFunctions/InterpTable1D.h): nodal slopes of the cubic kind from five-point
// differences (:134-179), the element look-up (:181-197), value / first / second derivative of the Hermite cubic and of the
// linear kind written out as the reference writes them (:199-267), and the scalar-type overloads InterpFunction1D supplies
// (:337-400): for the AD2 scalar the chain rule through the table's argument.  The product's own table (vf/functions.py:
// InterpTable1D) builds the interpolant as an expression and differentiates it symbolically -- a different route to the same numbers.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "ad2.h"

namespace oracle_odes {

struct OTable {
  int n = 0, vlen = 0;
  bool cubic = true, even = true;
  std::vector<double> ts, vs, ds;     // vs, ds: [vlen][n]

  // InterpTable1D.h:80-131
  void set(const std::vector<double>& t, const std::vector<double>& v, int vl, bool cub) {
    ts = t, vs = v, vlen = vl, cubic = cub, n = static_cast<int>(t.size());
    const double total = ts[n - 1] - ts[0];
    double terr = 0.0;
    for (int i = 0; i < n; i++) terr = std::max(terr, std::fabs(ts[i] - (ts[0] + total * i / (n - 1))));
    even = !(terr > std::fabs(total) * 1.0e-12);
    ds.assign(vs.size(), 0.0);
    if (cubic) slopes();
  }

  // InterpTable1D.h:134-179: weights w with sum_k w_k s_k^m = [m == 1], s_k the offsets of the five abscissae in units of the local step
  void slopes() {
    for (int i = 0; i < n; i++) {
      int start;
      if (i + 2 <= n - 1 && i - 2 >= 0) start = i - 2;
      else if (i < n - 1 - i) start = 0;
      else start = n - 5;
      const int dir = (i < n - 1) ? 1 : -1;
      const double step = std::fabs(ts[i + dir] - ts[i]);
      double A[5][6];
      for (int k = 0; k < 5; k++) {
        const double s = (ts[start + k] - ts[i]) / step;
        double pw = 1.0;
        for (int m = 0; m < 5; m++) A[m][k] = pw, pw *= s;
      }
      for (int m = 0; m < 5; m++) A[m][5] = (m == 1) ? 1.0 : 0.0;
      for (int c = 0; c < 5; c++) {                      // elimination with row pivoting
        int piv = c;
        for (int r = c + 1; r < 5; r++) if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
        for (int k = 0; k < 6; k++) std::swap(A[c][k], A[piv][k]);
        for (int r = 0; r < 5; r++) {
          if (r == c) continue;
          const double m = A[r][c] / A[c][c];
          for (int k = c; k < 6; k++) A[r][k] -= m * A[c][k];
        }
      }
      for (int q = 0; q < vlen; q++) {
        double acc = 0.0;
        for (int k = 0; k < 5; k++) acc += vs[q * n + start + k] * (A[k][5] / A[k][k] / step);
        ds[q * n + i] = acc;
      }
    }
  }

  // InterpTable1D.h:181-197
  int elem(double t) const {
    int e;
    if (even) e = std::min(static_cast<int>((t - ts[0]) / (ts[1] - ts[0])), n - 2);
    else e = static_cast<int>(std::upper_bound(ts.begin(), ts.end(), t) - ts.begin()) - 1;
    return std::max(std::min(e, n - 2), 0);
  }

  // InterpTable1D.h:215-266: row q; v, dv/dt, d2v/dt2
  void interp(int q, double t, double& v, double& dv, double& d2v) const {
    const int e = elem(t);
    const double step = ts[e + 1] - ts[e], x = (t - ts[e]) / step;
    const double v0 = vs[q * n + e], v1 = vs[q * n + e + 1];
    if (cubic) {
      const double d0 = ds[q * n + e], d1 = ds[q * n + e + 1];
      const double x2 = x * x, x3 = x2 * x;
      v = v0 * (2.0 * x3 - 3.0 * x2 + 1.0) + v1 * (-2.0 * x3 + 3.0 * x2) + d0 * ((x3 - 2.0 * x2 + x) * step) + d1 * ((x3 - x2) * step);
      dv = v0 * ((6.0 * x2 - 6.0 * x) / step) + v1 * ((-6.0 * x2 + 6.0 * x) / step) + d0 * (3.0 * x2 - 4.0 * x + 1.0) + d1 * (3.0 * x2 - 2.0 * x);
      d2v = v0 * ((12.0 * x - 6.0) / (step * step)) + v1 * ((-12.0 * x + 6.0) / (step * step)) + d0 * ((6.0 * x - 4.0) / step) +
            d1 * ((6.0 * x - 2.0) / step);
    } else {
      v = v0 * (1.0 - x) + v1 * x;
      dv = (v1 - v0) / step;
      d2v = 0.0;
    }
  }
};

static inline double tab_eval(const OTable& T, int q, double t) {
  double v, dv, d2v;
  T.interp(q, t, v, dv, d2v);
  return v;
}
template <int N>
static inline AD2<N> tab_eval(const OTable& T, int q, const AD2<N>& t) {
  double v, dv, d2v;
  T.interp(q, t.v, v, dv, d2v);
  return ad2_unary(t, v, dv, d2v);
}

// The three tables of the `tabulated` ODE (odes.h): plain arithmetic, so tests/helpers.py: make_tabulated builds the same numbers.
//   0  "density":  cubic,  UNEVEN abscissae h_i = -1.5 + 0.1 i + 0.004 i^2 (i = 0..24),       one value 1.2 / (1 + 0.5 (h_i + 1)^2)
//   1  "thrust":   linear, even abscissae   t_i = -0.25 + 0.537 i        (i = 0..20),          one value 1 + 0.05 i - 0.004 i^2
//   2  "wind":     cubic,  even abscissae   s_i = -2 + 0.25 i            (i = 0..16),          two values 0.3 s^2 - 0.1 s, 1 / (2.5 + s)
static inline const OTable& tabulated_table(int which) {
  static const std::vector<OTable> tabs = [] {
    std::vector<OTable> T(3);
    std::vector<double> t, v;
    for (int i = 0; i <= 24; i++) {
      const double h = -1.5 + 0.1 * i + 0.004 * i * i;
      t.push_back(h), v.push_back(1.2 / (1.0 + 0.5 * (h + 1.0) * (h + 1.0)));
    }
    T[0].set(t, v, 1, true);
    t.clear(), v.clear();
    for (int i = 0; i <= 20; i++) t.push_back(-0.25 + 0.537 * i), v.push_back(1.0 + 0.05 * i - 0.004 * i * i);
    T[1].set(t, v, 1, false);
    t.clear(), v.clear();
    std::vector<double> v2;
    for (int i = 0; i <= 16; i++) {
      const double s = -2.0 + 0.25 * i;
      t.push_back(s), v.push_back(0.3 * s * s - 0.1 * s), v2.push_back(1.0 / (2.5 + s));
    }
    v.insert(v.end(), v2.begin(), v2.end());
    T[2].set(t, v, 2, true);
    return T;
  }();
  return tabs[which];
}

}  // namespace oracle_odes
