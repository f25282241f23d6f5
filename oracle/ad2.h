// ORACLE (test infrastructure, never shipped or imported by the product path).
//
// Second-order forward-mode AD scalar: every value carries its full gradient and Hessian with
// respect to N independent inputs.  This is how the oracle obtains df/dy and lam^T d2f/dy2 of a
// user ODE *independently* of the product's symbolic code generator (asset_asrl_amd/vf/codegen.py):
// the ODE right-hand sides in odes.h are written once as templates over the scalar type and
// instantiated with double (value) and AD2<N> (derivatives).  It stands in for the reference's
// analytic per-node chain rules (/root/reference/src/VectorFunctions/CommonFunctions/
// NestedFunction.h:140-270, CwiseOperators.h) -- same calculus, exact to round-off, no shared code.
#pragma once
#include <cmath>

template <int N>
struct AD2 {
  double v;
  double g[N];
  double h[N * N];  // full symmetric storage, h[i*N+j]

  AD2() : v(0.0) { clear(); }
  AD2(double c) : v(c) { clear(); }  // NOLINT(implicit): constants promote
  void clear() {
    for (int i = 0; i < N; i++) g[i] = 0.0;
    for (int i = 0; i < N * N; i++) h[i] = 0.0;
  }
  static AD2 variable(double val, int idx) {
    AD2 r(val);
    r.g[idx] = 1.0;
    return r;
  }
};

// z = F(a):  z' = F' a',  z'' = F' a'' + F'' a' a'^T
template <int N>
static inline AD2<N> ad2_unary(const AD2<N>& a, double v, double d1, double d2) {
  AD2<N> z;
  z.v = v;
  for (int i = 0; i < N; i++) z.g[i] = d1 * a.g[i];
  for (int i = 0; i < N; i++) {
    const double d2gi = d2 * a.g[i];
    for (int j = 0; j <= i; j++) {
      const double t = d1 * a.h[i * N + j] + d2gi * a.g[j];
      z.h[i * N + j] = t;
      z.h[j * N + i] = t;
    }
  }
  return z;
}

// z = F(a,b) with partials fa, fb, faa, fab, fbb
template <int N>
static inline AD2<N> ad2_binary(const AD2<N>& a, const AD2<N>& b, double v, double fa, double fb, double faa,
                                double fab, double fbb) {
  AD2<N> z;
  z.v = v;
  for (int i = 0; i < N; i++) z.g[i] = fa * a.g[i] + fb * b.g[i];
  for (int i = 0; i < N; i++) {
    for (int j = 0; j <= i; j++) {
      const double t = fa * a.h[i * N + j] + fb * b.h[i * N + j] + faa * a.g[i] * a.g[j] +
                       fab * (a.g[i] * b.g[j] + b.g[i] * a.g[j]) + fbb * b.g[i] * b.g[j];
      z.h[i * N + j] = t;
      z.h[j * N + i] = t;
    }
  }
  return z;
}

template <int N> static inline AD2<N> operator+(const AD2<N>& a, const AD2<N>& b) { return ad2_binary(a, b, a.v + b.v, 1.0, 1.0, 0.0, 0.0, 0.0); }
template <int N> static inline AD2<N> operator-(const AD2<N>& a, const AD2<N>& b) { return ad2_binary(a, b, a.v - b.v, 1.0, -1.0, 0.0, 0.0, 0.0); }
template <int N> static inline AD2<N> operator*(const AD2<N>& a, const AD2<N>& b) { return ad2_binary(a, b, a.v * b.v, b.v, a.v, 0.0, 1.0, 0.0); }
template <int N> static inline AD2<N> operator/(const AD2<N>& a, const AD2<N>& b) {
  const double q = a.v / b.v, ib = 1.0 / b.v;
  return ad2_binary(a, b, q, ib, -q * ib, 0.0, -ib * ib, 2.0 * q * ib * ib);
}
template <int N> static inline AD2<N> operator-(const AD2<N>& a) { return ad2_unary(a, -a.v, -1.0, 0.0); }
template <int N> static inline AD2<N> operator+(const AD2<N>& a, double c) { return ad2_unary(a, a.v + c, 1.0, 0.0); }
template <int N> static inline AD2<N> operator+(double c, const AD2<N>& a) { return ad2_unary(a, c + a.v, 1.0, 0.0); }
template <int N> static inline AD2<N> operator-(const AD2<N>& a, double c) { return ad2_unary(a, a.v - c, 1.0, 0.0); }
template <int N> static inline AD2<N> operator-(double c, const AD2<N>& a) { return ad2_unary(a, c - a.v, -1.0, 0.0); }
template <int N> static inline AD2<N> operator*(const AD2<N>& a, double c) { return ad2_unary(a, a.v * c, c, 0.0); }
template <int N> static inline AD2<N> operator*(double c, const AD2<N>& a) { return ad2_unary(a, c * a.v, c, 0.0); }
template <int N> static inline AD2<N> operator/(const AD2<N>& a, double c) { return ad2_unary(a, a.v / c, 1.0 / c, 0.0); }
template <int N> static inline AD2<N> operator/(double c, const AD2<N>& a) {
  const double q = c / a.v;
  return ad2_unary(a, q, -q / a.v, 2.0 * q / (a.v * a.v));
}

template <int N> static inline AD2<N> sin(const AD2<N>& a) { const double s = std::sin(a.v), c = std::cos(a.v); return ad2_unary(a, s, c, -s); }
template <int N> static inline AD2<N> cos(const AD2<N>& a) { const double s = std::sin(a.v), c = std::cos(a.v); return ad2_unary(a, c, -s, -c); }
template <int N> static inline AD2<N> tan(const AD2<N>& a) { const double t = std::tan(a.v), d = 1.0 + t * t; return ad2_unary(a, t, d, 2.0 * t * d); }
template <int N> static inline AD2<N> exp(const AD2<N>& a) { const double e = std::exp(a.v); return ad2_unary(a, e, e, e); }
template <int N> static inline AD2<N> log(const AD2<N>& a) { return ad2_unary(a, std::log(a.v), 1.0 / a.v, -1.0 / (a.v * a.v)); }
template <int N> static inline AD2<N> sqrt(const AD2<N>& a) { const double r = std::sqrt(a.v); return ad2_unary(a, r, 0.5 / r, -0.25 / (r * a.v)); }
template <int N> static inline AD2<N> tanh(const AD2<N>& a) { const double t = std::tanh(a.v), d = 1.0 - t * t; return ad2_unary(a, t, d, -2.0 * t * d); }
template <int N> static inline AD2<N> pow(const AD2<N>& a, double c) {
  const double p = std::pow(a.v, c);
  return ad2_unary(a, p, c * p / a.v, c * (c - 1.0) * p / (a.v * a.v));
}
template <int N> static inline AD2<N> powi(const AD2<N>& a, int k) {
  const double p2 = (k >= 2) ? std::pow(a.v, k - 2) : 0.0;
  const double p1 = (k >= 2) ? p2 * a.v : 1.0;
  return ad2_unary(a, p1 * a.v, k * p1, (k >= 2) ? k * (k - 1.0) * p2 : 0.0);
}
static inline double powi(double a, int k) { return std::pow(a, k); }
