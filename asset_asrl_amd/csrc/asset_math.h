// Elementary functions of the generated ODE functors.
//
// sin / cos / tan go through one argument reduction and the two fdlibm polynomial kernels (k_sin.c / k_cos.c
// coefficients; < 1 ulp on [-pi/4, pi/4]).  The library sincos() carries the Payne-Hanek path for arbitrarily large
// arguments inline and branch-free (v_trig_preop_f64 ...), about 5x the instructions; ODE angles are O(1), so a
// three-term Cody-Waite reduction is used instead: its products are exact for |x| < 2^19 * pi/2 (error <= 2e-16
// absolute, checked against long double), and beyond that the absolute error grows like |x| * 2^-53, which is what one
// ulp of the argument itself moves sin(x) by.  NaN / Inf give NaN.  The ODE stage of the defect kernels is one
// instruction stream per SIMD, so instructions saved here are time saved.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define ASSET_MATH_FN __host__ __device__ inline
#else
#define ASSET_MATH_FN inline
#endif

// Scheduling fence of the generated bodies (vf/codegen.py: TRANS_FENCE): nothing moves across it, so the elementary-function
// evaluations of a body that has dozens of them run one after the other instead of all at once.
#if defined(__HIP_DEVICE_COMPILE__)
#define ASSET_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ASSET_SCHED_FENCE() ((void)0)
#endif

ASSET_MATH_FN void asset_sincos(double x, double* sp, double* cp) {
  const double fn = rint(x * 6.36619772367581382433e-01);          // x * 2/pi
  double r = fma(-fn, 1.57079632673412561417e+00, x);              // pi/2, first 33 bits: the product is exact
  r = fma(-fn, 6.07710050630396597660e-11, r);                     //       second 33 bits
  r = fma(-fn, 2.02226624879595063154e-21, r);                     //       remainder
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                              2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                              -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                               -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double s = fma(r * z, ps, r);
  const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = static_cast<int>(fn - 4.0 * floor(fn * 0.25));   // fn mod 4 without leaving the double range
  const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sp = (q & 2) ? -ss : ss;
  *cp = ((q + 1) & 2) ? -cc : cc;
}
ASSET_MATH_FN double asset_sin(double x) { double s, c; asset_sincos(x, &s, &c); return s; }
ASSET_MATH_FN double asset_cos(double x) { double s, c; asset_sincos(x, &s, &c); return c; }
ASSET_MATH_FN double asset_tan(double x) { double s, c; asset_sincos(x, &s, &c); return s / c; }

// Tabulated data in a generated body (vf.InterpTable1D; the reference's InterpTable1D::get_telem, CommonFunctions/InterpTable1D.h:181-197):
// the element of the abscissae t falls into, clamped to [0, n-2] -- by division where they are evenly spaced, by bisection elsewhere.
// Returned as a double: the generated bodies hold doubles only, and the conversions fold into the address arithmetic.
ASSET_MATH_FN double asset_tab_even(double t, double t0, double step, int n) {
  int e = static_cast<int>((t - t0) / step);
  e = e < n - 2 ? e : n - 2;
  return static_cast<double>(e > 0 ? e : 0);
}
ASSET_MATH_FN double asset_tab_find(const double* ts, int n, double t) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ts[mid] <= t) lo = mid + 1; else hi = mid;
  }
  int e = lo - 1;
  e = e < n - 2 ? e : n - 2;
  return static_cast<double>(e > 0 ? e : 0);
}
