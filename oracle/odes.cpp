// ORACLE (test infrastructure, never shipped or imported by the product path).
//
// ODE registry.  Provider 0 ("ad2") derives df/dy and lam^T d2f/dy2 from the templated right-hand
// sides in odes.h with the AD2 scalar -- independent of the product's code generator and used for
// every parity check.  Provider 1 ("gen") links the plain-C analytic derivatives printed by the
// product's generator (oracle/gen/odes_gen.c, made by oracle/gen_odes.py); it exists only so that
// bench.py's cpu_baseline is not handicapped by AD2's O(N^2)-per-operation cost, and is itself
// checked against provider 0 in tests/test_oracle.py.
#include <cstring>

#include "batch4.h"
#include "odes.h"
#include "oracle.h"

namespace {

double g_synth32[96];

template <int XV, int NIN, void (*FD)(const double*, double*, const void*),
          void (*FA)(const AD2<NIN>*, AD2<NIN>*, const void*)>
struct Ad2Provider {
  static void f(const double* y, double* fx, const void* ctx) { FD(y, fx, ctx); }
  static void run(const double* y, AD2<NIN>* out, const void* ctx) {
    AD2<NIN> in[NIN];
    for (int i = 0; i < NIN; i++) in[i] = AD2<NIN>::variable(y[i], i);
    FA(in, out, ctx);
  }
  static void fj(const double* y, double* fx, double* J, const void* ctx) {
    AD2<NIN> out[XV];
    run(y, out, ctx);
    for (int k = 0; k < XV; k++) {
      fx[k] = out[k].v;
      for (int i = 0; i < NIN; i++) J[k * NIN + i] = out[k].g[i];
    }
  }
  static void fjgh(const double* y, const double* lam, double* fx, double* J, double* g, double* H, const void* ctx) {
    AD2<NIN> out[XV];
    run(y, out, ctx);
    for (int i = 0; i < NIN; i++) g[i] = 0.0;
    for (int i = 0; i < NIN * NIN; i++) H[i] = 0.0;
    for (int k = 0; k < XV; k++) {
      fx[k] = out[k].v;
      for (int i = 0; i < NIN; i++) {
        J[k * NIN + i] = out[k].g[i];
        g[i] += lam[k] * out[k].g[i];
      }
      for (int i = 0; i < NIN * NIN; i++) H[i] += lam[k] * out[k].h[i];
    }
  }
};

#define AD2_ODE(NAME, XV, UV, PV)                                                                      \
  using P_##NAME = Ad2Provider<XV, XV + 1 + UV + PV, &oracle_odes::NAME<double>,                       \
                               &oracle_odes::NAME<AD2<XV + 1 + UV + PV>>>;

AD2_ODE(brachistochrone, 3, 1, 0)
AD2_ODE(reentry, 5, 2, 0)
AD2_ODE(twobody_lt, 6, 3, 0)
AD2_ODE(betts_lowthrust, 7, 3, 1)
AD2_ODE(synthetic32, 32, 0, 0)
AD2_ODE(vanderpol, 2, 1, 1)
AD2_ODE(switched, 2, 1, 0)
AD2_ODE(tabulated, 2, 1, 0)
AD2_ODE(coupled12, 12, 3, 2)
AD2_ODE(coupled16, 16, 3, 2)
AD2_ODE(driven14, 14, 3, 0)
AD2_ODE(driven20, 20, 3, 0)
AD2_ODE(pathcon, 2, 3, 0)
AD2_ODE(integrand_quad2, 1, 0, 0)
AD2_ODE(pairprod, 1, 2, 0)
AD2_ODE(integrand_powp, 1, 2, 0)
AD2_ODE(reentry_heating, 1, 1, 0)
AD2_ODE(cartpole, 4, 1, 0)
AD2_ODE(integrand_usq, 1, 0, 0)
AD2_ODE(freeflyingrobot, 6, 4, 0)
AD2_ODE(integrand_sum4, 1, 2, 0)
AD2_ODE(cannon, 4, 0, 1)
AD2_ODE(cannon_energy, 1, 0, 0)
AD2_ODE(integrand_lq, 1, 0, 0)
AD2_ODE(integrand_lq_pi, 1, 0, 0)
AD2_ODE(lq1, 1, 1, 0)
AD2_ODE(delta3_1, 7, 3, 0)
AD2_ODE(delta3_2, 7, 3, 0)
AD2_ODE(delta3_3, 7, 3, 0)
AD2_ODE(delta3_4, 7, 3, 0)
AD2_ODE(norm3, 1, 1, 0)
AD2_ODE(delta3_orbit, 5, 0, 0)
AD2_ODE(shape_1_0_0, 1, 0, 0)
AD2_ODE(shape_1_1_0, 1, 1, 0)
AD2_ODE(shape_2_1_0, 2, 1, 0)
AD2_ODE(shape_3_0_1, 3, 0, 1)
AD2_ODE(shape_4_4_0, 4, 4, 0)
AD2_ODE(shape_5_3_2, 5, 3, 2)
AD2_ODE(shape_6_0_0, 6, 0, 0)
AD2_ODE(shape_8_3_1, 8, 3, 1)
AD2_ODE(shape_10_4_0, 10, 4, 0)
AD2_ODE(shape_11_4_0, 11, 4, 0)

}  // namespace

// generated analytic C (optional at link time)
#define GEN_DECL(NAME)                                                                                  \
  extern "C" void ode_##NAME##_f(const double*, double*) __attribute__((weak));                         \
  extern "C" void ode_##NAME##_fj(const double*, double*, double*) __attribute__((weak));               \
  extern "C" void ode_##NAME##_fjgh(const double*, const double*, double*, double*, double*, double*)   \
      __attribute__((weak));                                                                            \
  extern "C" void ode_##NAME##_f4(const v4d*, v4d*) __attribute__((weak));                              \
  extern "C" void ode_##NAME##_fjgh4(const v4d*, const v4d*, v4d*, v4d*, v4d*, v4d*) __attribute__((weak)); \
  namespace {                                                                                           \
  void gen_##NAME##_f(const double* y, double* f, const void*) { ode_##NAME##_f(y, f); }                \
  void gen_##NAME##_fj(const double* y, double* f, double* J, const void*) { ode_##NAME##_fj(y, f, J); } \
  void gen_##NAME##_fjgh(const double* y, const double* l, double* f, double* J, double* g, double* H,  \
                         const void*) {                                                                 \
    ode_##NAME##_fjgh(y, l, f, J, g, H);                                                                \
  }                                                                                                     \
  }
GEN_DECL(brachistochrone)
GEN_DECL(reentry)
GEN_DECL(twobody_lt)
GEN_DECL(betts_lowthrust)
GEN_DECL(synthetic32)
GEN_DECL(vanderpol)
GEN_DECL(switched)
GEN_DECL(tabulated)
GEN_DECL(coupled12)
GEN_DECL(coupled16)
GEN_DECL(driven14)
GEN_DECL(driven20)
GEN_DECL(pathcon)
GEN_DECL(integrand_quad2)
GEN_DECL(pairprod)
GEN_DECL(integrand_powp)
GEN_DECL(reentry_heating)
GEN_DECL(cartpole)
GEN_DECL(integrand_usq)
GEN_DECL(freeflyingrobot)
GEN_DECL(integrand_sum4)
GEN_DECL(cannon)
GEN_DECL(cannon_energy)
GEN_DECL(integrand_lq)
GEN_DECL(integrand_lq_pi)
GEN_DECL(lq1)
GEN_DECL(delta3_1)
GEN_DECL(delta3_2)
GEN_DECL(delta3_3)
GEN_DECL(delta3_4)
GEN_DECL(norm3)
GEN_DECL(delta3_orbit)
GEN_DECL(shape_1_0_0)
GEN_DECL(shape_1_1_0)
GEN_DECL(shape_2_1_0)
GEN_DECL(shape_3_0_1)
GEN_DECL(shape_4_4_0)
GEN_DECL(shape_5_3_2)
GEN_DECL(shape_6_0_0)
GEN_DECL(shape_8_3_1)
GEN_DECL(shape_10_4_0)
GEN_DECL(shape_11_4_0)

extern "C" {

void oracle_set_synthetic32(const double* abc) { std::memcpy(g_synth32, abc, sizeof g_synth32); }

// the restated InterpTable1D on the tables of the `tabulated` ODE (interp_table.h): sizes, arrays, and one evaluation
int oracle_table_sizes(int which, int* n, int* vlen, int* even, int* cubic) {
  if (which < 0 || which > 2) return -1;
  const oracle_odes::OTable& T = oracle_odes::tabulated_table(which);
  *n = T.n, *vlen = T.vlen, *even = T.even, *cubic = T.cubic;
  return 0;
}
int oracle_table_data(int which, double* ts, double* vs, double* ds) {
  if (which < 0 || which > 2) return -1;
  const oracle_odes::OTable& T = oracle_odes::tabulated_table(which);
  std::memcpy(ts, T.ts.data(), sizeof(double) * T.n);
  std::memcpy(vs, T.vs.data(), sizeof(double) * T.n * T.vlen);
  std::memcpy(ds, T.ds.data(), sizeof(double) * T.n * T.vlen);
  return 0;
}
int oracle_table_interp(int which, double t, double* v, double* dv, double* d2v) {
  if (which < 0 || which > 2) return -1;
  const oracle_odes::OTable& T = oracle_odes::tabulated_table(which);
  for (int q = 0; q < T.vlen; q++) T.interp(q, t, v[q], dv[q], d2v[q]);
  return 0;
}

#define TRY(NAME, XV, UV, PV, CTX)                                                  \
  if (!std::strcmp(name, #NAME)) {                                                  \
    out->xv = XV, out->uv = UV, out->pv = PV, out->ctx = CTX;                       \
    if (provider == 0) {                                                            \
      out->f = &P_##NAME::f, out->fj = &P_##NAME::fj, out->fjgh = &P_##NAME::fjgh;  \
      return 0;                                                                     \
    }                                                                               \
    if (provider == 1 && ode_##NAME##_fjgh) {                                       \
      out->f = &gen_##NAME##_f, out->fj = &gen_##NAME##_fj, out->fjgh = &gen_##NAME##_fjgh; \
      return 0;                                                                     \
    }                                                                               \
    return -2;                                                                      \
  }

}  // extern "C"

// four-wide twin of a generated registry entry (batch4.h)
#define TRY4(NAME)                                                                  \
  if (ode->fjgh == &gen_##NAME##_fjgh && ode_##NAME##_fjgh4 && ode_##NAME##_f4) {   \
    out->f = &ode_##NAME##_f4, out->fjgh = &ode_##NAME##_fjgh4;                     \
    return 0;                                                                       \
  }
int oracle_get_ode4(const oracle_ode* ode, oracle_ode4* out) {
  TRY4(brachistochrone)
  TRY4(reentry)
  TRY4(twobody_lt)
  TRY4(betts_lowthrust)
  TRY4(synthetic32)
  TRY4(vanderpol)
  TRY4(switched)
  TRY4(tabulated)
  TRY4(coupled12)
  TRY4(coupled16)
  TRY4(driven14)
  TRY4(driven20)
  TRY4(pathcon)
  TRY4(integrand_quad2)
  TRY4(pairprod)
  TRY4(integrand_powp)
  TRY4(reentry_heating)
  TRY4(cartpole)
  TRY4(integrand_usq)
  TRY4(freeflyingrobot)
  TRY4(integrand_sum4)
  TRY4(cannon)
  TRY4(cannon_energy)
  TRY4(integrand_lq)
  TRY4(integrand_lq_pi)
  TRY4(lq1)
  TRY4(delta3_1)
  TRY4(delta3_2)
  TRY4(delta3_3)
  TRY4(delta3_4)
  TRY4(norm3)
  TRY4(delta3_orbit)
  TRY4(shape_1_0_0)
  TRY4(shape_1_1_0)
  TRY4(shape_2_1_0)
  TRY4(shape_3_0_1)
  TRY4(shape_4_4_0)
  TRY4(shape_5_3_2)
  TRY4(shape_6_0_0)
  TRY4(shape_8_3_1)
  TRY4(shape_10_4_0)
  TRY4(shape_11_4_0)
  return -1;
}

extern "C" {

int oracle_get_ode(const char* name, int provider, oracle_ode* out) {
  TRY(brachistochrone, 3, 1, 0, nullptr)
  TRY(reentry, 5, 2, 0, nullptr)
  TRY(twobody_lt, 6, 3, 0, nullptr)
  TRY(betts_lowthrust, 7, 3, 1, nullptr)
  TRY(synthetic32, 32, 0, 0, g_synth32)
  TRY(vanderpol, 2, 1, 1, nullptr)
  TRY(switched, 2, 1, 0, nullptr)
  TRY(tabulated, 2, 1, 0, nullptr)
  TRY(coupled12, 12, 3, 2, nullptr)
  TRY(coupled16, 16, 3, 2, nullptr)
  TRY(driven14, 14, 3, 0, nullptr)
  TRY(driven20, 20, 3, 0, nullptr)
  TRY(pathcon, 2, 3, 0, nullptr)
  TRY(integrand_quad2, 1, 0, 0, nullptr)
  TRY(pairprod, 1, 2, 0, nullptr)
  TRY(integrand_powp, 1, 2, 0, nullptr)
  TRY(reentry_heating, 1, 1, 0, nullptr)
  TRY(cartpole, 4, 1, 0, nullptr)
  TRY(integrand_usq, 1, 0, 0, nullptr)
  TRY(freeflyingrobot, 6, 4, 0, nullptr)
  TRY(integrand_sum4, 1, 2, 0, nullptr)
  TRY(cannon, 4, 0, 1, nullptr)
  TRY(cannon_energy, 1, 0, 0, nullptr)
  TRY(integrand_lq, 1, 0, 0, nullptr)
  TRY(integrand_lq_pi, 1, 0, 0, nullptr)
  TRY(lq1, 1, 1, 0, nullptr)
  TRY(delta3_1, 7, 3, 0, nullptr)
  TRY(delta3_2, 7, 3, 0, nullptr)
  TRY(delta3_3, 7, 3, 0, nullptr)
  TRY(delta3_4, 7, 3, 0, nullptr)
  TRY(norm3, 1, 1, 0, nullptr)
  TRY(delta3_orbit, 5, 0, 0, nullptr)
  TRY(shape_1_0_0, 1, 0, 0, nullptr)
  TRY(shape_1_1_0, 1, 1, 0, nullptr)
  TRY(shape_2_1_0, 2, 1, 0, nullptr)
  TRY(shape_3_0_1, 3, 0, 1, nullptr)
  TRY(shape_4_4_0, 4, 4, 0, nullptr)
  TRY(shape_5_3_2, 5, 3, 2, nullptr)
  TRY(shape_6_0_0, 6, 0, 0, nullptr)
  TRY(shape_8_3_1, 8, 3, 1, nullptr)
  TRY(shape_10_4_0, 10, 4, 0, nullptr)
  TRY(shape_11_4_0, 11, 4, 0, nullptr)
  return -1;
}
}
