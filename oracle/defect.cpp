// ORACLE (test infrastructure, never shipped or imported by the product path).
//
// Restatement of the three per-segment defect methods, keeping the reference's data flow so that
// intermediate quantities (C_XS, C_DXS, I_XS, DI_DCS, C_AVS, HTpar ...) can be diffed:
//   LGL value                 /root/reference/src/OptimalControl/LGLDefects.h:57-122
//   LGL value + Jacobian      LGLDefects.h:123-286
//   LGL value+J+adjgrad+adjH  LGLDefects.h:289-551
//   Trapezoidal (three forms) /root/reference/src/OptimalControl/TrapezoidalDefects.h:146-184, 186-260, 263-435
//   BlockConstant relabelling /root/reference/src/OptimalControl/Blocked_ODE_Wrapper.h:7-27
// Strict IEEE double, no -ffast-math.  Outputs follow Eigen's column-major convention.
#include <cstring>
#include <vector>

#include "lgl_coeffs.h"
#include "oracle.h"
#include "batch4.h"

namespace {

struct Sizes {
  int n, m, p;   // defect-level X/U/P sizes (after BlockConstant relabelling)
  int q;         // XtU = n+1+m
  int N;         // ODE input rows = q+p
  int T;         // time index inside a node = n
  int CS, K, IR, OR;
};

bool make_sizes(const oracle_ode* ode, int mode, int blocked, Sizes& s) {
  if (mode == ORACLE_FUNCTION) {   // the function itself is the constraint: one application = one evaluation
    s.n = ode->xv, s.m = ode->uv, s.p = ode->pv;
    s.q = s.n + 1 + s.m, s.N = s.q + s.p, s.T = s.n, s.CS = 1, s.K = 0;
    s.IR = s.N, s.OR = s.n;
    return true;
  }
  if (mode < ORACLE_TRAPEZOIDAL || mode > ORACLE_LGL7) return false;
  s.n = ode->xv;
  if (blocked) {  // Blocked_ODE_Wrapper: UV:=0, PV:=UV+PV, XtUV:=XtV ; the ODE input layout is unchanged
    s.m = 0;
    s.p = ode->uv + ode->pv;
  } else {
    s.m = ode->uv;
    s.p = ode->pv;
  }
  s.q = s.n + 1 + s.m;
  s.N = s.q + s.p;
  s.T = s.n;
  s.CS = (mode == ORACLE_TRAPEZOIDAL) ? 2 : mode;
  s.K = s.CS - 1;
  s.IR = s.CS * s.q + s.p;   // TranscriptionSizing.h:7-14
  s.OR = s.K * s.n;
  return true;
}

// The LGL bodies below are templates over the arithmetic type R: double (the parity oracle) or four segments side by
// side in one AVX register (v4d; the reference's SuperScalar batching, DenseFunctionBase.h:1318-1380 with
// DefaultSuperScalar = Array<double,4,1>, TypeDefs/EigenTypes.h:72-75).  With R = double the operations and their order
// are exactly those of the untemplated text this replaced.
template <class R> inline R bc(double c);
template <> inline double bc<double>(double c) { return c; }
template <> inline v4d bc<v4d>(double c) { return v4d{c, c, c, c}; }

struct OdeScalar {   // the registry's function pointers (odes.cpp)
  const oracle_ode* o;
  void f(const double* y, double* fx) const { o->f(y, fx, o->ctx); }
  void fj(const double* y, double* fx, double* J) const { o->fj(y, fx, J, o->ctx); }
  void fjgh(const double* y, const double* lam, double* fx, double* J, double* g, double* H) const {
    o->fjgh(y, lam, fx, J, g, H, o->ctx);
  }
};
struct OdeV4 {       // generated analytic derivatives compiled for four segments at once (gen/odes_gen4.c)
  const oracle_ode4* o;
  void f(const v4d* y, v4d* fx) const { o->f(y, fx); }
  void fjgh(const v4d* y, const v4d* lam, v4d* fx, v4d* J, v4d* g, v4d* H) const { o->fjgh(y, lam, fx, J, g, H); }
};

// column-major element accessors
#define DI(a, c) di[(a) + (size_t)(c) * S.N]
#define JX(r, c) jx[(r) + (size_t)(c) * S.OR]
#define HX(r, c) hx[(r) + (size_t)(c) * S.IR]
// ODE derivative storage: J row-major (n x N), H full (N x N)
#define OJ(J, k, a) (J)[(size_t)(k) * S.N + (a)]
#define OH(H, a, b) (H)[(size_t)(a) * S.N + (b)]

template <class R>
void load_cardinals(const Sizes& S, const R* x, std::vector<R>& C_XS) {
  C_XS.assign((size_t)S.CS * S.N, bc<R>(0.0));
  for (int i = 0; i < S.CS; i++) {
    R* c = &C_XS[(size_t)i * S.N];
    for (int k = 0; k < S.q; k++) c[k] = x[i * S.q + k];
    for (int k = 0; k < S.p; k++) c[S.q + k] = x[S.CS * S.q + k];
  }
}

// ------------------------------------------------------------------------------------------- LGL
void lgl_compute(const oracle_ode* ode, const Sizes& S, const double* x, double* fx) {
  const lgl_tables* L = lgl_get(S.CS);
  std::vector<double> C_XS, C_DXS((size_t)S.CS * S.n, 0.0), I_XS((size_t)S.K * S.N, 0.0), I_DXS(S.n, 0.0);
  load_cardinals(S, x, C_XS);
  for (int i = 0; i < S.CS; i++) ode->f(&C_XS[(size_t)i * S.N], &C_DXS[(size_t)i * S.n], ode->ctx);  // :65-75
  const double h = C_XS[(size_t)(S.CS - 1) * S.N + S.T] - C_XS[S.T];                                   // :77
  for (int i = 0; i < S.K; i++) {
    double* ix = &I_XS[(size_t)i * S.N];
    ix[S.T] = C_XS[S.T] + h * L->s[i];                                                                 // :80
    for (int k = 0; k < S.p; k++) ix[S.q + k] = x[S.CS * S.q + k];
    for (int j = 0; j < S.CS; j++) {                                                                   // :87-100
      const double* cx = &C_XS[(size_t)j * S.N];
      const double* cd = &C_DXS[(size_t)j * S.n];
      for (int k = 0; k < S.n; k++) ix[k] += (L->A[i][j] * cx[k] + (L->B[i][j] * h) * cd[k]);
      for (int k = 0; k < S.m; k++) ix[S.n + 1 + k] += L->U[i][j] * cx[S.n + 1 + k];
      for (int k = 0; k < S.n; k++) fx[i * S.n + k] += (L->C[i][j] * cx[k] + (L->D[i][j] * h) * cd[k]);
    }
    ode->f(ix, I_DXS.data(), ode->ctx);                                                                // :101
    for (int k = 0; k < S.n; k++) fx[i * S.n + k] += (h * L->E[i]) * I_DXS[k];                         // :102-103
  }
}

// Builds DI_DCS = d(x^_i, tau_i, u^_i, P)/dz for interior i (LGLDefects.h:162-216 / :415-458)
template <class R>
void build_DI(const Sizes& S, const lgl_tables* L, int i, R h, const std::vector<R>& C_DXS,
              const std::vector<R>& C_JDXS, std::vector<R>& di) {
  std::fill(di.begin(), di.end(), bc<R>(0.0));
  const int tf = S.q * (S.CS - 1) + S.T;
  DI(S.T, S.T) = bc<R>(1.0 - L->s[i]);
  DI(S.T, tf) = bc<R>(L->s[i]);
  for (int k = 0; k < S.p; k++) DI(S.q + k, S.CS * S.q + k) = bc<R>(1.0);
  for (int j = 0; j < S.CS; j++) {
    const R* Jj = &C_JDXS[(size_t)j * S.n * S.N];
    const R* fj = &C_DXS[(size_t)j * S.n];
    for (int k = 0; k < S.n; k++) DI(k, j * S.q + k) = bc<R>(L->A[i][j]);  // diagonal .setConstant (assignment)
    const R bh = L->B[i][j] * h;
    for (int k = 0; k < S.n; k++)
      for (int c = 0; c < S.q; c++) DI(k, j * S.q + c) += bh * OJ(Jj, k, c);
    for (int k = 0; k < S.n; k++)
      for (int c = 0; c < S.p; c++) DI(k, S.CS * S.q + c) += bh * OJ(Jj, k, S.q + c);
    for (int k = 0; k < S.n; k++) DI(k, S.T) -= L->B[i][j] * fj[k];
    for (int k = 0; k < S.n; k++) DI(k, tf) += L->B[i][j] * fj[k];
    for (int k = 0; k < S.m; k++) DI(S.n + 1 + k, j * S.q + S.n + 1 + k) = bc<R>(L->U[i][j]);
  }
}

// Cardinal contributions to fx rows / jx rows of interior i  (LGLDefects.h:218-246 / :460-488)
template <class R>
void cardinal_fx_jx(const Sizes& S, const lgl_tables* L, int i, R h, const std::vector<R>& C_XS,
                    const std::vector<R>& C_DXS, const std::vector<R>& C_JDXS, R* fx, R* jx) {
  const int tf = S.q * (S.CS - 1) + S.T;
  for (int j = 0; j < S.CS; j++) {
    const R* cx = &C_XS[(size_t)j * S.N];
    const R* cd = &C_DXS[(size_t)j * S.n];
    const R* Jj = &C_JDXS[(size_t)j * S.n * S.N];
    for (int k = 0; k < S.n; k++) fx[i * S.n + k] += (L->C[i][j] * cx[k] + (L->D[i][j] * h) * cd[k]);
    for (int k = 0; k < S.n; k++) JX(i * S.n + k, j * S.q + k) = bc<R>(L->C[i][j]);
    const R dh = L->D[i][j] * h;
    for (int k = 0; k < S.n; k++)
      for (int c = 0; c < S.q; c++) JX(i * S.n + k, j * S.q + c) += dh * OJ(Jj, k, c);
    for (int k = 0; k < S.n; k++)
      for (int c = 0; c < S.p; c++) JX(i * S.n + k, S.CS * S.q + c) += dh * OJ(Jj, k, S.q + c);
    for (int k = 0; k < S.n; k++) JX(i * S.n + k, S.T) -= L->D[i][j] * cd[k];
    for (int k = 0; k < S.n; k++) JX(i * S.n + k, tf) += L->D[i][j] * cd[k];
  }
}

// Interior contribution to fx / jx rows of interior i  (LGLDefects.h:251-260 / :491-500)
template <class R>
void interior_fx_jx(const Sizes& S, const lgl_tables* L, int i, R h, const R* I_DXS, const R* I_JDXS,
                    const std::vector<R>& di, R* fx, R* jx) {
  const int tf = S.q * (S.CS - 1) + S.T;
  const R he = h * L->E[i];
  for (int k = 0; k < S.n; k++) fx[i * S.n + k] += he * I_DXS[k];
  for (int k = 0; k < S.n; k++)
    for (int c = 0; c < S.IR; c++) {
      R acc = bc<R>(0.0);
      for (int a = 0; a < S.N; a++) acc += (he * OJ(I_JDXS, k, a)) * DI(a, c);
      JX(i * S.n + k, c) += acc;
    }
  for (int k = 0; k < S.n; k++) JX(i * S.n + k, S.T) -= L->E[i] * I_DXS[k];
  for (int k = 0; k < S.n; k++) JX(i * S.n + k, tf) += L->E[i] * I_DXS[k];
}

void lgl_jacobian(const oracle_ode* ode, const Sizes& S, const double* x, double* fx, double* jx) {
  const lgl_tables* L = lgl_get(S.CS);
  std::vector<double> C_XS, C_DXS((size_t)S.CS * S.n, 0.0), C_JDXS((size_t)S.CS * S.n * S.N, 0.0);
  std::vector<double> I_XS((size_t)S.K * S.N, 0.0), I_DXS(S.n, 0.0), I_JDXS((size_t)S.n * S.N, 0.0);
  std::vector<double> di((size_t)S.N * S.IR, 0.0);
  load_cardinals(S, x, C_XS);
  for (int i = 0; i < S.CS; i++)
    ode->fj(&C_XS[(size_t)i * S.N], &C_DXS[(size_t)i * S.n], &C_JDXS[(size_t)i * S.n * S.N], ode->ctx);  // :139-149
  const double h = C_XS[(size_t)(S.CS - 1) * S.N + S.T] - C_XS[S.T];
  for (int i = 0; i < S.K; i++) {
    double* ix = &I_XS[(size_t)i * S.N];
    ix[S.T] = C_XS[S.T] + h * L->s[i];
    for (int k = 0; k < S.p; k++) ix[S.q + k] = x[S.CS * S.q + k];
    for (int j = 0; j < S.CS; j++) {
      const double* cx = &C_XS[(size_t)j * S.N];
      const double* cd = &C_DXS[(size_t)j * S.n];
      for (int k = 0; k < S.n; k++) ix[k] += (L->A[i][j] * cx[k] + (L->B[i][j] * h) * cd[k]);
      for (int k = 0; k < S.m; k++) ix[S.n + 1 + k] += L->U[i][j] * cx[S.n + 1 + k];
    }
    build_DI(S, L, i, h, C_DXS, C_JDXS, di);
    cardinal_fx_jx(S, L, i, h, C_XS, C_DXS, C_JDXS, fx, jx);
    ode->fj(ix, I_DXS.data(), I_JDXS.data(), ode->ctx);                                                  // :249
    interior_fx_jx(S, L, i, h, I_DXS.data(), I_JDXS.data(), di, fx, jx);
  }
}

template <class R, class ODE>
void lgl_all_t(const ODE& ode, const Sizes& S, const R* x, const R* lam, R* fx, R* jx, R* agx, R* hx) {
  const lgl_tables* L = lgl_get(S.CS);
  const int tf = S.q * (S.CS - 1) + S.T;
  std::vector<R> C_XS, C_DXS((size_t)S.CS * S.n, bc<R>(0.0)), C_JDXS((size_t)S.CS * S.n * S.N, bc<R>(0.0));
  std::vector<R> C_AGXS(S.N, bc<R>(0.0)), C_AVS((size_t)S.CS * S.n, bc<R>(0.0)), C_HDXS((size_t)S.N * S.N, bc<R>(0.0));
  std::vector<R> I_XS((size_t)S.K * S.N, bc<R>(0.0)), I_DXS((size_t)S.K * S.n, bc<R>(0.0));
  std::vector<R> I_JDXS((size_t)S.K * S.n * S.N, bc<R>(0.0)), I_AGXS((size_t)S.K * S.N, bc<R>(0.0));
  std::vector<R> I_AVS((size_t)S.K * S.n, bc<R>(0.0)), I_HDXS((size_t)S.K * S.N * S.N, bc<R>(0.0));
  std::vector<R> di((size_t)S.N * S.IR, bc<R>(0.0)), HTpar(S.IR, bc<R>(0.0)), tmp((size_t)S.N * S.IR, bc<R>(0.0));

  load_cardinals(S, x, C_XS);
  // (1) cardinal values only (:325-337).  The reference evaluates the cardinal ODE value a second time in
  // step (3); ODE nodes assign their outputs, so one evaluation is equivalent.
  for (int i = 0; i < S.CS; i++) ode.f(&C_XS[(size_t)i * S.N], &C_DXS[(size_t)i * S.n]);
  const R h = C_XS[(size_t)(S.CS - 1) * S.N + S.T] - C_XS[S.T];                                   // :339

  // (2) interiors: interpolate, full ODE evaluation with lambda_i, accumulate cardinal adjoint weights (:341-375)
  for (int i = 0; i < S.K; i++) {
    R* ix = &I_XS[(size_t)i * S.N];
    ix[S.T] = C_XS[S.T] + h * L->s[i];
    for (int k = 0; k < S.p; k++) ix[S.q + k] = x[S.CS * S.q + k];
    R* iav = &I_AVS[(size_t)i * S.n];
    for (int k = 0; k < S.n; k++) iav[k] = lam[i * S.n + k];
    for (int j = 0; j < S.CS; j++) {
      const R* cx = &C_XS[(size_t)j * S.N];
      const R* cd = &C_DXS[(size_t)j * S.n];
      for (int k = 0; k < S.n; k++) ix[k] += (L->A[i][j] * cx[k] + (L->B[i][j] * h) * cd[k]);
      for (int k = 0; k < S.m; k++) ix[S.n + 1 + k] += L->U[i][j] * cx[S.n + 1 + k];
    }
    ode.fjgh(ix, iav, &I_DXS[(size_t)i * S.n], &I_JDXS[(size_t)i * S.n * S.N], &I_AGXS[(size_t)i * S.N],
             &I_HDXS[(size_t)i * S.N * S.N]);
    for (int j = 0; j < S.CS; j++) {
      const double scale = L->E[i] * L->B[i][j];
      R* cav = &C_AVS[(size_t)j * S.n];
      for (int k = 0; k < S.n; k++) cav[k] += I_AGXS[(size_t)i * S.N + k] * (scale * h * h);
      for (int k = 0; k < S.n; k++) cav[k] += iav[k] * (L->D[i][j] * h);
    }
  }

  // (3) cardinals: Jacobian + Hessian weighted by the accumulated adjoint (:377-412)
  for (int j = 0; j < S.CS; j++) {
    std::fill(C_AGXS.begin(), C_AGXS.end(), bc<R>(0.0));
    std::fill(C_HDXS.begin(), C_HDXS.end(), bc<R>(0.0));
    ode.fjgh(&C_XS[(size_t)j * S.N], &C_AVS[(size_t)j * S.n], &C_DXS[(size_t)j * S.n],
             &C_JDXS[(size_t)j * S.n * S.N], C_AGXS.data(), C_HDXS.data());
    const int o = j * S.q, P0 = S.CS * S.q;
    for (int a = 0; a < S.q; a++)
      for (int b = 0; b < S.q; b++) HX(o + a, o + b) += OH(C_HDXS, a, b);
    for (int a = 0; a < S.q; a++)
      for (int b = 0; b < S.p; b++) HX(o + a, P0 + b) += OH(C_HDXS, a, S.q + b);
    for (int a = 0; a < S.p; a++)
      for (int b = 0; b < S.q; b++) HX(P0 + a, o + b) += OH(C_HDXS, S.q + a, b);
    for (int a = 0; a < S.p; a++)
      for (int b = 0; b < S.p; b++) HX(P0 + a, P0 + b) += OH(C_HDXS, S.q + a, S.q + b);
    for (int a = 0; a < S.q; a++) HTpar[o + a] += C_AGXS[a] * (1.0 / h);
    for (int a = 0; a < S.p; a++) HTpar[P0 + a] += C_AGXS[S.q + a] * (1.0 / h);
  }

  // (4) interiors: rebuild DI_DCS, Jacobian rows, Hessian congruence, time-partial vector (:414-506)
  for (int i = 0; i < S.K; i++) {
    build_DI(S, L, i, h, C_DXS, C_JDXS, di);
    cardinal_fx_jx(S, L, i, h, C_XS, C_DXS, C_JDXS, fx, jx);
    interior_fx_jx(S, L, i, h, &I_DXS[(size_t)i * S.n], &I_JDXS[(size_t)i * S.n * S.N], di, fx, jx);
    const R he = h * L->E[i];
    const R* Hi = &I_HDXS[(size_t)i * S.N * S.N];
    // tmp = (H_i * he) * DI   (N x IR)
    for (int a = 0; a < S.N; a++)
      for (int c = 0; c < S.IR; c++) {
        R acc = bc<R>(0.0);
        for (int b = 0; b < S.N; b++) acc += (OH(Hi, a, b) * he) * DI(b, c);
        tmp[a + (size_t)c * S.N] = acc;
      }
    for (int r = 0; r < S.IR; r++)
      for (int c = 0; c < S.IR; c++) {
        R acc = bc<R>(0.0);
        for (int a = 0; a < S.N; a++) acc += DI(a, r) * tmp[a + (size_t)c * S.N];
        HX(r, c) += acc;
      }
    for (int c = 0; c < S.IR; c++) {
      R acc = bc<R>(0.0);
      for (int a = 0; a < S.N; a++) acc += (I_AGXS[(size_t)i * S.N + a] * L->E[i]) * DI(a, c);
      HTpar[c] += acc;
    }
  }

  // (5) rank-2 time update and adjoint gradient (:508-512)
  for (int r = 0; r < S.IR; r++) HX(r, S.T) -= HTpar[r];
  for (int r = 0; r < S.IR; r++) HX(r, tf) += HTpar[r];
  for (int c = 0; c < S.IR; c++) HX(S.T, c) -= HTpar[c];
  for (int c = 0; c < S.IR; c++) HX(tf, c) += HTpar[c];
  for (int c = 0; c < S.IR; c++) {
    R acc = bc<R>(0.0);
    for (int r = 0; r < S.OR; r++) acc += lam[r] * JX(r, c);
    agx[c] = acc;
  }
}

void lgl_all(const oracle_ode* ode, const Sizes& S, const double* x, const double* lam, double* fx, double* jx,
             double* agx, double* hx) {
  lgl_all_t<double>(OdeScalar{ode}, S, x, lam, fx, jx, agx, hx);
}

// ----------------------------------------------------------------------------------- Trapezoidal
void trap_load(const Sizes& S, const double* x, std::vector<double>& X0, std::vector<double>& X1) {
  X0.assign(S.N, 0.0);
  X1.assign(S.N, 0.0);
  for (int k = 0; k < S.q; k++) {
    X0[k] = x[k];
    X1[k] = x[S.q + k];
  }
  for (int k = 0; k < S.p; k++) X0[S.q + k] = X1[S.q + k] = x[2 * S.q + k];
}

void trap_compute(const oracle_ode* ode, const Sizes& S, const double* x, double* fx) {
  std::vector<double> X0, X1, F0(S.n), F1(S.n);
  trap_load(S, x, X0, X1);
  const double h = X1[S.T] - X0[S.T];
  ode->f(X0.data(), F0.data(), ode->ctx);
  ode->f(X1.data(), F1.data(), ode->ctx);
  for (int k = 0; k < S.n; k++) fx[k] = (X1[k] - X0[k]) - (h / 2.0) * (F0[k] + F1[k]);   // :171-173
  for (int k = 0; k < S.n; k++) fx[k] *= -1.0;                                             // :183
}

void trap_jac_core(const Sizes& S, double h, const std::vector<double>& X0, const std::vector<double>& X1,
                   const std::vector<double>& F0, const std::vector<double>& F1, const std::vector<double>& J0,
                   const std::vector<double>& J1, double* fx, double* jx) {
  for (int k = 0; k < S.n; k++) fx[k] = (X1[k] - X0[k]) - (h / 2.0) * (F0[k] + F1[k]);
  for (int k = 0; k < S.n; k++) JX(k, k) = -1.0;                                           // :213-218
  for (int k = 0; k < S.n; k++) JX(k, S.q + k) = 1.0;
  for (int k = 0; k < S.n; k++) {                                                          // :220-222
    const double tds = -0.5 * (F0[k] + F1[k]);
    JX(k, S.T) -= tds;
    JX(k, S.q + S.T) += tds;
  }
  for (int k = 0; k < S.n; k++)
    for (int c = 0; c < S.q; c++) {
      JX(k, c) += (-h / 2.0) * OJ(J0, k, c);                                               // :224-231
      JX(k, S.q + c) += (-h / 2.0) * OJ(J1, k, c);
    }
  for (int k = 0; k < S.n; k++)
    for (int c = 0; c < S.p; c++) JX(k, 2 * S.q + c) = (-h / 2.0) * (OJ(J0, k, S.q + c) + OJ(J1, k, S.q + c));
}

void trap_jacobian(const oracle_ode* ode, const Sizes& S, const double* x, double* fx, double* jx) {
  std::vector<double> X0, X1, F0(S.n), F1(S.n), J0((size_t)S.n * S.N), J1((size_t)S.n * S.N);
  trap_load(S, x, X0, X1);
  const double h = X1[S.T] - X0[S.T];
  ode->fj(X0.data(), F0.data(), J0.data(), ode->ctx);
  ode->fj(X1.data(), F1.data(), J1.data(), ode->ctx);
  trap_jac_core(S, h, X0, X1, F0, F1, J0, J1, fx, jx);
  for (int k = 0; k < S.n; k++) fx[k] *= -1.0;                                             // :258-259
  for (size_t k = 0; k < (size_t)S.OR * S.IR; k++) jx[k] *= -1.0;
}

void trap_all(const oracle_ode* ode, const Sizes& S, const double* x, const double* lam, double* fx, double* jx,
              double* agx, double* hx) {
  std::vector<double> X0, X1, F0(S.n), F1(S.n), J0((size_t)S.n * S.N), J1((size_t)S.n * S.N);
  std::vector<double> G0(S.N), G1(S.N), H0((size_t)S.N * S.N), H1((size_t)S.N * S.N), HTpar(S.IR, 0.0);
  trap_load(S, x, X0, X1);
  const double h = X1[S.T] - X0[S.T];
  ode->fjgh(X0.data(), lam, F0.data(), J0.data(), G0.data(), H0.data(), ode->ctx);        // :357-358
  ode->fjgh(X1.data(), lam, F1.data(), J1.data(), G1.data(), H1.data(), ode->ctx);
  trap_jac_core(S, h, X0, X1, F0, F1, J0, J1, fx, jx);
  for (int c = 0; c < S.IR; c++) {                                                         // :391
    double acc = 0.0;
    for (int r = 0; r < S.OR; r++) acc += lam[r] * JX(r, c);
    agx[c] = acc;
  }
  const double mh2 = -h / 2.0;
  const int P0 = 2 * S.q;
  for (int a = 0; a < S.q; a++)
    for (int b = 0; b < S.q; b++) {
      HX(a, b) = mh2 * OH(H0, a, b);                                                       // :395-401
      HX(S.q + a, S.q + b) = mh2 * OH(H1, a, b);
    }
  for (int a = 0; a < S.p; a++)
    for (int b = 0; b < S.p; b++) HX(P0 + a, P0 + b) = mh2 * (OH(H0, S.q + a, S.q + b) + OH(H1, S.q + a, S.q + b));
  const std::vector<double>* Hs[2] = {&H0, &H1};
  const std::vector<double>* Gs[2] = {&G0, &G1};
  for (int j = 0; j < 2; j++) {                                                            // :408-449
    const std::vector<double>& Hj = *Hs[j];
    const std::vector<double>& Gj = *Gs[j];
    for (int a = 0; a < S.q; a++)
      for (int b = 0; b < S.p; b++) {
        HX(j * S.q + a, P0 + b) += mh2 * OH(Hj, a, S.q + b);
        HX(P0 + b, j * S.q + a) += mh2 * OH(Hj, S.q + b, a);
      }
    for (int a = 0; a < S.q; a++) HTpar[j * S.q + a] += -Gj[a] * 0.5;
    for (int a = 0; a < S.p; a++) HTpar[P0 + a] += -Gj[S.q + a] * 0.5;
  }
  const int tf = S.q + S.T;
  for (int r = 0; r < S.IR; r++) HX(r, S.T) -= HTpar[r];                                   // :451-454
  for (int r = 0; r < S.IR; r++) HX(r, tf) += HTpar[r];
  for (int c = 0; c < S.IR; c++) HX(S.T, c) -= HTpar[c];
  for (int c = 0; c < S.IR; c++) HX(tf, c) += HTpar[c];
  for (int k = 0; k < S.OR; k++) fx[k] *= -1.0;                                            // :431-434
  for (size_t k = 0; k < (size_t)S.OR * S.IR; k++) jx[k] *= -1.0;
  for (int k = 0; k < S.IR; k++) agx[k] *= -1.0;
  for (size_t k = 0; k < (size_t)S.IR * S.IR; k++) hx[k] *= -1.0;
}

}  // namespace

// Four segments at once (nlp.cpp, bench.py's cpu_baseline leg): x, lam, and every output are arrays of v4d, lane = segment.
int oracle_defect_all_v4(const oracle_ode* ode, const oracle_ode4* ode4, int mode, int blocked, const v4d* x, const v4d* lam,
                         v4d* fx, v4d* jx, v4d* agx, v4d* hx) {
  Sizes S;
  if (!make_sizes(ode, mode, blocked, S) || mode < ORACLE_LGL3 || !ode4 || !ode4->fjgh) return -1;
  std::fill(fx, fx + S.OR, bc<v4d>(0.0));
  std::fill(jx, jx + (size_t)S.OR * S.IR, bc<v4d>(0.0));
  std::fill(agx, agx + S.IR, bc<v4d>(0.0));
  std::fill(hx, hx + (size_t)S.IR * S.IR, bc<v4d>(0.0));
  lgl_all_t<v4d>(OdeV4{ode4}, S, x, lam, fx, jx, agx, hx);
  return 0;
}

extern "C" {

int oracle_defect_sizes(int mode, int xv, int uv, int pv, int blocked, int* irows, int* orows) {
  oracle_ode o;
  std::memset(&o, 0, sizeof o);
  o.xv = xv, o.uv = uv, o.pv = pv;
  Sizes S;
  if (!make_sizes(&o, mode, blocked, S)) return -1;
  *irows = S.IR;
  *orows = S.OR;
  return 0;
}

// All three entry points zero their outputs first: the reference's defect bodies accumulate with +=
// into buffers the caller has cleared (DenseFunctionBase.h:1304-1307).
int oracle_defect_compute(const oracle_ode* ode, int mode, int blocked, const double* x, double* fx) {
  Sizes S;
  if (!make_sizes(ode, mode, blocked, S)) return -1;
  std::fill(fx, fx + S.OR, 0.0);
  if (mode == ORACLE_FUNCTION) { ode->f(x, fx, ode->ctx); return 0; }
  if (mode == ORACLE_TRAPEZOIDAL) trap_compute(ode, S, x, fx);
  else lgl_compute(ode, S, x, fx);
  return 0;
}

int oracle_defect_jacobian(const oracle_ode* ode, int mode, int blocked, const double* x, double* fx, double* jx) {
  Sizes S;
  if (!make_sizes(ode, mode, blocked, S)) return -1;
  std::fill(fx, fx + S.OR, 0.0);
  std::fill(jx, jx + (size_t)S.OR * S.IR, 0.0);
  if (mode == ORACLE_FUNCTION) {   // row-major (OR x IR) from the provider -> column-major
    std::vector<double> J((size_t)S.OR * S.IR);
    ode->fj(x, fx, J.data(), ode->ctx);
    for (int r = 0; r < S.OR; r++)
      for (int c = 0; c < S.IR; c++) JX(r, c) = J[(size_t)r * S.IR + c];
    return 0;
  }
  if (mode == ORACLE_TRAPEZOIDAL) trap_jacobian(ode, S, x, fx, jx);
  else lgl_jacobian(ode, S, x, fx, jx);
  return 0;
}

int oracle_defect_all(const oracle_ode* ode, int mode, int blocked, const double* x, const double* lam, double* fx,
                      double* jx, double* agx, double* hx) {
  Sizes S;
  if (!make_sizes(ode, mode, blocked, S)) return -1;
  std::fill(fx, fx + S.OR, 0.0);
  std::fill(jx, jx + (size_t)S.OR * S.IR, 0.0);
  std::fill(agx, agx + S.IR, 0.0);
  std::fill(hx, hx + (size_t)S.IR * S.IR, 0.0);
  if (mode == ORACLE_FUNCTION) {   // value, J, adjoint gradient J^T lam, adjoint Hessian sum_k lam_k grad^2 f_k
    std::vector<double> J((size_t)S.OR * S.IR);
    ode->fjgh(x, lam, fx, J.data(), agx, hx, ode->ctx);   // H is symmetric: row- and column-major coincide
    for (int r = 0; r < S.OR; r++)
      for (int c = 0; c < S.IR; c++) JX(r, c) = J[(size_t)r * S.IR + c];
    return 0;
  }
  if (mode == ORACLE_TRAPEZOIDAL) trap_all(ode, S, x, lam, fx, jx, agx, hx);
  else lgl_all(ode, S, x, lam, fx, jx, agx, hx);
  return 0;
}

const double* oracle_lgl_table(int cs, const char* which) {
  const lgl_tables* L = lgl_get(cs);
  if (!L || !which) return nullptr;
  if (!std::strcmp(which, "tc")) return L->tc;
  if (!std::strcmp(which, "s")) return L->s;
  if (!std::strcmp(which, "A")) return &L->A[0][0];
  if (!std::strcmp(which, "B")) return &L->B[0][0];
  if (!std::strcmp(which, "U")) return &L->U[0][0];
  if (!std::strcmp(which, "C")) return &L->C[0][0];
  if (!std::strcmp(which, "D")) return &L->D[0][0];
  if (!std::strcmp(which, "E")) return L->E;
  return nullptr;
}
}
