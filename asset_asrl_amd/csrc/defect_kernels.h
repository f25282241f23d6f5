// Batched LGL collocation-defect kernels for gfx950 (CDNA4).  HIP only -- no CUDA path.
//
// What is computed, per mesh segment (SURVEY.md section 8 "Mathematical statement"):
//   value      d_i = sum_j (C_ij x_j + h D_ij f_j) + h E_i f(x^_i, tau_i, u^_i, P)          i = 0..K-1
//   Jacobian   J   = d d / d z                                                (OR x IR, dense)
//   adjoint    g   = J^T lam
//   Hessian    H   = sum_k lam_k grad^2 d_k                                   (IR x IR, lower triangle)
// i.e. the three methods of the reference's LGLDefects (/root/reference/src/OptimalControl/
// LGLDefects.h:57-122, 123-286, 289-551) evaluated for every segment of a phase and written as the
// per-application blocks the solver interface scatters into the KKT matrix
// (/root/reference/src/VectorFunctions/DenseFunctionBase.h:1097-1129, 1276-1391) -- or, in the ASM instantiations,
// added straight into the solver's value array (DenseFunctionBase.h:1413-1523).
//
// One evaluation is one launch of lgl_defect_kernel -- both stages in one kernel (STAGE 3 / 4, meshes whose shares fit one
// group per workgroup: up to 14 336 Reentry-LGL7 segments on 256 CUs) -- or two (STAGE 1, then STAGE 2): one 64-lane
// wavefront per workgroup (two in STAGE 4), persistent workgroups over contiguous shares of the segments; the stages
// hand over through a per-segment workspace slot (Dims: w_*) that the fused forms read back from L2 / the Infinity Cache:
//   STAGE 1, ODE stage -- lane <-> evaluation point, up to G segments of the workgroup at a time:
//     P0  gather z = X[Vindex], lam = L[Cindex] into the slots
//     P1  (segment, cardinal node):   f_j, and every transcendental sub-expression of it        (f_save)
//     P2  (segment, interior point):  x^,tau,u^ ; f^, J^, g^ = J^^T lam_i, H^ = lam_i^T d2f     (fjgh)
//     P3  (segment, cardinal node):   w_j ; J_j, g_j = J_j^T w_j, H_j = w_j^T d2f               (fjgh_load)
//     J / H non-zeros go through LDS staging rows so that the workspace is written with coalesced stores.
//   STAGE 2, dense stage -- all 64 lanes on ONE segment at a time.  With DI_i = d(x^_i,tau_i,u^_i,P)/dz (N x IR):
//         M_i^T   = DI_i^T [hE_i H^_i | E_i g^_i]        16x16x4 f64 MFMA, A = DI_i^T tiles from LDS
//         H      += DI_i^T M_i   (lower-triangle tiles)  same A fragments, B = M_i from LDS
//         J^T    += DI_i^T (hE_i J^_i)^T                  same A fragments (or spare columns of the M product)
//     The sparse remainder (cardinal diagonal blocks, cardinal part of J, the two time rows / columns) enters as
//     accumulator initial values and one rank-2 product; accumulators are stored straight to the KKT block.
// The ODE is an inlined generated functor (asset_asrl_amd/vf/codegen.py).  DESIGN.md section 4 has the measurements.
#pragma once
#include "defect_dims.h"

namespace asset_hip {

// ---------------------------------------------------------------------------------------------- ODE phases
// Kept out of line: each is a long straight-line generated body, and separating their register allocation from
// the dense phase keeps the latter's accumulators and fragments in registers.
// Where a phase of the ODE stage reads what an earlier phase left: the segment's workspace slot, or (MIR) its short
// LDS mirror (defect_dims.h: Dims::MIRROR).
template <class D, bool MIR>
struct PhaseIn {
  using P = std::conditional_t<MIR, const lds_double*, const double*>;
  P z, lam, Cf, Ig, SV;
  __device__ PhaseIn(const double* S, const lds_double* M)
      : z(pick(S + D::w_z, M + D::m_z)), lam(pick(S + D::w_lam, M + D::m_lam)), Cf(pick(S + D::w_Cf, M + D::m_Cf)),
        Ig(pick(S + D::w_Ig, M + D::m_Ig)), SV(pick(S + D::w_SV, M + D::m_SV)) {}
  __device__ static P pick(const double* s, const lds_double* m) {
    if constexpr (MIR) { (void)s; return m; } else { (void)m; return s; }
  }
};

template <class Ode, class D, int LEVEL, bool MIR>
__device__ __attribute__((noinline)) void interior_eval(double* S, lds_double* M, int i, const LglTab* tabp, lds_double* row) {
  constexpr int K = D::K, n = D::n, m = D::m, p = D::p, q = D::q, N = D::N, T = D::T, CS = D::CS;
  const LglTab& tab = *tabp;
  const PhaseIn<D, MIR> pin(S, M);
  const auto z = pin.z;
  const double h = z[D::TF] - z[T];
  double y[N];
  double li[n > 0 ? n : 1];
#pragma unroll
  for (int k = 0; k < n; k++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < CS; j++) acc += (tab.A[i][j] * z[j * q + k] + (tab.B[i][j] * h) * pin.Cf[j * n + k]);
    y[k] = acc;
  }
  y[T] = z[T] + h * tab.s[i];
#pragma unroll
  for (int k = 0; k < m; k++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < CS; j++) acc += tab.U[i][j] * z[j * q + n + 1 + k];
    y[n + 1 + k] = acc;
  }
#pragma unroll
  for (int k = 0; k < p; k++) y[q + k] = z[D::P0 + k];
#pragma unroll
  for (int k = 0; k < n; k++) li[k] = (LEVEL >= 1) ? pin.lam[i * n + k] : 0.0;
  RegIn<D> in{y, li};
  // g^_i feeds the adjoint weights of P3 (mirror copy); f^_i is only read by the dense stage
  OdeOutStaged<D, LEVEL == 1, false, false, MIR && LEVEL >= 2> out{S + D::w_If + i * n, S + D::w_Ig + i * N,
                                                                   stage_or<D>(row, S + D::w_IJ + i * D::NZJ),
                                                                   stage_or<D>(row + D::NZJ, S + D::w_IH + i * D::NZH)};
  if constexpr (MIR && LEVEL >= 2) out.g2_ = M + D::m_Ig + i * N;
  (void)K;
  if constexpr (LEVEL == 0) Ode::f(in, out);
  else if constexpr (LEVEL == 1) {
    out.lamv_ = li;
    Ode::fj(in, out);
#pragma unroll
    for (int b = 0; b < N; b++) S[D::w_Ig + i * N + b] = out.gacc_[b];   // adjoint gradient of the point, for D5
  } else Ode::fjgh(in, out);
}

template <class Ode, class D, bool MIR>
__device__ __attribute__((noinline)) void cardinal_eval2(double* S, lds_double* M, int j, const LglTab* tabp, lds_double* row) {
  constexpr int K = D::K, n = D::n, N = D::N, T = D::T;
  const LglTab& tab = *tabp;
  const PhaseIn<D, MIR> pin(S, M);
  const auto z = pin.z;
  const double h = z[D::TF] - z[T];
  double w[n > 0 ? n : 1];
#pragma unroll
  for (int k = 0; k < n; k++) {  // C_AVS[j]  (LGLDefects.h:369-374)
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < K; i++) {
      acc += pin.Ig[i * N + k] * ((tab.E[i] * tab.B[i][j]) * h * h);
      acc += pin.lam[i * n + k] * (tab.D[i][j] * h);
    }
    w[k] = acc;
  }
  CardIn<D, typename PhaseIn<D, MIR>::P> in{z, w, j, pin.SV + j * Ode::NSAVE};
  OdeOutStaged<D> out{S + D::w_Cf + j * n, S + D::w_Cg + j * N, stage_or<D>(row, S + D::w_CJ + j * D::NZJ),
                      stage_or<D>(row + D::NZJ, S + D::w_CH + j * D::NZH)};
  Ode::fjgh_load(in, out);   // the transcendental sub-expressions of f at this node were stored by P1
}

template <class D>
struct GatherIn {  // y = X[Vindex(node j, component i)]: the first ODE phase reads the solver vector directly
  const double* X;
  const int* vi;   // this segment's Vindex column
  int j;
  __device__ double y(int i) const { return X[i < D::q ? vi[j * D::q + i] : vi[D::P0 + (i - D::q)]]; }
  __device__ double lam(int) const { return 0.0; }
  __device__ double saved(int) const { return 0.0; }
};

template <class Ode, class D, int LEVEL, bool MIR, class In>
__device__ inline void cardinal_eval1_body(const In& in, double* S, lds_double* M, int j, lds_double* row) {
  OdeOutStaged<D, false, MIR, MIR> out{S + D::w_Cf + j * D::n, nullptr, stage_or<D>(row, S + D::w_CJ + j * D::NZJ), nullptr};
  if constexpr (MIR) {                                   // f_j: workspace (dense stage) and mirror (P2); saved values: mirror only
    out.f2_ = M + D::m_Cf + j * D::n;
    out.sv_ = M + D::m_SV + j * Ode::NSAVE;
  } else {
    out.sv_ = S + D::w_SV + j * Ode::NSAVE;
  }
  if constexpr (LEVEL == 1) Ode::fj(in, out);
  else if constexpr (LEVEL == 2) Ode::f_save(in, out);
  else Ode::f(in, out);
}

template <class Ode, class D, int LEVEL, bool MIR>
__device__ __attribute__((noinline)) void cardinal_eval1(double* S, lds_double* M, int j, lds_double* row, const double* X, const int* vi) {
  // (reads the solver vector itself rather than P0's copy in the mirror: its loads then overlap P0's -- measured 0.25 us)
  cardinal_eval1_body<Ode, D, LEVEL, MIR>(GatherIn<D>{X, vi, j}, S, M, j, row);
}

// Coalesced copy of `npt` LDS staging rows (NC doubles each) to the workspace: element idx = row*NC + k goes to
// dst_of(e0 + row, k).  Four elements per lane are in flight (all LDS reads of a batch are issued before its stores;
// one read-wait-store per trip exposes a full LDS latency each time) and (row, k) advance incrementally.
template <int NC, class F>
__device__ inline void copy_rows(const double* stage, int stg_ld, int npt, int e0, int lane, F&& dst_of) {
  constexpr int RSTEP = 64 / NC, KSTEP = 64 % NC;
  int row = lane / NC, k = lane - row * NC;
  const int total = npt * NC;
  for (int base = 0; base < total; base += 256) {
    double v[4];
    double* d[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      ok[u] = base + 64 * u + lane < total;
      v[u] = stage[ok[u] ? row * stg_ld + k : 0];
      d[u] = dst_of(e0 + row, k);
      row += RSTEP;
      k += KSTEP;
      if (k >= NC) { k -= NC; row++; }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (ok[u]) *d[u] = v[u];
  }
}

// ---------------------------------------------------------------------------------------------- pipelined ODE stage
// (PIPE shapes: derivative level 2, LDS mirror, staged rows, one pass per phase.)  gfx9 tracks loads and stores in one
// in-order vmcnt, a non-inlined device function drains it on entry AND before it returns, and a wait for any load that
// follows stores under lane conditions becomes vmcnt(0).  In the plain layout of the stage that serialised: the gather's
// stores (one full round trip each), every phase's own result stores (drained at its return) and -- the large one --
// the coalesced copy-out of a phase's J / H rows, which the next phase's entry waited for at the HBM write rate with
// every workgroup of the device doing the same at the same moment.  Here
//   * the ODE bodies write LDS only (staging row, mirror; f^ in the rows the interior phase leaves unused), except g_j;
//   * the copy-out of the interior rows is issued INSIDE the out-of-line function of the next phase, behind its entry
//     wait, so it drains under that phase's arithmetic (its exit wait then finds the stores done);
//   * what the dense stage needs of the mirror (z, lam, f_j, g^_i) is copied once, coalesced, at the end of the group.
template <class D, bool HASROW, bool FL, bool FG, bool GL, bool GG, bool SV>
struct OdeOutPipe {
  lds_double* row_;     // [J | H] staging row of this lane
  lds_double* fl_;      // LDS destination of f (FL)
  glb_double* fg_;      // global destination of f (FG)
  lds_double* gl_;      // LDS destination of g (GL)
  glb_double* gg_;      // global destination of g (GG)
  lds_double* sv_;      // saved transcendentals (SV)
  __device__ void f(int k, double v) {
    if constexpr (FL) fl_[k] = v;
    if constexpr (FG) fg_[k] = v;
  }
  __device__ void J(int k, int i, double v) {
    if constexpr (HASROW) { const int c = D::ode_t::JPOS[k * D::N + i]; if (c >= 0) row_[c] = v; }
  }
  __device__ void g(int i, double v) {
    if constexpr (GL) gl_[i] = v;
    if constexpr (GG) gg_[i] = v;
  }
  __device__ void H(int i, int j, double v) {
    if constexpr (HASROW) { const int c = D::ode_t::HPOS[i * (i + 1) / 2 + j]; if (c >= 0) row_[D::NZJ + c] = v; }
  }
  __device__ void save(int k, double v) { if constexpr (SV) sv_[k] = v; }
};

template <class D, int GP>
struct PipeDims {   // GP = segments per group: 64 / CS in the ODE-stage launch, Dims::GF in the fused kernel
  static constexpr int K = D::K, CS = D::CS, n = D::n;
  // f^_i of the interior phase: staged in the rows that phase leaves unused (GP*K of the GP*CS), else stored directly
  static constexpr bool IFROW = (GP * CS - GP * K) * D::STG_LD >= GP * K * n;
  static constexpr int if_off = GP * K * D::STG_LD;     // (doubles from the start of the staging rows)
};

// P1: f_j and its transcendental sub-expressions -> mirror.  No global store.
template <class Ode, class D>
__device__ __attribute__((noinline)) void pipe_cardinal_value(lds_double* M, int j, const double* X, const int* vi) {
  OdeOutPipe<D, false, true, false, false, false, true> out{nullptr, M + D::m_Cf + j * D::n, nullptr, nullptr, nullptr,
                                                           M + D::m_SV + j * Ode::NSAVE};
  GatherIn<D> in{X, vi, j};
  Ode::f_save(in, out);
}

// P2: interior point i of one segment: J^, H^ -> row; g^ -> mirror; f^ -> spare rows (or the slot).
template <class Ode, class D, bool IFROW>
__device__ __attribute__((noinline)) void pipe_interior(glb_double* S, lds_double* M, int i, const LglTab* tabp, lds_double* row,
                                                        lds_double* ifrow) {
  constexpr int n = D::n, m = D::m, p = D::p, q = D::q, N = D::N, T = D::T, CS = D::CS;
  const LglTab& tab = *tabp;
  const lds_double* z = M + D::m_z;
  const lds_double* Cf = M + D::m_Cf;
  const lds_double* lam = M + D::m_lam;
  const double h = z[D::TF] - z[T];
  double y[N];
  double li[n > 0 ? n : 1];
#pragma unroll
  for (int k = 0; k < n; k++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < CS; j++) acc += (tab.A[i][j] * z[j * q + k] + (tab.B[i][j] * h) * Cf[j * n + k]);
    y[k] = acc;
  }
  y[T] = z[T] + h * tab.s[i];
#pragma unroll
  for (int k = 0; k < m; k++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < CS; j++) acc += tab.U[i][j] * z[j * q + n + 1 + k];
    y[n + 1 + k] = acc;
  }
#pragma unroll
  for (int k = 0; k < p; k++) y[q + k] = z[D::P0 + k];
#pragma unroll
  for (int k = 0; k < n; k++) li[k] = lam[i * n + k];
  RegIn<D> in{y, li};
  OdeOutPipe<D, true, IFROW, !IFROW, true, false, false> out{
      row, ifrow, S + D::w_If + i * n, M + D::m_Ig + i * N, nullptr, nullptr};
  Ode::fjgh(in, out);
}

// Coalesced copy of `npt` staging rows ([J | H], NSTG doubles each) to the workspace sections (wJ, wH) of the points
// they belong to: point e = g * PER + k of segment g.  Four elements per lane in flight.
template <class D, int PER>
__device__ inline void pipe_copy_rows(const lds_double* stage, int npt, int lane, glb_double* Wg, int wJ, int wH,
                                      int row0 = 0) {   // rows [row0, row0 + npt) of the staging area (point = row)
  constexpr int NC = D::NSTG, RSTEP = 64 / NC, KSTEP = 64 % NC;
  int row = row0 + lane / NC, k = lane - (lane / NC) * NC;
  const int total = npt * NC;
  for (int base = 0; base < total; base += 256) {
    double v[4];
    glb_double* d[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      ok[u] = base + 64 * u + lane < total;
      v[u] = stage[ok[u] ? row * D::STG_LD + k : 0];
      const int g = row / PER, i = row - g * PER;
      d[u] = Wg + g * D::WSLOT + ((k < D::NZJ) ? (wJ + i * D::NZJ + k) : (wH + i * D::NZH + (k - D::NZJ)));
      row += RSTEP;
      k += KSTEP;
      if (k >= NC) { k -= NC; row++; }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (ok[u]) *d[u] = v[u];
  }
}

// P3 of a whole group: first the copy-out of the interior phase (rows + staged f^), then -- while those stores drain --
// the cardinal second derivatives of this lane's point into the same rows.  g_j goes straight to the slot.
template <class Ode, class D, int GP, bool COPY = true>
__device__ __attribute__((noinline)) void pipe_cardinal_second(glb_double* Wg, lds_double* mirror, lds_double* stage,
                                                               const LglTab* tabp, int gcount, int lane) {
  constexpr int K = D::K, n = D::n, N = D::N, T = D::T, CS = D::CS;
#if defined(ASSET_EXP_NOWS)
  if constexpr (false) {
#else
  if constexpr (!D::TRAP && COPY) {
#endif
    if constexpr (D::NSTG > 0) pipe_copy_rows<D, K>(stage, gcount * K, lane, Wg, D::w_IJ, D::w_IH);
    if constexpr (PipeDims<D, GP>::IFROW) {
      for (int e = lane; e < gcount * K * n; e += 64) {
        const int g = e / (K * n);
        Wg[g * D::WSLOT + D::w_If + (e - g * K * n)] = stage[PipeDims<D, GP>::if_off + e];
      }
    }
    wave_lds_sync();   // the rows are free once their reads have returned (the stores carry the data in registers)
  }
  if (lane < gcount * CS) {
    const int g = lane / CS, j = lane - g * CS;
    const LglTab& tab = *tabp;
    const lds_double* M = mirror + g * D::MSLOT;
    const lds_double* z = M + D::m_z;
    const double h = z[D::TF] - z[T];
    double w[n > 0 ? n : 1];
#pragma unroll
    for (int k = 0; k < n; k++) {  // C_AVS[j]  (LGLDefects.h:369-374)
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < K; i++) {
        acc += M[D::m_Ig + i * N + k] * ((tab.E[i] * tab.B[i][j]) * h * h);
        acc += M[D::m_lam + i * n + k] * (tab.D[i][j] * h);
      }
      w[k] = acc;
    }
    CardIn<D, const lds_double*> in{z, w, j, M + D::m_SV + j * Ode::NSAVE};
    OdeOutPipe<D, true, false, false, false, true, false> out{stage + lane * D::STG_LD, nullptr, nullptr, nullptr,
                                                              Wg + g * D::WSLOT + D::w_Cg + j * N, nullptr};
    Ode::fjgh_load(in, out);   // the transcendental sub-expressions of f at this node were stored by P1
  }
}

// Coalesced copy of NW consecutive doubles of every segment's mirror slot (from m0) to its workspace slot (from w0).
template <class D, int NW>
__device__ inline void pipe_copy_mirror(const lds_double* mirror, int gcount, int lane, glb_double* Wg, int m0, int w0,
                                        int g0 = 0) {   // segments [g0, g0 + gcount) of the group
  const int total = gcount * NW;
  for (int base = 0; base < total; base += 256) {
    double v[4];
    glb_double* d[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = base + 64 * u + lane;
      ok[u] = e < total;
      const int g = g0 + (ok[u] ? e / NW : 0), r = ok[u] ? e - (e / NW) * NW : 0;
      v[u] = mirror[g * D::MSLOT + m0 + r];
      d[u] = Wg + g * D::WSLOT + w0 + r;
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (ok[u]) *d[u] = v[u];
  }
}

// The ODE stage of one group of at most GP segments (lane <-> evaluation point, one pass per phase).
#if defined(ASSET_TIMING)
#define ASSET_PTS_PARAMS , long long* pts_, int* npts_
#define PTS() do { if (pts_ && *npts_ < 24) pts_[(*npts_)++] = clock64(); } while (0)
#else
#define ASSET_PTS_PARAMS
#define PTS() do {} while (0)
#endif
template <class Ode, class D, int GP, class Pub>
__device__ inline void pipe_ode_group(const EvalArgs& a, int lane, int seg0, int gcount, double* Wg, double* stage,
                                      lds_double* mirror, const LglTab* tabp, Pub&& publish_tables ASSET_PTS_PARAMS) {
  constexpr int CS = D::CS, K = D::K, n = D::n, N = D::N, IR = D::IR, OR = D::OR, STG_LD = D::STG_LD;
  static_assert(GP * CS <= 64, "one pass per phase");
  {   // P0: gather into the mirror only -- index loads, value loads, LDS writes; no store to wait behind
    constexpr int NZ = (GP * IR + 63) / 64, NL = (GP * OR + 63) / 64;
    const int* vseg = a.vindex + size_t(seg0) * IR;      // this group's Vindex / Cindex columns are contiguous
    const int* cseg = a.cindex + size_t(seg0) * OR;
    int vi[NZ], ci[NL];
#pragma unroll
    for (int t = 0; t < NZ; t++) vi[t] = (lane + 64 * t < gcount * IR) ? vseg[lane + 64 * t] : -1;
#pragma unroll
    for (int t = 0; t < NL; t++) ci[t] = (lane + 64 * t < gcount * OR) ? cseg[lane + 64 * t] : -1;
    double zv[NZ], lv[NL];
#pragma unroll
    for (int t = 0; t < NZ; t++) zv[t] = (vi[t] >= 0) ? a.X[vi[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NL; t++) lv[t] = (ci[t] >= 0) ? a.L[ci[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NZ; t++) {
      const int e = lane + 64 * t, g = e / IR, r = e - g * IR;
      if (e < gcount * IR) mirror[g * D::MSLOT + D::m_z + r] = zv[t];
    }
#pragma unroll
    for (int t = 0; t < NL; t++) {
      const int e = lane + 64 * t, g = e / OR, r = e - g * OR;
      if (e < gcount * OR) mirror[g * D::MSLOT + D::m_lam + r] = lv[t];
    }
  }
  PTS();                             // (timing builds) gather issued and written to the mirror
  if (lane < gcount * CS) {          // P1 (reads the solver vector itself: its loads overlap the gather's)
    const int g = lane / CS, j = lane - g * CS;
    pipe_cardinal_value<Ode, D>(mirror + g * D::MSLOT, j, a.X, a.vindex + size_t(seg0 + g) * IR);
  }
  wave_lds_sync();
  publish_tables();
  PTS();                             // P1
  if constexpr (!D::TRAP) {          // P2
    if (lane < gcount * K) {
      const int g = lane / K, i = lane - g * K;
      pipe_interior<Ode, D, PipeDims<D, GP>::IFROW>((glb_double*)Wg + g * D::WSLOT, mirror + g * D::MSLOT, i, tabp,
                                                    (lds_double*)(stage + lane * STG_LD),
                                                    (lds_double*)(stage + PipeDims<D, GP>::if_off + lane * n));
    }
    wave_lds_sync();
  }
  PTS();                             // P2
  // P3, preceded (inside the function) by the copy-out of P2
  pipe_cardinal_second<Ode, D, GP>((glb_double*)Wg, mirror, (lds_double*)stage, tabp, gcount, lane);
  wave_lds_sync();
  PTS();                             // copy-out of P2 + P3
#if !defined(ASSET_EXP_NOWS)
  pipe_copy_rows<D, CS>((const lds_double*)stage, D::NSTG > 0 ? gcount * CS : 0, lane, (glb_double*)Wg, D::w_CJ, D::w_CH);
#endif
#if defined(ASSET_EXP_NOWS)
  if constexpr (false)
#endif
  {   // what the dense stage reads of the mirror: [z | lam | f_j] (contiguous in both layouts) and g^_i
    constexpr int NA = IR + OR + CS * n, NB = K * N;
    static_assert(D::w_z == 0 && D::w_lam == IR && D::w_Cf == IR + OR && D::m_z == 0 && D::m_lam == IR && D::m_Cf == IR + OR,
                  "slot and mirror share the layout of their first three sections");
    pipe_copy_mirror<D, NA>(mirror, gcount, lane, (glb_double*)Wg, 0, 0);
    pipe_copy_mirror<D, NB>(mirror, gcount, lane, (glb_double*)Wg, D::m_Ig, D::w_Ig);
  }
  wave_lds_sync();   // (whatever follows rewrites the mirror and the rows)
  PTS();             // copy-out of P3 and of the mirror issued
}

// The ODE stage of one group for the Jacobian kinds (derivative level 1) inside the fused kernel: gather, cardinal f_j and
// J_j, interior f^_i, J^_i and g^_i = J^_i^T lam_i (accumulated while J^_i is emitted) -- the phases of the plain ODE-stage
// launch on the group's GP <= GF segments, rows and mirror in the LDS the dense phase uses afterwards.
template <class Ode, class D, int GP, class Pub>
__device__ inline void plain_ode_group_l1(const EvalArgs& a, int lane, int seg0, int gcount, double* Wg, double* stage,
                                          lds_double* mirror, const LglTab* tabp, Pub&& publish_tables) {
  constexpr int CS = D::CS, K = D::K, IR = D::IR, OR = D::OR, STG_LD = D::STG_LD;
  static_assert(GP * CS <= 64 && D::STAGED && D::MIRROR, "one pass per phase, staged rows, mirror");
  {   // P0: gather into the slots (the dense phase reads z, lam there) and the mirror (the interior phase does)
    constexpr int NZ = (GP * IR + 63) / 64, NL = (GP * OR + 63) / 64;
    const int* vseg = a.vindex + size_t(seg0) * IR;
    const int* cseg = a.cindex + size_t(seg0) * OR;
    int vi[NZ], ci[NL];
#pragma unroll
    for (int t = 0; t < NZ; t++) vi[t] = (lane + 64 * t < gcount * IR) ? vseg[lane + 64 * t] : -1;
#pragma unroll
    for (int t = 0; t < NL; t++) ci[t] = (a.L && lane + 64 * t < gcount * OR) ? cseg[lane + 64 * t] : -1;
    double zv[NZ], lv[NL];
#pragma unroll
    for (int t = 0; t < NZ; t++) zv[t] = (vi[t] >= 0) ? a.X[vi[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NL; t++) lv[t] = (ci[t] >= 0) ? a.L[ci[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NZ; t++) {
      const int e = lane + 64 * t, g = e / IR, r = e - g * IR;
      if (e < gcount * IR) {
        Wg[g * D::WSLOT + D::w_z + r] = zv[t];
        mirror[g * D::MSLOT + D::m_z + r] = zv[t];
      }
    }
#pragma unroll
    for (int t = 0; t < NL; t++) {
      const int e = lane + 64 * t, g = e / OR, r = e - g * OR;
      if (e < gcount * OR) {
        Wg[g * D::WSLOT + D::w_lam + r] = lv[t];
        mirror[g * D::MSLOT + D::m_lam + r] = lv[t];
      }
    }
  }
  // P1: cardinal f_j (slot + mirror) and J_j (row), then the rows to the slots
  if (lane < gcount * CS) {
    const int g = lane / CS, j = lane - g * CS;
    cardinal_eval1<Ode, D, 1, true>(Wg + g * D::WSLOT, mirror + g * D::MSLOT, j, (lds_double*)(stage + lane * STG_LD), a.X,
                                    a.vindex + size_t(seg0 + g) * IR);
  }
  wave_lds_sync();
  copy_rows<(D::NZJ > 0 ? D::NZJ : 1)>(stage, STG_LD, D::NZJ > 0 ? gcount * CS : 0, 0, lane, [&](int ee, int k) {
    const int g = ee / CS, j = ee - g * CS;
    return Wg + g * D::WSLOT + D::w_CJ + j * D::NZJ + k;
  });
  wave_lds_sync();
  publish_tables();
  // P2: interior f^_i, J^_i (row), g^_i (slot)
  if constexpr (!D::TRAP) {
    if (lane < gcount * K) {
      const int g = lane / K, i = lane - g * K;
      interior_eval<Ode, D, 1, true>(Wg + g * D::WSLOT, mirror + g * D::MSLOT, i, tabp, (lds_double*)(stage + lane * STG_LD));
    }
    wave_lds_sync();
    copy_rows<(D::NZJ > 0 ? D::NZJ : 1)>(stage, STG_LD, D::NZJ > 0 ? gcount * K : 0, 0, lane, [&](int ee, int k) {
      const int g = ee / K, i = ee - g * K;
      return Wg + g * D::WSLOT + D::w_IJ + i * D::NZJ + k;
    });
    wave_lds_sync();
  }
}

// The ODE stage of a TWO-WAVE workgroup (fused kernel, STAGE 4): the group is both waves' segments, wave 0 evaluates the
// cardinal phases (P1, P3), wave 1 the interior phase (P2) and the copy-out of its rows.  The generated bodies are bound
// by instruction issue, whatever the number of active lanes: one wave with 40 points costs the SIMD it runs on as much as
// one with 20, so letting each wave of a pair evaluate its own half (STAGE 3) issues every body twice per pair.  Here a
// body is issued once per pair, and while a wave waits at a barrier the other workgroup's wave on its SIMD has the issue
// slots.  Rows: region A = GP2*K rows (P2), region B = GP2*CS rows (P3) -- P3 runs while wave 1 still copies region A out.
template <class Ode, class D, int GP2, class Pub>
__device__ inline void pipe_ode_group2(const EvalArgs& a, int tid, int seg0, int gcount, double* Wg, double* stage,
                                       lds_double* mirror, const LglTab* tabp, Pub&& publish_tables) {
  constexpr int CS = D::CS, K = D::K, n = D::n, N = D::N, IR = D::IR, OR = D::OR, STG_LD = D::STG_LD;
  static_assert(GP2 * CS <= 64, "one pass per phase");
#define ASSET_ROLE_SHIFT 8
  // which wave of the pair takes the cardinal phases alternates from one workgroup of a CU to the next (workgroups are
  // dealt to the 8 XCDs x 32 CUs breadth first: workgroup b is the (b >> 8)-th of its CU), so that the cardinal waves of
  // the workgroups sharing a CU do not pile up on the same SIMDs
  const int lane = tid & 63, wave = (tid >> 6) ^ ((ASSET_ROLE_SHIFT >= 0) ? ((int(blockIdx.x) >> ASSET_ROLE_SHIFT) & 1) : 0);
  {   // P0: gather into the mirror, both waves
    constexpr int NZ = (GP2 * IR + 127) / 128, NL = (GP2 * OR + 127) / 128;
    const int* vseg = a.vindex + size_t(seg0) * IR;
    const int* cseg = a.cindex + size_t(seg0) * OR;
    int vi[NZ], ci[NL];
#pragma unroll
    for (int t = 0; t < NZ; t++) vi[t] = (tid + 128 * t < gcount * IR) ? vseg[tid + 128 * t] : -1;
#pragma unroll
    for (int t = 0; t < NL; t++) ci[t] = (tid + 128 * t < gcount * OR) ? cseg[tid + 128 * t] : -1;
    double zv[NZ], lv[NL];
#pragma unroll
    for (int t = 0; t < NZ; t++) zv[t] = (vi[t] >= 0) ? a.X[vi[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NL; t++) lv[t] = (ci[t] >= 0) ? a.L[ci[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NZ; t++) {
      const int e = tid + 128 * t, g = e / IR, r = e - g * IR;
      if (e < gcount * IR) mirror[g * D::MSLOT + D::m_z + r] = zv[t];
    }
#pragma unroll
    for (int t = 0; t < NL; t++) {
      const int e = tid + 128 * t, g = e / OR, r = e - g * OR;
      if (e < gcount * OR) mirror[g * D::MSLOT + D::m_lam + r] = lv[t];
    }
  }
  if (wave == 0 && lane < gcount * CS) {   // P1 (reads the solver vector itself: no barrier before it)
    const int g = lane / CS, j = lane - g * CS;
    pipe_cardinal_value<Ode, D>(mirror + g * D::MSLOT, j, a.X, a.vindex + size_t(seg0 + g) * IR);
  }
  publish_tables();                         // (each wave its own copy of the weight tables)
  __syncthreads();
  lds_double* const rowsA = (lds_double*)stage;
  lds_double* const rowsB = (lds_double*)stage + GP2 * K * STG_LD;
  if (wave == 1 && lane < gcount * K) {     // P2
    const int g = lane / K, i = lane - g * K;
    pipe_interior<Ode, D, false>((glb_double*)Wg + g * D::WSLOT, mirror + g * D::MSLOT, i, tabp, rowsA + lane * STG_LD, nullptr);
  }
  __syncthreads();
  if (wave == 1) {                          // copy-out of P2, while wave 0 evaluates P3 into its own rows
    if constexpr (D::NSTG > 0) pipe_copy_rows<D, K>(rowsA, gcount * K, lane, (glb_double*)Wg, D::w_IJ, D::w_IH);
  } else {
    pipe_cardinal_second<Ode, D, GP2, false>((glb_double*)Wg, mirror, rowsB, tabp, gcount, lane);
  }
  __syncthreads();
  {   // copy-out of P3 and of the mirror sections the dense phase reads: each wave takes half of the rows / segments
    const int half_r = (gcount * CS + 1) / 2, r0 = wave * half_r, nr = min(half_r, gcount * CS - r0);
    if constexpr (D::NSTG > 0)
      if (nr > 0) pipe_copy_rows<D, CS>(rowsB, nr, lane, (glb_double*)Wg, D::w_CJ, D::w_CH, r0);
    constexpr int NA = IR + OR + CS * n, NB = K * N;
    const int half_g = (gcount + 1) / 2, g0 = wave * half_g, ng = min(half_g, gcount - g0);
    if (ng > 0) {
      pipe_copy_mirror<D, NA>(mirror, ng, lane, (glb_double*)Wg, 0, 0, g0);
      pipe_copy_mirror<D, NB>(mirror, ng, lane, (glb_double*)Wg, D::m_Ig, D::w_Ig, g0);
    }
  }
  wave_loads_landed();                      // this wave's slot stores have completed ...
  __syncthreads();                          // ... and so have the other wave's: either may read any slot of the group now
}

// Per-lane constants of the dense stage: every index decode, table weight, fragment offset and store offset that
// depends only on the lane, never on the segment.  They are the same for every workgroup of every launch of a handle,
// and deriving them costs ~10 k cycles (dependent look-ups in the sparsity tables), a quarter of the time a workgroup
// spends on its five segments of the bench workload -- so lane_setup_kernel computes them once per handle and the
// dense stage loads its lane's record.
template <class Ode, class D, int LEVEL>
struct LaneConsts {
  static constexpr int K = D::K, n = D::n, p = D::p, q = D::q, N = D::N, CS = D::CS, KS = D::KS;
  static constexpr int IR = D::IR, OR = D::OR, IRP = D::IRP, P0 = D::P0;
  static constexpr int CW = D::CW, RPW = 64 / CW, ROWS = K * n;
  static constexpr bool BOTH = (RPW == 1);                 // one lane builds the DI and the DC entry of its column
  static constexpr int CMAIN = (P0 < CW) ? P0 : CW;        // block columns handled by the structured pass
  // Offsets are relative to the slot base S and always readable: "no entry" points at the zero cell, so the
  // per-segment loads need no branch (a conditional LDS load costs a full exposed latency each).
  static constexpr int ZERO = D::WSLOTD + D::s_Z0;
  // JFUSE (small ODEs: all defect rows fit one 16-lane tile next to the N columns of H^ and the g^ column): the B
  // operand of the M product of interior i carries (hE_i J^_i)^T in the lanes [i*n, i*n+n) -- the lanes that own the
  // defect rows (i, .) of the J tile -- the g^ column in lane 15 and the H^ columns in the remaining lanes, so the
  // interior part of J^T falls out of the same 2*TI MFMAs per interior and the separate J product disappears.
  static constexpr bool JFUSE = (LEVEL >= 2) && (D::TJ == 1) && (D::MT == 1) && (K * n <= 15) && (N <= 15 - n);

  double wa[K], wb[K], wa2[BOTH ? K : 1], wb2[BOTH ? K : 1];
  double tB[CS], tD[CS], tC[CS], tE;                       // weights of the row (i,r) this lane owns in the time-column pass
  int bo[D::MT][KS], bst[D::MT][KS];                       // B fragment of [hE H^ | E g^]: offset for i = 0 and stride in i
  int cho[LEVEL >= 2 ? D::NTH : 1][4];                     // cardinal Hessian entry feeding accumulator (tile, v)
  int chp[(LEVEL >= 2 && p > 0) ? D::NTH : 1][4];          // parameter-parameter entry (summed over the cardinal nodes) or -1
  int jo[D::TJ][K][KS];                                    // (hE J^)^T fragment
  int avb[KS], avs[KS];                                    // DI_i^T fragment: offset in the dense scratch for i = 0, stride in i
  int cjo[n];                                              // dfdy_j[r][cc] of this lane's block column (D1)
  int bo2[JFUSE ? K : 1][KS];                              // B operand offsets of the fused product
  int hst[D::NTH][4], jst[D::TI * D::TJ][4];               // KKT slot of accumulator entry (tile, v) or -1

  __device__ void compute(const LglTab& tab, int lane) {
    const int lr = lane & 15, lk = lane >> 4;
    const int d1c = lane & (CW - 1), d1h = lane / CW;
    const bool d1ok = (d1c < CMAIN) && (BOTH || d1h < 2);
    const int d1j = d1ok ? d1c / q : 0, d1cc = d1ok ? d1c - d1j * q : 0;
    tE = 0.0;
  #pragma unroll
      for (int i = 0; i < K; i++) {
        const bool dcrole = (!BOTH && d1h == 1);
        wa[i] = dcrole ? tab.C[i][d1j] : tab.A[i][d1j];
        wb[i] = dcrole ? tab.D[i][d1j] : tab.B[i][d1j];
        if constexpr (BOTH) { wa2[i] = tab.C[i][d1j]; wb2[i] = tab.D[i][d1j]; }
      }
      {
        const int e = (lane < ROWS) ? lane : 0, i = e / n;
  #pragma unroll
        for (int jj = 0; jj < CS; jj++) { tB[jj] = tab.B[i][jj]; tD[jj] = tab.D[i][jj]; tC[jj] = tab.C[i][jj]; }
        tE = tab.E[i];
      }
  #pragma unroll
      for (int mt = 0; mt < D::MT; mt++)
  #pragma unroll
        for (int kk = 0; kk < KS; kk++) {
          const int b = 4 * kk + lk, acol = 16 * mt + lr;
          int v = ZERO, st = 0;
          if (b < N) {
            if (acol < N) {
              const int hp = Ode::HPOS[(b >= acol) ? b * (b + 1) / 2 + acol : acol * (acol + 1) / 2 + b];
              if (hp >= 0) { v = D::w_IH + hp; st = D::NZH; }
            } else if (acol == N) { v = D::w_Ig + b; st = N; }   // the g^ column is scaled by E_i, the others by h E_i
          }
          bo[mt][kk] = v;
          bst[mt][kk] = st;
        }
      if constexpr (JFUSE) {
  #pragma unroll
        for (int i = 0; i < K; i++) {
          const int jk = lr - i * n;
          const bool isj = (jk >= 0 && jk < n);
          const int hr = (isj || lr == 15) ? -1 : ((lr < i * n) ? lr : lr - n);   // rank among the H^ lanes
          const bool ish = (hr >= 0 && hr < N);
  #pragma unroll
          for (int kk = 0; kk < KS; kk++) {
            const int b = 4 * kk + lk;
            int o = ZERO;
            if (b < N) {
              if (isj) { const int jp = Ode::JPOS[jk * N + b]; if (jp >= 0) o = D::w_IJ + i * D::NZJ + jp; }
              else if (ish) { const int hp = Ode::HPOS[(b >= hr) ? b * (b + 1) / 2 + hr : hr * (hr + 1) / 2 + b]; if (hp >= 0) o = D::w_IH + i * D::NZH + hp; }
              else if (lr == 15) o = D::w_Ig + i * N + b;
            }
            bo2[i][kk] = o;
          }
        }
      }
  #pragma unroll
      for (int r = 0; r < n; r++) {
        const int jp = Ode::JPOS[r * N + d1cc];
        cjo[r] = (d1ok && jp >= 0) ? D::w_CJ + d1j * D::NZJ + jp : ZERO;
      }
  #pragma unroll
      for (int kk = 0; kk < KS; kk++) {
        const int r = 4 * kk + lk;
        avb[kk] = ((r < n) ? D::s_DIx + r * IRP : D::s_DIc + (r - n) * IRP) + lr;
        avs[kk] = ((r < n) ? n : D::NCR) * IRP;
      }
  #pragma unroll
      for (int jt = 0; jt < D::TJ; jt++) {
        const int jr = 16 * jt + lr;
        const int ji = (jr < OR) ? jr / n : K, jk = (jr < OR) ? jr - ji * n : 0;
  #pragma unroll
        for (int i = 0; i < K; i++)
  #pragma unroll
          for (int kk = 0; kk < KS; kk++) {
            const int aa = 4 * kk + lk;
            const int jp = (ji == i && aa < N) ? Ode::JPOS[jk * N + aa] : -1;
            jo[jt][i][kk] = (jp >= 0) ? D::w_IJ + i * D::NZJ + jp : ZERO;
          }
      }
  #pragma unroll
      for (int ct = 0; ct < D::TI; ct++)
  #pragma unroll
        for (int v = 0; v < 4; v++) {
          const int c = 16 * ct + lk + 4 * v;
  #pragma unroll
          for (int jt = 0; jt < D::TJ; jt++) {
            const int jr = 16 * jt + lr;
            jst[ct * D::TJ + jt][v] = (c < IR && jr < OR) ? D::jcol(c) + jr : -1;      // (block layout: defect_dims.h, Dims::KL)
          }
  #pragma unroll
          for (int rt = ct; rt < D::TI; rt++) {
            const int r = 16 * rt + lr, tix = rt * (rt + 1) / 2 + ct;
            const bool ok = (c < IR && r < IR && r >= c);
            hst[tix][v] = ok ? D::hcol(c) + r : -1;
            if constexpr (LEVEL >= 2) {
              int ch = ZERO, cp = -1;
              if (ok) {
                if (c < P0) {
                  const int jn = c / q, cc = c - jn * q;
                  if (r < P0) {
                    if (r / q == jn) { const int rr = r - jn * q, hp = Ode::HPOS[rr * (rr + 1) / 2 + cc]; if (hp >= 0) ch = D::w_CH + jn * D::NZH + hp; }
                  } else {
                    const int rr = q + (r - P0), hp = Ode::HPOS[rr * (rr + 1) / 2 + cc];
                    if (hp >= 0) ch = D::w_CH + jn * D::NZH + hp;
                  }
                } else {
                  const int rr = q + (r - P0), c2 = q + (c - P0);
                  cp = Ode::HPOS[rr * (rr + 1) / 2 + c2];    // parameter-parameter: summed over the cardinal nodes
                }
              }
              cho[tix][v] = ch;
              if constexpr (p > 0) chp[tix][v] = cp;
            }
          }
        }
    }
};

// The 64 records are stored word-interleaved -- word k of lane l at [k*64 + l] -- so that a wave reads its records with
// coalesced loads (record-major storage costs 64 cache lines per load instruction).
template <class LC>
struct LaneRecord {
  static_assert(sizeof(LC) % 4 == 0, "record must be a whole number of words");
  static constexpr int NW = int(sizeof(LC) / 4);
  union { LC lc; unsigned int w[NW]; };
  __device__ LaneRecord() {}
};

template <class Ode, int SCH, bool BLOCKED, int LEVEL>
__global__ __launch_bounds__(64) void lane_setup_kernel(unsigned int* out) {
  if constexpr (!Dims<Ode, SCH, BLOCKED>::WIDE) {   // (wide shapes: wide_setup_kernel)
    using LC = LaneConsts<Ode, Dims<Ode, SCH, BLOCKED>, LEVEL>;
    LaneRecord<LC> r;
    for (int k = 0; k < LaneRecord<LC>::NW; k++) r.w[k] = 0u;
    r.lc.compute(d_lgl_tab[Dims<Ode, SCH, BLOCKED>::TAB], threadIdx.x);
    for (int k = 0; k < LaneRecord<LC>::NW; k++) out[k * 64 + threadIdx.x] = r.w[k];
  }
}

// ---------------------------------------------------------------------------------------------- kernel
// LEVEL 0: value only (constraints).  LEVEL 1: value + Jacobian (+ J^T lam).  LEVEL 2: + adjoint Hessian.
// STAGE 1: ODE phases only (P0-P3; results -> workspace slot of every segment).  STAGE 2: dense phase only (P4).
// They are separate launches because their resource shapes differ: the ODE bodies need ~250 VGPRs and wide LDS
// staging rows, the dense phase needs few registers and 23 KiB of LDS, so it runs at a higher occupancy.
// STAGE 3: both in one launch (FUSED shapes, derivative level 2, meshes of at most GF segments per workgroup): every
//          wave runs the ODE stage for its own few segments, waits for its own slot stores and goes on with the dense
//          phase -- one ramp-up (dispatch, first loads, cold instruction cache) per evaluation instead of two, no
//          device-wide drain of the slot stores between the stages, and the slots are read back from L2.
// STAGE 4: as STAGE 3 with TWO-wave workgroups (FUSED2 shapes): the ODE stage of both waves' segments is evaluated once
//          per pair (pipe_ode_group2), then each wave runs the dense phase of its own segments in its own half of the LDS.
template <class Ode, int SCH, bool BLOCKED, int G, int LEVEL, int STAGE, bool ASM>
__device__ __forceinline__ void lgl_defect_body(const EvalArgs& a) {
  using D = Dims<Ode, SCH, BLOCKED>;
  constexpr int CS = D::CS;
  constexpr int K = D::K, n = D::n, m = D::m, p = D::p, q = D::q, N = D::N, T = D::T, TF = D::TF, P0 = D::P0;
  constexpr int IR = D::IR, OR = D::OR, IRP = D::IRP, ORP = D::ORP, NP = D::NP, KS = D::KS;
  constexpr int LC = D::LC, NSTG = D::NSTG, STG_LD = D::STG_LD;
  static_assert(N == Ode::NIN, "ODE input size mismatch");
  static_assert(G * CS <= 64 * 8, "group too large");
  (void)m; (void)p; (void)ORP;

  extern __shared__ __attribute__((aligned(16))) double lds[];
  // STAGE 4: [tables wave 0 | tables wave 1 | body wave 0 | body wave 1]; the ODE stage of the pair uses both bodies
  const int wave = (STAGE == 4) ? int(threadIdx.x >> 6) : 0;
  double* tabL = lds + (STAGE == 4 ? wave * D::TABSZ : 0);                       // weight tables (persistent)
  double* body = lds + (STAGE == 4 ? 2 * D::TABSZ + wave * D::DENSE : D::TABSZ);   // [slot buffer | dense scratch], aliased by the ODE staging rows
  double* slotb = body;
  double* scr = body + D::WSLOTD;
  double* stage = body;
  static_assert(STAGE != 3 || (D::FUSED && LEVEL >= 1), "no fused kernel for this shape / level");
  static_assert(STAGE != 4 || (D::FUSED2 && LEVEL == 2), "no two-wave fused kernel for this shape / level");
  constexpr bool MIR = D::MIRROR && LEVEL >= 1 && STAGE == 1;
#define ASSET_ODE_PIPE 1
  constexpr bool PIPE = ASSET_ODE_PIPE && MIR && LEVEL == 2 && D::STAGED && LC == 64 && G * CS <= 64;
  lds_double* const mirror = (lds_double*)(body + (STAGE == 3 ? D::GF * CS : LC) * STG_LD);   // [G][MSLOT] (ODE stage, MIR)
  static_assert(!MIR || G <= D::GM, "the LDS mirror holds one slot per segment of a group");
  const int lane = (STAGE == 4) ? int(threadIdx.x & 63) : int(threadIdx.x);
  const int lr = lane & 15, lk = lane >> 4;
  // the scheme's weight tables are read with lane-dependent indices all over the kernel: keep them in LDS
  // (ODE stage: the copy is finished just before its first use in P2, so its latency hides behind P0 / P1)
  constexpr int NTAB = (D::TABSZ + 63) / 64;
  double tabv[NTAB];
#pragma unroll
  for (int t = 0; t < NTAB; t++)
    tabv[t] = (lane + 64 * t < D::TABSZ) ? reinterpret_cast<const double*>(&d_lgl_tab[D::TAB])[lane + 64 * t] : 0.0;
  auto publish_tables = [&]() {
#pragma unroll
    for (int t = 0; t < NTAB; t++)
      if (lane + 64 * t < D::TABSZ) tabL[lane + 64 * t] = tabv[t];
    wave_lds_sync();
  };
  const LglTab& tab = *reinterpret_cast<const LglTab*>(tabL);

  // this workgroup's share of the mesh: contiguous, balanced (same rule as IndexingData.h:117-146)
  // (STAGE 4: every WAVE has a share; the pair's shares are adjacent)
  const int nshare = int(gridDim.x) * (STAGE == 4 ? 2 : 1), share = int(blockIdx.x) * (STAGE == 4 ? 2 : 1) + wave;
  const int per = a.nseg / nshare, rem = a.nseg % nshare;
  const int wg_first = share * per + min(share, rem);
  const int wg_count = per + (share < rem ? 1 : 0);
  (void)G;

  // ---- per-lane constants of the dense phase, computed once per launch: every index decode, table lookup and
  //      store offset below depends only on the lane, never on the segment
  using LCT = LaneConsts<Ode, D, (LEVEL >= 2 ? 2 : 1)>;
  constexpr int CW = LCT::CW, RPW = LCT::RPW, ROWS = LCT::ROWS, CMAIN = LCT::CMAIN, ZERO = LCT::ZERO;
  constexpr bool BOTH = LCT::BOTH, JFUSE = LCT::JFUSE;
  (void)RPW;
  const int d1c = lane & (CW - 1), d1h = lane / CW;        // column; role 0 -> DI row, 1 -> DC row (when RPW >= 2)
  const bool d1ok = (d1c < CMAIN) && (BOTH || d1h < 2);
  const int d1cc = d1ok ? d1c - (d1c / q) * q : 0;
  // slot offset of dfdy_j[r][cc] for run-time indices (the rare paths; table look-ups)
  auto cj_at = [](int j, int r, int cc) { const int jp = Ode::JPOS[r * N + cc]; return jp >= 0 ? D::w_CJ + j * D::NZJ + jp : ZERO; };
  (void)cj_at;
#if defined(ASSET_TIMING)
  long long tstamp[24];
  int nts = 0;
#define TS() do { if (nts < 24) tstamp[nts++] = clock64(); } while (0)
#define ASSET_PTS_ARGS , tstamp, &nts
#define ASSET_PTS_NONE , nullptr, nullptr
#else
#define TS() do {} while (0)
#define ASSET_PTS_ARGS
#define ASSET_PTS_NONE
#endif
// sub-phase stamps of the dense stage, taken for the second segment of the workgroup (-DASSET_TIMING, tools/dbg_time.py)
#define ASSET_TSG_SEG 1
#define TSG() do { if (g == ASSET_TSG_SEG) TS(); } while (0)
  if constexpr (STAGE >= 3) TS();   // (timing builds: kernel start)
#if defined(ASSET_WALLCLOCK)
  const long long wall_t0 = wall_clock64();
#endif
  if constexpr (STAGE == 3) {
    // the workgroup's segments are one group (the host sizes the grid so): ODE stage first, while the kernel holds
    // nothing else in registers; its results go to the slots and are read back below once the stores have landed
    if constexpr (LEVEL == 2)
      pipe_ode_group<Ode, D, D::GF>(a, lane, wg_first, min(wg_count, D::GF), a.work + size_t(wg_first) * D::WSLOT, stage, mirror,
                                    &tab, publish_tables ASSET_PTS_ARGS);
    else   // the Jacobian kinds (evalSOE / evalAUG)
      plain_ode_group_l1<Ode, D, D::GF>(a, lane, wg_first, min(wg_count, D::GF), a.work + size_t(wg_first) * D::WSLOT, stage, mirror,
                                        &tab, publish_tables);
  }
  if constexpr (STAGE == 4) {
    const int s0 = share - wave, first0 = s0 * per + min(s0, rem);            // the pair's first segment and count
    const int cnt = (per + (s0 < rem ? 1 : 0)) + (per + (s0 + 1 < rem ? 1 : 0));
    double* const area = lds + 2 * D::TABSZ;                                   // both bodies
    pipe_ode_group2<Ode, D, D::GF2>(a, int(threadIdx.x), first0, min(cnt, D::GF2), a.work + size_t(first0) * D::WSLOT, area,
                                    (lds_double*)(area + D::GF2 * (K + CS) * STG_LD), &tab, publish_tables);
  }
  LaneRecord<LCT> lrec;
  if constexpr (STAGE >= 2 && LEVEL >= 1) {              // computed once per handle (lane_setup_kernel)
    const unsigned int* rec = static_cast<const unsigned int*>(a.lane_consts) +
                              size_t(blockIdx.x % ASSET_LANE_REPLICAS) * (LaneRecord<LCT>::NW * 64);   // this workgroup's copy
#pragma unroll
    for (int k = 0; k < LaneRecord<LCT>::NW; k++) lrec.w[k] = rec[k * 64 + lane];
  }
  const LCT& lc = lrec.lc;
  if constexpr (STAGE == 2) publish_tables();            // after the record loads are in flight: one latency, not two
  if constexpr (STAGE == 3) {
    TS();                                                // (timing builds) record loads issued
    wave_loads_landed();                                 // the slot stores of the ODE stage (and the record loads)
    TS();                                                // ... landed
  }
  // (STAGE 4: pipe_ode_group2 ends with the wait and a barrier)
  const auto& wa = lc.wa;
  const auto& wb = lc.wb;
  const auto& wa2 = lc.wa2;
  const auto& wb2 = lc.wb2;
  const auto& tB = lc.tB;
  const auto& tD = lc.tD;
  const auto& tC = lc.tC;
  const auto& bo = lc.bo;
  const auto& bst = lc.bst;
  const auto& cho = lc.cho;
  const auto& chp = lc.chp;
  const auto& jo = lc.jo;
  const auto& avb = lc.avb;
  const auto& avs = lc.avs;
  const auto& cjo = lc.cjo;
  const auto& bo2 = lc.bo2;
  const auto& hst = lc.hst;
  const auto& jst = lc.jst;
  const double& tE = lc.tE;
  (void)wa2; (void)wb2; (void)chp; (void)bo2; (void)bo; (void)bst; (void)jo; (void)cho;

  for (int g0 = 0; g0 < wg_count; g0 += G) {
    const int seg0 = wg_first + g0;
    TS();
    const int gcount = min(G, wg_count - g0);
    double* Wg = a.work + size_t(seg0) * D::WSLOT;  // ODE result slots of this group's segments (HBM / L2)

    if constexpr (STAGE == 1 && PIPE) {
      // ---------------------------------------------------------------- pipelined ODE stage (see OdeOutPipe above)
      bool first = (g0 == 0);
      pipe_ode_group<Ode, D, G>(a, lane, seg0, gcount, Wg, stage, mirror, &tab, [&]() { if (first) publish_tables(); } ASSET_PTS_NONE);
      TS();
      continue;
    }
    if constexpr (STAGE == 1 && !PIPE) {
    // ------------------------------------------------------------------ P0: gather z = X[Vindex], lam = L[Cindex]
    // Two dependent HBM round trips (index, then value): every index load is issued before the first value load,
    // so the whole gather costs two latencies instead of two per 64 elements.
    {
      constexpr int NZ = (G * IR + 63) / 64, NL = (LEVEL >= 1) ? (G * OR + 63) / 64 : 0;
      const int* vseg = a.vindex + size_t(seg0) * IR;      // this group's Vindex / Cindex columns are contiguous
      const int* cseg = a.cindex + size_t(seg0) * OR;
      int vi[NZ], ci[NL > 0 ? NL : 1];
#pragma unroll
      for (int t = 0; t < NZ; t++) vi[t] = (lane + 64 * t < gcount * IR) ? vseg[lane + 64 * t] : -1;
#pragma unroll
      for (int t = 0; t < NL; t++) ci[t] = (a.L && lane + 64 * t < gcount * OR) ? cseg[lane + 64 * t] : -1;
      double zv[NZ], lv[NL > 0 ? NL : 1];
#pragma unroll
      for (int t = 0; t < NZ; t++) zv[t] = (vi[t] >= 0) ? a.X[vi[t]] : 0.0;
#pragma unroll
      for (int t = 0; t < NL; t++) lv[t] = (ci[t] >= 0) ? a.L[ci[t]] : 0.0;
#pragma unroll
      for (int t = 0; t < NZ; t++) {
        const int e = lane + 64 * t, g = e / IR, r = e - g * IR;
        if (e < gcount * IR) {
          Wg[g * D::WSLOT + D::w_z + r] = zv[t];
          if constexpr (MIR) mirror[g * D::MSLOT + D::m_z + r] = zv[t];
        }
      }
#pragma unroll
      for (int t = 0; t < NL; t++) {
        const int e = lane + 64 * t, g = e / OR, r = e - g * OR;
        if (e < gcount * OR) {
          Wg[g * D::WSLOT + D::w_lam + r] = lv[t];
          if constexpr (MIR) mirror[g * D::MSLOT + D::m_lam + r] = lv[t];
        }
      }
    }
    // (no wait here: P1 reads X itself; the barrier after P1 also covers these stores before P2 reads the slots)

    TS();
    // ------------------------------------------------------------------ P1: cardinal ODE values (and J for LEVEL 1)
    for (int e0 = 0; e0 < gcount * CS; e0 += LC) {
      const int e = e0 + lane;
      if (lane < LC && e < gcount * CS) {
        const int g = e / CS, j = e - g * CS;
        cardinal_eval1<Ode, D, LEVEL, MIR>(Wg + g * D::WSLOT, mirror + g * D::MSLOT, j, (lds_double*)(stage + lane * STG_LD), a.X,
                                           a.vindex + size_t(seg0 + g) * IR);
      }
      if constexpr (LEVEL == 1 && D::STAGED) {
        wave_lds_sync();   // staging rows are LDS-only hand-offs: do not wait for the global stores
        const int npt = min(LC, gcount * CS - e0);
        copy_rows<(D::NZJ > 0 ? D::NZJ : 1)>(stage, STG_LD, D::NZJ > 0 ? npt : 0, e0, lane, [&](int ee, int k) {
          const int g = ee / CS, j = ee - g * CS;
          return Wg + g * D::WSLOT + D::w_CJ + j * D::NZJ + k;
        });
        wave_lds_sync();   // staging rows are LDS-only hand-offs: do not wait for the global stores
      }
    }
    if constexpr (MIR) wave_lds_sync();   // P2 reads the mirror: no wait for the global stores
    else wave_mem_sync();

    TS();
    if (g0 == 0) publish_tables();
    // ------------------------------------------------------------------ P2: interior points
    // (Trapezoidal: the interior point has weight 0; it is not evaluated and its slot sections keep their zeros)
    if constexpr (!D::TRAP)
    for (int e0 = 0; e0 < gcount * K; e0 += LC) {
      const int e = e0 + lane;
      if (lane < LC && e < gcount * K) {
        const int g = e / K, i = e - g * K;
        interior_eval<Ode, D, LEVEL, MIR>(Wg + g * D::WSLOT, mirror + g * D::MSLOT, i, &tab, (lds_double*)(stage + lane * STG_LD));
      }
      if constexpr (LEVEL >= 1 && D::STAGED) {
        wave_lds_sync();   // staging rows are LDS-only hand-offs: do not wait for the global stores
        const int npt = min(LC, gcount * K - e0);
        constexpr int NC = (LEVEL >= 2) ? NSTG : D::NZJ;   // LEVEL 1 has no Hessian part
        copy_rows<(NC > 0 ? NC : 1)>(stage, STG_LD, NC > 0 ? npt : 0, e0, lane, [&](int ee, int k) {
          const int g = ee / K, i = ee - g * K;
          return Wg + g * D::WSLOT + ((k < D::NZJ) ? (D::w_IJ + i * D::NZJ + k) : (D::w_IH + i * D::NZH + (k - D::NZJ)));
        });
        wave_lds_sync();   // staging rows are LDS-only hand-offs: do not wait for the global stores
      }
    }
    if constexpr (MIR) wave_lds_sync();
    else wave_mem_sync();

    TS();
    // ------------------------------------------------------------------ P3: cardinal second derivatives
    if constexpr (LEVEL >= 2) {
      for (int e0 = 0; e0 < gcount * CS; e0 += LC) {
        const int e = e0 + lane;
        if (lane < LC && e < gcount * CS) {
          const int g = e / CS, j = e - g * CS;
          cardinal_eval2<Ode, D, MIR>(Wg + g * D::WSLOT, mirror + g * D::MSLOT, j, &tab, (lds_double*)(stage + lane * STG_LD));
        }
        wave_lds_sync();   // staging rows are LDS-only hand-offs: do not wait for the global stores
        const int npt = D::STAGED ? min(LC, gcount * CS - e0) : 0;
        copy_rows<(NSTG > 0 ? NSTG : 1)>(stage, STG_LD, NSTG > 0 ? npt : 0, e0, lane, [&](int ee, int k) {
          const int g = ee / CS, j = ee - g * CS;
          return Wg + g * D::WSLOT + ((k < D::NZJ) ? (D::w_CJ + j * D::NZJ + k) : (D::w_CH + j * D::NZH + (k - D::NZJ)));
        });
        wave_lds_sync();   // staging rows are LDS-only hand-offs: do not wait for the global stores
      }
    }

    if constexpr (LEVEL == 0) {
      // ---- value only: lanes over (segment, defect row); everything needed is in the workspace slots
      if (a.FX) {
        for (int e = lane; e < gcount * OR; e += 64) {
          const int g = e / OR, jr = e - g * OR;
          const int i = jr / n, k = jr - i * n;
          const double* S = Wg + g * D::WSLOT;
          const double* z = S + D::w_z;
          const double h = z[TF] - z[T];
          double fxv = 0.0;
#pragma unroll
          for (int j = 0; j < CS; j++) fxv += (tab.C[i][j] * z[j * q + k] + (tab.D[i][j] * h) * S[D::w_Cf + j * n + k]);
          fxv += (h * tab.E[i]) * S[D::w_If + i * n + k];
          a.FX[size_t(seg0 + g) * OR + jr] = fxv;
        }
      }
      wave_mem_sync();
    }
    }  // STAGE == 1
    if constexpr (STAGE == 1 || LEVEL == 0) continue;

    TS();
    constexpr int NPRE = (D::WSLOTD + 63) / 64;
    double pre[NPRE];                      // next segment's slot, in flight while the current one is processed
#pragma unroll
#if defined(ASSET_EXP_NOWS)
    for (int t = 0; t < NPRE; t++) pre[t] = 0.001 * (lane + t) + 0.5;
#else
    for (int t = 0; t < NPRE; t++) pre[t] = (t + 1 < NPRE || lane + 64 * t < D::WSLOTD) ? Wg[lane + 64 * t] : 0.0;
#endif
    // (the first slot's loads fly while the constant tiles below are built)
    // ---- per-group constants of the dense scratch (the staging rows aliased it): the rows of DI_i that do not
    //      depend on the segment (tau row, control-interpolation rows, parameter identity rows, zero padding;
    //      LGLDefects.h:417-458), the rank-2 direction d = e_TF - e_T, zero padding of M and DC
    for (int e = lane; e < K * D::NCR * IRP; e += 64) {
      const int i = e / (D::NCR * IRP), rem2 = e - i * D::NCR * IRP;
      const int r = n + rem2 / IRP, c = rem2 % IRP;
      double v = 0.0;
      if (c < IR) {
        if (r == T) v = (c == T) ? (1.0 - tab.s[i]) : ((c == TF) ? tab.s[i] : 0.0);
        else if (r > T && r < q) { if (c < P0 && (c % q) == r) v = tab.U[i][c / q]; }
        else if (r >= q && r < N) { if (c == P0 + (r - q)) v = 1.0; }
      }
      scr[D::s_DIc + e] = v;
    }
    if constexpr (IR < IRP) {                          // padding columns of the state rows (D1 writes every column < IR)
      for (int e = lane; e < K * n * IRP; e += 64) scr[D::s_DIx + e] = 0.0;
    }
    for (int e = lane; e < IRP; e += 64) {
      scr[D::s_R2 + e] = (e == TF) ? 1.0 : ((e == T) ? -1.0 : 0.0);
      scr[D::s_R2 + 2 * IRP + e] = 0.0;
    }
    if constexpr (IR < IRP) {                          // DC: padding columns and rows
      for (int e = lane; e < ORP * D::LDC; e += 64) scr[D::s_DC + e] = 0.0;
    } else {                                           // only the padding rows (columns >= IRP of a row are never read)
      for (int e = lane; e < (ORP - OR) * D::LDC; e += 64) scr[D::s_DC + OR * D::LDC + e] = 0.0;
    }
    if (lane < 2) scr[D::s_Z0 + lane] = 0.0;
    wave_lds_sync();
    // No load may be outstanding when the segment loop is entered (see wave_loads_landed): the first slot's loads have
    // had the constant-tile build to arrive.
    wave_store_fence();

    TS();
    // ------------------------------------------------------------------ P4: per-segment dense phase
    for (int g = 0; g < gcount; g++) {
      // slot: workspace -> LDS (coalesced); the loads were issued one segment ago
#pragma unroll
      for (int t = 0; t < NPRE; t++)
        if (t + 1 < NPRE || lane + 64 * t < D::WSLOTD) slotb[lane + 64 * t] = pre[t];   // only the last row is partial
      wave_lds_sync();
#if !defined(ASSET_EXP_NOWS)
      if (g + 1 < gcount) {
#pragma unroll
        for (int t = 0; t < NPRE; t++)
          pre[t] = (t + 1 < NPRE || lane + 64 * t < D::WSLOTD) ? Wg[(g + 1) * D::WSLOT + lane + 64 * t] : 0.0;
      }
#endif
      const double* S = slotb;
      const double* z = S + D::w_z;
      const double* lam = S + D::w_lam;
      const double h = z[TF] - z[T];
      const size_t seg = size_t(seg0 + g);
      TSG();   // slot in LDS

      if constexpr (LEVEL == 0) continue;

      double* DIx = scr + D::s_DIx;
      const double* DIc = scr + D::s_DIc;
      double* Mt = scr + D::s_M;
      double* DC = scr + D::s_DC;
      double* R2 = scr + D::s_R2;
      double* HI = scr + D::s_HI;

      // ---- D1: one pass over (interior i, state row r) builds, from a single read of dfdy_j per block column c:
      //        DI_i[r][c]  = A_ij [cc==r] + h B_ij J_j[r][cc]        (LGLDefects.h:430-444)
      //        DC[(i,r)][c] = C_ij [cc==r] + h D_ij J_j[r][cc]        (LGLDefects.h:467-482)  -- cardinal part of J
      //      lanes [0,CW) write the DI row, lanes [CW,2CW) the DC row of the same (i,r); row indices are compile-time
      if (d1ok) {
        double wbh[K], wb2h[BOTH ? K : 1];
#pragma unroll
        for (int i = 0; i < K; i++) {
          wbh[i] = wb[i] * h;
          if constexpr (BOTH) wb2h[i] = wb2[i] * h;
        }
        const bool dcrole = (!BOTH && d1h == 1);
        double* dstb = (dcrole ? DC : DIx) + d1c;
        const int dld = dcrole ? D::LDC : IRP;
        double jvr[n];                                   // every LDS read is issued before the first write: the compiler
#pragma unroll                                           // cannot reorder them itself (it must assume the tiles alias the slot)
        for (int r = 0; r < n; r++) jvr[r] = S[cjo[r]];
#pragma unroll
        for (int row = 0; row < ROWS; row++) {
          const int i = row / n, r = row - i * n;
          const double jv = jvr[r];
          double v = wbh[i] * jv;
          if (d1cc == r) v += wa[i];
          dstb[row * dld] = v;
          if constexpr (BOTH) {
            double v2 = wb2h[i] * jv;
            if (d1cc == r) v2 += wa2[i];
            DC[row * D::LDC + d1c] = v2;
          }
        }
      }
      if constexpr (CMAIN < IR) {                        // parameter columns and columns beyond one wave pass
        for (int e = lane; e < ROWS * (IR - CMAIN); e += 64) {
          const int row = e / (IR - CMAIN), c2 = CMAIN + e - row * (IR - CMAIN);
          const int i = row / n, r = row - i * n;
          double vi = 0.0, vc = 0.0;
          if (c2 < P0) {
            const int j2 = c2 / q, cc2 = c2 - j2 * q;
            const double jv = S[cj_at(j2, r, cc2)];
            vi = (tab.B[i][j2] * h) * jv;
            vc = (tab.D[i][j2] * h) * jv;
            if (cc2 == r) { vi += tab.A[i][j2]; vc += tab.C[i][j2]; }
          } else {
            for (int jj = 0; jj < CS; jj++) {
              const double jv = S[cj_at(jj, r, q + (c2 - P0))];
              vi += (tab.B[i][jj] * h) * jv;
              vc += (tab.D[i][jj] * h) * jv;
            }
          }
          DIx[row * IRP + c2] = vi;
          DC[row * D::LDC + c2] = vc;
        }
      }
      wave_lds_sync();
      TSG();   // D1: DI / DC tiles
      // time columns: DI rows -+ sum_j B_ij f_j (LGLDefects.h:446-450), DC rows -+ (sum_j D_ij f_j + E_i f^_i) (:484-500)
      double fx_hold = 0.0;
      auto time_columns = [&](int e, auto own_) {
        constexpr bool own = decltype(own_)::value;      // first pass: the weights are the precomputed per-lane ones
        const int i = e / n, r = e - i * n;
        double fv[CS], zv[CS];
        const double fi = S[D::w_If + i * n + r];
#pragma unroll
        for (int jj = 0; jj < CS; jj++) { fv[jj] = S[D::w_Cf + jj * n + r]; zv[jj] = z[jj * q + r]; }
        const double dit = DIx[e * IRP + T], ditf = DIx[e * IRP + TF], dct = DC[e * D::LDC + T], dctf = DC[e * D::LDC + TF];
        double sb = 0.0, sd = (own ? tE : tab.E[i]) * fi;
#pragma unroll
        for (int jj = 0; jj < CS; jj++) {
          sb += (own ? tB[jj] : tab.B[i][jj]) * fv[jj];
          sd += (own ? tD[jj] : tab.D[i][jj]) * fv[jj];
        }
        DIx[e * IRP + T] = dit - sb;
        DIx[e * IRP + TF] = ditf + sb;
        DC[e * D::LDC + T] = dct - sd;
        DC[e * D::LDC + TF] = dctf + sd;
        double fxv = h * sd;                             // defect value of row (i,r)  (LGLDefects.h:96-103)
#pragma unroll
        for (int jj = 0; jj < CS; jj++) fxv += (own ? tC[jj] : tab.C[i][jj]) * zv[jj];
        if constexpr (own) fx_hold = fxv;                // stored with the segment's other results (after the load fence)
        else if (a.FX) a.FX[seg * OR + e] = fxv;
      };
      if (lane < ROWS) time_columns(lane, std::true_type{});
      if constexpr (ROWS > 64) {
        wave_loads_landed();                             // (stores follow: see below)
        for (int e = lane + 64; e < ROWS; e += 64) time_columns(e, std::false_type{});
      }
      wave_lds_sync();
      TSG();   // time columns, FX

      // ---- D2: A fragments (DI_i^T tiles) for every tile row, reused by all three products
      double av[D::TI][K][KS];
#pragma unroll
      for (int ct = 0; ct < D::TI; ct++)
#pragma unroll
        for (int i = 0; i < K; i++)
#pragma unroll
          for (int kk = 0; kk < KS; kk++)
            av[ct][i][kk] = scr[avb[kk] + i * avs[kk] + 16 * ct];

      // Small shapes keep every accumulator tile until D6 (the stores then share a handful of lane-condition
      // branches); wide ones store each tile as it completes -- holding them all would spill.
#define ASSET_HOLD_TILES 6
      constexpr bool HOLD = (D::NTH + D::TI * D::TJ) <= ASSET_HOLD_TILES;
      constexpr bool CFULL = (IR == IRP);
      d4 accH[(LEVEL >= 2 && HOLD) ? D::NTH : 1];
      d4 accJ[HOLD ? D::TI * D::TJ : 1];
      // Where an entry goes: its slot of the segment's KKT block, or (ASM) the solver's value array through the
      // slot -> location map.  The map entries of a tile are loaded before its product so the stores never wait for
      // them; locations several slots share (boundary nodes of neighbouring segments, phase parameters) take a
      // no-return f64 atomic, the others a plain store (10 M atomics per evaluation measured ~65 us on their own).
      // ASM: the map is stored in fragment order -- entry (f, lane) of a segment, f = 4*tile + v with the H tiles
      // first -- so a lane reads its own locations with coalesced loads and needs no slot arithmetic; -1 marks an
      // accumulator entry that is no KKT slot (upper triangle of a diagonal tile, padding).
      constexpr int NFRAG = (D::NTH + D::TI * D::TJ) * 4;
      double* const kkt_dst = ASM ? a.values : (a.KKT ? a.KKT + seg * size_t(D::KSTRIDE) : nullptr);
      const int* const kmap_seg = ASM ? a.kmap + seg * size_t(NFRAG) * 64 + lane : nullptr;
      int hmap[ASM ? D::NTH : 1][4], jmap[ASM ? D::TI * D::TJ : 1][4];
      auto load_hmap = [&](int tix) {
        if constexpr (ASM) {
#pragma unroll
          for (int v = 0; v < 4; v++) hmap[tix][v] = kmap_seg[(tix * 4 + v) * 64];
        }
      };
      auto load_jmap = [&](int t) {
        if constexpr (ASM) {
#pragma unroll
          for (int v = 0; v < 4; v++) jmap[t][v] = kmap_seg[((D::NTH + t) * 4 + v) * 64];
        }
      };
      auto put_asm = [&](int off, double val) { asm_put(a, kkt_dst, off, val); };   // map encoding: defect_dims.h
      auto store_H_tile = [&](int rt, int ct, const d4& acc) {   // entry v: row r = 16rt + lr, column c = 16ct + lk + 4v
        const int tix = rt * (rt + 1) / 2 + ct;
        if constexpr (ASM) {
#pragma unroll
          for (int v = 0; v < 4; v++) put_asm(hmap[tix][v], acc[v]);
        } else if (rt > ct) {
          if (CFULL || 16 * rt + lr < IR) {
#pragma unroll
            for (int v = 0; v < 4; v++) kkt_dst[hst[tix][v]] = acc[v];
          }
        } else {
#pragma unroll
          for (int v = 0; v < 4; v++)
            if (lr >= lk + 4 * v && (CFULL || rt + 1 < D::TI || 16 * rt + lr < IR)) kkt_dst[hst[tix][v]] = acc[v];
        }
      };
      auto store_J_tile = [&](int jt, int ct, const d4& acc) {   // entry v: defect row 16jt + lr, column c
        if constexpr (ASM) {
#pragma unroll
          for (int v = 0; v < 4; v++) put_asm(jmap[ct * D::TJ + jt][v], acc[v]);
        } else if (16 * jt + lr < OR) {
#pragma unroll
          for (int v = 0; v < 4; v++)
            if (CFULL || ct + 1 < D::TI || 16 * ct + lk + 4 * v < IR) kkt_dst[jst[ct * D::TJ + jt][v]] = acc[v];
        }
      };
      if constexpr (ASM) {                                 // all of the segment's map entries, ahead of the products
        if constexpr (LEVEL >= 2) {
#pragma unroll
          for (int tix = 0; tix < D::NTH; tix++) load_hmap(tix);
        }
#pragma unroll
        for (int t = 0; t < D::TI * D::TJ; t++) load_jmap(t);
      }
      // Optional load fence (defect_dims.h: wave_store_fence, off by default): called once per segment, after the last
      // load of the iteration has been issued and before its first store, at one unconditional call site per
      // instantiation (the waitcnt pass is path-insensitive: a fence under a lane- or pointer-condition does not count).
      constexpr int FENCE_AT = JFUSE ? 0 : (HOLD ? 2 : 1);
      // The product phases run at raised wave priority: the two waves of a SIMD otherwise walk through the same
      // phases nearly in step and the LDS / VALU / MFMA pipes take turns; any asymmetry in arbitration helps (measured
      // 0.5 us, the same for every priority assignment tried).
      __builtin_amdgcn_s_setprio(1);
      // ---- D3: M_i^T = DI_i^T [hE_i H^_i | E_i g^_i]; column N of the product is sum_b E_i g^_i[b] DI_i[b,c]
      if constexpr (JFUSE) {
        double jacc[D::TI][4];                           // interior part of J^T, entry v = (c = 16ct + lk + 4v, jr = lr)
        double hi_acc[D::TI][4];
#pragma unroll
        for (int ct = 0; ct < D::TI; ct++)
#pragma unroll
          for (int v = 0; v < 4; v++) hi_acc[ct][v] = 0.0, jacc[ct][v] = 0.0;
        double bvall[K][KS];                             // all B operands are read before the first M^T write-back
#pragma unroll
        for (int i = 0; i < K; i++)
#pragma unroll
          for (int kk = 0; kk < KS; kk++) bvall[i][kk] = S[bo2[i][kk]];
#pragma unroll
        for (int i = 0; i < K; i++) {
          const double sc = (lr == 15) ? tab.E[i] : h * tab.E[i];   // the g^ column is scaled by E_i, the others by h E_i
          const bool isj = (lr >= i * n && lr < i * n + n);
          const int hr = (isj || lr == 15) ? -1 : ((lr < i * n) ? lr : lr - n);   // rank among the H^ lanes
          const bool ish = (hr >= 0 && hr < N);
          const int mcol = (hr >= 0 && hr < NP) ? i * NP + hr : K * NP;          // k-padding columns get zeros, others the spare
          const double jf = isj ? 1.0 : 0.0;
          double bv[KS];
#pragma unroll
          for (int kk = 0; kk < KS; kk++) bv[kk] = bvall[i][kk] * sc;
#pragma unroll
          for (int ct = 0; ct < D::TI; ct++) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < KS; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ct][i][kk], bv[kk], acc, 0, 0, 0);
            // acc[v] = row c = 16ct + lk + 4v of DI_i^T times this lane's column: an M_i^T column, E_i g^_i . DI_i (lane
            // 15, summed over i into HI) or row (i, lr - i*n) of the interior part of J (LGLDefects.h:452-500)
#pragma unroll
            for (int v = 0; v < 4; v++) {
              hi_acc[ct][v] += acc[v];
              jacc[ct][v] = fma(jf, acc[v], jacc[ct][v]);
            }
            // only the lanes on an H^ (or k-padding) column write M^T back: letting the others write a common spare cell
            // costs an 8-way same-address conflict per store (measured: 224 conflict cycles per segment)
            if (hr >= 0 && hr < NP) {
#pragma unroll
              for (int v = 0; v < 4; v++) Mt[(16 * ct + lk + 4 * v) * D::LDM + mcol] = (N == NP || ish) ? acc[v] : 0.0;
            }
          }
        }
        if (lr == 15) {
#pragma unroll
          for (int ct = 0; ct < D::TI; ct++)
#pragma unroll
            for (int v = 0; v < 4; v++) HI[16 * ct + lk + 4 * v] = hi_acc[ct][v];
        }
        // J^T = interior part + cardinal part DC^T; stored right away (its registers are free for the H products)
        if constexpr (FENCE_AT == 0) wave_store_fence();
        if (kkt_dst) {
#pragma unroll
          for (int ct = 0; ct < D::TI; ct++) {
            d4 acc;
#pragma unroll
            for (int v = 0; v < 4; v++) acc[v] = jacc[ct][v] + DC[lr * D::LDC + 16 * ct + lk + 4 * v];   // row 15: zero padding
            store_J_tile(0, ct, acc);
          }
        }
      } else if constexpr (LEVEL >= 2) {
        double hi_acc[D::TI][4];
#pragma unroll
        for (int ct = 0; ct < D::TI; ct++)
#pragma unroll
          for (int v = 0; v < 4; v++) hi_acc[ct][v] = 0.0;
        double bvall[K][D::MT][KS];                      // all B operands are read before the first M^T write-back
#pragma unroll
        for (int i = 0; i < K; i++)
#pragma unroll
          for (int mt = 0; mt < D::MT; mt++)
#pragma unroll
            for (int kk = 0; kk < KS; kk++) bvall[i][mt][kk] = S[bo[mt][kk] + i * bst[mt][kk]];
#pragma unroll
        for (int i = 0; i < K; i++) {
          const double he = h * tab.E[i];
#pragma unroll
          for (int mt = 0; mt < D::MT; mt++) {
            const int acol = 16 * mt + lr;               // column of [hE H^ | E g^]
            double bv[KS];
#pragma unroll
            for (int kk = 0; kk < KS; kk++) bv[kk] = bvall[i][mt][kk] * ((acol == N) ? tab.E[i] : he);
#pragma unroll
            for (int ct = 0; ct < D::TI; ct++) {
              d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int kk = 0; kk < KS; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ct][i][kk], bv[kk], acc, 0, 0, 0);
              // acc[v] = (M_i^T)[c = 16ct + lk + 4v][acol].  Every lane accumulates; only the lanes owning column N publish
              // hi_acc and only the lanes on a column < NP write M^T back.
              if (mt == N / 16) {
#pragma unroll
                for (int v = 0; v < 4; v++) hi_acc[ct][v] += acc[v];
              }
              if (acol < NP) {                           // (a common spare cell for the other lanes would be an 8-way write conflict)
#pragma unroll
                for (int v = 0; v < 4; v++) Mt[(16 * ct + lk + 4 * v) * D::LDM + i * NP + acol] = (acol < N) ? acc[v] : 0.0;
              }
            }
          }
        }
        if ((N & 15) == lr) {                            // the lanes that own column N
#pragma unroll
          for (int ct = 0; ct < D::TI; ct++)
#pragma unroll
            for (int v = 0; v < 4; v++) HI[16 * ct + lk + 4 * v] = hi_acc[ct][v];
        }
      }
      if constexpr (LEVEL >= 2) {
        wave_lds_sync();
        TSG();   // D2 + D3: fragments, M product
        // full time-partial vector HTpar (LGLDefects.h:403-411, 504-505) -> rank-2 rows:
        //   H += d HT^T + HT d^T  with d = e_TF - e_T   (the four updates of LGLDefects.h:508-511)
        const double ih = 1.0 / h;
        for (int c = lane; c < IRP; c += 64) {
          const int jn = c / q;
          double v = HI[c] + S[(c < P0) ? D::w_Cg + jn * N + (c - jn * q) : ZERO] * ih;   // HI is zero on padding columns
          if constexpr (p > 0) {
            if (c >= P0 && c < IR) {
#pragma unroll
              for (int j = 0; j < CS; j++) v += S[D::w_Cg + j * N + q + (c - P0)] * ih;
            }
          }
          R2[IRP + c] = v;          // A-side row 1 / B-side row 0 share this copy
        }
        wave_lds_sync();
        TSG();   // rank-2 rows
      }

      // ---- D4: H (lower-triangle tiles) and J^T
      if constexpr (FENCE_AT == 1) wave_store_fence();    // tiles are stored as they complete
      if constexpr (LEVEL >= 2) {
        // rank-2 time fragments: k=0 -> (A: d, B: HT), k=1 -> (A: HT, B: d), k=2,3 -> 0
        double a2[D::TI], b2[D::TI];
#pragma unroll
        for (int t = 0; t < D::TI; t++) {
          // the k index of the lane picks the row: no select (row 2 holds zeros)
          a2[t] = R2[(lk == 0 ? 0 : (lk == 1 ? IRP : 2 * IRP)) + 16 * t + lr];
          b2[t] = R2[(lk == 0 ? IRP : (lk == 1 ? 0 : 2 * IRP)) + 16 * t + lr];
        }
#pragma unroll
        for (int rt = 0; rt < D::TI; rt++) {
          double bm[K][KS];
#pragma unroll
          for (int i = 0; i < K; i++)
#pragma unroll
            for (int kk = 0; kk < KS; kk++) bm[i][kk] = Mt[(16 * rt + lr) * D::LDM + i * NP + 4 * kk + lk];
#pragma unroll
          for (int ct = 0; ct <= rt; ct++) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            // cardinal diagonal / parameter blocks (LGLDefects.h:386-402) enter as the initial accumulator value
            if (tiles_share_node<D>(ct, rt)) {
#pragma unroll
              for (int v = 0; v < 4; v++) {
                double val = S[cho[rt * (rt + 1) / 2 + ct][v]];
                if constexpr (p > 0) {
                  const int o = chp[rt * (rt + 1) / 2 + ct][v];
                  if (o >= 0) {
#pragma unroll
                    for (int j = 0; j < CS; j++) val += S[D::w_CH + j * D::NZH + o];
                  }
                }
                acc[v] = val;
              }
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[ct], b2[rt], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < K; i++)
#pragma unroll
              for (int kk = 0; kk < KS; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ct][i][kk], bm[i][kk], acc, 0, 0, 0);
            if constexpr (HOLD) accH[rt * (rt + 1) / 2 + ct] = acc;
            else if (kkt_dst) store_H_tile(rt, ct, acc);
          }
        }
      }
      if constexpr (!JFUSE) {
#pragma unroll
      for (int jt = 0; jt < D::TJ; jt++) {
        double bj[K][KS];                                // (hE_i J^_i)^T[aa][jr], non-zero only on interior i's rows
#pragma unroll
        for (int i = 0; i < K; i++)
#pragma unroll
          for (int kk = 0; kk < KS; kk++) {
            bj[i][kk] = (h * tab.E[i]) * S[jo[jt][i][kk]];
          }
#pragma unroll
        for (int ct = 0; ct < D::TI; ct++) {
          // cardinal part of J^T (DC^T) is the initial accumulator value: entry v is (c = 16ct + lk + 4v, jr = 16jt + lr)
          d4 acc;
#pragma unroll
          for (int v = 0; v < 4; v++) acc[v] = DC[(16 * jt + lr) * D::LDC + 16 * ct + lk + 4 * v];   // rows >= OR: zero padding
#pragma unroll
          for (int i = 0; i < K; i++)
#pragma unroll
            for (int kk = 0; kk < KS; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ct][i][kk], bj[i][kk], acc, 0, 0, 0);
          if constexpr (HOLD) accJ[ct * D::TJ + jt] = acc;
          else if (kkt_dst) store_J_tile(jt, ct, acc);
        }
      }
      }

      __builtin_amdgcn_s_setprio(0);
      TSG();   // D4: H (and J) products
      // ---- D5: adjoint gradient  g = J^T lam  without touching the J tile:
      //      interior part  h * sum_i E_i g^_i^T DI_i  (= h * HI), cardinal part = DC^T lam
      if constexpr (FENCE_AT == 2) wave_store_fence();
      if (a.FX && lane < ROWS) a.FX[seg * OR + lane] = fx_hold;
      if (a.AGX) {
        for (int c = lane; c < IR; c += 64) {
          double v = 0.0;
          if constexpr (LEVEL >= 2) {
            v = h * HI[c];
          } else {                                       // no M product: g^_i = J^_i^T lam_i comes from the ODE stage
#pragma unroll
            for (int i = 0; i < K; i++)
#pragma unroll
              for (int b = 0; b < N; b++)
                v += ((h * tab.E[i]) * S[D::w_Ig + i * N + b]) *
                     ((b < n) ? DIx[(i * n + b) * IRP + c] : DIc[(i * D::NCR + (b - n)) * IRP + c]);
          }
#pragma unroll
          for (int jr = 0; jr < OR; jr++) v += lam[jr] * DC[jr * D::LDC + c];
          a.AGX[seg * IR + c] = v;
        }
      }

      // ---- D6: store.  Entry (v) of a tile held by this lane: block column c = 16*ct + lk + 4v,
      //      row (H) r = 16*rt + lr or (J) jr = 16*jt + lr; 16 consecutive lanes cover 128 contiguous bytes.
      if (kkt_dst) {
        if constexpr (!HOLD) {
          if (LEVEL < 2 && !ASM && !(a.flags & 1)) {       // Jacobian-only kinds write the Hessian slots as zero (unless told not to)
            const d4 zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int rt = 0; rt < D::TI; rt++)
#pragma unroll
              for (int ct = 0; ct <= rt; ct++) store_H_tile(rt, ct, zero);
          }
        } else {
        // Lanes are grouped by the condition that makes their entry a KKT slot, so a handful of branches cover all
        // stores: off-diagonal H tiles are complete, a diagonal tile's entry v is kept when r >= c (lr >= lk + 4v),
        // a J tile's when its defect row exists (lr < OR - 16jt); padded sizes add the c < IR / r < IR tests.
        auto hval = [&](int tix, int v) { if constexpr (LEVEL >= 2) return accH[tix][v]; else return 0.0; };
        if constexpr (ASM) {
          if constexpr (LEVEL >= 2) {
#pragma unroll
            for (int rt = 0; rt < D::TI; rt++)
#pragma unroll
              for (int ct = 0; ct <= rt; ct++) store_H_tile(rt, ct, accH[rt * (rt + 1) / 2 + ct]);
          }
          if constexpr (!JFUSE) {
#pragma unroll
            for (int jt = 0; jt < D::TJ; jt++)
#pragma unroll
              for (int ct = 0; ct < D::TI; ct++) store_J_tile(jt, ct, accJ[ct * D::TJ + jt]);
          }
        } else {
        if (LEVEL >= 2 || !(a.flags & 1)) {   // Jacobian-only kinds write the Hessian slots as zero (unless told not to)
#pragma unroll
          for (int rt = 0; rt < D::TI; rt++)
#pragma unroll
            for (int ct = 0; ct < rt; ct++) {
              const int tix = rt * (rt + 1) / 2 + ct;
              if (CFULL || 16 * rt + lr < IR) {            // columns of a tile left of the diagonal are always < IR
#pragma unroll
                for (int v = 0; v < 4; v++) kkt_dst[hst[tix][v]] = hval(tix, v);
              }
            }
#pragma unroll
          for (int v = 0; v < 4; v++) {
            if (lr >= lk + 4 * v) {
#pragma unroll
              for (int t = 0; t < D::TI; t++) {
                const int tix = t * (t + 1) / 2 + t;
                if (CFULL || t + 1 < D::TI || 16 * t + lr < IR) kkt_dst[hst[tix][v]] = hval(tix, v);
              }
            }
          }
        }
        if constexpr (!JFUSE) {                            // (JFUSE stored the J tile right after the M product)
#pragma unroll
        for (int jt = 0; jt < D::TJ; jt++) {
          if (16 * jt + lr < OR) {
#pragma unroll
            for (int ct = 0; ct < D::TI; ct++)
#pragma unroll
              for (int v = 0; v < 4; v++)
                if (CFULL || ct + 1 < D::TI || 16 * ct + lk + 4 * v < IR)
                  kkt_dst[jst[ct * D::TJ + jt][v]] = accJ[ct * D::TJ + jt][v];
          }
        }
        }
        }
        }
      }
      wave_lds_sync();  // the next segment rewrites the DI / M / DC tiles
      TSG();   // D5 + D6: adjoint gradient, stores
    }
    TS();
  }
#if defined(ASSET_WALLCLOCK)   // (tuning builds) 100 MHz wall-clock at the start and the end of every workgroup, left in FX
  if (lane == 0 && a.FX && wg_count > 0) {
    a.FX[size_t(wg_first) * OR + 0] = double(wall_t0);
    a.FX[size_t(wg_first) * OR + 1] = double(wall_clock64());
    a.FX[size_t(wg_first) * OR + 2] = double(__builtin_amdgcn_s_getreg(63492));   // HW_ID
    a.FX[size_t(wg_first) * OR + 3] = double(__builtin_amdgcn_s_getreg(63508));   // XCC_ID
  }
#endif
#if defined(ASSET_TIMING)
  if (blockIdx.x == 7 && lane == 0 && a.FX)
    for (int t = 0; t + 1 < nts; t++) a.FX[size_t(wg_first) * OR + t] = double(tstamp[t + 1] - tstamp[t]);
#endif
#undef TS
}

// The kernel proper: the body above for the variants the shape has (lgl_variant_valid), nothing otherwise.
template <class Ode, int SCH, bool BLOCKED, int G, int LEVEL, int STAGE, bool ASM = false>
__global__ __launch_bounds__(STAGE == 4 ? 128 : 64, STAGE >= 2 ? (Dims<Ode, SCH, BLOCKED>::lds_bytes_dense() * 4 * ASSET_DENSE_WAVES_PER_SIMD <= 160 * 1024
                                                   ? ASSET_DENSE_WAVES_PER_SIMD : 1)   // LDS-bound to one wave per SIMD anyway: take the registers
                                             : ASSET_ODE_WAVES_PER_SIMD) void lgl_defect_kernel(EvalArgs a) {
  if constexpr (lgl_variant_valid<Dims<Ode, SCH, BLOCKED>, LEVEL, STAGE>()) lgl_defect_body<Ode, SCH, BLOCKED, G, LEVEL, STAGE, ASM>(a);
}

}  // namespace asset_hip
