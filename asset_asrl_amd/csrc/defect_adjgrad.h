// Value + adjoint gradient (constraints_adjointgradient, ComputableBase.h:315-335: what PSIOPT's evalRHS calls in every
// line-search step) without forming a Jacobian.  The reference multiplies the dense J it has just built by lam; the
// kernels of defect_kernels.h did the same through the derivative-level-1 path (ODE-stage launch with every Jacobian,
// dense-stage launch: 34.6 us for 10 000 Reentry-LGL7 segments, against 42.9 us for the whole evalKKT).  But J^T lam is
// a sum of vector-Jacobian products of the ODE, which the generated functor delivers directly (its `g` output; with the
// J / H outputs discarded the compiler drops their arithmetic):
//   g^_i = (d f^_i)^T lam_i                                   interior points      (N-vector)
//   w_j  = sum_i (h D_ij lam_i + h^2 E_i B_ij g^_i[0:n])      cardinal adjoint weights (LGLDefects.h:369-374)
//   g_j  = (d f_j)^T w_j                                      cardinal nodes       (N-vector)
//   (J^T lam)[node j, state k]   = g_j[k] + sum_i (C_ij lam_i[k] + h E_i A_ij g^_i[k])
//   (J^T lam)[node j, time]      = g_j[T]  -/+ at t_0 / t_f:  sum_i lam_i.(sum_j D_ij f_j + E_i f^_i)
//                                            + sum_i h E_i ( -/+ g^_i[0:n].(sum_j B_ij f_j) + (1 - s_i | s_i) g^_i[T] )
//   (J^T lam)[node j, control k] = g_j[n+1+k] + sum_i h E_i U_ij g^_i[n+1+k]
//   (J^T lam)[parameter k]       = sum_j g_j[q+k] + sum_i h E_i g^_i[q+k]
// which is LGLDefects.h:218-260 transposed and applied to lam (DI_i^T g^_i written out row by row, :167-216).  Trapezoidal:
// the same with E = 0.  One launch, lane <-> evaluation point, one wave per workgroup, everything between the phases in LDS.
#pragma once
#include "defect_dims.h"

namespace asset_hip {

template <class D>
struct AdjDims {
  static constexpr int CS = D::CS, K = D::K, n = D::n, N = D::N;
  static constexpr int m_z = 0, m_lam = D::IR, m_Cf = m_lam + D::OR, m_If = m_Cf + CS * n, m_Ig = m_If + K * n, m_Cg = m_Ig + K * N;
  static constexpr int m_S = m_Cg + CS * N;        // per interior: lam_i.(sum_j D_ij f_j + E_i f^_i), g^_i[0:n].(sum_j B_ij f_j)
  static constexpr int MS = (m_S + 2 * K) | 1;     // per-segment LDS slot: z | lam | f_j | f^_i | g^_i | g_j | time sums
  // segments per workgroup: as many as lanes serve (64 / CS) within 64 KiB of LDS
  static constexpr int GP_LDS = int((64 * 1024 - size_t(D::TABSZ) * 8) / (size_t(MS) * 8));
  static constexpr int GP = (GP_LDS < 1) ? 1 : (GP_LDS < 64 / CS ? GP_LDS : 64 / CS);
  static constexpr size_t lds_bytes() { return size_t(D::TABSZ + GP * MS) * 8; }
};

template <bool WANT_G>
struct OdeOutFG {   // f and (WANT_G) g = J^T lam of one point into LDS; everything else of the functor's outputs is dropped
  lds_double* f_;
  lds_double* g_;
  __device__ void f(int k, double v) { f_[k] = v; }
  __device__ void J(int, int, double) {}
  __device__ void g(int i, double v) { if constexpr (WANT_G) g_[i] = v; }
  __device__ void H(int, int, double) {}
  __device__ void save(int, double) {}
};

struct OdeOutG {    // only g (the value of a cardinal node is already in LDS)
  lds_double* g_;
  __device__ void f(int, double) {}
  __device__ void J(int, int, double) {}
  __device__ void g(int i, double v) { g_[i] = v; }
  __device__ void H(int, int, double) {}
  __device__ void save(int, double) {}
};

// ADJ = false: the value-only kind (constraints, evalOCC) through the same phases without the gradient parts -- nothing
// goes through the workspace (the value path of lgl_defect_kernel writes f_j and f^_i to the slots and reads them back).
template <class Ode, int SCH, bool BLOCKED, bool ADJ = true>
__global__ __launch_bounds__(64) void lgl_adjgrad_kernel(EvalArgs a) {
  using D = Dims<Ode, SCH, BLOCKED>;
  using AD = AdjDims<D>;
  constexpr int CS = D::CS, K = D::K, n = D::n, m = D::m, p = D::p, q = D::q, N = D::N, T = D::T, TF = D::TF, P0 = D::P0;
  constexpr int IR = D::IR, OR = D::OR, GP = AD::GP, MS = AD::MS;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  lds_double* const tabL = (lds_double*)lds;
  lds_double* const mir = (lds_double*)(lds + D::TABSZ);
  const int lane = threadIdx.x;
  const int seg0 = int(blockIdx.x) * GP;
  const int gcount = min(GP, a.nseg - seg0);
  // ---- gather: every index of the group first, then every value, then the LDS writes -- two trips to memory for the whole group
  //      (index rows that are runs, EvalArgs::affine: one).  As loops over the entries with an index -> value chain inside, the
  //      compiler had left 8 + 4 + 2 iterations of load, wait, load, wait: 26 dependent round trips at the head of a 6 us kernel.
  {
    constexpr int NZ = (GP * IR + 63) / 64, NL = (GP * OR + 63) / 64, NTAB = (D::TABSZ + 63) / 64;
    int vi[NZ], ci[NL];
    if (a.affine) {
#pragma unroll
      for (int t = 0; t < NZ; t++) {
        const int e = lane + 64 * t, g = e / IR, r = e - g * IR;
        vi[t] = (e < gcount * IR) ? a.aff_v0 + (seg0 + g) * a.aff_vs + r : -1;
      }
#pragma unroll
      for (int t = 0; t < NL; t++) {
        const int e = lane + 64 * t, g = e / OR, r = e - g * OR;
        ci[t] = (e < gcount * OR) ? a.aff_c0 + (seg0 + g) * a.aff_cs + r : -1;
      }
    } else {
      const int* vseg = a.vindex + size_t(seg0) * IR;      // (the group's Vindex / Cindex columns are contiguous)
      const int* cseg = a.cindex + size_t(seg0) * OR;
#pragma unroll
      for (int t = 0; t < NZ; t++) vi[t] = (lane + 64 * t < gcount * IR) ? vseg[lane + 64 * t] : -1;
#pragma unroll
      for (int t = 0; t < NL; t++) ci[t] = (lane + 64 * t < gcount * OR) ? cseg[lane + 64 * t] : -1;
    }
    double tabv[NTAB];
#pragma unroll
    for (int t = 0; t < NTAB; t++)
      tabv[t] = (lane + 64 * t < D::TABSZ) ? reinterpret_cast<const double*>(&d_lgl_tab[D::TAB])[lane + 64 * t] : 0.0;
    double zv[NZ], lv[NL];
#pragma unroll
    for (int t = 0; t < NZ; t++) zv[t] = (vi[t] >= 0) ? a.X[vi[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NL; t++) lv[t] = (ADJ && a.L && ci[t] >= 0) ? a.L[ci[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NTAB; t++)
      if (lane + 64 * t < D::TABSZ) tabL[lane + 64 * t] = tabv[t];
#pragma unroll
    for (int t = 0; t < NZ; t++) {
      const int e = lane + 64 * t, g = e / IR, r = e - g * IR;
      if (e < gcount * IR) mir[g * MS + AD::m_z + r] = zv[t];
    }
#pragma unroll
    for (int t = 0; t < NL; t++) {
      const int e = lane + 64 * t, g = e / OR, r = e - g * OR;
      if (e < gcount * OR) mir[g * MS + AD::m_lam + r] = lv[t];
    }
    if constexpr (D::TRAP) {   // no interior evaluation: its value and gradient read as zero
      for (int e = lane; e < gcount * (K * n + K * N); e += 64) {
        const int g = e / (K * n + K * N), r = e - g * (K * n + K * N);
        mir[g * MS + AD::m_If + r] = 0.0;
      }
    }
  }
  wave_lds_sync();
  const LglTab& tab = *reinterpret_cast<const LglTab*>(lds);
  // ---- cardinal values f_j
  if (lane < gcount * CS) {
    const int g = lane / CS, j = lane - g * CS;
    lds_double* M = mir + g * MS;
    CardIn<D, const lds_double*> in{M + AD::m_z, nullptr, j, nullptr};
    OdeOutFG<false> out{M + AD::m_Cf + j * n, nullptr};
    Ode::f(in, out);
  }
  wave_lds_sync();
  // ---- interior points: f^_i and g^_i = (d f^_i)^T lam_i
  if constexpr (!D::TRAP) {
    if (lane < gcount * K) {
      const int g = lane / K, i = lane - g * K;
      lds_double* M = mir + g * MS;
      const lds_double* z = M + AD::m_z;
      const lds_double* Cf = M + AD::m_Cf;
      const double h = z[TF] - z[T];
      double y[N], li[n > 0 ? n : 1];
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
      for (int k = 0; k < p; k++) y[q + k] = z[P0 + k];
#pragma unroll
      for (int k = 0; k < n; k++) li[k] = M[AD::m_lam + i * n + k];
      RegIn<D> in{y, li};
      if constexpr (ADJ) {
        OdeOutFG<true> out{M + AD::m_If + i * n, M + AD::m_Ig + i * N};
        Ode::fjgh(in, out);
      } else {
        OdeOutFG<false> out{M + AD::m_If + i * n, nullptr};
        Ode::f(in, out);
      }
    }
    wave_lds_sync();
  }
  // ---- cardinal nodes: g_j = (d f_j)^T w_j
  if (ADJ && lane < gcount * CS) {
    const int g = lane / CS, j = lane - g * CS;
    lds_double* M = mir + g * MS;
    const lds_double* z = M + AD::m_z;
    const double h = z[TF] - z[T];
    double w[n > 0 ? n : 1];
#pragma unroll
    for (int k = 0; k < n; k++) {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < K; i++) {
        acc += M[AD::m_Ig + i * N + k] * ((tab.E[i] * tab.B[i][j]) * h * h);
        acc += M[AD::m_lam + i * n + k] * (tab.D[i][j] * h);
      }
      w[k] = acc;
    }
    CardIn<D, const lds_double*> in{z, w, j, nullptr};
    OdeOutG out{M + AD::m_Cg + j * N};
    Ode::fjgh(in, out);
  }
  // ---- the two inner products per interior that the t_0 / t_f entries need (one lane per interior, straight-line: inside
  //      the entry loop below they were a 60-trip chain of dependent LDS reads that every pass of the wave waited for)
  if (ADJ && lane < gcount * K) {
    const int g = lane / K, i = lane - g * K;
    lds_double* M = mir + g * MS;
    double lsd = 0.0, gb = 0.0;
#pragma unroll
    for (int k = 0; k < n; k++) {
      double sd = tab.E[i] * M[AD::m_If + i * n + k], bf = 0.0;
#pragma unroll
      for (int jj = 0; jj < CS; jj++) {
        const double cf = M[AD::m_Cf + jj * n + k];
        sd += tab.D[i][jj] * cf;
        bf += tab.B[i][jj] * cf;
      }
      lsd += M[AD::m_lam + i * n + k] * sd;
      gb += M[AD::m_Ig + i * N + k] * bf;
    }
    M[AD::m_S + 2 * i] = lsd;
    M[AD::m_S + 2 * i + 1] = gb;
  }
  wave_lds_sync();
  // ---- defect values (LGLDefects.h:96-103)
  if (a.FX) {
    for (int e = lane; e < gcount * OR; e += 64) {
      const int g = e / OR, jr = e - g * OR, i = jr / n, k = jr - i * n;
      const lds_double* M = mir + g * MS;
      const lds_double* z = M + AD::m_z;
      const double h = z[TF] - z[T];
      double fxv = 0.0;
#pragma unroll
      for (int j = 0; j < CS; j++) fxv += (tab.C[i][j] * z[j * q + k] + (tab.D[i][j] * h) * M[AD::m_Cf + j * n + k]);
      fxv += (h * tab.E[i]) * M[AD::m_If + i * n + k];
      a.FX[size_t(seg0 + g) * OR + jr] = fxv;
    }
  }
  // ---- adjoint gradient
  if (ADJ && a.AGX) {
    for (int e = lane; e < gcount * IR; e += 64) {
      const int g = e / IR, c = e - g * IR;
      const lds_double* M = mir + g * MS;
      const lds_double* z = M + AD::m_z;
      const lds_double* lam = M + AD::m_lam;
      const lds_double* Ig = M + AD::m_Ig;
      const lds_double* Cg = M + AD::m_Cg;
      const double h = z[TF] - z[T];
      double v;
      if (c < P0) {
        const int j = c / q, cc = c - j * q;
        v = Cg[j * N + cc];
        if (cc < n) {
          for (int i = 0; i < K; i++) v += tab.C[i][j] * lam[i * n + cc] + (h * tab.E[i] * tab.A[i][j]) * Ig[i * N + cc];
        } else if (cc == T) {
          if (j == 0 || j == CS - 1) {
            double lsd = 0.0, tt = 0.0;
#pragma unroll
            for (int i = 0; i < K; i++) {
              const double he = h * tab.E[i], gb = M[AD::m_S + 2 * i + 1], gt = Ig[i * N + T];
              lsd += M[AD::m_S + 2 * i];
              tt += (j == 0) ? he * (-gb + (1.0 - tab.s[i]) * gt) : he * (gb + tab.s[i] * gt);
            }
            v += (j == 0) ? (tt - lsd) : (tt + lsd);
          }
        } else {
          const int k = cc - n - 1;
          for (int i = 0; i < K; i++) v += (h * tab.E[i] * tab.U[i][j]) * Ig[i * N + n + 1 + k];
        }
      } else {
        const int kk = c - P0;
        v = 0.0;
        for (int i = 0; i < K; i++) v += (h * tab.E[i]) * Ig[i * N + q + kk];
        for (int j = 0; j < CS; j++) v += Cg[j * N + q + kk];
      }
      a.AGX[size_t(seg0 + g) * IR + c] = v;
    }
  }
}

}  // namespace asset_hip
