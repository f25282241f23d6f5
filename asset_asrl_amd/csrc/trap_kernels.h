#pragma once
#include "defect_dims.h"

// =============================================================================================== Trapezoidal
// d = -[(x1 - x0) - (h/2)(f0 + f1)], its Jacobian, adjoint gradient and adjoint Hessian
// (/root/reference/src/OptimalControl/TrapezoidalDefects.h:146-184, 186-260, 263-435).  No interior point and
// no congruence product: every block entry is a closed form of the two cardinal ODE evaluations, so the kernel
// is two phases -- lane <-> (segment, node) for the ODE, then lanes stride over the block slots of each segment
// and store straight to HBM (coalesced, reference slot order).
namespace asset_hip {

template <class Ode, bool BLOCKED_>
struct TrapDims {
  static constexpr int n = Ode::XV;
  static constexpr int m = BLOCKED_ ? 0 : Ode::UV;
  static constexpr int p = BLOCKED_ ? Ode::UV + Ode::PV : Ode::PV;
  static constexpr int q = n + 1 + m;
  static constexpr int N = q + p;
  static constexpr int T = n, TF = q + n, P0 = 2 * q;
  static constexpr int IR = 2 * q + p, OR = n;
  static constexpr int NKKT = IR * (IR + 1) / 2 + OR * IR;
  static constexpr int NH = N * (N + 1) / 2;
  static constexpr int o_z = 0, o_lam = o_z + IR, o_F = o_lam + OR, o_J = o_F + 2 * n, o_G = o_J + 2 * n * N,
                       o_H = o_G + 2 * N, SLOT = o_H + 2 * NH;
  static_assert(IR + OR < 256, "slot map packs the row index in 8 bits");
  template <int G>
  static constexpr size_t lds_bytes() { return size_t(G) * SLOT * 8 + size_t((NKKT + 3) / 4 * 4) * 2; }
};

template <class D>
struct TrapIn {
  const double* z;
  const double* l;
  int j;
  __device__ double y(int i) const { return i < D::q ? z[j * D::q + i] : z[D::P0 + (i - D::q)]; }
  __device__ double lam(int k) const { return l[k]; }
};
template <class D>
struct TrapOut {
  double* f_;
  double* J_;
  double* g_;
  double* H_;
  __device__ void f(int k, double v) { f_[k] = v; }
  __device__ void J(int k, int i, double v) { J_[k * D::N + i] = v; }
  __device__ void g(int i, double v) { g_[i] = v; }
  __device__ void H(int i, int j, double v) { H_[i * (i + 1) / 2 + j] = v; }
};

// out of line for the same reason as the LGL ODE phases: a long generated body with its own register allocation
template <class Ode, class D, int LEVEL>
__device__ __attribute__((noinline)) void trap_node_eval(double* S, int j) {
  TrapIn<D> in{S + D::o_z, S + D::o_lam, j};
  TrapOut<D> out{S + D::o_F + j * D::n, S + D::o_J + j * D::n * D::N, S + D::o_G + j * D::N, S + D::o_H + j * D::NH};
  if constexpr (LEVEL == 0) Ode::f(in, out);
  else if constexpr (LEVEL == 1) Ode::fj(in, out);
  else Ode::fjgh(in, out);
}

template <class Ode, bool BLOCKED, int G, int LEVEL>
__global__ __launch_bounds__(64) void trap_defect_kernel(EvalArgs a) {
  using D = TrapDims<Ode, BLOCKED>;
  constexpr int n = D::n, q = D::q, N = D::N, T = D::T, TF = D::TF, P0 = D::P0, IR = D::IR, OR = D::OR;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  unsigned short* kmap = reinterpret_cast<unsigned short*>(lds + G * D::SLOT);
  const int lane = threadIdx.x;
  if constexpr (LEVEL >= 1) {
    for (int k = lane; k < D::NKKT; k += 64) {
      int lo = 0, hi = IR - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (mid * (IR + OR) - mid * (mid - 1) / 2 <= k) lo = mid; else hi = mid - 1;
      }
      const int r = k - (lo * (IR + OR) - lo * (lo - 1) / 2);
      kmap[k] = static_cast<unsigned short>((lo << 8) | (r < IR - lo ? lo + r : IR + (r - (IR - lo))));
    }
  }
  __syncthreads();
  const int ngroups = (a.nseg + G - 1) / G;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int seg0 = grp * G, gcount = min(G, a.nseg - seg0);
    for (int e = lane; e < gcount * IR; e += 64) {
      const int g = e / IR, r = e - g * IR;
      lds[g * D::SLOT + D::o_z + r] = a.X[a.vindex[size_t(seg0 + g) * IR + r]];
    }
    if constexpr (LEVEL >= 1) {
      for (int e = lane; e < gcount * OR; e += 64) {
        const int g = e / OR, r = e - g * OR;
        lds[g * D::SLOT + D::o_lam + r] = a.L ? a.L[a.cindex[size_t(seg0 + g) * OR + r]] : 0.0;
      }
    }
    __syncthreads();
    for (int e = lane; e < gcount * 2; e += 64) trap_node_eval<Ode, D, LEVEL>(lds + (e >> 1) * D::SLOT, e & 1);
    __syncthreads();
    for (int g = 0; g < gcount; g++) {
      const double* S = lds + g * D::SLOT;
      const double* z = S + D::o_z;
      const double* lam = S + D::o_lam;
      const double* F0 = S + D::o_F;
      const double* F1 = F0 + n;
      const double* J0 = S + D::o_J;
      const double* J1 = J0 + n * N;
      const double* G0 = S + D::o_G;
      const double* G1 = G0 + N;
      const double* H0 = S + D::o_H;
      const double* H1 = H0 + D::NH;
      const double h = z[TF] - z[T];
      const double mh2 = -h / 2.0;
      const size_t seg = size_t(seg0 + g);
      if (a.FX)
        for (int k = lane; k < OR; k += 64)
          a.FX[seg * OR + k] = -((z[q + k] - z[k]) - (h / 2.0) * (F0[k] + F1[k]));
      if constexpr (LEVEL == 0) continue;
      // final (already negated) Jacobian entry
      auto jac = [&](int k, int c) -> double {
        double v;
        if (c < q) {
          v = mh2 * J0[k * N + c];
          if (c == k) v += -1.0;
          if (c == T) v -= -0.5 * (F0[k] + F1[k]);
        } else if (c < P0) {
          const int cc = c - q;
          v = mh2 * J1[k * N + cc];
          if (cc == k) v += 1.0;
          if (cc == T) v += -0.5 * (F0[k] + F1[k]);
        } else {
          v = mh2 * (J0[k * N + q + (c - P0)] + J1[k * N + q + (c - P0)]);
        }
        return -v;
      };
      auto htpar = [&](int c) -> double {
        if (c < q) return -G0[c] * 0.5;
        if (c < P0) return -G1[c - q] * 0.5;
        return -G0[q + (c - P0)] * 0.5 + -G1[q + (c - P0)] * 0.5;
      };
      auto hess = [&](int r, int c) -> double {  // r >= c, final sign
        double v = 0.0;
        if (r < q) v = mh2 * hsym(H0, r, c);
        else if (r < P0) { if (c >= q) v = mh2 * hsym(H1, r - q, c - q); }
        else if (c >= P0) v = mh2 * (hsym(H0, q + r - P0, q + c - P0) + hsym(H1, q + r - P0, q + c - P0));
        else if (c < q) v = mh2 * hsym(H0, q + r - P0, c);
        else v = mh2 * hsym(H1, q + r - P0, c - q);
        if (c == T) v -= htpar(r);
        if (c == TF) v += htpar(r);
        if (r == T) v -= htpar(c);
        if (r == TF) v += htpar(c);
        return -v;
      };
      if (a.AGX)
        for (int c = lane; c < IR; c += 64) {
          double acc = 0.0;
          for (int k = 0; k < OR; k++) acc += lam[k] * jac(k, c);
          a.AGX[seg * IR + c] = acc;
        }
      if (a.KKT) {
        double* dst = a.KKT + seg * size_t(D::NKKT);
        for (int k = lane; k < D::NKKT; k += 64) {
          const int code = kmap[k], c = code >> 8, r = code & 255;
          double v;
          if (r < IR) v = (LEVEL >= 2) ? hess(r, c) : 0.0;
          else v = jac(r - IR, c);
          dst[k] = v;
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace asset_hip
