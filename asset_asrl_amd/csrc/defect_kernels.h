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
// (/root/reference/src/VectorFunctions/DenseFunctionBase.h:1097-1129, 1276-1391).
//
// Execution shape (one 64-lane wavefront per workgroup, grid-stride over groups of G segments):
//   phase P0  gather z = X[Vindex], lam = L[Cindex] for G segments into LDS (index reads coalesced)
//   phase P1  lane <-> (segment, cardinal node):   f_j                      (ODE value)
//   phase P2  lane <-> (segment, interior point):  x^,tau,u^ ; f^, J^, g^ = J^^T lam_i, H^ = lam_i^T d2f
//   phase P3  lane <-> (segment, cardinal node):   w_j ; J_j, g_j = J_j^T w_j, H_j = w_j^T d2f
//   phase P4  all 64 lanes on ONE segment at a time: stack DI = d(x^,tau,u^,P)/dz for the K interiors
//             into a (K*N) x IR LDS tile, M = (h E_i H^_i) DI, then the lower triangle of
//             H = DI^T M and J^T = DI^T (h E J^)^T as 16x16x4 f64 MFMA tiles out of LDS; the result tile
//             is staged in LDS, the sparse cardinal / time-column terms are added there, and the finished
//             block is streamed to HBM with fully coalesced 8-byte stores in the reference's slot order.
// The ODE is an inlined generated functor (asset_asrl_amd/vf/codegen.py), so P1-P3 are straight-line
// register code; the only HBM traffic is the gather and the block stores.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lgl_tables.h"

namespace asset_hip {

static __constant__ LglTab d_lgl_tab[3] = ASSET_LGL_TABLE_INIT;

struct EvalArgs {
  int nseg;
  const double* X;     // NLP primal vector (device)
  const double* L;     // equality multipliers (device); unused for value-only
  const int* vindex;   // [IR x nseg] column-major (device)
  const int* cindex;   // [OR x nseg] column-major (device)
  double* FX;          // [nseg x OR] blocks or null
  double* AGX;         // [nseg x IR] blocks or null
  double* KKT;         // [nseg x NKKT] blocks or null
};

// ---------------------------------------------------------------------------------------------- sizes
template <class Ode, int CS_, bool BLOCKED_>
struct Dims {
  static constexpr int CS = CS_, K = CS_ - 1;
  static constexpr int n = Ode::XV;
  static constexpr int m = BLOCKED_ ? 0 : Ode::UV;                    // Blocked_ODE_Wrapper.h:7-27
  static constexpr int p = BLOCKED_ ? Ode::UV + Ode::PV : Ode::PV;
  static constexpr int q = n + 1 + m;
  static constexpr int N = q + p;
  static constexpr int T = n;
  static constexpr int IR = CS * q + p;                               // TranscriptionSizing.h:7-14
  static constexpr int OR = K * n;
  static constexpr int TF = q * (CS - 1) + T;
  static constexpr int P0 = CS * q;
  static constexpr int NKKT = IR * (IR + 1) / 2 + OR * IR;            // DenseFunctionBase.h:1070-1088
  static constexpr int NH = N * (N + 1) / 2;                          // packed lower ODE Hessian
  static constexpr int IRP = (IR + 15) / 16 * 16;
  static constexpr int ORP = (OR + 15) / 16 * 16;
  static constexpr int KT = (K * N + 3) / 4 * 4;                      // stacked contraction depth, MFMA k=4
  // LDS leading dimensions (doubles).  Operand tiles are read as 16 consecutive doubles per 16-lane
  // group with consecutive groups one row apart: stride = 16 (mod 32) doubles keeps ds_read_b64 conflict-free.
  static constexpr int LDD = (IRP % 32 == 0) ? IRP + 16 : IRP;        // DI / M tiles
  static constexpr int LDJ = (ORP % 32 == 0) ? ORP + 16 : ORP;        // (hE J^)^T tile
  static constexpr int LDO = IRP;                                     // staged output: H part  [IRP][LDO]
  static constexpr int LDOJ = ORP;                                    //                J^T part [IRP][LDOJ]
  static constexpr int TI = IRP / 16, TJ = ORP / 16;
  static constexpr int NACC = TI * (TI + 1) / 2 + TI * TJ;            // accumulator tiles per lane

  // ---- LDS map (in doubles) : per in-flight segment slot
  static constexpr int o_z = 0;
  static constexpr int o_lam = o_z + IR;
  static constexpr int o_Cf = o_lam + OR;
  static constexpr int o_CJ = o_Cf + CS * n;
  static constexpr int o_Cg = o_CJ + CS * n * N;
  static constexpr int o_CH = o_Cg + CS * N;
  static constexpr int o_If = o_CH + CS * NH;
  static constexpr int o_IJ = o_If + K * n;
  static constexpr int o_Ig = o_IJ + K * n * N;
  static constexpr int o_IH = o_Ig + K * N;
  static constexpr int SLOT = o_IH + K * NH;
  // ---- dense scratch (one segment at a time)
  static constexpr int OUTSZ = IRP * (LDO + LDOJ);
  static constexpr int OPSZ = 2 * KT * LDD;
  static constexpr int s_DI = 0;                 // DI tile   [KT][LDD]
  static constexpr int s_M = KT * LDD;           // M tile    [KT][LDD]
  static constexpr int s_OUT = 0;                // output staging aliases DI/M (MFMA path, after a barrier)
  static constexpr int REGION = OUTSZ > OPSZ ? OUTSZ : OPSZ;
  static constexpr int s_LJ = REGION;            // (hE J^)^T [KT][LDJ]
  static constexpr int s_HT = s_LJ + KT * LDJ;   // HTpar [IRP]
  static constexpr int SCRATCH = s_HT + IRP;

  template <int G>
  static constexpr int lds_doubles() { return G * SLOT + SCRATCH; }
  template <int G>
  static constexpr size_t lds_bytes() { return size_t(lds_doubles<G>()) * 8 + size_t((NKKT + 3) / 4 * 4) * 2; }
};

using d4 = __attribute__((ext_vector_type(4))) double;

// ---------------------------------------------------------------------------------------------- ODE accessors
template <class D>
struct CardIn {  // y = [z_j (q), P (p)] read from the LDS copy of z; lam = adjoint weights in registers
  const double* z;
  const double* w;
  int j;
  __device__ double y(int i) const { return i < D::q ? z[j * D::q + i] : z[D::P0 + (i - D::q)]; }
  __device__ double lam(int k) const { return w[k]; }
};
template <class D>
struct RegIn {
  const double* yv;
  const double* lv;
  __device__ double y(int i) const { return yv[i]; }
  __device__ double lam(int k) const { return lv[k]; }
};
template <class D>
struct OdeOut {  // routes every derivative entry to its LDS slot (J row-major n x N, H packed lower)
  double* f_;
  double* J_;
  double* g_;
  double* H_;
  __device__ void f(int k, double v) { f_[k] = v; }
  __device__ void J(int k, int i, double v) { J_[k * D::N + i] = v; }
  __device__ void g(int i, double v) { g_[i] = v; }
  __device__ void H(int i, int j, double v) { H_[i * (i + 1) / 2 + j] = v; }
};

__device__ inline double hsym(const double* Hp, int a, int b) {
  return a >= b ? Hp[a * (a + 1) / 2 + b] : Hp[b * (b + 1) / 2 + a];
}

// ---------------------------------------------------------------------------------------------- kernel
// LEVEL 0: value only (constraints).  LEVEL 1: value + Jacobian (+ J^T lam).  LEVEL 2: + adjoint Hessian.
// MFMA: use v_mfma_f64_16x16x4_f64 for the congruence; false = plain FMA loops (cross-check / fallback sizes).
template <class Ode, int CS, bool BLOCKED, int G, int LEVEL, bool MFMA>
__global__ __launch_bounds__(64) void lgl_defect_kernel(EvalArgs a) {
  using D = Dims<Ode, CS, BLOCKED>;
  constexpr int K = D::K, n = D::n, m = D::m, p = D::p, q = D::q, N = D::N, T = D::T;
  constexpr int IR = D::IR, OR = D::OR;
  static_assert(N == Ode::NIN, "ODE input size mismatch");
  const LglTab& tab = d_lgl_tab[CS - 2];

  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* slots = lds;
  double* scr = lds + G * D::SLOT;
  unsigned short* kmap = reinterpret_cast<unsigned short*>(lds + D::template lds_doubles<G>());
  const int lane = threadIdx.x;

  // ---- slot order of the KKT block -> offset in the staged output tile (DenseFunctionBase.h:1112-1123)
  if constexpr (LEVEL >= 1) {
    for (int k = lane; k < D::NKKT; k += 64) {
      // column i owns (IR - i) Hessian slots then OR Jacobian slots; start(i) = i*(IR+OR) - i(i-1)/2
      int i = 0;
      int lo = 0, hi = IR - 1;
      while (lo < hi) {  // largest i with start(i) <= k
        const int mid = (lo + hi + 1) >> 1;
        const int st = mid * (IR + OR) - mid * (mid - 1) / 2;
        if (st <= k) lo = mid; else hi = mid - 1;
      }
      i = lo;
      const int r = k - (i * (IR + OR) - i * (i - 1) / 2);
      const int off = (r < IR - i) ? (i * D::LDO + (i + r)) : (D::IRP * D::LDO + i * D::LDOJ + (r - (IR - i)));
      kmap[k] = static_cast<unsigned short>(off);
    }
  }
  __syncthreads();

  const int ngroups = (a.nseg + G - 1) / G;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int seg0 = grp * G;
    const int gcount = min(G, a.nseg - seg0);

    // ------------------------------------------------------------------ P0: gather
    for (int e = lane; e < gcount * IR; e += 64) {
      const int g = e / IR, r = e - g * IR;
      slots[g * D::SLOT + D::o_z + r] = a.X[a.vindex[size_t(seg0 + g) * IR + r]];
    }
    if constexpr (LEVEL >= 1) {
      for (int e = lane; e < gcount * OR; e += 64) {
        const int g = e / OR, r = e - g * OR;
        slots[g * D::SLOT + D::o_lam + r] = a.L ? a.L[a.cindex[size_t(seg0 + g) * OR + r]] : 0.0;
      }
    }
    __syncthreads();

    // ------------------------------------------------------------------ P1: cardinal ODE values (and J for LEVEL 1)
    for (int e = lane; e < gcount * CS; e += 64) {
      const int g = e / CS, j = e - g * CS;
      double* S = slots + g * D::SLOT;
      CardIn<D> in{S + D::o_z, nullptr, j};
      OdeOut<D> out{S + D::o_Cf + j * n, S + D::o_CJ + j * n * N, nullptr, nullptr};
      if constexpr (LEVEL == 1) Ode::fj(in, out);
      else Ode::f(in, out);
    }
    __syncthreads();

    // ------------------------------------------------------------------ P2: interior points
    for (int e = lane; e < gcount * K; e += 64) {
      const int g = e / K, i = e - g * K;
      double* S = slots + g * D::SLOT;
      const double* z = S + D::o_z;
      const double h = z[D::TF] - z[T];
      double y[N];
      double li[n > 0 ? n : 1];
#pragma unroll
      for (int k = 0; k < n; k++) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < CS; j++) acc += (tab.A[i][j] * z[j * q + k] + (tab.B[i][j] * h) * S[D::o_Cf + j * n + k]);
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
      for (int k = 0; k < n; k++) li[k] = (LEVEL >= 2) ? S[D::o_lam + i * n + k] : 0.0;
      RegIn<D> in{y, li};
      OdeOut<D> out{S + D::o_If + i * n, S + D::o_IJ + i * n * N, S + D::o_Ig + i * N, S + D::o_IH + i * D::NH};
      if constexpr (LEVEL == 0) Ode::f(in, out);
      else if constexpr (LEVEL == 1) Ode::fj(in, out);
      else Ode::fjgh(in, out);
    }
    __syncthreads();

    // ------------------------------------------------------------------ P3: cardinal second derivatives
    if constexpr (LEVEL >= 2) {
      for (int e = lane; e < gcount * CS; e += 64) {
        const int g = e / CS, j = e - g * CS;
        double* S = slots + g * D::SLOT;
        const double* z = S + D::o_z;
        const double h = z[D::TF] - z[T];
        double w[n > 0 ? n : 1];
#pragma unroll
        for (int k = 0; k < n; k++) {  // C_AVS[j]  (LGLDefects.h:369-374)
          double acc = 0.0;
#pragma unroll
          for (int i = 0; i < K; i++) {
            acc += S[D::o_Ig + i * N + k] * ((tab.E[i] * tab.B[i][j]) * h * h);
            acc += S[D::o_lam + i * n + k] * (tab.D[i][j] * h);
          }
          w[k] = acc;
        }
        CardIn<D> in{z, w, j};
        OdeOut<D> out{S + D::o_Cf + j * n, S + D::o_CJ + j * n * N, S + D::o_Cg + j * N, S + D::o_CH + j * D::NH};
        Ode::fjgh(in, out);
      }
      __syncthreads();
    }

    // ------------------------------------------------------------------ P4: per-segment dense phase
    for (int g = 0; g < gcount; g++) {
      const double* S = slots + g * D::SLOT;
      const double* z = S + D::o_z;
      const double* lam = S + D::o_lam;
      const double h = z[D::TF] - z[T];
      const size_t seg = size_t(seg0 + g);

      // ---- value (every level): lanes over (i,k)
      if (a.FX) {
        for (int e = lane; e < OR; e += 64) {
          const int i = e / n, k = e - i * n;
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < CS; j++) acc += (tab.C[i][j] * z[j * q + k] + (tab.D[i][j] * h) * S[D::o_Cf + j * n + k]);
          acc += (h * tab.E[i]) * S[D::o_If + i * n + k];
          a.FX[seg * OR + e] = acc;
        }
      }
      if constexpr (LEVEL == 0) continue;

      double* DI = scr + D::s_DI;
      double* Mt = scr + D::s_M;
      double* LJ = scr + D::s_LJ;
      double* HT = scr + D::s_HT;
      double* OH = scr + D::s_OUT;
      double* OJ = OH + D::IRP * D::LDO;

      // ---- D1: DI tile (stacked over interiors) and (hE J^)^T tile
      for (int e = lane; e < D::KT * D::IRP; e += 64) {
        const int kk = e / D::IRP, c = e - kk * D::IRP;
        double v = 0.0;
        if (kk < K * N && c < IR) {
          const int i = kk / N, r = kk - i * N;
          if (r < n) {
            if (c < D::P0) {
              const int j = c / q, cc = c - j * q;
              v = (tab.B[i][j] * h) * S[D::o_CJ + (j * n + r) * N + cc];
              if (cc == r) v += tab.A[i][j];
              if (cc == T && (j == 0 || j == CS - 1)) {
                double sf = 0.0;
#pragma unroll
                for (int jj = 0; jj < CS; jj++) sf += tab.B[i][jj] * S[D::o_Cf + jj * n + r];
                v += (j == 0 && CS > 1) ? -sf : sf;
              }
            } else {
              const int cp = c - D::P0;
#pragma unroll
              for (int jj = 0; jj < CS; jj++) v += (tab.B[i][jj] * h) * S[D::o_CJ + (jj * n + r) * N + q + cp];
            }
          } else if (r == T) {
            v = (c == T) ? (1.0 - tab.s[i]) : ((c == D::TF) ? tab.s[i] : 0.0);
          } else if (r < q) {
            const int ku = r - (n + 1);
            if (c < D::P0) {
              const int j = c / q, cc = c - j * q;
              if (cc == n + 1 + ku) v = tab.U[i][j];
            }
          } else {
            if (c == D::P0 + (r - q)) v = 1.0;
          }
        }
        DI[kk * D::LDD + c] = v;
      }
      for (int e = lane; e < D::KT * D::ORP; e += 64) {
        const int kk = e / D::ORP, jr = e - kk * D::ORP;
        double v = 0.0;
        if (kk < K * N && jr < OR) {
          const int i = kk / N, r = kk - i * N;
          const int i2 = jr / n, k2 = jr - i2 * n;
          if (i2 == i) v = (h * tab.E[i]) * S[D::o_IJ + (i * n + k2) * N + r];
        }
        LJ[kk * D::LDJ + jr] = v;
      }
      __syncthreads();

      // ---- D2: M = (hE_i H^_i) DI_i ; time-partial vector HTpar
      if constexpr (LEVEL >= 2) {
        for (int e = lane; e < D::KT * D::IRP; e += 64) {
          const int kk = e / D::IRP, c = e - kk * D::IRP;
          double v = 0.0;
          if (kk < K * N && c < IR) {
            const int i = kk / N, r = kk - i * N;
            const double* Hp = S + D::o_IH + i * D::NH;
            const double he = h * tab.E[i];
#pragma unroll
            for (int b = 0; b < N; b++) v += (hsym(Hp, r, b) * he) * DI[(i * N + b) * D::LDD + c];
          }
          Mt[kk * D::LDD + c] = v;
        }
        for (int c = lane; c < D::IRP; c += 64) {
          double v = 0.0;
          if (c < IR) {
            const double ih = 1.0 / h;
            if (c < D::P0) {
              const int j = c / q, cc = c - j * q;
              v = S[D::o_Cg + j * N + cc] * ih;
            } else {
#pragma unroll
              for (int j = 0; j < CS; j++) v += S[D::o_Cg + j * N + q + (c - D::P0)] * ih;
            }
#pragma unroll
            for (int i = 0; i < K; i++) {
              double acc = 0.0;
#pragma unroll
              for (int r = 0; r < N; r++) acc += (S[D::o_Ig + i * N + r] * tab.E[i]) * DI[(i * N + r) * D::LDD + c];
              v += acc;
            }
          }
          HT[c] = v;
        }
        __syncthreads();
      }

      // ---- D3/D4: congruence products into the staged output tile
      if constexpr (MFMA) {
        d4 acc[D::NACC];
#pragma unroll
        for (int t = 0; t < D::NACC; t++) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int ct = 0; ct < D::TI; ct++) {
          const int tb = ct * (D::TI + D::TJ) - ct * (ct - 1) / 2;  // first accumulator tile of this tile-row
          double av[D::KT / 4];
#pragma unroll
          for (int kk = 0; kk < D::KT / 4; kk++) av[kk] = DI[(4 * kk + lk) * D::LDD + 16 * ct + lr];
          if constexpr (LEVEL >= 2) {
#pragma unroll
            for (int rt = ct; rt < D::TI; rt++) {
#pragma unroll
              for (int kk = 0; kk < D::KT / 4; kk++) {
                const double bv = Mt[(4 * kk + lk) * D::LDD + 16 * rt + lr];
                acc[tb + rt - ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv, acc[tb + rt - ct], 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int jt = 0; jt < D::TJ; jt++) {
#pragma unroll
            for (int kk = 0; kk < D::KT / 4; kk++) {
              const double bv = LJ[(4 * kk + lk) * D::LDJ + 16 * jt + lr];
              acc[tb + D::TI - ct + jt] =
                  __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv, acc[tb + D::TI - ct + jt], 0, 0, 0);
            }
          }
        }
        __syncthreads();  // every lane has finished reading DI / M: the staging tile may overwrite them
#pragma unroll
        for (int ct = 0; ct < D::TI; ct++) {
          const int tb = ct * (D::TI + D::TJ) - ct * (ct - 1) / 2;
#pragma unroll
          for (int rt = ct; rt < D::TI; rt++) {
#pragma unroll
            for (int v = 0; v < 4; v++) OH[(16 * ct + lk + 4 * v) * D::LDO + 16 * rt + lr] = acc[tb + rt - ct][v];
          }
#pragma unroll
          for (int jt = 0; jt < D::TJ; jt++) {
#pragma unroll
            for (int v = 0; v < 4; v++)
              OJ[(16 * ct + lk + 4 * v) * D::LDOJ + 16 * jt + lr] = acc[tb + D::TI - ct + jt][v];
          }
        }
      } else {
        // plain FMA reference path: each lane owns output elements and keeps them in registers until the barrier
        constexpr int NE_H = D::IRP * D::IRP, NE_J = D::IRP * D::ORP;
        constexpr int PER = (NE_H + NE_J + 63) / 64;
        double accv[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
          const int e = lane + 64 * u;
          double v = 0.0;
          if (e < NE_H) {
            const int c = e / D::IRP, r = e - c * D::IRP;
            if (LEVEL >= 2 && r >= c)
              for (int kk = 0; kk < K * N; kk++) v += DI[kk * D::LDD + c] * Mt[kk * D::LDD + r];
          } else if (e < NE_H + NE_J) {
            const int e2 = e - NE_H;
            const int c = e2 / D::ORP, jr = e2 - c * D::ORP;
            for (int kk = 0; kk < K * N; kk++) v += DI[kk * D::LDD + c] * LJ[kk * D::LDJ + jr];
          }
          accv[u] = v;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; u++) {
          const int e = lane + 64 * u;
          if (e < NE_H) {
            const int c = e / D::IRP, r = e - c * D::IRP;
            OH[c * D::LDO + r] = accv[u];
          } else if (e < NE_H + NE_J) {
            const int e2 = e - NE_H;
            const int c = e2 / D::ORP, jr = e2 - c * D::ORP;
            OJ[c * D::LDOJ + jr] = accv[u];
          }
        }
      }
      __syncthreads();

      // ---- D5: sparse terms.  Jacobian: cardinal blocks + time columns (LGLDefects.h:467-500)
      for (int e = lane; e < OR * IR; e += 64) {
        const int c = e / OR, jr = e - c * OR;
        const int i = jr / n, k = jr - i * n;
        double v = 0.0;
        if (c < D::P0) {
          const int j = c / q, cc = c - j * q;
          v = (tab.D[i][j] * h) * S[D::o_CJ + (j * n + k) * N + cc];
          if (cc == k) v += tab.C[i][j];
          if (cc == T && (j == 0 || j == CS - 1)) {
            double sf = 0.0;
#pragma unroll
            for (int jj = 0; jj < CS; jj++) sf += tab.D[i][jj] * S[D::o_Cf + jj * n + k];
            sf += tab.E[i] * S[D::o_If + i * n + k];
            v += (j == 0) ? -sf : sf;
          }
        } else {
#pragma unroll
          for (int jj = 0; jj < CS; jj++) v += (tab.D[i][jj] * h) * S[D::o_CJ + (jj * n + k) * N + q + (c - D::P0)];
        }
        OJ[c * D::LDOJ + jr] += v;
      }
      if constexpr (LEVEL >= 2) {
        // Hessian: cardinal diagonal / parameter blocks (LGLDefects.h:386-402), lower triangle only (r >= c)
        for (int e = lane; e < CS * q * q; e += 64) {
          const int j = e / (q * q), rem = e - j * q * q;
          const int aa = rem / q, bb = rem - aa * q;  // H(jq+aa, jq+bb), keep aa >= bb
          if (aa >= bb) OH[(j * q + bb) * D::LDO + (j * q + aa)] += S[D::o_CH + j * D::NH + aa * (aa + 1) / 2 + bb];
        }
        if constexpr (p > 0) {
          for (int e = lane; e < CS * q * p; e += 64) {
            const int j = e / (q * p), rem = e - j * q * p;
            const int aa = rem / p, bb = rem - aa * p;  // row P0+bb, col jq+aa
            OH[(j * q + aa) * D::LDO + (D::P0 + bb)] += hsym(S + D::o_CH + j * D::NH, q + bb, aa);
          }
          for (int e = lane; e < p * p; e += 64) {
            const int aa = e / p, bb = e - aa * p;  // row P0+aa, col P0+bb, aa >= bb
            if (aa >= bb) {
              double v = 0.0;
#pragma unroll
              for (int j = 0; j < CS; j++) v += hsym(S + D::o_CH + j * D::NH, q + aa, q + bb);
              OH[(D::P0 + bb) * D::LDO + (D::P0 + aa)] += v;
            }
          }
        }
      }
      __syncthreads();

      // ---- D6/D7: rank-2 time update (LGLDefects.h:508-511) and adjoint gradient (:512)
      if constexpr (LEVEL >= 2) {
        for (int r = lane; r < IR; r += 64) {  // columns T and TF, rows r >= column
          if (r >= T) OH[T * D::LDO + r] -= HT[r];
          if (r >= D::TF) OH[D::TF * D::LDO + r] += HT[r];
        }
        __syncthreads();
        for (int c = lane; c < IR; c += 64) {  // rows T and TF, columns c <= row
          if (c <= T) OH[c * D::LDO + T] -= HT[c];
          if (c <= D::TF) OH[c * D::LDO + D::TF] += HT[c];
        }
      }
      if (a.AGX) {
        for (int c = lane; c < IR; c += 64) {
          double acc = 0.0;
          for (int r = 0; r < OR; r++) acc += lam[r] * OJ[c * D::LDOJ + r];
          a.AGX[seg * IR + c] = acc;
        }
      }
      __syncthreads();

      // ---- D8: stream the finished block, reference slot order, coalesced
      if (a.KKT) {
        double* dst = a.KKT + seg * size_t(D::NKKT);
        for (int k = lane; k < D::NKKT; k += 64) {
          const int off = kmap[k];
          double v = scr[D::s_OUT + off];
          if (LEVEL < 2 && off < D::IRP * D::LDO) v = 0.0;
          dst[k] = v;
        }
      }
      __syncthreads();
    }
  }
}

}  // namespace asset_hip
