// Heavy right-hand sides in ONE launch (round 5): eight-wave workgroups, a group of G segments resident in LDS.
//
// STATUS: built, parity-exact (7e-16 / 9e-16 against the oracle at every size tried) and SLOWER than the shipped unit kernels + dense part --
// 31.6 against 20.1 us (Betts-LGL5 x 1 000), 93.5 against 90.0 us (Betts-LGL7 x 5 000); the dense part alone in this form
// (lgl_ures_dense_kernel) 28.2 against 20.2 us.  Compiled only with -DASSET_URES=1 / -DASSET_URES_DENSE=1 (defect_rowdpp.h: UResDims); the
// launcher takes what the kernel table holds.  Timeline and reasons: profiles/r5_ures_timeline.txt, DESIGN.md section 4.5.
//
// The unit kernels of defect_units.h spread an ODE whose derivative bodies do not fit a lane's registers over one single-wave
// workgroup per OUTPUT UNIT and group of segments; the results travel through the workspace in HBM to a second launch, the dense part
// (defect_resident.h, GIVEN form): for 1 000 Betts low-thrust LGL5 segments 9.7 us + 10.9 us, two launch latencies and 2.8 x the
// algorithmic bytes in memory traffic.  Here the units of a group are the WAVES of one workgroup:
//
//   P0  all waves     z, lam of the group's G segments and the weight tables -> LDS slots
//   P1  waves < U     cardinal values f_j (every unit wave for itself: the value body is light, the copies are bit-identical, and the
//                     interior phase then needs no workgroup barrier)
//   P2  wave u < U    interior points (lane = point x segment): unit u's share of [f^, J^, g^, H^] -> the slots  (LGLDefects.h:325-367)
//       -- barrier --
//   P3  wave u < U    cardinal nodes with the adjoint weights w_j (LGLDefects.h:369-374): unit u's share of [J, g, H]
//       -- barrier --
//   P4  all waves     the dense part by output rows (defect_rowdpp.h), its passes dealt round robin; blocks stored as they complete
//
// Nothing but the inputs is read from memory and nothing but the blocks is written.  Level 2, blocks (no on-device assembly); the
// other kinds keep the two-launch path.
#pragma once
#include "defect_resident.h"
#include "defect_units.h"

namespace asset_hip {

// P2 of one unit: interior point i of the segment whose slot is S -- x^, tau, u^ ; the unit's share of [f^, J^, g^, H^]
// (out of line, one function per unit: each gets its own register allocation -- inlined into the kernel, what the group loop keeps
//  alive is spilled and reloaded all through the bodies, and every reload is a round trip to memory)
template <class Ode, class D, int UN>
__device__ __attribute__((noinline, not_tail_called)) void ures_interior(lds_double* S, int i, const LglTab* tabp) {
  using U = UResDims<Ode, D>;
  constexpr int CS = D::CS, K = D::K, n = D::n, m = D::m, p = D::p, q = D::q, N = D::N, T = D::T;
  const LglTab& tab = *tabp;
  const lds_double* z = S + D::w_z;
  const lds_double* Cf = S + D::w_Cf;
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
  for (int k = 0; k < n; k++) li[k] = S[D::w_lam + i * n + k];
  if constexpr (UN == 0) {   // what the time rows of the dense part need (defect_rowdpp.h): FB_i[k] = sum_j B_ij f_j[k], and the step
#pragma unroll
    for (int k = 0; k < n; k++) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < CS; j++) acc = fma(tab.B[i][j], Cf[j * n + k], acc);
      S[U::s_FB + i * n + k] = acc;
    }
    if (i == 0) S[U::s_FB + K * n] = h;
  }
  RegIn<D> in{y, li};
  OdeOutRes<D> out{S + D::w_If + i * n, S + D::w_IJ + i * D::NZJ, S + D::w_Ig + i * N, S + D::w_IH + i * D::NZH, nullptr};
  Ode::template fjgh_unit<UN>(in, out);
}

// P3 of one unit: cardinal node j with the adjoint weights w_j (LGLDefects.h:369-374): the unit's share of [J_j, g_j, H_j]
template <class Ode, class D, int UN>
__device__ __attribute__((noinline, not_tail_called)) void ures_cardinal(lds_double* S, int j, const LglTab* tabp) {
  constexpr int K = D::K, n = D::n, N = D::N, T = D::T;
  const LglTab& tab = *tabp;
  const lds_double* z = S + D::w_z;
  const double h = z[D::TF] - z[T];
  double w[n > 0 ? n : 1];
#pragma unroll
  for (int k = 0; k < n; k++) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < K; i++) {
      acc += S[D::w_Ig + i * N + k] * ((tab.E[i] * tab.B[i][j]) * h * h);
      acc += S[D::w_lam + i * n + k] * (tab.D[i][j] * h);
    }
    w[k] = acc;
  }
  CardInRes<D> in{z, w, nullptr, j};
  OdeOutRes<D> out{nullptr, S + D::w_CJ + j * D::NZJ, S + D::w_Cg + j * N, S + D::w_CH + j * D::NZH, nullptr};
  Ode::template fjgh_unit<UN>(in, out);
}

template <class Ode, class D, int UN = 0>
__device__ __forceinline__ void ures_interior_unit(int unit, lds_double* S, int i, const LglTab* tabp) {
  if constexpr (UN < Ode::NUNITS) {
    if (unit == UN) ures_interior<Ode, D, UN>(S, i, tabp);
    else ures_interior_unit<Ode, D, UN + 1>(unit, S, i, tabp);
  }
}
template <class Ode, class D, int UN = 0>
__device__ __forceinline__ void ures_cardinal_unit(int unit, lds_double* S, int j, const LglTab* tabp) {
  if constexpr (UN < Ode::NUNITS) {
    if (unit == UN) ures_cardinal<Ode, D, UN>(S, j, tabp);
    else ures_cardinal_unit<Ode, D, UN + 1>(unit, S, j, tabp);
  }
}

template <class Ode, int SCH, bool BLOCKED, bool GIVEN>
__device__ __forceinline__ void lgl_ures_body(const EvalArgs& a, const int G) {
  using D = Dims<Ode, SCH, BLOCKED>;
  using U = UResDims<Ode, D>;
  using X = RdDims<Ode, D>;
  constexpr int CS = D::CS, K = D::K, n = D::n, m = D::m, p = D::p, q = D::q, N = D::N, T = D::T, IR = D::IR, OR = D::OR;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  lds_double* const tabL = (lds_double*)lds;
  lds_double* const slots = (lds_double*)(lds + D::TABSZ);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LglTab& tab = *reinterpret_cast<const LglTab*>(lds);
  const unsigned int* const rectab = static_cast<const unsigned int*>(a.lane_consts_res) +
                                     size_t(blockIdx.x % ASSET_LANE_REPLICAS) * size_t(res_table_words<Ode, D>()) + res_table_words_tile<Ode, D>();
  unsigned int recw[X::NQH * 4];
  bool have_rec = false;
  const int ngrp = (a.nseg + G - 1) / G;
#if defined(ASSET_TIMING)   // (tuning builds: clock64 stamps of every wave of workgroup 7, left in the first KKT block of its last group)
  long long tstamp[12];
  int nts = 0;
#define UTS() do { if (nts < 12) tstamp[nts++] = clock64(); } while (0)
#else
#define UTS() do {} while (0)
#endif
  UTS();
  for (int t = tid; t < D::TABSZ; t += 64 * U::NWV) tabL[t] = reinterpret_cast<const double*>(&d_lgl_tab[D::TAB])[t];

  for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
    const int seg0 = grp * G;
    const int gcount = min(G, a.nseg - seg0);
    if constexpr (GIVEN) {
      // ---- the group's ODE results from the workspace (written by the unit kernels, defect_units.h): G slots of WSLOT doubles, contiguous
      // there, SLOT apart here -- every thread 8 bytes of every 4 KB
      const double* const W = a.work + size_t(seg0) * D::WSLOT;
      constexpr int NC = (U::GMAX * D::WSLOT + 64 * U::NWV - 1) / (64 * U::NWV);
      double cv[NC];
#pragma unroll
      for (int t = 0; t < NC; t++) cv[t] = (tid + 64 * U::NWV * t < gcount * D::WSLOT) ? W[tid + 64 * U::NWV * t] : 0.0;
#pragma unroll
      for (int t = 0; t < NC; t++) {
        const int e = tid + 64 * U::NWV * t, g = e / D::WSLOT;
        if (e < gcount * D::WSLOT) slots[g * U::SLOT + (e - g * D::WSLOT)] = cv[t];
      }
      if (tid < gcount) slots[tid * U::SLOT + U::s_Z0] = 0.0;
      UTS();
      __syncthreads();
      UTS();
      // FB_i[k] = sum_j B_ij f_j[k] and the step: what the time rows of the passes read (defect_rowdpp.h)
      for (int e = tid; e < gcount * K * n; e += 64 * U::NWV) {
        const int g = e / (K * n), r = e - g * (K * n), i = r / n, k = r - i * n;
        lds_double* S = slots + g * U::SLOT;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < CS; j++) acc = fma(tab.B[i][j], S[D::w_Cf + j * n + k], acc);
        S[U::s_FB + r] = acc;
        if (r == 0) S[U::s_FB + K * n] = S[D::w_z + D::TF] - S[D::w_z + T];
      }
      UTS();
      __syncthreads();
      UTS();
    } else {
    // ---- P0: the group's inputs (every index first, then every value)
    {
      const int* vseg = a.vindex + size_t(seg0) * IR;
      const int* cseg = a.cindex + size_t(seg0) * OR;
      constexpr int NZ = (U::GMAX * IR + 64 * U::NWV - 1) / (64 * U::NWV), NL = (U::GMAX * OR + 64 * U::NWV - 1) / (64 * U::NWV);
      int zi[NZ], li[NL];
#pragma unroll
      for (int t = 0; t < NZ; t++) {
        const int e = tid + 64 * U::NWV * t, g = e / IR;
        zi[t] = (e < gcount * IR) ? (a.affine ? a.aff_v0 + (seg0 + g) * a.aff_vs + (e - g * IR) : vseg[e]) : 0;
      }
#pragma unroll
      for (int t = 0; t < NL; t++) {
        const int e = tid + 64 * U::NWV * t, g = e / OR;
        li[t] = (e < gcount * OR) ? (a.affine ? a.aff_c0 + (seg0 + g) * a.aff_cs + (e - g * OR) : cseg[e]) : 0;
      }
      double zv[NZ], lv[NL];
#pragma unroll
      for (int t = 0; t < NZ; t++) zv[t] = (tid + 64 * U::NWV * t < gcount * IR) ? a.X[zi[t]] : 0.0;
#pragma unroll
      for (int t = 0; t < NL; t++) lv[t] = (tid + 64 * U::NWV * t < gcount * OR && a.L) ? a.L[li[t]] : 0.0;
#pragma unroll
      for (int t = 0; t < NZ; t++) {
        const int e = tid + 64 * U::NWV * t, g = e / IR;
        if (e < gcount * IR) slots[g * U::SLOT + D::w_z + (e - g * IR)] = zv[t];
      }
#pragma unroll
      for (int t = 0; t < NL; t++) {
        const int e = tid + 64 * U::NWV * t, g = e / OR;
        if (e < gcount * OR) slots[g * U::SLOT + D::w_lam + (e - g * OR)] = lv[t];
      }
      if (tid < gcount) slots[tid * U::SLOT + U::s_Z0] = 0.0;
    }
    UTS();
    __syncthreads();
    UTS();

    if (wv < Ode::NUNITS) {
      // ---- P1: cardinal values f_j (lane = node x segment)
      if (lane < gcount * CS) {
        const int g = lane / CS, j = lane - g * CS;
        lds_double* S = slots + g * U::SLOT;
        CardInRes<D> in{S + D::w_z, nullptr, nullptr, j};
        OdeOutRes<D> out{S + D::w_Cf + j * n, nullptr, nullptr, nullptr, nullptr};
        Ode::f(in, out);
      }
      wave_lds_sync();
      UTS();
      // ---- P2: interior points, this wave's unit
      if (lane < gcount * K) {
        const int g = lane / K, i = lane - g * K;
        ures_interior_unit<Ode, D>(wv, slots + g * U::SLOT, i, &tab);
      }
    }
    UTS();
    __syncthreads();
    UTS();
    // ---- P3: cardinal nodes, second derivatives with the adjoint weights
    if (wv < Ode::NUNITS && lane < gcount * CS) {
      const int g = lane / CS, j = lane - g * CS;
      ures_cardinal_unit<Ode, D>(wv, slots + g * U::SLOT, j, &tab);
    }
    UTS();
    __syncthreads();
    UTS();
    }
    // ---- P4: the dense part
    rowdpp_dense<Ode, D, U::s_Z0, U::s_FB, 2>(
        a, (const lds_double*)tabL, rectab, gcount, seg0, seg0 + gcount, wv, U::NWV, lane,
        [&](int g) { return (const lds_double*)(slots + g * U::SLOT); }, [&](int g) { return seg0 + g; }, recw, have_rec);
    have_rec = true;
    UTS();
    __syncthreads();   // (the next group's inputs go where this one's passes read)
#if defined(ASSET_TIMING)
    if (blockIdx.x == 7 && lane == 0 && a.KKT && grp + int(gridDim.x) >= ngrp) {
      __builtin_amdgcn_s_waitcnt(0);
      for (int t = 0; t < nts; t++) a.KKT[size_t(seg0) * D::NKKT + wv * 16 + t] = double(tstamp[t] - tstamp[0]);
      a.KKT[size_t(seg0) * D::NKKT + wv * 16 + 15] = double(nts);
    }
#endif
  }
#undef UTS
}

// The kernel proper (an empty kernel for the shapes without it, which a run-time compiled module still names)
template <class Ode, int SCH, bool BLOCKED>
__global__ __launch_bounds__(512, 1) void lgl_ures_kernel(EvalArgs a, int G) {
  if constexpr (UResDims<Ode, Dims<Ode, SCH, BLOCKED>>::ONE_LAUNCH) lgl_ures_body<Ode, SCH, BLOCKED, false>(a, G);
}

// The dense part alone, behind the unit kernels (level 2, blocks): a group's slots from the workspace, then the passes of
// defect_rowdpp.h dealt over the workgroup's eight waves
template <class Ode, int SCH, bool BLOCKED>
__global__ __launch_bounds__(512, 1) void lgl_ures_dense_kernel(EvalArgs a, int G) {
  if constexpr (UResDims<Ode, Dims<Ode, SCH, BLOCKED>>::OK) lgl_ures_body<Ode, SCH, BLOCKED, true>(a, G);
}

}  // namespace asset_hip
