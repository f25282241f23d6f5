// ODE stage for heavy right-hand sides: one WORKGROUP (a single wave) per output unit and group of segments.
//
// The ODE stage of defect_kernels.h maps a lane to an evaluation point and runs the whole generated body
// f + J + g + H of that point in the lane.  For a heavy ODE (Betts' modified-equinoctial low-thrust dynamics with J2-J4
// gravity: 7 792 operations, ~780 values live at the peak) that body spills thousands of bytes per lane, and a small mesh
// (BASELINE.json configs[1]: 1 000 segments = 5 000 points, 80 waves on a 1 024-SIMD device) leaves nothing to hide the
// scratch round trips behind: the stage is one long latency chain (0.24 ms whatever the mesh size).
//
// Here the body is cut by OUTPUT: the generator (vf/codegen.py, plan_units) emits Ode::NUNITS bodies fjgh_unit<U>,
// unit "column k" = {J[:,k], H[i>=k,k]} (the second derivatives in input direction k) and one unit for f, g and the cheap
// columns; each recomputes the forward values it needs (6 x ~1 700 + 1 200 operations instead of 7 792, no spills).  The
// grid is (groups of segments) x (units); a workgroup evaluates its unit for every point of its group (lane <-> point), so
// the chain a point waits for is one unit long and the idle SIMDs of a small mesh do the rest in parallel.  Phases as in the reference's
// adjoint-Hessian method (LGLDefects.h:325-412): cardinal values -> interior points (all units) -> cardinal second
// derivatives with the adjoint weights w_j (all units), hand-offs through an LDS mirror of the group's small vectors.
// Results go to the same per-segment workspace slots the dense stage (defect_kernels.h, STAGE 2) reads.
#pragma once
#include "defect_dims.h"

namespace asset_hip {

template <class D>
struct UnitsDims {
  static constexpr int CS = D::CS, K = D::K, n = D::n, N = D::N;
  static constexpr int GP = 64 / CS;                    // segments per group at most (lane <-> cardinal point)
  // mirror slot of one segment: z | lam | f_j | g^_i
  static constexpr int m_z = 0, m_lam = D::IR, m_Cf = m_lam + D::OR, m_Ig = m_Cf + CS * n;
  static constexpr int MS = (m_Ig + K * N) | 1;
  static constexpr size_t lds_bytes(int gp) { return size_t(D::TABSZ + gp * MS) * 8; }
};

// outputs of one unit go straight to the segment's slot (the unit bodies emit every entry of their share once)
template <class D, bool MIRROR_G, bool WRITE_F>
struct OdeOutUnit {
  glb_double* f_;
  glb_double* J_;
  glb_double* g_;
  glb_double* H_;
  lds_double* gm_;      // mirror copy of g (interior points: the cardinal weights need it)
  __device__ void f(int k, double v) { if constexpr (WRITE_F) f_[k] = v; }
  __device__ void J(int k, int i, double v) {
    const int c = D::ode_t::JPOS[k * D::N + i];
    if (c >= 0) J_[c] = v;
  }
  __device__ void g(int i, double v) {
    g_[i] = v;
    if constexpr (MIRROR_G) gm_[i] = v;
  }
  __device__ void H(int i, int j, double v) {
    const int c = D::ode_t::HPOS[i * (i + 1) / 2 + j];
    if (c >= 0) H_[c] = v;
  }
};

// Jacobian kinds (evalSOE / evalAUG): f and the unit's columns of J; g^ = J^^T lam accumulated per owned column while the
// column is emitted (the bodies are inlined: the accumulators are registers)
template <class D>
struct OdeOutUnitJ {
  glb_double* f_;        // null: dropped
  glb_double* J_;
  const double* lam_;    // null: no g^
  double gacc_[D::N];
  __device__ void f(int k, double v) { if (f_) f_[k] = v; }
  __device__ void J(int k, int i, double v) {
    const int c = D::ode_t::JPOS[k * D::N + i];
    if (c >= 0) J_[c] = v;
    if (lam_) gacc_[i] = fma(lam_[k], v, gacc_[i]);
  }
};

template <class D>
struct OdeOutGx {   // g^_i[0:n] of an interior point -> the group's mirror (PHASE 4: the cardinal units form it themselves)
  lds_double* gm_;
  __device__ void g(int i, double v) { gm_[i] = v; }
};

template <class D, bool SLOT>
struct OdeOutValue {   // cardinal values: mirror (the interior points read them) and, from one unit, the slot (the dense stage does)
  lds_double* fm_;
  glb_double* f_;
  __device__ void f(int k, double v) {
    fm_[k] = v;
    if constexpr (SLOT) f_[k] = v;
  }
};

template <class Ode, class In, class Out, int U = 0>
__device__ inline void run_unit_j(int unit, const In& in, Out& out) {
  if constexpr (U < Ode::NUNITS) {
    if (unit == U) Ode::template fj_unit<U>(in, out);
    else run_unit_j<Ode, In, Out, U + 1>(unit, in, out);
  }
}

template <class Ode, class In, class Out, int U = 0>
__device__ inline void run_unit(int unit, const In& in, Out& out) {
  if constexpr (U < Ode::NUNITS) {
    if (unit == U) Ode::template fjgh_unit<U>(in, out);
    else run_unit<Ode, In, Out, U + 1>(unit, in, out);
  }
}

// PHASE 0: gather, cardinal values (every unit's workgroup computes them for itself: the value body is light), interior
//          points -- unit u = blockIdx.y.   PHASE 1 (second launch: it needs g^ of the interior pass): cardinal
//          second derivatives.   (One launch for both, with g^ handed over through agent-scope stores and a counter
//          per group of segments, was built and measured in round 3: 54.1 us against 43.5 us for 1 000 Betts-LGL5 segments --
//          the unit bodies read their inputs through scratch memory, and the polling and the L2-bypassing traffic of the
//          workgroups that are done stretch the interior pass of the others from 7.2 us to 9-23 us)
//          PHASE 4 (round 4): the whole stage in ONE launch without a hand-over -- the grid is (groups) x (2 NUNITS); the upper
//          half are the cardinal units, and each of their workgroups forms g^_i[0:n] of its group's interior points ITSELF
//          (Ode::gx, a vector-Jacobian product: 782 operations for Betts against ~1 700 of a column unit) instead of waiting
//          for the interior units: what they need of the interior pass is that vector and nothing else (LGLDefects.h:369-374).
//          The units recompute what they share, so the form pays while the mesh leaves SIMDs idle (registry.h).
//          Single-wave workgroups at one wave per SIMD: each unit body has the whole register file
//          (512 with the accumulation registers as spill space), where seven waves in one workgroup had 256 each and
//          spilled ~1 KB per lane to scratch (110 MB of scratch traffic per evaluation of 1 000 Betts segments).
template <class Ode, int SCH, bool BLOCKED, int PHASE>
__device__ __forceinline__ void lgl_ode_units_body(const EvalArgs& a, int gp) {
  using D = Dims<Ode, SCH, BLOCKED>;
  using UD = UnitsDims<D>;
  constexpr int CS = D::CS, K = D::K, n = D::n, m = D::m, p = D::p, q = D::q, N = D::N, T = D::T, IR = D::IR, OR = D::OR;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  lds_double* const tabL = (lds_double*)lds;
  lds_double* const mirror = (lds_double*)(lds + D::TABSZ);
  const int lane = threadIdx.x;
#if defined(ASSET_EXP_ONEUNIT)
  const bool cardinal_wg = false;
  const int unit = ASSET_EXP_ONEUNIT;   // (experiment: every workgroup runs the same unit body)
#else
  // PHASE 4 (one launch for the whole stage): blockIdx.y < NUNITS -- interior unit y; otherwise cardinal unit y - NUNITS
  const bool cardinal_wg = PHASE == 4 && int(blockIdx.y) >= Ode::NUNITS;
  const int unit = cardinal_wg ? int(blockIdx.y) - Ode::NUNITS : int(blockIdx.y);
#endif
  const int seg0 = int(blockIdx.x) * gp;
  const int gcount = min(gp, a.nseg - seg0);
  if (gcount <= 0) return;
  glb_double* const Wg = (glb_double*)(a.work + size_t(seg0) * D::WSLOT);
#if defined(ASSET_WALLCLOCK)   // (tuning builds, with ASSET_HIP_SKIP_DENSE: 100 MHz stamps of every workgroup, left in AGX)
  double* const wstamp = a.AGX + (size_t(blockIdx.y) * gridDim.x + blockIdx.x) * 8;
  int wn = 0;
#define ASSET_WALLCLOCK_PHASE 0      // (two launches write the same cells: which one is kept)
#define UTS() do { if (lane == 0 && a.AGX && PHASE == ASSET_WALLCLOCK_PHASE) wstamp[wn] = double(wall_clock64()); wn++; } while (0)
#else
#define UTS() do {} while (0)
#endif
  UTS();

  // ---- P0: weight tables and the group's z, lam -> LDS (unit 0 of the first launch also fills the slots' copies)
  constexpr int NTAB = (D::TABSZ + 63) / 64;
  double tabv[NTAB];                    // (requested here, written behind the other requests)
#pragma unroll
  for (int t = 0; t < NTAB; t++) tabv[t] = (lane + 64 * t < D::TABSZ) ? reinterpret_cast<const double*>(&d_lgl_tab[D::TAB])[lane + 64 * t] : 0.0;
  {
    const int* vseg = a.vindex + size_t(seg0) * IR;
    const int* cseg = a.cindex + size_t(seg0) * OR;
    // (every index first, then every value: two trips to memory instead of two per pass)
    constexpr int NZ = (UD::GP * IR + 63) / 64, NL = (UD::GP * OR + 63) / 64;
    int zi[NZ], li[NL];
    if (a.affine) {                     // index rows that are runs (EvalArgs::affine): the addresses without the tables
#pragma unroll
      for (int t = 0; t < NZ; t++) { const int e = lane + 64 * t, g = e / IR; zi[t] = (e < gcount * IR) ? a.aff_v0 + (seg0 + g) * a.aff_vs + (e - g * IR) : 0; }
#pragma unroll
      for (int t = 0; t < NL; t++) { const int e = lane + 64 * t, g = e / OR; li[t] = (e < gcount * OR) ? a.aff_c0 + (seg0 + g) * a.aff_cs + (e - g * OR) : 0; }
    } else {
#pragma unroll
      for (int t = 0; t < NZ; t++) zi[t] = (lane + 64 * t < gcount * IR) ? vseg[lane + 64 * t] : 0;
#pragma unroll
      for (int t = 0; t < NL; t++) li[t] = (lane + 64 * t < gcount * OR) ? cseg[lane + 64 * t] : 0;
    }
    double zv[NZ], lv[NL];
#pragma unroll
    for (int t = 0; t < NZ; t++) zv[t] = (lane + 64 * t < gcount * IR) ? a.X[zi[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NL; t++) lv[t] = (lane + 64 * t < gcount * OR && a.L) ? a.L[li[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NTAB; t++)
      if (lane + 64 * t < D::TABSZ) tabL[lane + 64 * t] = tabv[t];
#pragma unroll
    for (int t = 0; t < NZ; t++) {
      const int e = lane + 64 * t, g = e / IR, r = e - g * IR;
      if (e < gcount * IR) {
        mirror[g * UD::MS + UD::m_z + r] = zv[t];
        if ((PHASE == 0 || PHASE == 3 || PHASE == 4) && unit == 0 && !cardinal_wg) Wg[g * D::WSLOT + D::w_z + r] = zv[t];
      }
    }
#pragma unroll
    for (int t = 0; t < NL; t++) {
      const int e = lane + 64 * t, g = e / OR, r = e - g * OR;
      if (e < gcount * OR) {
        mirror[g * UD::MS + UD::m_lam + r] = lv[t];
        if ((PHASE == 0 || PHASE == 3 || PHASE == 4) && unit == 0 && !cardinal_wg) Wg[g * D::WSLOT + D::w_lam + r] = lv[t];
      }
    }
    if constexpr (PHASE == 1 && !D::TRAP) {   // g^_i of the first launch (every unit contributed its share)
      for (int e = lane; e < gcount * K * N; e += 64) {
        const int g = e / (K * N), r = e - g * K * N;
        mirror[g * UD::MS + UD::m_Ig + r] = Wg[g * D::WSLOT + D::w_Ig + r];
      }
    }
  }
  wave_lds_sync();
  UTS();
  const LglTab& tab = *reinterpret_cast<const LglTab*>(lds);

  if constexpr (PHASE == 0 || PHASE == 3 || PHASE == 4) {
    // ---- P1: cardinal values f_j -> mirror (this unit's own copy); unit 0 writes the slots
    if (lane < gcount * CS) {
      const int g = lane / CS, j = lane - g * CS;
      const lds_double* M = mirror + g * UD::MS;
      CardIn<D, const lds_double*> in{M + UD::m_z, nullptr, j, nullptr};
      if (unit == 0 && !cardinal_wg) {
        OdeOutValue<D, true> out{mirror + g * UD::MS + UD::m_Cf + j * n, Wg + g * D::WSLOT + D::w_Cf + j * n};
        Ode::f(in, out);
      } else {
        OdeOutValue<D, false> out{mirror + g * UD::MS + UD::m_Cf + j * n, nullptr};
        Ode::f(in, out);
      }
    }
    wave_lds_sync();
    UTS();
    // ---- P2: interior points, this unit's share of [f^, J^, g^, H^]  (Trapezoidal: none)
    if constexpr (!D::TRAP) {
      if (lane < gcount * K) {
        const int g = lane / K, i = lane - g * K;
        const lds_double* M = mirror + g * UD::MS;
        const lds_double* z = M + UD::m_z;
        const double h = z[D::TF] - z[T];
        double y[N];
        double li[n > 0 ? n : 1];
#pragma unroll
        for (int k = 0; k < n; k++) {
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < CS; j++) acc += (tab.A[i][j] * z[j * q + k] + (tab.B[i][j] * h) * M[UD::m_Cf + j * n + k]);
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
        for (int k = 0; k < n; k++) li[k] = M[UD::m_lam + i * n + k];
        RegIn<D> in{y, li};
        glb_double* S = Wg + g * D::WSLOT;
        if (cardinal_wg) {                // PHASE 4, a cardinal unit's workgroup: g^_i[0:n] for its own adjoint weights
          if constexpr (PHASE == 4) {
            OdeOutGx<D> out{mirror + g * UD::MS + UD::m_Ig + i * N};
            Ode::gx(in, out);
          }
        } else
        if constexpr (PHASE == 3) {       // Jacobian kinds: f^ (from the unit that holds f), this unit's columns of J^ and of g^ = J^^T lam
          OdeOutUnitJ<D> out{S + D::w_If + i * n, S + D::w_IJ + i * D::NZJ, li, {}};
#pragma unroll
          for (int b = 0; b < N; b++) out.gacc_[b] = 0.0;
          run_unit_j<Ode>(unit, in, out);
          const unsigned long long own = Ode::UNIT_COLS[unit];
#pragma unroll
          for (int b = 0; b < N; b++)
            if ((own >> b) & 1ull) S[D::w_Ig + i * N + b] = out.gacc_[b];
        } else {
        OdeOutUnit<D, false, true> out{S + D::w_If + i * n, S + D::w_IJ + i * D::NZJ, S + D::w_Ig + i * N, S + D::w_IH + i * D::NZH,
                                       nullptr};
#if defined(ASSET_EXP_UNITREP)   // (experiment: the unit body again -- the second pass finds its code in the instruction cache)
        for (int rep = 1; rep < ASSET_EXP_UNITREP; rep++) { run_unit<Ode>(unit, in, out); asm volatile("" ::: "memory"); }
#endif
        run_unit<Ode>(unit, in, out);
        }
      }
    }
  }
  if constexpr (PHASE == 3) {
    // ---- cardinal nodes, Jacobian kinds: this unit's columns of J_j (f_j is P1's)
    if (lane < gcount * CS) {
      const int g = lane / CS, j = lane - g * CS;
      CardIn<D, const lds_double*> in{mirror + g * UD::MS + UD::m_z, nullptr, j, nullptr};
      OdeOutUnitJ<D> out{nullptr, Wg + g * D::WSLOT + D::w_CJ + j * D::NZJ, nullptr, {}};
      run_unit_j<Ode>(unit, in, out);
    }
  }
  UTS();
  if constexpr (PHASE == 4) wave_lds_sync();     // (g^_i in the mirror)
  if constexpr (PHASE == 1 || PHASE == 4) {
    // ---- P3: cardinal nodes with the adjoint weights w_j (LGLDefects.h:369-374), this unit's share of [J, g, H]
    if ((PHASE == 1 || cardinal_wg) && lane < gcount * CS) {
      const int g = lane / CS, j = lane - g * CS;
      const lds_double* M = mirror + g * UD::MS;
      const lds_double* z = M + UD::m_z;
      const double h = z[D::TF] - z[T];
      double w[n > 0 ? n : 1];
#pragma unroll
      for (int k = 0; k < n; k++) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < K; i++) {
          if constexpr (!D::TRAP) acc += M[UD::m_Ig + i * N + k] * ((tab.E[i] * tab.B[i][j]) * h * h);
          acc += M[UD::m_lam + i * n + k] * (tab.D[i][j] * h);
        }
        w[k] = acc;
      }
      CardIn<D, const lds_double*> in{z, w, j, nullptr};
      glb_double* S = Wg + g * D::WSLOT;
      // (the unit that owns f emits f_j again: dropped, P1's value is the one every reader uses)
      OdeOutUnit<D, false, false> out{S + D::w_Cf + j * n, S + D::w_CJ + j * D::NZJ, S + D::w_Cg + j * N, S + D::w_CH + j * D::NZH, nullptr};
#if defined(ASSET_EXP_UNITREP)
      for (int rep = 1; rep < ASSET_EXP_UNITREP; rep++) { run_unit<Ode>(unit, in, out); asm volatile("" ::: "memory"); }
#endif
      run_unit<Ode>(unit, in, out);
    }
  }
  UTS();
#undef UTS
}

// The kernel proper (ODEs whose generated functor is cut into units; an empty kernel for the others, which a run-time
// compiled module still names).
template <class Ode, int SCH, bool BLOCKED, int PHASE>
__global__ __launch_bounds__(64, 1) void lgl_ode_units_kernel(EvalArgs a, int gp) {
  if constexpr (Ode::NUNITS > 1) lgl_ode_units_body<Ode, SCH, BLOCKED, PHASE>(a, gp);
}

}  // namespace asset_hip
