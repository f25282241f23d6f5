// Plain vector functions batched over applications.
//
// The other per-segment functions of a phase -- user path constraints, mesh-spacing and control-spline relations,
// integrands -- go through the same solver interface as the defect: gather x = X[Vindex(:,V)], lam = L[Cindex(:,V)],
// evaluate, write FX / AGX and the KKT block of application V (/root/reference/src/VectorFunctions/
// ComputableBase.h:246-335, DenseFunctionBase.h:1145-1391; block slot order DenseFunctionBase.h:1112-1123).  Here the
// function is a generated functor (value, Jacobian, adjoint gradient and adjoint Hessian from the expression graph,
// vf/codegen.py) and one thread evaluates one application: these functions are a few dozen operations on a handful
// of inputs, there is no contraction to share between lanes, and the block of an application is at most a few
// hundred bytes.  Registered under transcription id 0, so every entry point of the C ABI that takes a defect handle
// takes such a function as well.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "defect_dims.h"

namespace asset_hip {

template <class F>
struct FuncDims {
  static constexpr int IR = F::NIN, OR = F::XV;
  static constexpr int NKKT = IR * (IR + 1) / 2 + OR * IR;
  __host__ __device__ static constexpr int col_start(int c) { return c * (IR + OR) - c * (c - 1) / 2; }
};

template <class F>
struct FuncIn {
  using D = FuncDims<F>;
  double yv[D::IR];                 // x = X[Vindex(:,V)]
  double lv[D::OR > 0 ? D::OR : 1]; // lam = L[Cindex(:,V)] (zero without multipliers)
  const double* ac;                 // this application's constants (vf.ApplConst; asset_hip_defect_set_appl_consts) or null
  // The gather, up front and in two batches: every index load is issued before the first value load, the multipliers with
  // the variables.  Read where the generated body uses them -- `L ? L[ci[k]] : 0.0` at every use -- the multipliers of a
  // four-row function became four serial index-then-value round trips (8 memory latencies; 13 of the 20 us the control
  // spline of a 10 000-segment phase took).
  __device__ void gather(const EvalArgs& a, int V) {
    const int* vi = a.vindex + size_t(V) * D::IR;
    const int* ci = a.cindex + size_t(V) * D::OR;
    int vix[D::IR], cix[D::OR > 0 ? D::OR : 1];
#pragma unroll
    for (int i = 0; i < D::IR; i++) vix[i] = vi[i];
    const bool hasl = a.L != nullptr;
#pragma unroll
    for (int k = 0; k < D::OR; k++) cix[k] = hasl ? ci[k] : 0;
#pragma unroll
    for (int i = 0; i < D::IR; i++) yv[i] = a.X[vix[i]];
#pragma unroll
    for (int k = 0; k < D::OR; k++) lv[k] = hasl ? a.L[cix[k]] : 0.0;
    ac = F::NACONST > 0 ? a.appl_consts + size_t(V) * F::NACONST : nullptr;
  }
  __device__ double y(int i) const { return yv[i]; }
  __device__ double lam(int k) const { return lv[k]; }
  __device__ double aconst(int k) const { return ac[k]; }
};

// Applications per 64-lane workgroup when the blocks are staged in LDS (block kinds, not assembled): every lane that
// evaluates writes its block into its own LDS row and the wave copies the rows -- the blocks of consecutive applications
// are one contiguous range of the output -- with coalesced stores.  One thread per application storing straight to its
// block puts the 64 lanes of every store instruction on 64 different cache lines (a control-spline block is 2.5 KB:
// 10 000 of them took 21 us).  As many applications (a power of two, at least 4) as fit 40 KiB of LDS: four such workgroups per CU.
#define ASSET_FUNC_LDS_BUDGET (40 * 1024 - 64)   // four such workgroups per CU (one per SIMD)
#define ASSET_FUNC_COPY_UNROLL 8
#define ASSET_FUNC_WAVES 1
template <class F>
struct FuncStage {
  using D = FuncDims<F>;
  static constexpr int LD = D::NKKT | 1;                      // odd row stride: conflict-free row-wise writes
  static constexpr int BUDGET = ASSET_FUNC_LDS_BUDGET;
  static constexpr int APW = (64 * LD * 8 <= BUDGET) ? 64 : (32 * LD * 8 <= BUDGET) ? 32 : (16 * LD * 8 <= BUDGET) ? 16
                             : (8 * LD * 8 <= BUDGET) ? 8 : (4 * LD * 8 <= BUDGET) ? 4 : 0;   // 0: too large, store directly
  // (the rows start two doubles into the LDS: an LDS pointer to offset 0 compares equal to null, and "no block wanted" is a
  //  null block pointer in FuncOut)
  static constexpr int ROW0 = 2;
  static constexpr size_t lds_bytes() { return size_t(ROW0 + APW * LD) * 8; }
};

// out.J / out.g / out.H place an entry in FX, AGX or the block slot; ASM: into the solver's value array through the
// slot-ordered location map (encoding: defect_dims.h, EvalArgs::kmap); STG: the block is this lane's LDS row
template <class F, bool ASM, bool STG = false>
struct FuncOut {
  using D = FuncDims<F>;
  using KP = std::conditional_t<STG, lds_double*, double*>;
  double* fx;
  double* agx;
  KP kkt;             // block base, or the value array (ASM)
  const int* kmap;    // this application's slot -> location entries (ASM)
  bool hess;
  const EvalArgs* args;
  __device__ void put(int slot, double v) {
    if constexpr (ASM) asm_put(*args, kkt, kmap[slot], v);   // (-1: a dropped slot, e.g. the Jacobian of an objective)
    else kkt[slot] = v;
  }
  __device__ void f(int k, double v) { if (fx) fx[k] = v; }
  __device__ void J(int k, int i, double v) { if (kkt) put(D::col_start(i) + (D::IR - i) + k, v); }
  __device__ void g(int i, double v) { if (agx) agx[i] = v; }
  __device__ void H(int i, int j, double v) { if (kkt && hess) put(D::col_start(j) + (i - j), v); }   // j <= i
};

// LEVEL 0: value; 1: value + Jacobian (+ adjoint gradient, Hessian slots written as zero); 2: + adjoint Hessian.
// Launch (registry.h: launch_func_table): 64 lanes per workgroup; the block kinds (LEVEL >= 1, not ASM) of a function with
// FuncStage::APW > 0 take APW applications per workgroup and FuncStage::lds_bytes() of dynamic LDS, everything else 64.
template <class F, int LEVEL, bool ASM>
__device__ __forceinline__ void func_body(const EvalArgs& a, int block) {   // block: index of this workgroup within the function's grid
  using D = FuncDims<F>;
  using ST = FuncStage<F>;
  constexpr bool STG = !ASM && LEVEL >= 1 && ST::APW > 0;
  constexpr int APW = STG ? ST::APW : 64;
  const int lane = threadIdx.x;
  const int V0 = block * APW, V = V0 + lane;
  const bool active = lane < APW && V < a.nseg;
  if constexpr (STG) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const bool staged = a.KKT != nullptr;
#if defined(ASSET_FUNC_TIMING)
    long long ts0 = clock64(), ts1 = 0, ts2 = 0, ts3 = 0;
#endif
    if (active) {
      FuncIn<F> in;
      in.gather(a, V);
#if defined(ASSET_FUNC_TIMING)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ts1 = clock64();
#endif
      FuncOut<F, false, true> out{a.FX ? a.FX + size_t(V) * D::OR : nullptr, a.AGX ? a.AGX + size_t(V) * D::IR : nullptr,
                                  staged ? (lds_double*)lds + ST::ROW0 + lane * ST::LD : nullptr, nullptr, LEVEL >= 2, &a};
      F::fjgh(in, out);   // (LEVEL 1: its Hessian entries are dropped, out.hess == false ...
      if constexpr (LEVEL == 1) {   // ... and the Hessian slots of the block are written as zero, as the LGL kernels do)
        if (out.kkt)
          for (int c = 0; c < D::IR; c++)
            for (int r = c; r < D::IR; r++) out.kkt[D::col_start(c) + (r - c)] = 0.0;
      }
    }
#if defined(ASSET_FUNC_TIMING)
    ts2 = clock64();
#endif
    if (staged) {
      wave_lds_sync();
      const int napp = min(APW, a.nseg - V0);
      double* dst = a.KKT + size_t(V0) * D::NKKT;
      const lds_double* src = (const lds_double*)lds + ST::ROW0;
      const int total = napp * D::NKKT;
      constexpr int CU = ASSET_FUNC_COPY_UNROLL;
      for (int base = 0; base < total; base += CU * 64) {   // CU elements per lane in flight: all LDS reads of a trip
        double v[CU];                                       // are issued before its first store (one read-wait-store per
#pragma unroll                                              // trip exposes a full LDS latency every time)
        for (int u = 0; u < CU; u++) {
          const int e = base + 64 * u + lane;
          const int app = (e < total) ? e / D::NKKT : 0, sl = (e < total) ? e - app * D::NKKT : 0;
          v[u] = src[app * ST::LD + sl];
        }
#pragma unroll
        for (int u = 0; u < CU; u++) {
          const int e = base + 64 * u + lane;
          if (e < total) dst[e] = v[u];
        }
      }
#if defined(ASSET_FUNC_TIMING)   // (tuning builds) gather / body / copy-out cycles of workgroup 7, left in its first application's FX
      ts3 = clock64();
      if (block == 7 && lane == 0 && a.FX && D::OR >= 3) {
        a.FX[size_t(V0) * D::OR + 0] = double(ts1 - ts0);
        a.FX[size_t(V0) * D::OR + 1] = double(ts2 - ts1);
        a.FX[size_t(V0) * D::OR + 2] = double(ts3 - ts2);
      }
#endif
    }
  } else {
    if (!active) return;
    FuncIn<F> in;
    in.gather(a, V);
    FuncOut<F, ASM> out{a.FX ? a.FX + size_t(V) * D::OR : nullptr, a.AGX ? a.AGX + size_t(V) * D::IR : nullptr,
                        ASM ? a.values : (a.KKT ? a.KKT + size_t(V) * D::NKKT : nullptr),
                        ASM ? a.kmap + size_t(V) * D::NKKT : nullptr, LEVEL >= 2, &a};
    if constexpr (LEVEL == 0) F::f(in, out);
    else if constexpr (LEVEL == 2) F::fjgh(in, out);
    else {
      // fjgh also delivers g = J^T lam; its Hessian entries are dropped (out.hess == false) ...
      F::fjgh(in, out);
      if constexpr (!ASM) {   // ... and the Hessian slots of the block are written as zero, as the LGL kernels do
        if (out.kkt)
          for (int c = 0; c < D::IR; c++)
            for (int r = c; r < D::IR; r++) out.kkt[D::col_start(c) + (r - c)] = 0.0;
      }
    }
  }
}

template <class F, int LEVEL, bool ASM>
__global__ __launch_bounds__(64, ASSET_FUNC_WAVES) void func_kernel(EvalArgs a) {
  func_body<F, LEVEL, ASM>(a, int(blockIdx.x));
}

// ---- several functions in ONE launch ----------------------------------------------------------------------------------
// What a phase hands the solver beside its defects -- mesh spacing, nodal spacing, control splines, path constraints,
// integrands -- are five or six functions of a few hundred to a few thousand workgroups each: launched one after the
// other every one of them costs its own ramp (3.5-6 us apiece for 10 000 segments, none fills the device).  A bundle is a
// kernel over a list of functors: workgroups [start[k], start[k+1]) evaluate function k with its own arguments (index
// tables, outputs, multipliers).  Compiled at run time for the list at hand (rtc_device.h: ASSET_RTC_BUNDLE).
constexpr int BUNDLE_MAX = 8;
struct BundleArgs {
  EvalArgs a[BUNDLE_MAX];
  int start[BUNDLE_MAX + 1];   // first workgroup of every function; start[n] = the grid
  int n;
};
template <int LEVEL, int I, class F, class... Rest>
__device__ __forceinline__ void bundle_run(const BundleArgs& b, int k, int block) {
  if (k == I) func_body<F, LEVEL, false>(b.a[I], block);
  else if constexpr (sizeof...(Rest) > 0) bundle_run<LEVEL, I + 1, Rest...>(b, k, block);
}
template <int LEVEL, class... Fs>
__global__ __launch_bounds__(64, ASSET_FUNC_WAVES) void func_bundle_kernel(BundleArgs b) {
  static_assert(sizeof...(Fs) >= 1 && sizeof...(Fs) <= BUNDLE_MAX, "a bundle holds 1..BUNDLE_MAX functions");
  int k = 0, first = 0;
#pragma unroll
  for (int i = 1; i < int(sizeof...(Fs)); i++)
    if (int(blockIdx.x) >= b.start[i]) k = i, first = b.start[i];
  bundle_run<LEVEL, 0, Fs...>(b, k, int(blockIdx.x) - first);
}

}  // namespace asset_hip
