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
  const double* X;
  const double* L;
  const int* vi;
  const int* ci;
  const double* ac;    // this application's constants (vf.ApplConst; asset_hip_defect_set_appl_consts) or null
  __device__ double y(int i) const { return X[vi[i]]; }
  __device__ double lam(int k) const { return L ? L[ci[k]] : 0.0; }
  __device__ double aconst(int k) const { return ac[k]; }
};

// out.J / out.g / out.H place an entry in FX, AGX or the block slot; ASM: into the solver's value array through the
// slot-ordered location map (encoding: defect_dims.h, EvalArgs::kmap)
template <class F, bool ASM>
struct FuncOut {
  using D = FuncDims<F>;
  double* fx;
  double* agx;
  double* kkt;        // block base, or the value array (ASM)
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

// LEVEL 0: value; 1: value + Jacobian (+ adjoint gradient, Hessian slots written as zero); 2: + adjoint Hessian
template <class F, int LEVEL, bool ASM>
__global__ __launch_bounds__(64) void func_kernel(EvalArgs a) {
  using D = FuncDims<F>;
  const int V = blockIdx.x * blockDim.x + threadIdx.x;
  if (V >= a.nseg) return;
  FuncIn<F> in{a.X, a.L, a.vindex + size_t(V) * D::IR, a.cindex + size_t(V) * D::OR,
               F::NACONST > 0 ? a.appl_consts + size_t(V) * F::NACONST : nullptr};
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

}  // namespace asset_hip
