// De Boor mesh-error estimate of a phase trajectory on the device.
//
// Replaces ODEPhase<DODE>::get_meshinfo_deboor (/root/reference/src/OptimalControl/ODEPhase.h:442-585): the ODE
// right-hand side at every node, the leading-power combination y_i of each block's nodes and derivatives, and the
// neighbour differences that give the per-state error and node-density estimates.  The per-block infinity norms the
// callers take next (ODEPhaseBase.h:1369-1370, ODEPhaseBase.cpp:1461-1462) are produced here as well.
// Work per block is a handful of ODE value calls, so the mapping is simply thread <-> block; the trajectory is read
// in the reference's node-row layout [x, t, u, p].
#pragma once
#include <hip/hip_runtime.h>

#include "defect_dims.h"

namespace asset_hip {

// Leading-power weights of the scheme's Hermite interpolant and its error constant:
// Cardinal_XPower_Weights[j][0], Cardinal_DXPower_Weights[j][0], Order, ErrorWeight
// (LGLCoeffs.h:44-55, 139-161, 293-392, literals digit for digit; Trapezoidal: ODEPhase.h:499-504).
// The LGL weights are multiplied by Order! when used (ODEPhase.h:475-479), the Trapezoidal ones are not.
struct MeshScheme {
  int cs;
  double order, error_weight, factorial;
  double xw[4], dxw[4];
};
__host__ __device__ constexpr MeshScheme mesh_scheme(int sch) {
  switch (sch) {
    case 1: return {2, 2.0, 1.0 / 12.0, 1.0, {0.0, 0.0, 0.0, 0.0}, {-1.0, 1.0, 0.0, 0.0}};
    case 2: return {2, 3.0, 0.0026041666661458227, 6.0, {2.0, -2.0, 0.0, 0.0}, {1.0, 1.0, 0.0, 0.0}};
    case 3: return {3, 5.0, 3.100198409908181e-06, 120.0, {24.0, 0.0, -24.0, 0.0}, {4.0, 16.0, 4.0, 0.0}};
    default:
      return {4, 7.0, 2.9357939455472746e-09, 5040.0,
              {322.113192893432, -64.79204848488, 64.7920484849059, -322.11319289346},
              {26.2862997682608, 119.581459799146, 119.581459799146, 26.2862997682629}};
  }
}

struct MeshArgs {
  int nb;               // blocks (= mesh segments); nodes = nb*(cs-1) + 1
  const double* traj;   // [nodes][N] node rows [x, t, u, p]
  double* yvec;         // [nb][n]   scratch
  double* hs;           // [nb]      scratch
  double* tsnd;         // [nb+1]
  double* errors;       // [nb+1][n]  (= xv x (nb+1) column-major, the reference's Eigen layout)
  double* dist;         // [nb+1][n]
  double* error_max;    // [nb+1] infinity norm over the states
  double* dist_max;     // [nb+1]
};

template <class D>
struct NodeIn {  // ODE input = a trajectory row, optionally with another row's controls (BlockConstant: ODEPhase.h:529-537)
  const double* row;
  const double* urow;
  __device__ double y(int i) const { return (i > D::n && i < D::n + 1 + D::ode_t::UV) ? urow[i] : row[i]; }
  __device__ double lam(int) const { return 0.0; }
};
template <int NX>
struct ValueOut {
  double v[NX > 0 ? NX : 1];
  __device__ void f(int k, double x) { v[k] = x; }
};

template <class Ode, int SCH, bool BLOCKED>
__global__ __launch_bounds__(64) void mesh_yvec_kernel(MeshArgs a) {
  using D = Dims<Ode, SCH, false>;   // sizes of the trajectory rows do not depend on the control mode
  constexpr MeshScheme sc = mesh_scheme(SCH);
  constexpr int n = D::n, N = Ode::NIN, CS = sc.cs, T = n;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nb) return;
  const double* first = a.traj + size_t(i) * (CS - 1) * N;
  const double h = first[(CS - 1) * N + T] - first[T];
  const double t0 = a.traj[T], tf = a.traj[size_t(a.nb) * (CS - 1) * N + T];
  const double ipowh = 1.0 / pow(h, sc.order);
  double y[n];
#pragma unroll
  for (int k = 0; k < n; k++) y[k] = 0.0;
#pragma unroll
  for (int j = 0; j < CS; j++) {
    const double* row = first + j * N;
    NodeIn<D> in{row, (BLOCKED && Ode::UV > 0 && j == CS - 1) ? first : row};
    ValueOut<n> out;
    Ode::f(in, out);
#pragma unroll
    for (int k = 0; k < n; k++) {
      y[k] += row[k] * (sc.xw[j] * sc.factorial) * ipowh;
      y[k] += out.v[k] * (sc.dxw[j] * sc.factorial) * h * ipowh;
    }
  }
#pragma unroll
  for (int k = 0; k < n; k++) a.yvec[size_t(i) * n + k] = y[k];
  a.hs[i] = h;
  a.tsnd[i] = (first[T] - t0) / (tf - t0);
  if (i == a.nb - 1) a.tsnd[a.nb] = 1.0;
}

// neighbour differences of y (ODEPhase.h:563-582); thread <-> block, the last column repeats the one before
// (a template only so that every translation unit may hold a copy)
template <int UNUSED = 0>
__global__ __launch_bounds__(64) void mesh_error_kernel(MeshArgs a, int n, double order, double error_weight) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nb) return;
  const double hi = a.hs[i];
  const double scale = pow(fabs(hi), order + 1.0) * error_weight, ipow = 1.0 / (order + 1.0);
  double emax = 0.0, dmax = 0.0;
  for (int k = 0; k < n; k++) {
    const double yi = a.yvec[size_t(i) * n + k];
    double e;
    if (i > 0 && i < a.nb - 1)
      e = fabs((yi - a.yvec[size_t(i - 1) * n + k]) / (hi + a.hs[i - 1])) +
          fabs((a.yvec[size_t(i + 1) * n + k] - yi) / (hi + a.hs[i + 1]));
    else if (i == 0)
      e = fabs(2.0 * (yi - a.yvec[size_t(i + 1) * n + k]) / (hi + a.hs[i + 1]));
    else
      e = fabs(2.0 * (yi - a.yvec[size_t(i - 1) * n + k]) / (hi + a.hs[i - 1]));
    const double dv = pow(e, ipow), ev = e * scale;
    a.dist[size_t(i) * n + k] = dv;
    a.errors[size_t(i) * n + k] = ev;
    emax = fmax(emax, fabs(ev));
    dmax = fmax(dmax, fabs(dv));
    if (i == a.nb - 1) {
      a.dist[size_t(a.nb) * n + k] = dv;
      a.errors[size_t(a.nb) * n + k] = ev;
    }
  }
  a.error_max[i] = emax;
  a.dist_max[i] = dmax;
  if (i == a.nb - 1) a.error_max[a.nb] = emax, a.dist_max[a.nb] = dmax;
}

}  // namespace asset_hip
