// ORACLE (test infrastructure): de Boor mesh-error estimate of a phase trajectory.
// Follows /root/reference/src/OptimalControl/ODEPhase.h:442-585 (get_meshinfo_deboor) statement by statement:
//   derivatives f_k = ode(traj_k) at every node (:460-462); leading-power weights of the scheme
//   (LGLCoeffs.h:44-55, 139-161, 293-392; Trapezoidal :499-504); per block
//   yvec = sum_j [ x_j XerrW_j + f_j DXerrW_j h ] / h^Order (:513-546), with the last node's derivative re-evaluated
//   under the first node's control for BlockConstant control (:529-537); neighbour differences -> err (:563-575),
//   mesh_dist = err^(1/(Order+1)), mesh_errors = err |h|^(Order+1) ErrorWeight (:577-578); the last column repeats the
//   one before (:581-582).  AutoScaling (:551-559) is not part of the accelerated path and is not restated.
#include <cmath>
#include <vector>

#include "oracle.h"

namespace {
struct Scheme {
  int cs;
  double order, error_weight;
  double xw[4], dxw[4];   // Cardinal_XPower_Weights[j][0], Cardinal_DXPower_Weights[j][0]
  bool scale_by_factorial;
};
const Scheme kSchemes[5] = {
    {},
    {2, 2.0, 1.0 / 12.0, {0.0, 0.0}, {-1.0, 1.0}, false},                                                   // Trapezoidal
    {2, 3.0, 0.0026041666661458227, {2.0, -2.0}, {1.0, 1.0}, true},                                         // LGL3
    {3, 5.0, 3.100198409908181e-06, {24.0, 0.0, -24.0}, {4.0, 16.0, 4.0}, true},                            // LGL5
    {4, 7.0, 2.9357939455472746e-09, {322.113192893432, -64.79204848488, 64.7920484849059, -322.11319289346},
     {26.2862997682608, 119.581459799146, 119.581459799146, 26.2862997682629}, true},                      // LGL7
};
}  // namespace

extern "C" int oracle_mesh_error_deboor(const oracle_ode* ode, int mode, int blocked, const double* traj, int nnodes,
                                        double* tsnd, double* mesh_errors, double* mesh_dist) {
  if (!ode || mode < 1 || mode > 4 || !traj) return -1;
  const Scheme& sc = kSchemes[mode];
  const int xv = ode->xv, uv = ode->uv, nin = xv + 1 + uv + ode->pv, tvar = xv;
  const int bs = sc.cs, nb = (nnodes - 1) / (bs - 1);
  if (nb < 2 || nb * (bs - 1) + 1 != nnodes) return -2;
  double fact = 1.0;
  for (int i = 1; i <= int(sc.order); i++) fact *= i;
  double xw[4], dxw[4];
  for (int j = 0; j < bs; j++) {
    xw[j] = sc.xw[j] * (sc.scale_by_factorial ? fact : 1.0);
    dxw[j] = sc.dxw[j] * (sc.scale_by_factorial ? fact : 1.0);
  }
  std::vector<double> derivs(size_t(nnodes) * xv);
  for (int k = 0; k < nnodes; k++) ode->f(traj + size_t(k) * nin, derivs.data() + size_t(k) * xv, ode->ctx);
  const double T0 = traj[tvar], TF = traj[size_t(nnodes - 1) * nin + tvar];
  std::vector<double> yv(size_t(nb) * xv), hs(nb), ftmp(xv), ytmp(nin);
  for (int i = 0; i < nb; i++) {
    const int start = (bs - 1) * i;
    hs[i] = traj[size_t(start + bs - 1) * nin + tvar] - traj[size_t(start) * nin + tvar];
    tsnd[i] = (traj[size_t(start) * nin + tvar] - T0) / (TF - T0);
    const double powh = std::pow(hs[i], sc.order);
    for (int k = 0; k < xv; k++) yv[size_t(i) * xv + k] = 0.0;
    for (int j = 0; j < bs; j++) {
      const double* x = traj + size_t(start + j) * nin;
      const double* f = derivs.data() + size_t(start + j) * xv;
      if (blocked && uv != 0 && j == bs - 1) {   // last node of the block under the block's (first node's) control
        for (int c = 0; c < nin; c++) ytmp[c] = x[c];
        for (int c = 0; c < uv; c++) ytmp[xv + 1 + c] = traj[size_t(start) * nin + xv + 1 + c];
        ode->f(ytmp.data(), ftmp.data(), ode->ctx);
        f = ftmp.data();
      }
      for (int k = 0; k < xv; k++) {
        yv[size_t(i) * xv + k] += x[k] * xw[j] / powh;
        yv[size_t(i) * xv + k] += f[k] * dxw[j] * hs[i] / powh;
      }
    }
  }
  tsnd[nb] = 1.0;
  for (int i = 0; i < nb; i++)
    for (int k = 0; k < xv; k++) {
      const double* y = yv.data();
      double e;
      if (i > 0 && i < nb - 1)
        e = std::fabs((y[size_t(i) * xv + k] - y[size_t(i - 1) * xv + k]) / (hs[i] + hs[i - 1])) +
            std::fabs((y[size_t(i + 1) * xv + k] - y[size_t(i) * xv + k]) / (hs[i] + hs[i + 1]));
      else if (i == 0)
        e = std::fabs(2 * (y[size_t(i) * xv + k] - y[size_t(i + 1) * xv + k]) / (hs[i] + hs[i + 1]));
      else
        e = std::fabs(2 * (y[size_t(i) * xv + k] - y[size_t(i - 1) * xv + k]) / (hs[i] + hs[i - 1]));
      mesh_dist[size_t(i) * xv + k] = std::pow(e, 1.0 / (sc.order + 1));                       // column-major xv x (nb+1)
      mesh_errors[size_t(i) * xv + k] = e * std::pow(std::fabs(hs[i]), sc.order + 1) * sc.error_weight;
    }
  for (int k = 0; k < xv; k++) {
    mesh_dist[size_t(nb) * xv + k] = mesh_dist[size_t(nb - 1) * xv + k];
    mesh_errors[size_t(nb) * xv + k] = mesh_errors[size_t(nb - 1) * xv + k];
  }
  return 0;
}
