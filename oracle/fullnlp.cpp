// ORACLE (test infrastructure): a NonLinearProgram with objectives, equality and inequality constraints -- the full KKT
// layout, where oracle/nlp.cpp holds the single-equality program the defect parity tests need.  Restated from
//   countElems / getMATSpace / getRHSSpace / setMATDimensions / finalizeData   /root/reference/src/Solvers/NonLinearProgram.cpp:41-254
//   analyzeSparsity                                                             NonLinearProgram.cpp:267-344
//   evalKKT / evalAUG / evalRHS / evalOCC / evalSOE                              NonLinearProgram.cpp:347-683
//   RHSFillOP, fillSolverCoeffs                                                  NonLinearProgram.h:264-290,379-407
//   objective bodies                                                             VectorFunctions/DenseScalarFunctionBase.h:14-126
//   constraint bodies / block layout                                             VectorFunctions/DenseFunctionBase.h:1070-1523
// One thread (the reference's i-thread / j-thread invariance is checked on nlp.cpp).  A function is anything with the
// "all" signature (value, Jacobian, adjoint gradient, adjoint Hessian): a defect or plain function of oracle/defect.cpp,
// or one of the path functions of oracle/pathfuncs.cpp.
#include <algorithm>
#include <cstring>
#include <functional>
#include <vector>

#include "oracle.h"

namespace {
struct Fn {
  int kind;   // 0 objective, 1 equality, 2 inequality
  int ir, orr, nappl;
  std::vector<int> vindex, cindex;   // col-major [ir x nappl], [orr x nappl]
  std::vector<int> con_starts, grad_starts, kkt_starts;
  // x[ir], lam[orr] -> fx[orr], jx[orr x ir col-major], agx[ir], hx[ir x ir col-major]
  std::function<void(int, const double*, const double*, double*, double*, double*, double*)> all;   // (application, ...)
  int VLoc(int i, int V) const { return vindex[(size_t)V * ir + i]; }
  int CLoc(int j, int V) const { return cindex[(size_t)V * orr + j]; }
};
}  // namespace

struct oracle_fullnlp {
  int primal, equal, inequal;
  std::vector<Fn> fns;
  int kktdim = 0, num_user = 0, num_solver = 0;
  std::vector<int> kkt_rows, kkt_cols, kkt_locs, outer, inner;
  std::vector<int> pgx_rows, agx_rows, econ_rows, icon_rows;
  std::vector<double> pgx_c, agx_c, econ_c, icon_c, solver_coeffs;
};

namespace {
int add_fn(oracle_fullnlp* n, int kind, int ir, int orr, int nappl, const int* vindex, const int* cindex,
           std::function<void(int, const double*, const double*, double*, double*, double*, double*)> all) {
  Fn f;
  f.kind = kind, f.ir = ir, f.orr = orr, f.nappl = nappl;
  f.vindex.assign(vindex, vindex + (size_t)ir * nappl);
  if (kind != 0) f.cindex.assign(cindex, cindex + (size_t)orr * nappl);
  f.all = std::move(all);
  n->fns.push_back(std::move(f));
  return (int)n->fns.size() - 1;
}
}  // namespace

extern "C" {

oracle_fullnlp* oracle_fullnlp_create(int primal, int equal, int inequal) {
  oracle_fullnlp* n = new oracle_fullnlp;
  n->primal = primal, n->equal = equal, n->inequal = inequal;
  return n;
}
void oracle_fullnlp_destroy(oracle_fullnlp* n) { delete n; }

// a defect (mode LGL / Trapezoidal) or a plain function (ORACLE_FUNCTION) of an oracle_ode record
int oracle_fullnlp_add(oracle_fullnlp* n, int kind, const oracle_ode* fn, int mode, int blocked, int nappl,
                       const int* vindex, const int* cindex) {
  int ir, orr;
  if (oracle_defect_sizes(mode, fn->xv, fn->uv, fn->pv, blocked, &ir, &orr)) return -1;
  if (kind == 0 && orr != 1) return -2;
  const oracle_ode o = *fn;
  return add_fn(n, kind, ir, orr, nappl, vindex, cindex,
                [o, mode, blocked](int, const double* x, const double* l, double* fx, double* jx, double* ag, double* hx) {
                  oracle_defect_all(&o, mode, blocked, x, l, fx, jx, ag, hx);
                });
}
int oracle_fullnlp_add_integral(oracle_fullnlp* n, int kind, const oracle_ode* integrand, int cs, int xv, int pv,
                                int nappl, const int* vindex, const int* cindex) {
  const oracle_ode o = *integrand;
  return add_fn(n, kind, cs * (xv + 1) + pv, 1, nappl, vindex, cindex,
                [o, cs, xv, pv](int, const double* x, const double* l, double* fx, double* jx, double* ag, double* hx) {
                  oracle_lgl_integral_all(&o, cs, xv, pv, x, l, fx, jx, ag, hx);
                });
}
int oracle_fullnlp_add_mesh_spacing(oracle_fullnlp* n, int kind, int cs, int nappl, const int* vindex, const int* cindex) {
  return add_fn(n, kind, cs, cs - 2, nappl, vindex, cindex,
                [cs](int, const double* x, const double* l, double* fx, double* jx, double* ag, double* hx) {
                  oracle_lgl_mesh_spacing_all(cs, x, l, fx, jx, ag, hx);
                });
}
int oracle_fullnlp_add_control_spline(oracle_fullnlp* n, int kind, int cs, int usize, int nappl, const int* vindex,
                                      const int* cindex) {
  return add_fn(n, kind, (2 * cs - 1) * (usize + 1), usize * (cs - 2), nappl, vindex, cindex,
                [cs, usize](int, const double* x, const double* l, double* fx, double* jx, double* ag, double* hx) {
                  oracle_control_spline_all(cs, usize, 0, x, l, fx, jx, ag, hx);
                });
}

// SingleMeshSpacing objects, one per application, each with its own spacing (ODEPhaseBase.cpp:962-985 addPartitionedEquality)
int oracle_fullnlp_add_single_mesh_spacing(oracle_fullnlp* n, int kind, const double* spacings, double scale, int nappl,
                                           const int* vindex, const int* cindex) {
  std::vector<double> sp(spacings, spacings + nappl);
  return add_fn(n, kind, 3, 1, nappl, vindex, cindex,
                [sp, scale](int V, const double* x, const double* l, double* fx, double* jx, double* ag, double* hx) {
                  oracle_single_mesh_spacing_all(sp[V], scale, x, l, fx, jx, ag, hx);
                });
}

int oracle_fullnlp_analyze(oracle_fullnlp* n) {
  const int slack = n->inequal;
  n->kktdim = n->primal + slack + n->equal + n->inequal;                     // setMATDimensions :197-199
  size_t npgx = 0, nagx = 0, nec = 0, nic = 0, nk = 0;                      // countElems :41-70
  for (auto& f : n->fns) {
    const size_t na = f.nappl;
    nk += na * ((size_t)f.ir * (f.ir + 1) / 2 + (f.kind == 0 ? 0 : (size_t)f.orr * f.ir));
    if (f.kind == 0) npgx += na * f.ir;
    else nagx += na * f.ir;
    if (f.kind == 1) nec += na * f.orr;
    if (f.kind == 2) nic += na * f.orr;
  }
  n->num_user = (int)nk;
  n->num_solver = slack + n->primal + slack + n->equal + n->inequal;         // :201-205
  n->kkt_rows.assign(nk + n->num_solver, -1);
  n->kkt_cols.assign(nk + n->num_solver, -1);
  n->pgx_rows.assign(npgx, -1), n->agx_rows.assign(nagx, -1), n->econ_rows.assign(nec, -1), n->icon_rows.assign(nic, -1);
  n->pgx_c.assign(npgx, 0.0), n->agx_c.assign(nagx, 0.0), n->econ_c.assign(nec, 0.0), n->icon_c.assign(nic, 0.0);
  n->solver_coeffs.assign(n->num_solver, 0.0);
  const int eqoffset = n->primal + slack, iqoffset = n->primal + slack + n->equal;
  int pfree = 0, gfree = 0, efree = 0, ifree = 0, kfree = 0;
  for (int kind = 0; kind < 3; kind++)                                      // getRHSSpace :177-195, getMATSpace :111-139
    for (auto& f : n->fns) {
      if (f.kind != kind) continue;
      f.grad_starts.resize(f.nappl), f.con_starts.resize(f.nappl), f.kkt_starts.resize(f.nappl);
      for (int V = 0; V < f.nappl; V++) {                                   // getGradientSpace (IndexingData.h:96-104)
        f.grad_starts[V] = (kind == 0) ? pfree : gfree;
        for (int i = 0; i < f.ir; i++) {
          if (kind == 0) n->pgx_rows[pfree++] = f.VLoc(i, V);
          else n->agx_rows[gfree++] = f.VLoc(i, V);
        }
      }
      if (kind != 0)
        for (int V = 0; V < f.nappl; V++) {                                 // getConstraintSpace (IndexingData.h:106-115)
          f.con_starts[V] = (kind == 1) ? efree : ifree;
          for (int j = 0; j < f.orr; j++) {
            if (kind == 1) n->econ_rows[efree++] = f.CLoc(j, V);
            else n->icon_rows[ifree++] = f.CLoc(j, V);
          }
        }
      const int conoffset = kind == 1 ? eqoffset : iqoffset;
      for (int V = 0; V < f.nappl; V++) {                                   // getKKTSpace (DenseFunctionBase.h:1097-1129)
        f.kkt_starts[V] = kfree;
        for (int i = 0; i < f.ir; i++) {
          for (int j = i; j < f.ir; j++) n->kkt_rows[kfree] = f.VLoc(j, V), n->kkt_cols[kfree] = f.VLoc(i, V), kfree++;
          if (kind != 0)
            for (int j = 0; j < f.orr; j++)
              n->kkt_rows[kfree] = f.CLoc(j, V) + conoffset, n->kkt_cols[kfree] = f.VLoc(i, V), kfree++;
        }
      }
    }
  {                                                                         // finalizeData :236-254
    size_t s = nk;
    for (int i = 0; i < n->inequal; i++, s++) n->kkt_cols[s] = n->primal + i, n->kkt_rows[s] = iqoffset + i;
    for (int i = 0; i < n->primal; i++, s++) n->kkt_cols[s] = n->kkt_rows[s] = i;
    for (int i = 0; i < n->inequal; i++, s++) n->kkt_cols[s] = n->kkt_rows[s] = n->primal + i;
    for (int i = 0; i < n->equal; i++, s++) n->kkt_cols[s] = n->kkt_rows[s] = eqoffset + i;
    for (int i = 0; i < n->inequal; i++, s++) n->kkt_cols[s] = n->kkt_rows[s] = iqoffset + i;
  }
  // analyzeSparsity :267-344
  const size_t ne = n->kkt_rows.size();
  std::vector<std::pair<int, int>> trip(ne);
  for (size_t i = 0; i < ne; i++) {
    int row = n->kkt_rows[i], col = n->kkt_cols[i];
    if (col > row) std::swap(row, col), n->kkt_rows[i] = row, n->kkt_cols[i] = col;
    trip[i] = {col, row};
  }
  std::vector<std::pair<int, int>> uniq(trip);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  n->outer.assign(n->kktdim + 1, 0);
  n->inner.resize(uniq.size());
  for (size_t k = 0; k < uniq.size(); k++) n->outer[uniq[k].first + 1]++, n->inner[k] = uniq[k].second;
  for (int r = 0; r < n->kktdim; r++) n->outer[r + 1] += n->outer[r];
  n->kkt_locs.assign(ne, -1);
  for (size_t i = 0; i < ne; i++) {
    const int* b = n->inner.data() + n->outer[trip[i].first];
    const int* e = n->inner.data() + n->outer[trip[i].first + 1];
    n->kkt_locs[i] = (int)(std::lower_bound(b, e, trip[i].second) - n->inner.data());
  }
  return 0;
}

int oracle_fullnlp_kkt_dim(const oracle_fullnlp* n) { return n->kktdim; }
int oracle_fullnlp_nnz(const oracle_fullnlp* n) { return (int)n->inner.size(); }
int oracle_fullnlp_num_user_kkt(const oracle_fullnlp* n) { return n->num_user; }
int oracle_fullnlp_num_solver_kkt(const oracle_fullnlp* n) { return n->num_solver; }
void oracle_fullnlp_csr(const oracle_fullnlp* n, int* outer, int* inner) {
  std::memcpy(outer, n->outer.data(), sizeof(int) * n->outer.size());
  std::memcpy(inner, n->inner.data(), sizeof(int) * n->inner.size());
}
void oracle_fullnlp_kkt_locations(const oracle_fullnlp* n, int* locs) {
  std::memcpy(locs, n->kkt_locs.data(), sizeof(int) * n->kkt_locs.size());
}
double* oracle_fullnlp_solver_coeffs(oracle_fullnlp* n) { return n->solver_coeffs.data(); }

// level: 0 values (evalOCC), 1 values + gradients (evalRHS), 2 constraint values + Jacobians (evalSOE),
//        3 values + gradients + Jacobians (evalAUG), 4 everything (evalKKT).  Outputs are ACCUMULATED into (caller zeroes).
int oracle_fullnlp_eval(oracle_fullnlp* n, int level, double ObjScale, const double* X, const double* LE, const double* LI,
                        double* val, double* PGX, double* AGX, double* FXE, double* FXI, double* vals) {
  std::fill(n->pgx_c.begin(), n->pgx_c.end(), 0.0);                         // setRHSCoeffsZero
  std::fill(n->agx_c.begin(), n->agx_c.end(), 0.0);
  std::fill(n->econ_c.begin(), n->econ_c.end(), 0.0);
  std::fill(n->icon_c.begin(), n->icon_c.end(), 0.0);
  const bool grads = (level == 1 || level == 3 || level == 4), mats = level >= 2, hess = level == 4;
  for (auto& f : n->fns) {
    if (f.kind == 0 && level == 2) continue;
    std::vector<double> x(f.ir), l(f.orr), fx(f.orr), jx((size_t)f.orr * f.ir), ag(f.ir), hx((size_t)f.ir * f.ir);
    const double* L = f.kind == 1 ? LE : LI;
    for (int V = 0; V < f.nappl; V++) {
      for (int i = 0; i < f.ir; i++) x[i] = X[f.VLoc(i, V)];
      if (f.kind == 0) l[0] = ObjScale;                                     // DenseScalarFunctionBase.h:62-63
      else
        for (int j = 0; j < f.orr; j++) l[j] = (L && grads) ? L[f.CLoc(j, V)] : 0.0;
      std::fill(fx.begin(), fx.end(), 0.0), std::fill(jx.begin(), jx.end(), 0.0);
      std::fill(ag.begin(), ag.end(), 0.0), std::fill(hx.begin(), hx.end(), 0.0);
      f.all(V, x.data(), l.data(), fx.data(), jx.data(), ag.data(), hx.data());
      if (f.kind == 0) {
        *val += fx[0] * ObjScale;
        if (grads)
          for (int i = 0; i < f.ir; i++) n->pgx_c[f.grad_starts[V] + i] = jx[i] * ObjScale;   // gx = jx^T * ObjScale
      } else {
        double* c = (f.kind == 1 ? n->econ_c.data() : n->icon_c.data()) + f.con_starts[V];
        for (int j = 0; j < f.orr; j++) c[j] = fx[j];
        if (grads)
          for (int i = 0; i < f.ir; i++) n->agx_c[f.grad_starts[V] + i] = ag[i];
      }
      if (!mats) continue;
      int freeloc = f.kkt_starts[V];                                        // KKTFillAll / KKTFillJac / KKTFillHess
      for (int i = 0; i < f.ir; i++) {
        if (f.kind == 0) {
          if (hess)
            for (int j = i; j < f.ir; j++) vals[n->kkt_locs[freeloc++]] += hx[j + (size_t)i * f.ir];
          else freeloc += f.ir - i;
          continue;
        }
        if (hess)
          for (int j = i; j < f.ir; j++) vals[n->kkt_locs[freeloc++]] += hx[j + (size_t)i * f.ir];
        else freeloc += f.ir - i;
        for (int j = 0; j < f.orr; j++) vals[n->kkt_locs[freeloc++]] += jx[j + (size_t)i * f.orr];
      }
    }
  }
  if (FXE)                                                                  // fillRHS (NonLinearProgram.h:379-407)
    for (size_t i = 0; i < n->econ_c.size(); i++) FXE[n->econ_rows[i]] += n->econ_c[i];
  if (FXI)
    for (size_t i = 0; i < n->icon_c.size(); i++) FXI[n->icon_rows[i]] += n->icon_c[i];
  if (grads && AGX)
    for (size_t i = 0; i < n->agx_c.size(); i++) AGX[n->agx_rows[i]] += n->agx_c[i];
  if (grads && PGX)
    for (size_t i = 0; i < n->pgx_c.size(); i++) PGX[n->pgx_rows[i]] += n->pgx_c[i];
  if (mats)                                                                 // fillSolverCoeffs (.h:264-290)
    for (int i = 0; i < n->num_solver; i++) vals[n->kkt_locs[n->num_user + i]] += n->solver_coeffs[i];
  return 0;
}

}  // extern "C"
