#include "kkt_assembly.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <stdexcept>

namespace asset_hip_host {

KktAssembly::KktAssembly(int primal_vars, int equal_cons, int inequal_cons)
    : primal_(primal_vars), equal_(equal_cons), inequal_(inequal_cons) {
  if (primal_vars <= 0 || equal_cons < 0 || inequal_cons < 0) throw std::invalid_argument("KktAssembly: bad dimensions");
}

int KktAssembly::add(Kind k, BatchedDefectConstraint& fn, const SolverIndexingData& data) {
  if (analyzed_) throw std::logic_error("KktAssembly: add_* after analyze");
  if (data.input_size != fn.IRows() || data.output_size != fn.ORows())
    throw std::invalid_argument("KktAssembly: index data does not match the function");
  if (k == OBJ && fn.ORows() != 1) throw std::invalid_argument("KktAssembly: an objective has one output");
  fns_.push_back(Entry{&fn, data, k});
  return int(fns_.size()) - 1;
}
int KktAssembly::add_equality(BatchedDefectConstraint& fn, const SolverIndexingData& d) { return add(EQ, fn, d); }
int KktAssembly::add_inequality(BatchedDefectConstraint& fn, const SolverIndexingData& d) { return add(IQ, fn, d); }
int KktAssembly::add_objective(BatchedDefectConstraint& fn, const SolverIndexingData& d) { return add(OBJ, fn, d); }

void KktAssembly::analyze() {
  // ---- countElems (NonLinearProgram.cpp:41-70)
  size_t n_pgx = 0, n_agx = 0, n_econ = 0, n_icon = 0, n_kkt = 0;
  for (auto& e : fns_) {
    const size_t na = size_t(e.data.NumAppl());
    n_kkt += na * e.fn->numKKTEles(e.kind != OBJ, true);
    if (e.kind == OBJ) n_pgx += na * e.data.input_size;
    else n_agx += na * e.data.input_size;
    if (e.kind == EQ) n_econ += na * e.data.output_size;
    if (e.kind == IQ) n_icon += na * e.data.output_size;
  }
  pgx_rows_.assign(n_pgx, -1), agx_rows_.assign(n_agx, -1), econ_rows_.assign(n_econ, -1), icon_rows_.assign(n_icon, -1);
  pgx_coeffs_.assign(n_pgx, 0.0), agx_coeffs_.assign(n_agx, 0.0), econ_coeffs_.assign(n_econ, 0.0), icon_coeffs_.assign(n_icon, 0.0);
  num_user_ = int(n_kkt);
  const int slack = inequal_;
  const int num_solver = slack + primal_ + slack + equal_ + inequal_;          // setMATDimensions :201-205
  kkt_rows_.assign(n_kkt + num_solver, 0), kkt_cols_.assign(n_kkt + num_solver, 0);
  // ---- getRHSSpace (:177-195) and getMATSpace (:111-139): objectives, then equalities, then inequalities
  int pfree = 0, gfree = 0, efree = 0, ifree = 0, kfree = 0;
  const int eqoffset = primal_ + slack, iqoffset = primal_ + slack + equal_;
  for (Kind k : {OBJ, EQ, IQ})
    for (auto& e : fns_) {
      if (e.kind != k) continue;
      if (k == OBJ) {
        e.data.getGradientSpace(pgx_rows_.data(), pfree);
        e.fn->getKKTSpace(kkt_rows_.data(), kkt_cols_.data(), kfree, 0, false, true, e.data);
      } else if (k == EQ) {
        e.data.getGradientSpace(agx_rows_.data(), gfree);
        e.data.getConstraintSpace(econ_rows_.data(), efree);
        e.fn->getKKTSpace(kkt_rows_.data(), kkt_cols_.data(), kfree, eqoffset, true, true, e.data);
      } else {
        e.data.getGradientSpace(agx_rows_.data(), gfree);
        e.data.getConstraintSpace(icon_rows_.data(), ifree);
        e.fn->getKKTSpace(kkt_rows_.data(), kkt_cols_.data(), kfree, iqoffset, true, true, e.data);
      }
    }
  // ---- finalizeData (:236-254): the solver's slots, in storage order SlackJac | PrimalDiag | SlackDiag | EPivot | IPivot
  {
    size_t s = n_kkt;
    for (int i = 0; i < inequal_; i++, s++) kkt_cols_[s] = primal_ + i, kkt_rows_[s] = iqoffset + i;
    for (int i = 0; i < primal_; i++, s++) kkt_cols_[s] = kkt_rows_[s] = i;
    for (int i = 0; i < inequal_; i++, s++) kkt_cols_[s] = kkt_rows_[s] = primal_ + i;
    for (int i = 0; i < equal_; i++, s++) kkt_cols_[s] = kkt_rows_[s] = eqoffset + i;
    for (int i = 0; i < inequal_; i++, s++) kkt_cols_[s] = kkt_rows_[s] = iqoffset + i;
  }
  solver_coeffs_.assign(num_solver, 0.0);

  // ---- analyzeSparsity: the slots name the lower triangle; the solver wants the upper triangle of a row-major CSR, so
  //      an entry (row >= col) is filed in CSR row `col`, column `row`.  Duplicates share one location.
  const size_t total = kkt_rows_.size();
  const int dim = kkt_dim();
  for (size_t i = 0; i < total; i++)
    if (kkt_cols_[i] > kkt_rows_[i]) std::swap(kkt_rows_[i], kkt_cols_[i]);
  std::vector<int> count(dim + 1, 0);
  for (size_t i = 0; i < total; i++) count[kkt_cols_[i] + 1]++;
  std::partial_sum(count.begin(), count.end(), count.begin());
  std::vector<int> bucket(total);                       // CSR-row buckets of (column = original row), unsorted
  {
    std::vector<int> fill(count.begin(), count.end() - 1);
    for (size_t i = 0; i < total; i++) bucket[fill[kkt_cols_[i]]++] = kkt_rows_[i];
  }
  outer_.assign(dim + 1, 0);
  inner_.clear();
  inner_.reserve(total);
  for (int r = 0; r < dim; r++) {
    auto b = bucket.begin() + count[r], e = bucket.begin() + count[r + 1];
    std::sort(b, e);
    e = std::unique(b, e);
    inner_.insert(inner_.end(), b, e);
    outer_[r + 1] = int(inner_.size());
  }
  locs_.assign(total, -1);
  for (size_t i = 0; i < total; i++) {
    const int r = kkt_cols_[i], c = kkt_rows_[i];
    auto b = inner_.begin() + outer_[r], e = inner_.begin() + outer_[r + 1];
    locs_[i] = int(std::lower_bound(b, e, c) - inner_.begin());
  }
  for (auto& e : fns_) e.fn->enable_device_assembly(nnz());
  analyzed_ = true;
}

double KktAssembly::eval(int what, double ObjScale, const double* X, const double* LE, const double* LI, double* PGX,
                         double* AGX, double* FXE, double* FXI, double* vals) {
  if (!analyzed_) throw std::logic_error("KktAssembly: analyze() has not been called");
  const bool want_agx = (what == ASSET_HIP_CON_ADJGRAD || what == ASSET_HIP_JAC_ADJGRAD || what == ASSET_HIP_JAC_ADJGRAD_HESS);
  const bool want_kkt = what >= ASSET_HIP_JAC;
  // setRHSCoeffsZero / setMatrixZero (NonLinearProgram.cpp:487, PSIOPT.cpp:107)
  if (FXE) std::fill(FXE, FXE + equal_, 0.0);
  if (FXI) std::fill(FXI, FXI + inequal_, 0.0);
  if (want_agx && AGX) std::fill(AGX, AGX + primal_, 0.0);
  if (want_agx && PGX) std::fill(PGX, PGX + primal_, 0.0);
  if (want_kkt) std::fill(vals, vals + nnz(), 0.0);
  double val = 0.0;
  const int* lpt = locs_.data();
  for (auto& e : fns_) {
    BatchedDefectConstraint& c = *e.fn;
    if (e.kind == OBJ) {
      if (what == ASSET_HIP_JAC) continue;               // evalSOE: constraints only
      double* pg = pgx_coeffs_.data();
      switch (what) {
        case ASSET_HIP_CON: c.objective(ObjScale, X, val, e.data); break;
        case ASSET_HIP_CON_ADJGRAD:
        case ASSET_HIP_JAC_ADJGRAD: c.objective_gradient(ObjScale, X, val, pg, e.data); break;   // (evalAUG: no objective Hessian)
        default: c.objective_gradient_hessian(ObjScale, X, val, pg, vals, lpt, e.data);
      }
      continue;
    }
    const double* L = (e.kind == EQ) ? LE : LI;
    double* fx = (e.kind == EQ) ? econ_coeffs_.data() : icon_coeffs_.data();
    double* ag = agx_coeffs_.data();
    switch (what) {
      case ASSET_HIP_CON: c.constraints(X, fx, e.data); break;
      case ASSET_HIP_CON_ADJGRAD: c.constraints_adjointgradient(X, L, fx, ag, e.data); break;
      case ASSET_HIP_JAC: c.constraints_jacobian(X, fx, vals, lpt, e.data); break;
      case ASSET_HIP_JAC_ADJGRAD: c.constraints_jacobian_adjointgradient(X, L, fx, ag, vals, lpt, e.data); break;
      default: c.constraints_jacobian_adjointgradient_adjointhessian(X, L, fx, ag, vals, lpt, e.data);
    }
  }
  // RHSFillOP (NonLinearProgram.h:401-407)
  if (FXE)
    for (size_t i = 0; i < econ_rows_.size(); i++) FXE[econ_rows_[i]] += econ_coeffs_[i];
  if (FXI)
    for (size_t i = 0; i < icon_rows_.size(); i++) FXI[icon_rows_[i]] += icon_coeffs_[i];
  if (want_agx && AGX)
    for (size_t i = 0; i < agx_rows_.size(); i++) AGX[agx_rows_[i]] += agx_coeffs_[i];
  if (want_agx && PGX)
    for (size_t i = 0; i < pgx_rows_.size(); i++) PGX[pgx_rows_[i]] += pgx_coeffs_[i];
  if (want_kkt)   // fillSolverCoeffs (NonLinearProgram.h:264-290)
    for (size_t i = 0; i < solver_coeffs_.size(); i++) vals[locs_[num_user_ + i]] += solver_coeffs_[i];
  return val;
}

double KktAssembly::evalKKT(double ObjScale, const double* X, const double* LE, const double* LI, double* PGX, double* AGX,
                            double* FXE, double* FXI, double* vals) {
  return eval(ASSET_HIP_JAC_ADJGRAD_HESS, ObjScale, X, LE, LI, PGX, AGX, FXE, FXI, vals);
}
double KktAssembly::evalRHS(double ObjScale, const double* X, const double* LE, const double* LI, double* PGX, double* AGX,
                            double* FXE, double* FXI) {
  return eval(ASSET_HIP_CON_ADJGRAD, ObjScale, X, LE, LI, PGX, AGX, FXE, FXI, nullptr);
}
double KktAssembly::evalOCC(double ObjScale, const double* X, double* FXE, double* FXI) {
  return eval(ASSET_HIP_CON, ObjScale, X, nullptr, nullptr, nullptr, nullptr, FXE, FXI, nullptr);
}
void KktAssembly::evalSOE(const double* X, double* FXE, double* FXI, double* vals) {
  eval(ASSET_HIP_JAC, 1.0, X, nullptr, nullptr, nullptr, nullptr, FXE, FXI, vals);
}
// the solver's initialisation pass (NonLinearProgram.cpp:627-683): values, gradients and J -- no adjoint Hessian
double KktAssembly::evalAUG(double ObjScale, const double* X, const double* LE, const double* LI, double* PGX, double* AGX,
                            double* FXE, double* FXI, double* vals) {
  return eval(ASSET_HIP_JAC_ADJGRAD, ObjScale, X, LE, LI, PGX, AGX, FXE, FXI, vals);
}

}  // namespace asset_hip_host
