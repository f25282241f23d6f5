#include "kkt_assembly.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <stdexcept>

namespace asset_hip_host {

KktAssembly::KktAssembly(int primal_vars, int equal_cons) : primal_(primal_vars), equal_(equal_cons) {
  if (primal_vars <= 0 || equal_cons < 0) throw std::invalid_argument("KktAssembly: bad dimensions");
}

int KktAssembly::add_equality(BatchedDefectConstraint& con, const SolverIndexingData& data) {
  if (analyzed_) throw std::logic_error("KktAssembly: add_equality after analyze");
  if (data.input_size != con.IRows() || data.output_size != con.ORows())
    throw std::invalid_argument("KktAssembly: index data does not match the constraint");
  cons_.push_back(Entry{&con, data, 0});
  return int(cons_.size()) - 1;
}

void KktAssembly::analyze() {
  // ---- getRHSSpace / getMATSpace: every function claims its coefficient rows and its (row, col) slots
  size_t n_agx = 0, n_econ = 0, n_kkt = 0;
  for (auto& e : cons_) {
    n_agx += size_t(e.data.NumAppl()) * e.data.input_size;
    n_econ += size_t(e.data.NumAppl()) * e.data.output_size;
    n_kkt += size_t(e.data.NumAppl()) * e.con->numKKTEles(true, true);
  }
  agx_rows_.assign(n_agx, -1), econ_rows_.assign(n_econ, -1);
  agx_coeffs_.assign(n_agx, 0.0), econ_coeffs_.assign(n_econ, 0.0);
  num_user_ = int(n_kkt);
  const int num_solver = primal_ + equal_;
  kkt_rows_.assign(n_kkt + num_solver, 0), kkt_cols_.assign(n_kkt + num_solver, 0);
  int gfree = 0, cfree = 0, kfree = 0;
  for (auto& e : cons_) {
    e.data.getGradientSpace(agx_rows_.data(), gfree);
    e.data.getConstraintSpace(econ_rows_.data(), cfree);
    e.kkt_start = kfree;
    e.con->getKKTSpace(kkt_rows_.data(), kkt_cols_.data(), kfree, /*conoffset=*/primal_, true, true, e.data);
  }
  for (int i = 0; i < primal_; i++) kkt_rows_[n_kkt + i] = kkt_cols_[n_kkt + i] = i;                       // PrimalDiag
  for (int i = 0; i < equal_; i++) kkt_rows_[n_kkt + primal_ + i] = kkt_cols_[n_kkt + primal_ + i] = primal_ + i;   // EPivot
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
  for (auto& e : cons_) e.con->enable_device_assembly(nnz());
  analyzed_ = true;
}

void KktAssembly::eval(int what, const double* X, const double* L, double* FXE, double* AGX, double* vals) {
  if (!analyzed_) throw std::logic_error("KktAssembly: analyze() has not been called");
  const bool want_agx = (what == ASSET_HIP_CON_ADJGRAD || what == ASSET_HIP_JAC_ADJGRAD || what == ASSET_HIP_JAC_ADJGRAD_HESS);
  const bool want_kkt = what >= ASSET_HIP_JAC;
  // setRHSCoeffsZero / setMatrixZero (NonLinearProgram.cpp:487, PSIOPT.cpp:107)
  std::fill(FXE, FXE + equal_, 0.0);
  if (want_agx) std::fill(AGX, AGX + primal_, 0.0);
  if (want_kkt) std::fill(vals, vals + nnz(), 0.0);
  for (auto& e : cons_) {
    BatchedDefectConstraint& c = *e.con;
    double* fx = econ_coeffs_.data();
    double* ag = agx_coeffs_.data();
    const int* lpt = locs_.data();
    switch (what) {
      case ASSET_HIP_CON: c.constraints(X, fx, e.data); break;
      case ASSET_HIP_CON_ADJGRAD: c.constraints_adjointgradient(X, L, fx, ag, e.data); break;
      case ASSET_HIP_JAC: c.constraints_jacobian(X, fx, vals, lpt, e.data); break;
      case ASSET_HIP_JAC_ADJGRAD: c.constraints_jacobian_adjointgradient(X, L, fx, ag, vals, lpt, e.data); break;
      default: c.constraints_jacobian_adjointgradient_adjointhessian(X, L, fx, ag, vals, lpt, e.data);
    }
  }
  // RHSFillOP (NonLinearProgram.h:401-407)
  for (size_t i = 0; i < econ_rows_.size(); i++) FXE[econ_rows_[i]] += econ_coeffs_[i];
  if (want_agx)
    for (size_t i = 0; i < agx_rows_.size(); i++) AGX[agx_rows_[i]] += agx_coeffs_[i];
  if (want_kkt)   // fillSolverCoeffs (NonLinearProgram.h:264-290)
    for (size_t i = 0; i < solver_coeffs_.size(); i++) vals[locs_[num_user_ + i]] += solver_coeffs_[i];
}

void KktAssembly::evalKKT(const double* X, const double* L, double* FXE, double* AGX, double* vals) {
  eval(ASSET_HIP_JAC_ADJGRAD_HESS, X, L, FXE, AGX, vals);
}
void KktAssembly::evalSOE(const double* X, double* FXE, double* vals) { eval(ASSET_HIP_JAC, X, nullptr, FXE, nullptr, vals); }
void KktAssembly::evalRHS(const double* X, const double* L, double* FXE, double* AGX) {
  eval(ASSET_HIP_CON_ADJGRAD, X, L, FXE, AGX, nullptr);
}
void KktAssembly::evalOCC(const double* X, double* FXE) { eval(ASSET_HIP_CON, X, nullptr, FXE, nullptr, nullptr); }
// the solver's initialisation pass (NonLinearProgram.cpp:627-683): value, J^T L and J -- no adjoint Hessian
void KktAssembly::evalAUG(const double* X, const double* L, double* FXE, double* AGX, double* vals) {
  eval(ASSET_HIP_JAC_ADJGRAD, X, L, FXE, AGX, vals);
}

}  // namespace asset_hip_host
