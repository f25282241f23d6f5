// Host-side sparse block assembly for device-evaluated functions: the part of the reference's NonLinearProgram that sits
// between the function plug-ins and the solver's KKT / RHS buffers, for a program whose objectives, equality and
// inequality constraints are BatchedDefectConstraint objects (defects, plain path functions, segment quadratures).
//
// Mirrored, relative to /root/reference/src/Solvers:
//   countElems / getMATSpace / getRHSSpace / setMATDimensions / finalizeData   NonLinearProgram.cpp:41-254
//   analyzeSparsity  (upper-triangular row-major CSR, KKTLocations)              NonLinearProgram.cpp:267-344
//   evalKKT / evalSOE / evalOCC / evalRHS / evalAUG                               NonLinearProgram.cpp:347-449,473-537,590-683
//   RHSFillOP, fillSolverCoeffs                                                   NonLinearProgram.h:264-290,379-407
// KKTdim = PrimalVars + SlackVars + EqualCons + InequalCons with one slack per inequality row (setMATDimensions,
// NonLinearProgram.cpp:197-199).  Objectives claim Hessian slots only (conoffset 0, dojac = false), equalities their
// Jacobian rows at PrimalVars + SlackVars + CLoc, inequalities at PrimalVars + SlackVars + EqualCons + CLoc
// (getMATSpace, :111-139).  Behind the user entries come the solver's own slots in the reference's storage order
// (setMATDimensions :201-215, finalizeData :236-254): slack Jacobian ones (row = inequality row, col = its slack),
// primal Hessian diagonal, slack Hessian diagonal, equality pivots, inequality pivots; `solver_coeffs()` holds their
// values and every evaluation that fills the matrix adds them (fillSolverCoeffs).
// Evaluation uses the functions' on-device assembly, so an evaluation is: zero the buffers, one device call per
// function that returns its FX / AGX (PGX) blocks and adds its KKT entries, then the RHS fills.
#pragma once
#include <memory>
#include <vector>

#include "batched_defect_constraint.h"

namespace asset_hip_host {

class KktAssembly {
 public:
  KktAssembly(int primal_vars, int equal_cons, int inequal_cons = 0);

  // PhaseIndexer::addEquality / addInequality / addObjective: the assembly keeps a reference to `fn` (owned by the caller)
  // and a copy of its index data.  An objective function has one output and was constructed with equal_cons = 1.
  int add_equality(BatchedDefectConstraint& fn, const SolverIndexingData& data);
  int add_inequality(BatchedDefectConstraint& fn, const SolverIndexingData& data);
  int add_objective(BatchedDefectConstraint& fn, const SolverIndexingData& data);

  // space + sparsity analysis; call once after the last add_*
  void analyze();

  int primal_vars() const { return primal_; }
  int slack_vars() const { return inequal_; }
  int kkt_dim() const { return primal_ + inequal_ + equal_ + inequal_; }
  int nnz() const { return int(inner_.size()); }
  int num_user_kkt() const { return num_user_; }
  int num_solver_kkt() const { return int(solver_coeffs_.size()); }
  const std::vector<int>& outer() const { return outer_; }          // CSR row starts  [kkt_dim + 1]
  const std::vector<int>& inner() const { return inner_; }          // CSR column indices [nnz]
  const std::vector<int>& kkt_locations() const { return locs_; }   // per user slot, then the solver's slots
  // SlackJac (inequal) | PrimalDiag (primal) | SlackDiag (inequal) | EPivot (equal) | IPivot (inequal)
  std::vector<double>& solver_coeffs() { return solver_coeffs_; }

  // The reference's evaluation entry points (NonLinearProgram.cpp:347-683).  Every output array is overwritten (the
  // reference zeroes them first); LI / FXI / PGX may be null when the program has no inequalities / objectives.
  // Returns the objective value (ObjScale * sum of the objectives).
  double evalKKT(double ObjScale, const double* X, const double* LE, const double* LI, double* PGX, double* AGX,
                 double* FXE, double* FXI, double* kkt_values);                                  // value, gradients, J, H
  double evalRHS(double ObjScale, const double* X, const double* LE, const double* LI, double* PGX, double* AGX,
                 double* FXE, double* FXI);                                                      // value, gradients
  double evalOCC(double ObjScale, const double* X, double* FXE, double* FXI);                  // values
  void evalSOE(const double* X, double* FXE, double* FXI, double* kkt_values);                 // constraint values + J
  double evalAUG(double ObjScale, const double* X, const double* LE, const double* LI, double* PGX, double* AGX,
                 double* FXE, double* FXI, double* kkt_values);                                  // value, gradients, J (init pass)

  // equality-only programs (what round 1 offered)
  void evalKKT(const double* X, const double* L, double* FXE, double* AGX, double* kkt_values) {
    evalKKT(1.0, X, L, nullptr, nullptr, AGX, FXE, nullptr, kkt_values);
  }
  void evalSOE(const double* X, double* FXE, double* kkt_values) { evalSOE(X, FXE, nullptr, kkt_values); }
  void evalRHS(const double* X, const double* L, double* FXE, double* AGX) {
    evalRHS(1.0, X, L, nullptr, nullptr, AGX, FXE, nullptr);
  }
  void evalOCC(const double* X, double* FXE) { evalOCC(1.0, X, FXE, nullptr); }
  void evalAUG(const double* X, const double* L, double* FXE, double* AGX, double* kkt_values) {
    evalAUG(1.0, X, L, nullptr, nullptr, AGX, FXE, nullptr, kkt_values);
  }

 private:
  enum Kind { OBJ = 0, EQ = 1, IQ = 2 };
  struct Entry {
    BatchedDefectConstraint* fn;
    SolverIndexingData data;
    Kind kind;
  };
  int add(Kind k, BatchedDefectConstraint& fn, const SolverIndexingData& data);
  double eval(int what, double ObjScale, const double* X, const double* LE, const double* LI, double* PGX, double* AGX,
              double* FXE, double* FXI, double* kkt_values);
  int primal_, equal_, inequal_, num_user_ = 0;
  std::vector<Entry> fns_;
  std::vector<int> kkt_rows_, kkt_cols_, locs_, outer_, inner_;
  std::vector<int> pgx_rows_, agx_rows_, econ_rows_, icon_rows_;
  std::vector<double> pgx_coeffs_, agx_coeffs_, econ_coeffs_, icon_coeffs_, solver_coeffs_;
  bool analyzed_ = false;
};

}  // namespace asset_hip_host
