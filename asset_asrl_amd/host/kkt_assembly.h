// Host-side sparse block assembly for device-evaluated equality constraints: the part of the reference's
// NonLinearProgram that sits between the constraint plug-ins and the solver's KKT / RHS buffers, for a program whose
// equality constraints are BatchedDefectConstraint objects (defects, plain path functions).
//
// Mirrored, relative to /root/reference/src/Solvers:
//   countElems / getMATSpace / getRHSSpace / setMATDimensions / finalizeData   NonLinearProgram.cpp:41-254
//   analyzeSparsity  (upper-triangular row-major CSR, KKTLocations)              NonLinearProgram.cpp:267-344
//   evalKKT / evalSOE / evalOCC / evalRHS / evalAUG                               NonLinearProgram.cpp:347-449,473-537,590-683
//   RHSFillOP, fillSolverCoeffs                                                   NonLinearProgram.h:264-290,379-407
// The KKT system has dimension PrimalVars + EqualCons here (no slacks / inequalities: they stay with the host solver's
// own NLP); besides the user entries it carries one diagonal slot per primal variable and per equality constraint
// (PrimalDiag / EPivot coefficients), as the reference's does.
// Evaluation uses the constraints' on-device assembly, so an evaluation is: zero the value array, one device call per
// constraint that returns FX / AGX blocks and adds its KKT entries, then the two RHS fills.
#pragma once
#include <memory>
#include <vector>

#include "batched_defect_constraint.h"

namespace asset_hip_host {

class KktAssembly {
 public:
  KktAssembly(int primal_vars, int equal_cons);

  // PhaseIndexer::addEquality: the assembly keeps a reference to `con` (owned by the caller) and a copy of its index data
  int add_equality(BatchedDefectConstraint& con, const SolverIndexingData& data);

  // space + sparsity analysis; call once after the last add_equality
  void analyze();

  int kkt_dim() const { return primal_ + equal_; }
  int nnz() const { return int(inner_.size()); }
  int num_user_kkt() const { return num_user_; }
  const std::vector<int>& outer() const { return outer_; }          // CSR row starts  [kkt_dim + 1]
  const std::vector<int>& inner() const { return inner_; }          // CSR column indices [nnz]
  const std::vector<int>& kkt_locations() const { return locs_; }   // per slot, then the solver's diagonal slots
  std::vector<double>& solver_coeffs() { return solver_coeffs_; }   // PrimalDiag | EPivot values added by evalKKT/SOE

  // FXE[equal_cons], AGX[primal_vars], kkt_values[nnz] are overwritten (the reference zeroes them first)
  void evalKKT(const double* X, const double* L, double* FXE, double* AGX, double* kkt_values);   // value + J^T L + J + H
  void evalSOE(const double* X, double* FXE, double* kkt_values);                                  // value + J
  void evalRHS(const double* X, const double* L, double* FXE, double* AGX);                        // value + J^T L
  void evalOCC(const double* X, double* FXE);                                                      // value
  void evalAUG(const double* X, const double* L, double* FXE, double* AGX, double* kkt_values);   // value + J^T L + J (init pass)

 private:
  struct Entry {
    BatchedDefectConstraint* con;
    SolverIndexingData data;
    int kkt_start;
  };
  void eval(int what, const double* X, const double* L, double* FXE, double* AGX, double* kkt_values);
  int primal_, equal_, num_user_ = 0;
  std::vector<Entry> cons_;
  std::vector<int> kkt_rows_, kkt_cols_, locs_, outer_, inner_;
  std::vector<int> agx_rows_, econ_rows_;
  std::vector<double> agx_coeffs_, econ_coeffs_, solver_coeffs_;
  bool analyzed_ = false;
};

}  // namespace asset_hip_host
