// Host-side shim: a phase's defect equality constraint, evaluated for all of its applications at once on an
// MI355X through the C ABI (include/asset_hip.h), exposing the method set of the reference's constraint
// plug-in interface so that it can stand where the threaded CPU evaluator stood.
//
// Interface mirrored (names, argument meaning, accumulate-vs-overwrite semantics), relative to /root/reference/src:
//   SolverConstraintSpec::Concept          VectorFunctions/VectorFunctionTypeErasure/SolverInterfaceSpecs.h:41-92
//   SizableSpec  name/IRows/ORows/thread_safe   VectorFunctions/VectorFunctionTypeErasure/SizingSpecs.h:29-39
//   SolverIndexingData                     VectorFunctions/IndexingData.h:30-210
// Differences forced by the absence of Eigen in this tree: vectors are plain pointers, the KKT matrix is
// its CSR value array (what `KKTmat.valuePtr()` is in the reference), and the per-column mutex table
// (KKTClashes / KKTLocks) is not taken: the shim evaluates every application in one device call and scatters
// on the calling thread, i.e. it is meant to be registered with ThreadMode = MainThread
// (Solvers/NonLinearProgram.cpp:86-104), as SURVEY.md section 8(b) prescribes.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/asset_hip.h"

namespace asset_hip_host {

struct SolverIndexingData {
  int input_size = 0, output_size = 0, num_funcappl = 0;
  std::vector<int> Vindex;   // [input_size x num_funcappl] column-major
  std::vector<int> Cindex;   // [output_size x num_funcappl] column-major
  std::vector<int> InnerConstraintStarts, InnerGradientStarts, InnerKKTStarts;
  int NumAppl() const { return num_funcappl; }
  int VLoc(int loc, int col) const { return Vindex[size_t(col) * input_size + loc]; }
  int CLoc(int loc, int col) const { return Cindex[size_t(col) * output_size + loc]; }
  // IndexingData.h:96-115
  void getGradientSpace(int* GXrows, int& freeloc);
  void getConstraintSpace(int* FXrows, int& freeloc);
  // IndexingData.h:117-146: contiguous chunks of the applications, the first num_funcappl % Threads one longer; fewer chunks
  // than Threads when there are fewer applications.  (The sharded constraint below splits by the same rule inside the library;
  // this is for callers that keep one index-data object per shard, as the reference's ConstraintFunction::thread_split does.)
  std::vector<SolverIndexingData> thread_split(int Threads) const;
};

class BatchedDefectConstraint {
 public:
  // mode: ASSET_HIP_LGL3/5/7 or ASSET_HIP_TRAPEZOIDAL; throws std::invalid_argument / std::runtime_error
  BatchedDefectConstraint(const std::string& ode, int mode, bool blocked, const SolverIndexingData& data,
                          int primal_vars, int equal_cons, int device = 0);
  // The same constraint as one handle per device IN THIS PROCESS (include/asset_hip.h: asset_hip_defect_create_sharded): the
  // applications are split by the ByApplication rule (SolverIndexingData::thread_split) over devices[0..n), every evaluation
  // enqueues all shards and every shard's blocks come back over its own PCIe link.  The method set, the index data and the
  // results are those of the single-handle object (the blocks bitwise) -- this is what stands where the reference registers a
  // constraint with ThreadMode::ByApplication (Solvers/NonLinearProgram.cpp:71-109), still called from ONE solver thread.
  BatchedDefectConstraint(const std::string& ode, int mode, bool blocked, const SolverIndexingData& data,
                          int primal_vars, int equal_cons, const std::vector<int>& devices);
  ~BatchedDefectConstraint();
  int num_shards() const;
  // EnableHessianSparsity of the Trapezoidal defects (OptimalControl/TrapezoidalDefects.h:39, 75-141): the block of the adjoint
  // Hessian that couples the two nodes of a segment -- without the rows / columns of their times and of the parameters -- is
  // structurally zero; with the switch on it claims no KKT slots (numKKTEles / getKKTSpace via HessianElemIsNonZero) and the fill
  // steps over it (AddHessianElem).  The device still evaluates whole blocks; the masked slots are dropped by the scatter, or by
  // the slot map of the on-device assembly.  Set before getKKTSpace; ignored for the LGL transcriptions (the reference has no
  // such mask there).
  void EnableHessianSparsity(bool on);
  bool HessianElemIsNonZero(int row, int col) const { return hess_nz_.empty() || hess_nz_[size_t(row) + size_t(ir_) * col] != 0; }
  BatchedDefectConstraint(const BatchedDefectConstraint&) = delete;
  BatchedDefectConstraint& operator=(const BatchedDefectConstraint&) = delete;

  // new index data for the same function (adaptive mesh refinement: another number of applications); keeps the device handle
  void rebind(const SolverIndexingData& data, int primal_vars, int equal_cons);
  // DeepCopySpecs.h:36-60 (deep_copy_into): an independent object with a device handle of its own
  std::unique_ptr<BatchedDefectConstraint> deep_copy(const SolverIndexingData& data) const;

  // constants of the function's applications (a plain function built with vf.ApplConst): [NumAppl][per_application]
  void set_appl_consts(const double* consts, int per_application);

  std::string name() const;
  int IRows() const { return ir_; }
  int ORows() const { return or_; }
  bool thread_safe() const { return false; }  // one evaluation at a time per handle

  // DenseFunctionBase.h:1070-1088 / 1097-1129 (every Jacobian and lower-triangle Hessian entry is structural).  The ORDER in which
  // getKKTSpace tells the solver the (row, col) of an application's slots is this function's own business -- the solver maps every
  // pair to a matrix location whatever the order (NonLinearProgram.cpp:282-330) -- and here it is the order the device writes its
  // blocks in (asset_hip_defect_kkt_layout: for the narrow transcriptions the Jacobian column-major, then the packed lower
  // triangle; the reference's `for i: {H(j>=i, i); J(:, i)}` for plain functions and wide shapes), so that the fill walks block
  // and space side by side.
  int numKKTEles(bool dojac, bool dohess) const;
  void getKKTSpace(int* KKTrows, int* KKTcols, int& freeloc, int conoffset, bool dojac, bool dohess,
                   SolverIndexingData& data) const;

  // ComputableBase.h:246-335, DenseFunctionBase.h:1145-1391.  FX/AGX slots are overwritten, KKT values accumulated.
  void constraints(const double* X, double* FX, const SolverIndexingData& data);
  void constraints_adjointgradient(const double* X, const double* L, double* FX, double* AGX,
                                   const SolverIndexingData& data);
  void constraints_jacobian(const double* X, double* FX, double* KKTvals, const int* KKTLocations,
                            const SolverIndexingData& data);
  void constraints_jacobian_adjointgradient(const double* X, const double* L, double* FX, double* AGX,
                                            double* KKTvals, const int* KKTLocations, const SolverIndexingData& data);
  void constraints_jacobian_adjointgradient_adjointhessian(const double* X, const double* L, double* FX, double* AGX,
                                                           double* KKTvals, const int* KKTLocations,
                                                           const SolverIndexingData& data);

  // On-device assembly: after this call the three Jacobian kinds add their KKT entries
  // into KKTvals on the GPU (include/asset_hip.h, asset_hip_defect_eval_assembled) and the host-side scatter below
  // is not used.  nvalues = KKTmat.nonZeros().  The slot -> value-location map is gathered from KKTLocations and
  // data.InnerKKTStarts on the first evaluation and again whenever a different KKTLocations array (or a changed one,
  // detected on a sample) is passed, i.e. after the solver re-analyses the sparsity.
  void enable_device_assembly(long long nvalues);
  void disable_device_assembly() { nvalues_ = 0; }
  bool device_assembly() const { return nvalues_ > 0; }

  // block scatter (public so it can be checked on its own): KKTFillAll / KKTFillJac, DenseFunctionBase.h:1413-1523, over blocks in
  // the handle's layout ([NumAppl][kkt_stride()]); dojac = false: KKTFillHess of a scalar objective
  // (DenseScalarFunctionBase.h:82-126), whose space holds no Jacobian slots
  void scatter_kkt(const double* kkt_blocks, bool dohess, double* KKTvals, const int* KKTLocations, const SolverIndexingData& data,
                   bool dojac = true) const;
  int kkt_stride() const { return kstride_; }
  // slot k of a block: (row, col) as asset_hip_defect_kkt_layout gives them (row >= IRows(): Jacobian row row - IRows(); -1: padding)
  const std::vector<int32_t>& kkt_rows() const { return lrows_; }
  const std::vector<int32_t>& kkt_cols() const { return lcols_; }

  // ---- the same function used as an OBJECTIVE (one output): SolverObjectiveSpec::Concept, SolverInterfaceSpecs.h:252-281;
  //      bodies DenseScalarFunctionBase.h:14-80.  Val is accumulated (+= ObjScale * f over the applications), the GX
  //      slots are overwritten with ObjScale * grad f, the Hessian entries ObjScale * hess f are accumulated into KKTvals.
  //      The index data of an objective has one Cindex row that points at multiplier 0 (the device reads ObjScale there);
  //      its KKT space must have been claimed with getKKTSpace(..., dojac = false, dohess = true, ...), and the object
  //      must have been constructed with equal_cons = 1 (its multiplier vector is the single number ObjScale).
  void objective(double ObjScale, const double* X, double& Val, const SolverIndexingData& data);
  void objective_gradient(double ObjScale, const double* X, double& Val, double* GX, const SolverIndexingData& data);
  void objective_gradient_hessian(double ObjScale, const double* X, double& Val, double* GX, double* KKTvals,
                                  const int* KKTLocations, const SolverIndexingData& data);

 private:
  void eval(int what, const double* X, const double* L, double* FX, double* AGX, double* KKTvals,
            const int* KKTLocations, const SolverIndexingData& data, bool hess_only = false);
  void unpin();
  void create(const SolverIndexingData& data);
  asset_hip_defect_t h_ = nullptr;
  asset_hip_sharded_t hs_ = nullptr;   // the sharded form (then h_ is the first shard's handle, owned by hs_)
  std::vector<int> devices_;
  std::vector<char> hess_nz_;          // empty: every Hessian entry claims a slot
  std::string ode_;
  int mode_, ir_ = 0, or_ = 0, nkkt_ = 0, nappl_ = 0, n_equal_ = 0;
  int kstride_ = 0;                    // doubles per block, and (row, col) of every slot of a block: the handle's layout
  std::vector<int32_t> lrows_, lcols_;
  bool blocked_ = false;
  int n_primal_ = 0, device_ = 0;
  std::vector<double> fx_, agx_, kkt_;
  bool pinned_ = false;
  // device assembly state
  void ensure_kkt_map(const int* KKTLocations, const SolverIndexingData& data, bool hess_only);
  bool map_hess_only_ = false;
  long long nvalues_ = 0;
  const int* map_source_ = nullptr;
  std::vector<int> map_;          // [nappl][nkkt] value location of every block entry, canonical numbering (asset_hip_defect_set_kkt_map)
};

}  // namespace asset_hip_host
