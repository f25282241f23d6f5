#include "batched_defect_constraint.h"

#include <cstring>
#include <stdexcept>

namespace asset_hip_host {

void SolverIndexingData::getGradientSpace(int* GXrows, int& freeloc) {
  InnerGradientStarts.resize(NumAppl());
  for (int V = 0; V < NumAppl(); V++) {
    InnerGradientStarts[V] = freeloc;
    for (int i = 0; i < input_size; i++) GXrows[freeloc++] = VLoc(i, V);
  }
}
void SolverIndexingData::getConstraintSpace(int* FXrows, int& freeloc) {
  InnerConstraintStarts.resize(NumAppl());
  for (int V = 0; V < NumAppl(); V++) {
    InnerConstraintStarts[V] = freeloc;
    for (int j = 0; j < output_size; j++) FXrows[freeloc++] = CLoc(j, V);
  }
}

static void check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + asset_hip_last_error());
}

BatchedDefectConstraint::BatchedDefectConstraint(const std::string& ode, int mode, bool blocked,
                                                 const SolverIndexingData& data, int primal_vars, int equal_cons,
                                                 int device)
    : ode_(ode), mode_(mode), nappl_(data.NumAppl()) {
  if (data.NumAppl() <= 0) throw std::invalid_argument("BatchedDefectConstraint: no function applications");
  asset_hip_defect_desc d;
  std::memset(&d, 0, sizeof d);
  d.mode = mode;
  d.blocked = blocked ? 1 : 0;
  d.ode = ode_.c_str();
  d.nseg = data.NumAppl();
  d.vindex = data.Vindex.data();
  d.cindex = data.Cindex.data();
  d.n_primal = primal_vars;
  d.n_equal = equal_cons;
  d.device = device;
  int rc = asset_hip_defect_create(&d, &h_);
  if (rc == ASSET_HIP_EINVAL || rc == ASSET_HIP_ENOODE || rc == ASSET_HIP_ERANGE)
    throw std::invalid_argument(std::string("BatchedDefectConstraint: ") + asset_hip_last_error());
  check(rc, "asset_hip_defect_create");
  check(asset_hip_defect_sizes(h_, &ir_, &or_, &nkkt_), "asset_hip_defect_sizes");
  if (ir_ != data.input_size || or_ != data.output_size) {
    asset_hip_defect_destroy(h_);
    h_ = nullptr;
    throw std::invalid_argument("BatchedDefectConstraint: index data rows do not match the defect's IRows/ORows");
  }
  fx_.resize(size_t(nappl_) * or_);
  agx_.resize(size_t(nappl_) * ir_);
  kkt_.resize(size_t(nappl_) * nkkt_);
  // the staging arrays live as long as the constraint: page-lock them so the block copies run at PCIe rate
  pinned_ = asset_hip_host_register(fx_.data(), fx_.size() * sizeof(double)) == 0 &&
            asset_hip_host_register(agx_.data(), agx_.size() * sizeof(double)) == 0 &&
            asset_hip_host_register(kkt_.data(), kkt_.size() * sizeof(double)) == 0;
}

BatchedDefectConstraint::~BatchedDefectConstraint() {
  if (pinned_) {
    (void)asset_hip_host_unregister(fx_.data());
    (void)asset_hip_host_unregister(agx_.data());
    (void)asset_hip_host_unregister(kkt_.data());
  }
  asset_hip_defect_destroy(h_);
}

std::string BatchedDefectConstraint::name() const {
  const char* m = mode_ == ASSET_HIP_TRAPEZOIDAL ? "Trapezoidal" : (mode_ == 2 ? "LGL3" : (mode_ == 3 ? "LGL5" : "LGL7"));
  return std::string("HIP_") + m + "Defects<" + ode_ + ">";
}

int BatchedDefectConstraint::numKKTEles(bool dojac, bool dohess) const {
  return (dohess ? ir_ * (ir_ + 1) / 2 : 0) + (dojac ? or_ * ir_ : 0);
}

void BatchedDefectConstraint::getKKTSpace(int* KKTrows, int* KKTcols, int& freeloc, int conoffset, bool dojac,
                                          bool dohess, SolverIndexingData& data) const {
  data.InnerKKTStarts.resize(data.NumAppl());
  for (int V = 0; V < data.NumAppl(); V++) {
    data.InnerKKTStarts[V] = freeloc;
    for (int i = 0; i < ir_; i++) {
      if (dohess)
        for (int j = i; j < ir_; j++) {
          KKTrows[freeloc] = data.VLoc(j, V);
          KKTcols[freeloc] = data.VLoc(i, V);
          freeloc++;
        }
      if (dojac)
        for (int j = 0; j < or_; j++) {
          KKTrows[freeloc] = data.CLoc(j, V) + conoffset;
          KKTcols[freeloc] = data.VLoc(i, V);
          freeloc++;
        }
    }
  }
}

void BatchedDefectConstraint::scatter_kkt(const double* blocks, int nkkt, int ir, int orr, bool dohess,
                                          double* KKTvals, const int* lpt, const SolverIndexingData& data) {
  for (int V = 0; V < data.NumAppl(); V++) {
    const double* blk = blocks + size_t(V) * nkkt;
    int freeloc = data.InnerKKTStarts[V];
    int k = 0;
    for (int i = 0; i < ir; i++) {
      if (dohess) {
        for (int j = i; j < ir; j++) KKTvals[lpt[freeloc++]] += blk[k++];
      } else {  // KKTFillJac: the Hessian slots exist in the layout but are skipped
        freeloc += ir - i;
        k += ir - i;
      }
      for (int j = 0; j < orr; j++) KKTvals[lpt[freeloc++]] += blk[k++];
    }
  }
}

void BatchedDefectConstraint::eval(int what, const double* X, const double* L, double* FX, double* AGX,
                                   double* KKTvals, const int* KKTLocations, const SolverIndexingData& data) {
  if (data.NumAppl() != nappl_) throw std::invalid_argument("index data does not belong to this constraint");
  const bool want_agx = (what == ASSET_HIP_CON_ADJGRAD || what == ASSET_HIP_JAC_ADJGRAD || what == ASSET_HIP_JAC_ADJGRAD_HESS);
  const bool want_kkt = what >= ASSET_HIP_JAC;
  const bool assembled = want_kkt && nvalues_ > 0;
  if (assembled) {
    ensure_kkt_map(KKTLocations, data);
    check(asset_hip_defect_eval_assembled(h_, what, X, L, fx_.data(), want_agx ? agx_.data() : nullptr, KKTvals),
          "asset_hip_defect_eval_assembled");
  } else {
    check(asset_hip_defect_eval(h_, what, X, L, fx_.data(), want_agx ? agx_.data() : nullptr,
                                want_kkt ? kkt_.data() : nullptr),
          "asset_hip_defect_eval");
  }
  for (int V = 0; V < nappl_; V++) {  // callee overwrites its FX / AGX slots (fx.setZero(); compute)
    std::memcpy(FX + data.InnerConstraintStarts[V], fx_.data() + size_t(V) * or_, sizeof(double) * or_);
    if (want_agx) std::memcpy(AGX + data.InnerGradientStarts[V], agx_.data() + size_t(V) * ir_, sizeof(double) * ir_);
  }
  if (want_kkt && !assembled)
    scatter_kkt(kkt_.data(), nkkt_, ir_, or_, what == ASSET_HIP_JAC_ADJGRAD_HESS, KKTvals, KKTLocations, data);
}

void BatchedDefectConstraint::enable_device_assembly(long long nvalues) {
  if (nvalues <= 0) throw std::invalid_argument("enable_device_assembly: the value array length must be positive");
  nvalues_ = nvalues;
  map_source_ = nullptr;
}

void BatchedDefectConstraint::ensure_kkt_map(const int* lpt, const SolverIndexingData& data) {
  if (!lpt) throw std::invalid_argument("KKTLocations is null");
  if (int(data.InnerKKTStarts.size()) != nappl_) throw std::invalid_argument("InnerKKTStarts not filled: call getKKTSpace first");
  bool fresh = (lpt == map_source_) && map_.size() == size_t(nappl_) * nkkt_;
  if (fresh) {  // same array: make sure the solver did not re-fill it in place (cheap sample, 64 slots)
    const size_t n = map_.size(), step = n / 64 + 1;
    for (size_t s = 0; s < n && fresh; s += step) {
      const size_t V = s / nkkt_, k = s - V * nkkt_;
      fresh = (map_[s] == lpt[data.InnerKKTStarts[V] + k]);
    }
  }
  if (fresh) return;
  map_.resize(size_t(nappl_) * nkkt_);
  for (int V = 0; V < nappl_; V++)
    std::memcpy(map_.data() + size_t(V) * nkkt_, lpt + data.InnerKKTStarts[V], sizeof(int) * nkkt_);
  check(asset_hip_defect_set_kkt_map(h_, map_.data(), nvalues_, 0), "asset_hip_defect_set_kkt_map");
  map_source_ = lpt;
}

void BatchedDefectConstraint::constraints(const double* X, double* FX, const SolverIndexingData& data) {
  eval(ASSET_HIP_CON, X, nullptr, FX, nullptr, nullptr, nullptr, data);
}
void BatchedDefectConstraint::constraints_adjointgradient(const double* X, const double* L, double* FX, double* AGX,
                                                          const SolverIndexingData& data) {
  eval(ASSET_HIP_CON_ADJGRAD, X, L, FX, AGX, nullptr, nullptr, data);
}
void BatchedDefectConstraint::constraints_jacobian(const double* X, double* FX, double* KKTvals, const int* loc,
                                                   const SolverIndexingData& data) {
  eval(ASSET_HIP_JAC, X, nullptr, FX, nullptr, KKTvals, loc, data);
}
void BatchedDefectConstraint::constraints_jacobian_adjointgradient(const double* X, const double* L, double* FX,
                                                                   double* AGX, double* KKTvals, const int* loc,
                                                                   const SolverIndexingData& data) {
  eval(ASSET_HIP_JAC_ADJGRAD, X, L, FX, AGX, KKTvals, loc, data);
}
void BatchedDefectConstraint::constraints_jacobian_adjointgradient_adjointhessian(const double* X, const double* L,
                                                                                  double* FX, double* AGX,
                                                                                  double* KKTvals, const int* loc,
                                                                                  const SolverIndexingData& data) {
  eval(ASSET_HIP_JAC_ADJGRAD_HESS, X, L, FX, AGX, KKTvals, loc, data);
}

}  // namespace asset_hip_host
