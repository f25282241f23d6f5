#include "batched_defect_constraint.h"

#include <memory>

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

std::vector<SolverIndexingData> SolverIndexingData::thread_split(int Threads) const {
  std::vector<SolverIndexingData> split;
  if (Threads <= 0) return split;
  const int cols = num_funcappl, per = cols / Threads, rem = cols % Threads, range = per > 0 ? Threads : rem;
  int start = 0;
  for (int i = 0; i < range; i++) {
    const int cnt = per + (i < rem ? 1 : 0);
    SolverIndexingData d;
    d.input_size = input_size, d.output_size = output_size, d.num_funcappl = cnt;
    d.Vindex.assign(Vindex.begin() + size_t(start) * input_size, Vindex.begin() + size_t(start + cnt) * input_size);
    if (!Cindex.empty()) d.Cindex.assign(Cindex.begin() + size_t(start) * output_size, Cindex.begin() + size_t(start + cnt) * output_size);
    split.push_back(std::move(d));
    start += cnt;
  }
  return split;
}

BatchedDefectConstraint::BatchedDefectConstraint(const std::string& ode, int mode, bool blocked,
                                                 const SolverIndexingData& data, int primal_vars, int equal_cons,
                                                 int device)
    : ode_(ode), mode_(mode), nappl_(data.NumAppl()), n_equal_(equal_cons), blocked_(blocked), n_primal_(primal_vars), device_(device) {
  create(data);
}

BatchedDefectConstraint::BatchedDefectConstraint(const std::string& ode, int mode, bool blocked,
                                                 const SolverIndexingData& data, int primal_vars, int equal_cons,
                                                 const std::vector<int>& devices)
    : ode_(ode), mode_(mode), nappl_(data.NumAppl()), n_equal_(equal_cons), blocked_(blocked), n_primal_(primal_vars),
      device_(devices.empty() ? 0 : devices[0]), devices_(devices) {
  if (devices.empty()) throw std::invalid_argument("BatchedDefectConstraint: no devices");
  create(data);
}

int BatchedDefectConstraint::num_shards() const { return hs_ ? asset_hip_sharded_shards(hs_) : 1; }

void BatchedDefectConstraint::create(const SolverIndexingData& data) {
  if (data.NumAppl() <= 0) throw std::invalid_argument("BatchedDefectConstraint: no function applications");
  asset_hip_defect_desc d;
  std::memset(&d, 0, sizeof d);
  d.mode = mode_;
  d.blocked = blocked_ ? 1 : 0;
  d.ode = ode_.c_str();
  d.nseg = data.NumAppl();
  d.vindex = data.Vindex.data();
  d.cindex = data.Cindex.data();
  d.n_primal = n_primal_;
  d.n_equal = n_equal_;
  d.device = device_;
  int rc;
  if (devices_.empty()) rc = asset_hip_defect_create(&d, &h_);
  else {
    rc = asset_hip_defect_create_sharded(&d, int(devices_.size()), devices_.data(), &hs_);
    if (rc == 0) h_ = asset_hip_sharded_handle(hs_, 0);
  }
  if (rc == ASSET_HIP_EINVAL || rc == ASSET_HIP_ENOODE || rc == ASSET_HIP_ERANGE)
    throw std::invalid_argument(std::string("BatchedDefectConstraint: ") + asset_hip_last_error());
  check(rc, devices_.empty() ? "asset_hip_defect_create" : "asset_hip_defect_create_sharded");
  check(asset_hip_defect_sizes(h_, &ir_, &or_, &nkkt_), "asset_hip_defect_sizes");
  if (asset_hip_defect_kkt_layout(h_, &kstride_, nullptr, nullptr) < 0) check(ASSET_HIP_EINVAL, "asset_hip_defect_kkt_layout");
  lrows_.resize(kstride_), lcols_.resize(kstride_);
  if (asset_hip_defect_kkt_layout(h_, nullptr, lrows_.data(), lcols_.data()) < 0) check(ASSET_HIP_EINVAL, "asset_hip_defect_kkt_layout");
  if (ir_ != data.input_size || or_ != data.output_size) {
    if (hs_) asset_hip_sharded_destroy(hs_); else asset_hip_defect_destroy(h_);
    h_ = nullptr, hs_ = nullptr;
    throw std::invalid_argument("BatchedDefectConstraint: index data rows do not match the defect's IRows/ORows");
  }
  fx_.resize(size_t(nappl_) * or_);
  agx_.resize(size_t(nappl_) * ir_);
  kkt_.resize(size_t(nappl_) * kstride_);
  // the staging arrays live as long as the constraint: page-lock them so the block copies run at PCIe rate
  pinned_ = asset_hip_host_register(fx_.data(), fx_.size() * sizeof(double)) == 0 &&
            asset_hip_host_register(agx_.data(), agx_.size() * sizeof(double)) == 0 &&
            asset_hip_host_register(kkt_.data(), kkt_.size() * sizeof(double)) == 0;
}

BatchedDefectConstraint::~BatchedDefectConstraint() {
  unpin();
  if (hs_) asset_hip_sharded_destroy(hs_);
  else asset_hip_defect_destroy(h_);
}

void BatchedDefectConstraint::unpin() {
  if (pinned_) {
    (void)asset_hip_host_unregister(fx_.data());
    (void)asset_hip_host_unregister(agx_.data());
    (void)asset_hip_host_unregister(kkt_.data());
    pinned_ = false;
  }
}

// New index data for the same function: the re-meshing step of the adaptive mesh loop (ODEPhaseBase.cpp:1443-1542).  The
// device handle keeps its code, lane tables, stream and every buffer that still fits (asset_hip_defect_rebind).
void BatchedDefectConstraint::rebind(const SolverIndexingData& data, int primal_vars, int equal_cons) {
  if (data.NumAppl() <= 0 || data.input_size != ir_ || data.output_size != or_)
    throw std::invalid_argument("BatchedDefectConstraint::rebind: index data does not fit this function");
  if (hs_) {   // the sharded form: another split of the applications -- new shards (the device code stays loaded in the library)
    unpin();
    asset_hip_sharded_destroy(hs_);
    hs_ = nullptr, h_ = nullptr;
    nappl_ = data.NumAppl(), n_primal_ = primal_vars, n_equal_ = equal_cons;
    map_source_ = nullptr;
    map_.clear();
    create(data);
    return;
  }
  const int rc = asset_hip_defect_rebind(h_, data.NumAppl(), data.Vindex.data(), data.Cindex.data(), primal_vars, equal_cons);
  if (rc == ASSET_HIP_EINVAL || rc == ASSET_HIP_ERANGE)
    throw std::invalid_argument(std::string("BatchedDefectConstraint::rebind: ") + asset_hip_last_error());
  check(rc, "asset_hip_defect_rebind");
  unpin();
  nappl_ = data.NumAppl(), n_primal_ = primal_vars, n_equal_ = equal_cons;
  fx_.resize(size_t(nappl_) * or_);
  agx_.resize(size_t(nappl_) * ir_);
  kkt_.resize(size_t(nappl_) * kstride_);
  pinned_ = asset_hip_host_register(fx_.data(), fx_.size() * sizeof(double)) == 0 &&
            asset_hip_host_register(agx_.data(), agx_.size() * sizeof(double)) == 0 &&
            asset_hip_host_register(kkt_.data(), kkt_.size() * sizeof(double)) == 0;
  map_source_ = nullptr;          // (the KKT map was the old tables': gathered and uploaded again at the next evaluation)
  map_.clear();
}

// DeepCopySpecs.h:36-60: an independent function object -- a device handle of its own, created from the same descriptor.
// (Copying the object itself is deleted: ConstraintFunction copies its function by value, ConstraintFunction.h:40-62, and two
// copies must not release one handle twice; an adapter that needs value semantics holds this class -- or the raw handle --
// in a std::shared_ptr)
std::unique_ptr<BatchedDefectConstraint> BatchedDefectConstraint::deep_copy(const SolverIndexingData& data) const {
  auto c = devices_.empty() ? std::make_unique<BatchedDefectConstraint>(ode_, mode_, blocked_, data, n_primal_, n_equal_, device_)
                            : std::make_unique<BatchedDefectConstraint>(ode_, mode_, blocked_, data, n_primal_, n_equal_, devices_);
  if (nvalues_ > 0) c->enable_device_assembly(nvalues_);
  return c;
}

void BatchedDefectConstraint::set_appl_consts(const double* consts, int per_application) {
  if (hs_) {
    for (int i = 0; i < asset_hip_sharded_shards(hs_); i++) {
      int first = 0, count = 0;
      check(asset_hip_sharded_range(hs_, i, &first, &count, nullptr), "asset_hip_sharded_range");
      check(asset_hip_defect_set_appl_consts(asset_hip_sharded_handle(hs_, i), consts + size_t(first) * per_application, per_application),
            "asset_hip_defect_set_appl_consts");
    }
    return;
  }
  check(asset_hip_defect_set_appl_consts(h_, consts, per_application), "asset_hip_defect_set_appl_consts");
}

std::string BatchedDefectConstraint::name() const {
  const char* m = mode_ == ASSET_HIP_TRAPEZOIDAL ? "Trapezoidal" : (mode_ == 2 ? "LGL3" : (mode_ == 3 ? "LGL5" : "LGL7"));
  return std::string("HIP_") + m + "Defects<" + ode_ + ">";
}

void BatchedDefectConstraint::EnableHessianSparsity(bool on) {
  hess_nz_.clear();
  map_source_ = nullptr;
  if (!on || mode_ != ASSET_HIP_TRAPEZOIDAL) return;
  // TrapezoidalDefects::setODE, TrapezoidalDefects.h:75-121: IR = 2 q + p; ones on the two node blocks, on everything that touches
  // a parameter, and on the rows and columns of the two node times
  int xv = 0, uv = 0, pv = 0;
  check(asset_hip_ode_sizes(ode_.c_str(), &xv, &uv, &pv), "asset_hip_ode_sizes");
  const int q = xv + 1 + (blocked_ ? 0 : uv), p = blocked_ ? uv + pv : pv, T = xv;
  if (2 * q + p != ir_) throw std::runtime_error("EnableHessianSparsity: sizes do not add up");
  hess_nz_.assign(size_t(ir_) * ir_, 0);
  auto set = [&](int r, int c) { hess_nz_[size_t(r) + size_t(ir_) * c] = 1; };
  for (int r = 0; r < q; r++)
    for (int c = 0; c < q; c++) { set(r, c); set(q + r, q + c); }
  for (int r = 0; r < p; r++)
    for (int c = 0; c < p; c++) set(2 * q + r, 2 * q + c);
  for (int j = 0; j < 2; j++)
    for (int r = 0; r < q; r++)
      for (int c = 0; c < p; c++) { set(j * q + r, 2 * q + c); set(2 * q + c, j * q + r); }
  for (int k = 0; k < ir_; k++) { set(k, T); set(k, T + q); set(T, k); set(T + q, k); }
}

int BatchedDefectConstraint::numKKTEles(bool dojac, bool dohess) const {   // DenseFunctionBase.h:1070-1088
  int h = 0;
  if (dohess)
    for (int i = 0; i < ir_; i++)
      for (int j = i; j < ir_; j++) h += HessianElemIsNonZero(j, i) ? 1 : 0;
  return h + (dojac ? or_ * ir_ : 0);
}

void BatchedDefectConstraint::getKKTSpace(int* KKTrows, int* KKTcols, int& freeloc, int conoffset, bool dojac,
                                          bool dohess, SolverIndexingData& data) const {
  // (row, col) of an application's slots in the order of the device's blocks (see the header): DenseFunctionBase.h:1097-1129 with
  // its two loops replaced by one walk over the layout
  data.InnerKKTStarts.resize(data.NumAppl());
  for (int V = 0; V < data.NumAppl(); V++) {
    data.InnerKKTStarts[V] = freeloc;
    for (int k = 0; k < kstride_; k++) {
      const int r = lrows_[k], i = lcols_[k];
      if (r < 0) continue;                                       // padding
      if (r < ir_) {
        if (!dohess || !HessianElemIsNonZero(r, i)) continue;
        KKTrows[freeloc] = data.VLoc(r, V);
      } else {
        if (!dojac) continue;
        KKTrows[freeloc] = data.CLoc(r - ir_, V) + conoffset;
      }
      KKTcols[freeloc] = data.VLoc(i, V);
      freeloc++;
    }
  }
}

void BatchedDefectConstraint::scatter_kkt(const double* blocks, bool dohess, double* KKTvals, const int* lpt,
                                          const SolverIndexingData& data, bool dojac) const {
  for (int V = 0; V < data.NumAppl(); V++) {
    const double* blk = blocks + size_t(V) * kstride_;
    int freeloc = data.InnerKKTStarts[V];
    for (int k = 0; k < kstride_; k++) {
      const int r = lrows_[k];
      if (r < 0) continue;
      if (r < ir_) {
        if (!HessianElemIsNonZero(r, lcols_[k])) continue;         // claims no slot (AddHessianElem)
        if (dohess) KKTvals[lpt[freeloc]] += blk[k];               // KKTFillJac: the slot exists in the space and is stepped over
        freeloc++;
      } else if (dojac) {                                          // KKTFillHess: an objective's space has no Jacobian slots
        KKTvals[lpt[freeloc++]] += blk[k];
      }
    }
  }
}

void BatchedDefectConstraint::eval(int what, const double* X, const double* L, double* FX, double* AGX,
                                   double* KKTvals, const int* KKTLocations, const SolverIndexingData& data,
                                   bool hess_only) {
  if (data.NumAppl() != nappl_) throw std::invalid_argument("index data does not belong to this constraint");
  const bool want_agx = (what == ASSET_HIP_CON_ADJGRAD || what == ASSET_HIP_JAC_ADJGRAD || what == ASSET_HIP_JAC_ADJGRAD_HESS);
  const bool want_kkt = what >= ASSET_HIP_JAC;
  const bool assembled = want_kkt && nvalues_ > 0;
  if (assembled) {
    ensure_kkt_map(KKTLocations, data, hess_only);
    if (hs_) check(asset_hip_sharded_eval_assembled(hs_, what, X, L, fx_.data(), want_agx ? agx_.data() : nullptr, KKTvals),
                   "asset_hip_sharded_eval_assembled");
    else check(asset_hip_defect_eval_assembled(h_, what, X, L, fx_.data(), want_agx ? agx_.data() : nullptr, KKTvals),
               "asset_hip_defect_eval_assembled");
  } else {
    // the Jacobian kinds scatter with KKTFillJac below, which never reads the Hessian slots: do not have them written
    const int keep = (what == ASSET_HIP_JAC || what == ASSET_HIP_JAC_ADJGRAD) ? ASSET_HIP_KEEP_HESSIAN_SLOTS : 0;
    if (hs_) check(asset_hip_sharded_eval(hs_, what | keep, X, L, fx_.data(), want_agx ? agx_.data() : nullptr,
                                          want_kkt ? kkt_.data() : nullptr), "asset_hip_sharded_eval");
    else check(asset_hip_defect_eval(h_, what | keep, X, L, fx_.data(), want_agx ? agx_.data() : nullptr,
                                     want_kkt ? kkt_.data() : nullptr), "asset_hip_defect_eval");
  }
  for (int V = 0; V < nappl_; V++) {  // callee overwrites its FX / AGX slots (fx.setZero(); compute)
    std::memcpy(FX + data.InnerConstraintStarts[V], fx_.data() + size_t(V) * or_, sizeof(double) * or_);
    if (want_agx) std::memcpy(AGX + data.InnerGradientStarts[V], agx_.data() + size_t(V) * ir_, sizeof(double) * ir_);
  }
  if (want_kkt && !assembled)
    scatter_kkt(kkt_.data(), what == ASSET_HIP_JAC_ADJGRAD_HESS, KKTvals, KKTLocations, data, !hess_only);
}

// ---- objective: the function's single output weighted by ObjScale (DenseScalarFunctionBase.h:14-80)
void BatchedDefectConstraint::objective(double ObjScale, const double* X, double& Val, const SolverIndexingData& data) {
  if (or_ != 1 || n_equal_ != 1)
    throw std::invalid_argument("objective: the function must have one output and be constructed with equal_cons = 1");
  std::vector<double> fx(size_t(nappl_), 0.0);
  SolverIndexingData d = data;                       // value slots: one per application, in order
  d.InnerConstraintStarts.resize(nappl_);
  for (int V = 0; V < nappl_; V++) d.InnerConstraintStarts[V] = V;
  eval(ASSET_HIP_CON, X, nullptr, fx.data(), nullptr, nullptr, nullptr, d);
  for (int V = 0; V < nappl_; V++) Val += fx[V] * ObjScale;
}
void BatchedDefectConstraint::objective_gradient(double ObjScale, const double* X, double& Val, double* GX,
                                                 const SolverIndexingData& data) {
  if (or_ != 1 || n_equal_ != 1)
    throw std::invalid_argument("objective: the function must have one output and be constructed with equal_cons = 1");
  std::vector<double> fx(size_t(nappl_), 0.0);
  SolverIndexingData d = data;
  d.InnerConstraintStarts.resize(nappl_);
  for (int V = 0; V < nappl_; V++) d.InnerConstraintStarts[V] = V;
  eval(ASSET_HIP_CON_ADJGRAD, X, &ObjScale, fx.data(), GX, nullptr, nullptr, d);    // J^T lam with lam = ObjScale
  for (int V = 0; V < nappl_; V++) Val += fx[V] * ObjScale;
}
void BatchedDefectConstraint::objective_gradient_hessian(double ObjScale, const double* X, double& Val, double* GX,
                                                         double* KKTvals, const int* KKTLocations,
                                                         const SolverIndexingData& data) {
  if (or_ != 1 || n_equal_ != 1)
    throw std::invalid_argument("objective: the function must have one output and be constructed with equal_cons = 1");
  std::vector<double> fx(size_t(nappl_), 0.0);
  SolverIndexingData d = data;
  d.InnerConstraintStarts.resize(nappl_);
  for (int V = 0; V < nappl_; V++) d.InnerConstraintStarts[V] = V;
  eval(ASSET_HIP_JAC_ADJGRAD_HESS, X, &ObjScale, fx.data(), GX, KKTvals, KKTLocations, d, /*hess_only=*/true);
  for (int V = 0; V < nappl_; V++) Val += fx[V] * ObjScale;
}

void BatchedDefectConstraint::enable_device_assembly(long long nvalues) {
  if (nvalues <= 0) throw std::invalid_argument("enable_device_assembly: the value array length must be positive");
  nvalues_ = nvalues;
  map_source_ = nullptr;
}

void BatchedDefectConstraint::ensure_kkt_map(const int* lpt, const SolverIndexingData& data, bool hess_only) {
  if (!lpt) throw std::invalid_argument("KKTLocations is null");
  if (int(data.InnerKKTStarts.size()) != nappl_) throw std::invalid_argument("InnerKKTStarts not filled: call getKKTSpace first");
  // The library takes the map in the CANONICAL numbering of a block's entries (the reference's order `for i: {H(j>=i,i); J(:,i)}`,
  // include/asset_hip.h); the caller's KKT space was claimed by getKKTSpace above, i.e. in the order of the handle's layout, without
  // the entries that claim no slot (a Hessian mask -- EnableHessianSparsity; the Jacobian slots of an objective, hess_only).
  // space_of[k] = position of block slot k among the claimed slots of an application, or -1.
  std::vector<int> space_of(kstride_, -1);
  {
    int pos = 0;
    for (int k = 0; k < kstride_; k++) {
      const int r = lrows_[k];
      if (r < 0) continue;
      if (r < ir_) { if (HessianElemIsNonZero(r, lcols_[k])) space_of[k] = pos++; }
      else if (!hess_only) space_of[k] = pos++;
    }
  }
  // canonical entry -> block slot
  std::vector<int> slot_h(size_t(ir_) * ir_, -1), slot_j(size_t(or_) * ir_, -1);
  for (int k = 0; k < kstride_; k++) {
    const int r = lrows_[k], i = lcols_[k];
    if (r < 0) continue;
    if (r < ir_) slot_h[size_t(r) + size_t(ir_) * i] = k; else slot_j[size_t(r - ir_) + size_t(or_) * i] = k;
  }
  auto space_slot = [&](int V, int k) -> int { return space_of[k] < 0 ? -1 : lpt[data.InnerKKTStarts[V] + space_of[k]]; };
  const int k_first = slot_h[0], k_last = slot_h[size_t(ir_ - 1) + size_t(ir_) * (ir_ - 1)];
  bool fresh = (lpt == map_source_) && map_.size() == size_t(nappl_) * nkkt_ && map_hess_only_ == hess_only;
  if (fresh) {  // same array: make sure the solver did not re-fill it in place (cheap sample)
    for (int V = 0; V < nappl_ && fresh; V += nappl_ / 16 + 1)
      fresh = (map_[size_t(V) * nkkt_] == space_slot(V, k_first)) &&
              (map_[size_t(V) * nkkt_ + nkkt_ - or_ - 1] == space_slot(V, k_last));
  }
  if (fresh) return;
  map_.resize(size_t(nappl_) * nkkt_);
  for (int V = 0; V < nappl_; V++) {
    int* m = map_.data() + size_t(V) * nkkt_;
    int c = 0;
    for (int i = 0; i < ir_; i++) {
      for (int j = i; j < ir_; j++) m[c++] = space_slot(V, slot_h[size_t(j) + size_t(ir_) * i]);
      for (int j = 0; j < or_; j++) m[c++] = space_slot(V, slot_j[size_t(j) + size_t(or_) * i]);
    }
  }
  if (hs_) check(asset_hip_sharded_set_kkt_map(hs_, map_.data(), nvalues_), "asset_hip_sharded_set_kkt_map");
  else check(asset_hip_defect_set_kkt_map(h_, map_.data(), nvalues_, 0), "asset_hip_defect_set_kkt_map");
  map_source_ = lpt;
  map_hess_only_ = hess_only;
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
