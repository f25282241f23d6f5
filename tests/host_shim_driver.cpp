// Test driver for the C++ host shim (asset_asrl_amd/host/batched_defect_constraint.h): builds the index data the way
// NonLinearProgram::getMATSpace/getRHSSpace do for a single-thread NLP, runs one evaluation kind through the shim's
// Concept-style methods and returns the scattered results.  Compiled by tests/test_gpu_host_shim.py.
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../asset_asrl_amd/host/batched_defect_constraint.h"

using namespace asset_hip_host;

// 0: one handle on device 0; n > 0: the constraint as n in-process shards, all on device 0 (one GPU on the test box)
static int g_hess_sparsity = 0;   // EnableHessianSparsity of the Trapezoidal defects (TrapezoidalDefects.h:39)
extern "C" void shim_set_hessian_sparsity(int on) { g_hess_sparsity = on; }
static int g_shards = 0;
extern "C" void shim_set_shards(int n) { g_shards = n; }
static std::unique_ptr<BatchedDefectConstraint> make_con(const char* ode, int mode, bool blocked, const SolverIndexingData& d, int primal, int equal) {
  if (g_shards > 0) return std::make_unique<BatchedDefectConstraint>(ode, mode, blocked, d, primal, equal, std::vector<int>(size_t(g_shards), 0));
  return std::make_unique<BatchedDefectConstraint>(ode, mode, blocked, d, primal, equal, 0);
}

// The KKT space the constraint claims -- (row, col) of every slot in ITS order (the order of the device's blocks); the caller turns
// them into KKTLocations the way NonLinearProgram::analyzeSparsity does (NonLinearProgram.cpp:282-330), whatever the order.
extern "C" int shim_space(const char* ode, int mode, int blocked, int ir, int orr, int nappl, const int* vindex, const int* cindex,
                          int primal, int equal, int* kkt_rows_out, int* kkt_cols_out, char* err, int errcap) {
  try {
    SolverIndexingData data;
    data.input_size = ir, data.output_size = orr, data.num_funcappl = nappl;
    data.Vindex.assign(vindex, vindex + size_t(ir) * nappl);
    data.Cindex.assign(cindex, cindex + size_t(orr) * nappl);
    auto conp = make_con(ode, mode, blocked != 0, data, primal, equal);
    conp->EnableHessianSparsity(g_hess_sparsity != 0);
    int kfree = 0;
    conp->getKKTSpace(kkt_rows_out, kkt_cols_out, kfree, primal, true, true, data);
    return kfree;
  } catch (const std::exception& e) {
    std::strncpy(err, e.what(), errcap - 1);
    err[errcap - 1] = 0;
    return -1;
  }
}

extern "C" int shim_run(const char* ode, int mode, int blocked, int ir, int orr, int nappl, const int* vindex,
                        const int* cindex, int primal, int equal, int what, const double* X, const double* L,
                        const int* kkt_locations, int* kkt_rows_out, int* kkt_cols_out, double* kkt_vals,
                        double* FXE, double* AGX, char* err, int errcap, long long assembly_nvalues) {
  try {
    SolverIndexingData data;
    data.input_size = ir, data.output_size = orr, data.num_funcappl = nappl;
    data.Vindex.assign(vindex, vindex + size_t(ir) * nappl);
    data.Cindex.assign(cindex, cindex + size_t(orr) * nappl);
    auto conp = make_con(ode, mode, blocked != 0, data, primal, equal);
    BatchedDefectConstraint& con = *conp;
    con.EnableHessianSparsity(g_hess_sparsity != 0);
    if (g_shards > 0) {   // the split is the reference's (IndexingData.h:117-146)
      auto parts = data.thread_split(g_shards);
      int tot = 0;
      for (auto& q : parts) tot += q.NumAppl();
      if (int(parts.size()) != con.num_shards() || tot != nappl) throw std::runtime_error("thread_split / shard count mismatch");
    }
    std::vector<int> gxrows(size_t(ir) * nappl), fxrows(size_t(orr) * nappl);
    int gfree = 0, cfree = 0, kfree = 0;
    data.getGradientSpace(gxrows.data(), gfree);
    data.getConstraintSpace(fxrows.data(), cfree);
    con.getKKTSpace(kkt_rows_out, kkt_cols_out, kfree, primal, true, true, data);
    if (kfree != con.numKKTEles(true, true) * nappl) throw std::runtime_error("KKT space count mismatch");
    if (assembly_nvalues > 0) con.enable_device_assembly(assembly_nvalues);   // KKT entries added on the GPU
    std::vector<double> fxc(size_t(orr) * nappl, 0.0), agxc(size_t(ir) * nappl, 0.0);
    switch (what) {
      case ASSET_HIP_CON: con.constraints(X, fxc.data(), data); break;
      case ASSET_HIP_CON_ADJGRAD: con.constraints_adjointgradient(X, L, fxc.data(), agxc.data(), data); break;
      case ASSET_HIP_JAC: con.constraints_jacobian(X, fxc.data(), kkt_vals, kkt_locations, data); break;
      case ASSET_HIP_JAC_ADJGRAD:
        con.constraints_jacobian_adjointgradient(X, L, fxc.data(), agxc.data(), kkt_vals, kkt_locations, data);
        break;
      default:
        con.constraints_jacobian_adjointgradient_adjointhessian(X, L, fxc.data(), agxc.data(), kkt_vals,
                                                                kkt_locations, data);
    }
    // fillRHS (NonLinearProgram.h:401-407)
    for (size_t i = 0; i < fxc.size(); i++) FXE[fxrows[i]] += fxc[i];
    for (size_t i = 0; i < agxc.size(); i++) AGX[gxrows[i]] += agxc[i];
    return 0;
  } catch (const std::exception& e) {
    std::strncpy(err, e.what(), errcap - 1);
    err[errcap - 1] = 0;
    return 1;
  }
}

// The same evaluation through a constraint that was created for another mesh (the first nappl0 applications of a smaller
// program), re-bound to this one (BatchedDefectConstraint::rebind, the adaptive mesh loop's step), and then deep-copied
// (DeepCopySpecs.h:36-60): the COPY evaluates, with the original destroyed first -- it must own a handle of its own.
extern "C" int shim_rebind_run(const char* ode, int mode, int blocked, int ir, int orr, int nappl, const int* vindex,
                               const int* cindex, int primal, int equal, int nappl0, const double* X, const double* L,
                               const int* kkt_locations, double* kkt_vals, double* FXE, double* AGX, char* err, int errcap,
                               long long assembly_nvalues) {
  try {
    SolverIndexingData small, data;
    small.input_size = ir, small.output_size = orr, small.num_funcappl = nappl0;
    small.Vindex.assign(vindex, vindex + size_t(ir) * nappl0);
    small.Cindex.assign(cindex, cindex + size_t(orr) * nappl0);
    data.input_size = ir, data.output_size = orr, data.num_funcappl = nappl;
    data.Vindex.assign(vindex, vindex + size_t(ir) * nappl);
    data.Cindex.assign(cindex, cindex + size_t(orr) * nappl);
    auto con = make_con(ode, mode, blocked != 0, small, primal, equal);
    if (assembly_nvalues > 0) con->enable_device_assembly(assembly_nvalues);
    {
      std::vector<int> fr0(size_t(orr) * nappl0);
      int c0 = 0;
      small.getConstraintSpace(fr0.data(), c0);
      std::vector<double> f0(size_t(orr) * nappl0);
      con->constraints(X, f0.data(), small);                      // (the handle has been used on the old mesh)
    }
    con->rebind(data, primal, equal);
    std::vector<int> gxrows(size_t(ir) * nappl), fxrows(size_t(orr) * nappl), krows(size_t(con->numKKTEles(true, true)) * nappl),
        kcols(krows.size());
    int gfree = 0, cfree = 0, kfree = 0;
    data.getGradientSpace(gxrows.data(), gfree);
    data.getConstraintSpace(fxrows.data(), cfree);
    con->getKKTSpace(krows.data(), kcols.data(), kfree, primal, true, true, data);
    std::unique_ptr<BatchedDefectConstraint> copy = con->deep_copy(data);
    con.reset();
    std::vector<double> fxc(size_t(orr) * nappl, 0.0), agxc(size_t(ir) * nappl, 0.0);
    copy->constraints_jacobian_adjointgradient_adjointhessian(X, L, fxc.data(), agxc.data(), kkt_vals, kkt_locations, data);
    for (size_t i = 0; i < fxc.size(); i++) FXE[fxrows[i]] += fxc[i];
    for (size_t i = 0; i < agxc.size(); i++) AGX[gxrows[i]] += agxc[i];
    return 0;
  } catch (const std::exception& e) {
    std::strncpy(err, e.what(), errcap - 1);
    err[errcap - 1] = 0;
    return 1;
  }
}

// ---- KktAssembly (asset_asrl_amd/host/kkt_assembly.h): structure + the four evaluation entry points -------------
#include "../asset_asrl_amd/host/kkt_assembly.h"

// One constraint (ode, mode): returns nnz; outer[dim+1], inner[cap], locs[num_user + dim]; what: 0 OCC, 1 RHS, 2 SOE, 4 KKT
extern "C" int assembly_run(const char* ode, int mode, int blocked, int ir, int orr, int nappl, const int* vindex,
                            const int* cindex, int primal, int equal, int what, const double* X, const double* L,
                            int* outer, int* inner, int inner_cap, int* locs, double* kkt_vals, double* FXE, double* AGX,
                            char* err, int errcap) {
  try {
    SolverIndexingData data;
    data.input_size = ir, data.output_size = orr, data.num_funcappl = nappl;
    data.Vindex.assign(vindex, vindex + size_t(ir) * nappl);
    data.Cindex.assign(cindex, cindex + size_t(orr) * nappl);
    BatchedDefectConstraint con(ode, mode, blocked != 0, data, primal, equal, 0);
    KktAssembly nlp(primal, equal);
    nlp.add_equality(con, data);
    nlp.analyze();
    if (nlp.nnz() > inner_cap) throw std::runtime_error("inner capacity too small");
    std::memcpy(outer, nlp.outer().data(), sizeof(int) * (nlp.kkt_dim() + 1));
    std::memcpy(inner, nlp.inner().data(), sizeof(int) * nlp.nnz());
    std::memcpy(locs, nlp.kkt_locations().data(), sizeof(int) * nlp.kkt_locations().size());
    switch (what) {
      case 0: nlp.evalOCC(X, FXE); break;
      case 1: nlp.evalRHS(X, L, FXE, AGX); break;
      case 2: nlp.evalSOE(X, FXE, kkt_vals); break;
      case 3: nlp.evalAUG(X, L, FXE, AGX, kkt_vals); break;
      default: nlp.evalKKT(X, L, FXE, AGX, kkt_vals);
    }
    return nlp.nnz();
  } catch (const std::exception& e) {
    std::strncpy(err, e.what(), errcap - 1);
    err[errcap - 1] = 0;
    return -1;
  }
}

// ---- KktAssembly with objectives, equalities and inequalities (slacks): the whole KKT layout ------------------------
struct FnDesc {
  int kind;            // 0 objective, 1 equality, 2 inequality
  const char* name;    // device-side name (library ODE or run-time compiled function)
  int mode, blocked, ir, orr, nappl;
  const int* vindex;   // [ir x nappl] column-major
  const int* cindex;   // [orr x nappl]; objectives: ignored (all applications read multiplier 0 = ObjScale)
  const double* consts;   // per-application constants [nappl][nconst] of a function built with vf.ApplConst, or null
  int nconst;
};

// A program that stays alive between evaluations (a solver loop calls it once per iteration).
struct FullNlp {
  std::vector<SolverIndexingData> datas;
  std::vector<std::unique_ptr<BatchedDefectConstraint>> cons;
  std::unique_ptr<KktAssembly> nlp;
};

static void set_err(char* err, int errcap, const char* what) {
  if (err && errcap > 0) { std::strncpy(err, what, errcap - 1); err[errcap - 1] = 0; }
}

extern "C" void* fullnlp_create(const FnDesc* fns, int nfn, int primal, int equal, int inequal, char* err, int errcap) {
  try {
    std::unique_ptr<FullNlp> p(new FullNlp);
    p->datas.resize(nfn);
    for (int k = 0; k < nfn; k++) {
      const FnDesc& f = fns[k];
      SolverIndexingData& d = p->datas[k];
      d.input_size = f.ir, d.output_size = f.orr, d.num_funcappl = f.nappl;
      d.Vindex.assign(f.vindex, f.vindex + size_t(f.ir) * f.nappl);
      if (f.kind == 0) d.Cindex.assign(size_t(f.orr) * f.nappl, 0);
      else d.Cindex.assign(f.cindex, f.cindex + size_t(f.orr) * f.nappl);
      const int ncon = f.kind == 0 ? 1 : (f.kind == 1 ? equal : inequal);
      p->cons.emplace_back(new BatchedDefectConstraint(f.name, f.mode, f.blocked != 0, d, primal, ncon, 0));
      if (f.consts && f.nconst > 0) p->cons.back()->set_appl_consts(f.consts, f.nconst);
    }
    p->nlp.reset(new KktAssembly(primal, equal, inequal));
    for (int k = 0; k < nfn; k++) {
      if (fns[k].kind == 0) p->nlp->add_objective(*p->cons[k], p->datas[k]);
      else if (fns[k].kind == 1) p->nlp->add_equality(*p->cons[k], p->datas[k]);
      else p->nlp->add_inequality(*p->cons[k], p->datas[k]);
    }
    p->nlp->analyze();
    return p.release();
  } catch (const std::exception& e) {
    set_err(err, errcap, e.what());
    return nullptr;
  }
}

extern "C" void fullnlp_destroy(void* h) { delete static_cast<FullNlp*>(h); }

// sizes: [kkt_dim, nnz, number of KKT locations (user slots + solver slots), number of solver slots]
extern "C" void fullnlp_sizes(void* h, int* out) {
  KktAssembly& nlp = *static_cast<FullNlp*>(h)->nlp;
  out[0] = nlp.kkt_dim(), out[1] = nlp.nnz(), out[2] = int(nlp.kkt_locations().size()), out[3] = nlp.num_solver_kkt();
}

extern "C" void fullnlp_structure(void* h, int* outer, int* inner, int* locs) {
  KktAssembly& nlp = *static_cast<FullNlp*>(h)->nlp;
  std::memcpy(outer, nlp.outer().data(), sizeof(int) * (nlp.kkt_dim() + 1));
  std::memcpy(inner, nlp.inner().data(), sizeof(int) * nlp.nnz());
  std::memcpy(locs, nlp.kkt_locations().data(), sizeof(int) * nlp.kkt_locations().size());
}

extern "C" void fullnlp_set_solver_coeffs(void* h, const double* c) {
  KktAssembly& nlp = *static_cast<FullNlp*>(h)->nlp;
  std::memcpy(nlp.solver_coeffs().data(), c, sizeof(double) * nlp.num_solver_kkt());
}

// level: 0 evalOCC, 1 evalRHS, 2 evalSOE, 3 evalAUG, 4 evalKKT.  Returns 0 (or -1); *val = objective value.
extern "C" int fullnlp_eval(void* h, int level, double ObjScale, const double* X, const double* LE, const double* LI,
                            double* val, double* PGX, double* AGX, double* FXE, double* FXI, double* kkt_vals, char* err,
                            int errcap) {
  try {
    KktAssembly& nlp = *static_cast<FullNlp*>(h)->nlp;
    switch (level) {
      case 0: *val = nlp.evalOCC(ObjScale, X, FXE, FXI); break;
      case 1: *val = nlp.evalRHS(ObjScale, X, LE, LI, PGX, AGX, FXE, FXI); break;
      case 2: nlp.evalSOE(X, FXE, FXI, kkt_vals); *val = 0.0; break;
      case 3: *val = nlp.evalAUG(ObjScale, X, LE, LI, PGX, AGX, FXE, FXI, kkt_vals); break;
      default: *val = nlp.evalKKT(ObjScale, X, LE, LI, PGX, AGX, FXE, FXI, kkt_vals);
    }
    return 0;
  } catch (const std::exception& e) {
    set_err(err, errcap, e.what());
    return -1;
  }
}

// One-shot form: create, copy the structure out, evaluate once.  Returns nnz (or -1).
extern "C" int fullnlp_run(const FnDesc* fns, int nfn, int primal, int equal, int inequal, int level, double ObjScale,
                           const double* X, const double* LE, const double* LI, const double* solver_coeffs,
                           int* outer, int* inner, int inner_cap, int* locs, int locs_cap, double* val, double* PGX,
                           double* AGX, double* FXE, double* FXI, double* kkt_vals, char* err, int errcap) {
  void* h = fullnlp_create(fns, nfn, primal, equal, inequal, err, errcap);
  if (!h) return -1;
  int sz[4];
  fullnlp_sizes(h, sz);
  int rc = -1;
  if (sz[1] > inner_cap || sz[2] > locs_cap) set_err(err, errcap, "capacity too small");
  else {
    fullnlp_structure(h, outer, inner, locs);
    if (solver_coeffs) fullnlp_set_solver_coeffs(h, solver_coeffs);
    if (fullnlp_eval(h, level, ObjScale, X, LE, LI, val, PGX, AGX, FXE, FXI, kkt_vals, err, errcap) == 0) rc = sz[1];
  }
  fullnlp_destroy(h);
  return rc;
}
