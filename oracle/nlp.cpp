// ORACLE (test infrastructure, never shipped or imported by the product path).
//
// Restatement of everything between the per-segment defect function and the solver's KKT/RHS buffers:
//   phase variable layout + defect Vindex/Cindex   /root/reference/src/OptimalControl/PhaseIndexer.h:63-99,168-197
//                                                  /root/reference/src/OptimalControl/PhaseIndexer.cpp:132-189,361-391
//   per-application block layout                   /root/reference/src/VectorFunctions/DenseFunctionBase.h:1070-1129
//   batched solver-interface loops                 /root/reference/src/VectorFunctions/ComputableBase.h:246-378
//                                                  DenseFunctionBase.h:1145-1391
//   KKT scatter with per-column locks              DenseFunctionBase.h:1413-1523
//   ByApplication thread split                     /root/reference/src/VectorFunctions/IndexingData.h:96-146
//   NLP space / sparsity / eval entry points       /root/reference/src/Solvers/NonLinearProgram.cpp:25-344,347-683
//                                                  /root/reference/src/Solvers/NonLinearProgram.h:264-290,379-407
// The NLP here holds exactly one equality constraint (the phase's defect); objectives, inequalities
// and slacks are absent, so KKTdim = PrimalVars + EqualCons.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "batch4.h"
#include "oracle.h"

namespace {

struct IndexData {  // SolverIndexingData
  int ir = 0, orr = 0, nappl = 0;
  std::vector<int> vindex, cindex;  // col-major [ir x nappl], [orr x nappl]
  std::vector<int> con_starts, grad_starts, kkt_starts;
  int VLoc(int i, int V) const { return vindex[(size_t)V * ir + i]; }
  int CLoc(int j, int V) const { return cindex[(size_t)V * orr + j]; }
};

}  // namespace

// The reference keeps its evaluation threads alive between evaluations (a ctpl pool owned by the NonLinearProgram,
// NonLinearProgram.h:38,126-130; jobs pushed per evaluation, NonLinearProgram.cpp:519-526): a persistent pool of
// T - 1 workers, the caller runs the last slice itself.  Workers spin briefly on the job counter before they block, so a
// back-to-back stream of evaluations (bench.py's cpu_baseline leg) pays no wake-up latency.
class WorkerPool {
 public:
  explicit WorkerPool(int workers) : done_(0) {
    for (int t = 0; t < workers; t++) th_.emplace_back([this, t] { loop(t); });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; gen_.fetch_add(1); }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int size() const { return int(th_.size()); }
  // runs job(t) for t = 0..size()-1 on the workers; returns at once (wait() joins)
  void post(std::function<void(int)> job) {
    { std::lock_guard<std::mutex> g(m_); job_ = std::move(job); done_.store(0); gen_.fetch_add(1); }
    cv_.notify_all();
  }
  void wait() {
    for (int spin = 0; done_.load(std::memory_order_acquire) < size(); spin++)
      if (spin > 2000) std::this_thread::yield();
  }

 private:
  void loop(int t) {
    unsigned seen = 0;
    for (;;) {
      for (int spin = 0; gen_.load(std::memory_order_acquire) == seen && spin < 20000; spin++) {}
      if (gen_.load(std::memory_order_acquire) == seen) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_.load() != seen; });
      }
      std::function<void(int)> job;
      { std::lock_guard<std::mutex> g(m_); if (stop_) return; seen = gen_.load(); job = job_; }
      job(t);
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void(int)> job_;
  std::atomic<unsigned> gen_{0};
  std::atomic<int> done_;
  bool stop_ = false;
};

struct oracle_nlp {
  std::unique_ptr<WorkerPool> pool;          // created at the first multi-threaded evaluation
  oracle_ode ode;
  oracle_ode4 ode4 = {nullptr, nullptr};   // four-wide twin of `ode` (batch4.h), when the registry has one
  bool batch4 = false;                      // oracle_nlp_set_batch4: LGL evalKKT processes four applications at a time
  int mode, blocked;
  int ir, orr, nkkt_per_appl;
  // HessianElemIsNonZero (TrapezoidalDefects.h:124-130 with EnableHessianSparsity; all ones otherwise): [row + IR * col], and the
  // number of claimed Hessian slots of block column i (rows j >= i)
  std::vector<char> hess_nz;
  std::vector<int> hess_count;
  int primal, equal, threads;
  int kktdim;
  std::vector<IndexData> thr;  // one slice per thread (thread_split)
  int num_user_kkt = 0, num_solver_kkt = 0, num_kkt = 0;
  std::vector<int> kkt_rows, kkt_cols, kkt_thr, kkt_locs;
  std::vector<int> clashes;  // per KKT column: -1 or mutex id
  std::vector<std::mutex> locks;
  // RHS coefficient buffers (AGX then ECon) and their target rows
  std::vector<double> agx_coeffs, econ_coeffs, solver_coeffs;
  std::vector<int> agx_rows, econ_rows;
  // CSR upper-triangular structure
  std::vector<int> outer, inner;
};

namespace {

int num_kkt_eles(const oracle_nlp* n, int ir, int orr, bool dojac, bool dohess) {  // DenseFunctionBase.h:1070-1088
  int e = 0;
  for (int i = 0; i < ir; i++) {
    if (dohess) e += n->hess_count[i];
    if (dojac) e += orr;
  }
  return e;
}

// TrapezoidalDefects::setODE (TrapezoidalDefects.h:75-121): the structural non-zeros of the defect's adjoint Hessian -- the two
// node blocks, everything that touches a parameter, the rows and columns of the two node times; i.e. every entry but the
// cross-node block without its time rows / columns
void trapezoidal_hessian_mask(oracle_nlp* n, int q, int p, int T) {
  const int IR = n->ir;
  std::fill(n->hess_nz.begin(), n->hess_nz.end(), 0);
  auto set = [&](int r, int c) { n->hess_nz[r + (size_t)IR * c] = 1; };
  for (int r = 0; r < q; r++)
    for (int c = 0; c < q; c++) { set(r, c); set(q + r, q + c); }
  for (int r = 0; r < p; r++)
    for (int c = 0; c < p; c++) set(2 * q + r, 2 * q + c);
  for (int j = 0; j < 2; j++)
    for (int r = 0; r < q; r++)
      for (int c = 0; c < p; c++) { set(j * q + r, 2 * q + c); set(2 * q + c, j * q + r); }
  for (int k = 0; k < IR; k++) { set(k, T); set(k, T + q); set(T, k); set(T + q, k); }
}

// DenseFunctionBase::getKKTSpace, dojac = dohess = true
void get_kkt_space(oracle_nlp* n, IndexData& d, int& freeloc, int conoffset) {
  d.kkt_starts.resize(d.nappl);
  for (int V = 0; V < d.nappl; V++) {
    d.kkt_starts[V] = freeloc;
    for (int i = 0; i < d.ir; i++) {
      for (int j = i; j < d.ir; j++) {
        if (!n->hess_nz[j + (size_t)d.ir * i]) continue;
        n->kkt_rows[freeloc] = d.VLoc(j, V);
        n->kkt_cols[freeloc] = d.VLoc(i, V);
        freeloc++;
      }
      for (int j = 0; j < d.orr; j++) {
        n->kkt_rows[freeloc] = d.CLoc(j, V) + conoffset;
        n->kkt_cols[freeloc] = d.VLoc(i, V);
        freeloc++;
      }
    }
  }
}

void thread_split(const IndexData& all, int threads, std::vector<IndexData>& out) {  // IndexingData.h:117-146
  const int cols = all.nappl, per = cols / threads, rem = cols % threads;
  const int range = per > 0 ? threads : rem;
  int start = 0;
  out.clear();
  for (int i = 0; i < range; i++) {
    const int cnt = per + (i < rem ? 1 : 0);
    IndexData d;
    d.ir = all.ir, d.orr = all.orr, d.nappl = cnt;
    d.vindex.assign(all.vindex.begin() + (size_t)start * all.ir, all.vindex.begin() + (size_t)(start + cnt) * all.ir);
    d.cindex.assign(all.cindex.begin() + (size_t)start * all.orr,
                    all.cindex.begin() + (size_t)(start + cnt) * all.orr);
    out.push_back(std::move(d));
    start += cnt;
  }
}

void analyze_sparsity(oracle_nlp* n) {  // NonLinearProgram.cpp:267-344
  const size_t ne = (size_t)n->num_kkt;
  std::vector<std::pair<int, int>> trip(ne);
  for (size_t i = 0; i < ne; i++) {
    int row = n->kkt_rows[i], col = n->kkt_cols[i];
    if (col <= row) {
      trip[i] = {col, row};  // lower-triangular entry stored transposed: CSR row = col
    } else {
      n->kkt_rows[i] = col;
      n->kkt_cols[i] = row;
      trip[i] = {row, col};
    }
  }
  std::vector<std::pair<int, int>> uniq(trip);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  n->outer.assign(n->kktdim + 1, 0);
  n->inner.resize(uniq.size());
  for (size_t k = 0; k < uniq.size(); k++) {
    n->outer[uniq[k].first + 1]++;
    n->inner[k] = uniq[k].second;
  }
  for (int r = 0; r < n->kktdim; r++) n->outer[r + 1] += n->outer[r];
  n->kkt_locs.assign(ne, -1);
  for (size_t i = 0; i < ne; i++) {
    const int r = trip[i].first, c = trip[i].second;
    const int* b = n->inner.data() + n->outer[r];
    const int* e = n->inner.data() + n->outer[r + 1];
    const int* it = std::lower_bound(b, e, c);
    n->kkt_locs[i] = (int)(it - n->inner.data());
  }
}

inline void gather(const IndexData& d, int V, const double* X, const double* L, double* x, double* l) {
  for (int i = 0; i < d.ir; i++) x[i] = X[d.VLoc(i, V)];
  if (L)
    for (int j = 0; j < d.orr; j++) l[j] = L[d.CLoc(j, V)];
}

// One thread-slice of one solver-interface method.  If kkt_blocks != null the dense per-application
// block is written there (block order of getKKTSpace) instead of being scattered.
void eval_slice(oracle_nlp* n, const IndexData& d, int what, const double* X, const double* L, double* fx_out,
                double* agx_out, double* kktvals, double* kkt_blocks, bool blocks, size_t appl_base) {
  const int IR = d.ir, OR = d.orr;
  std::vector<double> x(IR), l(OR), fx(OR), jx((size_t)OR * IR), agx(IR), hx((size_t)IR * IR);
  // what happens to one application's results: value / adjoint-gradient slots, then the block or the KKT scatter
  auto emit = [&](int V) {
    // value / adjoint-gradient slots: callee overwrites (fx.setZero(); compute)
    const size_t a = appl_base + V;
    double* fdst = fx_out ? fx_out + (blocks ? a * OR : (size_t)d.con_starts[V]) : nullptr;
    if (fdst) std::memcpy(fdst, fx.data(), sizeof(double) * OR);
    if (agx_out && what != ORACLE_CON && what != ORACLE_JAC) {
      double* gdst = agx_out + (blocks ? a * IR : (size_t)d.grad_starts[V]);
      std::memcpy(gdst, agx.data(), sizeof(double) * IR);
    }
    if (what < ORACLE_JAC) return;
    const bool dohess = (what == ORACLE_JAC_ADJGRAD_HESS);
    if (blocks) {
      if (!kkt_blocks) return;
      double* blk = kkt_blocks + a * (size_t)(IR * (IR + 1) / 2 + OR * IR);   // (blocks are always dense: the mask is the NLP's)
      int k = 0;
      for (int i = 0; i < IR; i++) {
        for (int j = i; j < IR; j++) blk[k++] = dohess ? hx[j + (size_t)i * IR] : 0.0;
        for (int j = 0; j < OR; j++) blk[k++] = jx[j + (size_t)i * OR];
      }
      return;
    }
    // KKTFillAll / KKTFillJac (DenseFunctionBase.h:1413-1523), unique_constraints = true
    int freeloc = d.kkt_starts[V];
    const int* lpt = n->kkt_locs.data();
    for (int i = 0; i < IR; i++) {
      const int var = d.VLoc(i, V);
      if (dohess) {
        const int lk = n->clashes[var];
        if (lk >= 0) n->locks[lk].lock();
        for (int j = i; j < IR; j++)
          if (n->hess_nz[j + (size_t)IR * i]) kktvals[lpt[freeloc++]] += hx[j + (size_t)i * IR];   // AddHessianElem, TrapezoidalDefects.h:131-141
        if (lk >= 0) n->locks[lk].unlock();
      } else {
        freeloc += n->hess_count[i];
      }
      for (int j = 0; j < OR; j++) kktvals[lpt[freeloc++]] += jx[j + (size_t)i * OR];
    }
  };
  int V0 = 0;
  // SuperScalar batching (DenseFunctionBase.h:1318-1380): whole packs of four applications through the four-wide body,
  // results emitted pack lane by pack lane (ScalarCallBack), then the remainder one application at a time
  if (n->batch4 && n->ode4.fjgh && what == ORACLE_JAC_ADJGRAD_HESS && n->mode >= ORACLE_LGL3) {
    std::vector<v4d> x4(IR), l4(OR), fx4(OR), jx4((size_t)OR * IR), agx4(IR), hx4((size_t)IR * IR);
    for (; V0 + 4 <= d.nappl; V0 += 4) {
      for (int k = 0; k < 4; k++) {
        gather(d, V0 + k, X, L, x.data(), l.data());
        for (int i = 0; i < IR; i++) x4[i][k] = x[i];
        for (int j = 0; j < OR; j++) l4[j][k] = l[j];
      }
      oracle_defect_all_v4(&n->ode, &n->ode4, n->mode, n->blocked, x4.data(), l4.data(), fx4.data(), jx4.data(),
                           agx4.data(), hx4.data());
      for (int k = 0; k < 4; k++) {
        for (int j = 0; j < OR; j++) fx[j] = fx4[j][k];
        for (int i = 0; i < IR; i++) agx[i] = agx4[i][k];
        for (size_t e = 0; e < jx.size(); e++) jx[e] = jx4[e][k];
        for (size_t e = 0; e < hx.size(); e++) hx[e] = hx4[e][k];
        emit(V0 + k);
      }
    }
  }
  for (int V = V0; V < d.nappl; V++) {
    gather(d, V, X, (what == ORACLE_CON || what == ORACLE_JAC) ? nullptr : L, x.data(), l.data());
    switch (what) {
      case ORACLE_CON:
        oracle_defect_compute(&n->ode, n->mode, n->blocked, x.data(), fx.data());
        break;
      case ORACLE_CON_ADJGRAD:
      case ORACLE_JAC:
      case ORACLE_JAC_ADJGRAD:
        oracle_defect_jacobian(&n->ode, n->mode, n->blocked, x.data(), fx.data(), jx.data());
        if (what != ORACLE_JAC)
          for (int c = 0; c < IR; c++) {
            double acc = 0.0;
            for (int r = 0; r < OR; r++) acc += l[r] * jx[r + (size_t)c * OR];
            agx[c] = acc;
          }
        break;
      default:
        oracle_defect_all(&n->ode, n->mode, n->blocked, x.data(), l.data(), fx.data(), jx.data(), agx.data(),
                          hx.data());
    }
    emit(V);
  }
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------ phase indexer
int oracle_phase_num_vars(int xv, int uv, int pv, int spv, int cs, int ndefects, int blocked) {
  const int num_states = (cs - 1) * ndefects + 1;  // PhaseIndexer.h:66
  if (blocked) return num_states * (xv + 1) + ndefects * uv + pv + spv;  // :72-73
  return num_states * (xv + 1 + uv) + pv + spv;                           // :86
}

int oracle_phase_defect_index(int xv, int uv, int pv, int spv, int cs, int ndefects, int blocked, int var_offset,
                              int con_offset, int* vindex, int* cindex) {
  const int xt = xv + 1, xtu = xv + 1 + uv;
  const int num_states = (cs - 1) * ndefects + 1;
  const int nvars = oracle_phase_num_vars(xv, uv, pv, spv, cs, ndefects, blocked);
  const int param0 = nvars - pv - spv + var_offset;                       // ODEParamLocs, PhaseIndexer.h:93-95
  const int orows = (cs - 1) * xv;
  int irows, next_c = con_offset;
  if (!blocked) {
    irows = cs * xtu + pv;
    for (int i = 0; i < ndefects; i++) {                                  // PhaseIndexer.cpp:361-372
      int loc = 0;
      for (int j = 0; j < cs; j++) {
        const int state = i * (cs - 1) + j;
        for (int k = 0; k < xtu; k++) vindex[(size_t)i * irows + loc++] = var_offset + k + state * xtu;
      }
      for (int k = 0; k < pv; k++) vindex[(size_t)i * irows + loc++] = param0 + k;
    }
  } else {
    irows = cs * xt + uv + pv;
    const int ubase = var_offset + xt * num_states;                       // ODEFirstStateLocs.tail(UV), PhaseIndexer.h:75-77
    for (int i = 0; i < ndefects; i++) {                                  // PhaseIndexer.cpp:373-391
      int loc = 0;
      for (int j = 0; j < cs; j++) {
        const int state = i * (cs - 1) + j;
        for (int k = 0; k < xt; k++) vindex[(size_t)i * irows + loc++] = var_offset + k + state * xt;
      }
      for (int k = 0; k < uv; k++) vindex[(size_t)i * irows + loc++] = ubase + k + i * uv;
      for (int k = 0; k < pv; k++) vindex[(size_t)i * irows + loc++] = param0 + k;
    }
  }
  for (int i = 0; i < ndefects; i++)                                      // CinSet, PhaseIndexer.cpp:179-189
    for (int j = 0; j < orows; j++) cindex[(size_t)i * orows + j] = next_c++;
  return irows;
}

// ------------------------------------------------------------------------------------ NLP
oracle_nlp* oracle_nlp_create(const oracle_ode* ode, int mode, int blocked, int nappl, const int* vindex,
                              const int* cindex, int primal_vars, int equal_cons, int threads) {
  return oracle_nlp_create_ex(ode, mode, blocked, nappl, vindex, cindex, primal_vars, equal_cons, threads, 0);
}

// flags bit 0: EnableHessianSparsity of the Trapezoidal defects (TrapezoidalDefects.h:39, 124-141)
oracle_nlp* oracle_nlp_create_ex(const oracle_ode* ode, int mode, int blocked, int nappl, const int* vindex,
                                 const int* cindex, int primal_vars, int equal_cons, int threads, int flags) {
  int ir, orr;
  if (oracle_defect_sizes(mode, ode->xv, ode->uv, ode->pv, blocked, &ir, &orr)) return nullptr;
  oracle_nlp* n = new oracle_nlp;
  n->ode = *ode;
  n->mode = mode, n->blocked = blocked, n->ir = ir, n->orr = orr;
  n->primal = primal_vars, n->equal = equal_cons, n->threads = std::max(1, threads);
  n->kktdim = primal_vars + equal_cons;                                   // setMATDimensions (no slacks / inequalities)
  n->hess_nz.assign((size_t)ir * ir, 1);
  if ((flags & 1) && mode == ORACLE_TRAPEZOIDAL) {
    const int nn = ode->xv, m = blocked ? 0 : ode->uv, p = blocked ? ode->uv + ode->pv : ode->pv;
    trapezoidal_hessian_mask(n, nn + 1 + m, p, nn);
  }
  n->hess_count.assign(ir, 0);
  for (int i = 0; i < ir; i++)
    for (int j = i; j < ir; j++) n->hess_count[i] += n->hess_nz[j + (size_t)ir * i] ? 1 : 0;
  n->nkkt_per_appl = num_kkt_eles(n, ir, orr, true, true);
  IndexData all;
  all.ir = ir, all.orr = orr, all.nappl = nappl;
  all.vindex.assign(vindex, vindex + (size_t)ir * nappl);
  all.cindex.assign(cindex, cindex + (size_t)orr * nappl);
  thread_split(all, n->threads, n->thr);                                  // analyzeThreading, ByApplication

  n->num_user_kkt = n->nkkt_per_appl * nappl;                             // countElems
  n->num_solver_kkt = primal_vars + equal_cons;                           // primal diags + equality pivots
  n->num_kkt = n->num_user_kkt + n->num_solver_kkt;
  n->kkt_rows.assign(n->num_kkt, -1);
  n->kkt_cols.assign(n->num_kkt, -1);
  n->kkt_thr.assign(n->num_kkt, 0);
  n->solver_coeffs.assign(n->num_solver_kkt, 0.0);

  int freeloc = 0;                                                        // getMATSpace
  for (size_t t = 0; t < n->thr.size(); t++) {
    const int start = freeloc;
    get_kkt_space(n, n->thr[t], freeloc, primal_vars);
    std::fill(n->kkt_thr.begin() + start, n->kkt_thr.begin() + freeloc, (int)t);
  }
  std::vector<char> touched((size_t)n->thr.size() * n->kktdim, 0);
  for (int i = 0; i < n->num_user_kkt; i++) touched[(size_t)n->kkt_thr[i] * n->kktdim + n->kkt_cols[i]] = 1;
  n->clashes.assign(n->kktdim, -1);
  int nclash = 0;
  for (int c = 0; c < n->kktdim; c++) {
    int s = 0;
    for (size_t t = 0; t < n->thr.size(); t++) s += touched[t * n->kktdim + c];
    if (s > 1) n->clashes[c] = nclash++;
  }
  n->locks = std::vector<std::mutex>(nclash);

  n->agx_coeffs.assign((size_t)ir * nappl, 0.0);                          // setRHSDimensions / getRHSSpace
  n->econ_coeffs.assign((size_t)orr * nappl, 0.0);
  n->agx_rows.resize(n->agx_coeffs.size());
  n->econ_rows.resize(n->econ_coeffs.size());
  int gfree = 0, cfree = 0;
  for (auto& d : n->thr) {
    d.grad_starts.resize(d.nappl);
    d.con_starts.resize(d.nappl);
    for (int V = 0; V < d.nappl; V++) {
      d.grad_starts[V] = gfree;
      for (int i = 0; i < ir; i++) n->agx_rows[gfree++] = d.VLoc(i, V);
    }
    for (int V = 0; V < d.nappl; V++) {
      d.con_starts[V] = cfree;
      for (int j = 0; j < orr; j++) n->econ_rows[cfree++] = d.CLoc(j, V);
    }
  }
  for (int i = 0; i < primal_vars; i++) {                                 // finalizeData
    n->kkt_rows[n->num_user_kkt + i] = i;
    n->kkt_cols[n->num_user_kkt + i] = i;
  }
  for (int i = 0; i < equal_cons; i++) {
    n->kkt_rows[n->num_user_kkt + primal_vars + i] = primal_vars + i;
    n->kkt_cols[n->num_user_kkt + primal_vars + i] = primal_vars + i;
  }
  analyze_sparsity(n);
  return n;
}

/* bench.py's cpu_baseline leg: evaluate LGL evalKKT four applications at a time (batch4.h).  Returns 0, or -1 when this
 * library / ODE has no four-wide body (the scalar loop stays). */
int oracle_nlp_set_batch4(oracle_nlp* n, int on) {
  if (!on) { n->batch4 = false; return 0; }
  if (oracle_get_ode4(&n->ode, &n->ode4) != 0 || n->mode < ORACLE_LGL3) return -1;
  n->batch4 = true;
  return 0;
}

void oracle_nlp_destroy(oracle_nlp* n) { delete n; }
int oracle_nlp_kkt_dim(const oracle_nlp* n) { return n->kktdim; }
int oracle_nlp_nnz(const oracle_nlp* n) { return (int)n->inner.size(); }
int oracle_nlp_num_user_kkt(const oracle_nlp* n) { return n->num_user_kkt; }
void oracle_nlp_csr(const oracle_nlp* n, int* outer, int* inner) {
  std::memcpy(outer, n->outer.data(), sizeof(int) * n->outer.size());
  std::memcpy(inner, n->inner.data(), sizeof(int) * n->inner.size());
}
void oracle_nlp_kkt_locations(const oracle_nlp* n, int* locs) {
  std::memcpy(locs, n->kkt_locs.data(), sizeof(int) * n->kkt_locs.size());
}
void oracle_nlp_kkt_coords(const oracle_nlp* n, int* rows, int* cols) {
  std::memcpy(rows, n->kkt_rows.data(), sizeof(int) * n->kkt_rows.size());
  std::memcpy(cols, n->kkt_cols.data(), sizeof(int) * n->kkt_cols.size());
}

static void run_threads(oracle_nlp* n, int what, const double* X, const double* LE, double* fx, double* agx,
                        double* kktvals, double* kkt_blocks, bool blocks) {
  const int T = (int)n->thr.size();
  std::vector<size_t> base(T, 0);
  for (int t = 1; t < T; t++) base[t] = base[t - 1] + n->thr[t - 1].nappl;
  if (T > 1) {
    if (!n->pool || n->pool->size() != T - 1) n->pool.reset(new WorkerPool(T - 1));
    n->pool->post([&, n](int t) { eval_slice(n, n->thr[t], what, X, LE, fx, agx, kktvals, kkt_blocks, blocks, base[t]); });
  }
  if (T > 0) eval_slice(n, n->thr[T - 1], what, X, LE, fx, agx, kktvals, kkt_blocks, blocks, base[T - 1]);  // caller runs last slice
  if (T > 1) n->pool->wait();
}

int oracle_nlp_eval(oracle_nlp* n, int what, const double* X, const double* LE, double* FXE, double* AGX,
                    double* kktvals) {
  if (what < ORACLE_CON || what > ORACLE_JAC_ADJGRAD_HESS) return -1;
  if (what >= ORACLE_JAC && !kktvals) return -2;
  std::fill(n->agx_coeffs.begin(), n->agx_coeffs.end(), 0.0);             // setRHSCoeffsZero
  std::fill(n->econ_coeffs.begin(), n->econ_coeffs.end(), 0.0);
  run_threads(n, what, X, LE, n->econ_coeffs.data(), n->agx_coeffs.data(), kktvals, nullptr, false);
  // fillRHS: serial scatter-add (NonLinearProgram.h:401-407)
  if (FXE)
    for (size_t i = 0; i < n->econ_coeffs.size(); i++) FXE[n->econ_rows[i]] += n->econ_coeffs[i];
  if (AGX && what != ORACLE_CON && what != ORACLE_JAC)
    for (size_t i = 0; i < n->agx_coeffs.size(); i++) AGX[n->agx_rows[i]] += n->agx_coeffs[i];
  if (what >= ORACLE_JAC)                                                 // fillSolverCoeffs
    for (int i = 0; i < n->num_solver_kkt; i++) kktvals[n->kkt_locs[n->num_user_kkt + i]] += n->solver_coeffs[i];
  return 0;
}

int oracle_nlp_eval_blocks(oracle_nlp* n, int what, const double* X, const double* LE, double* fx_blocks,
                           double* agx_blocks, double* kkt_blocks) {
  if (what < ORACLE_CON || what > ORACLE_JAC_ADJGRAD_HESS) return -1;
  run_threads(n, what, X, LE, fx_blocks, agx_blocks, nullptr, kkt_blocks, true);
  return 0;
}
}
