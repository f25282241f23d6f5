// Dense stage for wide shapes (Dims::WIDE): one four-wave workgroup per CU works on one segment at a time.
//
// The products are those of lgl_defect_kernel's dense stage (LGLDefects.h:414-512): H = sum_i DI_i^T (hE_i H^_i) DI_i
// plus the cardinal blocks and the rank-2 time update, J = DC + sum_i hE_i J^_i DI_i, g = J^T lam.  What differs is
// what stays in LDS: with 32 states in LGL7 the M^T tile (124 KB) and the DC tile (113 KB) of the narrow kernel do
// not fit beside DI, so
//   * only DI (state rows per segment, constant rows per launch) is resident, shared by the four waves;
//   * H is produced one 16-row tile row at a time: the wave that takes tile row rt forms the 16 columns of
//     M_i = (hE_i H^_i) DI_i it needs in registers, interior by interior -- the accumulator layout of that product is
//     the B-operand layout of the H product, so M never goes through LDS -- and keeps the row's accumulators;
//   * the cardinal part of J (the DC tile) is formed from the slot's ODE Jacobians where it is used: as the initial
//     accumulator of a J^T tile, and in closed form for g = J^T lam;
//   * tile rows and J tiles are handed out to the waves through an LDS counter, largest first.
#pragma once
#include "defect_dims.h"

namespace asset_hip {

// LDS hand-off between the waves of the workgroup without draining the global stores (__syncthreads waits for them)
__device__ inline void wg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Where accumulator entry (tile (jt, ct), v, lane) of J^T finds its cardinal part -- entry (jr, c) of DC,
// C_ij [cc == r] + h D_ij dfdy_j[r][cc] -+ sd (LGLDefects.h:467-500): everything but the values depends only on the
// lane, so it is decoded once per handle into one word per entry (the per-entry index arithmetic cost more VALU
// instructions than the tile's MFMAs take cycles).  Word: bits 0-15 slot offset of dfdy_j[r][cc] (or the zero cell),
// 16-17 node j, 18-19 interior i, 20 diagonal (cc == r), 21 column T, 22 column TF.  Stored [(jt*TI + ct)*4 + v][lane].
template <class Ode, int SCH, bool BLOCKED>
__global__ __launch_bounds__(64) void wide_setup_kernel(unsigned int* out) {
  using D = Dims<Ode, SCH, BLOCKED>;
  constexpr int n = D::n, q = D::q, N = D::N, P0 = D::P0, OR = D::OR, TI = D::TI, TJ = D::TJ;
  constexpr int ZERO = D::WSLOTD + D::s_Z0;
  static_assert(ZERO < 65536, "slot offsets are packed into 16 bits");
  const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  for (int jt = 0; jt < TJ; jt++)
    for (int ct = 0; ct < TI; ct++)
      for (int v = 0; v < 4; v++) {
        const int jr = 16 * jt + lr, c = 16 * ct + lk + 4 * v;
        unsigned int w = unsigned(ZERO);
        if (jr < OR && c < P0) {
          const int i = jr / n, r = jr - i * n, j = c / q, cc = c - j * q;
          const int jp = Ode::JPOS[r * N + cc];
          w = unsigned(jp >= 0 ? D::w_CJ + j * D::NZJ + jp : ZERO) | (unsigned(j) << 16) | (unsigned(i) << 18) |
              (unsigned(cc == r) << 20) | (unsigned(c == D::T) << 21) | (unsigned(c == D::TF) << 22);
        }
        out[((jt * TI + ct) * 4 + v) * 64 + lane] = w;
      }
}

// Structurally zero operand fragments are known at compile time (Ode::HPOS / JPOS): k-step kk of the M product for row
// tile mt reads the 16 x 4 block H^[16mt.., 4kk..], k-step kk of a J^T tile reads J^[rows of the tile, 4kk..]; a block
// with no structural entry contributes nothing and its MFMA is skipped (32-state BASELINE ODE: 13 of 27 and 7 of 9).
template <class Ode, class D>
struct WideSparsity {
  static constexpr int N = D::N, n = D::n, KS = D::KS, MT = (D::NP + 15) / 16, TJ = D::TJ, K = D::K, OR = D::OR;
  static constexpr unsigned hmask(int mt) {              // bit kk: block (mt, kk) of H^ has a structural entry
    unsigned m = 0;
    for (int kk = 0; kk < KS; kk++)
      for (int r = 16 * mt; r < 16 * mt + 16 && r < N; r++)
        for (int c = 4 * kk; c < 4 * kk + 4 && c < N; c++)
          if (Ode::HPOS[r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r] >= 0) m |= 1u << kk;
    return m;
  }
  static constexpr unsigned jmask(int jt) {              // bit kk: some defect row of J tile jt has an entry in inputs 4kk..4kk+3
    unsigned m = 0;
    for (int kk = 0; kk < KS; kk++)
      for (int jr = 16 * jt; jr < 16 * jt + 16 && jr < OR; jr++)
        for (int c = 4 * kk; c < 4 * kk + 4 && c < N; c++)
          if (Ode::JPOS[(jr % n) * N + c] >= 0) m |= 1u << kk;
    return m;
  }
  static constexpr bool di_entry(int b, int c) {         // can DI_i[b][c] be non-zero (LGLDefects.h:417-458)?
    constexpr int q = D::q, P0 = D::P0, IR = D::IR, T = D::T, TF = D::TF;
    if (c >= IR || b >= N) return false;
    if (b < n) {
      if (c == T || c == TF) return true;                // time columns: -+ sum_j B_ij f_j
      if (c < P0) { const int cc = c % q; return cc == b || Ode::JPOS[b * N + cc] >= 0; }
      return Ode::JPOS[b * N + q + (c - P0)] >= 0;       // parameter columns
    }
    if (b == T) return c == T || c == TF;
    if (b < q) return c < P0 && (c % q) == b;            // control-interpolation rows
    return c == P0 + (b - q);                            // parameter identity rows
  }
  static constexpr unsigned dmask(int ct) {              // bit kk: rows 4kk..4kk+3 of DI have an entry in column tile ct
    unsigned m = 0;
    for (int kk = 0; kk < KS; kk++)
      for (int b = 4 * kk; b < 4 * kk + 4; b++)
        for (int c = 16 * ct; c < 16 * ct + 16; c++)
          if (di_entry(b, c)) m |= 1u << kk;
    return m;
  }
  struct DTable {
    unsigned m[D::TI];
    constexpr DTable() : m{} {
      for (int ct = 0; ct < D::TI; ct++) m[ct] = dmask(ct);
    }
  };
  static constexpr DTable DT{};
  struct JTable {
    unsigned m[TJ];
    constexpr JTable() : m{} {
      for (int jt = 0; jt < TJ; jt++) m[jt] = jmask(jt);
    }
  };
  static constexpr JTable JT{};
  struct HTable {                                        // (tables, so that nothing of this is evaluated on the device)
    unsigned m[MT];
    constexpr HTable() : m{} {
      for (int mt = 0; mt < MT; mt++) m[mt] = hmask(mt);
    }
  };
  static constexpr HTable HT{};
};

#define ASSET_WIDE_WGS 2              // workgroups per CU when two working sets fit its LDS (halves the register budget)
template <class Ode, int SCH, bool BLOCKED, int LEVEL, bool ASM>
__device__ __forceinline__ void lgl_wide_dense_body(const EvalArgs& a) {
  using D = Dims<Ode, SCH, BLOCKED>;
  static_assert(D::WIDE && LEVEL >= 1, "wide shapes, derivative kinds only (the value comes from the ODE stage)");
  constexpr int CS = D::CS, K = D::K, n = D::n, p = D::p, q = D::q, N = D::N, T = D::T, TF = D::TF, P0 = D::P0;
  constexpr int IR = D::IR, OR = D::OR, IRP = D::IRP, NP = D::NP, KS = D::KS, MT = (NP + 15) / 16;
  constexpr int TI = D::TI, TJ = D::TJ, NTH = D::NTH, NCR = D::NCR, ROWS = K * n;
  constexpr int ZERO = D::WSLOTD + D::s_Z0;              // slot-relative offset of a cell that holds 0.0
  // column tiles of one J work unit: a whole tile row when the workgroup has the CU's registers to itself (3 / 5 / 9
  // tiles per unit measured 19.4 / 16.0 / 15.1 ms on the 32-state LGL7 shape), half a row with two workgroups per CU
  constexpr bool FULLREG = !(D::lds_bytes_dense() * ASSET_WIDE_WGS <= 160 * 1024);
  constexpr int CTC = (FULLREG || TI <= 5) ? TI : (TI + 1) / 2;
  constexpr int NJC = (TI + CTC - 1) / CTC;
  constexpr int NHU = (LEVEL >= 2 || !ASM) ? TI : 0;     // H work units (Jacobian-only block kinds store zeros there)
  constexpr int NUNITS = NHU + TJ * NJC;
  using WS = WideSparsity<Ode, D>;
  // the interior loop of an H unit is unrolled when the workgroup has a CU's registers to itself (two per CU: it spills)
  constexpr int IUNROLL = FULLREG ? K : 1;
  static_assert(KS <= 32 && TJ <= 32, "fragment masks are 32-bit");
  (void)p;

  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* tabL = lds;
  double* slotb = lds + D::TABSZ;
  double* scr = slotb + D::WSLOTD;
  const int tid = threadIdx.x, lane = tid & 63;
  [[maybe_unused]] const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // (timing rows, -DASSET_TIMING)
  const int lr = lane & 15, lk = lane >> 4;
  for (int e = tid; e < D::TABSZ; e += 256) tabL[e] = reinterpret_cast<const double*>(&d_lgl_tab[D::TAB])[e];
  const LglTab& tab = *reinterpret_cast<const LglTab*>(tabL);

  const int per = a.nseg / int(gridDim.x), rem = a.nseg % int(gridDim.x);
  const int wg_first = int(blockIdx.x) * per + min(int(blockIdx.x), rem);
  const int wg_count = per + (int(blockIdx.x) < rem ? 1 : 0);

  double* DIx = scr + D::s_DIx;
  double* DIc = scr + D::s_DIc;
  double* SD = scr + D::s_DC;                            // sum_j D_ij f_j + E_i f^_i per defect row
  double* WL = scr + D::s_WL;                            // sum_i D_ij lam_(i,r)
  const unsigned int* dcinfo = static_cast<const unsigned int*>(a.lane_consts);   // wide_setup_kernel
  int* counter = reinterpret_cast<int*>(scr + D::s_CNT);
  double* LSD = scr + D::s_CNT + 1;
  double* R2 = scr + D::s_R2;
  double* HI = scr + D::s_HI;

  // run-time look-ups of the ODE's sparsity tables go through 16-bit LDS copies (a global look-up per entry costs more
  // than the entry's arithmetic)
  static_assert(D::NZJ < 32768 && D::NZH < 32768, "16-bit sparsity tables");
  short* jposL = reinterpret_cast<short*>(scr + D::s_JP);
  short* hposL = reinterpret_cast<short*>(scr + D::s_HP);
  for (int e = tid; e < n * N; e += 256) jposL[e] = short(Ode::JPOS[e]);
  for (int e = tid; e < D::NH; e += 256) hposL[e] = short(Ode::HPOS[e]);
  auto cj_at = [&](int j, int r, int cc) { const int jp = jposL[r * N + cc]; return jp >= 0 ? D::w_CJ + j * D::NZJ + jp : ZERO; };
  // slot offset of dfdy_j[r][cc], r = 0..n-1, for the block column this thread owns in the column-parallel passes
  constexpr int CMAIN = (P0 < 256) ? P0 : 256;
  const int ccol = (tid < CMAIN) ? tid : 0, cj_ = ccol / q, ccc = ccol - cj_ * q;

  // ---- per-lane constants (once per launch; a workgroup walks hundreds of segments)
  int avo[KS], avs[KS];                                  // DI_i^T fragment: row 4kk + lk of interior 0, stride in i
  int bo[LEVEL >= 2 ? MT : 1][KS];                       // fragment of H^_i for i = 0 (or the zero cell); stride NZH in i
#pragma unroll
  for (int kk = 0; kk < KS; kk++) {
    const int r = 4 * kk + lk;
    avo[kk] = ((r < n) ? D::s_DIx + r * IRP : ((r < N) ? D::s_DIc + (r - n) * IRP : D::s_R2 + 2 * IRP)) + lr;   // k-padding: zero row
    avs[kk] = (r < n) ? n * IRP : ((r < N) ? NCR * IRP : 0);
    if constexpr (LEVEL >= 2) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        const int acol = 16 * mt + lr;
        int v = ZERO;
        if (r < N && acol < N) {
          const int hp = Ode::HPOS[(r >= acol) ? r * (r + 1) / 2 + acol : acol * (acol + 1) / 2 + r];
          if (hp >= 0) v = D::w_IH + hp;
        }
        bo[mt][kk] = v;
      }
    }
  }
  auto avf = [&](int ct, int i, int kk) { return scr[avo[kk] + i * avs[kk] + 16 * ct]; };

  // ---- per-launch constants of the scratch: constant rows of DI_i (LGLDefects.h:417-458), rank-2 direction, zeros
  wg_lds_barrier();                                      // weight tables
  for (int e = tid; e < K * NCR * IRP; e += 256) {
    const int i = e / (NCR * IRP), rem2 = e - i * NCR * IRP;
    const int r = n + rem2 / IRP, c = rem2 % IRP;
    double v = 0.0;
    if (c < IR) {
      if (r == T) v = (c == T) ? (1.0 - tab.s[i]) : ((c == TF) ? tab.s[i] : 0.0);
      else if (r > T && r < q) { if (c < P0 && (c % q) == r) v = tab.U[i][c / q]; }
      else if (r >= q && r < N) { if (c == P0 + (r - q)) v = 1.0; }
    }
    DIc[e] = v;
  }
  if constexpr (IR < IRP) {
    for (int e = tid; e < K * n * IRP; e += 256) DIx[e] = 0.0;   // padding columns (the per-segment pass writes c < IR)
  }
  for (int e = tid; e < IRP; e += 256) {
    R2[e] = (e == TF) ? 1.0 : ((e == T) ? -1.0 : 0.0);
    R2[IRP + e] = 0.0;
    R2[2 * IRP + e] = 0.0;
  }
  if (tid < 2) scr[D::s_Z0 + tid] = 0.0;
  if (tid == 0) *counter = 0;

  constexpr int NPRE = (D::WSLOTD + 255) / 256;
  double pre[NPRE];
  {
    const double* W0 = a.work + size_t(wg_first) * D::WSLOT;
#pragma unroll
    for (int t = 0; t < NPRE; t++) pre[t] = (tid + 256 * t < D::WSLOTD && wg_count > 0) ? W0[tid + 256 * t] : 0.0;
  }

#if defined(ASSET_TIMING)   // cycle stamps of the workgroup's second segment, one row per wave (tools/dbg_time.py)
  long long tstamp[12];
  long long tunit[3] = {0, 0, 0};                        // cycles in H units / J units, units taken
  int nts = 0;
#define TSW() do { if (g == 1 && nts < 12) tstamp[nts++] = clock64(); } while (0)
#else
#define TSW() do {} while (0)
#endif
  for (int g = 0; g < wg_count; g++) {
    const size_t seg = size_t(wg_first + g);
    TSW();
    // slot: workspace -> LDS; the loads were issued one segment ago
#pragma unroll
    for (int t = 0; t < NPRE; t++)
      if (tid + 256 * t < D::WSLOTD) slotb[tid + 256 * t] = pre[t];
    wg_lds_barrier();
    if (g + 1 < wg_count) {
      const double* W1 = a.work + (seg + 1) * D::WSLOT;
#pragma unroll
      for (int t = 0; t < NPRE; t++) pre[t] = (tid + 256 * t < D::WSLOTD) ? W1[tid + 256 * t] : 0.0;
    }
    const double* S = slotb;
    const double* z = S + D::w_z;
    const double* lam = S + D::w_lam;
    const double h = z[TF] - z[T];
    TSW();   // slot

    // ---- state rows of DI_i, one thread per column:  A_ij [cc == r] + h B_ij dfdy_j[r][cc]   (LGLDefects.h:430-444)
    int cjo[n];                                           // (looked up per segment: 32 registers less across the tile work)
#pragma unroll
    for (int r = 0; r < n; r++) { const int jp = jposL[r * N + ccc]; cjo[r] = jp >= 0 ? D::w_CJ + cj_ * D::NZJ + jp : ZERO; }
    if (tid < CMAIN) {
      double wbh[K], wah[K];
#pragma unroll
      for (int i = 0; i < K; i++) { wbh[i] = tab.B[i][cj_] * h; wah[i] = tab.A[i][cj_]; }
      double jvr[n];                                      // every read is issued before the first write
#pragma unroll
      for (int r = 0; r < n; r++) jvr[r] = S[cjo[r]];
#pragma unroll
      for (int r = 0; r < n; r++)
#pragma unroll
        for (int i = 0; i < K; i++) {
          double v = wbh[i] * jvr[r];
          if (ccc == r) v += wah[i];
          DIx[(i * n + r) * IRP + tid] = v;
        }
    }
    for (int c = CMAIN + tid; c < IR; c += 256) {         // parameter columns, columns beyond one pass
      for (int r = 0; r < n; r++) {
        double v[K];
#pragma unroll
        for (int i = 0; i < K; i++) v[i] = 0.0;
        if (c < P0) {
          const int j = c / q, cc = c - j * q;
          const double jv = S[cj_at(j, r, cc)];
#pragma unroll
          for (int i = 0; i < K; i++) v[i] = (tab.B[i][j] * h) * jv + ((cc == r) ? tab.A[i][j] : 0.0);
        } else {
          for (int jj = 0; jj < CS; jj++) {
            const double jv = S[cj_at(jj, r, q + (c - P0))];
#pragma unroll
            for (int i = 0; i < K; i++) v[i] += (tab.B[i][jj] * h) * jv;
          }
        }
#pragma unroll
        for (int i = 0; i < K; i++) DIx[(i * n + r) * IRP + c] = v[i];
      }
    }
    for (int e = tid; e < CS * n; e += 256) {             // multiplier weights of the cardinal part of J^T lam
      const int j = e / n, r = e - j * n;
      double v = 0.0;
#pragma unroll
      for (int i = 0; i < K; i++) v += tab.D[i][j] * lam[i * n + r];
      WL[e] = v;
    }
    wg_lds_barrier();
    TSW();   // DI state rows
    // ---- time columns of DI (LGLDefects.h:446-450), the time-column vector of DC (:484-500), defect values (:96-103)
    for (int e = tid; e < ROWS; e += 256) {
      const int i = e / n, r = e - i * n;
      double sb = 0.0, sd = tab.E[i] * S[D::w_If + i * n + r], fxv = 0.0;
#pragma unroll
      for (int jj = 0; jj < CS; jj++) {
        const double fv = S[D::w_Cf + jj * n + r];
        sb += tab.B[i][jj] * fv;
        sd += tab.D[i][jj] * fv;
        fxv += tab.C[i][jj] * z[jj * q + r];
      }
      DIx[e * IRP + T] -= sb;
      DIx[e * IRP + TF] += sb;
      SD[e] = sd;
      if (a.FX) a.FX[seg * OR + e] = h * sd + fxv;
    }
    wg_lds_barrier();
    TSW();   // time columns
    // ---- HI = sum_i E_i g^_i^T DI_i  (interior part of J^T lam; with the cardinal g it gives the time partial HTpar)
    for (int c = tid; c < IRP; c += 256) {
      double v = 0.0;
#pragma unroll
      for (int i = 0; i < K; i++) {
        double sacc = 0.0;
#pragma unroll
        for (int b = 0; b < N; b++)
          sacc += S[D::w_Ig + i * N + b] * ((b < n) ? DIx[(i * n + b) * IRP + c] : DIc[(i * NCR + (b - n)) * IRP + c]);
        v += tab.E[i] * sacc;
      }
      HI[c] = v;
      if constexpr (LEVEL >= 2) {                         // rank-2 row (LGLDefects.h:403-411, 504-511)
        const double ih = 1.0 / h;
        const int jn = c / q;
        double w = v + S[(c < P0) ? D::w_Cg + jn * N + (c - jn * q) : ZERO] * ih;
        if constexpr (p > 0) {
          if (c >= P0 && c < IR) {
            for (int j = 0; j < CS; j++) w += S[D::w_Cg + j * N + q + (c - P0)] * ih;
          }
        }
        R2[IRP + c] = w;
      }
    }
    if (tid == 255) {                                     // lam . (time-column vector): the T / TF entries of J^T lam
      double lsd = 0.0;
#pragma unroll 8
      for (int jr = 0; jr < OR; jr++) lsd += lam[jr] * SD[jr];
      LSD[0] = lsd;
    }
    wg_lds_barrier();
    TSW();   // HI, rank-2 row

    // ---- g = J^T lam: interior part h HI, cardinal part in closed form from the slot
    if (a.AGX) {
      for (int c = tid; c < IR; c += 256) {
        double v = h * HI[c];
        if (c < P0) {
          const int j = c / q, cc = c - j * q;
          double sacc = 0.0;
          if (c < CMAIN) {
#pragma unroll
            for (int r = 0; r < n; r++) sacc += WL[cj_ * n + r] * S[cjo[r]];
          } else {
            for (int r = 0; r < n; r++) sacc += WL[j * n + r] * S[cj_at(j, r, cc)];
          }
          v += h * sacc;
          if (cc < n) {
#pragma unroll
            for (int i = 0; i < K; i++) v += tab.C[i][j] * lam[i * n + cc];
          }
          if (c == T) v -= LSD[0];
          if (c == TF) v += LSD[0];
        } else {
          double sacc = 0.0;
          for (int jj = 0; jj < CS; jj++)
            for (int r = 0; r < n; r++) sacc += WL[jj * n + r] * S[cj_at(jj, r, q + (c - P0))];
          v += h * sacc;
        }
        a.AGX[seg * IR + c] = v;
      }
    }
    TSW();   // adjoint gradient
    // ---- tiles.  Destination of an accumulator entry: its slot of the segment's KKT block, or (ASM) the solver's
    //      value array through the fragment-ordered slot -> location map (defect_dims.h, EvalArgs::kmap)
    double* const kkt_dst = ASM ? a.values : (a.KKT ? a.KKT + seg * size_t(D::NKKT) : nullptr);
    const int* const kmap_seg = ASM ? a.kmap + seg * size_t((NTH + TI * TJ) * 4) * 64 + lane : nullptr;
    auto put = [&](int frag, int slot, double val) {
      if constexpr (ASM) {
        asm_put(a, kkt_dst, kmap_seg[frag * 64], val);
      } else {
        if (slot >= 0) kkt_dst[slot] = val;
      }
    };
    auto col_start = [](int c) { return c * (IR + OR) - c * (c - 1) / 2; };   // first slot of block column c

    if (kkt_dst) {
      for (;;) {
        int u = 0;
        // (an LDS atomic proper: through a generic pointer it becomes a flat atomic, whose return waits for every
        //  outstanding global store of the wave -- measured 10k cycles per unit)
        if (lane == 0)
          u = __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) int*>(
                                         reinterpret_cast<uintptr_t>(counter) & 0xffffffffu),
                                     1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= NUNITS) break;
#if defined(ASSET_TIMING)
        const long long tu0 = clock64();
#endif
        if (u < NHU) {
          if (LEVEL < 2 && (a.flags & 1)) continue;        // Jacobian kinds, ASSET_HIP_KEEP_HESSIAN_SLOTS: nothing to store
          // ------------------------------------------------ tile row rt of H (lower triangle): tiles ct = 0..rt
          const int rt = TI - 1 - u;                       // largest first
          const int r = 16 * rt + lr;
          d4 acc[TI];
          if constexpr (LEVEL >= 2) {
            const double y2 = R2[(lk == 0 ? IRP : (lk == 1 ? 0 : 2 * IRP)) + 16 * rt + lr];
#pragma unroll
            for (int ct = 0; ct < TI; ct++) {
              acc[ct] = d4{0.0, 0.0, 0.0, 0.0};            // (all of them: initialising only ct <= rt measured 15 % slower on
              if (ct > rt) continue;                       //  the shapes that run two workgroups per CU)
              if (tiles_share_node<D>(ct, rt)) {           // cardinal diagonal / parameter blocks (LGLDefects.h:386-402)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                  // every look-up is unconditional (clamped indices, "no entry" -> the zero cell): a lane-dependent
                  // branch around an LDS read costs a full exposed latency
                  const int c = 16 * ct + lk + 4 * v;
                  const bool ok = (c < IR && r < IR && r >= c);
                  const int cn = (c < P0) ? c : P0 - 1, jn = cn / q, cc = cn - jn * q;      // node column
                  const int rr = (r >= P0) ? q + (r - P0) : r - jn * q;                    // ODE input index of the row
                  const bool okn = ok && c < P0 && (r >= P0 || r / q == jn);
                  const int hp = hposL[okn ? rr * (rr + 1) / 2 + cc : 0];
                  double val = S[(okn && hp >= 0) ? D::w_CH + jn * D::NZH + hp : ZERO];
                  if constexpr (p > 0) {                   // parameter-parameter entries: summed over the cardinal nodes
                    const bool okp = ok && c >= P0;
                    const int r2 = q + (r - P0), c2 = q + (c - P0);
                    const int hq = hposL[okp ? r2 * (r2 + 1) / 2 + c2 : 0];
#pragma unroll
                    for (int j = 0; j < CS; j++) val += S[(okp && hq >= 0) ? D::w_CH + j * D::NZH + hq : ZERO];
                  }
                  acc[ct][v] = val;
                }
              }
              const double x2 = R2[(lk == 0 ? 0 : (lk == 1 ? IRP : 2 * IRP)) + 16 * ct + lr];
              acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, y2, acc[ct], 0, 0, 0);   // rank-2 time update
            }
#pragma unroll(IUNROLL)
            for (int i = 0; i < K; i++) {                  // (a rolled loop copies all accumulators at its back edge)
              const double he = h * tab.E[i];
              // (an MFMA fed by an LDS read issued right before it runs at half rate -- measured 135 vs 74 cycles -- so
              //  every operand is read a batch ahead of the MFMAs that use it)
              double afr[KS], bv[MT][KS];
#pragma unroll
              for (int kk = 0; kk < KS; kk++) afr[kk] = avf(rt, i, kk);   // (a run-time DI mask for tile rt did not pay here)
#pragma unroll
              for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int kk = 0; kk < KS; kk++)
                  if (WS::HT.m[mt] >> kk & 1) bv[mt][kk] = S[bo[mt][kk] + ((bo[mt][kk] != ZERO) ? i * D::NZH : 0)];
              // M_i[:, 16rt..16rt+16) = (hE_i H^_i) DI_i[:, tile]: with H^ as the A operand the accumulator entry v of
              // row tile mt -- row 16mt + lk + 4v of M_i, column lr -- is exactly the B operand of k-step 4mt + v of the
              // H product below, so M never leaves the registers
              double bm[KS];
#pragma unroll
              for (int mt = 0; mt < MT; mt++) {
                d4 am = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < KS; kk++)
                  if (WS::HT.m[mt] >> kk & 1) am = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[mt][kk], afr[kk], am, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; v++)
                  if (4 * mt + v < KS) bm[4 * mt + v] = am[v] * he;   // hE_i scales the product (rows >= N of H^ are zero: the k-padding)
              }
              double af[2][KS];                            // fragments of tile ct + 1 fly while tile ct's MFMAs issue
#pragma unroll
              for (int kk = 0; kk < KS; kk++)
                if (WS::DT.m[0] >> kk & 1) af[0][kk] = avf(0, i, kk);
#pragma unroll
              for (int ct = 0; ct < TI; ct++) {
                if (ct > rt) continue;
                if (ct + 1 < TI && ct + 1 <= rt) {
#pragma unroll
                  for (int kk = 0; kk < KS; kk++)
                    if (WS::DT.m[ct + 1 < TI ? ct + 1 : 0] >> kk & 1) af[(ct + 1) & 1][kk] = avf(ct + 1, i, kk);
                }
#pragma unroll
                for (int kk = 0; kk < KS; kk++)
                  if (WS::DT.m[ct] >> kk & 1)
                    acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ct & 1][kk], bm[kk], acc[ct], 0, 0, 0);
              }
            }
          } else {
#pragma unroll
            for (int ct = 0; ct < TI; ct++) acc[ct] = d4{0.0, 0.0, 0.0, 0.0};   // Jacobian-only kinds: zero Hessian slots
          }
#pragma unroll
          for (int ct = 0; ct < TI; ct++) {
            if (ct > rt) continue;
            const int tix = rt * (rt + 1) / 2 + ct;
#pragma unroll
            for (int v = 0; v < 4; v++) {
              const int c = 16 * ct + lk + 4 * v;
              put(tix * 4 + v, (c < IR && r < IR && r >= c) ? col_start(c) + (r - c) : -1, acc[ct][v]);
            }
          }
        } else {
          // ------------------------------------------------ J^T tiles (jt, a chunk of column tiles):
          //   (hE_i J^_i DI_i)^T by MFMA, then the cardinal part, formed from the slot, added to the accumulators.
          //   Everything a tile needs is requested before the MFMAs of the tile ahead of it issue: one wave per SIMD
          //   has no other wave to hide a latency behind.
          const int uj = u - NHU, jt = uj / NJC, c0 = (uj - jt * NJC) * CTC;
          const int jr = 16 * jt + lr;
          const int ji = (jr < OR) ? jr / n : 0, jk = (jr < OR) ? jr - ji * n : 0;
          unsigned int dcw[CTC][4];
#pragma unroll
          for (int cq = 0; cq < CTC; cq++)
#pragma unroll
            for (int v = 0; v < 4; v++)
              dcw[cq][v] = (c0 + cq < TI) ? dcinfo[((jt * TI + c0 + cq) * 4 + v) * 64 + lane] : unsigned(ZERO);
          const double sdv = SD[(jr < OR) ? jr : 0];
          d4 acc[CTC];
#pragma unroll
          for (int cq = 0; cq < CTC; cq++) acc[cq] = d4{0.0, 0.0, 0.0, 0.0};
          const int i_lo = (16 * jt) / n, i_hi = min(K - 1, (16 * jt + 15) / n);   // interiors with defect rows in this tile
          const unsigned jm = WS::JT.m[jt];                // k-steps whose J^ fragment has a structural entry
#pragma unroll 1
          for (int i = i_lo; i <= i_hi; i++) {
            const double he = h * tab.E[i];
            double bj[KS];
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
              const int aa = 4 * kk + lk;
              const int jp = jposL[jk * N + ((aa < N) ? aa : N - 1)];
              bj[kk] = he * S[(jr < OR && ji == i && aa < N && jp >= 0) ? D::w_IJ + i * D::NZJ + jp : ZERO];
            }
            double af[2][KS];
#pragma unroll
            for (int kk = 0; kk < KS; kk++)
              if ((jm & WS::DT.m[c0]) >> kk & 1) af[0][kk] = avf(c0, i, kk);
#pragma unroll
            for (int cq = 0; cq < CTC; cq++) {
              if (c0 + cq >= TI) break;
              if (cq + 1 < CTC && c0 + cq + 1 < TI) {
#pragma unroll
                for (int kk = 0; kk < KS; kk++)
                  if ((jm & WS::DT.m[c0 + cq + 1]) >> kk & 1) af[(cq + 1) & 1][kk] = avf(c0 + cq + 1, i, kk);
              }
#pragma unroll
              for (int kk = 0; kk < KS; kk++)
                if ((jm & WS::DT.m[c0 + cq]) >> kk & 1)
                  acc[cq] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cq & 1][kk], bj[kk], acc[cq], 0, 0, 0);
            }
          }
#pragma unroll
          for (int cq = 0; cq < CTC; cq++) {
            const int ct = c0 + cq;
            if (ct >= TI) break;
#pragma unroll
            for (int v = 0; v < 4; v++) {
              const int c = 16 * ct + lk + 4 * v;
              // entry (jr, c) of DC (LGLDefects.h:467-500) from its pre-decoded word; branch-free
              const unsigned int wd = dcw[cq][v];
              const int j = (wd >> 16) & 3, i2 = (wd >> 18) & 3;
              double val = (tab.D[i2][j] * h) * S[wd & 0xffffu];
              val += (wd & (1u << 20)) ? tab.C[i2][j] : 0.0;
              val += (wd & (1u << 22)) ? sdv : ((wd & (1u << 21)) ? -sdv : 0.0);
              if constexpr (p > 0) {
                const bool okp = (jr < OR && c >= P0 && c < IR);
                const int jq = jposL[jk * N + (okp ? q + (c - P0) : 0)];
#pragma unroll
                for (int jj = 0; jj < CS; jj++) val += (tab.D[ji][jj] * h) * S[(okp && jq >= 0) ? D::w_CJ + jj * D::NZJ + jq : ZERO];
              }
              put((NTH + ct * TJ + jt) * 4 + v, (c < IR && jr < OR) ? col_start(c) + (IR - c) + jr : -1, acc[cq][v] + val);
            }
          }
        }
#if defined(ASSET_TIMING)
        if (g == 1) { tunit[u < NHU ? 0 : 1] += clock64() - tu0; tunit[2] += 1; }
#endif
      }
    }
    TSW();   // tiles of this wave
    wg_lds_barrier();                                    // every wave is done with the slot, DI and the counter
    TSW();   // wait for the other waves
    if (tid == 0) *counter = 0;                          // (visible after the next segment's first barrier)
  }
#if defined(ASSET_TIMING)
  if (blockIdx.x == 7 && lane == 0 && a.FX)
  {
    for (int t = 0; t + 1 < nts; t++) a.FX[size_t(wg_first) * OR + wv * 12 + t] = double(tstamp[t + 1] - tstamp[t]);
    for (int t = 0; t < 3; t++) a.FX[size_t(wg_first) * OR + wv * 12 + 8 + t] = double(tunit[t]);
  }
#endif
#undef TSW
}

// The kernel proper (wide shapes only; an empty kernel for the others, which a run-time compiled module still names).
template <class Ode, int SCH, bool BLOCKED, int LEVEL, bool ASM>
__global__ __launch_bounds__(256, (Dims<Ode, SCH, BLOCKED>::lds_bytes_dense() * ASSET_WIDE_WGS <= 160 * 1024 ? ASSET_WIDE_WGS : 1))
void lgl_wide_dense_kernel(EvalArgs a) {
  if constexpr (Dims<Ode, SCH, BLOCKED>::WIDE && LEVEL >= 1) lgl_wide_dense_body<Ode, SCH, BLOCKED, LEVEL, ASM>(a);
}

}  // namespace asset_hip
