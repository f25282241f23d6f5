// Dense stage for wide shapes, by output ROWS and without matrix instructions (round 3).
//
// On MI355X an f64 MFMA runs at the rate of f64 vector FMAs (78.6 TFLOP/s both), so the matrix form of
//     H = sum_i DI_i^T (hE_i H^_i) DI_i,   J = DC + sum_i hE_i J^_i DI_i                      (LGLDefects.h:414-512)
// buys operand delivery, not arithmetic -- and pays for padded tiles and for ignoring what DI_i is made of:
//     DI_i[:, column (j, cc)] = w_i(j, cc) e_cc + h B_ij dfdy_j[:, cc]   (+ the time-column terms)  (LGLDefects.h:417-458)
// an identity entry and one column of the SPARSE cardinal Jacobian of node j.  With a lane per output row r the products
// collapse (H is symmetric: row r of the stored lower triangle is column r of M):
//     d_i  = DI_i[:, r]                                 the lane's own column: n gathered values, two FMAs each
//     M_i  = hE_i H^_i d_i                              one FMA per structural entry of H^ (uniform operand)
//     WB_j = sum_i B_ij M_i[0:n]                        per node j
//     H(r, (j, cc)) = sum_i w_i(j, cc) M_i[cc] + h sum_k dfdy_j[k][cc] WB_j[k] + time / rank-2 / cardinal terms
// i.e. one FMA per structural entry of column cc of dfdy_j -- four on average for the 32-state BASELINE ODE, n for a dense
// one -- where the matrix form spends K N / 4 = 27 k-steps of a padded 16 x 16 tile.  The operands that do not depend on
// the lane (entries of dfdy_j, H^_i, g^_i) are held lane-distributed in registers -- entry e of a list in lane e % 64 of
// register e / 64 -- and reach the FMAs through v_readlane pairs as scalar operands (scalar loads return out of order and
// share their counter with the LDS, so every use drained both; an LDS broadcast costs the LDS what a full read does and,
// with the compiler issuing them far ahead, the registers the accumulators need: DESIGN 4.4a has the numbers).  J the same
// way with a lane per defect row (i, r): its row of J^_i in registers,
//     J((i, r), (j, cc)) = C_ij [cc == r] + h D_ij dfdy_j[r][cc] + hE_i (w J^_i[r][cc] + h B_ij sum_k J^_i[r][k] dfdy_j[k][cc]) + time terms.
// Stores: for a fixed block column the lanes' rows are contiguous (DenseFunctionBase.h:1112-1123) -- 512-byte runs.
//
// One four-wave workgroup per CU, one segment at a time; the segment's slot is staged in LDS and the next one prefetched while
// the rows are prepared (a global load behind a wave's block stores waits for every one of them); per segment the sparse
// ODE results are scattered into dense, oddly strided LDS arrays (what a lane gathers by its own row / column), then every
// wave runs the row blocks it owns (H rows in equal blocks, defect rows in blocks of 64), dealt at compile time by cost.
// Block entries are written with raw buffer stores: a lane outside the triangle gets an out-of-range offset.
#pragma once
#include "defect_dims.h"
#include "defect_wide.h"
#include "defect_rowdpp.h"
#include <utility>

namespace asset_hip {

#define ASSET_EXP_ROWS 0             // (experiments: bit 0 -- no H columns, bit 1 -- no defect-row columns, bit 2 -- no row preparation)

template <class D>
struct RowsDims {
  using Ode = typename D::ode_t;
  static constexpr int K = D::K, CS = D::CS, n = D::n, q = D::q, N = D::N, IR = D::IR, OR = D::OR;
  static constexpr bool OK_SHAPE = D::WIDE && !D::TRAP;
  static constexpr int WAVES = 4;
  static constexpr int LDK = n | 1, LDH = N | 1, LDJ = N | 1;       // odd strides: a lane per row reads conflict-free
  // (what is read with compile-time addresses -- the staged slots, the short vectors -- first: an LDS instruction's immediate
  //  offset reaches 64 KB, an address beyond that needs a register of its own)
  static constexpr int o_tab = 0;
  static constexpr int o_ST = o_tab + D::TABSZ;                     // [2][WSLOTD] the slot of this segment and of the next one (prefetched)
  static constexpr int o_SB = o_ST + 2 * D::WSLOTD;                 // [K][n]   sum_j B_ij f_j[k]
  static constexpr int o_WL = o_SB + K * n;                         // [CS][n]  sum_i D_ij lam_(i,k)
  static constexpr int o_CL = o_WL + CS * n;                        // [CS][n]  sum_i C_ij lam_(i,k)
  static constexpr int o_SD = o_CL + CS * n;                        // [OR]     sum_j D_ij f_j[r] + E_i f^_i[r]
  static constexpr int o_HT = o_SD + OR;                            // [IR]     full time-partial vector HTpar
  static constexpr int o_X = o_HT + IR;                             // [0]: sum lam sd
  static constexpr int o_Fd = o_X + 2;                              // [CS][N][LDK]   dfdy_j[k][cc]  at ((j N + cc) LDK + k)   (cc >= q: parameter columns)
  static constexpr int o_Hd = o_Fd + CS * N * LDK;                  // [CS][N][LDH]   H_j[a][b]      at ((j N + a) LDH + b), both halves
  static constexpr int o_Jd = o_Hd + CS * N * LDH;                  // [K][n][LDJ]    J^_i[r][a]     at ((i n + r) LDJ + a)
  static constexpr int o_END = o_Jd + K * n * LDJ;
  static constexpr int LDS_DOUBLES = o_END;
  static constexpr size_t lds_bytes() { return size_t(LDS_DOUBLES) * 8; }
  // (what the form is built for: at most four H blocks, the lane-independent operands of a segment in a few lane-distributed
  //  registers -- a dense 32-state Jacobian would take 68 of them and spill; such shapes stay with the tile kernel)
  static constexpr bool OK = OK_SHAPE && lds_bytes() <= 160 * 1024 && N <= 64 && (IR + 63) / 64 <= WAVES &&
                             Ode::NNZ_J <= 256 && Ode::NNZ_H <= 256;
  // row blocks: H blocks 0 .. NHB-1, then defect-row blocks; owner wave by greedy cost (H block b: its last row + 1 column
  // steps, a defect-row block: IR)
  static constexpr int NHB = (IR + 63) / 64, NJB = (OR + 63) / 64, NITEM = NHB + NJB;
  static constexpr int RB = (IR + NHB - 1) / NHB;                   // rows of an H block (equal blocks: the last one, whose rows have
                                                                    //  every column, is the critical one whatever its size)
  static constexpr int item_cost(int it) { return it < NHB ? 4 * (RB * it + RB < IR ? RB * it + RB : IR) : 3 * IR; }   // (a column of H ~ 4/3 of one of J)
  static constexpr int owner(int it) {                              // items are dealt in the order H_(NHB-1) .. H_0, J_0 ..
    int load[WAVES] = {0, 0, 0, 0};
    int own = 0;
    for (int s = 0; s < NITEM; s++) {
      const int cur = s < NHB ? NHB - 1 - s : s;
      int w = 0;
      for (int x = 1; x < WAVES; x++)
        if (load[x] < load[w]) w = x;
      load[w] += item_cost(cur);
      if (cur == it) own = w;
    }
    return own;
  }
};

// position -> (row, column) of the structural entries of dfdy (JPOS) and of the packed lower Hessian (HPOS)
template <class Ode>
struct NzIndex {
  static constexpr int N = Ode::NIN, n = Ode::XV, NZJ = Ode::NNZ_J, NZH = Ode::NNZ_H;
  struct Tab { short jr[NZJ > 0 ? NZJ : 1], jc[NZJ > 0 ? NZJ : 1], ha[NZH > 0 ? NZH : 1], hb[NZH > 0 ? NZH : 1]; };
  static constexpr Tab make() {
    Tab t{};
    for (int k = 0; k < n; k++)
      for (int c = 0; c < N; c++)
        if (Ode::JPOS[k * N + c] >= 0) { t.jr[Ode::JPOS[k * N + c]] = short(k); t.jc[Ode::JPOS[k * N + c]] = short(c); }
    for (int a = 0; a < N; a++)
      for (int b = 0; b <= a; b++)
        if (Ode::HPOS[a * (a + 1) / 2 + b] >= 0) { t.ha[Ode::HPOS[a * (a + 1) / 2 + b]] = short(a); t.hb[Ode::HPOS[a * (a + 1) / 2 + b]] = short(b); }
    return t;
  }
  static constexpr Tab v = make();
};

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<int, N - 1>{}) -- what indexes the sparsity tables has to
// be a constant expression (a run-time index turns the register arrays below into indexed register accesses)
// (by halves: a fold expression over a pack of several hundred indices exceeds the run-time compiler's bracket depth)
template <int B, int E, class F>
__device__ __forceinline__ void static_for_range(F& f) {
  if constexpr (E - B == 1) f(std::integral_constant<int, B>{});
  else if constexpr (E - B > 1) {
    constexpr int M = (B + E) / 2;
    static_for_range<B, M>(f);
    static_for_range<M, E>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_range<0, N>(f); }

// value of lane `l` (compile time) of a double every lane holds one of: two v_readlane, the result a scalar operand.  How the
// lane-independent operands reach the FMAs: a coalesced vector load puts 64 of them into one register, and nothing waits --
// scalar loads return out of order and share their counter with the LDS, so every use of one would drain both
// ... and read WHERE it is used: as volatile assembly with a fake input `after` -- the result of the FMA that precedes the use in
// program order.  As plain builtins the compiler issues every v_readlane of a loop ahead of the FMAs (they have no inputs to wait
// for), runs out of scalar registers and parks the values in the lanes of spare registers -- v_readlane, v_writelane, v_readlane
// per operand: 1 100 of the row preparation's 4 900 instructions (25.7 k -> 16.6 k cycles without them; a dependence on the
// FMA's OTHER operand is not enough: those are all ready before the loop starts).  A wave issues in order anyway.
template <int l>
__device__ __forceinline__ double lane_value_at(double v, double after) {
  int lo, hi;
  asm volatile("v_readlane_b32 %0, %2, %4\n\tv_readlane_b32 %1, %3, %4"
               : "=s"(lo), "=s"(hi) : "v"(__double2loint(v)), "v"(__double2hiint(v)), "n"(l), "v"(after));
  return __hiloint2double(hi, lo);
}

template <class Ode, int SCH, bool BLOCKED, int LEVEL>
__device__ __forceinline__ void lgl_rows_body(const EvalArgs& a, const double* __restrict__ work_ro) {
  using D = Dims<Ode, SCH, BLOCKED>;
  using R = RowsDims<D>;
  using NZ = NzIndex<Ode>;
  constexpr int K = D::K, CS = D::CS, n = D::n, q = D::q, N = D::N, IR = D::IR, OR = D::OR, T = D::T, TF = D::TF;
  constexpr int NZJ = D::NZJ, NZH = D::NZH, LDK = R::LDK, LDH = R::LDH, LDJ = R::LDJ;
  constexpr int p = D::p, P0 = D::P0;
  static_assert(N <= 64, "one register of lane-distributed values per vector of ODE inputs");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  lds_double* const L = (lds_double*)lds;
  const int tid = int(threadIdx.x), wave = tid >> 6, lane = tid & 63;
  constexpr LglTab ctab = c_lgl_tab[D::TAB];                        // compile-time indices: constants in the code
  const LglTab& tab = *reinterpret_cast<const LglTab*>(lds);        // lane-dependent indices: the LDS copy

  // once per workgroup: tables, and zeros where the dense arrays have no structural entry
  for (int e = tid; e < D::TABSZ; e += 256) L[R::o_tab + e] = reinterpret_cast<const double*>(&d_lgl_tab[D::TAB])[e];
  for (int e = tid; e < R::o_END - R::o_Fd; e += 256) L[R::o_Fd + e] = 0.0;

  const int nshare = int(gridDim.x), share = int(blockIdx.x);
  const int per = a.nseg / nshare, rem = a.nseg % nshare;
  const int wg_first = share * per + min(share, rem), wg_count = per + (share < rem ? 1 : 0);
  if (wg_count > 0)                                                 // the first slot: everybody; the later ones: one wave, a segment ahead
    for (int e = tid; e < D::WSLOTD; e += 256) L[R::o_ST + e] = work_ro[size_t(wg_first) * D::WSLOT + e];

  for (int s = 0; s < wg_count; s++) {
    const size_t seg = size_t(wg_first + s);
    // the segment's slot, in LDS since the previous segment's row preparation: nothing below waits for global memory (a load
    // behind this wave's block stores would wait for every one of them), and what every lane reads the same entry of --
    // dfdy_j, H^_i, g^_i -- is an LDS broadcast: ~5 per block column, a third of the LDS's time with four waves at it
    const lds_double* const ws = L + R::o_ST + (s & 1) * D::WSLOTD;
    double* const kkt = a.KKT ? a.KKT + seg * size_t(D::NKKT) : nullptr;
#if defined(ASSET_TIMING)
    long long rts[8]; int nrt = 0;
#define RWTS() do { if (nrt < 8) rts[nrt++] = clock64(); } while (0)
#else
#define RWTS() do {} while (0)
#endif
    RWTS();
    wg_lds_barrier();                                               // (the previous segment's readers are done; the first pass: the zeros)
    RWTS();
    // ---- S0: scatter the slot's sparse blocks into the dense arrays; the short vectors
    for (int e = tid; e < CS * NZJ; e += 256) {
      const int j = e / NZJ, pz = e - j * NZJ;
      L[R::o_Fd + (j * N + NZ::v.jc[pz]) * LDK + NZ::v.jr[pz]] = ws[D::w_CJ + e];
    }
    for (int e = tid; LEVEL >= 2 && e < CS * NZH; e += 256) {
      const int j = e / NZH, pz = e - j * NZH;
      const double v = ws[D::w_CH + e];
      L[R::o_Hd + (j * N + NZ::v.ha[pz]) * LDH + NZ::v.hb[pz]] = v;
      L[R::o_Hd + (j * N + NZ::v.hb[pz]) * LDH + NZ::v.ha[pz]] = v;
    }
    for (int e = tid; e < K * NZJ; e += 256) {
      const int i = e / NZJ, pz = e - i * NZJ;
      L[R::o_Jd + (i * n + NZ::v.jr[pz]) * LDJ + NZ::v.jc[pz]] = ws[D::w_IJ + e];
    }
    const double h = ws[D::w_z + TF] - ws[D::w_z + T];
    for (int e = tid; e < K * n; e += 256) {                        // SB[i][k]
      const int i = e / n, k = e - i * n;
      double sb = 0.0;
#pragma unroll
      for (int j = 0; j < CS; j++) sb = fma(tab.B[i][j], ws[D::w_Cf + j * n + k], sb);
      L[R::o_SB + e] = sb;
    }
    for (int e = tid; e < CS * n; e += 256) {                       // WL[j][k], CL[j][k]
      const int j = e / n, k = e - j * n;
      double wl = 0.0, cl = 0.0;
#pragma unroll
      for (int i = 0; i < K; i++) {
        const double l = a.L ? ws[D::w_lam + i * n + k] : 0.0;
        wl = fma(tab.D[i][j], l, wl);
        cl = fma(tab.C[i][j], l, cl);
      }
      L[R::o_WL + e] = wl;
      L[R::o_CL + e] = cl;
    }
    if (wave == R::WAVES - 1) {                                     // sd, the defect values, sum lam sd
      double lsum = 0.0;
      for (int jr = lane; jr < OR; jr += 64) {
        const int i = jr / n, r = jr - i * n;
        double sd = tab.E[i] * ws[D::w_If + jr], fx = 0.0;
#pragma unroll
        for (int j = 0; j < CS; j++) {
          sd = fma(tab.D[i][j], ws[D::w_Cf + j * n + r], sd);
          fx = fma(tab.C[i][j], ws[D::w_z + j * q + r], fx);
        }
        L[R::o_SD + jr] = sd;
        if (a.FX) a.FX[seg * OR + jr] = fma(h, sd, fx);
        lsum = fma(a.L ? ws[D::w_lam + jr] : 0.0, sd, lsum);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o);
      if (lane == 0) L[R::o_X] = lsum;
    }
    RWTS();
    wg_lds_barrier();
    RWTS();

    // ---- the next segment's slot -> the other staging buffer: the last wave, while the others prepare their rows (it has no H
    //      block when there are fewer than WAVES of them, and its block stores of the previous segment are long acknowledged)
    if ((LEVEL < 2 || wave == R::WAVES - 1) && s + 1 < wg_count) {  // (Jacobian kinds: no long row preparation to hide it behind -- a quarter per wave)
      constexpr int CH = 8;
      constexpr int NW = LEVEL < 2 ? R::WAVES : 1;
      const int pl = LEVEL < 2 ? tid : lane;
      const double* __restrict__ nxt = work_ro + (seg + 1) * D::WSLOT;
      lds_double* const dst = L + R::o_ST + ((s + 1) & 1) * D::WSLOTD;
      for (int e0 = 0; e0 < D::WSLOTD; e0 += 64 * NW * CH) {
        double v[CH];
#pragma unroll
        for (int t = 0; t < CH; t++) v[t] = (e0 + 64 * NW * t + pl < D::WSLOTD) ? nxt[e0 + 64 * NW * t + pl] : 0.0;
#pragma unroll
        for (int t = 0; t < CH; t++)
          if (e0 + 64 * NW * t + pl < D::WSLOTD) dst[e0 + 64 * NW * t + pl] = v[t];
      }
    }
    // ---- the row blocks of this wave
    // (row preparation: the lane-independent operands H^_i, g^_i, SB_i as lane-distributed registers + v_readlane -- entry e of a
    //  list in lane e % 64 of register e / 64; as LDS broadcasts the compiler issues them far ahead and spills the accumulators)
    // (round 5) the lane-independent operands sixteen to a register, entry e in lane e % 16 of EVERY 16-lane row, and broadcast
    // inside the FMA: v_fmac_f64_dpp row_newbcast (defect_rowdpp.h: fmac_bc) -- where rounds 3-4 read them with a v_readlane pair
    // and a hazard s_nop per use.  They are loaded where they are used (a node's dfdy_j list ahead of its columns, an interior's
    // H^_i / g^_i / SB_i lists ahead of its products): five to eight registers live at a time.
    constexpr int NH16 = (NZH + 15) / 16, NF16 = (NZJ + 15) / 16, NG16 = (N + 15) / 16, NS16 = (n + 15) / 16;
    const int l16 = lane & 15;
    auto load_F = [&](double (&Fr)[NF16], int j) {
#pragma unroll
      for (int t = 0; t < NF16; t++) Fr[t] = ws[D::w_CJ + j * NZJ + 16 * t + l16];     // (past the list: the next section's entries, never broadcast)
    };
    auto load_S = [&](double (&sr)[NS16], int i) {
#pragma unroll
      for (int t = 0; t < NS16; t++) sr[t] = L[R::o_SB + i * n + 16 * t + l16];
    };
    double Mi[K][N];                                                // H blocks: M_i[:, r]
    double TX = 0.0, HTr = 0.0;
    int hitem = -1;                                                 // (a wave owns at most one H block: NHB <= WAVES)
#pragma unroll
    for (int it = 0; it < R::NHB; it++)
      if (R::owner(it) == wave) hitem = it;
    if (hitem >= 0 && !(ASSET_EXP_ROWS & 4) && (LEVEL >= 2 || (a.AGX && a.L))) {   // (Jacobian kinds: only the adjoint gradient needs the rows)
      const int it = hitem;                                         // (run time: one copy of the code for every block)
      const int r = R::RB * it + lane;
      const bool rv = lane < R::RB && r < IR;
      const int rc = rv ? r : IR - 1;
      // a node's row (jn, ccr), or (segment parameters) a parameter row: component ccr = q + pa of EVERY node
      const bool par = p > 0 && rc >= P0;
      const int jn = par ? 0 : rc / q, ccr = par ? q + (rc - P0) : rc - jn * q;
      const bool pblock = p > 0 && R::RB * it + R::RB > P0;          // (uniform) the block holds parameter rows
      const double tsr = !rv ? 0.0 : ((r == T) ? -1.0 : ((r == TF) ? 1.0 : 0.0));
      const lds_double* const fcol = L + R::o_Fd + (jn * N + ccr) * LDK;   // the row's own column of dfdy (read where used: registers)
      double ht = 0.0;
#pragma unroll
      for (int i = 0; i < K; i++) {
        double d[N];
        const double hb = par ? 0.0 : h * tab.B[i][jn], ar = par ? 0.0 : tab.A[i][jn], ur = par ? 0.0 : tab.U[i][jn];
#pragma unroll
        for (int k = 0; k < n; k++) d[k] = fma(hb, fcol[k], (k == ccr) ? ar : 0.0);
        if constexpr (p > 0) {
          if (pblock) {                                             // parameter rows: h sum_j B_ij dfdy_j[:, q + pa] (LGLDefects.h:440-444)
#pragma unroll
            for (int j = 0; j < CS; j++) {
              const double hbj = par ? h * ctab.B[i][j] : 0.0;
#pragma unroll
              for (int k = 0; k < n; k++) d[k] = fma(hbj, L[R::o_Fd + (j * N + ccr) * LDK + k], d[k]);
            }
          }
#pragma unroll
          for (int pb = 0; pb < p; pb++) d[q + pb] = (par && ccr == q + pb) ? 1.0 : 0.0;
        }
        if ((R::RB * it <= T && T < R::RB * it + R::RB) || (R::RB * it <= TF && TF < R::RB * it + R::RB)) {   // a time row: -+ sum_j B_ij f_j on it
          double sr[NS16];
          load_S(sr, i);
          static_for<n>([&](auto KK) { constexpr int k = decltype(KK)::value; fmac_bc<(k & 15)>(d[k], sr[k >> 4], tsr); });
        }
        d[T] = !rv ? 0.0 : ((r == T) ? 1.0 - ctab.s[i] : ((r == TF) ? ctab.s[i] : 0.0));   // (T, TF < P0: never a parameter row)
#pragma unroll
        for (int cc = n + 1; cc < q; cc++) d[cc] = (!par && ccr == cc) ? ur : 0.0;
        {
          double gr[NG16], hti[2] = {0.0, 0.0};                     // (two partial sums: a dependent v_fmac_f64 issues every 9 cycles)
#pragma unroll
          for (int t = 0; t < NG16; t++) gr[t] = ws[D::w_Ig + i * N + 16 * t + l16];
          static_for<N>([&](auto BB) { constexpr int b = decltype(BB)::value; fmac_bc<(b & 15)>(hti[b & 1], gr[b >> 4], d[b]); });
          ht = fma(ctab.E[i], hti[0] + hti[1], ht);
        }
        if constexpr (LEVEL >= 2) {
#pragma unroll
        for (int b = 0; b < N; b++) Mi[i][b] = 0.0;
        double Hr[NH16];
#pragma unroll
        for (int t = 0; t < NH16; t++) Hr[t] = ws[D::w_IH + i * NZH + 16 * t + l16];
        static_for<N*(N + 1) / 2>([&](auto E) {
          constexpr int e = decltype(E)::value;
          constexpr int hp = Ode::HPOS[e];
          if constexpr (hp >= 0) {
            constexpr int b = NZ::v.ha[hp], l = NZ::v.hb[hp];
            fmac_bc<(hp & 15)>(Mi[i][b], Hr[hp >> 4], d[l]);
            if constexpr (b != l) fmac_bc<(hp & 15)>(Mi[i][l], Hr[hp >> 4], d[b]);
          }
        });
        const double he = h * ctab.E[i];
#pragma unroll
        for (int b = 0; b < N; b++) Mi[i][b] *= he;
        }
      }
      // full time partial of the row (LGLDefects.h:403-411, 504-505), the adjoint gradient of its column (:512)
      if constexpr (LEVEL >= 2) {
        double gs = ws[D::w_Cg + jn * N + ccr];
        if constexpr (p > 0) {
          if (pblock) {                                             // a parameter row: g_j[q + pa] summed over the nodes
#pragma unroll
            for (int j = 1; j < CS; j++) gs += par ? ws[D::w_Cg + j * N + ccr] : 0.0;
          }
        }
        HTr = ht + gs / h;
        if (rv) L[R::o_HT + r] = HTr;
      }
      if constexpr (LEVEL >= 2) {
#pragma unroll
      for (int i = 0; i < K; i++) {
        double sr[NS16], txp[2] = {0.0, 0.0};
        load_S(sr, i);
        static_for<n>([&](auto KK) { constexpr int k = decltype(KK)::value; fmac_bc<(k & 15)>(txp[k & 1], sr[k >> 4], Mi[i][k]); });
        TX += txp[0] + txp[1];
      }
      }
      if (a.AGX && a.L && rv) {
        double wl = 0.0;
#pragma unroll
        for (int k = 0; k < n; k++) wl = fma(L[R::o_WL + jn * n + k], fcol[k], wl);
        if constexpr (p > 0) {
          if (pblock) {
#pragma unroll
            for (int j = 1; j < CS; j++)
#pragma unroll
              for (int k = 0; k < n; k++) wl = fma(par ? L[R::o_WL + j * n + k] : 0.0, L[R::o_Fd + (j * N + ccr) * LDK + k], wl);
          }
        }
        const double cl = (ccr < n) ? L[R::o_CL + jn * n + (ccr < n ? ccr : 0)] : 0.0;
        a.AGX[seg * IR + r] = fma(h, ht + wl, fma(tsr, L[R::o_X], cl));
      }
    }
    double jrow[N], he_l = 0.0, sd_l = 0.0, TXJ = 0.0, Al[CS], Bl[CS], Cl[CS], Dl[CS], Ul[CS], s_l = 0.0;
    int jitem = -1;
    // (the defect-row blocks of this wave start after the barrier: their loop reads nothing the H blocks write, but a wave
    //  may own both kinds)
    RWTS();
    wg_lds_barrier();                                               // HT complete
    RWTS();

    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(kkt, 0, int(D::NKKT * 8), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;                           // an offset beyond the block: the store is dropped (no exec masking)
    if (LEVEL < 2 && hitem >= 0 && kkt && !(a.flags & 1)) {         // Jacobian kinds: the Hessian slots hold zeros, unless the caller never reads them
      const int r0 = R::RB * hitem + lane;
      const bool rv = lane < R::RB && r0 < IR;
      const int rmax = min(R::RB * hitem + R::RB - 1, IR - 1);
      int r = rv ? r0 : -1;
      u2 zero2;
      zero2.x = 0u; zero2.y = 0u;
      static_for<IR>([&](auto Ct) {
        constexpr int c = decltype(Ct)::value;
        if constexpr ((c & 15) == 0) asm volatile("" : "+v"(r));     // (the row >= column masks where they are used)
        if (c <= rmax)
          __builtin_amdgcn_raw_buffer_store_b64(zero2, rsrc, (r >= c) ? unsigned(r) * 8u : OOB, (c * (IR + OR - 1) - ((c * (c - 1)) >> 1)) * 8, 0);
      });
    }
    if (LEVEL >= 2 && hitem >= 0 && kkt && !(ASSET_EXP_ROWS & 1)) {
      const int r0 = R::RB * hitem + lane;
      const bool rv = lane < R::RB && r0 < IR;
      const int rc = rv ? r0 : IR - 1;
      const bool par = p > 0 && rc >= P0;                           // a parameter row: component ccr = q + pa of every node
      const int jn = par ? -1 : rc / q, ccr = par ? q + (rc - P0) : rc - (rc / q) * q;
      const int rmax = min(R::RB * hitem + R::RB - 1, IR - 1);
      const double tsrow = !rv ? 0.0 : ((r0 == T) ? -1.0 : ((r0 == TF) ? 1.0 : 0.0));   // rows T / TF: the transposed rank-2 rows -+ HTpar[c]
      const double txh = TX + HTr;
      int r = -1;
      unsigned voff = 0;
      double indj = 0.0;
      int hbase = R::o_Hd + ccr;                                    // (opaque: the column's offset then fits the instruction's immediate
      asm volatile("" : "+v"(hbase));                             //  field instead of taking an address register per column)
      double hcur = L[hbase];                                       // H_j[cc][ccr] of the column ahead: requested one column early
      // the block row (j N + cc) of the dense arrays that block column c reads: node columns (j, cc), parameter columns (0, q + pc)
      auto brow = [](int c) constexpr { return c < P0 ? (c / q) * N + (c % q) : q + (c - P0); };
      double Fr[NF16], htr = 0.0;                                   // the node's dfdy_j list; HTpar of sixteen columns
      // BM_j[k] = h sum_i B_ij M_i[k], formed when the loop reaches node j: a column of the node is then ONE FMA per structural entry
      // of its dfdy column instead of K (and K more to combine them) -- K n FMAs per node bought, ~ 2 K nnz(dfdy_j) saved
      // -- OFF: its n doubles are the registers the kernel does not have (512 of 512 and 340 bytes of scratch per lane instead of 482
      // and none; a scratch reload behind block stores waits for them all: 12 500 32-state segments 1.78 ms against 1.08 ms)
#define ASSET_ROWS_NODE_BM 0
      double BMn[ASSET_ROWS_NODE_BM ? n : 1];
      static_for<IR>([&](auto Ct) {
        constexpr int c = decltype(Ct)::value;
        constexpr bool pcol = c >= P0;                              // a parameter column (segment parameters)
        constexpr int j = pcol ? 0 : c / q, cc = pcol ? q + (c - P0) : c % q, CC = cc;
        if (c <= rmax) {                                            // (uniform)
          if constexpr ((c & 15) == 0) htr = L[R::o_HT + c + l16];
          if constexpr (cc == 0 && !pcol) {
            load_F(Fr, j);
            if constexpr (ASSET_ROWS_NODE_BM) {
              static_for<n>([&](auto KK) {
                constexpr int k = decltype(KK)::value;
                double b = 0.0;
#pragma unroll
                for (int i = 0; i < K; i++) b = fma(ctab.B[i][j], Mi[i][k], b);
                BMn[k] = h * b;
              });
            }
          }
          if constexpr (cc == 0 || c == P0) {                       // a new node (or the parameter columns)
            r = rv ? r0 : -1;                                       // (opaque per node: the row >= column masks are formed where they
            asm volatile("" : "+v"(r));                             //  are used, not all of them ahead of the loop)
            voff = unsigned(r) * 8u;
            indj = (par || jn == j) ? 1.0 : 0.0;                    // rows of node j take its cardinal block H_j; parameter rows every node's
          }
          double val = 0.0;
#pragma unroll
          for (int i = 0; i < K; i++) {
            const double w = pcol ? 1.0 : (cc < n ? ctab.A[i][j] : (cc == T ? (c == T ? 1.0 - ctab.s[i] : (c == TF ? ctab.s[i] : 0.0)) : ctab.U[i][j]));
            if (w != 0.0) val = fma(w, Mi[i][cc], val);
          }
          // h sum_i B_ij (dfdy_j[:, cc] . M_i); a parameter column: summed over the nodes
          double acc = 0.0;
          if constexpr (ASSET_ROWS_NODE_BM && !pcol) {
            double a2[2] = {0.0, 0.0};
            int cnt = 0;
            static_for<n>([&](auto KK) {
              constexpr int k = decltype(KK)::value;
              constexpr int jp = Ode::JPOS[k * N + CC];
              if constexpr (jp >= 0) { fmac_bc<(jp & 15)>(a2[cnt & 1], Fr[jp >> 4], BMn[k]); cnt++; }
            });
            val += a2[0] + a2[1];
          } else
          static_for<(pcol ? CS : 1)>([&](auto Jt) {
            constexpr int jj = pcol ? decltype(Jt)::value : j;
            double acci[K];
#pragma unroll
            for (int i = 0; i < K; i++) acci[i] = 0.0;
            double Fp[pcol ? NF16 : 1];
            if constexpr (pcol) load_F(Fp, jj);
            static_for<n>([&](auto KK) {
              constexpr int k = decltype(KK)::value;
              constexpr int jp = Ode::JPOS[k * N + CC];
              if constexpr (jp >= 0) {
                static_for<K>([&](auto II) {
                  constexpr int i = decltype(II)::value;
                  if constexpr (pcol) fmac_bc<(jp & 15)>(acci[i], Fp[jp >> 4], Mi[i][k]);
                  else fmac_bc<(jp & 15)>(acci[i], Fr[jp >> 4], Mi[i][k]);
                });
              }
            });
#pragma unroll
            for (int i = 0; i < K; i++) acc = fma(ctab.B[i][jj], acci[i], acc);
          });
          val = fma(h, acc, val);
          if constexpr (c == T) val -= txh;
          if constexpr (c == TF) val += txh;
          fmac_bc<(c & 15)>(val, htr, tsrow);
          const double hc = hcur;
          if constexpr (c + 1 < IR) hcur = L[hbase + brow(c + 1) * LDH];
          if constexpr (!pcol) val = fma(indj, hc, val);
          else {                                                    // parameter-parameter entries: H_j[q + pc][q + pa] summed over the nodes
            double hs = hc;
#pragma unroll
            for (int jj = 1; jj < CS; jj++) hs += L[hbase + (jj * N + cc) * LDH];
            val += par ? hs : 0.0;
          }
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, val), rsrc, (r >= c) ? voff : OOB,
                                                (c * (IR + OR - 1) - ((c * (c - 1)) >> 1)) * 8, 0);
        }
      });
    }

    RWTS();
    // (Jacobian kinds: the H blocks cost next to nothing, so every defect-row block is split in two column halves and the four
    //  halves dealt round robin)
    constexpr int JH = LEVEL >= 2 ? 1 : 2;                          // column ranges per defect-row block
    for (int jt = 0; jt < R::NJB * JH; jt++) {
      const int jb = jt / JH, half = jt - jb * JH;
      bool mine = false;
#pragma unroll
      for (int x = 0; x < R::NJB; x++) mine = mine || (x == jb && (LEVEL >= 2 ? R::owner(R::NHB + x) : jt % R::WAVES) == wave);
      if (!mine) continue;
      const int cfirst = half * (IR / JH), clast = (half + 1 == JH) ? IR : (half + 1) * (IR / JH);
      jitem = jb;
      const int jr = 64 * jb + lane;
      const bool jv = jr < OR;
      const int jc = jv ? jr : OR - 1;
      const int i = jc / n, r = jc - i * n;
#pragma unroll
      for (int b = 0; b < N; b++) jrow[b] = L[R::o_Jd + (i * n + r) * LDJ + b];
#pragma unroll
      for (int j = 0; j < CS; j++) { Al[j] = tab.A[i][j]; Bl[j] = tab.B[i][j]; Cl[j] = tab.C[i][j]; Dl[j] = tab.D[i][j]; Ul[j] = tab.U[i][j]; }
      s_l = tab.s[i];
      he_l = h * tab.E[i];
      sd_l = L[R::o_SD + jc];
      TXJ = 0.0;
#pragma unroll
      for (int k = 0; k < n; k++) TXJ = fma(jrow[k], L[R::o_SB + i * n + k], TXJ);
      if (!kkt || (ASSET_EXP_ROWS & 2)) continue;
      int ro = r;
      int fbase = R::o_Fd + r;
      asm volatile("" : "+v"(fbase));
      double fcur = L[fbase];                                       // dfdy_j[r][cc] of the column ahead
      auto brow = [](int c) constexpr { return c < P0 ? (c / q) * N + (c % q) : q + (c - P0); };   // (as for the H blocks)
      double Fr[NF16];
      static_for<IR>([&](auto Ct) {
        constexpr int c = decltype(Ct)::value;
        constexpr bool pcol = c >= P0;
        constexpr int j = pcol ? 0 : c / q, cc = pcol ? q + (c - P0) : c % q, CC = cc;
        if constexpr (cc == 0 && !pcol) {
          if (JH == 1 || (c + q > cfirst && c < clast)) load_F(Fr, j);   // (uniform: the node has columns in this wave's range)
        }
        if constexpr (cc == 0) {                                    // (opaque per node: [cc == r] is formed where it is used)
          ro = r;
          asm volatile("" : "+v"(ro));
        }
        const double fd = fcur;
        if constexpr (c + 1 < IR) fcur = L[fbase + brow(c + 1) * LDK];
        if (JH == 1 || (c >= cfirst && c < clast)) {                // (uniform)
        const double w = pcol ? 1.0 : (cc < n ? Al[j] : (cc == T ? (c == T ? 1.0 - s_l : (c == TF ? s_l : 0.0)) : Ul[j]));
        double acc = 0.0, fdd = h * Dl[j] * fd;                     // h B_il,j (J^_i[r][:] . dfdy_j[:, cc]),  h D_il,j dfdy_j[r][cc]
        static_for<(pcol ? CS : 1)>([&](auto Jt) {                  // (a parameter column: summed over the nodes)
          constexpr int jj = pcol ? decltype(Jt)::value : j;
          double a1 = 0.0;
          double Fp[pcol ? NF16 : 1];
          if constexpr (pcol) load_F(Fp, jj);
          static_for<n>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            constexpr int jp = Ode::JPOS[k * N + CC];
            if constexpr (jp >= 0) {
              if constexpr (pcol) fmac_bc<(jp & 15)>(a1, Fp[jp >> 4], jrow[k]);
              else fmac_bc<(jp & 15)>(a1, Fr[jp >> 4], jrow[k]);
            }
          });
          acc = fma(Bl[jj], a1, acc);
          if constexpr (pcol && jj > 0) fdd = fma(h * Dl[jj], L[fbase + (jj * N + cc) * LDK], fdd);
        });
        constexpr double tsc = (c == T) ? -1.0 : ((c == TF) ? 1.0 : 0.0);
        double val = he_l * fma(w, jrow[cc], fma(h, acc, tsc * TXJ));
        val += fdd;
        if constexpr (cc < n) val += (ro == cc) ? Cl[j] : 0.0;
        val = fma(tsc, sd_l, val);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, val), rsrc, jv ? unsigned(jr) * 8u : OOB,
                                              (c * (IR + OR - 1) - ((c * (c - 1)) >> 1) + IR) * 8, 0);
        }
      });
    }
    (void)jitem;
    RWTS();
#if defined(ASSET_TIMING)
    if (blockIdx.x == 7 && s == 2 && lane == 0 && a.FX)
      for (int t = 0; t + 1 < nrt; t++) a.FX[seg * OR + wave * 8 + t] = double(rts[t + 1] - rts[t]);
#endif
#undef RWTS
  }
}

template <class Ode, int SCH, bool BLOCKED, int LEVEL = 2>
__global__ __launch_bounds__(256, 1) void lgl_rows_kernel(EvalArgs a, const double* __restrict__ work_ro) {
  if constexpr (RowsDims<Dims<Ode, SCH, BLOCKED>>::OK) lgl_rows_body<Ode, SCH, BLOCKED, LEVEL>(a, work_ro);
}

}  // namespace asset_hip
