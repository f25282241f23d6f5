// The RESIDENT form of one evaluation (value + Jacobian + adjoint gradient + adjoint Hessian) of narrow LGL shapes:
// one launch, and the ODE results of a wave's segments never leave the chip.
//
// What the fused kernel of defect_kernels.h (STAGE 3) still pays, measured on the north-star phase (10 000 Reentry-LGL7
// segments): the per-segment slot of ODE results (460 doubles) is written to the workspace in HBM by the ODE stage and read
// back by the dense stage -- 37 MB each way beside 88 MB of blocks, plus the copy-out phases of the ODE stage (12 k of its
// 32 k cycles).  The slots did not fit the 20 KiB of LDS a wave has at 8 waves per CU because the dense stage needs 14 KiB
// of scratch tiles of its own (DI_i state rows / M^T, the constant rows of DI_i, the DC tile).  This kernel needs none of them:
//   * the fragments of DI_i = d(x^_i, tau_i, u^_i)/dz are built in registers straight from the slot (one LDS read per
//     fragment element and tile, shared by the K interiors: DI_i[b][c] = A_ij [cc == b] + h B_ij J_j[b][cc]);
//   * M_i = (h E_i H^_i) DI_i is formed with H^_i as the A operand, so that accumulator entry v of a column tile IS the B
//     operand of k-step v of  H += DI_i^T M_i  (the trick of defect_wide.h): M never goes through LDS;
//   * the cardinal part of J (DC) is formed per accumulator entry from the slot, as the initial value of the J^T tiles;
//   * J^T lam is taken column by column in closed form (no pass over a DC tile).
// LDS of a wave: weight tables, GR slots, a few short vectors -- GR = 5 Reentry-LGL7 segments in 19.7 KiB.  The ODE stage
// (lane <-> evaluation point, one pass per phase, generated bodies out of line) writes its results into the slots and
// nowhere else; the only global traffic of the kernel is the gather of z / lam, the per-lane constant record and the
// result blocks.  Reference: /root/reference/src/OptimalControl/LGLDefects.h:289-551 (phase order :341-412, the products
// :414-506, the time rows / columns :508-511, adjgrad :512); slot order of the blocks DenseFunctionBase.h:1112-1123.
//
// Shapes: LGL3/5/7 and (round 4) Trapezoidal -- segment parameters (ODE parameters, BlockConstant controls) included --, N + 1 <= 16,
// defect rows in at most two 16-row tiles (two tiles: built for one wave per SIMD, ResDims::WPS).  Every mesh size: one group of
// at most GR segments per wave in the one-group kernel -- since round 4 two-wave workgroups that share the ODE stage (ResDims::PAIR) --
// and the looped instantiation beyond.  Heavy right-hand sides (defect_units.h) take the dense part alone (GIVEN).  Wide shapes
// (IR >= 64) take defect_rows.h / defect_wide.h.
#pragma once
#include "defect_kernels.h"

namespace asset_hip {

template <class D>
struct ResDims {
  using Ode = typename D::ode_t;
  static constexpr int K = D::K, CS = D::CS, n = D::n, N = D::N, q = D::q, IR = D::IR, OR = D::OR, IRP = D::IRP;
  // interior points that are evaluated: none for Trapezoidal (TrapezoidalDefects.h:263-435 has no interior point; Dims carries
  // one of weight E = 0 so that the sizes and the tables keep their shape) -- its sections of the slot are never written, and
  // nothing below may read them
  static constexpr int KE = D::TRAP ? 0 : K;
  // slot of one segment (doubles): the sections of Dims (z | lam | Cf | CJ | Cg | CH | If | IJ | Ig | IH), a cell that always
  // holds 0.0 (target of every "no entry" offset), then the saved transcendentals of the cardinal nodes -- which alias the Cg
  // section when they fit (P1 writes them, P3 reads its own into registers before it writes g_j there)
  static constexpr bool SV_ALIAS = Ode::NSAVE <= N;
  static constexpr int s_Z0 = D::WSLOTD;
  static constexpr int s_SV = SV_ALIAS ? D::w_Cg : D::WSLOTD + 1;
  static constexpr int SV_LD = SV_ALIAS ? N : Ode::NSAVE;
  // (round 5) shapes that take the row-wise dense part (RD_SHAPE, below) are built for two waves per SIMD whatever their number of row
  // tiles: that part needs ~ 250 registers at any width.  Their ASSEMBLED kinds, which keep the tile form, would spill at 256 registers
  // (650 bytes per lane, TwoBody-LGL7) and go to the fused / two-launch kernels instead (ASM_OK, MF_RES_ASM).
#ifndef ASSET_RES_ROWDPP
#define ASSET_RES_ROWDPP 1
#endif
  static constexpr bool RD_SHAPE = ASSET_RES_ROWDPP == 2 || (ASSET_RES_ROWDPP == 1 && !((q % 4 == 0) && D::p == 0));
#define ASSET_RES_WPS ((Ode::NUNITS > 1 || (D::TJ > 1 && !(RD_SHAPE && !D::TRAP))) ? 1 : 2)
  static constexpr int WPS = ASSET_RES_WPS;
  static constexpr bool ASM_OK = !(D::TJ > 1 && RD_SHAPE && !D::TRAP && Ode::NUNITS == 1);
#ifndef ASSET_RES_PAIR
#define ASSET_RES_PAIR 1
#endif
  // ROWDPP (round 5, defect_rowdpp.h): the dense part of the one-group pair kernel by output rows, sixteen lanes to a row group, the
  // lane-independent operands broadcast inside the FMA (v_fmac_f64_dpp row_newbcast) -- no matrix instructions.  Its time rows need
  // FB_i[a] = sum_j B_ij f_j[a], which the interior phase leaves in a section of its own behind the saved values.
  // Which shapes take it -- measured per shape, 10 000 segments, both forms built from the same tree (tools/r5_shapes.sh; us, row-wise /
  // tile): TwoBody-LGL3 19.1 / 22.1, BlockConstant 18.5 / 30.2; TwoBody-LGL5 26.3 / 30.8, BlockConstant 24.0 / 33.1; Brachistochrone-LGL7
  // 17.7 / 25.6, -LGL5 BlockConstant 11.7 / 16.0; Reentry-LGL5 BlockConstant 22.7 / 27.2, -LGL7 BlockConstant 29.5 / 33.5 -- and
  // Reentry without segment parameters 15.4 / 15.3 (LGL3), 24.3 / 23.4 (LGL5), 31.5 / 28.4 (LGL7): the shapes whose node stride is a
  // multiple of four and that have no parameter columns (ResLane::QFAST) fill their 16-column tiles without padding and keep their row
  // weights in registers; there the tile form issues 500 instructions and 33 matrix instructions per segment against ~ 800 vector
  // instructions of the row-wise form, and both end at the drain of the block stores.  ASSET_RES_ROWDPP: 0 never, 1 by that rule, 2 always.
  static constexpr bool RD_CAP = ASSET_RES_PAIR && !D::TRAP && WPS == 2 && Ode::NUNITS == 1;   // the shape CAN run the row-wise part
  static constexpr bool ROWDPP = RD_SHAPE && RD_CAP;                  // ... and it is its default form
  // (round 6) BOTH forms for the shapes that default to tiles (Reentry without segment parameters): with the J | H block layout the
  // row-wise part wins where a workgroup has enough segments to fill its passes and on the large meshes -- Reentry-LGL7 x 7 500 22.8
  // against 24.8 us, x 10 000 27.5 / 28.5, x 1 000 000 2.80 / 2.89 ms; LGL5 x 100 000 205 / 224 us -- and loses on small meshes (LGL3 x 1 000
  // 9.75 / 9.12, LGL7 x 5 000 21.0 / 20.6) and where a wave walks two or three groups (LGL7 x 30 000 96 / 88): profiles/r6_forms.txt.
  // The launcher picks by mesh size (registry.h: K_RES_ALT, K_RESLP).
  static constexpr bool RD_ALT = !RD_SHAPE && RD_CAP && ASSET_RES_ROWDPP == 1;
  static constexpr bool RD_ANY = ROWDPP || RD_ALT;
  static constexpr int s_FB = D::WSLOTD + 1 + (SV_ALIAS ? 0 : CS * Ode::NSAVE);
  static constexpr int SLOT = (s_FB + (RD_ANY ? K * n + 1 : 0)) | 1;   // odd: conflict-free across segments
  // JRIDE: the rows behind the H^ rows of the A operand of the M product carry h E_i J^_i, so the interior part of J comes out of
  // the same matrix instructions (as J_i[r][c] in the lanes of column c: transposed with respect to the store order).  It is
  // turned through LDS, one 16-column tile and interior at a time, in buffers T_i [16][n] laid over sections of the segment's
  // own slot that are dead by then: CJ (read into registers / the DC values before) and [If | IJ | Ig | IH] (all in the
  // A operands and the row sums by then).  Buffers are placed greedily in the slot; shapes whose slot has no room (TwoBody: 15
  // structural entries of J^) keep them in the wave's scratch behind the slots instead (T_XTRA).
  static constexpr int TB = 16 * n;
  static constexpr int deadA0 = D::w_CJ, deadA1 = D::w_CJ + CS * D::NZJ, deadB0 = D::w_If, deadB1 = D::WSLOTD;
  static constexpr int t_off_slot(int i) {         // slot offset of T_i, or -1
    int a = deadA0, b = deadB0;
    for (int k = 0; k <= i; k++) {
      if (a + TB <= deadA1) { if (k == i) return a; a += TB; }
      else if (b + TB <= deadB1) { if (k == i) return b; b += TB; }
      else return -1;
    }
    return -1;
  }
#define ASSET_RES_JRIDE 1
  // GROW: row N of the A operand carries E_i g^_i, so that sum_i E_i g^_i . DI_i falls out of the M product too.  Shapes where the
  // H^ rows, that row and the J^ rows do not fit 16 rows together but the first and the last do (TwoBody: N = 10, n = 6) give the
  // g^ row up -- the sum is then K KS vector FMAs and two cross-lane adds per column tile -- and keep the ride: 12 of 45 matrix
  // instructions per TwoBody-LGL5 segment gone.
#define ASSET_RES_GROWLESS 1
  // (only where the wave has a SIMD's registers to itself: at two waves per SIMD the TwoBody-LGL5-BlockConstant kernel, which
  //  then keeps its J^T accumulators through the H tile columns, spills 48 values into the segment loop and runs 71.5 us
  //  instead of 35.6 -- every scratch reload there waits for the block stores in flight; TwoBody-LGL7: 81.4 -> 79.6 us)
  static constexpr bool GROW = (N + 1 + n <= 16) || !ASSET_RES_GROWLESS || N + n > 16 || WPS != 1;
  static constexpr int JR0 = GROW ? N + 1 : N;    // first J^ row of the A operand
  static constexpr bool JRIDE_ROWS = ASSET_RES_JRIDE && !D::TRAP && (JR0 + n <= 16);
  // (without the g^ row the sum reads g^_i from the slot while the buffers are in use: no buffer may lie over it)
  static constexpr bool T_XTRA = JRIDE_ROWS && (t_off_slot(K - 1) < 0 || !GROW);
  static constexpr bool JRIDE = JRIDE_ROWS && (T_XTRA || t_off_slot(K - 1) >= 0);
  static constexpr int t_off(int i) { return T_XTRA ? i * TB : t_off_slot(i); }   // relative to the slot, or (T_XTRA) to x_T
  // 16 dead cells behind the last buffer: where the lanes without a J row in an accumulator entry write instead (a store under
  // a lane condition costs an exec-mask round trip each; 24 of them per segment)
  static constexpr int t_end = t_off(K - 1) + TB;
  static constexpr int t_dummy = T_XTRA ? K * TB
                                 : ((t_off(K - 1) >= deadB0 && t_end + 16 <= deadB1) ? t_end
                                    : ((t_off(K - 1) < deadB0 && deadB0 + 16 <= deadB1) ? deadB0 : -1));
  // wave-level scratch behind the slots
  static constexpr int x_AUX = 0;                  // [K][4]: 1 - s_i, s_i, 0, 1  (tau row / parameter rows of DI_i, same row stride as the tables)
  static constexpr int x_HT = x_AUX + 4 * K;       // [IRP] full time-partial vector (rank-2 rows)
  static constexpr int x_CL = x_HT + IRP;          // [CS][n]  sum_i C_ij lam_(i,r)
  static constexpr int x_WL = x_CL + CS * n;       // [CS][n]  sum_i D_ij lam_(i,r)
  static constexpr int x_Z4 = x_WL + CS * n;       // four zeros (a weight row of the lanes without a defect row)
  static constexpr int x_T = x_Z4 + 4;             // (T_XTRA) [K][16][n] the J^ buffers of the ride + 16 dummy cells
  static constexpr int x_FLAG = x_T + (T_XTRA ? K * TB + 16 : 0);   // (EARLYC) the group whose cardinal second derivatives wave A has finished
  static constexpr int XTRA = x_FLAG + 2;
  // (round 6, row-wise dense part at level 2: the rows of J as what they are -- adjoint gradients for UNIT multiplier vectors -- and the
  //  adjoint gradient itself from the H passes, defect_rowdpp.h: UNITC.)  The time rows of the gradient need the segment's
  //  FB = sum_j f_j . BM_j + sum_i E_i f^_i . lam_i: the ODE phases leave its CS + K terms -- a lane per node / interior point -- in eight
  //  cells per slot laid over x_HT / x_CL / x_WL, which only the tile form uses.
  static constexpr int x_FBL = x_HT, FBL_LD = 8;
  // waves per SIMD the kernel is built for (registers: 512 / WPS per lane; LDS: 160 KiB / 4 WPS per wave).  Shapes with two
  // row tiles of defect rows keep two more accumulators and a third column tile's fragments: at 256 registers they spill
  // 650 bytes per lane (TwoBody-LGL7 x 10 000: 149.5 us), with the SIMD to themselves they do not (89.1 us; round 2's kernel 102.7)

  static constexpr int LDS_WAVE = 160 * 1024 / (4 * WPS);
  static constexpr int GR_FIT = (LDS_WAVE / 8 - D::TABSZ - XTRA) / SLOT;
  // PAIR: two-wave workgroups whose waves share the ODE stage.  A wave's ODE phases keep 15-20 of 64 lanes busy, and two waves
  // of a SIMD issue them at the SIMD's f64 rate (tools/ubench_issue.hip: 7.5 cycles per instruction and wave in pairs, 4.5-6
  // alone) -- the lane count costs nothing, the instruction count does.  In a pair ONE wave runs a phase for both waves'
  // segments (40 / 30 lanes for Reentry-LGL7) while the other loads its lane record or waits at the barrier: half the ODE
  // instructions per SIMD.  Shapes built for two waves per SIMD that run their own ODE stage; each wave keeps its own LDS
  // region [tables | GR slots | scratch], so every offset of the lane record holds for both.
  static constexpr bool PAIR = ASSET_RES_PAIR && WPS == 2 && Ode::NUNITS == 1;
  // masked stores of the segment loop as raw buffer stores (out-of-range offsets for the masked lanes) and the column role computed by
  // every lane: the loop is one basic block.  Not the two-row-tile shapes at one wave per SIMD: at 500 of 512 registers the freer
  // schedule spills (TwoBody-LGL7: 13 registers into the loop, 78.3 -> 81.4 us)
#define ASSET_RES_BSTORE (!(WPS == 1 && D::TJ > 1 && Ode::NUNITS == 1))
  static constexpr bool BSTORE = ASSET_RES_BSTORE;
  static constexpr int NWV = PAIR ? 2 : 1;                          // waves per workgroup
#ifndef ASSET_RES_LOOP_PAIR
#define ASSET_RES_LOOP_PAIR 1
#endif
  // the looped level-2 kernel in two-wave workgroups too (row-wise part) -- for right-hand sides heavy enough that sharing the ODE stage
  // pays for the pair's barriers (measured after UNITC, profiles/r6_forms2.txt: Reentry, 411 operations, every shape and every looped mesh
  // size 5-15 % faster as pairs -- LGL7 x 30 000 74 against 88 us, LGL5-BlockConstant x 1 000 000 1.79 / 1.94 ms; TwoBody, 186, every shape
  // 10-25 % SLOWER -- LGL5-BlockConstant x 100 000 195 against 169 us, LGL3 x 1 000 000 1.29 / 1.01 ms; Brachistochrone, 26, within 3 %
  // either way).  Light right-hand sides keep single-wave workgroups, whose waves drift apart and run one's ODE stage under another's stores.
  static constexpr int LOOP_PAIR_MIN_OPS = 300;
  static constexpr bool LOOP_PAIR = ASSET_RES_LOOP_PAIR && RD_ANY && Ode::OPS_FJGH >= LOOP_PAIR_MIN_OPS;
  // EARLYC (round 5): the cardinal value phase also leaves the Jacobians J_j in the slot (Ode::fj_save), so that the rows of [J ; g^T]
  // -- the C passes of the row-wise dense part, 45 % of a segment's bytes -- are formed and stored by one wave of the pair WHILE the
  // other runs the cardinal second-derivative phase: the stores start one phase earlier.
#ifndef ASSET_RES_EARLYC
#define ASSET_RES_EARLYC 1
#endif
  static constexpr bool EARLYC = ASSET_RES_EARLYC && RD_ANY && PAIR;   // (of the kernels that run the row-wise part)
  static constexpr int GR_PASS = 64 / (CS * NWV);                   // one pass per phase covers the workgroup's group
  static constexpr int GR = GR_FIT < GR_PASS ? GR_FIT : GR_PASS;
#ifndef ASSET_RD_UNITC
#define ASSET_RD_UNITC 1
#endif
  static constexpr bool UNITC = ASSET_RD_UNITC && RD_ANY && CS + K <= FBL_LD && FBL_LD * (GR > 0 ? GR : 0) <= IRP + 2 * CS * n;
  static constexpr int REGION = D::TABSZ + (GR > 0 ? GR : 0) * SLOT + XTRA;   // a wave's LDS (doubles)
  static constexpr size_t lds_bytes() { return size_t(NWV * REGION) * 8; }
  // (two row tiles of defect rows -- TwoBody-LGL7, K n = 18: with three column tiles beside them a wave needs ~420 registers; such
  //  shapes are built for one wave per SIMD, WPS above: 78 us for 10 000 segments against 149.5 us spilling at two)
#define ASSET_RES_MAX_TJ 2
  static constexpr bool DENSE_OK = !D::WIDE && D::TJ <= ASSET_RES_MAX_TJ && N + 1 <= 16 && GR >= 2 && D::STAGED;
  static constexpr bool OK = DENSE_OK && Ode::NUNITS == 1;
  // heavy right-hand sides (one workgroup per output unit, defect_units.h): the ODE results are in the workspace when the
  // dense part starts -- the kernel's GIVEN form copies a group's slots from there and goes on as usual
  static constexpr bool GIVEN_OK = DENSE_OK && Ode::NUNITS > 1;
  static constexpr int lkN = N & 3, vN = N >> 2;   // accumulator entry that holds row N of an M tile (the E g^ row)
  // lower-triangle H tiles that can hold a cardinal Hessian block (tiles_share_node), numbered among themselves
  static constexpr int sh_index(int tix_want) {
    int k = 0;
    for (int rt = 0; rt < D::TI; rt++)
      for (int ct = 0; ct <= rt; ct++) {
        if (rt * (rt + 1) / 2 + ct == tix_want) return tiles_share_node<D>(ct, rt) ? k : -1;
        if (tiles_share_node<D>(ct, rt)) k++;
      }
    return -1;
  }
  static constexpr int NSH = [] {
    int k = 0;
    for (int rt = 0; rt < D::TI; rt++)
      for (int ct = 0; ct <= rt; ct++) k += tiles_share_node<D>(ct, rt) ? 1 : 0;
    return k > 0 ? k : 1;
  }();
};

// ---------------------------------------------------------------------------------------------- ODE stage: accessors
template <class D, bool ACCG = false>
struct OdeOutRes {   // every result of an ODE body goes to the segment's LDS slot
  lds_double* f_;
  lds_double* J_;
  lds_double* g_;
  lds_double* H_;
  lds_double* sv_;
  const double* lamv_ = nullptr;                 // ACCG (the Jacobian kinds: fj has no g output): multipliers of the point's
  double gacc_[ACCG ? D::N : 1];                 // rows, and g^ = J^^T lam accumulated while J is emitted
  __device__ void f(int k, double v) { if (f_) f_[k] = v; }
  __device__ void J(int k, int i, double v) {
    const int c = D::ode_t::JPOS[k * D::N + i];
    if (c >= 0 && J_) J_[c] = v;
    if constexpr (ACCG) { if (c >= 0) gacc_[i] += lamv_[k] * v; }
  }
  __device__ void g(int i, double v) { if (g_) g_[i] = v; }
  __device__ void H(int i, int j, double v) {
    const int c = D::ode_t::HPOS[i * (i + 1) / 2 + j];
    if (c >= 0 && H_) H_[c] = v;
  }
  __device__ void save(int k, double v) { if (sv_) sv_[k] = v; }
};
template <class D>
struct CardInRes {   // y = [z_j (q)] from the slot; lam = adjoint weights and saved transcendentals in registers
  const lds_double* z;
  const double* w;
  const double* sv;
  int j;
  __device__ double y(int i) const { return i < D::q ? z[j * D::q + i] : z[D::P0 + (i - D::q)]; }
  __device__ double lam(int k) const { return w[k]; }
  __device__ double saved(int k) const { return sv[k]; }
};

template <class D>
struct GatherRun {   // y = X[first index of the segment + j q + i]: index rows that are runs (EvalArgs::affine; p == 0)
  const double* Xs;
  int j;
  __device__ double y(int i) const { return Xs[j * D::q + i]; }
  __device__ double lam(int) const { return 0.0; }
  __device__ double saved(int) const { return 0.0; }
};

// P1: f_j and its transcendental sub-expressions at cardinal node j (reads the solver vector itself: its loads overlap P0's);
// vi == nullptr: the segment's inputs are the run of X that starts at Xs
// (LEVEL 1, the Jacobian kinds: f_j and J_j, nothing saved -- there is no second cardinal phase)
// (WITHJ, level 2: f_j, J_j and the saved values -- Ode::fj_save, ResDims::EARLYC)
template <class Ode, class D, int LEVEL = 2, bool WITHJ = false>
__device__ __attribute__((noinline, not_tail_called)) void res_cardinal_value(lds_double* S, int j, const double* Xs, const int* vi) {
  using R = ResDims<D>;
  OdeOutRes<D> out{S + D::w_Cf + j * D::n, (LEVEL == 1 || WITHJ) ? S + D::w_CJ + j * D::NZJ : nullptr, nullptr, nullptr,
                   LEVEL == 1 ? nullptr : S + R::s_SV + j * R::SV_LD};
  if (vi) {
    GatherIn<D> in{Xs, vi, j};
    if constexpr (LEVEL == 1) Ode::fj(in, out); else if constexpr (WITHJ) Ode::fj_save(in, out); else Ode::f_save(in, out);
  } else {
    GatherRun<D> in{Xs, j};
    if constexpr (LEVEL == 1) Ode::fj(in, out); else if constexpr (WITHJ) Ode::fj_save(in, out); else Ode::f_save(in, out);
  }
}

// P2: interior point i: x^, tau, u^ ; f^, J^, g^ = J^^T lam_i, H^ = lam_i^T d2f
template <class Ode, class D, int LEVEL = 2, bool RDF = ResDims<D>::ROWDPP>
__device__ __attribute__((noinline, not_tail_called)) void res_interior(lds_double* S, int i, const LglTab* tabp, bool have_lam,
                                                                       lds_double* fbl = nullptr) {
  constexpr int n = D::n, m = D::m, q = D::q, N = D::N, T = D::T, CS = D::CS;
  const LglTab& tab = *tabp;
  const lds_double* z = S + D::w_z;
  const lds_double* Cf = S + D::w_Cf;
  const lds_double* lam = S + D::w_lam;
  const double h = z[D::TF] - z[T];
  double y[N];
  double li[n];
#pragma unroll
  for (int k = 0; k < n; k++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < CS; j++) acc += (tab.A[i][j] * z[j * q + k] + (tab.B[i][j] * h) * Cf[j * n + k]);
    y[k] = acc;
  }
  y[T] = z[T] + h * tab.s[i];
#pragma unroll
  for (int k = 0; k < m; k++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < CS; j++) acc += tab.U[i][j] * z[j * q + n + 1 + k];
    y[n + 1 + k] = acc;
  }
#pragma unroll
  for (int k = 0; k < D::p; k++) y[q + k] = z[D::P0 + k];
#pragma unroll
  for (int k = 0; k < n; k++) li[k] = have_lam ? lam[i * n + k] : 0.0;
  if constexpr (RDF && LEVEL >= 1) {   // FB_i[k] = sum_j B_ij f_j[k]: the time rows of the row-wise dense part (defect_rowdpp.h)
#pragma unroll
    for (int k = 0; k < n; k++) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < CS; j++) acc = fma(tab.B[i][j], Cf[j * n + k], acc);
      S[ResDims<D>::s_FB + i * n + k] = acc;
    }
    if (i == 0) S[ResDims<D>::s_FB + D::K * n] = h;   // (the step: the C pass of a segment lays the gradient row over z, which the H passes would read it from)
  }
  RegIn<D> in{y, li};
  if constexpr (LEVEL == 1) {          // f^, J^ and g^ = J^^T lam_i (accumulated while J^ is emitted)
    OdeOutRes<D, true> out{S + D::w_If + i * n, S + D::w_IJ + i * D::NZJ, nullptr, nullptr, nullptr};
    out.lamv_ = li;
#pragma unroll
    for (int b = 0; b < N; b++) out.gacc_[b] = 0.0;
    Ode::fj(in, out);
#pragma unroll
    for (int b = 0; b < N; b++) S[D::w_Ig + i * N + b] = out.gacc_[b];
  } else {
    OdeOutRes<D> out{S + D::w_If + i * n, S + D::w_IJ + i * D::NZJ, S + D::w_Ig + i * N, S + D::w_IH + i * D::NZH, nullptr};
    Ode::fjgh(in, out);
    if constexpr (RDF && ResDims<D>::UNITC) {   // E_i f^_i . lam_i: this point's term of the gradient's time rows (ResDims::x_FBL)
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < n; k++) acc = fma(have_lam ? lam[i * n + k] : 0.0, S[D::w_If + i * n + k], acc);
      fbl[CS + i] = tab.E[i] * acc;
    }
  }
}

// P3: cardinal node j: adjoint weights w_j (LGLDefects.h:369-374) ; J_j, g_j = J_j^T w_j, H_j = w_j^T d2f
// (NOJ: J_j is in the slot already -- EARLYC -- and the partner wave is reading it: not stored again)
template <class Ode, class D, bool NOJ = false, bool RDF = false>
__device__ __attribute__((noinline, not_tail_called)) void res_cardinal_second(lds_double* S, int j, const LglTab* tabp, lds_double* fbl = nullptr) {
  constexpr int K = D::K, n = D::n, N = D::N, T = D::T;
  using R = ResDims<D>;
  const LglTab& tab = *tabp;
  const lds_double* z = S + D::w_z;
  const double h = z[D::TF] - z[T];
  double w[n];
#pragma unroll
  for (int k = 0; k < n; k++) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < K; i++) {
      acc += S[D::w_Ig + i * N + k] * ((tab.E[i] * tab.B[i][j]) * h * h);
      acc += S[D::w_lam + i * n + k] * (tab.D[i][j] * h);
    }
    w[k] = acc;
  }
  if constexpr (RDF && ResDims<D>::UNITC) {        // f_j . w_j = h f_j . BM_j: this node's term of the gradient's time rows (ResDims::x_FBL)
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < n; k++) acc = fma(S[D::w_Cf + j * n + k], w[k], acc);
    fbl[j] = acc;
  }
  double sv[Ode::NSAVE > 0 ? Ode::NSAVE : 1];      // (read before g_j is written: the two may share their cells)
#pragma unroll
  for (int k = 0; k < Ode::NSAVE; k++) sv[k] = S[R::s_SV + j * R::SV_LD + k];
  wave_lds_order();
  CardInRes<D> in{z, w, sv, j};
  OdeOutRes<D> out{nullptr, NOJ ? nullptr : S + D::w_CJ + j * D::NZJ, S + D::w_Cg + j * N, S + D::w_CH + j * D::NZH, nullptr};
  Ode::fjgh_load(in, out);
}

// Trapezoidal, all derivatives (TrapezoidalDefects.h:263-435): there is no interior point, so the adjoint weights of the
// cardinal nodes do not wait for anything -- w_j = lam D_j h (:300-305 with D = 1/2) -- and ONE phase evaluates f_j, J_j,
// g_j = J_j^T w_j and H_j = w_j^T d2f at node j (the whole body: nothing is saved, nothing reloaded)
template <class Ode, class D>
__device__ __attribute__((noinline, not_tail_called)) void res_cardinal_all(lds_double* S, int j, const LglTab* tabp, bool have_lam) {
  constexpr int n = D::n, N = D::N, T = D::T;
  const LglTab& tab = *tabp;
  const lds_double* z = S + D::w_z;
  const double h = z[D::TF] - z[T];
  double w[n];
#pragma unroll
  for (int k = 0; k < n; k++) w[k] = have_lam ? S[D::w_lam + k] * (tab.D[0][j] * h) : 0.0;
  CardInRes<D> in{z, w, nullptr, j};
  OdeOutRes<D> out{S + D::w_Cf + j * n, S + D::w_CJ + j * D::NZJ, S + D::w_Cg + j * N, S + D::w_CH + j * D::NZH, nullptr};
  Ode::fjgh(in, out);
}

// ---------------------------------------------------------------------------------------------- per-lane constants
// Everything of the dense part that depends on the lane and not on the segment: computed once per handle
// (res_lane_setup_kernel), loaded by every wave after its ODE stage.  Offsets are relative to the segment's slot and
// always readable ("no entry" -> the slot's zero cell), offsets into the weight tables are relative to the LDS copy of the
// tables (`tab`, then ResDims::x_AUX) and address rows of stride 4 (LglTab: [K][4] arrays).
template <class Ode, class D>
struct ResLane {
  using R = ResDims<D>;
  static constexpr int K = D::K, n = D::n, q = D::q, N = D::N, CS = D::CS, KS = D::KS, TI = D::TI;
  static constexpr int IR = D::IR, OR = D::OR, P0 = D::P0, T = D::T, TF = D::TF;
  static constexpr int ZERO = R::s_Z0;
  static constexpr int p = D::p;
  static constexpr bool QFAST = (q % 4 == 0) && p == 0;   // the node of column 16ct + lk + 4v does not depend on lk
  // row weights C_il,j / D_il,j of the lane's defect rows in the record (registers), or as rows of the LDS tables: in registers
  // where there is room -- QFAST shapes, and shapes built for one wave per SIMD that are not behind the unit kernels (TwoBody-LGL7:
  // 82.1 us with the weights in registers, 83.4 us from the tables; TwoBody-LGL5 BlockConstant at two waves: 37.4 / 35.9 us)
  static constexpr bool TREG = QFAST || (R::WPS == 1 && Ode::NUNITS == 1);

  // DI fragments (B operand of the M product, A operand of the H and J products): lane (lr, lk) <-> column c = 16ct + lr,
  // row b = 4kk + lk:   dv[ct][i][kk] = tab[cao[ct][kk] + 4i] + h tab[cbo[ct] + 4i] * S[cjo[ct][kk]] -/+ sbv[i][kk] on columns T / TF
  int cao[TI][KS];       // constant part: A_ij [cc == b], tau row, U_ij [cc == b] or a zero
  int cbo[TI];           // B_ij of the column's node (or a zero row)
  int cjo[TI][KS];       // J_j[b][cc]
  int wlo[TI][KS];       // WL[j][b] (state rows) or a zero: this lane's share of  sum_r WL[j][r] J_j[r][cc]  (adjoint gradient)
  // parameter columns (c >= P0; p > 0): DI_i[b][c] = h sum_j B_ij J_j[b][q + pc], [b == q + pc] on the parameter rows
  int cpo[p > 0 ? TI : 1][KS];   // J_0[b][q + pc] (stride NZJ in j) or the zero cell; the lane's column is a parameter column iff cpo differs from it
  int wpo[p > 0 ? TI : 1][KS];   // WL[0][b] (stride n in j) or a zero
  // M product, A operand: lane (lr, lk) <-> row lr of [h E_i H^_i ; E_i g^_i ; h E_i J^_i (JRIDE)], column 4kk + lk
  int ao[KS], ast[KS];   // offset for i = 0, stride in i
  // J product, B operand: lane (lr, lk) <-> defect row jr = lr = (il, rl), row 4kk + lk of (h E_il J^_il)^T
  // (defect rows: row tile jt, jr = 16jt + lr = (il, rl); lanes without a row: il = rl = 0 and every weight below zero)
  static constexpr int TJ = D::TJ;
  int jo[TJ][KS];
  int il[TJ], rl[TJ];
  // defect row weights: sd = sum_jj tD[jj] f_jj[rl] + tE f^_il[rl];  fx = sum_jj tC[jj] z_jj[rl] + h sd
  // (QFAST: in the record; otherwise rows of the weight tables in LDS, dCo / dDo: 32 registers less, and what is selected from
  //  them per accumulator entry cannot be hoisted out of the segment loop into yet more registers)
  double tC[TREG ? TJ : 1][CS], tD[TREG ? TJ : 1][CS], tE[TJ];
  int dCo[TREG ? 1 : TJ], dDo[TREG ? 1 : TJ];
  // DC: initial value of J^T accumulator entry (ct, jt, v), column c = 16ct + lk + 4v:
  //   [cc == rl] C_il,j(c) + h D_il,j(c) S[dco[ct][jt][v]] -/+ sd on the time columns
  // (QFAST: the node of the column is known at compile time; otherwise it is packed two bits per entry, and [cc == rl] one)
  int dco[TI][TJ][4];
  unsigned jnb[QFAST ? 1 : TI];
  unsigned dbt[QFAST ? 1 : TI][TJ];
  // cardinal Hessian blocks: initial value of H accumulator entry (tile, v) of the tiles that can hold one (LGLDefects.h:386-402)
  int cho[R::NSH][4];
  // column role (lanes lk == lkN hold row N of the M tiles): column c = 16ct + lr
  int cgg[TI];           // g_j[cc]  (parameter column: g_0[q + pc], summed over the nodes with stride N)
  int clo[TI];           // CL[j][cc] (cc < n) or a zero

  __device__ void compute(int lane) {
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int oA = __builtin_offsetof(LglTab, A) / 8, oB = __builtin_offsetof(LglTab, B) / 8, oU = __builtin_offsetof(LglTab, U) / 8;
    constexpr int oAUX = D::TABSZ + R::GR * R::SLOT + R::x_AUX;   // (the scratch follows the slots)
    constexpr int oZERO = oAUX + 2;                                // a row of zeros, same stride
    auto jofs = [](int j, int r, int cc) { const int jp = Ode::JPOS[r * N + cc]; return jp >= 0 ? D::w_CJ + j * D::NZJ + jp : ZERO; };
    constexpr int oXT = D::TABSZ + R::GR * R::SLOT;               // start of the wave-level scratch
    for (int ct = 0; ct < TI; ct++) {
      const int c = 16 * ct + lr;
      const bool col = c < P0, par = c >= P0 && c < IR;   // a node's column / a parameter column
      const int j = col ? c / q : 0, cc = col ? c - j * q : 0, pc = par ? c - P0 : 0;
      cbo[ct] = col ? oB + j : oZERO;
      for (int kk = 0; kk < KS; kk++) {
        const int b = 4 * kk + lk;
        int ca = oZERO, cj = ZERO, cp = ZERO, wp = oZERO;
        if (col && b < N) {
          if (b < n) {
            if (cc == b) ca = oA + j;
            cj = jofs(j, b, cc);
          } else if (b == T) {
            if (c == T) ca = oAUX + 0;
            else if (c == TF) ca = oAUX + 1;
          } else if (b < q && cc == b) ca = oU + j;
        }
        if (par && b < N) {
          if (b < n) { cp = jofs(0, b, q + pc); wp = oXT + R::x_WL + b; }
          else if (b == q + pc) ca = oAUX + 3;               // identity on the parameter rows
        }
        cao[ct][kk] = ca;
        cjo[ct][kk] = cj;
        wlo[ct][kk] = (col && b < n) ? oXT + R::x_WL + j * n + b : oZERO;
        if constexpr (p > 0) { cpo[ct][kk] = cp; wpo[ct][kk] = wp; }
      }
      cgg[ct] = col ? D::w_Cg + j * N + cc : (par ? D::w_Cg + q + pc : ZERO);
      clo[ct] = (col && cc < n) ? oXT + R::x_CL + j * n + cc : oZERO;
    }
    for (int kk = 0; kk < KS; kk++) {
      const int b = 4 * kk + lk;
      int o = ZERO, st = 0;
      if (b < N) {
        if (lr < N) {
          const int hp = Ode::HPOS[(b >= lr) ? b * (b + 1) / 2 + lr : lr * (lr + 1) / 2 + b];
          if (hp >= 0) { o = D::w_IH + hp; st = D::NZH; }
        } else if (lr == N && R::GROW) { o = D::w_Ig + b; st = N; }
        else if (R::JRIDE && lr >= R::JR0 && lr < R::JR0 + n) {
          const int jp = Ode::JPOS[(lr - R::JR0) * N + b];
          if (jp >= 0) { o = D::w_IJ + jp; st = D::NZJ; }
        }
      }
      ao[kk] = o;
      ast[kk] = st;
    }
    const LglTab& tab = d_lgl_tab[D::TAB];
    for (int jt = 0; jt < TJ; jt++) {
      const int jr = 16 * jt + lr;
      const bool row = jr < OR;
      il[jt] = row ? jr / n : 0;
      rl[jt] = row ? jr - il[jt] * n : 0;
      for (int kk = 0; kk < KS; kk++) {
        const int b = 4 * kk + lk;
        const int jp = (row && b < N) ? Ode::JPOS[rl[jt] * N + b] : -1;
        jo[jt][kk] = jp >= 0 ? D::w_IJ + il[jt] * D::NZJ + jp : ZERO;
      }
      if constexpr (TREG) {
        for (int jj = 0; jj < CS; jj++) { tC[jt][jj] = row ? tab.C[il[jt]][jj] : 0.0; tD[jt][jj] = row ? tab.D[il[jt]][jj] : 0.0; }
      } else {
        constexpr int oC = __builtin_offsetof(LglTab, C) / 8, oD = __builtin_offsetof(LglTab, D) / 8;
        dCo[jt] = row ? oC + 4 * il[jt] : oXT + R::x_Z4;
        dDo[jt] = row ? oD + 4 * il[jt] : oXT + R::x_Z4;
      }
      tE[jt] = row ? tab.E[il[jt]] : 0.0;
      for (int ct = 0; ct < TI; ct++) {
        unsigned db = 0;
        for (int v = 0; v < 4; v++) {
          const int c = 16 * ct + lk + 4 * v;
          const bool ok = row && c < P0, par = row && c >= P0 && c < IR;
          const int j = ok ? c / q : 0, cc = ok ? c - j * q : 0;
          dco[ct][jt][v] = ok ? jofs(j, rl[jt], cc) : (par ? jofs(0, rl[jt], q + (c - P0)) : ZERO);   // (parameter column: stride NZJ in j)
          if (ok && cc == rl[jt]) db |= 1u << v;
        }
        if constexpr (!QFAST) dbt[ct][jt] = db;
      }
    }
    if constexpr (!QFAST)
      for (int ct = 0; ct < TI; ct++) {
        unsigned jb = 0;
        for (int v = 0; v < 4; v++) {
          const int c = 16 * ct + lk + 4 * v;
          jb |= unsigned(c < P0 ? c / q : 0) << (2 * v);
        }
        jnb[ct] = jb;
      }
    for (int ct = 0; ct < TI; ct++)
      for (int rt = ct; rt < TI; rt++)
        for (int v = 0; v < 4; v++) {
          const int c = 16 * ct + lk + 4 * v, r = 16 * rt + lr, tix = rt * (rt + 1) / 2 + ct;
          int ch = ZERO;
          if (c < IR && r < IR && r >= c) {
            if (c < P0) {                       // a node's column: its own node's rows, or a parameter row (same node's Hessian)
              const int jn = c / q, cc = c - jn * q;
              const int rr = r < P0 ? (r / q == jn ? r - jn * q : -1) : q + (r - P0);
              if (rr >= 0) { const int hp = Ode::HPOS[rr * (rr + 1) / 2 + cc]; if (hp >= 0) ch = D::w_CH + jn * D::NZH + hp; }
            } else {                            // parameter-parameter: summed over the nodes (stride NZH) by the kernel
              const int rr = q + (r - P0), c2 = q + (c - P0), hp = Ode::HPOS[rr * (rr + 1) / 2 + c2];
              if (hp >= 0) ch = D::w_CH + hp;
            }
          }
          if (R::sh_index(tix) >= 0) cho[R::sh_index(tix)][v] = ch;
        }
  }
};

typedef __attribute__((ext_vector_type(2))) unsigned int res_u2;

// The record as the kernel loads it: in quads of words -- table entry [quad][lane] is 16 bytes, one load per four words (77 single-word
// loads took 1.7 k cycles of a wave, 0.7 k in the pair form)
template <class LC>
struct ResRecord {
  static_assert(sizeof(LC) % 4 == 0, "record must be a whole number of words");
  static constexpr int NW = int(sizeof(LC) / 4), NQ = (NW + 3) / 4;
  union { LC lc; unsigned int w[NQ * 4]; };
  __device__ ResRecord() {}
};

}  // namespace asset_hip
#include "defect_rowdpp.h"
namespace asset_hip {

// The handle's table of the resident kernel (one of ASSET_LANE_REPLICAS copies): the ResLane records of the tile form, then --
// shapes with the row-wise dense part -- the row records of defect_rowdpp.h.  Sizes in 32-bit words.
template <class Ode, class D>
constexpr long long res_table_words_tile() {
  if constexpr (ResDims<D>::DENSE_OK) return (long long)ResRecord<ResLane<Ode, D>>::NQ * 256;
  else return 0;
}
template <class Ode, class D>
constexpr long long res_table_words() {
  if constexpr (ResDims<D>::DENSE_OK && ResDims<D>::RD_ANY) return res_table_words_tile<Ode, D>() + RdDims<Ode, D>::table_bytes() / 4;
  else return res_table_words_tile<Ode, D>();
}

template <class Ode, int SCH, bool BLOCKED>
__global__ __launch_bounds__(64) void res_lane_setup_kernel(unsigned int* out) {
  using D = Dims<Ode, SCH, BLOCKED>;
  if constexpr (ResDims<D>::DENSE_OK) {
    using LC = ResLane<Ode, D>;
    ResRecord<LC> r;
    for (int k = 0; k < ResRecord<LC>::NQ * 4; k++) r.w[k] = 0u;
    r.lc.compute(threadIdx.x);
    for (int k = 0; k < ResRecord<LC>::NQ * 4; k++) out[(k >> 2) * 256 + threadIdx.x * 4 + (k & 3)] = r.w[k];
    if constexpr (ResDims<D>::RD_ANY) {
      for (int rec = threadIdx.x; rec < RdDims<Ode, D>::NRECH; rec += 64)
        rd_lane_setup<Ode, D, ResDims<D>::s_Z0>(out + res_table_words_tile<Ode, D>(), rec);
      for (int rec = threadIdx.x; rec < RdDims<Ode, D>::NRECC; rec += 64)
        rd_lane_setup_c<Ode, D, ResDims<D>::s_Z0>(out + res_table_words_tile<Ode, D>(), rec);
    }
  }
}

// Rows of 64 dwords from global memory straight into LDS, no registers in between: global_load_lds_dword -- M0 holds the LDS byte
// address of lane 0's dword, lane l's lands 4 l behind it, the instruction offset moves both sides (tools/ubench_ldsdma.hip).
// gbase wave-uniform; completion is counted by vmcnt like any load's.  (Inline assembly: the compiler would not know which LDS
// reads the transfer may alias and would wait for all memory operations in flight ahead of each of them.)
__device__ inline void lds_dma_rows4(unsigned lds_addr, const void* gbase, unsigned voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
               "global_load_lds_dword %2, %3\n\tglobal_load_lds_dword %2, %3 offset:256\n\t"
               "global_load_lds_dword %2, %3 offset:512\n\tglobal_load_lds_dword %2, %3 offset:768\n\t"
               "s_mov_b32 m0, %0" : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(gbase) : "memory");
}
__device__ inline void lds_dma_row(unsigned lds_addr, const void* gbase, unsigned voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(gbase) : "memory");
}

// Exchanges between the four rows of a wave (row = 16 lanes) in the vector ALU: gfx950's v_permlane16_swap_b32 (old, src) returns
// { [old.r0, src.r0, old.r2, src.r2], [old.r1, src.r1, old.r3, src.r3] } and v_permlane32_swap_b32
// { [old.lo, src.lo], [old.hi, src.hi] } (tools/ubench_permlane.hip) -- no LDS round trip as with ds_bpermute (__shfl).
template <int W>   // W = 16: rows 2k, 2k+1 exchanged;  W = 32: the halves
__device__ inline void rows_split(double x, double& even, double& odd) {
  res_u2 lo, hi;
  if constexpr (W == 16) {
    lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
    hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
  } else {
    lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
    hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
  }
  even = __hiloint2double(hi[0], lo[0]);
  odd = __hiloint2double(hi[1], lo[1]);
}
// x summed over the four rows (the lanes with the same lr), every lane gets the total; bitwise what the two __shfl_xor steps give
__device__ inline double rows4_sum(double x) {
  double e, o;
  rows_split<16>(x, e, o); x = e + o;
  rows_split<32>(x, e, o); return e + o;
}
// the value row ROW's lane lr holds, in rows 0 and 1 (the other rows: unspecified)
template <int ROW>
__device__ inline double row_to_rows01(double x, int lr) {
  double e, o;
  if constexpr (ROW >= 2) { rows_split<32>(x, e, o); x = o; }   // [r2, r3, r2, r3]
  rows_split<16>(x, e, o);
  return (ROW & 1) ? o : e;
}

// sum over the 16 lanes of a row (lanes with the same lk): butterfly in DPP, every lane gets the total
__device__ inline double row16_sum(double x) {
  auto step = [](double v, auto ctrl) {
    constexpr int C = decltype(ctrl)::value;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), C, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), C, 0xf, 0xf, false);
    return v + __hiloint2double(hi, lo);
  };
  x = step(x, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  x = step(x, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  x = step(x, std::integral_constant<int, 0x141>{});   // row_half_mirror
  x = step(x, std::integral_constant<int, 0x140>{});   // row_mirror
  return x;
}

// ---------------------------------------------------------------------------------------------- kernel
// LEVEL 2: value + Jacobian + adjoint gradient + adjoint Hessian.  LEVEL 1 (the Jacobian kinds, evalSOE / evalAUG): two ODE
// phases (cardinal f_j, J_j; interior f^, J^, g^), no cardinal second derivatives, no Hessian products; the Hessian slots of
// the blocks are written as zeros unless the caller says it never reads them (ASSET_HIP_KEEP_HESSIAN_SLOTS).
// LOOP: meshes of more than GR segments per wave (a second instantiation: the one-group form is the north-star case and
// loses 1 us to the loop's bookkeeping).
template <class Ode, int SCH, bool BLOCKED, int LEVEL, bool ASM, bool LOOP, bool GIVEN = false, bool LPAIR = false,
          bool RDF = ResDims<Dims<Ode, SCH, BLOCKED>>::ROWDPP>       // RDF: the dense part by output rows (defect_rowdpp.h) instead of tiles
__device__ __forceinline__ void lgl_resident_body(const EvalArgs& a) {
  using D = Dims<Ode, SCH, BLOCKED>;
  using R = ResDims<D>;
  constexpr bool ROWDPP = RDF;
  static_assert(!RDF || R::RD_ANY, "the row-wise dense part: shapes built for it");
  using LCT = ResLane<Ode, D>;
  constexpr int CS = D::CS, K = D::K, KE = R::KE, n = D::n, q = D::q, N = D::N, T = D::T, TF = D::TF;
  constexpr int IR = D::IR, OR = D::OR, IRP = D::IRP, KS = D::KS, TI = D::TI, TJ = D::TJ, GR = R::GR, SLOT = R::SLOT;
  constexpr bool CFULL = (IR == IRP);
  static_assert(GR * CS * R::NWV <= 64, "one pass per phase");

  // (the pair form is the one-group kernel's: the looped instantiation keeps single-wave workgroups -- as a pair it ran 100 000
  //  Reentry-LGL7 segments 2.7 % faster, 328 against 337 us, but its TwoBody-LGL5-BlockConstant instantiation came out of the
  //  compiler wrong with the per-group opaque lane index and 25 % slower without it)
  // (round 5: with the row-wise dense part the looped level-2 kernel exists as a pair as well -- LPAIR, ResDims::LOOP_PAIR: the launcher
  //  takes it on the large meshes; the tile form's looped instantiations stay single waves)
  static_assert(!LPAIR || (LOOP && R::LOOP_PAIR && LEVEL == 2 && !ASM && !GIVEN), "the looped pair form: row-wise dense part, level 2, blocks");
  constexpr bool PAIR = R::PAIR && !GIVEN && (!LOOP || LPAIR);
  constexpr int NWV = PAIR ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wv = PAIR ? __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6) : 0;   // this wave of the workgroup
  lds_double* const tabL = (lds_double*)lds + wv * R::REGION;      // weight tables, then x_AUX ... behind the slots
  lds_double* const slots = tabL + D::TABSZ;
  lds_double* const xtra = slots + GR * SLOT;
  const int lane0 = int(threadIdx.x) & 63;
  const LglTab& tab = *reinterpret_cast<const LglTab*>((double*)tabL);
  const LglTab& ctab = d_lgl_tab[D::TAB];                           // compile-time indices: scalar loads

  // this wave's share of the mesh: contiguous, balanced (same rule as IndexingData.h:117-146), walked in groups of at most GR
  // segments (equal groups: 7 segments are 4 + 3).  One group per wave up to GR segments per wave; on larger meshes the waves
  // drift apart from group to group, so that the ODE stage of some runs under the block stores of the others.
  const int nshare = int(gridDim.x) * NWV, share = int(blockIdx.x) * NWV + wv;
  const int per = a.nseg / nshare, rem = a.nseg % nshare;
  int wg_first = share * per + min(share, rem), wg_count = per + (share < rem ? 1 : 0);
  // (PAIR) which shares take the `rem` segments of an uneven split: wave 0 of every workgroup first, then wave 1.  A SIMD hosts wave 1
  // of one workgroup and wave 0 of another (tools/ubench_place.hip), so the longer shares are spread one to a SIMD instead of two
  // to the SIMDs of the first rem / 2 workgroups: at 5 000 Reentry-LGL7 segments (shares of 3 and 2) the busiest SIMD has 5
  // segments instead of 6.
  auto pair_range = [&](int s, int& first, int& count) {
    const int b = s >> 1, w = s & 1, nwg = int(gridDim.x);
    const int rem0 = min(rem, nwg), rem1 = max(rem - nwg, 0);      // extras of the waves 0 / of the waves 1, workgroups in order
    first = s * per + min(b, rem0) + min(b, rem1) + ((w == 1 && b < rem0) ? 1 : 0);
    count = per + ((w == 0 ? b < rem0 : b < rem1) ? 1 : 0);
  };
  // (measured, Reentry-LGL7: 3 000 segments 18.8 -> 18.2 us, 5 000: 23.5 -> 22.8, 7 000: 27.6 -> 26.7, 9 000: 31.7 -> 30.7; with more
  //  extras than workgroups -- 10 000: 1 808 for 1 024 -- the busiest SIMD has two long shares either way and the plain rule is
  //  0.15 us faster)
  const bool w0first = rem <= int(gridDim.x);
  if constexpr (PAIR) { if (w0first) pair_range(share, wg_first, wg_count); }
  if constexpr (GIVEN) {
    // Behind the one-launch unit stage (registry.h): group g of units_gp segments was evaluated on XCD g % 8, and this workgroup runs
    // on XCD blockIdx % 8 -- it takes its segments from that XCD's groups (every group split evenly among the waves the XCD has
    // for it), so the slots come out of the L2 they were written into instead of from memory: the copy of a Betts-LGL5 slot was
    // 4-5 us of the dense part's 13.
    if (a.units_gp > 0 && (int(gridDim.x) & 7) == 0) {
      const int gp = a.units_gp, x = int(blockIdx.x) & 7, k = int(blockIdx.x) >> 3, nW = int(gridDim.x) >> 3;
      const int G = (a.nseg + gp - 1) / gp, Gx = x < G ? (G - x + 7) >> 3 : 0;     // groups of this XCD: x, x + 8, ...
      const int wpg = Gx > 0 ? nW / Gx : 0, gi = wpg > 0 ? k / wpg : 0, wi = wpg > 0 ? k - gi * wpg : 0;
      wg_first = 0, wg_count = 0;
      if (wpg > 0 && gi < Gx) {
        const int g = x + 8 * gi, gfirst = g * gp, gcnt = min(gp, a.nseg - gfirst);
        const int p2 = gcnt / wpg, r2 = gcnt - p2 * wpg;
        wg_first = gfirst + wi * p2 + min(wi, r2), wg_count = p2 + (wi < r2 ? 1 : 0);
      }
    }
  }
  // (the extra segments of an uneven split to the workgroups of XCDs 0-3 first -- they start 2-4 us before XCDs 4-7 in every launch --
  //  was measured again in round 4 and is slower at every size: 10 000 segments 32.26 against 31.93 us, 5 000: 24.0 / 23.5, 3 000: 19.7 / 18.9)
  // (PAIR: the partner's share -- the waves of a pair walk the same number of groups, they meet at barriers)
  const int oshare = share ^ 1;
  int o_first = oshare * per + min(oshare, rem), o_count = per + (oshare < rem ? 1 : 0);
  if constexpr (PAIR) { if (w0first) pair_range(oshare, o_first, o_count); }
  const int cmax = PAIR ? max(wg_count, o_count) : wg_count;
  const int ngroups = LOOP ? (cmax + GR - 1) / GR : (cmax > 0 ? 1 : 0);
  const int gbase = LOOP ? (ngroups > 0 ? wg_count / ngroups : 0) : min(wg_count, GR), gextra = (LOOP && ngroups > 0) ? wg_count % ngroups : 0;
  const int o_gbase = LOOP ? (ngroups > 0 ? o_count / ngroups : 0) : min(o_count, GR), o_gextra = (LOOP && ngroups > 0) ? o_count % ngroups : 0;
  // roles of the pair's waves in the ODE stage: wave 0 runs the cardinal phases (P1, P3), wave 1 the interior phase (P2).  At two
  // 256-register waves per SIMD the dispatcher puts wave 1 of workgroup b and wave 0 of workgroup b + gridDim/4 on one SIMD
  // (tools/ubench_place.hip: 1 002 of 1 024 SIMDs host one wave 0 and one wave 1), so with the roles fixed by the wave index a
  // SIMD runs one ODE phase at a time -- one workgroup's P1, the other's P2, the first one's P3 -- at a lone wave's issue rate.
  // (roles swapped in the upper half of the grid: measured slower, 33.1 against 32.6 us)
  constexpr int wa = 0;
  const bool roleA = !PAIR || wv == wa, roleB = !PAIR || wv != wa;
  auto pair_sync = [&]() {   // LDS hand-over between the pair's waves (no wait for stores in flight)
    if constexpr (PAIR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else wave_lds_sync();
  };
  // (shifting segments from the younger wave of every SIMD to the older one -- 52 / 54 / 56 % to the first half of the grid --
  //  changes nothing: 35.1-35.2 us each)
  // the second wave of its SIMD (workgroups are dealt breadth first: b and b + gridDim/2 share a SIMD; in the pair form b's wave 1
  // and (b + gridDim/4)'s wave 0 do)
  const bool young = PAIR ? (((4 * int(blockIdx.x)) / max(int(gridDim.x), 1)) & 1) != 0
                                                    : 2 * int(blockIdx.x) >= int(gridDim.x);
  lds_double* const region0 = (lds_double*)lds;                     // (PAIR) wave 0's region; wave 1's follows

#if defined(ASSET_TIMING)
  long long tstamp[24];
  int nts = 0;
#define RTS() do { if (nts < 24) tstamp[nts++] = clock64(); } while (0)
#define RTSG() do { if (g == 1) RTS(); } while (0)
#if defined(ASSET_TIMING_FINE)          // (inside the tile columns: products + T writes | LDS hand-over | J^T tile + stores | column role, rank-2 | H stores)
#define RTSF() do { if (g == 1) RTS(); } while (0)
#else
#define RTSF() do {} while (0)
#endif
#else
#define RTS() do {} while (0)
#define RTSF() do {} while (0)
#define RTSG() do {} while (0)
#endif
#if defined(ASSET_WALLCLOCK)
  const long long wall_t0 = wall_clock64();
#endif
  // (GIVEN) the slots arrive one segment ahead of their use, global -> LDS without registers: segment g's dense part starts by asking
  // for slot g + 1 (the last segment of a group: for slot 0 of the next group), and the wait at the top of the next segment is for
  // loads issued a segment's worth of stores ago.  Before: all of a group's slots copied through registers ahead of its first
  // segment, 8 loads per lane in flight -- 22 k cycles per group of four Betts-LGL7 slots at one wave per SIMD, nothing to run under them.
  bool slot0_ahead = false;
  for (int grp = 0, seg0 = wg_first, o_seg0 = o_first; grp < ngroups; grp++) {
  const int gcount = gbase + (grp < gextra ? 1 : 0);
  const int o_gcount = PAIR ? o_gbase + (grp < o_gextra ? 1 : 0) : 0;
  // (PAIR) the workgroup's group: wave 0's segments, then wave 1's; segment g of it -> slot and mesh segment
  const int gc0 = wv == 0 ? gcount : o_gcount, gc1 = PAIR ? (wv == 0 ? o_gcount : gcount) : 0;
  const int sg0 = wv == 0 ? seg0 : o_seg0, sg1 = wv == 0 ? o_seg0 : seg0;
  auto pslot = [&](int g) -> lds_double* {
    if constexpr (PAIR) return region0 + D::TABSZ + (g < gc0 ? g * SLOT : R::REGION + (g - gc0) * SLOT);
    else return slots + g * SLOT;
  };
  auto pseg = [&](int g) -> int { if constexpr (PAIR) return g < gc0 ? sg0 + g : sg1 + (g - gc0); else return seg0 + g; };
  auto pfbl = [&](int g) -> lds_double* {        // the eight cells of group member g for the terms of its FB (ResDims::x_FBL)
    constexpr int o = D::TABSZ + GR * SLOT + R::x_FBL;
    if constexpr (PAIR) return region0 + o + (g < gc0 ? g * R::FBL_LD : R::REGION + (g - gc0) * R::FBL_LD);
    else return tabL + o + g * R::FBL_LD;
  };
  const int gall = PAIR ? gc0 + gc1 : gcount;
  // (the lane index is made opaque per group: what derives from it is then recomputed in every group instead of being
  //  computed once before the loop and kept -- in scratch, there being no registers to keep it in across the ODE bodies)
  int lane = lane0;
  if constexpr (LOOP) asm volatile("" : "+v"(lane));
  const int lr = lane & 15, lk = lane >> 4;
  RTS();
  auto slot_dma = [&](int slot, int seg_) {             // slot <- the workspace's slot of mesh segment seg_ (WSLOTD doubles)
    const void* src = a.work + size_t(__builtin_amdgcn_readfirstlane(seg_)) * D::WSLOT;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(slots + slot * SLOT));
    constexpr int ROWS = D::WSLOTD * 2 / 64, TAIL = D::WSLOTD * 2 % 64;
#pragma unroll
    for (int r = 0; r + 4 <= ROWS; r += 4) lds_dma_rows4(dst + r * 256, src, lane * 4 + r * 256);
#pragma unroll
    for (int r = ROWS / 4 * 4; r < ROWS; r++) lds_dma_row(dst + r * 256, src, lane * 4 + r * 256);
    if (TAIL > 0 && lane < TAIL) lds_dma_row(dst + ROWS * 256, src, lane * 4 + ROWS * 256);
  };
  // (round 5 kept this to the one-group kernel: in the looped pair form the waves meet at a barrier per group and wave B, with all the C passes
  //  of a group in their general form, was the longer one every time -- TwoBody-LGL5-BlockConstant x 1 000 000: 2.53 ms with it, 2.07 ms
  //  without.  Round 6: a unit-multiplier C pass is half the instructions, and the looped pair kernel is built for heavy right-hand sides only
  //  (ResDims::LOOP_PAIR), whose cardinal second-derivative phase is long: wave B's C passes now fit under it -- Reentry-LGL7 x 100 000
  //  264 -> 254 us, x 1 000 000 2.55 -> 2.45 ms; LGL5-BlockConstant x 1 000 000 1.88 -> 1.76 ms; LGL3 x 100 000 83.5 -> 77.8 us: profiles/r6_ecl.txt)
  constexpr bool EARLYC = R::EARLYC && ROWDPP && PAIR && (!LOOP || LPAIR) && LEVEL == 2 && !ASM && !GIVEN && !D::TRAP;
  unsigned int rd_rec[R::RD_ANY ? RdDims<Ode, D>::NQH * 4 : 1];       // the row record of the row-wise dense part (defect_rowdpp.h)
  const unsigned int* const rd_rectab = static_cast<const unsigned int*>(a.lane_consts_res) +
                                        size_t(blockIdx.x % ASSET_LANE_REPLICAS) * size_t(res_table_words<Ode, D>()) + res_table_words_tile<Ode, D>();
  ResRecord<LCT> lrec;
  auto load_record = [&]() {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    const u4* rec = reinterpret_cast<const u4*>(static_cast<const unsigned int*>(a.lane_consts_res) +
                                                size_t((blockIdx.x + grp) % ASSET_LANE_REPLICAS) * size_t(res_table_words<Ode, D>()));   // (reloaded per
#pragma unroll                                                                                                             //  group: the ODE bodies need the registers)
    for (int k = 0; k < ResRecord<LCT>::NQ; k++) {
      const u4 v = rec[k * 64 + lane];
      lrec.w[4 * k] = v.x, lrec.w[4 * k + 1] = v.y, lrec.w[4 * k + 2] = v.z, lrec.w[4 * k + 3] = v.w;
    }
  };
  if constexpr (GIVEN) {
    // ------------------------------------------------------------------ slots from the workspace (defect_units.h wrote them)
    // (the record first: there is no ODE body here that needs the registers, and its round trip to memory runs under the slots')
    if (!slot0_ahead) slot_dma(0, seg0);                // (later groups: the last segment of the group before has asked for it)
    load_record();
    // (eight requests in flight per lane; 32 -- the slots of two Betts-LGL5 segments in one round trip instead of four -- changes
    //  nothing: 30.6 against 30.1 us for 1 000 segments, 106.1 / 106.7 for 5 000: the 11.8 k cycles of this copy are the slots arriving
    //  from where the unit kernels of seven other workgroups, on other XCDs, have just written them)
    constexpr int NTAB = (D::TABSZ + 63) / 64;
    double tabv[NTAB];
#pragma unroll
    for (int t = 0; t < NTAB; t++)
      tabv[t] = (lane + 64 * t < D::TABSZ) ? reinterpret_cast<const double*>(&d_lgl_tab[D::TAB])[lane + 64 * t] : 0.0;
#pragma unroll
    for (int t = 0; t < NTAB; t++)
      if (lane + 64 * t < D::TABSZ) tabL[lane + 64 * t] = tabv[t];
    if (lane < 4 * K) {
      const int i = lane >> 2, w = lane & 3;
      xtra[R::x_AUX + lane] = w == 0 ? 1.0 - ctab.s[i] : (w == 1 ? ctab.s[i] : (w == 3 ? 1.0 : 0.0));
    }
    if (lane < GR) slots[lane * SLOT + R::s_Z0] = 0.0;
    if (lane < 4) xtra[R::x_Z4 + lane] = 0.0;
  } else {
  // ------------------------------------------------------------------ ODE stage
#if defined(ASSET_EXP_ODEREP)   // (measurement build: the stage twice -- the second pass finds its code in the instruction cache)
  for (int oderep = 0; oderep < ASSET_EXP_ODEREP; oderep++) {
  if (oderep > 0) { pair_sync(); RTS(); }
#endif
  // P0: gather z = X[Vindex], lam = L[Cindex] into the slots -- index loads, value loads, LDS writes.  In a pair wave 1 gathers for
  // both waves (and fills both regions' tables) while wave 0 is already in P1, which reads the solver vector itself and writes
  // nothing P0 writes: the gather's round trip to memory -- 2.7 k cycles, 6.6 k at the cold start of a launch -- is off the
  // critical path of the stage.
  if (roleB) {
    constexpr int NZ = (NWV * GR * IR + 63) / 64, NL = (NWV * GR * OR + 63) / 64, NTAB = (D::TABSZ + 63) / 64;
    int vi[NZ], ci[NL];
    if (a.affine) {                      // (uniform) the rows of the index tables are runs: no index loads
#pragma unroll
      for (int t = 0; t < NZ; t++) {
        const int e = lane + 64 * t, g = e / IR, r = e - g * IR;
        vi[t] = (e < gall * IR) ? a.aff_v0 + pseg(g) * a.aff_vs + r : -1;
      }
#pragma unroll
      for (int t = 0; t < NL; t++) {
        const int e = lane + 64 * t, g = e / OR, r = e - g * OR;
        ci[t] = (e < gall * OR) ? a.aff_c0 + pseg(g) * a.aff_cs + r : -1;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NZ; t++) {
        const int e = lane + 64 * t, g = e / IR, r = e - g * IR;
        vi[t] = (e < gall * IR) ? a.vindex[size_t(pseg(g)) * IR + r] : -1;
      }
#pragma unroll
      for (int t = 0; t < NL; t++) {
        const int e = lane + 64 * t, g = e / OR, r = e - g * OR;
        ci[t] = (e < gall * OR) ? a.cindex[size_t(pseg(g)) * OR + r] : -1;
      }
    }
    double tabv[NTAB];
#pragma unroll
    for (int t = 0; t < NTAB; t++)
      tabv[t] = (lane + 64 * t < D::TABSZ) ? reinterpret_cast<const double*>(&d_lgl_tab[D::TAB])[lane + 64 * t] : 0.0;
    double zv[NZ], lv[NL];
#pragma unroll
    for (int t = 0; t < NZ; t++) zv[t] = (vi[t] >= 0) ? a.X[vi[t]] : 0.0;
#pragma unroll
    for (int t = 0; t < NL; t++) lv[t] = (ci[t] >= 0 && a.L) ? a.L[ci[t]] : 0.0;
#pragma unroll
    for (int w = 0; w < NWV; w++) {      // every region of the workgroup: tables, constant rows, zero cells
      lds_double* const tb = PAIR ? region0 + w * R::REGION : tabL;
      lds_double* const xt = tb + D::TABSZ + GR * SLOT;
#pragma unroll
      for (int t = 0; t < NTAB; t++)
        if (lane + 64 * t < D::TABSZ) tb[lane + 64 * t] = tabv[t];
      if (lane < 4 * K) {
        const int i = lane >> 2, wq = lane & 3;
        xt[R::x_AUX + lane] = wq == 0 ? 1.0 - ctab.s[i] : (wq == 1 ? ctab.s[i] : (wq == 3 ? 1.0 : 0.0));
      }
      if (lane < GR) tb[D::TABSZ + lane * SLOT + R::s_Z0] = 0.0;
      if (lane < 4) xt[R::x_Z4 + lane] = 0.0;
      if (grp == 0 && lane == 0) xt[R::x_FLAG] = 0.0;
    }
#pragma unroll
    for (int t = 0; t < NZ; t++) {
      const int e = lane + 64 * t, g = e / IR, r = e - g * IR;
      if (e < gall * IR) pslot(g)[D::w_z + r] = zv[t];
    }
#pragma unroll
    for (int t = 0; t < NL; t++) {
      const int e = lane + 64 * t, g = e / OR, r = e - g * OR;
      if (e < gall * OR) pslot(g)[D::w_lam + r] = lv[t];
    }
  }
  RTS();
  // lane <-> evaluation point, NODE-major: lane = j PW + g.  A ds_write_b64 is serviced in four groups of 16 consecutive lanes over 16
  // 8-byte banks (MI355X_MICROARCH.md, LDS): with the segments of a node in consecutive lanes a group's addresses differ by the odd
  // slot stride -- distinct banks -- whereas segment-major order put the nodes of one segment, N = 8 or NZH = 24 doubles apart, two
  // to a bank in every write of g / H (SQ_LDS_BANK_CONFLICT 497 -> cycles per wave, profiles/r4_*).
  constexpr int PW = (CS * 16 <= 64 && NWV * GR <= 16) ? 16 : NWV * GR;
  const int pj = lane / PW, pg = lane - pj * PW;       // node (or interior point) and segment of the group this lane evaluates
  if constexpr (D::TRAP && LEVEL >= 2) {   // Trapezoidal: one phase (the weights w_j need lam and h: after the gather)
    pair_sync();
    if (roleA && pj < CS && pg < gall) {
      const int g = pg, j = pj;
      res_cardinal_all<Ode, D>(pslot(g), j, &tab, a.L != nullptr);
    }
  } else {
  if (roleA && pj < CS && pg < gall) { // P1 (reads X itself, writes f_j and the saved values: nothing of P0's)
    const int g = pg, j = pj;
    if (a.affine && D::p == 0) res_cardinal_value<Ode, D, LEVEL, EARLYC>(pslot(g), j, a.X + (a.aff_v0 + pseg(g) * a.aff_vs), nullptr);
    else res_cardinal_value<Ode, D, LEVEL, EARLYC>(pslot(g), j, a.X, a.vindex + size_t(pseg(g)) * IR);
  }
  pair_sync();
  RTS();
  // (Round 5 also ran the interior phase and the cardinal second-derivative phase in HALVES -- both waves of a pair, each one half of
  //  the outputs, generated bodies fjgh_half<H>: parity-green and slower, Reentry-LGL7 x 10 000 29.9-30.2 against 28.8-28.9 us: the
  //  halves recompute what they share, and the wave that used to wait shared its SIMD with ANOTHER workgroup's phase.  Removed.)
  if constexpr (!D::TRAP) {
    if (roleB && pj < K && pg < gall) {   // P2
      const int g = pg, i = pj;
      res_interior<Ode, D, LEVEL, ROWDPP>(pslot(g), i, &tab, a.L != nullptr, pfbl(g));
    }
    if constexpr (LEVEL >= 2) pair_sync();
  }
  RTS();
  if constexpr (LEVEL >= 2) {
    if constexpr (EARLYC) {
      // P3 by wave A -- and meanwhile wave B forms and stores the rows of [J ; g^T] of the whole group (defect_rowdpp.h, MODE 1)
      if (roleA) {
        if (pj < CS && pg < gall) res_cardinal_second<Ode, D, true, ROWDPP>(pslot(pg), pj, &tab, pfbl(pg));
        // ... and says so -- a flag in wave 0's region, not a barrier: wave A goes on to its H passes, wave B looks at the flag when
        // its C passes are done (LDS instructions of a wave execute in issue order: whoever sees the flag sees the phase's results)
        wave_lds_sync();
        if (lane == 0) *reinterpret_cast<volatile lds_double*>(region0 + D::TABSZ + GR * SLOT + R::x_FLAG) = double(grp + 1);
      } else {
        int slo = 0x7fffffff, shi = 0;
        if (gc0 > 0) { slo = min(slo, sg0); shi = max(shi, sg0 + gc0); }
        if (gc1 > 0) { slo = min(slo, sg1); shi = max(shi, sg1 + gc1); }
        rowdpp_dense<Ode, D, R::s_Z0, R::s_FB, LEVEL, 1>(a, (const lds_double*)tabL, rd_rectab, gall, slo, shi, wv, NWV, lane,
                                                      [&](int g) -> const lds_double* { return pslot(g); }, pseg,
                                                      [&](int g) -> const lds_double* { return pfbl(g); }, rd_rec);
      }
    } else if (roleA && pj < CS && pg < gall) {  // P3
      const int g = pg, j = pj;
      res_cardinal_second<Ode, D, false, ROWDPP>(pslot(g), j, &tab, pfbl(g));
    }
  }
  }
#if defined(ASSET_EXP_ODEREP)
  }
#endif
  }   // (!GIVEN)
  RTS();
  if constexpr (ROWDPP && LEVEL >= 1 && !ASM && !GIVEN) {
    // ------------------------------------------------------------------ dense part by output rows (defect_rowdpp.h): the workgroup's
    // (segment, row group) tasks in passes of four, dealt to the two waves alternately
    if constexpr (EARLYC) {
      if (!roleA) {                    // wave B: the cardinal second derivatives of the group are in the slots?
        volatile lds_double* const flag = reinterpret_cast<volatile lds_double*>(region0 + D::TABSZ + GR * SLOT + R::x_FLAG);
        while (*flag != double(grp + 1)) __builtin_amdgcn_s_sleep(2);
      }
    } else pair_sync();                // (the last ODE phase's results, for both waves)
    RTS();
    int seg_lo = 0x7fffffff, seg_hi = 0;
    if (gc0 > 0) { seg_lo = min(seg_lo, sg0); seg_hi = max(seg_hi, sg0 + gc0); }
    if (gc1 > 0) { seg_lo = min(seg_lo, sg1); seg_hi = max(seg_hi, sg1 + gc1); }
    if constexpr (EARLYC) {
      // (wave B has stored the C passes -- nCP x 540 instructions -- while wave A ran the cardinal second-derivative phase, ~ 550)
      const int nCP = (gall + (4 / (RdDims<Ode, D>::CRG > 0 ? RdDims<Ode, D>::CRG : 1) > 0 ? 4 / RdDims<Ode, D>::CRG : 1) - 1) /
                      (4 / RdDims<Ode, D>::CRG > 0 ? 4 / RdDims<Ode, D>::CRG : 1);
      const int lB = nCP * (R::UNITC ? ASSET_RD_CCOST_UNIT : 540), lA = 550;   // (weights 900 / 370 -- wave A four H passes of five instead of three -- measured the same at 10 000 Reentry segments and slower at 5 000)
      rowdpp_dense<Ode, D, R::s_Z0, R::s_FB, LEVEL, 2>(a, (const lds_double*)tabL, rd_rectab, gall, seg_lo, seg_hi, wv, NWV, lane,
                                                    [&](int g) -> const lds_double* { return pslot(g); }, pseg,
                                                    [&](int g) -> const lds_double* { return pfbl(g); }, rd_rec, /*have_rec=*/!roleA,
                                                    wa == 0 ? lA : lB, wa == 0 ? lB : lA
#if defined(ASSET_TIMING)
                                                    , tstamp, &nts
#endif
                                                    );
    } else
    rowdpp_dense<Ode, D, R::s_Z0, R::s_FB, LEVEL>(a, (const lds_double*)tabL, rd_rectab, gall, seg_lo, seg_hi, wv, NWV, lane,
                                                  [&](int g) -> const lds_double* { return pslot(g); }, pseg,
                                                  [&](int g) -> const lds_double* { return pfbl(g); }, rd_rec, false, 0, 0
#if defined(ASSET_TIMING)
                                                  , tstamp, &nts
#endif
                                                  );
    RTS();
    seg0 += gcount;
    o_seg0 += o_gcount;
    // (the next group's gather rewrites the slots: in a pair BOTH waves must be through with this group's passes first -- without
    //  the barrier wave 1 gathers group g + 1 into slots wave 0 still reads: wrong blocks from the third group on, 100 003 TwoBody
    //  segments; very likely what round 4 saw as a "miscompiling" looped pair form of the tile kernel, whose group end has the
    //  same wave-local wait)
    pair_sync();
    continue;
  }
  // the per-lane record of the dense part (its loads fly while P3's LDS writes land; in a pair the wave that is not in the
  // last phase loads it while the other computes)
  if constexpr (!GIVEN) load_record();
  const LCT& lc = lrec.lc;
  if constexpr (!GIVEN) pair_sync(); else wave_lds_sync();   // (the last ODE phase's results, for both waves)
  wave_loads_landed();
  RTS();

  // ------------------------------------------------------------------ dense part, one segment at a time
  auto tabrow = [&](int o, int i) -> double { return tabL[o + 4 * i]; };
  // weights C_il,jj / D_il,jj of the lane's defect row of row tile jt (zeros without a row)
  auto tCw = [&](int jt, int jj) -> double { if constexpr (LCT::TREG) return lc.tC[jt][jj]; else return tabL[lc.dCo[jt] + jj]; };
  auto tDw = [&](int jt, int jj) -> double { if constexpr (LCT::TREG) return lc.tD[jt][jj]; else return tabL[lc.dDo[jt] + jj]; };   // row i of a [K][4] weight array (or of x_AUX)
  // -1 on column T, +1 on column TF of the lane's column 16t + lr: the direction d = e_TF - e_T of the rank-2 update and the
  // sign of the time-column terms
  auto tsA = [&](int t) -> double { return (16 * t + lr == T) ? -1.0 : ((16 * t + lr == TF) ? 1.0 : 0.0); };
  lds_double* const CL = xtra + R::x_CL;
  lds_double* const WL = xtra + R::x_WL;
  constexpr int NFRAG = (D::NTH + TI * TJ) * 4;

  const int nb_kkt = (!ASM && a.KKT) ? int(D::KSTRIDE * 8) : 0, nb_fx = a.FX ? OR * 8 : 0, nb_agx = (a.AGX && a.L) ? IR * 8 : 0;
  const int nb_h = (LEVEL >= 2 || !(a.flags & 1)) ? nb_kkt : 0;   // (Jacobian kinds: zeros, unless the caller never reads them)
  auto seg_lds_sync = [&]() { wave_lds_order(); };   // (hand-offs inside a segment, one wave: an ordering, not a wait -- defect_dims.h)
  auto segment = [&](const int g) __attribute__((always_inline)) {
    const lds_double* S = slots + g * SLOT;
    const size_t seg = size_t(__builtin_amdgcn_readfirstlane(seg0 + g));   // (wave-uniform, and the compiler must know: buffer resources)
    if constexpr (GIVEN) {
      // this segment's slot: asked for at the top of the segment before -- every store of that segment was issued behind it, and
      // memory operations complete in the order of issue -- or, the first one, just now.  NST store instructions per segment,
      // none under a branch (BSTORE): vmcnt(NST) is "the slot has landed" without a wait for the stores themselves.
      constexpr int NST = (R::BSTORE && !ASM) ? (TI * TJ + D::NTH) * 4 : 0, NW = NST > 63 ? 63 : NST;
      if (g == 0 && !slot0_ahead) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NW) : "memory");
      slot0_ahead = false;
      if (g + 1 < gcount) slot_dma(g + 1, seg0 + g + 1);
      else if (LOOP && grp + 1 < ngroups && g >= 1) { slot_dma(0, seg0 + gcount); slot0_ahead = true; }
    }
    const double h = S[D::w_z + TF] - S[D::w_z + T];
    const double rh = 1.0 / h;   // (one division per segment, off the tile columns' critical chain)
    double* const kkt_dst = ASM ? a.values : (a.KKT ? a.KKT + seg * size_t(D::KSTRIDE) : nullptr);
    // lane-masked block stores as raw buffer stores whose masked lanes carry an out-of-range offset (dropped by the bounds check):
    // no exec-mask region, no basic-block boundary per store group; an output the caller did not ask for is a resource of zero
    // records (sizes nb_* formed once, ahead of the loop) -- no branch on the pointers inside the segment either
    struct SegOut { __amdgpu_buffer_rsrc_t rs; double* p; int nb; };
    auto seg_out = [&](double* seg_base, int bytes) { return SegOut{__builtin_amdgcn_make_buffer_rsrc(seg_base, 0, bytes, 0x00020000), seg_base, bytes}; };
    const SegOut o_kkt = seg_out(a.KKT + seg * size_t(D::KSTRIDE), nb_kkt), o_h = seg_out(a.KKT + seg * size_t(D::KSTRIDE), nb_h);
    const SegOut o_fx = seg_out(a.FX + seg * size_t(OR), nb_fx), o_agx = seg_out(a.AGX + seg * size_t(IR), nb_agx);
    auto bst = [&](const SegOut& o, unsigned idx, bool ok, double v) {
      if constexpr (R::BSTORE) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(res_u2, v), o.rs, ok ? idx * 8u : 0xFFFFFFF0u, 0, 0);
      else if (o.nb != 0 && ok) o.p[idx] = v;
    };
    const int* const kmap_seg = ASM ? a.kmap + seg * size_t(NFRAG) * 64 + lane : nullptr;
    int hmap[ASM ? D::NTH : 1][4], jmap[ASM ? TI * TJ : 1][4];
    if constexpr (ASM) {                              // all of the segment's map entries, ahead of the products
#pragma unroll
      for (int t = 0; t < D::NTH; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) hmap[t][v] = kmap_seg[(t * 4 + v) * 64];
#pragma unroll
      for (int t = 0; t < TI * TJ; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) jmap[t][v] = kmap_seg[((D::NTH + t) * 4 + v) * 64];
    }

    RTSG();
    // ---- R1: defect rows (il, rl) of this lane, one per row tile: sd, value; the multiplier sums of the adjoint gradient
    double sd[TJ], fxv[TJ], lsd = 0.0;
#pragma unroll
    for (int jt = 0; jt < TJ; jt++) {
      const int jr = 16 * jt + lr;
      double fj[CS], zj[CS];
      const double fi = KE > 0 ? S[D::w_If + (jr < OR ? jr : 0)] : 0.0;   // f^_il[rl] (its weight tE is zero in the lanes without a row)
#pragma unroll
      for (int jj = 0; jj < CS; jj++) { fj[jj] = S[D::w_Cf + jj * n + lc.rl[jt]]; zj[jj] = S[D::w_z + jj * q + lc.rl[jt]]; }
      double sdv = lc.tE[jt] * fi, fx = 0.0;
#pragma unroll
      for (int jj = 0; jj < CS; jj++) { sdv += tDw(jt, jj) * fj[jj]; fx += tCw(jt, jj) * zj[jj]; }
      sd[jt] = sdv;
      fxv[jt] = fx + h * sdv;
      const double lamr = S[D::w_lam + (jr < OR ? jr : 0)];      // (read, then select: a load under a lane condition is an exec-mask
      lsd += ((jr < OR) ? lamr : 0.0) * sdv;                       //  region of its own, and a basic-block boundary for the scheduler)
    }
    int lkv = lk, lkb = lk * D::HCA - ((lk * (lk - 1)) >> 1);          // (opaque per iteration: what is derived from them is
    asm volatile("" : "+v"(lkv), "+v"(lkb));                              //  recomputed, not kept in registers across the loop)
    const double sls = row16_sum(lsd);                // sum_(i,r) lam_(i,r) sd_(i,r), in every lane
    {                                                 // CL[j][r] = sum_i C_ij lam_(i,r), WL[j][r] = sum_i D_ij lam_(i,r)
      // (by every lane: the ones past the CS n entries compute entry 0 again and write it to cells nobody reads -- the HT
      //  vector of the old time-row path -- instead of sitting out an exec-mask region)
      const bool own = lane < CS * n;
      const int e = own ? lane : 0, j = e / n, r = e - j * n;
      double cl = 0.0, wl = 0.0;
#pragma unroll
      for (int i = 0; i < K; i++) {
        const double l = S[D::w_lam + i * n + r];
        cl += tab.C[i][j] * l;
        wl += tab.D[i][j] * l;
      }
      if constexpr (R::BSTORE) {
        xtra[own ? R::x_CL + lane : R::x_HT + (lane & 15)] = cl;      // (sixteen dead cells, one per lane of a write's 16-lane group:
        xtra[own ? R::x_WL + lane : R::x_HT + (lane & 15)] = wl;      //  forty lanes on ONE cell were forty bank-conflict cycles per write)
      } else if (own) {
        CL[lane] = cl;
        WL[lane] = wl;
      }
    }

    RTSG();
    // ---- R2: fragments of DI_i, straight from the slot; this lane's rows of  sum_r WL[j][r] J_j[r][cc]
    seg_lds_sync();                                   // (CL / WL)
    double dv[TI][K][KS], agJ[TI];
    {
      double sbv[K][KS];
#pragma unroll
      for (int kk = 0; kk < KS; kk++) {
        const int b = 4 * kk + lk;                    // f_jj[b] for the state rows, nothing for the others
        double f[CS];
#pragma unroll
        for (int jj = 0; jj < CS; jj++) f[jj] = S[D::w_Cf + jj * n + (b < n ? b : 0)];   // (in range for every lane; selected below:
#pragma unroll                                                                      //  KE selects per k-step instead of CS)
        for (int i = 0; i < KE; i++) {
          double s = 0.0;
#pragma unroll
          for (int jj = 0; jj < CS; jj++) s += ctab.B[i][jj] * f[jj];
          sbv[i][kk] = (b < n) ? s : 0.0;
        }
      }
#pragma unroll
      for (int ct = 0; ct < TI; ct++) {
        double jv[KS];
#pragma unroll
        for (int kk = 0; kk < KS; kk++) jv[kk] = S[lc.cjo[ct][kk]];
        const double tsa = tsA(ct);
        double part = 0.0;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) part = fma(tabL[lc.wlo[ct][kk]], jv[kk], part);
        agJ[ct] = part;
#pragma unroll
        for (int i = 0; i < KE; i++) {
          const double hb = h * tabrow(lc.cbo[ct], i);
#pragma unroll
          for (int kk = 0; kk < KS; kk++)
            dv[ct][i][kk] = fma(tsa, sbv[i][kk], fma(hb, jv[kk], tabrow(lc.cao[ct][kk], i)));
        }
        if constexpr (D::p > 0) {
          if (16 * ct + 15 >= D::P0) {               // this tile has parameter columns: h sum_j B_ij J_j[b][q + pc] on the state rows
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
              const bool par = lc.cpo[ct][kk] != LCT::ZERO;
              const int jst = par ? D::NZJ : 0, wst = par ? n : 0;
              double jp[CS];
#pragma unroll
              for (int jj = 0; jj < CS; jj++) {
                jp[jj] = S[lc.cpo[ct][kk] + jj * jst];
                agJ[ct] = fma(tabL[lc.wpo[ct][kk] + jj * wst], jp[jj], agJ[ct]);
              }
#pragma unroll
              for (int i = 0; i < KE; i++) {
                double sp = 0.0;
#pragma unroll
                for (int jj = 0; jj < CS; jj++) sp = fma(ctab.B[i][jj], jp[jj], sp);
                dv[ct][i][kk] = fma(h, sp, dv[ct][i][kk]);
              }
            }
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < TI; ct++) {              // the four lanes of a column (lk = 0..3) hold its row groups: add them up
        agJ[ct] = rows4_sum(agJ[ct]);
      }
    }

    RTSG();
    // The products are ordered so that every result is stored as soon as it is complete and its stores drain under the
    // matrix instructions that follow (a wave's store phase is otherwise a stall at the rate the memory system accepts the
    // data): J^T first, then one tile column rt of H at a time -- M_i[:, tile rt] is the B operand of the tiles (ct <= rt, rt)
    // only, the rank-2 time rows of those tiles need HT on the tiles up to rt.
    // Entry v of a tile: block column c = 16ct + lk + 4v, row (H) r = 16rt + lr or (J) jr = lr; 16 consecutive lanes cover
    // 128 contiguous bytes of the block in either layout (defect_dims.h, Dims::KL; the reference's order:
    // DenseFunctionBase.h:1112-1123): H(r, c) sits at hcb + r, J(jr, c) at jcb + jr.
    auto hcb = [&](int ct, int v) {                  // c = c0 + lk:  hcol(c) = hcol(c0) + [lk HCA - lk (lk-1) / 2] - c0 lk
      const int c0 = 16 * ct + 4 * v;
      return D::HOFF + c0 * D::HCA - ((c0 * (c0 - 1)) >> 1) + lkb - c0 * lkv;
    };
    auto jcb = [&](int ct, int v) {
      if constexpr (D::KL != 0) return (16 * ct + 4 * v) * OR + lkv * OR;
      else return hcb(ct, v) + IR;
    };
    double ah[K][KS];                                  // A operand of the M products: [h E_i H^_i ; E_i g^_i]
#pragma unroll
    for (int i = 0; i < KE; i++) {
      const double sc = (R::GROW && lr == N) ? ctab.E[i] : h * ctab.E[i];     // the g^ row is scaled by E_i, the H^ / J^ rows by h E_i
#pragma unroll
      for (int kk = 0; kk < KS; kk++)           // (LEVEL 1: no H^ -- its section of the slot holds nothing)
        ah[i][kk] = (LEVEL >= 2 || lr >= N) ? sc * S[lc.ao[kk] + i * lc.ast[kk]] : 0.0;
    }
#define ASSET_RES_PRIO (PAIR ? 0 : 1)
    // The two waves of a SIMD are workgroups b and b + gridDim/2; left alone the older one wins every arbitration and ends
    // ~5 us before the younger one (36.8 against 31.5 us), i.e. the kernel ends 2.5 us later than it would with both ending
    // together.  ASSET_RES_PRIO 1: the younger wave runs its products at a higher priority than the older one -- the looped
    // kernel (100 000 Reentry segments: 308.3 against 313.5 us with equal priorities).  The pair form, whose waves meet at barriers
    // through the ODE stage and start their segments together, runs them at equal priority since the segment loop is one block
    // (10 000 segments 28.65 against 29.15 us, 5 000: 20.79 / 20.94, Trapezoidal 11.1 / 11.3, TwoBody-LGL5 33.13 / 33.32), and
    // the loop body exists once instead of once per priority.
    if (ASSET_RES_PRIO == 1) { if (young) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
    else if (ASSET_RES_PRIO == 2) { if ((g & 1) == int(young)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
    else __builtin_amdgcn_s_setprio(1);
    // ---- R3: cardinal part of J (DC): the initial value of the J^T tiles; without JRIDE the interior part right away
    //      J^T = DC^T + sum_i DI_i^T (h E_i J^_i)^T, stored at once
    auto dc_tile = [&](int ct, int jt) -> d4 {
      d4 r;
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const int c0 = 16 * ct + 4 * v;                         // column = c0 + lk
          double dd, cw;
          if constexpr (LCT::QFAST) {                              // node and component of the column: (c0 + lk) / q, (c0 + lk) % q
            const int jn = c0 / q < CS ? c0 / q : 0;
            dd = (c0 < IR) ? lc.tD[jt][jn] : 0.0;
            cw = (c0 < IR && lk == lc.rl[jt] - c0 % q) ? lc.tC[jt][jn] : 0.0;
          } else if constexpr (LCT::TREG) {                        // the node from its two bits; weights D_il,j / C_il,j by selection
            const unsigned jn = (lc.jnb[ct] >> (2 * v)) & 3u;
            const bool node = c0 + lk < D::P0;
            double dsel = lc.tD[jt][0], csel = lc.tC[jt][0];
#pragma unroll
            for (int jj = 1; jj < CS; jj++) { dsel = (jn == unsigned(jj)) ? lc.tD[jt][jj] : dsel; csel = (jn == unsigned(jj)) ? lc.tC[jt][jj] : csel; }
            dd = node ? dsel : 0.0;
            cw = ((lc.dbt[ct][jt] >> v) & 1u) ? csel : 0.0;
          } else {                                                 // ... or read from the tables' rows
            const int jn = int((lc.jnb[ct] >> (2 * v)) & 3u);
            constexpr int oZ4 = D::TABSZ + GR * SLOT + R::x_Z4;
            const bool node = c0 + lk < D::P0;
            dd = tabL[node ? lc.dDo[jt] + jn : oZ4];
            cw = tabL[((lc.dbt[ct][jt] >> v) & 1u) ? lc.dCo[jt] + jn : oZ4];
          }
          double val = fma(h * dd, S[lc.dco[ct][jt][v]], cw);
          if constexpr (D::p > 0) {
            if (c0 + 3 >= D::P0 && c0 < IR) {            // parameter columns: h sum_j D_il,j J_j[rl][q + pc]
              const bool par = c0 + lk >= D::P0 && lc.dco[ct][jt][v] != LCT::ZERO;   // (no entry: the zero cell, no stride)
              const int jst = par ? D::NZJ : 0;
              double ps = 0.0;
#pragma unroll
              for (int jj = 0; jj < CS; jj++) ps = fma(tDw(jt, jj), S[lc.dco[ct][jt][v] + jj * jst], ps);
              val = par ? h * ps : val;
            }
          }
          if (c0 <= T && T < c0 + 4) val -= (lk == T - c0) ? sd[jt] : 0.0;        // DC rows -+ (sum_j D_ij f_j + E_i f^_i)
          if (c0 <= TF && TF < c0 + 4) val += (lk == TF - c0) ? sd[jt] : 0.0;     // (LGLDefects.h:484-500)
          r[v] = val;
        }
      return r;
    };
    d4 accJ[TI][TJ];
    if constexpr (R::JRIDE) {                           // (read now: the T buffers of R4 lie over the CJ section)
#pragma unroll
      for (int ct = 0; ct < TI; ct++)
#pragma unroll
        for (int jt = 0; jt < TJ; jt++) accJ[ct][jt] = dc_tile(ct, jt);
    }
    auto store_J_tile = [&](int ct, int jt, const d4& acc) {
      if constexpr (ASM) {
        if (!kkt_dst) return;
#pragma unroll
        for (int v = 0; v < 4; v++) asm_put(a, kkt_dst, jmap[ct * TJ + jt][v], acc[v]);
      } else {
#pragma unroll
        for (int v = 0; v < 4; v++)
          bst(o_kkt, unsigned(jcb(ct, v) + 16 * jt + lr), 16 * jt + lr < OR && (CFULL || ct + 1 < TI || 16 * ct + lk + 4 * v < IR), acc[v]);
      }
    };
    if constexpr (!R::JRIDE) {
#pragma unroll
      for (int jt = 0; jt < TJ; jt++) {
        double bj[KS], hel = 0.0;
#pragma unroll
        for (int ct = 0; ct < TI; ct++) accJ[ct][jt] = dc_tile(ct, jt);     // (one row tile at a time: TI accumulators live, not TI TJ)
#pragma unroll
        for (int i = 0; i < KE; i++) hel = (lc.il[jt] == i) ? h * ctab.E[i] : hel;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) bj[kk] = KE > 0 ? hel * S[lc.jo[jt][kk]] : 0.0;
#pragma unroll
        for (int i = 0; i < KE; i++) {
          if (i * n >= 16 * jt + 16 || i * n + n <= 16 * jt) continue;      // (no defect row of interior i in this row tile)
#pragma unroll
          for (int kk = 0; kk < KS; kk++) {
            const double b = (lc.il[jt] == i) ? bj[kk] : 0.0;
#pragma unroll
            for (int ct = 0; ct < TI; ct++) accJ[ct][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[ct][i][kk], b, accJ[ct][jt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int ct = 0; ct < TI; ct++) store_J_tile(ct, jt, accJ[ct][jt]);
      }
    }
#pragma unroll
    for (int jt = 0; jt < TJ; jt++)
      bst(o_fx, unsigned(16 * jt + lr), lk == 0 && 16 * jt + lr < OR, fxv[jt]);
    seg_lds_sync();                                     // (every read of the sections the T buffers lie over is issued)
    RTSG();
    // ---- R4: tile column rt of H: M_i[:, rt], H(ct, rt) += DI_i[:, ct]^T M_i[:, rt]; HT and the adjoint gradient on its columns;
    //      rank-2 time update  H += d HT^T + HT d^T,  d = e_TF - e_T  (the four updates of LGLDefects.h:508-511)
    double a2[TI];                                      // A-side rank-2 fragments of the tiles done so far
#pragma unroll
    for (int rt = 0; rt < TI; rt++) {
      d4 accH[TI];                                      // tiles (ct, rt), ct <= rt
#pragma unroll
      for (int ct = 0; ct <= rt; ct++) {                // cardinal Hessian blocks (LGLDefects.h:386-402): the initial value
        if constexpr (LEVEL < 2) { accH[ct] = d4{0.0, 0.0, 0.0, 0.0}; continue; }
        const int tix = rt * (rt + 1) / 2 + ct, sh = R::sh_index(tix);
#pragma unroll
        for (int v = 0; v < 4; v++) {
          double val = sh >= 0 ? S[lc.cho[sh >= 0 ? sh : 0][v]] : 0.0;
          if constexpr (D::p > 0) {
            if (sh >= 0 && 16 * ct + 4 * v + 3 >= D::P0) {     // parameter-parameter entries: summed over the cardinal nodes
              const bool par = 16 * ct + 4 * v + lk >= D::P0 && lc.cho[sh >= 0 ? sh : 0][v] != LCT::ZERO;
              const int hst = par ? D::NZH : 0;
              double ps = 0.0;
#pragma unroll
              for (int jj = 1; jj < CS; jj++) ps += S[lc.cho[sh >= 0 ? sh : 0][v] + jj * hst];
              val = par ? val + ps : val;
            }
          }
          accH[ct][v] = val;
        }
      }
      double hi = 0.0;
      if constexpr ((LEVEL < 2 && !R::JRIDE) || !R::GROW) {   // no M product (or no g^ row in it) to take the sum from: E_i g^_i . DI_i over this lane's rows
#pragma unroll
        for (int i = 0; i < KE; i++)
#pragma unroll
          for (int kk = 0; kk < KS; kk++) {
            const int b = 4 * kk + lk;
            hi = fma(ctab.E[i] * ((b < N) ? S[D::w_Ig + i * N + (b < N ? b : 0)] : 0.0), dv[rt][i][kk], hi);
          }
        hi = rows4_sum(hi);
      }
      // (all K M products first, then their uses: a result is read -- written to T_i, fed to the H products -- K - 1 products
      //  after it was issued instead of right behind it, where the wave would sit out the 64 cycles of the instruction)
      d4 Mi[K];
#pragma unroll
      for (int i = 0; i < KE; i++) {
        if constexpr (LEVEL < 2 && !R::JRIDE) continue;
        Mi[i] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KS; kk++) Mi[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ah[i][kk], dv[rt][i][kk], Mi[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < KE; i++) {
        if constexpr (LEVEL < 2 && !R::JRIDE) continue;
        if constexpr (R::GROW) hi += Mi[i][R::vN];   // entry v: row lk + 4v of M_i (row N: E_i g^_i . DI_i), column 16rt + lr
        if constexpr (R::JRIDE) {        // rows JR0 .. JR0 + n - 1: J_i[r][16rt + lr] -> T_i[lr][r]
          lds_double* const Tbase = R::T_XTRA ? xtra + R::x_T : (lds_double*)S;   // the buffers: in the slot's dead sections, or the wave's own
          lds_double* const Ti = Tbase + R::t_off(i) + lr * n;
#pragma unroll
          for (int v = 0; v < 4; v++) {
            if (4 * v + 3 < R::JR0 || 4 * v > R::JR0 + n - 1) continue;   // (no lane has such a row in this entry)
            const int r = lk + 4 * v - R::JR0;
            if constexpr (R::t_dummy >= 0) {
              Tbase[(r >= 0 && r < n) ? R::t_off(i) + lr * n + r : R::t_dummy + lr] = Mi[i][v];
            } else if (r >= 0 && r < n) Ti[r] = Mi[i][v];
          }
        }
        if constexpr (LEVEL >= 2) {
#pragma unroll
          for (int kk = 0; kk < KS; kk++)
#pragma unroll
            for (int ct = 0; ct <= rt; ct++) accH[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[ct][i][kk], Mi[i][kk], accH[ct], 0, 0, 0);
        }
      }
      RTSF();
      if constexpr (R::JRIDE) {          // J^T tile rt: entry v <-> (column 16rt + lk + 4v, defect row jr = lr = (il, rl))
        seg_lds_sync();
        RTSF();
#pragma unroll
        for (int jt = 0; jt < TJ; jt++) {
          int tb = R::t_off(0);
#pragma unroll
          for (int i = 1; i < K; i++) tb = (lc.il[jt] == i) ? R::t_off(i) : tb;
          const lds_double* const Tr = (R::T_XTRA ? (const lds_double*)(xtra + R::x_T) : S) + tb + lk * n + lc.rl[jt];
          d4 acc = accJ[rt][jt];
#pragma unroll
          for (int v = 0; v < 4; v++) {
            acc[v] += Tr[4 * v * n];   // (a lane without a defect row has il = rl = 0: a valid address, and what it adds up is never
                                       //  stored -- no load under an exec mask, no select)
          }
          store_J_tile(rt, jt, acc);
        }
      }
      RTSF();
      // column role (lanes lk == lkN hold row N of the M tiles): full time-partial vector HT (LGLDefects.h:403-411, 504-505)
      // and the adjoint gradient  g = J^T lam = h sum_i E_i g^_i^T DI_i + DC^T lam  (LGLDefects.h:512) of column 16rt + lr
      double htv = 0.0;
      if (R::BSTORE || lk == R::lkN) {                   // (computed by every lane -- the loads are in range for all of them -- and
        const int c = 16 * rt + lr;                     //  used / stored from the lanes lk == lkN only)
        double gs = S[lc.cgg[rt]];
        if constexpr (D::p > 0) {
          if (16 * rt + 15 >= D::P0) {                  // a parameter column: g_j summed over the nodes
            const bool par = c >= D::P0 && c < IR;
            const int gst = par ? N : 0;
            double ps = 0.0;
#pragma unroll
            for (int jj = 1; jj < CS; jj++) ps += S[lc.cgg[rt] + jj * gst];
            gs = par ? gs + ps : gs;
          }
        }
        if constexpr (LEVEL >= 2) {
          htv = fma(gs, rh, hi);
        }
        bst(o_agx, unsigned(c), lk == R::lkN && (CFULL || c < IR), fma(h, hi + agJ[rt], fma(tsA(rt), sls, tabL[lc.clo[rt]])));
      }
      if constexpr (LEVEL >= 2) {
        // the column's value from its lk == lkN lane: one cross-lane read instead of an LDS write, a wait and a read (only lanes
        // lk = 0, 1 use it)
        const double ht = row_to_rows01<R::lkN>(htv, lr);
        // k-step of the rank-2 product: lanes lk == 0 carry (d, HT), lanes lk == 1 (HT, d), the others zeros -- as arithmetic with
        // the lane's 0 / 1 weights (nested selects on doubles came out of the compiler as branches)
        const double w0 = lk == 0 ? 1.0 : 0.0, w1 = lk == 1 ? 1.0 : 0.0, ts = tsA(rt);
        a2[rt] = fma(w0, ts, w1 * ht);
        const double b2 = fma(w0, ht, w1 * ts);
#pragma unroll
        for (int ct = 0; ct <= rt; ct++) accH[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[ct], b2, accH[ct], 0, 0, 0);
      }
      RTSF();
      if constexpr (ASM) {
        if (kkt_dst && LEVEL >= 2) {
#pragma unroll
          for (int ct = 0; ct <= rt; ct++)
#pragma unroll
            for (int v = 0; v < 4; v++) asm_put(a, kkt_dst, hmap[rt * (rt + 1) / 2 + ct][v], accH[ct][v]);
        }
      } else {
#pragma unroll
        for (int ct = 0; ct < rt; ct++)                 // tiles left of the diagonal: every column < IR
#pragma unroll
          for (int v = 0; v < 4; v++) bst(o_h, unsigned(hcb(ct, v) + 16 * rt + lr), CFULL || 16 * rt + lr < IR, accH[ct][v]);
#pragma unroll
        for (int v = 0; v < 4; v++)                     // diagonal tile: r >= c
          bst(o_h, unsigned(hcb(rt, v) + 16 * rt + lr), lr >= lk + 4 * v && (CFULL || rt + 1 < TI || 16 * rt + lr < IR), accH[rt][v]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    RTSG();
  };
  for (int g = 0; g < gcount; g++) segment(g);
  RTS();
  seg0 += gcount;
  o_seg0 += o_gcount;
  wave_lds_sync();                     // (the next group's gather rewrites the slots)
  }
#if defined(ASSET_TIMING)
  if (blockIdx.x == 7 && wv == 0 && lane0 == 0 && a.FX)
    for (int t = 0; t + 1 < nts; t++) a.FX[size_t(wg_first) * OR + t] = double(tstamp[t + 1] - tstamp[t]);
#endif
#if defined(ASSET_WALLCLOCK)   // (tuning builds) 100 MHz wall-clock at the start and the end of every wave, left in FX
  if (lane0 == 0 && a.FX && wg_count > 0) {
    a.FX[size_t(wg_first) * OR + 0] = double(wall_t0);
    a.FX[size_t(wg_first) * OR + 1] = double(wall_clock64());
    a.FX[size_t(wg_first) * OR + 2] = double(__builtin_amdgcn_s_getreg(63492));   // HW_ID
    a.FX[size_t(wg_first) * OR + 3] = double(__builtin_amdgcn_s_getreg(63508));   // XCC_ID
  }
#endif
#undef RTS
#undef RTSG
}

// FORM 1: the row-wise dense part for a shape whose default form is tiles (ResDims::RD_ALT; one-group kernel, level 2, blocks)
template <class Ode, int SCH, bool BLOCKED, int LEVEL = 2, bool ASM = false, bool LOOP = false, bool GIVEN = false, bool LPAIR = false, int FORM = -1>
__global__ __launch_bounds__((!GIVEN && ResDims<Dims<Ode, SCH, BLOCKED>>::PAIR && (!LOOP || LPAIR)) ? 128 : 64, (ResDims<Dims<Ode, SCH, BLOCKED>>::WPS))
void lgl_resident_kernel(EvalArgs a) {
#if defined(ASSET_EXP_NULL)   // (experiment: the cost of the launch itself)
  if (a.nseg > 0) return;
#endif
  using R = ResDims<Dims<Ode, SCH, BLOCKED>>;
  if constexpr (FORM == 1) {
    if constexpr (R::OK && R::RD_ALT && LEVEL == 2 && !ASM && !LOOP && !GIVEN && !LPAIR) lgl_resident_body<Ode, SCH, BLOCKED, 2, false, false, false, false, true>(a);
  } else if constexpr (LPAIR) {
    if constexpr (R::OK && R::LOOP_PAIR) lgl_resident_body<Ode, SCH, BLOCKED, LEVEL, ASM, LOOP, GIVEN, true, true>(a);
  } else if constexpr ((GIVEN ? R::GIVEN_OK : R::OK) && (!ASM || GIVEN || R::ASM_OK)) lgl_resident_body<Ode, SCH, BLOCKED, LEVEL, ASM, LOOP, GIVEN>(a);
}

}  // namespace asset_hip
