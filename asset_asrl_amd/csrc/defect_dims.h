// Sizes, memory maps and ODE accessors shared by the LGL defect kernels (defect_kernels.h).
//
// Notation: n = XV states, m = UV controls, p = PV parameters (BlockConstant control:
// m := 0, p := UV + PV, Blocked_ODE_Wrapper.h:7-27), q = n + 1 + m, N = q + p ODE inputs, CS cardinal nodes,
// K = CS - 1 interior points, IR = CS*q + p segment inputs, OR = K*n defect rows.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "lgl_tables.h"

#define ASSET_ODE_WAVES_PER_SIMD 1    // register budget of the ODE-stage kernel
#define ASSET_DENSE_WAVES_PER_SIMD 2  // register budget of the dense-phase kernel: 512 / 2 = 256 per lane

#define ASSET_LANE_REPLICAS 64        // copies of the dense stage's per-lane constant table (capi.hip)

// Measurement builds (tools/build_one.py with -DASSET_TUNING_BUILD): clock stamps written over FX / AGX (ASSET_TIMING,
// ASSET_WALLCLOCK, ASSET_FUNC_TIMING) and elimination experiments that compute WRONG results on purpose (ASSET_EXP_*).  A
// production build that defines one of them -- through ASSET_HIP_JIT flags or a build environment -- is refused.
#if (defined(ASSET_TIMING) || defined(ASSET_WALLCLOCK) || defined(ASSET_FUNC_TIMING) || defined(ASSET_EXP_NOWS) ||       \
     defined(ASSET_EXP_NULL) || defined(ASSET_EXP_ODEREP) || defined(ASSET_EXP_ONEUNIT) || defined(ASSET_EXP_UNITREP) || defined(ASSET_EXP_ROWS)) && \
    !defined(ASSET_TUNING_BUILD)
#error "ASSET_TIMING / ASSET_WALLCLOCK / ASSET_EXP_* change what the kernels write: measurement builds only (-DASSET_TUNING_BUILD)"
#endif

namespace asset_hip {

static __constant__ LglTab d_lgl_tab[4] = ASSET_LGL_TABLE_INIT;

struct EvalArgs {
  int nseg;
  const double* X;     // NLP primal vector (device)
  const double* L;     // equality multipliers (device); unused for value-only
  const int* vindex;   // [IR x nseg] column-major (device)
  const int* cindex;   // [OR x nseg] column-major (device)
  double* FX;          // [nseg x OR] blocks or null
  double* AGX;         // [nseg x IR] blocks or null
  double* KKT;         // [nseg x KSTRIDE] blocks (layout: Dims::KL below) or null
  double* work;        // [grid][G][SLOT] per-workgroup ODE result slots (L2-resident scratch in HBM)
  // On-device assembly (dense stage, ASM kernels): KKT entries go into the solver's value array instead of being
  // stored as blocks (DenseFunctionBase.h:1413-1523 KKTFillAll / KKTFillJac; locations NonLinearProgram.cpp:316-330;
  // the reference serialises clashing columns with mutexes, KKTClashes / KKTLocks).  kmap holds, per segment and in
  // accumulator-fragment order, one entry per lane and accumulator entry:  m >= 0 -- value location used by this slot
  // alone: the entry is STORED (the caller hands over zeros there, as the reference zeroes the KKT values before every
  // evaluation, PSIOPT.cpp:107);  m <= -2 -- location -(m+2) is shared with other slots of this constraint (boundary
  // nodes of adjacent segments, phase parameters): no-return f64 atomic add;  -1 -- no KKT slot.  In accumulate mode
  // every slot is encoded as shared, which makes the evaluation a true += at about 1.4e11 atomics/s.
  const int* kmap = nullptr;
  double* values = nullptr;
  // Locations that three or more slots share (entries between phase parameters: one contribution per segment) are not
  // added to atomically -- the order of the additions, hence the last bits of the sum, would change from run to run --
  // but STAGED: the map names such a location as nvalues + k, slot k of `stage`, every staged slot owns one cell, and
  // asm_reduce_kernel sums the cells of each location in slot order afterwards (capi.hip).  Locations with exactly two
  // contributors (boundary nodes of adjacent segments) keep the atomic: a + b = b + a, bit for bit.
  double* stage = nullptr;
  int nvalues = 0;
  // per-lane constants of the dense stage, computed once per handle (defect_kernels.h: LaneConsts, lane_setup_kernel)
  const void* lane_consts = nullptr;
  const void* lane_consts_res = nullptr;   // the record of the resident kernel (defect_resident.h: ResLane)
  // Index tables whose rows are runs -- vindex[s][k] = aff_v0 + s aff_vs + k, cindex[s][k] = aff_c0 + s aff_cs + k, what a
  // phase without BlockConstant controls and ODE parameters produces (PhaseIndexer.cpp:361-372, 179-189) -- are recognised
  // when the handle is created: the resident kernel then forms the addresses of z and lam itself instead of loading them
  // (one dependent memory round trip less at the start of every wave).
  int affine = 0, aff_v0 = 0, aff_vs = 0, aff_c0 = 0, aff_cs = 0;
  // plain functions (func_kernels.h): constants of every application, [nseg][F::NACONST] (vf.ApplConst) or null
  const double* appl_consts = nullptr;
  // bit 0 (ASSET_HIP_KEEP_HESSIAN_SLOTS, Jacobian kinds): the Hessian slots of the KKT blocks are not written at all
  // instead of being written as zeros -- KKTFillJac (DenseFunctionBase.h:1468-1523) never reads them
  int flags = 0;
  // Heavy right-hand sides: segments per group of the unit kernels that wrote the workspace slots (0: unknown).  The dense
  // part behind them then takes every group's segments on the XCD whose L2 holds them (defect_resident.h, GIVEN).
  int units_gp = 0;
};

// ---------------------------------------------------------------------------------------------- sizes
template <class Ode, int CS_, bool BLOCKED_>
struct Dims {
  // CS_ is the transcription id of the C ABI: 2, 3, 4 = LGL3 / LGL5 / LGL7 (the number of cardinal nodes); 1 = Trapezoidal,
  // which runs through the same kernels as a two-node scheme whose interior point has weight E = 0 and is never
  // evaluated (lgl_tables.h): its interior sections of the workspace stay at the zeros they are created with.
  static constexpr bool TRAP = (CS_ == 1);
  static constexpr int TAB = TRAP ? 3 : CS_ - 2;        // index into d_lgl_tab
  static constexpr int CS = TRAP ? 2 : CS_, K = CS - 1;
  static constexpr int n = Ode::XV;
  static constexpr int m = BLOCKED_ ? 0 : Ode::UV;                    // Blocked_ODE_Wrapper.h:7-27
  static constexpr int p = BLOCKED_ ? Ode::UV + Ode::PV : Ode::PV;
  static constexpr int q = n + 1 + m;
  static constexpr int N = q + p;
  static constexpr int T = n;
  static constexpr int IR = CS * q + p;                               // TranscriptionSizing.h:7-14
  static constexpr int OR = K * n;
  static constexpr int TF = q * (CS - 1) + T;
  static constexpr int P0 = CS * q;
  static constexpr int NKKT = IR * (IR + 1) / 2 + OR * IR;            // DenseFunctionBase.h:1070-1088
  static constexpr int NH = N * (N + 1) / 2;                          // packed lower ODE Hessian
  static constexpr int IRP = (IR + 15) / 16 * 16;
  static constexpr int ORP = (OR + 15) / 16 * 16;
  static constexpr int NP = (N + 3) / 4 * 4;                          // rows of one interior's DI tile (MFMA k = 4)
  static constexpr int KS = NP / 4;                                   // k-steps per interior
  static constexpr int MT = (N + 1 + 15) / 16;                        // 16-wide column tiles of [hE H^ | E g^]
  static constexpr int TI = IRP / 16, TJ = ORP / 16;
  static constexpr int NTH = TI * (TI + 1) / 2;                       // lower-triangle H tiles
  static constexpr int CW = IRP <= 16 ? 16 : (IRP <= 32 ? 32 : 64);   // lanes per DI row pass (power of two)

  // ---- per-segment slot of ODE results (in doubles): workspace layout in HBM, copied verbatim into LDS by the dense
  //      stage.  J and H blocks hold only their structural non-zeros, in Ode::JIDX / HIDX order (Ode::JPOS / HPOS
  //      give the position of a dense entry or -1); readers point "no entry" at a zero cell instead.
  using ode_t = Ode;
  static constexpr int NZJ = Ode::NNZ_J, NZH = Ode::NNZ_H;
  static constexpr int w_z = 0;
  static constexpr int w_lam = w_z + IR;
  static constexpr int w_Cf = w_lam + OR;
  static constexpr int w_CJ = w_Cf + CS * n;
  static constexpr int w_Cg = w_CJ + CS * NZJ;
  static constexpr int w_CH = w_Cg + CS * N;
  static constexpr int w_If = w_CH + CS * NZH;
  static constexpr int w_IJ = w_If + K * n;
  static constexpr int w_Ig = w_IJ + K * NZJ;
  static constexpr int w_IH = w_Ig + K * N;
  static constexpr int w_SV = w_IH + K * NZH;           // transcendental values of f at the cardinal nodes (P1 -> P3)
  static constexpr int WSLOTD = w_SV;                   // what the dense stage reads of a slot
  // ---- dense scratch (one segment at a time)
  // DI_i is kept as two tiles: state rows (r < n), rewritten for every segment, and the remaining rows (tau / control
  // / parameter / padding), constant per launch.  M^T is produced after every A fragment has been read into
  // registers, so it re-uses the state-row tile's memory.
  static constexpr int TABSZ = (sizeof(LglTab) + 7) / 8;   // LDS copy of the scheme's weight tables
  static constexpr int LDM = K * NP + 1;               // M is stored column-major [IRP][LDM]: conflict-free MFMA write-back
  static constexpr int LDC = IRP + 4;                  // DC row stride: row- and column-wise fragment reads both conflict-free
  static constexpr int XM_ALL = (K * n * IRP > IRP * LDM) ? K * n * IRP : IRP * LDM;
  // WIDE: with all of M^T and the DC tile the working set of one segment exceeds the 160 KiB of a CU (32 states in
  // LGL7: M^T 124 KB + DC 113 KB beside DI).  Such shapes run the dense stage as one four-wave workgroup per CU
  // (defect_wide.h) that keeps only DI resident: every wave produces the 16 columns of M_i it is about to use in
  // registers, and the cardinal part of J is formed from the slot where it is used instead of being kept as a tile
  // (s_DC then holds the time-column vector, one entry per defect row, and the multiplier weights of J^T lam).
  // Shapes from IR = 64 on take the same kernel even though they fit: holding their accumulators, fragments and lane
  // constants in one wave spills (32 states in LGL5, 10 000 segments: 6.7 ms single-wave, 1.2 ms four-wave; LGL3:
  // 0.64 -> 0.38 ms), while at IR = 40 (TwoBody-LGL7) the single-wave layout is still twice as fast.
#define ASSET_WIDE_MIN_IR 64
  static constexpr bool WIDE =
      size_t(TABSZ + WSLOTD + XM_ALL + K * (NP - n) * IRP + ORP * LDC + 4 * IRP + 2) * 8 > 160 * 1024 ||
      (!TRAP && IR >= ASSET_WIDE_MIN_IR);
  // ---- layout of a KKT block in memory (round 6).  The order of a block's slots is private to the function: the solver sees
  // (row, col) per slot through getKKTSpace -- a method the function itself implements (SolverInterfaceSpecs.h:41-92) -- and maps
  // each to its location in the matrix whatever the order (NonLinearProgram.cpp:282-330).  KL 0: the order of the reference's
  // dense functions, `for c: { H(r, c), r >= c ; J(j, c) }` (DenseFunctionBase.h:1112-1123), stride NKKT.  KL 1 (narrow shapes):
  // the Jacobian, column-major, then the packed lower triangle of H, column-major, each region a whole number of 128-byte lines:
  //     J(j, c) at  c OR + j ,       H(r, c), r >= c, at  HOFF + c IR - c (c - 1) / 2 + (r - c) ,      stride KSTRIDE
  // A dense part that writes the rows of [J ; g^T] and the rows of H in different passes (defect_rowdpp.h) then never writes a
  // 32-byte sector from two passes -- in the reference's order a block column's H part and J part share sectors, which reached
  // memory once per pass (WRITE_SIZE 1.29-1.38 x the block bytes, profiles/r5_*_pmc.json).  The handle exports the order
  // (asset_hip_defect_kkt_layout); padding slots are never written and never read.
#ifndef ASSET_KKT_LAYOUT
#define ASSET_KKT_LAYOUT 1
#endif
  static constexpr int KL = WIDE ? 0 : ASSET_KKT_LAYOUT;
  static constexpr int JREG = (OR * IR + 15) / 16 * 16, HREG = (IR * (IR + 1) / 2 + 15) / 16 * 16;
  static constexpr int KSTRIDE = KL ? JREG + HREG : NKKT;
  static constexpr int HOFF = KL ? JREG : 0;
  static constexpr int HCA = KL ? IR - 1 : IR + OR - 1;                  // hcol(c) = HOFF + c HCA - c (c - 1) / 2
  static constexpr int hcol(int c) { return HOFF + c * HCA - c * (c - 1) / 2; }          // H(r, c) sits at hcol(c) + r
  static constexpr int jcol(int c) { return KL ? c * OR : hcol(c) + IR; }                // J(j, c) sits at jcol(c) + j
  // (Measured and rejected, round 6 -- profiles/r6_aligned.txt: every block column on 128-byte lines of its own, the triangles'
  //  columns starting on 32-byte sectors, i.e. whole-line stores from every 16-lane group.  The store path likes it -- TwoBody-LGL5-
  //  BlockConstant x 10 000 21.0 -> 20.1 us although WRITE_SIZE grows 49.9 -> 60.5 MB -- but the padding is bytes: Reentry-LGL7 x 10 000
  //  27.5 -> 29.8 us (84.8 -> 101.7 MB), x 100 000 280 -> 360 us, TwoBody x 100 000 200 -> 260 us.  The packed regions stay.)
  static constexpr int WNW = 4;                        // waves of the wide dense kernel
  static constexpr int NCR = WIDE ? N - n : NP - n;    // constant rows per interior (wide: the k-padding rows read a zero row)
  static constexpr int s_DIx = 0;                      // [K][n][IRP]
  static constexpr int s_M = 0;                        // M^T [IRP][K*NP+1], aliases s_DIx  (wide: M stays in registers)
  static constexpr int XM = WIDE ? K * n * IRP : XM_ALL;
  static constexpr int s_DIc = XM;                     // [K][NCR][IRP]
  static constexpr int s_DC = s_DIc + K * NCR * IRP;   // cardinal part of J, rows = defect rows  [ORP][LDC], padding rows zero
  static constexpr int s_WL = s_DC + ORP;              // wide: sum_i D_ij lam_(i,r)  [CS][n], then the work counter
  static constexpr int s_CNT = s_WL + CS * n;
  static constexpr int s_R2 = s_DC + (WIDE ? ORP + CS * n + 2 : ORP * LDC);   // rank-2 time rows: [0] = d = e_TF - e_T (constant), [1] = HTpar, [2] = 0
  static constexpr int s_HI = s_R2 + 3 * IRP;          // sum_i E_i g^_i^T DI_i     [IRP]
  static constexpr int s_Z0 = s_HI + IRP;              // a cell that always holds 0.0: target of every "no entry" offset
  static constexpr int s_JP = s_Z0 + 2;                // wide: Ode::JPOS / HPOS as 16-bit LDS tables (run-time look-ups)
  static constexpr int s_HP = s_JP + (WIDE ? (n * N + 3) / 4 : 0);
  static constexpr int SCRATCH = s_HP + (WIDE ? (NH + 3) / 4 : 0);

  // ---- ODE-phase staging: every evaluating lane writes its dense J (n x N) and packed H into an LDS row, the wave
  //      then copies the rows to the workspace with coalesced stores.  Row stride is odd: conflict-free ds_write.
  static constexpr int NSTG = NZJ + NZH;               // a staging row holds one point's non-zeros [J | H]
  static constexpr int STG_LD = NSTG | 1;
  static constexpr int DENSE = WSLOTD + SCRATCH;         // slot buffer + dense scratch (staging aliases both)
  // lanes per ODE pass: as many as fit in the LDS the dense phase needs anyway (occupancy is LDS-bound)
#define ASSET_LC_BUDGET (64 * 1024)
  // very wide ODEs: no LDS row fits -> the evaluating lanes write J/H straight to the workspace (uncoalesced, correct)
  static constexpr bool STAGED = (16 * STG_LD * 8 <= ASSET_LC_BUDGET);
  static constexpr int LC = !STAGED ? 64
                            : (64 * STG_LD * 8 <= ASSET_LC_BUDGET) ? 64 : ((32 * STG_LD * 8 <= ASSET_LC_BUDGET) ? 32 : 16);
  // ODE-stage hand-offs: what a later phase of the stage reads of an earlier one (z, lam, f_j, g^_i, the saved
  // transcendentals) is mirrored in LDS, one short slot per segment of the group, so that the phases are separated by
  // an LDS wait instead of a drain of the wave's global stores plus a round trip to L2 (the workspace copy is still
  // written: the dense stage reads it).  Only when it fits beside the staging rows at four workgroups per CU.
  static constexpr int GM = 64 / CS;                   // segments per group of the ODE stage (build.py: pick_group)
  static constexpr int m_z = 0, m_lam = IR, m_Cf = m_lam + OR, m_Ig = m_Cf + CS * n, m_SV = m_Ig + K * N;
  static constexpr int MSLOT = (m_SV + CS * Ode::NSAVE) | 1;
  // (not for Trapezoidal: its interior sections are never written and must read as the zeros the workspace is created with)
  static constexpr bool MIRROR = !TRAP && STAGED && size_t(TABSZ + LC * STG_LD + GM * MSLOT) * 8 <= 40 * 1024;
  static constexpr int WSLOT = MIRROR ? w_SV : w_SV + CS * Ode::NSAVE;   // (the saved values live in the mirror when there is one)
  // Fused single launch (defect_kernels.h, STAGE 3): the ODE stage of a wave's own GF segments runs in the LDS the
  // dense phase needs anyway (rows of GF*CS points and GF mirror slots alias the slot buffer and the dense scratch).
  static constexpr int GF_FIT = int((size_t(DENSE) * 8) / (size_t(CS * STG_LD + MSLOT) * 8));
  static constexpr int GF = GF_FIT < 64 / CS ? GF_FIT : 64 / CS;
  static constexpr bool FUSED = MIRROR && STAGED && LC == 64 && !WIDE && GF >= 2;
  // two-wave form (STAGE 4): the pair's ODE stage -- rows of GF2*(K+CS) points (interior and cardinal regions side by
  // side) and GF2 mirror slots -- in the two waves' bodies
  static constexpr int GF2_FIT = int((size_t(2 * DENSE) * 8) / (size_t((K + CS) * STG_LD + MSLOT) * 8));
  static constexpr int GF2 = (GF2_FIT < 64 / CS ? GF2_FIT : 64 / CS) & ~1;   // even: both waves get GF2/2
  static constexpr bool FUSED2 = FUSED && GF2 >= 4;
  static constexpr size_t lds_bytes_fused2() { return size_t(2 * (TABSZ + DENSE)) * 8; }
  // LDS of the two launches: [weight tables | staging rows | mirror] and [weight tables | slot buffer | dense scratch]
  static constexpr size_t lds_bytes_ode() { return size_t(TABSZ + (STAGED ? LC * STG_LD : 0) + (MIRROR ? GM * MSLOT : 0)) * 8; }
  static constexpr size_t lds_bytes_dense() { return size_t(TABSZ + DENSE) * 8; }
  static constexpr size_t lds_bytes() { return lds_bytes_ode() > lds_bytes_dense() ? lds_bytes_ode() : lds_bytes_dense(); }
};

// Which (derivative level, stage) variants of lgl_defect_kernel a shape has.  The launcher (registry.h) uses no other; a
// run-time compiled module (rtc_device.h) names every variant and gets the others as empty kernels.
template <class D, int LEVEL, int STAGE>
constexpr bool lgl_variant_valid() {
  if (STAGE == 1) return true;                 // ODE stage (and the whole value-only kind): every shape
  if (D::WIDE) return false;                   // wide shapes: dense stage in defect_wide.h
  if (STAGE == 2) return LEVEL >= 1;
  if (STAGE == 3) return D::FUSED && LEVEL >= 1;
  if (STAGE == 4) return D::FUSED2 && LEVEL == 2;
  return false;
}

using d4 = __attribute__((ext_vector_type(4))) double;

// ---------------------------------------------------------------------------------------------- ODE accessors
template <class D, class ZP = const double*>
struct CardIn {  // y = [z_j (q), P (p)] read from the slot's copy of z; lam = adjoint weights in registers
  ZP z;
  const double* w;
  int j;
  ZP sv = nullptr;             // saved transcendental values of f at this node (fjgh_load)
  __device__ double y(int i) const { return i < D::q ? z[j * D::q + i] : z[D::P0 + (i - D::q)]; }
  __device__ double lam(int k) const { return w[k]; }
  __device__ double saved(int k) const { return sv[k]; }
};
template <class D>
struct RegIn {
  const double* yv;
  const double* lv;
  __device__ double y(int i) const { return yv[i]; }
  __device__ double lam(int k) const { return lv[k]; }
};
// LDS-address-space pointer: stores through it are ds_write (tracked by lgkmcnt only), never flat
typedef __attribute__((address_space(3))) double lds_double;
// global-address-space pointer: stores through it are global_store even inside an out-of-line device function (a
// generic pointer there means flat_store, which also occupies the LDS counter)
typedef __attribute__((address_space(1))) double glb_double;

template <class D, bool ACCG = false, bool MIR = false, bool MF = false, bool MG = false>
struct OdeOutStaged {  // f, g -> workspace slot (few scalars); J, H non-zeros -> this lane's LDS staging row [J | H]
  using JP = std::conditional_t<D::STAGED, lds_double*, double*>;
  using SP = std::conditional_t<MIR, lds_double*, double*>;     // MIR: what a later phase reads goes to the LDS mirror
  double* f_;
  double* g_;
  JP J_;
  JP H_;
  SP sv_ = nullptr;
  lds_double* f2_ = nullptr;                           // mirror copy of f (MF) / of g (MG)
  lds_double* g2_ = nullptr;
  const double* lamv_ = nullptr;                       // ACCG: multipliers of this point's defect rows ...
  double gacc_[ACCG ? D::N : 1];                       // ... and g^ = J^^T lam accumulated while J is emitted (fj has no g)
  __device__ void f(int k, double v) {
    f_[k] = v;
    if constexpr (MF) f2_[k] = v;
  }
  __device__ void J(int k, int i, double v) {          // (k, i) are literals in the generated bodies: the lookup folds
    const int c = D::ode_t::JPOS[k * D::N + i];
    if (c >= 0) {
      J_[c] = v;
      if constexpr (ACCG) gacc_[i] += lamv_[k] * v;
    }
  }
  __device__ void g(int i, double v) {
    g_[i] = v;
    if constexpr (MG) g2_[i] = v;
  }
  __device__ void H(int i, int j, double v) {
    const int c = D::ode_t::HPOS[i * (i + 1) / 2 + j];
    if (c >= 0) H_[c] = v;
  }
  __device__ void save(int k, double v) { sv_[k] = v; }
};

template <class D>
__device__ inline auto stage_or(lds_double* row, double* slot) {
  if constexpr (D::STAGED) { (void)slot; return row; } else { (void)row; return slot; }
}

// true when block columns [16ct,16ct+16) and rows [16rt,16rt+16) can hold a cardinal diagonal / parameter entry
template <class D>
__device__ constexpr bool tiles_share_node(int ct, int rt) {
  const int c0 = 16 * ct, c1 = (16 * ct + 15 < D::IR - 1) ? 16 * ct + 15 : D::IR - 1;
  const int r0 = 16 * rt, r1 = (16 * rt + 15 < D::IR - 1) ? 16 * rt + 15 : D::IR - 1;
  if (c0 > c1 || r0 > r1) return false;
  if (D::p > 0 && r1 >= D::P0) return true;                 // parameter rows couple to every column
  const int jc0 = c0 / D::q, jc1 = c1 / D::q, jr0 = r0 / D::q, jr1 = r1 / D::q;
  return !(jr1 < jc0 || jc1 < jr0);
}

// Assembled store of one KKT entry through its map word (EvalArgs::kmap): m >= 0 plain store, -1 dropped, m <= -2 shared
// location -(m + 2): atomic add, or (>= nvalues) its own staging cell.
__device__ inline void asm_put(const EvalArgs& a, double* values, int m, double v) {
  if (m >= 0) values[m] = v;
  else if (m != -1) {
    const int loc = -(m + 2);
    if (loc < a.nvalues) unsafeAtomicAdd(values + loc, v);       // global_atomic_add_f64, no return
    else a.stage[loc - a.nvalues] = v;
  }
}

// One wave per workgroup.  LDS instructions of a wave execute in issue order, so LDS hand-offs between lanes only
// need the compiler kept from reordering (and the reads returned); crucially this does NOT wait for outstanding
// global stores the way __syncthreads() (vmcnt(0)) does -- the block stores of a segment drain behind the next one.
__device__ inline void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// The same hand-off as an ordering only: the LDS unit executes a wave's instructions in issue order -- a read issued after a write
// sees it, a write issued after a read does not disturb it, whether or not the read's data has come back -- so nothing has to be
// waited for; the compiler must keep the order of issue, and the fence (wavefront scope: no instruction) tells it to.  Unlike
// wave_lds_sync() it does not drain the wave's LDS reads in flight: each of those was a full LDS latency with nothing to issue.
__device__ inline void wave_lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
// Waits until every outstanding vector-memory operation of the wave has completed (s_waitcnt vmcnt(0) as a compiler
// builtin, so the waitcnt insertion pass knows that no load is pending afterwards).  gfx9 counts loads and stores in
// the one in-order vmcnt: a wait for a load that was issued before some stores whose number the compiler cannot count
// (stores under lane conditions) becomes vmcnt(0) and drains those stores too.  Loops that prefetch while they store
// therefore call this once per iteration, after the last load has been issued and before the first store: the loads
// are then known to have landed and nothing later in the iteration waits on vmcnt, so the stores drain behind the
// next iteration's compute.   encoding: vmcnt = 0 (bits 3:0 and 15:14), expcnt = 7 (6:4), lgkmcnt = 15 (11:8)
__device__ inline void wave_loads_landed() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// The per-segment fence of the dense stage's store phase (defect_kernels.h), OFF by default.  Measured (round 2): with
// the fence the prefetched slot is consumed without a wait at the top of the next iteration, yet a 10 000-segment phase
// runs no faster (two launches 53.3 vs 53.4 us, fused 43.1 vs 42.9 us) and large meshes run 6-7 % slower (100 000
// segments 537 vs 506 us, 1 000 000 segments 5.01 vs 4.66 ms): a wave's time in the store phase is the issue of its
// own stores (about 2.4 bytes per cycle and wave, tools/ubench_store.hip), not the wait the compiler places.
#define ASSET_STORE_FENCE 0
__device__ inline void wave_store_fence() { if constexpr (ASSET_STORE_FENCE != 0) wave_loads_landed(); }
// Hand-offs through the global workspace (same wave writes, then reads): wait for the stores as well.
__device__ inline void wave_mem_sync() { __syncthreads(); }

}  // namespace asset_hip
