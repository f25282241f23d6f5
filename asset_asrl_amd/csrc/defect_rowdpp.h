// The dense part of the resident kernel BY OUTPUT ROWS, sixteen lanes to a row group, without matrix instructions (round 5).
//
// defect_rows.h showed for the wide shapes what the node-wise form of DI_i buys (LGLDefects.h:417-458):
//     DI_i[:, column (j, cc)] = w_i(j, cc) e_cc + h B_ij dfdy_j[:, cc]      (+ the time-column terms)
// With a lane per output row r:  d_i = DI_i[:, r],  M_i = hE_i H^_i d_i,  BM_j = sum_i B_ij M_i[0:n]  and
//     H(r, (j, cc)) = sum_i w_i(j, cc) M_i[cc] + h sum_a dfdy_j[a][cc] BM_j[a] + cardinal / time terms
// -- one FMA per structural entry of a column of the sparse cardinal Jacobian.  What kept that form away from the narrow shapes
// was operand delivery: the multiplier dfdy_j[a][cc] does not depend on the lane, and a v_readlane pair or an LDS broadcast per
// operand costs as much as the FMA it feeds.  gfx950 has the answer in the instruction itself:
//     v_fmac_f64_dpp  acc, op, x  row_newbcast:k      acc += (lane k of op's own 16-lane row) * x
// ("DP ALU DPP": tools/ubench_dpp.hip -- the rate of the plain v_fmac_f64, 4.9 cycles per wave alone, 2.5 per instruction with
// four waves on the SIMD).  So: entry e of the segment's slot lives in lane e % 16 of operand register e / 16 of EVERY ROW of the
// wave -- one ds_read_b64 per sixteen operands, no delivery instruction at all -- and because the broadcast is per row, the four
// rows of a wave work on four DIFFERENT segments.  A task is (segment, group of sixteen output rows):
//   * H task (g, rg): lane lr <-> row r = 16 rg + lr of the lower triangle of H; columns c <= r;
//   * C task (g, crg): lane lr <-> defect row jr = 16 crg + lr of J, the lane behind the last defect row <-> the adjoint gradient.
//     Every row of a C task is "the adjoint gradient for a multiplier vector": e_(i0, r0) for the defect row (i0, r0), lam for the
//     gradient row -- J^T e = that row of J (LGLDefects.h:512 read row-wise), so one code path serves both:
//         M_i = hE_i J^_i^T l_i ,  BM_j = sum_i (B_ij M_i + D_ij l_i) ,  out(j, cc) = sum_i (A_ij M_i[cc] + C_ij l_i[cc]) + h sum_a dfdy_j[a][cc] BM_j[a] .
// The time terms fold into the same sums: with FB = sum_j f_j . BM_j (+ E f^ . l) the columns t_0 / t_f get -/+ FB, and the rank-2
// update of H (LGLDefects.h:508-511) is  M_i -/+= E_i g^_i  in the lanes of rows t_0 / t_f (the row part) and -/+ HT[r] on the two
// time columns (the column part), HT[r] = (g_j[cc] + sum_i g^_i . hE_i d_i) / h from the lane's own d_i.
// For a fixed block column the lanes' rows are contiguous in the block (either layout of defect_dims.h, Dims::KL -- the reference's slot
// order, DenseFunctionBase.h:1112-1123, or J | H): a row group stores 128 contiguous bytes per column.  Lanes above the diagonal carry an
// out-of-range buffer offset.  With the J | H layout (round 6) the rows of [J ; g^T] and the rows of H -- different passes -- never write one
// 32-byte sector twice: WRITE_SIZE 1.29-1.38 x the block bytes in the reference's order, 1.00 x now (profiles/r6_layout_ab.txt).
//
// Counted on the ISA of Reentry-LGL7 (tools/isa_count.py): an H pass (the lower triangles of two segments) is ~ 1 000 vector
// instructions, 473 of them v_fmac_f64_dpp; a C pass (the Jacobian and gradient rows of four segments) ~ 1 000 with 456: ~ 800 per
// segment, no matrix instruction -- the tile form issues 500 per segment AND 33 matrix instructions of 64 cycles each.  Which shapes
// take which form is decided by measurement (ResDims::RD_SHAPE).
#pragma once
#include <utility>

namespace asset_hip {

template <int B, int E, class F>
__device__ __forceinline__ void rd_for_range(F& f) {
  if constexpr (E - B == 1) f(std::integral_constant<int, B>{});
  else if constexpr (E - B > 1) {
    constexpr int M = B + (E - B) / 2;
    rd_for_range<B, M>(f);
    rd_for_range<M, E>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void rd_for(F&& f) { rd_for_range<0, N>(f); }

// acc += (lane KL of op's 16-lane row) * x.  `op` must come straight from a load: a vector-ALU write of a register within two
// instructions of a DPP read of it is a hazard the compiler does not see inside inline assembly (tools/isa_dpp_hazard.py checks
// the built code objects for it).
template <int KL>
__device__ __forceinline__ void fmac_bc(double& acc, double op, double x) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(op), "v"(x), "n"(KL));
}

constexpr int rd_tri_row(int e) { int x = 0; while ((x + 1) * (x + 2) / 2 <= e) x++; return x; }   // row of packed-lower entry e

// 1.0 / 0.0 and v / 0.0 by a lane condition as INTEGER selects on the halves: a select between doubles with a load on one side
// comes out of the compiler as a branch around the load (an exec-mask region, a fence for the scheduler)
__device__ __forceinline__ double rd_sel01(bool c) { return __hiloint2double(c ? 0x3FF00000 : 0, 0); }
__device__ __forceinline__ double rd_keep(bool c, double v) { return __hiloint2double(c ? __double2hiint(v) : 0, c ? __double2loint(v) : 0); }

// A slot entry by a BYTE offset out of a lane record (LaneH / LaneC hold byte offsets in 16 bits: the address is ONE add with a word
// select -- v_add_u32_sdwa -- instead of an unpack, a shift and an add)
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ double rd_at(const lds_double* S, unsigned short byte_off) {
  return *reinterpret_cast<const lds_double*>(reinterpret_cast<const lds_char*>(S) + byte_off);
}

// Operand registers over entries [LO, HI) of an LDS array: register k holds base[16 (R0 + k) + lr] in lane lr of every row
// (base differs from row to row: each row's own segment)
template <int LO, int HI>
struct RdOps {
  static constexpr int R0 = LO / 16, NR = HI > LO ? (HI - 1) / 16 - R0 + 1 : 1;
  double r[NR];
  __device__ __forceinline__ void load(const lds_double* base_plus_lr) {
#pragma unroll
    for (int k = 0; k < NR; k++) r[k] = base_plus_lr[16 * (R0 + k)];
  }
  template <int OFF>
  __device__ __forceinline__ void fm(double& acc, double x) const {
    static_assert(OFF >= LO && OFF < HI, "operand outside the loaded range");
    fmac_bc<OFF % 16>(acc, r[OFF / 16 - R0], x);
  }
};

template <class Ode, class D>
struct RdDims {
  static constexpr int K = D::K, CS = D::CS, n = D::n, m = D::m, p = D::p, q = D::q, N = D::N, T = D::T, TF = D::TF;
  static constexpr int IR = D::IR, OR = D::OR, P0 = D::P0, NZJ = D::NZJ, NZH = D::NZH;
  static constexpr int RG = (IR + 15) / 16;            // row groups of H
  static constexpr int CRG = (OR + 1 + 15) / 16;       // row groups of [J ; g^T]
  static constexpr int NJC = p > 0 ? CS * n : n;
  // what an H row needs that depends on the row and not on the segment (computed once per handle: rd_lane_setup)
  struct LaneH {
    double wE[K];                  // E_i x weight of the row's unit entry of DI_i: A_ij | 1 - s_i / s_i | U_ij | 1 (parameter row)
    double bE[K];                  // E_i B_i,j(r)  (p == 0: the row's column of dfdy_j is scaled by it)
    double tsg;                    // -1 / +1 on the rows t_0 / t_f, 0 elsewhere
    int ar;                        // component of the unit entry: cc(r), or q + pr
    unsigned short jc[NJC];        // slot offsets of the row's column of dfdy: p == 0: dfdy_j(r)[a][cc(r)]; p > 0: per node (zero cell off the row's node; parameter rows: every node)
    unsigned short gs[CS];         // g_j[cc(r)] (zero cell off the row's node; parameter rows: every node)
    unsigned short hr[CS][N];      // row cc(r) of the cardinal Hessian of node jj (zero cell off the row's node; parameter rows: every node)
    unsigned short hc[K][N];       // column a(r) of the interior Hessian H^_i: what the UNIT entry of DI_i[:, r] picks out of it (zero cell: a structural zero)
  };
  static constexpr int NWH = int((sizeof(LaneH) + 3) / 4), NQH = (NWH + 3) / 4;
  static constexpr int NRECH = RG * 16;
  // what a row of J needs as the adjoint gradient of a UNIT multiplier vector (UNITC, below): where row r0 of J^_i0 sits in the slot
  struct LaneC {
    unsigned short jo[N];          // J^_i0[r0][aa] (zero cell: a structural zero, or a lane without a defect row)
  };
  static constexpr int NWC = int((sizeof(LaneC) + 3) / 4), NQC = (NWC + 3) / 4;
  static constexpr int NRECC = CRG * 16;
  static constexpr long long table_bytes() { return (long long)NQH * 16 * NRECH + (long long)NQC * 16 * NRECC; }
};

template <class Ode, class D, int ZERO>
__device__ void rd_lane_setup(unsigned int* out, int rec) {
  using X = RdDims<Ode, D>;
  constexpr int K = X::K, CS = X::CS, n = X::n, q = X::q, N = X::N, T = X::T, TF = X::TF, IR = X::IR, P0 = X::P0, p = X::p;
  union { typename X::LaneH h; unsigned int w[X::NQH * 4]; } u;
  for (int k = 0; k < X::NQH * 4; k++) u.w[k] = 0u;
  const LglTab& tab = d_lgl_tab[D::TAB];
  const int r = rec;                                   // rec = 16 rg + lr = the row
  auto jofs = [](int j, int a, int cc) { const int jp = Ode::JPOS[a * N + cc]; return jp >= 0 ? D::w_CJ + j * D::NZJ + jp : ZERO; };
  auto hofs = [](int j, int a, int b) {
    const int hp = Ode::HPOS[(a >= b) ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a];
    return hp >= 0 ? D::w_CH + j * D::NZH + hp : ZERO;
  };
  typename X::LaneH& L = u.h;
  for (int k = 0; k < X::NJC; k++) L.jc[k] = (unsigned short)ZERO;
  for (int i = 0; i < K; i++)
    for (int c = 0; c < N; c++) L.hc[i][c] = (unsigned short)ZERO;
  for (int jj = 0; jj < CS; jj++) {
    L.gs[jj] = (unsigned short)ZERO;
    for (int c = 0; c < N; c++) L.hr[jj][c] = (unsigned short)ZERO;
  }
  if (r < IR) {
    const bool node = r < P0;
    const int j = node ? r / q : 0, cc = node ? r - j * q : 0, pr = node ? 0 : r - P0;
    L.ar = node ? cc : q + pr;
    L.tsg = (r == T) ? -1.0 : ((r == TF) ? 1.0 : 0.0);
    for (int i = 0; i < K; i++)
      for (int c = 0; c < N; c++) {
        const int hp = Ode::HPOS[(c >= L.ar) ? c * (c + 1) / 2 + L.ar : L.ar * (L.ar + 1) / 2 + c];
        if (hp >= 0) L.hc[i][c] = (unsigned short)(D::w_IH + i * D::NZH + hp);
      }
    for (int i = 0; i < K; i++) {
      double w = 0.0;
      if (!node) w = 1.0;
      else if (cc < n) w = tab.A[i][j];
      else if (cc == T) w = (r == T) ? 1.0 - tab.s[i] : ((r == TF) ? tab.s[i] : 0.0);
      else w = tab.U[i][j];
      L.wE[i] = tab.E[i] * w;
      L.bE[i] = node ? tab.E[i] * tab.B[i][j] : 0.0;
    }
    for (int a = 0; a < n; a++) {
      if constexpr (p == 0) L.jc[a] = (unsigned short)jofs(j, a, cc);
      else
        for (int jj = 0; jj < CS; jj++)
          L.jc[jj * n + a] = (unsigned short)(node ? (jj == j ? jofs(jj, a, cc) : ZERO) : jofs(jj, a, q + pr));
    }
    for (int jj = 0; jj < CS; jj++) {
      const bool mine = node ? jj == j : true;
      const int row = node ? cc : q + pr;
      L.gs[jj] = (unsigned short)(mine ? D::w_Cg + jj * N + row : ZERO);
      for (int c = 0; c < N; c++) L.hr[jj][c] = (unsigned short)(mine ? hofs(jj, row, c) : ZERO);
    }
  }
  // (byte offsets: rd_at)
  for (int k = 0; k < X::NJC; k++) L.jc[k] = (unsigned short)(8 * L.jc[k]);
  for (int jj = 0; jj < CS; jj++) {
    L.gs[jj] = (unsigned short)(8 * L.gs[jj]);
    for (int c = 0; c < N; c++) L.hr[jj][c] = (unsigned short)(8 * L.hr[jj][c]);
  }
  for (int i = 0; i < K; i++)
    for (int c = 0; c < N; c++) L.hc[i][c] = (unsigned short)(8 * L.hc[i][c]);
  for (int k = 0; k < X::NQH * 4; k++) out[(k >> 2) * (X::NRECH * 4) + rec * 4 + (k & 3)] = u.w[k];
}
template <class Ode, class D, int ZERO>
__device__ void rd_lane_setup_c(unsigned int* out, int rec) {      // (the C records follow the H records of the table)
  using X = RdDims<Ode, D>;
  union { typename X::LaneC c; unsigned int w[X::NQC * 4]; } u;
  for (int k = 0; k < X::NQC * 4; k++) u.w[k] = 0u;
  const int jr = rec, i0 = jr < X::OR ? jr / X::n : 0, r0 = jr < X::OR ? jr - i0 * X::n : 0;
  for (int aa = 0; aa < X::N; aa++) {
    const int jp = jr < X::OR ? Ode::JPOS[r0 * X::N + aa] : -1;
    u.c.jo[aa] = (unsigned short)(8 * (jp >= 0 ? D::w_IJ + i0 * D::NZJ + jp : ZERO));   // (byte offset: rd_at)
  }
  unsigned int* const base = out + X::NQH * 4 * X::NRECH;
  for (int k = 0; k < X::NQC * 4; k++) base[(k >> 2) * (X::NRECC * 4) + rec * 4 + (k & 3)] = u.w[k];
}

// (Round 5 built this dense part into a one-launch kernel of heavy right-hand sides as well -- eight-wave workgroups, units and
//  passes in one launch, no workspace: bit-for-bit as good and slower (1 000 Betts-LGL5 segments 31.6 us against 20.1 us of the unit
//  kernels + the tile form's dense part, 5 000 Betts-LGL7 segments 93.5 against 90.0: profiles/r5_ures_timeline.txt).  It is kept, unbuilt,
//  in tools/attic/defect_ures.h.)

// ---------------------------------------------------------------------------------------------------------------- the passes
// Ctx: what the kernel hands over -- the workgroup's segments (slot pointer and mesh segment of group member g), the LDS copy of
// the weight tables, the outputs.
// MODE 0: every pass of the workgroup's group, dealt between its waves.  MODE 1 / 2 (ResDims::EARLYC, pair workgroups): the C passes
// alone -- ALL of them, by the calling wave, while its partner is still in the cardinal second-derivative phase, which the rows of
// [J ; g^T] need nothing of -- and later the H passes alone, dealt with what each wave has done by then counted in (load0 / load1).
// `recw`: the calling wave's row record; MODE 1 loads it ahead of its first store for the H passes that follow, MODE 2 loads it when
// `have_rec` is false.
template <class Ode, class D, int SLOTZERO, int S_FB, int LEVEL, int MODE = 0, class PSlot, class PSeg, class PFbl>
__device__ __forceinline__ void rowdpp_dense(const EvalArgs& a, const lds_double* tabL, const unsigned int* rectab, int gall, int seg_lo,
                                            int seg_hi, int wv, int nwv, int lane, PSlot pslot, PSeg pseg, PFbl pfbl,
                                            unsigned int (&recw)[RdDims<Ode, D>::NQH * 4], bool have_rec = false, int load0 = 0, int load1 = 0,
                                            long long* tsp = nullptr, int* ntsp = nullptr) {
#define RDTS() do { if (tsp && *ntsp < 24) tsp[(*ntsp)++] = clock64(); } while (0)
  using X = RdDims<Ode, D>;
  constexpr int K = X::K, CS = X::CS, n = X::n, q = X::q, N = X::N, T = X::T, TF = X::TF, IR = X::IR, OR = X::OR, P0 = X::P0, p = X::p;
  constexpr int NZJ = X::NZJ, NZH = X::NZH, RG = X::RG, CRG = X::CRG, KSTRIDE = D::KSTRIDE;   // (block layout: defect_dims.h, Dims::KL)
  constexpr int oS = __builtin_offsetof(LglTab, s) / 8, oA = __builtin_offsetof(LglTab, A) / 8, oB = __builtin_offsetof(LglTab, B) / 8;
  constexpr int oU = __builtin_offsetof(LglTab, U) / 8, oC = __builtin_offsetof(LglTab, C) / 8, oD = __builtin_offsetof(LglTab, D) / 8;
  constexpr int oE = __builtin_offsetof(LglTab, E) / 8;
  // UNITC (round 6; level 2): every row of J is the adjoint gradient for a UNIT multiplier vector e_(i0, r0) -- M_i = 0 off i0 and
  // M_i0 = hE_i0 (row r0 of J^_i0), BM_j = B_i0j M_i0 + D_i0j e_r0 -- so the C pass carries ONE interior's row through lane-own weights
  // instead of K through the general path (a third of its instructions), and the adjoint gradient itself -- the one row with a full
  // multiplier vector -- comes out of the H passes, whose lane of row r already holds most of entry r:
  //     agx[r] = sum_i hE_i w_i(r) g^_i[a(r)] + sum_j g_j[cc(r)] + sum_i C_ij(r) lam_i[cc(r)] -/+ FB on the rows t_0 / t_f
  // (g_j = J_j^T w_j carries the B and D parts: LGLDefects.h:369-374, :512), FB from the terms the ODE phases left (ResDims::x_FBL).
  constexpr bool UNITC = ResDims<D>::UNITC && LEVEL >= 2;
  static_assert(ResDims<D>::SLOT * 8 < 65536, "the lane records hold byte offsets into a slot in 16 bits (rd_at)");
  typedef __attribute__((ext_vector_type(2))) unsigned int u2;
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  const LglTab& ctab = d_lgl_tab[D::TAB];                           // compile-time indices: scalar loads
  const int lr = lane & 15, rs = lane >> 4;
  constexpr unsigned INVALID = 0xF0000000u;
#define ASSET_RD_NB 8
#define ASSET_RD_CCOST_UNIT 300      // a UNITC C pass in the units of the dealing below (instructions of a pass / 1.6; 540 in the general form; 200 / 400 measure the same: profiles/r6_unitc.txt)
  constexpr int NB = ASSET_RD_NB;                                    // block columns evaluated side by side

  // outputs of the workgroup's segments [seg_lo, seg_hi): one buffer resource each, offsets relative to seg_lo (an output the caller
  // did not ask for: zero records, every store dropped)
  const int nsegs = seg_hi > seg_lo ? seg_hi - seg_lo : 0;
  const size_t s0 = size_t(seg_lo);
  const __amdgpu_buffer_rsrc_t rs_kkt = __builtin_amdgcn_make_buffer_rsrc(a.KKT + s0 * size_t(KSTRIDE), 0, a.KKT ? nsegs * KSTRIDE * 8 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_fx = __builtin_amdgcn_make_buffer_rsrc(a.FX + s0 * size_t(OR), 0, a.FX ? nsegs * OR * 8 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_agx = __builtin_amdgcn_make_buffer_rsrc(a.AGX + s0 * size_t(IR), 0, (a.AGX && a.L) ? nsegs * IR * 8 : 0, 0x00020000);
  auto bst = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, voff, soff, 0);
  };

  RdOps<0, D::TABSZ> tb;                                             // the weight tables: the same in every row
  tb.load(tabL + lr);


  // ================================================================================================ H pass
  // Row slot rs of a wave (lanes 16 rs .. 16 rs + 15) always works on row group rs % RG of H: a lane's record never changes, and it
  // is loaded ONCE, ahead of the wave's first store -- on gfx9 loads and stores complete through one in-order counter, so a load
  // behind a pass's block stores waits for every one of them (measured with the record reloaded per pass: 9.8 k cycles for the
  // reload of a wave's second pass against 1.5 k for the first).  An H pass is then the whole lower triangle of SPP = 4 / RG
  // segments, a C pass the Jacobian and gradient rows of SPC = 4 / CRG segments; a segment's passes follow each other closely, so
  // that the partially written lines of its block columns meet in the L2 instead of going to memory half filled.
  constexpr int SPP = 4 / RG > 0 ? 4 / RG : 1, SPC = 4 / CRG > 0 ? 4 / CRG : 1;
  static_assert(RG <= 4 && CRG <= 4, "a segment's row groups fit one pass");
  const int hsub = rs / RG, rg = rs - hsub * RG;
  if (LEVEL >= 2 && !(MODE != 1 && have_rec)) {
    const u4* src = reinterpret_cast<const u4*>(rectab) + (16 * rg + lr);
#pragma unroll
    for (int k = 0; k < X::NQH; k++) {
      const u4 v = src[k * X::NRECH];
      recw[4 * k] = v.x, recw[4 * k + 1] = v.y, recw[4 * k + 2] = v.z, recw[4 * k + 3] = v.w;
    }
  }
  const typename X::LaneH& recL = *reinterpret_cast<const typename X::LaneH*>(recw);
  unsigned int reccw[X::NQC * 4];                                    // (UNITC) the lane's C record: its row of [J ; g^T] never changes either
  if constexpr (UNITC && MODE != 2) {
    const u4* src = reinterpret_cast<const u4*>(rectab) + X::NQH * X::NRECH + (16 * ((lane >> 4) % CRG) + lr);
#pragma unroll
    for (int k = 0; k < X::NQC; k++) {
      const u4 v = src[k * X::NRECC];
      reccw[4 * k] = v.x, reccw[4 * k + 1] = v.y, reccw[4 * k + 2] = v.z, reccw[4 * k + 3] = v.w;
    }
  }
  const typename X::LaneC& recC = *reinterpret_cast<const typename X::LaneC*>(reccw);
  auto hpass = [&](const int pl) __attribute__((always_inline)) {
    const int g0 = SPP * pl + hsub;
    const bool tv = hsub < SPP && g0 < gall;
    const int g = tv ? g0 : 0;
    const int r = 16 * rg + lr;
    const bool rv = tv && r < IR;
    const lds_double* const S = pslot(g);
    const lds_double* const Sl = S + lr;
    const unsigned kb = rv ? unsigned((pseg(g) - seg_lo) * (KSTRIDE * 8) + 8 * r) : INVALID;   // slot (r, c) at kb + 8 hcol(c)
    if constexpr (LEVEL < 2) {
      // the Jacobian kinds (evalSOE / evalAUG): the Hessian slots hold zeros -- unless the caller never reads them
      // (ASSET_HIP_KEEP_HESSIAN_SLOTS: the pass is not run at all, below)
      rd_for<IR>([&](auto CC) {
        constexpr int c = decltype(CC)::value;
        bst(rs_kkt, (c <= r) ? kb : INVALID, 8 * D::hcol(c), 0.0);
      });
      return;
    }
    const typename X::LaneH& L = recL;
    const double h = S[S_FB + K * n];                  // (t_f - t_0, left there by the interior phase)
    const double rh = 1.0 / h;

    // (every chain of dependent FMAs below is interleaved with others by hand: the compiler does not know the latency of an inline
    //  assembly statement and leaves them in source order -- a dependent v_fmac_f64 issues every 9 cycles, an independent one every 5)
    // ---- d'_i = hE_i DI_i[:, r] WITHOUT its unit entry hw_i e_a(r): what that entry contributes to M_i is hw_i times column a(r) of H^_i,
    //      N values a lane reads by its own offsets (LaneH::hc) -- LDS reads, which do not take the f64 pipe -- instead of N selects per
    //      interior to place it and an FMA for every structural entry of H^_i it meets.  The rest of d'_i lives on the state rows alone.
    double d[K][n], hw[K];
    {
      RdOps<S_FB, S_FB + K * n> fbv;                                 // FB_i[a] = sum_j B_ij f_j[a]  (written by the interior phase)
      fbv.load(Sl);
      double jcv[X::NJC];
#pragma unroll
      for (int k = 0; k < X::NJC; k++) jcv[k] = rd_at(S, L.jc[k]);
      double tsh[K], hhb[K];
#pragma unroll
      for (int i = 0; i < K; i++) hw[i] = h * L.wE[i], tsh[i] = L.tsg * (h * ctab.E[i]), hhb[i] = h * h * (p == 0 ? L.bE[i] : ctab.E[i]);
      double sj[p > 0 ? K : 1][p > 0 ? n : 1];
      if constexpr (p > 0) {
        rd_for<K>([&](auto I) { constexpr int i = decltype(I)::value; rd_for<n>([&](auto A) { sj[i][decltype(A)::value] = 0.0; }); });
        rd_for<CS>([&](auto JJ) {
          constexpr int jj = decltype(JJ)::value;
          rd_for<K>([&](auto I) {
            constexpr int i = decltype(I)::value;
            rd_for<n>([&](auto A) { constexpr int aa = decltype(A)::value; tb.template fm<oB + 4 * i + jj>(sj[i][aa], jcv[jj * n + aa]); });
          });
        });
      }
      rd_for<K>([&](auto I) {
        constexpr int i = decltype(I)::value;
        rd_for<n>([&](auto A) {
          constexpr int aa = decltype(A)::value;
          d[i][aa] = hhb[i] * (p == 0 ? jcv[aa] : sj[p > 0 ? i : 0][p > 0 ? aa : 0]);
        });
      });
      rd_for<n>([&](auto A) {
        constexpr int aa = decltype(A)::value;
        rd_for<K>([&](auto I) { constexpr int i = decltype(I)::value; fbv.template fm<S_FB + i * n + aa>(d[i][aa], tsh[i]); });
      });
    }
    RDTS();
    // ---- M_i = H^_i d'_i  (-/+ E_i g^_i in the rows t_0 / t_f: the row part of the rank-2 update), HT[r]
    double M[K][N], hto;
    {
      RdOps<D::w_Ig, D::w_Ig + K * N> ig;
      ig.load(Sl);
      RdOps<D::w_IH, D::w_IH + K * NZH> ih;
      ih.load(Sl);
      double gsum = 0.0;
#pragma unroll
      for (int jj = 0; jj < CS; jj++) gsum += rd_at(S, L.gs[jj]);
      double hsp[K], tse[K];
#pragma unroll
      for (int i = 0; i < K; i++) hsp[i] = 0.0, tse[i] = L.tsg * ctab.E[i];
      double gu[K];                                                  // hw_i g^_i[a(r)]: the unit entry's share of g^_i . d'_i -- and of the adjoint gradient's entry r
#pragma unroll
      for (int i = 0; i < K; i++) gu[i] = hw[i] * S[D::w_Ig + i * N + L.ar];
      rd_for<N>([&](auto A) {
        constexpr int aa = decltype(A)::value;
        rd_for<K>([&](auto I) {
          constexpr int i = decltype(I)::value;
          M[i][aa] = hw[i] * rd_at(S, L.hc[i][aa]);
          ig.template fm<D::w_Ig + i * N + aa>(M[i][aa], tse[i]);
          if constexpr (aa < n) ig.template fm<D::w_Ig + i * N + aa>(hsp[i], d[i][aa]);
        });
      });
      rd_for<N*(N + 1) / 2>([&](auto E) {
        constexpr int e = decltype(E)::value;
        constexpr int hp = Ode::HPOS[e];
        if constexpr (hp >= 0) {
          constexpr int ra = rd_tri_row(e);
          constexpr int cb = e - ra * (ra + 1) / 2;
          if constexpr (cb < n)
            rd_for<K>([&](auto I) {
              constexpr int i = decltype(I)::value;
              ih.template fm<D::w_IH + i * NZH + hp>(M[i][ra], d[i][cb]);
            });
          if constexpr (ra != cb && ra < n)
            rd_for<K>([&](auto I) {
              constexpr int i = decltype(I)::value;
              ih.template fm<D::w_IH + i * NZH + hp>(M[i][cb], d[i][ra]);
            });
        }
      });
#pragma unroll
      for (int i = 0; i < K; i++) hsp[i] += gu[i];
      if constexpr (UNITC) {
        const int jn = r < P0 ? r / q : CS - 1;                      // (rows without a state entry read lam's zero cell: any weight will do)
        const int lo = L.ar < n ? D::w_lam + L.ar : SLOTZERO, lst = L.ar < n ? n : 0;
        double ag = gsum;
#pragma unroll
        for (int i = 0; i < K; i++) ag += gu[i];
#pragma unroll
        for (int i = 0; i < K; i++) ag = fma(tabL[oC + 4 * i + jn], S[lo + i * lst], ag);
        const lds_double* const F = pfbl(g);
        double fc = 0.0, fi2 = 0.0;
#pragma unroll
        for (int j = 0; j < CS; j++) fc += F[j];
#pragma unroll
        for (int i = 0; i < K; i++) fi2 += F[CS + i];
        ag = fma(L.tsg, fma(rh, fc, fi2), ag);
        bst(rs_agx, rv ? unsigned((pseg(g) - seg_lo) * (IR * 8) + 8 * r) : INVALID, 0, ag);
      }
#pragma unroll
      for (int i = 0; i < K; i++) gsum += hsp[i];
      hto = rh * gsum;
    }
    RDTS();
    // ---- BM_j = sum_i B_ij M_i[0:n], FB = sum_j f_j . BM_j, the time sums
    double BM[CS][n], fb, tm1 = 0.0, tms = 0.0;
    {
      RdOps<D::w_Cf, D::w_Cf + CS * n> cf;
      cf.load(Sl);
      double fbp[CS];
      rd_for<CS>([&](auto J) {
        constexpr int j = decltype(J)::value;
        rd_for<n>([&](auto A) { BM[j][decltype(A)::value] = 0.0; });
        rd_for<K>([&](auto I) {
          constexpr int i = decltype(I)::value;
          rd_for<n>([&](auto A) { constexpr int aa = decltype(A)::value; tb.template fm<oB + 4 * i + j>(BM[j][aa], M[i][aa]); });
        });
        fbp[j] = 0.0;
      });
      rd_for<n>([&](auto A) {
        constexpr int aa = decltype(A)::value;
        rd_for<CS>([&](auto J) { constexpr int j = decltype(J)::value; cf.template fm<D::w_Cf + j * n + aa>(fbp[j], BM[j][aa]); });
      });
      fb = 0.0;
#pragma unroll
      for (int j = 0; j < CS; j++) fb += fbp[j];
#pragma unroll
      for (int j = 0; j < CS; j++)
#pragma unroll
        for (int aa = 0; aa < n; aa++) BM[j][aa] *= h;
      rd_for<K>([&](auto I) {
        constexpr int i = decltype(I)::value;
        tb.template fm<oS + i>(tm1, M[i][T]);
        tms += M[i][T];
      });
    }
    RDTS();
    const double w_t0 = (tms - tm1) - fb - hto, w_tf = tm1 + fb + hto;
    const double tg = L.tsg * rh;
    RdOps<D::w_CJ, D::w_CJ + CS * NZJ> cj;
    cj.load(Sl);
    RdOps<D::w_Cg, D::w_Cg + CS * N> cg;
    cg.load(Sl);
    // ---- the columns, sixteen at a time (a range only while some row group of the pass reaches it), NB columns side by side:
    //      term s of every column of a batch before term s + 1 of any
    auto col_init = [&](auto CC) -> double {
      constexpr int c = decltype(CC)::value;
      if constexpr (c < P0) {
        constexpr int j = c / q, cc = c - j * q;
        if constexpr (cc == T) return (j == 0) ? w_t0 : ((j == CS - 1) ? w_tf : 0.0);
        else return rd_at(S, L.hr[j][cc]);                    // (the cardinal Hessian entry: the sum starts from it -- a read, not a read and an add)
      } else {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < K; i++) acc += M[i][q + (c - P0)];
        return acc;
      }
    };
    auto col_term = [&](auto CC, auto SS, double& acc) {
      constexpr int c = decltype(CC)::value, s = decltype(SS)::value;
      if constexpr (c < P0) {
        constexpr int j = c / q, cc = c - j * q;
        if constexpr (s < K) {
          if constexpr (cc != T) tb.template fm<(cc < n ? oA : oU) + 4 * s + j>(acc, M[s][cc]);
        } else if constexpr (s < K + n) {
          constexpr int aa = s - K, jp = Ode::JPOS[aa * N + cc];
          if constexpr (jp >= 0) cj.template fm<D::w_CJ + j * NZJ + jp>(acc, BM[j][aa]);
        } else if constexpr (s == K + n) { if constexpr (cc == T) acc += rd_at(S, L.hr[j][cc]); }
        else if constexpr (s == K + n + 1) cg.template fm<D::w_Cg + j * N + cc>(acc, tg);
      } else {
        constexpr int pc = c - P0;
        if constexpr (s < CS * n) {
          constexpr int j = s / n, aa = s - j * n, jp = Ode::JPOS[aa * N + q + pc];
          if constexpr (jp >= 0) cj.template fm<D::w_CJ + j * NZJ + jp>(acc, BM[j][aa]);
        } else if constexpr (s < CS * n + CS) acc += rd_at(S, L.hr[s - CS * n][q + pc]);
        else if constexpr (s < CS * n + 2 * CS) cg.template fm<D::w_Cg + (s - CS * n - CS) * N + q + pc>(acc, tg);
      }
    };
    constexpr int NT = (p > 0 && CS * n + 2 * CS > K + n + 2) ? CS * n + 2 * CS : K + n + 2;
    rd_for<RG>([&](auto RNG) {
      constexpr int c0 = 16 * decltype(RNG)::value, c1 = c0 + 16 < IR ? c0 + 16 : IR;
      {
        rd_for<(c1 - c0 + NB - 1) / NB>([&](auto BB) {
          constexpr int b0 = c0 + NB * decltype(BB)::value, nb = b0 + NB <= c1 ? NB : c1 - b0;
          double acc[NB];
          rd_for<nb>([&](auto Bx) { acc[decltype(Bx)::value] = col_init(std::integral_constant<int, b0 + decltype(Bx)::value>{}); });
          rd_for<NT>([&](auto SS) {
            rd_for<nb>([&](auto Bx) { col_term(std::integral_constant<int, b0 + decltype(Bx)::value>{}, SS, acc[decltype(Bx)::value]); });
          });
          rd_for<nb>([&](auto Bx) {
            constexpr int c = b0 + decltype(Bx)::value;
            bst(rs_kkt, (c <= r) ? kb : INVALID, 8 * D::hcol(c), acc[decltype(Bx)::value]);
          });
        });
      }
    });
  };

  // ================================================================================================ C pass
  auto cpass = [&](const int pl) __attribute__((always_inline)) {
    const int csub = rs / CRG, crg = rs - csub * CRG;
    const int g0 = SPC * pl + csub;
    const bool tv = csub < SPC && g0 < gall;
    const int g = tv ? g0 : SPC * pl;                  // (an idle row slot reads along with the pass's first segment)
    const int jr = 16 * crg + lr;
    const bool isJ = tv && jr < OR, isG = tv && jr == OR;
    const int jrr = isJ ? jr : 0, i0 = jrr / n, r0 = jrr - i0 * n;
    const lds_double* const S = pslot(g);
    const lds_double* const Sl = S + lr;
    const int srel = pseg(g) - seg_lo;
    const unsigned kb = isJ ? unsigned(srel * (KSTRIDE * 8) + 8 * jr) : INVALID;            // slot (jr, c) at kb + 8 jcol(c)
    const double h = S[S_FB + K * n];                  // (t_f - t_0, left there by the interior phase)
    const double rh = 1.0 / h;
    // the defect value of the row: sum_j C_ij z_j[r] + h (sum_j D_ij f_j[r] + E_i f^_i[r])    (LGLDefects.h:460-500)
    double sdv;
    {
      double cz = 0.0, sd = tabL[oE + i0] * S[D::w_If + jrr];
#pragma unroll
      for (int j = 0; j < CS; j++) {
        cz = fma(tabL[oC + 4 * i0 + j], S[D::w_z + j * q + r0], cz);
        sd = fma(tabL[oD + 4 * i0 + j], S[D::w_Cf + j * n + r0], sd);
      }
      bst(rs_fx, isJ ? unsigned(srel * (OR * 8) + 8 * jr) : INVALID, 0, fma(h, sd, cz));
      sdv = sd;
    }
    if constexpr (UNITC) {
      // ---- the rows of J alone, each for its unit multiplier vector e_(i0, r0) (the gradient row: hpass)
      const lds_double* const tw = tabL + 4 * i0;                    // the lane's rows of the weight tables
      double wA[CS], wB[CS], wC[CS], wD[CS], wU[CS];
#pragma unroll
      for (int j = 0; j < CS; j++) wA[j] = tw[oA + j], wB[j] = tw[oB + j], wC[j] = tw[oC + j], wD[j] = tw[oD + j], wU[j] = tw[oU + j];
      const double hE = h * tabL[oE + i0];
      double M[N];
#pragma unroll
      for (int aa = 0; aa < N; aa++) M[aa] = hE * rd_at(S, recC.jo[aa]);
      // FB = E f^ . l + sum_j f_j . BM_j = (sum_j D_i0j f_j[r0] + E_i0 f^_i0[r0]) + sum_a FB_i0[a] M[a]
      double fb = sdv;
#pragma unroll
      for (int aa = 0; aa < n; aa++) fb = fma(S[S_FB + i0 * n + aa], M[aa], fb);
      double dsel[n], hB[CS], hD[CS];
#pragma unroll
      for (int aa = 0; aa < n; aa++) dsel[aa] = rd_sel01(isJ && r0 == aa);
#pragma unroll
      for (int j = 0; j < CS; j++) hB[j] = h * wB[j], hD[j] = h * wD[j];
      double BM[CS][n];
#pragma unroll
      for (int j = 0; j < CS; j++)
#pragma unroll
        for (int aa = 0; aa < n; aa++) BM[j][aa] = fma(hB[j], M[aa], hD[j] * dsel[aa]);
      const double tm1 = tabL[oS + i0] * M[T];
      const double w_t0 = (M[T] - tm1) - fb, w_tf = tm1 + fb;
      RdOps<D::w_CJ, D::w_CJ + CS * NZJ> cj;
      cj.load(Sl);
      auto col_init = [&](auto CC) -> double {
        constexpr int c = decltype(CC)::value;
        if constexpr (c < P0) {
          constexpr int j = c / q, cc = c - j * q;
          if constexpr (cc == T) return (j == 0) ? w_t0 : ((j == CS - 1) ? w_tf : 0.0);
          else if constexpr (cc < n) return wC[j] * dsel[cc];
          else return 0.0;
        } else return M[q + (c - P0)];
      };
      auto col_term = [&](auto CC, auto SS, double& acc) {
        constexpr int c = decltype(CC)::value, s = decltype(SS)::value;
        if constexpr (c < P0) {
          constexpr int j = c / q, cc = c - j * q;
          if constexpr (s == 0) {
            if constexpr (cc != T) acc = fma(cc < n ? wA[j] : wU[j], M[cc], acc);
          } else if constexpr (s <= n) {
            constexpr int aa = s - 1, jp = Ode::JPOS[aa * N + cc];
            if constexpr (jp >= 0) cj.template fm<D::w_CJ + j * NZJ + jp>(acc, BM[j][aa]);
          }
        } else {
          constexpr int pc = c - P0;
          if constexpr (s < CS * n) {
            constexpr int j = s / n, aa = s - j * n, jp = Ode::JPOS[aa * N + q + pc];
            if constexpr (jp >= 0) cj.template fm<D::w_CJ + j * NZJ + jp>(acc, BM[j][aa]);
          }
        }
      };
      constexpr int NT = (p > 0 && CS * n > n + 1) ? CS * n : n + 1;
      rd_for<(IR + NB - 1) / NB>([&](auto BB) {
        constexpr int b0 = NB * decltype(BB)::value, nb = b0 + NB <= IR ? NB : IR - b0;
        double acc[NB];
        rd_for<nb>([&](auto Bx) { acc[decltype(Bx)::value] = col_init(std::integral_constant<int, b0 + decltype(Bx)::value>{}); });
        rd_for<NT>([&](auto SS) {
          rd_for<nb>([&](auto Bx) { col_term(std::integral_constant<int, b0 + decltype(Bx)::value>{}, SS, acc[decltype(Bx)::value]); });
        });
        rd_for<nb>([&](auto Bx) {
          constexpr int c = b0 + decltype(Bx)::value;
          bst(rs_kkt, kb, 8 * D::jcol(c), acc[decltype(Bx)::value]);
        });
      });
      return;
    }
    // the row's multiplier vector l (unit vector of the defect row, lam for the gradient row), and hE_i l_i
    double lw[K][n], ls[K][n];
    const double gsel = rd_sel01(isG);
    const int jsel = isJ ? jr : -1;
#pragma unroll
    for (int i = 0; i < K; i++)
#pragma unroll
      for (int rr = 0; rr < n; rr++) {
        const double lam = S[D::w_lam + i * n + rr];
        lw[i][rr] = fma(gsel, lam, rd_sel01(jsel == i * n + rr));
        ls[i][rr] = (h * ctab.E[i]) * lw[i][rr];
      }
    // M_i = hE_i J^_i^T l_i
    double M[K][N];
    double fbe;
    {
      RdOps<D::w_IJ, D::w_IJ + K * NZJ> ij;
      ij.load(Sl);
      RdOps<D::w_If, D::w_If + K * n> fi;
      fi.load(Sl);
      rd_for<K>([&](auto I) { constexpr int i = decltype(I)::value; rd_for<N>([&](auto A) { M[i][decltype(A)::value] = 0.0; }); });
      rd_for<n>([&](auto RR) {                        // (row rr of J^_i: a term for each of its columns -- independent accumulators side by side)
        constexpr int rr = decltype(RR)::value;
        rd_for<N>([&](auto A) {
          constexpr int aa = decltype(A)::value;
          constexpr int jp = Ode::JPOS[rr * N + aa];
          if constexpr (jp >= 0)
            rd_for<K>([&](auto I) { constexpr int i = decltype(I)::value; ij.template fm<D::w_IJ + i * NZJ + jp>(M[i][aa], ls[i][rr]); });
        });
      });
      double fbp[K];
#pragma unroll
      for (int i = 0; i < K; i++) fbp[i] = 0.0;
      rd_for<n>([&](auto RR) {
        constexpr int rr = decltype(RR)::value;
        rd_for<K>([&](auto I) { constexpr int i = decltype(I)::value; fi.template fm<D::w_If + i * n + rr>(fbp[i], ls[i][rr]); });
      });
      fbe = 0.0;
#pragma unroll
      for (int i = 0; i < K; i++) fbe += fbp[i];
    }
    double BM[CS][n], fb = rh * fbe, tm1 = 0.0, tms = 0.0;
    {
      RdOps<D::w_Cf, D::w_Cf + CS * n> cf;
      cf.load(Sl);
      double fbp[CS];
      rd_for<CS>([&](auto J) {
        constexpr int j = decltype(J)::value;
        rd_for<n>([&](auto A) { BM[j][decltype(A)::value] = 0.0; });
        rd_for<K>([&](auto I) {
          constexpr int i = decltype(I)::value;
          rd_for<n>([&](auto A) { constexpr int aa = decltype(A)::value; tb.template fm<oB + 4 * i + j>(BM[j][aa], M[i][aa]); });
          rd_for<n>([&](auto A) { constexpr int aa = decltype(A)::value; tb.template fm<oD + 4 * i + j>(BM[j][aa], lw[i][aa]); });
        });
        fbp[j] = 0.0;
      });
      rd_for<n>([&](auto A) {
        constexpr int aa = decltype(A)::value;
        rd_for<CS>([&](auto J) { constexpr int j = decltype(J)::value; cf.template fm<D::w_Cf + j * n + aa>(fbp[j], BM[j][aa]); });
      });
#pragma unroll
      for (int j = 0; j < CS; j++) fb += fbp[j];
#pragma unroll
      for (int j = 0; j < CS; j++)
#pragma unroll
        for (int aa = 0; aa < n; aa++) BM[j][aa] *= h;
      rd_for<K>([&](auto I) {
        constexpr int i = decltype(I)::value;
        tb.template fm<oS + i>(tm1, M[i][T]);
        tms += M[i][T];
      });
    }
    const double w_t0 = (tms - tm1) - fb, w_tf = tm1 + fb;
    RdOps<D::w_CJ, D::w_CJ + CS * NZJ> cj;
    cj.load(Sl);
    const unsigned gbd = isG ? unsigned(srel * (IR * 8)) : INVALID;
    lds_double* const agx_z = const_cast<lds_double*>(S) + D::w_z;
    lds_double* const agx_dummy = const_cast<lds_double*>(S) + D::w_lam + (lr < OR ? lr : 0);
    auto col_init = [&](auto CC) -> double {
      constexpr int c = decltype(CC)::value;
      if constexpr (c < P0) {
        constexpr int j = c / q, cc = c - j * q;
        if constexpr (cc == T) return (j == 0) ? w_t0 : ((j == CS - 1) ? w_tf : 0.0);
        else return 0.0;
      } else {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < K; i++) acc += M[i][q + (c - P0)];
        return acc;
      }
    };
    auto col_term = [&](auto CC, auto SS, double& acc) {
      constexpr int c = decltype(CC)::value, s = decltype(SS)::value;
      if constexpr (c < P0) {
        constexpr int j = c / q, cc = c - j * q;
        if constexpr (s < K) {
          if constexpr (cc != T) tb.template fm<(cc < n ? oA : oU) + 4 * s + j>(acc, M[s][cc]);
        } else if constexpr (s < K + n) {
          constexpr int aa = s - K, jp = Ode::JPOS[aa * N + cc];
          if constexpr (jp >= 0) cj.template fm<D::w_CJ + j * NZJ + jp>(acc, BM[j][aa]);
        } else if constexpr (s < K + n + K) {
          if constexpr (cc < n) tb.template fm<oC + 4 * (s - K - n) + j>(acc, lw[s - K - n][cc]);
        }
      } else {
        constexpr int pc = c - P0;
        if constexpr (s < CS * n) {
          constexpr int j = s / n, aa = s - j * n, jp = Ode::JPOS[aa * N + q + pc];
          if constexpr (jp >= 0) cj.template fm<D::w_CJ + j * NZJ + jp>(acc, BM[j][aa]);
        }
      }
    };
    constexpr int NT = (p > 0 && CS * n > 2 * K + n) ? CS * n : 2 * K + n;
    rd_for<(IR + NB - 1) / NB>([&](auto BB) {
      constexpr int b0 = NB * decltype(BB)::value, nb = b0 + NB <= IR ? NB : IR - b0;
      double acc[NB];
      rd_for<nb>([&](auto Bx) { acc[decltype(Bx)::value] = col_init(std::integral_constant<int, b0 + decltype(Bx)::value>{}); });
      rd_for<NT>([&](auto SS) {
        rd_for<nb>([&](auto Bx) { col_term(std::integral_constant<int, b0 + decltype(Bx)::value>{}, SS, acc[decltype(Bx)::value]); });
      });
      rd_for<nb>([&](auto Bx) {
        constexpr int c = b0 + decltype(Bx)::value;
        bst(rs_kkt, kb, 8 * D::jcol(c), acc[decltype(Bx)::value]);
        // the gradient row's entry: into the segment's z section (dead: this pass has read it), the other lanes into lam cells of
        // their own (dead as well) -- one coalesced store per sixteen entries below instead of a store of four lanes per column
        if constexpr (MODE == 1) bst(rs_agx, gbd, 8 * c, acc[decltype(Bx)::value]);   // (the partner's phase still reads z and lam: no staging there)
        else *(isG ? agx_z + c : agx_dummy) = acc[decltype(Bx)::value];
      });
    });
    wave_lds_order();
    if constexpr (MODE != 1) {
      // (the row slot that holds the segment's gradient row -- crg of row OR -- stores it; with CRG > 1 the other slots of the segment sit out)
      const bool gslot = tv && (OR / 16) == crg;
      const unsigned gb2 = gslot ? unsigned(srel * (IR * 8) + 8 * lr) : INVALID;
#pragma unroll
      for (int k = 0; k < (IR + 15) / 16; k++)
        bst(rs_agx, (16 * k + lr < IR) ? gb2 : INVALID, 128 * k, Sl[D::w_z + 16 * k]);
    }
  };

  const int nHP = (LEVEL < 2 && (a.flags & 1)) ? 0 : (gall + SPP - 1) / SPP, nCP = (gall + SPC - 1) / SPC;
  if constexpr (MODE == 1) {
    for (int cp = 0; cp < nCP; cp++) {
      RDTS();
      cpass(cp);
    }
  } else if constexpr (MODE == 2) {
    for (int hp = 0; hp < nHP; hp++) {
      const int w = (nwv > 1 && load1 < load0) ? 1 : 0;
      (w ? load1 : load0) += 640;
      if (w != wv) continue;
      RDTS();
      hpass(hp);
    }
  } else {
    // the workgroup's passes in segment order -- the H passes of four segments, then their C pass(es) -- each to the wave with less to do
    // so far (cost: the instructions of a pass, tools/isa_count.py); both waves walk the same list
    int hp = 0, cp = 0;
    while (hp < nHP || cp < nCP) {
      const bool isH = hp < nHP && (cp >= nCP || SPP * hp < SPC * (cp + 1));     // H passes up to the segments of the next C pass first
      const int w = (nwv > 1 && load1 < load0) ? 1 : 0;
      (w ? load1 : load0) += isH ? (LEVEL >= 2 ? 640 : 40) : (UNITC ? ASSET_RD_CCOST_UNIT : 540);
      if (w == wv) {
        RDTS();
        if (isH) hpass(hp); else cpass(cp);
      }
      if (isH) hp++; else cp++;
    }
  }
  RDTS();
#undef RDTS
}

}  // namespace asset_hip
