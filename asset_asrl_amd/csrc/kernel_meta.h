// What the host has to know about one compiled (ODE, transcription, control mode) -- or one plain function -- to create
// handles for it and to launch its kernels: a flat table of integers, all of them compile-time constants of Dims<...>.
// Translation units compiled into libasset_hip.so read the table directly (registry.h); a module compiled at run time
// (hiprtc, rtc_device.h) carries it as the device constant `asset_rtc_meta`, which the loader copies back
// (capi.hip: asset_hip_jit_plugin).  One launcher serves both.
#pragma once
#include "defect_adjgrad.h"
#include "defect_kernels.h"
#include "defect_resident.h"
#include "defect_rows.h"
#include "defect_units.h"
#include "defect_wide.h"
#include "func_kernels.h"

namespace asset_hip {

enum MetaField {
  MF_KIND = 0,       // 1: transcription of an ODE, 2: plain function
  MF_XV, MF_UV, MF_PV,
  MF_MODE,           // ASSET_HIP_* transcription id (0 for plain functions)
  MF_BLOCKED,
  MF_IR, MF_OR, MF_NKKT,
  MF_G,              // segments per group of the ODE stage
  MF_LDS_BYTES,
  MF_WORK_DOUBLES,   // workspace doubles per segment (ODE result slot)
  MF_NACONST,        // plain functions: constants per application (vf.ApplConst)
  MF_BYTES_ODE, MF_BYTES_DENSE,
  MF_WIDE, MF_WIDE_WGS,
  MF_FUSED, MF_GF, MF_FUSED2, MF_GF2, MF_BYTES_FUSED2,
  MF_NUNITS, MF_UNITS_BASE_BYTES, MF_UNITS_SLOT_BYTES, MF_CS,
  MF_LANE_BYTES1, MF_LANE_BYTES2,
  MF_ADJ_GP, MF_ADJ_LDS_BYTES,   // value + adjoint gradient kernel (defect_adjgrad.h): segments per workgroup, its LDS
  MF_RES_GR, MF_RES_LDS_BYTES, MF_LANE_BYTES_RES,   // resident kernel (defect_resident.h): segments per wave (0: none), LDS, record table
  MF_RES_WPS,                                       // ... and the waves per SIMD it is built for
  MF_RESD_GR,                                       // ... its dense part alone behind the unit kernels of a heavy ODE (0: none)
  MF_ROWS_LDS_BYTES,                                // wide shapes, dense stage by output rows (defect_rows.h): its LDS (0: none)
  MF_RES_NWV,                                       // resident kernel: waves per workgroup (2: the pair form, ResDims::PAIR)
  MF_RES_LOOP_NWV,                                  // ... of its looped level-2 block kernel (2 with the row-wise dense part)
  MF_RES_ASM,                                       // ... 1: its assembled kinds exist (0: shapes of two row tiles with the row-wise dense part)
  MF_RES_ALT,                                       // resident kernel: 1 when the row-wise form exists beside a default tile form (ResDims::RD_ALT)
  MF_KL, MF_KSTRIDE,                                // layout of the KKT blocks the kernels write (defect_dims.h: Dims::KL) and their stride in doubles (0: NKKT)
  MF_COUNT
};

template <class Ode, int SCH, bool BLOCKED>
constexpr long long lgl_lane_table_bytes(int level) {
  using D = Dims<Ode, SCH, BLOCKED>;
  // bytes of the whole table: 64 word-interleaved records (defect_kernels.h: lane_setup_kernel; defect_wide.h: wide_setup_kernel)
  if constexpr (D::WIDE) return level >= 1 ? (long long)(D::TJ) * D::TI * 4 * 64 * sizeof(unsigned int) : 0;
  else return level >= 2 ? (long long)sizeof(LaneConsts<Ode, D, 2>) * 64 : (level == 1 ? (long long)sizeof(LaneConsts<Ode, D, 1>) * 64 : 0);
}

template <class Ode, int SCH, bool BLOCKED>
constexpr long long res_lane_table_bytes() {
  using D = Dims<Ode, SCH, BLOCKED>;
  return res_table_words<Ode, D>() * 4;   // quads of words, one per lane (+ the row records of defect_rowdpp.h)
}

template <class Ode, int SCH, bool BLOCKED, int G>
struct LglMeta {
  using D = Dims<Ode, SCH, BLOCKED>;
  using UD = UnitsDims<D>;
  static constexpr long long v[MF_COUNT] = {
      1, Ode::XV, Ode::UV, Ode::PV, SCH, BLOCKED ? 1 : 0, D::IR, D::OR, D::NKKT, G, (long long)D::lds_bytes(), D::WSLOT, 0,
      (long long)D::lds_bytes_ode(), (long long)D::lds_bytes_dense(), D::WIDE ? 1 : 0,
      (D::lds_bytes_dense() * ASSET_WIDE_WGS <= 160 * 1024) ? ASSET_WIDE_WGS : 1,
      D::FUSED ? 1 : 0, D::GF, D::FUSED2 ? 1 : 0, D::GF2, (long long)D::lds_bytes_fused2(),
      Ode::NUNITS, (long long)D::TABSZ * 8, (long long)UD::MS * 8, D::CS,
      lgl_lane_table_bytes<Ode, SCH, BLOCKED>(1), lgl_lane_table_bytes<Ode, SCH, BLOCKED>(2),
      AdjDims<D>::GP, (long long)AdjDims<D>::lds_bytes(),
      ResDims<D>::OK ? ResDims<D>::GR : 0, (long long)ResDims<D>::lds_bytes(), res_lane_table_bytes<Ode, SCH, BLOCKED>(),
      ResDims<D>::WPS, ResDims<D>::GIVEN_OK ? ResDims<D>::GR : 0,
      RowsDims<D>::OK ? (long long)RowsDims<D>::lds_bytes() : 0, ResDims<D>::NWV, ResDims<D>::LOOP_PAIR ? 2 : 1, ResDims<D>::ASM_OK ? 1 : 0,
      ResDims<D>::RD_ALT ? 1 : 0, D::KL, D::KSTRIDE};
};

template <class F>
struct FuncMeta {
  using D = FuncDims<F>;
  // MF_G: applications per workgroup of the block kinds (func_kernels.h: FuncStage; 0: 64, stored directly), MF_LDS_BYTES: their LDS
  static constexpr long long v[MF_COUNT] = {2, F::XV, F::UV, F::PV, 0, 0, D::IR, D::OR, D::NKKT, FuncStage<F>::APW,
                                            (long long)FuncStage<F>::lds_bytes(), 0, F::NACONST};
};

// A bundle of plain functions (func_kernels.h: func_bundle_kernel): MF_KIND 3, MF_XV = number of functions, and from
// MF_BYTES_ODE on one word per function, IR * 65536 + OR, so that the loader can check the members it is given.
template <class... Fs>
struct BundleMeta {
  static_assert(MF_BYTES_ODE + BUNDLE_MAX <= MF_COUNT, "signature words fit the table");
  static constexpr long long v[MF_COUNT] = {3, sizeof...(Fs), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                            (long long)(FuncDims<Fs>::IR) * 65536 + FuncDims<Fs>::OR...};
};

}  // namespace asset_hip
