// Kernel registry: every (ODE functor, transcription mode, blocked) instantiation compiled into
// libasset_hip.so registers one entry; the C ABI (capi.hip) looks entries up by name at create time.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "defect_kernels.h"
#include "defect_wide.h"
#include "defect_units.h"
#include "func_kernels.h"
#include "mesh_kernels.h"

namespace asset_hip {

struct KernelEntry {
  const char* ode;
  int xv, uv, pv;
  int mode;     // ASSET_HIP_* transcription mode
  int blocked;
  int ir, orr, nkkt;
  int seg_per_group;      // segments whose ODE results one workgroup keeps in its workspace at a time
  size_t lds_bytes;
  size_t work_doubles;    // workspace doubles per segment (ODE result slot)
  // level 0/1/2 ; returns hipError_t
  hipError_t (*launch)(int level, const EvalArgs& a, int cus, hipStream_t st);
  KernelEntry* next;
  hipError_t (*mesh)(const MeshArgs& a, hipStream_t st);   // de Boor mesh-error estimate (mesh_kernels.h)
  // per-lane constants of the dense stage for derivative level 1 / 2: table size in bytes (0: none) and the kernel filling it
  size_t (*lane_bytes)(int level);
  hipError_t (*lane_setup)(int level, void* out, hipStream_t st);
  int naconst = 0;        // plain functions: constants per application the function reads (vf.ApplConst)
};

#if defined(ASSET_PLUGIN)
// A plugin (one run-time compiled translation unit, see asset_asrl_amd/jit.py) collects its entries in a list of
// its own and exports it through asset_hip_plugin_entries(); asset_hip_load_plugin() splices it into the registry
// of libasset_hip.so.  Nothing here touches the host library's symbols, so the plugin needs no link against it.
namespace {
KernelEntry* g_plugin_head = nullptr;
}
struct Registrar {
  explicit Registrar(KernelEntry* e) {
    e->next = g_plugin_head;
    g_plugin_head = e;
  }
};
#define ASSET_PLUGIN_EXPORT()                                                                            \
  extern "C" __attribute__((visibility("default"))) ::asset_hip::KernelEntry* asset_hip_plugin_entries() { \
    return ::asset_hip::g_plugin_head;                                                                   \
  }
#else
inline KernelEntry*& registry_head() {
  static KernelEntry* head = nullptr;
  return head;
}

struct Registrar {
  explicit Registrar(KernelEntry* e) {
    e->next = registry_head();
    registry_head() = e;
  }
};
#endif

template <class Ode, int SCH, bool BLOCKED, int G>
hipError_t launch_lgl(int level, const EvalArgs& a, int cus, hipStream_t st) {
  using D = Dims<Ode, SCH, BLOCKED>;
  constexpr size_t bytes_ode = D::lds_bytes_ode(), bytes_dense = D::lds_bytes_dense();
  static_assert(bytes_ode <= 160 * 1024 && bytes_dense <= 160 * 1024, "per-workgroup LDS exceeds the 160 KiB of a gfx950 CU");
  // ODE launch: the three ODE phases are latency chains, so spread the segments over every resident wave (fewest
  // passes per wave); a workgroup walks its share in groups of at most G segments (= 64 points of the widest phase)
  int per_cu_a = int((160 * 1024) / bytes_ode);
  per_cu_a = per_cu_a < 1 ? 1 : (per_cu_a >= 8 ? 8 : (per_cu_a >= 4 ? 4 : per_cu_a));
  const int grid_a = a.nseg < cus * per_cu_a ? a.nseg : cus * per_cu_a;
  // dense launch: persistent single-wave workgroups, as many as the LDS lets be resident
  // (an even number of waves per CU spreads evenly over the 4 SIMDs; 7 per CU measured 30% slower than 6)
  int per_cu_b = int((160 * 1024) / bytes_dense);
  per_cu_b = per_cu_b < 1 ? 1 : (per_cu_b >= 8 ? 8 : (per_cu_b >= 6 ? 6 : (per_cu_b >= 4 ? 4 : per_cu_b)));
  int grid_b = a.nseg < cus * per_cu_b ? a.nseg : cus * per_cu_b;
  const int wide_wgs = (bytes_dense * ASSET_WIDE_WGS <= 160 * 1024) ? ASSET_WIDE_WGS : 1;   // four-wave workgroups per CU (defect_wide.h)
  (void)wide_wgs;
  static const int env_b = std::getenv("ASSET_HIP_GRID_B") ? std::atoi(std::getenv("ASSET_HIP_GRID_B")) : 0;  // tuning only
  if (env_b > 0) grid_b = env_b < a.nseg ? env_b : a.nseg;
  static const bool skip_dense = std::getenv("ASSET_HIP_SKIP_DENSE") != nullptr;                               // tuning only
#define ASSET_LAUNCH(LV, STG, GRID, BYTES) ASSET_LAUNCH_K((lgl_defect_kernel<Ode, SCH, BLOCKED, G, LV, STG, false>), GRID, 64, BYTES)
  // dense stage: single-wave workgroups, or (wide shapes, defect_wide.h) one four-wave workgroup per CU
#define ASSET_LAUNCH_DENSE(LV, ASMV)                                                                              \
  do {                                                                                                            \
    if constexpr (D::WIDE) {                                                                                      \
      ASSET_LAUNCH_K((lgl_wide_dense_kernel<Ode, SCH, BLOCKED, LV, ASMV>), (a.nseg < cus * wide_wgs ? a.nseg : cus * wide_wgs), 256, bytes_dense); \
    } else {                                                                                                      \
      ASSET_LAUNCH_K((lgl_defect_kernel<Ode, SCH, BLOCKED, G, LV, 2, ASMV>), grid_b, 64, bytes_dense);            \
    }                                                                                                             \
  } while (0)
#define ASSET_LAUNCH_K(KERN, GRID, BLOCK, BYTES)                                                                       \
  do {                                                                                                            \
    auto kern = KERN;                                                                                             \
    if (BYTES > 64 * 1024) {                                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(BYTES));                 \
      if (e != hipSuccess) return e;                                                                              \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, dim3(GRID), dim3(BLOCK), BYTES, st, a);                                                \
    hipError_t e2 = hipGetLastError();                                                                            \
    if (e2 != hipSuccess) return e2;                                                                              \
  } while (0)
  switch (level) {
    case 0: ASSET_LAUNCH(0, 1, grid_a, bytes_ode); return hipSuccess;
    case 1:
      ASSET_LAUNCH(1, 1, grid_a, bytes_ode);
      if (skip_dense) return hipSuccess;
      if (a.kmap) ASSET_LAUNCH_DENSE(1, true);   // KKT entries added straight into the solver's value array
      else ASSET_LAUNCH_DENSE(1, false);
      return hipSuccess;
    case 2:
      if constexpr (D::FUSED) {
        // single launch when every workgroup's share fits one group of the fused kernel (defect_kernels.h, STAGE 3)
        static const bool no_fuse = std::getenv("ASSET_HIP_NO_FUSE") != nullptr;                               // tuning only
        if constexpr (D::FUSED2) {
          // two-wave workgroups: the ODE bodies are issued once per pair of waves (defect_kernels.h, STAGE 4)
          static const bool no_fuse2 = std::getenv("ASSET_HIP_NO_FUSE2") != nullptr;                           // tuning only
          const int pairs = grid_b / 2;
          // (measured, 10 000 segments: TwoBody-LGL5-BlockConstant 42.2 -> 39.6 us, Reentry-LGL7 43.2 -> 43.0 us; with 2-3
          //  segments per wave -- Reentry-LGL7 x 5 000 -- the pair's barriers cost more than the shared bodies save:
          //  29.6 -> 32.9 us, so short shares keep the one-wave form)
          const int share = (a.nseg + 2 * pairs - 1) / (pairs > 0 ? 2 * pairs : 1);
          if (!a.kmap && !no_fuse && !no_fuse2 && !skip_dense && pairs > 0 && share >= 4 && share <= D::GF2 / 2) {
            ASSET_LAUNCH_K((lgl_defect_kernel<Ode, SCH, BLOCKED, G, 2, 4, false>), pairs, 128, D::lds_bytes_fused2());
            return hipSuccess;
          }
        }
        if (!a.kmap && !no_fuse && !skip_dense && (a.nseg + grid_b - 1) / grid_b <= D::GF) {
          ASSET_LAUNCH_K((lgl_defect_kernel<Ode, SCH, BLOCKED, G, 2, 3, false>), grid_b, 64, bytes_dense);
          return hipSuccess;
        }
      }
      if constexpr (Ode::NUNITS > 1) {
        // heavy right-hand side: the ODE stage runs one wave per output unit (defect_units.h)
        static const bool no_units = std::getenv("ASSET_HIP_NO_UNITS") != nullptr;                             // tuning only
        if (!no_units) {
          constexpr int GPMAX = 64 / D::CS;
          // about one workgroup per SIMD: groups x units ~ 4 per CU
          int gp = (a.nseg * Ode::NUNITS + 4 * cus - 1) / (4 * cus);
          gp = gp < 1 ? 1 : (gp > GPMAX ? GPMAX : gp);
          const size_t bytes_units = UnitsDims<D>::lds_bytes(gp);
          const dim3 grid((a.nseg + gp - 1) / gp, Ode::NUNITS);
          hipLaunchKernelGGL((lgl_ode_units_kernel<Ode, SCH, BLOCKED, 0>), grid, dim3(64), bytes_units, st, a, gp);
          hipLaunchKernelGGL((lgl_ode_units_kernel<Ode, SCH, BLOCKED, 1>), grid, dim3(64), bytes_units, st, a, gp);
          hipError_t e2 = hipGetLastError();
          if (e2 != hipSuccess) return e2;
        } else {
          ASSET_LAUNCH(2, 1, grid_a, bytes_ode);
        }
      } else {
        ASSET_LAUNCH(2, 1, grid_a, bytes_ode);
      }
      if (skip_dense) return hipSuccess;
      if (a.kmap) ASSET_LAUNCH_DENSE(2, true);
      else ASSET_LAUNCH_DENSE(2, false);
      return hipSuccess;
  }
#undef ASSET_LAUNCH
#undef ASSET_LAUNCH_DENSE
#undef ASSET_LAUNCH_K
  return hipErrorInvalidValue;
}

// A plain function batched over applications: transcription id 0 (func_kernels.h)
#define ASSET_REGISTER_FUNC(FN)                                                                                   \
  static ::asset_hip::KernelEntry entry_##FN##_func = {                                                           \
      FN::name(), FN::XV, FN::UV, FN::PV, 0, 0,                                                                   \
      ::asset_hip::FuncDims<FN>::IR, ::asset_hip::FuncDims<FN>::OR, ::asset_hip::FuncDims<FN>::NKKT, 0, 0, 0,    \
      &::asset_hip::launch_func<FN>, nullptr, nullptr, nullptr, nullptr, FN::NACONST};                            \
  static ::asset_hip::Registrar reg_##FN##_func(&entry_##FN##_func);

template <class Ode, int SCH, bool BLOCKED>
size_t lgl_lane_bytes(int level) {
  using D = Dims<Ode, SCH, BLOCKED>;
  // bytes of the whole table: 64 word-interleaved records (defect_kernels.h: lane_setup_kernel)
  if constexpr (D::WIDE) return level >= 1 ? size_t(D::TJ) * D::TI * 4 * 64 * sizeof(unsigned int) : 0;   // defect_wide.h: wide_setup_kernel
  else return level >= 2 ? sizeof(LaneConsts<Ode, D, 2>) * 64 : (level == 1 ? sizeof(LaneConsts<Ode, D, 1>) * 64 : 0);
}
template <class Ode, int SCH, bool BLOCKED>
hipError_t lgl_lane_setup(int level, void* out, hipStream_t st) {
  if constexpr (Dims<Ode, SCH, BLOCKED>::WIDE) {
    hipLaunchKernelGGL((wide_setup_kernel<Ode, SCH, BLOCKED>), dim3(1), dim3(64), 0, st, static_cast<unsigned int*>(out));
    return hipGetLastError();
  } else if (level >= 2)
    hipLaunchKernelGGL((lane_setup_kernel<Ode, SCH, BLOCKED, 2>), dim3(1), dim3(64), 0, st, static_cast<unsigned int*>(out));
  else
    hipLaunchKernelGGL((lane_setup_kernel<Ode, SCH, BLOCKED, 1>), dim3(1), dim3(64), 0, st, static_cast<unsigned int*>(out));
  return hipGetLastError();
}

// Trapezoidal = transcription id 1 of the same kernels (defect_dims.h: Dims::TRAP)
#define ASSET_REGISTER_TRAP(ODE, BLK, G) ASSET_REGISTER_LGL(ODE, 1, BLK, G)

#define ASSET_REGISTER_LGL(ODE, CSV, BLK, G)                                                                      \
  static ::asset_hip::KernelEntry entry_##ODE##_##CSV##_##BLK = {                                                 \
      ODE::name(), ODE::XV, ODE::UV, ODE::PV, CSV, BLK,                                                           \
      ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::IR, ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::OR,                   \
      ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::NKKT, G,                                                           \
      ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::lds_bytes(),                                                       \
      size_t(::asset_hip::Dims<ODE, CSV, (BLK != 0)>::WSLOT),                                                      \
      &::asset_hip::launch_lgl<ODE, CSV, (BLK != 0), G>, nullptr,        \
      &::asset_hip::launch_mesh<ODE, CSV, (BLK != 0)>, &::asset_hip::lgl_lane_bytes<ODE, CSV, (BLK != 0)>,         \
      &::asset_hip::lgl_lane_setup<ODE, CSV, (BLK != 0)>};                                                         \
  static ::asset_hip::Registrar reg_##ODE##_##CSV##_##BLK(&entry_##ODE##_##CSV##_##BLK);

}  // namespace asset_hip
