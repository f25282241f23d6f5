// Kernel registry: every (ODE functor, transcription mode, blocked) instantiation compiled into
// libasset_hip.so registers one entry; the C ABI (capi.hip) looks entries up by name at create time.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>

#include "defect_kernels.h"

namespace asset_hip {

struct KernelEntry {
  const char* ode;
  int xv, uv, pv;
  int mode;     // ASSET_HIP_* transcription mode
  int blocked;
  int ir, orr, nkkt;
  int seg_per_group;      // segments whose ODE results one workgroup keeps in its workspace at a time
  size_t lds_bytes;
  size_t work_doubles;    // workspace doubles per workgroup
  // level 0/1/2 ; returns hipError_t
  hipError_t (*launch)(int level, const EvalArgs& a, int grid, hipStream_t st);
  KernelEntry* next;
};

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

template <class Ode, int CS, bool BLOCKED, int G>
hipError_t launch_lgl(int level, const EvalArgs& a, int grid, hipStream_t st) {
  using D = Dims<Ode, CS, BLOCKED>;
  constexpr size_t bytes = D::lds_bytes();
  static_assert(bytes <= 160 * 1024, "per-workgroup LDS exceeds the 160 KiB of a gfx950 CU");
#define ASSET_LAUNCH(LV)                                                                                          \
  do {                                                                                                            \
    auto kern = lgl_defect_kernel<Ode, CS, BLOCKED, G, LV>;                                                       \
    if (bytes > 64 * 1024) {                                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));                 \
      if (e != hipSuccess) return e;                                                                              \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), bytes, st, a);                                                 \
    return hipGetLastError();                                                                                     \
  } while (0)
  switch (level) {
    case 0: ASSET_LAUNCH(0);
    case 1: ASSET_LAUNCH(1);
    case 2: ASSET_LAUNCH(2);
  }
#undef ASSET_LAUNCH
  return hipErrorInvalidValue;
}

template <class Ode, bool BLOCKED, int G>
hipError_t launch_trap(int level, const EvalArgs& a, int grid, hipStream_t st) {
  using D = TrapDims<Ode, BLOCKED>;
  constexpr size_t bytes = D::template lds_bytes<G>();
  static_assert(bytes <= 64 * 1024, "trapezoidal group does not fit the default dynamic LDS window");
  switch (level) {
    case 0: hipLaunchKernelGGL((trap_defect_kernel<Ode, BLOCKED, G, 0>), dim3(grid), dim3(64), bytes, st, a); break;
    case 1: hipLaunchKernelGGL((trap_defect_kernel<Ode, BLOCKED, G, 1>), dim3(grid), dim3(64), bytes, st, a); break;
    case 2: hipLaunchKernelGGL((trap_defect_kernel<Ode, BLOCKED, G, 2>), dim3(grid), dim3(64), bytes, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

#define ASSET_REGISTER_TRAP(ODE, BLK, G)                                                                          \
  static ::asset_hip::KernelEntry entry_##ODE##_trap_##BLK = {                                                    \
      ODE::name(), ODE::XV, ODE::UV, ODE::PV, 1, BLK,                                                             \
      ::asset_hip::TrapDims<ODE, (BLK != 0)>::IR, ::asset_hip::TrapDims<ODE, (BLK != 0)>::OR,                     \
      ::asset_hip::TrapDims<ODE, (BLK != 0)>::NKKT, G,                                                            \
      ::asset_hip::TrapDims<ODE, (BLK != 0)>::template lds_bytes<G>(), 0,                                         \
      &::asset_hip::launch_trap<ODE, (BLK != 0), G>, nullptr};                                                    \
  static ::asset_hip::Registrar reg_##ODE##_trap_##BLK(&entry_##ODE##_trap_##BLK);

#define ASSET_REGISTER_LGL(ODE, CSV, BLK, G)                                                                      \
  static ::asset_hip::KernelEntry entry_##ODE##_##CSV##_##BLK = {                                                 \
      ODE::name(), ODE::XV, ODE::UV, ODE::PV, CSV, BLK,                                                           \
      ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::IR, ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::OR,                   \
      ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::NKKT, G,                                                           \
      ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::lds_bytes(),                                                       \
      size_t(G) * ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::SLOT,                                                  \
      &::asset_hip::launch_lgl<ODE, CSV, (BLK != 0), G>, nullptr};                                                \
  static ::asset_hip::Registrar reg_##ODE##_##CSV##_##BLK(&entry_##ODE##_##CSV##_##BLK);

}  // namespace asset_hip
