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
  int seg_per_group;
  size_t lds_bytes;
  // level 0/1/2 ; returns hipError_t
  hipError_t (*launch)(int level, bool mfma, const EvalArgs& a, int grid, hipStream_t st);
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
hipError_t launch_lgl(int level, bool mfma, const EvalArgs& a, int grid, hipStream_t st) {
  using D = Dims<Ode, CS, BLOCKED>;
  constexpr size_t bytes = D::template lds_bytes<G>();
  static_assert(bytes <= 160 * 1024, "per-workgroup LDS exceeds the 160 KiB of a gfx950 CU");
#define ASSET_LAUNCH(LV, MF)                                                                                      \
  do {                                                                                                            \
    auto kern = lgl_defect_kernel<Ode, CS, BLOCKED, G, LV, MF>;                                                   \
    if (bytes > 64 * 1024) {                                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));                 \
      if (e != hipSuccess) return e;                                                                              \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), bytes, st, a);                                                 \
    return hipGetLastError();                                                                                     \
  } while (0)
  switch (level * 2 + (mfma ? 1 : 0)) {
    case 0: case 1: ASSET_LAUNCH(0, true);
    case 2: ASSET_LAUNCH(1, false);
    case 3: ASSET_LAUNCH(1, true);
    case 4: ASSET_LAUNCH(2, false);
    case 5: ASSET_LAUNCH(2, true);
  }
#undef ASSET_LAUNCH
  return hipErrorInvalidValue;
}

#define ASSET_REGISTER_LGL(ODE, CSV, BLK, G)                                                                      \
  static ::asset_hip::KernelEntry entry_##ODE##_##CSV##_##BLK = {                                                 \
      ODE::name(), ODE::XV, ODE::UV, ODE::PV, CSV, BLK,                                                           \
      ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::IR, ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::OR,                   \
      ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::NKKT, G,                                                           \
      ::asset_hip::Dims<ODE, CSV, (BLK != 0)>::template lds_bytes<G>(),                                           \
      &::asset_hip::launch_lgl<ODE, CSV, (BLK != 0), G>, nullptr};                                                \
  static ::asset_hip::Registrar reg_##ODE##_##CSV##_##BLK(&entry_##ODE##_##CSV##_##BLK);

}  // namespace asset_hip
