// Device side of a module compiled at run time, in process (hiprtc; asset_asrl_amd/jit.py, capi.hip: asset_hip_jit_plugin).
// The generated source is
//     #include "ode.h"                      the generated functor
//     #include "<csrc>/rtc_device.h"
//     ASSET_RTC_LGL(OdeName, CSV, BLK, G)   or   ASSET_RTC_FUNC(FnName)   or   ASSET_RTC_BUNDLE(Fn0, Fn1, ...)
// and the loader names the kernels it wants (hiprtcAddNameExpression): every variant the launcher of registry.h may use.
// Variants a shape does not have compile to empty kernels (lgl_variant_valid, the guards of the wide / units / setup
// kernels), so the list does not depend on the shape.
#pragma once
#include "kernel_meta.h"
#include "mesh_kernels.h"

#define ASSET_RTC_LGL(ODE, CSV, BLK, G)                                                                            \
  extern "C" __device__ const long long asset_rtc_meta[::asset_hip::MF_COUNT] = {                                  \
      ASSET_RTC_META_LIST((::asset_hip::LglMeta<ODE, CSV, (BLK != 0), G>::v))};
#define ASSET_RTC_FUNC(FN)                                                                                         \
  extern "C" __device__ const long long asset_rtc_meta[::asset_hip::MF_COUNT] = {                                  \
      ASSET_RTC_META_LIST((::asset_hip::FuncMeta<FN>::v))};
#define ASSET_RTC_BUNDLE(...)                                                                                      \
  extern "C" __device__ const long long asset_rtc_meta[::asset_hip::MF_COUNT] = {                                  \
      ASSET_RTC_META_LIST((::asset_hip::BundleMeta<__VA_ARGS__>::v))};
// (an array cannot be initialised from another array: spell the elements out)
#define ASSET_RTC_META_LIST(V)                                                                                     \
  V[0], V[1], V[2], V[3], V[4], V[5], V[6], V[7], V[8], V[9], V[10], V[11], V[12], V[13], V[14], V[15], V[16], V[17],  \
      V[18], V[19], V[20], V[21], V[22], V[23], V[24], V[25], V[26], V[27], V[28], V[29], V[30], V[31], V[32], V[33], V[34], V[35], V[36], V[37], V[38], V[39], V[40], V[41]
static_assert(::asset_hip::MF_COUNT == 42, "ASSET_RTC_META_LIST spells out MF_COUNT elements");
