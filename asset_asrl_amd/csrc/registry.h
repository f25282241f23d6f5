// Kernel registry: every (ODE functor, transcription mode, blocked) instantiation compiled into libasset_hip.so registers
// one entry; so does every module compiled at run time (capi.hip: asset_hip_jit_plugin).  The C ABI (capi.hip) looks
// entries up by name at create time.  An entry is data -- the integers of kernel_meta.h and one reference per kernel
// variant -- and ONE launcher (launch_lgl_table below) serves the kernels linked into the library (host stubs) and the
// kernels of a run-time module (hipFunction_t) alike.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "kernel_meta.h"
#include "mesh_kernels.h"

namespace asset_hip {

// ---- kernel references ---------------------------------------------------------------------------------------------
// A module compiled at run time: the code object and the lowered kernel names stay with it, and it is loaded on every
// device it is used on -- a hipModule_t (and its hipFunction_t handles) belongs to the device that was current when it
// was loaded, while a handle of the C ABI may live on any device (asset_hip_defect_desc::device).  Loading is lazy: the
// first launch on a device loads the code object there.
struct RtcModule {
  std::vector<char> code;
  std::vector<std::pair<int, std::string>> names;   // (kernel slot, lowered name)
  struct PerDevice {
    hipModule_t mod = nullptr;
    std::vector<hipFunction_t> fn;                  // by kernel slot
  };
  std::mutex m;
  std::vector<PerDevice> dev;                       // by device ordinal; sized ONCE (hipGetDeviceCount) and never resized
  ~RtcModule() {
    for (auto& pd : dev)
      if (pd.mod) (void)hipModuleUnload(pd.mod);
  }
  // kernel `slot` on the CURRENT device (loads the module there at first use).  The function handle is copied out under the
  // lock: no pointer into `dev` outlives it.
  hipError_t on_current_device(int slot, hipFunction_t* out) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(m);
    if (dev.empty()) {
      int nd = 0;
      if ((e = hipGetDeviceCount(&nd)) != hipSuccess) return e;
      dev.resize(nd > d + 1 ? nd : d + 1);
    }
    if (d >= int(dev.size())) return hipErrorInvalidDevice;
    PerDevice& pd = dev[d];
    if (!pd.mod) {
      hipModule_t mod = nullptr;
      if ((e = hipModuleLoadData(&mod, code.data())) != hipSuccess) return e;
      int nslot = 0;
      for (auto& n : names) nslot = n.first + 1 > nslot ? n.first + 1 : nslot;
      std::vector<hipFunction_t> fn(nslot, nullptr);
      for (auto& n : names)
        if ((e = hipModuleGetFunction(&fn[n.first], mod, n.second.c_str())) != hipSuccess) {
          hipModuleUnload(mod);
          return e;
        }
      pd.fn.swap(fn);
      pd.mod = mod;
    }
    if (slot < 0 || slot >= int(pd.fn.size()) || !pd.fn[slot]) return hipErrorInvalidDeviceFunction;
    *out = pd.fn[slot];
    return hipSuccess;
  }
  // the module handle of the current device (loaded if need be): for reading the module's constants
  hipError_t module_on_current_device(hipModule_t* out) {
    int first = -1;
    for (auto& n : names) { first = n.first; break; }
    hipFunction_t f = nullptr;
    if (first >= 0) {
      hipError_t e = on_current_device(first, &f);
      if (e != hipSuccess) return e;
    }
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(m);
    if (d >= int(dev.size()) || !dev[d].mod) return hipErrorInvalidValue;
    *out = dev[d].mod;
    return hipSuccess;
  }
};

struct KRef {
  const void* host = nullptr;   // host stub of a kernel linked into this process (hipLaunchKernel)
  RtcModule* rtc = nullptr;     // kernel `slot` of a run-time module (hipModuleLaunchKernel on the current device)
  int slot = -1;
  explicit operator bool() const { return host || rtc; }
};
inline hipError_t klaunch(const KRef& k, dim3 grid, dim3 block, size_t shmem, hipStream_t st, void** args) {
  if (k.rtc) {
    hipFunction_t fn = nullptr;
    hipError_t e = k.rtc->on_current_device(k.slot, &fn);
    if (e != hipSuccess) return e;
    return hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, unsigned(shmem), st, args, nullptr);
  }
  if (!k.host) return hipErrorInvalidDeviceFunction;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(k.host, hipFuncAttributeMaxDynamicSharedMemorySize, int(shmem));
    if (e != hipSuccess) return e;
  }
  return hipLaunchKernel(k.host, grid, block, args, shmem, st);
}

// kernel slots of a table (the same numbering for the static fill and for the name expressions of a run-time module)
constexpr int K_LGL(int level, int stage, bool asmb) { return level * 8 + (stage - 1) * 2 + (asmb ? 1 : 0); }   // 0..23
constexpr int K_WIDE(int level, bool asmb) { return 24 + (level - 1) * 2 + (asmb ? 1 : 0); }                     // 24..27
constexpr int K_WIDE_SETUP = 28, K_LANE_SETUP1 = 29, K_LANE_SETUP2 = 30, K_UNITS0 = 31, K_UNITS1 = 32;
constexpr int K_MESH_YVEC = 33, K_MESH_ERROR = 34;
constexpr int K_FUNC(int level, bool asmb) { return 35 + level * 2 + (asmb ? 1 : 0); }                           // 35..40
constexpr int K_BUNDLE(int level) { return 41 + level; }                                                         // 41..43
constexpr int K_ADJGRAD = 44;   // value + adjoint gradient without a Jacobian (defect_adjgrad.h)
constexpr int K_VALUE = 45;     // value only, the same kernel without the gradient parts
constexpr int K_RES(bool asmb) { return 46 + (asmb ? 1 : 0); }   // resident single launch (defect_resident.h), level 2
constexpr int K_RES_SETUP = 48;
constexpr int K_RES1(bool asmb) { return 49 + (asmb ? 1 : 0); }  // ... the Jacobian kinds
constexpr int K_RESL(int level, bool asmb) { return 51 + (level - 1) * 2 + (asmb ? 1 : 0); }   // ... looped over groups (large meshes)
constexpr int K_RESD(bool asmb) { return 55 + (asmb ? 1 : 0); }   // ... its dense part alone, slots from the workspace (heavy ODEs)
constexpr int K_UNITSJ = 57;   // heavy right-hand sides, Jacobian kinds: the unit kernel of the ODE stage (defect_units.h, PHASE 3)
constexpr int K_ROWS = 58;     // wide shapes: dense stage by output rows, no matrix instructions (defect_rows.h)
constexpr int K_ROWS1 = 59;    // ... the Jacobian kinds
constexpr int K_UNITS4 = 60;   // heavy right-hand sides: interior and cardinal units in one launch (defect_units.h, PHASE 4)
constexpr int K_RESLP = 61;   // resident kernel, looped, level 2, blocks, as two-wave workgroups (row-wise dense part: defect_rowdpp.h)
constexpr int K_RES_ALT = 62;   // resident kernel, one group, level 2, blocks: the row-wise dense part of a shape that defaults to tiles
constexpr int K_COUNT = 63;

struct KernelTable {
  long long meta[MF_COUNT] = {};
  KRef k[K_COUNT];
};

struct KernelEntry {
  const char* ode;
  int xv, uv, pv;
  int mode;     // ASSET_HIP_* transcription mode
  int blocked;
  int ir, orr, nkkt;
  int kl, kstride;        // layout of the KKT blocks the kernels write (defect_dims.h: Dims::KL) and the block stride in doubles
  int seg_per_group;      // segments whose ODE results one workgroup keeps in its workspace at a time
  size_t lds_bytes;
  size_t work_doubles;    // workspace doubles per segment (ODE result slot)
  int naconst;            // plain functions: constants per application the function reads (vf.ApplConst)
  const KernelTable* table;
  KernelEntry* next;
};
inline void entry_from_table(KernelEntry& e, const char* name, const KernelTable* t) {
  const long long* m = t->meta;
  e.ode = name, e.xv = int(m[MF_XV]), e.uv = int(m[MF_UV]), e.pv = int(m[MF_PV]), e.mode = int(m[MF_MODE]);
  e.blocked = int(m[MF_BLOCKED]), e.ir = int(m[MF_IR]), e.orr = int(m[MF_OR]), e.nkkt = int(m[MF_NKKT]);
  e.kl = int(m[MF_KL]), e.kstride = m[MF_KSTRIDE] > 0 ? int(m[MF_KSTRIDE]) : e.nkkt;   // (plain functions, bundles: the reference's order)
  e.seg_per_group = int(m[MF_G]), e.lds_bytes = size_t(m[MF_LDS_BYTES]), e.work_doubles = size_t(m[MF_WORK_DOUBLES]);
  e.naconst = int(m[MF_NACONST]), e.table = t, e.next = nullptr;
}

#if defined(ASSET_PLUGIN)
// A plugin (one translation unit compiled by hipcc at run time: asset_asrl_amd/jit.py with ASSET_HIP_JIT=hipcc) collects
// its entries in a list of its own and exports it through asset_hip_plugin_entries(); asset_hip_load_plugin() splices it
// into the registry of libasset_hip.so.  Nothing here touches the host library's symbols, so the plugin needs no link
// against it.
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

// ---- the launcher ---------------------------------------------------------------------------------------------------
#define ASSET_UNITS_ONE_LAUNCH_ROUNDS 1000   // heavy ODEs: the one-launch unit stage ALWAYS (the limit, in rounds of the SIMDs, is beyond every mesh) -- measured faster at every size
                                            // tried (Betts-LGL5 x 1 000: 30.3 against 41.5 us, x 10 000: 191.7 / 200.0; Betts-LGL7 x 5 000: 145.1 / 151.1)
// Dispatch knobs of the measurement scripts (tools/): read ONLY when the process opts in with ASSET_HIP_TUNING=1, so that a
// stray variable in a production environment cannot change which kernels run.  With the opt-in, every knob that takes effect is
// reported once on stderr.
inline const char* tuning_env(const char* name) {
  static const bool on = [] { const char* v = std::getenv("ASSET_HIP_TUNING"); return v && std::atoi(v) != 0; }();
  if (!on) {
    // a knob that is set but not in effect: say so once, so that a measurement script which forgot the opt-in does not quote
    // the default path under the knob's name
    static bool warned = false;
    if (!warned && std::getenv(name)) {
      warned = true;
      std::fprintf(stderr, "asset_hip: %s is set but IGNORED (dispatch knobs need ASSET_HIP_TUNING=1)\n", name);
    }
    return nullptr;
  }
  const char* v = std::getenv(name);
  if (v) std::fprintf(stderr, "asset_hip: tuning knob %s=%s is in effect (ASSET_HIP_TUNING=1)\n", name, v);
  return v;
}
inline hipError_t launch_lgl_table(const KernelTable& t, int level, const EvalArgs& a, int cus, hipStream_t st) {
  const long long* m = t.meta;
  const size_t bytes_ode = size_t(m[MF_BYTES_ODE]), bytes_dense = size_t(m[MF_BYTES_DENSE]);
  const bool wide = m[MF_WIDE] != 0;
  EvalArgs args = a;
  void* kargs[] = {&args};
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
  const int wide_wgs = int(m[MF_WIDE_WGS]);   // four-wave workgroups per CU (defect_wide.h)
  static const int env_b = tuning_env("ASSET_HIP_GRID_B") ? std::atoi(std::getenv("ASSET_HIP_GRID_B")) : 0;  // tuning only
  if (env_b > 0) grid_b = env_b < a.nseg ? env_b : a.nseg;
  static const bool skip_dense = tuning_env("ASSET_HIP_SKIP_DENSE") != nullptr;                               // tuning only
  auto ode_stage = [&](int lv) { return klaunch(t.k[K_LGL(lv, 1, false)], dim3(grid_a), dim3(64), bytes_ode, st, kargs); };
  // dense stage: single-wave workgroups, or (wide shapes, defect_wide.h) one four-wave workgroup per CU
  auto dense_stage = [&](int lv) {
    const bool asmb = a.kmap != nullptr;   // KKT entries added straight into the solver's value array
    static const bool no_rows = tuning_env("ASSET_HIP_NO_ROWS") != nullptr;                                   // tuning only
    // (LGL3: two nodes, IR = 2 q -- one of the two H blocks nearly empty: 487 against 399 us for 12 500 32-state segments; kept
    //  with the tile kernel)
    if (wide && lv >= 1 && !asmb && !no_rows && m[MF_ROWS_LDS_BYTES] > 0 && m[MF_CS] >= 3 && t.k[lv == 2 ? K_ROWS : K_ROWS1]) {
      const double* work_ro = a.work;
      void* rargs[] = {&args, &work_ro};
      return klaunch(t.k[lv == 2 ? K_ROWS : K_ROWS1], dim3(a.nseg < cus ? a.nseg : cus), dim3(256), size_t(m[MF_ROWS_LDS_BYTES]), st, rargs);
    }
    if (wide)
      return klaunch(t.k[K_WIDE(lv, asmb)], dim3(a.nseg < cus * wide_wgs ? a.nseg : cus * wide_wgs), dim3(256), bytes_dense, st, kargs);
    return klaunch(t.k[K_LGL(lv, 2, asmb)], dim3(grid_b), dim3(64), bytes_dense, st, kargs);
  };
  hipError_t e;
  // constraints_adjointgradient (evalRHS): value and J^T lam wanted, no Jacobian -- one launch of the vector-Jacobian kernel
  static const bool no_adj = tuning_env("ASSET_HIP_NO_ADJGRAD_KERNEL") != nullptr;                            // tuning only
  if (level == 1 && !a.KKT && !a.kmap && a.AGX && a.L && !no_adj && t.k[K_ADJGRAD]) {
    const int gp = int(m[MF_ADJ_GP]);
    return klaunch(t.k[K_ADJGRAD], dim3((a.nseg + gp - 1) / gp), dim3(64), size_t(m[MF_ADJ_LDS_BYTES]), st, kargs);
  }
  if (level == 0 && !no_adj && t.k[K_VALUE]) {   // constraints (evalOCC): the same kernel without its gradient parts
    const int gp = int(m[MF_ADJ_GP]);
    return klaunch(t.k[K_VALUE], dim3((a.nseg + gp - 1) / gp), dim3(64), size_t(m[MF_ADJ_LDS_BYTES]), st, kargs);
  }
  switch (level) {
    case 0: return ode_stage(0);
    case 1: {
      static const bool no_res1 = tuning_env("ASSET_HIP_NO_RESIDENT") != nullptr;                             // tuning only
      if (m[MF_RES_GR] > 0 && !no_res1 && !skip_dense && a.lane_consts_res && t.k[K_RES1(a.kmap != nullptr)] && (!a.kmap || m[MF_RES_ASM])) {
        const int waves = cus * 4 * int(m[MF_RES_WPS]);   // resident kernel, Jacobian kinds (defect_resident.h, LEVEL 1)
        const bool one = (a.nseg + waves - 1) / waves <= int(m[MF_RES_GR]);
        const KRef& kr = one ? t.k[K_RES1(a.kmap != nullptr)] : t.k[K_RESL(1, a.kmap != nullptr)];
        // (the one-group kernel of a pair shape: two waves per workgroup, a region of LDS each; the looped one: single waves)
        const int nwv = (one && m[MF_RES_NWV] > 1) ? 2 : 1, nw = a.nseg < waves ? a.nseg : waves;
        const size_t lds = size_t(m[MF_RES_LDS_BYTES]) / size_t(m[MF_RES_NWV] > 1 ? 2 : 1) * size_t(nwv);
        if (kr) return klaunch(kr, dim3((nw + nwv - 1) / nwv), dim3(64 * nwv), lds, st, kargs);
      }
      static const bool no_fuse1 = tuning_env("ASSET_HIP_NO_FUSE") != nullptr;                                // tuning only
      if (m[MF_FUSED] && !no_fuse1 && !skip_dense && (a.nseg + grid_b - 1) / grid_b <= int(m[MF_GF]) && t.k[K_LGL(1, 3, a.kmap != nullptr)])
        return klaunch(t.k[K_LGL(1, 3, a.kmap != nullptr)], dim3(grid_b), dim3(64), bytes_dense, st, kargs);   // one launch
      static const bool no_units1 = tuning_env("ASSET_HIP_NO_UNITS") != nullptr;                              // tuning only
      if (m[MF_NUNITS] > 1 && !no_units1 && t.k[K_UNITSJ] && a.nseg * int(m[MF_CS]) <= 64 * cus) {
        // heavy right-hand side: one wave per output unit (defect_units.h, PHASE 3), one launch -- while the mesh leaves SIMDs
        // idle: the units recompute what they share, and from ~16 cardinal points per SIMD on the one-body-per-lane stage
        // is the faster one (Betts-LGL5: 1 000 segments 21.5 against 48.1 us, 5 000: 48.8 / 53.8, 10 000: 85.0 / 72.7)
        const int nunits = int(m[MF_NUNITS]), gpmax = 64 / int(m[MF_CS]);
        int gp = (a.nseg * nunits + 4 * cus - 1) / (4 * cus);
        gp = gp < 1 ? 1 : (gp > gpmax ? gpmax : gp);
        const size_t bytes_units = size_t(m[MF_UNITS_BASE_BYTES]) + size_t(gp) * size_t(m[MF_UNITS_SLOT_BYTES]);
        void* uargs[] = {&args, &gp};
        if ((e = klaunch(t.k[K_UNITSJ], dim3((a.nseg + gp - 1) / gp, nunits), dim3(64), bytes_units, st, uargs)) != hipSuccess) return e;
      } else if ((e = ode_stage(1)) != hipSuccess) {
        return e;
      }
      if (skip_dense) return hipSuccess;
      return dense_stage(1);
    }
    case 2: {
      // resident kernel (defect_resident.h): the ODE results stay in LDS; meshes of at most GR segments per wave
      static const bool no_res = tuning_env("ASSET_HIP_NO_RESIDENT") != nullptr;                              // tuning only
      if (m[MF_RES_GR] > 0 && !no_res && !skip_dense && a.lane_consts_res && t.k[K_RES(a.kmap != nullptr)] && (!a.kmap || m[MF_RES_ASM])) {
        int waves = cus * 4 * int(m[MF_RES_WPS]);   // one group per wave up to GR segments per wave, the looped instantiation beyond
        static const int env_max = tuning_env("ASSET_HIP_RESIDENT_MAX_GROUPS") ? std::atoi(std::getenv("ASSET_HIP_RESIDENT_MAX_GROUPS")) : 0;   // tuning only
        static const int env_grid = tuning_env("ASSET_HIP_RESIDENT_GRID") ? std::atoi(std::getenv("ASSET_HIP_RESIDENT_GRID")) : 0;   // tuning only
        if (env_grid > 0) waves = env_grid;
        const bool one = (a.nseg + waves - 1) / waves <= int(m[MF_RES_GR]);
        const KRef& kr = one ? t.k[K_RES(a.kmap != nullptr)] : t.k[K_RESL(2, a.kmap != nullptr)];
        if (kr && (one || env_max <= 0 || (a.nseg + waves - 1) / waves <= env_max * int(m[MF_RES_GR])))
        {
          // two-wave workgroups: the one-group kernel of a pair shape -- and, on every looped mesh, the looped block kernel of the shapes
          // that are built with one (ResDims::LOOP_PAIR: row-wise dense part, a right-hand side heavy enough for the shared ODE stage to
          // pay for the pair's barriers -- profiles/r6_forms2.txt)
          static const bool no_alt = tuning_env("ASSET_HIP_NO_ALT_FORM") != nullptr;                           // tuning only
          static const int lpair_min = tuning_env("ASSET_HIP_LPAIR_MIN") ? std::atoi(std::getenv("ASSET_HIP_LPAIR_MIN")) : 1;   // tuning only (groups per wave from which the pair form is taken)
          static const int alt_min = tuning_env("ASSET_HIP_ALT_MIN") ? std::atoi(std::getenv("ASSET_HIP_ALT_MIN")) : 5;         // tuning only (HALF segments per workgroup)
          const bool lpair = !one && !a.kmap && m[MF_RES_LOOP_NWV] > 1 && t.k[K_RESLP] && !(no_alt && m[MF_RES_ALT]) &&
                             (a.nseg + waves - 1) / waves >= lpair_min * int(m[MF_RES_GR]);
          const int nwv = ((one && m[MF_RES_NWV] > 1) || lpair) ? 2 : 1, nw = a.nseg < waves ? a.nseg : waves;
          const size_t lds = size_t(m[MF_RES_LDS_BYTES]) / size_t(m[MF_RES_NWV] > 1 ? 2 : 1) * size_t(nwv);
          // shapes with both forms of the dense part (ResDims::RD_ALT): rows in the one-group kernel from two and a half segments per
          // workgroup on (with UNITC the row-wise part is the cheaper one wherever its passes -- four / two segments -- are not mostly
          // empty: Reentry-LGL7 x 2 500 15.8 against 16.1 us, x 5 000 20.5 / 20.9, x 10 000 26.4 / 28.4; x 1 000 13.8 / 13.2), tiles below;
          // looped meshes: the looped pair kernel (rows) -- profiles/r6_forms2.txt
          const int nwg = (nw + nwv - 1) / nwv;
          const bool alt = one && nwv == 2 && !a.kmap && !no_alt && m[MF_RES_ALT] && t.k[K_RES_ALT] && 2 * a.nseg >= alt_min * nwg;
          return klaunch(lpair ? t.k[K_RESLP] : (alt ? t.k[K_RES_ALT] : kr), dim3(nwg), dim3(64 * nwv), lds, st, kargs);
        }
      }
      if (m[MF_FUSED]) {
        // single launch when every workgroup's share fits one group of the fused kernel (defect_kernels.h, STAGE 3)
        static const bool no_fuse = tuning_env("ASSET_HIP_NO_FUSE") != nullptr;                               // tuning only
        if (m[MF_FUSED2]) {
          // two-wave workgroups: the ODE bodies are issued once per pair of waves (defect_kernels.h, STAGE 4)
          static const bool no_fuse2 = tuning_env("ASSET_HIP_NO_FUSE2") != nullptr;                           // tuning only
          const int pairs = grid_b / 2;
          // (measured, 10 000 segments: TwoBody-LGL5-BlockConstant 42.2 -> 39.6 us, Reentry-LGL7 43.2 -> 43.0 us; with 2-3
          //  segments per wave -- Reentry-LGL7 x 5 000 -- the pair's barriers cost more than the shared bodies save:
          //  29.6 -> 32.9 us, so short shares keep the one-wave form)
          const int share = (a.nseg + 2 * pairs - 1) / (pairs > 0 ? 2 * pairs : 1);
          if (!a.kmap && !no_fuse && !no_fuse2 && !skip_dense && pairs > 0 && share >= 4 && share <= int(m[MF_GF2]) / 2)
            return klaunch(t.k[K_LGL(2, 4, false)], dim3(pairs), dim3(128), size_t(m[MF_BYTES_FUSED2]), st, kargs);
        }
        // (also with on-device assembly: the dense part places its entries through the map either way)
        if (!no_fuse && !skip_dense && (a.nseg + grid_b - 1) / grid_b <= int(m[MF_GF]))
          return klaunch(t.k[K_LGL(2, 3, a.kmap != nullptr)], dim3(grid_b), dim3(64), bytes_dense, st, kargs);
      }
      static const bool no_units = tuning_env("ASSET_HIP_NO_UNITS") != nullptr;                               // tuning only
      if (m[MF_NUNITS] > 1 && !no_units) {
        // heavy right-hand side: the ODE stage runs one wave per output unit (defect_units.h)
        const int nunits = int(m[MF_NUNITS]), gpmax = 64 / int(m[MF_CS]);
        // about one workgroup per SIMD: groups x units ~ 4 per CU
        int gp = (a.nseg * nunits + 4 * cus - 1) / (4 * cus);
        gp = gp < 1 ? 1 : (gp > gpmax ? gpmax : gp);
        const size_t bytes_units = size_t(m[MF_UNITS_BASE_BYTES]) + size_t(gp) * size_t(m[MF_UNITS_SLOT_BYTES]);
        // One launch (PHASE 4: every cardinal unit forms the interior gradients it needs itself) while its 2 x units x groups
        // workgroups are about one round of the device's SIMDs; two launches (interior units, then cardinal units) beyond
        static const int units1 = tuning_env("ASSET_HIP_UNITS_ONE_LAUNCH") ? std::atoi(std::getenv("ASSET_HIP_UNITS_ONE_LAUNCH")) : -1;   // tuning only
        int gp4 = (a.nseg * 2 * nunits + 4 * cus - 1) / (4 * cus);
        gp4 = gp4 < 1 ? 1 : (gp4 > gpmax ? gpmax : gp4);
        const long long wgs4 = (long long)((a.nseg + gp4 - 1) / gp4) * 2 * nunits;
        const bool one_launch = t.k[K_UNITS4] && (units1 >= 0 ? units1 != 0 : wgs4 <= (long long)(ASSET_UNITS_ONE_LAUNCH_ROUNDS * 4) * cus);
        if (one_launch) {
          // XCD-aware placement (MI355X: 8 XCDs; on a part with another count the mapping below is still a valid split of the work --
          // it only stops coinciding with the L2s): workgroups go to the eight XCDs round robin in launch order (x fastest), so with the number of
          // groups padded to a multiple of eight every unit of group g runs on XCD g % 8 -- its slot is assembled in ONE L2 (no
          // 32-byte sector written back half-filled by several of them) -- and the dense part reads it there (units_gp below)
          const size_t bytes4 = size_t(m[MF_UNITS_BASE_BYTES]) + size_t(gp4) * size_t(m[MF_UNITS_SLOT_BYTES]);
          void* uargs4[] = {&args, &gp4};
          const int ng4 = (a.nseg + gp4 - 1) / gp4, ng4p = (ng4 + 7) & ~7;
          if ((e = klaunch(t.k[K_UNITS4], dim3(ng4p, 2 * nunits), dim3(64), bytes4, st, uargs4)) != hipSuccess) return e;
          static const bool no_xcd = tuning_env("ASSET_HIP_NO_XCD_PLACEMENT") != nullptr;                       // tuning only
          // (the dense part follows only while its shares stay as even as the plain split's -- every group is divided among a whole
          //  number of waves: Betts-LGL5 x 1 000: 29.9 -> 28.6 us; x 2 000, where that leaves 3 segments to some waves and 2 to others:
          //  51.3 -> 55.7 us.  The padded unit grid alone: Betts-LGL5 x 5 000 106.7 -> 101.2 us, Betts-LGL7 x 5 000 145.4 -> 138.0 us)
          const int dwaves = cus * 4 * int(m[MF_RES_WPS]), gx = (ng4 + 7) / 8, wpg = gx > 0 ? (dwaves / 8) / gx : 0;
          if (!no_xcd && dwaves % 8 == 0 && wpg > 0 && (gp4 + wpg - 1) / wpg <= (a.nseg + dwaves - 1) / dwaves) args.units_gp = gp4;
        } else {
        const dim3 grid((a.nseg + gp - 1) / gp, nunits);
        void* uargs[] = {&args, &gp};
        if ((e = klaunch(t.k[K_UNITS0], grid, dim3(64), bytes_units, st, uargs)) != hipSuccess) return e;
        if ((e = klaunch(t.k[K_UNITS1], grid, dim3(64), bytes_units, st, uargs)) != hipSuccess) return e;
        }
      } else if ((e = ode_stage(2)) != hipSuccess) {
        return e;
      }
      if (skip_dense) return hipSuccess;
      static const bool no_resd = tuning_env("ASSET_HIP_NO_RESIDENT") != nullptr;                             // tuning only
      if (m[MF_RESD_GR] > 0 && !no_resd && a.lane_consts_res && t.k[K_RESD(a.kmap != nullptr)]) {
        const int waves = cus * 4 * int(m[MF_RES_WPS]);   // dense part of the resident kernel over the slots the units wrote
        // (XCD-aware placement: the whole grid, a wave's segments follow from its XCD; otherwise a wave per segment at most)
        const int nwg = args.units_gp > 0 ? waves : (a.nseg < waves ? a.nseg : waves);
        return klaunch(t.k[K_RESD(a.kmap != nullptr)], dim3(nwg), dim3(64), size_t(m[MF_RES_LDS_BYTES]), st, kargs);
      }
      return dense_stage(2);
    }
  }
  return hipErrorInvalidValue;
}

// A plain function batched over applications: transcription id 0 (func_kernels.h)
inline hipError_t launch_func_table(const KernelTable& t, int level, const EvalArgs& a, hipStream_t st) {
  if (level < 0 || level > 2) return hipErrorInvalidValue;
  EvalArgs args = a;
  void* kargs[] = {&args};
  const bool asmb = level >= 1 && a.kmap != nullptr;
  // block kinds: APW applications per workgroup, their blocks staged in LDS (func_kernels.h: FuncStage)
  const bool staged = level >= 1 && !asmb && t.meta[MF_G] > 0;
  const int apw = staged ? int(t.meta[MF_G]) : 64;
  const size_t shmem = staged ? size_t(t.meta[MF_LDS_BYTES]) : 0;
  return klaunch(t.k[K_FUNC(level, asmb)], dim3((a.nseg + apw - 1) / apw), dim3(64), shmem, st, kargs);
}

inline hipError_t entry_launch(const KernelEntry* ke, int level, const EvalArgs& a, int cus, hipStream_t st) {
  return ke->table->meta[MF_KIND] == 2 ? launch_func_table(*ke->table, level, a, st) : launch_lgl_table(*ke->table, level, a, cus, st);
}

// de Boor mesh-error estimate (mesh_kernels.h); only transcriptions of an ODE have it
inline bool entry_has_mesh(const KernelEntry* ke) { return bool(ke->table->k[K_MESH_YVEC]); }
inline hipError_t entry_mesh(const KernelEntry* ke, const MeshArgs& a, hipStream_t st) {
  const MeshScheme sc = mesh_scheme(ke->mode);
  MeshArgs args = a;
  const int grid = (a.nb + 63) / 64;
  void* yargs[] = {&args};
  hipError_t e = klaunch(ke->table->k[K_MESH_YVEC], dim3(grid), dim3(64), 0, st, yargs);
  if (e != hipSuccess) return e;
  int xv = ke->xv;
  double order = sc.order, weight = sc.error_weight;
  void* eargs[] = {&args, &xv, &order, &weight};
  return klaunch(ke->table->k[K_MESH_ERROR], dim3(grid), dim3(64), 0, st, eargs);
}

// per-lane constants of the dense stage for derivative level 1 / 2: table size in bytes (0: none) and the kernel filling it
// (level 0 stands for the record of the resident kernel)
inline size_t entry_lane_bytes(const KernelEntry* ke, int level) {
  if (ke->table->meta[MF_KIND] != 1) return 0;
  return level >= 2 ? size_t(ke->table->meta[MF_LANE_BYTES2])
                    : (level == 1 ? size_t(ke->table->meta[MF_LANE_BYTES1]) : size_t(ke->table->meta[MF_LANE_BYTES_RES]));
}
inline hipError_t entry_lane_setup(const KernelEntry* ke, int level, void* out, hipStream_t st) {
  void* kargs[] = {&out};
  const KernelTable& t = *ke->table;
  if (level == 0) return klaunch(t.k[K_RES_SETUP], dim3(1), dim3(64), 0, st, kargs);
  return klaunch(t.meta[MF_WIDE] ? t.k[K_WIDE_SETUP] : (level >= 2 ? t.k[K_LANE_SETUP2] : t.k[K_LANE_SETUP1]), dim3(1), dim3(64), 0, st, kargs);
}

// ---- tables of the kernels linked into this translation unit ---------------------------------------------------------
#define ASSET_KPTR(...) reinterpret_cast<const void*>(&__VA_ARGS__)
template <class Ode, int SCH, bool BLOCKED, int G>
const KernelTable* lgl_static_table() {
  using D = Dims<Ode, SCH, BLOCKED>;
  static KernelTable t = [] {
    KernelTable r;
    for (int i = 0; i < MF_COUNT; i++) r.meta[i] = LglMeta<Ode, SCH, BLOCKED, G>::v[i];
    r.k[K_LGL(0, 1, false)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 0, 1, false>);
    r.k[K_LGL(1, 1, false)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 1, 1, false>);
    r.k[K_LGL(2, 1, false)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 2, 1, false>);
    if constexpr (D::WIDE) {
      r.k[K_WIDE(1, false)].host = ASSET_KPTR(lgl_wide_dense_kernel<Ode, SCH, BLOCKED, 1, false>);
      r.k[K_WIDE(1, true)].host = ASSET_KPTR(lgl_wide_dense_kernel<Ode, SCH, BLOCKED, 1, true>);
      r.k[K_WIDE(2, false)].host = ASSET_KPTR(lgl_wide_dense_kernel<Ode, SCH, BLOCKED, 2, false>);
      r.k[K_WIDE(2, true)].host = ASSET_KPTR(lgl_wide_dense_kernel<Ode, SCH, BLOCKED, 2, true>);
      r.k[K_WIDE_SETUP].host = ASSET_KPTR(wide_setup_kernel<Ode, SCH, BLOCKED>);
      if constexpr (RowsDims<D>::OK) {
        r.k[K_ROWS].host = ASSET_KPTR(lgl_rows_kernel<Ode, SCH, BLOCKED, 2>);
        r.k[K_ROWS1].host = ASSET_KPTR(lgl_rows_kernel<Ode, SCH, BLOCKED, 1>);
      }
    } else {
      r.k[K_LGL(1, 2, false)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 1, 2, false>);
      r.k[K_LGL(1, 2, true)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 1, 2, true>);
      r.k[K_LGL(2, 2, false)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 2, 2, false>);
      r.k[K_LGL(2, 2, true)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 2, 2, true>);
      r.k[K_LANE_SETUP1].host = ASSET_KPTR(lane_setup_kernel<Ode, SCH, BLOCKED, 1>);
      r.k[K_LANE_SETUP2].host = ASSET_KPTR(lane_setup_kernel<Ode, SCH, BLOCKED, 2>);
      if constexpr (D::FUSED) {
        r.k[K_LGL(2, 3, false)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 2, 3, false>);
        r.k[K_LGL(2, 3, true)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 2, 3, true>);
        r.k[K_LGL(1, 3, false)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 1, 3, false>);
        r.k[K_LGL(1, 3, true)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 1, 3, true>);
      }
      if constexpr (D::FUSED2) r.k[K_LGL(2, 4, false)].host = ASSET_KPTR(lgl_defect_kernel<Ode, SCH, BLOCKED, G, 2, 4, false>);
      if constexpr (ResDims<D>::OK) {
        r.k[K_RES(false)].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 2, false>);
        r.k[K_RES(true)].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 2, true>);
        r.k[K_RES1(false)].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 1, false>);
        r.k[K_RES1(true)].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 1, true>);
        r.k[K_RESL(2, false)].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 2, false, true>);
        r.k[K_RESL(2, true)].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 2, true, true>);
        r.k[K_RESL(1, false)].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 1, false, true>);
        r.k[K_RESL(1, true)].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 1, true, true>);
        r.k[K_RES_SETUP].host = ASSET_KPTR(res_lane_setup_kernel<Ode, SCH, BLOCKED>);
        if constexpr (ResDims<D>::LOOP_PAIR) r.k[K_RESLP].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 2, false, true, false, true>);
        if constexpr (ResDims<D>::RD_ALT) r.k[K_RES_ALT].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 2, false, false, false, false, 1>);
      }
      if constexpr (ResDims<D>::GIVEN_OK) {
        r.k[K_RESD(false)].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 2, false, true, true>);
        r.k[K_RESD(true)].host = ASSET_KPTR(lgl_resident_kernel<Ode, SCH, BLOCKED, 2, true, true, true>);
        r.k[K_RES_SETUP].host = ASSET_KPTR(res_lane_setup_kernel<Ode, SCH, BLOCKED>);
      }
    }
    if constexpr (Ode::NUNITS > 1) {
      r.k[K_UNITS0].host = ASSET_KPTR(lgl_ode_units_kernel<Ode, SCH, BLOCKED, 0>);
      r.k[K_UNITS1].host = ASSET_KPTR(lgl_ode_units_kernel<Ode, SCH, BLOCKED, 1>);
      r.k[K_UNITSJ].host = ASSET_KPTR(lgl_ode_units_kernel<Ode, SCH, BLOCKED, 3>);
      r.k[K_UNITS4].host = ASSET_KPTR(lgl_ode_units_kernel<Ode, SCH, BLOCKED, 4>);
    }
    r.k[K_ADJGRAD].host = ASSET_KPTR(lgl_adjgrad_kernel<Ode, SCH, BLOCKED, true>);
    r.k[K_VALUE].host = ASSET_KPTR(lgl_adjgrad_kernel<Ode, SCH, BLOCKED, false>);
    r.k[K_MESH_YVEC].host = ASSET_KPTR(mesh_yvec_kernel<Ode, SCH, BLOCKED>);
    r.k[K_MESH_ERROR].host = ASSET_KPTR(mesh_error_kernel<0>);
    return r;
  }();
  return &t;
}
template <class F>
const KernelTable* func_static_table() {
  static KernelTable t = [] {
    KernelTable r;
    for (int i = 0; i < MF_COUNT; i++) r.meta[i] = FuncMeta<F>::v[i];
    r.k[K_FUNC(0, false)].host = ASSET_KPTR(func_kernel<F, 0, false>);
    r.k[K_FUNC(1, false)].host = ASSET_KPTR(func_kernel<F, 1, false>);
    r.k[K_FUNC(1, true)].host = ASSET_KPTR(func_kernel<F, 1, true>);
    r.k[K_FUNC(2, false)].host = ASSET_KPTR(func_kernel<F, 2, false>);
    r.k[K_FUNC(2, true)].host = ASSET_KPTR(func_kernel<F, 2, true>);
    return r;
  }();
  return &t;
}
#undef ASSET_KPTR

struct StaticEntry {   // (static initialisation: the table is filled and the entry registered before main)
  KernelEntry e;
  Registrar* reg;
  StaticEntry(const char* name, const KernelTable* t) {
    entry_from_table(e, name, t);
    reg = new Registrar(&e);
  }
};

#define ASSET_REGISTER_FUNC(FN) \
  static ::asset_hip::StaticEntry entry_##FN##_func(FN::name(), ::asset_hip::func_static_table<FN>());

// Trapezoidal = transcription id 1 of the same kernels (defect_dims.h: Dims::TRAP)
#define ASSET_REGISTER_TRAP(ODE, BLK, G) ASSET_REGISTER_LGL(ODE, 1, BLK, G)

#define ASSET_REGISTER_LGL(ODE, CSV, BLK, G) \
  static ::asset_hip::StaticEntry entry_##ODE##_##CSV##_##BLK(ODE::name(), ::asset_hip::lgl_static_table<ODE, CSV, (BLK != 0), G>());

// ---- name expressions of the kernels of a run-time module (capi.hip: asset_hip_jit_plugin) -----------------------------
// kind 1: `type` is the ODE functor, kind 2: the function functor, kind 3: the functor list of a bundle.  Slots without a kernel for that kind: empty string.
inline std::string rtc_kernel_expr(int slot, int kind, const std::string& type, int csv, bool blocked, int g) {
  const std::string b = blocked ? "true" : "false";
  const std::string lgl = type + ", " + std::to_string(csv) + ", " + b;
  auto tf = [](bool v) { return std::string(v ? "true" : "false"); };
  if (kind == 3) {   // a bundle: `type` is the comma-separated functor list
    for (int lv = 0; lv <= 2; lv++)
      if (slot == K_BUNDLE(lv)) return "asset_hip::func_bundle_kernel<" + std::to_string(lv) + ", " + type + ">";
    return "";
  }
  if (kind == 2) {
    for (int lv = 0; lv <= 2; lv++)
      for (int as = 0; as <= (lv >= 1 ? 1 : 0); as++)
        if (slot == K_FUNC(lv, as != 0)) return "asset_hip::func_kernel<" + type + ", " + std::to_string(lv) + ", " + tf(as != 0) + ">";
    return "";
  }
  for (int lv = 0; lv <= 2; lv++)
    for (int stg = 1; stg <= 4; stg++)
      for (int as = 0; as <= 1; as++) {
        if (slot != K_LGL(lv, stg, as != 0)) continue;
        const bool used = (stg == 1 && !as) || (stg == 2 && lv >= 1) || (stg == 3 && lv >= 1) || (stg == 4 && lv == 2 && !as);
        if (!used) return "";
        return "asset_hip::lgl_defect_kernel<" + lgl + ", " + std::to_string(g) + ", " + std::to_string(lv) + ", " +
               std::to_string(stg) + ", " + tf(as != 0) + ">";
      }
  for (int lv = 1; lv <= 2; lv++)
    for (int as = 0; as <= 1; as++)
      if (slot == K_WIDE(lv, as != 0)) return "asset_hip::lgl_wide_dense_kernel<" + lgl + ", " + std::to_string(lv) + ", " + tf(as != 0) + ">";
  if (slot == K_WIDE_SETUP) return "asset_hip::wide_setup_kernel<" + lgl + ">";
  if (slot == K_ROWS) return "asset_hip::lgl_rows_kernel<" + lgl + ", 2>";
  if (slot == K_ROWS1) return "asset_hip::lgl_rows_kernel<" + lgl + ", 1>";
  if (slot == K_RES(false)) return "asset_hip::lgl_resident_kernel<" + lgl + ", 2, false>";
  if (slot == K_RES(true)) return "asset_hip::lgl_resident_kernel<" + lgl + ", 2, true>";
  if (slot == K_RES1(false)) return "asset_hip::lgl_resident_kernel<" + lgl + ", 1, false>";
  if (slot == K_RES1(true)) return "asset_hip::lgl_resident_kernel<" + lgl + ", 1, true>";
  for (int lv = 1; lv <= 2; lv++)
    for (int as = 0; as <= 1; as++)
      if (slot == K_RESL(lv, as != 0)) return "asset_hip::lgl_resident_kernel<" + lgl + ", " + std::to_string(lv) + ", " + tf(as != 0) + ", true>";
  if (slot == K_RESLP) return "asset_hip::lgl_resident_kernel<" + lgl + ", 2, false, true, false, true>";
  if (slot == K_RES_ALT) return "asset_hip::lgl_resident_kernel<" + lgl + ", 2, false, false, false, false, 1>";
  if (slot == K_RESD(false)) return "asset_hip::lgl_resident_kernel<" + lgl + ", 2, false, true, true>";
  if (slot == K_RESD(true)) return "asset_hip::lgl_resident_kernel<" + lgl + ", 2, true, true, true>";
  if (slot == K_RES_SETUP) return "asset_hip::res_lane_setup_kernel<" + lgl + ">";
  if (slot == K_LANE_SETUP1) return "asset_hip::lane_setup_kernel<" + lgl + ", 1>";
  if (slot == K_LANE_SETUP2) return "asset_hip::lane_setup_kernel<" + lgl + ", 2>";
  if (slot == K_UNITS0) return "asset_hip::lgl_ode_units_kernel<" + lgl + ", 0>";
  if (slot == K_UNITS1) return "asset_hip::lgl_ode_units_kernel<" + lgl + ", 1>";
  if (slot == K_UNITSJ) return "asset_hip::lgl_ode_units_kernel<" + lgl + ", 3>";
  if (slot == K_UNITS4) return "asset_hip::lgl_ode_units_kernel<" + lgl + ", 4>";
  if (slot == K_ADJGRAD) return "asset_hip::lgl_adjgrad_kernel<" + lgl + ", true>";
  if (slot == K_VALUE) return "asset_hip::lgl_adjgrad_kernel<" + lgl + ", false>";
  if (slot == K_MESH_YVEC) return "asset_hip::mesh_yvec_kernel<" + lgl + ">";
  if (slot == K_MESH_ERROR) return "asset_hip::mesh_error_kernel<0>";
  return "";
}

}  // namespace asset_hip
