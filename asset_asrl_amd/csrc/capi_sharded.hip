// One constraint, several device handles IN ONE PROCESS: the C-ABI form of the reference's thread_split
// (/root/reference/src/Solvers/ConstraintFunction.h:55-62 -> VectorFunctions/IndexingData.h:117-146: contiguous chunks of the
// function applications, the first `nappl % n` one longer; placement Solvers/NonLinearProgram.cpp:71-109; all shards launched
// before any is waited for, :519-526).  Where the reference hands every chunk to a CPU thread, every chunk here is a handle of
// its own on a device of the caller's choice, with its own stream: inputs go to every device over that device's PCIe link, all
// shards are enqueued, and every shard's blocks (or assembled values) come back over its own link into the caller's arrays.
// Built on the public C ABI (include/asset_hip.h) and the HIP runtime alone: no collective library, no shared memory segment,
// no Python.  The same device may be named several times (N handles on one GPU: how the form is tested on a one-GPU box).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/asset_hip.h"

extern "C" __attribute__((visibility("hidden"))) void asset_hip_set_last_error(const char* msg);   // capi.hip

namespace {
struct Shard {
  int device = 0, first = 0, count = 0;
  asset_hip_defect_t h = nullptr;
  hipStream_t stream = nullptr;
  double *dX = nullptr, *dL = nullptr, *dfx = nullptr, *dagx = nullptr, *dkkt = nullptr;
  // assembled form: this shard's range [lo, hi) of the solver's value array, on the device and in page-locked host memory
  long long lo = 0, hi = 0;
  double* dvals = nullptr;
  double* hvals = nullptr;
};
int sfail(int rc, const std::string& msg) {
  asset_hip_set_last_error(msg.c_str());
  return rc;
}
int hfail(hipError_t e, const char* where) {
  return sfail(int(e), std::string(where) + ": " + hipGetErrorString(e));
}
// Every entry point below walks the shards' devices (hipSetDevice is per thread): the calling thread's current device is put back on
// every exit path, so that a caller who also uses torch or another HIP client on this thread does not find its later allocations and
// launches on the last shard's GPU.
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
  ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};
// Is the range a device-to-host copy may land in page-locked (hipHostMalloc / asset_hip_host_register)?  Into pageable memory
// hipMemcpyAsync is synchronous for the calling thread.
bool page_locked(const void* p) {
  if (!p) return true;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}
}  // namespace

struct asset_hip_sharded {
  std::vector<Shard> shards;
  int ir = 0, orr = 0, nkkt = 0, nseg = 0, n_primal = 0, n_equal = 0;
  int kstride = 0;                       // doubles per KKT block in the handles' layout (asset_hip_defect_kkt_layout)
  long long nvalues = 0;
};

extern "C" {

void asset_hip_sharded_destroy(asset_hip_sharded_t s) {
  if (!s) return;
  DeviceGuard guard;
  for (Shard& sh : s->shards) {
    if (sh.stream || sh.dX) (void)hipSetDevice(sh.device);
    if (sh.stream) (void)hipStreamSynchronize(sh.stream);
    for (double* p : {sh.dX, sh.dL, sh.dfx, sh.dagx, sh.dkkt, sh.dvals})
      if (p) (void)hipFree(p);
    if (sh.hvals) (void)hipHostFree(sh.hvals);
    if (sh.stream) (void)hipStreamDestroy(sh.stream);
    if (sh.h) asset_hip_defect_destroy(sh.h);
  }
  delete s;
}

int asset_hip_defect_create_sharded(const asset_hip_defect_desc* d, int nshards, const int* devices, asset_hip_sharded_t* out) {
  if (!d || !out || !devices || nshards <= 0) return sfail(ASSET_HIP_EINVAL, "create_sharded: null descriptor / devices / output, or no shards");
  *out = nullptr;
  if (!d->vindex || !d->cindex || d->nseg <= 0) return sfail(ASSET_HIP_EINVAL, "create_sharded: descriptor fields missing or non-positive");
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      (void)hipGetLastError();
      return sfail(ASSET_HIP_ENODEV, "no HIP device visible: the evaluator has no CPU fallback");
    }
    for (int i = 0; i < nshards; i++)
      if (devices[i] < 0 || devices[i] >= ndev) return sfail(ASSET_HIP_EINVAL, "create_sharded: device ordinal out of range");
  }
  DeviceGuard guard;
  asset_hip_sharded* s = new (std::nothrow) asset_hip_sharded;
  if (!s) return sfail(ASSET_HIP_EINVAL, "out of host memory");
  s->nseg = d->nseg, s->n_primal = d->n_primal, s->n_equal = d->n_equal;
  // IndexingData.h:117-146: cols / Threads each, the first cols % Threads one more; fewer shards than asked for when there are
  // fewer applications than shards
  const int per = d->nseg / nshards, rem = d->nseg % nshards, range = per > 0 ? nshards : rem;
  int start = 0;
  for (int i = 0; i < range; i++) {
    Shard sh;
    sh.device = devices[i], sh.first = start, sh.count = per + (i < rem ? 1 : 0);
    start += sh.count;
    s->shards.push_back(sh);
  }
  for (size_t i = 0; i < s->shards.size(); i++) {
    Shard& sh = s->shards[i];
    asset_hip_defect_desc sd = *d;
    sd.device = sh.device;
    sd.nseg = sh.count;
    int rc = 0;
    if (i == 0) {                                       // sizes from the first handle (the tables of the others start IR / OR rows further on)
      rc = asset_hip_defect_create(&sd, &sh.h);
      if (rc == 0) rc = asset_hip_defect_sizes(sh.h, &s->ir, &s->orr, &s->nkkt);
      if (rc == 0 && asset_hip_defect_kkt_layout(sh.h, &s->kstride, nullptr, nullptr) < 0) rc = ASSET_HIP_EINVAL;
    } else {
      sd.vindex = d->vindex + size_t(sh.first) * s->ir;
      sd.cindex = d->cindex + size_t(sh.first) * s->orr;
      rc = asset_hip_defect_create(&sd, &sh.h);
    }
    if (rc) { asset_hip_sharded_destroy(s); return rc; }
    hipError_t e;
    auto bail = [&](hipError_t err, const char* w) { int r = hfail(err, w); asset_hip_sharded_destroy(s); return r; };
    if ((e = hipSetDevice(sh.device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipMalloc(&sh.dX, size_t(d->n_primal) * 8)) != hipSuccess) return bail(e, "hipMalloc(X)");
    if ((e = hipMalloc(&sh.dL, size_t(d->n_equal) * 8)) != hipSuccess) return bail(e, "hipMalloc(L)");
    if ((e = hipMalloc(&sh.dfx, size_t(sh.count) * s->orr * 8)) != hipSuccess) return bail(e, "hipMalloc(FX blocks)");
    if ((e = hipMalloc(&sh.dagx, size_t(sh.count) * s->ir * 8)) != hipSuccess) return bail(e, "hipMalloc(AGX blocks)");
    if ((e = hipMalloc(&sh.dkkt, size_t(sh.count) * s->kstride * 8)) != hipSuccess) return bail(e, "hipMalloc(KKT blocks)");
  }
  *out = s;
  return 0;
}

int asset_hip_sharded_shards(asset_hip_sharded_t s) { return s ? int(s->shards.size()) : 0; }

int asset_hip_sharded_range(asset_hip_sharded_t s, int shard, int* first, int* count, int* device) {
  if (!s || shard < 0 || shard >= int(s->shards.size())) return sfail(ASSET_HIP_EINVAL, "sharded_range: no such shard");
  if (first) *first = s->shards[shard].first;
  if (count) *count = s->shards[shard].count;
  if (device) *device = s->shards[shard].device;
  return 0;
}

asset_hip_defect_t asset_hip_sharded_handle(asset_hip_sharded_t s, int shard) {
  return (s && shard >= 0 && shard < int(s->shards.size())) ? s->shards[shard].h : nullptr;
}

}  // extern "C"

// inputs to every shard's device, on its stream
static int upload(asset_hip_sharded_t s, Shard& sh, const double* X, const double* L) {
  hipError_t e;
  if ((e = hipSetDevice(sh.device)) != hipSuccess) return hfail(e, "hipSetDevice");
  if ((e = hipMemcpyAsync(sh.dX, X, size_t(s->n_primal) * 8, hipMemcpyHostToDevice, sh.stream)) != hipSuccess) return hfail(e, "hipMemcpyAsync(X)");
  if (L && (e = hipMemcpyAsync(sh.dL, L, size_t(s->n_equal) * 8, hipMemcpyHostToDevice, sh.stream)) != hipSuccess) return hfail(e, "hipMemcpyAsync(L)");
  return 0;
}
static int drain(asset_hip_sharded_t s) {
  int rc = 0;
  for (Shard& sh : s->shards) {
    hipError_t e = hipSetDevice(sh.device);
    if (e == hipSuccess) e = hipStreamSynchronize(sh.stream);
    if (e != hipSuccess && rc == 0) rc = hfail(e, "hipStreamSynchronize(shard)");
  }
  return rc;
}

// One shard's share of an evaluation on ITS stream: inputs in, launch, blocks out into the caller's arrays.  `wait`: also wait for it
// (the per-shard host threads of the pageable path).
static int shard_eval(asset_hip_sharded_t s, Shard& sh, int what, const double* X, const double* L, double* fx_blocks, double* agx_blocks,
                      double* kkt_blocks, bool want_agx, bool want_kkt, bool wait) {
  int rc = upload(s, sh, X, L);
  if (rc) return rc;
  rc = asset_hip_defect_eval_device(sh.h, what, sh.dX, L ? sh.dL : nullptr, fx_blocks ? sh.dfx : nullptr, want_agx ? sh.dagx : nullptr,
                                    want_kkt ? sh.dkkt : nullptr, sh.stream);
  if (rc) return rc;
  hipError_t e = hipSuccess;
  if (fx_blocks) e = hipMemcpyAsync(fx_blocks + size_t(sh.first) * s->orr, sh.dfx, size_t(sh.count) * s->orr * 8, hipMemcpyDeviceToHost, sh.stream);
  if (e == hipSuccess && want_agx)
    e = hipMemcpyAsync(agx_blocks + size_t(sh.first) * s->ir, sh.dagx, size_t(sh.count) * s->ir * 8, hipMemcpyDeviceToHost, sh.stream);
  if (e == hipSuccess && want_kkt)
    e = hipMemcpyAsync(kkt_blocks + size_t(sh.first) * s->kstride, sh.dkkt, size_t(sh.count) * s->kstride * 8, hipMemcpyDeviceToHost, sh.stream);
  if (e == hipSuccess && wait) e = hipStreamSynchronize(sh.stream);
  return e == hipSuccess ? 0 : hfail(e, "hipMemcpyAsync(blocks)");
}

// Every shard through `body(shard)`: on the calling thread, one after the other, when `async` (nothing in body blocks: page-locked
// targets) -- or one host thread per shard, as the reference launches its thread-split functions (NonLinearProgram.cpp:519-526),
// when a copy of body would block the thread that issues it (pageable targets: the driver stages them synchronously) and the
// next shard's launch would wait for this shard's blocks to cross PCIe.  The first error (rc and text) is the call's.
template <class Body>
static int for_shards(asset_hip_sharded_t s, bool async, Body body) {
  const size_t n = s->shards.size();
  if (async || n == 1) {
    int rc = 0;
    for (size_t i = 0; i < n && rc == 0; i++) rc = body(s->shards[i]);
    return rc;
  }
  std::vector<int> rcs(n, 0);
  std::vector<std::string> msgs(n);
  std::vector<std::thread> th;
  th.reserve(n);
  for (size_t i = 0; i < n; i++)
    th.emplace_back([&, i] {
      rcs[i] = body(s->shards[i]);
      if (rcs[i]) msgs[i] = asset_hip_last_error();      // (the error text is per thread)
    });
  for (auto& t : th) t.join();
  for (size_t i = 0; i < n; i++)
    if (rcs[i]) return sfail(rcs[i], msgs[i]);
  return 0;
}

extern "C" {

int asset_hip_sharded_eval(asset_hip_sharded_t s, int what, const double* X, const double* L, double* fx_blocks, double* agx_blocks,
                           double* kkt_blocks) {
  if (!s || !X) return sfail(ASSET_HIP_EINVAL, "sharded_eval: null handle / X");
  DeviceGuard guard;
  const int kind = what & 0xff;
  const bool want_agx = agx_blocks && (kind == ASSET_HIP_CON_ADJGRAD || kind == ASSET_HIP_JAC_ADJGRAD || kind == ASSET_HIP_JAC_ADJGRAD_HESS);
  const bool want_kkt = kkt_blocks && kind >= ASSET_HIP_JAC;
  // every shard enqueued before any is waited for (NonLinearProgram.cpp:519-526) -- which a single enqueueing thread can only do
  // when no copy blocks it
  // (X / L: a pageable source costs the issuing thread the staging of 8 (n_primal + n_equal) bytes per shard, ahead of the shard's
  //  launch -- not a wait for another shard's blocks)
  const bool async = page_locked(fx_blocks) && (!want_agx || page_locked(agx_blocks)) && (!want_kkt || page_locked(kkt_blocks));
  const int rc = for_shards(s, async, [&](Shard& sh) {
    return shard_eval(s, sh, what, X, L, fx_blocks, agx_blocks, kkt_blocks, want_agx, want_kkt, !async);
  });
  const int rd = drain(s);                  // (also after an error: nothing of this call stays in flight)
  return rc ? rc : rd;
}

int asset_hip_sharded_set_kkt_map(asset_hip_sharded_t s, const int32_t* slot_locations, long long nvalues) {
  if (!s || !slot_locations || nvalues <= 0) return sfail(ASSET_HIP_EINVAL, "sharded_set_kkt_map: null handle / map, or no values");
  DeviceGuard guard;
  // Failure-atomic towards the evaluation: no map is in effect (nvalues = 0, asset_hip_sharded_eval_assembled refuses) from here until
  // EVERY shard has its new map and its new buffers; the buffers are built into locals and committed together.
  s->nvalues = 0;
  const size_t n = s->shards.size();
  struct Fresh { long long lo = 0, hi = 0; double* d = nullptr; double* h = nullptr; };
  std::vector<Fresh> fresh(n);
  auto undo = [&](int rc) {
    for (size_t i = 0; i < n; i++) {
      if (fresh[i].d || fresh[i].h) (void)hipSetDevice(s->shards[i].device);
      if (fresh[i].d) (void)hipFree(fresh[i].d);
      if (fresh[i].h) (void)hipHostFree(fresh[i].h);
    }
    return rc;
  };
  std::vector<int32_t> local;
  for (size_t i = 0; i < n; i++) {
    Shard& sh = s->shards[i];
    const int32_t* m = slot_locations + size_t(sh.first) * s->nkkt;    // (the map is given in the canonical slot order: include/asset_hip.h)
    const size_t len = size_t(sh.count) * s->nkkt;
    long long lo = nvalues, hi = 0;
    for (size_t k = 0; k < len; k++)
      if (m[k] >= 0) { lo = std::min<long long>(lo, m[k]); hi = std::max<long long>(hi, (long long)m[k] + 1); }
    if (hi <= lo) lo = 0, hi = 1;
    if (hi > nvalues) return undo(sfail(ASSET_HIP_ERANGE, "sharded_set_kkt_map: a slot location is outside the value array"));
    local.resize(len);
    for (size_t k = 0; k < len; k++) local[k] = m[k] >= 0 ? int32_t(m[k] - lo) : -1;   // the shard's own array starts at its lowest location
    const int rc = asset_hip_defect_set_kkt_map(sh.h, local.data(), hi - lo, 0);
    if (rc) return undo(rc);
    hipError_t e;
    if ((e = hipSetDevice(sh.device)) != hipSuccess) return undo(hfail(e, "hipSetDevice"));
    if ((e = hipMalloc(&fresh[i].d, size_t(hi - lo) * 8)) != hipSuccess) return undo(hfail(e, "hipMalloc(shard values)"));
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&fresh[i].h), size_t(hi - lo) * 8, hipHostMallocDefault)) != hipSuccess)
      return undo(hfail(e, "hipHostMalloc(shard values)"));
    fresh[i].lo = lo, fresh[i].hi = hi;
  }
  for (size_t i = 0; i < n; i++) {
    Shard& sh = s->shards[i];
    if (sh.dvals || sh.hvals) (void)hipSetDevice(sh.device);
    if (sh.dvals) (void)hipFree(sh.dvals);
    if (sh.hvals) (void)hipHostFree(sh.hvals);
    sh.dvals = fresh[i].d, sh.hvals = fresh[i].h, sh.lo = fresh[i].lo, sh.hi = fresh[i].hi;
  }
  s->nvalues = nvalues;
  return 0;
}

int asset_hip_sharded_eval_assembled(asset_hip_sharded_t s, int what, const double* X, const double* L, double* fx_blocks,
                                     double* agx_blocks, double* kkt_values) {
  if (!s || !X || !kkt_values) return sfail(ASSET_HIP_EINVAL, "sharded_eval_assembled: null handle / X / values");
  if (s->nvalues <= 0) return sfail(ASSET_HIP_EINVAL, "sharded_eval_assembled: no KKT map (asset_hip_sharded_set_kkt_map)");
  const int kind = what & 0xff;
  if (kind < ASSET_HIP_JAC) return sfail(ASSET_HIP_EINVAL, "sharded_eval_assembled: a Jacobian kind is expected");
  DeviceGuard guard;
  const bool want_agx = agx_blocks && (kind == ASSET_HIP_JAC_ADJGRAD || kind == ASSET_HIP_JAC_ADJGRAD_HESS);
  // (the values always land in the shards' own page-locked staging; the FX / AGX blocks in the caller's arrays)
  const bool async = page_locked(fx_blocks) && (!want_agx || page_locked(agx_blocks));
  const int rc = for_shards(s, async, [&](Shard& sh) {
    int r = upload(s, sh, X, L);
    if (r) return r;
    hipError_t e = hipMemsetAsync(sh.dvals, 0, size_t(sh.hi - sh.lo) * 8, sh.stream);
    if (e != hipSuccess) return hfail(e, "hipMemsetAsync(shard values)");
    r = asset_hip_defect_eval_assembled_device(sh.h, what, sh.dX, L ? sh.dL : nullptr, fx_blocks ? sh.dfx : nullptr,
                                               want_agx ? sh.dagx : nullptr, sh.dvals, sh.stream);
    if (r) return r;
    e = hipMemcpyAsync(sh.hvals, sh.dvals, size_t(sh.hi - sh.lo) * 8, hipMemcpyDeviceToHost, sh.stream);
    if (e == hipSuccess && fx_blocks)
      e = hipMemcpyAsync(fx_blocks + size_t(sh.first) * s->orr, sh.dfx, size_t(sh.count) * s->orr * 8, hipMemcpyDeviceToHost, sh.stream);
    if (e == hipSuccess && want_agx)
      e = hipMemcpyAsync(agx_blocks + size_t(sh.first) * s->ir, sh.dagx, size_t(sh.count) * s->ir * 8, hipMemcpyDeviceToHost, sh.stream);
    if (e == hipSuccess && !async) e = hipStreamSynchronize(sh.stream);
    return e == hipSuccess ? 0 : hfail(e, "hipMemcpyAsync(shard values / blocks)");
  });
  const int rd = drain(s);
  if (rc || rd) return rc ? rc : rd;
  // Every shard's range into the caller's array, in shard order: locations two neighbouring shards share (the node between their
  // segments) receive a + b exactly as the single handle's atomic pair does -- bitwise the same values for a phase without
  // parameters; entries between phase parameters are partial sums per shard here, one running sum there (same to rounding).
  // In parallel by DESTINATION: the touched part of the value array is cut into contiguous pieces, a host thread per piece adds
  // every shard's overlap with its piece in shard order -- no two threads write one location, and every location sees its
  // contributions in the order of the serial loop.
  long long lo = s->nvalues, hi = 0;
  for (Shard& sh : s->shards) lo = std::min(lo, sh.lo), hi = std::max(hi, sh.hi);
  const long long total = hi > lo ? hi - lo : 0;
  unsigned hw = std::thread::hardware_concurrency();
  const int nt = int(std::max<long long>(1, std::min<long long>({(long long)(hw ? hw : 1), 16LL, total / (1 << 18)})));   // (>= 2 MiB of values per thread)
  auto add_piece = [&](long long a, long long b) {
    for (Shard& sh : s->shards) {
      const long long x = std::max(a, sh.lo), y = std::min(b, sh.hi);
      double* dst = kkt_values + x;
      const double* src = sh.hvals + (x - sh.lo);
      for (long long k = 0; k < y - x; k++) dst[k] += src[k];
    }
  };
  if (nt <= 1) add_piece(lo, hi);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back(add_piece, lo + total * t / nt, lo + total * (t + 1) / nt);
    for (auto& t : th) t.join();
  }
  return 0;
}

}  // extern "C"
