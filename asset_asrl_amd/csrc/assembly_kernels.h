// On-device KKT assembly: the per-application blocks are added into the solver's CSR
// value array on the GPU, so the host receives nnz values (shared entries already summed) instead of nseg*NKKT block
// slots and performs no indexed += at all.
//
// What it replaces, relative to /root/reference/src:
//   KKTFillAll / KKTFillJac      VectorFunctions/DenseFunctionBase.h:1413-1523   mpt[KKTLocations[freeloc]] += value
//   KKTLocations                 Solvers/NonLinearProgram.cpp:316-330            offset of a triplet in valuePtr()
//   KKTClashes / KKTLocks        Solvers/NonLinearProgram.cpp:86-104,222-260     which columns several threads share
// The reference serialises clashing columns with mutexes; here a slot whose value location is shared with another
// slot of this constraint (boundary nodes of adjacent segments, phase parameters) is flagged once at map-upload time
// and added with a hardware f64 atomic, every other slot with a plain read-add-write.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace asset_hip {

// map[i] >= 0: unshared value location; map[i] <= -2: shared location -(map[i]+2); map[i] == -1: slot not scattered
__global__ __launch_bounds__(256) void kkt_scatter_kernel(const double* __restrict__ blocks,
                                                          const int32_t* __restrict__ map, size_t nslots,
                                                          double* values) {
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  // four independent slots per trip: the unshared ones are a read-add-write whose latency would otherwise serialise
  for (; i + 3 * stride < nslots; i += 4 * stride) {
    int32_t m[4];
    double v[4], old[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { m[u] = map[i + u * stride]; v[u] = blocks[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < 4; u++) old[u] = (m[u] >= 0) ? values[m[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (m[u] >= 0) values[m[u]] = old[u] + v[u];
      else if (m[u] != -1) unsafeAtomicAdd(values + (-(m[u] + 2)), v[u]);   // global_atomic_add_f64
    }
  }
  for (; i < nslots; i += stride) {
    const int32_t mm = map[i];
    if (mm == -1) continue;
    const double vv = blocks[i];
    if (mm >= 0) values[mm] += vv;
    else unsafeAtomicAdd(values + (-(mm + 2)), vv);
  }
}

}  // namespace asset_hip
