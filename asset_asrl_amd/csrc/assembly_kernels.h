// On-device KKT assembly: the per-application blocks are added into the solver's CSR
// value array on the GPU, so the host receives nnz values (shared entries already summed) instead of nseg*NKKT block
// slots and performs no indexed += at all.
//
// What it replaces, relative to /root/reference/src:
//   KKTFillAll / KKTFillJac      VectorFunctions/DenseFunctionBase.h:1413-1523   mpt[KKTLocations[freeloc]] += value
//   KKTLocations                 Solvers/NonLinearProgram.cpp:316-330            offset of a triplet in valuePtr()
//   KKTClashes / KKTLocks        Solvers/NonLinearProgram.cpp:86-104,222-260     which columns several threads share
// The reference serialises clashing columns with mutexes; here value locations shared by several slots of this
// constraint (boundary nodes of adjacent segments, phase parameters) are handled by hardware f64 atomics.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace asset_hip {

// Map encoding (built by asset_hip_defect_set_kkt_map): m >= 0 -- value location used by this slot alone: the entry
// is STORED (the caller hands over zeros there, as the reference zeroes the KKT values before every evaluation,
// PSIOPT.cpp:107); m <= -2 -- location -(m+2) is shared with other slots of this constraint: no-return f64 atomic add.
// In accumulate mode every slot is encoded as shared, which makes the whole evaluation a true += at about 1.4e11
// atomics/s.
//
// kkt_scatter_kernel serves the kernels without a fused path (Trapezoidal); the LGL dense stage places its
// accumulators itself (defect_kernels.h, ASM).
__global__ __launch_bounds__(256) void kkt_scatter_kernel(const double* __restrict__ blocks,
                                                          const int32_t* __restrict__ map, size_t nslots,
                                                          double* values) {
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nslots; i += stride) {
    const int32_t m = map[i];
    if (m >= 0) values[m] = blocks[i];
    else unsafeAtomicAdd(values + (-(m + 2)), blocks[i]);   // global_atomic_add_f64
  }
}

}  // namespace asset_hip
