// Micro-benchmark (design aid, not product code): how fast can persistent single-wave workgroups write per-segment
// blocks of 1008 doubles (the Reentry-LGL7 KKT block, 8 064 B) to HBM, by store pattern, with and without a compute
// phase between the blocks of a wave.   hipcc --offload-arch=gfx950 -O3 tools/ubench_store.hip -o gpurun_out/ubench_store
//   pattern 0: dwordx2 per lane, 64 lanes contiguous (512 B per instruction, 16 instructions per block)
//   pattern 1: dwordx4 per lane, 64 lanes contiguous (1 KiB per instruction, 8 instructions per block)
//   pattern 2: dwordx2, four 16-lane groups per instruction, each 128 B contiguous at an 8-byte-aligned column start
//              of the block's slot order (what the dense stage's accumulator-fragment stores do), 20 instructions
//   pattern 3: pattern 2's values staged through LDS in slot order, then written as pattern 1
// `spin`: dependent FMA chain of that many iterations before each block's stores (stands for the products of a segment).
// `fence`: s_waitcnt vmcnt(0) at the top of every block (what the compiler did before the load fence was added).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NK = 1008, IR = 32, OR = 15;

template <int PATTERN>
__global__ __launch_bounds__(64) void store_kernel(double* out, const int* offs, int nseg, int spin, int fence, double seed) {
  __shared__ double stage[NK + 16];
  const int lane = threadIdx.x;
  const int per = nseg / int(gridDim.x), rem = nseg % int(gridDim.x);
  const int first = int(blockIdx.x) * per + min(int(blockIdx.x), rem);
  const int count = per + (int(blockIdx.x) < rem ? 1 : 0);
  int off[20];
  if constexpr (PATTERN >= 2) {
#pragma unroll
    for (int f = 0; f < 20; f++) off[f] = offs[f * 64 + lane];
  }
  double acc = seed + lane;
  for (int g = 0; g < count; g++) {
    if (fence & 1) __builtin_amdgcn_s_waitcnt(0x0F70);
    if (fence & 2) {                      // antiphase: odd wave slots of a SIMD start half a period late
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      if (g == 0 && (hwid & 1u))
        for (int it = 0; it < spin / 2 + 55; it++) __builtin_amdgcn_s_sleep(1);
    }
    for (int it = 0; it < spin; it++) __builtin_amdgcn_s_sleep(1);        // 64 idle cycles per iteration: "compute" that uses no pipe
    double* blk = out + size_t(first + g) * NK;
    if constexpr (PATTERN == 0) {
#pragma unroll
      for (int t = 0; t < 16; t++)
        if (t * 64 + lane < NK) blk[t * 64 + lane] = acc + t;
    } else if constexpr (PATTERN == 1) {
      using d2 = __attribute__((ext_vector_type(2))) double;
#pragma unroll
      for (int t = 0; t < 8; t++)
        if (t * 128 + lane * 2 < NK) *reinterpret_cast<d2*>(blk + t * 128 + lane * 2) = d2{acc + t, acc - t};
    } else if constexpr (PATTERN == 2) {
#pragma unroll
      for (int f = 0; f < 20; f++)
        if (off[f] >= 0) blk[off[f]] = acc + f;
    } else {
      using d2 = __attribute__((ext_vector_type(2))) double;
#pragma unroll
      for (int f = 0; f < 20; f++)
        if (off[f] >= 0) stage[off[f]] = acc + f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      d2 v[8];
#pragma unroll
      for (int t = 0; t < 8; t++) v[t] = *reinterpret_cast<d2*>(stage + ((t * 128 + lane * 2 < NK) ? t * 128 + lane * 2 : 0));
#pragma unroll
      for (int t = 0; t < 8; t++)
        if (t * 128 + lane * 2 < NK) *reinterpret_cast<d2*>(blk + t * 128 + lane * 2) = v[t];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

int main(int argc, char** argv) {
  const int nseg = argc > 1 ? atoi(argv[1]) : 10000;
  // accumulator-fragment slot offsets of the Reentry-LGL7 block (defect_kernels.h: LaneConsts::hst / jst): 3 H tiles
  // (lower triangle of 2x2 tiles), 2 J tiles, 4 entries each
  std::vector<int> offs(20 * 64, -1);
  for (int l = 0; l < 64; l++) {
    const int lr = l & 15, lk = l >> 4;
    int f = 0;
    for (int rt = 0; rt < 2; rt++)
      for (int ct = 0; ct <= rt; ct++)
        for (int v = 0; v < 4; v++, f++) {
          const int c = 16 * ct + lk + 4 * v, r = 16 * rt + lr, cst = c * (IR + OR) - c * (c - 1) / 2;
          if (r >= c) offs[f * 64 + l] = cst + (r - c);
        }
    for (int ct = 0; ct < 2; ct++)
      for (int v = 0; v < 4; v++, f++) {
        const int c = 16 * ct + lk + 4 * v, cst = c * (IR + OR) - c * (c - 1) / 2;
        if (lr < OR) offs[f * 64 + l] = cst + (IR - c) + lr;
      }
  }
  double* out;
  int* doffs;
  CHECK(hipMalloc(&out, size_t(nseg) * NK * 8));
  CHECK(hipMalloc(&doffs, offs.size() * 4));
  CHECK(hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const double mb = double(nseg) * NK * 8 / 1e6;
  printf("nseg %d, %.1f MB per pass\n", nseg, mb);
  auto run = [&](int pattern, int grid, int spin, int fence) {
    auto launch = [&]() {
      switch (pattern) {
        case 0: hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(64), 0, 0, out, doffs, nseg, spin, fence, 1.0); break;
        case 1: hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(64), 0, 0, out, doffs, nseg, spin, fence, 1.0); break;
        case 2: hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(64), 0, 0, out, doffs, nseg, spin, fence, 1.0); break;
        default: hipLaunchKernelGGL(store_kernel<3>, dim3(grid), dim3(64), 0, 0, out, doffs, nseg, spin, fence, 1.0); break;
      }
    };
    for (int i = 0; i < 5; i++) launch();
    CHECK(hipEventRecord(e0, 0));
    const int iters = 50;
    for (int i = 0; i < iters; i++) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    printf("pattern %d grid %5d spin %5d fence %d : %8.2f us  %7.1f GB/s\n", pattern, grid, spin, fence, us, mb / us * 1e3 / 1e3);
  };
  for (int grid : {1024, 2048, 4096})
    for (int pattern = 0; pattern < 4; pattern++) run(pattern, grid, 0, 0);
  CHECK(hipMemset(out, 0, size_t(nseg) * NK * 8));
  {
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < 50; i++) CHECK(hipMemsetAsync(out, 0, size_t(nseg) * NK * 8, 0));
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipMemsetAsync: %8.2f us %7.1f GB/s\n", ms * 1e3 / 50, mb / (ms * 1e3 / 50));
  }
  // idle time between the blocks of a wave (s_sleep: 64 cycles per unit): 94 units ~ 6 k cycles, what a dense-stage
  // segment computes before its stores.  fence bit 0: vmcnt(0) at the top of every block; bit 1: antiphase start.
  for (int spin : {47, 94})
    for (int pattern : {1, 2})
      for (int fence : {0, 1, 2}) run(pattern, 2048, spin, fence);
  return 0;
}
