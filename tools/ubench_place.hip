// Where do the waves of a grid of two-wave (or one-wave) workgroups land?  Prints, per SIMD, which (workgroup, wave) it hosts.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_place.hip -o exp_build/ubench_place && exp_build/ubench_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
__global__ void probe(unsigned* out, int spin) {
  extern __shared__ double lds[];
  const int w = threadIdx.x >> 6;
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}           // stay resident until the whole grid is placed
  asm volatile("" ::: "v247");               // as many registers as the resident kernel: two waves fill a SIMD
  if ((threadIdx.x & 63) == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[(blockIdx.x * 2 + w) * 2 + 0] = hw;
    out[(blockIdx.x * 2 + w) * 2 + 1] = xcc;
    lds[w] = 1.0;
  }
}
void run(int wgs, int threads, size_t shmem) {
  unsigned* d; hipMalloc(&d, wgs * 2 * 2 * 4); hipMemset(d, 0xff, wgs * 2 * 2 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, int(shmem));
  hipLaunchKernelGGL(probe, dim3(wgs), dim3(threads), shmem, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(wgs * 4); hipMemcpy(h.data(), d, wgs * 16, hipMemcpyDeviceToHost);
  std::map<std::tuple<int,int,int,int>, std::vector<std::pair<int,int>>> simd;   // (xcc, se, cu, simd) -> (wg, wave)
  const int nw = threads / 64;
  for (int b = 0; b < wgs; b++) for (int w = 0; w < nw; w++) {
    unsigned hw = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1] & 0xf;
    int simdid = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    simd[{int(xcc), se * 2 + sh, cu, simdid}].push_back({b, w});
  }
  printf("grid %d x %d threads, %zu B LDS: %zu SIMDs used\n", wgs, threads, shmem, simd.size());
  int shown = 0;
  std::map<std::string, int> pattern;
  for (auto& kv : simd) {
    char buf[256]; int n = 0;
    for (auto& p : kv.second) n += snprintf(buf + n, sizeof buf - n, "(+%d,w%d) ", p.first - kv.second[0].first, p.second);
    pattern[buf]++;
    if (shown++ < 6) {
      printf("  xcc %d se %d cu %2d simd %d:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
      for (auto& p : kv.second) printf(" (wg %d, w%d)", p.first, p.second);
      printf("\n");
    }
  }
  for (auto& kv : pattern) printf("  pattern %s x %d\n", kv.first.c_str(), kv.second);
  hipFree(d);
}
int main() {
  run(2048, 64, 19616);
  run(1024, 128, 2 * 19616);
  run(512, 128, 2 * 19616);
  run(1024, 64, 40000);
  run(512, 256, 4 * 19616);
  return 0;
}
