// Micro-benchmark: how fast does ONE wave issue f64 FMAs (dependent chains of ILP 1/2/4/8), alone on its SIMD and with a partner?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o exp_build/ubench_issue && exp_build/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int ILP>
__global__ __launch_bounds__(64) void chain(double* out, long long* cyc, int iters, int active) {
  double a[ILP];
  const double b = 1.0000001, c = 1e-9;
  for (int k = 0; k < ILP; k++) a[k] = threadIdx.x * 1e-3 + k;
  if (int(threadIdx.x) >= active) return;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int k = 0; k < ILP; k++) a[k] = fma(a[k], b, c);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int k = 0; k < ILP; k++) s += a[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP>
void run(int wgs, int active, const char* what) {
  double* out; long long* cyc;
  hipMalloc(&out, wgs * 64 * 8); hipMalloc(&cyc, wgs * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; rep++) chain<ILP><<<wgs, 64>>>(out, cyc, iters, active);
  hipDeviceSynchronize();
  std::vector<long long> h(wgs);
  hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += double(v); avg /= wgs;
  printf("%-28s ILP %d lanes %2d: %.2f cycles per FMA instruction per wave\n", what, ILP, active, avg / (double(iters) * 16 * ILP));
  hipFree(out); hipFree(cyc);
}
int main() {
  // 256 CUs x 4 SIMDs: 1024 single-wave workgroups = one wave per SIMD; 2048 = two; 4096 = four
  for (int wgs : {1024, 2048, 4096}) {
    const char* w = wgs == 1024 ? "one wave per SIMD" : (wgs == 2048 ? "two waves per SIMD" : "four waves per SIMD");
    run<1>(wgs, 64, w); run<2>(wgs, 64, w); run<4>(wgs, 64, w); run<8>(wgs, 64, w);
    run<1>(wgs, 16, w); run<4>(wgs, 16, w); run<4>(wgs, 32, w);
  }
  return 0;
}
