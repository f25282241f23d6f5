// Issue cost of single f64 vector instructions, one wave per SIMD, 8 independent chains (tools/ubench_issue.hip for the FMA).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_ops.hip -o exp_build/ubench_ops && exp_build/ubench_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(64) void chain(double* out, long long* cyc, int iters, double b, double c) {
  double a[8];
  for (int k = 0; k < 8; k++) a[k] = threadIdx.x * 1e-3 + k + 1.0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if constexpr (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if constexpr (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if constexpr (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        if constexpr (OP == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "s"(b));
        if constexpr (OP == 4) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if constexpr (OP == 5) asm volatile("v_rcp_f64_e32 %0, %0" : "+v"(a[k]));
        if constexpr (OP == 6) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(reinterpret_cast<int*>(&a[k])[0]) : "v"(reinterpret_cast<const int*>(&b)[0]));
        if constexpr (OP == 7) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(reinterpret_cast<int*>(&a[k])[0]) : "v"(reinterpret_cast<const int*>(&b)[0]));
        if constexpr (OP == 8) asm volatile("v_fma_f64 %0, %0, %1, 0.5" : "+v"(a[k]) : "s"(b));
        if constexpr (OP == 9) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(a[k]));
      }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int k = 0; k < 8; k++) s += a[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* what, int wgs = 1024) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, wgs * 64 * 8); (void)hipMalloc(&cyc, wgs * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; rep++) chain<OP><<<wgs, 64>>>(out, cyc, iters, 1.0000001, 1e-9);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(wgs);
  (void)hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += double(v); avg /= wgs;
  printf("%-44s %d waves: %.2f cycles per instruction per wave\n", what, wgs, avg / (double(iters) * 64));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("v_fma_f64 v,v,v,v"); run<1>("v_mul_f64 v,v,v"); run<2>("v_add_f64 v,v,v"); run<3>("v_mul_f64 v,v,s");
  run<4>("v_fmac_f64_e32"); run<5>("v_rcp_f64"); run<6>("v_mov_b32"); run<7>("v_cndmask_b32"); run<8>("v_fma_f64 v,v,s,0.5"); run<9>("v_mul_f64 v,v,v (same)");
  run<0>("v_fma_f64 v,v,v,v", 2048); run<1>("v_mul_f64 v,v,v", 2048); run<6>("v_mov_b32", 2048);
  return 0;
}
