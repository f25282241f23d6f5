// Micro-benchmark (round 5): v_fmac_f64_dpp ... row_newbcast:k on gfx950 -- an f64 FMA whose multiplier is lane k of each 16-lane row of
// another register ("DP ALU DPP").  (1) what it returns, incl. row_mask / bank_mask as write predicates; (2) how fast a wave issues it,
// alone on its SIMD and with partners, against the plain v_fmac_f64 and against v_readlane + FMA / ds_read broadcast + FMA operand delivery.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dpp.hip -o exp_build/ubench_dpp && exp_build/ubench_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define FMAC_DPP(acc, op, x, K) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(op), "v"(x))

__global__ void semantics(const double* a, const double* b, double* o) {
  const int l = threadIdx.x;
  double op = a[l], x = b[l], acc0 = 0.0, acc1 = 100.0, acc2 = 200.0;
  FMAC_DPP(acc0, op, x, 3);
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0x5 bank_mask:0xf" : "+v"(acc1) : "v"(op), "v"(x));
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0x3" : "+v"(acc2) : "v"(op), "v"(x));
  o[l] = acc0, o[64 + l] = acc1, o[128 + l] = acc2;
}

template <int MODE, int ILP>
__global__ __launch_bounds__(64) void chain(double* out, long long* cyc, const double* in, int iters) {
  __shared__ double lds[64];
  double a[ILP];
  double op = in[threadIdx.x], x = in[64 + threadIdx.x];
  lds[threadIdx.x] = op;
  __syncthreads();
  for (int k = 0; k < ILP; k++) a[k] = threadIdx.x * 1e-3 + k;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int k = 0; k < ILP; k++) {
        if constexpr (MODE == 0) {          // plain FMA, vector operands
          asm("v_fmac_f64 %0, %1, %2" : "+v"(a[k]) : "v"(op), "v"(x));
        } else if constexpr (MODE == 1) {   // DPP broadcast operand
          FMAC_DPP(a[k], op, x, 5);
        } else if constexpr (MODE == 2) {   // v_readlane pair -> scalar operand
          const unsigned lo = __builtin_amdgcn_readlane(__double2loint(op), (r + k) & 63), hi = __builtin_amdgcn_readlane(__double2hiint(op), (r + k) & 63);
          const double s = __hiloint2double(hi, lo);
          asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[k]) : "s"(s), "v"(x));
        } else {                            // LDS broadcast read
          const double s = ((volatile double*)lds)[(r + k) & 63];
          asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[k]) : "v"(s), "v"(x));
        }
      }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int k = 0; k < ILP; k++) s += a[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int ILP>
void run(int wgs, const char* what, const double* in) {
  double* out; long long* cyc;
  hipMalloc(&out, wgs * 64 * 8); hipMalloc(&cyc, wgs * 8);
  const int iters = 1000;
  for (int rep = 0; rep < 2; rep++) chain<MODE, ILP><<<wgs, 64>>>(out, cyc, in, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(wgs);
  hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += double(v); avg /= wgs;
  const char* m[] = {"v_fmac_f64", "v_fmac_f64_dpp newbcast", "2 x v_readlane + v_fmac_f64", "ds_read_b64 + v_fmac_f64"};
  printf("%-22s %-28s ILP %d: %.2f cycles per FMA per wave\n", what, m[MODE], ILP, avg / (double(iters) * 16 * ILP));
  hipFree(out); hipFree(cyc);
}
int main() {
  std::vector<double> ha(128), hb(64), ho(192);
  for (int i = 0; i < 128; i++) ha[i] = 1000.0 + i;
  for (int i = 0; i < 64; i++) hb[i] = 1.0;
  double *a, *b, *o;
  hipMalloc(&a, 128 * 8); hipMalloc(&b, 64 * 8); hipMalloc(&o, 192 * 8);
  hipMemcpy(a, ha.data(), 128 * 8, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 64 * 8, hipMemcpyHostToDevice);
  semantics<<<1, 64>>>(a, b, o);
  hipMemcpy(ho.data(), o, 192 * 8, hipMemcpyDeviceToHost);
  for (int s = 0; s < 3; s++) {
    printf("%s:", s == 0 ? "newbcast:3 full masks      " : (s == 1 ? "newbcast:5 row_mask 0x5 (+100)" : "newbcast:7 bank_mask 0x3 (+200)"));
    for (int l = 0; l < 64; l++) printf(" %g", ho[s * 64 + l]);
    printf("\n");
  }
  for (int wgs : {1024, 2048, 4096}) {
    const char* w = wgs == 1024 ? "one wave per SIMD" : (wgs == 2048 ? "two waves per SIMD" : "four waves per SIMD");
    run<0, 1>(wgs, w, a); run<0, 8>(wgs, w, a);
    run<1, 1>(wgs, w, a); run<1, 8>(wgs, w, a);
    run<2, 8>(wgs, w, a); run<3, 8>(wgs, w, a);
  }
  return 0;
}
