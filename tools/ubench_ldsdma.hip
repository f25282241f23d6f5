// global -> LDS without registers (global_load_lds_dword, M0 = LDS byte address of lane 0's word, the instruction offset moves both
// sides): what defect_resident.h's slot prefetch relies on.  Copies n doubles to an 8-byte-aligned (not 16) LDS address and back.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_ldsdma.hip -o exp_build/ubench_ldsdma && exp_build/ubench_ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline void lds_dma4(unsigned lds_byte_addr, const void* gbase, unsigned voff) {   // four 256-byte rows
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
               "global_load_lds_dword %2, %3\n\tglobal_load_lds_dword %2, %3 offset:256\n\t"
               "global_load_lds_dword %2, %3 offset:512\n\tglobal_load_lds_dword %2, %3 offset:768\n\t"
               "s_mov_b32 m0, %0" : "=&s"(keep) : "s"(lds_byte_addr), "v"(voff), "s"(gbase) : "memory");
}
__global__ void k(const double* in, double* out, int n, long long* cyc) {
  extern __shared__ double lds[];
  const unsigned lane = threadIdx.x;
  const unsigned base = (unsigned)(size_t)(lds + 3);
  const long long t0 = clock64();
  for (int e = 0; e < 2 * n; e += 256) lds_dma4(__builtin_amdgcn_readfirstlane(base + e * 4), in, (e + lane) * 4);
  const long long t1 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = clock64();
  __syncthreads();
  for (int e = lane; e < n; e += 64) out[e] = lds[3 + e];
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
  const int n = 1152; static double h[1152], r[1152]; double *d, *o; long long *c, hc[2];
  for (int i = 0; i < n; i++) h[i] = i + 0.5;
  hipMalloc(&d, n * 8); hipMalloc(&o, n * 8); hipMalloc(&c, 16); hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++) {
    hipMemset(o, 0, n * 8); k<<<1, 64, 16384>>>(d, o, n, c); hipMemcpy(r, o, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < n; i++) bad += r[i] != h[i];
    printf("lds dma: %d mismatches of %d doubles; issue %lld cycles, wait %lld cycles\n", bad, n, hc[0], hc[1]);
  }
  return 0;
}
