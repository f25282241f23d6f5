// What v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950) return: prints, for every lane, the source lane of both results.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_permlane.hip -o exp_build/ubench_permlane && exp_build/ubench_permlane
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) unsigned int u2;
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  const u2 a = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
  const u2 b = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
  out[l] = a[0]; out[64 + l] = a[1]; out[128 + l] = b[0]; out[192 + l] = b[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[4] = {"permlane16_swap(old = lane, src = lane + 100)[0]", "permlane16_swap[1]", "permlane32_swap[0]", "permlane32_swap[1]"};
  for (int r = 0; r < 4; r++) { printf("%s:\n", nm[r]); for (int l = 0; l < 64; l++) printf("%4u%s", h[64 * r + l], (l & 15) == 15 ? "\n" : ""); }
  return 0;
}
