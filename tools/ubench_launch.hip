// What a launch costs by grid shape: an (almost) empty kernel with the resident kernel's LDS per wave, back to back in one stream.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o exp_build/ubench_launch && exp_build/ubench_launch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* p, int n) {
  extern __shared__ double lds[];
  if (n < 0) { lds[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[blockIdx.x] = lds[(threadIdx.x + 1) % blockDim.x]; }
}
int main() {
  double* d; hipMalloc(&d, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int shapes[][2] = {{2048, 64}, {1024, 128}, {512, 256}, {256, 512}, {256, 1024}, {1024, 64}, {512, 128}, {256, 256}};
  for (int rep = 0; rep < 2; rep++)
    for (auto& s : shapes) {
      const size_t lds = size_t(s[1] / 64) * 20480 > 163840 ? 163840 : size_t(s[1] / 64) * 20480;
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      for (int i = 0; i < 50; i++) k<<<s[0], s[1], lds>>>(d, 0);
      hipEventRecord(e0);
      for (int i = 0; i < 2000; i++) k<<<s[0], s[1], lds>>>(d, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("grid %5d x %4d threads, %6zu B LDS: %.2f us per launch\n", s[0], s[1], lds, ms * 1e3 / 2000);
    }
  return 0;
}
