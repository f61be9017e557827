// How long does a launch take whose workgroups exit at once, all of them or all but those with blockIdx % 8 == 0 (which spin ~5 us)?
// hipcc --offload-arch=gfx950 -O3 -o empty_blocks empty_blocks.hip && ./empty_blocks
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int mode, long long ticks, int *sink) {
  if (mode == 0) return;
  if ((blockIdx.x & 7) != 0) return;
  if (threadIdx.x >= 64) return;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks < 0) *sink = 1;
}
int main() {
  int *d; hipMalloc(&d, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grids[] = {8, 512, 2048, 4096, 8192, 16384};
  for (int mode = 0; mode < 2; mode++)
    for (int g : grids) {
      for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, mode, 500LL, d);
      hipDeviceSynchronize();
      hipEventRecord(a, 0);
      for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, mode, 500LL, d);
      hipEventRecord(b, 0); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("mode %d (%s) grid %6d x 256 threads: %.2f us per launch\n", mode, mode ? "1/8 of the workgroups spin 5 us" : "all exit at once", g, ms / 20 * 1e3);
    }
  return 0;
}
