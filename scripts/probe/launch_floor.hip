// Back-to-back dependent (same stream) kernels that do nothing or touch a little memory: time per launch by grid shape.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int n) { if (n == 12345) printf("x"); }
__global__ void __launch_bounds__(512) k_lds(int n) { extern __shared__ double s[]; if (n == 12345) s[threadIdx.x] = 1.0; }
__global__ void k_touch(double *a, int n) { const int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) a[t] += 1.0; }
template <class F> float timeit(F f, int reps, hipStream_t st) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; i++) f();
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / reps;
}
int main() {
  hipStream_t st; hipStreamCreate(&st);
  double *a; hipMalloc(&a, 64 << 20); hipMemset(a, 0, 64 << 20);
  const int reps = 2000;
  printf("empty 1x64        : %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, 0); }, reps, st));
  printf("empty 256x512     : %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, 0); }, reps, st));
  printf("empty 1024x128    : %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(128), 0, st, 0); }, reps, st));
  printf("lds48K 256x512    : %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 48 * 1024, st, 0); }, reps, st));
  printf("lds48K 64x512     : %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_lds, dim3(64), dim3(512), 48 * 1024, st, 0); }, reps, st));
  printf("touch 4 MB        : %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, st, a, 1 << 19); }, reps, st));
  printf("touch 32 KB       : %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_touch, dim3(16), dim3(256), 0, st, a, 1 << 12); }, reps, st));
  // graph of 20 empty kernels
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 48 * 1024, st, 0);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  printf("graph lds48K 256x512: %.2f us per kernel\n", timeit([&] { hipGraphLaunch(ge, st); }, 200, st) / 20);
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_touch, dim3(16), dim3(256), 0, st, a, 1 << 12);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  printf("graph touch 32 KB : %.2f us per kernel\n", timeit([&] { hipGraphLaunch(ge, st); }, 200, st) / 20);
  return 0;
}
