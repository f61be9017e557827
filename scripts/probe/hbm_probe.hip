// What does this MI355X deliver for the access patterns of the solver?  (calibration for the rooflines in DESIGN.md; not part of the library)
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/hbm_probe.hip -o /tmp/hbm_probe && /tmp/hbm_probe
// 1. contiguous streaming read (16 B per lane, grid-stride)          2. contiguous copy
// 3. "column walk": the solver's layout -- a wave reads 512 contiguous bytes of a row (64 columns) and climbs nz rows of stride RS, for NA
//    arrays at once, U rows in flight; planes of nz*RS doubles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ a, size_t n2, double *out) {
  double acc = 0.0;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n2; q += (size_t)gridDim.x * 256) { const double2 v = a[q]; acc += v.x + v.y; }
  if (acc == 123.456) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n2) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n2; q += (size_t)gridDim.x * 256) b[q] = a[q];
}
// block (64, 4): 64 columns x 4 planes; grid (ny/64, nx/4).  NA arrays, each plane*nx doubles; U rows requested before the first is used
template <int NA, int U>
__global__ __launch_bounds__(256) void k_walk(const double *__restrict__ base, size_t arr, int RS, int nz, size_t plane, double *out) {
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  const double *p = base + (size_t)i * plane + j;
  double acc = 0.0;
  for (int k = 0; k < nz; k += U) {
    double v[NA][U];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int q = 0; q < NA; q++) v[q][u] = p[(size_t)q * arr + (size_t)(k + u) * RS];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int q = 0; q < NA; q++) acc += v[q][u];
  }
  if (acc == 123.456) out[0] = acc;
}
// the same bytes, but a wave owns ONE row segment of all planes... (k fastest over the grid): block = 64 lanes, grid = (ny/64) * nx * (nz/U)
template <int NA, int U>
__global__ __launch_bounds__(64) void k_flat(const double *__restrict__ base, size_t arr, int RS, int nz, size_t plane, int gx, double *out) {
  const int nks = nz / U;
  const int ks = blockIdx.x % nks, grp = blockIdx.x / nks, bx = grp % gx, i = grp / gx;
  const double *p = base + (size_t)i * plane + bx * 64 + threadIdx.x + (size_t)ks * U * RS;
  double v[NA][U], acc = 0.0;
#pragma unroll
  for (int u = 0; u < U; u++)
#pragma unroll
    for (int q = 0; q < NA; q++) v[q][u] = p[(size_t)q * arr + (size_t)u * RS];
#pragma unroll
  for (int u = 0; u < U; u++)
#pragma unroll
    for (int q = 0; q < NA; q++) acc += v[q][u];
  if (acc == 123.456) out[0] = acc;
}

// 4. what does a vector-memory INSTRUCTION cost?  The walk of (3) with every row requested DUP times at neighbouring columns (cache hits, as the
//    stencil's neighbour values are), 8 B per lane, or the same bytes as 16-B requests: time against instructions per cell
template <int NA, int DUP, bool WIDE>
__global__ __launch_bounds__(256) void k_stencil(const double *__restrict__ base, size_t arr, int RS, int nz, size_t plane, double *out) {
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  const double *p = base + (size_t)i * plane + j;
  double acc = 0.0;
  for (int k = 0; k < nz; k++) {
#pragma unroll
    for (int q = 0; q < NA; q++) {
      const double *s = p + (size_t)q * arr + (size_t)k * RS;
      if (WIDE) {
#pragma unroll
        for (int d = 0; d < DUP; d += 2) { double2 t; __builtin_memcpy(&t, s + d, 16); acc += t.x + t.y; }
      } else {
#pragma unroll
        for (int d = 0; d < DUP; d++) acc += s[d];
      }
    }
  }
  if (acc == 123.456) out[0] = acc;
}

// 5. the walk of (3) at the occupancy of the solver's register-heavy kernels: one wave per SIMD (a block of 64 lanes that asks for 40 KB of LDS:
//    four blocks per CU), NA arrays x U rows requested before the first use
template <int NA, int U>
__global__ __launch_bounds__(64) void k_walk_1w(const double *__restrict__ base, size_t arr, int RS, int nz, size_t plane, double *out) {
  extern __shared__ double pad[];
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y;
  const double *p = base + (size_t)i * plane + j;
  double acc = 0.0;
  for (int k = 0; k < nz; k += U) {
    double v[NA][U];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int q = 0; q < NA; q++) v[q][u] = p[(size_t)q * arr + (size_t)(k + u) * RS];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int q = 0; q < NA; q++) acc += v[q][u];
  }
  if (acc == 123.456) { out[0] = acc; pad[threadIdx.x] = acc; }
}

// 6. the same with the wave split over TWO planes (lanes 0-31 plane i, lanes 32-63 plane i+1: 256-byte segments), as the fused
//    residual+restriction kernel is laid out
template <int NA, int U>
__global__ __launch_bounds__(64) void k_walk_split(const double *__restrict__ base, size_t arr, int RS, int nz, size_t plane, double *out) {
  extern __shared__ double pad[];
  const int j = blockIdx.x * 32 + (threadIdx.x & 31), i = blockIdx.y * 2 + (threadIdx.x >> 5);
  const double *p = base + (size_t)i * plane + j;
  double acc = 0.0;
  for (int k = 0; k < nz; k += U) {
    double v[NA][U];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int q = 0; q < NA; q++) v[q][u] = p[(size_t)q * arr + (size_t)(k + u) * RS];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int q = 0; q < NA; q++) acc += v[q][u];
  }
  if (acc == 123.456) { out[0] = acc; pad[threadIdx.x] = acc; }
}

template <class F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; r++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int nx = 512, ny = 512, nz = 64, RS = 528;  // one array = 512 planes of 64 rows of 528 doubles = 138 MB
  const size_t plane = (size_t)nz * RS, arr = plane * nx, NAmax = 8;
  double *a, *b, *out;
  CK(hipMalloc(&a, arr * NAmax * 8)); CK(hipMalloc(&b, arr * NAmax * 8)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, arr * NAmax * 8)); CK(hipMemset(b, 0, arr * NAmax * 8));
  const size_t n2 = arr * NAmax / 2;
  const double GB = arr * NAmax * 8 / 1e9;
  for (int g : {2048, 8192, 32768}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, (const double2 *)a, n2, out); }, 10);
    printf("contiguous read  grid %6d: %7.1f us  %6.2f TB/s\n", g, ms * 1e3, GB / ms);
  }
  for (int g : {2048, 8192, 32768}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, n2); }, 10);
    printf("contiguous copy  grid %6d: %7.1f us  %6.2f TB/s (read + write)\n", g, ms * 1e3, 2 * GB / ms);
  }
  const double useful8 = (double)nx * ny * nz * 8 * 8 / 1e9, useful1 = useful8 / 8;
#define WALK(NA, U) { float ms = timeit([&] { hipLaunchKernelGGL((k_walk<NA, U>), dim3(ny / 64, nx / 4), dim3(64, 4), 0, 0, a, arr, RS, nz, plane, out); }, 10); \
    printf("column walk  %d arrays, %2d rows in flight: %7.1f us  %6.2f TB/s\n", NA, U, ms * 1e3, (NA == 8 ? useful8 : useful1) / ms); }
  WALK(1, 1) WALK(1, 4) WALK(1, 16) WALK(1, 64) WALK(8, 1) WALK(8, 2) WALK(8, 4) WALK(8, 8)
#define FLAT(NA, U) { const int gx = ny / 64; float ms = timeit([&] { hipLaunchKernelGGL((k_flat<NA, U>), dim3(gx * nx * (nz / U)), dim3(64), 0, 0, a, arr, RS, nz, plane, gx, out); }, 10); \
    printf("flat (k fastest) %d arrays, %2d rows per wave: %7.1f us  %6.2f TB/s\n", NA, U, ms * 1e3, (NA == 8 ? useful8 : useful1) / ms); }
  FLAT(8, 1) FLAT(8, 2) FLAT(8, 4) FLAT(8, 8) FLAT(1, 8)
#define STEN(NA, DUP, WIDE) { float ms = timeit([&] { hipLaunchKernelGGL((k_stencil<NA, DUP, WIDE>), dim3(ny / 64, nx / 4), dim3(64, 4), 0, 0, a, arr, RS, nz, plane, out); }, 10); \
    printf("stencil walk %d arrays x %d requests per cell (%s): %7.1f us  %6.2f TB/s of new data, %5.1f instr/cell, %5.1f cycles per instruction and CU\n", NA, DUP, WIDE ? "16 B" : " 8 B", \
           ms * 1e3, useful8 / ms, (double)NA * DUP / (WIDE ? 2 : 1), ms * 1e-3 * 2.4e9 / ((double)nx * ny * nz / 64 / 256 * NA * DUP / (WIDE ? 2 : 1))); }
  STEN(8, 1, false) STEN(8, 2, false) STEN(8, 4, false) STEN(8, 2, true) STEN(8, 4, true) STEN(8, 8, true)
#define W1(NA, U) { float ms = timeit([&] { hipLaunchKernelGGL((k_walk_1w<NA, U>), dim3(ny / 64, nx), dim3(64), 40 * 1024, 0, a, arr, RS, nz, plane, out); }, 10); \
    printf("column walk, ONE wave per SIMD, %d arrays x %d rows in flight (%3d requests): %7.1f us  %6.2f TB/s\n", NA, U, NA * U, ms * 1e3, useful8 / ms); }
  W1(8, 1) W1(8, 2) W1(8, 4) W1(8, 8)
#define WS(NA, U) { float ms = timeit([&] { hipLaunchKernelGGL((k_walk_split<NA, U>), dim3(ny / 32, nx / 2), dim3(64), 40 * 1024, 0, a, arr, RS, nz, plane, out); }, 10); \
    printf("column walk, one wave per SIMD, wave split over two planes, %d arrays x %d rows in flight: %7.1f us  %6.2f TB/s\n", NA, U, ms * 1e3, useful8 / ms); }
  WS(8, 1) WS(8, 2) WS(8, 4)
  return 0;
}
