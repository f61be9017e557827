// Feasibility probe for the peer-to-peer halo transport: two processes share fine-grained device memory through
// hipIpc handles; each pushes a block into the other's receive buffer and raises a flag there, the other spins on
// its local flag (bounded) and checks the data.  Usage: ipc_probe <rank 0|1> <dir> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("rank %d: %s failed: %s\n", rank, #x, hipGetErrorString(e)); return 2; } } while (0)
static int rank;

__global__ void k_push(double *remote_buf, unsigned long long *remote_flag, int n, unsigned long long seq) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) remote_buf[t] = (double)seq + 1e-6 * t;
}
__global__ void k_signal(unsigned long long *remote_flag, unsigned long long seq) {
  __threadfence_system();
  __hip_atomic_store(remote_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_wait_check(const double *buf, unsigned long long *flag, int n, unsigned long long seq, int *err) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = 0;
    long long t0 = wall_clock64();
    while (true) {
      if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= seq) { ok = 1; break; }
      if (wall_clock64() - t0 > 300000000LL) break;  // 3 s at 100 MHz
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (!ok) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(err, 1); return; }
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
    if (buf[t] != (double)seq + 1e-6 * t) atomicOr(err, 2);
}

static bool read_file(const std::string &p, void *dst, size_t n) {
  for (int tries = 0; tries < 600; tries++) {
    FILE *f = fopen(p.c_str(), "rb");
    if (f) { size_t g = fread(dst, 1, n, f); fclose(f); if (g == n) return true; }
    usleep(50000);
  }
  return false;
}
static void write_file(const std::string &p, const void *src, size_t n) {
  std::string tmp = p + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb"); fwrite(src, 1, n, f); fclose(f); rename(tmp.c_str(), p.c_str());
}

int main(int argc, char **argv) {
  rank = atoi(argv[1]);
  std::string dir = argv[2];
  int iters = argc > 3 ? atoi(argv[3]) : 1000;
  const int n = 64 * 1024;  // doubles per message (512 KB)
  CK(hipSetDevice(0));
  double *buf; unsigned long long *flag; int *err;
  CK(hipExtMallocWithFlags((void **)&buf, 2 * n * sizeof(double), hipDeviceMallocFinegrained));
  CK(hipExtMallocWithFlags((void **)&flag, 4096, hipDeviceMallocFinegrained));
  CK(hipMemset(flag, 0, 4096)); CK(hipMemset(buf, 0, 2 * n * sizeof(double)));
  CK(hipMalloc((void **)&err, 4)); CK(hipMemset(err, 0, 4));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t hb, hf, pb, pf;
  CK(hipIpcGetMemHandle(&hb, buf)); CK(hipIpcGetMemHandle(&hf, flag));
  write_file(dir + "/hb" + std::to_string(rank), &hb, sizeof hb);
  write_file(dir + "/hf" + std::to_string(rank), &hf, sizeof hf);
  if (!read_file(dir + "/hb" + std::to_string(1 - rank), &pb, sizeof pb) || !read_file(dir + "/hf" + std::to_string(1 - rank), &pf, sizeof pf)) { printf("rank %d: no peer handle\n", rank); return 3; }
  double *rbuf; unsigned long long *rflag;
  CK(hipIpcOpenMemHandle((void **)&rbuf, pb, hipIpcMemLazyEnablePeerAccess));
  CK(hipIpcOpenMemHandle((void **)&rflag, pf, hipIpcMemLazyEnablePeerAccess));
  hipStream_t st; CK(hipStreamCreate(&st));
  auto t0 = std::chrono::steady_clock::now();
  for (int it = 1; it <= iters; it++) {
    const int par = it & 1;
    hipLaunchKernelGGL(k_push, dim3(n / 256), dim3(256), 0, st, rbuf + par * n, rflag + par, n, (unsigned long long)it);
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, st, rflag + par, (unsigned long long)it);
    hipLaunchKernelGGL(k_wait_check, dim3(64), dim3(256), 0, st, buf + par * n, flag + par, n, (unsigned long long)it, err);
  }
  CK(hipStreamSynchronize(st));
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  printf("rank %d: %d exchanges of %d KB, %.1f us each, err=%d (1=timeout 2=data)\n", rank, iters, (int)(n * 8 / 1024), dt / iters * 1e6, herr);
  CK(hipIpcCloseMemHandle(rbuf)); CK(hipIpcCloseMemHandle(rflag));
  return herr ? 1 : 0;
}
