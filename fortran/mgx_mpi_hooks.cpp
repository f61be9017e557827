// MPI implementation of libmgx's three communication hooks (include/mgx.h: mgx_set_comm), for a Fortran/C model that
// runs under MPI like the reference does (src/mg_mpi_exchange.f90, src/mg_gather.f90).  One MPI rank per GPU process.
//
//   mgx_mpi_install(fcomm)   install the hooks on the Fortran communicator handle fcomm (call before nhydro_init)
//   mgx_mpi_connect_p2p()    after nhydro_init: all-gather the hipIpc handles and switch the cycle's halo fills and
//                            coarse-level gathers to the peer-to-peer pushes (no MPI call inside a V-cycle any more)
//
// The hooks themselves stage through host memory (hipMemcpy + MPI on host buffers), which works with any MPI library;
// with a GPU-aware MPI the staging copies can be dropped (MGX_MPI_GPU_AWARE=1 passes the device pointers to MPI).
// They carry the set-up halos, the norm all-reduce and -- when the peer-to-peer transport is not connected -- the
// halo exchanges of the cycle (mg_mpi_exchange.f90:504-718) and gather_3D (mg_gather.f90:126).
#include <hip/hip_runtime.h>
#include <mpi.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/mgx.h"

namespace {
MPI_Comm g_comm = MPI_COMM_NULL;
int g_rank = 0, g_size = 1, g_aware = 0;
std::vector<double> g_hs[8], g_hr[8];

int hook_exchange(void *, int n, const int *peer, double *const *sendbuf, double *const *recvbuf, const int *count) {
  MPI_Request rq[16];
  if (hipDeviceSynchronize() != hipSuccess) return 1;  // the packs were launched asynchronously
  for (int q = 0; q < n; q++) {
    double *s = sendbuf[q], *r = recvbuf[q];
    if (!g_aware) {
      g_hs[q].resize(count[q]); g_hr[q].resize(count[q]);
      if (hipMemcpy(g_hs[q].data(), sendbuf[q], (size_t)count[q] * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
      s = g_hs[q].data(); r = g_hr[q].data();
    }
    MPI_Irecv(r, count[q], MPI_DOUBLE, peer[q], 7, g_comm, &rq[2 * q]);
    MPI_Isend(s, count[q], MPI_DOUBLE, peer[q], 7, g_comm, &rq[2 * q + 1]);
  }
  MPI_Waitall(2 * n, rq, MPI_STATUSES_IGNORE);
  if (!g_aware)
    for (int q = 0; q < n; q++)
      if (hipMemcpy(recvbuf[q], g_hr[q].data(), (size_t)count[q] * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}

int hook_allreduce(void *, double *buf, int n) {
  std::vector<double> h(n), o(n);
  if (hipMemcpy(h.data(), buf, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  MPI_Allreduce(h.data(), o.data(), n, MPI_DOUBLE, MPI_SUM, g_comm);
  return hipMemcpy(buf, o.data(), n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess;
}

// all-gather inside the <=4-member colour group of a gathered level, as point-to-point messages (no sub-communicator)
int hook_allgather(void *, const int *group, int ng, const double *sendbuf, double *recvbuf, int count) {
  std::vector<double> hs(count), hr((size_t)count * ng);
  if (hipMemcpy(hs.data(), sendbuf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  MPI_Request rq[8]; int nr = 0;
  for (int q = 0; q < ng; q++) {
    if (group[q] == g_rank) { memcpy(hr.data() + (size_t)q * count, hs.data(), (size_t)count * sizeof(double)); continue; }
    MPI_Irecv(hr.data() + (size_t)q * count, count, MPI_DOUBLE, group[q], 9, g_comm, &rq[nr++]);
    MPI_Isend(hs.data(), count, MPI_DOUBLE, group[q], 9, g_comm, &rq[nr++]);
  }
  MPI_Waitall(nr, rq, MPI_STATUSES_IGNORE);
  return hipMemcpy(recvbuf, hr.data(), (size_t)count * ng * sizeof(double), hipMemcpyHostToDevice) != hipSuccess;
}
}  // namespace

extern "C" {
int mgx_mpi_install(int fcomm) {
  g_comm = MPI_Comm_f2c((MPI_Fint)fcomm);
  MPI_Comm_rank(g_comm, &g_rank);
  MPI_Comm_size(g_comm, &g_size);
  const char *e = getenv("MGX_MPI_GPU_AWARE");
  g_aware = e && atoi(e) != 0;
  return mgx_set_comm(hook_exchange, hook_allreduce, hook_allgather, nullptr);
}

int mgx_mpi_connect_p2p(void) {
  const int nb = mgx_p2p_handle_bytes();
  std::vector<char> mine(nb), all((size_t)nb * g_size);
  int ok = mgx_p2p_prepare(mine.data()) == 0, allok = 0;
  MPI_Allgather(mine.data(), nb, MPI_BYTE, all.data(), nb, MPI_BYTE, g_comm);
  if (ok) ok = mgx_p2p_connect(all.data(), g_size) == 0;
  MPI_Allreduce(&ok, &allok, 1, MPI_INT, MPI_MIN, g_comm);
  if (!allok) { mgx_set_option("p2p", 0); return 1; }  // everybody stays on the MPI hooks
  MPI_Barrier(g_comm);
  return 0;
}
}
