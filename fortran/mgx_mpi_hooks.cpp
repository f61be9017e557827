// MPI implementation of libmgx's three communication hooks (include/mgx.h: mgx_set_comm), for a Fortran/C model that
// runs under MPI like the reference does (src/mg_mpi_exchange.f90, src/mg_gather.f90).  One MPI rank per GPU process.
//
//   mgx_mpi_install(fcomm)   install the hooks on the Fortran communicator handle fcomm (call before nhydro_init)
//   mgx_mpi_connect_p2p()    after nhydro_init: all-gather the hipIpc handles and switch the cycle's halo fills and
//                            coarse-level gathers to the peer-to-peer pushes (no MPI call inside a V-cycle any more);
//                            ends with a rank-coded halo fill through both transports, which must agree on every rank
//   mgx_mpi_connect_rccl()   before or after nhydro_init: MPI only broadcasts the 128-byte RCCL id, then libmgx.so's native
//                            RCCL transport (include/mgx.h) replaces the three hooks -- needs one GPU per rank
//
// The hooks themselves stage through host memory (hipMemcpy + MPI on host buffers), which works with any MPI library;
// with a GPU-aware MPI the staging copies can be dropped (MGX_MPI_GPU_AWARE=1 passes the device pointers to MPI).
// They carry the set-up halos, the norm all-reduce and -- when the peer-to-peer transport is not connected -- the
// halo exchanges of the cycle (mg_mpi_exchange.f90:504-718) and gather_3D (mg_gather.f90:126).
#include <hip/hip_runtime.h>
#include <mpi.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
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
  if (hipDeviceSynchronize() != hipSuccess) return 1;  // the producer ran on the solver's stream, hipMemcpy orders with the null stream only
  if (hipMemcpy(h.data(), buf, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  MPI_Allreduce(h.data(), o.data(), n, MPI_DOUBLE, MPI_SUM, g_comm);
  return hipMemcpy(buf, o.data(), n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess;
}

// all-gather inside the <=4-member colour group of a gathered level, as point-to-point messages (no sub-communicator)
int hook_allgather(void *, const int *group, int ng, const double *sendbuf, double *recvbuf, int count) {
  std::vector<double> hs(count), hr((size_t)count * ng);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
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
  if (!getenv("MGX_MPI_NO_SETDEVICE")) {  // one rank per GPU: the node-local rank picks the device (all ranks share it on a one-GPU box)
    MPI_Comm node; int lrank = 0, ndev = 0;
    MPI_Comm_split_type(g_comm, MPI_COMM_TYPE_SHARED, g_rank, MPI_INFO_NULL, &node);
    MPI_Comm_rank(node, &lrank);
    MPI_Comm_free(&node);
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) (void)hipSetDevice(lrank % ndev);
  }
  return mgx_set_comm(hook_exchange, hook_allreduce, hook_allgather, nullptr);
}

// libmgx.so's native RCCL transport instead of the MPI hooks: MPI carries the bootstrap id only.  Collective; on any
// failure every rank stays on (returns to) the MPI hooks and 1 is returned.
int mgx_mpi_connect_rccl(void) {
  std::vector<char> id(mgx_rccl_unique_id_bytes());
  int ok = 1, allok = 0;
  if (g_rank == 0) ok = mgx_rccl_get_unique_id(id.data()) == 0;
  MPI_Bcast(id.data(), (int)id.size(), MPI_BYTE, 0, g_comm);
  MPI_Allreduce(&ok, &allok, 1, MPI_INT, MPI_MIN, g_comm);
  if (allok) {
    ok = mgx_rccl_connect(id.data(), g_size, g_rank) == 0;
    MPI_Allreduce(&ok, &allok, 1, MPI_INT, MPI_MIN, g_comm);
  }
  if (!allok) { mgx_rccl_disconnect(); mgx_set_comm(hook_exchange, hook_allreduce, hook_allgather, nullptr); return 1; }
  return 0;
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
  // data self-test (as mgroms_amd.parallel.Comm._p2p_selftest): one level-1 halo fill of a rank-coded field through the
  // pushes and through the hooks; the pushes stay on only if both give every rank the same halos
  int nx, ny, nz;
  if (mgx_level_dims(1, &nx, &ny, &nz)) { mgx_set_option("p2p", 0); return 1; }
  const size_t n3 = (size_t)nz * (ny + 2) * (nx + 2);
  std::vector<double> pat(n3), a(n3), b(n3);
  for (size_t q = 0; q < n3; q++) pat[q] = 1000.0 * (g_rank + 1) + (double)(q % 997) * 1e-3;
  ok = mgx_set_field(1, MGX_P, pat.data()) == 0 && mgx_fill_halo(1, MGX_P) == 0 && mgx_get_field(1, MGX_P, a.data()) == 0;
  mgx_set_option("p2p", 0);
  const int ok2 = mgx_set_field(1, MGX_P, pat.data()) == 0 && mgx_fill_halo(1, MGX_P) == 0 && mgx_get_field(1, MGX_P, b.data()) == 0;
  ok = ok && ok2 && memcmp(a.data(), b.data(), n3 * sizeof(double)) == 0;
  std::fill(pat.begin(), pat.end(), 0.0);
  (void)mgx_set_field(1, MGX_P, pat.data());
  MPI_Allreduce(&ok, &allok, 1, MPI_INT, MPI_MIN, g_comm);
  if (!allok) return 1;   // p2p is off on every rank
  mgx_set_option("p2p", 1);
  return 0;
}
}
