// Native RCCL transport of libmgx.so: the reference's MPI traffic (mg_mpi_exchange.f90:504-718 halo exchange, :1555-1571
// global_sum, mg_gather.f90:126 gather) as RCCL calls enqueued on the solver's stream -- no host language in the loop.
//
//   fill_halo_*  : ncclGroupStart; ncclRecv/ncclSend to the <= 8 neighbours; ncclGroupEnd   (one group per halo fill: on the
//                  point-to-point xGMI fabric every neighbour is one hop, all edges of a fill travel concurrently)
//   global_sum   : ncclAllReduce of the one double
//   gather_3D    : grouped send/recv inside the reference's colour group (<= 4 members; no sub-communicator needed)
//
// librccl is bound at run time (dlopen), not at link time: a process that already carries an RCCL (PyTorch ships its own,
// built against the HIP runtime it loaded) must use THAT one, and a single-rank run needs none at all.
// Bootstrap: rank 0 calls mgx_rccl_get_unique_id, the caller broadcasts the 128 bytes by whatever it has (MPI_Bcast,
// torch.distributed, a file), every rank calls mgx_rccl_connect (collective, one rank per GPU).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include <rccl.h>

#include "../../include/mgx.h"

namespace {

struct Api {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  int (*GetVersion)(int *) = nullptr;
};

Api A;
ncclComm_t g_comm = nullptr;
int g_nranks = 0, g_rank = -1;
std::string g_err, g_where;

template <typename F> bool bind(F &fn, void *h, const char *name) {
  fn = reinterpret_cast<F>(dlsym(h, name));
  return fn != nullptr;
}

bool bind_all(void *h) {
  return bind(A.GetUniqueId, h, "ncclGetUniqueId") && bind(A.CommInitRank, h, "ncclCommInitRank") && bind(A.CommDestroy, h, "ncclCommDestroy") &&
         bind(A.GroupStart, h, "ncclGroupStart") && bind(A.GroupEnd, h, "ncclGroupEnd") && bind(A.Send, h, "ncclSend") && bind(A.Recv, h, "ncclRecv") &&
         bind(A.AllReduce, h, "ncclAllReduce") && bind(A.GetErrorString, h, "ncclGetErrorString");
}

// the RCCL already in the process first (global symbols, then a library a host framework loaded privately), else the system one
bool load_api() {
  if (A.GetUniqueId) return true;
  if (bind_all(RTLD_DEFAULT)) { g_where = "already loaded (global scope)"; return true; }
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (int pass = 0; pass < 2; pass++)
    for (const char *n : names) {
      void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (!h) continue;
      if (bind_all(h)) { A.handle = h; g_where = std::string(n) + (pass == 0 ? " (already loaded)" : ""); return true; }
      dlclose(h);
    }
  A = Api();
  g_err = "librccl could not be loaded (dlopen of librccl.so / librccl.so.1 failed)";
  return false;
}

int nfail(ncclResult_t r, const char *what) {
  g_err = std::string(what) + ": " + (A.GetErrorString ? A.GetErrorString(r) : "RCCL error");
  return 1;
}
#define NCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return nfail(r_, #call); } while (0)

}  // namespace

extern "C" {

const char *mgxr_last_error(void) { return g_err.c_str(); }
const char *mgxr_library(void) { return g_where.c_str(); }
int mgxr_connected(void) { return g_comm != nullptr; }
int mgxr_nranks(void) { return g_nranks; }

int mgxr_get_unique_id(void *out) {
  if (!load_api()) return 1;
  ncclUniqueId id;
  NCHK(A.GetUniqueId(&id));
  memcpy(out, &id, sizeof id);
  return 0;
}

int mgxr_connect(const void *idbytes, int nranks, int rank) {
  if (!load_api()) return 1;
  if (g_comm) { (void)A.CommDestroy(g_comm); g_comm = nullptr; }
  ncclUniqueId id;
  memcpy(&id, idbytes, sizeof id);
  NCHK(A.CommInitRank(&g_comm, nranks, id, rank));
  g_nranks = nranks; g_rank = rank;
  return 0;
}

void mgxr_disconnect(void) {
  if (g_comm && A.CommDestroy) (void)A.CommDestroy(g_comm);
  g_comm = nullptr; g_nranks = 0; g_rank = -1;
}

// fill_halo_*: one group of receives and sends (mg_mpi_exchange.f90:504-666 posts all MPI_IRecv, then all MPI_ISend)
int mgxr_exchange(hipStream_t st, int n, const int *peer, double *const *sendbuf, double *const *recvbuf, const int *count) {
  if (!g_comm) { g_err = "RCCL transport is not connected"; return 1; }
  NCHK(A.GroupStart());
  for (int q = 0; q < n; q++) NCHK(A.Recv(recvbuf[q], (size_t)count[q], ncclDouble, peer[q], g_comm, st));
  for (int q = 0; q < n; q++) NCHK(A.Send(sendbuf[q], (size_t)count[q], ncclDouble, peer[q], g_comm, st));
  NCHK(A.GroupEnd());
  return 0;
}

// global_sum (mg_mpi_exchange.f90:1555-1571): in place
int mgxr_allreduce(hipStream_t st, double *buf, int n) {
  if (!g_comm) { g_err = "RCCL transport is not connected"; return 1; }
  NCHK(A.AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, g_comm, st));
  return 0;
}

// gather_3D (mg_gather.f90:126): MPI_ALLGATHER on the colour group = every member sends its block to the others
int mgxr_allgather(hipStream_t st, const int *group, int ng, const double *sendbuf, double *recvbuf, int count) {
  if (!g_comm) { g_err = "RCCL transport is not connected"; return 1; }
  int me = -1;
  for (int q = 0; q < ng; q++) if (group[q] == g_rank) me = q;
  if (me < 0) { g_err = "allgather: this rank is not a member of the group"; return 1; }
  if (hipMemcpyAsync(recvbuf + (size_t)me * count, sendbuf, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) {
    g_err = "allgather: device copy of the own block failed"; return 1;
  }
  NCHK(A.GroupStart());
  for (int q = 0; q < ng; q++) if (q != me) NCHK(A.Recv(recvbuf + (size_t)q * count, (size_t)count, ncclDouble, group[q], g_comm, st));
  for (int q = 0; q < ng; q++) if (q != me) NCHK(A.Send(sendbuf, (size_t)count, ncclDouble, group[q], g_comm, st));
  NCHK(A.GroupEnd());
  return 0;
}

}  // extern "C"
