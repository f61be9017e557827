// Host side of libmgx.so: solver state, level hierarchy, halo / gather plumbing, cycle control and the
// C ABI declared in include/mgx.h.  Mirrors the reference's module structure:
//   mg_grids.f90 (levels, neighbours, gather groups)      -> define_levels()
//   mg_define_matrix.f90 (define_matrices_topo)           -> define_matrices()
//   mg_mpi_exchange.f90 (fill_halo_*, global_sum)         -> fill_halo_js(), rl_fill_halo(), global_sum()
//   mg_gather.f90 (gather, split)                         -> inside fine2coarse()/coarse2fine()
//   mg_relax.f90 / mg_intergrids.f90 / mg_solvers.f90     -> relax(), residual(), fine2coarse(), ...
// There is no CPU compute path: every operator is a HIP kernel launch (mgx_kernels.hip, mgx_setup.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mgx.h"
#include "mgx_internal.h"

struct RectOp { int op, nzz, nh, ny, j0, j1, i0, i1, mj, cj, mi, ci, mj2, cj2, mi2, ci2; };
struct ModelView { double *u, *v, *w, *rmask; int bmask; };

extern "C" {
int mgxk_relax_ks_pair(hipStream_t, const LevView *, int, int, int, Sides);
int mgxk_relax_ks_persist(hipStream_t, const LevView *, int, int, Sides, unsigned int *, unsigned int, int *, int);
int mgxk_set_ksp_timeout(double);
int mgxk_relax_colour(hipStream_t, const LevView *, int, int, int, int, int, int, int, Sides);
int mgxk_relax_small(hipStream_t, const LevView *, int, int, int, Sides, int);
int mgxk_relax_wave_fused(hipStream_t, const LevView *, const LevView *, int, int, int, Sides, int, int);
int mgxk_relax_gs_sweep(hipStream_t, const LevView *, int);
void mgxk_snapshot_k1(hipStream_t, const LevView *);
void mgxk_rbseq_setup(hipStream_t, const LevView *);
int mgxk_rbseq_scan(hipStream_t, const LevView *, int, int);
int mgxk_rbseq_wants_d0(const LevView *);
int mgxk_rbseq_walk_apply(hipStream_t, const LevView *, int, Sides, int);
int mgxk_rbseq_scan_apply(hipStream_t, const LevView *, int, Sides, int, int, unsigned int *, unsigned int, int *, int, long long);
int mgxk_set_rbseq_timeout(double);
void mgxk_rbseq_apply(hipStream_t, const LevView *, int, Sides, int);
void mgxk_rbseq_rho(hipStream_t, const LevView *, double *);
int mgxk_rbseq_window_planes(double);
int mgxk_rbseq_window(hipStream_t, const LevView *, int, Sides, int, int, int);
void mgxk_rbseq_gdecay(hipStream_t, const LevView *, double *);
int mgxk_rbseq_window_rows(const double *, int);
void mgxk_rbseq_d0(hipStream_t, const LevView *, int);
int mgxk_coarse_direct_cells(const LevView *);
int mgxk_coarse_direct_slabs(int);
int mgxk_coarse_direct_build(hipStream_t, const LevView *, int, int, int, Sides, int, double *, long long, double *);
int mgxk_coarse_direct_apply(hipStream_t, const LevView *, const double *, double *, unsigned int *, Sides);
int mgxk_has_reg_kernel(const LevView *);
int mgxk_residual_nblocks(const LevView *);
void mgxk_residual(hipStream_t, const LevView *, double *, double *, int, int, Sides);
void mgxk_sumsq(hipStream_t, const LevView *, const double *, double *, double *);
void mgxk_dot(hipStream_t, const LevView *, const double *, const double *, double *, double *);
void mgxk_fine2coarse(hipStream_t, const LevView *, const LevView *, double *, Sides, double *dup, double *zero);
void mgxk_restrict_chain(hipStream_t, const LevView *const *, int, Sides);
int mgxk_residual_restrict(hipStream_t, const LevView *, const LevView *, double *, int real, Sides, double *zero);
int mgxk_residual_restrict_ex(hipStream_t, const LevView *, const LevView *, double *, int real, Sides, double *zero, double *partial, double *dup);
int mgxk_residual_restrict_grid(const LevView *, const LevView *);
void mgxk_reduce(hipStream_t, const double *, int, double *);
void mgxk_coarse2fine(hipStream_t, const LevView *, const LevView *, const double *, int, Sides, int, int);
void mgxk_divc_selftest(hipStream_t, const double *, const double *, int, unsigned long long *);
void mgxk_halo_phys(hipStream_t, const LevView *, double *, Sides);
void mgxk_halo_mixed_corners(hipStream_t, const LevView *, double *, int, int, int, int);
void mgxk_halo_p2p(hipStream_t, const LevView *, double *, double *const *, double *const *, unsigned long long *const *,
                   unsigned long long *const *, const int *, unsigned long long, unsigned int *, int *, const int *, int);
int mgxk_set_p2p_timeout(double);
void mgxk_err_to_double(hipStream_t, const int *, int, double *);
void mgxk_halo_pack_all(hipStream_t, const LevView *, double *, double *const *, const int *, int);
void mgxk_convert(hipStream_t, const LevView *, double *, double *, int, int, int);
void mgxk_convert8(hipStream_t, const LevView *, const double *);
void mgxk_gather_place(hipStream_t, const LevView *, double *, const double *, int, int, int, int);
void mgxk_block_to_ref(hipStream_t, const LevView *, const double *, double *);
void mgxk_split(hipStream_t, const LevView *, const LevView *, const double *, double *, int, int);
void mgxk_gather_push(hipStream_t, const LevView *, const double *, double *const *, unsigned long long *const *, int, int, unsigned long long,
                      unsigned int *, int *);
void mgxk_gather_place_wait(hipStream_t, const LevView *, double *, const double *, int, int, int, int, unsigned long long *, unsigned long long,
                            int *);
void mgxs_coarsen2d(hipStream_t, const double *, double *, int, int, int, double);
void mgxs_rect(hipStream_t, double *, double *, const RectOp *);
void mgxs_halo_ref_closed(hipStream_t, double *, int, int, int, int);
void mgxs_zr_zw(hipStream_t, const GeoView *, double, double, double);
void mgxs_define_matrix(hipStream_t, const GeoView *, int lev1, int phase);
void mgxs_pivots(hipStream_t, const LevView *);
void mgxs_slopes_ref(hipStream_t, const GeoView *);
void mgxk_convert2(hipStream_t, const LevView *, double *, double *, const double *);
void mgxs_zw_js(hipStream_t, const GeoView *, const LevView *, double, double, double);
void mgxm_ref2model(hipStream_t, const double *, double *, int rows, int nh, int nx, int ny);
void mgxm_ref2model_2d(hipStream_t, const double *, double *, int nx, int ny);
void mgxm_js_model(hipStream_t, const LevView *, double *js, double *md, int dir);
void mgxm_rhs_uf(hipStream_t, const GeoView *, const ModelView *, double *);
void mgxm_rhs_vf(hipStream_t, const GeoView *, const ModelView *, double *);
void mgxm_rhs_wf(hipStream_t, const GeoView *, const ModelView *, double *);
void mgxm_flux_zero_face(hipStream_t, const GeoView *, double *, int face, int pl);
void mgxm_flux_face_copy(hipStream_t, const GeoView *, double *, double *, int face, int pl, int unpack);
void mgxm_rhs_accum(hipStream_t, const GeoView *, double *, const double *, const double *, const double *);
void mgxm_correct_uvw(hipStream_t, const GeoView *, const double *, const ModelView *);
// native RCCL transport (mgx_rccl.cpp)
const char *mgxr_last_error(void);
const char *mgxr_library(void);
int mgxr_connected(void);
int mgxr_nranks(void);
int mgxr_get_unique_id(void *);
int mgxr_connect(const void *, int, int);
void mgxr_disconnect(void);
int mgxr_exchange(hipStream_t, int, const int *, double *const *, double *const *, const int *);
int mgxr_allreduce(hipStream_t, double *, int);
int mgxr_allgather(hipStream_t, const int *, int, const double *, double *, int);
}

// a HIP error that was pending when a kernel wrapper started (mgx_before_launch, mgx_device.h): reported by the next synchronising call
thread_local hipError_t mgx_pending_error = hipSuccess;

namespace {

enum { M_GS = 0, M_RB = 1, M_FC = 2 };
inline bool all_physical(const Sides &s) { return s.S && s.E && s.N && s.W; }
inline bool any_physical(const Sides &s) { return s.S || s.E || s.N || s.W; }

struct Level {
  int nx, ny, nz, npx, npy, incx, incy, gather, ngx, ngy, key, color;
  int neighb[8];
  LevView v;    // solver fields (JS)
  LevView vs;   // pre-gather / split block (gathered levels): vs.b = restricted block, vs.p = split block
  GeoView g;    // set-up arrays (reference layout)
  double *tmp2[4];  // pre-gather coarse dx,dy,zeta,h
  double *blk, *gbuf;  // all-gather send / receive (reference layout blocks incl. halo)
  int group[4], ngroup;
  size_t n3js;  // doubles in one JS array
  bool r_halo_stale = false, b_halo_stale = false;  // deferred neighbour exchanges (multi-rank)
  size_t p2p_off[8][2];         // doubles into the receive slab: direction x parity
  unsigned long long p2p_seq = 0;  // exchanges done on this level through the peer-to-peer transport
  size_t p2p_goff[2];           // gathered levels: ngroup blocks of the peer-to-peer gather, by parity
  unsigned long long p2p_gseq = 0;
  unsigned int *ksp_done = nullptr; unsigned int ksp_seq = 0;  // per-plane progress counters of the persistent mid-level relax (k_relax_ksp) and their common value
  unsigned int *rbs_flag = nullptr; unsigned int rbs_seq = 0;  // progress word of the sequential-order red-black walk and the number of its launches (mgx_rbseq.hip: k_rbseq_scan, FUSE)
  double *gdec = nullptr; std::vector<double> gdec_h; int rbs_rows = 0;  // per row the largest |g(k) / g(1)| of the level (k_rbseq_gdecay) and the rows the correction reaches (mgxk_rbseq_window_rows)
  double rbs_rho = -1.0; int rbs_m = 0;  // sequential-order red-black, windowed walk (k_rbseq_window): rho = max |ag5| + |ag8| of the level, found at set-up, and the planes of warm-up it asks for (0 = none: the walk over the whole level)
  double *p1b = nullptr;        // second k=1 snapshot buffer (red-black on closed levels: one snapshot launch per relax call)
  double *zy_store, *zx_store;  // slope arrays; v.zy/v.zx point here while the matrix is the one define_matrices built
  double *f2d_store[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, *tab_store[2] = {nullptr, nullptr};  // m4,d4,m7,d7,h2,hi2,ze2 and cffw,csw (LevView)
  double *zg_store[4] = {nullptr, nullptr, nullptr, nullptr};  // dx2,dy2,cffr,csr (LevView)
};

struct TicRec { int lev, sub; hipEvent_t e0, e1; };
struct HostTic { int lev, sub; std::chrono::steady_clock::time_point t0; };  // a section the caller opened with mgx_tic

struct State {
  bool inited = false, have_matrix = false;
  mgx_params par;
  int method = M_RB, real = 1, linear = 1;
  int nlevs = 0, npx = 1, npy = 1, nranks = 1, rank = 0, pi = 0, pj = 0;
  std::vector<Level> lev;
  double hlim = 0, theta_b = 0, theta_s = 0;
  hipStream_t stream = nullptr;
  mgx_exchange_fn ex = nullptr; mgx_allreduce_fn ar = nullptr; mgx_allgather_fn ag = nullptr; void *ctx = nullptr;
  bool native_rccl = false;  // the hooks are the library's own RCCL transport (mgx_rccl_connect)
  double *d_partial = nullptr; int npartial = 0;
  double *d_scalar = nullptr; double *h_scalar = nullptr;
  double *ref_scratch = nullptr; size_t ref_scratch_n = 0;  // reference-layout staging (8 x level-1 field)
  double *slope_scratch = nullptr;                          // zy, zx of the level in work (2 x level-1 field), mgx_setup.hip
  double *xbuf[16]; size_t xbuf_n = 0;                       // 8 send + 8 receive halo buffers
  // peer-to-peer halo transport (mgx_p2p_prepare / mgx_p2p_connect): receive slab + flags in fine-grained device memory,
  // the same slab and flags of every other rank opened through hipIpc
  bool p2p_ready = false, p2p_on = false, p2p_borrowed = false;  // borrowed: peers are plain pointers (mgx_p2p_connect_pointers)
  double *p2p_slab = nullptr; size_t p2p_slab_n = 0;
  unsigned long long *p2p_flags = nullptr;
  std::vector<double *> peer_slab; std::vector<unsigned long long *> peer_flags;
  unsigned int *p2p_counter = nullptr;
  int *p2p_err = nullptr;   // host-mapped
  int *kerr = nullptr;      // host-mapped error word of the persistent relax kernel (a plane's neighbour never showed up)
  long long n_p2p = 0;
  int p2p_failed = 0;       // a wait of this rank timed out since the ranks last agreed (global_sum): reported collectively there
  int p2p_test_drop = 0;    // test hook (option "p2p_test_drop" = n): the n-th halo exchange from now does not raise its flags
  double *d_u = nullptr, *d_v = nullptr, *d_w = nullptr, *d_fx = nullptr, *d_fy = nullptr, *d_fz = nullptr, *d_bm = nullptr;  // model-layout scratch: the three fluxes of compute_rhs, divergence / pressure
  // the mask handed to nhydro_solve / nhydro_check_nondivergence on THIS call (nhydro.f90:72,82,98): staging copy in the
  // caller's layout and the i-fastest copy the model-space kernels read; call_mask = a mask came with the current call
  double *d_rmask_ref = nullptr, *d_rmask_m = nullptr; bool call_mask = false;
  std::vector<void *> allocs;
  int verbose = 1;
  int warm_start = 0;   // keep p between solves instead of the reference's cold start (mg_solvers.f90:35)
  int tictoc = 0;       // per-(level,name) GPU timers in the shape of mg_tictoc.f90
  int rb_chain = 1;     // red-black: chained k=1 snapshots on closed levels (0 = one snapshot launch per colour pass, for A/B tests)
  int keep_r = 0;       // cycles also store the interpolated correction in the fine r (dead state of the reference's coarse2fine)
  int rb_seq = 1;       // red-black with cmatrix='real' in the reference's sequential order by the parallel pass + a scan over the planes of the k=1 couplings + a rank-one correction per column (mgx_rbseq.hip): within a few ulp of mg_relax.f90:170-186, the DEFAULT; 0 = the plain parallel pass (old same-colour diagonals everywhere, 1e-4 per sweep away)
  int rb_exact = 0;     // red-black with cmatrix='real' in the reference's SEQUENTIAL order (plane after plane): bit-identical to mg_relax.f90:170-186, slow
  int exact_halos = 0;  // MGX_EXACT_HALOS=1: exchange r and b halos eagerly like the reference
  int no_mf = 0;      // MGX_NO_MF=1: always use the stored slots 3,5,6,8 (A/B tests)
  int use_small = 1;  // one-launch relax on small levels (MGX_NO_SMALL=1 disables, for A/B tests)
  int ksp_test_stall = 0;  // test hook (option "ksp_test_stall" = i): in the next persistent relax the workgroup of plane i returns at once
  int use_fuse = 1;   // option "fuse_tail" / MGX_NO_WAVE_FUSE=1: coarse2fine / residual+restriction folded into the one-workgroup relax of the level below the coarsest (A/B)
  int async_ops = 0;  // option "async": mgx_vcycle / mgx_fcycle / mgx_relax / mgx_fine2coarse / mgx_coarse2fine return without waiting for the stream
  int use_ksp = 1;    // option "ksp" / MGX_NO_KSP=1: one launch per colour pair instead of the persistent relax kernel (A/B)
  int ksp_down = 0;   // the persistent relax kernel timed out in this solver (its workgroups were not all resident): off until the next mgx_init
  // halo exchange beside the interior sweep (relax(), four colours on a level with neighbours, pushes on): a second stream carries the
  // boundary part of a colour pass and the exchange behind it while the solver's stream sweeps the interior
  hipStream_t stream2 = nullptr; hipEvent_t ev_a = nullptr, ev_s = nullptr, ev_x = nullptr;
  // OFF by default.  Measured (profiles/r04_overlap_2ranks.txt: two ranks of 512x512x64 sharing the one GPU of a test box): 8.4 ms per V-cycle with
  // it, 4.7 ms without.  A colour pass of such a block is ONE 512-register wave per SIMD for its whole duration: an exchange wave (or the boundary
  // part's) on a SIMD keeps the interior part's wave off it, and the boundary part alone takes as long as a whole pass (every wave runs the full
  // ~50 us), so the chain exchange -> boundary part -> exchange is no shorter than the serial one; the two cross-stream waits per colour come on top.
  int overlap = 0;       // option "overlap" / MGX_OVERLAP=1 (the same bits either way)
  long long n_overlap = 0;  // colour passes run that way
  int rbseq_fuse_min = 4 << 20;  // option "rbseq_fuse_min": cells of a colour (nx * ny/2 * nz) from which on the fused launch is used (below, the hand-off costs more than the correction's own launch: 256x256x32 0.111 ms per sweep fused, 0.099 separate)
  int rbseq_d0_in_pass = 1;  // option "rbseq_d0_in_pass" (A/B): 0 = k_rbseq_d0 as a launch of its own
  int rbseq_test_stall = 0;  // test hook: the walk of the fused launch never reports its progress (the bounded waits must end the launch)
  // option "coarsest_direct": the coarsest-level solve of a cycle (ns_coarsest sweeps from p = 0: a fixed linear map of b) as one matrix-vector product with the
  // operator the level's own relax kernel builds from the unit vectors when the matrix changes (mgx_relax_coarse.hip: k_coarse_direct).  The same map in
  // another association (1e-15 of max|p|), so: 1 (default) = only where the iteration is tolerance-based anyway (red-black in the sequential order at speed),
  // 2 = every method (four colours then lose their bit parity with the reference's loop), 0 = never
  int coarsest_direct = 1;
  double *cd_pb = nullptr, *cd_M = nullptr, *cd_part = nullptr; unsigned int *cd_cnt = nullptr;
  int cd_n = 0, cd_valid = 0, cd_method = -1, cd_mode = -1, cd_nsweeps = -1;   // cd_n: -1 = the level has no instance
  long long n_direct = 0;   // coarsest solves done that way
  int rbseq_rowcut = 1;  // option "rbseq_rowcut" (A/B): the windowed walk's correction stops at the last row it reaches to 2^-64 (Level::rbs_rows); 0 = every row
  int rbseq_window = 1;  // option "rbseq_window" / MGX_NO_RBSEQ_WINDOW=1: walk and correction of a colour by the windowed walk (k_rbseq_window: no hand-off, no walk over the whole level) on the levels whose contraction bound allows it (Level::rbs_m)
  long long n_window = 0;  // colours done that way
  double *rho_dev = nullptr, rho_host[32];  // the levels' rho (k_rbseq_rho) on the device and after the set-up's copy
  int rbseq_fuse = 1;    // option "rbseq_fuse" / MGX_NO_RBSEQ_FUSE=1: the correction of the sequential-order red-black inside the walk's launch (k_rbseq_scan, FUSE) instead of a launch behind it (A/B)
  int use_chain = 1;     // option "restrict_chain" / MGX_NO_RESTRICT_CHAIN=1: Fcycle's first-leg restrictions below level 1 as one launch (A/B)
  int fuse_closing = 1;  // option "fuse_closing" / MGX_NO_FUSE_CLOSING=1: the closing compute_residual(1) of a solve_p iteration also restricts its r for the next Fcycle, one kernel, no r written (A/B)
  int c2f_skip = 1;   // the cycles' prolongation leaves the columns alone that the first colour of the following four-colour relax overwrites unread (option "c2f_skip", MGX_C2F_NOSKIP=1)
  long long n_launch = 0, n_halo = 0, n_exch = 0, n_allred = 0;
  std::string err, transport_name;
  // mg_tictoc.f90's module variables (subname, time, calls, nblev) + the HIP events still in flight
  std::vector<std::string> tt_names;
  std::vector<TicRec> tt_open, tt_done;
  std::vector<HostTic> tt_host;
  double tt_time[32][32] = {};
  long long tt_calls[32][32] = {};
  int tt_nblev = 0;
};

// Instances.  The reference keeps ONE solver per process in module-global state (grid(:), mg_grids.f90:113-117), and so does every
// caller that never asks for more: instance 0 exists from the start and every thread acts on it.  A thread may select another
// instance (mgx_instance_create / mgx_instance_select): all mgx_* calls of THAT thread then act on it.  Used to couple several
// domains from one process and to run several ranks of one job as threads of one process (tests: BASELINE config 5's 4x2 grid on
// the one GPU of a test box, which admits fewer processes than that).
State S0;
std::vector<State *> g_instances = {&S0};
std::mutex g_instances_mu;
thread_local State *Sp = &S0;
#define S (*Sp)
int sync_stream();
void p2p_release();
// solvers that hold device state right now: the persistent relax kernel needs all its workgroups resident together, which nothing
// guarantees once several instances (thread-ranks, coupled domains) put kernels on the same device
int live_instances() {
  std::lock_guard<std::mutex> lk(g_instances_mu);
  int n = 0;
  for (State *q : g_instances) if (q && q->inited) n++;
  return n;
}

int fail(const char *fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  S.err = buf;
  if (S.verbose) fprintf(stderr, "mgx error: %s\n", buf);
  return 1;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define CHK(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)
#define NEED_INIT() do { if (!S.inited) return fail("mgx_init has not been called"); } while (0)
#define NEED_LEV(l) do { NEED_INIT(); if ((l) < 1 || (l) > S.nlevs) return fail("level %d out of range 1..%d", (l), S.nlevs); } while (0)

int dmalloc(double **p, size_t n) {
  void *q = nullptr;
  HIPCHK(hipMalloc(&q, (n ? n : 1) * sizeof(double)));
  HIPCHK(hipMemsetAsync(q, 0, (n ? n : 1) * sizeof(double), S.stream));
  S.allocs.push_back(q);
  *p = (double *)q;
  return 0;
}

int roundup(int a, int m) { return (a + m - 1) / m * m; }

void make_view(LevView &v, int nx, int ny, int nz) {
  v.nx = nx; v.ny = ny; v.nz = nz;
  v.EO = 15;
  v.HO = roundup(16 + ny / 2, 16);
  v.RS = roundup(v.HO + ny / 2 + 1, 16);
  v.plane = (long long)nz * v.RS;
}

// ---- mg_grids.f90:468-738 -------------------------------------------------------------------------
int find_grid_levels(int npxg, int npyg, int nx, int ny, int nz) {
  const int nxg = npxg * nx, nyg = npyg * ny, nzg = nz, ncoarsest = 4, nzmin = 2;
  const int nhoriz = nxg < nyg ? nxg : nyg;
  const int nl1 = 1 + (int)floor(log(nhoriz * 1.0 / ncoarsest * 1.0) / log(2.0));
  const int nl2 = 1 + (int)floor(log(nzg * 1.0 / nzmin * 1.0) / log(2.0));
  return nl1 < nl2 ? nl1 : nl2;
}

// level table of an arbitrary rank (needed to form gather groups without communication)
void rank_level_table(int rank, std::vector<Level> &T, int npx0, int npy0, int nsmall) {
  const int pi = rank % npx0, pj = rank / npx0;
  int nx = T[0].nx, ny = T[0].ny, nz = T[0].nz, npx = npx0, npy = npy0, incx = 1, incy = 1;
  T[0].npx = npx; T[0].npy = npy; T[0].incx = 1; T[0].incy = 1; T[0].gather = 0; T[0].ngx = 1; T[0].ngy = 1; T[0].key = 0; T[0].color = 0;
  for (int l = 1; l < (int)T.size(); l++) {  // define_grid_dims :503-577
    Level &L = T[l];
    if (nz == 1) { nx /= 2; ny /= 2; } else { nx /= 2; ny /= 2; nz /= 2; }
    L.gather = 0; L.ngx = 1; L.ngy = 1; L.key = 0; L.color = 0;
    if (((nx < ny ? nx : ny) < nsmall) && (npx * npy > 1)) {
      L.gather = 1;
      if (npx > 1) { npx /= 2; nx *= 2; L.ngx = 2; }
      if (npy > 1) { npy /= 2; ny *= 2; L.ngy = 2; }
      incx *= 2; incy *= 2;
    }
    L.nx = nx; L.ny = ny; L.nz = nz; L.npx = npx; L.npy = npy; L.incx = incx; L.incy = incy;
  }
  for (auto &L : T) {  // define_neighbours :580-661
    const int ix = L.incx, iy = L.incy;
    L.neighb[0] = (pj >= iy) ? (pj - iy) * npx0 + pi : -1;
    L.neighb[1] = (pi < npx0 - ix) ? pj * npx0 + pi + ix : -1;
    L.neighb[2] = (pj < npy0 - iy) ? (pj + iy) * npx0 + pi : -1;
    L.neighb[3] = (pi >= ix) ? pj * npx0 + pi - ix : -1;
    L.neighb[4] = (pj >= iy && pi >= ix) ? (pj - iy) * npx0 + pi - ix : -1;
    L.neighb[5] = (pj >= iy && pi < npx0 - ix) ? (pj - iy) * npx0 + pi + ix : -1;
    L.neighb[6] = (pj < npy0 - iy && pi < npx0 - ix) ? (pj + iy) * npx0 + pi + ix : -1;
    L.neighb[7] = (pj < npy0 - iy && pi >= ix) ? (pj + iy) * npx0 + pi - ix : -1;
  }
  for (int l = 1; l < (int)T.size(); l++) {  // define_gather_informations :664-738
    Level &L = T[l];
    if (!L.gather) continue;
    const int ix = L.incx / 2, iy = L.incy / 2;
    const int family = (pi / ix) * ix * iy + npx0 * iy * (pj / iy);
    const int nextfamily = (pi / (2 * ix)) * ix * iy * 4 + npx0 * 2 * iy * (pj / (iy * 2));
    L.color = nextfamily + (pi % ix) + (pj % iy) * ix;
    const int N = ix * npx0;
    L.key = ((family % N) / (ix * iy)) % 2 + 2 * ((family / N) % 2);
  }
}

// ---- halo exchange buffers ---------------------------------------------------------------------------
int exchange(int n, const int *peer, double *const *sb, double *const *rb, const int *cnt) {
  if (!S.ex) return fail("a halo exchange is needed (npx*npy > 1) but mgx_set_comm was not called");
  S.n_exch++;
  if (S.ex(S.ctx, n, peer, sb, rb, cnt)) return fail("exchange callback failed%s%s", S.native_rccl ? ": " : "", S.native_rccl ? mgxr_last_error() : "");
  return 0;
}

// the three hooks of mgx_set_comm served by the library's own RCCL communicator (mgx_rccl_connect), on the solver's stream
int rccl_exchange_hook(void *, int n, const int *peer, double *const *sb, double *const *rb, const int *cnt) { return mgxr_exchange(S.stream, n, peer, sb, rb, cnt); }
int rccl_allreduce_hook(void *, double *buf, int n) { return mgxr_allreduce(S.stream, buf, n); }
int rccl_allgather_hook(void *, const int *group, int ng, const double *sb, double *rb, int cnt) { return mgxr_allgather(S.stream, group, ng, sb, rb, cnt); }

// fill_halo_3D_relax / fill_halo_3D for the JS fields p,b,r (nh = 1): mg_mpi_exchange.f90:396-745
// xonly: fill_halo_4D's rule (mg_mpi_exchange.f90:1247-1534): nothing but the exchange with existing neighbours
int fill_halo_js(Level &L, double *a, bool phys_done = false, bool xonly = false) {
  S.n_halo++;
  const int *nb = L.neighb;
  Sides ph = {nb[0] < 0, nb[1] < 0, nb[2] < 0, nb[3] < 0};
  if (!phys_done && any_physical(ph)) { mgxk_halo_phys(S.stream, &L.v, a, ph); S.n_launch++; }
  int n = 0, peer[8], cnt[8], present[8];
  double *sb[8], *rb[8];
  for (int d = 0; d < 8; d++) {
    present[d] = nb[d] >= 0;
    if (nb[d] < 0) continue;
    const int c = L.nz * ((d == 0 || d == 2) ? L.nx : ((d == 1 || d == 3) ? L.ny : 1));
    peer[n] = nb[d]; cnt[n] = c; sb[n] = S.xbuf[d]; rb[n] = S.xbuf[8 + d]; n++;
  }
  int m[4] = {0, 0, 0, 0};  // mixed corners SW,SE,NE,NW: 1 = copy across the physical W/E side, 2 = across S/N (:720-743)
  bool any = false;
  if (n) {
    const int side1[4] = {0, 0, 2, 2}, side2[4] = {3, 1, 1, 3};  // SW:(S,W) SE:(S,E) NE:(N,E) NW:(N,W)
    for (int c = 0; c < 4; c++) {
      if (nb[4 + c] < 0) { if (nb[side1[c]] >= 0) m[c] = 1; else if (nb[side2[c]] >= 0) m[c] = 2; }
      if (xonly) m[c] = 0;
      any |= m[c] != 0;
    }
  }
  if (n && S.p2p_on) {  // push into the neighbours' receive buffers over xGMI, then wait on the local flags: no host step
    static const int opp[8] = {2, 3, 0, 1, 6, 7, 4, 5};
    const unsigned long long seq = ++L.p2p_seq;
    const int par = (int)(seq & 1), li = (int)(&L - &S.lev[0]);
    double *rbuf[8], *lbuf[8];
    unsigned long long *rflag[8], *lflag[8];
    for (int d = 0; d < 8; d++) {
      rbuf[d] = lbuf[d] = nullptr; rflag[d] = lflag[d] = nullptr;
      if (nb[d] < 0) continue;
      rbuf[d] = S.peer_slab[nb[d]] + L.p2p_off[opp[d]][par];
      rflag[d] = S.peer_flags[nb[d]] + (li * 8 + opp[d]) * 2 + par;
      lbuf[d] = S.p2p_slab + L.p2p_off[d][par];
      lflag[d] = S.p2p_flags + (li * 8 + d) * 2 + par;
    }
    int drop = 0;
    if (S.p2p_test_drop > 0 && --S.p2p_test_drop == 0) drop = 1;
    mgxk_halo_p2p(S.stream, &L.v, a, rbuf, lbuf, rflag, lflag, present, seq, S.p2p_counter, S.p2p_err, m, drop);  // push, wait, unpack, mixed corners
    S.n_launch++; S.n_p2p++;
  } else if (n) {
    mgxk_halo_pack_all(S.stream, &L.v, a, S.xbuf, present, 0); S.n_launch++;       // all edges + corners, one launch
    CHK(exchange(n, peer, sb, rb, cnt));
    mgxk_halo_pack_all(S.stream, &L.v, a, S.xbuf + 8, present, 1); S.n_launch++;
    if (any) { mgxk_halo_mixed_corners(S.stream, &L.v, a, m[0], m[1], m[2], m[3]); S.n_launch++; }
  }
  return 0;
}

// generic halo fill of a reference-layout array a(nzz,1-nh:ny+nh,1-nh:nx+nh); lbc = 0,'u','v'
// (mg_mpi_exchange.f90:23-352 2D, :750-1242 3D incl. nh=2 extrapolation and lbc_null)
void rect(double *a, double *buf, int op, int nzz, int nh, int ny, int j0, int j1, int i0, int i1, int mj = 0, int cj = 0, int mi = 0,
          int ci = 0, int mj2 = 0, int cj2 = 0, int mi2 = 0, int ci2 = 0) {
  RectOp R = {op, nzz, nh, ny, j0, j1, i0, i1, mj, cj, mi, ci, mj2, cj2, mi2, ci2};
  mgxs_rect(S.stream, a, buf, &R);
  S.n_launch++;
}

// xonly = fill_halo_4D (mg_mpi_exchange.f90:1245-1552): only the exchange with existing neighbours
int rl_fill_halo(Level &L, double *a, int nzz, int nh, char c, bool xonly = false) {
  S.n_halo++;
  const int nx = L.nx, ny = L.ny;
  const int *nb = L.neighb;
  const int So = nb[0], E = nb[1], N = nb[2], W = nb[3], SW = nb[4], SE = nb[5], NE = nb[6], NW = nb[7];
  const bool zSW = (c == 'u' && W < 0), zSE = (c == 'u' && E < 0), zNE = (c == 'u' && E < 0) || c == 'v', zNW = (c == 'u' && W < 0) || c == 'v';
  if (!xonly && c == 0 && So < 0 && E < 0 && N < 0 && W < 0 && (nh == 1 || nh == 2) && nx >= 2 && ny >= 2) {
    // no neighbour at all: every halo cell is an image (or, nh = 2, an extrapolation) of interior cells -- one launch for all sides and corners
    mgxs_halo_ref_closed(S.stream, a, nzz, nh, ny, nx); S.n_launch++;
    return 0;
  }
  if (!xonly) {
  // phase 1: physical sides, in the reference's order S,E,N,W then the corners
  if (So < 0) {
    if (c == 'v') rect(a, 0, 2, nzz, nh, ny, 1, 1, 1 - nh, nx + nh);
    else { rect(a, 0, 0, nzz, nh, ny, 0, 0, 1, nx, 0, 1, 0, 0); if (nh == 2) rect(a, 0, 1, nzz, nh, ny, -1, -1, 1, nx, 0, 2, 0, 0, 0, 3, 0, 0); }
  }
  if (E < 0) {
    if (c == 'u') rect(a, 0, 2, nzz, nh, ny, 1 - nh, ny + nh, nx + 1, nx + 1);
    else { rect(a, 0, 0, nzz, nh, ny, 1, ny, nx + 1, nx + 1, 0, 0, 0, -1); if (nh == 2) rect(a, 0, 1, nzz, nh, ny, 1, ny, nx + 2, nx + 2, 0, 0, 0, -2, 0, 0, 0, -3); }
  }
  if (N < 0) {
    if (c == 'v') rect(a, 0, 2, nzz, nh, ny, ny + 1, ny + 1, 1 - nh, nx + nh);
    else { rect(a, 0, 0, nzz, nh, ny, ny + 1, ny + 1, 1, nx, 0, -1, 0, 0); if (nh == 2) rect(a, 0, 1, nzz, nh, ny, ny + 2, ny + 2, 1, nx, 0, -2, 0, 0, 0, -3, 0, 0); }
  }
  if (W < 0) {
    if (c == 'u') rect(a, 0, 2, nzz, nh, ny, 1 - nh, ny + nh, 1, 1);
    else { rect(a, 0, 0, nzz, nh, ny, 1, ny, 0, 0, 0, 0, 0, 1); if (nh == 2) rect(a, 0, 1, nzz, nh, ny, 1, ny, -1, -1, 0, 0, 0, 2, 0, 0, 0, 3); }
  }
  if (SW < 0) { if (zSW) rect(a, 0, 2, nzz, nh, ny, 1 - nh, 0, 1 - nh, 0); else if (So < 0 && W < 0) rect(a, 0, 0, nzz, nh, ny, 1 - nh, 0, 1 - nh, 0, 1, 1, 1, 1); }
  if (SE < 0) { if (zSE) rect(a, 0, 2, nzz, nh, ny, 1 - nh, 0, nx + 1, nx + nh); else if (So < 0 && E < 0) rect(a, 0, 0, nzz, nh, ny, 1 - nh, 0, nx + 1, nx + nh, 1, 1, 1, 2 * nx + 1); }
  if (NE < 0) { if (zNE) rect(a, 0, 2, nzz, nh, ny, ny + 1, ny + nh, nx + 1, nx + nh); else if (N < 0 && E < 0) rect(a, 0, 0, nzz, nh, ny, ny + 1, ny + nh, nx + 1, nx + nh, 1, 2 * ny + 1, 1, 2 * nx + 1); }
  if (NW < 0) { if (zNW) rect(a, 0, 2, nzz, nh, ny, ny + 1, ny + nh, 1 - nh, 0); else if (N < 0 && W < 0) rect(a, 0, 0, nzz, nh, ny, ny + 1, ny + nh, 1 - nh, 0, 1, 2 * ny + 1, 1, 1); }
  }
  // phase 2: exchange with the existing neighbours
  int n = 0, peer[8], cnt[8];
  double *sb[8], *rb[8];
  int rr[8][4];
  for (int d = 0; d < 8; d++) {
    if (nb[d] < 0) continue;
    int sj0, sj1, si0, si1, rj0, rj1, ri0, ri1;
    const bool south = (d == 0 || d == 4 || d == 5), north = (d == 2 || d == 6 || d == 7);
    const bool east = (d == 1 || d == 5 || d == 6), west = (d == 3 || d == 4 || d == 7);
    if (south) { sj0 = 1; sj1 = nh; rj0 = 1 - nh; rj1 = 0; } else if (north) { sj0 = ny - nh + 1; sj1 = ny; rj0 = ny + 1; rj1 = ny + nh; } else { sj0 = rj0 = 1; sj1 = rj1 = ny; }
    if (west) { si0 = 1; si1 = nh; ri0 = 1 - nh; ri1 = 0; } else if (east) { si0 = nx - nh + 1; si1 = nx; ri0 = nx + 1; ri1 = nx + nh; } else { si0 = ri0 = 1; si1 = ri1 = nx; }
    const int count = nzz * (sj1 - sj0 + 1) * (si1 - si0 + 1);
    if ((size_t)count > S.xbuf_n) return fail("halo buffer too small");
    rect(a, S.xbuf[d], 3, nzz, nh, ny, sj0, sj1, si0, si1);
    peer[n] = nb[d]; cnt[n] = count; sb[n] = S.xbuf[d]; rb[n] = S.xbuf[8 + d];
    rr[n][0] = rj0; rr[n][1] = rj1; rr[n][2] = ri0; rr[n][3] = ri1; n++;
  }
  if (n) {
    CHK(exchange(n, peer, sb, rb, cnt));
    for (int q = 0; q < n; q++) rect(a, rb[q], 4, nzz, nh, ny, rr[q][0], rr[q][1], rr[q][2], rr[q][3]);
  }
  if (xonly) return 0;
  // phase 3: mixed corners (:1216-1240)
  if (SW < 0 && !zSW) { if (So >= 0) rect(a, 0, 0, nzz, nh, ny, 1 - nh, 0, 1 - nh, 0, 0, 0, 1, 1); else if (W >= 0) rect(a, 0, 0, nzz, nh, ny, 1 - nh, 0, 1 - nh, 0, 1, 1, 0, 0); }
  if (SE < 0 && !zSE) { if (So >= 0) rect(a, 0, 0, nzz, nh, ny, 1 - nh, 0, nx + 1, nx + nh, 0, 0, 1, 2 * nx + 1); else if (E >= 0) rect(a, 0, 0, nzz, nh, ny, 1 - nh, 0, nx + 1, nx + nh, 1, 1, 0, 0); }
  if (NE < 0 && !zNE) { if (N >= 0) rect(a, 0, 0, nzz, nh, ny, ny + 1, ny + nh, nx + 1, nx + nh, 0, 0, 1, 2 * nx + 1); else if (E >= 0) rect(a, 0, 0, nzz, nh, ny, ny + 1, ny + nh, nx + 1, nx + nh, 1, 2 * ny + 1, 0, 0); }
  if (NW < 0 && !zNW) { if (N >= 0) rect(a, 0, 0, nzz, nh, ny, ny + 1, ny + nh, 1 - nh, 0, 0, 0, 1, 1); else if (W >= 0) rect(a, 0, 0, nzz, nh, ny, ny + 1, ny + nh, 1 - nh, 0, 1, 2 * ny + 1, 0, 0); }
  return 0;
}

// global_sum (mg_mpi_exchange.f90:1555-1571) of the value in d_scalar[0]; returns it on the host
// The all-reduce doubles as the point where the ranks AGREE on the health of the peer-to-peer transport: a second value carries
// "a wait of mine timed out" (the device-side error word, read in stream order, or a time-out an earlier sync saw).  If any rank
// says so, every rank switches the pushes off, rewinds its sequence numbers and flags, and returns the same error: nobody is
// left pushing to, or waiting for, a rank that fell back alone.
int global_sum(const Level &L, double *out) {
  // The count is the same on every rank of a multi-rank job whatever this rank's transport state (a rank whose hipIpc mapping failed keeps
  // running on the hooks while its neighbours may have connected: a count chosen from the rank-local p2p_ready would mismatch): always two
  // values, the second one 0 from a rank without pushes.
  const bool agree = S.nranks > 1;
  if (S.nranks > 1) {
    if (!S.ar) return fail("an all-reduce is needed (npx*npy > 1) but mgx_set_comm was not called");
    S.n_allred++;
    if (S.p2p_ready && S.p2p_err) { mgxk_err_to_double(S.stream, S.p2p_err, S.p2p_failed, S.d_scalar + 1); S.n_launch++; }
    else HIPCHK(hipMemsetAsync(S.d_scalar + 1, 0, sizeof(double), S.stream));
    if (S.ar(S.ctx, S.d_scalar, 2)) return fail("allreduce callback failed");
  }
  HIPCHK(hipMemcpyAsync(S.h_scalar, S.d_scalar, 2 * sizeof(double), hipMemcpyDeviceToHost, S.stream));
  CHK(sync_stream());
  if (agree && S.h_scalar[1] > 0.0) {
    S.p2p_on = false; S.p2p_failed = 0;
    if (S.p2p_err) *S.p2p_err = 0;
    for (auto &Lv : S.lev) { Lv.p2p_seq = 0; Lv.p2p_gseq = 0; }
    if (S.p2p_flags) HIPCHK(hipMemsetAsync(S.p2p_flags, 0, 4096 * sizeof(unsigned long long), S.stream));
    HIPCHK(hipStreamSynchronize(S.stream));
    return fail("the peer-to-peer halo transport timed out on %d rank(s): ALL ranks have switched to the hooks together (sequence numbers "
                "rewound); the halos of the affected exchanges were stale, so the current solve is void -- repeat it", (int)S.h_scalar[1]);
  }
  *out = S.h_scalar[0] * (L.npx * L.npy) / (S.lev[0].npx * S.lev[0].npy);
  return 0;
}


// ---- mg_tictoc.f90: tic(lev,name) / toc(lev,name) / print_tictoc, timed with HIP events on the solver's stream ----

int tt_sub(const char *name) {
  for (size_t q = 0; q < S.tt_names.size(); q++) if (S.tt_names[q] == name) return (int)q;
  S.tt_names.push_back(name);
  return (int)S.tt_names.size() - 1;
}
void tic(int lev, const char *name) {
  if (!S.tictoc) return;
  TicRec r; r.lev = lev; r.sub = tt_sub(name);
  if (r.sub >= 32 || lev > 32) return;
  (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, S.stream);
  S.tt_open.push_back(r);
}
void toc(int lev, const char *name) {
  if (!S.tictoc) return;
  const int sub = tt_sub(name);
  for (int q = (int)S.tt_open.size() - 1; q >= 0; q--)
    if (S.tt_open[q].lev == lev && S.tt_open[q].sub == sub) {
      (void)hipEventRecord(S.tt_open[q].e1, S.stream);
      S.tt_done.push_back(S.tt_open[q]);
      S.tt_open.erase(S.tt_open.begin() + q);
      if (lev > S.tt_nblev) S.tt_nblev = lev;
      return;
    }
}
void tt_collect() {
  if (S.tt_done.empty()) return;
  (void)hipStreamSynchronize(S.stream);
  for (auto &r : S.tt_done) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { S.tt_time[r.lev - 1][r.sub] += ms * 1e-3; S.tt_calls[r.lev - 1][r.sub]++; }
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
  }
  S.tt_done.clear();
}
struct TicScope { int lev; const char *name; TicScope(int l, const char *n) : lev(l), name(n) { tic(l, n); } ~TicScope() { toc(lev, name); } };

// ---- operators ------------------------------------------------------------------------------------
// mg_relax.f90:16-47 relax ; :151-190 RB ; :193-234 FC
int relax(int lev, int nsweeps) {
  Level &L = S.lev[lev - 1];
  TicScope ts(lev, S.method == M_RB ? "relax_3D_8_RB" : (S.method == M_FC ? "relax_3D_8_FC" : "relax_3D_8_GS"));  // mg_relax.f90:128,167,209
  if (S.tictoc && S.tt_done.size() > 4096) tt_collect();
  if (S.method == M_GS) {  // exact lexicographic order by hyperplanes; halo fill once per sweep (mg_relax.f90:131-141)
    for (int it = 1; it <= nsweeps; it++) {
      if (!mgxk_relax_gs_sweep(S.stream, &L.v, S.real)) return fail("relax_method='GS': nz=%d has no register-resident kernel (nz must be a power of two <= 64)", L.nz);
      S.n_launch += L.ny + 2 * L.nx - 2;
      CHK(fill_halo_js(L, L.v.p));
    }
    return 0;
  }
  const Sides ph = {L.neighb[0] < 0, L.neighb[1] < 0, L.neighb[2] < 0, L.neighb[3] < 0};
  const int exact = S.method == M_RB && S.real && S.rb_exact;
  // sequential-order red-black (mgx_rbseq.hip); the one-workgroup kernels of the small levels run the reference's plane loop itself
  const int seq = S.method == M_RB && S.real && S.rb_seq && !exact && L.v.gk != nullptr;
  if (S.use_small && nsweeps > 0 && mgxk_relax_small(S.stream, &L.v, nsweeps, S.method, S.real, ph, exact ? 1 : (seq ? 2 : 0))) { S.n_launch++; return 0; }
  const bool closed = all_physical(ph);
  // closed mid levels, four colours: the whole call in one persistent launch, one workgroup per plane (mgx_relax_ks.hip: k_relax_ksp)
  if (S.method == M_FC && closed && S.use_ksp && !S.ksp_down && live_instances() == 1 && mgxk_relax_ks_persist(S.stream, &L.v, nsweeps, S.real, ph, L.ksp_done, L.ksp_seq, S.kerr, S.ksp_test_stall)) {
    S.ksp_test_stall = 0;
    L.ksp_seq += (unsigned int)nsweeps; S.n_launch++;
    return 0;
  }
  double *const p1a = L.v.p1;
  for (int it = 1; it <= nsweeps; it++) {
    if (exact) {
      // The reference's red-black loop is sequential (mg_relax.f90:170-186): with cmatrix='real' a column of plane i reads the
      // same-colour k=1 diagonals (j+-1,i-1) already updated and (j+-1,i+1) not yet (:271-276).  Columns of one colour inside a
      // plane are independent, so one launch per plane, in order, reproduces the loop bit for bit -- on one rank and, with the
      // halo filled after each colour as in the reference, its decomposition-dependent result on several.
      for (int rb = 1; rb <= 2; rb++) {
        int fused = 0;
        for (int i = 1; i <= L.nx; i++) { fused = mgxk_relax_colour(S.stream, &L.v, i, 1, 1, -1, rb, 1, 0, ph); S.n_launch++; }
        CHK(fill_halo_js(L, L.v.p, fused));
      }
      continue;
    }
    if (S.method == M_RB) {
      // cmatrix='real': the k=1 diagonal neighbours have the column's own colour and must be read as they were before the
      // pass (snapshot).  On a closed level the register kernels write the next sweep's snapshot themselves (two buffers
      // swapped per sweep: a pass reads only entries of its own colour, which the other colour's pass never touches), so
      // one snapshot launch per relax call suffices; with neighbours the halo part changes after every exchange.
      const bool chain = S.rb_chain && S.real && closed && mgxk_has_reg_kernel(&L.v) && !seq;
      if (chain) {
        if (it == 1) { mgxk_snapshot_k1(S.stream, &L.v); S.n_launch++; }
        L.v.p1w = (L.v.p1 == p1a) ? L.p1b : p1a;
      }
      for (int rb = 1; rb <= 2; rb++) {
        // seq on a closed level: the correction keeps the snapshot current (its colour's new bottom values and their physical images), so one
        // snapshot launch per relax call; with neighbours the halo part changes with every exchange
        if (S.real && !chain && !(seq && closed && !(it == 1 && rb == 1))) { mgxk_snapshot_k1(S.stream, &L.v); S.n_launch++; }
        // (seq, wide half-rows: the pass also leaves the walk's d0 = y(k=1) - snapshot in u1 where its kernel can -- one launch less)
        L.v.d0w = (seq && S.rbseq_d0_in_pass && (mgxk_rbseq_wants_d0(&L.v) || (S.rbseq_window && L.rbs_m > 0))) ? L.v.u1 : nullptr;
        const int pass = mgxk_relax_colour(S.stream, &L.v, 1, 1, L.nx, -1, rb, S.real, S.real, ph); S.n_launch++;
        L.v.d0w = nullptr;
        int fused = pass & 1;
        const int have_d0 = (pass & 2) ? 1 : 0;
        if (seq) {
          // y is in p; the walk over the planes, then p += g s with the mirrors (mgx_rbseq.hip).  A level wider than the walk takes
          // (ny > 2048) would have to run plane by plane: refuse loudly rather than fall back to another iteration
          // small levels whose pass left d0 in u1: walk and correction in one launch, every workgroup walking for itself (k_rbseq_walk_apply)
          // the windowed walk where the level's contraction bound allows it (k_rbseq_window): one launch, no hand-off, no walk over the level
          if (S.rbseq_window && L.rbs_m > 0) {
            if (!have_d0) { mgxk_rbseq_d0(S.stream, &L.v, rb); S.n_launch++; }   // (the nz = 128 colour pass does not leave it)
          }
          if (S.rbseq_window && L.rbs_m > 0 && mgxk_rbseq_window(S.stream, &L.v, rb, ph, closed ? 1 : 0, L.rbs_m, S.rbseq_rowcut ? L.rbs_rows : L.nz)) {
            S.n_launch++; S.n_window++; fused = 1;
            CHK(fill_halo_js(L, L.v.p, fused));
            continue;
          }
          if (have_d0 && S.rbseq_fuse && mgxk_rbseq_walk_apply(S.stream, &L.v, rb, ph, closed ? 1 : 0)) {
            S.n_launch++; fused = 1;
            CHK(fill_halo_js(L, L.v.p, fused));
            continue;
          }
          // (where an instance exists the correction runs inside the walk's launch, chasing it: option "rbseq_fuse")
          const int ran = S.rbseq_fuse ? mgxk_rbseq_scan_apply(S.stream, &L.v, rb, ph, closed ? 1 : 0, have_d0, L.rbs_flag, ++L.rbs_seq, S.kerr, S.rbseq_test_stall, (long long)S.rbseq_fuse_min) : mgxk_rbseq_scan(S.stream, &L.v, rb, have_d0);
          if (ran == 2) S.rbseq_test_stall = 0;
          if (!ran) return fail("rb_seq: level %d (ny = %d) has no scan instance; set option rb_exact or rb_seq = 0", lev, L.ny);
          if (ran == 1) { mgxk_rbseq_apply(S.stream, &L.v, rb, ph, closed ? 1 : 0); S.n_launch++; }
          S.n_launch += 2 - have_d0;
          fused = 1;  // the correction stores the physical images of every column it updates
        }
        CHK(fill_halo_js(L, L.v.p, fused));
      }
      if (chain) { L.v.p1 = L.v.p1w; L.v.p1w = nullptr; if (it == nsweeps) L.v.p1 = p1a; }
    } else {
      // A level with neighbours, halos by the pushes: the boundary part of a colour (the waves that hold a column next to a neighbour's
      // halo -- what the exchange sends, and all that reads what the last exchange delivered) and the exchange behind it go to a second
      // stream; the interior part runs beside them on the solver's stream and waits only for the previous colour's boundary part
      // (mg_relax.f90:181,224 exchange after every colour; SURVEY 7 "split boundary columns from interior, exchange while the interior runs").
      const bool ov = S.overlap && S.p2p_on && S.stream2 && !closed && mgxk_has_reg_kernel(&L.v);
      for (int fc1 = 1; fc1 <= 2; fc1++) {
        // closed mid levels: the two colours of a plane set in one launch (mgx_relax_ks.hip)
        if (closed && mgxk_relax_ks_pair(S.stream, &L.v, fc1, L.nx / 2, S.real, ph)) { S.n_launch++; continue; }
        for (int fc2 = 1; fc2 <= 2; fc2++) {
          if (ov) {
            Sides ps = ph;
            HIPCHK(hipEventRecord(S.ev_a, S.stream));                 // the interior of the previous colour (and whatever came before)
            HIPCHK(hipStreamWaitEvent(S.stream2, S.ev_a, 0));
            ps.part = 1;
            const int fused = mgxk_relax_colour(S.stream2, &L.v, 1 + (fc1 - 1) % 2, 2, L.nx / 2, fc2 == 1 ? 1 : 0, 0, S.real, 0, ps);
            HIPCHK(hipEventRecord(S.ev_s, S.stream2));
            { hipStream_t keep = S.stream; S.stream = S.stream2; const int rc = fill_halo_js(L, L.v.p, fused); S.stream = keep; if (rc) return rc; }
            ps.part = 2;
            mgxk_relax_colour(S.stream, &L.v, 1 + (fc1 - 1) % 2, 2, L.nx / 2, fc2 == 1 ? 1 : 0, 0, S.real, 0, ps);
            HIPCHK(hipStreamWaitEvent(S.stream, S.ev_s, 0));          // what follows on the solver's stream reads this colour's boundary part -- not its exchange
            S.n_launch += 2; S.n_overlap++;
            continue;
          }
          const int fused = mgxk_relax_colour(S.stream, &L.v, 1 + (fc1 - 1) % 2, 2, L.nx / 2, fc2 == 1 ? 1 : 0, 0, S.real, 0, ph); S.n_launch++;
          CHK(fill_halo_js(L, L.v.p, fused));
        }
      }
      if (ov && it == nsweeps) {  // the call ends: the solver's stream continues behind the last exchange
        HIPCHK(hipEventRecord(S.ev_x, S.stream2));
        HIPCHK(hipStreamWaitEvent(S.stream, S.ev_x, 0));
      }
    }
  }
  return 0;
}

// mg_relax.f90:337-383 compute_residual.  res == nullptr: the caller discards the norm (mg_solvers.f90:140),
// so neither the reduction nor the all-reduce is issued.
int residual(int lev, double *res) {
  Level &L = S.lev[lev - 1];
  TicScope ts(lev, "residual_3D_8");  // mg_relax.f90:367
  const Sides ph = {L.neighb[0] < 0, L.neighb[1] < 0, L.neighb[2] < 0, L.neighb[3] < 0};
  mgxk_residual(S.stream, &L.v, S.d_partial, S.d_scalar, S.real, res != nullptr, ph); S.n_launch += res ? 2 : 1;
  // the kernel wrote the physical mirrors of r; the neighbour part of r's halo is never read by the cycle (restriction
  // uses interior cells only), so the exchange is deferred until someone asks for r (mgx_get_field / mgx_fill_halo)
  if (S.exact_halos) CHK(fill_halo_js(L, L.v.r, true)); else L.r_halo_stale = true;
  if (res) { double s; CHK(global_sum(L, &s)); *res = sqrt(s); }
  return 0;
}

// mg_intergrids.f90:16-72.  with_residual: the caller is the down leg of a V-cycle, which would call compute_residual(lev)
// right before and discards both the norm and r (mg_solvers.f90:138-142): residual and restriction then run as ONE kernel
// that never writes r (mgx_resrest.hip), when the level has its matrix-free slopes; otherwise the two kernels in sequence.
int fine2coarse(int lev, bool dup_r = false, bool with_residual = false) {
  Level &F = S.lev[lev - 1], &C = S.lev[lev];
  const Sides phc = {C.neighb[0] < 0, C.neighb[1] < 0, C.neighb[2] < 0, C.neighb[3] < 0}, none = {0, 0, 0, 0};
  bool fused = false;
  // returns true when the fused residual+restriction kernel took the job
  auto down = [&](const LevView *Cv, double *dst, Sides ph, double *zero) -> int {
    if (!with_residual) return 0;
    if (!S.exact_halos && !S.keep_r && !dup_r) {  // keep_r: the caller wants the reference's r, which the fused kernel never writes
      TicScope ts(lev, "residual_3D_8");
      if (mgxk_residual_restrict(S.stream, &F.v, Cv, dst, S.real, ph, zero)) { S.n_launch++; return 1; }
    }
    return residual(lev, nullptr) ? -1 : 0;
  };
  if (!C.gather) {
    // closed level: the kernel also zeroes p_c and, for Fcycle, duplicates b_c into r_c (whole arrays through the mirrors)
    fused = all_physical(phc);
    const int d = down(&C.v, C.v.b, phc, fused ? C.v.p : nullptr);
    if (d < 0) return 1;
    if (!d) { mgxk_fine2coarse(S.stream, &F.v, &C.v, C.v.b, phc, fused && dup_r ? C.v.r : nullptr, fused ? C.v.p : nullptr); S.n_launch++; }
  } else {
    const int d = down(&C.vs, C.vs.b, none, nullptr);
    if (d < 0) return 1;
    if (!d) { mgxk_fine2coarse(S.stream, &F.v, &C.vs, C.vs.b, none, nullptr, nullptr); S.n_launch++; }
    const int Ng = C.nz * (C.vs.ny + 2) * (C.vs.nx + 2);
    if (S.p2p_on) {  // gather_3D (mg_gather.f90:95-174) as pushes into the members' gather buffers
      const unsigned long long seq = ++C.p2p_gseq;
      const int par = (int)(seq & 1), li = lev;
      int me = -1;
      for (int q = 0; q < C.ngroup; q++) if (C.group[q] == S.rank) me = q;
      if (me < 0) return fail("gather: rank %d is not in its own group on level %d", S.rank, lev + 1);
      double *dst[4]; unsigned long long *rflag[4];
      for (int q = 0; q < C.ngroup; q++) {
        // my own copy stays in ordinary device memory (C.blk): stores to the fine-grained slab are uncached and slow
        dst[q] = q == me ? C.blk : S.peer_slab[C.group[q]] + C.p2p_goff[par] + (size_t)me * Ng;
        rflag[q] = S.peer_flags[C.group[q]] + 1024 + (li * 4 + me) * 2 + par;
      }
      mgxk_gather_push(S.stream, &C.vs, C.vs.b, dst, rflag, C.ngroup, me, seq, S.p2p_counter, S.p2p_err); S.n_launch++;
      for (int q = 0; q < C.ngroup; q++) {
        unsigned long long *lflag = q == me ? nullptr : S.p2p_flags + 1024 + (li * 4 + q) * 2 + par;
        mgxk_gather_place_wait(S.stream, &C.v, C.v.b, q == me ? C.blk : S.p2p_slab + C.p2p_goff[par] + (size_t)q * Ng, C.vs.nx, C.vs.ny, q % C.ngx, q / C.ngx, lflag, seq, S.p2p_err);
        S.n_launch++;
      }
      S.n_p2p++;
    } else {
      mgxk_block_to_ref(S.stream, &C.vs, C.vs.b, C.blk); S.n_launch++;
      if (!S.ag) return fail("a gather is needed but mgx_set_comm was not called");
      if (S.ag(S.ctx, C.group, C.ngroup, C.blk, C.gbuf, Ng)) return fail("allgather callback failed");
      for (int q = 0; q < C.ngroup; q++) {
        mgxk_gather_place(S.stream, &C.v, C.v.b, C.gbuf + (size_t)q * Ng, C.vs.nx, C.vs.ny, q % C.ngx, q / C.ngx); S.n_launch++;
      }
    }
  }
  // b's halo is not read by relax/residual either: physical mirrors are in place, neighbour exchange deferred
  if (S.exact_halos || C.gather) CHK(fill_halo_js(C, C.v.b, !C.gather)); else C.b_halo_stale = true;
  if (!fused) {
    HIPCHK(hipMemsetAsync(C.v.p, 0, C.n3js * sizeof(double), S.stream));
    if (dup_r) HIPCHK(hipMemcpyAsync(C.v.r, C.v.b, C.n3js * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
  }
  return 0;
}

// mg_intergrids.f90:167-228.  keep_r: also leave the interpolated correction in the fine r, as the reference does (the C-ABI operator
// and exact_halos = 1); the cycles do not -- nothing reads it before compute_residual overwrites it.
// skip1: a four-colour relax(lev, n >= 1) follows immediately -- its first colour overwrites the (i odd, j odd) columns without reading
// them, so the prolongation leaves them alone (never together with keep_r).
int coarse2fine(int lev, bool keep_r = true, bool skip1 = false) {
  Level &F = S.lev[lev - 1], &C = S.lev[lev];
  const Sides phf = {F.neighb[0] < 0, F.neighb[1] < 0, F.neighb[2] < 0, F.neighb[3] < 0};
  if (!C.gather) {
    mgxk_coarse2fine(S.stream, &F.v, &C.v, C.v.p, S.linear, phf, keep_r, skip1 && !keep_r); S.n_launch++;
  } else {
    mgxk_split(S.stream, &C.v, &C.vs, C.v.p, C.vs.p, C.key % 2, C.key / 2); S.n_launch++;
    mgxk_coarse2fine(S.stream, &F.v, &C.vs, C.vs.p, S.linear, phf, keep_r, skip1 && !keep_r); S.n_launch++;
  }
  if (S.exact_halos && keep_r) CHK(fill_halo_js(F, F.v.r, true)); else F.r_halo_stale = true;
  // p = p + r over the whole array: the interior was updated by the kernel; the halo of p + halo of r
  // equals the halo fill of the updated p (both are images of the same interior cells)
  CHK(fill_halo_js(F, F.v.p, true));
  return 0;
}

// relax(lev, nsweeps) of the level below the coarsest one (a closed level the one-workgroup kernel serves), with coarse2fine(lev) folded
// in front (flags & 1) and / or compute_residual(lev) + fine2coarse(lev) folded behind (flags & 2): mgx_relax_coarse.hip.  Returns 1 when
// the fused kernel took the job (same bits as the separate operators), 0 = run them.
int relax_fused(int lev, int nsweeps, int flags) {
  if (lev >= S.nlevs || !S.use_small || !S.use_fuse || S.method == M_GS || S.tictoc || S.keep_r || S.exact_halos || !S.linear) return 0;
  if (S.method == M_RB && S.real && S.rb_exact) return 0;
  Level &F = S.lev[lev - 1], &C = S.lev[lev];
  const int mode = (S.method == M_RB && S.real && S.rb_seq && F.v.gk != nullptr) ? 2 : 0;
  const Sides phf = {F.neighb[0] < 0, F.neighb[1] < 0, F.neighb[2] < 0, F.neighb[3] < 0}, phc = {C.neighb[0] < 0, C.neighb[1] < 0, C.neighb[2] < 0, C.neighb[3] < 0};
  if (!all_physical(phf) || !all_physical(phc) || C.gather) return 0;
  if (!mgxk_relax_wave_fused(S.stream, &F.v, &C.v, nsweeps, S.method, S.real, phf, flags, mode)) return 0;
  S.n_launch++;
  if (flags & 1) F.r_halo_stale = true;  // what coarse2fine leaves (the correction is not stored in r inside a cycle)
  return 1;
}

// relax(nlevs, ns_coarsest) inside a cycle (mg_solvers.f90:117,144), where the coarsest level is entered with p = 0 (fine2coarse, mg_intergrids.f90:70):
// where option "coarsest_direct" allows it, one matrix-vector product with the operator the level's relax kernel built (mgx_relax_coarse.hip)
// p_zero: the caller has just restricted onto the level (Vcycle(nlevs) called as an operator relaxes whatever p it finds: the sweeps)
int coarsest_solve(bool p_zero) {
  Level &L = S.lev[S.nlevs - 1];
  const Sides ph = {L.neighb[0] < 0, L.neighb[1] < 0, L.neighb[2] < 0, L.neighb[3] < 0};
  const int exact = S.method == M_RB && S.real && S.rb_exact, seq = S.method == M_RB && S.real && S.rb_seq && !exact && L.v.gk != nullptr;
  const bool want = S.coarsest_direct == 2 || (S.coarsest_direct == 1 && seq);
  if (want && p_zero && S.nlevs >= 2 && S.use_small && !S.tictoc && S.method != M_GS && !exact && all_physical(ph) && !L.gather && S.par.ns_coarsest >= 1 && S.cd_n >= 0) {
    const int n = mgxk_coarse_direct_cells(&L.v), mode = seq ? 2 : 0;
    if (n > 0) {
      if (!S.cd_M) {
        CHK(dmalloc(&S.cd_pb, (size_t)2 * n * L.n3js)); CHK(dmalloc(&S.cd_M, (size_t)n * n));
        CHK(dmalloc(&S.cd_part, (size_t)mgxk_coarse_direct_slabs(n) * n));
        { double *q = nullptr; CHK(dmalloc(&q, 512)); S.cd_cnt = (unsigned int *)q; }   // one word per 64 rows, 64 bytes apart (zeroed by dmalloc)
        S.cd_n = n;
      }
      if (!S.cd_valid || S.cd_method != S.method || S.cd_mode != mode || S.cd_nsweeps != S.par.ns_coarsest) {
        if (mgxk_coarse_direct_build(S.stream, &L.v, S.par.ns_coarsest, S.method, S.real, ph, mode, S.cd_pb, (long long)L.n3js, S.cd_M)) {
          S.cd_valid = 1; S.cd_method = S.method; S.cd_mode = mode; S.cd_nsweeps = S.par.ns_coarsest; S.n_launch += 3;
        } else { S.cd_valid = 0; S.cd_n = -1; }   // no one-workgroup kernel for this level: the sweeps
      }
      if (S.cd_valid && mgxk_coarse_direct_apply(S.stream, &L.v, S.cd_M, S.cd_part, S.cd_cnt, ph)) { S.n_launch++; S.n_direct++; return 0; }
    } else S.cd_n = -1;
  }
  return relax(S.nlevs, S.par.ns_coarsest);
}

// mg_solvers.f90:129-151.  lead_c2f: the caller is Fcycle, whose coarse2fine(lev1) comes right before (:119-120)
int vcycle(int lev1, bool lead_c2f = false) {
  for (int lev = lev1; lev <= S.nlevs - 1; lev++) {
    const bool lead = lead_c2f && lev == lev1;
    if (relax_fused(lev, S.par.ns_pre, lead ? 3 : 2)) continue;
    if (lead) CHK(coarse2fine(lev, S.exact_halos || S.keep_r, S.c2f_skip && S.method == M_FC && S.par.ns_pre >= 1));
    CHK(relax(lev, S.par.ns_pre));
    CHK(fine2coarse(lev, false, true));  // compute_residual(lev) + fine2coarse(lev)
  }
  CHK(coarsest_solve(lev1 < S.nlevs));
  for (int lev = S.nlevs - 1; lev >= lev1; lev--) {
    if (relax_fused(lev, S.par.ns_post, 1)) continue;
    CHK(coarse2fine(lev, S.exact_halos || S.keep_r, S.c2f_skip && S.method == M_FC && S.par.ns_post >= 1));
    CHK(relax(lev, S.par.ns_post));
  }
  return 0;
}

// mg_solvers.f90:155-177: partial V-cycle down to level lev2
int vcycle2(int lev1, int lev2) {
  for (int lev = lev1; lev <= lev2 - 1; lev++) {
    CHK(relax(lev, S.par.ns_pre));
    CHK(fine2coarse(lev, false, true));  // compute_residual(lev) + fine2coarse(lev)
  }
  CHK(relax(lev2, S.par.ns_coarsest));
  for (int lev = lev2 - 1; lev >= lev1; lev--) {
    CHK(coarse2fine(lev, S.exact_halos || S.keep_r, S.c2f_skip && S.method == M_FC && S.par.ns_post >= 1));
    CHK(relax(lev, S.par.ns_post));
  }
  return 0;
}

// mg_solvers.f90:104-126
// have_r2: grid(2)%r already holds the restriction of the level-1 residual (the closing compute_residual of the previous solve_p
// iteration wrote it, residual_closing below): the first fine2coarse is then grid(2)%b = grid(2)%r and grid(2)%p = 0, two small copies
int fcycle(bool have_r2 = false) {
  TicScope ts(1, "Fcycle");  // mg_solvers.f90:108
  for (int lev = 1; lev <= S.nlevs - 1; lev++) {
    if (lev == 1 && have_r2) {
      Level &C = S.lev[1];
      HIPCHK(hipMemcpyAsync(C.v.b, C.v.r, C.n3js * sizeof(double), hipMemcpyDeviceToDevice, S.stream));   // physical images included (the kernel stored them)
      HIPCHK(hipMemsetAsync(C.v.p, 0, C.n3js * sizeof(double), S.stream));
      C.b_halo_stale = true; S.n_launch += 2;
      continue;
    }
    if (lev >= 2 && S.use_chain && !S.exact_halos) {
      // the rest of the first leg (closed, un-gathered levels: a single rank, or everything below the gathers) as ONE launch, up to four levels at a time
      int dep = 0;
      const LevView *vs[5] = {&S.lev[lev - 1].v, nullptr, nullptr, nullptr, nullptr};
      bool ok = true;
      for (int q = lev - 1; q < S.nlevs && ok; q++) { const Level &Lq = S.lev[q]; ok = Lq.neighb[0] < 0 && Lq.neighb[1] < 0 && Lq.neighb[2] < 0 && Lq.neighb[3] < 0 && (q == lev - 1 || !Lq.gather); }
      if (ok) {
        while (dep < 4 && lev + dep < S.nlevs) { dep++; vs[dep] = &S.lev[lev - 1 + dep].v; }
        const Level &F = S.lev[lev - 1];
        if (dep >= 2 && F.nx % (1 << dep) == 0 && F.ny % (1 << dep) == 0 && F.nz % (1 << dep) == 0) {
          const Sides all = {1, 1, 1, 1};
          mgxk_restrict_chain(S.stream, vs, dep, all); S.n_launch++;
          lev += dep - 1;
          continue;
        }
      }
    }
    CHK(fine2coarse(lev, true));  // + grid(lev+1)%r = grid(lev+1)%b (mg_solvers.f90:113)
  }
  CHK(coarsest_solve(S.nlevs >= 2));
  for (int lev = S.nlevs - 1; lev >= 1; lev--) CHK(vcycle(lev, true));  // coarse2fine(lev) + Vcycle(lev), :119-120
  return 0;
}

// compute_residual(1, res) at the end of a solve_p iteration (mg_solvers.f90:65).  If the loop goes on, the next thing that happens to this r
// is Fcycle's fine2coarse(1) (:112-115): the fused residual+restriction kernel (mgx_resrest.hip) forms the norm's partial sums AND the
// restricted r in one pass, into grid(2)%r only -- grid(2)%b and %p keep what the last cycle left, should the loop stop here.  The level-1 r
// is NOT written; the caller materialises it after the loop.  Returns 1 = fused (grid(2)%r is ready), 0 = the caller runs residual(1).
int residual_closing(double *res) {
  if (!S.fuse_closing || S.nlevs < 2 || S.exact_halos || S.keep_r || S.tictoc) return 0;
  Level &F = S.lev[0], &C = S.lev[1];
  if (C.gather) return 0;
  const int np = mgxk_residual_restrict_grid(&F.v, &C.v);
  if (np > S.npartial) return 0;
  const Sides phc = {C.neighb[0] < 0, C.neighb[1] < 0, C.neighb[2] < 0, C.neighb[3] < 0};
  if (!mgxk_residual_restrict_ex(S.stream, &F.v, &C.v, C.v.r, S.real, phc, nullptr, S.d_partial, nullptr)) return 0;
  mgxk_reduce(S.stream, S.d_partial, np, S.d_scalar); S.n_launch += 2;
  C.r_halo_stale = true;
  double s;
  if (global_sum(F, &s)) return -1;
  *res = sqrt(s);
  return 1;
}

// Fortran's Ew.3 edit descriptor (0.dddE+ee), so that the printed history reads like the reference's (format 10, mg_solvers.f90:99)
std::string fortran_e3(double v, int width) {
  char buf[32];
  if (v == 0.0 || !std::isfinite(v)) snprintf(buf, sizeof buf, v == 0.0 ? "0.000E+00" : "%f", v);
  else {
    const double a = fabs(v);
    int e = (int)floor(log10(a)) + 1;
    long m = lround(a / pow(10.0, e) * 1000.0);
    if (m >= 1000) { m = 100; e++; }
    if (m < 100) { m *= 10; e--; }
    if (abs(e) < 100) snprintf(buf, sizeof buf, "%s0.%03ldE%c%02d", v < 0 ? "-" : "", m, e < 0 ? '-' : '+', abs(e));
    else snprintf(buf, sizeof buf, "%s0.%03ld%c%03d", v < 0 ? "-" : "", m, e < 0 ? '-' : '+', abs(e));
  }
  std::string t(buf);
  if ((int)t.size() < width) t.insert(0, width - t.size(), ' ');
  return t;
}

// mg_solvers.f90:17-101
int solve_p(double tol, int maxite, int *nite_out, double *res_out, double *hist) {
  Level &L = S.lev[0];
  if (S.verbose && S.rank == 0) printf(" - solve p:\n");
  TicScope ts(1, "solve");  // mg_solvers.f90:45
  const auto tstart = std::chrono::steady_clock::now();  // cpu_time(tstart) (:46); wall clock here, the work is on the GPU
  if (!S.warm_start) HIPCHK(hipMemsetAsync(L.v.p, 0, L.n3js * sizeof(double), S.stream));  // grid(1)%p = 0 (:35)
  mgxk_sumsq(S.stream, &L.v, L.v.b, S.d_partial, S.d_scalar); S.n_launch += 2;
  double bnorm; CHK(global_sum(L, &bnorm)); bnorm = sqrt(bnorm);
  int nite = 0;
  double rnorm; CHK(residual(1, &rnorm));
  double res0 = rnorm / bnorm;
  const double rnorm0 = res0;
  if (hist) hist[0] = res0;
  FILE *f100 = (S.verbose && S.rank == 0) ? fopen("fort.100", "a") : nullptr;
  if (f100) fprintf(f100, " %24.16E %d\n", res0, nite);
  bool have_r2 = false;  // grid(2)%r = restriction of the current level-1 residual, and grid(1)%r not written (residual_closing)
  while (nite < maxite && res0 > tol) {
    CHK(fcycle(have_r2));
    const int fz = residual_closing(&rnorm);
    if (fz < 0) return 1;
    have_r2 = fz == 1;
    if (!fz) CHK(residual(1, &rnorm));
    rnorm = rnorm / bnorm;
    const double conv = res0 / rnorm;
    res0 = rnorm;
    nite++;
    if (hist) hist[nite] = rnorm;
    if (S.verbose && S.rank == 0) printf("ite = %2d: res = %s / conv = %10.3f\n", nite, fortran_e3(rnorm, 10).c_str(), conv);
    if (f100) fprintf(f100, " %24.16E %24.16E\n", rnorm, conv);
  }
  if (f100) fclose(f100);
  if (have_r2) CHK(residual(1, nullptr));  // grid(1)%r of the final iterate, which the fused closing residual did not write (once per solve)
  if (S.verbose && S.rank == 0) {  // the summary block (mg_solvers.f90:83-97)
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - tstart).count();
    const double np = (double)L.npx * L.npy, ncell = (double)L.nx * L.npx * (double)L.ny * L.npy * (double)L.nz;
    const double perf = dt * np / (-log(res0 / rnorm0) / log(10.0)) / ncell;
    printf(" --- summary ---\ntime spent to solve :%8.3f s\nrescaled performance:%s\n ---------------\n", dt, fortran_e3(perf, 10).c_str());
  }
  if (nite_out) *nite_out = nite;
  if (res_out) *res_out = res0;
  return 0;
}

// ---- set-up: mg_define_matrix.f90:28-208 ----------------------------------------------------------
int gather2d(Level &L, double *src_tmp, double *dst) {
  const int nxc = L.nx / L.ngx, nyc = L.ny / L.ngy, Ng = nxc * nyc;
  rect(src_tmp, L.blk, 3, 1, 1, nyc, 1, nyc, 1, nxc);
  if (!S.ag) return fail("a gather is needed but mgx_set_comm was not called");
  if (S.ag(S.ctx, L.group, L.ngroup, L.blk, L.gbuf, Ng)) return fail("allgather callback failed");
  for (int q = 0; q < L.ngroup; q++) {
    const int l = q % L.ngx, m = q / L.ngx;
    rect(dst, L.gbuf + (size_t)q * Ng, 4, 1, 1, L.ny, 1 + m * nyc, (m + 1) * nyc, 1 + l * nxc, (l + 1) * nxc);
  }
  return 0;
}

// the planes of warm-up each level's windowed red-black walk needs (mgx_rbseq.hip: k_rbseq_window), from the rho the set-up has just copied back
static void set_window_planes() {
  for (int l = 0; l < S.nlevs && l < 32; l++) {
    Level &L = S.lev[l];
    if (!L.v.gk || !S.rho_dev) { L.rbs_rho = -1.0; L.rbs_m = 0; L.rbs_rows = L.nz; continue; }
    L.rbs_rho = S.rho_host[l];
    L.rbs_m = mgxk_rbseq_window_planes(L.rbs_rho);
    L.rbs_rows = (L.gdec && !L.gdec_h.empty()) ? mgxk_rbseq_window_rows(L.gdec_h.data(), L.nz) : L.nz;
    if (S.verbose > 1 && S.rank == 0) printf(" level %d: red-black walk contracts by %.4g per plane: %d planes of warm-up%s\n", l + 1, L.rbs_rho, L.rbs_m, L.rbs_m ? "" : " (none: the walk over the whole level stays)");
  }
}

int define_matrices() {
  if (S.rho_dev) HIPCHK(hipMemsetAsync(S.rho_dev, 0, sizeof S.rho_host, S.stream));
  for (int l = 0; l < S.nlevs; l++) {
    Level &L = S.lev[l];
    if (l > 0) {
      Level &F = S.lev[l - 1];
      const int nxc = L.gather ? L.nx / L.ngx : L.nx, nyc = L.gather ? L.ny / L.ngy : L.ny;
      double *src[4] = {F.g.dx, F.g.dy, F.g.zeta, F.g.h};
      double *own[4] = {L.g.dx, L.g.dy, L.g.zeta, L.g.h};
      const double fac[4] = {0.5, 0.5, 0.25, 0.25};
      for (int q = 0; q < 4; q++) {
        mgxs_coarsen2d(S.stream, src[q], L.gather ? L.tmp2[q] : own[q], F.ny, nyc, nxc, fac[q]); S.n_launch++;
        if (L.gather) CHK(gather2d(L, L.tmp2[q], own[q]));
      }
    }
    CHK(rl_fill_halo(L, L.g.dx, 1, 1, 0));
    CHK(rl_fill_halo(L, L.g.dy, 1, 1, 0));
    CHK(rl_fill_halo(L, L.g.zeta, 1, 1, 0));
    CHK(rl_fill_halo(L, L.g.h, 1, 1, 0));
    mgxs_zr_zw(S.stream, &L.g, S.hlim, S.theta_b, S.theta_s); S.n_launch++;
    CHK(rl_fill_halo(L, L.g.zr, L.nz, 2, 0));
    CHK(rl_fill_halo(L, L.g.zw, L.nz + 1, 2, 0));
    // (no clearing of the cA scratch: k_cA_offdiag stores every slot of every cell, zeros included)
    L.g.bmask = S.par.bmask ? 1 : 0;
    if (l > 0) {  // boundary mask of a coarse level = 1, 0 in the physical halo when bmask (:157-161, fill_halo_2D_bmask)
      rect(L.g.rmask, 0, 5, 1, 1, L.ny, 0, L.ny + 1, 0, L.nx + 1);
      if (S.par.bmask) {
        if (L.neighb[0] < 0) rect(L.g.rmask, 0, 2, 1, 1, L.ny, 0, 0, 0, L.nx + 1);
        if (L.neighb[1] < 0) rect(L.g.rmask, 0, 2, 1, 1, L.ny, 0, L.ny + 1, L.nx + 1, L.nx + 1);
        if (L.neighb[2] < 0) rect(L.g.rmask, 0, 2, 1, 1, L.ny, L.ny + 1, L.ny + 1, 0, L.nx + 1);
        if (L.neighb[3] < 0) rect(L.g.rmask, 0, 2, 1, 1, L.ny, 0, L.ny + 1, 0, 0);
      }
    }
    L.g.szx = L.g.szy + (size_t)L.nz * (L.ny + 2) * (L.nx + 2);
    mgxs_slopes_ref(S.stream, &L.g); S.n_launch++;  // zy, zx once per cell: the cross coefficients and the smoother's matrix-free slopes both come from here
    mgxs_define_matrix(S.stream, &L.g, l == 0, 0); S.n_launch += 2;
    // fill_halo(lev,cA), mg_define_matrix.f90:611-613: the 4-D exchange, slot by slot (the set-up scratch is slot-major)
    if (S.par.bmask) for (int s = 0; s < 8; s++) CHK(rl_fill_halo(L, L.g.cA + (size_t)s * L.nz * (L.ny + 2) * (L.nx + 2), L.nz, 1, 0, true));
    mgxs_define_matrix(S.stream, &L.g, l == 0, 1); S.n_launch++;
    if (l == 0) {  // i-fastest copies for compute_rhs / correct_uvw (mgx_model.hip)
      mgxm_ref2model(S.stream, L.g.zw, L.g.mzw, L.nz + 1, 2, L.nx, L.ny);
      mgxm_ref2model(S.stream, L.g.dzw, L.g.mdzw, L.nz + 1, 1, L.nx, L.ny);
      mgxm_ref2model(S.stream, L.g.cw, L.g.mcw, L.nz + 1, 1, L.nx, L.ny);
      mgxm_ref2model(S.stream, L.g.zxdy, L.g.mzxdy, L.nz, 1, L.nx, L.ny);
      mgxm_ref2model(S.stream, L.g.zydx, L.g.mzydx, L.nz, 1, L.nx, L.ny);
      mgxm_ref2model_2d(S.stream, L.g.dx, L.g.mdx, L.nx, L.ny);
      mgxm_ref2model_2d(S.stream, L.g.dy, L.g.mdy, L.nx, L.ny);
      mgxm_ref2model_2d(S.stream, L.g.rmask, L.g.mrmask, L.nx, L.ny);
      S.n_launch += 8;
    }
    if (L.nz <= 1024) { mgxk_convert8(S.stream, &L.v, L.g.cA); S.n_launch++; }  // LDS-tiled transposition, one slot per block
    else for (int s = 0; s < 8; s++) { mgxk_convert(S.stream, &L.v, L.v.cA[s], L.g.cA + (size_t)s * L.nz * (L.ny + 2) * (L.nx + 2), 1, 0, 0); S.n_launch++; }
    mgxs_pivots(S.stream, &L.v); S.n_launch++;
    if (L.v.gk) { mgxk_rbseq_setup(S.stream, &L.v); S.n_launch++; }
    if (L.v.gk && S.rho_dev && l < 32) { mgxk_rbseq_rho(S.stream, &L.v, S.rho_dev + l); S.n_launch++; }
    if (L.v.gk && L.gdec) {
      HIPCHK(hipMemsetAsync(L.gdec, 0, (size_t)L.nz * sizeof(double), S.stream));
      mgxk_rbseq_gdecay(S.stream, &L.v, L.gdec); S.n_launch++;
      HIPCHK(hipMemcpyAsync(L.gdec_h.data(), L.gdec, (size_t)L.nz * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    }
    L.v.zy = L.zy_store; L.v.zx = L.zx_store;
    if (L.nz <= 1024) { mgxk_convert2(S.stream, &L.v, L.zy_store, L.zx_store, L.g.szy); S.n_launch++; }
    else { mgxk_convert(S.stream, &L.v, L.zy_store, L.g.szy, 1, 0, 0); mgxk_convert(S.stream, &L.v, L.zx_store, L.g.szx, 1, 0, 0); S.n_launch += 2; }
    L.v.m4 = L.f2d_store[0]; L.v.d4 = L.f2d_store[1]; L.v.m7 = L.f2d_store[2]; L.v.d7 = L.f2d_store[3];
    L.v.h2 = L.f2d_store[4]; L.v.hi2 = L.f2d_store[5]; L.v.ze2 = L.f2d_store[6]; L.v.cffw = L.tab_store[0]; L.v.csw = L.tab_store[1];
    L.v.dx2 = L.zg_store[0]; L.v.dy2 = L.zg_store[1]; L.v.cffr = L.zg_store[2]; L.v.csr = L.zg_store[3];
    mgxs_zw_js(S.stream, &L.g, &L.v, S.hlim, S.theta_b, S.theta_s); S.n_launch += 3;
    if (S.no_mf || S.par.bmask) { L.v.zy = L.v.zx = nullptr; L.v.m4 = nullptr; }  // masked coefficients are not rebuilt from the slopes
  }
  if (S.rho_dev) HIPCHK(hipMemcpyAsync(S.rho_host, S.rho_dev, sizeof S.rho_host, hipMemcpyDeviceToHost, S.stream));
  CHK(sync_stream());
  set_window_planes();
  S.cd_valid = 0;
  S.have_matrix = true;
  return 0;
}

// the model fields + the mask of the call.  The reference multiplies the w cross terms of compute_rhs by the rmask of the
// call whatever bmask says, and builds umask / vmask from it only when bmask (mg_compute_rhs.f90:56-72,110-111,
// mg_correct_uvw.f90:51-68).  Without a per-call mask (NULL): the level-1 mask of nhydro_matrices when bmask, else all ones.
ModelView model_view() {
  double *m = S.call_mask ? S.d_rmask_m : (S.par.bmask ? S.lev[0].g.mrmask : nullptr);
  return ModelView{S.d_u, S.d_v, S.d_w, m, S.par.bmask ? 1 : 0};
}

// rmaska of nhydro_solve / nhydro_check_nondivergence: (0:ny+1,0:nx+1), j fastest -- the layout the reference's drivers
// allocate (mg_testseamount.f90:97) and compute_rhs indexes (rmask(j,i)).  `dev`: the pointer is a device pointer.
int set_call_mask(const double *rmask, bool dev) {
  S.call_mask = rmask != nullptr;
  if (!rmask) return 0;
  Level &L = S.lev[0];
  const size_t n2 = (size_t)(L.ny + 2) * (L.nx + 2) * sizeof(double);
  HIPCHK(hipMemcpyAsync(S.d_rmask_ref, rmask, n2, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, S.stream));
  mgxm_ref2model_2d(S.stream, S.d_rmask_ref, S.d_rmask_m, L.nx, L.ny); S.n_launch++;
  return 0;
}

// fill_halo(1,uf,lbc_null='u') / fill_halo(1,vf,lbc_null='v') (mg_compute_rhs.f90:171,272), reduced to the entries the
// divergence reads: the first and last face.  Physical side: zero flux.  Neighbour: my last face is the neighbour's first
// face, computed over there from its own copy of the shared velocity (the reference takes that value too).
int flux_halo(Level &L, int face, double *fx) {
  const int lo = face == 0 ? L.neighb[3] : L.neighb[0], hi = face == 0 ? L.neighb[1] : L.neighb[2];  // W,E or S,N
  const int last = face == 0 ? L.nx + 1 : L.ny + 1, cnt = L.nz * (face == 0 ? L.ny : L.nx);
  if (lo < 0) { mgxm_flux_zero_face(S.stream, &L.g, fx, face, 1); S.n_launch++; }
  if (hi < 0) { mgxm_flux_zero_face(S.stream, &L.g, fx, face, last); S.n_launch++; }
  if (lo < 0 && hi < 0) return 0;
  if ((size_t)cnt > S.xbuf_n) return fail("halo buffer too small");
  // the exchange callback moves equal counts both ways with every peer: the unused direction carries a zero buffer
  int n = 0, peer[2], cn[2]; double *sb[2], *rb[2];
  if (lo >= 0) { mgxm_flux_face_copy(S.stream, &L.g, fx, S.xbuf[0], face, 1, 0); S.n_launch++; peer[n] = lo; cn[n] = cnt; sb[n] = S.xbuf[0]; rb[n] = S.xbuf[8]; n++; }
  if (hi >= 0) { peer[n] = hi; cn[n] = cnt; sb[n] = S.xbuf[2]; rb[n] = S.xbuf[9]; n++; }
  if (hi >= 0) HIPCHK(hipMemsetAsync(S.xbuf[2], 0, (size_t)cnt * sizeof(double), S.stream));
  CHK(exchange(n, peer, sb, rb, cn));
  if (hi >= 0) { mgxm_flux_face_copy(S.stream, &L.g, fx, S.xbuf[9], face, last, 1); S.n_launch++; }
  return 0;
}

// mg_compute_rhs.f90:14-379 on the device copies of u,v,w
int compute_rhs_dev() {
  Level &L = S.lev[0];
  TicScope ts(1, "compute_rhs");  // nhydro.f90:81
  ModelView M = model_view();
  HIPCHK(hipMemsetAsync(L.v.b, 0, L.n3js * sizeof(double), S.stream));
  mgxm_rhs_uf(S.stream, &L.g, &M, S.d_fx); S.n_launch++;
  if (!S.par.bmask) CHK(flux_halo(L, 0, S.d_fx));  // mg_compute_rhs.f90:170-172
  mgxm_rhs_vf(S.stream, &L.g, &M, S.d_fy); S.n_launch++;
  if (!S.par.bmask) CHK(flux_halo(L, 1, S.d_fy));  // :271-273
  mgxm_rhs_wf(S.stream, &L.g, &M, S.d_fz); S.n_launch++;
  mgxm_rhs_accum(S.stream, &L.g, S.d_bm, S.d_fx, S.d_fy, S.d_fz); S.n_launch++;  // :173, :274, :362-370 in one pass, same order
  mgxm_js_model(S.stream, &L.v, L.v.b, S.d_bm, 1); S.n_launch++;  // interior of b in the solver's layout
  return 0;
}

// mg_correct_uvw.f90:15-115 on the device copies of u,v,w
int correct_uvw_dev() {
  Level &L = S.lev[0];
  TicScope ts(1, "correct_uvw");
  ModelView M = model_view();
  mgxm_js_model(S.stream, &L.v, L.v.p, S.d_bm, 0); S.n_launch++;
  mgxm_correct_uvw(S.stream, &L.g, S.d_bm, &M); S.n_launch++;
  return 0;
}

int upload_uvw(const double *u, const double *v, const double *w) {
  Level &L = S.lev[0];
  const size_t nu = (size_t)(L.nx + 1) * (L.ny + 2) * L.nz, nv = (size_t)(L.nx + 2) * (L.ny + 1) * L.nz, nw = (size_t)(L.nx + 2) * (L.ny + 2) * (L.nz + 1);
  HIPCHK(hipMemcpyAsync(S.d_u, u, nu * sizeof(double), hipMemcpyHostToDevice, S.stream));
  HIPCHK(hipMemcpyAsync(S.d_v, v, nv * sizeof(double), hipMemcpyHostToDevice, S.stream));
  HIPCHK(hipMemcpyAsync(S.d_w, w, nw * sizeof(double), hipMemcpyHostToDevice, S.stream));
  return 0;
}

bool streq(const char *a, const char *b) { return strcmp(a, b) == 0; }

// ---- peer-to-peer halo transport: set-up / tear-down ---------------------------------------------------------
void p2p_release() {
  for (int r = 0; r < (int)S.peer_slab.size(); r++) {
    if (r == S.rank || S.p2p_borrowed) continue;
    if (S.peer_slab[r]) (void)hipIpcCloseMemHandle(S.peer_slab[r]);
    if (S.peer_flags[r]) (void)hipIpcCloseMemHandle(S.peer_flags[r]);
  }
  S.peer_slab.clear(); S.peer_flags.clear();
  if (S.p2p_slab) (void)hipFree(S.p2p_slab);
  if (S.p2p_flags) (void)hipFree(S.p2p_flags);
  if (S.p2p_counter) (void)hipFree(S.p2p_counter);
  if (S.p2p_err) (void)hipHostFree(S.p2p_err);
  S.p2p_slab = nullptr; S.p2p_flags = nullptr; S.p2p_counter = nullptr; S.p2p_err = nullptr;
  S.p2p_ready = S.p2p_on = S.p2p_borrowed = false;
}

// stream synchronise + the peer-to-peer error word (a neighbour that never raised its flag)
int sync_stream() {
  HIPCHK(hipStreamSynchronize(S.stream));
  // a kernel launch this thread issued since the last check was refused (launch configuration, LDS or register demand on this
  // device / ROCm): the operator it belonged to did not run, so the fields are not what the caller thinks -- fail loudly
  {
    hipError_t le = hipGetLastError();
    if (le == hipSuccess && mgx_pending_error != hipSuccess) le = mgx_pending_error;
    mgx_pending_error = hipSuccess;
    if (le != hipSuccess) return fail("a HIP call of this thread failed since the last synchronisation (a rejected kernel launch, or an earlier call of the host program): %s", hipGetErrorString(le));
  }
  if (S.kerr && *S.kerr == 2) {
    // the fused sequential-order red-black launch: a forwarding wave did not see the walk's progress within 2 s, or found itself on another
    // XCD than the walk (mgx_rbseq.hip).  The correction of that colour used stale values: the fused launch is OFF from now on.
    *S.kerr = 0;
    S.rbseq_fuse = 0; S.rbseq_test_stall = 0;
    return fail("the fused red-black walk + correction launch lost its hand-off (forwarding waves timed out or ran on another XCD than the walk); "
                "the fields of that level are wrong -- it is now OFF (option rbseq_fuse = 0: the correction in a launch of its own)");
  }
  if (S.kerr && *S.kerr) {
    // a workgroup of the persistent relax kernel waited 2 s for its neighbour plane: some of its workgroups were kept off the chip
    // (the GPU is shared with kernels that do not finish).  The sweep is incomplete: counters back to zero, the separate launches from now on.
    *S.kerr = 0;
    for (auto &L : S.lev) { if (L.ksp_done) (void)hipMemsetAsync(L.ksp_done, 0, (size_t)(L.nx + 2) * sizeof(unsigned int), S.stream); L.ksp_seq = 0; }
    S.ksp_down = 1;
    return fail("the persistent relax kernel timed out waiting for a neighbouring plane (its workgroups were not all resident); "
                "the fields of that level are incomplete -- it is now OFF (one launch per colour pair)");
  }
  if (S.p2p_err && *S.p2p_err) {
    // A wait on a neighbour's flag timed out (the edge it was waiting for stayed stale).  This rank must NOT fall back alone -- its
    // neighbours would go on pushing to flags nobody reads and waiting for pushes that never come: it keeps exchanging (the
    // sequence numbers stay in step, flags are compared with >=) and remembers; the ranks agree at the next global_sum (every
    // solve_p iteration, every norm), where all of them switch to the hooks together and report the error.
    *S.p2p_err = 0;
    S.p2p_failed = 1;
    if (S.verbose) fprintf(stderr, "mgx warning: rank %d: a peer-to-peer halo wait timed out; reported collectively at the next norm\n", S.rank);
  }
  return 0;
}

// end of a cycle / operator entry point: wait for the stream (and report what the device flagged), unless the caller asked for asynchronous
// operators (option "async"): then the work is only enqueued, as a GPU-resident model would want, and mgx_synchronize reports later
int op_sync() { return S.async_ops ? 0 : sync_stream(); }

int apply_params(const mgx_params &p) {
  if (streq(p.relax_method, "GS") || streq(p.relax_method, "Gauss-Seidel")) S.method = M_GS;
  else if (streq(p.relax_method, "RB") || streq(p.relax_method, "Red-Black")) S.method = M_RB;
  else if (streq(p.relax_method, "FC") || streq(p.relax_method, "Four-Color")) S.method = M_FC;
  else return fail("unknown relax_method '%s'", p.relax_method);
  S.real = streq(p.cmatrix, "real") ? 1 : 0;
  if (streq(p.interp_type, "linear")) S.linear = 1; else if (streq(p.interp_type, "nearest")) S.linear = 0; else return fail("unknown interp_type '%s'", p.interp_type);
  if (S.linear && streq(p.restrict_type, "linear")) return fail("linear interp + linear restrict is not permitted");
  if (p.aggressive) return fail("aggressive=.true.: coarse2fine_aggressive is not available in the reference either (mg_intergrids.f90:243)");
  S.par = p;
  return 0;
}

// A level-1 halo fill of a rank-coded field through the CURRENT neighbour transport (the hooks, or the pushes when they are on), all
// eight directions: every halo cell must hold the value its owner encoded (mg_testhalo.f90:75-92 with positions, not just ranks).
// Collective.  Leaves level-1 p zeroed.
int halo_rank_coded_check(const char *who) {
  Level &L = S.lev[0];
  const int nx = L.nx, ny = L.ny, nz = L.nz;
  const size_t n3 = (size_t)nz * (ny + 2) * (nx + 2);
  std::vector<double> h(n3, -1.0);
  auto at = [&](int k, int j, int i) -> size_t { return (size_t)k + (size_t)nz * ((size_t)j + (size_t)(ny + 2) * i); };
  auto code = [&](int r, int k, int j, int i) { return 1.0e7 * (r + 1) + (double)at(k, j, i); };
  for (int i = 1; i <= nx; i++) for (int j = 1; j <= ny; j++) for (int k = 0; k < nz; k++) h[at(k, j, i)] = code(S.rank, k, j, i);
  HIPCHK(hipMemcpyAsync(S.ref_scratch, h.data(), n3 * sizeof(double), hipMemcpyHostToDevice, S.stream));
  mgxk_convert(S.stream, &L.v, L.v.p, S.ref_scratch, 1, 0, 0);
  CHK(fill_halo_js(L, L.v.p));
  mgxk_convert(S.stream, &L.v, L.v.p, S.ref_scratch, 1, 0, 1);
  HIPCHK(hipMemcpyAsync(h.data(), S.ref_scratch, n3 * sizeof(double), hipMemcpyDeviceToHost, S.stream));
  CHK(sync_stream());
  HIPCHK(hipMemsetAsync(L.v.p, 0, L.n3js * sizeof(double), S.stream));
  const int *nb = L.neighb;
  // halo cell (j,i) of direction d is the owner's cell (js,is): S,E,N,W,SW,SE,NE,NW
  for (int d = 0; d < 8; d++) {
    if (nb[d] < 0) continue;
    const bool south = (d == 0 || d == 4 || d == 5), north = (d == 2 || d == 6 || d == 7), east = (d == 1 || d == 5 || d == 6), west = (d == 3 || d == 4 || d == 7);
    const int j0 = south ? 0 : (north ? ny + 1 : 1), j1 = south ? 0 : (north ? ny + 1 : ny);
    const int i0 = west ? 0 : (east ? nx + 1 : 1), i1 = west ? 0 : (east ? nx + 1 : nx);
    for (int i = i0; i <= i1; i++) for (int j = j0; j <= j1; j++) for (int k = 0; k < nz; k++) {
      const int js = south ? ny : (north ? 1 : j), is = west ? nx : (east ? 1 : i);
      if (h[at(k, j, i)] != code(nb[d], k, js, is)) return fail("%s: halo cell (k=%d,j=%d,i=%d) of direction %d does not hold rank %d's value", who, k + 1, j, i, d, nb[d]);
    }
  }
  return 0;
}

}  // namespace

// ====================================================================================================
extern "C" {

// ---- instances (see the comment at State S0) --------------------------------------------------------------------------------
int mgx_instance_create(void) {
  std::lock_guard<std::mutex> lk(g_instances_mu);
  for (size_t q = 1; q < g_instances.size(); q++) if (!g_instances[q]) { g_instances[q] = new State(); return (int)q; }
  g_instances.push_back(new State());
  return (int)g_instances.size() - 1;
}
int mgx_instance_select(int id) {
  std::lock_guard<std::mutex> lk(g_instances_mu);
  if (id < 0 || id >= (int)g_instances.size() || !g_instances[id]) return fail("mgx_instance_select: no instance %d", id);
  Sp = g_instances[id];
  return 0;
}
int mgx_instance_current(void) {
  std::lock_guard<std::mutex> lk(g_instances_mu);
  for (size_t q = 0; q < g_instances.size(); q++) if (g_instances[q] == Sp) return (int)q;
  return -1;
}
// the calling thread must have selected another instance (or 0) before; instance 0 cannot be destroyed
int mgx_instance_destroy(int id) {
  State *victim = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_instances_mu);
    if (id < 1 || id >= (int)g_instances.size() || !g_instances[id]) return fail("mgx_instance_destroy: no instance %d (instance 0 is permanent)", id);
    victim = g_instances[id];
    g_instances[id] = nullptr;
  }
  State *mine = Sp;
  Sp = victim;
  mgx_clean();
  Sp = (mine == victim) ? &S0 : mine;
  delete victim;
  return 0;
}

const char *mgx_last_error(void) { return S.err.c_str(); }
const char *mgx_version(void) { return "mgx 0.1 (gfx950)"; }
int mgx_set_verbose(int v) { S.verbose = v; return 0; }
int mgx_set_stream(void *st) { S.stream = (hipStream_t)st; return 0; }
int mgx_set_comm(mgx_exchange_fn ex, mgx_allreduce_fn ar, mgx_allgather_fn ag, void *ctx) { S.ex = ex; S.ar = ar; S.ag = ag; S.ctx = ctx; S.native_rccl = false; return 0; }

// ---- native RCCL transport -------------------------------------------------------------------------------------
int mgx_rccl_unique_id_bytes(void) { return 128; }
int mgx_rccl_get_unique_id(void *id_out) { if (mgxr_get_unique_id(id_out)) return fail("mgx_rccl_get_unique_id: %s", mgxr_last_error()); return 0; }
int mgx_rccl_connect(const void *id, int nranks, int rank) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail("mgx_rccl_connect: rank %d of %d", rank, nranks);
  if (mgxr_connect(id, nranks, rank)) return fail("mgx_rccl_connect: %s", mgxr_last_error());
  S.ex = rccl_exchange_hook; S.ar = rccl_allreduce_hook; S.ag = rccl_allgather_hook; S.ctx = nullptr; S.native_rccl = true;
  return 0;
}
int mgx_rccl_disconnect(void) {
  if (S.native_rccl) { S.ex = nullptr; S.ar = nullptr; S.ag = nullptr; S.native_rccl = false; }
  mgxr_disconnect();
  return 0;
}
// Collective self-test of the native transport (any world size, after mgx_init): one grouped exchange of rank-coded buffers with the
// next and the previous rank (with itself on one rank), an all-reduce of rank+1, an all-gather inside groups of up to four ranks and a
// level-1 halo fill of a position-coded field over all eight neighbour directions -- through the same hooks the solver uses.
// 0 = every value arrived.  Level-1 p is zero afterwards.
int mgx_rccl_selftest(void) {
  NEED_INIT();
  if (!S.native_rccl || !mgxr_connected()) return fail("mgx_rccl_selftest: the native RCCL transport is not connected");
  const int n = mgxr_nranks(), me = S.rank;
  if (n != S.nranks) return fail("mgx_rccl_selftest: communicator has %d ranks, the solver %d", n, S.nranks);
  const int cnt = (int)std::min<size_t>(1000, S.xbuf_n);
  int peers[2], np = 0;
  peers[np++] = (me + 1) % n;
  if ((me - 1 + n) % n != peers[0]) peers[np++] = (me - 1 + n) % n;
  std::vector<double> h(cnt);
  double *sb[2], *rb[2]; int cn[2];
  for (int q = 0; q < np; q++) {
    for (int t = 0; t < cnt; t++) h[t] = 1000.0 * me + peers[q] + 1e-3 * t;
    HIPCHK(hipMemcpyAsync(S.xbuf[q], h.data(), cnt * sizeof(double), hipMemcpyHostToDevice, S.stream));
    HIPCHK(hipStreamSynchronize(S.stream));
    HIPCHK(hipMemsetAsync(S.xbuf[8 + q], 0, cnt * sizeof(double), S.stream));
    sb[q] = S.xbuf[q]; rb[q] = S.xbuf[8 + q]; cn[q] = cnt;
  }
  CHK(exchange(np, peers, sb, rb, cn));
  for (int q = 0; q < np; q++) {
    HIPCHK(hipMemcpyAsync(h.data(), S.xbuf[8 + q], cnt * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    CHK(sync_stream());
    for (int t = 0; t < cnt; t++) if (h[t] != 1000.0 * peers[q] + me + 1e-3 * t) return fail("mgx_rccl_selftest: wrong data from rank %d (element %d)", peers[q], t);
  }
  S.h_scalar[0] = me + 1.0;
  HIPCHK(hipMemcpyAsync(S.d_scalar, S.h_scalar, sizeof(double), hipMemcpyHostToDevice, S.stream));
  if (S.ar(S.ctx, S.d_scalar, 1)) return fail("mgx_rccl_selftest: all-reduce failed: %s", mgxr_last_error());
  HIPCHK(hipMemcpyAsync(S.h_scalar, S.d_scalar, sizeof(double), hipMemcpyDeviceToHost, S.stream));
  CHK(sync_stream());
  if (S.h_scalar[0] != 0.5 * n * (n + 1)) return fail("mgx_rccl_selftest: all-reduce gave %g, expected %g", S.h_scalar[0], 0.5 * n * (n + 1));
  {  // all-gather leg (gather_3D's hook): groups of up to four consecutive ranks, the shape of the reference's colour groups
    const int g0 = me / 4 * 4, ng = std::min(4, n - g0), gc = (int)std::min<size_t>(500, S.ref_scratch_n / 8);
    int grp[4];
    for (int q = 0; q < ng; q++) grp[q] = g0 + q;
    std::vector<double> hs(gc), hr((size_t)gc * ng);
    for (int t = 0; t < gc; t++) hs[t] = 7000.0 * me + t;
    double *sb = S.ref_scratch, *rb = S.ref_scratch + gc;
    HIPCHK(hipMemcpyAsync(sb, hs.data(), gc * sizeof(double), hipMemcpyHostToDevice, S.stream));
    HIPCHK(hipMemsetAsync(rb, 0, (size_t)gc * ng * sizeof(double), S.stream));
    HIPCHK(hipStreamSynchronize(S.stream));
    if (S.ag(S.ctx, grp, ng, sb, rb, gc)) return fail("mgx_rccl_selftest: all-gather failed: %s", mgxr_last_error());
    HIPCHK(hipMemcpyAsync(hr.data(), rb, (size_t)gc * ng * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    CHK(sync_stream());
    for (int q = 0; q < ng; q++)
      for (int t = 0; t < gc; t++) if (hr[(size_t)q * gc + t] != 7000.0 * grp[q] + t) return fail("mgx_rccl_selftest: all-gather slot %d holds wrong data (element %d)", q, t);
  }
  CHK(halo_rank_coded_check("mgx_rccl_selftest"));
  return 0;
}
// which transport carries the neighbour traffic right now
const char *mgx_transport(void) {
  std::string &t = S.transport_name;
  if (S.nranks <= 1 && !S.native_rccl) t = "none (one rank)";
  else {
    t = S.native_rccl ? std::string("RCCL, native (") + mgxr_library() + ")" : (S.ex ? "host callbacks (mgx_set_comm)" : "none");
    if (S.p2p_on) t = "peer-to-peer pushes over hipIpc-shared buffers for the cycle's halos and gathers; " + t + " for set-up halos and the norm";
  }
  return t.c_str();
}

int mgx_params_default(mgx_params *p) {
  memset(p, 0, sizeof(*p));
  p->solver_prec = 1e-6; p->solver_maxiter = 50; p->nsmall = 8; p->ns_coarsest = 40; p->ns_pre = 3; p->ns_post = 2;
  strcpy(p->cmatrix, "real"); strcpy(p->relax_method, "RB"); strcpy(p->interp_type, "linear"); strcpy(p->restrict_type, "avg");
  return 0;
}

int mgx_read_namelist(const char *path, mgx_params *p) {
  FILE *f = fopen(path ? path : "nh_namelist", "r");
  if (!f) return 0;  // defaults stay (mg_namelist.f90:75-86)
  // Fortran namelist rules as far as /nhparam/ needs them (checked against read_nhnamelist of the reference compiled with flang,
  // tests/golden/ref_namelist.json): group and member names in any case; `!` starts a comment outside a string; assignments are
  // separated by commas, blanks or line ends; the group ends at `/`.
  std::string txt; char line[1024];
  while (fgets(line, sizeof(line), f)) {
    std::string s(line); char q = 0; size_t c = std::string::npos;
    for (size_t t = 0; t < s.size(); t++) {
      if (q) { if (s[t] == q) q = 0; }
      else if (s[t] == '\'' || s[t] == '"') q = s[t];
      else if (s[t] == '!') { c = t; break; }
    }
    if (c != std::string::npos) s = s.substr(0, c);
    txt += s + "\n";
  }
  fclose(f);
  std::string low = txt;
  for (auto &ch : low) ch = (char)tolower(ch);
  size_t a = low.find("&nhparam");
  if (a == std::string::npos) return fail("namelist group &nhparam not found in %s", path ? path : "nh_namelist");
  size_t pos = a + 8;
  const std::string ws = " \t\r\n,";
  for (;;) {
    pos = txt.find_first_not_of(ws, pos);
    if (pos == std::string::npos || txt[pos] == '/') break;
    size_t ke = pos;
    while (ke < txt.size() && (isalnum((unsigned char)txt[ke]) || txt[ke] == '_')) ke++;
    size_t eq = txt.find_first_not_of(" \t\r\n", ke);
    if (ke == pos || eq == std::string::npos || txt[eq] != '=') return fail("cannot parse namelist statement near '%s'", txt.substr(pos, 24).c_str());
    std::string key = low.substr(pos, ke - pos);
    size_t vb = txt.find_first_not_of(" \t\r\n", eq + 1), ve;
    if (vb == std::string::npos) return fail("namelist member '%s' has no value", key.c_str());
    std::string val, sv;
    if (txt[vb] == '\'' || txt[vb] == '"') {
      ve = txt.find(txt[vb], vb + 1);
      if (ve == std::string::npos) return fail("unterminated string for namelist member '%s'", key.c_str());
      sv = txt.substr(vb + 1, ve - vb - 1); val = sv; ve++;
    } else {
      ve = txt.find_first_of(" \t\r\n,/", vb);
      if (ve == std::string::npos) ve = txt.size();
      val = txt.substr(vb, ve - vb); sv = val;
    }
    pos = ve;
    auto num = [&](void) { std::string t = val; for (auto &ch : t) if (ch == 'd' || ch == 'D') ch = 'e'; return atof(t.c_str()); };
    auto lg = [&](void) { std::string t = val; for (auto &ch : t) ch = (char)tolower(ch); return (t.find(".t") == 0 || t.find("t") == 0) ? 1 : 0; };
    if (key == "solver_prec") p->solver_prec = num();
    else if (key == "solver_maxiter") p->solver_maxiter = (int)num();
    else if (key == "nsmall") p->nsmall = (int)num();
    else if (key == "ns_coarsest") p->ns_coarsest = (int)num();
    else if (key == "ns_pre") p->ns_pre = (int)num();
    else if (key == "ns_post") p->ns_post = (int)num();
    else if (key == "cmatrix") snprintf(p->cmatrix, 16, "%s", sv.c_str());
    else if (key == "relax_method") snprintf(p->relax_method, 16, "%s", sv.c_str());
    else if (key == "interp_type") snprintf(p->interp_type, 16, "%s", sv.c_str());
    else if (key == "restrict_type") snprintf(p->restrict_type, 16, "%s", sv.c_str());
    else if (key == "aggressive") p->aggressive = lg();
    else if (key == "netcdf_output") p->netcdf_output = lg();
    else if (key == "bmask") p->bmask = lg();
    else return fail("'%s' is not a member of namelist /nhparam/", key.c_str());  // a Fortran read would abort too
  }
  if (streq(p->interp_type, "linear") && streq(p->restrict_type, "linear")) return fail("linear interp + linear restrict is not permitted");
  return 0;
}

void mgx_clean(void) {
  if (S.stream || S.inited) (void)hipStreamSynchronize(S.stream);
  p2p_release();
  for (void *q : S.allocs) (void)hipFree(q);
  if (S.h_scalar) (void)hipHostFree(S.h_scalar);
  if (S.kerr) (void)hipHostFree(S.kerr);
  if (S.stream2) { (void)hipStreamSynchronize(S.stream2); (void)hipStreamDestroy(S.stream2); }
  if (S.ev_a) (void)hipEventDestroy(S.ev_a);
  if (S.ev_s) (void)hipEventDestroy(S.ev_s);
  if (S.ev_x) (void)hipEventDestroy(S.ev_x);
  tt_collect();
  hipStream_t st = S.stream; int vb = S.verbose, ws = S.warm_start, tc = S.tictoc, eh = S.exact_halos, rx = S.rb_exact, rq = S.rb_seq, kr = S.keep_r, cs = S.c2f_skip, fc = S.fuse_closing, uc = S.use_chain, rf = S.rbseq_fuse, rw = S.rbseq_window, cdo = S.coarsest_direct, rfm = S.rbseq_fuse_min, ovl = S.overlap, kp = S.use_ksp, fz = S.use_fuse, ao = S.async_ops;
  mgx_exchange_fn ex = S.ex; mgx_allreduce_fn ar = S.ar; mgx_allgather_fn ag = S.ag; void *ctx = S.ctx; const bool nat = S.native_rccl;
  // the timer table is module state of mg_tictoc in the reference: it outlives nhydro_clean (the drivers print it afterwards, mg_testseamount.f90:220-221)
  std::vector<std::string> tn = S.tt_names; std::vector<HostTic> th = S.tt_host; const int tnb = S.tt_nblev;
  static thread_local double tsave[32][32]; static thread_local long long csave[32][32];
  memcpy(tsave, S.tt_time, sizeof tsave); memcpy(csave, S.tt_calls, sizeof csave);
  S = State();
  S.tt_names = tn; S.tt_host = th; S.tt_nblev = tnb; memcpy(S.tt_time, tsave, sizeof tsave); memcpy(S.tt_calls, csave, sizeof csave);
  S.native_rccl = nat;
  S.stream = st; S.verbose = vb; S.warm_start = ws; S.tictoc = tc; S.exact_halos = eh; S.rb_exact = rx; S.rb_seq = rq; S.keep_r = kr; S.c2f_skip = cs; S.fuse_closing = fc; S.use_chain = uc; S.rbseq_fuse = rf; S.rbseq_window = rw; S.coarsest_direct = cdo; S.rbseq_fuse_min = rfm; S.overlap = ovl; S.use_ksp = kp; S.use_fuse = fz; S.async_ops = ao; S.ex = ex; S.ar = ar; S.ag = ag; S.ctx = ctx;
}

int mgx_init(int nx, int ny, int nz, int npx, int npy, int rank, const mgx_params *par) {
  if (S.inited) mgx_clean();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device visible: libmgx has no CPU path");
  mgx_params p;
  if (par) p = *par; else { mgx_params_default(&p); CHK(mgx_read_namelist(nullptr, &p)); }
  CHK(apply_params(p));
  if (nx < 2 || ny < 2 || nz < 2 || (nx & 1) || (ny & 1) || (nz & 1)) return fail("nx,ny,nz must be even and >= 2 (got %d %d %d)", nx, ny, nz);
  if (npx < 1 || npy < 1 || (npx & (npx - 1)) || (npy & (npy - 1))) return fail("the process grid must be powers of two in both directions (got %d x %d)", npx, npy);
  if (rank < 0 || rank >= npx * npy) return fail("rank %d outside the %d x %d process grid", rank, npx, npy);
  S.npx = npx; S.npy = npy; S.nranks = npx * npy; S.rank = rank; S.pi = rank % npx; S.pj = rank / npx;
  S.nlevs = find_grid_levels(npx, npy, nx, ny, nz);
  if (S.nlevs < 1) return fail("grid %dx%dx%d too small for a multigrid hierarchy", nx * npx, ny * npy, nz);
  S.lev.assign(S.nlevs, Level());
  S.lev[0].nx = nx; S.lev[0].ny = ny; S.lev[0].nz = nz;
  rank_level_table(rank, S.lev, npx, npy, S.par.nsmall);
  for (int l = 0; l < S.nlevs; l++) {
    const Level &L = S.lev[l];
    if ((L.nx & 1) || (L.ny & 1) || L.nz < 2) return fail("level %d has local size %dx%dx%d: odd sizes are not supported (assumptions:1-2)", l + 1, L.nx, L.ny, L.nz);
  }
  // gather groups: ranks with my colour, ordered by (key, rank)  (MPI_COMM_SPLIT, mg_grids.f90:717)
  for (int l = 1; l < S.nlevs; l++) {
    Level &L = S.lev[l];
    if (!L.gather) continue;
    std::vector<std::pair<int, int>> mem;
    for (int r = 0; r < S.nranks; r++) {
      std::vector<Level> T(S.nlevs);
      T[0].nx = nx; T[0].ny = ny; T[0].nz = nz;
      rank_level_table(r, T, npx, npy, S.par.nsmall);
      if (T[l].color == L.color) mem.push_back({T[l].key, r});
    }
    std::sort(mem.begin(), mem.end());
    L.ngroup = (int)mem.size();
    if (L.ngroup != L.ngx * L.ngy) return fail("gather group of level %d has %d members, expected %d", l + 1, L.ngroup, L.ngx * L.ngy);
    for (int q = 0; q < L.ngroup; q++) L.group[q] = mem[q].second;
  }
  // allocations
  size_t max_part = 1;
  for (int l = 0; l < S.nlevs; l++) {
    Level &L = S.lev[l];
    make_view(L.v, L.nx, L.ny, L.nz);
    L.n3js = (size_t)(L.nx + 2) * L.v.plane;
    CHK(dmalloc(&L.v.p, L.n3js)); CHK(dmalloc(&L.v.b, L.n3js)); CHK(dmalloc(&L.v.r, L.n3js));
    for (int s = 0; s < 8; s++) CHK(dmalloc(&L.v.cA[s], L.n3js));
    CHK(dmalloc(&L.v.bet, L.n3js)); CHK(dmalloc(&L.v.gam, L.n3js));
    CHK(dmalloc(&L.v.p1, (size_t)(L.nx + 2) * L.v.RS)); CHK(dmalloc(&L.p1b, (size_t)(L.nx + 2) * L.v.RS)); L.v.p1w = nullptr;
    { double *q = nullptr; CHK(dmalloc(&q, (size_t)(L.nx + 2) / 2 + 1)); L.ksp_done = (unsigned int *)q; L.ksp_seq = 0; }  // zeroed by dmalloc
    CHK(dmalloc(&L.zy_store, L.n3js)); CHK(dmalloc(&L.zx_store, L.n3js));
    L.v.zy = L.v.zx = nullptr;
    for (int q = 0; q < 7; q++) CHK(dmalloc(&L.f2d_store[q], (size_t)(L.nx + 2) * L.v.RS));
    for (int q = 0; q < 2; q++) CHK(dmalloc(&L.tab_store[q], (size_t)L.nz + 1));
    L.v.m4 = L.v.d4 = L.v.m7 = L.v.d7 = L.v.h2 = L.v.hi2 = L.v.ze2 = nullptr; L.v.cffw = L.v.csw = nullptr;
    for (int q = 0; q < 2; q++) CHK(dmalloc(&L.zg_store[q], (size_t)(L.nx + 2) * L.v.RS));
    for (int q = 2; q < 4; q++) CHK(dmalloc(&L.zg_store[q], (size_t)L.nz + 1));
    L.v.dx2 = L.v.dy2 = nullptr; L.v.cffr = L.v.csr = nullptr;
    L.v.gk = L.v.ag58 = L.v.u1 = nullptr; L.v.d0w = nullptr;
    if (S.method == M_RB && S.real) {  // sequential-order red-black (mgx_rbseq.hip): +8 B per cell
      CHK(dmalloc(&L.v.gk, L.n3js));
      CHK(dmalloc(&L.gdec, (size_t)L.nz)); L.gdec_h.assign((size_t)L.nz, 0.0);
      CHK(dmalloc(&L.v.ag58, (size_t)2 * (L.nx + 2) * L.v.RS)); CHK(dmalloc(&L.v.u1, (size_t)(L.nx + 2) * L.v.RS));
      { double *q = nullptr; CHK(dmalloc(&q, (size_t)(L.nx / 8 + 2) * 8 + 16)); L.rbs_flag = (unsigned int *)q; L.rbs_seq = 0; }  // one word per chunk of 8 planes, 64 B apart (zeroed by dmalloc): what the walk has handed to the correction workers
    }
    const size_t n2 = (size_t)(L.ny + 2) * (L.nx + 2);
    L.g.nx = L.nx; L.g.ny = L.ny; L.g.nz = L.nz;
    CHK(dmalloc(&L.g.dx, n2)); CHK(dmalloc(&L.g.dy, n2)); CHK(dmalloc(&L.g.zeta, n2)); CHK(dmalloc(&L.g.h, n2));
    CHK(dmalloc(&L.g.rmask, n2)); L.g.bmask = 0;
    CHK(dmalloc(&L.g.zr, (size_t)(L.ny + 4) * (L.nx + 4) * L.nz));
    CHK(dmalloc(&L.g.zw, (size_t)(L.ny + 4) * (L.nx + 4) * (L.nz + 1)));
    CHK(dmalloc(&L.g.cw, n2 * (L.nz + 1)));
    L.g.dzw = L.g.zxdy = L.g.zydx = nullptr;
    L.g.mzw = L.g.mdzw = L.g.mzxdy = L.g.mzydx = L.g.mcw = L.g.mdx = L.g.mdy = L.g.mrmask = nullptr;
    if (l == 0) {
      CHK(dmalloc(&L.g.dzw, n2 * (L.nz + 1))); CHK(dmalloc(&L.g.zxdy, n2 * L.nz)); CHK(dmalloc(&L.g.zydx, n2 * L.nz));
      CHK(dmalloc(&L.g.mzw, n2 * (L.nz + 1))); CHK(dmalloc(&L.g.mdzw, n2 * (L.nz + 1))); CHK(dmalloc(&L.g.mcw, n2 * (L.nz + 1)));
      CHK(dmalloc(&L.g.mzxdy, n2 * L.nz)); CHK(dmalloc(&L.g.mzydx, n2 * L.nz));
      CHK(dmalloc(&L.g.mdx, n2)); CHK(dmalloc(&L.g.mdy, n2)); CHK(dmalloc(&L.g.mrmask, n2));
    }
    if (L.gather) {
      const int nxc = L.nx / L.ngx, nyc = L.ny / L.ngy;
      L.vs = L.v;
      make_view(L.vs, nxc, nyc, L.nz);
      const size_t ns = (size_t)(nxc + 2) * L.vs.plane;
      CHK(dmalloc(&L.vs.b, ns)); CHK(dmalloc(&L.vs.p, ns));
      const size_t Ng = (size_t)L.nz * (nyc + 2) * (nxc + 2);
      CHK(dmalloc(&L.blk, Ng)); CHK(dmalloc(&L.gbuf, Ng * L.ngroup));
      for (int q = 0; q < 4; q++) CHK(dmalloc(&L.tmp2[q], (size_t)(nyc + 2) * (nxc + 2)));
    }
    size_t np = (size_t)mgxk_residual_nblocks(&L.v);
    if (l == 1) { const size_t nf = (size_t)mgxk_residual_restrict_grid(&S.lev[0].v, &L.v); if (nf > np) np = nf; }  // the fused closing residual of solve_p
    if (np > max_part) max_part = np;
  }
  S.npartial = (int)max_part;
  CHK(dmalloc(&S.d_partial, max_part));
  CHK(dmalloc(&S.d_scalar, 8));
  if (S.method == M_RB && S.real) CHK(dmalloc(&S.rho_dev, 32));
  HIPCHK(hipHostMalloc((void **)&S.h_scalar, 8 * sizeof(double)));
  HIPCHK(hipHostMalloc((void **)&S.kerr, 64, hipHostMallocMapped));
  *S.kerr = 0;
  if (S.nranks > 1) {
    HIPCHK(hipStreamCreateWithFlags(&S.stream2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&S.ev_a, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&S.ev_s, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&S.ev_x, hipEventDisableTiming));
  }
  Level &L1 = S.lev[0];
  S.ref_scratch_n = (size_t)8 * L1.nz * (L1.ny + 2) * (L1.nx + 2);
  CHK(dmalloc(&S.ref_scratch, S.ref_scratch_n));
  CHK(dmalloc(&S.slope_scratch, S.ref_scratch_n / 4));  // 2 x level-1 field: the slopes zy, zx in the reference layout
  for (auto &L : S.lev) { L.g.cA = S.ref_scratch; L.g.szy = S.slope_scratch; L.g.szx = nullptr; }
  S.xbuf_n = (size_t)(L1.nz + 1) * 2 * ((L1.nx > L1.ny ? L1.nx : L1.ny) + 4);
  if (S.par.bmask && S.nranks > 1) {  // the 4-D cA halo of define_matrix travels through the same buffers
    const size_t n4 = (size_t)8 * L1.nz * (L1.nx > L1.ny ? L1.nx : L1.ny);
    if (n4 > S.xbuf_n) S.xbuf_n = n4;
  }
  for (int q = 0; q < 16; q++) CHK(dmalloc(&S.xbuf[q], S.xbuf_n));
  CHK(dmalloc(&S.d_u, (size_t)(L1.nx + 1) * (L1.ny + 2) * L1.nz));
  CHK(dmalloc(&S.d_v, (size_t)(L1.nx + 2) * (L1.ny + 1) * L1.nz));
  CHK(dmalloc(&S.d_w, (size_t)(L1.nx + 2) * (L1.ny + 2) * (L1.nz + 1)));
  CHK(dmalloc(&S.d_fx, (size_t)(L1.nx + 2) * (L1.ny + 2) * (L1.nz + 1)));
  CHK(dmalloc(&S.d_fy, (size_t)(L1.nx + 2) * (L1.ny + 2) * (L1.nz + 1)));
  CHK(dmalloc(&S.d_fz, (size_t)(L1.nx + 2) * (L1.ny + 2) * (L1.nz + 1)));
  CHK(dmalloc(&S.d_bm, (size_t)(L1.nx + 2) * (L1.ny + 2) * L1.nz));
  CHK(dmalloc(&S.d_rmask_ref, (size_t)(L1.nx + 2) * (L1.ny + 2))); CHK(dmalloc(&S.d_rmask_m, (size_t)(L1.nx + 2) * (L1.ny + 2)));
  S.call_mask = false;
  CHK(sync_stream());
  S.use_small = getenv("MGX_NO_SMALL") ? 0 : 1;
  S.no_mf = getenv("MGX_NO_MF") ? 1 : 0;
  if (getenv("MGX_C2F_NOSKIP")) S.c2f_skip = 0;
  if (getenv("MGX_NO_FUSE_CLOSING")) S.fuse_closing = 0;
  if (getenv("MGX_NO_RESTRICT_CHAIN")) S.use_chain = 0;
  if (getenv("MGX_NO_RBSEQ_FUSE")) S.rbseq_fuse = 0;
  if (getenv("MGX_NO_RBSEQ_WINDOW")) S.rbseq_window = 0;
  if (getenv("MGX_COARSEST_DIRECT")) S.coarsest_direct = atoi(getenv("MGX_COARSEST_DIRECT"));
  if (getenv("MGX_OVERLAP")) S.overlap = atoi(getenv("MGX_OVERLAP"));
  if (getenv("MGX_NO_KSP")) S.use_ksp = 0;
  if (getenv("MGX_P2P_TIMEOUT_MS")) (void)mgxk_set_p2p_timeout(atof(getenv("MGX_P2P_TIMEOUT_MS")));
  if (getenv("MGX_EXACT_HALOS")) S.exact_halos = 1;
  if (getenv("MGX_TICTOC")) S.tictoc = 1;
  if (getenv("MGX_RB_EXACT")) S.rb_exact = atoi(getenv("MGX_RB_EXACT"));
  if (getenv("MGX_RB_SEQ")) S.rb_seq = atoi(getenv("MGX_RB_SEQ"));
  S.inited = true;
  if (S.verbose && S.rank == 0) {  // read_nhnamelist prints (mg_namelist.f90:108-124) and print_grids (mg_grids.f90:741-762)
    printf(" Non hydrostatic parameters:\n   - solver_prec   : %g\n   - solver_maxiter: %d\n   - nsmall        : %d\n   - ns_coarsest   : %d\n"
           "   - ns_pre        : %d\n   - ns_post       : %d\n   - cmatrix       : %s\n   - relax_method  : %s\n   - interp_type   : %s\n"
           "   - restrict_type : %s\n   - aggressive    : %c\n   - netcdf_output : %c\n   - bmask         : %c\n\n",
           p.solver_prec, p.solver_maxiter, p.nsmall, p.ns_coarsest, p.ns_pre, p.ns_post, p.cmatrix, p.relax_method, p.interp_type,
           p.restrict_type, p.aggressive ? 'T' : 'F', p.netcdf_output ? 'T' : 'F', p.bmask ? 'T' : 'F');
    printf(" - print grid information:\n");
    for (int l = 0; l < S.nlevs; l++) {
      const Level &L = S.lev[l];
      printf("  lev=%2d: %3d x%3d x%3d on %3d x%3d procs%s\n", l + 1, L.nx, L.ny, L.nz, L.npx, L.npy, L.gather ? " / gather" : "");
    }
  }
  return 0;
}

int mgx_matrices(const double *dx, const double *dy, const double *zeta, const double *h, const double *rmask, double hc,
                 double theta_b, double theta_s) {
  NEED_INIT();
  if (S.par.bmask && !rmask) return fail("bmask=.true. needs rmask in mgx_matrices (nhydro.f90:52-55)");
  if (S.verbose && S.rank == 0) printf("  nhydro_matrices:\n");
  S.hlim = hc; S.theta_b = theta_b; S.theta_s = theta_s;
  Level &L = S.lev[0];
  const size_t n2 = (size_t)(L.ny + 2) * (L.nx + 2) * sizeof(double);
  HIPCHK(hipMemcpyAsync(L.g.dx, dx, n2, hipMemcpyHostToDevice, S.stream));
  HIPCHK(hipMemcpyAsync(L.g.dy, dy, n2, hipMemcpyHostToDevice, S.stream));
  HIPCHK(hipMemcpyAsync(L.g.zeta, zeta, n2, hipMemcpyHostToDevice, S.stream));
  HIPCHK(hipMemcpyAsync(L.g.h, h, n2, hipMemcpyHostToDevice, S.stream));
  if (S.par.bmask) HIPCHK(hipMemcpyAsync(L.g.rmask, rmask, n2, hipMemcpyHostToDevice, S.stream));  // grid(1)%rmask = rmask
  return define_matrices();
}

int mgx_compute_rhs(const double *u, const double *v, const double *w, const double *rmask) {
  NEED_INIT();
  if (!S.have_matrix) return fail("mgx_matrices must be called before compute_rhs");
  CHK(set_call_mask(rmask, false));
  CHK(upload_uvw(u, v, w));
  CHK(compute_rhs_dev());
  CHK(sync_stream());
  return 0;
}

int mgx_solve(double *u, double *v, double *w, const double *rmask) {
  NEED_INIT();
  if (!S.have_matrix) return fail("mgx_matrices must be called before mgx_solve");
  if (S.verbose && S.rank == 0) printf("  nhydro_solve:\n");
  CHK(set_call_mask(rmask, false));
  CHK(upload_uvw(u, v, w));
  CHK(compute_rhs_dev());
  CHK(solve_p(S.par.solver_prec, S.par.solver_maxiter, nullptr, nullptr, nullptr));
  Level &L = S.lev[0];
  CHK(correct_uvw_dev());
  const size_t nu = (size_t)(L.nx + 1) * (L.ny + 2) * L.nz, nv = (size_t)(L.nx + 2) * (L.ny + 1) * L.nz, nw = (size_t)(L.nx + 2) * (L.ny + 2) * (L.nz + 1);
  HIPCHK(hipMemcpyAsync(u, S.d_u, nu * sizeof(double), hipMemcpyDeviceToHost, S.stream));
  HIPCHK(hipMemcpyAsync(v, S.d_v, nv * sizeof(double), hipMemcpyDeviceToHost, S.stream));
  HIPCHK(hipMemcpyAsync(w, S.d_w, nw * sizeof(double), hipMemcpyDeviceToHost, S.stream));
  CHK(sync_stream());
  return 0;
}

// Device-resident variant of nhydro_solve: u,v,w are DEVICE pointers in the model's (i,j,k) layout
// (e.g. torch tensors); nothing crosses PCIe.  The library's own staging copies are bypassed.
int mgx_solve_device(double *u_dev, double *v_dev, double *w_dev, const double *rmask) {
  NEED_INIT();
  if (!S.have_matrix) return fail("mgx_matrices must be called before mgx_solve_device");
  CHK(set_call_mask(rmask, true));
  double *su = S.d_u, *sv = S.d_v, *sw = S.d_w;
  S.d_u = u_dev; S.d_v = v_dev; S.d_w = w_dev;
  int rc = compute_rhs_dev();
  if (!rc) rc = solve_p(S.par.solver_prec, S.par.solver_maxiter, nullptr, nullptr, nullptr);
  if (!rc) rc = correct_uvw_dev();
  if (!rc) rc = sync_stream();
  S.d_u = su; S.d_v = sv; S.d_w = sw;
  return rc;
}

int mgx_check_nondivergence(double *u, double *v, double *w, const double *rmask) {
  if (S.verbose && S.rank == 0) printf(" - check non-divergence:\n");
  return mgx_compute_rhs(u, v, w, rmask);
}

int mgx_solve_p(double tol, int maxite, int *nite, double *res, double *hist) {
  NEED_INIT();
  if (!S.have_matrix) return fail("no matrix: call mgx_matrices (or mgx_set_field(lev, MGX_CA, ...)) first");
  return solve_p(tol, maxite, nite, res, hist);
}
int mgx_fcycle(void) { NEED_INIT(); CHK(fcycle()); CHK(op_sync()); return 0; }
int mgx_vcycle(int lev) { NEED_LEV(lev); CHK(vcycle(lev)); CHK(op_sync()); return 0; }
int mgx_vcycle2(int lev1, int lev2) { NEED_LEV(lev1); NEED_LEV(lev2); if (lev2 < lev1) return fail("Vcycle2: lev2 < lev1"); CHK(vcycle2(lev1, lev2)); CHK(op_sync()); return 0; }
int mgx_relax(int lev, int nsweeps) { NEED_LEV(lev); CHK(relax(lev, nsweeps)); CHK(op_sync()); return 0; }
int mgx_residual(int lev, double *res) { NEED_LEV(lev); double r; CHK(residual(lev, &r)); if (res) *res = r; return 0; }
int mgx_fine2coarse(int lev) { NEED_LEV(lev); if (lev >= S.nlevs) return fail("fine2coarse(%d): no coarser level", lev); CHK(fine2coarse(lev)); CHK(op_sync()); return 0; }
int mgx_coarse2fine(int lev) { NEED_LEV(lev); if (lev >= S.nlevs) return fail("coarse2fine(%d): no coarser level", lev); CHK(coarse2fine(lev)); CHK(op_sync()); return 0; }
// the generic fill_halo(lev, field) of mg_mpi_exchange.f90:10-16: 3-D solver fields p, b, r (fill_halo_3D[_relax], nh = 1), the 2-D
// geometry dx, dy, zeta, h (fill_halo_2D), zr / zw (fill_halo_3D with nh = 2: extrapolation at physical sides, :956-964) and the
// 4-D cA (fill_halo_4D: neighbour exchange only).  Collective over the ranks.
int mgx_fill_halo(int lev, int field) {
  NEED_LEV(lev);
  Level &L = S.lev[lev - 1];
  switch (field) {
    case MGX_P: CHK(fill_halo_js(L, L.v.p)); break;
    case MGX_B: CHK(fill_halo_js(L, L.v.b)); L.b_halo_stale = false; break;
    case MGX_R: CHK(fill_halo_js(L, L.v.r)); L.r_halo_stale = false; break;
    case MGX_CA: for (int s = 0; s < 8; s++) CHK(fill_halo_js(L, L.v.cA[s], true, true)); break;
    case MGX_DX: CHK(rl_fill_halo(L, L.g.dx, 1, 1, 0)); break;
    case MGX_DY: CHK(rl_fill_halo(L, L.g.dy, 1, 1, 0)); break;
    case MGX_ZETA: CHK(rl_fill_halo(L, L.g.zeta, 1, 1, 0)); break;
    case MGX_H: CHK(rl_fill_halo(L, L.g.h, 1, 1, 0)); break;
    case MGX_ZR: CHK(rl_fill_halo(L, L.g.zr, L.nz, 2, 0)); break;
    case MGX_ZW: CHK(rl_fill_halo(L, L.g.zw, L.nz + 1, 2, 0)); break;
    default: return fail("fill_halo: field %d has no halo rule (p, b, r, cA, dx, dy, zeta, h, zr, zw)", field);
  }
  CHK(op_sync());  // option "async": enqueued only, like the cycles
  return 0;
}

// testgalerkin(lev) (mg_solvers.f90:203-288): energy of a coarse field under the coarse operator against the energy of its
// interpolation under the fine one.  The reference fills grid(lev)%p with random_number; here the caller provides it
// (mgx_set_field(lev, MGX_P, ...)), everything after that is the reference's sequence.  b of both levels is zeroed, as there.
int mgx_testgalerkin(int lev, double *norm_c, double *norm_f) {
  NEED_LEV(lev);
  if (lev < 2) return fail("testgalerkin(%d): needs a finer level lev-1", lev);
  if (!S.have_matrix) return fail("testgalerkin: no matrix");
  double nc = 0, nf = 0;
  for (int pass = 0; pass < 2; pass++) {
    Level &L = S.lev[pass == 0 ? lev - 1 : lev - 2];
    const int l = pass == 0 ? lev : lev - 1;
    if (pass == 0) CHK(fill_halo_js(L, L.v.p));                                   // call fill_halo(lev,grid(lev)%p)
    else {
      HIPCHK(hipMemsetAsync(L.v.p, 0, L.n3js * sizeof(double), S.stream));        // grid(lev-1)%p = 0
      CHK(coarse2fine(l));                                                         // interpolate p to r and add r to p
    }
    HIPCHK(hipMemsetAsync(L.v.b, 0, L.n3js * sizeof(double), S.stream));          // grid(.)%b = 0
    CHK(residual(l, nullptr));
    mgxk_dot(S.stream, &L.v, L.v.p, L.v.r, S.d_partial, S.d_scalar); S.n_launch += 2;  // norm(lev,p,r,...) -> global_sum
    double s; CHK(global_sum(L, &s));
    (pass == 0 ? nc : nf) = s;
  }
  if (S.verbose && S.rank == 0)
    printf(" ======== lev %12d ===========\n norm coarse = %24.16E\n norm fine   = %24.16E\n ratio       = %24.16E\n", lev, nc, nf / 4, nc / nf * 4);
  if (norm_c) *norm_c = nc;
  if (norm_f) *norm_f = nf;
  return 0;
}

int mgx_level_table(int nx, int ny, int nz, int npx, int npy, int rank, int nsmall, int maxlev, int *out) {
  if (nx < 2 || ny < 2 || nz < 2 || npx < 1 || npy < 1 || rank < 0 || rank >= npx * npy) return -1;
  const int nl = find_grid_levels(npx, npy, nx, ny, nz);
  if (nl < 1 || nl > maxlev) return -1;
  std::vector<Level> T(nl);
  T[0].nx = nx; T[0].ny = ny; T[0].nz = nz;
  rank_level_table(rank, T, npx, npy, nsmall);
  for (int l = 0; l < nl; l++) {
    const Level &L = T[l];
    const int v[12] = {L.nx, L.ny, L.nz, L.npx, L.npy, L.incx, L.incy, L.gather, L.ngx, L.ngy, L.key, L.color};
    memcpy(out + 20 * l, v, sizeof(v)); memcpy(out + 20 * l + 12, L.neighb, 8 * sizeof(int));
  }
  return nl;
}

int mgx_set_option(const char *name, int value) {
  if (streq(name, "warm_start")) S.warm_start = value;
  else if (streq(name, "tictoc")) S.tictoc = value;
  else if (streq(name, "exact_halos")) S.exact_halos = value;
  else if (streq(name, "verbose")) S.verbose = value;
  else if (streq(name, "rb_chain")) S.rb_chain = value;
  else if (streq(name, "rb_exact")) S.rb_exact = value;
  else if (streq(name, "rb_seq")) S.rb_seq = value;
  else if (streq(name, "keep_r")) S.keep_r = value;
  else if (streq(name, "c2f_skip")) S.c2f_skip = value;
  else if (streq(name, "fuse_closing")) S.fuse_closing = value;
  else if (streq(name, "restrict_chain")) S.use_chain = value;
  else if (streq(name, "rbseq_fuse")) S.rbseq_fuse = value;
  else if (streq(name, "rbseq_window")) S.rbseq_window = value;
  else if (streq(name, "rbseq_rowcut")) S.rbseq_rowcut = value;
  else if (streq(name, "coarsest_direct")) S.coarsest_direct = value;
  else if (streq(name, "overlap")) S.overlap = value;
  else if (streq(name, "ksp")) { S.use_ksp = value; if (value) S.ksp_down = 0; }  // switching it on again also clears a time-out of this solver
  else if (streq(name, "async")) S.async_ops = value;
  else if (streq(name, "fuse_tail")) S.use_fuse = value;
  else if (streq(name, "ksp_test_stall")) S.ksp_test_stall = value;
  else if (streq(name, "rbseq_test_stall")) S.rbseq_test_stall = value;
  else if (streq(name, "rbseq_fuse_min")) S.rbseq_fuse_min = value;
  else if (streq(name, "rbseq_d0_in_pass")) S.rbseq_d0_in_pass = value;
  else if (streq(name, "rbseq_timeout_ms")) { if (mgxk_set_rbseq_timeout((double)value)) return fail("rbseq_timeout_ms: could not set the device constant"); }
  else if (streq(name, "ksp_timeout_ms")) { if (mgxk_set_ksp_timeout((double)value)) return fail("ksp_timeout_ms: could not set the device constant"); }
  else if (streq(name, "p2p_test_drop")) S.p2p_test_drop = value;
  else if (streq(name, "p2p_timeout_ms")) { if (mgxk_set_p2p_timeout((double)value)) return fail("p2p_timeout_ms: could not set the device constant"); }
  else if (streq(name, "p2p")) {  // collective: every rank switches together, between exchanges
    if (value && !S.p2p_ready) return fail("p2p: mgx_p2p_prepare / mgx_p2p_connect have not been called");
    S.p2p_on = value != 0;
    // the ranks decide this together (it is collective), so whatever a rank remembered about its own waits is settled here
    S.p2p_failed = 0;
    if (S.p2p_err) *S.p2p_err = 0;
  }
  else return fail("unknown option '%s'", name);
  return 0;
}

// read back a namelist member (the reference's drivers `use mg_namelist` and read e.g. `bmask` directly) or an option
int mgx_get_option(const char *name, int *value) {
  if (!value) return fail("mgx_get_option: value is NULL");
  if (streq(name, "bmask")) *value = S.par.bmask;
  else if (streq(name, "nsmall")) *value = S.par.nsmall;
  else if (streq(name, "solver_maxiter")) *value = S.par.solver_maxiter;
  else if (streq(name, "ns_coarsest")) *value = S.par.ns_coarsest;
  else if (streq(name, "ns_pre")) *value = S.par.ns_pre;
  else if (streq(name, "ns_post")) *value = S.par.ns_post;
  else if (streq(name, "netcdf_output")) *value = S.par.netcdf_output;
  else if (streq(name, "aggressive")) *value = S.par.aggressive;
  else if (streq(name, "warm_start")) *value = S.warm_start;
  else if (streq(name, "tictoc")) *value = S.tictoc;
  else if (streq(name, "exact_halos")) *value = S.exact_halos;
  else if (streq(name, "verbose")) *value = S.verbose;
  else if (streq(name, "rb_chain")) *value = S.rb_chain;
  else if (streq(name, "rb_exact")) *value = S.rb_exact;
  else if (streq(name, "rb_seq")) *value = S.rb_seq;
  else if (streq(name, "keep_r")) *value = S.keep_r;
  else if (streq(name, "c2f_skip")) *value = S.c2f_skip;
  else if (streq(name, "fuse_closing")) *value = S.fuse_closing;
  else if (streq(name, "restrict_chain")) *value = S.use_chain;
  else if (streq(name, "rbseq_fuse")) *value = S.rbseq_fuse;
  else if (streq(name, "rbseq_window")) *value = S.rbseq_window;
  else if (streq(name, "rbseq_rowcut")) *value = S.rbseq_rowcut;
  else if (streq(name, "coarsest_direct")) *value = S.coarsest_direct;
  else if (streq(name, "coarsest_direct_solves")) *value = (int)S.n_direct;
  else if (streq(name, "rbseq_window_colours")) *value = (int)S.n_window;
  else if (streq(name, "rbseq_fuse_min")) *value = S.rbseq_fuse_min;
  else if (streq(name, "rbseq_d0_in_pass")) *value = S.rbseq_d0_in_pass;
  else if (streq(name, "overlap")) *value = S.overlap;
  else if (streq(name, "overlapped_passes")) *value = (int)S.n_overlap;
  else if (streq(name, "ksp")) *value = (S.use_ksp && !S.ksp_down) ? 1 : 0;
  else if (streq(name, "async")) *value = S.async_ops;
  else if (streq(name, "fuse_tail")) *value = S.use_fuse;
  else if (streq(name, "p2p_failed")) *value = S.p2p_failed;
  else if (streq(name, "p2p")) *value = S.p2p_on ? 1 : 0;
  else return fail("unknown option '%s'", name);
  return 0;
}

// print_tictoc (mg_tictoc.f90:114-153): name, total and per-level seconds, then the call counts
int mgx_print_tictoc(const char *path) {
  tt_collect();
  FILE *f = fopen(path ? path : "fort.10", "w");
  if (!f) return fail("cannot open %s", path ? path : "fort.10");
  // the reference's formats (mg_tictoc.f90:128-150): t22 + A10, (x,I9) per level; per timer (x,A20), (x,E9.3) total and per level, then
  // the call counts under them -- byte for byte what flang writes (tests/golden/ref_tictoc.txt), E9.3 in Fortran's 0.dddE+ee form
  fprintf(f, "%21s%10s", "", "Total");
  for (int l = 1; l <= S.tt_nblev; l++) fprintf(f, " %9d", l);
  fprintf(f, "\n");
  for (size_t q = 0; q < S.tt_names.size(); q++) {
    double tot = 0; long long nc = 0;
    for (int l = 0; l < S.tt_nblev; l++) { tot += S.tt_time[l][q]; nc += S.tt_calls[l][q]; }
    fprintf(f, " %20s %s", S.tt_names[q].c_str(), fortran_e3(tot, 9).c_str());
    for (int l = 0; l < S.tt_nblev; l++) fprintf(f, " %s", fortran_e3(S.tt_time[l][q], 9).c_str());
    fprintf(f, "\n%21s %9lld", "", nc);
    for (int l = 0; l < S.tt_nblev; l++) fprintf(f, " %9lld", S.tt_calls[l][q]);
    fprintf(f, "\n");
  }
  fclose(f);
  return 0;
}

int mgx_tic(int lev, const char *name) {
  if (lev < 1 || lev > 32 || !name) return fail("tic: level %d outside 1..32", lev);
  const int sub = tt_sub(name);
  if (sub >= 32) return fail("tic: more than 32 timer names (mg_tictoc.f90: submax)");
  S.tt_host.push_back(HostTic{lev, sub, std::chrono::steady_clock::now()});
  return 0;
}
int mgx_toc(int lev, const char *name) {
  if (lev < 1 || lev > 32 || !name) return fail("toc: level %d outside 1..32", lev);
  const int sub = tt_sub(name);
  for (int q = (int)S.tt_host.size() - 1; q >= 0; q--)
    if (S.tt_host[q].lev == lev && S.tt_host[q].sub == sub) {
      if (S.inited) (void)hipStreamSynchronize(S.stream);
      S.tt_time[lev - 1][sub] += std::chrono::duration<double>(std::chrono::steady_clock::now() - S.tt_host[q].t0).count();
      S.tt_calls[lev - 1][sub]++;
      if (lev > S.tt_nblev) S.tt_nblev = lev;
      S.tt_host.erase(S.tt_host.begin() + q);
      return 0;
    }
  return fail("toc(%d,'%s') without a matching tic", lev, name);  // the reference prints "Error: tictoc" and goes on (mg_tictoc.f90:104-108)
}

// wait for everything enqueued on the solver's stream and report device-side errors (time-outs, rejected launches); the counterpart of option "async"
int mgx_synchronize(void) { NEED_INIT(); return sync_stream(); }
int mgx_nlevs(void) { return S.inited ? S.nlevs : 0; }
int mgx_level_dims(int lev, int *nx, int *ny, int *nz) { NEED_LEV(lev); const Level &L = S.lev[lev - 1]; *nx = L.nx; *ny = L.ny; *nz = L.nz; return 0; }
int mgx_rbseq_window_info(int lev, double *rho, int *planes) { NEED_LEV(lev); const Level &L = S.lev[lev - 1]; *rho = L.rbs_rho; *planes = L.rbs_m; return 0; }
int mgx_rbseq_window_rows(int lev, int *rows) { NEED_LEV(lev); *rows = S.lev[lev - 1].rbs_rows; return 0; }
int mgx_level_info(int lev, int *out) {
  NEED_LEV(lev);
  const Level &L = S.lev[lev - 1];
  const int v[10] = {L.npx, L.npy, L.incx, L.incy, L.gather, L.ngx, L.ngy, L.key, L.color, 0};
  memcpy(out, v, sizeof(v)); memcpy(out + 10, L.neighb, 8 * sizeof(int));
  return 0;
}

static int field_ptr(Level &L, int field, double **a, size_t *n) {
  const size_t n2 = (size_t)(L.ny + 2) * (L.nx + 2);
  switch (field) {
    case MGX_DX: *a = L.g.dx; *n = n2; return 0;
    case MGX_DY: *a = L.g.dy; *n = n2; return 0;
    case MGX_ZETA: *a = L.g.zeta; *n = n2; return 0;
    case MGX_H: *a = L.g.h; *n = n2; return 0;
    case MGX_ZR: *a = L.g.zr; *n = (size_t)(L.ny + 4) * (L.nx + 4) * L.nz; return 0;
    case MGX_ZW: *a = L.g.zw; *n = (size_t)(L.ny + 4) * (L.nx + 4) * (L.nz + 1); return 0;
    case MGX_CW: *a = L.g.cw; *n = n2 * (L.nz + 1); return 0;
    case MGX_RMASK: *a = L.g.rmask; *n = n2; return 0;
  }
  return 1;
}

int mgx_get_field(int lev, int field, double *host) {
  NEED_LEV(lev);
  Level &L = S.lev[lev - 1];
  const size_t n3 = (size_t)L.nz * (L.ny + 2) * (L.nx + 2);
  double *a; size_t n;
  if (field == MGX_P || field == MGX_B || field == MGX_R) {
    double *js = field == MGX_P ? L.v.p : (field == MGX_B ? L.v.b : L.v.r);
    if (field == MGX_R && L.r_halo_stale) { CHK(fill_halo_js(L, L.v.r, true)); L.r_halo_stale = false; }
    if (field == MGX_B && L.b_halo_stale) { CHK(fill_halo_js(L, L.v.b, true)); L.b_halo_stale = false; }
    mgxk_convert(S.stream, &L.v, js, S.ref_scratch, 1, 0, 1);
    HIPCHK(hipMemcpyAsync(host, S.ref_scratch, n3 * sizeof(double), hipMemcpyDeviceToHost, S.stream));
  } else if (field == MGX_CA) {
    for (int s = 0; s < 8; s++) mgxk_convert(S.stream, &L.v, L.v.cA[s], S.ref_scratch, 8, s, 1);
    HIPCHK(hipMemcpyAsync(host, S.ref_scratch, 8 * n3 * sizeof(double), hipMemcpyDeviceToHost, S.stream));
  } else if (!field_ptr(L, field, &a, &n)) {
    HIPCHK(hipMemcpyAsync(host, a, n * sizeof(double), hipMemcpyDeviceToHost, S.stream));
  } else return fail("get_field: unknown field id %d", field);
  CHK(sync_stream());
  return 0;
}

int mgx_set_field(int lev, int field, const double *host) {
  NEED_LEV(lev);
  Level &L = S.lev[lev - 1];
  const size_t n3 = (size_t)L.nz * (L.ny + 2) * (L.nx + 2);
  double *a; size_t n;
  if (field == MGX_P || field == MGX_B || field == MGX_R) {
    double *js = field == MGX_P ? L.v.p : (field == MGX_B ? L.v.b : L.v.r);
    HIPCHK(hipMemcpyAsync(S.ref_scratch, host, n3 * sizeof(double), hipMemcpyHostToDevice, S.stream));
    mgxk_convert(S.stream, &L.v, js, S.ref_scratch, 1, 0, 0);
  } else if (field == MGX_CA) {
    HIPCHK(hipMemcpyAsync(S.ref_scratch, host, 8 * n3 * sizeof(double), hipMemcpyHostToDevice, S.stream));
    for (int s = 0; s < 8; s++) mgxk_convert(S.stream, &L.v, L.v.cA[s], S.ref_scratch, 8, s, 0);
    mgxs_pivots(S.stream, &L.v);
    if (L.v.gk) mgxk_rbseq_setup(S.stream, &L.v);
    if (L.v.gk && S.rho_dev && lev <= 32) {
      HIPCHK(hipMemsetAsync(S.rho_dev + lev - 1, 0, sizeof(double), S.stream));
      mgxk_rbseq_rho(S.stream, &L.v, S.rho_dev + lev - 1);
      if (L.gdec) {
        HIPCHK(hipMemsetAsync(L.gdec, 0, (size_t)L.nz * sizeof(double), S.stream));
        mgxk_rbseq_gdecay(S.stream, &L.v, L.gdec);
        HIPCHK(hipMemcpyAsync(L.gdec_h.data(), L.gdec, (size_t)L.nz * sizeof(double), hipMemcpyDeviceToHost, S.stream));
      }
      HIPCHK(hipMemcpyAsync(S.rho_host, S.rho_dev, sizeof S.rho_host, hipMemcpyDeviceToHost, S.stream));
      CHK(sync_stream());
      set_window_planes();
    }
    L.v.zy = L.v.zx = nullptr;  // a user-supplied matrix is used as stored
    L.v.m4 = nullptr;
    S.cd_valid = 0;
    S.have_matrix = true;
  } else if (!field_ptr(L, field, &a, &n)) {
    HIPCHK(hipMemcpyAsync(a, host, n * sizeof(double), hipMemcpyHostToDevice, S.stream));
  } else return fail("set_field: unknown field id %d", field);
  CHK(sync_stream());
  return 0;
}

static int time_op(int lev, int reps, float *ms, int which) {
  NEED_LEV(lev);
  if (reps < 1) return fail("reps must be >= 1");
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, S.stream));
  for (int q = 0; q < reps; q++) { if (which == 0) CHK(relax(lev, 1)); else CHK(residual(lev, nullptr)); }
  HIPCHK(hipEventRecord(e1, S.stream));
  HIPCHK(hipEventSynchronize(e1));
  float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *ms = t / reps;
  return 0;
}
// a[i] / b[i] by the hardware division sequence against the refined-reciprocal quotient the colour pass uses for its per-column
// divisors (DIVC, mgx_device.h): *nbad = number of pairs whose bits differ.  Needs no mgx_init.
int mgx_selftest_divc(const double *a, const double *b, int n, long long *nbad) {
  if (n < 1) return fail("mgx_selftest_divc: n must be >= 1");
  double *da = nullptr, *db = nullptr; unsigned long long *dbad = nullptr, h = 0;
  HIPCHK(hipMalloc((void **)&da, (size_t)n * sizeof(double))); HIPCHK(hipMalloc((void **)&db, (size_t)n * sizeof(double))); HIPCHK(hipMalloc((void **)&dbad, sizeof h));
  HIPCHK(hipMemcpy(da, a, (size_t)n * sizeof(double), hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(db, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dbad, 0, sizeof h));
  mgxk_divc_selftest(nullptr, da, db, n, dbad);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(&h, dbad, sizeof h, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dbad);
  *nbad = (long long)h;
  return 0;
}
int mgx_time_relax(int lev, int reps, float *ms) { return time_op(lev, reps, ms, 0); }
int mgx_time_residual(int lev, int reps, float *ms) { return time_op(lev, reps, ms, 1); }
int mgx_counters(long long *out) { out[0] = S.n_launch; out[1] = S.n_halo; out[2] = S.n_exch; out[3] = S.n_allred; return 0; }
long long mgx_p2p_exchanges(void) { return S.n_p2p; }

int mgx_p2p_prepare(void *handles_out) {
  NEED_INIT();
  if (S.p2p_slab) return fail("mgx_p2p_prepare called twice");
  size_t off = 0;
  for (auto &L : S.lev)
    for (int d = 0; d < 8; d++) {
      const size_t c = (size_t)L.nz * ((d == 0 || d == 2) ? L.nx : ((d == 1 || d == 3) ? L.ny : 1));
      for (int par = 0; par < 2; par++) { L.p2p_off[d][par] = off; off += (c + 31) / 32 * 32; }
      L.p2p_seq = 0;
    }
  for (auto &L : S.lev) {
    L.p2p_gseq = 0; L.p2p_goff[0] = L.p2p_goff[1] = 0;
    if (!L.gather) continue;
    const size_t Ng = (size_t)L.nz * (L.vs.ny + 2) * (L.vs.nx + 2);
    for (int par = 0; par < 2; par++) { L.p2p_goff[par] = off; off += ((size_t)L.ngroup * Ng + 31) / 32 * 32; }
  }
  S.p2p_slab_n = off;
  HIPCHK(hipExtMallocWithFlags((void **)&S.p2p_slab, off * sizeof(double), hipDeviceMallocFinegrained));
  HIPCHK(hipExtMallocWithFlags((void **)&S.p2p_flags, 4096 * sizeof(unsigned long long), hipDeviceMallocFinegrained));
  HIPCHK(hipMalloc((void **)&S.p2p_counter, 64));
  HIPCHK(hipHostMalloc((void **)&S.p2p_err, 64, hipHostMallocMapped));
  *S.p2p_err = 0;
  HIPCHK(hipMemset(S.p2p_slab, 0, off * sizeof(double)));
  HIPCHK(hipMemset(S.p2p_flags, 0, 4096 * sizeof(unsigned long long)));
  HIPCHK(hipMemset(S.p2p_counter, 0, 64));
  HIPCHK(hipDeviceSynchronize());
  hipIpcMemHandle_t h[2];
  HIPCHK(hipIpcGetMemHandle(&h[0], S.p2p_slab));
  HIPCHK(hipIpcGetMemHandle(&h[1], S.p2p_flags));
  memcpy(handles_out, h, sizeof h);
  return 0;
}

int mgx_p2p_handle_bytes(void) { return (int)(2 * sizeof(hipIpcMemHandle_t)); }

// the receive slab and the flag page of THIS instance (after mgx_p2p_prepare), for ranks that live in the same process
int mgx_p2p_local_pointers(void **slab, void **flags) {
  NEED_INIT();
  if (!S.p2p_slab) return fail("mgx_p2p_local_pointers: call mgx_p2p_prepare first");
  *slab = S.p2p_slab; *flags = S.p2p_flags;
  return 0;
}

// mgx_p2p_connect for ranks whose buffers are directly addressable (other instances of this process; memory the caller mapped
// itself): slabs[r], flags[r] = what rank r's mgx_p2p_local_pointers returned.  Nothing is opened and nothing is closed later.
int mgx_p2p_connect_pointers(void *const *slabs, void *const *flags, int nranks) {
  NEED_INIT();
  if (!S.p2p_slab) return fail("mgx_p2p_connect_pointers: call mgx_p2p_prepare first");
  if (nranks != S.nranks) return fail("mgx_p2p_connect_pointers: %d pointer pairs for %d ranks", nranks, S.nranks);
  if ((int)S.lev.size() * 16 > 1024 || (int)S.lev.size() * 8 > 3072) return fail("mgx_p2p_connect_pointers: too many levels");
  S.peer_slab.assign(nranks, nullptr); S.peer_flags.assign(nranks, nullptr);
  for (int r = 0; r < nranks; r++) { S.peer_slab[r] = (double *)slabs[r]; S.peer_flags[r] = (unsigned long long *)flags[r]; }
  S.peer_slab[S.rank] = S.p2p_slab; S.peer_flags[S.rank] = S.p2p_flags;
  for (auto &L : S.lev) {
    for (int d = 0; d < 8; d++) if (L.neighb[d] >= 0 && !S.peer_slab[L.neighb[d]]) return fail("mgx_p2p_connect_pointers: no buffers for neighbour rank %d", L.neighb[d]);
    if (L.gather) for (int q = 0; q < L.ngroup; q++) if (!S.peer_slab[L.group[q]]) return fail("mgx_p2p_connect_pointers: no buffers for group member %d", L.group[q]);
  }
  S.p2p_borrowed = true;
  S.p2p_ready = true; S.p2p_on = true;
  return 0;
}

int mgx_p2p_connect(const void *all_handles, int nranks) {
  NEED_INIT();
  if (!S.p2p_slab) return fail("mgx_p2p_connect: call mgx_p2p_prepare first");
  if (nranks != S.nranks) return fail("mgx_p2p_connect: %d handle sets for %d ranks", nranks, S.nranks);
  if ((int)S.lev.size() * 16 > 1024 || (int)S.lev.size() * 8 > 3072) return fail("mgx_p2p_connect: too many levels");
  // test hook: this rank behaves as if hipIpcOpenMemHandle had refused (a rank that fails alone while its neighbours connect)
  if (getenv("MGX_P2P_TEST_FAIL_CONNECT") && atoi(getenv("MGX_P2P_TEST_FAIL_CONNECT")) == S.rank) return fail("mgx_p2p_connect: refused on rank %d (test hook MGX_P2P_TEST_FAIL_CONNECT)", S.rank);
  S.peer_slab.assign(nranks, nullptr); S.peer_flags.assign(nranks, nullptr);
  S.peer_slab[S.rank] = S.p2p_slab; S.peer_flags[S.rank] = S.p2p_flags;
  std::vector<char> need(nranks, 0);  // only the ranks that are a neighbour on some level are opened
  for (auto &L : S.lev) {
    for (int d = 0; d < 8; d++) if (L.neighb[d] >= 0) need[L.neighb[d]] = 1;
    if (L.gather) for (int q = 0; q < L.ngroup; q++) need[L.group[q]] = 1;
  }
  const hipIpcMemHandle_t *h = (const hipIpcMemHandle_t *)all_handles;
  for (int r = 0; r < nranks; r++) {
    if (r == S.rank || !need[r]) continue;
    void *p = nullptr, *f = nullptr;
    if (hipIpcOpenMemHandle(&p, h[2 * r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return fail("hipIpcOpenMemHandle(slab of rank %d) failed", r); }
    S.peer_slab[r] = (double *)p;
    if (hipIpcOpenMemHandle(&f, h[2 * r + 1], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return fail("hipIpcOpenMemHandle(flags of rank %d) failed", r); }
    S.peer_flags[r] = (unsigned long long *)f;
  }
  S.p2p_ready = true; S.p2p_on = true;
  return 0;
}

}  // extern "C"

#ifdef MGX_RBSEQ_TRACE
extern "C" int mgx_debug_rbs(int lev, unsigned long long *out8) {
  auto &L = S.lev[lev - 1];
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(out8, L.rbs_flag + (L.nx / 8 + 2) * 16, 64, hipMemcpyDeviceToHost);
  (void)hipMemset(L.rbs_flag + (L.nx / 8 + 2) * 16, 0, 64);
  return 0;
}
#endif
