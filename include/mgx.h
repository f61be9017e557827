/*
 * mgx.h -- C ABI of the MI355X-native multigrid pressure solver (libmgx.so).
 *
 * Drop-in boundary for the hot path of CESR-lab/mgroms: every entry point below
 * replaces one Fortran module procedure of the reference (file:line relative to
 * the reference's src/).  The reference has no C interop; a Fortran module
 * `nhydro` with the reference's five signatures forwards to these through
 * ISO_C_BINDING (fortran/nhydro.f90).  Like the reference
 * (module-global `grid(:)`, mg_grids.f90:113-117) the library holds ONE solver
 * instance per process unless the caller asks for more (mgx_instance_*, below);
 * one process drives one GPU.
 *
 * Conventions
 *  - all arrays are host pointers to double (real(kind=8)); the library owns
 *    every device array until mgx_clean().
 *  - 2-D geometry arrays are (0:ny+1, 0:nx+1), j fastest (mg_define_matrix.f90:71-76).
 *  - 3-D solver fields are (nz, 0:ny+1, 0:nx+1), k fastest (mg_grids.f90:207-209);
 *    cA is (8, nz, 0:ny+1, 0:nx+1) (mg_grids.f90:221).
 *  - model velocities are (i,j,k)-ordered, i fastest (nhydro.f90:57-59):
 *    u(1:nx+1,0:ny+1,1:nz)  v(0:nx+1,1:ny+1,1:nz)  w(0:nx+1,0:ny+1,0:nz).
 *  - every function returns 0 on success; on failure a non-zero code, and
 *    mgx_last_error() describes it (the reference would `stop -1`).
 */
#ifndef MGX_API_H_INCLUDED
#define MGX_API_H_INCLUDED

#ifdef __cplusplus
extern "C" {
#endif

/* The 13 members of namelist /nhparam/ (mg_namelist.f90:37-50), defaults :11-35 */
typedef struct mgx_params {
  double solver_prec;     /* 1e-6 */
  int solver_maxiter;     /* 50 */
  int nsmall;             /* 8 */
  int ns_coarsest;        /* 40 */
  int ns_pre;             /* 3 */
  int ns_post;            /* 2 */
  char cmatrix[16];       /* 'real' | 'simple' */
  char relax_method[16];  /* 'Gauss-Seidel','GS' | 'Red-Black','RB' | 'Four-Color','FC' */
  char interp_type[16];   /* 'linear' | 'nearest' */
  char restrict_type[16]; /* 'avg' (never read by the reference outside the namelist) */
  int aggressive;         /* .false. ; .true. is rejected (unimplemented in the reference, mg_intergrids.f90:243) */
  int netcdf_output;      /* .false. ; ignored (debug I/O, out of scope) */
  int bmask;              /* .false. ; .true. = masked coefficients, needs rmask in mgx_matrices */
} mgx_params;

/* field ids for mgx_get_field / mgx_set_field / mgx_fill_halo */
enum { MGX_P = 0, MGX_B = 1, MGX_R = 2, MGX_CA = 3, MGX_DX = 4, MGX_DY = 5, MGX_ZETA = 6, MGX_H = 7,
       MGX_ZR = 8, MGX_ZW = 9, MGX_CW = 10, MGX_RMASK = 14 };

/* fills *p with the defaults of mg_namelist.f90:11-35 */
int mgx_params_default(mgx_params *p);
/* read_nhnamelist (mg_namelist.f90:55-127): parse &nhparam from `path` (NULL = "nh_namelist");
 * a missing file keeps *p unchanged; linear+linear is rejected (:95-98). */
int mgx_read_namelist(const char *path, mgx_params *p);

/* nhydro_init(nx,ny,nz,npxg,npyg) (nhydro.f90:18-33).  nx,ny,nz: local block; npx,npy: process grid;
 * rank: this process's rank, placed at pi=mod(rank,npx), pj=rank/npx (mg_grids.f90:593-594).
 * par == NULL: read ./nh_namelist if present, else defaults (what the reference does). */
int mgx_init(int nx, int ny, int nz, int npx, int npy, int rank, const mgx_params *par);
/* nhydro_matrices(dx,dy,zeta,h,rmask,hc,theta_b,theta_s) (nhydro.f90:36-50) -> define_matrices
 * (mg_define_matrix.f90:28-208).  rmask may be NULL (= all ones; only read when bmask). */
int mgx_matrices(const double *dx, const double *dy, const double *zeta, const double *h, const double *rmask,
                 double hc, double theta_b, double theta_s);
/* nhydro_solve(nx,ny,nz,rmask,u,v,w) (nhydro.f90:53-102): compute_rhs, solve_p, correct_uvw; u,v,w updated in place.
 * rmask = the mask of THIS call (rmaska, nhydro.f90:56,72): the same memory layout as in mgx_matrices, (0:ny+1,0:nx+1)
 * with j fastest -- what the reference's drivers allocate (mg_testseamount.f90:97) and what compute_rhs / correct_uvw
 * index as rmask(j,i) (mg_compute_rhs.f90:61,110; the explicit-shape dummy of nhydro_solve declares it (0:nx+1,0:ny+1),
 * which is the same memory for the square blocks the reference is run on; on a non-square block with a non-trivial mask the
 * reference's own indexing reads element j+(nx+2)*i of an array its drivers fill at j+(ny+2)*i -- the library keeps the
 * drivers' layout, deliberately).  As in the reference it multiplies the w
 * cross terms of compute_rhs whatever bmask says, and yields umask / vmask when bmask (mg_compute_rhs.f90:56-72,
 * mg_correct_uvw.f90:51-68).  NULL = the level-1 mask of mgx_matrices when bmask, else all ones. */
int mgx_solve(double *u, double *v, double *w, const double *rmask);
/* nhydro_solve with the model state already on the GPU: u,v,w (and rmask, when not NULL) are DEVICE pointers (same layouts); no PCIe traffic.
 * This is what a GPU-resident ocean model calls every time step. */
int mgx_solve_device(double *u_dev, double *v_dev, double *w_dev, const double *rmask);
/* nhydro_check_nondivergence (nhydro.f90:105-134): recompute the divergence into grid(1)%b */
int mgx_check_nondivergence(double *u, double *v, double *w, const double *rmask);
/* nhydro_clean (nhydro.f90:137-141) */
void mgx_clean(void);

/* mg_solvers.f90:17-101 solve_p(tol,maxite).  *nite = iterations done, *res = last ||r||/||b||,
 * hist (may be NULL, else >= maxite+1 doubles) = normalised residual after each iteration, hist[0] = initial. */
int mgx_solve_p(double tol, int maxite, int *nite, double *res, double *hist);
/* mg_solvers.f90:104-126.  Fcycle restricts grid(1)%r on its way down: call mgx_residual for level 1 first, as solve_p does
 * (:50); the r a previous cycle's coarse2fine would have left there is not kept unless option "keep_r" is set. */
int mgx_fcycle(void);
int mgx_vcycle(int lev);                  /* mg_solvers.f90:129-151 */
int mgx_vcycle2(int lev1, int lev2);      /* mg_solvers.f90:155-177 partial V-cycle down to lev2 */
int mgx_relax(int lev, int nsweeps);      /* mg_relax.f90:16-47   */
int mgx_residual(int lev, double *res);   /* mg_relax.f90:337-383: r = b - A p, halo fill of r, *res = global ||r||_2 */
int mgx_fine2coarse(int lev);             /* mg_intergrids.f90:16-72  */
int mgx_coarse2fine(int lev);             /* mg_intergrids.f90:167-228 */
int mgx_fill_halo(int lev, int field);    /* generic fill_halo (mg_mpi_exchange.f90:10-16): p,b,r (:396-745); dx,dy,zeta,h (2D :23-352); zr,zw (nh=2 :750-1242); cA (4D :1247-1534); collective */
/* testgalerkin(lev) (mg_solvers.f90:203-288): *norm_c = <p,A p> on level lev, *norm_f = <I p, A I p> on level lev-1 (the reference
 * prints norm_c, norm_f/4 and norm_c/norm_f*4).  The coarse field is grid(lev)%p as the caller set it (the reference draws
 * it with random_number); b of both levels is zeroed. */
int mgx_testgalerkin(int lev, double *norm_c, double *norm_f);
/* compute_rhs / correct_uvw on the device-resident model state (mg_compute_rhs.f90:14, mg_correct_uvw.f90:15) */
int mgx_compute_rhs(const double *u, const double *v, const double *w, const double *rmask);

/* grid(lev)%<field> accessors (mg_grids.f90:24-65), host layout as described above */
int mgx_nlevs(void);
int mgx_level_dims(int lev, int *nx, int *ny, int *nz);
/* sequential-order red-black, windowed walk (option "rbseq_window"): the level's contraction bound rho = max |ag5| + |ag8| (-1: not a red-black
 * solver with cmatrix='real') and the planes of warm-up chosen from it (0 = the walk over the whole level stays).  No Fortran counterpart. */
int mgx_rbseq_window_info(int lev, double *rho, int *planes);
/* ... and the rows (from the bottom) its correction reaches: above the last row where max over the columns of |g(k) / g(1)| exceeds 2^-64 nothing is
 * read or written (option "rbseq_rowcut"); nz = every row. */
int mgx_rbseq_window_rows(int lev, int *rows);
/* out[0..9] = npx,npy,incx,incy,gather,ngx,ngy,key,color,0 ; out[10..17] = neighbours S,E,N,W,SW,SE,NE,NW (-1 = none) */
int mgx_level_info(int lev, int *out);
/* Pure host logic of find_grid_levels / define_grid_dims / define_neighbours / define_gather_informations
 * (mg_grids.f90:468-738) for any rank, usable before mgx_init and without a GPU.  out: 20 ints per level =
 * nx,ny,nz,npx,npy,incx,incy,gather,ngx,ngy,key,color, neighbours S,E,N,W,SW,SE,NE,NW.  Returns nlevs, or -1. */
int mgx_level_table(int nx, int ny, int nz, int npx, int npy, int rank, int nsmall, int maxlev, int *out);
/* In a multi-rank run mgx_get_field(lev, MGX_R | MGX_B, ...) is COLLECTIVE when the neighbour part of that field's halo is still
 * pending (the cycle defers the exchanges nothing reads): every rank must ask, as with mgx_fill_halo. */
int mgx_get_field(int lev, int field, double *host);
int mgx_set_field(int lev, int field, const double *host);

/* ---- multi-rank plumbing (replaces MPI in mg_mpi_exchange.f90 / mg_gather.f90) ----
 * The data path stays on the GPU: the library packs edges into device buffers and asks the host layer
 * (torch.distributed over RCCL) to move them.  All pointers handed to the callbacks are DEVICE pointers.
 *  exchange : for q in 0..n-1 send sendbuf[q] (count[q] doubles) to peer[q] and receive recvbuf[q]
 *             (count[q] doubles) from the same peer                           (fill_halo_*, :504-718)
 *  allreduce: in-place sum of n doubles over all ranks                          (global_sum, :1555-1571)
 *  allgather: gather `count` doubles from each of the `ng` ranks in `group` (ordered as the reference's
 *             localcomm, mg_grids.f90:702-718) into recvbuf                     (gather_3D, mg_gather.f90:126)
 * Each returns 0 on success.  Work must be enqueued on / ordered with the stream given to mgx_set_stream. */
typedef int (*mgx_exchange_fn)(void *ctx, int n, const int *peer, double *const *sendbuf, double *const *recvbuf,
                               const int *count);
typedef int (*mgx_allreduce_fn)(void *ctx, double *devbuf, int n);
typedef int (*mgx_allgather_fn)(void *ctx, const int *group, int ng, const double *sendbuf, double *recvbuf, int count);
int mgx_set_comm(mgx_exchange_fn ex, mgx_allreduce_fn ar, mgx_allgather_fn ag, void *ctx);

/* ---- native RCCL transport: the same three operations served inside libmgx.so by RCCL calls on the solver's stream
 * (ncclGroupStart / ncclRecv / ncclSend / ncclGroupEnd per halo fill, a one-double ncclAllReduce, grouped send/recv inside
 * the gather groups), replacing the MPI calls of mg_mpi_exchange.f90:504-718,1555-1571 and mg_gather.f90:126 with no host
 * language in the loop.  librccl is bound at run time (the copy already in the process, else librccl.so.1).
 * Bootstrap (collective, one rank per GPU, device selected by the caller): rank 0 obtains the id, the caller broadcasts its
 * mgx_rccl_unique_id_bytes() bytes (MPI_Bcast, torch.distributed, ...), every rank connects; connecting installs the
 * hooks (as mgx_set_comm would).  mgx_rccl_selftest (after mgx_init, collective): rank-coded exchange + all-reduce. */
int mgx_rccl_unique_id_bytes(void);
int mgx_rccl_get_unique_id(void *id_out);
int mgx_rccl_connect(const void *id, int nranks, int rank);
int mgx_rccl_disconnect(void);
int mgx_rccl_selftest(void);
/* human-readable name of the transport that carries the halos right now */
const char *mgx_transport(void);

/* HIP stream (hipStream_t) every kernel is launched on; NULL = the default stream */
int mgx_set_stream(void *hip_stream);
/* 0 = silent, 1 = the reference's rank-0 prints (parameter block, level table, "ite = ..: res = .. / conv = ..") */
int mgx_set_verbose(int level);

/* Options (0/1): "warm_start" keep p between solves instead of the cold start of mg_solvers.f90:35;
 * "tictoc" per-(level,name) timers like mg_tictoc.f90 (HIP events); "exact_halos" exchange the never-read r/b halos
 * eagerly as the reference does; "keep_r" (default 0) the cycles' coarse2fine also leaves the interpolated correction in the
 * fine level's r as mg_intergrids.f90:218-226 does (dead state: compute_residual rewrites r before anything reads it; the
 * mgx_coarse2fine operator always stores it, and so do the cycles under "exact_halos"); "verbose"; "p2p" (see below); "rb_chain" (default 1) red-black with cmatrix='real' on a
 * single-rank level: the colour passes write the next sweep's k=1 snapshot themselves, 0 = one snapshot launch per pass;
 * "c2f_skip" (default 1; MGX_C2F_NOSKIP=1 = 0): inside a cycle the prolongation before a four-colour relax does not update the
 * (i odd, j odd) columns, which that relax's first colour overwrites without reading them (mg_relax.f90:212-216) -- the coupling is
 * asserted by tests/test_gpu_parity.py::test_c2f_skip_is_invisible; 0 updates every column as the reference does;
 * Red-black with cmatrix='real' (the reference default) depends on the reference's SEQUENTIAL loop order (mg_relax.f90:170-186: a column
 * reads the same-colour k=1 diagonals of plane i-1 already updated, :271-276).  Three modes:
 * "rb_seq" (default 1; environment MGX_RB_SEQ): that order at speed -- parallel colour pass, a walk over the planes for the k=1 couplings,
 *   a rank-one correction per column (mgx_rbseq.hip): the reference's iterates up to the association of one sum per column (tests: 1e-12
 *   per relax call, histories within 1e-10), including the decomposition dependence on several ranks;
 * "rb_exact" (default 0; MGX_RB_EXACT): the same order bit for bit, one launch per i-plane (a parity mode, ~90 times slower);
 * both 0: the plain parallel sweep, same-colour diagonals as they were before the pass (the reference's own results differ by 2.5e-6
 *   between decompositions for the same reason; 5e-5 on the history).
 * "fuse_closing" (default 1; MGX_NO_FUSE_CLOSING): inside solve_p the closing compute_residual(1) of an iteration also restricts its r
 *   for the next Fcycle in the same pass (into grid(2)%r; grid(1)%r is materialised once, after the loop); 0 = the two operators.
 * "restrict_chain" (default 1; MGX_NO_RESTRICT_CHAIN): Fcycle's first-leg restrictions below level 1 on closed levels as one launch
 *   (mgx_kernels.hip: k_restrict_chain), 0 = one launch per level; the same bits.
 * "rbseq_fuse" (default 1; MGX_NO_RBSEQ_FUSE): with "rb_seq", walk and per-column correction of a colour in ONE launch -- on large levels
 *   the correction chases the walk across the XCDs (mgx_rbseq.hip: k_rbseq_scan, FUSE), on small levels (half-rows of at most 64 columns,
 *   at most 128 planes) every workgroup redoes the walk for its own planes (k_rbseq_walk_apply); 0 = the correction in a launch of its
 *   own behind the walk; the same bits.  A lost hand-off inside the large-level launch (its waits are bounded) makes the next
 *   synchronising call fail and switches the option off (read it back).
 * "rbseq_window" (default 1; MGX_NO_RBSEQ_WINDOW): with "rb_seq", walk and correction of a colour in one launch WITHOUT a walk over the whole
 *   level (mgx_rbseq.hip: k_rbseq_window).  The walk's recurrence contracts by rho = max |ag5| + |ag8| per plane (a property of the matrix, found
 *   when the coefficients are built: mgx_rbseq_window_info); a walk started from zero m planes before a plane has forgotten its start to
 *   rho^m, and m is chosen so that rho^m <= 2^-64 (half an ulp of the largest increment).  Every workgroup walks the m planes in front of
 *   its own over its chunk of columns +- 32; nothing is handed from one workgroup to another.  Used on every level where m <= 48 (rho <~ 0.39:
 *   0.03-0.04 on the seamount problem); otherwise, and with 0, the walk over the whole level ("rbseq_fuse").  Not the same bits as that
 *   walk (a truncation of 2^-64 of the largest increment), inside the same tolerances (tests: 1e-12 per relax call against "rb_exact").
 *   Read-only: "rbseq_window_colours" (colours done that way since mgx_init).
 * "rbseq_rowcut" (default 1): the correction p = y + g s of the windowed walk stops at the last row it reaches -- g = T^-1 e1 decays away from the bottom
 *   row, and above the last row where max over the columns of |g(k) / g(1)| exceeds 2^-64 (found with the coefficients: mgx_rbseq_window_rows) the
 *   correction is below 2^-63 of the largest increment: those rows are neither read nor written (55 of 64 rows on level 1 of 512x512x64, 56 of 128 at
 *   nz = 128); 0 = every row (A/B).
 * "coarsest_direct" (default 1; MGX_COARSEST_DIRECT=n): inside a cycle the coarsest level is entered with p = 0 and left after relax(nlevs, ns_coarsest)
 *   (mg_solvers.f90:117,144): a fixed linear map of b.  Its matrix is built with the level's own relax kernel from the unit vectors whenever the
 *   coefficients change (lazily, at the first cycle after) and the solve becomes one matrix-vector product (mgx_relax_coarse.hip: k_coarse_direct; closed,
 *   un-gathered coarsest levels of at most 2048 cells).  The same map in another association -- not the same bits as the sweeps (1e-15 of max|p|) -- so:
 *   1 = only where the iteration is tolerance-based anyway (relax_method='RB', cmatrix='real' in the sequential order at speed, "rb_seq"), 2 = every method
 *   (four colours then lose their bit parity with the reference's loop), 0 = never.  relax(nlevs, n) called as an operator always sweeps.
 *   Read-only: "coarsest_direct_solves".
 * "rbseq_fuse_min" (default 4194304): cells of a colour (nx * ny/2 * nz) from which on a level counts as large for "rbseq_fuse".
 * "rbseq_timeout_ms" (default 2000; write-only): bound of the waits inside that launch.
 * "rbseq_d0_in_pass" (default 1): the colour pass also writes the walk's d0 (0 = a launch of its own; the same bits).
 * "overlap" (default 0; MGX_OVERLAP=1): four colours on a level with neighbours, halos by the pushes: the boundary part of a colour pass and
 *   the exchange on a second stream beside the interior part.  Same bits; slower where it could be measured.
 * "ksp" (default 1): the persistent relax of the closed mid levels; read back 0 after it timed out (then off until mgx_init or "ksp" = 1).
 * Read-only through mgx_get_option: "p2p_failed" (a peer-to-peer wait of THIS rank timed out since the ranks last agreed: see below),
 *   "overlapped_passes". */
int mgx_set_option(const char *name, int value);
/* Option "async" (default 0): the cycle / operator entry points (mgx_vcycle, mgx_vcycle2, mgx_fcycle, mgx_relax, mgx_fine2coarse,
 * mgx_coarse2fine) only ENQUEUE their kernels on the solver's stream and return -- what a GPU-resident model wants between its own
 * kernels.  mgx_synchronize() waits for the stream and reports what the device flagged meanwhile (a peer that never showed up, a
 * rejected launch); every entry point that returns data to the host (mgx_residual, mgx_solve_p, mgx_get_field ...) synchronises anyway. */
int mgx_synchronize(void);
/* read back an option, or an integer / logical member of /nhparam/ as mgx_init took it from nh_namelist ("bmask", "nsmall",
 * "solver_maxiter", "ns_coarsest", "ns_pre", "ns_post", "netcdf_output", "aggressive"): the reference's drivers read these module
 * variables of mg_namelist directly (e.g. `if (bmask)`, mg_testseamount.f90) */
int mgx_get_option(const char *name, int *value);
/* print_tictoc (mg_tictoc.f90:114-153): timer table (seconds, calls per level) to `path` (NULL = "fort.10") */
int mgx_print_tictoc(const char *path);
/* tic(lev, name) / toc(lev, name) (mg_tictoc.f90:21-111) for the CALLER's own sections -- the reference's drivers bracket their main
 * program with them (mg_testseamount.f90:37,220).  Host wall clock like the reference's system_clock; toc waits for the solver's stream
 * first, so that the section contains the GPU work enqueued inside it.  They share the table of mgx_print_tictoc with the library's own
 * timers (option "tictoc"); the table survives mgx_clean, as the reference's module variables survive nhydro_clean. */
int mgx_tic(int lev, const char *name);
int mgx_toc(int lev, const char *name);

/* ---- measurement helpers used by bench.py (timed with HIP events on the solver's stream) ---- */
/* run `reps` smoother sweeps on level `lev` (reps x relax(lev,1)); *ms = average milliseconds per sweep */
int mgx_time_relax(int lev, int reps, float *ms);
/* self-test of the arithmetic identity the level-1 colour pass relies on for its per-column divisors: a[i] / b[i] by
 * the hardware fp64 division sequence against the refined-reciprocal quotient; *nbad = pairs whose bits differ (0 for operands whose
 * quotient and operands sit in the normal range).  Host arrays; needs no mgx_init. */
int mgx_selftest_divc(const double *a, const double *b, int n, long long *nbad);
int mgx_time_residual(int lev, int reps, float *ms);
/* counters since mgx_init: out[0]=kernel launches, out[1]=halo fills, out[2]=exchanges, out[3]=allreduces */
int mgx_counters(long long *out);

/* ---- peer-to-peer halo transport (replaces the MPI_Isend/MPI_Irecv/MPI_Waitall of fill_halo_3D[_relax],
 * mg_mpi_exchange.f90:504-718, for the p/b/r halos of the cycle) ----
 * Ranks of one node write their edges straight into the neighbours' receive buffers over xGMI (device memory shared
 * through hipIpc) and raise a flag there; the receiver waits on its local flag and unpacks -- one kernel per halo fill,
 * no host step, no callback.  Set-up is collective: every rank calls mgx_p2p_prepare (after mgx_init), the handle blobs
 * (mgx_p2p_handle_bytes() each) are all-gathered in rank order by the caller, every rank calls mgx_p2p_connect with
 * the concatenation.  mgx_set_option("p2p", 0|1) switches between this transport and the exchange callback (all
 * ranks together).  A neighbour that never shows up ends the wait after 5 s (option "p2p_timeout_ms" / MGX_P2P_TIMEOUT_MS); the rank
 * that waited does NOT fall back alone: it keeps exchanging and the ranks agree at the next global_sum (every solve_p iteration, every
 * mgx_residual norm), whose all-reduce carries a second value "a wait of mine timed out": if any rank says so, EVERY rank switches
 * to the hooks, rewinds its sequence numbers and returns the same error (the solve in progress is void; repeat it).  Test hook:
 * option "p2p_test_drop" = n makes the n-th halo exchange of this rank keep its flags down; "p2p_failed" (mgx_get_option) reads the
 * local marker. */
int mgx_p2p_handle_bytes(void);
int mgx_p2p_prepare(void *handles_out);
int mgx_p2p_connect(const void *all_handles, int nranks);
long long mgx_p2p_exchanges(void);

/* ---- more than one solver in a process (the reference has exactly one: module-global grid(:), mg_grids.f90:113-117) ----
 * Instance 0 exists from the start and is what every thread acts on.  mgx_instance_create returns the id of a new, empty
 * instance; mgx_instance_select(id) makes all later mgx_* calls OF THE CALLING THREAD act on it (thread-local selection:
 * one thread per instance may run concurrently; two threads must not drive the same instance at the same time).  Uses:
 * several nested domains coupled from one process; several ranks of one job as threads of one process (each with its own
 * stream, mgx_set_stream) where a box admits fewer processes than ranks -- they are connected with mgx_set_comm hooks or with
 * mgx_p2p_prepare + mgx_p2p_local_pointers + mgx_p2p_connect_pointers (same-process buffers need no hipIpc).
 * mgx_instance_destroy frees an instance (never 0); mgx_instance_current returns the calling thread's id. */
int mgx_instance_create(void);
int mgx_instance_select(int id);
int mgx_instance_current(void);
int mgx_instance_destroy(int id);
int mgx_p2p_local_pointers(void **slab, void **flags);
int mgx_p2p_connect_pointers(void *const *slabs, void *const *flags, int nranks);

const char *mgx_last_error(void);
const char *mgx_version(void);

#ifdef __cplusplus
}
#endif
#endif
