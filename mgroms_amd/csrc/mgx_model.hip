// Model coupling: compute_rhs (mg_compute_rhs.f90:14-379) and correct_uvw (mg_correct_uvw.f90:15-115) on the
// model's (i,j,k)-ordered velocities, which stay where the model keeps them.  Everything here works in "model space":
// one lane = one (i,j) column with lanes running along i, the fastest index of u, v, w, and every array the formulas
// read has an i-fastest copy made at set-up (zw, dzw, zxdy, zydx, cw of level 1; dx, dy, rmask), so each load and store
// of a wave is one contiguous run.  The result b goes to the solver's JS layout through an LDS-tiled transpose, and the
// pressure comes back the same way for correct_uvw.  Operation order follows the reference line by line
// (compiled with -ffp-contract=off): results are bit-identical to the CPU oracle.
#include "mgx_internal.h"

struct ModelView { double *u, *v, *w, *rmask; int bmask; };  // rmask: i-fastest copy of the level-1 mask (only read when bmask)
#define U(i, j, k) M.u[(((long long)((k)-1)) * (ny + 2) + (j)) * (nx + 1) + ((i)-1)]
#define V(i, j, k) M.v[(((long long)((k)-1)) * (ny + 1) + ((j)-1)) * (nx + 2) + (i)]
#define Wv(i, j, k) M.w[(((long long)(k)) * (ny + 2) + (j)) * (nx + 2) + (i)]
// model-layout index of a level-1 array with rows k = 1.., j,i = 0..n+1 (i fastest)
#define MI3(k, j, i) ((((long long)((k)-1)) * (ny + 2) + (j)) * (nx + 2) + (i))
#define M2(a, j, i) a[((long long)(j)) * (nx + 2) + (i)]
#define MZW(k, j, i) G.mzw[MI3(k, j, i)]
#define MDZW(k, j, i) G.mdzw[MI3(k, j, i)]
#define MZXDY(k, j, i) G.mzxdy[MI3(k, j, i)]
#define MZYDX(k, j, i) G.mzydx[MI3(k, j, i)]
#define MCW(k, j, i) G.mcw[MI3(k, j, i)]
#define MDX(j, i) M2(G.mdx, j, i)
#define MDY(j, i) M2(G.mdy, j, i)
#define RM(j, i) (M.rmask ? M2(M.rmask, j, i) : 1.0)
#define UMK(j, i) (M.bmask ? (((i) >= 1) ? RM(j, (i)-1) * RM(j, i) : 0.0) : 1.0)
#define VMK(j, i) (M.bmask ? (((j) >= 1) ? RM((j)-1, i) * RM(j, i) : 0.0) : 1.0)

#define COLUMN_THREAD_I(jlo, jhi, ilo, ihi)                                   \
  const int i = (ilo) + blockIdx.x * blockDim.x + threadIdx.x;                \
  const int j = (jlo) + blockIdx.y * blockDim.y + threadIdx.y;                \
  if (j > (jhi) || i > (ihi)) return;                                         \
  const int nx = G.nx, ny = G.ny, nz = G.nz;                                  \
  (void)nx; (void)ny; (void)nz;

// ---- layout changes (LDS-tiled transposes, 32x32 tiles, block 32x8) ---------------------------------------------
// reference layout a(rows, 1-nh:ny+nh, 1-nh:nx+nh) (k fastest) -> model layout (rows, 0:ny+1, 0:nx+1) (i fastest)
__global__ __launch_bounds__(256) void k_ref2model(const double *__restrict__ src, double *__restrict__ dst, int rows, int nh, int nx, int ny) {
  __shared__ double t[32][33];
  const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32, j = blockIdx.z;
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int i = i0 + r, k = k0 + threadIdx.x;
    if (i <= nx + 1 && k < rows) t[r][threadIdx.x] = src[(((long long)(i + nh - 1)) * (ny + 2 * nh) + (j + nh - 1)) * rows + k];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int k = k0 + r, i = i0 + threadIdx.x;
    if (i <= nx + 1 && k < rows) dst[((long long)k * (ny + 2) + j) * (nx + 2) + i] = t[threadIdx.x][r];
  }
}
// 2-D (0:ny+1,0:nx+1) j fastest -> i fastest
__global__ void k_ref2model_2d(const double *__restrict__ src, double *__restrict__ dst, int nx, int ny) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i <= nx + 1) dst[(long long)j * (nx + 2) + i] = src[(long long)i * (ny + 2) + j];
}
// JS field (rows of L) <-> model layout.  dir 0: JS -> model over j,i = 0..n+1 ; dir 1: model -> JS over the interior
__global__ __launch_bounds__(256) void k_js_model(LevView L, double *__restrict__ js, double *__restrict__ md, int dir) {
  __shared__ double t[32][33];
  const int nx = L.nx, ny = L.ny;
  const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32, k = blockIdx.z;
  const int lo = dir ? 1 : 0, ihi = dir ? nx : nx + 1, jhi = dir ? ny : ny + 1;
  if (dir == 0) {
    for (int r = threadIdx.y; r < 32; r += 8) {
      const int i = i0 + r, j = j0 + threadIdx.x;
      if (i <= ihi && j <= jhi) t[r][threadIdx.x] = js[(long long)i * L.plane + (long long)k * L.RS + jpos(L, j)];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
      const int j = j0 + r, i = i0 + threadIdx.x;
      if (i <= ihi && j <= jhi) md[((long long)k * (ny + 2) + j) * (nx + 2) + i] = t[threadIdx.x][r];
    }
  } else {
    for (int r = threadIdx.y; r < 32; r += 8) {
      const int j = j0 + r, i = i0 + threadIdx.x;
      if (i >= lo && j >= lo && i <= ihi && j <= jhi) t[r][threadIdx.x] = md[((long long)k * (ny + 2) + j) * (nx + 2) + i];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
      const int i = i0 + r, j = j0 + threadIdx.x;
      if (i >= lo && j >= lo && i <= ihi && j <= jhi) js[(long long)i * L.plane + (long long)k * L.RS + jpos(L, j)] = t[threadIdx.x][r];
    }
  }
}

// ---- compute_rhs: horizontal and vertical fluxes (mg_compute_rhs.f90:76-168, :177-269, :278-357) -----------------
__global__ void k_rhs_uf(GeoView G, ModelView M, double *__restrict__ fx) {
  COLUMN_THREAD_I(1, G.ny, 1, G.nx + 1)
  const double two = 2.0, hlf = 0.5, qrt = 0.25;
  int k = 1;
  fx[MI3(k, j, i)] =
      (qrt * (MZW(k + 1, j, i) - MZW(k, j, i) + MZW(k + 1, j, i - 1) - MZW(k, j, i - 1)) * (MDY(j, i) + MDY(j, i - 1)) * U(i, j, k)
       - qrt * (+MZXDY(k, j, i) * MDZW(k + 1, j, i) * Wv(i, j, k + 1 - 1) * RM(j, i) +
                MZXDY(k, j, i - 1) * MDZW(k + 1, j, i - 1) * Wv(i - 1, j, k + 1 - 1) * RM(j, i - 1))
       - (+MZXDY(k, j, i) * MZXDY(k, j, i) / (MCW(k, j, i) + MCW(k + 1, j, i)) +
          MZXDY(k, j, i - 1) * MZXDY(k, j, i - 1) / (MCW(k, j, i - 1) + MCW(k + 1, j, i - 1))) *
             (hlf * (MDX(j, i) + MDX(j, i - 1))) * U(i, j, k)
       - (+MZXDY(k, j, i) * MZYDX(k, j, i) / (MCW(k, j, i) + MCW(k + 1, j, i)) * hlf *
              (hlf * (MDY(j, i) + MDY(j - 1, i)) * V(i, j, k) * VMK(j, i) + hlf * (MDY(j + 1, i) + MDY(j, i)) * V(i, j + 1, k) * VMK(j + 1, i)) +
          MZXDY(k, j, i - 1) * MZYDX(k, j, i - 1) / (MCW(k, j, i - 1) + MCW(k + 1, j, i - 1)) * hlf *
              (hlf * (MDY(j, i - 1) + MDY(j - 1, i - 1)) * V(i - 1, j, k) * VMK(j, i - 1) +
               hlf * (MDY(j + 1, i - 1) + MDY(j, i - 1)) * V(i - 1, j + 1, k) * VMK(j + 1, i - 1)))) * UMK(j, i);
  for (k = 2; k <= nz - 1; k++)
    fx[MI3(k, j, i)] =
        (qrt * (MZW(k + 1, j, i) - MZW(k, j, i) + MZW(k + 1, j, i - 1) - MZW(k, j, i - 1)) * (MDY(j, i) + MDY(j, i - 1)) * U(i, j, k)
         - qrt * (+MZXDY(k, j, i) * MDZW(k, j, i) * Wv(i, j, k - 1) * RM(j, i) +
                  MZXDY(k, j, i) * MDZW(k + 1, j, i) * Wv(i, j, k + 1 - 1) * RM(j, i) +
                  MZXDY(k, j, i - 1) * MDZW(k, j, i - 1) * Wv(i - 1, j, k - 1) * RM(j, i - 1) +
                  MZXDY(k, j, i - 1) * MDZW(k + 1, j, i - 1) * Wv(i - 1, j, k + 1 - 1) * RM(j, i - 1))) * UMK(j, i);
  k = nz;
  fx[MI3(k, j, i)] =
      (qrt * (MZW(k + 1, j, i) - MZW(k, j, i) + MZW(k + 1, j, i - 1) - MZW(k, j, i - 1)) * (MDY(j, i) + MDY(j, i - 1)) * U(i, j, k)
       - qrt * (+MZXDY(k, j, i) * MDZW(k, j, i) * Wv(i, j, k - 1) * RM(j, i) +
                MZXDY(k, j, i) * two * MDZW(k + 1, j, i) * Wv(i, j, k + 1 - 1) * RM(j, i) +
                MZXDY(k, j, i - 1) * MDZW(k, j, i - 1) * Wv(i - 1, j, k - 1) * RM(j, i - 1) +
                MZXDY(k, j, i - 1) * two * MDZW(k + 1, j, i - 1) * Wv(i - 1, j, k + 1 - 1) * RM(j, i - 1))) * UMK(j, i);
}

__global__ void k_rhs_vf(GeoView G, ModelView M, double *__restrict__ fx) {
  COLUMN_THREAD_I(1, G.ny + 1, 1, G.nx)
  const double two = 2.0, hlf = 0.5, qrt = 0.25;
  int k = 1;
  fx[MI3(k, j, i)] =
      (qrt * (MZW(k + 1, j, i) - MZW(k, j, i) + MZW(k + 1, j - 1, i) - MZW(k, j - 1, i)) * (MDX(j, i) + MDX(j - 1, i)) * V(i, j, k)
       - qrt * (+MZYDX(k, j, i) * MDZW(k + 1, j, i) * Wv(i, j, k + 1 - 1) * RM(j, i) +
                MZYDX(k, j - 1, i) * MDZW(k + 1, j - 1, i) * Wv(i, j - 1, k + 1 - 1) * RM(j - 1, i))
       - (+MZYDX(k, j, i) * MZYDX(k, j, i) / (MCW(k, j, i) + MCW(k + 1, j, i)) +
          MZYDX(k, j - 1, i) * MZYDX(k, j - 1, i) / (MCW(k, j - 1, i) + MCW(k + 1, j - 1, i))) *
             hlf * (MDY(j, i) + MDY(j - 1, i)) * V(i, j, k)
       - (+MZXDY(k, j, i) * MZYDX(k, j, i) / (MCW(k, j, i) + MCW(k + 1, j, i)) * hlf *
              (hlf * (MDX(j, i) + MDX(j, i - 1)) * U(i, j, k) * UMK(j, i) + hlf * (MDX(j, i + 1) + MDX(j, i)) * U(i + 1, j, k) * UMK(j, i + 1)) +
          MZXDY(k, j - 1, i) * MZYDX(k, j - 1, i) / (MCW(k, j - 1, i) + MCW(k + 1, j - 1, i)) * hlf *
              (hlf * (MDX(j - 1, i) + MDX(j - 1, i - 1)) * U(i, j - 1, k) * UMK(j - 1, i) +
               hlf * (MDX(j - 1, i + 1) + MDX(j - 1, i)) * U(i + 1, j - 1, k) * UMK(j - 1, i + 1)))) * VMK(j, i);
  for (k = 2; k <= nz - 1; k++)
    fx[MI3(k, j, i)] =
        (qrt * (MZW(k + 1, j, i) - MZW(k, j, i) + MZW(k + 1, j - 1, i) - MZW(k, j - 1, i)) * (MDX(j, i) + MDX(j - 1, i)) * V(i, j, k)
         - qrt * (+MZYDX(k, j, i) * MDZW(k, j, i) * Wv(i, j, k - 1) * RM(j, i) +
                  MZYDX(k, j, i) * MDZW(k + 1, j, i) * Wv(i, j, k + 1 - 1) * RM(j, i) +
                  MZYDX(k, j - 1, i) * MDZW(k, j - 1, i) * Wv(i, j - 1, k - 1) * RM(j - 1, i) +
                  MZYDX(k, j - 1, i) * MDZW(k + 1, j - 1, i) * Wv(i, j - 1, k + 1 - 1) * RM(j - 1, i))) * VMK(j, i);
  k = nz;
  fx[MI3(k, j, i)] =
      (qrt * (MZW(k + 1, j, i) - MZW(k, j, i) + MZW(k + 1, j - 1, i) - MZW(k, j - 1, i)) * (MDX(j, i) + MDX(j - 1, i)) * V(i, j, k)
       - qrt * (+MZYDX(k, j, i) * MDZW(k, j, i) * Wv(i, j, k - 1) * RM(j, i) +
                MZYDX(k, j, i) * two * MDZW(k + 1, j, i) * Wv(i, j, k + 1 - 1) * RM(j, i) +
                MZYDX(k, j - 1, i) * MDZW(k, j - 1, i) * Wv(i, j - 1, k - 1) * RM(j - 1, i) +
                MZYDX(k, j - 1, i) * two * MDZW(k + 1, j - 1, i) * Wv(i, j - 1, k + 1 - 1) * RM(j - 1, i))) * VMK(j, i);
}

__global__ void k_rhs_wf(GeoView G, ModelView M, double *__restrict__ fz) {
  COLUMN_THREAD_I(1, G.ny, 1, G.nx)
  const double hlf = 0.5, qrt = 0.25;
  fz[MI3(1, j, i)] = 0.0;
  for (int k = 2; k <= nz; k++) {
    double t = MCW(k, j, i) * MDZW(k, j, i) * Wv(i, j, k - 1) -
               qrt * hlf * (+MZXDY(k, j, i) * (MDX(j, i) + MDX(j, i - 1)) * U(i, j, k) * UMK(j, i) +
                            MZXDY(k, j, i) * (MDX(j, i + 1) + MDX(j, i)) * U(i + 1, j, k) * UMK(j, i + 1) +
                            MZXDY(k - 1, j, i) * (MDX(j, i) + MDX(j, i - 1)) * U(i, j, k - 1) * UMK(j, i) +
                            MZXDY(k - 1, j, i) * (MDX(j, i + 1) + MDX(j, i)) * U(i + 1, j, k - 1) * UMK(j, i + 1));
    t = t - qrt * hlf * (+MZYDX(k, j, i) * (MDY(j, i) + MDY(j - 1, i)) * V(i, j, k) * VMK(j, i) +
                         MZYDX(k, j, i) * (MDY(j + 1, i) + MDY(j, i)) * V(i, j + 1, k) * VMK(j + 1, i) +
                         MZYDX(k - 1, j, i) * (MDY(j, i) + MDY(j - 1, i)) * V(i, j, k - 1) * VMK(j, i) +
                         MZYDX(k - 1, j, i) * (MDY(j + 1, i) + MDY(j, i)) * V(i, j + 1, k - 1) * VMK(j + 1, i));
    fz[MI3(k, j, i)] = t;
  }
  const int k = nz + 1;
  fz[MI3(k, j, i)] = MCW(k, j, i) * MDZW(k, j, i) * Wv(i, j, k - 1) -
                     hlf * hlf * (+MZXDY(k - 1, j, i) * (MDX(j, i) + MDX(j, i - 1)) * U(i, j, k - 1) * UMK(j, i) +
                                  MZXDY(k - 1, j, i) * (MDX(j, i + 1) + MDX(j, i)) * U(i + 1, j, k - 1) * UMK(j, i + 1)) -
                     hlf * hlf * (+MZYDX(k - 1, j, i) * (MDY(j, i) + MDY(j - 1, i)) * V(i, j, k - 1) * VMK(j, i) +
                                  MZYDX(k - 1, j, i) * (MDY(j + 1, i) + MDY(j, i)) * V(i, j + 1, k - 1) * VMK(j + 1, i));
}

// lbc_null of fill_halo(1,uf,'u') / (1,vf,'v') (mg_compute_rhs.f90:171,272) restricted to what the divergence reads:
// the flux through a physical boundary face is zero.  face 0: i = ipl (all j,k) ; face 1: j = jpl (all i,k)
__global__ void k_flux_zero_face(GeoView G, double *__restrict__ f, int face, int pl) {
  const int nx = G.nx, ny = G.ny;
  const int q = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y + 1;
  if (face == 0) { if (q <= ny + 1) f[MI3(k, q, pl)] = 0.0; }
  else { if (q <= nx + 1) f[MI3(k, pl, q)] = 0.0; }
}
// neighbour exchange of the same fill: pack my first face (i=1 or j=1) for the W/S neighbour, unpack the E/N neighbour's
// into i=nx+1 / j=ny+1.  Buffer order (k, q) with q fastest.
__global__ void k_flux_face_copy(GeoView G, double *__restrict__ f, double *__restrict__ buf, int face, int pl, int unpack) {
  const int nx = G.nx, ny = G.ny;
  const int n = face == 0 ? ny : nx;
  const int q = 1 + blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y + 1;
  if (q > n) return;
  const long long e = face == 0 ? MI3(k, q, pl) : MI3(k, pl, q);
  const long long t = (long long)(k - 1) * n + (q - 1);
  if (unpack) f[e] = buf[t]; else buf[t] = f[e];
}

// divergence, accumulated in the reference's order (:173, :274, :362-370) in a model-layout scratch bm
// mode 0: bm = uf(i+1)-uf(i) ; 1: bm += vf(j+1)-vf(j) ; 2: bm += wf(k+1)-wf(k)
__global__ void k_rhs_accum_m(GeoView G, double *__restrict__ bm, const double *__restrict__ f, int mode) {
  COLUMN_THREAD_I(1, G.ny, 1, G.nx)
  for (int k = 1; k <= nz; k++) {
    const long long e = MI3(k, j, i);
    if (mode == 0) bm[e] = f[MI3(k, j, i + 1)] - f[e];
    else if (mode == 1) bm[e] = bm[e] + f[MI3(k, j + 1, i)] - f[e];
    else bm[e] = bm[e] + f[MI3(k + 1, j, i)] - f[e];
  }
}

// correct_uvw (mg_correct_uvw.f90:73-108); pm = level-1 pressure in model layout (halo included).
// dzw(k) of the reference (zr(k)-zr(k-1), top: zw(nz+1)-zr(nz)) is the same expression as grid(1)%dzw.
__global__ void k_correct_uvw_m(GeoView G, const double *__restrict__ pm, ModelView M) {
  COLUMN_THREAD_I(0, G.ny + 1, 0, G.nx + 1)
  const double one = 1.0, hlf = 0.5;
#define PM(k, jj, ii) pm[MI3(k, jj, ii)]
  if (i >= 1) {
    const double dxu = hlf * (MDX(j, i) + MDX(j, i - 1));
    for (int k = 1; k <= nz; k++) U(i, j, k) = U(i, j, k) - one / dxu * (PM(k, j, i) - PM(k, j, i - 1)) * UMK(j, i);
  }
  if (j >= 1) {
    const double dyv = hlf * (MDY(j, i) + MDY(j - 1, i));
    for (int k = 1; k <= nz; k++) V(i, j, k) = V(i, j, k) - one / dyv * (PM(k, j, i) - PM(k, j - 1, i)) * VMK(j, i);
  }
  for (int k = 2; k <= nz; k++) {
    const double dzw = MDZW(k, j, i);
    Wv(i, j, k - 1) = Wv(i, j, k - 1) - one / dzw * (PM(k, j, i) - PM(k - 1, j, i));
  }
  const int k = nz + 1;
  const double dzw = MDZW(nz + 1, j, i);
  Wv(i, j, k - 1) = Wv(i, j, k - 1) - one / dzw * (-PM(k - 1, j, i));
#undef PM
}

// ------------------------------------------------------------------------------------------------
static inline dim3 igrid(int ni, int nj) { return dim3((ni + 63) / 64, (nj + 3) / 4); }
static const dim3 IBLK(64, 4);

extern "C" {
void mgxm_ref2model(hipStream_t st, const double *src, double *dst, int rows, int nh, int nx, int ny) {
  hipLaunchKernelGGL(k_ref2model, dim3((nx + 2 + 31) / 32, (rows + 31) / 32, ny + 2), dim3(32, 8), 0, st, src, dst, rows, nh, nx, ny);
}
void mgxm_ref2model_2d(hipStream_t st, const double *src, double *dst, int nx, int ny) {
  hipLaunchKernelGGL(k_ref2model_2d, dim3((nx + 2 + 63) / 64, ny + 2), dim3(64), 0, st, src, dst, nx, ny);
}
void mgxm_js_model(hipStream_t st, const LevView *L, double *js, double *md, int dir) {
  hipLaunchKernelGGL(k_js_model, dim3((L->nx + 2 + 31) / 32, (L->ny + 2 + 31) / 32, L->nz), dim3(32, 8), 0, st, *L, js, md, dir);
}
void mgxm_rhs_uf(hipStream_t st, const GeoView *G, const ModelView *M, double *fx) { hipLaunchKernelGGL(k_rhs_uf, igrid(G->nx + 1, G->ny), IBLK, 0, st, *G, *M, fx); }
void mgxm_rhs_vf(hipStream_t st, const GeoView *G, const ModelView *M, double *fx) { hipLaunchKernelGGL(k_rhs_vf, igrid(G->nx, G->ny + 1), IBLK, 0, st, *G, *M, fx); }
void mgxm_rhs_wf(hipStream_t st, const GeoView *G, const ModelView *M, double *fz) { hipLaunchKernelGGL(k_rhs_wf, igrid(G->nx, G->ny), IBLK, 0, st, *G, *M, fz); }
void mgxm_flux_zero_face(hipStream_t st, const GeoView *G, double *f, int face, int pl) {
  const int n = (face == 0 ? G->ny : G->nx) + 2;
  hipLaunchKernelGGL(k_flux_zero_face, dim3((n + 63) / 64, G->nz), dim3(64), 0, st, *G, f, face, pl);
}
void mgxm_flux_face_copy(hipStream_t st, const GeoView *G, double *f, double *buf, int face, int pl, int unpack) {
  const int n = face == 0 ? G->ny : G->nx;
  hipLaunchKernelGGL(k_flux_face_copy, dim3((n + 63) / 64, G->nz), dim3(64), 0, st, *G, f, buf, face, pl, unpack);
}
void mgxm_rhs_accum(hipStream_t st, const GeoView *G, double *bm, const double *f, int mode) { hipLaunchKernelGGL(k_rhs_accum_m, igrid(G->nx, G->ny), IBLK, 0, st, *G, bm, f, mode); }
void mgxm_correct_uvw(hipStream_t st, const GeoView *G, const double *pm, const ModelView *M) { hipLaunchKernelGGL(k_correct_uvw_m, igrid(G->nx + 2, G->ny + 2), IBLK, 0, st, *G, pm, *M); }
}
