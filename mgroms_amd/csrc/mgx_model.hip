// Model coupling: compute_rhs (mg_compute_rhs.f90:14-379) and correct_uvw (mg_correct_uvw.f90:15-115) on the
// model's (i,j,k)-ordered velocities, which stay where the model keeps them.  Everything here works in "model space":
// one lane = one (i,j) column with lanes running along i, the fastest index of u, v, w, and every array the formulas
// read has an i-fastest copy made at set-up (zw, dzw, zxdy, zydx, cw of level 1; dx, dy, rmask), so each load and store
// of a wave is one contiguous run.  The result b goes to the solver's JS layout through an LDS-tiled transpose, and the
// pressure comes back the same way for correct_uvw.  Operation order follows the reference line by line
// (compiled with -ffp-contract=off): results are bit-identical to the CPU oracle.
#include <cstdlib>

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
// rows ka .. kb of 1 .. klast for blockIdx.z (the divergence pass: nothing in it is sequential in k)
#define ROW_RUN(klast) const int ka = 1 + blockIdx.z * KR, kb = ka + KR - 1 < (klast) ? ka + KR - 1 : (klast);

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
// A lane climbs its column and keeps what row k+1 of this step is to row k of the next (zw, dzw, w; in k_rhs_wf zxdy, zydx, u, v of row
// k-1) in registers: rows are 2 MB apart, so a value asked for again one step later comes from HBM again -- k_rhs_uf moved 1.5 GB for
// 0.8 GB of operands (264 us = 5.6 TB/s: at the memory's rate, on the wrong bytes).  Same values in the same expressions: same bits.
__global__ void k_rhs_uf(GeoView G, ModelView M, double *__restrict__ fx, int KR) {
  COLUMN_THREAD_I(1, G.ny, 1, G.nx + 1)
  ROW_RUN(nz)
  const double two = 2.0, hlf = 0.5, qrt = 0.25;
  const double dys = MDY(j, i) + MDY(j, i - 1), rmA = RM(j, i), rmB = RM(j, i - 1), um = UMK(j, i);
  double zwA, zwB, zwA1 = MZW(ka, j, i), zwB1 = MZW(ka, j, i - 1);            // zw of rows k (A: column i, B: i-1) and k+1
  double dzA, dzB, dzA1 = MDZW(ka, j, i), dzB1 = MDZW(ka, j, i - 1);          // dzw likewise
  double wA0, wB0, wA1 = ka >= 2 ? Wv(i, j, ka - 1) : 0.0, wB1 = ka >= 2 ? Wv(i - 1, j, ka - 1) : 0.0;  // w(k-1) and w(k)
  for (int k = ka; k <= kb; k++) {
    zwA = zwA1; zwB = zwB1; zwA1 = MZW(k + 1, j, i); zwB1 = MZW(k + 1, j, i - 1);
    dzA = dzA1; dzB = dzB1; dzA1 = MDZW(k + 1, j, i); dzB1 = MDZW(k + 1, j, i - 1);
    wA0 = wA1; wB0 = wB1; wA1 = Wv(i, j, k); wB1 = Wv(i - 1, j, k);
    const double zxA = MZXDY(k, j, i), zxB = MZXDY(k, j, i - 1);
    if (k == 1)
      fx[MI3(k, j, i)] =
          (qrt * (zwA1 - zwA + zwB1 - zwB) * dys * U(i, j, k)
           - qrt * (+zxA * dzA1 * wA1 * rmA +
                    zxB * dzB1 * wB1 * rmB)
           - (+zxA * zxA / (MCW(k, j, i) + MCW(k + 1, j, i)) +
              zxB * zxB / (MCW(k, j, i - 1) + MCW(k + 1, j, i - 1))) *
                 (hlf * (MDX(j, i) + MDX(j, i - 1))) * U(i, j, k)
           - (+zxA * MZYDX(k, j, i) / (MCW(k, j, i) + MCW(k + 1, j, i)) * hlf *
                  (hlf * (MDY(j, i) + MDY(j - 1, i)) * V(i, j, k) * VMK(j, i) + hlf * (MDY(j + 1, i) + MDY(j, i)) * V(i, j + 1, k) * VMK(j + 1, i)) +
              zxB * MZYDX(k, j, i - 1) / (MCW(k, j, i - 1) + MCW(k + 1, j, i - 1)) * hlf *
                  (hlf * (MDY(j, i - 1) + MDY(j - 1, i - 1)) * V(i - 1, j, k) * VMK(j, i - 1) +
                   hlf * (MDY(j + 1, i - 1) + MDY(j, i - 1)) * V(i - 1, j + 1, k) * VMK(j + 1, i - 1)))) * um;
    else if (k <= nz - 1)
      fx[MI3(k, j, i)] =
          (qrt * (zwA1 - zwA + zwB1 - zwB) * dys * U(i, j, k)
           - qrt * (+zxA * dzA * wA0 * rmA +
                    zxA * dzA1 * wA1 * rmA +
                    zxB * dzB * wB0 * rmB +
                    zxB * dzB1 * wB1 * rmB)) * um;
    else
      fx[MI3(k, j, i)] =
          (qrt * (zwA1 - zwA + zwB1 - zwB) * dys * U(i, j, k)
           - qrt * (+zxA * dzA * wA0 * rmA +
                    zxA * two * dzA1 * wA1 * rmA +
                    zxB * dzB * wB0 * rmB +
                    zxB * two * dzB1 * wB1 * rmB)) * um;
  }
}

__global__ void k_rhs_vf(GeoView G, ModelView M, double *__restrict__ fx, int KR) {
  COLUMN_THREAD_I(1, G.ny + 1, 1, G.nx)
  ROW_RUN(nz)
  const double two = 2.0, hlf = 0.5, qrt = 0.25;
  const double dxs = MDX(j, i) + MDX(j - 1, i), rmA = RM(j, i), rmB = RM(j - 1, i), vm = VMK(j, i);
  double zwA, zwB, zwA1 = MZW(ka, j, i), zwB1 = MZW(ka, j - 1, i);            // A: row j, B: row j-1
  double dzA, dzB, dzA1 = MDZW(ka, j, i), dzB1 = MDZW(ka, j - 1, i);
  double wA0, wB0, wA1 = ka >= 2 ? Wv(i, j, ka - 1) : 0.0, wB1 = ka >= 2 ? Wv(i, j - 1, ka - 1) : 0.0;
  for (int k = ka; k <= kb; k++) {
    zwA = zwA1; zwB = zwB1; zwA1 = MZW(k + 1, j, i); zwB1 = MZW(k + 1, j - 1, i);
    dzA = dzA1; dzB = dzB1; dzA1 = MDZW(k + 1, j, i); dzB1 = MDZW(k + 1, j - 1, i);
    wA0 = wA1; wB0 = wB1; wA1 = Wv(i, j, k); wB1 = Wv(i, j - 1, k);
    const double zyA = MZYDX(k, j, i), zyB = MZYDX(k, j - 1, i);
    if (k == 1)
      fx[MI3(k, j, i)] =
          (qrt * (zwA1 - zwA + zwB1 - zwB) * dxs * V(i, j, k)
           - qrt * (+zyA * dzA1 * wA1 * rmA +
                    zyB * dzB1 * wB1 * rmB)
           - (+zyA * zyA / (MCW(k, j, i) + MCW(k + 1, j, i)) +
              zyB * zyB / (MCW(k, j - 1, i) + MCW(k + 1, j - 1, i))) *
                 hlf * (MDY(j, i) + MDY(j - 1, i)) * V(i, j, k)
           - (+MZXDY(k, j, i) * zyA / (MCW(k, j, i) + MCW(k + 1, j, i)) * hlf *
                  (hlf * (MDX(j, i) + MDX(j, i - 1)) * U(i, j, k) * UMK(j, i) + hlf * (MDX(j, i + 1) + MDX(j, i)) * U(i + 1, j, k) * UMK(j, i + 1)) +
              MZXDY(k, j - 1, i) * zyB / (MCW(k, j - 1, i) + MCW(k + 1, j - 1, i)) * hlf *
                  (hlf * (MDX(j - 1, i) + MDX(j - 1, i - 1)) * U(i, j - 1, k) * UMK(j - 1, i) +
                   hlf * (MDX(j - 1, i + 1) + MDX(j - 1, i)) * U(i + 1, j - 1, k) * UMK(j - 1, i + 1)))) * vm;
    else if (k <= nz - 1)
      fx[MI3(k, j, i)] =
          (qrt * (zwA1 - zwA + zwB1 - zwB) * dxs * V(i, j, k)
           - qrt * (+zyA * dzA * wA0 * rmA +
                    zyA * dzA1 * wA1 * rmA +
                    zyB * dzB * wB0 * rmB +
                    zyB * dzB1 * wB1 * rmB)) * vm;
    else
      fx[MI3(k, j, i)] =
          (qrt * (zwA1 - zwA + zwB1 - zwB) * dxs * V(i, j, k)
           - qrt * (+zyA * dzA * wA0 * rmA +
                    zyA * two * dzA1 * wA1 * rmA +
                    zyB * dzB * wB0 * rmB +
                    zyB * two * dzB1 * wB1 * rmB)) * vm;
  }
}

__global__ void k_rhs_wf(GeoView G, ModelView M, double *__restrict__ fz, int KR) {
  COLUMN_THREAD_I(1, G.ny, 1, G.nx)
  ROW_RUN(nz + 1)
  const double hlf = 0.5, qrt = 0.25;
  const double dxA = MDX(j, i) + MDX(j, i - 1), dxB = MDX(j, i + 1) + MDX(j, i), dyA = MDY(j, i) + MDY(j - 1, i), dyB = MDY(j + 1, i) + MDY(j, i);
  const double umA = UMK(j, i), umB = UMK(j, i + 1), vmA = VMK(j, i), vmB = VMK(j + 1, i);
  // row k-1 of the step: zxdy, zydx, u at i and i+1, v at j and j+1
  const int k0 = ka >= 2 ? ka - 1 : 1;
  double zx0 = MZXDY(k0, j, i), zy0 = MZYDX(k0, j, i), uA0 = U(i, j, k0), uB0 = U(i + 1, j, k0), vA0 = V(i, j, k0), vB0 = V(i, j + 1, k0);
  for (int k = ka; k <= kb; k++) {
    if (k == 1) fz[MI3(1, j, i)] = 0.0;
    else if (k <= nz) {
      const double zx1 = MZXDY(k, j, i), zy1 = MZYDX(k, j, i), uA1 = U(i, j, k), uB1 = U(i + 1, j, k), vA1 = V(i, j, k), vB1 = V(i, j + 1, k);
      double t = MCW(k, j, i) * MDZW(k, j, i) * Wv(i, j, k - 1) -
                 qrt * hlf * (+zx1 * dxA * uA1 * umA +
                              zx1 * dxB * uB1 * umB +
                              zx0 * dxA * uA0 * umA +
                              zx0 * dxB * uB0 * umB);
      t = t - qrt * hlf * (+zy1 * dyA * vA1 * vmA +
                           zy1 * dyB * vB1 * vmB +
                           zy0 * dyA * vA0 * vmA +
                           zy0 * dyB * vB0 * vmB);
      fz[MI3(k, j, i)] = t;
      zx0 = zx1; zy0 = zy1; uA0 = uA1; uB0 = uB1; vA0 = vA1; vB0 = vB1;
    } else
      fz[MI3(k, j, i)] = MCW(k, j, i) * MDZW(k, j, i) * Wv(i, j, k - 1) -
                         hlf * hlf * (+zx0 * dxA * uA0 * umA +
                                      zx0 * dxB * uB0 * umB) -
                         hlf * hlf * (+zy0 * dyA * vA0 * vmA +
                                      zy0 * dyB * vB0 * vmB);
  }
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

// divergence, accumulated in the reference's order (:173, :274, :362-370) in a model-layout scratch bm:
// bm = uf(i+1)-uf(i) ; bm = bm + vf(j+1)-vf(j) ; bm = bm + wf(k+1)-wf(k) -- one pass over the three flux arrays (the same operations in the
// same order as three passes over bm)
__global__ void k_rhs_accum_m(GeoView G, double *__restrict__ bm, const double *__restrict__ fu, const double *__restrict__ fv, const double *__restrict__ fw, int KR) {
  COLUMN_THREAD_I(1, G.ny, 1, G.nx)
  ROW_RUN(nz)
  for (int k = ka; k <= kb; k++) {
    const long long e = MI3(k, j, i);
    double t = fu[MI3(k, j, i + 1)] - fu[e];
    t = t + fv[MI3(k, j + 1, i)] - fv[e];
    t = t + fw[MI3(k + 1, j, i)] - fw[e];
    bm[e] = t;
  }
}

// correct_uvw (mg_correct_uvw.f90:73-108); pm = level-1 pressure in model layout (halo included).
// dzw(k) of the reference (zr(k)-zr(k-1), top: zw(nz+1)-zr(nz)) is the same expression as grid(1)%dzw.
__global__ void k_correct_uvw_m(GeoView G, const double *__restrict__ pm, ModelView M, int KR) {
  COLUMN_THREAD_I(0, G.ny + 1, 0, G.nx + 1)
  ROW_RUN(nz)
  const double one = 1.0, hlf = 0.5;
#define PM(k, jj, ii) pm[MI3(k, jj, ii)]
  // one climb of the run for the three components: p(k) is read once (and kept for the row above), not once per component
  const bool doU = i >= 1, doV = j >= 1;
  const double dxu = doU ? hlf * (MDX(j, i) + MDX(j, i - 1)) : 1.0, dyv = doV ? hlf * (MDY(j, i) + MDY(j - 1, i)) : 1.0;
  const double um = doU ? UMK(j, i) : 0.0, vm = doV ? VMK(j, i) : 0.0;
  double pk0 = ka >= 2 ? PM(ka - 1, j, i) : 0.0;  // p(k-1)
  for (int k = ka; k <= kb; k++) {
    const double pk = PM(k, j, i);
    if (doU) U(i, j, k) = U(i, j, k) - one / dxu * (pk - PM(k, j, i - 1)) * um;
    if (doV) V(i, j, k) = V(i, j, k) - one / dyv * (pk - PM(k, j - 1, i)) * vm;
    if (k >= 2) {
      const double dzw = MDZW(k, j, i);
      Wv(i, j, k - 1) = Wv(i, j, k - 1) - one / dzw * (pk - pk0);
    }
    pk0 = pk;
  }
  if (kb == nz) {
    const int k = nz + 1;
    const double dzw = MDZW(nz + 1, j, i);
    Wv(i, j, k - 1) = Wv(i, j, k - 1) - one / dzw * (-pk0);
  }
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
// Rows per block in z.  The divergence pass: eight.  The flux kernels and correct_uvw carry values from row to row (a run re-reads one
// row of its neighbour run): runs as long as possible while there are >= 16 384 waves to hide the load -> use chain of a row
// (scripts/probe/ab_model_runs.sh at 512x512x64: runs of 16 rows 0.86 ms for the four kernels, 32 rows 0.91, whole columns 0.91).
static inline dim3 igrid_k(int ni, int nj, int klast, int *KR) { *KR = klast >= 16 ? 8 : klast; return dim3((ni + 63) / 64, (nj + 3) / 4, (klast + *KR - 1) / *KR); }
static inline dim3 igrid_run(int ni, int nj, int klast, int *KR) {
  static const int krenv = getenv("MGX_MODEL_KR") ? atoi(getenv("MGX_MODEL_KR")) : 0;
  const long long waves = (long long)((ni + 63) / 64) * nj;
  long long nrun = krenv > 0 ? (klast + krenv - 1) / krenv : (16384 + waves - 1) / waves;
  if (nrun < 1) nrun = 1;
  if (nrun > (klast + 7) / 8) nrun = (klast + 7) / 8;  // at least eight rows per run
  *KR = (int)((klast + nrun - 1) / nrun);
  return dim3((ni + 63) / 64, (nj + 3) / 4, (klast + *KR - 1) / *KR);
}
void mgxm_rhs_uf(hipStream_t st, const GeoView *G, const ModelView *M, double *fx) { int KR; const dim3 g = igrid_run(G->nx + 1, G->ny, G->nz, &KR); hipLaunchKernelGGL(k_rhs_uf, g, IBLK, 0, st, *G, *M, fx, KR); }
void mgxm_rhs_vf(hipStream_t st, const GeoView *G, const ModelView *M, double *fx) { int KR; const dim3 g = igrid_run(G->nx, G->ny + 1, G->nz, &KR); hipLaunchKernelGGL(k_rhs_vf, g, IBLK, 0, st, *G, *M, fx, KR); }
void mgxm_rhs_wf(hipStream_t st, const GeoView *G, const ModelView *M, double *fz) { int KR; const dim3 g = igrid_run(G->nx, G->ny, G->nz + 1, &KR); hipLaunchKernelGGL(k_rhs_wf, g, IBLK, 0, st, *G, *M, fz, KR); }
void mgxm_flux_zero_face(hipStream_t st, const GeoView *G, double *f, int face, int pl) {
  const int n = (face == 0 ? G->ny : G->nx) + 2;
  hipLaunchKernelGGL(k_flux_zero_face, dim3((n + 63) / 64, G->nz), dim3(64), 0, st, *G, f, face, pl);
}
void mgxm_flux_face_copy(hipStream_t st, const GeoView *G, double *f, double *buf, int face, int pl, int unpack) {
  const int n = face == 0 ? G->ny : G->nx;
  hipLaunchKernelGGL(k_flux_face_copy, dim3((n + 63) / 64, G->nz), dim3(64), 0, st, *G, f, buf, face, pl, unpack);
}
void mgxm_rhs_accum(hipStream_t st, const GeoView *G, double *bm, const double *fu, const double *fv, const double *fw) {
  int KR; const dim3 g = igrid_k(G->nx, G->ny, G->nz, &KR);
  hipLaunchKernelGGL(k_rhs_accum_m, g, IBLK, 0, st, *G, bm, fu, fv, fw, KR);
}
void mgxm_correct_uvw(hipStream_t st, const GeoView *G, const double *pm, const ModelView *M) {
  int KR; const dim3 g = igrid_run(G->nx + 2, G->ny + 2, G->nz, &KR);
  hipLaunchKernelGGL(k_correct_uvw_m, g, IBLK, 0, st, *G, pm, *M, KR);
}
}
