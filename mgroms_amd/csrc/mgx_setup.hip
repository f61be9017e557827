// Set-up kernels (run once per geometry): coarsening of the 2-D geometry, sigma-coordinate depths,
// the 8 stored coefficients per cell, the tridiagonal pivots (compute_rhs / correct_uvw live in mgx_model.hip).
// These work in the reference (Fortran) layout; the coefficients
// are then repacked into the solver's JS layout (mgx_kernels.hip:k_convert).  One lane = one (j,i) column.
// Operation order follows the reference line by line (compiled with -ffp-contract=off).
#include "mgx_internal.h"

#define ZR(k, j, i) G.zr[(((long long)((i) + 1)) * (ny + 4) + ((j) + 1)) * nz + ((k)-1)]
#define ZW(k, j, i) G.zw[(((long long)((i) + 1)) * (ny + 4) + ((j) + 1)) * (nz + 1) + ((k)-1)]
#define A2(a, j, i) a[((long long)(i)) * (ny + 2) + (j)]
#define DX(j, i) A2(G.dx, j, i)
#define DY(j, i) A2(G.dy, j, i)
#define I3(k, j, i) ((((long long)(i)) * (ny + 2) + (j)) * nz + ((k)-1))
#define I3P(k, j, i) ((((long long)(i)) * (ny + 2) + (j)) * (nz + 1) + ((k)-1))
#define CW(k, j, i) G.cw[I3P(k, j, i)]
// set-up scratch of the eight slots: SLOT-MAJOR, eight arrays (nz, 0:ny+1, 0:nx+1) one after the other.  (The reference's record layout
// cA(8,k,j,i) made every store of a k-lane wave hit 64 different 64-byte records with 8 bytes each: 2.2 GB written for 0.8 GB of
// coefficients at 512x512x64, counters of round 3; slot-major a wave writes one 512-byte run per slot.)
#define CA(s, k, j, i) G.cA[(long long)((s)-1) * ((long long)nz * (ny + 2) * (nx + 2)) + I3(k, j, i)]

// umask(j,i) = rmask(j,i-1)*rmask(j,i) for i>=1, vmask(j,i) = rmask(j-1,i)*rmask(j,i) for j>=1, 0 elsewhere; all 1 without bmask
#define RMK(j, i) A2(G.rmask, j, i)
#define UM(j, i) (G.bmask ? (((i) >= 1) ? RMK(j, (i)-1) * RMK(j, i) : 0.0) : 1.0)
#define VM(j, i) (G.bmask ? (((j) >= 1) ? RMK((j)-1, i) * RMK(j, i) : 0.0) : 1.0)

#define COLUMN_THREAD(jlo, jhi, ilo, ihi)                                     \
  const int j = (jlo) + blockIdx.x * blockDim.x + threadIdx.x;                \
  const int i = (ilo) + blockIdx.y * blockDim.y + threadIdx.y;                \
  if (j > (jhi) || i > (ihi)) return;                                         \
  const int nx = G.nx, ny = G.ny, nz = G.nz;                                  \
  (void)nx; (void)ny; (void)nz;

// k-parallel variant for the coefficient kernels: lanes run along k, the fastest index of the reference layout, so a
// wave reads and writes whole k-runs of a column (cA: 64 contiguous bytes per lane).  nk = rows handled per column.
#define KCOL_THREAD(nk, jlo, jhi, ilo, ihi)                                   \
  const int nx = G.nx, ny = G.ny, nz = G.nz;                                  \
  (void)nx; (void)ny; (void)nz;                                               \
  /* 32-bit index arithmetic (a level has < 2^31 cells): the 64-bit divisions of a flat index were a quarter of these kernels' */ \
  /* instructions (they are bound by the count of vector instructions: counters of round 3) */ \
  const unsigned t_ = blockIdx.x * blockDim.x + threadIdx.x;                  \
  const unsigned nk_ = (unsigned)(nk), c_ = t_ / nk_;                         \
  const int k = 1 + (int)(t_ - c_ * nk_);                                     \
  const unsigned nj_ = (unsigned)((jhi) - (jlo) + 1), ci_ = c_ / nj_;         \
  const int j = (jlo) + (int)(c_ - ci_ * nj_);                                \
  const int i = (ilo) + (int)ci_;                                             \
  if (i > (ihi)) return;

// The same for kernels whose first and last level have formulas of their own (k_cA_offdiag, k_cA_diag): with k simply along the lanes every
// wave holds one lane of each special row and runs their long branches for it (the interior rows were a third of k_cA_offdiag's time).
// Here the first nint = nz - 2 rows x columns threads are the interior rows 2 .. nz-1 (k along the lanes), the remaining 2 x columns
// threads the rows 1 and nz (columns along the lanes): a wave runs one branch.  Launch with kgrid(nz, nj, ni) as before.
#define KCOL_THREAD_ENDS(jlo, jhi, ilo, ihi)                                  \
  const int nx = G.nx, ny = G.ny, nz = G.nz;                                  \
  (void)nx; (void)ny; (void)nz;                                               \
  const unsigned t_ = blockIdx.x * blockDim.x + threadIdx.x;                  \
  const unsigned nj_ = (unsigned)((jhi) - (jlo) + 1), ni_ = (unsigned)((ihi) - (ilo) + 1), ncol_ = nj_ * ni_;  \
  const unsigned nint_ = (unsigned)(nz - 2), tint_ = nint_ * ncol_;           \
  unsigned c_; int k;                                                         \
  if (t_ < tint_) { c_ = t_ / nint_; k = 2 + (int)(t_ - c_ * nint_); }        \
  else { const unsigned e_ = t_ - tint_, w_ = e_ / ncol_; if (w_ > 1) return; c_ = e_ - w_ * ncol_; k = w_ ? nz : 1; }  \
  const unsigned ci_ = c_ / nj_;                                              \
  const int j = (jlo) + (int)(c_ - ci_ * nj_);                                \
  const int i = (ilo) + (int)ci_;

// mg_define_matrix.f90:116-138: dx,dy = 1/2 sum4 ; zeta,h = 1/4 sum4.  dst is (0:nyc+1,0:nxc+1)
__global__ void k_coarsen2d(const double *__restrict__ src, double *__restrict__ dst, int nyf, int nyc, int nxc, double fac) {
  const int j = 1 + blockIdx.x * blockDim.x + threadIdx.x, i = 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (j > nyc || i > nxc) return;
  const int fj = 2 * j - 1, fi = 2 * i - 1;
  const long long sf = nyf + 2;
  dst[(long long)i * (nyc + 2) + j] = fac * (src[fi * sf + fj] + src[fi * sf + fj + 1] + src[(fi + 1) * sf + fj] + src[(fi + 1) * sf + fj + 1]);
}

// generic rectangle operation on a reference-layout array a(nzz, 1-nh:ny+nh, 1-nh:nx+nh):
//   op 0: a(dst) = a(src)                    src_j = mj ? cj - j : j + cj ; src_i likewise
//   op 1: a(dst) = 2*a(src) - a(src2)        (nh=2 extrapolation, mg_mpi_exchange.f90:956-964)
//   op 2: a(dst) = 0
//   op 3: buf = a(dst rect)  (pack)          op 4: a(dst rect) = buf (unpack); buffer order (k fastest, then j, then i)
struct RectOp { int op, nzz, nh, ny, j0, j1, i0, i1, mj, cj, mi, ci, mj2, cj2, mi2, ci2; };
__global__ void k_rect(double *__restrict__ a, double *__restrict__ buf, RectOp R) {
  const long long nj = R.j1 - R.j0 + 1, ni = R.i1 - R.i0 + 1, n = nj * ni * R.nzz;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int k = (int)(t % R.nzz);
  const int j = R.j0 + (int)((t / R.nzz) % nj), i = R.i0 + (int)(t / (R.nzz * nj));
  const long long sj = R.ny + 2 * R.nh;
#define AI(jj, ii) (((long long)((ii) + R.nh - 1)) * sj + ((jj) + R.nh - 1)) * R.nzz + k
  const long long d = AI(j, i);
  if (R.op == 2) { a[d] = 0.0; return; }
  if (R.op == 5) { a[d] = 1.0; return; }
  if (R.op == 3) { buf[t] = a[d]; return; }
  if (R.op == 4) { a[d] = buf[t]; return; }
  const int sjj = R.mj ? R.cj - j : j + R.cj, sii = R.mi ? R.ci - i : i + R.ci;
  if (R.op == 0) { a[d] = a[AI(sjj, sii)]; return; }
  const int sj2 = R.mj2 ? R.cj2 - j : j + R.cj2, si2 = R.mi2 ? R.ci2 - i : i + R.ci2;
  a[d] = 2.0 * a[AI(sjj, sii)] - a[AI(sj2, si2)];
#undef AI
}

// fill_halo_2D / fill_halo_3D of a reference-layout array a(nzz, 1-nh:ny+nh, 1-nh:nx+nh) on a level WITHOUT neighbours (every side physical,
// mg_mpi_exchange.f90:509-537 edges, :552-597 corners, :956-964 the nh = 2 extrapolation) in ONE launch: every halo cell's sources are
// interior cells, so the reference's order S, E, N, W, corners does not matter.  (As k_rect operations a fill was 8-12 launches of a few
// hundred threads; 341 launches, 0.9 ms, per rebuild of the 512x512x64 hierarchy.)  One thread per halo cell and level k.
__global__ void k_halo_ref_closed(double *__restrict__ a, int nzz, int nh, int ny, int nx) {
  const unsigned W = ny + 2 * nh, band = nh * W, ring = 2 * band + (unsigned)nx * 2 * nh;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned r = t / nzz;
  if (r >= ring) return;
  const int k = (int)(t - r * nzz);
  int j, i;
  if (r < band) { i = 1 - nh + (int)(r / W); j = 1 - nh + (int)(r % W); }
  else if (r < 2 * band) { const unsigned q = r - band; i = nx + 1 + (int)(q / W); j = 1 - nh + (int)(q % W); }
  else { const unsigned q = r - 2 * band, ii = q / (2 * nh), jj = q - ii * 2 * nh; i = 1 + (int)ii; j = (int)jj < nh ? 1 - nh + (int)jj : ny + 1 + ((int)jj - nh); }
#define AI(jj, ii) ((((long long)((ii) + nh - 1)) * W + ((jj) + nh - 1)) * nzz + k)
  const bool jin = j >= 1 && j <= ny, iin = i >= 1 && i <= nx;
  double v;
  if (iin) {        // south / north edge
    if (j == 0) v = a[AI(1, i)];
    else if (j < 0) v = 2.0 * a[AI(1, i)] - a[AI(2, i)];
    else if (j == ny + 1) v = a[AI(ny, i)];
    else v = 2.0 * a[AI(ny, i)] - a[AI(ny - 1, i)];
  } else if (jin) {  // west / east edge
    if (i == 0) v = a[AI(j, 1)];
    else if (i < 0) v = 2.0 * a[AI(j, 1)] - a[AI(j, 2)];
    else if (i == nx + 1) v = a[AI(j, nx)];
    else v = 2.0 * a[AI(j, nx)] - a[AI(j, nx - 1)];
  } else {          // corner where both sides are physical: the diagonal mirror
    const int sj = j <= 0 ? 1 - j : 2 * ny + 1 - j, si = i <= 0 ? 1 - i : 2 * nx + 1 - i;
    v = a[AI(sj, si)];
  }
  a[AI(j, i)] = v;
#undef AI
}

// mg_zr_zw.f90:98-170 setup_zr_zw_croco, 'new_s_coord', computed on 0:n+1 (:91).  Lanes run along k, the fastest index of zr / zw: a
// wave writes whole k-runs (with a lane per column every store hit 64 lines: 1.3 GB written for 0.27 GB of depths, counters of round 3).
__global__ void k_zr_zw(GeoView G, double hlim, double theta_b, double theta_s) {
  KCOL_THREAD(G.nz + 1, 0, G.ny + 1, 0, G.nx + 1)
  const double one = 1.0, hlf = 0.5, nul = 0.0;
  const double cff = one / (double)nz;
  const double h = A2(G.h, j, i), zeta = A2(G.zeta, j, i);
  const double hinv = one / (h + hlim);
  {
    double cswf, cs_w;
    const double sc_w = cff * (double)(k - 1 - nz);
    if (theta_s > nul) cswf = (one - cosh(theta_s * sc_w)) / (cosh(theta_s) - one); else cswf = -(sc_w * sc_w);
    if (theta_b > nul) cs_w = (exp(theta_b * cswf) - one) / (one - exp(-theta_b)); else cs_w = cswf;
    const double cff_w = hlim * sc_w;
    const double z_w0 = cff_w + cs_w * h;
    ZW(k, j, i) = z_w0 * h * hinv + zeta * (1. + z_w0 * hinv);
    if (k <= nz) {
      double csrf, cs_r;
      const double sc_r = cff * ((double)(k - nz) - hlf);
      if (theta_s > nul) csrf = (one - cosh(theta_s * sc_r)) / (cosh(theta_s) - one); else csrf = -(sc_r * sc_r);
      if (theta_b > nul) cs_r = (exp(theta_b * csrf) - one) / (one - exp(-theta_b)); else cs_r = csrf;
      const double cff_r = hlim * sc_r;
      const double z_r0 = cff_r + cs_r * h;
      ZR(k, j, i) = z_r0 * h * hinv + zeta * (1. + z_r0 * hinv);
    }
  }
}

// mg_define_matrix.f90:283-336: dzw, zxdy, zydx (level 1) and cw, on 0:n+1
__global__ void k_cw(GeoView G, int lev1) {
  KCOL_THREAD(G.nz + 1, 0, G.ny + 1, 0, G.nx + 1)
  const double one = 1.0, hlf = 0.5;
  if (lev1) {
    if (k == 1) G.dzw[I3P(1, j, i)] = ZR(1, j, i) - ZW(1, j, i);
    else if (k <= nz) G.dzw[I3P(k, j, i)] = ZR(k, j, i) - ZR(k - 1, j, i);
    else G.dzw[I3P(nz + 1, j, i)] = ZW(nz + 1, j, i) - ZR(nz, j, i);
    if (k <= nz) {
      G.zydx[I3(k, j, i)] = hlf * ((ZR(k, j + 1, i) - ZR(k, j - 1, i)) / DY(j, i)) * DX(j, i);
      G.zxdy[I3(k, j, i)] = hlf * ((ZR(k, j, i + 1) - ZR(k, j, i - 1)) / DX(j, i)) * DY(j, i);
    }
  }
  const double Arz = DX(j, i) * DY(j, i);
  const double sx = (hlf * (ZW(k, j, i + 1) - ZW(k, j, i - 1)) / DX(j, i));
  const double sy = (hlf * (ZW(k, j + 1, i) - ZW(k, j - 1, i)) / DY(j, i));
  const double den = (k == 1) ? (ZR(k, j, i) - ZW(k, j, i)) : ((k == nz + 1) ? (ZW(k, j, i) - ZR(k - 1, j, i)) : (ZR(k, j, i) - ZR(k - 1, j, i)));
  CW(k, j, i) = (Arz / den) * (one + sx * sx + sy * sy);
}

// The slopes every cross coefficient is built from, once per cell with the reference's inline expression
//   zy(k,j,i) = ( hlf*(zr(k,j+1,i)-zr(k,j-1,i)) / dy(j,i) ) * dx(j,i) ,  zx likewise in i        (mg_define_matrix.f90:358, 398)
// on 0:n+1.  k_cA_offdiag used to evaluate these quotients up to ten times per cell (it is bound by its vector-instruction count: 1360
// per wave, counters of round 3); the smoother's matrix-free slopes (LevView zy / zx) are a layout conversion of the same two arrays.
#define SZY(k, j, i) G.szy[I3(k, j, i)]
#define SZX(k, j, i) G.szx[I3(k, j, i)]
__global__ void k_slopes_ref(GeoView G) {
  KCOL_THREAD(G.nz, 0, G.ny + 1, 0, G.nx + 1)
  const double hlf = 0.5;
  SZY(k, j, i) = (hlf * (ZR(k, j + 1, i) - ZR(k, j - 1, i)) / DY(j, i)) * DX(j, i);
  SZX(k, j, i) = (hlf * (ZR(k, j, i + 1) - ZR(k, j, i - 1)) / DX(j, i)) * DY(j, i);
}

// mg_define_matrix.f90:352-609: off-diagonal slots (bmask = .false., umask = vmask = 1).
// Loop ranges of the reference: slots 3,4,5(k>1): i=1..nx, j=1..ny+1 ; slots 6,7,8(k>1): i=1..nx+1, j=1..ny ;
// cA(5,1): i=1..nx+1, j=0..ny ; cA(8,1): i=1..nx+1, j=1..ny+1 ; cA(2): interior.
// Every thread stores all eight slots of its cell -- the computed value or the zero the reference's zero-initialised array holds outside
// the loop ranges (slot 1: zero here, k_cA_diag fills the interior afterwards) -- over the WHOLE array including the plane i = 0: the
// 1.08 GB scratch needs no clearing pass before every rebuild (0.25 ms of 4 at 512x512x64).
__global__ void k_cA_offdiag(GeoView G) {
  KCOL_THREAD_ENDS(0, G.ny + 1, 0, G.nx + 1)
  const double one = 1.0, qrt = 0.25, hlf = 0.5;
  const bool in345 = (i >= 1) && (i <= nx) && (j >= 1);
  const bool in678 = (i >= 1) && (j >= 1) && (j <= ny);
  double v[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (k == 1) {
  if (in345) {
    v[3] = qrt * ((hlf * (ZR(k + 1, j + 1, i) - ZR(k + 1, j - 1, i)) / DY(j, i)) * DX(j, i) +
                            (hlf * (ZR(k, j, i) - ZR(k, j - 2, i)) / DY(j - 1, i)) * DX(j - 1, i)) * VM(j, i);
    const double t1 = ((hlf * (ZR(k, j + 1, i) - ZR(k, j - 1, i)) / DY(j, i)) * DX(j, i));
    const double t2 = ((hlf * (ZR(k, j, i) - ZR(k, j - 2, i)) / DY(j - 1, i)) * DX(j - 1, i));
    v[4] =
        (qrt * (ZW(k + 1, j, i) - ZW(k, j, i) + ZW(k + 1, j - 1, i) - ZW(k, j - 1, i)) * (DX(j, i) + DX(j - 1, i))) /
            (hlf * (DY(j, i) + DY(j - 1, i)))
        - ((t1 * t1) / (CW(k, j, i) + CW(k + 1, j, i)) + (t2 * t2) / (CW(k, j - 1, i) + CW(k + 1, j - 1, i)))
        - qrt * ((hlf * (ZR(k, j, i) - ZR(k, j - 2, i)) / DY(j - 1, i)) * DX(j - 1, i) -
                 (hlf * (ZR(k, j + 1, i) - ZR(k, j - 1, i)) / DY(j, i)) * DX(j, i));
    if (G.bmask)  // :375-389
      v[4] = (v[4]
          - (hlf * ((hlf * (ZR(k, j - 1, i + 1) - ZR(k, j - 1, i - 1)) / DX(j - 1, i)) * DY(j - 1, i)) *
                 ((hlf * (ZR(k, j, i) - ZR(k, j - 2, i)) / DY(j - 1, i)) * DX(j - 1, i)) /
                 (CW(k, j - 1, i) + CW(k + 1, j - 1, i)) * (UM(j - 1, i + 1) - UM(j - 1, i))
             - hlf * ((hlf * (ZR(k, j, i + 1) - ZR(k, j, i - 1)) / DY(j, i)) * DX(j, i)) *
                   ((hlf * (ZR(k, j + 1, i) - ZR(k, j - 1, i)) / DY(j, i)) * DX(j, i)) /
                   (CW(k, j, i) + CW(k + 1, j, i)) * (UM(j, i + 1) - UM(j, i)))) * VM(j, i);
  }
  if (in678) {
    v[6] = qrt * ((hlf * (ZR(k + 1, j, i + 1) - ZR(k + 1, j, i - 1)) / DX(j, i)) * DY(j, i) +
                            (hlf * (ZR(k, j, i) - ZR(k, j, i - 2)) / DX(j, i - 1)) * DY(j, i - 1)) * UM(j, i);
    const double t1 = ((hlf * (ZR(k, j, i + 1) - ZR(k, j, i - 1)) / DX(j, i)) * DY(j, i));
    const double t2 = ((hlf * (ZR(k, j, i) - ZR(k, j, i - 2)) / DX(j, i - 1)) * DY(j, i - 1));
    v[7] =
        (qrt * (ZW(k + 1, j, i) - ZW(k, j, i) + ZW(k + 1, j, i - 1) - ZW(k, j, i - 1)) * (DY(j, i) + DY(j, i - 1))) /
            (hlf * (DX(j, i) + DX(j, i - 1)))
        - ((t1 * t1) / (CW(k, j, i) + CW(k + 1, j, i)) + (t2 * t2) / (CW(k, j, i - 1) + CW(k + 1, j, i - 1)))
        - qrt * ((hlf * (ZR(k, j, i) - ZR(k, j, i - 2)) / DX(j, i - 1)) * DY(j, i - 1) -
                 (hlf * (ZR(k, j, i + 1) - ZR(k, j, i - 1)) / DX(j, i)) * DY(j, i));
    if (G.bmask)  // :417-433
      v[7] = (v[7]
          - (hlf * ((hlf * (ZR(k, j, i) - ZR(k, j, i - 2)) / DX(j, i - 1)) * DY(j, i - 1)) *
                 ((hlf * (ZR(k, j + 1, i - 1) - ZR(k, j - 1, i - 1)) / DY(j, i - 1)) * DX(j, i - 1)) /
                 (CW(k, j, i - 1) + CW(k + 1, j, i - 1)) * (VM(j + 1, i - 1) - VM(j, i - 1))
             - hlf * ((hlf * (ZR(k, j, i + 1) - ZR(k, j, i - 1)) / DY(j, i)) * DX(j, i)) *
                   ((hlf * (ZR(k, j + 1, i) - ZR(k, j - 1, i)) / DY(j, i)) * DX(j, i)) /
                   (CW(k, j, i) + CW(k + 1, j, i)) * (VM(j + 1, i) - VM(j, i)))) * UM(j, i);
  }
  if (i >= 1 && j <= ny) {
    v[5] =
        +hlf * ((hlf * (ZR(k, j + 1, i + 1) - ZR(k, j + 1, i - 1)) / DX(j + 1, i)) * DY(j + 1, i)) *
                ((hlf * (ZR(k, j + 2, i) - ZR(k, j, i)) / DY(j + 1, i)) * DX(j + 1, i)) /
                (CW(k, j + 1, i) + CW(k + 1, j + 1, i)) * UM(j + 1, i) * VM(j + 1, i)
        + hlf * ((hlf * (ZR(k, j, i) - ZR(k, j, i - 2)) / DX(j, i - 1)) * DY(j, i - 1)) *
                ((hlf * (ZR(k, j + 1, i - 1) - ZR(k, j - 1, i - 1)) / DY(j, i - 1)) * DX(j, i - 1)) /
                (CW(k, j, i - 1) + CW(k + 1, j, i - 1)) * UM(j, i) * VM(j + 1, i - 1);
  }
  if (i >= 1 && j >= 1) {
    v[8] =
        -hlf * ((hlf * (ZR(k, j - 1, i + 1) - ZR(k, j - 1, i - 1)) / DX(j - 1, i)) * DY(j - 1, i)) *
                ((hlf * (ZR(k, j, i) - ZR(k, j - 2, i)) / DY(j - 1, i)) * DX(j - 1, i)) /
                (CW(k, j - 1, i) + CW(k + 1, j - 1, i)) * UM(j - 1, i) * VM(j, i)
        - hlf * ((hlf * (ZR(k, j, i) - ZR(k, j, i - 2)) / DX(j, i - 1)) * DY(j, i - 1)) *
                ((hlf * (ZR(k, j + 1, i - 1) - ZR(k, j - 1, i - 1)) / DY(j, i - 1)) * DX(j, i - 1)) /
                (CW(k, j, i - 1) + CW(k + 1, j, i - 1)) * UM(j, i) * VM(j, i - 1);
  }
  } else if (k <= nz - 1) {
    if (in345 && j <= ny) {
      v[2] = CW(k, j, i);
      if (G.bmask)  // :497-509
        v[2] = v[2]
            - qrt * ((hlf * (ZR(k - 1, j, i + 1) - ZR(k - 1, j, i - 1)) / DX(j, i)) * DY(j, i) -
                     (hlf * (ZR(k, j, i + 1) - ZR(k, j, i - 1)) / DX(j, i)) * DY(j, i)) * (UM(j, i + 1) - UM(j, i))
            - qrt * ((hlf * (ZR(k - 1, j + 1, i) - ZR(k - 1, j - 1, i)) / DY(j, i)) * DX(j, i) -
                     (hlf * (ZR(k, j + 1, i) - ZR(k, j - 1, i)) / DY(j, i)) * DX(j, i)) * (VM(j + 1, i) - VM(j, i));
    }
    if (in345) {  // the slope terms are the stored zy (same expression, evaluated once per cell by k_slopes_ref)
      v[3] = qrt * (SZY(k + 1, j, i) + SZY(k, j - 1, i)) * VM(j, i);
      v[4] = (qrt * (ZW(k + 1, j, i) - ZW(k, j, i) + ZW(k + 1, j - 1, i) - ZW(k, j - 1, i)) *
                        (DX(j, i) + DX(j - 1, i))) / (hlf * (DY(j, i) + DY(j - 1, i))) * VM(j, i);
      v[5] = -qrt * ((SZY(k - 1, j, i)) + (SZY(k, j - 1, i))) * VM(j, i);
    }
    if (in678) {
      v[6] = qrt * ((SZX(k + 1, j, i)) + (SZX(k, j, i - 1))) * UM(j, i);
      v[7] = (qrt * (ZW(k + 1, j, i) - ZW(k, j, i) + ZW(k + 1, j, i - 1) - ZW(k, j, i - 1)) *
                        (DY(j, i) + DY(j, i - 1))) / (hlf * (DX(j, i) + DX(j, i - 1))) * UM(j, i);
      v[8] = -qrt * ((SZX(k - 1, j, i)) + (SZX(k, j, i - 1))) * UM(j, i);
    }
  } else {
  if (in345 && j <= ny) v[2] = CW(k, j, i);
  if (in345) {
    v[4] =
        (qrt * (ZW(k + 1, j, i) - ZW(k, j, i) + ZW(k + 1, j - 1, i) - ZW(k, j - 1, i)) * (DX(j, i) + DX(j - 1, i)) /
             (hlf * (DY(j, i) + DY(j - 1, i)))
         + qrt * (-((hlf * (ZR(k, j, i) - ZR(k, j - 2, i)) / DY(j - 1, i)) * DX(j - 1, i))
                  + ((hlf * (ZR(k, j + 1, i) - ZR(k, j - 1, i)) / DY(j, i)) * DX(j, i)))) * VM(j, i);
    v[5] = -qrt * (((hlf * (ZR(k - 1, j + 1, i) - ZR(k - 1, j - 1, i)) / DY(j, i)) * DX(j, i)) +
                             ((hlf * (ZR(k, j, i) - ZR(k, j - 2, i)) / DY(j - 1, i)) * DX(j - 1, i))) * VM(j, i);
  }
  if (in678) {
    v[7] =
        (qrt * (ZW(k + 1, j, i) - ZW(k, j, i) + ZW(k + 1, j, i - 1) - ZW(k, j, i - 1)) * (DY(j, i) + DY(j, i - 1)) /
             (hlf * (DX(j, i) + DX(j, i - 1)))
         + qrt * (-((hlf * (ZR(k, j, i) - ZR(k, j, i - 2)) / DX(j, i - 1)) * DY(j, i - 1))
                  + ((hlf * (ZR(k, j, i + 1) - ZR(k, j, i - 1)) / DX(j, i)) * DY(j, i)))) * UM(j, i);
    v[8] = -qrt * (((hlf * (ZR(k - 1, j, i + 1) - ZR(k - 1, j, i - 1)) / DX(j, i)) * DY(j, i)) +
                             ((hlf * (ZR(k, j, i) - ZR(k, j, i - 2)) / DX(j, i - 1)) * DY(j, i - 1))) * UM(j, i);
  }
  }
#pragma unroll
  for (int s = 1; s <= 8; s++) CA(s, k, j, i) = v[s];
}

// mg_define_matrix.f90:616-657: diagonal, interior columns
__global__ void k_cA_diag(GeoView G) {
  KCOL_THREAD_ENDS(1, G.ny, 1, G.nx)
  const double hlf = 0.5;
  if (k == 1)
  CA(1, k, j, i) = -CA(2, k + 1, j, i) - CA(4, k, j, i) - CA(4, k, j + 1, i) - CA(7, k, j, i) - CA(7, k, j, i + 1)
                   - CA(6, k, j, i) - CA(8, k + 1, j, i + 1) - CA(3, k, j, i) - CA(5, k + 1, j + 1, i)
                   - CA(5, k, j, i) - CA(5, k, j - 1, i + 1) - CA(8, k, j, i) - CA(8, k, j + 1, i + 1);
  else if (k <= nz - 1)
    CA(1, k, j, i) = -CA(2, k, j, i) - CA(2, k + 1, j, i) - CA(4, k, j, i) - CA(4, k, j + 1, i) - CA(7, k, j, i)
                     - CA(7, k, j, i + 1) - CA(6, k, j, i) - CA(6, k - 1, j, i + 1) - CA(8, k, j, i)
                     - CA(8, k + 1, j, i + 1) - CA(3, k, j, i) - CA(3, k - 1, j + 1, i) - CA(5, k, j, i)
                     - CA(5, k + 1, j + 1, i);
  else
  CA(1, k, j, i) = -CA(2, k, j, i) - CW(k + 1, j, i)
                   + hlf * (hlf * (ZR(k, j, i + 2) - ZR(k, j, i)) / DX(j, i + 1)) * DY(j, i + 1)
                   - hlf * (hlf * (ZR(k, j, i) - ZR(k, j, i - 2)) / DX(j, i - 1)) * DY(j, i - 1)
                   + hlf * (hlf * (ZR(k, j + 2, i) - ZR(k, j, i)) / DY(j + 1, i)) * DX(j + 1, i)
                   - hlf * (hlf * (ZR(k, j, i) - ZR(k, j - 2, i)) / DY(j - 1, i)) * DX(j - 1, i)
                   - CA(4, k, j, i) - CA(4, k, j + 1, i) - CA(7, k, j, i) - CA(7, k, j, i + 1)
                   - CA(6, k - 1, j, i + 1) - CA(8, k, j, i) - CA(3, k - 1, j + 1, i) - CA(5, k, j, i);
}

// What the colour pass needs to rebuild the interior rows of slots 4 and 7 instead of streaming them (mg_define_matrix.f90:532-534,
// 549-551):
//   cA(4,k,j,i) = ( qrt*(zw(k+1,j,i)-zw(k,j,i)+zw(k+1,j-1,i)-zw(k,j-1,i)) * (dx(j,i)+dx(j-1,i)) ) / ( hlf*(dy(j,i)+dy(j-1,i)) )
//   cA(7,k,j,i) = ( qrt*(zw(k+1,j,i)-zw(k,j,i)+zw(k+1,j,i-1)-zw(k,j,i-1)) * (dy(j,i)+dy(j,i-1)) ) / ( hlf*(dx(j,i)+dx(j,i-1)) )
// with zw regenerated from h, hinv, zeta and the two sigma tables (mg_zr_zw.f90:104-145): the 2-D factors and fields, JS j-order.
__global__ void k_zw_js(GeoView G, LevView L, double hlim) {
  COLUMN_THREAD(0, G.ny + 1, 0, G.nx + 1)
  const double hlf = 0.5, one = 1.0;
  const long long o2 = (long long)i * L.RS + jpos(L, j);
  L.m4[o2] = j >= 1 ? DX(j, i) + DX(j - 1, i) : 0.0;
  L.d4[o2] = j >= 1 ? hlf * (DY(j, i) + DY(j - 1, i)) : 1.0;
  L.m7[o2] = i >= 1 ? DY(j, i) + DY(j, i - 1) : 0.0;
  L.d7[o2] = i >= 1 ? hlf * (DX(j, i) + DX(j, i - 1)) : 1.0;
  const double h = A2(G.h, j, i);
  L.h2[o2] = h;
  L.hi2[o2] = one / (h + hlim);   // hinv of mg_zr_zw.f90:106, the same expression as k_zr_zw
  L.ze2[o2] = A2(G.zeta, j, i);
  if (L.dx2) { L.dx2[o2] = DX(j, i); L.dy2[o2] = DY(j, i); }
}
// sigma tables of the rho-points, k = 1..nz: cffr(k) = hlim*sc_r, csr(k) = Cs_r (mg_zr_zw.f90:108-122), same expressions as k_zr_zw
__global__ void k_sigma_tables_r(int nz, double hlim, double theta_b, double theta_s, double *cffr, double *csr) {
  const int k = 1 + blockIdx.x * blockDim.x + threadIdx.x;
  if (k > nz) return;
  const double one = 1.0, hlf = 0.5, nul = 0.0;
  const double cff = one / (double)nz;
  double csrf, cs_r;
  const double sc_r = cff * ((double)(k - nz) - hlf);
  if (theta_s > nul) csrf = (one - cosh(theta_s * sc_r)) / (cosh(theta_s) - one); else csrf = -(sc_r * sc_r);
  if (theta_b > nul) cs_r = (exp(theta_b * csrf) - one) / (one - exp(-theta_b)); else cs_r = csrf;
  cffr[k - 1] = hlim * sc_r;
  csr[k - 1] = cs_r;
}
// sigma tables of the w-points, k = 1..nz+1: cffw(k) = hlim*sc_w, csw(k) = Cs_w (mg_zr_zw.f90:124-141), same expressions as k_zr_zw
__global__ void k_sigma_tables(int nz, double hlim, double theta_b, double theta_s, double *cffw, double *csw) {
  const int k = 1 + blockIdx.x * blockDim.x + threadIdx.x;
  if (k > nz + 1) return;
  const double one = 1.0, nul = 0.0;
  const double cff = one / (double)nz;
  double cswf, cs_w;
  const double sc_w = cff * (double)(k - 1 - nz);
  if (theta_s > nul) cswf = (one - cosh(theta_s * sc_w)) / (cosh(theta_s) - one); else cswf = -(sc_w * sc_w);
  if (theta_b > nul) cs_w = (exp(theta_b * cswf) - one) / (one - exp(-theta_b)); else cs_w = cswf;
  cffw[k - 1] = hlim * sc_w;
  csw[k - 1] = cs_w;
}

// tridiagonal pivots of every interior column (mg_relax.f90:322-327): bet(1)=1/d(1);
// gam(k)=dd(k-1)*bet ; bet(k)=1/(d(k)-dd(k-1)*gam(k)) with d=cA(1,:), dd(k-1)=cA(2,k)
__global__ void k_pivots(LevView L) {
  const int jj = 1 + blockIdx.x * blockDim.x + threadIdx.x, i = 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (jj > L.ny || i > L.nx) return;
  const long long o = (long long)i * L.plane + jpos(L, jj);
  const double *__restrict__ d = L.cA[0], *__restrict__ dd = L.cA[1];
  double bet = 1.0 / d[o];
  L.bet[o] = bet;
  L.gam[o] = 0.0;
  for (int k = 2; k <= L.nz; k++) {
    const long long ko = o + (long long)(k - 1) * L.RS;
    const double g = dd[ko] * bet;
    bet = 1.0 / (d[ko] - dd[ko] * g);
    L.gam[ko] = g;
    L.bet[ko] = bet;
  }
}

// ------------------------------------------------------------------------------------------------
// compute_rhs (mg_compute_rhs.f90:14-379, bmask = .false.) on level 1.  u,v,w are the model's
// (i,j,k)-ordered arrays on the device; uf/vf share `fx` (nz,0:ny+1,0:nx+1), wf is `fz` (nz+1,..).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
static inline dim3 cgrid(int nj, int ni) { return dim3((nj + 63) / 64, (ni + 3) / 4); }
static const dim3 CBLK(64, 4);
static inline dim3 kgrid(int nk, int nj, int ni) { return dim3((unsigned)(((long long)nk * nj * ni + 255) / 256)); }

extern "C" {
void mgxs_coarsen2d(hipStream_t st, const double *src, double *dst, int nyf, int nyc, int nxc, double fac) {
  hipLaunchKernelGGL(k_coarsen2d, cgrid(nyc, nxc), CBLK, 0, st, src, dst, nyf, nyc, nxc, fac);
}
void mgxs_rect(hipStream_t st, double *a, double *buf, const RectOp *R) {
  const long long n = (long long)(R->j1 - R->j0 + 1) * (R->i1 - R->i0 + 1) * R->nzz;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_rect, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, buf, *R);
}
void mgxs_halo_ref_closed(hipStream_t st, double *a, int nzz, int nh, int ny, int nx) {
  const long long n = (long long)nzz * (2LL * nh * (ny + 2 * nh) + 2LL * nh * nx);
  hipLaunchKernelGGL(k_halo_ref_closed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, nzz, nh, ny, nx);
}
void mgxs_zr_zw(hipStream_t st, const GeoView *G, double hlim, double theta_b, double theta_s) {
  hipLaunchKernelGGL(k_zr_zw, kgrid(G->nz + 1, G->ny + 2, G->nx + 2), dim3(256), 0, st, *G, hlim, theta_b, theta_s);
}
void mgxs_define_matrix(hipStream_t st, const GeoView *G, int lev1, int phase) {
  if (phase == 0) {
    hipLaunchKernelGGL(k_cw, kgrid(G->nz + 1, G->ny + 2, G->nx + 2), dim3(256), 0, st, *G, lev1);
    hipLaunchKernelGGL(k_cA_offdiag, kgrid(G->nz, G->ny + 2, G->nx + 2), dim3(256), 0, st, *G);
  } else {
    hipLaunchKernelGGL(k_cA_diag, kgrid(G->nz, G->ny, G->nx), dim3(256), 0, st, *G);
  }
}
void mgxs_slopes_ref(hipStream_t st, const GeoView *G) { hipLaunchKernelGGL(k_slopes_ref, kgrid(G->nz, G->ny + 2, G->nx + 2), dim3(256), 0, st, *G); }
void mgxs_zw_js(hipStream_t st, const GeoView *G, const LevView *L, double hlim, double theta_b, double theta_s) {
  hipLaunchKernelGGL(k_zw_js, cgrid(G->ny + 2, G->nx + 2), CBLK, 0, st, *G, *L, hlim);
  hipLaunchKernelGGL(k_sigma_tables, dim3((G->nz + 1 + 63) / 64), dim3(64), 0, st, G->nz, hlim, theta_b, theta_s, (double *)L->cffw, (double *)L->csw);
  if (L->cffr) hipLaunchKernelGGL(k_sigma_tables_r, dim3((G->nz + 63) / 64), dim3(64), 0, st, G->nz, hlim, theta_b, theta_s, (double *)L->cffr, (double *)L->csr);
}
void mgxs_pivots(hipStream_t st, const LevView *L) { hipLaunchKernelGGL(k_pivots, cgrid(L->ny, L->nx), CBLK, 0, st, *L); }
}
