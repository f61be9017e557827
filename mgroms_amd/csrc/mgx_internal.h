// Internal declarations shared by the HIP translation units of libmgx.so.
#pragma once
#include <hip/hip_runtime.h>

// Device view of one grid level.  Solver fields use the "JS" layout:
//   element (k,j,i), k=1..nz, j=0..ny+1, i=0..nx+1  ->  i*plane + (k-1)*RS + pos(j)
//   pos(j) = (j odd) ? HO + j/2 : EO + j/2
// i.e. i-planes of nz rows; inside a row the even-j and the odd-j columns are stored as two contiguous
// halves.  A colour of the z-line smoother (fixed j parity in a plane) is then a unit-stride run of
// columns, and EO/HO are chosen so that the first interior column of either half is 128-byte aligned.
struct LevView {
  int nx, ny, nz;
  int RS, EO, HO;
  long long plane;  // nz * RS
  double *p, *b, *r;
  double *cA[8];    // slots 1..8 of the reference's cA(8,k,j,i), one JS array each
  double *bet;      // reciprocal pivots of the column tridiagonal (tridiag's `bet`, mg_relax.f90:322-327)
  double *gam;      // tridiag's `gam(k)` (mg_relax.f90:325)
  double *p1;       // snapshot of p(k=1,:,:) for the parallel red-black sweep, (nx+2) rows of RS
  double *p1w;      // when set: the colour pass also writes its new k=1 values here (= the snapshot of the NEXT sweep)
  double *d0w;      // when set (sequential-order red-black, wide half-rows): the colour pass also writes y(k=1) - snapshot of its columns here (2-D, rows of RS): the scan's d0, no launch of its own
  double *zy, *zx;  // slopes ZY, ZX (JS layout) for the matrix-free cross terms; nullptr = use the stored slots
  // Interior rows of slots 4 and 7 rebuilt in the kernel (mg_define_matrix.f90:532-534,549-551) from the interface depths zw, and zw
  // itself from its generating formula (mg_zr_zw.f90:140-145): zw(k,j,i) = z0*h*hinv + zeta*(1.+z0*hinv), z0 = cffw(k) + csw(k)*h --
  // two 1-D tables of nz+1 entries (cffw = hlim*sc_w, csw = Cs_w) and three 2-D fields (h, hinv = 1/(h+hlim), zeta), no 3-D array.
  // 2-D arrays: one row of RS per plane i, JS j-order.  m4 = dx(j,i)+dx(j-1,i), d4 = hlf*(dy(j,i)+dy(j-1,i)),
  // m7 = dy(j,i)+dy(j,i-1), d7 = hlf*(dx(j,i)+dx(j,i-1)).  nullptr = stored slots.
  double *m4, *d4, *m7, *d7, *h2, *hi2, *ze2;
  const double *cffw, *csw;
  // The column's OWN slopes zy, zx rebuilt in the kernel too (ZG, mgx_relax.hip): zr of the four face neighbours from the same formula at
  // the rho points (tables cffr = hlim*sc_r, csr = Cs_r of nz entries, mg_zr_zw.f90:108-122), then the reference's slope expression
  // with the column's dx, dy (dx2, dy2: JS 2-D order).  nullptr = streamed.
  double *dx2, *dy2;
  const double *cffr, *csr;
  // Red-black in the reference's SEQUENTIAL order at streaming speed (option "rb_seq", mgx_rbseq.hip): gk = T^-1 e1 of every column
  // (the response of the column's tridiagonal system to a unit source in its bottom row; matrix only), ag58 = the pairs gk(1) * cA(5,1,j,i),
  // gk(1) * cA(8,1,j,i) (2-D, two values per column, rows of 2 * RS per plane) and u1 = what the plane-by-plane scan found: new minus old
  // p(k=1) of every column of the colour in work (2-D, one row of RS per plane, zero in the halo).  nullptr = not allocated (four colours, cmatrix='simple').
  double *gk, *ag58, *u1;
};

__host__ __device__ inline int jpos(const LevView &L, int j) { return (j & 1) ? L.HO + (j >> 1) : L.EO + (j >> 1); }

// Reference-layout (Fortran order) views used by set-up, compute_rhs and correct_uvw.
struct GeoView {
  int nx, ny, nz;
  double *dx, *dy, *zeta, *h;  // (0:ny+1, 0:nx+1)
  double *zr;                  // (nz,   -1:ny+2, -1:nx+2)
  double *zw;                  // (nz+1, -1:ny+2, -1:nx+2)
  double *cw;                  // (nz+1, 0:ny+1, 0:nx+1)
  double *cA;                  // set-up scratch shared by all levels: SLOT-MAJOR, eight arrays (nz, 0:ny+1, 0:nx+1) one after the other
  double *szy, *szx;           // slopes zy, zx (nz, 0:ny+1, 0:nx+1), scratch shared by all levels (szx = szy + the level's array size)
  double *dzw, *zxdy, *zydx;   // level 1 only
  double *mzw, *mdzw, *mzxdy, *mzydx, *mcw, *mdx, *mdy, *mrmask;  // level 1 only: i-fastest copies read by compute_rhs / correct_uvw (mgx_model.hip)
  double *rmask;               // (0:ny+1, 0:nx+1) boundary / land mask of the level (mg_define_matrix.f90:78-79,157-161)
  int bmask;                   // namelist bmask: masked coefficients
};

// physical-boundary flags of a sub-domain (1 = no neighbour on that side)
// part (colour passes of a level with neighbours, mgx_api.cpp relax()): 0 = every column of the colour; 1 = only the waves that hold a
// column next to a NEIGHBOUR's halo (plane 1 / nx, first / last j-chunk on an open side): the columns the next exchange sends and the only
// ones that read what the last exchange delivered; 2 = all the others.  The two parts run on two streams, the exchange behind part 1.
struct Sides { int S, E, N, W; int part; };
// does the wave of plane i, j-chunk bx (of gx), j parity jodd belong to the part asked for?
__host__ __device__ inline bool sides_part_skip(const Sides &ph, int i, int nx, int jodd, int bx, int gx) {
  if (!ph.part) return false;
  const bool edge = (!ph.W && i == 1) || (!ph.E && i == nx) || (!ph.S && jodd && bx == 0) || (!ph.N && !jodd && bx == gx - 1);
  return edge != (ph.part == 1);
}
