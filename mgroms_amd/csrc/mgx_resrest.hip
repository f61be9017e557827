// Down leg of the V-cycle in one kernel: residual r = b - A p of the fine level (mg_relax.f90:421-515) and its restriction to
// the coarse right-hand side, b_c = sum of the 8 fine r (mg_intergrids.f90:139-162), without r ever leaving the chip.
//
// Vcycle discards the residual norm (mg_solvers.f90:140) and the r of the down leg is dead: nothing but fine2coarse reads
// it, and coarse2fine overwrites it on the way up (mg_intergrids.f90:387-448).  Separately the two operators move
// 87 + 9 B per fine cell at level 1 (r written, r read again, the stored diagonal, neighbour streams fetched by two
// different waves); here one lane owns a COARSE column = a 2x2 block of fine columns, walks it bottom to top with
// three-row windows in registers, shares the block's p / slope / coefficient values between its four residuals, rebuilds the
// diagonal from the couplings (as the smoother does, mgx_relax.hip) and keeps the running 8-cell sum: 48 loads per fine
// row for 4 cells (12 per cell instead of 18 + 1) and no r traffic.
// The four residuals and the 8-term sum use the reference's expressions in the reference's order: b_c is bit-identical
// to compute_residual followed by fine2coarse.  Matrix-free cross terms (needs the slopes zy, zx: the matrix must be the
// one define_matrices built); other cases keep the two separate kernels.
#include <cstdlib>

#include "mgx_device.h"

namespace {

// values of one fine row needed with their k-1 / k+1 neighbours (three-row windows)
struct RowW {
  double P[4][4];   // p at j = jA-1, jA, jB, jB+1 (first index) x i = iA-1, iA, iB, iB+1 (second); corners only at k = 1
  double ZY[2][2];  // own slopes zy at [jj][ii]
  double ZX[2][2];
  double A2[2][2];  // slot 2 at [jj][ii] (couples k with k-1)
};
// values needed at the row itself only
struct RowR {
  double ZYn[2][2];  // zy at jA-1 ([0][ii]) and jB+1 ([1][ii])
  double ZXn[2][2];  // zx at iA-1 ([jj][0]) and iB+1 ([jj][1])
  double A4[3][2];   // slot 4 at jA, jB, jB+1 x ii
  double A7[2][3];   // slot 7 at jj x iA, iB, iB+1
  double B[2][2];
};

struct Geo {
  long long o[4];  // plane offsets of iA-1, iA, iB, iB+1
  int c[4];        // row positions of jA-1, jA, jB, jB+1
};

template <bool CORNERS>
__device__ __forceinline__ void load_w(RowW &w, const LevView &F, const Geo &g, const long long ro) {
  const double *__restrict__ p = F.p, *__restrict__ zy = F.zy, *__restrict__ zx = F.zx, *__restrict__ a2 = F.cA[1];
#pragma unroll
  for (int jx = 0; jx < 4; jx++)
#pragma unroll
    for (int ix = 0; ix < 4; ix++) {
      const bool corner = (jx == 0 || jx == 3) && (ix == 0 || ix == 3);
      if (!corner || CORNERS) w.P[jx][ix] = p[g.o[ix] + ro + g.c[jx]];
    }
#pragma unroll
  for (int jj = 0; jj < 2; jj++)
#pragma unroll
    for (int ii = 0; ii < 2; ii++) {
      const long long e = g.o[ii + 1] + ro + g.c[jj + 1];
      w.ZY[jj][ii] = zy[e]; w.ZX[jj][ii] = zx[e]; w.A2[jj][ii] = a2[e];
    }
}

__device__ __forceinline__ void load_r(RowR &r, const LevView &F, const Geo &g, const long long ro) {
  const double *__restrict__ b = F.b, *__restrict__ zy = F.zy, *__restrict__ zx = F.zx, *__restrict__ a4 = F.cA[3], *__restrict__ a7 = F.cA[6];
#pragma unroll
  for (int ii = 0; ii < 2; ii++) {
    r.ZYn[0][ii] = zy[g.o[ii + 1] + ro + g.c[0]]; r.ZYn[1][ii] = zy[g.o[ii + 1] + ro + g.c[3]];
#pragma unroll
    for (int jx = 1; jx < 4; jx++) r.A4[jx - 1][ii] = a4[g.o[ii + 1] + ro + g.c[jx]];
  }
#pragma unroll
  for (int jj = 0; jj < 2; jj++) {
    r.ZXn[jj][0] = zx[g.o[0] + ro + g.c[jj + 1]]; r.ZXn[jj][1] = zx[g.o[3] + ro + g.c[jj + 1]];
#pragma unroll
    for (int ix = 1; ix < 4; ix++) r.A7[jj][ix - 1] = a7[g.o[ix] + ro + g.c[jj + 1]];
#pragma unroll
    for (int ii = 0; ii < 2; ii++) r.B[jj][ii] = b[g.o[ii + 1] + ro + g.c[jj + 1]];
  }
}

}  // namespace

// grid: 1-D, gx j-chunks of 64 coarse columns x gy groups of blockDim.y coarse planes, XCD-aware as k_relax_nz
template <bool REAL>
__global__ __launch_bounds__(128, 1) void k_residual_restrict(LevView F, LevView C, double *__restrict__ dst, Sides ph, double *__restrict__ zero, int gx, int gy) {
  int bx, by;
  if ((gy & 7) == 0) { const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3; by = xcd * (gy >> 3) + local / gx; bx = local - (local / gx) * gx; }
  else { by = blockIdx.x / gx; bx = blockIdx.x - by * gx; }
  const int j2 = 1 + bx * WAVE + threadIdx.x;
  const int i2 = 1 + by * blockDim.y + threadIdx.y;
  if (j2 > C.ny || i2 > C.nx) return;
  const int nz = F.nz;
  const long long RS = F.RS;
  Geo g;
  const int iA = 2 * i2 - 1;
  g.o[1] = (long long)iA * F.plane; g.o[0] = g.o[1] - F.plane; g.o[2] = g.o[1] + F.plane; g.o[3] = g.o[2] + F.plane;
  g.c[1] = F.HO + (j2 - 1); g.c[2] = F.EO + j2; g.c[0] = F.EO + (j2 - 1); g.c[3] = F.HO + j2;  // jA = 2 j2 - 1 (odd), jB = 2 j2 (even)
  const double qrt = 0.25;
  const double *__restrict__ a1 = F.cA[0], *__restrict__ a5 = F.cA[4], *__restrict__ a8 = F.cA[7];

  RowW Wm, W0, Wp, Wn;
  RowR R0, Rn;
  // first and last row: stored diagonal (their formulas differ, mg_define_matrix.f90:619-627,642-654); k = 1 diagonal slots
  double dfirst[2][2], dlast[2][2], e1[2][2], e2[2][2], e3[2][2], e4[2][2];
#pragma unroll
  for (int jj = 0; jj < 2; jj++)
#pragma unroll
    for (int ii = 0; ii < 2; ii++) {
      const long long o = g.o[ii + 1], op = g.o[ii + 2];
      const int c = g.c[jj + 1], jm = g.c[jj], jp = g.c[jj + 2];
      dfirst[jj][ii] = a1[o + c]; dlast[jj][ii] = a1[o + (long long)(nz - 1) * RS + c];
      if (REAL) { e1[jj][ii] = a5[o + c]; e2[jj][ii] = a5[op + jm]; e3[jj][ii] = a8[o + c]; e4[jj][ii] = a8[op + jp]; }
      else { e1[jj][ii] = e2[jj][ii] = e3[jj][ii] = e4[jj][ii] = 0.0; }
    }
  load_w<true>(W0, F, g, 0);
  load_w<false>(Wp, F, g, RS);
  load_r(R0, F, g, 0);
  Wm = W0;  // never read at k = 1
  double dk1[2][2];  // k = 1 horizontal-diagonal terms (cmatrix = 'real', mg_relax.f90:475-479)
#pragma unroll
  for (int jj = 0; jj < 2; jj++)
#pragma unroll
    for (int ii = 0; ii < 2; ii++) dk1[jj][ii] = 0.0;
  double z = 0.0;
  const long long oc = (long long)i2 * C.plane + jpos(C, j2);
  for (int k = 1; k <= nz; k++) {
    const long long ro = (long long)(k - 1) * RS;
    if (k + 2 <= nz) load_w<false>(Wn, F, g, ro + 2 * RS);
    if (k + 1 <= nz) load_r(Rn, F, g, ro + RS);
    double r[2][2];
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int ii = 0; ii < 2; ii++) {
        const double pc_m = Wm.P[jj + 1][ii + 1], pc_0 = W0.P[jj + 1][ii + 1], pc_p = Wp.P[jj + 1][ii + 1];
        const double pjm_m = Wm.P[jj][ii + 1], pjm_0 = W0.P[jj][ii + 1], pjm_p = Wp.P[jj][ii + 1];
        const double pjp_m = Wm.P[jj + 2][ii + 1], pjp_0 = W0.P[jj + 2][ii + 1], pjp_p = Wp.P[jj + 2][ii + 1];
        const double pim_m = Wm.P[jj + 1][ii], pim_0 = W0.P[jj + 1][ii], pim_p = Wp.P[jj + 1][ii];
        const double pip_m = Wm.P[jj + 1][ii + 2], pip_0 = W0.P[jj + 1][ii + 2], pip_p = Wp.P[jj + 1][ii + 2];
        const double zy_m = Wm.ZY[jj][ii], zy_p = Wp.ZY[jj][ii], zx_m = Wm.ZX[jj][ii], zx_p = Wp.ZX[jj][ii];
        // neighbour slopes of the row: the other column / plane of the block, or the ring outside it
        const double zyjm = jj == 0 ? R0.ZYn[0][ii] : W0.ZY[0][ii], zyjp = jj == 0 ? W0.ZY[1][ii] : R0.ZYn[1][ii];
        const double zxim = ii == 0 ? R0.ZXn[jj][0] : W0.ZX[jj][0], zxip = ii == 0 ? W0.ZX[jj][1] : R0.ZXn[jj][1];
        const double a4o = R0.A4[jj][ii], a4jp = R0.A4[jj + 1][ii], a7o = R0.A7[jj][ii], a7ip = R0.A7[jj][ii + 1];
        const double a2_0 = W0.A2[jj][ii], a2_p = Wp.A2[jj][ii];
        const double c3 = qrt * (zy_p + zyjm), c3m = qrt * (zyjp + zy_m), c5 = -qrt * (zy_m + zyjm), c5m = -qrt * (zyjp + zy_p);
        const double c6 = qrt * (zx_p + zxim), c6m = qrt * (zxip + zx_m), c8 = -qrt * (zx_m + zxim), c8m = -qrt * (zxip + zx_p);
        double rr;
        if (k == 1) {  // mg_relax.f90:464-482
          rr = R0.B[jj][ii] - dfirst[jj][ii] * pc_0 - a2_p * pc_p - c3 * pjm_p - a4o * pjm_0 - a4jp * pjp_0
                            - c5m * pjp_p - c6 * pim_p - a7o * pim_0 - a7ip * pip_0 - c8m * pip_p;
          if (REAL)
            rr = rr - e1[jj][ii] * W0.P[jj + 2][ii] - e2[jj][ii] * W0.P[jj][ii + 2] - e3[jj][ii] * W0.P[jj][ii] - e4[jj][ii] * W0.P[jj + 2][ii + 2];
        } else if (k < nz) {  // :484-496; the diagonal is minus the sum of the fourteen couplings (mg_define_matrix.f90:632-639)
          const double dk = -a2_0 - a2_p - a4o - a4jp - a7o - a7ip - c6 - c6m - c8 - c8m - c3 - c3m - c5 - c5m;
          rr = R0.B[jj][ii] - dk * pc_0 - a2_0 * pc_m - a2_p * pc_p - c3 * pjm_p - c3m * pjp_m
                            - a4o * pjm_0 - a4jp * pjp_0 - c5 * pjm_m - c5m * pjp_p
                            - c6 * pim_p - c6m * pip_m - a7o * pim_0 - a7ip * pip_0
                            - c8 * pim_m - c8m * pip_p;
        } else {  // :498-509
          rr = R0.B[jj][ii] - dlast[jj][ii] * pc_0 - a2_0 * pc_m - c3m * pjp_m - a4o * pjm_0 - a4jp * pjp_0
                            - c5 * pjm_m - c6m * pip_m - a7o * pim_0 - a7ip * pip_0 - c8 * pim_m;
        }
        r[jj][ii] = rr;
      }
    // fine2coarse_3D (mg_intergrids.f90:149-160): (k,jA,iA) + (k,jA,iB) + (k,jB,iA) + (k,jB,iB), then the same of k+1
    if (k & 1) z = r[0][0] + r[0][1] + r[1][0] + r[1][1];
    else {
      z = z + r[0][0] + r[0][1] + r[1][0] + r[1][1];
      const long long rc = (long long)((k >> 1) - 1) * C.RS;
      dst[oc + rc] = z;
      mirror_store(C, dst, rc, j2, i2, jpos(C, j2), z, ph);
      if (zero) { zero[oc + rc] = 0.0; mirror_store(C, zero, rc, j2, i2, jpos(C, j2), 0.0, ph); }
    }
    Wm = W0; W0 = Wp; Wp = Wn; R0 = Rn;
  }
}

extern "C" {

// returns 1 when launched (matrix-free slopes present), 0 = use mgxk_residual + mgxk_fine2coarse
int mgxk_residual_restrict(hipStream_t st, const LevView *F, const LevView *C, double *dst, int real, Sides ph, double *zero) {
  static const bool off = getenv("MGX_NO_RESREST") != nullptr;
  if (off || F->zy == nullptr || F->nz < 2 || (F->nz & 1)) return 0;
  if (C->nx * 2 != F->nx || C->ny * 2 != F->ny) return 0;
  const int by = 2, gx = (C->ny + WAVE - 1) / WAVE, gy = (C->nx + by - 1) / by;
  dim3 blk(WAVE, by), grd(gx * gy);
  if (real) hipLaunchKernelGGL((k_residual_restrict<true>), grd, blk, 0, st, *F, *C, dst, ph, zero, gx, gy);
  else hipLaunchKernelGGL((k_residual_restrict<false>), grd, blk, 0, st, *F, *C, dst, ph, zero, gx, gy);
  return 1;
}

}  // extern "C"
