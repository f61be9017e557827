// Colour pass of the z-line smoother for the MID levels (nz = 32, 16 of the 512x512x64 hierarchy): rows split over the
// waves of a workgroup ("k-split").  mg_relax.f90:237-305 (relax_3D_8_heart) + :308-334 (tridiag), matrix-free cross terms.
//
// Why: a colour of a 256x256x32 level is 16 384 columns = 256 waves for 1024 SIMDs, and a wave that walks its column row
// by row keeps at most 63 loads in flight (the vmcnt limit) through ~9 dependent round trips: the level-2 pass ran at
// 3.2 TB/s out of the Infinity Cache, 15 us per pass, latency-bound.  But only two flops per row are sequential in k
// (x(k) = (rhs(k) - a2(k) x(k-1)) bet(k)); the right-hand side -- 15 of the 17 streams and 50 of the 54 flops -- is not.
// So NW waves share one set of 64 columns: wave w builds rhs(k) for its NZ/NW rows (all its loads issued at once),
// parks them in LDS, wave 0 runs the two sweeps of the recurrence from LDS, and all waves store their rows of p.
// Four (or eight) times the loads in flight per column, two round trips instead of nine.
// Same expressions in the same order as relax_col_mf: bit-identical results.
#include <cstdlib>

#include "mgx_device.h"

template <int NZ, int NW, bool REAL, bool SNAP>
__global__ __launch_bounds__(64 * NW, 1) void k_relax_ks(LevView L, int i0, int istep, int nplanes, int jodd_fixed, int rb, Sides ph, int gx) {
  constexpr int R = NZ / NW;  // rows per wave
  __shared__ double sh[NZ * WAVE], sa2[NZ * WAVE], sbt[NZ * WAVE];  // rhs(k) then x(k); a2(k); bet(k): [k-1][lane]
  // XCD-aware block -> (j-chunk, plane) map, as k_relax_nz: each XCD owns a contiguous range of planes (speed only)
  int bx, ipl;
  if (gx < 0) { gx = -gx; ipl = blockIdx.x / gx; bx = blockIdx.x - ipl * gx; }
  else if ((nplanes & 7) == 0) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    ipl = xcd * (nplanes >> 3) + local / gx;
    bx = local - (local / gx) * gx;
  } else { ipl = blockIdx.x / gx; bx = blockIdx.x - ipl * gx; }
  const int lane = threadIdx.x, w = threadIdx.y;
  const int jh = bx * WAVE + lane;
  const bool live = jh < (L.ny >> 1);  // ragged last chunk: dead lanes still join the barriers
  const int i = i0 + istep * ipl;
  // RB: j = 1+mod(i+rb,2),ny,2 (mg_relax.f90:174) ; FC: fixed parity (:216-217)
  const int jodd = jodd_fixed >= 0 ? jodd_fixed : (((i + rb) & 1) == 0);
  int c, jm, jp;
  if (jodd) { c = L.HO + jh; jm = L.EO + jh; jp = jm + 1; }
  else      { c = L.EO + jh + 1; jm = L.HO + jh; jp = jm + 1; }
  const long long RS = L.RS;
  double *__restrict__ p = L.p;
  const double *__restrict__ b = L.b;
  const double *__restrict__ a2 = L.cA[1], *__restrict__ a4 = L.cA[3], *__restrict__ a5 = L.cA[4], *__restrict__ a7 = L.cA[6],
               *__restrict__ a8 = L.cA[7], *__restrict__ bet = L.bet, *__restrict__ zy = L.zy, *__restrict__ zx = L.zx;
  const long long o = (long long)i * L.plane, om = o - L.plane, op = o + L.plane;
  const double qrt = 0.25;
  const int ka = w * R + 1;  // first row of this wave

  // ---- phase 1: right-hand sides of rows ka .. ka+R-1 (every load of the wave is issued before the first use)
  double pjm[R + 2], pim[R + 2], pjp[R + 2], pip[R + 2], zyo[R + 2], zxo[R + 2];  // rows ka-1 .. ka+R
  double bb[R], a4o[R], a7o[R], a4n[R], a7n[R], zyjm[R], zyjp[R], zxim[R], zxip[R];
  double oa2[R], obt[R];  // the recurrence's coefficients of the wave's rows: handed to wave 0 through LDS
  double d1 = 0, d2 = 0, d3 = 0, d4 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
  if (live) {
#pragma unroll
    for (int r = 0; r < R + 2; r++) {
      const int k = ka - 1 + r;
      if (k >= 1 && k <= NZ) {
        const long long ro = (long long)(k - 1) * RS;
        pjm[r] = p[o + ro + jm]; pim[r] = p[om + ro + c]; pjp[r] = p[o + ro + jp]; pip[r] = p[op + ro + c];
        zyo[r] = zy[o + ro + c]; zxo[r] = zx[o + ro + c];
      } else { pjm[r] = pim[r] = pjp[r] = pip[r] = zyo[r] = zxo[r] = 0.0; }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const long long ro = (long long)(ka + r - 1) * RS;
      bb[r] = b[o + ro + c]; a4o[r] = a4[o + ro + c]; a7o[r] = a7[o + ro + c];
      a4n[r] = a4[o + ro + jp]; a7n[r] = a7[op + ro + c];
      zyjm[r] = zy[o + ro + jm]; zyjp[r] = zy[o + ro + jp]; zxim[r] = zx[om + ro + c]; zxip[r] = zx[op + ro + c];
      oa2[r] = a2[o + ro + c]; obt[r] = bet[o + ro + c];
    }
    if (w == 0) {
      if (REAL) {
        const double *__restrict__ q1 = SNAP ? L.p1 : p;
        const long long s = SNAP ? (long long)i * RS : o, sm = SNAP ? s - RS : om, sp = SNAP ? s + RS : op;
        d1 = q1[sm + jp]; d2 = q1[sp + jm]; d3 = q1[sm + jm]; d4 = q1[sp + jp];
        e1 = a5[o + c]; e2 = a5[op + jm]; e3 = a8[o + c]; e4 = a8[op + jp];
      }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int k = ka + r;
      // three-row windows: index r = row k-1, r+1 = row k, r+2 = row k+1
      const double pjm_m = pjm[r], pjm_0 = pjm[r + 1], pjm_p = pjm[r + 2], pim_m = pim[r], pim_0 = pim[r + 1], pim_p = pim[r + 2];
      const double pjp_m = pjp[r], pjp_0 = pjp[r + 1], pjp_p = pjp[r + 2], pip_m = pip[r], pip_0 = pip[r + 1], pip_p = pip[r + 2];
      const double zy_m = zyo[r], zy_p = zyo[r + 2], zx_m = zxo[r], zx_p = zxo[r + 2];
      const double c3 = qrt * (zy_p + zyjm[r]), c3m = qrt * (zyjp[r] + zy_m), c5 = -qrt * (zy_m + zyjm[r]), c5m = -qrt * (zyjp[r] + zy_p);
      const double c6 = qrt * (zx_p + zxim[r]), c6m = qrt * (zxip[r] + zx_m), c8 = -qrt * (zx_m + zxim[r]), c8m = -qrt * (zxip[r] + zx_p);
      double rhs;
      if (k == 1) {
        rhs = bb[r] - c3 * pjm_p - a4o[r] * pjm_0 - a4n[r] * pjp_0 - c5m * pjp_p
                    - c6 * pim_p - a7o[r] * pim_0 - a7n[r] * pip_0 - c8m * pip_p;
        if (REAL) rhs = rhs - e1 * d1 - e2 * d2 - e3 * d3 - e4 * d4;
      } else if (k < NZ) {
        rhs = bb[r] - c3 * pjm_p - c3m * pjp_m - a4o[r] * pjm_0 - a4n[r] * pjp_0
                    - c5 * pjm_m - c5m * pjp_p
                    - c6 * pim_p - c6m * pip_m - a7o[r] * pim_0 - a7n[r] * pip_0
                    - c8 * pim_m - c8m * pip_p;
      } else {
        rhs = bb[r] - c3m * pjp_m - a4o[r] * pjm_0 - a4n[r] * pjp_0 - c5 * pjm_m
                    - c6m * pip_m - a7o[r] * pim_0 - a7n[r] * pip_0 - c8 * pim_m;
      }
      sh[(k - 1) * WAVE + lane] = rhs;
      sa2[(k - 1) * WAVE + lane] = oa2[r];
      sbt[(k - 1) * WAVE + lane] = obt[r];
    }
  }
  __syncthreads();
  // ---- phase 2 (wave 0): tridiag (mg_relax.f90:322-332) on the parked right-hand sides
  if (w == 0 && live) {
    double x[NZ], g[NZ], ra2[NZ], rbt[NZ];
#pragma unroll
    for (int k = 1; k <= NZ; k++) { x[k - 1] = sh[(k - 1) * WAVE + lane]; ra2[k - 1] = sa2[(k - 1) * WAVE + lane]; rbt[k - 1] = sbt[(k - 1) * WAVE + lane]; }
    double xv = x[0] * rbt[0];
    x[0] = xv;
    g[0] = 0.0;
#pragma unroll
    for (int k = 2; k <= NZ; k++) {
      g[k - 1] = ra2[k - 1] * rbt[k - 2];                 // gam(k) = dd(k-1)*bet(k-1)
      xv = (x[k - 1] - ra2[k - 1] * xv) * rbt[k - 1];     // xc(k) = (b(k) - dd(k-1)*xc(k-1))*bet(k)
      x[k - 1] = xv;
    }
#pragma unroll
    for (int k = NZ - 1; k >= 1; k--) x[k - 1] = x[k - 1] - g[k] * x[k];
#pragma unroll
    for (int k = 1; k <= NZ; k++) sh[(k - 1) * WAVE + lane] = x[k - 1];
    if (SNAP && L.p1w != nullptr) {  // next sweep's k=1 snapshot entry and its physical mirrors (see relax_col_nz)
      const int j = jodd ? 2 * jh + 1 : 2 * jh + 2;
      const bool mS = ph.S && j == 1, mN = ph.N && j == L.ny, mW = ph.W && i == 1, mE = ph.E && i == L.nx;
      const int cS = L.EO, cN = jpos(L, L.ny + 1);
      double *w1 = L.p1w, *r1 = L.p1;
      const long long so = (long long)i * RS, sW = 0, sE = (long long)(L.nx + 1) * RS;
      const double v1 = x[0];
      w1[so + c] = v1;
#define SNAP_MIRROR(idx) { w1[idx] = v1; r1[idx] = v1; }
      if (mS) SNAP_MIRROR(so + cS)
      if (mN) SNAP_MIRROR(so + cN)
      if (mW) { SNAP_MIRROR(sW + c) if (mS) SNAP_MIRROR(sW + cS) if (mN) SNAP_MIRROR(sW + cN) }
      if (mE) { SNAP_MIRROR(sE + c) if (mS) SNAP_MIRROR(sE + cS) if (mN) SNAP_MIRROR(sE + cN) }
#undef SNAP_MIRROR
    }
  }
  __syncthreads();
  // ---- phase 3: every wave stores its rows (+ the physical-boundary mirrors, mg_mpi_exchange.f90:509-537,552-597)
  if (!live) return;
  const int j = jodd ? 2 * jh + 1 : 2 * jh + 2;
  const bool mS = ph.S && j == 1, mN = ph.N && j == L.ny, mW = ph.W && i == 1, mE = ph.E && i == L.nx;
  const int cS = L.EO, cN = jpos(L, L.ny + 1);
  const long long oW = 0, oE = (long long)(L.nx + 1) * L.plane;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int k = ka + r;
    const long long ro = (long long)(k - 1) * RS;
    const double v = sh[(k - 1) * WAVE + lane];
    p[o + ro + c] = v;
    if (mS) p[o + ro + cS] = v;
    if (mN) p[o + ro + cN] = v;
    if (mW) { p[oW + ro + c] = v; if (mS) p[oW + ro + cS] = v; if (mN) p[oW + ro + cN] = v; }
    if (mE) { p[oE + ro + c] = v; if (mS) p[oE + ro + cS] = v; if (mN) p[oE + ro + cN] = v; }
  }
}

extern "C" {

// returns 1 when the pass was launched here (the level qualifies), 0 to let the row-by-row kernels take it
int mgxk_relax_ks(hipStream_t st, const LevView *L, int i0, int istep, int nplanes, int jodd_fixed, int rb, int real, int snap, Sides ph) {
  static const bool off = getenv("MGX_NO_KS") != nullptr, noxcd = getenv("MGX_NO_XCD") != nullptr;
  static const int nw_env = getenv("MGX_KS_NW") ? atoi(getenv("MGX_KS_NW")) : 0;
  static const bool ks8 = getenv("MGX_NO_KS8") == nullptr;
  if (off || L->zy == nullptr || (L->nz != 32 && L->nz != 16 && !(L->nz == 8 && ks8))) return 0;
  const int gx0 = (L->ny / 2 + WAVE - 1) / WAVE;
  // worth it only while a colour has fewer waves than the chip has SIMDs (1024); a bandwidth-bound level keeps one wave per column set
  if (gx0 * nplanes > 512) return 0;
  // the level must live in the caches (no streaming hints here)
  const int gx = noxcd ? -gx0 : gx0;
  dim3 grd(gx0 * nplanes);
#define KS_LAUNCH(NZV, NWV)                                                                                                     \
  {                                                                                                                              \
    dim3 blk(WAVE, NWV);                                                                                                         \
    if (real && snap) hipLaunchKernelGGL((k_relax_ks<NZV, NWV, true, true>), grd, blk, 0, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx);   \
    else if (real) hipLaunchKernelGGL((k_relax_ks<NZV, NWV, true, false>), grd, blk, 0, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx);     \
    else hipLaunchKernelGGL((k_relax_ks<NZV, NWV, false, false>), grd, blk, 0, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx);              \
    return 1;                                                                                                                    \
  }
  // measured (256x256x32 / 128x128x16, four-colour sweep): 8 waves 47.0 / 21.6 us, 4 waves 53.3 / 23.3, row by row 57.6 / 28.2
  if (L->nz == 32) { if (nw_env == 4) KS_LAUNCH(32, 4) else KS_LAUNCH(32, 8) }
  if (L->nz == 16) { if (nw_env == 4) KS_LAUNCH(16, 4) else KS_LAUNCH(16, 8) }
  if (nw_env == 4) KS_LAUNCH(8, 4) else KS_LAUNCH(8, 8)
#undef KS_LAUNCH
}

}  // extern "C"
