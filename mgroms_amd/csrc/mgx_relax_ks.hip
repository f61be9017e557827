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

// H > 1 (nz = 64, the second level of an nz = 128 hierarchy): a wave's NZ/NW rows are built in H runs of R rows one after the other (the register
// arrays of a run are those of the nz = 32 kernel), the recurrence wave keeps only x(k) in registers and takes a2(k), bet(k) from LDS as it
// goes, and the three parked arrays are dynamic LDS (96 KB).
#ifdef MGX_KS_STAMP  // diagnostic build only (scripts/probe/ks_timeline.py): in-kernel time stamps of the phases of a pass
__device__ long long g_ks_stamp[1024 * 8];
#define KS_STAMP(q) { if (threadIdx.x == 0 && (w == 0 || w == NW - 1) && blockIdx.x < 512) g_ks_stamp[(blockIdx.x * 2 + (w ? 1 : 0)) * 8 + (q)] = (q) == 7 ? (long long)wall_clock64() : (long long)__builtin_amdgcn_s_memtime(); }
#else
#define KS_STAMP(q)
#endif
template <int NZ, int NW, bool REAL, bool SNAP, int H = 1>
__global__ __launch_bounds__(64 * NW, 1) void k_relax_ks(LevView L, int i0, int istep, int nplanes, int jodd_fixed, int rb, Sides ph, int gx) {
  constexpr int R = NZ / NW / H;  // rows per wave and run
  extern __shared__ double ks_lds[];
  double *__restrict__ sh = ks_lds, *__restrict__ sa2 = ks_lds + NZ * WAVE, *__restrict__ sbt = ks_lds + 2 * NZ * WAVE;  // rhs(k) then x(k); a2(k); bet(k): [k-1][lane]
  // XCD-aware block -> (j-chunk, plane) map, as k_relax_nz: each XCD owns a contiguous range of planes (speed only)
  int bx, ipl;
  if (gx < 0) { gx = -gx; ipl = blockIdx.x / gx; bx = blockIdx.x - ipl * gx; }
  else if ((nplanes & 7) == 0) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    ipl = xcd * (nplanes >> 3) + local / gx;
    bx = local - (local / gx) * gx;
  } else { ipl = blockIdx.x / gx; bx = blockIdx.x - ipl * gx; }
  const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y);  // wave-uniform: row ranges in scalar registers
  KS_STAMP(0) KS_STAMP(7)
  const int jh = bx * WAVE + lane;
  const bool live = jh < (L.ny >> 1);  // ragged last chunk: dead lanes still join the barriers
  const int i = i0 + istep * ipl;
  // RB: j = 1+mod(i+rb,2),ny,2 (mg_relax.f90:174) ; FC: fixed parity (:216-217)
  const int jodd = jodd_fixed >= 0 ? jodd_fixed : (((i + rb) & 1) == 0);
  if (sides_part_skip(ph, i, L.nx, jodd, bx, gx)) return;  // workgroup-uniform (before any barrier)
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
  const int kw = w * R * H + 1;  // first row of this wave

  // ---- phase 1: right-hand sides of rows ka .. ka+R-1 (every load of the wave is issued before the first use)
  double pjm[R + 2], pim[R + 2], pjp[R + 2], pip[R + 2], zyo[R + 2], zxo[R + 2];  // rows ka-1 .. ka+R
  double bb[R], a4o[R], a7o[R], a4n[R], a7n[R], zyjm[R], zyjp[R], zxim[R], zxip[R];
  double oa2[R], obt[R];  // the recurrence's coefficients of the wave's rows: handed to wave 0 through LDS
  double d1 = 0, d2 = 0, d3 = 0, d4 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
#pragma unroll 1
  for (int run = 0; run < H; run++) {
  const int ka = kw + run * R;  // first row of this run
  if (live) {
#pragma unroll
    for (int r = 0; r < R + 2; r++) {
      // rows 0 and NZ+1 do not exist and are never used (the first and the last row have their own expressions): clamped, no branch
      const int k = ka - 1 + r < 1 ? 1 : (ka - 1 + r > NZ ? NZ : ka - 1 + r);
      const long long ro = (long long)(k - 1) * RS;
      pjm[r] = p[o + ro + jm]; pim[r] = p[om + ro + c]; pjp[r] = p[o + ro + jp]; pip[r] = p[op + ro + c];
      zyo[r] = zy[o + ro + c]; zxo[r] = zx[o + ro + c];
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const long long ro = (long long)(ka + r - 1) * RS;
      bb[r] = b[o + ro + c]; a4o[r] = a4[o + ro + c]; a7o[r] = a7[o + ro + c];
      a4n[r] = a4[o + ro + jp]; a7n[r] = a7[op + ro + c];
      zyjm[r] = zy[o + ro + jm]; zyjp[r] = zy[o + ro + jp]; zxim[r] = zx[om + ro + c]; zxip[r] = zx[op + ro + c];
      oa2[r] = a2[o + ro + c]; obt[r] = bet[o + ro + c];
    }
    if (w == 0 && run == 0) {
      if (REAL) {
        const double *__restrict__ q1 = SNAP ? L.p1 : p;
        const long long s = SNAP ? (long long)i * RS : o, sm = SNAP ? s - RS : om, sp = SNAP ? s + RS : op;
        d1 = q1[sm + jp]; d2 = q1[sp + jm]; d3 = q1[sm + jm]; d4 = q1[sp + jp];
        e1 = a5[o + c]; e2 = a5[op + jm]; e3 = a8[o + c]; e4 = a8[op + jp];
      }
    }
#ifdef MGX_KS_STAMP
    KS_STAMP(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KS_STAMP(2)
#endif
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
  }
  __syncthreads();
  KS_STAMP(3)
  // ---- phase 2 (wave 0): tridiag (mg_relax.f90:322-332) on the parked right-hand sides
  if (H > 1) {
    if (w == 0 && live) {  // x(k) in registers only; a2(k), bet(k) stream from LDS (their addresses do not depend on the recurrence)
      double x[NZ];
      double xv = sh[lane] * sbt[lane];
      x[0] = xv;
#pragma unroll
      for (int k = 2; k <= NZ; k++) {
        xv = (sh[(k - 1) * WAVE + lane] - sa2[(k - 1) * WAVE + lane] * xv) * sbt[(k - 1) * WAVE + lane];
        x[k - 1] = xv;
      }
#pragma unroll
      for (int k = NZ - 1; k >= 1; k--) x[k - 1] = x[k - 1] - (sa2[k * WAVE + lane] * sbt[(k - 1) * WAVE + lane]) * x[k];  // gam(k+1) = dd(k)*bet(k)
#pragma unroll
      for (int k = 1; k <= NZ; k++) sh[(k - 1) * WAVE + lane] = x[k - 1];
    }
  } else
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
    if (SNAP && L.d0w != nullptr) L.d0w[(long long)i * RS + c] = x[0] - L.p1[(long long)i * RS + c];  // mgx_rbseq.hip (b): the walk's d0
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
  KS_STAMP(4)
  // ---- phase 3: every wave stores its rows (+ the physical-boundary mirrors, mg_mpi_exchange.f90:509-537,552-597)
  if (!live) return;
  const int j = jodd ? 2 * jh + 1 : 2 * jh + 2;
  const bool mS = ph.S && j == 1, mN = ph.N && j == L.ny, mW = ph.W && i == 1, mE = ph.E && i == L.nx;
  const int cS = L.EO, cN = jpos(L, L.ny + 1);
  const long long oW = 0, oE = (long long)(L.nx + 1) * L.plane;
#pragma unroll
  for (int r = 0; r < R * H; r++) {
    const int k = kw + r;
    const long long ro = (long long)(k - 1) * RS;
    const double v = sh[(k - 1) * WAVE + lane];
    p[o + ro + c] = v;
    if (mS) p[o + ro + cS] = v;
    if (mN) p[o + ro + cN] = v;
    if (mW) { p[oW + ro + c] = v; if (mS) p[oW + ro + cS] = v; if (mN) p[oW + ro + cN] = v; }
    if (mE) { p[oE + ro + c] = v; if (mS) p[oE + ro + cS] = v; if (mN) p[oE + ro + cN] = v; }
  }
#ifdef MGX_KS_STAMP
  KS_STAMP(5)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  KS_STAMP(6)
#endif
}
#ifdef MGX_KS_STAMP
extern "C" int mgxk_ks_stamps(long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ks_stamp), sizeof(long long) * 1024 * 8) == hipSuccess ? 0 : 1; }
#endif

// ------------------------------------------------------------------------------------------------
// Two colours of the four-colour sweep in one launch.  Colours 1 and 2 (mg_relax.f90:214-217: i odd with j odd, then i odd with
// j even) touch the same planes, and the second reads nothing the first changes except its j-1 / j+1 neighbours in the SAME plane
// (the planes i-1, i+1 belong to colours 3 and 4) -- and, on a physical west / east boundary, the k=1 diagonals in the mirrored
// halo plane.  So one workgroup that owns a whole plane (ny/2 <= 64 columns per colour: the 128x128x16 and 64x64x8 levels) runs
// colour a, keeps its result in LDS, and runs colour b from there; everything else colour b needs is requested at the start and
// arrives while colour a is computed.  The mid levels are launch- and latency-bound (6 and 4.8 us per pass for 4096 and 1024
// columns): 10 launches per V-cycle-level instead of 20.  Closed levels only (no exchange between the two colours).
// Same expressions in the same order: bit-identical to the two separate passes.
// ------------------------------------------------------------------------------------------------
namespace {
template <int R>
struct KsRows {
  double pim[R + 2], pip[R + 2], zyo[R + 2], zxo[R + 2];  // rows ka-1 .. ka+R
  double bb[R], a4o[R], a7o[R], a4n[R], a7n[R], zyjm[R], zyjp[R], zxim[R], zxip[R], oa2[R], obt[R];
  double e1, e2, e3, e4;
};
// everything of rows ka .. ka+R-1 that colour a does not change (the j-1 / j+1 neighbours of p and the k=1 diagonals are the caller's)
template <int NZ, int R, bool REAL>
__device__ __forceinline__ void ks_load(KsRows<R> &q, const LevView &L, const long long o, const int c, const int jm, const int jp, const int ka, const bool diag) {
  const long long RS = L.RS, om = o - L.plane, op = o + L.plane;
  const double *__restrict__ p = L.p, *__restrict__ b = L.b;
  const double *__restrict__ a2 = L.cA[1], *__restrict__ a4 = L.cA[3], *__restrict__ a5 = L.cA[4], *__restrict__ a7 = L.cA[6],
               *__restrict__ a8 = L.cA[7], *__restrict__ bet = L.bet, *__restrict__ zy = L.zy, *__restrict__ zx = L.zx;
#pragma unroll
  for (int r = 0; r < R + 2; r++) {
    // rows 0 and NZ+1 do not exist and are never used (the first and the last row have their own expressions): clamped, no branch
    const int k = ka - 1 + r < 1 ? 1 : (ka - 1 + r > NZ ? NZ : ka - 1 + r);
    const long long ro = (long long)(k - 1) * RS;
    q.pim[r] = p[om + ro + c]; q.pip[r] = p[op + ro + c];
    q.zyo[r] = zy[o + ro + c]; q.zxo[r] = zx[o + ro + c];
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    const long long ro = (long long)(ka + r - 1) * RS;
    q.bb[r] = b[o + ro + c]; q.a4o[r] = a4[o + ro + c]; q.a7o[r] = a7[o + ro + c];
    q.a4n[r] = a4[o + ro + jp]; q.a7n[r] = a7[op + ro + c];
    q.zyjm[r] = zy[o + ro + jm]; q.zyjp[r] = zy[o + ro + jp]; q.zxim[r] = zx[om + ro + c]; q.zxip[r] = zx[op + ro + c];
    q.oa2[r] = a2[o + ro + c]; q.obt[r] = bet[o + ro + c];
  }
  q.e1 = q.e2 = q.e3 = q.e4 = 0.0;
  if (REAL && diag) {  // the wave that owns row 1: coefficients of the four k=1 horizontal diagonals (mg_relax.f90:271-276)
    q.e1 = a5[o + c]; q.e2 = a5[op + jm]; q.e3 = a8[o + c]; q.e4 = a8[op + jp];
  }
}
template <int NZ, int R, bool REAL>
__device__ __forceinline__ void ks_rhs(const KsRows<R> &q, const double *pjm, const double *pjp, const double d1, const double d2, const double d3, const double d4,
                                       const int ka, const int lane, double *__restrict__ sh, double *__restrict__ sa2, double *__restrict__ sbt) {
  const double qrt = 0.25;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int k = ka + r;
    const double pjm_m = pjm[r], pjm_0 = pjm[r + 1], pjm_p = pjm[r + 2], pim_m = q.pim[r], pim_0 = q.pim[r + 1], pim_p = q.pim[r + 2];
    const double pjp_m = pjp[r], pjp_0 = pjp[r + 1], pjp_p = pjp[r + 2], pip_m = q.pip[r], pip_0 = q.pip[r + 1], pip_p = q.pip[r + 2];
    const double zy_m = q.zyo[r], zy_p = q.zyo[r + 2], zx_m = q.zxo[r], zx_p = q.zxo[r + 2];
    const double c3 = qrt * (zy_p + q.zyjm[r]), c3m = qrt * (q.zyjp[r] + zy_m), c5 = -qrt * (zy_m + q.zyjm[r]), c5m = -qrt * (q.zyjp[r] + zy_p);
    const double c6 = qrt * (zx_p + q.zxim[r]), c6m = qrt * (q.zxip[r] + zx_m), c8 = -qrt * (zx_m + q.zxim[r]), c8m = -qrt * (q.zxip[r] + zx_p);
    double rhs;
    if (k == 1) {
      rhs = q.bb[r] - c3 * pjm_p - q.a4o[r] * pjm_0 - q.a4n[r] * pjp_0 - c5m * pjp_p
                    - c6 * pim_p - q.a7o[r] * pim_0 - q.a7n[r] * pip_0 - c8m * pip_p;
      if (REAL) rhs = rhs - q.e1 * d1 - q.e2 * d2 - q.e3 * d3 - q.e4 * d4;
    } else if (k < NZ) {
      rhs = q.bb[r] - c3 * pjm_p - c3m * pjp_m - q.a4o[r] * pjm_0 - q.a4n[r] * pjp_0
                    - c5 * pjm_m - c5m * pjp_p
                    - c6 * pim_p - c6m * pip_m - q.a7o[r] * pim_0 - q.a7n[r] * pip_0
                    - c8 * pim_m - c8m * pip_p;
    } else {
      rhs = q.bb[r] - c3m * pjp_m - q.a4o[r] * pjm_0 - q.a4n[r] * pjp_0 - c5 * pjm_m
                    - c6m * pip_m - q.a7o[r] * pim_0 - q.a7n[r] * pip_0 - c8 * pim_m;
    }
    sh[(k - 1) * WAVE + lane] = rhs;
    sa2[(k - 1) * WAVE + lane] = q.oa2[r];
    sbt[(k - 1) * WAVE + lane] = q.obt[r];
  }
}
// tridiag (mg_relax.f90:322-332) on the parked right-hand sides of one lane; x(k) goes to out[(k-1)*stride + lane]
template <int NZ>
__device__ __forceinline__ void ks_tridiag(const double *sh, const double *__restrict__ sa2, const double *__restrict__ sbt, const int lane,
                                           double *out, const int stride) {  // out may be sh itself
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
  for (int k = 1; k <= NZ; k++) out[(k - 1) * stride + lane] = x[k - 1];
}
// the same with a2(k), bet(k) taken from LDS as the recurrence goes (their addresses do not depend on it): only x(k) in registers
template <int NZ>
__device__ __forceinline__ void ks_tridiag_lds(const double *sh, const double *__restrict__ sa2, const double *__restrict__ sbt, const int lane,
                                               double *out, const int stride) {  // out may be sh itself
  double x[NZ];
  double xv = sh[lane] * sbt[lane];
  x[0] = xv;
#pragma unroll
  for (int k = 2; k <= NZ; k++) {
    xv = (sh[(k - 1) * WAVE + lane] - sa2[(k - 1) * WAVE + lane] * xv) * sbt[(k - 1) * WAVE + lane];
    x[k - 1] = xv;
  }
#pragma unroll
  for (int k = NZ - 1; k >= 1; k--) x[k - 1] = x[k - 1] - (sa2[k * WAVE + lane] * sbt[(k - 1) * WAVE + lane]) * x[k];  // gam(k+1) = dd(k)*bet(k)
#pragma unroll
  for (int k = 1; k <= NZ; k++) out[(k - 1) * stride + lane] = x[k - 1];
}
// store rows ka .. ka+R-1 of a column and the physical-boundary mirrors (mg_mpi_exchange.f90:509-537,552-597)
// WT: write-through stores (agent scope, `sc1`): the persistent kernel hands planes to other workgroups inside one launch
template <bool WT> __device__ __forceinline__ void ks_st(double *q, const double v) {
  if (WT) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *q = v;
}
template <bool WT> __device__ __forceinline__ double ks_ld(const double *q) {
  return WT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
}
template <int R, bool WT = false>
__device__ __forceinline__ void ks_store(const LevView &L, const long long o, const int c, const int i, const int j, const int ka, const int lane,
                                         const double *__restrict__ x, const int stride, const Sides ph) {
  double *__restrict__ p = L.p;
  const bool mS = ph.S && j == 1, mN = ph.N && j == L.ny, mW = ph.W && i == 1, mE = ph.E && i == L.nx;
  const int cS = L.EO, cN = jpos(L, L.ny + 1);
  const long long oW = 0, oE = (long long)(L.nx + 1) * L.plane;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int k = ka + r;
    const long long ro = (long long)(k - 1) * L.RS;
    const double v = x[(k - 1) * stride + lane];
    ks_st<WT>(p + o + ro + c, v);
    if (mS) ks_st<WT>(p + o + ro + cS, v);
    if (mN) ks_st<WT>(p + o + ro + cN, v);
    if (mW) { ks_st<WT>(p + oW + ro + c, v); if (mS) ks_st<WT>(p + oW + ro + cS, v); if (mN) ks_st<WT>(p + oW + ro + cN, v); }
    if (mE) { ks_st<WT>(p + oE + ro + c, v); if (mS) ks_st<WT>(p + oE + ro + cS, v); if (mN) ks_st<WT>(p + oE + ro + cN, v); }
  }
}
// the two halves of ks_load for the persistent kernel: what no sweep changes (once per launch) ...
template <int NZ, int R, bool REAL>
__device__ __forceinline__ void ks_load_coef(KsRows<R> &q, const LevView &L, const long long o, const int c, const int jm, const int jp, const int ka, const bool diag) {
  const long long RS = L.RS, om = o - L.plane, op = o + L.plane;
  const double *__restrict__ b = L.b;
  const double *__restrict__ a2 = L.cA[1], *__restrict__ a4 = L.cA[3], *__restrict__ a5 = L.cA[4], *__restrict__ a7 = L.cA[6],
               *__restrict__ a8 = L.cA[7], *__restrict__ bet = L.bet, *__restrict__ zy = L.zy, *__restrict__ zx = L.zx;
#pragma unroll
  for (int r = 0; r < R + 2; r++) {
    const int k = ka - 1 + r < 1 ? 1 : (ka - 1 + r > NZ ? NZ : ka - 1 + r);
    const long long ro = (long long)(k - 1) * RS;
    q.zyo[r] = zy[o + ro + c]; q.zxo[r] = zx[o + ro + c];
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    const long long ro = (long long)(ka + r - 1) * RS;
    q.bb[r] = b[o + ro + c]; q.a4o[r] = a4[o + ro + c]; q.a7o[r] = a7[o + ro + c];
    q.a4n[r] = a4[o + ro + jp]; q.a7n[r] = a7[op + ro + c];
    q.zyjm[r] = zy[o + ro + jm]; q.zyjp[r] = zy[o + ro + jp]; q.zxim[r] = zx[om + ro + c]; q.zxip[r] = zx[op + ro + c];
    q.oa2[r] = a2[o + ro + c]; q.obt[r] = bet[o + ro + c];
  }
  q.e1 = q.e2 = q.e3 = q.e4 = 0.0;
  if (REAL && diag) { q.e1 = a5[o + c]; q.e2 = a5[op + jm]; q.e3 = a8[o + c]; q.e4 = a8[op + jp]; }
}
// ... and the i-1 / i+1 neighbours of p, which the neighbouring planes' workgroups rewrite every sweep
template <int NZ, int R, bool WT>
__device__ __forceinline__ void ks_load_p(KsRows<R> &q, const LevView &L, const long long o, const int c, const int ka) {
  const long long RS = L.RS, om = o - L.plane, op = o + L.plane;
  const double *p = L.p;
#pragma unroll
  for (int r = 0; r < R + 2; r++) {
    const int k = ka - 1 + r < 1 ? 1 : (ka - 1 + r > NZ ? NZ : ka - 1 + r);
    const long long ro = (long long)(k - 1) * RS;
    q.pim[r] = ks_ld<WT>(p + om + ro + c); q.pip[r] = ks_ld<WT>(p + op + ro + c);
  }
}
}  // namespace

template <int NZ, int NW, bool REAL>
__global__ __launch_bounds__(64 * NW, 1) void k_relax_ks2(LevView L, int i0, int nplanes, Sides ph) {
  constexpr int R = NZ / NW;     // rows per wave
  constexpr int XS = WAVE + 1;   // colour a's result: [k-1][lane], slot nh = the halo column ny+1 behind the last column
  __shared__ double sh[NZ * WAVE], sa2[NZ * WAVE], sbt[NZ * WAVE], xa[NZ * XS];
  int ipl = blockIdx.x;
  if ((nplanes & 7) == 0) { const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3; ipl = xcd * (nplanes >> 3) + local; }  // contiguous planes per XCD
  const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y);  // wave-uniform: row ranges in scalar registers
  const int nh = L.ny >> 1;      // columns of one colour in a plane (<= 64)
  const bool live = lane < nh;
  const int i = i0 + 2 * ipl;
  const long long o = (long long)i * L.plane;
  const int ka = w * R + 1;
  // colour a: j = 2 lane + 1 ; colour b: j = 2 lane + 2, whose j-1 / j+1 neighbours are colour a's columns lane and lane + 1
  const int cA = L.HO + lane, jmA = L.EO + lane, jpA = jmA + 1;
  const int cB = L.EO + lane + 1, jmB = L.HO + lane, jpB = jmB + 1;
  const bool mW = ph.W && i == 1, mE = ph.E && i == L.nx;
  KsRows<R> A, B;
  double pjm[R + 2] = {}, pjp[R + 2] = {}, d1 = 0, d2 = 0, d3 = 0, d4 = 0;     // colour a: from memory
  double bd1 = 0, bd2 = 0, bd3 = 0, bd4 = 0;                          // colour b's k=1 diagonals when they sit in unchanged planes
  const long long om = o - L.plane, op = o + L.plane;
  if (live) {
#pragma unroll
    for (int r = 0; r < R + 2; r++) {
      const int k = ka - 1 + r < 1 ? 1 : (ka - 1 + r > NZ ? NZ : ka - 1 + r);
      pjm[r] = L.p[o + (long long)(k - 1) * L.RS + jmA]; pjp[r] = L.p[o + (long long)(k - 1) * L.RS + jpA];
    }
    if (REAL && w == 0) {
      d1 = L.p[om + jpA]; d2 = L.p[op + jmA]; d3 = L.p[om + jmA]; d4 = L.p[op + jpA];
      bd1 = L.p[om + jpB]; bd2 = L.p[op + jmB]; bd3 = L.p[om + jmB]; bd4 = L.p[op + jpB];
    }
    ks_load<NZ, R, REAL>(A, L, o, cA, jmA, jpA, ka, w == 0);
    ks_load<NZ, R, REAL>(B, L, o, cB, jmB, jpB, ka, w == 0);
  }
  if (lane == 0) {  // the halo column ny+1 of this plane (image of column ny: colour b's own, not touched by colour a)
#pragma unroll
    for (int r = 0; r < R; r++) xa[(ka + r - 1) * XS + nh] = L.p[o + (long long)(ka + r - 1) * L.RS + L.HO + nh];
  }
  if (live) ks_rhs<NZ, R, REAL>(A, pjm, pjp, d1, d2, d3, d4, ka, lane, sh, sa2, sbt);
  __syncthreads();
  if (w == 0 && live) ks_tridiag<NZ>(sh, sa2, sbt, lane, xa, XS);
  __syncthreads();
  if (live) {
    ks_store<R>(L, o, cA, i, 2 * lane + 1, ka, lane, xa, XS, ph);
    double qjm[R + 2], qjp[R + 2];
#pragma unroll
    for (int r = 0; r < R + 2; r++) {
      const int k = ka - 1 + r < 1 ? 1 : (ka - 1 + r > NZ ? NZ : ka - 1 + r);
      qjm[r] = xa[(k - 1) * XS + lane]; qjp[r] = xa[(k - 1) * XS + lane + 1];
    }
    if (REAL && w == 0) {
      // k=1 diagonals in a mirrored halo plane are images of colour a's new values (plane 0 of plane 1, plane nx+1 of plane nx)
      if (mW) { bd1 = xa[lane + 1]; bd3 = xa[lane]; }
      if (mE) { bd2 = xa[lane]; bd4 = xa[lane + 1]; }
    }
    ks_rhs<NZ, R, REAL>(B, qjm, qjp, bd1, bd2, bd3, bd4, ka, lane, sh, sa2, sbt);
  }
  __syncthreads();
  if (w == 0 && live) ks_tridiag<NZ>(sh, sa2, sbt, lane, sh, WAVE);
  __syncthreads();
  if (live) ks_store<R>(L, o, cB, i, 2 * lane + 2, ka, lane, sh, WAVE, ph);
}

// ------------------------------------------------------------------------------------------------
// A whole relax call -- nsweeps four-colour sweeps (mg_relax.f90:193-234) -- of a closed mid level in ONE launch.
//
// k_relax_ks2 above needs two launches per sweep (the odd planes' colours 1+2, then the even planes' colours 3+4): 10 dependent
// launches per level visit at 9.2 us (128x128x16) / 5.9 us (64x64x8) each, of which the arithmetic is a small part -- the rest is the
// kernel boundary, the ramp of a new grid and the re-load of coefficients that no sweep changes.  Here ONE workgroup per plane (odd and even
// planes alike) stays resident for the whole call: it loads its columns' coefficients once into registers, then alternates with its
// two neighbour planes through a per-plane progress counter in global memory:
//     odd  plane i, sweep s : waits until planes i-1, i+1 have finished s phases, runs colours 1+2, publishes s+1
//     even plane i, sweep s : waits until planes i-1, i+1 have finished s+1 phases, runs colours 3+4, publishes s+1
// A plane's only inter-workgroup dependencies are its two neighbour planes (p at i-1 / i+1 incl. the k=1 diagonals); the waits
// above order every read after the write it needs and every write after the last read of the value it replaces.
// Hand-off (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").  FENCE = true (default): plain
// stores, each storing wave drains them (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one lane issues an agent-scope release
// fence and stores the counter; the consumer polls it with one lane (relaxed agent-scope loads, s_sleep between polls), issues an
// agent-scope acquire fence and releases its workgroup through a barrier, after which plain loads see the neighbour planes.
// FENCE = false (MGX_KSP_SC1=1, A/B): no fences, every load and store of p is an agent-scope (sc1) access instead -- write-through
// stores, L1-bypassing loads -- which the guide measured as sufficient on gfx950 but does not call an architectural guarantee.  All nx workgroups must be resident together (<= 128 of
// them, one per CU: 512 threads at up to 256 registers); the poll is bounded (2 s of the constant clock -> error word, reported by
// the next synchronising call) so that a co-tenant that keeps a workgroup off the chip cannot hang the stream.
// Same expressions in the same order as k_relax_ks2: bit-identical.
// ------------------------------------------------------------------------------------------------
// bound of a plane's wait for its neighbours, ticks of the 100 MHz constant clock (2 s; mgxk_set_ksp_timeout shortens it for the test)
__device__ long long g_ksp_timeout_ticks = 200000000LL;
template <int NZ, int NW, bool REAL, bool FENCE>
__global__ __launch_bounds__(64 * NW, 1) void k_relax_ksp(LevView L, int nsweeps, Sides ph, unsigned int *done, unsigned int base, int *err, int stall) {
  constexpr int R = NZ / NW;     // rows per wave
  constexpr int XS = WAVE + 1;
  constexpr bool WT = !FENCE;
  __shared__ double sh[NZ * WAVE], sa2[NZ * WAVE], sbt[NZ * WAVE], xa[NZ * XS];
  __shared__ int s_bail;
  int i = blockIdx.x + 1;
  if ((L.nx & 7) == 0) { const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3; i = xcd * (L.nx >> 3) + local + 1; }  // neighbour planes share an XCD (speed only)
  if (i == stall) return;  // test hook: this plane's workgroup never shows up (as if a co-tenant kept it off the chip)
  const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int nh = L.ny >> 1;
  const bool live = lane < nh, odd = (i & 1) != 0;
  const long long o = (long long)i * L.plane, om = o - L.plane, op = o + L.plane;
  const int ka = w * R + 1;
  const int cA = L.HO + lane, jmA = L.EO + lane, jpA = jmA + 1;
  const int cB = L.EO + lane + 1, jmB = L.HO + lane, jpB = jmB + 1;
  const bool mW = ph.W && i == 1, mE = ph.E && i == L.nx;
  KsRows<R> A, B;
  if (live) {
    ks_load_coef<NZ, R, REAL>(A, L, o, cA, jmA, jpA, ka, w == 0);
    ks_load_coef<NZ, R, REAL>(B, L, o, cB, jmB, jpB, ka, w == 0);
  }
  if (threadIdx.x == 0 && threadIdx.y == 0) s_bail = 0;
  __syncthreads();
#ifdef MGX_KS_STAMP
#define KSP_STAMP(q) { if (threadIdx.x == 0 && threadIdx.y == 0 && blockIdx.x < 128 && s == 1) g_ks_stamp[blockIdx.x * 8 + (q)] = (long long)wall_clock64(); }
#else
#define KSP_STAMP(q)
#endif
  for (int s = 0; s < nsweeps; s++) {
    KSP_STAMP(0)
    // the addresses of a phase do not change from sweep to sweep: left alone, the compiler keeps all ~60 of them in registers across
    // the loop (and spills the coefficients instead); an opaque copy of the base pointer makes it rebuild them, a few scalar adds
    LevView Lc = L;
    { double *pl = L.p; asm volatile("" : "+s"(pl)); Lc.p = pl; }
    const double *pr = Lc.p;
    if (!(odd && s == 0)) {  // the first phase of an odd plane reads what the previous launch left
      if (threadIdx.x == 0 && threadIdx.y == 0) {
        const unsigned int need = base + (unsigned int)s + (odd ? 0u : 1u);
        const long long t0 = wall_clock64(), tmax = g_ksp_timeout_ticks;
        for (;;) {
          bool ok = true;
          if (i > 1) ok = ok && (int)(__hip_atomic_load(done + i - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) >= 0;
          if (i < L.nx) ok = ok && (int)(__hip_atomic_load(done + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) >= 0;
          if (ok) break;
          if (wall_clock64() - t0 > tmax) { *err = 1; s_bail = 1; break; }  // 2 s at 100 MHz
          __builtin_amdgcn_s_sleep(1);
        }
        if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      }
      __syncthreads();
      if (s_bail) return;
    }
    KSP_STAMP(1)
    double pjm[R + 2] = {}, pjp[R + 2] = {}, d1 = 0, d2 = 0, d3 = 0, d4 = 0, bd1 = 0, bd2 = 0, bd3 = 0, bd4 = 0;
    if (live) {
#pragma unroll
      for (int r = 0; r < R + 2; r++) {
        const int k = ka - 1 + r < 1 ? 1 : (ka - 1 + r > NZ ? NZ : ka - 1 + r);
        pjm[r] = ks_ld<WT>(pr + o + (long long)(k - 1) * L.RS + jmA); pjp[r] = ks_ld<WT>(pr + o + (long long)(k - 1) * L.RS + jpA);
      }
      if (REAL && w == 0) {
        d1 = ks_ld<WT>(pr + om + jpA); d2 = ks_ld<WT>(pr + op + jmA); d3 = ks_ld<WT>(pr + om + jmA); d4 = ks_ld<WT>(pr + op + jpA);
        bd1 = ks_ld<WT>(pr + om + jpB); bd2 = ks_ld<WT>(pr + op + jmB); bd3 = ks_ld<WT>(pr + om + jmB); bd4 = ks_ld<WT>(pr + op + jpB);
      }
      ks_load_p<NZ, R, WT>(A, Lc, o, cA, ka);
      ks_load_p<NZ, R, WT>(B, Lc, o, cB, ka);
    }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; r++) xa[(ka + r - 1) * XS + nh] = ks_ld<WT>(pr + o + (long long)(ka + r - 1) * L.RS + L.HO + nh);
    }
    if (live) ks_rhs<NZ, R, REAL>(A, pjm, pjp, d1, d2, d3, d4, ka, lane, sh, sa2, sbt);
    __syncthreads();
    KSP_STAMP(2)
    if (w == 0 && live) ks_tridiag_lds<NZ>(sh, sa2, sbt, lane, xa, XS);
    __syncthreads();
    KSP_STAMP(3)
    if (live) {
      ks_store<R, WT>(Lc, o, cA, i, 2 * lane + 1, ka, lane, xa, XS, ph);
      double qjm[R + 2], qjp[R + 2];
#pragma unroll
      for (int r = 0; r < R + 2; r++) {
        const int k = ka - 1 + r < 1 ? 1 : (ka - 1 + r > NZ ? NZ : ka - 1 + r);
        qjm[r] = xa[(k - 1) * XS + lane]; qjp[r] = xa[(k - 1) * XS + lane + 1];
      }
      if (REAL && w == 0) {
        if (mW) { bd1 = xa[lane + 1]; bd3 = xa[lane]; }
        if (mE) { bd2 = xa[lane]; bd4 = xa[lane + 1]; }
      }
      ks_rhs<NZ, R, REAL>(B, qjm, qjp, bd1, bd2, bd3, bd4, ka, lane, sh, sa2, sbt);
    }
    __syncthreads();
    if (w == 0 && live) ks_tridiag_lds<NZ>(sh, sa2, sbt, lane, sh, WAVE);
    __syncthreads();
    KSP_STAMP(4)
    if (live) ks_store<R, WT>(Lc, o, cB, i, 2 * lane + 2, ka, lane, sh, WAVE, ph);
    // publish: every storing wave drains its stores, the workgroup meets, one lane raises the plane's counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    KSP_STAMP(5)
    if (threadIdx.x == 0 && threadIdx.y == 0) {
      if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __hip_atomic_store(done + i, base + (unsigned int)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    KSP_STAMP(6)
  }
}

extern "C" {

// returns 1 when the pass was launched here (the level qualifies), 0 to let the row-by-row kernels take it
int mgxk_relax_ks(hipStream_t st, const LevView *L, int i0, int istep, int nplanes, int jodd_fixed, int rb, int real, int snap, Sides ph) {
  mgx_before_launch();
  static const bool off = getenv("MGX_NO_KS") != nullptr, noxcd = getenv("MGX_NO_XCD") != nullptr;
  static const int nw_env = getenv("MGX_KS_NW") ? atoi(getenv("MGX_KS_NW")) : 0;
  static const bool ks8 = getenv("MGX_NO_KS8") == nullptr;
  static const bool ks64 = getenv("MGX_NO_KS64") == nullptr;
  if (off || L->zy == nullptr || (L->nz != 32 && L->nz != 16 && !(L->nz == 8 && ks8) && !(L->nz == 64 && ks64 && !snap))) return 0;
  const int gx0 = (L->ny / 2 + WAVE - 1) / WAVE;
  // worth it only while a colour has fewer waves than the chip has SIMDs (1024); a bandwidth-bound level keeps one wave per column set
  if (gx0 * nplanes > 512) return 0;
  // the level must live in the caches (no streaming hints here)
  const int gx = noxcd ? -gx0 : gx0;
  dim3 grd(gx0 * nplanes);
#define KS_LAUNCH(NZV, NWV)                                                                                                     \
  {                                                                                                                              \
    dim3 blk(WAVE, NWV);                                                                                                         \
    const size_t lds = (size_t)3 * NZV * WAVE * sizeof(double);                                                                  \
    if (real && snap) { hipLaunchKernelGGL((k_relax_ks<NZV, NWV, true, true>), grd, blk, lds, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx); return mgx_launched() ? (L->d0w != nullptr ? 3 : 1) : 0; } \
    else if (real) hipLaunchKernelGGL((k_relax_ks<NZV, NWV, true, false>), grd, blk, lds, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx);     \
    else hipLaunchKernelGGL((k_relax_ks<NZV, NWV, false, false>), grd, blk, lds, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx);              \
    return mgx_launched();                                                                                                       \
  }
  if (L->nz == 64) {  // 96 KB of dynamic LDS: above the 64 KB a kernel gets without asking
    static bool attr = false;
    const size_t lds = (size_t)3 * 64 * WAVE * sizeof(double);
    if (!attr) {
      if (hipFuncSetAttribute((const void *)k_relax_ks<64, 8, true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
          hipFuncSetAttribute((const void *)k_relax_ks<64, 8, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
      attr = true;
    }
    dim3 blk(WAVE, 8);
    if (real) hipLaunchKernelGGL((k_relax_ks<64, 8, true, false, 2>), grd, blk, lds, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx);
    else hipLaunchKernelGGL((k_relax_ks<64, 8, false, false, 2>), grd, blk, lds, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx);
    return mgx_launched();
  }
  // measured (256x256x32 / 128x128x16, four-colour sweep): 8 waves 47.0 / 21.6 us, 4 waves 53.3 / 23.3, row by row 57.6 / 28.2
  if (L->nz == 32) { if (nw_env == 4) KS_LAUNCH(32, 4) else KS_LAUNCH(32, 8) }
  if (L->nz == 16) { if (nw_env == 4) KS_LAUNCH(16, 4) else KS_LAUNCH(16, 8) }
  if (nw_env == 4) KS_LAUNCH(8, 4) else KS_LAUNCH(8, 8)
#undef KS_LAUNCH
}

// (mgxk_relax_ks returns 0 = not launched, 1 = launched, 3 = launched and L->d0w written)
// both colours of the planes i0, i0+2, ... of a four-colour sweep in one launch; returns 1 when launched (closed level, one column set per plane)
int mgxk_relax_ks_pair(hipStream_t st, const LevView *L, int i0, int nplanes, int real, Sides ph) {
  mgx_before_launch();
  static const bool off = getenv("MGX_NO_KS") != nullptr || getenv("MGX_NO_KS2") != nullptr;
  if (off || L->zy == nullptr || !(ph.S && ph.E && ph.N && ph.W) || (L->ny & 1) || L->ny / 2 > WAVE) return 0;
  if (L->nz != 16 && L->nz != 8) return 0;
  dim3 grd(nplanes), blk(WAVE, 8);
  if (L->nz == 16) { if (real) hipLaunchKernelGGL((k_relax_ks2<16, 8, true>), grd, blk, 0, st, *L, i0, nplanes, ph);
                     else hipLaunchKernelGGL((k_relax_ks2<16, 8, false>), grd, blk, 0, st, *L, i0, nplanes, ph); }
  else { if (real) hipLaunchKernelGGL((k_relax_ks2<8, 8, true>), grd, blk, 0, st, *L, i0, nplanes, ph);
         else hipLaunchKernelGGL((k_relax_ks2<8, 8, false>), grd, blk, 0, st, *L, i0, nplanes, ph); }
  return mgx_launched();
}

// all nsweeps four-colour sweeps of a closed mid level in one persistent launch (k_relax_ksp); returns 1 when launched.
// done: nx + 2 progress counters of the level (zero at init), base: their common value now; the caller adds nsweeps afterwards.
int mgxk_set_ksp_timeout(double ms) {
  const long long ticks = (long long)(ms * 1e5);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ksp_timeout_ticks), &ticks, sizeof ticks) == hipSuccess ? 0 : 1;
}
// stall: test hook, the plane whose workgroup returns at once (0 = none)
int mgxk_relax_ks_persist(hipStream_t st, const LevView *L, int nsweeps, int real, Sides ph, unsigned int *done, unsigned int base, int *err, int stall) {
  mgx_before_launch();
  static const bool off = getenv("MGX_NO_KS") != nullptr || getenv("MGX_NO_KS2") != nullptr || getenv("MGX_NO_KSP") != nullptr;
  // default: plain accesses between agent-scope release / acquire fences (the architecturally guaranteed hand-off); MGX_KSP_SC1=1: the
  // fence-free form with sc1 stores and loads (measured on gfx950 only) -- the two time the same (F-cycle 283-285 vs 285 it/s)
  static const bool fence = getenv("MGX_KSP_SC1") == nullptr;
  if (off || nsweeps < 1 || L->zy == nullptr || !(ph.S && ph.E && ph.N && ph.W) || (L->ny & 1) || (L->nx & 1) || L->ny / 2 > WAVE) return 0;
  if ((L->nz != 16 && L->nz != 8 && L->nz != 4) || L->nx > 128 || done == nullptr || err == nullptr) return 0;
  // nz = 16: four waves of four rows (both colours' coefficients of a lane: 288 registers, one wave per SIMD); nz = 8: eight waves of one row;
  // nz = 4: four waves of one row (the 128x64x4 level that eight GPUs gather; a single GPU's 32x32x4 level is k_relax_wave's)
  dim3 grd(L->nx), blk(WAVE, L->nz == 8 ? 8 : 4);
  // The workgroups wait for each other: ALL nx of them must be resident together.  The occupancy the runtime reports for this kernel times
  // the device's compute units must cover the grid (asked once per instance of the template), else the separate launches take the call.
  static int ncu = 0;
  if (ncu == 0) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount; else { (void)hipGetLastError(); ncu = -1; } }
#define KSP(NZV, RV, FV)                                                                                                          \
  {                                                                                                                                \
    static int cap = -1;                                                                                                           \
    if (cap < 0) {                                                                                                                 \
      int nb = 0;                                                                                                                  \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_relax_ksp<NZV, (NZV == 8 ? 8 : 4), RV, FV>, WAVE * (NZV == 8 ? 8 : 4), 0) != hipSuccess) { (void)hipGetLastError(); nb = 0; } \
      cap = ncu > 0 ? nb * ncu : 0;                                                                                                \
    }                                                                                                                              \
    if (cap < (int)grd.x) return 0;                                                                                                \
    hipLaunchKernelGGL((k_relax_ksp<NZV, (NZV == 8 ? 8 : 4), RV, FV>), grd, blk, 0, st, *L, nsweeps, ph, done, base, err, stall); \
  }
  if (L->nz == 16) { if (real) { if (fence) KSP(16, true, true) else KSP(16, true, false) } else { if (fence) KSP(16, false, true) else KSP(16, false, false) } }
  else if (L->nz == 8) { if (real) { if (fence) KSP(8, true, true) else KSP(8, true, false) } else { if (fence) KSP(8, false, true) else KSP(8, false, false) } }
  else { if (real) { if (fence) KSP(4, true, true) else KSP(4, true, false) } else { if (fence) KSP(4, false, true) else KSP(4, false, false) } }
#undef KSP
  return mgx_launched();
}

}  // extern "C"
