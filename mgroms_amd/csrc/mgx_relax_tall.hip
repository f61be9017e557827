// z-line smoother for tall columns (nz = 128, BASELINE config 5): one colour pass, matrix-free cross terms, the lower 64 rows'
// forward values in LDS.  mg_relax.f90:237-305 + :308-334.  Its own translation unit: x and gam of 64 rows stay in registers here
// (500 of 512), it needs a larger `#pragma unroll` budget than the others (Makefile), and the 16-byte pair loads of mgx_relax.hip
// do not fit next to them (0.8 KB/lane of scratch); slots 4 / 7 from regenerated zw do.
#include <cstdlib>

#include "mgx_device.h"

// Tall columns (nz = 128, BASELINE config 5): x and gam of 128 rows do not fit the register file next to the load rings.
// The forward-eliminated values of the lower LOW rows wait in LDS (xf: LOW rows x 64 lanes x 8 B = 32 KB per wave, one wave per
// SIMD = 128 KB of the CU's 160 KB) instead of going out to p and coming back; their gam is rebuilt on the way down from
// a2(k+1) and bet(k), re-read ahead of use (addresses are known: no dependent loads).  The upper NZ-LOW rows are handled exactly
// like relax_col_mf.  Same expressions, same order: bit-identical to the reference.
template <int NZ, int LOW, bool REAL, bool SNAP, int D, bool ST>
__device__ __forceinline__ void relax_col_mf_tall(const LevView &L, const int i, const int jh, const int jodd, const Sides ph, double *__restrict__ xf) {
  int c, jm, jp;
  if (jodd) { c = L.HO + jh; jm = L.EO + jh; jp = jm + 1; }
  else      { c = L.EO + jh + 1; jm = L.HO + jh; jp = jm + 1; }
  const long long RS = L.RS;
  double *__restrict__ p = L.p;
  const double *__restrict__ b = L.b;
  const double *__restrict__ a1 = L.cA[0], *__restrict__ a2 = L.cA[1], *__restrict__ a4 = L.cA[3], *__restrict__ a5 = L.cA[4],
               *__restrict__ a7 = L.cA[6], *__restrict__ a8 = L.cA[7], *__restrict__ bet = L.bet,
               *__restrict__ zy = L.zy, *__restrict__ zx = L.zx;
  const long long o = (long long)i * L.plane, om = o - L.plane, op = o + L.plane;
  const double qrt = 0.25;

  constexpr int RN = D + 2;  // rows k .. k+1+D are live at iteration k (row k is still read after the look-ahead load is issued)
  constexpr int RO = D + 2;  // own rows are needed one row early (zy(k+1), zx(k+1))
  double r_pjm[RN], r_pim[RN], r_pjp[RN], r_pip[RN], r_zyjm[RN], r_zyjp[RN], r_zxim[RN], r_zxip[RN], r_a4[RN], r_a7[RN];
  constexpr bool ZW = MGX_ZW;  // slots 4 and 7 of the interior rows from regenerated interface depths, see relax_col_mf (mgx_relax.hip)
  constexpr bool ZG = ZW && MGX_ZG;  // and the column's own slopes from regenerated zr of the face neighbours (same place)
  double o_b[RO], o_a2[RO], o_a4[RO], o_a7[RO], o_bet[RO], o_zy[ZG ? 1 : RO], o_zx[ZG ? 1 : RO];
  constexpr int UP = NZ - LOW;
  double x[UP], g[UP];
  double zw0[5], zw1[5], hh[5], hv[5], hz[5];
  const double *__restrict__ cffw = L.cffw, *__restrict__ csw = L.csw, *__restrict__ cffr = L.cffr, *__restrict__ csr = L.csr;
  double r4c = 0, r4p = 0, r7c = 0, r7p = 0, gdx = 1, gdy = 1, rdx = 0, rdy = 0;  // reciprocals of the constant divisors (DIVC, mgx_device.h)
  const double hlf = 0.5;

#define NB_LOAD(q)                                                               \
  if ((q) <= NZ) {                                                               \
    const long long ro_ = (long long)((q)-1) * RS; const int s_ = (q) % RN;      \
    r_pjm[s_] = p[o + ro_ + jm]; r_pim[s_] = p[om + ro_ + c];                    \
    r_pjp[s_] = p[o + ro_ + jp]; r_pip[s_] = p[op + ro_ + c];                    \
    r_zyjm[s_] = *(zy + o + ro_ + jm); r_zyjp[s_] = *(zy + o + ro_ + jp); \
    r_zxim[s_] = *(zx + om + ro_ + c); r_zxip[s_] = *(zx + op + ro_ + c); \
    if (!ZW) { r_a4[s_] = *(a4 + o + ro_ + jp); r_a7[s_] = *(a7 + op + ro_ + c); }      \
  }
#define OW_LOAD(q)                                                               \
  if ((q) <= NZ) {                                                               \
    const long long ko_ = o + (long long)((q)-1) * RS + c; const int s_ = (q) % RO; \
    o_b[s_] = ld_stream<ST>(b + ko_); o_a2[s_] = ld_stream<ST>(a2 + ko_); \
    if (!ZW) { o_a4[s_] = ld_stream<ST>(a4 + ko_); o_a7[s_] = ld_stream<ST>(a7 + ko_); } \
    if (!MGX_PV) o_bet[s_] = ld_stream<ST>(bet + ko_); \
    if (!ZG) { o_zy[ZG ? 0 : s_] = ld_stream<ST>(zy + ko_); o_zx[ZG ? 0 : s_] = ld_stream<ST>(zx + ko_); }       \
  }
  double dg1 = 0, dgn = 0;
  if (MGX_PV) { dg1 = a1[o + c]; dgn = a1[o + (long long)(NZ - 1) * RS + c]; }
  double a4_1 = 0, a4j_1 = 0, a7_1 = 0, a7i_1 = 0, a4_n = 0, a4j_n = 0, a7_n = 0, a7i_n = 0, m4c = 0, m4p = 0, d4c = 1, d4p = 1, m7c = 0, m7p = 0, d7c = 1, d7p = 1;
  if (ZW) {
    const long long rn = (long long)(NZ - 1) * RS;
    a4_1 = a4[o + c]; a4j_1 = a4[o + jp]; a7_1 = a7[o + c]; a7i_1 = a7[op + c];
    a4_n = a4[o + rn + c]; a4j_n = a4[o + rn + jp]; a7_n = a7[o + rn + c]; a7i_n = a7[op + rn + c];
    const long long q2 = (long long)i * RS;
    m4c = L.m4[q2 + c]; m4p = L.m4[q2 + jp]; d4c = L.d4[q2 + c]; d4p = L.d4[q2 + jp];
    m7c = L.m7[q2 + c]; m7p = L.m7[q2 + RS + c]; d7c = L.d7[q2 + c]; d7p = L.d7[q2 + RS + c];
    const long long cq[5] = {q2 + c, q2 + jm, q2 + jp, q2 - RS + c, q2 + RS + c};
#pragma unroll
    for (int q = 0; q < 5; q++) { hh[q] = L.h2[cq[q]]; hv[q] = L.hi2[cq[q]]; hz[q] = L.ze2[cq[q]]; }
    r4c = RCP_REF(d4c); r4p = RCP_REF(d4p); r7c = RCP_REF(d7c); r7p = RCP_REF(d7p);
    if (ZG) { gdx = L.dx2[q2 + c]; gdy = L.dy2[q2 + c]; rdx = RCP_REF(gdx); rdy = RCP_REF(gdy); }
  }
#define ZR_GEN(kk, q) ({ const double z0_ = cffr[(kk)-1] + csr[(kk)-1] * hh[q]; z0_ * hh[q] * hv[q] + hz[q] * (1. + z0_ * hv[q]); })
#define OWN_SLOPES(kk, ZY, ZX) { const double zn1_ = ZR_GEN(kk, 1), zn2_ = ZR_GEN(kk, 2), zn3_ = ZR_GEN(kk, 3), zn4_ = ZR_GEN(kk, 4); \
    ZY = DIVC(hlf * (zn2_ - zn1_), gdy, rdy) * gdx; ZX = DIVC(hlf * (zn4_ - zn3_), gdx, rdx) * gdy; }
#define ZW_GEN(kk, q) ({ const double z0_ = cffw[(kk)-1] + csw[(kk)-1] * hh[q]; z0_ * hh[q] * hv[q] + hz[q] * (1. + z0_ * hv[q]); })
  double d1 = 0, d2 = 0, d3 = 0, d4 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
  if (REAL) {
    const double *__restrict__ q1 = SNAP ? L.p1 : p;
    const long long s = SNAP ? (long long)i * RS : o, sm = SNAP ? s - RS : om, sp = SNAP ? s + RS : op;
    d1 = q1[sm + jp]; d2 = q1[sp + jm]; d3 = q1[sm + jm]; d4 = q1[sp + jp];
    e1 = a5[o + c]; e2 = a5[op + jm]; e3 = a8[o + c]; e4 = a8[op + jp];
  }
#pragma unroll
  for (int q = 1; q <= 1 + D; q++) { NB_LOAD(q) }
#pragma unroll
  for (int q = 1; q <= 1 + D; q++) { OW_LOAD(q) }

  // three-row windows (k-1, k, k+1) of the neighbour columns' p and of the own slopes
  double pjm_m = 0, pjm_0 = r_pjm[1 % RN], pjm_p = 0, pim_m = 0, pim_0 = r_pim[1 % RN], pim_p = 0;
  double pjp_m = 0, pjp_0 = r_pjp[1 % RN], pjp_p = 0, pip_m = 0, pip_0 = r_pip[1 % RN], pip_p = 0;
  double zy_m = 0, zy_0 = 0, zy_p = 0, zx_m = 0, zx_0 = 0, zx_p = 0;
  if (ZG) OWN_SLOPES(1, zy_0, zx_0)
  else { zy_0 = o_zy[ZG ? 0 : 1 % RO]; zx_0 = o_zx[ZG ? 0 : 1 % RO]; }
  double xv = 0.0, betp = 0.0;
  const int lane = threadIdx.x;
#define FWD_ROW(k)                                                        \
  {                                                                     \
    NB_LOAD(k + 1 + D) \
    OW_LOAD(k + 1 + D) \
    if (k < NZ) { \
      const int s1 = (k + 1) % RN, t1 = (k + 1) % RO; \
      pjm_p = r_pjm[s1]; pim_p = r_pim[s1]; pjp_p = r_pjp[s1]; pip_p = r_pip[s1]; \
      if (ZG) OWN_SLOPES(k + 1, zy_p, zx_p) \
      else { zy_p = o_zy[ZG ? 0 : t1]; zx_p = o_zx[ZG ? 0 : t1]; } \
    } \
    const int s = k % RO, n = k % RN; \
    const double zyjm = r_zyjm[n], zyjp = r_zyjp[n], zxim = r_zxim[n], zxip = r_zxip[n]; \
    const double c3 = qrt * (zy_p + zyjm), c3m = qrt * (zyjp + zy_m), c5 = -qrt * (zy_m + zyjm), c5m = -qrt * (zyjp + zy_p); \
    const double c6 = qrt * (zx_p + zxim), c6m = qrt * (zxip + zx_m), c8 = -qrt * (zx_m + zxim), c8m = -qrt * (zxip + zx_p); \
    double a4o, a4jp, a7o, a7ip; \
    if (!ZW) { a4o = o_a4[s]; a4jp = r_a4[n]; a7o = o_a7[s]; a7ip = r_a7[n]; } \
    else if (k == 1) { a4o = a4_1; a4jp = a4j_1; a7o = a7_1; a7ip = a7i_1; } \
    else if (k == NZ) { a4o = a4_n; a4jp = a4j_n; a7o = a7_n; a7ip = a7i_n; } \
    else { \
      if (k == 2) { _Pragma("unroll") for (int q = 0; q < 5; q++) zw0[q] = ZW_GEN(2, q); } \
      _Pragma("unroll") for (int q = 0; q < 5; q++) zw1[q] = ZW_GEN(k + 1, q); \
      const double wo0 = zw0[0], wop1 = zw1[0]; \
      a4o = DIVC(qrt * (wop1 - wo0 + zw1[1] - zw0[1]) * m4c, d4c, r4c); \
      a4jp = DIVC(qrt * (zw1[2] - zw0[2] + wop1 - wo0) * m4p, d4p, r4p); \
      a7o = DIVC(qrt * (wop1 - wo0 + zw1[3] - zw0[3]) * m7c, d7c, r7c); \
      a7ip = DIVC(qrt * (zw1[4] - zw0[4] + wop1 - wo0) * m7p, d7p, r7p); \
      _Pragma("unroll") for (int q = 0; q < 5; q++) zw0[q] = zw1[q]; \
    } \
    double betk; \
    if (MGX_PV) { /* pivots in the kernel, see relax_col_mf */ \
      double dk; \
      if (k == 1) dk = dg1; \
      else if (k == NZ) dk = dgn; \
      else dk = -o_a2[s] - o_a2[(k + 1) % RO] - a4o - a4jp - a7o - a7ip - c6 - c6m - c8 - c8m - c3 - c3m - c5 - c5m; \
      if (k == 1) betk = 1.0 / dk; \
      else { const double gk = o_a2[s] * betp; if (k > LOW + 1) g[k - LOW - 1] = gk; betk = 1.0 / (dk - o_a2[s] * gk); } \
    } else { \
      if (k > LOW + 1) g[k - LOW - 1] = o_a2[s] * betp; \
      betk = o_bet[s]; \
    } \
    betp = betk; \
    double rhs; \
    if (k == 1) { \
      rhs = o_b[s] - c3 * pjm_p - a4o * pjm_0 - a4jp * pjp_0 - c5m * pjp_p \
                   - c6 * pim_p - a7o * pim_0 - a7ip * pip_0 - c8m * pip_p; \
      if (REAL) rhs = rhs - e1 * d1 - e2 * d2 - e3 * d3 - e4 * d4; \
      xv = rhs * betk; \
    } else if (k < NZ) { \
      rhs = o_b[s] - c3 * pjm_p - c3m * pjp_m - a4o * pjm_0 - a4jp * pjp_0 \
                   - c5 * pjm_m - c5m * pjp_p \
                   - c6 * pim_p - c6m * pip_m - a7o * pim_0 - a7ip * pip_0 \
                   - c8 * pim_m - c8m * pip_p; \
      xv = (rhs - o_a2[s] * xv) * betk; \
    } else { \
      rhs = o_b[s] - c3m * pjp_m - a4o * pjm_0 - a4jp * pjp_0 - c5 * pjm_m \
                   - c6m * pip_m - a7o * pim_0 - a7ip * pip_0 - c8 * pim_m; \
      xv = (rhs - o_a2[s] * xv) * betk; \
    } \
    if (k > LOW) x[k - LOW - 1] = xv; else xf[(k - 1) * WAVE + lane] = xv; \
    if (k == LOW + 1) g0 = o_a2[s] * bet_low_in; \
    if (k == LOW) bet_low_in = betk; \
    pjm_m = pjm_0; pjm_0 = pjm_p; pim_m = pim_0; pim_0 = pim_p; \
    pjp_m = pjp_0; pjp_0 = pjp_p; pip_m = pip_0; pip_0 = pip_p; \
    zy_m = zy_0; zy_0 = zy_p; zx_m = zx_0; zx_0 = zx_p; \
  }
  double g0 = 0.0, bet_low_in = 0.0;  // g0 = gam(LOW+1) = a2(LOW+1)*bet(LOW): links the register half to the LDS half
#pragma unroll
  for (int k = 1; k <= LOW; k++) FWD_ROW(k)
#pragma unroll
  for (int k = LOW + 1; k <= NZ; k++) FWD_ROW(k)
#undef FWD_ROW
#undef ZW_GEN
#undef ZR_GEN
#undef OWN_SLOPES
#pragma unroll
  for (int k = UP - 1; k >= 1; k--) x[k - 1] = x[k - 1] - g[k] * x[k];

  const int j = jodd ? 2 * jh + 1 : 2 * jh + 2;
  const bool mS = ph.S && j == 1, mN = ph.N && j == L.ny, mW = ph.W && i == 1, mE = ph.E && i == L.nx;
  const int cS = L.EO, cN = jpos(L, L.ny + 1);
  const long long oW = 0, oE = (long long)(L.nx + 1) * L.plane;
#define STORE_ROW(k, v)                                                                                   \
  {                                                                                                       \
    const long long ro = (long long)((k)-1) * RS;                                                         \
    if (ST) NT2_STORE(v, p + o + ro + c); else p[o + ro + c] = v;                                         \
    if (mS) p[o + ro + cS] = v;                                                                           \
    if (mN) p[o + ro + cN] = v;                                                                           \
    if (mW) { p[oW + ro + c] = v; if (mS) p[oW + ro + cS] = v; if (mN) p[oW + ro + cN] = v; }             \
    if (mE) { p[oE + ro + c] = v; if (mS) p[oE + ro + cS] = v; if (mN) p[oE + ro + cN] = v; }             \
  }
  // lower rows, top down: x(k) = xf(k) - gam(k+1)*x(k+1), gam(k+1) = a2(k+1)*bet(k) (mg_relax.f90:325,330).  With the pivots
  // computed in the kernel the downward pass needs bet(k) again: the recurrence only runs upward, so bet(k), k < LOW, is
  // re-read from the array define_matrices left in memory (same bits), together with a2(k+1), DB rows ahead of use.
  constexpr int DB = 8;
  double r_a2[DB], r_bt[DB];
#define LOW_LOAD(q)                                                                                       \
  if ((q) >= 1 && (q) < LOW) {                                                                            \
    const long long ko_ = o + (long long)(LOW - (q)-1) * RS + c;                                          \
    r_a2[(q) % DB] = ld_stream<ST>(a2 + ko_ + RS); r_bt[(q) % DB] = ld_stream<ST>(bet + ko_);           \
  }
#pragma unroll
  for (int q = 1; q < DB; q++) { LOW_LOAD(q) }
#pragma unroll
  for (int k = LOW + 1; k <= NZ; k++) STORE_ROW(k, x[k - LOW - 1])
  double xn = x[0];
#pragma unroll
  for (int q = 0; q < LOW; q++) {  // row LOW - q
    const double gg = q == 0 ? g0 : r_a2[q % DB] * r_bt[q % DB];
    const double xk = xf[(LOW - q - 1) * WAVE + lane] - gg * xn;
    LOW_LOAD(q + DB)
    STORE_ROW(LOW - q, xk)
    xn = xk;
  }
  if (SNAP && L.p1w != nullptr) {  // next sweep's k=1 snapshot entry of this column (and its physical mirrors): no snapshot launch per pass
    // A mirrored halo cell is read (as a k=1 diagonal) only by columns of the OTHER colour, i.e. by the next pass of this
    // same sweep, which must see it updated: mirrors go to the buffer being read as well (no column of this pass reads them,
    // except a corner column its own corner, after which it is the one to overwrite it).
    double *w1 = L.p1w, *r1 = L.p1;
    const long long so = (long long)i * RS, sW = 0, sE = (long long)(L.nx + 1) * RS;
    const double v1 = xn;
    w1[so + c] = v1;
#define SNAP_MIRROR(idx) { w1[idx] = v1; r1[idx] = v1; }
    if (mS) SNAP_MIRROR(so + cS)
    if (mN) SNAP_MIRROR(so + cN)
    if (mW) { SNAP_MIRROR(sW + c) if (mS) SNAP_MIRROR(sW + cS) if (mN) SNAP_MIRROR(sW + cN) }
    if (mE) { SNAP_MIRROR(sE + c) if (mS) SNAP_MIRROR(sE + cS) if (mN) SNAP_MIRROR(sE + cN) }
#undef SNAP_MIRROR
  }
#undef LOW_LOAD
#undef STORE_ROW
#undef NB_LOAD
#undef OW_LOAD
}

// same launch geometry for the tall-column routine (nz = 128)
template <int NZ, int LOW, bool REAL, bool SNAP, int D, bool ST>
__global__ __launch_bounds__(128, 1) void k_relax_tall(LevView L, int i0, int istep, int nplanes, int jodd_fixed, int rb, Sides ph, int gx) {
  // XCD-aware block -> (j-chunk, plane pair) map.  Blocks are dealt round-robin over the 8 XCDs (b and b+8 share one),
  // each with its own 4 MB L2.  Give every XCD a contiguous range of planes: the pass over plane i and the pass over
  // plane i+2 both read p and the slopes of plane i+1.  The two waves of a block take two consecutive planes of the
  // colour, so those two readers also sit on one CU (speed only; any placement gives the same result).
  const int npair = (nplanes + blockDim.y - 1) / blockDim.y;
  int bx, ipr;
  if (gx < 0) { gx = -gx; ipr = blockIdx.x / gx; bx = blockIdx.x - ipr * gx; }  // MGX_NO_XCD=1 (A/B measurements)
  else if ((npair & 7) == 0) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    ipr = xcd * (npair >> 3) + local / gx;
    bx = local - (local / gx) * gx;
  } else { ipr = blockIdx.x / gx; bx = blockIdx.x - ipr * gx; }
  const int ipl = ipr * blockDim.y + threadIdx.y;
  const int jh = bx * WAVE + threadIdx.x;
  if (jh >= (L.ny >> 1) || ipl >= nplanes) return;
  const int i = i0 + istep * ipl;
  // RB: j = 1+mod(i+rb,2),ny,2 (mg_relax.f90:174) ; FC: fixed parity (:216-217)
  const int jodd = jodd_fixed >= 0 ? jodd_fixed : (((i + rb) & 1) == 0);
  if (sides_part_skip(ph, i, L.nx, jodd, bx, gx)) return;  // wave-uniform
  extern __shared__ double xf_lds[];  // blockDim.y waves x LOW rows x 64 lanes
  relax_col_mf_tall<NZ, LOW, REAL, SNAP, D, ST>(L, i, jh, jodd, ph, xf_lds + (size_t)threadIdx.y * LOW * WAVE);
}


// nz = 128 (BASELINE config 5): matrix-free form only, lower 64 rows through memory (relax_col_mf_tall)
extern "C" int mgxk_relax_nz128(hipStream_t st, const LevView *L, int i0, int istep, int nplanes, int jodd_fixed, int rb, int real, int snap, Sides ph) {
  if (L->zy == nullptr) return 0;
  mgx_before_launch();
  static const bool noxcd = getenv("MGX_NO_XCD") != nullptr, notall = getenv("MGX_NO_TALL") != nullptr;
  if (notall) return 0;
  const int gx0 = (L->ny / 2 + WAVE - 1) / WAVE, gx = noxcd ? -gx0 : gx0;
  const int by = gx0 * nplanes >= 2048 ? 2 : 1;
  dim3 blk(WAVE, by), grd(gx0 * ((nplanes + by - 1) / by));
  const bool stream = (double)L->nx * L->ny * 128 * 72.0 > 256e6;
  const size_t lds = (size_t)by * 64 * WAVE * sizeof(double);  // the lower 64 rows' forward values: 32 KB per wave
#define LAUNCH128_ONE(RV, SV, STV)                                                                                       \
  {                                                                                                                     \
    static bool attr = false;                                                                                           \
    if (!attr) {                                                                                                        \
      if (hipFuncSetAttribute((const void *)k_relax_tall<128, 64, RV, SV, 3, STV>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * WAVE * (int)sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return 0; } \
      attr = true;                                                                                                      \
    }                                                                                                                   \
    hipLaunchKernelGGL((k_relax_tall<128, 64, RV, SV, 3, STV>), grd, blk, lds, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx); \
  }
#define LAUNCH128(STV)                                                                                                  \
  {                                                                                                                     \
    if (real && snap) LAUNCH128_ONE(true, true, STV)                                                                    \
    else if (real) LAUNCH128_ONE(true, false, STV)                                                                      \
    else LAUNCH128_ONE(false, false, STV)                                                                               \
  }
  if (stream) LAUNCH128(true) else LAUNCH128(false)
#undef LAUNCH128
#undef LAUNCH128_ONE
  return mgx_launched();
}

