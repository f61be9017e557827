// The z-line smoother of the multigrid cycle (mg_relax.f90:16-334) as HIP kernels for gfx950 (MI355X): one colour pass per
// launch on the bandwidth-bound levels, whole relax calls in one workgroup on the launch-bound ones.  Hand-written, fp64.
//
// All arithmetic keeps the reference's operation order and is compiled with -ffp-contract=off, so a colour pass of the
// four-colour smoother is bit-identical to the reference's CPU loop.
//
// Thread mapping everywhere: one lane = one (j,i) column, lanes run along the unit-stride half-row of
// the JS layout (mgx_internal.h), so every global access of a wave is one contiguous 512-byte run.
#include <cstdlib>

#include "mgx_device.h"

// ------------------------------------------------------------------------------------------------
// z-line smoother, one colour pass.  mg_relax.f90:237-305 (relax_3D_8_heart) + :308-334 (tridiag).
// Columns of one colour never read each other (four-colour), or only through the k=1 horizontal
// diagonals (red-black): those are then read from the snapshot L.p1 taken before the pass (SNAP).
// The tridiagonal pivots (bet, gam) depend on the matrix only and are precomputed at set-up.
// ------------------------------------------------------------------------------------------------
template <bool REAL, bool SNAP>
__global__ __launch_bounds__(256) void k_relax_colour(LevView L, int i0, int istep, int nplanes, int jodd_fixed, int rb, Sides ph) {
  const int jh = blockIdx.x * WAVE + threadIdx.x;
  const int ipl = blockIdx.y * blockDim.y + threadIdx.y;
  if (jh >= (L.ny >> 1) || ipl >= nplanes) return;
  const int i = i0 + istep * ipl;
  // RB: j = 1+mod(i+rb,2),ny,2 (mg_relax.f90:174) ; FC: fixed parity (:216-217)
  const int jodd = jodd_fixed >= 0 ? jodd_fixed : (((i + rb) & 1) == 0);
  if (sides_part_skip(ph, i, L.nx, jodd, blockIdx.x, gridDim.x)) return;
  int c, jm, jp;  // positions of columns j, j-1, j+1 inside a row
  if (jodd) { c = L.HO + jh; jm = L.EO + jh; jp = jm + 1; }
  else      { c = L.EO + jh + 1; jm = L.HO + jh; jp = jm + 1; }
  const long long RS = L.RS;
  const int nz = L.nz;
  double *__restrict__ p = L.p;
  const double *__restrict__ b = L.b;
  const double *__restrict__ a2 = L.cA[1], *__restrict__ a3 = L.cA[2], *__restrict__ a4 = L.cA[3],
               *__restrict__ a5 = L.cA[4], *__restrict__ a6 = L.cA[5], *__restrict__ a7 = L.cA[6],
               *__restrict__ a8 = L.cA[7], *__restrict__ bet = L.bet, *__restrict__ gam = L.gam;
  const long long o = (long long)i * L.plane, om = o - L.plane, op = o + L.plane;

  // neighbour rows: p of (j-1,i) and (j,i-1); products coef*p of (j+1,i) [slots 3,4,5] and (j,i+1) [6,7,8]
  double pjm_m, pjm_0, pjm_p, pim_m, pim_0, pim_p;
  double m3_m, m3_0, m4_0, m5_p, n6_m, n6_0, n7_0, n8_p;
  double m3_p, m4_p, n6_p, n7_p;
#define LOAD_ROW(q, PJM, PIM, M3, M4, M5, N6, N7, N8)                         \
  {                                                                            \
    const long long ro = (long long)((q)-1) * RS;                              \
    PJM = p[o + ro + jm];                                                      \
    PIM = p[om + ro + c];                                                      \
    const double pj_ = p[o + ro + jp], pi_ = p[op + ro + c];                   \
    M3 = a3[o + ro + jp] * pj_; M4 = a4[o + ro + jp] * pj_; M5 = a5[o + ro + jp] * pj_; \
    N6 = a6[op + ro + c] * pi_; N7 = a7[op + ro + c] * pi_; N8 = a8[op + ro + c] * pi_; \
  }
  double dum5, dum8;
  LOAD_ROW(1, pjm_0, pim_0, m3_0, m4_0, dum5, n6_0, n7_0, dum8);
  LOAD_ROW(2, pjm_p, pim_p, m3_p, m4_p, m5_p, n6_p, n7_p, n8_p);
  (void)dum5; (void)dum8;

  // ---- k = 1 (mg_relax.f90:262-279)
  double rhs = b[o + c] - a3[o + c] * pjm_p - a4[o + c] * pjm_0 - m4_0 - m5_p - a6[o + c] * pim_p - a7[o + c] * pim_0 - n7_0 - n8_p;
  if (REAL) {
    const double *__restrict__ q1 = SNAP ? L.p1 : p;
    const long long s = SNAP ? (long long)i * RS : o, sm = SNAP ? s - RS : om, sp = SNAP ? s + RS : op;
    rhs = rhs - a5[o + c] * q1[sm + jp] - a5[op + jm] * q1[sp + jm] - a8[o + c] * q1[sm + jm] - a8[op + jp] * q1[sp + jp];
  }
  double x = rhs * bet[o + c];
  p[o + c] = x;

  // ---- k = 2 .. nz-1 (:281-291)
  for (int k = 2; k <= nz - 1; k++) {
    pjm_m = pjm_0; pjm_0 = pjm_p; pim_m = pim_0; pim_0 = pim_p;
    m3_m = m3_0; m3_0 = m3_p; m4_0 = m4_p; n6_m = n6_0; n6_0 = n6_p; n7_0 = n7_p;
    LOAD_ROW(k + 1, pjm_p, pim_p, m3_p, m4_p, m5_p, n6_p, n7_p, n8_p);
    const long long ko = o + (long long)(k - 1) * RS + c;
    rhs = b[ko] - a3[ko] * pjm_p - m3_m - a4[ko] * pjm_0 - m4_0 - a5[ko] * pjm_m - m5_p
                - a6[ko] * pim_p - n6_m - a7[ko] * pim_0 - n7_0 - a8[ko] * pim_m - n8_p;
    x = (rhs - a2[ko] * x) * bet[ko];
    p[ko] = x;
  }
  // ---- k = nz (:293-301)
  {
    pjm_m = pjm_0; pjm_0 = pjm_p; pim_m = pim_0; pim_0 = pim_p;
    m3_m = m3_0; m4_0 = m4_p; n6_m = n6_0; n7_0 = n7_p;
    const long long ko = o + (long long)(nz - 1) * RS + c;
    rhs = b[ko] - m3_m - a4[ko] * pjm_0 - m4_0 - a5[ko] * pjm_m - n6_m - a7[ko] * pim_0 - n7_0 - a8[ko] * pim_m;
    x = (rhs - a2[ko] * x) * bet[ko];
    p[ko] = x;
  }
  // ---- back substitution (:330-332): xc(k) = xc(k) - gam(k+1)*xc(k+1)
  for (int k = nz - 1; k >= 1; k--) {
    const long long ko = o + (long long)(k - 1) * RS + c;
    x = p[ko] - gam[ko + RS] * x;
    p[ko] = x;
  }
#undef LOAD_ROW
}


// ------------------------------------------------------------------------------------------------
// Register-resident variant of the colour pass for nz in {2,...,64} (level 1 of the 512x512x64 problem
// is NZ=64).  One wave = 64 columns; a colour of a 512^2 level is only 1024 waves (1 per SIMD), so the
// kernel is written for ONE wave per SIMD and spends its 512-VGPR budget on memory-level parallelism:
//   * the k loop is fully unrolled and software-pipelined: raw neighbour/own rows are loaded D rows ahead
//     into register rings, so ~D*19 independent 512-byte loads are in flight per wave;
//   * the forward solution x(k) and the back-substitution factors gam(k) stay in registers: p is written
//     once (no forward store + backward re-read), and the backward sweep does no dependent loads;
//   * physical-boundary mirrors of the updated columns (mg_mpi_exchange.f90:509-537,552-597) are stored by
//     the lane that owns the column, which removes the separate halo kernel after every colour.
// Arithmetic and its order are identical to k_relax_colour (bit-identical results).
// ------------------------------------------------------------------------------------------------
template <int NZ, bool REAL, bool SNAP, int D>
__device__ __forceinline__ void relax_col_nz(const LevView &L, const int i, const int jh, const int jodd, const Sides ph) {
  int c, jm, jp;
  if (jodd) { c = L.HO + jh; jm = L.EO + jh; jp = jm + 1; }
  else      { c = L.EO + jh + 1; jm = L.HO + jh; jp = jm + 1; }
  const long long RS = L.RS;
  double *__restrict__ p = L.p;
  const double *__restrict__ b = L.b;
  const double *__restrict__ a2 = L.cA[1], *__restrict__ a3 = L.cA[2], *__restrict__ a4 = L.cA[3],
               *__restrict__ a5 = L.cA[4], *__restrict__ a6 = L.cA[5], *__restrict__ a7 = L.cA[6],
               *__restrict__ a8 = L.cA[7], *__restrict__ bet = L.bet;
  const long long o = (long long)i * L.plane, om = o - L.plane, op = o + L.plane;

  constexpr bool ST = false;  // stored-slot path: coarser levels / user matrices, which live in the caches
  constexpr int RN = D + 1;  // raw neighbour rows in flight
  constexpr int RO = D + 1;  // raw own rows in flight
  double r_pjm[RN], r_pim[RN], r_pjp[RN], r_pip[RN], r_a3[RN], r_a4[RN], r_a5[RN], r_a6[RN], r_a7[RN], r_a8[RN];
  double o_b[RO], o_a2[RO], o_a3[RO], o_a4[RO], o_a5[RO], o_a6[RO], o_a7[RO], o_a8[RO], o_bet[RO];
  double x[NZ], g[NZ];

#define NB_LOAD(q)                                                               \
  if ((q) <= NZ) {                                                               \
    const long long ro_ = (long long)((q)-1) * RS; const int s_ = (q) % RN;      \
    r_pjm[s_] = p[o + ro_ + jm]; r_pim[s_] = p[om + ro_ + c];                    \
    r_pjp[s_] = p[o + ro_ + jp]; r_pip[s_] = p[op + ro_ + c];                    \
    r_a3[s_] = a3[o + ro_ + jp]; r_a4[s_] = a4[o + ro_ + jp]; r_a5[s_] = a5[o + ro_ + jp]; \
    r_a6[s_] = a6[op + ro_ + c]; r_a7[s_] = a7[op + ro_ + c]; r_a8[s_] = a8[op + ro_ + c]; \
  }
#define OW_LOAD(q)                                                               \
  if ((q) <= NZ) {                                                               \
    const long long ko_ = o + (long long)((q)-1) * RS + c; const int s_ = (q) % RO; \
    o_b[s_] = ld_stream<ST>(b + ko_); o_a2[s_] = ld_stream<ST>(a2 + ko_); o_a3[s_] = ld_stream<ST>(a3 + ko_); o_a4[s_] = ld_stream<ST>(a4 + ko_); o_a5[s_] = ld_stream<ST>(a5 + ko_); \
    o_a6[s_] = ld_stream<ST>(a6 + ko_); o_a7[s_] = ld_stream<ST>(a7 + ko_); o_a8[s_] = ld_stream<ST>(a8 + ko_); o_bet[s_] = ld_stream<ST>(bet + ko_); \
  }
  // products of a raw neighbour row (computed when the row is first needed)
#define NB_USE(q, PJM, PIM, M3, M4, M5, N6, N7, N8)                              \
  { const int s_ = (q) % RN; PJM = r_pjm[s_]; PIM = r_pim[s_];                   \
    M3 = r_a3[s_] * r_pjp[s_]; M4 = r_a4[s_] * r_pjp[s_]; M5 = r_a5[s_] * r_pjp[s_]; \
    N6 = r_a6[s_] * r_pip[s_]; N7 = r_a7[s_] * r_pip[s_]; N8 = r_a8[s_] * r_pip[s_]; }

  // k = 1 horizontal-diagonal terms (issued first: independent of everything else)
  double d1 = 0, d2 = 0, d3 = 0, d4 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
  if (REAL) {
    const double *__restrict__ q1 = SNAP ? L.p1 : p;
    const long long s = SNAP ? (long long)i * RS : o, sm = SNAP ? s - RS : om, sp = SNAP ? s + RS : op;
    d1 = q1[sm + jp]; d2 = q1[sp + jm]; d3 = q1[sm + jm]; d4 = q1[sp + jp];
    e1 = a5[o + c]; e2 = a5[op + jm]; e3 = a8[o + c]; e4 = a8[op + jp];
  }
  // prologue: neighbour rows 1..1+D... and own rows 1..D
#pragma unroll
  for (int q = 1; q <= 1 + D; q++) { NB_LOAD(q) }
#pragma unroll
  for (int q = 1; q <= D; q++) { OW_LOAD(q) }

  double pjm_m = 0, pjm_0, pjm_p, pim_m = 0, pim_0, pim_p;
  double m3_m = 0, m3_0, m4_0, m5_p, n6_m = 0, n6_0, n7_0, n8_p, m3_p, m4_p, n6_p, n7_p, dum5, dum8;
  NB_USE(1, pjm_0, pim_0, m3_0, m4_0, dum5, n6_0, n7_0, dum8)
  (void)dum5; (void)dum8;
  double xv = 0.0, betp = 0.0;
#pragma unroll
  for (int k = 1; k <= NZ; k++) {
    // keep the pipeline full
    NB_LOAD(k + 1 + D)
    OW_LOAD(k + D)
    if (k < NZ) { NB_USE(k + 1, pjm_p, pim_p, m3_p, m4_p, m5_p, n6_p, n7_p, n8_p) }
    const int s = k % RO;
    double rhs;
    // gam(k) = dd(k-1)*bet(k-1) (mg_relax.f90:325), from values already in registers: no gam stream from HBM
    if (k > 1) g[k - 1] = o_a2[s] * betp;
    betp = o_bet[s];
    if (k == 1) {
      rhs = o_b[s] - o_a3[s] * pjm_p - o_a4[s] * pjm_0 - m4_0 - m5_p - o_a6[s] * pim_p - o_a7[s] * pim_0 - n7_0 - n8_p;
      if (REAL) rhs = rhs - e1 * d1 - e2 * d2 - e3 * d3 - e4 * d4;
      xv = rhs * o_bet[s];
    } else if (k < NZ) {
      rhs = o_b[s] - o_a3[s] * pjm_p - m3_m - o_a4[s] * pjm_0 - m4_0 - o_a5[s] * pjm_m - m5_p
                   - o_a6[s] * pim_p - n6_m - o_a7[s] * pim_0 - n7_0 - o_a8[s] * pim_m - n8_p;
      xv = (rhs - o_a2[s] * xv) * o_bet[s];
    } else {
      rhs = o_b[s] - m3_m - o_a4[s] * pjm_0 - m4_0 - o_a5[s] * pjm_m - n6_m - o_a7[s] * pim_0 - n7_0 - o_a8[s] * pim_m;
      xv = (rhs - o_a2[s] * xv) * o_bet[s];
    }
    x[k - 1] = xv;
    // rotate the three-row window
    pjm_m = pjm_0; pjm_0 = pjm_p; pim_m = pim_0; pim_0 = pim_p;
    m3_m = m3_0; m3_0 = m3_p; m4_0 = m4_p; n6_m = n6_0; n6_0 = n6_p; n7_0 = n7_p;
  }
  // back substitution in registers, then one store per cell (+ mirrors on physical boundaries)
#pragma unroll
  for (int k = NZ - 1; k >= 1; k--) x[k - 1] = x[k - 1] - g[k] * x[k];

  const int j = jodd ? 2 * jh + 1 : 2 * jh + 2;
  const bool mS = ph.S && j == 1, mN = ph.N && j == L.ny, mW = ph.W && i == 1, mE = ph.E && i == L.nx;
  const int cS = L.EO, cN = jpos(L, L.ny + 1);
  const long long oW = 0, oE = (long long)(L.nx + 1) * L.plane;
#pragma unroll
  for (int k = 1; k <= NZ; k++) {
    const long long ro = (long long)(k - 1) * RS;
    const double v = x[k - 1];
    p[o + ro + c] = v;
    if (mS) p[o + ro + cS] = v;
    if (mN) p[o + ro + cN] = v;
    if (mW) { p[oW + ro + c] = v; if (mS) p[oW + ro + cS] = v; if (mN) p[oW + ro + cN] = v; }
    if (mE) { p[oE + ro + c] = v; if (mS) p[oE + ro + cS] = v; if (mN) p[oE + ro + cN] = v; }
  }
  if (SNAP && L.d0w != nullptr) L.d0w[(long long)i * RS + c] = x[0] - L.p1[(long long)i * RS + c];  // mgx_rbseq.hip (b): d0, what k_rbseq_d0 would compute from the stored y
  if (SNAP && L.p1w != nullptr) {  // next sweep's k=1 snapshot entry of this column (and its physical mirrors): no snapshot launch per pass
    // A mirrored halo cell is read (as a k=1 diagonal) only by columns of the OTHER colour, i.e. by the next pass of this
    // same sweep, which must see it updated: mirrors go to the buffer being read as well (no column of this pass reads them,
    // except a corner column its own corner, after which it is the one to overwrite it).
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
#undef NB_LOAD
#undef OW_LOAD
#undef NB_USE
}

// ------------------------------------------------------------------------------------------------
// Matrix-free cross terms.  Away from the special k=1 diagonals, slots 3,5,6,8 are sums of two slope values:
//   cA3(k,j,i) =  qrt*(ZY(k+1,j,i) + ZY(k,j-1,i))     cA5(k,j,i) = -qrt*(ZY(k-1,j,i) + ZY(k,j-1,i))
//   cA6(k,j,i) =  qrt*(ZX(k+1,j,i) + ZX(k,j,i-1))     cA8(k,j,i) = -qrt*(ZX(k-1,j,i) + ZX(k,j,i-1))
// (mg_define_matrix.f90:357-359,397-399,519-555,584-606) with ZY = ((hlf*(zr(k,j+1,i)-zr(k,j-1,i)))/dy)*dx and
// ZX likewise in i.  A column update needs slots 3,5 of itself AND of its j+1 neighbour (6,8: i+1): four stored
// values per direction, but only three slope values (own column window + one row of each neighbour).  Rebuilding
// the four coefficients in registers with the reference's own expression gives bit-identical values and removes
// 2 of the 19 streams of the colour pass (16 B per updated cell).  Slots 2,4,7, the pivots and the k=1 diagonal
// terms stay stored.  Used when the matrix came from define_matrices (not after mgx_set_field(cA)).
// ------------------------------------------------------------------------------------------------
// GL: gam(k) waits in LDS (gl: NZ rows x 64 lanes per wave) instead of registers.  At NZ = 64 x and gam together take half of
// the 512 registers; with the in-kernel pivots on top the kernel spilled (268 B/lane of scratch) -- 32 KB of LDS per wave
// (one wave per SIMD: 128 KB of the CU's 160 KB) frees 128 registers, and the backward sweep's reads are independent of its
// dependency chain.
// ZW: the interior rows of slots 4 and 7 (own and of the j+1 / i+1 neighbour: four streams) are rebuilt from the interface depths
// zw of the column and of its four face neighbours with the reference's expressions (mg_define_matrix.f90:532-534,549-551; the 2-D
// factors come precomputed, k_zw_js), and those depths are not streamed either: the sigma coordinate generates them,
// zw(k,j,i) = z0*h*hinv + zeta*(1.+z0*hinv) with z0 = cffw(k) + csw(k)*h (mg_zr_zw.f90:140-145), from three 2-D values per column
// (h, hinv, zeta: loaded once) and two table entries per row that are the same for every lane (scalar loads).  Same expressions,
// bit-identical values (the halo columns too: h and zeta carry the same mirror / exchange rules as zw), FOUR streams less, at seven
// flops per depth and four fp64 divisions per row under the loads.  Rows 1 and nz of slots 4 / 7 have other formulas
// (:361-372,:577-590): they read the stored slots.
// ZG (with ZW): the column's OWN slopes are not streamed either.  zy(k,j,i) = ((hlf*(zr(k,j+1,i)-zr(k,j-1,i)))/dy(j,i))*dx(j,i), zx likewise in
// i (mg_define_matrix.f90:358,398), need zr of the four face neighbours, whose h, hinv, zeta ZW holds already: zr = z0*h*hinv +
// zeta*(1.+z0*hinv), z0 = cffr(k)+csr(k)*h (mg_zr_zw.f90:112-122) -- four depths, two slopes per row, TWO streams less (the slopes of the
// face neighbours, which would need zr of the second ring, stay streamed).  All six divisors of the generated coefficients (dy, dx of the
// column; the four of slots 4 / 7) are per-column constants: DIVC (mgx_device.h).
template <int NZ, bool REAL, bool SNAP, int D, bool ST, bool GL = false, bool ZW = false, bool ZG = false>
__device__ __forceinline__ void relax_col_mf(const LevView &L, const int i, const int jh, const int jodd, const Sides ph, double *__restrict__ gl = nullptr) {
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
  double o_b[RO], o_a2[RO], o_a4[RO], o_a7[RO], o_bet[RO], o_zy[ZG ? 1 : RO], o_zx[ZG ? 1 : RO];
  double x[NZ], g[GL ? 1 : NZ];
  double zw0[5], zw1[5];  // ZW: generated zw of the column and of its j-1, j+1, i-1, i+1 neighbours, rows k and k+1
  const int lane = threadIdx.x;
#define G_PUT(kk, v) { if (GL) gl[((kk)-1) * WAVE + lane] = (v); else g[GL ? 0 : (kk)-1] = (v); }

#define NB_LOAD(q)                                                               \
  if ((q) <= NZ) {                                                               \
    const long long ro_ = (long long)((q)-1) * RS; const int s_ = (q) % RN;      \
    LD_PAIR(p + o + ro_ + jm, r_pjm[s_], r_pjp[s_]) r_pim[s_] = p[om + ro_ + c];  \
    r_pip[s_] = p[op + ro_ + c];                                                 \
    LD_PAIR(zy + o + ro_ + jm, r_zyjm[s_], r_zyjp[s_])                            \
    r_zxim[s_] = *(zx + om + ro_ + c); r_zxip[s_] = *(zx + op + ro_ + c); \
    if (!ZW) { r_a4[s_] = *(a4 + o + ro_ + jp); r_a7[s_] = *(a7 + op + ro_ + c); }  \
  }
#define OW_LOAD(q)                                                               \
  if ((q) <= NZ) {                                                               \
    const long long ko_ = o + (long long)((q)-1) * RS + c; const int s_ = (q) % RO; \
    o_b[s_] = ld_stream<ST>(b + ko_); o_a2[s_] = ld_stream<ST>(a2 + ko_); \
    if (!ZW) { o_a4[s_] = ld_stream<ST>(a4 + ko_); o_a7[s_] = ld_stream<ST>(a7 + ko_); } \
    if (!MGX_PV) o_bet[s_] = ld_stream<ST>(bet + ko_); \
    if (!ZG) { o_zy[ZG ? 0 : s_] = ld_stream<ST>(zy + ko_); o_zx[ZG ? 0 : s_] = ld_stream<ST>(zx + ko_); }       \
  }
  // PV: the diagonal of the first and the last row is read (two rows of the stored slot 1), the interior rows rebuild it
  double dg1 = 0, dgn = 0;
  if (MGX_PV) { dg1 = a1[o + c]; dgn = a1[o + (long long)(NZ - 1) * RS + c]; }
  // ZW: stored slots 4 and 7 of the first and the last row, and the per-column factors of the interior formula
  double hh[5], hv[5], hz[5];  // h, hinv, zeta of the five columns
  const double *__restrict__ cffw = L.cffw, *__restrict__ csw = L.csw, *__restrict__ cffr = L.cffr, *__restrict__ csr = L.csr;
  double r4c = 0, r4p = 0, r7c = 0, r7p = 0;                 // refined reciprocals of d4c, d4p, d7c, d7p
  double gdx = 1, gdy = 1, rdx = 0, rdy = 0;                 // ZG: dx, dy of the column and their reciprocals
  const double hlf = 0.5;
  double a4_1 = 0, a4j_1 = 0, a7_1 = 0, a7i_1 = 0, a4_n = 0, a4j_n = 0, a7_n = 0, a7i_n = 0, m4c = 0, m4p = 0, d4c = 1, d4p = 1, m7c = 0, m7p = 0, d7c = 1, d7p = 1;
  if (ZW) {
    const long long rn = (long long)(NZ - 1) * RS;
    a4_1 = a4[o + c]; a4j_1 = a4[o + jp]; a7_1 = a7[o + c]; a7i_1 = a7[op + c];
    a4_n = a4[o + rn + c]; a4j_n = a4[o + rn + jp]; a7_n = a7[o + rn + c]; a7i_n = a7[op + rn + c];
    const long long q2 = (long long)i * RS;
    m4c = L.m4[q2 + c]; m4p = L.m4[q2 + jp]; d4c = L.d4[q2 + c]; d4p = L.d4[q2 + jp];
    m7c = L.m7[q2 + c]; m7p = L.m7[q2 + RS + c]; d7c = L.d7[q2 + c]; d7p = L.d7[q2 + RS + c];
    const long long cq[5] = {q2 + c, q2 + jm, q2 + jp, q2 - RS + c, q2 + RS + c};  // the column, j-1, j+1, i-1, i+1
#pragma unroll
    for (int q = 0; q < 5; q++) { hh[q] = L.h2[cq[q]]; hv[q] = L.hi2[cq[q]]; hz[q] = L.ze2[cq[q]]; }
    r4c = RCP_REF(d4c); r4p = RCP_REF(d4p); r7c = RCP_REF(d7c); r7p = RCP_REF(d7p);
    if (ZG) { gdx = L.dx2[q2 + c]; gdy = L.dy2[q2 + c]; rdx = RCP_REF(gdx); rdy = RCP_REF(gdy); }
  }
  // zr(kk, column q) by its generating formula (mg_zr_zw.f90:112-122); own slopes of row kk from the four face neighbours
#define ZR_GEN(kk, q) ({ const double z0_ = cffr[(kk)-1] + csr[(kk)-1] * hh[q]; z0_ * hh[q] * hv[q] + hz[q] * (1. + z0_ * hv[q]); })
#define OWN_SLOPES(kk, ZY, ZX) { const double zn1_ = ZR_GEN(kk, 1), zn2_ = ZR_GEN(kk, 2), zn3_ = ZR_GEN(kk, 3), zn4_ = ZR_GEN(kk, 4); \
    ZY = DIVC(hlf * (zn2_ - zn1_), gdy, rdy) * gdx; ZX = DIVC(hlf * (zn4_ - zn3_), gdx, rdx) * gdy; }
  // zw(kk, column q) by its generating formula (mg_zr_zw.f90:140-145)
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
#pragma unroll
  for (int k = 1; k <= NZ; k++) {
    NB_LOAD(k + 1 + D)
    OW_LOAD(k + 1 + D)
    if (k < NZ) {
      const int s1 = (k + 1) % RN, t1 = (k + 1) % RO;
      pjm_p = r_pjm[s1]; pim_p = r_pim[s1]; pjp_p = r_pjp[s1]; pip_p = r_pip[s1];
      if (ZG) OWN_SLOPES(k + 1, zy_p, zx_p)
      else { zy_p = o_zy[ZG ? 0 : t1]; zx_p = o_zx[ZG ? 0 : t1]; }
    }
    const int s = k % RO, n = k % RN;
    const double zyjm = r_zyjm[n], zyjp = r_zyjp[n], zxim = r_zxim[n], zxip = r_zxip[n];
    // the eight cross coefficients of this row, rebuilt from the slopes (header comment): own slots 3,5,6,8 and the
    // mirrored ones stored at the j+1 / i+1 neighbours
    const double c3 = qrt * (zy_p + zyjm), c3m = qrt * (zyjp + zy_m), c5 = -qrt * (zy_m + zyjm), c5m = -qrt * (zyjp + zy_p);
    const double c6 = qrt * (zx_p + zxim), c6m = qrt * (zxip + zx_m), c8 = -qrt * (zx_m + zxim), c8m = -qrt * (zxip + zx_p);
    double a4o, a4jp, a7o, a7ip;  // slots 4 and 7 of the cell and of its j+1 / i+1 neighbour
    if (!ZW) { a4o = o_a4[s]; a4jp = r_a4[n]; a7o = o_a7[s]; a7ip = r_a7[n]; }
    else if (k == 1) { a4o = a4_1; a4jp = a4j_1; a7o = a7_1; a7ip = a7i_1; }
    else if (k == NZ) { a4o = a4_n; a4jp = a4j_n; a7o = a7_n; a7ip = a7i_n; }
    else {
      if (k == 2) {
#pragma unroll
        for (int q = 0; q < 5; q++) zw0[q] = ZW_GEN(2, q);
      }
#pragma unroll
      for (int q = 0; q < 5; q++) zw1[q] = ZW_GEN(k + 1, q);
      const double wo0 = zw0[0], wop1 = zw1[0];
      a4o = DIVC(qrt * (wop1 - wo0 + zw1[1] - zw0[1]) * m4c, d4c, r4c);
      a4jp = DIVC(qrt * (zw1[2] - zw0[2] + wop1 - wo0) * m4p, d4p, r4p);
      a7o = DIVC(qrt * (wop1 - wo0 + zw1[3] - zw0[3]) * m7c, d7c, r7c);
      a7ip = DIVC(qrt * (zw1[4] - zw0[4] + wop1 - wo0) * m7p, d7p, r7p);
#pragma unroll
      for (int q = 0; q < 5; q++) zw0[q] = zw1[q];
    }
    double betk;
    if (MGX_PV) {
      // pivots in the kernel: d(k) = cA(1,k,j,i) is minus the sum of the fourteen couplings of the row, added in the order of
      // mg_define_matrix.f90:632-639 (bit-identical to the stored value; rows 1 and nz, which have their own formulas
      // :619-627,:642-654, are read), then tridiag's recurrence (mg_relax.f90:322-326) -- no bet stream from HBM
      double dk;
      if (k == 1) dk = dg1;
      else if (k == NZ) dk = dgn;
      else dk = -o_a2[s] - o_a2[(k + 1) % RO] - a4o - a4jp - a7o - a7ip - c6 - c6m - c8 - c8m - c3 - c3m - c5 - c5m;
      if (k == 1) betk = 1.0 / dk;
      else { const double gk = o_a2[s] * betp; G_PUT(k, gk) betk = 1.0 / (dk - o_a2[s] * gk); }
    } else {
      if (k > 1) G_PUT(k, o_a2[s] * betp)
      betk = o_bet[s];
    }
    betp = betk;
    double rhs;
    if (k == 1) {
      rhs = o_b[s] - c3 * pjm_p - a4o * pjm_0 - a4jp * pjp_0 - c5m * pjp_p
                   - c6 * pim_p - a7o * pim_0 - a7ip * pip_0 - c8m * pip_p;
      if (REAL) rhs = rhs - e1 * d1 - e2 * d2 - e3 * d3 - e4 * d4;
      xv = rhs * betk;
    } else if (k < NZ) {
      rhs = o_b[s] - c3 * pjm_p - c3m * pjp_m - a4o * pjm_0 - a4jp * pjp_0
                   - c5 * pjm_m - c5m * pjp_p
                   - c6 * pim_p - c6m * pip_m - a7o * pim_0 - a7ip * pip_0
                   - c8 * pim_m - c8m * pip_p;
      xv = (rhs - o_a2[s] * xv) * betk;
    } else {
      rhs = o_b[s] - c3m * pjp_m - a4o * pjm_0 - a4jp * pjp_0 - c5 * pjm_m
                   - c6m * pip_m - a7o * pim_0 - a7ip * pip_0 - c8 * pim_m;
      xv = (rhs - o_a2[s] * xv) * betk;
    }
    x[k - 1] = xv;
    pjm_m = pjm_0; pjm_0 = pjm_p; pim_m = pim_0; pim_0 = pim_p;
    pjp_m = pjp_0; pjp_0 = pjp_p; pip_m = pip_0; pip_0 = pip_p;
    zy_m = zy_0; zy_0 = zy_p; zx_m = zx_0; zx_0 = zx_p;
  }
#pragma unroll
  for (int k = NZ - 1; k >= 1; k--) x[k - 1] = x[k - 1] - (GL ? gl[k * WAVE + lane] : g[GL ? 0 : k]) * x[k];
#undef G_PUT

  const int j = jodd ? 2 * jh + 1 : 2 * jh + 2;
  const bool mS = ph.S && j == 1, mN = ph.N && j == L.ny, mW = ph.W && i == 1, mE = ph.E && i == L.nx;
  const int cS = L.EO, cN = jpos(L, L.ny + 1);
  const long long oW = 0, oE = (long long)(L.nx + 1) * L.plane;
#pragma unroll
  for (int k = 1; k <= NZ; k++) {
    const long long ro = (long long)(k - 1) * RS;
    const double v = x[k - 1];
    if (ST) NT2_STORE(v, p + o + ro + c); else p[o + ro + c] = v;
    if (mS) p[o + ro + cS] = v;
    if (mN) p[o + ro + cN] = v;
    if (mW) { p[oW + ro + c] = v; if (mS) p[oW + ro + cS] = v; if (mN) p[oW + ro + cN] = v; }
    if (mE) { p[oE + ro + c] = v; if (mS) p[oE + ro + cS] = v; if (mN) p[oE + ro + cN] = v; }
  }
  if (SNAP && L.d0w != nullptr) L.d0w[(long long)i * RS + c] = x[0] - L.p1[(long long)i * RS + c];  // mgx_rbseq.hip (b): d0, what k_rbseq_d0 would compute from the stored y
  if (SNAP && L.p1w != nullptr) {  // next sweep's k=1 snapshot entry of this column (and its physical mirrors): no snapshot launch per pass
    // A mirrored halo cell is read (as a k=1 diagonal) only by columns of the OTHER colour, i.e. by the next pass of this
    // same sweep, which must see it updated: mirrors go to the buffer being read as well (no column of this pass reads them,
    // except a corner column its own corner, after which it is the one to overwrite it).
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
#undef NB_LOAD
#undef OW_LOAD
#undef ZW_GEN
#undef ZR_GEN
#undef OWN_SLOPES
}

template <int NZ, bool REAL, bool SNAP, int D, bool MF, bool ST>
__global__ __launch_bounds__(128, 1) void k_relax_nz(LevView L, int i0, int istep, int nplanes, int jodd_fixed, int rb, Sides ph, int gx) {
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
  constexpr bool GL = MF && NZ == 64 && MGX_GL;  // keep in step with launch_relax_nz_d
  constexpr bool ZW = MF && NZ >= 32 && MGX_ZW;
  if (GL) {
    extern __shared__ double g_lds[];  // blockDim.y waves x NZ rows x 64 lanes
    relax_col_mf<NZ, REAL, SNAP, D, ST, GL, ZW, ZW && MGX_ZG>(L, i, jh, jodd, ph, g_lds + (size_t)threadIdx.y * NZ * WAVE);
  } else if (MF) relax_col_mf<NZ, REAL, SNAP, D, ST, false, ZW, ZW && MGX_ZG>(L, i, jh, jodd, ph);
  else relax_col_nz<NZ, REAL, SNAP, D>(L, i, jh, jodd, ph);
}

// Lexicographic Gauss-Seidel (mg_relax.f90:116-148) on the device, EXACTLY: column (j,i) of the reference's
// `do i; do j` sweep reads new values of (j-1,i), (j,i-1), (j+1,i-1), (j-1,i-1) and old values of the rest, so all
// columns with equal t = j + 2 i are independent and hyperplanes t = 3 .. ny+2nx are processed in order (one
// launch each).  Slow (launch-bound, ~ny+2nx launches per sweep) but bit-identical to the sequential loop.
template <int NZ, bool REAL>
__global__ __launch_bounds__(64, 1) void k_relax_gs_front(LevView L, int t) {
  int ilo = (t - L.ny + 1) / 2; if (ilo < 1) ilo = 1;        // j = t - 2i <= ny
  const int i = ilo + blockIdx.x * WAVE + threadIdx.x;
  const int j = t - 2 * i;
  if (i > L.nx || j < 1 || j > L.ny) return;
  const Sides none = {0, 0, 0, 0};  // halo is refreshed once per sweep, after the loop (mg_relax.f90:141)
  relax_col_nz<NZ, REAL, false, 1>(L, i, (j - 1) >> 1, j & 1, none);
}

// Whole relax(lev, nsweeps) of a SMALL level (<= 1024 columns per colour, no neighbours) in ONE launch of ONE
// workgroup: colours are separated by __syncthreads() instead of kernel boundaries.  The coarsest-level solve of
// the reference (40 sweeps, mg_solvers.f90:117,144) is 160 colour passes of a 16x16x2 grid: launch-bound as
// separate kernels, ~1 us per pass here.  method: 1 = RB, 2 = FC.
template <int NZ, bool REAL, int MAXT>
__global__ __launch_bounds__(MAXT) void k_relax_small(LevView L, int nsweeps, int method, Sides ph, int exact) {
  const int tid = threadIdx.x, nth = blockDim.x;
  const int nyh = L.ny >> 1;
  for (int it = 0; it < nsweeps; it++) {
    const int ncolour = method == 2 ? 4 : 2;
    for (int cidx = 0; cidx < ncolour; cidx++) {
      if (method == 1 && REAL && exact) {
        // the reference's sequential red-black order (mg_relax.f90:173-176): planes i = 1..nx one after the other, so that a
        // column reads the already updated same-colour k=1 diagonals of plane i-1 and the old ones of plane i+1 (:271-276)
        for (int i = 1; i <= L.nx; i++) {
          const int jodd = ((i + cidx + 1) & 1) == 0;
          for (int t = tid; t < nyh; t += nth) relax_col_nz<NZ, REAL, false, 1>(L, i, t, jodd, ph);
          __syncthreads();
        }
        continue;
      }
      if (method == 1 && REAL) {  // snapshot of p(k=1) for the same-colour diagonals of red-black
        for (int t = tid; t < (L.nx + 2) * L.RS; t += nth) L.p1[t] = L.p[(long long)(t / L.RS) * L.plane + (t % L.RS)];
        __syncthreads();
      }
      const int ncol = method == 2 ? (L.nx >> 1) * nyh : L.nx * nyh;
      for (int t = tid; t < ncol; t += nth) {
        const int ipl = t / nyh, jh = t - ipl * nyh;
        int i, jodd;
        if (method == 2) { i = 1 + (cidx >> 1) + 2 * ipl; jodd = (cidx & 1) == 0; }
        else { i = 1 + ipl; jodd = ((i + cidx + 1) & 1) == 0; }
        if (method == 1) relax_col_nz<NZ, REAL, REAL, 1>(L, i, jh, jodd, ph);
        else relax_col_nz<NZ, REAL, false, 1>(L, i, jh, jodd, ph);
      }
      __syncthreads();
    }
  }
}

// snapshot of p(k=1,:,:) for the parallel red-black pass
__global__ void k_snapshot_k1(LevView L) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (t < L.RS) L.p1[(long long)i * L.RS + t] = L.p[(long long)i * L.plane + t];
}


// Register-resident relax of a level with <= 1024 columns (the coarsest levels: 16x16x2, and 16x16x4 / 32x32x2 class
// grids): ONE workgroup, one thread per column for the whole call.  Everything that does not change between colour
// passes -- b, the 7 own off-diagonal slots and pivots, the 6 neighbour slots the symmetric storage makes a column
// read, the k=1 diagonal slots, gam -- is loaded into registers once; only p lives in LDS (with its mirrored halo) and
// is exchanged there.  A colour pass is then ~4 NZ LDS reads, ~30 NZ flops and one barrier (0.1-0.2 us instead of
// 0.8 us with every operand re-read from LDS, and ~5 us as a separate launch): relax(nlevs, ns_coarsest=40) is 160
// dependent passes.  Same expressions in the same order as relax_col_nz: bit-identical.
template <int NZ, bool REAL, int MAXT>
__global__ __launch_bounds__(MAXT) void k_relax_reg(LevView G, int nsweeps, int method, Sides ph, int exact_in) {
  extern __shared__ double ldsr[];
  double *lds = ldsr;
  const int nx = G.nx, ny = G.ny, W = ny + 2, PL = (nx + 2) * W;  // P[k][i][j]
  double *__restrict__ P = lds, *__restrict__ P1 = lds + NZ * PL;  // P1: k=1 snapshot of the parallel red-black pass
  const int tid = threadIdx.x, nth = blockDim.x;
#define GI(k0, jj, ii) ((long long)(ii) * G.plane + (long long)(k0) * G.RS + jpos(G, jj))
  for (int t = tid; t < NZ * PL; t += nth) {
    const int k0 = t / PL, r = t - k0 * PL, i = r / W, j = r - i * W;
    P[t] = G.p[GI(k0, j, i)];
  }
  const int j = 1 + tid % ny, i = 1 + tid / ny;
  const bool mine = tid < nx * ny;
  double ob[NZ], a2[NZ], a3[NZ], a4[NZ], a5[NZ], a6[NZ], a7[NZ], a8[NZ], bet[NZ], g[NZ];
  double r3[NZ], r4[NZ], r5[NZ], r6[NZ], r7[NZ], r8[NZ], e2 = 0, e4 = 0;
  if (mine) {
#pragma unroll
    for (int k = 0; k < NZ; k++) {
      const long long c = GI(k, j, i), cj = GI(k, j + 1, i), ci = GI(k, j, i + 1);
      ob[k] = G.b[c]; a2[k] = G.cA[1][c]; a3[k] = G.cA[2][c]; a4[k] = G.cA[3][c]; a5[k] = G.cA[4][c];
      a6[k] = G.cA[5][c]; a7[k] = G.cA[6][c]; a8[k] = G.cA[7][c]; bet[k] = G.bet[c];
      r3[k] = G.cA[2][cj]; r4[k] = G.cA[3][cj]; r5[k] = G.cA[4][cj];
      r6[k] = G.cA[5][ci]; r7[k] = G.cA[6][ci]; r8[k] = G.cA[7][ci];
    }
    if (REAL) { e2 = G.cA[4][GI(0, j - 1, i + 1)]; e4 = G.cA[7][GI(0, j + 1, i + 1)]; }
    g[0] = 0.0;
#pragma unroll
    for (int k = 1; k < NZ; k++) g[k] = a2[k] * bet[k - 1];  // gam(k) = dd(k-1)*bet(k-1) (mg_relax.f90:325)
  }
  __syncthreads();
  const bool mS = ph.S && j == 1, mN = ph.N && j == ny, mW = ph.W && i == 1, mE = ph.E && i == nx;
  const int ncolour = method == 2 ? 4 : 2;
  // exact: the reference's sequential red-black order (mg_relax.f90:173-176) -- the planes of a colour one after the other,
  // k=1 diagonals read from p itself (new in plane i-1, old in plane i+1) instead of the snapshot
  const bool exact = exact_in && method == 1 && REAL;
  for (int it = 0; it < nsweeps; it++) {
    for (int cidx = 0; cidx < ncolour; cidx++) {
      if (method == 1 && REAL && !exact) {
        for (int t = tid; t < PL; t += nth) P1[t] = P[t];
        __syncthreads();
      }
      bool colour;
      if (method == 2) colour = ((i & 1) == ((cidx >> 1) ? 0 : 1)) && ((j & 1) == ((cidx & 1) ? 0 : 1));  // mg_relax.f90:214-217
      else colour = (j & 1) == ((((i + cidx + 1) & 1) == 0) ? 1 : 0);                                        // :174
      const int nstep = exact ? nx : 1;
      for (int ip = 1; ip <= nstep; ip++) {
      const bool active = colour && (!exact || i == ip);
      if (mine && active) {
        double x[NZ];
        const double *__restrict__ Q1 = (method == 1 && REAL && !exact) ? P1 : P;
        double d1 = 0, d2 = 0, d3 = 0, d4 = 0;
        if (REAL) { d1 = Q1[(i - 1) * W + j + 1]; d2 = Q1[(i + 1) * W + j - 1]; d3 = Q1[(i - 1) * W + j - 1]; d4 = Q1[(i + 1) * W + j + 1]; }
        double pjm[NZ], pjp[NZ], pim[NZ], pip[NZ];
#pragma unroll
        for (int k = 0; k < NZ; k++) {
          const int o = k * PL + i * W + j;
          pjm[k] = P[o - 1]; pjp[k] = P[o + 1]; pim[k] = P[o - W]; pip[k] = P[o + W];
        }
        double xv = 0.0;
#pragma unroll
        for (int k = 0; k < NZ; k++) {
          double rhs;
          if (k == 0) {
            rhs = ob[k] - a3[k] * pjm[k + 1] - a4[k] * pjm[k] - r4[k] * pjp[k] - r5[k + 1] * pjp[k + 1]
                        - a6[k] * pim[k + 1] - a7[k] * pim[k] - r7[k] * pip[k] - r8[k + 1] * pip[k + 1];
            if (REAL) rhs = rhs - a5[0] * d1 - e2 * d2 - a8[0] * d3 - e4 * d4;
            xv = rhs * bet[k];
          } else if (k < NZ - 1) {
            rhs = ob[k] - a3[k] * pjm[k + 1] - r3[k - 1] * pjp[k - 1] - a4[k] * pjm[k] - r4[k] * pjp[k] - a5[k] * pjm[k - 1] - r5[k + 1] * pjp[k + 1]
                        - a6[k] * pim[k + 1] - r6[k - 1] * pip[k - 1] - a7[k] * pim[k] - r7[k] * pip[k] - a8[k] * pim[k - 1] - r8[k + 1] * pip[k + 1];
            xv = (rhs - a2[k] * xv) * bet[k];
          } else {
            rhs = ob[k] - r3[k - 1] * pjp[k - 1] - a4[k] * pjm[k] - r4[k] * pjp[k] - a5[k] * pjm[k - 1]
                        - r6[k - 1] * pip[k - 1] - a7[k] * pim[k] - r7[k] * pip[k] - a8[k] * pim[k - 1];
            xv = (rhs - a2[k] * xv) * bet[k];
          }
          x[k] = xv;
        }
#pragma unroll
        for (int k = NZ - 2; k >= 0; k--) x[k] = x[k] - g[k + 1] * x[k + 1];
#pragma unroll
        for (int k = 0; k < NZ; k++) {
          const int o = k * PL;
          const double v = x[k];
          P[o + i * W + j] = v;
          if (mS) P[o + i * W] = v;
          if (mN) P[o + i * W + ny + 1] = v;
          if (mW) { P[o + j] = v; if (mS) P[o] = v; if (mN) P[o + ny + 1] = v; }
          if (mE) { P[o + (nx + 1) * W + j] = v; if (mS) P[o + (nx + 1) * W] = v; if (mN) P[o + (nx + 1) * W + ny + 1] = v; }
        }
      }
      __syncthreads();
      }
    }
  }
  for (int t = tid; t < NZ * PL; t += nth) {
    const int k0 = t / PL, r = t - k0 * PL, ii = r / W, jj = r - ii * W;
    G.p[GI(k0, jj, ii)] = P[t];
  }
#undef GI
}

// Coarsest-level solve entirely out of LDS: when p, b, slots 2..8 and the pivots of a level fit in 64 KB (16x16x2:
// 57 KB), ONE workgroup copies them in (compact JS layout), runs all nsweeps x colours with the same column routine
// (its pointers now address LDS), and writes p back.  relax(nlevs, ns_coarsest=40) = 160 dependent colour passes:
// ~0.3 us each from LDS instead of ~1.1 us through L2 (and ~5 us as separate launches).
template <int NZ, bool REAL>
__global__ __launch_bounds__(256) void k_relax_tiny(LevView G, int nsweeps, int method, Sides ph) {
  extern __shared__ double lds[];
  LevView L = G;
  L.EO = 0; L.HO = (G.ny >> 1) + 1; L.RS = G.ny + 2; L.plane = (long long)NZ * L.RS;
  const int n3 = (G.nx + 2) * (int)L.plane;
  double *base = lds;
  L.p = base; base += n3; L.b = base; base += n3;
  for (int q = 1; q < 8; q++) { L.cA[q] = base; base += n3; }
  L.cA[0] = nullptr; L.bet = base; base += n3; L.gam = nullptr; L.zy = L.zx = nullptr;
  L.p1 = base;
  const int tid = threadIdx.x, nth = blockDim.x;
  for (int t = tid; t < n3; t += nth) {  // copy in: compact index t -> (i,k,j) -> padded global index
    const int i = t / (int)L.plane, rem = t - i * (int)L.plane, k = rem / L.RS, pos = rem - k * L.RS;
    const int j = pos < L.HO ? 2 * pos : 2 * (pos - L.HO) + 1;
    const long long gidx = (long long)i * G.plane + (long long)k * G.RS + jpos(G, j);
    L.p[t] = G.p[gidx]; L.b[t] = G.b[gidx]; L.bet[t] = G.bet[gidx];
    for (int q = 1; q < 8; q++) L.cA[q][t] = G.cA[q][gidx];
  }
  __syncthreads();
  const int nyh = L.ny >> 1;
  for (int it = 0; it < nsweeps; it++) {
    const int ncolour = method == 2 ? 4 : 2;
    for (int cidx = 0; cidx < ncolour; cidx++) {
      if (method == 1 && REAL) {
        for (int t = tid; t < (L.nx + 2) * L.RS; t += nth) L.p1[t] = L.p[(t / L.RS) * (int)L.plane + (t % L.RS)];
        __syncthreads();
      }
      const int ncol = method == 2 ? (L.nx >> 1) * nyh : L.nx * nyh;
      for (int t = tid; t < ncol; t += nth) {
        const int ipl = t / nyh, jh = t - ipl * nyh;
        int i, jodd;
        if (method == 2) { i = 1 + (cidx >> 1) + 2 * ipl; jodd = (cidx & 1) == 0; }
        else { i = 1 + ipl; jodd = ((i + cidx + 1) & 1) == 0; }
        if (method == 1) relax_col_nz<NZ, REAL, REAL, 1>(L, i, jh, jodd, ph);
        else relax_col_nz<NZ, REAL, false, 1>(L, i, jh, jodd, ph);
      }
      __syncthreads();
    }
  }
  for (int t = tid; t < n3; t += nth) {  // copy p back (halo included: the column routine kept it mirrored)
    const int i = t / (int)L.plane, rem = t - i * (int)L.plane, k = rem / L.RS, pos = rem - k * L.RS;
    const int j = pos < L.HO ? 2 * pos : 2 * (pos - L.HO) + 1;
    G.p[(long long)i * G.plane + (long long)k * G.RS + jpos(G, j)] = L.p[t];
  }
}

template <int NZ, int D>
static void launch_relax_nz_d(hipStream_t st, const LevView *L, int i0, int istep, int nplanes, int jodd_fixed, int rb, int real, int snap, Sides ph) {
  static const bool noxcd = getenv("MGX_NO_XCD") != nullptr;
  const int gx0 = (L->ny / 2 + WAVE - 1) / WAVE, gx = noxcd ? -gx0 : gx0;
  // two planes per workgroup only when there are plenty of workgroups (it halves the number of CUs a small level uses)
  const int by = gx0 * nplanes >= 2048 ? 2 : 1;
  dim3 blk(WAVE, by), grd(gx0 * ((nplanes + by - 1) / by));
  const bool mf = L->zy != nullptr && NZ >= 16;  // matrix-free cross terms on the bandwidth-bound levels
  // streaming (nontemporal) hints only when the level cannot live in the 256 MB Infinity Cache between passes:
  // measured +14 % on the 1.2 GB level 1, -20 % on the 150 MB level 2 which is otherwise re-read from cache
  const bool stream = mf && (double)L->nx * L->ny * NZ * 72.0 > 256e6;
  // gam in LDS (GL, see relax_col_mf): 32 KB per wave of the matrix-free NZ = 64 kernel
#define LAUNCH_NZ(MFV, STV)                                                                                              \
  {                                                                                                                       \
    const size_t lds = (MFV && NZ == 64 && MGX_GL) ? (size_t)by * NZ * WAVE * sizeof(double) : 0;                          \
    if (real && snap) hipLaunchKernelGGL((k_relax_nz<NZ, true, true, D, MFV, STV>), grd, blk, lds, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx); \
    else if (real) hipLaunchKernelGGL((k_relax_nz<NZ, true, false, D, MFV, STV>), grd, blk, lds, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx);   \
    else hipLaunchKernelGGL((k_relax_nz<NZ, false, false, D, MFV, STV>), grd, blk, lds, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph, gx);            \
  }
  if (mf && stream) LAUNCH_NZ(true, true)
  else if (mf) LAUNCH_NZ(true, false)
  else LAUNCH_NZ(false, false)
#undef LAUNCH_NZ
}
template <int NZ>
static void launch_relax_nz(hipStream_t st, const LevView *L, int i0, int istep, int nplanes, int jodd_fixed, int rb, int real, int snap, Sides ph) {
  // look-ahead depth D (rows of loads in flight)
  // measured: NZ=16 6.3 us/pass at D=2 vs 9.3 at D=3; NZ>=32 flat for D=2..5
#ifdef MGX_D64
  constexpr int D = NZ == 64 ? MGX_D64 : (NZ >= 32 ? 3 : (NZ >= 8 ? 2 : 1));
#else
  constexpr int D = NZ >= 32 ? 3 : (NZ >= 8 ? 2 : 1);
#endif
#ifdef MGX_TUNE_D
  if (NZ >= 8) {
    static const int dd = getenv("MGX_D") ? atoi(getenv("MGX_D")) : 3;
    if (dd == 1) return launch_relax_nz_d<NZ, 1>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph);
    if (dd == 2) return launch_relax_nz_d<NZ, 2>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph);
    if (dd == 5) return launch_relax_nz_d<NZ, 5>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph);
    if (dd == 7) return launch_relax_nz_d<NZ, 7>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph);
  }
#endif
  launch_relax_nz_d<NZ, D>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph);
}

extern "C" {

// one Gauss-Seidel sweep as ny+2nx-2 hyperplane launches; returns 0 when nz has no register-resident variant
int mgxk_relax_gs_sweep(hipStream_t st, const LevView *L, int real) {
  for (int t = 3; t <= L->ny + 2 * L->nx; t++) {
    int ilo = (t - L->ny + 1) / 2; if (ilo < 1) ilo = 1;
    int ihi = (t - 1) / 2; if (ihi > L->nx) ihi = L->nx;
    if (ihi < ilo) continue;
    dim3 grd((ihi - ilo + 1 + WAVE - 1) / WAVE), blk(WAVE);
#define GS_CASE(NZV) case NZV: if (real) hipLaunchKernelGGL((k_relax_gs_front<NZV, true>), grd, blk, 0, st, *L, t); \
                               else hipLaunchKernelGGL((k_relax_gs_front<NZV, false>), grd, blk, 0, st, *L, t); break;
    switch (L->nz) { GS_CASE(2) GS_CASE(4) GS_CASE(8) GS_CASE(16) GS_CASE(32) GS_CASE(64) default: return 0; }
#undef GS_CASE
  }
  return 1;
}

// one-launch relax of a small level; returns 0 if the level does not qualify
int mgxk_relax_wave(hipStream_t, const LevView *, int, int, int, Sides, int);  // mgx_relax_coarse.hip
// mode (red-black with cmatrix='real'): 0 = parallel colour passes (snapshot), 1 = the reference's plane loop bit for bit (rb_exact), 2 = the
// same order by the walk of mgx_rbseq.hip where a kernel has it (k_relax_wave), else the plane loop
int mgxk_relax_small(hipStream_t st, const LevView *L, int nsweeps, int method, int real, Sides ph, int mode) {
  mgx_before_launch();
  if (mgxk_relax_wave(st, L, nsweeps, method, real, ph, mode)) return 1;  // <= 256 columns, nz = 2: the whole level in one wave
  const int exact = mode != 0;
  {  // one thread per column, coefficients in registers, p in LDS
    static const bool noreg = getenv("MGX_NO_REG") != nullptr;
    const int ncols = L->nx * L->ny;
    const bool closed = ph.S && ph.E && ph.N && ph.W;
    if (!noreg && closed && method != 0 && ((L->nz == 2 && ncols <= 1024) || (L->nz == 4 && ncols <= 256))) {
      const size_t bytes = ((size_t)L->nz + 1) * (L->nx + 2) * (L->ny + 2) * sizeof(double);
      const int nth = (ncols + 63) / 64 * 64;
      if (L->nz == 2) { if (real) hipLaunchKernelGGL((k_relax_reg<2, true, 1024>), dim3(1), dim3(nth), bytes, st, *L, nsweeps, method, ph, exact);
                        else hipLaunchKernelGGL((k_relax_reg<2, false, 1024>), dim3(1), dim3(nth), bytes, st, *L, nsweeps, method, ph, exact); }
      else { if (real) hipLaunchKernelGGL((k_relax_reg<4, true, 256>), dim3(1), dim3(nth), bytes, st, *L, nsweeps, method, ph, exact);
             else hipLaunchKernelGGL((k_relax_reg<4, false, 256>), dim3(1), dim3(nth), bytes, st, *L, nsweeps, method, ph, exact); }
      return mgx_launched();
    }
  }
  {  // everything in LDS? (11 arrays of the compact level + the k=1 snapshot)
    static const bool notiny = getenv("MGX_NO_TINY") != nullptr;
    const size_t n3 = (size_t)(L->nx + 2) * (L->ny + 2) * L->nz, bytes = (11 * n3 + (size_t)(L->nx + 2) * (L->ny + 2)) * sizeof(double);
    if (!notiny && !(exact && method == 1 && real) && bytes <= 64 * 1024 && (ph.S && ph.E && ph.N && ph.W) && (L->nz == 2 || L->nz == 4)) {
      const int ncolt = method == 2 ? (L->nx / 2) * (L->ny / 2) : L->nx * (L->ny / 2);
      const int ntht = ncolt <= 64 ? 64 : 256;
      if (L->nz == 2) { if (real) hipLaunchKernelGGL((k_relax_tiny<2, true>), dim3(1), dim3(ntht), bytes, st, *L, nsweeps, method, ph);
                        else hipLaunchKernelGGL((k_relax_tiny<2, false>), dim3(1), dim3(ntht), bytes, st, *L, nsweeps, method, ph); }
      else { if (real) hipLaunchKernelGGL((k_relax_tiny<4, true>), dim3(1), dim3(ntht), bytes, st, *L, nsweeps, method, ph);
             else hipLaunchKernelGGL((k_relax_tiny<4, false>), dim3(1), dim3(ntht), bytes, st, *L, nsweeps, method, ph); }
      return mgx_launched();
    }
  }
  const int ncol = method == 2 ? (L->nx / 2) * (L->ny / 2) : L->nx * (L->ny / 2);
  // (Measured and dropped: one cooperative launch per relax call with a grid-wide barrier between colour passes on the
  // 1024-16384-column levels -- cg grid.sync and a hand-written atomic barrier both cost more than the kernel boundary
  // they replace: V-cycle 2.81 -> 3.03 ms.)
  // one CU streams ~25-50 GB/s: worth it only while the level is launch-bound, not bandwidth-bound (measured:
  // 16x16x2 and 32x32x4 win, 64x64x8 loses 2x against separate launches over 256 CUs).  A 2-level-deep coarsest grid
  // is still launch-bound at 1024 columns per colour (the gathered 64x32x2 grid of an 8-GPU run): 1024 threads.
  if (!(ph.S && ph.E && ph.N && ph.W)) return 0;
  if (L->nz == 2 && ncol > 256 && ncol <= 1024) {
    if (real) hipLaunchKernelGGL((k_relax_small<2, true, 1024>), dim3(1), dim3(1024), 0, st, *L, nsweeps, method, ph, exact);
    else hipLaunchKernelGGL((k_relax_small<2, false, 1024>), dim3(1), dim3(1024), 0, st, *L, nsweeps, method, ph, exact);
    return mgx_launched();
  }
  if (ncol > 256 || L->nz > 8) return 0;
  // 32x32x8 (fifth level of an nz = 128 hierarchy), four colours: two colour-pair launches per sweep (k_relax_ks2, 5.9 us each) beat this
  // kernel's 22-32 us per sweep, which re-reads every operand through L2
  if (L->nz == 8 && method == 2 && L->zy != nullptr && ncol > 64 && L->ny / 2 <= WAVE) return 0;
  const int nth = ncol <= 64 ? 64 : 256;
#define SMALL_CASE(NZV) case NZV: if (real) hipLaunchKernelGGL((k_relax_small<NZV, true, 256>), dim3(1), dim3(nth), 0, st, *L, nsweeps, method, ph, exact); \
                                  else hipLaunchKernelGGL((k_relax_small<NZV, false, 256>), dim3(1), dim3(nth), 0, st, *L, nsweeps, method, ph, exact); return mgx_launched();
  switch (L->nz) { SMALL_CASE(2) SMALL_CASE(4) SMALL_CASE(8) default: return 0; }
#undef SMALL_CASE
}

// returns 1 when the launched kernel also wrote the physical-boundary mirrors of p (no k_halo_phys needed)
int mgxk_relax_ks(hipStream_t, const LevView *, int, int, int, int, int, int, int, Sides);  // mgx_relax_ks.hip
int mgxk_relax_nz128(hipStream_t, const LevView *, int, int, int, int, int, int, int, Sides);  // mgx_relax_tall.hip
// returns bit 0: the kernel stored the physical mirrors itself; bit 1: it wrote L->d0w
int mgxk_relax_colour(hipStream_t st, const LevView *L, int i0, int istep, int nplanes, int jodd_fixed, int rb, int real, int snap, Sides ph) {
  if (const int ks = mgxk_relax_ks(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph)) return ks;  // mid levels: rows split over the waves of a workgroup
  switch (L->nz) {
#ifndef MGX_QUICK  // -DMGX_QUICK: only the nz=64 instantiations (resource-usage checks of the level-1 kernel in seconds)
    case 2: launch_relax_nz<2>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph); return (real && snap && L->d0w != nullptr) ? 3 : 1;
    case 4: launch_relax_nz<4>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph); return (real && snap && L->d0w != nullptr) ? 3 : 1;
    case 8: launch_relax_nz<8>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph); return (real && snap && L->d0w != nullptr) ? 3 : 1;
    case 16: launch_relax_nz<16>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph); return (real && snap && L->d0w != nullptr) ? 3 : 1;
    case 32: launch_relax_nz<32>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph); return (real && snap && L->d0w != nullptr) ? 3 : 1;
#endif
    case 64: launch_relax_nz<64>(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph); return (real && snap && L->d0w != nullptr) ? 3 : 1;
#ifndef MGX_QUICK
    case 128: if (mgxk_relax_nz128(st, L, i0, istep, nplanes, jodd_fixed, rb, real, snap, ph)) return 1; break;
#endif
    default: break;
  }
  dim3 blk(WAVE, 4), grd = col_grid(L->ny / 2, nplanes);
  if (real && snap) hipLaunchKernelGGL((k_relax_colour<true, true>), grd, blk, 0, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph);
  else if (real) hipLaunchKernelGGL((k_relax_colour<true, false>), grd, blk, 0, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph);
  else hipLaunchKernelGGL((k_relax_colour<false, false>), grd, blk, 0, st, *L, i0, istep, nplanes, jodd_fixed, rb, ph);
  return 0;
}
// does mgxk_relax_colour run a register-resident kernel (which writes mirrors and chained snapshots) on this level?
int mgxk_has_reg_kernel(const LevView *L) {
  switch (L->nz) { case 2: case 4: case 8: case 16: case 32: case 64: return 1; case 128: return L->zy != nullptr && getenv("MGX_NO_TALL") == nullptr; default: return 0; }
}
void mgxk_snapshot_k1(hipStream_t st, const LevView *L) {
  hipLaunchKernelGGL(k_snapshot_k1, dim3((L->RS + 255) / 256, L->nx + 2), dim3(256), 0, st, *L);
}

}  // extern "C"
