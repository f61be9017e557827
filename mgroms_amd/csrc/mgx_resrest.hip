// Down leg of the V-cycle in one kernel: residual r = b - A p of the fine level (mg_relax.f90:421-515) and its restriction to
// the coarse right-hand side, b_c = sum of the 8 fine r (mg_intergrids.f90:139-162), without r ever leaving the chip.
//
// Vcycle discards the residual norm (mg_solvers.f90:140) and the r of the down leg is dead: nothing but fine2coarse reads
// it, and coarse2fine overwrites it on the way up (mg_intergrids.f90:387-448).  Separately the two operators issue 18 + 1
// loads per fine cell (every neighbour value fetched again by every lane that needs it), write r and read it back, and
// stream the stored diagonal.  Here one lane owns one fine plane of a COARSE column (two fine columns jA, jB; lanes 0-31 of a
// wave take plane iA, lanes 32-63 plane iB of the same 32 coarse columns), walks it bottom to top with three-row windows in
// registers, shares the pair's p / slope / coefficient values between its two residuals (24 loads per row = 12 per
// cell), rebuilds the diagonal from the couplings (as the smoother does, mgx_relax.hip), and the 8-cell sum is closed with
// the partner lane's two values (one cross-lane exchange per row): no r traffic, no diagonal stream, one launch.
// Measured (rocprofv3, MI355X): 512x512x64 211-224 us against 279 + 36 for the two kernels (round 2: 236-250 with guarded look-ahead);
// levels of up to 256x256x64 cells take k_residual_restrict_flat below (256x256x32: 24 us against 47 + 13).
// The residuals and the 8-term sum use the reference's expressions in the reference's order: b_c is bit-identical to
// compute_residual followed by fine2coarse.  Matrix-free cross terms (needs the slopes zy, zx: the matrix must be the one
// define_matrices built); other cases keep the two separate kernels.
#include <cstdlib>

#include "mgx_device.h"

namespace {

// One lane = one fine plane (iA for lanes 0-31, iB for lanes 32-63 of the wave) of a coarse column: two fine columns jA, jB.
// values of one fine row needed with their k-1 / k+1 neighbours (three-row windows)
struct RowW {
  double P[4];      // p of the own plane at j = jA-1, jA, jB, jB+1
  double Pm[2];     // p of plane i-1 at jA, jB
  double Pp[2];     // p of plane i+1
  double ZY4[4];    // slopes zy of the own plane at jA-1, jA, jB, jB+1 (the inner two with their k-1 / k+1 neighbours)
  double ZX[2];     // own slopes zx at jA, jB
  double A2[2];     // slot 2 (couples k with k-1)
};
// values needed at the row itself only
struct RowR {
  double ZXm[2];    // zx of plane i-1 at jA, jB
  double ZXp[2];    // zx of plane i+1
  double A4[3];     // slot 4 at jA, jB, jB+1
  double A7[2];     // slot 7 own plane at jA, jB
  double A7p[2];    // slot 7 of plane i+1
  double B[2];
};

struct Geo {
  long long o, om, op;  // plane offsets of i, i-1, i+1
  int c[4];             // row positions of jA-1, jA, jB, jB+1
};

// jA-1 and jB sit side by side in the even half-row, jA and jB+1 in the odd one (c[2] = c[0] + 1, c[3] = c[1] + 1): one 16-byte
// load per pair instead of two 8-byte loads whose wave-wide footprints overlap
#define LD_PAIR(ptr, A, B) { double2 t2_; __builtin_memcpy(&t2_, (ptr), 16); A = t2_.x; B = t2_.y; }
__device__ __forceinline__ void load_w(RowW &w, const LevView &F, const Geo &g, const long long ro) {
  const double *__restrict__ p = F.p, *__restrict__ zy = F.zy, *__restrict__ zx = F.zx, *__restrict__ a2 = F.cA[1];
  LD_PAIR(p + g.o + ro + g.c[0], w.P[0], w.P[2])
  LD_PAIR(p + g.o + ro + g.c[1], w.P[1], w.P[3])
  LD_PAIR(zy + g.o + ro + g.c[0], w.ZY4[0], w.ZY4[2])
  LD_PAIR(zy + g.o + ro + g.c[1], w.ZY4[1], w.ZY4[3])
#pragma unroll
  for (int jj = 0; jj < 2; jj++) {
    const long long e = g.o + ro + g.c[jj + 1];
    w.Pm[jj] = p[g.om + ro + g.c[jj + 1]]; w.Pp[jj] = p[g.op + ro + g.c[jj + 1]];
    w.ZX[jj] = zx[e]; w.A2[jj] = a2[e];
  }
}

__device__ __forceinline__ void load_r(RowR &r, const LevView &F, const Geo &g, const long long ro) {
  const double *__restrict__ b = F.b, *__restrict__ zx = F.zx, *__restrict__ a4 = F.cA[3], *__restrict__ a7 = F.cA[6];
  LD_PAIR(a4 + g.o + ro + g.c[1], r.A4[0], r.A4[2])
  r.A4[1] = a4[g.o + ro + g.c[2]];
#pragma unroll
  for (int jj = 0; jj < 2; jj++) {
    const int c = g.c[jj + 1];
    r.ZXm[jj] = zx[g.om + ro + c]; r.ZXp[jj] = zx[g.op + ro + c];
    r.A7[jj] = a7[g.o + ro + c]; r.A7p[jj] = a7[g.op + ro + c];
    r.B[jj] = b[g.o + ro + c];
  }
}

}  // namespace

// grid: 1-D, gx j-chunks of 32 coarse columns x gy groups of blockDim.y coarse planes, XCD-aware as k_relax_nz.
// One wave per SIMD (AW = AR = 1: four window rows + two row buffers, ~450 registers with the requests kept ahead of their use;
// squeezed to 256 registers for two waves per SIMD it spills and loses: 250 vs 236 us at 512x512x64 in round 2).
// NORM: also the sum of r^2 over the block's fine cells -> partial[blockIdx.x] (the closing compute_residual of a solve_p iteration,
// mg_solvers.f90:65, whose r the next Fcycle restricts first thing, :112-115: one pass over the level instead of two and no r written).
// dup: a second destination of the coarse sums (Fcycle's grid(lev+1)%r = grid(lev+1)%b, :113).
template <bool REAL, int AW, int AR, bool NORM = false>
__global__ __launch_bounds__(256, 1) void k_residual_restrict(LevView F, LevView C, double *__restrict__ dst, Sides ph, double *__restrict__ zero, int gx, int gy,
                                                             double *__restrict__ partial = nullptr, double *__restrict__ dup = nullptr) {
  __shared__ double red_ss[4];
  double ss = 0.0;
  int bx, by;
  if ((gy & 7) == 0) { const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3; by = xcd * (gy >> 3) + local / gx; bx = local - (local / gx) * gx; }
  else { by = blockIdx.x / gx; bx = blockIdx.x - by * gx; }
  const int half = threadIdx.x >> 5;                     // 0: fine plane iA = 2 i2 - 1, 1: iB = 2 i2
  int j2 = 1 + bx * 32 + (threadIdx.x & 31);
  const int i2 = 1 + by * blockDim.y + threadIdx.y;
  if (i2 > C.nx) {                                       // wave-uniform
    if (NORM) { if ((threadIdx.x & 63) == 0) red_ss[threadIdx.y] = 0.0; __syncthreads(); if (threadIdx.x == 0 && threadIdx.y == 0) partial[blockIdx.x] = red_ss[0] + red_ss[1] + red_ss[2] + red_ss[3]; }
    return;
  }
  const bool live = j2 <= C.ny;
  if (!live) j2 = C.ny;                                  // ragged chunk: dead lanes shadow a live column (they take part in the shuffles), store nothing
  const int nz = F.nz;
  const long long RS = F.RS;
  Geo g;
  g.o = (long long)(2 * i2 - 1 + half) * F.plane; g.om = g.o - F.plane; g.op = g.o + F.plane;
  g.c[1] = F.HO + (j2 - 1); g.c[2] = F.EO + j2; g.c[0] = F.EO + (j2 - 1); g.c[3] = F.HO + j2;  // jA = 2 j2 - 1 (odd), jB = 2 j2 (even)
  const double qrt = 0.25;
  const double *__restrict__ a1 = F.cA[0], *__restrict__ a5 = F.cA[4], *__restrict__ a8 = F.cA[7];

  // last row: stored diagonal (its formula differs, mg_define_matrix.f90:642-654)
  double dlast[2];
#pragma unroll
  for (int jj = 0; jj < 2; jj++) dlast[jj] = a1[g.o + (long long)(nz - 1) * RS + g.c[jj + 1]];
  // first row: stored diagonal, the k = 1 diagonal slots and the four corner values of p (horizontal diagonals of cmatrix = 'real',
  // mg_relax.f90:475-479) -- requested before the rows, so that the first step does not wait for the look-ahead behind them
  double dedge[2], e[2][8];
#pragma unroll
  for (int jj = 0; jj < 2; jj++) {
    const int c = g.c[jj + 1], jm = g.c[jj], jp = g.c[jj + 2];
    dedge[jj] = a1[g.o + c];
    if (REAL) {
      e[jj][0] = a5[g.o + c]; e[jj][1] = F.p[g.om + jp]; e[jj][2] = a5[g.op + jm]; e[jj][3] = F.p[g.op + jm];
      e[jj][4] = a8[g.o + c]; e[jj][5] = F.p[g.om + jm]; e[jj][6] = a8[g.op + jp]; e[jj][7] = F.p[g.op + jp];
    }
  }
  // Look-ahead: the window rows are requested AW steps and the rows' own values AR steps before their first use (A/B of the depths,
  // scripts/probe/ab_resrest_ahead.sh: (1,1) 224 us, (1,2) 233, (2,1) 225 -- once the requests are unconditional, below, the depth is
  // not what bounds the kernel).
  constexpr int NWB = 3 + AW, NRB = 1 + AR;  // buffers: rows k-1, k, k+1 + AW ahead; row k + AR ahead
  constexpr int U = (NWB % NRB == 0) ? NWB : NWB * NRB;  // steps after which both rotations are back where they started
  static_assert(U <= 12, "unroll");
  RowW W[NWB];
  RowR R[NRB];
  // before the peeled k = 1 step: W[0 .. AW+1] = rows 1 .. AW+2, R[NRB-1] = row 1, R[0 .. AR-2] = rows 2 .. AR.
  // Every request is UNCONDITIONAL (rows past the top are clamped to nz and never used): a request inside a branch makes the number of
  // outstanding loads path-dependent, the compiler then waits for vmcnt(0) at every step and the look-ahead is void (measured: one memory
  // round trip per row, whatever AW / AR).
#pragma unroll
  for (int q = 0; q < AW + 2; q++) load_w(W[q], F, g, (long long)(q + 1 <= nz ? q : nz - 1) * RS);
  load_r(R[NRB - 1], F, g, 0);
#pragma unroll
  for (int q = 0; q < AR - 1; q++) load_r(R[q], F, g, (long long)(q + 2 <= nz ? q + 1 : nz - 1) * RS);
  double z = 0.0;
  const long long oc = (long long)i2 * C.plane + jpos(C, j2);

  // one fine row k: Wm / W0 / Wp hold rows k-1, k, k+1; Wn receives row k+1+AW and Rn row k+AR while row k is computed
#define RR_LOADS(k, Wn, Rn)                                                                                                 \
    load_w(Wn, F, g, (long long)((k) + 1 + AW <= nz ? (k) + AW : nz - 1) * RS);                                              \
    load_r(Rn, F, g, (long long)((k) + AR <= nz ? (k) + AR - 1 : nz - 1) * RS);
#define RR_CELL_IN(Wm, W0, Wp, R0)                                                                                          \
      const double pc_m = Wm.P[jj + 1], pc_0 = W0.P[jj + 1], pc_p = Wp.P[jj + 1];                                            \
      const double pjm_m = Wm.P[jj], pjm_0 = W0.P[jj], pjm_p = Wp.P[jj];                                                     \
      const double pjp_m = Wm.P[jj + 2], pjp_0 = W0.P[jj + 2], pjp_p = Wp.P[jj + 2];                                         \
      const double pim_m = Wm.Pm[jj], pim_0 = W0.Pm[jj], pim_p = Wp.Pm[jj];                                                  \
      const double pip_m = Wm.Pp[jj], pip_0 = W0.Pp[jj], pip_p = Wp.Pp[jj];                                                  \
      const double zy_m = Wm.ZY4[jj + 1], zy_p = Wp.ZY4[jj + 1], zx_m = Wm.ZX[jj], zx_p = Wp.ZX[jj];                         \
      const double zyjm = W0.ZY4[jj], zyjp = W0.ZY4[jj + 2];                                                                \
      const double zxim = R0.ZXm[jj], zxip = R0.ZXp[jj];                                                                    \
      const double a4o = R0.A4[jj], a4jp = R0.A4[jj + 1], a7o = R0.A7[jj], a7ip = R0.A7p[jj];                                \
      const double a2_0 = W0.A2[jj], a2_p = Wp.A2[jj];                                                                      \
      const double c3 = qrt * (zy_p + zyjm), c3m = qrt * (zyjp + zy_m), c5 = -qrt * (zy_m + zyjm), c5m = -qrt * (zyjp + zy_p); \
      const double c6 = qrt * (zx_p + zxim), c6m = qrt * (zxip + zx_m), c8 = -qrt * (zx_m + zxim), c8m = -qrt * (zxip + zx_p); \
      (void)pc_m; (void)pjm_m; (void)pjp_m; (void)pim_m; (void)pip_m; (void)c3m; (void)c5; (void)c6m; (void)c8; (void)a2_0;
  /* fine2coarse_3D (mg_intergrids.f90:149-160): (k,jA,iA) + (k,jA,iB) + (k,jB,iA) + (k,jB,iB), then the same of k+1;
     the iB values come from the partner lane (lane ^ 32) */
#define RR_SUM(k, r)                                                                                                        \
    if (NORM && live) ss = ss + (r[0] * r[0] + r[1] * r[1]);                                                                \
    const double rA_B = __shfl_xor(r[0], 32, 64), rB_B = __shfl_xor(r[1], 32, 64);                                          \
    if ((k) & 1) z = r[0] + rA_B + r[1] + rB_B;                                                                             \
    else {                                                                                                                  \
      z = z + r[0] + rA_B + r[1] + rB_B;                                                                                    \
      if (half == 0 && live) {                                                                                              \
        const long long rc = (long long)(((k) >> 1) - 1) * C.RS;                                                            \
        dst[oc + rc] = z;                                                                                                   \
        mirror_store(C, dst, rc, j2, i2, jpos(C, j2), z, ph);                                                               \
        if (dup) { dup[oc + rc] = z; mirror_store(C, dup, rc, j2, i2, jpos(C, j2), z, ph); }                                 \
        if (zero) { zero[oc + rc] = 0.0; mirror_store(C, zero, rc, j2, i2, jpos(C, j2), 0.0, ph); }                          \
      }                                                                                                                     \
    }
  /* interior rows, mg_relax.f90:484-496; diagonal = minus the sum of the fourteen couplings (mg_define_matrix.f90:632-639) */
#define RR_GENERAL(rr, R0)                                                                                                  \
      {                                                                                                                     \
        const double dk = -a2_0 - a2_p - a4o - a4jp - a7o - a7ip - c6 - c6m - c8 - c8m - c3 - c3m - c5 - c5m;               \
        rr = R0.B[jj] - dk * pc_0 - a2_0 * pc_m - a2_p * pc_p - c3 * pjm_p - c3m * pjp_m                                    \
                      - a4o * pjm_0 - a4jp * pjp_0 - c5 * pjm_m - c5m * pjp_p                                                \
                      - c6 * pim_p - c6m * pip_m - a7o * pim_0 - a7ip * pip_0                                                \
                      - c8 * pim_m - c8m * pip_p;                                                                            \
      }
  /* last row, :498-509 (stored diagonal) */
#define RR_LAST(rr, R0)                                                                                                     \
      {                                                                                                                     \
        rr = R0.B[jj] - dlast[jj] * pc_0 - a2_0 * pc_m - c3m * pjp_m - a4o * pjm_0 - a4jp * pjp_0                           \
                      - c5 * pjm_m - c6m * pip_m - a7o * pim_0 - a7ip * pip_0 - c8 * pim_m;                                  \
      }
#define RR_STEP(k, Wm, W0, Wp, Wn, R0, Rn)                                                                                  \
  {                                                                                                                         \
    RR_LOADS(k, Wn, Rn)                                                                                                     \
    double r[2];                                                                                                            \
    _Pragma("unroll") for (int jj = 0; jj < 2; jj++) {                                                                     \
      RR_CELL_IN(Wm, W0, Wp, R0)                                                                                            \
      double rr;                                                                                                            \
      if ((k) < nz) RR_GENERAL(rr, R0)                                                                                       \
      else RR_LAST(rr, R0)                                                                                                  \
      r[jj] = rr;                                                                                                           \
    }                                                                                                                       \
    RR_SUM(k, r)                                                                                                            \
  }
  {  // k = 1 (mg_relax.f90:464-482), peeled
    RR_LOADS(1, W[AW + 2], R[AR - 1])
    double r[2];
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      RR_CELL_IN(W[0], W[0], W[1], R[NRB - 1])
      double rr = R[NRB - 1].B[jj] - dedge[jj] * pc_0 - a2_p * pc_p - c3 * pjm_p - a4o * pjm_0 - a4jp * pjp_0
                                   - c5m * pjp_p - c6 * pim_p - a7o * pim_0 - a7ip * pip_0 - c8m * pip_p;
      if (REAL) rr = rr - e[jj][0] * e[jj][1] - e[jj][2] * e[jj][3] - e[jj][4] * e[jj][5] - e[jj][6] * e[jj][7];
      r[jj] = rr;
    }
    RR_SUM(1, r)
  }
  // rows 2 .. nz, U steps per trip: at step q of a trip rows k-1, k, k+1 sit in W[q], W[q+1], W[q+2] (mod NWB) and row k's own values in
  // R[q mod NRB].  Whole trips run without a branch around a step; the last nz - 1 mod U rows follow, guarded (the rotation is back at 0).
  int k = 2;
  for (; k + U - 1 <= nz; k += U) {
#pragma unroll
    for (int q = 0; q < U; q++) RR_STEP(k + q, W[q % NWB], W[(q + 1) % NWB], W[(q + 2) % NWB], W[(q + 2 + AW) % NWB], R[q % NRB], R[(q + AR) % NRB])
  }
#pragma unroll
  for (int q = 0; q < U - 1; q++) {
    if (k + q <= nz) RR_STEP(k + q, W[q % NWB], W[(q + 1) % NWB], W[(q + 2) % NWB], W[(q + 2 + AW) % NWB], R[q % NRB], R[(q + AR) % NRB])
  }
#undef RR_STEP
#undef RR_LOADS
  if (NORM) {  // wave sum (lanes in index order pairs), then the block's four waves in order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((threadIdx.x & 63) == 0) red_ss[threadIdx.y] = ss;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) partial[blockIdx.x] = red_ss[0] + red_ss[1] + red_ss[2] + red_ss[3];
  }
}

// The same operator without the walk.  A lane that climbs its column pays one memory round trip per row step (20 us for nz = 16 and 13 us
// for nz = 8 on 8-32 CUs, measured), but nothing in r = b - A p is sequential in k.  Here one wave owns S fine rows (S/2 coarse rows) of 32
// coarse columns (both fine planes, as above): the S + 2 window rows and the S rows' own values are all requested at once, the rows are
// computed and the 8-cell sums closed with the partner lane -- one round trip, nz/S times the waves.  Same expressions (the macros above),
// same order: bit-identical.
template <bool REAL, int S, bool NORM = false>
__global__ __launch_bounds__(64) void k_residual_restrict_flat(LevView F, LevView C, double *__restrict__ dst, Sides ph, double *__restrict__ zero, int gx,
                                                              double *__restrict__ partial = nullptr, double *__restrict__ dup = nullptr) {
  double ss = 0.0;
  // 1-D grid, k fastest: the nz/S waves of one (plane, j-chunk) share two of their window rows with the wave above and below, so they
  // are kept together in time and on ONE XCD (workgroups are dealt to the eight XCDs round-robin), where that re-use is an L2 hit
  const int nz = F.nz, nks = (nz + S - 1) / S, ng = gx * C.nx;
  int grp, ks;
  if ((ng & 7) == 0) { const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3; ks = local % nks; grp = xcd * (ng >> 3) + local / nks; }
  else { ks = blockIdx.x % nks; grp = blockIdx.x / nks; }
  const int i2 = 1 + grp / gx, bx = grp - (grp / gx) * gx;
  const int half = threadIdx.x >> 5;
  int j2 = 1 + bx * 32 + (threadIdx.x & 31);
  const bool live = j2 <= C.ny;
  if (!live) j2 = C.ny;
  const int k = ks * S + 1;  // fine rows k (odd) .. k + S - 1 (those <= nz; nz is even)
  const long long RS = F.RS;
  Geo g;
  g.o = (long long)(2 * i2 - 1 + half) * F.plane; g.om = g.o - F.plane; g.op = g.o + F.plane;
  g.c[1] = F.HO + (j2 - 1); g.c[2] = F.EO + j2; g.c[0] = F.EO + (j2 - 1); g.c[3] = F.HO + j2;
  const double qrt = 0.25;
  const double *__restrict__ a1 = F.cA[0], *__restrict__ a5 = F.cA[4], *__restrict__ a8 = F.cA[7];
  RowW W[S + 2];  // rows k-1 .. k+S, clamped to 1 .. nz (the clamped copies are never used: the first and the last row have their own expressions)
  RowR R[S];
#pragma unroll
  for (int r = 0; r < S + 2; r++) {
    const int kr = k - 1 + r < 1 ? 1 : (k - 1 + r > nz ? nz : k - 1 + r);
    load_w(W[r], F, g, (long long)(kr - 1) * RS);
  }
#pragma unroll
  for (int r = 0; r < S; r++) {
    const int kr = k + r > nz ? nz : k + r;
    load_r(R[r], F, g, (long long)(kr - 1) * RS);
  }
  double dedge[2] = {0.0, 0.0}, e[2][8];
  if (k == 1) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      const int c = g.c[jj + 1], jm = g.c[jj], jp = g.c[jj + 2];
      dedge[jj] = a1[g.o + c];
      if (REAL) {
        e[jj][0] = a5[g.o + c]; e[jj][1] = F.p[g.om + jp]; e[jj][2] = a5[g.op + jm]; e[jj][3] = F.p[g.op + jm];
        e[jj][4] = a8[g.o + c]; e[jj][5] = F.p[g.om + jm]; e[jj][6] = a8[g.op + jp]; e[jj][7] = F.p[g.op + jp];
      }
    }
  }
  double dlast[2] = {0.0, 0.0};
  if (k + S > nz) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++) dlast[jj] = a1[g.o + (long long)(nz - 1) * RS + g.c[jj + 1]];
  }
  double z = 0.0;
  const long long oc = (long long)i2 * C.plane + jpos(C, j2);
#pragma unroll
  for (int r = 0; r < S; r++) {
    const int kr = k + r;
    if (kr <= nz) {  // wave-uniform
      double rv[2];
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
        RR_CELL_IN(W[r], W[r + 1], W[r + 2], R[r])
        double rr;
        if (r == 0 && kr == 1) {
          rr = R[r].B[jj] - dedge[jj] * pc_0 - a2_p * pc_p - c3 * pjm_p - a4o * pjm_0 - a4jp * pjp_0
                          - c5m * pjp_p - c6 * pim_p - a7o * pim_0 - a7ip * pip_0 - c8m * pip_p;
          if (REAL) rr = rr - e[jj][0] * e[jj][1] - e[jj][2] * e[jj][3] - e[jj][4] * e[jj][5] - e[jj][6] * e[jj][7];
        } else if (kr < nz) RR_GENERAL(rr, R[r])
        else RR_LAST(rr, R[r])
        rv[jj] = rr;
      }
      RR_SUM(kr, rv)  /* odd row: z = its four values; even row: z += its four values, store coarse row kr/2 */
    }
  }
  if (NORM) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (threadIdx.x == 0) partial[blockIdx.x] = ss;
  }
}
#undef RR_SUM
#undef RR_CELL_IN
#undef RR_GENERAL
#undef RR_LAST

extern "C" {

// returns 1 when launched (matrix-free slopes present, level large enough to be bandwidth-bound), 0 = use mgxk_residual + mgxk_fine2coarse
// partial != nullptr: also sum r^2 (partials, one per workgroup; *npartial_out = their number); dup: second destination of the coarse sums
int mgxk_residual_restrict_grid(const LevView *F, const LevView *C) {  // workgroups (= norm partials) of the launch below
  static const long long flatmax = getenv("MGX_RESREST_FLAT_MAX") ? atoll(getenv("MGX_RESREST_FLAT_MAX")) : 256LL * 256 * 64;
  const int gx = (C->ny + 31) / 32;
  if ((long long)F->nx * F->ny * F->nz <= flatmax) {
    static const int senv = getenv("MGX_RESREST_FLAT_S") ? atoi(getenv("MGX_RESREST_FLAT_S")) : 0;
    const int Sv = (senv >= 4 && F->nz >= 4) ? 4 : 2;
    return gx * C->nx * ((F->nz + Sv - 1) / Sv);
  }
  return gx * ((C->nx + 3) / 4);
}
int mgxk_residual_restrict_ex(hipStream_t st, const LevView *F, const LevView *C, double *dst, int real, Sides ph, double *zero, double *partial, double *dup) {
  mgx_before_launch();
  static const bool off = getenv("MGX_NO_RESREST") != nullptr;
  static const long long mincells = getenv("MGX_RESREST_MIN") ? atoll(getenv("MGX_RESREST_MIN")) : 0;
  if (off || F->zy == nullptr || F->nz < 2 || (F->nz & 1)) return 0;
  if (C->nx * 2 != F->nx || C->ny * 2 != F->ny) return 0;
  if ((long long)F->nx * F->ny * F->nz < mincells) return 0;
  static const long long flatmax = getenv("MGX_RESREST_FLAT_MAX") ? atoll(getenv("MGX_RESREST_FLAT_MAX")) : 256LL * 256 * 64;
  if ((long long)F->nx * F->ny * F->nz <= flatmax) {  // no walk: one wave per S fine rows, one round trip
    static const int senv = getenv("MGX_RESREST_FLAT_S") ? atoi(getenv("MGX_RESREST_FLAT_S")) : 0;
    const int gx = (C->ny + 31) / 32, Sr = senv ? senv : 2;
    dim3 blk(WAVE);
#define RRF(REALV, SV, NV) hipLaunchKernelGGL((k_residual_restrict_flat<REALV, SV, NV>), dim3((unsigned)gx * C->nx * ((F->nz + SV - 1) / SV)), blk, 0, st, *F, *C, dst, ph, zero, gx, partial, dup)
#define RRF2(SV) { if (partial) { if (real) RRF(true, SV, true); else RRF(false, SV, true); } else { if (real) RRF(true, SV, false); else RRF(false, SV, false); } }
    if (Sr >= 4 && F->nz >= 4) RRF2(4) else RRF2(2)
#undef RRF2
#undef RRF
    return mgx_launched();
  }
  const int by = 4, gx = (C->ny + 31) / 32, gy = (C->nx + by - 1) / by;
  dim3 blk(WAVE, by), grd(gx * gy);
  static const int deep = getenv("MGX_RESREST_AHEAD") ? atoi(getenv("MGX_RESREST_AHEAD")) : 11;  // 10 * AW + AR (A/B: scripts/probe/ab_resrest_ahead.sh -- 11, 12, 21 within 5 % of each other once the requests are unconditional)
#define RRW(REALV, AWV, ARV, NV) hipLaunchKernelGGL((k_residual_restrict<REALV, AWV, ARV, NV>), grd, blk, 0, st, *F, *C, dst, ph, zero, gx, gy, partial, dup)
#define RRW2(AWV, ARV) { if (partial) { if (real) RRW(true, AWV, ARV, true); else RRW(false, AWV, ARV, true); } else { if (real) RRW(true, AWV, ARV, false); else RRW(false, AWV, ARV, false); } }
  switch (deep) {
    case 12: RRW2(1, 2) break;
    case 21: RRW2(2, 1) break;
    default: RRW2(1, 1) break;
  }
#undef RRW2
#undef RRW
  return mgx_launched();
}
int mgxk_residual_restrict(hipStream_t st, const LevView *F, const LevView *C, double *dst, int real, Sides ph, double *zero) {
  return mgxk_residual_restrict_ex(st, F, C, dst, real, ph, zero, nullptr, nullptr);
}

}  // extern "C"
