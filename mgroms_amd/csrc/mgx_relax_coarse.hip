// Coarsest-level solve, relax(nlevs, ns_coarsest = 40) of mg_solvers.f90:117,144, as ONE wave.
//
// 40 sweeps of a 16x16x2 grid are 160 dependent colour passes.  k_relax_reg (mgx_relax.hip) runs them in one workgroup with a
// thread per column: in a four-colour pass a quarter of the threads work, three waves idle at the barrier, 0.56 us per pass.
// Here a level of <= 256 columns is ONE wave: every lane owns a 2x2 block of columns, i.e. one column of each of the four
// colours (two of each red-black colour), so all 64 lanes work in every pass and the passes are separated by a
// single-wave barrier only.  The level above it (32x32x4, 1024 columns) runs the same way on four waves of one workgroup
// (k_relax_small re-read every operand through L2 there: 4.5 us per pass).  Everything that does not change between passes -- b, the own slots and pivots, the slots of the
// j+1 / i+1 neighbours, gam -- sits in registers (4 columns x NZ rows x 16 values); p lives in LDS with its mirrored halo.
// The level is closed (every side physical), so every halo cell is the image of an interior cell at all times (mg_mpi_exchange.f90:509-537,
// corners :552-597): p sits in LDS WITHOUT a halo and a column reads its j-1 / j+1 / i-1 / i+1 neighbours and the four k=1 diagonals
// through indices clamped to the interior -- no mirror stores, no branches in a pass (they were a third of its instructions); the halo of
// the global array is rebuilt once, when p is written back.
// Same expressions in the same order as relax_col_nz / k_relax_reg: bit-identical.
#include <cstdlib>

#include "mgx_device.h"

// flags (round 3; the level below the coarsest one in a cycle): 1 = coarse2fine(lev) first -- p += the tri-linear interpolation of the
// coarse p of C (mg_intergrids.f90:366-450 + :226), applied while p is loaded into LDS; 2 = compute_residual(lev) + fine2coarse(lev)
// afterwards (mg_relax.f90:421-515, mg_intergrids.f90:139-162, p_c = 0 :70) from the coefficients the lane holds anyway -- a lane's 2x2
// block of columns is exactly one coarse column, so the 8-cell sum closes inside the lane.  Two launches less per level visit; the same
// expressions in the same order as k_coarse2fine, k_residual and k_fine2coarse (the stored diagonal; the matrix-free kernels rebuild
// the same bits).
// FZ: the instance that can fold the transfers in (instantiated for nz = 4 only: the plain kernels keep their register budget)
// SEQ: red-black in the reference's sequential order (the walk below; REAL only) -- its own instances, so that the others keep their registers
struct VecD4 { double v[4]; };
template <int NZ, bool REAL, int NT, bool FZ = false, bool SEQ = false>
__global__ __launch_bounds__(NT, 1) void k_relax_wave(LevView G, int nsweeps, int method, Sides ph, LevView C, int flags, long long bstride) {
  constexpr bool seq = SEQ && REAL;
  // a batch of independent solves with the level's matrix (mgxk_coarse_direct_build): workgroup q works on its own p and b
  if (bstride) { G.p += (long long)blockIdx.x * bstride; G.b += (long long)blockIdx.x * bstride; }
  extern __shared__ double ldsw[];
  const int nx = G.nx, ny = G.ny, W = ny, PL = nx * ny;  // P[k][i-1][j-1], interior only
  double *__restrict__ P = ldsw, *__restrict__ P1 = ldsw + NZ * PL;  // P1: k=1 snapshot of the parallel red-black pass
  // seq (the walk below): Q4 = per column the quad {d0 (then s) of the colour in work, cA(5,1), cA(8,1), g(1)} -- one 32-byte read per
  // plane step -- plus one all-zero quad (index PL) for the lanes past the half-row; GK = g = T^-1 e1 of every column (set once)
  double *__restrict__ Q4 = ldsw + (NZ + 1) * PL, *__restrict__ GK = Q4 + 4 * (PL + 1);
  // register walk (one-wave instance, nz = 2, at most RWN planes: the coarsest level of the 512x512x64 hierarchy, 16x16x2): the three constants
  // of a plane step stay in registers for the whole call and d0 / s cross between the column owners and the walk lanes through DS, one row
  // per walk lane (DS[jh * RWP + i - 1]): 16-byte LDS accesses, two planes each, instead of a 32-byte quad per plane and an 8-byte store
  constexpr int RWN = 16, RWP = RWN + 2;
  constexpr bool rwc = seq && NZ == 2 && NT == 64;
  double *__restrict__ DS = GK + NZ * PL;
  const bool rw = rwc && nx <= RWN;
  const int lane = threadIdx.x;
  // 32-bit element offsets (these levels have a few thousand cells): base pointer in scalar registers + one 32-bit byte offset per load
  // (the 64-bit index arithmetic of ~200 loads per lane and the divisions of a flat cell index were most of this kernel's fixed cost)
  const int pl32 = (int)G.plane;
#define GI(k0, jj, ii) ((ii) * pl32 + (k0) * G.RS + jpos(G, jj))
#define LD32(arr, idx) (*(const double *)((const char *)(arr) + (unsigned)((idx) << 3)))
  {  // p -> LDS: at most 4 cells per lane and level row (<= 256 columns per wave), all loads in flight before the first LDS store
    double tp[4 * NZ];
    int cell[4];
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int r = lane + v * NT;
      const int i = r / W, j = r - i * W;
      cell[v] = r < PL ? GI(0, j + 1, i + 1) : -1;
    }
#pragma unroll
    for (int k0 = 0; k0 < NZ; k0++)
#pragma unroll
      for (int v = 0; v < 4; v++) tp[k0 * 4 + v] = cell[v] >= 0 ? LD32(G.p, cell[v] + k0 * G.RS) : 0.0;
    if (FZ && (flags & 1)) {  // prolongation + correction: the fine cell (k,j,i) takes the coarse cell (k2,j2,i2) it lies in, that cell's neighbours
                      // towards the fine cell's side (j-1 / i-1 for odd j / i, else j+1 / i+1) and the same four of level k2-1 (odd k) / k2+1
      const double wa = 9. / 16., wb = 3. / 16., wc = 1. / 16., wd = 27. / 64., we = 9. / 64., wf = 3. / 64., wg = 1. / 64.;
      const int cpl = (int)C.plane;
#pragma unroll
      for (int v = 0; v < 4; v++) {
        if (cell[v] < 0) continue;
        const int r = lane + v * NT;
        const int i = r / W + 1, j = r - (i - 1) * W + 1;
        const int i2 = (i + 1) >> 1, j2 = (j + 1) >> 1, si = (i & 1) ? -1 : 1, sj = (j & 1) ? -1 : 1;
        const int c00 = i2 * cpl + jpos(C, j2), cj_ = i2 * cpl + jpos(C, j2 + sj), ci_ = (i2 + si) * cpl + jpos(C, j2), cd_ = (i2 + si) * cpl + jpos(C, j2 + sj);
#pragma unroll
        for (int k0 = 0; k0 < NZ; k0++) {
          const int k = k0 + 1, k2 = (k + 1) >> 1, ro = (k2 - 1) * C.RS;
          const double x00 = LD32(C.p, c00 + ro), xd = LD32(C.p, cd_ + ro), xjn = LD32(C.p, cj_ + ro), xin = LD32(C.p, ci_ + ro);
          double val;
          if (k == 1) val = +wa * x00 + wc * xd + wb * xjn + wb * xin;
          else if (k == NZ) val = 0.5 * (wa * x00 + wc * xd + wb * xjn + wb * xin);
          else {
            const int kp = k2 - ((k % 2) * 2 - 1), rp = (kp - 1) * C.RS;
            const double y00 = LD32(C.p, c00 + rp), yd = LD32(C.p, cd_ + rp), yjn = LD32(C.p, cj_ + rp), yin = LD32(C.p, ci_ + rp);
            val = +wd * x00 + wf * xd + we * xjn + we * xin + we * y00 + wg * yd + wf * yjn + wf * yin;
          }
          tp[k0 * 4 + v] = tp[k0 * 4 + v] + val;
        }
      }
    }
#pragma unroll
    for (int k0 = 0; k0 < NZ; k0++)
#pragma unroll
      for (int v = 0; v < 4; v++) if (cell[v] >= 0) P[k0 * PL + lane + v * NT] = tp[k0 * 4 + v];
  }
  const int nbj = ny >> 1;
  const bool mine = lane < (nx >> 1) * nbj;
  const int bi = lane / nbj, bj = lane - bi * nbj;
  // column q of the lane: (i,j) = (2 bi + 1 + (q >> 1), 2 bj + 1 + (q & 1)); q = 0..3 are the four colours in the reference's order.
  // The symmetric storage makes a column read slots 3,4,5 of its j+1 neighbour and 6,7,8 of its i+1 neighbour: inside the block those
  // are the lane's own registers (q -> q+1, q -> q+2), only the columns j+2 / i+2 beyond the block are held separately (xj, xi).
  double ob[4][NZ], a2[4][NZ], a3[4][NZ], a4[4][NZ], a5[4][NZ], a6[4][NZ], a7[4][NZ], a8[4][NZ], bet[4][NZ];
  double xj[2][3][NZ], xi[2][3][NZ], e2[4], e4[4];
  if (mine) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int i = 2 * bi + 1 + (q >> 1), j = 2 * bj + 1 + (q & 1);
      const int c0 = GI(0, j, i), cj0 = GI(0, j + 1, i), ci0 = c0 + pl32;
#pragma unroll
      for (int k = 0; k < NZ; k++) {
        const int c = c0 + k * G.RS;
        ob[q][k] = LD32(G.b, c); a2[q][k] = LD32(G.cA[1], c); a3[q][k] = LD32(G.cA[2], c); a4[q][k] = LD32(G.cA[3], c); a5[q][k] = LD32(G.cA[4], c);
        a6[q][k] = LD32(G.cA[5], c); a7[q][k] = LD32(G.cA[6], c); a8[q][k] = LD32(G.cA[7], c); bet[q][k] = LD32(G.bet, c);
        if (q & 1) { const int cj = cj0 + k * G.RS; xj[q >> 1][0][k] = LD32(G.cA[2], cj); xj[q >> 1][1][k] = LD32(G.cA[3], cj); xj[q >> 1][2][k] = LD32(G.cA[4], cj); }
        if (q >> 1) { const int ci = ci0 + k * G.RS; xi[q & 1][0][k] = LD32(G.cA[5], ci); xi[q & 1][1][k] = LD32(G.cA[6], ci); xi[q & 1][2][k] = LD32(G.cA[7], ci); }
      }
      e2[q] = e4[q] = 0.0;
      if (REAL) { e2[q] = LD32(G.cA[4], GI(0, j - 1, i + 1)); e4[q] = LD32(G.cA[7], GI(0, j + 1, i + 1)); }
    }
  }
  // seq (red-black in the reference's sequential order, mgx_rbseq.hip): g = T^-1 e1 of the lane's four columns, by tridiag's recurrences, into LDS
  if (seq && mine) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int oc = (2 * bi + (q >> 1)) * W + 2 * bj + (q & 1);
      double gq[NZ];
      double x = bet[q][0];
      gq[0] = x;
#pragma unroll
      for (int k = 1; k < NZ; k++) { x = (0.0 - a2[q][k] * x) * bet[q][k]; gq[k] = x; }
#pragma unroll
      for (int k = NZ - 2; k >= 0; k--) gq[k] = gq[k] - (a2[q][k + 1] * bet[q][k]) * gq[k + 1];
#pragma unroll
      for (int k = 0; k < NZ; k++) GK[k * PL + oc] = gq[k];
      Q4[4 * oc + 1] = a5[q][0]; Q4[4 * oc + 2] = a8[q][0]; Q4[4 * oc + 3] = gq[0];
    }
  }
  if (seq && threadIdx.x < 4) Q4[4 * PL + threadIdx.x] = 0.0;
  const bool walk_on = REAL && seq && method == 1;
  __syncthreads();
  // register walk: (cA5, cA8, g1) of the lane's column in every plane, for either colour (planes past nx and lanes past the half-row: zeros -> u = 0)
  double k5[rwc ? 2 : 1][rwc ? RWN : 1], k8[rwc ? 2 : 1][rwc ? RWN : 1], kg[rwc ? 2 : 1][rwc ? RWN : 1];
  if (rwc && rw && walk_on) {
#pragma unroll
    for (int c_ = 0; c_ < 2; c_++)
#pragma unroll
      for (int d_ = 0; d_ < RWN; d_++) {
        const bool jo_ = ((d_ & 1) == 0) == (c_ == 0);   // plane d_ + 1 holds the colour's odd j
        const int oc_ = (lane < (ny >> 1) && d_ < nx) ? d_ * W + 2 * lane + (jo_ ? 0 : 1) : PL;
        k5[c_][d_] = Q4[4 * oc_ + 1]; k8[c_][d_] = Q4[4 * oc_ + 2]; kg[c_][d_] = Q4[4 * oc_ + 3];
      }
  }
#define R3(q, k) (((q) & 1) ? xj[(q) >> 1][0][k] : a3[((q) + 1) & 3][k])
#define R4(q, k) (((q) & 1) ? xj[(q) >> 1][1][k] : a4[((q) + 1) & 3][k])
#define R5(q, k) (((q) & 1) ? xj[(q) >> 1][2][k] : a5[((q) + 1) & 3][k])
#define R6(q, k) (((q) >> 1) ? xi[(q) & 1][0][k] : a6[((q) + 2) & 3][k])
#define R7(q, k) (((q) >> 1) ? xi[(q) & 1][1][k] : a7[((q) + 2) & 3][k])
#define R8(q, k) (((q) >> 1) ? xi[(q) & 1][2][k] : a8[((q) + 2) & 3][k])
  // one column solve; Q1 = where the k=1 horizontal diagonals are read (the snapshot for red-black, p itself for four-colour)
#define COLUMN(q)                                                                                                          \
  {                                                                                                                         \
    const int i = 2 * bi + 1 + ((q) >> 1), j = 2 * bj + 1 + ((q) & 1);                                                       \
    /* the cell and its neighbours, clamped: at a side the neighbour is the image of the column itself */                    \
    const int oc = (i - 1) * W + (j - 1);                                                                                   \
    const int sjm = j > 1 ? -1 : 0, sjp = j < ny ? 1 : 0, sim = i > 1 ? -W : 0, sip = i < nx ? W : 0;                       \
    double x[NZ];                                                                                                           \
    double d1 = 0, d2 = 0, d3 = 0, d4 = 0;                                                                                  \
    if (REAL) { d1 = Q1[oc + sim + sjp]; d2 = Q1[oc + sip + sjm]; d3 = Q1[oc + sim + sjm]; d4 = Q1[oc + sip + sjp]; }        \
    double pjm[NZ], pjp[NZ], pim[NZ], pip[NZ];                                                                              \
    _Pragma("unroll") for (int k = 0; k < NZ; k++) {                                                                        \
      const int o = k * PL + oc;                                                                                            \
      pjm[k] = P[o + sjm]; pjp[k] = P[o + sjp]; pim[k] = P[o + sim]; pip[k] = P[o + sip];                                   \
    }                                                                                                                       \
    double xv = 0.0;                                                                                                        \
    _Pragma("unroll") for (int k = 0; k < NZ; k++) {                                                                        \
      double rhs;                                                                                                           \
      if (k == 0) {                                                                                                         \
        rhs = ob[q][k] - a3[q][k] * pjm[k + 1] - a4[q][k] * pjm[k] - R4(q, k) * pjp[k] - R5(q, k + 1) * pjp[k + 1]          \
                       - a6[q][k] * pim[k + 1] - a7[q][k] * pim[k] - R7(q, k) * pip[k] - R8(q, k + 1) * pip[k + 1];          \
        if (REAL) rhs = rhs - a5[q][0] * d1 - e2[q] * d2 - a8[q][0] * d3 - e4[q] * d4;                                      \
        xv = rhs * bet[q][k];                                                                                               \
      } else if (k < NZ - 1) {                                                                                              \
        rhs = ob[q][k] - a3[q][k] * pjm[k + 1] - R3(q, k - 1) * pjp[k - 1] - a4[q][k] * pjm[k] - R4(q, k) * pjp[k]          \
                       - a5[q][k] * pjm[k - 1] - R5(q, k + 1) * pjp[k + 1]                                                  \
                       - a6[q][k] * pim[k + 1] - R6(q, k - 1) * pip[k - 1] - a7[q][k] * pim[k] - R7(q, k) * pip[k]          \
                       - a8[q][k] * pim[k - 1] - R8(q, k + 1) * pip[k + 1];                                                 \
        xv = (rhs - a2[q][k] * xv) * bet[q][k];                                                                             \
      } else {                                                                                                              \
        rhs = ob[q][k] - R3(q, k - 1) * pjp[k - 1] - a4[q][k] * pjm[k] - R4(q, k) * pjp[k] - a5[q][k] * pjm[k - 1]          \
                       - R6(q, k - 1) * pip[k - 1] - a7[q][k] * pim[k] - R7(q, k) * pip[k] - a8[q][k] * pim[k - 1];          \
        xv = (rhs - a2[q][k] * xv) * bet[q][k];                                                                             \
      }                                                                                                                     \
      x[k] = xv;                                                                                                            \
    }                                                                                                                       \
    /* gam(k) = dd(k-1)*bet(k-1) (mg_relax.f90:325): the same product as the stored pivot table, recomputed */              \
    _Pragma("unroll") for (int k = NZ - 2; k >= 0; k--) x[k] = x[k] - (a2[q][k + 1] * bet[q][k]) * x[k + 1];                \
    _Pragma("unroll") for (int k = 0; k < NZ; k++) P[k * PL + oc] = x[k];                                                   \
    if (walk_on) { const double d0_ = x[0] - P1[oc];   /* d0 of the walk */                                                 \
      if (rw) DS[((j - 1) >> 1) * RWP + (i - 1)] = d0_; else Q4[4 * oc] = d0_; }                                            \
  }
  for (int it = 0; it < nsweeps; it++) {
    if (method == 2) {  // four colours (mg_relax.f90:212-230): (i odd,j odd), (i odd,j even), (i even,j odd), (i even,j even)
      const double *__restrict__ Q1 = P;
      if (mine) COLUMN(0)
      __syncthreads();
      if (mine) COLUMN(1)
      __syncthreads();
      if (mine) COLUMN(2)
      __syncthreads();
      if (mine) COLUMN(3)
      __syncthreads();
    } else {  // red-black (mg_relax.f90:170-186), parallel semantics: same-colour k=1 diagonals from the snapshot taken before the pass
      const double *__restrict__ Q1 = REAL ? P1 : P;
      // seq: the reference's plane-after-plane order on top of the parallel pass (header of mgx_rbseq.hip): the walk over the planes finds
      // u = new - old bottom value of every column of the colour (from plane i-1's u and the two k=1 couplings cA(5), cA(8) of the column;
      // zero beyond the sides: a halo cell is not refreshed during a colour), then p += g s.  Column q of the lane lies in plane i.
      // The walk runs on the lanes of wave 0, one lane per column of the colour's half-row, plane after plane with everything in
      // registers but the operands, which are asked for four planes ahead (one 32-byte LDS read per plane: at most 15 LDS requests
      // are counted per wave): s = -cA5 u(j+1,i-1) - cA8 u(j-1,i-1), u = d0 + g1 s; the neighbour j-1 / j+1 of the previous plane is the
      // lane itself and the next lane (a DPP wave shift).  s replaces d0 in the quad.  Lanes past the half-row read the zero quad: u = 0.
      // Measured (16x16x2, 40 sweeps = 80 colours, box of the pool): parallel passes 55 us per call, + d0 / correction / barriers 86, + the
      // walk 169; i.e. ~155 cycles per plane step, with the operands one plane ahead (four 8-byte reads, a conditional store) as with four
      // planes ahead and no branch: ONE wave gets a fraction of the LDS request rate (three LDS operations per step), not its latency.
      // quad index of the lane's column in plane i; planes past nx and lanes past the half-row: the zero quad (no branch anywhere in the walk)
#define WALK_OC(i, JODD) ((wact && (i) <= nx) ? ((i) - 1) * W + ((JODD) ? jA : jA + 1) : PL)
#define WALK_LOAD4(i0_, RBV, R)                                                                                              \
      { _Pragma("unroll") for (int d_ = 0; d_ < 4; d_++) __builtin_memcpy(&R[d_], Q4 + 4 * WALK_OC((i0_) + d_, ((d_ & 1) == 0) == ((RBV) == 1)), 32); }
#define WALK_STEP4(i0_, RBV, R)                                                                                              \
      { _Pragma("unroll") for (int d_ = 0; d_ < 4; d_++) {                                                                    \
          const bool jo_ = ((d_ & 1) == 0) == ((RBV) == 1);   /* the plane i0_ + d_ holds the colour's odd j (i0_ is odd) */    \
          const double sh_ = jo_ ? wave_shr1(up) : wave_shl1(up);                                                             \
          const double ua_ = jo_ ? up : sh_, ub_ = jo_ ? sh_ : up;                                                            \
          const double s_ = __builtin_fma(-R[d_].v[2], ub_, -(R[d_].v[1] * ua_));                                             \
          up = __builtin_fma(R[d_].v[3], s_, R[d_].v[0]);                                                                     \
          Q4[4 * WALK_OC((i0_) + d_, jo_)] = s_;                                                                              \
        } }
#define WALK_APPLY(q)                                                                                                        \
      {                                                                                                                       \
        const int oc_ = (2 * bi + ((q) >> 1)) * W + 2 * bj + ((q) & 1);                                                       \
        const double s_ = rw ? DS[(bj) * RWP + 2 * bi + ((q) >> 1)] : Q4[4 * oc_];   /* column (i,j) = (2bi+1+(q>>1), 2bj+1+(q&1)): jh = bj */ \
        _Pragma("unroll") for (int k = 0; k < NZ; k++) P[k * PL + oc_] = P[k * PL + oc_] + GK[k * PL + oc_] * s_;              \
        P1[oc_] = P[oc_];   /* the snapshot stays current: no copy of the level in front of the next pass */                  \
      }
      // RBV = 1: odd planes hold the colour's odd j (QA = 0), even planes its even j (QB = 3); RBV = 2: the other way round (QA = 1, QB = 2)
#define SEQ_WALK(QA, QB, RBV)                                                                                                \
      if (REAL && seq) {                                                                                                      \
        if (rwc && rw) {                                                                                                      \
          double dv[RWN], up = 0.0;                                                                                           \
          _Pragma("unroll") for (int d_ = 0; d_ < RWN; d_ += 2) __builtin_memcpy(&dv[d_], DS + lane * RWP + d_, 16);           \
          _Pragma("unroll") for (int d_ = 0; d_ < RWN; d_++) {                                                                \
            const bool jo_ = ((d_ & 1) == 0) == ((RBV) == 1);                                                                 \
            const double sh_ = jo_ ? wave_shr1(up) : wave_shl1(up);                                                           \
            const double ua_ = jo_ ? up : sh_, ub_ = jo_ ? sh_ : up;                                                          \
            const double s_ = __builtin_fma(-k8[RBV - 1][d_], ub_, -(k5[RBV - 1][d_] * ua_));                                 \
            up = __builtin_fma(kg[RBV - 1][d_], s_, (d_ < nx && lane < (ny >> 1)) ? dv[d_] : 0.0);                            \
            dv[d_] = s_;                                                                                                      \
          }                                                                                                                   \
          _Pragma("unroll") for (int d_ = 0; d_ < RWN; d_ += 2) __builtin_memcpy(DS + lane * RWP + d_, &dv[d_], 16);           \
        } else if (threadIdx.x < 64) {                                                                                        \
          const bool wact = lane < (ny >> 1);                                                                                 \
          const int jA = 2 * lane;                                                                                            \
          double up = 0.0;                                                                                                    \
          VecD4 ra[4], rb[4];                                                                                                 \
          WALK_LOAD4(1, RBV, ra)                                                                                              \
          for (int i0 = 1; i0 <= nx; i0 += 8) {                                                                               \
            WALK_LOAD4(i0 + 4, RBV, rb)                                                                                       \
            WALK_STEP4(i0, RBV, ra)                                                                                           \
            WALK_LOAD4(i0 + 8, RBV, ra)                                                                                       \
            WALK_STEP4(i0 + 4, RBV, rb)                                                                                       \
          }                                                                                                                   \
        }                                                                                                                     \
        __syncthreads();                                                                                                      \
        if (mine) { WALK_APPLY(QA) WALK_APPLY(QB) }                                                                           \
        __syncthreads();                                                                                                      \
      }
      // (walk_on: the correction keeps the snapshot current -- a pass reads and changes entries of its own colour only -- one copy per call)
      if (REAL && !(walk_on && it > 0)) { for (int t = lane; t < PL; t += NT) P1[t] = P[t]; __syncthreads(); }
      if (mine) { COLUMN(0) COLUMN(3) }   // rb = 1: j = 1+mod(i+1,2): (i odd, j odd) and (i even, j even)
      __syncthreads();
      SEQ_WALK(0, 3, 1)
      if (REAL && !walk_on) { for (int t = lane; t < PL; t += NT) P1[t] = P[t]; __syncthreads(); }
      if (mine) { COLUMN(1) COLUMN(2) }   // rb = 2
      __syncthreads();
      SEQ_WALK(1, 2, 2)
#undef SEQ_WALK
#undef WALK_APPLY
#undef WALK_STEP4
#undef WALK_LOAD4
#undef WALK_OC
    }
  }
  if (FZ && (flags & 2) && mine) {
    // compute_residual of the lane's four columns (stored slots, mg_relax.f90:464-509), then fine2coarse_3D: (k,jA,iA) + (k,jA,iB) + (k,jB,iA)
    // + (k,jB,iB), then the same of k+1 (mg_intergrids.f90:149-160) -- columns q = 0, 2, 1, 3 of the block
    double rr[4][NZ];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int i = 2 * bi + 1 + (q >> 1), j = 2 * bj + 1 + (q & 1);
      const int oc = (i - 1) * W + (j - 1);
      const int sjm = j > 1 ? -1 : 0, sjp = j < ny ? 1 : 0, sim = i > 1 ? -W : 0, sip = i < nx ? W : 0;
      const int c0 = GI(0, j, i);
      double pc[NZ], pjm[NZ], pjp[NZ], pim[NZ], pip[NZ], a1[NZ];
#pragma unroll
      for (int k = 0; k < NZ; k++) {
        const int o = k * PL + oc;
        pc[k] = P[o]; pjm[k] = P[o + sjm]; pjp[k] = P[o + sjp]; pim[k] = P[o + sim]; pip[k] = P[o + sip];
        a1[k] = LD32(G.cA[0], c0 + k * G.RS);
      }
#pragma unroll
      for (int k = 0; k < NZ; k++) {
        double r;
        if (k == 0) {
          r = ob[q][k] - a1[k] * pc[k] - a2[q][k + 1] * pc[k + 1] - a3[q][k] * pjm[k + 1] - a4[q][k] * pjm[k] - R4(q, k) * pjp[k] - R5(q, k + 1) * pjp[k + 1]
                       - a6[q][k] * pim[k + 1] - a7[q][k] * pim[k] - R7(q, k) * pip[k] - R8(q, k + 1) * pip[k + 1];
          if (REAL) r = r - a5[q][0] * P[oc + sim + sjp] - e2[q] * P[oc + sip + sjm] - a8[q][0] * P[oc + sim + sjm] - e4[q] * P[oc + sip + sjp];
        } else if (k < NZ - 1) {
          r = ob[q][k] - a1[k] * pc[k] - a2[q][k] * pc[k - 1] - a2[q][k + 1] * pc[k + 1] - a3[q][k] * pjm[k + 1] - R3(q, k - 1) * pjp[k - 1]
                       - a4[q][k] * pjm[k] - R4(q, k) * pjp[k] - a5[q][k] * pjm[k - 1] - R5(q, k + 1) * pjp[k + 1]
                       - a6[q][k] * pim[k + 1] - R6(q, k - 1) * pip[k - 1] - a7[q][k] * pim[k] - R7(q, k) * pip[k]
                       - a8[q][k] * pim[k - 1] - R8(q, k + 1) * pip[k + 1];
        } else {
          r = ob[q][k] - a1[k] * pc[k] - a2[q][k] * pc[k - 1] - R3(q, k - 1) * pjp[k - 1] - a4[q][k] * pjm[k] - R4(q, k) * pjp[k]
                       - a5[q][k] * pjm[k - 1] - R6(q, k - 1) * pip[k - 1] - a7[q][k] * pim[k] - R7(q, k) * pip[k] - a8[q][k] * pim[k - 1];
        }
        rr[q][k] = r;
      }
    }
    const int i2 = bi + 1, j2 = bj + 1, cjp = jpos(C, j2);
    const long long oc2 = (long long)i2 * C.plane + cjp;
#pragma unroll
    for (int k2 = 0; k2 < NZ / 2; k2++) {
      const int k = 2 * k2;
      const double z = rr[0][k] + rr[2][k] + rr[1][k] + rr[3][k] + rr[0][k + 1] + rr[2][k + 1] + rr[1][k + 1] + rr[3][k + 1];
      const long long rc = (long long)k2 * C.RS;
      C.b[oc2 + rc] = z;
      mirror_store(C, C.b, rc, j2, i2, cjp, z, ph);
      C.p[oc2 + rc] = 0.0;
      mirror_store(C, C.p, rc, j2, i2, cjp, 0.0, ph);
    }
  }
#undef COLUMN
#undef R3
#undef R4
#undef R5
#undef R6
#undef R7
#undef R8
  // write back, halo included: every halo cell is the image of the interior cell its indices clamp to.  (ii, jj) of a lane's cells
  // once (no division inside the row loop), 32-bit offsets as above
  const int WH = ny + 2, PLH = (nx + 2) * WH;
  for (int r = lane; r < PLH; r += NT) {
    const int ii = r / WH, jj = r - ii * WH;
    const int ci = ii < 1 ? 1 : (ii > nx ? nx : ii), cj = jj < 1 ? 1 : (jj > ny ? ny : jj);
    const int g0 = GI(0, jj, ii), l0 = (ci - 1) * W + (cj - 1);
#pragma unroll
    for (int k0 = 0; k0 < NZ; k0++) *(double *)((char *)G.p + (unsigned)((g0 + k0 * G.RS) << 3)) = P[k0 * PL + l0];
  }
#undef LD32
#undef GI
}

extern "C" {

// returns 1 when launched: a closed level of <= 256 columns with nz = 2 (the coarsest grid of every BASELINE configuration) as one
// wave, or of <= 1024 columns with nz = 2 / 4 (the 32x32x4 level above it) as four waves, or -- nz = 2 only -- of <= 2048 columns as
// eight waves: the 64x32x2 coarsest grid that eight GPUs gather (4x2 ranks of 512x512x64: its 40 sweeps took 436 us in k_relax_small,
// which re-reads every operand through L2, a fifth of a V-cycle on every rank)
// Cv, flags: see k_relax_wave (0 / nullptr = the plain relax call); mgxk_relax_wave_fused is the entry the cycles use
// mode (red-black with cmatrix='real' only): 0 = parallel colour passes, 1 = the bit-exact plane loop (not here: k_relax_reg), 2 = the
// reference's sequential order by the walk (k_relax_wave's seq)
static int relax_wave_launch(hipStream_t st, const LevView *L, int nsweeps, int method, int real, Sides ph, int mode, const LevView *Cv, int flags, int nbatch = 1, long long bstride = 0) {
  mgx_before_launch();
  static const bool off = getenv("MGX_NO_WAVE") != nullptr, off4 = getenv("MGX_NO_WAVE4") != nullptr, off8 = getenv("MGX_NO_WAVE8") != nullptr;
  if (off || (L->nz != 2 && L->nz != 4) || method == 0 || (mode == 1 && method == 1 && real)) return 0;
  const int seq = (mode == 2 && method == 1 && real) ? 1 : 0;
  if (seq && L->ny / 2 > WAVE) return 0;  // the walk keeps a half-row on the lanes of one wave; wider levels: k_relax_reg's plane loop
  const int nblk = (L->nx / 2) * (L->ny / 2);
  if (!(ph.S && ph.E && ph.N && ph.W) || (L->nx & 1) || (L->ny & 1) || nblk > (L->nz == 2 && !off8 ? 8 : 4) * WAVE) return 0;
  if ((nblk > WAVE || L->nz == 4) && off4) return 0;
  size_t bytes = ((size_t)L->nz + 1) * (L->nx + 2) * (L->ny + 2) * sizeof(double);
  if (seq) {  // p, the snapshot and u, interiors only
    bytes = ((2 * (size_t)L->nz + 5) * L->nx * L->ny + 4 + 64 * 18) * sizeof(double);  // p, the snapshot, the operand quads (+ a zero quad), g, the register walk's d0 / s rows
    if (bytes > 160 * 1024) return 0;
  }
  const LevView Cc = Cv ? *Cv : *L;
#define WAVE_LDS(KERNEL)                                                                                                      \
  { static size_t granted = 65536;                                                                                               \
    if (bytes > granted) { if (hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { (void)hipGetLastError(); return 0; } granted = 160 * 1024; } }
#define WAVE_CASE(NZV, NTV)                                                                                                   \
  { if (seq) { WAVE_LDS((k_relax_wave<NZV, true, NTV, false, true>))                                                           \
      hipLaunchKernelGGL((k_relax_wave<NZV, true, NTV, false, true>), dim3(nbatch), dim3(NTV), bytes, st, *L, nsweeps, method, ph, Cc, flags, bstride); } \
    else if (real) hipLaunchKernelGGL((k_relax_wave<NZV, true, NTV>), dim3(nbatch), dim3(NTV), bytes, st, *L, nsweeps, method, ph, Cc, flags, bstride); \
    else hipLaunchKernelGGL((k_relax_wave<NZV, false, NTV>), dim3(nbatch), dim3(NTV), bytes, st, *L, nsweeps, method, ph, Cc, flags, bstride);          \
    return mgx_launched(); }
#define WAVE_CASE_FZ(NTV)                                                                                                     \
  { if (seq) { WAVE_LDS((k_relax_wave<4, true, NTV, true, true>))                                                              \
      hipLaunchKernelGGL((k_relax_wave<4, true, NTV, true, true>), dim3(nbatch), dim3(NTV), bytes, st, *L, nsweeps, method, ph, Cc, flags, bstride); } \
    else if (real) hipLaunchKernelGGL((k_relax_wave<4, true, NTV, true>), dim3(nbatch), dim3(NTV), bytes, st, *L, nsweeps, method, ph, Cc, flags, bstride); \
    else hipLaunchKernelGGL((k_relax_wave<4, false, NTV, true>), dim3(nbatch), dim3(NTV), bytes, st, *L, nsweeps, method, ph, Cc, flags, bstride);      \
    return mgx_launched(); }
  if (flags & 3) {
    if (L->nz != 4) return 0;
    if (nblk <= WAVE) WAVE_CASE_FZ(64) else WAVE_CASE_FZ(256)
  }
  if (L->nz == 2) { if (nblk <= WAVE) WAVE_CASE(2, 64) else if (nblk <= 4 * WAVE) WAVE_CASE(2, 256) else WAVE_CASE(2, 512) }
  if (nblk <= WAVE) WAVE_CASE(4, 64) else WAVE_CASE(4, 256)
#undef WAVE_CASE_FZ
#undef WAVE_LDS
#undef WAVE_CASE
}

int mgxk_relax_wave(hipStream_t st, const LevView *L, int nsweeps, int method, int real, Sides ph, int mode) {
  return relax_wave_launch(st, L, nsweeps, method, real, ph, mode, nullptr, 0);
}
// ---- the coarsest-level solve of a cycle as ONE matrix-vector product (option "coarsest_direct") -----------------------------------
// Inside a cycle the coarsest level is entered with p = 0 (fine2coarse, mg_intergrids.f90:70) and left after relax(nlevs, ns_coarsest)
// (mg_solvers.f90:117,144): ns_coarsest sweeps of a FIXED linear iteration from zero, i.e. p = B b with B = (sum_k R^k) N a property of
// the level's matrix alone.  B is built when the coefficients are (mgxk_coarse_direct_build): the n = nx ny nz unit vectors are relaxed as a
// batch by the very kernel that serves the level (k_relax_wave, one workgroup per unit vector: two rounds of workgroups on the 256 CUs for the
// 512 cells of 16x16x2), so that every column of B holds that kernel's own bits; k_coarse_direct then forms p = B b (n^2 multiply-adds spread
// over n / 64 x n / 64 one-wave workgroups; partial sums combined by the last workgroup of each row tile in a fixed order: deterministic) and stores p
// with its physical mirrors.  ~7 us instead of 121 (red-black in the sequential order) / 34 (four colours) for the 160 / 80 dependent colour
// passes of the 16x16x2 level.  The same linear map in another association: NOT the same bits as the sweeps (1e-15 of max|p|; tests: 1e-12),
// which is why the default uses it only where the iteration is tolerance-based anyway (red-black in the sequential order at speed).
constexpr int CDB = 64, CDT = 64;   // columns and rows of B per workgroup (one wave)
__device__ __forceinline__ void cd_cell(const LevView &L, int r, int *k0, int *j, int *i) { *k0 = r % L.nz; const int q = r / L.nz; *j = 1 + q % L.ny; *i = 1 + q / L.ny; }
__device__ __forceinline__ long long cd_js(const LevView &L, int r) { int k0, j, i; cd_cell(L, r, &k0, &j, &i); return (long long)i * L.plane + (long long)k0 * L.RS + jpos(L, j); }
// batch set-up: workgroup c zeroes its p and b and sets b = e_c
__global__ void k_cd_init(LevView L, double *pb, long long stride, int n) {
  const int c = blockIdx.x;
  double *pp = pb + (long long)c * stride, *bb = pb + (long long)(n + c) * stride;
  for (long long t = threadIdx.x; t < stride; t += blockDim.x) { pp[t] = 0.0; bb[t] = 0.0; }
  __syncthreads();
  if (threadIdx.x == 0) bb[cd_js(L, c)] = 1.0;
}
// M[c * n + r] = column c of B at cell r
__global__ void k_cd_compact(LevView L, const double *pb, long long stride, double *M, int n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (r < n) M[(long long)c * n + r] = pb[(long long)c * stride + cd_js(L, r)];
}
// One wave = CDT rows x CDB columns of B.  The tickets of a row tile are atomics on ONE word, which the memory side serialises (~0.4 us each with the
// workgroups spread over the eight XCDs: the first version, 64 slabs of 8 columns, spent 26 us there): few, wide slabs, one word per 64-byte line.
__global__ __launch_bounds__(CDT) void k_coarse_direct(LevView L, const double *__restrict__ M, double *part, unsigned int *cnt, Sides ph, int n) {
  __shared__ double bs[CDB];
  __shared__ int last;
  const int r = blockIdx.x * CDT + threadIdx.x, c0 = blockIdx.y * CDB, nslab = gridDim.y;
  const int rc = r < n ? r : n - 1;
  for (int t = threadIdx.x; t < CDB; t += CDT) bs[t] = c0 + t < n ? L.b[cd_js(L, c0 + t)] : 0.0;
  double m[CDB];
#pragma unroll
  for (int t = 0; t < CDB; t++) { const int c = c0 + t < n ? c0 + t : n - 1; m[t] = M[(long long)c * n + rc]; }   // (columns past n meet b = 0)
  __syncthreads();
  double acc = 0.0;
#pragma unroll
  for (int t = 0; t < CDB; t++) acc += m[t] * bs[t];
  if (r < n) part[(long long)blockIdx.y * n + r] = acc;
  __threadfence();   // release: this workgroup's partial sums before its ticket
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(cnt + blockIdx.x * 16, 1u) == (unsigned int)(nslab - 1);
  __syncthreads();
  if (!last) return;
  __threadfence();   // acquire: every slab's partial sums of this row tile
  if (threadIdx.x == 0) cnt[blockIdx.x * 16] = 0;   // for the next call (calls on one level are ordered by the stream)
  if (r >= n) return;
  double v = 0.0;
  for (int sl = 0; sl < nslab; sl++) v += __hip_atomic_load(part + (long long)sl * n + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int k0, j, i;
  cd_cell(L, r, &k0, &j, &i);
  const int c = jpos(L, j);
  const long long ro = (long long)k0 * L.RS;
  L.p[(long long)i * L.plane + ro + c] = v;
  mirror_store(L, L.p, ro, j, i, c, v, ph);
}

// cells of a level the direct solve serves (a closed level of the one-workgroup kernel, at most 2048 cells), 0 = none
int mgxk_coarse_direct_cells(const LevView *L) {
  const long long n = (long long)L->nx * L->ny * L->nz;
  return ((L->nz == 2 || L->nz == 4) && n <= 2048 && !(L->nx & 1) && !(L->ny & 1)) ? (int)n : 0;
}
// build B (M: n * n doubles) with the level's own relax kernel; pb: scratch of 2 * n * stride doubles (stride = the level's array size).
// Returns 1 when built (all of it enqueued on st), 0 = the level has no one-workgroup kernel for this method / mode.
int mgxk_coarse_direct_build(hipStream_t st, const LevView *L, int nsweeps, int method, int real, Sides ph, int mode, double *pb, long long stride, double *M) {
  const int n = mgxk_coarse_direct_cells(L);
  if (!n || nsweeps < 1) return 0;
  hipLaunchKernelGGL(k_cd_init, dim3(n), dim3(256), 0, st, *L, pb, stride, n);
  LevView B = *L;
  B.p = pb; B.b = pb + (long long)n * stride;
  if (!relax_wave_launch(st, &B, nsweeps, method, real, ph, mode, nullptr, 0, n, stride)) return 0;
  hipLaunchKernelGGL(k_cd_compact, dim3((n + 255) / 256, n), dim3(256), 0, st, *L, pb, stride, M, n);
  return mgx_launched();
}
// p = B b with the mirrors; part: (n / CDB rounded up) * n doubles, cnt: one word per 64 rows, 64 bytes apart, zero before the first call
int mgxk_coarse_direct_apply(hipStream_t st, const LevView *L, const double *M, double *part, unsigned int *cnt, Sides ph) {
  const int n = mgxk_coarse_direct_cells(L);
  if (!n) return 0;
  mgx_before_launch();
  hipLaunchKernelGGL(k_coarse_direct, dim3((n + CDT - 1) / CDT, (n + CDB - 1) / CDB), dim3(CDT), 0, st, *L, M, part, cnt, ph, n);
  return mgx_launched();
}
int mgxk_coarse_direct_slabs(int n) { return (n + CDB - 1) / CDB; }

// relax(lev, nsweeps) of a closed level the one-workgroup kernel serves, with coarse2fine(lev) folded in front (flags & 1) and / or
// compute_residual(lev) + fine2coarse(lev) folded behind (flags & 2); C = level lev+1 (closed, exactly half the size, not gathered).
// Returns 1 when launched, 0 = the caller runs the separate operators.
int mgxk_relax_wave_fused(hipStream_t st, const LevView *L, const LevView *C, int nsweeps, int method, int real, Sides ph, int flags, int mode) {
  static const bool off = getenv("MGX_NO_WAVE_FUSE") != nullptr;
  if (off || !C || C->nx * 2 != L->nx || C->ny * 2 != L->ny || C->nz * 2 != L->nz || nsweeps < 0) return 0;
  return relax_wave_launch(st, L, nsweeps, method, real, ph, mode, C, flags);
}

}  // extern "C"
