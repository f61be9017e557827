// HIP kernels of the multigrid cycle for gfx950 (MI355X).  Hand-written, fp64, bandwidth-bound stencils.
//
// All arithmetic keeps the reference's operation order and is compiled with -ffp-contract=off, so a
// colour pass of the four-colour smoother, the residual and the transfers are bit-identical to the
// reference's CPU loops (only reductions differ in summation order).
//
// Thread mapping everywhere: one lane = one (j,i) column, lanes run along the unit-stride half-row of
// the JS layout (mgx_internal.h), so every global access of a wave is one contiguous 512-byte run.
#include <cstdlib>

#include "mgx_device.h"

// ------------------------------------------------------------------------------------------------
// residual r = b - A p on the interior and per-block partial sums of r^2.  mg_relax.f90:421-515.
// gridDim.z = 2: z = 0 handles the odd-j half-rows, z = 1 the even-j ones.
// ------------------------------------------------------------------------------------------------
template <bool REAL>
__global__ __launch_bounds__(256) void k_residual(LevView L, double *__restrict__ partial, int want_norm, int gx, int gy, Sides ph, int stream) {
  // 1-D grid of gx*gy*2 blocks; XCD-aware map (see k_relax_nz): each XCD owns a contiguous range of plane groups,
  // and the two j-parities of a plane group run back to back on the same XCD (they read the same rows)
  int bx, by, bz;
  {
    const int per = gx * 2;
    int grp, local;
    if ((gy & 7) == 0) { const int xcd = blockIdx.x & 7; local = blockIdx.x >> 3; grp = xcd * (gy >> 3) + local / per; local -= (local / per) * per; }
    else { grp = blockIdx.x / per; local = blockIdx.x - grp * per; }
    by = grp; bz = local / gx; bx = local - bz * gx;
  }
  const int jh = bx * WAVE + threadIdx.x;
  const int i = 1 + by * blockDim.y + threadIdx.y;
  const int jodd = bz == 0;
  double acc = 0.0;
  if (jh < (L.ny >> 1) && i <= L.nx) {
    int c, jm, jp;
    if (jodd) { c = L.HO + jh; jm = L.EO + jh; jp = jm + 1; }
    else      { c = L.EO + jh + 1; jm = L.HO + jh; jp = jm + 1; }
    const long long RS = L.RS;
    const int nz = L.nz;
    const double *__restrict__ p = L.p, *__restrict__ b = L.b;
    double *__restrict__ r = L.r;
    const double *__restrict__ a1 = L.cA[0], *__restrict__ a2 = L.cA[1], *__restrict__ a3 = L.cA[2],
                 *__restrict__ a4 = L.cA[3], *__restrict__ a5 = L.cA[4], *__restrict__ a6 = L.cA[5],
                 *__restrict__ a7 = L.cA[6], *__restrict__ a8 = L.cA[7];
    const long long o = (long long)i * L.plane, om = o - L.plane, op = o + L.plane;
    double pjm_m, pjm_0, pjm_p, pim_m, pim_0, pim_p, pc_m, pc_0, pc_p, a2_0, a2_p;
    double m3_m, m3_0, m4_0, m5_p, n6_m, n6_0, n7_0, n8_p, m3_p, m4_p, n6_p, n7_p;
#define LOAD_ROW(q, PJM, PIM, PC, A2, M3, M4, M5, N6, N7, N8)                  \
  {                                                                            \
    const long long ro = (long long)((q)-1) * RS;                              \
    PJM = p[o + ro + jm]; PIM = p[om + ro + c]; PC = p[o + ro + c]; A2 = a2[o + ro + c]; \
    const double pj_ = p[o + ro + jp], pi_ = p[op + ro + c];                   \
    M3 = a3[o + ro + jp] * pj_; M4 = a4[o + ro + jp] * pj_; M5 = a5[o + ro + jp] * pj_; \
    N6 = a6[op + ro + c] * pi_; N7 = a7[op + ro + c] * pi_; N8 = a8[op + ro + c] * pi_; \
  }
    double dum5, dum8;
    LOAD_ROW(1, pjm_0, pim_0, pc_0, a2_0, m3_0, m4_0, dum5, n6_0, n7_0, dum8);
    LOAD_ROW(2, pjm_p, pim_p, pc_p, a2_p, m3_p, m4_p, m5_p, n6_p, n7_p, n8_p);
    (void)dum5; (void)dum8;
    // k = 1 (:464-482)
    double rr = b[o + c] - a1[o + c] * pc_0 - a2_p * pc_p - a3[o + c] * pjm_p - a4[o + c] * pjm_0 - m4_0 - m5_p
                - a6[o + c] * pim_p - a7[o + c] * pim_0 - n7_0 - n8_p;
    if (REAL)
      rr = rr - a5[o + c] * p[om + jp] - a5[op + jm] * p[op + jm] - a8[o + c] * p[om + jm] - a8[op + jp] * p[op + jp];
    const int jcol = jodd ? 2 * jh + 1 : 2 * jh + 2;
    r[o + c] = rr;
    mirror_store(L, r, 0, jcol, i, c, rr, ph);
    acc = acc + rr * rr;
    for (int k = 2; k <= nz - 1; k++) {  // (:484-496)
      pjm_m = pjm_0; pjm_0 = pjm_p; pim_m = pim_0; pim_0 = pim_p; pc_m = pc_0; pc_0 = pc_p; a2_0 = a2_p;
      m3_m = m3_0; m3_0 = m3_p; m4_0 = m4_p; n6_m = n6_0; n6_0 = n6_p; n7_0 = n7_p;
      LOAD_ROW(k + 1, pjm_p, pim_p, pc_p, a2_p, m3_p, m4_p, m5_p, n6_p, n7_p, n8_p);
      const long long ko = o + (long long)(k - 1) * RS + c;
      rr = b[ko] - a1[ko] * pc_0 - a2_0 * pc_m - a2_p * pc_p - a3[ko] * pjm_p - m3_m - a4[ko] * pjm_0 - m4_0
                 - a5[ko] * pjm_m - m5_p - a6[ko] * pim_p - n6_m - a7[ko] * pim_0 - n7_0 - a8[ko] * pim_m - n8_p;
      r[ko] = rr;
      mirror_store(L, r, (long long)(k - 1) * RS, jcol, i, c, rr, ph);
      acc = acc + rr * rr;
    }
    {  // k = nz (:498-509)
      pjm_m = pjm_0; pjm_0 = pjm_p; pim_m = pim_0; pim_0 = pim_p; pc_m = pc_0; pc_0 = pc_p; a2_0 = a2_p;
      m3_m = m3_0; m4_0 = m4_p; n6_m = n6_0; n7_0 = n7_p;
      const long long ko = o + (long long)(nz - 1) * RS + c;
      rr = b[ko] - a1[ko] * pc_0 - a2_0 * pc_m - m3_m - a4[ko] * pjm_0 - m4_0 - a5[ko] * pjm_m - n6_m
                 - a7[ko] * pim_0 - n7_0 - a8[ko] * pim_m;
      r[ko] = rr;
      mirror_store(L, r, (long long)(nz - 1) * RS, jcol, i, c, rr, ph);
      acc = acc + rr * rr;
    }
#undef LOAD_ROW
  }
  if (!want_norm) return;
  // block reduction: wave shuffle, then LDS across the waves of the block (deterministic order)
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double red[16];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, w = tid >> 6;
  if ((tid & 63) == 0) red[w] = acc;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    const int nw = (blockDim.x * blockDim.y + 63) >> 6;
    for (int q = 0; q < nw; q++) s += red[q];
    partial[blockIdx.x] = s;
  }
}

// residual with matrix-free cross terms (see relax_col_mf): 17 streams per cell instead of 22.  The diagonal of the interior rows is
// rebuilt from the fourteen couplings the row holds anyway (mg_define_matrix.f90:632-639, summed in the reference's order: the same bits
// as the stored slot 1 -- as the smoother and the fused residual+restriction do); rows 1 and nz, whose formula differs, read the stored one.
// The j-1 / j+1 neighbours of p and zy sit side by side in the other half-row: one 16-byte load each.  What only this lane reads (b, a2)
// is streamed past the caches on a level that does not fit them.
template <bool REAL>
__global__ __launch_bounds__(256) void k_residual_mf(LevView L, double *__restrict__ partial, int want_norm, int gx, int gy, Sides ph, int stream) {
  // 1-D grid of gx*gy*2 blocks; XCD-aware map (see k_relax_nz): each XCD owns a contiguous range of plane groups,
  // and the two j-parities of a plane group run back to back on the same XCD (they read the same rows)
  int bx, by, bz;
  {
    const int per = gx * 2;
    int grp, local;
    if ((gy & 7) == 0) { const int xcd = blockIdx.x & 7; local = blockIdx.x >> 3; grp = xcd * (gy >> 3) + local / per; local -= (local / per) * per; }
    else { grp = blockIdx.x / per; local = blockIdx.x - grp * per; }
    by = grp; bz = local / gx; bx = local - bz * gx;
  }
  const int jh = bx * WAVE + threadIdx.x;
  const int i = 1 + by * blockDim.y + threadIdx.y;
  const int jodd = bz == 0;
  double acc = 0.0;
  if (jh < (L.ny >> 1) && i <= L.nx) {
    int c, jm, jp;
    if (jodd) { c = L.HO + jh; jm = L.EO + jh; jp = jm + 1; }
    else      { c = L.EO + jh + 1; jm = L.HO + jh; jp = jm + 1; }
    const long long RS = L.RS;
    const int nz = L.nz;
    const double *__restrict__ p = L.p, *__restrict__ b = L.b;
    double *__restrict__ r = L.r;
    const double *__restrict__ a1 = L.cA[0], *__restrict__ a2 = L.cA[1], *__restrict__ a4 = L.cA[3], *__restrict__ a5 = L.cA[4],
                 *__restrict__ a7 = L.cA[6], *__restrict__ a8 = L.cA[7], *__restrict__ zy = L.zy, *__restrict__ zx = L.zx;
    const long long o = (long long)i * L.plane, om = o - L.plane, op = o + L.plane;
    const double qrt = 0.25;
    // Every request is unconditional (rows past the top clamped to nz, never used) and issued ONE STEP before its first use: the window row
    // k+2 and the own-row values of k+1 are in flight while row k is computed.  (A request inside `if (k + 2 <= nz)` made the number of
    // outstanding loads path-dependent: the compiler then waits for vmcnt(0) at every step and the look-ahead is void.)
    double pc_m = 0, pc_0, pc_p, pc_n, pjm_m = 0, pjm_0, pjm_p, pjm_n, pim_m = 0, pim_0, pim_p, pim_n, pjp_m = 0, pjp_0, pjp_p, pjp_n, pip_m = 0, pip_0, pip_p, pip_n;
    double zy_m = 0, zy_0, zy_p, zy_n, zx_m = 0, zx_0, zx_p, zx_n, a2_0, a2_p, a2_n;
    double zyjm, zyjp, zxim, zxip, a4o, a4jp, a7o, a7ip, bk, zyjm_n, zyjp_n, zxim_n, zxip_n, a4o_n, a4jp_n, a7o_n, a7ip_n, bk_n;
#define LOAD_WIN(q, PC, PJM, PIM, PJP, PIP, ZY, ZX, A2)                        \
  { const long long ro = (long long)(((q) <= nz ? (q) : nz) - 1) * RS;         \
    PC = p[o + ro + c]; LD_PAIR(p + o + ro + jm, PJM, PJP) PIM = p[om + ro + c]; PIP = p[op + ro + c]; \
    ZY = *(zy + o + ro + c); ZX = *(zx + o + ro + c); A2 = ld_rt(a2 + o + ro + c, stream); }
#define LOAD_ROWV(q, ZYJM, ZYJP, ZXIM, ZXIP, A4O, A4JP, A7O, A7IP, BK)         \
  { const long long ro = (long long)(((q) <= nz ? (q) : nz) - 1) * RS, ko = o + ro + c; \
    LD_PAIR(zy + o + ro + jm, ZYJM, ZYJP) ZXIM = zx[om + ro + c]; ZXIP = zx[op + ro + c]; \
    A4O = *(a4 + ko); A4JP = a4[o + ro + jp]; A7O = *(a7 + ko); A7IP = a7[op + ro + c]; BK = ld_rt(b + ko, stream); }
    // rows 1 and nz: stored diagonal; row 1: the k = 1 diagonal slots and the four corner values of p (cmatrix = 'real', mg_relax.f90:475-479)
    const double d_first = a1[o + c], d_last = a1[o + (long long)(nz - 1) * RS + c];
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0, e7 = 0;
    if (REAL) { e0 = a5[o + c]; e1 = p[om + jp]; e2 = a5[op + jm]; e3 = p[op + jm]; e4 = a8[o + c]; e5 = p[om + jm]; e6 = a8[op + jp]; e7 = p[op + jp]; }
    LOAD_WIN(1, pc_0, pjm_0, pim_0, pjp_0, pip_0, zy_0, zx_0, a2_0)
    LOAD_ROWV(1, zyjm, zyjp, zxim, zxip, a4o, a4jp, a7o, a7ip, bk)
    LOAD_WIN(2, pc_p, pjm_p, pim_p, pjp_p, pip_p, zy_p, zx_p, a2_p)
    for (int k = 1; k <= nz; k++) {
      const long long ro = (long long)(k - 1) * RS, ko = o + ro + c;
      LOAD_WIN(k + 2, pc_n, pjm_n, pim_n, pjp_n, pip_n, zy_n, zx_n, a2_n)
      LOAD_ROWV(k + 1, zyjm_n, zyjp_n, zxim_n, zxip_n, a4o_n, a4jp_n, a7o_n, a7ip_n, bk_n)
      double rr;
      if (k == 1) {
        rr = bk - d_first * pc_0 - a2_p * pc_p - (qrt * (zy_p + zyjm)) * pjm_p - a4o * pjm_0 - a4jp * pjp_0
                   - (-qrt * (zyjp + zy_p)) * pjp_p - (qrt * (zx_p + zxim)) * pim_p - a7o * pim_0 - a7ip * pip_0
                   - (-qrt * (zxip + zx_p)) * pip_p;
        if (REAL) rr = rr - e0 * e1 - e2 * e3 - e4 * e5 - e6 * e7;
      } else if (k < nz) {
        const double c3 = qrt * (zy_p + zyjm), c3m = qrt * (zyjp + zy_m), c5 = -qrt * (zy_m + zyjm), c5m = -qrt * (zyjp + zy_p);
        const double c6 = qrt * (zx_p + zxim), c6m = qrt * (zxip + zx_m), c8 = -qrt * (zx_m + zxim), c8m = -qrt * (zxip + zx_p);
        const double dk = -a2_0 - a2_p - a4o - a4jp - a7o - a7ip - c6 - c6m - c8 - c8m - c3 - c3m - c5 - c5m;  // = cA(1,k,j,i), mg_define_matrix.f90:632-639
        rr = bk - dk * pc_0 - a2_0 * pc_m - a2_p * pc_p - c3 * pjm_p - c3m * pjp_m
                   - a4o * pjm_0 - a4jp * pjp_0 - c5 * pjm_m - c5m * pjp_p
                   - c6 * pim_p - c6m * pip_m - a7o * pim_0 - a7ip * pip_0
                   - c8 * pim_m - c8m * pip_p;
      } else {
        rr = bk - d_last * pc_0 - a2_0 * pc_m - (qrt * (zyjp + zy_m)) * pjp_m - a4o * pjm_0 - a4jp * pjp_0
                   - (-qrt * (zy_m + zyjm)) * pjm_m - (qrt * (zxip + zx_m)) * pip_m - a7o * pim_0 - a7ip * pip_0
                   - (-qrt * (zx_m + zxim)) * pim_m;
      }
      st_rt(r + ko, rr, stream);
      mirror_store(L, r, ro, jodd ? 2 * jh + 1 : 2 * jh + 2, i, c, rr, ph);
      acc = acc + rr * rr;
      pc_m = pc_0; pc_0 = pc_p; pc_p = pc_n; pjm_m = pjm_0; pjm_0 = pjm_p; pjm_p = pjm_n; pim_m = pim_0; pim_0 = pim_p; pim_p = pim_n;
      pjp_m = pjp_0; pjp_0 = pjp_p; pjp_p = pjp_n; pip_m = pip_0; pip_0 = pip_p; pip_p = pip_n;
      zy_m = zy_0; zy_0 = zy_p; zy_p = zy_n; zx_m = zx_0; zx_0 = zx_p; zx_p = zx_n; a2_0 = a2_p; a2_p = a2_n;
      zyjm = zyjm_n; zyjp = zyjp_n; zxim = zxim_n; zxip = zxip_n; a4o = a4o_n; a4jp = a4jp_n; a7o = a7o_n; a7ip = a7ip_n; bk = bk_n;
    }
#undef LOAD_ROWV
#undef LOAD_WIN
  }
  if (!want_norm) return;
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double red[16];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, w = tid >> 6;
  if ((tid & 63) == 0) red[w] = acc;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    const int nw = (blockDim.x * blockDim.y + 63) >> 6;
    for (int q = 0; q < nw; q++) s += red[q];
    partial[blockIdx.x] = s;
  }
}

// sum of squares of the interior of a JS field (bnorm of solve_p, mg_solvers.f90:50)
__global__ __launch_bounds__(256) void k_sumsq(LevView L, const double *__restrict__ a, double *__restrict__ partial) {
  const int jh = blockIdx.x * WAVE + threadIdx.x;
  const int i = 1 + blockIdx.y * blockDim.y + threadIdx.y;
  double acc = 0.0;
  if (jh < (L.ny >> 1) && i <= L.nx) {
    const int c = blockIdx.z == 0 ? L.HO + jh : L.EO + jh + 1;
    const long long o = (long long)i * L.plane + c;
    for (int k = 0; k < L.nz; k++) { const double v = a[o + (long long)k * L.RS]; acc += v * v; }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double red[16];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, w = tid >> 6;
  if ((tid & 63) == 0) red[w] = acc;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    const int nw = (blockDim.x * blockDim.y + 63) >> 6;
    for (int q = 0; q < nw; q++) s += red[q];
    partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
  }
}

// inner product of the interiors of two JS fields (norm(lev,x,y), mg_solvers.f90:180-200)
__global__ __launch_bounds__(256) void k_dot(LevView L, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ partial) {
  const int jh = blockIdx.x * WAVE + threadIdx.x;
  const int i = 1 + blockIdx.y * blockDim.y + threadIdx.y;
  double acc = 0.0;
  if (jh < (L.ny >> 1) && i <= L.nx) {
    const int c = blockIdx.z == 0 ? L.HO + jh : L.EO + jh + 1;
    const long long o = (long long)i * L.plane + c;
    for (int k = 0; k < L.nz; k++) acc += a[o + (long long)k * L.RS] * b[o + (long long)k * L.RS];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double red[16];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, w = tid >> 6;
  if ((tid & 63) == 0) red[w] = acc;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    const int nw = (blockDim.x * blockDim.y + 63) >> 6;
    for (int q = 0; q < nw; q++) s += red[q];
    partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
  }
}

// second stage: one block sums the partials in index order -> out[0]
__global__ __launch_bounds__(256) void k_reduce_partials(const double *__restrict__ partial, int n, double *__restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int q = threadIdx.x; q < n; q += 256) s += partial[q];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// ------------------------------------------------------------------------------------------------
// Fcycle's first leg below level 1 (mg_solvers.f90:110-115): for lev = l0 .. l0+DEP-1: grid(lev+1)%b = restriction of grid(lev)%r,
// grid(lev+1)%r = grid(lev+1)%b, grid(lev+1)%p = 0 -- a chain in which every level is the 8-cell sum of the one above and nothing else.
// One workgroup takes a (2^DEP)^3 block of the finest level of the chain into LDS and carries it down DEP levels there (4096, 512, 64,
// 8, 1 cells), storing every level's b, r and zeroed p with their physical images on the way: one launch instead of DEP (the small
// levels are launch-bound: 6.3 + 5.5 + 4.6 + 4.3 us and three kernel boundaries for the 256x256x32 -> 16x16x2 chain of the bench).
// Same 8-term sum in the same order as k_fine2coarse (left to right: (k,jA,iA) + (k,jA,iB) + (k,jB,iA) + (k,jB,iB), then the same of
// k+1): bit-identical.  Closed levels only (the mirrors then reach every halo cell), none of them gathered.
struct LevChain { LevView v[5]; };
template <int DEP>
__global__ __launch_bounds__(256) void k_restrict_chain(LevChain Ch, Sides ph) {
  constexpr int E = 1 << DEP;
  __shared__ double buf[E * E * E + (E / 2) * (E / 2) * (E / 2)];
  double *cur = buf, *nxt = buf + E * E * E;
  const LevView &F = Ch.v[0];
  const int nbj = F.ny / E, nbk = F.nz / E;
  int bb = blockIdx.x;
  const int kb = bb % nbk; bb /= nbk;
  const int jb = bb % nbj, ib = bb / nbj;
  // the fine block, (i, k, j) with j split by parity so that a lane run is a contiguous half-row run: idx = (ii * E + kk) * E + jj
  for (int t = threadIdx.x; t < E * E * E; t += 256) {
    const int hj = t % (E / 2), par = (t / (E / 2)) & 1, kk = (t / E) % E, ii = t / (E * E);
    const int jj = 2 * hj + par;               // 0-based inside the block: even jj = odd j (1-based)
    const int i = ib * E + ii + 1, j = jb * E + jj + 1, k = kb * E + kk + 1;
    cur[(ii * E + kk) * E + jj] = F.r[(long long)i * F.plane + (long long)(k - 1) * F.RS + jpos(F, j)];
  }
  __syncthreads();
#pragma unroll
  for (int d = 1; d <= DEP; d++) {
    const int e = E >> (d - 1), h = e >> 1;    // edge of the level above and of this one
    const LevView &C = Ch.v[d];
    for (int t = threadIdx.x; t < h * h * h; t += 256) {
      const int jc = t % h, kc = (t / h) % h, ic = t / (h * h);
      const int iA = 2 * ic, iB = iA + 1, k0 = 2 * kc, k1 = k0 + 1, jA = 2 * jc, jB = jA + 1;
#define CH(ii_, kk_, jj_) cur[((ii_) * e + (kk_)) * e + (jj_)]
      const double z = CH(iA, k0, jA) + CH(iB, k0, jA) + CH(iA, k0, jB) + CH(iB, k0, jB) + CH(iA, k1, jA) + CH(iB, k1, jA) + CH(iA, k1, jB) + CH(iB, k1, jB);
#undef CH
      nxt[(ic * h + kc) * h + jc] = z;
      const int i2 = ib * h + ic + 1, j2 = jb * h + jc + 1, k2 = kb * h + kc + 1, cj = jpos(C, j2);
      const long long rc = (long long)(k2 - 1) * C.RS, oc = (long long)i2 * C.plane + rc + cj;
      C.b[oc] = z; mirror_store(C, C.b, rc, j2, i2, cj, z, ph);
      C.r[oc] = z; mirror_store(C, C.r, rc, j2, i2, cj, z, ph);
      C.p[oc] = 0.0; mirror_store(C, C.p, rc, j2, i2, cj, 0.0, ph);
    }
    __syncthreads();
    double *sw = cur; cur = nxt; nxt = sw;
  }
}

// ------------------------------------------------------------------------------------------------
// restriction: coarse b = sum of the 8 fine r.  mg_intergrids.f90:139-162.  One lane = one coarse column.
// `dst` is the coarse b, or the pre-gather block (nxc x nyc) when the coarse level is gathered.
// ------------------------------------------------------------------------------------------------
// blockIdx.z: a run of KC coarse levels -- nothing here is sequential in k, and one wave per coarse column-chunk (1024 waves at 512x512x64:
// one per SIMD, a memory round trip per level, 41 us) is a latency chain, not a stream.
__global__ __launch_bounds__(256) void k_fine2coarse(LevView F, LevView C, double *__restrict__ dst, Sides ph, int stream, double *__restrict__ dup, double *__restrict__ zero, int KC) {
  const int j2 = 1 + blockIdx.x * WAVE + threadIdx.x;
  const int i2 = 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (j2 > C.ny || i2 > C.nx) return;
  const int ka = 1 + blockIdx.z * KC, kb = ka + KC - 1 < C.nz ? ka + KC - 1 : C.nz;
  const int i = 2 * i2 - 1;
  const int po = F.HO + (j2 - 1), pe = F.EO + j2;  // fine j = 2*j2-1 (odd) and 2*j2 (even)
  const double *__restrict__ x = F.r;
  const long long o0 = (long long)i * F.plane, o1 = o0 + F.plane;
  const long long oc = (long long)i2 * C.plane + jpos(C, j2);
  for (int k2 = ka; k2 <= kb; k2++) {
    const long long r0 = (long long)(2 * k2 - 2) * F.RS, r1 = r0 + F.RS;
    const double z = ld_rt(x + o0 + r0 + po, stream) + ld_rt(x + o1 + r0 + po, stream) + ld_rt(x + o0 + r0 + pe, stream) + ld_rt(x + o1 + r0 + pe, stream)
                   + ld_rt(x + o0 + r1 + po, stream) + ld_rt(x + o1 + r1 + po, stream) + ld_rt(x + o0 + r1 + pe, stream) + ld_rt(x + o1 + r1 + pe, stream);
    dst[oc + (long long)(k2 - 1) * C.RS] = z;
    mirror_store(C, dst, (long long)(k2 - 1) * C.RS, j2, i2, jpos(C, j2), z, ph);
    // closed levels only (the mirrors then reach every halo cell): r_c = b_c of Fcycle (mg_solvers.f90:113) and p_c = 0
    // (mg_intergrids.f90:70) written here instead of a copy and a memset launch
    if (dup) { dup[oc + (long long)(k2 - 1) * C.RS] = z; mirror_store(C, dup, (long long)(k2 - 1) * C.RS, j2, i2, jpos(C, j2), z, ph); }
    if (zero) { zero[oc + (long long)(k2 - 1) * C.RS] = 0.0; mirror_store(C, zero, (long long)(k2 - 1) * C.RS, j2, i2, jpos(C, j2), 0.0, ph); }
  }
}

// ------------------------------------------------------------------------------------------------
// prolongation + correction: fine r = interp(coarse p); fine p += fine r (interior).
// mg_intergrids.f90:366-450 (tri-linear, top level x 1/2), :336-363 (nearest), :226 (p = p + r).
// One lane = one coarse column = 2x2 fine columns.  `src` is the coarse p (or the split block).
// WR = false: the interpolated correction is added to p without being stored in the fine r.  Inside a cycle nothing reads that r
// before the next compute_residual rewrites it (mg_solvers.f90:129-151), and at level 1 it is a third of this kernel's traffic.
// ------------------------------------------------------------------------------------------------
template <bool LINEAR, bool WR>
__global__ __launch_bounds__(256) void k_coarse2fine(LevView F, LevView C, const double *__restrict__ xc, Sides ph, int stream) {
  const int j2 = 1 + blockIdx.x * WAVE + threadIdx.x;
  const int k2 = 1 + blockIdx.y * blockDim.y + threadIdx.y;
  const int i2 = 1 + blockIdx.z;
  if (j2 > C.ny || k2 > C.nz) return;
  const int i = 2 * i2 - 1;
  const int po = F.HO + (j2 - 1), pe = F.EO + j2;  // fine j (odd) and j+1 (even)
  const int c0 = jpos(C, j2), cm = jpos(C, j2 - 1), cp = jpos(C, j2 + 1);
  const long long q0 = (long long)i2 * C.plane, qm = q0 - C.plane, qp = q0 + C.plane;
  const long long o0 = (long long)i * F.plane, o1 = o0 + F.plane;
  double *__restrict__ rf = F.r;
  double *__restrict__ pf = F.p;
  const int nz = C.nz;
#define XC(kk, JJ, QQ) xc[QQ + (long long)((kk)-1) * C.RS + JJ]
#define PUT(k, OO, PP, val) { const long long ro_ = (long long)((k)-1) * F.RS, t_ = OO + ro_ + PP; const double v_ = (val), w_ = ld_rt(pf + t_, stream) + v_; if (WR) st_rt(rf + t_, v_, stream); st_rt(pf + t_, w_, stream); \
    const int jf_ = (PP == po) ? 2 * j2 - 1 : 2 * j2, if_ = (OO == o0) ? i : i + 1; \
    if (WR) mirror_store(F, rf, ro_, jf_, if_, PP, v_, ph); mirror_store(F, pf, ro_, jf_, if_, PP, w_, ph); }
  if (!LINEAR) {
    const double v = XC(k2, c0, q0);
    const int k = 2 * k2 - 1;
    PUT(k, o0, po, v); PUT(k + 1, o0, po, v); PUT(k, o0, pe, v); PUT(k + 1, o0, pe, v);
    PUT(k, o1, po, v); PUT(k + 1, o1, po, v); PUT(k, o1, pe, v); PUT(k + 1, o1, pe, v);
    return;
  }
  const double a = 9. / 16., b = 3. / 16., c = 1. / 16., d = 27. / 64., e = 9. / 64., f = 3. / 64., g = 1. / 64.;
  // the 9 coarse values of level k2 around (j2,i2)
  const double x00 = XC(k2, c0, q0), xmm = XC(k2, cm, qm), xm0 = XC(k2, cm, q0), x0m = XC(k2, c0, qm),
               xpm = XC(k2, cp, qm), xp0 = XC(k2, cp, q0), xmp = XC(k2, cm, qp), x0p = XC(k2, c0, qp), xpp = XC(k2, cp, qp);
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const int k = 2 * k2 - 1 + half;  // fine level
    if (k == 1) {                      // bottom level: bilinear (mg_intergrids.f90:392-405)
      PUT(1, o0, po, +a * x00 + c * xmm + b * xm0 + b * x0m);
      PUT(1, o0, pe, +a * x00 + c * xpm + b * xp0 + b * x0m);
      PUT(1, o1, po, +a * x00 + c * xmp + b * xm0 + b * x0p);
      PUT(1, o1, pe, +a * x00 + c * xpp + b * xp0 + b * x0p);
    } else if (k == 2 * nz) {          // top level: 1/2 bilinear (:434-446)
      PUT(k, o0, po, 0.5 * (a * x00 + c * xmm + b * xm0 + b * x0m));
      PUT(k, o0, pe, 0.5 * (a * x00 + c * xpm + b * xp0 + b * x0m));
      PUT(k, o1, po, 0.5 * (a * x00 + c * xmp + b * xm0 + b * x0p));
      PUT(k, o1, pe, 0.5 * (a * x00 + c * xpp + b * xp0 + b * x0p));
    } else {                           // interior: tri-linear, kp = k2-1 for odd k, k2+1 for even k (:407-432)
      const int kp = k2 - ((k % 2) * 2 - 1);
      const double y00 = XC(kp, c0, q0), ymm = XC(kp, cm, qm), ym0 = XC(kp, cm, q0), y0m = XC(kp, c0, qm),
                   ypm = XC(kp, cp, qm), yp0 = XC(kp, cp, q0), ymp = XC(kp, cm, qp), y0p = XC(kp, c0, qp), ypp = XC(kp, cp, qp);
      PUT(k, o0, po, +d * x00 + f * xmm + e * xm0 + e * x0m + e * y00 + g * ymm + f * ym0 + f * y0m);
      PUT(k, o0, pe, +d * x00 + f * xpm + e * xp0 + e * x0m + e * y00 + g * ypm + f * yp0 + f * y0m);
      PUT(k, o1, po, +d * x00 + f * xmp + e * xm0 + e * x0p + e * y00 + g * ymp + f * ym0 + f * y0p);
      PUT(k, o1, pe, +d * x00 + f * xpp + e * xp0 + e * x0p + e * y00 + g * ypp + f * yp0 + f * y0p);
    }
  }
#undef XC
#undef PUT
}

// Tri-linear prolongation + correction, one lane = a run of KC coarse levels of one coarse column.
// k_coarse2fine above fetches 18 coarse values per lane for 8 fine cells (the 3x3 neighbourhood of two coarse levels): 2.25
// cache accesses per fine cell next to the one load and one store the cell itself needs -- the texture path was busy 80 % of
// the kernel (rocprofv3 MemUnitStalled) and the level-1 launch ran at 4 TB/s.  Here a lane reads only its OWN column of the three
// coarse planes i2-1, i2, i2+1, takes the j2-1 / j2+1 columns from its neighbour lanes (the first and last lane of the wave
// fetch theirs), and keeps a three-level window while it walks up: 3 / (8 KC) coarse loads per fine cell instead of 2.25; the fine
// p of the next coarse level are requested before the current level is stored (without that look-ahead the run is a chain of
// load -> store round trips and loses: 106 us; with it 71 us against 83 us of k_coarse2fine at 512x512x64, WR = false).
// Same expressions in the same order as above (mg_intergrids.f90:392-446): bit-identical.
// SK: the fine columns (i odd, j odd) are left alone -- the first colour of the four-colour sweep that follows overwrites them without
// reading them (a line solve reads only the other columns; mg_relax.f90:212-230), which is a quarter of this kernel's traffic.  Except
// next to a physical south / west boundary: there the column reads its own old value through the mirrored halo cell (p(0) = p(1)),
// so columns j = 1 and i = 1 are updated (and mirrored) as usual.
template <bool WR, bool SK>
__global__ __launch_bounds__(256) void k_coarse2fine_run(LevView F, LevView C, const double *__restrict__ xc, Sides ph, int stream, int KC) {
  const int lane = threadIdx.x;
  const int j2 = 1 + blockIdx.x * WAVE + lane;
  const int i2 = 1 + blockIdx.z;
  const int nz = C.nz;
  const int ka = 1 + (blockIdx.y * blockDim.y + threadIdx.y) * KC;   // wave-uniform
  if (ka > nz) return;
  const int kb = ka + KC - 1 < nz ? ka + KC - 1 : nz;
  const bool live = j2 <= C.ny;
  const int jl = live ? j2 : C.ny + 1;  // lanes past the row still hand their neighbour a value (the halo column ny+1)
  const int i = 2 * i2 - 1;
  const int po = F.HO + (j2 - 1), pe = F.EO + j2;  // fine j (odd) and j+1 (even)
  const int c0 = jpos(C, jl), cm = jpos(C, jl - 1), cp = jpos(C, jl + 1 <= C.ny + 1 ? jl + 1 : jl);
  const long long q0 = (long long)i2 * C.plane, qm = q0 - C.plane, qp = q0 + C.plane;
  const long long o0 = (long long)i * F.plane, o1 = o0 + F.plane;
  double *__restrict__ rf = F.r;
  double *__restrict__ pf = F.p;
  // one coarse level: v[3*a + b], a = plane (i2-1, i2, i2+1), b = column (j2-1, j2, j2+1)
#define ROW(kk, v)                                                                                                  \
  {                                                                                                                 \
    const long long ro_ = (long long)(((kk) < 1 ? 1 : ((kk) > nz ? nz : (kk))) - 1) * C.RS;                          \
    const double t0_ = xc[qm + ro_ + c0], t1_ = xc[q0 + ro_ + c0], t2_ = xc[qp + ro_ + c0];                         \
    v[1] = t0_; v[4] = t1_; v[7] = t2_;                                                                             \
    v[0] = __shfl_up(t0_, 1, WAVE); v[3] = __shfl_up(t1_, 1, WAVE); v[6] = __shfl_up(t2_, 1, WAVE);                 \
    v[2] = __shfl_down(t0_, 1, WAVE); v[5] = __shfl_down(t1_, 1, WAVE); v[8] = __shfl_down(t2_, 1, WAVE);           \
    /* the wave's two outer columns: ONE request per plane, in which lane 0 asks for its j2-1, the last lane for its j2+1 (the others */ \
    /* repeat their own column) -- selected, not branched to */                                                      \
    const double e0_ = xc[qm + ro_ + ce], e1_ = xc[q0 + ro_ + ce], e2_ = xc[qp + ro_ + ce];                         \
    if (lane == 0) { v[0] = e0_; v[3] = e1_; v[6] = e2_; }                                                          \
    if (lane == WAVE - 1) { v[2] = e0_; v[5] = e1_; v[8] = e2_; }                                                   \
  }
  // the eight fine p of coarse level kk: [4*half + 2*(plane i+1) + (column j+1)], loaded one level ahead of their use
  // (every request of the walk is unconditional -- levels past the run are clamped, lanes past the row shadow the last column: a request
  // inside a branch makes the number of outstanding loads path-dependent and the compiler then waits for vmcnt(0) at every step)
#define LOADP(kk, v)                                                                                                \
  {                                                                                                                 \
    const int kq_ = (kk) <= kb ? (kk) : kb;                                                                         \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; h_++) {                                                              \
      const long long ro_ = (long long)(2 * kq_ - 2 + h_) * F.RS;                                                   \
      if (!SK) v[4 * h_ + 0] = ld_rt(pf + o0 + ro_ + pol, stream);                                                  \
      v[4 * h_ + 1] = ld_rt(pf + o0 + ro_ + pel, stream);                                                           \
      v[4 * h_ + 2] = ld_rt(pf + o1 + ro_ + pol, stream); v[4 * h_ + 3] = ld_rt(pf + o1 + ro_ + pel, stream);       \
    }                                                                                                               \
  }
  // SK: the (i odd, j odd) column is read only next to a physical south / west boundary (a lane- or plane-dependent branch: kept apart)
#define LOADP0(kk, v)                                                                                               \
  if (SK && edge0 && live) {                                                                                        \
    const int kq_ = (kk) <= kb ? (kk) : kb;                                                                         \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; h_++) v[4 * h_ + 0] = ld_rt(pf + o0 + (long long)(2 * kq_ - 2 + h_) * F.RS + po, stream); \
  }
  // fine cell Q of the 2x2 block: bit 1 = plane i+1, bit 0 = column j+1 (Q = 0 is the first colour's column)
#define PUT(k, Q, val) if (!(SK && (Q) == 0) || edge0) { const long long OO_ = ((Q) & 2) ? o1 : o0; const int PP_ = ((Q) & 1) ? pe : po;                      \
    const long long ro_ = (long long)((k)-1) * F.RS, t_ = OO_ + ro_ + PP_; const double v_ = (val), w_ = pc[4 * half + (Q)] + v_;                     \
    if (WR) st_rt(rf + t_, v_, stream); st_rt(pf + t_, w_, stream);                                                                                    \
    const int jf_ = ((Q) & 1) ? 2 * j2 : 2 * j2 - 1, if_ = ((Q) & 2) ? i + 1 : i;                                                                      \
    if (WR) mirror_store(F, rf, ro_, jf_, if_, PP_, v_, ph); mirror_store(F, pf, ro_, jf_, if_, PP_, w_, ph); }
  const bool edge0 = (ph.S && j2 == 1) || (ph.W && i2 == 1);  // SK: the (i odd, j odd) column of this lane is read through a mirror
  const int ce = lane == 0 ? cm : (lane == WAVE - 1 ? cp : c0);
  const int pol = live ? po : F.HO + (C.ny - 1), pel = live ? pe : F.EO + C.ny;  // columns the loads of a lane past the row fall on
  const double a = 9. / 16., b = 3. / 16., c = 1. / 16., d = 27. / 64., e = 9. / 64., f = 3. / 64., g = 1. / 64.;
  double pc[8], pa[8], pb[8];  // fine p of the level in work and of the next one (two levels ahead measured slower: 76 vs 71 us)
  double lo[9], x[9], hi[9], nw[9];  // coarse levels k2-1, k2, k2+1 and, requested one step before its first use, k2+2
#pragma unroll
  for (int q = 0; q < 9; q++) lo[q] = hi[q] = nw[q] = 0.0;
#pragma unroll
  for (int q = 0; q < 8; q++) pa[q] = pb[q] = 0.0;
  LOADP(ka, pa) LOADP0(ka, pa)
  ROW(ka - 1, lo)  // (level 0 does not exist and is never used: clamped)
  ROW(ka, x)
  ROW(ka + 1, hi)
  // one coarse level: CUR holds its fine p, NXT receives those of level kk + 1
#define STEP(kk, CUR, NXT)                                                                                          \
  {                                                                                                                 \
    const int k2 = (kk);                                                                                            \
    LOADP(k2 + 1, NXT) LOADP0(k2 + 1, NXT)                                                                          \
    ROW(k2 + 2, nw)                                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 8; q++) pc[q] = CUR[q];                                                   \
    if (live) {                                                                                                     \
      const double xmm = x[0], x0m = x[1], xpm = x[2], xm0 = x[3], x00 = x[4], xp0 = x[5], xmp = x[6], x0p = x[7], xpp = x[8]; \
      _Pragma("unroll") for (int half = 0; half < 2; half++) {                                                      \
        const int k = 2 * k2 - 1 + half;  /* fine level */                                                          \
        if (k == 1) {                      /* bottom level: bilinear (mg_intergrids.f90:392-405) */                  \
          PUT(1, 0, +a * x00 + c * xmm + b * xm0 + b * x0m);                                                   \
          PUT(1, 1, +a * x00 + c * xpm + b * xp0 + b * x0m);                                                   \
          PUT(1, 2, +a * x00 + c * xmp + b * xm0 + b * x0p);                                                   \
          PUT(1, 3, +a * x00 + c * xpp + b * xp0 + b * x0p);                                                   \
        } else if (k == 2 * nz) {          /* top level: 1/2 bilinear (:434-446) */                                  \
          PUT(k, 0, 0.5 * (a * x00 + c * xmm + b * xm0 + b * x0m));                                            \
          PUT(k, 1, 0.5 * (a * x00 + c * xpm + b * xp0 + b * x0m));                                            \
          PUT(k, 2, 0.5 * (a * x00 + c * xmp + b * xm0 + b * x0p));                                            \
          PUT(k, 3, 0.5 * (a * x00 + c * xpp + b * xp0 + b * x0p));                                            \
        } else {                           /* interior: tri-linear, kp = k2-1 for odd k, k2+1 for even k (:407-432) */ \
          const double *__restrict__ y = half ? hi : lo;                                                            \
          const double ymm = y[0], y0m = y[1], ypm = y[2], ym0 = y[3], y00 = y[4], yp0 = y[5], ymp = y[6], y0p = y[7], ypp = y[8]; \
          PUT(k, 0, +d * x00 + f * xmm + e * xm0 + e * x0m + e * y00 + g * ymm + f * ym0 + f * y0m);           \
          PUT(k, 1, +d * x00 + f * xpm + e * xp0 + e * x0m + e * y00 + g * ypm + f * yp0 + f * y0m);           \
          PUT(k, 2, +d * x00 + f * xmp + e * xm0 + e * x0p + e * y00 + g * ymp + f * ym0 + f * y0p);           \
          PUT(k, 3, +d * x00 + f * xpp + e * xp0 + e * x0p + e * y00 + g * ypp + f * yp0 + f * y0p);           \
        }                                                                                                           \
      }                                                                                                             \
    }                                                                                                               \
    _Pragma("unroll") for (int q = 0; q < 9; q++) { lo[q] = x[q]; x[q] = hi[q]; hi[q] = nw[q]; }                    \
  }
  int kk = ka;
  for (; kk + 1 <= kb; kk += 2) {  // no branch around a step inside the loop
    STEP(kk, pa, pb)
    STEP(kk + 1, pb, pa)
  }
  if (kk <= kb) STEP(kk, pa, pb)
#undef STEP
#undef ROW
#undef PUT
#undef LOADP
#undef LOADP0
}

// ------------------------------------------------------------------------------------------------
// physical-boundary halo (homogeneous Neumann mirror), nh = 1.  mg_mpi_exchange.f90:509-537 (edges),
// :552,567,582,597 (corners where both sides are physical).  All sources are interior cells.
// grid: x = perimeter index, y = k.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halo_phys(LevView L, double *__restrict__ a, Sides ph) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const long long ro = (long long)blockIdx.y * L.RS;
  const int nx = L.nx, ny = L.ny;
  if (q < nx) {  // south and north edges of plane i
    const long long o = (long long)(q + 1) * L.plane + ro;
    if (ph.S) a[o + jpos(L, 0)] = a[o + jpos(L, 1)];
    if (ph.N) a[o + jpos(L, ny + 1)] = a[o + jpos(L, ny)];
  } else if (q < nx + ny) {  // west and east planes, j = 1..ny
    const int j = q - nx + 1, c = jpos(L, j);
    if (ph.W) a[ro + c] = a[L.plane + ro + c];
    if (ph.E) a[(long long)(nx + 1) * L.plane + ro + c] = a[(long long)nx * L.plane + ro + c];
  } else if (q == nx + ny) {
    const long long oE = (long long)(nx + 1) * L.plane + ro, oI = (long long)nx * L.plane + ro;
    if (ph.S && ph.W) a[ro + jpos(L, 0)] = a[L.plane + ro + jpos(L, 1)];
    if (ph.S && ph.E) a[oE + jpos(L, 0)] = a[oI + jpos(L, 1)];
    if (ph.N && ph.E) a[oE + jpos(L, ny + 1)] = a[oI + jpos(L, ny)];
    if (ph.N && ph.W) a[ro + jpos(L, ny + 1)] = a[L.plane + ro + jpos(L, ny)];
  }
}

// mixed corners, after the edge halos have been received (mg_mpi_exchange.f90:720-743).
// mode per corner (SW,SE,NE,NW): 0 nothing, 1 copy along i from the received S/N halo row, 2 copy along j
// from the received W/E halo plane.
__global__ void k_halo_mixed_corners(LevView L, double *__restrict__ a, int mSW, int mSE, int mNE, int mNW) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= L.nz) return;
  const long long ro = (long long)k * L.RS;
  const int nx = L.nx, ny = L.ny;
  const long long W0 = ro, W1 = L.plane + ro, E0 = (long long)(nx + 1) * L.plane + ro, E1 = (long long)nx * L.plane + ro;
  const int j0 = jpos(L, 0), j1 = jpos(L, 1), jn = jpos(L, ny), jn1 = jpos(L, ny + 1);
  if (mSW == 1) a[W0 + j0] = a[W1 + j0]; else if (mSW == 2) a[W0 + j0] = a[W0 + j1];
  if (mSE == 1) a[E0 + j0] = a[E1 + j0]; else if (mSE == 2) a[E0 + j0] = a[E0 + j1];
  if (mNE == 1) a[E0 + jn1] = a[E1 + jn1]; else if (mNE == 2) a[E0 + jn1] = a[E0 + jn];
  if (mNW == 1) a[W0 + jn1] = a[W1 + jn1]; else if (mNW == 2) a[W0 + jn1] = a[W0 + jn];
}

// pack / unpack of the 8 exchange buffers (edges nz*nx, nz*ny; corners nz).  dir: 0 S,1 E,2 N,3 W,4 SW,5 SE,6 NE,7 NW.
// pack reads the interior edge that the neighbour in direction `dir` needs; unpack writes my halo on side `dir`.
// all present directions in one launch: blockIdx.z = direction (absent ones return), buffers passed by value.
// Buffer element (q,k) of an edge sits at k*n + q (q = position along the edge, lane-contiguous), so the writes of a
// push into a neighbour GPU's memory leave each wave as whole 512-byte runs.
struct HaloBufs { double *b[8]; int present[8]; };
__device__ __forceinline__ bool halo_elem(const LevView &L, int dir, int q, int k, int unpack, long long &e, long long &t) {
  const int nx = L.nx, ny = L.ny;
  int i, j, n;
  switch (dir) {
    case 0: n = nx; i = q + 1; j = unpack ? 0 : 1; break;
    case 1: n = ny; j = q + 1; i = unpack ? nx + 1 : nx; break;
    case 2: n = nx; i = q + 1; j = unpack ? ny + 1 : ny; break;
    case 3: n = ny; j = q + 1; i = unpack ? 0 : 1; break;
    case 4: n = 1; i = unpack ? 0 : 1; j = unpack ? 0 : 1; break;
    case 5: n = 1; i = unpack ? nx + 1 : nx; j = unpack ? 0 : 1; break;
    case 6: n = 1; i = unpack ? nx + 1 : nx; j = unpack ? ny + 1 : ny; break;
    default: n = 1; i = unpack ? 0 : 1; j = unpack ? ny + 1 : ny; break;
  }
  if (q >= n) return false;
  e = (long long)i * L.plane + (long long)k * L.RS + jpos(L, j);
  t = (long long)k * n + q;
  return true;
}
__global__ void k_halo_pack_all(LevView L, double *__restrict__ a, HaloBufs hb, int unpack) {
  const int dir = blockIdx.z;
  if (!hb.present[dir]) return;
  long long e, t;
  if (!halo_elem(L, dir, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y, unpack, e, t)) return;
  double *__restrict__ buf = hb.b[dir];
  if (unpack) a[e] = buf[t]; else buf[t] = a[e];
}

// ---- peer-to-peer halo transport (xGMI, no host in the loop) --------------------------------------------------
// Push: edges are written straight into the NEIGHBOURS' receive buffers (fine-grained device memory opened through
// hipIpc), every block fences at system scope, and the last block to finish raises the sequence number in each
// neighbour's flag.  Unpack: a block spins (bounded) on the LOCAL flag of its direction, then copies the received edge
// into the halo.  Receive buffers alternate by the parity of the per-level sequence number:
// a rank cannot push exchange n+2 before it has unpacked n+1, which its neighbour pushed after unpacking n.
// bound of every wait on a peer's flag, in ticks of the 100 MHz constant clock (5 s; MGX_P2P_TIMEOUT_MS / mgxk_set_p2p_timeout shorten it for tests)
__device__ long long g_p2p_timeout_ticks = 500000000LL;
struct HaloP2P {
  int drop;                     // test hook: this launch does not raise the neighbours' flags (a rank that went silent)
  unsigned long long *flag[8];  // push: the neighbour's flag to raise; unpack: the local flag to wait on
  unsigned long long seq;
  unsigned int *counter;        // blocks-done counter of the push launch (device memory, left at 0)
  int *err;                     // host-mapped error word: 1 = a wait timed out
  int mSW, mSE, mNE, mNW;       // unpack: mixed corners (one side physical, the other a neighbour), see k_halo_mixed_corners
  int blk0[9];                  // compact 1-D grid: blocks blk0[d] .. blk0[d+1]-1 serve direction d (empty when absent)
  int ipt;                      // items per thread (> 1 only for very long edges: keeps the grid within what is resident)
};
// item w of direction d is buffer element w = k*n_d + q: consecutive lanes = consecutive buffer elements
__device__ __forceinline__ int halo_dir(const HaloP2P &pp) {
  int dir = 0;
#pragma unroll
  for (int d = 1; d < 8; d++) if ((int)blockIdx.x >= pp.blk0[d]) dir = d;
  return dir;
}
__device__ __forceinline__ bool halo_item(const LevView &L, const HaloP2P &pp, int dir, int r, int unpack, int &q, int &k, long long &e, long long &t) {
  const int n = (dir == 0 || dir == 2) ? L.nx : ((dir == 1 || dir == 3) ? L.ny : 1);
  const int w = (((int)blockIdx.x - pp.blk0[dir]) * pp.ipt + r) * blockDim.x + threadIdx.x;
  if (w >= n * L.nz) return false;
  k = w / n; q = w - k * n;
  return halo_elem(L, dir, q, k, unpack, e, t);
}
// One launch per halo fill: every block first pushes its part of direction d into the neighbour's receive buffer, the last
// block to finish pushing raises the neighbours' flags, then every block waits (bounded) on the LOCAL flag of its
// direction and unpacks the same part of the edge it received.  The grid is at most ~128 blocks (compact, present
// directions only, several items per thread on long edges), well below what the GPU keeps resident, so a block that
// spins never keeps a pushing block from starting; should that ever fail the 5 s time-out turns it into an error.
struct HaloXchg { double *rbuf[8]; double *lbuf[8]; unsigned long long *rflag[8]; unsigned long long *lflag[8]; int present[8]; };
__global__ __launch_bounds__(256) void k_halo_exchange(LevView L, double *__restrict__ a, HaloXchg hx, HaloP2P pp) {
  const int dir = halo_dir(pp);  // block-uniform
  int q, k; long long e, t;
  for (int r = 0; r < pp.ipt; r++)
    if (halo_item(L, pp, dir, r, 0, q, k, e, t)) hx.rbuf[dir][t] = a[e];
  // every storing wave drains its remote writes, the block meets, ONE lane issues the system-scope release before the block reports in
  // (256 lanes fencing cost 2-4x one lane's: MI355X_MICROARCH.md, inter-workgroup visibility)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__hip_atomic_fetch_add(pp.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
      __hip_atomic_store(pp.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int d = 0; d < 8; d++)
        if (hx.present[d] && !pp.drop) __hip_atomic_store(hx.rflag[d], pp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    ok = 0;
    const long long t0 = wall_clock64(), tmax = g_p2p_timeout_ticks;
    while (true) {
      if (__hip_atomic_load(hx.lflag[dir], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= pp.seq) { ok = 1; break; }
      if (wall_clock64() - t0 > tmax) break;  // 5 s of the 100 MHz constant clock: the neighbour is gone
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  if (!ok) { if (threadIdx.x == 0) *pp.err = 1; return; }
  __threadfence_system();
  const int nx = L.nx, ny = L.ny;
  for (int r = 0; r < pp.ipt; r++) {
    if (!halo_item(L, pp, dir, r, 1, q, k, e, t)) continue;
    const double v = __builtin_nontemporal_load(hx.lbuf[dir] + t);
    a[e] = v;
    // mixed corners (mg_mpi_exchange.f90:720-743): the corner next to a physical side mirrors the edge halo cell that was
    // just received -- written here by the thread that unpacked that cell instead of a separate launch
    const long long ro = (long long)k * L.RS, W0 = ro, E0 = (long long)(nx + 1) * L.plane + ro;
    if (dir == 0) { if (q == 0 && pp.mSW == 1) a[W0 + jpos(L, 0)] = v; if (q == nx - 1 && pp.mSE == 1) a[E0 + jpos(L, 0)] = v; }
    else if (dir == 2) { if (q == 0 && pp.mNW == 1) a[W0 + jpos(L, ny + 1)] = v; if (q == nx - 1 && pp.mNE == 1) a[E0 + jpos(L, ny + 1)] = v; }
    else if (dir == 3) { if (q == 0 && pp.mSW == 2) a[W0 + jpos(L, 0)] = v; if (q == ny - 1 && pp.mNW == 2) a[W0 + jpos(L, ny + 1)] = v; }
    else if (dir == 1) { if (q == 0 && pp.mSE == 2) a[E0 + jpos(L, 0)] = v; if (q == ny - 1 && pp.mNE == 2) a[E0 + jpos(L, ny + 1)] = v; }
  }
}

// ------------------------------------------------------------------------------------------------
// layout conversion between the reference layout (k,j,i) k fastest and JS.  dir 0: ref -> JS, 1: JS -> ref
// ------------------------------------------------------------------------------------------------
__global__ void k_convert(LevView L, double *__restrict__ js, double *__restrict__ ref, int nslot, int slot, int dir) {
  const long long n = (long long)L.nz * (L.ny + 2) * (L.nx + 2);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int k = (int)(t % L.nz);
  const long long ji = t / L.nz;
  const int j = (int)(ji % (L.ny + 2)), i = (int)(ji / (L.ny + 2));
  const long long e = (long long)i * L.plane + (long long)k * L.RS + jpos(L, j);
  if (dir == 0) js[e] = ref[t * nslot + slot]; else ref[t * nslot + slot] = js[e];
}

// Set-up scratch -> JS for NS slot-major arrays (nz, 0:ny+1, 0:nx+1) at ref, ref + n3, ... (the eight coefficient slots; the two slope
// arrays).  A block takes one slot (blockIdx.z), one plane and TJ consecutive columns: one contiguous run of TJ*nz doubles, staged in
// LDS with the even and the odd columns apart ([parity][column pair][k], column stride padded by one double), then written row by row:
// for every k one wave stores TJ/2 consecutive even-j entries and one TJ/2 consecutive odd-j entries -- 512-byte runs at TJ = 128
// (the first version moved all eight slots of 16 columns per block: 64-byte runs, 2.1 TB/s at 512x512x64).
struct Slots8 { double *s[8]; };
__global__ __launch_bounds__(256) void k_convert_slots(LevView L, Slots8 out, const double *__restrict__ ref, int TJ) {
  extern __shared__ double lds[];
  const int nz = L.nz, cs = nz + 1, H = (TJ >> 1) * cs;
  const int i = blockIdx.y, j0 = blockIdx.x * TJ, sl = blockIdx.z;
  const int nj = min(TJ, L.ny + 2 - j0);
  const long long n3 = (long long)nz * (L.ny + 2) * (L.nx + 2);
  const double *__restrict__ src = ref + (long long)sl * n3 + ((long long)i * (L.ny + 2) + j0) * nz;
  const int run = nj * nz;
  const bool p2 = (nz & (nz - 1)) == 0;
  const int lz = 31 - __builtin_clz(nz);
  // eight loads in flight per thread before the first LDS store (one load per iteration made the block a chain of 32 memory round trips)
  constexpr int U = 8;
  const int nt = blockDim.x;
  for (int r0 = threadIdx.x; r0 < run; r0 += U * nt) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int r = r0 + u * nt; v[u] = r < run ? src[r] : 0.0; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int r = r0 + u * nt;
      if (r < run) { const int jl = p2 ? (r >> lz) : r / nz, k = r - jl * nz; lds[(jl & 1) * H + (jl >> 1) * cs + k] = v[u]; }
    }
  }
  __syncthreads();
  double *__restrict__ os = out.s[sl] + (long long)i * L.plane;
  const int hj = TJ >> 1;  // column pairs per tile; j0 is even (TJ is), so local parity = global parity
  for (int t = threadIdx.x; t < TJ * nz; t += nt) {
    const int q = t % hj, par = (t / hj) & 1, k = t / TJ;
    const int jl = 2 * q + par;
    if (jl < nj) os[(long long)k * L.RS + (par ? L.HO : L.EO) + ((j0 + jl) >> 1)] = lds[par * H + q * cs + k];
  }
}

// gather (mg_gather.f90:95-174): copy the interior of member block (l,m) (reference layout incl. halo, as received)
// into the JS coarse b at offset (l*nxc, m*nyc)
__global__ void k_gather_place(LevView C, double *__restrict__ dstjs, const double *__restrict__ blk, int nxc, int nyc, int l, int m) {
  const long long n = (long long)C.nz * nyc * nxc;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int k = (int)(t % C.nz);
  const long long ji = t / C.nz;
  const int j = 1 + (int)(ji % nyc), i = 1 + (int)(ji / nyc);
  const double v = blk[((long long)i * (nyc + 2) + j) * C.nz + k];
  dstjs[(long long)(i + l * nxc) * C.plane + (long long)k * C.RS + jpos(C, j + m * nyc)] = v;
}
// pre-gather block (JS level view Cs of the small block) -> contiguous reference-layout block incl. halo
__global__ void k_block_to_ref(LevView Cs, const double *__restrict__ js, double *__restrict__ blk) {
  const long long n = (long long)Cs.nz * (Cs.ny + 2) * (Cs.nx + 2);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int k = (int)(t % Cs.nz);
  const long long ji = t / Cs.nz;
  const int j = (int)(ji % (Cs.ny + 2)), i = (int)(ji / (Cs.ny + 2));
  blk[t] = js[(long long)i * Cs.plane + (long long)k * Cs.RS + jpos(Cs, j)];
}
// Peer-to-peer gather (same protocol as the halo pushes): my restricted block, converted to the reference layout on
// the fly, is written into slot `me` of every group member's gather buffer (mine included); the last block raises my
// flag at the other members.  k_gather_place_wait then waits for member q's flag before placing its block.
struct GatherP2P { double *dst[4]; unsigned long long *flag[4]; int ng, me; unsigned long long seq; unsigned int *counter; int *err; };
__global__ void k_gather_push(LevView Cs, const double *__restrict__ js, GatherP2P gp) {
  // block layout in the gather buffers: (i, k, j) with j fastest (halo included) -- lanes run along j on both sides
  const long long n = (long long)Cs.nz * (Cs.ny + 2) * (Cs.nx + 2);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(t % (Cs.ny + 2));
    const long long ik = t / (Cs.ny + 2);
    const int k = (int)(ik % Cs.nz), i = (int)(ik / Cs.nz);
    const double v = js[(long long)i * Cs.plane + (long long)k * Cs.RS + jpos(Cs, j)];
    for (int q = 0; q < gp.ng; q++) gp.dst[q][t] = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // as k_halo_exchange: waves drain, the block meets, one lane releases at system scope
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__hip_atomic_fetch_add(gp.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
      __hip_atomic_store(gp.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int q = 0; q < gp.ng; q++)
        if (q != gp.me) __hip_atomic_store(gp.flag[q], gp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
__global__ void k_gather_place_wait(LevView C, double *__restrict__ dstjs, const double *__restrict__ blk, int nxc, int nyc, int l, int m,
                                    unsigned long long *flag, unsigned long long seq, int *err) {
  if (flag) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
      ok = 0;
      const long long t0 = wall_clock64(), tmax = g_p2p_timeout_ticks;
      while (true) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= seq) { ok = 1; break; }
        if (wall_clock64() - t0 > tmax) break;
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) *err = 1; return; }
    __threadfence_system();
  }
  const long long n = (long long)C.nz * nyc * nxc;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int j = 1 + (int)(t % nyc);
    const long long ik = t / nyc;
    const int k = (int)(ik % C.nz), i = 1 + (int)(ik / C.nz);
    const double v = __builtin_nontemporal_load(blk + ((long long)i * C.nz + k) * (nyc + 2) + j);
    dstjs[(long long)(i + l * nxc) * C.plane + (long long)k * C.RS + jpos(C, j + m * nyc)] = v;
  }
}
// split (mg_gather.f90:177-220): own quadrant of the gathered p, halo included, into the small JS block
__global__ void k_split(LevView C, LevView Cs, const double *__restrict__ pc, double *__restrict__ dst, int l, int m) {
  const long long n = (long long)Cs.nz * (Cs.ny + 2) * (Cs.nx + 2);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int k = (int)(t % Cs.nz);
  const long long ji = t / Cs.nz;
  const int j = (int)(ji % (Cs.ny + 2)), i = (int)(ji / (Cs.ny + 2));
  dst[(long long)i * Cs.plane + (long long)k * Cs.RS + jpos(Cs, j)] =
      pc[(long long)(i + l * Cs.nx) * C.plane + (long long)k * C.RS + jpos(C, j + m * Cs.ny)];
}


// ------------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------------
// self-test of the constant-divisor quotient (mgx_device.h, DIVC): a / b by the hardware sequence against the refined-reciprocal form
__global__ void k_divc_selftest(const double *__restrict__ a, const double *__restrict__ b, int n, unsigned long long *bad) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double q = a[t] / b[t], rb = RCP_REF(b[t]), c = DIVC(a[t], b[t], rb);
  if (__double_as_longlong(q) != __double_as_longlong(c)) atomicAdd(bad, 1ull);
}

extern "C" {

int mgxk_residual_nblocks(const LevView *L) { dim3 g = col_grid(L->ny / 2, L->nx, 2); return g.x * g.y * g.z; }
void mgxk_residual(hipStream_t st, const LevView *L, double *partial, double *out, int real, int want_norm, Sides ph) {
  dim3 blk(WAVE, 4), g3 = col_grid(L->ny / 2, L->nx, 2), grd(g3.x * g3.y * 2);
  const int gx = g3.x, gy = g3.y;
  if (L->zy != nullptr && L->nz >= 3) {
    if (real) hipLaunchKernelGGL((k_residual_mf<true>), grd, blk, 0, st, *L, partial, want_norm, gx, gy, ph, level_streams(L));
    else hipLaunchKernelGGL((k_residual_mf<false>), grd, blk, 0, st, *L, partial, want_norm, gx, gy, ph, level_streams(L));
  } else if (real) hipLaunchKernelGGL((k_residual<true>), grd, blk, 0, st, *L, partial, want_norm, gx, gy, ph, level_streams(L));
  else hipLaunchKernelGGL((k_residual<false>), grd, blk, 0, st, *L, partial, want_norm, gx, gy, ph, level_streams(L));
  if (want_norm) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, partial, (int)grd.x, out);
}
// second stage of a norm whose partials another kernel wrote (one per workgroup, summed in index order)
void mgxk_reduce(hipStream_t st, const double *partial, int n, double *out) { hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, partial, n, out); }
void mgxk_sumsq(hipStream_t st, const LevView *L, const double *a, double *partial, double *out) {
  dim3 blk(WAVE, 4), grd = col_grid(L->ny / 2, L->nx, 2);
  hipLaunchKernelGGL(k_sumsq, grd, blk, 0, st, *L, a, partial);
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, partial, (int)(grd.x * grd.y * grd.z), out);
}
void mgxk_dot(hipStream_t st, const LevView *L, const double *a, const double *b, double *partial, double *out) {
  dim3 blk(WAVE, 4), grd = col_grid(L->ny / 2, L->nx, 2);
  hipLaunchKernelGGL(k_dot, grd, blk, 0, st, *L, a, b, partial);
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, partial, (int)(grd.x * grd.y * grd.z), out);
}
// levels[0 .. dep]: the views of lev, lev+1, ... (dep restrictions, 1 <= dep <= 4); every level closed, not gathered, level 0's nx, ny, nz multiples of 2^dep
void mgxk_restrict_chain(hipStream_t st, const LevView *const *levels, int dep, Sides ph) {
  LevChain Ch;
  for (int q = 0; q <= dep && q < 5; q++) Ch.v[q] = *levels[q];
  for (int q = dep + 1; q < 5; q++) Ch.v[q] = *levels[dep];
  const LevView &F = Ch.v[0];
  const int E = 1 << dep;
  const dim3 grd((unsigned)((F.nx / E) * (F.ny / E) * (F.nz / E)));
  switch (dep) {
    case 1: hipLaunchKernelGGL(k_restrict_chain<1>, grd, dim3(256), 0, st, Ch, ph); break;
    case 2: hipLaunchKernelGGL(k_restrict_chain<2>, grd, dim3(256), 0, st, Ch, ph); break;
    case 3: hipLaunchKernelGGL(k_restrict_chain<3>, grd, dim3(256), 0, st, Ch, ph); break;
    default: hipLaunchKernelGGL(k_restrict_chain<4>, grd, dim3(256), 0, st, Ch, ph); break;
  }
}
void mgxk_fine2coarse(hipStream_t st, const LevView *F, const LevView *C, double *dst, Sides ph, double *dup, double *zero) {
  dim3 grd = col_grid(C->ny, C->nx);
  const long long waves = (long long)grd.x * grd.y * 4;
  long long nchunk = (8192 + waves - 1) / waves;  // enough waves to hide the latency of the eight loads of a level
  if (nchunk > C->nz) nchunk = C->nz;
  if (nchunk < 1) nchunk = 1;
  const int KC = (int)((C->nz + nchunk - 1) / nchunk);
  grd.z = (C->nz + KC - 1) / KC;
  hipLaunchKernelGGL(k_fine2coarse, grd, dim3(WAVE, 4), 0, st, *F, *C, dst, ph, level_streams(F), dup, zero, KC);
}
// skip1: the caller guarantees that a four-colour relax of the fine level follows (cycles only; see k_coarse2fine_run, SK)
void mgxk_coarse2fine(hipStream_t st, const LevView *F, const LevView *C, const double *src, int linear, Sides ph, int keep_r, int skip1) {
  static const bool norun = getenv("MGX_C2F_OLD") != nullptr;
  if (linear && !norun) {
    // runs of KC coarse levels per lane: long enough to amortise the three-level window, short enough to keep >= ~2 waves per SIMD
    static const int kcenv = getenv("MGX_C2F_KC") ? atoi(getenv("MGX_C2F_KC")) : 0;
    int KC = kcenv > 0 ? kcenv : 16;
    const long long wav = (long long)((C->ny + WAVE - 1) / WAVE) * C->nx;
    while (KC > 1 && wav * ((C->nz + KC - 1) / KC) < 2048) KC >>= 1;
    if (KC > C->nz) KC = C->nz;
    const int nrun = (C->nz + KC - 1) / KC, byr = nrun >= 4 ? 4 : nrun;
    dim3 blk(WAVE, byr), grd((C->ny + WAVE - 1) / WAVE, (nrun + byr - 1) / byr, C->nx);
    static const int ntenv = getenv("MGX_C2F_NT") ? atoi(getenv("MGX_C2F_NT")) : -1;  // A/B: force the streaming hints on / off
    const int nt = ntenv >= 0 ? ntenv : level_streams(F);
    static const bool nosk = getenv("MGX_C2F_NOSKIP") != nullptr;
    if (keep_r) hipLaunchKernelGGL((k_coarse2fine_run<true, false>), grd, blk, 0, st, *F, *C, src, ph, nt, KC);
    else if (skip1 && !nosk && !(F->nx & 1) && !(F->ny & 1)) hipLaunchKernelGGL((k_coarse2fine_run<false, true>), grd, blk, 0, st, *F, *C, src, ph, nt, KC);
    else hipLaunchKernelGGL((k_coarse2fine_run<false, false>), grd, blk, 0, st, *F, *C, src, ph, nt, KC);
    return;
  }
  const int by = C->nz >= 4 ? 4 : C->nz;
  dim3 blk(WAVE, by), grd((C->ny + WAVE - 1) / WAVE, (C->nz + by - 1) / by, C->nx);
#define C2F(LIN, WRV) hipLaunchKernelGGL((k_coarse2fine<LIN, WRV>), grd, blk, 0, st, *F, *C, src, ph, level_streams(F))
  if (linear) { if (keep_r) C2F(true, true); else C2F(true, false); }
  else { if (keep_r) C2F(false, true); else C2F(false, false); }
#undef C2F
}
void mgxk_divc_selftest(hipStream_t st, const double *a, const double *b, int n, unsigned long long *bad) {
  hipLaunchKernelGGL(k_divc_selftest, dim3((n + 255) / 256), dim3(256), 0, st, a, b, n, bad);
}
void mgxk_halo_phys(hipStream_t st, const LevView *L, double *a, Sides ph) {
  const int n = L->nx + L->ny + 1;
  hipLaunchKernelGGL(k_halo_phys, dim3((n + 255) / 256, L->nz), dim3(256), 0, st, *L, a, ph);
}
void mgxk_halo_mixed_corners(hipStream_t st, const LevView *L, double *a, int mSW, int mSE, int mNE, int mNW) {
  hipLaunchKernelGGL(k_halo_mixed_corners, dim3((L->nz + 63) / 64), dim3(64), 0, st, *L, a, mSW, mSE, mNE, mNW);
}
void mgxk_halo_pack_all(hipStream_t st, const LevView *L, double *a, double *const *bufs, const int *present, int unpack) {
  HaloBufs hb;
  for (int d = 0; d < 8; d++) { hb.b[d] = bufs[d]; hb.present[d] = present[d]; }
  const int n = L->nx > L->ny ? L->nx : L->ny;
  hipLaunchKernelGGL(k_halo_pack_all, dim3((n + 63) / 64, L->nz, 8), dim3(64), 0, st, *L, a, hb, unpack);
}
void mgxk_halo_p2p(hipStream_t st, const LevView *L, double *a, double *const *rbuf, double *const *lbuf, unsigned long long *const *rflag,
                   unsigned long long *const *lflag, const int *present, unsigned long long seq, unsigned int *counter, int *err, const int *mixed, int drop) {
  HaloXchg hx; HaloP2P pp;
  pp.drop = drop;
  // Every block both pushes and waits, so all of them must be resident together, and a waiting wave, however light, keeps
  // a 512-VGPR smoother wave off its SIMD: harmless when the GPU belongs to one rank (the smoother is behind us in the
  // stream), fatal when several ranks share one GPU as in the tests (the neighbour's smoother would never start).  So the
  // grid is kept small -- at most 128 blocks = 512 of the 1024 SIMDs -- and longer edges give each thread several items.
  static const int maxblk = getenv("MGX_P2P_MAXBLK") ? atoi(getenv("MGX_P2P_MAXBLK")) : 128;
  long long items = 0;
  for (int d = 0; d < 8; d++) if (present[d]) items += (long long)L->nz * ((d == 0 || d == 2) ? L->nx : ((d == 1 || d == 3) ? L->ny : 1));
  pp.ipt = (int)((items + 256LL * maxblk - 1) / (256LL * maxblk)) + 1;  // +1: per-direction round-up never exceeds maxblk
  static const int ipt_min = getenv("MGX_P2P_IPT") ? atoi(getenv("MGX_P2P_IPT")) : 1;  // test hook for the multi-item path
  if (pp.ipt < ipt_min) pp.ipt = ipt_min;
  const int per = 256 * pp.ipt;
  int nb = 0;
  for (int d = 0; d < 8; d++) {
    hx.rbuf[d] = rbuf[d]; hx.lbuf[d] = lbuf[d]; hx.rflag[d] = rflag[d]; hx.lflag[d] = lflag[d]; hx.present[d] = present[d];
    pp.flag[d] = nullptr;
    pp.blk0[d] = nb;
    if (present[d]) nb += (L->nz * ((d == 0 || d == 2) ? L->nx : ((d == 1 || d == 3) ? L->ny : 1)) + per - 1) / per;
  }
  pp.blk0[8] = nb;
  // halo_item picks the LAST d with blk0[d] <= block index: absent directions are moved past the end
  for (int d = 7; d >= 0; d--) if (!present[d]) pp.blk0[d] = nb + 1;
  pp.seq = seq; pp.counter = counter; pp.err = err;
  pp.mSW = mixed[0]; pp.mSE = mixed[1]; pp.mNE = mixed[2]; pp.mNW = mixed[3];
  if (nb == 0) return;
  hipLaunchKernelGGL(k_halo_exchange, dim3(nb), dim3(256), 0, st, *L, a, hx, pp);
}
// the bound of every p2p flag wait, in milliseconds (device global of this library; all instances of the process)
int mgxk_set_p2p_timeout(double ms) {
  const long long ticks = (long long)(ms * 1e5);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_p2p_timeout_ticks), &ticks, sizeof ticks) == hipSuccess ? 0 : 1;
}
// out[0] = 1.0 when the host-mapped error word is set or `extra` is non-zero, else 0.0 -- in stream order, so that a time-out of a kernel
// still in flight reaches the all-reduce that follows (collective agreement on the health of the peer-to-peer transport)
__global__ void k_err_to_double(const int *err, int extra, double *out) { out[0] = (err && *err != 0) || extra ? 1.0 : 0.0; }
void mgxk_err_to_double(hipStream_t st, const int *err, int extra, double *out) { hipLaunchKernelGGL(k_err_to_double, dim3(1), dim3(1), 0, st, err, extra, out); }
void mgxk_convert(hipStream_t st, const LevView *L, double *js, double *ref, int nslot, int slot, int dir) {
  const long long n = (long long)L->nz * (L->ny + 2) * (L->nx + 2);
  hipLaunchKernelGGL(k_convert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *L, js, ref, nslot, slot, dir);
}
// tile width of k_convert_slots: 128 columns while the tile fits the LDS (nz <= 128), else the largest multiple of 16 that does
static int convert_tile(const LevView *L) {
  int tj = 128;
  while (tj > 16 && (size_t)tj * (L->nz + 1) * sizeof(double) > 150 * 1024) tj -= 16;
  return tj;
}
static void convert_slots(hipStream_t st, const LevView *L, const Slots8 &o, int ns, const double *ref) {
  const int tj = convert_tile(L);
  const size_t lds = (size_t)tj * (L->nz + 1) * sizeof(double);
  // > 64 KB from nz = 64 on; a refusal shows up as a launch error at the next synchronising call (sync_stream)
  static bool attr = false;
  // a refused attribute makes the launch below fail, and THAT is reported by the next synchronising call; the attribute call's own status is consumed here
  if (!attr) { if (hipFuncSetAttribute((const void *)k_convert_slots, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError(); attr = true; }
  hipLaunchKernelGGL(k_convert_slots, dim3((L->ny + 2 + tj - 1) / tj, L->nx + 2, ns), dim3(256), lds, st, *L, o, ref, tj);
}
void mgxk_convert8(hipStream_t st, const LevView *L, const double *ref) {
  Slots8 o;
  for (int q = 0; q < 8; q++) o.s[q] = L->cA[q];
  convert_slots(st, L, o, 8, ref);
}
// two slot-major arrays (nz, 0:ny+1, 0:nx+1) at ref, ref + n3 -> the JS arrays out0, out1 (the slopes zy, zx)
void mgxk_convert2(hipStream_t st, const LevView *L, double *out0, double *out1, const double *ref) {
  Slots8 o;
  for (int q = 0; q < 8; q++) o.s[q] = nullptr;
  o.s[0] = out0; o.s[1] = out1;
  convert_slots(st, L, o, 2, ref);
}
void mgxk_gather_place(hipStream_t st, const LevView *C, double *dstjs, const double *blk, int nxc, int nyc, int l, int m) {
  const long long n = (long long)C->nz * nyc * nxc;
  hipLaunchKernelGGL(k_gather_place, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *C, dstjs, blk, nxc, nyc, l, m);
}
void mgxk_block_to_ref(hipStream_t st, const LevView *Cs, const double *js, double *blk) {
  const long long n = (long long)Cs->nz * (Cs->ny + 2) * (Cs->nx + 2);
  hipLaunchKernelGGL(k_block_to_ref, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *Cs, js, blk);
}
void mgxk_gather_push(hipStream_t st, const LevView *Cs, const double *js, double *const *dst, unsigned long long *const *flags, int ng, int me,
                      unsigned long long seq, unsigned int *counter, int *err) {
  GatherP2P gp;
  for (int q = 0; q < 4; q++) { gp.dst[q] = q < ng ? dst[q] : nullptr; gp.flag[q] = q < ng ? flags[q] : nullptr; }
  gp.ng = ng; gp.me = me; gp.seq = seq; gp.counter = counter; gp.err = err;
  const long long n = (long long)Cs->nz * (Cs->ny + 2) * (Cs->nx + 2);
  long long nb = (n + 255) / 256;
  if (nb > 256) nb = 256;  // every wave ends with a system-scope fence and every block with an atomic: few, fat blocks
  hipLaunchKernelGGL(k_gather_push, dim3((unsigned)nb), dim3(256), 0, st, *Cs, js, gp);
}
void mgxk_gather_place_wait(hipStream_t st, const LevView *C, double *dstjs, const double *blk, int nxc, int nyc, int l, int m,
                            unsigned long long *flag, unsigned long long seq, int *err) {
  const long long n = (long long)C->nz * nyc * nxc;
  long long nb = (n + 255) / 256;
  if (flag && nb > 128) nb = 128;  // blocks that may spin stay few (see mgxk_halo_p2p)
  hipLaunchKernelGGL(k_gather_place_wait, dim3((unsigned)nb), dim3(256), 0, st, *C, dstjs, blk, nxc, nyc, l, m, flag, seq, err);
}
void mgxk_split(hipStream_t st, const LevView *C, const LevView *Cs, const double *pc, double *dst, int l, int m) {
  const long long n = (long long)Cs->nz * (Cs->ny + 2) * (Cs->nx + 2);
  hipLaunchKernelGGL(k_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *C, *Cs, pc, dst, l, m);
}

}  // extern "C"
