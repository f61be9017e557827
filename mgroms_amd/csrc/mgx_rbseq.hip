// Red-black smoothing in the reference's SEQUENTIAL order at streaming speed (option "rb_seq"; mg_relax.f90:151-190 with cmatrix='real').
//
// The reference sweeps the columns of a colour plane after plane (do i; do j, mg_relax.f90:173-176).  With cmatrix='real' a column's
// bottom row reads four neighbours of its OWN colour (:271-276): p(1,j+-1,i-1), which the loop has already updated, and p(1,j+-1,i+1),
// which it has not.  That is the only coupling between columns of one colour, it acts on the right-hand side of row k = 1 only, it comes
// from plane i-1 only, and the column solve is linear in its right-hand side.  So, per colour:
//   (a) the parallel colour pass (every kernel of mgx_relax*.hip, k=1 diagonals read from the snapshot p1 = "old" everywhere) gives y;
//   (b) d0 = y(1) - p1 (k_rbseq_d0), then ONE wave walks the planes i = 1..nx (k_rbseq_scan):
//         u(j,i) = d0(j,i) - ag5(j,i) u(j+1,i-1) - ag8(j,i) u(j-1,i-1),   ag5|8 = g(1,j,i) cA(5|8,1,j,i),  g = T^-1 e1  (stored as pairs: ag58)
//       u = new minus old bottom value of the column in the sequential order (zero in the halo: a halo cell is refreshed after the
//       colour, mg_relax.f90:181, on one rank and on several alike);
//   (c) p(:,j,i) = y(:,j,i) + g(:,j,i) s(j,i),  s = -cA(5,1,j,i) u(j+1,i-1) - cA(8,1,j,i) u(j-1,i-1)   (k_rbseq_apply, with the mirrors).
// Not bit-identical to the sequential loop (the same sum in another association: a few ulp), far inside north_star's 1e-10; the
// bit-exact order stays available as "rb_exact" (one launch per plane) and is what the tests compare against.
//
// Why one wave: the recurrence is sequential in i by nature (the cone of (j,i) widens by one column per plane in both directions), a
// step is two fused multiply-adds per column; the wave keeps D planes of operands in flight in registers (CPL columns per lane,
// contiguous, so the neighbours j+-1 are the lane's own registers but for one value that crosses to the next lane by a DPP wave
// shift).  What a plane step costs is its instruction issue (~190 cycles: 4-8 memory instructions, the DPP + FMA chain) and, where the
// operands live in HBM (level 1), the latency that at most 63 requests in flight per wave leave uncovered -- helper workgroups pull them
// into the walking wave's L2 (k_rbseq_scan).
#include <cstdlib>

#include "mgx_device.h"

// g = T^-1 e1 of every interior column with tridiag's own recurrences (mg_relax.f90:320-332; bet, gam from k_pivots), and the two
// multipliers of the scan.  One thread per column, lanes along j.
__global__ void k_rbseq_setup(LevView L) {
  const int jj = 1 + blockIdx.x * blockDim.x + threadIdx.x, i = 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (jj > L.ny || i > L.nx) return;
  const int c = jpos(L, jj);
  const long long o = (long long)i * L.plane + c;
  const double *__restrict__ dd = L.cA[1];
  double x = L.bet[o];  // xc(1) = b(1)*bet, b = e1
  L.gk[o] = x;
  for (int k = 2; k <= L.nz; k++) {
    const long long ko = o + (long long)(k - 1) * L.RS;
    x = (0.0 - dd[ko] * x) * L.bet[ko];
    L.gk[ko] = x;
  }
  for (int k = L.nz - 1; k >= 1; k--) {
    const long long ko = o + (long long)(k - 1) * L.RS;
    x = L.gk[ko] - L.gam[ko + L.RS] * x;
    L.gk[ko] = x;
  }
  const long long q = (long long)i * L.RS + c;
  L.ag58[2 * q] = x * L.cA[4][o];       // the two multipliers of a column side by side: one 16-byte request in the walk
  L.ag58[2 * q + 1] = x * L.cA[7][o];
}

// position of the first column of the colour inside a row of plane i: odd j -> HO + jh, even j -> EO + 1 + jh  (jh = 0..ny/2-1)
__device__ __forceinline__ int rb_jodd(int i, int rb) { return ((i + rb) & 1) == 0; }  // mg_relax.f90:174: j = 1+mod(i+rb,2),ny,2

// d0 = y(k=1) - snapshot, for the columns of colour rb
__global__ void k_rbseq_d0(LevView L, int rb) {
  const int jh = blockIdx.x * WAVE + threadIdx.x, i = 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (jh >= (L.ny >> 1) || i > L.nx) return;
  const int c = rb_jodd(i, rb) ? L.HO + jh : L.EO + 1 + jh;
  const long long q = (long long)i * L.RS + c;
  L.u1[q] = L.p[(long long)i * L.plane + c] - L.p1[q];
}

template <int N> struct VecD { double v[N]; };

// The walk over the planes.  NW waves share a half-row, CPL columns per lane (wave w: columns w*64*CPL ...); D planes of operands in
// flight per wave (nx is a multiple of D: the wrapper picks D);
// RBP = rb & 1 makes the j parity of a plane a compile-time property of its ring slot (planes start at the odd i = 1, D is even);
// FULL: the colour's half-row is exactly NW*64*CPL columns (vector accesses, no lane predicates);
// D0IN: d0 = p(k=1) - snapshot is formed here (one launch less; a fourth stream) instead of read from u1 (k_rbseq_d0).
// Memory-level parallelism is what bounds the walk on the large levels: a wave holds at most 63 vector-memory operations in flight
// (vmcnt), i.e. D * (loads + stores per plane) must stay below that, and every step ends in a scheduling barrier -- without it the
// compiler sinks all loads of a loop trip behind its last step and the trip waits for a full memory latency (measured: 99 us for the
// 512 planes of level 1).  With NW > 1 the two values that cross between neighbouring waves go through LDS, one barrier per plane.
template <int CPL, int D, int RBP, bool FULL, int NW, bool D0IN>
__global__ __launch_bounds__(64 * NW) void k_rbseq_scan(LevView L, int nhelp, int rb) {
  if (blockIdx.x != 0) {
    // Helper workgroups (nhelp of them, those with blockIdx % 8 == 0: dealt to the walking workgroup's XCD).  Every colour pass is a kernel
    // boundary, after which the walk's operands come from the Infinity Cache / HBM (~1.4 us per request: 16 planes of look-ahead make
    // 86 ns per plane, measured); the helpers ask for the same lines, many at a time, so that the walking wave finds them in the L2 it
    // shares with them.  Speed only: nothing depends on where a workgroup lands or on whether a line is still there.
    if ((blockIdx.x & 7) != 0) return;
    const int h = (blockIdx.x >> 3) - 1, per = (L.nx + nhelp - 1) / nhelp, nyh_ = L.ny >> 1;
    const int ia = 1 + h * per, ib = ia + per - 1 < L.nx ? ia + per - 1 : L.nx;
    double acc = 0.0;
    for (int i = ia; i <= ib; i++) {
      const int off = rb_jodd(i, rb) ? L.HO : L.EO + 1;
      const long long q = (long long)i * L.RS + off;
      for (int t = threadIdx.x * 2; t < nyh_; t += 2 * 64 * NW) {   // 16-byte requests; a half-row is a multiple of 2 columns and starts 16-byte aligned
        double2 a, b, c, d;
        if (D0IN) { __builtin_memcpy(&a, L.p + (long long)i * L.plane + off + t, 16); __builtin_memcpy(&b, L.p1 + q + t, 16); }
        else { __builtin_memcpy(&a, L.u1 + q + t, 16); b = a; }
        __builtin_memcpy(&c, L.ag58 + 2 * (q + t), 16); __builtin_memcpy(&d, L.ag58 + 2 * (q + t) + 2, 16);
        acc += a.x + b.y + c.x + d.y;
      }
    }
    if (acc == 1.2345678e-301) L.u1[0] = acc;   // never: keeps the requests alive (row 0 of u1 is halo and stays zero)
    return;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nyh = L.ny >> 1, nx = L.nx;
  const int jh0 = (wv * 64 + lane) * CPL;
  const unsigned int ujh0 = (unsigned int)jh0;
  const long long RS = L.RS;
  double *__restrict__ u1 = L.u1;
  const double *__restrict__ g58 = L.ag58, *__restrict__ p = L.p, *__restrict__ p1 = L.p1;
  __shared__ double edges[2][NW > 1 ? NW : 1][2];
  VecD<CPL> rd[D], ro[D0IN ? D : 1];
  VecD<2 * CPL> r58[D];   // (ag5, ag8) of the lane's columns
  double up[CPL];
  bool ok[CPL];
  int jc[CPL];
#pragma unroll
  for (int q = 0; q < CPL; q++) { up[q] = 0.0; ok[q] = jh0 + q < nyh; jc[q] = ok[q] ? jh0 + q : nyh - 1; }
  // every request is unconditional (planes past nx clamp to nx, lanes past the row to its last column): the waits stay counted
#define LOADP(ip, slot)                                                                                          \
  {                                                                                                              \
    const int i_ = (ip) <= nx ? (ip) : nx;                                                                       \
    const int off_ = (((((slot) + 1 + RBP) & 1) == 0) ? L.HO : L.EO + 1);                                        \
    const long long q_ = (long long)i_ * RS + off_, qp_ = (long long)i_ * L.plane + off_;                        \
    if (FULL) {                                                                                                  \
      /* wave-uniform row pointers (scalar registers) + the lane's 32-bit column offset: one address per request, no 64-bit vector arithmetic */ \
      const double *rp_ = p + qp_, *r1_ = p1 + q_, *ru_ = u1 + q_, *rg_ = g58 + 2 * q_;                           \
      if (D0IN) { __builtin_memcpy(&rd[slot], rp_ + ujh0, sizeof(VecD<CPL>)); __builtin_memcpy(&ro[D0IN ? (slot) : 0], r1_ + ujh0, sizeof(VecD<CPL>)); } \
      else __builtin_memcpy(&rd[slot], ru_ + ujh0, sizeof(VecD<CPL>));                                           \
      __builtin_memcpy(&r58[slot], rg_ + 2u * ujh0, sizeof(VecD<2 * CPL>));                                      \
    } else {                                                                                                     \
      _Pragma("unroll") for (int q = 0; q < CPL; q++) {                                                          \
        if (D0IN) { rd[slot].v[q] = p[qp_ + jc[q]]; ro[D0IN ? (slot) : 0].v[q] = p1[q_ + jc[q]]; }               \
        else rd[slot].v[q] = u1[q_ + jc[q]];                                                                     \
        r58[slot].v[2 * q] = g58[2 * (q_ + jc[q])]; r58[slot].v[2 * q + 1] = g58[2 * (q_ + jc[q]) + 1];          \
      }                                                                                                          \
    }                                                                                                            \
  }
  if (NW > 1) {  // plane 0 is halo: nothing crosses into plane 1
    if (threadIdx.x < 4 * NW) (&edges[0][0][0])[threadIdx.x] = 0.0;
    __syncthreads();
  }
#pragma unroll
  for (int d = 0; d < D; d++) { LOADP(1 + d, d) asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }  // slot by slot, as the loop issues them: the waits at the loop head are the minimum of both orders
  for (int i0 = 1; i0 <= nx; i0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      const int i = i0 + d;
      const bool jodd = (((d + 1 + RBP) & 1) == 0);  // = rb_jodd(i, rb): i0 is odd
      VecD<CPL> un;
      // odd j (position HO + jh):  j+1 <-> previous plane's jh,     j-1 <-> its jh - 1
      // even j (EO + 1 + jh):      j+1 <-> previous plane's jh + 1, j-1 <-> its jh
      double edge = jodd ? wave_shr1(up[CPL - 1]) : wave_shl1(up[0]);
      if (NW > 1) {  // the value that crosses from the neighbouring wave (written at the end of the previous step, before its barrier)
        const int par = (i - 1) & 1;
        if (jodd) { const double e = wv > 0 ? edges[par][wv > 0 ? wv - 1 : 0][1] : 0.0; if (lane == 0) edge = e; }
        else { const double e = wv < NW - 1 ? edges[par][wv < NW - 1 ? wv + 1 : 0][0] : 0.0; if (lane == 63) edge = e; }
      }
#pragma unroll
      for (int q = 0; q < CPL; q++) {
        double ua, ub;  // u(j+1,i-1), u(j-1,i-1)
        if (jodd) { ua = up[q]; ub = q > 0 ? up[q > 0 ? q - 1 : 0] : edge; }
        else { ub = up[q]; ua = q < CPL - 1 ? up[q < CPL - 1 ? q + 1 : 0] : edge; }
        const double d0 = D0IN ? rd[d].v[q] - ro[D0IN ? d : 0].v[q] : rd[d].v[q];
        double t = __builtin_fma(-r58[d].v[2 * q], ua, d0);
        t = __builtin_fma(-r58[d].v[2 * q + 1], ub, t);
        un.v[q] = (FULL || ok[q]) ? t : 0.0;
      }
      if (NW > 1) {
        if (lane == 0) edges[i & 1][wv][0] = un.v[0];
        if (lane == 63) edges[i & 1][wv][1] = un.v[CPL - 1];
      }
      const long long qo = (long long)i * RS + (jodd ? L.HO : L.EO + 1);
      if (FULL) { double *wu_ = u1 + qo; __builtin_memcpy(wu_ + ujh0, &un, sizeof(VecD<CPL>)); }
      else {
#pragma unroll
        for (int q = 0; q < CPL; q++) if (ok[q]) u1[qo + jh0 + q] = un.v[q];
      }
#pragma unroll
      for (int q = 0; q < CPL; q++) up[q] = un.v[q];
      LOADP(i + D, d)
      if (NW > 1) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef LOADP
}

// Measured and dropped (round 4): the walk with LOADER waves -- eight waves keep four chunks of requests in flight each, form d0 on the way
// and hand a plane's three operands to ONE walking wave through a double-buffered LDS ring, a workgroup barrier per chunk of eight planes; the
// walking wave issues no global load at all.  It lost on every level (level 1: 124 us against 73 + 5, level 2: 28 against 21, levels 3 / 4:
// 11.8 / 8.5 against 10.8 / 7.5): a single wave gets a fraction of the LDS request rate (~50 cycles per request, as in k_relax_wave's walk), and
// six 16-byte LDS reads per plane cost more than the six global loads they replace (profiles/r04_rbseq_ring_negative.txt).

// (c): p(:,j,i) += g(:,j,i) * s(j,i) for the columns of colour rb, with the physical mirrors (the pass wrote y into them), and -- SNAPW --
// the new bottom value into the snapshot p1 (and its mirrors), so that a closed level needs no snapshot launch before the next pass.
// One wave = 64 columns of a plane x the rows [kz*KR, (kz+1)*KR).
template <int KU, bool SNAPW>
__global__ __launch_bounds__(256) void k_rbseq_apply(LevView L, int rb, Sides ph, int KR, int nt) {
  const int jh = blockIdx.x * WAVE + threadIdx.x, i = 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (jh >= (L.ny >> 1) || i > L.nx) return;
  const int jodd = rb_jodd(i, rb);
  int c, jm, jp;
  if (jodd) { c = L.HO + jh; jm = L.EO + jh; jp = jm + 1; }
  else      { c = L.EO + jh + 1; jm = L.HO + jh; jp = jm + 1; }
  const int j = jodd ? 2 * jh + 1 : 2 * jh + 2;
  const long long o = (long long)i * L.plane;
  const long long qm = (long long)(i - 1) * L.RS;
  const double s = 0.0 - L.cA[4][o + c] * L.u1[qm + jp] - L.cA[7][o + c] * L.u1[qm + jm];
  double *__restrict__ p = L.p;
  const double *__restrict__ g = L.gk;
  const int k0 = blockIdx.z * KR;
  for (int kb = k0; kb < k0 + KR; kb += KU) {
    double pv[KU], gv[KU];
#pragma unroll
    for (int t = 0; t < KU; t++) {
      const long long ko = o + (long long)(kb + t) * L.RS + c;
      pv[t] = p[ko]; gv[t] = ld_rt(g + ko, nt);
    }
#pragma unroll
    for (int t = 0; t < KU; t++) {
      const long long ro = (long long)(kb + t) * L.RS;
      const double v = pv[t] + gv[t] * s;
      p[o + ro + c] = v;
      mirror_store(L, p, ro, j, i, c, v, ph);
      if (SNAPW && kb + t == 0) {
        LevView L2 = L; L2.plane = L.RS;  // the snapshot: one row per plane
        L.p1[(long long)i * L.RS + c] = v;
        mirror_store(L2, L.p1, 0, j, i, c, v, ph);
      }
    }
  }
}

extern "C" {

void mgxk_rbseq_setup(hipStream_t st, const LevView *L) {
  hipLaunchKernelGGL(k_rbseq_setup, dim3((L->ny + 63) / 64, (L->nx + 3) / 4), dim3(64, 4), 0, st, *L);
}

// (b): d0 and the walk; returns 0 when the level has no instance (more than 1024 columns per half-row): the caller then runs the planes one by one
int mgxk_rbseq_scan(hipStream_t st, const LevView *L, int rb) {
  const int nyh = L->ny / 2, nx = L->nx;
  if (L->gk == nullptr || nyh > 16 * WAVE || (nx & 1)) return 0;
  static const bool two_waves = getenv("MGX_RBSEQ_TWO_WAVES") != nullptr, d0_out = getenv("MGX_RBSEQ_D0_KERNEL") != nullptr;
  const int rbp = rb & 1;
  // helper workgroups that pull the walk's operands into its L2 (k_rbseq_scan): one per ~32 KB of operands, at most 32 (one per compute unit of an XCD)
  static const int help_env = getenv("MGX_RBSEQ_HELPERS") ? atoi(getenv("MGX_RBSEQ_HELPERS")) : -1;
  // Measured (512x512x64, rocprofv3): level 1 (4.2 MB of operands, HBM-resident) one wave 88.4 us without helpers, 59.7 with 32; two waves
  // (a barrier per plane) 73; levels 2-4 (1 MB and less) 20.9 / 11.0 / 7.6 us with or without them -- there the walk is bound by the ~190
  // cycles a plane step costs to issue (4-5 memory instructions, the dependent DPP + FMA chain), and a few dozen extra workgroups only add
  // launch time (level 4: 9.0 against 7.6).  So: helpers from 2 MB of operands on.
  const long long opbytes = (long long)nx * nyh * 32;
  int nhelp = opbytes >= (2LL << 20) ? (int)((opbytes + 131071) / 131072) : 0;
  if (nhelp > 32) nhelp = 32;
  if (nhelp > nx) nhelp = nx;
  if (help_env >= 0) nhelp = help_env < nx ? help_env : nx;
#define SCAN_CASE(CPLV, DV, FULLV, NWV, D0V)                                                                         \
  { if (rbp) hipLaunchKernelGGL((k_rbseq_scan<CPLV, DV, 1, FULLV, NWV, D0V>), dim3(1 + 8 * nhelp), dim3(WAVE * NWV), 0, st, *L, nhelp, rb); \
    else hipLaunchKernelGGL((k_rbseq_scan<CPLV, DV, 0, FULLV, NWV, D0V>), dim3(1 + 8 * nhelp), dim3(WAVE * NWV), 0, st, *L, nhelp, rb);     \
    return 1; }
  // ring depth: D * (loads + stores per plane) < 63 (vmcnt), a divisor of nx
#define SCAN_CPL(CPLV, DMAX, NWV, D0V)                                                                               \
  { const bool full = nyh == CPLV * WAVE * NWV;                                                                      \
    if (nx % DMAX == 0) { if (full) SCAN_CASE(CPLV, DMAX, true, NWV, D0V) else SCAN_CASE(CPLV, DMAX, false, NWV, D0V) } \
    if (nx % 4 == 0) { if (full) SCAN_CASE(CPLV, 4, true, NWV, D0V) else SCAN_CASE(CPLV, 4, false, NWV, D0V) }       \
    if (full) SCAN_CASE(CPLV, 2, true, NWV, D0V) else SCAN_CASE(CPLV, 2, false, NWV, D0V) }
  // small half-rows: one wave forms d0 itself (the level lives in L2; the walk is bound by its dependent chain, not by its requests)
  if (nyh <= 2 * WAVE && !d0_out) {
    if (nyh <= WAVE) SCAN_CPL(1, 16, 1, true)   // y, snapshot, the multiplier pair, the store of u: 4 operations per plane, 16 planes deep
    SCAN_CPL(2, 8, 1, true)
  }
  hipLaunchKernelGGL(k_rbseq_d0, dim3((nyh + WAVE - 1) / WAVE, (nx + 3) / 4), dim3(WAVE, 4), 0, st, *L, rb);
  if (nyh <= WAVE) SCAN_CPL(1, 16, 1, false)
  if (nyh <= 2 * WAVE) SCAN_CPL(2, 16, 1, false)
  // wide half-rows: the requests of ONE wave (at most 63 in flight) do not cover the latency of a level that lives in HBM: several waves
  // (256 columns per half-row: ONE wave with the helpers beats two waves with a barrier per plane, 59.7 against 73 us; wider half-rows
  // -- 512 and 1024 columns, BASELINE config 5 -- would need 16 / 32 memory instructions per plane in one wave: several waves there, unmeasured)
  if (nyh == 4 * WAVE && two_waves) SCAN_CPL(2, 16, 2, false)
  if (nyh == 8 * WAVE) SCAN_CPL(2, 16, 4, false)
  if (nyh == 16 * WAVE) SCAN_CPL(2, 16, 8, false)
  if (nyh <= 4 * WAVE) SCAN_CPL(4, 8, 1, false)
  if (nyh <= 8 * WAVE) SCAN_CPL(8, 4, 1, false)
  SCAN_CPL(16, 2, 1, false)
#undef SCAN_CPL
#undef SCAN_CASE
}

void mgxk_rbseq_apply(hipStream_t st, const LevView *L, int rb, Sides ph, int snapw) {
  const int nyh = L->ny / 2, nz = L->nz;
  const int ku = nz % 8 == 0 ? 8 : (nz % 4 == 0 ? 4 : 2);
  // rows per wave: enough waves to fill the chip on the large levels, whole columns on the small ones
  int kr = nz;
  const long long waves = (long long)((nyh + WAVE - 1) / WAVE) * L->nx;
  while (kr > ku && kr % 2 == 0 && (kr / 2) % ku == 0 && waves * (nz / kr) < 4096) kr /= 2;
  const dim3 grd((nyh + WAVE - 1) / WAVE, (L->nx + 3) / 4, nz / kr), blk(WAVE, 4);
  const int nt = level_streams(L);
#define APPLY_CASE(KUV)                                                                                              \
  { if (snapw) hipLaunchKernelGGL((k_rbseq_apply<KUV, true>), grd, blk, 0, st, *L, rb, ph, kr, nt);                 \
    else hipLaunchKernelGGL((k_rbseq_apply<KUV, false>), grd, blk, 0, st, *L, rb, ph, kr, nt); }
  if (ku == 8) APPLY_CASE(8) else if (ku == 4) APPLY_CASE(4) else APPLY_CASE(2)
#undef APPLY_CASE
}

}  // extern "C"
