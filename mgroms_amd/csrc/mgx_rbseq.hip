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
//
// FUSE (1: with the correction, 2: ... which also keeps the snapshot current): the correction (c) runs INSIDE this launch, chasing the walk.  The
// walk is one wave on one compute unit for ~55 us on level 1 while the other 255 idle, and the correction of plane i needs nothing but
// u of plane i-1: the workgroups whose index is not a multiple of 8 (dealt to the seven XCDs the walk does not run on: its operands stay
// in ITS L2) hold four correction workers each, one per wave, in plane order (rbseq_worker).
// Hand-off (MI355X_MICROARCH.md "Valid forms"; the XCDs' L2s are not coherent with each other).  The walking wave cannot store u
// write-through itself: vmcnt retires loads and stores in issue order, so every operand request would wait for the ~2.5 us a write-through
// acknowledge takes (measured: 117 -> 320 ns per plane).  It stores u as before (acknowledged by its L2) and reports in a word of its own
// how far it is -- plane i - D once step i has computed: the operands of step i were requested after u(i-D) was stored, and they are
// back.  Eight FORWARDING waves -- two workgroups of the walk's XCD (index a multiple of 8: the same L2), not of its compute unit (there
// their write-through traffic slowed the walk from 54 to 82 us) -- take the chunks of RBF_CH planes in turn: wait for the chunk (sc1
// poll of the progress word: served by the L2 the walk writes to), read its u past the L1 (sc1 loads), store it back write-through
// (sc1), wait for every acknowledge (vmcnt(0)), then set the chunk's word (sc1 store of this launch's number).  The workers poll that
// word with sc1 loads and read u with sc1 loads only.  All words count on from launch to launch (compared as signed differences), so
// nothing has to reset them.
// Every wait is bounded.  The walk waits for nobody and its workgroup is dispatched first; the forwarding waves wait for the walk,
// the workers for the forwarding waves.  What the forwarding waves rely on beyond that -- being on the walk's XCD -- is checked
// before the first use (rbseq_placement_ok: workgroups 0, 8, 16 ... of a launch share an XCD) and again in the kernel (the walk leaves
// its XCC_ID next to the progress word); a wait that lasts 2 s or a wrong XCD sets *err (reported by the next synchronising call,
// which turns the fused launch off), releases every word and lets the launch drain.
struct RbFuse { Sides ph; int KR, nt, nchunk, nkz, nworkers, test_stall; long long min_cells; unsigned int *flag; unsigned int seq; int *err; };
#ifdef MGX_RBSEQ_TRACE   // time stamps of the last fused launch (scripts/probe_rbseq_sweep.py), 100 MHz ticks, behind the words
#define RBT(slot, op) { unsigned long long *T_ = (unsigned long long *)(F.flag + (L.nx / RBF_CH + 2) * RBF_FS); const unsigned long long t_ = wall_clock64(); op(T_ + (slot), t_); }
__device__ __forceinline__ void rbt_set(unsigned long long *a, unsigned long long t) { *a = t; }
__device__ __forceinline__ void rbt_max(unsigned long long *a, unsigned long long t) { atomicMax(a, t); }
#else
#define RBT(slot, op) {}
#endif
// forwarding waves, workers per workgroup, planes per chunk, words between two chunk words (64 B), batches of eight rows a worker holds at most,
// chunks between the walk and the workers that request their rows
constexpr int RBF_NF = 8, RBF_WPB = 4, RBF_CH = 8, RBF_FS = 16, RBF_NB = 4, RBF_LEAD = 10;
__device__ long long g_rbs_timeout_ticks = 200000000LL;  // 2 s of the 100 MHz clock (mgxk_set_rbseq_timeout shortens it for the test)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned int rbs_xcc_id() { return (unsigned int)__builtin_amdgcn_s_getreg(20 | (3 << 11)); }  // hwreg(HW_REG_XCC_ID, 0, 4)
// wait until *word has reached `need` (signed difference); false = timed out
template <int NAPS>
__device__ __forceinline__ bool rbs_wait(const unsigned int *word, unsigned int need) {
  if ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) >= 0) return true;
  const long long t0 = wall_clock64(), tmax = g_rbs_timeout_ticks;
  for (;;) {
    __builtin_amdgcn_s_sleep(NAPS);
    if ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) >= 0) return true;
    if (wall_clock64() - t0 > tmax) return false;
  }
}
template <bool SNAPW>
__device__ __forceinline__ void rbseq_worker(const LevView &L, int rb, const RbFuse &F, int jh, int i, int kz, bool live, int c_lo, int c_hi);

template <int CPL, int D, int RBP, bool FULL, int NW, bool D0IN, int FUSE>
__global__ __launch_bounds__(FUSE ? 64 * RBF_WPB : 64 * NW) void k_rbseq_scan(LevView L, int nhelp, int rb, RbFuse F) {
  static_assert(!FUSE || (NW == 1 && FULL && D % RBF_CH == 0), "fused instances: one walking wave, full half-rows, whole chunks per ring");
  if (FUSE && (blockIdx.x & 7) != 0) {
    // four workers per workgroup, consecutive in plane order: ONE wave polls, for the last of them (1792 resident waves polling every
    // ~0.1 us would saturate the memory channel the words live on)
    const int wid0 = ((int)(blockIdx.x >> 3) * 7 + (int)(blockIdx.x & 7) - 1) * RBF_WPB, wid = wid0 + (int)(threadIdx.x >> 6);
    if (wid0 >= F.nworkers) return;
    const int per = F.nkz * F.nchunk;
    const int widl = wid0 + RBF_WPB - 1 < F.nworkers ? wid0 + RBF_WPB - 1 : F.nworkers - 1, il = 1 + widl / per;
    const bool live = wid < F.nworkers;
    const int w = live ? wid : F.nworkers - 1;
    const int kz = w % F.nkz, ch = (w / F.nkz) % F.nchunk, i = 1 + w / per;
    // Pacing: the workgroups resident from the start (two per compute unit) would all request their rows at once -- tens of MB in front of
    // the walk's first requests and its helpers' (measured: the helpers' first lines after 5.2 us, the walk's plane 33 after 10).  A worker
    // asks for its rows when the word of the chunk RBF_LEAD before its own is set (the words trail the walk by ~5 us = 6 chunks: the walk
    // is then ~4 chunks away, the rows take ~2.5 us to arrive), the first ones after ~1.5 us + half the time the walk needs to get there.
    {
      const int cl = il > 1 ? (il - 2) / RBF_CH - RBF_LEAD : -1;
      if (threadIdx.x < 64) {
        if (cl >= 0) { if (!rbs_wait<32>(F.flag + cl * RBF_FS, F.seq) && threadIdx.x == 0) *F.err = 2; }
        else { __builtin_amdgcn_s_sleep(60); for (int k = 0; k < (il >> 2); k++) __builtin_amdgcn_s_sleep(8); }   // ~1.5 us + 0.05 us per plane
      }
      __syncthreads();
    }
    if (wid == F.nworkers - 1 && (threadIdx.x & 63) == 0) RBT(3, rbt_set)
    // plane 0 is halo: u = 0 there, always
    // the chunks that hold plane i-1 of the workgroup's first and last worker (the words are set by different forwarding waves, in no
    // particular order: a workgroup whose workers lie on either side of a chunk boundary waits for both)
    const int i1 = 1 + wid0 / per;
    rbseq_worker<FUSE == 2>(L, rb, F, ch * WAVE + (int)(threadIdx.x & 63), i, kz, live, i1 > 1 ? (i1 - 2) / RBF_CH : 0, il > 1 ? (il - 2) / RBF_CH : -1);
    return;
  }
  if (FUSE && blockIdx.x == 0) {
    if (threadIdx.x >= 64) return;
    if (threadIdx.x == 0) F.flag[(L.nx / RBF_CH) * RBF_FS + 1] = (F.seq << 4) | rbs_xcc_id();
    if (threadIdx.x == 0) RBT(0, rbt_set)
  }
  if (FUSE && (blockIdx.x == 8u * (nhelp + 1) || blockIdx.x == 8u * (nhelp + 2))) {
    // the forwarding waves (two workgroups: a chunk takes a wave ~4 us -- poll, read, write through, acknowledge -- and the walk finishes one
    // every 0.86 us; with four waves the words fell 5-9 us behind by the end of the walk)
    const int f = (int)(threadIdx.x >> 6) + (blockIdx.x == 8u * (nhelp + 1) ? 0 : RBF_WPB), lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(L.u1, 0, 0x7fffffff, 0x00020000);
    constexpr int LPP = CPL >= 2 ? CPL / 2 : 1;   // 16-byte requests per lane and plane (half-row = CPL * 64 columns)
    const unsigned int *pw = F.flag + (L.nx / RBF_CH) * RBF_FS;
    bool bail = false;
    for (int c = f; c * RBF_CH < L.nx; c += RBF_NF) {
      const int ia = c * RBF_CH + 1;
      if (!bail) {
        bail = !rbs_wait<8>(pw, (F.seq << 13) + (unsigned int)(ia + RBF_CH - 1));
        if (!bail && c == f) bail = __hip_atomic_load(pw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ((F.seq << 4) | rbs_xcc_id());
        if (bail && lane == 0) *F.err = 2;
      }
      if (!bail) {
        u32x4 v[RBF_CH * LPP];
        unsigned int off[RBF_CH * LPP];
#pragma unroll
        for (int d = 0; d < RBF_CH; d++) {
          const int i = ia + d;
#pragma unroll
          for (int t = 0; t < LPP; t++) {
            off[d * LPP + t] = (unsigned int)(((long long)i * L.RS + (rb_jodd(i, rb) ? L.HO : L.EO + 1) + (t * 64 + lane) * 2) * 8);
            if (CPL >= 2 || lane < 32) v[d * LPP + t] = __builtin_amdgcn_raw_buffer_load_b128(urs, (int)off[d * LPP + t], 0, 16);  // aux 16 = sc1
          }
        }
#pragma unroll
        for (int q = 0; q < RBF_CH * LPP; q++) if (CPL >= 2 || lane < 32) __builtin_amdgcn_raw_buffer_store_b128(v[q], urs, (int)off[q], 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (lane == 0) __hip_atomic_store(F.flag + c * RBF_FS, F.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) RBT(2, rbt_max)
    }
    return;
  }
  if (blockIdx.x != 0) {
    // Helper workgroups (nhelp of them, those with blockIdx % 8 == 0: dealt to the walking workgroup's XCD).  Every colour pass is a kernel
    // boundary, after which the walk's operands come from the Infinity Cache / HBM (~1.4 us per request: 16 planes of look-ahead make
    // 86 ns per plane, measured); the helpers ask for the same lines, many at a time, so that the walking wave finds them in the L2 it
    // shares with them.  Speed only: nothing depends on where a workgroup lands or on whether a line is still there.
    if ((blockIdx.x & 7) != 0) return;
    const int h = (blockIdx.x >> 3) - 1;
    if (h >= nhelp) return;   // (a fused launch has more workgroups on this XCD than helpers)
    // Helper h takes the planes [h*per+1, (h+1)*per] and touches every 64 bytes of their operand rows ONCE, all requests in flight
    // before the first is waited for.  (The workgroups of a launch start ~0.15 us apart; a helper that walked its planes one after the
    // other, a memory latency each, left the walk right behind the helpers' start-up for its first planes: plane 33 after 7.8 us.)
    const int nyh_ = L.ny >> 1, per = (L.nx + nhelp - 1) / nhelp, ia = 1 + h * per, ib = ia + per - 1 < L.nx ? ia + per - 1 : L.nx;
    const int seg = (nyh_ * 8 + 63) / 64, ppp = seg * (D0IN ? 4 : 3), total = (ib - ia + 1) * ppp;   // 64-byte pieces: an 8-byte-per-column row has seg of them
    float acc = 0.f;
    for (int t0 = threadIdx.x; t0 < total; t0 += 8 * (int)blockDim.x) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int t = t0 + e * (int)blockDim.x;
        v[e] = 0.f;
        if (t < total) {
          const int i = ia + t / ppp, r = t % ppp, off = rb_jodd(i, rb) ? L.HO : L.EO + 1;
          const long long q = (long long)i * L.RS + off;
          const char *base;
          int piece = r;
          if (D0IN) {
            if (r < seg) base = (const char *)(L.p + (long long)i * L.plane + off);
            else if (r < 2 * seg) { base = (const char *)(L.p1 + q); piece = r - seg; }
            else { base = (const char *)(L.ag58 + 2 * q); piece = r - 2 * seg; }
          } else {
            if (r < seg) base = (const char *)(L.u1 + q);
            else { base = (const char *)(L.ag58 + 2 * q); piece = r - seg; }
          }
          // (the last piece of a row is clamped to the row's last 4 bytes)
          const int rowbytes = (r < (D0IN ? 2 : 1) * seg ? 8 : 16) * nyh_;
          int byte = piece * 64;
          if (byte > rowbytes - 4) byte = rowbytes - 4;
          v[e] = *(const float *)(base + byte);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; e++) acc += v[e];
    }
    if (acc == 1.2345678e-30f) L.u1[0] = acc;   // never: keeps the requests alive (row 0 of u1 is halo and stays zero)
    return;
  }
  const int lane = threadIdx.x & 63, wv = FUSE ? 0 : threadIdx.x >> 6, nyh = L.ny >> 1, nx = L.nx;
  const int jh0 = (wv * 64 + lane) * CPL;
  const unsigned int ujh0 = (unsigned int)jh0;
  const long long RS = L.RS;
  double *__restrict__ u1 = L.u1;
  const double *__restrict__ g58 = L.ag58, *__restrict__ p = L.p, *__restrict__ p1 = L.p1;
  __shared__ double edges[2][NW > 1 ? NW : 1][2];
  constexpr int PUB = D < 4 ? D : 4;  // planes between two publications of the progress word
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
#ifdef MGX_RBSEQ_TRACE

#endif
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
      if (FUSE && ((d + 1) % PUB) == 0 && i > D) {
        // this step has consumed operands requested after u(i - D) was stored: planes <= i - D are in the L2
        if (lane == 0 && !F.test_stall) F.flag[(nx / RBF_CH) * RBF_FS] = (F.seq << 13) + (unsigned int)(i - D);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#undef LOADP
  if (FUSE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && !F.test_stall) F.flag[(nx / RBF_CH) * RBF_FS] = (F.seq << 13) + (unsigned int)nx;
  }
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
__device__ __forceinline__ void rbseq_apply_cols(const LevView &L, int rb, const Sides &ph, int KR, int nt, int jh, int i, int kz) {
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
  const int k0 = kz * KR;
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

template <int KU, bool SNAPW>
__global__ __launch_bounds__(256) void k_rbseq_apply(LevView L, int rb, Sides ph, int KR, int nt) {
  rbseq_apply_cols<KU, SNAPW>(L, rb, ph, KR, nt, blockIdx.x * WAVE + threadIdx.x, 1 + blockIdx.y * blockDim.y + threadIdx.y, blockIdx.z);
}

// (c) as a worker of the fused launch (k_rbseq_scan, FUSE): the same arithmetic as rbseq_apply_cols.  What does not depend on the walk -- y
// and g of the worker's rows (at most 32: four batches of eight), the two couplings -- is requested BEFORE the wait for the walk, so a
// worker that has waited finishes one memory latency after its word is set (the tail of the launch behind the walk's last plane).
// c_lo .. c_hi (wave 0 of the workgroup polls their words, the others wait at the barrier): the chunks that hold plane i-1 of the workgroup's workers;
// every load of u is an sc1 load.
template <bool SNAPW>
__device__ __forceinline__ void rbseq_worker(const LevView &L, int rb, const RbFuse &F, int jh, int i, int kz, bool live, int c_lo, int c_hi) {
  const int jodd = rb_jodd(i, rb);
  int c, jm, jp;
  if (jodd) { c = L.HO + jh; jm = L.EO + jh; jp = jm + 1; }
  else      { c = L.EO + jh + 1; jm = L.HO + jh; jp = jm + 1; }
  const int j = jodd ? 2 * jh + 1 : 2 * jh + 2;
  const long long o = (long long)i * L.plane;
  const long long qm = (long long)(i - 1) * L.RS;
  double *__restrict__ p = L.p;
  const double *__restrict__ g = L.gk;
  const int k0 = kz * F.KR, nb = F.KR / 8, nt = F.nt;
  double pv[RBF_NB][8], gv[RBF_NB][8], c5 = 0.0, c8 = 0.0;
  if (live) {
    c5 = L.cA[4][o + c]; c8 = L.cA[7][o + c];
#pragma unroll
    for (int b = 0; b < RBF_NB; b++)
      if (b < nb) {
#pragma unroll
        for (int t = 0; t < 8; t++) { const long long ko = o + (long long)(k0 + 8 * b + t) * L.RS + c; pv[b][t] = p[ko]; gv[b][t] = ld_rt(g + ko, nt); }
      }
  }
  if (threadIdx.x < 64)
    for (int c_ = c_lo; c_ <= c_hi; c_++)
      if (!rbs_wait<16>(F.flag + c_ * RBF_FS, F.seq) && threadIdx.x == 0) *F.err = 2;
  __syncthreads();
  if (!live) return;
#ifdef MGX_RBSEQ_TRACE
  const bool lastw = i == L.nx && kz == F.nkz - 1 && jh == (L.ny >> 1) - 64;
  if (lastw) RBT(4, rbt_set)
#endif
  const double ujp = __hip_atomic_load(L.u1 + qm + jp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double ujm = __hip_atomic_load(L.u1 + qm + jm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double s = 0.0 - c5 * ujp - c8 * ujm;
#ifdef MGX_RBSEQ_TRACE
  if (lastw) { if (s == 1.234e-300) L.u1[0] = s; RBT(5, rbt_set) }
#endif
#pragma unroll
  for (int b = 0; b < RBF_NB; b++)
    if (b < nb) {
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const int k = k0 + 8 * b + t;
        const long long ro = (long long)k * L.RS;
        const double v = pv[b][t] + gv[b][t] * s;
        p[o + ro + c] = v;
        mirror_store(L, p, ro, j, i, c, v, F.ph);
        if (SNAPW && k == 0) {
          LevView L2 = L; L2.plane = L.RS;  // the snapshot: one row per plane
          L.p1[(long long)i * L.RS + c] = v;
          mirror_store(L2, L.p1, 0, j, i, c, v, F.ph);
        }
      }
    }
#ifdef MGX_RBSEQ_TRACE
  if (lastw) RBT(1, rbt_set)
#endif
}

// (b) + (c) of a SMALL level in one launch without any hand-off: every workgroup walks the planes 1 .. ib-1 itself (wave 0, u into LDS),
// then its four waves correct the planes ia .. ib it owns (rows split over the waves).  The walk of such a level is a few microseconds of
// dependent steps on operands that live in the L2 / Infinity Cache; redoing it in ~64 workgroups costs no time -- the launch lasts as
// long as the longest walk plus one correction -- and saves the correction's launch and the kernel boundary in front of it (levels 3 and
// 4 of the 512x512x64 problem: 10.5 + 5 and 7.6 + 5 us in two launches).  d0 must come from a buffer no workgroup of this launch writes
// (the colour pass left it in u1: LevView::d0w): another workgroup may already be correcting p and the snapshot of a plane this one
// still walks over.  Half-rows of at most 64 columns (one column per lane), at most 128 planes (u: 74 KB of LDS).
// Measured and dropped: the same with two columns per lane and u of the workgroup's own planes only (any number of planes), to serve
// level 2 (256 planes of 128 columns) as well: Vcycle 3.52 against 3.19 ms with level 2 included (128 workgroups each redoing a 20 us
// walk), and 3.24 against 3.19 on levels 3-4 alone (the generic walk is slower than this one).
template <int D, int RBP, bool SNAPW>
__global__ __launch_bounds__(256) void k_rbseq_walk_apply(LevView L, int rb, Sides ph, int PB, int nt) {
  extern __shared__ double ul[];   // u(jh, plane) at plane * 64 + jh; plane 0 (halo) and the lanes past the half-row: zero
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nyh = L.ny >> 1, nx = L.nx;
  const int ia = blockIdx.x * PB + 1, ib = ia + PB - 1 < nx ? ia + PB - 1 : nx;
  const int nwalk = ib - 1;
  if (threadIdx.x < 64) ul[threadIdx.x] = 0.0;
  if (wv == 0 && nwalk >= 1) {
    const bool ok = lane < nyh;
    const int jc = ok ? lane : nyh - 1;
    const long long RS = L.RS;
    const double *__restrict__ d0 = L.u1, *__restrict__ g58 = L.ag58;
    double rd[D], ra[D], rbb[D], up = 0.0;
#define WLOADP(ip, slot)                                                                                         \
    {                                                                                                            \
      const int i_ = (ip) <= nx ? (ip) : nx;                                                                     \
      const long long q_ = (long long)i_ * RS + (((((slot) + 1 + RBP) & 1) == 0) ? L.HO : L.EO + 1) + jc;        \
      rd[slot] = d0[q_]; ra[slot] = g58[2 * q_]; rbb[slot] = g58[2 * q_ + 1];                                    \
    }
#pragma unroll
    for (int d = 0; d < D; d++) { WLOADP(1 + d, d) asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    for (int i0 = 1; i0 <= nwalk; i0 += D) {
#pragma unroll
      for (int d = 0; d < D; d++) {
        const int i = i0 + d;
        const bool jodd = (((d + 1 + RBP) & 1) == 0);  // = rb_jodd(i, rb): i0 is odd, D even
        const double edge = jodd ? wave_shr1(up) : wave_shl1(up);
        const double ua = jodd ? up : edge, ub = jodd ? edge : up;   // u(j+1,i-1), u(j-1,i-1)
        double t = __builtin_fma(-ra[d], ua, rd[d]);
        t = __builtin_fma(-rbb[d], ub, t);
        up = ok ? t : 0.0;
        ul[i * 64 + lane] = up;
        WLOADP(i + D, d)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef WLOADP
  }
  __syncthreads();
  if (lane >= nyh) return;
  const int R = L.nz >> 2, k0 = wv * R;   // rows of this wave
  double *__restrict__ p = L.p;
  const double *__restrict__ g = L.gk;
  for (int i = ia; i <= ib; i++) {
    const int jh = lane, jodd = rb_jodd(i, rb);
    const int c = jodd ? L.HO + jh : L.EO + jh + 1;
    const int j = jodd ? 2 * jh + 1 : 2 * jh + 2;
    // odd j: u(j+1,i-1) is the previous plane's jh, u(j-1,i-1) its jh - 1; even j: jh + 1 and jh (zero beyond the half-row: halo)
    const double *um = ul + (i - 1) * 64;
    const double ujp = jodd ? um[jh] : (jh + 1 < nyh ? um[jh + 1] : 0.0);
    const double ujm = jodd ? (jh > 0 ? um[jh - 1] : 0.0) : um[jh];
    const long long o = (long long)i * L.plane;
    const double s = 0.0 - L.cA[4][o + c] * ujp - L.cA[7][o + c] * ujm;
    for (int t = 0; t < R; t++) {
      const int k = k0 + t;
      const long long ro = (long long)k * L.RS;
      const double v = p[o + ro + c] + ld_rt(g + o + ro + c, nt) * s;
      p[o + ro + c] = v;
      mirror_store(L, p, ro, j, i, c, v, ph);
      if (SNAPW && k == 0) {
        LevView L2 = L; L2.plane = L.RS;  // the snapshot: one row per plane
        L.p1[(long long)i * L.RS + c] = v;
        mirror_store(L2, L.p1, 0, j, i, c, v, ph);
      }
    }
  }
}

// (b) + (c) in one launch with NO hand-off and no sequential walk over the level: the WINDOWED walk (option "rbseq_window").
// The recurrence of (b), u(.,i) = d0(.,i) + M_i u(.,i-1), contracts: every row of M_i holds the two multipliers ag5, ag8 of one column, so
// ||M_i||_inf <= rho = max over the level of |ag5| + |ag8| (k_rbseq_rho, found at set-up: a property of the matrix; 0.03-0.04 on every level
// of the seamount problem).  A walk started from zero m planes before plane i therefore gives u(.,i) to within rho^m * max|u|, and m is
// chosen at set-up so that rho^m <= 2^-64: below half an ulp of max|u|, the same order as the reassociation (a) + (c) make anyway.  The
// dependence cone of a 64-column chunk widens by one column every two planes, so a workgroup walks a window of its chunk +- 32 columns
// (one wave, two columns per lane; the whole half-row where that is at most 64 columns) over the m planes in front of its own, with u in
// registers and the last plane's in LDS -- operands that the colour pass and the set-up left in the L2 / Infinity Cache, ~25 KB per
// workgroup -- while its four other waves have the rows of y and g of the workgroup's plane in flight; then p = y + g s as in
// k_rbseq_apply.  Nothing is handed from one workgroup to another: no progress words, no forwarding waves, no placement assumption.
// Where rho is too close to one (m > RBW_MAXM) the walk over the whole level stays (k_rbseq_scan).
// Workgroup = chunk ch of plane i, rows [kz*4*KR, (kz+1)*4*KR): waves 0-3 hold KR rows each (KR = 16 at nz = 64 ... 1 at nz = 4), wave 4 walks.
constexpr int RBW_MAXM = 48;
// Measured on level 1 of 512x512x64 (sweep of two passes + two of these launches, HIP events): ring depth 8 at 4 waves per SIMD 0.304-0.315 ms,
// depth 4 or 2 at 5-6 waves per SIMD 0.304-0.313: the same.  With no plane walked at all (timing probe MGX_RBW_PROBE_M=0) 0.295-0.296, with 14 planes
// 0.306-0.310: the walk costs ~6 us of the launch's ~45 (201 MB algorithmic, 210 MB counted: profiles/r04_pmc_traffic_rb_window.json) -- the dependent
// chain of 14 steps in ONE wave that shares its SIMD's issue slots with three row waves; at raised priority (s_setprio) ~4 of the 6 come back.
template <int CPL, int KR, bool SNAPW, int RBW_D = 8>
__global__ __launch_bounds__(320) void k_rbseq_window(LevView L, int rb, Sides ph, int m, int nt, int xmap, int nch, int nkz, int prio, int kcut) {
  __shared__ double ul[64 * CPL + 2];   // u of plane i-1 over the window at 1 + (column - w0); a zero on either side
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nyh = L.ny >> 1;
  // block -> (chunk, plane, row group).  XCD-aware where the planes divide by 8 (xmap): workgroup b runs on XCD b % 8, which takes a contiguous
  // eighth of the planes -- the windows of consecutive planes overlap in all but one plane, so the walks' operands are fetched into ONE L2
  // (with the plain order every XCD reads every plane's: +12 % of the launch's traffic on level 1 of 512x512x64)
  int ch, i, kz;
  if (xmap) {
    const int b = blockIdx.x, x = b & 7, r = b >> 3, per = nch * nkz;
    i = 1 + x * (L.nx >> 3) + r / per; ch = r % nch; kz = (r / nch) % nkz;
  } else { ch = blockIdx.x; i = 1 + blockIdx.y; kz = blockIdx.z; }
  if (kz * 4 * KR >= kcut) return;   // (rows the correction does not reach: see kcut below; the whole workgroup)
  const int w0 = CPL == 1 ? 0 : ch * 64 - 32;
  const long long RS = L.RS;
  const int jh = ch * 64 + lane;
  const bool live = jh < nyh;
  const int jodd = rb_jodd(i, rb);
  const int jhc = live ? jh : nyh - 1;
  const int c = jodd ? L.HO + jhc : L.EO + jhc + 1;
  const int j = jodd ? 2 * jhc + 1 : 2 * jhc + 2;
  const long long o = (long long)i * L.plane;
  double *__restrict__ p = L.p;
  const int k0 = (kz * 4 + (wv & 3)) * KR;
  double pv[KR], gv[KR], c5 = 0.0, c8 = 0.0;
  if (wv == 4) {
    if (prio) __builtin_amdgcn_s_setprio(3);   // the workgroup waits for this wave's dependent chain: let it issue ahead of the row waves
    double up[CPL];
    bool ok[CPL];
    int jc[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) {
      const int jw = w0 + lane * CPL + q;
      up[q] = 0.0; ok[q] = jw >= 0 && jw < nyh; jc[q] = jw < 0 ? 0 : (jw < nyh ? jw : nyh - 1);
    }
    const int ia = i - m > 1 ? i - m : 1;   // planes ia .. i-1 (plane 0 is halo: u = 0)
    if (ia < i) {
      const double *__restrict__ d0 = L.u1, *__restrict__ g58 = L.ag58;
      double rd[RBW_D][CPL], ra[RBW_D][CPL], rbb[RBW_D][CPL];
      // every request is unconditional (planes past i-1 clamp to i-1, columns outside the half-row to its ends): the waits stay counted
#define WLOADW(ip, slot)                                                                                         \
      {                                                                                                          \
        const int i_ = (ip) < i ? (ip) : i - 1;                                                                  \
        const long long q_ = (long long)i_ * RS + (rb_jodd(i_, rb) ? L.HO : L.EO + 1);                           \
        _Pragma("unroll") for (int q = 0; q < CPL; q++) {                                                        \
          rd[slot][q] = d0[q_ + jc[q]]; LD_PAIR(g58 + 2 * (q_ + jc[q]), ra[slot][q], rbb[slot][q])               \
        }                                                                                                        \
      }
#pragma unroll
      for (int d = 0; d < RBW_D; d++) { WLOADW(ia + d, d) asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
      for (int i0 = ia; i0 < i; i0 += RBW_D) {
#pragma unroll
        for (int d = 0; d < RBW_D; d++) {
          const int ip = i0 + d;
          if (ip < i) {
            const bool jo = rb_jodd(ip, rb);
            // odd j (position HO + jh):  j+1 <-> previous plane's jh,     j-1 <-> its jh - 1
            // even j (EO + 1 + jh):      j+1 <-> previous plane's jh + 1, j-1 <-> its jh
            const double e_r = wave_shr1(up[CPL - 1]), e_l = wave_shl1(up[0]);
            double un[CPL];
#pragma unroll
            for (int q = 0; q < CPL; q++) {
              const double lo = q > 0 ? up[q > 0 ? q - 1 : 0] : e_r, hi = q < CPL - 1 ? up[q < CPL - 1 ? q + 1 : 0] : e_l;
              const double ua = jo ? up[q] : hi, ub = jo ? lo : up[q];   // u(j+1,i-1), u(j-1,i-1)
              double t = __builtin_fma(-ra[d][q], ua, rd[d][q]);
              t = __builtin_fma(-rbb[d][q], ub, t);
              un[q] = ok[q] ? t : 0.0;
            }
#pragma unroll
            for (int q = 0; q < CPL; q++) up[q] = un[q];
          }
          WLOADW(ip + RBW_D, d)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#undef WLOADW
    }
#pragma unroll
    for (int q = 0; q < CPL; q++) ul[1 + lane * CPL + q] = up[q];
    if (lane == 0) { ul[0] = 0.0; ul[64 * CPL + 1] = 0.0; }
  } else {
    const double *__restrict__ g = L.gk;
    c5 = L.cA[4][o + c]; c8 = L.cA[7][o + c];
#pragma unroll
    for (int t = 0; t < KR; t++) if (k0 + t < kcut) { const long long ko = o + (long long)(k0 + t) * RS + c; pv[t] = p[ko]; gv[t] = ld_rt(g + ko, nt); }
  }
  __syncthreads();
  if (wv == 4 || !live) return;
  const int wi = 1 + jh - w0;   // the column's own place in ul
  const double ujp = jodd ? ul[wi] : ul[wi + 1], ujm = jodd ? ul[wi - 1] : ul[wi];
  const double s = 0.0 - c5 * ujp - c8 * ujm;
#pragma unroll
  for (int t = 0; t < KR; t++) {
    const int k = k0 + t;
    if (k >= kcut) break;   // (wave-uniform)
    const long long ro = (long long)k * RS;
    const double v = pv[t] + gv[t] * s;
    p[o + ro + c] = v;
    mirror_store(L, p, ro, j, i, c, v, ph);
    if (SNAPW && k == 0) {
      LevView L2 = L; L2.plane = L.RS;  // the snapshot: one row per plane
      L.p1[(long long)i * L.RS + c] = v;
      mirror_store(L2, L.p1, 0, j, i, c, v, ph);
    }
  }
}

// kcut: how far up a column the correction reaches.  g = T^-1 e1 decays away from the bottom row (the column matrix is diagonally dominant: by 0.4
// per row on level 1 of the seamount problem), and the correction of row k is g(k) s = (g(k) / g(1)) * (g(1) s), g(1) s being the bottom row's -- at most
// twice the largest increment u.  Above the last row where max over the columns of |g(k) / g(1)| exceeds 2^-64 the correction is below 2^-63 of the
// largest increment -- the same order as the window's truncation -- and those rows are neither read nor written: 55 of 64 rows at 512x512x64,
// 56 of 128 at nz = 128 (BASELINE config 5's columns), every row on the coarser levels.  out[k-1] = that maximum, per row (non-negative doubles
// order like their bit patterns).
__global__ __launch_bounds__(256) void k_rbseq_gdecay(LevView L, unsigned long long *out) {
  __shared__ double red[4];
  const int k = blockIdx.y, ncol = L.nx * L.ny;
  double v = 0.0;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < ncol; t += gridDim.x * 256) {
    const int i = 1 + t / L.ny, jj = 1 + t % L.ny;
    const long long o = (long long)i * L.plane + jpos(L, jj);
    const double g1 = L.gk[o], gkv = L.gk[o + (long long)k * L.RS];
    const double r = __builtin_fabs(gkv) / __builtin_fabs(g1);
    v = !(r <= v) ? r : v;   // (a NaN -- 0 / 0 in a degenerate column -- wins: the cut is then refused)
  }
  for (int sft = 32; sft >= 1; sft >>= 1) { const double ov = __shfl_xor(v, sft); v = !(ov <= v) ? ov : v; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) v = !(red[w] <= v) ? red[w] : v;
    if (v != v) v = __builtin_inf();
    atomicMax(out + k, (unsigned long long)__double_as_longlong(v));
  }
}

// rho = max over the interior columns of |ag5| + |ag8| (a bound of the walk's step in the maximum norm); non-negative doubles order like their bit patterns
__global__ void k_rbseq_rho(LevView L, unsigned long long *out) {
  const int jj = 1 + blockIdx.x * blockDim.x + threadIdx.x, i = 1 + blockIdx.y * blockDim.y + threadIdx.y;
  double v = 0.0;
  if (jj <= L.ny && i <= L.nx) {
    const long long q = (long long)i * L.RS + jpos(L, jj);
    v = __builtin_fabs(L.ag58[2 * q]) + __builtin_fabs(L.ag58[2 * q + 1]);
  }
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  for (int s = 32; s >= 1; s >>= 1) { const unsigned long long ob = __shfl_xor(b, s); b = ob > b ? ob : b; }
  if (threadIdx.x == 0) atomicMax(out, b);
}

template <int CPL, int D, int NW, bool D0IN>
static bool rbseq_fused_launch(hipStream_t st, const LevView *L, int nhelp, int rb, const RbFuse &F, int snapw) {
  if constexpr (NW == 1 && D % RBF_CH == 0) {
    const int per = 7 * RBF_WPB, octets = (F.nworkers + per - 1) / per, grid = 8 * (octets > nhelp + 3 ? octets : nhelp + 3);
    const dim3 blk(WAVE * RBF_WPB);
    if (snapw) { if (rb & 1) hipLaunchKernelGGL((k_rbseq_scan<CPL, D, 1, true, 1, D0IN, 2>), dim3(grid), blk, 0, st, *L, nhelp, rb, F);
                 else hipLaunchKernelGGL((k_rbseq_scan<CPL, D, 0, true, 1, D0IN, 2>), dim3(grid), blk, 0, st, *L, nhelp, rb, F); }
    else { if (rb & 1) hipLaunchKernelGGL((k_rbseq_scan<CPL, D, 1, true, 1, D0IN, 1>), dim3(grid), blk, 0, st, *L, nhelp, rb, F);
           else hipLaunchKernelGGL((k_rbseq_scan<CPL, D, 0, true, 1, D0IN, 1>), dim3(grid), blk, 0, st, *L, nhelp, rb, F); }
    return true;
  } else return false;
}

// Do the workgroups 0, 8, 16 ... of a launch share an XCD (what the forwarding waves of the fused launch rely on)?  Asked once per device.
__global__ void k_rbseq_xcc_probe(unsigned int *o) { if (threadIdx.x == 0) o[blockIdx.x] = rbs_xcc_id(); }
static bool rbseq_placement_ok(hipStream_t st) {
  static int known[64];   // per device: 0 = not asked yet, 1 = yes, 2 = no
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (known[dev]) return known[dev] == 1;
  unsigned int *d = nullptr, h[256];
  bool ok = hipMalloc((void **)&d, sizeof h) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_rbseq_xcc_probe, dim3(256), dim3(64), 0, st, d);
    ok = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    for (int b = 8; ok && b < 256; b += 8) ok = h[b] == h[0];
    (void)hipFree(d);
  }
  known[dev] = ok ? 1 : 2;
  return ok;
}

extern "C" {

void mgxk_rbseq_setup(hipStream_t st, const LevView *L) {
  hipLaunchKernelGGL(k_rbseq_setup, dim3((L->ny + 63) / 64, (L->nx + 3) / 4), dim3(64, 4), 0, st, *L);
}

// rows per correction wave: enough waves to fill the chip on the large levels, whole columns on the small ones
static void rbseq_apply_shape(const LevView *L, int *ku, int *kr) {
  const int nyh = L->ny / 2, nz = L->nz;
  *ku = nz % 8 == 0 ? 8 : (nz % 4 == 0 ? 4 : 2);
  *kr = nz;
  const long long waves = (long long)((nyh + WAVE - 1) / WAVE) * L->nx;
  while (*kr > *ku && *kr % 2 == 0 && (*kr / 2) % *ku == 0 && waves * (nz / *kr) < 4096) *kr /= 2;
}

// (b): d0 and the walk -- and, fz != nullptr, the correction (c) inside the walk's launch where an instance exists (returns 2 then).  Returns 0
// when the level has no instance (more than 1024 columns per half-row): the caller then runs the planes one by one
static int rbseq_scan_launch(hipStream_t st, const LevView *L, int rb, RbFuse *fz, int snapw, int have_d0) {
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
  RbFuse none = {};
  // the correction inside the walk's launch: instances for full half-rows, the deepest ring, nz a multiple of 8
  int ku, kr;
  rbseq_apply_shape(L, &ku, &kr);
  // ... and where it pays: the correction of a colour is 24 B per cell; below ~100 MB it takes less than what the hand-off adds (the
  // forwarding waves lag the walk by 2-5 us, the last workers finish ~4 us later: 256x256x32, 0.111 ms per sweep fused against 0.099)
  const bool can_fuse = fz != nullptr && ku == 8 && (long long)(nx + 2) * L->RS * 8 < (1LL << 31) && (long long)nx * nyh * L->nz >= fz->min_cells;
  // eight rows per worker there: what a worker does after its word is set -- 16 stores with the mirrors, ~1.4 us per eight rows under load --
  // is the tail of the launch behind the walk's last plane (level-1 sweep at 512x512x64 with 32 / 16 / 8 rows: 0.346 / 0.345-0.357 / 0.339 ms)
  if (can_fuse) kr = 8;
  if (can_fuse) { fz->KR = kr; fz->nt = level_streams(L); fz->nchunk = (nyh + WAVE - 1) / WAVE; fz->nkz = L->nz / kr; fz->nworkers = nx * fz->nchunk * fz->nkz; }
#define SCAN_LAUNCH(CPLV, DV, RBPV, FULLV, NWV, D0V, FUSEV, GRID, FARG)                                                \
  hipLaunchKernelGGL((k_rbseq_scan<CPLV, DV, RBPV, FULLV, NWV, D0V, FUSEV>), dim3(GRID), dim3(WAVE * NWV), 0, st, *L, nhelp, rb, FARG)
#define SCAN_CASE(CPLV, DV, FULLV, NWV, D0V)                                                                         \
  { if (rbp) SCAN_LAUNCH(CPLV, DV, 1, FULLV, NWV, D0V, 0, 1 + 8 * nhelp, none); else SCAN_LAUNCH(CPLV, DV, 0, FULLV, NWV, D0V, 0, 1 + 8 * nhelp, none); \
    return 1; }
  // fused: workgroup b holds 1 + RBF_NF workers unless b % 8 == 0 (those: the walk with its forwarding waves, the helpers, nothing)
#define FUSE_CASE(CPLV, DV, NWV, D0V)                                                                                \
  if (can_fuse && NWV == 1 && nyh == CPLV * WAVE && nx % DV == 0 && rbseq_fused_launch<CPLV, DV, NWV, D0V>(st, L, nhelp, rb, *fz, snapw)) return 2;
  // ring depth: D * (loads + stores per plane) < 63 (vmcnt), a divisor of nx
#define SCAN_CPL(CPLV, DMAX, NWV, D0V)                                                                               \
  { const bool full = nyh == CPLV * WAVE * NWV;                                                                      \
    FUSE_CASE(CPLV, DMAX, NWV, D0V)                                                                                  \
    if (nx % DMAX == 0) { if (full) SCAN_CASE(CPLV, DMAX, true, NWV, D0V) else SCAN_CASE(CPLV, DMAX, false, NWV, D0V) } \
    if (nx % 4 == 0) { if (full) SCAN_CASE(CPLV, 4, true, NWV, D0V) else SCAN_CASE(CPLV, 4, false, NWV, D0V) }       \
    if (full) SCAN_CASE(CPLV, 2, true, NWV, D0V) else SCAN_CASE(CPLV, 2, false, NWV, D0V) }
  // small half-rows: one wave forms d0 itself (the level lives in L2; the walk is bound by its dependent chain, not by its requests)
  // (half-rows of 65..128 columns whose pass wrote d0: the walk reads it -- three requests and a store per plane, 16 planes of look-ahead
  // instead of 8 with five: level 2 of 512x512x64 20.5 -> ~16 us per colour, Vcycle 3.12 -> 3.08 ms; MGX_RBSEQ_NO_D0_MID: A/B)
  static const bool d0_mid = getenv("MGX_RBSEQ_NO_D0_MID") == nullptr;
  if (nyh <= 2 * WAVE && !d0_out && !(have_d0 && d0_mid && nyh > WAVE)) {
    if (nyh <= WAVE) SCAN_CPL(1, 16, 1, true)   // y, snapshot, the multiplier pair, the store of u: 4 operations per plane, 16 planes deep
    SCAN_CPL(2, 8, 1, true)
  }
  if (!have_d0) hipLaunchKernelGGL(k_rbseq_d0, dim3((nyh + WAVE - 1) / WAVE, (nx + 3) / 4), dim3(WAVE, 4), 0, st, *L, rb);   // (the colour pass may have written it: LevView::d0w)
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
#undef FUSE_CASE
#undef SCAN_CASE
#undef SCAN_LAUNCH
}

int mgxk_rbseq_scan(hipStream_t st, const LevView *L, int rb, int have_d0) { return rbseq_scan_launch(st, L, rb, nullptr, 0, have_d0); }
// does the walk of this level read d0 from u1 (wide half-rows) rather than form it itself?  Then the colour pass should write it (LevView::d0w)
int mgxk_rbseq_wants_d0(const LevView *L) {
  static const bool d0_out = getenv("MGX_RBSEQ_D0_KERNEL") != nullptr;
  // (wide half-rows: the walk reads d0; small levels: k_rbseq_walk_apply needs it where no workgroup of its launch writes)
  static const bool d0_mid = getenv("MGX_RBSEQ_NO_D0_MID") == nullptr;
  return L->gk != nullptr && (L->ny / 2 > 2 * WAVE || d0_out || (L->ny / 2 <= WAVE && L->nx <= 128) || (d0_mid && L->ny / 2 > WAVE));
}

// (b) + (c) in one launch where an instance exists (returns 2), else (b) alone (returns 1: the caller launches mgxk_rbseq_apply) or nothing (0).
// words: this level's hand-off words (device memory, zero at allocation: nx / 8 chunk words, the walk's progress word, 64 B apart);
// seq: the number of this launch on the level (1, 2, ...: the caller counts); err: host-mapped error word; test_stall: the walk keeps
// its progress to itself (the test of the bounded waits); min_cells: cells of a colour from which on the fused launch is used
int mgxk_rbseq_scan_apply(hipStream_t st, const LevView *L, int rb, Sides ph, int snapw, int have_d0, unsigned int *words, unsigned int seq, int *err, int test_stall, long long min_cells) {
  RbFuse f = {};
  f.ph = ph; f.flag = words; f.seq = seq; f.err = err; f.test_stall = test_stall; f.min_cells = min_cells;
  if (L->nx >= (1 << 13) || err == nullptr || !rbseq_placement_ok(st)) return rbseq_scan_launch(st, L, rb, nullptr, 0, have_d0);
  return rbseq_scan_launch(st, L, rb, &f, snapw, have_d0);
}

// (b) + (c) of a small level in ONE launch (k_rbseq_walk_apply); needs d0 in u1 (the colour pass wrote it).  Returns 1 when launched.
int mgxk_rbseq_walk_apply(hipStream_t st, const LevView *L, int rb, Sides ph, int snapw) {
  const int nyh = L->ny / 2, nx = L->nx, nz = L->nz;
  static const bool off = getenv("MGX_NO_RBSEQ_WALK_APPLY") != nullptr;
  if (off || L->gk == nullptr || nyh > WAVE || nx > 128 || (nx & 1) || nz < 4 || (nz & 3)) return 0;
  mgx_before_launch();
  constexpr int DW = 16;
  const size_t lds = (size_t)(nx + DW + 1) * 64 * sizeof(double);
  static bool attr = false;
  if (!attr) {
    const int mx = (128 + DW + 1) * 64 * (int)sizeof(double);
    if (hipFuncSetAttribute((const void *)k_rbseq_walk_apply<DW, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_rbseq_walk_apply<DW, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_rbseq_walk_apply<DW, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_rbseq_walk_apply<DW, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) { (void)hipGetLastError(); return 0; }
    attr = true;
  }
  const int pb = nx > 64 ? (nx + 63) / 64 : 1, nblk = (nx + pb - 1) / pb, nt = level_streams(L);
  if (snapw) { if (rb & 1) hipLaunchKernelGGL((k_rbseq_walk_apply<DW, 1, true>), dim3(nblk), dim3(256), lds, st, *L, rb, ph, pb, nt);
               else hipLaunchKernelGGL((k_rbseq_walk_apply<DW, 0, true>), dim3(nblk), dim3(256), lds, st, *L, rb, ph, pb, nt); }
  else { if (rb & 1) hipLaunchKernelGGL((k_rbseq_walk_apply<DW, 1, false>), dim3(nblk), dim3(256), lds, st, *L, rb, ph, pb, nt);
         else hipLaunchKernelGGL((k_rbseq_walk_apply<DW, 0, false>), dim3(nblk), dim3(256), lds, st, *L, rb, ph, pb, nt); }
  return mgx_launched();
}

// rho of the level into *out (device memory, zero before the first call; mgx_api.cpp reads it back with the set-up's own synchronisation)
void mgxk_rbseq_rho(hipStream_t st, const LevView *L, double *out) {
  hipLaunchKernelGGL(k_rbseq_rho, dim3((L->ny + 63) / 64, (L->nx + 3) / 4), dim3(64, 4), 0, st, *L, (unsigned long long *)out);
}
// per row k = 1..nz the maximum over the columns of |g(k) / g(1)| into out[0..nz-1] (device memory, zero before the call)
void mgxk_rbseq_gdecay(hipStream_t st, const LevView *L, double *out) {
  const int ncol = L->nx * L->ny;
  int gx = (ncol + 255) / 256; if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_rbseq_gdecay, dim3(gx, L->nz), dim3(256), 0, st, *L, (unsigned long long *)out);
}
// rows (from the bottom) the correction reaches: the last one whose decay figure exceeds 2^-64 (all of them when none falls below, or when one is not a number)
int mgxk_rbseq_window_rows(const double *decay, int nz) {
  int k = nz;
  while (k > 1 && decay[k - 1] <= 5.421010862427522e-20) k--;   // 2^-64
  for (int q = 0; q < nz; q++) if (!(decay[q] >= 0.0) || decay[q] > 1e300) return nz;
  return k;
}
// d0 = y(k=1) - snapshot of the colour into u1, where the colour pass could not leave it (k_relax_tall)
void mgxk_rbseq_d0(hipStream_t st, const LevView *L, int rb) {
  const int nyh = L->ny / 2;
  hipLaunchKernelGGL(k_rbseq_d0, dim3((nyh + WAVE - 1) / WAVE, (L->nx + 3) / 4), dim3(WAVE, 4), 0, st, *L, rb);
}
// planes of warm-up after which a walk started from zero has forgotten its start to 2^-64: rho^m <= 2^-64; 0 = too many (or rho not a number)
int mgxk_rbseq_window_planes(double rho) {
  if (!(rho >= 0.0) || rho >= 1.0) return 0;
  if (rho < 1e-300) return 1;
  const double m = __builtin_ceil(64.0 * 0.6931471805599453 / -__builtin_log(rho));
  return m < 2.0 ? 2 : (m > (double)RBW_MAXM ? 0 : (int)m);
}
// (b) + (c) by the windowed walk (k_rbseq_window); needs d0 in u1 (the colour pass or k_rbseq_d0 wrote it) and m from mgxk_rbseq_window_planes.
// Returns 1 when launched.
int mgxk_rbseq_window(hipStream_t st, const LevView *L, int rb, Sides ph, int snapw, int m, int kcut) {
  const int nyh = L->ny / 2, nz = L->nz;
  if (L->gk == nullptr || m < 1 || m > RBW_MAXM || nyh < 1 || (L->ny & 1) || L->nx > 65535) return 0;
  int kr = nz % 64 == 0 ? 16 : (nz % 32 == 0 ? 8 : (nz % 16 == 0 ? 4 : (nz % 8 == 0 ? 2 : (nz % 4 == 0 ? 1 : 0))));
  if (!kr || nz / (4 * kr) > 65535) return 0;
  if (kcut < 1 || kcut > nz) kcut = nz;
  // the rows the correction reaches, spread over the four row waves of a workgroup (16 of 64 rows: four rows per wave rather than one wave with all
  // sixteen and three idle), and only the row groups that hold any are launched
  static const bool no_spread = getenv("MGX_RBW_NO_SPREAD") != nullptr;   // A/B
  while (!no_spread && kr > 1 && 4 * (kr / 2) >= kcut) kr /= 2;
  mgx_before_launch();
  static const bool no_xmap = getenv("MGX_RBSEQ_WINDOW_NO_XMAP") != nullptr;   // A/B
  // the walking wave at raised priority (s_setprio 3): level-1 sweep 0.3075-0.3150 -> 0.2989-0.3073 ms (three runs each, alternating); MGX_RBW_PRIO=0: A/B
  static const int prio = getenv("MGX_RBW_PRIO") ? atoi(getenv("MGX_RBW_PRIO")) : 1;
  static const int probe_m = getenv("MGX_RBW_PROBE_M") ? atoi(getenv("MGX_RBW_PROBE_M")) : -1;   // timing probe only (wrong results): another number of planes walked
  if (probe_m >= 0) m = probe_m;
  const int nt = level_streams(L), cpl = nyh <= WAVE ? 1 : 2, nch = (nyh + WAVE - 1) / WAVE, nkz = (kcut + 4 * kr - 1) / (4 * kr);
  const int xmap = !no_xmap && L->nx % 8 == 0 && (long long)nch * L->nx * nkz < (1LL << 31);
  const dim3 grd = xmap ? dim3(nch * L->nx * nkz) : dim3(nch, L->nx, nkz), blk(320);
#define WIN_CASE(CPLV, KRV)                                                                                          \
  { if (snapw) hipLaunchKernelGGL((k_rbseq_window<CPLV, KRV, true, (KRV <= 4 ? 4 : 8)>), grd, blk, 0, st, *L, rb, ph, m, nt, xmap, nch, nkz, prio, kcut); \
    else hipLaunchKernelGGL((k_rbseq_window<CPLV, KRV, false, (KRV <= 4 ? 4 : 8)>), grd, blk, 0, st, *L, rb, ph, m, nt, xmap, nch, nkz, prio, kcut); }
#define WIN_KR(CPLV) { if (kr == 16) WIN_CASE(CPLV, 16) else if (kr == 8) WIN_CASE(CPLV, 8) else if (kr == 4) WIN_CASE(CPLV, 4) else if (kr == 2) WIN_CASE(CPLV, 2) else WIN_CASE(CPLV, 1) }
  if (cpl == 1) WIN_KR(1) else WIN_KR(2)
#undef WIN_KR
#undef WIN_CASE
  return mgx_launched();
}

int mgxk_set_rbseq_timeout(double ms) {
  const long long ticks = (long long)(ms * 1e5);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_rbs_timeout_ticks), &ticks, sizeof ticks) == hipSuccess ? 0 : 1;
}

void mgxk_rbseq_apply(hipStream_t st, const LevView *L, int rb, Sides ph, int snapw) {
  const int nyh = L->ny / 2, nz = L->nz;
  int ku, kr;
  rbseq_apply_shape(L, &ku, &kr);
  const dim3 grd((nyh + WAVE - 1) / WAVE, (L->nx + 3) / 4, nz / kr), blk(WAVE, 4);
  const int nt = level_streams(L);
#define APPLY_CASE(KUV)                                                                                              \
  { if (snapw) hipLaunchKernelGGL((k_rbseq_apply<KUV, true>), grd, blk, 0, st, *L, rb, ph, kr, nt);                 \
    else hipLaunchKernelGGL((k_rbseq_apply<KUV, false>), grd, blk, 0, st, *L, rb, ph, kr, nt); }
  if (ku == 8) APPLY_CASE(8) else if (ku == 4) APPLY_CASE(4) else APPLY_CASE(2)
#undef APPLY_CASE
}

}  // extern "C"
