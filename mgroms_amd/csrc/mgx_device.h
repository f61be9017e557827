// Device helpers shared by the HIP translation units of libmgx.so (mgx_relax.hip, mgx_kernels.hip).
#pragma once
#include "mgx_internal.h"

#define WAVE 64
// Cache policy.  A level that cannot live in the 256 MB Infinity Cache (level 1 of the 512x512x64 problem: 1.2 GB)
// is streamed: what a pass reads exactly once, and what it writes, carries the non-temporal hint so that it does not
// evict the lines that ARE re-read inside the pass (the other j-half of a row, the i+-1 planes shared by two waves).
// Measured on the level-1 four-colour sweep: 0.35 -> 0.29 ms (loads alone 0.32, store alone 0.31).  Levels that fit
// the cache keep the default policy (the hint costs 20 % there), and so do loads whose lines another block of the same
// launch re-reads (the residual's two j-parities).  Chosen at compile time (ST) or per launch (level_streams).
#define NT_LOAD(ptr) __builtin_nontemporal_load(ptr)
#define NT2_STORE(v, ptr) __builtin_nontemporal_store(v, ptr)
template <bool NT> __device__ __forceinline__ double ld_stream(const double *p) { return NT ? NT_LOAD(p) : *p; }
__device__ __forceinline__ void st_rt(double *p, double v, int nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ double ld_rt(const double *p, int nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
static inline int level_streams(const LevView *L) { return (double)L->nx * L->ny * L->nz * 72.0 > 256e6; }

// store the physical-boundary images of an interior value (homogeneous Neumann mirror incl. the corner where two
// physical sides meet, mg_mpi_exchange.f90:509-537,552-597): lets the producing kernel fill its own halo
__device__ __forceinline__ void mirror_store(const LevView &L, double *__restrict__ a, const long long ro, const int j, const int i,
                                             const int c, const double v, const Sides ph) {
  const bool mS = ph.S && j == 1, mN = ph.N && j == L.ny, mW = ph.W && i == 1, mE = ph.E && i == L.nx;
  if (!(mS | mN | mW | mE)) return;
  const int cS = L.EO, cN = jpos(L, L.ny + 1);
  const long long o = (long long)i * L.plane + ro, oW = ro, oE = (long long)(L.nx + 1) * L.plane + ro;
  if (mS) a[o + cS] = v;
  if (mN) a[o + cN] = v;
  if (mW) { a[oW + c] = v; if (mS) a[oW + cS] = v; if (mN) a[oW + cN] = v; }
  if (mE) { a[oE + c] = v; if (mS) a[oE + cS] = v; if (mN) a[oE + cN] = v; }
}

// A rejected launch (more LDS or registers than this ROCm / device grants) must not pass for a completed colour pass: wrappers that
// have a generic fallback end in `return mgx_launched();` -- 0 = the launch was refused, nothing ran, the caller falls back; whatever
// has no fallback is caught by the sticky-error check of the next synchronising call (sync_stream in mgx_api.cpp).
// hipGetLastError() is sticky per thread for ANY earlier HIP call (an ignored attribute call, the caller's own runtime use): a wrapper that
// ends in mgx_launched() starts with mgx_before_launch(), which moves whatever is pending into mgx_pending_error (reported by the next
// synchronising call) so that mgx_launched() sees the status of THIS launch only.
extern thread_local hipError_t mgx_pending_error;  // mgx_api.cpp
static inline void mgx_before_launch() { const hipError_t e = hipGetLastError(); if (e != hipSuccess && mgx_pending_error == hipSuccess) mgx_pending_error = e; }
static inline int mgx_launched() { return hipGetLastError() == hipSuccess ? 1 : 0; }

static inline dim3 col_grid(int ncol_half, int nplanes, int z = 1) { return dim3((ncol_half + WAVE - 1) / WAVE, (nplanes + 3) / 4, z); }

// ---- switches and helpers shared by the smoother translation units ---------------------------------------------------
// MGX_PV: the matrix-free colour pass rebuilds the diagonal and the tridiagonal pivots in the kernel instead of streaming
// `bet` from HBM (relax_col_mf); -DMGX_NO_PV keeps the stored pivots for A/B measurements (same bits either way).
#ifdef MGX_NO_PV
#define MGX_PV 0
#else
#define MGX_PV 1
#endif
#ifndef MGX_GL
#define MGX_GL 1
#endif
// The j-1 and j+1 neighbours of a column sit side by side in the other half-row (jp = jm + 1): ONE 16-byte load per lane fetches
// both, instead of two 8-byte loads whose wave-wide footprints overlap by 63/64 (half the wave-level requests for these streams;
// 8-byte alignment only: gfx950 global loads do not need natural alignment)
#ifndef MGX_PAIR
#define MGX_PAIR 1
#endif
#ifndef MGX_ZW
#define MGX_ZW 1
#endif
// MGX_ZG: with MGX_ZW, the column's own slopes zy, zx are rebuilt in the kernel as well (relax_col_mf); -DMGX_ZG=0 streams them (A/B)
#ifndef MGX_ZG
#define MGX_ZG 1
#endif
// A quotient whose divisor is a per-column constant: the divisor's reciprocal is refined once with the two Newton steps of the hardware
// division sequence (v_rcp_f64, fma, fma, fma, fma) and each quotient then takes the sequence's last three operations (mul, fma, fma) --
// the same operations on the same values as `/` while no operand needs v_div_scale's rescaling (depths, metric factors: normal range),
// hence the same bits, at 3 instructions instead of ~11 per division.
#define RCP_REF(b) ({ const double b_ = (b); double r_ = __builtin_amdgcn_rcp(b_); double e_ = __builtin_fma(-b_, r_, 1.0); r_ = __builtin_fma(r_, e_, r_); \
                      e_ = __builtin_fma(-b_, r_, 1.0); __builtin_fma(r_, e_, r_); })
#define DIVC(a, b, rb) ({ const double a_ = (a); const double q_ = a_ * (rb); const double e_ = __builtin_fma(-(b), q_, a_); __builtin_fma(e_, (rb), q_); })
#if MGX_PAIR
#define LD_PAIR(ptr, A, B) { double2 t2_; __builtin_memcpy(&t2_, (ptr), 16); A = t2_.x; B = t2_.y; }
#else
#define LD_PAIR(ptr, A, B) { A = (ptr)[0]; B = (ptr)[1]; }
#endif

// the value of the neighbouring lane across the whole wave (DPP wave shifts of gfx9: one v_mov_b32_dpp per half, no LDS)
__device__ __forceinline__ double wave_shr1(double x) {  // lane n takes lane n-1's value, lane 0 takes 0
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1(double x) {  // lane n takes lane n+1's value, lane 63 takes 0
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

