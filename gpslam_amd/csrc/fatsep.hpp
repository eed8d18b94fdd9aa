// fatsep.hpp -- landmark elimination at scale (BASELINE config 4: 1e6 poses + 5e4 range landmarks with local
// visibility; the graph is matlab/PlazaPose2.m:55-66, :147-178 scaled, factors gpslam/slam/GPInterpolatedRangeFactorPose2.h:64-98).
//
// A dense landmark border (kernels.hpp k_lm_*) carries every landmark column through the whole chain: R = 1 + L ld
// right-hand sides, impossible for L = 5e4.  Here the chain is cut by nested dissection instead:
//   * CUT states every C states; every landmark is attached to ONE cut such that all states its factors touch lie in the
//     two segments next to that cut.  A cut state plus its landmarks is a FAT SEPARATOR (NB = 2d + ld * landmarks <= 80
//     columns).  compile() picks the smallest C for which every landmark fits (a landmark seen from more than two
//     segments does not; the segment length is doubled until all do).
//   * every segment interior (a plain block-tridiagonal chain) is eliminated against its NC = 2 NB + 1 border columns
//     [left fat | right fat | rhs]:  k_fs_factor (block Cholesky along the segment, one wave per segment),
//     k_fs_sweep (forward substitution of the border columns, one thread per column: Y = L^-1 G),
//     k_fs_syrk (the segment's Schur complement Y^T Y: a genuine dense GEMM with K = 2d * segment length, on
//     v_mfma_f64_16x16x4_f64),  k_fs_fat_assemble (direct terms - Schur complements -> fat blocks).
//   * the fat blocks form a block-tridiagonal system of K = N / C dense NB x NB blocks: block cyclic reduction over
//     level sets (k_fat_elim_* / k_fat_update per level, k_fat_top, k_fat_back_* per level in reverse).
//   * k_fs_rhs / k_fs_solve1: interior states by a single-rhs forward / backward sweep with the stored factors.
// Everything is summed in a fixed order (no atomics): results are bit-reproducible from run to run.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "devbuf.hpp"
#include "dpp.hpp"

namespace gps {

constexpr int kFatMax = 128;  // largest fat block (cut state + landmark columns): what one workgroup's LDS holds of ONE NB x NB operand
                              // (k_fat_elim_wide_mfma, k_fat_top, k_fat_back_w4: 132 KB of 160).  Config 4's graph needs 28-36, twice its
                              // landmark density 62, three times 96-100.
// columns of [H | H | g] per pass of k_fat_elim_wide_mfma: what is left of 160 KB beside the block (and the 4 x 4 factors in Ld)
inline int fat_wide_panel(int elem_bytes, int NB) {
  const long avail = 160L * 1024 - (long)(kFatMax / 4) * 10 * elem_bytes - (long)NB * (NB + 1) * elem_bytes - 512;
  long pw = avail / ((long)NB * elem_bytes);
  if (pw > 2 * NB + 1) pw = 2 * NB + 1;
  return (int)(pw < 4 ? 4 : pw);
}
constexpr int kFatLds = 80;   // round 3: 64 -> 80, what the LDS holds of all three NB x NB operands of k_fat_elim_mfma (155 KB of 160).
                              // Round 5: wider blocks keep the factor in LDS and stream [H | H | g] through it in column panels
                              // (k_fat_elim_wide_mfma).

template <typename T, typename TR = T> struct FsArgs {   // TR: type of the Jacobian row tables (kernels.hpp, LmArgs)
  int N, B, ld, L, K, NB, NC, NCP;   // K cuts / fat blocks; NC = 2 NB + 1 border columns; NCP = NC rounded up to 16
  int BS;                            // doubles per level-0 block record [D | O | g] (R = 1)
  const int *cuts;                   // K
  const int *segid;                  // N: segment of an interior state, -1 - k for cut k
  const int *fat_lm_ptr, *fat_lm;    // K + 1, landmarks of each fat block in slot order
  const int *lm_fat, *lm_slot;       // L
  const int *lmrow_ptr, *lmrow, *lmrow_state;   // rows touching each landmark, sorted by left state
  const int *lmpri_ptr, *lmpri;      // L + 1, prior ids per landmark
  // what a landmark's rows put on the right-hand side of the border columns, grouped by the interior state it lands on
  // (row rho's left half at its own state, its right half at the next one): groups per landmark sorted by state, members
  // (rho << 1 | half) per group; Gev[(g * ld + q) * B + r] = sum over the members of J[rho][half][r] * m[rho][q] (k_fs_gev)
  const int *lg_ptr, *lg_state, *lg_mptr, *lg_m;
  int ngroups;
  T *Gev;
  // the same right-hand sides as ENTRIES per chunk of a segment (k_fs_sweep_syrk): chunk ci = sc_base[seg] + (state - first interior
  // state) / (24 / B) holds entries [sc_ptr[ci], sc_ptr[ci + 1]); entry = (se_pk: (state - first interior state) << 8 | border
  // column,  se_src: g * ld + q, the B values at Gev + se_src * B)
  const int *sc_base, *sc_ptr, *se_pk, *se_src;
  // rows of each landmark that touch a cut state, by POSITION p of the landmark in fat_lm (k_fs_fat_assemble): items rho * 2 +
  // half (half 0: the row's left state is the cut, 1: its right state is), in the order of the landmark's row list.
  // fa: the landmark's own cut, fb: the next cut, fc: the previous one.  Null: walk the landmark's whole row list (fs_state_lm)
  const int *fa_ptr, *fa_it, *fb_ptr, *fb_it, *fc_ptr, *fc_it;
  T *lmMM;                           // L x ld x ld: sum over a landmark's rows of m m^T (+ its priors' weights on the diagonal), k_fs_lm_terms
  const double *pri_meas, *pri_sig;   // inputs are fp64 whatever T is (kernels.hpp, GpArgs)
  const double *lmk;
  const int *rowptr, *rowLm;
  const TR *rowLR, *rowE, *rowM;
  const T *blk;                      // N records [D | O | g]
  T *fac;                            // N x 2 B^2: [W = L^-1 (lower) | E = W O^T] of the interior states
  T *Y;                              // N x B x NCP
  T *Aseg;                           // (K - 1) x NCP x NCP
  T *Dfat, *link, *gfat, *Qbuf, *S1, *S2, *sv;
  T *xfat;                           // K x NB
  T *gL, *dL;                        // L ld
  T *x;                              // N x B solution (R = 1 layout of the level-0 solution array)
  T *rhs;                            // N x B
  T lambda;
  int last_fat_shared;               // split chains: the neighbour piece owns the damping of the last fat block's diagonal
  int *flag;
};

__device__ __forceinline__ double fs_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);          // the caller adds a residual step; without it the last-bit differences from
  y = y * fma(-0.5 * x * y, y, 1.5);          // sqrt() moved the small-graph comparison with the dense-border path by 1.7e-9
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
__device__ __forceinline__ float fs_rsqrt(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  return y * fmaf(-0.5f * x * y, y, 1.5f);
}
__device__ __forceinline__ void fs_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- block Cholesky along every segment interior.  One wave per segment.
// D~_j = D_j + lambda I - E_{j-1}^T E_{j-1} = L_j L_j^T;  W_j = L_j^-1;  E_j = W_j O_j^T  (O_j = H[j+1, j])
// A state is three LDS hand-overs: lanes r * B + c form D~_j; EVERY lane then factors the whole B x B block in registers
// (redundant, but a cooperative Cholesky is a barrier per pivot and column: the first version spent 4.6 us per state on 21
// of them, 1.18 ms per iteration at 1e6 states) and inverts L; lane r publishes row r of W; lanes r * B + c form E_j.
// The factors [W | E] leave in bursts of FL states through an LDS staging area: on gfx9-family parts loads and stores share
// one in-order counter (vmcnt), so a loop that stores every step and waits for a prefetched load every step drains BOTH
// every step.  (Measured at 1e6 states: 1.18 ms cooperative, 1.18 ms with the redundant factorisation alone, 1.04 ms with
// the store bursts, 0.89 ms with v_rsq in place of sqrt and division; what remains is VALU work -- 3908 waves x 255 states
// x ~700 instructions.)
template <typename T, int B, typename TR = T> __global__ void __launch_bounds__(64) k_fs_factor(FsArgs<T, TR> a) {
  const int seg = blockIdx.x, lane = threadIdx.x;
  const int j0 = a.cuts[seg] + 1, n = a.cuts[seg + 1] - a.cuts[seg] - 1;
  __shared__ T A[B * B], E[B * B], W[B * B], Os[B * B];
  constexpr int FL = 8;                                   // states per store burst
  __shared__ __attribute__((aligned(16))) T FACB[FL * 2 * B * B];
  typedef T V2 __attribute__((ext_vector_type(2)));
  auto flush = [&](int first, int cnt) {                  // states first .. first + cnt - 1 (contiguous in fac)
    V2 *dst = reinterpret_cast<V2 *>(a.fac + (size_t)first * 2 * B * B);
    const V2 *src = reinterpret_cast<const V2 *>(FACB);
    for (int q = lane; q < cnt * B * B; q += 64) dst[q] = src[q];     // (a nontemporal store changes nothing here)
  };
  // the chain is walked strictly in order (every step needs the previous state's E), so a step's only memory latency is
  // the fetch of its own block record [D | O]: a ring of DEPTH records is kept in flight (a step is shorter than a load)
  constexpr int PF = (B * B + 63) / 64, DEPTH = 4;
  T preD[DEPTH][PF], preO[DEPTH][PF];
  auto fetch = [&](int slot, int jn) {     // unconditional (clamped): a load under a branch would be waited for at once
    const T *bp = a.blk + (size_t)(j0 + min(jn, max(n - 1, 0))) * a.BS;
#pragma unroll
    for (int u = 0; u < PF; u++) {
      const int idx = min(lane + 64 * u, B * B - 1);
      preD[slot][u] = bp[idx];
      preO[slot][u] = bp[B * B + idx];
    }
  };
#pragma unroll
  for (int q = 0; q < DEPTH; q++) fetch(q, q);
  for (int j4 = 0; j4 < n; j4 += DEPTH) {
#pragma unroll
   for (int q = 0; q < DEPTH; q++) {
    const int jj = j4 + q;
    if (jj >= n) break;
    const int s = j0 + jj;
#pragma unroll
    for (int u = 0; u < PF; u++) {
      const int idx = lane + 64 * u;
      if (idx < B * B) {
        const int r = idx / B, c = idx - r * B;
        T v = preD[q][u] + (r == c ? a.lambda : T(0));
        if (jj > 0)
          for (int k = 0; k < B; k++) v -= E[k * B + r] * E[k * B + c];
        A[idx] = v;
        Os[idx] = preO[q][u];
      }
    }
    fetch(q, jj + DEPTH);
    fs_wave_sync();
    // ---- every lane: L (lower triangle, row-major packed) and W = L^-1 in registers
    T Lm[B][B], Wm[B][B], inv[B];
#pragma unroll
    for (int r = 0; r < B; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) Lm[r][c] = A[r * B + c];
    bool bad = false;
#pragma unroll
    for (int p = 0; p < B; p++) {
      T dd = Lm[p][p];
      if (!(dd > T(0))) { bad = true; dd = T(1); }
      // 1 / sqrt(dd) by v_rsq + two Newton steps instead of the IEEE sqrt and division sequences (~60 instructions of the
      // ~750 every lane executes per state; the kernel is VALU-bound: 3908 waves of redundant 6 x 6 factorisations)
      T y = fs_rsqrt(dd);
      T l = dd * y;
      l = fma(T(0.5) * y, fma(-l, l, dd), l);      // one residual step each: l and 1/l to the last bit or two
      y = fma(y, fma(-l, y, T(1)), y);
      inv[p] = y;
      Lm[p][p] = l;
#pragma unroll
      for (int r = p + 1; r < B; r++) Lm[r][p] *= inv[p];
#pragma unroll
      for (int r = p + 1; r < B; r++)
#pragma unroll
        for (int c = p + 1; c <= r; c++) Lm[r][c] -= Lm[r][p] * Lm[c][p];
    }
    if (bad && lane == 0) *a.flag = 1;
#pragma unroll
    for (int c = 0; c < B; c++) {          // column c of W by forward substitution
#pragma unroll
      for (int r = 0; r < B; r++) {
        if (r < c) { Wm[r][c] = T(0); continue; }
        T sacc = (r == c) ? T(1) : T(0);
#pragma unroll
        for (int k = c; k < r; k++) sacc -= Lm[r][k] * Wm[k][c];
        Wm[r][c] = sacc * inv[r];
      }
    }
#pragma unroll
    for (int r = 0; r < B; r++)
      if (lane == r) {
#pragma unroll
        for (int c = 0; c < B; c++) W[r * B + c] = Wm[r][c];
      }
    fs_wave_sync();
    T *fp = FACB + (size_t)(jj % FL) * 2 * B * B;
    for (int idx = lane; idx < B * B; idx += 64) {
      const int r = idx / B, c = idx - r * B;
      T e = T(0);
      for (int k = 0; k <= r; k++) e += W[r * B + k] * Os[c * B + k];   // (W O^T)[r][c] = sum_k W[r][k] O[c][k]
      fp[idx] = W[idx];
      fp[B * B + idx] = e;
      E[idx] = e;
    }
    fs_wave_sync();
    if (jj % FL == FL - 1 || jj == n - 1) {
      flush(s - (jj % FL), (jj % FL) + 1);
      fs_wave_sync();
    }
   }
  }
}

// ---- the same factorisation in the ROW LAYOUT of the chain solver (round 3; planar chains, B = 6, fp64): FOUR segments per wave,
// one per 16-lane DPP row; lane r < 6 of a row holds ROW r of the state's block.  The wave-per-segment kernel above lets every
// lane factor the whole 6 x 6 block redundantly (~700 VALU instructions per state and segment: 0.91 ms at config 4, VALU
// bound); here the block is factored cooperatively without a single barrier -- what a lane needs from another row arrives
// fused into the multiply-add (v_fmac_f64_dpp row_newbcast, dpp.hpp):
//   Cholesky, pivot p:  dd = A[p][p] broadcast;  every lane forms 1 / sqrt(dd);  L[r][p] = A[r][p] / l_pp;
//                       A[r][k] -= L[k][p] L[r][p] for all k at once (gather form: one source register, six lanes)
//   W = L^-1 by rows:   W_r = (e_r - sum_{k < r} L[r][k] W_k) / l_rr, row k broadcast as soon as it is final
//   E = W O^T:          E_r += W[r][m] (column m of O, held by lane m)
//   next block:         A' = D' + lambda I - E^T E,  (E^T E)[r][c] = sum_k E[k][r] E_k[c]; column r of E through a 288-byte LDS transpose
// ~370 instructions per state for four segments.  Records are fetched three states ahead (the only memory latency of a
// step); with that distance the in-order vmcnt counter never makes a step wait for its own stores.
template <int UNUSED = 0>   // (a template only so that the header may be included by several translation units)
__global__ void __launch_bounds__(64) k_fs_factor_rows6(FsArgs<double, double> a) {
  constexpr int B = 6, DEPTH = 3;
  const int lane = threadIdx.x, row = lane >> 4, r = lane & 15;
  const int nseg = a.K - 1;
  const int seg = min((int)blockIdx.x * 4 + row, nseg - 1);
  const bool segok = ((int)blockIdx.x * 4 + row) < nseg;
  const bool rowlane = r < B;
  const int rr = rowlane ? r : 0;
  const int j0 = a.cuts[seg] + 1, n = segok ? a.cuts[seg + 1] - a.cuts[seg] - 1 : 0;
  int nmax = n;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
  nmax = __builtin_amdgcn_readfirstlane(nmax);
  __shared__ double ET[4][B * B];
  const double lambda = a.lambda;
  double pD[DEPTH][B], pO[DEPTH][B];              // row r of D, column r of O of the states in flight
  auto fetch = [&](int slot, int jj) {            // unconditional (clamped): a load under a branch is waited for at once
    const double *bp = a.blk + (size_t)(j0 + min(jj, max(n - 1, 0))) * a.BS;
#pragma unroll
    for (int c = 0; c < B; c++) { pD[slot][c] = bp[rr * B + c]; pO[slot][c] = bp[B * B + c * B + rr]; }
  };
#pragma unroll
  for (int q = 0; q < DEPTH; q++) fetch(q, q);
  double Ecol[B], Er[B];
#pragma unroll
  for (int c = 0; c < B; c++) { Ecol[c] = 0.0; Er[c] = 0.0; }
  for (int j3 = 0; j3 < nmax; j3 += DEPTH) {
#pragma unroll
    for (int q = 0; q < DEPTH; q++) {
      const int jj = j3 + q;
      if (jj >= nmax) break;
      const bool live = jj < n;
      // ---- A = D + lambda I - E_{j-1}^T E_{j-1}
      double Ar[B], OT[B];
#pragma unroll
      for (int c = 0; c < B; c++) { Ar[c] = pD[q][c] + ((c == r) ? lambda : 0.0); OT[c] = pO[q][c]; }
      fetch(q, jj + DEPTH);
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, B>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        fmac_bcast6<k>(Ar, Er, -Ecol[k]);
      });
      // ---- Cholesky by rows
      double Lr[B], inv[B];
      bool bad = false;
      static_for<0, B>([&](auto pp) {
        constexpr int p = decltype(pp)::value;
        double dd = row_bcast<p>(Ar[p]);
        if (!(dd > 0.0)) { bad = true; dd = 1.0; }
        double y = fs_rsqrt(dd), l = dd * y;
        l = fma(0.5 * y, fma(-l, l, dd), l);        // one residual step each, as in k_fs_factor
        y = fma(y, fma(-l, y, 1.0), y);
        inv[p] = y;
        Lr[p] = (r > p) ? Ar[p] * y : ((r == p) ? l : 0.0);
        fmac_gather<6>(Ar, Lr[p], (r > p) ? -Lr[p] : 0.0);
      });
      if (bad && live && rowlane) *a.flag = 1;
      // ---- W = L^-1 by rows
      double Wr[B];
#pragma unroll
      for (int c = 0; c < B; c++) Wr[c] = (c == r) ? 1.0 : 0.0;
      static_for<0, B>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const double sc = (r == k) ? inv[k] : 1.0;
#pragma unroll
        for (int c = 0; c < B; c++) Wr[c] *= sc;
        fmac_self6<k>(Wr, (r > k) ? -Lr[k] : 0.0);
      });
      // ---- E = W O^T
#pragma unroll
      for (int c = 0; c < B; c++) Er[c] = 0.0;
      static_for<0, B>([&](auto mm) {
        constexpr int m = decltype(mm)::value;
        fmac_bcast6<m>(Er, OT, Wr[m]);
      });
      // ---- out: [W | E] of this state; column r of E for the next state's E^T E
      if (rowlane) {
#pragma unroll
        for (int c = 0; c < B; c++) ET[row][r * B + c] = Er[c];
      }
      if (live && rowlane) {
        double *fp = a.fac + (size_t)(j0 + jj) * 2 * B * B;
#pragma unroll
        for (int c = 0; c < B; c++) { fp[r * B + c] = Wr[c]; fp[B * B + r * B + c] = Er[c]; }
      }
      fs_wave_sync();
#pragma unroll
      for (int k = 0; k < B; k++) Ecol[k] = ET[row][k * B + rr];
      fs_wave_sync();
    }
  }
}

// ---- right-hand sides of the landmark border columns, one thread per (group, landmark component): the row tables are
// gathered HERE, fully parallel, so that k_fs_sweep's sequential steps read a dense, prefetchable stream instead of chasing
// lmrow_state -> lmrow -> rowM / rowLR (three dependent round trips per step: 5 us per step, 2.0 ms per iteration at 1e6 states)
template <typename T, int B, typename TR = T> __global__ void __launch_bounds__(256) k_fs_gev(FsArgs<T, TR> a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.ngroups * a.ld) return;
  const int g = t / a.ld, q = t - g * a.ld;
  T G[B];
#pragma unroll
  for (int r = 0; r < B; r++) G[r] = T(0);
  for (int k = a.lg_mptr[g]; k < a.lg_mptr[g + 1]; k++) {
    const int mm = a.lg_m[k], rho = mm >> 1, half = mm & 1;
    const T m = a.rowM[(size_t)rho * a.ld + q];
    const TR *row = a.rowLR + (size_t)rho * 2 * B + half * B;
#pragma unroll
    for (int r = 0; r < B; r++) G[r] += row[r] * m;
  }
#pragma unroll
  for (int r = 0; r < B; r++) a.Gev[(size_t)t * B + r] = G[r];
}

// ---- forward substitution of the border columns.  One thread per (segment, column):
// G~_j = G_j - E_{j-1}^T Y_{j-1},  Y_j = W_j G~_j.  Columns: [0, NB) left fat block, [NB, 2 NB) right fat block, 2 NB rhs.
// (round 5: up to 2 * 128 + 1 = 257 border columns, five waves)
template <typename T, int B, typename TR = T> __global__ void __launch_bounds__(320) k_fs_sweep(FsArgs<T, TR> a) {
  const int seg = blockIdx.x, c = threadIdx.x;
  const int cutL = a.cuts[seg], cutR = a.cuts[seg + 1];
  const int j0 = cutL + 1, n = cutR - cutL - 1;
  if (n <= 0) return;
  const int slast = j0 + n - 1;
  const bool is_rhs = (c == 2 * a.NB);
  const bool right = (!is_rhs && c >= a.NB);
  const int cc = is_rhs ? 0 : (right ? c - a.NB : c);
  const int kf = seg + (right ? 1 : 0);
  const bool is_state = !is_rhs && cc < B;
  bool active = c < a.NC;
  int lm = -1, q = 0;
  if (active && !is_rhs && !is_state) {
    const int li = (cc - B) / a.ld;
    q = (cc - B) - li * a.ld;
    if (a.fat_lm_ptr[kf] + li < a.fat_lm_ptr[kf + 1]) lm = a.fat_lm[a.fat_lm_ptr[kf] + li];
    if (lm < 0) active = false;     // padding column: Y stays zero (cleared once at compile time)
  }
  // ---- the column's right-hand sides.  Landmark columns: the groups of k_fs_gev, walked in order: the state of the next
  // group and its B values sit in registers one group ahead (gs, gn), so a step never waits for an index before it can ask
  // for data.  Groups of other segments (and of the cut states) come first in the landmark's list: skipped once.
  int gcur = 0, gend = 0;
  if (lm >= 0) { gcur = a.lg_ptr[lm]; gend = a.lg_ptr[lm + 1]; }
  while (gcur < gend && a.lg_state[gcur] < j0) gcur++;
  const int glast = max(a.ngroups - 1, 0);
  int gs = (gcur < gend) ? a.lg_state[gcur] : 0x7fffffff;
  T gn[B];
  {
    const T *gp = a.Gev + ((size_t)min(gcur, glast) * a.ld + q) * B;
#pragma unroll
    for (int r = 0; r < B; r++) gn[r] = gp[r];
  }
  T y[B];
#pragma unroll
  for (int r = 0; r < B; r++) y[r] = T(0);
  bool started = false;
  // [W_s | E_{s-1}] of the current state are the same for every column: staged through LDS, the next state's being
  // fetched (one value per thread, from a clamped address and without a branch around the load: the compiler waits for a
  // load under a branch right where it is issued) while the current step computes.  The same goes for the right-hand side
  // of the rhs column (every lane fetches the six values of the next state; one lane uses them).
  constexpr int FW = 2 * B * B;
  __shared__ T Fs[2][FW];
  const int tid = threadIdx.x;
  constexpr int PF = (FW + 63) / 64;          // values per thread (the block has at least 64 threads)
  const int nt = blockDim.x;
  for (int k = tid; k < FW; k += nt)
    Fs[0][k] = (k < B * B) ? a.fac[(size_t)j0 * FW + k] : T(0);       // E_{j0 - 1} = 0: the first interior state
  T rn[B];
  {
    const T *gp = a.blk + (size_t)j0 * a.BS + 2 * B * B;
#pragma unroll
    for (int r = 0; r < B; r++) rn[r] = gp[r];
  }
  __syncthreads();
  for (int jj = 0; jj < n; jj++) {
    const int s = j0 + jj;
    const T *fw = Fs[jj & 1], *fep = Fs[jj & 1] + B * B;
    // ---- this step's right-hand side (registers only, except the two cut-state couplings at the segment's ends)
    T G[B];
#pragma unroll
    for (int r = 0; r < B; r++) G[r] = T(0);
    bool consume = false;
    if (active) {
      if (is_rhs) {
        started = true;
#pragma unroll
        for (int r = 0; r < B; r++) G[r] = rn[r];
      } else if (is_state) {
        if (!right && jj == 0) {            // H[cutL + 1, cutL] = O_cutL
          started = true;
          const T *op = a.blk + (size_t)cutL * a.BS + B * B;
#pragma unroll
          for (int r = 0; r < B; r++) G[r] = op[r * B + cc];
        } else if (right && jj == n - 1) {  // H[cutR - 1, cutR] = O_{cutR-1}^T
          started = true;
          const T *op = a.blk + (size_t)s * a.BS + B * B;
#pragma unroll
          for (int r = 0; r < B; r++) G[r] = op[cc * B + r];
        }
      } else if (gs == s) {
        started = true;
        consume = true;
#pragma unroll
        for (int r = 0; r < B; r++) G[r] = gn[r];
      }
    }
    // ---- requests for the next step, all issued before this step's arithmetic: the next group of a column that has just
    // used one, the next state's factors, the next state's rhs.  (Y is identically zero until a column's first group /
    // coupling -- structurally, so nothing is computed or stored before that: the rows of Y were zeroed at compile time.)
    if (consume) gcur++;
    const int gidx = min(gcur, glast);
    const int gs_new = a.lg_state[gidx];
    T gnew[B];
    {
      const T *gp = a.Gev + ((size_t)gidx * a.ld + q) * B;
#pragma unroll
      for (int r = 0; r < B; r++) gnew[r] = gp[r];
    }
    const int sn = min(s + 1, slast);
    T pre[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) {
      const int k = min(tid + u * nt, FW - 1);
      pre[u] = a.fac[(size_t)(k < B * B ? sn : sn - 1) * FW + k];      // W_{s+1}[k] or E_s[k - B*B]
    }
    T rnew[B];
    {
      const T *gp = a.blk + (size_t)sn * a.BS + 2 * B * B;
#pragma unroll
      for (int r = 0; r < B; r++) rnew[r] = gp[r];
    }
    if (active && started) {
      if (jj > 0) {
#pragma unroll
        for (int k = 0; k < B; k++)
#pragma unroll
          for (int r = 0; r < B; r++) G[r] -= fep[k * B + r] * y[k];
      }
#pragma unroll
      for (int r = 0; r < B; r++) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k <= r; k++) acc += fw[r * B + k] * G[k];
        y[r] = acc;
      }
    }
    // ---- hand the prefetched values over (this is where the step waits for memory), then store
    if (consume) {
      gs = (gcur < gend) ? gs_new : 0x7fffffff;
#pragma unroll
      for (int r = 0; r < B; r++) gn[r] = gnew[r];
    }
#pragma unroll
    for (int r = 0; r < B; r++) rn[r] = rnew[r];
    if (jj + 1 < n) {
#pragma unroll
      for (int u = 0; u < PF; u++)
        if (tid + u * nt < FW) Fs[(jj + 1) & 1][tid + u * nt] = pre[u];
    }
    if (active && started) {
      T *yp = a.Y + (size_t)s * B * a.NCP + c;
#pragma unroll
      for (int r = 0; r < B; r++) {
        // Y is written once and read once, 2 ms later, by k_fs_syrk: a streaming (nontemporal) store
        __builtin_nontemporal_store(y[r], yp + (size_t)r * a.NCP);
      }
    }
    __syncthreads();
  }
}

// ---- Schur complement of a segment: Aseg = Y^T Y (NCP x NCP, K dimension = B * interior states) on the fp64 matrix
// cores.  One wave per (segment, 16-row tile): v_mfma_f64_16x16x4_f64, A[i][k] = Y[k][i0 + i], B[k][j] = Y[k][j0 + j]
// (lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15]); C: col = lane & 15, row = (lane >> 4) + 4 * reg.
typedef double fs_d4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline int fs_lds_stride(int ncp) { return ((ncp + 31) / 32) * 32 + 16; }   // (k_fs_syrk: of the instantiation's NCM)
// Y^T Y is symmetric: only the tiles on and below the diagonal (tile column <= tile row) are formed; readers use fs_sym.
// One 4-wave workgroup per segment.  The K dimension is walked in chunks of KC rows that the workgroup stages ONCE in
// LDS (coalesced 16-byte loads into registers while the matrix cores work on the current chunk, committed to the other
// LDS buffer afterwards); every wave owns a fixed subset of the tiles (round robin over the lower triangle) and feeds
// its MFMAs from LDS.  The first version let every (segment, tile row) wave stream its operand columns from L2 / HBM
// itself: 4.5x the traffic of Y, 4.8 ms at 1e6 states, bandwidth bound; staged, Y is read once (2.1 ms: 31 TFLOP/s of
// fp64 MFMA).  A variant with the tile loop specialised per tile count (branch-free k loop) needed 156 VGPRs and was
// slower (2.9 ms).  The sweep fused in front of this loop (thread per border column writing its Y rows straight into the LDS
// chunk, k_fs_sweep_syrk: no 4.6 GB Y buffer at all) was measured SLOWER as well: 5.0 ms against 2.34 + 2.11 ms -- 252 VGPRs
// (12 accumulator tiles + the sweep's state) leave two workgroups per CU, and per 24-row chunk the workgroup then pays the
// sweep's four dependent steps (LDS-fed 6 x 6 products on two of its four waves), the gather of the next chunk's right-hand
// sides and the MFMAs one after the other: 10 us per chunk instead of 4.3.
// Round 3: two instantiations (TPW accumulator tiles per wave, NCM = widest staged chunk row = LDS row stride): <7, 112> serves
// borders up to 112 columns (config 4: 80), <12, 144> everything wider.  Also measured: chunks requested two iterations ahead
// (two register sets, LDS-only barriers) -- no change, the kernel does not wait for its prefetch; piece / tile offsets formed
// once before the chunk loop with a compile-time stride (kept) -- 2 %.  SQ_INSTS_VALU / SQ_WAVES: ~1600 VALU instructions per
// wave and chunk around 24 MFMAs (profiles/round3_c4_kernel_stats.md).
template <int TPW, int NCM, typename TR = double> __global__ void __launch_bounds__(256) k_fs_syrk(FsArgs<double, TR> a) {
  constexpr int KC = 24;                                   // rows per chunk: 4 states of 6, 2 of 12, 6 of 4
  // LDS row stride: a multiple of 32 doubles plus 16, so that the four chunk rows an MFMA operand load touches (16 lanes x
  // 8 bytes each) fall into different bank groups; with the plain stride NCP = 96 they all hit the same banks (4-way
  // conflict on every operand load: the kernel was LDS-bandwidth bound at 0.40 of the matrix peak).  Round 3: the stride is
  // a COMPILE-TIME constant (that of the widest border the instantiation serves) and every per-thread piece / tile offset is
  // formed once, before the chunk loop: the loop had spent ~1600 VALU instructions per wave and chunk on the index
  // arithmetic of its runtime geometry (integer divisions by the row width, tile origins) around 24 MFMAs.
  constexpr int LSP = ((NCM + 31) / 32) * 32 + 16;
  extern __shared__ __align__(16) unsigned char syrk_smem[];
  double *buf = reinterpret_cast<double *>(syrk_smem);      // 2 x KC x LSP
  const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int NCP = a.NCP;
  // When the right-hand side column (index 2 NB) sits alone in the last 16-column tile (NB a multiple of 8: 40 at config 4's
  // round-2 landmark density), the matrix cores only see the 2 NB fat columns and the row of the Schur complement that belongs
  // to the right-hand side (2 NB dot products over the chunk rows) is summed on the vector ALU by two of the four waves out
  // of the same LDS chunk; the fetch then skips the padding columns as well.
  const bool rhs_alone = ((2 * a.NB) % 16) == 0;
  const int T16 = rhs_alone ? (2 * a.NB) / 16 : NCP / 16, ntiles = T16 * (T16 + 1) / 2;
  const int NCF = rhs_alone ? 2 * a.NB + 2 : NCP;           // columns fetched per chunk row (even: 16-byte pieces)
  // waves that also sum the rhs row, 64 columns each (its 2 NB entries left of the diagonal are the ones k_fs_fat_assemble reads).
  // Round 5: was "waves 0 and 1" whatever NB -- at NB = 72 and NB = 80 exactly the columns from 128 on were never summed
  const int nrw = (rhs_alone && blockIdx.y == 0) ? min((2 * a.NB + 63) / 64, 4) : 0;
  const int j0 = a.cuts[seg] + 1, n = a.cuts[seg + 1] - a.cuts[seg] - 1;
  const int kdim = n * a.B;
  const double *Yb = a.Y + (size_t)j0 * a.B * NCP;
  const int kl = lane >> 4, cl = lane & 15;
  fs_d4 acc[TPW];
  int aoff[TPW], boff[TPW], ooff[TPW];                     // operand offsets inside a chunk (row kl), output offset of the tile
  int nq = 0;                                               // tiles of this wave: p = (3 - wv) + 4 q of the lower triangle, row-major
#pragma unroll
  for (int q = 0; q < TPW; q++) {
    acc[q] = fs_d4{0.0, 0.0, 0.0, 0.0};
    // (the low waves get one tile fewer: they also sum the rhs row.  Round 5: borders beyond 176 columns deal the tiles over
    //  gridDim.y = 2 workgroups per segment -- both stage the whole chunk -- so that a wave's accumulators still fit its registers)
    const int pidx = (3 - wv) + 4 * (q * (int)gridDim.y + (int)blockIdx.y);
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= pidx) ti++;
    const int tj = pidx - ti * (ti + 1) / 2;
    aoff[q] = kl * LSP + ti * 16 + cl;
    boff[q] = kl * LSP + tj * 16 + cl;
    ooff[q] = (ti * 16 + kl) * NCP + tj * 16 + cl;
    if (pidx < ntiles) nq = q + 1;
  }
  nq = __builtin_amdgcn_readfirstlane(nq);
  typedef double V2 __attribute__((ext_vector_type(2)));
  const int chunk_v2 = KC * NCF / 2;                        // 16-byte pieces per chunk
  constexpr int PV = (KC * NCM / 2 + 255) / 256;            // pieces per thread at the widest chunk row of this instantiation
  V2 pre[PV];
  int prow[PV], goff[PV], loff[PV];                         // chunk row, offset in Y, offset in the LDS chunk of the thread's pieces
#pragma unroll
  for (int u = 0; u < PV; u++) {
    const int v = min(tid + u * 256, chunk_v2 - 1);
    const int row = (2 * v) / NCF, col = 2 * v - row * NCF;
    prow[u] = (tid + u * 256 < chunk_v2) ? row : 0x3fffffff;   // pieces beyond the chunk: never valid
    goff[u] = row * NCP + col;
    loff[u] = row * LSP + col;
  }
  auto fetch = [&](int c) {                                 // global -> registers (in flight under the MFMAs)
    const int k0 = c * KC;
    const double *yc = Yb + (size_t)k0 * NCP;
#pragma unroll
    for (int u = 0; u < PV; u++) {
      pre[u] = V2{0.0, 0.0};
      if (k0 + prow[u] < kdim) pre[u] = *reinterpret_cast<const V2 *>(yc + goff[u]);   // (a nontemporal load: no change)
    }
  };
  auto commit = [&](int which) {                            // registers -> LDS
    double *bc = buf + (size_t)which * KC * LSP;
#pragma unroll
    for (int u = 0; u < PV; u++)
      if (prow[u] < KC) *reinterpret_cast<V2 *>(bc + loff[u]) = pre[u];
  };
  const int nchunks = (kdim + KC - 1) / KC;
  if (nchunks > 0) { fetch(0); commit(0); }
  __syncthreads();
  double racc = 0.0;
  for (int c = 0; c < nchunks; c++) {
    if (c + 1 < nchunks) fetch(c + 1);
    const double *bb = buf + (size_t)(c & 1) * KC * LSP;
#pragma unroll
    for (int k4 = 0; k4 < KC; k4 += 4) {
#pragma unroll
      for (int q = 0; q < TPW; q++) {
        if (q < nq) {                                       // (wave-uniform)
          const double av = bb[aoff[q] + k4 * LSP], bv = bb[boff[q] + k4 * LSP];
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[q], 0, 0, 0);
        }
      }
    }
    if (wv < nrw) {                                          // row 2 NB of Y^T Y: column jr against the rhs column
      const int jr = min(lane + 64 * wv, 2 * a.NB);
#pragma unroll
      for (int k = 0; k < KC; k++) racc += bb[k * LSP + jr] * bb[k * LSP + 2 * a.NB];
    }
    if (c + 1 < nchunks) commit((c + 1) & 1);
    __syncthreads();
  }
  double *out = a.Aseg + (size_t)seg * NCP * NCP;
  if (wv < nrw && lane + 64 * wv <= 2 * a.NB) out[(size_t)(2 * a.NB) * NCP + lane + 64 * wv] = racc;
#pragma unroll
  for (int q = 0; q < TPW; q++) {
    if (q < nq) {
#pragma unroll
      for (int rg = 0; rg < 4; rg++) out[ooff[q] + 4 * rg * NCP] = acc[q][rg];
    }
  }
}
// ---- round 3: sweep and Schur complement in ONE launch, the border columns never reach memory.
// k_fs_sweep writes Y (N x B x NCP doubles: 3.84 GB at config 4) with streaming stores at 2.4 TB/s -- 1.6 ms -- and k_fs_syrk
// reads it back in 1.2 ms; together 2.8 of the iteration's 6.0 ms.  The round-2 fusion (every wave sweeps four steps, then
// multiplies: 5.0 ms against 4.45) serialised two latency chains in each wave.  Here the workgroup's waves are SPECIALISED, as
// in k_fused_level0: waves 4 and 5 are the sweep (one thread per border column, 64 columns each, state after state), waves
// 0..3 the matrix cores (tiles p = wave + 4 q of the lower triangle).  The sweep waves fill one half of a two-chunk LDS ring
// with the 24 rows of Y of the next chunk while the MFMA waves multiply the other half; one LDS-only barrier per chunk.
// What a step needs arrives a chunk ahead:
//   * the chunk's factors [W_s | E_{s-1}] and right-hand sides g_s: 24 / B states x (2 B^2 + B) doubles, one to five values per
//     lane, committed to a wave-private LDS copy (so the two sweep waves never have to meet);
//   * the landmark columns' right-hand sides as ENTRIES (FsArgs::sc_*, se_*): lane e of a sweep wave fetches entry e of the
//     chunk (descriptor two chunks ahead, its B values one chunk ahead) and drops it into the ring at (row, column) -- the
//     ring slot is zeroed by the column's own thread first; DS operations of one wave execute in order.  k_fs_sweep walked a
//     per-column list of groups instead: one dependent index -> data round trip per step and column.
// A step then is LDS and registers only: G = ring[row][c], G -= E^T y, y = W G, ring[row][c] = y (in place) -- the same
// expressions in the same order as k_fs_sweep, and the MFMA waves accumulate chunk after chunk in k_fs_syrk's order: the
// Schur complements are bit-identical to the two-launch path (test_gpu_segmented.py).
constexpr int fs_tri_row(int p) { int t = 0; while ((t + 1) * (t + 2) / 2 <= p) t++; return t; }
// Which of the workgroup's four waves owns tile p of the lower triangle.  Waves 0 and 1 are the sweep; a chunk of it costs what
// kFsSweepTiles tiles cost on the matrix cores (measured: ~2700 cycles against 6 x 64 per tile -- on MI355X the fp64 MFMA and
// the fp64 vector multiply-adds of one SIMD do not overlap, the matrix peak equals the vector peak), so the tiles are dealt
// greedily to the least loaded wave with that head start: every wave, hence every SIMD, carries the same work.
#ifndef GPS_FS_SWEEP_TILES
#define GPS_FS_SWEEP_TILES 7
#endif
constexpr int kFsSweepTiles = GPS_FS_SWEEP_TILES;
// (round 4: a border of at most 64 columns -- T16 <= 4, what the segment-length search now finds for config 4 -- is ONE sweep
//  wave's worth of columns: wave 1 then only multiplies, instead of walking the sweep's instruction stream with no column to own)
constexpr int fs_sweep_waves(int T16) { return T16 > 4 ? 2 : 1; }
constexpr int fs_tile_owner(int T16, int p) {
  const int NT = T16 * (T16 + 1) / 2;
  int load[4] = {kFsSweepTiles, fs_sweep_waves(T16) == 2 ? kFsSweepTiles : 0, 0, 0};
  int owner = 3;
  for (int q = 0; q <= p && q < NT; q++) {
    owner = 3;
    for (int w = 2; w >= 0; w--) if (load[w] < load[owner]) owner = w;
    load[owner]++;
  }
  return owner;
}
constexpr int fs_tile_rank(int T16, int p) {       // index of tile p among its owner's tiles
  const int o = fs_tile_owner(T16, p);
  int r = 0;
  for (int q = 0; q < p; q++) if (fs_tile_owner(T16, q) == o) r++;
  return r;
}
constexpr bool fs_panel_used(int T16, int w, int t) {   // does wave w touch the 16-column panel t?
  for (int q = 0; q < T16 * (T16 + 1) / 2; q++) {
    if (fs_tile_owner(T16, q) != w) continue;
    const int ti = fs_tri_row(q), tj = q - ti * (ti + 1) / 2;
    if (ti == t || tj == t) return true;
  }
  return false;
}
constexpr int fs_tile_count(int T16, int w) {
  int r = 0;
  for (int q = 0; q < T16 * (T16 + 1) / 2; q++) if (fs_tile_owner(T16, q) == w) r++;
  return r;
}

template <int T16, int WV, int LSP> struct FsTiles {
  static constexpr int NT = T16 * (T16 + 1) / 2, KC = 24, CNT = fs_tile_count(T16, WV), NA = CNT > 0 ? CNT : 1;
  fs_d4 acc[NA];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int q = 0; q < NA; q++) acc[q] = fs_d4{0.0, 0.0, 0.0, 0.0};
  }
  // one chunk: slot = the 24 x LSP rows of Y in LDS.  Every 16-column panel is fetched once per four rows in operand layout
  // (lane l: row l >> 4, column l & 15) and serves as A and as B operand of the tiles that touch it.
  __device__ __forceinline__ void chunk(const double *slot, int lane) {
    if constexpr (CNT > 0) {
      const double *bb = slot + (lane >> 4) * LSP + (lane & 15);
#pragma unroll
      for (int k4 = 0; k4 < KC; k4 += 4) {
        double pn[T16];
        static_for<0, T16>([&](auto tt) {                   // (only the panels this wave's tiles touch)
          constexpr int t = decltype(tt)::value;
          if constexpr (fs_panel_used(T16, WV, t)) pn[t] = bb[k4 * LSP + 16 * t];
          else pn[t] = 0.0;
        });
        static_for<0, NT>([&](auto pp) {
          constexpr int p = decltype(pp)::value;
          if constexpr (fs_tile_owner(T16, p) == WV) {
            constexpr int ti = fs_tri_row(p), tj = p - ti * (ti + 1) / 2, q = fs_tile_rank(T16, p);
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(pn[ti], pn[tj], acc[q], 0, 0, 0);
          }
        });
      }
    }
  }
  __device__ __forceinline__ void store(double *out, int NCP, int lane) const {
    const int kl = lane >> 4, cl = lane & 15;
    static_for<0, NT>([&](auto pp) {
      constexpr int p = decltype(pp)::value;
      if constexpr (fs_tile_owner(T16, p) == WV) {
        constexpr int ti = fs_tri_row(p), tj = p - ti * (ti + 1) / 2, q = fs_tile_rank(T16, p);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) out[(size_t)(ti * 16 + kl + 4 * rg) * NCP + tj * 16 + cl] = acc[q][rg];
      }
    });
  }
};

template <int T16, int WV, int LSP>
__device__ __forceinline__ void fs_mfma_role(const double *ring, int nchunks, int lane, double *out, int NCP) {
  constexpr int KC = 24;
  FsTiles<T16, WV, LSP> tl;
  tl.init();
#pragma unroll 1
  for (int i = 0; i <= nchunks; i++) {
    if (i >= 1) tl.chunk(ring + ((i - 1) & 1) * KC * LSP, lane);
    lds_barrier();
  }
  tl.store(out, NCP, lane);
}

#ifndef GPS_FS_PRIO
#define GPS_FS_PRIO 3
#endif
// the sweep wave SWV (0 / 1: border columns 64 SWV ..) of k_fs_sweep_syrk, including the few tiles fs_tile_owner deals to it
template <int B, int T16, int SWV, typename TR>
__device__ __forceinline__ void fs_sweep_role(const FsArgs<double, TR> &a, double *ring, double *FsAll, int seg, int lane, int cutL, int j0, int n,
                                              int nchunks, double *out) {
  constexpr int KC = 24, SPC = KC / B;
  constexpr int NCP = 16 * T16;
  constexpr int LSP = ((NCP + 31) / 32) * 32 + 16;
  constexpr int FW = 2 * B * B, SW = FW + B;                // staged per state: W_s | E_{s-1} | g_s
  constexpr int MP = ((FW + 15) / 16) * 16, MQ = MP / 16;   // the matrices of a state in LDS: FW elements padded to whole 16-lane rows
  constexpr int PF = (SPC * SW + 63) / 64;
  FsTiles<T16, SWV, LSP> tl;
  tl.init();
  constexpr int sw = SWV;
  const int c = sw * 64 + lane;                             // border column
  const int NB = a.NB;
  double *Fs = FsAll + sw * SPC * MP;
  const bool colok = c < NCP;
  const bool own_rhs = ((2 * NB) >> 6) == sw;               // (wave-uniform) this wave's columns include the right-hand side
  const bool lstate = c < B, rstate = (c >= NB && c < NB + B);
  const int slast = j0 + max(n, 1) - 1;
  const double *opL = a.blk + (size_t)cutL * a.BS + B * B, *opR = a.blk + (size_t)slast * a.BS + B * B;
  const int cb = a.sc_base[seg];
  double y[B];
#pragma unroll
  for (int r = 0; r < B; r++) y[r] = 0.0;
  double pre[PF], val[B];
  const int clast = max(nchunks - 1, 0);
  // every staged value of a lane has a fixed place in the chunk: (state t of the chunk, element k of [W | E | g]) -> base
  // pointer, stride per state and LDS destination are formed once
  const double *sbase[PF];
  int sstr[PF], sst[PF], sdst[PF];                          // sdst: >= 0 offset in Fs, -1 - r: rhs element r (ring), INT_MIN: nothing
  bool szero[PF], sneg[PF];
  static_assert(MP > FW, "the matrices' last 16-lane row has a pad word");
#pragma unroll
  for (int u = 0; u < PF; u++) {
    const int v = min(lane + 64 * u, SPC * SW - 1);
    const int t = v / SW, k = v - t * SW;
    sst[u] = t;
    sbase[u] = (k < B * B) ? a.fac + k : (k < FW) ? a.fac + k - FW : a.blk + FW + (k - FW);
    sstr[u] = (k < FW) ? FW : a.BS;
    sdst[u] = (lane + 64 * u >= SPC * SW) ? (int)0x80000000 : (k < FW ? t * MP + k : -1 - (t * B + (k - FW)));
    szero[u] = (k >= B * B && k < FW && t == 0);            // E_{j0 - 1} = 0 (first chunk only): the first interior state
    sneg[u] = (k >= B * B && k < FW);
  }
  auto stage = [&](int i) {                                 // factors + rhs of chunk i -> pre (clamped: never out of the segment)
    const int s0 = j0 + min(i, clast) * SPC;
#pragma unroll
    for (int u = 0; u < PF; u++) pre[u] = sbase[u][(size_t)min(s0 + sst[u], slast) * sstr[u]];
  };
  auto ent_range = [&](int i, int &e0, int &e1) {
    const int ci = cb + min(i, clast);
    e0 = a.sc_ptr[ci]; e1 = a.sc_ptr[ci + 1];
    if (i >= nchunks) e1 = e0;
  };
  auto ent_desc = [&](int e0, int e1, int &pk, int &src) {  // (the entry arrays carry 64 entries of slack)
    const int e = e0 + lane;
    pk = a.se_pk[e]; src = a.se_src[e];
    if (e >= e1) { pk = -1; src = 0; }
  };
  auto ent_vals = [&](int src) {
    const double *gp = a.Gev + (size_t)src * B;
#pragma unroll
    for (int r = 0; r < B; r++) val[r] = gp[r];
  };
  // the pipeline: ranges three chunks ahead, descriptors two, values one -- no request waits for one of the same chunk
  int pk_cur = -1, pk_n1 = -1, src_n1 = 0;
  int e0c = 0, e1c = 0, e0n1 = 0, e1n1 = 0, e0n2 = 0, e1n2 = 0;
  if (nchunks > 0) {
    stage(0);
    ent_range(0, e0c, e1c);
    ent_range(1, e0n1, e1n1);
    ent_range(2, e0n2, e1n2);
    int src0;
    ent_desc(e0c, e1c, pk_cur, src0);
    ent_desc(e0n1, e1n1, pk_n1, src_n1);
    ent_vals(src0);
  }
#pragma unroll 1
  for (int i = 0; i <= nchunks; i++) {
#if GPS_FS_PRIO
    __builtin_amdgcn_s_setprio(GPS_FS_PRIO);   // the sweep is what the workgroup's other waves wait for at the chunk barrier
#endif
    if (i < nchunks) {
      double *slot = ring + (i & 1) * KC * LSP;
      // ---- the chunk's right-hand sides: zero the column, then the staged values and the entries
      if (colok) {
#pragma unroll
        for (int row = 0; row < KC; row++) slot[row * LSP + c] = 0.0;
      }
      // (branch-free: a value that has no place goes to a word nobody reads -- the pad of the matrices' last 16-lane row, a
      // ring column beyond NCP)
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const double v = (szero[u] && i == 0) ? 0.0 : (sneg[u] ? -pre[u] : pre[u]);                     // W, -E
        Fs[sdst[u] >= 0 ? sdst[u] : MP - 1] = (sdst[u] >= 0) ? v : 0.0;
      }
      if (own_rhs) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
          const bool isg = sdst[u] < 0 && sdst[u] != (int)0x80000000 && i * SPC + sst[u] < n;
          const int rw = isg ? -1 - sdst[u] : 0;
          slot[rw * LSP + (isg ? 2 * NB : NCP + 8)] = pre[u];
        }
      }
      if (pk_cur >= 0) {
        const int col = pk_cur & 255, row = ((pk_cur >> 8) - i * SPC) * B;
        if ((col >> 6) == sw) {
#pragma unroll
          for (int r = 0; r < B; r++) slot[(row + r) * LSP + col] = val[r];
        }
      }
      // the cut states' couplings, twice per segment.  HERE, where the wave waits for its staged values anyway: a load under
      // a branch costs every lane a full vmcnt(0) at the join, taken or not -- inside the steps that drained the requests
      // of the chunks ahead four times per chunk (4.0 ms instead of 2.8 for the two launches)
      if (i == 0 && lstate) {
#pragma unroll
        for (int r = 0; r < B; r++) slot[r * LSP + c] = opL[r * B + c];                                    // H[cutL + 1, cutL] = O_cutL
      }
      if (i == nchunks - 1 && rstate) {
        const int row = (n - 1 - i * SPC) * B;
#pragma unroll
        for (int r = 0; r < B; r++) slot[(row + r) * LSP + c] = opR[(c - NB) * B + r];                     // H[cutR - 1, cutR] = O_{cutR-1}^T
      }
      for (int e = e0c + 64 + lane; e < e1c; e += 64) {     // more than 64 entries in one chunk: the rest, synchronously
        const int pk = a.se_pk[e];
        const double *gp = a.Gev + (size_t)a.se_src[e] * B;
        const int col = pk & 255, row = ((pk >> 8) - i * SPC) * B;
        if ((col >> 6) == sw) {
#pragma unroll
          for (int r = 0; r < B; r++) slot[(row + r) * LSP + col] = gp[r];
        }
      }
      fs_wave_sync();
      // ---- requests of the chunks ahead (in flight under this chunk's steps)
      stage(i + 1);
      pk_cur = pk_n1; e0c = e0n1; e1c = e1n1;
      ent_vals(src_n1);
      e0n1 = e0n2; e1n1 = e1n2;
      ent_desc(e0n1, e1n1, pk_n1, src_n1);
      ent_range(i + 3, e0n2, e1n2);
      // ---- the steps.  The matrices are the same for every column: lane j of each 16-lane row holds elements j, 16 + j, ...
      // of [W_s | -E_{s-1}] (MQ registers) and every multiply-add takes its matrix element by DPP row broadcast
      // (v_fmac_f64_dpp, dpp.hpp fmac_mat).  The first version read them from LDS as broadcast operands, two per ds_read_b128
      // right in front of the multiply-adds that use them: ~30 exposed LDS round trips per step, 12 800 cycles per chunk.
#pragma unroll
      for (int t = 0; t < SPC; t++) {
        const int jj = i * SPC + t;
        if (jj < n) {
          double Mr[MQ];
#pragma unroll
          for (int q = 0; q < MQ; q++) Mr[q] = Fs[t * MP + q * 16 + (lane & 15)];
          double G[B], yn[B];
#pragma unroll
          for (int r = 0; r < B; r++) { G[r] = colok ? slot[(t * B + r) * LSP + c] : 0.0; yn[r] = 0.0; }
          static_for<0, B>([&](auto kk) {                  // G -= E^T y  (E_{j0-1} is staged as zero)
            constexpr int k = decltype(kk)::value;
            fmac_mat<B, B * B + k * B, 1>(G, Mr, y[k]);
          });
          static_for<0, B>([&](auto kk) {                  // y = W G, W lower triangular: column k serves rows k .. B - 1
            constexpr int k = decltype(kk)::value;
            fmac_mat<B - k, k * B + k, B>(yn + k, Mr, G[k]);
          });
#pragma unroll
          for (int r = 0; r < B; r++) y[r] = yn[r];
          if (colok) {
#pragma unroll
            for (int r = 0; r < B; r++) slot[(t * B + r) * LSP + c] = y[r];
          }
        }
      }
    }
#if GPS_FS_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (i >= 1) tl.chunk(ring + ((i - 1) & 1) * KC * LSP, lane);
    lds_barrier();
  }
  tl.store(out, NCP, lane);
}

// NCP == 16 * T16 exactly (the caller picks the instantiation); 2 NB + 1 <= 128 columns; 256 threads: waves 0, 1 sweep (wave 1 only
// when there are more than 64 columns), 2, 3 multiply
template <int B, int T16, typename TR = double> __global__ void __launch_bounds__(256, 3) k_fs_sweep_syrk(FsArgs<double, TR> a) {
  constexpr int KC = 24, SPC = KC / B;                      // rows / states per chunk
  constexpr int NCP = 16 * T16;
  constexpr int LSP = ((NCP + 31) / 32) * 32 + 16;          // (see k_fs_syrk)
  static_assert(B <= 6, "fmac_mat blocks hold up to six multiply-adds");
  extern __shared__ __align__(16) unsigned char fsy_smem[];
  double *ring = reinterpret_cast<double *>(fsy_smem);      // 2 x KC x LSP
  double *FsAll = ring + 2 * KC * LSP;                      // 2 sweep waves x SPC x MP: [W_s | -E_{s-1}] per state of the chunk
  const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cutL = a.cuts[seg], cutR = a.cuts[seg + 1];
  const int j0 = cutL + 1, n = cutR - cutL - 1;
  const int nchunks = n > 0 ? (n + SPC - 1) / SPC : 0;
  double *out = a.Aseg + (size_t)seg * NCP * NCP;
  switch (wv) {
    case 0: fs_sweep_role<B, T16, 0, TR>(a, ring, FsAll, seg, lane, cutL, j0, n, nchunks, out); break;
    case 1:
      if constexpr (fs_sweep_waves(T16) == 2) fs_sweep_role<B, T16, 1, TR>(a, ring, FsAll, seg, lane, cutL, j0, n, nchunks, out);
      else fs_mfma_role<T16, 1, LSP>(ring, nchunks, lane, out, NCP);
      break;
    case 2: fs_mfma_role<T16, 2, LSP>(ring, nchunks, lane, out, NCP); break;
    default: fs_mfma_role<T16, 3, LSP>(ring, nchunks, lane, out, NCP); break;
  }
}

// dynamic LDS of k_fs_sweep_syrk<B, NCP / 16>: the two-chunk ring + the two sweep waves' factor copies
inline size_t fs_fused_smem(int b, int ncp) {
  return ((size_t)2 * 24 * fs_lds_stride(ncp) + (size_t)2 * (24 / b) * ((2 * b * b + 15) / 16 * 16)) * sizeof(double);
}
// entry (i, j) of a segment's symmetric Schur complement, stored by its lower tile triangle
template <typename T> __device__ __forceinline__ T fs_sym(const T *A, int ncp, int i, int j) {
  return (i >= j) ? A[(size_t)i * ncp + j] : A[(size_t)j * ncp + i];
}

// ---- fat blocks: direct terms minus the Schur complements of the two neighbouring segments.
// variable v of fat block k: v < B -> component v of the cut state; else landmark fat_lm[ptr[k] + (v - B) / ld],
// component (v - B) % ld, or padding (unit diagonal).
template <typename T> struct FsVar { int kind, lm, q; };   // kind 0 state, 1 landmark, 2 padding
template <typename T, typename TR> __device__ __forceinline__ FsVar<T> fs_var(const FsArgs<T, TR> &a, int k, int v) {
  FsVar<T> o;
  if (v < a.B) { o.kind = 0; o.lm = -1; o.q = v; return o; }
  const int li = (v - a.B) / a.ld;
  o.q = (v - a.B) - li * a.ld;
  if (a.fat_lm_ptr[k] + li < a.fat_lm_ptr[k + 1]) { o.kind = 1; o.lm = a.fat_lm[a.fat_lm_ptr[k] + li]; }
  else { o.kind = 2; o.lm = -1; }
  return o;
}
// sum over the rows of landmark lm of (Jacobian entry of state `st`, component r) * m_q
template <typename T, typename TR> __device__ __forceinline__ T fs_state_lm(const FsArgs<T, TR> &a, int st, int r, int lm, int q) {
  T v = T(0);
  for (int t = a.lmrow_ptr[lm]; t < a.lmrow_ptr[lm + 1]; t++) {
    const int ls = a.lmrow_state[t];
    if (ls != st && ls != st - 1) continue;
    const int rho = a.lmrow[t];
    v += a.rowLR[(size_t)rho * 2 * a.B + (ls == st ? 0 : a.B) + r] * a.rowM[(size_t)rho * a.ld + q];
  }
  return v;
}
// ---- per landmark: its own block of the normal equations, sum_rows m m^T (+ prior weights), and its gradient -sum_rows m e
// (- prior terms).  One thread per landmark, the sums in the order k_fs_fat_assemble used to form them in (round 3: that kernel
// re-walked the landmark's row list for each of its ld x ld + ld entries, two dependent loads per row, inside the fat block's
// workgroup).
template <typename T, typename TR = T> __global__ void __launch_bounds__(128) k_fs_lm_terms(FsArgs<T, TR> a) {
  const int lm = blockIdx.x * blockDim.x + threadIdx.x;
  if (lm >= a.L) return;
  constexpr int LDM = 3;                       // landmark dimension <= 3 (Point2 / Point3)
  const int ld = a.ld;
  T v[LDM][LDM], g[LDM];
#pragma unroll
  for (int q = 0; q < LDM; q++) {
    g[q] = T(0);
#pragma unroll
    for (int q2 = 0; q2 < LDM; q2++) v[q][q2] = T(0);
  }
  for (int t = a.lmrow_ptr[lm]; t < a.lmrow_ptr[lm + 1]; t++) {     // ONE pass over the rows; every sum keeps its row order
    const int rho = a.lmrow[t];
    TR m[LDM];                                 // (products in the row tables' type, as k_fs_fat_assemble formed them)
#pragma unroll
    for (int q = 0; q < LDM; q++) m[q] = (q < ld) ? a.rowM[(size_t)rho * ld + q] : TR(0);
    const TR e = a.rowE[rho];
#pragma unroll
    for (int q = 0; q < LDM; q++) {
#pragma unroll
      for (int q2 = 0; q2 < LDM; q2++) v[q][q2] += m[q] * m[q2];
      g[q] -= m[q] * e;
    }
  }
  for (int t = a.lmpri_ptr[lm]; t < a.lmpri_ptr[lm + 1]; t++) {
    const int pk = a.lmpri[t];
#pragma unroll
    for (int q = 0; q < LDM; q++) {
      if (q >= ld) continue;
      const T w = T(1) / T(a.pri_sig[(size_t)pk * ld + q]);
      v[q][q] += w * w;
      g[q] -= w * w * T(a.lmk[(size_t)lm * ld + q] - a.pri_meas[(size_t)pk * ld + q]);
    }
  }
#pragma unroll
  for (int q = 0; q < LDM; q++) {
    if (q >= ld) continue;
#pragma unroll
    for (int q2 = 0; q2 < LDM; q2++) if (q2 < ld) a.lmMM[((size_t)lm * ld + q) * ld + q2] = v[q][q2];
    a.gL[(size_t)lm * ld + q] = g[q];          // undamped gradient (LM model)
  }
}

// the same sum from the precomputed items of the landmark at position p (round 3): fs_state_lm scans all ~9-18 rows of the landmark
// through three dependent loads each to find the zero to two that touch the cut -- 720 such scans per fat block, 0.32 ms per
// iteration at config 4; the items are found once, at compile() time.  Same rows in the same order: bit-identical.
template <typename T, typename TR> __device__ __forceinline__ T fs_state_lm_items(const FsArgs<T, TR> &a, const int *ptr, const int *it, int p, int r, int q) {
  T v = T(0);
  for (int t = ptr[p]; t < ptr[p + 1]; t++) {
    const int rho = it[t] >> 1, half = it[t] & 1;
    v += a.rowLR[(size_t)rho * 2 * a.B + (half ? a.B : 0) + r] * a.rowM[(size_t)rho * a.ld + q];
  }
  return v;
}
template <typename T, typename TR = T> __global__ void __launch_bounds__(256) k_fs_fat_assemble(FsArgs<T, TR> a) {
  const int k = blockIdx.x, NB = a.NB, B = a.B;
  const bool items = a.fa_ptr != nullptr;
  const int p0 = a.fat_lm_ptr[k], p1 = (k < a.K - 1) ? a.fat_lm_ptr[k + 1] : 0;     // positions of the landmarks of blocks k, k + 1
  const int cut = a.cuts[k];
  const T *AL = (k > 0) ? a.Aseg + (size_t)(k - 1) * a.NCP * a.NCP : nullptr;   // segment on the left: this block is its RIGHT fat
  const T *AR = (k < a.K - 1) ? a.Aseg + (size_t)k * a.NCP * a.NCP : nullptr;
  const T *bp = a.blk + (size_t)cut * a.BS;
  for (int idx = threadIdx.x; idx < NB * NB; idx += blockDim.x) {
    const int r = idx / NB, c = idx - r * NB;
    const FsVar<T> vr = fs_var(a, k, r), vc = fs_var(a, k, c);
    T v = T(0);
    if (vr.kind == 2 || vc.kind == 2) {
      v = (r == c) ? T(1) : T(0);
    } else {
      if (vr.kind == 0 && vc.kind == 0) v = bp[r * B + c];
      else if (vr.kind == 0) v = items ? fs_state_lm_items(a, a.fa_ptr, a.fa_it, p0 + (c - B) / a.ld, r, vc.q) : fs_state_lm(a, cut, r, vc.lm, vc.q);
      else if (vc.kind == 0) v = items ? fs_state_lm_items(a, a.fa_ptr, a.fa_it, p0 + (r - B) / a.ld, c, vr.q) : fs_state_lm(a, cut, c, vr.lm, vr.q);
      else if (vr.lm == vc.lm && items) v = a.lmMM[((size_t)vr.lm * a.ld + vr.q) * a.ld + vc.q];
      else if (vr.lm == vc.lm) {
        for (int t = a.lmrow_ptr[vr.lm]; t < a.lmrow_ptr[vr.lm + 1]; t++) {
          const int rho = a.lmrow[t];
          v += a.rowM[(size_t)rho * a.ld + vr.q] * a.rowM[(size_t)rho * a.ld + vc.q];
        }
        if (vr.q == vc.q)
          for (int t = a.lmpri_ptr[vr.lm]; t < a.lmpri_ptr[vr.lm + 1]; t++) {
            const T w = T(1) / T(a.pri_sig[(size_t)a.lmpri[t] * a.ld + vr.q]);
            v += w * w;
          }
      }
      if (r == c && !(a.last_fat_shared && k == a.K - 1)) v += a.lambda;
      if (AL) v -= fs_sym(AL, a.NCP, NB + r, NB + c);
      if (AR) v -= fs_sym(AR, a.NCP, r, c);
    }
    a.Dfat[(size_t)k * NB * NB + idx] = v;
    if (k < a.K - 1) {       // link k -> k + 1: H[fat k+1 variable r, fat k variable c]
      const FsVar<T> wr = fs_var(a, k + 1, r);
      const int cut1 = a.cuts[k + 1];
      T o = T(0);
      if (wr.kind != 2 && vc.kind != 2) {
        if (wr.kind == 0 && vc.kind == 0) { if (cut1 == cut + 1) o = bp[B * B + r * B + c]; }
        else if (wr.kind == 0 && vc.kind == 1) o = items ? fs_state_lm_items(a, a.fb_ptr, a.fb_it, p0 + (c - B) / a.ld, r, vc.q) : fs_state_lm(a, cut1, r, vc.lm, vc.q);
        else if (wr.kind == 1 && vc.kind == 0) o = items ? fs_state_lm_items(a, a.fc_ptr, a.fc_it, p1 + (r - B) / a.ld, c, wr.q) : fs_state_lm(a, cut, c, wr.lm, wr.q);
        o -= fs_sym(AR, a.NCP, NB + r, c);
      }
      a.link[(size_t)k * NB * NB + idx] = o;
    }
  }
  for (int r = threadIdx.x; r < NB; r += blockDim.x) {
    const FsVar<T> vr = fs_var(a, k, r);
    T g = T(0);
    if (vr.kind == 0) g = bp[2 * B * B + r];
    else if (vr.kind == 1 && items) g = a.gL[(size_t)vr.lm * a.ld + vr.q];
    else if (vr.kind == 1) {
      for (int t = a.lmrow_ptr[vr.lm]; t < a.lmrow_ptr[vr.lm + 1]; t++) {
        const int rho = a.lmrow[t];
        g -= a.rowM[(size_t)rho * a.ld + vr.q] * a.rowE[rho];
      }
      for (int t = a.lmpri_ptr[vr.lm]; t < a.lmpri_ptr[vr.lm + 1]; t++) {
        const int pk = a.lmpri[t];
        const T w = T(1) / T(a.pri_sig[(size_t)pk * a.ld + vr.q]);
        g -= w * w * T(a.lmk[(size_t)vr.lm * a.ld + vr.q] - a.pri_meas[(size_t)pk * a.ld + vr.q]);
      }
      a.gL[(size_t)vr.lm * a.ld + vr.q] = g;   // undamped gradient (LM model)
    }
    if (vr.kind != 2) {
      if (AL) g -= fs_sym(AL, a.NCP, NB + r, 2 * NB);
      if (AR) g -= fs_sym(AR, a.NCP, r, 2 * NB);
    }
    a.gfat[(size_t)k * NB + r] = g;
  }
}

// ---- block cyclic reduction over the fat blocks.  One level: eliminate every other active block m (left neighbour l,
// right neighbour r or -1):  D_m = L L^T,  P = L^-1 H[m, l],  Q = L^-1 H[m, r],  z = L^-1 g_m;
// S1 = P^T P -> D_l,  S2 = Q^T Q -> D_r,  new link H[r, l] = -Q^T P,  P^T z -> g_l,  Q^T z -> g_r.
struct FatLevel {
  const int *elim;     // 6 ints per eliminated block: m, l, r, link(l -> m), link(m -> r), new link(l -> r)
  int nelim;
  const int *upd;      // 3 ints per survivor: a, eliminated block on its left (or -1), on its right (or -1)
  int nupd;
};

// Cholesky of the NB x NB block in Lm (lower triangle, stride LS; NB is a multiple of 4) and X <- L^-1 X for the NX columns of X
// (stride XS), four pivots at a time: every thread factors the 4 x 4 diagonal block in registers (ten LDS broadcasts, no
// barrier), the rows below it and the four rows of X are multiplied by its inverse (one row / one column per thread), a
// barrier, then ONE rank-4 update of the trailing triangle and of the rows of X below (eight operand reads that do not depend
// on each other per entry), a barrier.  NB / 4 sequential steps of two barriers instead of NB steps of three: the unblocked
// version spent 79 us per block at NB = 40, of which its waves sat 2/3 at barriers or behind single exposed LDS round trips
// (one wave per SIMD: nothing else to issue).  The 4 x 4 factors go to Ld (10 per step: l00 l10 l11 l20 l21 l22 l30 l31 l32 l33);
// the diagonal blocks of Lm keep their pre-factor values, so nobody waits before overwriting them.
template <typename T> __device__ __forceinline__ void fat_factor_panel4(T *Lm, T *X, T *Ld, int NB, int LS, int XS, int NX, int *flag) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int tx = tid & 63, ty = tid >> 6, nty = nt >> 6;
  for (int p = 0; p < NB; p += 4) {
    // ---- 4 x 4 diagonal block: A = L L^T, W = L^-1 (both lower), every thread
    T A[4][4], L[4][4], W[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) A[i][j] = Lm[(p + i) * LS + p + j];
    bool bad = false;
    T inv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      T dd = A[j][j];
#pragma unroll
      for (int k = 0; k < j; k++) dd -= L[j][k] * L[j][k];
      if (!(dd > T(0))) { bad = true; dd = T(1); }
      T y = fs_rsqrt(dd), l = dd * y;
      l = fma(T(0.5) * y, fma(-l, l, dd), l);        // one residual step each, as in k_fs_factor
      y = fma(y, fma(-l, y, T(1)), y);
      L[j][j] = l; inv[j] = y;
#pragma unroll
      for (int i = j + 1; i < 4; i++) {
        T v = A[i][j];
#pragma unroll
        for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
        L[i][j] = v * y;
      }
    }
    if (bad && tid == 0) *flag = 1;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int r = c; r < 4; r++) {
        T sacc = (r == c) ? T(1) : T(0);
#pragma unroll
        for (int k = c; k < r; k++) sacc -= L[r][k] * W[k][c];
        W[r][c] = sacc * inv[r];
      }
    if (tid == 0) {
      int q = 0;
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) Ld[(p >> 2) * 10 + q++] = L[i][j];
    }
    // ---- rows below: L[i][p..p+3] = A[i][p..p+3] W^T;  the four rows of X: X[p..p+3][c] = W X[p..p+3][c]
    const int m2 = NB - p - 4;
    for (int t = tid; t < m2 + NX; t += nt) {
      if (t < m2) {
        T *row = Lm + (p + 4 + t) * LS + p;
        const T a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3];
        row[0] = a0 * W[0][0];
        row[1] = a0 * W[1][0] + a1 * W[1][1];
        row[2] = a0 * W[2][0] + a1 * W[2][1] + a2 * W[2][2];
        row[3] = a0 * W[3][0] + a1 * W[3][1] + a2 * W[3][2] + a3 * W[3][3];
      } else {
        T *cp = X + p * XS + (t - m2);
        const T b0 = cp[0], b1 = cp[XS], b2 = cp[2 * XS], b3 = cp[3 * XS];
        cp[0] = W[0][0] * b0;
        cp[XS] = W[1][0] * b0 + W[1][1] * b1;
        cp[2 * XS] = W[2][0] * b0 + W[2][1] * b1 + W[2][2] * b2;
        cp[3 * XS] = W[3][0] * b0 + W[3][1] * b1 + W[3][2] * b2 + W[3][3] * b3;
      }
    }
    __syncthreads();
    // ---- rank-4 update of the trailing triangle and of the rows of X below
    for (int i = p + 4 + ty; i < NB; i += nty) {
      const T *li = Lm + i * LS + p;
      const T l0 = li[0], l1 = li[1], l2 = li[2], l3 = li[3];
      for (int cc = tx; cc < m2 + NX; cc += 64) {
        if (cc < m2) {
          const int j = p + 4 + cc;
          if (j <= i) {
            const T *lj = Lm + j * LS + p;
            Lm[i * LS + j] -= l0 * lj[0] + l1 * lj[1] + l2 * lj[2] + l3 * lj[3];
          }
        } else {
          const T *xp = X + p * XS + (cc - m2);
          X[i * XS + (cc - m2)] -= l0 * xp[0] + l1 * xp[XS] + l2 * xp[2 * XS] + l3 * xp[3 * XS];
        }
      }
    }
    __syncthreads();
  }
}
// entry (i, j), j <= i, of the factor after fat_factor_panel4
template <typename T> __device__ __forceinline__ T fat_l_entry(const T *Lm, const T *Ld, int LS, int i, int j) {
  if ((i >> 2) != (j >> 2)) return Lm[i * LS + j];
  const int a = i & 3, b = j & 3;
  return Ld[(i >> 2) * 10 + a * (a + 1) / 2 + b];
}

// ---- one elimination of the cyclic reduction for fat blocks of 52 .. 80 columns (kFatLds): D_m and [H(m,l) | H(m,r) | g] in LDS
// (155 KB at 80 columns, one workgroup per CU), the blocked Cholesky + L^-1 [H | H | g] of fat_factor_panel4, the five products.
// Round 6 (VERDICT r5 "tune the wide path"): the multiply-adds are on the matrix cores and the workgroup is 16 waves.  Until then
// (k_fat_elim: 256 threads, the panel's rank-4 updates six LDS operands per entry for four multiply-adds on one wave per SIMD, the
// products scalar loops of four LDS reads per three multiply-adds with the symmetric ones formed twice) an 80-column block took
// 245 us, 6.9 of the 17.9 ms per iteration of config 4's graph at three times its landmark density.  A rank-4 update of a 16 x 16
// tile IS one v_mfma_f64_16x16x4_f64: a tile C of the trailing triangle / of the rows of X below takes -L[i, p..p+3] L[j, p..p+3]^T
// (or ... X[p..p+3, c]), ten LDS accesses per lane per tile; the products are the tiles of k_fat_elim_rows over X where it lies; the
// 4 x 4 diagonal factor and the row scalings are fat_factor_panel4's.  92 us per block: 13.4 ms per iteration at 3 x, 5.5 at 2 x (6.0).
template <typename TR = double> __global__ void __launch_bounds__(1024) k_fat_elim_mfma(FsArgs<double, TR> a, FatLevel lv) {
  extern __shared__ __align__(16) unsigned char fat_smem[];
  const int NB = a.NB, LS = NB + 1, XS = 2 * NB + 1, NB2 = NB * NB, NX = 2 * NB + 1;
  double *Lm = reinterpret_cast<double *>(fat_smem);
  double *X = Lm + NB * LS;
  __shared__ double Ld[(kFatMax / 4) * 10];
  const int *e = lv.elim + 6 * blockIdx.x;
  const int m = e[0], r = e[2], lk_lm = e[3], lk_mr = e[4], lk_new = e[5];
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
  const int kl = lane >> 4, cl = lane & 15;
  for (int idx = tid; idx < NB2; idx += nt) {
    const int i = idx / NB, j = idx - i * NB;
    Lm[i * LS + j] = a.Dfat[(size_t)m * NB2 + idx];
    X[i * XS + j] = a.link[(size_t)lk_lm * NB2 + idx];                               // H[m, l]
    X[j * XS + NB + i] = (r >= 0) ? a.link[(size_t)lk_mr * NB2 + idx] : 0.0;           // H[m, r] = H[r, m]^T: read along H[r, m]'s rows, transposed in LDS
  }
  for (int i = tid; i < NB; i += nt) X[i * XS + 2 * NB] = a.gfat[(size_t)m * NB + i];
  __syncthreads();
  const int CTX = (NX + 15) / 16;
  for (int p = 0; p < NB; p += 4) {
    // ---- 4 x 4 diagonal block: A = L L^T, W = L^-1 (both lower), every thread (fat_factor_panel4's arithmetic)
    double A[4][4], L[4][4], W[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) A[i][j] = Lm[(p + i) * LS + p + j];
    bool bad = false;
    double inv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double dd = A[j][j];
#pragma unroll
      for (int k = 0; k < j; k++) dd -= L[j][k] * L[j][k];
      if (!(dd > 0.0)) { bad = true; dd = 1.0; }
      double y = fs_rsqrt(dd), l = dd * y;
      l = fma(0.5 * y, fma(-l, l, dd), l);
      y = fma(y, fma(-l, y, 1.0), y);
      L[j][j] = l; inv[j] = y;
#pragma unroll
      for (int i = j + 1; i < 4; i++) {
        double v = A[i][j];
#pragma unroll
        for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
        L[i][j] = v * y;
      }
    }
    if (bad && tid == 0) *a.flag = 1;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int rr = c; rr < 4; rr++) {
        double sacc = (rr == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = c; k < rr; k++) sacc -= L[rr][k] * W[k][c];
        W[rr][c] = sacc * inv[rr];
      }
    if (tid == 0) {
      int q = 0;
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) Ld[(p >> 2) * 10 + q++] = L[i][j];
    }
    // ---- rows below: L[i][p..p+3] = A[i][p..p+3] W^T;  the four rows of X: X[p..p+3][c] = W X[p..p+3][c]
    const int r0 = p + 4, m2 = NB - r0;
    for (int t = tid; t < m2 + NX; t += nt) {
      if (t < m2) {
        double *row = Lm + (r0 + t) * LS + p;
        const double a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3];
        row[0] = a0 * W[0][0];
        row[1] = a0 * W[1][0] + a1 * W[1][1];
        row[2] = a0 * W[2][0] + a1 * W[2][1] + a2 * W[2][2];
        row[3] = a0 * W[3][0] + a1 * W[3][1] + a2 * W[3][2] + a3 * W[3][3];
      } else {
        double *cp = X + p * XS + (t - m2);
        const double b0 = cp[0], b1 = cp[XS], b2 = cp[2 * XS], b3 = cp[3 * XS];
        cp[0] = W[0][0] * b0;
        cp[XS] = W[1][0] * b0 + W[1][1] * b1;
        cp[2 * XS] = W[2][0] * b0 + W[2][1] * b1 + W[2][2] * b2;
        cp[3 * XS] = W[3][0] * b0 + W[3][1] * b1 + W[3][2] * b2 + W[3][3] * b3;
      }
    }
    __syncthreads();
    // ---- rank-4 update, one MFMA per 16 x 16 tile: row tile rt (rows r0 + 16 rt ..) x [triangle tiles 0 .. rt | the CTX tiles of X]
    const int nrt = (m2 + 15) / 16;
    const int ntile = nrt * CTX + nrt * (nrt + 1) / 2;
    for (int t = wv; t < ntile; t += nw) {
      int rt, ct;
      bool tri;
      if (t < nrt * CTX) { rt = t / CTX; ct = t - rt * CTX; tri = false; }
      else {
        const int u = t - nrt * CTX;
        rt = 0;
        while ((rt + 1) * (rt + 2) / 2 <= u) rt++;
        ct = u - rt * (rt + 1) / 2; tri = true;
      }
      const int ia = r0 + 16 * rt + cl;                               // A operand: row ia, k = kl
      const double av = (ia < NB) ? -Lm[ia * LS + p + kl] : 0.0;
      double bv;
      if (tri) { const int jb = r0 + 16 * ct + cl; bv = (jb < NB) ? Lm[jb * LS + p + kl] : 0.0; }
      else { const int cb = 16 * ct + cl; bv = (cb < NX) ? X[(p + kl) * XS + cb] : 0.0; }
      double *cbase = tri ? (Lm + r0 + 16 * ct + cl) : (X + 16 * ct + cl);
      const int cstride = tri ? LS : XS;
      const bool colok = tri ? (r0 + 16 * ct + cl < NB) : (16 * ct + cl < NX);
      fs_d4 acc;
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int ic = r0 + 16 * rt + kl + 4 * rg;
        acc[rg] = (colok && ic < NB) ? cbase[ic * cstride] : 0.0;
      }
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int ic = r0 + 16 * rt + kl + 4 * rg;
        if (colok && ic < NB) cbase[ic * cstride] = acc[rg];
      }
    }
    __syncthreads();
  }
  // ---- results: L, P, Q, z (whole lines by consecutive threads)
  for (int idx = tid; idx < NB2; idx += nt) {
    const int i = idx / NB, j = idx - i * NB;
    a.Dfat[(size_t)m * NB2 + idx] = (j <= i) ? fat_l_entry(Lm, Ld, LS, i, j) : 0.0;
    a.link[(size_t)lk_lm * NB2 + idx] = X[i * XS + j];           // P
    a.Qbuf[(size_t)m * NB2 + idx] = X[i * XS + NB + j];          // Q
  }
  for (int i = tid; i < NB; i += nt) a.gfat[(size_t)m * NB + i] = X[i * XS + 2 * NB];   // z
  // ---- products: tile (ti, tj), tj <= ti, of X^T X (rows / columns = columns of X: P | Q | z), k over the NB rows
  const int ntiles = CTX * (CTX + 1) / 2;
  for (int t = wv; t < ntiles; t += nw) {
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
    const int tj = t - ti * (ti + 1) / 2;
    const int ca = ti * 16 + cl, cb = tj * 16 + cl;
    const bool oka = ca < NX, okb = cb < NX;
    const double *pa = X + kl * XS + min(ca, NX - 1), *pb = X + kl * XS + min(cb, NX - 1);
    fs_d4 acc = fs_d4{0.0, 0.0, 0.0, 0.0};
    for (int k4 = 0; k4 < NB; k4 += 4) {
      const double av = oka ? pa[k4 * XS] : 0.0, bv = okb ? pb[k4 * XS] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const int ci = ti * 16 + kl + 4 * rg, cj = tj * 16 + cl;      // entry (ci, cj) of X^T X
      const double v = acc[rg];
      if (ci >= NX || cj > ci) continue;                            // padding; the upper half of a diagonal tile
      if (ci < NB) {                                                // P^T P (cj < NB as well)
        a.S1[(size_t)m * NB2 + (size_t)ci * NB + cj] = v;
        a.S1[(size_t)m * NB2 + (size_t)cj * NB + ci] = v;
      } else if (ci < 2 * NB) {
        if (cj < NB) { if (r >= 0) a.link[(size_t)lk_new * NB2 + (size_t)(ci - NB) * NB + cj] = -v; }   // H[r, l] = -(Q^T P)
        else {                                                      // Q^T Q
          a.S2[(size_t)m * NB2 + (size_t)(ci - NB) * NB + (cj - NB)] = v;
          a.S2[(size_t)m * NB2 + (size_t)(cj - NB) * NB + (ci - NB)] = v;
        }
      } else if (cj < 2 * NB) {                                     // ci == 2 NB: P^T z | Q^T z
        a.sv[(size_t)m * 2 * NB + cj] = v;
      }
    }
  }
}

// ---- fat blocks wider than kFatLds (84 .. 128 columns; round 5).  The three NB x NB operands no longer fit the LDS together
// (NB = 128: 395 KB); the block D_m does (132 KB).  It is factored there once, then the columns of [H(m,l) | H(m,r) | g] pass through
// the rest of the LDS in panels of PW columns (fat_wide_panel): X <- L^-1 X by the same four-pivot steps (the 4 x 4 diagonal factors
// come back from Ld), P / Q / z go out to the places k_fat_elim_mfma writes them, and the five products are formed from P and Q where
// they lie (L2).  Round 6: on the matrix cores, 16 waves -- k_fat_elim_mfma's tiles in all three phases: the factorisation's rank-4
// updates of the trailing triangle, the panels' rank-4 updates of the rows below, the products as tiles of
// [P | Q | z]^T [P | Q | z].  The scalar kernel (k_fat_elim_wide: 256 threads, 4 x 4 register tiles) took 1.19 ms per level launch at
// 104 columns, 15.5 of the 33 ms per iteration of config 4's graph at four times its landmark density; this one 0.56 (25 ms).
template <typename TR = double> __global__ void __launch_bounds__(1024) k_fat_elim_wide_mfma(FsArgs<double, TR> a, FatLevel lv, int PW) {
  extern __shared__ __align__(16) unsigned char fat_smem[];
  const int NB = a.NB, LS = NB + 1, XS = PW, NB2 = NB * NB, NXT = 2 * NB + 1;
  double *Lm = reinterpret_cast<double *>(fat_smem);
  double *X = Lm + NB * LS;
  __shared__ double Ld[(kFatMax / 4) * 10];
  const int *e = lv.elim + 6 * blockIdx.x;
  const int m = e[0], r = e[2], lk_lm = e[3], lk_mr = e[4], lk_new = e[5];
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
  const int kl = lane >> 4, cl = lane & 15;
  for (int idx = tid; idx < NB2; idx += nt) Lm[(idx / NB) * LS + idx % NB] = a.Dfat[(size_t)m * NB2 + idx];
  __syncthreads();
  // ---- D_m = L L^T (fat_factor_panel4's 4 x 4 diagonal factors and row scalings, the trailing triangle by MFMA tiles)
  for (int p = 0; p < NB; p += 4) {
    double A[4][4], L[4][4], W[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) A[i][j] = Lm[(p + i) * LS + p + j];
    bool bad = false;
    double inv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double dd = A[j][j];
#pragma unroll
      for (int k = 0; k < j; k++) dd -= L[j][k] * L[j][k];
      if (!(dd > 0.0)) { bad = true; dd = 1.0; }
      double y = fs_rsqrt(dd), l = dd * y;
      l = fma(0.5 * y, fma(-l, l, dd), l);
      y = fma(y, fma(-l, y, 1.0), y);
      L[j][j] = l; inv[j] = y;
#pragma unroll
      for (int i = j + 1; i < 4; i++) {
        double v = A[i][j];
#pragma unroll
        for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
        L[i][j] = v * y;
      }
    }
    if (bad && tid == 0) *a.flag = 1;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int rr = c; rr < 4; rr++) {
        double sacc = (rr == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = c; k < rr; k++) sacc -= L[rr][k] * W[k][c];
        W[rr][c] = sacc * inv[rr];
      }
    if (tid == 0) {
      int q = 0;
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) Ld[(p >> 2) * 10 + q++] = L[i][j];
    }
    const int r0 = p + 4, m2 = NB - r0;
    for (int t = tid; t < m2; t += nt) {
      double *row = Lm + (r0 + t) * LS + p;
      const double a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3];
      row[0] = a0 * W[0][0];
      row[1] = a0 * W[1][0] + a1 * W[1][1];
      row[2] = a0 * W[2][0] + a1 * W[2][1] + a2 * W[2][2];
      row[3] = a0 * W[3][0] + a1 * W[3][1] + a2 * W[3][2] + a3 * W[3][3];
    }
    __syncthreads();
    const int nrt = (m2 + 15) / 16, ntile = nrt * (nrt + 1) / 2;
    for (int t = wv; t < ntile; t += nw) {
      int rt = 0;
      while ((rt + 1) * (rt + 2) / 2 <= t) rt++;
      const int ct = t - rt * (rt + 1) / 2;
      const int ia = r0 + 16 * rt + cl, jb = r0 + 16 * ct + cl;
      const double av = (ia < NB) ? -Lm[ia * LS + p + kl] : 0.0, bv = (jb < NB) ? Lm[jb * LS + p + kl] : 0.0;
      double *cbase = Lm + jb;
      const bool colok = jb < NB;
      fs_d4 acc;
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int ic = r0 + 16 * rt + kl + 4 * rg;
        acc[rg] = (colok && ic < NB) ? cbase[ic * LS] : 0.0;
      }
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int ic = r0 + 16 * rt + kl + 4 * rg;
        if (colok && ic < NB) cbase[ic * LS] = acc[rg];
      }
    }
    __syncthreads();
  }
  for (int idx = tid; idx < NB2; idx += nt) {
    const int i = idx / NB, j = idx - i * NB;
    a.Dfat[(size_t)m * NB2 + idx] = (j <= i) ? fat_l_entry(Lm, Ld, LS, i, j) : 0.0;
  }
  // ---- [P | Q | z] = L^-1 [H(m,l) | H(m,r) | g], PW columns at a time
  double *Pg = a.link + (size_t)lk_lm * NB2, *Qg = a.Qbuf + (size_t)m * NB2, *zg = a.gfat + (size_t)m * NB;
  for (int c0 = 0; c0 < NXT; c0 += PW) {
    const int pw = min(PW, NXT - c0);
    __syncthreads();
    for (int idx = tid; idx < NB * pw; idx += nt) {
      const int i = idx / pw, cc = idx - i * pw, c = c0 + cc;
      double v;
      if (c < NB) v = Pg[(size_t)i * NB + c];                                                             // H[m, l]
      else if (c < 2 * NB) v = (r >= 0) ? a.link[(size_t)lk_mr * NB2 + (size_t)(c - NB) * NB + i] : 0.0;  // H[m, r] = H[r, m]^T
      else v = zg[i];
      X[i * XS + cc] = v;
    }
    __syncthreads();
    const int CTP = (pw + 15) / 16;
    for (int p = 0; p < NB; p += 4) {
      double L[4][4], W[4][4], inv[4];
      {
        int q = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j <= i; j++) L[i][j] = Ld[(p >> 2) * 10 + q++];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) inv[j] = 1.0 / L[j][j];
#pragma unroll
      for (int c = 0; c < 4; c++)
#pragma unroll
        for (int rr = c; rr < 4; rr++) {
          double sacc = (rr == c) ? 1.0 : 0.0;
#pragma unroll
          for (int k = c; k < rr; k++) sacc -= L[rr][k] * W[k][c];
          W[rr][c] = sacc * inv[rr];
        }
      for (int t = tid; t < pw; t += nt) {
        double *cp = X + p * XS + t;
        const double b0 = cp[0], b1 = cp[XS], b2 = cp[2 * XS], b3 = cp[3 * XS];
        cp[0] = W[0][0] * b0;
        cp[XS] = W[1][0] * b0 + W[1][1] * b1;
        cp[2 * XS] = W[2][0] * b0 + W[2][1] * b1 + W[2][2] * b2;
        cp[3 * XS] = W[3][0] * b0 + W[3][1] * b1 + W[3][2] * b2 + W[3][3] * b3;
      }
      __syncthreads();
      const int r0 = p + 4, m2 = NB - r0, nrt = (m2 + 15) / 16;
      for (int t = wv; t < nrt * CTP; t += nw) {
        const int rt = t / CTP, ct = t - rt * CTP;
        const int ia = r0 + 16 * rt + cl, cb = 16 * ct + cl;
        const double av = (ia < NB) ? -Lm[ia * LS + p + kl] : 0.0, bv = (cb < pw) ? X[(p + kl) * XS + cb] : 0.0;
        double *cbase = X + cb;
        const bool colok = cb < pw;
        fs_d4 acc;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int ic = r0 + 16 * rt + kl + 4 * rg;
          acc[rg] = (colok && ic < NB) ? cbase[ic * XS] : 0.0;
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int ic = r0 + 16 * rt + kl + 4 * rg;
          if (colok && ic < NB) cbase[ic * XS] = acc[rg];
        }
      }
      __syncthreads();
    }
    for (int idx = tid; idx < NB * pw; idx += nt) {
      const int i = idx / pw, cc = idx - i * pw, c = c0 + cc;
      const double v = X[i * XS + cc];
      if (c < NB) Pg[(size_t)i * NB + c] = v;                  // P
      else if (c < 2 * NB) Qg[(size_t)i * NB + (c - NB)] = v;  // Q
      else zg[i] = v;                                          // z
    }
  }
  __threadfence_block();
  __syncthreads();
  // ---- products: tile (ti, tj), tj <= ti, of [P | Q | z]^T [P | Q | z], operands from where the panels went
  const int CT = (NXT + 15) / 16, ntiles = CT * (CT + 1) / 2;
  auto col_ptr = [&](int c) -> const double * {        // column c of [P | Q | z]: base and row stride
    return c < NB ? Pg + c : (c < 2 * NB ? Qg + (c - NB) : zg);
  };
  for (int t = wv; t < ntiles; t += nw) {
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
    const int tj = t - ti * (ti + 1) / 2;
    const int ca = ti * 16 + cl, cb = tj * 16 + cl;
    const bool oka = ca < NXT, okb = cb < NXT;
    const double *pa = col_ptr(min(ca, NXT - 1)), *pb = col_ptr(min(cb, NXT - 1));
    const int sa = (min(ca, NXT - 1) < 2 * NB) ? NB : 1, sb = (min(cb, NXT - 1) < 2 * NB) ? NB : 1;
    fs_d4 acc = fs_d4{0.0, 0.0, 0.0, 0.0};
    for (int k4 = 0; k4 < NB; k4 += 4) {
      const double av = oka ? pa[(size_t)(k4 + kl) * sa] : 0.0, bv = okb ? pb[(size_t)(k4 + kl) * sb] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const int ci = ti * 16 + kl + 4 * rg, cj = tj * 16 + cl;
      const double v = acc[rg];
      if (ci >= NXT || cj > ci) continue;
      if (ci < NB) {
        a.S1[(size_t)m * NB2 + (size_t)ci * NB + cj] = v;
        a.S1[(size_t)m * NB2 + (size_t)cj * NB + ci] = v;
      } else if (ci < 2 * NB) {
        if (cj < NB) { if (r >= 0) a.link[(size_t)lk_new * NB2 + (size_t)(ci - NB) * NB + cj] = -v; }
        else {
          a.S2[(size_t)m * NB2 + (size_t)(ci - NB) * NB + (cj - NB)] = v;
          a.S2[(size_t)m * NB2 + (size_t)(cj - NB) * NB + (ci - NB)] = v;
        }
      } else if (cj < 2 * NB) {
        a.sv[(size_t)m * 2 * NB + cj] = v;
      }
    }
  }
}

// ---- round 3: the same elimination with the whole factorisation in REGISTERS (fat blocks up to 48 columns, fp64).
// k_fat_elim (the LDS kernel of rounds 2-5) took 37-40 us for ONE 36-column block whatever the level's size (ablations: load 6, the blocked
// factorisation + L^-1 [H | H | g] through LDS 22, the five products 10) and the cyclic reduction has twelve levels of it, nine
// of them smaller than the chip.  Here lane i < NBP of a wave holds ROW i of the block D_m (NBP = NB rounded up to 8, padded with
// the identity) and the remaining 64 - NBP lanes hold rows of [H(m,l) | H(m,r) | g]^T: the right-looking Cholesky's row
// operations turn the stacked matrix [D; X^T] into [L; (L^-1 X)^T] -- the triangular solves ride along as extra rows.  Column
// j's pivot row entries reach the other lanes through v_readlane (scalar operands of the multiply-adds); no LDS, no barrier.
// Every wave factors the block redundantly (they run on different SIMDs) and carries its own 64 - NBP columns of X, so the
// waves never meet before the products.  Those -- P^T P, Q^T Q, Q^T P, P^T z, Q^T z, all of them blocks of X'^T X' with
// X' = L^-1 X -- are 16 x 16 tiles on v_mfma_f64_16x16x4_f64 over the rows X'^T left in LDS.
template <int NBP, typename TR = double>
__global__ void __launch_bounds__(NBP <= 40 ? 256 : 512) k_fat_elim_rows(FsArgs<double, TR> a, FatLevel lv) {
  constexpr int NBL = 64 - NBP;                       // rows of X^T per wave
  constexpr int KS = NBP + 1;                         // LDS row stride of X'^T (odd: operand loads spread over the banks)
  constexpr int CTM = (2 * NBP + 1 + 15) / 16;        // 16-column panels of X at the largest NB this instantiation serves
  __shared__ double PT[16 * CTM * KS];
  const int NB = a.NB, NB2 = NB * NB, NX = 2 * NB + 1;
  const int *e = lv.elim + 6 * blockIdx.x;
  const int m = e[0], r = e[2], lk_lm = e[3], lk_mr = e[4], lk_new = e[5];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  const bool isA = lane < NBP;
  const int c = wv * NBL + (lane - NBP);              // column of X of a lane behind the block rows
  // ---- loads: one row per lane, element k at base[min(k, NB - 1) * stride]
  const double *base = a.Dfat + (size_t)m * NB2;      // (a valid address for the lanes that hold padding)
  int stride = 0;
  bool valid = false;
  if (isA) {
    if (lane < NB) { base = a.Dfat + (size_t)m * NB2 + (size_t)lane * NB; stride = 1; valid = true; }
  } else if (c < NB) { base = a.link + (size_t)lk_lm * NB2 + c; stride = NB; valid = true; }                       // H[m, l][k][c]
  else if (c < 2 * NB) { if (r >= 0) { base = a.link + (size_t)lk_mr * NB2 + (size_t)(c - NB) * NB; stride = 1; valid = true; } }   // H[r, m][c - NB][k]
  else if (c == 2 * NB) { base = a.gfat + (size_t)m * NB; stride = 1; valid = true; }
  double av[NBP];
#pragma unroll
  for (int k = 0; k < NBP; k++) av[k] = base[(size_t)min(k, NB - 1) * stride];
#pragma unroll
  for (int k = 0; k < NBP; k++) {
    const double pad = (isA && lane == k) ? 1.0 : 0.0;
    av[k] = (valid && k < NB) ? av[k] : pad;
  }
  // ---- [D; X^T] -> [L; X'^T]
  bool bad = false;
  static_for<0, NBP>([&](auto jj) {
    constexpr int j = decltype(jj)::value;
    double dd = lane_bcast(av[j], j);
    if (!(dd > 0.0)) { bad = true; dd = 1.0; }
    double y = fs_rsqrt(dd), l = dd * y;
    l = fma(0.5 * y, fma(-l, l, dd), l);              // one residual step each, as in fat_factor_panel4
    y = fma(y, fma(-l, y, 1.0), y);
    const double lj = (lane == j) ? l : av[j] * y;
    av[j] = lj;
    static_for<j + 1, NBP>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      av[k] = fma(-lj, lane_bcast(lj, k), av[k]);
    });
  });
  if (bad && tid == 0) *a.flag = 1;
  // ---- results: L (wave 0), P / Q / z (the lanes that own the columns), X'^T rows to LDS for the products
  if (isA) {
    if (wv == 0 && lane < NB) {
      double *dp = a.Dfat + (size_t)m * NB2 + (size_t)lane * NB;
#pragma unroll
      for (int k = 0; k < NBP; k++) if (k < NB) dp[k] = (k <= lane) ? av[k] : 0.0;
    }
  } else {
    if (c < 16 * CTM) {
#pragma unroll
      for (int k = 0; k < NBP; k++) PT[c * KS + k] = av[k];
    }
    if (c < NB) {
      double *dp = a.link + (size_t)lk_lm * NB2 + c;                                  // P
#pragma unroll
      for (int k = 0; k < NBP; k++) if (k < NB) dp[(size_t)k * NB] = av[k];
    } else if (c < 2 * NB) {
      double *dp = a.Qbuf + (size_t)m * NB2 + (c - NB);                               // Q
#pragma unroll
      for (int k = 0; k < NBP; k++) if (k < NB) dp[(size_t)k * NB] = av[k];
    } else if (c == 2 * NB) {
      double *dp = a.gfat + (size_t)m * NB;                                           // z
#pragma unroll
      for (int k = 0; k < NBP; k++) if (k < NB) dp[k] = av[k];
    }
  }
  __syncthreads();
  // ---- products: tile (ti, tj), tj <= ti, of X'^T X' (rows / columns = columns of X: P | Q | z)
  const int CT = (NX + 15) / 16, ntiles = CT * (CT + 1) / 2;
  const int kl = lane >> 4, cl = lane & 15;
  for (int p = wv; p < ntiles; p += nw) {
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= p) ti++;
    const int tj = p - ti * (ti + 1) / 2;
    const double *pa = PT + (ti * 16 + cl) * KS + kl, *pb = PT + (tj * 16 + cl) * KS + kl;
    fs_d4 acc = fs_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k4 = 0; k4 < NBP; k4 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k4], pb[k4], acc, 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const int ci = ti * 16 + kl + 4 * rg, cj = tj * 16 + cl;      // entry (ci, cj) of X'^T X'
      const double v = acc[rg];
      if (ci >= NX || cj > ci) continue;                            // padding; the upper half of a diagonal tile
      if (ci < NB) {                                                // P^T P (cj < NB as well)
        a.S1[(size_t)m * NB2 + (size_t)ci * NB + cj] = v;
        a.S1[(size_t)m * NB2 + (size_t)cj * NB + ci] = v;
      } else if (ci < 2 * NB) {
        if (cj < NB) { if (r >= 0) a.link[(size_t)lk_new * NB2 + (size_t)(ci - NB) * NB + cj] = -v; }   // H[r, l] = -(Q^T P)
        else {                                                      // Q^T Q
          a.S2[(size_t)m * NB2 + (size_t)(ci - NB) * NB + (cj - NB)] = v;
          a.S2[(size_t)m * NB2 + (size_t)(cj - NB) * NB + (ci - NB)] = v;
        }
      } else if (cj < 2 * NB) {                                     // ci == 2 NB: P^T z | Q^T z
        a.sv[(size_t)m * 2 * NB + cj] = v;
      }
    }
  }
}

template <typename T, typename TR = T> __global__ void __launch_bounds__(256) k_fat_update(FsArgs<T, TR> a, FatLevel lv) {
  const int NB = a.NB, NB2 = NB * NB;
  const int *u = lv.upd + 3 * blockIdx.x;
  const int blk = u[0], ml = u[1], mr = u[2];     // ml: eliminated block whose RIGHT neighbour this is; mr: ... LEFT ...
  for (int idx = threadIdx.x; idx < NB2; idx += blockDim.x) {
    T v = a.Dfat[(size_t)blk * NB2 + idx];
    if (ml >= 0) v -= a.S2[(size_t)ml * NB2 + idx];
    if (mr >= 0) v -= a.S1[(size_t)mr * NB2 + idx];
    a.Dfat[(size_t)blk * NB2 + idx] = v;
  }
  for (int i = threadIdx.x; i < NB; i += blockDim.x) {
    T v = a.gfat[(size_t)blk * NB + i];
    if (ml >= 0) v -= a.sv[(size_t)ml * 2 * NB + NB + i];
    if (mr >= 0) v -= a.sv[(size_t)mr * 2 * NB + i];
    a.gfat[(size_t)blk * NB + i] = v;
  }
}

// the last active block: dense Cholesky solve
template <typename T, typename TR = T> __global__ void __launch_bounds__(256) k_fat_top(FsArgs<T, TR> a, int top) {
  extern __shared__ __align__(16) unsigned char fat_smem[];
  const int NB = a.NB, LS = NB + 1;
  T *Lm = reinterpret_cast<T *>(fat_smem);
  T *y = Lm + NB * LS;
  __shared__ T Ld[(kFatMax / 4) * 10];
  for (int idx = threadIdx.x; idx < NB * NB; idx += blockDim.x) Lm[(idx / NB) * LS + idx % NB] = a.Dfat[(size_t)top * NB * NB + idx];
  for (int i = threadIdx.x; i < NB; i += blockDim.x) y[i] = a.gfat[(size_t)top * NB + i];
  __syncthreads();
  fat_factor_panel4(Lm, y, Ld, NB, LS, 1, 1, a.flag);          // y <- L^-1 g
  if (threadIdx.x < 64) {                                        // x = L^-T y, right-looking, one wave
    const int lane = threadIdx.x;
    for (int i = NB - 1; i >= 0; i--) {
      const T xi = y[i] / fat_l_entry(Lm, Ld, LS, i, i);
      if (lane == 0) y[i] = xi;
      for (int k = lane; k < i; k += 64) y[k] -= fat_l_entry(Lm, Ld, LS, i, k) * xi;
      fs_wave_sync();
    }
    for (int i = lane; i < NB; i += 64) a.xfat[(size_t)top * NB + i] = y[i];
  }
}

// ---- x_m = L^-T (z - P x_l - Q x_r) for fat blocks of 52 .. 128 columns (the factor through LDS: NB (NB + 1) + 3 NB values, 135 KB
// at NB = 128).  Round 6: four waves.  Until then (k_fat_back) ONE wave did it: a lane walked a row of P and of Q (160 dependent
// loads of its own line each) and the triangular solve was NB steps of two wave-level syncs: 77 us per level at 80 columns,
// 13 levels = 1.0 ms of the 5.5 ms iteration of config 4's graph at twice its landmark density.  Now the factor arrives in whole
// lines, a wave forms a row's two dot products with its lanes along the row, and the triangular solve takes four rows per step
// (the 4 x 4 diagonal block solved by every lane from ten LDS broadcasts, one rank-4 update of the rest).
template <typename TR = double> __global__ void __launch_bounds__(256) k_fat_back_w4(FsArgs<double, TR> a, FatLevel lv) {
  extern __shared__ __align__(16) unsigned char fat_smem[];
  const int NB = a.NB, LS = NB + 1, NB2 = NB * NB, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double *Ls = reinterpret_cast<double *>(fat_smem), *xs = Ls + NB * LS, *ts = xs + 2 * NB;
  const int *e = lv.elim + 6 * blockIdx.x;
  const int m = e[0], l = e[1], r = e[2], lk_lm = e[3];
  for (int idx = tid; idx < NB2; idx += 256) Ls[(idx / NB) * LS + idx % NB] = a.Dfat[(size_t)m * NB2 + idx];
  for (int i = tid; i < NB; i += 256) {
    xs[i] = a.xfat[(size_t)l * NB + i];
    xs[NB + i] = (r >= 0) ? a.xfat[(size_t)r * NB + i] : 0.0;
  }
  __syncthreads();
  const double *P = a.link + (size_t)lk_lm * NB2, *Q = a.Qbuf + (size_t)m * NB2;
  for (int i = wv; i < NB; i += 4) {                  // t_i = z_i - P[i, :] x_l - Q[i, :] x_r
    double sacc = 0.0;
    for (int j = lane; j < NB; j += 64) {
      sacc = fma(P[(size_t)i * NB + j], xs[j], sacc);
      if (r >= 0) sacc = fma(Q[(size_t)i * NB + j], xs[NB + j], sacc);
    }
    for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
    if (lane == 0) ts[i] = a.gfat[(size_t)m * NB + i] - sacc;
  }
  __syncthreads();
  if (wv == 0) {                                      // L^T x = t from the last four rows up (NB is a multiple of 4)
    for (int p = NB - 4; p >= 0; p -= 4) {
      const double l00 = Ls[p * LS + p], l10 = Ls[(p + 1) * LS + p], l11 = Ls[(p + 1) * LS + p + 1];
      const double l20 = Ls[(p + 2) * LS + p], l21 = Ls[(p + 2) * LS + p + 1], l22 = Ls[(p + 2) * LS + p + 2];
      const double l30 = Ls[(p + 3) * LS + p], l31 = Ls[(p + 3) * LS + p + 1], l32 = Ls[(p + 3) * LS + p + 2], l33 = Ls[(p + 3) * LS + p + 3];
      const double y3 = ts[p + 3] / l33;
      const double y2 = (ts[p + 2] - l32 * y3) / l22;
      const double y1 = (ts[p + 1] - l21 * y2 - l31 * y3) / l11;
      const double y0 = (ts[p] - l10 * y1 - l20 * y2 - l30 * y3) / l00;
      fs_wave_sync();                                 // (every lane has read t[p .. p + 3])
      for (int k = lane; k < p; k += 64)
        ts[k] -= Ls[p * LS + k] * y0 + Ls[(p + 1) * LS + k] * y1 + Ls[(p + 2) * LS + k] * y2 + Ls[(p + 3) * LS + k] * y3;
      if (lane == 0) { ts[p] = y0; ts[p + 1] = y1; ts[p + 2] = y2; ts[p + 3] = y3; }
      fs_wave_sync();
    }
    for (int i = lane; i < NB; i += 64) a.xfat[(size_t)m * NB + i] = ts[i];
  }
}

// ---- round 3: the same back-substitution without LDS (fat blocks up to 64 columns, fp64).  Lane k holds COLUMN k of L
// (L[i][k] for all i: coalesced loads) and entry k of t = z - P x_l - Q x_r; step i broadcasts x_i = t_i / L_ii from lane i
// (v_readlane) and every lane k < i takes L[i][k] x_i off its entry.  The one-wave LDS kernel of round 2 spent 12-13 us per block (one wave,
// 2 NB wave-level syncs around LDS round trips); this is one dependent multiply-add + broadcast per step.
template <int NBP, typename TR = double> __global__ void __launch_bounds__(64) k_fat_back_rows(FsArgs<double, TR> a, FatLevel lv) {
  const int NB = a.NB, NB2 = NB * NB, lane = threadIdx.x;
  const int *e = lv.elim + 6 * blockIdx.x;
  const int m = e[0], l = e[1], r = e[2], lk_lm = e[3];
  const int kc = min(lane, NB - 1);
  const bool on = lane < NB;
  double Lc[NBP];                                   // column `lane` of L
  const double *dp = a.Dfat + (size_t)m * NB2 + kc;
#pragma unroll
  for (int i = 0; i < NBP; i++) Lc[i] = dp[(size_t)min(i, NB - 1) * NB];
  // t = z - P x_l - Q x_r: row `lane` of P and Q against the neighbours' solutions, broadcast entry by entry
  const double xl = a.xfat[(size_t)l * NB + kc], xr = (r >= 0) ? a.xfat[(size_t)r * NB + kc] : 0.0;
  double t = a.gfat[(size_t)m * NB + kc];
  const double *P = a.link + (size_t)lk_lm * NB2 + (size_t)kc * NB, *Q = a.Qbuf + (size_t)m * NB2 + (size_t)kc * NB;
  static_for<0, NBP / 8>([&](auto cc) {             // eight columns at a time (all of P and Q at once would not fit the registers)
    constexpr int j0 = 8 * decltype(cc)::value;
    double pr[8], qr[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { pr[j] = P[min(j0 + j, NB - 1)]; qr[j] = Q[min(j0 + j, NB - 1)]; }
    static_for<0, 8>([&](auto jj) {
      constexpr int j = j0 + decltype(jj)::value;
      if (j < NB) {                                 // (uniform)
        t = fma(-pr[j - j0], lane_bcast(xl, j), t);
        t = fma(-qr[j - j0], lane_bcast(xr, j), t);
      }
    });
  });
  double dinv = 1.0;
  static_for<0, NBP>([&](auto ii) {                 // lane i keeps 1 / L_ii
    constexpr int i = decltype(ii)::value;
    if (lane == i) dinv = 1.0 / Lc[i];
  });
  static_for<0, NBP>([&](auto ii) {
    constexpr int i = NBP - 1 - decltype(ii)::value;
    if (i < NB) {                                   // (uniform)
      const double xi = lane_bcast(t * dinv, i);
      t = (lane == i) ? xi : ((lane < i) ? fma(-Lc[i], xi, t) : t);
    }
  });
  if (on) a.xfat[(size_t)m * NB + lane] = t;
}

// ---- a chain split across GPUs (every piece runs from one shared cut state to the next; the shared cut and the landmarks
// seen from both sides form a fat separator both neighbours hold).  The cyclic reduction of a piece stops at its two end
// blocks; what is left of it is the interface record  [Dff | H(last, first) | Dll | g_first | g_last]  in blocks of NT
// (the widest fat block of any piece; unit diagonal beyond the piece's own NB).
template <typename T, typename TR = T> __global__ void __launch_bounds__(256) k_fs_pack_record(FsArgs<T, TR> a, int end_link, int NT, T *rec) {
  const int NB = a.NB, NB2 = NB * NB, NT2 = NT * NT, last = a.K - 1;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < NT2; idx += gridDim.x * blockDim.x) {
    const int i = idx / NT, j = idx - i * NT;
    const bool in = i < NB && j < NB;
    const T pad = (i == j) ? T(1) : T(0);
    rec[idx] = in ? a.Dfat[i * NB + j] : pad;
    rec[NT2 + idx] = in ? a.link[(size_t)end_link * NB2 + i * NB + j] : T(0);
    rec[2 * NT2 + idx] = in ? a.Dfat[(size_t)last * NB2 + i * NB + j] : pad;
    if (idx < NT) {
      rec[3 * NT2 + idx] = idx < NB ? a.gfat[idx] : T(0);
      rec[3 * NT2 + NT + idx] = idx < NB ? a.gfat[(size_t)last * NB + idx] : T(0);
    }
  }
}
// the P gathered records -> the (P + 1)-block tridiagonal system of the shared separators (block j sits between piece j - 1
// and piece j; blocks 0 and P are the two ends of the whole chain)
template <typename T> __global__ void __launch_bounds__(256) k_fs_top_build(const T *recs, int P, int NT, T *D, T *link, T *g) {
  const int NT2 = NT * NT, j = blockIdx.x;
  const size_t RS = (size_t)3 * NT2 + 2 * NT;
  const T *left = (j > 0) ? recs + (size_t)(j - 1) * RS : nullptr, *right = (j < P) ? recs + (size_t)j * RS : nullptr;
  for (int idx = threadIdx.x; idx < NT2; idx += blockDim.x) {
    T v = T(0);
    if (left) v += left[2 * NT2 + idx];
    if (right) v += right[idx];
    D[(size_t)j * NT2 + idx] = v;
    if (right) link[(size_t)j * NT2 + idx] = right[NT2 + idx];
  }
  for (int i = threadIdx.x; i < NT; i += blockDim.x) {
    T v = T(0);
    if (left) v += left[3 * NT2 + NT + i];
    if (right) v += right[3 * NT2 + i];
    g[(size_t)j * NT + i] = v;
  }
}
// the solution of this piece's two end blocks out of the top system
template <typename T, typename TR = T> __global__ void __launch_bounds__(64) k_fs_top_take(FsArgs<T, TR> a, const T *xtop, int rank, int NT) {
  for (int i = threadIdx.x; i < a.NB; i += blockDim.x) {
    a.xfat[i] = xtop[(size_t)rank * NT + i];
    a.xfat[(size_t)(a.K - 1) * a.NB + i] = xtop[(size_t)(rank + 1) * NT + i];
  }
}

// out[i] = x[i] for the landmarks this piece counts (own[landmark] != 0), 0 for those shared with the piece on its right
template <typename T> __global__ void __launch_bounds__(256) k_fs_mask_mul(const T *x, const int *own, int ld, int n, T *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = own[i / ld] ? x[i] : T(0);
}

// ---- scatter the fat solution: cut states -> x, landmarks -> dL
template <typename T, typename TR = T> __global__ void __launch_bounds__(256) k_fs_scatter(FsArgs<T, TR> a) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid < a.K * a.B) {
    const int k = tid / a.B, r = tid - k * a.B;
    a.x[(size_t)a.cuts[k] * a.B + r] = a.xfat[(size_t)k * a.NB + r];
  }
  if (tid < a.L * a.ld) {
    const int l = tid / a.ld, q = tid - l * a.ld;
    a.dL[tid] = a.xfat[(size_t)a.lm_fat[l] * a.NB + a.B + a.lm_slot[l] * a.ld + q];
  }
}

// ---- right-hand side of the interior states once the fat solution is known: rhs_s = g_s - G_s x_fat
template <typename T, int B, typename TR = T> __global__ void __launch_bounds__(128) k_fs_rhs(FsArgs<T, TR> a) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.N || a.segid[s] < 0) return;
  T r[B];
  const T *bp = a.blk + (size_t)s * a.BS;
#pragma unroll
  for (int k = 0; k < B; k++) r[k] = bp[2 * B * B + k];
  if (s > 0 && a.segid[s - 1] < 0) {            // left neighbour is a cut: H[s, s-1] = O_{s-1}
    const T *op = a.blk + (size_t)(s - 1) * a.BS + B * B;
    const T *xc = a.x + (size_t)(s - 1) * B;
#pragma unroll
    for (int k = 0; k < B; k++)
#pragma unroll
      for (int j = 0; j < B; j++) r[k] -= op[k * B + j] * xc[j];
  }
  if (s + 1 < a.N && a.segid[s + 1] < 0) {      // right neighbour is a cut: H[s, s+1] = O_s^T
    const T *op = bp + B * B;
    const T *xc = a.x + (size_t)(s + 1) * B;
#pragma unroll
    for (int k = 0; k < B; k++)
#pragma unroll
      for (int j = 0; j < B; j++) r[k] -= op[j * B + k] * xc[j];
  }
  for (int half = 0; half < 2; half++) {        // rows of state s (left halves), rows of state s - 1 (right halves)
    const int st = s - half;
    if (st < 0) continue;
    for (int rho = a.rowptr[st]; rho < a.rowptr[st + 1]; rho++) {
      const int lm = a.rowLm[rho];
      if (lm < 0) continue;
      T t = T(0);
      for (int q = 0; q < a.ld; q++) t += a.rowM[(size_t)rho * a.ld + q] * a.dL[(size_t)lm * a.ld + q];
      const TR *row = a.rowLR + (size_t)rho * 2 * B + half * B;
#pragma unroll
      for (int k = 0; k < B; k++) r[k] -= row[k] * t;
    }
  }
#pragma unroll
  for (int k = 0; k < B; k++) a.rhs[(size_t)s * B + k] = r[k];
}

// ---- interior states: single right-hand side forward / backward sweep with the stored factors.  One WAVE per segment:
// lane r < B owns row r of the 6-vector recurrences (y_s = W_s (rhs_s - E_{s-1}^T y_{s-1}) forward, x_s = W_s^T (y_s - E_s x_{s+1})
// backward); the other lanes' values reach it through v_readlane.  All 64 lanes fetch: the [W | E] records and right-hand
// sides of FL states at a time, coalesced, into LDS, one burst ahead; the results leave in bursts as well (loads and stores
// share the in-order vmcnt counter: a step that loads and stores drains both).  The first version ran one THREAD per segment
// (62 waves on the whole chip, every lane walking its own 576-byte records): 1.0 ms per iteration at 1e6 states.
__device__ __forceinline__ double fs_readlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float fs_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <typename T, int B, typename TR = T> __global__ void __launch_bounds__(64) k_fs_solve1(FsArgs<T, TR> a) {
  const int seg = blockIdx.x, lane = threadIdx.x;
  const int j0 = a.cuts[seg] + 1, n = a.cuts[seg + 1] - a.cuts[seg] - 1;
  if (n <= 0) return;
  constexpr int FL = 8, FW = 2 * B * B, SL = FW + B;       // slot: [W | E | rhs]
  constexpr int PF = (FL * FW + 63) / 64;
  __shared__ T buf[2][FL * SL];
  __shared__ T outb[FL * B];
  const int nb = (n + FL - 1) / FL;
  const int r = lane < B ? lane : 0;                       // lanes >= B compute row 0 again (harmless, keeps the code branch-free)
  constexpr int PR = (FL * B + 63) / 64;                   // right-hand-side values per lane and burst (2 for B = 12)
  T pre[PF], prer[PR];
  // fetch burst `bi` of a sweep over the states j0 + lo .. j0 + lo + FL - 1 (clamped to the segment)
  auto fetch = [&](int lo) {
#pragma unroll
    for (int u = 0; u < PF; u++) {
      const int k = lane + 64 * u, f = k / FW;
      const int st = min(max(lo + f, 0), n - 1);
      pre[u] = (k < FL * FW) ? a.fac[(size_t)(j0 + st) * FW + (k - f * FW)] : T(0);
    }
#pragma unroll
    for (int u = 0; u < PR; u++) {
      const int k = lane + 64 * u, f = k / B, st = min(max(lo + f, 0), n - 1);
      prer[u] = (k < FL * B) ? a.rhs[(size_t)(j0 + st) * B + (k - f * B)] : T(0);
    }
  };
  auto commit = [&](int w) {
#pragma unroll
    for (int u = 0; u < PF; u++) {
      const int k = lane + 64 * u, f = k / FW;
      if (k < FL * FW) buf[w][f * SL + (k - f * FW)] = pre[u];
    }
#pragma unroll
    for (int u = 0; u < PR; u++) {
      const int k = lane + 64 * u, f = k / B;
      if (k < FL * B) buf[w][f * SL + FW + (k - f * B)] = prer[u];
    }
  };
  // ---- forward
  T y = T(0), ep[B];                                       // ep[k] = E_{s-1}[k][r]
#pragma unroll
  for (int k = 0; k < B; k++) ep[k] = T(0);
  fetch(0);
  commit(0);
  fs_wave_sync();
  for (int bi = 0; bi < nb; bi++) {
    const int base = bi * FL, cnt = min(FL, n - base);
    fetch(base + FL);
    const T *cb = buf[bi & 1];
#pragma unroll
    for (int f = 0; f < FL; f++) {
      if (f >= cnt) break;
      const T *sp = cb + f * SL;
      T t = sp[FW + r];
#pragma unroll
      for (int k = 0; k < B; k++) t -= ep[k] * fs_readlane(y, k);
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < B; k++) acc += sp[r * B + k] * fs_readlane(t, k);     // W is stored with its zeros above the diagonal
      y = acc;
#pragma unroll
      for (int k = 0; k < B; k++) ep[k] = sp[B * B + k * B + r];
      if (lane < B) outb[f * B + lane] = y;
    }
    fs_wave_sync();
    for (int k = lane; k < cnt * B; k += 64) a.rhs[(size_t)(j0 + base) * B + k] = outb[k];   // y~ of the burst, read back by the backward sweep
    commit((bi + 1) & 1);
    fs_wave_sync();
  }
  // ---- backward (bursts and states in descending order; E of the last interior state multiplies x = 0)
  T x = T(0);
  const int last_lo = (nb - 1) * FL;
  fetch(last_lo);
  commit(0);
  fs_wave_sync();
  for (int bi = nb - 1, it = 0; bi >= 0; bi--, it++) {
    const int base = bi * FL, cnt = min(FL, n - base);
    fetch(base - FL);
    const T *cb = buf[it & 1];
#pragma unroll
    for (int f = FL - 1; f >= 0; f--) {
      if (f >= cnt) continue;
      const T *sp = cb + f * SL;
      T t = sp[FW + r];
      if (base + f < n - 1) {
#pragma unroll
        for (int k = 0; k < B; k++) t -= sp[B * B + r * B + k] * fs_readlane(x, k);
      }
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < B; k++) acc += sp[k * B + r] * fs_readlane(t, k);     // x = W^T t
      x = acc;
      if (lane < B) outb[f * B + lane] = x;
    }
    fs_wave_sync();
    for (int k = lane; k < cnt * B; k += 64) a.x[(size_t)(j0 + base) * B + k] = outb[k];
    commit((it + 1) & 1);
    fs_wave_sync();
  }
}

// landmarks += dL; *out_max = max(*out_max, |dL|_inf) -- L can be 5e4: a grid-stride kernel + one partial per block
template <typename T> __global__ void __launch_bounds__(256) k_fs_lm_update(double *lmk, const T *dL, int nl, const int *flag, T *partial) {
  __shared__ T red[256];
  const bool bad = flag && *flag;
  T mx = T(0);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += gridDim.x * blockDim.x) {
    const T v = dL[i];
    if (!bad) lmk[i] += v;
    mx = fmax(mx, (v == v) ? fabs(v) : T(INFINITY));
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
template <typename T> __global__ void __launch_bounds__(64) k_fs_max_into(const T *partial, int n, double *out) {
  T mx = T(0);
  for (int i = threadIdx.x; i < n; i += 64) mx = fmax(mx, partial[i]);
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
  if (threadIdx.x == 0) *out = fmax(*out, (double)mx);
}
// error of the landmark priors, grid-stride
template <typename T> __global__ void __launch_bounds__(256) k_fs_lmprior_err(const double *lmk, const int *pri_lm, const double *pri_meas, const double *pri_sig,
                                                                             int npri, int ld, T *partial) {
  __shared__ T red[256];
  T err = T(0);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < npri; k += gridDim.x * blockDim.x)
    for (int q = 0; q < ld; q++) {
      const T we = T((lmk[(size_t)pri_lm[k] * ld + q] - pri_meas[(size_t)k * ld + q]) / pri_sig[(size_t)k * ld + q]);
      err += we * we;
    }
  red[threadIdx.x] = T(0.5) * err;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// =================================================================== host side

struct FatSepPlan {
  bool active = false;
  int N = 0, B = 0, ld = 0, L = 0, K = 0, NB = 0, NC = 0, NCP = 0, C = 0, nlinks = 0, top = 0;
  std::vector<int> cuts;
  struct LevelHost { int elim_off, nelim, upd_off, nupd; };
  std::vector<LevelHost> levels;
  DevBuf d_cuts, d_segid, d_fat_lm_ptr, d_fat_lm, d_lm_fat, d_lm_slot, d_lmpri_ptr, d_lmpri, d_elim, d_upd;
  DevBuf fac, Y, Aseg, Dfat, link, gfat, Qbuf, S1, S2, sv, xfat, gL, rhs, partial;
  std::string err;
  // a chain split across GPUs (gpslam_hip_fs_set_split): this handle holds piece `rank` of `nranks`; its first / last fat
  // block is shared with the neighbour piece and holds exactly the landmarks first_lm / last_lm, in that order
  bool split = false;
  int rank = 0, nranks = 1, nb_top = 0, end_link = 0, ttop = 0;
  std::vector<int> first_lm, last_lm;
  std::vector<LevelHost> tlevels;
  DevBuf send, recv, tD, tlink, tg, tQ, tS1, tS2, tsv, tx, d_telim, d_tupd, d_lm_own, lm_tmp;
  // right-hand-side groups of the landmark border columns (FsArgs::lg_*)
  DevBuf d_lg_ptr, d_lg_state, d_lg_mptr, d_lg_m, Gev;
  DevBuf d_sc_base, d_sc_ptr, d_se_pk, d_se_src;    // the same as entries per chunk (k_fs_sweep_syrk)
  DevBuf d_fa_ptr, d_fa_it, d_fb_ptr, d_fb_it, d_fc_ptr, d_fc_it;   // rows of each landmark that touch a cut state (k_fs_fat_assemble)
  DevBuf lmMM;
  bool fused_sweep = false;                         // k_fs_sweep_syrk serves this plan (else k_fs_sweep + k_fs_syrk through Y)
  int ngroups = 0;
  std::vector<int> h_lmrow, h_lmstate, h_lmptr;     // compile(): rows per landmark, sorted by left state

  void release() {
    for (DevBuf *b : {&d_cuts, &d_segid, &d_fat_lm_ptr, &d_fat_lm, &d_lm_fat, &d_lm_slot, &d_lmpri_ptr, &d_lmpri, &d_elim, &d_upd,
                      &fac, &Y, &Aseg, &Dfat, &link, &gfat, &Qbuf, &S1, &S2, &sv, &xfat, &gL, &rhs, &partial,
                      &send, &recv, &tD, &tlink, &tg, &tQ, &tS1, &tS2, &tsv, &tx, &d_telim, &d_tupd, &d_lm_own, &lm_tmp,
                      &d_lg_ptr, &d_lg_state, &d_lg_mptr, &d_lg_m, &Gev, &d_sc_base, &d_sc_ptr, &d_se_pk, &d_se_src,
                      &d_fa_ptr, &d_fa_it, &d_fb_ptr, &d_fb_it, &d_fc_ptr, &d_fc_it, &lmMM})
      b->release();
    active = false;
  }

  // Level sets of the block cyclic reduction over K blocks in chain order: every level eliminates every other active block.
  // keep_last: the last block is never eliminated (a piece of a split chain stops at its two end blocks; end_link then is
  // the link between them).  elim: 6 ints per eliminated block, upd: 3 per survivor with an eliminated neighbour (FatLevel).
  static void build_levels(int K, bool keep_last, std::vector<int> &elim, std::vector<int> &upd, std::vector<LevelHost> &levels,
                           int &top, int &nlinks, int &end_link) {
    elim.clear(); upd.clear(); levels.clear();
    std::vector<int> active(K), linkidx(K > 0 ? K - 1 : 0);
    for (int k = 0; k < K; k++) active[k] = k;
    for (int k = 0; k + 1 < K; k++) linkidx[k] = k;
    int next_link = K - 1;
    while ((int)active.size() > (keep_last ? 2 : 1)) {
      LevelHost lv;
      lv.elim_off = (int)elim.size() / 6; lv.upd_off = (int)upd.size() / 3;
      const int n = (int)active.size();
      auto eliminated = [&](int q) { return q >= 0 && q < n && (q & 1) && !(keep_last && q == n - 1); };
      std::vector<int> nact, nlink;
      for (int i = 0; i < n; i++) {
        if (eliminated(i)) continue;
        nact.push_back(active[i]);
        if (!eliminated(i - 1) && !eliminated(i + 1)) continue;
        upd.push_back(active[i]);
        upd.push_back(eliminated(i - 1) ? active[i - 1] : -1);
        upd.push_back(eliminated(i + 1) ? active[i + 1] : -1);
      }
      for (int q = 1; q < n; q += 2) {
        if (!eliminated(q)) { nlink.push_back(linkidx[q - 1]); continue; }   // the kept last block: its link stays
        const int r = (q + 1 < n) ? active[q + 1] : -1;
        const int lk_new = (r >= 0) ? next_link++ : -1;
        elim.push_back(active[q]); elim.push_back(active[q - 1]); elim.push_back(r);
        elim.push_back(linkidx[q - 1]); elim.push_back(r >= 0 ? linkidx[q] : -1); elim.push_back(lk_new);
        if (r >= 0) nlink.push_back(lk_new);
      }
      lv.nelim = (int)elim.size() / 6 - lv.elim_off; lv.nupd = (int)upd.size() / 3 - lv.upd_off;
      levels.push_back(lv);
      active.swap(nact);
      linkidx.swap(nlink);
    }
    top = active[0];
    nlinks = std::max(next_link, 1);
    end_link = linkidx.empty() ? 0 : linkidx[0];
  }

  // even: segments of (nearly) equal length <= C instead of a short last one -- the pieces of a split chain, whose LAST
  // segment has to hold the whole window of every landmark shared with the neighbour
  static std::vector<int> make_cuts(int N, int C, bool even = false) {
    std::vector<int> c;
    if (even) {
      const int nseg = std::max(1, (N - 1 + C - 1) / C);
      for (int i = 0; i <= nseg; i++) c.push_back((int)(((long long)i * (N - 1)) / nseg));
      return c;
    }
    for (int s = 0; s < N - 1; s += C) c.push_back(s);
    if (c.empty() || c.back() != N - 1) c.push_back(N - 1);
    if (c.size() >= 3 && c[c.size() - 1] - c[c.size() - 2] < 2) c.erase(c.end() - 2);
    return c;
  }

  // touch_lo / touch_hi: first / last state touched by the factors of each landmark (-1: none).
  // Returns false (err set) when no segment length fits.
  bool choose(int N_, int B_, int ld_, int L_, const std::vector<int> &touch_lo, const std::vector<int> &touch_hi, int c_forced,
              std::vector<int> &fat_of, std::vector<int> &slot_of, std::vector<int> &counts) {
    N = N_; B = B_; ld = ld_; L = L_;
    if (N < 2) { err = "the segmented landmark elimination needs at least two states"; return false; }
    // one segmentation: cuts every Ctry states, every landmark on one admissible cut, balanced.  Returns whether every landmark
    // found a cut; nb_out = the fat block size it needs (B + ld * the fullest cut's landmarks)
    auto attempt = [&](int Ctry, int &nb_out) -> bool {
      cuts = make_cuts(N, Ctry, split);
      K = (int)cuts.size();
      counts.assign(K, 0);
      fat_of.assign(L, 0);
      slot_of.assign(L, 0);
      bool ok = true;
      // split chains: the shared end blocks hold exactly the listed landmarks, in the listed order (both neighbours build the
      // same block); every other landmark keeps away from them
      std::vector<int> forced(L, -1);
      const bool lsh = split && rank > 0, rsh = split && rank < nranks - 1;
      if (split) {
        for (int l : first_lm) {
          if (touch_lo[l] >= 0 && K >= 2 && touch_hi[l] > cuts[1]) { ok = false; break; }
          forced[l] = 0; fat_of[l] = 0; slot_of[l] = counts[0]++;
        }
        for (int l : last_lm) {
          if (!ok) break;
          if (touch_lo[l] >= 0 && K >= 2 && touch_lo[l] < cuts[K - 2]) { ok = false; break; }
          forced[l] = K - 1; fat_of[l] = K - 1; slot_of[l] = counts[K - 1]++;
        }
      }
      // The fat block size is set by the FULLEST cut (NB = B + ld * max count; the border of every segment, the Y buffer and
      // the cubic cost of the fat chain all scale with it), so the landmarks are balanced: those with a single admissible
      // cut first, then the others to the emptier of their cuts, then single moves off the fullest cuts while that lowers
      // the maximum (round 3: 17 -> 14 landmarks per cut on the config-4 graph, NB 40 -> 36).
      std::vector<int> lo_of(L, 0), hi_of(L, -1);
      for (int l = 0; l < L && ok; l++) {
        if (forced[l] >= 0) continue;
        int lo, hi;
        if (touch_lo[l] < 0) { lo = hi = std::min(std::max(l % K, lsh ? 1 : 0), rsh ? K - 2 : K - 1); }
        else {
          const int k_lo = (int)(std::upper_bound(cuts.begin(), cuts.end(), touch_lo[l]) - cuts.begin()) - 1;
          const int k_hi = (int)(std::lower_bound(cuts.begin(), cuts.end(), touch_hi[l]) - cuts.begin());
          if (k_hi - k_lo > 2) { ok = false; break; }
          lo = std::max(k_hi - 1, 0);
          hi = std::min(k_lo + 1, K - 1);
        }
        if (lsh && lo == 0) lo = 1;
        if (rsh && hi == K - 1) hi = K - 2;
        if (lo > hi) { ok = false; break; }
        lo_of[l] = lo; hi_of[l] = hi;
      }
      if (ok) {
        for (int pass = 0; pass < 2; pass++)          // pass 0: no choice; pass 1: the emptier admissible cut
          for (int l = 0; l < L; l++) {
            if (forced[l] >= 0 || (pass == 0) != (lo_of[l] == hi_of[l])) continue;
            int best = lo_of[l];
            for (int k = lo_of[l] + 1; k <= hi_of[l]; k++) if (counts[k] < counts[best]) best = k;
            fat_of[l] = best;
            counts[best]++;
          }
        std::vector<std::vector<int>> members(K);
        for (int l = 0; l < L; l++) if (forced[l] < 0) members[fat_of[l]].push_back(l);
        for (int round = 0; round < 64; round++) {    // move a landmark off a fullest cut to a cut at least two emptier
          int mxc = 0;
          for (int k = 0; k < K; k++) mxc = std::max(mxc, counts[k]);
          bool moved = false;
          for (int k = 0; k < K; k++) {
            if (counts[k] != mxc) continue;
            for (size_t q = 0; q < members[k].size(); q++) {
              const int l = members[k][q];
              int to = -1;
              for (int k2 = lo_of[l]; k2 <= hi_of[l]; k2++) if (k2 != k && counts[k2] + 1 < counts[k] && (to < 0 || counts[k2] < counts[to])) to = k2;
              if (to < 0) continue;
              members[k].erase(members[k].begin() + (long)q);
              members[to].push_back(l);
              fat_of[l] = to; counts[k]--; counts[to]++;
              moved = true;
              break;
            }
          }
          if (!moved) break;
        }
        // slots: forced landmarks keep theirs (the shared end blocks' order is the caller's); the rest in landmark order
        std::vector<int> next(K, 0);
        for (int k = 0; k < K; k++) next[k] = 0;
        for (int l = 0; l < L; l++) if (forced[l] >= 0) next[forced[l]]++;
        for (int l = 0; l < L; l++) if (forced[l] < 0) slot_of[l] = next[fat_of[l]]++;
      }
      int mx = 0;
      for (int k = 0; k < K; k++) mx = std::max(mx, counts[k]);
      nb_out = B + ld * mx;
      return ok;
    };
    const char *too_many = "too many landmarks per cut for the fat separators (2d + landmark_dim * landmarks must be <= 128)";
    auto no_fit = [&]() {
      return split ? "no segmentation of this piece keeps its private landmarks off the shared end blocks and the shared ones inside the "
                     "end segments: the piece is too short for the landmarks' windows of visibility (use fewer, longer pieces)"
                   : "a landmark is seen from more than two segments of the chain: no local-visibility segmentation exists";
    };
    // What a segmentation costs per state: the Schur complement of a segment is (NCP / 16)(NCP / 16 + 1) / 2 MFMA tiles per four
    // rows, the border sweep NC columns (measured on the config-4 graph: 0.064 ms per tile and 0.025 ms per column and 1e6
    // states) -- both set by the FULLEST cut, so a shorter segment with one landmark less on it can drop a whole tile row.
    auto score = [&](int nb) {
      const int nbr = std::min((nb + 3) & ~3, kFatMax), nc = 2 * nbr + 1, t = ((nc + 15) & ~15) / 16;
      return 0.064 * (t * (t + 1) / 2) + 0.025 * nc;
    };
    int nb = 0;
    if (c_forced > 0) {
      const bool ok = attempt(c_forced, nb);
      if (!ok || nb > kFatMax) { err = ok ? too_many : no_fit(); return false; }
      C = c_forced;
    } else {
      // doubling finds the first segment length that fits; the lengths between it and half of it (in sixteenths of it, at least 16
      // states) are then tried as well -- round 4: the config-4 graph fits at 256 (NB 36, 15 tiles) and at 208 (NB 28, 10 tiles:
      // the solve phase 3.32 -> 2.58 ms).  Among equal costs the longest segments win (fewer fat blocks).
      int Chi = 0;
      for (int Ctry = 32;; Ctry *= 2) {
        const bool ok = attempt(Ctry, nb);
        if (ok && nb <= kFatMax) { Chi = Ctry; break; }
        if (K <= 2) { err = ok ? too_many : no_fit(); return false; }
        if (ok && nb > kFatMax) { err = too_many; return false; }   // longer segments only add landmarks per cut
      }
      int best = Chi;
      double best_score = score(nb);
      const int step = std::max(16, Chi / 16);
      for (int Ctry = Chi - step; Ctry > Chi / 2; Ctry -= step) {
        const bool ok = attempt(Ctry, nb);
        if (!ok || nb > kFatMax) continue;
        if (score(nb) < best_score - 1e-12) { best = Ctry; best_score = score(nb); }
      }
      (void)attempt(best, nb);
      C = best;
    }
    NB = (nb + 3) & ~3;
    if (NB > kFatMax) NB = kFatMax;
    NC = 2 * NB + 1;
    NCP = (NC + 15) & ~15;
    return true;
  }
};

}  // namespace gps
