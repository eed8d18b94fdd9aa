// cr_step.hpp -- one elimination of the block cyclic reduction in the row layout (shared by upper.hip and by the tail of the
// level-0 kernels in kernels.hpp).  Lane r < B of a 16-lane DPP row holds ROW r of the panel [D_j | O_j^T | F | g_j]; the
// records live in LDS as [D (B x B, row-major) | O (B x B, row-major, O_i = H[right neighbour of i, i]) | g (B)].
//
// Pair (s, j), n = the block to the right of j (possibly the virtual block beyond the group):
//   D_j x_j + O_j^T x_n + F x_s = g_j,  F = O_s
//   U = D_j^-1 O_j^T,  V = D_j^-1 F,  Y = D_j^-1 g_j        -> record j becomes [V | U | Y] (column-major, as level 0 stores it)
//   D_s -= F^T V,  g_s -= F^T Y,  O_s <- -O_j V             (s now couples to n)                      store_own()
//   D_n -= O_j U,  g_n -= O_j Y                             (n is the s of the next pair: after a barrier) add_right()
#pragma once

#include "dpp.hpp"

namespace gps {

template <int B> struct CrStep {
  static constexpr int BS = 2 * B * B + B;
  typedef double V2 __attribute__((ext_vector_type(2)));
  double Or[B], Fr[B], Ar[B], Dn[B], Fn[B];
  double gr, as_, gn;

  // rr = r for the row lanes, 0 for the idle lanes of the DPP row (they shadow row 0 and never store).  Returns true in the lane
  // of a pivot that was not positive (its reciprocal is not).  Every lane of the wave must call this (DPP): idle DPP rows pass any valid (s, j).
  __device__ __forceinline__ bool compute(const double *REC, int s, int j, int r, int rr) {
    const double *Rj = REC + j * BS, *Rs = REC + s * BS;
    double Dr[B], Gr[B], Ol[B];
    {
      const V2 *dj = reinterpret_cast<const V2 *>(Rj + rr * B), *oj = reinterpret_cast<const V2 *>(Rj + B * B + rr * B);
      const V2 *ds = reinterpret_cast<const V2 *>(Rs + rr * B), *os = reinterpret_cast<const V2 *>(Rs + B * B + rr * B);
#pragma unroll
      for (int k = 0; k < B / 2; k++) {
        const V2 a = dj[k], b = oj[k], c = ds[k], d = os[k];
        Dr[2 * k] = a.x; Dr[2 * k + 1] = a.y;        // row r of D_j
        Ol[2 * k] = b.x; Ol[2 * k + 1] = b.y;        // row r of O_j
        Ar[2 * k] = c.x; Ar[2 * k + 1] = c.y;        // row r of D_s
        Fr[2 * k] = d.x; Fr[2 * k + 1] = d.y;        // row r of F = O_s
      }
#pragma unroll
      for (int k = 0; k < B; k++) {
        Or[k] = Rj[B * B + k * B + rr];              // row r of O_j^T
        Gr[k] = Rs[B * B + k * B + rr];              // row r of F^T
      }
    }
    gr = Rj[2 * B * B + rr];
    as_ = Rs[2 * B * B + rr];
    gn = 0.0;
    __builtin_amdgcn_sched_barrier(0);
    // Gauss-Jordan on D_j by row operations; the pivot row stays unscaled until the end (scaling commutes).  The reciprocal
    // of the NEXT pivot (hardware reciprocal + two Newton steps, a chain of dependent instructions) is formed under the row
    // operations of the current one: its D entry is final as soon as the D part of the current step is done.
    double invs = 1.0;
    double piv = row_bcast<0>(Dr[0]);
    double inv = fast_rcp(piv);
    static_for<0, B>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      const bool isk = (r == k);
      invs = isk ? inv : invs;
      const double nmp = isk ? 0.0 : -(Dr[k] * inv);
      fmac_self_n<k, B>(Dr, nmp);              // (entries at or left of the pivot become garbage that nothing reads again)
      double pn = 1.0, r0 = 1.0;
      if constexpr (k + 1 < B) {
        pn = row_bcast<(k + 1 < B ? k + 1 : 0)>(Dr[k + 1 < B ? k + 1 : 0]);
        r0 = __builtin_amdgcn_rcp(pn);
      }
      __builtin_amdgcn_sched_barrier(0);
      fmac_self_n<k, B>(Or, nmp);
      if constexpr (k + 1 < B) r0 = fma(fma(-pn, r0, 1.0), r0, r0);
      __builtin_amdgcn_sched_barrier(0);
      fmac_self_n<k, B>(Fr, nmp);
      if constexpr (k + 1 < B) r0 = fma(fma(-pn, r0, 1.0), r0, r0);
      __builtin_amdgcn_sched_barrier(0);
      fmac_self1<k>(gr, nmp);
      piv = pn; inv = r0;
    });
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < B; k++) { Or[k] *= invs; Fr[k] *= invs; Dn[k] = 0.0; Fn[k] = 0.0; }   // U_j, V_j: row r
    gr *= invs;                                                                                  // Y_j
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, B>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      const double ol = Ol[i], gg = Gr[i];          // subtracted: the negation is the instructions' source modifier
      fmac_bcast_n<i, B, true>(Dn, Or, ol);         // -O_j U_j: row r
      fmac_bcast2<i, true>(gn, as_, gr, ol, gg);   // -O_j Y_j,  g_s -= F^T Y_j
      fmac_bcast_n<i, B, true>(Fn, Fr, ol);         // -O_j V_j: the coupling of s to n
      fmac_bcast_n<i, B, true>(Ar, Fr, gg);         // D_s -= F^T V_j
    });
    __builtin_amdgcn_sched_barrier(0);
    return !(invs > 0.0);     // a pivot that is not positive (or not a number) leaves such a reciprocal in its own lane
  }

  // the pair's own blocks: s in place, j as the factor record (row lanes of active pairs only)
  __device__ __forceinline__ void store_own(double *REC, int s, int j, int r) const {
    double *Ws = REC + s * BS, *Wj = REC + j * BS;
    V2 *wd = reinterpret_cast<V2 *>(Ws + r * B), *wo = reinterpret_cast<V2 *>(Ws + B * B + r * B);
#pragma unroll
    for (int k = 0; k < B / 2; k++) {
      V2 a, b;
      a.x = Ar[2 * k]; a.y = Ar[2 * k + 1];
      b.x = Fn[2 * k]; b.y = Fn[2 * k + 1];
      wd[k] = a;
      wo[k] = b;
    }
#pragma unroll
    for (int k = 0; k < B; k++) {
      Wj[k * B + r] = Fr[k];                    // the factor record is column-major [V | U | Y]
      Wj[B * B + k * B + r] = Or[k];
    }
    Ws[2 * B * B + r] = as_;
    Wj[2 * B * B + r] = gr;
  }

  // the right neighbour's share (it is the s of the next pair, whose store_own must have happened: barrier in between)
  __device__ __forceinline__ void add_right(double *REC, int n, int r) const {
    double *Wn = REC + n * BS;
    V2 *wd = reinterpret_cast<V2 *>(Wn + r * B);
#pragma unroll
    for (int k = 0; k < B / 2; k++) {
      V2 a = wd[k];
      a.x += Dn[2 * k]; a.y += Dn[2 * k + 1];
      wd[k] = a;
    }
    Wn[2 * B * B + r] += gn;
  }
};

// The same elimination on TWO adjacent DPP rows (the later sub-levels of a group leave rows idle): both rows run the
// Gauss-Jordan on their own copy of D_j (and of g_j), row `half == 0` carries O_j^T along and produces U, Y, the right
// neighbour's share (-O_j U, -O_j Y); row `half == 1` carries F = O_s and produces V, the new coupling -O_j V and the pair's
// own Schur complement (D_s - F^T V, g_s - F^T Y).  Per lane 546 multiply-adds instead of 834, every one of them with the
// operands and in the order of CrStep: bit-identical results.  (An elimination is VALU-issue bound for a single wave --
// DESIGN.md section 4 -- so the width of the panel, not its latency, is what a sub-level costs.)
template <int B> struct CrStepWide {
  static constexpr int BS = 2 * B * B + B;
  typedef double V2 __attribute__((ext_vector_type(2)));
  double X[B], P[B], Ar[B];        // half 0: X = row r of O_j^T -> U, P = -O_j U;   half 1: X = row r of F -> V, P = -O_j V, Ar = D_s - F^T V
  double gr, as_, gn;

  __device__ __forceinline__ bool compute(const double *REC, int s, int j, int r, int rr, int half) {
    const double *Rj = REC + j * BS, *Rs = REC + s * BS;
    double Dr[B], Ol[B], Gr[B];
    {
      const V2 *dj = reinterpret_cast<const V2 *>(Rj + rr * B), *oj = reinterpret_cast<const V2 *>(Rj + B * B + rr * B);
      const V2 *ds = reinterpret_cast<const V2 *>(Rs + rr * B);
#pragma unroll
      for (int k = 0; k < B / 2; k++) {
        const V2 a = dj[k], b = oj[k], c = ds[k];
        Dr[2 * k] = a.x; Dr[2 * k + 1] = a.y;        // row r of D_j
        Ol[2 * k] = b.x; Ol[2 * k + 1] = b.y;        // row r of O_j
        Ar[2 * k] = c.x; Ar[2 * k + 1] = c.y;        // row r of D_s (half 1)
      }
      // half 0: row r of O_j^T = column r of O_j (stride B);  half 1: row r of F = O_s (contiguous)
      const double *xb = half ? Rs + B * B + rr * B : Rj + B * B + rr;
      const int xs = half ? 1 : B;
#pragma unroll
      for (int k = 0; k < B; k++) {
        X[k] = xb[k * xs];
        Gr[k] = Rs[B * B + k * B + rr];              // row r of F^T (half 1)
      }
    }
    gr = Rj[2 * B * B + rr];
    as_ = Rs[2 * B * B + rr];
    gn = 0.0;
    __builtin_amdgcn_sched_barrier(0);
    double invs = 1.0;
    double piv = row_bcast<0>(Dr[0]);
    double inv = fast_rcp(piv);
    static_for<0, B>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      const bool isk = (r == k);
      invs = isk ? inv : invs;
      const double nmp = isk ? 0.0 : -(Dr[k] * inv);
      fmac_self_n<k, B>(Dr, nmp);
      double pn = 1.0, r0 = 1.0;
      if constexpr (k + 1 < B) {
        pn = row_bcast<(k + 1 < B ? k + 1 : 0)>(Dr[k + 1 < B ? k + 1 : 0]);
        r0 = __builtin_amdgcn_rcp(pn);
      }
      __builtin_amdgcn_sched_barrier(0);
      fmac_self_n<k, B>(X, nmp);
      if constexpr (k + 1 < B) { r0 = fma(fma(-pn, r0, 1.0), r0, r0); r0 = fma(fma(-pn, r0, 1.0), r0, r0); }
      __builtin_amdgcn_sched_barrier(0);
      fmac_self1<k>(gr, nmp);
      piv = pn; inv = r0;
    });
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < B; k++) { X[k] *= invs; P[k] = 0.0; }     // U_j / V_j: row r
    gr *= invs;                                                     // Y_j
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, B>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      const double ol = Ol[i], gg = half ? Gr[i] : 0.0;
      fmac_bcast_n<i, B, true>(P, X, ol);           // half 0: -O_j U_j;  half 1: -O_j V_j (the coupling of s to n)
      fmac_bcast2<i, true>(gn, as_, gr, ol, gg);   // -O_j Y_j (half 0 uses it),  g_s -= F^T Y_j (half 1)
      fmac_bcast_n<i, B, true>(Ar, X, gg);          // half 1: D_s -= F^T V_j  (half 0: += 0)
    });
    __builtin_amdgcn_sched_barrier(0);
    return !(invs > 0.0);     // a pivot that is not positive (or not a number) leaves such a reciprocal in its own lane
  }

  // row lanes of active pairs only
  __device__ __forceinline__ void store_own(double *REC, int s, int j, int r, int half) const {
    double *Ws = REC + s * BS, *Wj = REC + j * BS;
    if (half) {
      V2 *wd = reinterpret_cast<V2 *>(Ws + r * B), *wo = reinterpret_cast<V2 *>(Ws + B * B + r * B);
#pragma unroll
      for (int k = 0; k < B / 2; k++) {
        V2 a, b;
        a.x = Ar[2 * k]; a.y = Ar[2 * k + 1];
        b.x = P[2 * k]; b.y = P[2 * k + 1];
        wd[k] = a;
        wo[k] = b;
      }
#pragma unroll
      for (int k = 0; k < B; k++) Wj[k * B + r] = X[k];              // V, column-major
      Ws[2 * B * B + r] = as_;
    } else {
#pragma unroll
      for (int k = 0; k < B; k++) Wj[B * B + k * B + r] = X[k];      // U, column-major
      Wj[2 * B * B + r] = gr;                                        // Y
    }
  }
  // half 0 only
  __device__ __forceinline__ void add_right(double *REC, int n, int r) const {
    double *Wn = REC + n * BS;
    V2 *wd = reinterpret_cast<V2 *>(Wn + r * B);
#pragma unroll
    for (int k = 0; k < B / 2; k++) {
      V2 a = wd[k];
      a.x += P[2 * k]; a.y += P[2 * k + 1];
      wd[k] = a;
    }
    Wn[2 * B * B + r] += gn;
  }
};

// x_j = Y_j - U_j x_n - V_j x_s for every eliminated block of a group of G = 2^Q blocks, sub-levels in reverse; 16 lanes per
// pair (the caller provides >= 16 * G / 2 threads, or fewer pairs than it has 16-lane groups).  XS[0] (the group's first
// block) and XS[G] (the block beyond the group, or zero) are given.  sync(): LDS visibility between the lanes involved.
template <int B, int G, int Q, typename Sync>
__device__ __forceinline__ void cr_group_backward(const double *REC, double *XS, int cnt, int tid, Sync sync) {
  constexpr int BS = 2 * B * B + B;
  const int p = tid >> 4, r = tid & 15;
#pragma unroll 1
  for (int q = Q - 1; q >= 0; q--) {
    const int h = 1 << q, np = G >> (q + 1);
    const int s = p * 2 * h, j = s + h;
    if (p < np && j < cnt && r < B) {
      const int n = (j + h < cnt) ? j + h : G;
      const double *Rj = REC + j * BS;
      double v = Rj[2 * B * B + r];
#pragma unroll
      for (int k = 0; k < B; k++) v = fma(-Rj[B * B + k * B + r], XS[n * B + k], v);
#pragma unroll
      for (int k = 0; k < B; k++) v = fma(-Rj[k * B + r], XS[s * B + k], v);
      XS[j * B + r] = v;
    }
    sync();
  }
}

}  // namespace gps
