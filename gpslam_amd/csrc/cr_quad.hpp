// cr_quad.hpp -- the elimination of cr_step.hpp on FOUR adjacent DPP rows (one wave per pair): the sub-levels of a group that
// leave most of the workgroup idle (8, 4, 2, 1 pairs of a group of 32) cost what ONE lane's instruction stream costs, so the
// panel is cut once more.  Rows 0 / 1 carry the two column halves of O_j^T (-> U, the right neighbour's share -O_j U; row 0 also
// Y and -O_j Y), rows 2 / 3 the two column halves of F = O_s (-> V, the new coupling -O_j V, D_s - F^T V; row 2 also
// g_s - F^T Y).  Every row runs the Gauss-Jordan on its own copy of D_j (only the columns right of the pivot, in blocks of four
// where the block size allows it) and of g_j.  Per lane 336 multiply-adds for B = 12 instead of CrStepWide's 612, each
// with the operands and in the order of CrStep: bit-identical results.
#pragma once

#include "dpp.hpp"

namespace gps {

template <int K> __device__ __forceinline__ void fmac_self3(double *d, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %0, %3 row_newbcast:%4 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %1, %3 row_newbcast:%4 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])
      : "v"(m), "n"(K));
}
template <int K> __device__ __forceinline__ void fmac_self2(double *d, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %0, %2 row_newbcast:%3 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1])
      : "v"(m), "n"(K));
}
template <int K, bool NEG = false> __device__ __forceinline__ void fmac_bcast3(double *d, const double *s, double m) {
  if constexpr (NEG) {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %3, -%6 row_newbcast:%7 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %4, -%6 row_newbcast:%7 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %2, %5, -%6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(m), "n"(K));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %3, %6 row_newbcast:%7 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%7 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %2, %5, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(m), "n"(K));
  }
}
template <int K, bool NEG = false> __device__ __forceinline__ void fmac_bcast2v(double *d, const double *s, double m) {
  if constexpr (NEG) {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %2, -%4 row_newbcast:%5 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %3, -%4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1])
        : "v"(s[0]), "v"(s[1]), "v"(m), "n"(K));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%5 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1])
        : "v"(s[0]), "v"(s[1]), "v"(m), "n"(K));
  }
}
// half-width front ends: N = B / 2 in {6, 3, 2}
template <int K, int N> __device__ __forceinline__ void fmac_self_h(double *d, double m) {
  static_assert(N == 6 || N == 3 || N == 2, "half panels of the chain solver's block sizes");
  if constexpr (N == 6) fmac_self6<K>(d, m);
  else if constexpr (N == 3) fmac_self3<K>(d, m);
  else fmac_self2<K>(d, m);
}
template <int K, int N, bool NEG = false> __device__ __forceinline__ void fmac_bcast_h(double *d, const double *s, double m) {
  static_assert(N == 6 || N == 3 || N == 2, "half panels of the chain solver's block sizes");
  if constexpr (N == 6) fmac_bcast6<K, NEG>(d, s, m);
  else if constexpr (N == 3) fmac_bcast3<K, NEG>(d, s, m);
  else fmac_bcast2v<K, NEG>(d, s, m);
}

template <int B> struct CrStepQuad {
  static constexpr int BS = 2 * B * B + B, H = B / 2;
  typedef double V2 __attribute__((ext_vector_type(2)));
  // part 0 (rows 0, 1): X = row r of O_j^T, columns [ch H, ch H + H) -> U;  P = -O_j U
  // part 1 (rows 2, 3): X = row r of F                                  -> V;  P = -O_j V,  Ar = D_s - F^T V
  double X[H], P[H], Ar[H];
  double gr, as_, gn;

  // qrow = the lane's DPP row inside the wave (0..3).  Every lane of the wave calls this (DPP).
  __device__ __forceinline__ bool compute(const double *REC, int s, int j, int r, int rr, int qrow) {
    const int part = qrow >> 1, ch = qrow & 1;
    const double *Rj = REC + j * BS, *Rs = REC + s * BS;
    double Dr[B], Ol[B], Gr[B];
    {
      const V2 *dj = reinterpret_cast<const V2 *>(Rj + rr * B), *oj = reinterpret_cast<const V2 *>(Rj + B * B + rr * B);
#pragma unroll
      for (int k = 0; k < B / 2; k++) {
        const V2 a = dj[k], b = oj[k];
        Dr[2 * k] = a.x; Dr[2 * k + 1] = a.y;        // row r of D_j
        Ol[2 * k] = b.x; Ol[2 * k + 1] = b.y;        // row r of O_j
      }
      // part 0: row r of O_j^T = column r of O_j (stride B);  part 1: row r of F = O_s (contiguous)
      const double *xb = part ? Rs + B * B + rr * B + ch * H : Rj + B * B + ch * H * B + rr;
      const int xs = part ? 1 : B;
      const double *ab = Rs + rr * B + ch * H;        // row r of D_s, the lane's columns (part 1)
#pragma unroll
      for (int k = 0; k < H; k++) {
        X[k] = xb[k * xs];
        Ar[k] = ab[k];
      }
#pragma unroll
      for (int k = 0; k < B; k++) Gr[k] = Rs[B * B + k * B + rr];   // row r of F^T (part 1)
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
      // of D only the columns right of the pivot still matter (entries at or left of it inside a block become garbage that
      // nothing reads again)
      if constexpr (B == 12) {
        if (k < 3) fmac_self4<k>(Dr, nmp);
        if (k < 7) fmac_self4<k>(Dr + 4, nmp);
        if (k < 11) fmac_self4<k>(Dr + 8, nmp);
      } else {
        fmac_self_n<k, B>(Dr, nmp);
      }
      double pn = 1.0, r0 = 1.0;
      if constexpr (k + 1 < B) {
        pn = row_bcast<(k + 1 < B ? k + 1 : 0)>(Dr[k + 1 < B ? k + 1 : 0]);
        r0 = __builtin_amdgcn_rcp(pn);
      }
      __builtin_amdgcn_sched_barrier(0);
      fmac_self_h<k, H>(X, nmp);
      if constexpr (k + 1 < B) { r0 = fma(fma(-pn, r0, 1.0), r0, r0); r0 = fma(fma(-pn, r0, 1.0), r0, r0); }
      __builtin_amdgcn_sched_barrier(0);
      fmac_self1<k>(gr, nmp);
      piv = pn; inv = r0;
    });
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < H; k++) { X[k] *= invs; P[k] = 0.0; }     // U_j / V_j: row r, the lane's columns
    gr *= invs;                                                     // Y_j
    __builtin_amdgcn_sched_barrier(0);
    // one instruction stream for the four rows (a branch by DPP row would run both sides one after the other): rows 0 / 1 add
    // zeros to their copy of Ar
    static_for<0, B>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      const double ol = Ol[i], gg = part ? Gr[i] : 0.0;
      fmac_bcast_h<i, H, true>(P, X, ol);           // part 0: -O_j U_j;  part 1: -O_j V_j (the coupling of s to n)
      fmac_bcast2<i, true>(gn, as_, gr, ol, gg);   // -O_j Y_j (row 0 uses it),  g_s -= F^T Y_j (row 2)
      fmac_bcast_h<i, H, true>(Ar, X, gg);          // part 1: D_s -= F^T V_j
    });
    __builtin_amdgcn_sched_barrier(0);
    return !(invs > 0.0);     // a pivot that is not positive (or not a number) leaves such a reciprocal in its own lane
  }

  // the pair's own blocks: s in place, j as the factor record [V | U | Y] (row lanes of active pairs only)
  __device__ __forceinline__ void store_own(double *REC, int s, int j, int r, int qrow) const {
    const int part = qrow >> 1, ch = qrow & 1, c0 = ch * H;
    double *Ws = REC + s * BS, *Wj = REC + j * BS;
    if (part) {
#pragma unroll
      for (int k = 0; k < H; k++) {
        Ws[r * B + c0 + k] = Ar[k];
        Ws[B * B + r * B + c0 + k] = P[k];
        Wj[(c0 + k) * B + r] = X[k];              // V, column-major
      }
      if (ch == 0) Ws[2 * B * B + r] = as_;
    } else {
#pragma unroll
      for (int k = 0; k < H; k++) Wj[B * B + (c0 + k) * B + r] = X[k];   // U, column-major
      if (ch == 0) Wj[2 * B * B + r] = gr;                                // Y
    }
  }
  // the right neighbour's share: rows 0 / 1 (after the barrier behind store_own)
  __device__ __forceinline__ void add_right(double *REC, int n, int r, int qrow) const {
    const int ch = qrow & 1, c0 = ch * H;
    double *Wn = REC + n * BS;
#pragma unroll
    for (int k = 0; k < H; k++) Wn[r * B + c0 + k] += P[k];
    if (ch == 0) Wn[2 * B * B + r] += gn;
  }
};

}  // namespace gps
