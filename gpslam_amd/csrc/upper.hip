// upper.hip -- the solver levels above level 0: LDS-resident block cyclic reduction, 32 blocks per workgroup and launch.
// See upper.hpp for the scheme.  The arithmetic of one elimination is the row-layout block step of
// k_chunk_forward_rows (kernels.hpp): lane r of a 16-lane DPP row holds ROW r of the panel [D~_j | O_j^T | F | g~_j],
// Gauss-Jordan multipliers are lane-local, the pivot / source rows travel fused into the multiply-adds
// (v_fmac_f64_dpp row_newbcast, dpp.hpp).
//
// One elimination (pair (s, j), n = the block 2^q to the right of j, possibly the virtual block G beyond the group):
//   D_j x_j + O_j^T x_n + F x_s = g_j,  F = O_s          (O_i = H[right neighbour of i, i])
//   U = D_j^-1 O_j^T,  V = D_j^-1 F,  Y = D_j^-1 g_j        -> record j becomes [V | U | Y] (column-major, as level 0 stores it)
//   D_s -= F^T V,  g_s -= F^T Y,  O_s <- -O_j V             (s now couples to n)
//   D_n -= O_j U,  g_n -= O_j Y                             (added after a barrier: n is the s of the next pair)
// Back-substitution: x_j = Y_j - U_j x_n - V_j x_s, sub-levels in reverse.
#include "upper.hpp"
#include "dpp.hpp"

namespace gps {

namespace {

template <int B> struct UpDims {
  static constexpr int BS = 2 * B * B + B, AS = B * B + B;
  static constexpr int DP = B * B / 2, GP = B / 2, NPC = BS / 2;     // 16-byte pieces of D (or O), of g, of a record
  static constexpr size_t lds_fwd(bool top) { return ((size_t)(kUpG + 1) * BS + (top ? (size_t)(kUpG + 1) * B : 0)) * sizeof(double); }
  static constexpr size_t lds_bwd() { return ((size_t)kUpG * BS + (size_t)(kUpG + 1) * B) * sizeof(double); }
};

typedef double V2 __attribute__((ext_vector_type(2)));

// x_j = Y_j - U_j x_n - V_j x_s for every eliminated block of the group, sub-levels in reverse; 16 lanes per pair.
// XS[0] (the group's first block) and XS[G] (the block beyond the group, or zero) are given.
template <int B>
__device__ __forceinline__ void group_backward(const double *REC, double *XS, int cnt, int tid) {
  constexpr int BS = UpDims<B>::BS, G = kUpG;
  const int p = tid >> 4, r = tid & 15;
#pragma unroll 1
  for (int q = kUpQ - 1; q >= 0; q--) {
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
    lds_barrier();
  }
}

template <int B, bool TOP>
__global__ void __launch_bounds__(256) k_multi_forward(UpFwdArgs a) {
  constexpr int BS = UpDims<B>::BS, AS = UpDims<B>::AS, DP = UpDims<B>::DP, GP = UpDims<B>::GP, NPC = UpDims<B>::NPC, G = kUpG;
  extern __shared__ __attribute__((aligned(16))) double REC[];   // (G + 1) records; TOP: + (G + 1) solutions
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, r = lane & 15;
  const int g = blockIdx.x, base = g * G;
  const int cnt = min(G, a.n - base);
  const bool rowlane = r < B;
  const int rr = rowlane ? r : 0;            // idle lanes shadow row 0 (they never store)

  // ---- the group's records (+ the addends the level below sent them) into LDS, as 16-byte pieces.  The group's first
  // block keeps its addend out: the previous group carries it upward (TOP: there is no level above, it joins here).
  for (int idx = tid; idx < cnt * NPC; idx += 256) {
    const int i = idx / NPC, t = idx - i * NPC;
    V2 v = reinterpret_cast<const V2 *>(a.blk + (size_t)(base + i) * BS)[t];
    if (a.add != nullptr && (i >= 1 || TOP) && (t < DP || t >= 2 * DP)) {
      const V2 w = reinterpret_cast<const V2 *>(a.add + (size_t)(base + i) * AS)[t < DP ? t : t - DP];
      v.x += w.x; v.y += w.y;
    }
    reinterpret_cast<V2 *>(REC + i * BS)[t] = v;
  }
  {   // the virtual block G: what is already owed to the block beyond the group
    const int xi = base + cnt;
    const bool have = a.add != nullptr && ((xi < a.n) || (a.ext != 0));
    for (int t = tid; t < NPC; t += 256) {
      V2 v = {0.0, 0.0};
      if (have && (t < DP || t >= 2 * DP)) v = reinterpret_cast<const V2 *>(a.add + (size_t)xi * AS)[t < DP ? t : t - DP];
      reinterpret_cast<V2 *>(REC + G * BS)[t] = v;
    }
  }
  __syncthreads();

#pragma unroll 1
  for (int q = 0; q < kUpQ; q++) {
    const int h = 1 << q, np = G >> (q + 1);
    const int p = row * 4 + wave;            // the pairs of a sub-level spread over the waves first, then over DPP rows
    const int s = p * 2 * h, j = s + h;
    const bool act = (p < np) && (j < cnt);
    const int n = (j + h < cnt) ? j + h : G;
    const bool wact = __ballot(act) != 0ull;
    double Or[B], Fr[B], Ar[B], Dn[B], Fn[B];
    double gr = 0.0, as_ = 0.0, gn = 0.0;
    if (wact) {
      const double *Rj = REC + (act ? j : 0) * BS, *Rs = REC + (act ? s : 0) * BS;   // idle rows recompute block 0 (never stored)
      double Dr[B], Gr[B], Ol[B];
#pragma unroll
      for (int k = 0; k < B; k++) {
        Dr[k] = Rj[rr * B + k];                    // row r of D_j
        Or[k] = Rj[B * B + k * B + rr];            // row r of O_j^T
        Ol[k] = Rj[B * B + rr * B + k];            // row r of O_j
        Ar[k] = Rs[rr * B + k];                    // row r of D_s
        Fr[k] = Rs[B * B + rr * B + k];            // row r of F = O_s
        Gr[k] = Rs[B * B + k * B + rr];            // row r of F^T
      }
      gr = Rj[2 * B * B + rr];
      as_ = Rs[2 * B * B + rr];
      __builtin_amdgcn_sched_barrier(0);
      // Gauss-Jordan on D_j by row operations; the pivot row stays unscaled until the end (scaling commutes)
      double invs = 1.0;
      bool bad = false;
      static_for<0, B>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const double piv = row_bcast<k>(Dr[k]);
        bad = bad || !(piv > 0.0);
        const double inv = fast_rcp(piv);
        const bool isk = (r == k);
        invs = isk ? inv : invs;
        const double nmp = isk ? 0.0 : -(Dr[k] * inv);
        fmac_self_n<k, B>(Dr, nmp);              // (entries at or left of the pivot become garbage that nothing reads again)
        fmac_self_n<k, B>(Or, nmp);
        fmac_self_n<k, B>(Fr, nmp);
        fmac_self1<k>(gr, nmp);
      });
      if (bad && act && r == 0) *a.flag = 1;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < B; k++) { Or[k] *= invs; Fr[k] *= invs; Dn[k] = 0.0; Fn[k] = 0.0; }   // U_j, V_j: row r
      gr *= invs;                                                                                  // Y_j
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, B>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        const double nol = -Ol[i], ngg = -Gr[i];
        fmac_bcast_n<i, B>(Dn, Or, nol);         // -O_j U_j: row r
        fmac_bcast2<i>(gn, as_, gr, nol, ngg);   // -O_j Y_j,  g_s -= F^T Y_j
        fmac_bcast_n<i, B>(Fn, Fr, nol);         // -O_j V_j: the coupling of s to n
        fmac_bcast_n<i, B>(Ar, Fr, ngg);         // D_s -= F^T V_j
      });
      __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();                            // every pair has read its operands
    if (act && rowlane) {
      double *Ws = REC + s * BS, *Wj = REC + j * BS;
#pragma unroll
      for (int k = 0; k < B; k++) {
        Ws[r * B + k] = Ar[k];
        Ws[B * B + r * B + k] = Fn[k];
        Wj[k * B + r] = Fr[k];                  // the factor record is column-major [V | U | Y]
        Wj[B * B + k * B + r] = Or[k];
      }
      Ws[2 * B * B + r] = as_;
      Wj[2 * B * B + r] = gr;
    }
    lds_barrier();                            // the pairs' own blocks are in place: now the right neighbours' shares
    if (act && rowlane) {
      double *Wn = REC + n * BS;
#pragma unroll
      for (int k = 0; k < B; k++) Wn[r * B + k] += Dn[k];
      Wn[2 * B * B + r] += gn;
    }
    if (!TOP) {   // this sub-level's factor records leave for the back-substitution launch
      for (int idx = tid; idx < np * NPC; idx += 256) {
        const int pp = idx / NPC, t = idx - pp * NPC;
        const int jj = pp * 2 * h + h;
        if (jj < cnt) reinterpret_cast<V2 *>(a.blk + (size_t)(base + jj) * BS)[t] = reinterpret_cast<const V2 *>(REC + jj * BS)[t];
      }
    }
    lds_barrier();
  }

  if (!TOP) {
    // what is left of the group: its first block (now coupled to the block beyond the group) and what that block is owed
    for (int t = tid; t < NPC; t += 256)
      reinterpret_cast<V2 *>(a.up_blk + (size_t)g * BS)[t] = reinterpret_cast<const V2 *>(REC)[t];
    for (int t = tid; t < DP + GP; t += 256)
      reinterpret_cast<V2 *>(a.up_add + (size_t)(g + 1) * AS)[t] = reinterpret_cast<const V2 *>(REC + G * BS)[t < DP ? t : t + DP];
    return;
  }

  // ---- TOP: solve the last block, back-substitute the group in LDS
  double *XS = REC + (G + 1) * BS;
  if (wave == 0) {
    double Dr[B];
#pragma unroll
    for (int k = 0; k < B; k++) Dr[k] = REC[rr * B + k];
    double gr = REC[2 * B * B + rr];
    double invs = 1.0;
    bool bad = false;
    static_for<0, B>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      const double piv = row_bcast<k>(Dr[k]);
      bad = bad || !(piv > 0.0);
      const double inv = fast_rcp(piv);
      const bool isk = (r == k);
      invs = isk ? inv : invs;
      const double nmp = isk ? 0.0 : -(Dr[k] * inv);
      fmac_self_n<k, B>(Dr, nmp);
      fmac_self1<k>(gr, nmp);
    });
    if (bad && lane == 0) *a.flag = 1;
    if (row == 0 && rowlane) XS[r] = gr * invs;
  }
  if (tid < B) XS[G * B + tid] = 0.0;          // nothing beyond the top level
  lds_barrier();
  group_backward<B>(REC, XS, cnt, tid);
  for (int idx = tid; idx < cnt * B; idx += 256) a.x[(size_t)base * B + idx] = XS[idx];
}

template <int B>
__global__ void __launch_bounds__(256) k_multi_backward(UpBwdArgs a) {
  constexpr int BS = UpDims<B>::BS, NPC = UpDims<B>::NPC, G = kUpG;
  extern __shared__ __attribute__((aligned(16))) double REC[];   // G records, then (G + 1) solutions
  double *XS = REC + G * BS;
  const int tid = threadIdx.x;
  const int g = blockIdx.x, base = g * G;
  const int cnt = min(G, a.n - base);
  for (int idx = tid + NPC; idx < cnt * NPC; idx += 256) {       // (the group's first block was not eliminated here)
    const int i = idx / NPC, t = idx - i * NPC;
    reinterpret_cast<V2 *>(REC + i * BS)[t] = reinterpret_cast<const V2 *>(a.blk + (size_t)(base + i) * BS)[t];
  }
  const bool have = (base + cnt < a.n) || (a.ext != 0);
  if (tid < B) {
    XS[tid] = a.xup[(size_t)g * B + tid];
    XS[G * B + tid] = have ? a.xup[(size_t)(g + 1) * B + tid] : 0.0;
  }
  __syncthreads();
  group_backward<B>(REC, XS, cnt, tid);
  for (int idx = tid; idx < cnt * B; idx += 256) a.x[(size_t)base * B + idx] = XS[idx];
  // the block beyond the level lives on the next rank: park its solution in the extra slot, where the level below
  // expects the solution of "block n"
  if (a.ext != 0 && base + cnt == a.n && tid < B) a.x[(size_t)a.n * B + tid] = XS[G * B + tid];
}

template <typename K> hipError_t allow_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int B> int fwd_b(bool top, const UpFwdArgs &a, hipStream_t st) {
  static bool ready = false;                  // (the attribute is per kernel and process-wide: set once, to the one size)
  hipError_t e;
  if (!ready) {
    if ((e = allow_lds(&k_multi_forward<B, true>, UpDims<B>::lds_fwd(true))) != hipSuccess) return (int)e;
    if ((e = allow_lds(&k_multi_forward<B, false>, UpDims<B>::lds_fwd(false))) != hipSuccess) return (int)e;
    ready = true;
  }
  const int groups = (a.n + kUpG - 1) / kUpG;
  if (top) k_multi_forward<B, true><<<dim3(1), dim3(256), UpDims<B>::lds_fwd(true), st>>>(a);
  else k_multi_forward<B, false><<<dim3(groups), dim3(256), UpDims<B>::lds_fwd(false), st>>>(a);
  return (int)hipGetLastError();
}
template <int B> int bwd_b(const UpBwdArgs &a, hipStream_t st) {
  static bool ready = false;
  hipError_t e;
  if (!ready) {
    if ((e = allow_lds(&k_multi_backward<B>, UpDims<B>::lds_bwd())) != hipSuccess) return (int)e;
    ready = true;
  }
  const int groups = (a.n + kUpG - 1) / kUpG;
  k_multi_backward<B><<<dim3(groups), dim3(256), UpDims<B>::lds_bwd(), st>>>(a);
  return (int)hipGetLastError();
}

}  // namespace

int upper_forward(int B, bool top, const UpFwdArgs &a, hipStream_t st) {
  if (a.n <= 0 || (top && a.n > kUpG)) return (int)hipErrorInvalidValue;
  switch (B) {
    case 4: return fwd_b<4>(top, a, st);
    case 6: return fwd_b<6>(top, a, st);
    case 12: return fwd_b<12>(top, a, st);
  }
  return (int)hipErrorInvalidValue;
}
int upper_backward(int B, const UpBwdArgs &a, hipStream_t st) {
  if (a.n <= 0) return (int)hipErrorInvalidValue;
  switch (B) {
    case 4: return bwd_b<4>(a, st);
    case 6: return bwd_b<6>(a, st);
    case 12: return bwd_b<12>(a, st);
  }
  return (int)hipErrorInvalidValue;
}

}  // namespace gps
