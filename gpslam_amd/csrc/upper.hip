// upper.hip -- the solver levels above level 0: LDS-resident block cyclic reduction, G blocks per workgroup and launch.
// See upper.hpp for the scheme and cr_step.hpp for one elimination (the row-layout block step of k_chunk_forward_rows:
// lane r of a 16-lane DPP row holds ROW r of the panel [D~_j | O_j^T | F | g~_j], Gauss-Jordan multipliers are lane-local,
// the pivot / source rows travel fused into the multiply-adds -- v_fmac_f64_dpp row_newbcast, dpp.hpp).
// Back-substitution: x_j = Y_j - U_j x_n - V_j x_s, sub-levels in reverse.
#include "upper.hpp"
#include "cr_step.hpp"
#include "cr_quad.hpp"

#include <cstdio>
#include <cstdlib>

namespace gps {

namespace {

template <int B, int G> struct UpDims {
  static_assert(G == 32 || G == 4, "group sizes: 32 (a launch of its own), 4 (the level the row-layout level-0 kernels fold into their tail)");
  static constexpr int BS = 2 * B * B + B, AS = B * B + B;
  static constexpr int DP = B * B / 2, GP = B / 2, NPC = BS / 2;     // 16-byte pieces of D (or O), of g, of a record
  static constexpr int Q = (G == 32) ? 5 : 2;                         // sub-levels
  static constexpr int NT = (G == 32) ? 512 : 64, NW = NT / 64;       // threads, waves: a pair takes two DPP rows (G / 2 pairs), later four
  static constexpr int UF = (G * NPC + NT - 1) / NT;                  // 16-byte pieces per thread of a group's records
  static constexpr size_t lds_fwd(bool top) { return ((size_t)(G + 1) * BS + (top ? (size_t)(G + 1) * B : 0)) * sizeof(double); }
  static constexpr size_t lds_bwd() { return ((size_t)G * BS + (size_t)(G + 1) * B) * sizeof(double); }
};

typedef double V2 __attribute__((ext_vector_type(2)));

#ifdef GPS_TRACE_UPPER
// debug builds only (scripts/trace_upper.py): s_memrealtime stamps of the waves of workgroup 0 of the last !TOP and TOP forward launches
__device__ unsigned long long g_up_trace[2 * 8 * 64];
#define UP_TR(slot) do { if (blockIdx.x == 0 && lane == 0) g_up_trace[((TOP ? 1 : 0) * 8 + wave) * 64 + (slot)] = wall_clock64(); } while (0)
#else
#define UP_TR(slot) do { } while (0)
#endif

template <int B, int G, bool TOP>
__global__ void __launch_bounds__((UpDims<B, G>::NT)) k_multi_forward(UpFwdArgs a) {
  typedef UpDims<B, G> DM;
  constexpr int BS = DM::BS, AS = DM::AS, DP = DM::DP, GP = DM::GP, NPC = DM::NPC, Q = DM::Q, NT = DM::NT, NW = DM::NW, UF = DM::UF;
  extern __shared__ __attribute__((aligned(16))) double REC[];   // (G + 1) records; TOP: + (G + 1) solutions
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, r = lane & 15;
  const int g = blockIdx.x, base = g * G;
  const int cnt = min(G, a.n - base);
  const bool rowlane = r < B;
  const int rr = rowlane ? r : 0;            // idle lanes shadow row 0 (they never store)
  UP_TR(0);

  // ---- the group's records (+ the addends the level below sent them) into LDS, as 16-byte pieces; every load of the
  // thread is issued before the first one is consumed (a load per loop iteration had exposed a memory round trip each:
  // 17 500 of the kernel's 79 000 cycles).  The group's first block keeps its addend out: the previous group carries it
  // upward (TOP: there is no level above, it joins here).
  {
    V2 v[UF], w[UF];
    const V2 *blk2 = reinterpret_cast<const V2 *>(a.blk + (size_t)base * BS);
    const V2 *add2 = reinterpret_cast<const V2 *>(a.add != nullptr ? a.add + (size_t)base * AS : a.blk + (size_t)base * BS);
#pragma unroll
    for (int u = 0; u < UF; u++) {
      const int idx = min(tid + NT * u, cnt * NPC - 1);
      const int i = idx / NPC, t = idx - i * NPC;
      v[u] = blk2[idx];
      w[u] = add2[i * (AS / 2) + (t < DP ? t : (t >= 2 * DP ? t - DP : 0))];
    }
#pragma unroll
    for (int u = 0; u < UF; u++) {
      const int idx = tid + NT * u;
      const int i = idx / NPC, t = idx - i * NPC;
      const bool joins = a.add != nullptr && (i >= 1 || TOP) && (t < DP || t >= 2 * DP);
      V2 x = v[u];
      if (joins) { x.x += w[u].x; x.y += w[u].y; }
      if (idx < cnt * NPC) reinterpret_cast<V2 *>(REC)[idx] = x;
    }
  }
  {   // the virtual block G: what is already owed to the block beyond the group
    const int xi = base + cnt;
    const bool have = a.add != nullptr && ((xi < a.n) || (a.ext != 0));
    for (int t = tid; t < NPC; t += NT) {
      V2 v = {0.0, 0.0};
      if (have && (t < DP || t >= 2 * DP)) v = reinterpret_cast<const V2 *>(a.add + (size_t)xi * AS)[t < DP ? t : t - DP];
      reinterpret_cast<V2 *>(REC + G * BS)[t] = v;
    }
  }
  UP_TR(1);
  __syncthreads();
  UP_TR(2);

  // A sub-level costs what ONE lane's instruction stream costs (an elimination is VALU-issue bound even for a single wave), so
  // the panel of a pair is spread over as many DPP rows as the sub-level leaves free: two (CrStepWide: G / 2 pairs on NW * 4
  // rows) in the first sub-level, four (CrStepQuad, one wave per pair) in all the others.  Same values in every form
  // (and as CrStep, one row per pair, which the tail of the level-0 kernels runs).
  static_assert((G >> 1) <= 2 * NW && (G >> 2) <= NW, "two rows per pair in the first sub-level, four from the second on");
  auto sub_level = [&](int q, auto form) {
    constexpr bool quad = decltype(form)::value;
    const int h = 1 << q, np = G >> (q + 1);
    const int half = row & 1;
    const int p = quad ? wave : (row >> 1) * NW + wave;      // pairs spread over the waves first, then over DPP rows
    const int s = p * 2 * h, j = s + h;
    const bool act = (p < np) && (j < cnt);
    const int n = (j + h < cnt) ? j + h : G;
    std::conditional_t<quad, CrStepQuad<B>, CrStepWide<B>> st;
    if (__ballot(act) != 0ull) {             // (idle DPP rows of a working wave recompute block 0; they never store)
      const bool bad = st.compute(REC, act ? s : 0, act ? j : 0, r, rr, quad ? row : half);
      if (bad && act) *a.flag = 1;             // (the lane of the failed pivot reports)
    }
    UP_TR(3 + 6 * q);
    lds_barrier();                            // every pair has read its operands
    UP_TR(4 + 6 * q);
    if (act && rowlane) st.store_own(REC, s, j, r, quad ? row : half);
    UP_TR(5 + 6 * q);
    lds_barrier();                            // the pairs' own blocks are in place: now the right neighbours' shares
    UP_TR(6 + 6 * q);
    if (act && rowlane) {
      if constexpr (quad) { if (row < 2) st.add_right(REC, n, r, row); }
      else { if (half == 0) st.add_right(REC, n, r); }
    }
    UP_TR(7 + 6 * q);
    lds_barrier();
    UP_TR(8 + 6 * q);
  };
  if constexpr ((G >> 1) > NW) sub_level(0, std::false_type{});
  else sub_level(0, std::true_type{});
#pragma unroll 1
  for (int q = 1; q < Q; q++) sub_level(q, std::true_type{});

  if (!TOP) {
    // the factor records of every sub-level leave for the back-substitution launch in one pass (blocks 1 .. cnt - 1 were all
    // eliminated, each at the sub-level of its lowest set bit, and nothing touched their records afterwards)
    for (int idx = tid; idx < (cnt - 1) * NPC; idx += NT)
      reinterpret_cast<V2 *>(a.blk + (size_t)(base + 1) * BS)[idx] = reinterpret_cast<const V2 *>(REC + BS)[idx];
    // what is left of the group: its first block (now coupled to the block beyond the group) and what that block is owed
    for (int t = tid; t < NPC; t += NT)
      reinterpret_cast<V2 *>(a.up_blk + (size_t)g * BS)[t] = reinterpret_cast<const V2 *>(REC)[t];
    for (int t = tid; t < DP + GP; t += NT)
      reinterpret_cast<V2 *>(a.up_add + (size_t)(g + 1) * AS)[t] = reinterpret_cast<const V2 *>(REC + G * BS)[t < DP ? t : t + DP];
    UP_TR(40);
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
    static_for<0, B>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      const double piv = row_bcast<k>(Dr[k]);
      const double inv = fast_rcp(piv);
      const bool isk = (r == k);
      invs = isk ? inv : invs;
      const double nmp = isk ? 0.0 : -(Dr[k] * inv);
      fmac_self_n<k, B>(Dr, nmp);
      fmac_self1<k>(gr, nmp);
    });
    if (!(invs > 0.0)) *a.flag = 1;
    if (row == 0 && rowlane) XS[r] = gr * invs;
  }
  if (tid < B) XS[G * B + tid] = 0.0;          // nothing beyond the top level
  UP_TR(41);
  lds_barrier();
  cr_group_backward<B, G, Q>(REC, XS, cnt, tid, [] { lds_barrier(); });
  UP_TR(42);
  for (int idx = tid; idx < cnt * B; idx += NT) a.x[(size_t)base * B + idx] = XS[idx];
  UP_TR(43);
}

template <int B, int G>
__global__ void __launch_bounds__((UpDims<B, G>::NT)) k_multi_backward(UpBwdArgs a) {
  typedef UpDims<B, G> DM;
  constexpr int BS = DM::BS, NPC = DM::NPC, Q = DM::Q, NT = DM::NT, UF = DM::UF;
  extern __shared__ __attribute__((aligned(16))) double REC[];   // G records, then (G + 1) solutions
  double *XS = REC + G * BS;
  const int tid = threadIdx.x;
  const int g = blockIdx.x, base = g * G;
  const int cnt = min(G, a.n - base);
  {   // (all loads first, as in the forward launch; the group's first block was not eliminated here but comes along)
    V2 v[UF];
    const V2 *blk2 = reinterpret_cast<const V2 *>(a.blk + (size_t)base * BS);
#pragma unroll
    for (int u = 0; u < UF; u++) v[u] = blk2[min(tid + NT * u, cnt * NPC - 1)];
#pragma unroll
    for (int u = 0; u < UF; u++)
      if (tid + NT * u < cnt * NPC) reinterpret_cast<V2 *>(REC)[tid + NT * u] = v[u];
  }
  const bool have = (base + cnt < a.n) || (a.ext != 0);
  if (tid < B) {
    XS[tid] = a.xup[(size_t)g * B + tid];
    XS[G * B + tid] = have ? a.xup[(size_t)(g + 1) * B + tid] : 0.0;
  }
  __syncthreads();
  cr_group_backward<B, G, Q>(REC, XS, cnt, tid, [] { lds_barrier(); });
  for (int idx = tid; idx < cnt * B; idx += NT) a.x[(size_t)base * B + idx] = XS[idx];
  // the block beyond the level lives on the next rank: park its solution in the extra slot, where the level below
  // expects the solution of "block n"
  if (a.ext != 0 && base + cnt == a.n && tid < B) a.x[(size_t)a.n * B + tid] = XS[G * B + tid];
}

template <typename K> hipError_t allow_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
// The attribute belongs to the CURRENT DEVICE's function object, not to the process: a process that drives a second GPU
// (ChainSolver(device=1) after device 0) must set it there as well.  One flag per device ordinal.
constexpr int kMaxDevices = 64;
inline bool *ready_slot(bool (&tab)[kMaxDevices]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;   // unknown ordinal: set it every time
  return &tab[dev];
}

template <int B, int G> int fwd_b(bool top, const UpFwdArgs &a0, hipStream_t st) {
  typedef UpDims<B, G> DM;
  static bool ready_tab[kMaxDevices] = {};   // (the attribute is per kernel and per device: set once each, to the one size)
  bool *ready = ready_slot(ready_tab);
  hipError_t e;
  const UpFwdArgs &a = a0;
  if (!ready || !*ready) {
    if ((e = allow_lds(&k_multi_forward<B, G, true>, DM::lds_fwd(true))) != hipSuccess) return (int)e;
    if ((e = allow_lds(&k_multi_forward<B, G, false>, DM::lds_fwd(false))) != hipSuccess) return (int)e;
    if (ready) *ready = true;
  }
  const int groups = (a.n + G - 1) / G;
  if (top) k_multi_forward<B, G, true><<<dim3(1), dim3(DM::NT), DM::lds_fwd(true), st>>>(a);
  else k_multi_forward<B, G, false><<<dim3(groups), dim3(DM::NT), DM::lds_fwd(false), st>>>(a);
  return (int)hipGetLastError();
}
template <int B, int G> int bwd_b(const UpBwdArgs &a, hipStream_t st) {
  typedef UpDims<B, G> DM;
  static bool ready_tab[kMaxDevices] = {};
  bool *ready = ready_slot(ready_tab);
  hipError_t e;
  if (!ready || !*ready) {
    if ((e = allow_lds(&k_multi_backward<B, G>, DM::lds_bwd())) != hipSuccess) return (int)e;
    if (ready) *ready = true;
  }
  const int groups = (a.n + G - 1) / G;
  k_multi_backward<B, G><<<dim3(groups), dim3(DM::NT), DM::lds_bwd(), st>>>(a);
  return (int)hipGetLastError();
}

}  // namespace

#ifdef GPS_TRACE_UPPER
extern "C" int gpslam_hip_debug_upper_trace(unsigned long long *out) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_up_trace), sizeof(g_up_trace)) == hipSuccess ? 0 : -2;
}
#endif
int upper_forward(int B, int G, bool top, const UpFwdArgs &a, hipStream_t st) {
  if (a.n <= 0 || (top && a.n > G) || (G != 32 && G != 4)) return (int)hipErrorInvalidValue;
  switch (B) {
    case 4: return G == 32 ? fwd_b<4, 32>(top, a, st) : fwd_b<4, 4>(top, a, st);
    case 6: return G == 32 ? fwd_b<6, 32>(top, a, st) : fwd_b<6, 4>(top, a, st);
    case 12: return G == 32 ? fwd_b<12, 32>(top, a, st) : fwd_b<12, 4>(top, a, st);
  }
  return (int)hipErrorInvalidValue;
}
int upper_backward(int B, int G, const UpBwdArgs &a, hipStream_t st) {
  if (a.n <= 0 || (G != 32 && G != 4)) return (int)hipErrorInvalidValue;
  switch (B) {
    case 4: return G == 32 ? bwd_b<4, 32>(a, st) : bwd_b<4, 4>(a, st);
    case 6: return G == 32 ? bwd_b<6, 32>(a, st) : bwd_b<6, 4>(a, st);
    case 12: return G == 32 ? bwd_b<12, 32>(a, st) : bwd_b<12, 4>(a, st);
  }
  return (int)hipErrorInvalidValue;
}

}  // namespace gps
