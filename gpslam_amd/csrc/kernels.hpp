// kernels.hpp -- HIP kernels of the GP-SLAM Gauss-Newton iteration (gfx950, wave64).
//
// HBM data layout (DESIGN.md has the full picture):
//   states      SoA   pose[k * stride + i], vel[k * stride + i]      (coalesced across factors)
//   row table   AoS   one record per whitened Jacobian row: rowLR[rho][2b] = [dL | dR], rowE[rho]
//                     rows are grouped by the LEFT state of their factor; rowptr[s] .. rowptr[s+1]
//   blocks      AoS   per state one record [D (b x b) | O (b x b) | G (R cols of b)], O_s = H[s+1, s]
//                     (k_fused_level0 never writes them: it forms them in LDS)
//   solver      the same records, overwritten in place by the elimination with [V | U | Y]
//
// Kernels:
//   k_gp          batched GaussianProcessPrior*::evaluateError + H1..H4 (+ whitening)     [K1]
//   k_simple      PriorFactor / BetweenFactor rows;  k_meas  measurement factors           [K2]
//   k_assemble_ghost   J^T J / J^T e per state from the row table (no atomics, fixed order) [K3]
//   k_lm_*        landmark border: Schur complement on the chain solution                 [K5]
//   k_interp_query     batched interpolatePose of the current estimate
//   k_chunk_forward / k_chunk_backward   partitioned block Gauss-Jordan, one wave per chunk,
//                 one panel column per lane, pivot broadcast through v_readlane            [K4]
//   k_chunk_forward_rows   level 0 for block size 12: four chunks per wave on 16-lane DPP rows, panel ROWS in lanes,
//                 pivot rows by v_mov_b64_dpp row_newbcast                                  [K4]
//   k_fused_level0     K3 inside K4 level 0: two-wave workgroups, one wave assembles the block records of four
//                 chunks from the row tables, the other eliminates them (LDS hand-over)      [K3 + K4]
//   k_retract     x <- x (+) delta, |delta|_inf                                            [K6]
#pragma once

#include "factors.hpp"
#include "dpp.hpp"
#include "cr_step.hpp"
#include "cr_quad.hpp"
#include <type_traits>

namespace gps {

constexpr int kMaxRhs = 28;  // 3B + R <= 64 for B = 12

template <typename T> struct UMat { T u[36]; };  // chol_upper(Qc^-1), d x d row-major

// ------------------------------------------------------------------ reductions

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// |v| for a maximum that must not lose NaNs (fmax drops them): a NaN update reads as +inf
template <typename T> __device__ __forceinline__ T abs_or_inf(T v) { return (v == v) ? fabs(v) : T(INFINITY); }
template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
// block-wide sum (blockDim.x multiple of 64, <= 256); result valid in thread 0
template <typename T> __device__ __forceinline__ T block_sum(T v) {
  __shared__ T red[4];
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  T r = T(0);
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)((blockDim.x + 63) >> 6); i++) r += red[i];
  return r;
}
template <typename T> __device__ __forceinline__ T block_max(T v) {
  __shared__ T redm[4];
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) redm[w] = v;
  __syncthreads();
  T r = T(0);
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)((blockDim.x + 63) >> 6); i++) r = fmax(r, redm[i]);
  return r;
}

// out[slot] = sum (mode 0) or max (mode 1) of in[0..n), single block, fixed order -> deterministic
// (round 6: a thread's operands are requested sixteen at a time and then added in the order they always were -- the error partials of
//  a 1e6-state graph with measurement factors are 47 000 numbers, 183 dependent load round trips per thread: 40-70 us of config 5's
//  SE(3) iteration, 2.5 x 20 us of config 4's)
template <typename T, int NT = 256> __device__ __forceinline__ T strided_fold(const T *in, int n, int mode) {
  T acc = T(0);
  int i = threadIdx.x;
  for (; i + 15 * NT < n; i += 16 * NT) {
    T v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = in[i + u * NT];
#pragma unroll
    for (int u = 0; u < 16; u++) acc = mode ? fmax(acc, v[u]) : acc + v[u];
  }
  for (; i < n; i += NT) acc = mode ? fmax(acc, in[i]) : acc + in[i];
  return acc;
}
template <typename T> __global__ void __launch_bounds__(256) k_final_reduce(const T *in, int n, double *out, int mode) {
  const T acc = strided_fold(in, n, mode);
  const T r = mode ? block_max(acc) : block_sum(acc);
  if (threadIdx.x == 0) *out = (double)r;
}


// ------------------------------------------------------------------ K1: GP prior rows

// fp32 arithmetic only: every factor here is invariant under a common translation of the variables it touches, and the
// states sit in HBM as fp64.  Before the arithmetic drops to float the translations are re-centred on the factor's
// first pose IN FP64, so that a Jacobian computed 1e5 m from the origin is as accurate as one computed at the origin
// (a float holds 1e5 m to 8 mm: the 0.1 m step between consecutive states would carry 4 digits).  The fp64
// instantiations do not re-centre: their results stay bit-identical to what the oracle parity tests pinned.
// (LINEAR3 is (x, y, theta) for OdometryFactor2DLinear / RangeBearingFactor2DLinear, whose Jacobians are evaluated AT theta:
//  only x and y are re-centred -- ADVICE r2; a plain GaussianProcessPriorLinear<3> chain loses nothing by that, its Jacobians
//  are constants.)
template <int MF> struct TransPart {
  static constexpr int n = (MF == POSE3) ? 3 : ((MF == POSE2 || MF == LINEAR3) ? 2 : ((MF == ROT3 || MF == ROT3_BIAS) ? 0 : MTraits<MF>::pd));
  static constexpr int off = (MF == POSE3) ? 9 : 0;
};
template <typename T> struct IsF64 { static constexpr bool v = false; };
template <> struct IsF64<double> { static constexpr bool v = true; };

// Inputs (states, landmarks, factor parameters) are ALWAYS fp64 in HBM; T is the arithmetic / row-table type.  In the
// fp32 mode (GPSLAM_FP32) the Jacobian rows, the normal equations and the solver run with T = float, while the residual
// is evaluated by the T = double error pass of the same kernels, which then also deposits its whitened error as the
// fp32 right-hand side (rowE32): fp32 linear algebra + fp64 residual = iterative refinement through the Gauss-Newton loop.
// Pending update (round 5).  Inside a fixed-count Gauss-Newton run the retraction of iteration k is folded into the linearisation
// of iteration k + 1: K1 reads a state, applies the update the previous solve left in the level-0 solution array, linearises at the
// updated values, and the thread that OWNS the state (the GP prior whose left state it is; the last state rides with the last
// factor) writes it into the OTHER state buffer -- the host swaps the two buffers behind the launch.  One launch (k_retract) and one
// read + write of every state less per iteration; Values::retract and NonlinearFactorGraph::linearize of GaussNewtonOptimizer::iterate
//  in one pass over the states.  The same arithmetic in the same order as k_retract: bit-identical states.
struct PendUpd {
  const double *dx = nullptr;   // N x b updates (level-0 solutions, R = 1), or null: nothing pending
  double *pose_w = nullptr, *vel_w = nullptr;   // the other state buffer (same SoA stride)
  const int *flag = nullptr;    // non-SPD flag of the solve that produced dx: set -> the states stay as they are
  int chart = 0, last = 0;      // retract chart; index of the chain's last state
};
// state i as K1 sees it: the stored value, or -- with an update pending -- its retraction; `own`: this thread writes it back
template <int MF> __device__ __forceinline__ void load_state_upd(const double *pose, const double *vel, int stride, int i, const PendUpd &u,
                                                                  bool own, double *p, double *v) {
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd, b = 2 * d;
#pragma unroll
  for (int k = 0; k < pd; k++) p[k] = pose[(size_t)k * stride + i];
#pragma unroll
  for (int k = 0; k < d; k++) v[k] = vel[(size_t)k * stride + i];
  if (u.dx != nullptr) {
    if (!*u.flag) {
      double dl[b], q[pd];
#pragma unroll
      for (int k = 0; k < b; k++) dl[k] = u.dx[(size_t)i * b + k];
      PoseFactors<double, MF, false>::retract(p, dl, u.chart, q);
#pragma unroll
      for (int k = 0; k < pd; k++) p[k] = q[k];
#pragma unroll
      for (int k = 0; k < d; k++) v[k] += dl[d + k];
    }
    if (own) {
#pragma unroll
      for (int k = 0; k < pd; k++) u.pose_w[(size_t)k * stride + i] = p[k];
#pragma unroll
      for (int k = 0; k < d; k++) u.vel_w[(size_t)k * stride + i] = v[k];
    }
  }
}
// states i and i + 1 of a two-state factor the same way, every request in flight before the first one is waited for (round 6): the
// single-state form behind a branch on the solve's flag put four dependent memory round trips in front of K1's arithmetic -- state,
// flag, update, and the same again for the right state -- 11 of a K1 wave's 33 us (scripts/trace_klin.py)
template <int MF> __device__ __forceinline__ void load_two_states_upd(const double *pose, const double *vel, int stride, int i, const PendUpd &u,
                                                                      bool own1, bool own2, double *p1, double *v1, double *p2, double *v2) {
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd, b = 2 * d;
#pragma unroll
  for (int k = 0; k < pd; k++) { p1[k] = pose[(size_t)k * stride + i]; p2[k] = pose[(size_t)k * stride + i + 1]; }
#pragma unroll
  for (int k = 0; k < d; k++) { v1[k] = vel[(size_t)k * stride + i]; v2[k] = vel[(size_t)k * stride + i + 1]; }
  if (u.dx != nullptr) {
    const int bad = *u.flag;
    double dl1[b], dl2[b];
#pragma unroll
    for (int k = 0; k < b; k++) { dl1[k] = u.dx[(size_t)i * b + k]; dl2[k] = u.dx[(size_t)(i + 1) * b + k]; }
    if (!bad) {
      double q[pd];
      PoseFactors<double, MF, false>::retract(p1, dl1, u.chart, q);
#pragma unroll
      for (int k = 0; k < pd; k++) p1[k] = q[k];
#pragma unroll
      for (int k = 0; k < d; k++) v1[k] += dl1[d + k];
      PoseFactors<double, MF, false>::retract(p2, dl2, u.chart, q);
#pragma unroll
      for (int k = 0; k < pd; k++) p2[k] = q[k];
#pragma unroll
      for (int k = 0; k < d; k++) v2[k] += dl2[d + k];
    }
    if (own1) {
#pragma unroll
      for (int k = 0; k < pd; k++) u.pose_w[(size_t)k * stride + i] = p1[k];
#pragma unroll
      for (int k = 0; k < d; k++) u.vel_w[(size_t)k * stride + i] = v1[k];
    }
    if (own2) {
#pragma unroll
      for (int k = 0; k < pd; k++) u.pose_w[(size_t)k * stride + i + 1] = p2[k];
#pragma unroll
      for (int k = 0; k < d; k++) u.vel_w[(size_t)k * stride + i + 1] = v2[k];
    }
  }
}

template <typename T> struct GpArgs {
  PendUpd pend;                // fp64 SE(3) record launches inside run_gn: the previous iteration's update, applied here
  const double *pose, *vel;    // SoA
  int stride;             // SoA stride (>= N + 1)
  int count;
  const int *left;        // left state of factor f
  const double *dt;
  const int *row0;        // first row of factor f in the row table
  float *rowE32;          // MODE 1 (error only, T = double): also store the whitened error here, or null
  T *rowLR, *rowE;        // MODE 0
  T *partial;             // per-block error partial sums
  T *out_e, *out_H;       // MODE 2: API layout
  UMat<T> U;
  int vw;                 // Pose3 only: velocities are world-frame [v; w] (the *Pose3VW factors)
  T *gps;                 // Pose3, MODE 0: structured records (kGps* below) instead of rows in rowLR / rowE, or null
};

// Structured record of one GaussianProcessPriorPose3 (what k_fused_level0's assembly wave reads when K1 feeds it directly).
// Round 4: the record holds what the whitened 12 x 24 Jacobian [L | R] is a FUNCTION of, not its columns.  With
//   X = Jinv = Jr^-1(r) = [[XA, 0], [XC, XA]]                      (GaussianProcessPriorPose3.h:76; 18 distinct entries)
//   J = Hlog Hcomp1 Hinv = -Jinv Ad(h^-1) = [[JA, 0], [JC, JA]]   (:79-84; the product of two such matrices is one again)
//   F = FD = d(Jinv v2)/dr by central differences = [[FA, 0], [FC, FD]]     (:81, :90; Pose3utils.cpp:167-179; 27 entries)
// the Jacobian is H1 = [J; F J], H2 = [-dt I; -I], H3 = [X; F X], H4 = [0; X] (:79-95), whitened by
// R_w = [[sa U, sb U], [0, sc U]], U = chol_upper(Qc^-1) the same for every factor.  Column c of the whitened halves is
//   L[:, c] = [U (sa J_c + sb F J_c); sc U F J_c]        c < 6 (pose)         [k2 U_c'; -sc U_c']           velocity column c'
//   R[:, c] = [U (sa X_c + sb F X_c); sc U F X_c]                              [sb U X_c'; sc U X_c']
// which the assembly wave forms lane by lane (lane c of a chunk's 16-lane DPP row holds column c): two 6 x 6 matrix-vector
// products with F and four with U per lane and state, all as v_fmac_f64_dpp blocks.  80 doubles = five 128-byte lines per
// factor instead of the 196 doubles of round 2/3's column records (and the 312 of plain rows + errors):
//   [0..8] XA  [9..17] XC  [18..26] JA  [27..35] JC  [36..44] FA  [45..53] FC  [54..62] FD   (3 x 3 blocks, row-major)
//   [63] 0 (the velocity lanes take their zero coefficients from here)
//   [64..75] whitened error,  [76] k2 = -(sa dt + sb), [77] sb, [78] sc, [79] sa
// K1 stages 16 doubles per lane and wave and writes whole 128-byte lines (wave_store_part): the order above is the order in
// which it produces the blocks.  Record F (one past the last factor) is all zeros: states without a GP prior read it.
#ifndef GPS_KLIN_WAVES
#define GPS_KLIN_WAVES 2   /* two K1 waves per SIMD (<= 256 VGPRs): linearise phase 87.8 vs 93.3 us with structured records */
#endif
// The same idea for the d = 3 manifolds (SE(2), SO(3), 3-D linear: block size 6; round 4).  There H1 = [J1; 0], H3 = [J3; 0] and
// H2 = [h2t I; h2b I], H4 = [0; h4b I] with constants (GaussianProcessPriorPose2.h:76-79, GaussianProcessPriorRot3.h:73-76,
// GaussianProcessPriorLinear.h:72-81), so of the whitened 6 x 12 Jacobian only two 3 x 3 blocks are data:
//   [0..8] A1 = U (sa J1)   [9..17] A3 = U (sa J3)   (row-major)        rows 0..2: [A1 | kLt U | A3 | kRt U]
//   [18..23] whitened error   [24] kLt [25] kRt [26] kLb [27] kRb      rows 3..5: [ 0 | kLb U |  0 | kRb U]
//   [28..31] 0                                                           (kLt = sa h2t + sb h2b, kRt = sb h4b, kLb = sc h2b, kRb = sc h4b)
// 32 doubles = two 128-byte lines per factor instead of 6 rows x 13 doubles = 624 bytes written as 96-byte fragments (1.45x write
// amplification measured: profiles/round4_c4_v0).  Consumers: k_assemble_ghost<6> and k_fused_level0<1, double, 6>.
constexpr int kGp3Len = 32, kGp3A1 = 0, kGp3A3 = 9, kGp3E = 18, kGp3S = 24;
// BetweenFactor<Pose3>(x_i, x_i+1) of a chain that runs the structured path (round 4): e = Log(measured^-1 x_i^-1 x_i+1),
// H2 = Jr^-1(e) = [[RA, 0], [RC, RA]], H1 = -Jr^-1(e) Ad((x_i^-1 x_i+1)^-1) = [[LA, 0], [LC, LA]] (GTSAM BetweenFactor.h through
// Pose3::between / Logmap; call site matlab/PlazaPose2.m:125), rows weighted by 1 / sigma:
//   [0..8] RA  [9..17] RC  [18..26] LA  [27..35] LC  (3 x 3, row-major)   [36..41] 1 / sigma   [42..47] whitened error
// 48 doubles = three whole lines instead of six compact rows of 12 + 1 doubles (624 bytes in 96-byte fragments).  The assembly wave
// of k_fused_level0 builds column c of [H1 | H2] in lane c < 6 exactly as it builds the columns of the GP prior's J and X.
constexpr int kBtwLen = 48, kBtwRA = 0, kBtwRC = 9, kBtwLA = 18, kBtwLC = 27, kBtwW = 36, kBtwE = 42;
// Interpolated measurement rows of an SE(3) chain on the structured path (round 5; GPInterpolatedGPSFactorPose3.h:66-95 over
// GaussianProcessInterpolatorPose3.h:82-98).  A whitened row of such a factor is  w Hp [H1 H2 | H3 H4]  with
//   H1 = Ad(Exp(xi)^-1) + He (p11 J + p12 F J),  H2 = l12 He,  H3 = He (p11 X + p12 F X),  H4 = p12 He X
// where X = Jr^-1(r), J = -X Ad(h^-1) and F = d(X v2)/dr are the interval's GP record (kGps*) and l12, p11, p12 the interpolation
// coefficients of tau.  With mu = w Hp He (6) the row is
//   L = [ Lp | l12 mu ],   R = [ mu (p11 X + p12 F X) | p12 mu X ],   Lp = w Hp Ad(Exp(xi)^-1) + mu (p11 J + p12 F J)
// so k_meas writes ONE 128-byte line per row -- [Lp (6) | mu (6) | whitened error, p11, p12, l12] -- instead of 24 + 1 doubles, and
// the assembly wave of k_fused_level0<4> forms R from mu and the X / F X columns it holds for the GP prior anyway: lane c < 16 of
// a chunk's DPP row loads element c of the line (one load per row and lane), mu travels by row_newbcast from lanes 6..11, the four
// scalars from lanes 12..15.
constexpr int kIRowLen = 16, kIRowLp = 0, kIRowMu = 6, kIRowE = 12, kIRowP11 = 13, kIRowP12 = 14, kIRowL12 = 15;
constexpr int kGpsLen = 80, kGpsXA = 0, kGpsXC = 9, kGpsJA = 18, kGpsJC = 27, kGpsFA = 36, kGpsFC = 45, kGpsFD = 54, kGpsZ = 63,
              kGpsE = 64, kGpsS = 76;

// Cooperative row store: every lane of a wave has deposited one row (W doubles, W even) of ITS factor in the wave's
// LDS staging buffer; the wave then writes the 64 rows as 16-byte pieces, consecutive lanes on consecutive pieces of
// the same row, so that each store instruction covers whole 64-byte sectors instead of 64 scattered 16-byte
// fragments (per-lane row stores are store-issue bound and write 2.4x the bytes to HBM: profiles/round1_v2).
// srow[l] = first row of lane l's factor in the row table, or -1.
// wave_store_part: the lanes staged HW (even) doubles each with stride HW + 2; they land at columns
// [coloff, coloff + HW) of row (first row of the lane's factor) + roff of a table with W doubles per row.
template <typename T, int W, int HW, int LS = HW + 2, bool STREAM = false>
__device__ __forceinline__ void wave_store_part(const T *st, const int *srow, int lane, int roff, int coloff, T *table) {
  constexpr int P = HW / 2;
  typedef T V2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int t = 0; t < P; t++) {
    const int q = t * 64 + lane;
    const int fl = q / P, piece = q - fl * P;
    const int r0 = srow[fl];
    if (r0 >= 0) {
      const V2 v = *reinterpret_cast<const V2 *>(st + fl * LS + piece * 2);
      V2 *dst = reinterpret_cast<V2 *>(table + (size_t)(r0 + roff) * W + coloff + piece * 2);
      // STREAM (the structured records: written once in whole 128-byte lines, read once by the next launch): 1e6 Pose3 states
      // K1 0.376 -> 0.337 ms, 1e5: the iteration -5..-12 us.  Not for row tables: their 16-byte fragments of 96-byte half rows
      // had DOUBLED K1 with this hint (round 2); the factor records of k_fused_level0 and the solutions of the
      // back-substitution, which the next launch finds in the caches, lose 3 and 20 us of the iteration with it.
      if constexpr (STREAM) __builtin_nontemporal_store(v, dst);
      else *dst = v;
    }
  }
}
template <typename T, int W>
__device__ __forceinline__ void wave_store_rows(const T *st, const int *srow, int lane, int roff, T *table) {
  wave_store_part<T, W, W>(st, srow, lane, roff, 0, table);
}
// HW consecutive scalars per lane (stride HW + 2) to table[first row + k]: 8-byte pieces (the first row may be odd)
template <typename T, int HW>
__device__ __forceinline__ void wave_store_scalars(const T *st, const int *srow, int lane, T *table) {
  constexpr int LS = HW + 2;
#pragma unroll
  for (int t = 0; t < HW; t++) {
    const int q = t * 64 + lane;
    const int fl = q / HW, k = q - fl * HW;
    const int r0 = srow[fl];
    if (r0 >= 0) table[r0 + k] = st[fl * LS + k];
  }
}

// Row rho of U * M, U upper triangular d x d (row-major, d = 6), M block lower-triangular [[A, 0], [C, D]]:
// only the structurally non-zero products are formed.
template <typename T>
__device__ __forceinline__ void utri_times_bl6_row(const T *U, const BL6<T> &M, int rho, T *out) {
#pragma unroll
  for (int c = 0; c < 6; c++) {
    T acc = T(0);
#pragma unroll
    for (int r = 0; r < 6; r++) {
      if (r < rho) continue;
      if (r < 3 && c >= 3) continue;
      acc += U[rho * 6 + r] * bl6_at(M, r, c);
    }
    out[c] = acc;
  }
}

// K1 for GaussianProcessPriorPose3 (GaussianProcessPriorPose3.h:60-98), whitened rows straight into the row table.
// The 12 x 24 Jacobian is never held as a whole: with Jinv = Jr^-1(r), FD = d(Jinv v2)/dr, the right half
// [H3 H4] = [[Jinv, 0], [FD Jinv, Jinv]] is whitened and stored first, then the left half
// [H1 H2] = [[J, -dt I], [FD J, -I]], J = -Jinv Ad(h^-1); whitening R = [[sa U, sb U], [0, sc U]] is applied as
// U x (block lower-triangular) products row by row, skipping the structural zeros.  Peak live state is FD + two
// 6x6 blocks instead of two 6 x 24 arrays, which is what lets two waves share a SIMD.
// K1 for a chain whose GP priors reach the fused level-0 kernel as structured records (kGps* above): Log, Jr^-1, the
// finite-difference block and J = -Jr^-1 Ad(h^-1) per factor, the whitened error -- and none of the products that turn them
// into Jacobian columns (those moved to the consumer, where they are DPP row operations).
#ifdef GPS_TRACE_KLIN
// debug builds only (scripts/trace_klin.py): s_memrealtime stamps of the K1 waves -- 16 per wave, wave = f / 64
static __device__ unsigned long long g_klin_trace[4096 * 16];
#define KL_TR(slot) do { if (lane == 0 && (f >> 6) < 4096) g_klin_trace[(size_t)(f >> 6) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define KL_TR(slot) do { } while (0)
#endif
template <typename T>
__device__ __forceinline__ void gp_pose3_record(const GpArgs<T> &a, bool valid, int f, T *st, int *sr, int lane, T &err) {
  KL_TR(0);
  T *mine = st + lane * 20;
  sr[lane] = valid ? f : -1;                 // structured records are indexed by factor, not by row
  const T *U = a.U.u;
  T p1[12], p2[12], v1[6], v2[6];
  T dt = T(1);
#pragma unroll
  for (int k = 0; k < 12; k++) { p1[k] = (k == 0 || k == 4 || k == 8) ? T(1) : T(0); p2[k] = p1[k]; }   // idle lane: identity
#pragma unroll
  for (int k = 0; k < 6; k++) { v1[k] = T(0); v2[k] = T(0); }
  if (valid) {
    const int i = a.left[f];
    dt = a.dt[f];
    if constexpr (std::is_same<T, double>::value) {
      // (with an update pending this thread owns state i, and the chain's last state if that is its right one)
      load_two_states_upd<POSE3>(a.pose, a.vel, a.stride, i, a.pend, true, i + 1 == a.pend.last, p1, v1, p2, v2);
    } else {
#pragma unroll
      for (int k = 0; k < 12; k++) { p1[k] = T(a.pose[(size_t)k * a.stride + i]); p2[k] = T(a.pose[(size_t)k * a.stride + i + 1]); }
#pragma unroll
      for (int k = 0; k < 6; k++) { v1[k] = a.vel[(size_t)k * a.stride + i]; v2[k] = a.vel[(size_t)k * a.stride + i + 1]; }
    }
  }
  KL_TR(1);
  const SE3<T> h = se3_between(as_se3(p1), as_se3(p2));
  const V6<T> r = se3_log(h);                 // GaussianProcessPriorPose3.h:72
#ifdef GPS_TRACE_KLIN
  { V6<T> rr_ = r; pin(rr_); }
  KL_TR(2);
#endif
  const JrK<T> k0 = jr_coefs(r.w);
  const BL6<T> Jinv = se3_jrinv_k(k0, r);     // :76
  const V6<T> u1 = as_v6(v1), u2 = as_v6(v2);
  const T sq = sqrt(dt);
  const T sa = T(3.4641016151377545870548926830117) / (dt * sq);  // sqrt(12 / dt^3)
  const T sb = T(-1.7320508075688772935274463415059) / sq;        // (-6 / dt^2) / sa
  const T sc = T(1) / sq;                                          // sqrt(4/dt - sb^2)
  {   // the whitened error and the scalars: line 4 of the record
    const V6<T> top = r - dt * u1;            // :97
    const V6<T> bot = Jinv * u2 - u1;
    const T e[12] = {top.w.x, top.w.y, top.w.z, top.v.x, top.v.y, top.v.z, bot.w.x, bot.w.y, bot.w.z, bot.v.x, bot.v.y, bot.v.z};
#pragma unroll
    for (int rho = 0; rho < 6; rho++) {
      T wt = T(0), wb = T(0);
#pragma unroll
      for (int q = rho; q < 6; q++) {
        wt += U[rho * 6 + q] * (sa * e[q] + sb * e[6 + q]);
        wb += U[rho * 6 + q] * e[6 + q];
      }
      wb *= sc;
      err += wt * wt + wb * wb;
      mine[rho] = wt;
      mine[6 + rho] = wb;
    }
    mine[12] = -(sa * dt + sb); mine[13] = sb; mine[14] = sc; mine[15] = sa;
    KL_TR(3);
    wave_store_part<T, kGpsLen, 16, 20, true>(st, sr, lane, 0, kGpsE, a.gps);
    KL_TR(4);
  }
  // the matrix blocks in record order, 16 doubles (one 128-byte line) staged per lane at a time
  auto put = [&](int idx, T v) {              // (idx is a compile-time constant once the loops below are unrolled)
    mine[idx & 15] = v;
    if ((idx & 15) == 15) wave_store_part<T, kGpsLen, 16, 20, true>(st, sr, lane, 0, idx - 15, a.gps);
  };
#pragma unroll
  for (int k = 0; k < 9; k++) put(kGpsXA + k, Jinv.A.m[k]);
#pragma unroll
  for (int k = 0; k < 9; k++) put(kGpsXC + k, Jinv.C.m[k]);
  {
    const SE3<T> hi = se3_inverse(h);
    // J = -Jinv Ad(h^-1), Ad = [[R, 0], [t^ R, R]]: the diagonal blocks of J are equal (those of both factors are)
    const M3<T> JA = neg(Jinv.A * hi.R);
    const M3<T> JC = neg(Jinv.C * hi.R + Jinv.A * (skew(hi.t) * hi.R));
#pragma unroll
    for (int k = 0; k < 9; k++) put(kGpsJA + k, JA.m[k]);
#pragma unroll
    for (int k = 0; k < 9; k++) put(kGpsJC + k, JC.m[k]);
  }
  KL_TR(5);
  {   // last, with nothing but r, the coefficients and v2 alive next to it: the twelve evaluations of the difference quotient
    // (the factor's error sum waits in a spare staging slot of the lane: across this block it was what the allocator spilled)
    mine[16] = err;
    BL6<T> FD = se3_jrinv_times_x_fd_k(k0, r, u2);   // (:81, :90)
#ifdef GPS_TRACE_KLIN
    pin(FD.A); pin(FD.C); pin(FD.D);
    KL_TR(6);
#endif
#pragma unroll
    for (int k = 0; k < 9; k++) put(kGpsFA + k, FD.A.m[k]);
#pragma unroll
    for (int k = 0; k < 9; k++) put(kGpsFC + k, FD.C.m[k]);
#pragma unroll
    for (int k = 0; k < 9; k++) put(kGpsFD + k, FD.D.m[k]);
    put(kGpsZ, T(0));
    err = mine[16];
  }
  KL_TR(7);
}

template <typename T, bool VW>
__device__ __forceinline__ void gp_pose3_rows(const GpArgs<T> &a, bool valid, int f, T *st, int *sr, int lane, T &err) {
  constexpr int LS = 14;
  T *mine = st + lane * LS;
  const T *U = a.U.u;
  T p1[12], p2[12], v1[6], v2[6];
  T dt = T(1);
#pragma unroll
  for (int k = 0; k < 12; k++) { p1[k] = (k == 0 || k == 4 || k == 8) ? T(1) : T(0); p2[k] = p1[k]; }   // idle lane: identity
#pragma unroll
  for (int k = 0; k < 6; k++) { v1[k] = T(0); v2[k] = T(0); }
  if (valid) {
    const int i = a.left[f];
    dt = a.dt[f];
    double q1[12], q2[12];
#pragma unroll
    for (int k = 0; k < 12; k++) { q1[k] = a.pose[(size_t)k * a.stride + i]; q2[k] = a.pose[(size_t)k * a.stride + i + 1]; }
    if (!IsF64<T>::v) {
#pragma unroll
      for (int k = 9; k < 12; k++) { q2[k] -= q1[k]; q1[k] = 0.0; }
    }
#pragma unroll
    for (int k = 0; k < 12; k++) { p1[k] = T(q1[k]); p2[k] = T(q2[k]); }
#pragma unroll
    for (int k = 0; k < 6; k++) { v1[k] = a.vel[(size_t)k * a.stride + i]; v2[k] = a.vel[(size_t)k * a.stride + i + 1]; }
  }
  if (VW) {   // GaussianProcessPriorPose3VW.h:87-88: body velocities of the world-frame (v, w)
    T b1[6], b2[6];
    vw_to_vb(p1, v1, b1);
    vw_to_vb(p2, v2, b2);
#pragma unroll
    for (int k = 0; k < 6; k++) { v1[k] = b1[k]; v2[k] = b2[k]; }
  }
  const SE3<T> h = se3_between(as_se3(p1), as_se3(p2));
  const V6<T> r = se3_log(h);                 // GaussianProcessPriorPose3.h:72
  const JrK<T> k0 = jr_coefs(r.w);
  const BL6<T> Jinv = se3_jrinv_k(k0, r);     // :76
  const V6<T> u1 = as_v6(v1), u2 = as_v6(v2);
  const T sq = sqrt(dt);
  const T sa = T(3.4641016151377545870548926830117) / (dt * sq);  // sqrt(12 / dt^3)
  const T sb = T(-1.7320508075688772935274463415059) / sq;        // (-6 / dt^2) / sa
  const T sc = T(1) / sq;                                          // sqrt(4/dt - sb^2)
  {
    const V6<T> top = r - dt * u1;            // :97
    const V6<T> bot = Jinv * u2 - u1;
    const T e[12] = {top.w.x, top.w.y, top.w.z, top.v.x, top.v.y, top.v.z, bot.w.x, bot.w.y, bot.w.z, bot.v.x, bot.v.y, bot.v.z};
#pragma unroll
    for (int rho = 0; rho < 6; rho++) {
      T wt = T(0), wb = T(0);
#pragma unroll
      for (int q = rho; q < 6; q++) {
        wt += U[rho * 6 + q] * (sa * e[q] + sb * e[6 + q]);
        wb += U[rho * 6 + q] * e[6 + q];
      }
      wb *= sc;
      err += wt * wt + wb * wb;
      mine[rho] = wt;
      mine[6 + rho] = wb;
    }
    wave_store_scalars<T, 12>(st, sr, lane, a.rowE);
  }
  const BL6<T> FD = se3_jrinv_times_x_fd_k(k0, r, u2);   // (:81, :90) -- computed once, used twice
  {   // right state: H3 = [Jinv; FD Jinv] (:88-93), H4 = [0; Jinv] (:95)
    const BL6<T> P3 = FD * Jinv;
#pragma unroll
    for (int rho = 0; rho < 6; rho++) {
      T x[6], y[6];
      utri_times_bl6_row(U, Jinv, rho, x);
      utri_times_bl6_row(U, P3, rho, y);
#pragma unroll
      for (int c = 0; c < 6; c++) { mine[c] = sa * x[c] + sb * y[c]; mine[6 + c] = sb * x[c]; }
      if (VW) vw_row_transform(p2, v2, mine);
      wave_store_part<T, 24, 12>(st, sr, lane, rho, 12, a.rowLR);
#pragma unroll
      for (int c = 0; c < 6; c++) { mine[c] = sc * y[c]; mine[6 + c] = sc * x[c]; }
      if (VW) vw_row_transform(p2, v2, mine);
      wave_store_part<T, 24, 12>(st, sr, lane, 6 + rho, 12, a.rowLR);
    }
  }
  {   // left state: H1 = [J; FD J], J = Hlog Hcomp1 Hinv = -Jinv Ad(h^-1) (:79-84), H2 = [-dt I; -I] (:86)
    const BL6<T> J = neg(Jinv * se3_adjoint(se3_inverse(h)));
    const BL6<T> P1 = FD * J;
    const T k2 = -(sa * dt + sb);
#pragma unroll
    for (int rho = 0; rho < 6; rho++) {
      T x[6], y[6];
      utri_times_bl6_row(U, J, rho, x);
      utri_times_bl6_row(U, P1, rho, y);
#pragma unroll
      for (int c = 0; c < 6; c++) { mine[c] = sa * x[c] + sb * y[c]; mine[6 + c] = (c >= rho) ? k2 * U[rho * 6 + c] : T(0); }
      if (VW) vw_row_transform(p1, v1, mine);
      wave_store_part<T, 24, 12>(st, sr, lane, rho, 0, a.rowLR);
#pragma unroll
      for (int c = 0; c < 6; c++) { mine[c] = sc * y[c]; mine[6 + c] = (c >= rho) ? -sc * U[rho * 6 + c] : T(0); }
      if (VW) vw_row_transform(p1, v1, mine);
      wave_store_part<T, 24, 12>(st, sr, lane, 6 + rho, 0, a.rowLR);
    }
  }
}

// One 128-thread block of GP-prior factors; bid = block index among the GP blocks.  stage / srow: the block's LDS staging
// area (2 * 64 * (2b + 2) elements of T and 128 ints when MODE == 0), owned by the calling kernel so that k_lin can share
// one area between its factor types.
// REC (Pose3 body-velocity chains, fp64, MODE 0): structured records for k_fused_level0 (a.gps) instead of rows -- a template
// parameter, not a run-time branch: the two bodies in one kernel cost both their register budget (256 VGPRs + 368 B of scratch)
template <typename T, int MF, int MODE, bool VW, bool REC = false>
__device__ __forceinline__ void gp_block(const GpArgs<T> &a, const int bid, T *stage, int *srow) {
  static_assert(!REC || (MODE == 0 && !VW && IsF64<T>::v && (MF == POSE3 || MF == POSE2 || MF == ROT3 || MF == LINEAR3)),
                "structured GP records: fp64, SE(3) body-velocity chains and the d = 3 manifolds");
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd, b = 2 * d;
  constexpr bool JAC = (MODE != 1);
  constexpr int LS = 2 * b + 2;                       // staging stride (16-byte aligned, conflict-free for b128)
  const int f = bid * 128 + threadIdx.x;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool valid = f < a.count;
  T err = T(0);
  if constexpr (MF == POSE3 && MODE == 0) {
    srow[threadIdx.x] = valid ? a.row0[f] : -1;
    if constexpr (REC) gp_pose3_record<T>(a, valid, f, stage + wv * 64 * 20 /* 16 staged doubles per lane, stride 20 */, srow + wv * 64, lane, err);
    else gp_pose3_rows<T, VW>(a, valid, f, stage + wv * 64 * 20 /* half rows: 14 per lane */, srow + wv * 64, lane, err);
    const T tot = block_sum(T(0.5) * err);
    if (threadIdx.x == 0) a.partial[bid] = tot;
    return;
  }
  T e[b];
  T Jt[JAC ? d * 2 * b : 1], Jb[JAC ? d * 2 * b : 1];
  T dt = T(1);
  if (valid) {
    const int i = a.left[f];
    dt = a.dt[f];
    T p1[pd], p2[pd], v1[d], v2[d];
    if constexpr (REC && d == 3 && IsF64<T>::v) {
      // the d = 3 record launches inside run_gn: the states as the pending update leaves them (PendUpd; nothing pending: as stored);
      // this factor owns its left state, the last factor the chain's last state as well
      double q1[pd], q2[pd], w1[d], w2[d];
      load_two_states_upd<MF>(a.pose, a.vel, a.stride, i, a.pend, true, i + 1 == a.pend.last, q1, w1, q2, w2);
#pragma unroll
      for (int k = 0; k < pd; k++) { p1[k] = T(q1[k]); p2[k] = T(q2[k]); }
#pragma unroll
      for (int k = 0; k < d; k++) { v1[k] = T(w1[k]); v2[k] = T(w2[k]); }
    } else {
    {
      double q1[pd], q2[pd];
#pragma unroll
      for (int k = 0; k < pd; k++) { q1[k] = a.pose[(size_t)k * a.stride + i]; q2[k] = a.pose[(size_t)k * a.stride + i + 1]; }
      if (!IsF64<T>::v) {
#pragma unroll
        for (int k = TransPart<MF>::off; k < TransPart<MF>::off + TransPart<MF>::n; k++) { q2[k] -= q1[k]; q1[k] = 0.0; }
      }
#pragma unroll
      for (int k = 0; k < pd; k++) { p1[k] = T(q1[k]); p2[k] = T(q2[k]); }
    }
#pragma unroll
    for (int k = 0; k < d; k++) { v1[k] = a.vel[(size_t)k * a.stride + i]; v2[k] = a.vel[(size_t)k * a.stride + i + 1]; }
    }
    if constexpr (MF == POSE3) {
      if (a.vw) {
        T b1[6], b2[6];
        vw_to_vb(p1, v1, b1);
        vw_to_vb(p2, v2, b2);
#pragma unroll
        for (int k = 0; k < 6; k++) { v1[k] = b1[k]; v2[k] = b2[k]; }
      }
    }
    GpPrior<T, MF, JAC>::eval(p1, v1, p2, v2, dt, e, Jt, Jb);
    if constexpr (MF == POSE3 && JAC) {
      if (a.vw) {
#pragma unroll
        for (int r = 0; r < d; r++) {
          vw_row_transform(p1, v1, Jt + r * 2 * b); vw_row_transform(p2, v2, Jt + r * 2 * b + b);
          vw_row_transform(p1, v1, Jb + r * 2 * b); vw_row_transform(p2, v2, Jb + r * 2 * b + b);
        }
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < b; k++) e[k] = T(0);
    if (JAC) {
#pragma unroll
      for (int k = 0; k < d * 2 * b; k++) { Jt[k] = T(0); Jb[k] = T(0); }
    }
  }
  if (MODE == 2) {
    if (valid) {
#pragma unroll
      for (int k = 0; k < b; k++) a.out_e[(size_t)f * b + k] = e[k];
      if (a.out_H) {
        T *H = a.out_H + (size_t)f * 4 * b * d;
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
          for (int r = 0; r < d; r++)
#pragma unroll
            for (int c = 0; c < d; c++) {
              H[(q * b + r) * d + c] = Jt[r * 2 * b + q * d + c];
              H[(q * b + d + r) * d + c] = Jb[r * 2 * b + q * d + c];
            }
      }
    }
  } else {
    // whitening R = chol_upper(Q^-1(dt)) = [[sa, sb], [0, sc]] (x) U, U = chol_upper(Qc^-1)
    // (noiseModel::Gaussian::Covariance(calcQ(Qc, dt)), GaussianProcessPriorPose3.h:46)
    const T sq = sqrt(dt);
    const T sa = T(3.4641016151377545870548926830117) / (dt * sq);  // sqrt(12 / dt^3)
    const T sb = T(-1.7320508075688772935274463415059) / sq;        // (-6 / dt^2) / sa
    const T sc = T(1) / sq;                                          // sqrt(4/dt - sb^2)
    if constexpr (REC && d == 3) {
      // the d = 3 record (kGp3*): 16 doubles staged per lane at a time (stride 20), whole 128-byte lines out
      T *st = stage + wv * 64 * 20, *mine = st + lane * 20;
      int *sr = srow + wv * 64;
      sr[lane] = valid ? f : -1;
      auto put = [&](int idx, T v) {            // (idx: a compile-time constant once the loops are unrolled)
        mine[idx & 15] = v;
        if ((idx & 15) == 15) wave_store_part<T, kGp3Len, 16, 20, true>(st, sr, lane, 0, idx - 15, a.gps);
      };
#pragma unroll
      for (int blk = 0; blk < 2; blk++)         // A1 = U (sa J1 + sb .): columns 0..2 of the top rows; A3: columns 6..8
#pragma unroll
        for (int rho = 0; rho < 3; rho++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            T v = T(0);
#pragma unroll
            for (int r = rho; r < 3; r++) v += a.U.u[rho * 3 + r] * (sa * Jt[r * 12 + 6 * blk + c] + sb * Jb[r * 12 + 6 * blk + c]);
            put((blk ? kGp3A3 : kGp3A1) + rho * 3 + c, v);
          }
#pragma unroll
      for (int rho = 0; rho < 3; rho++) {
        T wt = T(0), wb = T(0);
#pragma unroll
        for (int r = rho; r < 3; r++) {
          wt += a.U.u[rho * 3 + r] * (sa * e[r] + sb * e[3 + r]);
          wb += a.U.u[rho * 3 + r] * e[3 + r];
        }
        wb *= sc;
        err += wt * wt + wb * wb;
        put(kGp3E + rho, wt);
        put(kGp3E + 3 + rho, wb);               // (positions 21..23 of the record: written out of order, both inside the second line)
      }
      put(kGp3S + 0, sa * Jt[3] + sb * Jb[3]);  // kLt: the (0, 0) entry of H2's halves
      put(kGp3S + 1, sa * Jt[9] + sb * Jb[9]);  // kRt: H4's
      put(kGp3S + 2, sc * Jb[3]);               // kLb
      put(kGp3S + 3, sc * Jb[9]);               // kRb
      put(28, T(0)); put(29, T(0)); put(30, T(0));
      put(31, T(0));
      const T tot = block_sum(T(0.5) * err);
      if (threadIdx.x == 0) a.partial[bid] = tot;
      return;
    }
    T *st = stage + (MODE == 0 ? wv * 64 * LS : 0);
    T *mine = st + (MODE == 0 ? lane * LS : 0);
    const int *sr = srow + (MODE == 0 ? wv * 64 : 0);
    int row0 = 0;
    if (MODE == 0) {
      row0 = valid ? a.row0[f] : -1;
      srow[threadIdx.x] = row0;
    }
    if (MODE == 1 && a.rowE32 && valid) row0 = a.row0[f];
#pragma unroll
    for (int rho = 0; rho < d; rho++) {
      T wt = T(0), wb = T(0);
#pragma unroll
      for (int r = rho; r < d; r++) {
        wt += a.U.u[rho * d + r] * (sa * e[r] + sb * e[d + r]);
        wb += a.U.u[rho * d + r] * e[d + r];
      }
      wb *= sc;
      err += wt * wt + wb * wb;
      if (MODE == 1 && a.rowE32 && valid) {
        a.rowE32[row0 + rho] = (float)wt;
        a.rowE32[row0 + d + rho] = (float)wb;
      }
      if (MODE == 0) {
        if (valid) {
          a.rowE[row0 + rho] = wt;
          a.rowE[row0 + d + rho] = wb;
        }
#pragma unroll
        for (int col = 0; col < 2 * b; col++) {
          T vt = T(0);
#pragma unroll
          for (int r = rho; r < d; r++) vt += a.U.u[rho * d + r] * (sa * Jt[r * 2 * b + col] + sb * Jb[r * 2 * b + col]);
          mine[col] = vt;
        }
        wave_store_rows<T, 2 * b>(st, sr, lane, rho, a.rowLR);
#pragma unroll
        for (int col = 0; col < 2 * b; col++) {
          T vb = T(0);
#pragma unroll
          for (int r = rho; r < d; r++) vb += a.U.u[rho * d + r] * Jb[r * 2 * b + col];
          mine[col] = sc * vb;
        }
        wave_store_rows<T, 2 * b>(st, sr, lane, d + rho, a.rowLR);
      }
    }
  }
  if (MODE != 2) {
    const T tot = block_sum(T(0.5) * err);
    if (threadIdx.x == 0) a.partial[bid] = tot;
  }
}

// MODE 0: whitened rows + error; MODE 1: error only; MODE 2: unwhitened e + H1..H4 in API layout
template <typename T, int MF, int MODE, bool VW = false, bool REC = false>
__global__ void __launch_bounds__(128) k_gp(GpArgs<T> a) {
  constexpr int LS = REC ? 20 : 4 * MTraits<MF>::d + 2;
  __shared__ T stage[MODE == 0 ? 2 * 64 * LS : 1];
  __shared__ int srow[MODE == 0 ? 128 : 1];
  gp_block<T, MF, MODE, VW, REC>(a, blockIdx.x, stage, srow);
}

// ------------------------------------------------------------------ unary / between rows

template <typename T> struct FacArgs {
  PendUpd pend;         // see GpArgs: these factors read the updated states too (they own none)
  const double *pose, *vel;
  int stride, count, chart;
  const int *idx;       // state (left state for between)
  const double *meas;   // count x (pd or d)
  const double *sig;    // count x d
  const int *row0;
  float *rowE32;        // error-only pass: fp32 copy of the whitened error (fp32 mode), or null
  T *rowLR, *rowE;
  T *partial;
  T *rec;               // KIND 2, Pose3, fp64: structured records (kBtw*) instead of compact rows, or null
};

// BetweenFactor<Pose3> as a structured record (kBtw*): the block-triangular halves of H1 and H2, the weights, the whitened error
template <typename T>
__device__ __forceinline__ void between_pose3_record(const FacArgs<T> &a, const int bid, T *stage, int *srow) {
  const int f = bid * 128 + threadIdx.x;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool valid = f < a.count;
  T *st = stage + wv * 64 * 20, *mine = st + lane * 20;
  int *sr = srow + wv * 64;
  sr[lane] = valid ? f : -1;
  T x1[12], x2[12], m[12];
#pragma unroll
  for (int k = 0; k < 12; k++) { x1[k] = (k == 0 || k == 4 || k == 8) ? T(1) : T(0); x2[k] = x1[k]; m[k] = x1[k]; }   // idle lane: identities
  if (valid) {
    const int i = a.idx[f];
    if constexpr (std::is_same<T, double>::value) {
      double vv[6];
      double vv2[6];
      load_two_states_upd<POSE3>(a.pose, a.vel, a.stride, i, a.pend, false, false, x1, vv, x2, vv2);
#pragma unroll
      for (int k = 0; k < 12; k++) m[k] = T(a.meas[(size_t)f * 12 + k]);
    } else {
#pragma unroll
      for (int k = 0; k < 12; k++) {
        x1[k] = T(a.pose[(size_t)k * a.stride + i]);
        x2[k] = T(a.pose[(size_t)k * a.stride + i + 1]);
        m[k] = T(a.meas[(size_t)f * 12 + k]);
      }
    }
  }
  auto put = [&](int idx, T v) {              // (idx: a compile-time constant once the loops are unrolled)
    mine[idx & 15] = v;
    if ((idx & 15) == 15) wave_store_part<T, kBtwLen, 16, 20, true>(st, sr, lane, 0, idx - 15, a.rec);
  };
  const SE3<T> hx = se3_between(as_se3(x1), as_se3(x2));
  const V6<T> xi = se3_log(se3_between(as_se3(m), hx));          // PoseFactors<T, POSE3>::between
  const BL6<T> HL = se3_jrinv(xi);
#pragma unroll
  for (int k = 0; k < 9; k++) put(kBtwRA + k, HL.A.m[k]);
#pragma unroll
  for (int k = 0; k < 9; k++) put(kBtwRC + k, HL.C.m[k]);
  {
    const SE3<T> hi = se3_inverse(hx);
    const M3<T> LA = neg(HL.A * hi.R);
    const M3<T> LC = neg(HL.C * hi.R + HL.A * (skew(hi.t) * hi.R));
#pragma unroll
    for (int k = 0; k < 9; k++) put(kBtwLA + k, LA.m[k]);
#pragma unroll
    for (int k = 0; k < 9; k++) put(kBtwLC + k, LC.m[k]);
  }
  const T e[6] = {xi.w.x, xi.w.y, xi.w.z, xi.v.x, xi.v.y, xi.v.z};
  T err = T(0), we[6];
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const T w = valid ? T(1) / T(a.sig[(size_t)f * 6 + r]) : T(0);
    we[r] = e[r] * w;
    err += we[r] * we[r];
    put(kBtwW + r, w);
  }
#pragma unroll
  for (int r = 0; r < 6; r++) put(kBtwE + r, we[r]);
  const T tot = block_sum(T(0.5) * err);
  if (threadIdx.x == 0) a.partial[bid] = tot;
}

// KIND 0: PriorFactor<Pose>, 1: PriorFactor<Vector> on the velocity, 2: BetweenFactor<Pose>(x_i, x_i+1)
template <typename T, int MF, int KIND, bool JAC>
__device__ __forceinline__ void simple_block(const FacArgs<T> &a, const int bid, T *stage, int *srow) {
  if constexpr (KIND == 2 && MF == POSE3 && JAC && IsF64<T>::v) {
    if (a.rec != nullptr) { between_pose3_record<T>(a, bid, stage, srow); return; }
  }
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd, b = 2 * d;
  // PriorFactor<Pose> and BetweenFactor<Pose> have no velocity columns: they go to the compact row table,
  // [d/dpose_left (d) | d/dpose_right (d)] per row, which halves what K3 has to read for them
  constexpr int W = (KIND == 1) ? 2 * b : 2 * d;
  constexpr int LS = W + 2;
  const int f = bid * 128 + threadIdx.x;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool valid = f < a.count;
  T err = T(0);
  T e[d], H1[JAC ? d * d : 1], H2[JAC ? d * d : 1];
#pragma unroll
  for (int k = 0; k < d; k++) e[k] = T(0);
  if (JAC) {
#pragma unroll
    for (int k = 0; k < d * d; k++) { H1[k] = T(0); H2[k] = T(0); }
  }
  if (valid) {
    const int i = a.idx[f];
    if (KIND == 1) {
      double pp[pd], vv[d];
      load_state_upd<MF>(a.pose, a.vel, a.stride, i, a.pend, false, pp, vv);
#pragma unroll
      for (int k = 0; k < d; k++) e[k] = T(vv[k] - a.meas[(size_t)f * d + k]);
    } else {
      T x1[pd], m[pd], x2[pd];
      {
        double q1[pd], qm[pd], q2[pd], vv[d];
        load_state_upd<MF>(a.pose, a.vel, a.stride, i, a.pend, false, q1, vv);
        if (KIND == 2) load_state_upd<MF>(a.pose, a.vel, a.stride, i + 1, a.pend, false, q2, vv);
#pragma unroll
        for (int k = 0; k < pd; k++) {
          qm[k] = a.meas[(size_t)f * pd + k];
          if (KIND != 2) q2[k] = 0.0;
        }
        if (!IsF64<T>::v) {   // re-centre on x1: PriorFactor compares (prior, x1), BetweenFactor (x1, x2) with a relative measurement
#pragma unroll
          for (int k = TransPart<MF>::off; k < TransPart<MF>::off + TransPart<MF>::n; k++) {
            if (KIND == 0) qm[k] -= q1[k]; else q2[k] -= q1[k];
            q1[k] = 0.0;
          }
        }
#pragma unroll
        for (int k = 0; k < pd; k++) { x1[k] = T(q1[k]); m[k] = T(qm[k]); x2[k] = T(q2[k]); }
      }
      if (KIND == 0) {
        PoseFactors<T, MF, JAC>::prior(m, x1, a.chart, e, H1);
      } else {
        PoseFactors<T, MF, JAC>::between(m, x1, x2, a.chart, e, H1, H2);
      }
    }
  }
  const int row0 = ((JAC || a.rowE32) && valid) ? a.row0[f] : -1;
  T *st = stage + (JAC ? wv * 64 * LS : 0);
  T *mine = st + (JAC ? lane * LS : 0);
  if (JAC) srow[threadIdx.x] = row0;
#pragma unroll
  for (int r = 0; r < d; r++) {
    const T w = valid ? T(1) / T(a.sig[(size_t)f * d + r]) : T(0);
    const T we = e[r] * w;
    err += we * we;
    if (!JAC && a.rowE32 && valid) a.rowE32[row0 + r] = (float)we;
    if (JAC) {
      if (valid) a.rowE[row0 + r] = we;
#pragma unroll
      for (int c = 0; c < W; c++) mine[c] = T(0);
      if (KIND == 1) {
        mine[d + r] = w;
      } else {
#pragma unroll
        for (int c = 0; c < d; c++) mine[c] = w * H1[r * d + c];
        if (KIND == 2) {
#pragma unroll
          for (int c = 0; c < d; c++) mine[d + c] = w * H2[r * d + c];
        }
      }
      wave_store_rows<T, W>(st, srow + wv * 64, lane, r, a.rowLR);
    }
  }
  const T tot = block_sum(T(0.5) * err);
  if (threadIdx.x == 0) a.partial[bid] = tot;
}

template <typename T, int MF, int KIND, bool JAC>
__global__ void __launch_bounds__(128) k_simple(FacArgs<T> a) {
  constexpr int LS = ((KIND == 1) ? 4 * MTraits<MF>::d : 2 * MTraits<MF>::d) + 2;
  __shared__ T stage[JAC ? 2 * 64 * LS : 1];
  __shared__ int srow[JAC ? 128 : 1];
  simple_block<T, MF, KIND, JAC>(a, blockIdx.x, stage, srow);
}

// K1 of a Gauss-Newton iteration in ONE launch: the GP priors, the pose / velocity priors and the between factors share the
// grid, the register-heavy GP blocks first and the light streaming blocks behind them, which fill the issue slots and the
// store bandwidth the GP waves leave (measured on 1e5 Pose3 states: 0.539 ms per iteration; the same kernels on two streams
// with an event fork / join 0.556; GP and between blocks alternating 0.568).
template <typename T> struct LinArgs {
  GpArgs<T> gp;
  FacArgs<T> fac[3];      // pose priors, velocity priors, between factors
  int nb_gp, nb[3];       // blocks per type (0: type absent)
  // deferred reduction (round 3): the per-block |delta|_inf maxima the PREVIOUS iteration's k_retract left behind are
  // reduced here, by one extra workgroup at the end of the grid, instead of by a launch of their own (fixed order:
  // deterministic; nothing is skipped, the value lands in *red_out one launch later).  red_in == null: nothing pending.
  const double *red_in = nullptr;
  int red_n = 0;
  double *red_out = nullptr;
};
// VP: the launch contains velocity priors (full-width rows).  Without them a Pose3 launch stages half rows only (the GP
// prior writes its rows in halves, pose priors / between factors have compact rows): 14 KB of LDS instead of 27 KB, so that
// every workgroup of a 1e5-state launch is resident at once.
template <typename T, int MF, bool VW, bool VP, bool REC = false>
__global__ void __launch_bounds__(128, GPS_KLIN_WAVES) k_lin(LinArgs<T> a) {
  constexpr int LS = (MF == POSE3 && !VP) ? 20 : (4 * MTraits<MF>::d + 2 > 20 || !REC ? 4 * MTraits<MF>::d + 2 : 20);
  __shared__ T stage[2 * 64 * LS];
  __shared__ int srow[128];
  int bid = blockIdx.x;
  if (a.red_in != nullptr && bid == (int)gridDim.x - 1) {     // the extra workgroup: pending maximum of the previous retraction
    const double acc = strided_fold<double, 128>(a.red_in, a.red_n, 1);
    const double r = block_max(acc);
    if (threadIdx.x == 0) *a.red_out = r;
    return;
  }
  // GP blocks first, then the light ones (measured in round 2: 0.539 ms per iteration; GP and between blocks alternating 0.568)
  if (bid < a.nb_gp) { gp_block<T, MF, 0, VW, REC>(a.gp, bid, stage, srow); return; }
  bid -= a.nb_gp;
  if (bid < a.nb[2]) { simple_block<T, MF, 2, true>(a.fac[2], bid, stage, srow); return; }
  bid -= a.nb[2];
  if (bid < a.nb[0]) { simple_block<T, MF, 0, true>(a.fac[0], bid, stage, srow); return; }
  bid -= a.nb[0];
  if constexpr (VP) simple_block<T, MF, 1, true>(a.fac[1], bid, stage, srow);
}

// ------------------------------------------------------------------ K2: measurement factors

enum FKind : int { FK_INTERP_RANGE = 0, FK_RANGE = 1, FK_INTERP_ATT = 2, FK_INTERP_GPS = 3, FK_ODOM2D = 4, FK_BEARING_RANGE = 5, FK_INTERP_PROJ = 6, FK_AHRS = 7 };
constexpr int kNumMeasKinds = 8;
constexpr int kAhrsWidth = 34;   // per-factor parameters of FK_AHRS: the 25 of ahrs_factor() + sqrt information R (3 x 3, upper triangular)
constexpr int kMeasAux = 22;     // [body_P_sensor (12) | fx, fy, s, u0, v0 | has_sensor | k1, k2, p1, p2 (Cal3DS2; zeros: Cal3_S2)]
template <int FK> struct FKRows { static constexpr int rows = (FK == FK_INTERP_RANGE || FK == FK_RANGE) ? 1 : ((FK == FK_INTERP_ATT || FK == FK_BEARING_RANGE || FK == FK_INTERP_PROJ) ? 2 : 3); };

template <typename T> struct MeasArgs {
  PendUpd pend;        // d = 3 chains inside run_gn: the update k_lin is applying beside this launch (pose / vel: the buffer it READS)
  const double *pose, *vel;
  int stride;
  const double *lmk;   // L x ld (AoS)
  int ld, count, chart;
  const int *idx;      // left (or only) state
  const int *lm;       // landmark or null
  const double *meas;  // count x mw
  int mw;
  const double *sig;   // count x rows
  const double *coef;  // count x 4: l11, l12, p11, p12 (interpolated kinds)
  float *rowE32;       // error-only pass: fp32 copy of the whitened error (fp32 mode), or null
  T *out_e, *out_J;    // inspection (gpslam_hip_linearize_meas): unwhitened e (count x rows) and per row
                       // [H1 H2 | H3 H4 | H5 (3, zero padded)] exactly as evaluateError returns them, or null
  const double *aux;       // table of kMeasAux-wide entries [body_P_sensor (12) | fx, fy, s, u0, v0 | has_sensor | k1, k2, p1, p2]
  const int *aidx;     // count: entry of each factor (one body_P_sensor / calibration PER FACTOR, as in the reference:
                       // GPInterpolatedRangeFactorPose3.h:46-54), or null: no sensor transform anywhere
  const double *sqi;   // count x rows x rows square-root information R (upper triangular, R^T R = cov^-1) of factors with a
                       // noiseModel::Gaussian instead of diagonal sigmas (multi-row kinds; `sig` is then ignored), or null
  int vw;              // Pose3 only: velocities are world-frame [v; w]
  // Pose3, Jacobian pass (round 4): the structured records K1 has just written (kGps*) and the record index per left state (n + 2
  // entries, -1: no GP prior on that interval) -- interp_pose3 takes Jinv, Jinv Ad(h^-1) and the finite-difference block of the
  // interval from there instead of forming them again for every measurement factor on it.  Null: it forms them.
  const T *gps = nullptr;
  const int *gpidx = nullptr;
  T *rowI = nullptr;   // k_meas<..., IROW = true>: the table of 16-double interpolated rows (kIRow*); row0 then counts rows of THAT table
  const int *row0;
  T *rowLR, *rowE, *rowM;
  int *rowLm;
  T *partial;
};

// valid (manifold, kind) pairs; everything else is rejected on the host and compiles to an empty kernel
template <int MF, int FK> struct MeasValid {
  static constexpr bool v = ((FK == FK_INTERP_RANGE || FK == FK_RANGE) && (MF == POSE2 || MF == POSE3 || MF == LINEAR3)) ||
                            (FK == FK_INTERP_ATT && (MF == ROT3 || MF == ROT3_BIAS)) || (FK == FK_AHRS && MF == ROT3_BIAS) ||
                            ((FK == FK_INTERP_GPS || FK == FK_INTERP_PROJ) && MF == POSE3) ||
                            ((FK == FK_ODOM2D || FK == FK_BEARING_RANGE) && MF == LINEAR3);
};

template <typename T> __device__ __forceinline__ void put_v3(V3<T> a, T *row) { row[0] = a.x; row[1] = a.y; row[2] = a.z; }
template <typename T> __device__ __forceinline__ void put_v6(V6<T> a, T *row) { put_v3(a.w, row); put_v3(a.v, row + 3); }

// GPInterpolatedRangeFactorPose2/Pose3/2DLinear, RangeFactorPose2 / RangeFactor2DLinear,
// GPInterpolatedAttitudeFactorRot3, GPInterpolatedGPSFactorPose3, OdometryFactor2DLinear, RangeBearingFactor2DLinear
// (gpslam/slam/*.h, see the per-branch citations).  One thread per factor.
// IROW (round 5; SE(3), fp64, GPInterpolatedGPSFactorPose3): the rows leave as 16-double lines [Lp | mu | e, p11, p12, l12] (kIRow*)
// for k_fused_level0<4>, which forms the right halves from mu and the interval's GP record
template <typename T, int MF, int FK, bool JAC, bool IROW = false>
__global__ void __launch_bounds__(128) k_meas(MeasArgs<T> a) {
  static_assert(!IROW || (FK == FK_INTERP_GPS && MF == POSE3 && JAC && std::is_same<T, double>::value), "interpolated rows: SE(3) GPS factors, fp64");
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd, b = 2 * d, rows = FKRows<FK>::rows;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  T err = T(0);
  if constexpr (MeasValid<MF, FK>::v) {
    // whitened Jacobian rows leave through the per-wave staging buffer (wave_store_rows), like the GP prior's
    T JL[JAC ? rows * b : 1], JR[JAC ? rows * b : 1], wgt[rows];
    T ew_[IROW ? rows : 1];            // IROW: the whitened errors, staged with the rows
    int row0v = -1;
#pragma unroll
    for (int r = 0; r < rows; r++) wgt[r] = T(0);
    if constexpr (IROW) {
#pragma unroll
      for (int r = 0; r < rows; r++) ew_[r] = T(0);
    }
    if (f < a.count) {
      const int i = a.idx[f];
      constexpr bool two = (FK == FK_INTERP_RANGE || FK == FK_INTERP_ATT || FK == FK_INTERP_GPS || FK == FK_ODOM2D || FK == FK_INTERP_PROJ || FK == FK_AHRS);
      constexpr bool haslm = (FK == FK_INTERP_RANGE || FK == FK_RANGE || FK == FK_BEARING_RANGE || FK == FK_INTERP_PROJ);
      T p1[pd], v1[d], p2[pd], v2[d];
      double org[3] = {0.0, 0.0, 0.0};      // fp32 arithmetic: translations re-centred on the first pose (see TransPart)
      // (round 5) the d = 3 chains run this kernel BESIDE k_lin on a second stream: with an update pending (PendUpd) both read the
      // old state buffer and apply the update themselves; k_lin alone writes the other buffer
      constexpr bool PEND = JAC && IsF64<T>::v && two && (MF == POSE2 || MF == ROT3 || MF == LINEAR3);
      if constexpr (PEND) {
        double q1[pd], q2[pd], w1[d], w2[d];
        load_two_states_upd<MF>(a.pose, a.vel, a.stride, i, a.pend, false, false, q1, w1, q2, w2);
#pragma unroll
        for (int k = 0; k < pd; k++) { p1[k] = T(q1[k]); p2[k] = T(q2[k]); }
#pragma unroll
        for (int k = 0; k < d; k++) { v1[k] = T(w1[k]); v2[k] = T(w2[k]); }
      } else {
      {
        double q1[pd], q2[pd];
#pragma unroll
        for (int k = 0; k < pd; k++) { q1[k] = a.pose[(size_t)k * a.stride + i]; q2[k] = two ? a.pose[(size_t)k * a.stride + i + 1] : 0.0; }
        if (!IsF64<T>::v) {
#pragma unroll
          for (int k = 0; k < TransPart<MF>::n; k++) {
            if (k < 3) org[k] = q1[TransPart<MF>::off + k];
            if (two) q2[TransPart<MF>::off + k] -= q1[TransPart<MF>::off + k];
            q1[TransPart<MF>::off + k] = 0.0;
          }
        }
#pragma unroll
        for (int k = 0; k < pd; k++) { p1[k] = T(q1[k]); p2[k] = T(q2[k]); }
      }
#pragma unroll
      for (int k = 0; k < d; k++) { v1[k] = a.vel[(size_t)k * a.stride + i]; v2[k] = two ? a.vel[(size_t)k * a.stride + i + 1] : T(0); }
      }
      if constexpr (MF == POSE3 && two) {
        if (a.vw) {     // GaussianProcessInterpolatorPose3VW.h:79-80
          T b1[6], b2[6];
          vw_to_vb(p1, v1, b1);
          vw_to_vb(p2, v2, b2);
#pragma unroll
          for (int k = 0; k < 6; k++) { v1[k] = b1[k]; v2[k] = b2[k]; }
        }
      }
      int lm = -1;
      T pt[3] = {T(0), T(0), T(0)};
      if (haslm) {
        lm = a.lm[f];
        for (int q = 0; q < a.ld; q++) pt[q] = T(a.lmk[(size_t)lm * a.ld + q] - org[q]);
      }
      const T *gprec = nullptr;      // the interval's GaussianProcessPriorPose3 record, when K1 left one (see MeasArgs::gps)
      if constexpr (MF == POSE3 && JAC && std::is_same<T, double>::value) {
        if (a.gps != nullptr && two) {
          const int g = a.gpidx[i];
          if (g >= 0) gprec = a.gps + (size_t)g * kGpsLen;
        }
      }
      ICoef<T> kc = {T(0), T(0), T(0), T(0)};
      if (FK == FK_INTERP_RANGE || FK == FK_INTERP_ATT || FK == FK_INTERP_GPS || FK == FK_INTERP_PROJ)
        kc = {T(a.coef[4 * (size_t)f]), T(a.coef[4 * (size_t)f + 1]), T(a.coef[4 * (size_t)f + 2]), T(a.coef[4 * (size_t)f + 3])};
      T ms[6];
#pragma unroll
      for (int k = 0; k < 6; k++) ms[k] = (k < a.mw) ? T(a.meas[(size_t)f * a.mw + k]) : T(0);
      // per-factor body_P_sensor / calibration (identical entries are shared through the table)
      const double *ax = a.aidx ? a.aux + (size_t)a.aidx[f] * kMeasAux : nullptr;
      const bool has_sensor = ax && ax[17] != 0.0;
      T sens[12];
#pragma unroll
      for (int k = 0; k < 12; k++) sens[k] = (has_sensor && k < pd) ? T(ax[k]) : T(0);
      T e[rows];
      T Jm[JAC ? rows * 3 : 1];
      if (JAC) {
#pragma unroll
        for (int k = 0; k < rows * b; k++) { JL[k] = T(0); JR[k] = T(0); }
#pragma unroll
        for (int k = 0; k < rows * 3; k++) Jm[k] = T(0);
      }

      if constexpr ((FK == FK_INTERP_RANGE || FK == FK_RANGE) && MF == POSE3) {
        // GPInterpolatedRangeFactorPose3::evaluateError, gpslam/slam/GPInterpolatedRangeFactorPose3.h:64-98
        Interp6Out<T, JAC> o;
        SE3<T> pose = (FK == FK_INTERP_RANGE) ? interp_pose3<T, JAC>(p1, v1, p2, v2, kc, o, gprec) : as_se3(p1);
        const SE3<T> S = as_se3(sens);
        const SE3<T> sp = has_sensor ? se3_compose(pose, S) : pose;
        const V3<T> pw = {pt[0], pt[1], pt[2]};
        const V3<T> q = tmul(sp.R, pw - sp.t);                       // Pose3::transform_to
        const T r = sqrt(dot(q, q));
        const V3<T> qh = (T(1) / r) * q;
        e[0] = r - ms[0];
        if (JAC) {
          V6<T> Hr = {rowmul(qh, skew(q)), -qh};                     // D_r_local * [skew(q), -I]
          if (has_sensor) Hr = rowmul(Hr, se3_adjoint(se3_inverse(S)));   // Hpose * H0 (:87)
          put_v3(sp.R * qh, Jm);                                     // D_r_local * R^T
          if (FK == FK_INTERP_RANGE) {
            put_v6(rowmul(Hr, o.H1), JL); put_v6(rowmul(Hr, o.H2), JL + 6);
            put_v6(rowmul(Hr, o.H3), JR); put_v6(rowmul(Hr, o.H4), JR + 6);
          } else {
            put_v6(Hr, JL);
          }
        }
      } else if constexpr ((FK == FK_INTERP_RANGE || FK == FK_RANGE) && MF == POSE2) {
        // GPInterpolatedRangeFactorPose2::evaluateError, gpslam/slam/GPInterpolatedRangeFactorPose2.h:64-98
        Interp3Out<T, JAC> o;
        SE2<T> pose = (FK == FK_INTERP_RANGE) ? interp_pose2<T, JAC>(p1, v1, p2, v2, kc, o) : SE2<T>{p1[0], p1[1], p1[2]};
        const SE2<T> S = {sens[0], sens[1], sens[2]};
        const SE2<T> sp = has_sensor ? se2_compose(pose, S) : pose;
        const T dx = pt[0] - sp.x, dy = pt[1] - sp.y;
        const T r = sqrt(dx * dx + dy * dy);
        const T hx = dx / r, hy = dy / r;
        e[0] = r - ms[0];
        if (JAC) {
          const T c = cos(sp.th), sn = sin(sp.th);
          V3<T> Hr = {-c * hx - sn * hy, sn * hx - c * hy, T(0)};   // D_r_d * [[-c, s, 0], [-s, -c, 0]]
          if (has_sensor) Hr = rowmul(Hr, se2_adjoint(se2_inverse(S)));
          Jm[0] = hx; Jm[1] = hy;
          if (FK == FK_INTERP_RANGE) {
            put_v3(rowmul(Hr, o.H1), JL); put_v3(rowmul(Hr, o.H2), JL + 3);
            put_v3(rowmul(Hr, o.H3), JR); put_v3(rowmul(Hr, o.H4), JR + 3);
          } else {
            put_v3(Hr, JL);
          }
        }
      } else if constexpr ((FK == FK_INTERP_RANGE || FK == FK_RANGE) && MF == LINEAR3) {
        // GPInterpolatedRangeFactor2DLinear (GPInterpolatedRangeFactor2DLinear.h:60-88) / RangeFactor2DLinear (:43-56)
        T px, py;
        if (FK == FK_INTERP_RANGE) {
          px = kc.l11 * p1[0] + kc.l12 * v1[0] + kc.p11 * p2[0] + kc.p12 * v2[0];
          py = kc.l11 * p1[1] + kc.l12 * v1[1] + kc.p11 * p2[1] + kc.p12 * v2[1];
        } else { px = p1[0]; py = p1[1]; }
        const T dx = pt[0] - px, dy = pt[1] - py;
        const T r = sqrt(dx * dx + dy * dy);
        const T hx = dx / r, hy = dy / r;
        e[0] = r - ms[0];
        if (JAC) {
          Jm[0] = hx; Jm[1] = hy;
          if (FK == FK_INTERP_RANGE) {
            JL[0] = -kc.l11 * hx; JL[1] = -kc.l11 * hy; JL[3] = -kc.l12 * hx; JL[4] = -kc.l12 * hy;
            JR[0] = -kc.p11 * hx; JR[1] = -kc.p11 * hy; JR[3] = -kc.p12 * hx; JR[4] = -kc.p12 * hy;
          } else { JL[0] = -hx; JL[1] = -hy; }
        }
      } else if constexpr (FK == FK_INTERP_ATT) {
        // GPInterpolatedAttitudeFactorRot3::evaluateError, gpslam/slam/GPInterpolatedAttitudeFactorRot3.h:61-83
        Interp3Out<T, JAC> o;
        const M3<T> R = interp_rot3<T, JAC>(p1, v1, p2, v2, kc, o);
        const V3<T> nZ = {ms[0], ms[1], ms[2]}, bRef = {ms[3], ms[4], ms[5]};
        V3<T> q = R * bRef;
        q = (T(1) / sqrt(dot(q, q))) * q;
        V3<T> z1, z2;
        unit3_basis(nZ, z1, z2);
        e[0] = dot(z1, q); e[1] = dot(z2, q);
        if (JAC) {
          V3<T> q1, q2;
          unit3_basis(q, q1, q2);
          const M3<T> RS = R * skew(bRef);
          const V3<T> n1 = -rowmul(q1, RS), n2 = -rowmul(q2, RS);       // D_nRef_R = -Bq^T R [bRef]x
          const T d00 = dot(z1, q1), d01 = dot(z1, q2), d10 = dot(z2, q1), d11 = dot(z2, q2);   // Bz^T Bq
          const V3<T> h0 = d00 * n1 + d01 * n2, h1 = d10 * n1 + d11 * n2;
          // (the velocity columns start at d: 3 for Rot3 states, 6 for the AHRS state whose pose slot also holds the bias)
          put_v3(rowmul(h0, o.H1), JL); put_v3(rowmul(h0, o.H2), JL + d); put_v3(rowmul(h0, o.H3), JR); put_v3(rowmul(h0, o.H4), JR + d);
          put_v3(rowmul(h1, o.H1), JL + b); put_v3(rowmul(h1, o.H2), JL + b + d); put_v3(rowmul(h1, o.H3), JR + b); put_v3(rowmul(h1, o.H4), JR + b + d);
        }
      } else if constexpr (FK == FK_AHRS) {
        // gtsam::AHRSFactor(x_i, x_i+1, b_i, pim) (matlab/GPAHRSexample.m:131-137); noise model = Gaussian with the
        // pre-integrated covariance: the rows are whitened here by its square-root information R (sigmas = 1 below),
        // except in the inspection call, which returns the unwhitened evaluateError() values
        const double *pp = a.meas + (size_t)f * a.mw;
        T prm[25];
#pragma unroll
        for (int k = 0; k < 25; k++) prm[k] = T(pp[k]);
        M3<T> A, B, C;
        ahrs_factor<T, JAC>(as_m3(p1), as_m3(p2), V3<T>{p1[9], p1[10], p1[11]}, prm, e, A, B, C);
        const bool whiten = !(JAC && a.out_e);
        T Rw[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Rw[k] = whiten ? T(pp[25 + k]) : ((k == 0 || k == 4 || k == 8) ? T(1) : T(0));
        const V3<T> ew = as_m3(Rw) * V3<T>{e[0], e[1], e[2]};
        e[0] = ew.x; e[1] = ew.y; e[2] = ew.z;
        if (JAC) {
          put_m3(as_m3(Rw) * A, JL, b, 0, 0);     // d/dx_i
          put_m3(as_m3(Rw) * C, JL, b, 0, 3);     // d/db_i
          put_m3(as_m3(Rw) * B, JR, b, 0, 0);     // d/dx_i+1
        }
      } else if constexpr (FK == FK_INTERP_GPS) {
        // GPInterpolatedGPSFactorPose3::evaluateError, gpslam/slam/GPInterpolatedGPSFactorPose3.h:66-95
        Interp6Out<T, JAC> o;
        BL6<T> He, Hc21, s1;
        SE3<T> pose;
        if constexpr (IROW) pose = interp_pose3_parts<T>(p1, v1, p2, v2, kc, gprec, He, Hc21, s1);
        else pose = interp_pose3<T, JAC>(p1, v1, p2, v2, kc, o, gprec);
        const SE3<T> S = as_se3(sens);
        const SE3<T> sp = has_sensor ? se3_compose(pose, S) : pose;
        e[0] = sp.t.x - ms[0]; e[1] = sp.t.y - ms[1]; e[2] = sp.t.z - ms[2];
        if constexpr (IROW) {
          // JL row r = [Lp | mu] (the left half of the Jacobian is [Lp | l12 mu]); JR stays zero: the consumer forms it
          const BL6<T> AdS = se3_adjoint(se3_inverse(S));
#pragma unroll
          for (int r = 0; r < 3; r++) {
            V6<T> Hp = {{T(0), T(0), T(0)}, {sp.R.m[3 * r], sp.R.m[3 * r + 1], sp.R.m[3 * r + 2]}};   // translation(H) = [0, R]
            if (has_sensor) Hp = rowmul(Hp, AdS);
            const V6<T> mu = rowmul(Hp, He);
            put_v6(rowmul(Hp, Hc21) + rowmul(mu, s1), JL + r * b); put_v6(mu, JL + r * b + 6);
          }
        } else if (JAC) {
          const BL6<T> AdS = se3_adjoint(se3_inverse(S));
#pragma unroll
          for (int r = 0; r < 3; r++) {
            V6<T> Hp = {{T(0), T(0), T(0)}, {sp.R.m[3 * r], sp.R.m[3 * r + 1], sp.R.m[3 * r + 2]}};   // translation(H) = [0, R]
            if (has_sensor) Hp = rowmul(Hp, AdS);
            put_v6(rowmul(Hp, o.H1), JL + r * b); put_v6(rowmul(Hp, o.H2), JL + r * b + 6);
            put_v6(rowmul(Hp, o.H3), JR + r * b); put_v6(rowmul(Hp, o.H4), JR + r * b + 6);
          }
        }
      } else if constexpr (FK == FK_INTERP_PROJ) {
        // GPInterpolatedProjectionFactorPose3<CALIBRATION>::evaluateError, GPInterpolatedProjectionFactorPose3.h:82-139:
        // PinholeCamera(pose * body_P_sensor, K).project(point); a landmark behind the camera is masked, not thrown
        // (throwCheirality = false): error = 2 fx, all Jacobians zero (:122-138).  CALIBRATION = Cal3_S2, or Cal3DS2 when the
        // entry carries distortion coefficients (radial k1, k2, tangential p1, p2: Cal3DS2_Base::uncalibrate)
        Interp6Out<T, JAC> o;
        const SE3<T> pose = interp_pose3<T, JAC>(p1, v1, p2, v2, kc, o, gprec);
        const SE3<T> S = as_se3(sens);
        const SE3<T> cam = has_sensor ? se3_compose(pose, S) : pose;
        const V3<T> pw = {pt[0], pt[1], pt[2]};
        const V3<T> q = tmul(cam.R, pw - cam.t);
        const T fx = ax ? T(ax[12]) : T(1), fy = ax ? T(ax[13]) : T(1), sk = ax ? T(ax[14]) : T(0), cu0 = ax ? T(ax[15]) : T(0), cv0 = ax ? T(ax[16]) : T(0);
        if (!(q.z > T(0))) {
          e[0] = T(2) * fx; e[1] = T(2) * fx;
        } else {
          const T dz = T(1) / q.z, u = q.x * dz, v = q.y * dz;
          const T k1 = ax ? T(ax[18]) : T(0), k2 = ax ? T(ax[19]) : T(0), t1 = ax ? T(ax[20]) : T(0), t2 = ax ? T(ax[21]) : T(0);
          const T xx = u * u, yy = v * v, xy = u * v, rr = xx + yy;
          const T g = T(1) + k1 * rr + k2 * rr * rr;
          const T pdx = g * u + T(2) * t1 * xy + t2 * (rr + T(2) * xx);     // the distorted intrinsic point (= (u, v) for Cal3_S2)
          const T pdy = g * v + T(2) * t2 * xy + t1 * (rr + T(2) * yy);
          e[0] = fx * pdx + sk * pdy + cu0 - ms[0];
          e[1] = fy * pdy + cv0 - ms[1];
          if (JAC) {
            // PinholeBase::Dpose / Dpoint, then uncalibrate's [[fx, s], [0, fy]] D(pd)/D(pn)
            const T dgx = T(2) * k1 * u + T(4) * k2 * rr * u, dgy = T(2) * k1 * v + T(4) * k2 * rr * v;
            const T d00 = g + u * dgx + T(2) * t1 * v + T(6) * t2 * u, d01 = u * dgy + T(2) * t1 * u + T(2) * t2 * v;
            const T d10 = v * dgx + T(2) * t2 * v + T(2) * t1 * u, d11 = g + v * dgy + T(2) * t2 * u + T(6) * t1 * v;
            const T m00 = fx * d00 + sk * d10, m01 = fx * d01 + sk * d11, m10 = fy * d10, m11 = fy * d11;
            const V6<T> r0 = {{u * v, T(-1) - u * u, v}, {-dz, T(0), dz * u}};
            const V6<T> r1 = {{T(1) + v * v, -u * v, -u}, {T(0), -dz, dz * v}};
            V6<T> h0 = m00 * r0 + m01 * r1, h1 = m10 * r0 + m11 * r1;
            if (has_sensor) {
              const BL6<T> AdS = se3_adjoint(se3_inverse(S));
              h0 = rowmul(h0, AdS);
              h1 = rowmul(h1, AdS);
            }
            const M3<T> &R = cam.R;
            const V3<T> c0 = {R.m[0], R.m[3], R.m[6]}, c1 = {R.m[1], R.m[4], R.m[7]}, c2 = {R.m[2], R.m[5], R.m[8]};
            const V3<T> d0 = dz * (c0 - u * c2), d1 = dz * (c1 - v * c2);
            put_v3(m00 * d0 + m01 * d1, Jm);
            put_v3(m10 * d0 + m11 * d1, Jm + 3);
            put_v6(rowmul(h0, o.H1), JL); put_v6(rowmul(h0, o.H2), JL + 6); put_v6(rowmul(h0, o.H3), JR); put_v6(rowmul(h0, o.H4), JR + 6);
            put_v6(rowmul(h1, o.H1), JL + b); put_v6(rowmul(h1, o.H2), JL + b + 6); put_v6(rowmul(h1, o.H3), JR + b); put_v6(rowmul(h1, o.H4), JR + b + 6);
          }
        }
      } else if constexpr (FK == FK_ODOM2D) {
        // OdometryFactor2DLinear::evaluateError, gpslam/slam/OdometryFactor2DLinear.h:50-75
        const T dx = p2[0] - p1[0], dy = p2[1] - p1[1], dth = p2[2] - p1[2];
        const T c = cos(p1[2]), sn = sin(p1[2]);
        const T qx = c * dx + sn * dy, qy = -sn * dx + c * dy;
        e[0] = qx - ms[0]; e[1] = qy - ms[1]; e[2] = dth - ms[2];
        if (JAC) {
          JL[0] = -c; JL[1] = -sn; JL[2] = qy;
          JL[b + 0] = sn; JL[b + 1] = -c; JL[b + 2] = -qx;
          JL[2 * b + 2] = T(-1);
          JR[0] = c; JR[1] = sn;
          JR[b + 0] = -sn; JR[b + 1] = c;
          JR[2 * b + 2] = T(1);
        }
      } else if constexpr (FK == FK_BEARING_RANGE) {
        // RangeBearingFactor2DLinear::evaluateError, gpslam/slam/RangeBearingFactor2DLinear.h:47-84; meas = (bearing, range)
        const T c = cos(p1[2]), sn = sin(p1[2]);
        const T dx = pt[0] - p1[0], dy = pt[1] - p1[1];
        const T rx = c * dx + sn * dy, ry = -sn * dx + c * dy;           // pose2.transform_to(point)
        const T ed = sqrt(dx * dx + dy * dy);
        const T hx = dx / ed, hy = dy / ed;
        const T df = atan2(ry, rx) - ms[0];
        e[0] = atan2(sin(df), cos(df));
        e[1] = ed - ms[1];
        if (JAC) {
          T t0 = T(0), t1 = T(0);
          if (ed > T(1e-5)) { t0 = -ry / (ed * ed); t1 = rx / (ed * ed); }   // :62
          JL[0] = t0 * (-c) + t1 * sn; JL[1] = t0 * (-sn) + t1 * (-c); JL[2] = t0 * ry + t1 * (-rx);   // tmp * [-R^T, t]
          JL[b + 0] = -hx; JL[b + 1] = -hy;
          Jm[0] = t0 * c + t1 * (-sn); Jm[1] = t0 * sn + t1 * c;       // tmp * R^T
          Jm[3 + 0] = hx; Jm[3 + 1] = hy;
        }
      }

      if constexpr (MF == POSE3 && two && JAC) {
        if (a.vw) {     // chain rule through convertVWtoVb (v1 / v2 hold the body velocities here)
#pragma unroll
          for (int r = 0; r < rows; r++) { vw_row_transform(p1, v1, JL + r * b); vw_row_transform(p2, v2, JR + r * b); }
        }
      }
      if (JAC && a.out_e) {
#pragma unroll
        for (int r = 0; r < rows; r++) {
          a.out_e[(size_t)f * rows + r] = e[r];
          T *o = a.out_J + ((size_t)f * rows + r) * (2 * b + 3);
#pragma unroll
          for (int c = 0; c < b; c++) { o[c] = JL[r * b + c]; o[b + c] = JR[r * b + c]; }
#pragma unroll
          for (int c = 0; c < 3; c++) o[2 * b + c] = Jm[r * 3 + c];
        }
      }
      const int row0 = (JAC || a.rowE32) ? a.row0[f] : 0;
      row0v = row0;
      constexpr bool kFullNoise = rows > 1 && FK != FK_AHRS;     // (AHRSFactor whitens by its pre-integrated covariance above)
      if constexpr (kFullNoise) {
        if (a.sqi) {   // noiseModel::Gaussian: rows <- R rows; R is upper triangular, so ascending in place
          const double *Rw = a.sqi + (size_t)f * rows * rows;
#pragma unroll
          for (int r = 0; r < rows; r++) {
            T acc = T(0);
#pragma unroll
            for (int q = r; q < rows; q++) acc += T(Rw[r * rows + q]) * e[q];
            e[r] = acc;
            if (JAC) {
#pragma unroll
              for (int c = 0; c < b; c++) {
                T aL = T(0), aR = T(0);
#pragma unroll
                for (int q = r; q < rows; q++) { aL += T(Rw[r * rows + q]) * JL[q * b + c]; aR += T(Rw[r * rows + q]) * JR[q * b + c]; }
                JL[r * b + c] = aL; JR[r * b + c] = aR;
              }
#pragma unroll
              for (int c = 0; c < 3; c++) {
                T am = T(0);
#pragma unroll
                for (int q = r; q < rows; q++) am += T(Rw[r * rows + q]) * Jm[q * 3 + c];
                Jm[r * 3 + c] = am;
              }
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < rows; r++) {
        const T w = (kFullNoise && a.sqi) ? T(1) : T(1) / T(a.sig[(size_t)f * rows + r]);
        const T we = e[r] * w;
        err += we * we;
        wgt[r] = w;
        if constexpr (IROW) ew_[r] = we;
        if (!JAC && a.rowE32) a.rowE32[row0 + r] = (float)we;
        if (JAC && a.rowLR) {      // (rowLR == null: the inspection call, gpslam_hip_linearize_meas, leaves the row tables alone)
          a.rowE[row0 + r] = we;
          if (a.ld > 0) {
            a.rowLm[row0 + r] = lm;
            for (int q = 0; q < a.ld; q++) a.rowM[(size_t)(row0 + r) * a.ld + q] = w * Jm[r * 3 + q];
          }
        }
      }
    }
    if constexpr (IROW) {
      __shared__ T istage[2 * 64 * 20];
      __shared__ int isrow[128];
      const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
      T *st = istage + wv * 64 * 20, *mine = st + lane * 20;
      isrow[threadIdx.x] = row0v;
      const bool live = f < a.count;
      const T p11 = live ? T(a.coef[4 * (size_t)f + 2]) : T(0), p12 = live ? T(a.coef[4 * (size_t)f + 3]) : T(0), l12 = live ? T(a.coef[4 * (size_t)f + 1]) : T(0);
#pragma unroll
      for (int r = 0; r < rows; r++) {
#pragma unroll
        for (int c = 0; c < 12; c++) mine[c] = wgt[r] * JL[r * b + c];
        mine[kIRowE] = ew_[r]; mine[kIRowP11] = p11; mine[kIRowP12] = p12; mine[kIRowL12] = l12;
        wave_store_part<T, kIRowLen, 16, 20, true>(st, isrow + wv * 64, lane, r, 0, a.rowI);
      }
    } else if constexpr (JAC) {
      constexpr int LS = 2 * b + 2;
      __shared__ T stage[2 * 64 * LS];
      __shared__ int srow[128];
      const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
      T *st = stage + wv * 64 * LS, *mine = st + lane * LS;
      srow[threadIdx.x] = row0v;
      if (a.rowLR) {
#pragma unroll
        for (int r = 0; r < rows; r++) {
#pragma unroll
          for (int c = 0; c < b; c++) { mine[c] = wgt[r] * JL[r * b + c]; mine[b + c] = wgt[r] * JR[r * b + c]; }
          wave_store_rows<T, 2 * b>(st, srow + wv * 64, lane, r, a.rowLR);
        }
      }
    }
  }
  const T tot = block_sum(T(0.5) * err);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = tot;
}

// GPInterpolatedGPSFactorPose3 on the structured path, as 16-double lines (kIRow*) -- the kernel k_meas<double, POSE3, FK_INTERP_GPS,
// true, true> is, written for its register count (round 6).  The general kernel keeps every row of the factor (rows x 24 doubles), the
// interpolator's three 6 x 6 blocks and the interval's record alive at once: 256 VGPRs + AGPR spill space, ONE wave per SIMD, 0.88 ms
// for 4e6 factors at 0.37 of the HBM roof -- a latency chain nobody covers.  Here the whitening enters through Hp (the rows are
// linear in it: diag(1 / sigma) or the square-root information R of a Gaussian model), so a row is final the moment it is formed;
// the interval's Jinv is dead once xi exists, the record's J and F blocks are requested when the lines are half done, and the pose's
// 12 numbers are the only thing that lives from the first load to the last row.  Same device functions on the same operands as
// interp_pose3_parts (GaussianProcessInterpolatorPose3.h:57-105, GPInterpolatedGPSFactorPose3.h:66-95): products with the weight
// commute by one rounding.  The interval must carry a GP prior (compile(): irow_ok), body-frame velocities (struct_ok).
// materialise values HERE: an empty volatile asm that "modifies" them keeps the compiler from sinking their computation towards the
// use (and, on a pointer, from hoisting the loads behind it) -- the phases of k_gps_lines below stay phases
template <typename P> __device__ __forceinline__ void pin_ptr(P *&p) { asm volatile("" : "+v"(p)); }
#ifndef GPS_GPS_LINES_WAVES
#define GPS_GPS_LINES_WAVES 2   /* 0: the general k_meas<..., IROW> kernel (round 5) */
#endif
// se3_Q (lie.hpp; Pose3utils.cpp:92-113) term by term: the same products and the same order of every sum, each term folded into
// the result before the next one's operands exist (the expression as one statement keeps eight 3 x 3 temporaries alive)
__device__ __forceinline__ M3<double> se3_Q_seq(V3<double> w, V3<double> rho) {
  typedef double T;
  const T th = sqrt(dot(w, w));
  T a, b, c;
  if (fabs(th) > T(1e-5)) {
    const T s = sin(th), co = cos(th);
    const T t2 = th * th, t3 = t2 * th, t4 = t3 * th, t5 = t4 * th;
    a = (th - s) / t3;
    b = (T(1) - T(0.5) * t2 - co) / t4;
    c = T(-0.5) * ((T(1) - T(0.5) * t2 - co) / t4 - T(3) * (th - s - t3 / T(6)) / t5);
  } else {
    a = T(1) / T(6);
    b = T(1) / T(24);
    c = T(-0.5) * (T(1) / T(24) + T(3) / T(120));
  }
  pin(a); pin(b); pin(c);
  const M3<T> X = skew(w), Y = skew(rho);
  M3<T> XY = X * Y, YX = Y * X, XYX = X * YX;
  M3<T> Q = T(-0.5) * Y + a * (XY + YX - XYX);
  pin(Q);
  Q = Q + b * (X * XY + YX * X - T(3) * XYX);
  pin(Q);
  Q = Q + c * (XYX * X + X * XYX);
  return Q;
}
// SENS: some factor of the launch carries a body_P_sensor (a.aidx != null).  Without one Hp = W [0 | R]: its rotational half is
// exactly zero and only the lower block rows of He and Ad(Exp(xi)^-1) are touched -- half of Hp's registers and products.
template <int WAVES, bool SENS>
__global__ void __launch_bounds__(128, WAVES) k_gps_lines(MeasArgs<double> a) {
  typedef double T;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = f < a.count;
  const int ff = live ? f : 0;                   // (idle lanes of the last block recompute factor 0; they store nothing)
  const int i = a.idx[ff];
  const T l12 = a.coef[4 * (size_t)ff + 1], p11 = a.coef[4 * (size_t)ff + 2], p12 = a.coef[4 * (size_t)ff + 3];
  const T *gp = a.gps + (size_t)a.gpidx[i] * kGpsLen;
  const T *gp2 = gp;
  auto m3 = [&](int off) { M3<T> m; for (int q = 0; q < 9; q++) m.m[q] = gp[off + q]; return m; };
  SE3<T> A1;
  V6<T> xi;
  {
    // SoA states through ONE 32-bit byte offset per lane and uniform bases (global_load saddr + voffset): 36 loads with 64-bit
    // per-lane addresses were 72 address registers, more than everything they fetched
    const unsigned off = (unsigned)i * 8u;
    auto ldu = [&](const double *base) { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + off); };
    T p1[12], p2[12];
#pragma unroll
    for (int k = 0; k < 12; k++) { p1[k] = ldu(a.pose + (size_t)k * a.stride); p2[k] = ldu(a.pose + (size_t)k * a.stride + 1); }
    V6<T> u1, u2;
    {
      T v1[6], v2[6];
#pragma unroll
      for (int k = 0; k < 6; k++) { v1[k] = ldu(a.vel + (size_t)k * a.stride); v2[k] = ldu(a.vel + (size_t)k * a.stride + 1); }
      u1 = as_v6(v1); u2 = as_v6(v2);
    }
    A1 = as_se3(p1);
    const V6<T> r = se3_log(se3_between(A1, as_se3(p2)));
    BL6<T> Jinv;
    Jinv.A = m3(kGpsXA); Jinv.C = m3(kGpsXC); Jinv.D = Jinv.A;
    xi = l12 * u1 + p11 * r + p12 * (Jinv * u2);
  }
  pin(xi);
  SE3<T> ex = se3_exp(xi);
  pin(ex.R); pin(ex.t);
  // the sensor pose and the whitened Hp = W [0 | R_sp] (Ad(S^-1)): three 6-vectors (SENS) or their translational halves
  V6<T> Hp[SENS ? 3 : 1];
  V3<T> hv[SENS ? 1 : 3];
  T ew[3];
  {
    const SE3<T> pose = se3_compose(A1, ex);
    SE3<T> sp = pose;
    BL6<T> AdS;
    bool has_sensor = false;
    if constexpr (SENS) {
      const double *ax = a.aux + (size_t)a.aidx[ff] * kMeasAux;
      has_sensor = ax[17] != 0.0;
      if (has_sensor) {
        T sens[12];
#pragma unroll
        for (int k = 0; k < 12; k++) sens[k] = ax[k];
        const SE3<T> S = as_se3(sens);
        sp = se3_compose(pose, S);
        AdS = se3_adjoint(se3_inverse(S));
      }
    }
    T e[3] = {sp.t.x - a.meas[(size_t)ff * a.mw], sp.t.y - a.meas[(size_t)ff * a.mw + 1], sp.t.z - a.meas[(size_t)ff * a.mw + 2]};
    // W: diag(1 / sigma), or the upper triangular square-root information of a noiseModel::Gaussian (rows <- R rows)
    T W[9] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
    if (a.sqi) {
      const double *Rw = a.sqi + (size_t)ff * 9;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = r; q < 3; q++) W[r * 3 + q] = Rw[r * 3 + q];
    } else {
#pragma unroll
      for (int r = 0; r < 3; r++) W[r * 4] = T(1) / a.sig[(size_t)ff * 3 + r];
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
      T acc = T(0);
      V3<T> hr = {T(0), T(0), T(0)};
#pragma unroll
      for (int q = r; q < 3; q++) {
        acc += W[r * 3 + q] * e[q];
        hr = hr + W[r * 3 + q] * V3<T>{sp.R.m[3 * q], sp.R.m[3 * q + 1], sp.R.m[3 * q + 2]};   // translation(H) = [0, R]
      }
      ew[r] = acc;
      if constexpr (SENS) {
        Hp[r] = {{T(0), T(0), T(0)}, hr};
        if (has_sensor) Hp[r] = rowmul(Hp[r], AdS);
        pin(Hp[r]);
      } else {
        hv[r] = hr;
        pin(hv[r]);
      }
      pin(ew[r]);
    }
  }
  T err = live ? ew[0] * ew[0] + ew[1] * ew[1] + ew[2] * ew[2] : T(0);
  V6<T> mu[3], Lp[3];
  {
    // He = se3_jr(xi) = [[Jr, 0], [Q, Jr]] (rightJacobianPose3, Pose3utils.cpp:182-189), the Q block first
    M3<T> HQ = se3_Q_seq(xi.w, xi.v);
    pin(HQ);
    const M3<T> Jw = so3_jr(xi.w);
#pragma unroll
    for (int r = 0; r < 3; r++) {
      if constexpr (SENS) mu[r] = rowmul(Hp[r], BL6<T>{Jw, HQ, Jw});
      else mu[r] = {rowmul(hv[r], HQ), rowmul(hv[r], Jw)};
      pin(mu[r]);
    }
  }
  {
    const BL6<T> Hc21 = se3_adjoint(se3_inverse(ex));
#pragma unroll
    for (int r = 0; r < 3; r++) {
      if constexpr (SENS) Lp[r] = rowmul(Hp[r], Hc21);
      else Lp[r] = {rowmul(hv[r], Hc21.C), rowmul(hv[r], Hc21.D)};
      pin(Lp[r]);
    }
  }
  pin_ptr(gp2);      // the record's J and F blocks are requested from here on, not before
  {   // Lp += mu (p11 J + p12 F J) as (p11 mu + p12 (mu F)) J, one 3 x 3 block of the record at a time: F J is never formed and no
      // two blocks are alive together (mu F = [mu_w FA + mu_v FC | mu_v FD], b J = [b_w JA + b_v JC | b_v JA])
    auto n3 = [&](int off) { M3<T> m; for (int q = 0; q < 9; q++) m.m[q] = gp2[off + q]; return m; };
    V6<T> bq[3];
    {
      M3<T> blk = n3(kGpsFA);
#pragma unroll
      for (int r = 0; r < 3; r++) { bq[r].w = rowmul(mu[r].w, blk); pin(bq[r].w); }
      blk = n3(kGpsFC);
#pragma unroll
      for (int r = 0; r < 3; r++) { bq[r].w = p11 * mu[r].w + p12 * (bq[r].w + rowmul(mu[r].v, blk)); pin(bq[r].w); }
      blk = n3(kGpsFD);
#pragma unroll
      for (int r = 0; r < 3; r++) { bq[r].v = p11 * mu[r].v + p12 * rowmul(mu[r].v, blk); pin(bq[r].v); }
    }
    pin_ptr(gp2);
    {
      M3<T> blk = n3(kGpsJA);
#pragma unroll
      for (int r = 0; r < 3; r++) { Lp[r].w = Lp[r].w + rowmul(bq[r].w, blk); Lp[r].v = Lp[r].v + rowmul(bq[r].v, blk); pin(Lp[r]); }
      blk = n3(kGpsJC);
#pragma unroll
      for (int r = 0; r < 3; r++) { Lp[r].w = Lp[r].w + rowmul(bq[r].v, blk); pin(Lp[r].w); }
    }
  }
  __shared__ T istage[2 * 64 * 20];
  __shared__ int isrow[128];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T *st = istage + wv * 64 * 20, *mine = st + lane * 20;
  isrow[threadIdx.x] = live ? a.row0[f] : -1;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    put_v6(Lp[r], mine + kIRowLp); put_v6(mu[r], mine + kIRowMu);
    mine[kIRowE] = ew[r]; mine[kIRowP11] = p11; mine[kIRowP12] = p12; mine[kIRowL12] = l12;
    wave_store_part<T, kIRowLen, 16, 20, true>(st, isrow + wv * 64, lane, r, 0, a.rowI);
  }
  const T tot = block_sum(T(0.5) * err);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = tot;
}

// ------------------------------------------------------------------ landmark border (Schur complement)

// TR: type of the Jacobian row tables (float on fp32 handles, whose normal equations and solver stay fp64)
template <typename T, typename TR = T> struct LmArgs {
  int N, R, B, L, ld, nl;   // R = 1 + nl
  int Nx;                   // solution slots in x: N, or N + 1 when a halo state follows the segment
  const TR *rowLR, *rowE, *rowM;
  const int *lmrow;         // row ids of rows that touch a landmark, grouped by landmark
  const int *lmrow_state;   // left state of each such row
  const int *lmrow_ptr;     // L + 1
  int nlmrows;
  const int *chunk_lm, *chunk_j0, *chunk_j1;   // row chunks of the landmark rows (each inside one landmark)
  const int *chunk_ptr;     // L + 1: chunks of landmark l are chunk_ptr[l] .. chunk_ptr[l + 1]
  int nchunks;
  T *part;                  // nchunks x ld x R partial sums of the Schur complement
  T *x;                     // level-0 solutions, N x R x B (column 0 is corrected in place)
  T *t;                     // nlmrows x R
  // landmark priors
  int npri;
  const int *pri_lm;
  const double *pri_meas, *pri_sig;
  double *lmk;              // L x ld (fp64 like the states)
  T *S;                     // nl x (nl + 1): column 0 = rhs, columns 1.. = Schur complement
  T *gL;                    // nl (undamped gradient, for LM)
  T *dL;                    // nl solution
  T lambda;
  int *flag;
  T *partial;
};

// t[j][c] = JL_rho . X_s[:, c] + JR_rho . X_{s+1}[:, c]
template <typename T, typename TR = T> __global__ void __launch_bounds__(128) k_lm_t(LmArgs<T, TR> a) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = tid / a.R, c = tid - j * a.R;
  if (j >= a.nlmrows) return;
  const int rho = a.lmrow[j], s = a.lmrow_state[j];
  const TR *row = a.rowLR + (size_t)rho * 2 * a.B;
  const T *x0 = a.x + ((size_t)s * a.R + c) * a.B;
  T acc = T(0);
  for (int k = 0; k < a.B; k++) acc += row[k] * x0[k];
  if (s + 1 < a.Nx) {
    const T *x1 = a.x + ((size_t)(s + 1) * a.R + c) * a.B;
    for (int k = 0; k < a.B; k++) acc += row[a.B + k] * x1[k];
  }
  a.t[(size_t)j * a.R + c] = acc;
}

// Landmark Schur complement S[al][c]: c = 0 -> rhs gL - B^T x0; c >= 1 -> (H_LL + lambda I - B^T Z)[al][c-1], in two
// stages with a fixed summation order: a landmark may be seen from tens of thousands of rows (Plaza-rate ranges on a
// 1e6-state chain: 55k rows per landmark), so the rows are cut into chunks that one workgroup each reduces, thread
// (q, c) = (component of the landmark, column) running down the chunk; a second kernel adds the chunks of a landmark
// in order, the priors and the damping.
constexpr int kLmChunk = 64;
template <typename T, typename TR = T> __global__ void __launch_bounds__(256) k_lm_reduce_part(LmArgs<T, TR> a) {
  const int ch = blockIdx.x;
  const int W = a.ld * a.R;                       // (q, c) pairs, <= 84
  const int nsub = 256 / W;                       // row phases per pair (the loop is a chain of dependent loads)
  const int sub = threadIdx.x / W, qc = threadIdx.x - sub * W;
  const int q = qc / a.R, c = qc - q * a.R;
  __shared__ T red[2][256];
  T acc = T(0), g = T(0);
  if (sub < nsub) {
    const int lm = a.chunk_lm[ch];
    const int cl = c - 1;
    const bool same = (c >= 1) && (cl / a.ld == lm);
    const int q2 = same ? cl - lm * a.ld : 0;
    for (int j = a.chunk_j0[ch] + sub; j < a.chunk_j1[ch]; j += nsub) {
      const int rho = a.lmrow[j];
      const T m = a.rowM[(size_t)rho * a.ld + q];
      if (c == 0) g -= m * a.rowE[rho];
      else if (same) acc += m * a.rowM[(size_t)rho * a.ld + q2];
      acc -= m * a.t[(size_t)j * a.R + c];
    }
    red[0][threadIdx.x] = acc;
    red[1][threadIdx.x] = g;
  }
  __syncthreads();
  if (sub == 0) {                                  // fixed order over the phases
    T sa = T(0), sg = T(0);
    for (int u = 0; u < nsub; u++) { sa += red[0][u * W + qc]; sg += red[1][u * W + qc]; }
    a.part[((size_t)ch * a.ld + q) * a.R + c] = sa + sg;          // c = 0: gL - B^T x0
    if (c == 0) a.part[((size_t)a.nchunks * a.ld + (size_t)ch * a.ld + q) * a.R] = sg;   // gL alone feeds the LM model
  }
}
// one wave per (al, c): lanes stride over the chunks of the landmark, then a fixed-order wave sum
template <typename T, typename TR = T> __global__ void __launch_bounds__(64) k_lm_reduce(LmArgs<T, TR> a) {
  const int al = blockIdx.x / a.R, c = blockIdx.x - al * a.R;
  const int lm = al / a.ld, q = al - lm * a.ld;
  const int cl = c - 1;
  T acc = T(0), g = T(0);
  for (int ch = a.chunk_ptr[lm] + (int)threadIdx.x; ch < a.chunk_ptr[lm + 1]; ch += 64) {
    acc += a.part[((size_t)ch * a.ld + q) * a.R + c];
    if (c == 0) g += a.part[((size_t)a.nchunks * a.ld + (size_t)ch * a.ld + q) * a.R];
  }
  acc = wave_sum(acc);
  g = wave_sum(g);
  if (threadIdx.x != 0) return;
  for (int k = 0; k < a.npri; k++) {
    if (a.pri_lm[k] != lm) continue;
    const T w = T(1) / a.pri_sig[(size_t)k * a.ld + q];
    if (c == 0) { const T gp = -w * w * (a.lmk[(size_t)lm * a.ld + q] - a.pri_meas[(size_t)k * a.ld + q]); g += gp; acc += gp; }
    else if (cl == al) acc += w * w;
  }
  if (c == 0) a.gL[al] = g;
  else if (cl == al) acc += a.lambda;
  a.S[(size_t)al * a.R + c] = acc;
}

// Dense SPD solve of the (small) landmark system, S = columns 1..nl of a.S (upper part), rhs = column 0 -> dL.
// One wave, right-looking Cholesky in LDS: step j scales column j (lane per row) and applies the rank-1 update to the
// trailing block and to the rhs (lane per entry); then the back-substitution.  nl <= 27.
template <typename T, typename TR = T> __global__ void __launch_bounds__(64) k_lm_solve(LmArgs<T, TR> a) {
  constexpr int NM = kMaxRhs - 1;
  __shared__ T M[NM][NM + 1];      // lower triangle becomes L; +1: no bank conflicts down a column
  __shared__ T y[NM];
  const int n = a.nl, R = a.R, lane = threadIdx.x;
  for (int idx = lane; idx < n * n; idx += 64) {
    const int i = idx / n, j = idx - i * n;
    M[i][j] = (j <= i) ? a.S[(size_t)j * R + 1 + i] : T(0);         // S is stored by its upper part: S[j][i], j <= i
  }
  for (int i = lane; i < n; i += 64) y[i] = a.S[(size_t)i * R];
  wave_lds_sync();
  for (int j = 0; j < n; j++) {
    T dd = M[j][j];
    if (!(dd > T(0))) { if (lane == 0) *a.flag = 1; dd = T(1); }
    const T l = sqrt(dd), linv = T(1) / l;
    wave_lds_sync();
    for (int i = j + lane; i < n; i += 64) M[i][j] = (i == j) ? l : M[i][j] * linv;
    if (lane == 0) y[j] *= linv;
    wave_lds_sync();
    const int m = n - j - 1;                                        // trailing block (j+1 .. n-1)^2, lower part, + rhs
    for (int idx = lane; idx < m * m + m; idx += 64) {
      if (idx < m * m) {
        const int i = j + 1 + idx / m, k = j + 1 + idx % m;
        if (k <= i) M[i][k] -= M[i][j] * M[k][j];
      } else {
        const int i = j + 1 + (idx - m * m);
        y[i] -= M[i][j] * y[j];
      }
    }
    wave_lds_sync();
  }
  // L^T dL = y
  for (int i = n - 1; i >= 0; i--) {
    if (lane == 0) {
      const T v = y[i] / M[i][i];
      y[i] = v;
      a.dL[i] = v;
    }
    wave_lds_sync();
    const T v = y[i];
    for (int k = lane; k < i; k += 64) y[k] -= M[i][k] * v;
    wave_lds_sync();
  }
}

// delta_p = x0 - Z dL  (in place in column 0 of x)
template <typename T, typename TR = T> __global__ void __launch_bounds__(256) k_lm_correct(LmArgs<T, TR> a) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = tid / a.B, k = tid - s * a.B;
  if (s >= a.N) return;
  T *xs = a.x + (size_t)s * a.R * a.B;
  T v = xs[k];
  for (int q = 0; q < a.nl; q++) v -= xs[(size_t)(1 + q) * a.B + k] * a.dL[q];
  xs[k] = v;
}

// landmarks += dL; out[0] = max |dL| (single block)
template <typename T, typename TR = T> __global__ void __launch_bounds__(64) k_lm_update(LmArgs<T, TR> a, double *out_max) {
  T mx = T(0);
  const bool bad = a.flag && *a.flag;          // indeterminate system: leave the landmarks where they are
  for (int i = threadIdx.x; i < a.nl; i += 64) {
    if (!bad) a.lmk[i] += a.dL[i];
    mx = fmax(mx, abs_or_inf(a.dL[i]));
  }
  mx = wave_max(mx);
  if (threadIdx.x == 0) *out_max = fmax(*out_max, (double)mx);
}

// error of the landmark priors (PriorFactor<Point>), single block
template <typename T, typename TR = T> __global__ void __launch_bounds__(128) k_lmprior_err(LmArgs<T, TR> a) {
  T err = T(0);
  for (int k = threadIdx.x; k < a.npri; k += 128)
    for (int q = 0; q < a.ld; q++) {
      const T we = (a.lmk[(size_t)a.pri_lm[k] * a.ld + q] - a.pri_meas[(size_t)k * a.ld + q]) / a.pri_sig[(size_t)k * a.ld + q];
      err += we * we;
    }
  const T tot = block_sum(T(0.5) * err);
  if (threadIdx.x == 0) a.partial[0] = tot;
}

// ---- small utilities for Levenberg-Marquardt
// partial[b] = sum over this block of a[i] * b[i]  (stride-aware: element i lives at a[i * sa], b[i * sb])
// the three dot products of a Levenberg-Marquardt trial in one launch (round 3): x . y0, x . y1 (y1 may be null), x . x, each with
// k_dot's partition into per-block partial sums -- partial[b], partial[nb + b], partial[2 nb + b] -- so that k_final_reduce3 adds
// them up in k_final_reduce's order: the model-fidelity test sees bit-identical numbers
template <typename T> __global__ void __launch_bounds__(256) k_dot3(const T *x, const T *y0, const T *y1, int n, T *partial) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = gridDim.x;
  const T xi = (i < n) ? x[i] : T(0);
  const T t0 = block_sum((i < n) ? xi * y0[i] : T(0));
  __syncthreads();
  const T t1 = block_sum((i < n && y1 != nullptr) ? xi * y1[i] : T(0));
  __syncthreads();
  const T t2 = block_sum((i < n) ? xi * xi : T(0));
  if (threadIdx.x == 0) { partial[blockIdx.x] = t0; partial[nb + blockIdx.x] = t1; partial[2 * nb + blockIdx.x] = t2; }
}
// block b: out[slot[b]] = sum of in[b * n .. b * n + n) in k_final_reduce's order
template <typename T> __global__ void __launch_bounds__(256) k_final_reduce3(const T *in, int n, double *out, int s0, int s1, int s2) {
  const T *p = in + (size_t)blockIdx.x * n;
  const T acc = strided_fold(p, n, 0);
  const T r = block_sum(acc);
  const int slot = blockIdx.x == 0 ? s0 : (blockIdx.x == 1 ? s1 : s2);
  if (threadIdx.x == 0 && slot >= 0) out[slot] = (double)r;
}
template <typename T> __global__ void __launch_bounds__(256) k_dot(const T *x, const T *y, int n, T *partial) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const T v = (i < n) ? x[i] * y[i] : T(0);
  const T tot = block_sum(v);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
// gather column 0 of the N x R x B solution array into a dense N x B vector
template <typename T> __global__ void __launch_bounds__(256) k_gather_delta(const T *x, int N, int R, int B, T *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * B) return;
  const int s = i / B, k = i - s * B;
  out[i] = x[(size_t)s * R * B + k];
}

// ------------------------------------------------------------------ loop closures (round 6)
// gtsam::BetweenFactor<Pose>(x_i, x_j, measured) between NON-adjacent states: the one factor of a SLAM graph that leaves the chain
// (the reference's factors take arbitrary keys the same way, gpslam/gp/GaussianProcessPriorPose3.h:43-47).  Its whitened rows
// U = [.. A_i .. A_j ..] (d x N b, pose columns of two states) make the normal equations H0 + U^T U with H0 the block-tridiagonal
// (+ landmark border) matrix of everything else.  The chain solver stays what it is: the d columns of U^T ride through it as extra
// right-hand sides behind the landmark columns (Z = H0^-1 U^T), and
//   (H0 + U^T U)^-1 [g + U^T r | B] = X + Z Y,   X = H0^-1 [g | B],   Y = (I + U Z)^-1 ([r | 0] - U X),   r = -(whitened error)
// (Sherman-Morrison-Woodbury; I + U Z is d K x d K, symmetric positive definite) corrects the solution column AND the landmark
// columns before the landmark Schur complement is formed, so that landmarks and closures mix freely.  K closures cost K d of the
// kMaxRhs - 1 border columns.  Four small kernels: evaluate, inject the columns into the level-0 records, solve, correct.
struct CloArgs {
  const double *pose;     // SoA states
  int stride, count, chart;
  const int *first, *second;
  const double *meas, *sig;   // count x pose_dim, count x d
  double *A;              // count records [A_i (d x d) | A_j (d x d) | r (d)]: whitened H1, H2 and right-hand side
  double *partial;        // the closures' 0.5 |R e|^2 (one value)
  double *blk;            // level-0 block records [D | O | G (B x R)]
  int BS, B, R, col0;     // record length, block size, right-hand sides, first closure column (1 + nl)
  double *gsave;          // Levenberg-Marquardt: the gradient copy takes U^T r as well, or null
  double *x;              // level-0 solutions N x R x B
  int N, ncols;           // columns 0 .. ncols - 1 (the update and the landmark columns) are corrected
  double *Y;              // nc x ncols
  int *flag;
};
constexpr int kCloLen(int d) { return 2 * d * d + d; }

template <int MF, bool JAC> __global__ void __launch_bounds__(128) k_clo_eval(CloArgs a) {
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd;
  double err = 0.0;
  for (int f = threadIdx.x; f < a.count; f += 128) {
    const int i = a.first[f], j = a.second[f];
    double x1[pd], x2[pd], m[pd], e[d], H1[JAC ? d * d : 1], H2[JAC ? d * d : 1];
#pragma unroll
    for (int k = 0; k < pd; k++) {
      x1[k] = a.pose[(size_t)k * a.stride + i];
      x2[k] = a.pose[(size_t)k * a.stride + j];
      m[k] = a.meas[(size_t)f * pd + k];
    }
    PoseFactors<double, MF, JAC>::between(m, x1, x2, a.chart, e, H1, H2);
    double *rec = a.A + (size_t)f * kCloLen(d);
#pragma unroll
    for (int r = 0; r < d; r++) {
      const double w = 1.0 / a.sig[(size_t)f * d + r];
      const double we = w * e[r];
      err += we * we;
      if (JAC) {
#pragma unroll
        for (int c = 0; c < d; c++) { rec[r * d + c] = w * H1[r * d + c]; rec[d * d + r * d + c] = w * H2[r * d + c]; }
        rec[2 * d * d + r] = -we;
      }
    }
  }
  const double tot = block_sum(0.5 * err);
  if (threadIdx.x == 0) a.partial[0] = tot;
}

// the columns of U^T into the records of the two states of every closure (k_assemble_ghost left those columns zero); one workgroup
template <int d> __global__ void __launch_bounds__(256) k_clo_inject(CloArgs a) {
  const int per = 2 * d * d;
  for (int t = threadIdx.x; t < a.count * per; t += 256) {
    const int k = t / per, u = t - k * per;
    const int side = u / (d * d), v = u - side * d * d;
    const int q = v / d, c = v - q * d;
    const int s = side ? a.second[k] : a.first[k];
    a.blk[(size_t)s * a.BS + 2 * a.B * a.B + (size_t)(a.col0 + k * d + q) * a.B + c] = a.A[(size_t)k * kCloLen(d) + u];
  }
  if (a.gsave && threadIdx.x == 0) {     // several closures may meet in one state: one thread, the order they were added in
    for (int k = 0; k < a.count; k++) {
      const double *rec = a.A + (size_t)k * kCloLen(d);
      for (int side = 0; side < 2; side++) {
        const int s = side ? a.second[k] : a.first[k];
        for (int c = 0; c < d; c++) {
          double acc = 0.0;
          for (int q = 0; q < d; q++) acc += rec[side * d * d + q * d + c] * rec[2 * d * d + q];
          a.gsave[(size_t)s * a.B + c] += acc;
        }
      }
    }
  }
}

// Y = (I + U Z)^-1 ([r | 0] - U X): one wave.  W = U [X | Z] from the solution columns of the closures' states, Cholesky of the
// symmetrised I + U Z in LDS, one lane per right-hand side for the two substitutions.
template <int d> __global__ void __launch_bounds__(64) k_clo_solve(CloArgs a) {
  constexpr int NM = kMaxRhs - 1;
  __shared__ double W[NM][kMaxRhs + 1];
  __shared__ double C[NM][NM + 1];
  const int lane = threadIdx.x, nc = a.count * d, R = a.R;
  for (int idx = lane; idx < nc * R; idx += 64) {
    const int p = idx / R, c = idx - p * R;
    const int k = p / d, q = p - k * d;
    const double *rec = a.A + (size_t)k * kCloLen(d);
    const double *xi = a.x + ((size_t)a.first[k] * R + c) * a.B, *xj = a.x + ((size_t)a.second[k] * R + c) * a.B;
    double acc = 0.0;
    for (int m = 0; m < d; m++) acc += rec[q * d + m] * xi[m];
    for (int m = 0; m < d; m++) acc += rec[d * d + q * d + m] * xj[m];
    W[p][c] = acc;
  }
  wave_lds_sync();
  for (int idx = lane; idx < nc * nc; idx += 64) {
    const int p = idx / nc, p2 = idx - p * nc;
    C[p][p2] = (p == p2 ? 1.0 : 0.0) + 0.5 * (W[p][a.col0 + p2] + W[p2][a.col0 + p]);
  }
  wave_lds_sync();
  for (int j = 0; j < nc; j++) {      // right-looking Cholesky, lower triangle
    double dd = C[j][j];
    if (!(dd > 0.0)) { if (lane == 0) *a.flag = 1; dd = 1.0; }
    const double l = sqrt(dd), linv = 1.0 / l;
    wave_lds_sync();
    for (int i = j + lane; i < nc; i += 64) C[i][j] = (i == j) ? l : C[i][j] * linv;
    wave_lds_sync();
    const int m = nc - j - 1;
    for (int idx = lane; idx < m * m; idx += 64) {
      const int i = j + 1 + idx / m, k = j + 1 + idx % m;
      if (k <= i) C[i][k] -= C[i][j] * C[k][j];
    }
    wave_lds_sync();
  }
  if (lane < a.ncols) {               // lane c: column c of [r | 0] - U X through L y = b, L^T z = y
    const int c = lane;
    double y[NM];
    for (int p = 0; p < nc; p++) {
      const int k = p / d, q = p - k * d;
      double v = (c == 0 ? a.A[(size_t)k * kCloLen(d) + 2 * d * d + q] : 0.0) - W[p][c];
      for (int m = 0; m < p; m++) v -= C[p][m] * y[m];
      y[p] = v / C[p][p];
    }
    for (int p = nc - 1; p >= 0; p--) {
      double v = y[p];
      for (int m = p + 1; m < nc; m++) v -= C[m][p] * y[m];
      y[p] = v / C[p][p];
    }
    for (int p = 0; p < nc; p++) a.Y[(size_t)p * a.ncols + c] = y[p];
  }
}

// X <- X + Z Y on the update column and the landmark columns of every state
template <int d> __global__ void __launch_bounds__(256) k_clo_correct(CloArgs a) {
  const int nc = a.count * d;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = tid / a.B, k = tid - s * a.B;
  if (s >= a.N) return;
  double *xs = a.x + (size_t)s * a.R * a.B;
  double z[kMaxRhs - 1];
  for (int q = 0; q < nc; q++) z[q] = xs[(size_t)(a.col0 + q) * a.B + k];
  for (int c = 0; c < a.ncols; c++) {
    double v = xs[(size_t)c * a.B + k];
    for (int q = 0; q < nc; q++) v += z[q] * a.Y[(size_t)q * a.ncols + c];
    xs[(size_t)c * a.B + k] = v;
  }
}

// ------------------------------------------------------------------ K3: assemble normal equations

template <typename T, typename TR = T> struct AsmArgs {
  int N, R;               // states, rhs columns (1 + border)
  const int *rowptr;      // N + 2 entries; rows of left state s: [rowptr[s], rowptr[s+1]); rowptr[-1] handled by s > 0
  const TR *rowLR, *rowE;
  const int *crowptr;     // compact rows (velocity-free Jacobians) of left state s: [crowptr[s], crowptr[s+1])
  const TR *rowC, *rowCE; // Mc x B = [d/dpose_left (B/2) | d/dpose_right (B/2)], Mc
  const TR *rowM;         // M x ld (border) or null
  const int *rowLm;       // landmark id per row or -1
  int ld;
  T *blk;                 // N records [D | O | G]
  T *gsave;               // N x B copy of the gradient column (for the LM model-fidelity test) or null
  T *halo_add;            // segment sharding: [RD | Rg] the rows of state N-1 owe to the next rank's first state
  // block size 6, fp64: the GP priors as structured records (kGp3*; K1 wrote no rows for them) or null
  const T *gps3;          // gp_count + 1 records, the last one all zeros
  const int *gpidx;       // record of the GP prior whose left state is s, or -1 (N + 2 entries)
  int gp_count;
  const T *Ud;            // chol_upper(Qc^-1), row-major 3 x 3
};

// Wave-cooperative assembly, every row fetched once.  A wave owns G - 1 = 64 / B - 1 consecutive states plus, in
// its first lane group, the state before them as a "ghost" that only contributes its rows' right halves.  The B
// lanes of a group each load ONE element of the left and of the right half of their state's current Jacobian row
// (coalesced 8-byte-per-lane loads, two rows in flight) and publish them in a per-wave LDS exchange buffer; a lane
// then reads back its own row's left half (-> D_s += L^T L, O_s += R^T L) and the right half published by the
// group of state s - 1 (-> D_s += R^T R).  Compared with letting every group re-read the rows of s - 1 this
// removes the second pass over the row table (2x HBM over-fetch measured, profiles/round1_v3) and a third of the
// loop iterations.  DS operations of one wave execute in order, so no barrier is needed; two buffers alternate.
template <typename T, int B, int WPB = 4, typename TR = T>
__global__ void __launch_bounds__(64 * WPB) k_assemble_ghost(AsmArgs<T, TR> a) {
  constexpr int G = 64 / B;                       // lane groups per wave (the first one is the ghost)
  static_assert(G >= 2, "needs at least one real state per wave");
  const int lane = threadIdx.x & 63;
  // blocks walk the chain from its END: the rows written last by the linearisation are the ones still in the
  // memory-side cache
  const int wave = (int)(((gridDim.x - 1 - blockIdx.x) * blockDim.x + threadIdx.x) >> 6);
  const int g = lane / B, c = lane - g * B;
  const int nstates = a.N + (a.halo_add ? 1 : 0);
  const int s = wave * (G - 1) + g - 1;           // ghost: the state before the wave's first
  const bool owner = (g < G) && s >= 0 && s < a.N;           // has rows to load
  const bool out = (g >= 1) && (g < G) && s < nstates;       // produces a block (or the halo addend)
  const int sc = out ? s : 0;
  const int gb = g * B, gp = (g >= 1 ? g - 1 : 0) * B;
  int rp_s = 0, n_own = 0;
  if (owner) {
    rp_s = a.rowptr[s];
    n_own = a.rowptr[s + 1] - rp_s;
  }
  // structured GP prior of the state (block size 6): its six rows are formed from the record below, the row loop starts behind them
  int gq = -1;
  if constexpr (B == 6 && std::is_same<T, double>::value && std::is_same<TR, double>::value) {
    if (a.gps3 && owner) gq = a.gpidx[s];
    if (gq >= 0) { rp_s += B; n_own -= B; }
  }
  int n_max = n_own;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n_max = max(n_max, __shfl_xor(n_max, o, 64));
  n_max = __builtin_amdgcn_readfirstlane(n_max);
  const int rp_prev = __shfl(rp_s, gp, 64);       // first row of state s - 1 (landmark columns of its rows)
  const int n_prev = __shfl(n_own, gp, 64);
  T D[B], O[B];
  T gsum = T(0);
#pragma unroll
  for (int k = 0; k < B; k++) { D[k] = T(0); O[k] = T(0); }
  constexpr int XS = 128 + ((G + 1) & ~1);        // per wave and parity: L[64] | R[64] | e[G]
  __shared__ T xch[WPB][2][XS];
  T *xw = &xch[threadIdx.x >> 6][0][0];
  const int BS = 2 * B * B + B * a.R;
  T *bp = a.blk + (size_t)sc * BS;
  const bool isblk = out && sc < a.N;
  // rhs columns: of this state's record, or -- for the virtual state behind the segment -- of the addend owed to the
  // next rank's first state (the rows of the last interval carry landmark columns too)
  T *gcol = isblk ? bp + 2 * B * B : ((out && a.halo_add) ? a.halo_add + B * B : nullptr);
  if (gcol) for (int r = 1; r < a.R; r++) gcol[r * B + c] = T(0);
  // loads are unconditional (row index clamped to the state's last row, the ghost group re-reads its R element
  // instead of an L element); validity is applied when the values are consumed.  A predicated load would make the
  // loaded value a phi with zero, and the copy that implements the phi waits for the load.
  const int n_last = max(n_own - 1, 0);
  auto ld = [&](int i, T &Lc, T &Rc, T &e) {
    const int rho = rp_s + min(i, n_last);
    const TR *row = a.rowLR + (size_t)rho * 2 * B;
    Lc = row[g >= 1 ? c : B + c];
    Rc = row[B + c];
    e = a.rowE[rho];
  };
  auto step = [&](int i, const bool valid, const T Lraw, const T Rraw, const T eraw) {
    const T Lc = (valid && g >= 1) ? Lraw : T(0), Rc = valid ? Rraw : T(0), e = valid ? eraw : T(0);
    T *buf = xw + (i & 1) * XS;
    buf[lane] = Lc;
    buf[64 + lane] = Rc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const T *Lrow = buf + gb;
    const T *Prow = buf + 64 + gp;                // right half of the current row of state s - 1
    const T Pc = Prow[c];
    const T ep = __shfl(e, gp, 64);               // e of state s - 1's current row (no branch: keeps the step one block)
#pragma unroll
    for (int k = 0; k < B; k++) {
      if (B > 8 && k == B / 2) __builtin_amdgcn_sched_barrier(0);   // halves the LDS read-back registers live at once
      const T Lk = Lrow[k];
      D[k] += Lc * Lk;
      O[k] += Rc * Lk;
      D[k] += Pc * Prow[k];
    }
    gsum -= Lc * e;
    gsum -= Pc * ep;
  };
  if constexpr (B == 6 && std::is_same<T, double>::value && std::is_same<TR, double>::value) {
    if (a.gps3) {
      // Lane c of the state's group holds column c of the six rows [A1 | kLt U | A3 | kRt U; 0 | kLb U | 0 | kRb U]: a pose lane
      // (c < 3) its column of A1 / A3 (and nothing in the bottom rows), a velocity lane column c - 3 of U times the record's
      // four coefficients.  States without a GP prior (and idle groups) read the all-zero record.
      const bool pc = c < 3;
      const int c3 = pc ? c : c - 3;
      const T *rec = a.gps3 + (size_t)(gq >= 0 ? gq : a.gp_count) * kGp3Len;
      T colL[3], colR[3], ew[6];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const T uk = a.Ud[3 * k + c3];
        colL[k] = pc ? rec[kGp3A1 + 3 * k + c3] : uk;
        colR[k] = pc ? rec[kGp3A3 + 3 * k + c3] : uk;
      }
      const T mLt = pc ? T(1) : rec[kGp3S + 0], mRt = pc ? T(1) : rec[kGp3S + 1];
      const T mLb = pc ? T(0) : rec[kGp3S + 2], mRb = pc ? T(0) : rec[kGp3S + 3];
#pragma unroll
      for (int i = 0; i < 6; i++) ew[i] = rec[kGp3E + i];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        step(i, true, (i < 3 ? mLt : mLb) * colL[i % 3], (i < 3 ? mRt : mRb) * colR[i % 3], ew[i]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // PF register sets rotate by full unrolling (no register copies: a copy of a set whose load is still in flight
  // would wait for it and cut the prefetch distance to one iteration -- measured 1000-2400 cycles of exposed
  // latency per row before this change)
  constexpr int PF = 3;
  T rL[PF], rR[PF], rE[PF];
  if (n_max > 0) {                                // (an empty row table may be a null pointer)
#pragma unroll
    for (int j = 0; j < PF; j++) {
      ld(j, rL[j], rR[j], rE[j]);
      __builtin_amdgcn_sched_barrier(0);          // issue order = consumption order (vmcnt counts in order)
    }
  }
  for (int i0 = 0; i0 < n_max; i0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; j++) {
      step(i0 + j, i0 + j < n_own, rL[j], rR[j], rE[j]);          // steps past n_max exchange zeros
      ld(i0 + j + PF, rL[j], rR[j], rE[j]);
      __builtin_amdgcn_sched_barrier(0);          // keeps the next step's LDS reads from being hoisted (VGPRs)
    }
  }
  // compact rows: the same exchange with half-width rows; only the pose rows / columns of D_s and O_s are touched
  {
    constexpr int Dh = B / 2;
    int crp = 0, cn = 0;
    if (owner) {
      crp = a.crowptr[s];
      cn = a.crowptr[s + 1] - crp;
    }
    int cmax = cn;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o, 64));
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    const int cn_last = max(cn - 1, 0);
    const int cc = c < Dh ? c : 0;
    auto ldc = [&](int i, T &Lc, T &Rc, T &e) {
      const int rho = crp + min(i, cn_last);
      const TR *row = a.rowC + (size_t)rho * B;
      Lc = row[g >= 1 ? cc : Dh + cc];
      Rc = row[Dh + cc];
      e = a.rowCE[rho];
    };
    T cL[PF], cR[PF], cE[PF];
    if (cmax > 0) {
#pragma unroll
      for (int j = 0; j < PF; j++) {
        ldc(j, cL[j], cR[j], cE[j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    auto cstep = [&](int i, const T Lraw, const T Rraw, const T eraw) {
      const bool valid = i < cn;
      const T Lc = (valid && g >= 1 && c < Dh) ? Lraw : T(0), Rc = (valid && c < Dh) ? Rraw : T(0), e = valid ? eraw : T(0);
      T *buf = xw + (i & 1) * XS;
      buf[lane] = Lc;
      buf[64 + lane] = Rc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const T *Lrow = buf + gb;
      const T *Prow = buf + 64 + gp;
      const T Pc = Prow[c];
      const T ep = __shfl(e, gp, 64);               // e of state s - 1's current row (no branch: keeps the step one block)
#pragma unroll
      for (int k = 0; k < Dh; k++) {
        const T Lk = Lrow[k];
        D[k] += Lc * Lk;
        O[k] += Rc * Lk;
        D[k] += Pc * Prow[k];
      }
      gsum -= Lc * e;
      gsum -= Pc * ep;
    };
    for (int i0 = 0; i0 < cmax; i0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; j++) {
        cstep(i0 + j, cL[j], cR[j], cE[j]);
        ldc(i0 + j + PF, cL[j], cR[j], cE[j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // landmark columns of the rows that have one (range / projection factors; a small minority of the rows): a second
  // pass that re-reads just those rows, own rows first, then the rows of state s - 1 -- kept out of the exchange
  // loop so that its steps stay branch-free
  if (a.rowM && gcol) {
    for (int i = 0; i < n_own; i++) {
      const int rho = rp_s + i;
      const int lm = a.rowLm[rho];
      if (lm < 0) continue;
      const T Lc = a.rowLR[(size_t)rho * 2 * B + c];
      for (int q = 0; q < a.ld; q++) gcol[(1 + lm * a.ld + q) * B + c] += Lc * a.rowM[(size_t)rho * a.ld + q];
    }
    for (int i = 0; i < n_prev; i++) {
      const int rho = rp_prev + i;
      const int lm = a.rowLm[rho];
      if (lm < 0) continue;
      const T Pc = a.rowLR[(size_t)rho * 2 * B + B + c];
      for (int q = 0; q < a.ld; q++) gcol[(1 + lm * a.ld + q) * B + c] += Pc * a.rowM[(size_t)rho * a.ld + q];
    }
  }
  if (out && sc == a.N) {   // halo: what the rows of state N-1 owe the neighbour's first state
#pragma unroll
    for (int k = 0; k < B; k++) a.halo_add[c * B + k] = D[k];
    a.halo_add[B * B + c] = gsum;
  }
  // [D | O | g] leave through a per-wave LDS image of the G - 1 records, then as 16-byte-per-lane stores that are
  // contiguous within a record.  Writing a lane's row of D directly (16-byte pieces at a B * 8 byte stride, one write
  // request per lane) cost 0.085 of the kernel's 0.20 ms on the benchmark chain.
  constexpr int RS = 2 * B * B + B;
  __shared__ T stg[WPB][(G - 1) * RS];
  T *sw = &stg[threadIdx.x >> 6][0];
  if (isblk) {
    T *r = sw + (g - 1) * RS;
#pragma unroll
    for (int k = 0; k < B; k++) { r[c * B + k] = D[k]; r[B * B + c * B + k] = O[k]; }
    r[2 * B * B + c] = gsum;
    if (a.gsave) a.gsave[(size_t)sc * B + c] = gsum;
  }
  wave_lds_sync();
  typedef T V2 __attribute__((ext_vector_type(2)));
  const int s_first = wave * (G - 1);
  constexpr int NPAIR = (G - 1) * RS / 2;
  static_assert(RS % 2 == 0, "records are copied in pairs");
#pragma unroll
  for (int t = 0; t < (NPAIR + 63) / 64; t++) {
    const int q = t * 64 + lane;
    const int j = (2 * q) / RS, off = 2 * q - j * RS;
    if (q < NPAIR && s_first + j < a.N)
      // non-temporal: the records are next touched by another kernel, and kept out of the caches they leave the rows
      // this kernel still has to read where the linearisation left them (0.150 -> 0.133 ms; the elimination pays 10 us)
      __builtin_nontemporal_store(*reinterpret_cast<const V2 *>(sw + 2 * q), reinterpret_cast<V2 *>(a.blk + (size_t)(s_first + j) * BS + off));
  }
}

// ------------------------------------------------------------------ trajectory queries

template <typename T> struct QueryArgs {
  const double *pose, *vel;   // SoA states
  int stride, count;
  const int *left;       // query q lies in the interval (left[q], left[q] + 1)
  const double *coef;        // count x 4: l11, l12, p11, p12 for (dt[q], tau[q])
  int vw;                // Pose3 only: velocities are world-frame [v; w]
  T *out;                // count x pose_dim (AoS, the layout of gpslam_hip_get_states' rows)
  T *out_H;              // JAC: count x 4 x d x d = H1..H4 of interpolatePose (GaussianProcessInterpolatorPose3.h:82-98)
};

// Batched GaussianProcessInterpolator{Linear,Pose2,Pose3,Rot3}::interpolatePose without Jacobians (the public
// query use of the interpolators, gpslam.h:57-86): thread per query.
template <typename T, int MF, bool JAC = false>
__global__ void __launch_bounds__(128) k_interp_query(QueryArgs<T> a) {
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.count) return;
  const int i = a.left[q];
  const ICoef<T> k = {T(a.coef[4 * (size_t)q]), T(a.coef[4 * (size_t)q + 1]), T(a.coef[4 * (size_t)q + 2]), T(a.coef[4 * (size_t)q + 3])};
  T p1[pd], p2[pd], v1[d], v2[d];
#pragma unroll
  for (int c = 0; c < pd; c++) { p1[c] = a.pose[(size_t)c * a.stride + i]; p2[c] = a.pose[(size_t)c * a.stride + i + 1]; }
#pragma unroll
  for (int c = 0; c < d; c++) { v1[c] = a.vel[(size_t)c * a.stride + i]; v2[c] = a.vel[(size_t)c * a.stride + i + 1]; }
  if constexpr (MF == POSE3) {
    if (a.vw) {
      T b1[6], b2[6];
      vw_to_vb(p1, v1, b1);
      vw_to_vb(p2, v2, b2);
#pragma unroll
      for (int c = 0; c < 6; c++) { v1[c] = b1[c]; v2[c] = b2[c]; }
    }
  }
  T *o = a.out + (size_t)q * pd;
  T *oh = JAC ? a.out_H + (size_t)q * 4 * d * d : nullptr;
  if constexpr (MF == LINEAR2 || MF == LINEAR3) {
    // p(tau) = Lambda_1 [p1; v1] + Psi_1 [p2; v2]   (GaussianProcessInterpolatorLinear.h:70-90)
#pragma unroll
    for (int c = 0; c < d; c++) o[c] = k.l11 * p1[c] + k.l12 * v1[c] + k.p11 * p2[c] + k.p12 * v2[c];
    if (JAC) {     // H_k = the d x d blocks of Lambda_1, Psi_1 (:83-86): scalar multiples of I for the shared Qc
      const T cf[4] = {k.l11, k.l12, k.p11, k.p12};
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int r = 0; r < d; r++)
#pragma unroll
          for (int c = 0; c < d; c++) oh[(m * d + r) * d + c] = (r == c) ? cf[m] : T(0);
    }
  } else if constexpr (MF == POSE2) {
    Interp3Out<T, JAC> jo;
    const SE2<T> r = interp_pose2<T, JAC>(p1, v1, p2, v2, k, jo);
    o[0] = r.x; o[1] = r.y; o[2] = r.th;
    if (JAC) { put_m3(jo.H1, oh, 3, 0, 0); put_m3(jo.H2, oh + 9, 3, 0, 0); put_m3(jo.H3, oh + 18, 3, 0, 0); put_m3(jo.H4, oh + 27, 3, 0, 0); }
  } else if constexpr (MF == ROT3) {
    Interp3Out<T, JAC> jo;
    const M3<T> r = interp_rot3<T, JAC>(p1, v1, p2, v2, k, jo);
#pragma unroll
    for (int c = 0; c < 9; c++) o[c] = r.m[c];
    if (JAC) { put_m3(jo.H1, oh, 3, 0, 0); put_m3(jo.H2, oh + 9, 3, 0, 0); put_m3(jo.H3, oh + 18, 3, 0, 0); put_m3(jo.H4, oh + 27, 3, 0, 0); }
  } else if constexpr (MF == ROT3_BIAS) {
    // the rotation is interpolated (GaussianProcessInterpolatorRot3); the bias slot returns the left state's bias, the one
    // an AHRSFactor over this interval uses
    Interp3Out<T, JAC> jo;
    const M3<T> r = interp_rot3<T, JAC>(p1, v1, p2, v2, k, jo);
#pragma unroll
    for (int c = 0; c < 9; c++) o[c] = r.m[c];
#pragma unroll
    for (int c = 0; c < 3; c++) o[9 + c] = p1[9 + c];
    if (JAC) {
#pragma unroll
      for (int c = 0; c < 4 * 36; c++) oh[c] = T(0);
      put_m3(jo.H1, oh, 6, 0, 0); put_m3(jo.H2, oh + 36, 6, 0, 0); put_m3(jo.H3, oh + 72, 6, 0, 0); put_m3(jo.H4, oh + 108, 6, 0, 0);
#pragma unroll
      for (int c = 0; c < 3; c++) oh[(3 + c) * 6 + 3 + c] = T(1);
    }
  } else {
    Interp6Out<T, JAC> jo;
    const SE3<T> r = interp_pose3<T, JAC>(p1, v1, p2, v2, k, jo);
#pragma unroll
    for (int c = 0; c < 9; c++) o[c] = r.R.m[c];
    o[9] = r.t.x; o[10] = r.t.y; o[11] = r.t.z;
    if (JAC) {
#pragma unroll
      for (int c = 0; c < 4 * 36; c++) oh[c] = T(0);
      put_bl6(jo.H1, oh, 6, 0, 0); put_bl6(jo.H2, oh + 36, 6, 0, 0); put_bl6(jo.H3, oh + 72, 6, 0, 0); put_bl6(jo.H4, oh + 108, 6, 0, 0);
      if (a.vw) {   // chain rule through convertVWtoVb, one 12-wide [d/dpose | d/dVb] segment per output row and state
#pragma unroll
        for (int rr = 0; rr < 6; rr++) {
          T seg[12];
#pragma unroll
          for (int c = 0; c < 6; c++) { seg[c] = oh[rr * 6 + c]; seg[6 + c] = oh[36 + rr * 6 + c]; }
          vw_row_transform(p1, v1, seg);
#pragma unroll
          for (int c = 0; c < 6; c++) { oh[rr * 6 + c] = seg[c]; oh[36 + rr * 6 + c] = seg[6 + c]; }
#pragma unroll
          for (int c = 0; c < 6; c++) { seg[c] = oh[72 + rr * 6 + c]; seg[6 + c] = oh[108 + rr * 6 + c]; }
          vw_row_transform(p2, v2, seg);
#pragma unroll
          for (int c = 0; c < 6; c++) { oh[72 + rr * 6 + c] = seg[c]; oh[108 + rr * 6 + c] = seg[6 + c]; }
        }
      }
    }
  }
}

// Batched GaussianProcessInterpolatorLinear<D>::interpolateVelocity (GaussianProcessInterpolatorLinear.h:106-126): the bottom
// block rows of Lambda and Psi.  coef = (l21, l22, p21, p22) per query; H_k = those scalars times I (:117-120).
template <typename T, int MF>
__global__ void __launch_bounds__(128) k_interp_velocity(QueryArgs<T> a) {
  constexpr int d = MTraits<MF>::d;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.count) return;
  if constexpr (MF == LINEAR2 || MF == LINEAR3) {
    const int i = a.left[q];
    const T cf[4] = {T(a.coef[4 * (size_t)q]), T(a.coef[4 * (size_t)q + 1]), T(a.coef[4 * (size_t)q + 2]), T(a.coef[4 * (size_t)q + 3])};
#pragma unroll
    for (int c = 0; c < d; c++) {
      const T p1 = a.pose[(size_t)c * a.stride + i], p2 = a.pose[(size_t)c * a.stride + i + 1];
      const T v1 = a.vel[(size_t)c * a.stride + i], v2 = a.vel[(size_t)c * a.stride + i + 1];
      a.out[(size_t)q * d + c] = cf[0] * p1 + cf[1] * v1 + cf[2] * p2 + cf[3] * v2;
    }
    if (a.out_H) {
      T *oh = a.out_H + (size_t)q * 4 * d * d;
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int r = 0; r < d; r++)
#pragma unroll
          for (int c = 0; c < d; c++) oh[(m * d + r) * d + c] = (r == c) ? cf[m] : T(0);
    }
  }
}

// getBodyCentricVb / getBodyCentricVs (gpslam/gp/Pose3utils.cpp:17-24; Barfoot14tro eq. 25), batched: thread per pose pair.
//   which 0: Vb = Logmap(pose1^-1 pose2) / dt      which 1: Vs = Logmap(pose2 pose1^-1) / dt
template <typename T>
__global__ void __launch_bounds__(128) k_body_velocity(const double *p1, const double *p2, const double *dt, int count, int which, double *out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= count) return;
  T a[12], b[12];
#pragma unroll
  for (int k = 0; k < 12; k++) { a[k] = T(p1[(size_t)q * 12 + k]); b[k] = T(p2[(size_t)q * 12 + k]); }
  const SE3<T> A = as_se3(a), Bm = as_se3(b);
  const SE3<T> h = which ? se3_compose(Bm, se3_inverse(A)) : se3_between(A, Bm);
  const V6<T> r = se3_log(h);
  const T s = T(1) / T(dt[q]);
  double *o = out + (size_t)q * 6;
  o[0] = s * r.w.x; o[1] = s * r.w.y; o[2] = s * r.w.z; o[3] = s * r.v.x; o[4] = s * r.v.y; o[5] = s * r.v.z;
}

// ------------------------------------------------------------------ K4: partitioned block Gauss-Jordan




template <typename T> struct FwdArgs {
  T *blk;            // n records of this level, eliminated in place
  const T *add;      // n addend records [RD | Rg] or null (level 0)
  T *up_blk;         // records of the next level (one per chunk)
  T *up_add;         // addends of the next level (nchunks + 1)
  int n, m, R;
  int no_sep;        // 1: top level, a single chunk with no separator (plain sequential elimination)
  int last_has_right;// the last chunk has a right separator outside this level (next rank)
  const T *remote_add; // [RD | Rg] already owed to that outside separator by lower levels / the assembly, or null
  T lambda;          // LM damping added to the diagonal of D while loading (level 0 only)
  int *flag;         // set to 1 if a pivot is not positive
  // k_fused_level0 only.  tail = 1: the wave's four chunk separators are reduced to the first of them at the end of the
  // kernel (two sub-levels of cyclic reduction, cr_step.hpp) -- the level of groups of four that upper.hip would otherwise
  // run as a launch of its own.  up_blk / up_add then belong to the level ABOVE that one (one record per workgroup), and
  // the factor records of the three eliminated separators go to l1_blk (the records of the level in between).
  T *l1_blk;
  int tail;
};

// One wave per chunk.  Panel columns: [ D~ (B) | O^T (B) | F (B) | rhs (R) ], one column per lane.
// Block j:  D~_j x_j + O_j^T x_{j+1} + F_j x_sep = g~_j.  Gauss-Jordan on D~_j turns the other columns into
// U_j = D~^-1 O_j^T, V_j = D~^-1 F_j, Y_j = D~^-1 g~_j (stored in place for the back-substitution) and
//   D~_{j+1} = D_{j+1} - O_j U_j,  F_{j+1} = -O_j V_j,  g~_{j+1} = g_{j+1} - O_j Y_j,
//   separator:  D_sep -= F_j^T V_j,  g_sep -= F_j^T Y_j.
// The lanes that held O^T become the D~ lanes of the next block (roles swap), so nothing moves.
// FAST (4B + 2R <= 64): the separator sums live in B + R otherwise idle lanes ("A lanes") and are formed by the
// very same instruction stream as the next-block update -- every lane executes  nxt -= Mat * col, the update lanes
// with Mat = O_j and their own column, the A lanes with Mat = F_j^T and a V_j / Y_j column fetched from LDS --
// which removes a whole B x B multiply-accumulate pass per block and keeps the live state at two B-vectors per lane.
template <typename T, int B, bool FAST>
__global__ void __launch_bounds__(64, 4) k_chunk_forward(FwdArgs<T> a) {
  const int R = a.R;
  const int BS = 2 * B * B + B * R;
  const int AS = B * B + B * R;
  const int c = blockIdx.x;
  const int s = c * a.m;
  const int e = min(s + a.m, a.n);
  const int lane = threadIdx.x;
  const bool has_sep = !a.no_sep;
  const bool right_exists = (e < a.n) || (a.last_has_right != 0);
  // FAST: the eliminated record [V_j | U_j | Y_j] is laid out in LDS exactly as it goes to memory: the A lanes take their
  // V / Y operands from there, and the wave writes it back as contiguous 16-byte pieces (3 wave stores per Pose3 block
  // instead of 6 stores of 16-byte fragments at a 96-byte stride)
  constexpr int kRecLen = FAST ? 2 * B * B + B * ((64 - 4 * B) / 2) : 2;
  __shared__ __attribute__((aligned(16))) T ldsM[2 * B * B];   // [O_j | F_j]: one array, so that Mat below is an offset
  T *const ldsO = ldsM, *const ldsF = ldsM + B * B;
  __shared__ __attribute__((aligned(16))) T ldsRec[kRecLen];
  if (FAST && a.no_sep) {   // no spike columns at the top level: keep the V part of the stored records defined
    for (int q = threadIdx.x; q < B * B; q += 64) ldsRec[q] = T(0);
  }
  int dbase = 0, obase = B;
  const int cF = lane - 2 * B;
  const bool isF = has_sep && cF >= 0 && cF < B;
  const int cR = lane - 3 * B;
  const bool isR = cR >= 0 && cR < R;
  const int cA = lane - (3 * B + R);                      // < B: column of the separator's D; >= B: rhs column
  const bool isA = FAST && has_sep && cA >= 0 && cA < B + R;
  T col[B];
  T nxt[B];                 // update lanes: operand of block j+1;  A lanes: minus the separator sums (persistent)
  T acc[FAST ? 1 : B];      // !FAST: separator sums in the F / rhs lanes
#pragma unroll
  for (int k = 0; k < B; k++) { col[k] = T(0); nxt[k] = T(0); }
#pragma unroll
  for (int k = 0; k < (FAST ? 1 : B); k++) acc[k] = T(0);

  auto load_D_row = [&](int j, int r, T *out) {
    const T *p = a.blk + (size_t)j * BS + r * B;
#pragma unroll
    for (int k = 0; k < B; k++) out[k] = p[k];
    if (a.add) {
      const T *q = a.add + (size_t)j * AS + r * B;
#pragma unroll
      for (int k = 0; k < B; k++) out[k] += q[k];
    }
#pragma unroll
    for (int k = 0; k < B; k++)
      if (k == r) out[k] += a.lambda;
  };
  auto load_O_row = [&](int j, int r, T *out) {
    const T *p = a.blk + (size_t)j * BS + B * B + r * B;
#pragma unroll
    for (int k = 0; k < B; k++) out[k] = p[k];
  };
  auto load_G_col = [&](int j, int r, T *out) {
    const T *p = a.blk + (size_t)j * BS + 2 * B * B + r * B;
#pragma unroll
    for (int k = 0; k < B; k++) out[k] = p[k];
    if (a.add) {
      const T *q = a.add + (size_t)j * AS + B * B + r * B;
#pragma unroll
      for (int k = 0; k < B; k++) out[k] += q[k];
    }
  };

  const int j0 = has_sep ? s + 1 : s;
  if (isA) {
    // the A lanes start from the separator's own row of D (or rhs column) and subtract the sums from it: the loads
    // fly with the first block's instead of being an exposed round trip at the end (one per hierarchy level)
    if (cA < B) load_D_row(s, cA, nxt);
    else load_G_col(s, cA - B, nxt);
  }
  if (j0 < e) {
    if (lane < B) load_D_row(j0, lane, col);
    else if (lane < 2 * B) load_O_row(j0, lane - B, col);
    else if (isF) {
      const T *p = a.blk + (size_t)s * BS + B * B;  // F_first = O_s (acts on x_sep in row s+1)
#pragma unroll
      for (int k = 0; k < B; k++) col[k] = p[k * B + cF];
    } else if (isR) load_G_col(j0, cR, col);
  }

  // The first block's operands are awaited HERE rather than inside the loop: with loads pending on `col` at the loop
  // entry the compiler's wait-count bookkeeping keeps a (partial) wait for memory at the top of every iteration,
  // which then catches the operand prefetch issued just above it.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  for (int j = j0; j < e; ++j) {
    const bool last = (j == e - 1);
    const int cD = lane - dbase, cO = lane - obase;
    const bool isD = cD >= 0 && cD < B, isO = cO >= 0 && cO < B;
    // Operands of block j+1, fetched early so that the HBM latency hides under the elimination.  The loads are
    // UNCONDITIONAL (every lane reads 12 doubles from a valid address of its role, or block j again) and neither the
    // addend of an upper level nor the damping is folded in here: a load under a branch merges with the zero of the
    // other path, an add consumes the value, and either makes the compiler wait for the load right where it was
    // issued -- 1.4 us per block step of a chunk that runs alone (measured with s_memtime), i.e. on every upper level.
    T araw[B];
    const bool use_n = !last && (isD || isO || isR);
    const bool use_a = !last && a.add != nullptr && (isO || isR);
    {
      const int jn = last ? j : j + 1;
      const T *pn = a.blk + (size_t)jn * BS + (isD ? B * B + cD * B : isO ? cO * B : isR ? 2 * B * B + cR * B : 0);
      const T *pq = a.add ? a.add + (size_t)jn * AS + (isR ? B * B + cR * B : isO ? cO * B : 0) : pn;
      if (!isA) {
#pragma unroll
        for (int k = 0; k < B; k++) nxt[k] = pn[k];
      }
#pragma unroll
      for (int k = 0; k < B; k++) araw[k] = pq[k];
    }
    if (isO) {
#pragma unroll
      for (int k = 0; k < B; k++) ldsO[cO * B + k] = col[k];
    }
    if (isF) {
#pragma unroll
      for (int k = 0; k < B; k++) ldsF[cF * B + k] = col[k];
    }
    // Gauss-Jordan on the D~ columns; every lane applies the same row operations to its column
    const int db = __builtin_amdgcn_readfirstlane(dbase);
#pragma unroll
    for (int k = 0; k < B; k++) {
      // the whole pivot column goes to SGPRs first, then the row operations: back-to-back v_readlane / v_fma pairs
      // on one SGPR pair cost a hazard nop each and serialise on the write-after-read of that pair
      const T p = lane_bcast(col[k], db + k);
      T mlt[B];
#pragma unroll
      for (int i = 0; i < B; i++) mlt[i] = (i != k) ? lane_bcast(col[i], db + k) : T(0);
      if (!(p > T(0)) && lane == 0) *a.flag = 1;
      const T rowk = col[k] * fast_rcp(p);
#pragma unroll
      for (int i = 0; i < B; i++)
        if (i != k) col[i] -= mlt[i] * rowk;
      col[k] = rowk;
    }
    T *bp = a.blk + (size_t)j * BS;
    T *rp = FAST ? ldsRec : bp;                                       // where this block's factors are assembled
    if (isO) {
#pragma unroll
      for (int k = 0; k < B; k++) rp[B * B + cO * B + k] = col[k];  // U_j, column cO
    } else if (isF) {
#pragma unroll
      for (int k = 0; k < B; k++) rp[cF * B + k] = col[k];          // V_j, column cF
    } else if (isR) {
#pragma unroll
      for (int k = 0; k < B; k++) rp[2 * B * B + cR * B + k] = col[k];  // Y_j
    }
    wave_lds_sync();
    if (FAST) {
      typedef T V2 __attribute__((ext_vector_type(2)));
      const V2 *src = reinterpret_cast<const V2 *>(ldsRec);
      V2 *dst = reinterpret_cast<V2 *>(bp);
      for (int q = lane; q < BS / 2; q += 64) dst[q] = src[q];
    }
    if (!FAST && has_sep && (isF || isR)) {   // wide borders: separator sums in a pass of their own
#pragma unroll
      for (int q = 0; q < B; q++) {
        T sacc = T(0);
#pragma unroll
        for (int k = 0; k < B; k++) sacc += ldsF[q * B + k] * col[k];
        acc[q] += sacc;
      }
    }
    // the fetched operands have landed by now: lanes without an operand start from zero, addend and damping join in
    // the summation order of the unfused form (the A lanes keep their running sums)
    if (!isA) {
#pragma unroll
      for (int k = 0; k < B; k++) {
        T v = use_n ? nxt[k] : T(0);
        v += use_a ? araw[k] : T(0);
        v += (use_n && isO && k == cO) ? a.lambda : T(0);
        nxt[k] = v;
      }
    }
    // one B x B product per lane:  nxt -= Mat * col
    const int moff = isA ? B * B : 0;
    if (isA) {
      const T *vp = ldsRec + (cA < B ? cA * B : 2 * B * B + (cA - B) * B);   // column of V_j, or of Y_j
#pragma unroll
      for (int k = 0; k < B; k++) col[k] = vp[k];
    } else if (!(isO || isF || isR)) {
#pragma unroll
      for (int k = 0; k < B; k++) col[k] = T(0);     // D lanes (and spare lanes) take the operand unchanged
    }
    {
      // rows of Mat are wave-uniform (broadcast) 16-byte LDS reads; row r + 1 is in flight while row r is used, so
      // the loop waits on LDS once per row instead of once per operand pair
      typedef T V2 __attribute__((ext_vector_type(2)));
      const V2 *M2 = reinterpret_cast<const V2 *>(ldsM + moff);
      V2 cur[B / 2], nx[B / 2];
#pragma unroll
      for (int k = 0; k < B / 2; k++) cur[k] = M2[k];
#pragma unroll
      for (int r = 0; r < B; r++) {
        if (r + 1 < B) {
#pragma unroll
          for (int k = 0; k < B / 2; k++) nx[k] = M2[(r + 1) * (B / 2) + k];
        }
        T sacc = T(0);
#pragma unroll
        for (int k = 0; k < B / 2; k++) {
          sacc += cur[k].x * col[2 * k];
          sacc += cur[k].y * col[2 * k + 1];
        }
        nxt[r] -= sacc;
#pragma unroll
        for (int k = 0; k < B / 2; k++) cur[k] = nx[k];
      }
    }
    if (!last) {
      if (!isA) {
#pragma unroll
        for (int r = 0; r < B; r++) col[r] = nxt[r];
      }
      const int tswap = dbase; dbase = obase; obase = tswap;
    } else if (right_exists && has_sep) {
      T *ua = a.up_add + (size_t)(c + 1) * AS;
      const T *ra = (e == a.n) ? a.remote_add : nullptr;   // carry what is already owed to the outside separator
      if (isO) {
#pragma unroll
        for (int k = 0; k < B; k++) ua[cO * B + k] = (ra ? ra[cO * B + k] : T(0)) + nxt[k];              // -O U  (symmetric)
      } else if (isR) {
#pragma unroll
        for (int k = 0; k < B; k++) ua[B * B + cR * B + k] = (ra ? ra[B * B + cR * B + k] : T(0)) + nxt[k];      // -O Y
      } else if (isF) {
        T *uo = a.up_blk + (size_t)c * BS + B * B;
#pragma unroll
        for (int r = 0; r < B; r++) uo[r * B + cF] = nxt[r];             // C = -O V, couples sep c -> sep c+1
      }
    }
    wave_lds_sync();
  }

  if (has_sep) {
    T *ub = a.up_blk + (size_t)c * BS;
    // separator block of the next level: D_sep - sum F^T V (row = column by symmetry), g_sep - sum F^T Y
    const bool wD = FAST ? (isA && cA < B) : isF;
    const bool wG = FAST ? (isA && cA >= B) : isR;
    const int rD = FAST ? cA : cF, rG = FAST ? cA - B : cR;
    if (wD) {
      T dr[B];
      if (!FAST) load_D_row(s, rD, dr);
#pragma unroll
      for (int k = 0; k < B; k++) ub[rD * B + k] = FAST ? nxt[k] : dr[k] - acc[FAST ? 0 : k];
    } else if (wG) {
      T gr[B];
      if (!FAST) load_G_col(s, rG, gr);
#pragma unroll
      for (int k = 0; k < B; k++) ub[2 * B * B + rG * B + k] = FAST ? nxt[k] : gr[k] - acc[FAST ? 0 : k];
    }
    if (j0 >= e) {  // chunk without interior: the separator keeps its original coupling
      if (lane < B) {
        T orow[B];
        load_O_row(s, lane, orow);
#pragma unroll
        for (int k = 0; k < B; k++) ub[B * B + lane * B + k] = right_exists ? orow[k] : T(0);
        if (right_exists) {
          T *ua = a.up_add + (size_t)(c + 1) * AS;
          const T *ra = (e == a.n) ? a.remote_add : nullptr;
#pragma unroll
          for (int k = 0; k < B; k++) ua[lane * B + k] = ra ? ra[lane * B + k] : T(0);
        }
      }
      if (isR && right_exists) {
        T *ua = a.up_add + (size_t)(c + 1) * AS;
        const T *ra = (e == a.n) ? a.remote_add : nullptr;
#pragma unroll
        for (int k = 0; k < B; k++) ua[B * B + cR * B + k] = ra ? ra[B * B + cR * B + k] : T(0);
      }
    } else if (!right_exists) {
      if (lane < B) {
#pragma unroll
        for (int k = 0; k < B; k++) ub[B * B + lane * B + k] = T(0);
      }
    }
  }
}

// ---------------------------------------------------------------- row-layout forward elimination (Pose3, R = 1)
//
// Same elimination as k_chunk_forward, same records in and out, different mapping: FOUR chunks per wave, one per
// 16-lane DPP row, lane r < 12 of a row holding ROW r of the panel [ D~_j | O_j^T | F_j | g~_j ].  The multipliers
// D~[r][k] of a Gauss-Jordan step are then lane-local and only the pivot row travels -- by `v_mov_b32_dpp
// row_newbcast:k` (full VALU rate, no SGPR round trip) instead of a v_readlane per multiplier.  The updates
//   D~_{j+1} = D_{j+1} - O_j U_j,  F_{j+1} = -O_j V_j,  g~_{j+1} = g_{j+1} - O_j Y_j
// take row r of O_j as local scalars against broadcast rows of U_j / V_j / Y_j; the separator sums need F_j^T,
// which is carried alongside as G_j = F_j^T with  G_{j+1} = -G_j U_j  (V_j^T = G_j D~_j^-1 by symmetry):
//   D_sep -= G_j V_j,  g_sep -= G_j Y_j.
// Records enter and leave as contiguous 16-byte pieces through a per-row LDS image (which also provides the
// transposed reads of O).  ~660 VALU instructions per chunk-step against ~830 (288 of them v_readlane) before.
// Round 3: the block size is a template parameter (12: SE(3); 6: SE(2), SO(3), 3-D linear; 4: 2-D linear) -- the planar /
// rotation chains had kept the column-layout kernel (k_chunk_forward), at 1.7 TB/s of record traffic where this one is bound
// by it.
template <int B>
__global__ void __launch_bounds__(64, 2) k_chunk_forward_rows(FwdArgs<double> a) {
  constexpr int BS = 2 * B * B + B, AS = B * B + B;   // R == 1
  constexpr int NPC = BS / 2;                                  // 16-byte pieces of a record
  constexpr int NV = (NPC + 15) / 16;                          // pieces per lane of a 16-lane row
  typedef double V2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x, grp = lane >> 4, r = lane & 15;
  const int c = blockIdx.x * 4 + grp;
  const int nch = (a.n + a.m - 1) / a.m;
  const bool valid = c < nch;
  const int s = valid ? c * a.m : 0;
  const int e = valid ? min(s + a.m, a.n) : 0;
  const int j0 = s + 1;
  const bool rowlane = r < B;
  const int rr = rowlane ? r : 0;                              // idle lanes shadow row 0 (they never store)
  const bool right_exists = (e < a.n) || (a.last_has_right != 0);
  const bool has_int = valid && j0 < e;
  const double lambda = a.lambda;
  __shared__ __attribute__((aligned(16))) double L[4 * BS];   // one record image per 16-lane row
  // Every LDS access below is  L[opaque per-lane offset + compile-time constant]: left to itself the compiler derives the
  // row / column / piece addresses from one another by strength reduction and ends up with a dozen VGPRs holding the
  // same address (40 VGPRs, which spilled -- and every scratch reload drains vmcnt, i.e. waits for the factor stores)
  int ro = grp * BS + rr * B;     // row r:      L[ro + k]
  int co = grp * BS + rr;         // column r:   L[co + k * B]
  int po = grp * BS + 2 * r;      // 16-byte piece q * 16 + r of the image: L[po + 32 * q]
  int dg = grp * BS + rr * B + rr;
  asm volatile("" : "+v"(ro), "+v"(co), "+v"(po), "+v"(dg));

  // record j as 16-byte pieces, unconditionally (clamped): whether it exists is applied when it is consumed
  auto load_rec = [&](int j, V2 *v) {
    const V2 *src = reinterpret_cast<const V2 *>(a.blk + (size_t)min(max(j, 0), a.n - 1) * BS);
#pragma unroll
    for (int t = 0; t < NV; t++) v[t] = src[min(t * 16 + r, NPC - 1)];
  };
  auto put_rec = [&](const V2 *v, bool ok) {
#pragma unroll
    for (int t = 0; t < NV; t++) {
      const int idx = t * 16 + r;
      V2 w = v[t];
      if (!ok) { w.x = 0.0; w.y = 0.0; }
      if (idx < NPC) *reinterpret_cast<V2 *>(&L[po + 32 * t]) = w;
    }
  };
  auto add_lambda = [&](bool ok) {
    if (lambda != 0.0) {
      if (rowlane && ok) L[dg] += lambda;
      wave_lds_sync();
    }
  };

  V2 pa[NV], pb[NV];
  load_rec(s, pa);
  load_rec(j0, pb);
  double Dr[B], Or[B], Fr[B], Gr[B], Ar[B];
  double gr, as_;
  put_rec(pa, valid);
  wave_lds_sync();
  add_lambda(valid);
#pragma unroll
  for (int k = 0; k < B; k++) {
    Ar[k] = L[ro + k];                       // row r of D_sep
    Fr[k] = L[ro + B * B + k];               // F_{s+1} = O_s: row r
    Gr[k] = L[co + B * B + k * B];           // G_{s+1} = O_s^T: row r
  }
  as_ = L[co + 2 * B * B];
  wave_lds_sync();
  put_rec(pb, has_int);
  wave_lds_sync();
  add_lambda(has_int);
#pragma unroll
  for (int k = 0; k < B; k++) {
    Dr[k] = L[ro + k];                       // row r of D_j
    Or[k] = L[co + B * B + k * B];           // row r of O_j^T
  }
  gr = L[co + 2 * B * B];
  // (the image of record j stays in LDS through the elimination: row r of O_j is fetched from it afterwards, which
  //  keeps 24 VGPRs free while the prefetched record j + 1 occupies 40)

  if (valid && !has_int && rowlane) {        // chunk without interior: the separator keeps its original coupling
    double *ub = a.up_blk + (size_t)c * BS;
#pragma unroll
    for (int k = 0; k < B; k++) {
      ub[r * B + k] = Ar[k];
      ub[B * B + r * B + k] = right_exists ? Fr[k] : 0.0;
    }
    ub[2 * B * B + r] = as_;
    if (right_exists) {
      double *ua = a.up_add + (size_t)(c + 1) * AS;
      const double *ra = (e == a.n) ? a.remote_add : nullptr;
#pragma unroll
      for (int k = 0; k < B; k++) ua[r * B + k] = ra ? ra[r * B + k] : 0.0;
      ua[B * B + r] = ra ? ra[B * B + r] : 0.0;
    }
  }

  // lane 0 belongs to the wave's first chunk, which is never shorter than the others
  const int steps = __builtin_amdgcn_readfirstlane(max(e - j0, 0));
  for (int t = 0; t < steps; t++) {
    const int j = j0 + t;
    const bool live = j < e, lastb = (j == e - 1), nextlive = (j + 1 < e);
    V2 pf[NV];
    load_rec(j + 1, pf);                     // record j + 1 flies under the elimination of block j
    __builtin_amdgcn_sched_barrier(0);
    // Gauss-Jordan on D~_j by row operations; the pivot row stays unscaled until the end (scaling commutes)
    double invs = 1.0;
    static_for<0, B>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      const double piv = row_bcast<k>(Dr[k]);
      const double inv = fast_rcp(piv);
      const bool isk = (r == k);
      invs = isk ? inv : invs;
      const double nmp = isk ? 0.0 : -(Dr[k] * inv);
      // row r -= (D~[r][k] / pivot) * row k, the pivot row fused into the multiply-add (fmac_self*): of D~ only the columns
      // right of the pivot still matter, in blocks of four (entries at or left of the pivot inside a block become garbage
      // that nothing reads again)
      if constexpr (B == 12) {
        if (k < 3) fmac_self4<k>(Dr, nmp);
        if (k < 7) fmac_self4<k>(Dr + 4, nmp);
        if (k < 11) fmac_self4<k>(Dr + 8, nmp);
      } else {
        fmac_self_n<k, B>(Dr, nmp);
      }
      fmac_self_n<k, B>(Or, nmp);
      fmac_self_n<k, B>(Fr, nmp);
      fmac_self1<k>(gr, nmp);
    });
    // a pivot that is not positive (or not a number) leaves a reciprocal that is not positive in its own lane: one test per
    // block step instead of one per pivot (lanes without a row keep 1)
    if (!(invs > 0.0) && live) *a.flag = 1;
    __builtin_amdgcn_sched_barrier(0);
    double Ol[B];
#pragma unroll
    for (int k = 0; k < B; k++) Ol[k] = L[ro + B * B + k];          // row r of O_j, from the image of record j
#pragma unroll
    for (int k = 0; k < B; k++) { Or[k] *= invs; Fr[k] *= invs; }   // U_j, V_j: row r
    gr *= invs;                                                      // Y_j
    wave_lds_sync();
    if (rowlane) {
#pragma unroll
      for (int k = 0; k < B; k++) {
        L[co + k * B] = Fr[k];               // the stored record is column-major [V | U | Y]
        L[co + B * B + k * B] = Or[k];
      }
      L[co + 2 * B * B] = gr;
    }
    wave_lds_sync();
    if (live) {
      V2 *dst = reinterpret_cast<V2 *>(a.blk + (size_t)j * BS);
#pragma unroll
      for (int q = 0; q < NV; q++) {
        const int idx = q * 16 + r;
        if (idx < NPC) dst[idx] = *reinterpret_cast<const V2 *>(&L[po + 32 * q]);
      }
    }
    wave_lds_sync();
    put_rec(pf, nextlive);                   // zeros after the chunk's last block: the products then are the addends
    wave_lds_sync();
    add_lambda(nextlive);
    double Dn[B], Fn[B], Gn[B], gn;
#pragma unroll
    for (int k = 0; k < B; k++) { Dn[k] = L[ro + k]; Gn[k] = 0.0; }
    gn = L[co + 2 * B * B];
    __builtin_amdgcn_sched_barrier(0);
    // two passes, so that at most seven 12-vectors are live: rows of U_j first (O_j^T is dead afterwards) ...
    static_for<0, B>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      const double ol = Ol[i], gg = Gr[i];          // subtracted: the negation is the instructions' source modifier
      fmac_bcast_n<i, B, true>(Dn, Or, ol);     // D~_{j+1} -= O_j[r][i] * (row i of U_j)
      fmac_bcast_n<i, B, true>(Gn, Or, gg);     // G_{j+1}  -= G_j[r][i] * (row i of U_j)
      fmac_bcast2<i, true>(gn, as_, gr, ol, gg);
    });
    // (pinned here: the compiler otherwise sinks the G_{j+1} sums below the output branch at the end of the step and
    //  spills all 144 broadcast values to feed them there)
#pragma unroll
    for (int k = 0; k < B; k++) asm volatile("" : "+v"(Gn[k]), "+v"(Dn[k]));
    __builtin_amdgcn_sched_barrier(0);
    // ... then rows of V_j
#pragma unroll
    for (int k = 0; k < B; k++) Fn[k] = 0.0;
    static_for<0, B>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      const double ol = Ol[i], gg = Gr[i];          // subtracted: the negation is the instructions' source modifier
      fmac_bcast_n<i, B, true>(Fn, Fr, ol);     // F_{j+1} -= O_j[r][i] * (row i of V_j)
      fmac_bcast_n<i, B, true>(Ar, Fr, gg);     // D_sep   -= G_j[r][i] * (row i of V_j)
    });
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < B; k++) asm volatile("" : "+v"(Fn[k]), "+v"(Ar[k]));
#pragma unroll
    for (int k = 0; k < B; k++) {
      Dr[k] = Dn[k]; Fr[k] = Fn[k]; Gr[k] = Gn[k];
      Or[k] = L[co + B * B + k * B];
    }
    gr = gn;
    __builtin_amdgcn_sched_barrier(0);
    if (live && lastb && rowlane) {          // after the chunk's last block the "next block" operands are the addends
      double *ub = a.up_blk + (size_t)c * BS;
#pragma unroll
      for (int k = 0; k < B; k++) {
        ub[r * B + k] = Ar[k];                                 // D_sep - sum G V
        ub[B * B + r * B + k] = right_exists ? Fr[k] : 0.0;   // C = -O V couples separator c to separator c + 1
      }
      ub[2 * B * B + r] = as_;
      if (right_exists) {
        double *ua = a.up_add + (size_t)(c + 1) * AS;
        const double *ra = (e == a.n) ? a.remote_add : nullptr;   // what is already owed to the outside separator
#pragma unroll
        for (int k = 0; k < B; k++) ua[r * B + k] = (ra ? ra[r * B + k] : 0.0) + Dr[k];   // -O U
        ua[B * B + r] = (ra ? ra[B * B + r] : 0.0) + gr;                                    // -O Y
      }
    }
    wave_lds_sync();
  }
}

// ---------------------------------------------------------------- assembly fused into the level-0 elimination
//
// k_fused_level0: a workgroup is TWO waves that share four chunks.  Wave 1 (ASM) forms the records [D_j | O_j | g_j]
// of the chunks' states straight from the row tables -- what k_assemble_ghost does, but chunk by chunk, state after
// state, in the row layout of the elimination: lane r of a 16-lane row accumulates row r of D and O; per Jacobian row
// it loads its own L and R element, gathers the 12 L (then the 12 R) elements of the row with
// `v_mov_b64_dpp row_newbcast:0..11` and adds L[r] L[k] -> D_j, R[r] L[k] -> O_j, R[r] R[k] -> the carry that opens
// D_{j+1}.  Wave 0 (ELIM) is k_chunk_forward_rows with the record images coming from ASM through LDS (two rotating
// images per chunk, one s_barrier per block step) instead of from memory.  The block records [D | O | g] never exist in
// HBM: -240 MB written, -262 MB read per iteration, and k_assemble_ghost's launch is gone.
// Conventions that differ from the unfused pair: a separator's own record holds only its rows' L^T L part -- the
// R^T R of the rows of the state before it (the previous chunk's last state) reaches it as part of the addend that
// chunk sends to the upper level anyway (the "virtual" record after a chunk's last state is [carry | 0 | carry_g]).
// TR: type of the Jacobian row tables (float on fp32 handles -- GPSLAM_FP32 --, whose accumulation, normal equations and
// elimination stay fp64: the rows are widened as they leave the ring)
template <typename T, typename TR = T> struct FusedArgs {
  FwdArgs<T> f;           // level-0 arguments of the elimination (blk: where the factors [V | U | Y] go)
  const int *rowptr;      // full-width rows of left state s: [rowptr[s], rowptr[s+1])
  const TR *rowLR, *rowE; // M x 24, M
  const int *crowptr;     // compact rows
  const TR *rowC, *rowCE; // Mc x 12, Mc
  const T *rowI = nullptr;       // k_fused_level0<4>: the interpolated measurement rows as 16-double lines (kIRow*), indexed by irowptr
  const int *irowptr = nullptr;  // n + 2 entries, like rowptr, counting only those rows (the GP priors have none: they are records)
  const T *gps;           // structured GP-prior records (kGpsLen each, see GpArgs::gps) or null: the GP rows are in rowLR
  int gp_count;           // number of records; record gp_count is all zeros (states without a GP prior read it)
  const T *brec;          // BetweenFactor<Pose3> records (kBtw*; K1 wrote no compact rows for them) or null (ST variants only)
  const int *btwidx;      // n + 2 entries: record of the between factor whose left state is s, or -1
  int btw_count;          // number of records; record btw_count is all zeros
  const int *gpidx;       // n + 2 entries: record of the GP prior whose left state is s, or -1
  int odd_rows;           // the structured chain has other full-width rows as well: 1 = a few (k_fused_level0<2>), 2 = many (<3>: a ring)
  int u_diag;             // Ud is diagonal (SE(3) records: k_fused_level0<1, double, 12, true>)
  const T *Ud;            // chol_upper(Qc^-1), row-major 6 x 6, in device memory: the structured velocity columns are multiples of its rows
  int *simd_cnt = nullptr;       // round 6 (GPS_ROLE_SWAP): wave-0 count per SIMD of the chip (8192 ints, zero between launches), or null
  T *gsave, *gsave2;      // Levenberg-Marquardt: the gradient g = -J^T e per state (gsave) and, for a chunk's separator, the part of
                          // it that the PREVIOUS chunk's last rows contribute (gsave2; zero elsewhere); null: not wanted
#ifdef GPS_TRACE_FUSED
  unsigned long long *trace = nullptr;   // debug builds only: 64 stamps per wave (scripts/trace_fused.py)
#endif
};
#ifdef GPS_TRACE_FUSED
#define GPS_TR(slot) do { if (u.trace && lane == 0) u.trace[((size_t)blockIdx.x * 2 + role) * 64 + (slot)] = wall_clock64(); } while (0)
#define GPS_TRA(slot) do { if (kimg == 13) GPS_TR(slot); } while (0)
#define GPS_TRE(slot) do { if (t == 10) GPS_TR(slot); } while (0)
#else
#define GPS_TR(slot) do { } while (0)
#define GPS_TRA(slot) do { } while (0)
#define GPS_TRE(slot) do { } while (0)
#endif


// ST: every full-width row of the chain belongs to a GP prior and K1 delivers those as structured records (u.gps): the
// full-width row ring is replaced by the record ring (a kernel with both spills: 256 VGPRs + 176 B of scratch, 0.34 ms).
// SV = 2: a structured chain that also has a few other full-width rows (a velocity prior or two): those are fetched where
// they are used, without a ring (the pure variant stays free of that loop's registers: with it the kernel spills again).
// (round 3: the block size is a template parameter -- 12: SE(3), with or without structured GP records; 6: SE(2), SO(3), 3-D linear
// chains, plain rows only)
// DG (round 4; SV = 1, B = 12): U = chol_upper(Qc^-1) is DIAGONAL (an isotropic or diagonal Qc, the common case; the host looks at
// the 36 numbers).  Then row q of the whitened L has, among its six velocity columns, only the one of its own component
// (L's velocity block is [k2 U; -sc U]): D and O need 7 instead of 12 multiply-adds per Jacobian row, and U Z is six products per
// six-vector instead of 21 multiply-adds -- 204 of the assembly wave's ~1170 instructions per block step.  What is skipped are
// products with exact zeros: the same values as the general kernel.
#ifndef GPS_FUSED_WAVES
#define GPS_FUSED_WAVES 2
#endif
// Wave priority inside k_fused_level0 (round 6).  The two waves a SIMD holds belong to two different workgroups, and the
// instruction arbiter serves the OLDER wave first: in a launch of one resident set (1e5 states: 1000 workgroups) the workgroups whose
// waves were placed first run at the speed of a wave that has its SIMD to itself and end at 115 us, those placed second take what is
// left and end at 145-154 us -- the launch lasts as long as the slowest (scripts/trace_fused.py, HW_ID wave slots).  GPS_PRIO = 1
// (the default): every block step a wave sets its priority to 3 - (step mod 4), so the wave that is BEHIND on its SIMD is the one
// that is served first and the workgroups of a CU advance together: the spread of a CU's workgroups falls from ~30 us to 4 us, the
// launch from 154 to 142-145 us at the same median (132 us: the work is what it was), 1e6 states unchanged (workgroups come and go
// there).  Measured and not kept: 2 / 3 (one role always first: no gain), 4 (two-step quantum: half the gain), 5 (half-step
// quantum: no better than 1).  0: no s_setprio (rounds 1-5).  HISTORY.md "Round 6" has the tables.
#ifndef GPS_PRIO
#define GPS_PRIO 1
#endif
// c: progress in half steps (2 t at the top of block step t, 2 t + 1 in its middle)
// Which wave of a workgroup eliminates (round 6, GPS_ROLE_SWAP).  The hardware puts wave 0 of a two-wave workgroup on SIMD a and wave 1
// on sigma(a), sigma the 4-cycle 3 -> 0 -> 2 -> 1 -> 3 (HW_ID of 1000 workgroups, scripts/trace_fused.py); a CU's four workgroups then
// either start on four different SIMDs -- every SIMD holds one wave 0 and one wave 1 -- or two by two on the same pair, and with
// "wave 0 eliminates" two SIMDs of that CU hold two elimination waves (5.4 us per block step instead of 4.8) and two hold two assembly
// waves: 27-76 of the chip's 1024 SIMDs per launch, and their workgroups are the launch's last.  With the switch on, wave 0 counts
// itself into a per-SIMD word (one returning atomic per workgroup, undone on exit): the SECOND wave 0 of a SIMD trades roles with its
// wave 1, which puts an elimination wave on sigma(a) -- where the first workgroup's assembly wave sits -- and every SIMD of every CU
// holds one wave of each kind.  (Exact for a launch of one resident set, where it matters: 1e5 states.  Where workgroups come and go the
// word counts wave 0s, not elimination waves -- a newcomer next to a workgroup that traded sees the count and trades as well -- and
// the launch's duration does not depend on the placement: 1e6 states 1.27 ms with or without.)
#ifndef GPS_ROLE_SWAP
#define GPS_ROLE_SWAP 1
#endif
// (block size 12 only: the three-waves-per-SIMD kernel of the d = 3 chains gains nothing at 1e5 states and loses 2 % at 1e6)
template <int B> __device__ __forceinline__ void fused_step_prio(int c, int role) {
  if constexpr (B != 12) return;
#if GPS_PRIO == 5
  const int t = c;
#else
  if (c & 1) return;
  const int t = c >> 1;
#endif
#if GPS_PRIO == 1 || GPS_PRIO == 4 || GPS_PRIO == 5
  const int q = (GPS_PRIO == 4 ? (t >> 1) : t) & 3;
  if (q == 0) __builtin_amdgcn_s_setprio(3);
  else if (q == 1) __builtin_amdgcn_s_setprio(2);
  else if (q == 2) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#elif GPS_PRIO == 2
  if (t == 0) { if (role == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
#elif GPS_PRIO == 3
  if (t == 0) { if (role == 0) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3); }
#else
  (void)t; (void)role;
#endif
}
template <int SV, typename TR = double, int B = 12, bool DG = false>
__global__ void __launch_bounds__(128, GPS_FUSED_WAVES) k_fused_level0(FusedArgs<double, TR> u) {
  static_assert(SV == 0 || std::is_same<TR, double>::value, "structured GP records are fp64");
  static_assert(!DG || ((SV == 1 || SV == 3 || SV == 4) && B == 12), "the diagonal-U form belongs to the SE(3) record variants");
  // SV = 3 (round 4): records AND a ring of full-width rows -- SE(3) chains with interpolated measurement factors (GPS, range,
  // projection: a dozen full-width rows per state), which had lost the records to the plain-row kernel
  // SV = 4 (round 5): records AND interpolated measurement rows as 16-double lines (kIRow*): one operand per row and lane, a whole
  // state's dozen rows in flight, the right halves formed here from mu and the record's X / F X columns
  constexpr bool ST = SV != 0, ODD = SV >= 2, ORING = SV == 3, IROW = SV == 4;
  static_assert(!IROW || B == 12, "interpolated rows are an SE(3) form");
  // ST12: SE(3) records (kGps*, kBtw*: the assembly wave forms the columns, no full-width row ring);  ST6 (round 4): the d = 3
  // records (kGp3*) of SE(2) / SO(3) / 3-D linear chains -- six rows per state from 11 operands per lane, next to the row ring
  // that still serves measurement factors and velocity priors
  constexpr bool ST12 = ST && B == 12, ST6 = ST && B == 6;
  const FwdArgs<double> &a = u.f;
  static_assert(B == 12 || B == 6, "block sizes with DPP gather blocks");
  static_assert(SV == 0 || B == 12 || SV == 1, "the variants with odd rows are SE(3) variants");
  constexpr int BS = 2 * B * B + B, AS = B * B + B;   // R == 1
  constexpr int NPC = BS / 2, NV = (NPC + 15) / 16;
  typedef double V2 __attribute__((ext_vector_type(2)));
  // (alternating which wave assembles and which eliminates between workgroups -- by bit 0, 1 or 2 of the workgroup index -- so that a
  //  SIMD gets one wave of each kind changes nothing: 0.303-0.307 ms per iteration at 1e5 states for all four assignments)
  const int lane = threadIdx.x & 63, grp = lane >> 4, r = lane & 15;
  int role = threadIdx.x >> 6;
#if GPS_ROLE_SWAP
  // (the variants that sit at 256 VGPRs -- a ring of full-width rows next to the records, the general-Qc line form -- keep "wave 0
  //  eliminates": the role bookkeeping costs them 16-20 bytes of scratch, and a scratch reload drains the loads in flight)
  // (block size 6 runs three waves per SIMD, six workgroups per CU: the two-by-two argument above is about two)
  constexpr bool kSwap = B == 12 && !(SV == 3 || (SV == 4 && !DG));
  __shared__ int swap_s;
  int simd_slot = -1;                             // (wave 0: the word it counted itself into)
  if (kSwap && u.simd_cnt != nullptr) {
    if (role == 0) {
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;   // HW_ID, XCC_ID
      simd_slot = (int)((((xc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf)) * 4 + ((hw >> 4) & 3));
      if (lane == 0) swap_s = atomicAdd(u.simd_cnt + simd_slot, 1) > 0 ? 1 : 0;
    }
    __syncthreads();
    role ^= __builtin_amdgcn_readfirstlane(swap_s);
  }
  auto leave = [&]() { if (simd_slot >= 0 && lane == 0) atomicSub(u.simd_cnt + simd_slot, 1); };
#else
  auto leave = [&]() {};
#endif
  const int c = blockIdx.x * 4 + grp;
  const int nch = (a.n + a.m - 1) / a.m;
  const bool valid = c < nch;
  const int s = valid ? c * a.m : 0;
  const int e = valid ? min(s + a.m, a.n) : 0;
  const int j0 = s + 1;
  const bool rowlane = r < B;
  const int rr = rowlane ? r : 0;
  const bool right_exists = (e < a.n) || (a.last_has_right != 0);
  // tail: the chain's last chunk is padded to full length with decoupled identity blocks (D = I, O = 0, g = 0: their
  // elimination changes nothing and their records are never stored), so that the four chunks of a wave end in the same
  // block step and their separator data is still in registers when the tail starts
  const bool tail = a.tail != 0;
  const int ep = (valid && tail) ? s + a.m : e;
  const bool has_int = valid && j0 < ep;
  const double lambda = a.lambda;
  // The images [D | O | g] the assembly wave hands over are private to this kernel: their rows are padded to BP = B + 1 doubles.
  // With a pitch of B the B lanes of a row access (stride 2 B dwords) collide pairwise in the 64 LDS banks (B = 12: lanes r and
  // r + 8; 32 banks for stores: three ways) -- SQ_LDS_BANK_CONFLICT was 49 % of SQ_LDS_IDX_ACTIVE (round 4); with an odd pitch
  // row accesses and column accesses are both conflict-free inside a 16-lane row.  OUTR keeps the record's own (unpadded,
  // column-major) layout: it leaves as contiguous 16-byte pieces.
  constexpr int BP = B + 1, IS = 2 * B * BP + B + (B & 1);            // image pitch, doubles per image (even)
  __shared__ __attribute__((aligned(16))) double IMG[2 * 4 * IS > 5 * BS + 4 * AS ? 2 * 4 * IS : 5 * BS + 4 * AS];   // two rotating images per chunk (the tail's group reuses the space)
  __shared__ __attribute__((aligned(16))) double OUTR[4 * BS];       // the factor record on its way out
  int ro = grp * IS + rr * BP;    // row r of an image:    IMG[buf * 4 IS + ro + k]          (D: + 0, O: + B BP, g: co + 2 B BP)
  int co = grp * IS + rr;         // column r:             IMG[... + co + k * BP]
  int po = grp * BS + 2 * r;      // 16-byte piece q * 16 + r of OUTR
  int oc = grp * BS + rr;         // column r of the factor record in OUTR: OUTR[oc + k * B]
  int tr = grp * BS + rr * BP;    // row r of the (padded) transpose scratch inside OUTR's V | U area
  int tc = grp * BS + rr;         // ... its column r: OUTR[tc + k * BP]
  asm volatile("" : "+v"(ro), "+v"(co), "+v"(po), "+v"(oc), "+v"(tr), "+v"(tc));
  const int steps = __builtin_amdgcn_readfirstlane(max(ep - j0, 0));   // lane 0: the wave's first (never shorter) chunk
#ifdef GPS_PROBE_ONLY_ROLE
  if (role != GPS_PROBE_ONLY_ROLE) return;      // (register probes: one role's code alone; never a product build)
#endif
  GPS_TR(0);
#ifdef GPS_TRACE_FUSED
  if (u.trace && lane == 0) {
    u.trace[((size_t)blockIdx.x * 2 + role) * 64 + 61] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
    u.trace[((size_t)blockIdx.x * 2 + role) * 64 + 62] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
  }
#endif

  if (role == 1) {
    // ================================================================ ASM: images 0 .. steps + 1
    // Everything that has a memory latency is requested one state ahead: the row pointers of state t + 2 and the first
    // PF rows (full-width and compact) of state t + 1 are in flight while state t is accumulated.
    // ring depths (rows in flight per table).  Whole-state rings (12 full-width rows: one GP prior) were measured SLOWER
    // (0.197 vs 0.182 ms): with all arithmetic of both waves ablated the kernel still takes 0.158 ms -- 313 MB of row reads
    // + 248 MB of factor writes at the mixed read / write rate this part sustains -- so deeper prefetch only costs registers.
    constexpr int PF = ST6 ? 3 : (ORING ? (DG ? 6 : 4) : 6), PC = ST12 ? (SV == 2 ? 4 : ((ORING || IROW) ? 2 : 1)) : (ST6 ? 3 : 6), Dh = B / 2;
    constexpr int PFI = 12;                               // interpolated rows in flight: one register each (four GPS factors of an interval)   // (ST6: three waves per SIMD need <= 168 VGPRs)   // (ST: the records take the registers of the compact ring: pose priors are what is left in it)
    const int rc = r < Dh ? r : 0;
    const int ptr_max = a.n + 1;                         // rowptr / crowptr have n + 2 entries
    // R^T R of the state's rows (and its share of the gradient): what opens the NEXT state's D -- kept where it is summed, the next
    // state takes it over when it starts (round 4: a copy out at the end of a state and a copy in at the start of the next were
    // 24 moves per state); dgpend: what write_img adds to the diagonal of the image (LM damping, identity of a padding block)
    double RRacc[B], grr = 0.0, dgpend = 0.0;
#pragma unroll
    for (int k = 0; k < B; k++) RRacc[k] = 0.0;
    double Dacc[B], Oacc[B], gacc;
    constexpr bool FRING = !ST12 || ORING;            // a ring of full-width rows (ORING: 4 deep -- 6 in the diagonal-Qc form -- + 2 compact rows is what 256 VGPRs hold next to the records)
    double fL[FRING ? PF : 1], fR[FRING ? PF : 1], fE[FRING ? PF : 1], cL[PC], cR[PC], cE[PC];   // the two operand rings
    double fI[IROW ? PFI : 1];                           // ... and the ring of interpolated rows: element (lane & 15) of each line
    // IROW (round 6): the compact rows (a between factor's six, a pose prior's) as ONE register per row as well -- lane c < 12 holds
    // element c of [L (6) | R (6)], lanes 12..15 the whitened error -- so that a whole state's rows are in flight where the three
    // registers per row of cL / cR / cE allowed two: their latency was exposed three times per block step (2.2 of its 7.1 us)
    constexpr int PCI = 6;
    double cV[IROW ? PCI : 1];
    const int *fptr = IROW ? u.irowptr : u.rowptr;       // the full-width table this variant walks
    int rp = 0, nf = 0, cp = 0, nc = 0;                  // rows of the state the rings belong to
    int rpn = fptr[min(s + 1, ptr_max)], cpn = u.crowptr[min(s + 1, ptr_max)];      // pointers one state ahead
    int rpnn = fptr[min(s + 2, ptr_max)], cpnn = u.crowptr[min(s + 2, ptr_max)];    // ... and two
    // structured GP prior of the state (u.gps, layout kGps*): lane c < 12 of the chunk's DPP row builds COLUMN c of the whitened
    // [L | R] from the record's blocks -- pose columns for c < 6, velocity columns for c >= 6 (see the record's description):
    //   column r6 = c mod 6 of X = Jinv and of J as six-vectors [top; bottom] (the translation columns have a zero top and
    //   the diagonal block below it), F times both (F's blocks live one entry per lane: element 3 i + j in lane 3 i + j),
    //   then  L = [U (aL J_c + b F J_c); U (c F J_c - dR J_c)],  R = [U (aR X_c + b F X_c); U (c F X_c + dR X_c)]
    // with per-lane coefficients fetched from the record (pose lanes: aL = aR = sa, b = sb, c = sc, dR = 0; velocity lanes:
    // J_c = the unit vector, aL = k2, aR = sb, b = c = 0, dR = sc -- their zeros are the record's zero slot).
    constexpr bool st_on = ST;
    constexpr int NRAW = ST12 ? 21 : (ST6 ? 11 : 1);
    double Ur[(ST12 && !DG) ? 3 : 1];
    if constexpr (ST12 && !DG) {
#pragma unroll
      for (int k = 0; k < 3; k++) Ur[k] = u.Ud[min(16 * k + r, 35)];       // U, row-major: entry e in lane e & 15 of Ur[e >> 4]
    }
    double udv = 0.0;                                                        // DG: the diagonal, entry k in lane k of every row
    if constexpr (DG) udv = u.Ud[7 * min(r, 5)];
    // where this lane's operands sit in a record (loop-invariant; the stride-3 walks down a column are immediate offsets)
    int oX1 = 0, oX2 = 0, oJ1 = 0, oJ2 = 0, oF = 0, oE = 0, oaL = 0, oaR = 0, ob = 0, oc = 0, od = 0;
    if constexpr (ST12) {
      const int r6 = r < Dh ? r : (r < B ? r - Dh : 0);
      const bool velc = r >= Dh && r < B, hi3 = r6 >= 3;
      const int j3 = hi3 ? r6 - 3 : r6;
      oX1 = kGpsXA + j3; oX2 = (hi3 ? kGpsXA : kGpsXC) + j3;      // top three of X's column (masked for the translation columns), bottom three
      oJ1 = kGpsJA + j3; oJ2 = (hi3 ? kGpsJA : kGpsJC) + j3;
      oF = min(r, 8); oE = kGpsE + min(r, 11);
      oaL = velc ? kGpsS + 0 : kGpsS + 3;    // aL: k2 | sa
      oaR = velc ? kGpsS + 1 : kGpsS + 3;    // aR: sb | sa
      ob = velc ? kGpsZ : kGpsS + 1;         // b:  0  | sb
      oc = velc ? kGpsZ : kGpsS + 2;         // c:  0  | sc
      od = velc ? kGpsS + 2 : kGpsZ;         // dR: sc | 0
    }
    double raw[NRAW];
    int gp = -1;
    // BetweenFactor<Pose3> record of the state (u.brec, kBtw*): lane c < 6 builds column c of [H1 | H2] from the block-triangular
    // halves exactly as above (the same column walk: oX1 / oX2 minus their bases), lanes 6..11 read the all-zero record
    constexpr int NBR = ST12 ? 14 : 1;
    double braw[NBR];
    // (not in the variant with odd full-width rows: its registers are spoken for -- the host hands it compact rows)
    const bool btw_on = ST12 && !ODD && u.brec != nullptr;
    int bq = -1, bqn = btw_on ? u.btwidx[min(s + 1, ptr_max)] : -1, bqnn = btw_on ? u.btwidx[min(s + 2, ptr_max)] : -1;
#pragma unroll
    for (int k = 0; k < NBR; k++) braw[k] = 0.0;
    auto ldbtw = [&]() {                      // operands of the between factor whose left state the rings point at (bq)
      if constexpr (ST12) {
        if (!btw_on) return;                  // (no records on this launch: u.brec is null)
        const double *rec = u.brec + (size_t)((bq >= 0 && r < Dh) ? bq : u.btw_count) * kBtwLen;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          braw[k] = rec[kBtwRA - kGpsXA + oX1 + 3 * k]; braw[3 + k] = rec[kBtwRA - kGpsXA + oX2 + 3 * k];
          braw[6 + k] = rec[kBtwLA - kGpsXA + oX1 + 3 * k]; braw[9 + k] = rec[kBtwLA - kGpsXA + oX2 + 3 * k];
        }
        braw[12] = rec[kBtwW + min(r, Dh - 1)];
        braw[13] = rec[kBtwE + min(r, Dh - 1)];
      }
    };
    int gpn = st_on ? u.gpidx[min(s + 1, ptr_max)] : -1, gpnn = st_on ? u.gpidx[min(s + 2, ptr_max)] : -1;
    // operands of the GP prior whose left state is s + kimg (record g; no such factor: the all-zero record behind the last one)
    auto ldraw = [&](int kimg, int g) {
      if constexpr (ST6) {
        // d = 3 record (kGp3*): lane c < 6 holds column c of the six rows [A1 | kLt U | A3 | kRt U; 0 | kLb U | 0 | kRb U] -- a pose
        // lane (c < 3) its column of A1 / A3, a velocity lane column c - 3 of U and the record's four coefficients; lane q < 6 also
        // fetches whitened error q (it travels by row_newbcast)
        const bool live = valid && (s + kimg) < e && g >= 0;
        const double *rec = u.gps + (size_t)(live ? g : u.gp_count) * kGp3Len;
        const bool pc = r < 3;
        const int c3 = pc ? r : (r < 6 ? r - 3 : 0);
#pragma unroll
        for (int k = 0; k < 3; k++) {
          raw[k] = pc ? rec[kGp3A1 + 3 * k + c3] : u.Ud[3 * k + c3];
          raw[3 + k] = pc ? rec[kGp3A3 + 3 * k + c3] : u.Ud[3 * k + c3];
        }
        // (the zero record makes the velocity lanes' coefficients zero: a state without a GP prior contributes nothing)
        raw[6] = rec[kGp3S + 0]; raw[7] = rec[kGp3S + 1]; raw[8] = rec[kGp3S + 2]; raw[9] = rec[kGp3S + 3];
        raw[10] = rec[kGp3E + min(r, 5)];
      }
      if constexpr (ST12) {
        const bool live = valid && (s + kimg) < e && g >= 0;
        const double *rec = u.gps + (size_t)(live ? g : u.gp_count) * kGpsLen;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          raw[k] = rec[oX1 + 3 * k]; raw[3 + k] = rec[oX2 + 3 * k];
          raw[6 + k] = rec[oJ1 + 3 * k]; raw[9 + k] = rec[oJ2 + 3 * k];
        }
        raw[12] = rec[kGpsFA + oF]; raw[13] = rec[kGpsFC + oF]; raw[14] = rec[kGpsFD + oF];
        raw[15] = rec[oE];
        raw[16] = rec[oaL]; raw[17] = rec[oaR]; raw[18] = rec[ob]; raw[19] = rec[oc]; raw[20] = rec[od];
      }
    };
    double Lcol[ST12 ? B : 1], Rcol[ST12 ? B : 1], newl = 0.0;
    auto reconstruct = [&]() {
      if constexpr (ST12) {
        double X6[6], J6[6], P3[6], P1[6];
        // the lane's role, recomputed per state from an opaque copy of its index: as loop invariants the masks and the unit
        // vector would occupy two dozen registers for the whole kernel (the wave spilled with them)
        int rq = r;
        asm volatile("" : "+v"(rq));
        const bool tcol = (rq >= 3 && rq < 6) || (rq >= 9), vcol = rq >= 6;   // translation column; velocity column
#pragma unroll
        for (int k = 0; k < 3; k++) {
          X6[k] = tcol ? 0.0 : raw[k]; X6[3 + k] = raw[3 + k];
          J6[k] = vcol ? ((rq == 6 + k) ? 1.0 : 0.0) : (tcol ? 0.0 : raw[6 + k]);
          J6[3 + k] = vcol ? ((rq == 9 + k) ? 1.0 : 0.0) : raw[9 + k];
        }
#pragma unroll
        for (int k = 0; k < 6; k++) { P3[k] = 0.0; P1[k] = 0.0; }
        const double *fa = &raw[12], *fc = &raw[13], *fd = &raw[14];
        static_for<0, 3>([&](auto jj) {            // F [top; bottom] = [FA top; FC top + FD bottom]
          constexpr int j = decltype(jj)::value;
          fmac_mat<3, j, 3>(P3, fa, X6[j]);
          fmac_mat<3, j, 3>(P3 + 3, fc, X6[j]);
          fmac_mat<3, j, 3>(P3 + 3, fd, X6[3 + j]);
          fmac_mat<3, j, 3>(P1, fa, J6[j]);
          fmac_mat<3, j, 3>(P1 + 3, fc, J6[j]);
          fmac_mat<3, j, 3>(P1 + 3, fd, J6[3 + j]);
        });
        const double aL = raw[16], aR = raw[17], bb = raw[18], cc = raw[19], dR = raw[20], ndR = -dR;
        double Z[4][6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
          Z[0][k] = fma(aL, J6[k], bb * P1[k]);      // top rows of L
          Z[1][k] = fma(cc, P1[k], ndR * J6[k]);     // bottom rows of L
          Z[2][k] = fma(aR, X6[k], bb * P3[k]);      // top rows of R
          Z[3][k] = fma(cc, P3[k], dR * X6[k]);      // bottom rows of R
        }
        if constexpr (DG) {
          static_for<0, 6>([&](auto kk) {            // U Z, U diagonal
            constexpr int k = decltype(kk)::value;
            const double uk = row_bcast<k>(udv);
            Lcol[k] = uk * Z[0][k]; Lcol[6 + k] = uk * Z[1][k];
            Rcol[k] = uk * Z[2][k]; Rcol[6 + k] = uk * Z[3][k];
            // (formed here: sunk into the row loop they would keep Z and the record's operands alive across it -- the wave spilled)
            asm volatile("" : "+v"(Lcol[k]), "+v"(Lcol[6 + k]), "+v"(Rcol[k]), "+v"(Rcol[6 + k]));
          });
        } else {
#pragma unroll
          for (int k = 0; k < B; k++) { Lcol[k] = 0.0; Rcol[k] = 0.0; }
          static_for<0, 6>([&](auto kk) {            // U Z, U upper triangular: out[i] += U[i][k] Z[k], i <= k
            constexpr int k = decltype(kk)::value;
            fmac_mat<k + 1, k, 6>(Lcol, Ur, Z[0][k]);
            fmac_mat<k + 1, k, 6>(Lcol + 6, Ur, Z[1][k]);
            fmac_mat<k + 1, k, 6>(Rcol, Ur, Z[2][k]);
            fmac_mat<k + 1, k, 6>(Rcol + 6, Ur, Z[3][k]);
          });
        }
        newl = -raw[15];                             // minus the whitened error of row (lane)
      }
    };
    auto ldf = [&](int i, double &Lv, double &Rv, double &ev) {
      const int rho = rp + min(i, max(nf - 1, 0));
      const TR *row = u.rowLR + (size_t)rho * 2 * B;
      Lv = (double)row[rr]; Rv = (double)row[B + rr]; ev = (double)u.rowE[rho];
    };
    auto ldfi = [&](int i, double &v) {      // element (lane & 15) of interpolated row i of the state the rings point at
      const int rho = rp + min(i, max(nf - 1, 0));
      v = u.rowI[(size_t)rho * kIRowLen + r];
    };
    // point the ring of interpolated rows at state s + kimg and request its first PFI lines.  assemble() does this for the NEXT
    // state as soon as the current state's lines are consumed (they then travel under the GP prior's, the between factor's and the
    // pose prior's rows: ~800 instructions); open_state() only for the very first state.
    bool iopened = false;
    auto open_irows = [&](int kimg, int p0, int p1) {
      const bool live = valid && (s + kimg) < e;
      rp = live ? p0 : 0; nf = live ? p1 - p0 : 0;
#pragma unroll
      for (int q = 0; q < PFI; q++) ldfi(q, fI[q]);
      iopened = true;
    };
    auto ldc = [&](int i, double &Lv, double &Rv, double &ev) {
      const int rho = cp + min(i, max(nc - 1, 0));
      const TR *row = u.rowC + (size_t)rho * B;
      Lv = (double)row[rc]; Rv = (double)row[Dh + rc]; ev = (double)u.rowCE[rho];
    };
    auto ldcv = [&](int i, double &v) {      // element (lane & 15) of compact row i of the state the rings point at: [L | R | e e e e]
      const int rho = cp + min(i, max(nc - 1, 0));
      const TR *p = (r < B) ? u.rowC + (size_t)rho * B + r : u.rowCE + rho;
      v = (double)*p;
    };
    // point the rings at state s + kimg (row range known from the pointers loaded earlier) and start their first loads
    auto open_state = [&](int kimg, int p0, int p1, int q0, int q1, int g, int bqv) {
      const bool live = valid && (s + kimg) < e;
      gp = (live && st_on) ? g : -1;
      bq = (live && btw_on) ? bqv : -1;
      if (gp >= 0 && !IROW) p0 += B;                     // its 12 rows lead the state's range in the row table: not used (the table of interpolated rows never held them)
      if (bq >= 0) q1 -= Dh;                             // ... and its between factor's six rows end its range in the compact table
      if constexpr (!IROW) { rp = (live && (!ST12 || ODD)) ? p0 : 0; nf = (live && (!ST12 || ODD)) ? p1 - p0 : 0; }   // (ODD: the few other full-width rows)
      cp = live ? q0 : 0; nc = live ? q1 - q0 : 0;
      if constexpr (FRING) {
#pragma unroll
        for (int q = 0; q < PF; q++) ldf(q, fL[q], fR[q], fE[q]);
      }
      if constexpr (IROW) {
        if (!iopened) open_irows(kimg, p0, p1);          // (the first state only: every other one was opened early, by assemble)
        iopened = false;
      } else {       // (IROW: the compact ring of a state is requested behind that state's interpolated rows, whose loop needs the registers)
#pragma unroll
        for (int q = 0; q < PC; q++) ldc(q, cL[q], cR[q], cE[q]);
      }
    };
    auto assemble = [&](int kimg) {
      const bool live = valid && (s + kimg) < e;
      GPS_TRA(40);
      int nfm = nf, ncm = nc;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { nfm = max(nfm, __shfl_xor(nfm, o, 64)); ncm = max(ncm, __shfl_xor(ncm, o, 64)); }
      nfm = __builtin_amdgcn_readfirstlane(nfm);
      ncm = __builtin_amdgcn_readfirstlane(ncm);
#pragma unroll
      for (int k = 0; k < B; k++) { Dacc[k] = RRacc[k]; Oacc[k] = 0.0; RRacc[k] = 0.0; }
      gacc = grr;
      grr = 0.0;
      if constexpr (IROW) {
        // The interpolated measurement rows of the state FIRST, while the record's operands are still whole (the columns of the
        // GP prior are formed after this loop, into the registers it frees).  Row = [Lp | mu | e, p11, p12, l12], one element per
        // lane:  L = [Lp | l12 mu],  R = [mu (p11 X + p12 F X) | p12 mu X] = cA (mu . Xc) + cB (mu . Pc)  with this lane's column
        // Xc of X (velocity lanes: column c - 6) and Pc = F Xc, (cA, cB) = (p11, p12) on pose lanes, (p12, 0) on velocity lanes.
        int rq = r;
        asm volatile("" : "+v"(rq));
        const bool tcol = (rq >= 3 && rq < 6) || (rq >= 9), posel = rq < Dh, vell = rq >= Dh && rq < B;
        double Xc[6], Pc[6];
#pragma unroll
        for (int k = 0; k < 3; k++) { Xc[k] = tcol ? 0.0 : raw[k]; Xc[3 + k] = raw[3 + k]; }
#pragma unroll
        for (int k = 0; k < 6; k++) Pc[k] = 0.0;
        static_for<0, 3>([&](auto jj) {                  // F [top; bottom] = [FA top; FC top + FD bottom]
          constexpr int j = decltype(jj)::value;
          fmac_mat<3, j, 3>(Pc, &raw[12], Xc[j]);
          fmac_mat<3, j, 3>(Pc + 3, &raw[13], Xc[j]);
          fmac_mat<3, j, 3>(Pc + 3, &raw[14], Xc[3 + j]);
        });
        for (int i0 = 0; i0 < nfm; i0 += PFI) {
          if (i0 > 0) {                                  // more than PFI lines on one state (rare): the ring is refilled in place
#pragma unroll
            for (int q = 0; q < PFI; q++) ldfi(i0 + q, fI[q]);
          }
#pragma unroll
          for (int q = 0; q < PFI; q++) {
            const int i = i0 + q;
            const double V = (i < nf) ? fI[q] : 0.0;
            double sc[4];
            row_bcast4_at<kIRowE>(V, sc);                // whitened error, p11, p12, l12
            const double Lv = posel ? V : (vell ? sc[3] * V : 0.0);
            double sX = 0.0, sP = 0.0;
            fmac_dot6x2_at<kIRowMu>(sX, sP, V, Xc, Pc);  // mu . Xc, mu . Pc
            const double cA = posel ? sc[1] : sc[2], cB = posel ? sc[2] : 0.0;
            const double Rv = rowlane ? fma(cA, sX, cB * sP) : 0.0;
            fmac_gather<B>(Dacc, Lv, Lv);
            fmac_gather<B>(Oacc, Lv, Rv);
            fmac_gather<B>(RRacc, Rv, Rv);
            gacc = fma(-Lv, sc[0], gacc);
            grr = fma(-Rv, sc[0], grr);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        open_irows(kimg + 1, rpn, rpnn);                 // the next state's lines: in flight under the rest of this state
#pragma unroll
        for (int q = 0; q < PCI; q++) ldcv(q, cV[q]);    // this state's compact rows: wanted behind the GP prior's twelve
      }
      if constexpr (ST6) {                               // the d = 3 record: six rows from the lane's column of [A1 | U | A3 | U]
        const bool pc = r < 3;
        const double mLt = pc ? 1.0 : raw[6], mRt = pc ? 1.0 : raw[7], mLb = pc ? 0.0 : raw[8], mRb = pc ? 0.0 : raw[9];
        const double ne3 = -raw[10];
        static_for<0, 6>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          const double Lv = (i < 3 ? mLt : mLb) * raw[i % 3], Rv = (i < 3 ? mRt : mRb) * raw[3 + i % 3];
          fmac_gather<B>(Dacc, Lv, Lv);
          fmac_gather<B>(Oacc, Lv, Rv);
          fmac_gather<B>(RRacc, Rv, Rv);
          fmac_bcast2<i>(gacc, grr, ne3, Lv, Rv);
          __builtin_amdgcn_sched_barrier(0);
        });
        ldraw(kimg + 1, gpn);                            // (the operands are consumed: the next state's record into their registers)
      }
      if constexpr (ST12) {                              // the structured GP prior: its 12 rows from the columns built above
        GPS_TRA(41);
        reconstruct();
        GPS_TRA(42);
        static_for<0, B>([&](auto qq) {
          constexpr int q = decltype(qq)::value;
          // the next state's record is requested once most of this state's columns are consumed (their registers take it)
          if constexpr (q == 8) ldraw(kimg + 1, gpn);
          if constexpr (q == 5) ldbtw();                 // this state's between record: wanted right behind these twelve rows
          const double Lv = Lcol[q], Rv = Rcol[q];
          // (round 6: unguarded multiply-adds -- what they read through DPP, the row's entries Lcol[q] / Rcol[q] and the whitened
          //  error, was written by reconstruct() before this loop began; one guard opens the loop)
          if constexpr (q == 0) dpp_guard();
          if constexpr (DG) {
            // row q of L: the pose columns and velocity column 6 + q mod 6.  X, J and F are block lower triangular ([[A, 0], [C, D]]),
            // so the rotation rows (q mod 6 < 3) of the whitened [L | R] are zero in every translation column (3..5, and 9..11 of R)
            constexpr int vq = Dh + q % Dh;
            if constexpr (q % Dh < 3) {
              fmac_gather_nn<3>(Dacc, Lv, Lv); fmac_bcast1_nn<vq>(Dacc[vq], Lv, Lv);
              fmac_gather_nn<3>(Oacc, Lv, Rv); fmac_bcast1_nn<vq>(Oacc[vq], Lv, Rv);
              fmac_gather_nn<3>(RRacc, Rv, Rv); fmac_gather_nn<3, Dh>(RRacc + Dh, Rv, Rv);
            } else {
              fmac_gather_nn<Dh>(Dacc, Lv, Lv); fmac_bcast1_nn<vq>(Dacc[vq], Lv, Lv);
              fmac_gather_nn<Dh>(Oacc, Lv, Rv); fmac_bcast1_nn<vq>(Oacc[vq], Lv, Rv);
              fmac_gather_nn<B>(RRacc, Rv, Rv);
            }
          } else {
            fmac_gather_nn<B>(Dacc, Lv, Lv);
            fmac_gather_nn<B>(Oacc, Lv, Rv);
            fmac_gather_nn<B>(RRacc, Rv, Rv);
          }
          fmac_bcast1_nn<q>(gacc, newl, Lv);             // g -= e[q] L[q][r]
          fmac_bcast1_nn<q>(grr, newl, Rv);              // carry_g -= e[q] R[q][r]
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      GPS_TRA(43);
      if (kimg >= 3) fused_step_prio<B>(2 * (kimg - 3) + 1, 1);   // (image t + 3 is assembled under block step t: its middle)
      if constexpr (ST12) {                              // the state's BetweenFactor<Pose3> record: six compact rows from its columns
        if (btw_on) {
          int rq = r;
          asm volatile("" : "+v"(rq));
          const bool tcol = rq >= 3;                     // translation columns: zero top, the diagonal block below
          double Lc6[6], Rc6[6];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            Rc6[k] = tcol ? 0.0 : braw[k]; Rc6[3 + k] = braw[3 + k];
            Lc6[k] = tcol ? 0.0 : braw[6 + k]; Lc6[3 + k] = braw[9 + k];
          }
          const double nbe = -braw[13];
          static_for<0, Dh>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const double wi = row_bcast<i>(braw[12]);     // 1 / sigma of row i
            const double Lv = wi * Lc6[i], Rv = wi * Rc6[i];
            dpp_guard();                                 // (Lv, Rv are one instruction old; everything else in the row is not)
            if constexpr (i < 3) {                       // rotation rows of [[A, 0], [C, A]]: zero in the translation columns
              fmac_gather_nn<3>(Dacc, Lv, Lv);
              fmac_gather_nn<3>(Oacc, Lv, Rv);
              fmac_gather_nn<3>(RRacc, Rv, Rv);
            } else {
              fmac_gather_nn<Dh>(Dacc, Lv, Lv);
              fmac_gather_nn<Dh>(Oacc, Lv, Rv);
              fmac_gather_nn<Dh>(RRacc, Rv, Rv);
            }
            fmac_bcast1_nn<i>(gacc, nbe, Lv);
            fmac_bcast1_nn<i>(grr, nbe, Rv);
            __builtin_amdgcn_sched_barrier(0);
          });
        }
      }
      GPS_TRA(44);
      if constexpr (SV == 2) {
        // the odd full-width row of a structured chain (host-checked to be few): fetched where it is used, no ring --
        // only the block step of a state that has one waits for it
        for (int i = 0; i < nfm; i++) {
          double Lv, Rv, ev;
          ldf(i, Lv, Rv, ev);
          const bool ok = i < nf;
          Lv = ok ? Lv : 0.0; Rv = ok ? Rv : 0.0; ev = ok ? ev : 0.0;
          fmac_gather<B>(Dacc, Lv, Lv);
          fmac_gather<B>(Oacc, Lv, Rv);
          fmac_gather<B>(RRacc, Rv, Rv);
          gacc = fma(-Lv, ev, gacc);
          grr = fma(-Rv, ev, grr);
        }
      }
      if constexpr (FRING)
      for (int i0 = 0; i0 < nfm; i0 += PF) {             // full-width rows
#pragma unroll
        for (int q = 0; q < PF; q++) {
          const int i = i0 + q;
          const bool ok = i < nf;
          const double Lv = ok ? fL[q] : 0.0, Rv = ok ? fR[q] : 0.0, ev = ok ? fE[q] : 0.0;
          fmac_gather<B>(Dacc, Lv, Lv);     // D[r][k] += L[k] L[r]: the row's element of lane k fused into the multiply-add
          fmac_gather<B>(Oacc, Lv, Rv);     // O[r][k] += L[k] R[r]
          fmac_gather<B>(RRacc, Rv, Rv);    // carry[r][k] += R[k] R[r]
          gacc = fma(-Lv, ev, gacc);
          grr = fma(-Rv, ev, grr);
          ldf(i + PF, fL[q], fR[q], fE[q]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (IROW) {
        for (int i0 = 0; i0 < ncm; i0 += PCI) {          // compact rows, one register each (see cV)
          if (i0 > 0) {                                  // more than PCI on one state (a pose prior next to a between factor): refilled in place
#pragma unroll
            for (int q = 0; q < PCI; q++) ldcv(i0 + q, cV[q]);
          }
#pragma unroll
          for (int q = 0; q < PCI; q++) {
            const int i = i0 + q;
            const double V = (i < nc) ? cV[q] : 0.0;
            // this lane's R entry sits six lanes up (row_shl:6, zero where the source lane falls off the row)
            const int rlo = __builtin_amdgcn_update_dpp(0, __double2loint(V), 0x106, 0xf, 0xf, true);
            const int rhi = __builtin_amdgcn_update_dpp(0, __double2hiint(V), 0x106, 0xf, 0xf, true);
            const bool pl = r < Dh;
            const double Lv = pl ? V : 0.0, Rv = pl ? __hiloint2double(rhi, rlo) : 0.0;
            const double ev = row_bcast<B>(V);
            fmac_gather<Dh>(Dacc, V, Lv);                 // D[r][k] += L[k] L[r]   (lane k < 6 of V holds L[k])
            fmac_gather<Dh>(Oacc, V, Rv);                 // O[r][k] += L[k] R[r]
            fmac_gather_nn<Dh, Dh>(RRacc, V, Rv);         // carry[r][k] += R[k] R[r]   (lane 6 + k holds R[k]; V was guarded above)
            gacc = fma(-Lv, ev, gacc);
            grr = fma(-Rv, ev, grr);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else
      for (int i0 = 0; i0 < ncm; i0 += PC) {             // compact rows (velocity-free): six columns each side
#pragma unroll
        for (int q = 0; q < PC; q++) {
          const int i = i0 + q;
          const bool ok = (i < nc) && (r < Dh);
          const double Lv = ok ? cL[q] : 0.0, Rv = ok ? cR[q] : 0.0, ev = (i < nc) ? cE[q] : 0.0;
          fmac_gather<Dh>(Dacc, Lv, Lv);
          fmac_gather<Dh>(Oacc, Lv, Rv);
          fmac_gather<Dh>(RRacc, Rv, Rv);
          gacc = fma(-Lv, ev, gacc);
          grr = fma(-Rv, ev, grr);
          ldc(i + PC, cL[q], cR[q], cE[q]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      GPS_TRA(45);
      {   // Levenberg-Marquardt damping on the diagonal of a real state's D; the padding blocks behind the chain's last state
          // (tail) are identities
        // (added by write_img to the image's diagonal entry, one LDS read-modify-write of the lane's own row: as a select over the
        //  twelve column registers it was 24 instructions per state, for a zero in every Gauss-Newton iteration)
        const bool pad = valid && (s + kimg) >= e && (s + kimg) < ep;
        dgpend = live ? lambda : (pad ? 1.0 : 0.0);
      }
      // the next state: its row range is known, open its rings; fetch the pointers of the state after it
      open_state(kimg + 1, rpn, rpnn, cpn, cpnn, gpn, bqn);
      rpn = rpnn; cpn = cpnn; gpn = gpnn; bqn = bqnn;
      if (btw_on) bqnn = u.btwidx[min(s + kimg + 3, ptr_max)];
      rpnn = fptr[min(s + kimg + 3, ptr_max)];
      cpnn = u.crowptr[min(s + kimg + 3, ptr_max)];
      if (st_on) gpnn = u.gpidx[min(s + kimg + 3, ptr_max)];
      GPS_TRA(46);
    };
    auto write_img = [&](int buf, int kimg) {
      if (rowlane) {
        double *img = IMG + buf * 4 * IS;
#pragma unroll
        for (int k = 0; k < B; k++) { img[ro + k] = Dacc[k]; img[ro + B * BP + k] = Oacc[k]; }
        if (dgpend != 0.0) { const double t = img[ro + rr]; img[ro + rr] = t + dgpend; }   // D[r][r] += lambda (or the padding block's 1)
        img[co + 2 * B * BP] = gacc;
        if (u.gsave && valid) {          // (LM) the gradient: a state's own record, and what the chunk's last rows owe the next separator
          const int jg = s + kimg;
          if (jg < e) u.gsave[(size_t)jg * B + r] = gacc;
          else if (jg == e && e < a.n) u.gsave2[(size_t)e * B + r] = gacc;
        }
      }
    };
    {
      const int g0 = st_on ? u.gpidx[min(s, ptr_max)] : -1;
      ldraw(0, g0);
      open_state(0, fptr[min(s, ptr_max)], rpn, u.crowptr[min(s, ptr_max)], cpn, g0, btw_on ? u.btwidx[min(s, ptr_max)] : -1);
    }
    assemble(0); write_img(0, 0);
    assemble(1); write_img(1, 1);
    GPS_TR(1);
    lds_barrier();                       // P: images 0 and 1 are there
    lds_barrier();                       // Q: ELIM has taken what it needs from image 0
    GPS_TR(2);
    // (round 3: image 2 is assembled AFTER Q, under the elimination of the first block -- ELIM only needs it at the barrier
    //  of step 0.  Before, ELIM sat at Q through this assembly: one block step of every chunk, 3 % of the kernel)
    assemble(2);
    write_img(0, 2);
    for (int t = 0; t < steps; t++) {
      fused_step_prio<B>(2 * t, 1);
      GPS_TR(3 + min(t, 50));
      lds_barrier();                     // step t: image t + 2 is there; image t + 1 is dead from here on
      if (t + 1 < steps) { assemble(t + 3); write_img((t + 1) & 1, t + 3); }
    }
    GPS_TR(60);
    leave();
    return;
  }

  // ================================================================== ELIM (k_chunk_forward_rows on LDS images)
  double Dr[B], Or[B], Fr[B], Gr[B], Ar[B];
  double gr, as_;
  lds_barrier();                         // P
  GPS_TR(1);
#pragma unroll
  for (int k = 0; k < B; k++) {
    Ar[k] = IMG[ro + k];                           // image 0: the separator
    Fr[k] = IMG[ro + B * BP + k];
    Gr[k] = IMG[co + B * BP + k * BP];
    Dr[k] = IMG[4 * IS + ro + k];                  // image 1: the first interior state (or the virtual end record)
    Or[k] = IMG[4 * IS + co + B * BP + k * BP];
  }
  as_ = IMG[co + 2 * B * BP];
  gr = IMG[4 * IS + co + 2 * B * BP];
  lds_barrier();                         // Q
  GPS_TR(2);
  if (valid && !has_int && rowlane) {    // chunk without interior: the separator keeps its coupling, its rows' R^T R is owed
    double *ub = a.up_blk + (size_t)c * BS;
#pragma unroll
    for (int k = 0; k < B; k++) {
      ub[r * B + k] = Ar[k];
      ub[B * B + r * B + k] = right_exists ? Fr[k] : 0.0;
    }
    ub[2 * B * B + r] = as_;
    if (right_exists) {
      double *ua = a.up_add + (size_t)(c + 1) * AS;
#pragma unroll
      for (int k = 0; k < B; k++) ua[r * B + k] = Dr[k];
      ua[B * B + r] = gr;
    }
  }
  for (int t = 0; t < steps; t++) {
    fused_step_prio<B>(2 * t, 0);
    const int j = j0 + t;
    const bool live = j < e, lastb = (j == e - 1);
    const double *cur = IMG + ((t + 1) & 1) * 4 * IS, *nxt = IMG + (t & 1) * 4 * IS;   // images t + 1 and t + 2
#ifdef GPS_ELIM_R5
    double invs = 1.0;
    static_for<0, B>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      const double piv = row_bcast<k>(Dr[k]);
      const double inv = fast_rcp(piv);
      const bool isk = (r == k);
      invs = isk ? inv : invs;
      const double nmp = isk ? 0.0 : -(Dr[k] * inv);
      // row r -= (D~[r][k] / pivot) * row k, the pivot row fused into the multiply-add (fmac_self*): of D~ only the columns
      // right of the pivot still matter, in blocks of four (entries at or left of the pivot inside a block become garbage
      // that nothing reads again)
      if constexpr (B == 12) {
        if (k < 3) fmac_self4<k>(Dr, nmp);
        if (k < 7) fmac_self4<k>(Dr + 4, nmp);
        if (k < 11) fmac_self4<k>(Dr + 8, nmp);
      } else {
        fmac_self_n<k, B>(Dr, nmp);
      }
      fmac_self_n<k, B>(Or, nmp);
      fmac_self_n<k, B>(Fr, nmp);
      fmac_self1<k>(gr, nmp);
    });
    // a pivot that is not positive (or not a number) leaves a reciprocal that is not positive in its own lane: one test per
    // block step instead of one per pivot (lanes without a row keep 1)
    if (!(invs > 0.0) && live) *a.flag = 1;
    __builtin_amdgcn_sched_barrier(0);
    double Ol[B];
#pragma unroll
    for (int k = 0; k < B; k++) Ol[k] = cur[ro + B * BP + k];         // row r of O_j
#pragma unroll
    for (int k = 0; k < B; k++) { Or[k] *= invs; Fr[k] *= invs; }
    gr *= invs;
    if (rowlane) {
#pragma unroll
      for (int k = 0; k < B; k++) {
        OUTR[oc + k * B] = Fr[k];
        OUTR[oc + B * B + k * B] = Or[k];
      }
      OUTR[oc + 2 * B * B] = gr;
    }
    GPS_TR(3 + min(t, 50));
    lds_barrier();                       // step t
    if (live) {
      V2 *dst = reinterpret_cast<V2 *>(a.blk + (size_t)j * BS);
#pragma unroll
      for (int q = 0; q < NV; q++) {
        const int idx = q * 16 + r;
        if (idx < NPC) dst[idx] = *reinterpret_cast<const V2 *>(&OUTR[po + 32 * q]);   // (nontemporal: the iteration +3 us at 1e5 states)
      }
    }
    double Dn[B], Fn[B], gn;
#pragma unroll
    for (int k = 0; k < B; k++) Dn[k] = nxt[ro + k];
    gn = nxt[co + 2 * B * BP];
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, B>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      const double ol = Ol[i], gg = Gr[i];          // subtracted: the negation is the instructions' source modifier
      fmac_bcast_n<i, B, true>(Dn, Or, ol);
      fmac_bcast2<i, true>(gn, as_, gr, ol, gg);
    });
#pragma unroll
    for (int k = 0; k < B; k++) asm volatile("" : "+v"(Dn[k]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < B; k++) Fn[k] = 0.0;
    static_for<0, B>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      const double ol = Ol[i], gg = Gr[i];          // subtracted: the negation is the instructions' source modifier
      fmac_bcast_n<i, B, true>(Fn, Fr, ol);        // F_{j+1} -= O_j[r][i] * (row i of V_j)
      fmac_bcast_n<i, B, true>(Ar, Fr, gg);        // D_sep   -= G_j[r][i] * (row i of V_j)
    });
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < B; k++) asm volatile("" : "+v"(Fn[k]), "+v"(Ar[k]));
    // G_{j+1} = F_{j+1}^T by a transpose through LDS (the V area of the factor image, copied out long ago) instead of
    // the recurrence G_{j+1} = -G_j U_j: 24 LDS operations for 144 multiply-adds
    if (rowlane) {
#pragma unroll
      for (int k = 0; k < B; k++) OUTR[tr + k] = Fn[k];
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < B; k++) {
      Dr[k] = Dn[k]; Fr[k] = Fn[k]; Gr[k] = OUTR[tc + k * BP];
      Or[k] = nxt[co + B * BP + k * BP];
    }
    gr = gn;
    wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
#else
    // Round 6: the elimination wave is the wave a block step waits for (scripts/trace_fused.py: the assembly wave sits at the
    // barrier 0.65-1.0 us of a 4.6-5.6 us step), and alone on a SIMD it issued one instruction per 8 cycles -- it waited for its own
    // dependent chains.  Same arithmetic, same operands, same order of every sum (bit-identical), rearranged so that no latency is
    // exposed to this wave:
    //  * the reciprocal of pivot k + 1 (broadcast, v_rcp_f64, two Newton steps, the multiplier: nine dependent instructions) is
    //    started as soon as column k + 1 of D~ has taken pivot k's update, and its steps are dealt between the row operations on
    //    O, F and g that follow;
    //  * the columns right of the pivot are updated exactly (66 instead of 84 multiply-adds per block: no blocks of four);
    //  * row r of O_j is requested before the elimination instead of behind it; the diagonal block of the next image and the
    //    pieces of the outgoing factor record are requested together behind the barrier and consumed behind the 288 products with
    //    V_j (F_{j+1}, D_sep), the transposed G_{j+1} and the next O^T behind the 144 products with U_j;
    //  * the multiply-adds carry no s_nop (dpp.hpp: the DPP source of every one of them was written a pivot step earlier).
    GPS_TRE(48);
    double Ol[B];
#pragma unroll
    for (int k = 0; k < B; k++) Ol[k] = cur[ro + B * BP + k];         // row r of O_j (image t + 1: published a step ago)
    double invs = 1.0;
    double inv = fast_rcp(row_bcast<0>(Dr[0]));
    static_for<0, B>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      const bool isk = (r == k);
      invs = isk ? inv : invs;
      const double nmp = isk ? 0.0 : -(Dr[k] * inv);
      // row r -= (D~[r][k] / pivot) * row k, the pivot row fused into the multiply-add
      double pn = 1.0, rn = 1.0;
      if constexpr (k + 1 < B) {
        fmac_self1_nn<k>(Dr[k + 1], nmp);                // the next pivot's column first ...
        pn = row_bcast<k + 1>(Dr[k + 1]);                // (guarded: Dr[k + 1] was written by the instruction before)
        rn = __builtin_amdgcn_rcp(pn);                   // ... and its reciprocal under way beneath what follows
        fmac_self_range_nn<k, (k + 2 < B ? k + 2 : B), B>(Dr, nmp);
      }
      __builtin_amdgcn_sched_barrier(0);
      const double e0 = fma(-pn, rn, 1.0);
      fmac_self_range_nn<k, 0, B / 2>(Or, nmp);
      __builtin_amdgcn_sched_barrier(0);
      rn = fma(e0, rn, rn);
      fmac_self_range_nn<k, B / 2, B>(Or, nmp);
      __builtin_amdgcn_sched_barrier(0);
      const double e1 = fma(-pn, rn, 1.0);
      fmac_self_range_nn<k, 0, B / 2>(Fr, nmp);
      __builtin_amdgcn_sched_barrier(0);
      rn = fma(e1, rn, rn);
      fmac_self_range_nn<k, B / 2, B>(Fr, nmp);
      fmac_self1_nn<k>(gr, nmp);
      __builtin_amdgcn_sched_barrier(0);
      inv = rn;
    });
    // a pivot that is not positive (or not a number) leaves a reciprocal that is not positive in its own lane: one test per
    // block step instead of one per pivot (lanes without a row keep 1)
    if (!(invs > 0.0) && live) *a.flag = 1;
    __builtin_amdgcn_sched_barrier(0);
    fused_step_prio<B>(2 * t + 1, 0);
    GPS_TRE(49);
#pragma unroll
    for (int k = 0; k < B; k++) { Or[k] *= invs; Fr[k] *= invs; }
    gr *= invs;
    if (rowlane) {
#pragma unroll
      for (int k = 0; k < B; k++) {
        OUTR[oc + k * B] = Fr[k];
        OUTR[oc + B * B + k * B] = Or[k];
      }
      OUTR[oc + 2 * B * B] = gr;
    }
    GPS_TR(3 + min(t, 50));
    lds_barrier();                       // step t
    GPS_TRE(50);
    V2 pc[NV];
#pragma unroll
    for (int q = 0; q < NV; q++) pc[q] = *reinterpret_cast<const V2 *>(&OUTR[po + 32 * (q * 16 + r < NPC ? q : 0)]);
    double Dn[B], Fn[B], gn;
#pragma unroll
    for (int k = 0; k < B; k++) Dn[k] = nxt[ro + k];
    gn = nxt[co + 2 * B * BP];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < B; k++) Fn[k] = 0.0;
    dpp_guard();
    static_for<0, B>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      const double ol = Ol[i], gg = Gr[i];          // subtracted: the negation is the instructions' source modifier
      fmac_bcast_n_nn<i, B, true>(Fn, Fr, ol);      // F_{j+1} -= O_j[r][i] * (row i of V_j)
      fmac_bcast_n_nn<i, B, true>(Ar, Fr, gg);      // D_sep   -= G_j[r][i] * (row i of V_j)
      fmac_bcast1_nn<i, true>(gn, gr, ol);          // g_{j+1} -= O_j[r][i] * Y_j[i]
      fmac_bcast1_nn<i, true>(as_, gr, gg);         // g_sep   -= G_j[r][i] * Y_j[i]
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int k = 0; k < B; k++) asm volatile("" : "+v"(Fn[k]), "+v"(Ar[k]));
    asm volatile("" : "+v"(gn), "+v"(as_));
    __builtin_amdgcn_sched_barrier(0);
    GPS_TRE(51);
    if (live) {
      V2 *dst = reinterpret_cast<V2 *>(a.blk + (size_t)j * BS);
#pragma unroll
      for (int q = 0; q < NV; q++) {
        const int idx = q * 16 + r;
        if (idx < NPC) dst[idx] = pc[q];             // (nontemporal: the iteration +3 us at 1e5 states)
      }
    }
    // G_{j+1} = F_{j+1}^T by a transpose through LDS (the V area of the factor image: its pieces were read above, and the DS
    // operations of a wave execute in order) instead of the recurrence G_{j+1} = -G_j U_j: 24 LDS operations for 144 multiply-adds
    if (rowlane) {
#pragma unroll
      for (int k = 0; k < B; k++) OUTR[tr + k] = Fn[k];
    }
    wave_lds_sync();
    double Gn[B], On[B];
#pragma unroll
    for (int k = 0; k < B; k++) { Gn[k] = OUTR[tc + k * BP]; On[k] = nxt[co + B * BP + k * BP]; }
    __builtin_amdgcn_sched_barrier(0);
    GPS_TRE(52);
    dpp_guard();
    static_for<0, B>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      fmac_bcast_n_nn<i, B, true>(Dn, Or, Ol[i]);   // D~_{j+1} -= O_j[r][i] * (row i of U_j)
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int k = 0; k < B; k++) asm volatile("" : "+v"(Dn[k]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < B; k++) { Dr[k] = Dn[k]; Fr[k] = Fn[k]; Gr[k] = Gn[k]; Or[k] = On[k]; }
    gr = gn;
    wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
    GPS_TRE(53);
#endif
    if (!tail && live && lastb && rowlane) {
      double *ub = a.up_blk + (size_t)c * BS;
#pragma unroll
      for (int k = 0; k < B; k++) {
        ub[r * B + k] = Ar[k];
        ub[B * B + r * B + k] = right_exists ? Fr[k] : 0.0;
      }
      ub[2 * B * B + r] = as_;
      if (right_exists) {
        double *ua = a.up_add + (size_t)(c + 1) * AS;
#pragma unroll
        for (int k = 0; k < B; k++) ua[r * B + k] = Dr[k];   // -O U + R^T R of the chunk's last rows
        ua[B * B + r] = gr;
      }
    }
  }
  GPS_TR(59);
  if (!tail) { leave(); return; }

  // ================================================================== the level of groups of four, in place
  // All four chunks ended in the same block step (padding above); row grp holds what the in-loop branch above would have
  // sent to memory: the separator's record [Ar | Fr | as_] and the addend [Dr | gr] it owes the next separator.  Both
  // images are dead (the assembly wave is past its last write), they now hold the group: REC[0..3] the records, REC[4] the
  // virtual block beyond the group, TA[0..3] the addends.
  {
    constexpr int G4 = 4;
    double *REC = IMG, *TA = IMG + (G4 + 1) * BS;
    const int c0 = blockIdx.x * 4;
    const int cnt = min(G4, nch - c0);
    wave_lds_sync();
    if (valid && rowlane) {
      double *rc = REC + grp * BS, *ta = TA + grp * AS;
#pragma unroll
      for (int k = 0; k < B; k++) {
        rc[r * B + k] = Ar[k];
        rc[B * B + r * B + k] = right_exists ? Fr[k] : 0.0;
        ta[r * B + k] = Dr[k];
      }
      rc[2 * B * B + r] = as_;
      ta[B * B + r] = gr;
    }
    wave_lds_sync();
    if (valid && rowlane && grp >= 1) {          // block i >= 1 of the group receives what chunk i - 1 owes it
      double *rc = REC + grp * BS;
      const double *ta = TA + (grp - 1) * AS;
#pragma unroll
      for (int k = 0; k < B; k++) rc[r * B + k] += ta[r * B + k];
      rc[2 * B * B + r] += ta[B * B + r];
    }
    if (grp == 0 && rowlane) {                   // the block beyond the group is owed the last chunk's addend
      const bool have = (c0 + cnt < nch);
      double *rc = REC + G4 * BS;
      const double *ta = TA + (cnt - 1) * AS;
#pragma unroll
      for (int k = 0; k < B; k++) {
        rc[r * B + k] = have ? ta[r * B + k] : 0.0;
        rc[B * B + r * B + k] = 0.0;
      }
      rc[2 * B * B + r] = have ? ta[B * B + r] : 0.0;
    }
    wave_lds_sync();
    // two sub-levels: pairs (0, 1), (2, 3) on two DPP rows each (CrStepWide), then the pair (0, 2) on all four (CrStepQuad) --
    // the wave is alone in its workgroup by now, every multiply-add it does not issue is time off the launch (round 4: one row
    // per pair had cost 2 x 834 multiply-adds per lane, this is 612 + 336)
    auto sub_level = [&](int q, auto form) {
      constexpr bool quad = decltype(form)::value;
      const int h = 1 << q;
      const int pq = quad ? 0 : (grp >> 1), role4 = quad ? grp : (grp & 1);
      const int sq = pq * 2 * h, jq = sq + h;
      const bool act = jq < cnt;
      const int nq = (jq + h < cnt) ? jq + h : G4;
      std::conditional_t<quad, CrStepQuad<B>, CrStepWide<B>> st;
      if (__ballot(act) != 0ull) {               // (idle DPP rows recompute block 0; they never store)
        const bool bad = st.compute(REC, act ? sq : 0, act ? jq : 0, r, rr, role4);
        if (bad && act) *a.flag = 1;
      }
      wave_lds_sync();
      if (act && rowlane) st.store_own(REC, sq, jq, r, role4);
      wave_lds_sync();
      if (act && rowlane) {
        if constexpr (quad) { if (role4 < 2) st.add_right(REC, nq, r, role4); }
        else { if (role4 == 0) st.add_right(REC, nq, r); }
      }
      wave_lds_sync();
    };
    sub_level(0, std::false_type{});
    sub_level(1, std::true_type{});
    {   // the factor records of the eliminated separators (blocks 1 .. cnt - 1): level 1's back-substitution reads them
      V2 *dst = reinterpret_cast<V2 *>(a.l1_blk + (size_t)(c0 + 1) * BS);
      const V2 *src = reinterpret_cast<const V2 *>(REC + BS);
      for (int t = lane; t < (cnt - 1) * NPC; t += 64) dst[t] = src[t];
    }
    V2 *ub = reinterpret_cast<V2 *>(a.up_blk + (size_t)blockIdx.x * BS);
    const V2 *s0 = reinterpret_cast<const V2 *>(REC);
    for (int t = lane; t < NPC; t += 64) ub[t] = s0[t];
    V2 *ua = reinterpret_cast<V2 *>(a.up_add + (size_t)(blockIdx.x + 1) * AS);
    const V2 *s4 = reinterpret_cast<const V2 *>(REC + G4 * BS);
    for (int t = lane; t < B * B / 2 + B / 2; t += 64) ua[t] = s4[t < B * B / 2 ? t : t + B * B / 2];
  }
  GPS_TR(60);
  leave();
}

template <typename T> struct BwdArgs {
  const T *blk;   // eliminated records of this level
  T *x;           // n x R x B solutions of this level (+ one slot for the neighbour rank's separator)
  const T *xup;   // solutions of the next level (separators) or null at the top
  int n, m, R, no_sep, last_has_right;
  // k_chunk_backward_rows only (round 4): the level above is the level of groups of four that k_fused_level0 reduced in its tail --
  // a wave's four chunks ARE one such group, so the wave back-substitutes the group itself (l1_blk: that level's factor records,
  // l1_xup: the solutions of the group's first block and of the next group's, one level further up) instead of a launch of
  // k_multi_backward<B, 4> writing xup for it.  Null: xup holds the separators' solutions.
  const T *l1_blk = nullptr, *l1_xup = nullptr;
  int l1_n = 0;   // blocks of that level (= chunks of this one)
};

// x_j = Y_j - U_j x_{j+1} - V_j x_sep, right to left through the chunk.  Lane (k, rr) owns component k of the rhs
// columns rr, rr + RG, ...  The pass is pure streaming (2400 B of factors per 24 FMAs on 12 lanes for Pose3), so it
// lives or dies by how the factors arrive: the wave fetches each record [V | U | Y] as contiguous 16-byte pieces
// (NP wave loads of 1 KiB instead of 25 loads of 96 useful bytes), PFB records ahead in a register ring, and hands
// them to the consuming lanes through LDS.  NP = 16-byte pieces per lane per record: 3 covers R <= 8 for B = 12.
template <typename T, int B, int NP>
__global__ void __launch_bounds__(64) k_chunk_backward(BwdArgs<T> a) {
  const int R = a.R;
  const int BS = 2 * B * B + B * R;
  const int c = blockIdx.x;
  const int s = c * a.m;
  const int e = min(s + a.m, a.n);
  const int lane = threadIdx.x;
  const bool has_sep = !a.no_sep;
  const bool right_exists = has_sep && ((e < a.n) || (a.last_has_right != 0));
  constexpr int RG = 64 / B;
  constexpr int PFB = 4;   // records in flight (round 3: 3 -> 4 with the longer chunks of large chains: 485 -> 452 us at 1e6 states; 6: 510 us)
  const int rr = lane / B, k = lane - rr * B;
  const bool active = rr < RG;
  typedef T V2 __attribute__((ext_vector_type(2)));
  __shared__ T xs[kMaxRhs * B];
  __shared__ T xa[kMaxRhs * B];
  __shared__ T xb[kMaxRhs * B];
  __shared__ __attribute__((aligned(16))) T rec[NP * 128];      // one record [V | U | Y]
  for (int idx = lane; idx < R * B; idx += 64) {
    xs[idx] = has_sep ? a.xup[(size_t)c * R * B + idx] : T(0);
    xa[idx] = right_exists ? a.xup[(size_t)(c + 1) * R * B + idx] : T(0);
  }
  wave_lds_sync();
  if (has_sep)
    for (int idx = lane; idx < R * B; idx += 64) a.x[(size_t)s * R * B + idx] = xs[idx];
  // the right separator of the last chunk lives on the next rank: park its solution in the extra slot x[n] so
  // that the level below finds it where it expects the solution of "separator n"
  if (right_exists && e == a.n)
    for (int idx = lane; idx < R * B; idx += 64) a.x[(size_t)a.n * R * B + idx] = xa[idx];
  const int j0 = has_sep ? s + 1 : s;
  const int pieces = BS / 2;                                      // BS is even (B is)
  V2 ring[PFB][NP];
  // The record loads are UNCONDITIONAL (record index clamped into the chunk, piece index into the record): a load under
  // a branch -- or a ring slot that is only sometimes refilled -- makes the compiler's wait-count bookkeeping give up
  // and drain vmcnt to zero at every step, which turns the 3-deep prefetch into none at all and additionally waits
  // for the x stores (2.6 us per block step measured, a full memory round trip).  Steps past the chunk's first
  // interior record recompute harmlessly; only their x store is masked.
  const int jlo = min(j0, e - 1);
  auto arm = [&](int u, int j) {
    const V2 *src = reinterpret_cast<const V2 *>(a.blk + (size_t)max(j, jlo) * BS);
#pragma unroll
    for (int p = 0; p < NP; p++) ring[u][p] = src[min(lane + 64 * p, pieces - 1)];
  };
  if (j0 < e) {
#pragma unroll
    for (int u = 0; u < PFB; u++) arm(u, e - 1 - u);
  }
  int ping = 0;
  for (int base = e - 1; base >= j0; base -= PFB) {
#pragma unroll
    for (int u = 0; u < PFB; u++) {
      const int j = base - u;
      V2 *rv = reinterpret_cast<V2 *>(rec);
#pragma unroll
      for (int p = 0; p < NP; p++) rv[lane + 64 * p] = ring[u][p];
      arm(u, j - PFB);
      wave_lds_sync();
      const T *cur = ping ? xb : xa;
      T *nx = ping ? xa : xb;
      if (active) {
        T Ur[B], Vr[B];
#pragma unroll
        for (int q = 0; q < B; q++) { Ur[q] = rec[B * B + q * B + k]; Vr[q] = has_sep ? rec[q * B + k] : T(0); }
        for (int r = rr; r < R; r += RG) {
          T v = rec[2 * B * B + r * B + k];
#pragma unroll
          for (int q = 0; q < B; q++) v -= Ur[q] * cur[r * B + q] + Vr[q] * xs[r * B + q];
          nx[r * B + k] = v;
          if (j >= j0) a.x[(size_t)j * R * B + r * B + k] = v;
        }
      }
      wave_lds_sync();
      ping ^= 1;
    }
  }
}

// ---------------------------------------------------------------- row-layout back-substitution (single right-hand side)
//
// k_chunk_backward_rows (round 4): the back-substitution of chains without landmark columns in the row layout of the forward
// kernels -- four chunks per wave, one per 16-lane DPP row, lane r < B holds ROW r of U_j and V_j and component r of every
// solution.  x_j = Y_j - U_j x_(j+1) - V_j x_sep is 2 B multiply-adds per lane with the solution of the state to the right
// broadcast inside the row (v_fmac_f64_dpp row_newbcast); V_j x_sep does not depend on the recurrence and is formed while the
// previous state's result is still in flight.  No LDS, no barriers: the record's 2 B + 1 operands per lane come straight from
// memory (for every q the B lanes of a row read B consecutive doubles of the column-major record [V | U | Y]), PF records ahead.
// The generic kernel above spends a wave per chunk on 12 active lanes and two LDS round trips per state: 4.2 TB/s at block size
// 12 and 2.9 TB/s at block size 6 (profiles/round4_v1, round4_c2_1e6_v1) where a streaming read reaches 5 - 6.
template <int B>
__global__ void __launch_bounds__(64) k_chunk_backward_rows(BwdArgs<double> a) {
  constexpr int BS = 2 * B * B + B;                 // R == 1
  constexpr int PF = (B == 12) ? 3 : 4;             // records in flight per chunk (vmcnt counts to 63: 3 x 25 and 4 x 13 loads, with the consumed slot excluded, stay below)
  const int lane = threadIdx.x, grp = lane >> 4, r = lane & 15;
  const int nch = (a.n + a.m - 1) / a.m;
  const int c = blockIdx.x * 4 + grp;
  const bool valid = c < nch;
  const int cc = valid ? c : 0;
  const int s = cc * a.m;
  const int e = min(s + a.m, a.n);
  const bool rowlane = r < B;
  const int rr = rowlane ? r : 0;
  const bool has_sep = !a.no_sep;
  const bool right_exists = has_sep && ((e < a.n) || (a.last_has_right != 0));
  double xs, xn;
  if (a.l1_blk != nullptr) {
    // The four separators of the wave are a group of the level above: x_0 and x_4 (the next group's first block, or zero behind
    // the chain) come from one level further up, x_2 = Y_2 - U_2 x_4 - V_2 x_0, then x_1 = Y_1 - U_1 x_2 - V_1 x_0 and
    // x_3 = Y_3 - U_3 x_4 - V_3 x_2 (cr_group_backward for G = 4; a block without a right neighbour inside the group takes x_4).
    // DPP row i holds record i of the group (25 operands per lane); two passes of one instruction stream -- the first is row 2's
    // x_2, the second rows 1 and 3 with their own operands -- and the results change rows through ds_bpermute.
    const int g = blockIdx.x, cnt = min(4, a.l1_n - 4 * g);
    const bool have = 4 * g + cnt < a.l1_n;
    const double x0 = rowlane ? a.l1_xup[(size_t)g * B + rr] : 0.0;
    const double x4 = (have && rowlane) ? a.l1_xup[(size_t)(g + 1) * B + rr] : 0.0;
    double rec[2 * B + 1];
    {
      const double *rp = a.l1_blk + (size_t)(4 * g + min(max(grp, 1), max(cnt - 1, 0))) * BS + rr;
#pragma unroll
      for (int q = 0; q < 2 * B; q++) rec[q] = rp[q * B];
      rec[2 * B] = rp[2 * B * B];
    }
    auto solve = [&](double xr, double xl) {                 // Y - U xr - V xl, summed in cr_group_backward's order
      double v = rec[2 * B];
      const double nl = -xl, nr = -xr;
      static_for<0, B>([&](auto qq) { constexpr int q = decltype(qq)::value; fmac_bcast1<q>(v, nr, rec[B + q]); });
      static_for<0, B>([&](auto qq) { constexpr int q = decltype(qq)::value; fmac_bcast1<q>(v, nl, rec[q]); });
      return v;
    };
    const double va = solve(x4, x0);                         // row 2: x_2
    const double x2 = __shfl(va, 32 + r, 64);
    const double vb = solve(grp == 1 ? ((2 < cnt) ? x2 : x4) : x4, grp == 3 ? x2 : x0);   // row 1: x_1, row 3: x_3
    const double x1 = __shfl(vb, 16 + r, 64), x3 = __shfl(vb, 48 + r, 64);
    // chunk grp of the group: its separator is block grp, the state right of it block grp + 1 (x_4 behind the group's last)
    xs = grp == 0 ? x0 : (grp == 1 ? x1 : (grp == 2 ? x2 : x3));
    xn = (grp + 1 < cnt) ? (grp == 0 ? x1 : (grp == 1 ? x2 : x3)) : x4;
    xs = (has_sep && rowlane) ? xs : 0.0;
    xn = (right_exists && rowlane) ? xn : 0.0;
  } else {
    xs = (has_sep && rowlane) ? a.xup[(size_t)cc * B + rr] : 0.0;                 // the chunk's own separator
    xn = (right_exists && rowlane) ? a.xup[(size_t)(cc + 1) * B + rr] : 0.0;      // the state right of the chunk
  }
  if (valid && rowlane) {
    if (has_sep) a.x[(size_t)s * B + r] = xs;
    // the right separator of the last chunk lives on the next rank: park its solution in the extra slot x[n]
    if (right_exists && e == a.n) a.x[(size_t)a.n * B + r] = xn;
  }
  const int j0 = has_sep ? s + 1 : s;
  const int steps = __builtin_amdgcn_readfirstlane(valid ? max(e - j0, 0) : 0);   // lane 0: the wave's first (never shorter) chunk
  if (steps <= 0) return;
  const int jlo = min(j0, e - 1);
  const int dump_off = (a.n + 1) * B + lane;       // 64 values behind the solutions (and the neighbour rank's slot): lanes without a row
  const double nxs = -xs;
  double ring[PF][2 * B + 1];
  // unconditional loads (record index clamped into the chunk): steps past a shorter chunk's first record recompute harmlessly
  auto arm = [&](int u, int j) {
    const double *rec = a.blk + (size_t)max(min(j, e - 1), jlo) * BS + rr;
#pragma unroll
    for (int q = 0; q < 2 * B; q++) ring[u][q] = rec[q * B];          // V[r][q] (q < B), U[r][q - B]
    ring[u][2 * B] = rec[2 * B * B];                                   // Y[r]
  };
  // (slot by slot, in consumption order: vmcnt counts in issue order, and the wait in front of the loop's first product is placed
  //  for the worse of its two predecessors -- with the scheduler free to interleave these loads it was vmcnt(0) on every trip)
#pragma unroll
  for (int u = 0; u < PF; u++) { arm(u, e - 1 - u); __builtin_amdgcn_sched_barrier(0); }
  for (int base = 0; base < steps; base += PF) {
#pragma unroll
    for (int u = 0; u < PF; u++) {
      const int j = e - 1 - base - u;
      // x_j = Y_j + V_j (-x_sep) + U_j (-x_(j+1)): the record's operands are consumed in load order, Y first
      double av = ring[u][2 * B], au = 0.0;
      const double nxn = -xn;
      static_for<0, B>([&](auto qq) {                // V_j x_sep: independent of the recurrence
        constexpr int q = decltype(qq)::value;
        fmac_bcast1<q>(av, nxs, ring[u][q]);
      });
      static_for<0, B>([&](auto qq) {                // U_j x_(j+1)
        constexpr int q = decltype(qq)::value;
        fmac_bcast1<q>(au, nxn, ring[u][B + q]);
      });
      double v = av + au;
      asm volatile("" : "+v"(v));                    // (formed here, not sunk into the loop latch)
      // the slot's operands are consumed: refill it (issued here, not before the products -- a refill in front of them makes the
      // compiler rotate the ring by register copies, and a copy of a register whose load is in flight drains vmcnt)
      arm(u, j - PF);
      const bool live = j >= j0;                     // (a shorter chunk of the wave is done: keep its last solution, store nothing new)
      xn = live ? v : xn;
      // unconditional, branch-free store (a store or an address under a branch breaks the wait-count bookkeeping: vmcnt(0) at the
      // next operand): finished chunks rewrite their first interior solution with the value it already has, lanes without a
      // row store to a dump
      const int off = (valid && rowlane && e > j0) ? max(j, jlo) * B + rr : dump_off;     // (a chunk that is only its separator stores nothing; n B < 2^31)
      a.x[off] = xn;
    }
  }
}

// ------------------------------------------------------------------ segment sharding: reduced interface system

// Every rank contributes one record [D | C | G | RD | Rg] (BS + AS values): its separator block (first state) after
// the local elimination, the coupling C to the next rank's separator and the addend it owes that separator.
// top[r] = [D_r + RD_{r-1} | C_r | G_r + Rg_{r-1}]
template <typename T> __global__ void __launch_bounds__(256) k_iface_build(const T *rec, int P, int B, int R, T *top) {
  const int BS = 2 * B * B + B * R, AS = B * B + B * R, RS = BS + AS;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * BS) return;
  const int r = i / BS, k = i - r * BS;
  T v = rec[(size_t)r * RS + k];
  if (r > 0) {
    const T *pa = rec + (size_t)(r - 1) * RS + BS;
    if (k < B * B) v += pa[k];
    else if (k >= 2 * B * B) v += pa[B * B + (k - 2 * B * B)];
  }
  if (r == P - 1 && k >= B * B && k < 2 * B * B) v = T(0);   // no coupling beyond the last rank
  top[i] = v;
}

// ------------------------------------------------------------------ K6: retract

template <typename T> struct RetractArgs {
  double *pose, *vel;   // the state is fp64 whatever T is: the update is applied in fp64
  int stride, N, R, chart;
  int first;        // first state to update (the halo state of a segment is updated by its own launch)
  const T *x;       // N x R x b, column 0 = delta (indexed from `first`)
  T *partial;       // per-block max |delta|
  const int *flag;  // non-SPD flag of the elimination: when set the states are left untouched (GTSAM throws
                    // IndeterminantLinearSystemException before Values::retract), only |delta|_inf is reported
  // deferred reduction (round 3): the per-block error sums of this iteration's linearisation, summed by one extra
  // workgroup at the end of the grid instead of by a launch of their own right behind the linearisation
  const double *red_in = nullptr;
  int red_n = 0;
  double *red_out = nullptr;
};

template <typename T, int MF>
__global__ void __launch_bounds__(128) k_retract(RetractArgs<T> a) {
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd, b = 2 * d;
  if (a.red_in != nullptr && blockIdx.x == gridDim.x - 1) {     // the extra workgroup: pending error sum, fixed order
    const double acc = strided_fold<double, 128>(a.red_in, a.red_n, 0);
    const double r = block_sum(acc);
    if (threadIdx.x == 0) *a.red_out = r;
    return;
  }
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = a.first + li;
  T mx = T(0);
  if (li < a.N) {
    const T *dl = a.x + (size_t)li * a.R * b;
    double dlt[b], x[pd], out[pd];
#pragma unroll
    for (int k = 0; k < b; k++) { dlt[k] = (double)dl[k]; mx = fmax(mx, abs_or_inf(dl[k])); }
    if (!(a.flag && *a.flag)) {
#pragma unroll
      for (int k = 0; k < pd; k++) x[k] = a.pose[(size_t)k * a.stride + i];
      PoseFactors<double, MF, false>::retract(x, dlt, a.chart, out);
#pragma unroll
      for (int k = 0; k < pd; k++) a.pose[(size_t)k * a.stride + i] = out[k];
#pragma unroll
      for (int k = 0; k < d; k++) a.vel[(size_t)k * a.stride + i] += dlt[d + k];
    }
  }
  const T r = block_max(mx);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
}

}  // namespace gps
