// kernels.hpp -- HIP kernels of the GP-SLAM Gauss-Newton iteration (gfx950, wave64).
//
// HBM data layout (DESIGN.md has the full picture):
//   states      SoA   pose[k * stride + i], vel[k * stride + i]      (coalesced across factors)
//   row table   AoS   one record per whitened Jacobian row: rowLR[rho][2b] = [dL | dR], rowE[rho]
//                     rows are grouped by the LEFT state of their factor; rowptr[s] .. rowptr[s+1]
//   blocks      AoS   per state one record [D (b x b) | O (b x b) | G (R cols of b)], O_s = H[s+1, s]
//   solver      the same records, overwritten in place by the elimination with [V | U | Y]
//
// Kernels:
//   k_gp          batched GaussianProcessPrior*::evaluateError + H1..H4 (+ whitening)     [K1]
//   k_unary/k_between   PriorFactor / BetweenFactor rows
//   k_assemble    J^T J / J^T e per state from the row table (no atomics, fixed order)     [K3]
//   k_chunk_forward / k_chunk_backward   partitioned block Gauss-Jordan, one wave per chunk,
//                 one panel column per lane, pivot broadcast through v_readlane            [K4]
//   k_retract     x <- x (+) delta, |delta|_inf                                            [K6]
#pragma once

#include "factors.hpp"

namespace gps {

constexpr int kMaxRhs = 28;  // 3B + R <= 64 for B = 12

template <typename T> struct UMat { T u[36]; };  // chol_upper(Qc^-1), d x d row-major

// ------------------------------------------------------------------ reductions

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
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
template <typename T> __global__ void __launch_bounds__(256) k_final_reduce(const T *in, int n, double *out, int mode) {
  T acc = T(0);
  for (int i = threadIdx.x; i < n; i += 256) acc = mode ? fmax(acc, in[i]) : acc + in[i];
  const T r = mode ? block_max(acc) : block_sum(acc);
  if (threadIdx.x == 0) *out = (double)r;
}

// ------------------------------------------------------------------ K1: GP prior rows

template <typename T> struct GpArgs {
  const T *pose, *vel;    // SoA
  int stride;             // SoA stride (>= N + 1)
  int count;
  const int *left;        // left state of factor f
  const T *dt;
  const int *row0;        // first row of factor f in the row table
  T *rowLR, *rowE;        // MODE 0
  T *partial;             // per-block error partial sums
  T *out_e, *out_H;       // MODE 2: API layout
  UMat<T> U;
};

// MODE 0: whitened rows + error; MODE 1: error only; MODE 2: unwhitened e + H1..H4 in API layout
template <typename T, int MF, int MODE>
__global__ void __launch_bounds__(128) k_gp(GpArgs<T> a) {
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd, b = 2 * d;
  constexpr bool JAC = (MODE != 1);
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  T err = T(0);
  if (f < a.count) {
    const int i = a.left[f];
    const T dt = a.dt[f];
    T p1[pd], p2[pd], v1[d], v2[d];
#pragma unroll
    for (int k = 0; k < pd; k++) { p1[k] = a.pose[(size_t)k * a.stride + i]; p2[k] = a.pose[(size_t)k * a.stride + i + 1]; }
#pragma unroll
    for (int k = 0; k < d; k++) { v1[k] = a.vel[(size_t)k * a.stride + i]; v2[k] = a.vel[(size_t)k * a.stride + i + 1]; }
    T e[b];
    T Jt[JAC ? d * 2 * b : 1], Jb[JAC ? d * 2 * b : 1];
    GpPrior<T, MF, JAC>::eval(p1, v1, p2, v2, dt, e, Jt, Jb);
    if (MODE == 2) {
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
    } else {
      // whitening R = chol_upper(Q^-1(dt)) = [[sa, sb], [0, sc]] (x) U, U = chol_upper(Qc^-1)
      // (noiseModel::Gaussian::Covariance(calcQ(Qc, dt)), GaussianProcessPriorPose3.h:46)
      const T sq = sqrt(dt);
      const T sa = T(3.4641016151377545870548926830117) / (dt * sq);  // sqrt(12 / dt^3)
      const T sb = T(-1.7320508075688772935274463415059) / sq;        // (-6 / dt^2) / sa
      const T sc = T(1) / sq;                                          // sqrt(4/dt - sb^2)
      const int row0 = (MODE == 0) ? a.row0[f] : 0;
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
        if (MODE == 0) {
          a.rowE[row0 + rho] = wt;
          a.rowE[row0 + d + rho] = wb;
          T *rt = a.rowLR + (size_t)(row0 + rho) * 2 * b;
          T *rb = a.rowLR + (size_t)(row0 + d + rho) * 2 * b;
#pragma unroll
          for (int col = 0; col < 2 * b; col++) {
            T vt = T(0), vb = T(0);
#pragma unroll
            for (int r = rho; r < d; r++) {
              vt += a.U.u[rho * d + r] * (sa * Jt[r * 2 * b + col] + sb * Jb[r * 2 * b + col]);
              vb += a.U.u[rho * d + r] * Jb[r * 2 * b + col];
            }
            rt[col] = vt;
            rb[col] = sc * vb;
          }
        }
      }
    }
  }
  if (MODE != 2) {
    const T tot = block_sum(T(0.5) * err);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = tot;
  }
}

// ------------------------------------------------------------------ unary / between rows

template <typename T> struct FacArgs {
  const T *pose, *vel;
  int stride, count, chart;
  const int *idx;       // state (left state for between)
  const T *meas;        // count x (pd or d)
  const T *sig;         // count x d
  const int *row0;
  T *rowLR, *rowE;
  T *partial;
};

// KIND 0: PriorFactor<Pose>, 1: PriorFactor<Vector> on the velocity, 2: BetweenFactor<Pose>(x_i, x_i+1)
template <typename T, int MF, int KIND, bool JAC>
__global__ void __launch_bounds__(128) k_simple(FacArgs<T> a) {
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd, b = 2 * d;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  T err = T(0);
  if (f < a.count) {
    const int i = a.idx[f];
    T e[d], H1[JAC ? d * d : 1], H2[JAC ? d * d : 1];
    if (KIND == 1) {
#pragma unroll
      for (int k = 0; k < d; k++) e[k] = a.vel[(size_t)k * a.stride + i] - a.meas[(size_t)f * d + k];
    } else {
      T x1[pd], m[pd];
#pragma unroll
      for (int k = 0; k < pd; k++) { x1[k] = a.pose[(size_t)k * a.stride + i]; m[k] = a.meas[(size_t)f * pd + k]; }
      if (KIND == 0) {
        PoseFactors<T, MF, JAC>::prior(m, x1, a.chart, e, H1);
      } else {
        T x2[pd];
#pragma unroll
        for (int k = 0; k < pd; k++) x2[k] = a.pose[(size_t)k * a.stride + i + 1];
        PoseFactors<T, MF, JAC>::between(m, x1, x2, a.chart, e, H1, H2);
      }
    }
    const int row0 = JAC ? a.row0[f] : 0;
#pragma unroll
    for (int r = 0; r < d; r++) {
      const T w = T(1) / a.sig[(size_t)f * d + r];
      const T we = e[r] * w;
      err += we * we;
      if (JAC) {
        a.rowE[row0 + r] = we;
        T *row = a.rowLR + (size_t)(row0 + r) * 2 * b;
#pragma unroll
        for (int c = 0; c < 2 * b; c++) row[c] = T(0);
        if (KIND == 1) {
          row[d + r] = w;
        } else {
#pragma unroll
          for (int c = 0; c < d; c++) row[c] = w * H1[r * d + c];
          if (KIND == 2) {
#pragma unroll
            for (int c = 0; c < d; c++) row[b + c] = w * H2[r * d + c];
          }
        }
      }
    }
  }
  const T tot = block_sum(T(0.5) * err);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = tot;
}

// ------------------------------------------------------------------ K3: assemble normal equations

template <typename T> struct AsmArgs {
  int N, R;               // states, rhs columns (1 + border)
  const int *rowptr;      // N + 2 entries; rows of left state s: [rowptr[s], rowptr[s+1]); rowptr[-1] handled by s > 0
  const T *rowLR, *rowE;
  const T *rowM;          // M x ld (border) or null
  const int *rowLm;       // landmark id per row or -1
  int ld;
  T *blk;                 // N records [D | O | G]
};

// thread (s, c) builds row c of D_s and O_s and entry c of every rhs column of state s
template <typename T, int B>
__global__ void __launch_bounds__(192) k_assemble(AsmArgs<T> a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = t / B, c = t - s * B;
  if (s >= a.N) return;
  const int BS = 2 * B * B + B * a.R;
  T *bp = a.blk + (size_t)s * BS;
  T D[B], O[B];
  T g = T(0);
#pragma unroll
  for (int k = 0; k < B; k++) { D[k] = T(0); O[k] = T(0); }
  for (int r = 1; r < a.R; r++) bp[2 * B * B + r * B + c] = T(0);
  // rows whose factor has left state s: left block -> D_s, O_s (with the right block), g_s
  for (int rho = a.rowptr[s]; rho < a.rowptr[s + 1]; rho++) {
    const T *row = a.rowLR + (size_t)rho * 2 * B;
    const T Lc = row[c], Rc = row[B + c];
    const T e = a.rowE[rho];
#pragma unroll
    for (int k = 0; k < B; k++) {
      const T Lk = row[k];
      D[k] += Lc * Lk;
      O[k] += Rc * Lk;
    }
    g -= Lc * e;
    if (a.rowM) {
      const int lm = a.rowLm[rho];
      if (lm >= 0)
        for (int q = 0; q < a.ld; q++) bp[2 * B * B + (1 + lm * a.ld + q) * B + c] += Lc * a.rowM[(size_t)rho * a.ld + q];
    }
  }
  // rows whose factor has left state s-1: right block -> D_s, g_s
  if (s > 0) {
    for (int rho = a.rowptr[s - 1]; rho < a.rowptr[s]; rho++) {
      const T *row = a.rowLR + (size_t)rho * 2 * B + B;
      const T Rc = row[c];
      const T e = a.rowE[rho];
#pragma unroll
      for (int k = 0; k < B; k++) D[k] += Rc * row[k];
      g -= Rc * e;
      if (a.rowM) {
        const int lm = a.rowLm[rho];
        if (lm >= 0)
          for (int q = 0; q < a.ld; q++) bp[2 * B * B + (1 + lm * a.ld + q) * B + c] += Rc * a.rowM[(size_t)rho * a.ld + q];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < B; k++) { bp[c * B + k] = D[k]; bp[B * B + c * B + k] = O[k]; }
  bp[2 * B * B + c] = g;
}

// ------------------------------------------------------------------ K4: partitioned block Gauss-Jordan

__device__ __forceinline__ double lane_bcast(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <typename T> struct FwdArgs {
  T *blk;            // n records of this level, eliminated in place
  const T *add;      // n addend records [RD | Rg] or null (level 0)
  T *up_blk;         // records of the next level (one per chunk)
  T *up_add;         // addends of the next level (nchunks + 1)
  int n, m, R;
  int no_sep;        // 1: top level, a single chunk with no separator (plain sequential elimination)
  int last_has_right;// the last chunk has a right separator outside this level (next rank)
  T lambda;          // LM damping added to the diagonal of D while loading (level 0 only)
  int *flag;         // set to 1 if a pivot is not positive
};

// One wave per chunk.  Panel columns: [ D~ (B) | O^T (B) | F (B) | rhs (R) ], one column per lane.
// Block j:  D~_j x_j + O_j^T x_{j+1} + F_j x_sep = g~_j.  Gauss-Jordan on D~_j turns the other columns into
// U_j = D~^-1 O_j^T, V_j = D~^-1 F_j, Y_j = D~^-1 g~_j (stored in place for the back-substitution) and
//   D~_{j+1} = D_{j+1} - O_j U_j,  F_{j+1} = -O_j V_j,  g~_{j+1} = g_{j+1} - O_j Y_j,
//   separator:  D_sep -= F_j^T V_j,  g_sep -= F_j^T Y_j.
// The lanes that held O^T become the D~ lanes of the next block (roles swap), so nothing moves.
template <typename T, int B>
__global__ void __launch_bounds__(64) k_chunk_forward(FwdArgs<T> a) {
  const int R = a.R;
  const int BS = 2 * B * B + B * R;
  const int AS = B * B + B * R;
  const int c = blockIdx.x;
  const int s = c * a.m;
  const int e = min(s + a.m, a.n);
  const int lane = threadIdx.x;
  const bool has_sep = !a.no_sep;
  const bool right_exists = (e < a.n) || (a.last_has_right != 0);
  __shared__ T ldsO[B * B];
  __shared__ T ldsF[B * B];
  int dbase = 0, obase = B;
  const int cF = lane - 2 * B;
  const bool isF = has_sep && cF >= 0 && cF < B;
  const int cR = lane - 3 * B;
  const bool isR = cR >= 0 && cR < R;
  T col[B], acc[B];
#pragma unroll
  for (int k = 0; k < B; k++) { col[k] = T(0); acc[k] = T(0); }

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
  if (j0 < e) {
    if (lane < B) load_D_row(j0, lane, col);
    else if (lane < 2 * B) load_O_row(j0, lane - B, col);
    else if (isF) {
      const T *p = a.blk + (size_t)s * BS + B * B;  // F_first = O_s (acts on x_sep in row s+1)
#pragma unroll
      for (int k = 0; k < B; k++) col[k] = p[k * B + cF];
    } else if (isR) load_G_col(j0, cR, col);
  }

  for (int j = j0; j < e; ++j) {
    const bool last = (j == e - 1);
    const int cD = lane - dbase, cO = lane - obase;
    const bool isD = cD >= 0 && cD < B, isO = cO >= 0 && cO < B;
    T nxt[B];
#pragma unroll
    for (int k = 0; k < B; k++) nxt[k] = T(0);
    if (!last) {  // operands of block j+1, fetched early so the HBM latency hides under the elimination
      if (isD) load_O_row(j + 1, cD, nxt);
      else if (isO) load_D_row(j + 1, cO, nxt);
      else if (isR) load_G_col(j + 1, cR, nxt);
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
      const T p = lane_bcast(col[k], db + k);
      if (!(p > T(0)) && lane == 0) *a.flag = 1;
      const T rowk = col[k] / p;
#pragma unroll
      for (int i = 0; i < B; i++) {
        if (i != k) {
          const T mlt = lane_bcast(col[i], db + k);
          col[i] -= mlt * rowk;
        }
      }
      col[k] = rowk;
    }
    T *bp = a.blk + (size_t)j * BS;
    if (isO) {
#pragma unroll
      for (int k = 0; k < B; k++) bp[B * B + cO * B + k] = col[k];  // U_j, column cO
    } else if (isF) {
#pragma unroll
      for (int k = 0; k < B; k++) bp[cF * B + k] = col[k];          // V_j, column cF
    } else if (isR) {
#pragma unroll
      for (int k = 0; k < B; k++) bp[2 * B * B + cR * B + k] = col[k];  // Y_j
    }
    __syncthreads();
    if (has_sep && (isF || isR)) {
#pragma unroll
      for (int q = 0; q < B; q++) {
        T sacc = T(0);
#pragma unroll
        for (int k = 0; k < B; k++) sacc += ldsF[q * B + k] * col[k];
        acc[q] += sacc;
      }
    }
    T nw[B];
    if (isO || isF || isR) {
#pragma unroll
      for (int r = 0; r < B; r++) {
        T sacc = T(0);
#pragma unroll
        for (int k = 0; k < B; k++) sacc += ldsO[r * B + k] * col[k];
        nw[r] = -sacc;
      }
    } else {
#pragma unroll
      for (int r = 0; r < B; r++) nw[r] = T(0);
    }
    if (!last) {
      if (isO || isR) {
#pragma unroll
        for (int r = 0; r < B; r++) col[r] = nxt[r] + nw[r];
      } else if (isF) {
#pragma unroll
        for (int r = 0; r < B; r++) col[r] = nw[r];
      } else if (isD) {
#pragma unroll
        for (int r = 0; r < B; r++) col[r] = nxt[r];
      }
      const int tswap = dbase; dbase = obase; obase = tswap;
    } else if (right_exists && has_sep) {
      T *ua = a.up_add + (size_t)(c + 1) * AS;
      if (isO) {
#pragma unroll
        for (int k = 0; k < B; k++) ua[cO * B + k] = nw[k];              // -O U  (symmetric)
      } else if (isR) {
#pragma unroll
        for (int k = 0; k < B; k++) ua[B * B + cR * B + k] = nw[k];      // -O Y
      } else if (isF) {
        T *uo = a.up_blk + (size_t)c * BS + B * B;
#pragma unroll
        for (int r = 0; r < B; r++) uo[r * B + cF] = nw[r];              // C = -O V, couples sep c -> sep c+1
      }
    }
    __syncthreads();
  }

  if (has_sep) {
    T *ub = a.up_blk + (size_t)c * BS;
    if (isF) {
      T dr[B];
      load_D_row(s, cF, dr);
#pragma unroll
      for (int k = 0; k < B; k++) ub[cF * B + k] = dr[k] - acc[k];
    } else if (isR) {
      T gr[B];
      load_G_col(s, cR, gr);
#pragma unroll
      for (int k = 0; k < B; k++) ub[2 * B * B + cR * B + k] = gr[k] - acc[k];
    }
    if (j0 >= e) {  // chunk without interior: the separator keeps its original coupling
      if (lane < B) {
        T orow[B];
        load_O_row(s, lane, orow);
#pragma unroll
        for (int k = 0; k < B; k++) ub[B * B + lane * B + k] = right_exists ? orow[k] : T(0);
        if (right_exists) {
          T *ua = a.up_add + (size_t)(c + 1) * AS;
#pragma unroll
          for (int k = 0; k < B; k++) ua[lane * B + k] = T(0);
        }
      }
      if (isR && right_exists) {
        T *ua = a.up_add + (size_t)(c + 1) * AS;
#pragma unroll
        for (int k = 0; k < B; k++) ua[B * B + cR * B + k] = T(0);
      }
    } else if (!right_exists) {
      if (lane < B) {
#pragma unroll
        for (int k = 0; k < B; k++) ub[B * B + lane * B + k] = T(0);
      }
    }
  }
}

template <typename T> struct BwdArgs {
  const T *blk;   // eliminated records of this level
  T *x;           // n x R x B solutions of this level
  const T *xup;   // solutions of the next level (separators) or null at the top
  int n, m, R, no_sep, last_has_right;
};

// x_j = Y_j - U_j x_{j+1} - V_j x_sep, right to left through the chunk
template <typename T, int B>
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
  const int rr = lane / B, k = lane - rr * B;
  const bool active = rr < RG;
  __shared__ T xs[kMaxRhs * B];
  __shared__ T xa[kMaxRhs * B];
  __shared__ T xb[kMaxRhs * B];
  for (int idx = lane; idx < R * B; idx += 64) {
    xs[idx] = has_sep ? a.xup[(size_t)c * R * B + idx] : T(0);
    xa[idx] = right_exists ? a.xup[(size_t)(c + 1) * R * B + idx] : T(0);
  }
  __syncthreads();
  if (has_sep)
    for (int idx = lane; idx < R * B; idx += 64) a.x[(size_t)s * R * B + idx] = xs[idx];
  const int j0 = has_sep ? s + 1 : s;
  int ping = 0;
  for (int j = e - 1; j >= j0; --j) {
    const T *bp = a.blk + (size_t)j * BS;
    const T *cur = ping ? xb : xa;
    T *nx = ping ? xa : xb;
    if (active) {
      T Ur[B], Vr[B];
#pragma unroll
      for (int q = 0; q < B; q++) {
        Ur[q] = bp[B * B + q * B + k];
        Vr[q] = has_sep ? bp[q * B + k] : T(0);
      }
      for (int r = rr; r < R; r += RG) {
        T v = bp[2 * B * B + r * B + k];
#pragma unroll
        for (int q = 0; q < B; q++) v -= Ur[q] * cur[r * B + q] + Vr[q] * xs[r * B + q];
        nx[r * B + k] = v;
        a.x[(size_t)j * R * B + r * B + k] = v;
      }
    }
    __syncthreads();
    ping ^= 1;
  }
}

// ------------------------------------------------------------------ K6: retract

template <typename T> struct RetractArgs {
  T *pose, *vel;
  int stride, N, R, chart;
  const T *x;       // N x R x b, column 0 = delta
  T *partial;       // per-block max |delta|
};

template <typename T, int MF>
__global__ void __launch_bounds__(128) k_retract(RetractArgs<T> a) {
  constexpr int d = MTraits<MF>::d, pd = MTraits<MF>::pd, b = 2 * d;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  T mx = T(0);
  if (i < a.N) {
    const T *dl = a.x + (size_t)i * a.R * b;
    T dlt[b], x[pd], out[pd];
#pragma unroll
    for (int k = 0; k < b; k++) { dlt[k] = dl[k]; mx = fmax(mx, fabs(dlt[k])); }
#pragma unroll
    for (int k = 0; k < pd; k++) x[k] = a.pose[(size_t)k * a.stride + i];
    PoseFactors<T, MF, false>::retract(x, dlt, a.chart, out);
#pragma unroll
    for (int k = 0; k < pd; k++) a.pose[(size_t)k * a.stride + i] = out[k];
#pragma unroll
    for (int k = 0; k < d; k++) a.vel[(size_t)k * a.stride + i] += dlt[d + k];
  }
  const T r = block_max(mx);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
}

}  // namespace gps
