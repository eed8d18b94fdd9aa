// api_common.hpp -- what the three translation units of the library share: the handle, the host helpers and the kernel launchers
// (round 3: api.hip was one 3.5-minute translation unit; the fp64 and fp32 halves compile side by side now)
#pragma once
// api.hip -- host side of libgpslam_hip.so: the opaque handle, the graph-compile pass and the C ABI
// declared in include/gpslam_hip.h.  No CPU fallback exists anywhere in this library: every compute entry
// point launches HIP kernels and fails with GPSLAM_E_HIP if the device is unusable.
#include "../../include/gpslam_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "devbuf.hpp"
#include "kernels.hpp"
#include "fatsep.hpp"
#include "upper.hpp"

using namespace gps;

namespace {

struct Level {
  int n = 0, m = 0, nch = 0;
  DevBuf blk, add, x;
};

struct SimpleSet {  // PriorFactor / BetweenFactor style factors: index + measurement + sigmas
  std::vector<int32_t> idx;
  std::vector<double> meas, sig;
  int width = 0;  // doubles per measurement
  DevBuf d_idx, d_meas, d_sig, d_row0;
  int count() const { return (int)idx.size(); }
  void release() { d_idx.release(); d_meas.release(); d_sig.release(); d_row0.release(); }
};

struct MeasSet {  // measurement factors (kernels.hpp FKind)
  int kind = 0, rows = 1, mw = 1;
  bool two = false, haslm = false, interp = false;
  std::vector<int32_t> idx, lm;
  std::vector<double> meas, sig, dt, tau;
  // body_P_sensor / Cal3_S2 per factor: a table of distinct kMeasAux-wide entries and one index per factor
  std::vector<double> aux;
  std::vector<int32_t> aidx;
  bool any_aux = false;                // some factor of this kind carries a sensor transform or a calibration
  // noiseModel::Gaussian on factors of this kind (gpslam_hip_set_meas_covariance): rows x rows square-root information per
  // factor, diag(1 / sigma) for those that kept their diagonal model; empty: every factor is diagonal
  std::vector<double> sqi;
  DevBuf d_idx, d_lm, d_meas, d_sig, d_coef, d_row0, d_aux, d_aidx, d_sqi;
  DevBuf d_irow0;                      // first row of each factor in the table of 16-double interpolated rows (handle: irow_ok)
  int count() const { return (int)idx.size(); }
  void release() { d_idx.release(); d_lm.release(); d_meas.release(); d_sig.release(); d_coef.release(); d_row0.release(); d_aux.release(); d_aidx.release(); d_sqi.release(); d_irow0.release(); }
};

}  // namespace

struct gpslam_hip_handle {
  gpslam_hip_config_v2 cfg;   // (gpslam_hip_create maps a v1 config onto it)
  int mf = 0, d = 0, pd = 0, b = 0, ld = 0;
  int vw = 0;                 // Pose3: velocities are world-frame [v; w] (cfg.velocity, the *Pose3VW factors)
  int N = 0, L = 0, stride = 0, R = 1, nl = 0;
  bool own_stream = true;
  hipStream_t stream = nullptr;
  hipStream_t aux_stream = nullptr;   // side stream for the light factor kernels (launch_factors)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // the fused level-0 launch of a timed iteration carries its own start / stop events (hipExtLaunchKernelGGL: the dispatch's own
  // time stamps, what rocprofv3's kernel trace reads) -- events recorded AROUND a launch add their marker packets to it (7 us of 149)
  hipEvent_t ev_l0a = nullptr, ev_l0b = nullptr;
  bool l0_ext = false;        // ... and this iteration's launch took them
  bool l0_stamps = false;     // gpslam_hip_set_level0_stamps: timed iterations do so (off: events around the launch, as in rounds 1-5)
  double Qc[36], U[36];
  std::vector<double> h_lmk;
  DevBuf pose, vel, lmk, pose_bak, vel_bak, lmk_bak;
  // factors
  std::vector<int32_t> gp_left;
  std::vector<double> gp_dt;
  DevBuf d_gp_left, d_gp_dt, d_gp_row0;
  // one Qc_model per GP prior (gpslam_hip_add_gp_priors_qc; GaussianProcessPriorPose3.h:43-49): gp_q[f] = 0: the handle's
  // shared Qc (set_qc), k >= 1: entry k - 1 of gp_Utab (36 doubles each, chol_upper(Qc^-1)).  With more than one distinct
  // Qc, compile() orders the device-side factor arrays by Qc (gp_perm: device position -> position in the order added)
  // and the linearisation runs one launch per group (gp_groups: {q, first, count}), each with its U as a kernel argument.
  std::vector<int32_t> gp_q;
  std::vector<double> gp_Utab;
  std::vector<int32_t> gp_perm;
  std::vector<int32_t> gp_groups;
  int gp_single_q = 0;      // > 0: every GP prior came through add_gp_priors_qc with the SAME Qc (entry gp_single_q - 1 of gp_Utab):
                            // one group, one launch, and the structured-record path of k_fused_level0 stays available
  SimpleSet pri, vpri, btw, lpri;
  // loop closures (gpslam_hip_add_between_pairs; kernels.hpp "loop closures"): clo.idx = first state, clo_second = second state
  SimpleSet clo;
  std::vector<int32_t> clo_second;
  DevBuf d_clo_second, clo_A, clo_Y;
  int nclo = 0, nc = 0;       // closures of the compiled graph, their right-hand-side columns (nclo * d)
  MeasSet ms[kNumMeasKinds];
  // row table
  int M = 0;
  DevBuf rowLR, rowE, rowM, rowLm, rowptr;
  DevBuf partial;
  // landmark border
  int nlmrows = 0;
  DevBuf rowC, rowCE, crowptr;   // compact row table (pose priors, between factors) and its row pointers
  int Mc = 0;
  DevBuf lmrow, lmrow_state, lmrow_ptr, lm_t, lm_S, lm_dL;   // lm_S = [S (nl x R) | gL (nl)]
  DevBuf lm_chunk_lm, lm_chunk_j0, lm_chunk_j1, lm_chunk_ptr, lm_part;   // chunked reduction of the landmark rows
  int nlmchunks = 0;
  // solver
  std::vector<Level> lv;
  DevBuf gsave, dvec;
  // segment sharding
  DevBuf halo_add, iface_send, iface_recv, top_blk, top_x;
  DevBuf scal, flag, api_e, api_H;
  // landmark elimination at scale (fatsep.hpp): segments + fat separators instead of the dense border
  FatSepPlan fs;
  DevBuf lm_gL;             // undamped landmark gradient of the segmented path (the dense path keeps it behind lm_S)
  bool upper_ok = false;    // the levels above level 0 run as LDS-resident cyclic reduction (upper.hip)
  bool fuse_ok = false;     // k_fused_level0 applies to this graph (compile())
  bool fuse_now = false;    // ... and the iteration being enqueued uses it (enqueue_gn)
  bool struct_ok = false;   // the GP priors may reach k_fused_level0 as structured records (GpArgs::gps) instead of rows
  bool struct_now = false;  // ... and the linearisation / elimination being enqueued do so
  // block size 6 (SE(2), SO(3), 3-D linear chains): the GP priors as 32-double records (kGp3*) that k_assemble_ghost and
  // k_fused_level0<1, double, 6> decode; rows3: the launch being enqueued needs real rows after all (gpslam_hip_get_rows)
  bool struct3_ok = false, rows3 = false;
  bool pure6 = false;       // block size 6: nothing but d = 3 GP records in the full-width row table (fused level 0 at every size)
  bool odd_many = false;    // SE(3) records: more other full-width rows than k_fused_level0<2> fetches without a ring (-> <3>)
  DevBuf gps, gpidx, dU, gsave2;
  DevBuf simd_cnt;          // k_fused_level0 (GPS_ROLE_SWAP): wave-0 count per SIMD of the chip, zero between launches
  // BetweenFactor<Pose3> of a chain on the structured path as 48-double records (kBtw*): at most one per left state
  bool btw_rec_ok = false;
  bool gp_rows_lead = true, btw_rows_trail = true;   // compile(): the row placement the record decoders rely on holds
  // SE(3) records + interpolated measurement rows as 16-double lines (round 5, k_fused_level0<4>): every full-width row besides the
  // GP priors' belongs to a GPInterpolatedGPSFactorPose3 on an interval that has a GP prior
  bool irow_ok = false;
  DevBuf rowI, irowptr;
  // the host's collectives (gpslam_hip_set_collectives): with them the optimiser loops run on sharded handles / split pieces
  gpslam_hip_all_gather_fn coll_gather = nullptr;
  gpslam_hip_all_reduce_sum_fn coll_reduce = nullptr;
  void *coll_user = nullptr;
  DevBuf coll_s, coll_r;     // 8 doubles of this rank's scalars, nranks x 8 gathered
  DevBuf brec, btwidx;
  // Gauss-Newton runs (gpslam_hip_run_gn): the retraction of an iteration folded into the next iteration's K1 (kernels.hpp: PendUpd).
  // pend_ok: compile() found the graph eligible;  pend_upd: a solve's update sits in the level-0 solution array, not yet applied
  bool pend_ok = false, pend_upd = false;
  bool keep_flag = false;   // inside gpslam_hip_run_gn on a sharded handle / split piece: phase 1 leaves the non-positive-pivot flag alone
  bool gsave_now = false;   // the fused kernel being enqueued stores the gradient (Levenberg-Marquardt trials)
  int U_version = 0, dU_version = -1;   // set_qc after compile(): the device copy of U is refreshed before its next use
  bool compiled = false;
  double last_ms[5] = {0, 0, 0, 0, 0};
  // deferred reductions inside run_gn / iterate_gn (chains without landmarks, unsharded): the error partial sums of the
  // linearisation are summed by an extra workgroup of k_retract, the |delta|_inf partial maxima of the retraction by an
  // extra workgroup of the NEXT iteration's k_lin -- two launches (+ their gaps) less per iteration, same values, same order
  DevBuf partial2;            // the retraction's per-block maxima (its own buffer: the linearisation reuses `partial`)
  bool defer_err = false, defer_dmax = false;   // what the call being enqueued may defer (enqueue_gn)
  int pend_err_n = 0, pend_err_slot = 0;        // pending: error partials in `partial`
  int pend_dmax_n = 0, pend_dmax_slot = 0;      // pending: maxima in `partial2`
  bool time_l0 = false;       // a timed iteration also stamps the end of the level-0 forward launch (ev[5])
  double l0_ms = 0.0;         // ... accumulated over the last timed run: the dominant kernel INSIDE an iteration
  double ph_lambda = 0.0;
  std::string err;
#ifdef GPS_TRACE_FUSED
  DevBuf dbg_trace;           // debug builds only: 64 stamps per wave of the last k_fused_level0 launch
  int dbg_trace_waves = 0;
#endif
};

#define HIPCHK(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);                             \
      return GPSLAM_E_HIP;                                                                    \
    }                                                                                         \
  } while (0)

namespace {

int fail(gpslam_hip_handle *h, int code, const char *msg) {
  h->err = msg;
  return code;
}

bool spd_chol_upper(int n, double *A) {
  for (int j = 0; j < n; j++) {
    double dd = A[j * n + j];
    for (int k = 0; k < j; k++) dd -= A[k * n + j] * A[k * n + j];
    if (!(dd > 0.0)) return false;
    dd = std::sqrt(dd);
    A[j * n + j] = dd;
    for (int i = j + 1; i < n; i++) {
      double s = A[j * n + i];
      for (int k = 0; k < j; k++) s -= A[k * n + j] * A[k * n + i];
      A[j * n + i] = s / dd;
    }
    for (int i = 0; i < j; i++) A[j * n + i] = 0.0;
  }
  return true;
}

// U = chol_upper(Qc^-1): invert Qc through its own Cholesky factor, then factor the inverse
bool make_U(int n, const double *Qc, double *U) {
  double C[36], Ci[36], Qi[36];
  std::memcpy(C, Qc, sizeof(double) * n * n);
  if (!spd_chol_upper(n, C)) return false;  // Qc = C^T C
  std::memset(Ci, 0, sizeof(Ci));
  for (int j = 0; j < n; j++) {
    Ci[j * n + j] = 1.0 / C[j * n + j];
    for (int i = j - 1; i >= 0; i--) {
      double s = 0.0;
      for (int k = i + 1; k <= j; k++) s += C[i * n + k] * Ci[k * n + j];
      Ci[i * n + j] = -s / C[i * n + i];
    }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double s = 0.0;
      for (int k = 0; k < n; k++) s += Ci[i * n + k] * Ci[j * n + k];
      Qi[i * n + j] = s;
    }
  std::memcpy(U, Qi, sizeof(double) * n * n);
  return spd_chol_upper(n, U);
}

// First block row of Lambda(tau), Psi(tau) (gpslam/gp/GPutils.h:54-71): [l11 I, l12 I], [p11 I, p12 I].
// Psi = Q(tau) Phi(dt - tau)^T Q^-1(dt) = (A(tau) Phi2(dt - tau)^T Ainv(dt)) (x) (Qc Qc^-1);  Lambda = Phi(tau) - Psi Phi(dt)
void interp_coef(double dt, double tau, double *out4) {
  const double s = dt - tau;
  const double a11 = tau * tau * tau / 3.0 + s * tau * tau / 2.0, a12 = tau * tau / 2.0;   // first row of A(tau) Phi2(s)^T
  const double p11 = a11 * (12.0 / (dt * dt * dt)) + a12 * (-6.0 / (dt * dt));
  const double p12 = a11 * (-6.0 / (dt * dt)) + a12 * (4.0 / dt);
  out4[0] = 1.0 - p11;
  out4[1] = tau - p11 * dt - p12;
  out4[2] = p11;
  out4[3] = p12;
}

// synchronous host -> device copy of a vector (the stream is drained so temporaries may die)
template <typename V> int upload(gpslam_hip_handle *h, DevBuf &buf, const std::vector<V> &v) {
  HIPCHK(buf.reserve(v.size() * sizeof(V)));
  if (!v.empty()) {
    HIPCHK(hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return 0;
}
inline int nblocks(int n, int bs) { return (n + bs - 1) / bs; }

// ---- manifold / block-size dispatch: call f(std::integral_constant<int, X>{}) for the runtime value
template <typename F> void dispatch_mf(int mf, F &&f) {
  switch (mf) {
    case LINEAR2: f(std::integral_constant<int, LINEAR2>{}); break;
    case LINEAR3: f(std::integral_constant<int, LINEAR3>{}); break;
    case POSE2: f(std::integral_constant<int, POSE2>{}); break;
    case POSE3: f(std::integral_constant<int, POSE3>{}); break;
    case ROT3: f(std::integral_constant<int, ROT3>{}); break;
    case ROT3_BIAS: f(std::integral_constant<int, ROT3_BIAS>{}); break;
  }
}
template <typename F> void dispatch_b(int b, F &&f) {
  switch (b) {
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 6: f(std::integral_constant<int, 6>{}); break;
    case 12: f(std::integral_constant<int, 12>{}); break;
  }
}
template <typename F> void dispatch_fk(int fk, F &&f) {
  switch (fk) {
    case 0: f(std::integral_constant<int, 0>{}); break;
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 5: f(std::integral_constant<int, 5>{}); break;
    case 6: f(std::integral_constant<int, 6>{}); break;
    case 7: f(std::integral_constant<int, 7>{}); break;
  }
}

// force_sharded = 1 forces the sharded code path on a single segment (self-test of the exchange plumbing)
bool sharded(const gpslam_hip_handle *h) { return h->cfg.nranks > 1 || h->cfg.force_sharded == 1; }
bool has_right_rank(const gpslam_hip_handle *h) { return sharded(h) && h->cfg.rank < h->cfg.nranks - 1; }

int read_scal(gpslam_hip_handle *h, double *out, int n, int *flag) {
  HIPCHK(hipMemcpyAsync(out, h->scal.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int collect_timing(gpslam_hip_handle *h, double *acc) {
  float ms;
  for (int i = 0; i < 4; i++) {
    HIPCHK(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    acc[i] += ms;
  }
  HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[4]));
  acc[4] += ms;
  if (h->time_l0) {            // the level-0 forward launch of this iteration: its own stamps, or (unfused) ev[2] = start of the solve phase
    if (h->l0_ext) HIPCHK(hipEventElapsedTime(&ms, h->ev_l0a, h->ev_l0b));
    else HIPCHK(hipEventElapsedTime(&ms, h->ev[2], h->ev[5]));
    h->l0_ms += ms;
  }
  return 0;
}

int need_compiled(gpslam_hip_handle *h) {
  if (!h) return GPSLAM_E_INVALID;
  if (!h->compiled) return fail(h, GPSLAM_E_NOT_COMPILED, "call gpslam_hip_compile() first");
  return 0;
}

int add_simple(gpslam_hip_handle *h, SimpleSet &s, int width, int sigw, int32_t count, const int32_t *idx,
               const double *meas, const double *sig, int max_idx) {
  if (!h || count < 0 || (count > 0 && (!idx || !meas || !sig))) return GPSLAM_E_INVALID;
  for (int k = 0; k < count; k++)
    if (idx[k] < 0 || idx[k] > max_idx) return fail(h, GPSLAM_E_INVALID, "factor index out of range");
  for (size_t k = 0; k < (size_t)count * sigw; k++)
    if (!(sig[k] > 0.0)) return fail(h, GPSLAM_E_INVALID, "sigmas must be positive");
  s.width = width;
  s.idx.insert(s.idx.end(), idx, idx + count);
  s.meas.insert(s.meas.end(), meas, meas + (size_t)count * width);
  s.sig.insert(s.sig.end(), sig, sig + (size_t)count * sigw);
  h->compiled = false;
  return 0;
}

// index of the last state a two-state factor may start at (the halo state extends a non-final segment by one)
int max_left(const gpslam_hip_handle *h) { return h->N - 2 + (has_right_rank(h) ? 1 : 0); }

int add_meas(gpslam_hip_handle *h, int fk, int rows, int mw, bool two, bool haslm, bool interp, bool ok_mf,
             int32_t count, const int32_t *idx, const int32_t *lm, const double *meas, const double *sig,
             const double *dt, const double *tau, const double *sensor, const double *calib = nullptr, const double *distortion = nullptr) {
  if (!h || count < 0) return GPSLAM_E_INVALID;
  if (!ok_mf) return fail(h, GPSLAM_E_INVALID, "this factor does not exist for the handle's manifold / landmark dimension");
  if (count > 0 && (!idx || !meas || !sig || (haslm && !lm) || (interp && (!dt || !tau)))) return GPSLAM_E_INVALID;
  MeasSet &s = h->ms[fk];
  const int mx = two ? max_left(h) : h->N - 1;
  for (int k = 0; k < count; k++) {
    if (idx[k] < 0 || idx[k] > mx) return fail(h, GPSLAM_E_INVALID, "factor state index out of range");
    if (haslm && (lm[k] < 0 || lm[k] >= h->L)) return fail(h, GPSLAM_E_INVALID, "landmark index out of range (set_landmarks first)");
    if (interp && !(dt[k] > 0.0)) return fail(h, GPSLAM_E_INVALID, "delta_t must be positive");
  }
  for (size_t k = 0; k < (size_t)count * rows; k++)
    if (!(sig[k] > 0.0)) return fail(h, GPSLAM_E_INVALID, "sigmas must be positive");
  s.kind = fk; s.rows = rows; s.mw = mw; s.two = two; s.haslm = haslm; s.interp = interp;
  s.idx.insert(s.idx.end(), idx, idx + count);
  if (haslm) s.lm.insert(s.lm.end(), lm, lm + count);
  s.meas.insert(s.meas.end(), meas, meas + (size_t)count * mw);
  s.sig.insert(s.sig.end(), sig, sig + (size_t)count * rows);
  if (!s.sqi.empty()) {   // the kind already has Gaussian factors: the newcomers' diagonal models as matrices
    for (int k = 0; k < count; k++)
      for (int r = 0; r < rows; r++)
        for (int q = 0; q < rows; q++) s.sqi.push_back(r == q ? 1.0 / sig[(size_t)k * rows + r] : 0.0);
  }
  if (interp) { s.dt.insert(s.dt.end(), dt, dt + count); s.tau.insert(s.tau.end(), tau, tau + count); }
  {   // this call's body_P_sensor / calibration: find it in (or append it to) the kind's table
    double ent[kMeasAux] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0};
    if (sensor) { std::memcpy(ent, sensor, sizeof(double) * h->pd); ent[17] = 1.0; }
    if (calib) std::memcpy(ent + 12, calib, sizeof(double) * 5);
    if (distortion) std::memcpy(ent + 18, distortion, sizeof(double) * 4);
    int slot = -1;
    const int nent = (int)(s.aux.size() / kMeasAux);
    for (int q = 0; q < nent && slot < 0; q++)
      if (std::memcmp(&s.aux[(size_t)q * kMeasAux], ent, sizeof(ent)) == 0) slot = q;
    if (slot < 0) { slot = nent; s.aux.insert(s.aux.end(), ent, ent + kMeasAux); }
    s.aidx.insert(s.aidx.end(), (size_t)count, slot);
    if (sensor || calib) s.any_aux = true;
  }
  h->compiled = false;
  return 0;
}

int sync_landmarks_to_host(gpslam_hip_handle *h) {
  if (h->L <= 0 || !h->lmk.p) return 0;
  std::vector<double> t((size_t)h->L * h->ld);
  HIPCHK(hipMemcpyAsync(t.data(), h->lmk.p, t.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < t.size(); i++) h->h_lmk[i] = (double)t[i];
  return 0;
}

int backup_state(gpslam_hip_handle *h, bool restore) {
  const size_t np = (size_t)h->pd * h->stride * sizeof(double), nv = (size_t)h->d * h->stride * sizeof(double);
  HIPCHK(h->pose_bak.reserve(np));
  HIPCHK(h->vel_bak.reserve(nv));
  if (restore) {
    HIPCHK(hipMemcpyAsync(h->pose.p, h->pose_bak.p, np, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->vel.p, h->vel_bak.p, nv, hipMemcpyDeviceToDevice, h->stream));
  } else {
    HIPCHK(hipMemcpyAsync(h->pose_bak.p, h->pose.p, np, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->vel_bak.p, h->vel.p, nv, hipMemcpyDeviceToDevice, h->stream));
  }
  if (h->nl > 0) {
    const size_t nb = (size_t)h->nl * sizeof(double);
    HIPCHK(h->lmk_bak.reserve(nb));
    if (restore) HIPCHK(hipMemcpyAsync(h->lmk.p, h->lmk_bak.p, nb, hipMemcpyDeviceToDevice, h->stream));
    else HIPCHK(hipMemcpyAsync(h->lmk_bak.p, h->lmk.p, nb, hipMemcpyDeviceToDevice, h->stream));
  }
  return 0;
}

}  // namespace

// kernels that exist for fp64 only (hand-written 64-bit DPP row layout, v_mfma_f64): the fp32 instantiation of the
// host code never selects them (rows_kernel_applies / compile()), these overloads only keep it compiling
namespace {
// ea / eb != null: the launch carries its own start / stop events (a timed iteration: the dispatch's own time stamps)
#define GPS_FUSED_LAUNCH(...)                                                                                      \
  do {                                                                                                             \
    if (ea) hipExtLaunchKernelGGL((__VA_ARGS__), dim3(grid), dim3(128), 0, st, ea, eb, 0, u);                      \
    else __VA_ARGS__<<<dim3(grid), dim3(128), 0, st>>>(u);                                                         \
  } while (0)
inline void launch_fused_k(int b, const FusedArgs<double, double> &u, int grid, hipStream_t st, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr) {
  if (b == 6) {
    if (u.gps) GPS_FUSED_LAUNCH(k_fused_level0<1, double, 6>);      // d = 3 records (kGp3*)
    else GPS_FUSED_LAUNCH(k_fused_level0<0, double, 6>);
    return;
  }
  if (u.gps && u.rowI) {                     // records + interpolated measurement rows as 16-double lines (round 5)
    if (u.u_diag) GPS_FUSED_LAUNCH(k_fused_level0<4, double, 12, true>);
    else GPS_FUSED_LAUNCH(k_fused_level0<4>);
  } else if (u.gps && u.odd_rows == 2) {     // records + a ring of full-width rows (measurement factors)
    if (u.u_diag) GPS_FUSED_LAUNCH(k_fused_level0<3, double, 12, true>);
    else GPS_FUSED_LAUNCH(k_fused_level0<3>);
  } else if (u.gps && u.odd_rows) GPS_FUSED_LAUNCH(k_fused_level0<2>);
  else if (u.gps && u.u_diag) GPS_FUSED_LAUNCH(k_fused_level0<1, double, 12, true>);   // diagonal chol(Qc^-1)
  else if (u.gps) GPS_FUSED_LAUNCH(k_fused_level0<1>);
  else GPS_FUSED_LAUNCH(k_fused_level0<0>);
}
// fp32 handles: fp32 row tables straight into the fused kernel's fp64 accumulation (round 3: the unfused assembly had cost the
// fp32 mode more than its halved row traffic saved)
inline void launch_fused_k(int b, const FusedArgs<double, float> &u, int grid, hipStream_t st, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr) {
  if (b == 6) GPS_FUSED_LAUNCH(k_fused_level0<0, float, 6>);
  else GPS_FUSED_LAUNCH(k_fused_level0<0, float>);
}
// GPInterpolatedGPSFactorPose3 as 16-double lines: fp64 only (compile(): irow_ok)
inline void launch_gps_lines_k(const MeasArgs<double> &a, int nb, hipStream_t st) {
#if GPS_GPS_LINES_WAVES > 0
  if (a.aidx != nullptr) k_gps_lines<GPS_GPS_LINES_WAVES, true><<<dim3(nb), dim3(128), 0, st>>>(a);   // (round 6: a kernel written for its register count)
  else k_gps_lines<GPS_GPS_LINES_WAVES, false><<<dim3(nb), dim3(128), 0, st>>>(a);
#else
  k_meas<double, POSE3, FK_INTERP_GPS, true, true><<<dim3(nb), dim3(128), 0, st>>>(a);
#endif
}
inline void launch_gps_lines_k(const MeasArgs<float> &, int, hipStream_t) {}
inline void launch_rows_k(int b, const FwdArgs<double> &a, int grid, hipStream_t st) {
  if (b == 12) k_chunk_forward_rows<12><<<dim3(grid), dim3(64), 0, st>>>(a);
  else if (b == 6) k_chunk_forward_rows<6><<<dim3(grid), dim3(64), 0, st>>>(a);
  else k_chunk_forward_rows<4><<<dim3(grid), dim3(64), 0, st>>>(a);
}
inline void launch_rows_k(int, const FwdArgs<float> &, int, hipStream_t) {}
inline void launch_bwd_rows_k(int b, const BwdArgs<double> &a, int grid, hipStream_t st) {
  if (b == 12) k_chunk_backward_rows<12><<<dim3(grid), dim3(64), 0, st>>>(a);
  else if (b == 6) k_chunk_backward_rows<6><<<dim3(grid), dim3(64), 0, st>>>(a);
  else k_chunk_backward_rows<4><<<dim3(grid), dim3(64), 0, st>>>(a);
}
inline void launch_bwd_rows_k(int, const BwdArgs<float> &, int, hipStream_t) {}
// segment interiors of the segmented landmark elimination: planar fp64 chains take the cooperative row-layout kernel (four
// segments per wave), everything else the wave-per-segment kernel
template <int BB, typename T, typename TR> inline void fs_launch_factor(const FsArgs<T, TR> &a, int nseg, hipStream_t st) {
  if constexpr (BB == 6 && std::is_same<T, double>::value && std::is_same<TR, double>::value) {
    k_fs_factor_rows6<0><<<dim3((nseg + 3) / 4), dim3(64), 0, st>>>(a);
    return;
  }
  k_fs_factor<T, BB, TR><<<dim3(nseg), dim3(64), 0, st>>>(a);
}
}  // namespace

