// api.hip -- host side of libgpslam_hip.so: the opaque handle, the graph-compile pass and the C ABI
// declared in include/gpslam_hip.h.  No CPU fallback exists anywhere in this library: every compute entry
// point launches HIP kernels and fails with GPSLAM_E_HIP if the device is unusable.
#include "../../include/gpslam_hip.h"

#include <hip/hip_runtime.h>

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
  int count() const { return (int)idx.size(); }
  void release() { d_idx.release(); d_lm.release(); d_meas.release(); d_sig.release(); d_coef.release(); d_row0.release(); d_aux.release(); d_aidx.release(); d_sqi.release(); }
};

}  // namespace

struct gpslam_hip_handle {
  gpslam_hip_config cfg;
  int mf = 0, d = 0, pd = 0, b = 0, ld = 0;
  int vw = 0;                 // Pose3: velocities are world-frame [v; w] (cfg.reserved[3], the *Pose3VW factors)
  int N = 0, L = 0, stride = 0, R = 1, nl = 0;
  bool own_stream = true;
  hipStream_t stream = nullptr;
  hipStream_t aux_stream = nullptr;   // side stream for the light factor kernels (launch_factors)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
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
  SimpleSet pri, vpri, btw, lpri;
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
  DevBuf gps, gpidx, dU, gsave2;
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

// reserved[0] = 1 forces the sharded code path on a single segment (self-test of the exchange plumbing)
bool sharded(const gpslam_hip_handle *h) { return h->cfg.nranks > 1 || h->cfg.reserved[0] == 1; }
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
  if (h->time_l0) {            // the level-0 forward launch of this iteration (ev[2] = start of the solve phase)
    HIPCHK(hipEventElapsedTime(&ms, h->ev[2], h->ev[5]));
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
             const double *dt, const double *tau, const double *sensor, const double *calib = nullptr) {
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
    double ent[kMeasAux] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0};
    if (sensor) { std::memcpy(ent, sensor, sizeof(double) * h->pd); ent[17] = 1.0; }
    if (calib) std::memcpy(ent + 12, calib, sizeof(double) * 5);
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
inline void launch_fused_k(int b, const FusedArgs<double, double> &u, int grid, hipStream_t st) {
  if (b == 6) { k_fused_level0<0, double, 6><<<dim3(grid), dim3(128), 0, st>>>(u); return; }
  if (u.gps && u.odd_rows) k_fused_level0<2><<<dim3(grid), dim3(128), 0, st>>>(u);
  else if (u.gps) k_fused_level0<1><<<dim3(grid), dim3(128), 0, st>>>(u);
  else k_fused_level0<0><<<dim3(grid), dim3(128), 0, st>>>(u);
}
// fp32 handles: fp32 row tables straight into the fused kernel's fp64 accumulation (round 3: the unfused assembly had cost the
// fp32 mode more than its halved row traffic saved)
inline void launch_fused_k(int b, const FusedArgs<double, float> &u, int grid, hipStream_t st) {
  if (b == 6) k_fused_level0<0, float, 6><<<dim3(grid), dim3(128), 0, st>>>(u);
  else k_fused_level0<0, float><<<dim3(grid), dim3(128), 0, st>>>(u);
}
inline void launch_rows_k(int b, const FwdArgs<double> &a, int grid, hipStream_t st) {
  if (b == 12) k_chunk_forward_rows<12><<<dim3(grid), dim3(64), 0, st>>>(a);
  else if (b == 6) k_chunk_forward_rows<6><<<dim3(grid), dim3(64), 0, st>>>(a);
  else k_chunk_forward_rows<4><<<dim3(grid), dim3(64), 0, st>>>(a);
}
inline void launch_rows_k(int, const FwdArgs<float> &, int, hipStream_t) {}
// segment interiors of the segmented landmark elimination: planar fp64 chains take the cooperative row-layout kernel (four
// segments per wave); GPSLAM_FS_FACTOR_ROWS=0 keeps the wave-per-segment kernel, for A/B measurements
template <int BB, typename T, typename TR> inline void fs_launch_factor(const FsArgs<T, TR> &a, int nseg, hipStream_t st) {
  if constexpr (BB == 6 && std::is_same<T, double>::value && std::is_same<TR, double>::value) {
    static const bool off = getenv("GPSLAM_FS_FACTOR_ROWS") && atoi(getenv("GPSLAM_FS_FACTOR_ROWS")) == 0;
    if (!off) { k_fs_factor_rows6<<<dim3((nseg + 3) / 4), dim3(64), 0, st>>>(a); return; }
  }
  k_fs_factor<T, BB, TR><<<dim3(nseg), dim3(64), 0, st>>>(a);
}
}  // namespace

// =================================================================== the precision-dependent half, once per precision
namespace impl64 {
typedef double Real;
typedef double RowT;
#define IMPL_NS impl64
#include "api_impl.inc"
#undef IMPL_NS
}  // namespace impl64
namespace impl32 {
typedef double Real;
typedef float RowT;
#define IMPL_NS impl32
#include "api_impl.inc"
#undef IMPL_NS
}  // namespace impl32

// =================================================================== C ABI

extern "C" {

void gpslam_hip_default_params(gpslam_hip_params *p) {
  p->max_iterations = 100;
  p->relative_error_tol = 1e-5;
  p->absolute_error_tol = 1e-5;
  p->error_tol = 0.0;
  p->delta_tol = 0.0;
  p->lambda_initial = 1e-5;
  p->lambda_factor = 10.0;
  p->lambda_upper_bound = 1e5;
  p->lambda_lower_bound = 0.0;
  p->min_model_fidelity = 1e-3;
  p->use_lm = 0;
  p->pad = 0;
}

int gpslam_hip_create(const gpslam_hip_config *cfg, gpslam_hip_handle **out) {
  if (!cfg || !out) return GPSLAM_E_INVALID;
  if (cfg->manifold < 0 || cfg->manifold > GPSLAM_ROT3_BIAS) return GPSLAM_E_INVALID;
  if (cfg->precision != GPSLAM_FP64 && cfg->precision != GPSLAM_FP32) return GPSLAM_E_INVALID;
  if (cfg->landmark_dim != 0 && cfg->landmark_dim != 2 && cfg->landmark_dim != 3) return GPSLAM_E_INVALID;
  if (cfg->nranks < 0 || (cfg->nranks > 1 && (cfg->rank < 0 || cfg->rank >= cfg->nranks))) return GPSLAM_E_INVALID;
  if (cfg->reserved[3] != 0 && (cfg->reserved[3] != GPSLAM_VELOCITY_WORLD_VW || cfg->manifold != GPSLAM_POSE3)) return GPSLAM_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GPSLAM_E_HIP;  // no GPU: fail loudly
  if (cfg->device < 0 || cfg->device >= ndev) return GPSLAM_E_INVALID;
  if (hipSetDevice(cfg->device) != hipSuccess) return GPSLAM_E_HIP;
  gpslam_hip_handle *h = new gpslam_hip_handle();
  h->cfg = *cfg;
  if (h->cfg.nranks < 1) h->cfg.nranks = 1;
  h->mf = cfg->manifold;
  static const int dd[6] = {2, 3, 3, 6, 3, 6}, pdd[6] = {2, 3, 3, 12, 9, 12};
  h->d = dd[h->mf];
  h->pd = pdd[h->mf];
  h->b = 2 * h->d;
  h->ld = cfg->landmark_dim;
  h->vw = (cfg->reserved[3] == GPSLAM_VELOCITY_WORLD_VW) ? 1 : 0;
  std::memset(h->Qc, 0, sizeof(h->Qc));
  std::memset(h->U, 0, sizeof(h->U));
  for (int i = 0; i < h->d; i++) h->Qc[i * h->d + i] = h->U[i * h->d + i] = 1.0;
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return GPSLAM_E_HIP; }
  if (hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) { delete h; return GPSLAM_E_HIP; }
  for (int i = 0; i < 6; i++)
    if (hipEventCreate(&h->ev[i]) != hipSuccess) { delete h; return GPSLAM_E_HIP; }
  if (h->scal.reserve(16 * sizeof(double)) != hipSuccess || h->flag.reserve(sizeof(int)) != hipSuccess) {
    delete h;
    return GPSLAM_E_HIP;
  }
  (void)hipMemsetAsync(h->scal.p, 0, 16 * sizeof(double), h->stream);
  (void)hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream);
  (void)hipStreamSynchronize(h->stream);
  *out = h;
  return 0;
}

int gpslam_hip_destroy(gpslam_hip_handle *h) {
  if (!h) return 0;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  DevBuf *bufs[] = {&h->pose, &h->vel, &h->lmk, &h->pose_bak, &h->vel_bak, &h->lmk_bak, &h->d_gp_left, &h->d_gp_dt,
                    &h->d_gp_row0, &h->rowLR, &h->rowE, &h->rowC, &h->rowCE, &h->crowptr, &h->rowM, &h->rowLm, &h->rowptr, &h->partial, &h->lmrow,
                    &h->lmrow_state, &h->lmrow_ptr, &h->lm_t, &h->lm_S, &h->lm_dL, &h->lm_chunk_lm, &h->lm_chunk_j0, &h->lm_chunk_j1, &h->lm_chunk_ptr, &h->lm_part, &h->gsave, &h->dvec,
                    &h->halo_add, &h->iface_send, &h->iface_recv, &h->top_blk, &h->top_x, &h->scal, &h->flag,
                    &h->api_e, &h->api_H, &h->gps, &h->gpidx, &h->dU, &h->gsave2, &h->partial2};
  for (DevBuf *b : bufs) b->release();
  for (SimpleSet *s : {&h->pri, &h->vpri, &h->btw, &h->lpri}) s->release();
  for (MeasSet &s : h->ms) s.release();
  h->fs.release();
  h->lm_gL.release();
  for (Level &v : h->lv) { v.blk.release(); v.add.release(); v.x.release(); }
  for (int i = 0; i < 6; i++) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->stream && h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

const char *gpslam_hip_last_error(const gpslam_hip_handle *h) { return h ? h->err.c_str() : "null handle"; }
void *gpslam_hip_stream(gpslam_hip_handle *h) { return h ? (void *)h->stream : nullptr; }

int gpslam_hip_set_stream(gpslam_hip_handle *h, void *stream) {
  if (!h) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  h->stream = (hipStream_t)stream;
  h->own_stream = false;
  return 0;
}

int gpslam_hip_set_qc(gpslam_hip_handle *h, const double *Qc) {
  if (!h || !Qc) return GPSLAM_E_INVALID;
  double U[36], Qpad[36];
  if (h->mf == ROT3_BIAS) {   // Qc is the 3 x 3 of GaussianProcessPriorRot3; bias and pad components: identity (see GpPrior<ROT3_BIAS>)
    std::memset(Qpad, 0, sizeof(Qpad));
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) Qpad[i * 6 + j] = Qc[i * 3 + j];
      Qpad[(3 + i) * 6 + 3 + i] = 1.0;
    }
    Qc = Qpad;
  }
  if (!make_U(h->d, Qc, U)) return fail(h, GPSLAM_E_NOT_SPD, "Qc is not positive definite");
  std::memcpy(h->Qc, Qc, sizeof(double) * h->d * h->d);
  std::memcpy(h->U, U, sizeof(double) * h->d * h->d);
  h->U_version++;
  return 0;
}

int gpslam_hip_set_states(gpslam_hip_handle *h, int32_t N, const double *pose, const double *vel) {
  if (!h || N <= 0 || !pose || !vel) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  const bool same = (N == h->N) && h->pose.p;
  if (N != h->N) h->compiled = false;
  h->N = N;
  h->stride = N + 1;  // one halo slot: the first state of the right neighbour segment
  std::vector<double> sp((size_t)h->pd * h->stride, 0.0), sv((size_t)h->d * h->stride, 0.0);
  if (same && sharded(h)) {  // keep the halo state
    HIPCHK(hipMemcpyAsync(sp.data(), h->pose.p, sp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(sv.data(), h->vel.p, sv.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  for (int i = 0; i < N; i++) {
    for (int k = 0; k < h->pd; k++) sp[(size_t)k * h->stride + i] = pose[(size_t)i * h->pd + k];
    for (int k = 0; k < h->d; k++) sv[(size_t)k * h->stride + i] = vel[(size_t)i * h->d + k];
  }
  HIPCHK(h->pose.reserve(sp.size() * sizeof(double)));
  HIPCHK(h->vel.reserve(sv.size() * sizeof(double)));
  HIPCHK(hipMemcpyAsync(h->pose.p, sp.data(), sp.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->vel.p, sv.data(), sv.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int gpslam_hip_set_halo_state(gpslam_hip_handle *h, const double *pose, const double *vel) {
  if (!h || h->N <= 0 || !pose || !vel || !h->pose.p) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  std::vector<double> pv(h->pd), vv(h->d);
  for (int k = 0; k < h->pd; k++) pv[k] = pose[k];
  for (int k = 0; k < h->d; k++) vv[k] = vel[k];
  for (int k = 0; k < h->pd; k++)
    HIPCHK(hipMemcpyAsync(h->pose.as<double>() + (size_t)k * h->stride + h->N, &pv[k], sizeof(double), hipMemcpyHostToDevice, h->stream));
  for (int k = 0; k < h->d; k++)
    HIPCHK(hipMemcpyAsync(h->vel.as<double>() + (size_t)k * h->stride + h->N, &vv[k], sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int gpslam_hip_get_states(gpslam_hip_handle *h, double *pose, double *vel) {
  if (!h || h->N <= 0) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  std::vector<double> sp((size_t)h->pd * h->stride), sv((size_t)h->d * h->stride);
  HIPCHK(hipMemcpyAsync(sp.data(), h->pose.p, sp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(sv.data(), h->vel.p, sv.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < h->N; i++) {
    if (pose) for (int k = 0; k < h->pd; k++) pose[(size_t)i * h->pd + k] = (double)sp[(size_t)k * h->stride + i];
    if (vel) for (int k = 0; k < h->d; k++) vel[(size_t)i * h->d + k] = (double)sv[(size_t)k * h->stride + i];
  }
  return 0;
}

int gpslam_hip_set_landmarks(gpslam_hip_handle *h, int32_t L, const double *pts) {
  if (!h || L < 0 || (L > 0 && (!pts || h->ld == 0))) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  if (L != h->L) h->compiled = false;
  h->L = L;
  h->h_lmk.assign(pts, pts + (size_t)L * h->ld);
  return upload(h, h->lmk, h->h_lmk);
}
int gpslam_hip_get_landmarks(gpslam_hip_handle *h, double *pts) {
  if (!h) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  int rc = sync_landmarks_to_host(h);
  if (rc) return rc;
  if (h->L > 0 && pts) std::memcpy(pts, h->h_lmk.data(), sizeof(double) * h->h_lmk.size());
  return 0;
}

int gpslam_hip_add_gp_priors(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt) {
  if (!h || count < 0 || (count > 0 && (!left || !dt))) return GPSLAM_E_INVALID;
  for (int k = 0; k < count; k++) {
    if (left[k] < 0 || left[k] > max_left(h)) return fail(h, GPSLAM_E_INVALID, "gp prior index out of range");
    if (!(dt[k] > 0.0)) return fail(h, GPSLAM_E_INVALID, "gp prior delta_t must be positive");
  }
  h->gp_left.insert(h->gp_left.end(), left, left + count);
  h->gp_dt.insert(h->gp_dt.end(), dt, dt + count);
  h->gp_q.insert(h->gp_q.end(), (size_t)count, 0);
  h->compiled = false;
  return 0;
}
int gpslam_hip_add_gp_priors_qc(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt, const double *Qc) {
  if (!h || count < 0 || (count > 0 && (!left || !dt || !Qc))) return GPSLAM_E_INVALID;
  const int dq = (h->mf == ROT3_BIAS) ? 3 : h->d;
  std::vector<int32_t> q((size_t)count);
  std::vector<double> tab = h->gp_Utab;
  for (int k = 0; k < count; k++) {
    if (left[k] < 0 || left[k] > max_left(h)) return fail(h, GPSLAM_E_INVALID, "gp prior index out of range");
    if (!(dt[k] > 0.0)) return fail(h, GPSLAM_E_INVALID, "gp prior delta_t must be positive");
    double U[36], Qpad[36];
    const double *qc = Qc + (size_t)k * dq * dq;
    if (h->mf == ROT3_BIAS) {   // as in set_qc: bias and pad components are identities
      std::memset(Qpad, 0, sizeof(Qpad));
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Qpad[i * 6 + j] = qc[i * 3 + j];
        Qpad[(3 + i) * 6 + 3 + i] = 1.0;
      }
      qc = Qpad;
    }
    std::memset(U, 0, sizeof(U));
    if (!make_U(h->d, qc, U)) return fail(h, GPSLAM_E_NOT_SPD, "Qc is not positive definite");
    int slot = -1;
    const int nent = (int)(tab.size() / 36);
    for (int e = 0; e < nent && slot < 0; e++)
      if (std::memcmp(&tab[(size_t)e * 36], U, sizeof(U)) == 0) slot = e;
    if (slot < 0) { slot = nent; tab.insert(tab.end(), U, U + 36); }
    q[k] = slot + 1;
  }
  h->gp_Utab.swap(tab);
  h->gp_left.insert(h->gp_left.end(), left, left + count);
  h->gp_dt.insert(h->gp_dt.end(), dt, dt + count);
  h->gp_q.insert(h->gp_q.end(), q.begin(), q.end());
  h->compiled = false;
  return 0;
}
int gpslam_hip_set_meas_covariance(gpslam_hip_handle *h, int32_t kind, int32_t count, const double *cov) {
  if (!h || kind < 0 || kind >= kNumMeasKinds || count < 0 || (count > 0 && !cov)) return GPSLAM_E_INVALID;
  if (kind == FK_AHRS) return fail(h, GPSLAM_E_INVALID, "AHRSFactor takes its covariance in gpslam_hip_add_ahrs");
  MeasSet &s = h->ms[kind];
  if (count > s.count()) return fail(h, GPSLAM_E_INVALID, "set_meas_covariance: more covariances than factors of this kind");
  const int rows = s.rows, n = s.count();
  if (count == 0) return 0;
  if (rows == 1) {   // a 1 x 1 Gaussian is a sigma
    for (int k = 0; k < count; k++) {
      if (!(cov[k] > 0.0)) return fail(h, GPSLAM_E_NOT_SPD, "covariance is not positive definite");
      s.sig[(size_t)(n - count + k)] = std::sqrt(cov[k]);
    }
    h->compiled = false;
    return 0;
  }
  std::vector<double> R((size_t)count * rows * rows, 0.0);
  for (int k = 0; k < count; k++)
    if (!make_U(rows, cov + (size_t)k * rows * rows, &R[(size_t)k * rows * rows])) return fail(h, GPSLAM_E_NOT_SPD, "covariance is not positive definite");
  if (s.sqi.empty()) {
    s.sqi.assign((size_t)n * rows * rows, 0.0);
    for (int f = 0; f < n; f++)
      for (int r = 0; r < rows; r++) s.sqi[((size_t)f * rows + r) * rows + r] = 1.0 / s.sig[(size_t)f * rows + r];
  }
  std::memcpy(&s.sqi[(size_t)(n - count) * rows * rows], R.data(), R.size() * sizeof(double));
  h->compiled = false;
  return 0;
}
int gpslam_hip_add_pose_priors(gpslam_hip_handle *h, int32_t count, const int32_t *idx, const double *prior,
                               const double *sigmas) {
  return h ? add_simple(h, h->pri, h->pd, h->d, count, idx, prior, sigmas, h->N - 1) : GPSLAM_E_INVALID;
}
int gpslam_hip_add_vel_priors(gpslam_hip_handle *h, int32_t count, const int32_t *idx, const double *prior,
                              const double *sigmas) {
  return h ? add_simple(h, h->vpri, h->d, h->d, count, idx, prior, sigmas, h->N - 1) : GPSLAM_E_INVALID;
}
int gpslam_hip_add_between(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *measured,
                           const double *sigmas) {
  return h ? add_simple(h, h->btw, h->pd, h->d, count, left, measured, sigmas, max_left(h)) : GPSLAM_E_INVALID;
}
int gpslam_hip_add_landmark_priors(gpslam_hip_handle *h, int32_t count, const int32_t *idx, const double *prior,
                                   const double *sigmas) {
  if (!h) return GPSLAM_E_INVALID;
  if (h->ld == 0) return fail(h, GPSLAM_E_INVALID, "handle was created without landmarks");
  return add_simple(h, h->lpri, h->ld, h->ld, count, idx, prior, sigmas, h->L - 1);
}
int gpslam_hip_add_interp_range(gpslam_hip_handle *h, int32_t count, const int32_t *left, const int32_t *landmark,
                                const double *z, const double *sigma, const double *dt, const double *tau,
                                const double *sensor) {
  if (!h) return GPSLAM_E_INVALID;
  const bool ok = (h->mf == POSE2 && h->ld == 2) || (h->mf == POSE3 && h->ld == 3) || (h->mf == LINEAR3 && h->ld == 2);
  if (h->mf == LINEAR3 && sensor) return fail(h, GPSLAM_E_INVALID, "GPInterpolatedRangeFactor2DLinear has no body_P_sensor");
  return add_meas(h, FK_INTERP_RANGE, 1, 1, true, true, true, ok, count, left, landmark, z, sigma, dt, tau, sensor);
}
int gpslam_hip_add_range(gpslam_hip_handle *h, int32_t count, const int32_t *idx, const int32_t *landmark,
                         const double *z, const double *sigma) {
  if (!h) return GPSLAM_E_INVALID;
  const bool ok = (h->mf == POSE2 && h->ld == 2) || (h->mf == POSE3 && h->ld == 3) || (h->mf == LINEAR3 && h->ld == 2);
  return add_meas(h, FK_RANGE, 1, 1, false, true, false, ok, count, idx, landmark, z, sigma, nullptr, nullptr, nullptr);
}
int gpslam_hip_add_interp_attitude(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *nZ,
                                   const double *bRef, const double *sigma, const double *dt, const double *tau) {
  if (!h) return GPSLAM_E_INVALID;
  if (count > 0 && (!nZ || !bRef)) return GPSLAM_E_INVALID;
  std::vector<double> m((size_t)std::max(count, 0) * 6);
  for (int k = 0; k < count; k++) {  // Unit3 normalises its argument
    for (int part = 0; part < 2; part++) {
      const double *v = (part == 0 ? nZ : bRef) + 3 * (size_t)k;
      const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      if (!(n > 0.0)) return fail(h, GPSLAM_E_INVALID, "attitude directions must be non-zero");
      for (int q = 0; q < 3; q++) m[6 * (size_t)k + 3 * part + q] = v[q] / n;
    }
  }
  return add_meas(h, FK_INTERP_ATT, 2, 6, true, false, true, h->mf == ROT3 || h->mf == ROT3_BIAS, count, left, nullptr, m.data(), sigma, dt, tau, nullptr);
}
int gpslam_hip_add_ahrs(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *delta_R,
                        const double *dR_dbias, const double *bias_hat, const double *delta_tij, const double *cov,
                        const double *omega_coriolis) {
  if (!h) return GPSLAM_E_INVALID;
  if (count > 0 && (!delta_R || !dR_dbias || !bias_hat || !delta_tij || !cov)) return GPSLAM_E_INVALID;
  std::vector<double> m((size_t)std::max(count, 0) * kAhrsWidth, 0.0), ones((size_t)std::max(count, 0) * 3, 1.0);
  for (int k = 0; k < count; k++) {
    double *o = &m[(size_t)k * kAhrsWidth];
    std::memcpy(o, delta_R + 9 * (size_t)k, 9 * sizeof(double));
    std::memcpy(o + 9, dR_dbias + 9 * (size_t)k, 9 * sizeof(double));
    std::memcpy(o + 18, bias_hat + 3 * (size_t)k, 3 * sizeof(double));
    o[21] = delta_tij[k];
    if (omega_coriolis) std::memcpy(o + 22, omega_coriolis, 3 * sizeof(double));
    // noiseModel::Gaussian::Covariance(preintMeasCov): whitening by R, R^T R = cov^-1
    if (!make_U(3, cov + 9 * (size_t)k, o + 25)) return fail(h, GPSLAM_E_NOT_SPD, "AHRS pre-integrated covariance is not positive definite");
  }
  return add_meas(h, FK_AHRS, 3, kAhrsWidth, true, false, false, h->mf == ROT3_BIAS, count, left, nullptr, m.data(), ones.data(),
                  nullptr, nullptr, nullptr);
}
int gpslam_hip_add_interp_gps(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *measured,
                              const double *sigmas, const double *dt, const double *tau, const double *sensor) {
  if (!h) return GPSLAM_E_INVALID;
  return add_meas(h, FK_INTERP_GPS, 3, 3, true, false, true, h->mf == POSE3, count, left, nullptr, measured, sigmas, dt, tau, sensor);
}
int gpslam_hip_add_interp_projection(gpslam_hip_handle *h, int32_t count, const int32_t *left, const int32_t *landmark,
                                     const double *measured, const double *sigmas, const double *dt, const double *tau,
                                     const double *K, const double *sensor) {
  if (!h || !K) return GPSLAM_E_INVALID;
  if (h->vw) return fail(h, GPSLAM_E_UNSUPPORTED, "the reference has no projection factor for the VW velocity family");
  return add_meas(h, FK_INTERP_PROJ, 2, 2, true, true, true, h->mf == POSE3 && h->ld == 3, count, left, landmark, measured, sigmas, dt, tau, sensor, K);
}
int gpslam_hip_add_odometry2d(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *measured,
                              const double *sigmas) {
  if (!h) return GPSLAM_E_INVALID;
  return add_meas(h, FK_ODOM2D, 3, 3, true, false, false, h->mf == LINEAR3, count, left, nullptr, measured, sigmas, nullptr, nullptr, nullptr);
}
int gpslam_hip_add_bearing_range(gpslam_hip_handle *h, int32_t count, const int32_t *idx, const int32_t *landmark,
                                 const double *bearing, const double *range, const double *sigmas) {
  if (!h) return GPSLAM_E_INVALID;
  if (count > 0 && (!bearing || !range)) return GPSLAM_E_INVALID;
  std::vector<double> m((size_t)std::max(count, 0) * 2);
  for (int k = 0; k < count; k++) { m[2 * (size_t)k] = bearing[k]; m[2 * (size_t)k + 1] = range[k]; }
  return add_meas(h, FK_BEARING_RANGE, 2, 2, false, true, false, h->mf == LINEAR3 && h->ld == 2, count, idx, landmark, m.data(), sigmas, nullptr, nullptr, nullptr);
}

int gpslam_hip_clear_factors(gpslam_hip_handle *h) {
  if (!h) return GPSLAM_E_INVALID;
  h->gp_left.clear(); h->gp_dt.clear(); h->gp_q.clear(); h->gp_Utab.clear(); h->gp_perm.clear(); h->gp_groups.clear();
  for (SimpleSet *s : {&h->pri, &h->vpri, &h->btw, &h->lpri}) { s->idx.clear(); s->meas.clear(); s->sig.clear(); }
  for (MeasSet &s : h->ms) {
    s.idx.clear(); s.lm.clear(); s.meas.clear(); s.sig.clear(); s.dt.clear(); s.tau.clear(); s.aux.clear(); s.aidx.clear(); s.sqi.clear();
    s.any_aux = false;
  }
  h->compiled = false;
  return 0;
}

int gpslam_hip_plan_info(gpslam_hip_handle *h, int32_t out8[8]) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!out8) return GPSLAM_E_INVALID;
  const bool fused = h->fuse_ok && h->lv.size() >= 2;
  const int32_t v[8] = {(int32_t)h->lv.size(), h->lv.empty() ? 0 : h->lv[0].m, h->lv.size() > 1 ? h->lv[1].m : 0, fused ? 1 : 0,
                        (fused && h->struct_ok) ? 1 : 0, h->M, h->Mc, h->R};
  for (int i = 0; i < 8; i++) out8[i] = v[i];
  return 0;
}

int gpslam_hip_segment_plan(gpslam_hip_handle *h, int32_t out8[8]) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!out8) return GPSLAM_E_INVALID;
  const FatSepPlan &p = h->fs;
  const int32_t v[8] = {p.active ? 1 : 0, p.C, p.K, p.NB, p.NC, p.NCP, (int32_t)p.levels.size(), p.nlinks};
  for (int i = 0; i < 8; i++) out8[i] = p.active ? v[i] : 0;
  return 0;
}

int gpslam_hip_last_level0_ms(gpslam_hip_handle *h, double *ms) {
  if (!h || !ms) return GPSLAM_E_INVALID;
  *ms = h->l0_ms;
  return 0;
}
int gpslam_hip_last_timing(gpslam_hip_handle *h, double *out5) {
  if (!h || !out5) return GPSLAM_E_INVALID;
  for (int i = 0; i < 5; i++) out5[i] = h->last_ms[i];
  return 0;
}

// interpolatePose (+ H1..H4 when out_H != NULL) of the current estimate; shared by the two entry points below
static int interpolate_impl(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt, const double *tau,
                            double *out_pose, double *out_H) {
  if (!h || count < 0 || (count > 0 && (!left || !dt || !tau || !out_pose))) return GPSLAM_E_INVALID;
  if (h->N < 2) return fail(h, GPSLAM_E_INVALID, "interpolation needs at least two states");
  const int mx = max_left(h);
  for (int q = 0; q < count; q++) {
    if (left[q] < 0 || left[q] > mx) return fail(h, GPSLAM_E_INVALID, "query interval out of range");
    if (!(dt[q] > 0.0)) return fail(h, GPSLAM_E_INVALID, "delta_t must be positive");
  }
  if (count == 0) return 0;
  (void)hipSetDevice(h->cfg.device);
  const int pd = h->pd, d = h->d;
  std::vector<double> coef((size_t)count * 4);
  for (int q = 0; q < count; q++) interp_coef(dt[q], tau[q], &coef[4 * (size_t)q]);
  std::vector<int> li(left, left + count);
  struct Scratch {   // query buffers live for this call only
    DevBuf left, coef, out, outH;
    ~Scratch() { left.release(); coef.release(); out.release(); outH.release(); }
  } sc;
  int rc;
  if ((rc = upload(h, sc.left, li))) return rc;
  if ((rc = upload(h, sc.coef, coef))) return rc;
  HIPCHK(sc.out.reserve((size_t)count * pd * sizeof(double)));
  if (out_H) HIPCHK(sc.outH.reserve((size_t)count * 4 * d * d * sizeof(double)));
  QueryArgs<double> a;
  a.pose = h->pose.as<double>(); a.vel = h->vel.as<double>(); a.stride = h->stride; a.count = count;
  a.left = sc.left.as<int>(); a.coef = sc.coef.as<double>(); a.out = sc.out.as<double>(); a.out_H = out_H ? sc.outH.as<double>() : nullptr;
  a.vw = h->vw;
  dispatch_mf(h->mf, [&](auto tag) {
    constexpr int MF = decltype(tag)::value;
    if (out_H) k_interp_query<double, MF, true><<<dim3(nblocks(count, 128)), dim3(128), 0, h->stream>>>(a);
    else k_interp_query<double, MF, false><<<dim3(nblocks(count, 128)), dim3(128), 0, h->stream>>>(a);
  });
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out_pose, sc.out.p, (size_t)count * pd * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out_H) HIPCHK(hipMemcpyAsync(out_H, sc.outH.p, (size_t)count * 4 * d * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
int gpslam_hip_interpolate_poses(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt,
                                 const double *tau, double *out_pose) {
  return interpolate_impl(h, count, left, dt, tau, out_pose, nullptr);
}
int gpslam_hip_interpolate_poses_jac(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt,
                                     const double *tau, double *out_pose, double *out_H) {
  if (!out_H) return GPSLAM_E_INVALID;
  return interpolate_impl(h, count, left, dt, tau, out_pose, out_H);
}

// GaussianProcessInterpolatorLinear<D>::interpolateVelocity (gpslam.h:193, GaussianProcessInterpolatorLinear.h:106-126) of the
// current estimate, batched.  Bottom block rows of Lambda(tau) / Psi(tau): l21 = -p21, l22 = 1 - p21 dt - p22 with
// (p21, p22) = second row of A(tau) Phi2(dt - tau)^T Ainv(dt) (Qc cancels, as in interp_coef).
int gpslam_hip_interpolate_velocities(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt,
                                      const double *tau, double *out_vel, double *out_H) {
  if (!h || count < 0 || (count > 0 && (!left || !dt || !tau || !out_vel))) return GPSLAM_E_INVALID;
  if (h->mf != LINEAR2 && h->mf != LINEAR3)
    return fail(h, GPSLAM_E_UNSUPPORTED, "interpolateVelocity exists for GaussianProcessInterpolatorLinear only: the reference declares it for the "
                                         "Lie-group interpolators without implementing it (GaussianProcessInterpolatorPose3.h:118-123)");
  if (h->N < 2) return fail(h, GPSLAM_E_INVALID, "interpolation needs at least two states");
  const int mx = max_left(h);
  for (int q = 0; q < count; q++) {
    if (left[q] < 0 || left[q] > mx) return fail(h, GPSLAM_E_INVALID, "query interval out of range");
    if (!(dt[q] > 0.0)) return fail(h, GPSLAM_E_INVALID, "delta_t must be positive");
  }
  if (count == 0) return 0;
  (void)hipSetDevice(h->cfg.device);
  const int d = h->d;
  std::vector<double> coef((size_t)count * 4);
  for (int q = 0; q < count; q++) {
    const double T = dt[q], t = tau[q], sft = T - t;
    const double a21 = t * t / 2.0 + t * sft, a22 = t;            // second row of A(tau) Phi2(dt - tau)^T
    const double p21 = a21 * (12.0 / (T * T * T)) + a22 * (-6.0 / (T * T));
    const double p22 = a21 * (-6.0 / (T * T)) + a22 * (4.0 / T);
    coef[4 * (size_t)q] = -p21;
    coef[4 * (size_t)q + 1] = 1.0 - p21 * T - p22;
    coef[4 * (size_t)q + 2] = p21;
    coef[4 * (size_t)q + 3] = p22;
  }
  std::vector<int> li(left, left + count);
  struct Scratch {
    DevBuf left, coef, out, outH;
    ~Scratch() { left.release(); coef.release(); out.release(); outH.release(); }
  } sc;
  int rc;
  if ((rc = upload(h, sc.left, li))) return rc;
  if ((rc = upload(h, sc.coef, coef))) return rc;
  HIPCHK(sc.out.reserve((size_t)count * d * sizeof(double)));
  if (out_H) HIPCHK(sc.outH.reserve((size_t)count * 4 * d * d * sizeof(double)));
  QueryArgs<double> a;
  a.pose = h->pose.as<double>(); a.vel = h->vel.as<double>(); a.stride = h->stride; a.count = count;
  a.left = sc.left.as<int>(); a.coef = sc.coef.as<double>(); a.out = sc.out.as<double>(); a.out_H = out_H ? sc.outH.as<double>() : nullptr;
  a.vw = 0;
  if (h->mf == LINEAR2) k_interp_velocity<double, LINEAR2><<<dim3(nblocks(count, 128)), dim3(128), 0, h->stream>>>(a);
  else k_interp_velocity<double, LINEAR3><<<dim3(nblocks(count, 128)), dim3(128), 0, h->stream>>>(a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out_vel, sc.out.p, (size_t)count * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out_H) HIPCHK(hipMemcpyAsync(out_H, sc.outH.p, (size_t)count * 4 * d * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// getBodyCentricVb / getBodyCentricVs (gpslam.h:161-164, gpslam/gp/Pose3utils.cpp:17-24), batched over pose pairs; any handle
// (its device and stream are used, its graph is not touched)
int gpslam_hip_body_centric_velocity(gpslam_hip_handle *h, int32_t which, int32_t count, const double *pose1, const double *pose2,
                                     const double *dt, double *out) {
  if (!h || count < 0 || (which != 0 && which != 1) || (count > 0 && (!pose1 || !pose2 || !dt || !out))) return GPSLAM_E_INVALID;
  for (int q = 0; q < count; q++)
    if (!(dt[q] != 0.0)) return fail(h, GPSLAM_E_INVALID, "delta_t must not be zero");
  if (count == 0) return 0;
  (void)hipSetDevice(h->cfg.device);
  struct Scratch {
    DevBuf a, b, t, o;
    ~Scratch() { a.release(); b.release(); t.release(); o.release(); }
  } sc;
  const size_t np = (size_t)count * 12 * sizeof(double);
  HIPCHK(sc.a.reserve(np)); HIPCHK(sc.b.reserve(np));
  HIPCHK(sc.t.reserve((size_t)count * sizeof(double))); HIPCHK(sc.o.reserve((size_t)count * 6 * sizeof(double)));
  HIPCHK(hipMemcpyAsync(sc.a.p, pose1, np, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(sc.b.p, pose2, np, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(sc.t.p, dt, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream));
  k_body_velocity<double><<<dim3(nblocks(count, 128)), dim3(128), 0, h->stream>>>(sc.a.as<double>(), sc.b.as<double>(), sc.t.as<double>(), count, which, sc.o.as<double>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, sc.o.p, (size_t)count * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// ---- precision dispatch: the handle was created GPSLAM_FP64 or GPSLAM_FP32
int gpslam_hip_compile(gpslam_hip_handle *h) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_compile(h) : impl64::gpslam_hip_compile(h);
}
int gpslam_hip_linearize_gp(gpslam_hip_handle *h, double *errors, double *jacobians) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_linearize_gp(h, errors, jacobians) : impl64::gpslam_hip_linearize_gp(h, errors, jacobians);
}
int gpslam_hip_linearize_meas(gpslam_hip_handle *h, int32_t kind, double *errors, double *jacobians) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_linearize_meas(h, kind, errors, jacobians) : impl64::gpslam_hip_linearize_meas(h, kind, errors, jacobians);
}
int gpslam_hip_error(gpslam_hip_handle *h, double *err) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_error(h, err) : impl64::gpslam_hip_error(h, err);
}
int gpslam_hip_iterate_gn(gpslam_hip_handle *h, gpslam_hip_stats *st) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_iterate_gn(h, st) : impl64::gpslam_hip_iterate_gn(h, st);
}
int gpslam_hip_run_gn(gpslam_hip_handle *h, int32_t iters, gpslam_hip_stats *st, double *out5) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_run_gn(h, iters, st, out5) : impl64::gpslam_hip_run_gn(h, iters, st, out5);
}
int gpslam_hip_iterate_lm(gpslam_hip_handle *h, double *lambda, const gpslam_hip_params *p, gpslam_hip_stats *st) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_iterate_lm(h, lambda, p, st) : impl64::gpslam_hip_iterate_lm(h, lambda, p, st);
}
int gpslam_hip_optimize(gpslam_hip_handle *h, const gpslam_hip_params *p, gpslam_hip_stats *st) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_optimize(h, p, st) : impl64::gpslam_hip_optimize(h, p, st);
}
int gpslam_hip_normal_equations(gpslam_hip_handle *h, double *D, double *O, double *g, double *B) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_normal_equations(h, D, O, g, B) : impl64::gpslam_hip_normal_equations(h, D, O, g, B);
}
int gpslam_hip_get_rows(gpslam_hip_handle *h, int32_t *n_rows, double *rowLR, double *rowE, double *rowM, int32_t *rowLm) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_get_rows(h, n_rows, rowLR, rowE, rowM, rowLm) : impl64::gpslam_hip_get_rows(h, n_rows, rowLR, rowE, rowM, rowLm);
}
int gpslam_hip_block_tridiag_solve(gpslam_hip_handle *h, int32_t N, const double *D, const double *O, const double *g, double *x) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_block_tridiag_solve(h, N, D, O, g, x) : impl64::gpslam_hip_block_tridiag_solve(h, N, D, O, g, x);
}
int gpslam_hip_time_kernel(gpslam_hip_handle *h, int32_t which, int32_t reps, double *avg_ms) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_time_kernel(h, which, reps, avg_ms) : impl64::gpslam_hip_time_kernel(h, which, reps, avg_ms);
}
int gpslam_hip_interface_send(gpslam_hip_handle *h, void **dev_ptr, size_t *bytes) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_interface_send(h, dev_ptr, bytes) : impl64::gpslam_hip_interface_send(h, dev_ptr, bytes);
}
int gpslam_hip_interface_recv(gpslam_hip_handle *h, void **dev_ptr, size_t *bytes) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_interface_recv(h, dev_ptr, bytes) : impl64::gpslam_hip_interface_recv(h, dev_ptr, bytes);
}
int gpslam_hip_iterate_phase1(gpslam_hip_handle *h, double lambda) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_iterate_phase1(h, lambda) : impl64::gpslam_hip_iterate_phase1(h, lambda);
}
int gpslam_hip_iterate_phase2(gpslam_hip_handle *h, gpslam_hip_stats *st) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_iterate_phase2(h, st) : impl64::gpslam_hip_iterate_phase2(h, st);
}
int gpslam_hip_iterate_phase2a(gpslam_hip_handle *h) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_iterate_phase2a(h) : impl64::gpslam_hip_iterate_phase2a(h);
}
int gpslam_hip_iterate_phase2b(gpslam_hip_handle *h, gpslam_hip_stats *st) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_iterate_phase2b(h, st) : impl64::gpslam_hip_iterate_phase2b(h, st);
}
int gpslam_hip_landmark_reduce_buffer(gpslam_hip_handle *h, void **dev_ptr, size_t *bytes) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_landmark_reduce_buffer(h, dev_ptr, bytes) : impl64::gpslam_hip_landmark_reduce_buffer(h, dev_ptr, bytes);
}
int gpslam_hip_fs_set_split(gpslam_hip_handle *h, int32_t rank, int32_t nranks, const int32_t *first_lm, int32_t n_first,
                            const int32_t *last_lm, int32_t n_last) {
  if (!h) return GPSLAM_E_INVALID;
  if (nranks < 1 || rank < 0 || rank >= nranks || n_first < 0 || n_last < 0 || (n_first && !first_lm) || (n_last && !last_lm))
    return fail(h, GPSLAM_E_INVALID, "fs_set_split: bad rank / nranks / landmark lists");
  if (sharded(h)) return fail(h, GPSLAM_E_INVALID, "fs_set_split: create the handle with nranks = 1 (the pieces overlap in their shared cut states, there is no halo)");
  if ((rank == 0 && n_first) || (rank == nranks - 1 && n_last)) return fail(h, GPSLAM_E_INVALID, "fs_set_split: the two ends of the whole chain share nothing");
  {   // a landmark listed twice, or in both lists, would corrupt the slot bookkeeping of the shared end blocks
    std::vector<int32_t> all(first_lm, first_lm + n_first);
    all.insert(all.end(), last_lm, last_lm + n_last);
    std::sort(all.begin(), all.end());
    if (std::adjacent_find(all.begin(), all.end()) != all.end())
      return fail(h, GPSLAM_E_INVALID, "fs_set_split: a landmark is listed twice (the first / last lists must be duplicate-free and disjoint)");
    if (!all.empty() && all.front() < 0) return fail(h, GPSLAM_E_INVALID, "fs_set_split: negative landmark index");
  }
  h->fs.split = true;
  h->fs.rank = rank; h->fs.nranks = nranks; h->fs.nb_top = 0;
  h->fs.first_lm.assign(first_lm, first_lm + n_first);
  h->fs.last_lm.assign(last_lm, last_lm + n_last);
  h->compiled = false;
  return 0;
}
int gpslam_hip_fs_split_info(gpslam_hip_handle *h, int32_t out4[4]) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_fs_split_info(h, out4) : impl64::gpslam_hip_fs_split_info(h, out4);
}
int gpslam_hip_fs_set_top(gpslam_hip_handle *h, int32_t nb_top) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_fs_set_top(h, nb_top) : impl64::gpslam_hip_fs_set_top(h, nb_top);
}
int gpslam_hip_fs_interface(gpslam_hip_handle *h, void **send, size_t *send_bytes, void **recv, size_t *recv_bytes) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_fs_interface(h, send, send_bytes, recv, recv_bytes)
                                         : impl64::gpslam_hip_fs_interface(h, send, send_bytes, recv, recv_bytes);
}
int gpslam_hip_fs_phase1(gpslam_hip_handle *h, double lambda) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_fs_phase1(h, lambda) : impl64::gpslam_hip_fs_phase1(h, lambda);
}
int gpslam_hip_fs_phase2(gpslam_hip_handle *h, gpslam_hip_stats *st) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_fs_phase2(h, st) : impl64::gpslam_hip_fs_phase2(h, st);
}
int gpslam_hip_fs_lm_trial_phase1(gpslam_hip_handle *h, double lambda) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_fs_lm_trial_phase1(h, lambda) : impl64::gpslam_hip_fs_lm_trial_phase1(h, lambda);
}
int gpslam_hip_fs_lm_trial_phase2(gpslam_hip_handle *h, double *out6) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_fs_lm_trial_phase2(h, out6) : impl64::gpslam_hip_fs_lm_trial_phase2(h, out6);
}
int gpslam_hip_lm_begin(gpslam_hip_handle *h) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_lm_begin(h) : impl64::gpslam_hip_lm_begin(h);
}
int gpslam_hip_lm_trial_phase1(gpslam_hip_handle *h, double lambda) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_lm_trial_phase1(h, lambda) : impl64::gpslam_hip_lm_trial_phase1(h, lambda);
}
int gpslam_hip_lm_trial_phase2(gpslam_hip_handle *h, double *out6) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_lm_trial_phase2(h, out6) : impl64::gpslam_hip_lm_trial_phase2(h, out6);
}
int gpslam_hip_lm_reject(gpslam_hip_handle *h) {
  if (!h) return GPSLAM_E_INVALID;
  return h->cfg.precision == GPSLAM_FP32 ? impl32::gpslam_hip_lm_reject(h) : impl64::gpslam_hip_lm_reject(h);
}

}  // extern "C"
