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

using namespace gps;
typedef double Real;  // GPSLAM_FP64; the kernels are templated on the scalar for the fp32 path

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
  DevBuf d_idx, d_lm, d_meas, d_sig, d_coef, d_row0, d_aux, d_aidx;
  int count() const { return (int)idx.size(); }
  void release() { d_idx.release(); d_lm.release(); d_meas.release(); d_sig.release(); d_coef.release(); d_row0.release(); d_aux.release(); d_aidx.release(); }
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
  bool fuse_ok = false;     // k_fused_level0 applies to this graph (compile())
  bool fuse_now = false;    // ... and the iteration being enqueued uses it (enqueue_gn)
  bool compiled = false;
  double last_ms[5] = {0, 0, 0, 0, 0};
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
int upload_real(gpslam_hip_handle *h, DevBuf &buf, const std::vector<double> &v) {
  std::vector<Real> t(v.begin(), v.end());
  return upload(h, buf, t);
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
  }
}

UMat<Real> make_umat(const gpslam_hip_handle *h) {
  UMat<Real> u;
  for (int i = 0; i < 36; i++) u.u[i] = 0;
  for (int i = 0; i < h->d * h->d; i++) u.u[i] = (Real)h->U[i];
  return u;
}

// reserved[0] = 1 forces the sharded code path on a single segment (self-test of the exchange plumbing)
bool sharded(const gpslam_hip_handle *h) { return h->cfg.nranks > 1 || h->cfg.reserved[0] == 1; }
bool has_right_rank(const gpslam_hip_handle *h) { return sharded(h) && h->cfg.rank < h->cfg.nranks - 1; }

GpArgs<Real> gp_args(gpslam_hip_handle *h, Real *partial) {
  GpArgs<Real> a;
  a.pose = h->pose.as<Real>(); a.vel = h->vel.as<Real>(); a.stride = h->stride;
  a.count = (int)h->gp_left.size();
  a.left = h->d_gp_left.as<int>(); a.dt = h->d_gp_dt.as<Real>(); a.row0 = h->d_gp_row0.as<int>();
  a.rowLR = h->rowLR.as<Real>(); a.rowE = h->rowE.as<Real>();
  a.partial = partial; a.out_e = nullptr; a.out_H = nullptr;
  a.U = make_umat(h);
  a.vw = h->vw;
  return a;
}

LmArgs<Real> lm_args(gpslam_hip_handle *h, double lambda) {
  LmArgs<Real> a;
  a.N = h->N; a.R = h->R; a.B = h->b; a.L = h->L; a.ld = h->ld; a.nl = h->nl;
  a.Nx = h->N + (has_right_rank(h) ? 1 : 0);
  a.rowLR = h->rowLR.as<Real>(); a.rowE = h->rowE.as<Real>(); a.rowM = h->rowM.as<Real>();
  a.lmrow = h->lmrow.as<int>(); a.lmrow_state = h->lmrow_state.as<int>(); a.lmrow_ptr = h->lmrow_ptr.as<int>();
  a.nlmrows = h->nlmrows;
  a.chunk_lm = h->lm_chunk_lm.as<int>(); a.chunk_j0 = h->lm_chunk_j0.as<int>(); a.chunk_j1 = h->lm_chunk_j1.as<int>();
  a.chunk_ptr = h->lm_chunk_ptr.as<int>(); a.nchunks = h->nlmchunks; a.part = h->lm_part.as<Real>();
  a.x = h->lv.empty() ? nullptr : h->lv[0].x.as<Real>();
  a.t = h->lm_t.as<Real>();
  a.npri = h->lpri.count(); a.pri_lm = h->lpri.d_idx.as<int>(); a.pri_meas = h->lpri.d_meas.as<Real>();
  a.pri_sig = h->lpri.d_sig.as<Real>();
  a.lmk = h->lmk.as<Real>(); a.S = h->lm_S.as<Real>(); a.gL = h->lm_S.as<Real>() + (size_t)h->nl * h->R; a.dL = h->lm_dL.as<Real>();
  a.lambda = (Real)lambda; a.flag = h->flag.as<int>(); a.partial = nullptr;
  return a;
}

// mode 0: Jacobian rows + error, 1: error only.  Error partial sums land in h->partial, reduced into scal[slot].
int launch_factors(gpslam_hip_handle *h, int mode, int slot) {
  Real *part = h->partial.as<Real>();
  int off = 0;
  // The factor kernels are independent of one another (disjoint rows, disjoint partial sums).  The GP-prior kernel
  // runs one register-heavy wave per SIMD and leaves most issue slots empty, the others are light streaming kernels:
  // they go to a second stream and run underneath it (fork / join with events; joined before the reduction).
  const bool fork = (mode == 0) && !h->gp_left.empty() && h->aux_stream != nullptr;
  hipStream_t side = fork ? h->aux_stream : h->stream;
  if (fork) {
    HIPCHK(hipEventRecord(h->ev_fork, h->stream));
    HIPCHK(hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0));
  }
  if (!h->gp_left.empty()) {
    GpArgs<Real> a = gp_args(h, part + off);
    const int nb = nblocks(a.count, 128);
    dispatch_mf(h->mf, [&](auto tag) {
      constexpr int MF = decltype(tag)::value;
      if (mode == 0) {
        if (MF == POSE3 && h->vw) k_gp<Real, MF, 0, true><<<dim3(nb), dim3(128), 0, h->stream>>>(a);
        else k_gp<Real, MF, 0><<<dim3(nb), dim3(128), 0, h->stream>>>(a);
      } else {
        k_gp<Real, MF, 1><<<dim3(nb), dim3(128), 0, h->stream>>>(a);
      }
    });
    off += nb;
  }
  SimpleSet *sets[3] = {&h->pri, &h->vpri, &h->btw};
  for (int kind = 0; kind < 3; kind++) {
    SimpleSet &s = *sets[kind];
    if (s.count() == 0) continue;
    FacArgs<Real> a;
    a.pose = h->pose.as<Real>(); a.vel = h->vel.as<Real>(); a.stride = h->stride;
    a.count = s.count(); a.chart = h->cfg.chart;
    a.idx = s.d_idx.as<int>(); a.meas = s.d_meas.as<Real>(); a.sig = s.d_sig.as<Real>(); a.row0 = s.d_row0.as<int>();
    const bool compact = (kind != 1);   // pose priors and between factors: velocity-free rows
    a.rowLR = compact ? h->rowC.as<Real>() : h->rowLR.as<Real>(); a.rowE = compact ? h->rowCE.as<Real>() : h->rowE.as<Real>();
    a.partial = part + off;
    const int nb = nblocks(a.count, 128);
    dispatch_mf(h->mf, [&](auto tag) {
      constexpr int MF = decltype(tag)::value;
      const dim3 g(nb), t(128);
      if (mode == 0) {
        if (kind == 0) k_simple<Real, MF, 0, true><<<g, t, 0, side>>>(a);
        else if (kind == 1) k_simple<Real, MF, 1, true><<<g, t, 0, side>>>(a);
        else k_simple<Real, MF, 2, true><<<g, t, 0, side>>>(a);
      } else {
        if (kind == 0) k_simple<Real, MF, 0, false><<<g, t, 0, side>>>(a);
        else if (kind == 1) k_simple<Real, MF, 1, false><<<g, t, 0, side>>>(a);
        else k_simple<Real, MF, 2, false><<<g, t, 0, side>>>(a);
      }
    });
    off += nb;
  }
  for (int fk = 0; fk < kNumMeasKinds; fk++) {
    MeasSet &s = h->ms[fk];
    if (s.count() == 0) continue;
    MeasArgs<Real> a;
    a.pose = h->pose.as<Real>(); a.vel = h->vel.as<Real>(); a.stride = h->stride;
    a.lmk = h->lmk.as<Real>(); a.ld = h->ld; a.count = s.count(); a.chart = h->cfg.chart;
    a.idx = s.d_idx.as<int>(); a.lm = s.d_lm.as<int>(); a.meas = s.d_meas.as<Real>(); a.mw = s.mw;
    a.sig = s.d_sig.as<Real>(); a.coef = s.d_coef.as<Real>();
    a.aux = s.any_aux ? s.d_aux.as<Real>() : nullptr; a.aidx = s.any_aux ? s.d_aidx.as<int>() : nullptr;
    a.vw = h->vw;
    a.row0 = s.d_row0.as<int>();
    a.rowLR = h->rowLR.as<Real>(); a.rowE = h->rowE.as<Real>(); a.rowM = h->rowM.as<Real>(); a.rowLm = h->rowLm.as<int>();
    a.partial = part + off;
    const int nb = nblocks(a.count, 128);
    dispatch_mf(h->mf, [&](auto tag) {
      constexpr int MF = decltype(tag)::value;
      dispatch_fk(fk, [&](auto ftag) {
        constexpr int FK = decltype(ftag)::value;
        if (mode == 0) k_meas<Real, MF, FK, true><<<dim3(nb), dim3(128), 0, side>>>(a);
        else k_meas<Real, MF, FK, false><<<dim3(nb), dim3(128), 0, side>>>(a);
      });
    });
    off += nb;
  }
  if (h->lpri.count() > 0 && h->fs.active) {   // tens of thousands of landmark priors: grid-stride, one partial per block
    const int nbl = std::min(nblocks(h->lpri.count(), 256), 256);
    k_fs_lmprior_err<Real><<<dim3(nbl), dim3(256), 0, side>>>(h->lmk.as<Real>(), h->lpri.d_idx.as<int>(), h->lpri.d_meas.as<Real>(),
                                                               h->lpri.d_sig.as<Real>(), h->lpri.count(), h->ld, part + off);
    off += nbl;
  } else if (h->lpri.count() > 0) {
    LmArgs<Real> a = lm_args(h, 0.0);
    a.partial = part + off;
    k_lmprior_err<Real><<<dim3(1), dim3(128), 0, side>>>(a);
    off += 1;
  }
  if (fork) {
    HIPCHK(hipEventRecord(h->ev_join, h->aux_stream));
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  }
  k_final_reduce<Real><<<dim3(1), dim3(256), 0, h->stream>>>(part, off, h->scal.as<double>() + slot, 0);
  HIPCHK(hipGetLastError());
  return 0;
}

int launch_assemble(gpslam_hip_handle *h, bool save_g) {
  AsmArgs<Real> a;
  a.N = h->N; a.R = h->R;
  a.rowptr = h->rowptr.as<int>();
  a.rowLR = h->rowLR.as<Real>(); a.rowE = h->rowE.as<Real>();
  a.crowptr = h->crowptr.as<int>(); a.rowC = h->rowC.as<Real>(); a.rowCE = h->rowCE.as<Real>();
  const bool border = h->nl > 0 && !h->fs.active;   // landmark columns ride in the records only with the dense border
  a.rowM = border ? h->rowM.as<Real>() : nullptr;
  a.rowLm = border ? h->rowLm.as<int>() : nullptr;
  a.ld = h->ld;
  a.blk = h->lv[0].blk.as<Real>();
  a.gsave = save_g ? h->gsave.as<Real>() : nullptr;
  a.halo_add = has_right_rank(h) ? h->halo_add.as<Real>() : nullptr;
  const int nstates = h->N + (a.halo_add ? 1 : 0);
  dispatch_b(h->b, [&](auto tag) {
    constexpr int BB = decltype(tag)::value;
    const int waves = nblocks(nstates, 64 / BB - 1);
    k_assemble_ghost<Real, BB><<<dim3(nblocks(waves, 4)), dim3(256), 0, h->stream>>>(a);   // 1 / 8 / 16 waves per block measured slower
  });
  HIPCHK(hipGetLastError());
  return 0;
}

// k_chunk_forward_rows covers block size 12 with a single right-hand side (GPSLAM_FWD_ROWS=0 keeps the column-layout
// kernel, for A/B measurements)
bool rows_kernel_applies(const gpslam_hip_handle *h) {
  static const bool off = getenv("GPSLAM_FWD_ROWS") && atoi(getenv("GPSLAM_FWD_ROWS")) == 0;
  return h->b == 12 && h->R == 1 && !off;
}
// k_fused_level0 (assembly inside the level-0 elimination): Pose3 chains without landmark columns; on a sharded handle
// the rows of the last local state carry their R^T R to the next rank inside the addend the last chunk sends upward,
// which is what k_assemble_ghost's halo addend used to hold
// (GPSLAM_FUSE_K3=0 keeps k_assemble_ghost + k_chunk_forward_rows, for A/B measurements)
bool fused_kernel_applies(const gpslam_hip_handle *h) {
  static const bool off = getenv("GPSLAM_FUSE_K3") && atoi(getenv("GPSLAM_FUSE_K3")) == 0;
  return rows_kernel_applies(h) && h->nl == 0 && !off;
}
void launch_fwd(gpslam_hip_handle *h, const FwdArgs<Real> &a, int grid) {
  const bool fast = (4 * h->b + 2 * h->R <= 64);   // room for the separator sums in spare lanes
  // level 0 of a Pose3 chain without landmark columns: four chunks per wave, panel rows in lanes
  if (h->fuse_now && !a.no_sep && !a.add) {
    FusedArgs<Real> u;
    u.f = a;
    u.rowptr = h->rowptr.as<int>(); u.rowLR = h->rowLR.as<Real>(); u.rowE = h->rowE.as<Real>();
    u.crowptr = h->crowptr.as<int>(); u.rowC = h->rowC.as<Real>(); u.rowCE = h->rowCE.as<Real>();
    k_fused_level0<<<dim3(nblocks(grid, 4)), dim3(128), 0, h->stream>>>(u);
    return;
  }
  if (rows_kernel_applies(h) && !a.no_sep && !a.add) {
    k_chunk_forward_rows<<<dim3(nblocks(grid, 4)), dim3(64), 0, h->stream>>>(a);
    return;
  }
  dispatch_b(h->b, [&](auto tag) {
    constexpr int BB = decltype(tag)::value;
    if (fast) k_chunk_forward<Real, BB, true><<<dim3(grid), dim3(64), 0, h->stream>>>(a);
    else k_chunk_forward<Real, BB, false><<<dim3(grid), dim3(64), 0, h->stream>>>(a);
  });
}
void launch_bwd(gpslam_hip_handle *h, const BwdArgs<Real> &a, int grid) {
  dispatch_b(h->b, [&](auto tag) {
    constexpr int BB = decltype(tag)::value;
    // 16-byte pieces per lane per record: 3 wave loads cover 2 b^2 + b R <= 384 doubles, 5 cover every admissible R
    if (2 * BB * BB + BB * h->R <= 384) k_chunk_backward<Real, BB, 3><<<dim3(grid), dim3(64), 0, h->stream>>>(a);
    else k_chunk_backward<Real, BB, 5><<<dim3(grid), dim3(64), 0, h->stream>>>(a);
  });
}

// Forward elimination through the local levels.  Unsharded: the last level is the sequential top solve.
// Sharded: every level keeps its first block as separator; the last level reduces to the interface record.
int launch_forward(gpslam_hip_handle *h, double lambda) {
  const int nl = (int)h->lv.size();
  const bool sh = sharded(h);
  const size_t BS = (size_t)2 * h->b * h->b + (size_t)h->b * h->R, AS = (size_t)h->b * h->b + (size_t)h->b * h->R;
  for (int l = 0; l < nl; l++) {
    Level &v = h->lv[l];
    const bool last = (l == nl - 1);
    FwdArgs<Real> a;
    a.blk = v.blk.as<Real>();
    a.add = (l > 0) ? v.add.as<Real>() : nullptr;
    a.n = v.n; a.R = h->R;
    a.lambda = (l == 0) ? (Real)lambda : Real(0);
    a.flag = h->flag.as<int>();
    a.remote_add = nullptr;
    int grid;
    if (!sh) {
      a.up_blk = last ? nullptr : h->lv[l + 1].blk.as<Real>();
      a.up_add = last ? nullptr : h->lv[l + 1].add.as<Real>();
      a.m = last ? v.n : v.m;
      a.no_sep = last ? 1 : 0;
      a.last_has_right = 0;
      grid = last ? 1 : v.nch;
    } else {
      a.up_blk = last ? h->iface_send.as<Real>() : h->lv[l + 1].blk.as<Real>();
      // the interface record is [blk (BS) | addend slot 1 (AS)]; addend slot 0 is unused, so point one slot back
      a.up_add = last ? h->iface_send.as<Real>() + BS - AS : h->lv[l + 1].add.as<Real>();
      a.m = last ? v.n : v.m;
      a.no_sep = 0;
      a.last_has_right = has_right_rank(h) ? 1 : 0;
      if (a.last_has_right) a.remote_add = (l == 0) ? h->halo_add.as<Real>() : v.add.as<Real>() + (size_t)v.n * AS;
      grid = last ? 1 : v.nch;
    }
    launch_fwd(h, a, grid);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// Back-substitution through the local levels; xtop = separator solutions for the last level (sharded) or null
int launch_backward(gpslam_hip_handle *h, const Real *xtop) {
  const int nl = (int)h->lv.size();
  const bool sh = sharded(h);
  for (int l = nl - 1; l >= 0; l--) {
    Level &v = h->lv[l];
    const bool last = (l == nl - 1);
    BwdArgs<Real> a;
    a.blk = v.blk.as<Real>(); a.x = v.x.as<Real>();
    a.n = v.n; a.R = h->R;
    if (!sh) {
      a.xup = last ? nullptr : h->lv[l + 1].x.as<Real>();
      a.m = last ? v.n : v.m; a.no_sep = last ? 1 : 0; a.last_has_right = 0;
    } else {
      a.xup = last ? xtop : h->lv[l + 1].x.as<Real>();
      a.m = last ? v.n : v.m; a.no_sep = 0; a.last_has_right = has_right_rank(h) ? 1 : 0;
    }
    launch_bwd(h, a, last ? 1 : v.nch);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// landmark Schur complement, landmark solve, pose correction (nl > 0)
// Landmark border in two halves: (a) this handle's contribution to the Schur complement [S | gL] -- summed over the
// ranks by the caller when the chain is sharded -- and (b) the landmark solve and the correction of the chain update.
int launch_landmarks_reduce(gpslam_hip_handle *h, double lambda) {
  if (h->nl <= 0) return 0;
  LmArgs<Real> a = lm_args(h, lambda);
  if (h->nlmrows > 0) k_lm_t<Real><<<dim3(nblocks(h->nlmrows * h->R, 128)), dim3(128), 0, h->stream>>>(a);
  if (h->nlmchunks > 0) k_lm_reduce_part<Real><<<dim3(h->nlmchunks), dim3(256), 0, h->stream>>>(a);
  k_lm_reduce<Real><<<dim3(h->nl * h->R), dim3(64), 0, h->stream>>>(a);
  HIPCHK(hipGetLastError());
  return 0;
}
int launch_landmarks_solve(gpslam_hip_handle *h, double lambda) {
  if (h->nl <= 0) return 0;
  LmArgs<Real> a = lm_args(h, lambda);
  k_lm_solve<Real><<<dim3(1), dim3(64), 0, h->stream>>>(a);
  k_lm_correct<Real><<<dim3(nblocks(h->N * h->b, 256)), dim3(256), 0, h->stream>>>(a);
  HIPCHK(hipGetLastError());
  return 0;
}
int launch_landmarks(gpslam_hip_handle *h, double lambda) {
  int rc = launch_landmarks_reduce(h, lambda);
  return rc ? rc : launch_landmarks_solve(h, lambda);
}


// ---- segmented landmark elimination (fatsep.hpp)
FsArgs<Real> fs_args(gpslam_hip_handle *h, double lambda) {
  FatSepPlan &p = h->fs;
  FsArgs<Real> a;
  a.N = h->N; a.B = h->b; a.ld = h->ld; a.L = h->L; a.K = p.K; a.NB = p.NB; a.NC = p.NC; a.NCP = p.NCP;
  a.BS = 2 * h->b * h->b + h->b;
  a.cuts = p.d_cuts.as<int>(); a.segid = p.d_segid.as<int>();
  a.fat_lm_ptr = p.d_fat_lm_ptr.as<int>(); a.fat_lm = p.d_fat_lm.as<int>();
  a.lm_fat = p.d_lm_fat.as<int>(); a.lm_slot = p.d_lm_slot.as<int>();
  a.lmrow_ptr = h->lmrow_ptr.as<int>(); a.lmrow = h->lmrow.as<int>(); a.lmrow_state = h->lmrow_state.as<int>();
  a.lmpri_ptr = p.d_lmpri_ptr.as<int>(); a.lmpri = p.d_lmpri.as<int>();
  a.pri_meas = h->lpri.d_meas.as<Real>(); a.pri_sig = h->lpri.d_sig.as<Real>();
  a.lmk = h->lmk.as<Real>();
  a.rowptr = h->rowptr.as<int>(); a.rowLm = h->rowLm.as<int>();
  a.rowLR = h->rowLR.as<Real>(); a.rowE = h->rowE.as<Real>(); a.rowM = h->rowM.as<Real>();
  a.blk = h->lv[0].blk.as<Real>();
  a.fac = p.fac.as<Real>(); a.Y = p.Y.as<Real>(); a.Aseg = p.Aseg.as<Real>();
  a.Dfat = p.Dfat.as<Real>(); a.link = p.link.as<Real>(); a.gfat = p.gfat.as<Real>(); a.Qbuf = p.Qbuf.as<Real>();
  a.S1 = p.S1.as<Real>(); a.S2 = p.S2.as<Real>(); a.sv = p.sv.as<Real>(); a.xfat = p.xfat.as<Real>();
  a.gL = h->lm_gL.as<Real>(); a.dL = h->lm_dL.as<Real>();
  a.x = h->lv[0].x.as<Real>(); a.rhs = p.rhs.as<Real>();
  a.lambda = (Real)lambda; a.flag = h->flag.as<int>();
  return a;
}

int fs_solve(gpslam_hip_handle *h, double lambda) {
  FatSepPlan &p = h->fs;
  FsArgs<Real> a = fs_args(h, lambda);
  hipStream_t st = h->stream;
  const int nseg = p.K - 1;
  const size_t smem_elim = ((size_t)p.NB * (p.NB + 1) + (size_t)p.NB * (2 * p.NB + 1)) * sizeof(Real);
  const size_t smem_top = ((size_t)p.NB * (p.NB + 1) + (size_t)p.NB) * sizeof(Real);
  dispatch_b(h->b, [&](auto tag) {
    constexpr int BB = decltype(tag)::value;
    k_fs_factor<Real, BB><<<dim3(nseg), dim3(64), 0, st>>>(a);
    k_fs_sweep<Real, BB><<<dim3(nseg), dim3(((p.NC + 63) / 64) * 64), 0, st>>>(a);
  });
  k_fs_syrk<9><<<dim3(nseg, p.NCP / 16), dim3(64), 0, st>>>(a);
  k_fs_fat_assemble<Real><<<dim3(p.K), dim3(256), 0, st>>>(a);
  for (const FatSepPlan::LevelHost &lv : p.levels) {
    FatLevel fl;
    fl.elim = p.d_elim.as<int>() + (size_t)6 * lv.elim_off; fl.nelim = lv.nelim;
    fl.upd = p.d_upd.as<int>() + (size_t)3 * lv.upd_off; fl.nupd = lv.nupd;
    k_fat_elim<Real><<<dim3(lv.nelim), dim3(256), smem_elim, st>>>(a, fl);
    k_fat_update<Real><<<dim3(lv.nupd), dim3(256), 0, st>>>(a, fl);
  }
  k_fat_top<Real><<<dim3(1), dim3(256), smem_top, st>>>(a, p.top);
  for (int li = (int)p.levels.size() - 1; li >= 0; li--) {
    const FatSepPlan::LevelHost &lv = p.levels[li];
    FatLevel fl;
    fl.elim = p.d_elim.as<int>() + (size_t)6 * lv.elim_off; fl.nelim = lv.nelim;
    fl.upd = nullptr; fl.nupd = 0;
    k_fat_back<Real><<<dim3(lv.nelim), dim3(64), 0, st>>>(a, fl);
  }
  k_fs_scatter<Real><<<dim3(nblocks(std::max(p.K * h->b, h->L * h->ld), 256)), dim3(256), 0, st>>>(a);
  dispatch_b(h->b, [&](auto tag) {
    constexpr int BB = decltype(tag)::value;
    k_fs_rhs<Real, BB><<<dim3(nblocks(h->N, 128)), dim3(128), 0, st>>>(a);
    k_fs_solve1<Real, BB><<<dim3(nblocks(nseg, 64)), dim3(64), 0, st>>>(a);
  });
  HIPCHK(hipGetLastError());
  return 0;
}

// compile(): segment plan, level sets of the cyclic reduction, buffers.  per_lm: rows touching each landmark (row, left
// state), sorted by state; touch_hi: last state touched.
int fs_build(gpslam_hip_handle *h, const std::vector<int> &touch_lo, const std::vector<int> &touch_hi) {
  FatSepPlan &p = h->fs;
  const int N = h->N, b = h->b, ld = h->ld, L = h->L;
  std::vector<int> fat_of, slot_of, counts;
  if (!p.choose(N, b, ld, L, touch_lo, touch_hi, h->cfg.reserved[4], fat_of, slot_of, counts)) return fail(h, GPSLAM_E_UNSUPPORTED, p.err.c_str());
  const int K = p.K, NB = p.NB;
  std::vector<int> segid(N, 0);
  for (int k = 0; k + 1 < K; k++)
    for (int s2 = p.cuts[k] + 1; s2 < p.cuts[k + 1]; s2++) segid[s2] = k;
  for (int k = 0; k < K; k++) segid[p.cuts[k]] = -1 - k;
  std::vector<int> fat_ptr(K + 1, 0), fat_lm(L, 0);
  for (int k = 0; k < K; k++) fat_ptr[k + 1] = fat_ptr[k] + counts[k];
  for (int l = 0; l < L; l++) fat_lm[fat_ptr[fat_of[l]] + slot_of[l]] = l;
  std::vector<int> pri_ptr(L + 1, 0), pri_ids(h->lpri.idx.size());
  for (int32_t i : h->lpri.idx) pri_ptr[i + 1]++;
  for (int l = 0; l < L; l++) pri_ptr[l + 1] += pri_ptr[l];
  { std::vector<int> cur(pri_ptr.begin(), pri_ptr.end() - 1);
    for (size_t k = 0; k < h->lpri.idx.size(); k++) pri_ids[cur[h->lpri.idx[k]]++] = (int)k; }
  // level sets of the block cyclic reduction
  std::vector<int> elim, upd;
  p.levels.clear();
  std::vector<int> active(K), linkidx(K > 0 ? K - 1 : 0);
  for (int k = 0; k < K; k++) active[k] = k;
  for (int k = 0; k + 1 < K; k++) linkidx[k] = k;
  int next_link = K - 1;
  while (active.size() > 1) {
    FatSepPlan::LevelHost lv;
    lv.elim_off = (int)elim.size() / 6; lv.upd_off = (int)upd.size() / 3;
    const int n = (int)active.size();
    std::vector<int> nact, nlink;
    for (int i = 0; i < n; i += 2) {
      nact.push_back(active[i]);
      upd.push_back(active[i]);
      upd.push_back(i - 1 >= 0 ? active[i - 1] : -1);
      upd.push_back(i + 1 < n ? active[i + 1] : -1);
    }
    for (int q = 1; q < n; q += 2) {
      const int r = (q + 1 < n) ? active[q + 1] : -1;
      const int lk_new = (r >= 0) ? next_link++ : -1;
      elim.push_back(active[q]); elim.push_back(active[q - 1]); elim.push_back(r);
      elim.push_back(linkidx[q - 1]); elim.push_back(r >= 0 ? linkidx[q] : -1); elim.push_back(lk_new);
      if (r >= 0) nlink.push_back(lk_new);
    }
    lv.nelim = (int)elim.size() / 6 - lv.elim_off; lv.nupd = (int)upd.size() / 3 - lv.upd_off;
    p.levels.push_back(lv);
    active.swap(nact);
    linkidx.swap(nlink);
  }
  p.top = active[0];
  p.nlinks = std::max(next_link, 1);
  hipStream_t st = h->stream;
  HIPCHK(upload_vec(st, p.d_cuts, p.cuts));
  HIPCHK(upload_vec(st, p.d_segid, segid));
  HIPCHK(upload_vec(st, p.d_fat_lm_ptr, fat_ptr));
  HIPCHK(upload_vec(st, p.d_fat_lm, fat_lm));
  HIPCHK(upload_vec(st, p.d_lm_fat, fat_of));
  HIPCHK(upload_vec(st, p.d_lm_slot, slot_of));
  HIPCHK(upload_vec(st, p.d_lmpri_ptr, pri_ptr));
  HIPCHK(upload_vec(st, p.d_lmpri, pri_ids));
  HIPCHK(upload_vec(st, p.d_elim, elim));
  HIPCHK(upload_vec(st, p.d_upd, upd));
  const size_t NB2 = (size_t)NB * NB;
  HIPCHK(p.fac.reserve((size_t)N * 2 * b * b * sizeof(Real)));
  HIPCHK(p.Y.reserve((size_t)N * b * p.NCP * sizeof(Real)));
  HIPCHK(hipMemsetAsync(p.Y.p, 0, (size_t)N * b * p.NCP * sizeof(Real), st));   // padding columns stay zero for good
  HIPCHK(p.Aseg.reserve((size_t)(K - 1) * p.NCP * p.NCP * sizeof(Real)));
  HIPCHK(p.Dfat.reserve((size_t)K * NB2 * sizeof(Real)));
  HIPCHK(p.link.reserve((size_t)p.nlinks * NB2 * sizeof(Real)));
  HIPCHK(p.gfat.reserve((size_t)K * NB * sizeof(Real)));
  HIPCHK(p.Qbuf.reserve((size_t)K * NB2 * sizeof(Real)));
  HIPCHK(p.S1.reserve((size_t)K * NB2 * sizeof(Real)));
  HIPCHK(p.S2.reserve((size_t)K * NB2 * sizeof(Real)));
  HIPCHK(p.sv.reserve((size_t)K * 2 * NB * sizeof(Real)));
  HIPCHK(p.xfat.reserve((size_t)K * NB * sizeof(Real)));
  HIPCHK(p.rhs.reserve((size_t)N * b * sizeof(Real)));
  HIPCHK(p.partial.reserve(1024 * sizeof(Real)));
  HIPCHK(h->lm_gL.reserve((size_t)std::max(h->nl, 1) * sizeof(Real)));
  HIPCHK(h->lm_dL.reserve((size_t)std::max(h->nl, 1) * sizeof(Real)));
  const size_t smem_elim = ((size_t)NB * (NB + 1) + (size_t)NB * (2 * NB + 1)) * sizeof(Real);
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fat_elim<Real>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_elim));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fat_top<Real>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_elim));
  p.active = true;
  return 0;
}

int launch_solve(gpslam_hip_handle *h, double lambda) {
  int rc;
  if (h->fs.active) return fs_solve(h, lambda);
  if ((rc = launch_forward(h, lambda))) return rc;
  if ((rc = launch_backward(h, nullptr))) return rc;
  return launch_landmarks(h, lambda);
}

// x <- x (+) delta for the local states (and landmarks); scal[slot] = |delta|_inf
int launch_retract(gpslam_hip_handle *h, int slot) {
  RetractArgs<Real> a;
  a.pose = h->pose.as<Real>(); a.vel = h->vel.as<Real>(); a.stride = h->stride; a.N = h->N; a.R = h->R;
  a.chart = h->cfg.chart; a.first = 0; a.x = h->lv[0].x.as<Real>(); a.partial = h->partial.as<Real>(); a.flag = h->flag.as<int>();
  const int nb = nblocks(h->N, 128);
  dispatch_mf(h->mf, [&](auto tag) {
    constexpr int MF = decltype(tag)::value;
    k_retract<Real, MF><<<dim3(nb), dim3(128), 0, h->stream>>>(a);
  });
  k_final_reduce<Real><<<dim3(1), dim3(256), 0, h->stream>>>(h->partial.as<Real>(), nb, h->scal.as<double>() + slot, 1);
  if (h->nl > 0 && h->fs.active) {
    const int nbl = std::min(nblocks(h->nl, 256), 1024);
    k_fs_lm_update<Real><<<dim3(nbl), dim3(256), 0, h->stream>>>(h->lmk.as<Real>(), h->lm_dL.as<Real>(), h->nl, h->flag.as<int>(), h->fs.partial.as<Real>());
    k_fs_max_into<Real><<<dim3(1), dim3(64), 0, h->stream>>>(h->fs.partial.as<Real>(), nbl, h->scal.as<double>() + slot);
  } else if (h->nl > 0) {
    LmArgs<Real> la = lm_args(h, 0.0);
    k_lm_update<Real><<<dim3(1), dim3(64), 0, h->stream>>>(la, h->scal.as<double>() + slot);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int read_scal(gpslam_hip_handle *h, double *out, int n, int *flag) {
  HIPCHK(hipMemcpyAsync(out, h->scal.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// one Gauss-Newton iteration enqueued on the stream (no host sync); records phase events when timed.
// eval_after = false skips the error-only pass over the retracted state: inside a fixed-count run the next
// iteration's linearisation evaluates exactly that error anyway (scal[0]), so it is computed once, not twice.
int enqueue_gn(gpslam_hip_handle *h, double lambda, bool timed, bool eval_after = true) {
  int rc;
  if (timed) HIPCHK(hipEventRecord(h->ev[0], h->stream));
  if ((rc = launch_factors(h, 0, 0))) return rc;
  if (timed) HIPCHK(hipEventRecord(h->ev[1], h->stream));
  const bool fused = h->fuse_ok && h->lv.size() >= 2;   // the assembly happens inside the level-0 elimination
  if (!fused && (rc = launch_assemble(h, false))) return rc;
  if (timed) HIPCHK(hipEventRecord(h->ev[2], h->stream));
  h->fuse_now = fused;
  rc = launch_solve(h, lambda);
  h->fuse_now = false;
  if (rc) return rc;
  if (timed) HIPCHK(hipEventRecord(h->ev[3], h->stream));
  if ((rc = launch_retract(h, 2))) return rc;
  if (eval_after && (rc = launch_factors(h, 1, 1))) return rc;
  if (timed) HIPCHK(hipEventRecord(h->ev[4], h->stream));
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
  std::vector<Real> t((size_t)h->L * h->ld);
  HIPCHK(hipMemcpyAsync(t.data(), h->lmk.p, t.size() * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < t.size(); i++) h->h_lmk[i] = (double)t[i];
  return 0;
}

int backup_state(gpslam_hip_handle *h, bool restore) {
  const size_t np = (size_t)h->pd * h->stride * sizeof(Real), nv = (size_t)h->d * h->stride * sizeof(Real);
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
    const size_t nb = (size_t)h->nl * sizeof(Real);
    HIPCHK(h->lmk_bak.reserve(nb));
    if (restore) HIPCHK(hipMemcpyAsync(h->lmk.p, h->lmk_bak.p, nb, hipMemcpyDeviceToDevice, h->stream));
    else HIPCHK(hipMemcpyAsync(h->lmk_bak.p, h->lmk.p, nb, hipMemcpyDeviceToDevice, h->stream));
  }
  return 0;
}

// scal[slot] = sum_i x[i] * y[i]
int launch_dot(gpslam_hip_handle *h, const Real *x, const Real *y, int n, int slot) {
  const int nb = nblocks(n, 256);
  k_dot<Real><<<dim3(nb), dim3(256), 0, h->stream>>>(x, y, n, h->partial.as<Real>());
  k_final_reduce<Real><<<dim3(1), dim3(256), 0, h->stream>>>(h->partial.as<Real>(), nb, h->scal.as<double>() + slot, 0);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // namespace

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
  if (cfg->manifold < 0 || cfg->manifold > 4) return GPSLAM_E_INVALID;
  if (cfg->precision != GPSLAM_FP64) return GPSLAM_E_UNSUPPORTED;
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
  static const int dd[5] = {2, 3, 3, 6, 3}, pdd[5] = {2, 3, 3, 12, 9};
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
                    &h->api_e, &h->api_H};
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
  double U[36];
  if (!make_U(h->d, Qc, U)) return fail(h, GPSLAM_E_NOT_SPD, "Qc is not positive definite");
  std::memcpy(h->Qc, Qc, sizeof(double) * h->d * h->d);
  std::memcpy(h->U, U, sizeof(double) * h->d * h->d);
  return 0;
}

int gpslam_hip_set_states(gpslam_hip_handle *h, int32_t N, const double *pose, const double *vel) {
  if (!h || N <= 0 || !pose || !vel) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  const bool same = (N == h->N) && h->pose.p;
  if (N != h->N) h->compiled = false;
  h->N = N;
  h->stride = N + 1;  // one halo slot: the first state of the right neighbour segment
  std::vector<Real> sp((size_t)h->pd * h->stride, Real(0)), sv((size_t)h->d * h->stride, Real(0));
  if (same && sharded(h)) {  // keep the halo state
    HIPCHK(hipMemcpyAsync(sp.data(), h->pose.p, sp.size() * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(sv.data(), h->vel.p, sv.size() * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  for (int i = 0; i < N; i++) {
    for (int k = 0; k < h->pd; k++) sp[(size_t)k * h->stride + i] = (Real)pose[(size_t)i * h->pd + k];
    for (int k = 0; k < h->d; k++) sv[(size_t)k * h->stride + i] = (Real)vel[(size_t)i * h->d + k];
  }
  HIPCHK(h->pose.reserve(sp.size() * sizeof(Real)));
  HIPCHK(h->vel.reserve(sv.size() * sizeof(Real)));
  HIPCHK(hipMemcpyAsync(h->pose.p, sp.data(), sp.size() * sizeof(Real), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->vel.p, sv.data(), sv.size() * sizeof(Real), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int gpslam_hip_set_halo_state(gpslam_hip_handle *h, const double *pose, const double *vel) {
  if (!h || h->N <= 0 || !pose || !vel || !h->pose.p) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  std::vector<Real> pv(h->pd), vv(h->d);
  for (int k = 0; k < h->pd; k++) pv[k] = (Real)pose[k];
  for (int k = 0; k < h->d; k++) vv[k] = (Real)vel[k];
  for (int k = 0; k < h->pd; k++)
    HIPCHK(hipMemcpyAsync(h->pose.as<Real>() + (size_t)k * h->stride + h->N, &pv[k], sizeof(Real), hipMemcpyHostToDevice, h->stream));
  for (int k = 0; k < h->d; k++)
    HIPCHK(hipMemcpyAsync(h->vel.as<Real>() + (size_t)k * h->stride + h->N, &vv[k], sizeof(Real), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int gpslam_hip_get_states(gpslam_hip_handle *h, double *pose, double *vel) {
  if (!h || h->N <= 0) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  std::vector<Real> sp((size_t)h->pd * h->stride), sv((size_t)h->d * h->stride);
  HIPCHK(hipMemcpyAsync(sp.data(), h->pose.p, sp.size() * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(sv.data(), h->vel.p, sv.size() * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
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
  return upload_real(h, h->lmk, h->h_lmk);
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
  return add_meas(h, FK_INTERP_ATT, 2, 6, true, false, true, h->mf == ROT3, count, left, nullptr, m.data(), sigma, dt, tau, nullptr);
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
  h->gp_left.clear(); h->gp_dt.clear();
  for (SimpleSet *s : {&h->pri, &h->vpri, &h->btw, &h->lpri}) { s->idx.clear(); s->meas.clear(); s->sig.clear(); }
  for (MeasSet &s : h->ms) {
    s.idx.clear(); s.lm.clear(); s.meas.clear(); s.sig.clear(); s.dt.clear(); s.tau.clear(); s.aux.clear(); s.aidx.clear();
    s.any_aux = false;
  }
  h->compiled = false;
  return 0;
}

int gpslam_hip_compile(gpslam_hip_handle *h) {
  if (!h || h->N <= 0) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  const int N = h->N, d = h->d, b = h->b;
  {   // factors were range-checked against the N and L of the moment they were added; set_states / set_landmarks may
      // have shrunk either since: re-validate every stored index before it is used to size or index anything
    const int ml = max_left(h);
    auto bad = [](const std::vector<int32_t> &v, int hi) { for (int32_t i : v) if (i < 0 || i > hi) return true; return false; };
    if (bad(h->gp_left, ml) || bad(h->btw.idx, ml) || bad(h->pri.idx, N - 1) || bad(h->vpri.idx, N - 1) || bad(h->lpri.idx, h->L - 1))
      return fail(h, GPSLAM_E_INVALID, "a stored factor refers to a state / landmark that no longer exists (set_states / set_landmarks shrank the problem)");
    for (const MeasSet &s : h->ms)
      if (bad(s.idx, s.two ? ml : N - 1) || (s.haslm && bad(s.lm, h->L - 1)))
        return fail(h, GPSLAM_E_INVALID, "a stored measurement factor refers to a state / landmark that no longer exists");
  }
  h->nl = h->L * h->ld;
  h->R = 1 + h->nl;
  h->fs.active = false;
  // Landmark columns: up to kMaxRhs - 1 of them ride through the whole chain solver as extra right-hand sides (the dense
  // border, Plaza's handful of beacons).  More than that -- config 4's 5e4 locally visible landmarks -- go through the
  // segmented elimination with fat separators (fatsep.hpp); reserved[5] = 1 forces that path for any landmark count.
  const bool segmented = h->nl > 0 && (h->cfg.reserved[5] == 1 || 3 * b + h->R > 64 || h->R > kMaxRhs);
  if (segmented) {
    if (sharded(h)) return fail(h, GPSLAM_E_UNSUPPORTED, "the segmented landmark elimination runs on unsharded handles only (this round)");
    h->R = 1;   // no landmark columns inside the chain solver
  }
  // ---- row layout: rows grouped by left state; inside a state: GP, pose prior, velocity prior, between, measurements
  std::vector<int> rows_in(N + 1, 0);
  for (int32_t l : h->gp_left) rows_in[l] += b;
  for (int32_t i : h->vpri.idx) rows_in[i] += d;
  std::vector<int> crows_in(N + 1, 0);       // compact rows: pose priors, between factors
  for (int32_t i : h->pri.idx) crows_in[i] += d;
  for (int32_t i : h->btw.idx) crows_in[i] += d;
  for (MeasSet &s : h->ms) for (int32_t i : s.idx) rows_in[i] += s.rows;
  std::vector<int> rowptr(N + 2, 0);
  for (int s = 0; s <= N; s++) rowptr[s + 1] = rowptr[s] + rows_in[s];
  h->M = rowptr[N + 1];
  std::vector<int> cursor(rowptr.begin(), rowptr.end() - 1);
  std::vector<int> gp_row0(h->gp_left.size());
  for (size_t f = 0; f < h->gp_left.size(); f++) { gp_row0[f] = cursor[h->gp_left[f]]; cursor[h->gp_left[f]] += b; }
  auto place = [&](const std::vector<int32_t> &idx, int rows, std::vector<int> &row0) {
    row0.resize(idx.size());
    for (size_t f = 0; f < idx.size(); f++) { row0[f] = cursor[idx[f]]; cursor[idx[f]] += rows; }
  };
  std::vector<int> crowptr(N + 2, 0);
  for (int s = 0; s <= N; s++) crowptr[s + 1] = crowptr[s] + crows_in[s];
  h->Mc = crowptr[N + 1];
  std::vector<int> ccursor(crowptr.begin(), crowptr.end() - 1);
  auto cplace = [&](const std::vector<int32_t> &idx, int rows, std::vector<int> &row0) {
    row0.resize(idx.size());
    for (size_t f = 0; f < idx.size(); f++) { row0[f] = ccursor[idx[f]]; ccursor[idx[f]] += rows; }
  };
  std::vector<int> r_pri, r_vpri, r_btw, r_ms[kNumMeasKinds];
  cplace(h->pri.idx, d, r_pri);
  place(h->vpri.idx, d, r_vpri);
  cplace(h->btw.idx, d, r_btw);
  for (int fk = 0; fk < kNumMeasKinds; fk++) place(h->ms[fk].idx, h->ms[fk].rows, r_ms[fk]);
  int rc;
  if ((rc = upload(h, h->rowptr, rowptr))) return rc;
  if ((rc = upload(h, h->d_gp_left, h->gp_left))) return rc;
  if ((rc = upload_real(h, h->d_gp_dt, h->gp_dt))) return rc;
  if ((rc = upload(h, h->d_gp_row0, gp_row0))) return rc;
  auto up_set = [&](SimpleSet &s, const std::vector<int> &row0) -> int {
    int r2;
    if ((r2 = upload(h, s.d_idx, s.idx))) return r2;
    if ((r2 = upload_real(h, s.d_meas, s.meas))) return r2;
    if ((r2 = upload_real(h, s.d_sig, s.sig))) return r2;
    return upload(h, s.d_row0, row0);
  };
  if ((rc = up_set(h->pri, r_pri))) return rc;
  if ((rc = up_set(h->vpri, r_vpri))) return rc;
  if ((rc = up_set(h->btw, r_btw))) return rc;
  { std::vector<int> none; if ((rc = up_set(h->lpri, none))) return rc; }
  int npart = nblocks((int)h->gp_left.size(), 128) + nblocks(h->pri.count(), 128) + nblocks(h->vpri.count(), 128) +
              nblocks(h->btw.count(), 128) + 1 + 256;
  for (int fk = 0; fk < kNumMeasKinds; fk++) {
    MeasSet &s = h->ms[fk];
    if ((rc = upload(h, s.d_idx, s.idx))) return rc;
    if ((rc = upload(h, s.d_lm, s.lm))) return rc;
    if ((rc = upload_real(h, s.d_meas, s.meas))) return rc;
    if ((rc = upload_real(h, s.d_sig, s.sig))) return rc;
    if ((rc = upload(h, s.d_row0, r_ms[fk]))) return rc;
    std::vector<double> coef((size_t)(s.interp ? s.count() : 0) * 4);
    for (int k = 0; k < (s.interp ? s.count() : 0); k++) interp_coef(s.dt[k], s.tau[k], &coef[4 * (size_t)k]);
    if ((rc = upload_real(h, s.d_coef, coef))) return rc;
    if ((rc = upload_real(h, s.d_aux, s.aux))) return rc;
    if ((rc = upload(h, s.d_aidx, s.aidx))) return rc;
    npart += nblocks(s.count(), 128);
  }
  const size_t Mrows = (size_t)std::max(h->M, 1);
  HIPCHK(h->rowLR.reserve(Mrows * 2 * b * sizeof(Real)));
  HIPCHK(h->rowE.reserve(Mrows * sizeof(Real)));
  HIPCHK(h->rowC.reserve((size_t)std::max(h->Mc, 1) * b * sizeof(Real)));
  HIPCHK(h->rowCE.reserve((size_t)std::max(h->Mc, 1) * sizeof(Real)));
  if ((rc = upload(h, h->crowptr, crowptr))) return rc;
  // ---- landmark border bookkeeping
  h->nlmrows = 0;
  std::vector<int> touch_lo(h->L, -1), touch_hi(h->L, -1);   // first / last state the factors of each landmark touch
  if (h->nl > 0) {
    HIPCHK(h->rowM.reserve(Mrows * h->ld * sizeof(Real)));
    HIPCHK(h->rowLm.reserve(Mrows * sizeof(int)));
    HIPCHK(hipMemsetAsync(h->rowM.p, 0, Mrows * h->ld * sizeof(Real), h->stream));
    HIPCHK(hipMemsetAsync(h->rowLm.p, 0xFF, Mrows * sizeof(int), h->stream));  // -1: row touches no landmark
    std::vector<std::vector<std::pair<int, int>>> per_lm(h->L);              // (row, state)
    for (int fk = 0; fk < kNumMeasKinds; fk++) {
      MeasSet &s = h->ms[fk];
      if (!s.haslm) continue;
      for (int f = 0; f < s.count(); f++) {
        for (int r = 0; r < s.rows; r++) per_lm[s.lm[f]].push_back({r_ms[fk][f] + r, s.idx[f]});
        int &lo = touch_lo[s.lm[f]], &hi = touch_hi[s.lm[f]];
        const int last = s.idx[f] + (s.two ? 1 : 0);
        lo = (lo < 0) ? s.idx[f] : std::min(lo, (int)s.idx[f]);
        hi = std::max(hi, last);
      }
    }
    std::vector<int> lmrow, lmstate, lmptr(h->L + 1, 0);
    for (int l = 0; l < h->L; l++) {
      // by left state (stable: equal states keep the order the factors were added in): the segmented path walks a
      // landmark's rows along the chain, and the dense path's summation order is then independent of the call order
      std::stable_sort(per_lm[l].begin(), per_lm[l].end(), [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.second < y.second; });
      for (auto &pr : per_lm[l]) { lmrow.push_back(pr.first); lmstate.push_back(pr.second); }
      lmptr[l + 1] = (int)lmrow.size();
    }
    h->nlmrows = (int)lmrow.size();
    if (!segmented) {   // chunks of at most kLmChunk consecutive rows of one landmark
      std::vector<int> clm, cj0, cj1, cptr(h->L + 1, 0);
      for (int l = 0; l < h->L; l++) {
        for (int j = lmptr[l]; j < lmptr[l + 1]; j += kLmChunk) {
          clm.push_back(l); cj0.push_back(j); cj1.push_back(std::min(j + kLmChunk, lmptr[l + 1]));
        }
        cptr[l + 1] = (int)clm.size();
      }
      h->nlmchunks = (int)clm.size();
      if ((rc = upload(h, h->lm_chunk_lm, clm))) return rc;
      if ((rc = upload(h, h->lm_chunk_j0, cj0))) return rc;
      if ((rc = upload(h, h->lm_chunk_j1, cj1))) return rc;
      if ((rc = upload(h, h->lm_chunk_ptr, cptr))) return rc;
      HIPCHK(h->lm_part.reserve((size_t)std::max(h->nlmchunks, 1) * 2 * h->ld * h->R * sizeof(Real)));
    }
    if ((rc = upload(h, h->lmrow, lmrow))) return rc;
    if ((rc = upload(h, h->lmrow_state, lmstate))) return rc;
    if ((rc = upload(h, h->lmrow_ptr, lmptr))) return rc;
    if (!segmented) {
      HIPCHK(h->lm_t.reserve((size_t)std::max(h->nlmrows, 1) * h->R * sizeof(Real)));
      HIPCHK(h->lm_S.reserve(((size_t)h->nl * h->R + h->nl) * sizeof(Real)));   // [S (nl x R) | gL (nl)]: one buffer, one all-reduce
    }
    HIPCHK(h->lm_dL.reserve((size_t)h->nl * sizeof(Real)));
    if (!h->lmk.p) return fail(h, GPSLAM_E_INVALID, "set_landmarks() before compile()");
  }
  npart = std::max(npart, std::max(nblocks(N, 128), nblocks(N * b, 256))) + 4200;
  HIPCHK(h->partial.reserve((size_t)npart * sizeof(Real)));
  HIPCHK(h->gsave.reserve((size_t)N * b * sizeof(Real)));
  HIPCHK(h->dvec.reserve((size_t)N * b * sizeof(Real)));
  // ---- solver hierarchy: chunks of m0 states at level 0, m1 above.  Unsharded: a single-wave sequential top
  // level of <= `top` blocks.  Sharded: reduce down to one block per rank (the rank separator).
  // reserved[1] / reserved[2] override the upper-level chunk length and the size of the sequential top level
  // Level-0 chunk length: `slots` chunks run at once and a launch costs (rounds of slots) x (block steps per chunk);
  // pick the length that minimises that product (1e5 Pose3 states on 256 CUs: 13 -> 7693 chunks, one round of 12 steps
  // of the row-layout kernel; the column-layout kernel gets 25 -> 4000 chunks, one round of 24 steps).
  int m0 = 16;
  if (h->cfg.chunk > 1) {
    m0 = h->cfg.chunk;
  } else {
    hipDeviceProp_t prop;
    const int cus = (hipGetDeviceProperties(&prop, h->cfg.device) == hipSuccess && prop.multiProcessorCount > 0)
                        ? prop.multiProcessorCount : 256;
    // column-layout kernel: one chunk per wave, 4 waves per SIMD; row-layout kernel: four chunks per wave, 2 waves per SIMD
    const bool rows = rows_kernel_applies(h);
    // (the fused kernel: four two-wave workgroups of four chunks per CU)
    const long slots = (long)cus * (fused_kernel_applies(h) ? 16 : rows ? 32 : 16);
    long best = -1;
    for (int m = rows ? 8 : 16; m <= 32; m++) {
      const long chunks = (N + m - 1) / m;
      const long cost = ((chunks + slots - 1) / slots) * (m - 1);
      if (best < 0 || cost <= best) { best = cost; m0 = m; }   // ties: the longer chunk leaves fewer separators
    }
  }
  const int m1 = h->cfg.reserved[1] > 1 ? h->cfg.reserved[1] : 4;
  const int top = h->cfg.reserved[2] > 0 ? h->cfg.reserved[2] : 8;
  for (Level &v : h->lv) { v.blk.release(); v.add.release(); v.x.release(); }
  h->lv.clear();
  const size_t BS = (size_t)2 * b * b + (size_t)b * h->R, AS = (size_t)b * b + (size_t)b * h->R;
  int n = N;
  for (int l = 0;; l++) {
    Level v;
    v.n = n;
    v.m = (l == 0) ? m0 : m1;
    v.nch = nblocks(n, v.m);
    h->lv.push_back(v);
    if (n <= top) break;
    n = v.nch;
  }
  for (size_t l = 0; l < h->lv.size(); l++) {
    Level &v = h->lv[l];
    HIPCHK(v.blk.reserve((size_t)v.n * BS * sizeof(Real)));
    HIPCHK(v.x.reserve((size_t)(v.n + 1) * b * h->R * sizeof(Real)));   // + 1: solution of the neighbour rank's separator
    if (l > 0) {
      HIPCHK(v.add.reserve((size_t)(v.n + 1) * AS * sizeof(Real)));
      HIPCHK(hipMemsetAsync(v.add.p, 0, (size_t)(v.n + 1) * AS * sizeof(Real), h->stream));
    }
  }
  if (sharded(h)) {
    const int P = h->cfg.nranks;
    HIPCHK(h->halo_add.reserve(AS * sizeof(Real)));
    HIPCHK(h->iface_send.reserve((BS + AS) * sizeof(Real)));
    HIPCHK(h->iface_recv.reserve((size_t)P * (BS + AS) * sizeof(Real)));
    HIPCHK(h->top_blk.reserve((size_t)P * BS * sizeof(Real)));
    HIPCHK(h->top_x.reserve((size_t)(P + 1) * b * h->R * sizeof(Real)));
    HIPCHK(hipMemsetAsync(h->halo_add.p, 0, AS * sizeof(Real), h->stream));
    HIPCHK(hipMemsetAsync(h->iface_send.p, 0, (BS + AS) * sizeof(Real), h->stream));
    HIPCHK(hipMemsetAsync(h->iface_recv.p, 0, (size_t)P * (BS + AS) * sizeof(Real), h->stream));
    HIPCHK(hipMemsetAsync(h->top_x.p, 0, (size_t)(P + 1) * b * h->R * sizeof(Real), h->stream));
  }
  h->fuse_ok = fused_kernel_applies(h);
  if (segmented && (rc = fs_build(h, touch_lo, touch_hi))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  h->compiled = true;
  return 0;
}

int gpslam_hip_linearize_gp(gpslam_hip_handle *h, double *errors, double *jacobians) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!errors) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  const int F = (int)h->gp_left.size(), b = h->b, d = h->d;
  if (F == 0) return 0;
  HIPCHK(h->api_e.reserve((size_t)F * b * sizeof(Real)));
  if (jacobians) HIPCHK(h->api_H.reserve((size_t)F * 4 * b * d * sizeof(Real)));
  GpArgs<Real> a = gp_args(h, nullptr);
  a.rowLR = nullptr; a.rowE = nullptr;
  a.out_e = h->api_e.as<Real>(); a.out_H = jacobians ? h->api_H.as<Real>() : nullptr;
  dispatch_mf(h->mf, [&](auto tag) {
    constexpr int MF = decltype(tag)::value;
    k_gp<Real, MF, 2><<<dim3(nblocks(F, 128)), dim3(128), 0, h->stream>>>(a);
  });
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(errors, h->api_e.p, (size_t)F * b * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
  if (jacobians)
    HIPCHK(hipMemcpyAsync(jacobians, h->api_H.p, (size_t)F * 4 * b * d * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return F;
}

int gpslam_hip_error(gpslam_hip_handle *h, double *err) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!err) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  if ((rc = launch_factors(h, 1, 1))) return rc;
  double s[4];
  int flag;
  if ((rc = read_scal(h, s, 4, &flag))) return rc;
  *err = s[1];
  return 0;
}

int gpslam_hip_iterate_gn(gpslam_hip_handle *h, gpslam_hip_stats *st) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (sharded(h)) return fail(h, GPSLAM_E_INVALID, "sharded handle: use iterate_phase1 / iterate_phase2");
  (void)hipSetDevice(h->cfg.device);
  HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  if ((rc = enqueue_gn(h, 0.0, true))) return rc;
  double s[4];
  int flag = 0;
  if ((rc = read_scal(h, s, 4, &flag))) return rc;
  for (int i = 0; i < 5; i++) h->last_ms[i] = 0;
  if ((rc = collect_timing(h, h->last_ms))) return rc;
  if (st) {
    std::memset(st, 0, sizeof(*st));
    st->error_before = s[0];
    st->error_after = s[1];
    st->delta_inf_norm = s[2];
    st->iterations = 1;
    st->accepted = 1;
    st->status = flag ? GPSLAM_E_NOT_SPD : 0;
  }
  if (flag) return fail(h, GPSLAM_E_NOT_SPD, "non-positive pivot in the block elimination (indeterminate system)");
  if (!(s[1] == s[1])) return fail(h, GPSLAM_E_NAN, "NaN error after update");
  return 0;
}

int gpslam_hip_run_gn(gpslam_hip_handle *h, int32_t iters, gpslam_hip_stats *st, double *out5) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (iters <= 0) return GPSLAM_E_INVALID;
  if (sharded(h)) return fail(h, GPSLAM_E_INVALID, "sharded handle: use iterate_phase1 / iterate_phase2");
  (void)hipSetDevice(h->cfg.device);
  HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  double acc[5] = {0, 0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
    const bool timed = (out5 != nullptr);
    if ((rc = enqueue_gn(h, 0.0, timed, it == iters - 1))) return rc;
    if (timed) {
      HIPCHK(hipEventSynchronize(h->ev[4]));
      if ((rc = collect_timing(h, acc))) return rc;
    }
  }
  double s[4];
  int flag = 0;
  if ((rc = read_scal(h, s, 4, &flag))) return rc;
  if (out5) for (int i = 0; i < 5; i++) out5[i] = acc[i];
  if (st) {
    std::memset(st, 0, sizeof(*st));
    st->error_before = s[0];
    st->error_after = s[1];
    st->delta_inf_norm = s[2];
    st->iterations = iters;
    st->accepted = 1;
    st->status = flag ? GPSLAM_E_NOT_SPD : 0;
  }
  return flag ? fail(h, GPSLAM_E_NOT_SPD, "non-positive pivot in the block elimination") : 0;
}

// LevenbergMarquardtOptimizer::iterate (GTSAM 4.0 defaults: diagonalDamping = false, fixed lambda factor):
// linearise once; loop { damp with lambda I; solve; rho = (err - newErr) / (linErr(0) - linErr(delta));
// accept if rho > minModelFidelity and lambda /= factor, else lambda *= factor until lambdaUpperBound }.
// linErr(0) - linErr(delta) = 0.5 delta.g + 0.5 lambda |delta|^2 because (H + lambda I) delta = g.
int gpslam_hip_iterate_lm(gpslam_hip_handle *h, double *lambda, const gpslam_hip_params *p, gpslam_hip_stats *st) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!lambda || !p) return GPSLAM_E_INVALID;
  if (sharded(h)) return fail(h, GPSLAM_E_UNSUPPORTED, "LM on a sharded handle: not supported yet");
  (void)hipSetDevice(h->cfg.device);
  HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  if ((rc = launch_factors(h, 0, 0))) return rc;        // scal[0] = current error
  if ((rc = backup_state(h, false))) return rc;
  double s[8];
  int flag = 0;
  bool accepted = false;
  double err0 = 0, new_err = 0, dinf = 0;
  const int nx = h->N * h->b;
  for (;;) {
    HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
    if ((rc = launch_assemble(h, true))) return rc;      // rows are still those of the linearisation point
    if ((rc = launch_solve(h, *lambda))) return rc;
    k_gather_delta<Real><<<dim3(nblocks(nx, 256)), dim3(256), 0, h->stream>>>(h->lv[0].x.as<Real>(), h->N, h->R, h->b, h->dvec.as<Real>());
    if ((rc = launch_dot(h, h->dvec.as<Real>(), h->gsave.as<Real>(), nx, 3))) return rc;   // delta . g
    if ((rc = launch_dot(h, h->dvec.as<Real>(), h->dvec.as<Real>(), nx, 4))) return rc;    // |delta|^2
    if (h->nl > 0) {
      const Real *gLp = h->fs.active ? h->lm_gL.as<Real>() : h->lm_S.as<Real>() + (size_t)h->nl * h->R;
      if ((rc = launch_dot(h, h->lm_dL.as<Real>(), gLp, h->nl, 5))) return rc;
      if ((rc = launch_dot(h, h->lm_dL.as<Real>(), h->lm_dL.as<Real>(), h->nl, 6))) return rc;
    }
    if ((rc = launch_retract(h, 2))) return rc;
    if ((rc = launch_factors(h, 1, 1))) return rc;       // scal[1] = trial error
    if ((rc = read_scal(h, s, 8, &flag))) return rc;
    err0 = s[0];
    bool ok = false;
    if (!flag) {
      const double dg = s[3] + (h->nl > 0 ? s[5] : 0.0), dd = s[4] + (h->nl > 0 ? s[6] : 0.0);
      const double lin_change = 0.5 * dg + 0.5 * (*lambda) * dd;
      if (lin_change >= 0.0) {
        const double cost_change = err0 - s[1];
        const double fidelity = (lin_change > 1e-20) ? cost_change / lin_change : 0.0;
        if (fidelity > p->min_model_fidelity) { ok = true; new_err = s[1]; dinf = s[2]; }
      }
    }
    if (ok) {
      *lambda /= p->lambda_factor;
      if (*lambda < p->lambda_lower_bound) *lambda = p->lambda_lower_bound;
      accepted = true;
      break;
    }
    if ((rc = backup_state(h, true))) return rc;          // reject: restore the linearisation point
    if (*lambda >= p->lambda_upper_bound) break;
    *lambda *= p->lambda_factor;
  }
  if (st) {
    std::memset(st, 0, sizeof(*st));
    st->error_before = err0;
    st->error_after = accepted ? new_err : err0;
    st->delta_inf_norm = accepted ? dinf : 0.0;
    st->lambda = *lambda;
    st->iterations = 1;
    st->accepted = accepted ? 1 : 0;
  }
  return 0;
}

int gpslam_hip_optimize(gpslam_hip_handle *h, const gpslam_hip_params *p, gpslam_hip_stats *st) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!p) return GPSLAM_E_INVALID;
  // NonlinearOptimizer::defaultOptimize: do { cur = error(); iterate(); } while (!converged)
  double err0;
  if ((rc = gpslam_hip_error(h, &err0))) return rc;
  gpslam_hip_stats it;
  std::memset(&it, 0, sizeof(it));
  double new_err = err0, dinf = 0.0, lambda = p->lambda_initial;
  int iters = 0;
  if (!(err0 <= p->error_tol)) {
    for (;;) {
      const double cur = new_err;
      rc = p->use_lm ? gpslam_hip_iterate_lm(h, &lambda, p, &it) : gpslam_hip_iterate_gn(h, &it);
      if (rc) break;
      iters++;
      new_err = it.error_after;
      dinf = it.delta_inf_norm;
      if (iters >= p->max_iterations) break;
      if (new_err <= p->error_tol) break;
      const double abs_dec = cur - new_err, rel_dec = abs_dec / cur;
      if (rel_dec <= p->relative_error_tol || abs_dec <= p->absolute_error_tol) break;
      if (p->delta_tol > 0.0 && dinf < p->delta_tol) break;
      if (p->use_lm && !it.accepted) break;
    }
  }
  if (st) {
    std::memset(st, 0, sizeof(*st));
    st->error_before = err0;
    st->error_after = new_err;
    st->delta_inf_norm = dinf;
    st->lambda = lambda;
    st->iterations = iters;
    st->status = rc;
    st->accepted = 1;
  }
  return rc;
}

int gpslam_hip_normal_equations(gpslam_hip_handle *h, double *D, double *O, double *g, double *B) {
  int rc = need_compiled(h);
  if (rc) return rc;
  (void)hipSetDevice(h->cfg.device);
  if (B && h->fs.active) return fail(h, GPSLAM_E_UNSUPPORTED, "the dense landmark coupling B does not exist on the segmented landmark path");
  if ((rc = launch_factors(h, 0, 0))) return rc;
  if ((rc = launch_assemble(h, false))) return rc;
  const int N = h->N, b = h->b, R = h->R;
  const size_t BS = (size_t)2 * b * b + (size_t)b * R;
  std::vector<Real> blk((size_t)N * BS);
  HIPCHK(hipMemcpyAsync(blk.data(), h->lv[0].blk.p, blk.size() * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int s = 0; s < N; s++) {
    const Real *p = blk.data() + (size_t)s * BS;
    if (D) for (int k = 0; k < b * b; k++) D[(size_t)s * b * b + k] = p[k];
    if (O) for (int k = 0; k < b * b; k++) O[(size_t)s * b * b + k] = p[b * b + k];
    if (g) for (int k = 0; k < b; k++) g[(size_t)s * b + k] = p[2 * b * b + k];
    if (B && R > 1)
      for (int k = 0; k < b; k++)
        for (int r = 1; r < R; r++) B[((size_t)s * b + k) * (R - 1) + (r - 1)] = p[2 * b * b + r * b + k];
  }
  return 0;
}

// whitened Jacobian rows of the current linearisation, in row-table order (rows grouped by left state; inside a
// state: GP priors, pose priors, velocity priors, between, then the measurement kinds in FKind order)
int gpslam_hip_get_rows(gpslam_hip_handle *h, int32_t *n_rows, double *rowLR, double *rowE, double *rowM, int32_t *rowLm) {
  int rc = need_compiled(h);
  if (rc) return rc;
  (void)hipSetDevice(h->cfg.device);
  // the M rows of the full-width table, then the Mc compact rows (pose priors, between factors) expanded to full width
  const size_t M = (size_t)h->M, Mc = (size_t)h->Mc, b = (size_t)h->b, d = (size_t)h->d;
  if (n_rows) *n_rows = (int32_t)(M + Mc);
  if (!rowLR && !rowE && !rowM && !rowLm) return 0;
  if ((rc = launch_factors(h, 0, 0))) return rc;
  if (M + Mc == 0) return 0;
  std::vector<Real> cLR(Mc * b), cE(Mc);
  if (M > 0) {
    if (rowLR) HIPCHK(hipMemcpyAsync(rowLR, h->rowLR.p, M * 2 * b * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
    if (rowE) HIPCHK(hipMemcpyAsync(rowE, h->rowE.p, M * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
    if (h->nl > 0) {
      if (rowM) HIPCHK(hipMemcpyAsync(rowM, h->rowM.p, M * h->ld * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
      if (rowLm) HIPCHK(hipMemcpyAsync(rowLm, h->rowLm.p, M * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
  }
  if (Mc > 0) {
    HIPCHK(hipMemcpyAsync(cLR.data(), h->rowC.p, Mc * b * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(cE.data(), h->rowCE.p, Mc * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  for (size_t r = 0; r < Mc; r++) {
    if (rowLR) {
      double *dst = rowLR + (M + r) * 2 * b;
      for (size_t k = 0; k < 2 * b; k++) dst[k] = 0.0;
      for (size_t k = 0; k < d; k++) { dst[k] = cLR[r * b + k]; dst[b + k] = cLR[r * b + d + k]; }
    }
    if (rowE) rowE[M + r] = cE[r];
    if (rowM) for (int q = 0; q < h->ld; q++) rowM[(M + r) * h->ld + q] = 0.0;
    if (rowLm) rowLm[M + r] = -1;
  }
  return 0;
}

int gpslam_hip_block_tridiag_solve(gpslam_hip_handle *h, int32_t N, const double *D, const double *O,
                                   const double *g, double *x) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (N != h->N || h->R != 1 || sharded(h) || !D || !O || !g || !x) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  const int b = h->b;
  const size_t BS = (size_t)2 * b * b + b;
  std::vector<Real> blk((size_t)N * BS);
  for (int s = 0; s < N; s++) {
    Real *p = blk.data() + (size_t)s * BS;
    for (int k = 0; k < b * b; k++) { p[k] = D[(size_t)s * b * b + k]; p[b * b + k] = O[(size_t)s * b * b + k]; }
    for (int k = 0; k < b; k++) p[2 * b * b + k] = g[(size_t)s * b + k];
  }
  HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  HIPCHK(hipMemcpyAsync(h->lv[0].blk.p, blk.data(), blk.size() * sizeof(Real), hipMemcpyHostToDevice, h->stream));
  if ((rc = launch_solve(h, 0.0))) return rc;
  std::vector<Real> xs((size_t)N * b);
  int flag = 0;
  HIPCHK(hipMemcpyAsync(xs.data(), h->lv[0].x.p, xs.size() * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(&flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < xs.size(); i++) x[i] = xs[i];
  return flag ? fail(h, GPSLAM_E_NOT_SPD, "non-positive pivot") : 0;
}

int gpslam_hip_interpolate_poses(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt,
                                 const double *tau, double *out_pose) {
  if (!h || count < 0 || (count > 0 && (!left || !dt || !tau || !out_pose))) return GPSLAM_E_INVALID;
  if (h->N < 2) return fail(h, GPSLAM_E_INVALID, "interpolation needs at least two states");
  const int mx = max_left(h);
  for (int q = 0; q < count; q++) {
    if (left[q] < 0 || left[q] > mx) return fail(h, GPSLAM_E_INVALID, "query interval out of range");
    if (!(dt[q] > 0.0)) return fail(h, GPSLAM_E_INVALID, "delta_t must be positive");
  }
  if (count == 0) return 0;
  (void)hipSetDevice(h->cfg.device);
  const int pd = h->pd;
  std::vector<double> coef((size_t)count * 4);
  for (int q = 0; q < count; q++) interp_coef(dt[q], tau[q], &coef[4 * (size_t)q]);
  std::vector<int> li(left, left + count);
  struct Scratch {   // query buffers live for this call only
    DevBuf left, coef, out;
    ~Scratch() { left.release(); coef.release(); out.release(); }
  } sc;
  DevBuf &d_left = sc.left, &d_coef = sc.coef, &d_out = sc.out;
  int rc;
  if ((rc = upload(h, d_left, li))) return rc;
  if ((rc = upload_real(h, d_coef, coef))) return rc;
  HIPCHK(d_out.reserve((size_t)count * pd * sizeof(Real)));
  QueryArgs<Real> a;
  a.pose = h->pose.as<Real>(); a.vel = h->vel.as<Real>(); a.stride = h->stride; a.count = count;
  a.left = d_left.as<int>(); a.coef = d_coef.as<Real>(); a.out = d_out.as<Real>(); a.vw = h->vw;
  dispatch_mf(h->mf, [&](auto tag) {
    constexpr int MF = decltype(tag)::value;
    k_interp_query<Real, MF><<<dim3(nblocks(count, 128)), dim3(128), 0, h->stream>>>(a);
  });
  HIPCHK(hipGetLastError());
  std::vector<Real> out((size_t)count * pd);
  HIPCHK(hipMemcpyAsync(out.data(), d_out.p, out.size() * sizeof(Real), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (size_t k = 0; k < out.size(); k++) out_pose[k] = out[k];
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

int gpslam_hip_last_timing(gpslam_hip_handle *h, double *out5) {
  if (!h || !out5) return GPSLAM_E_INVALID;
  for (int i = 0; i < 5; i++) out5[i] = h->last_ms[i];
  return 0;
}

int gpslam_hip_time_kernel(gpslam_hip_handle *h, int32_t which, int32_t reps, double *avg_ms) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (reps <= 0 || !avg_ms || which < 0 || which > 4 || sharded(h)) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  double total = 0.0;
  for (int r = 0; r < reps; r++) {
    // bring the inputs of the timed kernel into their real state (untimed)
    if (which >= 1 && (rc = launch_factors(h, 0, 0))) return rc;
    if (which >= 2 && (rc = launch_assemble(h, false))) return rc;
    HIPCHK(hipEventRecord(h->ev[0], h->stream));
    if (which == 0) {
      GpArgs<Real> a = gp_args(h, h->partial.as<Real>());
      const int nb = nblocks(a.count, 128);
      dispatch_mf(h->mf, [&](auto tag) {
        constexpr int MF = decltype(tag)::value;
        if (MF == POSE3 && h->vw) k_gp<Real, MF, 0, true><<<dim3(nb), dim3(128), 0, h->stream>>>(a);
        else k_gp<Real, MF, 0><<<dim3(nb), dim3(128), 0, h->stream>>>(a);
      });
    } else if (which == 1) {
      if (h->fuse_ok && h->lv.size() >= 2) { *avg_ms = 0.0; return 0; }   // no such launch: see the header
      if ((rc = launch_assemble(h, false))) return rc;
    } else if (which == 2 || which == 3) {
      Level &v = h->lv[0];
      const bool top = (h->lv.size() == 1);
      FwdArgs<Real> a;
      a.blk = v.blk.as<Real>(); a.add = nullptr;
      a.up_blk = top ? nullptr : h->lv[1].blk.as<Real>();
      a.up_add = top ? nullptr : h->lv[1].add.as<Real>();
      a.n = v.n; a.m = top ? v.n : v.m; a.R = h->R; a.no_sep = top ? 1 : 0; a.last_has_right = 0;
      a.remote_add = nullptr; a.lambda = Real(0); a.flag = h->flag.as<int>();
      const int grid = top ? 1 : v.nch;
      h->fuse_now = h->fuse_ok && !top;
      launch_fwd(h, a, grid);
      h->fuse_now = false;
      if (which == 3) {  // time the level-0 back-substitution instead (separator solutions = whatever lv[1].x holds)
        HIPCHK(hipEventRecord(h->ev[0], h->stream));
        BwdArgs<Real> bw;
        bw.blk = v.blk.as<Real>(); bw.x = v.x.as<Real>(); bw.xup = top ? nullptr : h->lv[1].x.as<Real>();
        bw.n = v.n; bw.m = top ? v.n : v.m; bw.R = h->R; bw.no_sep = top ? 1 : 0; bw.last_has_right = 0;
        launch_bwd(h, bw, grid);
      }
    } else {
      HIPCHK(hipMemsetAsync(h->lv[0].x.p, 0, (size_t)h->N * h->b * h->R * sizeof(Real), h->stream));
      HIPCHK(hipEventRecord(h->ev[0], h->stream));
      RetractArgs<Real> a;
      a.pose = h->pose.as<Real>(); a.vel = h->vel.as<Real>(); a.stride = h->stride; a.N = h->N; a.R = h->R;
      a.chart = h->cfg.chart; a.first = 0; a.x = h->lv[0].x.as<Real>(); a.partial = h->partial.as<Real>(); a.flag = h->flag.as<int>();
      dispatch_mf(h->mf, [&](auto tag) {
        constexpr int MF = decltype(tag)::value;
        k_retract<Real, MF><<<dim3(nblocks(h->N, 128)), dim3(128), 0, h->stream>>>(a);
      });
    }
    HIPCHK(hipEventRecord(h->ev[1], h->stream));
    HIPCHK(hipEventSynchronize(h->ev[1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    total += ms;
  }
  *avg_ms = total / reps;
  return 0;
}

// ---------------------------------------------------------------- segment sharding

int gpslam_hip_interface_send(gpslam_hip_handle *h, void **dev_ptr, size_t *bytes) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!sharded(h) || !dev_ptr || !bytes) return GPSLAM_E_INVALID;
  const size_t BS = (size_t)2 * h->b * h->b + (size_t)h->b * h->R, AS = (size_t)h->b * h->b + (size_t)h->b * h->R;
  *dev_ptr = h->iface_send.p;
  *bytes = (BS + AS) * sizeof(Real);
  return 0;
}
int gpslam_hip_interface_recv(gpslam_hip_handle *h, void **dev_ptr, size_t *bytes) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!sharded(h) || !dev_ptr || !bytes) return GPSLAM_E_INVALID;
  const size_t BS = (size_t)2 * h->b * h->b + (size_t)h->b * h->R, AS = (size_t)h->b * h->b + (size_t)h->b * h->R;
  *dev_ptr = h->iface_recv.p;
  *bytes = (size_t)h->cfg.nranks * (BS + AS) * sizeof(Real);
  return 0;
}

// phase 1: linearise, assemble, eliminate the local segment down to its separator -> interface record
int gpslam_hip_iterate_phase1(gpslam_hip_handle *h, double lambda) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!sharded(h)) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  h->ph_lambda = lambda;
  HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  if ((rc = launch_factors(h, 0, 0))) return rc;
  if (!h->fuse_ok && (rc = launch_assemble(h, false))) return rc;
  h->fuse_now = h->fuse_ok;
  rc = launch_forward(h, lambda);
  h->fuse_now = false;
  return rc;
}

// phase 2 (after the all-gather of the records into interface_recv): every rank solves the P-block reduced
// system redundantly and back-substitutes its segment (2a); with landmarks it then forms its share of the landmark
// Schur complement, which the caller sums over the ranks (one all-reduce of gpslam_hip_landmark_reduce_buffer);
// 2b solves the landmark system (redundantly), corrects the chain update, retracts the states, the landmarks and
// the local copy of the halo state.  st (optional) returns THIS RANK's error terms and |delta|_inf; the caller
// reduces them across ranks.
int gpslam_hip_iterate_phase2a(gpslam_hip_handle *h) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!sharded(h)) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  const int P = h->cfg.nranks, b = h->b, R = h->R;
  const size_t BS = (size_t)2 * b * b + (size_t)b * R;
  k_iface_build<Real><<<dim3(nblocks(P * (int)BS, 256)), dim3(256), 0, h->stream>>>(h->iface_recv.as<Real>(), P, b, R, h->top_blk.as<Real>());
  {
    FwdArgs<Real> a;
    a.blk = h->top_blk.as<Real>(); a.add = nullptr; a.up_blk = nullptr; a.up_add = nullptr;
    a.n = P; a.m = P; a.R = R; a.no_sep = 1; a.last_has_right = 0; a.remote_add = nullptr; a.lambda = Real(0);
    a.flag = h->flag.as<int>();
    launch_fwd(h, a, 1);
    BwdArgs<Real> bw;
    bw.blk = h->top_blk.as<Real>(); bw.x = h->top_x.as<Real>(); bw.xup = nullptr;
    bw.n = P; bw.m = P; bw.R = R; bw.no_sep = 1; bw.last_has_right = 0;
    launch_bwd(h, bw, 1);
  }
  const Real *xtop = h->top_x.as<Real>() + (size_t)h->cfg.rank * b * R;   // [x_sep(rank), x_sep(rank + 1)]
  if ((rc = launch_backward(h, xtop))) return rc;
  // the ranks' Schur complements are summed: the LM damping of the landmark block goes in exactly once
  return launch_landmarks_reduce(h, (h->cfg.nranks > 1 && h->cfg.rank != 0) ? 0.0 : h->ph_lambda);
}

int gpslam_hip_landmark_reduce_buffer(gpslam_hip_handle *h, void **dev_ptr, size_t *bytes) {
  if (!h || !dev_ptr || !bytes) return GPSLAM_E_INVALID;
  *dev_ptr = h->nl > 0 ? h->lm_S.p : nullptr;
  *bytes = h->nl > 0 ? ((size_t)h->nl * h->R + h->nl) * sizeof(Real) : 0;
  return 0;
}

int gpslam_hip_iterate_phase2b(gpslam_hip_handle *h, gpslam_hip_stats *st) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!sharded(h)) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  const int b = h->b, R = h->R;
  Real *xtop = h->top_x.as<Real>() + (size_t)h->cfg.rank * b * R;
  if ((rc = launch_landmarks_solve(h, h->ph_lambda))) return rc;
  if ((rc = launch_retract(h, 2))) return rc;
  if (has_right_rank(h)) {  // keep the local copy of the neighbour's first state in step with its owner
    if (h->nl > 0) {        // its update needs the landmark correction too: delta = x0 - Z dL on that one block
      LmArgs<Real> la = lm_args(h, 0.0);
      la.N = 1; la.x = xtop + (size_t)b * R;
      k_lm_correct<Real><<<dim3(nblocks(b, 256)), dim3(256), 0, h->stream>>>(la);
    }
    RetractArgs<Real> a;
    a.pose = h->pose.as<Real>(); a.vel = h->vel.as<Real>(); a.stride = h->stride; a.N = 1; a.R = R;
    a.chart = h->cfg.chart; a.first = h->N; a.x = xtop + (size_t)b * R; a.partial = h->partial.as<Real>() + nblocks(h->N, 128) + 4; a.flag = h->flag.as<int>();
    dispatch_mf(h->mf, [&](auto tag) {
      constexpr int MF = decltype(tag)::value;
      k_retract<Real, MF><<<dim3(1), dim3(128), 0, h->stream>>>(a);
    });
  }
  // the error of the new state: only when the caller wants statistics (inside a fixed-count run the next
  // iteration's linearisation evaluates it anyway)
  if (st && (rc = launch_factors(h, 1, 1))) return rc;
  HIPCHK(hipGetLastError());
  if (st) {
    double s[4];
    int flag = 0;
    if ((rc = read_scal(h, s, 4, &flag))) return rc;
    std::memset(st, 0, sizeof(*st));
    st->error_before = s[0];
    st->error_after = s[1];
    st->delta_inf_norm = s[2];
    st->iterations = 1;
    st->accepted = 1;
    st->status = flag ? GPSLAM_E_NOT_SPD : 0;
    if (flag) return fail(h, GPSLAM_E_NOT_SPD, "non-positive pivot in the block elimination");
  }
  return 0;
}

// ---- Levenberg-Marquardt on a sharded chain.  The decisions of gpslam_hip_iterate_lm need three global sums (error,
// delta . g, |delta|^2), so the loop lives with the caller (gpslam_amd/sharded.py: ShardedSolver.iterate_lm) and the
// library provides its device-side steps:
//   lm_begin            linearise once at the current estimate, remember it
//   lm_trial_phase1     damp with lambda, eliminate the segment            -> all-gather of the interface records
//   iterate_phase2a     reduced solve, back-substitution, landmark share   -> all-reduce of the landmark buffer
//   lm_trial_phase2     landmark solve, trial update, this rank's scalars  -> all-reduce of the scalars, decision
//   lm_reject           back to the linearisation point (an accepted trial needs nothing)
int gpslam_hip_lm_begin(gpslam_hip_handle *h) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!sharded(h)) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  if ((rc = launch_factors(h, 0, 0))) return rc;        // rows + scal[0] = this rank's error
  return backup_state(h, false);
}

int gpslam_hip_lm_trial_phase1(gpslam_hip_handle *h, double lambda) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!sharded(h)) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  h->ph_lambda = lambda;
  HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  if ((rc = launch_assemble(h, true))) return rc;       // rows are still those of the linearisation point
  return launch_forward(h, lambda);
}

// out6 = {error at the linearisation point, trial error, |delta|_inf, delta . g, |delta|^2, indefinite-pivot flag},
// all for THIS rank's states (landmark terms on rank 0 only); sums / maxima over the ranks give the global values
int gpslam_hip_lm_trial_phase2(gpslam_hip_handle *h, double *out6) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!sharded(h) || !out6) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  const int b = h->b, R = h->R, nx = h->N * h->b;
  Real *xtop = h->top_x.as<Real>() + (size_t)h->cfg.rank * b * R;
  if ((rc = launch_landmarks_solve(h, h->ph_lambda))) return rc;
  if (has_right_rank(h) && h->nl > 0) {
    LmArgs<Real> la = lm_args(h, 0.0);
    la.N = 1; la.x = xtop + (size_t)b * R;
    k_lm_correct<Real><<<dim3(nblocks(b, 256)), dim3(256), 0, h->stream>>>(la);
  }
  k_gather_delta<Real><<<dim3(nblocks(nx, 256)), dim3(256), 0, h->stream>>>(h->lv[0].x.as<Real>(), h->N, h->R, h->b, h->dvec.as<Real>());
  if ((rc = launch_dot(h, h->dvec.as<Real>(), h->gsave.as<Real>(), nx, 3))) return rc;   // delta . g
  if ((rc = launch_dot(h, h->dvec.as<Real>(), h->dvec.as<Real>(), nx, 4))) return rc;    // |delta|^2
  const bool lm_here = h->nl > 0 && h->cfg.rank == 0;    // replicated landmark update: counted once
  if (lm_here) {
    if ((rc = launch_dot(h, h->lm_dL.as<Real>(), h->lm_S.as<Real>() + (size_t)h->nl * h->R, h->nl, 5))) return rc;
    if ((rc = launch_dot(h, h->lm_dL.as<Real>(), h->lm_dL.as<Real>(), h->nl, 6))) return rc;
  }
  // the rows of the last local state also feed the gradient of the neighbour's first state (halo_add = [RD | Rg]):
  // that share of delta . g is only known here
  if (has_right_rank(h)) {
    if ((rc = launch_dot(h, xtop + (size_t)b * R, h->halo_add.as<Real>() + (size_t)b * b, b, 7))) return rc;
  }
  if ((rc = launch_retract(h, 2))) return rc;
  if (has_right_rank(h)) {
    RetractArgs<Real> a;
    a.pose = h->pose.as<Real>(); a.vel = h->vel.as<Real>(); a.stride = h->stride; a.N = 1; a.R = R;
    a.chart = h->cfg.chart; a.first = h->N; a.x = xtop + (size_t)b * R; a.partial = h->partial.as<Real>() + nblocks(h->N, 128) + 4; a.flag = h->flag.as<int>();
    dispatch_mf(h->mf, [&](auto tag) {
      constexpr int MF = decltype(tag)::value;
      k_retract<Real, MF><<<dim3(1), dim3(128), 0, h->stream>>>(a);
    });
  }
  if ((rc = launch_factors(h, 1, 1))) return rc;         // scal[1] = trial error
  double s[8];
  int flag = 0;
  if ((rc = read_scal(h, s, 8, &flag))) return rc;
  out6[0] = s[0]; out6[1] = s[1]; out6[2] = s[2];
  out6[3] = s[3] + (has_right_rank(h) ? s[7] : 0.0) + (lm_here ? s[5] : 0.0);
  out6[4] = s[4] + (lm_here ? s[6] : 0.0);
  out6[5] = flag ? 1.0 : 0.0;
  return 0;
}

int gpslam_hip_lm_reject(gpslam_hip_handle *h) {
  int rc = need_compiled(h);
  if (rc) return rc;
  if (!sharded(h)) return GPSLAM_E_INVALID;
  (void)hipSetDevice(h->cfg.device);
  return backup_state(h, true);
}

// 2a + 2b for chains without landmarks (or a single rank): nothing to reduce in between
int gpslam_hip_iterate_phase2(gpslam_hip_handle *h, gpslam_hip_stats *st) {
  if (h && h->nl > 0 && h->cfg.nranks > 1)
    return fail(h, GPSLAM_E_INVALID, "landmarks on a sharded chain: phase2a, all-reduce of the landmark buffer, phase2b");
  int rc = gpslam_hip_iterate_phase2a(h);
  return rc ? rc : gpslam_hip_iterate_phase2b(h, st);
}

}  // extern "C"
