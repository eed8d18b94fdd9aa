// api.hip -- the C ABI of include/gpslam_hip.h: handle life cycle, graph building, and the dispatch to the fp64-row /
// fp32-row halves (api_impl64.hip, api_impl32.hip; both are api_impl.inc)
#include "api_common.hpp"

namespace impl64 {
#include "api_decl.inc"
}
namespace impl32 {
#include "api_decl.inc"
}

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

// One LevenbergMarquardtOptimizer::tryLambda decision (GTSAM 4.0.x, as recalled: PARITY UNPINNED) from the scalars of a trial.
// Host arithmetic only; every Levenberg-Marquardt loop of this repository -- gpslam_hip_iterate_lm, the sharded / split loops
// of the C++ drivers and of the Python mirror -- takes its branch here, so that all of them follow one rule.
int gpslam_hip_lm_decide(const double *s6, const gpslam_hip_params *p, double *lambda, int32_t *accepted, int32_t *done) {
  if (!s6 || !p || !lambda || !accepted || !done) return GPSLAM_E_INVALID;
  bool ok = false, stop_searching = false;
  if (s6[5] == 0.0) {                                     // the damped system was solved (no indefinite pivot)
    const double lin_change = 0.5 * s6[3] + 0.5 * (*lambda) * s6[4];   // linErr(0) - linErr(delta), (H + lambda I) delta = g
    if (lin_change >= 0.0) {                              // "step is valid"
      const double cost_change = s6[0] - s6[1];
      if (lin_change > 1e-20) ok = cost_change / lin_change > p->min_model_fidelity;
      // the small-cost-change stop: a trial that moves the cost by less than relativeErrorTol * error ends the search for a
      // lambda whether or not it is kept
      if (std::fabs(cost_change) < p->relative_error_tol * s6[0]) stop_searching = true;
    }
  }
  if (ok) {                                               // decreaseLambda (useFixedLambdaFactor)
    *lambda /= p->lambda_factor;
    if (*lambda < p->lambda_lower_bound) *lambda = p->lambda_lower_bound;
    *accepted = 1; *done = 1;
    return 0;
  }
  *accepted = 0;
  if (stop_searching) { *done = 1; return 0; }            // lambda and the values stay as they were
  *lambda *= p->lambda_factor;                            // increaseLambda, then the bound (GTSAM's order)
  *done = (*lambda >= p->lambda_upper_bound) ? 1 : 0;     // "giving up because cannot decrease error with maximum lambda"
  return 0;
}

int gpslam_hip_set_collectives(gpslam_hip_handle *h, gpslam_hip_all_gather_fn all_gather, gpslam_hip_all_reduce_sum_fn all_reduce_sum, void *user) {
  if (!h) return GPSLAM_E_INVALID;
  h->coll_gather = all_gather; h->coll_reduce = all_reduce_sum; h->coll_user = user;
  return 0;
}

uint32_t gpslam_hip_abi_version(void) { return GPSLAM_HIP_ABI_VERSION; }
size_t gpslam_hip_struct_size(int32_t which) {
  switch (which) {
    case GPSLAM_STRUCT_CONFIG: return sizeof(gpslam_hip_config);
    case GPSLAM_STRUCT_CONFIG_V2: return sizeof(gpslam_hip_config_v2);
    case GPSLAM_STRUCT_STATS: return sizeof(gpslam_hip_stats);
    case GPSLAM_STRUCT_PARAMS: return sizeof(gpslam_hip_params);
  }
  return 0;
}

// v1: the eight anonymous words are today's named fields, in order ([7] must be 0)
int gpslam_hip_create(const gpslam_hip_config *c1, gpslam_hip_handle **out) {
  if (!c1 || !out) return GPSLAM_E_INVALID;
  if (c1->reserved[7] != 0) return GPSLAM_E_INVALID;
  gpslam_hip_config_v2 c;
  std::memset(&c, 0, sizeof(c));
  c.struct_size = (uint32_t)sizeof(c);
  c.manifold = c1->manifold; c.precision = c1->precision; c.device = c1->device; c.chart = c1->chart;
  c.landmark_dim = c1->landmark_dim; c.chunk = c1->chunk; c.rank = c1->rank; c.nranks = c1->nranks;
  c.force_sharded = c1->reserved[0]; c.upper_chunk = c1->reserved[1]; c.top_blocks = c1->reserved[2]; c.velocity = c1->reserved[3];
  c.segment_length = c1->reserved[4]; c.force_segmented = c1->reserved[5]; c.plan = c1->reserved[6];
  return gpslam_hip_create_v2(&c, out);
}

int gpslam_hip_create_v2(const gpslam_hip_config_v2 *in, gpslam_hip_handle **out) {
  if (!in || !out) return GPSLAM_E_INVALID;
  // the caller's struct may be shorter (an older header: the missing fields are 0) or longer (a newer one: accepted only if the
  // bytes this build does not know are zero -- a knob we cannot honour must not be dropped silently)
  if (in->struct_size < offsetof(gpslam_hip_config_v2, nranks) + sizeof(int32_t)) return GPSLAM_E_INVALID;
  gpslam_hip_config_v2 cv;
  std::memset(&cv, 0, sizeof(cv));
  std::memcpy(&cv, in, in->struct_size < sizeof(cv) ? in->struct_size : sizeof(cv));
  for (size_t i = sizeof(cv); i < in->struct_size; i++)
    if (reinterpret_cast<const unsigned char *>(in)[i] != 0) return GPSLAM_E_UNSUPPORTED;
  cv.struct_size = (uint32_t)sizeof(cv);
  const gpslam_hip_config_v2 *cfg = &cv;
  if (cfg->manifold < 0 || cfg->manifold > GPSLAM_ROT3_BIAS) return GPSLAM_E_INVALID;
  if (cfg->precision != GPSLAM_FP64 && cfg->precision != GPSLAM_FP32) return GPSLAM_E_INVALID;
  if (cfg->landmark_dim != 0 && cfg->landmark_dim != 2 && cfg->landmark_dim != 3) return GPSLAM_E_INVALID;
  if (cfg->nranks < 0 || (cfg->nranks > 1 && (cfg->rank < 0 || cfg->rank >= cfg->nranks))) return GPSLAM_E_INVALID;
  if (cfg->velocity != 0 && (cfg->velocity != GPSLAM_VELOCITY_WORLD_VW || cfg->manifold != GPSLAM_POSE3)) return GPSLAM_E_INVALID;
  constexpr int kPlanBits = GPSLAM_PLAN_UNFUSED_LEVEL0 | GPSLAM_PLAN_COLUMN_LEVEL0 | GPSLAM_PLAN_LEVELS_OF_FOUR | GPSLAM_PLAN_FS_TWO_LAUNCHES | GPSLAM_PLAN_GP_ROWS | GPSLAM_PLAN_GENERIC_QC | GPSLAM_PLAN_MEAS_ROWS | GPSLAM_PLAN_SEPARATE_RETRACT;
  if ((cfg->plan & ~kPlanBits) != 0) return GPSLAM_E_INVALID;
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
  h->vw = (cfg->velocity == GPSLAM_VELOCITY_WORLD_VW) ? 1 : 0;
  std::memset(h->Qc, 0, sizeof(h->Qc));
  std::memset(h->U, 0, sizeof(h->U));
  for (int i = 0; i < h->d; i++) h->Qc[i * h->d + i] = h->U[i * h->d + i] = 1.0;
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return GPSLAM_E_HIP; }
  if (hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) { delete h; return GPSLAM_E_HIP; }
  for (int i = 0; i < 6; i++)
    if (hipEventCreate(&h->ev[i]) != hipSuccess) { delete h; return GPSLAM_E_HIP; }
  if (hipEventCreate(&h->ev_l0a) != hipSuccess || hipEventCreate(&h->ev_l0b) != hipSuccess) { delete h; return GPSLAM_E_HIP; }
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
                    &h->api_e, &h->api_H, &h->gps, &h->gpidx, &h->dU, &h->gsave2, &h->partial2, &h->brec, &h->btwidx,
                    &h->rowI, &h->irowptr, &h->coll_s, &h->coll_r, &h->d_clo_second, &h->clo_A, &h->clo_Y, &h->simd_cnt};
  for (DevBuf *b : bufs) b->release();
  for (SimpleSet *s : {&h->pri, &h->vpri, &h->btw, &h->lpri, &h->clo}) s->release();
  for (MeasSet &s : h->ms) s.release();
  h->fs.release();
  h->lm_gL.release();
  for (Level &v : h->lv) { v.blk.release(); v.add.release(); v.x.release(); }
  for (int i = 0; i < 6; i++) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  if (h->ev_l0a) (void)hipEventDestroy(h->ev_l0a);
  if (h->ev_l0b) (void)hipEventDestroy(h->ev_l0b);
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
int gpslam_hip_add_between_pairs(gpslam_hip_handle *h, int32_t count, const int32_t *first, const int32_t *second,
                                 const double *measured, const double *sigmas) {
  if (!h || count < 0 || (count > 0 && (!first || !second || !measured || !sigmas))) return GPSLAM_E_INVALID;
  for (int k = 0; k < count; k++) {
    if (first[k] < 0 || first[k] >= h->N || second[k] < 0 || second[k] >= h->N) return fail(h, GPSLAM_E_INVALID, "factor index out of range");
    if (first[k] == second[k]) return fail(h, GPSLAM_E_INVALID, "a BetweenFactor needs two different states");
  }
  for (size_t k = 0; k < (size_t)count * h->d; k++)
    if (!(sigmas[k] > 0.0)) return fail(h, GPSLAM_E_INVALID, "sigmas must be positive");
  for (int k = 0; k < count; k++) {
    const double *m = measured + (size_t)k * h->pd, *sg = sigmas + (size_t)k * h->d;
    if (second[k] == first[k] + 1) {          // the chain's own BetweenFactor(x_i, x_i+1): the ordinary row path
      int rc = add_simple(h, h->btw, h->pd, h->d, 1, first + k, m, sg, max_left(h));
      if (rc) return rc;
      continue;
    }
    if (sharded(h)) return fail(h, GPSLAM_E_UNSUPPORTED, "loop closures on a sharded handle (nranks > 1): both states must live on one rank");
    h->clo.width = h->pd;
    h->clo.idx.push_back(first[k]);
    h->clo_second.push_back(second[k]);
    h->clo.meas.insert(h->clo.meas.end(), m, m + h->pd);
    h->clo.sig.insert(h->clo.sig.end(), sg, sg + h->d);
  }
  h->compiled = false;
  return 0;
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
int gpslam_hip_add_interp_projection_ds2(gpslam_hip_handle *h, int32_t count, const int32_t *left, const int32_t *landmark,
                                         const double *measured, const double *sigmas, const double *dt, const double *tau,
                                         const double *K9, const double *sensor) {
  if (!h || !K9) return GPSLAM_E_INVALID;
  if (h->vw) return fail(h, GPSLAM_E_UNSUPPORTED, "the reference has no projection factor for the VW velocity family");
  return add_meas(h, FK_INTERP_PROJ, 2, 2, true, true, true, h->mf == POSE3 && h->ld == 3, count, left, landmark, measured, sigmas, dt, tau, sensor, K9, K9 + 5);
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
  for (SimpleSet *s : {&h->pri, &h->vpri, &h->btw, &h->lpri, &h->clo}) { s->idx.clear(); s->meas.clear(); s->sig.clear(); }
  h->clo_second.clear();
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
                        ((fused && h->struct_ok) || h->struct3_ok) ? ((fused && h->struct_ok && h->irow_ok) ? 2 : 1) : 0, h->M, h->Mc, h->R};
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

#ifdef GPS_TRACE_FUSED
int gpslam_hip_debug_fused_trace(gpslam_hip_handle *h, unsigned long long *out, int32_t max_waves, int32_t *nwaves) {
  if (!h || !out || !nwaves) return GPSLAM_E_INVALID;
  *nwaves = h->dbg_trace_waves;
  const int n = h->dbg_trace_waves < max_waves ? h->dbg_trace_waves : max_waves;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return GPSLAM_E_HIP;
  if (n > 0 && hipMemcpy(out, h->dbg_trace.p, (size_t)n * 64 * 8, hipMemcpyDeviceToHost) != hipSuccess) return GPSLAM_E_HIP;
  return 0;
}
#endif
int gpslam_hip_set_level0_stamps(gpslam_hip_handle *h, int32_t on) {
  if (!h) return GPSLAM_E_INVALID;
  h->l0_stamps = on != 0;
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
