// gtsam_adapter.hpp -- run an EXISTING GTSAM + gpslam graph on the MI355X path without touching the reference.
//
// Compiled only where real GTSAM and the reference headers are installed (neither exists in the build image, so there
// this header is an empty translation unit; it is host-only code -- no HIP, no second code path on the device side).
//
// How a gpslam user adopts it:
//   1. construct the GP factors through the RECORDING subclasses below, e.g.
//          graph.add(gpslam_hip::GaussianProcessPriorPose3(x1, v1, x2, v2, delta_t, Qc_model));
//      instead of gpslam::GaussianProcessPriorPose3(...).  Each one IS-A reference factor (it derives from it and forwards
//      the constructor), so the graph still linearises, prints, serialises and optimises with stock GTSAM exactly as
//      before; the subclass only keeps the constructor arguments readable (the reference stores delta_t_, tau_, GPbase_
//      private and offers no getters: gpslam/gp/GaussianProcessPriorPose3.h:31, gpslam/slam/GPInterpolatedRangeFactorPose2.h:26-32).
//   2. replace   gtsam::LevenbergMarquardtOptimizer opt(graph, init, params);   by
//                gpslam_hip::HipChainOptimizerPose3 opt(graph, init, params);   (…Pose2 for SE(2) graphs)
//      iterate() / optimize() / error() / values() / iterations() / lambda() keep GTSAM's meaning
//      (call sites: matlab/PlazaPose2.m:208-228, gpslam/gp/tests/testGaussianProcessPriorPose3.cpp:185-188).
// Graph requirements are those of the C ABI (include/gpslam_hip.h): keys Symbol('x'|'v'|'l', i), chain order (a factor couples
// state i with state i + 1 only; checked, not assumed).  Every GP prior keeps its own Qc_model (gpslam_hip_add_gp_priors_qc).
// CI without GTSAM type-checks this header against the declaration-only stand-ins of tests/cpp/gtsam_decl/ (-fsyntax-only).
#pragma once

#if defined(__has_include)
#if __has_include(<gtsam/nonlinear/NonlinearFactorGraph.h>) && __has_include(<gpslam/gp/GaussianProcessPriorPose3.h>)
#define GPSLAM_HIP_HAVE_GTSAM 1
#endif
#endif

#ifdef GPSLAM_HIP_HAVE_GTSAM

#include <gtsam/geometry/Pose2.h>
#include <gtsam/geometry/Pose3.h>
#include <gtsam/inference/Symbol.h>
#include <gtsam/linear/NoiseModel.h>
#include <gtsam/nonlinear/LevenbergMarquardtParams.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/PriorFactor.h>

#include <gpslam/gp/GPutils.h>
#include <gpslam/gp/GaussianProcessPriorPose2.h>
#include <gpslam/gp/GaussianProcessPriorPose3.h>
#include <gpslam/slam/GPInterpolatedRangeFactorPose2.h>
#include <gpslam/slam/GPInterpolatedRangeFactorPose3.h>

#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gpslam_hip.h"

namespace gpslam_hip {

/// constructor arguments a reference factor does not expose again
struct GpParams {
  double delta_t = 0, tau = 0;
  gtsam::Matrix Qc;
};

#define GPSLAM_HIP_RECORDING_PRIOR(CLS)                                                                              \
  class CLS : public gpslam::CLS {                                                                                   \
   public:                                                                                                           \
    CLS(gtsam::Key poseKey1, gtsam::Key velKey1, gtsam::Key poseKey2, gtsam::Key velKey2, double delta_t,           \
        const gtsam::SharedNoiseModel &Qc_model)                                                                     \
        : gpslam::CLS(poseKey1, velKey1, poseKey2, velKey2, delta_t, Qc_model) {                                     \
      gp.delta_t = delta_t;                                                                                          \
      gp.Qc = gpslam::getQc(Qc_model);                                                                               \
    }                                                                                                                \
    gtsam::NonlinearFactor::shared_ptr clone() const override {                                                      \
      return gtsam::NonlinearFactor::shared_ptr(new CLS(*this));                                                     \
    }                                                                                                                \
    GpParams gp;                                                                                                     \
  };
GPSLAM_HIP_RECORDING_PRIOR(GaussianProcessPriorPose3)
GPSLAM_HIP_RECORDING_PRIOR(GaussianProcessPriorPose2)

#define GPSLAM_HIP_RECORDING_RANGE(CLS, POSE)                                                                        \
  class CLS : public gpslam::CLS {                                                                                   \
   public:                                                                                                           \
    CLS(double measured, const gtsam::SharedNoiseModel &meas_model, const gtsam::SharedNoiseModel &Qc_model,        \
        gtsam::Key poseKey1, gtsam::Key velKey1, gtsam::Key poseKey2, gtsam::Key velKey2, gtsam::Key pointKey,       \
        double delta_t, double tau, boost::optional<POSE> body_P_sensor = boost::none)                               \
        : gpslam::CLS(measured, meas_model, Qc_model, poseKey1, velKey1, poseKey2, velKey2, pointKey, delta_t, tau,  \
                      body_P_sensor), measured_value(measured), sensor(body_P_sensor) {                              \
      gp.delta_t = delta_t;                                                                                          \
      gp.tau = tau;                                                                                                  \
      gp.Qc = gpslam::getQc(Qc_model);                                                                               \
    }                                                                                                                \
    gtsam::NonlinearFactor::shared_ptr clone() const override {                                                      \
      return gtsam::NonlinearFactor::shared_ptr(new CLS(*this));                                                     \
    }                                                                                                                \
    GpParams gp;                                                                                                     \
    double measured_value;                                                                                           \
    boost::optional<POSE> sensor;                                                                                    \
  };
GPSLAM_HIP_RECORDING_RANGE(GPInterpolatedRangeFactorPose3, gtsam::Pose3)
GPSLAM_HIP_RECORDING_RANGE(GPInterpolatedRangeFactorPose2, gtsam::Pose2)

namespace detail {
inline std::vector<double> sigmas(const gtsam::SharedNoiseModel &m) {
  auto diag = boost::dynamic_pointer_cast<gtsam::noiseModel::Diagonal>(m);
  if (!diag) throw std::invalid_argument("HipChainOptimizer: the noise models of priors, odometry and range factors must be diagonal");
  const gtsam::Vector s = diag->sigmas();
  return std::vector<double>(s.data(), s.data() + s.size());
}
struct HandleDeleter { void operator()(gpslam_hip_handle *h) const { if (h) gpslam_hip_destroy(h); } };
inline void pack(const gtsam::Pose3 &p, double *o) {
  const gtsam::Matrix3 R = p.rotation().matrix();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) o[3 * i + j] = R(i, j);
  o[9] = p.translation().x(); o[10] = p.translation().y(); o[11] = p.translation().z();
}
inline void pack(const gtsam::Pose2 &p, double *o) { o[0] = p.x(); o[1] = p.y(); o[2] = p.theta(); }
inline gtsam::Pose3 unpack3(const double *o) {
  gtsam::Matrix3 R;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R(i, j) = o[3 * i + j];
  return gtsam::Pose3(gtsam::Rot3(R), gtsam::Point3(o[9], o[10], o[11]));
}
template <typename POSE> struct Traits;
template <> struct Traits<gtsam::Pose3> {
  enum { manifold = GPSLAM_POSE3, d = 6, pd = 12, ld = 3 };
  typedef gtsam::Vector6 Vel;
  typedef gtsam::Point3 Point;
  typedef GaussianProcessPriorPose3 Prior;
  typedef GPInterpolatedRangeFactorPose3 Range;
  static gtsam::Pose3 un(const double *o) { return unpack3(o); }
};
template <> struct Traits<gtsam::Pose2> {
  enum { manifold = GPSLAM_POSE2, d = 3, pd = 3, ld = 2 };
  typedef gtsam::Vector3 Vel;
  typedef gtsam::Point2 Point;
  typedef GaussianProcessPriorPose2 Prior;
  typedef GPInterpolatedRangeFactorPose2 Range;
  static gtsam::Pose2 un(const double *o) { return gtsam::Pose2(o[0], o[1], o[2]); }
};
}  // namespace detail

/// GaussNewtonOptimizer / LevenbergMarquardtOptimizer over a chain-structured gpslam graph, evaluated by libgpslam_hip.so
template <typename POSE> class HipChainOptimizerT {
  typedef detail::Traits<POSE> TR;

 public:
  HipChainOptimizerT(const gtsam::NonlinearFactorGraph &graph, const gtsam::Values &init,
                     const gtsam::LevenbergMarquardtParams &params = gtsam::LevenbergMarquardtParams(), bool use_lm = true, int device = 0)
      : init_(init) {
    // variables: pose i = Symbol('x', i), velocity i = Symbol('v', i), landmark j = Symbol('l', j), ordered by index
    for (const gtsam::Key k : init.keys()) {
      const gtsam::Symbol s(k);
      if (s.chr() == 'x') states_[s.index()] = (int)states_.size();
      if (s.chr() == 'l') lms_[s.index()] = (int)lms_.size();
    }
    int i = 0;
    for (auto &kv : states_) kv.second = i++;
    i = 0;
    for (auto &kv : lms_) kv.second = i++;
    const int N = (int)states_.size(), L = (int)lms_.size();
    std::vector<double> P((size_t)N * TR::pd), V((size_t)N * TR::d, 0.0), LM((size_t)L * TR::ld);
    for (auto &kv : states_) {
      detail::pack(init.at<POSE>(gtsam::Symbol('x', kv.first)), &P[(size_t)kv.second * TR::pd]);
      const gtsam::Key vk = gtsam::Symbol('v', kv.first);
      if (init.exists(vk)) {
        const typename TR::Vel v = init.at<typename TR::Vel>(vk);
        for (int q = 0; q < TR::d; q++) V[(size_t)kv.second * TR::d + q] = v(q);
      }
    }
    for (auto &kv : lms_) {
      const typename TR::Point p = init.at<typename TR::Point>(gtsam::Symbol('l', kv.first));
      for (int q = 0; q < TR::ld; q++) LM[(size_t)kv.second * TR::ld + q] = p(q);
    }
    gpslam_hip_config_v2 cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = (uint32_t)sizeof(cfg);
    cfg.manifold = TR::manifold; cfg.precision = GPSLAM_FP64; cfg.device = device; cfg.nranks = 1;
    cfg.chart = ((int)TR::manifold == (int)GPSLAM_POSE2) ? GPSLAM_CHART_FIRST_ORDER : GPSLAM_CHART_EXPMAP;
    cfg.landmark_dim = L > 0 ? TR::ld : 0;
    {   // (owned from here on: whatever throws below, the handle is destroyed -- ADVICE r2)
      gpslam_hip_handle *raw = nullptr;
      if ((gpslam_hip_abi_version() >> 16) != (uint32_t)GPSLAM_HIP_ABI_MAJOR || gpslam_hip_create_v2(&cfg, &raw) != 0) throw std::runtime_error("HipChainOptimizer: no usable HIP device");
      hh_.reset(raw);
      h_ = raw;
    }
    check(gpslam_hip_set_states(h_, N, P.data(), V.data()), "set_states");
    if (L > 0) check(gpslam_hip_set_landmarks(h_, L, LM.data()), "set_landmarks");
    bool qc_set = false;
    for (const auto &f : graph) {
      if (!f) continue;
      if (auto gp = boost::dynamic_pointer_cast<typename TR::Prior>(f)) {
        // the chain solver takes (x_i, v_i, x_i+1, v_i+1) only: verify it instead of reading key1() alone
        const int32_t left = chain_left(gp->key1(), gp->key2(), gp->key3(), gp->key4());
        if (!qc_set) { set_qc(gp->gp.Qc); qc_set = true; }      // (the shared Qc only serves interpolation queries)
        std::vector<double> q((size_t)TR::d * TR::d);
        for (int i2 = 0; i2 < TR::d; i2++)
          for (int j2 = 0; j2 < TR::d; j2++) q[(size_t)i2 * TR::d + j2] = gp->gp.Qc(i2, j2);
        check(gpslam_hip_add_gp_priors_qc(h_, 1, &left, &gp->gp.delta_t, q.data()), "add_gp_priors_qc");   // its own Qc_model
      } else if (auto rg = boost::dynamic_pointer_cast<typename TR::Range>(f)) {
        // (an interpolated factor's Qc_model has no effect on its error or Jacobians: Qc cancels in Lambda and Psi)
        const int32_t left = chain_left(rg->key1(), rg->key2(), rg->key3(), rg->key4()), lm = lm_of(rg->key5());
        const std::vector<double> sg = detail::sigmas(rg->noiseModel());
        double sensor[12];
        if (rg->sensor) detail::pack(*rg->sensor, sensor);
        check(gpslam_hip_add_interp_range(h_, 1, &left, &lm, &rg->measured_value, sg.data(), &rg->gp.delta_t, &rg->gp.tau,
                                          rg->sensor ? sensor : nullptr), "add_interp_range");
      } else if (auto pr = boost::dynamic_pointer_cast<gtsam::PriorFactor<POSE>>(f)) {
        const int32_t idx = state_of(pr->key());
        double m[12];
        detail::pack(pr->prior(), m);
        const std::vector<double> sg = detail::sigmas(pr->noiseModel());
        check(gpslam_hip_add_pose_priors(h_, 1, &idx, m, sg.data()), "add_pose_priors");
      } else if (auto pv = boost::dynamic_pointer_cast<gtsam::PriorFactor<typename TR::Vel>>(f)) {
        const int32_t idx = state_of(pv->key());
        const typename TR::Vel m = pv->prior();
        const std::vector<double> sg = detail::sigmas(pv->noiseModel());
        check(gpslam_hip_add_vel_priors(h_, 1, &idx, m.data(), sg.data()), "add_vel_priors");
      } else if (auto pl = boost::dynamic_pointer_cast<gtsam::PriorFactor<typename TR::Point>>(f)) {
        const int32_t idx = lm_of(pl->key());
        const typename TR::Point m = pl->prior();
        double mm[3] = {0, 0, 0};
        for (int q = 0; q < TR::ld; q++) mm[q] = m(q);
        const std::vector<double> sg = detail::sigmas(pl->noiseModel());
        check(gpslam_hip_add_landmark_priors(h_, 1, &idx, mm, sg.data()), "add_landmark_priors");
      } else if (auto bt = boost::dynamic_pointer_cast<gtsam::BetweenFactor<POSE>>(f)) {
        // consecutive states: a chain factor; any other pair: a loop closure (round 6, gpslam_hip_add_between_pairs)
        const int32_t first = state_of(bt->key1()), second = state_of(bt->key2());
        double m[12];
        detail::pack(bt->measured(), m);
        const std::vector<double> sg = detail::sigmas(bt->noiseModel());
        check(gpslam_hip_add_between_pairs(h_, 1, &first, &second, m, sg.data()), "add_between_pairs");
      } else {
        throw std::invalid_argument("HipChainOptimizer: factor type not covered by the chain solver (keep it on stock GTSAM)");
      }
    }
    check(gpslam_hip_compile(h_), "compile");
    gpslam_hip_default_params(&p_);
    p_.use_lm = use_lm ? 1 : 0;
    p_.max_iterations = (int)params.maxIterations;
    p_.relative_error_tol = params.relativeErrorTol;
    p_.absolute_error_tol = params.absoluteErrorTol;
    p_.error_tol = params.errorTol;
    p_.lambda_initial = params.lambdaInitial;
    p_.lambda_factor = params.lambdaFactor;
    p_.lambda_upper_bound = params.lambdaUpperBound;
    p_.lambda_lower_bound = params.lambdaLowerBound;
    p_.min_model_fidelity = params.minModelFidelity;
    lambda_ = p_.lambda_initial;
    check(gpslam_hip_error(h_, &error_), "error");
  }
  HipChainOptimizerT(const HipChainOptimizerT &) = delete;
  HipChainOptimizerT &operator=(const HipChainOptimizerT &) = delete;

  double error() const { return error_; }
  double lambda() const { return lambda_; }
  int iterations() const { return iterations_; }
  /// one GaussNewtonOptimizer::iterate() / LevenbergMarquardtOptimizer::iterate()
  void iterate() {
    gpslam_hip_stats st;
    const int rc = p_.use_lm ? gpslam_hip_iterate_lm(h_, &lambda_, &p_, &st) : gpslam_hip_iterate_gn(h_, &st);
    if (rc == GPSLAM_E_NOT_SPD) throw gtsam::IndeterminantLinearSystemException(0);
    check(rc, "iterate");
    error_ = st.error_after;
    iterations_++;
  }
  /// NonlinearOptimizer::optimize() with GTSAM's stop rules
  const gtsam::Values &optimize() {
    gpslam_hip_stats st;
    const int rc = gpslam_hip_optimize(h_, &p_, &st);
    if (rc == GPSLAM_E_NOT_SPD) throw gtsam::IndeterminantLinearSystemException(0);
    check(rc, "optimize");
    error_ = st.error_after; iterations_ += st.iterations; lambda_ = st.lambda;
    return values();
  }
  const gtsam::Values &values() {
    const int N = (int)states_.size(), L = (int)lms_.size();
    std::vector<double> P((size_t)N * TR::pd), V((size_t)N * TR::d), LM((size_t)(L > 0 ? L : 1) * TR::ld);
    check(gpslam_hip_get_states(h_, P.data(), V.data()), "get_states");
    if (L > 0) check(gpslam_hip_get_landmarks(h_, LM.data()), "get_landmarks");
    out_ = init_;
    for (auto &kv : states_) {
      out_.update(gtsam::Symbol('x', kv.first), TR::un(&P[(size_t)kv.second * TR::pd]));
      const gtsam::Key vk = gtsam::Symbol('v', kv.first);
      if (out_.exists(vk)) {
        typename TR::Vel v;
        for (int q = 0; q < TR::d; q++) v(q) = V[(size_t)kv.second * TR::d + q];
        out_.update(vk, v);
      }
    }
    for (auto &kv : lms_) {
      typename TR::Point p;
      for (int q = 0; q < TR::ld; q++) p(q) = LM[(size_t)kv.second * TR::ld + q];
      out_.update(gtsam::Symbol('l', kv.first), p);
    }
    return out_;
  }

 private:
  void check(int rc, const char *what) const {
    if (rc < 0) throw std::runtime_error(std::string("HipChainOptimizer: ") + what + " failed: " + gpslam_hip_last_error(h_));
  }
  void set_qc(const gtsam::Matrix &Qc) {
    std::vector<double> q((size_t)TR::d * TR::d);
    for (int i = 0; i < TR::d; i++)
      for (int j = 0; j < TR::d; j++) q[(size_t)i * TR::d + j] = Qc(i, j);
    check(gpslam_hip_set_qc(h_, q.data()), "set_qc");
  }
  int32_t state_of(gtsam::Key k) const {
    auto it = states_.find(gtsam::Symbol(k).index());
    if (it == states_.end()) throw std::invalid_argument("HipChainOptimizer: factor refers to an unknown state");
    return (int32_t)it->second;
  }
  /// left state of a factor on (pose1, vel1, pose2, vel2): 'x' / 'v' symbols, matching indices, consecutive states
  int32_t chain_left(gtsam::Key pose1, gtsam::Key vel1, gtsam::Key pose2, gtsam::Key vel2) const {
    const gtsam::Symbol p1(pose1), v1(vel1), p2(pose2), v2(vel2);
    if (p1.chr() != 'x' || p2.chr() != 'x' || v1.chr() != 'v' || v2.chr() != 'v')
      throw std::invalid_argument("HipChainOptimizer: GP factors take keys (x_i, v_i, x_j, v_j) in this order");
    if (v1.index() != p1.index() || v2.index() != p2.index())
      throw std::invalid_argument("HipChainOptimizer: the velocity keys of a GP factor must carry the indices of its pose keys");
    const int32_t left = state_of(pose1);
    if (state_of(pose2) != left + 1) throw std::invalid_argument("HipChainOptimizer: GP factors must join consecutive states (chain structure)");
    return left;
  }
  int32_t lm_of(gtsam::Key k) const {
    auto it = lms_.find(gtsam::Symbol(k).index());
    if (it == lms_.end()) throw std::invalid_argument("HipChainOptimizer: factor refers to an unknown landmark");
    return (int32_t)it->second;
  }
  std::unique_ptr<gpslam_hip_handle, detail::HandleDeleter> hh_;   // owns the handle, also when the constructor throws
  gpslam_hip_handle *h_ = nullptr;
  gpslam_hip_params p_;
  std::map<uint64_t, int> states_, lms_;
  gtsam::Values init_, out_;
  double error_ = 0.0, lambda_ = 0.0;
  int iterations_ = 0;
};
typedef HipChainOptimizerT<gtsam::Pose3> HipChainOptimizerPose3;
typedef HipChainOptimizerT<gtsam::Pose2> HipChainOptimizerPose2;

}  // namespace gpslam_hip

#endif  // GPSLAM_HIP_HAVE_GTSAM
