// End-to-end tests of the C++ host classes (gpslam_amd/host/gpslam_host.hpp), written the way the reference's own
// CppUnitLite tests are: build a NonlinearFactorGraph with the gpslam factor classes, optimise with
// GaussNewtonOptimizer / LevenbergMarquardtOptimizer, compare with the ground truth.  Scenarios and numbers are those
// of gpslam/gp/tests/testGaussianProcessPrior{Pose3,Pose2,Rot3,Linear}.cpp (Optimization),
// gpslam/slam/tests/testGPInterpolatedRangeFactorPose{2,3}.cpp (optimization) and
// gpslam/slam/tests/testRangeBearingFactor2DLinear.cpp (optimization), gpslam/gp/tests/testGaussianProcessPriorPose3VW.cpp
// (Optimization) and gpslam/slam/tests/testGPInterpolatedProjectionFactorPose3.cpp (optimization).
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <vector>
#include <stdexcept>

#include "../../gpslam_amd/host/gpslam_host.hpp"

using namespace gtsam;
using namespace gpslam;

static int failures = 0;
#define EXPECT(cond)                                                                  \
  do {                                                                                \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
  } while (0)
#define EXPECT_NEAR(a, b, tol) EXPECT(std::fabs((a) - (b)) <= (tol))

static bool near3(const Point3 &a, const Point3 &b, double tol) {
  return std::fabs(a.x - b.x) <= tol && std::fabs(a.y - b.y) <= tol && std::fabs(a.z - b.z) <= tol;
}
static bool nearR(const Rot3 &a, const Rot3 &b, double tol) {
  for (int i = 0; i < 9; i++) if (std::fabs(a.R[i] - b.R[i]) > tol) return false;
  return true;
}
template <int N> static bool nearV(const VectorN<N> &a, const VectorN<N> &b, double tol) {
  for (int i = 0; i < N; i++) if (std::fabs(a[i] - b[i]) > tol) return false;
  return true;
}

static void test_gp_prior_pose3_optimization() {
  auto model_prior = noiseModel::Isotropic::Sigma(6, 0.001);
  double delta_t = 1;
  Matrix Qc = 0.01 * Matrix::Identity(6);
  auto Qc_model = noiseModel::Gaussian::Covariance(Qc);
  Pose3 pose1(Rot3(), Point3(0, 0, 0)), pose2(Rot3(), Point3(1, 0, 0));
  Vector6 v1 = {0, 0, 0, 1, 0, 0};
  Vector6 v2 = {0.1, 0.2, -0.3, 2.0, -0.5, 0.6};
  NonlinearFactorGraph graph;
  graph.add(PriorFactor<Pose3>(Symbol('x', 1), pose1, model_prior));
  graph.add(PriorFactor<Pose3>(Symbol('x', 2), pose2, model_prior));
  graph.add(GaussianProcessPriorPose3(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), delta_t, Qc_model));
  Values init_values;
  init_values.insert(Symbol('x', 1), pose1);
  init_values.insert(Symbol('v', 1), v1);
  init_values.insert(Symbol('x', 2), pose2);
  init_values.insert(Symbol('v', 2), v2);
  GaussNewtonParams parameters;
  GaussNewtonOptimizer optimizer(graph, init_values, parameters);
  optimizer.optimize();
  Values values = optimizer.values();
  EXPECT_NEAR(0, graph.error(values), 1e-6);
  EXPECT(near3(pose1.t, values.at<Pose3>(Symbol('x', 1)).t, 1e-6) && nearR(pose1.R, values.at<Pose3>(Symbol('x', 1)).R, 1e-6));
  EXPECT(near3(pose2.t, values.at<Pose3>(Symbol('x', 2)).t, 1e-6) && nearR(pose2.R, values.at<Pose3>(Symbol('x', 2)).R, 1e-6));
  EXPECT(nearV(v1, values.at<Vector6>(Symbol('v', 1)), 1e-6));
  EXPECT(nearV(v1, values.at<Vector6>(Symbol('v', 2)), 1e-6));
  EXPECT(optimizer.iterations() > 0 && optimizer.iterations() < 100);
}

static void test_gp_prior_pose2_rot3_linear_optimization() {
  {
    auto model_prior = noiseModel::Isotropic::Sigma(3, 0.001);
    auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(3));
    Pose2 pose1(0, 0, 0), pose2(1, 0, 0);
    Vector3 v1 = {1, 0, 0}, v2 = {2.0, -0.5, 0.6};
    NonlinearFactorGraph graph;
    graph.add(PriorFactor<Pose2>(Symbol('x', 1), pose1, model_prior));
    graph.add(PriorFactor<Pose2>(Symbol('x', 2), pose2, model_prior));
    graph.add(GaussianProcessPriorPose2(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 1.0, Qc_model));
    Values init;
    init.insert(Symbol('x', 1), pose1); init.insert(Symbol('v', 1), v1); init.insert(Symbol('x', 2), pose2); init.insert(Symbol('v', 2), v2);
    GaussNewtonOptimizer optimizer(graph, init);
    Values values = optimizer.optimize();
    EXPECT_NEAR(0, graph.error(values), 1e-6);
    EXPECT_NEAR(values.at<Pose2>(Symbol('x', 2)).x, 1.0, 1e-6);
    EXPECT(nearV(v1, values.at<Vector3>(Symbol('v', 2)), 1e-6));
  }
  {
    auto model_prior = noiseModel::Isotropic::Sigma(3, 0.001);
    auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(3));
    Rot3 pose1, pose2(Rot3::Ypr(0, 0, 0.1));
    Vector3 v1 = {1, 0, 0}, v2 = {2.0, -0.5, 0.6};
    NonlinearFactorGraph graph;
    graph.add(PriorFactor<Rot3>(Symbol('x', 1), pose1, model_prior));
    graph.add(PriorFactor<Rot3>(Symbol('x', 2), pose2, model_prior));
    graph.add(GaussianProcessPriorRot3(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 0.1, Qc_model));
    Values init;
    init.insert(Symbol('x', 1), pose1); init.insert(Symbol('v', 1), v1); init.insert(Symbol('x', 2), pose2); init.insert(Symbol('v', 2), v2);
    GaussNewtonOptimizer optimizer(graph, init);
    Values values = optimizer.optimize();
    EXPECT_NEAR(0, graph.error(values), 1e-6);
    EXPECT(nearR(pose2, values.at<Rot3>(Symbol('x', 2)), 1e-6));
    EXPECT(nearV(v1, values.at<Vector3>(Symbol('v', 1)), 1e-6) && nearV(v1, values.at<Vector3>(Symbol('v', 2)), 1e-6));
  }
  {
    auto model_prior = noiseModel::Isotropic::Sigma(3, 0.001);
    auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(3));
    Vector3 p1 = {1, 0, 0}, p2 = {1.1, 0, 0}, v1 = {1, 0, 0}, v1init = {1, 0.1, 0.2}, v2init = {2.1, -1.2, 0.9};
    NonlinearFactorGraph graph;
    graph.add(PriorFactor<Vector3>(Symbol('x', 1), p1, model_prior));
    graph.add(PriorFactor<Vector3>(Symbol('x', 2), p2, model_prior));
    graph.add(GaussianProcessPriorLinear<3>(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 0.1, Qc_model));
    Values init;
    init.insert(Symbol('x', 1), p1); init.insert(Symbol('v', 1), v1init); init.insert(Symbol('x', 2), p2); init.insert(Symbol('v', 2), v2init);
    LevenbergMarquardtOptimizer optimizer(graph, init);   // the linear problem through the LM path
    Values values = optimizer.optimize();
    EXPECT_NEAR(0, graph.error(values), 1e-6);
    EXPECT(nearV(v1, values.at<Vector3>(Symbol('v', 1)), 1e-5) && nearV(v1, values.at<Vector3>(Symbol('v', 2)), 1e-5));
  }
}

static void test_interp_range_pose2_optimization() {
  auto model_prior = noiseModel::Isotropic::Sigma(3, 0.01);
  auto model_prior2_loss = noiseModel::Isotropic::Sigma(2, 0.1);
  auto model_cam = noiseModel::Isotropic::Sigma(1, 0.1);
  double delta_t = 0.5, tau1 = 0.05, tau2 = 0.25, tau3 = 0.45;
  auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(3));
  Pose2 p1(0, 0, 0), p2(5, 0, 0), pcam1(0.5, 0, 0), pcam2(2.5, 0, 0), pcam3(4.5, 0, 0);
  Vector3 v1 = {10, 0, 0}, v2 = {10, 0, 0};
  Pose2 p1i(0.1, 0.1, -0.1), p2i(5.1, -0.1, 0.1);
  Vector3 v1i = {9.8, 0, 0.2}, v2i = {10.2, 0, -0.1};
  Point2 land(2.4, 3.2), landi(2.3, 3.1);
  double meas1 = pcam1.range(land), meas2 = pcam2.range(land), meas3 = pcam3.range(land);
  NonlinearFactorGraph graph;
  graph.add(PriorFactor<Pose2>(Symbol('x', 1), p1, model_prior));
  graph.add(PriorFactor<Pose2>(Symbol('x', 2), p2, model_prior));
  graph.add(PriorFactor<Point2>(Symbol('l', 1), land, model_prior2_loss));
  graph.add(PriorFactor<Vector3>(Symbol('v', 1), v1, model_prior));
  graph.add(PriorFactor<Vector3>(Symbol('v', 2), v2, model_prior));
  graph.add(GaussianProcessPriorPose2(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), delta_t, Qc_model));
  graph.add(GPInterpolatedRangeFactorPose2(meas1, model_cam, Qc_model, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), delta_t, tau1));
  graph.add(GPInterpolatedRangeFactorPose2(meas2, model_cam, Qc_model, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), delta_t, tau2));
  graph.add(GPInterpolatedRangeFactorPose2(meas3, model_cam, Qc_model, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), delta_t, tau3));
  Values init_values;
  init_values.insert(Symbol('x', 1), p1i); init_values.insert(Symbol('v', 1), v1i);
  init_values.insert(Symbol('x', 2), p2i); init_values.insert(Symbol('v', 2), v2i);
  init_values.insert(Symbol('l', 1), landi);
  GaussNewtonParams parameters;
  parameters.setVerbosity("ERROR");
  GaussNewtonOptimizer optimizer(graph, init_values, parameters);
  optimizer.optimize();
  Values values = optimizer.values();
  EXPECT_NEAR(0, graph.error(values), 1e-4);
  EXPECT_NEAR(values.at<Pose2>(Symbol('x', 1)).x, 0.0, 1e-4);
  EXPECT_NEAR(values.at<Pose2>(Symbol('x', 2)).x, 5.0, 1e-4);
  EXPECT(nearV(v1, values.at<Vector3>(Symbol('v', 1)), 1e-4) && nearV(v2, values.at<Vector3>(Symbol('v', 2)), 1e-4));
  EXPECT_NEAR(values.at<Point2>(Symbol('l', 1)).x, 2.4, 1e-4);
  EXPECT_NEAR(values.at<Point2>(Symbol('l', 1)).y, 3.2, 1e-4);
}

static void test_interp_range_pose3_optimization_with_extrapolation() {
  auto model_prior = noiseModel::Isotropic::Sigma(6, 0.01);
  auto model_prior3_loss = noiseModel::Isotropic::Sigma(3, 0.1);
  auto model_cam = noiseModel::Isotropic::Sigma(1, 0.1);
  double delta_t = 0.1, tau1 = -0.1, tau2 = 0.05, tau3 = 0.2;    // tau outside [0, delta_t] on both sides
  auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(6));
  Pose3 p1(Rot3(), Point3(0, 0, 0)), p2(Rot3(), Point3(1, 0, 0));
  Pose3 pcam1(Rot3(), Point3(-1, 0, 0)), pcam2(Rot3(), Point3(0.5, 0, 0)), pcam3(Rot3(), Point3(2, 0, 0));
  Vector6 v1 = {0, 0, 0, 10, 0, 0}, v2 = {0, 0, 0, 10, 0, 0};
  Pose3 p1i(Rot3::Ypr(0.1, 0.2, 0.4), Point3(0.2, 0.3, -0.2)), p2i(Rot3::Ypr(-0.1, -0.2, -0.4), Point3(1.2, -0.3, 0.2));
  Vector6 v1i = {-0.1, 0, 0, 0.8, 0, 0.2}, v2i = {0, 0, 0.2, 1.2, 0, -0.1};
  Point3 land(0.4, 1.2, 3), landi(0.3, 1.1, 2.9);
  NonlinearFactorGraph graph;
  graph.add(PriorFactor<Pose3>(Symbol('x', 1), p1, model_prior));
  graph.add(PriorFactor<Pose3>(Symbol('x', 2), p2, model_prior));
  graph.add(PriorFactor<Point3>(Symbol('l', 1), land, model_prior3_loss));
  graph.add(PriorFactor<Vector6>(Symbol('v', 1), v1, model_prior));
  graph.add(PriorFactor<Vector6>(Symbol('v', 2), v2, model_prior));
  graph.add(GaussianProcessPriorPose3(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), delta_t, Qc_model));
  const double taus[3] = {tau1, tau2, tau3};
  const Pose3 cams[3] = {pcam1, pcam2, pcam3};
  for (int k = 0; k < 3; k++)
    graph.add(GPInterpolatedRangeFactorPose3(cams[k].range(land), model_cam, Qc_model, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2),
                                             Symbol('v', 2), Symbol('l', 1), delta_t, taus[k]));
  Values init_values;
  init_values.insert(Symbol('x', 1), p1i); init_values.insert(Symbol('v', 1), v1i);
  init_values.insert(Symbol('x', 2), p2i); init_values.insert(Symbol('v', 2), v2i);
  init_values.insert(Symbol('l', 1), landi);
  GaussNewtonOptimizer optimizer(graph, init_values);
  optimizer.optimize();
  Values values = optimizer.values();
  EXPECT_NEAR(0, graph.error(values), 1e-6);
  EXPECT(near3(p2.t, values.at<Pose3>(Symbol('x', 2)).t, 1e-6) && nearR(p2.R, values.at<Pose3>(Symbol('x', 2)).R, 1e-6));
  EXPECT(nearV(v1, values.at<Vector6>(Symbol('v', 1)), 1e-6) && nearV(v2, values.at<Vector6>(Symbol('v', 2)), 1e-6));
  EXPECT(near3(land, values.at<Point3>(Symbol('l', 1)), 1e-6));
}

static void test_range_bearing_2dlinear_optimization() {
  // gpslam/slam/tests/testRangeBearingFactor2DLinear.cpp:86-132: three poses without velocities
  auto meas_model = noiseModel::Isotropic::Sigma(2, 0.1);
  auto prior_model = noiseModel::Isotropic::Sigma(3, 1.0);
  auto between_model = noiseModel::Isotropic::Sigma(3, 0.1);
  Key key_p1 = Symbol('x', 1), key_p2 = Symbol('x', 2), key_p3 = Symbol('x', 3), key_lnd = Symbol('l', 1);
  Vector3 p1 = {0, 0, 0}, p2 = {1, 0, 0}, p3 = {2, 1, 0};
  Point2 lnd(0, 2);
  double dist1 = 2.0, dist2 = std::sqrt(5.0), dist3 = std::sqrt(5.0);
  double bear1 = 1.570796326794897, bear2 = 2.034443935795703, bear3 = 2.677945044588987;
  Vector3 btw12 = {1, 0, 0}, btw23 = {1, 1, 0};
  NonlinearFactorGraph graph;
  graph.add(PriorFactor<Vector3>(key_p1, p1, prior_model));
  graph.add(BetweenFactor<Vector3>(key_p1, key_p2, btw12, between_model));
  graph.add(BetweenFactor<Vector3>(key_p2, key_p3, btw23, between_model));
  graph.add(RangeBearingFactor2DLinear(key_p1, key_lnd, dist1, bear1, meas_model));
  graph.add(RangeBearingFactor2DLinear(key_p2, key_lnd, dist2, bear2, meas_model));
  graph.add(RangeBearingFactor2DLinear(key_p3, key_lnd, dist3, bear3, meas_model));
  Values init_values;
  init_values.insert(key_p1, Vector3{0.2, -0.5, 0.3});
  init_values.insert(key_p2, Vector3{0.8, 0.2, 0.1});
  init_values.insert(key_p3, Vector3{2.4, 1.3, -0.4});
  init_values.insert(key_lnd, Point2(0.1, 2.2));
  GaussNewtonOptimizer optimizer(graph, init_values);
  optimizer.optimize();
  Values values = optimizer.values();
  EXPECT_NEAR(0, graph.error(values), 1e-6);
  EXPECT(nearV(p1, values.at<Vector3>(key_p1), 1e-6) && nearV(p2, values.at<Vector3>(key_p2), 1e-6) && nearV(p3, values.at<Vector3>(key_p3), 1e-6));
  EXPECT_NEAR(values.at<Point2>(key_lnd).x, 0.0, 1e-6);
  EXPECT_NEAR(values.at<Point2>(key_lnd).y, 2.0, 1e-6);
}

static void test_gp_prior_pose3vw_optimization() {
  // testGaussianProcessPriorPose3VW.cpp:142-188 (the reference only checks that the optimisation runs; the optimum is
  // the initial configuration: both poses fixed one metre apart, dt = 1, world velocity (1, 0, 0))
  auto model_prior = noiseModel::Isotropic::Sigma(6, 0.001);
  double delta_t = 1;
  auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(6));
  Pose3 pose1(Rot3(), Point3(0, 0, 0)), pose2(Rot3(), Point3(1, 0, 0));
  Vector3 v1 = {1, 0, 0}, w1 = {0, 0, 0}, v2 = {1, 0, 0}, w2 = {0, 0, 0};
  NonlinearFactorGraph graph;
  graph.add(PriorFactor<Pose3>(Symbol('x', 1), pose1, model_prior));
  graph.add(PriorFactor<Pose3>(Symbol('x', 2), pose2, model_prior));
  graph.add(GaussianProcessPriorPose3VW(Symbol('x', 1), Symbol('v', 1), Symbol('w', 1), Symbol('x', 2), Symbol('v', 2), Symbol('w', 2),
                                        delta_t, Qc_model));
  Values init_values;
  init_values.insert(Symbol('x', 1), pose1);
  init_values.insert(Symbol('v', 1), v1);
  init_values.insert(Symbol('w', 1), w1);
  init_values.insert(Symbol('x', 2), pose2);
  init_values.insert(Symbol('v', 2), Vector3{0.7, 0.2, -0.1});   // perturbed: the prior must pull it back to (1, 0, 0)
  init_values.insert(Symbol('w', 2), Vector3{0.05, -0.02, 0.03});
  GaussNewtonParams parameters;
  GaussNewtonOptimizer optimizer(graph, init_values, parameters);
  optimizer.optimize();
  Values values = optimizer.values();
  EXPECT_NEAR(0, graph.error(values), 1e-6);
  EXPECT(nearV(v1, values.at<Vector3>(Symbol('v', 1)), 1e-6) && nearV(w1, values.at<Vector3>(Symbol('w', 1)), 1e-6));
  EXPECT(nearV(v2, values.at<Vector3>(Symbol('v', 2)), 1e-6) && nearV(w2, values.at<Vector3>(Symbol('w', 2)), 1e-6));
  EXPECT(near3(pose2.t, values.at<Pose3>(Symbol('x', 2)).t, 1e-6));
}

// the class is a template over CALIBRATION in the reference (GPInterpolatedProjectionFactorPose3.h:29): the same scenario with
// gtsam::Cal3_S2 (the reference's test) and with a distorting gtsam::Cal3DS2 (round 4)
template <class CAL> static void projection_scenario(const std::shared_ptr<CAL> &K) {
  // testGPInterpolatedProjectionFactorPose3.cpp:180-262
  typedef GPInterpolatedProjectionFactorPose3<CAL> ProjectionFactor;
  auto model_prior = noiseModel::Isotropic::Sigma(6, 0.01);
  auto model_cam = noiseModel::Isotropic::Sigma(2, 0.1);
  double delta_t = 0.1, tau1 = 0.02, tau2 = 0.06, tau3 = 0.09;
  auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(6));
  Pose3 p1(Rot3(), Point3(0, 0, 0)), p2(Rot3(), Point3(1, 0, 0));
  Pose3 pcam1(Rot3(), Point3(0.2, 0, 0)), pcam2(Rot3(), Point3(0.6, 0, 0)), pcam3(Rot3(), Point3(0.9, 0, 0));
  Vector6 v1 = {0, 0, 0, 10, 0, 0}, v2 = {0, 0, 0, 10, 0, 0};
  Pose3 p1i(Rot3::Ypr(0.1, 0.2, 0.4), Point3(0.2, 0.3, -0.2));
  Pose3 p2i(Rot3::Ypr(-0.1, -0.2, -0.4), Point3(1.2, -0.3, 0.2));
  Vector6 v1i = {-0.3, 0, 0, 0.7, 0, 0.2}, v2i = {0, 0, 0.4, 1.2, 0, -0.1};
  Point3 land(3.4, 1.2, 20), landi(3.3, 1.3, 18);
  Point2 meas1 = PinholeCamera<CAL>(pcam1, *K).project(land);
  Point2 meas2 = PinholeCamera<CAL>(pcam2, *K).project(land);
  Point2 meas3 = PinholeCamera<CAL>(pcam3, *K).project(land);
  NonlinearFactorGraph graph;
  graph.add(PriorFactor<Pose3>(Symbol('x', 1), p1, model_prior));
  graph.add(PriorFactor<Pose3>(Symbol('x', 2), p2, model_prior));
  graph.add(GaussianProcessPriorPose3(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), delta_t, Qc_model));
  graph.add(ProjectionFactor(meas1, model_cam, Qc_model, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), delta_t, tau1, K));
  graph.add(ProjectionFactor(meas2, model_cam, Qc_model, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), delta_t, tau2, K));
  graph.add(ProjectionFactor(meas3, model_cam, Qc_model, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), delta_t, tau3, K));
  Values init_values;
  init_values.insert(Symbol('x', 1), p1i);
  init_values.insert(Symbol('v', 1), v1i);
  init_values.insert(Symbol('x', 2), p2i);
  init_values.insert(Symbol('v', 2), v2i);
  init_values.insert(Symbol('l', 1), landi);
  GaussNewtonParams parameters;
  GaussNewtonOptimizer optimizer(graph, init_values, parameters);
  optimizer.optimize();
  Values values = optimizer.values();
  EXPECT_NEAR(0, graph.error(values), 1e-6);
  EXPECT(near3(p1.t, values.at<Pose3>(Symbol('x', 1)).t, 1e-6) && nearR(p1.R, values.at<Pose3>(Symbol('x', 1)).R, 1e-6));
  EXPECT(near3(p2.t, values.at<Pose3>(Symbol('x', 2)).t, 1e-6) && nearR(p2.R, values.at<Pose3>(Symbol('x', 2)).R, 1e-6));
  EXPECT(nearV(v1, values.at<Vector6>(Symbol('v', 1)), 1e-6));
  EXPECT(nearV(v2, values.at<Vector6>(Symbol('v', 2)), 1e-6));
  EXPECT(near3(land, values.at<Point3>(Symbol('l', 1)), 1e-6));
  // dense trajectory output: the camera poses the measurements were generated from (GaussianProcessInterpolatorPose3,
  // gpslam.h:72-77, as one batched query)
  std::vector<Pose3> q = optimizer.interpolatePoses<Pose3>({Symbol('x', 1), Symbol('x', 1), Symbol('x', 1)},
                                                           {delta_t, delta_t, delta_t}, {tau1, tau2, tau3});
  EXPECT(q.size() == 3 && near3(pcam1.t, q[0].t, 1e-6) && near3(pcam2.t, q[1].t, 1e-6) && near3(pcam3.t, q[2].t, 1e-6));
}
static void test_projection_optimization_and_trajectory_query() {
  projection_scenario(std::make_shared<Cal3_S2>(50, 50, 0, 40, 30));
  // (mild coefficients: the scenario starts 0.4 rad away from the truth, and a strongly non-monotonic radial polynomial gives
  //  Gauss-Newton local minima to find there)
  auto Kd = std::make_shared<Cal3DS2>(50, 50, 0, 40, 30, 0.05, -0.01, 0.002, -0.003);
  projection_scenario(Kd);
  // the distortion is visible off axis (a dropped term would not be caught by a zero-residual fixed point alone)
  Pose3 cam(Rot3(), Point3(0.2, 0, 0));
  Point2 a = PinholeCamera<Cal3_S2>(cam, Cal3_S2(50, 50, 0, 40, 30)).project(Point3(10.0, -6.0, 20));
  Point2 b = PinholeCamera<Cal3DS2>(cam, *Kd).project(Point3(10.0, -6.0, 20));
  EXPECT(std::fabs(a.x - b.x) > 1e-2 && std::fabs(a.y - b.y) > 1e-2);
}

// evaluateError / interpolatePose of single factors, the calls the reference's factor tests make
// (gpslam/gp/tests/testGaussianProcessPriorPose3.cpp:43-60, testGaussianProcessPriorLinear.cpp:45-50,
// gpslam/slam/tests/testGPInterpolatedRangeFactorPose2.cpp:55-64, gpslam/gp/tests/testGaussianProcessInterpolatorPose3.cpp:33-39)
static void test_evaluate_error_and_interpolators() {
  {   // GaussianProcessPriorPose3: constant forward velocity 1 m/s, dt = 1 -> zero error (testGaussianProcessPriorPose3.cpp:69-74)
    auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(6));
    GaussianProcessPriorPose3 factor(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 1.0, Qc_model);
    Pose3 p1, p2(Rot3(), Point3(1, 0, 0));
    Vector6 v = {0, 0, 0, 1, 0, 0};
    Matrix H1, H2, H3, H4;
    Vector e = factor.evaluateError(p1, v, p2, v, &H1, &H2, &H3, &H4);
    EXPECT(e.size() == 12);
    for (double x : e) EXPECT_NEAR(0.0, x, 1e-12);
    EXPECT(H1.rows == 12 && H1.cols == 6 && H4.rows == 12 && H4.cols == 6);
    for (int i = 0; i < 6; i++) { EXPECT_NEAR(-1.0, H2(i, i), 1e-12); EXPECT_NEAR(-1.0, H2(6 + i, i), 1e-12); }   // H2 = [-dt I; -I]
    for (int i = 0; i < 6; i++) EXPECT_NEAR(1.0, H3(i, i), 1e-9);                                                 // identity relative pose: Hlog = I
    // H1 numerically: perturb the first pose's translation along body x (first-order: t + R dx)
    const double h = 1e-6;
    Pose3 pp(Rot3(), Point3(h, 0, 0)), pm(Rot3(), Point3(-h, 0, 0));
    Vector ep = factor.evaluateError(pp, v, p2, v), em = factor.evaluateError(pm, v, p2, v);
    for (int r = 0; r < 12; r++) EXPECT_NEAR((ep[r] - em[r]) / (2 * h), H1(r, 3), 1e-6);
    // the reference's own spelling of the same call (testGaussianProcessPriorPose3.cpp:49-60: Matrix lvalues into
    // boost::optional<Matrix&> parameters, boost::none for what is not wanted): compiles unchanged (round 6)
    Matrix G1, G2, G3, G4;
    Vector e2 = factor.evaluateError(p1, v, p2, v, G1, G2, G3, G4);
    EXPECT(e2 == e && G1.a == H1.a && G2.a == H2.a && G3.a == H3.a && G4.a == H4.a);
    Matrix K2;
    Vector e3 = factor.evaluateError(p1, v, p2, v, boost::none, K2);
    EXPECT(e3 == e && K2.a == H2.a);
    boost::optional<Matrix &> none_of_them = boost::none, third(G3);
    G3 = Matrix();
    Vector e4 = factor.evaluateError(p1, v, p2, v, none_of_them, boost::none, third, boost::none);
    EXPECT(e4 == e && G3.a == H3.a);
  }
  {   // GaussianProcessPriorLinear<3>: e = [p1 + dt v1 - p2; v1 - v2], H1 = [I; 0], H2 = [dt I; I], H3 = [-I; 0], H4 = [0; -I]
    auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(3));
    GaussianProcessPriorLinear<3> factor(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 0.1, Qc_model);
    Vector3 p1 = {1, 2, 3}, v1 = {0.5, -0.5, 1}, p2 = {1.2, 1.7, 3.3}, v2 = {0.4, -0.2, 0.9};
    Matrix H1, H2, H3, H4;
    Vector e = factor.evaluateError(p1, v1, p2, v2, &H1, &H2, &H3, &H4);
    for (int i = 0; i < 3; i++) {
      EXPECT_NEAR(p1[i] + 0.1 * v1[i] - p2[i], e[i], 1e-14);
      EXPECT_NEAR(v1[i] - v2[i], e[3 + i], 1e-14);
      EXPECT_NEAR(1.0, H1(i, i), 0); EXPECT_NEAR(0.1, H2(i, i), 0); EXPECT_NEAR(1.0, H2(3 + i, i), 0);
      EXPECT_NEAR(-1.0, H3(i, i), 0); EXPECT_NEAR(-1.0, H4(3 + i, i), 0); EXPECT_NEAR(0.0, H4(i, i), 0);
    }
  }
  {   // GPInterpolatedRangeFactorPose2: H5 against central differences in the landmark (a vector space)
    auto Qc_model = noiseModel::Gaussian::Covariance(0.001 * Matrix::Identity(3));
    auto meas_model = noiseModel::Isotropic::Sigma(1, 0.1);
    Pose2 sensor(0.2, -0.1, 0.3);
    GPInterpolatedRangeFactorPose2 factor(4.5, meas_model, Qc_model, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2),
                                          Symbol('l', 1), 0.1, 0.04, &sensor);
    Pose2 p1(0, 0, 0.1), p2(0.1, 0.02, 0.15);
    Vector3 v1 = {1, 0.1, 0.3}, v2 = {1.1, 0.2, 0.4};
    Point2 land(3.0, 2.5);
    Matrix H1, H5;
    Vector e = factor.evaluateError(p1, v1, p2, v2, land, &H1, nullptr, nullptr, nullptr, &H5);
    EXPECT(e.size() == 1 && H5.rows == 1 && H5.cols == 2 && H1.rows == 1 && H1.cols == 3);
    const double h = 1e-6;
    const double dx = (factor.evaluateError(p1, v1, p2, v2, Point2(land.x + h, land.y))[0] - factor.evaluateError(p1, v1, p2, v2, Point2(land.x - h, land.y))[0]) / (2 * h);
    const double dy = (factor.evaluateError(p1, v1, p2, v2, Point2(land.x, land.y + h))[0] - factor.evaluateError(p1, v1, p2, v2, Point2(land.x, land.y - h))[0]) / (2 * h);
    EXPECT_NEAR(dx, H5(0, 0), 1e-7);
    EXPECT_NEAR(dy, H5(0, 1), 1e-7);
    EXPECT_NEAR(1.0, H5(0, 0) * H5(0, 0) + H5(0, 1) * H5(0, 1), 1e-12);   // d range / d point is a unit vector
  }
  {   // interpolators: tau = 0 reproduces the first state with H1 = I, H2..H4 = 0; the midpoint of a constant-velocity
      // motion is the halfway pose (testGaussianProcessInterpolatorPose3.cpp:60-66 scenario)
    auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(6));
    Pose3 p1, p2(Rot3(), Point3(0.1, 0, 0));
    Vector6 v = {0, 0, 0, 1, 0, 0};
    GaussianProcessInterpolatorPose3 at0(Qc_model, 0.1, 0.0), mid(Qc_model, 0.1, 0.05);
    Matrix H1, H2, H3, H4;
    Pose3 q0 = at0.interpolatePose(p1, v, p2, v, &H1, &H2, &H3, &H4);
    EXPECT(near3(p1.t, q0.t, 1e-12) && nearR(p1.R, q0.R, 1e-12));
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { EXPECT_NEAR(i == j ? 1.0 : 0.0, H1(i, j), 1e-9); EXPECT_NEAR(0.0, H3(i, j), 1e-9); EXPECT_NEAR(0.0, H4(i, j), 1e-9); }
    Pose3 qm = mid.interpolatePose(p1, v, p2, v);
    EXPECT(near3(Point3(0.05, 0, 0), qm.t, 1e-12) && nearR(p1.R, qm.R, 1e-12));
    auto Qc3 = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(3));
    GaussianProcessInterpolatorLinear<3> lin(Qc3, 0.1, 0.03);
    Vector3 a = {0, 0, 0}, va = {1, 2, 3}, b = {0.1, 0.2, 0.3};
    Vector3 ql = lin.interpolatePose(a, va, b, va);
    EXPECT_NEAR(0.03, ql[0], 1e-12); EXPECT_NEAR(0.06, ql[1], 1e-12); EXPECT_NEAR(0.09, ql[2], 1e-12);
    GaussianProcessInterpolatorPose2 i2(Qc3, 0.1, 0.05);
    Pose2 r2 = i2.interpolatePose(Pose2(0, 0, 0), Vector3{1, 0, 0}, Pose2(0.1, 0, 0), Vector3{1, 0, 0});
    EXPECT_NEAR(0.05, r2.x, 1e-12); EXPECT_NEAR(0.0, r2.y, 1e-12); EXPECT_NEAR(0.0, r2.theta, 1e-12);
    GaussianProcessInterpolatorRot3 i3(Qc3, 0.1, 0.05);
    Rot3 r3 = i3.interpolatePose(Rot3(), Vector3{0, 0, 1}, Rot3::Ypr(0.1, 0, 0), Vector3{0, 0, 1});
    EXPECT(nearR(Rot3::Ypr(0.05, 0, 0), r3, 1e-10));
  }
}

static void test_error_conventions() {
  auto model = noiseModel::Isotropic::Sigma(3, 0.1);
  auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(3));
  Values v;
  v.insert(Symbol('x', 1), Pose2(0, 0, 0)); v.insert(Symbol('v', 1), Vector3{0, 0, 0});
  v.insert(Symbol('x', 2), Pose2(1, 0, 0)); v.insert(Symbol('v', 2), Vector3{0, 0, 0});
  v.insert(Symbol('x', 3), Pose2(2, 0, 0)); v.insert(Symbol('v', 3), Vector3{0, 0, 0});
  bool threw = false;
  try { (void)v.at<Pose2>(Symbol('x', 9)); } catch (const std::out_of_range &) { threw = true; }   // ValuesKeyDoesNotExist
  EXPECT(threw);
  NonlinearFactorGraph g;
  g.add(GaussianProcessPriorPose2(Symbol('x', 1), Symbol('v', 1), Symbol('x', 3), Symbol('v', 3), 0.1, Qc_model));   // skips a state
  threw = false;
  try { GaussNewtonOptimizer o(g, v); } catch (const std::invalid_argument &) { threw = true; }
  EXPECT(threw);
  NonlinearFactorGraph g2;   // velocity of x1..x3 unconstrained and no priors: indeterminate system
  g2.add(PriorFactor<Pose2>(Symbol('x', 1), Pose2(0, 0, 0), model));
  threw = false;
  try { GaussNewtonOptimizer o(g2, v); o.iterate(); } catch (const std::runtime_error &) { threw = true; }   // IndeterminantLinearSystemException
  EXPECT(threw);
}

// matlab/GPAHRSexample.m in miniature, built with the GTSAM-named classes (PriorFactor<Rot3>, PriorFactor<Vector3> on a 'b'
// key, AHRSFactor + PreintegratedAhrsMeasurements, BetweenFactor<Vector3>, GaussianProcessPriorRot3, Rot3AttitudeFactor,
// GPInterpolatedAttitudeFactorRot3) and, with the same numbers, through the C ABI directly: identical results.
static Rot3 rot_exp(double wx, double wy, double wz) {
  const double th = std::sqrt(wx * wx + wy * wy + wz * wz);
  Rot3 r;
  if (th < 1e-12) return r;
  const double k[3] = {wx / th, wy / th, wz / th}, s = std::sin(th), c = 1.0 - std::cos(th);
  const double K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double kk = 0.0;
      for (int q = 0; q < 3; q++) kk += K[3 * i + q] * K[3 * q + j];
      r.R[3 * i + j] = (i == j ? 1.0 : 0.0) + s * K[3 * i + j] + c * kk;
    }
  return r;
}
static void test_ahrs_graph_with_bias_states() {
  const int N = 24;
  const double dt = 0.01, w[3] = {0.3, -0.2, 0.5}, bt[3] = {0.004, -0.003, 0.002};
  auto Qc_model = noiseModel::Gaussian::Covariance(1e4 * Matrix::Identity(3));
  auto acc_model = noiseModel::Isotropic::Sigma(2, 0.1);
  Matrix gyro_cov = 1e-3 * Matrix::Identity(3);
  NonlinearFactorGraph graph;
  Values init;
  std::vector<Rot3> truth;
  for (int k = 0; k < N; k++) truth.push_back(rot_exp(w[0] * k * dt, w[1] * k * dt, w[2] * k * dt));
  auto gravity_in_body = [&](const Rot3 &R) { return Unit3(R.R[6], R.R[7], R.R[8]); };   // R^T (0, 0, 1)
  graph.add(PriorFactor<Rot3>(Symbol('x', 1), truth[0], noiseModel::Isotropic::Sigma(3, 0.1)));
  graph.add(PriorFactor<Vector3>(Symbol('b', 1), Vector3{0, 0, 0}, noiseModel::Isotropic::Sigma(3, 1e-2)));
  // the same factors as plain arrays for the C ABI
  std::vector<int32_t> left;
  std::vector<double> dR, D, bh, tij, cov, gpdt, att_nz, att_b, att_sig, att_dt, att_tau;
  std::vector<int32_t> att_left;
  for (int k = 1; k < N; k++) {
    PreintegratedAhrsMeasurements pim(Vector3{0, 0, 0}, gyro_cov);
    for (int q = 0; q < 2; q++) pim.integrateMeasurement(Vector3{w[0] + bt[0], w[1] + bt[1], w[2] + bt[2]}, dt / 2);
    graph.add(AHRSFactor(Symbol('x', k), Symbol('x', k + 1), Symbol('b', k), pim));
    graph.add(BetweenFactor<Vector3>(Symbol('b', k), Symbol('b', k + 1), Vector3{0, 0, 0}, noiseModel::Isotropic::Sigma(3, 1e-4)));
    graph.add(GaussianProcessPriorRot3(Symbol('x', k), Symbol('v', k), Symbol('x', k + 1), Symbol('v', k + 1), dt, Qc_model));
    const std::vector<double> m = pim.packed();
    left.push_back(k - 1);
    dR.insert(dR.end(), m.begin(), m.begin() + 9); D.insert(D.end(), m.begin() + 9, m.begin() + 18);
    bh.insert(bh.end(), m.begin() + 18, m.begin() + 21); tij.push_back(m[21]); cov.insert(cov.end(), m.begin() + 22, m.end());
    gpdt.push_back(dt);
    const Unit3 g = gravity_in_body(truth[k]);
    double tau = 1.0, d_t = 1.0;   // Rot3AttitudeFactor on x_{k+1} (the host maps it to tau = delta_t = 1)
    if (k % 3 == 0) {                // every third one sits inside the interval instead
      const Rot3 mid = rot_exp(w[0] * (k - 0.6) * dt, w[1] * (k - 0.6) * dt, w[2] * (k - 0.6) * dt);
      const Unit3 gm = gravity_in_body(mid);
      graph.add(GPInterpolatedAttitudeFactorRot3(Symbol('x', k), Symbol('v', k), Symbol('x', k + 1), Symbol('v', k + 1), dt, 0.4 * dt, Qc_model,
                                                 acc_model, Unit3(0, 0, 1), gm));
      att_b.insert(att_b.end(), gm.p, gm.p + 3); tau = 0.4 * dt; d_t = dt;
    } else {
      graph.add(Rot3AttitudeFactor(Symbol('x', k + 1), Unit3(0, 0, 1), acc_model, g));
      att_b.insert(att_b.end(), g.p, g.p + 3);
    }
    att_left.push_back(k - 1); att_nz.insert(att_nz.end(), {0.0, 0.0, 1.0}); att_sig.insert(att_sig.end(), {0.1, 0.1});
    att_dt.push_back(d_t); att_tau.push_back(tau);
  }
  for (int k = 1; k <= N; k++) {
    init.insert(Symbol('x', k), Rot3());
    init.insert(Symbol('b', k), Vector3{0, 0, 0});
    init.insert(Symbol('v', k), Vector3{0, 0, 0});
  }
  LevenbergMarquardtOptimizer opt(graph, init);
  const double e0 = opt.error();
  opt.optimize();
  const Values res = opt.values();
  EXPECT(opt.error() < 1e-3 * e0 && opt.iterations() > 1 && opt.iterations() < 50);
  for (int k = 1; k <= N; k += 5) EXPECT(nearR(truth[k - 1], res.at<Rot3>(Symbol('x', k)), 0.02));
  EXPECT(nearV(Vector3{w[0], w[1], w[2]}, res.at<Vector3>(Symbol('v', N / 2)), 0.05));
  EXPECT(std::fabs(res.at<Vector3>(Symbol('b', N))[0]) < 1e-2);

  // ---- the same problem through the C ABI
  gpslam_hip_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.manifold = GPSLAM_ROT3_BIAS; cfg.nranks = 1;
  gpslam_hip_handle *h = nullptr;
  EXPECT(gpslam_hip_create(&cfg, &h) == 0);
  std::vector<double> P((size_t)N * 12, 0.0), V((size_t)N * 6, 0.0);
  for (int k = 0; k < N; k++) P[12 * k] = P[12 * k + 4] = P[12 * k + 8] = 1.0;
  EXPECT(gpslam_hip_set_states(h, N, P.data(), V.data()) == 0);
  const double Qc3[9] = {1e4, 0, 0, 0, 1e4, 0, 0, 0, 1e4};
  EXPECT(gpslam_hip_set_qc(h, Qc3) == 0);
  const double inf = INFINITY;
  int32_t zero = 0;
  {
    double m[12], sg[6] = {0.1, 0.1, 0.1, inf, inf, inf};
    std::memcpy(m, truth[0].R, sizeof(double) * 9); m[9] = m[10] = m[11] = 0.0;
    EXPECT(gpslam_hip_add_pose_priors(h, 1, &zero, m, sg) == 0);
    double mb[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, sb[6] = {inf, inf, inf, 1e-2, 1e-2, 1e-2};
    EXPECT(gpslam_hip_add_pose_priors(h, 1, &zero, mb, sb) == 0);
  }
  for (int k = 0; k < N - 1; k++) {
    double mb[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, sb[6] = {inf, inf, inf, 1e-4, 1e-4, 1e-4};
    int32_t l = k;
    EXPECT(gpslam_hip_add_between(h, 1, &l, mb, sb) == 0);
  }
  EXPECT(gpslam_hip_add_ahrs(h, N - 1, left.data(), dR.data(), D.data(), bh.data(), tij.data(), cov.data(), nullptr) == 0);
  EXPECT(gpslam_hip_add_gp_priors(h, N - 1, left.data(), gpdt.data()) == 0);
  EXPECT(gpslam_hip_add_interp_attitude(h, N - 1, att_left.data(), att_nz.data(), att_b.data(), att_sig.data(), att_dt.data(), att_tau.data()) == 0);
  EXPECT(gpslam_hip_compile(h) == 0);
  double e_abi = 0.0;
  EXPECT(gpslam_hip_error(h, &e_abi) == 0);
  EXPECT_NEAR(e_abi, e0, 1e-9 * e0);
  gpslam_hip_params prm;
  gpslam_hip_default_params(&prm);
  prm.use_lm = 1;
  gpslam_hip_stats st;
  EXPECT(gpslam_hip_optimize(h, &prm, &st) >= 0);
  EXPECT_NEAR(st.error_after, opt.error(), 1e-9 * std::max(1e-12, opt.error()) + 1e-15);
  EXPECT(gpslam_hip_get_states(h, P.data(), V.data()) == 0);
  for (int k = 0; k < N; k++) {
    const Rot3 r = res.at<Rot3>(Symbol('x', k + 1));
    for (int q = 0; q < 9; q++) EXPECT_NEAR(P[12 * k + q], r.R[q], 1e-10);
    const Vector3 b = res.at<Vector3>(Symbol('b', k + 1)), v = res.at<Vector3>(Symbol('v', k + 1));
    for (int q = 0; q < 3; q++) { EXPECT_NEAR(P[12 * k + 9 + q], b[q], 1e-10); EXPECT_NEAR(V[6 * k + q], v[q], 1e-9); EXPECT(V[6 * k + 3 + q] == 0.0); }
  }
  gpslam_hip_destroy(h);
}

// Round 3: what the reference's constructors accept and the ABI could not yet express -- one Qc_model per GP prior
// (GaussianProcessPriorPose2.h:42-48), a full Gaussian noise model on a measurement factor (GPInterpolatedGPSFactorPose3.h:46-54),
// GaussianProcessInterpolatorLinear::interpolateVelocity (GaussianProcessInterpolatorLinear.h:106-126) and getBodyCentricVb / Vs
// (Pose3utils.cpp:17-24; values of testPose3Utils.cpp's scenario: a pure translation and a pure rotation).
static void test_round3_boundary_additions() {
  {   // two GP priors with DIFFERENT Qc_models in one graph: zero-error fixed point, and a stiffer Qc pulls harder
    auto prior = noiseModel::Isotropic::Sigma(3, 0.001);
    auto QcA = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(3));
    Matrix qb = 0.5 * Matrix::Identity(3); qb(0, 1) = qb(1, 0) = 0.1;
    auto QcB = noiseModel::Gaussian::Covariance(qb);
    Vector3 v = {1, 0, 0};
    NonlinearFactorGraph graph;
    graph.add(PriorFactor<Pose2>(Symbol('x', 1), Pose2(0, 0, 0), prior));
    graph.add(PriorFactor<Pose2>(Symbol('x', 3), Pose2(2, 0, 0), prior));
    graph.add(GaussianProcessPriorPose2(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 1.0, QcA));
    graph.add(GaussianProcessPriorPose2(Symbol('x', 2), Symbol('v', 2), Symbol('x', 3), Symbol('v', 3), 1.0, QcB));
    Values init;
    init.insert(Symbol('x', 1), Pose2(0.1, 0.05, 0.02)); init.insert(Symbol('v', 1), Vector3{0.8, 0.1, 0.0});
    init.insert(Symbol('x', 2), Pose2(0.9, -0.1, 0.05)); init.insert(Symbol('v', 2), Vector3{1.2, 0.0, 0.1});
    init.insert(Symbol('x', 3), Pose2(2.1, 0.1, -0.03)); init.insert(Symbol('v', 3), Vector3{0.9, -0.1, 0.0});
    LevenbergMarquardtOptimizer opt(graph, init);
    opt.optimize();
    Values r = opt.values();
    EXPECT_NEAR(0, graph.error(r), 1e-6);
    EXPECT_NEAR(1.0, r.at<Pose2>(Symbol('x', 2)).x, 1e-4);
    EXPECT(nearV(v, r.at<Vector3>(Symbol('v', 2)), 1e-4));
  }
  {   // a GPS fix with a full covariance: the optimum of  prior + gps  on one coordinate pair is the covariance-weighted mean
    auto Qc_model = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(6));
    Matrix cov = 0.04 * Matrix::Identity(3); cov(0, 1) = cov(1, 0) = 0.02;
    auto gps_model = noiseModel::Gaussian::Covariance(cov);
    auto diag_model = noiseModel::Isotropic::Sigma(3, 0.2);
    Pose3 p1, p2(Rot3(), Point3(0.1, 0, 0));
    Vector6 v = {0, 0, 0, 1, 0, 0};
    for (int full = 0; full < 2; full++) {
      GPInterpolatedGPSFactorPose3 f(Point3(0.05, 0.01, 0), full ? gps_model : diag_model, Qc_model, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 0.1, 0.05);
      NonlinearFactorGraph g;
      g.add(f);
      Values vals;
      vals.insert(Symbol('x', 1), p1); vals.insert(Symbol('v', 1), v); vals.insert(Symbol('x', 2), p2); vals.insert(Symbol('v', 2), v);
      // e = (0, -0.01, 0): error = 0.5 e^T cov^-1 e
      const double e1 = -0.01;
      const double expect = full ? 0.5 * e1 * e1 * (0.04 / (0.04 * 0.04 - 0.02 * 0.02)) : 0.5 * e1 * e1 / 0.04;
      EXPECT_NEAR(expect, g.error(vals), 1e-12);
    }
  }
  {   // interpolateVelocity: constant velocity in, the same velocity out; H2 + H4 blocks at tau = 0 are [I, 0]
    auto Qc3 = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(3));
    GaussianProcessInterpolatorLinear<3> lin(Qc3, 0.1, 0.03), at0(Qc3, 0.1, 0.0);
    Vector3 a = {0, 0, 0}, va = {1, 2, 3}, b = {0.1, 0.2, 0.3};
    Vector3 qv = lin.interpolateVelocity(a, va, b, va);
    EXPECT(nearV(va, qv, 1e-12));
    Matrix H1, H2, H3, H4;
    Vector3 q0 = at0.interpolateVelocity(a, va, b, Vector3{4, 5, 6}, &H1, &H2, &H3, &H4);
    EXPECT(nearV(va, q0, 1e-12));
    for (int i = 0; i < 3; i++) { EXPECT_NEAR(1.0, H2(i, i), 1e-12); EXPECT_NEAR(0.0, H1(i, i), 1e-12); EXPECT_NEAR(0.0, H3(i, i), 1e-12); EXPECT_NEAR(0.0, H4(i, i), 1e-12); }
  }
  {   // body-centric velocities: a pure translation along x in 0.1 s, then a yaw of 0.1 rad in 0.1 s
    Pose3 a, b(Rot3(), Point3(0.1, 0, 0)), c(Rot3::Ypr(0.1, 0, 0), Point3(0, 0, 0));
    Vector6 vb = getBodyCentricVb(a, b, 0.1), vs = getBodyCentricVs(a, b, 0.1);
    EXPECT(nearV(Vector6{0, 0, 0, 1, 0, 0}, vb, 1e-12) && nearV(Vector6{0, 0, 0, 1, 0, 0}, vs, 1e-12));
    Vector6 wb = getBodyCentricVb(a, c, 0.1);
    EXPECT(nearV(Vector6{0, 0, 1, 0, 0, 0}, wb, 1e-12));
    std::vector<Vector6> many = getBodyCentricVb(std::vector<Pose3>{a, a}, std::vector<Pose3>{b, c}, std::vector<double>{0.1, 0.1});
    EXPECT(many.size() == 2 && nearV(vb, many[0], 0) && nearV(wb, many[1], 0));
  }
}

// Round 4: equals() / print() of the factor classes (gpslam/gp/GaussianProcessPriorPose3.h:104-115 and its siblings): same
// class + keys + noise model + the members the class compares; the 2D-linear interpolated range factor leaves its GP base
// out (GPInterpolatedRangeFactor2DLinear.h:96-100).  No device call anywhere here: runs on the CPU as well (--host-only).
static void test_equals_print_clone() {
  auto Qc = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(6));
  auto Qc2 = noiseModel::Gaussian::Covariance(0.02 * Matrix::Identity(6));
  GaussianProcessPriorPose3 a(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 0.1, Qc);
  GaussianProcessPriorPose3 same(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 0.1, Qc);
  GaussianProcessPriorPose3 other_dt(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 0.2, Qc);
  GaussianProcessPriorPose3 other_key(Symbol('x', 1), Symbol('v', 1), Symbol('x', 3), Symbol('v', 3), 0.1, Qc);
  GaussianProcessPriorPose3 other_qc(Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), 0.1, Qc2);
  GaussianProcessPriorPose3VW vw(Symbol('x', 1), Symbol('v', 1), Symbol('w', 1), Symbol('x', 2), Symbol('v', 2), Symbol('w', 2), 0.1, Qc);
  EXPECT(a.equals(same) && a.equals(same, 1e-12));
  EXPECT(!a.equals(other_dt) && !a.equals(other_key) && !a.equals(other_qc));
  EXPECT(!a.equals(vw) && !vw.equals(a));                     // another class: the dynamic_cast fails
  EXPECT(a.equals(other_dt, 0.5));                            // |delta_t - delta_t'| < tol, as the reference compares it
  NonlinearFactor::shared_ptr c = a.clone();
  EXPECT(c->equals(a) && a.equals(*c) && c->size() == 4);
  auto Qc3 = noiseModel::Gaussian::Covariance(0.001 * Matrix::Identity(3));
  auto Qc3b = noiseModel::Gaussian::Covariance(0.002 * Matrix::Identity(3));
  auto rm = noiseModel::Isotropic::Sigma(1, 0.1);
  Pose2 sensor(0.1, 0.2, 0.3);
  GPInterpolatedRangeFactorPose2 r1(2.5, rm, Qc3, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), 0.1, 0.04);
  GPInterpolatedRangeFactorPose2 r2(2.5, rm, Qc3, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), 0.1, 0.04);
  GPInterpolatedRangeFactorPose2 r_meas(2.6, rm, Qc3, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), 0.1, 0.04);
  GPInterpolatedRangeFactorPose2 r_tau(2.5, rm, Qc3, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), 0.1, 0.05);
  GPInterpolatedRangeFactorPose2 r_sens(2.5, rm, Qc3, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), 0.1, 0.04, &sensor);
  EXPECT(r1.equals(r2) && !r1.equals(r_meas) && !r1.equals(r_tau) && !r1.equals(r_sens) && r_sens.equals(r_sens));
  // the 2D-linear variant compares keys, noise model and the measurement only (measurement, then the keys, then the models: its own constructor order)
  GPInterpolatedRangeFactor2DLinear l1(2.5, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), rm, Qc3, 0.1, 0.04);
  GPInterpolatedRangeFactor2DLinear l_tau(2.5, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), rm, Qc3b, 0.2, 0.07);
  GPInterpolatedRangeFactor2DLinear l_meas(2.7, Symbol('x', 1), Symbol('v', 1), Symbol('x', 2), Symbol('v', 2), Symbol('l', 1), rm, Qc3, 0.1, 0.04);
  EXPECT(l1.equals(l_tau) && !l1.equals(l_meas));
  PriorFactor<Pose2> p1(Symbol('x', 1), Pose2(1, 2, 0.3), noiseModel::Isotropic::Sigma(3, 0.1));
  PriorFactor<Pose2> p2(Symbol('x', 1), Pose2(1, 2, 0.3), noiseModel::Isotropic::Sigma(3, 0.1));
  PriorFactor<Pose2> p3(Symbol('x', 1), Pose2(1, 2, 0.3), noiseModel::Isotropic::Sigma(3, 0.2));
  EXPECT(p1.equals(p2) && !p1.equals(p3) && !p1.equals(a));
  RangeBearingFactor2DLinear rb(Symbol('x', 1), Symbol('l', 1), 0.5, 3.0, noiseModel::Isotropic::Sigma(2, 0.1));
  EXPECT(rb.equals(*rb.clone()));
  // print(): the reference's first line, then keys through the formatter
  std::printf("--- print() samples\n");
  a.print("factor 1: ");
  r_sens.print("", [](Key k) { return std::string("<") + DefaultKeyFormatter(k) + ">"; });
  EXPECT(DefaultKeyFormatter(Symbol('x', 12)) == "x12");
  // plain integer keys are legal gtsam keys and 0 is one of them (print() used to drop a zero key after the first):
  // the range factor on (pose key 7, landmark key 0) and the between factor on (3, 0) print both
  RangeBearingFactor2DLinear rb0(Key(7), Key(0), 0.5, 3.0, noiseModel::Isotropic::Sigma(2, 0.1));
  rb0.print("zero landmark key: ", [](Key k) { return std::string("[k") + std::to_string((unsigned long long)k) + "]"; });
  BetweenFactor<Pose2> b0(Key(3), Key(0), Pose2(1, 0, 0), noiseModel::Isotropic::Sigma(3, 0.1));
  b0.print("zero second key: ", [](Key k) { return std::string("[k") + std::to_string((unsigned long long)k) + "]"; });
}

// a graph built through the host classes (every GP prior carries its own Qc_model, all the same matrix) keeps the
// structured-record path of the fused kernel
static void test_single_qc_graph_keeps_the_structured_path() {
  auto Qc = noiseModel::Gaussian::Covariance(0.01 * Matrix::Identity(6));
  auto prior = noiseModel::Isotropic::Sigma(6, 0.001);
  const int N = 600;
  NonlinearFactorGraph graph;
  Values init;
  Vector6 v = {0, 0, 0, 1, 0, 0};
  graph.add(PriorFactor<Pose3>(Symbol('x', 0), Pose3(), prior));
  graph.add(PriorFactor<Pose3>(Symbol('x', N - 1), Pose3(Rot3(), Point3(0.1 * (N - 1), 0, 0)), prior));   // (two poses pin the velocity as well)
  for (int i = 0; i < N; i++) {
    init.insert(Symbol('x', i), Pose3(Rot3(), Point3(0.1 * i, 0.001 * (i % 7), 0)));
    init.insert(Symbol('v', i), v);
    if (i + 1 < N) graph.add(GaussianProcessPriorPose3(Symbol('x', i), Symbol('v', i), Symbol('x', i + 1), Symbol('v', i + 1), 0.1, Qc));
  }
  GaussNewtonOptimizer opt(graph, init);
  int32_t info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  EXPECT(gpslam_hip_plan_info(opt.handle(), info) == 0);
  EXPECT(info[3] == 1 && info[4] == 1);     // fused level 0, structured GP records
  opt.optimize();
  EXPECT(graph.error(opt.values()) < 1e-6);
}

// ABI 2.0 (round 6): the version symbol, the struct sizes, and what gpslam_hip_create_v2 does with a caller's struct_size
static void test_abi_version_and_struct_sizes() {      // (host only: no handle is created)
  EXPECT(gpslam_hip_abi_version() == (uint32_t)GPSLAM_HIP_ABI_VERSION);
  EXPECT(gpslam_hip_struct_size(GPSLAM_STRUCT_CONFIG) == sizeof(gpslam_hip_config));
  EXPECT(gpslam_hip_struct_size(GPSLAM_STRUCT_CONFIG_V2) == sizeof(gpslam_hip_config_v2));
  EXPECT(gpslam_hip_struct_size(GPSLAM_STRUCT_STATS) == sizeof(gpslam_hip_stats));
  EXPECT(gpslam_hip_struct_size(GPSLAM_STRUCT_PARAMS) == sizeof(gpslam_hip_params));
  EXPECT(gpslam_hip_struct_size(99) == 0);
  EXPECT(sizeof(gpslam_hip_config) == 64 && sizeof(gpslam_hip_config_v2) == 64 && sizeof(gpslam_hip_stats) == 56);
  gtsam::detail::check_abi();                           // what the host classes do before their first handle
  gpslam_hip_handle *h = nullptr;
  gpslam_hip_config_v2 c;
  std::memset(&c, 0, sizeof(c));
  c.manifold = GPSLAM_POSE3; c.nranks = 1;
  c.struct_size = 8;                                    // shorter than the fields every version has
  EXPECT(gpslam_hip_create_v2(&c, &h) == GPSLAM_E_INVALID && h == nullptr);
  EXPECT(gpslam_hip_create_v2(nullptr, &h) == GPSLAM_E_INVALID);
  c.struct_size = (uint32_t)sizeof(c); c.plan = 1 << 20;     // an unknown plan bit is refused, by name
  EXPECT(gpslam_hip_create_v2(&c, &h) == GPSLAM_E_INVALID && h == nullptr);
  struct { gpslam_hip_config_v2 c; int32_t knob_of_a_newer_header; } big;
  std::memset(&big, 0, sizeof(big));
  big.c.struct_size = (uint32_t)sizeof(big); big.c.manifold = GPSLAM_POSE3; big.c.nranks = 1;
  big.knob_of_a_newer_header = 3;                       // a knob this build cannot honour is not dropped silently
  EXPECT(gpslam_hip_create_v2(&big.c, &h) == GPSLAM_E_UNSUPPORTED && h == nullptr);
}
static void test_config_v1_v2_and_truncated_v2_reach_the_same_handle() {      // (GPU)
  int32_t i1[8], i2[8], i3[8];
  gpslam_hip_handle *h1 = nullptr, *h2 = nullptr, *h3 = nullptr;
  gpslam_hip_config c1;                                 // v1: a caller built against the rounds 1-5 header
  std::memset(&c1, 0, sizeof(c1));
  c1.manifold = GPSLAM_POSE3; c1.nranks = 1; c1.reserved[6] = GPSLAM_PLAN_UNFUSED_LEVEL0 | GPSLAM_PLAN_GP_ROWS;
  EXPECT(gpslam_hip_create(&c1, &h1) == 0);
  gpslam_hip_config_v2 c2 = gtsam::detail::make_config(GPSLAM_POSE3);
  c2.chart = GPSLAM_CHART_EXPMAP; c2.plan = GPSLAM_PLAN_UNFUSED_LEVEL0 | GPSLAM_PLAN_GP_ROWS;
  EXPECT(gpslam_hip_create_v2(&c2, &h2) == 0);
  gpslam_hip_config_v2 c3 = gtsam::detail::make_config(GPSLAM_POSE3);     // a v2 caller whose header ended before `plan`
  c3.struct_size = (uint32_t)offsetof(gpslam_hip_config_v2, plan);
  c3.plan = 0x7fffffff;                                 // (beyond its struct_size: not read)
  EXPECT(gpslam_hip_create_v2(&c3, &h3) == 0);
  c1.reserved[7] = 1;
  gpslam_hip_handle *h4 = nullptr;
  EXPECT(gpslam_hip_create(&c1, &h4) == GPSLAM_E_INVALID);
  const int N = 600;
  std::vector<double> P((size_t)N * 12, 0.0), V((size_t)N * 6, 0.0);
  std::vector<int32_t> left(N - 1);
  std::vector<double> dt(N - 1, 0.1);
  for (int k = 0; k < N; k++) { P[12 * k] = P[12 * k + 4] = P[12 * k + 8] = 1.0; P[12 * k + 9] = 0.1 * k; V[6 * k + 3] = 1.0; }
  for (int k = 0; k + 1 < N; k++) left[k] = k;
  const int32_t i0 = 0; const double sig[6] = {1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3};
  for (gpslam_hip_handle *h : {h1, h2, h3}) {
    EXPECT(gpslam_hip_set_states(h, N, P.data(), V.data()) == 0);
    EXPECT(gpslam_hip_add_gp_priors(h, N - 1, left.data(), dt.data()) == 0);
    EXPECT(gpslam_hip_add_pose_priors(h, 1, &i0, P.data(), sig) == 0);
    EXPECT(gpslam_hip_compile(h) == 0);
  }
  EXPECT(gpslam_hip_plan_info(h1, i1) == 0 && gpslam_hip_plan_info(h2, i2) == 0 && gpslam_hip_plan_info(h3, i3) == 0);
  EXPECT(std::memcmp(i1, i2, sizeof(i1)) == 0);         // the same plan through either struct
  EXPECT(i1[3] == 0 && i1[4] == 0 && i3[3] == 1 && i3[4] == 1);   // ... the two-launch row path where asked for, the default plan where not
  gpslam_hip_stats s1, s2;
  EXPECT(gpslam_hip_iterate_gn(h1, &s1) == 0 && gpslam_hip_iterate_gn(h2, &s2) == 0);
  EXPECT(s1.error_before == s2.error_before && s1.error_after == s2.error_after);
  gpslam_hip_destroy(h1); gpslam_hip_destroy(h2); gpslam_hip_destroy(h3);
}

// A loop closure the way a GTSAM user writes one (round 6): BetweenFactor<Pose2> between the LAST and the FIRST state of a chain
// that drives once around a circle -- the reference's factors take arbitrary keys (gpslam/gp/GaussianProcessPriorPose3.h:43-47) and
// GTSAM's optimisers eliminate whatever graph results; rounds 1-5 of this header threw std::invalid_argument here.
static void test_loop_closure_between_non_adjacent_keys() {
  const int N = 48;
  const double dt = 0.25, w = 2 * M_PI / ((N - 1) * dt), v = 1.0;      // constant twist: one full turn over the N - 1 intervals
  auto step = [&](double bias) {                                        // Pose2::Expmap(dt * (v, 0, w + bias))
    const double th = dt * (w + bias);
    return Pose2(v * dt * std::sin(th) / th, v * dt * (1 - std::cos(th)) / th, th);
  };
  auto compose = [](const Pose2 &a, const Pose2 &b) {
    return Pose2(a.x + std::cos(a.theta) * b.x - std::sin(a.theta) * b.y, a.y + std::sin(a.theta) * b.x + std::cos(a.theta) * b.y, a.theta + b.theta);
  };
  std::vector<Pose2> truth(N), dead(N);
  for (int k = 0; k + 1 < N; k++) { truth[k + 1] = compose(truth[k], step(0.0)); dead[k + 1] = compose(dead[k], step(0.02)); }   // biased gyro: drift
  auto Qc_model = noiseModel::Gaussian::Covariance(1.0 * Matrix::Identity(3));
  auto build = [&](bool closure) {
    NonlinearFactorGraph graph;
    graph.add(PriorFactor<Pose2>(Symbol('x', 0), truth[0], noiseModel::Isotropic::Sigma(3, 1e-3)));
    for (int k = 0; k + 1 < N; k++) {
      graph.add(GaussianProcessPriorPose2(Symbol('x', k), Symbol('v', k), Symbol('x', k + 1), Symbol('v', k + 1), dt, Qc_model));
      graph.add(BetweenFactor<Pose2>(Symbol('x', k), Symbol('x', k + 1), step(0.02), noiseModel::Isotropic::Sigma(3, 2e-2)));
    }
    // the closure: the robot recognises its starting place; x_{N-1}^-1 x_0 = identity after a full turn (keys in "backward" order)
    if (closure) graph.add(BetweenFactor<Pose2>(Symbol('x', N - 1), Symbol('x', 0), Pose2(0, 0, -2 * M_PI), noiseModel::Isotropic::Sigma(3, 1e-3)));
    return graph;
  };
  Values init;
  for (int k = 0; k < N; k++) { init.insert(Symbol('x', k), dead[k]); init.insert(Symbol('v', k), Vector3{v, 0, w}); }
  auto dist = [&](const Values &val, int k) { const Pose2 q = val.at<Pose2>(Symbol('x', k)); return std::hypot(q.x - truth[k].x, q.y - truth[k].y); };
  NonlinearFactorGraph open_graph = build(false), closed_graph = build(true);
  Values open_values = LevenbergMarquardtOptimizer(open_graph, init).optimize();
  LevenbergMarquardtOptimizer opt(closed_graph, init);
  Values closed_values = opt.optimize();
  EXPECT(dist(open_values, N - 1) > 0.15);                       // odometry alone: the drift stays (0.02 rad/s of gyro bias over a turn)
  EXPECT(dist(closed_values, N - 1) < 5e-3);                     // with the closure the last pose is back at the start
  EXPECT(dist(closed_values, N / 2) < 0.6 * dist(open_values, N / 2));   // ... and the correction is spread along the chain
  EXPECT(closed_graph.error(closed_values) < closed_graph.error(init) * 1e-2);
  EXPECT(opt.iterations() >= 2);
  // evaluateError of the closure factor itself through the ABI at the optimum: small against its sigma-scaled initial value
  GaussNewtonOptimizer gn(closed_graph, closed_values);           // Gauss-Newton from the LM optimum: a fixed point
  Values again = gn.optimize();
  EXPECT_NEAR(closed_graph.error(again), closed_graph.error(closed_values), 1e-6 * closed_graph.error(closed_values));
}

int main(int argc, char **argv) {
  if (argc > 1 && std::strcmp(argv[1], "--host-only") == 0) {   // what needs no GPU (the CPU test run)
    test_equals_print_clone();
    test_abi_version_and_struct_sizes();
    if (failures == 0) std::printf("host_api_tests: all host-only tests passed\n");
    return failures == 0 ? 0 : 1;
  }
  test_equals_print_clone();
  test_abi_version_and_struct_sizes();
  test_config_v1_v2_and_truncated_v2_reach_the_same_handle();
  test_single_qc_graph_keeps_the_structured_path();
  test_gp_prior_pose3_optimization();
  test_gp_prior_pose2_rot3_linear_optimization();
  test_interp_range_pose2_optimization();
  test_interp_range_pose3_optimization_with_extrapolation();
  test_range_bearing_2dlinear_optimization();
  test_gp_prior_pose3vw_optimization();
  test_projection_optimization_and_trajectory_query();
  test_evaluate_error_and_interpolators();
  test_error_conventions();
  test_ahrs_graph_with_bias_states();
  test_round3_boundary_additions();
  test_loop_closure_between_non_adjacent_keys();
  if (failures == 0) std::printf("host_api_tests: all tests passed\n");
  return failures == 0 ? 0 : 1;
}
