// gtsam_decl.hpp -- DECLARATION-ONLY stand-ins for the handful of GTSAM / Boost / gpslam names that
// gpslam_amd/host/gtsam_adapter.hpp and bench/gtsam_reference.cpp use.
//
// Purpose: those two files compile only where real GTSAM and the reference are installed, which is nowhere this project
// is built -- a typo in either would ship silently.  With this directory on the include path they
// are TYPE-CHECKED (g++ -fsyntax-only, tests/test_cpp_host.py).  Nothing here has a body, nothing links, nothing runs:
// this pins no behaviour and is not an oracle.  Signatures follow the public GTSAM 4.0 API as the reference uses it
// (e.g. gpslam/gp/GaussianProcessPriorPose3.h:29-55, gpslam/slam/GPInterpolatedRangeFactorPose2.h:40-54,
// gpslam/gp/tests/testGaussianProcessPriorPose3.cpp:162-194); they are written from that usage, not copied from GTSAM.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace boost {
template <typename T> using shared_ptr = std::shared_ptr<T>;
template <typename T, typename U> shared_ptr<T> dynamic_pointer_cast(const shared_ptr<U> &p);
struct none_t {};
extern const none_t none;
template <typename T> class optional {
 public:
  optional();
  optional(none_t);
  optional(const T &v);
  explicit operator bool() const;
  const T &operator*() const;
  const T *operator->() const;
};
}  // namespace boost

namespace gtsam {

typedef std::uint64_t Key;

// just enough of a dense matrix / vector: element access, raw data, size, the expressions the two files write
class Matrix {
 public:
  Matrix();
  Matrix(int rows, int cols);
  double &operator()(int r, int c);
  double operator()(int r, int c) const;
  int rows() const;
  int cols() const;
  const double *data() const;
  static Matrix Identity(int rows, int cols);
};
Matrix operator*(double s, const Matrix &m);
class Matrix3 {
 public:
  double &operator()(int r, int c);
  double operator()(int r, int c) const;
};
class Vector {
 public:
  double &operator()(int i);
  double operator()(int i) const;
  const double *data() const;
  std::size_t size() const;
};
template <int N> class FixedVector {
 public:
  double &operator()(int i);
  double operator()(int i) const;
  const double *data() const;
  std::size_t size() const;
};
typedef FixedVector<3> Vector3;
typedef FixedVector<6> Vector6;

class Symbol {
 public:
  Symbol(unsigned char c, std::uint64_t j);
  Symbol(Key key);
  operator Key() const;
  unsigned char chr() const;
  std::uint64_t index() const;
};

class Point2 {
 public:
  Point2();
  Point2(double x, double y);
  double x() const;
  double y() const;
  double &operator()(int i);
  double operator()(int i) const;
};
class Point3 {
 public:
  Point3();
  Point3(double x, double y, double z);
  double x() const;
  double y() const;
  double z() const;
  double &operator()(int i);
  double operator()(int i) const;
};
class Rot3 {
 public:
  Rot3();
  explicit Rot3(const Matrix3 &R);
  Matrix3 matrix() const;
};
class Pose2 {
 public:
  Pose2();
  Pose2(double x, double y, double theta);
  double x() const;
  double y() const;
  double theta() const;
};
class Pose3 {
 public:
  Pose3();
  Pose3(const Rot3 &R, const Point3 &t);
  const Rot3 &rotation() const;
  const Point3 &translation() const;
};

namespace noiseModel {
class Base {
 public:
  virtual ~Base();
  std::size_t dim() const;
};
class Gaussian : public Base {
 public:
  static boost::shared_ptr<Gaussian> Covariance(const Matrix &cov);
};
class Diagonal : public Gaussian {
 public:
  const Vector &sigmas() const;
};
class Isotropic : public Diagonal {
 public:
  static boost::shared_ptr<Isotropic> Sigma(std::size_t dim, double sigma);
};
}  // namespace noiseModel
typedef boost::shared_ptr<noiseModel::Base> SharedNoiseModel;

class Values;
class NonlinearFactor {
 public:
  typedef boost::shared_ptr<NonlinearFactor> shared_ptr;
  virtual ~NonlinearFactor();
  virtual shared_ptr clone() const;
};
class NoiseModelFactor : public NonlinearFactor {
 public:
  const SharedNoiseModel &noiseModel() const;
};
template <typename T> class PriorFactor : public NoiseModelFactor {
 public:
  PriorFactor(Key key, const T &prior, const SharedNoiseModel &model);
  Key key() const;
  const T &prior() const;
};
template <typename T> class BetweenFactor : public NoiseModelFactor {
 public:
  BetweenFactor(Key key1, Key key2, const T &measured, const SharedNoiseModel &model);
  Key key1() const;
  Key key2() const;
  const T &measured() const;
};

class NonlinearFactorGraph {
 public:
  typedef std::vector<NonlinearFactor::shared_ptr>::const_iterator const_iterator;
  const_iterator begin() const;
  const_iterator end() const;
  template <typename F> void add(const F &factor);
};

class Values {
 public:
  Values();
  Values(const Values &);
  Values &operator=(const Values &);
  std::vector<Key> keys() const;
  bool exists(Key k) const;
  template <typename T> const T &at(Key k) const;
  template <typename T> void insert(Key k, const T &v);
  template <typename T> void update(Key k, const T &v);
};

class Ordering {
 public:
  void push_back(Key k);
};
struct NonlinearOptimizerParams {
  std::size_t maxIterations;
  double relativeErrorTol, absoluteErrorTol, errorTol;
  boost::optional<Ordering> ordering;
};
struct GaussNewtonParams : NonlinearOptimizerParams {};
struct LevenbergMarquardtParams : NonlinearOptimizerParams {
  double lambdaInitial, lambdaFactor, lambdaUpperBound, lambdaLowerBound, minModelFidelity;
};
class NonlinearOptimizer {
 public:
  virtual ~NonlinearOptimizer();
  void iterate();
  double error() const;
  const Values &values() const;
  const Values &optimize();
};
class GaussNewtonOptimizer : public NonlinearOptimizer {
 public:
  GaussNewtonOptimizer(const NonlinearFactorGraph &graph, const Values &init, const GaussNewtonParams &params = GaussNewtonParams());
};
class LevenbergMarquardtOptimizer : public NonlinearOptimizer {
 public:
  LevenbergMarquardtOptimizer(const NonlinearFactorGraph &graph, const Values &init, const LevenbergMarquardtParams &params = LevenbergMarquardtParams());
};
class IndeterminantLinearSystemException {
 public:
  explicit IndeterminantLinearSystemException(Key j);
};

}  // namespace gtsam

namespace gpslam {

gtsam::Matrix getQc(const gtsam::SharedNoiseModel &Qc_model);   // gpslam/gp/GPutils.cpp:16-20

// NoiseModelFactor4 / NoiseModelFactor5 expose key1() .. key5() (gpslam/gp/GaussianProcessPriorPose3.h:29-30)
class FourKeyFactor : public gtsam::NoiseModelFactor {
 public:
  gtsam::Key key1() const;
  gtsam::Key key2() const;
  gtsam::Key key3() const;
  gtsam::Key key4() const;
};
class FiveKeyFactor : public FourKeyFactor {
 public:
  gtsam::Key key5() const;
};
class GaussianProcessPriorPose3 : public FourKeyFactor {
 public:
  GaussianProcessPriorPose3(gtsam::Key poseKey1, gtsam::Key velKey1, gtsam::Key poseKey2, gtsam::Key velKey2, double delta_t,
                            const gtsam::SharedNoiseModel &Qc_model);
};
class GaussianProcessPriorPose2 : public FourKeyFactor {
 public:
  GaussianProcessPriorPose2(gtsam::Key poseKey1, gtsam::Key velKey1, gtsam::Key poseKey2, gtsam::Key velKey2, double delta_t,
                            const gtsam::SharedNoiseModel &Qc_model);
};
class GPInterpolatedRangeFactorPose3 : public FiveKeyFactor {
 public:
  GPInterpolatedRangeFactorPose3(double measured, const gtsam::SharedNoiseModel &meas_model, const gtsam::SharedNoiseModel &Qc_model,
                                 gtsam::Key poseKey1, gtsam::Key velKey1, gtsam::Key poseKey2, gtsam::Key velKey2, gtsam::Key pointKey,
                                 double delta_t, double tau, boost::optional<gtsam::Pose3> body_P_sensor = boost::none);
};
class GPInterpolatedRangeFactorPose2 : public FiveKeyFactor {
 public:
  GPInterpolatedRangeFactorPose2(double measured, const gtsam::SharedNoiseModel &meas_model, const gtsam::SharedNoiseModel &Qc_model,
                                 gtsam::Key poseKey1, gtsam::Key velKey1, gtsam::Key poseKey2, gtsam::Key velKey2, gtsam::Key pointKey,
                                 double delta_t, double tau, boost::optional<gtsam::Pose2> body_P_sensor = boost::none);
};

}  // namespace gpslam
