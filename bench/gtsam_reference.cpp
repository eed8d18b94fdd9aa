// bench/gtsam_reference.cpp -- the TRUE reference baseline of BASELINE.md section 2 item 2: the identical synthetic
// config-3 graph built with the real gpslam:: factors on real GTSAM, timing optimizer.iterate() on the host cores
// exactly as matlab/PlazaPose2.m:219-233 does.  Compiled by bench.py only where <gtsam/...> and <gpslam/...> are
// installed (neither is in the build image or on the GPU box: there bench.py reports "reference binary unavailable"
// and no number is invented).  Without the headers this file is an empty program.
//
//   g++ -O3 -std=c++11 bench/gtsam_reference.cpp -lgpslam -lgtsam -ltbb -o bench/gtsam_reference
//   bench/gtsam_reference problem.bin [iterations]
// problem.bin (written by bench.py: write_reference_problem): int64 N; double dt; double qc; then per state 12 doubles
// pose (R row-major, t) and 6 doubles velocity (initial values); N-1 odometry poses (12 doubles each); double
// sigma_odo; prior pose (12 doubles); double sigma_prior.
#if defined(__has_include)
#if __has_include(<gtsam/nonlinear/LevenbergMarquardtOptimizer.h>) && __has_include(<gpslam/gp/GaussianProcessPriorPose3.h>)
#define HAVE_REFERENCE 1
#endif
#endif

#include <cstdio>

#ifdef HAVE_REFERENCE
#include <gtsam/geometry/Pose3.h>
#include <gtsam/inference/Symbol.h>
#include <gtsam/nonlinear/GaussNewtonOptimizer.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/PriorFactor.h>

#include <gpslam/gp/GaussianProcessPriorPose3.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <vector>

using namespace gtsam;

static Pose3 read_pose(const double *p) {
  Matrix3 R;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R(i, j) = p[3 * i + j];
  return Pose3(Rot3(R), Point3(p[9], p[10], p[11]));
}

int main(int argc, char **argv) {
  if (argc < 2) { std::printf("usage: %s problem.bin [iterations]\n", argv[0]); return 2; }
  const int iters = argc > 2 ? std::atoi(argv[2]) : 3;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int64_t N;
  double dt, qc, sig_odo, sig_prior;
  if (std::fread(&N, 8, 1, f) != 1 || std::fread(&dt, 8, 1, f) != 1 || std::fread(&qc, 8, 1, f) != 1) return 2;
  std::vector<double> st((size_t)N * 18), odo((size_t)(N - 1) * 12), prior(12);
  if (std::fread(st.data(), 8, st.size(), f) != st.size() || std::fread(odo.data(), 8, odo.size(), f) != odo.size() ||
      std::fread(&sig_odo, 8, 1, f) != 1 || std::fread(prior.data(), 8, 12, f) != 12 || std::fread(&sig_prior, 8, 1, f) != 1) return 2;
  std::fclose(f);
  NonlinearFactorGraph graph;
  Values init;
  SharedNoiseModel Qc_model = noiseModel::Gaussian::Covariance(qc * Matrix::Identity(6, 6));
  SharedNoiseModel odo_model = noiseModel::Isotropic::Sigma(6, sig_odo), prior_model = noiseModel::Isotropic::Sigma(6, sig_prior);
  Ordering ordering;                                       // the explicit chain order both sides use
  for (int64_t i = 0; i < N; i++) {
    init.insert(Symbol('x', i), read_pose(&st[(size_t)i * 18]));
    Vector6 v;
    for (int q = 0; q < 6; q++) v(q) = st[(size_t)i * 18 + 12 + q];
    init.insert(Symbol('v', i), v);
    ordering.push_back(Symbol('x', i));
    ordering.push_back(Symbol('v', i));
  }
  graph.add(PriorFactor<Pose3>(Symbol('x', 0), read_pose(prior.data()), prior_model));
  for (int64_t i = 0; i + 1 < N; i++) {
    graph.add(BetweenFactor<Pose3>(Symbol('x', i), Symbol('x', i + 1), read_pose(&odo[(size_t)i * 12]), odo_model));
    graph.add(gpslam::GaussianProcessPriorPose3(Symbol('x', i), Symbol('v', i), Symbol('x', i + 1), Symbol('v', i + 1), dt, Qc_model));
  }
  for (int use_lm = 0; use_lm < 2; use_lm++) {
    double best = 1e300, err = 0.0;
    if (use_lm) {
      LevenbergMarquardtParams p;
      p.ordering = ordering;
      LevenbergMarquardtOptimizer opt(graph, init, p);
      for (int it = 0; it < iters; it++) {
        auto t0 = std::chrono::steady_clock::now();
        opt.iterate();
        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      }
      err = opt.error();
    } else {
      GaussNewtonParams p;
      p.ordering = ordering;
      GaussNewtonOptimizer opt(graph, init, p);
      for (int it = 0; it < iters; it++) {
        auto t0 = std::chrono::steady_clock::now();
        opt.iterate();
        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      }
      err = opt.error();
    }
    std::printf("{\"optimizer\": \"%s\", \"states\": %lld, \"seconds_per_iteration\": %.6f, \"error\": %.9e}\n",
                use_lm ? "LevenbergMarquardtOptimizer" : "GaussNewtonOptimizer", (long long)N, best, err);
  }
  return 0;
}
#else
int main() {
  std::printf("reference binary unavailable: GTSAM / gpslam headers not installed\n");
  return 3;
}
#endif
