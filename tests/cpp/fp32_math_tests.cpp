// Host-side unit tests of the fp32 arithmetic of the factor kernels (gpslam_amd/csrc/factors.hpp, "fp32 arithmetic"):
// the functions are __host__ __device__, so hipcc's host pass runs them on the CPU, no GPU needed.
//   1. series-blended coefficients == the reference's closed forms (Pose3utils.cpp:98-104, :219-223) in fp64, across the blend
//   2. the fp32 coefficients are accurate to fp32 rounding where the closed forms in fp32 are not
//   3. the exact derivative d(Jr^-1(xi) x)/d xi (forward-mode duals) == the reference's central difference
//      (jacobianMethodNumercialDiff, Pose3utils.cpp:167-179) evaluated in fp64
#include <cmath>
#include <cstdio>
#include <random>

#include "../../gpslam_amd/csrc/factors.hpp"

using namespace gps;

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fails++; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static void closed_forms(double th, double &c, double &qa, double &qb, double &qc) {
  const double s = std::sin(th), co = std::cos(th), t2 = th * th, t3 = t2 * th, t4 = t3 * th, t5 = t4 * th;
  c = 1.0 / t2 - (1.0 + co) / (2.0 * th * s);
  qa = (th - s) / t3;
  qb = (1.0 - 0.5 * t2 - co) / t4;
  qc = -0.5 * ((1.0 - 0.5 * t2 - co) / t4 - 3.0 * (th - s - t3 / 6.0) / t5);
}

int main() {
  // 1. fp64: blended == closed form wherever the closed form is itself accurate (th >= 0.05), continuous at the blend
  for (double th : {0.05, 0.1, 0.3, 0.4999, 0.5, 0.5001, 0.7, 1.0, 2.0, 3.0}) {
    double c, qa, qb, qc;
    closed_forms(th, c, qa, qb, qc);
    const JrK<double> k = jr_coefs_smooth<double>(V3<double>{th * 0.6, -th * 0.8, 0.0});
    const double tol = th < 0.2 ? 3e-9 : 1e-12;     // the CLOSED forms lose 1e-16 / th^4 at small th
    CHECK(std::fabs(k.c - c) < tol && std::fabs(k.qa - qa) < tol && std::fabs(k.qb - qb) < tol && std::fabs(k.qc - qc) < tol,
          "th=%g  c %.3e qa %.3e qb %.3e qc %.3e", th, k.c - c, k.qa - qa, k.qb - qb, k.qc - qc);
  }
  // 2. fp32 coefficients vs the fp64 blended ones: fp32 rounding everywhere, including th = 0.01 where the closed form of
  //    qb evaluated in fp32 has no correct digit
  for (double th : {1e-4, 1e-3, 0.01, 0.03, 0.1, 0.3, 0.49, 0.51, 1.0, 2.5}) {
    const JrK<double> kd = jr_coefs_smooth<double>(V3<double>{th * 0.6, -th * 0.8, 0.0});
    const JrK<float> kf = jr_coefs(V3<float>{(float)(th * 0.6), (float)(-th * 0.8), 0.f});
    const double tol = th > 2.0 ? 2e-5 : 1e-6;    // near pi the closed forms divide by sin(th)
    CHECK(std::fabs(kf.c - kd.c) < tol * (1 + std::fabs(kd.c)) && std::fabs(kf.qa - kd.qa) < tol && std::fabs(kf.qb - kd.qb) < tol && std::fabs(kf.qc - kd.qc) < tol,
          "fp32 th=%g  c %.3e qa %.3e qb %.3e qc %.3e", th, kf.c - kd.c, kf.qa - kd.qa, kf.qb - kd.qb, kf.qc - kd.qc);
  }
  // 3. exact derivative vs the reference's h = 1e-6 central difference in fp64
  std::mt19937 rng(7);
  std::normal_distribution<double> nd(0.0, 1.0);
  double worst32 = 0.0, worst64 = 0.0, worst_ref_small = 0.0, worst_ref_large = 0.0;
  for (double scale : {0.01, 0.05, 0.2, 0.6, 1.5, 2.5}) {
    for (int rep = 0; rep < 20; rep++) {
      double w[3] = {nd(rng), nd(rng), nd(rng)};
      const double n = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      for (double &q : w) q *= scale / n;
      const V6<double> xi = {{w[0], w[1], w[2]}, {nd(rng), nd(rng), nd(rng)}};
      const V6<double> x = {{nd(rng), nd(rng), nd(rng)}, {nd(rng), nd(rng), nd(rng)}};
      const BL6<double> FDref = se3_jrinv_times_x_fd(xi, x);     // the reference's construction: closed forms, h = 1e-6
      // yardstick: central difference with h = 1e-4 of the SAME map with the cancellation-free coefficients (noise 1e-12,
      // truncation ~1e-8).  The reference's own difference is noisier than that at small angles: the rounding error of
      // its closed-form coefficients (1e-16 / th^4) is divided by 2h.
      BL6<double> FD;
      FD.A = M3<double>::zero(); FD.C = M3<double>::zero(); FD.D = M3<double>::zero();
      for (int i = 0; i < 6; i++) {
        const double hh = 1e-4;
        V6<double> xp = xi, xn = xi;
        double *pp[6] = {&xp.w.x, &xp.w.y, &xp.w.z, &xp.v.x, &xp.v.y, &xp.v.z}, *pn[6] = {&xn.w.x, &xn.w.y, &xn.w.z, &xn.v.x, &xn.v.y, &xn.v.z};
        *pp[i] += hh; *pn[i] -= hh;
        const V6<double> col = (1.0 / (2 * hh)) * (se3_jrinv_apply_k(jr_coefs_smooth<double>(xp.w), xp, x) - se3_jrinv_apply_k(jr_coefs_smooth<double>(xn.w), xn, x));
        if (i < 3) { FD.A.m[i] = col.w.x; FD.A.m[3 + i] = col.w.y; FD.A.m[6 + i] = col.w.z; FD.C.m[i] = col.v.x; FD.C.m[3 + i] = col.v.y; FD.C.m[6 + i] = col.v.z; }
        else { FD.D.m[i - 3] = col.v.x; FD.D.m[3 + i - 3] = col.v.y; FD.D.m[6 + i - 3] = col.v.z; }
      }
      for (int q = 0; q < 9; q++) {
        const double dr = std::fmax(std::fabs(FDref.A.m[q] - FD.A.m[q]), std::fmax(std::fabs(FDref.C.m[q] - FD.C.m[q]), std::fabs(FDref.D.m[q] - FD.D.m[q])));
        if (scale < 0.1) worst_ref_small = std::fmax(worst_ref_small, dr); else worst_ref_large = std::fmax(worst_ref_large, dr);
      }
      const V6<float> xif = {{(float)xi.w.x, (float)xi.w.y, (float)xi.w.z}, {(float)xi.v.x, (float)xi.v.y, (float)xi.v.z}};
      const V6<float> xf = {{(float)x.w.x, (float)x.w.y, (float)x.w.z}, {(float)x.v.x, (float)x.v.y, (float)x.v.z}};
      const BL6<float> AN = se3_jrinv_times_x_fd_k(jr_coefs(xif.w), xif, xf);
      // the same derivative through fp64 duals: isolates the method from fp32 rounding
      typedef Dual3<double> D;
      const V3<D> wd = {D(xi.w.x, 0), D(xi.w.y, 1), D(xi.w.z, 2)};
      const V3<D> rho = {D(xi.v.x), D(xi.v.y), D(xi.v.z)}, xw = {D(x.w.x), D(x.w.y), D(x.w.z)}, xv = {D(x.v.x), D(x.v.y), D(x.v.z)};
      const JrK<D> k = jr_coefs_smooth<D>(wd);
      const V3<D> top = so3_jrinv_apply_k(k, wd, xw);
      const V3<D> bot = so3_jrinv_apply_k(k, wd, xv - se3_Q_apply_k(k, wd, rho, top));
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          worst64 = std::fmax(worst64, std::fabs(top[i].d[j] - FD.A.m[3 * i + j]));
          worst64 = std::fmax(worst64, std::fabs(bot[i].d[j] - FD.C.m[3 * i + j]));
          worst32 = std::fmax(worst32, std::fabs((double)AN.A.m[3 * i + j] - FD.A.m[3 * i + j]));
          worst32 = std::fmax(worst32, std::fabs((double)AN.C.m[3 * i + j] - FD.C.m[3 * i + j]));
          worst32 = std::fmax(worst32, std::fabs((double)AN.D.m[3 * i + j] - FD.D.m[3 * i + j]));
        }
    }
  }
  std::printf("exact derivative vs yardstick: fp64 duals %.2e, fp32 %.2e;  the reference's h = 1e-6 difference vs yardstick: %.2e (th < 0.1), %.2e (th >= 0.2)\n",
              worst64, worst32, worst_ref_small, worst_ref_large);
  CHECK(worst64 < 1e-7, "fp64 duals differ from the yardstick by %.3e", worst64);
  CHECK(worst32 < 5e-6, "fp32 exact derivative differs by %.3e", worst32);
  CHECK(worst_ref_large < 1e-7, "reference-style difference at larger angles differs by %.3e", worst_ref_large);
  if (fails == 0) std::printf("all fp32 math tests passed\n");
  return fails ? 1 : 0;
}
