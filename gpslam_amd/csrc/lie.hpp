// lie.hpp -- register-resident Lie-group maths for the batched factor kernels (gfx950).
//
// One thread evaluates one factor, so everything here is a fixed-size value type whose loops
// unroll completely and whose storage is VGPRs.  SE(3) objects are kept in 3x3 block form
// ([[A,0],[C,A']] lower block-triangular Jacobians) instead of dense 6x6, which removes the
// multiplications by structural zeros that the reference's Eigen expressions perform.
//
// Semantics follow GTSAM's conventions as used by gtrll/gpslam:
// Pose3 tangent (omega, v); Pose2 tangent (vx, vy, omega); right Jacobians.
// Reference call sites: gpslam/gp/GaussianProcessPriorPose3.h:72-76, gpslam/gp/Pose3utils.cpp:92-224.
#pragma once

#include <hip/hip_runtime.h>

#define GD __host__ __device__ __forceinline__

namespace gps {

// <cmath> for both scalar types under the unqualified names the templates use.  In the HOST pass ::cos(float) is the
// C function taking double (the fp32 instantiations would silently compute, and return, doubles); these overloads make
// the float instantiations single precision in both passes.
GD float sin(float x) { return ::sinf(x); }
GD double sin(double x) { return ::sin(x); }
GD float cos(float x) { return ::cosf(x); }
GD double cos(double x) { return ::cos(x); }
GD float tan(float x) { return ::tanf(x); }
GD double tan(double x) { return ::tan(x); }
GD float sqrt(float x) { return ::sqrtf(x); }
GD double sqrt(double x) { return ::sqrt(x); }
GD float acos(float x) { return ::acosf(x); }
GD double acos(double x) { return ::acos(x); }
GD float atan2(float y, float x) { return ::atan2f(y, x); }
GD double atan2(double y, double x) { return ::atan2(y, x); }
GD float fabs(float x) { return ::fabsf(x); }
GD double fabs(double x) { return ::fabs(x); }
GD float fmax(float a, float b) { return ::fmaxf(a, b); }
GD double fmax(double a, double b) { return ::fmax(a, b); }

template <typename T> struct Eps;
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };
template <> struct Eps<float> { static constexpr float v = 1.1920929e-07f; };

template <typename T> struct V3 {
  T x, y, z;
  GD T operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <typename T> GD V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> GD V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> GD V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <typename T> GD V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> GD T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> GD V3<T> cross(V3<T> a, V3<T> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// 3x3 matrix, row-major
template <typename T> struct M3 {
  T m[9];
  GD T &operator()(int i, int j) { return m[3 * i + j]; }
  GD T operator()(int i, int j) const { return m[3 * i + j]; }
  static GD M3 identity() { return {{T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}}; }
  static GD M3 zero() { return {{T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)}}; }
};
template <typename T> GD M3<T> operator*(const M3<T> &a, const M3<T> &b) {
  M3<T> c;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return c;
}
template <typename T> GD V3<T> operator*(const M3<T> &a, V3<T> v) {
  return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
          a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
template <typename T> GD M3<T> operator+(const M3<T> &a, const M3<T> &b) {
  M3<T> c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.m[i] = a.m[i] + b.m[i];
  return c;
}
template <typename T> GD M3<T> operator-(const M3<T> &a, const M3<T> &b) {
  M3<T> c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.m[i] = a.m[i] - b.m[i];
  return c;
}
template <typename T> GD M3<T> operator*(T s, const M3<T> &a) {
  M3<T> c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.m[i] = s * a.m[i];
  return c;
}
template <typename T> GD M3<T> neg(const M3<T> &a) { return T(-1) * a; }
template <typename T> GD M3<T> transpose(const M3<T> &a) {
  return {{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}};
}
// a^T * v
template <typename T> GD V3<T> tmul(const M3<T> &a, V3<T> v) {
  return {a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
          a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z};
}
template <typename T> GD M3<T> skew(V3<T> w) { return {{T(0), -w.z, w.y, w.z, T(0), -w.x, -w.y, w.x, T(0)}}; }

// ------------------------------------------------------------------ SO(3)

// Rodrigues (SO3::Expmap)
template <typename T> GD M3<T> so3_exp(V3<T> w) {
  const T th2 = dot(w, w);
  const M3<T> W = skew(w);
  if (th2 > Eps<T>::v) {
    const T th = sqrt(th2);
    const T a = sin(th) / th;
    const T h = sin(T(0.5) * th);
    const T b = T(2) * h * h / th2;
    return M3<T>::identity() + a * W + b * (W * W);
  }
  return M3<T>::identity() + W;
}

// SO3::Logmap including the trace -> -1 and trace -> 3 branches
template <typename T> GD V3<T> so3_log(const M3<T> &R) {
  const T tr = R.m[0] + R.m[4] + R.m[8];
  const T PI = T(3.14159265358979323846);
  if (fabs(tr + T(1)) < T(1e-10)) {
    if (fabs(R.m[8] + T(1)) > T(1e-10)) {
      const T k = PI / sqrt(T(2) + T(2) * R.m[8]);
      return {k * R.m[2], k * R.m[5], k * (T(1) + R.m[8])};
    } else if (fabs(R.m[4] + T(1)) > T(1e-10)) {
      const T k = PI / sqrt(T(2) + T(2) * R.m[4]);
      return {k * R.m[1], k * (T(1) + R.m[4]), k * R.m[7]};
    } else {
      const T k = PI / sqrt(T(2) + T(2) * R.m[0]);
      return {k * (T(1) + R.m[0]), k * R.m[3], k * R.m[6]};
    }
  }
  T mag;
  const T tr_3 = tr - T(3);
  if (tr_3 < T(-1e-7)) {
    const T th = acos((tr - T(1)) * T(0.5));
    mag = th / (T(2) * sin(th));
  } else {
    mag = T(0.5) - tr_3 * tr_3 / T(12);
  }
  return {mag * (R.m[7] - R.m[5]), mag * (R.m[2] - R.m[6]), mag * (R.m[3] - R.m[1])};
}

// Rot3::CayleyChart::Retract: the default Rot3 retraction of a GTSAM 4.0 build without GTSAM_ROT3_EXPMAP (SURVEY.md
// Appendix A); agrees with Expmap to second order, so the fixed point of an optimisation does not depend on it
template <typename T> GD M3<T> so3_cayley(V3<T> w) {
  const T x = w.x, y = w.y, z = w.z;
  const T x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
  const T f = T(1) / (T(4) + x2 + y2 + z2), f2 = T(2) * f;
  return {{(T(4) + x2 - y2 - z2) * f, (xy - T(2) * z) * f2, (xz + T(2) * y) * f2,
           (xy + T(2) * z) * f2, (T(4) - x2 + y2 - z2) * f, (yz - T(2) * x) * f2,
           (xz - T(2) * y) * f2, (yz + T(2) * x) * f2, (T(4) - x2 - y2 + z2) * f}};
}

// right Jacobian Jr(w) (SO3::ExpmapDerivative; rightJacobianRot3, Pose3utils.cpp:203-212)
template <typename T> GD M3<T> so3_jr(V3<T> w) {
  const T th2 = dot(w, w);
  if (th2 <= Eps<T>::v) return M3<T>::identity();
  const T th = sqrt(th2);
  const M3<T> Y = (T(1) / th) * skew(w);
  return M3<T>::identity() - ((T(1) - cos(th)) / th) * Y + (T(1) - sin(th) / th) * (Y * Y);
}

// inverse right Jacobian (SO3::LogmapDerivative; rightJacobianRot3inv, Pose3utils.cpp:215-224)
template <typename T> GD M3<T> so3_jrinv(V3<T> w) {
  const T th2 = dot(w, w);
  if (th2 <= Eps<T>::v) return M3<T>::identity();
  const T th = sqrt(th2);
  const M3<T> X = skew(w);
  const T c = T(1) / th2 - (T(1) + cos(th)) / (T(2) * th * sin(th));
  return M3<T>::identity() + T(0.5) * X + c * (X * X);
}

// ------------------------------------------------------------------ SE(3)

template <typename T> struct SE3 {
  M3<T> R;
  V3<T> t;
};

template <typename T> GD SE3<T> se3_between(const SE3<T> &a, const SE3<T> &b) {  // a^-1 * b
  const M3<T> Rt = transpose(a.R);
  return {Rt * b.R, Rt * (b.t - a.t)};
}
template <typename T> GD SE3<T> se3_compose(const SE3<T> &a, const SE3<T> &b) { return {a.R * b.R, a.t + a.R * b.t}; }
template <typename T> GD SE3<T> se3_inverse(const SE3<T> &a) {
  const M3<T> Rt = transpose(a.R);
  return {Rt, -(Rt * a.t)};
}

// 6-vector in (omega, v) halves
template <typename T> struct V6 {
  V3<T> w, v;
};
template <typename T> GD V6<T> operator+(V6<T> a, V6<T> b) { return {a.w + b.w, a.v + b.v}; }
template <typename T> GD V6<T> operator-(V6<T> a, V6<T> b) { return {a.w - b.w, a.v - b.v}; }
template <typename T> GD V6<T> operator*(T s, V6<T> a) { return {s * a.w, s * a.v}; }

// Block lower-triangular 6x6: [[A, 0], [C, D]]
template <typename T> struct BL6 {
  M3<T> A, C, D;
};
template <typename T> GD BL6<T> operator*(const BL6<T> &x, const BL6<T> &y) {
  return {x.A * y.A, x.C * y.A + x.D * y.C, x.D * y.D};
}
template <typename T> GD V6<T> operator*(const BL6<T> &x, V6<T> u) { return {x.A * u.w, x.C * u.w + x.D * u.v}; }
template <typename T> GD BL6<T> neg(const BL6<T> &x) { return {neg(x.A), neg(x.C), neg(x.D)}; }
template <typename T> GD BL6<T> operator-(const BL6<T> &x, const BL6<T> &y) { return {x.A - y.A, x.C - y.C, x.D - y.D}; }
template <typename T> GD BL6<T> operator+(const BL6<T> &x, const BL6<T> &y) { return {x.A + y.A, x.C + y.C, x.D + y.D}; }
template <typename T> GD BL6<T> operator*(T s, const BL6<T> &x) { return {s * x.A, s * x.C, s * x.D}; }
// materialise values HERE (device code): an empty volatile asm that "modifies" them keeps the compiler from sinking their computation
// towards the use and from interleaving what follows with what produced them -- phases stay phases (kernels.hpp: k_gps_lines, K1)
GD void pin(double &x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#else
  (void)x;
#endif
}
GD void pin(float &x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#else
  (void)x;
#endif
}
template <typename T> GD void pin(V3<T> &x) { pin(x.x); pin(x.y); pin(x.z); }
template <typename T> GD void pin(V6<T> &x) { pin(x.w); pin(x.v); }
template <typename T> GD void pin(M3<T> &x) { for (int q = 0; q < 9; q++) pin(x.m[q]); }
// entry (i, j) of the 6x6
template <typename T> GD T bl6_at(const BL6<T> &x, int i, int j) {
  if (i < 3) return j < 3 ? x.A.m[3 * i + j] : T(0);
  return j < 3 ? x.C.m[3 * (i - 3) + j] : x.D.m[3 * (i - 3) + (j - 3)];
}

// Pose3::AdjointMap = [[R, 0], [t^ R, R]]
template <typename T> GD BL6<T> se3_adjoint(const SE3<T> &g) { return {g.R, skew(g.t) * g.R, g.R}; }

// Q block of the SE(3) right Jacobian (Barfoot14tro eq. 102, signs for the right Jacobian):
// rightJacobianPose3Q, Pose3utils.cpp:92-113, including its |theta| > 1e-5 branch.
template <typename T> GD M3<T> se3_Q(V3<T> w, V3<T> rho) {
  const T th = sqrt(dot(w, w));
  const M3<T> X = skew(w), Y = skew(rho);
  const M3<T> XY = X * Y, YX = Y * X, XYX = X * YX;
  T a, b, c;
  if (fabs(th) > T(1e-5)) {
    const T s = sin(th), co = cos(th);
    const T t2 = th * th, t3 = t2 * th, t4 = t3 * th, t5 = t4 * th;
    a = (th - s) / t3;
    b = (T(1) - T(0.5) * t2 - co) / t4;
    c = T(-0.5) * ((T(1) - T(0.5) * t2 - co) / t4 - T(3) * (th - s - t3 / T(6)) / t5);
  } else {
    a = T(1) / T(6);
    b = T(1) / T(24);
    c = T(-0.5) * (T(1) / T(24) + T(3) / T(120));
  }
  const M3<T> t1 = XY + YX - XYX;
  const M3<T> t2m = X * XY + YX * X - T(3) * XYX;
  const M3<T> t3m = XYX * X + X * XYX;
  return T(-0.5) * Y + a * t1 + b * t2m + c * t3m;
}

// fp32: the coefficients by their series in th^2 below th^2 = 0.25 (see factors.hpp, "fp32 arithmetic")
GD M3<float> se3_Q(V3<float> w, V3<float> rho) {
  const float u = dot(w, w);
  float a, b, c;
  if (u < 0.25f) {
    a = 1.f / 6 - u * (1.f / 120 - u * (1.f / 5040 - u * (1.f / 362880 - u * (1.f / 39916800))));
    b = -1.f / 24 + u * (1.f / 720 - u * (1.f / 40320 - u * (1.f / 3628800 - u * (1.f / 479001600))));
    const float qd = -1.f / 120 + u * (1.f / 5040 - u * (1.f / 362880 - u * (1.f / 39916800)));
    c = -0.5f * (b - 3.f * qd);
  } else {
    const float th = sqrt(u), s = sin(th), co = cos(th);
    const float t3 = u * th, t4 = u * u, t5 = t4 * th;
    a = (th - s) / t3;
    b = (1.f - 0.5f * u - co) / t4;
    c = -0.5f * (b - 3.f * (th - s - t3 / 6.f) / t5);
  }
  const M3<float> X = skew(w), Y = skew(rho);
  const M3<float> XY = X * Y, YX = Y * X, XYX = X * YX;
  return -0.5f * Y + a * (XY + YX - XYX) + b * (X * XY + YX * X - 3.f * XYX) + c * (XYX * X + X * XYX);
}
GD M3<float> so3_jrinv(V3<float> w) {
  const float u = dot(w, w);
  const float c = (u < 0.25f) ? 1.f / 12 + u * (1.f / 720 + u * (1.f / 30240 + u * (1.f / 1209600)))
                              : 1.f / u - (1.f + cos(sqrt(u))) / (2.f * sqrt(u) * sin(sqrt(u)));
  const M3<float> X = skew(w);
  return M3<float>::identity() + 0.5f * X + c * (X * X);
}

// rightJacobianPose3inv (Pose3utils.cpp:192-200) = Pose3::LogmapDerivative in terms of xi
template <typename T> GD BL6<T> se3_jrinv(V6<T> xi) {
  const M3<T> Jw = so3_jrinv(xi.w);
  const M3<T> Q = se3_Q(xi.w, xi.v);
  return {Jw, neg(Jw * Q * Jw), Jw};
}
// rightJacobianPose3 (Pose3utils.cpp:182-189) = Pose3::ExpmapDerivative
template <typename T> GD BL6<T> se3_jr(V6<T> xi) {
  const M3<T> Jw = so3_jr(xi.w);
  return {Jw, se3_Q(xi.w, xi.v), Jw};
}

// Pose3::Logmap
template <typename T> GD V6<T> se3_log(const SE3<T> &g) {
  const V3<T> w = so3_log(g.R);
  const T t = sqrt(dot(w, w));
  if (t < T(1e-10)) return {w, g.t};
  const V3<T> wn = (T(1) / t) * w;
  const T Tan = tan(T(0.5) * t);
  const V3<T> WT = cross(wn, g.t);
  const V3<T> WWT = cross(wn, WT);
  return {w, g.t - (T(0.5) * t) * WT + (T(1) - t / (T(2) * Tan)) * WWT};
}

// Pose3::Expmap
template <typename T> GD SE3<T> se3_exp(V6<T> xi) {
  const M3<T> R = so3_exp(xi.w);
  const T th2 = dot(xi.w, xi.w);
  if (th2 > Eps<T>::v) {
    const V3<T> tpar = dot(xi.w, xi.v) * xi.w;
    const V3<T> wxv = cross(xi.w, xi.v);
    return {R, (T(1) / th2) * (wxv - R * wxv + tpar)};
  }
  return {R, xi.v};
}

// ------------------------------------------------------------------ SE(2), stored as (x, y, theta)

template <typename T> struct SE2 {
  T x, y, th;
};
template <typename T> GD T wrap_pi(T a) { return atan2(sin(a), cos(a)); }

template <typename T> GD SE2<T> se2_between(const SE2<T> &a, const SE2<T> &b) {
  const T c = cos(a.th), s = sin(a.th);
  const T dx = b.x - a.x, dy = b.y - a.y;
  return {c * dx + s * dy, -s * dx + c * dy, b.th - a.th};
}
template <typename T> GD SE2<T> se2_compose(const SE2<T> &a, const SE2<T> &b) {
  const T c = cos(a.th), s = sin(a.th);
  return {a.x + c * b.x - s * b.y, a.y + s * b.x + c * b.y, a.th + b.th};
}
template <typename T> GD SE2<T> se2_inverse(const SE2<T> &a) {
  const T c = cos(a.th), s = sin(a.th);
  return {-(c * a.x + s * a.y), -(-s * a.x + c * a.y), -a.th};
}
// Pose2::AdjointMap
template <typename T> GD M3<T> se2_adjoint(const SE2<T> &g) {
  const T c = cos(g.th), s = sin(g.th);
  return {{c, -s, g.y, s, c, -g.x, T(0), T(0), T(1)}};
}
// Pose2::Logmap -> (vx, vy, w)
template <typename T> GD V3<T> se2_log(const SE2<T> &g) {
  const T w = wrap_pi(g.th);
  if (fabs(w) < T(1e-10)) return {g.x, g.y, w};
  const T c = cos(g.th), s = sin(g.th);
  const T c1 = c - T(1), det = c1 * c1 + s * s;
  const T ux = c * g.x + s * g.y, uy = -s * g.x + c * g.y;
  const T dx = ux - g.x, dy = uy - g.y;
  const T k = w / det;
  return {k * (-dy), k * dx, w};
}
// Pose2::Expmap
template <typename T> GD SE2<T> se2_exp(V3<T> xi) {
  const T w = xi.z;
  if (fabs(w) < T(1e-10)) return {xi.x, xi.y, xi.z};
  const T c = cos(w), s = sin(w);
  const T ox = -xi.y, oy = xi.x;
  const T rx = c * ox - s * oy, ry = s * ox + c * oy;
  return {(ox - rx) / w, (oy - ry) / w, wrap_pi(w)};
}
template <typename T> GD M3<T> se2_ad(V3<T> v) { return {{T(0), -v.z, v.y, v.z, T(0), -v.x, T(0), T(0), T(0)}}; }
// Pose2::ExpmapDerivative
template <typename T> GD M3<T> se2_dexp(V3<T> v) {
  const T al = v.z;
  if (fabs(al) > T(1e-5)) {
    const T sZ = sin(al) / al, c1Z = (cos(al) - T(1)) / al;
    const T v1Z = v.x / al, v2Z = v.y / al;
    return {{sZ, -c1Z, v1Z + v2Z * c1Z - v1Z * sZ, c1Z, sZ, -v1Z * c1Z + v2Z - v2Z * sZ, T(0), T(0), T(1)}};
  }
  return M3<T>::identity() - T(0.5) * se2_ad(v);
}
// Pose2::LogmapDerivative in terms of v = Logmap(p)
template <typename T> GD M3<T> se2_dlog(V3<T> v) {
  const T al = v.z;
  if (fabs(al) > T(1e-5)) {
    const T ai = T(1) / al;
    // halfCotHalfAlpha = 0.5 sin(a) / (1 - cos(a)) evaluated without the cancellation in (1 - cos(a)):
    // identical value, but no 2e-16/a^2 relative error ("Pose2 LogmapDerivative")
    const T hc = T(0.5) / tan(T(0.5) * al);
    return {{al * hc, T(-0.5) * al, v.x * ai - v.x * hc + T(0.5) * v.y, T(0.5) * al, al * hc,
             v.y * ai - T(0.5) * v.x - v.y * hc, T(0), T(0), T(1)}};
  }
  return M3<T>::identity() + T(0.5) * se2_ad(v);
}

}  // namespace gps
