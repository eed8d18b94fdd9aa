// factors.hpp -- per-thread factor evaluation (error + Jacobians) for the batched linearisation kernels.
//
// Every function evaluates ONE factor entirely in registers.  Outputs are unwhitened, in the layout the
// reference's evaluateError() produces; whitening and the write to the HBM row table happen in kernels.hip.
// Reference functions restated here (file:line relative to /root/reference):
//   GaussianProcessPriorLinear<D>::evaluateError   gpslam/gp/GaussianProcessPriorLinear.h:63-83
//   GaussianProcessPriorPose2::evaluateError       gpslam/gp/GaussianProcessPriorPose2.h:58-82
//   GaussianProcessPriorRot3::evaluateError        gpslam/gp/GaussianProcessPriorRot3.h:58-79
//   GaussianProcessPriorPose3::evaluateError       gpslam/gp/GaussianProcessPriorPose3.h:60-98
//   jacobianMethodNumercialDiff                    gpslam/gp/Pose3utils.cpp:167-179
#pragma once

#include <type_traits>

#include "lie.hpp"

namespace gps {

// ROT3_BIAS: the AHRS state of matlab/GPAHRSexample.m:69-202 -- pose slot = (Rot3 x_i, gyroscope bias b_i), velocity slot =
// (angular velocity v_i, three pad components pinned to zero): a 12-wide block, so the chain solver's Pose3 kernels apply
enum Manifold : int { LINEAR2 = 0, LINEAR3 = 1, POSE2 = 2, POSE3 = 3, ROT3 = 4, ROT3_BIAS = 5 };
enum Chart : int { CHART_EXPMAP = 0, CHART_FIRST_ORDER = 1 };

template <int M> struct MTraits;
template <> struct MTraits<LINEAR2> { static constexpr int d = 2, pd = 2; };
template <> struct MTraits<LINEAR3> { static constexpr int d = 3, pd = 3; };
template <> struct MTraits<POSE2> { static constexpr int d = 3, pd = 3; };
template <> struct MTraits<POSE3> { static constexpr int d = 6, pd = 12; };
template <> struct MTraits<ROT3> { static constexpr int d = 3, pd = 9; };
template <> struct MTraits<ROT3_BIAS> { static constexpr int d = 6, pd = 12; };

// ---- conversions between flat register arrays and the Lie types
template <typename T> GD M3<T> as_m3(const T *p) {
  M3<T> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.m[i] = p[i];
  return r;
}
template <typename T> GD SE3<T> as_se3(const T *p) { return {as_m3(p), {p[9], p[10], p[11]}}; }
template <typename T> GD V3<T> as_v3(const T *p) { return {p[0], p[1], p[2]}; }
template <typename T> GD V6<T> as_v6(const T *p) { return {{p[0], p[1], p[2]}, {p[3], p[4], p[5]}}; }
template <typename T> GD void put_m3(const M3<T> &a, T *J, int ld, int r0, int c0) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) J[(r0 + i) * ld + c0 + j] = a.m[3 * i + j];
}
template <typename T> GD void put_bl6(const BL6<T> &a, T *J, int ld, int r0, int c0) {
  put_m3(a.A, J, ld, r0, c0);
  put_m3(a.C, J, ld, r0 + 3, c0);
  put_m3(a.D, J, ld, r0 + 3, c0 + 3);
}

// Jr^-1(xi) * x without forming the 6x6 matrix:  Jr^-1 = [[Jw, 0], [-Jw Q Jw, Jw]] (Pose3utils.cpp:192-200), so
//   top = Jw x_w,   bottom = Jw (x_v - Q (Jw x_w)),
// with Jw a = a + 1/2 w x a + c w x (w x a) (Pose3utils.cpp:215-224) and every product with the skew matrices of
// rightJacobianPose3Q (Pose3utils.cpp:92-113) written as nested cross products.
template <typename T> GD V3<T> so3_jrinv_apply(V3<T> w, V3<T> a) {
  const T th2 = dot(w, w);
  if (th2 <= Eps<T>::v) return a;
  const T th = sqrt(th2);
  const T c = T(1) / th2 - (T(1) + cos(th)) / (T(2) * th * sin(th));
  const V3<T> wa = cross(w, a);
  return a + T(0.5) * wa + c * cross(w, wa);
}
template <typename T> GD V3<T> se3_Q_apply(V3<T> w, V3<T> rho, V3<T> b) {
  const T th = sqrt(dot(w, w));
  T ca, cb, cc;
  if (fabs(th) > T(1e-5)) {
    const T s = sin(th), co = cos(th);
    const T t2 = th * th, t3 = t2 * th, t4 = t3 * th, t5 = t4 * th;
    ca = (th - s) / t3;
    cb = (T(1) - T(0.5) * t2 - co) / t4;
    cc = T(-0.5) * ((T(1) - T(0.5) * t2 - co) / t4 - T(3) * (th - s - t3 / T(6)) / t5);
  } else {
    ca = T(1) / T(6);
    cb = T(1) / T(24);
    cc = T(-0.5) * (T(1) / T(24) + T(3) / T(120));
  }
  const V3<T> Yb = cross(rho, b), Xb = cross(w, b);
  const V3<T> XYb = cross(w, Yb), YXb = cross(rho, Xb), XXb = cross(w, Xb);
  const V3<T> XYXb = cross(w, YXb);
  const V3<T> XXYb = cross(w, XYb), YXXb = cross(rho, XXb);
  const V3<T> XYXXb = cross(w, cross(rho, cross(w, Xb))), XXYXb = cross(w, XYXb);
  return T(-0.5) * Yb + ca * (XYb + YXb - XYXb) + cb * (XXYb + YXXb - T(3) * XYXb) + cc * (XYXXb + XXYXb);
}
template <typename T> GD V6<T> se3_jrinv_apply(V6<T> xi, V6<T> x) {
  const V3<T> top = so3_jrinv_apply(xi.w, x.w);
  return {top, so3_jrinv_apply(xi.w, x.v - se3_Q_apply(xi.w, xi.v, top))};
}

// The same maps with the trigonometric coefficients hoisted: one sincos per distinct rotation vector instead of one
// sin and one cos inside every so3 / Q evaluation (the finite difference below needs 7 distinct angles, not 78
// inlined trig call sites -- two thirds of the K1 instruction stream before this change, profiles/round1_v3).
template <typename T> struct JrK {
  T c;            // so3 Jr^-1: 1/th^2 - (1 + cos th) / (2 th sin th)         (Pose3utils.cpp:215-224)
  T qa, qb, qc;   // rightJacobianPose3Q coefficients incl. the |th| > 1e-5 branch (Pose3utils.cpp:92-113)
  bool ident;     // th^2 <= eps: Jr^-1 = I exactly
};
GD void sincos_t(double x, double *s, double *c) { sincos(x, s, c); }
GD void sincos_t(float x, float *s, float *c) { sincosf(x, s, c); }
template <typename T> GD JrK<T> jr_coefs(V3<T> w) {
  JrK<T> k;
  const T th2 = dot(w, w);
  const T th = sqrt(th2);
  T s, co;
  sincos_t(th, &s, &co);
  k.ident = th2 <= Eps<T>::v;
  k.c = k.ident ? T(0) : T(1) / th2 - (T(1) + co) / (T(2) * th * s);
  if (fabs(th) > T(1e-5)) {
    const T t2 = th * th, t3 = t2 * th, t4 = t3 * th, t5 = t4 * th;
    k.qa = (th - s) / t3;
    k.qb = (T(1) - T(0.5) * t2 - co) / t4;
    k.qc = T(-0.5) * ((T(1) - T(0.5) * t2 - co) / t4 - T(3) * (th - s - t3 / T(6)) / t5);
  } else {
    k.qa = T(1) / T(6);
    k.qb = T(1) / T(24);
    k.qc = T(-0.5) * (T(1) / T(24) + T(3) / T(120));
  }
  return k;
}
template <typename T> GD V3<T> so3_jrinv_apply_k(const JrK<T> &k, V3<T> w, V3<T> a) {
  if (k.ident) return a;
  const V3<T> wa = cross(w, a);
  return a + T(0.5) * wa + k.c * cross(w, wa);
}
template <typename T> GD V3<T> se3_Q_apply_k(const JrK<T> &k, V3<T> w, V3<T> rho, V3<T> b) {
  const V3<T> Yb = cross(rho, b), Xb = cross(w, b);
  const V3<T> XYb = cross(w, Yb), YXb = cross(rho, Xb), XXb = cross(w, Xb);
  const V3<T> XYXb = cross(w, YXb);
  const V3<T> XXYb = cross(w, XYb), YXXb = cross(rho, XXb);
  const V3<T> XYXXb = cross(w, cross(rho, cross(w, Xb))), XXYXb = cross(w, XYXb);
  return T(-0.5) * Yb + k.qa * (XYb + YXb - XYXb) + k.qb * (XXYb + YXXb - T(3) * XYXb) + k.qc * (XYXXb + XXYXb);
}
template <typename T> GD V6<T> se3_jrinv_apply_k(const JrK<T> &k, V6<T> xi, V6<T> x) {
  const V3<T> top = so3_jrinv_apply_k(k, xi.w, x.w);
  return {top, so3_jrinv_apply_k(k, xi.w, x.v - se3_Q_apply_k(k, xi.w, xi.v, top))};
}
// rightJacobianPose3inv as a matrix from hoisted coefficients (same formulas as se3_jrinv / se3_Q in lie.hpp)
template <typename T> GD BL6<T> se3_jrinv_k(const JrK<T> &k, V6<T> xi) {
  const M3<T> X = skew(xi.w), Y = skew(xi.v);
  const M3<T> Jw = k.ident ? M3<T>::identity() : M3<T>::identity() + T(0.5) * X + k.c * (X * X);
  const M3<T> XY = X * Y, YX = Y * X, XYX = X * YX;
  const M3<T> t1 = XY + YX - XYX;
  const M3<T> t2m = X * XY + YX * X - T(3) * XYX;
  const M3<T> t3m = XYX * X + X * XYX;
  const M3<T> Q = T(-0.5) * Y + k.qa * t1 + k.qb * t2m + k.qc * t3m;
  return {Jw, neg(Jw * Q * Jw), Jw};
}
// e_I x b for the unit vector e_I (the products with the zeros and ones of e_I are not left to the compiler: x * 0 is not 0 in
// IEEE arithmetic and is not folded)
template <int I, typename T> GD V3<T> cross_unit(V3<T> b) {
  if (I == 0) return {T(0), -b.z, b.y};
  if (I == 1) return {b.z, T(0), -b.x};
  return {-b.y, b.x, T(0)};
}
// Q(w, e_I) b of se3_Q_apply_k: rho = e_I
template <int I, typename T> GD V3<T> se3_Q_apply_unit(const JrK<T> &k, V3<T> w, V3<T> Xb, V3<T> XXb, V3<T> b) {
  const V3<T> Yb = cross_unit<I>(b);
  const V3<T> XYb = cross(w, Yb), YXb = cross_unit<I>(Xb);
  const V3<T> XYXb = cross(w, YXb);
  const V3<T> XXYb = cross(w, XYb), YXXb = cross_unit<I>(XXb);
  const V3<T> XYXXb = cross(w, YXXb), XXYXb = cross(w, XYXb);
  return T(-0.5) * Yb + k.qa * (XYb + YXb - XYXb) + k.qb * (XXYb + YXXb - T(3) * XYXb) + k.qc * (XYXXb + XXYXb);
}
// Derivative of Jr^-1(xi) x with respect to xi as jacobianMethodNumercialDiff forms it (Pose3utils.cpp:167-200; see
// se3_jrinv_times_x_fd below): the three ROTATIONAL columns are the reference's central difference, h = 1e-6, evaluated with the
// reference's formulas in the reference's order -- its rounding noise (the closed-form coefficients cancel) is part of what the
// parity tests compare.  The three TRANSLATIONAL columns are the same difference quotient in closed form (round 6): Jr^-1(xi) x
// is affine in rho -- lower half Jw (x_v - Q(w, rho) Jw x_w), Q linear in rho -- so (f(rho + h e) - f(rho - h e)) / 2h IS
// -Jw Q(w, e) Jw x_w; the quotient differed from it by its own rounding (1e-16 / 2h = 5e-11 of |f|, three orders below the
// 1e-7 the rotational columns are held to) and cost six full evaluations (890 of a K1 thread's ~5000 instructions) against 250.
// k0 = jr_coefs(xi.w).
template <typename T> GD BL6<T> se3_jrinv_times_x_fd_k(const JrK<T> &k0, V6<T> xi, V6<T> x) {
  const T h = T(1e-6);
  const T s = T(1) / (T(2) * h);
  BL6<T> D;
  D.A = M3<T>::zero();
  D.C = M3<T>::zero();
  D.D = M3<T>::zero();
#pragma unroll
  for (int i = 0; i < 3; i++) {
    V6<T> xp = xi, xn = xi;
    if (i == 0) { xp.w.x += h; xn.w.x -= h; }
    if (i == 1) { xp.w.y += h; xn.w.y -= h; }
    if (i == 2) { xp.w.z += h; xn.w.z -= h; }
    const JrK<T> kp = jr_coefs(xp.w), kn = jr_coefs(xn.w);
    const V6<T> col = s * (se3_jrinv_apply_k(kp, xp, x) - se3_jrinv_apply_k(kn, xn, x));
    D.A.m[0 + i] = col.w.x; D.A.m[3 + i] = col.w.y; D.A.m[6 + i] = col.w.z;
    D.C.m[0 + i] = col.v.x; D.C.m[3 + i] = col.v.y; D.C.m[6 + i] = col.v.z;
  }
  {
    const V3<T> top = so3_jrinv_apply_k(k0, xi.w, x.w);
    const V3<T> Xb = cross(xi.w, top), XXb = cross(xi.w, Xb);
    const V3<T> c0 = so3_jrinv_apply_k(k0, xi.w, se3_Q_apply_unit<0>(k0, xi.w, Xb, XXb, top));
    const V3<T> c1 = so3_jrinv_apply_k(k0, xi.w, se3_Q_apply_unit<1>(k0, xi.w, Xb, XXb, top));
    const V3<T> c2 = so3_jrinv_apply_k(k0, xi.w, se3_Q_apply_unit<2>(k0, xi.w, Xb, XXb, top));
    D.D.m[0] = -c0.x; D.D.m[3] = -c0.y; D.D.m[6] = -c0.z;
    D.D.m[1] = -c1.x; D.D.m[4] = -c1.y; D.D.m[7] = -c1.z;
    D.D.m[2] = -c2.x; D.D.m[5] = -c2.y; D.D.m[8] = -c2.z;
  }
  return D;
}

// =================================================================== fp32 arithmetic (GPSLAM_FP32)
// The reference's formulas cannot simply be instantiated for float:
//  * jacobianMethodNumercialDiff's step h = 1e-6 (Pose3utils.h:57, Pose3utils.cpp:167-179) is below float's resolution
//    of the rotation vector (eps32 = 6e-8 relative): the central difference is noise.  The fp32 path takes the EXACT
//    derivative d(Jr^-1(xi) x)/d xi instead: forward-mode differentiation (Dual3: value + 3 partials with respect to the
//    rotation vector) through the very same nested-cross-product evaluation, and the closed form for the translational
//    half (Q is linear in rho: column k of d/d rho is -Jw Q(w, e_k) Jw x_w).
//  * the closed-form coefficients (th - sin th)/th^3, (1 - th^2/2 - cos th)/th^4, ... (Pose3utils.cpp:98-104, :219-223)
//    cancel catastrophically at the th = 0.01 .. 0.1 rad of consecutive chain states: in fp32 the numerator of the th^-4
//    coefficient is all rounding error.  Below th^2 = 0.25 they are evaluated by their Taylor series in th^2 (5 terms,
//    truncation < 1e-9), above by the closed forms; the reference's thresholds 1e-5 / 1e-10 play no role in fp32.
template <typename T> struct Dual3 {
  T v, d[3];
  GD Dual3() : v(T(0)), d{T(0), T(0), T(0)} {}
  template <typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>
  GD Dual3(U x) : v(T(x)), d{T(0), T(0), T(0)} {}
  GD Dual3(T x, int k) : v(x), d{k == 0 ? T(1) : T(0), k == 1 ? T(1) : T(0), k == 2 ? T(1) : T(0)} {}
};
template <typename T> GD Dual3<T> operator+(Dual3<T> a, Dual3<T> b) { Dual3<T> r; r.v = a.v + b.v; for (int i = 0; i < 3; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
template <typename T> GD Dual3<T> operator-(Dual3<T> a, Dual3<T> b) { Dual3<T> r; r.v = a.v - b.v; for (int i = 0; i < 3; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
template <typename T> GD Dual3<T> operator-(Dual3<T> a) { Dual3<T> r; r.v = -a.v; for (int i = 0; i < 3; i++) r.d[i] = -a.d[i]; return r; }
template <typename T> GD Dual3<T> operator*(Dual3<T> a, Dual3<T> b) { Dual3<T> r; r.v = a.v * b.v; for (int i = 0; i < 3; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <typename T> GD Dual3<T> operator/(Dual3<T> a, Dual3<T> b) {
  Dual3<T> r;
  const T inv = T(1) / b.v;
  r.v = a.v * inv;
  for (int i = 0; i < 3; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
template <typename T> GD Dual3<T> sqrt(Dual3<T> a) { Dual3<T> r; r.v = sqrt(a.v); const T h = T(0.5) / r.v; for (int i = 0; i < 3; i++) r.d[i] = a.d[i] * h; return r; }
template <typename T> GD Dual3<T> sin(Dual3<T> a) { Dual3<T> r; r.v = sin(a.v); const T c = cos(a.v); for (int i = 0; i < 3; i++) r.d[i] = a.d[i] * c; return r; }
template <typename T> GD Dual3<T> cos(Dual3<T> a) { Dual3<T> r; r.v = cos(a.v); const T s = -sin(a.v); for (int i = 0; i < 3; i++) r.d[i] = a.d[i] * s; return r; }
GD float dual_val(float x) { return x; }
GD double dual_val(double x) { return x; }
template <typename T> GD T dual_val(Dual3<T> x) { return x.v; }

// coefficients of Jr^-1 and of Q as smooth functions of u = th^2 (S = float, double or a Dual3 of them)
template <typename S> GD JrK<S> jr_coefs_smooth(V3<S> w) {
  JrK<S> k;
  k.ident = false;
  const S u = dot(w, w);
  if (dual_val(u) < 0.25) {
    // (1 - (th/2) cot(th/2)) / th^2;  (th - sin th)/th^3;  (1 - th^2/2 - cos th)/th^4;  (th - sin th - th^3/6)/th^5
    k.c = S(1.0 / 12) + u * (S(1.0 / 720) + u * (S(1.0 / 30240) + u * (S(1.0 / 1209600) + u * S(1.0 / 47900160))));
    k.qa = S(1.0 / 6) - u * (S(1.0 / 120) - u * (S(1.0 / 5040) - u * (S(1.0 / 362880) - u * S(1.0 / 39916800))));
    k.qb = S(-1.0 / 24) + u * (S(1.0 / 720) - u * (S(1.0 / 40320) - u * (S(1.0 / 3628800) - u * S(1.0 / 479001600))));
    const S qd = S(-1.0 / 120) + u * (S(1.0 / 5040) - u * (S(1.0 / 362880) - u * (S(1.0 / 39916800) - u * S(1.0 / 6227020800.0))));
    k.qc = S(-0.5) * (k.qb - S(3.0) * qd);
  } else {
    const S th = sqrt(u);
    const S s = sin(th), co = cos(th);
    const S t3 = u * th, t4 = u * u, t5 = t4 * th;
    k.c = S(1.0) / u - (S(1.0) + co) / (S(2.0) * th * s);
    k.qa = (th - s) / t3;
    k.qb = (S(1.0) - S(0.5) * u - co) / t4;
    k.qc = S(-0.5) * (k.qb - S(3.0) * (th - s - t3 / S(6.0)) / t5);
  }
  return k;
}
// fp32: the blended coefficients replace the reference's closed forms + thresholds
GD JrK<float> jr_coefs(V3<float> w) { return jr_coefs_smooth<float>(w); }

// d(Jr^-1(xi) x)/d xi, exact (fp32 path; the fp64 path keeps the reference's central difference below)
GD BL6<float> se3_jrinv_times_x_fd_k(const JrK<float> &k0, V6<float> xi, V6<float> x) {
  typedef Dual3<float> D;
  const V3<D> w = {D(xi.w.x, 0), D(xi.w.y, 1), D(xi.w.z, 2)};
  const V3<D> rho = {D(xi.v.x), D(xi.v.y), D(xi.v.z)};
  const V3<D> xw = {D(x.w.x), D(x.w.y), D(x.w.z)}, xv = {D(x.v.x), D(x.v.y), D(x.v.z)};
  const JrK<D> k = jr_coefs_smooth<D>(w);
  const V3<D> top = so3_jrinv_apply_k(k, w, xw);
  const V3<D> bot = so3_jrinv_apply_k(k, w, xv - se3_Q_apply_k(k, w, rho, top));
  BL6<float> M;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    M.A.m[0 + j] = top.x.d[j]; M.A.m[3 + j] = top.y.d[j]; M.A.m[6 + j] = top.z.d[j];
    M.C.m[0 + j] = bot.x.d[j]; M.C.m[3 + j] = bot.y.d[j]; M.C.m[6 + j] = bot.z.d[j];
  }
  const V3<float> tv = {top.x.v, top.y.v, top.z.v};
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const V3<float> ej = {j == 0 ? 1.f : 0.f, j == 1 ? 1.f : 0.f, j == 2 ? 1.f : 0.f};
    const V3<float> col = so3_jrinv_apply_k(k0, xi.w, -se3_Q_apply_k(k0, xi.w, ej, tv));
    M.D.m[0 + j] = col.x; M.D.m[3 + j] = col.y; M.D.m[6 + j] = col.z;
  }
  return M;
}

// d( Jr^-1(xi) * x ) / d xi by central differences with h = 1e-6: the construction of
// jacobianMethodNumercialDiff(rightJacobianPose3inv, xi, x) (Pose3utils.cpp:167-179, default dxi Pose3utils.h:57),
// column i = (Jr^-1(xi + h e_i) x - Jr^-1(xi - h e_i) x) / (2h).  The reference subtracts the two 6x6 matrices
// before multiplying by x; here the two matrix-vector products are formed directly (a third of the flops, no 6x6
// temporaries) -- the same number up to rounding of order eps/h = 1e-10, which is the noise floor of this finite
// difference in the reference as well.  The result is block lower-triangular: perturbing rho leaves the rotation
// block untouched, so the top-right 3x3 block is an exact zero in the reference too.
template <typename T> GD BL6<T> se3_jrinv_times_x_fd(V6<T> xi, V6<T> x) {
  // (round 6: one function for every caller -- the rotational columns as described, the translational ones in closed form)
  return se3_jrinv_times_x_fd_k(jr_coefs(xi.w), xi, x);
}

// ---- world-frame velocity parameterisation: the *Pose3VW factor family (GaussianProcessPriorPose3VW.h,
// GaussianProcessInterpolatorPose3VW.h, GPInterpolatedGPSFactorPose3VW.h).  A state carries s = [v; w], world-frame
// translational and rotational velocity; every VW factor is the body-velocity factor evaluated at
// Vb = convertVWtoVb(v, w, pose) = [R^T w; R^T v] (Pose3utils.cpp:47-64), and its Jacobians follow by the chain rule
//   dVb/dpose = [[skew(R^T w), 0], [skew(R^T v), 0]],   dVb/d(v, w) = [[0, R^T], [R^T, 0]]        (:56-60)
// -- exactly the H1p / H1v / H1w terms of GaussianProcessPriorPose3VW.h:97-115.
template <typename T> GD void vw_to_vb(const T *pose, const T *s, T *vb) {
  const M3<T> R = as_m3(pose);
  const V3<T> wb = tmul(R, V3<T>{s[3], s[4], s[5]}), tb = tmul(R, V3<T>{s[0], s[1], s[2]});
  vb[0] = wb.x; vb[1] = wb.y; vb[2] = wb.z; vb[3] = tb.x; vb[4] = tb.y; vb[5] = tb.z;
}
// one state's segment of a Jacobian row, a[12] = [d/dpose (6) | d/dVb (6)]  ->  [d/dpose | d/dv (3) | d/dw (3)]
template <typename T> GD void vw_row_transform(const T *pose, const T *vb, T *a) {
  const M3<T> R = as_m3(pose);
  const V3<T> aw = {a[6], a[7], a[8]}, at = {a[9], a[10], a[11]};
  const V3<T> wb = {vb[0], vb[1], vb[2]}, tb = {vb[3], vb[4], vb[5]};
  const V3<T> dp = cross(aw, wb) + cross(at, tb);          // x^T skew(u) = (x cross u)^T
  a[0] += dp.x; a[1] += dp.y; a[2] += dp.z;
  const V3<T> nv = R * at, nw = R * aw;                    // x^T R^T = (R x)^T
  a[6] = nv.x; a[7] = nv.y; a[8] = nv.z; a[9] = nw.x; a[10] = nw.y; a[11] = nw.z;
}

// GP prior, unwhitened.  x1 = (p1, v1), x2 = (p2, v2) as flat arrays (pose layout of include/gpslam_hip.h).
// Outputs: e[2d]; if JAC: Jt[d][4d] = top d rows of [H1 H2 | H3 H4], Jb[d][4d] = bottom d rows.
template <typename T, int MF, bool JAC> struct GpPrior;

template <typename T, int D, bool JAC> struct GpPriorLinear {
  static GD void eval(const T *p1, const T *v1, const T *p2, const T *v2, T dt, T *e, T *Jt, T *Jb) {
#pragma unroll
    for (int i = 0; i < D; i++) {
      e[i] = p1[i] + dt * v1[i] - p2[i];
      e[D + i] = v1[i] - v2[i];
    }
    if (JAC) {
#pragma unroll
      for (int i = 0; i < D * 4 * D; i++) { Jt[i] = T(0); Jb[i] = T(0); }
#pragma unroll
      for (int i = 0; i < D; i++) {
        Jt[i * 4 * D + i] = T(1);           // H1 = [I; 0]
        Jt[i * 4 * D + D + i] = dt;         // H2 = [dt I; I]
        Jb[i * 4 * D + D + i] = T(1);
        Jt[i * 4 * D + 2 * D + i] = T(-1);  // H3 = [-I; 0]
        Jb[i * 4 * D + 3 * D + i] = T(-1);  // H4 = [0; -I]
      }
    }
  }
};
template <typename T, bool JAC> struct GpPrior<T, LINEAR2, JAC> : GpPriorLinear<T, 2, JAC> {};
template <typename T, bool JAC> struct GpPrior<T, LINEAR3, JAC> : GpPriorLinear<T, 3, JAC> {};

// shared tail of the d = 3 groups: e = [r - v1 dt; v2 - v1], H1 = [J1; 0], H2 = [-dt I; -I], H3 = [J3; 0], H4 = [0; I]
template <typename T, bool JAC>
GD void gp3_tail(V3<T> r, const T *v1, const T *v2, T dt, const M3<T> &J1, const M3<T> &J3, T *e, T *Jt, T *Jb) {
  e[0] = r.x - v1[0] * dt; e[1] = r.y - v1[1] * dt; e[2] = r.z - v1[2] * dt;
#pragma unroll
  for (int i = 0; i < 3; i++) e[3 + i] = v2[i] - v1[i];
  if (JAC) {
#pragma unroll
    for (int i = 0; i < 36; i++) { Jt[i] = T(0); Jb[i] = T(0); }
    put_m3(J1, Jt, 12, 0, 0);
    put_m3(J3, Jt, 12, 0, 6);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      Jt[i * 12 + 3 + i] = -dt;
      Jb[i * 12 + 3 + i] = T(-1);
      Jb[i * 12 + 9 + i] = T(1);
    }
  }
}

GD BL6<float> se3_jrinv_times_x_fd(V6<float> xi, V6<float> x) { return se3_jrinv_times_x_fd_k(jr_coefs(xi.w), xi, x); }

template <typename T, bool JAC> struct GpPrior<T, POSE2, JAC> {
  static GD void eval(const T *p1, const T *v1, const T *p2, const T *v2, T dt, T *e, T *Jt, T *Jb) {
    const SE2<T> a = {p1[0], p1[1], p1[2]}, b = {p2[0], p2[1], p2[2]};
    const SE2<T> h = se2_between(a, b);
    const V3<T> r = se2_log(h);
    M3<T> J1 = M3<T>::zero(), J3 = M3<T>::zero();
    if (JAC) {
      // d(a^-1 b)/da = Ad(b^-1) * (-Ad(a)) = -Ad(h^-1);  d/db = I
      J3 = se2_dlog(r);
      J1 = neg(J3 * se2_adjoint(se2_inverse(h)));
    }
    gp3_tail<T, JAC>(r, v1, v2, dt, J1, J3, e, Jt, Jb);
  }
};

template <typename T, bool JAC> struct GpPrior<T, ROT3, JAC> {
  static GD void eval(const T *p1, const T *v1, const T *p2, const T *v2, T dt, T *e, T *Jt, T *Jb) {
    const M3<T> R1 = as_m3(p1), R2 = as_m3(p2);
    const M3<T> h = transpose(R1) * R2;
    const V3<T> r = so3_log(h);
    M3<T> J1 = M3<T>::zero(), J3 = M3<T>::zero();
    if (JAC) {
      // Hcomp1 * Hinv = R2^T * (-R1) = -h^T
      J3 = so3_jrinv(r);
      J1 = neg(J3 * transpose(h));
    }
    gp3_tail<T, JAC>(r, v1, v2, dt, J1, J3, e, Jt, Jb);
  }
};

// GaussianProcessPriorRot3 on the (x, v) part of the AHRS state.  The bias has no GP prior in the recipe (it is tied by
// BetweenFactorVector, GPAHRSexample.m:143); the three pad components are pinned by unit rows (pad_1 in the top half,
// pad_2 in the bottom half), which keeps the 12-wide block non-singular without any extra factor set.  The handle's
// Qc is diag(Qc_rot, I_3) (set_qc pads it), so the whitening leaves the pad rows decoupled from the rotation rows.
template <typename T, bool JAC> struct GpPrior<T, ROT3_BIAS, JAC> {
  static GD void eval(const T *p1, const T *v1, const T *p2, const T *v2, T dt, T *e, T *Jt, T *Jb) {
    T e3[6], Jt3[JAC ? 36 : 1], Jb3[JAC ? 36 : 1];
    GpPrior<T, ROT3, JAC>::eval(p1, v1, p2, v2, dt, e3, Jt3, Jb3);
#pragma unroll
    for (int i = 0; i < 3; i++) { e[i] = e3[i]; e[3 + i] = v1[3 + i]; e[6 + i] = e3[3 + i]; e[9 + i] = v2[3 + i]; }
    if (JAC) {
#pragma unroll
      for (int i = 0; i < 6 * 24; i++) { Jt[i] = T(0); Jb[i] = T(0); }
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            Jt[r * 24 + 12 * s + c] = Jt3[r * 12 + 6 * s + c];           // d/d theta_s
            Jt[r * 24 + 12 * s + 6 + c] = Jt3[r * 12 + 6 * s + 3 + c];   // d/d omega_s
            Jb[r * 24 + 12 * s + c] = Jb3[r * 12 + 6 * s + c];
            Jb[r * 24 + 12 * s + 6 + c] = Jb3[r * 12 + 6 * s + 3 + c];
          }
#pragma unroll
      for (int c = 0; c < 3; c++) { Jt[(3 + c) * 24 + 9 + c] = T(1); Jb[(3 + c) * 24 + 21 + c] = T(1); }
    }
  }
};

template <typename T, bool JAC> struct GpPrior<T, POSE3, JAC> {
  static GD void eval(const T *p1, const T *v1, const T *p2, const T *v2, T dt, T *e, T *Jt, T *Jb) {
    const SE3<T> a = as_se3(p1), b = as_se3(p2);
    const SE3<T> h = se3_between(a, b);
    const V6<T> r = se3_log(h);                 // GaussianProcessPriorPose3.h:72
    const BL6<T> Jinv = se3_jrinv(r);           // :76
    const V6<T> u1 = as_v6(v1), u2 = as_v6(v2);
    const V6<T> top = r - dt * u1;              // :97
    const V6<T> bot = Jinv * u2 - u1;
    e[0] = top.w.x; e[1] = top.w.y; e[2] = top.w.z; e[3] = top.v.x; e[4] = top.v.y; e[5] = top.v.z;
    e[6] = bot.w.x; e[7] = bot.w.y; e[8] = bot.w.z; e[9] = bot.v.x; e[10] = bot.v.y; e[11] = bot.v.z;
    if (JAC) {
#pragma unroll
      for (int i = 0; i < 6 * 24; i++) { Jt[i] = T(0); Jb[i] = T(0); }
      // Hlogmap = LogmapDerivative = Jr^-1(r) (same closed form, Pose3utils.cpp:192-200)
      const BL6<T> J_Ti1 = Jinv;                                   // Hlogmap * Hcomp2, Hcomp2 = I   (:89)
      const BL6<T> J_Ti = neg(Jinv * se3_adjoint(se3_inverse(h))); // Hlogmap * Hcomp1 * Hinv = -Hlog Ad(h^-1) (:80)
      const BL6<T> FD = se3_jrinv_times_x_fd(r, u2);               // (:81, :90) -- computed once, used twice
      put_bl6(J_Ti, Jt, 24, 0, 0);                                 // H1 = [J_Ti; FD J_Ti]
      put_bl6(FD * J_Ti, Jb, 24, 0, 0);
      put_bl6(J_Ti1, Jt, 24, 0, 12);                               // H3 = [J_Ti1; FD J_Ti1]
      put_bl6(FD * J_Ti1, Jb, 24, 0, 12);
      put_bl6(Jinv, Jb, 24, 0, 18);                                // H4 = [0; Jinv] (:95)
#pragma unroll
      for (int i = 0; i < 6; i++) {                                // H2 = [-dt I; -I] (:86)
        Jt[i * 24 + 6 + i] = -dt;
        Jb[i * 24 + 6 + i] = T(-1);
      }
    }
  }
};

// ------------------------------------------------------------------ charts (retract / local at the origin)

// v = Local_origin(h) and its derivative for the pose manifolds; used by PriorFactor / BetweenFactor.
// Flat pose in, tangent out.  HL: d x d row-major (only if JAC).
template <typename T, int MF, bool JAC> struct ChartLocal;

template <typename T, bool JAC> struct ChartLocal<T, POSE3, JAC> {
  static GD void eval(const SE3<T> &h, int, T *v, T *HL) {
    const V6<T> xi = se3_log(h);
    v[0] = xi.w.x; v[1] = xi.w.y; v[2] = xi.w.z; v[3] = xi.v.x; v[4] = xi.v.y; v[5] = xi.v.z;
    if (JAC) {
#pragma unroll
      for (int i = 0; i < 36; i++) HL[i] = T(0);
      put_bl6(se3_jrinv(xi), HL, 6, 0, 0);
    }
  }
};

// PriorFactor<Pose>: e = Local(prior, x) = Local_origin(prior^-1 x), H = dLocal.
// BetweenFactor<Pose>: hx = x1^-1 x2, e = Local_origin(measured^-1 hx), H1 = -HL Ad(hx^-1), H2 = HL.
template <typename T, int MF, bool JAC> struct PoseFactors;

template <typename T, int D, bool JAC> struct PoseFactorsLinear {
  static GD void prior(const T *pr, const T *x, int, T *e, T *H) {
#pragma unroll
    for (int i = 0; i < D; i++) e[i] = x[i] - pr[i];
    if (JAC) {
#pragma unroll
      for (int i = 0; i < D * D; i++) H[i] = T(0);
#pragma unroll
      for (int i = 0; i < D; i++) H[i * D + i] = T(1);
    }
  }
  static GD void between(const T *m, const T *x1, const T *x2, int, T *e, T *H1, T *H2) {
#pragma unroll
    for (int i = 0; i < D; i++) e[i] = (x2[i] - x1[i]) - m[i];
    if (JAC) {
#pragma unroll
      for (int i = 0; i < D * D; i++) { H1[i] = T(0); H2[i] = T(0); }
#pragma unroll
      for (int i = 0; i < D; i++) { H1[i * D + i] = T(-1); H2[i * D + i] = T(1); }
    }
  }
  static GD void retract(const T *x, const T *dlt, int, T *out) {
#pragma unroll
    for (int i = 0; i < D; i++) out[i] = x[i] + dlt[i];
  }
};
template <typename T, bool JAC> struct PoseFactors<T, LINEAR2, JAC> : PoseFactorsLinear<T, 2, JAC> {};
template <typename T, bool JAC> struct PoseFactors<T, LINEAR3, JAC> : PoseFactorsLinear<T, 3, JAC> {};

template <typename T, bool JAC> struct PoseFactors<T, POSE2, JAC> {
  static GD void local0(const SE2<T> &h, int chart, T *v, M3<T> &HL) {
    if (chart == CHART_FIRST_ORDER) {
      v[0] = h.x; v[1] = h.y; v[2] = wrap_pi(h.th);
      if (JAC) {
        const T c = cos(h.th), s = sin(h.th);
        HL = {{c, -s, T(0), s, c, T(0), T(0), T(0), T(1)}};   // = h.rotation().matrix(): d(x, y, th)/d(delta) under h o (dx, dy, dth)
      }
    } else {
      const V3<T> xi = se2_log(h);
      v[0] = xi.x; v[1] = xi.y; v[2] = xi.z;
      if (JAC) HL = se2_dlog(xi);
    }
  }
  static GD void prior(const T *pr, const T *x, int chart, T *e, T *H) {
    const SE2<T> h = se2_between<T>({pr[0], pr[1], pr[2]}, {x[0], x[1], x[2]});
    M3<T> HL = M3<T>::identity();
    local0(h, chart, e, HL);
    if (JAC) put_m3(HL, H, 3, 0, 0);
  }
  static GD void between(const T *m, const T *x1, const T *x2, int chart, T *e, T *H1, T *H2) {
    const SE2<T> hx = se2_between<T>({x1[0], x1[1], x1[2]}, {x2[0], x2[1], x2[2]});
    const SE2<T> h = se2_between<T>({m[0], m[1], m[2]}, hx);
    M3<T> HL = M3<T>::identity();
    local0(h, chart, e, HL);
    if (JAC) {
      put_m3(neg(HL * se2_adjoint(se2_inverse(hx))), H1, 3, 0, 0);
      put_m3(HL, H2, 3, 0, 0);
    }
  }
  static GD void retract(const T *x, const T *dlt, int chart, T *out) {
    SE2<T> ex;
    if (chart == CHART_FIRST_ORDER) ex = {dlt[0], dlt[1], dlt[2]};
    else ex = se2_exp<T>({dlt[0], dlt[1], dlt[2]});
    const SE2<T> r = se2_compose<T>({x[0], x[1], x[2]}, ex);
    out[0] = r.x; out[1] = r.y; out[2] = r.th;
  }
};

template <typename T, bool JAC> struct PoseFactors<T, ROT3, JAC> {
  static GD void prior(const T *pr, const T *x, int, T *e, T *H) {
    const M3<T> h = transpose(as_m3(pr)) * as_m3(x);
    const V3<T> w = so3_log(h);
    e[0] = w.x; e[1] = w.y; e[2] = w.z;
    if (JAC) put_m3(so3_jrinv(w), H, 3, 0, 0);
  }
  static GD void between(const T *m, const T *x1, const T *x2, int, T *e, T *H1, T *H2) {
    const M3<T> hx = transpose(as_m3(x1)) * as_m3(x2);
    const M3<T> h = transpose(as_m3(m)) * hx;
    const V3<T> w = so3_log(h);
    e[0] = w.x; e[1] = w.y; e[2] = w.z;
    if (JAC) {
      const M3<T> HL = so3_jrinv(w);
      put_m3(neg(HL * transpose(hx)), H1, 3, 0, 0);
      put_m3(HL, H2, 3, 0, 0);
    }
  }
  // chart: CHART_EXPMAP R Exp(w) (GTSAM >= 4.1, GTSAM_ROT3_EXPMAP) or CHART_FIRST_ORDER = GTSAM 4.0's default R Cayley(w)
  static GD void retract(const T *x, const T *dlt, int chart, T *out) {
    const V3<T> w = {dlt[0], dlt[1], dlt[2]};
    const M3<T> r = as_m3(x) * (chart == CHART_FIRST_ORDER ? so3_cayley<T>(w) : so3_exp<T>(w));
#pragma unroll
    for (int i = 0; i < 9; i++) out[i] = r.m[i];
  }
};

template <typename T, bool JAC> struct PoseFactors<T, POSE3, JAC> {
  static GD void prior(const T *pr, const T *x, int chart, T *e, T *H) {
    ChartLocal<T, POSE3, JAC>::eval(se3_between(as_se3(pr), as_se3(x)), chart, e, H);
  }
  static GD void between(const T *m, const T *x1, const T *x2, int chart, T *e, T *H1, T *H2) {
    const SE3<T> hx = se3_between(as_se3(x1), as_se3(x2));
    const SE3<T> h = se3_between(as_se3(m), hx);
    const V6<T> xi = se3_log(h);
    e[0] = xi.w.x; e[1] = xi.w.y; e[2] = xi.w.z; e[3] = xi.v.x; e[4] = xi.v.y; e[5] = xi.v.z;
    if (JAC) {
      const BL6<T> HL = se3_jrinv(xi);
#pragma unroll
      for (int i = 0; i < 36; i++) { H1[i] = T(0); H2[i] = T(0); }
      put_bl6(neg(HL * se3_adjoint(se3_inverse(hx))), H1, 6, 0, 0);
      put_bl6(HL, H2, 6, 0, 0);
    }
  }
  // chart: CHART_EXPMAP T Expmap(xi) (GTSAM >= 4.1, GTSAM_POSE3_EXPMAP) or CHART_FIRST_ORDER = GTSAM 4.0's default
  // Pose3::ChartAtOrigin::Retract: T * Pose3(Rot3::Retract(w), v), i.e. (R Cayley(w), t + R v)
  static GD void retract(const T *x, const T *dlt, int chart, T *out) {
    const V6<T> xi = as_v6(dlt);
    const SE3<T> ex = (chart == CHART_FIRST_ORDER) ? SE3<T>{so3_cayley(xi.w), xi.v} : se3_exp(xi);
    const SE3<T> r = se3_compose(as_se3(x), ex);
#pragma unroll
    for (int i = 0; i < 9; i++) out[i] = r.R.m[i];
    out[9] = r.t.x; out[10] = r.t.y; out[11] = r.t.z;
  }
};

// SO(3) x R^3 (rotation, bias): PriorFactorRot3 / PriorFactorVector / BetweenFactorVector of the AHRS recipe
// (GPAHRSexample.m:118-121, :143) are the two halves of these 6-row factors; the half that is not wanted gets an infinite
// sigma (weight zero).
template <typename T, bool JAC> struct PoseFactors<T, ROT3_BIAS, JAC> {
  static GD void prior(const T *pr, const T *x, int chart, T *e, T *H) {
    T H3[JAC ? 9 : 1];
    PoseFactors<T, ROT3, JAC>::prior(pr, x, chart, e, H3);
#pragma unroll
    for (int i = 0; i < 3; i++) e[3 + i] = x[9 + i] - pr[9 + i];
    if (JAC) {
#pragma unroll
      for (int i = 0; i < 36; i++) H[i] = T(0);
#pragma unroll
      for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) H[r * 6 + c] = H3[r * 3 + c];
        H[(3 + r) * 6 + 3 + r] = T(1);
      }
    }
  }
  static GD void between(const T *m, const T *x1, const T *x2, int chart, T *e, T *H1, T *H2) {
    T A[JAC ? 9 : 1], B[JAC ? 9 : 1];
    PoseFactors<T, ROT3, JAC>::between(m, x1, x2, chart, e, A, B);
#pragma unroll
    for (int i = 0; i < 3; i++) e[3 + i] = (x2[9 + i] - x1[9 + i]) - m[9 + i];
    if (JAC) {
#pragma unroll
      for (int i = 0; i < 36; i++) { H1[i] = T(0); H2[i] = T(0); }
#pragma unroll
      for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) { H1[r * 6 + c] = A[r * 3 + c]; H2[r * 6 + c] = B[r * 3 + c]; }
        H1[(3 + r) * 6 + 3 + r] = T(-1);
        H2[(3 + r) * 6 + 3 + r] = T(1);
      }
    }
  }
  static GD void retract(const T *x, const T *dlt, int chart, T *out) {
    PoseFactors<T, ROT3, false>::retract(x, dlt, chart, out);
#pragma unroll
    for (int i = 0; i < 3; i++) out[9 + i] = x[9 + i] + dlt[3 + i];
  }
};

// gtsam::AHRSFactor::evaluateError (GTSAM 4.0 gtsam/navigation/AHRSFactor.cpp -- third-party, not under /root/reference;
// call site matlab/GPAHRSexample.m:131-137).  prm = [deltaRij (9) | delRdelBiasOmega (9) | biasHat (3) | deltaTij |
// omegaCoriolis (3)] of the PreintegratedAhrsMeasurements:
//   omega   = Log(deltaRij Exp(delRdelBiasOmega (bias - biasHat)))           PreintegratedAhrsMeasurements::predict
//   c       = Ri^T omegaCoriolis deltaTij                                     integrateCoriolis
//   fR      = Log(Exp(omega - c)^T Ri^T Rj)
//   H1 = Jr^-1(fR) (-(Ri^T Rj)^T + fRrot^T Jr(omega - c) [c]x),  H2 = Jr^-1(fR),
//   H3 = -Jr^-1(fR) fRrot^T Jr(omega - c) Jr^-1(omega) Jr(delRdelBiasOmega db) delRdelBiasOmega
template <typename T, bool JAC>
GD void ahrs_factor(const M3<T> &Ri, const M3<T> &Rj, V3<T> bias, const T *prm, T *e, M3<T> &H1, M3<T> &H2, M3<T> &H3) {
  const M3<T> dR = as_m3(prm), D = as_m3(prm + 9);
  const V3<T> binc = bias - V3<T>{prm[18], prm[19], prm[20]};
  const V3<T> bio = D * binc;
  const V3<T> om = so3_log(dR * so3_exp(bio));
  const V3<T> cor = prm[21] * tmul(Ri, V3<T>{prm[22], prm[23], prm[24]});
  const V3<T> com = om - cor;
  const M3<T> aR = transpose(Ri) * Rj;
  const M3<T> fRrot = transpose(so3_exp(com)) * aR;
  const V3<T> fR = so3_log(fRrot);
  e[0] = fR.x; e[1] = fR.y; e[2] = fR.z;
  if (JAC) {
    const M3<T> Dexp = so3_jr(com), Dlog = so3_jrinv(fR);
    const M3<T> fRt = transpose(fRrot);
    H1 = Dlog * (neg(transpose(aR)) + fRt * (Dexp * skew(cor)));
    H2 = Dlog;
    H3 = neg(Dlog * (fRt * (Dexp * (so3_jrinv(om) * (so3_jr(bio) * D)))));
  }
}

}  // namespace gps

// =====================================================================================================
// GP interpolators and the measurement factors that sit on an interpolated pose.
//   GaussianProcessInterpolatorLinear<D>::interpolatePose   gpslam/gp/GaussianProcessInterpolatorLinear.h:70-90
//   GaussianProcessInterpolatorPose2::interpolatePose       gpslam/gp/GaussianProcessInterpolatorPose2.h:56-89
//   GaussianProcessInterpolatorRot3::interpolatePose        gpslam/gp/GaussianProcessInterpolatorRot3.h:56-86
//   GaussianProcessInterpolatorPose3::interpolatePose       gpslam/gp/GaussianProcessInterpolatorPose3.h:57-105
// Lambda(tau) and Psi(tau) (gpslam/gp/GPutils.h:54-71) are Kronecker products P (x) (Qc Qc^-1) = P (x) I of a
// 2x2 matrix with the identity for ANY Qc, so only the first block row is needed and it is four scalars:
//   Lambda_1 = [l11 I, l12 I],  Psi_1 = [p11 I, p12 I]   (computed on the host per factor, see api.hip).
// =====================================================================================================
namespace gps {

template <typename T> struct ICoef { T l11, l12, p11, p12; };

// row-vector (1x3) times 3x3
template <typename T> GD V3<T> rowmul(V3<T> a, const M3<T> &m) {
  return {a.x * m.m[0] + a.y * m.m[3] + a.z * m.m[6], a.x * m.m[1] + a.y * m.m[4] + a.z * m.m[7],
          a.x * m.m[2] + a.y * m.m[5] + a.z * m.m[8]};
}
// row-vector (1x6) times block lower-triangular 6x6
template <typename T> GD V6<T> rowmul(V6<T> a, const BL6<T> &m) { return {rowmul(a.w, m.A) + rowmul(a.v, m.C), rowmul(a.v, m.D)}; }

// ---- d = 3 groups: interpolated pose + Hint1..4 as 3x3 matrices
template <typename T, bool JAC> struct Interp3Out { M3<T> H1, H2, H3, H4; };

template <typename T, bool JAC>
GD SE2<T> interp_pose2(const T *p1, const T *v1, const T *p2, const T *v2, ICoef<T> k, Interp3Out<T, JAC> &o) {
  const SE2<T> a = {p1[0], p1[1], p1[2]}, b = {p2[0], p2[1], p2[2]};
  const SE2<T> h = se2_between(a, b);
  const V3<T> r = se2_log(h);
  const V3<T> xi = {k.l12 * v1[0] + k.p11 * r.x + k.p12 * v2[0], k.l12 * v1[1] + k.p11 * r.y + k.p12 * v2[1],
                    k.l12 * v1[2] + k.p11 * r.z + k.p12 * v2[2]};
  const SE2<T> ex = se2_exp(xi);
  if (JAC) {
    const M3<T> He = se2_dexp(xi);                         // Hcomp22 * Hexp, Hcomp22 = I
    const M3<T> J3 = se2_dlog(r);
    const M3<T> J1 = neg(J3 * se2_adjoint(se2_inverse(h)));
    o.H1 = se2_adjoint(se2_inverse(ex)) + k.p11 * (He * J1);   // Hcomp21 + Hexpr1 Psi11 Hlog Hcomp11 Hinv
    o.H2 = k.l12 * He;
    o.H3 = k.p11 * (He * J3);
    o.H4 = k.p12 * He;
  }
  return se2_compose(a, ex);
}

template <typename T, bool JAC>
GD M3<T> interp_rot3(const T *p1, const T *v1, const T *p2, const T *v2, ICoef<T> k, Interp3Out<T, JAC> &o) {
  const M3<T> R1 = as_m3(p1), R2 = as_m3(p2);
  const M3<T> h = transpose(R1) * R2;
  const V3<T> r = so3_log(h);
  const V3<T> xi = {k.l12 * v1[0] + k.p11 * r.x + k.p12 * v2[0], k.l12 * v1[1] + k.p11 * r.y + k.p12 * v2[1],
                    k.l12 * v1[2] + k.p11 * r.z + k.p12 * v2[2]};
  const M3<T> ex = so3_exp(xi);
  if (JAC) {
    const M3<T> He = so3_jr(xi);
    const M3<T> J3 = so3_jrinv(r);
    const M3<T> J1 = neg(J3 * transpose(h));
    o.H1 = transpose(ex) + k.p11 * (He * J1);               // Rot3::compose H1 = R2^T
    o.H2 = k.l12 * He;
    o.H3 = k.p11 * (He * J3);
    o.H4 = k.p12 * He;
  }
  return R1 * ex;
}

// ---- SE(3): Hint1..4 are block lower-triangular
template <typename T, bool JAC> struct Interp6Out { BL6<T> H1, H2, H3, H4; };

// gp (round 4): the structured record of the GaussianProcessPriorPose3 on the SAME interval, or null.  What the interpolator
// shares with that prior -- Jinv = Jr^-1(r), Jinv Ad(h^-1) and the finite-difference block d(Jinv v2)/dr -- is three quarters of
// its arithmetic (the difference quotient alone is twelve Jr^-1 evaluations) and is a function of the two states only, not of tau:
// K1 has just written all three (kGps*: X, J = -Jinv Ad(h^-1), F), so the four or so measurement factors of an interval read them
// instead of forming them again.  Same device functions on the same inputs: the same numbers.
template <typename T, bool JAC>
GD SE3<T> interp_pose3(const T *p1, const T *v1, const T *p2, const T *v2, ICoef<T> k, Interp6Out<T, JAC> &o, const T *gp = nullptr) {
  const SE3<T> a = as_se3(p1), b = as_se3(p2);
  const SE3<T> h = se3_between(a, b);
  const V6<T> r = se3_log(h);                                          // :68
  const V6<T> u1 = as_v6(v1), u2 = as_v6(v2);
  BL6<T> Jinv, FD, tmp1;
  if (gp != nullptr) {
    auto m3 = [&](int off) { M3<T> m; for (int q = 0; q < 9; q++) m.m[q] = gp[off + q]; return m; };
    Jinv.A = m3(0); Jinv.C = m3(9); Jinv.D = Jinv.A;                   // kGpsXA, kGpsXC
    if (JAC) {
      tmp1.A = m3(18); tmp1.C = m3(27); tmp1.D = tmp1.A;               // kGpsJA, kGpsJC
      FD.A = m3(36); FD.C = m3(45); FD.D = m3(54);                     // kGpsFA, kGpsFC, kGpsFD
    }
  } else {
    const JrK<T> k0 = jr_coefs(r.w);                                   // trig coefficients shared by Jinv and the FD block
    Jinv = se3_jrinv_k(k0, r);                                         // :72
    if (JAC) {
      FD = se3_jrinv_times_x_fd_k(k0, r, u2);                          // (:84, :93) computed once
      tmp1 = neg(Jinv * se3_adjoint(se3_inverse(h)));                  // Hlogmap Hcomp11 Hinv
    }
  }
  const V6<T> xi = k.l12 * u1 + k.p11 * r + k.p12 * (Jinv * u2);       // Lambda_1 r1 + Psi_1 r2, r1 = [0; v1], r2 = [r; Jinv v2]
  const SE3<T> ex = se3_exp(xi);
  if (JAC) {
    const BL6<T> He = se3_jr(xi);                                      // Hcomp22 * Hexp (:80)
    const BL6<T> s1 = k.p11 * tmp1 + k.p12 * (FD * tmp1);              // Psi_1 * dr2_dT1
    o.H1 = se3_adjoint(se3_inverse(ex)) + He * s1;                     // Hcomp21 + ... (:87)
    o.H2 = k.l12 * He;                                                 // (:89)
    const BL6<T> s3 = k.p11 * Jinv + k.p12 * (FD * Jinv);              // Psi_1 * dr2_dT2, Hlogmap Hcomp12 = Jinv
    o.H3 = He * s3;                                                    // (:96)
    o.H4 = k.p12 * (He * Jinv);                                        // (:98)
  }
  return se3_compose(a, ex);
}

// The same interpolation in the parts the 16-double interpolated rows are made of (kernels.hpp: kIRow*): the pose, He = Hcomp22 Hexp
// (:80), Hc21 = Ad(Exp(xi)^-1) (:87) and s1 = p11 J + p12 F J, so that H1 = Hc21 + He s1 -- H2, H3, H4 are He times what the consumer
// holds already (l12 I, p11 X + p12 F X, p12 X).  gp as above (null: the blocks are formed here).
template <typename T>
GD SE3<T> interp_pose3_parts(const T *p1, const T *v1, const T *p2, const T *v2, ICoef<T> k, const T *gp, BL6<T> &He, BL6<T> &Hc21, BL6<T> &s1) {
  const SE3<T> a = as_se3(p1), b = as_se3(p2);
  const SE3<T> h = se3_between(a, b);
  const V6<T> r = se3_log(h);
  const V6<T> u1 = as_v6(v1), u2 = as_v6(v2);
  BL6<T> Jinv, FD, tmp1;
  if (gp != nullptr) {
    auto m3 = [&](int off) { M3<T> m; for (int q = 0; q < 9; q++) m.m[q] = gp[off + q]; return m; };
    Jinv.A = m3(0); Jinv.C = m3(9); Jinv.D = Jinv.A;
    tmp1.A = m3(18); tmp1.C = m3(27); tmp1.D = tmp1.A;
    FD.A = m3(36); FD.C = m3(45); FD.D = m3(54);
  } else {
    const JrK<T> k0 = jr_coefs(r.w);
    Jinv = se3_jrinv_k(k0, r);
    FD = se3_jrinv_times_x_fd_k(k0, r, u2);
    tmp1 = neg(Jinv * se3_adjoint(se3_inverse(h)));
  }
  const V6<T> xi = k.l12 * u1 + k.p11 * r + k.p12 * (Jinv * u2);
  const SE3<T> ex = se3_exp(xi);
  He = se3_jr(xi);
  s1 = k.p11 * tmp1 + k.p12 * (FD * tmp1);
  Hc21 = se3_adjoint(se3_inverse(ex));
  return se3_compose(a, ex);
}

// ---- Unit3 basis (GTSAM Unit3::basis): B = [b1 b2], b1 = n x axis_min, b2 = n x b1
template <typename T> GD void unit3_basis(V3<T> n, V3<T> &b1, V3<T> &b2) {
  const T mx = fabs(n.x), my = fabs(n.y), mz = fabs(n.z);
  V3<T> axis = {T(0), T(0), T(1)};
  if (mx <= my && mx <= mz) axis = {T(1), T(0), T(0)};
  else if (my <= mx && my <= mz) axis = {T(0), T(1), T(0)};
  b1 = cross(n, axis);
  b1 = (T(1) / sqrt(dot(b1, b1))) * b1;
  b2 = cross(n, b1);
  b2 = (T(1) / sqrt(dot(b2, b2))) * b2;
}

}  // namespace gps
