/*
 * orc_math.h -- tiny dense row-major matrix helpers for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is the checker for the HIP path; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * The reference leans on Eigen for this; there is no Eigen here, so these are
 * plain loops.  All matrices are row-major double arrays.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <string.h>

/* C(m x n) = A(m x k) * B(k x n) */
static inline void orc_mm(int m, int k, int n, const double *A, const double *B, double *C) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0.0;
      for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
      C[i * n + j] = s;
    }
}

/* C(m x n) = A^T * B, A is (k x m), B is (k x n) */
static inline void orc_mtm(int k, int m, int n, const double *A, const double *B, double *C) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0.0;
      for (int l = 0; l < k; l++) s += A[l * m + i] * B[l * n + j];
      C[i * n + j] = s;
    }
}

/* B(n x m) = A(m x n)^T */
static inline void orc_tr(int m, int n, const double *A, double *B) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) B[j * m + i] = A[i * n + j];
}

static inline void orc_eye(int n, double *A) {
  memset(A, 0, sizeof(double) * (size_t)n * (size_t)n);
  for (int i = 0; i < n; i++) A[i * n + i] = 1.0;
}

static inline void orc_zero(int n, double *A) { memset(A, 0, sizeof(double) * (size_t)n); }
static inline void orc_copy(int n, const double *A, double *B) { memcpy(B, A, sizeof(double) * (size_t)n); }
static inline void orc_scale(int n, double s, double *A) { for (int i = 0; i < n; i++) A[i] *= s; }
static inline void orc_axpy(int n, double a, const double *X, double *Y) { for (int i = 0; i < n; i++) Y[i] += a * X[i]; }

static inline double orc_dot(int n, const double *a, const double *b) {
  double s = 0.0;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

/* copy an (r x c) block of src (leading dim lds) at (sr, sc) into dst (leading dim ldd) at (dr, dc) */
static inline void orc_blk(int r, int c, const double *src, int lds, int sr, int sc,
                           double *dst, int ldd, int dr, int dc) {
  for (int i = 0; i < r; i++)
    for (int j = 0; j < c; j++) dst[(dr + i) * ldd + dc + j] = src[(sr + i) * lds + sc + j];
}

static inline void orc_cross(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* In-place Cholesky A = U^T U (upper, row-major, n x n).  Returns 0 on success, -1 if not SPD. */
static inline int orc_chol_upper(int n, double *A) {
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[k * n + j] * A[k * n + j];
    if (!(d > 0.0)) return -1;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[j * n + i];
      for (int k = 0; k < j; k++) s -= A[k * n + j] * A[k * n + i];
      A[j * n + i] = s / d;
    }
    for (int i = 0; i < j; i++) A[j * n + i] = 0.0;
  }
  return 0;
}

/* General inverse by Gauss-Jordan with partial pivoting (n <= 16). Returns 0 ok, -1 singular. */
static inline int orc_inv(int n, const double *A, double *Ainv) {
  double M[16 * 32];
  if (n > 16) return -1;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) {
      M[i * 2 * n + j] = A[i * n + j];
      M[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
    }
  }
  for (int c = 0; c < n; c++) {
    int p = c;
    double best = fabs(M[c * 2 * n + c]);
    for (int r = c + 1; r < n; r++)
      if (fabs(M[r * 2 * n + c]) > best) { best = fabs(M[r * 2 * n + c]); p = r; }
    if (best == 0.0) return -1;
    if (p != c)
      for (int j = 0; j < 2 * n; j++) {
        double t = M[c * 2 * n + j];
        M[c * 2 * n + j] = M[p * 2 * n + j];
        M[p * 2 * n + j] = t;
      }
    double piv = M[c * 2 * n + c];
    for (int j = 0; j < 2 * n; j++) M[c * 2 * n + j] /= piv;
    for (int r = 0; r < n; r++) {
      if (r == c) continue;
      double f = M[r * 2 * n + c];
      if (f == 0.0) continue;
      for (int j = 0; j < 2 * n; j++) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
    }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ainv[i * n + j] = M[i * 2 * n + n + j];
  return 0;
}

#endif /* ORC_MATH_H */
