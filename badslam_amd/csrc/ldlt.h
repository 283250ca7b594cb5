// ldlt.h -- small dense symmetric solve in binary64, host + device.
// What Eigen's  H.cast<double>().selfadjointView<Eigen::Upper>().ldlt().solve(b)  does in the
// reference (B/direct_ba_alternating.cc:206, B/kernel_opt_intrinsics.cc:171,272): LDL^T with
// symmetric diagonal pivoting and the pseudo-inverse rule on D (|D_ii| <= DBL_MIN -> 0).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

namespace bahip {

template <int N>
__host__ __device__ inline void ldlt_solve_sym(double* A /* N*N row-major, destroyed */, const double* b, double* x) {
  int perm[N];
  for (int c = 0; c < N; ++c) perm[c] = c;
  for (int k = 0; k < N; ++k) {
    int piv = k;
    double best = fabs(A[k * N + k]);
    for (int c = k + 1; c < N; ++c)
      if (fabs(A[c * N + c]) > best) { best = fabs(A[c * N + c]); piv = c; }
    if (piv != k) {
      for (int j = 0; j < N; ++j) { const double t = A[k * N + j]; A[k * N + j] = A[piv * N + j]; A[piv * N + j] = t; }
      for (int j = 0; j < N; ++j) { const double t = A[j * N + k]; A[j * N + k] = A[j * N + piv]; A[j * N + piv] = t; }
      const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    const double d = A[k * N + k];
    if (fabs(d) > 2.2250738585072014e-308) {
      for (int c = k + 1; c < N; ++c) A[c * N + k] /= d;
      for (int c = k + 1; c < N; ++c)
        for (int j = k + 1; j <= c; ++j) {
          A[c * N + j] -= A[c * N + k] * d * A[j * N + k];
          A[j * N + c] = A[c * N + j];
        }
    } else {
      for (int c = k + 1; c < N; ++c) A[c * N + k] = 0;
    }
  }
  double y[N];
  for (int c = 0; c < N; ++c) y[c] = b[perm[c]];
  for (int c = 0; c < N; ++c)
    for (int j = 0; j < c; ++j) y[c] -= A[c * N + j] * y[j];
  for (int c = 0; c < N; ++c) {
    const double d = A[c * N + c];
    y[c] = (fabs(d) > 2.2250738585072014e-308) ? y[c] / d : 0.0;
  }
  for (int c = N - 1; c >= 0; --c)
    for (int j = c + 1; j < N; ++j) y[c] -= A[j * N + c] * y[j];
  for (int c = 0; c < N; ++c) x[perm[c]] = y[c];
}

}  // namespace bahip
