// geofilter_oracle.cpp - TEST INFRASTRUCTURE ONLY: a plain C++ restatement (no Eigen, no openMVG headers) of the reference's
// a-contrario fundamental-matrix filter, SURVEY.md 8(f) N2. Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of the
// bench may call it; the product (openmvg_amd/) never does.
//
// What is restated (paths under /root/reference/src/openMVG):
//   matching_image_collection/F_ACRobust.hpp:65-122            GeometricFilter_FMatrix_AC::Robust_estimation (kernel set-up, the
//                                                               "more than 2.5 x 7 inliers" acceptance, m_dPrecision_robust)
//   robust_estimation/robust_estimator_ACRansacKernelAdaptator.hpp:104-202   ACKernelAdaptor (normalisation by image size,
//                                                               logalpha0 of the point-to-line model, multError 0.5, unormalizeError)
//   multiview/conditioning.cpp:44-67,87-89                      PreconditionerFromPoints(w, h), UnnormalizerT
//   multiview/solver_fundamental_kernel.cpp:37-93,157-166       SevenPointSolver (null space of the 7 x 9 epipolar system, cubic in the
//                                                               pencil), EpipolarDistanceError
//   numeric/poly.h:32-96                                        SolveCubicPolynomial (Cardano / Viete)
//   robust_estimation/robust_estimator_ACRansac.hpp:58-119,196-262,339-489   log-combinatorial tables (float), the quantified NFA
//                                                               on a 20-bin histogram of the residuals, the ACRANSAC loop (max-consensus
//                                                               warm-up, early exit, focused sampling among the best inliers)
//   robust_estimation/rand_sampling.hpp:43-110                  the two UniformSample forms (rejection / Fisher-Yates)
//   third_party/histogram/histogram.hpp:47-112                  bin of a value, bin centres
//   std::mt19937 (default seed 5489) and libstdc++ 11's std::uniform_int_distribution<uint32_t> on a 32-bit engine
//   (bits/uniform_int_dist.h: Lemire's multiply-shift with rejection) - the sample sequence is part of the result.
//
// The homography model (port_geofilter_h_acransac) shares everything above but the kernel adaptor:
//   matching_image_collection/H_ACRobust.hpp:49-113             GeometricFilter_HMatrix_AC::Robust_estimation (point-to-point, 2.5 x 4)
//   multiview/solver_homography_kernel.cpp:37-93, .hpp:60-64     FourPointSolver (null vector of the 8 x 9 DLT system; the reference: last
//                                                               right singular vector, numeric/nullspace.cpp:14-32), AsymmetricError
//   robust_estimator_ACRansacKernelAdaptator.hpp:38-81          logalpha0 / multError of the point-to-point parametrisation
//   multiview/conditioning.cpp:80-82                            UnnormalizerI
// Pinned to the compiled reference and its stored outputs by tests/test_geofilter_h.py.
//
// The angular essential models (port_geofilter_e_angular_acransac) and the orthographic one (port_geofilter_eo_acransac), same loop:
//   matching_image_collection/E_ACRobust_Angular.hpp:53-160     the a-contrario stage of GeometricFilter_ESphericalMatrix_AC_Angular<isUpright>
//                                                               (bound D2R(precision), 2.5 x MINIMUM_SAMPLES); its RelativePoseFromEssential
//                                                               stage (:126-143) is not restated: the compiled reference checks it
//   robust_estimator_ACRansacKernelAdaptator.hpp:85-100,465-541 RADIAN_ANGLE: log alpha0 = log10(1 / 2), multError 1 / 4, no normalisation
//   multiview/solver_essential_eight_point.cpp:17-61            EightPointRelativePoseSolver (null vector of the 8 x 9 system), AngularError
//   multiview/solver_essential_three_point.cpp:31-113           ThreePointUprightRelativePoseSolver; ThreePointsRelativePose (closed form)
//   matching_image_collection/Eo_Robust.hpp:50-144, robust_estimator_ACRansacKernelAdaptator.hpp:384-456, solver_essential_kernel.hpp:69-78
//                                                               ACKernelAdaptorEssentialOrtho, OrthographicSymmetricEpipolarDistanceError
// Pinned to the compiled reference's stored outputs by tests/test_geofilter_angular.py / test_geofilter_ortho.py (the orthographic model
// bit for bit: closed form, this file is compiled with -ffp-contract=off).
//
// One deliberate difference: the reference takes the two-dimensional null space of the 7 x 9 system from
// Eigen::SelfAdjointEigenSolver on A^T A (its two smallest eigenvectors); here it comes from Householder reflections of A^T (the
// last two columns of Q), the device code uses complete-pivoting elimination. The three bases span the same plane for a sample in
// general position, so the fundamental matrices of the pencil are the same up to scale and rounding - not bit for bit (they were
// not reproducible across Eigen builds either). Parity policy: identical inlier sets and F equal to 1e-6 after
// normalisation on fixtures whose decisive residuals are not within rounding of a histogram edge.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Mt19937 {   // std::mt19937
  uint32_t mt[624];
  int idx;
  explicit Mt19937(uint32_t seed = 5489u) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int i = 0; i < 624; ++i) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
};
// std::uniform_int_distribution<uint32_t>(a, b)(g) of libstdc++ 11 for a generator with a full 32-bit range
uint32_t uniform_u32(Mt19937& g, uint32_t a, uint32_t b) {
  const uint32_t urange = b - a;
  if (urange == 0xffffffffu) return g.next() + a;
  const uint32_t range = urange + 1;
  uint64_t product = (uint64_t)g.next() * range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) { product = (uint64_t)g.next() * range; low = (uint32_t)product; }
  }
  return (uint32_t)(product >> 32) + a;
}

float logcombi(uint32_t k, uint32_t n, const std::vector<float>& l10) {
  if (k >= n) return 0.f;
  if (n - k < k) k = n - k;
  float r = 0.f;
  for (uint32_t i = 1; i <= k; ++i) r += l10[n - i + 1] - l10[i];
  return r;
}

struct Model { double f[9]; };   // row-major 3 x 3

// null space of an R x 9 system (R = 7: seven-point, two vectors; R = 8: four-point DLT, one vector): A^T = Q R by Householder
// reflections, the last 9 - R columns of Q
template <int R>
void nullspace_rows(const double A[R][9], double (*out)[9]) {
  double M[9][R];   // A^T
  for (int r = 0; r < R; ++r) for (int c = 0; c < 9; ++c) M[c][r] = A[r][c];
  double V[R][9];   // Householder vectors
  for (int k = 0; k < R; ++k) {
    double norm = 0;
    for (int i = k; i < 9; ++i) norm += M[i][k] * M[i][k];
    norm = std::sqrt(norm);
    double v[9] = {0};
    for (int i = k; i < 9; ++i) v[i] = M[i][k];
    v[k] += (M[k][k] >= 0 ? norm : -norm);
    double vv = 0;
    for (int i = k; i < 9; ++i) vv += v[i] * v[i];
    for (int i = 0; i < 9; ++i) V[k][i] = v[i];
    if (vv == 0) continue;
    for (int j = k; j < R; ++j) {
      double dot = 0;
      for (int i = k; i < 9; ++i) dot += v[i] * M[i][j];
      const double s = 2 * dot / vv;
      for (int i = k; i < 9; ++i) M[i][j] -= s * v[i];
    }
  }
  for (int which = 0; which < 9 - R; ++which) {   // Q e_R .. Q e_8 with Q = H_0 H_1 ... H_(R-1)
    double q[9] = {0};
    q[R + which] = 1;
    for (int k = R - 1; k >= 0; --k) {
      double vv = 0, dot = 0;
      for (int i = 0; i < 9; ++i) { vv += V[k][i] * V[k][i]; dot += V[k][i] * q[i]; }
      if (vv == 0) continue;
      const double s = 2 * dot / vv;
      for (int i = 0; i < 9; ++i) q[i] -= s * V[k][i];
    }
    std::memcpy(out[which], q, sizeof(q));
  }
}
void nullspace7(const double A[7][9], double f1[9], double f2[9]) {
  double out[2][9];
  nullspace_rows<7>(A, out);
  std::memcpy(f1, out[0], sizeof(out[0])); std::memcpy(f2, out[1], sizeof(out[1]));
}

int solve_cubic(double a, double b, double c, double x[3]) {   // numeric/poly.h:32-75
  const double eps = std::numeric_limits<double>::epsilon();
  a /= 3;
  double p = (b - 3 * a * a) / 3;
  double q = (2 * a * a * a - a * b + c) / 2;
  double d = q * q + p * p * p;
  const double tolq = std::max(std::abs(2 * a * a * a), std::max(std::abs(a * b), std::abs(c)));
  const double tolp = std::max(std::abs(b), std::abs(3 * a * a));
  int n = (d > eps * std::max(p * p * tolp, std::abs(q) * tolq) ? 1 : 3);
  if (n == 1) {
    d = std::pow(std::abs(q) + std::sqrt(d), 1 / 3.0);
    x[0] = d - p / d;
    if (q > 0) x[0] = -x[0];
  } else {
    if (3 * p >= -eps * tolp) { n = 1; x[0] = 0; }
    else {
      p = std::sqrt(-p);
      q /= p * p * p;
      d = (q <= -1) ? M_PI : (q >= 1) ? 0 : std::acos(q);
      for (int i = 0; i < 3; ++i) x[i] = -2 * p * std::cos((d + 2 * M_PI * i) / 3);
    }
  }
  for (int i = 0; i < n; ++i) x[i] -= a;
  return n;
}

int seven_point(const double* x1, const double* x2, const uint32_t s[7], Model out[3]) {   // solver_fundamental_kernel.cpp:37-93
  double A[7][9];
  for (int r = 0; r < 7; ++r) {
    const double a1[3] = {x1[2 * s[r]], x1[2 * s[r] + 1], 1.0}, a2[3] = {x2[2 * s[r]], x2[2 * s[r] + 1], 1.0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[r][3 * i + j] = a2[i] * a1[j];
  }
  double F1[9], F2[9];
  nullspace7(A, F1, F2);
  const double a = F1[0], j = F2[0], b = F1[1], k = F2[1], c = F1[2], l = F2[2], d = F1[3], m = F2[3], e = F1[4], n = F2[4],
               f = F1[5], o = F2[5], g = F1[6], p = F2[6], h = F1[7], q = F2[7], i = F1[8], r = F2[8];
  const double P[4] = {
    a*e*i + b*f*g + c*d*h - a*f*h - b*d*i - c*e*g,
    a*e*r + a*i*n + b*f*p + b*g*o + c*d*q + c*h*m + d*h*l + e*i*j + f*g*k -
    a*f*q - a*h*o - b*d*r - b*i*m - c*e*p - c*g*n - d*i*k - e*g*l - f*h*j,
    a*n*r + b*o*p + c*m*q + d*l*q + e*j*r + f*k*p + g*k*o + h*l*m + i*j*n -
    a*o*q - b*m*r - c*n*p - d*k*r - e*l*p - f*j*q - g*l*n - h*j*o - i*k*m,
    j*n*r + k*o*p + l*m*q - j*o*q - k*m*r - l*n*p};
  if (P[0] == 0.0) return 0;   // poly.h:88-91
  double roots[3];
  const int nr = solve_cubic(P[2] / P[3], P[1] / P[3], P[0] / P[3], roots);
  for (int t = 0; t < nr; ++t)
    for (int u = 0; u < 9; ++u) out[t].f[u] = F1[u] + roots[t] * F2[u];
  return nr;
}

inline double epipolar_error(const Model& F, const double* x, const double* y) {   // EpipolarDistanceError, :157-166
  const double fx0 = F.f[0] * x[0] + F.f[1] * x[1] + F.f[2], fx1 = F.f[3] * x[0] + F.f[4] * x[1] + F.f[5],
               fx2 = F.f[6] * x[0] + F.f[7] * x[1] + F.f[8];
  const double dt = fx0 * y[0] + fx1 * y[1] + fx2;
  return dt * dt / (fx0 * fx0 + fx1 * fx1);
}

// FourPointSolver::Solve (multiview/solver_homography_kernel.cpp:37-93): the null vector of the 8 x 9 DLT system, row-major H
int four_point(const double* x1, const double* x2, const uint32_t s[4], Model out[1]) {
  double A[8][9];
  for (int i = 0; i < 4; ++i) {
    const double x = x1[2 * s[i]], y = x1[2 * s[i] + 1], u = x2[2 * s[i]], v = x2[2 * s[i] + 1];
    const double r0[9] = {x, y, 1.0, 0.0, 0.0, 0.0, -u * x, -u * y, -u}, r1[9] = {0.0, 0.0, 0.0, x, y, 1.0, -v * x, -v * y, -v};
    std::memcpy(A[2 * i], r0, sizeof(r0)); std::memcpy(A[2 * i + 1], r1, sizeof(r1));
  }
  double h[1][9];
  nullspace_rows<8>(A, h);
  std::memcpy(out[0].f, h[0], sizeof(h[0]));
  return 1;
}
inline double homography_error(const Model& H, const double* x, const double* y) {   // AsymmetricError, solver_homography_kernel.hpp:60-64
  const double v0 = (H.f[0] * x[0] + H.f[1] * x[1]) + H.f[2], v1 = (H.f[3] * x[0] + H.f[4] * x[1]) + H.f[5], v2 = (H.f[6] * x[0] + H.f[7] * x[1]) + H.f[8];
  const double dx = y[0] - v0 / v2, dy = y[1] - v1 / v2;
  return dx * dx + dy * dy;
}


// ---- FivePointSolver (multiview/solver_essential_five_point.cpp:35-230), restated with other linear algebra than both the reference
// (Eigen: SelfAdjointEigenSolver null space, FullPivLU, EigenSolver) and the device kernel (complete-pivoting elimination + Gram-Schmidt,
// lane-parallel QR iteration, closed-form eigenvector rows): Householder null space, Gauss-Jordan with partial pivoting, scalar
// Hessenberg reduction by stabilised elimination + the hqr iteration, eigenvectors by elimination of (A - lambda I). ----
namespace fivept {
typedef double P1[4];    // {x, y, z, 1}
typedef double P2[10];   // {xx, xy, yy, xz, yz, zz, x, y, z, 1}
void o1(const double* a, const double* b, double* r) {
  r[0] = a[0] * b[0]; r[1] = a[0] * b[1] + a[1] * b[0]; r[2] = a[1] * b[1]; r[3] = a[0] * b[2] + a[2] * b[0]; r[4] = a[1] * b[2] + a[2] * b[1];
  r[5] = a[2] * b[2]; r[6] = a[0] * b[3] + a[3] * b[0]; r[7] = a[1] * b[3] + a[3] * b[1]; r[8] = a[2] * b[3] + a[3] * b[2]; r[9] = a[3] * b[3];
}
void o2_add(const double* a, const double* b, double* r) {   // degree 2 x degree 1, reference column order
  const double axx = a[0], axy = a[1], ayy = a[2], axz = a[3], ayz = a[4], azz = a[5], ax = a[6], ay = a[7], az = a[8], a1 = a[9];
  const double bx = b[0], by = b[1], bz = b[2], b1 = b[3];
  r[0] += axx * bx; r[1] += axx * by + axy * bx; r[2] += axy * by + ayy * bx; r[3] += ayy * by; r[4] += axx * bz + axz * bx;
  r[5] += axy * bz + ayz * bx + axz * by; r[6] += ayy * bz + ayz * by; r[7] += axz * bz + azz * bx; r[8] += ayz * bz + azz * by; r[9] += azz * bz;
  r[10] += axx * b1 + ax * bx; r[11] += axy * b1 + ax * by + ay * bx; r[12] += ayy * b1 + ay * by; r[13] += axz * b1 + ax * bz + az * bx;
  r[14] += ayz * b1 + ay * bz + az * by; r[15] += azz * b1 + az * bz; r[16] += ax * b1 + a1 * bx; r[17] += ay * b1 + a1 * by; r[18] += az * b1 + a1 * bz;
  r[19] += a1 * b1;
}
// eigenvalues of a general real n x n matrix (n = 10): elimination to Hessenberg form (elmhes) + hqr (EISPACK / Numerical Recipes)
bool eigenvalues(double a[10][10], double wr[10], double wi[10]) {
  const int n = 10;
  for (int m = 1; m < n - 1; ++m) {
    double x = 0.0; int i = m;
    for (int j = m; j < n; ++j) if (std::fabs(a[j][m - 1]) > std::fabs(x)) { x = a[j][m - 1]; i = j; }
    if (i != m) { for (int j = m - 1; j < n; ++j) std::swap(a[i][j], a[m][j]); for (int j = 0; j < n; ++j) std::swap(a[j][i], a[j][m]); }
    if (x != 0.0)
      for (i = m + 1; i < n; ++i) {
        double y = a[i][m - 1];
        if (y != 0.0) { y /= x; a[i][m - 1] = y; for (int j = m; j < n; ++j) a[i][j] -= y * a[m][j]; for (int j = 0; j < n; ++j) a[j][m] += y * a[j][i]; }
      }
  }
  for (int i = 2; i < n; ++i) for (int j = 0; j < i - 1; ++j) a[i][j] = 0.0;
  double anorm = 0.0;
  for (int i = 0; i < n; ++i) for (int j = std::max(i - 1, 0); j < n; ++j) anorm += std::fabs(a[i][j]);
  int nn = n - 1; double t = 0.0, p = 0, q = 0, r = 0, z = 0, w, x, y, s;
  while (nn >= 0) {
    int its = 0, l;
    do {
      for (l = nn; l >= 1; --l) { s = std::fabs(a[l - 1][l - 1]) + std::fabs(a[l][l]); if (s == 0.0) s = anorm; if (std::fabs(a[l][l - 1]) + s == s) { a[l][l - 1] = 0.0; break; } }
      x = a[nn][nn];
      if (l == nn) { wr[nn] = x + t; wi[nn--] = 0.0; }
      else {
        y = a[nn - 1][nn - 1]; w = a[nn][nn - 1] * a[nn - 1][nn];
        if (l == nn - 1) {
          p = 0.5 * (y - x); q = p * p + w; z = std::sqrt(std::fabs(q)); x += t;
          if (q >= 0.0) { z = p + (p >= 0.0 ? std::fabs(z) : -std::fabs(z)); wr[nn - 1] = wr[nn] = x + z; if (z != 0.0) wr[nn] = x - w / z; wi[nn - 1] = wi[nn] = 0.0; }
          else { wr[nn - 1] = wr[nn] = x + p; wi[nn - 1] = z; wi[nn] = -z; }
          nn -= 2;
        } else {
          if (its == 30) return false;
          if (its == 10 || its == 20) { t += x; for (int i = 0; i <= nn; ++i) a[i][i] -= x; s = std::fabs(a[nn][nn - 1]) + std::fabs(a[nn - 1][nn - 2]); y = x = 0.75 * s; w = -0.4375 * s * s; }
          ++its;
          int m;
          for (m = nn - 2; m >= l; --m) {
            z = a[m][m]; r = x - z; s = y - z;
            p = (r * s - w) / a[m + 1][m] + a[m][m + 1]; q = a[m + 1][m + 1] - z - r - s; r = a[m + 2][m + 1];
            s = std::fabs(p) + std::fabs(q) + std::fabs(r); p /= s; q /= s; r /= s;
            if (m == l) break;
            const double u = std::fabs(a[m][m - 1]) * (std::fabs(q) + std::fabs(r)), v = std::fabs(p) * (std::fabs(a[m - 1][m - 1]) + std::fabs(z) + std::fabs(a[m + 1][m + 1]));
            if (u + v == v) break;
          }
          for (int i = m + 2; i <= nn; ++i) { a[i][i - 2] = 0.0; if (i != m + 2) a[i][i - 3] = 0.0; }
          for (int k = m; k <= nn - 1; ++k) {
            if (k != m) { p = a[k][k - 1]; q = a[k + 1][k - 1]; r = 0.0; if (k != nn - 1) r = a[k + 2][k - 1]; if ((x = std::fabs(p) + std::fabs(q) + std::fabs(r)) != 0.0) { p /= x; q /= x; r /= x; } }
            const double sq = std::sqrt(p * p + q * q + r * r);
            if ((s = p >= 0.0 ? sq : -sq) != 0.0) {
              if (k == m) { if (l != m) a[k][k - 1] = -a[k][k - 1]; } else a[k][k - 1] = -s * x;
              p += s; x = p / s; y = q / s; z = r / s; q /= p; r /= p;
              for (int j = k; j <= nn; ++j) { p = a[k][j] + q * a[k + 1][j]; if (k != nn - 1) { p += r * a[k + 2][j]; a[k + 2][j] -= p * z; } a[k + 1][j] -= p * y; a[k][j] -= p * x; }
              const int mmin = nn < k + 3 ? nn : k + 3;
              for (int i = l; i <= mmin; ++i) { p = x * a[i][k] + y * a[i][k + 1]; if (k != nn - 1) { p += z * a[i][k + 2]; a[i][k + 2] -= p * r; } a[i][k + 1] -= p * q; a[i][k] -= p; }
            }
          }
        }
      }
    } while (l < nn - 1);
  }
  return true;
}
// b1, b2: bearing vectors (3 doubles per correspondence), s: five sample indices. Essential matrices (row-major) to out; returns their number.
int five_point(const double* b1, const double* b2, const uint32_t* s, Model* out) {
  double A[5][9];
  for (int r = 0; r < 5; ++r) for (int c = 0; c < 9; ++c) A[r][c] = b2[3 * (size_t)s[r] + c / 3] * b1[3 * (size_t)s[r] + c % 3];
  double nb[4][9];
  nullspace_rows<5>(A, nb);   // orthonormal
  double E[9][4];             // E[3 i + j] = polynomial of entry (i, j)
  for (int u = 0; u < 9; ++u) for (int k = 0; k < 4; ++k) E[u][k] = nb[k][u];
  double M[10][20] = {{0}};
  { double p[10], q[10], d[10];
    const int t3[3][5] = {{1, 5, 2, 4, 6}, {2, 3, 0, 5, 7}, {0, 4, 1, 3, 8}};
    for (auto& t : t3) { o1(E[t[0]], E[t[1]], p); o1(E[t[2]], E[t[3]], q); for (int k = 0; k < 10; ++k) d[k] = p[k] - q[k]; o2_add(d, E[t[4]], M[0]); } }
  double EET[3][3][10], tr[10];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    for (int k = 0; k < 10; ++k) EET[i][j][k] = 0.0;
    for (int m = 0; m < 3; ++m) { double p[10]; o1(E[3 * i + m], E[3 * j + m], p); for (int k = 0; k < 10; ++k) EET[i][j][k] += p[k]; }
  }
  for (int k = 0; k < 10; ++k) tr[k] = 0.5 * (EET[0][0][k] + EET[1][1][k] + EET[2][2][k]);
  for (int i = 0; i < 3; ++i) for (int k = 0; k < 10; ++k) EET[i][i][k] -= tr[k];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int m = 0; m < 3; ++m) o2_add(EET[i][m], E[3 * m + j], M[1 + 3 * i + j]);
  // Gauss-Jordan with partial pivoting on the cubic columns: M -> [I | B]
  for (int c = 0; c < 10; ++c) {
    int piv = c;
    for (int r = c + 1; r < 10; ++r) if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
    if (M[piv][c] == 0.0) return 0;
    if (piv != c) for (int k = 0; k < 20; ++k) std::swap(M[piv][k], M[c][k]);
    const double ip = 1.0 / M[c][c];
    for (int k = 0; k < 20; ++k) M[c][k] *= ip;
    for (int r = 0; r < 10; ++r) if (r != c) { const double f = M[r][c]; if (f != 0.0) for (int k = 0; k < 20; ++k) M[r][k] -= f * M[c][k]; }
  }
  double At[10][10] = {{0}}, H[10][10];
  const int rows[6] = {0, 1, 2, 4, 5, 7};
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 10; ++c) At[r][c] = M[rows[r]][10 + c];
  At[6][0] = At[7][1] = At[8][3] = At[9][6] = -1.0;
  std::memcpy(H, At, sizeof(H));
  double wr[10], wi[10];
  if (!eigenvalues(H, wr, wi)) return 0;
  int n = 0;
  for (int e = 0; e < 10; ++e) {
    if (wi[e] != 0.0 || !(std::fabs(wr[e]) < 1e150)) continue;
    // null vector of (At - lambda I): elimination with complete pivoting, the column left without a pivot is free
    double G[10][10];
    for (int r = 0; r < 10; ++r) for (int c = 0; c < 10; ++c) G[r][c] = At[r][c] - (r == c ? wr[e] : 0.0);
    int prow[10], pcol[10]; bool ru[10] = {false}, cu[10] = {false};
    for (int st = 0; st < 9; ++st) {
      double best = 0.0; int br = -1, bc = -1;
      for (int r = 0; r < 10; ++r) if (!ru[r]) for (int c = 0; c < 10; ++c) if (!cu[c] && std::fabs(G[r][c]) > best) { best = std::fabs(G[r][c]); br = r; bc = c; }
      if (br < 0) { prow[st] = -1; continue; }
      ru[br] = cu[bc] = true; prow[st] = br; pcol[st] = bc;
      for (int r = 0; r < 10; ++r) if (r != br) { const double f = G[r][bc] / G[br][bc]; if (f != 0.0) for (int c = 0; c < 10; ++c) G[r][c] -= f * G[br][c]; }
    }
    int fc = 0; while (fc < 10 && cu[fc]) ++fc;
    double v[10] = {0}; v[fc] = 1.0;
    for (int st = 0; st < 9; ++st) if (prow[st] >= 0) v[pcol[st]] = -G[prow[st]][fc] / G[prow[st]][pcol[st]];
    for (int u = 0; u < 9; ++u) out[n].f[u] = E[u][0] * v[6] + E[u][1] * v[7] + E[u][2] * v[8] + E[u][3] * v[9];
    ++n;
  }
  return n;
}
}  // namespace fivept

// ---- the angular and the orthographic essential models (E_ACRobust_Angular.hpp:33-191, Eo_Robust.hpp:35-165) ----
// EightPointRelativePoseSolver::Solve on exactly eight bearing pairs (multiview/solver_essential_eight_point.cpp:17-47): the null vector of
// the 8 x 9 system A[r][3 i + j] = x2[i] x1[j]; with eight columns the projection onto the essential manifold is skipped (:36)
int eight_point_bearings(const double* b1, const double* b2, const uint32_t s[8], Model out[1]) {
  double A[8][9];
  for (int r = 0; r < 8; ++r)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A[r][3 * i + j] = b2[3 * (size_t)s[r] + i] * b1[3 * (size_t)s[r] + j];
  double o[1][9];
  nullspace_rows<8>(A, o);
  std::memcpy(out[0].f, o[0], sizeof(o[0]));
  return 1;
}
// ThreePointUprightRelativePoseSolver::Solve (multiview/solver_essential_three_point.cpp:84-113): null vector n of the 3 x 4 system with rows
// [a.x b.y, -a.z b.y, -b.x a.y, -b.z a.y] (here by its 3 x 3 minors), E = [0 n2 0; -n0 0 n1; 0 n3 0]
int three_point_upright(const double* b1, const double* b2, const uint32_t s[3], Model out[1]) {
  double A[3][4];
  for (int i = 0; i < 3; ++i) {
    const double* a = b1 + 3 * (size_t)s[i]; const double* b = b2 + 3 * (size_t)s[i];
    A[i][0] = a[0] * b[1]; A[i][1] = -a[2] * b[1]; A[i][2] = -b[0] * a[1]; A[i][3] = -b[2] * a[1];
  }
  auto det3 = [&](int c0, int c1, int c2) {
    return A[0][c0] * (A[1][c1] * A[2][c2] - A[1][c2] * A[2][c1]) - A[0][c1] * (A[1][c0] * A[2][c2] - A[1][c2] * A[2][c0]) +
           A[0][c2] * (A[1][c0] * A[2][c1] - A[1][c1] * A[2][c0]);
  };
  const double n0 = det3(1, 2, 3), n1 = -det3(0, 2, 3), n2 = det3(0, 1, 3), n3 = -det3(0, 1, 2);
  for (int u = 0; u < 9; ++u) out[0].f[u] = 0.0;
  out[0].f[1] = n2; out[0].f[3] = -n0; out[0].f[5] = n1; out[0].f[7] = n3;
  return 1;
}
// Square(AngularError::Error) (multiview/solver_essential_eight_point.cpp:50-61, ACKernelAdaptor_AngularRadianError::Errors)
inline double angular_error_sq(const Model& E, const double* a, const double* b) {
  double e[3];
  for (int r = 0; r < 3; ++r) e[r] = (E.f[3 * r] * a[0] + E.f[3 * r + 1] * a[1]) + E.f[3 * r + 2] * a[2];
  const double n2 = (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2];
  if (n2 > 0.0) { const double nn = std::sqrt(n2); e[0] /= nn; e[1] /= nn; e[2] /= nn; }
  const double ang = std::asin((b[0] * e[0] + b[1] * e[1]) + b[2] * e[2]);
  return ang * ang;
}
// ThreePointsRelativePose (multiview/solver_essential_three_point.cpp:31-79): the two closed-form orthographic essential matrices
int three_point_ortho(const double* x1, const double* x2, const uint32_t s[3], Model out[2]) {
  const double *p0 = x1 + 2 * (size_t)s[0], *p1 = x1 + 2 * (size_t)s[1], *p2 = x1 + 2 * (size_t)s[2];
  const double *q0 = x2 + 2 * (size_t)s[0], *q1 = x2 + 2 * (size_t)s[1], *q2 = x2 + 2 * (size_t)s[2];
  const double u1x = p1[0] - p0[0], u1y = p1[1] - p0[1], v1x = p2[0] - p0[0], v1y = p2[1] - p0[1];
  const double u2x = q1[0] - q0[0], u2y = q1[1] - q0[1], v2x = q2[0] - q0[0], v2y = q2[1] - q0[1];
  const double denom = u1x * v1y - u1y * v1x;
  const double ac = (u1y * v2x - u2x * v1y) / denom, ad = (u1y * v2y - u2y * v1y) / denom;
  const double bc = (u2x * v1x - u1x * v2x) / denom, bd = (u2y * v1x - u1x * v2y) / denom;
  const double ac2 = ac * ac;
  const double g2 = -ac2 + ad * ad - bc * bc + bd * bd;
  const double g1 = 2.0 * ac * ad + 2.0 * bc * bd;
  const double g0 = ac2 + bc * bc - 1.0;
  const double h4 = g1 * g1 + g2 * g2;
  const double h2 = -g1 * g1 + 2.0 * g0 * g2;
  const double h0 = g0 * g0;
  const double rdisc = std::sqrt(h2 * h2 - 4.0 * h4 * h0);
  for (int k = 0; k < 2; ++k) {
    const double root = k == 0 ? h2 + rdisc : h2 - rdisc;
    const double sd = std::sqrt(-root / h4 / 2.0);
    const double sc = -(g2 * sd * sd + ac2 + bc * bc - 1.0) / (2.0 * ac * ad * sd + 2.0 * bc * bd * sd);
    const double sa = ac * sc + ad * sd, sb = bc * sc + bd * sd;
    const double se = -sa * p0[0] - sb * p0[1] - sc * q0[0] - sd * q0[1];
    const double e[9] = {0, 0, sa, 0, 0, sb, sc, sd, se};
    std::memcpy(out[k].f, e, sizeof(e));
  }
  return 2;
}
// OrthographicSymmetricEpipolarDistanceError (multiview/solver_essential_kernel.hpp:69-78)
inline double ortho_error(const Model& E, const double* x, const double* y) {
  return std::abs(E.f[8] + x[0] * E.f[2] + x[1] * E.f[5] + y[0] * E.f[6] + y[1] * E.f[7]);
}

struct Pair {
  uint32_t n;
  std::vector<double> x1, x2;   // normalised
  double n2_00, t2[3];          // N2(0,0) and the entries of N1 / N2 needed to unnormalise: T = [[s,0,tx],[0,s,ty],[0,0,1]]
  double n1_00, t1[3];
  double logalpha0, max_threshold;
};

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// homography = false: ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError, UnnormalizerT> (point to line);
// homography = true: ACKernelAdaptor<FourPointSolver, AsymmetricError, UnnormalizerI> configured point to point (H_ACRobust.hpp:77-87)
// essential (K != NULL): ACKernelAdaptorEssential<FivePointSolver, EpipolarDistanceError> on the pixels, bearings = normalised Kinv (x, y, 1)
// extra: kAngular8 / kUpright3 = ACKernelAdaptor_AngularRadianError<EightPointRelativePoseSolver | ThreePointUprightRelativePoseSolver, AngularError> on
// the bearing vectors bI / bJ alone (precision in degrees, E_ACRobust_Angular.hpp:117-119); kOrtho = ACKernelAdaptorEssentialOrtho<ThreePointKernel,
// OrthographicSymmetricEpipolarDistanceError> on hnormalized bearing vectors in xI / xJ with the bound pair_bound[p] (Eo_Robust.hpp:96-121)
enum Extra { kNone = 0, kAngular8 = 3, kUpright3 = 4, kOrtho = 5 };
double port_acransac(bool homography, const double* xI, const double* xJ, const uint64_t* start, const uint32_t* wh, uint64_t n_pairs,
                     double precision, uint32_t max_iterations, uint8_t* inlier_mask, uint8_t* ok, double* Fout,
                     double* prec, double* nfa_out, const double* K = nullptr, const double* bI = nullptr, const double* bJ = nullptr,
                     int extra = kNone, const double* pair_bound = nullptr) {
  const bool essential = K != nullptr;
  const bool angular = extra == kAngular8 || extra == kUpright3, ortho = extra == kOrtho;
  const uint32_t kMin = extra == kAngular8 ? 8 : (extra == kUpright3 || ortho) ? 3 : homography ? 4 : essential ? 5 : 7;   // Solver::MINIMUM_SAMPLES
  const double max_models = angular ? 1.0 : ortho ? 2.0 : homography ? 1.0 : essential ? 10.0 : 3.0;      // Solver::MAX_MODELS
  const double mult_error = angular ? 0.25 : homography ? 1.0 : 0.5;         // ACParametrizationHelper::MultError
  const double inf = std::numeric_limits<double>::infinity();
  for (uint64_t pp = 0; pp < n_pairs; ++pp) {
    const uint64_t lo = start[pp];
    const uint32_t n = (uint32_t)(start[pp + 1] - lo);
    std::memset(inlier_mask + lo, 0, n);
    ok[pp] = 0; prec[pp] = 0.0; nfa_out[pp] = 0.0;   // nData <= sizeSample: {0, 0} (ACRansac.hpp:354-355)
    for (int u = 0; u < 9; ++u) Fout[9 * pp + u] = (u % 4 == 0) ? 1.0 : 0.0;   // m_F = Identity
    if (n <= kMin) continue;
    // ---- ACKernelAdaptor: normalisation by the image sizes (conditioning.cpp:44-53) ----
    double T[2][3] = {{1.0, 0.0, 0.0}, {1.0, 0.0, 0.0}};   // {s, tx, ty} of image I / J
    for (int im = 0; im < 2 && !angular; ++im) {
      const int w = (int)wh[4 * pp + 2 * im], h = (int)wh[4 * pp + 2 * im + 1];
      const double dNorm = 1.0 / std::sqrt(static_cast<double>(w * h));
      T[im][0] = dNorm; T[im][1] = -.5f * w * dNorm; T[im][2] = -.5 * h * dNorm;
      if (essential || ortho) { T[im][0] = 1.0; T[im][1] = 0.0; T[im][2] = 0.0; }   // N1 = N2 = I
    }
    // essential: F = K2^-T E K1^-1 (multiview/essential.cpp:48-53), the inverses by cofactors
    double k1i[9], k2i[9];
    if (essential) {
      for (int im = 0; im < 2; ++im) {
        const double* m = K + 18 * pp + 9 * im; double* inv = im ? k2i : k1i;
        const double c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
        const double id = 1.0 / (c00 * m[0] + c10 * m[1] + c20 * m[2]);
        inv[0] = c00 * id; inv[3] = c10 * id; inv[6] = c20 * id;
        inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
        inv[2] = (m[1] * m[5] - m[2] * m[4]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
      }
    }
    std::vector<double> x1(2 * n), x2(2 * n);
    for (uint32_t i = 0; i < n && !angular; ++i) {
      x1[2 * i] = T[0][0] * xI[2 * (lo + i)] + T[0][1]; x1[2 * i + 1] = T[0][0] * xI[2 * (lo + i) + 1] + T[0][2];
      x2[2 * i] = T[1][0] * xJ[2 * (lo + i)] + T[1][1]; x2[2 * i + 1] = T[1][0] * xJ[2 * (lo + i) + 1] + T[1][2];
    }
    const int w2 = angular ? 1 : (int)wh[4 * pp + 2], h2 = angular ? 1 : (int)wh[4 * pp + 3];
    const double logalpha0 = angular ? std::log10(1. / 2.)   // RADIAN_ANGLE (robust_estimator_ACRansacKernelAdaptator.hpp:85-94)
                             : ortho ? std::log10(2. * std::hypot(w2, h2) / (w2 * static_cast<double>(h2)) / 0.5)
                             : homography ? std::log10(M_PI / (w2 * static_cast<double>(h2)) / (T[1][0] * T[1][0]))   // point to point
                             : essential ? std::log10(2. * std::hypot(w2, h2) / (w2 * static_cast<double>(h2)) / 0.5)   // LogAlpha0(w2, h2, 0.5)
                                        : std::log10(2. * std::hypot(w2, h2) / (w2 * static_cast<double>(h2)) / T[1][0]);   // point to line
    const double upper = angular ? precision * M_PI / 180.0 : ortho ? pair_bound[pp] : precision * precision;   // (D2R(precision); the functor's bound)
    const bool quantified = upper != inf;
    if (!quantified) continue;   // the exhaustive NFA form (no precision bound) is not restated: main_GeometricFilter always passes one
    const double max_threshold = upper * T[1][0] * T[1][0];
    // ---- NFA_Interface: tables ----
    const double loge0 = std::log10(max_models * (n - kMin));
    std::vector<float> l10(n + 1), logc_n(n + 1), logc_k(n + 1);
    // (the reference calls the unqualified log10 on a float: with <cmath> of libstdc++ that is the C function on a double, rounded to float)
    for (uint32_t i = 0; i <= n; ++i) l10[i] = (float)::log10((double)static_cast<float>(i));
    for (uint32_t k = 0; k <= n; ++k) { logc_n[k] = logcombi(k, n, l10); logc_k[k] = logcombi(kMin, k, l10); }
    const int nBins = 20;
    const double bins_by_interval = nBins / (max_threshold - 0.0);
    double bin_value[20];
    { const double val = (max_threshold - 0.0) / static_cast<double>(nBins - 1);
      for (int i = 0; i < nBins; ++i) bin_value[i] = val * static_cast<double>(i) + 0.0; }
    // ---- ACRANSAC ----
    std::vector<uint32_t> vec_index(n), vec_sample(kMin), vec_inliers;
    for (uint32_t i = 0; i < n; ++i) vec_index[i] = i;
    std::vector<double> residuals(n);
    double minNFA = inf, errorMax = inf;
    Model best{};
    bool have_model = false;
    int nIterReserve = (int)(max_iterations / 10);
    unsigned nIter = max_iterations - nIterReserve;
    bool ac_mode = false;
    Mt19937 rng;
    for (unsigned iter = 0; iter < nIter && iter < max_iterations; ++iter) {
      if (ac_mode) {   // rand_sampling.hpp:84-110
        if (kMin <= vec_index.size()) {
          const uint32_t last = (uint32_t)vec_index.size() - 1;
          for (uint32_t i = 0; i < kMin; ++i) std::swap(vec_index[i], vec_index[uniform_u32(rng, i, last)]);
          for (uint32_t i = 0; i < kMin; ++i) vec_sample[i] = vec_index[i];
        }
      } else {   // :43-66
        vec_sample.clear();
        while (vec_sample.size() < kMin) {
          const uint32_t s = uniform_u32(rng, 0, n - 1);
          if (std::find(vec_sample.begin(), vec_sample.end(), s) == vec_sample.end()) vec_sample.push_back(s);
        }
      }
      Model models[10], emodels[10];
      int nm;
      if (extra == kAngular8) {
        nm = eight_point_bearings(bI + 3 * lo, bJ + 3 * lo, vec_sample.data(), models);
      } else if (extra == kUpright3) {
        nm = three_point_upright(bI + 3 * lo, bJ + 3 * lo, vec_sample.data(), models);
      } else if (ortho) {
        nm = three_point_ortho(x1.data(), x2.data(), vec_sample.data(), models);
      } else if (essential) {
        nm = fivept::five_point(bI + 3 * lo, bJ + 3 * lo, vec_sample.data(), emodels);
        for (int mi = 0; mi < nm; ++mi) {   // the model evaluated on the pixels: F = K2^-T E K1^-1
          double tmp[9];
          for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) tmp[3 * r + c] = (k2i[r] * emodels[mi].f[c] + k2i[3 + r] * emodels[mi].f[3 + c]) + k2i[6 + r] * emodels[mi].f[6 + c];
          for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) models[mi].f[3 * r + c] = (tmp[3 * r] * k1i[c] + tmp[3 * r + 1] * k1i[3 + c]) + tmp[3 * r + 2] * k1i[6 + c];
        }
      } else {
        nm = homography ? four_point(x1.data(), x2.data(), vec_sample.data(), models) : seven_point(x1.data(), x2.data(), vec_sample.data(), models);
      }
      bool better = false;
      for (int mi = 0; mi < nm; ++mi) {
        for (uint32_t i = 0; i < n; ++i)
          residuals[i] = angular ? angular_error_sq(models[mi], bI + 3 * (lo + i), bJ + 3 * (lo + i)) : ortho ? ortho_error(models[mi], &x1[2 * i], &x2[2 * i])
                         : homography ? homography_error(models[mi], &x1[2 * i], &x2[2 * i]) : epipolar_error(models[mi], &x1[2 * i], &x2[2 * i]);
        if (!ac_mode) {
          unsigned nInlier = 0;
          for (uint32_t i = 0; i < n; ++i) nInlier += residuals[i] <= max_threshold;
          if (nInlier > 2.5 * kMin) ac_mode = true;
        }
        if (ac_mode) {   // ComputeNFA_and_inliers, quantified form (:196-262)
          size_t freq[20] = {0};
          for (uint32_t i = 0; i < n; ++i) {
            const double x = residuals[i];
            if (!(x < 0.0)) {
              const double t = (x - 0.0) * bins_by_interval;
              // static_cast<size_t> of a value out of range (or NaN) is undefined; x86-64's conversion yields 2^63, i.e. "overflow bin"
              const size_t b = (t >= 0.0 && t < 1.8e19) ? static_cast<size_t>(t) : (size_t)1 << 63;
              if (b < (size_t)nBins) ++freq[b];
            }
          }
          double cb_nfa = inf, cb_thr = 0.0;
          unsigned cum = 0;
          for (int bin = 0; bin < nBins; ++bin) {
            cum += (unsigned)freq[bin];
            if (cum > kMin && bin_value[bin] > std::numeric_limits<float>::epsilon()) {
              const double logalpha = logalpha0 + mult_error * std::log10(bin_value[bin] + std::numeric_limits<float>::epsilon());
              const double cur = loge0 + logalpha * (double)(cum - kMin) + logc_n[cum] + logc_k[cum];
              if (cur < cb_nfa && cur < 0) { cb_nfa = cur; cb_thr = bin_value[bin]; }
            }
          }
          if (cb_nfa < minNFA) {
            vec_inliers.clear();   // (updated even when the function then reports "not better": size <= MINIMUM_SAMPLES)
            for (uint32_t i = 0; i < n; ++i) if (residuals[i] <= cb_thr) vec_inliers.push_back(i);
            if (vec_inliers.size() > kMin) {
              better = true; minNFA = cb_nfa; errorMax = cb_thr; best = essential ? emodels[mi] : models[mi]; have_model = true;
            }
          }
        }
      }
      if (!ac_mode && iter > (unsigned)(nIterReserve * 2)) { nIter = 0; continue; }
      if (ac_mode && ((better && minNFA < 0) || ((iter + 1) == nIter && nIterReserve > 0))) {
        if (vec_inliers.empty()) { ++nIter; --nIterReserve; }
        else {
          vec_index = vec_inliers;
          if (nIterReserve) { nIter = iter + 1 + nIterReserve; nIterReserve = 0; }
        }
      }
    }
    if (minNFA >= 0) vec_inliers.clear();
    double Fm[9];
    for (int u = 0; u < 9; ++u) Fm[u] = have_model ? best.f[u] : ((u % 4 == 0) ? 1.0 : 0.0);
    if (angular) {   // ACKernelAdaptor_AngularRadianError: no normalisation, unormalizeError(val) = sqrt(val)
      if (!vec_inliers.empty()) errorMax = std::sqrt(errorMax);
    } else if (!vec_inliers.empty() && !essential && !ortho) {   // (ACKernelAdaptorEssential{,Ortho}: Unnormalize does nothing, unormalizeError(val) = val)
      // Unnormalize: F = N2^T F N1 (conditioning.cpp:87-89), errorMax -> sqrt(errorMax) / N2(0,0)
      const double N1[9] = {T[0][0], 0, T[0][1], 0, T[0][0], T[0][2], 0, 0, 1}, N2[9] = {T[1][0], 0, T[1][1], 0, T[1][0], T[1][2], 0, 0, 1};
      double tmp[9], res[9];
      if (homography) {   // UnnormalizerI (conditioning.cpp:80-82): H = N2^-1 H N1
        const double is = 1.0 / T[1][0];
        const double N2i[9] = {is, 0, -T[1][1] * is, 0, is, -T[1][2] * is, 0, 0, 1};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += N2i[3 * r + k] * Fm[3 * k + c]; tmp[3 * r + c] = s; }
      } else
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += N2[3 * k + r] * Fm[3 * k + c]; tmp[3 * r + c] = s; }
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += tmp[3 * r + k] * N1[3 * k + c]; res[3 * r + c] = s; }
      std::memcpy(Fm, res, sizeof(res));
      errorMax = std::sqrt(errorMax) / T[1][0];
    }
    const bool good = vec_inliers.size() > kMin * 2.5;
    ok[pp] = good;
    prec[pp] = errorMax; nfa_out[pp] = minNFA;
    std::memcpy(Fout + 9 * pp, Fm, sizeof(Fm));
    if (good) for (uint32_t idx : vec_inliers) inlier_mask[lo + idx] = 1;
  }
  return 0.0;
}
}  // namespace

extern "C" {
// same interface as ref_geofilter_f_acransac / ref_geofilter_h_acransac (oracle/ref_shim_geofilter.cpp), one thread
double port_geofilter_f_acransac(const double* xI, const double* xJ, const uint64_t* start, const uint32_t* wh, uint64_t n_pairs,
                                 double precision, uint32_t max_iterations, uint8_t* inlier_mask, uint8_t* ok, double* Fout,
                                 double* prec, double* nfa_out) {
  return port_acransac(false, xI, xJ, start, wh, n_pairs, precision, max_iterations, inlier_mask, ok, Fout, prec, nfa_out);
}
double port_geofilter_h_acransac(const double* xI, const double* xJ, const uint64_t* start, const uint32_t* wh, uint64_t n_pairs,
                                 double precision, uint32_t max_iterations, uint8_t* inlier_mask, uint8_t* ok, double* Fout,
                                 double* prec, double* nfa_out) {
  return port_acransac(true, xI, xJ, start, wh, n_pairs, precision, max_iterations, inlier_mask, ok, Fout, prec, nfa_out);
}
// the essential model: K = 18 doubles per pair {K_I, K_J} row-major; bI / bJ = the cameras' bearing vectors of the correspondences (3 doubles
// each, as Pinhole_Intrinsic::operator() returns them) or NULL: then normalised Kinv (x, y, 1) is formed here
double port_geofilter_e_acransac(const double* xI, const double* xJ, const uint64_t* start, const uint32_t* wh, const double* K, const double* bI,
                                 const double* bJ, uint64_t n_pairs, double precision, uint32_t max_iterations, uint8_t* inlier_mask, uint8_t* ok,
                                 double* Fout, double* prec, double* nfa_out) {
  std::vector<double> b1, b2;
  if (!bI || !bJ) {
    const uint64_t N = start[n_pairs];
    b1.resize(3 * N + 3); b2.resize(3 * N + 3);
    for (uint64_t p = 0; p < n_pairs; ++p)
      for (int im = 0; im < 2; ++im) {
        const double* m = K + 18 * p + 9 * im;
        const double c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
        const double id = 1.0 / (c00 * m[0] + c10 * m[1] + c20 * m[2]);
        const double inv[9] = {c00 * id, (m[2] * m[7] - m[1] * m[8]) * id, (m[1] * m[5] - m[2] * m[4]) * id, c10 * id, (m[0] * m[8] - m[2] * m[6]) * id,
                               (m[2] * m[3] - m[0] * m[5]) * id, c20 * id, (m[1] * m[6] - m[0] * m[7]) * id, (m[0] * m[4] - m[1] * m[3]) * id};
        const double* x = im ? xJ : xI; double* b = im ? b2.data() : b1.data();
        for (uint64_t i = start[p]; i < start[p + 1]; ++i) {
          const double v0 = inv[0] * x[2 * i] + inv[1] * x[2 * i + 1] + inv[2], v1 = inv[3] * x[2 * i] + inv[4] * x[2 * i + 1] + inv[5], v2 = inv[6] * x[2 * i] + inv[7] * x[2 * i + 1] + inv[8];
          const double nn = std::sqrt(v0 * v0 + v1 * v1 + v2 * v2);
          b[3 * i] = v0 / nn; b[3 * i + 1] = v1 / nn; b[3 * i + 2] = v2 / nn;
        }
      }
    bI = b1.data(); bJ = b2.data();
  }
  return port_acransac(false, xI, xJ, start, wh, n_pairs, precision, max_iterations, inlier_mask, ok, Fout, prec, nfa_out, K, bI, bJ);
}
// the five-point restatement alone (tests: against ref_five_point and the device's mvgx_debug_five_point)
// same interface as ref_geofilter_e_angular_acransac without the pose stage (oracle/ref_shim_geofilter.cpp); F receives m_E
double port_geofilter_e_angular_acransac(const double* bI, const double* bJ, const uint64_t* start, uint64_t n_pairs, double precision_deg, uint32_t max_iterations,
                                         int upright, uint8_t* inlier_mask, uint8_t* ok, double* Fout, double* prec, double* nfa) {
  return port_acransac(false, nullptr, nullptr, start, nullptr, n_pairs, precision_deg, max_iterations, inlier_mask, ok, Fout, prec, nfa, nullptr, bI, bJ,
                       upright ? kUpright3 : kAngular8);
}
// the orthographic essential model on hnormalized bearing vectors hI / hJ with the functor's bound per pair (Eo_Robust.hpp:96-121)
double port_geofilter_eo_acransac(const double* hI, const double* hJ, const uint64_t* start, const uint32_t* wh, const double* pair_bound, uint64_t n_pairs,
                                  uint32_t max_iterations, uint8_t* inlier_mask, uint8_t* ok, double* Fout, double* prec, double* nfa) {
  return port_acransac(false, hI, hJ, start, wh, n_pairs, 1.0, max_iterations, inlier_mask, ok, Fout, prec, nfa, nullptr, nullptr, nullptr, kOrtho, pair_bound);
}
void port_five_point(const double* b1, const double* b2, double* Es_out, int* n_out) {
  const uint32_t s[5] = {0, 1, 2, 3, 4};
  Model out[10];
  *n_out = fivept::five_point(b1, b2, s, out);
  for (int m = 0; m < *n_out; ++m) std::memcpy(Es_out + 9 * m, out[m].f, sizeof(out[m].f));
}
}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------------
// Guided matching (test infrastructure like the rest of this file): restatement of
//   /root/reference/src/openMVG/robust_estimation/guided_matching.hpp:178-227 (the Regions overload the functors call) with
//   EpipolarDistanceError (multiview/solver_fundamental_kernel.cpp:157-166, kind 0) or AsymmetricError (solver_homography_kernel.hpp:59-63,
//   kind 1), distanceRatio<double> (:68-112) and L2<uint8_t> (matching/metric.hpp:55-93) for ONE image pair.
// Sums as the reference's build forms them: F x~ = (F_i0 x0 + F_i1 x1) + F_i2 (Eigen's homogeneous product), the three-element dot
// product c0 + (c1 + c2) (Eigen's unrolled reduction), no fused multiply-add (this file is compiled with -ffp-contract=off).
// Pinned to the compiled reference through ref_guided_match (oracle/ref_shim_geofilter.cpp) in tests/test_guided_matching.py.
// out_ij: capacity 2 nI; returns the number of matches (ascending i).
// ---------------------------------------------------------------------------------------------------------------------------------
extern "C" uint64_t port_guided_match(int kind, const double* M, const double* xyI, const uint8_t* descI, uint64_t nI, const double* xyJ,
                                       const uint8_t* descJ, uint64_t nJ, uint32_t desc_bytes, double error_th, double dist_ratio, uint32_t* out_ij) {
  uint64_t n_out = 0;
  if (!(error_th < std::numeric_limits<double>::infinity())) return 0;
  for (uint64_t i = 0; i < nI; ++i) {
    const double x0 = xyI[2 * i], x1 = xyI[2 * i + 1];
    const double v0 = (M[0] * x0 + M[1] * x1) + M[2], v1 = (M[3] * x0 + M[4] * x1) + M[5], v2 = (M[6] * x0 + M[7] * x1) + M[8];
    double bd = std::numeric_limits<double>::max(), sbd = bd;
    uint64_t idx = 0;
    for (uint64_t j = 0; j < nJ; ++j) {
      const double y0 = xyJ[2 * j], y1 = xyJ[2 * j + 1];
      double err;
      if (kind == 0) {
        const double dt = v0 * y0 + (v1 * y1 + v2);
        err = (dt * dt) / (v0 * v0 + v1 * v1);
      } else {
        const double dx = y0 - v0 / v2, dy = y1 - v1 / v2;
        err = dx * dx + dy * dy;
      }
      if (!(err < error_th)) continue;
      int d = 0;
      for (uint32_t k = 0; k < desc_bytes; ++k) { const int t = (int)descI[i * desc_bytes + k] - (int)descJ[j * desc_bytes + k]; d += t * t; }
      const double dist = (double)d;
      if (dist < bd) { idx = j; sbd = bd; bd = dist; }
      else if (dist < sbd) sbd = dist;
    }
    if (sbd != std::numeric_limits<double>::max() && bd < dist_ratio * sbd) { out_ij[2 * n_out] = (uint32_t)i; out_ij[2 * n_out + 1] = (uint32_t)idx; ++n_out; }
  }
  return n_out;
}

// The same for the other region types (round 6): desc_type 0 = uint8 rows (L2<uint8_t>), 1 = float rows (L2<float>, metric.hpp:95-131:
// float sums in groups of four, result += ((d0 d0 + d1 d1) + d2 d2) + d3 d3; this file is compiled with -ffp-contract=off), 2 = bit rows under
// the SQUARED Hamming distance (binary_regions.hpp:109-120). desc_len: elements per row (bytes for types 0 and 2, floats for type 1).
extern "C" uint64_t port_guided_match_typed(int kind, int desc_type, const double* M, const double* xyI, const void* descI_, uint64_t nI, const double* xyJ,
                                             const void* descJ_, uint64_t nJ, uint32_t desc_len, double error_th, double dist_ratio, uint32_t* out_ij) {
  uint64_t n_out = 0;
  if (!(error_th < std::numeric_limits<double>::infinity())) return 0;
  const uint8_t* bI = static_cast<const uint8_t*>(descI_); const uint8_t* bJ = static_cast<const uint8_t*>(descJ_);
  const float* fI = static_cast<const float*>(descI_); const float* fJ = static_cast<const float*>(descJ_);
  for (uint64_t i = 0; i < nI; ++i) {
    const double x0 = xyI[2 * i], x1 = xyI[2 * i + 1];
    const double v0 = (M[0] * x0 + M[1] * x1) + M[2], v1 = (M[3] * x0 + M[4] * x1) + M[5], v2 = (M[6] * x0 + M[7] * x1) + M[8];
    double bd = std::numeric_limits<double>::max(), sbd = bd;
    uint64_t idx = 0;
    for (uint64_t j = 0; j < nJ; ++j) {
      const double y0 = xyJ[2 * j], y1 = xyJ[2 * j + 1];
      double err;
      if (kind == 0) {
        const double dt = v0 * y0 + (v1 * y1 + v2);
        err = (dt * dt) / (v0 * v0 + v1 * v1);
      } else {
        const double dx = y0 - v0 / v2, dy = y1 - v1 / v2;
        err = dx * dx + dy * dy;
      }
      if (!(err < error_th)) continue;
      double dist;
      if (desc_type == 0) {
        int d = 0;
        for (uint32_t k = 0; k < desc_len; ++k) { const int t = (int)bI[i * desc_len + k] - (int)bJ[j * desc_len + k]; d += t * t; }
        dist = (double)d;
      } else if (desc_type == 1) {
        const float* a = fI + i * desc_len; const float* b = fJ + j * desc_len;
        volatile float result = 0.f;
        uint32_t k = 0;
        for (; k + 3 < desc_len; k += 4) {
          const float d0 = a[k] - b[k], d1 = a[k + 1] - b[k + 1], d2 = a[k + 2] - b[k + 2], d3 = a[k + 3] - b[k + 3];
          volatile float g = d0 * d0;
          g = g + d1 * d1; g = g + d2 * d2; g = g + d3 * d3;
          result = result + g;
        }
        for (; k < desc_len; ++k) { const float d0 = a[k] - b[k]; result = result + d0 * d0; }
        dist = (double)result;
      } else {
        unsigned h = 0;
        for (uint32_t k = 0; k < desc_len; ++k) h += (unsigned)__builtin_popcount((unsigned)(bI[i * desc_len + k] ^ bJ[j * desc_len + k]));
        dist = (double)(h * h);
      }
      if (dist < bd) { idx = j; sbd = bd; bd = dist; }
      else if (dist < sbd) sbd = dist;
    }
    if (sbd != std::numeric_limits<double>::max() && bd < dist_ratio * sbd) { out_ij[2 * n_out] = (uint32_t)i; out_ij[2 * n_out + 1] = (uint32_t)idx; ++n_out; }
  }
  return n_out;
}
