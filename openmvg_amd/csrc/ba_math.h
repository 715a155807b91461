// ba_math.h — closed-form per-observation math of the BA kernels (fp64), usable from device and host code.
//
// What the reference computes with Ceres autodiff Jets (paths under /root/reference/src):
//   openMVG/sfm/sfm_data_BA_ceres_camera_functor.hpp:124-164 (pinhole), :228-270 (radial K1), :337-382 (radial K3)
//   third_party/ceres-solver/include/ceres/rotation.h:563-622 (AngleAxisRotatePoint, both branches)
//   third_party/ceres-solver/internal/ceres/loss_function.cc:47-61 (HuberLoss), corrector.cc:41-155 (Corrector)
// is written here as explicit derivatives: no dual numbers, no Eigen, no Ceres on the device.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define MVGX_HD __host__ __device__ __forceinline__
#else
#define MVGX_HD inline
#endif

namespace mvgx_ba {

constexpr int kMaxIntr = 8;
constexpr int kCamPinhole = 1, kCamRadial1 = 2, kCamRadial3 = 3;

MVGX_HD int intr_param_count(int model) {
  return model == kCamPinhole ? 3 : model == kCamRadial1 ? 4 : model == kCamRadial3 ? 6 : -1;
}

// p = R(aa) X + t. When kJac: R = dp/dX (row-major 3x3) and A = dp/d(aa) (row-major 3x3).
template <bool kJac>
MVGX_HD void transform_point(const double* pose, const double* X, double p[3], double R[9], double A[9]) {
  const double a0 = pose[0], a1 = pose[1], a2 = pose[2];
  const double x0 = X[0], x1 = X[1], x2 = X[2];
  const double theta2 = a0 * a0 + a1 * a1 + a2 * a2;
  if (theta2 > 2.220446049250313e-16 /* DBL_EPSILON */) {
    const double theta = sqrt(theta2);
    const double c = cos(theta), s = sin(theta);
    const double ti = 1.0 / theta;
    const double w0 = a0 * ti, w1 = a1 * ti, w2 = a2 * ti;
    const double wx0 = w1 * x2 - w2 * x1, wx1 = w2 * x0 - w0 * x2, wx2 = w0 * x1 - w1 * x0;
    const double wdx = w0 * x0 + w1 * x1 + w2 * x2;
    const double k = 1.0 - c;
    const double tmp = wdx * k;
    p[0] = x0 * c + wx0 * s + w0 * tmp;
    p[1] = x1 * c + wx1 * s + w1 * tmp;
    p[2] = x2 * c + wx2 * s + w2 * tmp;
    if (kJac) {
      // R = c I + s [w]x + k w w^T
      R[0] = c + k * w0 * w0;      R[1] = -s * w2 + k * w0 * w1; R[2] = s * w1 + k * w0 * w2;
      R[3] = s * w2 + k * w1 * w0; R[4] = c + k * w1 * w1;       R[5] = -s * w0 + k * w1 * w2;
      R[6] = -s * w1 + k * w2 * w0; R[7] = s * w0 + k * w2 * w1; R[8] = c + k * w2 * w2;
      // dp/dtheta = -s X + c (w x X) + s (w.X) w
      const double q0 = -s * x0 + c * wx0 + s * wdx * w0;
      const double q1 = -s * x1 + c * wx1 + s * wdx * w1;
      const double q2 = -s * x2 + c * wx2 + s * wdx * w2;
      // M = dp/dw = -s [X]x + k (w X^T + (w.X) I)
      const double M[9] = {k * (w0 * x0 + wdx), s * x2 + k * w0 * x1,  -s * x1 + k * w0 * x2,
                           -s * x2 + k * w1 * x0, k * (w1 * x1 + wdx), s * x0 + k * w1 * x2,
                           s * x1 + k * w2 * x0,  -s * x0 + k * w2 * x1, k * (w2 * x2 + wdx)};
      // A = q w^T + M (I - w w^T) / theta
      const double w[3] = {w0, w1, w2};
      const double q[3] = {q0, q1, q2};
      for (int r = 0; r < 3; ++r) {
        const double mw = M[r * 3] * w0 + M[r * 3 + 1] * w1 + M[r * 3 + 2] * w2;
        for (int cidx = 0; cidx < 3; ++cidx) A[r * 3 + cidx] = q[r] * w[cidx] + (M[r * 3 + cidx] - mw * w[cidx]) * ti;
      }
    }
  } else {
    // first-order branch: p = X + aa x X
    p[0] = x0 + (a1 * x2 - a2 * x1);
    p[1] = x1 + (a2 * x0 - a0 * x2);
    p[2] = x2 + (a0 * x1 - a1 * x0);
    if (kJac) {
      R[0] = 1;   R[1] = -a2; R[2] = a1;
      R[3] = a2;  R[4] = 1;   R[5] = -a0;
      R[6] = -a1; R[7] = a0;  R[8] = 1;
      A[0] = 0;   A[1] = x2;  A[2] = -x1;   // d(aa x X)/d(aa) = -[X]x
      A[3] = -x2; A[4] = 0;   A[5] = x0;
      A[6] = x1;  A[7] = -x0; A[8] = 0;
    }
  }
  p[0] += pose[3]; p[1] += pose[4]; p[2] += pose[5];
}

// Residual r = project(intr, pose, X) - obs and, when kJac, the row-major Jacobians
//   Ji (2 x 8, columns beyond the model's parameter count are 0), Jc (2 x 6: angle-axis | t), Jp (2 x 3).
template <bool kJac>
MVGX_HD void eval_observation(int model, const double* intr, const double* pose, const double* X, const double* obs,
                              double r[2], double* Ji, double* Jc, double* Jp) {
  double p[3], R[9], A[9];
  transform_point<kJac>(pose, X, p, R, A);
  const double iz = 1.0 / p[2];
  const double u = p[0] * iz, v = p[1] * iz;
  const double f = intr[0];
  double coeff = 1.0, dc = 0.0, r2 = 0.0, r4 = 0.0, r6 = 0.0;
  if (model != kCamPinhole) {
    r2 = u * u + v * v;
    if (model == kCamRadial1) {
      coeff = 1.0 + intr[3] * r2;
      dc = intr[3];
    } else {
      r4 = r2 * r2;
      r6 = r4 * r2;
      coeff = 1.0 + intr[3] * r2 + intr[4] * r4 + intr[5] * r6;
      dc = intr[3] + 2.0 * intr[4] * r2 + 3.0 * intr[5] * r4;
    }
  }
  const double xd = u * coeff, yd = v * coeff;
  r[0] = intr[1] + xd * f - obs[0];
  r[1] = intr[2] + yd * f - obs[1];
  if (kJac) {
    // d(res)/d(u,v)
    const double j00 = f * (coeff + 2.0 * u * u * dc), j01 = f * (2.0 * u * v * dc);
    const double j10 = j01, j11 = f * (coeff + 2.0 * v * v * dc);
    // d(u,v)/dp = [[iz, 0, -u iz], [0, iz, -v iz]]
    const double g00 = j00 * iz, g01 = j01 * iz, g02 = -(j00 * u + j01 * v) * iz;
    const double g10 = j10 * iz, g11 = j11 * iz, g12 = -(j10 * u + j11 * v) * iz;
    for (int c = 0; c < 3; ++c) {
      Jp[c] = g00 * R[c] + g01 * R[3 + c] + g02 * R[6 + c];
      Jp[3 + c] = g10 * R[c] + g11 * R[3 + c] + g12 * R[6 + c];
      Jc[c] = g00 * A[c] + g01 * A[3 + c] + g02 * A[6 + c];
      Jc[6 + c] = g10 * A[c] + g11 * A[3 + c] + g12 * A[6 + c];
    }
    Jc[3] = g00; Jc[4] = g01; Jc[5] = g02;
    Jc[9] = g10; Jc[10] = g11; Jc[11] = g12;
    for (int c = 0; c < 16; ++c) Ji[c] = 0.0;
    Ji[0] = xd; Ji[8] = yd;        // d/d focal
    Ji[1] = 1.0; Ji[8 + 2] = 1.0;  // d/d ppx, d/d ppy
    if (model != kCamPinhole) {
      Ji[3] = f * u * r2; Ji[8 + 3] = f * v * r2;
      if (model == kCamRadial3) {
        Ji[4] = f * u * r4; Ji[8 + 4] = f * v * r4;
        Ji[5] = f * u * r6; Ji[8 + 5] = f * v * r6;
      }
    }
  }
}

// HuberLoss::Evaluate (a <= 0: TrivialLoss). rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s).
MVGX_HD void huber_rho(double a, double s, double rho[3]) {
  const double b = a * a;
  if (a > 0.0 && s > b) {
    const double rr = sqrt(s);
    rho[0] = 2.0 * a * rr - b;
    const double r1 = a / rr;
    rho[1] = r1 > 2.2250738585072014e-308 ? r1 : 2.2250738585072014e-308;
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

// Corrector for rho'' <= 0 (always the case for Huber / trivial loss): residual and Jacobian scale by sqrt(rho').
MVGX_HD double corrector_scale(const double rho[3]) { return sqrt(rho[1]); }

// inverse of a symmetric positive definite 3x3 (v = {a00, a01, a02, a11, a12, a22}) through its Cholesky factor
// (invert_psd_matrix.h:49-72, full-rank branch). Returns false if not positive definite.
MVGX_HD bool invert_spd3(const double v[6], double inv[6]) {
  if (!(v[0] > 0.0)) return false;
  const double l00 = sqrt(v[0]);
  const double l10 = v[1] / l00, l20 = v[2] / l00;
  const double d1 = v[3] - l10 * l10;
  if (!(d1 > 0.0)) return false;
  const double l11 = sqrt(d1);
  const double l21 = (v[4] - l20 * l10) / l11;
  const double d2 = v[5] - l20 * l20 - l21 * l21;
  if (!(d2 > 0.0)) return false;
  const double l22 = sqrt(d2);
  const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  inv[0] = i00 * i00 + i10 * i10 + i20 * i20;
  inv[1] = i10 * i11 + i20 * i21;
  inv[2] = i20 * i22;
  inv[3] = i11 * i11 + i21 * i21;
  inv[4] = i21 * i22;
  inv[5] = i22 * i22;
  return true;
}

}  // namespace mvgx_ba
