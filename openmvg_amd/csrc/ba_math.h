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
// numeric values of cameras::EINTRINSIC (cameras/Camera_Common.hpp:39-50)
constexpr int kCamPinhole = 1, kCamRadial1 = 2, kCamRadial3 = 3, kCamBrown = 4, kCamFisheye = 5, kCamSpherical = 7;

// size of the intrinsic parameter block (IntrinsicBase::getParams): -1 = no functor for the model
// (sfm_data_BA_ceres.cpp:84-108). The spherical camera has no parameter block; its intrinsics row carries {w, h} as data.
MVGX_HD int intr_param_count(int model) {
  return model == kCamPinhole ? 3 : model == kCamRadial1 ? 4 : model == kCamRadial3 ? 6 : model == kCamBrown ? 8 :
         model == kCamFisheye ? 7 : model == kCamSpherical ? 0 : -1;
}

// The terms of the rotation that depend on the pose alone (one sqrt, one sin, one cos, one division): a kernel that evaluates
// many observations of one pose forms them once. trig = {cos theta, sin theta, 1 - cos theta, w0, w1, w2, 1 / theta, flag};
// flag = 0 selects the first-order branch (theta^2 <= DBL_EPSILON, ceres/rotation.h:563-622), the other entries are unused then.
constexpr int kPoseTrig = 8;
MVGX_HD void pose_trig(const double* pose, double* trig) {
  const double a0 = pose[0], a1 = pose[1], a2 = pose[2];
  const double theta2 = a0 * a0 + a1 * a1 + a2 * a2;
  if (theta2 > 2.220446049250313e-16 /* DBL_EPSILON */) {
    const double theta = sqrt(theta2);
    const double c = cos(theta), s = sin(theta);
    const double ti = 1.0 / theta;
    trig[0] = c; trig[1] = s; trig[2] = 1.0 - c;
    trig[3] = a0 * ti; trig[4] = a1 * ti; trig[5] = a2 * ti;
    trig[6] = ti; trig[7] = 1.0;
  } else {
    for (int k = 0; k < 7; ++k) trig[k] = 0.0;
    trig[7] = 0.0;
  }
}

// p = R(aa) X + t with the pose's rotation terms given. When kJac: R = dp/dX (row-major 3x3) and A = dp/d(aa) (row-major 3x3).
template <bool kJac>
MVGX_HD void transform_point_t(const double* pose, const double* trig, const double* X, double p[3], double R[9], double A[9]) {
  const double x0 = X[0], x1 = X[1], x2 = X[2];
  if (trig[7] != 0.0) {
    const double c = trig[0], s = trig[1], k = trig[2];
    const double w0 = trig[3], w1 = trig[4], w2 = trig[5], ti = trig[6];
    const double wx0 = w1 * x2 - w2 * x1, wx1 = w2 * x0 - w0 * x2, wx2 = w0 * x1 - w1 * x0;
    const double wdx = w0 * x0 + w1 * x1 + w2 * x2;
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
    const double a0 = pose[0], a1 = pose[1], a2 = pose[2];
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

// p = R(aa) X + t. When kJac: R = dp/dX (row-major 3x3) and A = dp/d(aa) (row-major 3x3).
template <bool kJac>
MVGX_HD void transform_point(const double* pose, const double* X, double p[3], double R[9], double A[9]) {
  double trig[kPoseTrig];
  pose_trig(pose, trig);
  transform_point_t<kJac>(pose, trig, X, p, R, A);
}

// Residual r = project(intr, pose, X) - obs and, when kJac, the row-major Jacobians
//   Ji (2 x 8, columns beyond the model's parameter count are 0), Jc (2 x 6: angle-axis | t), Jp (2 x 3).
// Functors of sfm_data_BA_ceres_camera_functor.hpp: pinhole :103-194, radial K1 :207-300, radial K3 :313-412,
// Brown T2 :425-545, fisheye :548-660, spherical :662-760.
// kPinholeFamily: the caller guarantees model is one of pinhole / radial K1 / radial K3 / Brown T2 (the polynomial models): the
// branches of the spherical and fisheye functors (atan2, atan, sqrt, divisions) are compiled out - in the fused point-group kernels
// they cost registers on every problem although almost no scene uses them.
template <bool kJac, bool kPinholeFamily = false>
// off: the observation is switched off (mvgx_ba_update_subset). Its residual and rows are multiplied by zero afterwards, so they must be
// FINITE whatever the (possibly stale) point is: the point is evaluated as if it sat on the optical axis at unit depth - no division by a
// vanishing depth, no overflowing distortion polynomial, (0, 0, 1) is regular for every camera model (selecting the three
// camera-frame coordinates costs three v_cndmask, clearing the 34 Jacobian entries after the fact cost the fused kernels 100 - 160 bytes
// of scratch and 8 - 10 % of their time, call r5_09).
MVGX_HD void eval_observation_t(int model, const double* intr, const double* pose, const double* trig, const double* X, const double* obs,
                                double r[2], double* Ji, double* Jc, double* Jp, bool off = false) {
  double p[3], R[9], A[9];
  transform_point_t<kJac>(pose, trig, X, p, R, A);
  if (off) { p[0] = 0.0; p[1] = 0.0; p[2] = 1.0; }
  double g00, g01, g02, g10, g11, g12;   // G = d r / d p (2 x 3)
  if (kJac)
    for (int c = 0; c < 16; ++c) Ji[c] = 0.0;
  if (!kPinholeFamily && model == kCamSpherical) {
    // lon = atan2(x, z), lat = atan2(-y, |(x, z)|); r = (lon, -lat) size / 2pi + (w, h) / 2 - obs, size = max(w, h)
    const double w = intr[0], h = intr[1];
    const double size = w > h ? w : h;
    const double k = size / (2.0 * 3.14159265358979323846);
    const double rho2 = p[0] * p[0] + p[2] * p[2];
    const double rho = sqrt(rho2);
    const double lon = atan2(p[0], p[2]);
    const double lat = atan2(-p[1], rho);
    r[0] = lon * k + w / 2.0 - obs[0];
    r[1] = -lat * k + h / 2.0 - obs[1];
    if (kJac) {
      const double n2 = p[1] * p[1] + rho2;
      g00 = k * p[2] / rho2; g01 = 0.0; g02 = -k * p[0] / rho2;
      // d lat / d p = (y x / (rho n2), -rho / n2, y z / (rho n2));  r1 = -k lat
      g10 = -k * (p[1] * p[0] / (rho * n2)); g11 = k * rho / n2; g12 = -k * (p[1] * p[2] / (rho * n2));
    }
  } else {
    const double iz = 1.0 / p[2];
    const double u = p[0] * iz, v = p[1] * iz;
    const double f = intr[0];
    const double r2 = u * u + v * v;
    double xd = u, yd = v;                       // distorted normalised coordinates
    double m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0;   // d (xd, yd) / d (u, v)
    if (model == kCamRadial1 || model == kCamRadial3 || model == kCamBrown) {
      const double k1 = intr[3], k2 = model == kCamRadial1 ? 0.0 : intr[4], k3 = model == kCamRadial1 ? 0.0 : intr[5];
      const double r4 = r2 * r2, r6 = r4 * r2;
      const double coeff = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
      const double dc = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
      xd = u * coeff; yd = v * coeff;
      if (kJac) {
        m00 = coeff + 2.0 * u * u * dc; m01 = 2.0 * u * v * dc; m10 = m01; m11 = coeff + 2.0 * v * v * dc;
        Ji[3] = f * u * r2; Ji[8 + 3] = f * v * r2;
        if (model != kCamRadial1) {
          Ji[4] = f * u * r4; Ji[8 + 4] = f * v * r4;
          Ji[5] = f * u * r6; Ji[8 + 5] = f * v * r6;
        }
      }
      if (model == kCamBrown) {
        const double t1 = intr[6], t2 = intr[7];
        xd += t2 * (r2 + 2.0 * u * u) + 2.0 * t1 * u * v;
        yd += t1 * (r2 + 2.0 * v * v) + 2.0 * t2 * u * v;
        if (kJac) {
          m00 += 6.0 * t2 * u + 2.0 * t1 * v; m01 += 2.0 * t2 * v + 2.0 * t1 * u;
          m10 += 2.0 * t1 * u + 2.0 * t2 * v; m11 += 6.0 * t1 * v + 2.0 * t2 * u;
          Ji[6] = f * 2.0 * u * v;            Ji[8 + 6] = f * (r2 + 2.0 * v * v);
          Ji[7] = f * (r2 + 2.0 * u * u);     Ji[8 + 7] = f * 2.0 * u * v;
        }
      }
    } else if (!kPinholeFamily && model == kCamFisheye) {
      const double rr = sqrt(r2);
      if (rr > 1e-8) {   // else cdist = 1 (a constant in the reference's functor: zero derivative)
        const double th = atan(rr);
        const double th2 = th * th, th3 = th2 * th, th5 = th3 * th2, th7 = th5 * th2, th9 = th7 * th2;
        const double thd = th + intr[3] * th3 + intr[4] * th5 + intr[5] * th7 + intr[6] * th9;
        const double ir = 1.0 / rr;
        const double cdist = thd * ir;
        xd = u * cdist; yd = v * cdist;
        if (kJac) {
          const double dthd = 1.0 + 3.0 * intr[3] * th2 + 5.0 * intr[4] * th2 * th2 + 7.0 * intr[5] * th3 * th3 + 9.0 * intr[6] * th5 * th3;
          const double dcd = (dthd / (1.0 + r2) - cdist) * ir;      // d cdist / d r
          const double cu = dcd * u * ir, cv = dcd * v * ir;        // d cdist / d u, d v
          m00 = cdist + u * cu; m01 = u * cv; m10 = v * cu; m11 = cdist + v * cv;
          Ji[3] = f * u * th3 * ir; Ji[8 + 3] = f * v * th3 * ir;
          Ji[4] = f * u * th5 * ir; Ji[8 + 4] = f * v * th5 * ir;
          Ji[5] = f * u * th7 * ir; Ji[8 + 5] = f * v * th7 * ir;
          Ji[6] = f * u * th9 * ir; Ji[8 + 6] = f * v * th9 * ir;
        }
      }
    }
    r[0] = intr[1] + xd * f - obs[0];
    r[1] = intr[2] + yd * f - obs[1];
    if (kJac) {
      Ji[0] = xd; Ji[8] = yd;        // d/d focal
      Ji[1] = 1.0; Ji[8 + 2] = 1.0;  // d/d ppx, d/d ppy
      const double j00 = f * m00, j01 = f * m01, j10 = f * m10, j11 = f * m11;
      // d(u,v)/dp = [[iz, 0, -u iz], [0, iz, -v iz]]
      g00 = j00 * iz; g01 = j01 * iz; g02 = -(j00 * u + j01 * v) * iz;
      g10 = j10 * iz; g11 = j11 * iz; g12 = -(j10 * u + j11 * v) * iz;
    }
  }
  if (kJac) {
    for (int c = 0; c < 3; ++c) {
      Jp[c] = g00 * R[c] + g01 * R[3 + c] + g02 * R[6 + c];
      Jp[3 + c] = g10 * R[c] + g11 * R[3 + c] + g12 * R[6 + c];
      Jc[c] = g00 * A[c] + g01 * A[3 + c] + g02 * A[6 + c];
      Jc[6 + c] = g10 * A[c] + g11 * A[3 + c] + g12 * A[6 + c];
    }
    Jc[3] = g00; Jc[4] = g01; Jc[5] = g02;
    Jc[9] = g10; Jc[10] = g11; Jc[11] = g12;
  }
}

template <bool kJac>
MVGX_HD void eval_observation(int model, const double* intr, const double* pose, const double* X, const double* obs,
                              double r[2], double* Ji, double* Jc, double* Jp, bool off = false) {
  double trig[kPoseTrig];
  pose_trig(pose, trig);
  eval_observation_t<kJac>(model, intr, pose, trig, X, obs, r, Ji, Jc, Jp, off);
}

// PoseCenterConstraintCostFunction (sfm_data_BA_ceres.cpp:44-80): r = weight o (C(pose) - prior), C = -R(-aa) t.
// When kJac: Jc (3 x 6, row-major) = [d r / d aa | d r / d t].
// Unit ray of an image observation in world coordinates: R^T * bearing(get_ud_pixel(x)), normalised — the two vectors
// AngleBetweenRay (cameras/Camera_Intrinsics.hpp:263-280) compares, as RemoveOutliers_AngleError
// (sfm/sfm_data_filters.cpp:77-121) calls it. Undistortion per model:
//   pinhole / spherical: identity (Camera_Pinhole.hpp:268-271, Camera_Spherical.hpp:175)
//   radial K1 / K3: bisection on r^2 (Camera_Pinhole_Radial.hpp:37-70, :148-158, :273-277, :357-367, :482-486)
//   Brown T2: fixed-point iteration, Manhattan stop 1e-10 (Camera_Pinhole_Brown.hpp:97-110, :226-236); the reference loop
//             is unbounded — here it gives up after kBrownMaxIter rounds (a diverging model has no meaningful ray anyway)
//   fisheye: 10 fixed-point rounds on theta, then tan (Camera_Pinhole_Fisheye.hpp:112-136)
// Bearing: Kinv (x, y, 1) normalised (Camera_Pinhole.hpp:136-139); spherical lon/lat (Camera_Spherical.hpp:115-132).
constexpr int kBrownMaxIter = 10000;
constexpr int kBisectMaxIter = 40000;   // bracket search (ratio 1.05: < 30 000 rounds across the double range) + bisection together

MVGX_HD double radial_disto_r2(int model, const double* intr, double r2) {
  const double k1 = intr[3];
  if (model == kCamRadial1) {
    const double c = 1.0 + r2 * k1;
    return r2 * (c * c);
  }
  const double c = 1.0 + r2 * (k1 + r2 * (intr[4] + r2 * intr[5]));
  return r2 * (c * c);
}

MVGX_HD void observation_ray(int model, const double* intr, const double* pose, const double* obs, double ray[3]) {
  double b0, b1, b2;
  if (model == kCamSpherical) {
    const double w = intr[0], h = intr[1];
    const double size = w > h ? w : h;
    const double ux = (obs[0] - w / 2.0) / size, uy = (obs[1] - h / 2.0) / size;
    const double lon = ux * 2 * 3.14159265358979323846, lat = -uy * 2 * 3.14159265358979323846;
    b0 = cos(lat) * sin(lon); b1 = -sin(lat); b2 = cos(lat) * cos(lon);
  } else {
    const double f = intr[0], cx = intr[1], cy = intr[2];
    double px = (obs[0] - cx) / f, py = (obs[1] - cy) / f;   // ima2cam
    if (model == kCamRadial1 || model == kCamRadial3) {
      const double r2 = px * px + py * py;
      double radius = 1.0;
      if (r2 != 0.0) {
        // the reference's loops are unbounded; they end within a few hundred rounds for every finite input whose interval
        // can shrink below 1e-10 in double precision - a device thread must not spin on garbage (|x| ~ 1e300), hence the cap
        double lo = r2, up = r2;
        int guard = 0;
        while (radial_disto_r2(model, intr, lo) > r2 && guard++ < kBisectMaxIter) lo /= 1.05;
        while (radial_disto_r2(model, intr, up) < r2 && guard++ < kBisectMaxIter) up *= 1.05;
        while (1e-10 < up - lo && guard++ < kBisectMaxIter) {
          const double mid = .5 * (lo + up);
          if (radial_disto_r2(model, intr, mid) > r2) up = mid; else lo = mid;
        }
        radius = sqrt(.5 * (lo + up) / r2);
      }
      px *= radius; py *= radius;
    } else if (model == kCamBrown) {
      const double k1 = intr[3], k2 = intr[4], k3 = intr[5], t1 = intr[6], t2 = intr[7];
      double ux = px, uy = py, dx, dy;
      for (int it = 0;; ++it) {
        const double r2 = ux * ux + uy * uy, r4 = r2 * r2, r6 = r4 * r2;
        const double kd = k1 * r2 + k2 * r4 + k3 * r6;
        dx = ux * kd + (t2 * (r2 + 2 * ux * ux) + 2 * t1 * ux * uy);
        dy = uy * kd + (t1 * (r2 + 2 * uy * uy) + 2 * t2 * ux * uy);
        if (!(fabs(ux + dx - px) + fabs(uy + dy - py) > 1e-10) || it >= kBrownMaxIter) break;
        ux = px - dx; uy = py - dy;
      }
      px = ux; py = uy;
    } else if (model == kCamFisheye) {
      const double td = hypot(px, py);
      double scale = 1.0;
      if (td > 1e-8) {
        double th = td;
        for (int j = 0; j < 10; ++j) {
          const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
          th = td / (1 + intr[3] * t2 + intr[4] * t4 + intr[5] * t6 + intr[6] * t8);
        }
        scale = tan(th) / td;
      }
      px *= scale; py *= scale;
    }
    const double xu = f * px + cx, yu = f * py + cy;   // cam2ima: the undistorted pixel ...
    b0 = (xu - cx) / f; b1 = (yu - cy) / f; b2 = 1.0;  // ... and its bearing
    const double n = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
    b0 /= n; b1 /= n; b2 /= n;
  }
  double p[3], R[9], A[9];
  const double zero[3] = {0.0, 0.0, 0.0};
  transform_point<true>(pose, zero, p, R, A);   // R row-major
  const double r0 = R[0] * b0 + R[3] * b1 + R[6] * b2;
  const double r1 = R[1] * b0 + R[4] * b1 + R[7] * b2;
  const double r2_ = R[2] * b0 + R[5] * b1 + R[8] * b2;
  const double n = sqrt(r0 * r0 + r1 * r1 + r2_ * r2_);
  ray[0] = r0 / n; ray[1] = r1 / n; ray[2] = r2_ / n;
}

// R2D(acos(clamp(dot, -1 + 1e-8, 1 - 1e-8))) (Camera_Intrinsics.hpp:278-279, numeric.h:72-75,135-138)
MVGX_HD double ray_angle_deg(const double a[3], const double b[3]) {
  double dt = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  dt = dt < 1.0 - 1.e-8 ? dt : 1.0 - 1.e-8;
  dt = dt > -1.0 + 1.e-8 ? dt : -1.0 + 1.e-8;
  return acos(dt) / 3.14159265358979323846 * 180.0;
}

template <bool kJac>
MVGX_HD void eval_pose_center_prior(const double* pose, const double* center, const double* weight, double r[3], double* Jc) {
  const double neg[6] = {-pose[0], -pose[1], -pose[2], 0.0, 0.0, 0.0};
  double p[3], R[9], A[9];
  transform_point<kJac>(neg, pose + 3, p, R, A);   // p = R(-aa) t, R = dp/dt, A = dp/dw at w = -aa
  for (int k = 0; k < 3; ++k) r[k] = weight[k] * (-p[k] - center[k]);
  if (kJac)
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) {
        Jc[k * 6 + c] = weight[k] * A[k * 3 + c];        // C = -p(w(aa)), dw/daa = -1
        Jc[k * 6 + 3 + c] = -weight[k] * R[k * 3 + c];
      }
}

// HuberLoss::Evaluate. rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s). a = 0 is a legal (degenerate) scale the
// pose-centre priors can produce, so "no loss function" is a separate switch.
MVGX_HD void huber_rho_on(bool loss, double a, double s, double rho[3]) {
  const double b = a * a;
  if (loss && s > b) {
    const double rr = sqrt(s);
    rho[0] = 2.0 * a * rr - b;
    const double r1 = a / rr;
    rho[1] = r1 > 2.2250738585072014e-308 ? r1 : 2.2250738585072014e-308;
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}
// observation residuals: a <= 0 means the problem was built without a loss function (TrivialLoss)
MVGX_HD void huber_rho(double a, double s, double rho[3]) { huber_rho_on(a > 0.0, a, s, rho); }

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

// Inverse of the Cholesky factor of a symmetric positive definite 3x3 (v = {a00, a01, a02, a11, a12, a22}):
// V = L L^T, li = {i00, i10, i11, i20, i21, i22} = the lower-triangular L^-1, so that V^-1 = L^-T L^-1
// (what invert_psd_matrix.h:49-72 computes through Eigen's LLT). Returns false if not positive definite.
MVGX_HD bool chol_inv3(const double v[6], double li[6]) {
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
  li[0] = i00; li[1] = i10; li[2] = i11; li[3] = i20; li[4] = i21; li[5] = i22;
  return true;
}

}  // namespace mvgx_ba
