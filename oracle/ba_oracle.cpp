/*
 * ba_oracle.cpp — TEST INFRASTRUCTURE ONLY (never linked into, imported by, or called from the product path).
 *
 * Plain C++ CPU restatement of the reference's bundle-adjustment path: openMVG's problem construction and residual
 * functors, and the part of vendored Ceres 1.13.0 that a Solve() of that problem executes (autodiff Jacobians, Huber
 * corrector, Jacobi scaling, Levenberg-Marquardt trust region, Schur elimination of the point blocks, dense reduced
 * solve, back-substitution). Every function cites the reference lines it follows; paths are under
 * /root/reference/src ("ceres/" = third_party/ceres-solver).
 *
 * Pinned against the reference itself: tests/test_oracle_ba.py runs this next to oracle/_ref/libref_ba.so
 * (Bundle_Adjustment_Ceres::Adjust compiled in place) on the scenes of the reference's own sfm_data_BA_test.cpp shape
 * and asserts equal final RMSE (1e-9) and the reference's test assertions (RMSE decreases, Adjust() succeeds).
 *
 * Derivatives here are computed with forward-mode dual numbers ("Jets"), as Ceres' AutoDiffCostFunction does
 * (ceres/include/ceres/internal/autodiff.h:208, jet.h:172) — deliberately NOT the closed-form Jacobians the HIP
 * kernels use, so the two derivations check each other.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/mvgx.h"  // struct layouts only (mvgx_ba_problem / _options / _summary)

namespace {

// ---------------------------------------------------------------------------------------------------------
// Forward-mode dual number, ceres/include/ceres/jet.h:172 (value + N partials; +,-,*,/, sqrt, sin, cos)
// ---------------------------------------------------------------------------------------------------------
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0) { for (int i = 0; i < N; ++i) v[i] = 0; }
  Jet(double x) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0; }  // NOLINT
  Jet(double x, int k) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0; v[k] = 1.0; }
};
template <int N> Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> Jet<N> operator-(const Jet<N>& f) { Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi;
  return h;
}
template <int N> Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> Jet<N> operator+(double s, const Jet<N>& f) { return f + s; }
template <int N> Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> Jet<N> operator-(double s, const Jet<N>& f) { return Jet<N>(s) - f; }
template <int N> Jet<N> operator*(const Jet<N>& f, double s) { Jet<N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <int N> Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> Jet<N> operator/(double s, const Jet<N>& g) { return Jet<N>(s) / g; }
template <int N> Jet<N> operator/(const Jet<N>& f, double s) { return f * (1.0 / s); }
template <int N> bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> Jet<N> sqrt(const Jet<N>& f) { Jet<N> h; h.a = std::sqrt(f.a); const double d = 1.0 / (2.0 * h.a); for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * d; return h; }
template <int N> Jet<N> sin(const Jet<N>& f) { Jet<N> h; h.a = std::sin(f.a); const double c = std::cos(f.a); for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h; }
template <int N> Jet<N> cos(const Jet<N>& f) { Jet<N> h; h.a = std::cos(f.a); const double s = -std::sin(f.a); for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i]; return h; }
// jet.h: atan, atan2(g, f) = atan(g / f) with the quadrant of (f, g)
template <int N> Jet<N> atan(const Jet<N>& f) { Jet<N> h; h.a = std::atan(f.a); const double d = 1.0 / (1.0 + f.a * f.a); for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }
template <int N> Jet<N> atan2(const Jet<N>& g, const Jet<N>& f) {
  Jet<N> h; h.a = std::atan2(g.a, f.a); const double t = 1.0 / (f.a * f.a + g.a * g.a);
  for (int i = 0; i < N; ++i) h.v[i] = t * (-g.a * f.v[i] + f.a * g.v[i]);
  return h;
}
inline double value_of(double x) { return x; }
template <int N> double value_of(const Jet<N>& x) { return x.a; }
using std::atan;
using std::atan2;
using std::cos;
using std::sin;
using std::sqrt;

// ceres/include/ceres/rotation.h:563-622 — AngleAxisRotatePoint: Rodrigues if theta^2 > DBL_EPSILON, else pt + w x pt
template <typename T>
void angle_axis_rotate_point(const T aa[3], const T pt[3], T out[3]) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (value_of(theta2) > std::numeric_limits<double>::epsilon()) {
    const T theta = sqrt(theta2);
    const T costheta = cos(theta);
    const T sintheta = sin(theta);
    const T theta_inverse = T(1.0) / theta;
    const T w[3] = {aa[0] * theta_inverse, aa[1] * theta_inverse, aa[2] * theta_inverse};
    const T wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - costheta);
    out[0] = pt[0] * costheta + wxp[0] * sintheta + w[0] * tmp;
    out[1] = pt[1] * costheta + wxp[1] * sintheta + w[1] * tmp;
    out[2] = pt[2] * costheta + wxp[2] * sintheta + w[2] * tmp;
  } else {
    const T wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0]};
    out[0] = pt[0] + wxp[0];
    out[1] = pt[1] + wxp[1];
    out[2] = pt[2] + wxp[2];
  }
}

int intr_param_count(int model) {
  switch (model) {
    case MVGX_CAM_PINHOLE: return 3;          // cameras/Camera_Pinhole.hpp:215-223 getParams {f, ppx, ppy}
    case MVGX_CAM_PINHOLE_RADIAL1: return 4;  // cameras/Camera_Pinhole_Radial.hpp K1: + k1
    case MVGX_CAM_PINHOLE_RADIAL3: return 6;  // cameras/Camera_Pinhole_Radial.hpp:340-348 K3: + k1 k2 k3
    case MVGX_CAM_PINHOLE_BROWN: return 8;    // cameras/Camera_Pinhole_Brown.hpp getParams: + k1 k2 k3 t1 t2
    case MVGX_CAM_PINHOLE_FISHEYE: return 7;  // cameras/Camera_Pinhole_Fisheye.hpp getParams: + k1 k2 k3 k4
    case MVGX_CAM_SPHERICAL: return 0;        // cameras/Camera_Spherical.hpp: getParams() is empty; {w, h} are data
    default: return -1;
  }
}

// sfm/sfm_data_BA_ceres_camera_functor.hpp:124-164 (pinhole), :228-270 (radial K1), :337-382 (radial K3),
// :446-500 (Brown T2), :569-618 (fisheye), :681-711 (spherical):
// x_u = hnormalized(R(aa) X + t); r = pp + f * distort(x_u) - obs
template <typename T>
void reprojection_residual(int model, const T* intr, const T* pose, const T* X, const double obs[2], T r[2]) {
  T p[3];
  angle_axis_rotate_point(pose, X, p);
  p[0] = p[0] + pose[3]; p[1] = p[1] + pose[4]; p[2] = p[2] + pose[5];
  if (model == MVGX_CAM_SPHERICAL) {
    // intr = {w, h} (image size, data): lon/lat of the bearing vector, normalised by 2 pi, scaled by max(w, h)
    const double w = value_of(intr[0]), h = value_of(intr[1]);
    const T lon = atan2(p[0], p[2]);
    const T lat = atan2(-p[1], sqrt(p[0] * p[0] + p[2] * p[2]));
    const T c0 = lon / (2 * M_PI), c1 = -lat / (2 * M_PI);
    const double size = std::max(w, h);
    r[0] = c0 * size + w / 2.0 - obs[0];
    r[1] = c1 * size + h / 2.0 - obs[1];
    return;
  }
  const T u = p[0] / p[2], v = p[1] / p[2];
  const T& focal = intr[0];
  const T& ppx = intr[1];
  const T& ppy = intr[2];
  if (model == MVGX_CAM_PINHOLE) {
    r[0] = ppx + u * focal - obs[0];
    r[1] = ppy + v * focal - obs[1];
  } else if (model == MVGX_CAM_PINHOLE_RADIAL1) {
    const T r2 = u * u + v * v;
    const T coeff = T(1.0) + intr[3] * r2;
    r[0] = ppx + (u * coeff) * focal - obs[0];
    r[1] = ppy + (v * coeff) * focal - obs[1];
  } else if (model == MVGX_CAM_PINHOLE_RADIAL3) {
    const T r2 = u * u + v * v;
    const T r4 = r2 * r2;
    const T r6 = r4 * r2;
    const T coeff = T(1.0) + intr[3] * r2 + intr[4] * r4 + intr[5] * r6;
    r[0] = ppx + (u * coeff) * focal - obs[0];
    r[1] = ppy + (v * coeff) * focal - obs[1];
  } else if (model == MVGX_CAM_PINHOLE_BROWN) {
    const T& k1 = intr[3]; const T& k2 = intr[4]; const T& k3 = intr[5]; const T& t1 = intr[6]; const T& t2 = intr[7];
    const T r2 = u * u + v * v;
    const T r4 = r2 * r2;
    const T r6 = r4 * r2;
    const T r_coeff = T(1.0) + k1 * r2 + k2 * r4 + k3 * r6;
    const T t_x = t2 * (r2 + 2.0 * u * u) + 2.0 * t1 * u * v;
    const T t_y = t1 * (r2 + 2.0 * v * v) + 2.0 * t2 * u * v;
    r[0] = ppx + (u * r_coeff + t_x) * focal - obs[0];
    r[1] = ppy + (v * r_coeff + t_y) * focal - obs[1];
  } else {  // MVGX_CAM_PINHOLE_FISHEYE
    const T& k1 = intr[3]; const T& k2 = intr[4]; const T& k3 = intr[5]; const T& k4 = intr[6];
    const T r2 = u * u + v * v;
    const T rr = sqrt(r2);
    const T theta = atan(rr), theta2 = theta * theta, theta3 = theta2 * theta, theta4 = theta2 * theta2, theta5 = theta4 * theta,
            theta7 = theta3 * theta3 * theta, theta8 = theta4 * theta4, theta9 = theta8 * theta;
    const T theta_dist = theta + k1 * theta3 + k2 * theta5 + k3 * theta7 + k4 * theta9;
    const T inv_r = rr > T(1e-8) ? T(1.0) / rr : T(1.0);
    const T cdist = rr > T(1e-8) ? theta_dist * inv_r : T(1.0);
    r[0] = ppx + (u * cdist) * focal - obs[0];
    r[1] = ppy + (v * cdist) * focal - obs[1];
  }
}

// PoseCenterConstraintCostFunction (sfm_data_BA_ceres.cpp:44-80): weight o (-(R(-aa) t) - prior centre)
template <typename T>
void pose_center_residual(const T* pose, const double center[3], const double weight[3], T r[3]) {
  const T neg[3] = {-pose[0], -pose[1], -pose[2]};
  T c[3];
  angle_axis_rotate_point(neg, pose + 3, c);
  for (int k = 0; k < 3; ++k) r[k] = (c[k] * -1.0 - center[k]) * weight[k];
}

constexpr int kJetN = 8 + 6 + 3;  // intrinsics (<= 8) | pose (6) | point (3)

// AutoDiffCostFunction::Evaluate (ceres/include/ceres/autodiff_cost_function.h:202-219): residual + row-major 2 x k Jacobians
void eval_obs_autodiff(int model, const double* intr, const double* pose, const double* X, const double* obs,
                       double r[2], double Ji[16], double Jc[12], double Jp[6]) {
  typedef Jet<kJetN> J;
  J ji[8], jc[6], jx[3], jr[2];
  const int K = intr_param_count(model);
  for (int k = 0; k < 8; ++k) ji[k] = (k < K) ? J(intr[k], k) : J(intr[k]);   // beyond K: data (spherical {w, h}), no partials
  for (int k = 0; k < 6; ++k) jc[k] = J(pose[k], 8 + k);
  for (int k = 0; k < 3; ++k) jx[k] = J(X[k], 14 + k);
  reprojection_residual<J>(model, ji, jc, jx, obs, jr);
  for (int row = 0; row < 2; ++row) {
    r[row] = jr[row].a;
    for (int k = 0; k < 8; ++k) Ji[row * 8 + k] = jr[row].v[k];
    for (int k = 0; k < 6; ++k) Jc[row * 6 + k] = jr[row].v[8 + k];
    for (int k = 0; k < 3; ++k) Jp[row * 3 + k] = jr[row].v[14 + k];
  }
}

// ceres/internal/ceres/loss_function.cc:47-61 — HuberLoss(a): b = a^2
void huber_on(bool loss, double a, double s, double rho[3]) {
  const double b = a * a;
  if (loss && s > b) {
    const double r = std::sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}
void huber(double a, double s, double rho[3]) { huber_on(a > 0.0, a, s, rho); }   // a <= 0: problem built without a loss function

struct Problem {
  uint32_t n_poses = 0, n_intr = 0, n_points = 0;
  uint64_t n_obs = 0;
  std::vector<double> poses, intr, points;
  std::vector<int> model;
  std::vector<uint32_t> op, oi, ox;
  std::vector<double> oxy;
  std::vector<uint8_t> pose_mask, intr_mask;
  bool points_constant = false;
  double huber_a = 0;
  std::vector<double> oweight;          // per observation (empty: all unweighted)
  std::vector<uint8_t> octrl;           // per observation: control-point residual (no loss, not in the RMSE)
  std::vector<uint8_t> point_const;     // per point: SetParameterBlockConstant
  std::vector<uint32_t> prior_pose;     // pose-centre priors
  std::vector<double> prior_center, prior_weight;
  double prior_huber_a = 0;
  uint64_t n_obs_rmse = 0;
  bool is_const(uint32_t j) const { return points_constant || (!point_const.empty() && point_const[j]); }
  double weight(uint64_t k) const { return (!oweight.empty() && oweight[k] != 0.0) ? oweight[k] : 1.0; }
  bool ctrl(uint64_t k) const { return !octrl.empty() && octrl[k]; }
  // derived: local columns
  std::vector<int> pose_col, intr_col;             // first reduced-system column of the block or -1 (constant / unused)
  std::vector<std::vector<int>> pose_free, intr_free;  // free component indices of each block
  std::vector<uint8_t> point_used;
  int ncols = 0;
  std::vector<std::vector<uint64_t>> obs_of_point;
};

bool load(const mvgx_ba_problem* p, Problem& P) {
  P.n_poses = p->n_poses; P.n_intr = p->n_intrinsics; P.n_points = p->n_points; P.n_obs = p->n_obs;
  P.poses.assign(p->poses, p->poses + size_t(P.n_poses) * 6);
  P.intr.assign(p->intrinsics, p->intrinsics + size_t(P.n_intr) * MVGX_BA_MAX_INTR_PARAMS);
  P.points.assign(p->points, p->points + size_t(P.n_points) * 3);
  P.model.assign(p->intr_model, p->intr_model + P.n_intr);
  P.op.assign(p->obs_pose, p->obs_pose + P.n_obs);
  P.oi.assign(p->obs_intr, p->obs_intr + P.n_obs);
  P.ox.assign(p->obs_point, p->obs_point + P.n_obs);
  P.oxy.assign(p->obs_xy, p->obs_xy + 2 * P.n_obs);
  P.pose_mask.assign(P.n_poses, 0);
  P.intr_mask.assign(P.n_intr, 0);
  if (p->pose_const_mask) P.pose_mask.assign(p->pose_const_mask, p->pose_const_mask + P.n_poses);
  if (p->intr_const_mask) P.intr_mask.assign(p->intr_const_mask, p->intr_const_mask + P.n_intr);
  P.points_constant = p->points_constant != 0;
  P.huber_a = p->huber_a;
  if (p->obs_weight) P.oweight.assign(p->obs_weight, p->obs_weight + P.n_obs);
  if (p->obs_is_control) P.octrl.assign(p->obs_is_control, p->obs_is_control + P.n_obs);
  if (p->point_const_mask) P.point_const.assign(p->point_const_mask, p->point_const_mask + P.n_points);
  if (p->n_pose_priors) {
    P.prior_pose.assign(p->prior_pose, p->prior_pose + p->n_pose_priors);
    P.prior_center.assign(p->prior_center, p->prior_center + 3 * size_t(p->n_pose_priors));
    P.prior_weight.assign(p->prior_weight, p->prior_weight + 3 * size_t(p->n_pose_priors));
    P.prior_huber_a = p->prior_huber_a;
  }
  P.n_obs_rmse = 0;
  for (uint64_t k = 0; k < P.n_obs; ++k) if (!P.ctrl(k)) ++P.n_obs_rmse;
  for (uint32_t i = 0; i < P.n_intr; ++i) if (intr_param_count(P.model[i]) < 0) return false;
  // Blocks no residual references are dropped by Ceres' preprocessor (program.cc RemoveFixedBlocks), like constants.
  std::vector<uint8_t> pose_used(P.n_poses, 0), intr_used(P.n_intr, 0);
  P.point_used.assign(P.n_points, 0);
  P.obs_of_point.assign(P.n_points, {});
  for (uint64_t k = 0; k < P.n_obs; ++k) {
    if (P.op[k] >= P.n_poses || P.oi[k] >= P.n_intr || P.ox[k] >= P.n_points) return false;
    pose_used[P.op[k]] = 1; intr_used[P.oi[k]] = 1; P.point_used[P.ox[k]] = 1;
    P.obs_of_point[P.ox[k]].push_back(k);
  }
  for (uint32_t q : P.prior_pose) { if (q >= P.n_poses) return false; pose_used[q] = 1; }
  // reduced camera system column layout: poses in index order, then intrinsics
  P.pose_col.assign(P.n_poses, -1); P.intr_col.assign(P.n_intr, -1);
  P.pose_free.assign(P.n_poses, {}); P.intr_free.assign(P.n_intr, {});
  int col = 0;
  for (uint32_t i = 0; i < P.n_poses; ++i) {
    if (!pose_used[i]) continue;
    for (int c = 0; c < 6; ++c) if (!((P.pose_mask[i] >> c) & 1)) P.pose_free[i].push_back(c);
    if (P.pose_free[i].empty()) continue;
    P.pose_col[i] = col; col += int(P.pose_free[i].size());
  }
  for (uint32_t i = 0; i < P.n_intr; ++i) {
    if (!intr_used[i]) continue;
    const int K = intr_param_count(P.model[i]);
    for (int c = 0; c < K; ++c) if (!((P.intr_mask[i] >> c) & 1)) P.intr_free[i].push_back(c);
    if (P.intr_free[i].empty()) continue;   // constant, or no parameter block at all (spherical)
    P.intr_col[i] = col; col += int(P.intr_free[i].size());
  }
  P.ncols = col;
  return true;
}

struct ObsLin {           // one residual block after ResidualBlock::Evaluate (ceres/internal/ceres/residual_block.cc:68-196)
  double r[2];            // corrected residuals
  double E[6];            // 2 x 3 point Jacobian (corrected)
  double Fc[12];          // 2 x 6 pose Jacobian (global columns, corrected)
  double Fi[16];          // 2 x 8 intrinsic Jacobian
};

struct PriorLin { double r[3]; double Fc[18]; };   // pose-centre prior: corrected residual and 3 x 6 pose Jacobian

double prior_cost(const Problem& P, const std::vector<double>& poses) {
  double c = 0;
  for (size_t q = 0; q < P.prior_pose.size(); ++q) {
    double r[3];
    pose_center_residual<double>(&poses[size_t(P.prior_pose[q]) * 6], &P.prior_center[3 * q], &P.prior_weight[3 * q], r);
    double rho[3];
    huber_on(true, P.prior_huber_a, r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho);
    c += 0.5 * rho[0];
  }
  return c;
}

double linearize_priors(const Problem& P, const std::vector<double>& poses, std::vector<PriorLin>& L) {
  typedef Jet<6> J;
  L.resize(P.prior_pose.size());
  double c = 0;
  for (size_t q = 0; q < P.prior_pose.size(); ++q) {
    const double* pose = &poses[size_t(P.prior_pose[q]) * 6];
    J jp[6], jr[3];
    for (int k = 0; k < 6; ++k) jp[k] = J(pose[k], k);
    pose_center_residual<J>(jp, &P.prior_center[3 * q], &P.prior_weight[3 * q], jr);
    PriorLin& o = L[q];
    double s = 0;
    for (int k = 0; k < 3; ++k) { o.r[k] = jr[k].a; s += o.r[k] * o.r[k]; for (int cc = 0; cc < 6; ++cc) o.Fc[k * 6 + cc] = jr[k].v[cc]; }
    double rho[3];
    huber_on(true, P.prior_huber_a, s, rho);
    c += 0.5 * rho[0];
    const double sr = std::sqrt(rho[1]);   // rho'' <= 0: Corrector scales residual and Jacobian by sqrt(rho') (corrector.cc:81-85)
    for (int k = 0; k < 3; ++k) o.r[k] *= sr;
    for (int k = 0; k < 18; ++k) o.Fc[k] *= sr;
  }
  return c;
}

// Cost only: 0.5 * rho(|r|^2) summed (residual_block.cc:168-176); also the loss-free squared error for the RMSE.
void evaluate_cost(const Problem& P, const std::vector<double>& poses, const std::vector<double>& intr,
                   const std::vector<double>& points, double* cost, double* sq_err) {
  double c = 0, se = 0;
  for (uint64_t k = 0; k < P.n_obs; ++k) {
    double r[2];
    reprojection_residual<double>(P.model[P.oi[k]], &intr[size_t(P.oi[k]) * 8], &poses[size_t(P.op[k]) * 6],
                                  &points[size_t(P.ox[k]) * 3], &P.oxy[2 * k], r);
    const double w = P.weight(k);   // WeightedCostFunction (camera_functor.hpp:35-90)
    r[0] *= w; r[1] *= w;
    const double s = r[0] * r[0] + r[1] * r[1];
    double rho[3];
    huber_on(!P.ctrl(k) && P.huber_a > 0.0, P.huber_a, s, rho);   // control-point blocks carry no loss (:421-435)
    c += 0.5 * rho[0];
    if (!P.ctrl(k)) se += s;
  }
  c += prior_cost(P, poses);
  *cost = c; *sq_err = se;
}

// Residual + Jacobian + loss correction (corrector.cc:41-110,112-155: rho'' <= 0 for Huber -> plain sqrt(rho') scaling)
void linearize(const Problem& P, const std::vector<double>& poses, const std::vector<double>& intr,
               const std::vector<double>& points, std::vector<ObsLin>& L, double* cost) {
  L.resize(P.n_obs);
  double c = 0;
  for (uint64_t k = 0; k < P.n_obs; ++k) {
    ObsLin& o = L[k];
    eval_obs_autodiff(P.model[P.oi[k]], &intr[size_t(P.oi[k]) * 8], &poses[size_t(P.op[k]) * 6],
                      &points[size_t(P.ox[k]) * 3], &P.oxy[2 * k], o.r, o.Fi, o.Fc, o.E);
    const double w = P.weight(k);
    if (w != 1.0) {
      o.r[0] *= w; o.r[1] *= w;
      for (double& v : o.E) v *= w;
      for (double& v : o.Fc) v *= w;
      for (double& v : o.Fi) v *= w;
    }
    const double s = o.r[0] * o.r[0] + o.r[1] * o.r[1];
    double rho[3];
    huber_on(!P.ctrl(k) && P.huber_a > 0.0, P.huber_a, s, rho);
    c += 0.5 * rho[0];
    const double sqrt_rho1 = std::sqrt(rho[1]);
    double residual_scaling = sqrt_rho1, alpha_sq_norm = 0.0;
    if (!(s == 0.0 || rho[2] <= 0.0)) {
      const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
      const double alpha = 1.0 - std::sqrt(D);
      residual_scaling = sqrt_rho1 / (1 - alpha);
      alpha_sq_norm = alpha / s;
    }
    auto correct = [&](double* J, int ncol) {
      if (alpha_sq_norm == 0.0) { for (int i = 0; i < 2 * ncol; ++i) J[i] *= sqrt_rho1; return; }
      for (int cc = 0; cc < ncol; ++cc) {
        const double rtj = J[cc] * o.r[0] + J[ncol + cc] * o.r[1];
        for (int rr = 0; rr < 2; ++rr) J[rr * ncol + cc] = sqrt_rho1 * (J[rr * ncol + cc] - alpha_sq_norm * o.r[rr] * rtj);
      }
    };
    correct(o.E, 3); correct(o.Fc, 6); correct(o.Fi, 8);
    o.r[0] *= residual_scaling; o.r[1] *= residual_scaling;
  }
  *cost = c;
}

// Dense Cholesky solve of the reduced camera system (schur_complement_solver.cc:180-224: Eigen LLT<Upper>).
bool cholesky_solve(std::vector<double>& S, int n, std::vector<double>& b) {
  for (int j = 0; j < n; ++j) {
    double d = S[size_t(j) * n + j];
    for (int k = 0; k < j; ++k) d -= S[size_t(j) * n + k] * S[size_t(j) * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    S[size_t(j) * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double v = S[size_t(i) * n + j];
      for (int k = 0; k < j; ++k) v -= S[size_t(i) * n + k] * S[size_t(j) * n + k];
      S[size_t(i) * n + j] = v / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= S[size_t(i) * n + k] * b[k];
    b[i] = v / S[size_t(i) * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = b[i];
    for (int k = i + 1; k < n; ++k) v -= S[size_t(k) * n + i] * b[k];
    b[i] = v / S[size_t(i) * n + i];
  }
  return true;
}

// invert_psd_matrix.h:49-72 (full-rank branch): inverse of the 3x3 SPD block through its Cholesky factor
bool invert_spd3(const double A[9], double inv[9]) {
  const double l00 = std::sqrt(A[0]);
  if (!(A[0] > 0)) return false;
  const double l10 = A[3] / l00, l20 = A[6] / l00;
  const double d1 = A[4] - l10 * l10;
  if (!(d1 > 0)) return false;
  const double l11 = std::sqrt(d1);
  const double l21 = (A[7] - l20 * l10) / l11;
  const double d2 = A[8] - l20 * l20 - l21 * l21;
  if (!(d2 > 0)) return false;
  const double l22 = std::sqrt(d2);
  // inverse of L (lower)
  const double i00 = 1 / l00, i11 = 1 / l11, i22 = 1 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  // A^-1 = L^-T L^-1
  inv[0] = i00 * i00 + i10 * i10 + i20 * i20;
  inv[1] = inv[3] = i10 * i11 + i20 * i21;
  inv[2] = inv[6] = i20 * i22;
  inv[4] = i11 * i11 + i21 * i21;
  inv[5] = inv[7] = i21 * i22;
  inv[8] = i22 * i22;
  return true;
}

struct Solver {
  const Problem& P;
  mvgx_ba_options opt;
  std::vector<double> poses, intr, points;       // x_
  std::vector<double> cposes, cintr, cpoints;    // candidate_x_
  std::vector<ObsLin> L;                         // Jacobian at x_ (unscaled, loss-corrected)
  std::vector<PriorLin> LP;                      // rows of the pose-centre priors
  std::vector<double> scale_cam, scale_pt;       // jacobian_scaling_ (trust_region_minimizer.cc:239-254)
  std::vector<double> diag_cam, diag_pt;         // LM `diagonal_` (levenberg_marquardt_strategy.cc:75-87), already clamped
  std::vector<double> step_cam, step_pt;         // trust_region_step_ (scaled space)
  double x_cost = 0, candidate_cost = 0, model_cost_change = 0, radius = 0, decrease_factor = 2.0;
  double gradient_max_norm = 0;
  bool reuse_diagonal = false;
  explicit Solver(const Problem& p, const mvgx_ba_options& o) : P(p), opt(o) {}

  // squared column norms of the (optionally scaled) Jacobian and the gradient J^T r
  void column_norms_and_gradient(bool scaled, std::vector<double>& n_cam, std::vector<double>& n_pt,
                                 std::vector<double>& g_cam, std::vector<double>& g_pt) const {
    n_cam.assign(P.ncols, 0); g_cam.assign(P.ncols, 0);
    n_pt.assign(size_t(P.n_points) * 3, 0); g_pt.assign(size_t(P.n_points) * 3, 0);
    for (uint64_t k = 0; k < P.n_obs; ++k) {
      const ObsLin& o = L[k];
      const uint32_t ip = P.op[k], ii = P.oi[k], ix = P.ox[k];
      if (P.pose_col[ip] >= 0)
        for (size_t c = 0; c < P.pose_free[ip].size(); ++c) {
          const int gc = P.pose_free[ip][c], col = P.pose_col[ip] + int(c);
          const double s = scaled ? scale_cam[col] : 1.0;
          const double j0 = o.Fc[gc] * s, j1 = o.Fc[6 + gc] * s;
          n_cam[col] += j0 * j0 + j1 * j1; g_cam[col] += j0 * o.r[0] + j1 * o.r[1];
        }
      if (P.intr_col[ii] >= 0)
        for (size_t c = 0; c < P.intr_free[ii].size(); ++c) {
          const int gc = P.intr_free[ii][c], col = P.intr_col[ii] + int(c);
          const double s = scaled ? scale_cam[col] : 1.0;
          const double j0 = o.Fi[gc] * s, j1 = o.Fi[8 + gc] * s;
          n_cam[col] += j0 * j0 + j1 * j1; g_cam[col] += j0 * o.r[0] + j1 * o.r[1];
        }
      if (!P.is_const(ix))
        for (int c = 0; c < 3; ++c) {
          const double s = scaled ? scale_pt[size_t(ix) * 3 + c] : 1.0;
          const double j0 = o.E[c] * s, j1 = o.E[3 + c] * s;
          n_pt[size_t(ix) * 3 + c] += j0 * j0 + j1 * j1; g_pt[size_t(ix) * 3 + c] += j0 * o.r[0] + j1 * o.r[1];
        }
    }
    for (size_t q = 0; q < LP.size(); ++q) {   // prior rows: 3 residuals on one pose block
      const uint32_t ip = P.prior_pose[q];
      if (P.pose_col[ip] < 0) continue;
      for (size_t c = 0; c < P.pose_free[ip].size(); ++c) {
        const int gc = P.pose_free[ip][c], col = P.pose_col[ip] + int(c);
        const double s = scaled ? scale_cam[col] : 1.0;
        for (int k = 0; k < 3; ++k) { const double j = LP[q].Fc[k * 6 + gc] * s; n_cam[col] += j * j; g_cam[col] += j * LP[q].r[k]; }
      }
    }
  }

  // TrustRegionMinimizer::EvaluateGradientAndJacobian (trust_region_minimizer.cc:226-279)
  void evaluate_gradient_and_jacobian(bool iteration_zero) {
    linearize(P, poses, intr, points, L, &x_cost);
    x_cost += linearize_priors(P, poses, LP);
    std::vector<double> n_cam, n_pt, g_cam, g_pt;
    column_norms_and_gradient(false, n_cam, n_pt, g_cam, g_pt);
    if (iteration_zero) {
      scale_cam.assign(P.ncols, 1.0); scale_pt.assign(size_t(P.n_points) * 3, 1.0);
      if (opt.jacobi_scaling) {
        for (int i = 0; i < P.ncols; ++i) scale_cam[i] = 1.0 / (1.0 + std::sqrt(n_cam[i]));
        for (size_t i = 0; i < scale_pt.size(); ++i) scale_pt[i] = 1.0 / (1.0 + std::sqrt(n_pt[i]));
      }
    }
    gradient_max_norm = 0;  // |Plus(x,-g) - x|_inf = max |g| over the free components
    for (double g : g_cam) gradient_max_norm = std::max(gradient_max_norm, std::fabs(g));
    for (uint32_t j = 0; j < P.n_points; ++j)
      if (P.point_used[j] && !P.is_const(j)) for (int c = 0; c < 3; ++c) gradient_max_norm = std::max(gradient_max_norm, std::fabs(g_pt[size_t(j) * 3 + c]));
  }

  // LevenbergMarquardtStrategy::ComputeStep (levenberg_marquardt_strategy.cc:65-145) with the Schur-complement solver
  // (schur_complement_solver.cc:120-157, schur_eliminator_impl.h:176-410). Returns false on LINEAR_SOLVER_FAILURE.
  bool compute_step() {
    const int n = P.ncols;
    if (!reuse_diagonal) {
      std::vector<double> g_cam, g_pt;
      column_norms_and_gradient(true, diag_cam, diag_pt, g_cam, g_pt);
      for (double& d : diag_cam) d = std::min(std::max(d, opt.min_lm_diagonal), opt.max_lm_diagonal);
      for (double& d : diag_pt) d = std::min(std::max(d, opt.min_lm_diagonal), opt.max_lm_diagonal);
    }
    reuse_diagonal = true;
    std::vector<double> S(size_t(n) * n, 0.0), rhs(n, 0.0);
    for (int i = 0; i < n; ++i) S[size_t(i) * n + i] = diag_cam[i] / radius;  // D^2 on the f-block diagonals (:190-211)
    struct Blk { int col0; int w; double Y[24]; };  // E^T F of one f-block touched by the current point (3 x w)
    std::vector<double> Vinv(size_t(P.n_points) * 9, 0.0), gpt(size_t(P.n_points) * 3, 0.0);
    std::vector<Blk> blocks;
    auto scaled_row = [&](const ObsLin& o, uint64_t k, double* Fs /*2 x n_local*/, int* cols, int& nl) {
      nl = 0;
      const uint32_t ip = P.op[k], ii = P.oi[k];
      if (P.pose_col[ip] >= 0)
        for (size_t c = 0; c < P.pose_free[ip].size(); ++c) {
          const int col = P.pose_col[ip] + int(c), gc = P.pose_free[ip][c];
          Fs[nl] = o.Fc[gc] * scale_cam[col]; Fs[16 + nl] = o.Fc[6 + gc] * scale_cam[col]; cols[nl++] = col;
        }
      if (P.intr_col[ii] >= 0)
        for (size_t c = 0; c < P.intr_free[ii].size(); ++c) {
          const int col = P.intr_col[ii] + int(c), gc = P.intr_free[ii][c];
          Fs[nl] = o.Fi[gc] * scale_cam[col]; Fs[16 + nl] = o.Fi[8 + gc] * scale_cam[col]; cols[nl++] = col;
        }
    };
    // one chunk per 3-D point (rows of a point are contiguous after reorder_program.cc:259-...,531-538)
    for (uint32_t j = 0; j < P.n_points; ++j) {
      const auto& obs = P.obs_of_point[j];
      if (obs.empty()) continue;
      if (P.is_const(j)) {  // no e-block: NoEBlockRowsUpdate (:556-576): S += F^T F, rhs += F^T b
        for (uint64_t k : obs) {
          double Fs[32]; int cols[16]; int nl;
          scaled_row(L[k], k, Fs, cols, nl);
          for (int a = 0; a < nl; ++a) {
            rhs[cols[a]] += Fs[a] * L[k].r[0] + Fs[16 + a] * L[k].r[1];
            for (int b = 0; b < nl; ++b) S[size_t(cols[a]) * n + cols[b]] += Fs[a] * Fs[b] + Fs[16 + a] * Fs[16 + b];
          }
        }
        continue;
      }
      double V[9] = {0}, g[3] = {0};
      for (int c = 0; c < 3; ++c) V[c * 3 + c] = diag_pt[size_t(j) * 3 + c] / radius;  // ete = D^2 (:236-243)
      std::vector<int> ycols; std::vector<double> Y;  // dense 3 x (touched columns) buffer = E^T F (:434-490)
      for (uint64_t k : obs) {
        const ObsLin& o = L[k];
        double Es[6];
        for (int c = 0; c < 3; ++c) { Es[c] = o.E[c] * scale_pt[size_t(j) * 3 + c]; Es[3 + c] = o.E[3 + c] * scale_pt[size_t(j) * 3 + c]; }
        for (int a = 0; a < 3; ++a) {
          g[a] += Es[a] * o.r[0] + Es[3 + a] * o.r[1];
          for (int b = 0; b < 3; ++b) V[a * 3 + b] += Es[a] * Es[b] + Es[3 + a] * Es[3 + b];
        }
        double Fs[32]; int cols[16]; int nl;
        scaled_row(o, k, Fs, cols, nl);
        for (int a = 0; a < nl; ++a) {
          rhs[cols[a]] += Fs[a] * o.r[0] + Fs[16 + a] * o.r[1];                      // rhs += F^T b
          for (int b = 0; b < nl; ++b) S[size_t(cols[a]) * n + cols[b]] += Fs[a] * Fs[b] + Fs[16 + a] * Fs[16 + b];  // S += F^T F
          size_t pos = std::find(ycols.begin(), ycols.end(), cols[a]) - ycols.begin();
          if (pos == ycols.size()) { ycols.push_back(cols[a]); Y.insert(Y.end(), 3, 0.0); }
          for (int e = 0; e < 3; ++e) Y[pos * 3 + e] += Es[e] * Fs[a] + Es[3 + e] * Fs[16 + a];
        }
      }
      double Vi[9];
      if (!invert_spd3(V, Vi)) return false;
      std::memcpy(&Vinv[size_t(j) * 9], Vi, sizeof(Vi));
      std::memcpy(&gpt[size_t(j) * 3], g, sizeof(g));
      double Vig[3];
      for (int a = 0; a < 3; ++a) Vig[a] = Vi[a * 3] * g[0] + Vi[a * 3 + 1] * g[1] + Vi[a * 3 + 2] * g[2];
      const size_t m = ycols.size();
      for (size_t a = 0; a < m; ++a) {
        double T[3];  // (V^-1 Y_a)
        for (int e = 0; e < 3; ++e) T[e] = Vi[e * 3] * Y[a * 3] + Vi[e * 3 + 1] * Y[a * 3 + 1] + Vi[e * 3 + 2] * Y[a * 3 + 2];
        rhs[ycols[a]] -= Y[a * 3] * Vig[0] + Y[a * 3 + 1] * Vig[1] + Y[a * 3 + 2] * Vig[2];   // UpdateRhs (:374-410)
        for (size_t b = 0; b < m; ++b)                                                          // ChunkOuterProduct (:499-548)
          S[size_t(ycols[b]) * n + ycols[a]] -= Y[b * 3] * T[0] + Y[b * 3 + 1] * T[1] + Y[b * 3 + 2] * T[2];
      }
    }
    for (size_t q = 0; q < LP.size(); ++q) {   // prior rows have no e-block either: S += F^T F, rhs += F^T b
      const uint32_t ip = P.prior_pose[q];
      if (P.pose_col[ip] < 0) continue;
      for (size_t a = 0; a < P.pose_free[ip].size(); ++a) {
        const int ca = P.pose_col[ip] + int(a), ga = P.pose_free[ip][a];
        for (int k = 0; k < 3; ++k) rhs[ca] += LP[q].Fc[k * 6 + ga] * scale_cam[ca] * LP[q].r[k];
        for (size_t b = 0; b < P.pose_free[ip].size(); ++b) {
          const int cb = P.pose_col[ip] + int(b), gb = P.pose_free[ip][b];
          for (int k = 0; k < 3; ++k) S[size_t(ca) * n + cb] += LP[q].Fc[k * 6 + ga] * scale_cam[ca] * LP[q].Fc[k * 6 + gb] * scale_cam[cb];
        }
      }
    }
    step_cam.assign(n, 0.0);
    if (n > 0) {
      std::vector<double> z = rhs;
      if (!cholesky_solve(S, n, z)) return false;
      step_cam = z;
    }
    // BackSubstitute (:303-366): y_p = (E^T E + D^2)^-1 (E^T b - E^T F z)
    step_pt.assign(size_t(P.n_points) * 3, 0.0);
      for (uint32_t j = 0; j < P.n_points; ++j) {
        const auto& obs = P.obs_of_point[j];
        if (obs.empty() || P.is_const(j)) continue;
        double t[3] = {gpt[size_t(j) * 3], gpt[size_t(j) * 3 + 1], gpt[size_t(j) * 3 + 2]};
        for (uint64_t k : obs) {
          const ObsLin& o = L[k];
          double Fs[32]; int cols[16]; int nl;
          scaled_row(o, k, Fs, cols, nl);
          double fz0 = 0, fz1 = 0;
          for (int a = 0; a < nl; ++a) { fz0 += Fs[a] * step_cam[cols[a]]; fz1 += Fs[16 + a] * step_cam[cols[a]]; }
          for (int c = 0; c < 3; ++c) {
            const double s = scale_pt[size_t(j) * 3 + c];
            t[c] -= o.E[c] * s * fz0 + o.E[3 + c] * s * fz1;
          }
        }
        const double* Vi = &Vinv[size_t(j) * 9];
        for (int a = 0; a < 3; ++a) step_pt[size_t(j) * 3 + a] = Vi[a * 3] * t[0] + Vi[a * 3 + 1] * t[1] + Vi[a * 3 + 2] * t[2];
      }
    for (double& v : step_cam) { if (!std::isfinite(v)) return false; v = -v; }  // step = -solution (:120)
    for (double& v : step_pt) { if (!std::isfinite(v)) return false; v = -v; }
    // model_cost_change = -(J step)^T (r + J step / 2)   (trust_region_minimizer.cc:402-405)
    double mc = 0;
    for (uint64_t k = 0; k < P.n_obs; ++k) {
      const ObsLin& o = L[k];
      double Fs[32]; int cols[16]; int nl;
      scaled_row(o, k, Fs, cols, nl);
      double m0 = 0, m1 = 0;
      for (int a = 0; a < nl; ++a) { m0 += Fs[a] * step_cam[cols[a]]; m1 += Fs[16 + a] * step_cam[cols[a]]; }
      if (!P.is_const(P.ox[k])) {
        const uint32_t j = P.ox[k];
        for (int c = 0; c < 3; ++c) {
          const double s = scale_pt[size_t(j) * 3 + c] * step_pt[size_t(j) * 3 + c];
          m0 += o.E[c] * s; m1 += o.E[3 + c] * s;
        }
      }
      mc -= m0 * (o.r[0] + m0 / 2.0) + m1 * (o.r[1] + m1 / 2.0);
    }
    for (size_t q = 0; q < LP.size(); ++q) {
      const uint32_t ip = P.prior_pose[q];
      if (P.pose_col[ip] < 0) continue;
      for (int k = 0; k < 3; ++k) {
        double m = 0;
        for (size_t a = 0; a < P.pose_free[ip].size(); ++a) {
          const int ca = P.pose_col[ip] + int(a);
          m += LP[q].Fc[k * 6 + P.pose_free[ip][a]] * scale_cam[ca] * step_cam[ca];
        }
        mc -= m * (LP[q].r[k] + m / 2.0);
      }
    }
    model_cost_change = mc;
    return true;
  }

  // candidate_x_ = Plus(x_, delta_), delta_ = step o jacobian_scaling (:411, :718-734); returns |delta| and |x|
  void make_candidate(double* step_norm, double* x_norm) {
    cposes = poses; cintr = intr; cpoints = points;
    double sn = 0, xn = 0;
    for (uint32_t i = 0; i < P.n_poses; ++i) {
      if (P.pose_col[i] < 0) continue;
      for (size_t c = 0; c < P.pose_free[i].size(); ++c) {
        const int col = P.pose_col[i] + int(c);
        const double d = step_cam[col] * scale_cam[col];
        cposes[size_t(i) * 6 + P.pose_free[i][c]] += d; sn += d * d;
      }
      for (int c = 0; c < 6; ++c) xn += poses[size_t(i) * 6 + c] * poses[size_t(i) * 6 + c];
    }
    for (uint32_t i = 0; i < P.n_intr; ++i) {
      if (P.intr_col[i] < 0) continue;
      for (size_t c = 0; c < P.intr_free[i].size(); ++c) {
        const int col = P.intr_col[i] + int(c);
        const double d = step_cam[col] * scale_cam[col];
        cintr[size_t(i) * 8 + P.intr_free[i][c]] += d; sn += d * d;
      }
      for (int c = 0; c < intr_param_count(P.model[i]); ++c) xn += intr[size_t(i) * 8 + c] * intr[size_t(i) * 8 + c];
    }
      for (uint32_t j = 0; j < P.n_points; ++j) {
        if (!P.point_used[j] || P.is_const(j)) continue;
        for (int c = 0; c < 3; ++c) {
          const double d = step_pt[size_t(j) * 3 + c] * scale_pt[size_t(j) * 3 + c];
          cpoints[size_t(j) * 3 + c] += d; sn += d * d;
          xn += points[size_t(j) * 3 + c] * points[size_t(j) * 3 + c];
        }
      }
    *step_norm = std::sqrt(sn); *x_norm = std::sqrt(xn);
  }
};

}  // namespace

extern "C" {

int oracle_ba_eval_obs(int model, const double* intr, const double* pose, const double* X, const double* obs,
                       double* r, double* Ji, double* Jc, double* Jp) {
  if (intr_param_count(model) < 0) return 1;
  eval_obs_autodiff(model, intr, pose, X, obs, r, Ji, Jc, Jp);
  return 0;
}

int oracle_ba_eval_prior(const double* pose, const double* center, const double* weight, double* r, double* Jc) {
  typedef Jet<6> J;
  J jp[6], jr[3];
  for (int k = 0; k < 6; ++k) jp[k] = J(pose[k], k);
  pose_center_residual<J>(jp, center, weight, jr);
  for (int k = 0; k < 3; ++k) { r[k] = jr[k].a; for (int c = 0; c < 6; ++c) Jc[k * 6 + c] = jr[k].v[c]; }
  return 0;
}

// RemoveOutliers_AngleError (sfm/sfm_data_filters.cpp:77-121): per track the maximum over observation pairs of
// AngleBetweenRay (cameras/Camera_Intrinsics.hpp:263-280) on get_ud_pixel'd observations. Flat arrays in the layout of
// mvgx_ba_problem; observations in any order. out[n_points] in degrees (0 for tracks with < 2 observations).
namespace {
struct UdModel {
  int model;
  const double* q;   // intrinsics row (8)
  double radial(double r2) const {   // Camera_Pinhole_Radial.hpp:273-277 (K1), :482-486 (K3): distoFunctor
    if (model == MVGX_CAM_PINHOLE_RADIAL1) { const double c = 1. + r2 * q[3]; return r2 * (c * c); }
    const double c = 1. + r2 * (q[3] + r2 * (q[4] + r2 * q[5]));
    return r2 * (c * c);
  }
  void remove_disto(double p[2]) const {
    switch (model) {
      case MVGX_CAM_PINHOLE_RADIAL1:
      case MVGX_CAM_PINHOLE_RADIAL3: {   // Camera_Pinhole_Radial.hpp:148-158 / :357-367 + bisection :37-70
        const double r2 = p[0] * p[0] + p[1] * p[1];
        if (r2 == 0) return;
        double lowerbound = r2, upbound = r2;
        while (radial(lowerbound) > r2) lowerbound /= 1.05;
        while (radial(upbound) < r2) upbound *= 1.05;
        while (1e-10 < upbound - lowerbound) {
          const double mid = .5 * (lowerbound + upbound);
          if (radial(mid) > r2) upbound = mid; else lowerbound = mid;
        }
        const double radius = std::sqrt(.5 * (lowerbound + upbound) / r2);
        p[0] *= radius; p[1] *= radius;
        return;
      }
      case MVGX_CAM_PINHOLE_BROWN: {     // Camera_Pinhole_Brown.hpp:97-110, distoFunction :226-236
        auto disto = [&](const double u[2], double d[2]) {
          const double r2 = u[0] * u[0] + u[1] * u[1], r4 = r2 * r2, r6 = r4 * r2;
          const double k_diff = q[3] * r2 + q[4] * r4 + q[5] * r6;
          const double t_x = q[7] * (r2 + 2 * u[0] * u[0]) + 2 * q[6] * u[0] * u[1];
          const double t_y = q[6] * (r2 + 2 * u[1] * u[1]) + 2 * q[7] * u[0] * u[1];
          d[0] = u[0] * k_diff + t_x; d[1] = u[1] * k_diff + t_y;
        };
        double u[2] = {p[0], p[1]}, d[2];
        disto(u, d);
        int guard = 0;
        while (std::fabs(u[0] + d[0] - p[0]) + std::fabs(u[1] + d[1] - p[1]) > 1e-10 && guard++ < 10000) {
          u[0] = p[0] - d[0]; u[1] = p[1] - d[1];
          disto(u, d);
        }
        p[0] = u[0]; p[1] = u[1];
        return;
      }
      case MVGX_CAM_PINHOLE_FISHEYE: {   // Camera_Pinhole_Fisheye.hpp:112-136
        const double theta_dist = std::hypot(p[0], p[1]);
        if (theta_dist > 1e-8) {
          double theta = theta_dist;
          for (int j = 0; j < 10; ++j) {
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            theta = theta_dist / (1 + q[3] * t2 + q[4] * t4 + q[5] * t6 + q[6] * t8);
          }
          const double scale = std::tan(theta) / theta_dist;
          p[0] *= scale; p[1] *= scale;
        }
        return;
      }
      default: return;   // pinhole: Camera_Pinhole.hpp:268-271
    }
  }
  void bearing(const double x[2], double b[3]) const {
    if (model == MVGX_CAM_SPHERICAL) {   // Camera_Spherical.hpp:103-132
      const double size = std::max(q[0], q[1]);
      const double ux = (x[0] - q[0] / 2.0) / size, uy = (x[1] - q[1] / 2.0) / size;
      const double lon = ux * 2 * M_PI, lat = -uy * 2 * M_PI;
      b[0] = std::cos(lat) * std::sin(lon); b[1] = -std::sin(lat); b[2] = std::cos(lat) * std::cos(lon);
      return;
    }
    double p[2] = {(x[0] - q[1]) / q[0], (x[1] - q[2]) / q[0]};        // ima2cam
    remove_disto(p);
    const double xu[2] = {q[0] * p[0] + q[1], q[0] * p[1] + q[2]};     // cam2ima: get_ud_pixel
    b[0] = (xu[0] - q[1]) / q[0]; b[1] = (xu[1] - q[2]) / q[0]; b[2] = 1.0;   // Kinv * homogeneous (Camera_Pinhole.hpp:136-139)
    const double n = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    b[0] /= n; b[1] /= n; b[2] /= n;
  }
};
}  // namespace

int oracle_ba_track_angles(uint32_t n_points, uint64_t n_obs, const double* poses, const double* intrinsics,
                           const int32_t* intr_model, const uint32_t* obs_pose, const uint32_t* obs_intr,
                           const uint32_t* obs_point, const double* obs_xy, double* out) {
  std::vector<double> rays(3 * n_obs);
  std::vector<std::vector<uint64_t>> track(n_points);
  for (uint64_t o = 0; o < n_obs; ++o) {
    const UdModel m{intr_model[obs_intr[o]], intrinsics + 8 * size_t(obs_intr[o])};
    if (intr_param_count(m.model) < 0) return 1;
    double b[3], w[3];
    m.bearing(obs_xy + 2 * o, b);
    const double* aa = poses + 6 * size_t(obs_pose[o]);
    const double inv[3] = {-aa[0], -aa[1], -aa[2]};   // R^T = R(-aa)
    angle_axis_rotate_point<double>(inv, b, w);
    const double n = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int k = 0; k < 3; ++k) rays[3 * o + k] = w[k] / n;
    track[obs_point[o]].push_back(o);
  }
  for (uint32_t p = 0; p < n_points; ++p) {
    double max_angle = 0.0;
    const auto& t = track[p];
    for (size_t a = 0; a < t.size(); ++a)
      for (size_t b = a + 1; b < t.size(); ++b) {
        const double* r1 = &rays[3 * t[a]];
        const double* r2 = &rays[3 * t[b]];
        const double dt = r1[0] * r2[0] + r1[1] * r2[1] + r1[2] * r2[2];
        const double angle = std::acos(std::max(-1.0 + 1.e-8, std::min(dt, 1.0 - 1.e-8))) / M_PI * 180.0;
        max_angle = std::max(angle, max_angle);
      }
    out[p] = max_angle;
  }
  return 0;
}

// cost = 1/2 sum rho(|r|^2) (Ceres cost), rmse = sqrt(sum |r|^2 / (2 n_obs)) (sfm_data_BA_test.cpp:310-330)
int oracle_ba_evaluate(const mvgx_ba_problem* prob, double* cost, double* rmse) {
  Problem P;
  if (!load(prob, P)) return 1;
  double c, se;
  evaluate_cost(P, P.poses, P.intr, P.points, &c, &se);
  *cost = c;
  *rmse = P.n_obs_rmse ? std::sqrt(se / (2.0 * double(P.n_obs_rmse))) : 0.0;
  return 0;
}

// TrustRegionMinimizer::Minimize (trust_region_minimizer.cc:66-119). trace (optional): per iteration
// {cost, candidate_cost, model_cost_change, radius_after, accepted(1/0), gradient_max_norm}, trace_cap rows.
int oracle_ba_solve(const mvgx_ba_problem* prob, const mvgx_ba_options* options, double* poses_out, double* intr_out,
                    double* points_out, mvgx_ba_summary* sum, double* trace, int trace_cap) {
  Problem P;
  if (!load(prob, P)) return 1;
  Solver S(P, *options);
  S.poses = P.poses; S.intr = P.intr; S.points = P.points;
  S.radius = options->initial_radius;
  std::memset(sum, 0, sizeof(*sum));
  double c0, se0;
  evaluate_cost(P, S.poses, S.intr, S.points, &c0, &se0);
  sum->initial_rmse = P.n_obs_rmse ? std::sqrt(se0 / (2.0 * double(P.n_obs_rmse))) : 0.0;
  S.evaluate_gradient_and_jacobian(true);  // IterationZero (:177-212)
  sum->initial_cost = S.x_cost;
  sum->termination = 1;  // NO_CONVERGENCE until proven otherwise
  int iteration = 0, invalid = 0, ntrace = 0;
  bool last_successful = true;
  bool x_norm_valid = false;  // Init() leaves x_norm_ = -1 until the first accepted step (trust_region_minimizer.cc:169)
  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue (:288-340)
    if (last_successful) ++sum->num_successful_steps;
    if (iteration >= options->max_num_iterations) { sum->termination = 1; break; }
    if (last_successful && S.gradient_max_norm <= options->gradient_tolerance) { sum->termination = 0; break; }
    if (S.radius <= options->min_radius) { sum->termination = 0; break; }
    ++iteration;
    const bool ok = S.compute_step();
    const bool valid = ok && S.model_cost_change > 0.0;
    if (!valid) {  // HandleInvalidStep (:432-462) + LevenbergMarquardtStrategy::StepIsInvalid = StepRejected(0)
      if (++invalid >= options->max_consecutive_invalid_steps) { sum->termination = 2; break; }
      S.radius = S.radius / S.decrease_factor; S.decrease_factor *= 2.0; S.reuse_diagonal = true;
      last_successful = false;
      continue;
    }
    invalid = 0;
    double step_norm, x_norm;
    S.make_candidate(&step_norm, &x_norm);
    if (!x_norm_valid) x_norm = -1.0;
    double se;
    evaluate_cost(P, S.cposes, S.cintr, S.cpoints, &S.candidate_cost, &se);
    if (trace && ntrace < trace_cap) {
      double* t = trace + size_t(ntrace) * 6;
      t[0] = S.x_cost; t[1] = S.candidate_cost; t[2] = S.model_cost_change; t[3] = S.radius; t[4] = -1; t[5] = S.gradient_max_norm;
    }
    // ParameterToleranceReached (:667-687), FunctionToleranceReached (:690-708): both return BEFORE accepting
    if (step_norm <= options->parameter_tolerance * (x_norm + options->parameter_tolerance)) { sum->termination = 0; ++ntrace; break; }
    if (std::fabs(S.x_cost - S.candidate_cost) <= options->function_tolerance * S.x_cost) { sum->termination = 0; ++ntrace; break; }
    const double relative_decrease = (S.x_cost - S.candidate_cost) / S.model_cost_change;  // StepQuality, monotonic steps
    if (relative_decrease > options->min_relative_decrease) {  // HandleSuccessfulStep (:767-779)
      S.poses = S.cposes; S.intr = S.cintr; S.points = S.cpoints;
      S.evaluate_gradient_and_jacobian(false);
      S.radius = S.radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));  // StepAccepted (:147-154)
      S.radius = std::min(options->max_radius, S.radius);
      S.decrease_factor = 2.0; S.reuse_diagonal = false;
      last_successful = true; x_norm_valid = true;
      if (trace && ntrace < trace_cap) trace[size_t(ntrace) * 6 + 4] = 1;
    } else {  // HandleUnsuccessfulStep + StepRejected (:156-160)
      S.radius = S.radius / S.decrease_factor; S.decrease_factor *= 2.0; S.reuse_diagonal = true;
      last_successful = false;
      if (trace && ntrace < trace_cap) trace[size_t(ntrace) * 6 + 4] = 0;
    }
    if (trace && ntrace < trace_cap) trace[size_t(ntrace) * 6 + 3] = S.radius;
    ++ntrace;
  }
  sum->num_iterations = iteration;
  sum->final_cost = S.x_cost;
  double c1, se1;
  evaluate_cost(P, S.poses, S.intr, S.points, &c1, &se1);
  sum->final_rmse = P.n_obs_rmse ? std::sqrt(se1 / (2.0 * double(P.n_obs_rmse))) : 0.0;
  if (poses_out) std::memcpy(poses_out, S.poses.data(), S.poses.size() * sizeof(double));
  if (intr_out) std::memcpy(intr_out, S.intr.data(), S.intr.size() * sizeof(double));
  if (points_out) std::memcpy(points_out, S.points.data(), S.points.size() * sizeof(double));
  return sum->termination == 2 ? 2 : 0;
}

}  // extern "C"
