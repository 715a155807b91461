// ref_shim_ba.cpp — TEST INFRASTRUCTURE ONLY.
//
// extern "C" wrapper (our code) around the REFERENCE's bundle adjustment: openMVG::sfm::Bundle_Adjustment_Ceres::Adjust
// (sfm/sfm_data_BA_ceres.cpp:165-608) on vendored Ceres 1.13.0, both compiled in place from /root/reference/src by
// oracle/Makefile into oracle/_ref/libref_ba.so. No reference source is copied: this file builds an in-memory SfM_Data
// from flat arrays (the layout of mvgx_ba_problem in include/mvgx.h), calls the reference, and flattens the result.
// The RMSE helper restates sfm_data_BA_test.cpp:310-330 using the reference's own IntrinsicBase::residual.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

#include "ceres/rotation.h"
#include "ceres/types.h"

#include "openMVG/cameras/Camera_Common.hpp"
#include "openMVG/cameras/Camera_Pinhole.hpp"
#include "openMVG/cameras/Camera_Pinhole_Radial.hpp"
#include "openMVG/geometry/pose3.hpp"
#include "openMVG/sfm/sfm_data.hpp"
#include "openMVG/sfm/sfm_data_BA.hpp"
#include "openMVG/sfm/sfm_data_BA_ceres.hpp"

using namespace openMVG;
using namespace openMVG::sfm;
using namespace openMVG::cameras;
using namespace openMVG::geometry;

namespace {

double rmse_of(const SfM_Data& sfm_data) {
  double ss = 0.0;
  size_t n = 0;
  for (const auto& lm : sfm_data.GetLandmarks()) {
    for (const auto& ob : lm.second.obs) {
      const View* view = sfm_data.GetViews().find(ob.first)->second.get();
      const Pose3 pose = sfm_data.GetPoseOrDie(view);
      const std::shared_ptr<IntrinsicBase> intr = sfm_data.GetIntrinsics().find(view->id_intrinsic)->second;
      const Vec2 r = intr->residual(pose(lm.second.X), ob.second.x);
      ss += r(0) * r(0) + r(1) * r(1);
      n += 2;
    }
  }
  return n ? std::sqrt(ss / double(n)) : 0.0;
}

}  // namespace

extern "C" {

// poses: n_poses x 6 (angle-axis, t = -R C), intrinsics: n_intr x 8, points: n_points x 3 — all updated in place.
// Each pose becomes one View (view id = pose id) using the intrinsic of its first observation.
// intrinsics_opt / extrinsics_opt / structure_opt: numeric values of the openMVG option enums.
// linear_solver: 0 = reference default (SPARSE_SCHUR + EIGEN_SPARSE here), 1 = DENSE_SCHUR, 2 = SPARSE_SCHUR.
// out_stats[0..3] = {rmse_before, rmse_after, seconds in Adjust(), Adjust() return value}.
int ref_ba_adjust(uint32_t n_poses, uint32_t n_intr, uint32_t n_points, uint64_t n_obs, double* poses,
                  double* intrinsics, const int32_t* intr_model, double* points, const uint32_t* obs_pose,
                  const uint32_t* obs_intr, const uint32_t* obs_point, const double* obs_xy, int intrinsics_opt,
                  int extrinsics_opt, int structure_opt, int max_iterations, int num_threads, int linear_solver,
                  int use_loss, int print_summary, double* out_stats) {
  SfM_Data scene;
  std::vector<int64_t> pose_intr(n_poses, -1);
  for (uint64_t k = 0; k < n_obs; ++k)
    if (pose_intr[obs_pose[k]] < 0) pose_intr[obs_pose[k]] = obs_intr[k];
  for (uint64_t k = 0; k < n_obs; ++k)
    if (pose_intr[obs_pose[k]] != int64_t(obs_intr[k])) return -2;  // a pose seen through two intrinsics: not a View

  for (uint32_t i = 0; i < n_intr; ++i) {
    const double* p = intrinsics + size_t(i) * 8;
    if (intr_model[i] == PINHOLE_CAMERA)
      scene.intrinsics[i] = std::make_shared<Pinhole_Intrinsic>(1000, 1000, p[0], p[1], p[2]);
    else if (intr_model[i] == PINHOLE_CAMERA_RADIAL1)
      scene.intrinsics[i] = std::make_shared<Pinhole_Intrinsic_Radial_K1>(1000, 1000, p[0], p[1], p[2], p[3]);
    else if (intr_model[i] == PINHOLE_CAMERA_RADIAL3)
      scene.intrinsics[i] = std::make_shared<Pinhole_Intrinsic_Radial_K3>(1000, 1000, p[0], p[1], p[2], p[3], p[4], p[5]);
    else
      return -3;
  }
  for (uint32_t i = 0; i < n_poses; ++i) {
    const double* p = poses + size_t(i) * 6;
    Mat3 R;
    ceres::AngleAxisToRotationMatrix(p, R.data());  // column-major, as the reference's own write-back (:534-553)
    const Vec3 t(p[3], p[4], p[5]);
    const Vec3 C = -R.transpose() * t;
    scene.poses[i] = Pose3(R, C);
    const IndexT intr_id = pose_intr[i] < 0 ? 0 : IndexT(pose_intr[i]);
    scene.views[i] = std::make_shared<View>("", i, intr_id, i, 1000, 1000);
  }
  for (uint32_t j = 0; j < n_points; ++j)
    scene.structure[j].X = Vec3(points[3 * size_t(j)], points[3 * size_t(j) + 1], points[3 * size_t(j) + 2]);
  for (uint64_t k = 0; k < n_obs; ++k)
    scene.structure[obs_point[k]].obs[obs_pose[k]] = Observation(Vec2(obs_xy[2 * k], obs_xy[2 * k + 1]), IndexT(k));

  out_stats[0] = rmse_of(scene);

  Bundle_Adjustment_Ceres::BA_Ceres_options opt(false, num_threads != 1);
  if (num_threads > 0) opt.nb_threads_ = unsigned(num_threads);
  opt.bCeres_summary_ = print_summary != 0;
  opt.bUse_loss_function_ = use_loss != 0;
  if (max_iterations > 0) opt.max_num_iterations_ = max_iterations;
  if (linear_solver == 1) opt.linear_solver_type_ = ceres::DENSE_SCHUR;
  if (linear_solver == 2) opt.linear_solver_type_ = ceres::SPARSE_SCHUR;
  Bundle_Adjustment_Ceres ba(opt);
  const Optimize_Options oo(static_cast<Intrinsic_Parameter_Type>(intrinsics_opt),
                            static_cast<Extrinsic_Parameter_Type>(extrinsics_opt),
                            static_cast<Structure_Parameter_Type>(structure_opt != 0));
  const auto t0 = std::chrono::steady_clock::now();
  const bool ok = ba.Adjust(scene, oo);
  const auto t1 = std::chrono::steady_clock::now();
  out_stats[2] = std::chrono::duration<double>(t1 - t0).count();
  out_stats[3] = ok ? 1.0 : 0.0;
  out_stats[1] = rmse_of(scene);

  for (uint32_t i = 0; i < n_poses; ++i) {
    const Pose3& pose = scene.poses.at(i);
    const Mat3 R = pose.rotation();
    const Vec3 t = pose.translation();
    double* p = poses + size_t(i) * 6;
    ceres::RotationMatrixToAngleAxis((const double*)R.data(), p);
    p[3] = t(0); p[4] = t(1); p[5] = t(2);
  }
  for (uint32_t i = 0; i < n_intr; ++i) {
    const std::vector<double> prm = scene.intrinsics.at(i)->getParams();
    for (size_t k = 0; k < prm.size() && k < 8; ++k) intrinsics[size_t(i) * 8 + k] = prm[k];
  }
  for (uint32_t j = 0; j < n_points; ++j) {
    const Vec3& X = scene.structure.at(j).X;
    points[3 * size_t(j)] = X(0); points[3 * size_t(j) + 1] = X(1); points[3 * size_t(j) + 2] = X(2);
  }
  return ok ? 0 : 1;
}

int ref_ba_default_linear_solver_is_sparse(void) {
  Bundle_Adjustment_Ceres::BA_Ceres_options opt(false, true);
  return opt.linear_solver_type_ == ceres::SPARSE_SCHUR ? 1 : 0;
}

}  // extern "C"
