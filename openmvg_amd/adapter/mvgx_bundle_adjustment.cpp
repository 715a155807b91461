// mvgx_bundle_adjustment.cpp — Bundle_Adjustment_HIP::Adjust: SfM_Data -> flat mvgx_ba_problem -> libmvgx_hip.so
// -> SfM_Data, following the parameter-block layout and write-back rules of Bundle_Adjustment_Ceres::Adjust
// (sfm/sfm_data_BA_ceres.cpp): pose block = [angle-axis(3), t(3)] with t = -R*C (:260-271); intrinsic block =
// getParams() (:310-320); one residual per observation with blocks (intrinsic, pose, landmark) (:354-396); Huber
// loss with a = Square(4.0) (:249); constant blocks / subsets from Optimize_Options (:274-306, :321-344, :394-395);
// write-back :527-568. No Ceres and no Eigen solver is used: Eigen only as the reference's value types, and
// ceres/rotation.h (header-only templates vendored with openMVG) for the exact angle-axis conversions the reference
// performs on the host.
#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "ceres/rotation.h"

#include "openMVG/cameras/Camera_Common.hpp"
#include "openMVG/cameras/Camera_Intrinsics.hpp"
#include "openMVG/geometry/pose3.hpp"
#include "openMVG/sfm/sfm_data.hpp"
#include "openMVG/sfm/sfm_view_priors.hpp"
#include "openMVG/system/logger.hpp"
#include "openMVG/types.hpp"

#include "mvgx.h"
#include "mvgx_bundle_adjustment.hpp"

namespace openMVG {
namespace sfm {

using cameras::Intrinsic_Parameter_Type;
using geometry::Pose3;

bool Bundle_Adjustment_HIP::Adjust(SfM_Data& sfm_data, const Optimize_Options& options) {
  // --- parts of Adjust() the device path does not implement yet are refused loudly, never approximated ---
  if (options.control_point_opt.bUse_control_points && !sfm_data.control_points.empty()) {
    OPENMVG_LOG_ERROR << "mvgx BA: ground-control-point residuals (sfm_data_BA_ceres.cpp:398-451) are not on the device path.";
    return false;
  }
  if (options.use_motion_priors_opt) {
    for (const auto& v : sfm_data.GetViews()) {
      const auto* prior = dynamic_cast<const ViewPriors*>(v.second.get());
      if (prior && prior->b_use_pose_center_ && sfm_data.IsPoseAndIntrinsicDefined(prior)) {
        OPENMVG_LOG_ERROR << "mvgx BA: pose-centre priors (sfm_data_BA_ceres.cpp:184-240,454-473) are not on the device path.";
        return false;
      }
    }
  }

  // --- parameter blocks ---
  std::unordered_map<IndexT, uint32_t> pose_idx, intr_idx;
  std::vector<IndexT> pose_ids, intr_ids;
  std::vector<double> poses, intrinsics, points, obs_xy;
  std::vector<int32_t> intr_model;
  std::vector<uint8_t> pose_mask, intr_mask;
  std::vector<uint32_t> obs_pose, obs_intr, obs_point;

  uint8_t pmask = 0;
  if (options.extrinsics_opt == Extrinsic_Parameter_Type::NONE) pmask = 0x3F;
  else if (options.extrinsics_opt == Extrinsic_Parameter_Type::ADJUST_TRANSLATION) pmask = 0x07;  // rotation constant
  else if (options.extrinsics_opt == Extrinsic_Parameter_Type::ADJUST_ROTATION) pmask = 0x38;     // translation constant

  poses.reserve(sfm_data.poses.size() * 6);
  for (const auto& it : sfm_data.poses) {
    const Mat3 R = it.second.rotation();
    const Vec3 t = it.second.translation();
    double aa[3];
    ceres::RotationMatrixToAngleAxis(static_cast<const double*>(R.data()), aa);
    pose_idx.emplace(it.first, static_cast<uint32_t>(pose_ids.size()));
    pose_ids.push_back(it.first);
    poses.insert(poses.end(), {aa[0], aa[1], aa[2], t(0), t(1), t(2)});
    pose_mask.push_back(pmask);
  }
  for (const auto& it : sfm_data.intrinsics) {
    if (!cameras::isValid(it.second->getType())) {
      OPENMVG_LOG_ERROR << "Unsupported camera type.";
      continue;
    }
    const std::vector<double> prm = it.second->getParams();
    if (prm.empty() || prm.size() > MVGX_BA_MAX_INTR_PARAMS) {
      OPENMVG_LOG_ERROR << "mvgx BA: camera model " << static_cast<int>(it.second->getType()) << " has no device functor.";
      return false;
    }
    uint8_t m = 0;
    if (options.intrinsics_opt == Intrinsic_Parameter_Type::NONE) {
      m = static_cast<uint8_t>((1u << prm.size()) - 1u);
    } else {
      for (int c : it.second->subsetParameterization(options.intrinsics_opt)) m |= static_cast<uint8_t>(1u << c);
    }
    intr_idx.emplace(it.first, static_cast<uint32_t>(intr_ids.size()));
    intr_ids.push_back(it.first);
    intr_model.push_back(static_cast<int32_t>(it.second->getType()));
    for (size_t k = 0; k < MVGX_BA_MAX_INTR_PARAMS; ++k) intrinsics.push_back(k < prm.size() ? prm[k] : 0.0);
    intr_mask.push_back(m);
  }

  // --- observations (landmark X is refined in place, as the reference hands X.data() to the solver) ---
  std::vector<Landmark*> lm_of_point;
  lm_of_point.reserve(sfm_data.structure.size());
  points.reserve(sfm_data.structure.size() * 3);
  for (auto& lm : sfm_data.structure) {
    const uint32_t j = static_cast<uint32_t>(lm_of_point.size());
    lm_of_point.push_back(&lm.second);
    points.insert(points.end(), {lm.second.X(0), lm.second.X(1), lm.second.X(2)});
    for (const auto& ob : lm.second.obs) {
      const View* view = sfm_data.views.at(ob.first).get();
      const auto ii = intr_idx.find(view->id_intrinsic);
      if (ii == intr_idx.end()) {
        OPENMVG_LOG_ERROR << "Cannot create a CostFunction for this camera model.";
        return false;
      }
      obs_pose.push_back(pose_idx.at(view->id_pose));
      obs_intr.push_back(ii->second);
      obs_point.push_back(j);
      obs_xy.push_back(ob.second.x(0));
      obs_xy.push_back(ob.second.x(1));
    }
  }

  mvgx_ba_problem prob{};
  prob.n_poses = static_cast<uint32_t>(pose_ids.size());
  prob.n_intrinsics = static_cast<uint32_t>(intr_ids.size());
  prob.n_points = static_cast<uint32_t>(lm_of_point.size());
  prob.n_obs = obs_pose.size();
  prob.poses = poses.data(); prob.intrinsics = intrinsics.data(); prob.intr_model = intr_model.data();
  prob.points = points.data();
  prob.obs_pose = obs_pose.data(); prob.obs_intr = obs_intr.data(); prob.obs_point = obs_point.data();
  prob.obs_xy = obs_xy.data();
  prob.pose_const_mask = pose_mask.data(); prob.intr_const_mask = intr_mask.data();
  prob.points_constant = options.structure_opt == Structure_Parameter_Type::NONE ? 1 : 0;
  prob.huber_a = options_.bUse_loss_function_ ? Square(4.0) : 0.0;

  mvgx_ba_ctx* ctx = nullptr;
  int rc = mvgx_ba_create(options_.device_, &prob, &ctx);
  if (rc == MVGX_ERR_UNSUPPORTED) {
    OPENMVG_LOG_ERROR << "Cannot create a CostFunction for this camera model. (" << mvgx_last_error() << ")";
    return false;
  }
  if (rc != MVGX_OK) {
    OPENMVG_LOG_ERROR << "mvgx BA: " << mvgx_last_error();
    return false;
  }
  mvgx_ba_options opt;
  mvgx_ba_default_options(&opt);
  opt.max_num_iterations = options_.max_num_iterations_;
  opt.parameter_tolerance = options_.parameter_tolerance_;
  opt.gradient_tolerance = options_.gradient_tolerance_;
  mvgx_ba_summary summary{};
  rc = mvgx_ba_solve(ctx, &opt, &summary);
  // the solver state is written back to the landmarks in every case: the reference optimises X in place, so a failed
  // solve leaves moved points behind as well (sfm_data_BA_ceres.cpp:378, :503-507)
  const int rc_read = mvgx_ba_read_params(ctx, poses.data(), intrinsics.data(), points.data());
  mvgx_ba_destroy(ctx);
  if (rc_read != MVGX_OK) {
    OPENMVG_LOG_ERROR << "mvgx BA: " << mvgx_last_error();
    return false;
  }
  if (!prob.points_constant)
    for (size_t j = 0; j < lm_of_point.size(); ++j)
      lm_of_point[j]->X = Vec3(points[3 * j], points[3 * j + 1], points[3 * j + 2]);
  if (rc != MVGX_OK) {
    OPENMVG_LOG_ERROR << "IsSolutionUsable is false. Bundle Adjustment failed. (" << mvgx_last_error() << ")";
    return false;
  }

  if (options_.bVerbose_) {
    const double nres = 2.0 * static_cast<double>(prob.n_obs);
    OPENMVG_LOG_INFO << "\nBundle Adjustment statistics (approximated RMSE):\n"
                     << " #views: " << sfm_data.views.size() << "\n"
                     << " #poses: " << sfm_data.poses.size() << "\n"
                     << " #intrinsics: " << sfm_data.intrinsics.size() << "\n"
                     << " #tracks: " << sfm_data.structure.size() << "\n"
                     << " #residuals: " << static_cast<uint64_t>(nres) << "\n"
                     << " Initial RMSE: " << std::sqrt(summary.initial_cost / nres) << "\n"
                     << " Final RMSE: " << std::sqrt(summary.final_cost / nres) << "\n"
                     << " Time (s): " << summary.total_ms * 1e-3 << " (MI355X, " << summary.num_iterations
                     << " LM iterations)\n--\n Used motion prior: 0";
  }

  if (options.extrinsics_opt != Extrinsic_Parameter_Type::NONE) {
    for (size_t i = 0; i < pose_ids.size(); ++i) {
      const double* p = &poses[6 * i];
      Mat3 R_refined;
      ceres::AngleAxisToRotationMatrix(p, R_refined.data());
      const Vec3 t_refined(p[3], p[4], p[5]);
      Pose3& pose = sfm_data.poses.at(pose_ids[i]);
      if (options.extrinsics_opt == Extrinsic_Parameter_Type::ADJUST_ROTATION)
        pose.rotation() = R_refined;
      else if (options.extrinsics_opt == Extrinsic_Parameter_Type::ADJUST_TRANSLATION)
        pose.center() = -R_refined.transpose() * t_refined;
      else
        pose = Pose3(R_refined, -R_refined.transpose() * t_refined);
    }
  }
  if (options.intrinsics_opt != Intrinsic_Parameter_Type::NONE) {
    for (size_t k = 0; k < intr_ids.size(); ++k) {
      auto& cam = sfm_data.intrinsics.at(intr_ids[k]);
      const size_t np = cam->getParams().size();
      cam->updateFromParams(std::vector<double>(&intrinsics[8 * k], &intrinsics[8 * k] + np));
    }
  }
  return true;
}

}  // namespace sfm
}  // namespace openMVG
