// mvgx_outlier_filters.cpp - RemoveOutliers_PixelResidualError / RemoveOutliers_AngleError (sfm/sfm_data_filters.cpp:40-73, :77-121)
// with the per-observation residual norms and the per-track maximum ray angles computed on the MI355X (mvgx_ba_residuals,
// mvgx_ba_track_angles): the two passes of SequentialSfMReconstructionEngine::badTrackRejector (sequential_SfM.cpp:1226-1232) that
// sit between the consecutive Adjust() calls of the "do { BA } while (reject)" loops (:206-210). In the reference they are one
// host thread making three hash lookups and two virtual calls per observation (and per observation PAIR for the angles); after a
// device Adjust() of a few milliseconds they were what the loop waited for.
//
// The scene is flattened exactly as Adjust() does (mvgx_scene_arrays.hpp) and offered to the BA context that Adjust() left idle:
// the filters run right after it, on the structure it solved, so the context is re-bound (mvgx_ba_update, values only) and nothing
// is built; any other scene gets a new context, which then waits for the next Adjust().
//
// Decisions are the reference's: the device value decides unless it lies within 1e-9 (relative) of the threshold - then that
// observation / track is re-evaluated with the reference's own expression (IntrinsicBase::residual, AngleBetweenRay over
// get_ud_pixel) on the host. Erasure order and return values as in the reference.
//
// Link-time substitution: sfm/sfm_data_filters.cpp holds seven more functions that stay as they are, so that TU is
// compiled with -DRemoveOutliers_PixelResidualError=RemoveOutliers_PixelResidualError_cpu
// -DRemoveOutliers_AngleError=RemoveOutliers_AngleError_cpu (no source change) and this TU defines the two original names. The _cpu
// functions - the reference's own code, in the application anyway - finish the call when the device path cannot
// (mvgx_adapter_policy.hpp: logged once; MVGX_ON_DEVICE_ERROR=throw throws instead).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <unordered_map>
#include <vector>

#include "ceres/rotation.h"

#include "openMVG/cameras/Camera_Common.hpp"
#include "openMVG/cameras/Camera_Intrinsics.hpp"
#include "openMVG/geometry/pose3.hpp"
#include "openMVG/sfm/sfm_data.hpp"
#include "openMVG/sfm/sfm_data_filters.hpp"
#include "openMVG/sfm/sfm_landmark.hpp"
#include "openMVG/system/logger.hpp"
#include "openMVG/types.hpp"

#include "mvgx.h"
#include "mvgx_adapter_policy.hpp"
#include "mvgx_scene_arrays.hpp"

namespace openMVG {
namespace sfm {

// the reference's implementations under the names the build recipe gives them (see above)
IndexT RemoveOutliers_PixelResidualError_cpu(SfM_Data& sfm_data, const double dThresholdPixel, const unsigned int minTrackLength);
IndexT RemoveOutliers_AngleError_cpu(SfM_Data& sfm_data, const double dMinAcceptedAngle);

namespace {

using mvgx_adapter::FlatScene;

// Flat arrays of the scene (every block free: the filters evaluate, they do not solve) bound to a device context.
// false: the scene is not one the device path takes (a camera model without functor, a landmark observing a view without pose or
// intrinsic - the reference's .at() / GetPoseOrDie would throw or abort there, and does when the caller falls back to it) or a
// device call failed (logged by the policy).
bool bind_scene(SfM_Data& sfm_data, FlatScene& fs, mvgx_adapter::BoundContext& bound, const char* stage) {
  std::unordered_map<IndexT, uint32_t> pose_idx, intr_idx;
  fs.poses.clear(); fs.intrinsics.clear(); fs.intr_model.clear(); fs.pose_mask.clear(); fs.intr_mask.clear();
  fs.pose_ids.clear(); fs.intr_ids.clear();
  for (const auto& it : sfm_data.poses) {
    const Mat3 R = it.second.rotation();
    const Vec3 t = it.second.translation();
    double aa[3];
    ceres::RotationMatrixToAngleAxis(static_cast<const double*>(R.data()), aa);
    pose_idx.emplace(it.first, static_cast<uint32_t>(fs.pose_ids.size()));
    fs.pose_ids.push_back(it.first);
    fs.poses.insert(fs.poses.end(), {aa[0], aa[1], aa[2], t(0), t(1), t(2)});
  }
  for (const auto& it : sfm_data.intrinsics) {
    if (!cameras::isValid(it.second->getType())) continue;
    std::vector<double> prm = it.second->getParams();
    if (prm.size() > MVGX_BA_MAX_INTR_PARAMS) return false;
    if (prm.empty()) prm = {static_cast<double>(it.second->w()), static_cast<double>(it.second->h())};   // CAMERA_SPHERICAL: data of the functor
    intr_idx.emplace(it.first, static_cast<uint32_t>(fs.intr_ids.size()));
    fs.intr_ids.push_back(it.first);
    fs.intr_model.push_back(static_cast<int32_t>(it.second->getType()));
    for (size_t k = 0; k < MVGX_BA_MAX_INTR_PARAMS; ++k) fs.intrinsics.push_back(k < prm.size() ? prm[k] : 0.0);
  }
  int flatten_error = 0;
  const uint64_t n_obs = mvgx_adapter::flatten_observations(sfm_data, pose_idx, intr_idx, fs, &flatten_error, [](const char*) {});
  if (flatten_error) return false;
  mvgx_ba_problem prob{};
  prob.n_poses = static_cast<uint32_t>(pose_idx.size());
  prob.n_intrinsics = static_cast<uint32_t>(intr_idx.size());
  prob.n_points = static_cast<uint32_t>(fs.lm_of_point.size());
  prob.n_obs = n_obs;
  prob.poses = fs.poses.data(); prob.intrinsics = fs.intrinsics.data(); prob.intr_model = fs.intr_model.data();
  prob.points = fs.points.data();
  prob.obs_pose = fs.obs_pose.data(); prob.obs_intr = fs.obs_intr.data(); prob.obs_point = fs.obs_point.data();
  prob.obs_xy = fs.obs_xy.data();
  prob.huber_a = Square(4.0);   // (as Adjust() with its default loss: the kept context's structure does not depend on it)
  const bool inj = mvgx_adapter::injected("filters", stage);
  const int rc = inj ? MVGX_ERR_NODEV : mvgx_adapter::bind_context(mvgx_adapter::kAnyDevice, prob, fs, /* plain */ true, bound);
  if (rc != MVGX_OK) {
    if (rc != MVGX_ERR_UNSUPPORTED) mvgx_adapter::device_failure(mvgx_adapter::kFilters, "outlier filters", "mvgx_ba_create", rc, inj);
    return false;
  }
  return true;
}

bool near_threshold(double v, double thr) { return std::fabs(v - thr) <= 1e-9 * std::max(1.0, std::fabs(thr)); }

}  // namespace

IndexT RemoveOutliers_PixelResidualError(SfM_Data& sfm_data, const double dThresholdPixel, const unsigned int minTrackLength) {
  FlatScene& fs = mvgx_adapter::flat_scene();
  mvgx_adapter::BoundContext bound;
  if (!bind_scene(sfm_data, fs, bound, "residuals")) return RemoveOutliers_PixelResidualError_cpu(sfm_data, dThresholdPixel, minTrackLength);
  std::vector<double>& norm = fs.scratch;   // in the indexing of the context's structure (the kept one on the subset route)
  norm.resize(std::max<size_t>(bound.subset ? bound.kept->obs_pose.size() : fs.obs_pose.size(), 1));
  const int rc = mvgx_ba_residuals(bound.ctx, norm.data());
  // (the context goes back into the slot when this function is done with the flat arrays: the hand-over takes them along)
  struct Release {
    mvgx_adapter::BoundContext& b; FlatScene& fs; bool healthy;
    ~Release() { mvgx_adapter::release_bound_context(b, fs, true, healthy); }
  } release{bound, fs, rc == MVGX_OK};
  const std::vector<Landmark*>& lm_of_point = fs.lm_of_point;
  const std::vector<IndexT>& lm_key = fs.lm_key;
  const std::vector<uint64_t>&obs_first = fs.obs_first, &obs_old = bound.obs_old;
  const bool subset = bound.subset;
  if (rc != MVGX_OK) {
    mvgx_adapter::device_failure(mvgx_adapter::kFilters, "outlier filters", "mvgx_ba_residuals", rc, false);
    return RemoveOutliers_PixelResidualError_cpu(sfm_data, dThresholdPixel, minTrackLength);
  }
  mvgx_adapter::counters().device_pairs.fetch_add(1);
  // erasure: the observations of a landmark are visited in the order of the walk that numbered them (an unordered_map keeps the
  // order of the elements it keeps); landmarks are independent, so the host workers take ranges of them; the landmarks themselves
  // leave the (shared) structure map afterwards on this thread
  const size_t n_lm = lm_of_point.size(), per = 2048, n_ranges = (n_lm + per - 1) / per;
  std::vector<IndexT> removed(std::max<size_t>(n_ranges, 1), 0);
  std::vector<uint8_t> drop(std::max<size_t>(n_lm, 1), 0);
  mvgx_adapter::host_parallel(n_ranges, [&](uint64_t r, unsigned) {
    IndexT count = 0;
    for (size_t j = r * per, e = std::min(n_lm, (r + 1) * per); j < e; ++j) {
      Landmark& lm = *lm_of_point[j];
      Observations& obs = lm.obs;
      uint64_t k = obs_first[j];
      for (Observations::iterator it = obs.begin(); it != obs.end(); ++k) {
        double v = norm[subset ? obs_old[k] : k];
        if (near_threshold(v, dThresholdPixel)) {   // the reference's own expression decides a borderline observation
          const View* view = sfm_data.views.at(it->first).get();
          const geometry::Pose3 pose = sfm_data.GetPoseOrDie(view);
          const cameras::IntrinsicBase* intrinsic = sfm_data.intrinsics.at(view->id_intrinsic).get();
          v = intrinsic->residual(pose(lm.X), it->second.x).norm();
        }
        if (v > dThresholdPixel) { ++count; it = obs.erase(it); }
        else ++it;
      }
      drop[j] = obs.empty() || obs.size() < minTrackLength;
    }
    removed[r] = count;
  });
  IndexT outlier_count = 0;
  for (size_t r = 0; r < n_ranges; ++r) outlier_count += removed[r];
  for (size_t j = 0; j < n_lm; ++j)
    if (drop[j]) sfm_data.structure.erase(lm_key[j]);
  return outlier_count;
}

IndexT RemoveOutliers_AngleError(SfM_Data& sfm_data, const double dMinAcceptedAngle) {
  FlatScene& fs = mvgx_adapter::flat_scene();
  mvgx_adapter::BoundContext bound;
  if (!bind_scene(sfm_data, fs, bound, "angles")) return RemoveOutliers_AngleError_cpu(sfm_data, dMinAcceptedAngle);
  std::vector<double>& angle = fs.scratch;   // per point of the context's structure (the kept one on the subset route)
  angle.resize(std::max<size_t>(bound.subset ? bound.kept->lm_key.size() : fs.lm_of_point.size(), 1));
  const int rc = mvgx_ba_track_angles(bound.ctx, angle.data());
  struct Release {
    mvgx_adapter::BoundContext& b; FlatScene& fs; bool healthy;
    ~Release() { mvgx_adapter::release_bound_context(b, fs, true, healthy); }
  } release{bound, fs, rc == MVGX_OK};
  const std::vector<Landmark*>& lm_of_point = fs.lm_of_point;
  const std::vector<IndexT>& lm_key = fs.lm_key;
  const std::vector<uint32_t>& point_old = bound.point_old;
  const bool subset = bound.subset;
  if (rc != MVGX_OK) {
    mvgx_adapter::device_failure(mvgx_adapter::kFilters, "outlier filters", "mvgx_ba_track_angles", rc, false);
    return RemoveOutliers_AngleError_cpu(sfm_data, dMinAcceptedAngle);
  }
  mvgx_adapter::counters().device_pairs.fetch_add(1);
  IndexT removedTrack_count = 0;
  for (size_t j = 0; j < lm_of_point.size(); ++j) {
    double max_angle = angle[subset ? point_old[j] : j];
    if (near_threshold(max_angle, dMinAcceptedAngle)) {   // the reference's loop for a borderline track (:84-110)
      const Observations& obs = lm_of_point[j]->obs;
      max_angle = 0.0;
      for (Observations::const_iterator it1 = obs.begin(); it1 != obs.end(); ++it1) {
        const View* view1 = sfm_data.views.at(it1->first).get();
        const geometry::Pose3 pose1 = sfm_data.GetPoseOrDie(view1);
        const cameras::IntrinsicBase* intrinsic1 = sfm_data.intrinsics.at(view1->id_intrinsic).get();
        Observations::const_iterator it2 = it1;
        for (++it2; it2 != obs.end(); ++it2) {
          const View* view2 = sfm_data.views.at(it2->first).get();
          const geometry::Pose3 pose2 = sfm_data.GetPoseOrDie(view2);
          const cameras::IntrinsicBase* intrinsic2 = sfm_data.intrinsics.at(view2->id_intrinsic).get();
          max_angle = std::max(max_angle, AngleBetweenRay(pose1, intrinsic1, pose2, intrinsic2, intrinsic1->get_ud_pixel(it1->second.x),
                                                          intrinsic2->get_ud_pixel(it2->second.x)));
        }
      }
    }
    if (max_angle < dMinAcceptedAngle) {
      sfm_data.structure.erase(lm_key[j]);
      ++removedTrack_count;
    }
  }
  return removedTrack_count;
}

}  // namespace sfm
}  // namespace openMVG
