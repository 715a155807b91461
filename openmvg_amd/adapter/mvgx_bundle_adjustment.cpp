// mvgx_bundle_adjustment.cpp — Bundle_Adjustment_HIP::Adjust: SfM_Data -> flat mvgx_ba_problem -> libmvgx_hip.so
// -> SfM_Data, following the parameter-block layout and write-back rules of Bundle_Adjustment_Ceres::Adjust
// (sfm/sfm_data_BA_ceres.cpp): pose block = [angle-axis(3), t(3)] with t = -R*C (:260-271); intrinsic block =
// getParams() (:310-320); one residual per observation with blocks (intrinsic, pose, landmark) (:354-396); Huber
// loss with a = Square(4.0) (:249); constant blocks / subsets from Optimize_Options (:274-306, :321-344, :394-395);
// ground control points as weighted, loss-free residuals on constant points (:398-451); pose-centre priors with the
// robust pre-registration of the scene onto them (:180-240, :454-473, :575-606); write-back :527-568.
// No Ceres and no Eigen solver is used: Eigen only as the reference's value types, and
// ceres/rotation.h (header-only templates vendored with openMVG) for the exact angle-axis conversions the reference
// performs on the host.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "ceres/rotation.h"

#include "openMVG/cameras/Camera_Common.hpp"
#include "openMVG/cameras/Camera_Intrinsics.hpp"
#include "openMVG/geometry/Similarity3.hpp"
#include "openMVG/geometry/Similarity3_Kernel.hpp"
#include "openMVG/geometry/pose3.hpp"
#include "openMVG/numeric/eigen_alias_definition.hpp"
#include "openMVG/robust_estimation/robust_estimator_LMeds.hpp"
#include "openMVG/sfm/sfm_data.hpp"
#include "openMVG/sfm/sfm_data_transform.hpp"
#include "openMVG/sfm/sfm_view.hpp"
#include "openMVG/sfm/sfm_view_priors.hpp"
#include "openMVG/stl/stlMap.hpp"
#include "openMVG/system/logger.hpp"
#include "openMVG/types.hpp"

#include "mvgx.h"
#include "mvgx_adapter_policy.hpp"
#include "mvgx_bundle_adjustment.hpp"
#include "mvgx_scene_arrays.hpp"

namespace openMVG {
namespace sfm {

using cameras::Intrinsic_Parameter_Type;
using geometry::Pose3;

namespace {

// Motion priors (sfm_data_BA_ceres.cpp:180-240 before the solve, :454-473 the residuals, :575-606 after it). The reference
// walks the views three times with the same test; here the participating views are listed once and everything else reads
// the current centres through that list. Host work on a handful of 3-vectors, done with the reference's own library
// calls (Similarity3_Kernel + LeastMedianOfSquares, which seeds its sampler deterministically; ApplySimilarity) and the same
// arithmetic, so the scene handed to the solver is the reference's bit for bit.
class PosePriorFrame {
 public:
  PosePriorFrame(SfM_Data& scene, bool requested) : scene_(scene) {
    if (!requested || scene.GetViews().size() <= 3) return;
    for (const auto& v : scene.GetViews()) {
      const ViewPriors* p = dynamic_cast<const ViewPriors*>(v.second.get());
      if (p && p->b_use_pose_center_ && scene.IsPoseAndIntrinsicDefined(p)) views_.push_back(p);
    }
    if (views_.size() <= 3) {
      OPENMVG_LOG_WARNING << "Motion priors ignored: " << views_.size() << " usable pose-centre prior(s), at least 4 are needed";
      return;
    }
    // robust similarity pose centres -> prior centres
    geometry::Similarity3 fit;
    const Mat from = pose_centres(), to = prior_centres();
    geometry::kernel::Similarity3_Kernel kernel(from, to);
    if (robust::LeastMedianOfSquares(kernel, &fit) == std::numeric_limits<double>::max()) return;
    usable_ = true;
    // median distance of the registered centres to their priors: the scale of the priors' Huber loss
    Mat3X moved(3, views_.size());
    for (size_t k = 0; k < views_.size(); ++k) moved.col(k) = fit(Vec3(from.col(k)));
    Vec gap = (moved - Mat3X(to)).colwise().norm();
    std::nth_element(gap.data(), gap.data() + gap.size() / 2, gap.data() + gap.size());
    median_gap_ = gap(gap.size() / 2);
    ApplySimilarity(fit, scene_);
    // conditioning: origin at the centroid of the pose centres, priors moved along
    Vec3 centroid = Vec3::Zero();
    const double n_poses = static_cast<double>(scene_.poses.size());
    for (const auto& pose : scene_.poses) centroid += pose.second.center() / n_poses;
    to_centroid_ = geometry::Similarity3(Pose3(Mat3::Identity(), centroid), 1.0);
    ApplySimilarity(to_centroid_, scene_, true);
  }

  bool usable() const { return usable_; }
  double median_gap() const { return median_gap_; }
  size_t size() const { return usable_ ? views_.size() : 0; }
  const ViewPriors& view(size_t k) const { return *views_[k]; }

  // after the solve: back to the caller's origin, then the fitting statistics of the log
  void leave() {
    if (!usable_) return;
    ApplySimilarity(to_centroid_.inverse(), scene_, true);
    const Vec gap = (Mat3X(pose_centres()) - Mat3X(prior_centres())).colwise().norm();
    std::ostringstream os;
    os << "Pose prior statistics (user units):\n"
       << " - Starting median fitting error: " << median_gap_ << "\n"
       << " - Final fitting error:\n";
    minMaxMeanMedian<Vec::Scalar>(gap.data(), gap.data() + gap.size(), os);
    OPENMVG_LOG_INFO << os.str();
  }

 private:
  Mat pose_centres() const {
    Mat m(3, views_.size());
    for (size_t k = 0; k < views_.size(); ++k) m.col(k) = scene_.GetPoses().at(views_[k]->id_pose).center();
    return m;
  }
  Mat prior_centres() const {
    Mat m(3, views_.size());
    for (size_t k = 0; k < views_.size(); ++k) m.col(k) = views_[k]->pose_center_;
    return m;
  }
  SfM_Data& scene_;
  std::vector<const ViewPriors*> views_;
  bool usable_ = false;
  double median_gap_ = 0.0;
  geometry::Similarity3 to_centroid_;
};

}  // namespace

bool Bundle_Adjustment_HIP::Adjust(SfM_Data& sfm_data, const Optimize_Options& options) {
  // MVGX_ADAPTER_TIMING=1: the phases of this call on stderr
  const bool timing = std::getenv("MVGX_ADAPTER_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto tick = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[mvgx Adjust] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  PosePriorFrame priors(sfm_data, options.use_motion_priors_opt);   // may move the whole scene (undone by priors.leave())
  const bool b_usable_prior = priors.usable();

  // --- parameter blocks ---
  // The flat arrays live in a per-thread store that keeps its memory between calls: 30 MB (200 views / 1 M observations) to
  // 150 MB (1 000 / 5 M) of fresh pageable memory per call were a zero fill on this thread plus a page fault per 4 KB in the
  // flatten threads - about a third of the 5 ms the walk took.
  std::unordered_map<IndexT, uint32_t> pose_idx, intr_idx;
  mvgx_adapter::FlatScene& fs = mvgx_adapter::flat_scene();
  std::vector<IndexT>&pose_ids = fs.pose_ids, &intr_ids = fs.intr_ids;
  pose_ids.clear(); intr_ids.clear();
  std::vector<double>&poses = fs.poses, &intrinsics = fs.intrinsics, &points = fs.points, &obs_xy = fs.obs_xy;
  std::vector<int32_t>& intr_model = fs.intr_model;
  std::vector<uint8_t>&pose_mask = fs.pose_mask, &intr_mask = fs.intr_mask;
  std::vector<uint32_t>&obs_pose = fs.obs_pose, &obs_intr = fs.obs_intr, &obs_point = fs.obs_point;
  poses.clear(); intrinsics.clear(); intr_model.clear(); pose_mask.clear(); intr_mask.clear();

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
    std::vector<double> prm = it.second->getParams();
    if (prm.size() > MVGX_BA_MAX_INTR_PARAMS) {
      OPENMVG_LOG_ERROR << "mvgx BA: camera model " << static_cast<int>(it.second->getType()) << " has no device functor.";
      return false;
    }
    const bool no_block = prm.empty();   // CAMERA_SPHERICAL: residual blocks take (pose, point) only (:366-383)
    if (no_block) prm = {static_cast<double>(it.second->w()), static_cast<double>(it.second->h())};   // data of the functor
    uint8_t m = 0;
    if (no_block) {
      m = 0;
    } else if (options.intrinsics_opt == Intrinsic_Parameter_Type::NONE) {
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

  tick("priors, cameras");
  // --- observations (landmark X is refined in place, as the reference hands X.data() to the solver) ---
  // (the walk of the two unordered_maps on the library's host workers: mvgx_scene_arrays.hpp)
  int flatten_error = 0;   // 1: an observation of a view without usable intrinsic, 2: of a view without pose / unknown view
  const uint64_t n_structure_obs64 = mvgx_adapter::flatten_observations(sfm_data, pose_idx, intr_idx, fs, &flatten_error, tick);
  std::vector<Landmark*>& lm_of_point = fs.lm_of_point;
  tick("  observation rows");
  if (flatten_error == 1) {
    OPENMVG_LOG_ERROR << "Cannot create a CostFunction for this camera model.";
    return false;
  }
  if (flatten_error == 2) throw std::out_of_range("mvgx BA: a landmark observes a view without pose");   // the reference's .at() throws here as well

  // --- ground control points: constant points, weighted residuals without loss function (:398-451) ---
  const size_t n_structure_obs = obs_pose.size();
  std::vector<double> obs_weight;
  std::vector<uint8_t> obs_is_control, point_const;
  size_t n_points_total = lm_of_point.size();
  if (options.control_point_opt.bUse_control_points) {
    for (auto& gcp : sfm_data.control_points) {
      if (gcp.second.obs.empty()) {
        OPENMVG_LOG_ERROR << "Cannot use this GCP id: " << gcp.first << ". There is not linked image observation.";
        continue;
      }
      const uint32_t j = static_cast<uint32_t>(n_points_total++);
      points.insert(points.end(), {gcp.second.X(0), gcp.second.X(1), gcp.second.X(2)});
      for (const auto& ob : gcp.second.obs) {
        const View* view = sfm_data.views.at(ob.first).get();
        const auto ii = intr_idx.find(view->id_intrinsic);
        if (ii == intr_idx.end()) continue;   // IntrinsicsToCostFunction returned null: the reference skips the block
        obs_pose.push_back(pose_idx.at(view->id_pose));
        obs_intr.push_back(ii->second);
        obs_point.push_back(j);
        obs_xy.push_back(ob.second.x(0));
        obs_xy.push_back(ob.second.x(1));
      }
    }
    if (n_points_total > lm_of_point.size()) {
      obs_weight.assign(obs_pose.size(), 0.0);
      obs_is_control.assign(obs_pose.size(), 0);
      for (size_t k = n_structure_obs; k < obs_pose.size(); ++k) { obs_weight[k] = options.control_point_opt.weight; obs_is_control[k] = 1; }
      point_const.assign(n_points_total, 0);
      for (size_t j = lm_of_point.size(); j < n_points_total; ++j) point_const[j] = 1;
    }
  }
  // --- pose-centre priors (:454-473); the reference indexes the pose block with prior->id_view ---
  std::vector<uint32_t> prior_pose;
  std::vector<double> prior_center, prior_weight;
  for (size_t k = 0; k < priors.size(); ++k) {
    const ViewPriors& v = priors.view(k);
    prior_pose.push_back(pose_idx.at(v.id_view));
    for (int a = 0; a < 3; ++a) { prior_center.push_back(v.pose_center_(a)); prior_weight.push_back(v.center_weight_(a)); }
  }

  mvgx_ba_problem prob{};
  if (!obs_weight.empty()) { prob.obs_weight = obs_weight.data(); prob.obs_is_control = obs_is_control.data(); prob.point_const_mask = point_const.data(); }
  prob.n_pose_priors = static_cast<uint32_t>(prior_pose.size());
  prob.prior_pose = prior_pose.data(); prob.prior_center = prior_center.data(); prob.prior_weight = prior_weight.data();
  prob.prior_huber_a = Square(priors.median_gap());
  prob.n_poses = static_cast<uint32_t>(pose_ids.size());
  prob.n_intrinsics = static_cast<uint32_t>(intr_ids.size());
  prob.n_points = static_cast<uint32_t>(n_points_total);
  prob.n_obs = obs_pose.size();
  prob.poses = poses.data(); prob.intrinsics = intrinsics.data(); prob.intr_model = intr_model.data();
  prob.points = points.data();
  prob.obs_pose = obs_pose.data(); prob.obs_intr = obs_intr.data(); prob.obs_point = obs_point.data();
  prob.obs_xy = obs_xy.data();
  prob.pose_const_mask = pose_mask.data(); prob.intr_const_mask = intr_mask.data();
  prob.points_constant = options.structure_opt == Structure_Parameter_Type::NONE ? 1 : 0;
  prob.huber_a = options_.bUse_loss_function_ ? Square(4.0) : 0.0;

  tick("scene -> arrays");
  // the context: the kept one re-bound (same structure: values only; the same scene minus observations: those switched off), or a
  // new one (mvgx_scene_arrays.hpp)
  const bool plain = obs_weight.empty() && prior_pose.empty();
  mvgx_adapter::BoundContext bound;
  const bool inj_create = mvgx_adapter::injected("ba", "create");
  int rc = inj_create ? MVGX_ERR_NODEV : mvgx_adapter::bind_context(options_.device_, prob, fs, plain, bound, options_.linear_solver_);
  mvgx_ba_ctx* ctx = bound.ctx;
  tick(bound.route);
  if (rc == MVGX_ERR_UNSUPPORTED) {
    OPENMVG_LOG_ERROR << "Cannot create a CostFunction for this camera model. (" << mvgx_last_error() << ")";
    return false;
  }
  if (rc != MVGX_OK) {   // the reference's convention: log, return false (mvgx_adapter_policy.hpp; MVGX_ON_DEVICE_ERROR=throw throws)
    mvgx_adapter::device_failure(mvgx_adapter::kBundle, "bundle adjustment", "mvgx_ba_create", rc, inj_create);
    return false;
  }
  mvgx_ba_options opt;
  mvgx_ba_default_options(&opt);
  opt.max_num_iterations = options_.max_num_iterations_;
  opt.parameter_tolerance = options_.parameter_tolerance_;
  opt.gradient_tolerance = options_.gradient_tolerance_;
  mvgx_ba_summary summary{};
  rc = mvgx_ba_solve(ctx, &opt, &summary);
  tick("mvgx_ba_solve");
  // the solver state is written back to the landmarks in every case: the reference optimises X in place, so a failed
  // solve leaves moved points behind as well (sfm_data_BA_ceres.cpp:378, :503-507)
  const int rc_read = mvgx_ba_read_params(ctx, poses.data(), intrinsics.data(), bound.subset ? bound.points_old.data() : points.data());
  if (rc_read == MVGX_OK && bound.subset)   // the context holds the kept structure: this scene's points are a selection of its points
    for (size_t j = 0; j < bound.point_old.size(); ++j)
      for (int a = 0; a < 3; ++a) points[3 * j + a] = bound.points_old[3 * static_cast<size_t>(bound.point_old[j]) + a];
  // (the context goes back into the slot after the write-back: that hands the flat arrays over with it)
  struct Release {
    mvgx_adapter::BoundContext& b; mvgx_adapter::FlatScene& fs; bool plain, healthy;
    ~Release() { mvgx_adapter::release_bound_context(b, fs, plain, healthy); }
  } release{bound, fs, plain, rc_read == MVGX_OK && (rc == MVGX_OK || rc == MVGX_ERR_NUMERIC)};   // (a context whose device calls failed is not kept)
  tick("read_params");
  if (rc_read != MVGX_OK) {
    OPENMVG_LOG_ERROR << "mvgx BA: " << mvgx_last_error();
    return false;
  }
  if (!prob.points_constant) {
    const size_t n_lm = lm_of_point.size(), per = 4096;
    mvgx_adapter::host_parallel((n_lm + per - 1) / per, [&](uint64_t g, unsigned) {
      for (size_t j = g * per, e = std::min(n_lm, (g + 1) * per); j < e; ++j) lm_of_point[j]->X = Vec3(points[3 * j], points[3 * j + 1], points[3 * j + 2]);
    });
  }
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
                     << " LM iterations)\n--\n Used motion prior: " << static_cast<int>(b_usable_prior);
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
  priors.leave();
  tick("write-back");
  return true;
}

}  // namespace sfm
}  // namespace openMVG

// diagnostic / test entries of an adapter library that holds this TU: {contexts created, contexts re-bound by mvgx_ba_update};
// release: destroys the idle context (its device memory goes back to the library's slab cache)
extern "C" void mvgx_adapter_ba_context_stats(uint64_t out[2], int reset) {
  auto& c = mvgx_adapter::context_cache();
  if (out) { out[0] = c.created.load(); out[1] = c.reused.load() + c.subset.load(); }
  if (reset) { c.created = 0; c.reused = 0; c.subset = 0; }
}
// ... {created, re-bound with the same structure, re-bound with observations switched off}
extern "C" void mvgx_adapter_ba_context_stats3(uint64_t out[3], int reset) {
  auto& c = mvgx_adapter::context_cache();
  if (out) { out[0] = c.created.load(); out[1] = c.reused.load(); out[2] = c.subset.load(); }
  if (reset) { c.created = 0; c.reused = 0; c.subset = 0; }
}
// how the kept context solves its reduced camera system (mvgx_ba_get_solver_info; MVGX_ERR_STATE: no context is kept)
extern "C" int mvgx_adapter_ba_kept_solver_info(mvgx_ba_solver_info* out) {
  auto& c = mvgx_adapter::context_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  return c.idle ? mvgx_ba_get_solver_info(c.idle, out) : MVGX_ERR_STATE;
}
extern "C" void mvgx_adapter_ba_release_context() {
  mvgx_ba_ctx* ctx = mvgx_adapter::take_idle_context(std::numeric_limits<int>::min());   // (no device matches: the idle context and its kept arrays are destroyed)
  (void)ctx;
}
