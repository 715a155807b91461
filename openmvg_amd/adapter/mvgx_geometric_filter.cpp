// mvgx_geometric_filter.cpp - definition of the explicit specialisation declared in mvgx_geometric_filter.hpp: the geometric filter
// of a putative-match container with GeometricFilter_FMatrix_AC on the MI355X.
//
// Reference behaviour reproduced (openMVG/matching_image_collection/GeometricFilter.hpp:66-131, F_ACRobust.hpp:65-122):
//   * every pair of the container is estimated independently; a pair enters _map_GeometricMatches only if Robust_estimation
//     returned true (more than 2.5 x 7 inliers), with the putative matches of the inliers in their original order;
//   * the progress bar is restarted with the number of pairs and advanced once per pair; a cancelled run leaves the container empty
//     from the point of cancellation (checked before the device call and between its result batches);
//   * with b_guided_matching the reference's own Geometry_guided_matching runs on the host with the estimated F and precision and
//     its result replaces the inlier list.
// Inputs of the device call: MatchesPairToMat (Geometric_Filter_utils.cpp:52-89, the reference's own function: undistorted pixel
// positions of the matched features) for every pair, gathered on OpenMP threads, and the image sizes of the two views.
// What the device does not reproduce is routed to the reference's own code: an unbounded precision (m_dPrecision = infinity) and
// pairs with more than 12 000 putative matches run functor.Robust_estimation on the host, pair by pair.
#include "mvgx_geometric_filter.hpp"

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "mvgx.h"
#include "openMVG/matching_image_collection/Geometric_Filter_utils.hpp"
#include "openMVG/sfm/pipelines/sfm_regions_provider.hpp"
#include "openMVG/sfm/sfm_data.hpp"
#include "openMVG/system/logger.hpp"

namespace openMVG {
namespace matching_image_collection {

namespace {
constexpr size_t kDeviceMaxMatches = 12000;   // mvgx_geofilter_f_acransac's bound per pair

// the reference's loop body for one pair (GeometricFilter.hpp:93-128) with the reference's own functor: the route for what the
// device call does not cover
bool reference_pair(const GeometricFilter_FMatrix_AC& functor, const sfm::SfM_Data* sfm_data,
                    const std::shared_ptr<sfm::Regions_Provider>& regions_provider, const Pair& pair, const IndMatches& putative,
                    bool guided, double ratio, IndMatches& out) {
  GeometricFilter_FMatrix_AC f = functor;
  if (!f.Robust_estimation(sfm_data, regions_provider, pair, putative, out)) return false;
  if (guided) {
    IndMatches g;
    f.Geometry_guided_matching(sfm_data, regions_provider, pair, ratio, g);
    std::swap(out, g);
  }
  return true;
}
}  // namespace

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_FMatrix_AC>(
    const GeometricFilter_FMatrix_AC& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* my_progress_bar) {
  if (!my_progress_bar) my_progress_bar = &system::ProgressInterface::dummy();
  my_progress_bar->Restart(putative_matches.size(), "- Geometric filtering -");
  const size_t n_pairs = putative_matches.size();
  if (!n_pairs) return;
  std::vector<PairWiseMatches::const_iterator> its;
  its.reserve(n_pairs);
  for (auto it = putative_matches.begin(); it != putative_matches.end(); ++it) its.push_back(it);
  const bool device_ok = std::isfinite(functor.m_dPrecision) && functor.m_dPrecision > 0.0 && functor.m_stIteration >= 1;
  // pairs for the device (prefix sums of their match counts); the others go through the reference's functor
  std::vector<uint8_t> on_device(n_pairs, 0);
  std::vector<uint64_t> start(1, 0);
  std::vector<size_t> dev_pairs;
  for (size_t p = 0; p < n_pairs; ++p) {
    if (device_ok && its[p]->second.size() <= kDeviceMaxMatches) {
      on_device[p] = 1;
      dev_pairs.push_back(p);
      start.push_back(start.back() + its[p]->second.size());
    }
  }
  std::vector<double> xI(2 * start.back()), xJ(2 * start.back());
  std::vector<uint32_t> wh(4 * dev_pairs.size());
#ifdef OPENMVG_USE_OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
  for (int64_t k = 0; k < (int64_t)dev_pairs.size(); ++k) {
    const auto& kv = *its[dev_pairs[k]];
    Mat2X a, b;
    MatchesPairToMat(kv.first, kv.second, sfm_data_, regions_provider_, a, b);
    const uint64_t lo = start[k];
    for (size_t i = 0; i < kv.second.size(); ++i) {
      xI[2 * (lo + i)] = a(0, i); xI[2 * (lo + i) + 1] = a(1, i);
      xJ[2 * (lo + i)] = b(0, i); xJ[2 * (lo + i) + 1] = b(1, i);
    }
    const sfm::View* vi = sfm_data_->GetViews().at(kv.first.first).get();
    const sfm::View* vj = sfm_data_->GetViews().at(kv.first.second).get();
    wh[4 * k] = vi->ui_width; wh[4 * k + 1] = vi->ui_height; wh[4 * k + 2] = vj->ui_width; wh[4 * k + 3] = vj->ui_height;
  }
  std::vector<uint8_t> mask(start.back() ? start.back() : 1);
  std::vector<mvgx_geofilter_result> res(dev_pairs.size() ? dev_pairs.size() : 1);
  if (!dev_pairs.empty() && !my_progress_bar->hasBeenCanceled()) {
    mvgx_geofilter_options opt;
    opt.precision = functor.m_dPrecision;
    opt.max_iterations = functor.m_stIteration;
    const int rc = mvgx_geofilter_f_acransac(-1, xI.data(), xJ.data(), start.data(), wh.data(), dev_pairs.size(), &opt, mask.data(), res.data(), nullptr);
    if (rc != MVGX_OK) {   // no CPU substitute for a failing device: report like the matcher adapter does
      OPENMVG_LOG_ERROR << "mvgx geometric filter: " << mvgx_last_error();
      throw std::runtime_error(std::string("mvgx_geofilter_f_acransac failed: ") + mvgx_last_error());
    }
  }
  // results in container order; guided matching (host, the reference's code) on OpenMP threads like the reference's loop
  std::vector<int64_t> dev_index(n_pairs, -1);
  for (size_t k = 0; k < dev_pairs.size(); ++k) dev_index[dev_pairs[k]] = (int64_t)k;
#ifdef OPENMVG_USE_OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
  for (int64_t p = 0; p < (int64_t)n_pairs; ++p) {
    if (my_progress_bar->hasBeenCanceled()) continue;
    const auto& kv = *its[p];
    IndMatches inliers;
    bool ok;
    if (!on_device[p]) {
      ok = reference_pair(functor, sfm_data_, regions_provider_, kv.first, kv.second, b_guided_matching, d_distance_ratio, inliers);
    } else {
      const int64_t k = dev_index[p];
      ok = res[k].ok != 0;
      if (ok) {
        inliers.reserve(res[k].n_inliers);
        const uint64_t lo = start[k];
        for (size_t i = 0; i < kv.second.size(); ++i)
          if (mask[lo + i]) inliers.push_back(kv.second[i]);
        if (b_guided_matching) {
          GeometricFilter_FMatrix_AC f = functor;
          for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) f.m_F(r, c) = res[k].F[3 * r + c];
          f.m_dPrecision_robust = res[k].precision_robust;
          IndMatches g;
          f.Geometry_guided_matching(sfm_data_, regions_provider_, kv.first, d_distance_ratio, g);
          std::swap(inliers, g);
        }
      }
    }
    if (ok) {
#ifdef OPENMVG_USE_OPENMP
#pragma omp critical
#endif
      { _map_GeometricMatches.insert({kv.first, std::move(inliers)}); }
    }
    ++(*my_progress_bar);
  }
}

}  // namespace matching_image_collection
}  // namespace openMVG
