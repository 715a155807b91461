// mvgx_geometric_filter.cpp - definitions of the explicit specialisations declared in mvgx_geometric_filter.hpp: the geometric filter
// of a putative-match container with GeometricFilter_FMatrix_AC, GeometricFilter_HMatrix_AC or GeometricFilter_EMatrix_AC on the MI355X.
//
// Reference behaviour reproduced (openMVG/matching_image_collection/GeometricFilter.hpp:66-131, F_ACRobust.hpp:65-122,
// H_ACRobust.hpp:49-113):
//   * every pair of the container is estimated independently; a pair enters _map_GeometricMatches only if Robust_estimation
//     returned true (more than 2.5 x 7 inliers for F, 2.5 x 4 for H), with the putative matches of the inliers in their original order;
//   * the progress bar is restarted with the number of pairs and advanced once per pair; the device is called in batches of
//     kPairsPerCall pairs and cancellation is checked between them: a cancelled run leaves the container empty from that point;
//   * with b_guided_matching the functor's Geometry_guided_matching (F_ACRobust.hpp:109-152, H_ACRobust.hpp:136-180, E_ACRobust.hpp:153-215 ->
//     robust_estimation/guided_matching.hpp:178-227) replaces the inlier list. Round 5: for uint8 descriptors of 64 / 128 / 144 bytes it runs
//     on the device as well (mvgx_guided_match_u8: the estimated model, Square(m_dPrecision_robust), Square(dDistanceRatio), the
//     undistorted positions the estimation stage already holds and the regions' descriptor bytes); any other region type, a pair the
//     device did not estimate or a failing call takes the reference's own member function (mvgx_adapter_guided_counters tells which ran).
// Inputs of the device call (mvgx_geofilter_f_acransac_indexed): the undistorted pixel positions of the features of every view that
// occurs (the expressions of MatchesPointsToMat, Geometric_Filter_utils.cpp:33-49, once per feature on OpenMP threads), the index
// pairs of the putative matches as the container holds them, and the image sizes of the views.
// What the device does not reproduce is routed to the reference's own code: an unbounded precision (m_dPrecision = infinity) and
// pairs with more than 2^20 putative matches run functor.Robust_estimation on the host, pair by pair - and so do the pairs of a
// batch whose device call failed (logged once; mvgx_adapter_policy.hpp; MVGX_ON_DEVICE_ERROR=throw stops instead).
#include "mvgx_geometric_filter.hpp"
#include "openMVG/matching_image_collection/H_ACRobust.hpp"
#include "openMVG/matching_image_collection/E_ACRobust.hpp"
#include "openMVG/cameras/Camera_Pinhole.hpp"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "mvgx.h"
#include "mvgx_adapter_policy.hpp"
#include "openMVG/cameras/Camera_Intrinsics.hpp"
#include "openMVG/features/feature.hpp"
#include "openMVG/matching_image_collection/Geometric_Filter_utils.hpp"
#include "openMVG/multiview/essential.hpp"
#include "openMVG/sfm/pipelines/sfm_regions_provider.hpp"
#include "openMVG/sfm/sfm_data.hpp"
#include "openMVG/system/logger.hpp"

namespace openMVG {
namespace matching_image_collection {

namespace {
constexpr size_t kDeviceMaxMatches = size_t(1) << 20;   // mvgx_geofilter_f_acransac's bound per pair
constexpr size_t kPairsPerCall = 65536;                 // pairs per device call: progress / cancellation granularity, bounded staging memory

// the reference's loop body for one pair (GeometricFilter.hpp:93-128) with the reference's own functor: the route for what the
// device call does not cover
template <class Functor>
bool reference_pair(const Functor& functor, const sfm::SfM_Data* sfm_data,
                    const std::shared_ptr<sfm::Regions_Provider>& regions_provider, const Pair& pair, const IndMatches& putative,
                    bool guided, double ratio, IndMatches& out) {
  Functor f = functor;
  if (!f.Robust_estimation(sfm_data, regions_provider, pair, putative, out)) return false;
  if (guided) {
    IndMatches g;
    f.Geometry_guided_matching(sfm_data, regions_provider, pair, ratio, g);
    std::swap(out, g);
  }
  return true;
}

// what differs between the two functors: the member that holds the model and the C entry point
template <class Functor> struct ModelOf;
// (run: the indexed C entry with one argument list for the three models - the bearing vectors and calibration matrices only reach the
// essential entry)
template <> struct ModelOf<GeometricFilter_FMatrix_AC> {
  static Mat3& model(GeometricFilter_FMatrix_AC& f) { return f.m_F; }
  static constexpr const char* entry_name = "mvgx_geofilter_f_acransac_indexed";
  static constexpr bool essential = false, angular = false;
  static double precision(const GeometricFilter_FMatrix_AC& f) { return f.m_dPrecision; }
  static void set_robust_precision(GeometricFilter_FMatrix_AC& f, double v) { f.m_dPrecision_robust = v; }
  static int run(const double* xy, const double*, const uint64_t* fs, const uint32_t* wh, const double*, uint32_t nv, const uint32_t* pv, const uint64_t* st,
                 const uint32_t* ij, uint64_t nb, const mvgx_geofilter_options* o, uint8_t* m, mvgx_geofilter_result* r) {
    return mvgx_geofilter_f_acransac_indexed(-1, xy, fs, wh, nv, pv, st, ij, nb, o, m, r, nullptr);
  }
};
template <> struct ModelOf<GeometricFilter_HMatrix_AC> {
  static Mat3& model(GeometricFilter_HMatrix_AC& f) { return f.m_H; }
  static constexpr const char* entry_name = "mvgx_geofilter_h_acransac_indexed";
  static constexpr bool essential = false, angular = false;
  static double precision(const GeometricFilter_HMatrix_AC& f) { return f.m_dPrecision; }
  static void set_robust_precision(GeometricFilter_HMatrix_AC& f, double v) { f.m_dPrecision_robust = v; }
  static int run(const double* xy, const double*, const uint64_t* fs, const uint32_t* wh, const double*, uint32_t nv, const uint32_t* pv, const uint64_t* st,
                 const uint32_t* ij, uint64_t nb, const mvgx_geofilter_options* o, uint8_t* m, mvgx_geofilter_result* r) {
    return mvgx_geofilter_h_acransac_indexed(-1, xy, fs, wh, nv, pv, st, ij, nb, o, m, r, nullptr);
  }
};
template <> struct ModelOf<GeometricFilter_EMatrix_AC> {
  static Mat3& model(GeometricFilter_EMatrix_AC& f) { return f.m_E; }
  static constexpr const char* entry_name = "mvgx_geofilter_e_acransac_indexed";
  static constexpr bool essential = true, angular = false;
  static double precision(const GeometricFilter_EMatrix_AC& f) { return f.m_dPrecision; }
  static void set_robust_precision(GeometricFilter_EMatrix_AC& f, double v) { f.m_dPrecision_robust = v; }
  static int run(const double* xy, const double* bearing, const uint64_t* fs, const uint32_t* wh, const double* K, uint32_t nv, const uint32_t* pv,
                 const uint64_t* st, const uint32_t* ij, uint64_t nb, const mvgx_geofilter_options* o, uint8_t* m, mvgx_geofilter_result* r) {
    return mvgx_geofilter_e_acransac_indexed(-1, xy, bearing, fs, wh, K, nv, pv, st, ij, nb, o, m, r, nullptr);
  }
};

// the angular essential functors (E_ACRobust_Angular.hpp:33-191; -g a: eight-point solver, -g u: three-point upright solver): bearing
// vectors of any camera model, no pixels; their second stage (RelativePoseFromEssential on the a-contrario inliers, :126-143) runs
// below with the reference's own function
template <bool kUpright> struct ModelOf<GeometricFilter_ESphericalMatrix_AC_Angular<kUpright>> {
  using F = GeometricFilter_ESphericalMatrix_AC_Angular<kUpright>;
  static Mat3& model(F& f) { return f.m_E; }
  static constexpr const char* entry_name = "mvgx_geofilter_e_angular_acransac_indexed";
  static constexpr bool essential = false, angular = true;
  static constexpr int min_samples = kUpright ? 3 : 8;   // Solver::MINIMUM_SAMPLES
  static double precision(const F& f) { return f.m_precision_upper_bound; }
  static void set_robust_precision(F& f, double v) { f.m_precision_upper_bound_robust = v; }
  static int run(const double*, const double* bearing, const uint64_t* fs, const uint32_t*, const double*, uint32_t nv, const uint32_t* pv,
                 const uint64_t* st, const uint32_t* ij, uint64_t nb, const mvgx_geofilter_options* o, uint8_t* m, mvgx_geofilter_result* r) {
    return mvgx_geofilter_e_angular_acransac_indexed(-1, bearing, fs, nv, pv, st, ij, nb, kUpright ? 1 : 0, o, m, r, nullptr);
  }
};

// the orthographic essential functor (Eo_Robust.hpp:35-165; -g o): pinhole cameras only (the functor rejects the others, :85-88); the
// device takes the hnormalized bearing vector of every feature in place of its pixel position and the bound of every pair (the mean
// of the two cameras' imagePlane_toCameraPlaneError(precision^2), :96-100)
template <> struct ModelOf<GeometricFilter_EOMatrix_RA> {
  using F = GeometricFilter_EOMatrix_RA;
  static Mat3& model(F& f) { return f.m_E; }
  static constexpr const char* entry_name = "mvgx_geofilter_eo_acransac_indexed";
  static constexpr bool essential = false, angular = false;
  static double precision(const F& f) { return f.m_dPrecision; }
  static void set_robust_precision(F&, double) {}
};
template <class Functor> struct IsOrtho { static constexpr bool value = false; };
template <> struct IsOrtho<GeometricFilter_EOMatrix_RA> { static constexpr bool value = true; };

// the body of both specialisations; the members of ImageCollectionGeometricFilter it works on are passed in under their names
template <class Functor>
void filter_container(const sfm::SfM_Data* sfm_data_, const std::shared_ptr<sfm::Regions_Provider>& regions_provider_,
                      PairWiseMatches& _map_GeometricMatches, const Functor& functor, const PairWiseMatches& putative_matches,
                      const bool b_guided_matching, const double d_distance_ratio, system::ProgressInterface* my_progress_bar) {
  if (!my_progress_bar) my_progress_bar = &system::ProgressInterface::dummy();
  my_progress_bar->Restart(putative_matches.size(), "- Geometric filtering -");
  const size_t n_pairs = putative_matches.size();
  if (!n_pairs) return;
  std::vector<PairWiseMatches::const_iterator> its;
  its.reserve(n_pairs);
  for (auto it = putative_matches.begin(); it != putative_matches.end(); ++it) its.push_back(it);
  using M = ModelOf<Functor>;
  constexpr bool kBearings = M::essential || M::angular;
  constexpr bool kOrtho = IsOrtho<Functor>::value;
  const double precision = M::precision(functor);
  const bool device_ok = std::isfinite(precision) && precision > 0.0 && functor.m_stIteration >= 1;
  // pairs for the device (prefix sums of their match counts); the others go through the reference's functor
  std::vector<uint8_t> on_device(n_pairs, 0);
  std::vector<uint64_t> start(1, 0);
  std::vector<size_t> dev_pairs;
  // the essential model needs a pinhole intrinsic on both views (E_ACRobust.hpp:76-100): the other pairs take the reference's functor,
  // which warns and rejects them
  auto pinhole_view = [&](IndexT id) {
    const auto vit = sfm_data_->GetViews().find(id);
    if (vit == sfm_data_->GetViews().end()) return false;
    const auto iit = sfm_data_->GetIntrinsics().find(vit->second->id_intrinsic);
    return iit != sfm_data_->GetIntrinsics().end() && iit->second && cameras::isPinhole(iit->second->getType());
  };
  // the angular models take any camera model, but both views need one (E_ACRobust_Angular.hpp:72-88)
  auto calibrated_view = [&](IndexT id) {
    const auto vit = sfm_data_->GetViews().find(id);
    if (vit == sfm_data_->GetViews().end()) return false;
    const auto iit = sfm_data_->GetIntrinsics().find(vit->second->id_intrinsic);
    return iit != sfm_data_->GetIntrinsics().end() && iit->second;
  };
  for (size_t p = 0; p < n_pairs; ++p) {
    if (device_ok && its[p]->second.size() <= kDeviceMaxMatches &&
        (!(M::essential || kOrtho) || (pinhole_view(its[p]->first.first) && pinhole_view(its[p]->first.second))) &&
        (!M::angular || (calibrated_view(its[p]->first.first) && calibrated_view(its[p]->first.second)))) {
      on_device[p] = 1;
      dev_pairs.push_back(p);
      start.push_back(start.back() + its[p]->second.size());
    }
  }
  // The device call takes the container as it is: the (undistorted) positions of every feature of the views that occur - computed
  // once per view, with the expressions of MatchesPointsToMat (Geometric_Filter_utils.cpp:33-49), where the reference recomputes
  // them for every match of every pair - and the index pairs of the matches; the gather runs on the device.
  static_assert(sizeof(matching::IndMatch) == 2 * sizeof(uint32_t), "IndMatch is two 32-bit indices");
  std::map<IndexT, uint32_t> view_slot;
  for (size_t k : dev_pairs) { view_slot.emplace(its[k]->first.first, 0u); view_slot.emplace(its[k]->first.second, 0u); }
  std::vector<IndexT> slot_view;
  for (auto& kv : view_slot) { kv.second = (uint32_t)slot_view.size(); slot_view.push_back(kv.first); }
  const size_t n_views = slot_view.size();
  std::vector<features::PointFeatures> positions(n_views);
  std::vector<uint64_t> feat_start(n_views + 1, 0);
  std::vector<uint32_t> wh(2 * std::max<size_t>(n_views, 1));
#ifdef OPENMVG_USE_OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
  for (int64_t v = 0; v < (int64_t)n_views; ++v) {
    positions[v] = regions_provider_->get(slot_view[v])->GetRegionsPositions();
    feat_start[v + 1] = positions[v].size();
    const sfm::View* view = sfm_data_->GetViews().at(slot_view[v]).get();
    wh[2 * v] = view->ui_width; wh[2 * v + 1] = view->ui_height;
  }
  for (size_t v = 0; v < n_views; ++v) feat_start[v + 1] += feat_start[v];
  std::vector<double> feat_xy(2 * std::max<uint64_t>(feat_start[n_views], 1));
  // essential model: the bearing vector of every feature by the camera's own operator() on the undistorted position (what
  // Robust_estimation passes as (*cam_I)(xI), E_ACRobust.hpp:118-123) and the calibration matrix of every view
  std::vector<double> feat_bearing(kBearings ? 3 * std::max<uint64_t>(feat_start[n_views], 1) : 0);
  std::vector<double> view_K(ModelOf<Functor>::essential ? 9 * std::max<size_t>(n_views, 1) : 0);
#ifdef OPENMVG_USE_OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
  for (int64_t v = 0; v < (int64_t)n_views; ++v) {
    const sfm::View* view = sfm_data_->GetViews().at(slot_view[v]).get();
    const cameras::IntrinsicBase* cam =
        sfm_data_->GetIntrinsics().count(view->id_intrinsic) ? sfm_data_->GetIntrinsics().at(view->id_intrinsic).get() : nullptr;
    double* dst = feat_xy.data() + 2 * feat_start[v];
    for (size_t i = 0; i < positions[v].size(); ++i) {
      const Vec2 x = cam ? Vec2(cam->get_ud_pixel(positions[v][i].coords().cast<double>())) : Vec2(positions[v][i].coords().cast<double>());
      dst[2 * i] = x(0); dst[2 * i + 1] = x(1);
    }
    if (kOrtho && cam && cameras::isPinhole(cam->getType())) {   // (*cam)(x).colwise().hnormalized() in place of the position
      const size_t n = positions[v].size();
      Mat2X pts(2, n);
      for (size_t i = 0; i < n; ++i) pts.col(i) << dst[2 * i], dst[2 * i + 1];
      const Mat2X h = (*cam)(pts).colwise().hnormalized();
      for (size_t i = 0; i < n; ++i) { dst[2 * i] = h(0, i); dst[2 * i + 1] = h(1, i); }
    }
    if (kBearings) {
      const cameras::Pinhole_Intrinsic* pin = dynamic_cast<const cameras::Pinhole_Intrinsic*>(cam);
      if (M::angular ? cam != nullptr : pin != nullptr) {   // (views without a (pinhole) camera only occur in pairs that are not on the device)
        const size_t n = positions[v].size();
        Mat2X pts(2, n);
        for (size_t i = 0; i < n; ++i) pts.col(i) << dst[2 * i], dst[2 * i + 1];
        const Mat3X b = (*cam)(pts);
        double* bd = feat_bearing.data() + 3 * feat_start[v];
        for (size_t i = 0; i < n; ++i) { bd[3 * i] = b(0, i); bd[3 * i + 1] = b(1, i); bd[3 * i + 2] = b(2, i); }
        if (M::essential) {
          const Mat3& Km = pin->K();
          for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) view_K[9 * v + 3 * r + c] = Km(r, c);
        }
      }
    }
    features::PointFeatures().swap(positions[v]);
  }
  std::vector<uint32_t> pair_views(2 * std::max<size_t>(dev_pairs.size(), 1)), ij(2 * std::max<uint64_t>(start.back(), 1));
#ifdef OPENMVG_USE_OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
  for (int64_t k = 0; k < (int64_t)dev_pairs.size(); ++k) {
    const auto& kv = *its[dev_pairs[k]];
    pair_views[2 * k] = view_slot.at(kv.first.first); pair_views[2 * k + 1] = view_slot.at(kv.first.second);
    if (!kv.second.empty()) std::memcpy(ij.data() + 2 * start[k], kv.second.data(), kv.second.size() * sizeof(matching::IndMatch));
  }
  std::vector<double> pair_precision(kOrtho ? std::max<size_t>(dev_pairs.size(), 1) : 0);
  if (kOrtho)
    for (size_t k = 0; k < dev_pairs.size(); ++k) {
      const auto& kv = *its[dev_pairs[k]];
      const cameras::IntrinsicBase* cI = sfm_data_->GetIntrinsics().at(sfm_data_->GetViews().at(kv.first.first)->id_intrinsic).get();
      const cameras::IntrinsicBase* cJ = sfm_data_->GetIntrinsics().at(sfm_data_->GetViews().at(kv.first.second)->id_intrinsic).get();
      pair_precision[k] = (cI->imagePlane_toCameraPlaneError(Square(precision)) + cJ->imagePlane_toCameraPlaneError(Square(precision))) / 2.;
    }
  std::vector<uint8_t> mask(start.back() ? start.back() : 1);
  std::vector<mvgx_geofilter_result> res(dev_pairs.size() ? dev_pairs.size() : 1);
  size_t n_dev_done = 0;   // dev_pairs[0, n_dev_done): results valid
  if (!dev_pairs.empty()) {
    mvgx_geofilter_options opt;
    opt.precision = precision;
    opt.max_iterations = functor.m_stIteration;
    std::vector<uint64_t> start_b;
    for (size_t b0 = 0; b0 < dev_pairs.size() && !my_progress_bar->hasBeenCanceled(); b0 += kPairsPerCall) {
      const size_t nb = std::min(kPairsPerCall, dev_pairs.size() - b0);
      start_b.assign(nb + 1, 0);
      for (size_t k = 0; k <= nb; ++k) start_b[k] = start[b0 + k] - start[b0];   // a call's match_start begins at zero
      const bool inj = mvgx_adapter::injected("geofilter", "run");
      int rc = MVGX_ERR_NODEV;
      if constexpr (kOrtho) {
        if (!inj)
          rc = mvgx_geofilter_eo_acransac_indexed(-1, feat_xy.data(), feat_start.data(), wh.data(), (uint32_t)n_views, pair_views.data() + 2 * b0, start_b.data(),
                                                  ij.data() + 2 * start[b0], pair_precision.data() + b0, (uint64_t)nb, &opt, mask.data() + start[b0], res.data() + b0,
                                                  nullptr);
      } else if (!inj) {
        rc = ModelOf<Functor>::run(feat_xy.data(), feat_bearing.data(), feat_start.data(), wh.data(), view_K.data(), (uint32_t)n_views,
                                   pair_views.data() + 2 * b0, start_b.data(), ij.data() + 2 * start[b0], (uint64_t)nb, &opt, mask.data() + start[b0],
                                   res.data() + b0);
      }
      if (rc != MVGX_OK) {
        // logged once; the pairs from here on take the reference's own functor below (or the failure is thrown)
        mvgx_adapter::device_failure(mvgx_adapter::kGeofilter, "geometric filter", ModelOf<Functor>::entry_name, rc, inj);
        break;
      }
      n_dev_done = b0 + nb;
    }
    if (!my_progress_bar->hasBeenCanceled()) {
      for (size_t k = n_dev_done; k < dev_pairs.size(); ++k) on_device[dev_pairs[k]] = 0;
      mvgx_adapter::counters().fallback_pairs.fetch_add(dev_pairs.size() - n_dev_done);
    }
    mvgx_adapter::counters().device_pairs.fetch_add(n_dev_done);
  }
  // ---- second stage on the device: guided matching of the accepted pairs (the models and bounds of the first stage never leave the process) ----
  std::vector<uint8_t> guided_done(dev_pairs.size() ? dev_pairs.size() : 1, 0);
  // one entry per device call (the accepted pairs go in batches, below): the call's offsets and match list
  struct GuidedBatch { std::vector<uint64_t> start; uint32_t* ij = nullptr; };
  std::vector<GuidedBatch> guided_batches;
  struct FreeGuided { std::vector<GuidedBatch>& b; ~FreeGuided() { for (GuidedBatch& g : b) if (g.ij) mvgx_host_free(g.ij); } } free_guided{guided_batches};
  std::vector<size_t> guided_pairs;          // positions in dev_pairs of the pairs handed to the guided calls, in call order
  std::vector<uint32_t> guided_batch_of;     // per entry of guided_pairs: its call, and its position in that call
  std::vector<uint32_t> guided_pos_in_batch;
  // H_ACRobust.hpp:166-187: a NEGATIVE ratio selects the homography functor's geometry-only matching (nearest position under H, then
  // IndMatch and (x, y) de-duplication) - what main_GeometricFilter passes for -g h. The device kernel is the descriptor-ratio form:
  // such a call keeps the functor's own member function. The F and E functors square the ratio whatever its sign.
  const bool guided_on_device = !(std::is_same<Functor, GeometricFilter_HMatrix_AC>::value && d_distance_ratio < 0);
  if constexpr (!M::angular && !kOrtho) {
    if (b_guided_matching && guided_on_device && n_dev_done > 0 && !my_progress_bar->hasBeenCanceled()) {
      // the regions' descriptors, in the feature order of feat_start: regions of ONE type and length - uint8 scalar rows (SIFT, LIOP:
      // L2<uint8_t>), float scalar rows (AKAZE float: L2<float>) or binary rows (AKAZE binary: squared Hamming), the three metrics
      // Regions::SquaredDescriptorDistance resolves to (scalar_regions.hpp:107-116, binary_regions.hpp:109-120; round 6)
      uint32_t desc_bytes = 0;
      int desc_type = -1;
      bool usable = true;
      for (size_t v = 0; v < n_views && usable; ++v) {
        const std::shared_ptr<features::Regions> r = regions_provider_->get(slot_view[v]);
        usable = r && r->RegionCount() == feat_start[v + 1] - feat_start[v];
        if (!usable) break;
        const int t = r->IsBinary() ? MVGX_DESC_BINARY
                      : r->IsScalar() && r->Type_id() == typeid(unsigned char).name() ? MVGX_DESC_U8
                      : r->IsScalar() && r->Type_id() == typeid(float).name() ? MVGX_DESC_F32 : -1;
        const uint32_t nb = (uint32_t)(r->DescriptorLength() * (t == MVGX_DESC_F32 ? sizeof(float) : 1));
        usable = t >= 0 && (desc_type < 0 || (t == desc_type && nb == desc_bytes));
        if (usable) { desc_type = t; desc_bytes = nb; }
      }
      usable = usable && (desc_type == MVGX_DESC_U8 ? (desc_bytes == 64 || desc_bytes == 128 || desc_bytes == 144)
                          : desc_type == MVGX_DESC_F32 ? (desc_bytes == 256 || desc_bytes == 512)
                                                       : (desc_bytes == 32 || desc_bytes == 64));
      if (usable) {
        for (size_t k = 0; k < n_dev_done; ++k)
          if (res[k].ok) guided_pairs.push_back(k);
        std::vector<uint8_t> desc((size_t)feat_start[n_views] * desc_bytes + 1);
#ifdef OPENMVG_USE_OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
        for (int64_t v = 0; v < (int64_t)n_views; ++v) {
          const std::shared_ptr<features::Regions> r = regions_provider_->get(slot_view[v]);
          const size_t nb = (size_t)(feat_start[v + 1] - feat_start[v]) * desc_bytes;
          if (nb) std::memcpy(desc.data() + (size_t)feat_start[v] * desc_bytes, r->DescriptorRawData(), nb);
        }
        // The accepted pairs go to the device in batches: the call holds one word per left feature of every pair of its
        // batch, so a batch ends at kGuidedFeaturesPerCall left features (1 GB of device memory; 100 000 pairs of 2 000 features are one
        // call) - the cancellation flag is looked at between two calls, and a failing call sends only ITS pairs to the reference's
        // member function (logged once).
        constexpr uint64_t kGuidedFeaturesPerCall = 1ull << 28;
        const size_t ng_all = guided_pairs.size();
        guided_batch_of.assign(ng_all, 0); guided_pos_in_batch.assign(ng_all, 0);
        for (size_t q0 = 0; q0 < ng_all && !my_progress_bar->hasBeenCanceled();) {
          size_t q1 = q0; uint64_t feats = 0;
          while (q1 < ng_all) {
            const uint32_t vI = pair_views[2 * guided_pairs[q1]];
            const uint64_t nI = feat_start[vI + 1] - feat_start[vI];
            if (q1 > q0 && feats + nI > kGuidedFeaturesPerCall) break;
            feats += nI; ++q1;
          }
          const size_t ng = q1 - q0;
          std::vector<uint32_t> g_views(2 * ng);
          std::vector<double> g_model(9 * ng), g_th(ng);
          for (size_t q = 0; q < ng; ++q) {
            const size_t k = guided_pairs[q0 + q];
            g_views[2 * q] = pair_views[2 * k]; g_views[2 * q + 1] = pair_views[2 * k + 1];
            if constexpr (M::essential) {   // E_ACRobust.hpp:196-197: the epipolar error is taken in pixels, with F = K2^-T E K1^-1
              Mat3 E, K1, K2, F;
              for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
                E(r, c) = res[k].F[3 * r + c]; K1(r, c) = view_K[9 * (size_t)pair_views[2 * k] + 3 * r + c]; K2(r, c) = view_K[9 * (size_t)pair_views[2 * k + 1] + 3 * r + c];
              }
              FundamentalFromEssential(E, K1, K2, &F);
              for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g_model[9 * q + 3 * r + c] = F(r, c);
            } else {
              std::memcpy(g_model.data() + 9 * q, res[k].F, 9 * sizeof(double));
            }
            g_th[q] = Square(res[k].precision_robust);   // (infinity stays infinity: no guided matches, as the functors test)
          }
          guided_batches.emplace_back();
          GuidedBatch& gb = guided_batches.back();
          gb.start.assign(ng + 1, 0);
          const bool inj = mvgx_adapter::injected("geofilter", "guided");
          const int rc = inj ? MVGX_ERR_NODEV
                             : mvgx_guided_match(-1, feat_xy.data(), desc.data(), desc_type, desc_bytes, feat_start.data(), (uint32_t)n_views, g_views.data(),
                                                 g_model.data(), g_th.data(), (uint64_t)ng,
                                                 std::is_same<Functor, GeometricFilter_HMatrix_AC>::value ? MVGX_GUIDED_HOMOGRAPHY : MVGX_GUIDED_FUNDAMENTAL,
                                                 Square(d_distance_ratio), gb.start.data(), &gb.ij, nullptr);
          if (rc == MVGX_OK) {
            for (size_t q = 0; q < ng; ++q) {
              guided_done[guided_pairs[q0 + q]] = 1;
              guided_batch_of[q0 + q] = (uint32_t)(guided_batches.size() - 1); guided_pos_in_batch[q0 + q] = (uint32_t)q;
            }
            mvgx_adapter::counters().guided_device_pairs.fetch_add(ng);
          } else {
            // logged once; the pairs of this batch take the reference's own Geometry_guided_matching below (or the failure is thrown)
            mvgx_adapter::device_failure(mvgx_adapter::kGeofilter, "geometric filter", "mvgx_guided_match", rc, inj);
          }
          q0 = q1;
        }
      }
    }
  }
  std::vector<int64_t> guided_index(dev_pairs.size() ? dev_pairs.size() : 1, -1);
  for (size_t q = 0; q < guided_pairs.size(); ++q) guided_index[guided_pairs[q]] = (int64_t)q;
  // results in container order; what is left of the guided matching (host, the reference's code) on OpenMP threads like the reference's loop
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
      if ((size_t)k >= n_dev_done) continue;   // (cancelled between two device calls)
      ok = res[k].ok != 0;
      if (ok) {
        inliers.reserve(res[k].n_inliers);
        const uint64_t lo = start[k];
        if constexpr (M::angular) {
          // second stage of the functor (E_ACRobust_Angular.hpp:126-151): the relative pose whose triangulated inliers lie in front of both
          // cameras, with the reference's own RelativePoseFromEssential; its inliers are the geometric matches
          const size_t n = kv.second.size();
          const double* bI = feat_bearing.data() + 3 * feat_start[pair_views[2 * k]];
          const double* bJ = feat_bearing.data() + 3 * feat_start[pair_views[2 * k + 1]];
          Mat3X x1(3, n), x2(3, n);
          std::vector<uint32_t> vec_inliers;
          for (size_t i = 0; i < n; ++i) {
            for (int c = 0; c < 3; ++c) { x1(c, i) = bI[3 * (size_t)kv.second[i].i_ + c]; x2(c, i) = bJ[3 * (size_t)kv.second[i].j_ + c]; }
            if (mask[lo + i]) vec_inliers.push_back((uint32_t)i);
          }
          Mat3 E;
          for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) E(r, c) = res[k].F[3 * r + c];
          geometry::Pose3 relative_pose;
          std::vector<uint32_t> inliers_indexes;
          std::vector<Vec3> inliers_X;
          if (RelativePoseFromEssential(x1, x2, E, vec_inliers, &relative_pose, &inliers_indexes, &inliers_X)) vec_inliers.swap(inliers_indexes);
          else vec_inliers.clear();
          ok = vec_inliers.size() > M::min_samples * 2.5;
          if (ok) for (const uint32_t i : vec_inliers) inliers.push_back(kv.second[i]);
        } else {
          for (size_t i = 0; i < kv.second.size(); ++i)
            if (mask[lo + i]) inliers.push_back(kv.second[i]);
        }
        if (ok && b_guided_matching && guided_done[k]) {
          const int64_t q = guided_index[k];
          const GuidedBatch& gb = guided_batches[guided_batch_of[q]];
          const uint32_t qb = guided_pos_in_batch[q];
          IndMatches g;
          g.reserve(gb.start[qb + 1] - gb.start[qb]);
          for (uint64_t e = gb.start[qb]; e < gb.start[qb + 1]; ++e) g.emplace_back(gb.ij[2 * e], gb.ij[2 * e + 1]);
          std::swap(inliers, g);
        } else if (ok && b_guided_matching) {
          mvgx_adapter::counters().guided_host_pairs.fetch_add(1);
          Functor f = functor;
          for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) ModelOf<Functor>::model(f)(r, c) = res[k].F[3 * r + c];
          M::set_robust_precision(f, res[k].precision_robust);
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
}  // namespace

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_FMatrix_AC>(
    const GeometricFilter_FMatrix_AC& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* my_progress_bar) {
  filter_container(sfm_data_, regions_provider_, _map_GeometricMatches, functor, putative_matches, b_guided_matching, d_distance_ratio, my_progress_bar);
}

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_HMatrix_AC>(
    const GeometricFilter_HMatrix_AC& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* my_progress_bar) {
  filter_container(sfm_data_, regions_provider_, _map_GeometricMatches, functor, putative_matches, b_guided_matching, d_distance_ratio, my_progress_bar);
}

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_EMatrix_AC>(
    const GeometricFilter_EMatrix_AC& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* my_progress_bar) {
  filter_container(sfm_data_, regions_provider_, _map_GeometricMatches, functor, putative_matches, b_guided_matching, d_distance_ratio, my_progress_bar);
}

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_ESphericalMatrix_AC_Angular<false>>(
    const GeometricFilter_ESphericalMatrix_AC_Angular<false>& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* my_progress_bar) {
  filter_container(sfm_data_, regions_provider_, _map_GeometricMatches, functor, putative_matches, b_guided_matching, d_distance_ratio, my_progress_bar);
}

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_ESphericalMatrix_AC_Angular<true>>(
    const GeometricFilter_ESphericalMatrix_AC_Angular<true>& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* my_progress_bar) {
  filter_container(sfm_data_, regions_provider_, _map_GeometricMatches, functor, putative_matches, b_guided_matching, d_distance_ratio, my_progress_bar);
}

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_EOMatrix_RA>(
    const GeometricFilter_EOMatrix_RA& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* my_progress_bar) {
  filter_container(sfm_data_, regions_provider_, _map_GeometricMatches, functor, putative_matches, b_guided_matching, d_distance_ratio, my_progress_bar);
}

}  // namespace matching_image_collection
}  // namespace openMVG
