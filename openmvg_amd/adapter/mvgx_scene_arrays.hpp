// mvgx_scene_arrays.hpp - what the replacement TUs of the BA path share (mvgx_bundle_adjustment.cpp, mvgx_outlier_filters.cpp):
// the SfM_Data -> flat arrays walk on the library's host workers, the per-thread store of those arrays, and the process-wide slot
// that keeps the last BA context idle between calls. Header-only (inline functions with function-local statics: one instance
// per linked image).
#pragma once

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "openMVG/sfm/sfm_data.hpp"
#include "openMVG/sfm/sfm_landmark.hpp"
#include "openMVG/sfm/sfm_view.hpp"
#include "openMVG/types.hpp"

#include "mvgx.h"

namespace mvgx_adapter {

using openMVG::IndexT;
using openMVG::sfm::Landmark;
using openMVG::sfm::Landmarks;
using openMVG::sfm::SfM_Data;
using openMVG::sfm::View;

// f(item, worker) for item in [0, n) on the library's host workers. An exception out of f (the reference's own .at() in a
// borderline re-evaluation, a bad_alloc) must not unwind through the C frames of the library or end a worker thread: the first one
// is carried back and rethrown on the calling thread, the remaining items are skipped.
template <class F>
void host_parallel(uint64_t n, F&& f) {
  using Fn = typename std::remove_reference<F>::type;
  if (n <= 1) { for (uint64_t i = 0; i < n; ++i) f(i, 0u); return; }
  struct Job { Fn* f; std::atomic<bool> failed{false}; std::mutex mu; std::exception_ptr error; } job;
  job.f = &f;
  mvgx_host_parallel_for(n, 0, [](void* u, uint64_t i, unsigned w) {
    Job& j = *static_cast<Job*>(u);
    if (j.failed.load(std::memory_order_relaxed)) return;
    try { (*j.f)(i, w); }
    catch (...) {
      std::lock_guard<std::mutex> lock(j.mu);
      if (!j.error) j.error = std::current_exception();
      j.failed.store(true);
    }
  }, &job);
  if (job.error) std::rethrow_exception(job.error);
}

// The flattened scene of a call (see Adjust): one store per calling thread, capacity kept between calls.
struct FlatScene {
  std::vector<double> poses, intrinsics, points, obs_xy;
  std::vector<int32_t> intr_model;
  std::vector<uint8_t> pose_mask, intr_mask;
  std::vector<uint32_t> obs_pose, obs_intr, obs_point;
  std::vector<IndexT> pose_ids, intr_ids;   // dense index -> key in SfM_Data::poses / ::intrinsics
  std::vector<Landmark*> lm_of_point;
  std::vector<IndexT> lm_key;   // the landmark's key in SfM_Data::structure
  std::vector<uint64_t> obs_first;   // [point]: index of its first observation row
  std::vector<double> scratch;  // per-observation / per-point results of the outlier filters
};
inline FlatScene& flat_scene() {
  static thread_local FlatScene fs;
  return fs;
}

// The context of the last Adjust() of this process, kept idle between calls. The engines construct a Bundle_Adjustment_Ceres on the
// stack per call (sequential_SfM.cpp:1194-1210, global_SfM.cpp:379-446), so nothing of the object survives; what repeats is the
// scene. The next call (Adjust(), or one of the two outlier filters) offers its arrays to the kept context:
//   same structure (global_SfM.cpp's refinement passes with growing parameter sets; a rejection round that removed nothing; the
//     filters right after Adjust())                                     -> mvgx_ba_update: values only;
//   the same scene MINUS observations / tracks (what RemoveOutliers_* leave behind: the `do { BA } while (reject)` loop of
//     sequential_SfM.cpp:1190-1232)                                     -> mvgx_ba_update_subset: the kept structure with those
//     observations switched off - found by walking the kept arrays and the new ones side by side (erasing from an unordered_map
//     keeps the order of what stays);
//   anything else (views or tracks added: a resection)                  -> the context is destroyed and a new one created.
// A context taken out of the slot belongs to the calling thread; concurrent calls find the slot empty and create their own.
// MVGX_BA_CONTEXT_CACHE=0 turns this off (every call creates and destroys); mvgx_adapter_ba_release_context() frees the idle
// context at any time.
struct KeptStructure {   // the arrays the kept context was created from
  std::vector<IndexT> pose_ids, intr_ids, lm_key;
  std::vector<int32_t> intr_model;
  std::vector<uint32_t> obs_pose, obs_intr, obs_point;
  std::vector<uint64_t> obs_first;    // [point]: its first observation
  std::vector<double> points, obs_xy; // values of that call (what a switched-off observation / an absent point keeps)
  bool plain = false;                 // neither control points nor pose priors: the only kind the subset route takes
  void take_from(FlatScene& fs, bool is_plain) {   // (swaps: both sides keep their capacity)
    pose_ids.swap(fs.pose_ids); intr_ids.swap(fs.intr_ids); lm_key.swap(fs.lm_key); intr_model.swap(fs.intr_model);
    obs_pose.swap(fs.obs_pose); obs_intr.swap(fs.obs_intr); obs_point.swap(fs.obs_point); obs_first.swap(fs.obs_first);
    points.swap(fs.points); obs_xy.swap(fs.obs_xy);
    plain = is_plain;
  }
};
struct ContextCache {
  std::mutex mu;
  mvgx_ba_ctx* idle = nullptr;
  KeptStructure* kept = nullptr;
  int device = 0;
  std::atomic<uint64_t> created{0}, reused{0}, subset{0};
};
inline ContextCache& context_cache() {
  static ContextCache* c = new ContextCache;   // never destroyed: the HIP runtime may be gone when static destructors run
  return *c;
}
inline bool context_cache_enabled() {
  const char* env = std::getenv("MVGX_BA_CONTEXT_CACHE");
  return !(env && env[0] == '0');
}
// the idle context and its structure arrays leave the slot together (nullptr: nothing usable is kept). kAnyDevice (the outlier
// filters, which have no device option of their own): whatever device the kept context lives on - *device receives it.
constexpr int kAnyDevice = -1000000;
inline mvgx_ba_ctx* take_idle_context(int device, KeptStructure** kept = nullptr, int* device_of_context = nullptr) {
  ContextCache& c = context_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  mvgx_ba_ctx* ctx = c.idle;
  KeptStructure* ks = c.kept;
  c.idle = nullptr; c.kept = nullptr;
  if (ctx && device == kAnyDevice) device = c.device;
  if (device_of_context) *device_of_context = device == kAnyDevice ? -1 : device;
  if (ctx && (c.device != device || !context_cache_enabled())) { mvgx_ba_destroy(ctx); ctx = nullptr; }
  if (!ctx || !kept) { delete ks; ks = nullptr; }
  if (kept) *kept = ks;
  return ctx;
}
inline void keep_idle_context(mvgx_ba_ctx* ctx, int device, KeptStructure* kept = nullptr) {
  if (!context_cache_enabled()) { mvgx_ba_destroy(ctx); delete kept; return; }
  ContextCache& c = context_cache();
  mvgx_ba_ctx* old = nullptr;
  KeptStructure* old_kept = nullptr;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    old = c.idle; old_kept = c.kept;
    c.idle = ctx; c.kept = kept;
    c.device = device;
  }
  if (old) mvgx_ba_destroy(old);
  delete old_kept;
}

// Is the scene in `fs` the scene of `ks` minus observations / whole tracks? Both lists are in the walk's order (bucket order of
// the landmark map, each landmark's own observation order): erasing elements of an unordered_map does not move the ones that
// stay, so the new lists are subsequences of the kept ones exactly when only erasures happened. On success: enabled[k_old] = 1
// for the kept observations that are still there, point_old[j_new] / obs_old[k_new] = their indices in the kept arrays.
inline bool express_in_kept_structure(const KeptStructure& ks, const FlatScene& fs, std::vector<uint8_t>& enabled,
                                      std::vector<uint32_t>& point_old, std::vector<uint64_t>& obs_old) {
  if (ks.pose_ids != fs.pose_ids || ks.intr_ids != fs.intr_ids || ks.intr_model != fs.intr_model) return false;
  const size_t n_old = ks.lm_key.size(), n_new = fs.lm_key.size();
  const uint64_t no_old = ks.obs_pose.size(), no_new = fs.obs_pose.size();
  if (n_new > n_old || no_new > no_old) return false;
  enabled.assign(std::max<size_t>(no_old, 1), 0);
  point_old.resize(n_new);
  obs_old.resize(no_new);
  // the points on this thread (one pass over the two key lists), the observations of the matched points on the host workers
  size_t jo = 0;
  for (size_t jn = 0; jn < n_new; ++jn, ++jo) {
    while (jo < n_old && ks.lm_key[jo] != fs.lm_key[jn]) ++jo;
    if (jo == n_old) return false;
    point_old[jn] = static_cast<uint32_t>(jo);
  }
  std::atomic<int> failed{0};
  const size_t per = 4096;
  host_parallel((n_new + per - 1) / per, [&](uint64_t g, unsigned) {
    for (size_t jn = g * per, e = std::min(n_new, (g + 1) * per); jn < e; ++jn) {
      const size_t jp = point_old[jn];
      uint64_t ko = ks.obs_first[jp];
      const uint64_t ko_end = jp + 1 < n_old ? ks.obs_first[jp + 1] : no_old;
      const uint64_t kn_end = jn + 1 < n_new ? fs.obs_first[jn + 1] : no_new;
      for (uint64_t kn = fs.obs_first[jn]; kn < kn_end; ++kn, ++ko) {
        while (ko < ko_end && !(ks.obs_pose[ko] == fs.obs_pose[kn] && ks.obs_intr[ko] == fs.obs_intr[kn])) ++ko;
        if (ko == ko_end) { failed.store(1); return; }
        enabled[ko] = 1;
        obs_old[kn] = ko;
      }
    }
  });
  return failed.load() == 0;
}

// A device context bound to the scene in `fs` / `prob` by the cheapest of the three routes above.
struct BoundContext {
  mvgx_ba_ctx* ctx = nullptr;
  KeptStructure* kept = nullptr;     // the structure arrays that go back into the slot with the context (subset route: the old ones)
  int device = -1;                   // where the context lives (what it goes back into the slot with)
  bool subset = false;               // the context holds the KEPT structure: parameters / residuals / angles come back in its indexing
  std::vector<uint32_t> point_old;   // subset: [new point] -> kept point
  std::vector<uint64_t> obs_old;     // subset: [new observation] -> kept observation
  std::vector<double> points_old;    // subset: the kept-indexed point array handed to the library (read_params target)
  const char* route = "";
};
// rc of the library call that decided (MVGX_OK: bc.ctx is ready). `plain`: prob has neither control points nor priors.
// linear_solver: MVGX_BA_LINEAR_SOLVER_* of the caller's options, or kAnySolver (the outlier filters never solve). A kept context
// whose reduced solver is set up as the other kind is not re-used.
constexpr int kAnySolver = -1;
inline int bind_context(int device, const mvgx_ba_problem& prob, FlatScene& fs, bool plain, BoundContext& bc, int linear_solver = kAnySolver) {
  KeptStructure* ks = nullptr;
  mvgx_ba_ctx* ctx = take_idle_context(device, &ks, &device);   // (kAnyDevice becomes the kept context's device, or -1)
  bc.device = device;
  int rc = MVGX_ERR_STRUCTURE;
  if (ctx && linear_solver != kAnySolver && mvgx_ba_set_linear_solver(ctx, linear_solver) != MVGX_OK) {
    mvgx_ba_destroy(ctx); ctx = nullptr;
  }
  if (ctx) {
    // (a scene with other counts than the kept structure cannot have its structure - no need to fingerprint a million observations
    // to learn that; in the reject loop every call after the first erasure takes this way straight to the subset route)
    const bool counts_differ = ks && (prob.n_obs != ks->obs_pose.size() || prob.n_points != ks->lm_key.size());
    rc = counts_differ ? MVGX_ERR_STRUCTURE : mvgx_ba_update(ctx, &prob);
    if (rc == MVGX_OK) {
      context_cache().reused.fetch_add(1);
      bc.route = "mvgx_ba_update (context kept)";
    } else if (rc == MVGX_ERR_STRUCTURE && plain && ks && ks->plain) {
      std::vector<uint8_t> enabled;
      if (express_in_kept_structure(*ks, fs, enabled, bc.point_old, bc.obs_old)) {
        // the kept structure with this call's values: cameras and masks as given, points / image points where they still exist
        bc.points_old = ks->points;
        {
          const size_t n_new = bc.point_old.size(), no_new = bc.obs_old.size(), per = 16384;
          host_parallel((n_new + per - 1) / per, [&](uint64_t g, unsigned) {
            for (size_t jn = g * per, e = std::min(n_new, (g + 1) * per); jn < e; ++jn)
              for (int a = 0; a < 3; ++a) bc.points_old[3 * static_cast<size_t>(bc.point_old[jn]) + a] = fs.points[3 * jn + a];
          });
          host_parallel((no_new + per - 1) / per, [&](uint64_t g, unsigned) {
            for (size_t kn = g * per, e = std::min(no_new, (g + 1) * per); kn < e; ++kn) {
              ks->obs_xy[2 * bc.obs_old[kn]] = fs.obs_xy[2 * kn];
              ks->obs_xy[2 * bc.obs_old[kn] + 1] = fs.obs_xy[2 * kn + 1];
            }
          });
        }
        mvgx_ba_problem old = prob;
        old.n_points = static_cast<uint32_t>(ks->lm_key.size());
        old.n_obs = ks->obs_pose.size();
        old.points = bc.points_old.data();
        old.obs_pose = ks->obs_pose.data(); old.obs_intr = ks->obs_intr.data(); old.obs_point = ks->obs_point.data();
        old.obs_xy = ks->obs_xy.data();
        rc = mvgx_ba_update_subset(ctx, &old, enabled.data());
        if (rc == MVGX_OK) {
          context_cache().subset.fetch_add(1);
          bc.subset = true;
          bc.route = "mvgx_ba_update_subset (context kept, observations off)";
        }
      }
    }
    if (rc != MVGX_OK) { mvgx_ba_destroy(ctx); ctx = nullptr; }
  }
  if (!ctx) {
    delete ks; ks = nullptr;
    rc = mvgx_ba_create(device, &prob, &ctx);
    if (rc == MVGX_OK && linear_solver != kAnySolver && (rc = mvgx_ba_set_linear_solver(ctx, linear_solver)) != MVGX_OK) mvgx_ba_destroy(ctx);
    if (rc == MVGX_OK) { context_cache().created.fetch_add(1); bc.route = "mvgx_ba_create"; }
    else ctx = nullptr;
  }
  bc.ctx = ctx;
  bc.kept = ks;
  return rc;
}
// after the caller's last use of the context: back into the slot with the structure arrays it was built from
inline void release_bound_context(BoundContext& bc, FlatScene& fs, bool plain, bool healthy) {
  if (!bc.ctx) return;
  const int device = bc.device;
  if (!healthy) { mvgx_ba_destroy(bc.ctx); delete bc.kept; bc.ctx = nullptr; bc.kept = nullptr; return; }
  if (!bc.subset) {   // created from / re-bound to the arrays in fs: they become the kept ones
    if (!bc.kept) bc.kept = new KeptStructure;
    bc.kept->take_from(fs, plain);
  }
  keep_idle_context(bc.ctx, device, bc.kept);
  bc.ctx = nullptr; bc.kept = nullptr;
}

// Landmarks and Observations are std::unordered_map (types.hpp:67): walking one is a chain of dependent loads, one node per
// element - about a million nodes at 200 views. The walk therefore runs on the library's host workers (mvgx_host_parallel_for),
// split by BUCKET ranges of the landmark map: pass 1 counts the landmarks and observations of every range, a prefix sum fixes
// where each range writes, pass 2 fills the landmark pointers / keys / coordinates and the per-observation rows. Points are
// numbered in bucket order (a function of the container alone, like the reference's iteration order - which the reference itself
// calls unspecified); within a landmark the observations keep the order of its own map.
// (Round 3: pointers on one thread, rows on 16 threads started per call: 5.1 ms at 200 views / 1 M observations; now 1.6 ms.)
// Returns the number of observations; *error = 1: an observation of a view without usable intrinsic, 2: of a view without pose /
// of an unknown view (the arrays are then incomplete).
template <class Tick>
uint64_t flatten_observations(SfM_Data& sfm_data, const std::unordered_map<IndexT, uint32_t>& pose_idx,
                              const std::unordered_map<IndexT, uint32_t>& intr_idx, FlatScene& fs, int* error, Tick&& tick) {
  std::vector<double>&points = fs.points, &obs_xy = fs.obs_xy;
  std::vector<uint32_t>&obs_pose = fs.obs_pose, &obs_intr = fs.obs_intr, &obs_point = fs.obs_point;
  Landmarks& structure = sfm_data.structure;
  struct ViewBlocks { uint32_t pose, intr; bool has_pose, has_intr; };
  std::unordered_map<IndexT, ViewBlocks> view_blocks;
  view_blocks.reserve(sfm_data.views.size());
  for (const auto& v : sfm_data.views) {
    const View* view = v.second.get();
    const auto pi = pose_idx.find(view->id_pose);
    const auto ii = intr_idx.find(view->id_intrinsic);
    view_blocks.emplace(v.first, ViewBlocks{pi == pose_idx.end() ? 0u : pi->second, ii == intr_idx.end() ? 0u : ii->second, pi != pose_idx.end(),
                                            ii != intr_idx.end()});
  }
  tick("  view blocks");
  const size_t n_buckets = structure.bucket_count();
  const size_t n_ranges = structure.size() < 4096 ? 1 : std::min<size_t>(512, structure.size() / 512);   // (several per worker: the ranges are uneven)
  std::vector<uint64_t> range_lm(n_ranges + 1, 0), range_obs(n_ranges + 1, 0);
  auto bucket_lo = [&](size_t r) { return n_buckets * r / n_ranges; };
  host_parallel(n_ranges, [&](uint64_t r, unsigned) {
    uint64_t n_lm = 0, n_ob = 0;
    for (size_t b = bucket_lo(r), be = bucket_lo(r + 1); b < be; ++b)
      for (auto it = structure.begin(b), e = structure.end(b); it != e; ++it) { ++n_lm; n_ob += it->second.obs.size(); }
    range_lm[r + 1] = n_lm; range_obs[r + 1] = n_ob;
  });
  for (size_t r = 0; r < n_ranges; ++r) { range_lm[r + 1] += range_lm[r]; range_obs[r + 1] += range_obs[r]; }
  const uint64_t n_structure_obs64 = range_obs[n_ranges];
  tick("  landmark / observation counts");
  std::vector<Landmark*>& lm_of_point = fs.lm_of_point;
  lm_of_point.resize(range_lm[n_ranges]);
  fs.lm_key.resize(range_lm[n_ranges]);
  fs.obs_first.resize(range_lm[n_ranges]);
  points.resize(lm_of_point.size() * 3);
  obs_pose.resize(n_structure_obs64); obs_intr.resize(n_structure_obs64); obs_point.resize(n_structure_obs64);
  obs_xy.resize(2 * n_structure_obs64);
  std::atomic<int> flatten_error{0};
  host_parallel(n_ranges, [&](uint64_t r, unsigned) {
    uint64_t j = range_lm[r], k = range_obs[r];
    for (size_t b = bucket_lo(r), be = bucket_lo(r + 1); b < be; ++b)
      for (auto it = structure.begin(b), e = structure.end(b); it != e; ++it, ++j) {
        Landmark& lm = it->second;
        lm_of_point[j] = &lm;
        fs.lm_key[j] = it->first;
        fs.obs_first[j] = k;
        points[3 * j] = lm.X(0); points[3 * j + 1] = lm.X(1); points[3 * j + 2] = lm.X(2);
        for (const auto& ob : lm.obs) {
          const auto vb = view_blocks.find(ob.first);
          if (vb == view_blocks.end()) { flatten_error = 2; return; }                       // views.at(...) of the serial walk
          if (!vb->second.has_intr) { int none = 0; flatten_error.compare_exchange_strong(none, 1); return; }   // its intrinsic test comes first
          if (!vb->second.has_pose) { flatten_error = 2; return; }                          // pose_idx.at(...)
          obs_pose[k] = vb->second.pose;
          obs_intr[k] = vb->second.intr;
          obs_point[k] = static_cast<uint32_t>(j);
          obs_xy[2 * k] = ob.second.x(0);
          obs_xy[2 * k + 1] = ob.second.x(1);
          ++k;
        }
      }
  });
  *error = flatten_error.load();
  return n_structure_obs64;
}

}  // namespace mvgx_adapter
