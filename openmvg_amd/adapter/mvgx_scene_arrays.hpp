// mvgx_scene_arrays.hpp - what the replacement TUs of the BA path share (mvgx_bundle_adjustment.cpp, mvgx_outlier_filters.cpp):
// the SfM_Data -> flat arrays walk on the library's host workers, the per-thread store of those arrays, and the process-wide slot
// that keeps the last BA context idle between calls. Header-only (inline functions with function-local statics: one instance
// per linked image).
#pragma once

#include <atomic>
#include <cstdint>
#include <cstdlib>
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

// f(item, worker) for item in [0, n) on the library's host workers
template <class F>
void host_parallel(uint64_t n, F&& f) {
  using Fn = typename std::remove_reference<F>::type;
  if (n <= 1) { for (uint64_t i = 0; i < n; ++i) f(i, 0u); return; }
  mvgx_host_parallel_for(n, 0, [](void* u, uint64_t i, unsigned w) { (*static_cast<Fn*>(u))(i, w); }, &f);
}

// The flattened scene of a call (see Adjust): one store per calling thread, capacity kept between calls.
struct FlatScene {
  std::vector<double> poses, intrinsics, points, obs_xy;
  std::vector<int32_t> intr_model;
  std::vector<uint8_t> pose_mask, intr_mask;
  std::vector<uint32_t> obs_pose, obs_intr, obs_point;
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
// scene: global_SfM.cpp refines the same structure three times with growing parameter sets, sequential_SfM.cpp:1190-1232 calls
// Adjust again whenever its rejection step removed nothing, and callers re-run BA after changing options. The next call offers
// its arrays to the kept context (mvgx_ba_update): same structure -> only values are uploaded (the host structure build, the
// device allocations and the symbolic phase of the reduced solve are skipped); another structure -> the context is destroyed and
// a new one created, as before. A context taken out of the cache belongs to the calling thread; concurrent Adjust() calls simply
// find the cache empty. MVGX_BA_CONTEXT_CACHE=0 turns this off (every call creates and destroys);
// mvgx_adapter_ba_release_context() hands the idle context's device memory back at any time.
struct ContextCache {
  std::mutex mu;
  mvgx_ba_ctx* idle = nullptr;
  int device = 0;
  std::atomic<uint64_t> created{0}, reused{0};
};
inline ContextCache& context_cache() {
  static ContextCache* c = new ContextCache;   // never destroyed: the HIP runtime may be gone when static destructors run
  return *c;
}
inline bool context_cache_enabled() {
  const char* env = std::getenv("MVGX_BA_CONTEXT_CACHE");
  return !(env && env[0] == '0');
}
inline mvgx_ba_ctx* take_idle_context(int device) {
  ContextCache& c = context_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  mvgx_ba_ctx* ctx = c.idle;
  c.idle = nullptr;
  if (ctx && (c.device != device || !context_cache_enabled())) { mvgx_ba_destroy(ctx); ctx = nullptr; }
  return ctx;
}
inline void keep_idle_context(mvgx_ba_ctx* ctx, int device) {
  if (!context_cache_enabled()) { mvgx_ba_destroy(ctx); return; }
  ContextCache& c = context_cache();
  mvgx_ba_ctx* old = nullptr;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    old = c.idle;
    c.idle = ctx;
    c.device = device;
  }
  if (old) mvgx_ba_destroy(old);
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
