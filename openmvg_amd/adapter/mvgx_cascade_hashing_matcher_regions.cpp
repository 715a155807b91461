// mvgx_cascade_hashing_matcher_regions.cpp — link-time replacement for openMVG's
//   src/openMVG/matching_image_collection/Cascade_Hashing_Matcher_Regions.cpp
// (CASCADE_HASHING_L2: the default nearest-neighbour method of main_ComputeMatches for scalar regions,
//  main_ComputeMatches.cpp:254-258). Same header, same mangled symbols; link libmvgx_hip.so.
//
// Split of the work (reference lines are Cascade_Hashing_Matcher_Regions.cpp):
//   host, with the reference's own library code (so the values are the reference's bit for bit):
//     * the zero-mean descriptor :78-104 (mean over the images of the per-image mean, CascadeHasher::GetZeroMeanDescriptor), the
//       per-image means on all host threads;
//     * the two de-duplication steps :218-226 - IndMatch::getDeduplicated and IndMatchDecorator<float>::getDeduplicated
//       (whose std::set ordering is the library's) - on helper threads, the container is filled from the calling thread;
//   MI355X (mvgx_cascade_*):
//     * the rest of the hashing stage :66-76, :107-131 - the random projections of CascadeHasher::Init and CreateHashedDescriptions
//       per descriptor (mvgx_cascade_hash_regions: Eigen's single-precision product order reproduced, codes and bucket ids
//       bit-identical - tests/test_cascade.py, tests/test_adapter_*.py);
//     * the matching stage :166-215 - bucket candidates, Hamming ranking of the hash codes, exact L2 on the ten best, the two
//       nearest, the distance-ratio test - integer work on the hash outputs, bit-identical lists.
// 128-byte uint8 regions (SIFT), 144-byte uint8 (AKAZE_Liop_Regions) and 64-float regions (AKAZE_Float_Regions) take that route (round 5:
// the hashing stage of the last two as well - mvgx_cascade_hash_regions_typed; MVGX_CASCADE_HASH=host keeps it with CreateHashedDescriptions
// on the host threads and hands the codes over: mvgx_cascade_set_regions_typed); float distances in L2<float>'s summation order. Other
// lengths keep working through the reference's own CascadeHasher::Match_HashedDescriptions on the host.
// A failing device call is logged once and the remaining pairs run through the reference's own classes (mvgx_adapter_policy.hpp).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <stdexcept>
#include <string>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <type_traits>
#include <typeinfo>
#include <unordered_map>
#include <vector>

#include "openMVG/features/feature.hpp"
#include "openMVG/features/regions.hpp"
#include "openMVG/matching/cascade_hasher.hpp"
#include "openMVG/matching/indMatch.hpp"
#include "openMVG/matching/indMatchDecoratorXY.hpp"
#include "openMVG/matching/matching_filters.hpp"
#include "openMVG/matching_image_collection/Cascade_Hashing_Matcher_Regions.hpp"
#include "openMVG/numeric/numeric.h"
#include "openMVG/sfm/pipelines/sfm_regions_provider.hpp"
#include "openMVG/system/logger.hpp"
#include "openMVG/system/progressinterface.hpp"
#include "openMVG/types.hpp"

#include "mvgx.h"
#include "mvgx_adapter_policy.hpp"

namespace openMVG {
namespace matching_image_collection {

namespace {

// MVGX_CASCADE_HASH=check, last call: 0 not run, 1 the device's codes equal this build's CascadeHasher's on the probed view, 2 they differ
// (tests read it through mvgx_adapter_cascade_last_hash_check)
std::atomic<int> g_last_hash_check{0};
constexpr uint64_t kPairsPerCall = 1u << 16;   // cancellation / progress granularity of the device route

template <class F>
void on_host_threads(size_t n, F f) {   // f(index) for every index, indices handed out dynamically
  // (MVGX_ADAPTER_THREADS sets another limit than 32: the std::set work of the de-duplication is bound by the allocator - 128 threads
  // were measured 30 % slower than 32 on a 256-thread host, call r5_34)
  static const unsigned limit = [] { const char* e = std::getenv("MVGX_ADAPTER_THREADS"); const int v = e ? std::atoi(e) : 32; return (unsigned)std::max(1, v); }();
  const unsigned threads = (unsigned)std::min<size_t>(n, std::max(1u, std::min(limit, std::thread::hardware_concurrency())));
  if (threads <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
  std::atomic<size_t> next{0};
  auto body = [&]() { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < threads; ++t) pool.emplace_back(body);
  body();
  for (auto& t : pool) t.join();
}

// everything the matching stage needs to know about one view
template <typename ScalarT>
struct View {
  std::shared_ptr<features::Regions> regions;
  matching::HashedDescriptions hashed;
  std::vector<features::PointFeature> positions;
  size_t count() const { return regions ? regions->RegionCount() : 0; }
  const ScalarT* rows() const { return reinterpret_cast<const ScalarT*>(regions->DescriptorRawData()); }
};

// the reference's last two steps on one pair's putative list (entries (index in I, index in J))
void deduplicate(matching::IndMatches& v, const std::vector<features::PointFeature>& posI, const std::vector<features::PointFeature>& posJ) {
  // IndMatch::getDeduplicated = the list through a std::set<IndMatch> (ordered by (i, j)) and back: the identity on a strictly
  // increasing list - what the device delivers (one entry per query row, in row order) - so the set is only built when it is not
  bool increasing = true;
  for (size_t k = 1; k < v.size() && increasing; ++k) increasing = v[k - 1] < v[k];
  if (!increasing) matching::IndMatch::getDeduplicated(v);
  matching::IndMatchDecorator<float> by_position(v, posI, posJ);
  by_position.getDeduplicated(v);
}

template <typename ScalarT>
void match_collection(const sfm::Regions_Provider& provider, const Pair_Set& pairs, float dist_ratio,
                      matching::PairWiseMatchesContainer& out, system::ProgressInterface* progress) {
  using RowMajor = Eigen::Matrix<ScalarT, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
  progress->Restart(pairs.size(), "- Matching -");
  // views in ascending id = the row order of the zero-mean matrix (:88-108)
  std::map<IndexT, View<ScalarT>> views;
  for (const Pair& p : pairs) { views[p.first]; views[p.second]; }
  if (views.empty()) return;
  for (auto& kv : views) kv.second.regions = provider.get(kv.first);
  const size_t dimension = views.begin()->second.regions->DescriptorLength();
  matching::CascadeHasher hasher;
  hasher.Init(dimension);
  // the device covers both stages for the shapes openMVG's scalar describers produce: 128-byte uint8 (SIFT_Regions), 144-byte uint8
  // (AKAZE_Liop_Regions) and 64-float rows (AKAZE_Float_Regions; round 5: their hashing stage too - mvgx_cascade_hash_regions_typed)
  constexpr bool is_float = std::is_same<ScalarT, float>::value;
  const bool on_device = (std::is_same<ScalarT, unsigned char>::value && (dimension == 128 || dimension == 144)) || (is_float && dimension == 64);
  const bool device_hashing = on_device;
  std::vector<View<ScalarT>*> order;
  for (auto& kv : views) order.push_back(&kv.second);
  // the zero-mean descriptor (:78-104): the reference's own GetZeroMeanDescriptor, per view on the host threads, then over the views
  Eigen::VectorXf zero_mean;
  {
    Eigen::MatrixXf per_view(views.size(), dimension);
    per_view.fill(0.0f);
    on_host_threads(order.size(), [&](size_t k) {
      View<ScalarT>& v = *order[k];
      if (v.count() > 0) {
        Eigen::Map<RowMajor> m(const_cast<ScalarT*>(v.rows()), v.count(), dimension);
        per_view.row(k) = matching::CascadeHasher::GetZeroMeanDescriptor(m);   // (rows of a column-major matrix: disjoint elements)
      }
    });
    zero_mean = matching::CascadeHasher::GetZeroMeanDescriptor(per_view);
  }
  // hash codes and bucket ids: on the device for 128-byte uint8 regions (mvgx_cascade_hash_regions below, bit-identical to
  // CreateHashedDescriptions), else the reference's CreateHashedDescriptions on the host threads
  on_host_threads(order.size(), [&](size_t k) {
    View<ScalarT>& v = *order[k];
    if (!device_hashing) {
      Eigen::Map<RowMajor> m(const_cast<ScalarT*>(v.rows()), v.count(), dimension);
      v.hashed = hasher.CreateHashedDescriptions(m, zero_mean);   // const member, per-view outputs: thread safe
    }
    v.positions = v.regions->GetRegionsPositions();
  });
  // MVGX_CASCADE_HASH=host: the hashing stage stays with the reference's CascadeHasher on the host threads and only the matching stage
  // runs on the device (mvgx_cascade_set_regions) - for openMVG builds whose Eigen kernels are compiled with FMA (-march=native): the
  // device hashing reproduces the non-FMA operation order (mvgx.h), such a build's own CascadeHasher rounds differently near zero.
  // Default "device"; "check" hashes the first view both ways once and switches to "host" with a warning on a mismatch.
  const char* hash_env = std::getenv("MVGX_CASCADE_HASH");
  bool hash_host = !device_hashing || (hash_env && !std::strcmp(hash_env, "host"));
  const bool hash_check = device_hashing && hash_env && !std::strcmp(hash_env, "check");
  bool hashed_on_host = !device_hashing;   // v.hashed filled (above)

  // pairs in the reference's visiting order (grouped by I, ascending), minus the ones it skips (:151-176)
  std::vector<Pair> todo;
  uint32_t skipped = 0;
  for (const Pair& p : pairs) {
    const View<ScalarT>& vi = views.at(p.first);
    const View<ScalarT>& vj = views.at(p.second);
    if (vi.count() == 0 || vi.regions->Type_id() != vj.regions->Type_id()) { ++skipped; continue; }
    todo.push_back(p);
  }
  if (skipped) (*progress) += skipped;

  // the reference's own matching stage for the pairs todo[first ..), pair by pair on the host threads; the container is filled by this
  // thread. Route of the region types the device does not cover, and - after a device failure - of the pairs it did not deliver.
  auto host_route = [&](size_t first) {
    std::vector<matching::IndMatches> lists(todo.size() - first);
    on_host_threads(todo.size() - first, [&](size_t kk) {
      const size_t k = first + kk;
      if (progress->hasBeenCanceled()) return;
      const View<ScalarT>& vi = views.at(todo[k].first);
      const View<ScalarT>& vj = views.at(todo[k].second);
      using DistanceT = typename Accumulator<ScalarT>::Type;
      Eigen::Map<RowMajor> mI(const_cast<ScalarT*>(vi.rows()), vi.count(), dimension), mJ(const_cast<ScalarT*>(vj.rows()), vj.count(), dimension);
      matching::IndMatches nn;
      std::vector<DistanceT> dist;
      hasher.template Match_HashedDescriptions<RowMajor, DistanceT>(vj.hashed, mJ, vi.hashed, mI, &nn, &dist);
      std::vector<int> kept;
      matching::NNdistanceRatio(dist.begin(), dist.end(), 2, kept, Square(dist_ratio));
      matching::IndMatches& v = lists[kk];
      for (int q : kept) v.emplace_back(nn[q * 2].j_, nn[q * 2].i_);
      deduplicate(v, vi.positions, vj.positions);
    });
    for (size_t k = first; k < todo.size(); ++k) {
      if (!lists[k - first].empty()) out.insert({todo[k], std::move(lists[k - first])});
      ++(*progress);
    }
  };
  // hash codes and bucket ids of every view by the reference's own class (the non-device types; MVGX_CASCADE_HASH=host; fallback)
  auto hash_on_host = [&]() {
    on_host_threads(order.size(), [&](size_t k) {
      View<ScalarT>& v = *order[k];
      Eigen::Map<RowMajor> m(const_cast<ScalarT*>(v.rows()), v.count(), dimension);
      v.hashed = hasher.CreateHashedDescriptions(m, zero_mean);   // const member, per-view outputs: thread safe
    });
  };
  if (!on_device) {
    host_route(0);
    return;
  }

  // ---- device route: dense view numbering; hashing and matching stages both on the device ----
  std::unordered_map<IndexT, uint32_t> dense;
  std::vector<const uint8_t*> rows;
  std::vector<uint32_t> n_desc;
  std::vector<const View<ScalarT>*> view_of;
  for (auto& kv : views) {
    const View<ScalarT>& v = kv.second;
    dense[kv.first] = (uint32_t)rows.size();
    view_of.push_back(&v);
    const size_t n = v.count();
    n_desc.push_back((uint32_t)n);
    rows.push_back(n ? reinterpret_cast<const uint8_t*>(v.regions->DescriptorRawData()) : nullptr);
  }
  std::vector<uint32_t> dev_pairs;
  for (const Pair& p : todo) { dev_pairs.push_back(dense[p.first]); dev_pairs.push_back(dense[p.second]); }

  struct Ctx { mvgx_cascade_ctx* c = nullptr; ~Ctx() { if (c) mvgx_cascade_destroy(c); } } ctx;
  // Error convention (mvgx_adapter_policy.hpp): a failing device call is logged once and the pairs not yet delivered run through the
  // reference's own hashing + matching classes above (or the failure is thrown, MVGX_ON_DEVICE_ERROR=throw).
  using mvgx_adapter::injected;
  bool failed = false;
  uint64_t delivered = 0;
  auto step = [&](const char* stage, int rc_call, bool inj) {
    if (!inj && rc_call == MVGX_OK) return true;
    mvgx_adapter::device_failure(mvgx_adapter::kCascade, "cascade hashing", stage, inj ? MVGX_ERR_NODEV : rc_call, inj);
    failed = true;
    return false;
  };
  // host-side hash outputs in the layout of mvgx_cascade_set_regions_typed (codes: one bit per dimension, bucket ids: 6 x uint16 per descriptor)
  const size_t code_bytes = (dimension + 7) / 8;
  std::vector<std::vector<uint8_t>> codes;
  std::vector<std::vector<uint16_t>> buckets;
  std::vector<const uint8_t*> code_ptr;
  std::vector<const uint16_t*> bucket_ptr;
  auto pack_host_hashes = [&]() {
    codes.clear(); buckets.clear(); code_ptr.clear(); bucket_ptr.clear();
    for (const View<ScalarT>* v : view_of) {
      const size_t n = v->count();
      codes.emplace_back(n * code_bytes);
      buckets.emplace_back(n * 6);
      for (size_t r = 0; r < n; ++r) {
        const matching::HashedDescription& h = v->hashed.hashed_desc[r];
        std::memcpy(&codes.back()[r * code_bytes], h.hash_code.data(), code_bytes);
        for (int g = 0; g < 6; ++g) buckets.back()[r * 6 + g] = h.bucket_ids[g];
      }
    }
    for (size_t k = 0; k < codes.size(); ++k) { code_ptr.push_back(codes[k].data()); bucket_ptr.push_back(buckets[k].data()); }
  };
  bool inj = injected("cascade", "create");
  int rc = inj ? MVGX_OK : mvgx_cascade_create(-1, &ctx.c);
  if (step("create", rc, inj)) {
    if (hash_check && !rows.empty()) {
      // one view hashed both ways: the device's codes against this build's own CascadeHasher
      size_t probe = 0;
      while (probe + 1 < rows.size() && n_desc[probe] == 0) ++probe;
      const uint32_t n = n_desc[probe];
      std::vector<uint8_t> dev_codes((size_t)n * code_bytes);
      std::vector<uint16_t> dev_buckets((size_t)n * 6);
      uint8_t* cp = dev_codes.data(); uint16_t* bp = dev_buckets.data();
      rc = mvgx_cascade_hash_regions_typed(ctx.c, is_float ? 1 : 0, reinterpret_cast<const void* const*>(rows.data() + probe), n_desc.data() + probe, 1,
                                           (uint32_t)dimension, zero_mean.data(), 6, 10, std::mt19937::default_seed, &cp, &bp);
      if (rc == MVGX_OK && n) {
        View<ScalarT>& v = *order[probe];
        Eigen::Map<RowMajor> m(const_cast<ScalarT*>(v.rows()), v.count(), dimension);
        const matching::HashedDescriptions href = hasher.CreateHashedDescriptions(m, zero_mean);
        bool same = true;
        for (uint32_t r = 0; r < n && same; ++r) {
          same = !std::memcmp(&dev_codes[(size_t)r * code_bytes], href.hashed_desc[r].hash_code.data(), code_bytes);
          for (int g = 0; g < 6 && same; ++g) same = dev_buckets[(size_t)r * 6 + g] == href.hashed_desc[r].bucket_ids[g];
        }
        if (!same) {
          OPENMVG_LOG_WARNING << "mvgx cascade hashing: this build's CascadeHasher rounds differently from the device hashing stage "
                                 "(Eigen compiled with FMA?) - hashing on the host (MVGX_CASCADE_HASH=host)";
          hash_host = true;
        }
        g_last_hash_check.store(same ? 1 : 2);
      }
    }
    if (hash_host) {
      if (!hashed_on_host) { hash_on_host(); hashed_on_host = true; }
      pack_host_hashes();
      inj = injected("cascade", "hash");
      if (!inj)
        rc = mvgx_cascade_set_regions_typed(ctx.c, is_float ? 1 : 0, reinterpret_cast<const void* const*>(rows.data()), code_ptr.data(), bucket_ptr.data(),
                                            n_desc.data(), (uint32_t)rows.size(), (uint32_t)dimension, (uint32_t)code_bytes, 6, 10);
      step("set_regions", rc, inj);
    } else {
      // CascadeHasher::Init(dimension) defaults: 6 bucket groups, 10 bits per bucket, std::mt19937::default_seed
      inj = injected("cascade", "hash");
      if (!inj)
        rc = mvgx_cascade_hash_regions_typed(ctx.c, is_float ? 1 : 0, reinterpret_cast<const void* const*>(rows.data()), n_desc.data(), (uint32_t)rows.size(),
                                             (uint32_t)dimension, zero_mean.data(), 6, 10, std::mt19937::default_seed, nullptr, nullptr);
      step("hash_regions", rc, inj);
    }
  }
  const float ratio_sq = Square(dist_ratio);
  const uint64_t n_pairs = todo.size();
  // MVGX_ADAPTER_TIMING=1: wall time of the three things this loop does, on stderr at the end (device stage | the reference's two
  // de-duplication classes on the host threads | the container, filled by this thread). Measured at 1 000 images x 2 000 (call r5_33):
  // 0.44 | 1.45 | 0.02 s. The de-duplication is the reference's own std::set work and bound by the allocator (more threads: slower);
  // running the next batch on the device meanwhile was built and gains nothing (the device stage has host work of its own: call r5_35).
  const bool timing = std::getenv("MVGX_ADAPTER_TIMING") != nullptr;
  double t_dev = 0.0, t_dedup = 0.0, t_fill = 0.0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  for (uint64_t p0 = 0; !failed && p0 < n_pairs; p0 += kPairsPerCall) {
    if (progress->hasBeenCanceled()) break;
    const uint64_t nb = std::min<uint64_t>(kPairsPerCall, n_pairs - p0);
    inj = injected("cascade", "run");
    auto t0 = now();
    if (!inj) rc = mvgx_cascade_run(ctx.c, dev_pairs.data() + 2 * p0, nb, ratio_sq, nullptr);
    t_dev += since(t0);
    if (!step("run", rc, inj)) break;
    const uint64_t* offsets = nullptr;
    const uint32_t* ij = nullptr;
    mvgx_cascade_results(ctx.c, &offsets, &ij);
    std::vector<matching::IndMatches> lists(nb);
    t0 = now();
    on_host_threads((size_t)((nb + 255) / 256), [&](size_t chunk) {
      for (uint64_t k = chunk * 256, hi = std::min<uint64_t>(nb, k + 256); k < hi; ++k) {
        const uint64_t lo = offsets[k], n = offsets[k + 1] - lo;
        if (!n) continue;
        matching::IndMatches& v = lists[k];
        v.reserve(n);
        for (uint64_t m = 0; m < n; ++m) v.emplace_back(ij[2 * (lo + m)], ij[2 * (lo + m) + 1]);
        deduplicate(v, view_of[dev_pairs[2 * (p0 + k)]]->positions, view_of[dev_pairs[2 * (p0 + k) + 1]]->positions);
      }
    });
    t_dedup += since(t0);
    t0 = now();
    for (uint64_t k = 0; k < nb; ++k)
      if (!lists[k].empty()) out.insert({todo[p0 + k], std::move(lists[k])});
    t_fill += since(t0);
    (*progress) += (uint32_t)nb;
    delivered = p0 + nb;
  }
  if (timing)
    std::fprintf(stderr, "[mvgx cascade adapter] %llu pairs: device stage %.3f s | lists + de-duplication (host threads) %.3f s | container %.3f s\n",
                 (unsigned long long)delivered, t_dev, t_dedup, t_fill);
  mvgx_adapter::counters().device_pairs.fetch_add(delivered);
  if (failed && !progress->hasBeenCanceled()) {
    if (!hashed_on_host) hash_on_host();
    mvgx_adapter::counters().fallback_pairs.fetch_add(n_pairs - delivered);
    host_route((size_t)delivered);
  }
}

}  // namespace

int g_last_hash_check_value() { return g_last_hash_check.load(); }

Cascade_Hashing_Matcher_Regions::Cascade_Hashing_Matcher_Regions(float dist_ratio) : Matcher(), f_dist_ratio_(dist_ratio) {}

void Cascade_Hashing_Matcher_Regions::Match(const std::shared_ptr<sfm::Regions_Provider>& regions_provider, const Pair_Set& pairs,
                                            matching::PairWiseMatchesContainer& map_PutativeMatches,
                                            system::ProgressInterface* progress) const {
  if (!regions_provider || regions_provider->IsBinary()) return;
  if (!progress) progress = &system::ProgressInterface::dummy();
  const std::string type = regions_provider->Type_id();
  if (type == typeid(unsigned char).name())
    match_collection<unsigned char>(*regions_provider, pairs, f_dist_ratio_, map_PutativeMatches, progress);
  else if (type == typeid(float).name())
    match_collection<float>(*regions_provider, pairs, f_dist_ratio_, map_PutativeMatches, progress);
  else
    OPENMVG_LOG_ERROR << "Matcher not implemented for this region type: " << type;
}

}  // namespace matching_image_collection
}  // namespace openMVG

extern "C" int mvgx_adapter_cascade_last_hash_check(void) { return openMVG::matching_image_collection::g_last_hash_check_value(); }
