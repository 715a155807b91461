// mvgx_matcher_regions.cpp — link-time replacement for openMVG's
//   src/openMVG/matching_image_collection/Matcher_Regions.cpp
//
// Compile this translation unit INSTEAD of the reference's Matcher_Regions.cpp (same header, same mangled symbols:
// Matcher_Regions::Matcher_Regions(float, EMatcherType) and Matcher_Regions::Match(...) const) and link
// libmvgx_hip.so. main_ComputeMatches.cpp:248-252,318 and main_benchANN.cpp:168,186 then run unchanged:
//
//   -n BRUTEFORCEL2 on 128-D uint8 regions (SIFT_Regions) with dist_ratio <= 1
//        -> the MI355X path: every (I, J) of the Pair_Set goes to the C ABI in batches, match lists come back
//           bit-identical to RegionsMatcherT<ArrayMatcherBruteForce<uchar, L2<uchar>>>::MatchDistanceRatio
//           (regions_matcher.hpp:162-207) and are inserted in the container in ascending (I, J) order.
//           A failing device call is logged once and the remaining pairs run through the reference's own route
//           (mvgx_adapter_policy.hpp; MVGX_ON_DEVICE_ERROR=throw restores the exception).
//   -n BRUTEFORCEHAMMING on binary uint8 regions of <= 64 bytes (AKAZE_Binary_Regions) with dist_ratio <= 1
//        -> the MI355X popcount path (mvgx_hamming_*), bit-identical to
//           RegionsMatcherT<ArrayMatcherBruteForce<uchar, Hamming<uchar>>>::MatchDistanceRatio (regions_matcher.cpp:184-191)
//   -n BRUTEFORCEL2 on 64-D float regions (AKAZE_Float_Regions) with dist_ratio <= 1
//        -> the MI355X packed-fp32 path (mvgx_l2f_*): L2<float> in the reference's own summation order, bit-identical lists
//   -n BRUTEFORCEL2 on uint8 regions of 64 or 144 bytes (AKAZE_Liop_Regions) with dist_ratio <= 1
//        -> the MI355X integer dot-product path (mvgx_l2u8_*), bit-identical lists
//   anything else (other -n values, other descriptor lengths, ratio > 1 whose tie order is libstdc++'s)
//        -> the per-pair interface the reference itself uses for them (RegionMatcherFactory, regions_matcher.cpp:54),
//           which stays in the link; that code is not part of the accelerated path.
//
// Contract mirrored from Matcher_Regions.cpp:32-107: progress restart with pairs.size(); pairs whose I (or J) has no
// regions are skipped but counted; regions of different Type_id are skipped; only non-empty match vectors are
// inserted; insert() is called from the calling thread only; cancellation is polled between device batches.
// Several GPUs: MVGX_DEVICES=all (or a list of ordinals) in the environment of the unchanged main_ComputeMatches.
#include <cstddef>
#include <cstdint>
#include <map>
#include <type_traits>
#include <memory>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <typeinfo>
#include <unordered_map>
#include <vector>

#include "openMVG/features/regions.hpp"
#include "openMVG/matching/indMatch.hpp"
#include "openMVG/matching/regions_matcher.hpp"
#include "openMVG/matching_image_collection/Matcher_Regions.hpp"
#include "openMVG/numeric/numeric.h"
#include "openMVG/sfm/pipelines/sfm_regions_provider.hpp"
#include "openMVG/system/logger.hpp"
#include "openMVG/system/progressinterface.hpp"

#include "mvgx.h"
#include "mvgx_adapter_policy.hpp"

namespace openMVG {
namespace matching_image_collection {

namespace {

constexpr uint64_t kPairsPerCall = 1u << 17;  // cancellation / progress granularity of the device path (~60 ms of device work)

bool is_sift_u8(const features::Regions& r) {
  return r.IsScalar() && r.DescriptorLength() == 128 && r.Type_id() == typeid(unsigned char).name();
}

bool is_u8_other(const features::Regions& r) {   // uint8 scalar regions of the other supported lengths (AKAZE_Liop_Regions: 144)
  return r.IsScalar() && (r.DescriptorLength() == 64 || r.DescriptorLength() == 144) && r.Type_id() == typeid(unsigned char).name();
}

bool is_float64(const features::Regions& r) {
  return r.IsScalar() && r.DescriptorLength() == 64 && r.Type_id() == typeid(float).name();
}

bool is_binary_u8(const features::Regions& r) {
  return r.IsBinary() && r.DescriptorLength() >= 1 && r.DescriptorLength() <= 64 && r.Type_id() == typeid(unsigned char).name();
}

struct Sink {
  matching::PairWiseMatchesContainer* out;
  const std::vector<IndexT>* ids;  // dense image index -> view id
};

// The match lists of the device batches become IndMatches vectors on helper threads (allocating and first-touching up to a
// gigabyte per run is the host-side cost of a large collection: the page faults of fresh memory, not the copying) while the
// CALLING thread - and only it - inserts them into the container (indMatch.hpp:70-75: insert is not thread safe;
// Matcher_Regions.cpp:95-103 inserts from the thread that called Match). Helper threads live for one Match() call; a batch is
// a job of 256-pair chunks; jobs are built in submission order and taken out in submission order.
class ListBuilder {
 public:
  struct Job {
    const uint32_t* pairs_IJ; uint64_t nb;
    const uint32_t* off32; const uint64_t* off64;   // nb + 1 offsets (in matches) relative to `ij`, one of the two widths
    const uint32_t* ij;
    std::vector<matching::IndMatches> lists;
    uint64_t n_chunks = 0, next = 0, done = 0;       // guarded by the builder's mutex
    std::exception_ptr error;
    uint64_t offset(uint64_t k) const { return off32 ? off32[k] : off64[k]; }
  };

  explicit ListBuilder(unsigned helpers) {
    for (unsigned t = 0; t < helpers; ++t) pool_.emplace_back([this]() { work(); });
  }
  ~ListBuilder() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_work_.notify_all();
    for (auto& t : pool_) t.join();
  }
  void submit(const uint32_t* pairs_IJ, uint64_t nb, const uint32_t* off32, const uint64_t* off64, const uint32_t* ij) {
    std::unique_ptr<Job> j(new Job());
    j->pairs_IJ = pairs_IJ; j->nb = nb; j->off32 = off32; j->off64 = off64; j->ij = ij;
    j->lists.resize(nb);
    j->n_chunks = (nb + kChunk - 1) / kChunk;
    { std::lock_guard<std::mutex> lk(mu_); jobs_.push_back(std::move(j)); }
    cv_work_.notify_all();
  }
  size_t pending() const { return jobs_.size() - taken_; }   // calling thread only
  // waits for the oldest job that has not been taken yet (helping with its chunks meanwhile) and inserts its lists
  void take_oldest(const Sink& sink) {
    Job* j;
    { std::lock_guard<std::mutex> lk(mu_); j = jobs_[taken_].get(); }
    for (;;) {   // the calling thread builds chunks too instead of sleeping
      uint64_t c;
      {
        std::unique_lock<std::mutex> lk(mu_);
        if (j->next >= j->n_chunks) { cv_done_.wait(lk, [&]() { return j->done == j->n_chunks; }); break; }
        c = j->next++;
      }
      build_chunk(*j, c);
      std::lock_guard<std::mutex> lk(mu_);
      ++j->done;
    }
    if (j->error) std::rethrow_exception(j->error);
    static const int debug_skip = std::getenv("MVGX_ADAPTER_DEBUG_SKIP") ? std::atoi(std::getenv("MVGX_ADAPTER_DEBUG_SKIP")) : 0;
    if (debug_skip != 2)   // (measurement knob: lists built, container untouched)
      for (uint64_t k = 0; k < j->nb; ++k)
        if (!j->lists[k].empty())
          sink.out->insert({{(*sink.ids)[j->pairs_IJ[2 * k]], (*sink.ids)[j->pairs_IJ[2 * k + 1]]}, std::move(j->lists[k])});
    std::lock_guard<std::mutex> lk(mu_);
    jobs_[taken_].reset();
    ++taken_;
  }

 private:
  static constexpr uint64_t kChunk = 256;
  static void build_chunk(Job& j, uint64_t c) {
    try {
      for (uint64_t k = c * kChunk, hi = std::min(j.nb, k + kChunk); k < hi; ++k) {
        const uint64_t lo = j.offset(k), n = j.offset(k + 1) - lo;
        if (!n) continue;
        // one allocation and one block copy per pair: IndMatch is two IndexT side by side (indMatch.hpp:25-46), the device's (i, j) words
        static_assert(sizeof(matching::IndMatch) == 2 * sizeof(uint32_t) && std::is_trivially_copyable<matching::IndMatch>::value &&
                      offsetof(matching::IndMatch, i_) == 0 && offsetof(matching::IndMatch, j_) == sizeof(uint32_t), "IndMatch is the (i, j) pair of the device lists");
        const matching::IndMatch* first = reinterpret_cast<const matching::IndMatch*>(j.ij + 2 * lo);
        j.lists[k].assign(first, first + n);
      }
    } catch (...) {
      j.error = std::current_exception();   // (any chunk's failure fails the job; rethrown on the calling thread)
    }
  }
  void work() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      Job* j = nullptr;
      for (size_t q = taken_; q < jobs_.size() && !j; ++q)
        if (jobs_[q] && jobs_[q]->next < jobs_[q]->n_chunks) j = jobs_[q].get();
      if (!j) {
        if (stop_) return;
        cv_work_.wait(lk);
        continue;
      }
      const uint64_t c = j->next++;
      lk.unlock();
      build_chunk(*j, c);
      lk.lock();
      if (++j->done == j->n_chunks) cv_done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<std::unique_ptr<Job>> jobs_;
  size_t taken_ = 0;
  bool stop_ = false;
  std::vector<std::thread> pool_;
};

// The SIFT context of the last Match() call is KEPT (round 6): its device scratch (two batch slots), its page-locked result buffers and
// its streams are what create / destroy cost - 25 + 46 ms of a 0.6 s call at 1 000 x 2 000 - and none of it depends on the image set; the
// next call uploads its own regions into it (mvgx_match_set_regions reuses what fits). One context per process, taken by one Match() at
// a time (a concurrent call makes its own, as before); MVGX_ADAPTER_KEEP_CONTEXT=0 restores create / destroy per call, and
// mvgx_adapter_match_release_context() hands the memory back (tests; a host that is done matching).
struct KeptMatchContext {
  std::mutex mu;
  mvgx_match_ctx* ctx = nullptr;
  bool busy = false;
};
KeptMatchContext& kept_match_context() { static KeptMatchContext* k = new KeptMatchContext(); return *k; }   // (never destroyed: no HIP call at process exit)
bool keep_context_enabled() {
  const char* env = std::getenv("MVGX_ADAPTER_KEEP_CONTEXT");
  return !env || std::atoi(env) != 0;
}

unsigned builder_helpers() { return std::min(15u, std::max(1u, std::thread::hardware_concurrency()) - 1u); }

// The reference's generic per-pair route for matcher types the device path does not cover.
void match_generic(matching::EMatcherType type, float ratio, const std::shared_ptr<sfm::Regions_Provider>& provider,
                   const std::vector<Pair>& pairs, matching::PairWiseMatchesContainer& out,
                   system::ProgressInterface* progress) {
  size_t k = 0;
  while (k < pairs.size()) {
    const IndexT I = pairs[k].first;
    size_t end = k;
    while (end < pairs.size() && pairs[end].first == I) ++end;
    if (progress->hasBeenCanceled()) { k = end; continue; }
    const std::shared_ptr<features::Regions> regionsI = provider->get(I);
    std::unique_ptr<matching::RegionsMatcher> matcher;
    if (regionsI && regionsI->RegionCount() > 0) matcher = matching::RegionMatcherFactory(type, *regionsI);
    for (; k < end; ++k) {
      if (matcher) {
        const std::shared_ptr<features::Regions> regionsJ = provider->get(pairs[k].second);
        if (regionsJ && regionsJ->RegionCount() > 0 && regionsI->Type_id() == regionsJ->Type_id()) {
          matching::IndMatches v;
          matcher->MatchDistanceRatio(ratio, *regionsJ, v);
          if (!v.empty()) out.insert({pairs[k], std::move(v)});
        }
      }
      ++(*progress);
    }
  }
}

}  // namespace

Matcher_Regions::Matcher_Regions(float dist_ratio, matching::EMatcherType eMatcherType)
    : Matcher(), f_dist_ratio_(dist_ratio), eMatcherType_(eMatcherType) {}

void Matcher_Regions::Match(const std::shared_ptr<sfm::Regions_Provider>& regions_provider, const Pair_Set& pairs,
                            matching::PairWiseMatchesContainer& map_PutativeMatches,
                            system::ProgressInterface* progress) const {
  if (!progress) progress = &system::ProgressInterface::dummy();
  progress->Restart(pairs.size(), "- Matching -");
  // MVGX_ADAPTER_TIMING=1: phase times of this call on stderr
  const bool timing = std::getenv("MVGX_ADAPTER_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto tick = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[mvgx Matcher_Regions::Match] %-32s %8.2f ms\n", what,
                 std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };

  const float ratio_sq = Square(f_dist_ratio_);  // regions_matcher.hpp:196 (squared metric), numeric.h:56
  const bool hamming = eMatcherType_ == matching::BRUTE_FORCE_HAMMING;   // metric not squared: the ratio is used as given
  const bool device_type = (eMatcherType_ == matching::BRUTE_FORCE_L2 && ratio_sq <= 1.0f && ratio_sq >= 0.0f) ||
                           (hamming && f_dist_ratio_ <= 1.0f && f_dist_ratio_ >= 0.0f);
  size_t binary_len = 0;   // one descriptor length per device run (a Regions_Provider holds one region type)
  int l2_kind = -1;        // BRUTE_FORCE_L2: 0 = 128-D uint8, 1 = 64-D float, 2 = uint8 of another length; first usable regions decide

  // Pair_Set is ordered by (I, J): the order in which the reference visits and inserts.
  std::vector<Pair> generic_pairs;
  std::vector<IndexT> ids;                       // dense index -> view id
  std::unordered_map<IndexT, uint32_t> dense;    // view id -> dense index
  std::vector<std::shared_ptr<features::Regions>> keep;  // holds the descriptor memory alive for the whole call
  std::vector<uint32_t> dev_pairs;               // 2 x n: (dense I, dense J)
  uint32_t skipped = 0;

  auto dense_id = [&](IndexT view) -> int64_t {
    auto it = dense.find(view);
    if (it != dense.end()) return it->second;
    std::shared_ptr<features::Regions> r = regions_provider->get(view);
    if (!r) return -1;
    if (hamming) {
      if (!is_binary_u8(*r)) return -1;
      if (!binary_len) binary_len = r->DescriptorLength();
      if (r->DescriptorLength() != binary_len) return -1;
    } else {
      if (l2_kind < 0) { l2_kind = is_float64(*r) ? 1 : is_u8_other(*r) ? 2 : 0; if (l2_kind == 2) binary_len = r->DescriptorLength(); }
      if (l2_kind == 1 ? !is_float64(*r) : l2_kind == 2 ? !(is_u8_other(*r) && r->DescriptorLength() == binary_len) : !is_sift_u8(*r)) return -1;
    }
    const uint32_t k = static_cast<uint32_t>(ids.size());
    dense.emplace(view, k);
    ids.push_back(view);
    keep.push_back(std::move(r));
    return k;
  };

  for (const Pair& p : pairs) {
    if (!device_type) { generic_pairs.push_back(p); continue; }
    const int64_t a = dense_id(p.first), b = dense_id(p.second);
    if (a >= 0 && b >= 0) {
      dev_pairs.push_back(static_cast<uint32_t>(a));
      dev_pairs.push_back(static_cast<uint32_t>(b));
      continue;
    }
    // one side is not 128-D uint8: the reference skips mismatching Type_id, otherwise uses its own matcher
    const std::shared_ptr<features::Regions> ri = regions_provider->get(p.first), rj = regions_provider->get(p.second);
    if (ri && rj && ri->RegionCount() > 0 && rj->RegionCount() > 0 && ri->Type_id() == rj->Type_id())
      generic_pairs.push_back(p);
    else
      ++skipped;
  }
  if (skipped) (*progress) += skipped;
  tick("pair list + regions lookup");

  if (!dev_pairs.empty()) {
    std::vector<const uint8_t*> rows(ids.size());
    std::vector<uint32_t> n_desc(ids.size());
    for (size_t k = 0; k < ids.size(); ++k) {
      n_desc[k] = static_cast<uint32_t>(keep[k]->RegionCount());
      rows[k] = n_desc[k] ? static_cast<const uint8_t*>(keep[k]->DescriptorRawData()) : nullptr;
    }
    Sink sink{&map_PutativeMatches, &ids};
    const uint64_t n_pairs = dev_pairs.size() / 2;
    // the device paths have the same call shapes (include/mvgx.h)
    const bool f32 = !hamming && l2_kind == 1, u8o = !hamming && l2_kind == 2;
    // every exit path - including an exception out of the container or of a list allocation - releases the device context
    struct Contexts {
      mvgx_match_ctx* l2 = nullptr;
      mvgx_hamming_ctx* hm = nullptr;
      mvgx_l2f_ctx* lf = nullptr;
      mvgx_l2u8_ctx* lu = nullptr;
      bool l2_kept = false;    // l2 is the process's kept context, held by this call
      bool l2_healthy = true;  // ... and goes back to the holder (false after a failing device call: destroyed instead)
      void release() {
        if (hm) mvgx_hamming_destroy(hm);
        if (lf) mvgx_l2f_destroy(lf);
        if (lu) mvgx_l2u8_destroy(lu);
        if (l2 && l2_kept) {
          KeptMatchContext& k = kept_match_context();
          std::lock_guard<std::mutex> lk(k.mu);
          if (!l2_healthy) { mvgx_match_destroy(l2); k.ctx = nullptr; }
          k.busy = false;
        } else if (l2) {
          mvgx_match_destroy(l2);
        }
        hm = nullptr; lf = nullptr; lu = nullptr; l2 = nullptr; l2_kept = false;
      }
      ~Contexts() { release(); }
    } ctx;
    // device -1 = "no preference": MVGX_DEVICES (e.g. "all") makes the SIFT path one context over several GPUs of the node
    // Error convention (mvgx_adapter_policy.hpp): a failing device call is logged once; the pairs whose lists have not reached the
    // container go through the reference's own RegionMatcherFactory route below (or the failure is thrown, MVGX_ON_DEVICE_ERROR=throw).
    using mvgx_adapter::injected;
    bool failed = false;
    uint64_t delivered = 0;   // device pairs [0, delivered): lists in the container (or known to be empty), progress advanced
    auto step = [&](const char* stage, int rc_call, bool inj) {   // true = go on with the device
      if (!inj && rc_call == MVGX_OK) return true;
      mvgx_adapter::device_failure(mvgx_adapter::kMatch, "matching", stage, inj ? MVGX_ERR_NODEV : rc_call, inj);
      failed = true;
      ctx.l2_healthy = false;   // (a kept context that failed is not kept)
      return false;
    };
    const uint32_t n_img = static_cast<uint32_t>(ids.size());
    int rc = MVGX_OK;
    bool inj = injected("match", "create");
    const bool sift = !hamming && !f32 && !u8o;
    if (!inj && sift && keep_context_enabled()) {   // the kept context, when no other call holds it
      KeptMatchContext& k = kept_match_context();
      std::lock_guard<std::mutex> lk(k.mu);
      if (!k.busy) {
        if (!k.ctx) rc = mvgx_match_create(-1, &k.ctx);
        if (rc == MVGX_OK) { ctx.l2 = k.ctx; ctx.l2_kept = true; k.busy = true; }
      }
    }
    if (!inj && !ctx.l2_kept && rc == MVGX_OK)
      rc = hamming ? mvgx_hamming_create(-1, &ctx.hm) : f32 ? mvgx_l2f_create(-1, &ctx.lf) : u8o ? mvgx_l2u8_create(-1, &ctx.lu)
                                                                                                 : mvgx_match_create(-1, &ctx.l2);
    // (SIFT path) the page-locked result buffers of the stream are obtained on a helper thread WHILE the regions are uploaded: in the first
    // call of a process they cost what the plain-memory copies they replace cost (~100 ms at 1 000 x 2 000), later calls find them in place
    std::thread reserve_thread;
    struct JoinReserve { std::thread& t; ~JoinReserve() { if (t.joinable()) t.join(); } } join_reserve{reserve_thread};
    if (!inj && rc == MVGX_OK && ctx.l2) {
      const char* env = std::getenv("MVGX_ADAPTER_PINNED_RESULTS");
      const char* envb = std::getenv("MVGX_ADAPTER_BATCH_PAIRS");
      const int64_t batch_pairs = envb ? std::max(1, std::atoi(envb)) : (1 << 14);
      mvgx_match_set_option(ctx.l2, "stream_hold", 1);
      mvgx_match_set_option(ctx.l2, "pinned_stream", env ? std::atoi(env) : 1);
      mvgx_match_set_option(ctx.l2, "batch_pairs", batch_pairs);
      uint64_t sum_desc = 0;
      for (uint32_t v : n_desc) sum_desc += v;
      // a guess at a batch's lists: 0.4 matches per feature of the left image (dense synthetic sets reach 0.2, real image sets a tenth
      // of that), capped at 128 MB per buffer; a batch that exceeds its buffer re-pins it with headroom (the library's rule)
      const uint64_t words = std::min<uint64_t>((128u << 20) / 4, (uint64_t)(0.4 * 2.0 * (double)batch_pairs * (double)sum_desc / std::max<size_t>(n_desc.size(), 1)));
      mvgx_match_ctx* l2 = ctx.l2;
      reserve_thread = std::thread([l2, words]() { (void)mvgx_match_set_option(l2, "stream_reserve", (int64_t)words); });   // (a failure here: the run allocates itself)
    }
    if (step("create", rc, inj)) {
      inj = injected("match", "set_regions");
      if (!inj)
        rc = hamming ? mvgx_hamming_set_regions(ctx.hm, rows.data(), n_desc.data(), n_img, static_cast<uint32_t>(binary_len ? binary_len : 64))
             : f32   ? mvgx_l2f_set_regions(ctx.lf, reinterpret_cast<const float* const*>(rows.data()), n_desc.data(), n_img, 64)
             : u8o   ? mvgx_l2u8_set_regions(ctx.lu, rows.data(), n_desc.data(), n_img, static_cast<uint32_t>(binary_len))
                     : mvgx_match_set_regions(ctx.l2, rows.data(), n_desc.data(), n_img, 128);
      step("set_regions", rc, inj);
    }
    if (reserve_thread.joinable()) reserve_thread.join();
    tick("context + upload + tile build");
    if (!failed && ctx.l2) {
      // SIFT path: the lists arrive batch by batch on THIS thread (mvgx_match_run_stream) while the device(s) work on the
      // next batches; host memory beside the container itself is two batches per device. Cancellation is polled per batch.
      // "stream_hold": a batch's buffers outlive two further sink calls, so batch k is converted by the helper threads while
      // batches k + 1 and k + 2 arrive; the container takes batch k - 2 at call k (and the rest after the run) - on this thread.
      // (options set above, before the regions went up: "stream_hold" - a batch's buffers outlive two further sink calls -, the lists
      // through page-locked buffers - round 6: into plain memory the runtime stages the copies itself, 976 MB of lists at 1 000 x 2 000
      // took 130 ms longer than the device needs for the whole run (call r6_09: 336 against 208 ms with the sink switched off) -, batches
      // of 16 384 pairs: four buffers of ~70 MB at that size)
      ListBuilder list_builder(builder_helpers());
      struct Stream {
        ListBuilder& builder; const Sink* sink; const uint32_t* pairs; system::ProgressInterface* progress; std::exception_ptr error;
        uint64_t* delivered;
        static int on_batch(void* user, uint64_t first_pair, uint32_t nb, const uint32_t* offsets, const uint32_t* ij) {
          Stream& s = *static_cast<Stream*>(user);
          static const int debug_skip = std::getenv("MVGX_ADAPTER_DEBUG_SKIP") ? std::atoi(std::getenv("MVGX_ADAPTER_DEBUG_SKIP")) : 0;
          if (debug_skip == 1) return 0;   // (measurement knob: device + transfers only)
          try {
            s.builder.submit(s.pairs + 2 * first_pair, nb, offsets, nullptr, ij);
            while (s.builder.pending() > 2) s.builder.take_oldest(*s.sink);
            (*s.progress) += nb;
            *s.delivered = first_pair + nb;   // (batches arrive in order; what is still pending in the builder is flushed after the run)
          } catch (...) {   // never unwind through the C frames: stop the run, rethrow after it has returned
            s.error = std::current_exception();
            return 1;
          }
          return s.progress->hasBeenCanceled() ? 1 : 0;
        }
      } stream{list_builder, &sink, dev_pairs.data(), progress, nullptr, &delivered};
      if (!progress->hasBeenCanceled()) {
        inj = injected("match", "run");
        rc = inj ? MVGX_OK : mvgx_match_run_stream(ctx.l2, dev_pairs.data(), n_pairs, ratio_sq, &Stream::on_batch, &stream, nullptr);
        if (stream.error) std::rethrow_exception(stream.error);
        while (stream.builder.pending()) stream.builder.take_oldest(sink);   // the last batches: their buffers live until the next call on the context
        step("run", rc, inj);
      }
      tick("device runs + container fill");
    } else if (!failed) {
      ListBuilder builder(builder_helpers());
      for (uint64_t p0 = 0; p0 < n_pairs; p0 += kPairsPerCall) {
        const uint64_t nb = std::min<uint64_t>(kPairsPerCall, n_pairs - p0);
        if (progress->hasBeenCanceled()) break;
        inj = injected("match", "run");
        if (!inj)
          rc = hamming ? mvgx_hamming_run(ctx.hm, dev_pairs.data() + 2 * p0, nb, f_dist_ratio_, nullptr)
               : f32   ? mvgx_l2f_run(ctx.lf, dev_pairs.data() + 2 * p0, nb, ratio_sq, nullptr)
                       : mvgx_l2u8_run(ctx.lu, dev_pairs.data() + 2 * p0, nb, ratio_sq, nullptr);
        tick("device run");
        if (!step("run", rc, inj)) break;
        const uint64_t* offsets = nullptr;
        const uint32_t* ij = nullptr;
        if (hamming) mvgx_hamming_results(ctx.hm, &offsets, &ij);
        else if (f32) mvgx_l2f_results(ctx.lf, &offsets, &ij);
        else mvgx_l2u8_results(ctx.lu, &offsets, &ij);
        builder.submit(dev_pairs.data() + 2 * p0, nb, nullptr, offsets, ij);
        builder.take_oldest(sink);
        tick("container fill");
        (*progress) += static_cast<uint32_t>(nb);
        delivered = p0 + nb;
      }
    }
    mvgx_adapter::counters().device_pairs.fetch_add(delivered);
    if (failed && !progress->hasBeenCanceled()) {
      // the host application's own reference route for what the device did not deliver (Pair_Set order is kept: dev_pairs is in it)
      std::vector<Pair> rest;
      rest.reserve(n_pairs - delivered);
      for (uint64_t k = delivered; k < n_pairs; ++k) rest.emplace_back(ids[dev_pairs[2 * k]], ids[dev_pairs[2 * k + 1]]);
      mvgx_adapter::counters().fallback_pairs.fetch_add(rest.size());
      match_generic(eMatcherType_, f_dist_ratio_, regions_provider, rest, map_PutativeMatches, progress);
      tick("reference route after a device failure");
    }
    ctx.release();
    tick("context release");
  }

  if (!generic_pairs.empty())
    match_generic(eMatcherType_, f_dist_ratio_, regions_provider, generic_pairs, map_PutativeMatches, progress);
}

}  // namespace matching_image_collection
}  // namespace openMVG

// releases the kept SIFT context (device scratch, page-locked buffers); a Match() in flight keeps its hold: returns 0 then, 1 when released / nothing was kept
extern "C" int mvgx_adapter_match_release_context() {
  using openMVG::matching_image_collection::kept_match_context;
  auto& k = kept_match_context();
  std::lock_guard<std::mutex> lk(k.mu);
  if (k.busy) return 0;
  if (k.ctx) { mvgx_match_destroy(k.ctx); k.ctx = nullptr; }
  return 1;
}
