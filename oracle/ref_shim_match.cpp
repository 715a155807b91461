// ref_shim_match.cpp — TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" wrapper (our code) around the REFERENCE's own matching implementation, compiled in place from
// /root/reference/src by oracle/Makefile into oracle/_ref/libref_match.so. No reference source is copied into this
// repository: this file only #includes the reference headers and calls the reference classes:
//   openMVG::matching::L2<uint8_t>                                   matching/metric.hpp:55-93 (+ metric_simd.hpp AVX2)
//   openMVG::matching::ArrayMatcherBruteForce<uint8_t, L2<uint8_t>>  matching/matcher_brute_force.hpp:27-201
//   openMVG::matching_image_collection::Matcher_Regions              matching_image_collection/Matcher_Regions.cpp:22-107
//   openMVG::matching_image_collection::Cascade_Hashing_Matcher_Regions, openMVG::matching::CascadeHasher
//                                                                    Cascade_Hashing_Matcher_Regions.cpp:28-272, cascade_hasher.hpp
// fed through an in-memory sfm::Regions_Provider (sfm/pipelines/sfm_regions_provider.hpp:30-144, cache_ is protected).
// Used to (a) validate oracle/match_oracle.c, (b) serve as the "reference" CPU baseline in bench.py.
#include <chrono>
#include <cstdint>
#include <exception>
#include <cstring>
#include <memory>
#include <vector>

#include "openMVG/features/regions_factory.hpp"
#include "openMVG/matching/indMatch.hpp"
#include "openMVG/matching/matcher_brute_force.hpp"
#include "openMVG/matching/matching_filters.hpp"
#include "openMVG/numeric/numeric.h"
#include "openMVG/matching/metric.hpp"
#include "openMVG/matching/metric_hamming.hpp"
#include "openMVG/matching/cascade_hasher.hpp"
#include "openMVG/matching_image_collection/Cascade_Hashing_Matcher_Regions.hpp"
#include "openMVG/matching_image_collection/Matcher_Regions.hpp"
#include "openMVG/sfm/pipelines/sfm_regions_provider.hpp"

using namespace openMVG;

namespace {

struct InMemoryRegionsProvider : public sfm::Regions_Provider {
  void set(IndexT id, std::shared_ptr<features::Regions> r) { cache_[id] = std::move(r); }
  void set_type(features::Regions* t) { region_type_.reset(t); }
};

std::shared_ptr<features::Regions> make_sift_regions(const uint8_t* rows, uint32_t n) {
  auto r = std::make_shared<features::SIFT_Regions>();
  r->Features().resize(n);
  r->Descriptors().resize(n);
  for (uint32_t k = 0; k < n; ++k) {
    r->Features()[k] = features::SIOPointFeature(float(k), float(k), 1.f, 0.f);
    std::memcpy(r->Descriptors()[k].data(), rows + size_t(k) * 128, 128);
  }
  return r;
}

std::shared_ptr<features::Regions> make_binary64_regions(const uint8_t* rows, uint32_t n) {
  auto r = std::make_shared<features::AKAZE_Binary_Regions>();   // Binary_Regions<SIOPointFeature, 64>, regions_factory.hpp:26
  r->Features().resize(n);
  r->Descriptors().resize(n);
  for (uint32_t k = 0; k < n; ++k) {
    r->Features()[k] = features::SIOPointFeature(float(k), float(k), 1.f, 0.f);
    std::memcpy(r->Descriptors()[k].data(), rows + size_t(k) * 64, 64);
  }
  return r;
}

std::shared_ptr<features::Regions> make_float64_regions(const float* rows, uint32_t n) {
  auto r = std::make_shared<features::AKAZE_Float_Regions>();   // Scalar_Regions<SIOPointFeature, float, 64>, regions_factory.hpp:22
  r->Features().resize(n);
  r->Descriptors().resize(n);
  for (uint32_t k = 0; k < n; ++k) {
    r->Features()[k] = features::SIOPointFeature(float(k), float(k), 1.f, 0.f);
    std::memcpy(r->Descriptors()[k].data(), rows + size_t(k) * 64, 64 * sizeof(float));
  }
  return r;
}

std::shared_ptr<features::Regions> make_liop144_regions(const uint8_t* rows, uint32_t n) {
  auto r = std::make_shared<features::AKAZE_Liop_Regions>();   // Scalar_Regions<SIOPointFeature, unsigned char, 144>
  r->Features().resize(n);
  r->Descriptors().resize(n);
  for (uint32_t k = 0; k < n; ++k) {
    r->Features()[k] = features::SIOPointFeature(float(k), float(k), 1.f, 0.f);
    std::memcpy(r->Descriptors()[k].data(), rows + size_t(k) * 144, 144);
  }
  return r;
}

}  // namespace

extern "C" {

typedef void (*ref_match_sink)(void* user, uint32_t I, uint32_t J, const uint32_t* ij, uint32_t n);

// L2<uint8_t> on `size` elements (the AVX2 path of the reference engages for size == 128 when built with
// -DOPENMVG_USE_AVX2 and needs 32-byte aligned rows: inputs are copied to aligned storage here).
int ref_l2_u8(const uint8_t* a, const uint8_t* b, size_t size) {
  std::vector<uint8_t, Eigen::aligned_allocator<uint8_t>> aa(a, a + size), bb(b, b + size);
  matching::L2<uint8_t> metric;
  return metric(aa.data(), bb.data(), size);
}

int ref_uses_avx2(void) {
#ifdef OPENMVG_USE_AVX2
  return 1;
#else
  return 0;
#endif
}

// ArrayMatcherBruteForce<uint8_t>::Build + SearchNeighbours. out arrays sized nJ*NN. Returns 1 on success.
int ref_search_neighbours_u8(const uint8_t* db, int nI, const uint8_t* queries, int nJ, int dim, int NN,
                             int32_t* out_index, int32_t* out_dist) {
  std::vector<uint8_t, Eigen::aligned_allocator<uint8_t>> dbc(db, db + size_t(nI > 0 ? nI : 0) * dim);
  std::vector<uint8_t, Eigen::aligned_allocator<uint8_t>> qc(queries, queries + size_t(nJ > 0 ? nJ : 0) * dim);
  matching::ArrayMatcherBruteForce<uint8_t, matching::L2<uint8_t>> m;
  m.Build(dbc.data(), nI, dim);
  matching::IndMatches idx;
  std::vector<int> dist;
  if (!m.SearchNeighbours(qc.data(), nJ, &idx, &dist, size_t(NN))) return 0;
  for (size_t k = 0; k < idx.size(); ++k) {
    out_index[k] = int32_t(idx[k].j_);
    out_dist[k] = dist[k];
  }
  return 1;
}

// Matcher_Regions(dist_ratio, BRUTE_FORCE_L2).Match on in-memory SIFT_Regions. `sink` is called for every entry of
// the resulting PairWiseMatches map (sorted by (I, J)). Returns the number of pairs with matches.
uint64_t ref_matcher_regions_match_u8(const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                                      const uint32_t* pairs_IJ, uint64_t n_pairs, float dist_ratio,
                                      ref_match_sink sink, void* user) {
  auto provider = std::make_shared<InMemoryRegionsProvider>();
  provider->set_type(new features::SIFT_Regions());
  for (uint32_t k = 0; k < n_images; ++k) provider->set(k, make_sift_regions(desc_rows[k], n_desc[k]));
  Pair_Set pairs;
  for (uint64_t p = 0; p < n_pairs; ++p) pairs.insert({pairs_IJ[2 * p], pairs_IJ[2 * p + 1]});
  matching::PairWiseMatches out;
  matching_image_collection::Matcher_Regions matcher(dist_ratio, matching::BRUTE_FORCE_L2);
  std::shared_ptr<sfm::Regions_Provider> base = provider;
  try {   // (the reference's Match never throws; a replacement TU asked to - MVGX_ON_DEVICE_ERROR=throw - must not unwind into ctypes)
    matcher.Match(base, pairs, out, nullptr);
  } catch (const std::exception&) {
    return UINT64_MAX;
  }
  std::vector<uint32_t> flat;
  for (const auto& kv : out) {
    flat.resize(kv.second.size() * 2);
    for (size_t m = 0; m < kv.second.size(); ++m) {
      flat[2 * m] = kv.second[m].i_;
      flat[2 * m + 1] = kv.second[m].j_;
    }
    if (sink) sink(user, kv.first.first, kv.first.second, flat.data(), uint32_t(kv.second.size()));
  }
  return out.size();
}

// Hamming<unsigned char> on `size` bytes (matching/metric_hamming.hpp:36-107).
unsigned int ref_hamming_u8(const uint8_t* a, const uint8_t* b, size_t size) {
  std::vector<uint8_t, Eigen::aligned_allocator<uint8_t>> aa(a, a + size), bb(b, b + size);
  matching::Hamming<uint8_t> metric;
  return metric(aa.data(), bb.data(), size);
}

// Matcher_Regions(dist_ratio, BRUTE_FORCE_HAMMING).Match on in-memory AKAZE_Binary_Regions (64-byte descriptors); same
// output convention as ref_matcher_regions_match_u8.
uint64_t ref_matcher_regions_match_binary64(const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                                            const uint32_t* pairs_IJ, uint64_t n_pairs, float dist_ratio,
                                            ref_match_sink sink, void* user) {
  auto provider = std::make_shared<InMemoryRegionsProvider>();
  provider->set_type(new features::AKAZE_Binary_Regions());
  for (uint32_t k = 0; k < n_images; ++k) provider->set(k, make_binary64_regions(desc_rows[k], n_desc[k]));
  Pair_Set pairs;
  for (uint64_t p = 0; p < n_pairs; ++p) pairs.insert({pairs_IJ[2 * p], pairs_IJ[2 * p + 1]});
  matching::PairWiseMatches out;
  matching_image_collection::Matcher_Regions matcher(dist_ratio, matching::BRUTE_FORCE_HAMMING);
  std::shared_ptr<sfm::Regions_Provider> base = provider;
  matcher.Match(base, pairs, out, nullptr);
  std::vector<uint32_t> flat;
  for (const auto& kv : out) {
    flat.resize(kv.second.size() * 2);
    for (size_t m = 0; m < kv.second.size(); ++m) {
      flat[2 * m] = kv.second[m].i_;
      flat[2 * m + 1] = kv.second[m].j_;
    }
    if (sink) sink(user, kv.first.first, kv.first.second, flat.data(), uint32_t(kv.second.size()));
  }
  return out.size();
}

// L2<float> on `size` elements (matching/metric.hpp:98-135).
float ref_l2_f32(const float* a, const float* b, size_t size) {
  std::vector<float, Eigen::aligned_allocator<float>> aa(a, a + size), bb(b, b + size);
  matching::L2<float> metric;
  return metric(aa.data(), bb.data(), size);
}

// Matcher_Regions(dist_ratio, BRUTE_FORCE_L2).Match on in-memory AKAZE_Float_Regions (64 floats); same output convention.
uint64_t ref_matcher_regions_match_float64(const float* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                                           const uint32_t* pairs_IJ, uint64_t n_pairs, float dist_ratio,
                                           ref_match_sink sink, void* user) {
  auto provider = std::make_shared<InMemoryRegionsProvider>();
  provider->set_type(new features::AKAZE_Float_Regions());
  for (uint32_t k = 0; k < n_images; ++k) provider->set(k, make_float64_regions(desc_rows[k], n_desc[k]));
  Pair_Set pairs;
  for (uint64_t p = 0; p < n_pairs; ++p) pairs.insert({pairs_IJ[2 * p], pairs_IJ[2 * p + 1]});
  matching::PairWiseMatches out;
  matching_image_collection::Matcher_Regions matcher(dist_ratio, matching::BRUTE_FORCE_L2);
  std::shared_ptr<sfm::Regions_Provider> base = provider;
  matcher.Match(base, pairs, out, nullptr);
  std::vector<uint32_t> flat;
  for (const auto& kv : out) {
    flat.resize(kv.second.size() * 2);
    for (size_t m = 0; m < kv.second.size(); ++m) {
      flat[2 * m] = kv.second[m].i_;
      flat[2 * m + 1] = kv.second[m].j_;
    }
    if (sink) sink(user, kv.first.first, kv.first.second, flat.data(), uint32_t(kv.second.size()));
  }
  return out.size();
}

// Wall time of Matcher_Regions::Match itself (exhaustive pairs over n_images SIFT_Regions), container included, without the
// per-pair callback: the end-to-end figure a caller of main_ComputeMatches sees. out[0] = seconds, out[1] = matches,
// out[2] = pairs with matches.
int ref_matcher_regions_match_u8_timed(const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                                       float dist_ratio, double* out) {
  auto provider = std::make_shared<InMemoryRegionsProvider>();
  provider->set_type(new features::SIFT_Regions());
  for (uint32_t k = 0; k < n_images; ++k) provider->set(k, make_sift_regions(desc_rows[k], n_desc[k]));
  Pair_Set pairs;
  for (uint32_t i = 0; i < n_images; ++i)
    for (uint32_t j = i + 1; j < n_images; ++j) pairs.insert({i, j});
  matching::PairWiseMatches res;
  matching_image_collection::Matcher_Regions matcher(dist_ratio, matching::BRUTE_FORCE_L2);
  std::shared_ptr<sfm::Regions_Provider> base = provider;
  const auto t0 = std::chrono::steady_clock::now();
  matcher.Match(base, pairs, res, nullptr);
  out[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  uint64_t n = 0;
  for (const auto& kv : res) n += kv.second.size();
  out[1] = double(n);
  out[2] = double(res.size());
  return 0;
}

// Matcher_Regions(dist_ratio, BRUTE_FORCE_L2).Match on in-memory AKAZE_Liop_Regions (144 x uint8); same output convention.
uint64_t ref_matcher_regions_match_liop144(const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                                           const uint32_t* pairs_IJ, uint64_t n_pairs, float dist_ratio,
                                           ref_match_sink sink, void* user) {
  auto provider = std::make_shared<InMemoryRegionsProvider>();
  provider->set_type(new features::AKAZE_Liop_Regions());
  for (uint32_t k = 0; k < n_images; ++k) provider->set(k, make_liop144_regions(desc_rows[k], n_desc[k]));
  Pair_Set pairs;
  for (uint64_t p = 0; p < n_pairs; ++p) pairs.insert({pairs_IJ[2 * p], pairs_IJ[2 * p + 1]});
  matching::PairWiseMatches out;
  matching_image_collection::Matcher_Regions matcher(dist_ratio, matching::BRUTE_FORCE_L2);
  std::shared_ptr<sfm::Regions_Provider> base = provider;
  matcher.Match(base, pairs, out, nullptr);
  std::vector<uint32_t> flat;
  for (const auto& kv : out) {
    flat.resize(kv.second.size() * 2);
    for (size_t m = 0; m < kv.second.size(); ++m) {
      flat[2 * m] = kv.second[m].i_;
      flat[2 * m + 1] = kv.second[m].j_;
    }
    if (sink) sink(user, kv.first.first, kv.first.second, flat.data(), uint32_t(kv.second.size()));
  }
  return out.size();
}

// ---- CASCADE_HASHING_L2 ------------------------------------------------------------------------------------------------
// The hashing stage exactly as Cascade_Hashing_Matcher_Regions.cpp:66-131 runs it (CascadeHasher::Init(dimension), the zero-mean
// descriptor = mean over the images of the per-image mean, CreateHashedDescriptions per image), with the per-descriptor outputs
// copied out: hash_out[k] n x 16 bytes (dynamic_bitset blocks), bids_out[k] n x 6 uint16. Used by the tests as the input of the
// device / restatement matching stage.
int ref_cascade_hash_u8(const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images, uint8_t* const* hash_out,
                        uint16_t* const* bids_out) {
  using BaseMat = Eigen::Matrix<unsigned char, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
  matching::CascadeHasher hasher;
  hasher.Init(128);
  Eigen::MatrixXf per_image(n_images, 128);
  per_image.fill(0.0f);
  for (uint32_t k = 0; k < n_images; ++k)
    if (n_desc[k] > 0) {
      Eigen::Map<BaseMat> m(const_cast<unsigned char*>(desc_rows[k]), n_desc[k], 128);
      per_image.row(k) = matching::CascadeHasher::GetZeroMeanDescriptor(m);
    }
  const Eigen::VectorXf zero_mean = matching::CascadeHasher::GetZeroMeanDescriptor(per_image);
  for (uint32_t k = 0; k < n_images; ++k) {
    Eigen::Map<BaseMat> m(const_cast<unsigned char*>(desc_rows[k]), n_desc[k], 128);
    const matching::HashedDescriptions h = hasher.CreateHashedDescriptions(m, zero_mean);
    for (size_t r = 0; r < h.hashed_desc.size(); ++r) {
      if (h.hashed_desc[r].hash_code.num_blocks() != 16 || h.hashed_desc[r].bucket_ids.size() != 6) return 0;
      std::memcpy(hash_out[k] + r * 16, h.hashed_desc[r].hash_code.data(), 16);
      for (int g = 0; g < 6; ++g) bids_out[k][r * 6 + g] = h.hashed_desc[r].bucket_ids[g];
    }
  }
  return 1;
}

// The zero-mean descriptor of the hashing stage alone (Cascade_Hashing_Matcher_Regions.cpp:78-104: the mean over the images of the
// per-image mean, both CascadeHasher::GetZeroMeanDescriptor): 128 floats. The device hashing stage takes it as an input.
int ref_cascade_zero_mean_u8(const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images, float* out128) {
  using BaseMat = Eigen::Matrix<unsigned char, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
  Eigen::MatrixXf per_image(n_images, 128);
  per_image.fill(0.0f);
  for (uint32_t k = 0; k < n_images; ++k)
    if (n_desc[k] > 0) {
      Eigen::Map<BaseMat> m(const_cast<unsigned char*>(desc_rows[k]), n_desc[k], 128);
      per_image.row(k) = matching::CascadeHasher::GetZeroMeanDescriptor(m);
    }
  const Eigen::VectorXf zero_mean = matching::CascadeHasher::GetZeroMeanDescriptor(per_image);
  if (zero_mean.size() != 128) return 0;
  for (int j = 0; j < 128; ++j) out128[j] = zero_mean(j);
  return 1;
}

// Cascade_Hashing_Matcher_Regions(dist_ratio).Match on in-memory SIFT_Regions whose features sit at feat_xy[k] (n x 2 floats:
// the reference removes matches that repeat the same coordinates, Cascade_Hashing_Matcher_Regions.cpp:221-226). Same output
// convention as ref_matcher_regions_match_u8.
uint64_t ref_cascade_matcher_regions_match_u8(const uint8_t* const* desc_rows, const float* const* feat_xy, const uint32_t* n_desc,
                                              uint32_t n_images, const uint32_t* pairs_IJ, uint64_t n_pairs, float dist_ratio,
                                              ref_match_sink sink, void* user) {
  auto provider = std::make_shared<InMemoryRegionsProvider>();
  provider->set_type(new features::SIFT_Regions());
  for (uint32_t k = 0; k < n_images; ++k) {
    std::shared_ptr<features::Regions> r = make_sift_regions(desc_rows[k], n_desc[k]);
    auto* sr = static_cast<features::SIFT_Regions*>(r.get());
    for (uint32_t q = 0; q < n_desc[k]; ++q) sr->Features()[q] = features::SIOPointFeature(feat_xy[k][2 * q], feat_xy[k][2 * q + 1], 1.f, 0.f);
    provider->set(k, r);
  }
  Pair_Set pairs;
  for (uint64_t p = 0; p < n_pairs; ++p) pairs.insert({pairs_IJ[2 * p], pairs_IJ[2 * p + 1]});
  matching::PairWiseMatches out;
  matching_image_collection::Cascade_Hashing_Matcher_Regions matcher(dist_ratio);
  std::shared_ptr<sfm::Regions_Provider> base = provider;
  matcher.Match(base, pairs, out, nullptr);
  std::vector<uint32_t> flat;
  for (const auto& kv : out) {
    flat.resize(kv.second.size() * 2);
    for (size_t m = 0; m < kv.second.size(); ++m) {
      flat[2 * m] = kv.second[m].i_;
      flat[2 * m + 1] = kv.second[m].j_;
    }
    if (sink) sink(user, kv.first.first, kv.first.second, flat.data(), uint32_t(kv.second.size()));
  }
  return out.size();
}

// ... the same on in-memory AKAZE_Float_Regions (64 x float) and AKAZE_Liop_Regions (144 x uint8): the other scalar region types
// Cascade_Hashing_Matcher_Regions::Match dispatches on (Cascade_Hashing_Matcher_Regions.cpp:233-262)
extern "C++" {
template <class RegionsT, class Row>
static uint64_t cascade_match_typed(std::shared_ptr<features::Regions> (*make)(const Row*, uint32_t), const Row* const* desc_rows,
                                    const float* const* feat_xy, const uint32_t* n_desc, uint32_t n_images, const uint32_t* pairs_IJ,
                                    uint64_t n_pairs, float dist_ratio, ref_match_sink sink, void* user) {
  auto provider = std::make_shared<InMemoryRegionsProvider>();
  provider->set_type(new RegionsT());
  for (uint32_t k = 0; k < n_images; ++k) {
    std::shared_ptr<features::Regions> r = make(desc_rows[k], n_desc[k]);
    auto* sr = static_cast<RegionsT*>(r.get());
    for (uint32_t q = 0; q < n_desc[k]; ++q) sr->Features()[q] = features::SIOPointFeature(feat_xy[k][2 * q], feat_xy[k][2 * q + 1], 1.f, 0.f);
    provider->set(k, r);
  }
  Pair_Set pairs;
  for (uint64_t p = 0; p < n_pairs; ++p) pairs.insert({pairs_IJ[2 * p], pairs_IJ[2 * p + 1]});
  matching::PairWiseMatches out;
  matching_image_collection::Cascade_Hashing_Matcher_Regions matcher(dist_ratio);
  std::shared_ptr<sfm::Regions_Provider> base = provider;
  matcher.Match(base, pairs, out, nullptr);
  std::vector<uint32_t> flat;
  for (const auto& kv : out) {
    flat.resize(kv.second.size() * 2);
    for (size_t m = 0; m < kv.second.size(); ++m) { flat[2 * m] = kv.second[m].i_; flat[2 * m + 1] = kv.second[m].j_; }
    if (sink) sink(user, kv.first.first, kv.first.second, flat.data(), uint32_t(kv.second.size()));
  }
  return out.size();
}
}  // extern "C++"
uint64_t ref_cascade_matcher_regions_match_float64(const float* const* desc_rows, const float* const* feat_xy, const uint32_t* n_desc,
                                                   uint32_t n_images, const uint32_t* pairs_IJ, uint64_t n_pairs, float dist_ratio,
                                                   ref_match_sink sink, void* user) {
  return cascade_match_typed<features::AKAZE_Float_Regions, float>(&make_float64_regions, desc_rows, feat_xy, n_desc, n_images, pairs_IJ, n_pairs,
                                                                    dist_ratio, sink, user);
}
uint64_t ref_cascade_matcher_regions_match_liop144(const uint8_t* const* desc_rows, const float* const* feat_xy, const uint32_t* n_desc,
                                                   uint32_t n_images, const uint32_t* pairs_IJ, uint64_t n_pairs, float dist_ratio,
                                                   ref_match_sink sink, void* user) {
  return cascade_match_typed<features::AKAZE_Liop_Regions, uint8_t>(&make_liop144_regions, desc_rows, feat_xy, n_desc, n_images, pairs_IJ, n_pairs,
                                                                     dist_ratio, sink, user);
}

// CascadeHasher with a chosen bucket layout on ONE pair (queries = J, database = I): Init(128, n_groups, bits_per_bucket), the
// zero-mean descriptor over the two images, CreateHashedDescriptions, Match_HashedDescriptions (NN = 2) and NNdistanceRatio -
// the list before the de-duplication steps, as (index in I, index in J). The hash outputs are copied out as well (16 bytes and
// n_groups uint16 per descriptor), so that the restatement / the device stage can be run on exactly these inputs. Few bits per
// bucket put hundreds of candidates in a bucket: the regime where the top-ten selection and the repeat test do real work.
// Returns the number of matches.
uint32_t ref_cascade_match_pair_u8(const uint8_t* descI, uint32_t nI, const uint8_t* descJ, uint32_t nJ, uint32_t n_groups,
                                   uint32_t bits_per_bucket, float dist_ratio, uint8_t* hashI, uint16_t* bidsI, uint8_t* hashJ,
                                   uint16_t* bidsJ, uint32_t* out_ij /* capacity 2 * nJ */) {
  using BaseMat = Eigen::Matrix<unsigned char, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
  matching::CascadeHasher hasher;
  hasher.Init(128, uint8_t(n_groups), uint8_t(bits_per_bucket));
  Eigen::Map<BaseMat> mI(const_cast<unsigned char*>(descI), nI, 128), mJ(const_cast<unsigned char*>(descJ), nJ, 128);
  Eigen::MatrixXf per_image(2, 128);
  per_image.fill(0.0f);
  if (nI) per_image.row(0) = matching::CascadeHasher::GetZeroMeanDescriptor(mI);
  if (nJ) per_image.row(1) = matching::CascadeHasher::GetZeroMeanDescriptor(mJ);
  const Eigen::VectorXf zero_mean = matching::CascadeHasher::GetZeroMeanDescriptor(per_image);
  const matching::HashedDescriptions hI = hasher.CreateHashedDescriptions(mI, zero_mean), hJ = hasher.CreateHashedDescriptions(mJ, zero_mean);
  auto copy_out = [&](const matching::HashedDescriptions& h, uint8_t* hash, uint16_t* bids) {
    for (size_t r = 0; r < h.hashed_desc.size(); ++r) {
      std::memcpy(hash + r * 16, h.hashed_desc[r].hash_code.data(), 16);
      for (uint32_t g = 0; g < n_groups; ++g) bids[r * n_groups + g] = h.hashed_desc[r].bucket_ids[g];
    }
  };
  copy_out(hI, hashI, bidsI);
  copy_out(hJ, hashJ, bidsJ);
  matching::IndMatches nn;
  std::vector<float> dist;
  hasher.Match_HashedDescriptions<BaseMat, float>(hJ, mJ, hI, mI, &nn, &dist);
  std::vector<int> kept;
  matching::NNdistanceRatio(dist.begin(), dist.end(), 2, kept, Square(dist_ratio));
  for (size_t k = 0; k < kept.size(); ++k) {
    out_ij[2 * k] = nn[kept[k] * 2].j_;
    out_ij[2 * k + 1] = nn[kept[k] * 2].i_;
  }
  return uint32_t(kept.size());
}

}  // extern "C"
