// ref_shim_io.cpp — TEST INFRASTRUCTURE ONLY.
//
// extern "C" access to the REFERENCE's own file readers / writers, compiled in place from /root/reference/src by
// oracle/Makefile into oracle/_ref/libref_io.so, so that openmvg_amd/io.py can be checked against them:
//   features/descriptor.hpp:182-226   loadDescsFromBinFile / saveDescsToBinFile  (Descriptor<unsigned char, 128>)
//   features/feature_container.hpp    loadFeatsFromFile / saveFeatsToFile        (SIOPointFeature stream operators)
//   matching/indMatch.hpp:58-64       IndMatch stream operators; the "txt" loops of matching/indMatch_utils.cpp:28-131
//                                     are restated here around them (that TU itself needs cereal, absent from the tree)
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "openMVG/features/descriptor.hpp"
#include "openMVG/features/feature.hpp"
#include "openMVG/features/feature_container.hpp"
#include "openMVG/matching/indMatch.hpp"

using namespace openMVG;
using Desc = features::Descriptor<unsigned char, 128>;

extern "C" {

int ref_io_save_desc(const char* path, const uint8_t* rows, uint64_t n) {
  std::vector<Desc, Eigen::aligned_allocator<Desc>> v(n);
  for (uint64_t k = 0; k < n; ++k) std::memcpy(v[k].data(), rows + k * 128, 128);
  return features::saveDescsToBinFile(path, v) ? 0 : 1;
}

// rows may be null to query the count
int64_t ref_io_load_desc(const char* path, uint8_t* rows, uint64_t cap) {
  std::vector<Desc, Eigen::aligned_allocator<Desc>> v;
  if (!features::loadDescsFromBinFile(path, v)) return -1;
  if (rows) for (uint64_t k = 0; k < v.size() && k < cap; ++k) std::memcpy(rows + k * 128, v[k].data(), 128);
  return (int64_t)v.size();
}

int ref_io_save_feat(const char* path, const float* xyso, uint64_t n) {
  features::SIOPointFeatures v;
  for (uint64_t k = 0; k < n; ++k) v.emplace_back(xyso[4 * k], xyso[4 * k + 1], xyso[4 * k + 2], xyso[4 * k + 3]);
  return features::saveFeatsToFile(path, v) ? 0 : 1;
}

int64_t ref_io_load_feat(const char* path, float* xyso, uint64_t cap) {
  features::SIOPointFeatures v;
  if (!features::loadFeatsFromFile(path, v)) return -1;
  if (xyso) for (uint64_t k = 0; k < v.size() && k < cap; ++k) {
    xyso[4 * k] = v[k].x(); xyso[4 * k + 1] = v[k].y(); xyso[4 * k + 2] = v[k].scale(); xyso[4 * k + 3] = v[k].orientation();
  }
  return (int64_t)v.size();
}

// pairs: n_pairs x 2, offsets: n_pairs + 1, ij: matches x 2 (all uint32 / uint64 as in mvgx_match_results)
int ref_io_save_matches_txt(const char* path, const uint32_t* pairs, uint64_t n_pairs, const uint64_t* offsets, const uint32_t* ij) {
  matching::PairWiseMatches m;
  for (uint64_t k = 0; k < n_pairs; ++k) {
    if (offsets[k + 1] == offsets[k]) continue;
    matching::IndMatches v;
    for (uint64_t q = offsets[k]; q < offsets[k + 1]; ++q) v.emplace_back(ij[2 * q], ij[2 * q + 1]);
    m[{pairs[2 * k], pairs[2 * k + 1]}] = std::move(v);
  }
  std::ofstream stream(path);
  if (!stream) return 1;
  for (const auto& cur_match : m) {   // the "txt" branch of matching::Save
    stream << cur_match.first.first << " " << cur_match.first.second << '\n' << cur_match.second.size() << '\n';
    std::copy(cur_match.second.cbegin(), cur_match.second.cend(), std::ostream_iterator<matching::IndMatch>(stream, "\n"));
  }
  return stream ? 0 : 1;
}

// the "txt" branch of matching::Load; returns the number of pairs, fills up to cap matches (I, J, i, j) per row
int64_t ref_io_load_matches_txt(const char* path, uint32_t* rows, uint64_t cap, uint64_t* n_matches) {
  std::ifstream stream(path);
  if (!stream) return -1;
  size_t I, J, number;
  uint64_t total = 0;
  int64_t npairs = 0;
  while (stream >> I >> J >> number) {
    for (size_t i = 0; i < number; ++i) {
      matching::IndMatch im;
      stream >> im;
      if (rows && total < cap) { rows[4 * total] = (uint32_t)I; rows[4 * total + 1] = (uint32_t)J; rows[4 * total + 2] = im.i_; rows[4 * total + 3] = im.j_; }
      ++total;
    }
    ++npairs;
  }
  *n_matches = total;
  return npairs;
}

}  // extern "C"
