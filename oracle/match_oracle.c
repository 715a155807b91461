/*
 * match_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into, imported by, or called from the product path).
 *
 * Plain-C restatement of openMVG's brute-force L2 + distance-ratio matching path, used as the parity checker
 * for libmvgx_hip.so. Every function cites the reference lines (under /root/reference/src/openMVG) it follows.
 * Pinned against the reference itself: tests/test_oracle_matching.py checks it against the golden values of the
 * reference's own unit tests (metric_test.cpp:31-39,132-147; matching_test.cpp:26-87) and, when oracle/_ref is
 * built, against the reference's compiled ArrayMatcherBruteForce / Matcher_Regions on random and adversarial input.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include <limits.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* matching/metric.hpp:55-93 — L2<uint8_t>::operator(): ResultType = int (accumulator_trait.hpp:15-32),
 * sum of squared differences, 4-way unrolled then tail; integer arithmetic, so order does not matter. */
int oracle_l2_u8(const uint8_t* a, const uint8_t* b, size_t size) {
  int result = 0;
  for (size_t k = 0; k < size; ++k) {
    const int diff = (int)a[k] - (int)b[k];
    result += diff * diff;
  }
  return result;
}

/* Same metric for the other integral / floating instantiations exercised by metric_test.cpp:31-39. */
int oracle_l2_i32(const int32_t* a, const int32_t* b, size_t size) {
  int result = 0;
  for (size_t k = 0; k < size; ++k) {
    const int diff = a[k] - b[k];
    result += diff * diff;
  }
  return result;
}
float oracle_l2_f32(const float* a, const float* b, size_t size) {
  /* metric.hpp:60-92: processes 4 at a time accumulating diff0^2+diff1^2+diff2^2+diff3^2 into result */
  float result = 0.f;
  size_t k = 0;
  for (; k + 4 <= size; k += 4) {
    const float d0 = a[k] - b[k], d1 = a[k + 1] - b[k + 1], d2 = a[k + 2] - b[k + 2], d3 = a[k + 3] - b[k + 3];
    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  for (; k < size; ++k) {
    const float d0 = a[k] - b[k];
    result += d0 * d0;
  }
  return result;
}

/* matching/matcher_brute_force.hpp:163-200 — SearchNeighbours_func for NN neighbours:
 * all nI distances, then the NN smallest in ascending order (stl/indexed_sort.hpp:48-63, std::partial_sort on
 * (val, index) packets comparing val only). Ties between equal distances are implementation-defined in the
 * reference (heap order); this restatement breaks them by ascending database index. For NN=2 and ratio <= 1
 * the emitted matches do not depend on that order (an accepted query has d0 < d1 strictly).
 * Returns 0 (and writes nothing) in the reference's early-out cases, matcher_brute_force.hpp:108-113. */
int oracle_search_neighbours_u8(const uint8_t* db, int nI, const uint8_t* queries, int nJ, int dim, int NN,
                                int32_t* out_index /* nJ*NN */, int32_t* out_dist /* nJ*NN */) {
  if (db == NULL || nI < 1 || NN > nI || nJ < 1) return 0;
#pragma omp parallel for schedule(static)
  for (int q = 0; q < nJ; ++q) {
    int32_t* bi = out_index + (size_t)q * NN;
    int32_t* bd = out_dist + (size_t)q * NN;
    int filled = 0;
    for (int i = 0; i < nI; ++i) {
      const int d = oracle_l2_u8(queries + (size_t)q * dim, db + (size_t)i * dim, (size_t)dim);
      /* insertion into the sorted prefix of length <= NN; strict '<' keeps the earlier index first on ties */
      int pos = filled;
      while (pos > 0 && d < bd[pos - 1]) --pos;
      if (pos >= NN) continue;
      const int last = filled < NN ? filled : NN - 1;
      for (int k = last; k > pos; --k) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; }
      bd[pos] = d; bi[pos] = i;
      if (filled < NN) ++filled;
    }
  }
  return 1;
}

/* matching/matching_filters.hpp:39-60 — NNdistanceRatio on int distances: `(*iter) < fratio * (*iter2)` is
 * evaluated as (float)d0 < fratio * (float)d1 in binary32 (usual arithmetic conversions; x86-64 SSE, no x87). */
static int ratio_ok(int d0, int d1, float fratio) {
  volatile float rhs = fratio * (float)d1; /* volatile: forbid contraction / excess precision */
  return (float)d0 < rhs;
}

/* matching/regions_matcher.hpp:162-207 — RegionsMatcherT::MatchDistanceRatio with b_squared_metric_ = true
 * (regions_matcher.cpp:75-81): 2-NN search, ratio filter with Square(distance_ratio) (numeric.h:56: x*x in
 * float), emit IndMatch(i_ = index in I (database), j_ = index in J (query)) in ascending j.
 * out_ij must hold 2*nJ uint32. Returns the number of matches. */
uint32_t oracle_match_distance_ratio_u8(const uint8_t* dbI, int nI, const uint8_t* qJ, int nJ, int dim,
                                        float distance_ratio, uint32_t* out_ij) {
  if (nJ < 1 || nI < 2) return 0; /* SearchNeighbours returns false -> no matches */
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)nJ);
  int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)nJ);
  uint32_t n = 0;
  if (oracle_search_neighbours_u8(dbI, nI, qJ, nJ, dim, 2, idx, dist)) {
    const float fratio = distance_ratio * distance_ratio; /* Square(distance_ratio), float */
    for (int q = 0; q < nJ; ++q) {
      if (ratio_ok(dist[2 * q], dist[2 * q + 1], fratio)) {
        out_ij[2 * n] = (uint32_t)idx[2 * q];
        out_ij[2 * n + 1] = (uint32_t)q;
        ++n;
      }
    }
  }
  free(idx);
  free(dist);
  return n;
}

/* matching_image_collection/Matcher_Regions.cpp:32-107 — Matcher_Regions::Match for BRUTE_FORCE_L2 on uint8
 * 128-D regions. For every input pair (I, J) (processed grouped by I; the grouping does not change results):
 *   I empty -> skipped (:65-69); J empty -> skipped (:85-90); otherwise MatchDistanceRatio; the pair is reported
 *   only if it has matches (:99-102).
 * Output: offsets[n_pairs+1] (match counts prefix, in input pair order), ij (2 uint32 per match), capacity in
 * matches; returns total matches or (uint64_t)-1 if capacity is too small. */
uint64_t oracle_matcher_regions_match_u8(const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                                         uint32_t dim, const uint32_t* pairs_IJ, uint64_t n_pairs,
                                         float distance_ratio, uint64_t* offsets, uint32_t* ij, uint64_t capacity) {
  /* pairs are independent (the reference itself runs the J loop under OpenMP, Matcher_Regions.cpp:80): evaluate them in
   * parallel into per-pair lists (the 2-NN search called from inside a parallel region runs single-threaded), then
   * concatenate in input order. */
  uint32_t** lists = (uint32_t**)calloc(n_pairs ? n_pairs : 1, sizeof(uint32_t*));
  uint32_t* counts = (uint32_t*)calloc(n_pairs ? n_pairs : 1, sizeof(uint32_t));
  if (!lists || !counts) { free(lists); free(counts); return (uint64_t)-1; }
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t p = 0; p < (int64_t)n_pairs; ++p) {
    const uint32_t I = pairs_IJ[2 * p], J = pairs_IJ[2 * p + 1];
    if (I < n_images && J < n_images && n_desc[I] != 0 && n_desc[J] != 0) {
      lists[p] = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)n_desc[J]);
      if (lists[p])
        counts[p] = oracle_match_distance_ratio_u8(desc_rows[I], (int)n_desc[I], desc_rows[J], (int)n_desc[J], (int)dim,
                                                   distance_ratio, lists[p]);
    }
  }
  uint64_t total = 0;
  int overflow = 0;
  offsets[0] = 0;
  for (uint64_t p = 0; p < n_pairs; ++p) {
    if (!overflow && total + counts[p] > capacity) overflow = 1;
    if (!overflow && counts[p]) memcpy(ij + 2 * total, lists[p], sizeof(uint32_t) * 2 * (size_t)counts[p]);
    total += counts[p];
    offsets[p + 1] = total;
    free(lists[p]);
  }
  free(lists);
  free(counts);
  return overflow ? (uint64_t)-1 : total;
}

/* ---- binary descriptors: BRUTE_FORCE_HAMMING (matching/regions_matcher.cpp:184-191) ----
 * matching/metric_hamming.hpp:36-107 — Hamming<unsigned char>::operator(): ResultType = unsigned int, popcount of the XOR
 * over `size` bytes (the reference walks 8-, 4- or 1-byte words depending on size; the sum is the same). */
unsigned int oracle_hamming_u8(const uint8_t* a, const uint8_t* b, size_t size) {
  unsigned int result = 0;
  for (size_t k = 0; k < size; ++k) {
    unsigned int x = (unsigned int)(a[k] ^ b[k]);
    while (x) { result += x & 1u; x >>= 1; }
  }
  return result;
}

/* matching/regions_matcher.hpp:162-207 with b_squared_metric_ == false (regions_matcher.cpp:189): 2-NN search
 * (matcher_brute_force.hpp:95-200), then NNdistanceRatio (matching_filters.hpp:39-60) with the ratio as given:
 * `(*iter) < fratio * (*iter2)` on unsigned int distances = (float)d0 < fratio * (float)d1 in binary32.
 * Ties in the 2-NN are broken by ascending database index (irrelevant for ratio <= 1, see oracle_search_neighbours_u8). */
uint32_t oracle_match_distance_ratio_hamming(const uint8_t* dbI, int nI, const uint8_t* qJ, int nJ, int bytes,
                                             float distance_ratio, uint32_t* out_ij) {
  if (nJ < 1 || nI < 2) return 0;
  uint32_t n = 0;
  for (int q = 0; q < nJ; ++q) {
    unsigned int d0 = 0xFFFFFFFFu, d1 = 0xFFFFFFFFu;
    int i0 = -1;
    for (int i = 0; i < nI; ++i) {
      const unsigned int d = oracle_hamming_u8(qJ + (size_t)q * bytes, dbI + (size_t)i * bytes, (size_t)bytes);
      if (d < d0) { d1 = d0; d0 = d; i0 = i; }
      else if (d < d1) { d1 = d; }
    }
    volatile float rhs = distance_ratio * (float)d1;
    if ((float)d0 < rhs) {
      out_ij[2 * n] = (uint32_t)i0;
      out_ij[2 * n + 1] = (uint32_t)q;
      ++n;
    }
  }
  return n;
}

/* matching_image_collection/Matcher_Regions.cpp:32-107 for BRUTE_FORCE_HAMMING on binary regions of `bytes` bytes; same
 * output convention as oracle_matcher_regions_match_u8. */
uint64_t oracle_matcher_regions_match_hamming(const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                                              uint32_t bytes, const uint32_t* pairs_IJ, uint64_t n_pairs,
                                              float distance_ratio, uint64_t* offsets, uint32_t* ij, uint64_t capacity) {
  uint32_t** lists = (uint32_t**)calloc(n_pairs ? n_pairs : 1, sizeof(uint32_t*));
  uint32_t* counts = (uint32_t*)calloc(n_pairs ? n_pairs : 1, sizeof(uint32_t));
  if (!lists || !counts) { free(lists); free(counts); return (uint64_t)-1; }
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t p = 0; p < (int64_t)n_pairs; ++p) {
    const uint32_t I = pairs_IJ[2 * p], J = pairs_IJ[2 * p + 1];
    if (I < n_images && J < n_images && n_desc[I] != 0 && n_desc[J] != 0) {
      lists[p] = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)n_desc[J]);
      if (lists[p])
        counts[p] = oracle_match_distance_ratio_hamming(desc_rows[I], (int)n_desc[I], desc_rows[J], (int)n_desc[J], (int)bytes,
                                                        distance_ratio, lists[p]);
    }
  }
  uint64_t total = 0;
  int overflow = 0;
  offsets[0] = 0;
  for (uint64_t p = 0; p < n_pairs; ++p) {
    if (!overflow && total + counts[p] > capacity) overflow = 1;
    if (!overflow && counts[p]) memcpy(ij + 2 * total, lists[p], sizeof(uint32_t) * 2 * (size_t)counts[p]);
    total += counts[p];
    offsets[p + 1] = total;
    free(lists[p]);
  }
  free(lists);
  free(counts);
  return overflow ? (uint64_t)-1 : total;
}

/* ---- float descriptors: BRUTE_FORCE_L2 on Scalar_Regions<float> (matching/regions_matcher.cpp:119-124) ----
 * L2<float> is oracle_l2_f32 above (metric.hpp:98-135; dimensions that are a multiple of 8 take its scalar loop). 2-NN +
 * NNdistanceRatio with Square(distance_ratio) on float distances (regions_matcher.hpp:162-207, matching_filters.hpp:39-60).
 * This translation unit must be compiled without floating-point contraction and without -ffast-math (oracle/Makefile). */
uint32_t oracle_match_distance_ratio_f32(const float* dbI, int nI, const float* qJ, int nJ, int dim, float distance_ratio,
                                         uint32_t* out_ij) {
  if (nJ < 1 || nI < 2) return 0;
  const float fratio = distance_ratio * distance_ratio;
  uint32_t n = 0;
  for (int q = 0; q < nJ; ++q) {
    float d0 = 0.f, d1 = 0.f;
    int i0 = -1, have = 0;
    for (int i = 0; i < nI; ++i) {
      const float d = oracle_l2_f32(qJ + (size_t)q * dim, dbI + (size_t)i * dim, (size_t)dim);
      if (have == 0) { d0 = d; i0 = i; have = 1; }
      else if (d < d0) { d1 = d0; d0 = d; i0 = i; have = 2; }
      else if (have == 1 || d < d1) { d1 = d; have = 2; }
    }
    volatile float rhs = fratio * d1;
    if (d0 < rhs) {
      out_ij[2 * n] = (uint32_t)i0;
      out_ij[2 * n + 1] = (uint32_t)q;
      ++n;
    }
  }
  return n;
}

uint64_t oracle_matcher_regions_match_f32(const float* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                                          uint32_t dim, const uint32_t* pairs_IJ, uint64_t n_pairs, float distance_ratio,
                                          uint64_t* offsets, uint32_t* ij, uint64_t capacity) {
  uint32_t** lists = (uint32_t**)calloc(n_pairs ? n_pairs : 1, sizeof(uint32_t*));
  uint32_t* counts = (uint32_t*)calloc(n_pairs ? n_pairs : 1, sizeof(uint32_t));
  if (!lists || !counts) { free(lists); free(counts); return (uint64_t)-1; }
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t p = 0; p < (int64_t)n_pairs; ++p) {
    const uint32_t I = pairs_IJ[2 * p], J = pairs_IJ[2 * p + 1];
    if (I < n_images && J < n_images && n_desc[I] != 0 && n_desc[J] != 0) {
      lists[p] = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)n_desc[J]);
      if (lists[p])
        counts[p] = oracle_match_distance_ratio_f32(desc_rows[I], (int)n_desc[I], desc_rows[J], (int)n_desc[J], (int)dim,
                                                    distance_ratio, lists[p]);
    }
  }
  uint64_t total = 0;
  int overflow = 0;
  offsets[0] = 0;
  for (uint64_t p = 0; p < n_pairs; ++p) {
    if (!overflow && total + counts[p] > capacity) overflow = 1;
    if (!overflow && counts[p]) memcpy(ij + 2 * total, lists[p], sizeof(uint32_t) * 2 * (size_t)counts[p]);
    total += counts[p];
    offsets[p + 1] = total;
    free(lists[p]);
  }
  free(lists);
  free(counts);
  return overflow ? (uint64_t)-1 : total;
}

/* ------------------------------------------------------------------------------------------------------------------
 * CASCADE_HASHING_L2, the matching stage (the hashing stage - single-precision Eigen products - stays the reference's own
 * code: its outputs, the per-descriptor hash code and bucket ids, are inputs here).
 * Restates  matching/cascade_hasher.hpp:241-367  CascadeHasher::Match_HashedDescriptions (queries = descriptors of J,
 * database = descriptors of I, NN = 2, kNumTopCandidates = 10) followed by  matching_image_collection/
 * Cascade_Hashing_Matcher_Regions.cpp:196-215  (NNdistanceRatio with Square(ratio), IndMatch(database id, query id)).
 * The two de-duplication steps that follow in the reference (:218-226) work on feature coordinates and are host code
 * of the caller; this function returns the list BEFORE them, in ascending query order.
 *   hash: n x hash_bytes (stl::dynamic_bitset blocks, unsigned char);  bids: n x n_groups uint16 bucket ids
 * ------------------------------------------------------------------------------------------------------------------ */
static unsigned popcount8(unsigned v) { unsigned c = 0; while (v) { c += v & 1u; v >>= 1; } return c; }

uint32_t oracle_cascade_match_pair_u8(const uint8_t* descI, const uint8_t* hashI, const uint16_t* bidsI, uint32_t nI,
                                      const uint8_t* descJ, const uint8_t* hashJ, const uint16_t* bidsJ, uint32_t nJ,
                                      uint32_t dim, uint32_t hash_bytes, uint32_t n_groups, uint32_t bits_per_bucket,
                                      float ratio_sq, uint32_t* out_ij /* capacity 2 * nJ */) {
  const uint32_t n_buckets = 1u << bits_per_bucket, n_hash_bits = hash_bytes * 8;
  /* buckets of the database image: group -> bucket -> ascending descriptor ids (cascade_hasher.hpp:222-236) */
  uint32_t* start = (uint32_t*)calloc((size_t)n_groups * n_buckets + 1, sizeof(uint32_t));
  uint32_t* items = (uint32_t*)malloc(((size_t)n_groups * nI + 1) * sizeof(uint32_t));
  for (uint32_t g = 0; g < n_groups; ++g)
    for (uint32_t k = 0; k < nI; ++k) start[(size_t)g * n_buckets + bidsI[(size_t)k * n_groups + g] + 1]++;
  for (size_t b = 0; b < (size_t)n_groups * n_buckets; ++b) start[b + 1] += start[b];
  {
    uint32_t* fill = (uint32_t*)malloc((size_t)n_groups * n_buckets * sizeof(uint32_t));
    memcpy(fill, start, (size_t)n_groups * n_buckets * sizeof(uint32_t));
    for (uint32_t g = 0; g < n_groups; ++g)
      for (uint32_t k = 0; k < nI; ++k) items[fill[(size_t)g * n_buckets + bidsI[(size_t)k * n_groups + g]]++] = k;
    free(fill);
  }
  uint32_t* cand = (uint32_t*)malloc(((size_t)n_groups * nI + 1) * sizeof(uint32_t));
  uint8_t* used = (uint8_t*)malloc(nI + 1);
  uint32_t* by_ham = (uint32_t*)malloc(((size_t)(n_hash_bits + 1) * nI + 1) * sizeof(uint32_t));   /* column h: ids at distance h */
  uint32_t* n_ham = (uint32_t*)malloc((n_hash_bits + 1) * sizeof(uint32_t));
  uint32_t n_out = 0;
  for (uint32_t q = 0; q < nJ; ++q) {
    size_t nc = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
      const size_t b = (size_t)g * n_buckets + bidsJ[(size_t)q * n_groups + g];
      for (uint32_t e = start[b]; e < start[b + 1]; ++e) { cand[nc++] = items[e]; used[items[e]] = 0; }
    }
    if (nc <= 2) continue;   /* "not at least NN candidates": the raw count, duplicates included (:283) */
    memset(n_ham, 0, (n_hash_bits + 1) * sizeof(uint32_t));
    for (size_t c = 0; c < nc; ++c) {
      const uint32_t id = cand[c];
      if (used[id]) continue;
      used[id] = 1;
      unsigned h = 0;
      for (uint32_t w = 0; w < hash_bytes; ++w) h += popcount8((unsigned)(hashJ[(size_t)q * hash_bytes + w] ^ hashI[(size_t)id * hash_bytes + w]));
      by_ham[(size_t)h * nI + n_ham[h]++] = id;
    }
    /* the (up to) ten candidates of smallest Hamming distance, ties in order of first appearance (:338-352) */
    float dist[10];
    uint32_t ids[10];
    int nt = 0;
    for (uint32_t h = 0; h <= n_hash_bits && nt < 10; ++h)
      for (uint32_t k = 0; k < n_ham[h] && nt < 10; ++k) {
        const uint32_t id = by_ham[(size_t)h * nI + k];
        dist[nt] = (float)oracle_l2_u8(descI + (size_t)id * dim, descJ + (size_t)q * dim, dim);   /* DistanceType = float (:185-188) */
        ids[nt++] = id;
      }
    if (nt < 2) continue;
    /* std::partial_sort of (distance, id) pairs, first two: a total order, so "the two smallest pairs" (:356-365) */
    int i0 = 0, i1 = -1;
    for (int k = 1; k < nt; ++k)
      if (dist[k] < dist[i0] || (dist[k] == dist[i0] && ids[k] < ids[i0])) i0 = k;
    for (int k = 0; k < nt; ++k) {
      if (k == i0) continue;
      if (i1 < 0 || dist[k] < dist[i1] || (dist[k] == dist[i1] && ids[k] < ids[i1])) i1 = k;
    }
    if (dist[i0] < ratio_sq * dist[i1]) {   /* NNdistanceRatio, matching_filters.hpp:39-60 */
      out_ij[2 * n_out] = ids[i0];
      out_ij[2 * n_out + 1] = q;
      ++n_out;
    }
  }
  free(start); free(items); free(cand); free(used); free(by_ham); free(n_ham);
  return n_out;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
