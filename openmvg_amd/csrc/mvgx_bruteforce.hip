// mvgx_bruteforce.hip — the other brute-force instantiations behind RegionMatcherFactory on gfx950:
//   Hamming 2-NN + distance-ratio matching of binary descriptors, and L2<float> on float descriptors (second half of the file).
//
// Replaces, for binary regions (features::Binary_Regions<SIOPointFeature, 64> = AKAZE_Binary_Regions,
// features/regions_factory.hpp:26) behind the same factory as the L2 path:
//   matching/regions_matcher.cpp:184-191   BRUTE_FORCE_HAMMING -> RegionsMatcherT<ArrayMatcherBruteForce<uchar, Hamming<uchar>>>(regions, false)
//   matching/metric_hamming.hpp:36-107     Hamming<T>: ResultType unsigned int, popcount of the XOR
//   matching/matcher_brute_force.hpp:95-200 SearchNeighbours (all distances, the NN smallest)
//   matching/regions_matcher.hpp:162-207   MatchDistanceRatio with b_squared_metric_ == false: the ratio is used as given
//   matching/matching_filters.hpp:39-60    NNdistanceRatio: (float)d0 < ratio * (float)d1
//
// Formulation. Every lane owns one query descriptor of image J in registers (NW dwords). The database image I is not staged
// at all: the row index is wave-uniform, so the rows arrive through the scalar data path (s_load_dwordx8/16 from the
// constant cache) and are XORed as SGPR operands; per row and lane NW x (v_xor + v_bcnt accumulate) and three VALU for the
// running top-2 on packed keys (distance << 22 | row): best0 = min(best0, key), best1 = med3(best0, best1, key).
// Keys order equal distances by row, so best0 is the first minimum; for ratio <= 1 an accepted query has d0 < d1 strictly
// and the emitted index is the unique argmin (the reference's partial_sort tie order cannot matter).
// Bound: VALU issue (2 NW + 3 instructions per descriptor pair and lane); HBM traffic is negligible (every image is read
// once per work item from L2 / the scalar cache).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "mvgx_common.h"

namespace {

using mvgx::set_error;

constexpr int kQBlock = 256;                 // queries per workgroup
constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr uint32_t kRowBits = 22;            // rows per image < 2^22 (distance <= 512 fits the upper 10 bits)

struct HamParams {
  const uint32_t* words;        // all descriptors, NW dwords each (zero padded), image-major
  const uint64_t* img_row_off;  // first row of image k
  const uint32_t* img_n;        // rows of image k
  const uint2* pairs;           // (I, J) per pair of the batch
  const uint2* work;            // (pair in batch, first query)
  uint32_t* best;               // [pair][query] -> database row or kInvalid
  uint32_t* count;              // matches per pair
  uint32_t qstride;
  float ratio;
};

__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }

template <int NW>
__global__ __launch_bounds__(kQBlock) void hamming_top2_ratio_kernel(HamParams p) {
  __shared__ uint32_t sh_count;
  const uint2 w = p.work[blockIdx.x];
  const uint2 ij = p.pairs[w.x];
  const uint32_t nI = p.img_n[ij.x], nJ = p.img_n[ij.y];
  const uint32_t q = w.y + threadIdx.x;
  const bool active = q < nJ;
  if (threadIdx.x == 0) sh_count = 0;
  __syncthreads();
  uint32_t qv[NW];
  {
    const uint32_t* src = p.words + (p.img_row_off[ij.y] + (active ? q : 0)) * NW;
#pragma unroll
    for (int k = 0; k < NW; ++k) qv[k] = src[k];
  }
  const uint32_t* __restrict__ db = p.words + p.img_row_off[ij.x] * NW;   // wave-uniform: scalar loads
  uint32_t b0 = kInvalid, b1 = kInvalid;
  // rows in groups of four: the scalar loads of a group are issued back to back and their latency is hidden by the VALU work
  // of the other waves on the SIMD (35 VGPRs: the occupancy is bounded by the workgroup size, not by registers)
#pragma unroll 4
  for (uint32_t i = 0; i < nI; ++i) {
    const uint32_t* __restrict__ row = db + (size_t)i * NW;
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) d += __popc(row[k] ^ qv[k]);
    const uint32_t key = (d << kRowBits) | i;
    const uint32_t lo = umin32(b0, key), hi = umax32(b0, key);
    b1 = umin32(b1, hi);   // = med3(b0, b1, key) for b0 <= b1
    b0 = lo;
  }
  uint32_t out = kInvalid;
  if (active && nI >= 2) {   // matcher_brute_force.hpp:108-113: NN (= 2) > rows -> no result
    const float d0 = (float)(b0 >> kRowBits), d1 = (float)(b1 >> kRowBits);
    if (d0 < __fmul_rn(p.ratio, d1)) out = b0 & ((1u << kRowBits) - 1u);
  }
  if (active) p.best[(size_t)w.x * p.qstride + q] = out;
  if (out != kInvalid) atomicAdd(&sh_count, 1u);
  __syncthreads();
  if (threadIdx.x == 0 && sh_count) atomicAdd(&p.count[w.x], sh_count);
}

// ---- L2<float> (matching/metric.hpp:98-135) on Scalar_Regions<SIOPointFeature, float, DIM> (AKAZE_Float_Regions: 64) ----
// regions_matcher.cpp:119-124: ArrayMatcherBruteForce<float, L2<float>>, squared metric -> ratio test with Square(ratio).
// Bit-exactness: float addition is not associative, so the kernel performs the reference's operations in the reference's
// order - per group of four elements ((d0 d0 + d1 d1) + d2 d2) + d3 d3, then result += group - with separate IEEE
// multiplies and adds (fp contraction off: an FMA would round differently). The reference takes this scalar route for every
// dimension that is a multiple of 8 (metric.hpp:107-112 sends only `size % 8 != 0` to its AVX variant); other dimensions
// are refused by the C ABI. Same structure as the Hamming kernel: query in registers, database rows through the scalar
// path, two rows per step in the two halves of packed-fp32 instructions (rows stored pairwise interleaved). Top-2 on 64-bit keys (distance bits << 32 | row):
// a sum of squares is >= +0, and non-negative floats order like their bit patterns.
typedef float f2 __attribute__((ext_vector_type(2)));

struct L2fParams {
  const float* rows;            // all descriptors, DIM floats each, image-major
  const uint64_t* img_row_off;
  const uint32_t* img_n;
  const uint2* pairs;
  const uint2* work;
  uint32_t* best;
  uint32_t* count;
  uint32_t qstride;
  float ratio_sq;
};

__device__ __forceinline__ void top2_u64(uint64_t key, uint64_t& b0, uint64_t& b1) {
  const uint64_t lo = key < b0 ? key : b0, hi = key < b0 ? b0 : key;
  b1 = hi < b1 ? hi : b1;
  b0 = lo;
}

template <int DIM>
__global__ __launch_bounds__(kQBlock) void l2f_top2_ratio_kernel(L2fParams p) {
#pragma clang fp contract(off)
  __shared__ uint32_t sh_count;
  const uint2 w = p.work[blockIdx.x];
  const uint2 ij = p.pairs[w.x];
  const uint32_t nI = p.img_n[ij.x], nJ = p.img_n[ij.y];
  const uint32_t q = w.y + threadIdx.x;
  const bool active = q < nJ;
  if (threadIdx.x == 0) sh_count = 0;
  __syncthreads();
  // layout: rows of an image in pairs, element k of rows 2m and 2m+1 adjacent (an image holds an even number of rows, the
  // pad row is zero): one 64-bit scalar operand feeds both halves of a packed-fp32 instruction
  float qv[DIM];
  {
    const uint32_t qq = active ? q : 0;
    const float* src = p.rows + (p.img_row_off[ij.y] + (qq & ~1u)) * DIM + (qq & 1u);
#pragma unroll
    for (int k = 0; k < DIM; ++k) qv[k] = src[2 * k];
  }
  const f2* __restrict__ db = reinterpret_cast<const f2*>(p.rows + p.img_row_off[ij.x] * DIM);   // wave-uniform: scalar loads
  uint64_t b0 = ~0ull, b1 = ~0ull;
  for (uint32_t i = 0; i < nI; i += 2) {   // rows i (x half) and i + 1 (y half)
    const f2* __restrict__ rp = db + (size_t)(i >> 1) * DIM;
    f2 result = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < DIM; k += 4) {
      const f2 d0 = (f2){qv[k], qv[k]} - rp[k];
      const f2 d1 = (f2){qv[k + 1], qv[k + 1]} - rp[k + 1];
      const f2 d2 = (f2){qv[k + 2], qv[k + 2]} - rp[k + 2];
      const f2 d3 = (f2){qv[k + 3], qv[k + 3]} - rp[k + 3];
      const f2 g = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
      result = result + g;
    }
    top2_u64(((uint64_t)__float_as_uint(result.x) << 32) | i, b0, b1);
    const uint64_t k1 = i + 1 < nI ? (((uint64_t)__float_as_uint(result.y) << 32) | (i + 1)) : ~0ull;   // pad row: never a neighbour
    top2_u64(k1, b0, b1);
  }
  uint32_t out = kInvalid;
  if (active && nI >= 2) {
    const float d0 = __uint_as_float((uint32_t)(b0 >> 32)), d1 = __uint_as_float((uint32_t)(b1 >> 32));
    if (d0 < p.ratio_sq * d1) out = (uint32_t)b0;   // matching_filters.hpp:39-60 on float distances
  }
  if (active) p.best[(size_t)w.x * p.qstride + q] = out;
  if (out != kInvalid) atomicAdd(&sh_count, 1u);
  __syncthreads();
  if (threadIdx.x == 0 && sh_count) atomicAdd(&p.count[w.x], sh_count);
}

// ---- L2<unsigned char> for descriptor lengths other than 128 (AKAZE_Liop_Regions: 144, regions_factory.hpp:24) ----
// regions_matcher.cpp:75-81: ArrayMatcherBruteForce<unsigned char, L2<unsigned char>>, squared metric. L2<uint8_t>
// (metric.hpp:55-93) accumulates int: d = |a|^2 + |b|^2 - 2 a.b exactly, with a.b from v_dot4_u32_u8 on the raw bytes and
// the squared norms precomputed per row. Same structure as the kernels above (query in registers, database rows and their
// norms through the scalar path); the 128-byte SIFT case keeps its own MFMA path (mvgx_match.hip) - this one exists so that
// the other uint8 region types do not fall back to the host.
struct L2uParams {
  const uint32_t* words;        // all descriptors, NW dwords each (zero padded), image-major
  const uint32_t* norms;        // |a|^2 per row
  const uint64_t* img_row_off;
  const uint32_t* img_n;
  const uint2* pairs;
  const uint2* work;
  uint32_t* best;
  uint32_t* count;
  uint32_t qstride;
  float ratio_sq;
};

template <int NW>
__global__ __launch_bounds__(kQBlock) void l2u8_top2_ratio_kernel(L2uParams p) {
  __shared__ uint32_t sh_count;
  const uint2 w = p.work[blockIdx.x];
  const uint2 ij = p.pairs[w.x];
  const uint32_t nI = p.img_n[ij.x], nJ = p.img_n[ij.y];
  const uint32_t q = w.y + threadIdx.x;
  const bool active = q < nJ;
  if (threadIdx.x == 0) sh_count = 0;
  __syncthreads();
  uint32_t qv[NW];
  const uint64_t qrow = p.img_row_off[ij.y] + (active ? q : 0);
  {
    const uint32_t* src = p.words + qrow * NW;
#pragma unroll
    for (int k = 0; k < NW; ++k) qv[k] = src[k];
  }
  const uint32_t qn = p.norms[qrow];
  const uint32_t* __restrict__ db = p.words + p.img_row_off[ij.x] * NW;   // wave-uniform: scalar loads
  const uint32_t* __restrict__ dn = p.norms + p.img_row_off[ij.x];
  uint64_t b0 = ~0ull, b1 = ~0ull;
#pragma unroll 2
  for (uint32_t i = 0; i < nI; ++i) {
    const uint32_t* __restrict__ row = db + (size_t)i * NW;
    uint32_t dot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) dot = __builtin_amdgcn_udot4(row[k], qv[k], dot, false);
    const uint32_t d = dn[i] + qn - 2u * dot;   // exact: every term < 2^24 for lengths up to 256
    top2_u64(((uint64_t)d << 32) | i, b0, b1);
  }
  uint32_t out = kInvalid;
  if (active && nI >= 2) {
    const float d0 = (float)(uint32_t)(b0 >> 32), d1 = (float)(uint32_t)(b1 >> 32);
    if (d0 < __fmul_rn(p.ratio_sq, d1)) out = (uint32_t)b0;
  }
  if (active) p.best[(size_t)w.x * p.qstride + q] = out;
  if (out != kInvalid) atomicAdd(&sh_count, 1u);
  __syncthreads();
  if (threadIdx.x == 0 && sh_count) atomicAdd(&p.count[w.x], sh_count);
}

// ------------------------------------------------------------------------------------------------------------------
// CASCADE_HASHING_L2 (the default -n of main_ComputeMatches for scalar regions), matching stage on the device.
// Replaces matching/cascade_hasher.hpp:241-367 (CascadeHasher::Match_HashedDescriptions, NN = 2, kNumTopCandidates = 10) +
// matching_image_collection/Cascade_Hashing_Matcher_Regions.cpp:196-215 (NNdistanceRatio with Square(ratio)). The HASHING
// stage (hash code and bucket ids of every descriptor: single-precision Eigen products, cascade_hasher.hpp:176-220) stays
// host code of the caller - the openMVG adapter runs the reference's own CascadeHasher for it - so everything here is
// integer work on given codes and the lists are bit-identical:
//   one lane = one query descriptor of J. Candidates = the database descriptors of I in the query's bucket of each group
//   (CSR per image, ascending ids inside a bucket), in group order; a candidate met again in a later group is skipped
//   (its first appearance counts: it is a repeat iff it shares the query's bucket in an earlier group); queries with at most
//   two raw candidates are dropped. The ten candidates of smallest Hamming distance between the 128-bit codes, ties in
//   order of appearance = the ten smallest keys (distance << 24 | appearance index), kept in a register insertion network.
//   Their exact L2<uint8> distances (v_dot4_u32_u8), the two smallest (distance, id) pairs in lexicographic order - a total
//   order, unlike the brute-force path no tie can be ambiguous - and the fp32 ratio test of the reference.
// Latency / L2-traffic bound gather work (~12 candidates per query at 2 000 descriptors and 1 024 buckets per group).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kCasGroupsMax = 8;
constexpr int kCasTop = 10;
struct CasParams {
  const uint32_t* words;        // descriptors, DW dwords per row (uint8: length / 4 bytes packed; float: one element per dword)
  const uint32_t* hash;         // hash code per row: HW dwords in a slot of 2, 4 or 8 (zero padded)
  const uint4* bids;            // bucket ids per row: 8 x uint16 (groups beyond n_groups unused)
  const uint32_t* bstart;       // per image: n_groups * n_buckets + 1 offsets into its items
  const uint32_t* items;        // per image: n_groups lists of its rows, grouped by bucket (ascending row inside a bucket)
  const uint64_t* img_row_off;
  const uint32_t* img_n;
  const uint2* pairs;
  const uint2* work;
  uint32_t* best;
  uint32_t* count;
  uint32_t qstride, n_groups, n_buckets;
  float ratio_sq;
};
__device__ __forceinline__ uint32_t cas_bid(const uint4& b, int g) {
  const uint32_t w = g < 2 ? b.x : g < 4 ? b.y : g < 6 ? b.z : b.w;
  return (g & 1) ? (w >> 16) : (w & 0xFFFFu);
}
// DW: dwords per descriptor row, HW: dwords of the hash code (= descriptor length / 32, rounded up), kFloat: L2<float> on the ten
// candidates in the reference's summation order (metric.hpp: groups of four, separate multiplies and adds) instead of L2<uint8_t>.
// Instantiations: <32, 4, false> SIFT_Regions, <36, 5, false> AKAZE_Liop_Regions (144 bytes), <64, 2, true> AKAZE_Float_Regions.
template <int DW, int HW, bool kFloat>
__global__ __launch_bounds__(kQBlock) void cascade_match_kernel(CasParams p) {
#pragma clang fp contract(off)
  constexpr int HS = HW <= 2 ? 2 : HW <= 4 ? 4 : 8;   // dwords between the codes of two rows
  __shared__ uint32_t sh_count;
  const uint2 w = p.work[blockIdx.x];
  const uint2 ij = p.pairs[w.x];
  const uint32_t nJ = p.img_n[ij.y];
  const uint32_t q = w.y + threadIdx.x;
  const bool active = q < nJ;
  if (threadIdx.x == 0) sh_count = 0;
  __syncthreads();
  uint32_t out = kInvalid;
  if (active) {
    const uint64_t offI = p.img_row_off[ij.x], rowJ = p.img_row_off[ij.y] + q;
    const uint4 bq = p.bids[rowJ];
    uint32_t hq[HW];
#pragma unroll
    for (int k = 0; k < HW; ++k) hq[k] = p.hash[rowJ * HS + k];
    const int G = (int)p.n_groups;
    const uint32_t* __restrict__ bs = p.bstart + (size_t)ij.x * ((size_t)G * p.n_buckets + 1);
    const uint32_t* __restrict__ it = p.items + offI * (uint64_t)G;
    uint32_t s[kCasGroupsMax], len[kCasGroupsMax], raw = 0;
#pragma unroll
    for (int g = 0; g < kCasGroupsMax; ++g) {
      s[g] = 0; len[g] = 0;
      if (g < G) {
        const uint32_t b = (uint32_t)g * p.n_buckets + cas_bid(bq, g);
        s[g] = bs[b];
        len[g] = bs[b + 1] - s[g];
        raw += len[g];
      }
    }
    if (raw > 2) {
      uint32_t top[kCasTop];
#pragma unroll
      for (int k = 0; k < kCasTop; ++k) top[k] = 0xFFFFFFFFu;
      uint32_t t = 0;
#pragma unroll
      for (int g = 0; g < kCasGroupsMax; ++g) {
        for (uint32_t e = 0; e < len[g]; ++e, ++t) {
          const uint32_t id = it[s[g] + e];
          const uint4 bi = p.bids[offI + id];
          bool repeat = false;
#pragma unroll
          for (int g2 = 0; g2 < kCasGroupsMax; ++g2)
            if (g2 < g) repeat = repeat || cas_bid(bi, g2) == cas_bid(bq, g2);
          if (repeat) continue;
          uint32_t ham = 0;
#pragma unroll
          for (int k = 0; k < HW; ++k) ham += __popc(p.hash[(offI + id) * HS + k] ^ hq[k]);
          uint32_t key = (ham << 24) | t;
#pragma unroll
          for (int k = 0; k < kCasTop; ++k) {   // sorted insertion: the list keeps the ten smallest keys
            const uint32_t lo = umin32(key, top[k]), hi2 = umax32(key, top[k]);
            top[k] = lo; key = hi2;
          }
        }
      }
      // exact distances of the (up to) ten selected candidates; the two smallest (distance, id) pairs
      uint32_t qv[DW];
      {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(p.words + rowJ * DW);
#pragma unroll
        for (int k = 0; k < DW / 4; ++k) { const uint4 v = src[k]; qv[4 * k] = v.x; qv[4 * k + 1] = v.y; qv[4 * k + 2] = v.z; qv[4 * k + 3] = v.w; }
      }
      uint32_t qn = 0;
      if (!kFloat) {
#pragma unroll
        for (int k = 0; k < DW; ++k) qn = __builtin_amdgcn_udot4(qv[k], qv[k], qn, false);
      }
      uint64_t b0 = ~0ull, b1 = ~0ull;   // (distance << 32 | id): lexicographic (distance, id); non-negative floats order like their bits
      int n_top = 0;
#pragma unroll
      for (int k = 0; k < kCasTop; ++k) {
        if (top[k] == 0xFFFFFFFFu) continue;
        ++n_top;
        const uint32_t tt = top[k] & 0xFFFFFFu;
        uint32_t cum = 0, idx = 0;
#pragma unroll
        for (int g = 0; g < kCasGroupsMax; ++g) {
          if (tt >= cum && tt < cum + len[g]) idx = s[g] + (tt - cum);
          cum += len[g];
        }
        const uint32_t id = it[idx];
        const uint4* __restrict__ row = reinterpret_cast<const uint4*>(p.words + (offI + id) * DW);
        uint32_t dbits;
        if (kFloat) {
          // L2<float>::operator()(candidate, query, size) (metric.hpp:98-135): result += ((d0 d0 + d1 d1) + d2 d2) + d3 d3 per four elements
          float result = 0.f;
#pragma unroll
          for (int c4 = 0; c4 < DW / 4; ++c4) {
            const uint4 v = row[c4];
            const float d0 = __uint_as_float(v.x) - __uint_as_float(qv[4 * c4]), d1 = __uint_as_float(v.y) - __uint_as_float(qv[4 * c4 + 1]);
            const float d2 = __uint_as_float(v.z) - __uint_as_float(qv[4 * c4 + 2]), d3 = __uint_as_float(v.w) - __uint_as_float(qv[4 * c4 + 3]);
            const float g = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
            result = result + g;
          }
          dbits = __float_as_uint(result);
        } else {
          uint32_t dot = 0, rn = 0;
#pragma unroll
          for (int c4 = 0; c4 < DW / 4; ++c4) {
            const uint4 v = row[c4];
            dot = __builtin_amdgcn_udot4(v.x, qv[4 * c4], dot, false); rn = __builtin_amdgcn_udot4(v.x, v.x, rn, false);
            dot = __builtin_amdgcn_udot4(v.y, qv[4 * c4 + 1], dot, false); rn = __builtin_amdgcn_udot4(v.y, v.y, rn, false);
            dot = __builtin_amdgcn_udot4(v.z, qv[4 * c4 + 2], dot, false); rn = __builtin_amdgcn_udot4(v.z, v.z, rn, false);
            dot = __builtin_amdgcn_udot4(v.w, qv[4 * c4 + 3], dot, false); rn = __builtin_amdgcn_udot4(v.w, v.w, rn, false);
          }
          dbits = rn + qn - 2u * dot;   // exact (< 2^24)
        }
        top2_u64(((uint64_t)dbits << 32) | id, b0, b1);
      }
      if (n_top >= 2) {
        const float d0 = kFloat ? __uint_as_float((uint32_t)(b0 >> 32)) : (float)(uint32_t)(b0 >> 32);
        const float d1 = kFloat ? __uint_as_float((uint32_t)(b1 >> 32)) : (float)(uint32_t)(b1 >> 32);
        if (d0 < __fmul_rn(p.ratio_sq, d1)) out = (uint32_t)b0;
      }
    }
    p.best[(size_t)w.x * p.qstride + q] = out;
  }
  if (out != kInvalid) atomicAdd(&sh_count, 1u);
  __syncthreads();
  if (threadIdx.x == 0 && sh_count) atomicAdd(&p.count[w.x], sh_count);
}

// exclusive scan of the per-pair counts (one workgroup)
__global__ __launch_bounds__(1024) void hamming_scan_kernel(const uint32_t* __restrict__ count, uint32_t n, uint32_t* __restrict__ offsets) {
  __shared__ uint32_t part[1024];
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t lo = umin32(threadIdx.x * per, n), hi = umin32(lo + per, n);
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; ++k) s += count[k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t v = threadIdx.x >= (uint32_t)off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;
  for (uint32_t k = lo; k < hi; ++k) { offsets[k] = run; run += count[k]; }
  if (threadIdx.x == 1023) offsets[n] = part[1023];
}

// ordered gather: matches of a pair in ascending query index (regions_matcher.hpp:198-205 emits them in that order)
__global__ __launch_bounds__(256) void hamming_compact_kernel(const uint32_t* __restrict__ best, const uint32_t* __restrict__ offsets,
                                                              const uint2* __restrict__ pairs, const uint32_t* __restrict__ img_n,
                                                              uint32_t qstride, uint2* __restrict__ out) {
  __shared__ uint32_t sh[256];
  const uint32_t k = blockIdx.x;
  const uint32_t nJ = img_n[pairs[k].y];
  uint32_t base = offsets[k];
  if (offsets[k + 1] == base) return;
  for (uint32_t q0 = 0; q0 < nJ; q0 += 256) {
    const uint32_t q = q0 + threadIdx.x;
    const uint32_t b = q < nJ ? best[(size_t)k * qstride + q] : kInvalid;
    const uint32_t f = b != kInvalid ? 1u : 0u;
    sh[threadIdx.x] = f;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const uint32_t v = threadIdx.x >= (uint32_t)off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += v;
      __syncthreads();
    }
    if (f) out[base + sh[threadIdx.x] - 1] = make_uint2(b, q);
    base += sh[255];
    __syncthreads();
  }
}

template <typename T>
struct Buf {
  T* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  int ensure(size_t n) {
    if (n <= cap) return MVGX_OK;
    release();
    const size_t want = std::max<size_t>(n + n / 4, 16);
    if (pinned) MVGX_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), want * sizeof(T), 0));
    else MVGX_HIP(mvgx::device_malloc(reinterpret_cast<void**>(&p), want * sizeof(T)));
    cap = want;
    return MVGX_OK;
  }
  void release() {
    if (p) { if (pinned) (void)hipHostFree(p); else (void)hipFree(p); }
    p = nullptr; cap = 0;
  }
};

}  // namespace

namespace {
// HASHING stage of CASCADE_HASHING_L2 (matching/cascade_hasher.hpp:179-223, CreateHashedDescriptions): per descriptor
//   descriptor = row.cast<float>() - zero_mean;  primary_projection = P1 * descriptor  (128 x 128);  hash bit j = primary_projection(j) > 0
//   per bucket group g: secondary_projection = P2[g] * descriptor (bits x 128); bucket id = the bits (projection > 0), first row most significant
// The products are Eigen 3.4 column-major matrix x vector products in single precision (GeneralMatrixVector.h,
// general_matrix_vector_product<..., ColMajor, ...>::run): for 128 columns the kernel walks the columns in blocks of 16
// (block_cols = cols < 128 ? cols : (stride * 4 < 32000 ? 16 : 4)); inside a block every output row accumulates
// c = a(i, j) * x(j) + c from c = 0 in ascending j - a rounded product and a rounded sum, the reference objects are built without
// FMA - and the block's c is then added to the row's result (res = c * 1 + res). One lane per descriptor reproduces exactly that
// order with a rounded product and a rounded sum (cas_mul_rn / cas_add_rn below: no contraction); the projection rows are wave-uniform and come through the scalar cache.
// P: [128 + groups * bits][128] floats, row q = row q of the primary projection, then the rows of the secondary projections.
// (OCML's rounded operations: the toolchain's __fmul_rn / __fadd_rn are plain x * y / x + y unless OCML_BASIC_ROUNDED_OPERATIONS is defined on the
// command line, and hipcc's default contraction turned this loop into v_pk_fma_f32 - found in round 4, mvgx_geofilter.hip has the story.)
#ifdef __HIPCC__
__device__ __forceinline__ float cas_mul_rn(float a, float b) { return __ocml_mul_rte_f32(a, b); }
__device__ __forceinline__ float cas_add_rn(float a, float b) { return __ocml_add_rte_f32(a, b); }
#else   // the HIP emulation of the test-suite
__device__ __forceinline__ float cas_mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float cas_add_rn(float a, float b) { return __fadd_rn(a, b); }
#endif
constexpr int kCasDim = 128, kCasBlockCols = 16;
__global__ __launch_bounds__(256) void cascade_hash_kernel(const uint32_t* __restrict__ words, uint64_t n_rows, const float* __restrict__ zero_mean,
                                                           const float* __restrict__ P, int n_groups, int bits, uint4* __restrict__ hash,
                                                           uint4* __restrict__ bids) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = r < n_rows;
  float d[kCasDim];
  {
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(words + (live ? r : 0) * (kCasDim / 4));
#pragma unroll
    for (int q = 0; q < kCasDim / 16; ++q) {
      const uint4 w = src[q];
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int e = 0; e < 16; ++e) d[16 * q + e] = __fsub_rn((float)((ww[e >> 2] >> (8 * (e & 3))) & 255u), zero_mean[16 * q + e]);
    }
  }
  auto project = [&](const float* __restrict__ row) -> float {   // one output row: eight column blocks of sixteen
    float res = 0.0f;
#pragma unroll
    for (int b = 0; b < kCasDim / kCasBlockCols; ++b) {
      float c = 0.0f;
#pragma unroll
      for (int j = 0; j < kCasBlockCols; ++j) c = cas_add_rn(cas_mul_rn(row[kCasBlockCols * b + j], d[kCasBlockCols * b + j]), c);
      res = cas_add_rn(c, res);
    }
    return res;
  };
  uint32_t code[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int w = 0; w < 4; ++w)
    for (int b = 0; b < 32; ++b) {
      const float v = project(P + (size_t)(32 * w + b) * kCasDim);
      if (v > 0.0f) code[w] |= 1u << b;   // dynamic_bitset: bit q of the code = bit (q % 8) of byte q / 8
    }
  uint32_t ids[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g >= n_groups) break;   // uniform
    uint32_t id = 0;
    for (int k = 0; k < bits; ++k) {
      const float v = project(P + (size_t)(kCasDim + g * bits + k) * kCasDim);
      id = (id << 1) + (v > 0.0f ? 1u : 0u);
    }
    ids[g] = id & 0xFFFFu;   // uint16_t bucket_id
  }
  if (!live) return;
  hash[r] = make_uint4(code[0], code[1], code[2], code[3]);
  bids[r] = make_uint4(ids[0] | (ids[1] << 16), ids[2] | (ids[3] << 16), ids[4] | (ids[5] << 16), ids[6] | (ids[7] << 16));
}
// The same stage for the other shapes openMVG's scalar describers produce (round 5): 144-byte uint8 rows (AKAZE_Liop_Regions: nine column
// blocks of 16) and 64-float rows (AKAZE_Float_Regions: fewer than 128 columns are ONE block in Eigen's kernel, block_cols = cols). One code
// bit per dimension (CascadeHasher::Init(dimension)): DIM / 32 code dwords (rounded up) in a slot of HS dwords, the slot's tail zero.
template <int DIM, int BLOCK, bool FLOAT>
__global__ __launch_bounds__(256) void cascade_hash_typed_kernel(const uint32_t* __restrict__ words, uint64_t n_rows, const float* __restrict__ zero_mean,
                                                                 const float* __restrict__ P, int n_groups, int bits, uint32_t* __restrict__ hash,
                                                                 uint4* __restrict__ bids) {
  constexpr int HW = (DIM + 31) / 32, HS = HW <= 2 ? 2 : HW <= 4 ? 4 : 8;
  static_assert(DIM % BLOCK == 0 && DIM % 16 == 0, "column blocks");
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = r < n_rows;
  float d[DIM];
  if constexpr (FLOAT) {
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(words + (live ? r : 0) * DIM);
#pragma unroll
    for (int q = 0; q < DIM / 4; ++q) {
      const uint4 w = src[q];   // (four floats)
      d[4 * q] = __fsub_rn(__uint_as_float(w.x), zero_mean[4 * q]); d[4 * q + 1] = __fsub_rn(__uint_as_float(w.y), zero_mean[4 * q + 1]);
      d[4 * q + 2] = __fsub_rn(__uint_as_float(w.z), zero_mean[4 * q + 2]); d[4 * q + 3] = __fsub_rn(__uint_as_float(w.w), zero_mean[4 * q + 3]);
    }
  } else {
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(words + (live ? r : 0) * (DIM / 4));
#pragma unroll
    for (int q = 0; q < DIM / 16; ++q) {
      const uint4 w = src[q];
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int e = 0; e < 16; ++e) d[16 * q + e] = __fsub_rn((float)((ww[e >> 2] >> (8 * (e & 3))) & 255u), zero_mean[16 * q + e]);
    }
  }
  auto project = [&](const float* __restrict__ row) -> float {
    float res = 0.0f;
#pragma unroll
    for (int b = 0; b < DIM / BLOCK; ++b) {
      float c = 0.0f;
#pragma unroll
      for (int j = 0; j < BLOCK; ++j) c = cas_add_rn(cas_mul_rn(row[BLOCK * b + j], d[BLOCK * b + j]), c);
      res = cas_add_rn(c, res);
    }
    return res;
  };
  uint32_t code[HS];
#pragma unroll
  for (int w = 0; w < HS; ++w) code[w] = 0u;
#pragma unroll
  for (int w = 0; w < HW; ++w)
    for (int b = 0; b < 32; ++b) {
      if (32 * w + b >= DIM) break;
      const float v = project(P + (size_t)(32 * w + b) * DIM);
      if (v > 0.0f) code[w] |= 1u << b;
    }
  uint32_t ids[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g >= n_groups) break;   // uniform
    uint32_t id = 0;
    for (int k = 0; k < bits; ++k) {
      const float v = project(P + (size_t)(DIM + g * bits + k) * DIM);
      id = (id << 1) + (v > 0.0f ? 1u : 0u);
    }
    ids[g] = id & 0xFFFFu;
  }
  if (!live) return;
#pragma unroll
  for (int w = 0; w < HS; ++w) hash[r * HS + w] = code[w];
  bids[r] = make_uint4(ids[0] | (ids[1] << 16), ids[2] | (ids[3] << 16), ids[4] | (ids[5] << 16), ids[6] | (ids[7] << 16));
}
}  // namespace

struct BfCtx {
  int kind = 0;   // 0: Hamming on packed bits (nw dwords per row), 1: L2<float> (nw floats per row), 2: L2<uint8> (nw dwords per row)
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evk0 = nullptr, evk1 = nullptr;
  int64_t batch_pairs = 1 << 15;
  uint32_t n_images = 0, nw = 0, desc_bytes = 0, max_n = 0, qstride = 0;
  std::vector<uint32_t> h_n;
  Buf<uint32_t> d_words, d_n, d_best, d_count, d_offsets, d_norms;
  // kind 3 (cascade hashing): hash codes, bucket ids, per-image bucket lists
  Buf<uint4> d_hash, d_bids;
  Buf<uint32_t> d_bstart, d_items;
  uint32_t cas_groups = 0, cas_buckets = 0;
  bool cas_float = false;             // cascade hashing on float rows (AKAZE_Float_Regions): L2<float> on the candidates
  Buf<uint64_t> d_row_off;
  Buf<uint2> d_pairs, d_work, d_ij;
  Buf<uint2> hp_pairs, hp_work;
  Buf<uint32_t> hp_offsets;
  std::vector<uint64_t> res_offsets;
  std::vector<uint32_t> res_ij;
  BfCtx() { hp_pairs.pinned = hp_work.pinned = hp_offsets.pinned = true; }
};
struct mvgx_hamming_ctx : BfCtx {};
struct mvgx_l2f_ctx : BfCtx {};
struct mvgx_l2u8_ctx : BfCtx {};
struct mvgx_cascade_ctx : BfCtx {};

namespace {

int bf_create(int kind, int device, BfCtx* c) {
  c->kind = kind;
  const int rc = mvgx::select_device(device);
  if (rc) return rc;
  MVGX_HIP(hipGetDevice(&c->device));
  MVGX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  MVGX_HIP(hipEventCreate(&c->ev0)); MVGX_HIP(hipEventCreate(&c->ev1));
  MVGX_HIP(hipEventCreate(&c->evk0)); MVGX_HIP(hipEventCreate(&c->evk1));
  return MVGX_OK;
}

void bf_release(BfCtx* c) {
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->d_hash.release(); c->d_bids.release(); c->d_bstart.release(); c->d_items.release();
  c->d_words.release(); c->d_norms.release(); c->d_n.release(); c->d_best.release(); c->d_count.release(); c->d_offsets.release();
  c->d_row_off.release(); c->d_pairs.release(); c->d_work.release(); c->d_ij.release();
  c->hp_pairs.release(); c->hp_work.release(); c->hp_offsets.release();
  for (hipEvent_t e : {c->ev0, c->ev1, c->evk0, c->evk1}) if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
}

int bf_set_option(BfCtx* c, const char* key, int64_t value) {
  MVGX_REQUIRE(c && key, MVGX_ERR_ARG, "set_option: NULL argument");
  if (!strcmp(key, "batch_pairs")) {
    MVGX_REQUIRE(value >= 1 && value <= (1 << 20), MVGX_ERR_ARG, "batch_pairs must be in [1, 2^20]");
    c->batch_pairs = value;
    return MVGX_OK;
  }
  set_error("set_option: unknown key '%s'", key);
  return MVGX_ERR_ARG;
}

// rows of row_bytes bytes each -> device rows of nw dwords (zero padded: padding is equal in every row and adds nothing)
int bf_set_regions(BfCtx* c, const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images, uint32_t row_bytes,
                   uint32_t nw) {
  MVGX_HIP(hipSetDevice(c->device));
  c->n_images = n_images;
  c->desc_bytes = row_bytes;
  c->nw = nw;
  c->h_n.assign(n_desc, n_desc + n_images);
  std::vector<uint64_t> off(n_images + 1, 0);
  c->max_n = 0;
  for (uint32_t k = 0; k < n_images; ++k) {
    MVGX_REQUIRE(n_desc[k] == 0 || desc_rows[k] != nullptr, MVGX_ERR_ARG, "image %u: NULL descriptor array", k);
    MVGX_REQUIRE(n_desc[k] < (1u << kRowBits), MVGX_ERR_UNSUPPORTED, "image %u: %u descriptors (limit 2^22 - 1)", k, n_desc[k]);
    off[k + 1] = off[k] + (c->kind == 1 ? (n_desc[k] + 1u) / 2u * 2u : n_desc[k]);   // float path: whole row pairs
    c->max_n = std::max(c->max_n, n_desc[k]);
  }
  c->qstride = (c->max_n + kQBlock - 1) / kQBlock * kQBlock;
  const uint64_t rows = off[n_images];
  std::vector<uint32_t> words((size_t)std::max<uint64_t>(rows, 1) * nw, 0u);
  for (uint32_t k = 0; k < n_images; ++k)
    for (uint32_t r = 0; r < n_desc[k]; ++r) {
      if (c->kind == 1) {   // element e of row r -> pair (r / 2), slot 2 e + (r & 1)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(desc_rows[k] + (size_t)r * row_bytes);
        uint32_t* dst = words.data() + (off[k] + (r & ~1u)) * nw + (r & 1u);
        for (uint32_t e = 0; e < nw; ++e) dst[2 * e] = src[e];
      } else {
        memcpy(reinterpret_cast<uint8_t*>(words.data() + (off[k] + r) * nw), desc_rows[k] + (size_t)r * row_bytes, row_bytes);
      }
    }
  int rc;
  if ((rc = c->d_words.ensure(words.size())) || (rc = c->d_row_off.ensure(n_images + 1)) || (rc = c->d_n.ensure(std::max(n_images, 1u))))
    return rc;
  MVGX_HIP(hipMemcpyAsync(c->d_words.p, words.data(), words.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipMemcpyAsync(c->d_row_off.p, off.data(), (n_images + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  std::vector<uint32_t> norms;
  if (c->kind == 2) {   // |a|^2 per row (metric.hpp:55-93 accumulates int; < 2^24 here)
    norms.assign((size_t)std::max<uint64_t>(rows, 1), 0u);
    for (uint64_t r = 0; r < rows; ++r) {
      const uint8_t* b = reinterpret_cast<const uint8_t*>(words.data() + r * nw);
      uint32_t nn = 0;
      for (uint32_t e = 0; e < row_bytes; ++e) nn += (uint32_t)b[e] * b[e];
      norms[r] = nn;
    }
    if ((rc = c->d_norms.ensure(norms.size()))) return rc;
    MVGX_HIP(hipMemcpyAsync(c->d_norms.p, norms.data(), norms.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  }
  if (n_images)
    MVGX_HIP(hipMemcpyAsync(c->d_n.p, c->h_n.data(), n_images * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipStreamSynchronize(c->stream));
  return MVGX_OK;
}

// the I/J loops of Matcher_Regions::Match (Matcher_Regions.cpp:57-105) in batches: work list -> top-2 + ratio kernel ->
// per-pair counts -> exclusive scan -> ordered compaction -> host lists
int bf_run(BfCtx* c, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio, mvgx_match_stats* stats) {
  MVGX_REQUIRE(c && (pairs_IJ || n_pairs == 0), MVGX_ERR_ARG, "run: NULL argument");
  MVGX_REQUIRE(c->nw != 0 || c->n_images == 0, MVGX_ERR_STATE, "run before set_regions");
  // (cascade hashing orders its ten candidates by (distance, id): no ambiguous tie, any ratio is reproduced)
  MVGX_REQUIRE(c->kind == 3 || (ratio <= 1.0f && ratio >= 0.0f), MVGX_ERR_UNSUPPORTED,
               "ratio = %g: the device path reproduces the reference only for 0 <= ratio <= 1 "
               "(ties are libstdc++ partial_sort order beyond that)", (double)ratio);
  MVGX_HIP(hipSetDevice(c->device));
  for (uint64_t k = 0; k < n_pairs; ++k)
    MVGX_REQUIRE(pairs_IJ[2 * k] < c->n_images && pairs_IJ[2 * k + 1] < c->n_images, MVGX_ERR_ARG,
                 "pair %llu references image out of range", (unsigned long long)k);
  c->res_offsets.assign(n_pairs + 1, 0);
  c->res_ij.clear();
  mvgx_match_stats st;
  memset(&st, 0, sizeof(st));
  st.variant = (c->kind == 0 ? 100 : c->kind == 1 ? 200 : c->kind == 2 ? 300 : 400) + c->nw;
  int rc;
  float kernel_ms = 0.f;
  MVGX_HIP(hipEventRecord(c->ev0, c->stream));
  // pairs per batch: the option, capped so that the scratch (4 B per pair and query slot) stays near 2 GB when the images carry
  // tens of thousands of descriptors (the same rule as mvgx_match_run); match totals of a batch are 32-bit on the device
  const uint64_t B = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)c->batch_pairs, std::max<uint64_t>(16, (1ull << 29) / std::max<uint32_t>(c->qstride, 1))));
  MVGX_REQUIRE(B * std::max<uint32_t>(c->qstride, 1) < (1ull << 32), MVGX_ERR_UNSUPPORTED,
               "images of %u descriptor slots: a 16-pair batch overflows the 32-bit match offsets of the device path", c->qstride);
  for (uint64_t p0 = 0; p0 < n_pairs; p0 += B) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(B, n_pairs - p0);
    const uint32_t blocks_per_pair = std::max<uint32_t>(1, c->qstride / kQBlock);
    if ((rc = c->hp_pairs.ensure(nb)) || (rc = c->hp_work.ensure((size_t)nb * blocks_per_pair))) return rc;
    uint32_t n_work = 0;
    for (uint32_t k = 0; k < nb; ++k) {
      const uint32_t I = pairs_IJ[2 * (p0 + k)], J = pairs_IJ[2 * (p0 + k) + 1];
      c->hp_pairs.p[k] = make_uint2(I, J);
      const uint32_t nI = c->h_n[I], nJ = c->h_n[J];
      // matcher_brute_force.hpp:108-113, Matcher_Regions.cpp:65-69,85-90; cascade: an empty I is skipped (Cascade_Hashing_Matcher_Regions.cpp:153-157)
      if (c->kind == 3 ? (nI == 0 || nJ == 0) : (nI < 2 || nJ == 0)) continue;
      for (uint32_t q0 = 0; q0 < nJ; q0 += kQBlock) c->hp_work.p[n_work++] = make_uint2(k, q0);
      st.n_pairs += 1;
      st.n_desc_pairs += (uint64_t)nI * nJ;
    }
    if ((rc = c->d_pairs.ensure(nb)) || (rc = c->d_work.ensure(std::max<uint32_t>(n_work, 1))) ||
        (rc = c->d_best.ensure((size_t)nb * std::max<uint32_t>(c->qstride, 1))) || (rc = c->d_count.ensure(nb)) ||
        (rc = c->d_offsets.ensure((size_t)nb + 1)) || (rc = c->hp_offsets.ensure((size_t)nb + 1)))
      return rc;
    MVGX_HIP(hipMemcpyAsync(c->d_pairs.p, c->hp_pairs.p, nb * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
    if (n_work) MVGX_HIP(hipMemcpyAsync(c->d_work.p, c->hp_work.p, n_work * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
    MVGX_HIP(hipMemsetAsync(c->d_count.p, 0, nb * sizeof(uint32_t), c->stream));
    if (n_work) {
      MVGX_HIP(hipEventRecord(c->evk0, c->stream));
      if (c->kind == 3) {
        CasParams cp;
        cp.words = c->d_words.p; cp.hash = reinterpret_cast<const uint32_t*>(c->d_hash.p); cp.bids = c->d_bids.p; cp.bstart = c->d_bstart.p; cp.items = c->d_items.p;
        cp.img_row_off = c->d_row_off.p; cp.img_n = c->d_n.p; cp.pairs = c->d_pairs.p; cp.work = c->d_work.p; cp.best = c->d_best.p;
        cp.count = c->d_count.p; cp.qstride = c->qstride; cp.n_groups = c->cas_groups; cp.n_buckets = c->cas_buckets; cp.ratio_sq = ratio;
        if (c->cas_float) hipLaunchKernelGGL((cascade_match_kernel<64, 2, true>), dim3(n_work), dim3(kQBlock), 0, c->stream, cp);
        else if (c->nw == 36) hipLaunchKernelGGL((cascade_match_kernel<36, 5, false>), dim3(n_work), dim3(kQBlock), 0, c->stream, cp);
        else hipLaunchKernelGGL((cascade_match_kernel<32, 4, false>), dim3(n_work), dim3(kQBlock), 0, c->stream, cp);
      } else if (c->kind == 0) {
        HamParams hp;
        hp.words = c->d_words.p; hp.img_row_off = c->d_row_off.p; hp.img_n = c->d_n.p; hp.pairs = c->d_pairs.p; hp.work = c->d_work.p;
        hp.best = c->d_best.p; hp.count = c->d_count.p; hp.qstride = c->qstride; hp.ratio = ratio;
        if (c->nw == 8) hipLaunchKernelGGL(hamming_top2_ratio_kernel<8>, dim3(n_work), dim3(kQBlock), 0, c->stream, hp);
        else hipLaunchKernelGGL(hamming_top2_ratio_kernel<16>, dim3(n_work), dim3(kQBlock), 0, c->stream, hp);
      } else if (c->kind == 2) {
        L2uParams up;
        up.words = c->d_words.p; up.norms = c->d_norms.p; up.img_row_off = c->d_row_off.p; up.img_n = c->d_n.p; up.pairs = c->d_pairs.p;
        up.work = c->d_work.p; up.best = c->d_best.p; up.count = c->d_count.p; up.qstride = c->qstride; up.ratio_sq = ratio;
        if (c->nw == 16) hipLaunchKernelGGL(l2u8_top2_ratio_kernel<16>, dim3(n_work), dim3(kQBlock), 0, c->stream, up);
        else if (c->nw == 32) hipLaunchKernelGGL(l2u8_top2_ratio_kernel<32>, dim3(n_work), dim3(kQBlock), 0, c->stream, up);
        else hipLaunchKernelGGL(l2u8_top2_ratio_kernel<36>, dim3(n_work), dim3(kQBlock), 0, c->stream, up);
      } else {
        L2fParams fp;
        fp.rows = reinterpret_cast<const float*>(c->d_words.p); fp.img_row_off = c->d_row_off.p; fp.img_n = c->d_n.p;
        fp.pairs = c->d_pairs.p; fp.work = c->d_work.p; fp.best = c->d_best.p; fp.count = c->d_count.p; fp.qstride = c->qstride;
        fp.ratio_sq = ratio;
        hipLaunchKernelGGL(l2f_top2_ratio_kernel<64>, dim3(n_work), dim3(kQBlock), 0, c->stream, fp);
      }
      MVGX_HIP(hipGetLastError());
      MVGX_HIP(hipEventRecord(c->evk1, c->stream));
      st.n_kernel_launches += 1;
    }
    hipLaunchKernelGGL(hamming_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_count.p, nb, c->d_offsets.p);
    MVGX_HIP(hipGetLastError());
    MVGX_HIP(hipMemcpyAsync(c->hp_offsets.p, c->d_offsets.p, ((size_t)nb + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    MVGX_HIP(hipStreamSynchronize(c->stream));
    if (n_work) {
      float ms = 0.f;
      MVGX_HIP(hipEventElapsedTime(&ms, c->evk0, c->evk1));
      kernel_ms += ms;
    }
    const uint32_t total = c->hp_offsets.p[nb];
    const uint64_t base = c->res_offsets[p0];
    for (uint32_t k = 0; k <= nb; ++k) c->res_offsets[p0 + k] = base + c->hp_offsets.p[k];
    st.n_matches += total;
    if (total) {
      if ((rc = c->d_ij.ensure(total))) return rc;
      hipLaunchKernelGGL(hamming_compact_kernel, dim3(nb), dim3(256), 0, c->stream, c->d_best.p, c->d_offsets.p, c->d_pairs.p,
                         c->d_n.p, c->qstride, c->d_ij.p);
      MVGX_HIP(hipGetLastError());
      const size_t old = c->res_ij.size();
      c->res_ij.resize(old + (size_t)total * 2);
      MVGX_HIP(hipMemcpyAsync(c->res_ij.data() + old, c->d_ij.p, (size_t)total * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
      MVGX_HIP(hipStreamSynchronize(c->stream));
    }
  }
  MVGX_HIP(hipEventRecord(c->ev1, c->stream));
  MVGX_HIP(hipEventSynchronize(c->ev1));
  float ms = 0.f;
  MVGX_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  st.total_ms = ms;
  st.kernel_ms = kernel_ms;
  if (stats) *stats = st;
  return MVGX_OK;
}

template <typename Ctx>
int bf_create_as(int kind, int device, Ctx** out) {
  MVGX_REQUIRE(out, MVGX_ERR_ARG, "create: NULL out");
  *out = nullptr;
  Ctx* c = new Ctx();
  const int rc = bf_create(kind, device, c);
  if (rc) { bf_release(c); delete c; return rc; }
  *out = c;
  return MVGX_OK;
}

}  // namespace

extern "C" {

int mvgx_hamming_create(int device, mvgx_hamming_ctx** out) { return bf_create_as(0, device, out); }
int mvgx_hamming_destroy(mvgx_hamming_ctx* c) { if (c) { bf_release(c); delete c; } return MVGX_OK; }
int mvgx_hamming_set_option(mvgx_hamming_ctx* c, const char* key, int64_t value) { return bf_set_option(c, key, value); }

int mvgx_hamming_set_regions(mvgx_hamming_ctx* c, const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                             uint32_t desc_bytes) {
  MVGX_REQUIRE(c && (n_images == 0 || (desc_rows && n_desc)), MVGX_ERR_ARG, "mvgx_hamming_set_regions: NULL argument");
  MVGX_REQUIRE(desc_bytes >= 1 && desc_bytes <= 64, MVGX_ERR_UNSUPPORTED,
               "binary descriptor of %u bytes unsupported (device path: 1..64 bytes; AKAZE MLDB is 64)", desc_bytes);
  return bf_set_regions(c, desc_rows, n_desc, n_images, desc_bytes, desc_bytes <= 32 ? 8 : 16);
}

int mvgx_hamming_run(mvgx_hamming_ctx* c, const uint32_t* pairs_IJ, uint64_t n_pairs, float dist_ratio, mvgx_match_stats* stats) {
  return bf_run(c, pairs_IJ, n_pairs, dist_ratio, stats);
}

int mvgx_hamming_results(mvgx_hamming_ctx* c, const uint64_t** offsets, const uint32_t** ij) {
  MVGX_REQUIRE(c && offsets && ij, MVGX_ERR_ARG, "mvgx_hamming_results: NULL argument");
  *offsets = c->res_offsets.data();
  *ij = c->res_ij.data();
  return MVGX_OK;
}

int mvgx_l2f_create(int device, mvgx_l2f_ctx** out) { return bf_create_as(1, device, out); }
int mvgx_l2f_destroy(mvgx_l2f_ctx* c) { if (c) { bf_release(c); delete c; } return MVGX_OK; }
int mvgx_l2f_set_option(mvgx_l2f_ctx* c, const char* key, int64_t value) { return bf_set_option(c, key, value); }

int mvgx_l2f_set_regions(mvgx_l2f_ctx* c, const float* const* desc_rows, const uint32_t* n_desc, uint32_t n_images, uint32_t dim) {
  MVGX_REQUIRE(c && (n_images == 0 || (desc_rows && n_desc)), MVGX_ERR_ARG, "mvgx_l2f_set_regions: NULL argument");
  MVGX_REQUIRE(dim == 64, MVGX_ERR_UNSUPPORTED,
               "float descriptors of length %u unsupported (device path: 64 = AKAZE_Float_Regions)", dim);
  return bf_set_regions(c, reinterpret_cast<const uint8_t* const*>(desc_rows), n_desc, n_images, dim * 4, dim);
}

int mvgx_l2f_run(mvgx_l2f_ctx* c, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq, mvgx_match_stats* stats) {
  return bf_run(c, pairs_IJ, n_pairs, ratio_sq, stats);
}

int mvgx_l2f_results(mvgx_l2f_ctx* c, const uint64_t** offsets, const uint32_t** ij) {
  MVGX_REQUIRE(c && offsets && ij, MVGX_ERR_ARG, "mvgx_l2f_results: NULL argument");
  *offsets = c->res_offsets.data();
  *ij = c->res_ij.data();
  return MVGX_OK;
}

int mvgx_l2u8_create(int device, mvgx_l2u8_ctx** out) { return bf_create_as(2, device, out); }
int mvgx_l2u8_destroy(mvgx_l2u8_ctx* c) { if (c) { bf_release(c); delete c; } return MVGX_OK; }
int mvgx_l2u8_set_option(mvgx_l2u8_ctx* c, const char* key, int64_t value) { return bf_set_option(c, key, value); }

int mvgx_l2u8_set_regions(mvgx_l2u8_ctx* c, const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images, uint32_t dim) {
  MVGX_REQUIRE(c && (n_images == 0 || (desc_rows && n_desc)), MVGX_ERR_ARG, "mvgx_l2u8_set_regions: NULL argument");
  MVGX_REQUIRE(dim == 64 || dim == 128 || dim == 144, MVGX_ERR_UNSUPPORTED,
               "uint8 descriptors of length %u unsupported (this path: 64, 128 or 144 = AKAZE_Liop_Regions; SIFT's 128 has "
               "its own MFMA path, mvgx_match_*)", dim);
  return bf_set_regions(c, desc_rows, n_desc, n_images, dim, dim / 4);
}

int mvgx_l2u8_run(mvgx_l2u8_ctx* c, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq, mvgx_match_stats* stats) {
  return bf_run(c, pairs_IJ, n_pairs, ratio_sq, stats);
}

int mvgx_l2u8_results(mvgx_l2u8_ctx* c, const uint64_t** offsets, const uint32_t** ij) {
  MVGX_REQUIRE(c && offsets && ij, MVGX_ERR_ARG, "mvgx_l2u8_results: NULL argument");
  *offsets = c->res_offsets.data();
  *ij = c->res_ij.data();
  return MVGX_OK;
}

int mvgx_cascade_create(int device, mvgx_cascade_ctx** out) { return bf_create_as(3, device, out); }
int mvgx_cascade_destroy(mvgx_cascade_ctx* c) { if (c) { bf_release(c); delete c; } return MVGX_OK; }
int mvgx_cascade_set_option(mvgx_cascade_ctx* c, const char* key, int64_t value) { return bf_set_option(c, key, value); }

}  // extern "C"

namespace {
// the region types of the matching stage: scalar_type 0 = uint8 (128 = SIFT_Regions, 144 = AKAZE_Liop_Regions), 1 = float (64 =
// AKAZE_Float_Regions); the code has one bit per dimension (CascadeHasher::Init(dimension), stl::dynamic_bitset bytes)
int cas_check_shape(int scalar_type, uint32_t dim, uint32_t hash_bytes, uint32_t n_groups, uint32_t bits_per_bucket) {
  const bool known = (scalar_type == 0 && (dim == 128 || dim == 144)) || (scalar_type == 1 && dim == 64);
  MVGX_REQUIRE(known && hash_bytes == (dim + 7) / 8, MVGX_ERR_UNSUPPORTED,
               "cascade hashing on the device: uint8 descriptors of 128 or 144 bytes or float descriptors of length 64, with one code bit "
               "per dimension (CascadeHasher::Init(dimension)); got scalar type %d, length %u, %u code bytes", scalar_type, dim, hash_bytes);
  MVGX_REQUIRE(n_groups >= 1 && n_groups <= (uint32_t)kCasGroupsMax && bits_per_bucket >= 1 && bits_per_bucket <= 16, MVGX_ERR_UNSUPPORTED,
               "cascade hashing on the device: 1..8 bucket groups of 2^1..2^16 buckets (got %u groups, %u bits)", n_groups, bits_per_bucket);
  return MVGX_OK;
}

// per image and bucket group the lists of its descriptors by bucket, ascending descriptor inside a bucket (cascade_hasher.hpp:
// 225-237), from the bucket ids (8 x uint16 per descriptor); images are independent: host threads
int cas_build_buckets(BfCtx* c, const uint4* bids, const uint32_t* n_desc, uint32_t n_images, uint32_t G, uint32_t bits_per_bucket) {
  const uint32_t NB = 1u << bits_per_bucket;
  c->cas_groups = G; c->cas_buckets = NB;
  std::vector<uint64_t> off(n_images + 1, 0);
  for (uint32_t k = 0; k < n_images; ++k) {
    MVGX_REQUIRE(n_desc[k] < (1u << 24) / G, MVGX_ERR_UNSUPPORTED, "image %u: %u descriptors (limit 2^24 / groups: the appearance index has 24 bits)", k, n_desc[k]);
    off[k + 1] = off[k] + n_desc[k];
  }
  const uint64_t rows = off[n_images];
  std::vector<uint32_t> bstart((size_t)std::max<uint32_t>(n_images, 1) * ((size_t)G * NB + 1), 0u), items((size_t)std::max<uint64_t>(rows * G, 1), 0u);
  std::vector<int> bad(n_images, -1);
  auto image = [&](uint32_t k) {
    const uint32_t n = n_desc[k];
    uint32_t* bs = bstart.data() + (size_t)k * ((size_t)G * NB + 1);
    const uint16_t* b16 = reinterpret_cast<const uint16_t*>(bids + off[k]);
    for (uint32_t r = 0; r < n; ++r)
      for (uint32_t g = 0; g < G; ++g) {
        const uint32_t b = b16[(size_t)r * 8 + g];
        if (b >= NB) { bad[k] = (int)r; return; }
        bs[(size_t)g * NB + b + 1]++;
      }
    for (size_t b = 0; b < (size_t)G * NB; ++b) bs[b + 1] += bs[b];
    std::vector<uint32_t> fill(bs, bs + (size_t)G * NB);
    uint32_t* it = items.data() + off[k] * G;
    for (uint32_t g = 0; g < G; ++g)
      for (uint32_t r = 0; r < n; ++r) it[fill[(size_t)g * NB + b16[(size_t)r * 8 + g]]++] = r;   // ascending r inside a bucket
  };
  {
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)std::thread::hardware_concurrency(), (uint64_t)32, rows / 8192 + 1, (uint64_t)std::max(n_images, 1u)}));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back([&, t] { for (uint32_t k = t; k < n_images; k += T) image(k); });
    for (uint32_t k = 0; k < n_images; k += T) image(k);
    for (auto& x : th) x.join();
  }
  for (uint32_t k = 0; k < n_images; ++k)
    MVGX_REQUIRE(bad[k] < 0, MVGX_ERR_ARG, "image %u, descriptor %d: bucket id out of range", k, bad[k]);
  int rc;
  if ((rc = c->d_bstart.ensure(bstart.size())) || (rc = c->d_items.ensure(items.size()))) return rc;
  MVGX_HIP(hipMemcpyAsync(c->d_bstart.p, bstart.data(), bstart.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipMemcpyAsync(c->d_items.p, items.data(), items.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipStreamSynchronize(c->stream));
  return MVGX_OK;
}

// CascadeHasher::Init(dim, groups, bits, seed) (cascade_hasher.hpp:120-163): the projections are std::normal_distribution<>(0, 1)
// draws of a std::mt19937, converted to float, row by row - the primary projection first, then the secondary ones. The same
// standard-library facilities give the same values as the reference's own Init (libstdc++ on both sides of the boundary here).
const std::vector<float>& cas_projections(uint32_t dim, uint32_t groups, uint32_t bits, uint32_t seed) {
  struct Entry { uint32_t dim, groups, bits, seed; std::vector<float> P; };
  static std::mutex mu;
  static std::vector<Entry*> cache;   // never freed: a handful of parameter sets per process
  std::lock_guard<std::mutex> lk(mu);
  for (Entry* e : cache) if (e->dim == dim && e->groups == groups && e->bits == bits && e->seed == seed) return e->P;
  Entry* e = new Entry{dim, groups, bits, seed, {}};
  e->P.resize(((size_t)dim + (size_t)groups * bits) * dim);
  std::mt19937 gen(seed);
  std::normal_distribution<> nd(0, 1);
  for (float& v : e->P) v = (float)nd(gen);
  cache.push_back(e);
  return e->P;
}
}  // namespace

extern "C" {

int mvgx_cascade_set_regions_typed(mvgx_cascade_ctx* c, int scalar_type, const void* const* desc_rows, const uint8_t* const* hash_codes,
                                   const uint16_t* const* bucket_ids, const uint32_t* n_desc, uint32_t n_images, uint32_t dim,
                                   uint32_t hash_bytes, uint32_t n_groups, uint32_t bits_per_bucket) {
  MVGX_REQUIRE(c && (n_images == 0 || (desc_rows && hash_codes && bucket_ids && n_desc)), MVGX_ERR_ARG, "mvgx_cascade_set_regions: NULL argument");
  int rc = cas_check_shape(scalar_type, dim, hash_bytes, n_groups, bits_per_bucket);
  if (rc) return rc;
  const bool is_float = scalar_type == 1;
  if ((rc = bf_set_regions(c, reinterpret_cast<const uint8_t* const*>(desc_rows), n_desc, n_images, is_float ? dim * 4 : dim, is_float ? dim : dim / 4))) return rc;
  c->cas_float = is_float;
  const uint32_t G = n_groups;
  const uint32_t HW = (dim + 31) / 32, HS = HW <= 2 ? 2 : HW <= 4 ? 4 : 8;   // code dwords, slot dwords (cascade_match_kernel)
  std::vector<uint64_t> off(n_images + 1, 0);
  for (uint32_t k = 0; k < n_images; ++k) {
    MVGX_REQUIRE(n_desc[k] == 0 || (hash_codes[k] && bucket_ids[k]), MVGX_ERR_ARG, "image %u: NULL hash / bucket array", k);
    off[k + 1] = off[k] + n_desc[k];
  }
  const uint64_t rows = off[n_images];
  std::vector<uint32_t> hash((size_t)std::max<uint64_t>(rows, 1) * HS + 4, 0u);
  std::vector<uint4> bids((size_t)std::max<uint64_t>(rows, 1));
  for (uint32_t k = 0; k < n_images; ++k)
    for (uint32_t r = 0; r < n_desc[k]; ++r) {
      memcpy(&hash[(off[k] + r) * HS], hash_codes[k] + (size_t)r * hash_bytes, hash_bytes);   // (the slot's tail stays zero)
      uint16_t b8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (uint32_t g = 0; g < G; ++g) b8[g] = bucket_ids[k][(size_t)r * G + g];
      memcpy(&bids[off[k] + r], b8, 16);
    }
  if ((rc = c->d_hash.ensure(hash.size() / 4)) || (rc = c->d_bids.ensure(bids.size()))) return rc;
  MVGX_HIP(hipMemcpyAsync(c->d_hash.p, hash.data(), hash.size() / 4 * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipMemcpyAsync(c->d_bids.p, bids.data(), bids.size() * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
  return cas_build_buckets(c, bids.data(), n_desc, n_images, G, bits_per_bucket);
}
int mvgx_cascade_set_regions(mvgx_cascade_ctx* c, const uint8_t* const* desc_rows, const uint8_t* const* hash_codes,
                             const uint16_t* const* bucket_ids, const uint32_t* n_desc, uint32_t n_images, uint32_t dim,
                             uint32_t hash_bytes, uint32_t n_groups, uint32_t bits_per_bucket) {
  return mvgx_cascade_set_regions_typed(c, 0, reinterpret_cast<const void* const*>(desc_rows), hash_codes, bucket_ids, n_desc, n_images, dim, hash_bytes,
                                        n_groups, bits_per_bucket);
}

// Test hook (not declared in include/mvgx.h): out[0] = cas_add_rn(cas_mul_rn(a, b), c) as cascade_hash_kernel evaluates it, out[1] = fmaf(a, b, c)
__global__ void cas_rounded_ops_debug_kernel(const float* abc, float* out) {
  out[0] = cas_add_rn(cas_mul_rn(abc[0], abc[1]), abc[2]);
  out[1] = fmaf(abc[0], abc[1], abc[2]);
}
int mvgx_debug_rounded_ops_f32(const float* abc, float* out) {
  MVGX_REQUIRE(abc && out, MVGX_ERR_ARG, "mvgx_debug_rounded_ops_f32: NULL argument");
  int rc = mvgx::select_device(-1);
  if (rc) return rc;
  Buf<float> din, dout;
  struct Release { Buf<float>&a, &b; ~Release() { a.release(); b.release(); } } release{din, dout};
  if ((rc = din.ensure(3)) || (rc = dout.ensure(2))) return rc;
  MVGX_HIP(hipMemcpy(din.p, abc, 3 * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(cas_rounded_ops_debug_kernel, dim3(1), dim3(1), 0, nullptr, (const float*)din.p, dout.p);
  MVGX_HIP(hipGetLastError());
  MVGX_HIP(hipStreamSynchronize(nullptr));
  MVGX_HIP(hipMemcpy(out, dout.p, 2 * sizeof(float), hipMemcpyDeviceToHost));
  return MVGX_OK;
}

int mvgx_cascade_hash_regions_typed(mvgx_cascade_ctx* c, int scalar_type, const void* const* desc_rows, const uint32_t* n_desc, uint32_t n_images, uint32_t dim,
                                    const float* zero_mean, uint32_t n_groups, uint32_t bits_per_bucket, uint32_t random_seed,
                                    uint8_t* const* hash_codes_out, uint16_t* const* bucket_ids_out) {
  MVGX_REQUIRE(c && zero_mean && (n_images == 0 || (desc_rows && n_desc)), MVGX_ERR_ARG, "mvgx_cascade_hash_regions: NULL argument");
  const uint32_t hash_bytes = (dim + 7) / 8;
  int rc = cas_check_shape(scalar_type, dim, hash_bytes, n_groups, bits_per_bucket);
  if (rc) return rc;
  const bool is_float = scalar_type == 1;
  // the descriptors, row after row, on the device
  if ((rc = bf_set_regions(c, reinterpret_cast<const uint8_t* const*>(desc_rows), n_desc, n_images, is_float ? dim * 4 : dim, is_float ? dim : dim / 4))) return rc;
  c->cas_float = is_float;
  uint64_t rows = 0;
  for (uint32_t k = 0; k < n_images; ++k) rows += n_desc[k];
  const uint32_t HW = (dim + 31) / 32, HS = HW <= 2 ? 2 : HW <= 4 ? 4 : 8;   // code dwords, slot dwords (cascade_match_kernel)
  const std::vector<float>& P = cas_projections(dim, n_groups, bits_per_bucket, random_seed);
  Buf<float> d_P, d_zm;
  const size_t hash_u4 = ((size_t)std::max<uint64_t>(rows, 1) * HS + 4 + 3) / 4;
  if ((rc = d_P.ensure(P.size())) || (rc = d_zm.ensure(dim)) || (rc = c->d_hash.ensure(hash_u4)) || (rc = c->d_bids.ensure((size_t)std::max<uint64_t>(rows, 1))))
    return rc;
  struct Release { Buf<float>&a, &b; ~Release() { a.release(); b.release(); } } release{d_P, d_zm};
  MVGX_HIP(hipMemcpyAsync(d_P.p, P.data(), P.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipMemcpyAsync(d_zm.p, zero_mean, dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
  std::vector<uint4> bids((size_t)std::max<uint64_t>(rows, 1));
  std::vector<uint32_t> hash;
  if (rows) {
    const dim3 grid((unsigned)((rows + 255) / 256));
    uint32_t* const h32 = reinterpret_cast<uint32_t*>(c->d_hash.p);
    if (!is_float && dim == 128)
      hipLaunchKernelGGL(cascade_hash_kernel, grid, dim3(256), 0, c->stream, c->d_words.p, rows, d_zm.p, d_P.p, (int)n_groups, (int)bits_per_bucket, c->d_hash.p, c->d_bids.p);
    else if (!is_float)   // 144
      hipLaunchKernelGGL((cascade_hash_typed_kernel<144, 16, false>), grid, dim3(256), 0, c->stream, c->d_words.p, rows, d_zm.p, d_P.p, (int)n_groups,
                         (int)bits_per_bucket, h32, c->d_bids.p);
    else                  // 64 floats
      hipLaunchKernelGGL((cascade_hash_typed_kernel<64, 64, true>), grid, dim3(256), 0, c->stream, c->d_words.p, rows, d_zm.p, d_P.p, (int)n_groups,
                         (int)bits_per_bucket, h32, c->d_bids.p);
    MVGX_HIP(hipGetLastError());
    MVGX_HIP(hipMemcpyAsync(bids.data(), c->d_bids.p, rows * sizeof(uint4), hipMemcpyDeviceToHost, c->stream));
    if (hash_codes_out) {
      hash.resize(rows * HS);
      MVGX_HIP(hipMemcpyAsync(hash.data(), c->d_hash.p, rows * HS * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    }
  }
  MVGX_HIP(hipStreamSynchronize(c->stream));
  uint64_t at = 0;
  for (uint32_t k = 0; k < n_images; ++k) {   // the per-descriptor outputs in the reference's shapes, on request
    for (uint32_t r = 0; r < n_desc[k]; ++r) {
      if (hash_codes_out && hash_codes_out[k]) memcpy(hash_codes_out[k] + (size_t)r * hash_bytes, &hash[(at + r) * HS], hash_bytes);
      if (bucket_ids_out && bucket_ids_out[k]) memcpy(bucket_ids_out[k] + (size_t)r * n_groups, &bids[at + r], n_groups * sizeof(uint16_t));
    }
    at += n_desc[k];
  }
  return cas_build_buckets(c, bids.data(), n_desc, n_images, n_groups, bits_per_bucket);
}
int mvgx_cascade_hash_regions(mvgx_cascade_ctx* c, const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images, uint32_t dim,
                              const float* zero_mean, uint32_t n_groups, uint32_t bits_per_bucket, uint32_t random_seed,
                              uint8_t* const* hash_codes_out, uint16_t* const* bucket_ids_out) {
  return mvgx_cascade_hash_regions_typed(c, 0, reinterpret_cast<const void* const*>(desc_rows), n_desc, n_images, dim, zero_mean, n_groups, bits_per_bucket,
                                         random_seed, hash_codes_out, bucket_ids_out);
}

int mvgx_cascade_run(mvgx_cascade_ctx* c, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq, mvgx_match_stats* stats) {
  return bf_run(c, pairs_IJ, n_pairs, ratio_sq, stats);
}

int mvgx_cascade_results(mvgx_cascade_ctx* c, const uint64_t** offsets, const uint32_t** ij) {
  MVGX_REQUIRE(c && offsets && ij, MVGX_ERR_ARG, "mvgx_cascade_results: NULL argument");
  *offsets = c->res_offsets.data();
  *ij = c->res_ij.data();
  return MVGX_OK;
}

}  // extern "C"
