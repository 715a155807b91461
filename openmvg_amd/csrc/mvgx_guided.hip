// libmvgx_hip.so - guided matching on gfx950: the second stage of openMVG's a-contrario geometric filters.
//
// Reference semantics reproduced (paths under /root/reference/src/openMVG):
//   robust_estimation/guided_matching.hpp:178-227   GuidedMatching(model, camL, lRegions, camR, rRegions, errorTh, distRatio, out): for
//                                                   every left feature i, over the right features j whose GEOMETRIC error under the
//                                                   model is below errorTh, the best and second-best SquaredDescriptorDistance
//                                                   (distanceRatio<double>, :68-112: first minimum wins, a repeated minimum becomes the
//                                                   second best); i is kept iff a second best exists and bd < distRatio * sbd; IndMatch(i, idx)
//                                                   in ascending i (getDeduplicated sorts by (i, j): every i occurs once)
//   matching_image_collection/F_ACRobust.hpp:109-152   the caller: EpipolarDistanceError with m_F, Square(m_dPrecision_robust),
//   matching_image_collection/E_ACRobust.hpp:153-215   Square(dDistanceRatio); E: F = K2^-T E K1^-1 first (the caller passes that F);
//   matching_image_collection/H_ACRobust.hpp:136-180   H: AsymmetricError with m_H
//   multiview/solver_fundamental_kernel.cpp:157-166    EpipolarDistanceError = (y~ . F x~)^2 / |(F x~)_xy|^2
//   multiview/solver_homography_kernel.hpp:59-63       AsymmetricError = |y - hnormalized(H x~)|^2
//   features/regions_factory.hpp (Scalar_Regions::SquaredDescriptorDistance) + matching/metric.hpp:55-93: L2<uint8_t>, exact int
//   features/scalar_regions.hpp:107-116 + matching/metric.hpp:95-131: L2<float> (AKAZE_Float_Regions), float sums in groups of four -
//                                                   result += ((d0 d0 + d1 d1) + d2 d2) + d3 d3, no fused multiply-add - widened to double
//   features/binary_regions.hpp:109-120 + matching/metric_hamming.hpp: Hamming<unsigned char> SQUARED (AKAZE_Binary_Regions, 64 bytes)
//
// Device formulation: an O(nI nJ) predicate with a sparse descriptor stage - the brute-force matcher with a geometric mask. One lane
// owns one left feature; the right features arrive through the scalar data path (wave-uniform position, norm and descriptor row), so
// a geometric test is a handful of fp64 operations per lane and the descriptor distance (v_dot4_u32_u8 on the row the lane keeps in
// registers, |a|^2 + |b|^2 - 2 a.b, exact) runs only in the wave-iterations where some lane passed. The error is rounded operation by
// operation as the reference's build does (no FMA). For the epipolar error the quotient dt^2 / den is compared with the threshold
// exactly as written, but the division itself is only carried out when dt^2 is within 1e-15 (relative) of errorTh * den - everywhere
// else the comparison is decided without it (|fl(q / den) - q / den| <= 2^-53 q / den).
#include <hip/hip_runtime.h>

#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mvgx_common.h"

namespace {

using mvgx::set_error;

#ifdef __HIPCC__
__device__ __forceinline__ double g_mul(double a, double b) { return __ocml_mul_rte_f64(a, b); }
__device__ __forceinline__ double g_add(double a, double b) { return __ocml_add_rte_f64(a, b); }
__device__ __forceinline__ double g_sub(double a, double b) { return __ocml_sub_rte_f64(a, b); }
#else   // the HIP emulation of the test-suite
__device__ __forceinline__ double g_mul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double g_add(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double g_sub(double a, double b) { return __dadd_rn(a, -b); }
#endif

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kThreads = 256;
constexpr int kQueue = 16;   // right features a lane queues for the descriptor stage before the wave drains the queues (LDS: 16 KB per workgroup)

struct GuidedParams {
  const double2* xy;          // undistorted positions, all images
  const uint32_t* desc;       // descriptors as dwords, DW per feature
  const int* norm;            // |a|^2 per feature
  const uint64_t* feat_start; // n_images + 1
  const uint32_t* pairs;      // (I, J)
  const double* models;       // 9 per pair
  const double* th;           // per pair
  const uint2* work;          // (pair, first left feature of the block)
  const uint64_t* left_start; // per pair: first row of `best`
  uint32_t* best;             // per (pair, left feature): right feature or kNone
  uint32_t* count;            // per pair
  unsigned long long* counters;   // [0] geometric tests passed, [1] wave-iterations with a descriptor stage (statistics)
  double ratio_sq;
};

template <int DW>
__global__ __launch_bounds__(256) void desc_norms_kernel(const uint32_t* __restrict__ desc, uint64_t n, int* __restrict__ norm) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < DW; ++k) { const uint32_t v = desc[i * DW + k]; s = __builtin_amdgcn_udot4(v, v, s, false); }
  norm[i] = (int)s;
}

// KIND 0: EpipolarDistanceError (model = F), 1: AsymmetricError (model = H)
// TYPE (round 6): 0 uint8 rows under L2<uint8_t> (int), 1 float rows under L2<float> (the reference's float sums, in its order), 2 binary rows
// under the squared Hamming distance (int). distanceRatio<double> sees these values widened to double: comparing them in their own type is the
// same comparison, and "no second best yet" (its numeric_limits<double>::max()) is INT_MAX / +infinity here - a value no distance takes
// (a float distance that overflowed to +infinity fails `dist < max` in the reference as it fails `dist < infinity` here).
template <int TYPE> struct GuidedDist { using type = int; static __device__ __forceinline__ int none() { return INT_MAX; } };
template <> struct GuidedDist<1> { using type = float; static __device__ __forceinline__ float none() { return __builtin_inff(); } };
template <int KIND, int DW, int TYPE = 0>
__global__ __launch_bounds__(kThreads) void guided_match_kernel(GuidedParams P) {
  using dist_t = typename GuidedDist<TYPE>::type;
  const uint2 wk = P.work[blockIdx.x];
  const uint32_t p = wk.x;
  const uint32_t I = P.pairs[2 * p], J = P.pairs[2 * p + 1];
  const uint64_t fI = P.feat_start[I], fJ = P.feat_start[J];
  const uint32_t nI = (uint32_t)(P.feat_start[I + 1] - fI), nJ = (uint32_t)(P.feat_start[J + 1] - fJ);
  const double* __restrict__ M = P.models + 9 * (size_t)p;
  const double th = P.th[p];
  const uint32_t i = wk.y + threadIdx.x;
  const bool active = i < nI;
  const double2 x = P.xy[fI + (active ? i : 0)];
  // what depends on the left feature alone
  double a0, a1, a2, den = 0.0, t_lo = 0.0, t_hi = 0.0;
  {
    const double v0 = g_add(g_add(g_mul(M[0], x.x), g_mul(M[1], x.y)), M[2]);
    const double v1 = g_add(g_add(g_mul(M[3], x.x), g_mul(M[4], x.y)), M[5]);
    const double v2 = g_add(g_add(g_mul(M[6], x.x), g_mul(M[7], x.y)), M[8]);
    if (KIND == 0) {
      a0 = v0; a1 = v1; a2 = v2;
      den = g_add(g_mul(v0, v0), g_mul(v1, v1));
      const double t = th * den;   // (only brackets the exact comparison below)
      t_lo = t * (1.0 - 1e-15); t_hi = t * (1.0 + 1e-15);
    } else {
      a0 = v0 / v2; a1 = v1 / v2; a2 = 0.0;
    }
  }
  // (this lane's descriptor row stays in memory: held in registers across the test loop - 32 of them - it halved the waves per SIMD, and
  // the test loop lives on occupancy; a drain step reads both rows 16 bytes at a time while the other waves run their tests)
  const uint4* __restrict__ lrow = reinterpret_cast<const uint4*>(P.desc + (fI + (active ? i : 0)) * DW);
  const int na = TYPE == 0 ? P.norm[fI + (active ? i : 0)] : 0;
  dist_t bd = GuidedDist<TYPE>::none(), sbd = GuidedDist<TYPE>::none();
  uint32_t idx = 0;
  unsigned long long n_pass = 0, n_stage = 0;
  const double2* __restrict__ xyJ = P.xy + fJ;
  const int* __restrict__ normJ = P.norm + fJ;
  const uint32_t* __restrict__ descJ = P.desc + fJ * DW;
  // Right features four at a time: their positions are one 64-byte scalar load, the four geometric tests run back to back, and only a
  // group in which some lane passed looks at descriptors (one right feature after the other, in index order: distanceRatio's update
  // depends on the order). One feature per trip - load, wait, test - left the vector ALU idle two cycles out of three (call r5_13).
  auto geometric = [&](const double2 y) -> bool {
    if (KIND == 0) {
      // F_x . (y0, y1, 1) as Eigen's unrolled reduction of a three-element expression sums it: c0 + (c1 + c2)
      const double dt = g_add(g_mul(a0, y.x), g_add(g_mul(a1, y.y), a2));
      const double q = g_mul(dt, dt);
      bool pass = q < t_lo;
      if (!pass && !(q > t_hi)) pass = q / den < th;   // within rounding of the bound (or a degenerate line): the reference's expression itself
      return pass;
    } else {
      const double dx = g_sub(y.x, a0), dy = g_sub(y.y, a1);
      return g_add(g_mul(dx, dx), g_mul(dy, dy)) < th;
    }
  };
  // The descriptor stage, dense (round 5, second form). 0.4 % of the geometric tests pass under a fundamental matrix: a wave that looked at
  // a right feature's descriptor as soon as ANY of its lanes passed ran that stage for 1.1 useful lanes of 64, in two of three groups of
  // four right features - the larger half of the kernel. Instead every lane queues the right features that passed ITS test (a column of
  // kQueue words in LDS, filled in index order), and when a column is nearly full - or at the end - the wave drains the queues: step k,
  // every lane with more than k entries takes the distance to ITS k-th candidate (its row gathered through the vector memory path, 16 bytes
  // at a time) and updates its best / second best. A lane still walks its candidates in index order - distanceRatio's update depends on
  // nothing else - so the lists are the same; a drain step has ~30 useful lanes instead of 1.1 and there are ~50 x fewer of them.
  __shared__ uint32_t queue_lds[kQueue * kThreads];
  uint32_t* const q = queue_lds + (threadIdx.x >> 6) * (kQueue * 64) + (threadIdx.x & 63);   // entry k of this lane: q[64 k]
  uint32_t n_q = 0;
  auto drain = [&]() {
    uint32_t most = n_q;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) most = max(most, (uint32_t)__shfl_xor((int)most, off));
    for (uint32_t k = 0; k < most; ++k) {   // (wave-uniform bound)
      n_stage += 1;
      if (k < n_q) {
        const uint32_t j = q[64 * k];
        const uint4* __restrict__ rj = reinterpret_cast<const uint4*>(descJ + (size_t)j * DW);
        dist_t d;
        if (TYPE == 0) {
          unsigned dot = 0;
#pragma unroll 2
          for (int k4 = 0; k4 < DW / 4; ++k4) {
            const uint4 l = lrow[k4], r = rj[k4];
            dot = __builtin_amdgcn_udot4(l.x, r.x, dot, false);
            dot = __builtin_amdgcn_udot4(l.y, r.y, dot, false);
            dot = __builtin_amdgcn_udot4(l.z, r.z, dot, false);
            dot = __builtin_amdgcn_udot4(l.w, r.w, dot, false);
          }
          d = (dist_t)(na + normJ[j] - 2 * (int)dot);
        } else if (TYPE == 1) {   // metric.hpp:95-131: four differences, their squares summed left to right, added to the running float
          float result = 0.f;
#pragma unroll 2
          for (int k4 = 0; k4 < DW / 4; ++k4) {
            const uint4 l = lrow[k4], r = rj[k4];
            const float d0 = __fsub_rn(__uint_as_float(l.x), __uint_as_float(r.x)), d1 = __fsub_rn(__uint_as_float(l.y), __uint_as_float(r.y));
            const float d2 = __fsub_rn(__uint_as_float(l.z), __uint_as_float(r.z)), d3 = __fsub_rn(__uint_as_float(l.w), __uint_as_float(r.w));
            const float g = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)), __fmul_rn(d3, d3));
            result = __fadd_rn(result, g);
          }
          d = (dist_t)result;
        } else {
          int h = 0;
#pragma unroll 2
          for (int k4 = 0; k4 < DW / 4; ++k4) {
            const uint4 l = lrow[k4], r = rj[k4];
            h += __popc(l.x ^ r.x) + __popc(l.y ^ r.y) + __popc(l.z ^ r.z) + __popc(l.w ^ r.w);
          }
          d = (dist_t)(h * h);   // binary_regions.hpp:119: descDist * descDist
        }
        n_pass += 1;
        if (d < bd) { sbd = bd; bd = d; idx = j; }
        else if (d < sbd) sbd = d;
      }
    }
    n_q = 0;
  };
  static_assert(DW % 4 == 0, "descriptor rows are read 16 bytes at a time");
  uint32_t j = 0;
  for (; j + 4 <= nJ; j += 4) {
    const double2 y0 = xyJ[j], y1 = xyJ[j + 1], y2 = xyJ[j + 2], y3 = xyJ[j + 3];   // (wave-uniform: scalar loads)
    const bool p0 = active && geometric(y0), p1 = active && geometric(y1), p2 = active && geometric(y2), p3 = active && geometric(y3);
    if (__ballot(p0 || p1 || p2 || p3) == 0ull) continue;   // (uniform: under a homography 15 groups of 16 end here)
    if (p0) { q[64 * n_q] = j; ++n_q; }
    if (p1) { q[64 * n_q] = j + 1; ++n_q; }
    if (p2) { q[64 * n_q] = j + 2; ++n_q; }
    if (p3) { q[64 * n_q] = j + 3; ++n_q; }
    if (__ballot(n_q + 4 > (uint32_t)kQueue)) drain();   // (uniform) some column could not take another group
  }
  for (; j < nJ; ++j) {
    if (active && geometric(xyJ[j])) { q[64 * n_q] = j; ++n_q; }
    if (__ballot(n_q + 4 > (uint32_t)kQueue)) drain();
  }
  drain();
  const bool valid = active && sbd != GuidedDist<TYPE>::none() && (double)bd < P.ratio_sq * (double)sbd;
  if (active) P.best[P.left_start[p] + i] = valid ? idx : kNone;
  const unsigned long long m = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&P.count[p], (unsigned)__popcll(m));
  if (P.counters) {
    // one atomic per wave (statistics only; the counts of a lane are far below 2^53)
    double tot = (double)n_pass;
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&P.counters[0], (unsigned long long)tot); atomicAdd(&P.counters[1], n_stage); }
  }
}

// ascending i: one wave per pair walks its slice of `best`
__global__ __launch_bounds__(64) void guided_compact_kernel(const uint32_t* __restrict__ best, const uint64_t* __restrict__ left_start,
                                                            const uint64_t* __restrict__ feat_start, const uint32_t* __restrict__ pairs,
                                                            const uint64_t* __restrict__ match_start, uint32_t* __restrict__ ij) {
  const uint32_t p = blockIdx.x;
  const uint32_t I = pairs[2 * p];
  const uint32_t nI = (uint32_t)(feat_start[I + 1] - feat_start[I]);
  if (match_start[p + 1] == match_start[p]) return;
  const int lane = threadIdx.x;
  uint64_t at = match_start[p];
  for (uint32_t base = 0; base < nI; base += 64) {
    const uint32_t i = base + lane;
    const uint32_t v = i < nI ? best[left_start[p] + i] : kNone;
    const unsigned long long m = __ballot(v != kNone);
    if (v != kNone) {
      const uint64_t o = at + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
      ij[2 * o] = i; ij[2 * o + 1] = v;
    }
    at += (uint64_t)__popcll(m);
  }
}

template <typename T>
struct DevArray {
  T* p = nullptr;
  DevArray() = default;
  DevArray(const DevArray&) = delete;   // (a launch must be handed the raw pointer: the test-suite's emulation captures launch arguments by value)
  DevArray& operator=(const DevArray&) = delete;
  // (from the call's arena - slabs of the library's device cache: fourteen hipMalloc / hipFree pairs per call were up to 23 ms of it)
  int alloc(mvgx::Arena& a, size_t n) { return a.alloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T)); }
};

template <int KIND, int DW, int TYPE>
void launch_match(const GuidedParams& P, uint32_t n_work, hipStream_t s) {
  hipLaunchKernelGGL((guided_match_kernel<KIND, DW, TYPE>), dim3(n_work), dim3(kThreads), 0, s, P);
}

}  // namespace

extern "C" {

void mvgx_host_free(void* p) { free(p); }

int mvgx_guided_match_u8(int device, const double* feat_xy, const uint8_t* desc, uint32_t desc_bytes, const uint64_t* feat_start, uint32_t n_images,
                         const uint32_t* pairs, const double* models, const double* error_th, uint64_t n_pairs, int kind, double dist_ratio_sq,
                         uint64_t* match_start, uint32_t** matches_ij, mvgx_guided_stats* stats) {
  return mvgx_guided_match(device, feat_xy, desc, MVGX_DESC_U8, desc_bytes, feat_start, n_images, pairs, models, error_th, n_pairs, kind, dist_ratio_sq,
                           match_start, matches_ij, stats);
}

int mvgx_guided_match(int device, const double* feat_xy, const void* desc, int desc_type, uint32_t desc_bytes, const uint64_t* feat_start, uint32_t n_images,
                      const uint32_t* pairs, const double* models, const double* error_th, uint64_t n_pairs, int kind, double dist_ratio_sq,
                      uint64_t* match_start, uint32_t** matches_ij, mvgx_guided_stats* stats) {
  const auto t_enter = std::chrono::steady_clock::now();
  MVGX_REQUIRE(feat_start && match_start && matches_ij && (n_pairs == 0 || (pairs && models && error_th)), MVGX_ERR_ARG, "mvgx_guided_match: NULL argument");
  MVGX_REQUIRE(kind == MVGX_GUIDED_FUNDAMENTAL || kind == MVGX_GUIDED_HOMOGRAPHY, MVGX_ERR_ARG, "mvgx_guided_match: kind must be 0 (fundamental) or 1 (homography)");
  MVGX_REQUIRE(desc_type == MVGX_DESC_U8 || desc_type == MVGX_DESC_F32 || desc_type == MVGX_DESC_BINARY, MVGX_ERR_ARG, "mvgx_guided_match: descriptor type %d", desc_type);
  MVGX_REQUIRE(desc_type != MVGX_DESC_U8 || desc_bytes == 64 || desc_bytes == 128 || desc_bytes == 144, MVGX_ERR_UNSUPPORTED,
               "mvgx_guided_match: uint8 descriptors of %u bytes (64, 128 and 144 are built)", desc_bytes);
  MVGX_REQUIRE(desc_type != MVGX_DESC_F32 || desc_bytes == 256 || desc_bytes == 512, MVGX_ERR_UNSUPPORTED,
               "mvgx_guided_match: float descriptors of %u bytes (64 and 128 floats are built)", desc_bytes);
  MVGX_REQUIRE(desc_type != MVGX_DESC_BINARY || desc_bytes == 32 || desc_bytes == 64, MVGX_ERR_UNSUPPORTED,
               "mvgx_guided_match: binary descriptors of %u bytes (32 and 64 are built)", desc_bytes);
  MVGX_REQUIRE(dist_ratio_sq >= 0.0 && std::isfinite(dist_ratio_sq), MVGX_ERR_ARG, "mvgx_guided_match: distance ratio");
  *matches_ij = nullptr;
  const uint64_t n_feat = n_images ? feat_start[n_images] : 0;
  for (uint32_t k = 0; k < n_images; ++k) MVGX_REQUIRE(feat_start[k] <= feat_start[k + 1], MVGX_ERR_ARG, "mvgx_guided_match: feat_start must ascend");
  MVGX_REQUIRE(n_feat == 0 || (feat_xy && desc), MVGX_ERR_ARG, "mvgx_guided_match: NULL feature arrays");
  // work list; a pair whose bound is not finite gives no match (the functors test m_dPrecision_robust != infinity), neither does an empty image
  std::vector<uint64_t> left_start(n_pairs + 1, 0);
  std::vector<uint2> work;
  for (uint64_t p = 0; p < n_pairs; ++p) {
    const uint32_t I = pairs[2 * p], J = pairs[2 * p + 1];
    MVGX_REQUIRE(I < n_images && J < n_images, MVGX_ERR_ARG, "mvgx_guided_match: pair %llu names image %u / %u of %u", (unsigned long long)p, I, J, n_images);
    MVGX_REQUIRE(p < (uint64_t)UINT32_MAX, MVGX_ERR_ARG, "mvgx_guided_match: more than 2^32 pairs in one call");
    const uint64_t nI = feat_start[I + 1] - feat_start[I], nJ = feat_start[J + 1] - feat_start[J];
    MVGX_REQUIRE(nI < (1ull << 31) && nJ < (1ull << 31), MVGX_ERR_ARG, "mvgx_guided_match: image with 2^31 or more features");
    left_start[p + 1] = left_start[p] + nI;
    if (!(error_th[p] > 0.0) || !std::isfinite(error_th[p]) || nI == 0 || nJ == 0) continue;
    for (uint64_t i0 = 0; i0 < nI; i0 += kThreads) work.push_back(make_uint2((uint32_t)p, (uint32_t)i0));
  }
  std::memset(match_start, 0, (n_pairs + 1) * sizeof(uint64_t));
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (stats) stats->n_pairs = n_pairs;
  if (n_pairs == 0) return MVGX_OK;
  int rc = mvgx::select_device(device);
  if (rc) return rc;
  // (declared BEFORE the stream guard - destroyed after it: on every exit the stream is drained before the slabs of this call go back
  // to the process-wide cache, where a concurrent call could pick them up while kernels enqueued here still write - ADVICE r5)
  mvgx::Arena arena;   // device memory of this call
  hipStream_t stream = nullptr;
  if ((rc = mvgx::acquire_stream(&stream))) return rc;
  int dev = 0;
  (void)hipGetDevice(&dev);
  struct StreamGuard { int d; hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); mvgx::release_stream(d, s); } } guard{dev, stream};

  const int DW = (int)desc_bytes / 4;
  DevArray<double2> d_xy; DevArray<uint32_t> d_desc; DevArray<int> d_norm; DevArray<uint64_t> d_fs, d_ls, d_ms; DevArray<uint32_t> d_pairs, d_best, d_count, d_ij;
  DevArray<double> d_models, d_th; DevArray<uint2> d_work; DevArray<unsigned long long> d_ctr;
  if ((rc = d_xy.alloc(arena, n_feat)) || (rc = d_desc.alloc(arena, n_feat * DW)) || (rc = d_norm.alloc(arena, n_feat)) || (rc = d_fs.alloc(arena, n_images + 1)) || (rc = d_ls.alloc(arena, n_pairs + 1)) ||
      (rc = d_ms.alloc(arena, n_pairs + 1)) || (rc = d_pairs.alloc(arena, 2 * n_pairs)) || (rc = d_best.alloc(arena, left_start[n_pairs])) || (rc = d_count.alloc(arena, n_pairs)) ||
      (rc = d_models.alloc(arena, 9 * n_pairs)) || (rc = d_th.alloc(arena, n_pairs)) || (rc = d_work.alloc(arena, work.size())) || (rc = d_ctr.alloc(arena, 2)))
    return rc;
  if (n_feat) {
    MVGX_HIP(hipMemcpyAsync(d_xy.p, feat_xy, n_feat * sizeof(double2), hipMemcpyHostToDevice, stream));
    MVGX_HIP(hipMemcpyAsync(d_desc.p, desc, n_feat * (size_t)desc_bytes, hipMemcpyHostToDevice, stream));
  }
  MVGX_HIP(hipMemcpyAsync(d_fs.p, feat_start, (n_images + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
  MVGX_HIP(hipMemcpyAsync(d_ls.p, left_start.data(), (n_pairs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
  MVGX_HIP(hipMemcpyAsync(d_pairs.p, pairs, 2 * n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  MVGX_HIP(hipMemcpyAsync(d_models.p, models, 9 * n_pairs * sizeof(double), hipMemcpyHostToDevice, stream));
  MVGX_HIP(hipMemcpyAsync(d_th.p, error_th, n_pairs * sizeof(double), hipMemcpyHostToDevice, stream));
  if (!work.empty()) MVGX_HIP(hipMemcpyAsync(d_work.p, work.data(), work.size() * sizeof(uint2), hipMemcpyHostToDevice, stream));
  MVGX_HIP(hipMemsetAsync(d_count.p, 0, n_pairs * sizeof(uint32_t), stream));
  MVGX_HIP(hipMemsetAsync(d_ctr.p, 0, 2 * sizeof(unsigned long long), stream));
  if (left_start[n_pairs]) MVGX_HIP(hipMemsetAsync(d_best.p, 0xFF, left_start[n_pairs] * sizeof(uint32_t), stream));   // pairs without work: nothing kept
  hipEvent_t e0 = nullptr, e1 = nullptr;
  MVGX_HIP(hipEventCreate(&e0));
  MVGX_HIP(hipEventCreate(&e1));
  struct EventGuard { hipEvent_t a, b; ~EventGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } eguard{e0, e1};
  MVGX_HIP(hipEventRecord(e0, stream));
  if (n_feat && desc_type == MVGX_DESC_U8) {
    const unsigned nb = (unsigned)((n_feat + 255) / 256);
    const uint32_t* const pd = d_desc.p;
    int* const pn = d_norm.p;
    if (DW == 16) hipLaunchKernelGGL(desc_norms_kernel<16>, dim3(nb), dim3(256), 0, stream, pd, n_feat, pn);
    else if (DW == 32) hipLaunchKernelGGL(desc_norms_kernel<32>, dim3(nb), dim3(256), 0, stream, pd, n_feat, pn);
    else hipLaunchKernelGGL(desc_norms_kernel<36>, dim3(nb), dim3(256), 0, stream, pd, n_feat, pn);
  }
  GuidedParams P;
  P.xy = d_xy.p; P.desc = d_desc.p; P.norm = d_norm.p; P.feat_start = d_fs.p; P.pairs = d_pairs.p; P.models = d_models.p; P.th = d_th.p;
  P.work = d_work.p; P.left_start = d_ls.p; P.best = d_best.p; P.count = d_count.p; P.counters = stats ? d_ctr.p : nullptr; P.ratio_sq = dist_ratio_sq;
  if (!work.empty()) {
    const uint32_t nw = (uint32_t)work.size();
#define MVGX_GUIDED_CASE(K, D, TY) if (kind == K && DW == D && desc_type == TY) launch_match<K, D, TY>(P, nw, stream);
    MVGX_GUIDED_CASE(0, 16, 0) MVGX_GUIDED_CASE(0, 32, 0) MVGX_GUIDED_CASE(0, 36, 0) MVGX_GUIDED_CASE(1, 16, 0) MVGX_GUIDED_CASE(1, 32, 0) MVGX_GUIDED_CASE(1, 36, 0)
    MVGX_GUIDED_CASE(0, 64, 1) MVGX_GUIDED_CASE(0, 128, 1) MVGX_GUIDED_CASE(1, 64, 1) MVGX_GUIDED_CASE(1, 128, 1)
    MVGX_GUIDED_CASE(0, 8, 2) MVGX_GUIDED_CASE(0, 16, 2) MVGX_GUIDED_CASE(1, 8, 2) MVGX_GUIDED_CASE(1, 16, 2)
#undef MVGX_GUIDED_CASE
    MVGX_HIP(hipGetLastError());
  }
  MVGX_HIP(hipEventRecord(e1, stream));
  std::vector<uint32_t> count(n_pairs);
  MVGX_HIP(hipMemcpyAsync(count.data(), d_count.p, n_pairs * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  MVGX_HIP(hipStreamSynchronize(stream));
  for (uint64_t p = 0; p < n_pairs; ++p) match_start[p + 1] = match_start[p] + count[p];
  const uint64_t total = match_start[n_pairs];
  uint32_t* out = static_cast<uint32_t*>(malloc(std::max<uint64_t>(total, 1) * 2 * sizeof(uint32_t)));
  MVGX_REQUIRE(out, MVGX_ERR_HIP, "mvgx_guided_match: out of host memory (%llu matches)", (unsigned long long)total);
  if (total) {
    if ((rc = d_ij.alloc(arena, 2 * total))) { free(out); return rc; }
    hipError_t e = hipMemcpyAsync(d_ms.p, match_start, (n_pairs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
      const uint32_t *const pb = d_best.p, *const pp = d_pairs.p;
      const uint64_t *const pl = d_ls.p, *const pf = d_fs.p, *const pm = d_ms.p;
      uint32_t* const po = d_ij.p;
      hipLaunchKernelGGL(guided_compact_kernel, dim3((unsigned)n_pairs), dim3(64), 0, stream, pb, pl, pf, pp, pm, po);
      e = hipGetLastError();
    }
    // (through page-locked memory of the library's cache: a copy straight into the fresh malloc block pins its pages on the way - up to
    // 20 ms for 16 MB, every few calls)
    mvgx::HostArena staging;
    uint32_t* pinned = nullptr;
    if (e == hipSuccess && staging.array(&pinned, 2 * total) != MVGX_OK) pinned = nullptr;
    if (e == hipSuccess) e = hipMemcpyAsync(pinned ? pinned : out, d_ij.p, 2 * total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess && pinned) std::memcpy(out, pinned, 2 * total * sizeof(uint32_t));
    if (e != hipSuccess) {   // (drained before the page-locked staging block goes back to the cache: a copy into it may still be in flight)
      (void)hipStreamSynchronize(stream);
      free(out); set_error("mvgx_guided_match: %s", hipGetErrorString(e)); return MVGX_ERR_HIP;
    }
  }
  *matches_ij = out;
  if (stats) {
    unsigned long long ctr[2] = {0, 0};
    MVGX_HIP(hipMemcpy(ctr, d_ctr.p, sizeof(ctr), hipMemcpyDeviceToHost));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    stats->n_matches = total;
    stats->n_geometric_tests = 0;
    for (const uint2& w : work) {
      const uint32_t I = pairs[2 * (size_t)w.x], J = pairs[2 * (size_t)w.x + 1];
      const uint64_t nI = feat_start[I + 1] - feat_start[I], nJ = feat_start[J + 1] - feat_start[J];
      stats->n_geometric_tests += std::min<uint64_t>(kThreads, nI - w.y) * nJ;
    }
    stats->n_geometric_passed = ctr[0];
    stats->n_descriptor_stages = ctr[1];
    stats->kernel_ms = ms;
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
  }
  return MVGX_OK;
}

}  // extern "C"
