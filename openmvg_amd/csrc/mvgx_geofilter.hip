// libmvgx_hip.so - a-contrario fundamental-matrix filter of putative matches on gfx950.
//
// Reference path reproduced (paths under /root/reference/src/openMVG):
//   matching_image_collection/GeometricFilter.hpp:66-131     ImageCollectionGeometricFilter::Robust_model_estimation: one robust
//                                                             estimation per image pair of the putative-match container
//   matching_image_collection/F_ACRobust.hpp:65-122          GeometricFilter_FMatrix_AC::Robust_estimation
//   robust_estimation/robust_estimator_ACRansacKernelAdaptator.hpp:104-202   ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError>
//   robust_estimation/robust_estimator_ACRansac.hpp:339-489  ACRANSAC, :196-262 the quantified NFA (20-bin histogram), :58-119 tables
//   robust_estimation/rand_sampling.hpp:43-110               the two UniformSample forms; std::mt19937 + libstdc++ 11 uniform_int_distribution
//   multiview/solver_fundamental_kernel.cpp:37-93,157-166    seven-point solver, EpipolarDistanceError; numeric/poly.h:32-96 the cubic
//
// Mapping. AC-RANSAC is a sequential program per image pair - the sample of iteration k + 1 depends on whether iteration k found a
// better model (the sampling pool shrinks to its inliers) - and there are as many independent programs as image pairs with matches
// (configs[1]: up to 499 500). ONE WAVE runs one pair from start to end: control flow is wave-uniform (every lane tracks the same
// scalars), the per-correspondence work of an iteration (residuals of up to three models, histogram, inlier count, pool rebuild)
// is spread over the 64 lanes, the 7 x 9 elimination of the minimal solver over 63 lanes (one matrix element each, pivots and
// rows by cross-lane shuffles). The Mersenne-Twister state, the sampling pool and the pair's log-combinatorial tables live in
// the wave's slice of LDS; the normalised correspondences (32 bytes each) stay in L2 - a pair re-reads its few KiB once per model.
// No workgroup barrier anywhere: the four waves of a workgroup are four unrelated pairs.
//
// What is exact and what is not (parity policy): the sample sequence (generator, Lemire's bounded draw, both
// samplers), the float tables, histogram bins, NFA arithmetic (no contraction) and the control flow are the reference's. The
// null space of the 7 x 9 system comes from complete-pivoting elimination instead of Eigen's eigenvectors of A^T A: the pencil's
// fundamental matrices agree up to scale and rounding, so residuals differ in the last bits and - when the cubic is badly
// conditioned in one basis - occasionally by more; a pair whose decisive residual sits within that distance of a histogram edge
// can end with another inlier set. The compiled reference has the same sensitivity to its own Eigen build.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>
#include <vector>

#include "mvgx_common.h"

namespace {

using mvgx::set_error;

constexpr int kBins = 20;        // robust_estimator_ACRansac.hpp:221
constexpr int kMinSamples = 7, kMaxModels = 3;   // the fundamental-matrix model (SevenPointSolver)
// The estimated model: the a-contrario loop is the same program for both, what differs is the minimal solver, the residual, the
// sample size and - on the host - logalpha0 / multError of the NFA (point-to-line for F, point-to-point for H).
// kModelEA8 / kModelEU3: the essential matrix with the ANGULAR residual on bearing vectors (E_ACRobust_Angular.hpp:33-191, spherical
// cameras): EightPointRelativePoseSolver / ThreePointUprightRelativePoseSolver, one model per sample, no normalisation, no pixels.
// kModelEO: the orthographic essential matrix (Eo_Robust.hpp:35-165): ThreePointSolver (two models per sample) and
// OrthographicSymmetricEpipolarDistanceError on the hnormalized bearing vectors, which travel in the pixel arrays (no normalisation).
enum GeoModel { kModelF = 0, kModelH = 1, kModelE = 2, kModelEA8 = 3, kModelEU3 = 4, kModelEO = 5 };
template <int MODEL> constexpr bool model_is_angular() { return MODEL == kModelEA8 || MODEL == kModelEU3; }
template <int MODEL> constexpr int model_min_samples() {   // Solver::MINIMUM_SAMPLES
  return MODEL == kModelH ? 4 : MODEL == kModelE ? 5 : MODEL == kModelEA8 ? 8 : (MODEL == kModelEU3 || MODEL == kModelEO) ? 3 : kMinSamples;
}
template <int MODEL> constexpr int model_sample_slots() { return MODEL == kModelEA8 ? 8 : 7; }   // length of the kernel's sample array
constexpr int kMtN = 624, kMtM = 397;

struct GeoPair {   // per pair, prepared on the host (glibc's log10 / hypot / sqrt: the reference's values)
  uint64_t start;
  uint32_t n, pad_;
  double max_threshold, loge0, bins_by_interval;
  double bin_value[kBins];
  double logalpha_bin[kBins];   // logalpha0 + multError * log10(bin_value + FLT_EPSILON)
  double k2it[9], k1i[9];       // essential model only: K2^-T and K1^-1 (row-major), F = K2^-T E K1^-1 (multiview/essential.cpp:48-53)
};
struct GeoResult {
  double F[9];             // best model, normalised coordinates (row-major); valid if have_model
  double error_max, min_nfa;
  uint32_t n_inliers, have_model;
  uint32_t n_iterations, n_models;   // measurement: a-contrario iterations run / models evaluated by this pair's wave
  uint64_t clocks;                   // measurement: shader clocks (s_memtime) between the wave's first and last instruction
};

// ---- arithmetic that must not be contracted into FMAs (the reference's build has none) ----
// (The toolchain's __dmul_rn / __dadd_rn are plain x * y / x + y unless OCML_BASIC_ROUNDED_OPERATIONS is defined on the command line, and
// hipcc's default -ffp-contract=fast-honor-pragmas fuses those into v_fma_f64 - measured in round 4: the orthographic solver differed
// from the reference by 1e-9 until this was found. OCML's rounded operations are opaque to the contraction pass.)
#ifdef __HIPCC__
__device__ __forceinline__ double mul_rn(double a, double b) { return __ocml_mul_rte_f64(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __ocml_add_rte_f64(a, b); }
#else   // the HIP emulation of the test-suite (host compiler; its header keeps the two operations apart)
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
#endif

__device__ __forceinline__ void wave_sync() {   // LDS writes of this wave visible to its other lanes (a wave's LDS operations complete in order)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- std::mt19937: state in LDS, every lane tracks the same index ----
__device__ __forceinline__ void mt_twist(uint32_t* mt, int lane) {
  // mt[i] = mt[(i + 397) % 624] ^ twist(mt[i], mt[i + 1]) for i = 0 .. 623 in order; 64 consecutive i per round: mt[i + 1] is read
  // before any lane of the round writes, mt[i - 227] (i >= 227) was written at least three rounds earlier
  for (int base = 0; base < kMtN - 1; base += 64) {
    const int i = base + lane;
    uint32_t a = 0, b = 0, c = 0;
    if (i < kMtN - 1) { a = mt[i]; b = mt[i + 1]; c = mt[i + kMtM < kMtN ? i + kMtM : i + kMtM - kMtN]; }
    wave_sync();
    if (i < kMtN - 1) {
      const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
      mt[i] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    wave_sync();
  }
  const uint32_t y = (mt[kMtN - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
  const uint32_t v = mt[kMtM - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  wave_sync();
  if (lane == 0) mt[kMtN - 1] = v;
  wave_sync();
}
__device__ __forceinline__ uint32_t mt_next(uint32_t* mt, int& idx, int lane) {
  if (idx >= kMtN) { mt_twist(mt, lane); idx = 0; }
  uint32_t y = mt[idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}
// std::uniform_int_distribution<uint32_t>(a, b) of libstdc++ 11 on a 32-bit engine (bits/uniform_int_dist.h, _S_nd)
__device__ __forceinline__ uint32_t uniform_u32(uint32_t* mt, int& idx, int lane, uint32_t a, uint32_t b) {
  const uint32_t urange = b - a;
  if (urange == 0xffffffffu) return mt_next(mt, idx, lane) + a;
  const uint32_t range = urange + 1;
  uint64_t product = (uint64_t)mt_next(mt, idx, lane) * range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) { product = (uint64_t)mt_next(mt, idx, lane) * range; low = (uint32_t)product; }
  }
  return (uint32_t)(product >> 32) + a;
}

// The same two for a sample drawn AHEAD of the iteration that will use it (essential model, see the kernel): such a draw must be
// revocable, and a twist of the generator's state is not - with may_twist == false a draw that needs one fails instead (nothing changed).
__device__ __forceinline__ bool mt_try_next(uint32_t* mt, int& idx, int lane, bool may_twist, uint32_t& y) {
  if (idx >= kMtN) {
    if (!may_twist) return false;
    mt_twist(mt, lane); idx = 0;
  }
  y = mt[idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return true;
}
__device__ __forceinline__ bool uniform_try_u32(uint32_t* mt, int& idx, int lane, uint32_t a, uint32_t b, bool may_twist, uint32_t& out) {
  const uint32_t urange = b - a;
  uint32_t y;
  if (urange == 0xffffffffu) {
    if (!mt_try_next(mt, idx, lane, may_twist, y)) return false;
    out = y + a;
    return true;
  }
  const uint32_t range = urange + 1;
  if (!mt_try_next(mt, idx, lane, may_twist, y)) return false;
  uint64_t product = (uint64_t)y * range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      if (!mt_try_next(mt, idx, lane, may_twist, y)) return false;
      product = (uint64_t)y * range; low = (uint32_t)product;
    }
  }
  out = (uint32_t)(product >> 32) + a;
  return true;
}

// ---- numeric/poly.h:32-75 ----
__device__ __forceinline__ int solve_cubic(double a, double b, double c, double x[3]) {
  const double eps = 2.220446049250313e-16;
  a /= 3;
  double p = (b - 3 * a * a) / 3;
  double q = (2 * a * a * a - a * b + c) / 2;
  double d = q * q + p * p * p;
  const double tolq = fmax(fabs(2 * a * a * a), fmax(fabs(a * b), fabs(c)));
  const double tolp = fmax(fabs(b), fabs(3 * a * a));
  int n = (d > eps * fmax(p * p * tolp, fabs(q) * tolq) ? 1 : 3);
  if (n == 1) {
    d = pow(fabs(q) + sqrt(d), 1 / 3.0);
    x[0] = d - p / d;
    if (q > 0) x[0] = -x[0];
  } else {
    if (3 * p >= -eps * tolp) { n = 1; x[0] = 0; }
    else {
      p = sqrt(-p);
      q /= p * p * p;
      d = (q <= -1) ? 3.14159265358979323846 : (q >= 1) ? 0.0 : acos(q);
      for (int i = 0; i < 3; ++i) x[i] = -2 * p * cos((d + 2 * 3.14159265358979323846 * i) / 3);
    }
  }
  for (int i = 0; i < n; ++i) x[i] -= a;
  return n;
}

__device__ __forceinline__ double shfl_f64(double v, int src) { return __shfl(v, src); }
__device__ __forceinline__ double lane_value_f64(double v, int src_lane) {   // src_lane wave-uniform: two v_readlane, no LDS crossbar
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
// maximum of a 32-bit key over the wave, the same value in every lane: quad / half-row / row exchanges on the DPP path, the four
// rows through scalar registers (a butterfly of ds_bpermute round trips cost ~2000 clocks per pivot search here)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // lane ^ 1
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // lane ^ 2
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // the other quad of the half row
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // the other half of the row
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(r0, r1), max(r2, r3));
}

// EpipolarDistanceError (solver_fundamental_kernel.cpp:157-166)
__device__ __forceinline__ double epipolar_error(const double (&F)[9], double2 x, double2 y) {
  const double fx0 = F[0] * x.x + F[1] * x.y + F[2], fx1 = F[3] * x.x + F[4] * x.y + F[5], fx2 = F[6] * x.x + F[7] * x.y + F[8];
  const double dt = fx0 * y.x + fx1 * y.y + fx2;
  return dt * dt / (fx0 * fx0 + fx1 * fx1);
}

// AsymmetricError (multiview/solver_homography_kernel.hpp:60-64): |y - hnormalized(H [x; 1])|^2, products and sums rounded one by one
// (the reference's build has no fused multiply-add)
__device__ __forceinline__ double homography_error(const double (&H)[9], double2 x, double2 y) {
  const double v0 = add_rn(add_rn(mul_rn(H[0], x.x), mul_rn(H[1], x.y)), H[2]);
  const double v1 = add_rn(add_rn(mul_rn(H[3], x.x), mul_rn(H[4], x.y)), H[5]);
  const double v2 = add_rn(add_rn(mul_rn(H[6], x.x), mul_rn(H[7], x.y)), H[8]);
  const double dx = y.x - v0 / v2, dy = y.y - v1 / v2;
  return add_rn(mul_rn(dx, dx), mul_rn(dy, dy));
}
// OrthographicSymmetricEpipolarDistanceError (multiview/solver_essential_kernel.hpp:69-78): |E22 + x0 E02 + x1 E12 + y0 E20 + y1 E21|,
// summed left to right, no contraction
__device__ __forceinline__ double ortho_error(const double (&E)[9], double2 x, double2 y) {
  return fabs(add_rn(add_rn(add_rn(add_rn(E[8], mul_rn(x.x, E[2])), mul_rn(x.y, E[5])), mul_rn(y.x, E[6])), mul_rn(y.y, E[7])));
}
template <int MODEL>
__device__ __forceinline__ double model_error(const double (&M)[9], double2 x, double2 y) {
  return MODEL == kModelH ? homography_error(M, x, y) : MODEL == kModelEO ? ortho_error(M, x, y)
                                                                          : epipolar_error(M, x, y);   // (the essential model evaluates its F = K2^-T E K1^-1)
}
// Square(AngularError::Error) (multiview/solver_essential_eight_point.cpp:50-61; ACKernelAdaptor_AngularRadianError::Errors squares
// it): asin(x2 . normalized(E x1)), products and sums rounded one by one in Eigen's order, normalized() = v / sqrt(v . v) when positive
struct Pt3 { double x, y, z; };
__device__ __forceinline__ double angular_error(const double (&E)[9], Pt3 a, Pt3 b) {
  double ex = add_rn(add_rn(mul_rn(E[0], a.x), mul_rn(E[1], a.y)), mul_rn(E[2], a.z));
  double ey = add_rn(add_rn(mul_rn(E[3], a.x), mul_rn(E[4], a.y)), mul_rn(E[5], a.z));
  double ez = add_rn(add_rn(mul_rn(E[6], a.x), mul_rn(E[7], a.y)), mul_rn(E[8], a.z));
  const double n2 = add_rn(add_rn(mul_rn(ex, ex), mul_rn(ey, ey)), mul_rn(ez, ez));
  if (n2 > 0.0) { const double nn = sqrt(n2); ex = ex / nn; ey = ey / nn; ez = ez / nn; }
  const double d = add_rn(add_rn(mul_rn(b.x, ex), mul_rn(b.y, ey)), mul_rn(b.z, ez));
  const double ang = asin(d);
  return mul_rn(ang, ang);
}
template <int MODEL>
__device__ __forceinline__ double model_error(const double (&M)[9], Pt3 x, Pt3 y) { return angular_error(M, x, y); }
// correspondence i of a pair as the model's residual takes it: normalised pixel positions, or - angular models - bearing vectors
template <int MODEL> struct PointOf { using type = double2; };
template <> struct PointOf<kModelEA8> { using type = Pt3; };
template <> struct PointOf<kModelEU3> { using type = Pt3; };
template <int MODEL>
__device__ __forceinline__ typename PointOf<MODEL>::type load_point(const double2* __restrict__ x, const double* __restrict__ b, uint32_t i) {
  if constexpr (model_is_angular<MODEL>()) return Pt3{b[3 * (size_t)i], b[3 * (size_t)i + 1], b[3 * (size_t)i + 2]};
  else return x[i];
}

#include "geofilter_five_point.h"
#include "geofilter_five_point_x4.h"

// The null vector of an 8 x 9 system spread over 36 lanes: lane 9 r0 + c (r0 < 4) holds A[r0][c] in a0 and A[r0 + 4][c] in a1 (the
// other lanes run along with copies and never win a pivot). Gauss-Jordan elimination with complete pivoting - the pivot is the element
// of largest magnitude as a float, lowest (row, column) on ties: key = float bits with the low 7 bits replaced by 127 - (9 r + c) -, the
// pivot row and column reach the lanes through the LDS crossbar; the column left without a pivot carries the null space (H[that
// column] = 1). Every lane gets the vector. A rank-deficient system gives SOME null vector. (Until round 4 every lane of a group of
// eight held a whole row and the wave did the same work eight times over: 1 600 instructions against 760 - same arithmetic per
// element, same results.)
#pragma clang fp contract(on)
__device__ __forceinline__ void null_vector_8x9(double a0, double a1, int lane, double (&H)[9]) {
  const bool live = lane < 36;
  const int r0 = live ? lane / 9 : 3, c = live ? lane - 9 * (lane / 9) : 8;
  uint32_t row_used = 0, col_used = 0;
  int prow[8], pcol[8], n_piv = 0;
#pragma unroll
  for (int step = 0; step < 8; ++step) {
    const bool col_free = live && !((col_used >> c) & 1u);
    const float m0 = (float)fabs(a0), m1 = (float)fabs(a1);
    const uint32_t k0 = (col_free && !((row_used >> r0) & 1u) && m0 > 0.f && m0 == m0) ? ((__float_as_uint(m0) & ~127u) | (uint32_t)(127 - (9 * r0 + c))) : 0u;
    const uint32_t k1 = (col_free && !((row_used >> (r0 + 4)) & 1u) && m1 > 0.f && m1 == m1) ? ((__float_as_uint(m1) & ~127u) | (uint32_t)(127 - (9 * (r0 + 4) + c))) : 0u;
    const uint32_t best = wave_max_u32(k0 > k1 ? k0 : k1);
    if (best == 0u) break;   // rank deficient sample (wave-uniform)
    const int who = 127 - (int)(best & 127u);
    const int pr = who / 9, pc = who - 9 * pr;
    const double of_row = (pr >> 2) ? a1 : a0;                       // (wave-uniform choice: the half the pivot row lives in)
    const double rowv = shfl_f64(of_row, 9 * (pr & 3) + c);          // pivot row, my column
    const double piv = lane_value_f64(of_row, 9 * (pr & 3) + pc);
    const double colv0 = shfl_f64(a0, 9 * r0 + pc), colv1 = shfl_f64(a1, 9 * r0 + pc);   // my rows, pivot column
    const double ip = 1.0 / piv;
    if (r0 != pr) { const double f = colv0 * ip; a0 -= f * rowv; }
    if (r0 + 4 != pr) { const double f = colv1 * ip; a1 -= f * rowv; }
    row_used |= 1u << pr; col_used |= 1u << pc;
    prow[step] = pr; pcol[step] = pc;
    n_piv = step + 1;
  }
  int fc = 0;
  while ((col_used >> fc) & 1u) ++fc;
#pragma unroll
  for (int u = 0; u < 9; ++u) H[u] = (u == fc) ? 1.0 : 0.0;
#pragma unroll
  for (int step = 0; step < 8; ++step) {
    if (step < n_piv) {   // wave-uniform
      const double of_row = (prow[step] >> 2) ? a1 : a0;
      const int base = 9 * (prow[step] & 3);
      const double num = lane_value_f64(of_row, base + fc), den = lane_value_f64(of_row, base + pcol[step]);
      const double h = -num / den;
#pragma unroll
      for (int u = 0; u < 9; ++u) H[u] = (u == pcol[step]) ? h : H[u];
    }
  }
}

// FourPointSolver::Solve on the sample s[0..3] (wave-uniform; multiview/solver_homography_kernel.cpp:37-93): the null vector of the
// 8 x 9 DLT system (two rows per correspondence: [x^T 1 0 0 0 -x' x^T -x'] and [0 0 0 x^T 1 -y' x^T -y']), row-major 3 x 3, by
// null_vector_8x9 above (the reference takes the last right singular vector of the same matrix:
// the same line up to rounding and scale; a rank-deficient sample gives SOME null vector in both, not the same one).
__device__ __forceinline__ void four_point(const double2* __restrict__ x1, const double2* __restrict__ x2, const uint32_t (&s)[7], int lane, double (&H)[9]) {
  // lane 9 r0 + c holds the elements (r0, c) and (r0 + 4, c): rows 2 k / 2 k + 1 belong to sample point k
  const int r0 = lane < 36 ? lane / 9 : 3, c = lane < 36 ? lane - 9 * (lane / 9) : 8;
  const int cm = c % 3;
  auto element = [&](int r) {
    const int pt = r >> 1;
    const uint32_t si = pt == 0 ? s[0] : pt == 1 ? s[1] : pt == 2 ? s[2] : s[3];
    const double2 p1 = x1[si], p2 = x2[si];
    const bool second = r & 1;
    const double t = second ? p2.y : p2.x;
    const double h = cm == 0 ? p1.x : cm == 1 ? p1.y : 1.0;
    return c < 3 ? (second ? 0.0 : h) : c < 6 ? (second ? h : 0.0) : -t * h;
  };
  null_vector_8x9(element(r0), element(r0 + 4), lane, H);
}

// The same on FOUR samples, one per 16-lane row of the wave (samples ahead, see the kernel): lane gl < 9 of a row holds COLUMN gl of the
// row's 8 x 9 system (eight registers) instead of two elements per lane of 36; same pivot choices (keys, tie-breaks), same arithmetic
// per element, the pivot row / column travel through the LDS crossbar inside the row. Every lane of a row gets the row's vector.
__device__ __forceinline__ void null_vector_8x9_x4(double (&a)[8], int lane, double (&H)[9]) {
  const int gl = lane & 15;
  const bool live = gl < 9;
  const int c = live ? gl : 8;
  uint32_t row_used = 0, col_used = 0;
  int prow[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pcol[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_piv = 0;
  bool going = true;
#pragma unroll
  for (int step = 0; step < 8; ++step) {
    uint32_t key = 0u;
    if (live && !((col_used >> c) & 1u)) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float m = (float)fabs(a[r]);
        const uint32_t k = (!((row_used >> r) & 1u) && m > 0.f && m == m) ? ((__float_as_uint(m) & ~127u) | (uint32_t)(127 - (9 * r + c))) : 0u;
        key = k > key ? k : key;
      }
    }
    const uint32_t best = five_point::row_max_u32(key);
    going = going && best != 0u;
    const int who = going ? 127 - (int)(best & 127u) : 0;
    const int pr = who / 9, pc = who - 9 * pr;
    double rowv = a[0];   // pivot row, my column
#pragma unroll
    for (int r = 1; r < 8; ++r) rowv = (pr == r) ? a[r] : rowv;
    const double piv = five_point::row_value_f64(rowv, pc, lane);
    const double ip = 1.0 / piv;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const double colv = five_point::row_value_f64(a[r], pc, lane);   // row r, pivot column
      if (going && r != pr) { const double f = colv * ip; a[r] -= f * rowv; }
    }
    if (going) {
      row_used |= 1u << pr; col_used |= 1u << pc;
      prow[step] = pr; pcol[step] = pc;
      n_piv = step + 1;
    }
  }
  int fc = 0;
  while ((col_used >> fc) & 1u) ++fc;
#pragma unroll
  for (int u = 0; u < 9; ++u) H[u] = (u == fc) ? 1.0 : 0.0;
#pragma unroll
  for (int step = 0; step < 8; ++step) {
    double of_row = a[0];
#pragma unroll
    for (int r = 1; r < 8; ++r) of_row = (prow[step] == r) ? a[r] : of_row;
    const double num = five_point::row_value_f64(of_row, fc, lane), den = five_point::row_value_f64(of_row, pcol[step], lane);
    const double h = -num / den;
    if (step < n_piv) {
#pragma unroll
      for (int u = 0; u < 9; ++u) H[u] = (u == pcol[step]) ? h : H[u];
    }
  }
}
// FourPointSolver::Solve on the sample of this lane's row (s[0..3])
__device__ __forceinline__ void four_point_x4(const double2* __restrict__ x1, const double2* __restrict__ x2, const uint32_t (&s)[4], int lane, double (&H)[9]) {
  const int gl = lane & 15, c = gl < 9 ? gl : 8;
  const int cm = c % 3;
  double a[8];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const double2 p1 = x1[s[pt]], p2 = x2[s[pt]];
    const double h = cm == 0 ? p1.x : cm == 1 ? p1.y : 1.0;
    a[2 * pt] = c < 3 ? h : c < 6 ? 0.0 : -p2.x * h;
    a[2 * pt + 1] = c < 3 ? 0.0 : c < 6 ? h : -p2.y * h;
  }
  null_vector_8x9_x4(a, lane, H);
}
#pragma clang fp contract(fast)

// EightPointRelativePoseSolver::Solve on exactly eight bearing pairs (multiview/solver_essential_eight_point.cpp:17-47; with eight
// columns the projection onto the essential manifold is skipped, :36): E = the null vector of the 8 x 9 epipolar system
// A[r][3 i + j] = x2[i] x1[j] (EncodeEpipolarEquation, solver_fundamental_kernel.hpp:83-93), row-major 3 x 3. The reference takes the
// eigenvector of A^T A of smallest eigenvalue: the same line up to rounding, scale and sign (the residual is invariant to both).
__device__ __forceinline__ void eight_point(const double* __restrict__ b1, const double* __restrict__ b2, const uint32_t (&s)[8], int lane, double (&E)[9]) {
  const int r0 = lane < 36 ? lane / 9 : 3, c = lane < 36 ? lane - 9 * (lane / 9) : 8;
  const int ci = c / 3, cj = c - 3 * ci;
  auto element = [&](int r) {
    uint32_t si = s[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) si = (r == k) ? s[k] : si;
    return b2[3 * (size_t)si + ci] * b1[3 * (size_t)si + cj];
  };
  null_vector_8x9(element(r0), element(r0 + 4), lane, E);
}

// ThreePointsRelativePose (multiview/solver_essential_three_point.cpp:31-79; ThreePointSolver::Solve, solver_essential_kernel.cpp:49-56): the two
// orthographic essential matrices [0 0 a; 0 0 b; c d e] of three correspondences in closed form. Only + - x / sqrt in the reference's
// evaluation order, every operation rounded on its own: the models are the reference's bit for bit (a negative radicand gives NaN
// models in both, which no residual accepts). Every lane computes the same thing.
__device__ __forceinline__ void three_point_ortho(const double2* __restrict__ x1, const double2* __restrict__ x2, const uint32_t (&s)[7], double (&E1)[9],
                                                  double (&E2)[9]) {
  const double2 p0 = x1[s[0]], p1 = x1[s[1]], p2 = x1[s[2]], q0 = x2[s[0]], q1 = x2[s[1]], q2 = x2[s[2]];
  const double u1x = add_rn(p1.x, -p0.x), u1y = add_rn(p1.y, -p0.y), v1x = add_rn(p2.x, -p0.x), v1y = add_rn(p2.y, -p0.y);
  const double u2x = add_rn(q1.x, -q0.x), u2y = add_rn(q1.y, -q0.y), v2x = add_rn(q2.x, -q0.x), v2y = add_rn(q2.y, -q0.y);
  const double denom = add_rn(mul_rn(u1x, v1y), -mul_rn(u1y, v1x));
  const double ac = add_rn(mul_rn(u1y, v2x), -mul_rn(u2x, v1y)) / denom;
  const double ad = add_rn(mul_rn(u1y, v2y), -mul_rn(u2y, v1y)) / denom;
  const double bc = add_rn(mul_rn(u2x, v1x), -mul_rn(u1x, v2x)) / denom;
  const double bd = add_rn(mul_rn(u2y, v1x), -mul_rn(u1x, v2y)) / denom;
  const double ac2 = mul_rn(ac, ac), bc2 = mul_rn(bc, bc);
  const double g2 = add_rn(add_rn(add_rn(-ac2, mul_rn(ad, ad)), -bc2), mul_rn(bd, bd));
  const double g1 = add_rn(mul_rn(mul_rn(2.0, ac), ad), mul_rn(mul_rn(2.0, bc), bd));
  const double g0 = add_rn(add_rn(ac2, bc2), -1.0);
  const double h4 = add_rn(mul_rn(g1, g1), mul_rn(g2, g2));
  const double h2 = add_rn(mul_rn(-g1, g1), mul_rn(mul_rn(2.0, g0), g2));
  const double h0 = mul_rn(g0, g0);
  const double rdisc = sqrt(add_rn(mul_rn(h2, h2), -mul_rn(mul_rn(4.0, h4), h0)));
  auto essential = [&](double root, double (&E)[9]) {
    const double sd = sqrt(mul_rn(-root / h4, 0.5));   // (/ 2.0 is exact; sqrt and / round to nearest on the device: tools/ortho_probe.hip)
    const double num = add_rn(add_rn(add_rn(mul_rn(mul_rn(g2, sd), sd), ac2), bc2), -1.0);
    const double den = add_rn(mul_rn(mul_rn(mul_rn(2.0, ac), ad), sd), mul_rn(mul_rn(mul_rn(2.0, bc), bd), sd));
    const double sc = -num / den;
    const double sa = add_rn(mul_rn(ac, sc), mul_rn(ad, sd));
    const double sb = add_rn(mul_rn(bc, sc), mul_rn(bd, sd));
    const double se = add_rn(add_rn(add_rn(mul_rn(-sa, p0.x), -mul_rn(sb, p0.y)), -mul_rn(sc, q0.x)), -mul_rn(sd, q0.y));
    E[0] = 0.0; E[1] = 0.0; E[2] = sa; E[3] = 0.0; E[4] = 0.0; E[5] = sb; E[6] = sc; E[7] = sd; E[8] = se;
  };
  essential(add_rn(h2, rdisc), E1);
  essential(add_rn(h2, -rdisc), E2);
}

// ThreePointUprightRelativePoseSolver::Solve (multiview/solver_essential_three_point.cpp:84-113): the null vector n of the 3 x 4
// system with rows [a.x b.y, -a.z b.y, -b.x a.y, -b.z a.y] (a = bearing in the first view, b = in the second) through its four 3 x 3
// minors (the reference: eigenvector of A^T A of smallest eigenvalue - the same line), E = [0 n2 0; -n0 0 n1; 0 n3 0]. Every lane
// computes the same thing.
__device__ __forceinline__ void three_point_upright(const double* __restrict__ b1, const double* __restrict__ b2, const uint32_t (&s)[7], int lane, double (&E)[9]) {
  (void)lane;
  double A[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const uint32_t si = s[i];
    const double ax = b1[3 * (size_t)si], ay = b1[3 * (size_t)si + 1], az = b1[3 * (size_t)si + 2];
    const double bx = b2[3 * (size_t)si], by = b2[3 * (size_t)si + 1], bz = b2[3 * (size_t)si + 2];
    A[i][0] = ax * by; A[i][1] = -az * by; A[i][2] = -bx * ay; A[i][3] = -bz * ay;
  }
  auto det3 = [&](int c0, int c1, int c2) {
    return A[0][c0] * (A[1][c1] * A[2][c2] - A[1][c2] * A[2][c1]) - A[0][c1] * (A[1][c0] * A[2][c2] - A[1][c2] * A[2][c0]) +
           A[0][c2] * (A[1][c0] * A[2][c1] - A[1][c1] * A[2][c0]);
  };
  const double n0 = det3(1, 2, 3), n1 = -det3(0, 2, 3), n2 = det3(0, 1, 3), n3 = -det3(0, 1, 2);
#pragma unroll
  for (int u = 0; u < 9; ++u) E[u] = 0.0;
  E[1] = n2; E[3] = -n0; E[5] = n1; E[7] = n3;
}

// SevenPointSolver::Solve on the sample s[0..6] (wave-uniform): the two null vectors F1, F2 of the 7 x 9 system and the real
// roots of det(F1 + alpha F2) = 0; model t is F1 + roots[t] F2. Every lane gets everything.
__device__ __forceinline__ int seven_point(const double2* __restrict__ x1, const double2* __restrict__ x2, const uint32_t (&s)[7], int lane,
                                           double (&F1)[9], double (&F2)[9], double (&roots)[3]) {
  // lane = 9 r + c holds A[r][c] = x2h[c / 3] * x1h[c % 3] of sample point r (EncodeEpipolarEquation, solver_fundamental_kernel.hpp:83-93)
  const int r = lane < 63 ? lane / 9 : 6, c = lane < 63 ? lane - 9 * (lane / 9) : 8;
  uint32_t si = s[0];
#pragma unroll
  for (int k = 1; k < 7; ++k) si = (r == k) ? s[k] : si;
  const double2 p1 = x1[si], p2 = x2[si];
  const int ci = c / 3, cj = c - 3 * ci;
  const double h2 = ci == 0 ? p2.x : ci == 1 ? p2.y : 1.0, h1 = cj == 0 ? p1.x : cj == 1 ? p1.y : 1.0;
  double a = h2 * h1;
  // Gauss-Jordan elimination with complete pivoting; the two columns without a pivot span the null space. The pivot is the
  // element of largest magnitude as a float (a choice within 2^-18 of the largest is as good a pivot), lowest lane on ties:
  // key = float bits with the low 6 bits replaced by 63 - lane.
  uint32_t row_used = 0, col_used = 0;
  int prow[7], pcol[7], n_piv = 0;
  double ipivs[7];   // 1 / pivot of every step (wave-uniform): the pivot element is not touched again, the null vectors below divide by it
#pragma unroll
  for (int step = 0; step < 7; ++step) {
    const bool cand = lane < 63 && !((row_used >> r) & 1u) && !((col_used >> c) & 1u);
    const float mag = (float)fabs(a);
    const uint32_t key = (cand && mag > 0.f && mag == mag) ? ((__float_as_uint(mag) & ~63u) | (uint32_t)(63 - lane)) : 0u;
    const uint32_t best = wave_max_u32(key);
    if (best == 0u) break;   // rank deficient sample (wave-uniform): fewer pivots, the first two free columns are used
    const int who = 63 - (int)(best & 63u);
    const int pr = who / 9, pc = who - 9 * (who / 9);
    const double ipiv = 1.0 / lane_value_f64(a, who);
    const double rowv = shfl_f64(a, 9 * pr + c);   // pivot row, my column
    const double colv = shfl_f64(a, 9 * r + pc);   // my row, pivot column
    if (r != pr) a -= (colv * ipiv) * rowv;
    row_used |= 1u << pr; col_used |= 1u << pc;
    prow[step] = pr; pcol[step] = pc; ipivs[step] = ipiv;
    n_piv = step + 1;
  }
  int f1 = 0;
  while ((col_used >> f1) & 1u) ++f1;
  int f2 = f1 + 1;
  while ((col_used >> f2) & 1u) ++f2;
  // lane u < 9 gets component u of the two null vectors: 1 at the free column, -A[prow][free] / A[prow][pcol] at a pivot column
  double F1u = lane == f1 ? 1.0 : 0.0, F2u = lane == f2 ? 1.0 : 0.0;
#pragma unroll
  for (int step = 0; step < 7; ++step) {
    if (step < n_piv) {   // wave-uniform
      const double iden = ipivs[step];   // (= 1 / A[prow][pcol]: later steps only change the OTHER columns of a pivot row)
      const double n1 = lane_value_f64(a, 9 * prow[step] + f1), n2 = lane_value_f64(a, 9 * prow[step] + f2);
      if (lane == pcol[step]) { F1u = -n1 * iden; F2u = -n2 * iden; }
    }
  }
#pragma unroll
  for (int u = 0; u < 9; ++u) { F1[u] = lane_value_f64(F1u, u); F2[u] = lane_value_f64(F2u, u); }
  // det(F1 + alpha F2) = 0 (solver_fundamental_kernel.cpp:62-86), coefficients in ascending powers of alpha
  const double a_ = F1[0], j = F2[0], b = F1[1], k = F2[1], c_ = F1[2], l = F2[2], d = F1[3], m = F2[3], e = F1[4], n = F2[4],
               f = F1[5], o = F2[5], g = F1[6], p = F2[6], h = F1[7], q = F2[7], i = F1[8], rr = F2[8];
  const double P[4] = {
    a_*e*i + b*f*g + c_*d*h - a_*f*h - b*d*i - c_*e*g,
    a_*e*rr + a_*i*n + b*f*p + b*g*o + c_*d*q + c_*h*m + d*h*l + e*i*j + f*g*k -
    a_*f*q - a_*h*o - b*d*rr - b*i*m - c_*e*p - c_*g*n - d*i*k - e*g*l - f*h*j,
    a_*n*rr + b*o*p + c_*m*q + d*l*q + e*j*rr + f*k*p + g*k*o + h*l*m + i*j*n -
    a_*o*q - b*m*rr - c_*n*p - d*k*rr - e*l*p - f*j*q - g*l*n - h*j*o - i*k*m,
    j*n*rr + k*o*p + l*m*q - j*o*q - k*m*rr - l*n*p};
  roots[0] = roots[1] = roots[2] = 0.0;
  if (P[0] == 0.0) return 0;   // poly.h:88-91
  return solve_cubic(P[2] / P[3], P[1] / P[3], P[0] / P[3], roots);
}

// MVGX_GEO_STAMPS (measurement build): shader clocks of lane 0 between the stages of an a-contrario iteration, summed over all waves:
// 0 sampling | 1 minimal solver | 2 residuals + histogram | 3 NFA over the bins | 4 inlier count of a better model | 5 loop control, pool
#ifdef MVGX_GEO_STAMPS
__device__ unsigned long long g_geo_stamps[8];
#define GEO_STAMP(i) do { const long long t_now_ = __builtin_amdgcn_s_memtime(); geo_acc_[i] += (unsigned long long)(t_now_ - t_geo_); t_geo_ = t_now_; } while (0)   // (per wave, in registers: one atomic per stage and wave at the end)
#else
#define GEO_STAMP(i) do { } while (0)
#endif
// float logcombi(k, n) (robust_estimator_ACRansac.hpp:58-72) on the shared table of log10 values
__device__ __forceinline__ float logcombi(uint32_t k, uint32_t n, const float* __restrict__ l10) {
  if (k >= n) return 0.f;
  if (n - k < k) k = n - k;
  float r = 0.f;
  for (uint32_t i = 1; i <= k; ++i) r += l10[n - i + 1] - l10[i];
  return r;
}

// number of correspondences whose residual under F is at most thr (all lanes get the sum): ballots, no shuffles
template <int MODEL>
__device__ __forceinline__ uint32_t count_within(const double (&F)[9], const double2* __restrict__ x1, const double2* __restrict__ x2,
                                                 const double* __restrict__ b1, const double* __restrict__ b2, uint32_t n, double thr, int lane) {
  uint32_t cnt = 0;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t i = base + lane;
    const bool in = i < n && model_error<MODEL>(F, load_point<MODEL>(x1, b1, i < n ? i : 0), load_point<MODEL>(x2, b2, i < n ? i : 0)) <= thr;
    cnt += (uint32_t)__popcll(__ballot(in));
  }
  return cnt;
}

// ApplyTransformationToPoints with the preconditioner of the image size (conditioning.cpp:44-67): x' = s x + t per coordinate, one
// product and one sum each, no contraction - the values the reference's Eigen expression produces
__global__ __launch_bounds__(256) void geofilter_normalize_kernel(const GeoPair* __restrict__ pairs, const double* __restrict__ norm /* 6 per pair */,
                                                                  uint32_t n_pairs, const double2* __restrict__ xI, const double2* __restrict__ xJ,
                                                                  double2* __restrict__ x1n, double2* __restrict__ x2n) {
  const uint32_t p = blockIdx.x;
  if (p >= n_pairs) return;
  const uint64_t lo = pairs[p].start;
  const uint32_t n = pairs[p].n;
  const double* __restrict__ t = norm + 6 * (size_t)p;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const double2 a = xI[lo + i], b = xJ[lo + i];
    x1n[lo + i] = make_double2(add_rn(mul_rn(t[0], a.x), t[1]), add_rn(mul_rn(t[0], a.y), t[2]));
    x2n[lo + i] = make_double2(add_rn(mul_rn(t[3], b.x), t[4]), add_rn(mul_rn(t[3], b.y), t[5]));
  }
}

// The same with MatchesPairToMat (Geometric_Filter_utils.hpp:56-64) in front: the correspondences of pair (I, J) are index pairs
// (i, j) into the feature positions of the two images
__global__ __launch_bounds__(256) void geofilter_normalize_indexed_kernel(const GeoPair* __restrict__ pairs, const double* __restrict__ norm, uint32_t n_pairs,
                                                                          const double2* __restrict__ feat_xy, const uint64_t* __restrict__ feat_start,
                                                                          const uint2* __restrict__ pair_images, const uint2* __restrict__ ij,
                                                                          double2* __restrict__ x1n, double2* __restrict__ x2n) {
  const uint32_t p = blockIdx.x;
  if (p >= n_pairs) return;
  const uint64_t lo = pairs[p].start;
  const uint32_t n = pairs[p].n;
  const double* __restrict__ t = norm + 6 * (size_t)p;
  const uint2 im = pair_images[p];
  const double2* __restrict__ fI = feat_xy + feat_start[im.x];
  const double2* __restrict__ fJ = feat_xy + feat_start[im.y];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint2 m = ij[lo + i];
    const double2 a = fI[m.x], b = fJ[m.y];
    x1n[lo + i] = make_double2(add_rn(mul_rn(t[0], a.x), t[1]), add_rn(mul_rn(t[0], a.y), t[2]));
    x2n[lo + i] = make_double2(add_rn(mul_rn(t[3], b.x), t[4]), add_rn(mul_rn(t[3], b.y), t[5]));
  }
}

// essential model, indexed form: the bearing vectors of the matches gathered from the per-feature table
__global__ __launch_bounds__(256) void geofilter_gather_bearings_kernel(const GeoPair* __restrict__ pairs, uint32_t n_pairs, const double* __restrict__ feat_bearing,
                                                                        const uint64_t* __restrict__ feat_start, const uint2* __restrict__ pair_images,
                                                                        const uint2* __restrict__ ij, double* __restrict__ b1, double* __restrict__ b2) {
  const uint32_t p = blockIdx.x;
  if (p >= n_pairs) return;
  const uint64_t lo = pairs[p].start;
  const uint32_t n = pairs[p].n;
  const uint2 im = pair_images[p];
  const double* __restrict__ fI = feat_bearing + 3 * feat_start[im.x];
  const double* __restrict__ fJ = feat_bearing + 3 * feat_start[im.y];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint2 m = ij[lo + i];
#pragma unroll
    for (int k = 0; k < 3; ++k) { b1[3 * (lo + i) + k] = fI[3 * (size_t)m.x + k]; b2[3 * (lo + i) + k] = fJ[3 * (size_t)m.y + k]; }
  }
}

constexpr int kWaveScratch = 64;   // words behind a wave's tables: histogram (20) | the model of the current inlier list (18 words = 9 doubles at 8-byte alignment + 2)
// the essential model adds the five-point solver's workspace, its up to ten essential matrices and their fundamental matrices
// (round 5: four samples are solved side by side - four workspaces, four sets of essential matrices, the fundamental matrices of the one being evaluated)
constexpr int kAhead = 4;
constexpr int kEssentialScratch = 2 * (kAhead * five_point::kScratch + kAhead * 90 + 90);
// fundamental / homography models: the models of the four samples solved together - per row F1 | F2 | roots | their number (22 doubles)
constexpr int kRowModel = 22;
// (The seven-point solver of the fundamental model was built in this form too and is not kept: its elimination is 40 % of an iteration, not
// 75 - 85 %, and what four side by side save went into the exchange of the rows' models and 60 more spilled registers: 55 ms against 48 on
// 20 000 pairs, profiles/round5_geofilter_f_h_samples_ahead_ab_call_r5_30.txt.)
template <int MODEL> constexpr bool model_solves_ahead() { return MODEL == kModelE || MODEL == kModelH; }
template <int MODEL> constexpr int wave_scratch_words() {
  return kWaveScratch + (MODEL == kModelE ? kEssentialScratch : model_solves_ahead<MODEL>() ? 2 * kAhead * kRowModel : 0);
}

// kGlobalTables: the sampling pool and the two log-combinatorial tables of a wave (3 x n words) live in a global scratch block
// instead of LDS - the class of pairs with more correspondences than a workgroup's LDS holds (one wave per workgroup; the generator,
// the histogram and the model stay in LDS)
#ifndef MVGX_GEO_WGS
#define MVGX_GEO_WGS 3     // ... the other instantiations
#endif
#ifndef MVGX_GEO_E_WGS
#define MVGX_GEO_E_WGS 2   // workgroups per CU the essential instantiation is compiled for (1: twice the registers, half the waves)
#endif
// kAheadForm (essential and homography models): the instantiation that draws its samples ahead and solves four side by side; the other one
// holds the one-sample solver (`ahead` = 1: the form the equality tests compare with). Two kernels, not one with a run-time branch: with both
// solvers in one kernel the essential instantiation spilled 246 registers instead of 164 and ran 19 % slower (call r5_57).
template <int WAVES, bool kGlobalTables = false, int MODEL = kModelF, bool kAheadForm = false>
__global__ __launch_bounds__(64 * WAVES, MODEL == kModelE ? MVGX_GEO_E_WGS : MVGX_GEO_WGS) void geofilter_f_acransac_kernel(const GeoPair* __restrict__ pairs, const uint32_t* __restrict__ order,
                                                                          uint32_t n_work, uint32_t n_cap, const double2* __restrict__ x1n,
                                                                          const double2* __restrict__ x2n, const float* __restrict__ l10,
                                                                          const uint32_t* __restrict__ mt_init, uint32_t max_iterations,
                                                                          GeoResult* __restrict__ results, uint8_t* __restrict__ mask,
                                                                          uint32_t* __restrict__ table_scratch = nullptr,
                                                                          const double* __restrict__ bear1 = nullptr, const double* __restrict__ bear2 = nullptr,
                                                                          uint32_t ahead = 1) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u32[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t w = blockIdx.x * WAVES + wave;
  if (w >= n_work) return;   // wave-uniform; no workgroup barrier below
  const long long t_start = __builtin_amdgcn_s_memtime();
  uint32_t n_iter_run = 0, n_models_run = 0;
  constexpr int kMin = model_min_samples<MODEL>();
  const uint32_t pidx = order[w];
  const GeoPair& P = pairs[pidx];   // (read through the scalar data path: wave-uniform)
  const uint32_t n = P.n;
  // LDS form: the class's size for every pair; global form: the pair's own (its tables sit at 3 x (start + 4 x pair) words of the
  // scratch block - start = the pair's first correspondence: no two pairs overlap, whatever their sizes)
  const uint32_t cap1 = ((kGlobalTables ? n : n_cap) + 2) & ~1u;   // even: the doubles behind stay 8-byte aligned
  const uint32_t per_wave = kMtN + (kGlobalTables ? 0u : 3 * cap1) + wave_scratch_words<MODEL>();   // words: generator | pool | logc_n | logc_k | scratch
  uint32_t* const mt = lds_u32 + (size_t)wave * per_wave;
  uint32_t* const pool = kGlobalTables ? table_scratch + 3 * ((size_t)P.start + 4 * (size_t)pidx) : mt + kMtN;
  float* const logc_n = reinterpret_cast<float*>(pool + cap1);
  float* const logc_k = logc_n + cap1;
  uint32_t* const hist = kGlobalTables ? mt + kMtN : reinterpret_cast<uint32_t*>(logc_k + cap1);
  double* const inlF_lds = reinterpret_cast<double*>(hist + 24);   // the model behind the current inlier list / pool (rarely touched: kept out of registers)
  double* const e_scr = reinterpret_cast<double*>(hist + kWaveScratch);   // essential model: solver workspace | Es[10][9] | Fs[10][9]
  double* const e_Es4 = e_scr + kAhead * five_point::kScratch;   // the essential matrices of the four samples solved together
  double* const e_Fs = e_Es4 + kAhead * 90;
  double* e_Es = e_Es4;                                           // ... of the sample under evaluation
  const double max_threshold = P.max_threshold, bins_by_interval = P.bins_by_interval, loge0 = P.loge0;
  const double2* __restrict__ x1 = x1n + P.start;
  const double2* __restrict__ x2 = x2n + P.start;
  constexpr bool kBearings = MODEL == kModelE || model_is_angular<MODEL>();
  const double* __restrict__ bv1 = kBearings ? bear1 + 3 * P.start : nullptr;
  const double* __restrict__ bv2 = kBearings ? bear2 + 3 * P.start : nullptr;
  // lane b < 20 keeps the constants of histogram bin b
  const double my_bin_value = lane < kBins ? P.bin_value[lane] : 0.0, my_logalpha = lane < kBins ? P.logalpha_bin[lane] : 0.0;
  // ---- set-up: generator (std::mt19937(5489) before its first twist), pool = 0..n-1, tables ----
  for (int i = lane; i < kMtN; i += 64) mt[i] = mt_init[i];
  for (uint32_t i = lane; i < n; i += 64) pool[i] = i;
  for (uint32_t k = lane; k <= n; k += 64) { logc_n[k] = logcombi(k, n, l10); logc_k[k] = logcombi(kMin, k, l10); }
  if (lane < 9) inlF_lds[lane] = 0.0;
  wave_sync();
  int mt_idx = kMtN;
  uint32_t pool_size = n;
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  double minNFA = inf, errorMax = inf;
  double bestFu = 0.0;   // lane u < 9: component u of the best model
  double inl_thr = 0.0;
  uint32_t inl_count = 0, have_model = 0;
  int nIterReserve = (int)(max_iterations / 10);
  unsigned nIter = max_iterations - (unsigned)nIterReserve;
  bool ac_mode = false;
  constexpr int kS = model_sample_slots<MODEL>();
  uint32_t s[kS];
#pragma unroll
  for (int k = 0; k < kS; ++k) s[k] = 0;
#ifdef MVGX_GEO_STAMPS
  unsigned long long geo_acc_[6] = {0, 0, 0, 0, 0, 0};
  long long t_geo_ = __builtin_amdgcn_s_memtime();
#endif
  // One sample from the generator and the pool as they stand (rand_sampling.hpp), into s. may_twist == false - a sample drawn AHEAD of
  // the iteration that will use it (essential model, below): a draw that would twist the generator's state is revoked - index and pool as
  // they were, s undefined - and false comes back.
  // The kMin draws of a sample side by side: lane i < kMin takes word idx + i of the generator's state, tempers it and scales it to its
  // range [lo_i, hi] exactly as uniform_int_distribution does (product >> 32); out[i] = the value of lane i. Only when every draw is
  // accepted at once - the distribution rejects a word with probability range / 2^32 - and the state holds kMin more words; else false
  // and nothing has changed (the caller then draws one after the other: twists, rejections and all).
  auto draw_together = [&](uint32_t lo_step, uint32_t hi, uint32_t (&out)[kMin]) -> bool {
    if (mt_idx + kMin > kMtN) return false;
    const uint32_t i = (uint32_t)lane < (uint32_t)kMin ? (uint32_t)lane : 0u;
    const uint32_t lo = lo_step * i;                 // (a-contrario mode: element i is exchanged with one of [i, last]; warm-up: all from [0, n - 1])
    uint32_t y = mt[mt_idx + (int)i];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    const uint32_t range = hi - lo + 1;              // (never 2^32 here: hi < n)
    const uint64_t product = (uint64_t)y * range;
    if (__ballot((uint32_t)lane < (uint32_t)kMin && (uint32_t)product < range)) return false;   // (a word the distribution might reject: the exact path decides)
    const uint32_t v = (uint32_t)(product >> 32) + lo;
#pragma unroll
    for (int k = 0; k < kMin; ++k) out[k] = (uint32_t)__builtin_amdgcn_readlane((int)v, k);
    mt_idx += kMin;
    return true;
  };
  auto draw_sample = [&](bool may_twist) -> bool {
    const int idx0 = mt_idx;
    if (ac_mode) {
      if (pool_size >= (uint32_t)kMin) {   // else UniformSample returns false and vec_sample keeps its values
        // element i is exchanged with element jx_i in [i, last]: the draws do not depend on the pool, so they come first (a revoked
        // sample has touched nothing but the index); one lane then makes the exchanges in order (its LDS operations complete in
        // order: no rendezvous between them), and element i is final once exchange i is done (later ones touch elements >= their own i)
        const uint32_t last = pool_size - 1;
        uint32_t jx[kMin];
        if (!draw_together(1u, last, jx)) {
          bool ok = true;
#pragma unroll
          for (uint32_t i = 0; i < (uint32_t)kMin; ++i) {
            jx[i] = 0;
            ok = ok && uniform_try_u32(mt, mt_idx, lane, i, last, may_twist, jx[i]);
          }
          if (!ok) { mt_idx = idx0; return false; }
        }
        if (lane == 0) {
#pragma unroll
          for (uint32_t i = 0; i < (uint32_t)kMin; ++i) {
            const uint32_t vi = pool[i], vj = pool[jx[i]];
            pool[i] = vj; pool[jx[i]] = vi;
          }
        }
        wave_sync();
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)kMin; ++i) s[i] = pool[i];
        wave_sync();   // (read before the next sample's exchanges)
      }
    } else {
      // warm-up: kMin distinct indices of [0, n - 1], a repeated one drawn again - side by side when the first kMin draws are distinct
      // (which they are 9 times in 10 at 250 correspondences), one after the other otherwise
      uint32_t c[kMin];
      if (draw_together(0u, n - 1, c)) {
        bool distinct = true;
#pragma unroll
        for (int a = 1; a < kMin; ++a)
#pragma unroll
          for (int b = 0; b < a; ++b) distinct = distinct && c[a] != c[b];
        if (distinct) {
#pragma unroll
          for (int k = 0; k < kMin; ++k) s[k] = c[k];
          return true;
        }
        mt_idx = idx0;   // (the exact path from the same words)
      }
      int got = 0;
      while (got < kMin) {
        uint32_t cand = 0;
        if (!uniform_try_u32(mt, mt_idx, lane, 0, n - 1, may_twist, cand)) { mt_idx = idx0; return false; }
        bool found = false;
#pragma unroll
        for (int k = 0; k < kS; ++k) found = found || (k < got && s[k] == cand);
        if (!found) {
#pragma unroll
          for (int k = 0; k < kS; ++k) if (k == got) s[k] = cand;
          ++got;
        }
      }
    }
    return true;
  };
  // Essential model, SAMPLES AHEAD (round 5). The five-point solver is 85 % of an iteration and the dependent instruction stream of ten
  // lanes. What the generator and the pool hand out does not depend on the models found - only three events re-seat the sampling: the
  // switch from the max-consensus warm-up to the a-contrario mode, a better model (the pool becomes its inliers) and the end of the
  // loop. So up to four samples are drawn in a row (each from the state the one before left: the reference's sequence), solved side by
  // side - one per 16-lane row, five_point::solve4 - and evaluated in order; after an event the samples not yet used are void: the
  // generator's index goes back to where the last used sample left it (a sample drawn ahead never twists the state, so the index is
  // the whole state; the pool is either untouched - warm-up - or rebuilt by the event) and the next batch draws again. Results are those
  // of one sample per iteration, bit for bit (`ahead` = 1 is that form: tests/test_geofilter_e.py compares the two).
  constexpr bool x4 = kAheadForm && model_solves_ahead<MODEL>();
  double* const row_models = e_scr;   // (fundamental / homography models: the same words of the wave's scratch)
  unsigned iter = 0;
  while (iter < nIter && iter < max_iterations) {
    int kb = 1;
    [[maybe_unused]] uint32_t S[kAhead][kMin];
    [[maybe_unused]] int idx_after[kAhead] = {0, 0, 0, 0};
    [[maybe_unused]] int nm_rows = 0;
    if constexpr (model_solves_ahead<MODEL>()) {
      if (x4) {
        const unsigned left = (nIter < max_iterations ? nIter : max_iterations) - iter;
        const int kb_max = (int)(left < ahead ? left : ahead);
        kb = 0;
#pragma unroll
        for (int j = 0; j < kAhead; ++j) {
          if (j < kb_max && kb == j) {   // (kb == j: no draw of this batch was revoked)
            if (draw_sample(j == 0)) {
#pragma unroll
              for (int i = 0; i < kMin; ++i) S[j][i] = s[i];
              idx_after[j] = mt_idx;
              kb = j + 1;
            }
          }
        }
        GEO_STAMP(0);
        uint32_t mine[kMin];   // the sample of this lane's row (rows beyond the batch repeat its last sample; their results are not read)
        const int row = (lane >> 4) < kb ? (lane >> 4) : kb - 1;
#pragma unroll
        for (int i = 0; i < kMin; ++i) mine[i] = row == 0 ? S[0][i] : row == 1 ? S[1][i] : row == 2 ? S[2][i] : S[3][i];
        if constexpr (MODEL == kModelE) nm_rows = five_point::solve4(bv1, bv2, mine, lane, e_scr, e_Es4);
        else {
          // the row's models go to the wave's scratch: the evaluation below wants them in every lane, sample after sample
          double R1[9], R2[9], rr[3] = {0.0, 0.0, 0.0};
          int rn = 1;
          four_point_x4(x1, x2, mine, lane, R1);
#pragma unroll
          for (int u = 0; u < 9; ++u) R2[u] = 0.0;
          wave_sync();   // (the models of the batch before have been read)
          if ((lane & 15) == 0) {
            double* const dst = row_models + kRowModel * (lane >> 4);
#pragma unroll
            for (int u = 0; u < 9; ++u) { dst[u] = R1[u]; dst[9 + u] = R2[u]; }
            dst[18] = rr[0]; dst[19] = rr[1]; dst[20] = rr[2]; dst[21] = (double)rn;
          }
          wave_sync();
        }
        GEO_STAMP(1);
      }
    }
#pragma unroll 1
    for (int j = 0; j < kb; ++j) {
    const bool ac_at_start = ac_mode;
    // ---- sample ----
    if (!x4) (void)draw_sample(true);
    if constexpr (model_solves_ahead<MODEL>()) {
      if (x4) {
#pragma unroll
        for (int i = 0; i < kMin; ++i) s[i] = j == 0 ? S[0][i] : j == 1 ? S[1][i] : j == 2 ? S[2][i] : S[3][i];
      }
    }
    GEO_STAMP(0);
    // ---- fit, evaluate ----
    double F1[9], F2[9], roots[3] = {0.0, 0.0, 0.0};
    int nm = 1;
    if constexpr (MODEL == kModelH) {
      if (x4) {
#pragma unroll
        for (int u = 0; u < 9; ++u) F1[u] = row_models[kRowModel * j + u];
      } else four_point(x1, x2, s, lane, F1);   // (one model per sample: MAX_MODELS = 1)
#pragma unroll
      for (int u = 0; u < 9; ++u) F2[u] = 0.0;
    } else if constexpr (MODEL == kModelE) {
      // FivePointSolver on the sample's bearing vectors (up to ten essential matrices, LDS), then F = K2^-T E K1^-1 per model for
      // the pixel residuals (ACKernelAdaptorEssential::Errors; products and sums rounded one by one like the reference's 3 x 3 products)
      if (x4) { nm = __builtin_amdgcn_readlane(nm_rows, 16 * j); e_Es = e_Es4 + 90 * j; }
      else { nm = five_point::solve(bv1, bv2, s, lane, e_scr, e_Es4); e_Es = e_Es4; }
      for (int e = lane; e < 9 * nm; e += 64) {
        const int mi = e / 9, u = e - 9 * mi, i = u / 3, j = u - 3 * i;
        const double* __restrict__ E = e_Es + 9 * mi;
        double t3[3];   // row i of K2^-T E
#pragma unroll
        for (int c = 0; c < 3; ++c) t3[c] = add_rn(add_rn(mul_rn(P.k2it[3 * i], E[c]), mul_rn(P.k2it[3 * i + 1], E[3 + c])), mul_rn(P.k2it[3 * i + 2], E[6 + c]));
        e_Fs[9 * mi + u] = add_rn(add_rn(mul_rn(t3[0], P.k1i[j]), mul_rn(t3[1], P.k1i[3 + j])), mul_rn(t3[2], P.k1i[6 + j]));
      }
      wave_sync();
#pragma unroll
      for (int u = 0; u < 9; ++u) { F1[u] = 0.0; F2[u] = 0.0; }
    } else if constexpr (MODEL == kModelEA8) {
      eight_point(bv1, bv2, s, lane, F1);
#pragma unroll
      for (int u = 0; u < 9; ++u) F2[u] = 0.0;
    } else if constexpr (MODEL == kModelEU3) {
      three_point_upright(bv1, bv2, s, lane, F1);
#pragma unroll
      for (int u = 0; u < 9; ++u) F2[u] = 0.0;
    } else if constexpr (MODEL == kModelEO) {
      three_point_ortho(x1, x2, s, F1, F2);   // (MAX_MODELS = 2: model 0 = F1, model 1 = F2)
      nm = 2;
    } else {
      nm = seven_point(x1, x2, s, lane, F1, F2, roots);
    }
    GEO_STAMP(1);
    bool better = false;
    ++n_iter_run; n_models_run += (uint32_t)nm;
    for (int mi = 0; mi < nm; ++mi) {
      const double root = mi == 0 ? roots[0] : mi == 1 ? roots[1] : roots[2];
      double F[9];
#pragma unroll
      for (int u = 0; u < 9; ++u)
        F[u] = (MODEL == kModelH || model_is_angular<MODEL>()) ? F1[u] : MODEL == kModelEO ? (mi == 0 ? F1[u] : F2[u]) : MODEL == kModelE ? e_Fs[9 * mi + u]
                                                                                                                                                  : F1[u] + root * F2[u];
      if (lane < kBins) hist[lane] = 0;
      wave_sync();
      uint32_t n_le = 0;
      for (uint32_t base = 0; base < n; base += 256) {   // four rounds of loads in flight
        typename PointOf<MODEL>::type a1[4], a2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t i = base + 64 * q + lane;
          a1[q] = load_point<MODEL>(x1, bv1, i < n ? i : 0); a2[q] = load_point<MODEL>(x2, bv2, i < n ? i : 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t i = base + 64 * q + lane;
          const double e = model_error<MODEL>(F, a1[q], a2[q]);
          const bool in = i < n && e <= max_threshold;
          if (!ac_mode) n_le += (uint32_t)__popcll(__ballot(in));
          if (i < n && !(e < 0.0)) {   // Histogram::Add (histogram.hpp:72-86); a NaN / huge value ends in no bin, like the cast to size_t on x86-64
            const double t = mul_rn(e, bins_by_interval);
            if (t >= 0.0 && t < (double)kBins) atomicAdd(&hist[(int)t], 1u);
          }
        }
      }
      if (!ac_mode && (double)n_le > 2.5 * kMin) ac_mode = true;   // MAX-CONSENSUS warm-up (:404-414)
      wave_sync();
      GEO_STAMP(2);
      if (ac_mode) {   // ComputeNFA_and_inliers, quantified form (:196-262): lane b evaluates bin b, the first bin of minimal NFA wins
        // cumulated histogram: lane b reads its bin, an inclusive scan inside the 16-lane row (DPP row_shr), lanes 16..19 add row 0's total
        uint32_t cum = lane < kBins ? hist[lane] : 0u;
        cum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cum, 0x111, 0xF, 0xF, true);
        cum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cum, 0x112, 0xF, 0xF, true);
        cum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cum, 0x114, 0xF, 0xF, true);
        cum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cum, 0x118, 0xF, 0xF, true);
        {
          const uint32_t row0 = (uint32_t)__builtin_amdgcn_readlane((int)cum, 15);
          if (lane >= 16) cum += row0;
        }
        double cur = inf;
        if (lane < kBins && cum > (uint32_t)kMin && my_bin_value > 1.1920928955078125e-07) {
          cur = add_rn(add_rn(add_rn(loge0, mul_rn(my_logalpha, (double)(cum - kMin))), (double)logc_n[cum]), (double)logc_k[cum]);
          if (!(cur < 0)) cur = inf;
        }
        // minimum over the 20 lanes, lowest bin on ties (the sequential scan keeps the first strict improvement): wave maxima of an
        // order-reversing key of the value's bits (high word, then low word among its holders), then the first lane that holds both
        double cb_nfa = inf, cb_thr = 0.0;
        {
          const bool cand = lane < kBins && cur < inf;   // (a candidate is a finite negative value)
          unsigned long long bits = (unsigned long long)__double_as_longlong(cur);
          bits = (bits >> 63) ? bits : ~(bits | 0x8000000000000000ull);   // negative doubles: larger bit pattern = smaller value; the others below all of them
          const uint32_t hi = cand ? (uint32_t)(bits >> 32) : 0u;
          const uint32_t best_hi = wave_max_u32(hi);
          if (best_hi != 0u) {   // (wave-uniform; 0: no candidate)
            const bool top = cand && hi == best_hi;
            const uint32_t lo = top ? (uint32_t)bits : 0u;
            const uint32_t best_lo = wave_max_u32(lo);
            const unsigned long long holders = __ballot(top && lo == best_lo);
            const int b = (int)__builtin_ctzll(holders);
            cb_nfa = lane_value_f64(cur, b);
            cb_thr = lane_value_f64(my_bin_value, b);
          }
        }
        GEO_STAMP(3);
        if (cb_nfa < minNFA) {   // the inlier list is rebuilt even if it then turns out too short (the reference's behaviour)
          const uint32_t cnt = count_within<MODEL>(F, x1, x2, bv1, bv2, n, cb_thr, lane);
          wave_sync();
          if (lane < 9) {
#pragma unroll
            for (int u = 0; u < 9; ++u) if (lane == u) inlF_lds[u] = F[u];
          }
          wave_sync();
          inl_thr = cb_thr; inl_count = cnt;
          if (cnt > (uint32_t)kMin) {
            better = true; minNFA = cb_nfa; errorMax = cb_thr; have_model = 1;
#pragma unroll
            for (int u = 0; u < 9; ++u) if (lane == u) bestFu = MODEL == kModelE ? e_Es[9 * mi + u] : F[u];
          }
        }
      }
    }
    GEO_STAMP(4);
    // ---- loop control (:445-474) ----
    bool redraw = ac_mode != ac_at_start;   // samples drawn ahead in the other mode are void
    if (!ac_mode && iter > (unsigned)(nIterReserve * 2)) nIter = 0;
    else if (ac_mode && ((better && minNFA < 0) || ((iter + 1) == nIter && nIterReserve > 0))) {
      if (inl_count == 0) { ++nIter; --nIterReserve; }
      else {
        // vec_index = vec_inliers: the correspondences within inl_thr of the list's model, in index order
        double F[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) F[u] = inlF_lds[u];
        uint32_t m = 0;
        for (uint32_t base = 0; base < n; base += 64) {
          const uint32_t i = base + lane;
          const bool in = i < n && model_error<MODEL>(F, load_point<MODEL>(x1, bv1, i < n ? i : 0), load_point<MODEL>(x2, bv2, i < n ? i : 0)) <= inl_thr;
          const unsigned long long bal = __ballot(in);
          if (in) pool[m + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = i;
          m += (uint32_t)__popcll(bal);
        }
        wave_sync();
        pool_size = m;
        if (nIterReserve) { nIter = iter + 1 + (unsigned)nIterReserve; nIterReserve = 0; }
        redraw = true;   // ... from the old pool as well
      }
    }
    GEO_STAMP(5);
    ++iter;
    if (!(iter < nIter && iter < max_iterations)) break;
    if constexpr (model_solves_ahead<MODEL>()) {
      if (x4 && redraw && j + 1 < kb) {   // the generator as the last sample used left it
        mt_idx = j == 0 ? idx_after[0] : j == 1 ? idx_after[1] : idx_after[2];
        break;
      }
    }
    }   // samples of the batch
  }
#ifdef MVGX_GEO_STAMPS
  if (lane == 0) { for (int k = 0; k < 6; ++k) atomicAdd(&g_geo_stamps[k], geo_acc_[k]); }
#endif
  if (minNFA >= 0) inl_count = 0;   // no meaningful model (:477-478)
  const bool good = (double)inl_count > kMin * 2.5;   // F_ACRobust.hpp:103, H_ACRobust.hpp:98
  {
    double F[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) F[u] = inlF_lds[u];
    for (uint32_t i = lane; i < n; i += 64) mask[P.start + i] = (good && model_error<MODEL>(F, load_point<MODEL>(x1, bv1, i), load_point<MODEL>(x2, bv2, i)) <= inl_thr) ? 1 : 0;
  }
  if (lane < 9) results[pidx].F[lane] = bestFu;
  if (lane == 0) {
    results[pidx].error_max = errorMax; results[pidx].min_nfa = minNFA; results[pidx].n_inliers = inl_count; results[pidx].have_model = have_model;
    results[pidx].n_iterations = n_iter_run; results[pidx].n_models = n_models_run;
    results[pidx].clocks = (uint64_t)(__builtin_amdgcn_s_memtime() - t_start);
  }
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { MVGX_HIP(mvgx::device_malloc(&p, std::max<size_t>(bytes, 16))); return MVGX_OK; }
};

template <int WAVES, int MODEL>
int launch_class(const GeoPair* d_pairs, const uint32_t* d_order, uint32_t n_work, uint32_t n_cap, const double2* x1, const double2* x2,
                 const float* l10, const uint32_t* mt_init, uint32_t max_it, GeoResult* res, uint8_t* mask, hipStream_t stream,
                 const double* b1 = nullptr, const double* b2 = nullptr, uint32_t ahead = 1) {
  if (!n_work) return MVGX_OK;
  const size_t lds = (size_t)WAVES * (kMtN + 3 * (size_t)((n_cap + 2) & ~1u) + wave_scratch_words<MODEL>()) * sizeof(uint32_t);
  constexpr bool kHasAheadForm = model_solves_ahead<MODEL>();   // (the other models have one instantiation)
  if (kHasAheadForm && ahead > 1) {
    MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&geofilter_f_acransac_kernel<WAVES, false, MODEL, kHasAheadForm>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((geofilter_f_acransac_kernel<WAVES, false, MODEL, kHasAheadForm>), dim3((n_work + WAVES - 1) / WAVES), dim3(64 * WAVES), lds, stream, d_pairs, d_order, n_work,
                       n_cap, x1, x2, l10, mt_init, max_it, res, mask, (uint32_t*)nullptr, b1, b2, ahead);
  } else {
    MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&geofilter_f_acransac_kernel<WAVES, false, MODEL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((geofilter_f_acransac_kernel<WAVES, false, MODEL, false>), dim3((n_work + WAVES - 1) / WAVES), dim3(64 * WAVES), lds, stream, d_pairs, d_order, n_work,
                       n_cap, x1, x2, l10, mt_init, max_it, res, mask, (uint32_t*)nullptr, b1, b2, ahead);
  }
  MVGX_HIP(hipGetLastError());
  return MVGX_OK;
}
// the class above the LDS classes: tables in `scratch` (n_work x 3 x ((n_cap + 2) & ~1) words), workgroups of four waves like the small classes
// (the LDS a wave needs no longer grows with its pair: the occupancy of the small classes at any size)
template <int MODEL>
int launch_class_global(const GeoPair* d_pairs, const uint32_t* d_order, uint32_t n_work, uint32_t n_cap, const double2* x1, const double2* x2,
                        const float* l10, const uint32_t* mt_init, uint32_t max_it, GeoResult* res, uint8_t* mask, uint32_t* scratch, hipStream_t stream,
                        const double* b1 = nullptr, const double* b2 = nullptr, uint32_t ahead = 1) {
  if (!n_work) return MVGX_OK;
  constexpr int W = 4;
  const size_t lds = (size_t)W * (kMtN + wave_scratch_words<MODEL>()) * sizeof(uint32_t);
  constexpr bool kHasAheadForm = model_solves_ahead<MODEL>();
  if (kHasAheadForm && ahead > 1) {
    MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&geofilter_f_acransac_kernel<W, true, MODEL, kHasAheadForm>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((geofilter_f_acransac_kernel<W, true, MODEL, kHasAheadForm>), dim3((n_work + W - 1) / W), dim3(64 * W), lds, stream, d_pairs, d_order, n_work, n_cap, x1, x2,
                       l10, mt_init, max_it, res, mask, scratch, b1, b2, ahead);
  } else {
    MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&geofilter_f_acransac_kernel<W, true, MODEL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((geofilter_f_acransac_kernel<W, true, MODEL, false>), dim3((n_work + W - 1) / W), dim3(64 * W), lds, stream, d_pairs, d_order, n_work, n_cap, x1, x2,
                       l10, mt_init, max_it, res, mask, scratch, b1, b2, ahead);
  }
  MVGX_HIP(hipGetLastError());
  return MVGX_OK;
}

}  // namespace

namespace {

// Where the correspondences come from: gathered coordinates (xI / xJ, image_wh per pair) or, `indexed`, the feature positions of
// every image plus index pairs (image_wh per image)
struct GeoSource {
  bool indexed = false;
  const double* xI = nullptr; const double* xJ = nullptr;
  const double* feat_xy = nullptr; const uint64_t* feat_start = nullptr; uint32_t n_images = 0;
  const uint32_t* pair_images = nullptr; const uint32_t* ij = nullptr;
  // essential model: bearing vectors (3 per match, or 3 per feature when indexed) and the calibration matrices (row-major 3 x 3:
  // 18 doubles per pair {K_I, K_J}, or 9 per image when indexed)
  const double* bI = nullptr; const double* bJ = nullptr; const double* feat_bearing = nullptr; const double* K = nullptr;
  const double* pair_precision = nullptr;   // orthographic model: the bound of every pair (NULL: opt->precision for all)
};

// inverse of a 3 x 3 matrix by cofactors (what Eigen's Matrix3d::inverse() evaluates: cofactors x 1 / det)
inline void inverse3(const double* m, double* inv) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
  const double det = c00 * m[0] + c10 * m[1] + c20 * m[2];
  const double id = 1.0 / det;
  inv[0] = c00 * id; inv[3] = c10 * id; inv[6] = c20 * id;
  inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  inv[2] = (m[1] * m[5] - m[2] * m[4]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// the launches of the two forms for one model: pairs of more than kCap0 correspondences (largest first) with their tables in global
// scratch, the others with them in LDS. (Until the end of round 5 there were three more LDS classes - up to 1 024, 4 096 and 12 000
// correspondences at 4 / 2 / 1 waves per workgroup: their LDS left one workgroup per CU, and a pair of 2 000 correspondences ran 2.5 x
// slower per iteration than it does in the global form, call r5_61.)
template <int MODEL>
int launch_classes(const GeoPair* d_pairs, const uint32_t* ord, const std::vector<uint32_t>& order, uint32_t c_global, uint32_t cap0, uint64_t n_total, uint64_t n_pairs,
                   const double2* px1, const double2* px2, const float* l10, const uint32_t* mt, uint32_t max_it, GeoResult* res, uint8_t* mask, DevBuf& d_tables,
                   hipStream_t stream, const double* b1 = nullptr, const double* b2 = nullptr, uint32_t ahead = 1) {
  int rc;
  if (c_global) {
    if ((rc = d_tables.alloc(3 * ((size_t)n_total + 4 * (size_t)n_pairs + 4) * sizeof(uint32_t)))) return rc;
    if ((rc = launch_class_global<MODEL>(d_pairs, ord, c_global, 0, px1, px2, l10, mt, max_it, res, mask, static_cast<uint32_t*>(d_tables.p), stream, b1, b2, ahead))) return rc;
  }
  return launch_class<4, MODEL>(d_pairs, ord + c_global, (uint32_t)order.size() - c_global, cap0, px1, px2, l10, mt, max_it, res, mask, stream, b1, b2, ahead);
}

int geofilter_run(int device, int model, const GeoSource& src, const uint64_t* match_start, const uint32_t* image_wh,
                  uint64_t n_pairs, const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                  mvgx_geofilter_stats* stats) {
  const bool angular = model == kModelEA8 || model == kModelEU3;   // bearing vectors only: no pixels, no image sizes, no normalisation
  const bool bearings = model == kModelE || angular;
  const int min_samples = model == kModelH ? 4 : model == kModelE ? 5 : model == kModelEA8 ? 8 : (model == kModelEU3 || model == kModelEO) ? 3 : kMinSamples;
  const int max_models = (model == kModelH || angular) ? 1 : model == kModelE ? 10 : model == kModelEO ? 2 : kMaxModels;
  if (angular)
    MVGX_REQUIRE(!match_start || match_start[n_pairs] == 0 || (src.indexed ? src.feat_bearing != nullptr : (src.bI && src.bJ)), MVGX_ERR_ARG,
                 "mvgx_geofilter_e_angular_acransac: NULL bearing vectors");
  if (model == kModelE) {
    MVGX_REQUIRE(n_pairs == 0 || src.K, MVGX_ERR_ARG, "mvgx_geofilter_e_acransac: NULL calibration matrices");
    MVGX_REQUIRE(!match_start || match_start[n_pairs] == 0 || (src.indexed ? src.feat_bearing != nullptr : (src.bI && src.bJ)), MVGX_ERR_ARG,
                 "mvgx_geofilter_e_acransac: NULL bearing vectors");
  }
  MVGX_REQUIRE(opt && match_start && (n_pairs == 0 || ((image_wh || angular) && results)), MVGX_ERR_ARG, "mvgx_geofilter_f_acransac: NULL argument");
  const uint64_t n_total = match_start[n_pairs];
  MVGX_REQUIRE(n_total == 0 || inlier_mask, MVGX_ERR_ARG, "mvgx_geofilter_f_acransac: NULL inlier mask");
  if (src.indexed) {
    MVGX_REQUIRE(n_pairs == 0 || (src.pair_images && src.feat_start), MVGX_ERR_ARG, "mvgx_geofilter_f_acransac_indexed: NULL pair / feature-start array");
    MVGX_REQUIRE(n_total == 0 || ((src.feat_xy || angular) && src.ij), MVGX_ERR_ARG, "mvgx_geofilter_f_acransac_indexed: NULL position / index array");
    for (uint64_t p = 0; p < n_pairs; ++p)
      MVGX_REQUIRE(src.pair_images[2 * p] < src.n_images && src.pair_images[2 * p + 1] < src.n_images, MVGX_ERR_ARG,
                   "mvgx_geofilter_f_acransac_indexed: pair %llu names an image out of range", (unsigned long long)p);
  } else {
    MVGX_REQUIRE(n_total == 0 || angular || (src.xI && src.xJ), MVGX_ERR_ARG, "mvgx_geofilter_f_acransac: NULL correspondence array");
  }
  for (uint64_t p = 0; src.pair_precision && p < n_pairs; ++p)
    MVGX_REQUIRE(std::isfinite(src.pair_precision[p]) && src.pair_precision[p] > 0.0, MVGX_ERR_UNSUPPORTED,
                 "mvgx_geofilter_eo_acransac: pair %llu: precision must be a finite upper bound", (unsigned long long)p);
  MVGX_REQUIRE(src.pair_precision || (std::isfinite(opt->precision) && opt->precision > 0.0), MVGX_ERR_UNSUPPORTED,
               "mvgx_geofilter_f_acransac: precision must be a finite upper bound (the exhaustive NFA form of an unbounded precision is not "
               "reproduced on the device; main_GeometricFilter passes 4.0)");
  MVGX_REQUIRE(opt->max_iterations >= 1, MVGX_ERR_ARG, "mvgx_geofilter_f_acransac: max_iterations must be at least 1");
  constexpr uint32_t kCap0 = 256;             // correspondences per pair up to which a wave's tables (3 words per correspondence) live in LDS
  constexpr uint32_t kCapGlobal = 1u << 20;   // above kCap0 they live in global scratch (geofilter_f_acransac_kernel<4, true>)
  for (uint64_t p = 0; p < n_pairs; ++p) {
    MVGX_REQUIRE(match_start[p + 1] >= match_start[p], MVGX_ERR_ARG, "mvgx_geofilter_f_acransac: match_start must be non-decreasing");
    MVGX_REQUIRE(match_start[p + 1] - match_start[p] <= kCapGlobal, MVGX_ERR_UNSUPPORTED,
                 "mvgx_geofilter_f_acransac: pair %llu has %llu correspondences (limit %u)", (unsigned long long)p,
                 (unsigned long long)(match_start[p + 1] - match_start[p]), kCapGlobal);
  }
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = mvgx::select_device(device < 0 ? -1 : device);
  if (rc) return rc;
  // ---- host preparation (ACKernelAdaptor's constructor + NFA_Interface's constants), on host threads ----
  // (page-locked memory from the library's cache, not zeroed - every field of every pair is written below: a fresh std::vector of
  // 100 000 pairs was 50 MB of page faults per call, two thirds of this phase; the uploads from it are asynchronous DMA)
  mvgx::HostArena host_mem;
  GeoPair* hp = nullptr;
  double* norm = nullptr;   // {s1, tx1, ty1, s2, tx2, ty2}; the points are normalised on the device
  GeoResult* hr = nullptr;
  if ((rc = host_mem.array(&hp, std::max<uint64_t>(n_pairs, 1))) || (rc = host_mem.array(&norm, std::max<uint64_t>(n_pairs, 1) * 6)) ||
      (rc = host_mem.array(&hr, std::max<uint64_t>(n_pairs, 1))))
    return rc;
  const size_t norm_size = n_pairs * 6;
  uint32_t n_max = 0;
  for (uint64_t p = 0; p < n_pairs; ++p) n_max = std::max<uint32_t>(n_max, (uint32_t)(match_start[p + 1] - match_start[p]));
  const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), (size_t)32, (size_t)(n_pairs / 4096 + 1)}));
  std::atomic<int64_t> bad_pair{-1};
  auto prep = [&](unsigned tix) {
    for (uint64_t p = tix; p < n_pairs; p += T) {
      const uint64_t lo = match_start[p];
      const uint32_t n = (uint32_t)(match_start[p + 1] - lo);
      GeoPair& g = hp[p];
      g.start = lo; g.n = n; g.pad_ = 0;
      if (src.indexed) {
        const uint32_t I = src.pair_images[2 * p], J = src.pair_images[2 * p + 1];
        const uint64_t nI = src.feat_start[I + 1] - src.feat_start[I], nJ = src.feat_start[J + 1] - src.feat_start[J];
        bool bad = false;
        for (uint32_t m = 0; m < n; ++m) bad = bad || src.ij[2 * (lo + m)] >= nI || src.ij[2 * (lo + m) + 1] >= nJ;
        if (bad) bad_pair.store((int64_t)p);
      }
      double t[2][3] = {{1.0, 0.0, 0.0}, {1.0, 0.0, 0.0}};
      if (angular) for (int k = 0; k < 6; ++k) norm[6 * p + k] = 0.0;   // (uploaded, never read: no uninitialised bytes on the wire)
      for (int im = 0; im < 2 && !angular; ++im) {   // conditioning.cpp:44-53
        const uint32_t* whp = src.indexed ? image_wh + 2 * (size_t)src.pair_images[2 * p + im] : image_wh + 4 * p + 2 * im;
        const int w = (int)whp[0], h = (int)whp[1];
        const double dNorm = 1.0 / std::sqrt(static_cast<double>(w * h));
        t[im][0] = dNorm; t[im][1] = -.5f * w * dNorm; t[im][2] = -.5 * h * dNorm;
        if (model == kModelE || model == kModelEO) { t[im][0] = 1.0; t[im][1] = 0.0; t[im][2] = 0.0; }   // ACKernelAdaptorEssential{,Ortho}: N1 = N2 = I
        for (int k = 0; k < 3; ++k) norm[6 * p + 3 * im + k] = t[im][k];
      }
      if (model == kModelE) {
        const double* K1 = src.indexed ? src.K + 9 * (size_t)src.pair_images[2 * p] : src.K + 18 * p;
        const double* K2 = src.indexed ? src.K + 9 * (size_t)src.pair_images[2 * p + 1] : src.K + 18 * p + 9;
        double k2i[9];
        inverse3(K1, g.k1i);
        inverse3(K2, k2i);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) g.k2it[3 * a + b] = k2i[3 * b + a];
      } else {
        for (int u = 0; u < 9; ++u) { g.k1i[u] = 0.0; g.k2it[u] = 0.0; }
      }
      static const uint32_t kNoSize[2] = {1, 1};
      const uint32_t* wh2 = angular ? kNoSize : src.indexed ? image_wh + 2 * (size_t)src.pair_images[2 * p + 1] : image_wh + 4 * p + 2;
      const int w2 = (int)wh2[0], h2 = (int)wh2[1];
      // ACParametrizationHelper (robust_estimator_ACRansacKernelAdaptator.hpp:38-81): point-to-line for F, point-to-point for H
      // (RADIAN_ANGLE, :85-100: log10(1 / 2), multError 1 / 4 - the residual is a squared angle)
      const double logalpha0 = angular ? std::log10(1. / 2.)
                               : model == kModelH ? std::log10(M_PI / (w2 * static_cast<double>(h2)) / (t[1][0] * t[1][0]))
                               : (model == kModelE || model == kModelEO) ? std::log10(2. * std::hypot(w2, h2) / (w2 * static_cast<double>(h2)) / 0.5)   // LogAlpha0(w2, h2, 0.5)
                                                  : std::log10(2. * std::hypot(w2, h2) / (w2 * static_cast<double>(h2)) / t[1][0]);
      const double mult_error = angular ? 1. / 4. : model == kModelH ? 1.0 : 0.5;
      // angular models: E_ACRobust_Angular.hpp:117-119 hands ACRANSAC D2R(precision in degrees), which ACRANSAC compares with the
      // SQUARED angles as it is (robust_estimator_ACRansac.hpp:351-353: maxThreshold = precision * normalizer2()(0,0)^2, the identity here)
      // orthographic model: the value the functor hands to ACRANSAC (Eo_Robust.hpp:96-100: the mean of the two cameras'
      // imagePlane_toCameraPlaneError(precision^2)), per pair or one for all
      g.max_threshold = model == kModelEO ? (src.pair_precision ? src.pair_precision[p] : opt->precision)
                        : angular ? opt->precision * M_PI / 180.0 : opt->precision * opt->precision * t[1][0] * t[1][0];
      g.loge0 = n > (uint32_t)min_samples ? std::log10((double)max_models * (n - min_samples)) : 0.0;
      g.bins_by_interval = kBins / (g.max_threshold - 0.0);
      const double val = (g.max_threshold - 0.0) / static_cast<double>(kBins - 1);
      for (int b = 0; b < kBins; ++b) {
        g.bin_value[b] = val * static_cast<double>(b) + 0.0;
        g.logalpha_bin[b] = logalpha0 + mult_error * std::log10(g.bin_value[b] + std::numeric_limits<float>::epsilon());
      }
    }
  };
  {
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T; ++t) pool.emplace_back(prep, t);
    prep(0);
    for (auto& th : pool) th.join();
  }
  MVGX_REQUIRE(bad_pair.load() < 0, MVGX_ERR_ARG, "mvgx_geofilter_f_acransac_indexed: pair %lld has a feature index out of range", (long long)bad_pair.load());
  // pairs that run the estimation (more correspondences than a minimal sample), by size class, largest first inside a class
  std::vector<uint32_t> order;
  order.reserve(n_pairs);
  for (uint64_t p = 0; p < n_pairs; ++p) if (hp[p].n > (uint32_t)min_samples) order.push_back((uint32_t)p);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hp[a].n > hp[b].n; });
  // [0, c_global): more than kCap0 correspondences (tables in global scratch); the rest: tables in LDS. MVGX_GEO_GLOBAL_ABOVE = a smaller
  // bound (tests: the global form on small pairs)
  uint32_t global_above = kCap0;
  if (const char* e = getenv("MVGX_GEO_GLOBAL_ABOVE")) global_above = (uint32_t)std::min<int>((int)kCap0, std::max(1, atoi(e)));
  uint32_t c_global = 0;
  while (c_global < order.size() && hp[order[c_global]].n > global_above) ++c_global;
  std::vector<float> l10(n_max + 2);
  for (uint32_t i = 0; i <= n_max + 1; ++i) l10[i] = (float)::log10((double)static_cast<float>(i));   // log10 of a float through the C function, as the reference's unqualified call resolves
  uint32_t mt_init[kMtN];
  mt_init[0] = 5489u;
  for (int i = 1; i < kMtN; ++i) mt_init[i] = 1812433253u * (mt_init[i - 1] ^ (mt_init[i - 1] >> 30)) + (uint32_t)i;
  const auto t_prep = std::chrono::steady_clock::now();
  // MVGX_GEO_TIMING=1: where a call's time goes, on stderr (host preparation | device memory | uploads issued | kernels + downloads | results)
  const bool timing = getenv("MVGX_GEO_TIMING") != nullptr;
  auto t_mark = t_prep;
  double t_phase[4] = {0, 0, 0, 0};
  auto mark = [&](int k) { const auto t = std::chrono::steady_clock::now(); t_phase[k] += std::chrono::duration<double, std::milli>(t - t_mark).count(); t_mark = t; };
  // ---- device ----
  hipStream_t stream = nullptr;
  if ((rc = mvgx::acquire_stream(&stream))) return rc;
  int stream_device = 0;
  MVGX_HIP(hipGetDevice(&stream_device));
  struct StreamGuard { int dev; hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); mvgx::release_stream(dev, s); } } sg{stream_device, stream};
  DevBuf d_pairs, d_order, d_x1, d_x2, d_l10, d_mt, d_res, d_mask, d_raw1, d_raw2, d_norm, d_feat, d_fstart, d_pimg, d_b1, d_b2, d_fbear;
  const uint64_t n_feat = src.indexed && src.n_images ? src.feat_start[src.n_images] : 0;
  const uint64_t n_xy = angular ? 0 : n_total;   // (the angular models never touch pixel positions)
  if ((rc = d_pairs.alloc(n_pairs * sizeof(GeoPair))) || (rc = d_order.alloc(order.size() * sizeof(uint32_t))) || (rc = d_x1.alloc(n_xy * sizeof(double2))) ||
      (rc = d_x2.alloc(n_xy * sizeof(double2))) || (rc = d_l10.alloc(l10.size() * sizeof(float))) || (rc = d_mt.alloc(sizeof(mt_init))) ||
      (rc = d_res.alloc(n_pairs * sizeof(GeoResult))) || (rc = d_mask.alloc(n_total)) || (rc = d_norm.alloc(norm_size * sizeof(double))))
    return rc;
  if (src.indexed) {   // d_raw1: the index pairs
    if ((rc = d_raw1.alloc(n_total * sizeof(uint2))) || (rc = d_feat.alloc((angular ? 0 : n_feat) * sizeof(double2))) ||
        (rc = d_fstart.alloc(((size_t)src.n_images + 1) * sizeof(uint64_t))) || (rc = d_pimg.alloc(n_pairs * sizeof(uint2))))
      return rc;
  } else if ((rc = d_raw1.alloc(n_xy * sizeof(double2))) || (rc = d_raw2.alloc(n_xy * sizeof(double2)))) {
    return rc;
  }
  if (bearings) {
    if ((rc = d_b1.alloc(n_total * 3 * sizeof(double))) || (rc = d_b2.alloc(n_total * 3 * sizeof(double)))) return rc;
    if (src.indexed && (rc = d_fbear.alloc(n_feat * 3 * sizeof(double)))) return rc;
  }
  mark(0);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  MVGX_HIP(hipEventCreate(&e0));
  MVGX_HIP(hipEventCreate(&e1));
  struct EventGuard { hipEvent_t a, b; ~EventGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } eg{e0, e1};
  if (n_pairs) MVGX_HIP(hipMemcpyAsync(d_pairs.p, hp, n_pairs * sizeof(GeoPair), hipMemcpyHostToDevice, stream));
  if (!order.empty()) MVGX_HIP(hipMemcpyAsync(d_order.p, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  if (n_total) {
    if (src.indexed) {
      MVGX_HIP(hipMemcpyAsync(d_raw1.p, src.ij, n_total * sizeof(uint2), hipMemcpyHostToDevice, stream));
      if (!angular) MVGX_HIP(hipMemcpyAsync(d_feat.p, src.feat_xy, n_feat * sizeof(double2), hipMemcpyHostToDevice, stream));
      MVGX_HIP(hipMemcpyAsync(d_fstart.p, src.feat_start, ((size_t)src.n_images + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
      MVGX_HIP(hipMemcpyAsync(d_pimg.p, src.pair_images, n_pairs * sizeof(uint2), hipMemcpyHostToDevice, stream));
    } else if (!angular) {
      MVGX_HIP(hipMemcpyAsync(d_raw1.p, src.xI, n_total * sizeof(double2), hipMemcpyHostToDevice, stream));
      MVGX_HIP(hipMemcpyAsync(d_raw2.p, src.xJ, n_total * sizeof(double2), hipMemcpyHostToDevice, stream));
    }
    MVGX_HIP(hipMemcpyAsync(d_norm.p, norm, norm_size * sizeof(double), hipMemcpyHostToDevice, stream));
    MVGX_HIP(hipMemsetAsync(d_mask.p, 0, n_total, stream));
    if (bearings) {
      if (src.indexed) {
        MVGX_HIP(hipMemcpyAsync(d_fbear.p, src.feat_bearing, n_feat * 3 * sizeof(double), hipMemcpyHostToDevice, stream));
      } else {
        MVGX_HIP(hipMemcpyAsync(d_b1.p, src.bI, n_total * 3 * sizeof(double), hipMemcpyHostToDevice, stream));
        MVGX_HIP(hipMemcpyAsync(d_b2.p, src.bJ, n_total * 3 * sizeof(double), hipMemcpyHostToDevice, stream));
      }
    }
  }
  MVGX_HIP(hipMemcpyAsync(d_l10.p, l10.data(), l10.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  MVGX_HIP(hipMemcpyAsync(d_mt.p, mt_init, sizeof(mt_init), hipMemcpyHostToDevice, stream));
  if (n_pairs) MVGX_HIP(hipMemsetAsync(d_res.p, 0, n_pairs * sizeof(GeoResult), stream));
  MVGX_HIP(hipEventRecord(e0, stream));
  mark(1);
  const uint32_t* ord = static_cast<const uint32_t*>(d_order.p);
  const auto* px1 = static_cast<const double2*>(d_x1.p);
  const auto* px2 = static_cast<const double2*>(d_x2.p);
  if (n_total) {   // (plain pointers in the launch: the buffers own their memory and must not be captured by value)
    const auto* pp = static_cast<const GeoPair*>(d_pairs.p);
    const auto* pn = static_cast<const double*>(d_norm.p);
    const auto *r1 = static_cast<const double2*>(d_raw1.p), *r2 = static_cast<const double2*>(d_raw2.p);
    auto *o1 = static_cast<double2*>(d_x1.p), *o2 = static_cast<double2*>(d_x2.p);
    if (src.indexed) {
      const auto* ft = static_cast<const double2*>(d_feat.p);
      const auto* fs = static_cast<const uint64_t*>(d_fstart.p);
      const auto *pim = static_cast<const uint2*>(d_pimg.p), *mij = static_cast<const uint2*>(d_raw1.p);
      if (!angular) hipLaunchKernelGGL(geofilter_normalize_indexed_kernel, dim3((unsigned)n_pairs), dim3(256), 0, stream, pp, pn, (uint32_t)n_pairs, ft, fs, pim, mij, o1, o2);
      if (bearings) {
        const double* fb = static_cast<const double*>(d_fbear.p);
        double *ob1 = static_cast<double*>(d_b1.p), *ob2 = static_cast<double*>(d_b2.p);
        hipLaunchKernelGGL(geofilter_gather_bearings_kernel, dim3((unsigned)n_pairs), dim3(256), 0, stream, pp, (uint32_t)n_pairs, fb, fs, pim, mij, ob1, ob2);
      }
    } else if (!angular) {
      hipLaunchKernelGGL(geofilter_normalize_kernel, dim3((unsigned)n_pairs), dim3(256), 0, stream, pp, pn, (uint32_t)n_pairs, r1, r2, o1, o2);
    }
    MVGX_HIP(hipGetLastError());
  }
  DevBuf d_tables;
  {
    // fundamental / homography / essential models: samples drawn and solved ahead of their iterations, four side by side (1: one minimal solve per
    // iteration, the form of rounds 3-4; results equal)
    uint32_t e_ahead = kAhead;
    if (const char* env = getenv("MVGX_GEO_AHEAD")) e_ahead = (uint32_t)std::min(kAhead, std::max(1, atoi(env)));
    else if (const char* env2 = getenv("MVGX_GEO_E_AHEAD")) e_ahead = (uint32_t)std::min(kAhead, std::max(1, atoi(env2)));
    auto* dp = static_cast<const GeoPair*>(d_pairs.p);
    auto* dl = static_cast<const float*>(d_l10.p);
    auto* dm = static_cast<const uint32_t*>(d_mt.p);
    auto* dr = static_cast<GeoResult*>(d_res.p);
    auto* dk = static_cast<uint8_t*>(d_mask.p);
    const double *pb1 = static_cast<const double*>(d_b1.p), *pb2 = static_cast<const double*>(d_b2.p);
    rc = model == kModelH ? launch_classes<kModelH>(dp, ord, order, c_global, kCap0, n_total, n_pairs, px1, px2, dl, dm, opt->max_iterations, dr, dk, d_tables, stream, nullptr, nullptr, e_ahead)
         : model == kModelEO ? launch_classes<kModelEO>(dp, ord, order, c_global, kCap0, n_total, n_pairs, px1, px2, dl, dm, opt->max_iterations, dr, dk, d_tables, stream)
         : model == kModelEA8 ? launch_classes<kModelEA8>(dp, ord, order, c_global, kCap0, n_total, n_pairs, px1, px2, dl, dm, opt->max_iterations, dr, dk, d_tables, stream, pb1, pb2)
         : model == kModelEU3 ? launch_classes<kModelEU3>(dp, ord, order, c_global, kCap0, n_total, n_pairs, px1, px2, dl, dm, opt->max_iterations, dr, dk, d_tables, stream, pb1, pb2)
         : model == kModelE ? launch_classes<kModelE>(dp, ord, order, c_global, kCap0, n_total, n_pairs, px1, px2, dl, dm, opt->max_iterations, dr, dk, d_tables, stream,
                                                      static_cast<const double*>(d_b1.p), static_cast<const double*>(d_b2.p), e_ahead)
                            : launch_classes<kModelF>(dp, ord, order, c_global, kCap0, n_total, n_pairs, px1, px2, dl, dm, opt->max_iterations, dr, dk, d_tables, stream, nullptr, nullptr, e_ahead);
    if (rc) return rc;
  }
  MVGX_HIP(hipEventRecord(e1, stream));
  if (n_pairs) MVGX_HIP(hipMemcpyAsync(hr, d_res.p, n_pairs * sizeof(GeoResult), hipMemcpyDeviceToHost, stream));
  if (n_total) MVGX_HIP(hipMemcpyAsync(inlier_mask, d_mask.p, n_total, hipMemcpyDeviceToHost, stream));
  MVGX_HIP(hipStreamSynchronize(stream));
  mark(2);
  float kernel_ms = 0.f;
  (void)hipEventElapsedTime(&kernel_ms, e0, e1);
  // ---- results in the reference's terms: Unnormalize (conditioning.cpp:87-89), unormalizeError, the 2.5 x 7 acceptance ----
  uint64_t n_ok = 0, n_inl = 0, n_iter = 0, n_mod = 0, clocks = 0;
  for (uint64_t p = 0; p < n_pairs; ++p) {
    mvgx_geofilter_result& o = results[p];
    const GeoResult& r = hr[p];
    const bool ran = hp[p].n > (uint32_t)min_samples;
    if (ran) { n_iter += r.n_iterations; n_mod += r.n_models; clocks += r.clocks; }
    double Fm[9];
    for (int u = 0; u < 9; ++u) Fm[u] = (ran && r.have_model) ? r.F[u] : ((u % 4 == 0) ? 1.0 : 0.0);   // m_F starts as the identity
    double err = ran ? r.error_max : 0.0;
    if (ran && r.n_inliers > 0 && (model == kModelE || model == kModelEO)) {
      // ACKernelAdaptorEssential{,Ortho}: Unnormalize does nothing, unormalizeError(val) = val (the residual as it is)
    } else if (angular) {
      // ACKernelAdaptor_AngularRadianError: Unnormalize does nothing, unormalizeError(val) = sqrt(val): an angle in radians. The
      // reference's solvers return unit vectors (eigenvectors): the model is scaled to unit Frobenius norm (its sign stays free).
      if (ran && r.n_inliers > 0) err = std::sqrt(err);
      if (ran && r.have_model) {
        double n2 = 0.0;
        for (int u = 0; u < 9; ++u) n2 += Fm[u] * Fm[u];
        if (n2 > 0.0) for (int u = 0; u < 9; ++u) Fm[u] /= std::sqrt(n2);
      }
    } else if (ran && r.n_inliers > 0) {
      const double* t = &norm[6 * p];
      const double N1[9] = {t[0], 0, t[1], 0, t[0], t[2], 0, 0, 1}, N2[9] = {t[3], 0, t[4], 0, t[3], t[5], 0, 0, 1};
      double tmp[9], res[9];
      if (model == kModelH) {   // UnnormalizerI (conditioning.cpp:80-82): H = N2^-1 H N1; N2 = [s 0 tx; 0 s ty; 0 0 1]
        const double is = 1.0 / t[3];
        const double N2i[9] = {is, 0, -t[4] * is, 0, is, -t[5] * is, 0, 0, 1};
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double s_ = 0; for (int k = 0; k < 3; ++k) s_ += N2i[3 * a + k] * Fm[3 * k + b]; tmp[3 * a + b] = s_; }
      } else   // UnnormalizerT: F = N2^T F N1
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double s_ = 0; for (int k = 0; k < 3; ++k) s_ += N2[3 * k + a] * Fm[3 * k + b]; tmp[3 * a + b] = s_; }
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double s_ = 0; for (int k = 0; k < 3; ++k) s_ += tmp[3 * a + k] * N1[3 * k + b]; res[3 * a + b] = s_; }
      std::memcpy(Fm, res, sizeof(res));
      err = std::sqrt(err) / t[3];
    }
    std::memcpy(o.F, Fm, sizeof(Fm));
    o.precision_robust = err;
    o.nfa = ran ? r.min_nfa : 0.0;
    o.n_inliers = ran ? r.n_inliers : 0;
    o.ok = ran && (double)r.n_inliers > min_samples * 2.5;
    n_ok += o.ok; n_inl += o.ok ? o.n_inliers : 0;
  }
  if (stats) {
    stats->n_pairs = n_pairs; stats->n_pairs_estimated = order.size(); stats->n_pairs_ok = n_ok; stats->n_inliers = n_inl;
    stats->kernel_ms = kernel_ms;
    mark(3);
    if (timing)
      fprintf(stderr, "[mvgx geofilter] %llu pairs, %llu matches: host preparation %.2f ms | device memory %.2f | uploads issued %.2f | kernels + downloads %.2f (kernels %.2f) | results %.2f\n",
              (unsigned long long)n_pairs, (unsigned long long)n_total, std::chrono::duration<double, std::milli>(t_prep - t_begin).count(), t_phase[0], t_phase[1], t_phase[2],
              (double)kernel_ms, t_phase[3]);
    stats->host_prepare_ms = std::chrono::duration<double, std::milli>(t_prep - t_begin).count();
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    stats->n_iterations = n_iter; stats->n_models = n_mod; stats->wave_clocks = clocks;
  }
  return MVGX_OK;
}

}  // namespace

extern "C" {

int mvgx_geofilter_f_acransac(int device, const double* xI, const double* xJ, const uint64_t* match_start, const uint32_t* image_wh,
                              uint64_t n_pairs, const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                              mvgx_geofilter_stats* stats) {
  GeoSource src;
  src.xI = xI; src.xJ = xJ;
  return geofilter_run(device, kModelF, src, match_start, image_wh, n_pairs, opt, inlier_mask, results, stats);
}

int mvgx_geofilter_h_acransac(int device, const double* xI, const double* xJ, const uint64_t* match_start, const uint32_t* image_wh,
                              uint64_t n_pairs, const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                              mvgx_geofilter_stats* stats) {
  GeoSource src;
  src.xI = xI; src.xJ = xJ;
  return geofilter_run(device, kModelH, src, match_start, image_wh, n_pairs, opt, inlier_mask, results, stats);
}

int mvgx_geofilter_f_acransac_indexed(int device, const double* feat_xy, const uint64_t* feat_start, const uint32_t* image_wh, uint32_t n_images,
                                      const uint32_t* pairs, const uint64_t* match_start, const uint32_t* ij, uint64_t n_pairs,
                                      const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                                      mvgx_geofilter_stats* stats) {
  GeoSource src;
  src.indexed = true;
  src.feat_xy = feat_xy; src.feat_start = feat_start; src.n_images = n_images; src.pair_images = pairs; src.ij = ij;
  return geofilter_run(device, kModelF, src, match_start, image_wh, n_pairs, opt, inlier_mask, results, stats);
}

int mvgx_geofilter_h_acransac_indexed(int device, const double* feat_xy, const uint64_t* feat_start, const uint32_t* image_wh, uint32_t n_images,
                                      const uint32_t* pairs, const uint64_t* match_start, const uint32_t* ij, uint64_t n_pairs,
                                      const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                                      mvgx_geofilter_stats* stats) {
  GeoSource src;
  src.indexed = true;
  src.feat_xy = feat_xy; src.feat_start = feat_start; src.n_images = n_images; src.pair_images = pairs; src.ij = ij;
  return geofilter_run(device, kModelH, src, match_start, image_wh, n_pairs, opt, inlier_mask, results, stats);
}

// The essential-matrix model (E_ACRobust.hpp:39-150): ACKernelAdaptorEssential<FivePointSolver, EpipolarDistanceError> + ACRANSAC.
int mvgx_geofilter_e_acransac(int device, const double* xI, const double* xJ, const double* bearingI, const double* bearingJ, const uint64_t* match_start,
                              const uint32_t* image_wh, const double* K, uint64_t n_pairs, const mvgx_geofilter_options* opt, uint8_t* inlier_mask,
                              mvgx_geofilter_result* results, mvgx_geofilter_stats* stats) {
  GeoSource src;
  src.xI = xI; src.xJ = xJ; src.bI = bearingI; src.bJ = bearingJ; src.K = K;
  return geofilter_run(device, kModelE, src, match_start, image_wh, n_pairs, opt, inlier_mask, results, stats);
}

int mvgx_geofilter_e_acransac_indexed(int device, const double* feat_xy, const double* feat_bearing, const uint64_t* feat_start, const uint32_t* image_wh,
                                      const double* image_K, uint32_t n_images, const uint32_t* pairs, const uint64_t* match_start, const uint32_t* ij,
                                      uint64_t n_pairs, const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                                      mvgx_geofilter_stats* stats) {
  GeoSource src;
  src.indexed = true;
  src.feat_xy = feat_xy; src.feat_start = feat_start; src.n_images = n_images; src.pair_images = pairs; src.ij = ij;
  src.feat_bearing = feat_bearing; src.K = image_K;
  return geofilter_run(device, kModelE, src, match_start, image_wh, n_pairs, opt, inlier_mask, results, stats);
}

// The essential matrix with the angular residual on bearing vectors (E_ACRobust_Angular.hpp:33-191): upright = 0 the eight-point
// solver, 1 the three-point upright solver.
int mvgx_geofilter_e_angular_acransac(int device, const double* bearingI, const double* bearingJ, const uint64_t* match_start, uint64_t n_pairs, int upright,
                                      const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results, mvgx_geofilter_stats* stats) {
  GeoSource src;
  src.bI = bearingI; src.bJ = bearingJ;
  return geofilter_run(device, upright ? kModelEU3 : kModelEA8, src, match_start, nullptr, n_pairs, opt, inlier_mask, results, stats);
}

int mvgx_geofilter_e_angular_acransac_indexed(int device, const double* feat_bearing, const uint64_t* feat_start, uint32_t n_images, const uint32_t* pairs,
                                              const uint64_t* match_start, const uint32_t* ij, uint64_t n_pairs, int upright, const mvgx_geofilter_options* opt,
                                              uint8_t* inlier_mask, mvgx_geofilter_result* results, mvgx_geofilter_stats* stats) {
  GeoSource src;
  src.indexed = true;
  src.feat_start = feat_start; src.n_images = n_images; src.pair_images = pairs; src.ij = ij; src.feat_bearing = feat_bearing;
  return geofilter_run(device, upright ? kModelEU3 : kModelEA8, src, match_start, nullptr, n_pairs, opt, inlier_mask, results, stats);
}

// The orthographic essential matrix (Eo_Robust.hpp:35-165): xI / xJ (feat_xy) carry the hnormalized bearing vectors.
int mvgx_geofilter_eo_acransac(int device, const double* xI, const double* xJ, const uint64_t* match_start, const uint32_t* image_wh, const double* pair_precision,
                               uint64_t n_pairs, const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results, mvgx_geofilter_stats* stats) {
  GeoSource src;
  src.xI = xI; src.xJ = xJ; src.pair_precision = pair_precision;
  return geofilter_run(device, kModelEO, src, match_start, image_wh, n_pairs, opt, inlier_mask, results, stats);
}

int mvgx_geofilter_eo_acransac_indexed(int device, const double* feat_xy, const uint64_t* feat_start, const uint32_t* image_wh, uint32_t n_images,
                                       const uint32_t* pairs, const uint64_t* match_start, const uint32_t* ij, const double* pair_precision, uint64_t n_pairs,
                                       const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results, mvgx_geofilter_stats* stats) {
  GeoSource src;
  src.indexed = true;
  src.feat_xy = feat_xy; src.feat_start = feat_start; src.n_images = n_images; src.pair_images = pairs; src.ij = ij; src.pair_precision = pair_precision;
  return geofilter_run(device, kModelEO, src, match_start, image_wh, n_pairs, opt, inlier_mask, results, stats);
}

// Test hook (not declared in include/mvgx.h): out[0] = add_rn(mul_rn(a, b), c) as the kernels of this file evaluate it, out[1] = fma(a, b, c) -
// tests/test_rounded_ops_gpu.py feeds values on which the two differ, so that a toolchain that contracts the former is noticed
__global__ void rounded_ops_debug_kernel(const double* abc, double* out) {
  out[0] = add_rn(mul_rn(abc[0], abc[1]), abc[2]);
  out[1] = fma(abc[0], abc[1], abc[2]);
}
int mvgx_debug_rounded_ops(const double* abc, double* out) {
  MVGX_REQUIRE(abc && out, MVGX_ERR_ARG, "mvgx_debug_rounded_ops: NULL argument");
  int rc = mvgx::select_device(-1);
  if (rc) return rc;
  DevBuf din, dout;
  if ((rc = din.alloc(3 * sizeof(double))) || (rc = dout.alloc(2 * sizeof(double)))) return rc;
  MVGX_HIP(hipMemcpy(din.p, abc, 3 * sizeof(double), hipMemcpyHostToDevice));
  {
    const double* pi = static_cast<const double*>(din.p);
    double* po = static_cast<double*>(dout.p);
    hipLaunchKernelGGL(rounded_ops_debug_kernel, dim3(1), dim3(1), 0, nullptr, pi, po);
  }
  MVGX_HIP(hipGetLastError());
  MVGX_HIP(hipStreamSynchronize(nullptr));
  MVGX_HIP(hipMemcpy(out, dout.p, 2 * sizeof(double), hipMemcpyDeviceToHost));
  return MVGX_OK;
}

// Test hook (not declared in include/mvgx.h): the five-point solver alone on one sample of five bearing pairs (b1, b2: 5 x 3 doubles);
// Es_out receives up to ten essential matrices (row-major), *n_out their number.
__global__ void five_point_debug_kernel(const double* b1, const double* b2, double* Es_out, int* n_out) {
  __shared__ double scr[five_point::kScratch + 90];
  const int lane = threadIdx.x & 63;
  const uint32_t s[7] = {0, 1, 2, 3, 4, 0, 0};
  const int n = five_point::solve(b1, b2, s, lane, scr, scr + five_point::kScratch);
  for (int e = lane; e < 9 * n; e += 64) Es_out[e] = scr[five_point::kScratch + e];
  if (lane == 0) *n_out = n;
}
// The same for solve4: four samples (b1, b2: 4 x 5 x 3 doubles), one per 16-lane row; Es_out 4 x 90, n_out 4
__global__ __launch_bounds__(64) void five_point4_debug_kernel(const double* b1, const double* b2, double* Es_out, int* n_out) {
  __shared__ double scr[4 * five_point::kScratch + 4 * 90];
  const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15;
  const uint32_t s[5] = {5u * g, 5u * g + 1, 5u * g + 2, 5u * g + 3, 5u * g + 4};
  const int n = five_point::solve4(b1, b2, s, lane, scr, scr + 4 * five_point::kScratch);
  for (int e = gl; e < 9 * n; e += 16) Es_out[90 * g + e] = scr[4 * five_point::kScratch + 90 * g + e];
  if (gl == 0) n_out[g] = n;
}
// diagnostic (not declared in include/mvgx.h): five-point solves of this process whose eigenvalues fell back to hqr; reset != 0 clears
int mvgx_debug_five_point_fallbacks(unsigned long long* out, int reset) {
  unsigned long long v = 0;
  MVGX_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(five_point::g_hqr_fallbacks), sizeof(v)));
  if (out) *out = v;
  if (reset) { v = 0; MVGX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(five_point::g_hqr_fallbacks), &v, sizeof(v))); }
  return MVGX_OK;
}
#ifdef MVGX_GEO_STAMPS
int mvgx_debug_geo_stamps(unsigned long long* out8, int reset) {
  unsigned long long v[8];
  MVGX_HIP(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_geo_stamps), sizeof(v)));
  for (int k = 0; k < 8; ++k) out8[k] = v[k];
  if (reset) { for (auto& x : v) x = 0; MVGX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_geo_stamps), v, sizeof(v))); }
  return MVGX_OK;
}
#endif
int mvgx_debug_five_point(const double* b1, const double* b2, double* Es_out, int* n_out) {
  MVGX_REQUIRE(b1 && b2 && Es_out && n_out, MVGX_ERR_ARG, "mvgx_debug_five_point: NULL argument");
  int rc = mvgx::select_device(-1);
  if (rc) return rc;
  DevBuf d1, d2, dE, dn;
  if ((rc = d1.alloc(15 * sizeof(double))) || (rc = d2.alloc(15 * sizeof(double))) || (rc = dE.alloc(90 * sizeof(double))) || (rc = dn.alloc(sizeof(int)))) return rc;
  MVGX_HIP(hipMemcpy(d1.p, b1, 15 * sizeof(double), hipMemcpyHostToDevice));
  MVGX_HIP(hipMemcpy(d2.p, b2, 15 * sizeof(double), hipMemcpyHostToDevice));
  MVGX_HIP(hipMemset(dE.p, 0, 90 * sizeof(double)));
  {   // (plain pointers in the launch: the buffers own their memory and must not be captured by value)
    const double *p1 = static_cast<const double*>(d1.p), *p2 = static_cast<const double*>(d2.p);
    double* pE = static_cast<double*>(dE.p);
    int* pn = static_cast<int*>(dn.p);
    hipLaunchKernelGGL(five_point_debug_kernel, dim3(1), dim3(64), 0, nullptr, p1, p2, pE, pn);
  }
  MVGX_HIP(hipGetLastError());
  MVGX_HIP(hipStreamSynchronize(nullptr));
  MVGX_HIP(hipMemcpy(Es_out, dE.p, 90 * sizeof(double), hipMemcpyDeviceToHost));
  MVGX_HIP(hipMemcpy(n_out, dn.p, sizeof(int), hipMemcpyDeviceToHost));
  return MVGX_OK;
}

// Test hook (not declared in include/mvgx.h): five_point::solve4 on four samples of five bearing pairs (b1, b2: 4 x 5 x 3 doubles)
int mvgx_debug_five_point4(const double* b1, const double* b2, double* Es_out, int* n_out) {
  MVGX_REQUIRE(b1 && b2 && Es_out && n_out, MVGX_ERR_ARG, "mvgx_debug_five_point4: NULL argument");
  int rc = mvgx::select_device(-1);
  if (rc) return rc;
  DevBuf d1, d2, dE, dn;
  if ((rc = d1.alloc(60 * sizeof(double))) || (rc = d2.alloc(60 * sizeof(double))) || (rc = dE.alloc(360 * sizeof(double))) || (rc = dn.alloc(4 * sizeof(int)))) return rc;
  MVGX_HIP(hipMemcpy(d1.p, b1, 60 * sizeof(double), hipMemcpyHostToDevice));
  MVGX_HIP(hipMemcpy(d2.p, b2, 60 * sizeof(double), hipMemcpyHostToDevice));
  MVGX_HIP(hipMemset(dE.p, 0, 360 * sizeof(double)));
  {
    const double *p1 = static_cast<const double*>(d1.p), *p2 = static_cast<const double*>(d2.p);
    double* pE = static_cast<double*>(dE.p);
    int* pn = static_cast<int*>(dn.p);
    hipLaunchKernelGGL(five_point4_debug_kernel, dim3(1), dim3(64), 0, nullptr, p1, p2, pE, pn);
  }
  MVGX_HIP(hipGetLastError());
  MVGX_HIP(hipStreamSynchronize(nullptr));
  MVGX_HIP(hipMemcpy(Es_out, dE.p, 360 * sizeof(double), hipMemcpyDeviceToHost));
  MVGX_HIP(hipMemcpy(n_out, dn.p, 4 * sizeof(int), hipMemcpyDeviceToHost));
  return MVGX_OK;
}

}  // extern "C"
