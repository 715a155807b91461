// ref_shim_geofilter.cpp - TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" wrapper (our code) around the REFERENCE's own a-contrario fundamental-matrix filter, compiled in place from
// /root/reference/src by oracle/Makefile (target ref_geofilter) into oracle/_ref/libref_geofilter.so. It runs, per image pair, what
// GeometricFilter_FMatrix_AC::Robust_estimation (matching_image_collection/F_ACRobust.hpp:65-122) runs:
//   robust::ACKernelAdaptor<fundamental::kernel::SevenPointSolver, fundamental::kernel::EpipolarDistanceError, UnnormalizerT, Mat3>
//     (robust_estimation/robust_estimator_ACRansacKernelAdaptator.hpp:104-202, multiview/solver_fundamental_kernel.cpp:37-93,157-166)
//   robust::ACRANSAC (robust_estimation/robust_estimator_ACRansac.hpp:339-489)
// with the pairs spread over OpenMP threads as ImageCollectionGeometricFilter::Robust_model_estimation does
// (matching_image_collection/GeometricFilter.hpp:80-131). The feature positions arrive as the Mat2X pair MatchesPairToMat would
// build (Geometric_Filter_utils.hpp:56-64); the container bookkeeping around it is not part of this checker.
// Used (a) to pin oracle/geofilter_oracle.cpp and the device path, (b) as the "reference" CPU baseline of bench_geofilter.py.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include <omp.h>

#include "openMVG/multiview/solver_fundamental_kernel.hpp"
#include "openMVG/numeric/numeric.h"
#include "openMVG/robust_estimation/robust_estimator_ACRansac.hpp"
#include "openMVG/robust_estimation/robust_estimator_ACRansacKernelAdaptator.hpp"

using namespace openMVG;

extern "C" {

// pairs: for pair p the correspondences [start[p], start[p + 1]) of xI / xJ (2 doubles each, pixels), image sizes wh[4 p ..] =
// {w_I, h_I, w_J, h_J}. Outputs: inlier_mask per correspondence (1: geometric inlier of a pair whose estimation succeeded),
// ok[p] (Robust_estimation returned true), F[9 p ..] row-major (m_F), prec[p] (m_dPrecision_robust = ACRansacOut.first),
// nfa[p] (ACRansacOut.second). Returns the seconds spent.
double ref_geofilter_f_acransac(const double* xI, const double* xJ, const uint64_t* start, const uint32_t* wh, uint64_t n_pairs,
                                double precision, uint32_t max_iterations, int num_threads, uint8_t* inlier_mask, uint8_t* ok, double* F,
                                double* prec, double* nfa) {
  using KernelType = robust::ACKernelAdaptor<fundamental::kernel::SevenPointSolver, fundamental::kernel::EpipolarDistanceError, UnnormalizerT, Mat3>;
  const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic) num_threads(num_threads > 0 ? num_threads : omp_get_max_threads())
  for (int64_t p = 0; p < (int64_t)n_pairs; ++p) {
    const uint64_t lo = start[p], n = start[p + 1] - lo;
    Mat2X x1(2, n), x2(2, n);
    for (uint64_t i = 0; i < n; ++i) {
      x1.col(i) << xI[2 * (lo + i)], xI[2 * (lo + i) + 1];
      x2.col(i) << xJ[2 * (lo + i)], xJ[2 * (lo + i) + 1];
    }
    std::memset(inlier_mask + lo, 0, n);
    Mat3 model = Mat3::Identity();
    const KernelType kernel(x1, wh[4 * p], wh[4 * p + 1], x2, wh[4 * p + 2], wh[4 * p + 3], true);
    const double upper_bound_precision = Square(precision);   // F_ACRobust.hpp:98
    std::vector<uint32_t> vec_inliers;
    const std::pair<double, double> out = robust::ACRANSAC(kernel, vec_inliers, max_iterations, &model, upper_bound_precision);
    const bool good = vec_inliers.size() > KernelType::MINIMUM_SAMPLES * 2.5;   // F_ACRobust.hpp:103
    ok[p] = good ? 1 : 0;
    prec[p] = out.first; nfa[p] = out.second;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) F[9 * p + 3 * r + c] = model(r, c);
    if (good)
      for (const uint32_t idx : vec_inliers) inlier_mask[lo + idx] = 1;
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
