// oracle/ref_time_acransac.cpp - TEST / MEASUREMENT INFRASTRUCTURE ONLY (never linked into the product).
// Times the REFERENCE's geometric filter on synthetic two-view correspondences with the reference's own classes, compiled where
// they lie under /root/reference (oracle/Makefile target ref_acransac_time): what GeometricFilter_FMatrix_AC::Robust_estimation
// runs per image pair (matching_image_collection/F_ACRobust.hpp:65-122): ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError>
// + ACRANSAC (robust_estimation/robust_estimator_ACRansac.hpp:339-489) with the defaults of main_GeometricFilter (4 px, 2048
// iterations). The numbers back DESIGN.md section 7 (why SURVEY row N2 stays on the host).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "openMVG/multiview/solver_fundamental_kernel.hpp"
#include "openMVG/robust_estimation/robust_estimator_ACRansac.hpp"
#include "openMVG/robust_estimation/robust_estimator_ACRansacKernelAdaptator.hpp"

using namespace openMVG;
using Clock = std::chrono::steady_clock;
static double us(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

static void one_case(int n, double inlier_fraction) {
  std::mt19937 g(5);
  std::uniform_real_distribution<double> U(-1, 1);
  Mat2X x1(2, n), x2(2, n);
  const double f = 1000, c = 500, th = 0.1;   // two pinhole views of a point cloud, 0.3 px noise, the rest uniform outliers
  for (int i = 0; i < n; ++i) {
    const Vec3 X(U(g) * 2, U(g) * 2, 6 + U(g));
    const Vec3 Y(cos(th) * X(0) + sin(th) * X(2) - 0.5, X(1), -sin(th) * X(0) + cos(th) * X(2));
    x1.col(i) << f * X(0) / X(2) + c, f * X(1) / X(2) + c;
    x2.col(i) << f * Y(0) / Y(2) + c + 0.3 * U(g), f * Y(1) / Y(2) + c + 0.3 * U(g);
    if (i >= inlier_fraction * n) x2.col(i) << c + c * U(g), c + c * U(g);
  }
  using KernelType = robust::ACKernelAdaptor<fundamental::kernel::SevenPointSolver, fundamental::kernel::EpipolarDistanceError, UnnormalizerT, Mat3>;
  const KernelType kernel(x1, 1000, 1000, x2, 1000, 1000, true);
  std::vector<uint32_t> inliers;
  Mat3 F;
  auto t0 = Clock::now();
  const std::pair<double, double> out = robust::ACRANSAC(kernel, inliers, 2048, &F, 16.0);
  auto t1 = Clock::now();
  const double total_ms = us(t0, t1) / 1e3;
  std::vector<uint32_t> s = {1, 5, 9, 13, 17, 21, 25};
  std::vector<Mat3> models;
  size_t n_models = 0;
  t0 = Clock::now();
  for (int it = 0; it < 1000; ++it) { models.clear(); s[0] = (uint32_t)(it % 30 + 30) % n; kernel.Fit(s, &models); n_models += models.size(); }
  t1 = Clock::now();
  const double fit_us = us(t0, t1) / 1000;
  models.clear();
  kernel.Fit(s, &models);
  std::vector<double> res(n);
  double sink = 0;
  t0 = Clock::now();
  for (int it = 0; it < 1000 && !models.empty(); ++it) { kernel.Errors(models[0], res); sink += res[it % n]; }
  t1 = Clock::now();
  printf("{\"putative_matches\": %d, \"inlier_fraction\": %.2f, \"acransac_ms\": %.3f, \"inliers\": %zu, \"nfa\": %.1f, \"fit_us_per_sample\": %.2f, "
         "\"models_per_sample\": %.2f, \"errors_us_per_model\": %.2f, \"checksum\": %.3f}\n",
         n, inlier_fraction, total_ms, inliers.size(), out.second, fit_us, n_models / 1000.0, us(t0, t1) / 1000, sink);
}

int main() {
  one_case(100, 0.1);
  one_case(200, 0.5);
  one_case(500, 0.7);
  one_case(1000, 0.8);
  return 0;
}
