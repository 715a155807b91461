// geofilter_oracle.cpp - TEST INFRASTRUCTURE ONLY: a plain C++ restatement (no Eigen, no openMVG headers) of the reference's
// a-contrario fundamental-matrix filter, SURVEY.md 8(f) N2. Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of the
// bench may call it; the product (openmvg_amd/) never does.
//
// What is restated (paths under /root/reference/src/openMVG):
//   matching_image_collection/F_ACRobust.hpp:65-122            GeometricFilter_FMatrix_AC::Robust_estimation (kernel set-up, the
//                                                               "more than 2.5 x 7 inliers" acceptance, m_dPrecision_robust)
//   robust_estimation/robust_estimator_ACRansacKernelAdaptator.hpp:104-202   ACKernelAdaptor (normalisation by image size,
//                                                               logalpha0 of the point-to-line model, multError 0.5, unormalizeError)
//   multiview/conditioning.cpp:44-67,87-89                      PreconditionerFromPoints(w, h), UnnormalizerT
//   multiview/solver_fundamental_kernel.cpp:37-93,157-166       SevenPointSolver (null space of the 7 x 9 epipolar system, cubic in the
//                                                               pencil), EpipolarDistanceError
//   numeric/poly.h:32-96                                        SolveCubicPolynomial (Cardano / Viete)
//   robust_estimation/robust_estimator_ACRansac.hpp:58-119,196-262,339-489   log-combinatorial tables (float), the quantified NFA
//                                                               on a 20-bin histogram of the residuals, the ACRANSAC loop (max-consensus
//                                                               warm-up, early exit, focused sampling among the best inliers)
//   robust_estimation/rand_sampling.hpp:43-110                  the two UniformSample forms (rejection / Fisher-Yates)
//   third_party/histogram/histogram.hpp:47-112                  bin of a value, bin centres
//   std::mt19937 (default seed 5489) and libstdc++ 11's std::uniform_int_distribution<uint32_t> on a 32-bit engine
//   (bits/uniform_int_dist.h: Lemire's multiply-shift with rejection) - the sample sequence is part of the result.
//
// The homography model (port_geofilter_h_acransac) shares everything above but the kernel adaptor:
//   matching_image_collection/H_ACRobust.hpp:49-113             GeometricFilter_HMatrix_AC::Robust_estimation (point-to-point, 2.5 x 4)
//   multiview/solver_homography_kernel.cpp:37-93, .hpp:60-64     FourPointSolver (null vector of the 8 x 9 DLT system; the reference: last
//                                                               right singular vector, numeric/nullspace.cpp:14-32), AsymmetricError
//   robust_estimator_ACRansacKernelAdaptator.hpp:38-81          logalpha0 / multError of the point-to-point parametrisation
//   multiview/conditioning.cpp:80-82                            UnnormalizerI
// Pinned to the compiled reference and its stored outputs by tests/test_geofilter_h.py.
//
// One deliberate difference: the reference takes the two-dimensional null space of the 7 x 9 system from
// Eigen::SelfAdjointEigenSolver on A^T A (its two smallest eigenvectors); here it comes from Householder reflections of A^T (the
// last two columns of Q), the device code uses complete-pivoting elimination. The three bases span the same plane for a sample in
// general position, so the fundamental matrices of the pencil are the same up to scale and rounding - not bit for bit (they were
// not reproducible across Eigen builds either). Parity policy: identical inlier sets and F equal to 1e-6 after
// normalisation on fixtures whose decisive residuals are not within rounding of a histogram edge.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Mt19937 {   // std::mt19937
  uint32_t mt[624];
  int idx;
  explicit Mt19937(uint32_t seed = 5489u) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int i = 0; i < 624; ++i) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
};
// std::uniform_int_distribution<uint32_t>(a, b)(g) of libstdc++ 11 for a generator with a full 32-bit range
uint32_t uniform_u32(Mt19937& g, uint32_t a, uint32_t b) {
  const uint32_t urange = b - a;
  if (urange == 0xffffffffu) return g.next() + a;
  const uint32_t range = urange + 1;
  uint64_t product = (uint64_t)g.next() * range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) { product = (uint64_t)g.next() * range; low = (uint32_t)product; }
  }
  return (uint32_t)(product >> 32) + a;
}

float logcombi(uint32_t k, uint32_t n, const std::vector<float>& l10) {
  if (k >= n) return 0.f;
  if (n - k < k) k = n - k;
  float r = 0.f;
  for (uint32_t i = 1; i <= k; ++i) r += l10[n - i + 1] - l10[i];
  return r;
}

struct Model { double f[9]; };   // row-major 3 x 3

// null space of an R x 9 system (R = 7: seven-point, two vectors; R = 8: four-point DLT, one vector): A^T = Q R by Householder
// reflections, the last 9 - R columns of Q
template <int R>
void nullspace_rows(const double A[R][9], double (*out)[9]) {
  double M[9][R];   // A^T
  for (int r = 0; r < R; ++r) for (int c = 0; c < 9; ++c) M[c][r] = A[r][c];
  double V[R][9];   // Householder vectors
  for (int k = 0; k < R; ++k) {
    double norm = 0;
    for (int i = k; i < 9; ++i) norm += M[i][k] * M[i][k];
    norm = std::sqrt(norm);
    double v[9] = {0};
    for (int i = k; i < 9; ++i) v[i] = M[i][k];
    v[k] += (M[k][k] >= 0 ? norm : -norm);
    double vv = 0;
    for (int i = k; i < 9; ++i) vv += v[i] * v[i];
    for (int i = 0; i < 9; ++i) V[k][i] = v[i];
    if (vv == 0) continue;
    for (int j = k; j < R; ++j) {
      double dot = 0;
      for (int i = k; i < 9; ++i) dot += v[i] * M[i][j];
      const double s = 2 * dot / vv;
      for (int i = k; i < 9; ++i) M[i][j] -= s * v[i];
    }
  }
  for (int which = 0; which < 9 - R; ++which) {   // Q e_R .. Q e_8 with Q = H_0 H_1 ... H_(R-1)
    double q[9] = {0};
    q[R + which] = 1;
    for (int k = R - 1; k >= 0; --k) {
      double vv = 0, dot = 0;
      for (int i = 0; i < 9; ++i) { vv += V[k][i] * V[k][i]; dot += V[k][i] * q[i]; }
      if (vv == 0) continue;
      const double s = 2 * dot / vv;
      for (int i = 0; i < 9; ++i) q[i] -= s * V[k][i];
    }
    std::memcpy(out[which], q, sizeof(q));
  }
}
void nullspace7(const double A[7][9], double f1[9], double f2[9]) {
  double out[2][9];
  nullspace_rows<7>(A, out);
  std::memcpy(f1, out[0], sizeof(out[0])); std::memcpy(f2, out[1], sizeof(out[1]));
}

int solve_cubic(double a, double b, double c, double x[3]) {   // numeric/poly.h:32-75
  const double eps = std::numeric_limits<double>::epsilon();
  a /= 3;
  double p = (b - 3 * a * a) / 3;
  double q = (2 * a * a * a - a * b + c) / 2;
  double d = q * q + p * p * p;
  const double tolq = std::max(std::abs(2 * a * a * a), std::max(std::abs(a * b), std::abs(c)));
  const double tolp = std::max(std::abs(b), std::abs(3 * a * a));
  int n = (d > eps * std::max(p * p * tolp, std::abs(q) * tolq) ? 1 : 3);
  if (n == 1) {
    d = std::pow(std::abs(q) + std::sqrt(d), 1 / 3.0);
    x[0] = d - p / d;
    if (q > 0) x[0] = -x[0];
  } else {
    if (3 * p >= -eps * tolp) { n = 1; x[0] = 0; }
    else {
      p = std::sqrt(-p);
      q /= p * p * p;
      d = (q <= -1) ? M_PI : (q >= 1) ? 0 : std::acos(q);
      for (int i = 0; i < 3; ++i) x[i] = -2 * p * std::cos((d + 2 * M_PI * i) / 3);
    }
  }
  for (int i = 0; i < n; ++i) x[i] -= a;
  return n;
}

int seven_point(const double* x1, const double* x2, const uint32_t s[7], Model out[3]) {   // solver_fundamental_kernel.cpp:37-93
  double A[7][9];
  for (int r = 0; r < 7; ++r) {
    const double a1[3] = {x1[2 * s[r]], x1[2 * s[r] + 1], 1.0}, a2[3] = {x2[2 * s[r]], x2[2 * s[r] + 1], 1.0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[r][3 * i + j] = a2[i] * a1[j];
  }
  double F1[9], F2[9];
  nullspace7(A, F1, F2);
  const double a = F1[0], j = F2[0], b = F1[1], k = F2[1], c = F1[2], l = F2[2], d = F1[3], m = F2[3], e = F1[4], n = F2[4],
               f = F1[5], o = F2[5], g = F1[6], p = F2[6], h = F1[7], q = F2[7], i = F1[8], r = F2[8];
  const double P[4] = {
    a*e*i + b*f*g + c*d*h - a*f*h - b*d*i - c*e*g,
    a*e*r + a*i*n + b*f*p + b*g*o + c*d*q + c*h*m + d*h*l + e*i*j + f*g*k -
    a*f*q - a*h*o - b*d*r - b*i*m - c*e*p - c*g*n - d*i*k - e*g*l - f*h*j,
    a*n*r + b*o*p + c*m*q + d*l*q + e*j*r + f*k*p + g*k*o + h*l*m + i*j*n -
    a*o*q - b*m*r - c*n*p - d*k*r - e*l*p - f*j*q - g*l*n - h*j*o - i*k*m,
    j*n*r + k*o*p + l*m*q - j*o*q - k*m*r - l*n*p};
  if (P[0] == 0.0) return 0;   // poly.h:88-91
  double roots[3];
  const int nr = solve_cubic(P[2] / P[3], P[1] / P[3], P[0] / P[3], roots);
  for (int t = 0; t < nr; ++t)
    for (int u = 0; u < 9; ++u) out[t].f[u] = F1[u] + roots[t] * F2[u];
  return nr;
}

inline double epipolar_error(const Model& F, const double* x, const double* y) {   // EpipolarDistanceError, :157-166
  const double fx0 = F.f[0] * x[0] + F.f[1] * x[1] + F.f[2], fx1 = F.f[3] * x[0] + F.f[4] * x[1] + F.f[5],
               fx2 = F.f[6] * x[0] + F.f[7] * x[1] + F.f[8];
  const double dt = fx0 * y[0] + fx1 * y[1] + fx2;
  return dt * dt / (fx0 * fx0 + fx1 * fx1);
}

// FourPointSolver::Solve (multiview/solver_homography_kernel.cpp:37-93): the null vector of the 8 x 9 DLT system, row-major H
int four_point(const double* x1, const double* x2, const uint32_t s[4], Model out[1]) {
  double A[8][9];
  for (int i = 0; i < 4; ++i) {
    const double x = x1[2 * s[i]], y = x1[2 * s[i] + 1], u = x2[2 * s[i]], v = x2[2 * s[i] + 1];
    const double r0[9] = {x, y, 1.0, 0.0, 0.0, 0.0, -u * x, -u * y, -u}, r1[9] = {0.0, 0.0, 0.0, x, y, 1.0, -v * x, -v * y, -v};
    std::memcpy(A[2 * i], r0, sizeof(r0)); std::memcpy(A[2 * i + 1], r1, sizeof(r1));
  }
  double h[1][9];
  nullspace_rows<8>(A, h);
  std::memcpy(out[0].f, h[0], sizeof(h[0]));
  return 1;
}
inline double homography_error(const Model& H, const double* x, const double* y) {   // AsymmetricError, solver_homography_kernel.hpp:60-64
  const double v0 = (H.f[0] * x[0] + H.f[1] * x[1]) + H.f[2], v1 = (H.f[3] * x[0] + H.f[4] * x[1]) + H.f[5], v2 = (H.f[6] * x[0] + H.f[7] * x[1]) + H.f[8];
  const double dx = y[0] - v0 / v2, dy = y[1] - v1 / v2;
  return dx * dx + dy * dy;
}

struct Pair {
  uint32_t n;
  std::vector<double> x1, x2;   // normalised
  double n2_00, t2[3];          // N2(0,0) and the entries of N1 / N2 needed to unnormalise: T = [[s,0,tx],[0,s,ty],[0,0,1]]
  double n1_00, t1[3];
  double logalpha0, max_threshold;
};

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// homography = false: ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError, UnnormalizerT> (point to line);
// homography = true: ACKernelAdaptor<FourPointSolver, AsymmetricError, UnnormalizerI> configured point to point (H_ACRobust.hpp:77-87)
double port_acransac(bool homography, const double* xI, const double* xJ, const uint64_t* start, const uint32_t* wh, uint64_t n_pairs,
                     double precision, uint32_t max_iterations, uint8_t* inlier_mask, uint8_t* ok, double* Fout,
                     double* prec, double* nfa_out) {
  const uint32_t kMin = homography ? 4 : 7;                 // Solver::MINIMUM_SAMPLES
  const double max_models = homography ? 1.0 : 3.0;         // Solver::MAX_MODELS
  const double mult_error = homography ? 1.0 : 0.5;         // ACParametrizationHelper::MultError
  const double inf = std::numeric_limits<double>::infinity();
  for (uint64_t pp = 0; pp < n_pairs; ++pp) {
    const uint64_t lo = start[pp];
    const uint32_t n = (uint32_t)(start[pp + 1] - lo);
    std::memset(inlier_mask + lo, 0, n);
    ok[pp] = 0; prec[pp] = 0.0; nfa_out[pp] = 0.0;   // nData <= sizeSample: {0, 0} (ACRansac.hpp:354-355)
    for (int u = 0; u < 9; ++u) Fout[9 * pp + u] = (u % 4 == 0) ? 1.0 : 0.0;   // m_F = Identity
    if (n <= kMin) continue;
    // ---- ACKernelAdaptor: normalisation by the image sizes (conditioning.cpp:44-53) ----
    double T[2][3];   // {s, tx, ty} of image I / J
    for (int im = 0; im < 2; ++im) {
      const int w = (int)wh[4 * pp + 2 * im], h = (int)wh[4 * pp + 2 * im + 1];
      const double dNorm = 1.0 / std::sqrt(static_cast<double>(w * h));
      T[im][0] = dNorm; T[im][1] = -.5f * w * dNorm; T[im][2] = -.5 * h * dNorm;
    }
    std::vector<double> x1(2 * n), x2(2 * n);
    for (uint32_t i = 0; i < n; ++i) {
      x1[2 * i] = T[0][0] * xI[2 * (lo + i)] + T[0][1]; x1[2 * i + 1] = T[0][0] * xI[2 * (lo + i) + 1] + T[0][2];
      x2[2 * i] = T[1][0] * xJ[2 * (lo + i)] + T[1][1]; x2[2 * i + 1] = T[1][0] * xJ[2 * (lo + i) + 1] + T[1][2];
    }
    const int w2 = (int)wh[4 * pp + 2], h2 = (int)wh[4 * pp + 3];
    const double logalpha0 = homography ? std::log10(M_PI / (w2 * static_cast<double>(h2)) / (T[1][0] * T[1][0]))   // point to point
                                        : std::log10(2. * std::hypot(w2, h2) / (w2 * static_cast<double>(h2)) / T[1][0]);   // point to line
    const double upper = precision * precision;
    const bool quantified = upper != inf;
    if (!quantified) continue;   // the exhaustive NFA form (no precision bound) is not restated: main_GeometricFilter always passes one
    const double max_threshold = upper * T[1][0] * T[1][0];
    // ---- NFA_Interface: tables ----
    const double loge0 = std::log10(max_models * (n - kMin));
    std::vector<float> l10(n + 1), logc_n(n + 1), logc_k(n + 1);
    // (the reference calls the unqualified log10 on a float: with <cmath> of libstdc++ that is the C function on a double, rounded to float)
    for (uint32_t i = 0; i <= n; ++i) l10[i] = (float)::log10((double)static_cast<float>(i));
    for (uint32_t k = 0; k <= n; ++k) { logc_n[k] = logcombi(k, n, l10); logc_k[k] = logcombi(kMin, k, l10); }
    const int nBins = 20;
    const double bins_by_interval = nBins / (max_threshold - 0.0);
    double bin_value[20];
    { const double val = (max_threshold - 0.0) / static_cast<double>(nBins - 1);
      for (int i = 0; i < nBins; ++i) bin_value[i] = val * static_cast<double>(i) + 0.0; }
    // ---- ACRANSAC ----
    std::vector<uint32_t> vec_index(n), vec_sample(kMin), vec_inliers;
    for (uint32_t i = 0; i < n; ++i) vec_index[i] = i;
    std::vector<double> residuals(n);
    double minNFA = inf, errorMax = inf;
    Model best{};
    bool have_model = false;
    int nIterReserve = (int)(max_iterations / 10);
    unsigned nIter = max_iterations - nIterReserve;
    bool ac_mode = false;
    Mt19937 rng;
    for (unsigned iter = 0; iter < nIter && iter < max_iterations; ++iter) {
      if (ac_mode) {   // rand_sampling.hpp:84-110
        if (kMin <= vec_index.size()) {
          const uint32_t last = (uint32_t)vec_index.size() - 1;
          for (uint32_t i = 0; i < kMin; ++i) std::swap(vec_index[i], vec_index[uniform_u32(rng, i, last)]);
          for (uint32_t i = 0; i < kMin; ++i) vec_sample[i] = vec_index[i];
        }
      } else {   // :43-66
        vec_sample.clear();
        while (vec_sample.size() < kMin) {
          const uint32_t s = uniform_u32(rng, 0, n - 1);
          if (std::find(vec_sample.begin(), vec_sample.end(), s) == vec_sample.end()) vec_sample.push_back(s);
        }
      }
      Model models[3];
      const int nm = homography ? four_point(x1.data(), x2.data(), vec_sample.data(), models) : seven_point(x1.data(), x2.data(), vec_sample.data(), models);
      bool better = false;
      for (int mi = 0; mi < nm; ++mi) {
        for (uint32_t i = 0; i < n; ++i) residuals[i] = homography ? homography_error(models[mi], &x1[2 * i], &x2[2 * i]) : epipolar_error(models[mi], &x1[2 * i], &x2[2 * i]);
        if (!ac_mode) {
          unsigned nInlier = 0;
          for (uint32_t i = 0; i < n; ++i) nInlier += residuals[i] <= max_threshold;
          if (nInlier > 2.5 * kMin) ac_mode = true;
        }
        if (ac_mode) {   // ComputeNFA_and_inliers, quantified form (:196-262)
          size_t freq[20] = {0};
          for (uint32_t i = 0; i < n; ++i) {
            const double x = residuals[i];
            if (!(x < 0.0)) {
              const double t = (x - 0.0) * bins_by_interval;
              // static_cast<size_t> of a value out of range (or NaN) is undefined; x86-64's conversion yields 2^63, i.e. "overflow bin"
              const size_t b = (t >= 0.0 && t < 1.8e19) ? static_cast<size_t>(t) : (size_t)1 << 63;
              if (b < (size_t)nBins) ++freq[b];
            }
          }
          double cb_nfa = inf, cb_thr = 0.0;
          unsigned cum = 0;
          for (int bin = 0; bin < nBins; ++bin) {
            cum += (unsigned)freq[bin];
            if (cum > kMin && bin_value[bin] > std::numeric_limits<float>::epsilon()) {
              const double logalpha = logalpha0 + mult_error * std::log10(bin_value[bin] + std::numeric_limits<float>::epsilon());
              const double cur = loge0 + logalpha * (double)(cum - kMin) + logc_n[cum] + logc_k[cum];
              if (cur < cb_nfa && cur < 0) { cb_nfa = cur; cb_thr = bin_value[bin]; }
            }
          }
          if (cb_nfa < minNFA) {
            vec_inliers.clear();   // (updated even when the function then reports "not better": size <= MINIMUM_SAMPLES)
            for (uint32_t i = 0; i < n; ++i) if (residuals[i] <= cb_thr) vec_inliers.push_back(i);
            if (vec_inliers.size() > kMin) {
              better = true; minNFA = cb_nfa; errorMax = cb_thr; best = models[mi]; have_model = true;
            }
          }
        }
      }
      if (!ac_mode && iter > (unsigned)(nIterReserve * 2)) { nIter = 0; continue; }
      if (ac_mode && ((better && minNFA < 0) || ((iter + 1) == nIter && nIterReserve > 0))) {
        if (vec_inliers.empty()) { ++nIter; --nIterReserve; }
        else {
          vec_index = vec_inliers;
          if (nIterReserve) { nIter = iter + 1 + nIterReserve; nIterReserve = 0; }
        }
      }
    }
    if (minNFA >= 0) vec_inliers.clear();
    double Fm[9];
    for (int u = 0; u < 9; ++u) Fm[u] = have_model ? best.f[u] : ((u % 4 == 0) ? 1.0 : 0.0);
    if (!vec_inliers.empty()) {
      // Unnormalize: F = N2^T F N1 (conditioning.cpp:87-89), errorMax -> sqrt(errorMax) / N2(0,0)
      const double N1[9] = {T[0][0], 0, T[0][1], 0, T[0][0], T[0][2], 0, 0, 1}, N2[9] = {T[1][0], 0, T[1][1], 0, T[1][0], T[1][2], 0, 0, 1};
      double tmp[9], res[9];
      if (homography) {   // UnnormalizerI (conditioning.cpp:80-82): H = N2^-1 H N1
        const double is = 1.0 / T[1][0];
        const double N2i[9] = {is, 0, -T[1][1] * is, 0, is, -T[1][2] * is, 0, 0, 1};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += N2i[3 * r + k] * Fm[3 * k + c]; tmp[3 * r + c] = s; }
      } else
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += N2[3 * k + r] * Fm[3 * k + c]; tmp[3 * r + c] = s; }
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += tmp[3 * r + k] * N1[3 * k + c]; res[3 * r + c] = s; }
      std::memcpy(Fm, res, sizeof(res));
      errorMax = std::sqrt(errorMax) / T[1][0];
    }
    const bool good = vec_inliers.size() > kMin * 2.5;
    ok[pp] = good;
    prec[pp] = errorMax; nfa_out[pp] = minNFA;
    std::memcpy(Fout + 9 * pp, Fm, sizeof(Fm));
    if (good) for (uint32_t idx : vec_inliers) inlier_mask[lo + idx] = 1;
  }
  return 0.0;
}
}  // namespace

extern "C" {
// same interface as ref_geofilter_f_acransac / ref_geofilter_h_acransac (oracle/ref_shim_geofilter.cpp), one thread
double port_geofilter_f_acransac(const double* xI, const double* xJ, const uint64_t* start, const uint32_t* wh, uint64_t n_pairs,
                                 double precision, uint32_t max_iterations, uint8_t* inlier_mask, uint8_t* ok, double* Fout,
                                 double* prec, double* nfa_out) {
  return port_acransac(false, xI, xJ, start, wh, n_pairs, precision, max_iterations, inlier_mask, ok, Fout, prec, nfa_out);
}
double port_geofilter_h_acransac(const double* xI, const double* xJ, const uint64_t* start, const uint32_t* wh, uint64_t n_pairs,
                                 double precision, uint32_t max_iterations, uint8_t* inlier_mask, uint8_t* ok, double* Fout,
                                 double* prec, double* nfa_out) {
  return port_acransac(true, xI, xJ, start, wh, n_pairs, precision, max_iterations, inlier_mask, ok, Fout, prec, nfa_out);
}
}  // extern "C"
