// measurement: stage clocks of the five-point solver (one wave per sample, many waves) - hipcc -DMVGX_FIVE_POINT_STAMPS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
namespace {
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ double shfl_f64(double v, int src) { return __shfl(v, src); }
__device__ __forceinline__ double lane_value_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(r0, r1), max(r2, r3));
}
#include "geofilter_five_point.h"
#include "geofilter_five_point_x4.h"
// four samples per wave (solve4): a quarter of the waves
__global__ __launch_bounds__(256, 2) void k4(const double* b1, const double* b2, int n_samples, int* n_out) {
  __shared__ double scr[4][4 * five_point::kScratch + 4 * 90];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + wave;
  if (4 * w >= n_samples) return;
  const int mine = 4 * w + (lane >> 4) < n_samples ? 4 * w + (lane >> 4) : n_samples - 1;
  const uint32_t s[5] = {5u * mine, 5u * mine + 1, 5u * mine + 2, 5u * mine + 3, 5u * mine + 4};
  const int n = five_point::solve4(b1, b2, s, lane, scr[wave], scr[wave] + 4 * five_point::kScratch);
  if ((lane & 15) == 0 && 4 * w + (lane >> 4) < n_samples) n_out[4 * w + (lane >> 4)] = n;
}
__global__ __launch_bounds__(256, 2) void k(const double* b1, const double* b2, int n_samples, int* n_out) {
  __shared__ double scr[4][five_point::kScratch + 90];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + wave;
  if (w >= n_samples) return;
  const uint32_t s[7] = {0, 1, 2, 3, 4, 0, 0};
  const int n = five_point::solve(b1 + 15 * (size_t)w, b2 + 15 * (size_t)w, s, lane, scr[wave], scr[wave] + five_point::kScratch);
  if (lane == 0) n_out[w] = n;
}
}
int main() {
  const int N = 200000;
  std::mt19937_64 g(1);
  std::normal_distribution<double> nd;
  std::vector<double> b1(15 * (size_t)N), b2(15 * (size_t)N);
  for (int w = 0; w < N; ++w) {
    double t[3] = {nd(g), nd(g), nd(g)}, th = 0.2;
    for (int i = 0; i < 5; ++i) {
      double X[3] = {nd(g) * 0.5, nd(g) * 0.5, 4 + nd(g) * 0.3};
      double Y[3] = {std::cos(th) * X[0] + std::sin(th) * X[2] + t[0] * 0.3, X[1] + t[1] * 0.3, -std::sin(th) * X[0] + std::cos(th) * X[2] + t[2] * 0.1};
      double n1 = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]), n2 = std::sqrt(Y[0] * Y[0] + Y[1] * Y[1] + Y[2] * Y[2]);
      for (int k2 = 0; k2 < 3; ++k2) { b1[15 * (size_t)w + 3 * i + k2] = X[k2] / n1; b2[15 * (size_t)w + 3 * i + k2] = Y[k2] / n2; }
    }
  }
  double *d1, *d2; int* dn;
  hipMalloc(&d1, b1.size() * 8); hipMalloc(&d2, b2.size() * 8); hipMalloc(&dn, N * 4);
  hipMemcpy(d1, b1.data(), b1.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d2, b2.data(), b2.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    const bool x4 = rep >= 2;
    unsigned long long z[12] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(five_point::g_stamps), z, sizeof(z));
    hipEventRecord(e0);
    if (x4) hipLaunchKernelGGL(k4, dim3((N / 4 + 3) / 4), dim3(256), 0, 0, d1, d2, N, dn);
    else hipLaunchKernelGGL(k, dim3((N + 3) / 4), dim3(256), 0, 0, d1, d2, N, dn);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[12]; hipMemcpyFromSymbol(st, HIP_SYMBOL(five_point::g_stamps), sizeof(st));
    std::vector<int> n(N); hipMemcpy(n.data(), dn, N * 4, hipMemcpyDeviceToHost);
    double mean = 0; for (int v : n) mean += v;
    const unsigned long long D = x4 ? N / 4 : N;   // waves that stamped
#ifdef MVGX_FIVE_POINT_COUNT_ROUNDS
    { unsigned long long r = 0, sv = 0;
      hipMemcpyFromSymbol(&r, HIP_SYMBOL(five_point::g_aberth_rounds), sizeof(r)); hipMemcpyFromSymbol(&sv, HIP_SYMBOL(five_point::g_aberth_solves), sizeof(sv));
      unsigned long long c[8]; hipMemcpyFromSymbol(c, HIP_SYMBOL(five_point::g_fallback_cause), sizeof(c));
      printf("rows left to hqr by cause (cumulative): leading coefficient %llu | subdiagonal %llu | coefficients %llu | no convergence %llu | polish %llu | coincident %llu\n", c[0], c[1], c[2], c[3], c[4], c[5]);
      printf("Aberth rounds per solve (cumulative over the launches): %.2f\n", sv ? (double)r / (double)sv : 0.0); }
#endif
    printf("%s: %d solves in %.2f ms = %.0f ns per solve per device; mean models %.2f; clocks per solve: nullspace %llu | constraints %llu | gauss-jordan %llu | hessenberg %llu | hqr %llu | eigenvectors %llu\n",
           x4 ? "solve4 (clocks per batch of four)" : "solve", N, ms, ms * 1e6 / N, mean / N, st[0] / D, st[1] / D, st[2] / D, st[3] / D, st[4] / D, st[5] / D);
    if (x4) printf("  of the eigenvalue stage: polynomial + start radii %llu | iteration %llu | polish on the matrix %llu | coincident roots, hand-over %llu\n", st[6] / D, st[7] / D, st[8] / D, st[9] / D);
  }
  return 0;
}
