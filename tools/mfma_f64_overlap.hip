// mfma_f64_overlap — does v_mfma_f64_16x16x4_f64 run BESIDE fp64 vector arithmetic on gfx950, or do the two share the fp64 pipe?
// (round 6: the question behind fusing the Z^T Z of a point group with the evaluation of the next group's observations)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_f64_overlap tools/mfma_f64_overlap.hip && tools/_build/mfma_f64_overlap
// One workgroup per CU-sized slot, WAVES waves per SIMD (blockDim = 256 * WAVES); every wave times its own loop with s_memtime and
// the wall time of the launch gives the clock. Kernels:
//   mfma_only   : 4 independent accumulators, 4 MFMAs per iteration
//   valu_only   : KV v_fma_f64 per iteration on 8 independent chains (or v_fma_f32 / v_add_u32)
//   mixed       : 1 MFMA then KV/4 vector operations, four times per iteration (same wave)
//   split       : waves 0..3 (one per SIMD) run mfma_only, waves 4..7 run valu_only (two waves per SIMD, different pipes)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 4000;
enum VOp { kFma64 = 0, kFma32 = 1, kAddU32 = 2, kMul64 = 3 };

template <int OP>
__device__ __forceinline__ void vop(double (&x)[8], float (&f)[8], unsigned (&u)[8], int k, double y, float yf, unsigned yu) {
  if (OP == kFma64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[k]) : "v"(y));
  if (OP == kMul64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(y));
  if (OP == kFma32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k]) : "v"(yf));
  if (OP == kAddU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(yu));
}

// MODE 0: mfma only, 1: valu only, 2: mixed in one wave, 3: split by wave (wave index >= 4: valu, else mfma)
template <int MODE, int KV, int OP>
__global__ void probe_kernel(long long* cyc, double* sink) {
  d4 acc[4];
  for (int j = 0; j < 4; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
  double x[8]; float f[8]; unsigned u[8];
  for (int k = 0; k < 8; ++k) { x[k] = 1.0 + 1e-9 * (threadIdx.x + k); f[k] = 1.0f + 1e-6f * k; u[k] = threadIdx.x + k; }
  const double a = 1.0 + 1e-12 * threadIdx.x, b = 1.0 - 1e-12 * threadIdx.x, y = 1.0 - 1e-13;
  const float yf = 0.999999f; const unsigned yu = 3;
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 2) {
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
        for (int k = 0; k < KV / 4; ++k) vop<OP>(x, f, u, (j * (KV / 4) + k) & 7, y, yf, yu);
      }
    }
  } else if (do_mfma) {
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
    }
  } else if (do_valu) {
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
      for (int k = 0; k < KV; ++k) vop<OP>(x, f, u, k & 7, y, yf, yu);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  for (int k = 0; k < 8; ++k) s += x[k] + f[k] + u[k];
  if (s == 1.2345e300) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE, int KV, int OP>
static void run(const char* name, int waves_per_simd, long long* d_cyc, double* d_sink, int n_cu) {
  const int threads = 256 * waves_per_simd;
  std::vector<long long> h(n_cu * 16, 0);
  CHECK(hipMemset(d_cyc, 0, h.size() * sizeof(long long)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe_kernel<MODE, KV, OP>), dim3(n_cu), dim3(threads), 0, 0, d_cyc, d_sink);   // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe_kernel<MODE, KV, OP>), dim3(n_cu), dim3(threads), 0, 0, d_cyc, d_sink);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(h.data(), d_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  std::vector<long long> lo, hi;   // waves 0..3 and 4..7
  for (int b = 0; b < n_cu; ++b) for (int w = 0; w < threads / 64; ++w) (w < 4 ? lo : hi).push_back(h[b * 16 + w]);
  std::sort(lo.begin(), lo.end()); std::sort(hi.begin(), hi.end());
  const double us = ms * 1e3;
  printf("%-58s waves/SIMD %d  wall %8.1f us  per iteration: %7.1f ns", name, waves_per_simd, us, us * 1e3 / kIters);
  printf("  s_memtime ticks/iter waves0-3 %.2f", (double)lo[lo.size() / 2] / kIters);
  if (!hi.empty()) printf("  waves4-7 %.2f", (double)hi[hi.size() / 2] / kIters);
  printf("\n");
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %.0f MHz; every line: one workgroup per CU, %d iterations; an iteration = 4 MFMA (f64 16x16x4) and / or KV vector ops\n",
         prop.name, n_cu, prop.clockRate / 1e3, kIters);
  long long* d_cyc; double* d_sink;
  CHECK(hipMalloc(&d_cyc, n_cu * 16 * sizeof(long long))); CHECK(hipMalloc(&d_sink, 64));
  run<0, 0, kFma64>("mfma only (4 per iteration)", 1, d_cyc, d_sink, n_cu);
  run<0, 0, kFma64>("mfma only (4 per iteration)", 2, d_cyc, d_sink, n_cu);
  run<1, 32, kFma64>("v_fma_f64 only (32 per iteration)", 1, d_cyc, d_sink, n_cu);
  run<1, 32, kFma64>("v_fma_f64 only (32 per iteration)", 2, d_cyc, d_sink, n_cu);
  run<1, 32, kMul64>("v_mul_f64 only (32 per iteration)", 1, d_cyc, d_sink, n_cu);
  run<1, 32, kFma32>("v_fma_f32 only (32 per iteration)", 1, d_cyc, d_sink, n_cu);
  run<1, 32, kAddU32>("v_add_u32 only (32 per iteration)", 1, d_cyc, d_sink, n_cu);
  run<2, 16, kFma64>("same wave: 4 x (mfma + 4 v_fma_f64)", 1, d_cyc, d_sink, n_cu);
  run<2, 32, kFma64>("same wave: 4 x (mfma + 8 v_fma_f64)", 1, d_cyc, d_sink, n_cu);
  run<2, 48, kFma64>("same wave: 4 x (mfma + 12 v_fma_f64)", 1, d_cyc, d_sink, n_cu);
  run<2, 32, kFma32>("same wave: 4 x (mfma + 8 v_fma_f32)", 1, d_cyc, d_sink, n_cu);
  run<2, 48, kFma32>("same wave: 4 x (mfma + 12 v_fma_f32)", 1, d_cyc, d_sink, n_cu);
  run<2, 32, kAddU32>("same wave: 4 x (mfma + 8 v_add_u32)", 1, d_cyc, d_sink, n_cu);
  run<2, 48, kAddU32>("same wave: 4 x (mfma + 12 v_add_u32)", 1, d_cyc, d_sink, n_cu);
  run<3, 32, kFma64>("two waves per SIMD: one mfma, one 32 v_fma_f64", 2, d_cyc, d_sink, n_cu);
  run<3, 48, kFma64>("two waves per SIMD: one mfma, one 48 v_fma_f64", 2, d_cyc, d_sink, n_cu);
  run<3, 48, kFma32>("two waves per SIMD: one mfma, one 48 v_fma_f32", 2, d_cyc, d_sink, n_cu);
  run<3, 48, kAddU32>("two waves per SIMD: one mfma, one 48 v_add_u32", 2, d_cyc, d_sink, n_cu);
  return 0;
}
