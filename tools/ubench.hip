// ubench — instruction-rate microbenchmarks that size the matching kernel's epilogue on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench tools/ubench.hip && tools/_build/ubench
// Every kernel times its own loop with s_memtime (shader cycles) in wave 0 of each block and the host takes the
// median over blocks, so the numbers are per-SIMD issue costs at the stated waves/SIMD, independent of DVFS.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 2000;

enum Op { kMax = 0, kMed3, kLshlAdd, kMax3, kPkMaxI16, kMin, kNumOps };
static const char* kOpName[] = {"v_max_i32", "v_med3_i32", "v_lshl_add_u32", "v_max3_i32", "v_pk_max_i16", "v_min_i32"};

template <int OP>
__device__ __forceinline__ void valu8(int (&x)[8], int y, int z) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (OP == kMax) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[k]) : "v"(y));
    if (OP == kMed3) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));
    if (OP == kLshlAdd) asm volatile("v_lshl_add_u32 %0, %0, 9, %1" : "+v"(x[k]) : "v"(y));
    if (OP == kMax3) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));
    if (OP == kPkMaxI16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[k]) : "v"(y));
    if (OP == kMin) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x[k]) : "v"(y));
  }
}

// pure VALU: 64 ops per iteration, 8 independent chains
template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(long long* cyc, int* sink) {
  int x[8];
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 7 + k;
  const int y = threadIdx.x ^ 0x55, z = threadIdx.x + 3;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) valu8<OP>(x, y, z);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int k = 0; k < 8; ++k) s += x[k];
  if (s == 0x7fffffff) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// MFMA i8 32x32x32: NCHAIN dependent MFMAs per accumulator, two accumulators ping-pong; KV VALU ops (v_max_i32 on the
// OTHER accumulator's registers, like a software-pipelined epilogue) spread after each MFMA.
template <int NCHAIN, int KV, int OP>
__global__ __launch_bounds__(256) void mix_kernel(long long* cyc, int* sink) {
  v16i accA, accB;
  for (int k = 0; k < 16; ++k) { accA[k] = k; accB[k] = -k; }
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  int t[16];
  for (int k = 0; k < 16; ++k) t[k] = -1000000 + k;
  const int z = threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      v16i& acc = half ? accB : accA;
      v16i& other = half ? accA : accB;
      constexpr int per = (KV + NCHAIN - 1) / NCHAIN;
      int done = 0;
#pragma unroll
      for (int c = 0; c < NCHAIN; ++c) {
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
        for (int v = 0; v < per; ++v) {
          if (done < KV) {
            const int r = done & 15;
            if (OP == kMax) asm volatile("v_max_i32 %0, %0, %1" : "+v"(t[r]) : "v"(other[r]));
            if (OP == kMed3) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(t[r]) : "v"(other[r]), "v"(z));
            if (OP == kLshlAdd) asm volatile("v_lshl_add_u32 %0, %1, 9, %0" : "+v"(t[r]) : "v"(other[r]));
            if (OP == kMax3) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(t[r]) : "v"(other[r]), "v"(other[(r + 1) & 15]));
            ++done;
          }
        }
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int k = 0; k < 16; ++k) s += t[k] + accA[k] + accB[k];
  if (s == 0x7fffffff) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// MFMA 16x16x64 i8 variant (4 accumulator regs), for the rate comparison
template <int NIND>
__global__ __launch_bounds__(256) void mfma16_kernel(long long* cyc, int* sink) {
  v4i acc[NIND];
  for (int i = 0; i < NIND; ++i) acc[i] = v4i{i, 1, 2, 3};
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NIND; ++i) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int i = 0; i < NIND; ++i) s += acc[i][0];
  if (s == 0x7fffffff) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NIND>
__global__ __launch_bounds__(256) void mfma32_kernel(long long* cyc, int* sink) {
  v16i acc[NIND];
  for (int i = 0; i < NIND; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = i + k;
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NIND; ++i) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int i = 0; i < NIND; ++i) s += acc[i][0];
  if (s == 0x7fffffff) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static long long* d_cyc;
static int* d_sink;

template <typename F>
static double run(F launch, int blocks_per_cu, double* wall_ms) {
  const int nblk = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(nblk);  // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch(nblk);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  *wall_ms = ms;
  std::vector<long long> h(nblk * 4);
  CHECK(hipMemcpy(h.data(), d_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  return (double)h[h.size() / 2];
}

#define RUN_VALU(OP) for (int bpc : {1, 2, 4}) { double w; \
    double c = run([&](int n) { hipLaunchKernelGGL(valu_kernel<OP>, dim3(n), dim3(256), 0, 0, d_cyc, d_sink); }, bpc, &w); \
    printf("valu %-16s waves/SIMD %d : %.2f cyc/op/wave  -> %.2f cyc/op/SIMD  (wall %.3f ms)\n", kOpName[OP], bpc, c / (kIters * 64.0), c / (kIters * 64.0) / bpc, w); }

#define RUN_MIX(NC, KV, OP) for (int bpc : {1, 2}) { double w; \
    double c = run([&](int n) { hipLaunchKernelGGL((mix_kernel<NC, KV, OP>), dim3(n), dim3(256), 0, 0, d_cyc, d_sink); }, bpc, &w); \
    printf("mix chain %d + %2d x %-14s waves/SIMD %d : %.1f cyc per (chain+valu) per wave -> MFMA pipe util %.1f%%  (wall %.3f ms)\n", NC, KV, kOpName[OP], bpc, \
           c / (kIters * 2.0), 100.0 * (NC * 32.0 * bpc) / (c / (kIters * 2.0)), w); }

int main() {
  CHECK(hipMalloc(&d_cyc, 256 * 8 * 4 * sizeof(long long)));
  CHECK(hipMalloc(&d_sink, 64));
  RUN_VALU(kMax) RUN_VALU(kMed3) RUN_VALU(kLshlAdd) RUN_VALU(kMax3) RUN_VALU(kPkMaxI16)
  for (int bpc : {1, 2}) { double w;
    double c = run([&](int n) { hipLaunchKernelGGL(mfma32_kernel<4>, dim3(n), dim3(256), 0, 0, d_cyc, d_sink); }, bpc, &w);
    const double ops = (double)256 * bpc * 4 * kIters * 32 * 65536.0;
    printf("mfma_i32_32x32x32_i8 x4 indep, waves/SIMD %d: %.1f cyc/mfma/wave, %.1f cyc/mfma/SIMD, wall %.3f ms = %.0f TOPS\n", bpc, c / (kIters * 32.0), c / (kIters * 32.0) / bpc, w, ops / (w * 1e-3) / 1e12); }
  for (int bpc : {1, 2}) { double w;
    double c = run([&](int n) { hipLaunchKernelGGL(mfma32_kernel<1>, dim3(n), dim3(256), 0, 0, d_cyc, d_sink); }, bpc, &w);
    printf("mfma_i32_32x32x32_i8 dependent chain, waves/SIMD %d: %.1f cyc/mfma/wave (wall %.3f ms)\n", bpc, c / (kIters * 8.0), w); }
  for (int bpc : {1, 2}) { double w;
    double c = run([&](int n) { hipLaunchKernelGGL(mfma16_kernel<4>, dim3(n), dim3(256), 0, 0, d_cyc, d_sink); }, bpc, &w);
    const double ops = (double)256 * bpc * 4 * kIters * 32 * 32768.0;
    printf("mfma_i32_16x16x64_i8 x4 indep, waves/SIMD %d: %.1f cyc/mfma/wave, wall %.3f ms = %.0f TOPS\n", bpc, c / (kIters * 32.0), w, ops / (w * 1e-3) / 1e12); }
  RUN_MIX(4, 0, kMax) RUN_MIX(4, 16, kMax) RUN_MIX(4, 24, kMax) RUN_MIX(4, 32, kMax) RUN_MIX(4, 48, kMax) RUN_MIX(4, 64, kMax)
  RUN_MIX(4, 16, kMax3) RUN_MIX(4, 32, kMed3) RUN_MIX(4, 48, kMed3)
  return 0;
}
