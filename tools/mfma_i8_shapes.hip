// mfma_i8_shapes - the A/B VERDICT r4 item 4 asks for before anybody builds a 16x16x64 tile layout for the matching filter:
// sustained MFMA-only streams of v_mfma_i32_32x32x32_i8 and v_mfma_i32_16x16x64_i8 on REAL int8 operands (SIFT-like bytes - 128, the
// values the filter kernel multiplies; zero operands do not reach the power cap and say nothing about the clocks the kernel sees),
// long enough (>= 100 ms per run) for the power management to settle. Both shapes do 1 024 MAC per lane-instruction-cycle at peak:
//   32x32x32: 32 768 MAC per instruction, 16 passes (64 cycles at 4 cycles / pass is the documented issue cost of 8 passes x 2 ... measured below)
//   16x16x64: 16 384 MAC per instruction, 8 passes
// What is reported per shape and waves / SIMD: wall-clock TOPS (2 x MAC / s over the whole device) and shader cycles per MFMA per SIMD
// (s_memtime is a constant 100 MHz clock on gfx950, so the second figure is wall time as well - kept for the comparison with tools/ubench.hip).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_i8_shapes tools/mfma_i8_shapes.hip && tools/_build/mfma_i8_shapes
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kOperands = 8;   // distinct A and B fragments a wave cycles through (registers: 2 x 8 x 4)

// SHAPE 32: four independent 32x32 accumulators (64 registers, what the filter kernel holds per wave); SHAPE 16: eight independent
// 16x16 accumulators (32 registers - the "half the accumulator registers" of the proposal) or sixteen (same registers as SHAPE 32)
template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void stream_kernel(const v4i* __restrict__ data, int iters, int* sink) {
  v4i a[kOperands], b[kOperands];
  const int lane_id = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < kOperands; ++i) {
    a[i] = data[(size_t)(2 * i) * 65536 + (lane_id & 65535)];
    b[i] = data[(size_t)(2 * i + 1) * 65536 + (lane_id & 65535)];
  }
  int total = 0;
  if constexpr (SHAPE == 32) {
    v16i acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int o = 0; o < kOperands; ++o)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
          asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(o + i) % kOperands]), "v"(b[o]));
    }
    for (int i = 0; i < NACC; ++i) for (int k = 0; k < 16; ++k) total += acc[i][k];
  } else {
    v4i acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int o = 0; o < kOperands; ++o)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
          asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(o + i) % kOperands]), "v"(b[o]));
    }
    for (int i = 0; i < NACC; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  if (total == 0x7fffffff) sink[0] = total;
}

template <int SHAPE, int NACC>
static void run(const char* name, const v4i* data, int* sink, int blocks_per_cu, double target_ms, FILE* json, bool zero) {
  const int nblk = 256 * blocks_per_cu;
  const double mac_per_mfma = SHAPE == 32 ? 32768.0 : 16384.0;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int iters = 2000;
  float ms = 0;
  for (int round = 0; round < 3; ++round) {   // calibrate the loop count, then the measured run (the last, longest one)
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream_kernel<SHAPE, NACC>), dim3(nblk), dim3(256), 0, 0, data, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (round < 2) iters = (int)(iters * (round == 0 ? target_ms / 4 : target_ms) / ms) + 1;
  }
  const double mfma = (double)nblk * 4 * iters * kOperands * NACC;
  const double tops = 2.0 * mfma * mac_per_mfma / (ms * 1e-3) / 1e12;
  printf("%-34s %s waves/SIMD %d: %8.1f ms, %7.0f TOPS = %.3f of 5000\n", name, zero ? "zero operands" : "real operands", blocks_per_cu, ms, tops, tops / 5000.0);
  if (json)
    fprintf(json, "{\"shape\": \"%s\", \"accumulators_per_wave\": %d, \"operands\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.2f, \"tops\": %.1f, \"frac_of_5000\": %.4f}\n",
            name, NACC, zero ? "zero" : "real", blocks_per_cu, ms, tops, tops / 5000.0);
}

int main(int argc, char** argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 150.0;
  FILE* json = argc > 2 ? fopen(argv[2], "w") : nullptr;
  // SIFT-like bytes: most components small, a few large (the synthetic descriptors of openmvg_amd/synth.py have the same histogram shape), - 128
  std::vector<int8_t> h((size_t)2 * kOperands * 65536 * 16);
  std::mt19937 rng(12345);
  std::exponential_distribution<float> ex(1.0f / 28.0f);
  for (auto& v : h) { const int u = std::min(255, (int)ex(rng)); v = (int8_t)(u - 128); }
  v4i *d_real, *d_zero;
  int* sink;
  CHECK(hipMalloc(&d_real, h.size()));
  CHECK(hipMalloc(&d_zero, h.size()));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemcpy(d_real, h.data(), h.size(), hipMemcpyHostToDevice));
  CHECK(hipMemset(d_zero, 0, h.size()));
  for (int rep = 0; rep < 2; ++rep) {   // twice, alternating: the device warms up during the first round
    for (int bpc : {1, 2}) {
      run<32, 4>("v_mfma_i32_32x32x32_i8 x4 acc", d_real, sink, bpc, target_ms, rep ? json : nullptr, false);
      run<16, 8>("v_mfma_i32_16x16x64_i8 x8 acc", d_real, sink, bpc, target_ms, rep ? json : nullptr, false);
      run<16, 16>("v_mfma_i32_16x16x64_i8 x16 acc", d_real, sink, bpc, target_ms, rep ? json : nullptr, false);
    }
    if (rep == 0) continue;
    run<32, 4>("v_mfma_i32_32x32x32_i8 x4 acc", d_zero, sink, 2, target_ms, json, true);
    run<16, 8>("v_mfma_i32_16x16x64_i8 x8 acc", d_zero, sink, 2, target_ms, json, true);
    run<16, 8>("v_mfma_i32_16x16x64_i8 x8 acc", d_real, sink, 3, target_ms, json, false);   // the third wave per SIMD the smaller accumulators would allow
  }
  if (json) fclose(json);
  return 0;
}
