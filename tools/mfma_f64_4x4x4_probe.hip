// mfma_f64_4x4x4_probe - v_mfma_f64_4x4x4_4b_f64 on gfx950: how long it holds the pipe, and where its operands and results live.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_f64_4x4x4_probe tools/mfma_f64_4x4x4_probe.hip && tools/_build/mfma_f64_4x4x4_probe
// Layout probe: A = 1000 (lane + 1) in exactly one lane la (else 0), B = 1 in exactly one lane lb (else 0): the lanes where D != 0 tell which
// (A lane, B lane) pairs meet and where their product lands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void layout_kernel(double* out /* [64 la][64 lb][64 lanes] */) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1000.0 * (la + 1) : 0.0, b = lane == lb ? 1.0 : 0.0;
      double d = 0.0;
      asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n s_nop 15" : "+v"(d) : "v"(a), "v"(b));
      out[((size_t)la * 64 + lb) * 64 + lane] = d;
    }
}
__global__ void timing_kernel(long long* cyc, double* sink, int iters) {
  double d[8]; for (int k = 0; k < 8; ++k) d[k] = 0.0;
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(d[k]) : "v"(a), "v"(b));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for (int k = 0; k < 8; ++k) s += d[k];
  if (s == 1.2345e300) sink[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* d_out; CHECK(hipMalloc(&d_out, 64 * 64 * 64 * sizeof(double)));
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, d_out);
  std::vector<double> h(64 * 64 * 64);
  CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(double), hipMemcpyDeviceToHost));
  // for every A lane: which B lanes it meets, and the D lane of each product
  printf("A lane -> (B lane : D lane) pairs\n");
  for (int la = 0; la < 64; ++la) {
    printf("A %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" (%d:%d)", lb, l);
    printf("\n");
  }
  long long* d_cyc; double* d_sink; CHECK(hipMalloc(&d_cyc, 64 * sizeof(long long))); CHECK(hipMalloc(&d_sink, 64));
  const int iters = 4000;
  hipLaunchKernelGGL(timing_kernel, dim3(1), dim3(64), 0, 0, d_cyc, d_sink, iters);
  hipLaunchKernelGGL(timing_kernel, dim3(1), dim3(64), 0, 0, d_cyc, d_sink, iters);
  long long c; CHECK(hipMemcpy(&c, d_cyc, sizeof(c), hipMemcpyDeviceToHost));
  printf("v_mfma_f64_4x4x4_4b_f64: %.2f s_memtime ticks per instruction (8 independent accumulators, one wave)\n", (double)c / (iters * 8.0));
  return 0;
}
