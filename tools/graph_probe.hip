// measurement: a dependent chain of N small kernels - plain stream launches against the same chain captured once into a hipGraph and
// replayed (what an LM iteration of the BA solver would gain from graphs: its ~60 launches are such a chain).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void step_kernel(double* buf, int i) { if (threadIdx.x < 64) buf[i * 64 + threadIdx.x] = buf[(i - 1) * 64 + threadIdx.x] + 1.0; }
int main() {
  const int N = 64;
  double* buf;
  (void)hipMalloc(&buf, (N + 2) * 64 * 8);
  (void)hipMemset(buf, 0, (N + 2) * 64 * 8);
  hipStream_t s; (void)hipStreamCreate(&s);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float ms;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(a, s);
    for (int i = 1; i <= N; ++i) hipLaunchKernelGGL(step_kernel, dim3(1), dim3(256), 0, s, buf, i);
    (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
    printf("stream: %d launches, %.2f us per step\n", N, ms * 1e3 / N);
  }
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 1; i <= N; ++i) hipLaunchKernelGGL(step_kernel, dim3(1), dim3(256), 0, s, buf, i);
  if (hipStreamEndCapture(s, &g) != hipSuccess || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("graph capture failed\n"); return 1; }
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(a, s);
    (void)hipGraphLaunch(ge, s);
    (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
    printf("graph:  %d kernel nodes, %.2f us per step\n", N, ms * 1e3 / N);
  }
  return 0;
}
