// xcd_handoff — what a hand-over of a tile between two workgroups costs on gfx950 when both sit on the SAME XCD (one L2) and the data
// moves with agent-scope RELAXED accesses (sc1: miss the per-CU cache, meet in the L2; no buffer_wbl2 / buffer_inv), against the
// device-scope release / acquire pair the compiler emits (L2 write-back + invalidate: needed only ACROSS XCDs).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/xcd_handoff tools/xcd_handoff.hip && tools/_build/xcd_handoff
// Grid: 8 K workgroups of 256 threads; the workgroups with blockIdx % 8 == X take part (K participants), the others leave at once.
// A token goes round the participants R times; a participant waits for its turn, reads the 4 KB tile the one before wrote, adds 1,
// writes it, passes the token on. Every participant records the XCC_ID register it ran on.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kTile = 512;   // doubles handed over (two per thread)

// MODE 0: relaxed agent-scope accesses + s_waitcnt before the flag (same-XCD form); 1: release / acquire at agent scope (compiler's fences)
template <int MODE>
__global__ __launch_bounds__(256) void ring_kernel(int K, int R, int xcd, unsigned* flag, double* tile, unsigned* xcc_of, long long* cyc, int stride8) {
  const int b = blockIdx.x;
  if (stride8 ? (b % 8 != xcd) : (b >= K)) return;
  const int p = stride8 ? b / 8 : b;
  if (threadIdx.x == 0) xcc_of[p] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < R; ++r) {
    const unsigned my_turn = (unsigned)(r * K + p);
    if (threadIdx.x == 0) {   // (bounded: a stale line must not hang the device - the result check then reports WRONG)
      int polls = 0;
      if (MODE == 0) { while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != my_turn && ++polls < 200000) __builtin_amdgcn_s_sleep(1); }
      else { while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != my_turn && ++polls < 200000) __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    double v[2];
    for (int k = 0; k < 2; ++k) {
      double* q = tile + threadIdx.x + 256 * k;
      v[k] = MODE == 0 ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
      v[k] += 1.0;
      if (MODE == 0) __hip_atomic_store(q, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *q = v[k];
    }
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the stores above have reached the L2 (s_waitcnt vmcnt(0))
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 0) __hip_atomic_store(flag, my_turn + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(flag, my_turn + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (threadIdx.x == 0) cyc[p] = __builtin_amdgcn_s_memtime() - t0;
}

template <int MODE>
static void run(const char* name, int K, int R, int xcd, int stride8) {
  unsigned* flag; double* tile; unsigned* xcc; long long* cyc;
  CHECK(hipMalloc(&flag, 256)); CHECK(hipMalloc(&tile, kTile * sizeof(double))); CHECK(hipMalloc(&xcc, K * sizeof(unsigned))); CHECK(hipMalloc(&cyc, K * sizeof(long long)));
  float best = 1e30f;
  std::vector<double> h(kTile); std::vector<unsigned> hx(K);
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(flag, 0, 256)); CHECK(hipMemset(tile, 0, kTile * sizeof(double)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(ring_kernel<MODE>, dim3(stride8 ? 8 * K : K), dim3(256), 0, 0, K, R, xcd, flag, tile, xcc, cyc, stride8);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  CHECK(hipMemcpy(h.data(), tile, kTile * sizeof(double), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hx.data(), xcc, K * sizeof(unsigned), hipMemcpyDeviceToHost));
  bool ok = true;
  for (double v : h) ok = ok && v == (double)(K * R);
  std::sort(hx.begin(), hx.end());
  const bool one_xcd = hx.front() == hx.back();
  printf("%-64s K %2d  %7.2f us per hand-over  (%d hand-overs, %.1f us)  tile %s  XCC ids %u..%u%s\n", name, K, best * 1e3 / (K * R), K * R, best * 1e3,
         ok ? "correct" : "WRONG", hx.front(), hx.back(), one_xcd ? " (one XCD)" : " (several XCDs)");
  CHECK(hipFree(flag)); CHECK(hipFree(tile)); CHECK(hipFree(xcc)); CHECK(hipFree(cyc));
}

int main() {
  const int R = 200;
  for (int K : {2, 4, 16, 32}) {
    run<0>("same XCD (blockIdx % 8 == 0), relaxed sc1 accesses + waitcnt", K, R, 0, 1);
    run<1>("same XCD (blockIdx % 8 == 0), release / acquire at agent scope", K, R, 0, 1);
    run<1>("consecutive workgroups (all XCDs), release / acquire at agent scope", K, R, 0, 0);
    run<0>("consecutive workgroups (all XCDs), relaxed sc1 accesses (NOT coherent)", K, R, 0, 0);
  }
  run<0>("same XCD (blockIdx % 8 == 5), relaxed sc1 accesses + waitcnt", 8, R, 5, 1);
  return 0;
}
