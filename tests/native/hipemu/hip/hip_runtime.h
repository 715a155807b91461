// hipemu — TEST INFRASTRUCTURE ONLY. A minimal single-threaded emulation of the HIP execution model (workgroups of
// fibers, 64-lane waves, LDS, barriers, shuffles, ballots, DPP row permutations, the f64 and i8 MFMAs) so that the
// device code of the product (openmvg_amd/csrc/mvgx_ba.hip, mvgx_bruteforce.hip unchanged; mvgx_match.hip with its
// asynchronous staging helpers substituted, see tests/_emu.py) can be compiled for the host and exercised by the CPU
// test-suite, where no GPU exists. It is never part of libmvgx_hip.so and nothing in openmvg_amd/ loads it: the product has no CPU path.
//
// Semantics: the workgroups of a launch run one after the other; the threads of a workgroup are ucontext fibers on one
// OS thread, switched only at __syncthreads() and at wave-collective operations (__shfl*, MFMA), which wait for all
// live lanes of the wave. "Device memory" is host memory.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <functional>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
struct double2 { double x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

namespace hipemu {
extern thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void block_barrier();
void site(int n);
double shfl_f64(double v, int src_lane_of(int lane, int arg), int arg);
typedef double d4 __attribute__((ext_vector_type(4)));
d4 mfma_f64_16x16x4(double a, double b, d4 c, int, int, int);
double mfma_f64_4x4x4(double a, double b, double c, int, int, int);
int readlane_i32(int v, int src_lane);
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
v16i_t mfma_i32_32x32x32_i8(v4i_t a, v4i_t b, v16i_t c, int, int, int);
v4i_t mfma_i32_16x16x64_i8(v4i_t a, v4i_t b, v4i_t c, int, int, int);
unsigned long long ballot(bool pred);
int update_dpp(int old, int src, int dpp_ctrl, int row_mask, int bank_mask, bool bound_ctrl);
int lane_xor(int lane, int m);
int lane_down(int lane, int d);
int lane_abs(int lane, int s);
}  // namespace hipemu

#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ thread_local   // one workgroup at a time on one OS thread: a per-thread static is the workgroup's LDS

#define __syncthreads() hipemu::block_barrier()
#define MVGX_EMU_SITE(n) hipemu::site(n)   // source marker for the divergence check of the scheduler
static inline double __shfl_xor(double v, int m) { return hipemu::shfl_f64(v, hipemu::lane_xor, m); }
static inline double __shfl_down(double v, int d) { return hipemu::shfl_f64(v, hipemu::lane_down, d); }
static inline double __shfl(double v, int s) { return hipemu::shfl_f64(v, hipemu::lane_abs, s); }
static inline int __shfl_xor(int v, int m) { return hipemu::readlane_i32(v, (int)(hipemu::g_threadIdx.x & 63) ^ m); }
static inline int __shfl(int v, int s) { return hipemu::readlane_i32(v, s); }
static inline int __shfl_down(int v, int d) {
  const int lane = (int)(hipemu::g_threadIdx.x & 63);
  const int r = hipemu::readlane_i32(v, lane + d < 64 ? lane + d : lane);   // every lane takes part in the exchange
  return lane + d < 64 ? r : v;
}
static inline unsigned __shfl_down(unsigned v, int d) { return (unsigned)__shfl_down((int)v, d); }
#define __ballot(p) hipemu::ballot((p))
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline float __int2float_rn(int v) { return (float)v; }
#define __builtin_amdgcn_readfirstlane(v) hipemu::readlane_i32((int)(v), 0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_sched_barrier(a) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_update_dpp hipemu::update_dpp
#define __builtin_amdgcn_mfma_i32_32x32x32_i8 hipemu::mfma_i32_32x32x32_i8
#define __builtin_amdgcn_mfma_i32_16x16x64_i8 hipemu::mfma_i32_16x16x64_i8
static inline int __builtin_amdgcn_sdot4(int a, int b, int c, bool) {   // v_dot4_i32_i8: signed bytes
  for (int k = 0; k < 4; ++k) c += (int)(signed char)(a >> (8 * k)) * (int)(signed char)(b >> (8 * k));
  return c;
}
static inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool) {   // v_dot4_u32_u8
  for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xFFu) * ((b >> (8 * k)) & 0xFFu);
  return c;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64 hipemu::mfma_f64_16x16x4
#define __builtin_amdgcn_mfma_f64_4x4x4f64 hipemu::mfma_f64_4x4x4
#define __builtin_amdgcn_readlane(v, l) hipemu::readlane_i32((v), (l))
static inline int __double2loint(double d) { long long b; memcpy(&b, &d, 8); return (int)(b & 0xffffffffll); }
static inline int __double2hiint(double d) { long long b; memcpy(&b, &d, 8); return (int)(b >> 32); }
static inline double __hiloint2double(int hi, int lo) { long long b = (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo); double d; memcpy(&d, &b, 8); return d; }
#define __builtin_amdgcn_rcp(x) (1.0 / (x))   /* v_rcp_f64: the device refines it with Newton steps */
#define __builtin_amdgcn_rsq(x) (1.0 / sqrt((double)(x)))   /* v_rsq_f64: refined on the device likewise */

static inline int atomicExch(int* p, int v) { const int o = *p; *p = v; return o; }
static inline void __threadfence_system() {}
static inline void __threadfence() {}   // (one host thread runs the lanes: program order is memory order)
static inline double atomicAdd(double* p, double v) { const double o = *p; *p += v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p += v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p += v; return o; }
static inline long long __double_as_longlong(double v) { long long o; __builtin_memcpy(&o, &v, 8); return o; }
static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; if (v < o) *p = v; return o; }
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; if (v > o) *p = v; return o; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double __dadd_rn(double a, double b) { volatile double r = a + b; return r; }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
using std::max;
using std::min;

// ---- runtime API subset used by the BA solver ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorPeerAccessAlreadyEnabled = 704, hipErrorUnknown = 999 };
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline const char* hipGetErrorString(hipError_t) { return "hipemu error"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
#define HIP_SYMBOL(x) (x)
template <class T> static inline hipError_t hipMemcpyFromSymbol(void* dst, const T& sym, size_t n) { memcpy(dst, &sym, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyToSymbol(T& sym, const void* src, size_t n) { memcpy(&sym, src, n); return hipSuccess; }
static inline long long __builtin_amdgcn_s_memtime() { return 0; }
// a wave's lanes are separate fibers here: the compiler-only wave barrier of the hardware must really wait for all lanes
#define __builtin_amdgcn_wave_barrier() ((void)hipemu::readlane_i32(0, 0))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
// failure injection for the error paths: HIPEMU_FAIL_MALLOC_AFTER=n makes the (n+1)-th and later device allocations fail
static inline bool hipemu_malloc_should_fail() {
  static long seen = 0;
  const char* env = getenv("HIPEMU_FAIL_MALLOC_AFTER");
  if (!env) { seen = 0; return false; }
  return seen++ >= atol(env);
}
static inline hipError_t hipMalloc(void** p, size_t n) { if (hipemu_malloc_should_fail()) { *p = nullptr; return hipErrorUnknown; } *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = *total_b = (size_t)64 << 30; return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(uintptr_t(1)); return hipSuccess; }   // (a non-null dummy: callers test the handle)
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

// ---- stream capture / graphs: a captured launch is recorded (arguments by value) and replayed by hipGraphLaunch in
// issue order, which is a topological order of the fork / join structure the capture describes ----
struct hipemuGraph;
typedef hipemuGraph* hipGraph_t;
typedef hipemuGraph* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
enum { hipEventDisableTiming = 2 };
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode);
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*);
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t);
hipError_t hipGraphDestroy(hipGraph_t);
hipError_t hipGraphExecDestroy(hipGraphExec_t);
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(uintptr_t(1)); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
