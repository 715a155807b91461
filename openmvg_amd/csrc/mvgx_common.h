// Shared host-side helpers for libmvgx_hip.so (error plumbing, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "mvgx.h"

namespace mvgx {

std::string& last_error_ref();
void set_error(const char* fmt, ...);

// Evaluates a HIP runtime call; on failure records file:line + hipGetErrorString and returns MVGX_ERR_HIP
// from the enclosing function.
#define MVGX_HIP(call)                                                                        \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      ::mvgx::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return MVGX_ERR_HIP;                                                                    \
    }                                                                                         \
  } while (0)

#define MVGX_REQUIRE(cond, code, ...)    \
  do {                                   \
    if (!(cond)) {                       \
      ::mvgx::set_error(__VA_ARGS__);    \
      return (code);                     \
    }                                    \
  } while (0)

int select_device(int device);
// frees the idle device-memory slabs the Arena cache holds for the current device (see mvgx_common.hip)
void trim_device_cache();
// hipMalloc that, on failure, trims the Arena cache of the current device and tries once more
hipError_t device_malloc(void** p, size_t bytes);

// A non-blocking stream of the current device from a process-wide cache of idle ones (hipStreamCreate costs ~2 ms here, and an
// SfM engine creates a context per Adjust call); release_stream takes back a stream that has been synchronised.
int acquire_stream(hipStream_t* out);
void release_stream(int device, hipStream_t s);

// MVGX_DEVICES: "all" or a comma-separated list of device ordinals (an ordinal may repeat: several contexts on one device,
// used by the single-GPU tests of the multi-device paths). Unset / empty -> `out` stays empty (the current device).
int devices_from_env(std::vector<int>& out);

// Device memory of one context, sub-allocated from a few large slabs that are handed back to a process-wide cache when the
// context is destroyed: an SfM engine calls Bundle_Adjustment::Adjust hundreds of times (sequential_SfM.cpp:593-596, :1190-1215),
// and a context makes ~80 allocations - through hipMalloc / hipFree that is several milliseconds per call, from the cache a few
// microseconds. MVGX_DEVICE_CACHE_MB bounds what the cache keeps per device (default 4096; 0: no caching); trim_device_cache()
// returns the idle slabs of the current device to the driver.
class Arena {
 public:
  enum Kind { kDevice = 0, kHost = 1 };   // kHost: page-locked host memory (HostArena below)
  explicit Arena(Kind kind = kDevice) : kind_(kind) {}
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;
  ~Arena() { release(); }
  int alloc(void** out, size_t bytes);   // 256-byte aligned, uninitialised; MVGX_ERR_HIP when the memory cannot be had
  void release();                        // every slab back to the cache (or to the driver)
  size_t bytes_reserved() const;
 private:
  struct Slab { char* p; size_t size; int device; };
  std::vector<Slab> slabs_;   // slabs_[bump_] is the one small requests are carved from
  Kind kind_;
  int bump_ = -1;
  size_t off_ = 0, next_ = 0;
  int take_slab(size_t min_bytes, Slab* out);
};

// Host scratch of a create call (structure build, upload sources): page-locked slabs from a process-wide cache, like the device
// slabs. Fresh pageable memory is expensive at this size - measured on the bench host: malloc + first touch of 256 MB 50 ms
// (page faults), free 18 ms (munmap) - and a BA context of n observations builds ~50 n bytes of lists: through std::vector
// that was most of mvgx_ba_create. From the cache the memory is already mapped, and hipMemcpyAsync from it is a true asynchronous
// DMA (the uploads overlap the rest of the host work). MVGX_HOST_CACHE_MB bounds what the cache keeps (default 2048; 0: none).
// The memory is NOT zeroed.
class HostArena {
 public:
  HostArena() : a_(Arena::kHost) {}
  template <class T>
  int array(T** out, size_t n) {
    void* p = nullptr;
    const int rc = a_.alloc(&p, n * sizeof(T));
    *out = static_cast<T*>(p);
    return rc;
  }
  void release() { a_.release(); }
 private:
  Arena a_;
};
// returns the idle page-locked slabs to the driver (tests; a host that wants the memory back)
void trim_host_cache();

}  // namespace mvgx
