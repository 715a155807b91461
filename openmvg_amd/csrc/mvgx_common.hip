// libmvgx_hip.so — common entry points (error string, device enumeration).
#include "mvgx_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace mvgx {

std::string& last_error_ref() {
  static thread_local std::string err;
  return err;
}

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
}

int select_device(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    set_error("no HIP device visible (libmvgx_hip needs a gfx950 GPU; there is no CPU fallback)");
    return MVGX_ERR_NODEV;
  }
  if (device >= count) {
    set_error("device %d out of range (%d visible)", device, count);
    return MVGX_ERR_ARG;
  }
  if (device >= 0) MVGX_HIP(hipSetDevice(device));
  return MVGX_OK;
}

int devices_from_env(std::vector<int>& out) {
  out.clear();
  const char* env = getenv("MVGX_DEVICES");
  if (!env || !*env) return MVGX_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    set_error("no HIP device visible (libmvgx_hip needs a gfx950 GPU; there is no CPU fallback)");
    return MVGX_ERR_NODEV;
  }
  if (!strcmp(env, "all")) {
    for (int d = 0; d < count; ++d) out.push_back(d);
    return MVGX_OK;
  }
  for (const char* p = env; *p;) {
    char* end = nullptr;
    const long v = strtol(p, &end, 10);
    if (end == p || v < 0 || v >= count) {
      set_error("MVGX_DEVICES='%s': expected 'all' or a comma-separated list of ordinals below %d", env, count);
      return MVGX_ERR_ARG;
    }
    out.push_back((int)v);
    p = end;
    while (*p == ',' || *p == ' ') ++p;
  }
  return MVGX_OK;
}

namespace {
struct StreamCache {
  std::mutex mu;
  std::vector<std::pair<int, hipStream_t>> idle;
};
StreamCache& stream_cache() { static StreamCache* c = new StreamCache(); return *c; }   // never destroyed (see slab_cache)
}  // namespace

int acquire_stream(hipStream_t* out) {
  int device = 0;
  MVGX_HIP(hipGetDevice(&device));
  {
    StreamCache& c = stream_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    for (size_t k = 0; k < c.idle.size(); ++k)
      if (c.idle[k].first == device) {
        *out = c.idle[k].second;
        c.idle.erase(c.idle.begin() + k);
        return MVGX_OK;
      }
  }
  MVGX_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
  return MVGX_OK;
}

void release_stream(int device, hipStream_t s) {
  if (!s) return;
  {
    StreamCache& c = stream_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    size_t n = 0;
    for (const auto& e : c.idle) n += e.first == device;
    if (n < 8) { c.idle.emplace_back(device, s); return; }
  }
  (void)hipStreamDestroy(s);
}

namespace {
struct SlabCache {
  std::mutex mu;
  struct Entry { char* p; size_t size; int device; };
  std::vector<Entry> free_;
  const char* env; size_t default_mb;
  SlabCache(const char* e, size_t mb) : env(e), default_mb(mb) {}
  size_t limit() const {
    const char* e = getenv(env);
    return (e ? (size_t)strtoull(e, nullptr, 10) : default_mb) << 20;
  }
};
// never destroyed: they outlive the HIP runtime's teardown order
SlabCache& slab_cache(Arena::Kind k) {
  static SlabCache* dev = new SlabCache("MVGX_DEVICE_CACHE_MB", 4096);
  static SlabCache* host = new SlabCache("MVGX_HOST_CACHE_MB", 2048);
  return k == Arena::kHost ? *host : *dev;
}
constexpr size_t kSlabAlign = 256, kFirstSlab = 8u << 20, kMaxBumpSlab = 256u << 20, kOwnSlab = 32u << 20;
hipError_t slab_malloc(Arena::Kind k, void** p, size_t bytes) {
  return k == Arena::kHost ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
}
void slab_free(Arena::Kind k, void* p) { (void)(k == Arena::kHost ? hipHostFree(p) : hipFree(p)); }
}  // namespace

int Arena::take_slab(size_t min_bytes, Slab* out) {
  int device = 0;
  MVGX_HIP(hipGetDevice(&device));
  SlabCache& c = slab_cache(kind_);
  const bool host = kind_ == kHost;   // (page-locked host memory is not tied to a device)
  {
    std::lock_guard<std::mutex> lk(c.mu);
    int best = -1;
    for (size_t k = 0; k < c.free_.size(); ++k)   // smallest cached slab that fits and wastes at most half of itself
      if ((host || c.free_[k].device == device) && c.free_[k].size >= min_bytes && c.free_[k].size <= 2 * min_bytes + kFirstSlab &&
          (best < 0 || c.free_[k].size < c.free_[best].size)) best = (int)k;
    if (best >= 0) {
      *out = Slab{c.free_[best].p, c.free_[best].size, device};
      c.free_.erase(c.free_.begin() + best);
      return MVGX_OK;
    }
  }
  void* p = nullptr;
  hipError_t e = slab_malloc(kind_, &p, min_bytes);
  if (e != hipSuccess) {   // out of memory: give the cache back to the driver and retry once
    (void)hipGetLastError();
    std::vector<SlabCache::Entry> drop;
    { std::lock_guard<std::mutex> lk(c.mu); drop.swap(c.free_); }
    for (auto& d : drop) { if (!host) (void)hipSetDevice(d.device); slab_free(kind_, d.p); }
    (void)hipSetDevice(device);
    e = slab_malloc(kind_, &p, min_bytes);
  }
  if (e != hipSuccess) {
    set_error("%s(%zu bytes) -> %s", host ? "hipHostMalloc" : "hipMalloc", min_bytes, hipGetErrorString(e));
    return MVGX_ERR_HIP;
  }
  *out = Slab{static_cast<char*>(p), min_bytes, device};
  return MVGX_OK;
}

int Arena::alloc(void** out, size_t bytes) {
  bytes = (std::max<size_t>(bytes, 1) + kSlabAlign - 1) / kSlabAlign * kSlabAlign;
  if (bytes >= kOwnSlab) {   // large arrays get a slab of their own (reused by the next context of the same shape)
    // (host: sizes rounded up to 16 MB steps, so that a problem that grows from call to call - incremental SfM - still finds them)
    if (kind_ == kHost) bytes = (bytes + (16u << 20) - 1) / (16u << 20) * (16u << 20);
    Slab s;
    const int rc = take_slab(bytes, &s);
    if (rc) return rc;
    slabs_.push_back(s);
    *out = s.p;
    return MVGX_OK;
  }
  if (bump_ < 0 || off_ + bytes > slabs_[bump_].size) {
    next_ = next_ ? std::min(2 * next_, kMaxBumpSlab) : kFirstSlab;
    Slab s;
    const int rc = take_slab(std::max(next_, bytes), &s);
    if (rc) return rc;
    slabs_.push_back(s);
    bump_ = (int)slabs_.size() - 1;
    off_ = 0;
  }
  *out = slabs_[bump_].p + off_;
  off_ += bytes;
  return MVGX_OK;
}

void Arena::release() {
  if (slabs_.empty()) return;
  SlabCache& c = slab_cache(kind_);
  const bool host = kind_ == kHost;
  std::vector<Slab> drop;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    const size_t limit = c.limit();
    for (const Slab& s : slabs_) {   // MVGX_DEVICE_CACHE_MB bounds what the cache keeps PER DEVICE, MVGX_HOST_CACHE_MB in total
      size_t cached = 0;
      for (const auto& e : c.free_) cached += (host || e.device == s.device) ? e.size : 0;
      if (cached + s.size <= limit) c.free_.push_back(SlabCache::Entry{s.p, s.size, s.device});
      else drop.push_back(s);
    }
  }
  if (!drop.empty()) {   // the caller's current device is left as it was
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (const Slab& s : drop) { if (!host) (void)hipSetDevice(s.device); slab_free(kind_, s.p); }
    if (have) (void)hipSetDevice(cur);
  }
  slabs_.clear();
  bump_ = -1; off_ = 0; next_ = 0;
}

// Hands the idle slabs of the current device back to the driver: cached slabs are invisible to every other allocator, so a
// hipMemGetInfo reading counts them as used and a plain hipMalloc next to the cache can fail while gigabytes sit idle. Called
// before such a reading and after a failed hipMalloc (then the allocation is retried once).
void trim_device_cache() {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return;
  SlabCache& c = slab_cache(Arena::kDevice);
  std::vector<SlabCache::Entry> drop;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    for (size_t k = 0; k < c.free_.size();) {
      if (c.free_[k].device == device) { drop.push_back(c.free_[k]); c.free_.erase(c.free_.begin() + k); }
      else ++k;
    }
  }
  for (auto& d : drop) (void)hipFree(d.p);
}

void trim_host_cache() {
  SlabCache& c = slab_cache(Arena::kHost);
  std::vector<SlabCache::Entry> drop;
  { std::lock_guard<std::mutex> lk(c.mu); drop.swap(c.free_); }
  for (auto& d : drop) (void)hipHostFree(d.p);
}

hipError_t device_malloc(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    trim_device_cache();
    e = hipMalloc(p, bytes);
  }
  return e;
}

size_t Arena::bytes_reserved() const {
  size_t n = 0;
  for (const Slab& s : slabs_) n += s.size;
  return n;
}

}  // namespace mvgx

extern "C" {

const char* mvgx_last_error(void) { return mvgx::last_error_ref().c_str(); }

int mvgx_abi_version(void) { return 12; }

int mvgx_device_count(int* count) {
  if (!count) return MVGX_ERR_ARG;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) c = 0;
  *count = c;
  return MVGX_OK;
}

}  // extern "C"
