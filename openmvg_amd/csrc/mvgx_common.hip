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
struct SlabCache {
  std::mutex mu;
  struct Entry { char* p; size_t size; int device; };
  std::vector<Entry> free_;
  size_t limit() {
    static const size_t mb = [] { const char* e = getenv("MVGX_DEVICE_CACHE_MB"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)4096; }();
    return mb << 20;
  }
};
SlabCache& slab_cache() { static SlabCache* c = new SlabCache(); return *c; }   // never destroyed: outlives the HIP runtime's teardown order
constexpr size_t kSlabAlign = 256, kFirstSlab = 8u << 20, kMaxBumpSlab = 256u << 20, kOwnSlab = 32u << 20;
}  // namespace

int Arena::take_slab(size_t min_bytes, Slab* out) {
  int device = 0;
  MVGX_HIP(hipGetDevice(&device));
  SlabCache& c = slab_cache();
  {
    std::lock_guard<std::mutex> lk(c.mu);
    int best = -1;
    for (size_t k = 0; k < c.free_.size(); ++k)   // smallest cached slab that fits and wastes at most half of itself
      if (c.free_[k].device == device && c.free_[k].size >= min_bytes && c.free_[k].size <= 2 * min_bytes + kFirstSlab &&
          (best < 0 || c.free_[k].size < c.free_[best].size)) best = (int)k;
    if (best >= 0) {
      *out = Slab{c.free_[best].p, c.free_[best].size, device};
      c.free_.erase(c.free_.begin() + best);
      return MVGX_OK;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, min_bytes);
  if (e != hipSuccess) {   // out of memory: give the cache back to the driver and retry once
    (void)hipGetLastError();
    std::vector<SlabCache::Entry> drop;
    { std::lock_guard<std::mutex> lk(c.mu); drop.swap(c.free_); }
    for (auto& d : drop) { (void)hipSetDevice(d.device); (void)hipFree(d.p); }
    (void)hipSetDevice(device);
    e = hipMalloc(&p, min_bytes);
  }
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) -> %s", min_bytes, hipGetErrorString(e));
    return MVGX_ERR_HIP;
  }
  *out = Slab{static_cast<char*>(p), min_bytes, device};
  return MVGX_OK;
}

int Arena::alloc(void** out, size_t bytes) {
  bytes = (std::max<size_t>(bytes, 1) + kSlabAlign - 1) / kSlabAlign * kSlabAlign;
  if (bytes >= kOwnSlab) {   // large arrays get a slab of their own (reused by the next context of the same shape)
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
  SlabCache& c = slab_cache();
  std::vector<Slab> drop;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    for (const Slab& s : slabs_) {   // MVGX_DEVICE_CACHE_MB bounds what the cache keeps PER DEVICE
      size_t cached = 0;
      for (const auto& e : c.free_) cached += e.device == s.device ? e.size : 0;
      if (cached + s.size <= c.limit()) c.free_.push_back(SlabCache::Entry{s.p, s.size, s.device});
      else drop.push_back(s);
    }
  }
  if (!drop.empty()) {   // the caller's current device is left as it was
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (const Slab& s : drop) { (void)hipSetDevice(s.device); (void)hipFree(s.p); }
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
  SlabCache& c = slab_cache();
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

int mvgx_abi_version(void) { return 5; }

int mvgx_device_count(int* count) {
  if (!count) return MVGX_ERR_ARG;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) c = 0;
  *count = c;
  return MVGX_OK;
}

}  // extern "C"
