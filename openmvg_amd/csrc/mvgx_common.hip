// libmvgx_hip.so — common entry points (error string, device enumeration).
#include "mvgx_common.h"

#include <cstdlib>
#include <cstring>

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

}  // namespace mvgx

extern "C" {

const char* mvgx_last_error(void) { return mvgx::last_error_ref().c_str(); }

int mvgx_abi_version(void) { return 4; }

int mvgx_device_count(int* count) {
  if (!count) return MVGX_ERR_ARG;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) c = 0;
  *count = c;
  return MVGX_OK;
}

}  // extern "C"
