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

// MVGX_DEVICES: "all" or a comma-separated list of device ordinals (an ordinal may repeat: several contexts on one device,
// used by the single-GPU tests of the multi-device paths). Unset / empty -> `out` stays empty (the current device).
int devices_from_env(std::vector<int>& out);

}  // namespace mvgx
