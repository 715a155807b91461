// mvgx_adapter_policy.hpp - what a replacement TU does when the device path fails (shared by the four adapter TUs).
//
// The reference's convention ("Error conventions"): no exceptions out of Match() / Adjust(); problems are logged with
// OPENMVG_LOG_ERROR; Match returns void and skips what it cannot do (Matcher_Regions.cpp:65-69,74-75,85-90), Adjust returns false
// (sfm_data_BA_ceres.cpp:388-392,503-507). An unchanged main_ComputeMatches / main_GeometricFilter / main_SfM has no handler, so
// a throw would end in std::terminate. The replacement TUs therefore, on a failing mvgx_* call,
//   * log the failure ONCE per process and component, and
//   * (default, MVGX_ON_DEVICE_ERROR unset or "fallback") finish the call with the host application's OWN reference code, which is
//     linked into that application anyway: the RegionMatcherFactory route for Matcher_Regions, CascadeHasher + Match_HashedDescriptions
//     for Cascade_Hashing_Matcher_Regions, the functor's own Robust_estimation per pair for the geometric filter; Adjust() returns false;
//   * or throw std::runtime_error when MVGX_ON_DEVICE_ERROR=throw (what rounds 1 - 3 did unconditionally; the choice of a caller that
//     prefers to stop over running for hours on host cores).
// This is code of the host application, not of oracle/: it is never on a tested parity path or in a timed region - the tests prove
// which route ran through mvgx_adapter_counters (device pairs > 0 and no failure in every parity / timing test; the fallback only
// under an injected failure, tests/test_adapter_emu_cpu.py).
//
// MVGX_ADAPTER_INJECT_FAILURE=<component>:<stage> (test hook), e.g. "match:create", "match:run", "cascade:hash", "geofilter:run":
// the named mvgx_* call of the named component is treated as failed with MVGX_ERR_NODEV without being made.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "mvgx.h"
#include "openMVG/system/logger.hpp"

namespace mvgx_adapter {

struct Counters {
  std::atomic<uint64_t> device_pairs{0};     // image pairs whose result came from the device path
  std::atomic<uint64_t> fallback_pairs{0};   // image pairs finished by the host application's reference code after a device failure
  std::atomic<uint64_t> device_failures{0};  // failing mvgx_* calls seen (incl. injected ones)
  std::atomic<uint64_t> logged{0};           // bit per component: the failure of that component has been logged
  std::atomic<uint64_t> guided_device_pairs{0};   // geometric filter: pairs whose guided matching ran on the device (mvgx_guided_match_u8)
  std::atomic<uint64_t> guided_host_pairs{0};     // ... with the reference's own Geometry_guided_matching (other descriptor types, device failure)
};
inline Counters& counters() {   // one instance per linked image (function-local static of an inline function)
  static Counters c;
  return c;
}

inline bool throw_on_device_error() {
  const char* env = std::getenv("MVGX_ON_DEVICE_ERROR");
  return env && !std::strcmp(env, "throw");
}

// rc of a device call, or MVGX_ERR_NODEV without making it when the test hook names <component>:<stage>
inline bool injected(const char* component, const char* stage) {
  const char* env = std::getenv("MVGX_ADAPTER_INJECT_FAILURE");
  if (!env) return false;
  const std::string want = std::string(component) + ":" + stage;
  return want == env;
}

enum Component { kMatch = 0, kCascade = 1, kGeofilter = 2, kBundle = 3, kFilters = 4 };

// Logs once per process and component; throws when the caller asked for it. Returns normally otherwise: the caller continues on
// the reference route (or returns false).
inline void device_failure(Component comp, const char* component, const char* stage, int rc, bool was_injected) {
  Counters& c = counters();
  c.device_failures.fetch_add(1);
  const std::string msg = std::string("mvgx (MI355X ") + component + "): " + stage + " failed with status " + std::to_string(rc) + ": " +
                          (was_injected ? "injected by MVGX_ADAPTER_INJECT_FAILURE" : mvgx_last_error());
  const uint64_t bit = uint64_t(1) << comp;
  if (!(c.logged.fetch_or(bit) & bit))
    OPENMVG_LOG_ERROR << msg << (throw_on_device_error() ? "" : comp == kBundle ? " - Adjust() returns false" : comp == kFilters ? " - continuing with the reference's own CPU code"
                                                                                 : " - continuing with the reference's own CPU code for the remaining pairs");
  if (throw_on_device_error()) throw std::runtime_error(msg);
}

}  // namespace mvgx_adapter

// diagnostic / test entry of an adapter library: {device pairs, fallback pairs, device failures}; reset = 1 clears them afterwards
extern "C" __attribute__((weak)) void mvgx_adapter_counters(uint64_t out[3], int reset) {
  mvgx_adapter::Counters& c = mvgx_adapter::counters();
  if (out) { out[0] = c.device_pairs.load(); out[1] = c.fallback_pairs.load(); out[2] = c.device_failures.load(); }
  if (reset) { c.device_pairs = 0; c.fallback_pairs = 0; c.device_failures = 0; c.logged = 0; }
}
// ... of the geometric filter's second stage: {pairs guided on the device, pairs guided by the reference's host code}
extern "C" __attribute__((weak)) void mvgx_adapter_guided_counters(uint64_t out[2], int reset) {
  mvgx_adapter::Counters& c = mvgx_adapter::counters();
  if (out) { out[0] = c.guided_device_pairs.load(); out[1] = c.guided_host_pairs.load(); }
  if (reset) { c.guided_device_pairs = 0; c.guided_host_pairs = 0; }
}
