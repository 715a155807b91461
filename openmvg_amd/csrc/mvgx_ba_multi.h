// Bundle adjustment on several devices of one process (see mvgx_ba_multi.hip).
#pragma once
#include "mvgx.h"

#include <cstdint>

namespace mvgx {
struct BaMulti;
// identity of a problem's structure (mvgx_ba.hip::ba_fingerprint): what a context can be re-bound to by mvgx_ba_update
struct BaFingerprint {
  uint64_t h[2] = {0, 0};
  bool operator==(const BaFingerprint& o) const { return h[0] == o.h[0] && h[1] == o.h[1]; }
};
BaFingerprint ba_fingerprint(const mvgx_ba_problem* p);
int ba_multi_create(const int* devices, int n_devices, const mvgx_ba_problem* p, BaMulti** out);
void ba_multi_destroy(BaMulti* m);
int ba_multi_n_shards(const BaMulti* m);
int ba_multi_transport_is_rccl(const BaMulti* m);
int ba_multi_solve(BaMulti* m, const mvgx_ba_options* opt, mvgx_ba_summary* summary, bool one_iteration);
int ba_multi_evaluate(BaMulti* m, double* cost, double* rmse);
int ba_multi_read_params(BaMulti* m, double* poses, double* intrinsics, double* points);
int ba_multi_residuals(BaMulti* m, double* residual_norm);
int ba_multi_track_angles(BaMulti* m, double* max_angle_deg);
int ba_multi_solver_info(BaMulti* m, mvgx_ba_solver_info* out);
int ba_multi_set_linear_solver(BaMulti* m, int kind);
int ba_multi_update(BaMulti* m, const mvgx_ba_problem* p);   // MVGX_ERR_STRUCTURE: not the structure the shards were cut from
// internal (mvgx_ba.hip): fails the RCCL collectives a context has in flight (another shard of the same process failed);
// argument checks shared by mvgx_ba_create and mvgx_ba_create_multi
void ba_ctx_comm_abort(mvgx_ba_ctx* c);
// while set on the calling thread, mvgx_ba_create leaves the symbolic phase of the reduced solve to the first iteration (the shards
// of a multi-device context are bound to a transport right after, and plan on the union of their blocks)
void ba_create_defer_plan(bool on);
int ba_validate_problem(const mvgx_ba_problem* p);
}  // namespace mvgx
