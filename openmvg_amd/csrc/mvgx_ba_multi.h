// Bundle adjustment on several devices of one process (see mvgx_ba_multi.hip).
#pragma once
#include "mvgx.h"

namespace mvgx {
struct BaMulti;
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
}  // namespace mvgx
