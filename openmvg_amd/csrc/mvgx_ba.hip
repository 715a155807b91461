// libmvgx_hip.so — bundle adjustment entry points. PLACEHOLDER until the LM kernels land (same round):
// every call fails loudly with MVGX_ERR_UNSUPPORTED; nothing falls back to a CPU path.
#include "mvgx_common.h"

extern "C" {

void mvgx_ba_default_options(mvgx_ba_options* o) {
  if (!o) return;
  o->max_num_iterations = 50;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_radius = 1e4;
  o->max_radius = 1e16;
  o->min_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->verbose = 0;
}

#define MVGX_BA_TODO(name)                                         \
  mvgx::set_error(name ": BA kernels not built into this library"); \
  return MVGX_ERR_UNSUPPORTED

int mvgx_ba_create(int, const mvgx_ba_problem*, mvgx_ba_ctx**) { MVGX_BA_TODO("mvgx_ba_create"); }
int mvgx_ba_destroy(mvgx_ba_ctx*) { return MVGX_OK; }
int mvgx_ba_set_allreduce(mvgx_ba_ctx*, mvgx_allreduce_f64, void*) { MVGX_BA_TODO("mvgx_ba_set_allreduce"); }
int mvgx_ba_solve(mvgx_ba_ctx*, const mvgx_ba_options*, mvgx_ba_summary*) { MVGX_BA_TODO("mvgx_ba_solve"); }
int mvgx_ba_lm_iteration(mvgx_ba_ctx*, const mvgx_ba_options*, mvgx_ba_summary*) { MVGX_BA_TODO("mvgx_ba_lm_iteration"); }
int mvgx_ba_read_params(mvgx_ba_ctx*, double*, double*, double*) { MVGX_BA_TODO("mvgx_ba_read_params"); }
int mvgx_ba_evaluate(mvgx_ba_ctx*, double*, double*) { MVGX_BA_TODO("mvgx_ba_evaluate"); }

}  // extern "C"
