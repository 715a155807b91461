// Host build of openmvg_amd/csrc/ba_math.h (the closed-form math the HIP kernels run), so the CPU test-suite can
// compare it with the oracle's autodiff without a GPU. Test infrastructure; g++ -O2 -shared.
#include "ba_math.h"

extern "C" {
void host_eval_observation(int model, const double* intr, const double* pose, const double* X, const double* obs,
                           double* r, double* Ji, double* Jc, double* Jp) {
  mvgx_ba::eval_observation<true>(model, intr, pose, X, obs, r, Ji, Jc, Jp);
}
void host_eval_residual(int model, const double* intr, const double* pose, const double* X, const double* obs, double* r) {
  mvgx_ba::eval_observation<false>(model, intr, pose, X, obs, r, nullptr, nullptr, nullptr);
}
void host_eval_prior(const double* pose, const double* center, const double* weight, double* r, double* Jc) {
  mvgx_ba::eval_pose_center_prior<true>(pose, center, weight, r, Jc);
}
void host_huber(double a, double s, double* rho) { mvgx_ba::huber_rho(a, s, rho); }
int host_invert_spd3(const double* v, double* inv) { return mvgx_ba::invert_spd3(v, inv) ? 1 : 0; }
}
