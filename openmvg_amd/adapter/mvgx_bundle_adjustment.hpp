// mvgx_bundle_adjustment.hpp — openMVG::sfm::Bundle_Adjustment_HIP: the MI355X bundle adjustment behind openMVG's
// abstract sfm::Bundle_Adjustment interface (sfm/sfm_data_BA.hpp:92-105), for callers that can pick an implementation.
// Callers that name Bundle_Adjustment_Ceres (all 12 call sites of the reference) get the same
// engine through the link-time replacement TU mvgx_bundle_adjustment_ceres.cpp.
#ifndef MVGX_BUNDLE_ADJUSTMENT_HPP_
#define MVGX_BUNDLE_ADJUSTMENT_HPP_

#include "openMVG/sfm/sfm_data_BA.hpp"

namespace openMVG {
namespace sfm {

struct SfM_Data;

class Bundle_Adjustment_HIP : public Bundle_Adjustment {
 public:
  // field names and defaults of Bundle_Adjustment_Ceres::BA_Ceres_options that affect the solve
  // (sfm_data_BA_ceres.cpp:110-149); thread / linear-solver choices have no device meaning.
  struct Options {
    bool bVerbose_;
    double parameter_tolerance_;
    double gradient_tolerance_;
    bool bUse_loss_function_;
    int max_num_iterations_;
    int device_;  // HIP device ordinal, -1 = current
    // reduced camera system: MVGX_BA_LINEAR_SOLVER_AUTO (the library's rule), _DENSE, _SPARSE, _SPARSE_PREFERRED (mvgx.h). The
    // Bundle_Adjustment_Ceres replacement maps BA_Ceres_options::linear_solver_type_ onto it (mvgx_bundle_adjustment_ceres.cpp).
    int linear_solver_;
    Options()
        : bVerbose_(true), parameter_tolerance_(1e-8), gradient_tolerance_(1e-10), bUse_loss_function_(true),
          max_num_iterations_(50), device_(-1), linear_solver_(0) {}
  };

  Bundle_Adjustment_HIP() {}
  explicit Bundle_Adjustment_HIP(const Options& options) : options_(options) {}
  Options& options() { return options_; }

  bool Adjust(SfM_Data& sfm_data, const Optimize_Options& options) override;

 private:
  Options options_;
};

}  // namespace sfm
}  // namespace openMVG
#endif
