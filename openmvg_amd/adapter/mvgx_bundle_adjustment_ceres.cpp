// mvgx_bundle_adjustment_ceres.cpp — link-time replacement for openMVG's src/openMVG/sfm/sfm_data_BA_ceres.cpp.
//
// Compile this TU (with mvgx_bundle_adjustment.cpp) INSTEAD of the reference's sfm_data_BA_ceres.cpp: it defines the
// same symbols — Bundle_Adjustment_Ceres::BA_Ceres_options::BA_Ceres_options(bool, bool), the Bundle_Adjustment_Ceres
// constructor, ceres_options() and Adjust() (sfm_data_BA_ceres.hpp:31-69) — so sequential_SfM.cpp:1190-1215 and the
// other call sites that construct Bundle_Adjustment_Ceres by name run the MI355X solver unchanged. Only Ceres'
// enum header is needed (ceres/types.h); no Ceres object code is linked by this file. IntrinsicsToCostFunction
// (sfm_data_BA_ceres.hpp:24-29) builds ceres::CostFunction objects and is therefore NOT provided here: in the reference tree
// it is called only inside sfm_data_BA_ceres.cpp itself (:367, :415), i.e. from the Adjust() this file replaces.
#include <utility>

#include "ceres/types.h"

#include "openMVG/sfm/sfm_data_BA_ceres.hpp"

#include "mvgx_bundle_adjustment.hpp"

namespace openMVG {
namespace sfm {

Bundle_Adjustment_Ceres::BA_Ceres_options::BA_Ceres_options(const bool bVerbose, bool bmultithreaded)
    : bVerbose_(bVerbose),
      nb_threads_(1),  // host threads have no role on the device path; kept for source compatibility
      parameter_tolerance_(1e-8),
      gradient_tolerance_(1e-10),
      bUse_loss_function_(true),
      max_num_iterations_(50),
      max_linear_solver_iterations_(500) {
  (void)bmultithreaded;
  bCeres_summary_ = false;
  // The reduced camera system is always eliminated with the Schur complement and factored densely on the GPU.
  linear_solver_type_ = ceres::DENSE_SCHUR;
  preconditioner_type_ = ceres::JACOBI;
  sparse_linear_algebra_library_type_ = ceres::NO_SPARSE;
}

Bundle_Adjustment_Ceres::Bundle_Adjustment_Ceres(const Bundle_Adjustment_Ceres::BA_Ceres_options& options)
    : ceres_options_(options) {}

Bundle_Adjustment_Ceres::BA_Ceres_options& Bundle_Adjustment_Ceres::ceres_options() { return ceres_options_; }

bool Bundle_Adjustment_Ceres::Adjust(SfM_Data& sfm_data, const Optimize_Options& options) {
  Bundle_Adjustment_HIP::Options o;
  o.bVerbose_ = ceres_options_.bVerbose_;
  o.parameter_tolerance_ = ceres_options_.parameter_tolerance_;
  o.gradient_tolerance_ = ceres_options_.gradient_tolerance_;
  o.bUse_loss_function_ = ceres_options_.bUse_loss_function_;
  o.max_num_iterations_ = ceres_options_.max_num_iterations_;
  Bundle_Adjustment_HIP engine(o);
  return engine.Adjust(sfm_data, options);
}

}  // namespace sfm
}  // namespace openMVG
