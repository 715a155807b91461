// mvgx_bundle_adjustment_ceres.cpp — link-time replacement for openMVG's src/openMVG/sfm/sfm_data_BA_ceres.cpp.
//
// Compile this TU (with mvgx_bundle_adjustment.cpp) INSTEAD of the reference's sfm_data_BA_ceres.cpp: it defines the
// same symbols — Bundle_Adjustment_Ceres::BA_Ceres_options::BA_Ceres_options(bool, bool), the Bundle_Adjustment_Ceres
// constructor, ceres_options() and Adjust() (sfm_data_BA_ceres.hpp:31-69) — so sequential_SfM.cpp:1190-1215 and the
// other call sites that construct Bundle_Adjustment_Ceres by name run the MI355X solver unchanged. Only Ceres'
// enum header is needed (ceres/types.h); no Ceres object code is linked by this file. IntrinsicsToCostFunction
// (sfm_data_BA_ceres.hpp:24-29) builds ceres::CostFunction objects and is therefore NOT provided here: in the reference tree
// it is called only inside sfm_data_BA_ceres.cpp itself (:367, :415), i.e. from the Adjust() this file replaces.
#include <atomic>
#include <utility>

#include "ceres/types.h"

#include "openMVG/sfm/sfm_data_BA_ceres.hpp"
#include "openMVG/system/logger.hpp"

#include "mvgx.h"

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
  // The reference's default is SPARSE_SCHUR whenever ceres has a sparse library (sfm_data_BA_ceres.cpp:132-146); the device library
  // always has its block-sparse factorisation and uses it where its plan pays (Adjust below).
  linear_solver_type_ = ceres::SPARSE_SCHUR;
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
  // linear_solver_type_ (:483): the callers choose between the two direct Schur solvers by the pose count (sequential_SfM.cpp:1193-1205,
  // sequential_SfM2.cpp:517-521, sfm_stellar_engine.cpp:742-756) - DENSE_SCHUR -> the dense blocked Cholesky of the reduced camera
  // system, SPARSE_SCHUR -> sparsity is exploited: the block-sparse tile Cholesky wherever its plan needs fewer dependent launches and
  // no more tiles than the dense sweep (the library's rule, mvgx_ba_get_solver_info), the dense one otherwise. Every other ceres type
  // (DENSE_QR, DENSE_NORMAL_CHOLESKY, SPARSE_NORMAL_CHOLESKY, CGNR, ITERATIVE_SCHUR) solves the same normal equations; the device
  // library has no un-eliminated or iterative solver, so those run with its own choice (one notice). MVGX_BA_SOLVER outranks all.
  switch (ceres_options_.linear_solver_type_) {
    case ceres::DENSE_SCHUR: o.linear_solver_ = MVGX_BA_LINEAR_SOLVER_DENSE; break;
    case ceres::SPARSE_SCHUR: o.linear_solver_ = MVGX_BA_LINEAR_SOLVER_AUTO; break;
    default: {
      static std::atomic<bool> said{false};
      if (!said.exchange(true))
        OPENMVG_LOG_INFO << "mvgx bundle adjustment: linear_solver_type_ " << ceres_options_.linear_solver_type_
                         << " has no device counterpart - the Schur-complement solver of the library's own choice is used";
      o.linear_solver_ = MVGX_BA_LINEAR_SOLVER_AUTO;
    }
  }
  Bundle_Adjustment_HIP engine(o);
  return engine.Adjust(sfm_data, options);
}

}  // namespace sfm
}  // namespace openMVG
