// Hand-written equivalent of the config.h the reference's vendored Ceres 1.13 generates at configure time
// (third_party/ceres-solver/cmake/config.h.in) for openMVG's internal-Ceres build
// (third_party/ceres-solver/CMakeLists.txt:67-82: EIGENSPARSE ON, SUITESPARSE/CXSPARSE/LAPACK OFF, OPENMP ON,
//  SCHUR_SPECIALIZATIONS ON, CUSTOM_BLAS ON, MINIGLOG ON). Used where no configured openMVG build tree exists: the adapter TUs
// (whose openMVG headers include <ceres/...>) compiled against the bare source tree, and the oracle/_ref build of the checker.
// A maintainer building inside openMVG's CMake uses the generated config.h instead.
#ifndef CERES_PUBLIC_INTERNAL_CONFIG_H_
#define CERES_PUBLIC_INTERNAL_CONFIG_H_
#define CERES_USE_EIGEN_SPARSE
#define CERES_NO_LAPACK
#define CERES_NO_SUITESPARSE
#define CERES_NO_CXSPARSE
#define CERES_USE_OPENMP
#define CERES_HAVE_PTHREAD
#define CERES_HAVE_RWLOCK
#define CERES_STD_UNORDERED_MAP
#endif  // CERES_PUBLIC_INTERNAL_CONFIG_H_
