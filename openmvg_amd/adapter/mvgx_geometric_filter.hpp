// mvgx_geometric_filter.hpp - openMVG-side adapter of the MI355X geometric filter.
//
// ImageCollectionGeometricFilter::Robust_model_estimation is a member TEMPLATE defined in the reference header
// (openMVG/matching_image_collection/GeometricFilter.hpp:66-131); its callers (software/SfM/main_GeometricFilter.cpp:303-309,
// software/SfM/main_ComputeMatches... pipelines) instantiate it with a functor type. There is no translation unit to replace, so the
// drop-in is an explicit specialisation for the fundamental-matrix functor: a caller that includes THIS header (one line after its
// include of GeometricFilter.hpp, or -include on the command line) gets the declaration below, the compiler no longer instantiates
// the primary template for GeometricFilter_FMatrix_AC, and the linker takes the definition of mvgx_geometric_filter.cpp, which
// runs all image pairs of the container through mvgx_geofilter_f_acransac (include/mvgx.h). Same signature, same container
// (_map_GeometricMatches), same acceptance rule; the guided-matching step, if asked for, runs the reference's own
// Geometry_guided_matching with the estimated model. The homography functor (GeometricFilter_HMatrix_AC, H_ACRobust.hpp) has the same
// kind of specialisation over mvgx_geofilter_h_acransac_indexed, and the essential-matrix functor (GeometricFilter_EMatrix_AC, E_ACRobust.hpp:
// main_GeometricFilter -g e) over mvgx_geofilter_e_acransac_indexed - the bearing vectors of the features come from the cameras' own
// operator(), pairs without two pinhole cameras take the reference's functor (which warns and rejects them). The angular essential
// functors (GeometricFilter_ESphericalMatrix_AC_Angular<false | true>, E_ACRobust_Angular.hpp: -g a / -g u) run their a-contrario stage
// through mvgx_geofilter_e_angular_acransac_indexed and their cheirality stage with the reference's own RelativePoseFromEssential.
// The orthographic functor (GeometricFilter_EOMatrix_RA, Eo_Robust.hpp: -g o) goes through mvgx_geofilter_eo_acransac_indexed with the
// hnormalized bearing vectors of pinhole cameras and the functor's camera-plane bound per pair.
#ifndef MVGX_GEOMETRIC_FILTER_HPP
#define MVGX_GEOMETRIC_FILTER_HPP

#include "openMVG/matching_image_collection/E_ACRobust.hpp"
#include "openMVG/matching_image_collection/E_ACRobust_Angular.hpp"
#include "openMVG/matching_image_collection/Eo_Robust.hpp"
#include "openMVG/matching_image_collection/F_ACRobust.hpp"
#include "openMVG/matching_image_collection/GeometricFilter.hpp"
#include "openMVG/matching_image_collection/H_ACRobust.hpp"

namespace openMVG {
namespace matching_image_collection {

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_FMatrix_AC>(
    const GeometricFilter_FMatrix_AC& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* progress_bar);

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_HMatrix_AC>(
    const GeometricFilter_HMatrix_AC& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* progress_bar);

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_EMatrix_AC>(
    const GeometricFilter_EMatrix_AC& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* progress_bar);

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_ESphericalMatrix_AC_Angular<false>>(
    const GeometricFilter_ESphericalMatrix_AC_Angular<false>& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* progress_bar);

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_ESphericalMatrix_AC_Angular<true>>(
    const GeometricFilter_ESphericalMatrix_AC_Angular<true>& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* progress_bar);

template <>
void ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_EOMatrix_RA>(
    const GeometricFilter_EOMatrix_RA& functor, const PairWiseMatches& putative_matches, const bool b_guided_matching,
    const double d_distance_ratio, system::ProgressInterface* progress_bar);

}  // namespace matching_image_collection
}  // namespace openMVG

#endif  // MVGX_GEOMETRIC_FILTER_HPP
