/*
 * mvgx.h — C ABI of libmvgx_hip.so: MI355X (gfx950) accelerators behind two openMVG interfaces.
 *
 *   matching : exhaustive brute-force L2 2-NN + Lowe ratio on 128-D uint8 descriptors
 *              (drop-in for openMVG::matching_image_collection::Matcher_Regions(ratio, BRUTE_FORCE_L2))
 *   ba       : Levenberg-Marquardt bundle adjustment (Jacobians -> Schur -> reduced solve -> back-substitution)
 *              (drop-in for openMVG::sfm::Bundle_Adjustment_Ceres::Adjust)
 *
 * Plain C: pointers + sizes only, no C++/torch types. Every entry point returns an int status
 * (MVGX_OK == 0) and never throws. `mvgx_last_error()` returns a thread-local message.
 *
 * Reference interfaces each entry point replaces are cited as  <path under /root/reference/src>:<line>.
 */
#ifndef MVGX_H_
#define MVGX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVGX_OK 0
#define MVGX_ERR_ARG 1      /* bad argument (null pointer, dim != 128, index out of range ...)   */
#define MVGX_ERR_HIP 2      /* a HIP runtime call failed; see mvgx_last_error()                  */
#define MVGX_ERR_NODEV 3    /* no gfx950 device visible                                          */
#define MVGX_ERR_STATE 4    /* call order violated (e.g. run before set_regions)                 */
#define MVGX_ERR_UNSUPPORTED 5 /* semantics the device path does not reproduce (ratio > 1, camera model) */
#define MVGX_ERR_NUMERIC 6  /* BA: linear solve failed / non-finite cost                         */
#define MVGX_ERR_STRUCTURE 7 /* mvgx_ba_update: the problem's structure is not the context's (the context is untouched) */

const char* mvgx_last_error(void);
int mvgx_device_count(int* count);
/* abi version, bumped on any signature or struct-layout change (2: mvgx_ba_problem control points / priors;
 * 3: mvgx_ba_get_solver_info; 4: multi-device contexts, mvgx_match_run_stream; 5: geometric filter; 6: indexed filter entry,
 * cascade hashing on the device; 7: homography model of the geometric filter; 8: iteration / clock counters in
 * mvgx_geofilter_stats, essential-matrix model of the geometric filter; 9: mvgx_host_parallel_for; 10: guided matching;
 * 11: mvgx_cascade_hash_regions_typed; 12: mvgx_guided_match for float and binary regions) */
int mvgx_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * MATCHING
 * replaces: matching_image_collection/Matcher_Regions.cpp:32-107 (Matcher_Regions::Match),
 *           matching/regions_matcher.hpp:162-207 (RegionsMatcherT::MatchDistanceRatio),
 *           matching/matcher_brute_force.hpp:100-200 (ArrayMatcherBruteForce::SearchNeighbours),
 *           matching/metric.hpp:55-93 (L2<uint8_t>), matching/matching_filters.hpp:39-60 (NNdistanceRatio)
 * ---------------------------------------------------------------------------------------------- */

typedef struct mvgx_match_ctx mvgx_match_ctx;

/* Called serially on the calling thread, once per image pair that has >= 1 putative match
 * (Matcher_Regions.cpp:95-103 inserts only non-empty vectors). `ij` holds n (i_in_I, j_in_J) uint32
 * pairs in ascending j, exactly the IndMatch(i_, j_) order of regions_matcher.hpp:198-204. */
typedef void (*mvgx_match_sink)(void* user, uint32_t I, uint32_t J, const uint32_t* ij, uint32_t n);

typedef struct mvgx_match_stats {
  uint64_t n_pairs;            /* image pairs processed on the device                              */
  uint64_t n_desc_pairs;       /* sum over processed pairs of nI * nJ distance evaluations          */
  uint64_t n_matches;          /* putative matches emitted                                          */
  uint64_t n_kernel_launches;  /* launches of the dominant kernel (l2_top2_ratio)                   */
  double   kernel_ms;          /* HIP-event time summed over those launches (profile mode only)     */
  double   total_ms;           /* HIP-event time of the whole device pass (match + compaction)      */
  uint32_t kernel_vgprs;       /* informational                                                     */
  uint32_t variant;            /* kernel variant actually used                                      */
} mvgx_match_stats;

/* device >= 0: that device. device == -1 ("no preference", what the openMVG adapter passes): the devices named by the
 * environment variable MVGX_DEVICES - "all" or a comma-separated list of ordinals; two or more make a multi-device
 * context (below) - or, when it is unset, the current HIP device. device <= -2: the current HIP device, environment ignored. */
int mvgx_match_create(int device, mvgx_match_ctx** out);
/* One context over several devices of this process (Matcher_Regions.cpp:49-54 fans the pairs of an image I out over host
 * threads; here the pair list is cut into batches that the devices take in turn, one host thread per device, no
 * collective): descriptors are replicated by set_regions, every run call shares its batches out dynamically, results are
 * identical to a single-device run. An ordinal may repeat (several contexts on one device). */
int mvgx_match_create_multi(const int* devices, int n_devices, mvgx_match_ctx** out);
int mvgx_match_destroy(mvgx_match_ctx* ctx);

/* knobs: "variant" (kernel variant id), "profile" (1: HIP events around every match-kernel launch; 2: also a statistics
 * pass that counts the candidates the filter hands to the verify stage, reported in kernel_vgprs),
 * "batch_pairs" (pairs per device batch), "keep_host_results" (0: skip D2H of the match lists),
 * "overlap" (default 1: two batch slots, batch b filters while batch b-1 is verified/compacted/copied; 0: one at a time),
 * "double_buffer_results" (default 0, see mvgx_match_run),
 * "pinned_results" (default 1: the host match lists live in pinned memory - fastest when a context is run many times;
 * 0: plain memory, for one-shot use where pinning a gigabyte costs more than the staged copies; set before the first run),
 * "stream_hold" (mvgx_match_run_stream, default 0; 1: the buffers handed to the sink stay valid until TWO further sink
 * calls have returned or the run has returned - a second set of host buffers per batch slot - so that a caller can
 * convert batch k on its own threads while batches k + 1 and k + 2 arrive),
 * "pinned_stream" (default 1: the batch buffers of mvgx_match_run_stream are pinned; 0: plain memory, for one-shot use).
 * "stream_reserve" (uint32 words per buffer): page-locks the stream's host buffers ahead of the run - callable from another thread
 *   while mvgx_match_set_regions uploads (pinning ~100 MB takes tens of milliseconds the first batches would otherwise wait for).
 * "batch_pairs" (default 32 768): image pairs per device batch. */
int mvgx_match_set_option(mvgx_match_ctx* ctx, const char* key, int64_t value);

/* Load the descriptor arrays of n_images images into HBM (replaces Regions_Provider::get +
 * RegionsMatcherT ctor/Build, sfm_regions_provider.hpp:76-85, regions_matcher.hpp:119-132).
 * desc_rows[k] -> n_desc[k] x dim row-major uint8 (Scalar_Regions::DescriptorRawData, scalar_regions.hpp:93);
 * may be NULL when n_desc[k] == 0. dim must be 128. The arrays are copied; the caller keeps ownership. */
int mvgx_match_set_regions(mvgx_match_ctx* ctx, const uint8_t* const* desc_rows, const uint32_t* n_desc,
                           uint32_t n_images, uint32_t dim);

/* Same, descriptors already resident on the device: one concatenated (sum n_desc) x 128 row-major uint8
 * buffer (device pointer), image k starting at row sum_{m<k} n_desc[m]. */
int mvgx_match_set_regions_device(mvgx_match_ctx* ctx, const void* d_desc_concat, const uint32_t* n_desc,
                                  uint32_t n_images, uint32_t dim);

/* Match `n_pairs` image pairs. pairs_IJ = 2*n_pairs uint32 (I = database image, J = query image, as
 * Matcher_Regions.cpp:57-93). ratio_sq = Square(dist_ratio) computed in float by the caller
 * (regions_matcher.hpp:196, numeric.h:56). ratio_sq > 1 -> MVGX_ERR_UNSUPPORTED (tie order would be
 * libstdc++-specific, stl/indexed_sort.hpp:48-63). Pairs whose I has < 2 descriptors or whose J is
 * empty produce no matches (matcher_brute_force.hpp:108-113, Matcher_Regions.cpp:65-69,85-90).
 * Results stay valid until the next run/destroy; with the option "double_buffer_results" = 1 until the run AFTER the next
 * one: the context then alternates between two result buffers, so a caller may consume run k on another thread while
 * run k + 1 executes (the adapter fills the match container that way). */
int mvgx_match_run(mvgx_match_ctx* ctx, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq,
                   mvgx_match_stats* stats /* may be NULL */);

/* Streaming form of mvgx_match_run for runs whose match lists should not be held in host memory at once (10k images:
 * 5e7 pairs, ~1e10 matches): the lists are handed over batch by batch ("batch_pairs" image pairs each) and host memory
 * stays O(batch) - two pinned buffers per device. `sink` is entered on the CALLING thread only, one batch at a time:
 * pairs [first_pair, first_pair + n_pairs) of the input list, offsets[n_pairs + 1] relative to the batch (in matches),
 * ij = 2 uint32 per match, ascending j within a pair; the pointers are valid during the call only. Batches arrive in
 * ascending order on a single device and in any order on a multi-device context. A non-zero return stops the run (the
 * sink is not entered again; the call still returns MVGX_OK). mvgx_match_results is not affected by this call. */
typedef int (*mvgx_match_batch_sink)(void* user, uint64_t first_pair, uint32_t n_pairs, const uint32_t* offsets,
                                     const uint32_t* ij);
int mvgx_match_run_stream(mvgx_match_ctx* ctx, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq,
                          mvgx_match_batch_sink sink, void* user, mvgx_match_stats* stats /* may be NULL */);

/* Host view of the last run: offsets[n_pairs+1] into ij (in units of matches); ij = 2 uint32 per match. */
int mvgx_match_results(mvgx_match_ctx* ctx, const uint64_t** offsets, const uint32_t** ij);

/* One-shot convenience with the exact shape of Matcher_Regions::Match: upload, run, and feed `sink`
 * pair by pair in ascending (I, J) order of the input list. */
int mvgx_match_pairs_u8_l2(const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                           uint32_t dim, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq,
                           int device, mvgx_match_sink sink, void* user);

/* ------------------------------------------------------------------------------------------------
 * MATCHING OF BINARY DESCRIPTORS (BRUTE_FORCE_HAMMING)
 * replaces, behind the same factory as the L2 path (matching/regions_matcher.cpp:184-191):
 *   RegionsMatcherT<ArrayMatcherBruteForce<unsigned char, Hamming<unsigned char>>>(regions, false) on binary regions
 *   (features::Binary_Regions<SIOPointFeature, 64> = AKAZE_Binary_Regions, features/regions_factory.hpp:26):
 *   matching/metric_hamming.hpp:36-107 (popcount of the XOR, unsigned int), matching/matcher_brute_force.hpp:95-200,
 *   MatchDistanceRatio with the ratio as given (regions_matcher.hpp:162-207), inside the I/J loops of
 *   Matcher_Regions::Match (Matcher_Regions.cpp:57-105). Same call shapes as mvgx_match_*; the ratio is dist_ratio itself
 *   (the metric is not squared), accepted range 0 <= dist_ratio <= 1. desc_bytes: 1..64 (AKAZE MLDB: 64).
 * ---------------------------------------------------------------------------------------------- */
typedef struct mvgx_hamming_ctx mvgx_hamming_ctx;
int mvgx_hamming_create(int device, mvgx_hamming_ctx** out);
int mvgx_hamming_destroy(mvgx_hamming_ctx* ctx);
int mvgx_hamming_set_option(mvgx_hamming_ctx* ctx, const char* key /* "batch_pairs" */, int64_t value);
/* desc_rows[k] -> n_desc[k] x desc_bytes row-major bytes (Binary_Regions::DescriptorRawData, binary_regions.hpp) */
int mvgx_hamming_set_regions(mvgx_hamming_ctx* ctx, const uint8_t* const* desc_rows, const uint32_t* n_desc,
                             uint32_t n_images, uint32_t desc_bytes);
int mvgx_hamming_run(mvgx_hamming_ctx* ctx, const uint32_t* pairs_IJ, uint64_t n_pairs, float dist_ratio,
                     mvgx_match_stats* stats /* may be NULL */);
int mvgx_hamming_results(mvgx_hamming_ctx* ctx, const uint64_t** offsets, const uint32_t** ij);

/* ------------------------------------------------------------------------------------------------
 * MATCHING OF FLOAT DESCRIPTORS (BRUTE_FORCE_L2 on Scalar_Regions<SIOPointFeature, float, 64> = AKAZE_Float_Regions)
 * replaces matching/regions_matcher.cpp:119-124: RegionsMatcherT<ArrayMatcherBruteForce<float, L2<float>>>(regions, true);
 * L2<float> (matching/metric.hpp:98-135) is evaluated with the reference's own operation order in IEEE binary32 without
 * contraction, so distances - and therefore the match lists - are bit-identical, not merely close.
 * Same call shapes as mvgx_match_*; ratio_sq = Square(dist_ratio) in float, accepted range 0 <= ratio_sq <= 1; dim must be 64.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mvgx_l2f_ctx mvgx_l2f_ctx;
int mvgx_l2f_create(int device, mvgx_l2f_ctx** out);
int mvgx_l2f_destroy(mvgx_l2f_ctx* ctx);
int mvgx_l2f_set_option(mvgx_l2f_ctx* ctx, const char* key /* "batch_pairs" */, int64_t value);
/* desc_rows[k] -> n_desc[k] x dim row-major floats (Scalar_Regions::DescriptorRawData, scalar_regions.hpp:93) */
int mvgx_l2f_set_regions(mvgx_l2f_ctx* ctx, const float* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                         uint32_t dim);
int mvgx_l2f_run(mvgx_l2f_ctx* ctx, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq,
                 mvgx_match_stats* stats /* may be NULL */);
int mvgx_l2f_results(mvgx_l2f_ctx* ctx, const uint64_t** offsets, const uint32_t** ij);

/* ------------------------------------------------------------------------------------------------
 * MATCHING OF uint8 DESCRIPTORS OF OTHER LENGTHS (BRUTE_FORCE_L2 on Scalar_Regions<SIOPointFeature, unsigned char, N>,
 * N = 144: AKAZE_Liop_Regions, features/regions_factory.hpp:24)
 * replaces matching/regions_matcher.cpp:75-81 for those region types; L2<uint8_t> (matching/metric.hpp:55-93) in exact integer
 * arithmetic (v_dot4_u32_u8), so the lists are bit-identical. Same call shapes as mvgx_match_* (ratio_sq = Square(dist_ratio));
 * dim must be 64, 128 or 144 - SIFT's 128 is better served by mvgx_match_* (the MFMA path) and accepted here as a cross-check.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mvgx_l2u8_ctx mvgx_l2u8_ctx;
int mvgx_l2u8_create(int device, mvgx_l2u8_ctx** out);
int mvgx_l2u8_destroy(mvgx_l2u8_ctx* ctx);
int mvgx_l2u8_set_option(mvgx_l2u8_ctx* ctx, const char* key /* "batch_pairs" */, int64_t value);
int mvgx_l2u8_set_regions(mvgx_l2u8_ctx* ctx, const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                          uint32_t dim);
int mvgx_l2u8_run(mvgx_l2u8_ctx* ctx, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq,
                  mvgx_match_stats* stats /* may be NULL */);
int mvgx_l2u8_results(mvgx_l2u8_ctx* ctx, const uint64_t** offsets, const uint32_t** ij);

/* ------------------------------------------------------------------------------------------------
 * CASCADE HASHING (CASCADE_HASHING_L2, the default nearest-neighbour method of main_ComputeMatches for scalar regions)
 * replaces the MATCHING stage of matching_image_collection/Cascade_Hashing_Matcher_Regions.cpp:134-215:
 *   matching/cascade_hasher.hpp:241-367 CascadeHasher::Match_HashedDescriptions (candidates from the query's bucket in every
 *   bucket group, de-duplicated in order of appearance, the ten nearest in Hamming distance of the hash codes, exact
 *   L2<uint8> on those, the two smallest (distance, id) pairs) and the NNdistanceRatio test with Square(dist_ratio).
 * The HASHING stage (cascade_hasher.hpp:154-239: zero-mean descriptor, primary / secondary random projections - single-
 * precision Eigen products) stays host code of the caller; its per-descriptor outputs are inputs here, so the device part
 * is integer work and the lists are bit-identical for any ratio. The openMVG adapter runs the reference's own
 * CascadeHasher for the hashing stage and applies the reference's two de-duplication steps (:218-226) to the lists.
 *   desc_rows[k]  n_desc[k] x 128 uint8      hash_codes[k]  n_desc[k] x 16 bytes (stl::dynamic_bitset blocks)
 *   bucket_ids[k] n_desc[k] x n_groups uint16 (HashedDescription::bucket_ids), values < 2^bits_per_bucket
 * Results: per pair the list BEFORE de-duplication, (descriptor of I, descriptor of J) in ascending J.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mvgx_cascade_ctx mvgx_cascade_ctx;
int mvgx_cascade_create(int device, mvgx_cascade_ctx** out);
int mvgx_cascade_destroy(mvgx_cascade_ctx* ctx);
int mvgx_cascade_set_option(mvgx_cascade_ctx* ctx, const char* key /* "batch_pairs" */, int64_t value);
int mvgx_cascade_set_regions(mvgx_cascade_ctx* ctx, const uint8_t* const* desc_rows, const uint8_t* const* hash_codes,
                             const uint16_t* const* bucket_ids, const uint32_t* n_desc, uint32_t n_images, uint32_t dim,
                             uint32_t hash_bytes, uint32_t n_groups, uint32_t bits_per_bucket);
/* ... for the other scalar region types Cascade_Hashing_Matcher_Regions.cpp:233-262 accepts (ABI 9): scalar_type 0 = uint8 rows of
 * 128 (SIFT_Regions) or 144 bytes (AKAZE_Liop_Regions), 1 = float rows of length 64 (AKAZE_Float_Regions: the ten candidates are
 * ranked by L2<float> in the reference's summation order, the ratio test runs on the float distances). hash_bytes = (dim + 7) / 8:
 * CascadeHasher::Init(dimension) makes one code bit per dimension. The hashing stage of these types stays with the caller. */
int mvgx_cascade_set_regions_typed(mvgx_cascade_ctx* ctx, int scalar_type, const void* const* desc_rows, const uint8_t* const* hash_codes,
                                   const uint16_t* const* bucket_ids, const uint32_t* n_desc, uint32_t n_images, uint32_t dim,
                                   uint32_t hash_bytes, uint32_t n_groups, uint32_t bits_per_bucket);
/* The HASHING stage on the device (cascade_hasher.hpp:120-163 Init, :179-239 CreateHashedDescriptions) in place of
 * mvgx_cascade_set_regions: the projections are generated like CascadeHasher::Init(dim, n_groups, bits_per_bucket, random_seed)
 * (std::mt19937 + std::normal_distribution<>, the reference's default seed is 5489), every descriptor is centred on zero_mean
 * (dim floats: the caller's CascadeHasher::GetZeroMeanDescriptor result) and projected with the operation order of Eigen 3.4's
 * single-precision column-major matrix x vector kernel (column blocks of 16, products and sums rounded separately - no FMA, as
 * in a reference build without -mfma), so codes and bucket ids equal the reference's bit for bit (tests/test_cascade.py against
 * the compiled reference). hash_codes_out[k] (n_desc[k] x 16 bytes) / bucket_ids_out[k] (n_desc[k] x n_groups uint16) receive
 * the per-descriptor outputs when non-NULL. */
int mvgx_cascade_hash_regions(mvgx_cascade_ctx* ctx, const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images, uint32_t dim,
                              const float* zero_mean, uint32_t n_groups, uint32_t bits_per_bucket, uint32_t random_seed,
                              uint8_t* const* hash_codes_out /* may be NULL */, uint16_t* const* bucket_ids_out /* may be NULL */);
/* The same for every shape the device covers (ABI 11): scalar_type 0 = uint8 rows of 128 or 144 bytes, 1 = float rows of length 64 (the shapes
 * of mvgx_cascade_set_regions_typed); dim code bits per descriptor ((dim + 7) / 8 bytes in hash_codes_out). Eigen's kernel walks 144 columns
 * in nine blocks of 16 and 64 columns as one block (block_cols = cols below 128): the same order here, the same bits (tests/test_cascade_typed.py
 * against the compiled reference's CascadeHasher). */
int mvgx_cascade_hash_regions_typed(mvgx_cascade_ctx* ctx, int scalar_type, const void* const* desc_rows, const uint32_t* n_desc, uint32_t n_images, uint32_t dim,
                                    const float* zero_mean, uint32_t n_groups, uint32_t bits_per_bucket, uint32_t random_seed,
                                    uint8_t* const* hash_codes_out /* may be NULL */, uint16_t* const* bucket_ids_out /* may be NULL */);
int mvgx_cascade_run(mvgx_cascade_ctx* ctx, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq,
                     mvgx_match_stats* stats /* may be NULL */);
int mvgx_cascade_results(mvgx_cascade_ctx* ctx, const uint64_t** offsets, const uint32_t** ij);

/* ------------------------------------------------------------------------------------------------
 * BUNDLE ADJUSTMENT
 * replaces: sfm/sfm_data_BA_ceres.cpp:165-608 (Bundle_Adjustment_Ceres::Adjust) and, underneath it,
 *           vendored Ceres 1.13: program_evaluator.h:138-285, residual_block.cc:68-196, corrector.cc:41-155,
 *           schur_eliminator_impl.h:176-410, schur_complement_solver.cc:120-224,
 *           levenberg_marquardt_strategy.cc:65-160, trust_region_minimizer.cc:66-786
 * ---------------------------------------------------------------------------------------------- */

typedef struct mvgx_ba_ctx mvgx_ba_ctx;

/* camera models, numeric values of cameras::EINTRINSIC (cameras/Camera_Common.hpp:39-50) */
#define MVGX_CAM_PINHOLE 1          /* PINHOLE_CAMERA          params {f, ppx, ppy}              */
#define MVGX_CAM_PINHOLE_RADIAL1 2  /* PINHOLE_CAMERA_RADIAL1  params {f, ppx, ppy, k1}          */
#define MVGX_CAM_PINHOLE_RADIAL3 3  /* PINHOLE_CAMERA_RADIAL3  params {f, ppx, ppy, k1, k2, k3}  */
#define MVGX_CAM_PINHOLE_BROWN 4    /* PINHOLE_CAMERA_BROWN    params {f, ppx, ppy, k1, k2, k3, t1, t2} */
#define MVGX_CAM_PINHOLE_FISHEYE 5  /* PINHOLE_CAMERA_FISHEYE  params {f, ppx, ppy, k1, k2, k3, k4} */
#define MVGX_CAM_SPHERICAL 7        /* CAMERA_SPHERICAL: no parameter block (getParams() is empty, sfm_data_BA_ceres.cpp:366-383);
                                       the intrinsics row carries the image size {w, h} the functor needs           */
#define MVGX_BA_MAX_INTR_PARAMS 8

typedef struct mvgx_ba_problem {
  uint32_t n_poses, n_intrinsics, n_points;
  uint64_t n_obs;
  /* parameter blocks, layouts of sfm_data_BA_ceres.cpp:260-351 */
  const double* poses;        /* n_poses x 6: angle-axis(3), translation(3) with t = -R*C          */
  const double* intrinsics;   /* n_intrinsics x MVGX_BA_MAX_INTR_PARAMS (unused tail ignored)      */
  const int32_t* intr_model;  /* n_intrinsics: MVGX_CAM_*                                          */
  const double* points;       /* n_points x 3                                                      */
  /* observations (one residual block each, sfm_data_BA_ceres.cpp:354-396) */
  const uint32_t* obs_pose;   /* n_obs */
  const uint32_t* obs_intr;   /* n_obs */
  const uint32_t* obs_point;  /* n_obs */
  const double* obs_xy;       /* n_obs x 2 */
  /* constant-parameter masks (bit k set = component k of the block held constant;
   * all bits of the block set = SetParameterBlockConstant). NULL = everything free. */
  const uint8_t* pose_const_mask;   /* n_poses, bits 0..5   (Extrinsic_Parameter_Type, sfm_data_BA.hpp:28-34) */
  const uint8_t* intr_const_mask;   /* n_intrinsics, bits 0..7 (subsetParameterization, Camera_Intrinsics.hpp) */
  uint8_t points_constant;          /* Structure_Parameter_Type::NONE (sfm_data_BA.hpp:38-42)      */
  double huber_a;                   /* HuberLoss(a): sfm_data_BA_ceres.cpp:249 uses Square(4.0)=16; <=0: no loss */
  /* ---- optional (NULL / 0 = absent): ground control points and pose-centre priors ---- */
  const double* obs_weight;         /* n_obs: residual weight of WeightedCostFunction (camera_functor.hpp:35-90);
                                       0 = the unweighted functor (IntrinsicsToCostFunction's weight == 0.0 case)          */
  const uint8_t* obs_is_control;    /* n_obs: 1 = residual block of a control point: added WITHOUT loss function
                                       (sfm_data_BA_ceres.cpp:421-435) and not part of the RMSE (not a Landmark)          */
  const uint8_t* point_const_mask;  /* n_points: 1 = SetParameterBlockConstant on this point (control points, :447)       */
  uint32_t n_pose_priors;           /* PoseCenterConstraintCostFunction residuals (sfm_data_BA_ceres.cpp:44-80,454-473)   */
  const uint32_t* prior_pose;       /* n_pose_priors: pose block index                                                    */
  const double* prior_center;       /* n_pose_priors x 3: ViewPriors::pose_center_                                        */
  const double* prior_weight;       /* n_pose_priors x 3: ViewPriors::center_weight_                                      */
  double prior_huber_a;             /* HuberLoss(a) of the priors: Square(pose_center_robust_fitting_error); may be 0     */
} mvgx_ba_problem;

typedef struct mvgx_ba_options {
  int32_t max_num_iterations;   /* 50  (sfm_data_BA_ceres.cpp:120,478)           */
  double function_tolerance;    /* 1e-6 (ceres solver.h:91)                       */
  double gradient_tolerance;    /* 1e-10 (sfm_data_BA_ceres.cpp:117)              */
  double parameter_tolerance;   /* 1e-8  (sfm_data_BA_ceres.cpp:118)              */
  double initial_radius;        /* 1e4  (solver.h:84)                             */
  double max_radius;            /* 1e16                                           */
  double min_radius;            /* 1e-32                                          */
  double min_relative_decrease; /* 1e-3                                           */
  double min_lm_diagonal;       /* 1e-6                                           */
  double max_lm_diagonal;       /* 1e32                                           */
  int32_t max_consecutive_invalid_steps; /* 5                                     */
  int32_t jacobi_scaling;       /* 1                                              */
  int32_t verbose;              /* 0                                              */
} mvgx_ba_options;

typedef struct mvgx_ba_summary {
  int32_t num_iterations;         /* LM iterations executed (successful + unsuccessful)            */
  int32_t num_successful_steps;
  int32_t termination;            /* 0 convergence, 1 no-convergence (max iters), 2 failure         */
  double initial_cost, final_cost;      /* 1/2 sum rho(|r|^2), as ceres Solver::Summary             */
  double initial_rmse, final_rmse;      /* sqrt(sum |x - proj|^2 / (2 n_obs)), sfm_data_BA_test.cpp:310-330 */
  double total_ms;                /* device wall time of the solve (HIP events)                     */
  double iter_ms_mean;            /* mean time of one LM iteration                                  */
  double jacobian_ms, schur_ms, solve_ms, backsub_ms, cost_ms; /* per-phase device time summed over the context's life
                                   * when MVGX_BA_PHASE_TIMING=1 was set at create (HIP events), else 0            */
} mvgx_ba_summary;

void mvgx_ba_default_options(mvgx_ba_options* opt);
/* device >= 0: that device; -1 ("no preference", what the openMVG adapter passes): MVGX_DEVICES ("all" or a list of
 * ordinals) names the device(s) - two or more make a multi-device context when the problem has at least
 * MVGX_BA_MULTI_MIN_OBS observations (default 200 000; smaller problems use the first one) - or, unset, the current
 * device; <= -2: the current device, environment ignored. */
int mvgx_ba_create(int device, const mvgx_ba_problem* problem, mvgx_ba_ctx** out);
/* One context over several devices of THIS process (sfm_data_BA_ceres.cpp:165-608 is one call of one process): the
 * problem is cut as described under "multi-GPU" below (points + their observations partitioned by sum L_p^2, camera blocks
 * replicated, priors on the first shard), one single-device context and one host thread per shard, the per-iteration sums
 * go over RCCL (distinct devices, librccl present) or over peer-mapped device memory (MVGX_BA_TRANSPORT=rccl|peer forces
 * either; an ordinal may repeat with the peer transport). Every entry point below accepts such a context; results
 * (parameters, residuals, track angles) come back in the caller's numbering. */
int mvgx_ba_create_multi(const int* devices, int n_devices, const mvgx_ba_problem* problem, mvgx_ba_ctx** out);
int mvgx_ba_destroy(mvgx_ba_ctx* ctx);
/* Re-binds an existing context to a problem of the SAME structure and new values - what an SfM engine's consecutive Adjust() calls
 * on one scene are when nothing was added or removed in between (global_SfM.cpp's three refinement passes with growing parameter
 * sets; a caller's own re-runs; sequential_SfM.cpp:1190-1232 whenever the rejection step removed nothing): the host structure
 * build, the device allocations and the symbolic phase of the reduced solve of mvgx_ba_create are kept, only values are uploaded.
 *   structure (must be equal, compared through a 128-bit fingerprint taken at create): the counts, obs_pose / obs_intr / obs_point,
 *     intr_model, points_constant, point_const_mask, obs_is_control, presence of obs_weight, prior_pose;
 *   values (taken from `problem`): poses, intrinsics, points, obs_xy, obs_weight, prior_center / prior_weight, pose_const_mask,
 *     intr_const_mask, huber_a, prior_huber_a.
 * Returns MVGX_OK (the next mvgx_ba_solve starts from the new values with a fresh trust-region state; results are bit-identical
 * to a context created from `problem`), or MVGX_ERR_STRUCTURE when the structure differs: the context is untouched and still
 * solves its old problem; the caller destroys it and creates a new one. Works on single- and multi-device contexts. (ABI 9) */
int mvgx_ba_update(mvgx_ba_ctx* ctx, const mvgx_ba_problem* problem);
/* mvgx_ba_update with some observations switched off: `problem` still has the context's structure, obs_enabled[k] == 0 (k in the
 * order of the problem's observation arrays) removes observation k from the program - no residual, no Jacobian rows, no cost, not
 * in the RMSE, not in the track's angle - a point left without observations stops being a parameter (it keeps its value, takes no
 * step, does not count in the parameter norm) and a pose / intrinsic block left without observations leaves the program, as when the
 * reference builds its problem from a scene those observations and tracks were erased from (the rejection step of
 * sequential_SfM.cpp:1190-1232: RemoveOutliers_* erase, then Adjust() again). Equal to a context created from the reduced scene up to
 * the order of the sums (the point groups are the original ones): same iterations and decisions on the test scenes, final RMSE to
 * 1e-12. mvgx_ba_residuals / mvgx_ba_track_angles keep the indexing of `problem`. obs_enabled NULL = mvgx_ba_update; a later
 * mvgx_ba_update switches everything back on. MVGX_ERR_UNSUPPORTED on a multi-device context. (ABI 9) */
int mvgx_ba_update_subset(mvgx_ba_ctx* ctx, const mvgx_ba_problem* problem, const uint8_t* obs_enabled);
/* The library's persistent host workers (the ones the structure build of mvgx_ba_create runs on; started on first use, parked
 * between jobs), lent to the host side of a caller - the replacement TU flattens SfM_Data with them instead of starting threads
 * of its own per Adjust() call. fn(user, item, worker) runs once for every item in [0, n_items), items handed out one at a time in
 * ascending order, worker < max(1, min(max_workers or 32, hardware threads)) identifies the executing thread (0 = the caller);
 * returns when every item is done. Not re-entrant from inside fn. (ABI 9) */
typedef void (*mvgx_host_item_fn)(void* user, uint64_t item, unsigned worker);
int mvgx_host_parallel_for(uint64_t n_items, unsigned max_workers, mvgx_host_item_fn fn, void* user);
/* ---- multi-GPU (one process per GPU) --------------------------------------------------------------
 * Every rank creates its context from ITS shard of the problem: all poses and intrinsics (replicated, same
 * order on every rank) and a disjoint subset of the points together with ALL observations of those points
 * (a point's rows must be local to be eliminated, ceres schur_eliminator_impl.h:114-151). Per LM iteration
 * the solver sums across ranks: cost, camera column norms + gradient, the partial reduced camera system
 * (S, rhs), the model-cost change and the point parts of |step|^2 / |x|^2; max: gradient max-norm and the
 * failure flag. Every rank then factors the same S (no second exchange) and back-substitutes its points.
 * Either bind RCCL (mvgx_ba_comm_init: ncclAllReduce on the solver's stream, unique id from
 * mvgx_comm_unique_id on rank 0, distributed by the caller) or supply a callback transport. */
#define MVGX_REDUCE_SUM 0
#define MVGX_REDUCE_MAX 1
typedef int (*mvgx_allreduce_f64)(void* user, void* device_buffer, uint64_t count, int op, void* hip_stream);
int mvgx_ba_set_allreduce(mvgx_ba_ctx* ctx, mvgx_allreduce_f64 fn, void* user);
int mvgx_comm_unique_id(void* out128 /* 128 bytes: ncclUniqueId */);
int mvgx_ba_comm_init(mvgx_ba_ctx* ctx, int world, int rank, const void* unique_id128);
int mvgx_ba_solve(mvgx_ba_ctx* ctx, const mvgx_ba_options* opt, mvgx_ba_summary* summary);
/* one LM iteration (Jacobian + Schur + reduced solve + back-substitution + candidate cost + accept/reject) */
int mvgx_ba_lm_iteration(mvgx_ba_ctx* ctx, const mvgx_ba_options* opt, mvgx_ba_summary* summary);
/* copy the current parameter blocks back (same layouts as mvgx_ba_problem) */
int mvgx_ba_read_params(mvgx_ba_ctx* ctx, double* poses, double* intrinsics, double* points);
/* residual-only evaluation at the current parameters: cost (1/2 sum rho) and RMSE (no loss) */
int mvgx_ba_evaluate(mvgx_ba_ctx* ctx, double* cost, double* rmse);
/* |x - project(pose(X))| in pixels for every observation at the current parameters (unweighted, no loss), in the order
 * of the problem's observation arrays: the quantity RemoveOutliers_PixelResidualError (sfm/sfm_data_filters.cpp:40-73)
 * compares with its threshold in the "do { BA } while (badTrackRejector)" loops (sequential_SfM.cpp:206-210,1226-1232). */
int mvgx_ba_residuals(mvgx_ba_ctx* ctx, double* residual_norm /* n_obs */);
/* per point (track): the largest angle in degrees between the world rays of two of its observations at the current
 * poses / intrinsics (undistorted pixel -> bearing -> R^T, cameras/Camera_Intrinsics.hpp:263-280), 0 for tracks with fewer
 * than two observations: the quantity RemoveOutliers_AngleError (sfm/sfm_data_filters.cpp:77-121) compares with
 * dMinAcceptedAngle - the second half of badTrackRejector (sequential_SfM.cpp:1226-1232). */
int mvgx_ba_track_angles(mvgx_ba_ctx* ctx, double* max_angle_deg /* n_points */);

/* How the reduced camera system of this context is solved (decided at the first iteration, on the union of the ranks'
 * block patterns): the reference picks SPARSE_SCHUR above 100 poses (sfm/pipelines/sequential/sequential_SfM.cpp:1193-1205,
 * ceres schur_complement_solver.cc:241-347); here a block-sparse tile Cholesky in a nested-dissection order is used
 * whenever it needs fewer rounds of dependent launches than the dense sweep and no more tiles, the dense blocked Cholesky otherwise.
 * MVGX_BA_SOLVER=dense|sparse (environment, read at create) forces either. */
typedef struct mvgx_ba_solver_info {
  int32_t sparse;          /* 1: block-sparse tile Cholesky, 0: dense                                              */
  int32_t n_columns;       /* N = 6 n_poses + 8 n_intrinsics                                                       */
  int32_t n_padded;        /* sparse: columns after padding every part of the dissection to a multiple of 64      */
  int32_t n_parts;         /* sparse: parts of the nested dissection (incl. the dense border)                      */
  int32_t n_border_blocks; /* sparse: camera blocks ordered last as dense border (e.g. shared intrinsics)          */
  int32_t n_levels;        /* sparse: levels of the tile elimination tree = rounds of dependent launches           */
  int64_t n_factor_tiles;  /* 64 x 64 tiles of the factor's lower triangle (dense: nt (nt + 1) / 2)                */
  int64_t n_dense_tiles;   /* nt (nt + 1) / 2 with nt = ceil(N / 64)                                               */
  double flops;            /* floating-point operations of one factorisation + solve on the stored tiles           */
  /* Schur assembly: points whose pose x pose products are formed group-wise on the f64 matrix cores (groups of up to 42
   * points that share a set of at most 10 poses); the other points go through the flat product list                   */
  int32_t n_point_groups;
  int32_t n_grouped_points;
} mvgx_ba_solver_info;
int mvgx_ba_get_solver_info(mvgx_ba_ctx* ctx, mvgx_ba_solver_info* out);   /* MVGX_ERR_STATE before the first iteration */
/* The caller's choice of linear solver (ABI 9) - Bundle_Adjustment_Ceres::BA_Ceres_options::linear_solver_type_
 * (sfm/sfm_data_BA_ceres.cpp:132-146 default, :483 handed to ceres; sequential_SfM.cpp:1193-1205 and the other engines pick
 * DENSE_SCHUR or SPARSE_SCHUR by the pose count): both are direct solves of the Schur complement, what differs is the storage and
 * factorisation of the reduced camera system. AUTO is the library's rule above; DENSE the dense blocked Cholesky; SPARSE the block-
 * sparse tile Cholesky (MVGX_ERR_UNSUPPORTED when its task lists cannot be built); SPARSE_PREFERRED falls back to DENSE in that case
 * (what the replacement TU passes for SPARSE_SCHUR). Before the first iteration the symbolic phase is simply run again; afterwards
 * MVGX_OK when the solver in place is the one asked for, else MVGX_ERR_STATE. The environment's MVGX_BA_SOLVER (dense | sparse |
 * auto) outranks this call. ceres' iterative types (ITERATIVE_SCHUR, CGNR) have no counterpart: see INTEGRATION.md. */
enum { MVGX_BA_LINEAR_SOLVER_AUTO = 0, MVGX_BA_LINEAR_SOLVER_DENSE = 1, MVGX_BA_LINEAR_SOLVER_SPARSE = 2, MVGX_BA_LINEAR_SOLVER_SPARSE_PREFERRED = 3 };
int mvgx_ba_set_linear_solver(mvgx_ba_ctx* ctx, int kind);

/* ---- geometric filter of putative matches: a-contrario fundamental-matrix estimation --------------------------
 * Replaces, per image pair of a putative-match container, GeometricFilter_FMatrix_AC::Robust_estimation
 * (matching_image_collection/F_ACRobust.hpp:65-122): ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError, UnnormalizerT>
 * (robust_estimation/robust_estimator_ACRansacKernelAdaptator.hpp:104-202) + ACRANSAC
 * (robust_estimation/robust_estimator_ACRansac.hpp:339-489), as called for every pair by
 * ImageCollectionGeometricFilter::Robust_model_estimation (matching_image_collection/GeometricFilter.hpp:66-131).
 * Pair p owns the correspondences [match_start[p], match_start[p + 1]) of xI / xJ: (undistorted) pixel positions of the matched
 * features in image I / J, what MatchesPairToMat (Geometric_Filter_utils.hpp:56-64) builds, in the order of the pair's IndMatches;
 * image_wh[4 p ..] = {w_I, h_I, w_J, h_J} (View::ui_width / ui_height). Outputs: inlier_mask[m] = 1 for the geometric inliers of a
 * pair whose estimation succeeded (the reference's geometric_inliers are the putative matches with mask 1, in their order),
 * results[p]. One wave of the device runs one pair; pairs with at most 7 correspondences are rejected without estimation like the
 * reference does. Not reproduced (MVGX_ERR_UNSUPPORTED): an unbounded precision (the exhaustive NFA form), pairs with more than
 * 2^20 correspondences (up to 12 000 a wave's tables are in LDS, above that in global scratch). Parity: same sample sequence, tables and NFA arithmetic; the minimal solver's null space is computed by
 * elimination instead of Eigen's eigen-solver, so models agree to rounding, not bit for bit (parity policy). */
typedef struct mvgx_geofilter_options {
  double precision;          /* GeometricFilter_FMatrix_AC::m_dPrecision, pixels (main_GeometricFilter: 4.0)  */
  uint32_t max_iterations;   /* m_stIteration (main_GeometricFilter: 2048; constructor default 1024)          */
} mvgx_geofilter_options;
typedef struct mvgx_geofilter_result {
  double F[9];               /* m_F, row-major, in pixel coordinates (identity if no model was found)          */
  double precision_robust;   /* ACRansacOut.first -> m_dPrecision_robust (pixels)                              */
  double nfa;                /* ACRansacOut.second                                                             */
  uint32_t n_inliers;        /* |vec_inliers|                                                                  */
  uint32_t ok;               /* Robust_estimation's return value: n_inliers > 2.5 x 7                          */
} mvgx_geofilter_result;
typedef struct mvgx_geofilter_stats {
  uint64_t n_pairs, n_pairs_estimated, n_pairs_ok, n_inliers;
  double kernel_ms;          /* device time of the estimation kernels (HIP events)                             */
  double host_prepare_ms;    /* normalisation + per-pair constants on host threads                             */
  double total_ms;           /* the whole call incl. transfers                                                 */
  /* measurement (ABI version 8): what the estimation waves did, summed over the estimated pairs                  */
  uint64_t n_iterations;     /* a-contrario iterations run (sample + fit + evaluation of its models)           */
  uint64_t n_models;         /* models evaluated (F: 1 - 3 per iteration, H: 1)                                */
  uint64_t wave_clocks;      /* shader clocks (s_memtime) from the first to the last instruction of every wave */
} mvgx_geofilter_stats;
int mvgx_geofilter_f_acransac(int device, const double* xI, const double* xJ, const uint64_t* match_start, const uint32_t* image_wh,
                              uint64_t n_pairs, const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                              mvgx_geofilter_stats* stats /* may be NULL */);
/* The same estimation on a PairWiseMatches-shaped input: instead of gathered coordinates (MatchesPairToMat,
 * Geometric_Filter_utils.hpp:56-64, per pair) the caller hands over the feature positions of every image once - feat_xy, image k
 * owning rows [feat_start[k], feat_start[k + 1]) - and per pair its two images (pairs[2 p], pairs[2 p + 1]) and the index pairs
 * (i, j) of its putative matches (ij rows [match_start[p], match_start[p + 1])): exactly what Matcher_Regions::Match produces.
 * image_wh is per IMAGE here: {w, h}. The gather runs on the device; results, mask and statistics as above. Image or feature
 * indices out of range -> MVGX_ERR_ARG. */
int mvgx_geofilter_f_acransac_indexed(int device, const double* feat_xy, const uint64_t* feat_start, const uint32_t* image_wh, uint32_t n_images,
                                      const uint32_t* pairs, const uint64_t* match_start, const uint32_t* ij, uint64_t n_pairs,
                                      const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                                      mvgx_geofilter_stats* stats /* may be NULL */);

/* The homography model of the same filter: GeometricFilter_HMatrix_AC::Robust_estimation (matching_image_collection/H_ACRobust.hpp:49-113) -
 * ACKernelAdaptor<homography::kernel::FourPointSolver, AsymmetricError, UnnormalizerI> with the point-to-point error model
 * (multiview/solver_homography_kernel.cpp:37-93, solver_homography_kernel.hpp:60-64) + the same ACRANSAC. Same arguments, result
 * layout and limits as the two entries above; result.F holds m_H (x_J ~ H x_I, pixels), ok is n_inliers > 2.5 x 4, pairs with at most
 * 4 correspondences are rejected without estimation. Parity: as for F - the null vector of the 8 x 9 DLT system comes from
 * complete-pivoting elimination instead of Eigen's JacobiSVD (the same line to rounding; a rank-deficient sample has a family of
 * null vectors and the two methods pick different members). */
int mvgx_geofilter_h_acransac(int device, const double* xI, const double* xJ, const uint64_t* match_start, const uint32_t* image_wh,
                              uint64_t n_pairs, const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                              mvgx_geofilter_stats* stats /* may be NULL */);
int mvgx_geofilter_h_acransac_indexed(int device, const double* feat_xy, const uint64_t* feat_start, const uint32_t* image_wh, uint32_t n_images,
                                      const uint32_t* pairs, const uint64_t* match_start, const uint32_t* ij, uint64_t n_pairs,
                                      const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                                      mvgx_geofilter_stats* stats /* may be NULL */);
/* The essential-matrix model, GeometricFilter_EMatrix_AC (matching_image_collection/E_ACRobust.hpp:39-150; main_GeometricFilter -g e,
 * the model of the calibrated pipelines): ACKernelAdaptorEssential<FivePointSolver, EpipolarDistanceError> + ACRANSAC with samples of
 * five, up to ten models per sample (multiview/solver_essential_five_point.cpp:170-230 on the device: null space, constraint expansion,
 * Gauss-Jordan elimination, Hessenberg + shifted QR iteration of the 10 x 10 action matrix, one wave per image pair), residuals =
 * EpipolarDistanceError of F = K_J^-T E K_I^-1 on the PIXEL positions, precision in pixels (no normalisation), acceptance above
 * 2.5 x 5 inliers. Additional inputs: the cameras' bearing vectors of the positions (what Pinhole_Intrinsic::operator()(x) returns,
 * Camera_Pinhole.hpp:136-139: normalised Kinv (x, y, 1); 3 doubles per match, or per feature in the indexed form - computed by the
 * caller with the camera's own code) and the calibration matrices (row-major 3 x 3: 18 doubles per pair {K_I, K_J}, or 9 per image).
 * results[p].F receives m_E, results[p].precision_robust = ACRansacOut.first as the reference stores it (for this adaptor the SQUARED
 * pixel distance: ACKernelAdaptorEssential::unormalizeError returns its argument). Same parity policy as the F / H models. */
int mvgx_geofilter_e_acransac(int device, const double* xI, const double* xJ, const double* bearingI, const double* bearingJ,
                              const uint64_t* match_start, const uint32_t* image_wh, const double* K, uint64_t n_pairs,
                              const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                              mvgx_geofilter_stats* stats /* may be NULL */);
int mvgx_geofilter_e_acransac_indexed(int device, const double* feat_xy, const double* feat_bearing, const uint64_t* feat_start,
                                      const uint32_t* image_wh, const double* image_K, uint32_t n_images, const uint32_t* pairs,
                                      const uint64_t* match_start, const uint32_t* ij, uint64_t n_pairs, const mvgx_geofilter_options* opt,
                                      uint8_t* inlier_mask, mvgx_geofilter_result* results, mvgx_geofilter_stats* stats /* may be NULL */);

/* The essential matrix with the ANGULAR residual, GeometricFilter_ESphericalMatrix_AC_Angular<isUpright> (matching_image_collection/
 * E_ACRobust_Angular.hpp:33-191; main_GeometricFilter -g a and -g u, the models of spherical / upright rigs - ABI 9):
 * ACKernelAdaptor_AngularRadianError<Solver, AngularError> (robust_estimator_ACRansacKernelAdaptator.hpp:465-541) + ACRANSAC on the
 * cameras' bearing vectors alone - no pixel positions, image sizes or normalisation; log alpha0 = log10(1 / 2), multError = 1 / 4,
 * residual = the squared angle asin(x2 . normalized(E x1))^2 (multiview/solver_essential_eight_point.cpp:50-61). upright = 0:
 * EightPointRelativePoseSolver (:17-47, samples of eight, one model: the null vector of the 8 x 9 epipolar system), acceptance above
 * 2.5 x 8 inliers; upright = 1: ThreePointUprightRelativePoseSolver (multiview/solver_essential_three_point.cpp:84-113, samples of
 * three). opt->precision is the functor's precision_upper_bound in DEGREES (main_GeometricFilter passes 4.0): like the reference the
 * bound D2R(precision) is compared with the SQUARED angles as it is. results[p].F receives m_E scaled to unit Frobenius norm (the
 * reference's eigenvector; the sign is free), results[p].precision_robust = ACRansacOut.first (radians), inlier_mask / n_inliers / ok
 * describe the a-contrario result; the functor's second stage - RelativePoseFromEssential on those inliers (cheirality), E_ACRobust_
 * Angular.hpp:126-143 - is host work on a handful of points per pair and stays with the caller (the replacement TU runs the
 * reference's own function). Bearing vectors: 3 doubles per match, or per feature in the indexed form. */
int mvgx_geofilter_e_angular_acransac(int device, const double* bearingI, const double* bearingJ, const uint64_t* match_start, uint64_t n_pairs,
                                      int upright, const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                                      mvgx_geofilter_stats* stats /* may be NULL */);
int mvgx_geofilter_e_angular_acransac_indexed(int device, const double* feat_bearing, const uint64_t* feat_start, uint32_t n_images,
                                              const uint32_t* pairs, const uint64_t* match_start, const uint32_t* ij, uint64_t n_pairs, int upright,
                                              const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                                              mvgx_geofilter_stats* stats /* may be NULL */);
/* The ORTHOGRAPHIC essential matrix, GeometricFilter_EOMatrix_RA (matching_image_collection/Eo_Robust.hpp:35-165; main_GeometricFilter
 * -g o - ABI 9): ACKernelAdaptorEssentialOrtho<ThreePointKernel, OrthographicSymmetricEpipolarDistanceError>
 * (robust_estimator_ACRansacKernelAdaptator.hpp:384-456) + ACRANSAC with samples of three and the two closed-form models of
 * ThreePointsRelativePose per sample (multiview/solver_essential_three_point.cpp:31-79: + - x / sqrt only, evaluated in the
 * reference's order without contraction - the models are the reference's bit for bit), residual |E22 + x0 E02 + x1 E12 + y0 E20 +
 * y1 E21| (multiview/solver_essential_kernel.hpp:69-78), log alpha0 of the second image's size (point to line, scale 0.5), acceptance
 * above 2.5 x 3 inliers. xI / xJ (feat_xy in the indexed form) are the HNORMALIZED bearing vectors (b.x / b.z, b.y / b.z of what the
 * pinhole camera's operator() returns). pair_precision[p] = the bound the functor hands to ACRANSAC for pair p (Eo_Robust.hpp:96-100: the
 * mean of the two cameras' imagePlane_toCameraPlaneError(precision^2)); NULL: opt->precision for every pair. results[p].F = m_E,
 * precision_robust = ACRansacOut.first. */
int mvgx_geofilter_eo_acransac(int device, const double* xI, const double* xJ, const uint64_t* match_start, const uint32_t* image_wh,
                               const double* pair_precision, uint64_t n_pairs, const mvgx_geofilter_options* opt, uint8_t* inlier_mask,
                               mvgx_geofilter_result* results, mvgx_geofilter_stats* stats /* may be NULL */);
int mvgx_geofilter_eo_acransac_indexed(int device, const double* feat_xy, const uint64_t* feat_start, const uint32_t* image_wh, uint32_t n_images,
                                       const uint32_t* pairs, const uint64_t* match_start, const uint32_t* ij, const double* pair_precision,
                                       uint64_t n_pairs, const mvgx_geofilter_options* opt, uint8_t* inlier_mask, mvgx_geofilter_result* results,
                                       mvgx_geofilter_stats* stats /* may be NULL */);

/* ------------------------------------------------------------------------------------------------------------------------
 * Guided matching (ABI 10): the second stage of the a-contrario geometric filters - SURVEY.md 8(f) N2's files
 *   /root/reference/src/openMVG/robust_estimation/guided_matching.hpp:178-227  GuidedMatching(model, camL, lRegions, camR, rRegions, ...)
 *   /root/reference/src/openMVG/matching_image_collection/F_ACRobust.hpp:109-152, E_ACRobust.hpp:153-215, H_ACRobust.hpp:136-180
 * For every pair (I, J) and every feature i of I: over the features j of J whose geometric error under the pair's model is below
 * error_th[pair] (kind FUNDAMENTAL: EpipolarDistanceError of F - the essential functor passes F = K2^-T E K1^-1; kind HOMOGRAPHY:
 * AsymmetricError of H), the smallest and second smallest SquaredDescriptorDistance (L2<uint8_t>, exact); (i, j_best) is a match iff a
 * second one exists and best < dist_ratio_sq * second (double arithmetic, guided_matching.hpp:68-112). Matches of a pair in ascending
 * i (what IndMatch::getDeduplicated leaves). feat_xy: the positions the reference would compare - camL->get_ud_pixel(position) or the
 * position itself - as doubles, image k owning rows [feat_start[k], feat_start[k + 1]); desc: desc_bytes (64, 128 or 144) per feature,
 * same rows. A pair whose error_th is not a positive finite number gives no match (the functors' m_dPrecision_robust != infinity
 * test). match_start: n_pairs + 1 offsets (out); *matches_ij: 2 uint32 per match (i, j), allocated by the library - release it with
 * mvgx_host_free. */
#define MVGX_GUIDED_FUNDAMENTAL 0
#define MVGX_GUIDED_HOMOGRAPHY 1
typedef struct mvgx_guided_stats {
  uint64_t n_pairs, n_matches;
  uint64_t n_geometric_tests;     /* (i, j) evaluated                                                              */
  uint64_t n_geometric_passed;    /* ... with an error below the bound: descriptor distances that counted          */
  uint64_t n_descriptor_stages;   /* wave-iterations in which the descriptor stage ran (some lane had passed)      */
  double kernel_ms;               /* device time of the norm + matching kernels (HIP events)                       */
  double total_ms;                /* the whole call incl. transfers                                                 */
} mvgx_guided_stats;
int mvgx_guided_match_u8(int device, const double* feat_xy, const uint8_t* desc, uint32_t desc_bytes, const uint64_t* feat_start, uint32_t n_images,
                         const uint32_t* pairs, const double* models /* 9 per pair, row-major */, const double* error_th /* per pair */,
                         uint64_t n_pairs, int kind, double dist_ratio_sq, uint64_t* match_start, uint32_t** matches_ij,
                         mvgx_guided_stats* stats /* may be NULL */);
/* The same for every region type the reference's functors can meet (ABI 12; Regions::SquaredDescriptorDistance is virtual,
 * guided_matching.hpp:213): desc_type MVGX_DESC_U8 = mvgx_guided_match_u8; MVGX_DESC_F32 = float rows of 64 or 128 values (desc_bytes
 * 256 / 512: AKAZE_Float_Regions, scalar_regions.hpp:107-116 + L2<float>, metric.hpp:95-131 - the reference's float sums in its order, no
 * fused multiply-add, widened to double: equal lists, not a tolerance); MVGX_DESC_BINARY = rows of 32 or 64 bytes under the SQUARED
 * Hamming distance (AKAZE_Binary_Regions, binary_regions.hpp:109-120). Everything else as above. */
#define MVGX_DESC_U8 0
#define MVGX_DESC_F32 1
#define MVGX_DESC_BINARY 2
int mvgx_guided_match(int device, const double* feat_xy, const void* desc, int desc_type, uint32_t desc_bytes, const uint64_t* feat_start,
                      uint32_t n_images, const uint32_t* pairs, const double* models, const double* error_th, uint64_t n_pairs, int kind,
                      double dist_ratio_sq, uint64_t* match_start, uint32_t** matches_ij, mvgx_guided_stats* stats /* may be NULL */);
void mvgx_host_free(void* p);   /* releases an array the library allocated for the caller (mvgx_guided_match{,_u8}) */

#ifdef __cplusplus
}
#endif
#endif /* MVGX_H_ */
