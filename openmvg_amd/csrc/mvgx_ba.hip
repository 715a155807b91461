// libmvgx_hip.so — Levenberg-Marquardt bundle adjustment on gfx950 (fp64), no Ceres / no Eigen on the device.
//
// Reference path reproduced (paths under /root/reference/src; "ceres/" = third_party/ceres-solver/internal/ceres):
//   openMVG/sfm/sfm_data_BA_ceres.cpp:242-473      problem: one residual block per observation (intrinsic, pose, point),
//                                                   weighted loss-free residuals of ground control points (:398-451),
//                                                   pose-centre priors (:44-80, :454-473)
//   openMVG/sfm/sfm_data_BA_ceres_camera_functor.hpp  pinhole / radial K1 / radial K3 / Brown T2 / fisheye / spherical residuals
//   ceres/residual_block.cc:68-196, corrector.cc     Huber loss correction of residuals and Jacobians
//   ceres/trust_region_minimizer.cc:66-786           LM loop (Jacobi scaling frozen at iteration 0, step acceptance,
//                                                     parameter / function / gradient tolerances, invalid steps)
//   ceres/levenberg_marquardt_strategy.cc:65-160     D = sqrt(clamp(diag(J^T J)) / radius), radius update rules
//   ceres/schur_eliminator_impl.h:176-410            S = F^T F + D_f^2 - sum_p (E^T F)^T (E^T E + D_p^2)^-1 (E^T F), rhs, back-sub
//   ceres/schur_complement_solver.cc:180-224         dense Cholesky of the reduced camera system
//
// Device data layout (all fp64):
//   observations sorted by 3-D point (CSR pt_start), so a point's rows are contiguous like Ceres' chunks;
//   Jacobian as three arrays of per-observation records (loss-corrected, UNscaled), so that both the point-ordered
//     kernels and the gathers by pose / intrinsic read whole cache lines:
//     JA[o] = {r(2), E = d r/d point (2x3)} (64 B), JB[o] = {r(2), Fc = d r/d pose (2x6), pad} (128 B),
//     JC[o] = {Fi = d r/d intrinsic (2x8)} (128 B);
//   reduced camera system S: n = 6 n_poses + 8 n_intr columns (pose blocks first, then intrinsic blocks), leading
//     dimension n + 1; memory is simultaneously "row-major upper + rhs in column n" (how the assembly kernels write it)
//     and "column-major lower + rhs in row n" (how the Cholesky kernels read it). Constant / unused parameter
//     components keep their slot with Jacobi scale 0, unit diagonal and zero rhs, so they decouple exactly.
//
// The reduced system is S = G - Z^T Z with
//   G = Fs^T Fs, the Gram blocks of the scaled camera columns. They depend on the Jacobian only: accumulated once per
//       Jacobian evaluation (ba_pi_gram / ba_intr_gram), reused when only the LM radius changes;
//   Z = L_p^-1 Es^T Fs per point, V_p = Es^T Es + D_p^2 = L_p L_p^T: one 3 x 6 block per observation (pose columns) and
//       one 3 x 8 block per (point, intrinsic) slot (intrinsic columns). V^-1 never appears: Z^T Z = Y^T V^-1 Y.
// Every product -Z_a^T Z_b lands in the block (camera block of a, camera block of b). The products are listed once at
// create time, sorted by destination block and cut into chunks; one wave per chunk accumulates its products in
// registers (no atomics, fixed order), a second kernel sums the chunks of each block and writes it into S.
// Cholesky: per 64-column block step chol_diag_inv (factor the diagonal block, invert its factor), chol_panel_mfma
// (L21 = A21 L11^-T as a GEMM; the rhs row rides along = forward substitution), chol_update_mfma (A22 -= L21 L21^T,
// v_mfma_f64_16x16x4_f64); back substitution one launch per block step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <numeric>
#include <pthread.h>
#include <thread>
#include <type_traits>
#include <vector>

#include "ba_math.h"
#include "ba_sparse_plan.h"
#include "mvgx_ba_multi.h"
#include "mvgx_comm.h"
#include "mvgx_common.h"

namespace {

using mvgx::set_error;
using namespace mvgx_ba;

constexpr int kJA = 8, kJB = 16, kJC = 16;   // doubles per observation record: {r, E} | {r, Fc, pad} | {Fi}
constexpr int kPiChunk = 512;      // observations per workgroup of the (pose, intrinsic) Gram kernel
using d4_t = __attribute__((ext_vector_type(4))) double;   // accumulator of v_mfma_f64_16x16x4_f64
constexpr int kTripChunk = 1024;   // (entity a, entity b) products per wave of the Schur-product kernel
constexpr int kPoseGram = 27;      // per pose: Fc^T Fc upper triangle (21) | Fc^T r (6)
constexpr int kPiGram = 75;        // per (pose, intrinsic) pair: the 27 above | Fc^T Fi (6 x 8)
constexpr int kIntrGram = 44;      // per intrinsic: Fi^T Fi upper triangle (36) | Fi^T r (8)
constexpr int kPriorJ = 21;        // per pose-centre prior: corrected r (3) | corrected d r / d pose (3 x 6)

// kSCamStepSq..kSModelCam are contiguous (one reduction writes all six: ba_step_scalars_kernel; kSStepSq..kSModelPt are the rank-local
// point parts, summed over the ranks in one all-reduce); the camera parts are replicated on every rank, the
// point parts are rank-local and summed across ranks.
enum Scalar { kSCost = 0, kSSqErr, kSModel, kSCamStepSq, kSCamXSq, kSStepSq, kSXSq, kSModelPt, kSModelCam, kSGmax, kSFail, kSNobs, kSGmaxGrp, kSCount = 14 };

// One list of Schur products -Z_a^T Z_b, sorted by the (row block, column block) of S they add into. Entities are
// observations (pose blocks, width 6) or (point, intrinsic) slots (intrinsic blocks, width 8).
struct TripList {
  uint32_t n_trips = 0, n_chunks = 0, n_blocks = 0;
  uint2* trips = nullptr;             // (a, b)
  uint32_t* chunk_lo = nullptr;       // n_chunks: first product
  uint32_t* chunk_hi = nullptr;       // n_chunks: one past the last
  uint8_t* chunk_diag = nullptr;      // n_chunks: chunk of a diagonal block (carries the rhs)
  uint32_t* block_row = nullptr;      // n_blocks: camera block index (pose i -> i, intrinsic k -> n_poses + k)
  uint32_t* block_col = nullptr;
  uint32_t* block_chunk0 = nullptr;   // n_blocks + 1
  int32_t* block_own = nullptr;       // pose x intrinsic lists: the (pose, intrinsic) pair whose Fc^T Fi adds in, or -1
  double* part = nullptr;             // (n_chunks + n_ext) x (WA * WB + WA)
  // partial blocks written by the point-group kernel (GroupList), numbered in block order after the n_chunks chunks of the flat list
  uint32_t n_ext = 0;
  uint32_t* block_ext0 = nullptr;     // n_blocks + 1 (null: none)
};

// Point groups - the fused path. Free points whose observations fall on at most kGroupCams distinct poses and kGroupIntr distinct
// intrinsics are processed in groups of up to kGroupPts points that share such a camera set; consecutive groups with the SAME
// camera set form a supergroup = one workgroup. Nothing of the Jacobian is stored for these points: the workgroup evaluates the
// residuals and closed-form Jacobians of the group's observations from (observation, parameters) in registers (one observation
// per thread), reduces V_p / gradient / column norms per point in LDS, factors V_p, forms the Z blocks, stages them as a dense
// (3 n_points) x 80 matrix (6 columns per local pose, 8 per local intrinsic, column kGroupHCol = h_p, zero where a point does not
// see a camera) and computes Z^T Z - all Schur products of the group: pose x pose, pose x intrinsic, intrinsic x intrinsic and the
// rhs - on the f64 matrix cores, accumulating over the groups of the supergroup. The back-substitution runs the same code up to the
// Z blocks again (mode kGroupBacksub) instead of reading stored Z blocks: per LM iteration the points' observations are read twice
// (24 bytes each), no per-observation record is written. Long tracks, constant points, poses seen twice by a point and points
// with more intrinsics than kGroupIntr stay on the record-based path below (JA / JB / JC records, flat product lists).
constexpr int kGroupCams = 16;                                    // local poses per group AT MOST (tables, pair numbering): the wide form
constexpr int kNarrowCams = 10;                                   // ... of the usual form: 60 pose columns
constexpr int kGroupIntr = 2;                                     // local intrinsics per group: 16 intrinsic columns
// The staged matrix has one of two forms, chosen per supergroup (round 6: points seen by 11 .. 16 poses used to leave the fused path):
//   usual: up to kNarrowCams local poses - 60 pose columns | 16 intrinsic columns | h_p (column kGroupHCol = 76) in rows of kGroupCols = 80
//          doubles (5 column tiles, 15 tiles of Z^T Z; the compact form of a pinhole-size single intrinsic: h_p in column 63, 10 tiles),
//          up to kGroupPts points, the tiles accumulated in registers over the groups of the supergroup;
//   wide:  a local pose beyond kNarrowCams in use - 96 pose columns | 16 | h_p (column kWideHCol = 112) in rows of kWideCols = 128 doubles
//          (up to 8 column tiles, 36 tiles), up to kWidePts points (the same LDS region), a wave's tiles formed one
//          after the other and added straight to the partial blocks in memory (a supergroup of several groups: read - add - write by the same lane).
constexpr int kGroupHCol = 6 * kNarrowCams + 8 * kGroupIntr;      // the column of h_p (76)
constexpr int kGroupCols = 80;                                    // 5 x 16: the columns 77..79 are padding (zero)
constexpr int kGroupColTiles = kGroupCols / 16;
constexpr int kGroupTiles = kGroupColTiles * (kGroupColTiles + 1) / 2;   // upper 16 x 16 tiles of Z^T Z (15)
constexpr int kWideHCol = 6 * kGroupCams + 8 * kGroupIntr;        // 112
constexpr int kWideCols = 128;
constexpr int kWideColTiles = kWideCols / 16;
constexpr int kWideTiles = kWideColTiles * (kWideColTiles + 1) / 2;      // 36
constexpr int kGroupPairsPP = kGroupCams * (kGroupCams + 1) / 2;  // destination blocks of a group: pose x pose (136)
constexpr int kGroupPairsPI = kGroupCams * kGroupIntr;            // pose x intrinsic (32)
constexpr int kGroupPairsII = kGroupIntr * (kGroupIntr + 1) / 2;  // intrinsic x intrinsic (3)
constexpr int kNVpp = 6 * 6 + 6, kNVpi = 6 * 8 + 6, kNVii = 8 * 8 + 8;   // doubles per partial block (block | rhs) of the three product families
#ifndef MVGX_GROUP_PTS
#define MVGX_GROUP_PTS 32
#endif
constexpr int kGroupPts = MVGX_GROUP_PTS;                         // 3 x points rows of the staged matrix
#ifndef MVGX_GROUP_THREADS
#define MVGX_GROUP_THREADS 256
#endif
constexpr int kGroupThreads = MVGX_GROUP_THREADS;                 // one observation per thread: a group holds at most this many observations
constexpr int kGroupWaves = kGroupThreads / 64;
constexpr int kGroupTilesPerWave = (kGroupTiles + kGroupWaves - 1) / kGroupWaves;
// The staged matrix lies ROW-major in LDS (round 6; column-major with a padded column stride before): the six values a thread forms for
// a row of its pose's columns are three 16-byte writes side by side with its neighbours' (the column-major form scattered 18 8-byte
// writes of a thread over 18 columns, four distinct bank groups per half wave), and a half wave's MFMA operand - 16 consecutive
// columns of two rows - is two contiguous 128-byte runs 640 bytes apart: every bank twice, the minimum for 64 x 8 bytes.
constexpr int kGroupCS = kGroupCols;                              // doubles between rows (usual form; the wide form: kWideCols)
constexpr int kGroupRows = 3 * kGroupPts;
constexpr int kWidePts = kGroupRows * kGroupCols / kWideCols / 3; // 20: the wide form's rows fill the same region
static_assert(kGroupPts % 4 == 0 && kGroupPts <= 255 && kGroupIntr == 2 && (kGroupCS * 8) % 256 == 128 && (6 * 8) % 16 == 0, "group tile layout (the slot ranges assume two local intrinsics)");
static_assert(kGroupThreads >= 128 + 6 * kGroupCams + 8 * kGroupIntr + 8 * kGroupIntr && kGroupCams <= 16, "the tables of a supergroup are staged by thread ranges 0.., 64.., 128.., 240..; four bits of local pose in an entry word");
static_assert(kWidePts * 3 * kWideCols <= kGroupRows * kGroupCols && kWidePts * kGroupCams >= kGroupThreads, "the wide form in region M");
constexpr int kGroupOut = kGroupPairsPP * kNVpp + kGroupPairsPI * kNVpi + kGroupPairsII * kNVii;   // partial blocks of a supergroup on their way out
constexpr int kGroupM = kGroupRows * kGroupCS;                    // region M: the staged matrix; before that the per-observation terms (24 x threads)
static_assert(kGroupM >= 24 * (kGroupThreads + 1) && kGroupM >= kGroupOut && kGroupM >= 3 * (kGroupThreads + 1) + 3 * kGroupPts * kGroupIntr * 8, "LDS region M");
constexpr int kGroupSums = kGroupPts * 20, kGroupPtab = kGroupPts * 12;
constexpr int kGroupCamRow = 6 + kPoseTrig + 6;                     // per local pose: parameters | rotation terms | column scales
constexpr int kGroupIntrRow = 8 + 8;                              // per local intrinsic: parameters | column scales
constexpr int kGroupCandRow = 6 + kPoseTrig;                       // per local pose of the candidate x + delta: parameters | rotation terms
constexpr int kGroupCand = kGroupCams * kGroupCandRow + kGroupIntr * 8 + 3 * kGroupPts;   // back-substitution with the candidate's cost: the candidate's cameras, the point threads' running sums
constexpr int kGroupTabs = kGroupCams * kGroupCamRow + kGroupIntr * kGroupIntrRow + 6 * kGroupCams + 8 * kGroupIntr + kGroupCand;   // ... the solution's components, the candidate
constexpr int kGroupLds = (kGroupM + kGroupSums + kGroupPtab + kGroupTabs) * (int)sizeof(double) + (kGroupThreads + 2 * (kGroupPts + 1) + kGroupIntr + 3 + kGroupPairsPP + kGroupPairsPI + kGroupPairsII) * (int)sizeof(uint32_t);
// The norms and back-substitution modes never stage the matrix: their region M holds only the per-observation terms of the point
// sums (18 x (threads + 1) doubles), which lets a third workgroup onto the CU (49 KB instead of 74 KB each).
constexpr int kGroupMSmall = (18 * (kGroupThreads + 1) + 1) & ~1;
constexpr int kGroupLdsSmall = kGroupLds - (kGroupM - kGroupMSmall) * (int)sizeof(double);
template <int MODE> constexpr int group_m_doubles() { return MODE == 1 /* kGroupForward */ ? kGroupM : kGroupMSmall; }
template <int MODE> constexpr int group_lds_bytes() { return MODE == 1 ? kGroupLds : kGroupLdsSmall; }
constexpr int kGroupMinPts = 8;                                   // smaller groups go to the record-based path
constexpr uint32_t kNoChunk = 0xFFFFFFFFu;
enum GroupMode { kGroupNorms = 0, kGroupForward = 1, kGroupBacksub = 2 };
struct GroupList {
  uint32_t n_sg = 0, n_groups = 0;
  uint32_t* sg_start = nullptr;    // n_sg + 1 -> groups
  uint32_t* sg_order = nullptr;    // n_sg: workgroup -> supergroup, largest first (the dispatcher hands workgroups out in index order: the
                                   // short ones then fill the tail of the launch instead of a long one starting last)
  uint32_t* obs_start = nullptr;   // n_groups + 1 -> entries (one per observation, point by point)
  uint32_t* pt_start = nullptr;    // n_groups + 1 -> pts
  uint32_t* pts = nullptr;         // grouped point -> point
  uint32_t* pt_estart = nullptr;   // grouped point (+ 1) -> its first entry
  uint32_t* pt_ksplit = nullptr;   // grouped point -> the first of its entries with local intrinsic 1 (entries of a point: intrinsic 0 first)
  uint32_t* eq = nullptr;          // entry: local point | local pose << 8 | local intrinsic << 12
  uint32_t* eobs = nullptr;        // entry: observation index (weights / control flags)
  double2* exy = nullptr;          // entry: the observation
  uint32_t* cams = nullptr;        // n_sg x kGroupCams: pose of a local pose (unused: 0)
  uint32_t* intrs = nullptr;       // n_sg x kGroupIntr
  uint32_t* chunk_pp = nullptr;    // n_sg x kGroupPairsPP: row of tpp.part for local poses (x <= y), kNoChunk: no common point
  uint32_t* chunk_pi = nullptr;    // n_sg x kGroupPairsPI: row of tpi.part for (local pose x, local intrinsic k)
  uint32_t* chunk_ii = nullptr;    // n_sg x kGroupPairsII: row of tii.part for local intrinsics (k <= k')
  double* gmax_part = nullptr;     // n_sg: max |gradient| over the supergroup's points (forward pass)
  uint32_t* ungrouped = nullptr;   // observations of the points outside every group (record-based path)
  uint32_t n_ungrouped = 0;
};

// Block-sparse storage of the reduced camera system (ba_sparse_plan.h): 64 x 64 tiles of the permuted, padded matrix,
// lower triangle, only the tiles that are non-zero in S or fill in its factor; tile row nT carries the rhs (row 0 of each
// of its tiles). Inside a tile element (r, c) lives at c * 64 + r.
struct SpSys {
  int enabled = 0;
  int nT = 0, n_slots = 0, n_levels = 0;
  const int32_t* pcol = nullptr;      // N: original scalar column -> padded permuted column
  const int32_t* tmap = nullptr;      // (nT + 1) x nT: tile (I, J), I >= J -> slot, -1 = structurally zero
  const int32_t* tile_kb = nullptr;   // nT: valid columns of a diagonal tile (the rest is identity padding)
  double* A = nullptr;                // n_slots x 4096: S, updated in place by the sweep
  double* L = nullptr;                // n_slots x 4096: the factor's tiles below the diagonal and the forward-substituted rhs
  double* Linv = nullptr;             // nT x 4096: inverses of the diagonal factor tiles
  double* z = nullptr;                // nT x 64: solution in padded order
  const mvgx_sparse::GemmTask *t_tasks = nullptr, *u_tasks = nullptr;
  const mvgx_sparse::SlotPair *t_pairs = nullptr, *u_pairs = nullptr;
  const int32_t *f_cols = nullptr, *bs_start = nullptr, *bs_slot = nullptr, *bs_row = nullptr;
  const int32_t* level_of = nullptr;  // nT: level of a tile column in the elimination tree
  // look-ahead schedule (ba_sparse_plan.h): column of a slot; per tile column the level-before contributions to its diagonal tile
  const int32_t *slot_col = nullptr, *pre_start = nullptr, *pre_slot = nullptr, *pre_col = nullptr;
  const int32_t* bs_rec = nullptr;    // nT x 16: what a workgroup of the reverse sweep needs to start, per position in f_cols (ba_sparse_plan.h)
  unsigned* z_flag = nullptr;         // nT: the solve (epoch) whose z_k is in s.z - the one-launch reverse sweep (sp_backsolve_all_kernel)
  double* u_scratch = nullptr;        // partial sums of the U tasks with split contributor lists: 256 doubles per chunk
  unsigned* u_counter = nullptr;      // arrivals per split group (left at zero by the last arrival)
};

struct Dev {
  // sizes
  uint32_t n_poses = 0, n_intr = 0, n_pts = 0, n_priors = 0;
  uint64_t n_obs = 0;
  int N = 0, LD = 0;             // camera system size and leading dimension
  int n_islots = 0;              // (point, intrinsic) slots
  int n_pi = 0, n_pichunks = 0;
  double huber_a = 0, prior_huber_a = 0;
  // parameters
  double *poses = nullptr, *intr = nullptr, *pts = nullptr;      // x_
  double *cposes = nullptr, *cintr = nullptr, *cpts = nullptr;   // candidate_x_
  int* model = nullptr;
  // observations (sorted by point)
  uint32_t *opose = nullptr, *ointr = nullptr, *opt = nullptr;
  double* oxy = nullptr;
  double* oweight = nullptr;          // n_obs or null
  uint8_t* octrl = nullptr;           // n_obs or null
  uint8_t* odisabled = nullptr;       // n_obs or null (mvgx_ba_update_subset): the observation contributes nothing - residual, Jacobian rows, cost, ray
  uint32_t* oorig = nullptr;          // n_obs: index of the observation in the caller's arrays (null: identity)
  uint32_t* pt_start = nullptr;       // n_pts + 1
  uint32_t* ptk_start = nullptr;      // n_pts + 1 -> intrinsic slots of a point
  uint32_t* slot_intr = nullptr;      // n_islots
  uint32_t* slot_point = nullptr;     // n_islots
  uint8_t* cam_active = nullptr;      // N: free component of a block that has residuals
  uint8_t* cam_counts = nullptr;      // N: component belongs to a block that is in the reduced program (x-norm)
  uint8_t* pt_free = nullptr;         // n_pts: point is a free parameter block with residuals
  uint8_t* pt_grouped = nullptr;      // n_pts: point belongs to a point group (fused path: no records, no stored Z); null: no groups
  // (pose, intrinsic) pairs and intrinsics: observation lists for the Gram kernels
  uint32_t *pi_obs = nullptr, *pichunk_lo = nullptr, *pichunk_hi = nullptr, *pi_chunk0 = nullptr, *pose_pi_start = nullptr;
  uint32_t *pichunk_pose = nullptr, *pichunk_intr = nullptr;   // n_pichunks: the pair of the chunk
  uint32_t* pi_pt = nullptr;          // n_obs, (pose, intrinsic) order: point of the observation
  double2* pi_xy = nullptr;           // n_obs, (pose, intrinsic) order: the observation
  // the intrinsic's own Gram blocks come out of the same pass as the (pose, intrinsic) blocks; per intrinsic the list of the
  // (pose, intrinsic) chunks that belong to it
  uint32_t *intr_pichunk_start = nullptr, *intr_pichunk = nullptr;
  double* pichunk_ipart = nullptr;   // n_pichunks x kIntrGram
  double* intr_slice_part = nullptr; // n_intr x kIntrSlices x kIntrGram: slice sums of ba_intr_finish_kernel
  unsigned* intr_arrivals = nullptr; // n_intr: workgroups of ba_intr_finish_kernel that have delivered their slice (left at zero)
  // pose-centre priors
  uint32_t *prior_pose = nullptr, *pose_prior_start = nullptr, *pose_prior_idx = nullptr;
  double *prior_center = nullptr, *prior_weight = nullptr, *Jprior = nullptr;
  // Jacobian and derived
  double *JA = nullptr, *JB = nullptr, *JC = nullptr;   // n_obs records each
  double *cn_cam = nullptr, *g_cam = nullptr, *scale_cam = nullptr, *diag_cam = nullptr;   // N
  double *cn_pt = nullptr, *g_pt = nullptr, *scale_pt = nullptr, *diag_pt = nullptr;       // 3 n_pts
  double *pichunk_part = nullptr, *pi_gram = nullptr, *pose_gram = nullptr, *igram = nullptr;
  double *Linv3 = nullptr, *hp = nullptr;   // per point: L_p^-1 (6, lower) and h_p = L_p^-1 Es^T r (3)
  double *Zpose = nullptr, *Zint = nullptr; // n_obs x 18, n_islots x 24
  TripList tpp, tpi, tii;
  GroupList grp;
  double* S = nullptr;                // N x LD (dense mode only)
  SpSys sp;                           // block-sparse mode
  // multi-rank exchange of S: the union over ranks of the non-zero camera blocks, packed contiguously (+ the rhs column)
  uint32_t n_ublocks = 0;
  uint64_t n_packed = 0;
  uint32_t *ublk_row = nullptr, *ublk_col = nullptr;   // first row / column of the block in S
  uint8_t *ublk_h = nullptr, *ublk_w = nullptr;        // 6 or 8
  uint64_t* ublk_off = nullptr;
  double* packed = nullptr;
  double* linv = nullptr;             // per Cholesky block step: L11^-1 k-major (64 x 64), then row-major (64 x 64)
  double *zsol = nullptr, *step_cam = nullptr, *step_pt = nullptr;
  double asm_inv_radius = 0.0;        // != 0: the assemble launch also applies the LM diagonal (ba_finish_system_kernel's work), with this 1 / radius
  double* part = nullptr;             // partial sums (reductions)
  double* grp_part = nullptr;         // n_sg x 5: the supergroups' sums of the back-substitution pass with the candidate (ba_step_reduce_kernel)
  double* scalars = nullptr;          // kSCount
  int* fail = nullptr;
  const int* gate = nullptr;          // non-null: a Jacobian evaluation launched AHEAD of the host's accept decision - its kernels leave at once unless *gate (the device's own decision, ba_publish_scalars_kernel)
};

// ------------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) t += sh[i];
  return t;  // valid in thread 0
}
__device__ __forceinline__ double block_max(double v, double* sh) {
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) t = fmax(t, sh[i]);
  return t;
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int N>
__device__ __forceinline__ void load_rec(const double* __restrict__ p, double* v) {   // N doubles, 16-byte aligned
  const double2* __restrict__ q = reinterpret_cast<const double2*>(p);
#pragma unroll
  for (int k = 0; k < N / 2; ++k) { const double2 t = q[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
}

// Element (row, col) of the reduced system, row <= col in the original numbering, col == N: the rhs. Dense mode: the
// row-major upper triangle (= column-major lower, what the dense Cholesky reads). Sparse mode: the tile of the permuted matrix.
__device__ __forceinline__ double* sys_elem(const Dev& d, int row, int col) {
  if (!d.sp.enabled) return d.S + (size_t)row * d.LD + col;
  const int pr = d.sp.pcol[row];
  if (col == d.N) return d.sp.A + (size_t)d.sp.tmap[(size_t)d.sp.nT * d.sp.nT + (pr >> 6)] * 4096 + (size_t)(pr & 63) * 64;
  const int pc = d.sp.pcol[col];
  const int hi = pr > pc ? pr : pc, lo = pr > pc ? pc : pr;
  return d.sp.A + (size_t)d.sp.tmap[(size_t)(hi >> 6) * d.sp.nT + (lo >> 6)] * 4096 + (size_t)(lo & 63) * 64 + (hi & 63);
}

// out[k] = sum_i part[i * stride + k], k < nk (single block; deterministic order)
// One workgroup of 1024 threads folds the per-workgroup partials of a kernel: every thread takes a contiguous slice of the
// partials (fixed order: slice by slice, then wave by wave), eight loads in flight at a time - as a strided loop of 256 threads
// over 20k partials this took 24 microseconds, five times per LM iteration.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const double* __restrict__ part, int n, int stride, int nk,
                                                               double* __restrict__ out, int out_off, int is_max) {
  __shared__ double sh[16];
  // thread t takes rows t, t + 1024, ...: neighbouring lanes read neighbouring rows (a run of consecutive rows per thread made
  // every load a separate line; 25 microseconds for the 20 000 partials of C5), eight loads in flight
  for (int k = 0; k < nk; ++k) {
    double v = 0;
    int i = (int)threadIdx.x;
    for (; i + 7 * 1024 < n; i += 8 * 1024) {
      double t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = part[(size_t)(i + q * 1024) * stride + k];
#pragma unroll
      for (int q = 0; q < 8; ++q) v = is_max ? fmax(v, t[q]) : v + t[q];
    }
    for (; i < n; i += 1024) { const double t = part[(size_t)i * stride + k]; v = is_max ? fmax(v, t) : v + t; }
    const double t = is_max ? block_max(v, sh) : block_sum(v, sh);
    if (threadIdx.x == 0) out[out_off + k] = t;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------
// linearize: residual (+ Jacobian) per observation, weight, Huber loss correction, cost partial sums
// ------------------------------------------------------------------------------------------------------
// A wave's 64 records of SLOTS 16-byte slots each leave through LDS: every lane drops its record (padded stride: no bank
// conflicts), then lane l stores slots l, l + 64, ... of the wave's contiguous block - 1 KiB per store instruction instead of
// 64 partial lines (thread-per-record stores at a 64 / 128-byte stride cost four times their bytes in L2 write requests).
template <int SLOTS>
__device__ __forceinline__ void store_records_coalesced(double2* lds_wave /* 64 x (SLOTS + 1) */, const double2 (&v)[SLOTS], bool active,
                                                        double2* __restrict__ wave_base, int n_valid) {
  const int lane = threadIdx.x & 63;
  if (active) {
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) lds_wave[lane * (SLOTS + 1) + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < SLOTS; ++t) {
    const int i = t * 64 + lane, rec = i / SLOTS, slot = i - rec * SLOTS;
    if (rec < n_valid) wave_base[i] = lds_wave[rec * (SLOTS + 1) + slot];
  }
  __syncthreads();
}

// Loss-corrected residual and Jacobian of one observation at the current parameters, as ba_linearize_kernel forms them:
// WeightedCostFunction (camera_functor.hpp:35-90; weight 0 selects the unweighted functor), control points without loss
// function, Huber corrector (corrector.cc:81-85,126-129: residual and Jacobian scale by sqrt(rho')). o: the observation's index
// in the point-sorted arrays (weights / control flags). Returns r (corrected) and the factor sc of the Jacobian entries.
// An observation that mvgx_ba_update_subset switched off contributes NOTHING: zero residual, zero scale of its rows. The rows themselves
// must be finite for that product to be zero - the callers evaluate such an observation with `off` (ba_math.h: eval_observation_t), which
// keeps its projection regular whatever the stale point is (inf * 0 = NaN reached the cost and the Gram blocks).
__device__ __forceinline__ bool observation_is_off(const Dev& d, uint64_t o) { return d.odisabled && d.odisabled[o]; }
__device__ __forceinline__ double correct_observation(const Dev& d, uint64_t o, double (&r)[2]) {
  if (observation_is_off(d, o)) { r[0] = 0.0; r[1] = 0.0; return 0.0; }   // as if the observation were not in the problem
  double w = 1.0;
  if (d.oweight) { const double ww = d.oweight[o]; if (ww != 0.0) w = ww; }
  const bool ctrl = d.octrl && d.octrl[o];
  r[0] *= w; r[1] *= w;
  const double s = r[0] * r[0] + r[1] * r[1];
  double rho[3];
  huber_rho_on(!ctrl && d.huber_a > 0.0, d.huber_a, s, rho);
  const double sr = corrector_scale(rho);
  r[0] *= sr; r[1] *= sr;
  return sr * w;
}

// The records of the listed observations only (the points outside the point groups), one thread each, plain stores: the slow path.
__global__ __launch_bounds__(256) void ba_linearize_list_kernel(Dev d, const uint32_t* __restrict__ list, uint32_t n) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint64_t o = list[idx];
  const uint32_t ip = d.opose[o], ii = d.ointr[o], ix = d.opt[o];
  double pin[8], pp[6], px[3], obs[2], r[2], Ji[16], Jc[12], Jp[6];
#pragma unroll
  for (int k = 0; k < 8; ++k) pin[k] = d.intr[(size_t)ii * 8 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) pp[k] = d.poses[(size_t)ip * 6 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) px[k] = d.pts[(size_t)ix * 3 + k];
  obs[0] = d.oxy[2 * o]; obs[1] = d.oxy[2 * o + 1];
  eval_observation<true>(d.model[ii], pin, pp, px, obs, r, Ji, Jc, Jp, observation_is_off(d, o));
  const double sc = correct_observation(d, o, r);
  double* __restrict__ ja = d.JA + (size_t)o * kJA;
  double* __restrict__ jb = d.JB + (size_t)o * kJB;
  double* __restrict__ jc = d.JC + (size_t)o * kJC;
  ja[0] = r[0]; ja[1] = r[1]; jb[0] = r[0]; jb[1] = r[1];
#pragma unroll
  for (int k = 0; k < 6; ++k) ja[2 + k] = Jp[k] * sc;
#pragma unroll
  for (int k = 0; k < 12; ++k) jb[2 + k] = Jc[k] * sc;
  jb[14] = 0.0; jb[15] = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) jc[k] = Ji[k] * sc;
}

template <bool kJac>
__global__ __launch_bounds__(256) void ba_linearize_kernel(Dev d, const double* __restrict__ poses,
                                                           const double* __restrict__ intr, const double* __restrict__ pts,
                                                           double* __restrict__ part /* gridDim x 2 */) {
  __shared__ double sh[4];
  __shared__ double2 rec_lds[kJac ? 4 * 64 * 9 : 1];   // 36 KiB: the records of the four waves, one array at a time
  const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0, sq = 0;
  double2 va[4], vb[8], vc[8];
  if (o < d.n_obs) {
    const uint32_t ip = d.opose[o], ii = d.ointr[o], ix = d.opt[o];
    double pin[8], pp[6], px[3], obs[2], r[2], Ji[16], Jc[12], Jp[6];
#pragma unroll
    for (int k = 0; k < 8; ++k) pin[k] = intr[(size_t)ii * 8 + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) pp[k] = poses[(size_t)ip * 6 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) px[k] = pts[(size_t)ix * 3 + k];
    obs[0] = d.oxy[2 * o]; obs[1] = d.oxy[2 * o + 1];
    const bool off = observation_is_off(d, o);   // (mvgx_ba_update_subset: zero residual, zero rows, no cost; evaluated with a regular projection)
    eval_observation<kJac>(d.model[ii], pin, pp, px, obs, r, Ji, Jc, Jp, off);
    // WeightedCostFunction (camera_functor.hpp:35-90): weight 0 selects the unweighted functor
    double w = 1.0;
    if (d.oweight) { const double ww = d.oweight[o]; if (ww != 0.0) w = ww; }
    const bool ctrl = d.octrl && d.octrl[o];   // control point: no loss function, not in the RMSE
    if (off) w = 0.0;
    r[0] *= w; r[1] *= w;
    const double s = r[0] * r[0] + r[1] * r[1];
    double rho[3];
    huber_rho_on(!ctrl && d.huber_a > 0.0, d.huber_a, s, rho);
    cost = 0.5 * rho[0];
    sq = ctrl ? 0.0 : s;
    if (kJac) {
      const double sr = corrector_scale(rho);
      const double sc = sr * w;
      const double r0 = r[0] * sr, r1 = r[1] * sr;
      va[0] = make_double2(r0, r1);
#pragma unroll
      for (int k = 0; k < 3; ++k) va[1 + k] = make_double2(Jp[2 * k] * sc, Jp[2 * k + 1] * sc);
      vb[0] = make_double2(r0, r1);
#pragma unroll
      for (int k = 0; k < 6; ++k) vb[1 + k] = make_double2(Jc[2 * k] * sc, Jc[2 * k + 1] * sc);
      vb[7] = make_double2(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < 8; ++k) vc[k] = make_double2(Ji[2 * k] * sc, Ji[2 * k + 1] * sc);
    }
  }
  if (kJac) {   // uniform: the three record arrays leave coalesced (see store_records_coalesced)
    const int wave = threadIdx.x >> 6;
    const uint64_t o0 = (uint64_t)blockIdx.x * blockDim.x + (uint64_t)wave * 64;   // first observation of the wave
    const int n_valid = o0 >= d.n_obs ? 0 : (int)min<uint64_t>(64, d.n_obs - o0);
    const bool active = o < d.n_obs;
    double2* lw = rec_lds + wave * 64 * 9;
    store_records_coalesced<4>(lw, va, active, reinterpret_cast<double2*>(d.JA + (size_t)o0 * kJA), n_valid);
    store_records_coalesced<8>(lw, vb, active, reinterpret_cast<double2*>(d.JB + (size_t)o0 * kJB), n_valid);
    store_records_coalesced<8>(lw, vc, active, reinterpret_cast<double2*>(d.JC + (size_t)o0 * kJC), n_valid);
  }
  const double c = block_sum(cost, sh);
  const double q = block_sum(sq, sh);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = c; part[2 * blockIdx.x + 1] = q; }
}

// |x - project(X)| per observation at the given parameters, unweighted, no loss: what RemoveOutliers_PixelResidualError
// (sfm/sfm_data_filters.cpp:40-73) thresholds. out is indexed by the caller's observation order.
__global__ __launch_bounds__(256) void ba_residual_norm_kernel(Dev d, const uint32_t* __restrict__ orig_index, double* __restrict__ out) {
  const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= d.n_obs) return;
  const uint32_t ip = d.opose[o], ii = d.ointr[o], ix = d.opt[o];
  double pin[8], pp[6], px[3], obs[2], r[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) pin[k] = d.intr[(size_t)ii * 8 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) pp[k] = d.poses[(size_t)ip * 6 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) px[k] = d.pts[(size_t)ix * 3 + k];
  obs[0] = d.oxy[2 * o]; obs[1] = d.oxy[2 * o + 1];
  eval_observation<false>(d.model[ii], pin, pp, px, obs, r, nullptr, nullptr, nullptr);
  out[orig_index ? orig_index[o] : (uint32_t)o] = sqrt(r[0] * r[0] + r[1] * r[1]);
}

// world ray of every observation (3 doubles each, point-sorted order) at the current parameters
__global__ __launch_bounds__(256) void ba_obs_ray_kernel(Dev d, double* __restrict__ rays) {
  const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= d.n_obs) return;
  const uint32_t ip = d.opose[o], ii = d.ointr[o];
  double pin[8], pp[6], obs[2], ray[3];
#pragma unroll
  for (int k = 0; k < 8; ++k) pin[k] = d.intr[(size_t)ii * 8 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) pp[k] = d.poses[(size_t)ip * 6 + k];
  obs[0] = d.oxy[2 * o]; obs[1] = d.oxy[2 * o + 1];
  observation_ray(d.model[ii], pin, pp, obs, ray);
  rays[3 * o] = ray[0]; rays[3 * o + 1] = ray[1]; rays[3 * o + 2] = ray[2];
}

// per track: the largest angle (degrees) between the rays of two of its observations — what RemoveOutliers_AngleError
// (sfm/sfm_data_filters.cpp:77-121) compares with dMinAcceptedAngle; 0 for tracks with fewer than two observations
__global__ __launch_bounds__(256) void ba_track_angle_kernel(Dev d, const double* __restrict__ rays, double* __restrict__ out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.n_pts) return;
  const uint32_t lo = d.pt_start[p], hi = d.pt_start[p + 1];
  double best = 0.0;
  for (uint32_t a = lo; a < hi; ++a) {
    if (d.odisabled && d.odisabled[a]) continue;
    const double ra[3] = {rays[3 * (size_t)a], rays[3 * (size_t)a + 1], rays[3 * (size_t)a + 2]};
    for (uint32_t b = a + 1; b < hi; ++b) {
      if (d.odisabled && d.odisabled[b]) continue;
      const double rb[3] = {rays[3 * (size_t)b], rays[3 * (size_t)b + 1], rays[3 * (size_t)b + 2]};
      const double ang = ray_angle_deg(ra, rb);
      best = ang > best ? ang : best;   // std::max(angle, max_angle): a NaN angle never replaces the maximum
    }
  }
  out[p] = best;
}

// pose-centre priors (one workgroup): cost added onto scalars[kSCost]; with kJac the loss-corrected residual and
// Jacobian rows are kept for the Gram / gradient / model-cost kernels
template <bool kJac>
__global__ __launch_bounds__(256) void ba_prior_kernel(Dev d, const double* __restrict__ poses, int add_cost) {
  __shared__ double sh[4];
  double cost = 0;
  for (uint32_t q = threadIdx.x; q < d.n_priors; q += blockDim.x) {
    double r[3], Jc[18];
    eval_pose_center_prior<kJac>(poses + (size_t)d.prior_pose[q] * 6, d.prior_center + 3 * (size_t)q, d.prior_weight + 3 * (size_t)q, r, Jc);
    const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    double rho[3];
    huber_rho_on(true, d.prior_huber_a, s, rho);
    cost += 0.5 * rho[0];
    if (kJac) {
      const double sr = corrector_scale(rho);
      double* out = d.Jprior + (size_t)q * kPriorJ;
      for (int k = 0; k < 3; ++k) out[k] = r[k] * sr;
      for (int k = 0; k < 18; ++k) out[3 + k] = Jc[k] * sr;
    }
  }
  const double c = block_sum(cost, sh);
  if (threadIdx.x == 0 && add_cost) d.scalars[kSCost] += c;
}

// ------------------------------------------------------------------------------------------------------
// column norms (unscaled) and gradient J^T r
// ------------------------------------------------------------------------------------------------------
// Column norms and gradient of the point blocks on the record-based path: cn_p = sum_o diag(E_o^T E_o), g_p = sum_o E_o^T r_o over
// the point's observations, one thread per point (grouped points get theirs from ba_point_group_kernel).
__global__ __launch_bounds__(256) void ba_point_norms_kernel(Dev d) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.n_pts || (d.pt_grouped && d.pt_grouped[p])) return;
  double cn[3] = {0, 0, 0}, g[3] = {0, 0, 0};
  for (uint32_t o = d.pt_start[p]; o < d.pt_start[p + 1]; ++o) {
    double a[8];
    load_rec<8>(d.JA + (size_t)o * kJA, a);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      cn[c] += a[2 + c] * a[2 + c] + a[5 + c] * a[5 + c];
      g[c] += a[2 + c] * a[0] + a[5 + c] * a[1];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { d.cn_pt[(size_t)p * 3 + c] = cn[c]; d.g_pt[(size_t)p * 3 + c] = g[c]; }
}

// Lists that are permutations of uploaded ones are made on the device (mvgx_ba_create): out_pt[i] = opt[idx[i]] (optional),
// out_xy[i] = the image point of observation idx[i]
__global__ __launch_bounds__(256) void ba_gather_lists_kernel(const uint32_t* __restrict__ idx, uint32_t n, const uint32_t* __restrict__ opt,
                                                              const double* __restrict__ oxy, uint32_t* __restrict__ out_pt, double2* __restrict__ out_xy) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t o = idx[i];
  if (out_pt) out_pt[i] = opt[o];
  out_xy[i] = *reinterpret_cast<const double2*>(oxy + 2 * (size_t)o);
}

__device__ __forceinline__ constexpr int tri6(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }   // r <= c
__device__ __forceinline__ constexpr int tri8(int r, int c) { return r * 8 - (r * (r - 1)) / 2 + (c - r); }

// Gram blocks of the camera columns, by (pose, intrinsic) pair: the two rows of an observation's camera Jacobian are two rows of
// F = [Fc (6) | Fi (8) | r | 0], and F^T F of a chunk holds Fc^T Fc, Fc^T Fi, Fi^T Fi, Fc^T r and Fi^T r - UNscaled. One workgroup
// per chunk of the pair's observations (camera order: pi_xy / pi_pt are copies of the observations in that order, 20 contiguous
// bytes each; the points are gathered). Nothing is read back from a stored Jacobian: every thread evaluates its observation,
// drops the two rows into LDS, and each wave runs v_mfma_f64_16x16x4_f64 over its own 64 observations with lane (li, lk) feeding
// element li of row k0 + lk as BOTH operands; the four accumulators are summed in wave order.
// a value that is the same in every lane, moved to scalar registers (the compiler cannot prove the uniformity of a gathered load)
__device__ __forceinline__ double uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
constexpr int kGramRow = 34;   // doubles per staged observation (2 x 16 + pad: 16-byte aligned, conflict-free 16-byte stores)
template <bool kPinholeFamily>
__global__ __launch_bounds__(256, kPinholeFamily ? 4 : 2) void ba_cam_gram_kernel(Dev d) {
  // a wave stages HALF of its 64 observations at a time (lanes 0..31, then 32..63): 8.7 KB of LDS per wave instead of 17.4 KB lets
  // three workgroups share a CU (the polynomial-model variant needs 124 registers; with whole waves staged, LDS held it at two)
  __shared__ __attribute__((aligned(16))) double F[128 * kGramRow];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const uint32_t ch = blockIdx.x;
  if (d.gate && !*d.gate) return;   // (launched ahead of a step the device itself did not accept: nothing has changed)
  if (ch == 0 && tid == 0) { *d.fail = 0; d.scalars[kSGmax] = 0.0; }   // every Jacobian evaluation leaves the fail word of the following step clear (one memset launch less per iteration) and the gradient maximum ready for ba_gram_finish_kernel's atomic max
  const uint32_t lo = d.pichunk_lo[ch], hi = d.pichunk_hi[ch];
  const uint32_t ip = d.pichunk_pose[ch], ii = d.pichunk_intr[ch];
  // the chunk's camera is the same for every thread: its 14 parameters live in scalar registers (28 vector registers less - with
  // them in vector registers the polynomial-model variant spilled 4 - 14 registers at its 128-register budget: 20 - 60 bytes of
  // scratch traffic per observation)
  double pin[8], pp[6];
#pragma unroll
  for (int k = 0; k < 8; ++k) pin[k] = uniform_f64(d.intr[(size_t)ii * 8 + k]);
#pragma unroll
  for (int k = 0; k < 6; ++k) pp[k] = uniform_f64(d.poses[(size_t)ip * 6 + k]);
  const int model = __builtin_amdgcn_readfirstlane(d.model[ii]);
  double trig[kPoseTrig];
  pose_trig(pp, trig);   // (the pose is the chunk's: once per thread)
#pragma unroll
  for (int k = 0; k < kPoseTrig; ++k) trig[k] = uniform_f64(trig[k]);
  d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
  for (uint32_t base = lo; base < hi; base += 256) {
    const uint32_t e = base + tid;
    double2 rows[16];   // row 0: Fc (6) | Fi (8) | r | 0, then row 1
    if (e < hi) {
      const uint32_t ix = d.pi_pt[e];
      const double2 xy = d.pi_xy[e];
      double px[3], obs[2] = {xy.x, xy.y}, r[2], Ji[16], Jc[12], Jp[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) px[k] = d.pts[(size_t)ix * 3 + k];
      const uint64_t o = (d.oweight || d.octrl || d.odisabled) ? d.pi_obs[e] : 0;
      eval_observation_t<true, kPinholeFamily>(model, pin, pp, trig, px, obs, r, Ji, Jc, Jp, observation_is_off(d, o));
      const double sc = correct_observation(d, o, r);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double v[16];
#pragma unroll
        for (int c = 0; c < 6; ++c) v[c] = Jc[6 * h + c] * sc;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[6 + c] = Ji[8 * h + c] * sc;
        v[14] = r[h]; v[15] = 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c) rows[8 * h + c] = make_double2(v[2 * c], v[2 * c + 1]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) rows[c] = make_double2(0.0, 0.0);
    }
    const uint32_t w0 = base + 64u * wave;
    const int nw = w0 >= hi ? 0 : (int)min(64u, hi - w0);   // observations of this wave in this round
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((lane >> 5) == half) {
        double2* __restrict__ row = reinterpret_cast<double2*>(F + (32 * wave + (lane & 31)) * kGramRow);
#pragma unroll
        for (int c = 0; c < 16; ++c) row[c] = rows[c];
      }
      __syncthreads();
      const int nh = min(32, max(0, nw - 32 * half));   // observations of this half
      const double* __restrict__ src = F + (size_t)(32 * wave + (lk >> 1)) * kGramRow + 16 * (lk & 1) + li;
      for (int ks = 0; 2 * ks < nh; ++ks) {   // k-step ks: rows of observations 2 ks (lk = 0, 1) and 2 ks + 1 (lk = 2, 3)
        const double v = src[(size_t)(2 * ks) * kGramRow];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
      }
      __syncthreads();
    }
  }
  // D[i = lk + 4 reg][j = li] of the four waves, summed in wave order
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) F[(wave * 4 + reg) * 64 + lane] = acc[reg];
  __syncthreads();
  if (wave != 0) return;
  double* __restrict__ outp = d.pichunk_part + (size_t)ch * kPiGram;
  double* __restrict__ outi = d.pichunk_ipart + (size_t)ch * kIntrGram;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int i = lk + 4 * reg, j = li;
    const double val = ((F[(0 * 4 + reg) * 64 + lane] + F[(1 * 4 + reg) * 64 + lane]) + F[(2 * 4 + reg) * 64 + lane]) + F[(3 * 4 + reg) * 64 + lane];
    if (i < 6) {
      if (j < 6) { if (i <= j) outp[tri6(i, j)] = val; }
      else if (j < 14) outp[kPoseGram + i * 8 + (j - 6)] = val;
      else if (j == 14) outp[21 + i] = val;
    } else if (i < 14) {
      if (j >= 6 && j < 14) { if (i <= j) outi[tri8(i - 6, j - 6)] = val; }
      else if (j == 14) outi[36 + (i - 6)] = val;
    }
  }
}
// The three finish steps of the Gram blocks in one launch (they were three, the pose step waiting for the pair step): workgroups of
// 1 024 threads - eight (pose, intrinsic) pairs per workgroup, then 32 poses per workgroup (a pose sums its pairs' chunk partials
// itself, pair by pair: the same sums in the same order as via pi_gram), then the slices of the intrinsics. fold_diag (one rank, not
// iteration zero, every point grouped): the camera part of ba_lm_diag_kernel is done here as well - LM diagonal from the column
// norms, max |gradient| into scalars[kSGmax] (cleared by the Gram kernel; one atomic max per workgroup on the bits of a non-negative
// double: order-independent) - and that kernel and its reduction are not launched.
constexpr int kIntrSlices = 8;   // workgroups per intrinsic in ba_gram_finish_kernel: each sums a contiguous slice of the intrinsic's chunk list
__device__ __forceinline__ void finish_column(const Dev& d, int col, double cn, double dmin, double dmax) {
  const double s = d.scale_cam[col];
  d.diag_cam[col] = fmin(fmax(cn * s * s, dmin), dmax);
}
__global__ __launch_bounds__(1024) void ba_gram_finish_kernel(Dev d, uint32_t wg_pi, uint32_t wg_pose, int fold_diag, double dmin, double dmax) {
  __shared__ double sh[16][kIntrGram];
  __shared__ unsigned s_last;
  if (d.gate && !*d.gate) return;   // (as ba_cam_gram_kernel)
  const uint32_t wg = blockIdx.x;
  double gm = 0.0;   // fold_diag: max |gradient| over the active camera columns this thread finishes
  if (wg < wg_pi) {   // ---- (pose, intrinsic) pairs ----
    const uint32_t q = wg * 8 + (threadIdx.x >> 7);
    const int t = threadIdx.x & 127;
    if (q < (uint32_t)d.n_pi && t < kPiGram) {
      double v = 0;
      for (uint32_t ch = d.pi_chunk0[q]; ch < d.pi_chunk0[q + 1]; ++ch) v += d.pichunk_part[(size_t)ch * kPiGram + t];
      d.pi_gram[(size_t)q * kPiGram + t] = v;
    }
    return;
  }
  if (wg < wg_pi + wg_pose) {   // ---- poses ----
    const uint32_t i = (wg - wg_pi) * 32 + (threadIdx.x >> 5);
    const int t = threadIdx.x & 31;
    if (i < d.n_poses && t < kPoseGram) {
      double v = 0;
      for (uint32_t q = d.pose_pi_start[i]; q < d.pose_pi_start[i + 1]; ++q) {
        double w = 0;
        for (uint32_t ch = d.pi_chunk0[q]; ch < d.pi_chunk0[q + 1]; ++ch) w += d.pichunk_part[(size_t)ch * kPiGram + t];
        v += w;
      }
      if (d.n_priors) {
        int r = 0, c = 0;   // (r, c) of upper-triangle index t < 21
        if (t < 21) { int k = t; while (k >= 6 - r) { k -= 6 - r; ++r; } c = r + k; }
        for (uint32_t e = d.pose_prior_start[i]; e < d.pose_prior_start[i + 1]; ++e) {
          const double* jp = d.Jprior + (size_t)d.pose_prior_idx[e] * kPriorJ;
          if (t < 21) { for (int k = 0; k < 3; ++k) v += jp[3 + k * 6 + r] * jp[3 + k * 6 + c]; }
          else { for (int k = 0; k < 3; ++k) v += jp[3 + k * 6 + (t - 21)] * jp[k]; }
        }
      }
      d.pose_gram[(size_t)i * kPoseGram + t] = v;
      if (t >= 21) {
        d.g_cam[6 * i + (t - 21)] = v;
        if (fold_diag && d.cam_active[6 * i + (t - 21)]) gm = fabs(v);
      }
#pragma unroll
      for (int c = 0; c < 6; ++c)
        if (t == tri6(c, c)) { d.cn_cam[6 * i + c] = v; if (fold_diag) finish_column(d, 6 * (int)i + c, v, dmin, dmax); }
    }
    if (fold_diag) {
      const double m = block_max(gm, &sh[0][0]);
      if (threadIdx.x == 0 && m > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(d.scalars + kSGmax), (unsigned long long)__double_as_longlong(m));
    }
    return;
  }
  // ---- intrinsics: kIntrSlices workgroups each, the last arrival combines the slices ----
  const uint32_t wi = wg - wg_pi - wg_pose;
  const uint32_t k = wi / kIntrSlices, slice = wi % kIntrSlices;
  const int g = threadIdx.x >> 6, t = threadIdx.x & 63;
  if (t < kIntrGram) {
    double v = 0;
    {
      const uint32_t q00 = d.intr_pichunk_start[k], n_all = d.intr_pichunk_start[k + 1] - q00;
      const uint32_t per = (n_all + kIntrSlices - 1) / kIntrSlices;
      const uint32_t q0 = q00 + min(n_all, slice * per), q1 = q00 + min(n_all, (slice + 1) * per);
      for (uint32_t q = q0 + g; q < q1; q += 128) {
        uint32_t id[8];
        double x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) id[j] = d.intr_pichunk[min(q + 16 * j, q1 - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = d.pichunk_ipart[(size_t)id[j] * kIntrGram + t];
#pragma unroll
        for (int j = 0; j < 8; ++j) v += (q + 16 * j < q1) ? x[j] : 0.0;
      }
    }
    sh[g][t] = v;
  }
  __syncthreads();
  if (g == 0 && t < kIntrGram) {
    double v = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += sh[q][t];
    d.intr_slice_part[((size_t)k * kIntrSlices + slice) * kIntrGram + t] = v;
    __threadfence();   // the slice's sums are visible device-wide before the workgroup counts as arrived
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(d.intr_arrivals + k, 1u);
    s_last = prev + 1 == (unsigned)kIntrSlices;
    if (s_last) d.intr_arrivals[k] = 0;   // ready for the next evaluation
  }
  __syncthreads();
  if (!s_last || g != 0 || t >= kIntrGram) return;
  __threadfence();
  double v = 0;
#pragma unroll
  for (int q = 0; q < kIntrSlices; ++q) v += __builtin_nontemporal_load(d.intr_slice_part + ((size_t)k * kIntrSlices + q) * kIntrGram + t);
  d.igram[(size_t)k * kIntrGram + t] = v;
  const int col0 = 6 * (int)d.n_poses + 8 * (int)k;
  if (t >= 36) {
    d.g_cam[col0 + (t - 36)] = v;
    // (one wave finishes an intrinsic: its eight gradient entries go to the maximum one by one)
    if (fold_diag && d.cam_active[col0 + (t - 36)] && v != 0.0)
      atomicMax(reinterpret_cast<unsigned long long*>(d.scalars + kSGmax), (unsigned long long)__double_as_longlong(fabs(v)));
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (t == tri8(c, c)) { d.cn_cam[col0 + c] = v; if (fold_diag) finish_column(d, col0 + c, v, dmin, dmax); }
}

// jacobian_scaling_ = 1 / (1 + sqrt(|col|^2)) at iteration 0 (trust_region_minimizer.cc:239-254); 0 for inactive columns
__global__ void ba_make_scaling_kernel(Dev d, int jacobi) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)d.N) d.scale_cam[i] = d.cam_active[i] ? (jacobi ? 1.0 / (1.0 + sqrt(d.cn_cam[i])) : 1.0) : 0.0;
  if (i < (size_t)d.n_pts * 3) d.scale_pt[i] = d.pt_free[i / 3] ? (jacobi ? 1.0 / (1.0 + sqrt(d.cn_pt[i])) : 1.0) : 0.0;
}

// LM diagonal = clamp(diag(Js^T Js), min, max) (levenberg_marquardt_strategy.cc:75-87) + max |gradient| partials
// skip_grouped: the points of the point groups get their diagonal (and their share of max |gradient|) inside ba_point_group_kernel
__global__ __launch_bounds__(256) void ba_lm_diag_kernel(Dev d, double dmin, double dmax, double* __restrict__ part, int skip_grouped) {
  __shared__ double sh[4];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double gm = 0;
  if (i < (size_t)d.N) {
    const double s = d.scale_cam[i];
    d.diag_cam[i] = fmin(fmax(d.cn_cam[i] * s * s, dmin), dmax);
    if (d.cam_active[i]) gm = fabs(d.g_cam[i]);
  }
  if (i < (size_t)d.n_pts * 3 && !(skip_grouped && d.pt_grouped && d.pt_grouped[i / 3])) {
    const double s = d.scale_pt[i];
    d.diag_pt[i] = fmin(fmax(d.cn_pt[i] * s * s, dmin), dmax);
    if (s != 0.0) gm = fmax(gm, fabs(d.g_pt[i]));
  }
  const double t = block_max(gm, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ------------------------------------------------------------------------------------------------------
// per-point elimination
// ------------------------------------------------------------------------------------------------------
// per point: V = Es^T Es + D^2 = L L^T, L^-1, h = L^-1 Es^T r
__global__ __launch_bounds__(256) void ba_point_solve_kernel(Dev d, double inv_radius) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.n_pts || (d.pt_grouped && d.pt_grouped[p])) return;
  const uint32_t o0 = d.pt_start[p], o1 = d.pt_start[p + 1];
  const double sp[3] = {d.scale_pt[(size_t)p * 3], d.scale_pt[(size_t)p * 3 + 1], d.scale_pt[(size_t)p * 3 + 2]};
  double V[6] = {d.diag_pt[(size_t)p * 3] * inv_radius, 0, 0, d.diag_pt[(size_t)p * 3 + 1] * inv_radius, 0,
                 d.diag_pt[(size_t)p * 3 + 2] * inv_radius};
  double g[3] = {0, 0, 0};
  for (uint32_t o = o0; o < o1; ++o) {
    double a[8];
    load_rec<8>(d.JA + (size_t)o * kJA, a);
    const double r0 = a[0], r1 = a[1];
    double e0[3], e1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { e0[c] = a[2 + c] * sp[c]; e1[c] = a[5 + c] * sp[c]; }
    V[0] += e0[0] * e0[0] + e1[0] * e1[0]; V[1] += e0[0] * e0[1] + e1[0] * e1[1]; V[2] += e0[0] * e0[2] + e1[0] * e1[2];
    V[3] += e0[1] * e0[1] + e1[1] * e1[1]; V[4] += e0[1] * e0[2] + e1[1] * e1[2]; V[5] += e0[2] * e0[2] + e1[2] * e1[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] += e0[c] * r0 + e1[c] * r1;
  }
  double li[6] = {0, 0, 0, 0, 0, 0};
  const bool eliminate = sp[0] != 0.0;  // scale 0 <=> point constant / unused: no e-block (Z = 0, the rows stay in G)
  if (eliminate && o1 > o0) {
    if (!chol_inv3(V, li)) { atomicExch(d.fail, 1); for (int c = 0; c < 6; ++c) li[c] = 0.0; }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) d.Linv3[(size_t)p * 6 + c] = li[c];
  d.hp[(size_t)p * 3 + 0] = li[0] * g[0];
  d.hp[(size_t)p * 3 + 1] = li[1] * g[0] + li[2] * g[1];
  d.hp[(size_t)p * 3 + 2] = li[3] * g[0] + li[4] * g[1] + li[5] * g[2];
}

// per observation: Z = L_p^-1 Es^T Fc_s (3 x 6); a = {r, E} record, b = {r, Fc} record, sp / sc = point / pose column scales
__device__ __forceinline__ void obs_z_math(const double (&a)[8], const double (&b)[16], const double (&sp)[3], const double (&li)[6],
                                           const double (&sc)[6], double (&Z)[18]) {
  double e0[3], e1[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { e0[c] = a[2 + c] * sp[c]; e1[c] = a[5 + c] * sp[c]; }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double f0 = b[2 + c] * sc[c], f1 = b[8 + c] * sc[c];
    const double y0 = e0[0] * f0 + e1[0] * f1, y1 = e0[1] * f0 + e1[1] * f1, y2 = e0[2] * f0 + e1[2] * f1;
    Z[c] = li[0] * y0;
    Z[6 + c] = li[1] * y0 + li[2] * y1;
    Z[12 + c] = li[3] * y0 + li[4] * y1 + li[5] * y2;
  }
}
// list == nullptr: every observation; else the n listed ones (observations of points outside the point groups: the groups'
// own observations get their Z from ba_schur_group_kernel while it stages them)
__global__ __launch_bounds__(256) void ba_obs_z_kernel(Dev d, const uint32_t* __restrict__ list, uint64_t n) {
  const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint64_t o = list ? list[idx] : idx;
  const uint32_t p = d.opt[o], ip = d.opose[o];
  double a[8], b[16], sp[3], li[6], sc[6], Z[18];
  load_rec<8>(d.JA + (size_t)o * kJA, a);
  load_rec<16>(d.JB + (size_t)o * kJB, b);
#pragma unroll
  for (int c = 0; c < 3; ++c) sp[c] = d.scale_pt[(size_t)p * 3 + c];
#pragma unroll
  for (int c = 0; c < 6; ++c) { li[c] = d.Linv3[(size_t)p * 6 + c]; sc[c] = d.scale_cam[6 * ip + c]; }
  obs_z_math(a, b, sp, li, sc, Z);
  double2* __restrict__ zo = reinterpret_cast<double2*>(d.Zpose + (size_t)o * 18);
#pragma unroll
  for (int k = 0; k < 9; ++k) zo[k] = make_double2(Z[2 * k], Z[2 * k + 1]);
}

// per (point, intrinsic) slot and intrinsic column: Z[:, c] = L_p^-1 sum over the slot's observations of Es^T Fi_s[:, c]
__global__ __launch_bounds__(256) void ba_slot_z_kernel(Dev d) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t s = idx >> 3;
  const int c = idx & 7;
  if (s >= (uint32_t)d.n_islots) return;
  const uint32_t p = d.slot_point[s], ik = d.slot_intr[s];
  if (d.pt_grouped && d.pt_grouped[p]) return;
  const double sp[3] = {d.scale_pt[(size_t)p * 3], d.scale_pt[(size_t)p * 3 + 1], d.scale_pt[(size_t)p * 3 + 2]};
  const double sc = d.scale_cam[6 * d.n_poses + 8 * ik + c];
  double y0 = 0, y1 = 0, y2 = 0;
  for (uint32_t o = d.pt_start[p]; o < d.pt_start[p + 1]; ++o) {
    if (d.ointr[o] != ik) continue;
    double a[8];
    load_rec<8>(d.JA + (size_t)o * kJA, a);   // the 8 lanes of a slot read the same record (broadcast)
    const double f0 = d.JC[(size_t)o * kJC + c] * sc, f1 = d.JC[(size_t)o * kJC + 8 + c] * sc;
    y0 += (a[2] * f0 + a[5] * f1) * sp[0];
    y1 += (a[3] * f0 + a[6] * f1) * sp[1];
    y2 += (a[4] * f0 + a[7] * f1) * sp[2];
  }
  const double* li = d.Linv3 + (size_t)p * 6;
  d.Zint[(size_t)s * 24 + c] = li[0] * y0;
  d.Zint[(size_t)s * 24 + 8 + c] = li[1] * y0 + li[2] * y1;
  d.Zint[(size_t)s * 24 + 16 + c] = li[3] * y0 + li[4] * y1 + li[5] * y2;
}

// One wave per chunk of products of one destination block: acc += Z_a^T Z_b (WA x WB), and for the (a, a) products of a
// diagonal block rhs += Z_a^T h_p. Lanes stride the chunk; the 64 per-lane partial blocks are summed through LDS in a
// fixed order.
template <int WA, int WB>
__global__ __launch_bounds__(64) void ba_schur_products_kernel(TripList L, const double* __restrict__ Za, const double* __restrict__ Zb,
                                                               const double* __restrict__ hp, const uint32_t* __restrict__ a_point) {
  constexpr int NV = WA * WB + WA;
  __shared__ double red[64][NV + 1];
  // Workgroup ids are dealt round-robin to the 8 XCDs: give each XCD one contiguous eighth of the (row, column)-sorted
  // chunk list, so that the products that share Z records of a camera row meet in one 4 MiB L2.
  const uint32_t per = (L.n_chunks + 7) / 8;
  const uint32_t ch = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (ch >= L.n_chunks) return;
  const int lane = threadIdx.x;
  const uint32_t lo = L.chunk_lo[ch], hi = L.chunk_hi[ch];
  const bool diag = L.chunk_diag[ch] != 0;
  double acc[WA * WB], rhs[WA];
#pragma unroll
  for (int k = 0; k < WA * WB; ++k) acc[k] = 0.0;
#pragma unroll
  for (int k = 0; k < WA; ++k) rhs[k] = 0.0;
  for (uint32_t t = lo + lane; t < hi; t += 64) {
    const uint2 ab = L.trips[t];
    double za[3 * WA], zb[3 * WB];
    const double2* __restrict__ pa = reinterpret_cast<const double2*>(Za + (size_t)ab.x * (3 * WA));
    const double2* __restrict__ pb = reinterpret_cast<const double2*>(Zb + (size_t)ab.y * (3 * WB));
#pragma unroll
    for (int k = 0; k < 3 * WA / 2; ++k) { const double2 v = pa[k]; za[2 * k] = v.x; za[2 * k + 1] = v.y; }
#pragma unroll
    for (int k = 0; k < 3 * WB / 2; ++k) { const double2 v = pb[k]; zb[2 * k] = v.x; zb[2 * k + 1] = v.y; }
#pragma unroll
    for (int r = 0; r < WA; ++r)
#pragma unroll
      for (int c = 0; c < WB; ++c)
        acc[r * WB + c] += za[r] * zb[c] + za[WA + r] * zb[WB + c] + za[2 * WA + r] * zb[2 * WB + c];
    if (diag && ab.x == ab.y) {
      const uint32_t p = a_point[ab.x];
      const double h0 = hp[(size_t)p * 3], h1 = hp[(size_t)p * 3 + 1], h2 = hp[(size_t)p * 3 + 2];
#pragma unroll
      for (int r = 0; r < WA; ++r) rhs[r] += za[r] * h0 + za[WA + r] * h1 + za[2 * WA + r] * h2;
    }
  }
#pragma unroll
  for (int k = 0; k < WA * WB; ++k) red[lane][k] = acc[k];
#pragma unroll
  for (int k = 0; k < WA; ++k) red[lane][WA * WB + k] = rhs[k];
  __syncthreads();
  for (int idx = lane; idx < NV; idx += 64) {
    double v = 0;
#pragma unroll 8
    for (int l = 0; l < 64; ++l) v += red[l][idx];
    L.part[(size_t)ch * NV + idx] = v;
  }
}

// ------------------------------------------------------------------------------------------------------
// the fused point-group pass (see GroupList)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int group_pair_pp(int x, int y) { return x * kGroupCams - x * (x - 1) / 2 + (y - x); }   // x <= y
__device__ __forceinline__ int group_pair_ii(int k, int l) { return k * kGroupIntr - k * (k - 1) / 2 + (l - k); }   // k <= l
// upper tile t of the kGroupColTiles x kGroupColTiles tile grid, row by row: (0,0) (0,1) .. (0,4) (1,1) ..
__device__ __forceinline__ void group_tile(int t, int& ti, int& tj, int n = kGroupColTiles) {
  ti = 0;
  while (t >= n) { t -= n; --n; ++ti; }
  tj = ti + t;
}
// The compact form of the staged matrix (round 6): a supergroup that uses ONE local intrinsic whose model has at most kCompactIntr
// parameters (pinhole: its intrinsic columns beyond are zero by construction) keeps h_p in column 6 kGroupCams + kCompactIntr instead
// of kGroupHCol - 64 columns, four column tiles, ten tiles of Z^T Z instead of fifteen. Same sums, same bits; the zero elements the
// five dropped tiles used to deliver are written as zeros when the partial blocks go out.
constexpr int kCompactIntr = 3;
// The strip form (round 6): one local intrinsic of up to kStripIntr parameters. Columns: 60 pose columns | h_p (60) | the intrinsic's eight
// columns (61 ..; those beyond 67 are zero and are not read). Z^T Z = the ten 16 x 16 tiles of columns 0 .. 63 + the products of columns
// 64 .. 67 with columns 0 .. 67: five v_mfma_f64_4x4x4_4b_f64 per k-step (18 cycles each: 90) where the fifth column tile cost five
// 16 x 16 x 4 instructions (320). Lane (li, lk) of strip instruction s holds the product of columns 64 + lk and 16 s + li.
constexpr int kStripIntr = 7;
constexpr int kStripHCol = 6 * kNarrowCams, kStripICol0 = kStripHCol + 1, kStripCol0 = 64, kStripInstr = 5;
static_assert(kStripICol0 + kStripIntr <= kStripCol0 + 4, "the strip's four columns hold the intrinsic's last parameters");
constexpr int kCompactHCol = 6 * kNarrowCams + kCompactIntr;
constexpr int kCompactColTiles = (kCompactHCol + 16) / 16;
constexpr int kCompactTiles = kCompactColTiles * (kCompactColTiles + 1) / 2;
static_assert(kCompactHCol == 63 && kCompactColTiles == 4, "the compact form is sized for four column tiles");
// A finished 16 x 16 tile of the supergroup's Z^T Z goes to LDS as partial blocks of the three product families:
//   out + 0                      [pair (x <= y)][42]  6 x 6 block (+ the rhs of pose x for x == y, from the h column)
//   out + 55 * 42                [x * kGroupIntr + k][54]  6 x 8 block
//   out + 55 * 42 + 20 * 54      [pair (k <= l)][72]  8 x 8 block (+ the rhs of intrinsic k for k == l)
// Only elements with global column I <= J are defined (the upper triangle of a diagonal block is the whole block).
// (I <= J: columns of the staged matrix in its usual forms - pose columns below 6 kNarrowCams, 16 intrinsic columns from icol0, h_p in
// column hcol: behind the intrinsic columns, or in front of them in the strip form)
__device__ __forceinline__ void group_store_elem(int I, int J, double val, double* __restrict__ out, int icol0, int hcol) {
  constexpr int kPoseCols = 6 * kNarrowCams;
  double* __restrict__ out_pi = out + kGroupPairsPP * kNVpp;
  double* __restrict__ out_ii = out_pi + kGroupPairsPI * kNVpi;
  if (I > J) return;
  const bool i_pose = I < kPoseCols, j_pose = J < kPoseCols;   // (h_p first: in the compact form its column lies inside the intrinsic's range)
  const bool i_intr = I != hcol && I >= icol0 && I < icol0 + 8 * kGroupIntr, j_intr = J != hcol && J >= icol0 && J < icol0 + 8 * kGroupIntr;
  if (i_pose) {
    const int x = I / 6, r = I - 6 * x;
    if (j_pose) { const int y = J / 6, c = J - 6 * y; out[group_pair_pp(x, y) * kNVpp + r * 6 + c] = val; }
    else if (J == hcol) out[group_pair_pp(x, x) * kNVpp + 36 + r] = val;
    else if (j_intr) { const int l = (J - icol0) >> 3, c = (J - icol0) & 7; out_pi[(x * kGroupIntr + l) * kNVpi + r * 8 + c] = val; }
  } else if (i_intr) {
    const int k = (I - icol0) >> 3, r = (I - icol0) & 7;
    if (j_intr) { const int l = (J - icol0) >> 3, c = (J - icol0) & 7; out_ii[group_pair_ii(k, l) * kNVii + r * 8 + c] = val; }
    else if (J == hcol) out_ii[group_pair_ii(k, k) * kNVii + 64 + r] = val;
  } else if (I == hcol && j_intr) {   // (the strip form: h_p in front of the intrinsic columns)
    const int l = (J - icol0) >> 3, c = (J - icol0) & 7;
    out_ii[group_pair_ii(l, l) * kNVii + 64 + c] = val;
  }
}
__device__ __forceinline__ void group_store_tile(const d4_t& acc, int ti, int tj, double* __restrict__ out, int li, int lk, int icol0, int hcol) {
  const int J = 16 * tj + li;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) group_store_elem(16 * ti + lk + 4 * reg, J, acc[reg], out, icol0, hcol);
}

// The wide form: a tile's four elements of a lane go straight to the partial blocks in memory - where (chunk: the destination rows of
// the supergroup's blocks, pp | pi | ii as in LDS; null: the element has no destination). A supergroup of several groups adds to what its
// group before left there: the same lane owns the same elements in every group, the old values are fetched before the tile's k loop.
__device__ __forceinline__ void wide_tile_dst(int ti, int tj, const uint32_t* __restrict__ chunk, double* __restrict__ part_pp, double* __restrict__ part_pi,
                                              double* __restrict__ part_ii, int li, int lk, int kPoseCols, int hcol, double* (&dst)[4]) {
  const int J = 16 * tj + li;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    dst[reg] = nullptr;
    const int I = 16 * ti + lk + 4 * reg;
    if (I >= hcol || J > hcol || I > J) continue;
    const bool i_pose = I < kPoseCols;
    const int x = i_pose ? I / 6 : (I - kPoseCols) / 8;
    const int r = i_pose ? I - 6 * x : (I - kPoseCols) - 8 * x;
    if (J == hcol) {
      if (i_pose) { const uint32_t ch = chunk[group_pair_pp(x, x)]; if (ch != kNoChunk) dst[reg] = part_pp + (size_t)ch * kNVpp + 36 + r; }
      else { const uint32_t ch = chunk[kGroupPairsPP + kGroupPairsPI + group_pair_ii(x, x)]; if (ch != kNoChunk) dst[reg] = part_ii + (size_t)ch * kNVii + 64 + r; }
    } else if (J < kPoseCols) {
      const int y = J / 6, c = J - 6 * y;
      const uint32_t ch = chunk[group_pair_pp(x, y)];
      if (ch != kNoChunk) dst[reg] = part_pp + (size_t)ch * kNVpp + r * 6 + c;
    } else {
      const int l = (J - kPoseCols) / 8, c = (J - kPoseCols) - 8 * l;
      if (i_pose) { const uint32_t ch = chunk[kGroupPairsPP + x * kGroupIntr + l]; if (ch != kNoChunk) dst[reg] = part_pi + (size_t)ch * kNVpi + r * 8 + c; }
      else { const uint32_t ch = chunk[kGroupPairsPP + kGroupPairsPI + group_pair_ii(x, l)]; if (ch != kNoChunk) dst[reg] = part_ii + (size_t)ch * kNVii + r * 8 + c; }
    }
  }
}

// 1 / sqrt(x): v_rsq_f64 + two Newton steps (full precision for finite x > 0)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  r = fma(fma(-hx * r, r, 0.5), r, r);
  r = fma(fma(-hx * r, r, 0.5), r, r);
  return r;
}
// chol_inv3 (ba_math.h) for the point threads of ba_point_group_kernel: the three pivots through 1 / sqrt - no square root, no
// division (their IEEE sequences are ~20 dependent instructions each, and the phase runs on one lane in eight while the rest of the
// workgroup waits at the barrier). Same quantities to rounding.
__device__ __forceinline__ bool chol_inv3_fast(const double v[6], double li[6]) {
  if (!(v[0] > 0.0)) return false;
  const double i00 = fast_rsqrt(v[0]);
  const double l10 = v[1] * i00, l20 = v[2] * i00;
  const double d1 = v[3] - l10 * l10;
  if (!(d1 > 0.0)) return false;
  const double i11 = fast_rsqrt(d1);
  const double l21 = (v[4] - l20 * l10) * i11;
  const double d2 = v[5] - l20 * l20 - l21 * l21;
  if (!(d2 > 0.0)) return false;
  const double i22 = fast_rsqrt(d2);
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  li[0] = i00; li[1] = i10; li[2] = i11; li[3] = i20; li[4] = i21; li[5] = i22;
  return true;
}

// One workgroup per supergroup, one thread per observation of the current group. MODE:
//   kGroupNorms    column norms and gradient of the points (iteration zero: the Jacobi scaling needs the norms first)
//   kGroupForward  per point V = Es^T Es + D^2 = L L^T, h = L^-1 Es^T r, Z = L^-1 Es^T Fs; S -= Z^T Z, rhs -= Z^T h as
//                  partial blocks of the three product lists; g_pt, diag_pt and max |g_pt| on the way
//   kGroupBacksub  the same up to Z, then step_pt = -L^-T (h - sum Z z)
// MVGX_BA_GROUP_DEBUG=1: shader clocks of thread 0 of every workgroup between the phases of ba_point_group_kernel, summed per phase
// (0 observation, 1 point sums, 2 point factors, 3 slots, 4 back-substitution, 5 matrix staging, 6 MFMA, 7 partial blocks out)
__device__ unsigned long long g_group_stamps[3][8];   // [mode][phase]
__device__ int g_group_debug;
__device__ int g_group_compact = 1;   // MVGX_BA_GROUP_COMPACT=0: every supergroup stages the full 80 columns (A/B runs)
// (summed per workgroup in LDS and added to the global counters once, at the end of the workgroup: an atomic per stamp on eight shared
// addresses slowed the stamped kernel by 60 % - round 6 - and the wait ended up in whichever phase came next)
#define MVGX_GSTAMP(i) do { if (stamping) { const long long t_now = __builtin_amdgcn_s_memtime(); s_stamps[i] += (unsigned long long)(t_now - t_prev); t_prev = __builtin_amdgcn_s_memtime(); } } while (0)
#define MVGX_GSTAMP_FLUSH() do { if (stamping) { for (int i_ = 0; i_ < 8; ++i_) if (s_stamps[i_]) atomicAdd(&g_group_stamps[MODE][i_], s_stamps[i_]); } } while (0)
// WIDE: the launch covers the supergroups of the wide form (G.sg_order[sg_base ..]; a launch of its own - round 6: with the form a
// run-time flag the usual form paid for it, 543 -> 603 us at C5 - so the form is a compile-time constant of each of the two launches).
template <int MODE, bool kPinholeFamily = false, bool WIDE = false>
__global__ __launch_bounds__(kGroupThreads, (MODE == 1 /* kGroupForward */ || (MODE == 2 && !kPinholeFamily)) ? 2 : 3) void ba_point_group_kernel(Dev d, GroupList G, double inv_radius, double dmin, double dmax,
                                                                       double* __restrict__ part_pp, double* __restrict__ part_pi,
                                                                       double* __restrict__ part_ii, double* __restrict__ cand_part, uint32_t sg_base, int zero_system) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // (round 6) issue priority: a forward workgroup's waves go FIRST on their SIMDs outside the Z^T Z phase and last inside it. The matrix
  // instruction holds the SIMD's vector issue for 64 cycles (section 4.3 of DESIGN.md): a SIMD-mate in its evaluation / sum / factor phases -
  // dependent chains - otherwise gets one instruction in between and crawls; with priority it issues whenever it can and the matrix stream
  // takes what is left. C3 0.498 against 0.508 ms per iteration, C5 1.328 against 1.335 (calls r6_60 / r6_61, same box, three alternations;
  // the reverse assignment: half of it; the same in the Gram kernel: nothing). Same results: only the order of issue changes.
  if (MODE == kGroupForward) __builtin_amdgcn_s_setprio(3);
  double* const M = lds;                          // per-observation terms [value][thread] -> the staged matrix [row][kGroupCS] -> partial blocks
  double* const sums = lds + group_m_doubles<MODE>();   // [point][20]
  double* const ptab = sums + kGroupSums;         // [point][12]: L^-1 (6) | h (3)
  double* const ctab = ptab + kGroupPtab;         // [local pose][kGroupCamRow]: parameters | rotation terms | column scales
  double* const itab = ctab + kGroupCams * kGroupCamRow;   // [local intrinsic][kGroupIntrRow]: parameters | column scales
  double* const ztab = itab + kGroupIntr * kGroupIntrRow;  // back-substitution: the reduced solution at the group's columns (the staged matrix's column order)
  double* const ccand = ztab + 6 * kGroupCams + 8 * kGroupIntr;   // back-substitution + candidate: [local pose][kGroupCandRow] | [local intrinsic][8] of x + delta
  uint32_t* const ek = reinterpret_cast<uint32_t*>(ccand + kGroupCand);   // [thread]: entry word of the observation
  // kGroupBacksub with cand_part: the pass also forms the candidate x + delta of its points and the candidate's cost over its
  // observations (the cameras of x + delta are in d.cposes / d.cintr already), and what ba_step_scalars_kernel sums over the points -
  // five partial sums per supergroup; the observations are then not visited a fourth time in the iteration (ba_linearize_kernel<false>).
  const bool with_cand = MODE == kGroupBacksub && cand_part != nullptr;
  double c_cost = 0.0, c_sq = 0.0;
  double* const pacc = ccand + kGroupCams * kGroupCandRow + kGroupIntr * 8;   // [3][point thread]: |delta|^2, |x|^2, model cost terms (in LDS: registers are what this mode is short of)
  if (with_cand && threadIdx.x < 3 * kGroupPts) pacc[threadIdx.x] = 0.0;
  uint32_t* const pe = ek + kGroupThreads;        // [point + 1]: first entry (group-relative)
  uint32_t* const pks = pe + kGroupPts + 1;       // [point]: first entry of the point's observations with local intrinsic 1
  int* const imodel = reinterpret_cast<int*>(pks + kGroupPts + 1);   // [local intrinsic]: camera model
  int* const ncam_used = imodel + kGroupIntr;                         // forward: local poses of the supergroup that carry observations
  uint32_t* const chunk_ids = reinterpret_cast<uint32_t*>(ncam_used + 2);   // forward: destination rows of the supergroup's partial blocks (pp | pi | ii), fetched at the start
  int* const sg_flags = ncam_used + 1;
  int* const npose_cols = reinterpret_cast<int*>(chunk_ids + kGroupPairsPP + kGroupPairsPI + kGroupPairsII);   // forward: 6 x (the highest local pose in use + 1)                                // forward: bit 0 - only local intrinsic 0 is in use, bit 1 - the compact form (see kCompactIntr)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const uint32_t sg = G.sg_order[sg_base + blockIdx.x];
  const uint32_t g0 = G.sg_start[sg], g1 = G.sg_start[sg + 1];
  const uint32_t* __restrict__ cams = G.cams + (size_t)sg * kGroupCams;
  const uint32_t* __restrict__ intrs = G.intrs + (size_t)sg * kGroupIntr;
  constexpr bool wide = WIDE;   // (the host sorts the supergroups by form: a supergroup is wide when its camera set has more than kNarrowCams poses)
  constexpr int NT = kGroupThreads;
  constexpr int NSUM = MODE == kGroupBacksub ? 18 : 15;   // per-observation terms summed per point (back-substitution: + Es^T (Fs z))
  constexpr int NS = kGroupThreads + 1;   // stride of the per-observation terms [value][thread]: odd, so that the readers of one observation's values hit distinct banks
  d4_t acc[kGroupTilesPerWave];
  int tti[kGroupTilesPerWave], ttj[kGroupTilesPerWave];
#pragma unroll
  for (int j = 0; j < kGroupTilesPerWave; ++j) {
    acc[j] = d4_t{0.0, 0.0, 0.0, 0.0};
    group_tile(min(wave + j * kGroupWaves, kGroupTiles - 1), tti[j], ttj[j]);
  }
  // The cameras of a supergroup are the same for all of its groups: their parameters, the rotation terms that depend on the pose
  // alone (sqrt, sin, cos, division: once per pose here instead of once per observation), the Jacobi scales of their columns and -
  // for the back-substitution - the reduced solution at those columns go to LDS once; the observations then read LDS rows instead
  // of gathering 30 doubles each through the vector memory path.
  if (tid < kGroupCams) {
    const uint32_t ip = cams[tid];
    double pp[6], trig[kPoseTrig];
#pragma unroll
    for (int k = 0; k < 6; ++k) pp[k] = d.poses[(size_t)ip * 6 + k];
    pose_trig(pp, trig);
    double* __restrict__ row = ctab + tid * kGroupCamRow;
#pragma unroll
    for (int k = 0; k < 6; ++k) { row[k] = pp[k]; row[6 + kPoseTrig + k] = d.scale_cam[6 * (size_t)ip + k]; }
#pragma unroll
    for (int k = 0; k < kPoseTrig; ++k) row[6 + k] = trig[k];
    if (with_cand) {
#pragma unroll
      for (int k = 0; k < 6; ++k) pp[k] = d.cposes[(size_t)ip * 6 + k];
      pose_trig(pp, trig);
      double* __restrict__ crow2 = ccand + tid * kGroupCandRow;
#pragma unroll
      for (int k = 0; k < 6; ++k) crow2[k] = pp[k];
#pragma unroll
      for (int k = 0; k < kPoseTrig; ++k) crow2[6 + k] = trig[k];
    }
  } else if (tid >= 64 && tid < 64 + kGroupIntr * kGroupIntrRow) {
    const int k = (tid - 64) / kGroupIntrRow, j = (tid - 64) - k * kGroupIntrRow;
    const uint32_t ii = intrs[k];
    itab[k * kGroupIntrRow + j] = j < 8 ? d.intr[(size_t)ii * 8 + j] : d.scale_cam[6 * (size_t)d.n_poses + 8 * (size_t)ii + (j - 8)];
    if (j == 0) imodel[k] = d.model[ii];
  } else if (MODE == kGroupBacksub && tid >= 128 && tid < 128 + 6 * kGroupCams + 8 * kGroupIntr) {
    const int j = tid - 128;
    ztab[j] = j < 6 * kGroupCams ? d.zsol[6 * (size_t)cams[j / 6] + j % 6]
                                 : d.zsol[6 * (size_t)d.n_poses + 8 * (size_t)intrs[(j - 6 * kGroupCams) >> 3] + ((j - 6 * kGroupCams) & 7)];
  } else if (with_cand && tid >= 240 && tid < 240 + 8 * kGroupIntr) {
    const int j = tid - 240;
    ccand[kGroupCams * kGroupCandRow + j] = d.cintr[(size_t)intrs[j >> 3] * 8 + (j & 7)];
  }
  if (MODE == kGroupForward && tid >= 192) {   // (wave 3: the waves 0..2 stage the tables above)
    const int x = tid - 192;
    const bool used = x < kGroupCams && G.chunk_pp[(size_t)sg * kGroupPairsPP + group_pair_pp(x, x)] != kNoChunk;
    const unsigned long long used_mask = __ballot(used);
    const int n_used = (int)__popcll(used_mask);
    if (x == 0) {
      *ncam_used = n_used;
      *npose_cols = 0;
      for (int b = 0; b < kGroupCams; ++b) if ((used_mask >> b) & 1ull) *npose_cols = 6 * (b + 1);   // pose columns up to the last local pose in use
      const bool single = G.chunk_ii[(size_t)sg * kGroupPairsII + group_pair_ii(1, 1)] == kNoChunk;   // no point of the supergroup sees local intrinsic 1
      const int pc = intr_param_count(d.model[intrs[0]]);
      // (bit 2: the strip form - one local intrinsic of 4 .. kStripIntr parameters, radial K3 and its kin: h_p in column 60, the intrinsic
      // columns behind it - 68 columns: the ten tiles of four column tiles, and columns 64 .. 67 on v_mfma_f64_4x4x4_4b_f64)
      *sg_flags = (single ? 1 : 0) | (!wide && single && pc >= 0 && pc <= kCompactIntr && g_group_compact ? 2 : 0) |
                  (!wide && single && pc > kCompactIntr && pc <= kStripIntr && g_group_compact ? 4 : 0);
    }
  }
  if (MODE == kGroupForward && tid < kGroupPairsPP + kGroupPairsPI + kGroupPairsII) {
    // (the last phase used to fetch a block's destination element by element: fourteen dependent trips to memory per workgroup, 8 us)
    static_assert(kGroupPairsPP + kGroupPairsPI + kGroupPairsII <= kGroupThreads, "one destination row per thread");
    const int j = tid;
    chunk_ids[j] = j < kGroupPairsPP ? G.chunk_pp[(size_t)sg * kGroupPairsPP + j]
                 : j < kGroupPairsPP + kGroupPairsPI ? G.chunk_pi[(size_t)sg * kGroupPairsPI + (j - kGroupPairsPP)]
                 : G.chunk_ii[(size_t)sg * kGroupPairsII + (j - kGroupPairsPP - kGroupPairsPI)];
  }
  if (MODE == kGroupForward && zero_system) {
    // The reduced system is zeroed here, a slice per workgroup (of the first forward launch) - the assemble pass that follows this kernel
    // writes only the blocks that exist, the tiles of the fill must start at zero - instead of by a memset launch in front of this kernel.
    double2* const z2 = reinterpret_cast<double2*>(d.sp.enabled ? d.sp.A : d.S);
    const size_t n2 = (d.sp.enabled ? (size_t)d.sp.n_slots * 4096 : (size_t)d.N * d.LD) / 2;   // (N (N + 1) is even)
    const size_t per = (n2 + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (size_t i = lo + tid; i < hi; i += kGroupThreads) z2[i] = make_double2(0.0, 0.0);
  }
  double gmax = 0.0;
  __shared__ unsigned long long s_stamps[8];
  const bool stamping = g_group_debug && tid == 0;
  if (stamping) { for (int i_ = 0; i_ < 8; ++i_) s_stamps[i_] = 0; }
  long long t_prev = stamping ? __builtin_amdgcn_s_memtime() : 0;
  // The first words of a group (its ranges, this thread's entry, the point of a point thread) are fetched one group ahead: they
  // head the chains of dependent loads (entry -> ids -> parameters), which then start from registers.
  // forward: the form of the staged matrix (the flags were set by wave 3 above)
  int hcol = kGroupHCol, n_tiles = kGroupTiles;
  bool single_intr = false, strip = false;
  if (MODE == kGroupForward) {
    __syncthreads();
    const int fl = *sg_flags;
    single_intr = (fl & 1) != 0;
    if (fl & 6) {   // compact or strip: the ten tiles of four column tiles
      hcol = (fl & 2) ? kCompactHCol : kStripHCol; n_tiles = kCompactTiles;
#pragma unroll
      for (int j = 0; j < kGroupTilesPerWave; ++j) group_tile(min(wave + j * kGroupWaves, kCompactTiles - 1), tti[j], ttj[j], kCompactColTiles);
    }
    strip = !wide && (fl & 4) != 0;
    // (the wide form keeps its intrinsic columns and h_p right behind the pose columns IN USE: 11 poses are 83 columns - 6 column
    // tiles, 21 tiles of Z^T Z - where the full width has 36)
    if (wide) hcol = *npose_cols + 8 * kGroupIntr;
  }
  const bool compact = !wide && hcol == kCompactHCol;
  const int cs = wide ? kWideCols : kGroupCS;                          // doubles between the rows of the staged matrix
  const int icol0 = wide ? hcol - 8 * kGroupIntr : strip ? kStripICol0 : 6 * kNarrowCams;    // its first intrinsic column
  // the strip form's instructions: waves 0 / 1 carry three tiles, 2 / 3 two - the five strip instructions go 4 | - | 0, 1 | 2, 3
  const int strip_s0 = wave == 0 ? 4 : wave == 2 ? 0 : wave == 3 ? 2 : -1, strip_n = !strip ? 0 : wave == 0 ? 1 : wave >= 2 ? 2 : 0;
  double sacc[2] = {0.0, 0.0};
  const int wide_col_tiles = (hcol + 16) / 16;
  const int ncap = wide ? kGroupCams : kNarrowCams;                    // entries a point can have
  uint32_t nx_e0 = G.obs_start[g0], nx_ne = G.obs_start[g0 + 1] - nx_e0, nx_p0 = G.pt_start[g0], nx_np = G.pt_start[g0 + 1] - nx_p0;
  uint32_t nx_qxk = (uint32_t)tid < nx_ne ? G.eq[nx_e0 + tid] : 0u;
  double2 nx_xy = (uint32_t)tid < nx_ne ? G.exy[nx_e0 + tid] : make_double2(0.0, 0.0);
  uint32_t nx_pt = (uint32_t)tid < nx_np ? G.pts[nx_p0 + tid] : 0u;
  uint32_t nx_pe = (uint32_t)tid <= nx_np ? G.pt_estart[nx_p0 + tid] - nx_e0 : 0u;
  uint32_t nx_pk = MODE == kGroupForward && (uint32_t)tid < nx_np ? G.pt_ksplit[nx_p0 + tid] - nx_e0 : 0u;
  // ... and so is the point of this thread's observation, in two steps that each start a phase after the load they depend on has
  // landed (entry word -> point id -> coordinates and scales): an observation thread used to open every group with those two
  // dependent trips to memory while the rest of its wave waited
  uint32_t nx_ix = (uint32_t)tid < nx_ne ? G.pts[nx_p0 + (nx_qxk & 255u)] : 0u;
  double nx_px[3] = {0.0, 0.0, 0.0}, nx_spo[3] = {0.0, 0.0, 0.0};
  if ((uint32_t)tid < nx_ne) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { nx_px[k] = d.pts[(size_t)nx_ix * 3 + k]; if (MODE == kGroupForward) nx_spo[k] = d.scale_pt[(size_t)nx_ix * 3 + k]; }
  }
  for (uint32_t g = g0; g < g1; ++g) {
    const uint32_t e0 = nx_e0, ne = nx_ne, p0 = nx_p0, np = nx_np;
    const uint32_t qxk = nx_qxk, my_pt = nx_pt, my_pe = nx_pe, my_pk = nx_pk;
    const double2 xy = nx_xy;
    const double cur_px[3] = {nx_px[0], nx_px[1], nx_px[2]}, cur_spo[3] = {nx_spo[0], nx_spo[1], nx_spo[2]};
    const uint32_t cur_ix = nx_ix;   // (the scales are fetched ahead in the forward mode only: the back-substitution sits at its register budget of three waves per SIMD)
    if (g + 1 < g1) {
      nx_e0 = G.obs_start[g + 1]; nx_ne = G.obs_start[g + 2] - nx_e0; nx_p0 = G.pt_start[g + 1]; nx_np = G.pt_start[g + 2] - nx_p0;
      nx_qxk = (uint32_t)tid < nx_ne ? G.eq[nx_e0 + tid] : 0u;
      nx_xy = (uint32_t)tid < nx_ne ? G.exy[nx_e0 + tid] : make_double2(0.0, 0.0);
      nx_pt = (uint32_t)tid < nx_np ? G.pts[nx_p0 + tid] : 0u;
      nx_pe = (uint32_t)tid <= nx_np ? G.pt_estart[nx_p0 + tid] - nx_e0 : 0u;
      nx_pk = MODE == kGroupForward && (uint32_t)tid < nx_np ? G.pt_ksplit[nx_p0 + tid] - nx_e0 : 0u;
    }
    // a point thread's scales: fetched here, used after the point sums (a dependent load there would stall the whole workgroup)
    double my_sp[3] = {0.0, 0.0, 0.0};
    if (MODE != kGroupNorms && (uint32_t)tid < np) {
#pragma unroll
      for (int c = 0; c < 3; ++c) my_sp[c] = d.scale_pt[(size_t)my_pt * 3 + c];
    }
    __syncthreads();   // the previous group's readers of M / sums / ptab / ek / pe are done
    // ---- 1. the observation of this thread: residual, closed-form Jacobian, loss correction, column scales. What stays in
    // registers until the point factors exist: the scaled rows Es (2 x 3), Fc_s (2 x 6), Fi_s (2 x 8); the per-observation terms
    // of the point sums go straight to LDS ----
    const bool has = (uint32_t)tid < ne;
    int q = 0, x = 0;
    double es0[3], es1[3], fc0[6], fc1[6], fi0[8], fi1[8];
    if (has) {
      ek[tid] = qxk;
      q = (int)(qxk & 255u); x = (int)((qxk >> 8) & 15u);
      const int kk = (int)((qxk >> 12) & 15u);
      double pin[8], pp[6], trig[kPoseTrig], px[3] = {cur_px[0], cur_px[1], cur_px[2]}, obs[2] = {xy.x, xy.y}, r[2], Ji[16], Jc[12], Jp[6];
      if (with_cand) {   // the point thread takes x from here (every observation of the point writes the same three values)
#pragma unroll
        for (int k = 0; k < 3; ++k) ptab[q * 12 + 9 + k] = px[k];
      }
      const double* __restrict__ crow = ctab + x * kGroupCamRow;
      const double* __restrict__ irow = itab + kk * kGroupIntrRow;
#pragma unroll
      for (int k = 0; k < 6; ++k) pp[k] = crow[k];
#pragma unroll
      for (int k = 0; k < kPoseTrig; ++k) trig[k] = crow[6 + k];
#pragma unroll
      for (int k = 0; k < 8; ++k) pin[k] = irow[k];
      const uint64_t o_ = (d.oweight || d.octrl || d.odisabled) ? G.eobs[e0 + tid] : 0;
      eval_observation_t<true, kPinholeFamily>(imodel[kk], pin, pp, trig, px, obs, r, Ji, Jc, Jp, observation_is_off(d, o_));
      const double sc = correct_observation(d, o_, r);
      // unscaled point terms: column norms and gradient (what ba_point_norms_kernel sums from the records)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double e0u = Jp[c] * sc, e1u = Jp[3 + c] * sc;
        M[(9 + c) * NS + tid] = e0u * e0u + e1u * e1u;
        M[(12 + c) * NS + tid] = e0u * r[0] + e1u * r[1];
        if (MODE != kGroupNorms) { const double sp = MODE == kGroupForward ? cur_spo[c] : d.scale_pt[(size_t)cur_ix * 3 + c]; es0[c] = e0u * sp; es1[c] = e1u * sp; }
      }
      if (MODE != kGroupNorms) {
        M[0 * NS + tid] = es0[0] * es0[0] + es1[0] * es1[0]; M[1 * NS + tid] = es0[0] * es0[1] + es1[0] * es1[1];
        M[2 * NS + tid] = es0[0] * es0[2] + es1[0] * es1[2]; M[3 * NS + tid] = es0[1] * es0[1] + es1[1] * es1[1];
        M[4 * NS + tid] = es0[1] * es0[2] + es1[1] * es1[2]; M[5 * NS + tid] = es0[2] * es0[2] + es1[2] * es1[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) M[(6 + c) * NS + tid] = es0[c] * r[0] + es1[c] * r[1];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double s_ = crow[6 + kPoseTrig + c] * sc;
          fc0[c] = Jc[c] * s_; fc1[c] = Jc[6 + c] * s_;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const double s_ = irow[8 + c] * sc;
          fi0[c] = Ji[c] * s_; fi1[c] = Ji[8 + c] * s_;
        }
        if (MODE == kGroupBacksub) {
          // sum_e Z_e z = L^-1 sum_e Es^T (Fs z): the two-vector w = Fc_s z_pose + Fi_s z_intr of this observation, then Es^T w
          double w0 = 0.0, w1 = 0.0;
#pragma unroll
          for (int c = 0; c < 6; ++c) { const double z = ztab[6 * x + c]; w0 += fc0[c] * z; w1 += fc1[c] * z; }
#pragma unroll
          for (int c = 0; c < 8; ++c) { const double z = ztab[6 * kGroupCams + 8 * kk + c]; w0 += fi0[c] * z; w1 += fi1[c] * z; }
#pragma unroll
          for (int c = 0; c < 3; ++c) M[(15 + c) * NS + tid] = es0[c] * w0 + es1[c] * w1;
        }
      }
    }
    if ((uint32_t)tid <= np) pe[tid] = my_pe;
    if (MODE == kGroupForward && (uint32_t)tid < np) pks[tid] = my_pk;
    if (g + 1 < g1) nx_ix = (uint32_t)tid < nx_ne ? G.pts[nx_p0 + (nx_qxk & 255u)] : 0u;
    __syncthreads();
    MVGX_GSTAMP(0);
    // ---- 2. per point: the 15 sums over its observations, in observation order ----
    for (int it = tid; it < (int)np * NSUM; it += NT) {
      const int pq = it / NSUM, v = it - pq * NSUM;
      if (MODE == kGroupNorms && v < 9) continue;
      const uint32_t elo = pe[pq], ehi = pe[pq + 1];
      const double* __restrict__ src = M + v * NS;
      double term[kGroupCams];
#pragma unroll
      for (int j = 0; j < kNarrowCams; ++j) term[j] = src[min(elo + (uint32_t)j, ehi - 1)];   // independent loads, all in flight
      if (wide) {   // (uniform)
#pragma unroll
        for (int j = kNarrowCams; j < kGroupCams; ++j) term[j] = src[min(elo + (uint32_t)j, ehi - 1)];
      }
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < kNarrowCams; ++j) sum += elo + (uint32_t)j < ehi ? term[j] : 0.0;
      if (wide) {
#pragma unroll
        for (int j = kNarrowCams; j < kGroupCams; ++j) sum += elo + (uint32_t)j < ehi ? term[j] : 0.0;
      }
      sums[pq * 20 + v] = sum;
    }
    __syncthreads();
    MVGX_GSTAMP(1);
    if (g + 1 < g1 && (uint32_t)tid < nx_ne) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { nx_px[k] = d.pts[(size_t)nx_ix * 3 + k]; if (MODE == kGroupForward) nx_spo[k] = d.scale_pt[(size_t)nx_ix * 3 + k]; }
    }
    // ---- 3. per point: norms / gradient out; LM diagonal, V = L L^T, L^-1, h ----
    if ((uint32_t)tid < np) {
      const uint32_t p = my_pt;
      const double* __restrict__ sm = sums + tid * 20;
      if (MODE == kGroupNorms) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { d.cn_pt[(size_t)p * 3 + c] = sm[9 + c]; d.g_pt[(size_t)p * 3 + c] = sm[12 + c]; }
      } else {
        double V[6] = {sm[0], sm[1], sm[2], sm[3], sm[4], sm[5]}, li6[6];
        double dg[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) dg[c] = fmin(fmax(sm[9 + c] * my_sp[c] * my_sp[c], dmin), dmax);
        V[0] += dg[0] * inv_radius; V[3] += dg[1] * inv_radius; V[5] += dg[2] * inv_radius;
        if (!chol_inv3_fast(V, li6)) { atomicExch(d.fail, 1); for (int c = 0; c < 6; ++c) li6[c] = 0.0; }
        double* __restrict__ pt = ptab + tid * 12;
#pragma unroll
        for (int c = 0; c < 6; ++c) pt[c] = li6[c];
        pt[6] = li6[0] * sm[6];
        pt[7] = li6[1] * sm[6] + li6[2] * sm[7];
        pt[8] = li6[3] * sm[6] + li6[4] * sm[7] + li6[5] * sm[8];
        if (MODE == kGroupBacksub) {   // step = -L^-T (h - L^-1 sum_e Es^T (Fs z))
          const double t0 = pt[6] - li6[0] * sm[15];
          const double t1 = pt[7] - (li6[1] * sm[15] + li6[2] * sm[16]);
          const double t2 = pt[8] - (li6[3] * sm[15] + li6[4] * sm[16] + li6[5] * sm[17]);
          const double st[3] = {-(li6[0] * t0 + li6[1] * t1 + li6[3] * t2), -(li6[2] * t1 + li6[4] * t2), -(li6[5] * t2)};
#pragma unroll
          for (int c = 0; c < 3; ++c) d.step_pt[(size_t)p * 3 + c] = st[c];
          if (with_cand) {   // the point of x + delta; its terms of |delta|^2, |x|^2 and of the model cost change (as ba_step_scalars_kernel forms them)
            double c_dsq = pacc[tid], c_xsq = pacc[kGroupPts + tid], c_vp = pacc[2 * kGroupPts + tid];
            const double my_px[3] = {pt[9], pt[10], pt[11]};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const double delta = st[c] * my_sp[c];
              const double cp = my_px[c] + delta;
              d.cpts[(size_t)p * 3 + c] = cp;
              pt[9 + c] = cp;
              c_dsq += delta * delta;
              if (my_sp[c] != 0.0) c_xsq += my_px[c] * my_px[c];
              c_vp += 0.5 * st[c] * (st[c] * dg[c] * inv_radius - sm[12 + c] * my_sp[c]);
            }
            pacc[tid] = c_dsq; pacc[kGroupPts + tid] = c_xsq; pacc[2 * kGroupPts + tid] = c_vp;
          }
        }
        if (MODE == kGroupForward) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            d.g_pt[(size_t)p * 3 + c] = sm[12 + c]; d.diag_pt[(size_t)p * 3 + c] = dg[c];
            gmax = fmax(gmax, fabs(sm[12 + c]));
          }
        }
      }
    }
    MVGX_GSTAMP(2);
    if (with_cand) {   // (uniform) the observation of this thread at x + delta: cost terms only
      __syncthreads();
      if (has) {
        const int kk = (int)((qxk >> 12) & 15u);
        double pin[8], pp[6], trig[kPoseTrig], px[3], obs[2] = {xy.x, xy.y}, r[2], Ji[16], Jc[12], Jp[6];
        const double* __restrict__ crow2 = ccand + x * kGroupCandRow;
        const double* __restrict__ irow2 = ccand + kGroupCams * kGroupCandRow + kk * 8;
#pragma unroll
        for (int k = 0; k < 3; ++k) px[k] = ptab[q * 12 + 9 + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) pp[k] = crow2[k];
#pragma unroll
        for (int k = 0; k < kPoseTrig; ++k) trig[k] = crow2[6 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) pin[k] = irow2[k];
        const uint64_t o = (d.oweight || d.octrl || d.odisabled) ? G.eobs[e0 + tid] : 0;
        const bool off = observation_is_off(d, o);
        eval_observation_t<false, kPinholeFamily>(imodel[kk], pin, pp, trig, px, obs, r, Ji, Jc, Jp, off);
        double w = 1.0;
        if (d.oweight) { const double ww = d.oweight[o]; if (ww != 0.0) w = ww; }
        if (off) w = 0.0;
        const bool ctrl = d.octrl && d.octrl[o];
        r[0] *= w; r[1] *= w;
        const double s2 = r[0] * r[0] + r[1] * r[1];
        double rho[3];
        huber_rho_on(!ctrl && d.huber_a > 0.0, d.huber_a, s2, rho);
        c_cost += 0.5 * rho[0];
        c_sq += ctrl ? 0.0 : s2;
      }
    }
    if (MODE != kGroupForward) continue;
    // ---- 4. intrinsic slots: Zint[q][k][:, c] = L_q^-1 sum over the point's observations with local intrinsic k of Es^T Fi_s[:, c] ----
    // (the sums of step 2 have been read: the terms of this step may replace them)
    if (has) {
      if (compact) {   // (uniform) the columns beyond kCompactIntr are zero and have no place in the compact matrix
#pragma unroll
        for (int c = 0; c < kCompactIntr; ++c) {
#pragma unroll
          for (int e = 0; e < 3; ++e) M[(e * 8 + c) * NS + tid] = es0[e] * fi0[c] + es1[e] * fi1[c];
        }
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
          for (int e = 0; e < 3; ++e) M[(e * 8 + c) * NS + tid] = es0[e] * fi0[c] + es1[e] * fi1[c];
        }
      }
    }
    __syncthreads();
    // one item per (point, column): the entries of the point are at most kGroupCams, ordered by local intrinsic (host) - the first
    // pks - pe of them belong to intrinsic 0; all 3 x kGroupCams terms are fetched before the first sum (the two ranges used to be two
    // rounds of a loop with one dependent trip to LDS per entry), each range summed in entry order as before
    static_assert(kGroupPts * 8 <= kGroupThreads, "one slot item per thread");
    double zs[kGroupIntr][3];
#pragma unroll
    for (int k = 0; k < kGroupIntr; ++k) zs[k][0] = zs[k][1] = zs[k][2] = 0.0;
    const int slot_sh = compact ? 2 : 3;   // items per point: 4 (3 in use) or 8
    const int slot_pq = tid >> slot_sh, slot_c = tid & ((1 << slot_sh) - 1);
    const bool slot_item = slot_pq < (int)np && (!compact || slot_c < kCompactIntr);
    {
      const int pq = slot_pq, c = slot_c;
      // the entries of a point in rounds of kSlotRound (all terms of a round fetched before its first sum): two rounds cover the usual
      // form's ten entries, four the wide form's sixteen
      constexpr int kSlotRound = 5;
      const int n_rounds = wide ? (kGroupCams + kSlotRound - 1) / kSlotRound : (kNarrowCams + kSlotRound - 1) / kSlotRound;
      if (slot_item && single_intr) {
        // (uniform choice) every entry of the point belongs to local intrinsic 0: one range, its terms added in entry order behind a
        // 0 / 1 factor (y + t * 1 = y + t, y + t * 0 = y: the bits of the masked sums below)
        const uint32_t elo = pe[pq], ehi = pe[pq + 1];
        double y0 = 0.0, y1 = 0.0, y2 = 0.0;
        for (int half = 0; half < n_rounds; ++half) {
          double t0[kSlotRound], t1[kSlotRound], t2[kSlotRound];
#pragma unroll
          for (int j = 0; j < kSlotRound; ++j) {
            const uint32_t e = min(elo + (uint32_t)(half * kSlotRound + j), ehi - 1);
            t0[j] = M[c * NS + e]; t1[j] = M[(8 + c) * NS + e]; t2[j] = M[(16 + c) * NS + e];
          }
#pragma unroll
          for (int j = 0; j < kSlotRound; ++j) {
            const double m = elo + (uint32_t)(half * kSlotRound + j) < ehi ? 1.0 : 0.0;
            y0 = fma(t0[j], m, y0); y1 = fma(t1[j], m, y1); y2 = fma(t2[j], m, y2);
          }
        }
        const double* __restrict__ pt = ptab + pq * 12;
        zs[0][0] = pt[0] * y0;
        zs[0][1] = pt[1] * y0 + pt[2] * y1;
        zs[0][2] = pt[3] * y0 + pt[4] * y1 + pt[5] * y2;
      } else if (slot_item) {
        const uint32_t elo = pe[pq], ehi = pe[pq + 1], esp = pks[pq];
        double y[kGroupIntr][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        for (int half = 0; half < n_rounds; ++half) {
          double t0[kSlotRound], t1[kSlotRound], t2[kSlotRound];
#pragma unroll
          for (int j = 0; j < kSlotRound; ++j) {
            const uint32_t e = min(elo + (uint32_t)(half * kSlotRound + j), ehi - 1);
            t0[j] = M[c * NS + e]; t1[j] = M[(8 + c) * NS + e]; t2[j] = M[(16 + c) * NS + e];
          }
#pragma unroll
          for (int j = 0; j < kSlotRound; ++j) {
            const uint32_t e = elo + (uint32_t)(half * kSlotRound + j);
            const bool in0 = e < esp, in1 = e >= esp && e < ehi;
            y[0][0] += in0 ? t0[j] : 0.0; y[0][1] += in0 ? t1[j] : 0.0; y[0][2] += in0 ? t2[j] : 0.0;
            y[1][0] += in1 ? t0[j] : 0.0; y[1][1] += in1 ? t1[j] : 0.0; y[1][2] += in1 ? t2[j] : 0.0;
          }
        }
        const double* __restrict__ pt = ptab + pq * 12;
#pragma unroll
        for (int k = 0; k < kGroupIntr; ++k) {
          zs[k][0] = pt[0] * y[k][0];
          zs[k][1] = pt[1] * y[k][0] + pt[2] * y[k][1];
          zs[k][2] = pt[3] * y[k][0] + pt[4] * y[k][1] + pt[5] * y[k][2];
        }
      }
    }
    const double* __restrict__ ptq = ptab + q * 12;   // L_q^-1 of this thread's point
    __syncthreads();   // the per-observation terms in M have been read
    MVGX_GSTAMP(3);
    // ---- 5. the staged matrix: rows 3 q .. 3 q + 2, columns 6 x .. (poses), 60 + 8 k .. (intrinsics), kGroupHCol (h) ----
    // A group whose every point is observed by every local pose of the supergroup in use (points grouped by identical pose sets: the
    // common case) writes every cell of the columns that count - pose columns by the observation threads, intrinsic columns and h by
    // the point items - so only the padding rows behind the last point need zeros; the columns of local poses without observations
    // keep whatever the region held: an element of Z^T Z depends on its own two columns only, and the blocks of those poses have
    // no destination (kNoChunk). Any other group clears the region first.
    const int rows = ((int)(3 * np + 3)) & ~3;
    const bool dense = ne == np * (uint32_t)*ncam_used;   // (a point sees a pose at most once: groups hold no repeated pose)
    if (!dense) {
      for (int i = tid; i < kGroupM / 2; i += NT) reinterpret_cast<double2*>(M)[i] = make_double2(0.0, 0.0);
      __syncthreads();
    } else {
      const int npad = rows - 3 * (int)np;   // 0..3 rows of zeros behind the last point
      for (int i = tid; i < cs * npad; i += NT) M[3 * (int)np * cs + i] = 0.0;
    }
    if (has) {
      // Z = L^-1 Es^T Fc_s as (L^-1 Es^T) Fc_s: the two columns a0, a1 of L^-1 Es^T (3 x 2) first - 12 operations - then two per element
      // (the element-by-element form L^-1 (Es^T Fc_s) took 12 per column: 72 against 48)
      const double a0[3] = {ptq[0] * es0[0], ptq[1] * es0[0] + ptq[2] * es0[1], ptq[3] * es0[0] + ptq[4] * es0[1] + ptq[5] * es0[2]};
      const double a1[3] = {ptq[0] * es1[0], ptq[1] * es1[0] + ptq[2] * es1[1], ptq[3] * es1[0] + ptq[4] * es1[1] + ptq[5] * es1[2]};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        // (16-byte aligned: 640 or 1 024 (3 q + r) + 48 x; the wide form's rows are 1 024 bytes apart - every row on the same banks - so
        // its odd rows keep their columns exchanged in blocks of 16, `swz`: an MFMA operand's two rows then meet different banks)
        const int swz = wide ? (((3 * q + r) & 1) << 4) : 0;
        double* __restrict__ dst = M + (3 * q + r) * cs;
#pragma unroll
        for (int c = 0; c < 6; c += 2)
          *reinterpret_cast<double2*>(dst + ((6 * x + c) ^ swz)) = make_double2(a0[r] * fc0[c] + a1[r] * fc1[c], a0[r] * fc0[c + 1] + a1[r] * fc1[c + 1]);
      }
    }
    if (slot_item) {
      const int pq = slot_pq, c = slot_c;
#pragma unroll
      for (int k = 0; k < kGroupIntr; ++k) {   // column 8 k + c behind the pose columns (compact: local intrinsic 0 only)
        if (k == 0 || !(compact || strip)) {
          const int col = icol0 + 8 * k + c, r0 = 3 * pq;
#pragma unroll
          for (int r = 0; r < 3; ++r) M[(r0 + r) * cs + (wide ? col ^ (((r0 + r) & 1) << 4) : col)] = zs[k][r];
        }
      }
    }
    if ((uint32_t)tid < np) {
#pragma unroll
      for (int r = 0; r < 3; ++r) M[(3 * tid + r) * cs + (wide ? hcol ^ (((3 * tid + r) & 1) << 4) : hcol)] = ptab[tid * 12 + 6 + r];
    }
    __syncthreads();
    MVGX_GSTAMP(5);
    if (MODE == kGroupForward) __builtin_amdgcn_s_setprio(0);   // (see the head of the kernel)
    // ---- 6. Z^T Z: the upper tiles dealt round-robin to the waves, accumulated over the groups of the supergroup ----
    if (wide) {   // (uniform) the wide form: 36 tiles, a wave's nine one after the other, each straight to the partial blocks
      for (int t = wave; t < wide_col_tiles * (wide_col_tiles + 1) / 2; t += kGroupWaves) {
        int ti, tj;
        group_tile(t, ti, tj, wide_col_tiles);
        const double* __restrict__ ca = M + lk * kWideCols + ((16 * ti + li) ^ ((lk & 1) << 4));   // (row k0 + lk: k0 is a multiple of four)
        const double* __restrict__ cb = M + lk * kWideCols + ((16 * tj + li) ^ ((lk & 1) << 4));
        double* dst[4];
        wide_tile_dst(ti, tj, chunk_ids, part_pp, part_pi, part_ii, li, lk, icol0, hcol, dst);
        double old[4] = {0.0, 0.0, 0.0, 0.0};
        if (g != g0) {   // (uniform) what the groups before left
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) if (dst[reg]) old[reg] = *dst[reg];
        }
        d4_t a = d4_t{0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < rows; k0 += 4) a = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[k0 * kWideCols], cb[k0 * kWideCols], a, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) if (dst[reg]) *dst[reg] = old[reg] + a[reg];
      }
    } else {
#pragma unroll
    for (int j = 0; j < kGroupTilesPerWave; ++j) {
      if (wave + j * kGroupWaves < n_tiles) {   // wave-uniform
        const double* __restrict__ ca = M + lk * kGroupCS + 16 * tti[j] + li;
        const double* __restrict__ cb = M + lk * kGroupCS + 16 * ttj[j] + li;
        d4_t a = acc[j];
        for (int k0 = 0; k0 < rows; k0 += 4) a = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[k0 * kGroupCS], cb[k0 * kGroupCS], a, 0, 0, 0);
        acc[j] = a;
      }
    }
    if (strip_n) {   // (wave-uniform) columns 64 .. 67 against the columns of this wave's strip instructions
      const double* __restrict__ sa = M + lk * kGroupCS + kStripCol0 + (li & 3);
      const double* __restrict__ sb = M + lk * kGroupCS + 16 * strip_s0 + li;
      double s0 = sacc[0], s1 = sacc[1];
      if (strip_n == 2) {
        for (int k0 = 0; k0 < rows; k0 += 4) {
          const double av = sa[k0 * kGroupCS];
          s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, sb[k0 * kGroupCS], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, sb[k0 * kGroupCS + 16], s1, 0, 0, 0);
        }
      } else {
        for (int k0 = 0; k0 < rows; k0 += 4) s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(sa[k0 * kGroupCS], sb[k0 * kGroupCS], s0, 0, 0, 0);
      }
      sacc[0] = s0; sacc[1] = s1;
    }
    }
    if (MODE == kGroupForward) __builtin_amdgcn_s_setprio(3);
    MVGX_GSTAMP(6);
  }
  if (with_cand) {   // (uniform) the supergroup's five sums, in wave order
    __syncthreads();
    const bool ptt = tid < kGroupPts;
    const double v0 = block_sum(c_cost, sums), v1 = block_sum(c_sq, sums), v2 = block_sum(ptt ? pacc[tid] : 0.0, sums),
                 v3 = block_sum(ptt ? pacc[kGroupPts + tid] : 0.0, sums), v4 = block_sum(ptt ? pacc[2 * kGroupPts + tid] : 0.0, sums);
    if (tid == 0) { double* o = cand_part + 5 * (size_t)sg; o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3; o[4] = v4; }
  }
  if (MODE != kGroupForward) { MVGX_GSTAMP_FLUSH(); return; }
  // ---- 7. partial blocks out: tiles -> LDS -> contiguous runs in the three partial-sum buffers; max |g_pt| of the supergroup ----
  __syncthreads();
  if (wide) {   // (uniform) its tiles are out already
    const double gm = block_max(gmax, sums);
    if (tid == 0) G.gmax_part[sg] = gm;
    MVGX_GSTAMP(7);
    MVGX_GSTAMP_FLUSH();
    return;
  }
  double* const out = M;
  if (compact || strip) {   // (uniform) what the dropped tiles / columns held: zero rows / columns of the intrinsic's blocks beyond its parameters
    for (int i = tid; i < kGroupPairsPI * kNVpi + kGroupPairsII * kNVii; i += NT) out[kGroupPairsPP * kNVpp + i] = 0.0;
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < kGroupTilesPerWave; ++j)
    if (wave + j * kGroupWaves < n_tiles) group_store_tile(acc[j], tti[j], ttj[j], out, li, lk, icol0, hcol);
  for (int q = 0; q < strip_n; ++q)   // lane (li, lk) of strip instruction s: the product of columns 16 s + li (row) and 64 + lk (column)
    group_store_elem(16 * (strip_s0 + q) + li, kStripCol0 + lk, sacc[q], out, icol0, hcol);
  {
    const double gm = block_max(gmax, sums);   // (block_max synchronises: the tiles above are in LDS afterwards)
    if (tid == 0) G.gmax_part[sg] = gm;
  }
  __syncthreads();
  // a wave per block, a lane per element (elements a block does not define - the rhs of an off-diagonal block, the lower triangle of a
  // diagonal one - carry whatever the region held: ba_schur_assemble never uses them)
  static_assert(kNVpp <= 64 && kNVpi <= 64 && kNVii <= 128, "a block is one or two rows of lanes");
  {   // (pose x pose: 136 pair numbers, of which a supergroup of the usual form has at most 55 - a wave walks the LIVE ones of its quarter)
    constexpr int kPer = (kGroupPairsPP + kGroupWaves - 1) / kGroupWaves;
    static_assert(kPer <= 64, "a quarter of the pair numbers per wave, one per lane");
    const int p0 = wave * kPer;
    const uint32_t my_ch = lane < kPer && p0 + lane < kGroupPairsPP ? chunk_ids[p0 + lane] : kNoChunk;
    unsigned long long live = __ballot(my_ch != kNoChunk);
    while (live) {
      const int b = __builtin_ctzll(live);
      live &= live - 1;
      const uint32_t ch = (uint32_t)__builtin_amdgcn_readlane((int)my_ch, b);
      if (lane < kNVpp) part_pp[(size_t)ch * kNVpp + lane] = out[(p0 + b) * kNVpp + lane];
    }
  }
  for (int pair = wave; pair < kGroupPairsPI; pair += kGroupWaves) {
    const uint32_t ch = chunk_ids[kGroupPairsPP + pair];
    if (ch != kNoChunk && lane < kNVpi) part_pi[(size_t)ch * kNVpi + lane] = out[kGroupPairsPP * kNVpp + pair * kNVpi + lane];
  }
  for (int it = wave; it < 2 * kGroupPairsII; it += kGroupWaves) {
    const int pair = it >> 1, e = 64 * (it & 1) + lane;
    const uint32_t ch = chunk_ids[kGroupPairsPP + kGroupPairsPI + pair];
    if (ch != kNoChunk && e < kNVii) part_ii[(size_t)ch * kNVii + e] = out[kGroupPairsPP * kNVpp + kGroupPairsPI * kNVpi + pair * kNVii + e];
  }
  MVGX_GSTAMP(7);
  MVGX_GSTAMP_FLUSH();
}

// KIND 0: pose x pose, 1: pose x intrinsic, 2: intrinsic x intrinsic. Sums the chunks of one destination block, adds the
// scaled Gram block that belongs there, writes the block (and, for diagonal blocks, the rhs entries) into S.
// (b: destination block; e: element handled by this thread; sgi: which of the SG groups of 128 threads that stride the partial
// blocks of the destination this thread belongs to; red: SG x 128 doubles of LDS when SG > 1 - every thread of the workgroup then
// runs this function for the same b)
template <int WA, int WB, int KIND, int SG>
__device__ __forceinline__ void schur_assemble_block(const Dev& d, const TripList& L, uint32_t b, int e, int sgi, double (*red)[128]) {
  constexpr int NV = WA * WB + WA;
  const uint32_t rcb = L.block_row[b], ccb = L.block_col[b];
  const bool diag = rcb == ccb;
  const bool live = e < NV && (e < WA * WB || diag);
  double sum = 0;
  if (live) {   // eight loads in flight, added in list order
    const uint32_t c1 = L.block_chunk0[b + 1];
    uint32_t ch = L.block_chunk0[b] + sgi;
    const double* __restrict__ q = L.part + e;
    for (; ch + 7 * SG < c1; ch += 8 * SG) {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = q[(size_t)(ch + j * SG) * NV];
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[j];
    }
    for (; ch < c1; ch += SG) sum += q[(size_t)ch * NV];
  }
  if (live && L.block_ext0) {   // partial blocks of the point groups: loads four at a time, summed in list order
    const uint32_t x1 = L.block_ext0[b + 1];
    uint32_t x = L.block_ext0[b] + sgi;
    const double* __restrict__ q = L.part + (size_t)L.n_chunks * NV + e;
    for (; x + 3 * SG < x1; x += 4 * SG) {
      const double v0 = q[(size_t)x * NV], v1 = q[(size_t)(x + SG) * NV], v2 = q[(size_t)(x + 2 * SG) * NV], v3 = q[(size_t)(x + 3 * SG) * NV];
      sum += v0; sum += v1; sum += v2; sum += v3;
    }
    for (; x < x1; x += SG) sum += q[(size_t)x * NV];
  }
  if (SG > 1) {   // (a shared intrinsic's block collects a partial block from every supergroup that sees it: thousands)
    red[sgi][e] = sum;
    __syncthreads();
    if (sgi != 0) return;
    sum = 0;
#pragma unroll
    for (int j = 0; j < SG; ++j) sum += red[j][e];
  }
  if (!live) return;
  const uint32_t np = d.n_poses;
  const int row0 = rcb < np ? 6 * (int)rcb : 6 * (int)np + 8 * (int)(rcb - np);
  const int col0 = ccb < np ? 6 * (int)ccb : 6 * (int)np + 8 * (int)(ccb - np);
  if (e < WA * WB) {
    const int r = e / WB, c = e - r * WB;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    double own = 0;
    if (KIND == 0) {
      if (diag) own = d.pose_gram[(size_t)rcb * kPoseGram + tri6(lo, hi)];
    } else if (KIND == 1) {
      const int q = L.block_own[b];
      if (q >= 0) own = d.pi_gram[(size_t)q * kPiGram + kPoseGram + r * 8 + c];
    } else {
      if (diag) own = d.igram[(size_t)(rcb - np) * kIntrGram + tri8(lo, hi)];
    }
    if (diag && r > c) return;   // the upper triangle of a diagonal block is the whole block (sparse mode: one home per element)
    double val = own * d.scale_cam[row0 + r] * d.scale_cam[col0 + c] - sum;
    // asm_inv_radius != 0: ba_finish_system_kernel's work here (one rank, every camera block has its diagonal destination) -
    // S_jj += diag_j / radius for free components, a unit diagonal for constant / unused ones
    if (d.asm_inv_radius != 0.0 && diag && r == c) val = d.cam_active[row0 + r] ? val + d.diag_cam[row0 + r] * d.asm_inv_radius : 1.0;
    *sys_elem(d, row0 + r, col0 + c) = val;
  } else {
    const int r = e - WA * WB;
    const double g = KIND == 0 ? d.pose_gram[(size_t)rcb * kPoseGram + 21 + r] : d.igram[(size_t)(rcb - np) * kIntrGram + 36 + r];
    double val = g * d.scale_cam[row0 + r] - sum;
    if (d.asm_inv_radius != 0.0 && !d.cam_active[row0 + r]) val = 0.0;   // (... and a zero right-hand side)
    *sys_elem(d, row0 + r, d.N) = val;
  }
}

// One launch for the three product families (they write disjoint destination blocks): workgroups of 1 024 threads - eight pose x
// pose or pose x intrinsic destinations per workgroup (one per group of 128 threads), one intrinsic x intrinsic destination per
// workgroup (eight groups stride its thousands of partial blocks). As three launches the families ran one after the other
// (C5 17.8 + 7.2 + 9.4 us, C3 6.3 + 5.9 + 11.5 us).
__global__ __launch_bounds__(1024) void ba_schur_assemble_all_kernel(Dev d, uint32_t wg_pp, uint32_t wg_pi, uint32_t wg_ii) {
  __shared__ double red[8][128];
  const int e = threadIdx.x & 127, grp = threadIdx.x >> 7;
  const uint32_t wg = blockIdx.x;
  if (wg == wg_pp + wg_pi + wg_ii) {   // one more workgroup: max |gradient| over the supergroups of the forward pass (was a launch of its own)
    double v = 0;
    for (uint32_t i = threadIdx.x; i < d.grp.n_sg; i += 1024) v = fmax(v, d.grp.gmax_part[i]);
    const double t = block_max(v, &red[0][0]);
    if (threadIdx.x == 0) d.scalars[kSGmaxGrp] = t;
    return;
  }
  // (the intrinsic x intrinsic destinations first: their workgroups walk thousands of partial blocks each and should not be the ones
  // the dispatcher starts last)
  if (wg < wg_ii) {
    schur_assemble_block<8, 8, 2, 8>(d, d.tii, wg, e, grp, red);
  } else if (wg < wg_ii + wg_pi) {
    const uint32_t b = (wg - wg_ii) * 8 + grp;
    if (b < d.tpi.n_blocks) schur_assemble_block<6, 8, 1, 1>(d, d.tpi, b, e, 0, red);
  } else {
    const uint32_t b = (wg - wg_ii - wg_pi) * 8 + grp;
    if (b < d.tpp.n_blocks) schur_assemble_block<6, 6, 0, 1>(d, d.tpp, b, e, 0, red);
  }
}

// ---- multi-rank exchange of the reduced system --------------------------------------------------------------------
// S is block sparse (a pair of cameras shares a block only if some point sees both). Only the blocks that are non-zero
// on at least one rank travel: every rank marks its own blocks in an n_cb x n_cb flag matrix (upper triangle), one
// all-reduce(max) of the flags at the start of the solve gives all ranks the same union and the same packed layout.
template <int KIND>
__global__ void ba_mark_blocks_kernel(Dev d, TripList L, double* __restrict__ flags, int n_cb) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < L.n_blocks) flags[(size_t)L.block_row[b] * n_cb + L.block_col[b]] = 1.0;
}
// one workgroup per union block: S block -> packed (dir 0) or packed -> S block (dir 1); the last workgroup moves the rhs
__global__ __launch_bounds__(64) void ba_pack_system_kernel(Dev d, int dir) {
  const uint32_t b = blockIdx.x;
  const int t = threadIdx.x;
  if (b == d.n_ublocks) {
    double* pk = d.packed + (d.n_packed - (uint64_t)d.N);
    for (int i = t; i < d.N; i += 64) {
      double* sp = sys_elem(d, i, d.N);
      if (dir == 0) pk[i] = *sp; else *sp = pk[i];
    }
    return;
  }
  const int h = d.ublk_h[b], w = d.ublk_w[b];
  if (t >= h * w) return;
  const int r = t / w, c = t - r * w;
  if (d.ublk_row[b] == d.ublk_col[b] && r > c) {   // lower triangle of a diagonal block: not part of the system
    if (dir == 0) d.packed[d.ublk_off[b] + t] = 0.0;
    return;
  }
  double* sp = sys_elem(d, (int)d.ublk_row[b] + r, (int)d.ublk_col[b] + c);
  double* pk = d.packed + d.ublk_off[b] + t;
  if (dir == 0) *pk = *sp; else *sp = *pk;
}

// After the (cross-rank) sum of the partial systems: S_jj += D_j^2 = diag_j / radius for free components, unit diagonal
// and zero rhs for constant / unused ones (their off-diagonals are exactly zero: Jacobi scale 0).
__global__ __launch_bounds__(256) void ba_finish_system_kernel(Dev d, double inv_radius) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= d.N) return;
  double* dd = sys_elem(d, row, row);
  if (d.cam_active[row]) {
    *dd += d.diag_cam[row] * inv_radius;
  } else {
    *dd = 1.0;
    *sys_elem(d, row, d.N) = 0.0;
  }
}
__global__ void ba_pack_fail_kernel(Dev d) { d.scalars[kSFail] = (double)*d.fail; }

// ------------------------------------------------------------------------------------------------------
// Blocked Cholesky of the column-major lower matrix A (n x n, ld = n + 1) whose extra row n carries the rhs: after the
// sweep, row n holds y = L^-1 rhs. A(i, j) = S[j * ld + i].
// ------------------------------------------------------------------------------------------------------
constexpr int kLS = 65;    // LDS row stride of the 64 x 64 factor / inverse
constexpr int kTS = 80;    // LDS row stride of a k-major 64-wide MFMA operand tile (rows k, k+1 land 32 banks apart)
constexpr int kDiagLds = (2 * 64 * kLS + 128 * 17 + 64 + 8) * (int)sizeof(double);

__device__ __forceinline__ double fast_rcp(double x) {   // v_rcp_f64 + two Newton steps (full precision for finite x > 0)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}


__device__ __forceinline__ double readlane_f64(double v, int src_lane) {   // src_lane wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// Pieces of the in-LDS inverse of a 64 x 64 Cholesky factor (chol_diag_inv_body). All are executed by ONE wave.
// LDS writes of a wave followed by reads of other lanes of the SAME wave: no s_barrier needed (a wave's LDS operations
// complete in order), only the compiler has to keep the order.
__device__ __forceinline__ void lds_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// acc + X Y for 16 x 16 blocks in LDS arrays of row stride kLS: X[i][k] = Xp[i * kLS + k], Y[k][j] = Yp[k * kLS + j];
// the result element (lk + 4 reg, li) is in acc[reg] of lane (li, lk)
__device__ __forceinline__ d4_t block_mma(const double* Xp, const double* Yp, d4_t acc, int li, int lk) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xp[li * kLS + 4 * ks + lk], Yp[(4 * ks + lk) * kLS + li], acc, 0, 0, 0);
  return acc;
}
// the same with Y in a wave-private scratch block of row stride 17
__device__ __forceinline__ d4_t block_mma_t(const double* Xp, const double (*Y)[17], d4_t acc, int li, int lk) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xp[li * kLS + 4 * ks + lk], Y[4 * ks + lk][li], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void block_store(double* d, int stride, d4_t v, double scale, int li, int lk) {
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) d[(lk + 4 * reg) * stride + li] = scale * v[reg];
}
// inverse of the lower-triangular 16 x 16 diagonal block at b0 by forward substitution, one column per lane (16 lanes);
// rd[r] = 1 / L[r][r]
__device__ __forceinline__ void diag_block_inverse(const double (*L)[kLS], double (*Li)[kLS], const double* rd, int b0, int lane) {
  if (lane < 16) {
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double v = (r == lane) ? 1.0 : 0.0;
#pragma unroll
      for (int q = 0; q < r; ++q) v -= L[b0 + r][b0 + q] * x[q];
      x[r] = v * rd[b0 + r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Li[b0 + r][b0 + lane] = x[r];
  }
}

// One step of the blocked inverse, executed by a whole wave with wave-uniform parameters (scalar control flow only):
//   acc = [nothing | the wave's scratch block S0 | scratch block 1 of wave `from`] + sum_{p < np} L[I][K0 + p] Linv[K0 + p][J]
//   out 0 / 1: acc is left in the wave's scratch block S0 / S1;   out 2: Linv[I][J] = -Linv[I][I] acc
struct InvTask { int I, J, K0, np, init /* 0 zero, 1 own S0, 2 S1 of wave 2 */, out; };
// [slot: panel 2, panel 3, after the loop round A, round B][wave][step]; I < 0: idle
__device__ const InvTask kInvSchedule[4][4][2] = {
  // while panel 2 is factored (final: L columns 0..31, Linv00 from the slot before)
  {{{-1, 0, 0, 0, 0, 0}, {-1, 0, 0, 0, 0, 0}}, {{-1, 0, 0, 0, 0, 0}, {-1, 0, 0, 0, 0, 0}},
   {{1, 0, 0, 1, 0, 0}, {-1, 0, 0, 0, 0, 0}},                                  // wave 2: S0 = L10 Linv00
   {{-1, 0, 0, 0, 0, 0}, {-1, 0, 0, 0, 0, 0}}},
  // while panel 3 is factored (final: L columns 0..47, Linv11)
  {{{-1, 0, 0, 0, 0, 0}, {-1, 0, 0, 0, 0, 0}}, {{-1, 0, 0, 0, 0, 0}, {-1, 0, 0, 0, 0, 0}},
   {{1, 0, 0, 0, 1, 2}, {-1, 0, 0, 0, 0, 0}},                                  // wave 2: Linv10 = -Linv11 S0
   {{2, 1, 1, 1, 0, 0}, {-1, 0, 0, 0, 0, 0}}},                                 // wave 3: S0 = L21 Linv11
  // round A (final: L, all diagonal blocks of Linv, Linv10)
  {{{-1, 0, 0, 0, 0, 0}, {-1, 0, 0, 0, 0, 0}},
   {{2, 0, 0, 2, 0, 2}, {-1, 0, 0, 0, 0, 0}},                                  // wave 1: Linv20 = -Linv22 (L20 Linv00 + L21 Linv10)
   {{3, 0, 0, 2, 0, 1}, {3, 1, 1, 1, 0, 0}},                                   // wave 2: S1 = L30 Linv00 + L31 Linv10, S0 = L31 Linv11
   {{2, 1, 0, 0, 1, 2}, {3, 2, 2, 1, 0, 0}}},                                  // wave 3: Linv21 = -Linv22 S0, then S0 = L32 Linv22
  // round B: block row 3
  {{{-1, 0, 0, 0, 0, 0}, {-1, 0, 0, 0, 0, 0}},
   {{3, 0, 2, 1, 2, 2}, {-1, 0, 0, 0, 0, 0}},                                  // wave 1: Linv30 = -Linv33 (S1 of wave 2 + L32 Linv20)
   {{3, 1, 2, 1, 1, 2}, {-1, 0, 0, 0, 0, 0}},                                  // wave 2: Linv31 = -Linv33 (S0 + L32 Linv21)
   {{3, 2, 0, 0, 1, 2}, {-1, 0, 0, 0, 0, 0}}},                                 // wave 3: Linv32 = -Linv33 S0
};
__device__ __forceinline__ void inverse_task(const double (*L)[kLS], double (*Li)[kLS], double (*Tmp)[17], double (*S0)[17],
                                             const InvTask t, int li, int lk) {
  if (t.I < 0) return;
  d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
  if (t.init) {
    const double (*src)[17] = t.init == 1 ? S0 : Tmp + 32 * 2 + 16;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) acc[reg] = src[lk + 4 * reg][li];
  }
  for (int p = 0; p < t.np; ++p) acc = block_mma(&L[16 * t.I][16 * (t.K0 + p)], &Li[16 * (t.K0 + p)][16 * t.J], acc, li, lk);
  if (t.out < 2) {
    block_store(&S0[16 * t.out][0], 17, acc, 1.0, li, lk);
    return;
  }
  if (t.np > 0 || t.init != 1) {   // otherwise S0 already holds acc
    lds_wave_sync();               // all lanes have read S0 before it is overwritten
    block_store(&S0[0][0], 17, acc, 1.0, li, lk);
  }
  lds_wave_sync();
  block_store(&Li[16 * t.I][16 * t.J], kLS, block_mma_t(&Li[16 * t.I][16 * t.I], S0, d4_t{0.0, 0.0, 0.0, 0.0}, li, lk), -1.0, li, lk);
}

// A 16-row strip of L_ik = A_ik Linv_k^T rebuilt in registers exactly as the T task forms the tile (same MFMA order, so the same bits):
// lane (li, lk) of the product D[c][r] ends up holding L(16 b + li, 16 cb + lk + 4 reg) in strip[4 cb + reg], which IS the operand
// element of k-step 4 cb + reg of an update product - no exchange between lanes. Two phases, so that a caller can put other work between
// the issue of the loads and their use.
__device__ const int kPreBi[4][3] = {{0, 3, 3}, {1, 1, 3}, {2, 2, 2}, {3, -1, -1}}, kPreBj[4][3] = {{0, 0, 1}, {0, 1, 2}, {0, 1, 2}, {3, 0, 0}};
struct StripOperands { double xv[16], yv[40]; };   // yv: column blocks 0..3 of Linv, 4 (cb + 1) k-steps each (the inverse is triangular)
__device__ __forceinline__ void sp_strip_load(const double* __restrict__ X, const double* __restrict__ Linv, int li, int lk, StripOperands& o) {
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) o.xv[ks] = X[ks * 256];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const double* __restrict__ Y = Linv + 16 * cb + li + lk * 64;
#pragma unroll
    for (int ks = 0; ks < 4 * (cb + 1); ++ks) o.yv[2 * cb * (cb + 1) + ks] = Y[ks * 256];
  }
}
__device__ __forceinline__ void sp_strip_compute(const StripOperands& o, double (&strip)[16]) {
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4 * (cb + 1); ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.yv[2 * cb * (cb + 1) + ks], o.xv[ks], acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) strip[4 * cb + reg] = acc[reg];
  }
}

// Eight pivots of the panel factorisation without a single cross-lane operand: every lane holds the (partially updated) 8 x 8 diagonal
// block dg (lower triangle, row s at s (s + 1) / 2) and its own row a[0..7] of the same eight columns. Right-looking on unscaled columns,
// exactly the recurrence of the rows themselves: x_t -= (x_j / d_j) dg[t][j] for t > j. Out: the multipliers lj[j] = a_j / d_j of the own
// row, the pivots (1.0 where the block is not positive definite: ok turns false, uniformly - every lane sees the same dg).
template <int G>
__device__ __forceinline__ bool panel_factor(double (&dg)[G * (G + 1) / 2], double* __restrict__ a, double (&lj)[G], double* __restrict__ piv, bool ok) {
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const double dj = dg[j * (j + 1) / 2 + j];
    ok = ok && (dj > 0.0) && isfinite(dj);
    const double dsafe = ok ? dj : 1.0;
    piv[j] = dsafe;
    const double r = fast_rcp(dsafe);
    lj[j] = a[j] * r;
#pragma unroll
    for (int t = j + 1; t < G; ++t) a[t] -= lj[j] * dg[t * (t + 1) / 2 + j];
#pragma unroll
    for (int s_ = j + 1; s_ < G; ++s_) {
      const double m = dg[s_ * (s_ + 1) / 2 + j] * r;
#pragma unroll
      for (int t = j + 1; t <= s_; ++t) dg[s_ * (s_ + 1) / 2 + t] -= m * dg[t * (t + 1) / 2 + j];
    }
  }
  return ok;
}
// (round 6) The sub-panel width: 16 pivots as 16 / kPanelG sub-panels whose kPanelG x kPanelG diagonal block every lane factors redundantly.
// The redundant block costs G^3 / 6 multiply-adds per sub-panel in every lane: 84 at G = 8 (two sub-panels: 168 + the two hand-overs),
// 10 at G = 4 (four sub-panels: 40 + six hand-overs through the wave's scratch block). Same operations on the same values in the same order
// for every G (a_rt -= (a_rj / d_j) a_pt, j ascending): same bits. -DMVGX_BA_PANEL_G=8: the round-5 form (A/B runs).
#ifndef MVGX_BA_PANEL_G
#define MVGX_BA_PANEL_G 4
#endif
constexpr int kPanelG = MVGX_BA_PANEL_G;
static_assert(16 % kPanelG == 0 && kPanelG >= 2, "sub-panels of the 16-column panel");

// Factor the kb x kb diagonal block (kb <= 64, identity-padded to 64) and invert the factor. One workgroup; the block is
// processed as four 64 x 16 column panels: a register-resident panel factorisation by one wave, then a rank-16 update
// of the remaining columns by all four. The inverse is built block by block (16 x 16) by the three waves the panel
// factorisation leaves idle.
// kDense: the factor goes back into A and the inverse is stored twice (k-major for the panel GEMM, row-major for the
// back substitution); otherwise (block-sparse solver) only the k-major inverse is kept.
// MVGX_BA_FACTOR_DEBUG=1: shader-clock stamps of the phases of the factor-and-invert kernel (workgroup 0), printed at destroy
__device__ long long g_factor_stamps[24];
__device__ int g_factor_debug;
#define MVGX_STAMP(i) do { if (stamping) g_factor_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
// (inside the second panel slot, wave 0: waits for the wave's outstanding LDS operations first, so that a stamp closes what precedes it)
#define MVGX_STAMP_W(i) do { if (stamping && jb == 1) { __builtin_amdgcn_s_waitcnt(0xc07f); g_factor_stamps[i] = __builtin_amdgcn_s_memtime(); } } while (0)
// kPre (block-sparse solver, look-ahead schedule): before the factorisation the workgroup subtracts the contributions of the columns j of
// the level before from the tile - sum_j L_kj L_kj^T with L_kj = A_kj Linv_j^T rebuilt here, strip by strip, in the T task's arithmetic,
// and accumulated over j in the U task's order (16 x 16 sub-blocks, contributors ascending): the bits of the level-by-level schedule.
template <bool kDense, bool kPre = false>
__device__ __forceinline__ void chol_diag_inv_body(double* __restrict__ A, int ld, int k0, int kb,
                                                   double* __restrict__ linv /* [k][c] = Linv[c][k], then [r][c] */, int* fail,
                                                   double* lds, const SpSys* sp = nullptr, int kcol = 0) {
  double (*L)[kLS] = reinterpret_cast<double (*)[kLS]>(lds);
  double (*Li)[kLS] = reinterpret_cast<double (*)[kLS]>(lds + 64 * kLS);
  double (*Tmp)[17] = reinterpret_cast<double (*)[17]>(lds + 2 * 64 * kLS);     // two 16 x 16 scratch blocks per wave
  double* rd = lds + 2 * 64 * kLS + 128 * 17;                                     // 1 / L[r][r]
  double* flag = rd + 64;
  const int tid = threadIdx.x;
  const bool stamping = g_factor_debug && blockIdx.x == 0 && tid == 0;   // read once: one global load, not one per stamp
  MVGX_STAMP(0);
  int pre0 = 0, pre1 = 0;
  StripOperands so;
  if constexpr (kPre) {
    pre0 = sp->pre_start[kcol]; pre1 = sp->pre_start[kcol + 1];
    if (pre0 < pre1 && tid < 256)   // the first contributor's operands travel together with the tile itself
      sp_strip_load(sp->A + (size_t)sp->pre_slot[pre0] * 4096 + 16 * (tid >> 6) + (tid & 15) + ((tid & 63) >> 4) * 64,
                    sp->Linv + (size_t)sp->pre_col[pre0] * 4096, tid & 15, (tid & 63) >> 4, so);
  }
  {
    // all 16 loads of a thread are issued before the first LDS store: as a rolled loop every element paid a full memory
    // round trip (the compiler keeps load -> wait -> store per iteration), ~11 of the kernel's 25 microseconds
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = tid + 256 * i, c = q >> 6, r = q & 63;
      const bool inside = r < kb && c < kb;
      v[i] = (r == c && !inside) ? 1.0 : 0.0;   // identity padding of a partial last block
      if (inside && r >= c) v[i] = A[(size_t)(k0 + c) * ld + (k0 + r)];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = tid + 256 * i, c = q >> 6, r = q & 63;
      L[r][c] = v[i];
      Li[r][c] = 0.0;
    }
  }
  __syncthreads();
  // the wave index as a scalar: everything wave-dependent below is a scalar branch or a table lookup, no exec-masked regions
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
  if constexpr (kPre) {
    if (pre0 < pre1) {   // (uniform)
      // the ten 16 x 16 sub-blocks of the lower triangle over the four waves: (row block, column block) per wave and turn, -1: idle
      const int (*sbi)[3] = kPreBi, (*sbj)[3] = kPreBj;
      d4_t acc[3] = {d4_t{0.0, 0.0, 0.0, 0.0}, d4_t{0.0, 0.0, 0.0, 0.0}, d4_t{0.0, 0.0, 0.0, 0.0}};
      for (int q = pre0; q < pre1; ++q) {
        if (q > pre0) {
          sp_strip_load(sp->A + (size_t)sp->pre_slot[q] * 4096 + 16 * wave + li + lk * 64, sp->Linv + (size_t)sp->pre_col[q] * 4096, li, lk, so);
          __syncthreads();   // every wave has read the tile of the contributor before
        }
        double strip[16];
        sp_strip_compute(so, strip);
        // L_kj staged where the inverse will be built (Li is cleared again below): row 16 wave + li, column 16 cb + lk + 4 reg
#pragma unroll
        for (int e = 0; e < 16; ++e) Li[16 * wave + li][16 * (e >> 2) + lk + 4 * (e & 3)] = strip[e];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int bi = sbi[wave][t], bj = sbj[wave][t];
          if (bi < 0) continue;   // (wave-uniform)
#pragma unroll
          for (int ks = 0; ks < 16; ++ks)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[16 * bj + li][4 * ks + lk], Li[16 * bi + li][4 * ks + lk], acc[t], 0, 0, 0);
        }
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int bi = sbi[wave][t], bj = sbj[wave][t];
        if (bi < 0) continue;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {   // element (row 16 bi + li, column 16 bj + lk + 4 reg) of the tile: the U task's destination
          const int r = 16 * bi + li, cc = 16 * bj + lk + 4 * reg;
          if (r >= cc) L[r][cc] -= acc[t][reg];
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { const int q = tid + 256 * i; Li[q & 63][q >> 6] = 0.0; }
      __syncthreads();
    }
  }
  MVGX_STAMP(1);
  // The inverse of the factor, Linv[I][J] = -Linv[I][I] sum_{K=J}^{I-1} L[I][K] Linv[K][J] over 16 x 16 blocks, is built by waves
  // 1..3 WHILE wave 0 factors the next panel (the panel chain is serial and one wave wide): a piece is scheduled into the first
  // panel slot in which its inputs are final (kInvSchedule), so that only the last block row is left for after the loop.
  // Products run on the f64 matrix core with operands read from LDS; a wave keeps partial sums in two private 16 x 16 blocks.
  double (*T0)[17] = Tmp + 32 * wave;
  // (unrolled: as a loop - 2 000 instead of 6 600 instructions - every slot was 10 % slower, call r5_08: the kernel is not bound by instruction fetch)
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    const int j0 = jb * 16;
    MVGX_STAMP(2 + jb);
    // Panel of 16 columns, factored by wave 0 alone with lane = row and the row's 16 panel entries in registers: the pivot
    // column reaches the other lanes through v_readlane (scalar operands), so the 16 pivots cost neither LDS traffic nor
    // barriers. Right-looking on UNscaled columns: a_rt -= (a_rj / d_j) a_pt for t > j, p = j0 + j; the 1 / sqrt(d)
    // scaling is applied when the panel is written back. Rows above the pivot compute garbage that is never stored.
    if (wave == 0) {
      // Round 5: the 16 columns as sub-panels (two of 8 then; four of 4 since round 6) whose diagonal block every lane factors REDUNDANTLY in its
      // own registers (panel_factor): no pivot crosses a lane any more. The v_readlane-fed form (one pivot = 2 + 2 (15 - j) readlanes in front of the
      // multiply-adds, ~380 clocks) gave way to ~36 broadcast LDS reads per sub-panel; between the two sub-panels the eight rows
      // j0 + 8 .. j0 + 15 hand their entries to every lane through the wave's scratch block. Same operations on the same values in the
      // same order as the one-pivot-at-a-time form (a_rt -= (a_rj / d_j) a_pt, unscaled columns, 1 / sqrt(d) at the write-back): same bits.
      double a[16], dg[kPanelG * (kPanelG + 1) / 2], lj[kPanelG], piv[16];
      double* X = &T0[0][0];   // wave 0's scratch blocks (the inverse schedule never gives wave 0 a task): (16 - G) G + G G doubles used
      double* X2 = X + (16 - kPanelG) * kPanelG;
      MVGX_STAMP_W(10);
#pragma unroll
      for (int t = 0; t < 16; ++t) a[t] = L[lane][j0 + t];
#pragma unroll
      for (int s_ = 0; s_ < kPanelG; ++s_)
#pragma unroll
        for (int t = 0; t <= s_; ++t) dg[s_ * (s_ + 1) / 2 + t] = L[j0 + s_][j0 + t];   // wave-uniform addresses: broadcast reads
      MVGX_STAMP_W(11);
      bool ok = true;
#pragma unroll
      for (int c0 = 0; c0 < 16; c0 += kPanelG) {
        ok = panel_factor<kPanelG>(dg, a + c0, lj, piv + c0, ok);
        if (c0 + kPanelG >= 16) break;
        // the columns right of the sub-panel: a[c0 + G + t] -= sum_k (a_rk / d_k) U[t][k], U = the unscaled entries (columns c0 .. c0 + G - 1)
        // of rows j0 + c0 + G + t, k ascending - handed to every lane through the wave's scratch block
        const int nrem = 16 - (c0 + kPanelG);
        const int row0 = j0 + c0 + kPanelG;
        if ((unsigned)(lane - row0) < (unsigned)nrem) {
#pragma unroll
          for (int k = 0; k < kPanelG; ++k) X[(lane - row0) * kPanelG + k] = a[c0 + k];
        }
        lds_wave_sync();
#pragma unroll
        for (int t = 0; t < nrem; ++t) {
          double u[kPanelG];
#pragma unroll
          for (int k = 0; k < kPanelG; ++k) u[k] = X[t * kPanelG + k];
#pragma unroll
          for (int k = 0; k < kPanelG; ++k) a[c0 + kPanelG + t] -= lj[k] * u[k];
        }
        // the next diagonal block: rows row0 .. row0 + G - 1 hand over their entries of columns c0 + G .. c0 + 2 G - 1
        if ((unsigned)(lane - row0) < (unsigned)kPanelG) {
#pragma unroll
          for (int t = 0; t < kPanelG; ++t) X2[(lane - row0) * kPanelG + t] = a[c0 + kPanelG + t];
        }
        lds_wave_sync();   // (also: every lane has read U before the next sub-panel overwrites it)
#pragma unroll
        for (int s_ = 0; s_ < kPanelG; ++s_)
#pragma unroll
          for (int t = 0; t <= s_; ++t) dg[s_ * (s_ + 1) / 2 + t] = X2[s_ * kPanelG + t];
        lds_wave_sync();   // (X2 is read before the next hand-over writes it)
      }
      MVGX_STAMP_W(15);
      double dmine = 1.0;   // lane j keeps pivot j: the 16 square roots are taken once, in parallel, after the chain
#pragma unroll
      for (int j = 0; j < 16; ++j) dmine = (lane == j) ? piv[j] : dmine;
      const double rsm = 1.0 / sqrt(dmine);
      if (lane == 0) flag[0] = ok ? 1.0 : -1.0;   // for the uniform exit below
      if (lane < 16) rd[j0 + lane] = rsm;          // 1 / L[r][r]
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const double rst = readlane_f64(rsm, t);   // wave collective: outside the lane-dependent select
        L[lane][j0 + t] = (lane >= j0 + t) ? a[t] * rst : 0.0;
      }
      MVGX_STAMP_W(16);
      if (jb == 3) {   // the last diagonal block is inverted by the wave that produced it
        lds_wave_sync();
        diag_block_inverse(L, Li, rd, 48, lane);
      }
    } else if (jb >= 1) {
      if (wave == 1) diag_block_inverse(L, Li, rd, 16 * (jb - 1), lane);
      if (jb >= 2) {
        inverse_task(L, Li, Tmp, T0, kInvSchedule[jb - 2][wave][0], li, lk);
        inverse_task(L, Li, Tmp, T0, kInvSchedule[jb - 2][wave][1], li, lk);
      }
    }
    __syncthreads();
    MVGX_STAMP_W(17);
    if (!(flag[0] > 0.0)) { if (tid == 0) atomicExch(fail, 2); return; }   // uniform: not positive definite
    if (jb == 3) break;
    // rank-16 update of the columns right of the panel, one 16 x 16 block (I, J), I >= J > jb, per wave and turn:
    // D = P_I P_J^T with P_X = L[16 X .. 16 X + 15][j0 .. j0 + 15] on the f64 matrix core (4 k-steps of 4)
    {
      const int nbk = 3 - jb, ntile = nbk * (nbk + 1) / 2;
      for (int tile = wave; tile < ntile; tile += 4) {
        int bi = 0, bj = tile;
        while (bj > bi) { bj -= bi + 1; ++bi; }
        const int r0 = j0 + 16 + 16 * bi, c0 = j0 + 16 + 16 * bj;
        d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(L[r0 + li][j0 + 4 * ks + lk], L[c0 + li][j0 + 4 * ks + lk], acc, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) L[r0 + lk + 4 * reg][c0 + li] -= acc[reg];
      }
    }
    __syncthreads();
    MVGX_STAMP_W(18);
  }
  MVGX_STAMP(6);
  // what is left: block rows 2 (columns 0, 1) and 3, in two rounds
  inverse_task(L, Li, Tmp, T0, kInvSchedule[2][wave][0], li, lk);
  inverse_task(L, Li, Tmp, T0, kInvSchedule[2][wave][1], li, lk);
  __syncthreads();
  MVGX_STAMP(7);
  inverse_task(L, Li, Tmp, T0, kInvSchedule[3][wave][0], li, lk);
  __syncthreads();
  MVGX_STAMP(8);
  for (int q = tid; q < 4096; q += 256) {
    const int c = q >> 6, r = q & 63;
    if (kDense && r < kb && c < kb && r >= c) A[(size_t)(k0 + c) * ld + (k0 + r)] = L[r][c];
    linv[q] = Li[r][c];                  // k-major: linv[k = c][col = r] = Linv[r][c]
    if (kDense) linv[4096 + q] = Li[q >> 6][q & 63]; // row-major
  }
  MVGX_STAMP(9);
}
__global__ __launch_bounds__(256) void chol_diag_inv_kernel(double* __restrict__ A, int ld, int k0, int kb, double* __restrict__ linv, int* fail) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  chol_diag_inv_body<true>(A, ld, k0, kb, linv, fail, lds);
}

// ------------------------------------------------------------------------------------------------------
// Block-sparse Cholesky of the reduced camera system (symbolic phase: ba_sparse_plan.h). One level of the elimination
// tree of the tile columns per round of launches: F factors + inverts the diagonal tiles of the level, T forms the tiles
// below them, U applies their outer products to the tiles of later columns. A wave owns one 16 x 16 sub-block of a
// destination tile and walks its contributors in a fixed order: operands go straight from L2 into the f64 MFMA (the
// tiles are stored so that an operand fragment is 16 consecutive doubles), no LDS, no barriers, no atomics.
// ------------------------------------------------------------------------------------------------------
constexpr long long kZNotYet = 0x7FF84D5647580001ll;   // a quiet NaN no arithmetic produces: "this entry of z has not been solved in this solve"
__global__ __launch_bounds__(256) void sp_factor_kernel(SpSys s, int f0, int* fail) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int k = s.f_cols[f0 + blockIdx.x];
  // (every column is factored once per solve, before the reverse sweep: its part of the solution starts as "not there yet" - see
  // sp_backsolve_all_kernel - at no launch of its own)
  if (threadIdx.x < 64) s.z[(size_t)k * 64 + threadIdx.x] = __longlong_as_double(kZNotYet);
  double* A = s.A + (size_t)s.tmap[(size_t)k * s.nT + k] * 4096;
  chol_diag_inv_body<false>(A, 64, 0, s.tile_kb[k], s.Linv + (size_t)k * 4096, fail, lds);
}

// kUpdate false: T, dst(L) = X(A) Linv_k^T (only the q <= column part of the triangular inverse is walked);
// kUpdate true:  U, dst(A) -= sum over contributors X(L) Y(L)^T. The contributor list of a task is fetched 64 entries at a
// time (one per lane, handed out by v_readlane) and the 32 operand loads of contributor c + 1 are in flight while the 16
// MFMAs of contributor c run: as a plain loop every contributor cost two dependent memory round trips (index, then operands),
// which was the whole 23 microseconds of this kernel.
__device__ __forceinline__ void sp_load_operands(const double* __restrict__ X, const double* __restrict__ Y, double (&xv)[16], double (&yv)[16]) {
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) { xv[ks] = X[ks * 256]; yv[ks] = Y[ks * 256]; }
}
template <bool kUpdate>
__device__ __forceinline__ void sp_gemm_task(const SpSys& s, int task) {   // one wave, one task (wave-uniform)
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  const mvgx_sparse::GemmTask g = (kUpdate ? s.u_tasks : s.t_tasks)[task];
  const mvgx_sparse::SlotPair* __restrict__ pairs = kUpdate ? s.u_pairs : s.t_pairs;
  d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
  // lane (li, lk) feeds X[16 bi + li][4 ks + lk] and Y[16 bj + li][4 ks + lk]; element (r, q) of a tile sits at q * 64 + r
  const size_t xoff = 16 * g.bi + li + lk * 64, yoff = 16 * g.bj + li + lk * 64;
  double* __restrict__ dst = (kUpdate ? s.A : s.L) + (size_t)g.dst * 4096 + (size_t)(16 * g.bj + lk) * 64 + 16 * g.bi + li;
  double old[4] = {0.0, 0.0, 0.0, 0.0};
  if constexpr (kUpdate) {   // the destination is read up front, not after the last MFMA
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) old[reg] = dst[reg * 256];
  }
  if constexpr (!kUpdate) {
    const int ksteps = 4 * (g.bj + 1);
    const mvgx_sparse::SlotPair p = pairs[g.c0];
    const double* __restrict__ X = s.A + (size_t)p.a * 4096 + xoff;
    const double* __restrict__ Y = s.Linv + (size_t)p.b * 4096 + yoff;
    double xv[16], yv[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      xv[ks] = ks < ksteps ? X[ks * 256] : 0.0;
      yv[ks] = ks < ksteps ? Y[ks * 256] : 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (ks < ksteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(yv[ks], xv[ks], acc, 0, 0, 0);   // D[c][r]: stores coalesce along r
  } else {
    for (int cbase = g.c0; cbase < g.c1; cbase += 64) {
      const int n = min(64, g.c1 - cbase);
      const mvgx_sparse::SlotPair mine = lane < n ? pairs[cbase + lane] : mvgx_sparse::SlotPair{0, 0};
      double xa[16], ya[16], xb[16], yb[16];
      {
        const int pa = __builtin_amdgcn_readlane(mine.a, 0), pb = __builtin_amdgcn_readlane(mine.b, 0);
        sp_load_operands(s.L + (size_t)pa * 4096 + xoff, s.L + (size_t)pb * 4096 + yoff, xa, ya);
      }
      for (int i = 0; i < n; i += 2) {
        if (i + 1 < n) {
          const int pa = __builtin_amdgcn_readlane(mine.a, i + 1), pb = __builtin_amdgcn_readlane(mine.b, i + 1);
          sp_load_operands(s.L + (size_t)pa * 4096 + xoff, s.L + (size_t)pb * 4096 + yoff, xb, yb);
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[ks], xa[ks], acc, 0, 0, 0);
        if (i + 1 >= n) break;
        if (i + 2 < n) {
          const int pa = __builtin_amdgcn_readlane(mine.a, i + 2), pb = __builtin_amdgcn_readlane(mine.b, i + 2);
          sp_load_operands(s.L + (size_t)pa * 4096 + xoff, s.L + (size_t)pb * 4096 + yoff, xa, ya);
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(yb[ks], xb[ks], acc, 0, 0, 0);
      }
    }
  }
  if constexpr (kUpdate) {
    if (g.n_chunks > 1) {   // wave-uniform: a chunk of a long contributor list
      double* __restrict__ part = s.u_scratch + (size_t)(g.scratch + g.chunk) * 256 + lane;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) part[reg * 64] = acc[reg];
      __threadfence();   // the partial sums are visible device-wide before this chunk counts as arrived
      unsigned arrived = 0;
      if (lane == 0) arrived = atomicAdd(s.u_counter + g.group, 1u);
      arrived = __builtin_amdgcn_readfirstlane(arrived);
      if (arrived + 1 != (unsigned)g.n_chunks) return;
      __threadfence();   // last arrival: the other chunks' partial sums are read from memory, in chunk order
      double tot[4] = {0.0, 0.0, 0.0, 0.0};
      const double* __restrict__ all = s.u_scratch + (size_t)g.scratch * 256 + lane;
      for (int ch = 0; ch < g.n_chunks; ++ch) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) tot[reg] += __builtin_nontemporal_load(all + (size_t)ch * 256 + reg * 64);
      }
      if (lane == 0) s.u_counter[g.group] = 0;   // ready for the next factorisation
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) dst[reg * 256] = old[reg] - tot[reg];
      return;
    }
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) dst[reg * 256] = kUpdate ? old[reg] - acc[reg] : acc[reg];
}
// A level with ONE tile column k (the chain at the top of the elimination tree): its update tasks need nothing but the tiles A_ik,
// A_jk below the diagonal and Linv_k - the two 16-row strips of L they multiply are rebuilt in registers exactly as the T task
// forms them (same MFMA order, so the same bits): lane (li, lk) of the T product D[c][r] ends up holding L(16 b + li, 16 cb + lk +
// 4 reg), which IS the operand element of k-step 4 cb + reg of the update product - no exchange between lanes. The T tasks (the
// stored L tiles are still needed by the reverse sweep) then run beside the update tasks in one launch instead of in front of them:
// one launch boundary less per chain level.
__device__ __forceinline__ void sp_strip_from_a(const double* __restrict__ X, const double* __restrict__ Linv, int li, int lk, double (&strip)[16]) {
  double xv[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) xv[ks] = X[ks * 256];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const double* __restrict__ Y = Linv + 16 * cb + li + lk * 64;
    double yv[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) yv[ks] = ks < 4 * (cb + 1) ? Y[ks * 256] : 0.0;
    d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (ks < 4 * (cb + 1)) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(yv[ks], xv[ks], acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) strip[4 * cb + reg] = acc[reg];
  }
}
__global__ __launch_bounds__(256) void sp_chain_level_kernel(SpSys s, int t0, int n_t, int u0, int n_u, int k) {
  const int t = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (t >= n_t + n_u) return;   // wave-uniform
  if (t >= n_u) { sp_gemm_task<false>(s, t0 + (t - n_u)); return; }   // (the update tasks first: they are what the next level waits for)
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  const mvgx_sparse::GemmTask g = s.u_tasks[u0 + t];
  const mvgx_sparse::SlotPair p = s.u_pairs[g.c0];   // one contributor: the level's column
  double* __restrict__ dst = s.A + (size_t)g.dst * 4096 + (size_t)(16 * g.bj + lk) * 64 + 16 * g.bi + li;
  double old[4];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) old[reg] = dst[reg * 256];
  const double* __restrict__ Linv = s.Linv + (size_t)k * 4096;
  double la[16], lb[16];
  sp_strip_from_a(s.A + (size_t)p.a * 4096 + 16 * g.bi + li + lk * 64, Linv, li, lk, la);
  sp_strip_from_a(s.A + (size_t)p.b * 4096 + 16 * g.bj + li + lk * 64, Linv, li, lk, lb);
  d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lb[ks], la[ks], acc, 0, 0, 0);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) dst[reg * 256] = old[reg] - acc[reg];
}

template <bool kUpdate>
__global__ __launch_bounds__(256) void sp_gemm_kernel(SpSys s, int t0, int n_tasks) {
  const int t = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (t >= n_tasks) return;   // wave-uniform
  sp_gemm_task<kUpdate>(s, t0 + t);
}

// The U task of the look-ahead schedule: the tiles L_ik, L_jk it multiplies are being written by the T tasks of the SAME launch, so the two
// 16-row strips are rebuilt from A_ik, A_jk and Linv_k (sp_strip_*: the T task's arithmetic, the same bits) - the generalisation of
// sp_chain_level_kernel's update to any number of contributors and to split contributor lists. Same sums in the same order as
// sp_gemm_task<true>.
__device__ __forceinline__ void sp_update_task_rebuilding(const SpSys& s, int task) {   // one wave, one task (wave-uniform)
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  const mvgx_sparse::GemmTask g = s.u_tasks[task];
  const mvgx_sparse::SlotPair* __restrict__ pairs = s.u_pairs;
  d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
  const size_t xoff = 16 * g.bi + li + lk * 64, yoff = 16 * g.bj + li + lk * 64;
  double* __restrict__ dst = s.A + (size_t)g.dst * 4096 + (size_t)(16 * g.bj + lk) * 64 + 16 * g.bi + li;
  double old[4];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) old[reg] = dst[reg * 256];
  for (int cbase = g.c0; cbase < g.c1; cbase += 64) {
    const int n = min(64, g.c1 - cbase);
    const mvgx_sparse::SlotPair mine = lane < n ? pairs[cbase + lane] : mvgx_sparse::SlotPair{0, 0};
    const int mycol = s.slot_col[mine.a];
    for (int i = 0; i < n; ++i) {
      const int pa = __builtin_amdgcn_readlane(mine.a, i), pb = __builtin_amdgcn_readlane(mine.b, i), k = __builtin_amdgcn_readlane(mycol, i);
      const double* __restrict__ Linv = s.Linv + (size_t)k * 4096;
      double la[16], lb[16];
      {
        StripOperands o;
        sp_strip_load(s.A + (size_t)pa * 4096 + xoff, Linv, li, lk, o);
        sp_strip_compute(o, la);
        if (pa == pb && g.bi == g.bj) {   // (wave-uniform) a diagonal sub-block of a diagonal tile: one strip, twice
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) lb[ks] = la[ks];
        } else {
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) o.xv[ks] = s.A[(size_t)pb * 4096 + yoff + ks * 256];
          sp_strip_compute(o, lb);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lb[ks], la[ks], acc, 0, 0, 0);
    }
  }
  if (g.n_chunks > 1) {   // wave-uniform: a chunk of a long contributor list (see sp_gemm_task)
    double* __restrict__ part = s.u_scratch + (size_t)(g.scratch + g.chunk) * 256 + lane;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) part[reg * 64] = acc[reg];
    __threadfence();
    unsigned arrived = 0;
    if (lane == 0) arrived = atomicAdd(s.u_counter + g.group, 1u);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived + 1 != (unsigned)g.n_chunks) return;
    __threadfence();
    double tot[4] = {0.0, 0.0, 0.0, 0.0};
    const double* __restrict__ all = s.u_scratch + (size_t)g.scratch * 256 + lane;
    for (int ch = 0; ch < g.n_chunks; ++ch) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) tot[reg] += __builtin_nontemporal_load(all + (size_t)ch * 256 + reg * 64);
    }
    if (lane == 0) s.u_counter[g.group] = 0;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) dst[reg * 256] = old[reg] - tot[reg];
    return;
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) dst[reg * 256] = old[reg] - acc[reg];
}

// One launch per level of the elimination tree (look-ahead schedule): workgroups [0, nf) factor + invert the diagonal tiles of the level
// (their first four waves; with `pre` they first take the contributions of the level before, chol_diag_inv_body<false, true>), the others
// run the U tasks [u0, u0 + nu) and T tasks [t0, t0 + nt) of the level BEFORE, one per wave. Nothing in a launch depends on anything else
// in it: the factor workgroups read A_kk, A_kj, Linv_j - final since the launch before; the U tasks write tiles of later columns only,
// never a diagonal tile of this level; the T tasks write L. 84 KB of LDS per workgroup (one per CU) is what the factor needs; with eight
// waves a CU still holds 8 T / U tasks.
constexpr int kLevelThreads = 512;
__global__ __launch_bounds__(kLevelThreads) void sp_level_kernel(SpSys s, int f0, int nf, int pre, int u0, int nu, int t0, int nt, int* fail) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if ((int)blockIdx.x < nf) {
    if (threadIdx.x >= 256) return;   // (whole waves: the barriers of the factor count the four that stay)
    const int k = s.f_cols[f0 + blockIdx.x];
    if (threadIdx.x < 64) s.z[(size_t)k * 64 + threadIdx.x] = __longlong_as_double(kZNotYet);   // (as sp_factor_kernel)
    double* A = s.A + (size_t)s.tmap[(size_t)k * s.nT + k] * 4096;
    if (pre) chol_diag_inv_body<false, true>(A, 64, 0, s.tile_kb[k], s.Linv + (size_t)k * 4096, fail, lds, &s, k);
    else chol_diag_inv_body<false, false>(A, 64, 0, s.tile_kb[k], s.Linv + (size_t)k * 4096, fail, lds);
    return;
  }
  const int t = ((int)blockIdx.x - nf) * (kLevelThreads / 64) + (int)(threadIdx.x >> 6);
  if (t < nu) sp_update_task_rebuilding(s, u0 + t);                  // (the update tasks first: the longer ones)
  else if (t < nu + nt) sp_gemm_task<false>(s, t0 + (t - nu));
}

// Reverse sweep, one workgroup per tile column of the level: z_k = Linv_k^T (y_k - sum over the tiles below L_ik^T z_i).
// Round 5: the workgroup starts from ONE 64-byte record (column, rhs slot, its first six (slot, row) entries; wave-uniform, so scalar
// loads) and then issues the loads of data together - the tiles below with their parts of z two entries at a time, the rhs strip, the
// inverse: two trips to memory per level where the walk f_cols -> bs_start / tmap -> lists -> tiles -> rhs -> inverse took five or six
// (11 us per level for a few hundred multiply-adds). Same sums in the same order.
__device__ __forceinline__ void sp_backsolve_entry_load(const SpSys& s, int slot, int row, int c, int part, double (&tv)[16], double (&zv)[16]) {
  const double* __restrict__ tile = s.L + (size_t)slot * 4096 + c * 64 + part * 16;
  const double* __restrict__ zi = s.z + (size_t)row * 64 + part * 16;
#pragma unroll
  for (int q = 0; q < 16; ++q) { tv[q] = tile[q]; zv[q] = zi[q]; }
}
__device__ __forceinline__ void sp_backsolve_col(const SpSys& s, int f, double* w /* 64 doubles of LDS */) {
  const int tid = threadIdx.x, c = tid >> 2, part = tid & 3;
  const int32_t* __restrict__ rec = s.bs_rec + (size_t)f * 16;
  const int k = rec[0], rhs_slot = rec[1], n = rec[2], e0 = rec[3];
  const double* __restrict__ li = s.Linv + (size_t)k * 4096 + c * 64 + part * 16;   // Linv[q][c] at c * 64 + q
  double lv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) lv[q] = li[q];
  const double yk = part == 0 ? s.L[(size_t)rhs_slot * 4096 + c * 64] : 0.0;         // y_k[c]: row 0 of the rhs tile
  double v = 0;
#pragma unroll
  for (int pr = 0; pr < mvgx_sparse::kBsInline / 2; ++pr) {
    if (2 * pr >= n) break;   // (uniform)
    double ta[16], za[16], tb[16], zb[16];
    sp_backsolve_entry_load(s, rec[4 + 4 * pr], rec[5 + 4 * pr], c, part, ta, za);
    const bool two = 2 * pr + 1 < n;
    if (two) sp_backsolve_entry_load(s, rec[6 + 4 * pr], rec[7 + 4 * pr], c, part, tb, zb);
#pragma unroll
    for (int q = 0; q < 16; ++q) v += ta[q] * za[q];
    if (two) {
#pragma unroll
      for (int q = 0; q < 16; ++q) v += tb[q] * zb[q];
    }
  }
  // columns with more tiles below than a record holds: the rest of the list, its indices fetched 64 at a time (one per lane, v_readlane
  // hands them out)
  const int lane = tid & 63;
  for (int ebase = e0 + mvgx_sparse::kBsInline; ebase < e0 + n; ebase += 64) {
    const int m = min(64, e0 + n - ebase);
    const int my_slot = lane < m ? s.bs_slot[ebase + lane] : 0, my_row = lane < m ? s.bs_row[ebase + lane] : 0;
    for (int i = 0; i < m; ++i) {
      double tv[16], zv[16];
      sp_backsolve_entry_load(s, __builtin_amdgcn_readlane(my_slot, i), __builtin_amdgcn_readlane(my_row, i), c, part, tv, zv);
#pragma unroll
      for (int q = 0; q < 16; ++q) v += tv[q] * zv[q];
    }
  }
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  if (part == 0) w[c] = yk - v;
  __syncthreads();
  double u = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) u += lv[q] * w[part * 16 + q];
  u += __shfl_xor(u, 1);
  u += __shfl_xor(u, 2);
  if (part == 0) s.z[(size_t)k * 64 + c] = u;
}
__global__ __launch_bounds__(256) void sp_backsolve_kernel(SpSys s, int f0) {
  __shared__ double w[64];
  sp_backsolve_col(s, f0 + (int)blockIdx.x, w);
}

// The WHOLE reverse sweep as one launch (round 5): workgroup b takes the column at position nT - 1 - b of f_cols - the root first - so every
// column a workgroup waits for belongs to a workgroup with a SMALLER index: dispatched earlier, resident or finished, whatever else occupies
// the device (no co-residency assumption, no deadlock). A workgroup fetches everything that does not depend on the solution (its record, the
// inverse, the rhs strip, the tiles below) at once, then waits for the z of the rows it needs - a word per column holding the number of the
// solve whose z it carries, written after the values with device-scope release, polled with device-scope acquire - and only then loads
// those 64 doubles per entry. A level of the sweep then costs one flag hand-over and one trip to memory for z (~3.5 us) instead of a launch,
// its drain and three trips (10 us). Same sums in the same order as sp_backsolve_col.
// Round 6: the hand-over without cache maintenance. What made the round-5 form slow (a hand-over ~15 us) was not the distance between
// the XCDs but the fences: a device-scope release writes the producer's whole L2 back (buffer_wbl2), an acquire invalidates the
// consumer's (buffer_inv) - on every poll. Here every word that crosses workgroups - the flags AND the values - moves with RELAXED
// agent-scope atomic accesses (sc1: they miss the per-CU cache and are coherent across the XCDs by themselves), the producer waits
// for its stores to be acknowledged (s_waitcnt vmcnt(0): a workgroup-scope release fence) before it raises the flag, and nothing else is
// fenced: tools/xcd_handoff.hip measures 0.8 - 1.2 us per hand-over of a 4 KB tile this way, on one XCD or across them (call r6_15).
#ifdef __HIPCC__
__device__ __forceinline__ unsigned flag_load_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_store_relaxed(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double shared_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void shared_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stores_acknowledged() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(1); }
#else   // the HIP emulation runs the workgroups of a launch one after the other, in index order: a flag is always set when it is read
__device__ __forceinline__ unsigned flag_load_relaxed(const unsigned* p) { return *p; }
__device__ __forceinline__ void flag_store_relaxed(unsigned* p, unsigned v) { *p = v; }
__device__ __forceinline__ double shared_load(const double* p) { return *p; }
__device__ __forceinline__ void shared_store(double* p, double v) { *p = v; }
__device__ __forceinline__ void stores_acknowledged() {}
__device__ __forceinline__ void spin_pause() {}
#endif
constexpr int kBsPrefetch = 4;   // entries whose tiles a workgroup holds in registers while it waits
// Round 6: no flags. A column's part of z starts every solve as kZNotYet (written by the column's own factor workgroup) and turns into
// the solution by 64 single 8-byte stores; a consumer polls THE VALUES (lane i of wave w polls element i of the w-th entry it needs,
// into LDS) until none is kZNotYet: one store on the producer's side and one successful poll on the consumer's per level of the sweep,
// where flag + values were two dependent trips each (store, acknowledge, flag | poll, load). A NaN in the solution (a failed
// factorisation: the fail word is set anyway) can never equal kZNotYet; the polls are bounded all the same.
__global__ __launch_bounds__(256) void sp_backsolve_all_kernel(SpSys s, unsigned epoch, int* fail) {
  __shared__ double w[64];
  __shared__ double zsh[kBsPrefetch][64];
  __shared__ int give_up;
  const int tid = threadIdx.x, c = tid >> 2, part = tid & 3, lane = tid & 63, wave = tid >> 6;
  const int f = s.nT - 1 - (int)blockIdx.x;
  const int32_t* __restrict__ rec = s.bs_rec + (size_t)f * 16;
  const int k = rec[0], rhs_slot = rec[1], n = rec[2], e0 = rec[3];
  const double* __restrict__ li = s.Linv + (size_t)k * 4096 + c * 64 + part * 16;
  double lv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) lv[q] = li[q];
  const double yk = part == 0 ? s.L[(size_t)rhs_slot * 4096 + c * 64] : 0.0;
  if (tid == 0) give_up = 0;
  double v = 0;
  static_assert(kBsPrefetch == 4, "one polling wave per entry of a round");
  for (int base = 0; base < n; base += kBsPrefetch) {   // (one round for almost every column: at most four tiles below it)
    // the tile parts of this round's entries: they do not depend on the solution - in flight while the values are awaited
    double tv[kBsPrefetch][16];
#pragma unroll
    for (int i = 0; i < kBsPrefetch; ++i)
      if (base + i < n) {   // (uniform)
        const int e = base + i;
        const int slot = e < mvgx_sparse::kBsInline ? rec[4 + 2 * e] : s.bs_slot[e0 + e];
        const double* __restrict__ tile = s.L + (size_t)slot * 4096 + c * 64 + part * 16;
#pragma unroll
        for (int q = 0; q < 16; ++q) tv[i][q] = tile[q];
      }
    __syncthreads();   // (zsh of the round before has been read; give_up is set)
    if (base + wave < n) {   // wave w awaits entry base + w: 64 values, one per lane
      const int e = base + wave;
      const int row = e < mvgx_sparse::kBsInline ? rec[5 + 2 * e] : s.bs_row[e0 + e];
      const double* src = s.z + (size_t)row * 64 + lane;
      double val = shared_load(src);
      int spins = 0;
      while (__double_as_longlong(val) == kZNotYet) {
        spin_pause();
        if (++spins > (1 << 20)) { give_up = 1; break; }
        val = shared_load(src);
      }
      zsh[wave][lane] = val;
    }
    __syncthreads();
    if (give_up) { if (tid == 0) atomicExch(fail, 3); return; }   // (uniform)
#pragma unroll
    for (int i = 0; i < kBsPrefetch; ++i)
      if (base + i < n) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v += tv[i][q] * zsh[i][part * 16 + q];
      }
  }
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  if (part == 0) w[c] = yk - v;
  __syncthreads();
  double u = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) u += lv[q] * w[part * 16 + q];
  u += __shfl_xor(u, 1);
  u += __shfl_xor(u, 2);
  if (part == 0) shared_store(s.z + (size_t)k * 64 + c, u);
  (void)epoch;
}

// The top of the elimination tree is a chain - one tile column per level, each depending on all the ones above it - and its part
// of the reverse sweep ran as one launch per column, each a series of dependent trips to L2 (column id, list bounds, (slot, row)
// lists, tiles + z, store): ~10 us per level, 12 levels at C5. Here ONE workgroup walks the chain from the root down: the column
// ids, list bounds, lists and rhs slots of the whole chain are fetched into LDS up front, the chain's part of the solution stays
// in LDS, so a step is one trip to L2 for the tiles (independent of z) plus LDS work. Same arithmetic, same order as
// sp_backsolve_col. Columns f_cols[f_lo .. f_lo + n_chain) are the columns of levels level0 .. level0 + n_chain - 1; every tile
// below a chain column lies in a row of the chain (an ancestor).
constexpr int kChainMax = 32, kChainEntriesMax = 1024;
__global__ __launch_bounds__(256) void sp_backsolve_chain_kernel(SpSys s, int f_lo, int n_chain, int level0) {
  __shared__ double zc[kChainMax][64];
  __shared__ double w[64];
  __shared__ int col_k[kChainMax], col_e0[kChainMax + 1], col_y[kChainMax];
  __shared__ int e_slot[kChainEntriesMax], e_pos[kChainEntriesMax];
  const int tid = threadIdx.x, c = tid >> 2, part = tid & 3;
  if (tid < n_chain) {
    const int k = s.f_cols[f_lo + tid];
    col_k[tid] = k;
    col_y[tid] = s.tmap[(size_t)s.nT * s.nT + k];
  }
  __syncthreads();
  if (tid == 0) {   // list offsets of the chain's columns inside e_slot / e_pos
    int at = 0;
    for (int q = 0; q < n_chain; ++q) { col_e0[q] = at; at += s.bs_start[col_k[q] + 1] - s.bs_start[col_k[q]]; }
    col_e0[n_chain] = at;
  }
  __syncthreads();
  for (int q = 0; q < n_chain; ++q) {
    const int b0 = s.bs_start[col_k[q]], n = col_e0[q + 1] - col_e0[q];
    for (int i = tid; i < n; i += 256) {
      e_slot[col_e0[q] + i] = s.bs_slot[b0 + i];
      e_pos[col_e0[q] + i] = s.level_of[s.bs_row[b0 + i]] - level0;
    }
  }
  __syncthreads();
  for (int q = n_chain - 1; q >= 0; --q) {
    const int k = col_k[q];
    // (issued first: neither depends on the solution of the columns above)
    const double* __restrict__ li = s.Linv + (size_t)k * 4096 + c * 64 + part * 16;
    double lv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) lv[j] = li[j];
    const double yk = part == 0 ? s.L[(size_t)col_y[q] * 4096 + c * 64] : 0.0;
    double v = 0;
    for (int e = col_e0[q]; e < col_e0[q + 1]; ++e) {
      const double* __restrict__ tile = s.L + (size_t)e_slot[e] * 4096 + c * 64 + part * 16;
      const double* __restrict__ zi = &zc[e_pos[e]][part * 16];
      double tv[16], zv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) { tv[j] = tile[j]; zv[j] = zi[j]; }
#pragma unroll
      for (int j = 0; j < 16; ++j) v += tv[j] * zv[j];
    }
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    if (part == 0) w[c] = yk - v;
    __syncthreads();
    double u = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) u += lv[j] * w[part * 16 + j];
    u += __shfl_xor(u, 1);
    u += __shfl_xor(u, 2);
    if (part == 0) { zc[q][c] = u; s.z[(size_t)k * 64 + c] = u; }
    __syncthreads();
  }
}

__global__ void sp_gather_solution_kernel(Dev d) {   // also the camera part of the step (ba_backsub_kernel's first line: step = -solution)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  const double z = d.sp.z[d.sp.pcol[i]];
  d.zsol[i] = z;
  d.step_cam[i] = d.cam_active[i] ? -z : 0.0;
}

// 32 x 32 sub-tile of D[c][r] = sum_k Q[k][c] P[k][r] on one wave: 2 x 2 MFMA blocks, MFMA row index = c, column = r.
// Lane l feeds A[i = l & 15][k = l >> 4] = Q[k][cbase + i] and B[k = l >> 4][j = l & 15] = P[k][rbase + j]; it receives
// D[i = (l >> 4) + 4 reg][j = l & 15].
__device__ __forceinline__ void mfma_tile_32x32(const double (*P)[kTS], const double (*Q)[kTS], int kc, int rbase, int cbase,
                                                d4_t acc[2][2]) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  for (int k = 0; k < kc; k += 4) {
    const double q0 = Q[k + lk][cbase + li], q1 = Q[k + lk][cbase + 16 + li];
    const double p0 = P[k + lk][rbase + li], p1 = P[k + lk][rbase + 16 + li];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, p0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, p1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, p0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, p1, acc[1][1], 0, 0, 0);
  }
}

// rows k0 + kb .. n (row n = rhs) of block column k0: X = A21 Linv^T, 64 rows per workgroup. Latency-bound (few
// workgroups per launch): both operands are fetched in one go (2 x 40 KiB of LDS) so the launch costs one memory round trip.
constexpr int kPanelLds = 2 * 64 * kTS * (int)sizeof(double);
__global__ __launch_bounds__(256) void chol_panel_mfma_kernel(double* __restrict__ A, int n, int ld, int k0, int kb,
                                                              const double* __restrict__ linvT) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double (*P)[kTS] = reinterpret_cast<double (*)[kTS]>(lds);
  double (*Q)[kTS] = reinterpret_cast<double (*)[kTS]>(lds + 64 * kTS);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row0 = k0 + kb + (int)blockIdx.x * 64;
  const int rbase = (wave & 1) * 32, cbase = (wave >> 1) * 32;
  d4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = d4_t{0.0, 0.0, 0.0, 0.0};
  {
    double pv[16], qv[16];   // all 32 loads of a thread in flight before the first LDS store (one memory round trip, not sixteen)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = tid + 256 * i, k = q >> 6, r = q & 63;
      pv[i] = (k < kb && row0 + r <= n) ? A[(size_t)(k0 + k) * ld + (row0 + r)] : 0.0;
      qv[i] = linvT[k * 64 + r];   // zero beyond the factor's triangle, identity padding beyond kb
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = tid + 256 * i, k = q >> 6, r = q & 63;
      P[k][r] = pv[i];
      Q[k][r] = qv[i];
    }
  }
  __syncthreads();
  mfma_tile_32x32(P, Q, 64, rbase, cbase, acc);
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int c = cbase + 16 * ci + (lane >> 4) + 4 * reg, r = row0 + rbase + 16 * ri + (lane & 15);
        if (c < kb && r <= n) A[(size_t)(k0 + c) * ld + r] = acc[ci][ri][reg];
      }
}

// Trailing update A(r, c) -= sum_{q in [ks, ks + klen)} L(r, q) L(c, q) on the 64 x 64 tiles (ti >= tj) of the lower
// triangle whose first row / column is t0, for columns c < col_end and rows r <= n (the rhs row rides along). Two uses per
// outer panel of the two-level factorisation: the inner step (klen = 64, columns up to the end of the panel only) and the
// panel's deferred update of everything to its right (klen = panel width: K-times fewer passes over the trailing matrix).
__global__ __launch_bounds__(256) void chol_update_mfma_kernel(double* __restrict__ A, int n, int ld, int ks, int klen, int t0, int col_end) {
  __shared__ double P[32][kTS];   // rows of tile I, k-major
  __shared__ double Q[32][kTS];   // rows of tile J (= columns of the destination tile)
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (ti < tj) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i0 = t0 + ti * 64, j0 = t0 + tj * 64;
  const int rbase = (wave & 1) * 32, cbase = (wave >> 1) * 32;
  d4_t acc[2][2], dst[2][2];   // dst: the destination tile, requested before anything else (one round trip less)
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int c = j0 + cbase + 16 * ci + (lane >> 4) + 4 * reg, r = i0 + rbase + 16 * ri + (lane & 15);
        dst[ci][ri][reg] = (r <= n && c < col_end && r >= c) ? A[(size_t)c * ld + r] : 0.0;
        acc[ci][ri][reg] = 0.0;
      }
  for (int kc0 = 0; kc0 < klen; kc0 += 32) {
    if (kc0) __syncthreads();
    for (int q = tid; q < 32 * 64; q += 256) {
      const int k = q >> 6, r = q & 63, kk = kc0 + k;
      P[k][r] = (kk < klen && i0 + r <= n) ? A[(size_t)(ks + kk) * ld + (i0 + r)] : 0.0;
      Q[k][r] = (kk < klen && j0 + r < col_end) ? A[(size_t)(ks + kk) * ld + (j0 + r)] : 0.0;
    }
    __syncthreads();
    mfma_tile_32x32(P, Q, 32, rbase, cbase, acc);
  }
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int c = j0 + cbase + 16 * ci + (lane >> 4) + 4 * reg, r = i0 + rbase + 16 * ri + (lane & 15);
        if (r <= n && c < col_end && r >= c) A[(size_t)c * ld + r] = dst[ci][ri][reg] - acc[ci][ri][reg];
      }
}

// The deferred panel update on 128 x 128 tiles (each wave a 64 x 64 sub-tile = 4 x 4 MFMA blocks), software-pipelined:
// the next 32-deep slice of both operands is fetched into registers while the current one is multiplied out of LDS.
// PMC counters place the 64 x 64 kernel at C5 on the L2 / Infinity-cache line-traffic limit (~3.7 TB/s x 4 flop/B = 15
// TFLOP/s: S does not fit the 32 MiB of L2); a 128 x 128 tile with K = 256 moves 2.7x fewer bytes per flop.
constexpr int kTS2 = 144;   // LDS row stride of a k-major 128-wide operand tile (rows k, k + 1 land 32 banks apart)
constexpr int kUpd128Lds = 2 * 32 * kTS2 * (int)sizeof(double);
__global__ __launch_bounds__(256, 2) void chol_update128_kernel(double* __restrict__ A, int n, int ld, int ks, int klen, int t0, int col_end) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double (*P)[kTS2] = reinterpret_cast<double (*)[kTS2]>(lds);
  double (*Q)[kTS2] = reinterpret_cast<double (*)[kTS2]>(lds + 32 * kTS2);
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (ti < tj) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int i0 = t0 + ti * 128, j0 = t0 + tj * 128;
  const int rbase = (wave & 1) * 64, cbase = (wave >> 1) * 64;
  d4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = d4_t{0.0, 0.0, 0.0, 0.0};
  double pr[16], qr[16];
  const int lr = tid & 127, lk0 = tid >> 7;   // this thread's row of the slice and its first k (k = lk0, lk0 + 2, ...)
  const bool pin = i0 + lr <= n, qin = j0 + lr < col_end;
  auto fetch = [&](int kc0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kk = kc0 + lk0 + 2 * i;
      const size_t colbase = (size_t)(ks + kk) * ld;
      pr[i] = (kk < klen && pin) ? A[colbase + (i0 + lr)] : 0.0;
      qr[i] = (kk < klen && qin) ? A[colbase + (j0 + lr)] : 0.0;
    }
  };
  fetch(0);
  for (int kc0 = 0; kc0 < klen; kc0 += 32) {
    if (kc0) __syncthreads();   // everyone is done reading the previous slice
#pragma unroll
    for (int i = 0; i < 16; ++i) { P[lk0 + 2 * i][lr] = pr[i]; Q[lk0 + 2 * i][lr] = qr[i]; }
    __syncthreads();
    if (kc0 + 32 < klen) fetch(kc0 + 32);   // in flight during the MFMA work below
    for (int k = 0; k < 32; k += 4) {
      double qv[4], pv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { qv[u] = Q[k + lk][cbase + 16 * u + li]; pv[u] = P[k + lk][rbase + 16 * u + li]; }
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) acc[ci][ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(qv[ci], pv[ri], acc[ci][ri], 0, 0, 0);
    }
  }
#pragma unroll
  for (int ci = 0; ci < 4; ++ci)
#pragma unroll
    for (int ri = 0; ri < 4; ++ri)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int c = j0 + cbase + 16 * ci + lk + 4 * reg, r = i0 + rbase + 16 * ri + li;
        if (r <= n && c < col_end && r >= c) A[(size_t)c * ld + r] -= acc[ci][ri][reg];
      }
}

// Back substitution, block step b0: z_b = L_bb^-T y_b (every workgroup, in LDS; workgroup 0 stores it), then
// y_c -= sum_r L(b0 + r, c) z_b[r] for this workgroup's 256 columns c < b0: four lanes per column, 16 contiguous rows
// (one cache line) each. The step is a chain of tiny dependent phases, so every global load it needs (y_b, the inverse
// block, the 64 x 256 slab of L, the y_c to update) is issued up front, before the first barrier: one memory round trip
// per launch instead of four. (A variant that fused four block steps per launch measured slower: its in-workgroup
// triangular solve added dependent round trips.)
__global__ __launch_bounds__(256) void chol_backsolve_step_kernel(double* __restrict__ A, int n, int ld, int b0, int kb,
                                                                  const double* __restrict__ linv_rm, double* __restrict__ z) {
  __shared__ double yb[64];
  __shared__ double zp[4][64];
  __shared__ double zb[64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int part = lane & 3, cq = (int)blockIdx.x * 256 + wave * 64 + (lane >> 2);
  const double yv = (tid < kb) ? A[(size_t)(b0 + tid) * ld + n] : 0.0;   // tid < kb <= 64
  double lv[16], cv[4][16], yc[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) lv[i] = linv_rm[(wave + 4 * i) * 64 + lane];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int c = cq + pass * 16;
    const bool on = c < b0;
    const double* __restrict__ colp = A + (size_t)(on ? c : 0) * ld + (b0 + part * 16);
#pragma unroll
    for (int k = 0; k < 16; ++k) cv[pass][k] = (on && part * 16 + k < kb) ? colp[k] : 0.0;
    yc[pass] = (on && part == 0) ? A[(size_t)c * ld + n] : 0.0;
  }
  if (tid < 64) yb[tid] = yv;
  __syncthreads();
  {  // z[c] = sum_r Linv[r][c] y[r]; wave w takes r = w, w + 4, ...
    double v = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += lv[i] * yb[wave + 4 * i];
    zp[wave][lane] = v;
  }
  __syncthreads();
  if (tid < 64) {
    const double v = (zp[0][tid] + zp[1][tid]) + (zp[2][tid] + zp[3][tid]);
    zb[tid] = v;
    if (blockIdx.x == 0 && tid < kb) z[b0 + tid] = v;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int c = cq + pass * 16;
    double v = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += cv[pass][k] * zb[part * 16 + k];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    if (c < b0 && part == 0) A[(size_t)c * ld + n] = yc[pass] - v;
  }
}

// ------------------------------------------------------------------------------------------------------
// back-substitution of the points, steps, model cost, candidate
// ------------------------------------------------------------------------------------------------------
// step_pt = -L^-T (h - sum Z z)  ( = -V^-1 (Es^T r - sum Y z), schur_eliminator_impl.h:303-366 )
__global__ __launch_bounds__(256) void ba_backsub_kernel(Dev d) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)d.N) d.step_cam[i] = d.cam_active[i] ? -d.zsol[i] : 0.0;   // step = -solution
  if (p >= d.n_pts || (d.pt_grouped && d.pt_grouped[p])) return;
  double t[3] = {d.hp[(size_t)p * 3], d.hp[(size_t)p * 3 + 1], d.hp[(size_t)p * 3 + 2]};
  for (uint32_t o = d.pt_start[p]; o < d.pt_start[p + 1]; ++o) {
    const uint32_t ip = d.opose[o];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double z = d.zsol[6 * ip + c];
#pragma unroll
      for (int e = 0; e < 3; ++e) t[e] -= d.Zpose[(size_t)o * 18 + e * 6 + c] * z;
    }
  }
  for (uint32_t s = d.ptk_start[p]; s < d.ptk_start[p + 1]; ++s) {
    const int col0 = 6 * (int)d.n_poses + 8 * (int)d.slot_intr[s];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const double z = d.zsol[col0 + c];
#pragma unroll
      for (int e = 0; e < 3; ++e) t[e] -= d.Zint[(size_t)s * 24 + e * 8 + c] * z;
    }
  }
  const double* li = d.Linv3 + (size_t)p * 6;
  d.step_pt[(size_t)p * 3 + 0] = -(li[0] * t[0] + li[1] * t[1] + li[3] * t[2]);
  d.step_pt[(size_t)p * 3 + 1] = -(li[2] * t[1] + li[4] * t[2]);
  d.step_pt[(size_t)p * 3 + 2] = -(li[5] * t[2]);
}

// model_cost_change = -(Js step)^T (r + Js step / 2)  (trust_region_minimizer.cc:402-405)
__global__ __launch_bounds__(256) void ba_model_cost_kernel(Dev d, double* __restrict__ part) {
  __shared__ double sh[4];
  const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double v = 0;
  if (o < d.n_obs) {
    const uint32_t ip = d.opose[o], ik = d.ointr[o], p = d.opt[o];
    double a[8], b[16], h[16];
    load_rec<8>(d.JA + (size_t)o * kJA, a);
    load_rec<16>(d.JB + (size_t)o * kJB, b);
    load_rec<16>(d.JC + (size_t)o * kJC, h);
    double m0 = 0, m1 = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double s = d.scale_cam[6 * ip + c] * d.step_cam[6 * ip + c];
      m0 += b[2 + c] * s; m1 += b[8 + c] * s;
    }
    const int col0 = 6 * (int)d.n_poses + 8 * (int)ik;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const double s = d.scale_cam[col0 + c] * d.step_cam[col0 + c];
      m0 += h[c] * s; m1 += h[8 + c] * s;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double s = d.scale_pt[(size_t)p * 3 + c] * d.step_pt[(size_t)p * 3 + c];
      m0 += a[2 + c] * s; m1 += a[5 + c] * s;
    }
    v = -(m0 * (a[0] + m0 / 2.0) + m1 * (a[1] + m1 / 2.0));
  }
  const double t = block_sum(v, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// The same quantity from the normal equations, without a pass over the observations (ba_step_scalars_kernel below): the step solves
// (Js^T Js + D^2) s = -gs exactly (direct solver), hence -(Js s)^T (r + Js s / 2) = -s^T gs - s^T Js^T Js s / 2 = (s^T D^2 s - s^T gs) / 2,
// with gs = scale o g the gradient and D^2 = LM diagonal / radius of the scaled problem; the point components are rank-local in a
// multi-rank run, the camera components replicated.
// the prior rows' share of the model cost change, added onto scalars[kSModel] (one workgroup)
__global__ __launch_bounds__(256) void ba_prior_model_kernel(Dev d) {
  __shared__ double sh[4];
  double v = 0;
  for (uint32_t q = threadIdx.x; q < d.n_priors; q += blockDim.x) {
    const double* jp = d.Jprior + (size_t)q * kPriorJ;
    const uint32_t ip = d.prior_pose[q];
    for (int k = 0; k < 3; ++k) {
      double m = 0;
      for (int c = 0; c < 6; ++c) m += jp[3 + k * 6 + c] * d.scale_cam[6 * ip + c] * d.step_cam[6 * ip + c];
      v -= m * (jp[k] + m / 2.0);
    }
  }
  const double t = block_sum(v, sh);
  if (threadIdx.x == 0) d.scalars[kSModel] += t;
}

// candidate_x = x + step o scaling; partial sums of |delta|^2 and |x|^2 (over the blocks of the reduced program)
__global__ __launch_bounds__(256) void ba_candidate_kernel(Dev d, double* __restrict__ part) {
  __shared__ double sh[4];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double dsq = 0, xsq = 0, cdsq = 0, cxsq = 0;
  if (i < (size_t)d.N) {
    const int np6 = 6 * (int)d.n_poses;
    double* x; double* cx; size_t idx;
    if ((int)i < np6) { x = d.poses; cx = d.cposes; idx = i; } else { x = d.intr; cx = d.cintr; idx = i - np6; }
    const double delta = d.step_cam[i] * d.scale_cam[i];
    cx[idx] = x[idx] + delta;
    cdsq = delta * delta;
    if (d.cam_counts[i]) cxsq = x[idx] * x[idx];
  }
  if (i < (size_t)d.n_pts * 3) {
    const double delta = d.step_pt[i] * d.scale_pt[i];
    d.cpts[i] = d.pts[i] + delta;
    dsq += delta * delta;
    if (d.scale_pt[i] != 0.0) xsq += d.pts[i] * d.pts[i];
  }
  const double ca = block_sum(cdsq, sh);
  const double cb = block_sum(cxsq, sh);
  const double a = block_sum(dsq, sh);
  const double b = block_sum(xsq, sh);
  if (threadIdx.x == 0) {
    part[4 * blockIdx.x] = ca; part[4 * blockIdx.x + 1] = cb; part[4 * blockIdx.x + 2] = a; part[4 * blockIdx.x + 3] = b;
  }
}

// ba_model_cost_vec_kernel and ba_candidate_kernel in one pass over the step (both read it once; as two launches with a reduction each
// they were four launches): part[6 b + 0..3] = the candidate's sums, [4] / [5] = the point / camera parts of the model cost change
__global__ __launch_bounds__(256) void ba_step_scalars_kernel(Dev d, double inv_radius, double* __restrict__ part) {
  __shared__ double sh[4];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double dsq = 0, xsq = 0, cdsq = 0, cxsq = 0, vc = 0, vp = 0;
  if (i < (size_t)d.N) {
    const int np6 = 6 * (int)d.n_poses;
    double* x; double* cx; size_t idx;
    if ((int)i < np6) { x = d.poses; cx = d.cposes; idx = i; } else { x = d.intr; cx = d.cintr; idx = i - np6; }
    const double s = d.step_cam[i];
    const double delta = s * d.scale_cam[i];
    cx[idx] = x[idx] + delta;
    cdsq = delta * delta;
    if (d.cam_counts[i]) cxsq = x[idx] * x[idx];
    vc = 0.5 * s * (s * d.diag_cam[i] * inv_radius - d.g_cam[i] * d.scale_cam[i]);
  }
  if (i < (size_t)d.n_pts * 3) {
    const double s = d.step_pt[i];
    const double delta = s * d.scale_pt[i];
    d.cpts[i] = d.pts[i] + delta;
    dsq += delta * delta;
    if (d.scale_pt[i] != 0.0) xsq += d.pts[i] * d.pts[i];
    vp = 0.5 * s * (s * d.diag_pt[i] * inv_radius - d.g_pt[i] * d.scale_pt[i]);
  }
  const double ca = block_sum(cdsq, sh);
  const double cb = block_sum(cxsq, sh);
  const double a = block_sum(dsq, sh);
  const double b = block_sum(xsq, sh);
  const double tp = block_sum(vp, sh);
  const double tc = block_sum(vc, sh);
  if (threadIdx.x == 0) {
    double* o = part + 6 * (size_t)blockIdx.x;
    o[0] = ca; o[1] = cb; o[2] = a; o[3] = b; o[4] = tp; o[5] = tc;
  }
}

// The camera half of ba_step_scalars_kernel alone (the candidate's poses and intrinsics, |delta|^2, |x|^2 and the model cost change of
// the camera columns: part[3 b + 0..2]) - run BEFORE the back-substitution of the point groups when that pass forms the points'
// half and the candidate's cost itself (it reads the candidate's cameras).
// gather: sp_gather_solution_kernel's work first (the solution of the block-sparse solve at its column -> zsol, step_cam), same thread.
__global__ __launch_bounds__(256) void ba_step_scalars_cam_kernel(Dev d, double inv_radius, double* __restrict__ part, int gather) {
  __shared__ double sh[4];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double cdsq = 0, cxsq = 0, vc = 0;
  if (i < (size_t)d.N) {
    const int np6 = 6 * (int)d.n_poses;
    double* x; double* cx; size_t idx;
    if ((int)i < np6) { x = d.poses; cx = d.cposes; idx = i; } else { x = d.intr; cx = d.cintr; idx = i - np6; }
    double s;
    if (gather) {
      const double z = d.sp.z[d.sp.pcol[i]];
      d.zsol[i] = z;
      s = d.cam_active[i] ? -z : 0.0;
      d.step_cam[i] = s;
    } else s = d.step_cam[i];
    const double delta = s * d.scale_cam[i];
    cx[idx] = x[idx] + delta;
    cdsq = delta * delta;
    if (d.cam_counts[i]) cxsq = x[idx] * x[idx];
    vc = 0.5 * s * (s * d.diag_cam[i] * inv_radius - d.g_cam[i] * d.scale_cam[i]);
  }
  const double ca = block_sum(cdsq, sh);
  const double cb = block_sum(cxsq, sh);
  const double tc = block_sum(vc, sh);
  if (threadIdx.x == 0) { double* o = part + 3 * (size_t)blockIdx.x; o[0] = ca; o[1] = cb; o[2] = tc; }
}
// The eight sums of a step whose back-substitution formed the candidate: one workgroup per sum (all at once instead of one after the
// other), rows in index order as reduce_partials_kernel takes them. cam_part: n_cam x 3 (ba_step_scalars_cam_kernel), grp_part: n_sg x 5.
// (Publishing the scalars to the host from the workgroup that finishes last - arrival counter, fences, the stores to host memory -
// was measured: this kernel 4.7 -> 11.9 us against the 3.9 us of ba_publish_scalars_kernel's own launch. Not kept.)
__global__ __launch_bounds__(1024) void ba_step_reduce_kernel(const double* __restrict__ cam_part, int n_cam, const double* __restrict__ grp_part, int n_sg,
                                                              double* __restrict__ scalars) {
  __shared__ double sh[16];
  // sum -> (source, column, scalar)
  const int k = blockIdx.x;
  const bool cam = k >= 5;
  const int col = cam ? k - 5 : k;
  const int dst = cam ? (col == 0 ? kSCamStepSq : col == 1 ? kSCamXSq : kSModelCam)
                      : (col == 0 ? kSCost : col == 1 ? kSSqErr : col == 2 ? kSStepSq : col == 3 ? kSXSq : kSModelPt);
  const double* __restrict__ part = cam ? cam_part : grp_part;
  const int n = cam ? n_cam : n_sg, stride = cam ? 3 : 5;
  double v = 0;
  int i = (int)threadIdx.x;
  for (; i + 7 * 1024 < n; i += 8 * 1024) {
    double t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = part[(size_t)(i + q * 1024) * stride + col];
#pragma unroll
    for (int q = 0; q < 8; ++q) v += t[q];
  }
  for (; i < n; i += 1024) v += part[(size_t)i * stride + col];
  const double t = block_sum(v, sh);
  if (threadIdx.x == 0) scalars[dst] = t;
}

// ---- host side of the assembly: product lists sorted by destination block ----
struct TripHost {
  std::vector<uint2> trips;
  std::vector<uint32_t> chunk_lo, chunk_hi, block_row, block_col, block_chunk0;
  std::vector<uint8_t> chunk_diag;
  std::vector<int32_t> block_own;
  // external partial blocks (point groups): per block the range of ext rows, numbered in block order
  std::vector<uint32_t> block_ext0;
  uint32_t n_ext = 0;
};
// external partial blocks handed to build_trip_list: per row block a list of (column block, ext id), ascending; the list
// build assigns every ext id its row in the partial-sum buffer (ext_row[id], counted after the flat chunks)
struct TripExt {
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> rows;
  std::vector<uint32_t> ext_row;
};

// Host threads for the structure build (the analogue of Ceres' preprocessor). f(index, thread) is called for every index
// in [0, n), indices handed out dynamically; results must not depend on which thread ran an index.
inline unsigned host_threads(size_t work_items) {
  if (const char* env = getenv("MVGX_HOST_THREADS")) return (unsigned)std::min(64, std::max(1, atoi(env)));   // as told
  // one thread per ~16k items: waking a thread costs microseconds, and the small problems an SfM engine sends most often
  // (initial pair, local adjustments) are built faster by the calling thread alone
  // (32 at most: measured on the 2 x 64-core bench host, 5M observations: 16 threads 33 ms, 32 threads 21.5 ms, 64 threads 21.8 ms
  // typical but one call in three at 45 - 70 ms)
  const unsigned want = (unsigned)std::min<size_t>(work_items / 16384, 32);
  return std::max(1u, std::min(std::thread::hardware_concurrency(), want));
}

// The workers behind parallel_for_dynamic: started once per process (a structure build runs ~30 parallel loops; starting and
// joining 31 threads for each of them cost about a millisecond per loop), parked on a condition variable between jobs. One job
// at a time: a caller that finds the pool busy (several contexts being created at once) runs its loop on threads of its own.
class HostPool {
 public:
  static HostPool* instance() {
    static std::once_flag once;
    std::call_once(once, [] {
      pthread_atfork(nullptr, nullptr, [] { g_pool_ = nullptr; });   // a forked child has no workers: it starts its own on first use
    });
    HostPool* p = g_pool_.load(std::memory_order_acquire);
    if (p) return p;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    p = g_pool_.load(std::memory_order_acquire);
    if (!p) { p = new HostPool(); g_pool_.store(p, std::memory_order_release); }   // never destroyed (workers may outlive main's statics)
    return p;
  }
  // runs body(t) for t in [0, threads): t = 0 on the calling thread; false when the pool is busy (the caller then uses its own threads)
  bool run(unsigned threads, const std::function<void(unsigned)>& body) {
    std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
    if (!job.owns_lock()) return false;
    const unsigned helpers = std::min<unsigned>(threads - 1, (unsigned)workers_.size());
    {
      std::lock_guard<std::mutex> lk(mu_);   // the job and its epoch change together (workers read them under the lock); a worker about to sleep cannot miss it
      body_ = &body; want_ = helpers;
      remaining_.store(helpers, std::memory_order_relaxed);
      epoch_.fetch_add(1, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire)) cv_start_.notify_all();
    body(0);
    for (int i = 0; i < kSpin && remaining_.load(std::memory_order_acquire); ++i) cpu_relax();
    if (remaining_.load(std::memory_order_acquire)) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_done_.wait(lk, [&] { return remaining_.load(std::memory_order_acquire) == 0; });
    }
    return true;
  }
 private:
  // A structure build is ~30 loops a few hundred microseconds apart: a worker that went to sleep on the condition variable
  // between two of them cost 50 - 100 us to wake (x 30 loops: half of a 100k-observation build), so workers poll the epoch for
  // about that long before they sleep.
  static constexpr int kSpin = 1 << 15;
  static void cpu_relax() { __builtin_ia32_pause(); }
  HostPool() {
    const unsigned n = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u)) - 1;   // (host_threads() asks for at most 32, MVGX_HOST_THREADS for at most 64: beyond the pool the caller's own threads)
    for (unsigned t = 0; t < n; ++t) workers_.emplace_back([this, t] { work(t + 1); }), workers_.back().detach();
  }
  void work(unsigned tix) {
    uint64_t seen = 0;
    bool active = false;   // took part in the last job: only those poll (a build that uses 6 threads must not keep 31 polling)
    for (;;) {
      bool got = false;
      for (int i = 0; active && i < kSpin; ++i) {
        if (epoch_.load(std::memory_order_acquire) != seen) { got = true; break; }
        cpu_relax();
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1, std::memory_order_acq_rel);
        cv_start_.wait(lk, [&] { return epoch_.load(std::memory_order_acquire) != seen; });
        sleepers_.fetch_sub(1, std::memory_order_acq_rel);
      }
      const std::function<void(unsigned)>* body;
      unsigned want;
      { std::lock_guard<std::mutex> lk(mu_); seen = epoch_.load(std::memory_order_relaxed); body = body_; want = want_; }
      active = tix <= want;
      if (!active) continue;   // (a job cannot end - and the next one begin - before every worker it counts on has run it)
      (*body)(tix);
      if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> lk(mu_);
        cv_done_.notify_one();
      }
    }
  }
  static std::atomic<HostPool*> g_pool_;
  std::vector<std::thread> workers_;
  std::mutex job_mu_, mu_;
  std::condition_variable cv_start_, cv_done_;
  const std::function<void(unsigned)>* body_ = nullptr;
  unsigned want_ = 0;
  std::atomic<unsigned> remaining_{0}, sleepers_{0};
  std::atomic<uint64_t> epoch_{0};
};
std::atomic<HostPool*> HostPool::g_pool_{nullptr};

template <class F>
void parallel_for_dynamic(size_t n, size_t grain, unsigned threads, F f) {
  if (threads <= 1 || n <= grain) { for (size_t i = 0; i < n; ++i) f(i, 0u); return; }
  threads = (unsigned)std::min<size_t>(threads, (n + grain - 1) / grain);
  std::atomic<size_t> next{0};
  auto body = [&](unsigned tix) {
    for (;;) {
      const size_t lo = next.fetch_add(grain);
      if (lo >= n) return;
      const size_t hi = std::min(n, lo + grain);
      for (size_t i = lo; i < hi; ++i) f(i, tix);
    }
  };
  if (HostPool::instance()->run(threads, body)) return;
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < threads; ++t) pool.emplace_back(body, t);
  body(0);
  for (auto& th : pool) th.join();
}

// Stable counting sort of the indices 0..n-1 by key(i) in [0, n_keys): start[n_keys + 1] and order[n] (ascending index inside a
// key), with per-thread histograms over contiguous index ranges; the result does not depend on the thread count.
template <class Key>
void counting_sort_indices(uint64_t n, uint32_t n_keys, unsigned threads, Key key, std::vector<uint32_t>& start, uint32_t* order) {
  start.assign((size_t)n_keys + 1, 0);
  if (threads <= 1 || n < 4096 || (uint64_t)threads * n_keys > (1u << 24)) {
    for (uint64_t i = 0; i < n; ++i) start[key(i) + 1]++;
    for (uint32_t k = 0; k < n_keys; ++k) start[k + 1] += start[k];
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    for (uint64_t i = 0; i < n; ++i) order[fill[key(i)]++] = (uint32_t)i;
    return;
  }
  const uint64_t per = (n + threads - 1) / threads;
  std::vector<std::vector<uint32_t>> hist(threads);
  parallel_for_dynamic(threads, 1, threads, [&](size_t t, unsigned) {
    hist[t].assign(n_keys, 0);
    for (uint64_t i = t * per, e = std::min(n, (t + 1) * per); i < e; ++i) hist[t][key(i)]++;
  });
  uint32_t run = 0;
  for (uint32_t k = 0; k < n_keys; ++k) {
    start[k] = run;
    for (unsigned t = 0; t < threads; ++t) { const uint32_t c_ = hist[t][k]; hist[t][k] = run; run += c_; }
  }
  start[n_keys] = run;
  parallel_for_dynamic(threads, 1, threads, [&](size_t t, unsigned) {
    for (uint64_t i = t * per, e = std::min(n, (t + 1) * per); i < e; ++i) order[hist[t][key(i)]++] = (uint32_t)i;
  });
}
template <class Key>
void counting_sort_indices(uint64_t n, uint32_t n_keys, unsigned threads, Key key, std::vector<uint32_t>& start, std::vector<uint32_t>& order) {
  order.resize(n);
  counting_sort_indices(n, n_keys, threads, key, start, order.data());
}

// Product list of one family, generated row by row (row = camera block of the first factor). gen(r, emit) must call
// emit(col, a, b) for every product of row r in the canonical order (point, then first factor, then second factor); the
// products of a row are grouped by column with a stable counting sort, cut into blocks (one per (row, col)) and chunks of
// at most kTripChunk products. Rows are independent: they are generated by host threads and stitched in row order, so the
// list does not depend on the thread count. O(products) time, no comparison sort over the products.
template <class Gen>
int build_trip_list(size_t n_cb, Gen gen, TripHost& out, unsigned threads, TripExt* ext = nullptr) {
  struct RowMeta { std::vector<uint32_t> col, cnt; };
  std::vector<uint64_t> rstart(n_cb + 1, 0);
  parallel_for_dynamic(n_cb, 1, threads, [&](size_t r, unsigned) {
    uint64_t n = 0;
    gen((uint32_t)r, [&](uint32_t, uint32_t, uint32_t) { ++n; });
    rstart[r + 1] = n;
  });
  for (size_t r = 0; r < n_cb; ++r) rstart[r + 1] += rstart[r];
  MVGX_REQUIRE(rstart[n_cb] < (1ull << 32), MVGX_ERR_ARG, "mvgx_ba_create: too many co-visibility products for one device shard");
  out.trips.resize(rstart[n_cb]);
  std::vector<RowMeta> meta(n_cb);
  struct Scratch { std::vector<uint32_t> cnt, touched, cols; std::vector<uint64_t> off; std::vector<uint2> ab; };
  std::vector<Scratch> scratch(threads);
  for (auto& sc : scratch) { sc.cnt.assign(n_cb, 0); sc.off.assign(n_cb, 0); }
  parallel_for_dynamic(n_cb, 1, threads, [&](size_t r, unsigned tix) {
    const uint64_t lo = rstart[r], hi = rstart[r + 1];
    if (lo == hi) return;
    Scratch& sc = scratch[tix];
    sc.touched.clear(); sc.cols.clear(); sc.ab.clear();
    gen((uint32_t)r, [&](uint32_t cb, uint32_t a, uint32_t b) {
      if (sc.cnt[cb]++ == 0) sc.touched.push_back(cb);
      sc.cols.push_back(cb); sc.ab.push_back(make_uint2(a, b));
    });
    std::sort(sc.touched.begin(), sc.touched.end());
    uint64_t pos = lo;
    RowMeta& m = meta[r];
    for (uint32_t cb : sc.touched) { sc.off[cb] = pos; m.col.push_back(cb); m.cnt.push_back(sc.cnt[cb]); pos += sc.cnt[cb]; sc.cnt[cb] = 0; }
    for (size_t q = 0; q < sc.cols.size(); ++q) out.trips[sc.off[sc.cols[q]]++] = sc.ab[q];
  });
  out.block_chunk0.push_back(0);
  if (ext) out.block_ext0.push_back(0);
  for (size_t r = 0; r < n_cb; ++r) {
    uint64_t pos = rstart[r];
    const RowMeta& m = meta[r];
    static const std::vector<std::pair<uint32_t, uint32_t>> none;
    const auto& xr = ext ? ext->rows[r] : none;   // ascending (column, id)
    size_t k = 0, x = 0;
    while (k < m.col.size() || x < xr.size()) {   // union of the flat list's columns and the external ones, ascending
      const uint32_t cb = std::min(k < m.col.size() ? m.col[k] : UINT32_MAX, x < xr.size() ? xr[x].first : UINT32_MAX);
      out.block_row.push_back((uint32_t)r); out.block_col.push_back(cb); out.block_own.push_back(-1);
      if (k < m.col.size() && m.col[k] == cb) {
        const uint32_t cnt = m.cnt[k];
        for (uint64_t a = pos; a < pos + cnt; a += kTripChunk) {
          out.chunk_lo.push_back((uint32_t)a);
          out.chunk_hi.push_back((uint32_t)std::min<uint64_t>(a + kTripChunk, pos + cnt));
          out.chunk_diag.push_back(cb == (uint32_t)r ? 1 : 0);
        }
        pos += cnt;
        ++k;
      }
      out.block_chunk0.push_back((uint32_t)out.chunk_lo.size());
      if (ext) {
        for (; x < xr.size() && xr[x].first == cb; ++x) ext->ext_row[xr[x].second] = out.n_ext++;
        out.block_ext0.push_back(out.n_ext);
      }
    }
  }
  return MVGX_OK;
}

template <typename T>
int dev_alloc(mvgx::Arena& pool, T** p, size_t n) {
  return pool.alloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
}
template <typename T>
int dev_upload_n(mvgx::Arena& pool, T** p, const T* v, size_t n, hipStream_t s) {
  int rc = dev_alloc(pool, p, n);
  if (rc) return rc;
  if (n) MVGX_HIP(hipMemcpyAsync(*p, v, n * sizeof(T), hipMemcpyHostToDevice, s));
  return MVGX_OK;
}
template <typename T>
int dev_upload(mvgx::Arena& pool, T** p, const std::vector<T>& v, hipStream_t s) {
  int rc = dev_alloc(pool, p, v.size());
  if (rc) return rc;
  if (!v.empty()) MVGX_HIP(hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return MVGX_OK;
}

}  // namespace

struct mvgx_ba_ctx {
  mvgx::BaMulti* multi = nullptr;      // a context over several devices (mvgx_ba_create_multi / MVGX_DEVICES): everything below is unused
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  Dev d;
  mvgx::Arena pool;                    // every device allocation (slabs go back to the process-wide cache in destroy)
  double* h_scalars = nullptr;         // pinned: kSCount scalars | fail word | sequence number of the last publication
  double* h_scalars_dev = nullptr;     // the same block as the device addresses it
  unsigned long long publish_seq = 0;
  bool poll_scalars = true;
  int* h_fail = nullptr;               // the slot after h_scalars
  int* d_accept = nullptr;             // device word: the accept decision of the step just computed (ba_publish_scalars_kernel)
  bool speculate = true;               // MVGX_BA_SPECULATE=0: the next Jacobian evaluation is launched after the host's decision, as before round 6
  bool spec_done = false;              // this step's Jacobian evaluation at x + delta is already enqueued, gated by d_accept
  mvgx_allreduce_f64 allreduce = nullptr;
  void* allreduce_user = nullptr;
  mvgx::RcclComm* rccl = nullptr;
  double n_obs_rmse_local = 0;         // observations of this rank that count in the RMSE (not control points)
  double n_obs_global = 0;             // ... over all ranks
  bool update_incomplete = false;      // a mvgx_ba_update{,_subset} failed half way: no solve until one succeeds
  // LM state (persists across mvgx_ba_lm_iteration calls)
  bool started = false;
  double x_cost = 0, radius = 0, decrease_factor = 2.0, gradient_max_norm = 0;
  double dmin = 1e-6, dmax = 1e32;     // LM diagonal clamp of the options the current Jacobian evaluation ran with
  bool gmax_pending = false;           // max |gradient| of the last Jacobian evaluation is still on the device
  bool gmax_resolved = false;          // ... and arrived with this iteration's step
  bool reuse_diagonal = false, x_norm_valid = false, last_successful = true;
  int iteration = 0, invalid = 0, successful = 0, termination = 1;
  bool finished = false;
  double initial_cost = 0, initial_rmse = 0;
  int grid_obs = 0, grid_vec = 0;
  // reduced-system solver: block-sparse (tile) Cholesky with a nested-dissection order, or the dense one (auto: by fill)
  std::vector<std::pair<uint32_t, uint32_t>> h_blocks;   // non-zero camera blocks of this rank's S (row block, col block)
  uint32_t n_grouped_points = 0;
  bool model_cost_from_jacobian = false;   // MVGX_BA_MODEL_COST=jacobian: ba_model_cost_kernel instead of the normal-equation form
  bool solver_ready = false;
  bool pinhole_family = false;       // every intrinsic is pinhole / radial K1 / radial K3 / Brown T2: the point-group kernels run without the spherical / fisheye branches
  bool diag_blocks_complete = false;   // every pose / intrinsic block has a diagonal destination block in the assemble lists
  uint32_t n_sg_wide = 0;             // supergroups of the wide form: the tail of sg_order, a launch of their own
  std::vector<uint32_t> h_sg_order;   // (a member: the asynchronous upload may read it after mvgx_ba_create's locals are gone)
  // what mvgx_ba_update needs to re-bind the context to new values of the same structure
  mvgx::BaFingerprint fingerprint;            // of the problem the context was created from
  std::vector<uint8_t> h_pose_used, h_intr_used;   // blocks that carry a residual (the camera masks are applied on top of these)
  std::vector<uint32_t> h_perm;               // caller's observation index of point-order position k; empty: the caller's list was in point order
  uint64_t n_gentries = 0;                    // entries of the point groups (their copy of the image points is re-gathered)
  std::vector<uint8_t> h_pt_free;             // which points are free parameter blocks with residuals (structure)
  double n_obs_rmse_all = 0;                  // n_obs_rmse_local of the whole structure (a subset counts its own)
  uint8_t* odisabled_buf = nullptr;           // device array behind Dev::odisabled (allocated by the first mvgx_ba_update_subset)
  double* filter_scratch = nullptr;           // 3 n_obs doubles: residual norms / observation rays (mvgx_ba_residuals, mvgx_ba_track_angles)
  // block-sparse solve, MVGX_BA_LOOKAHEAD=1: one launch per level, the next level's factorisation beside the tasks of this one. Built and measured
  // in round 5 (calls r5_06 .. r5_10), bit-identical, and 7 % SLOWER than the level-by-level schedule (C5 0.445 against 0.416 ms, C3 0.266 against
  // 0.249): taking a contribution inside the factor workgroup costs the 6 us the separate task launch did, and the U tasks that rebuild their
  // strips of L do six times the MFMA work at a quarter of the occupancy. Off by default; kept as the cross-check of the schedule (tests).
  bool lookahead = false;
  // reverse sweep of the block-sparse solve in ONE launch, columns handed over through device-scope flags (MVGX_BA_BACKSOLVE_FLAGS=1). Built and
  // measured in round 5 (call r5_16): bit-identical, and SLOWER - C5 solve 0.474 against 0.409 ms, C3 0.246 against 0.245: a device-scope release on
  // this eight-XCD part writes the producer's whole L2 back, a hand-over costs ~15 us where a launch boundary costs 10. Off by default.
  bool bs_one_launch = true;    // the reverse sweep of the block-sparse solver as ONE launch (round 6: values polled, no flags - sp_backsolve_all_kernel); MVGX_BA_BACKSOLVE_FLAGS=0: a launch per level
  unsigned bs_epoch = 0;       // number of the solve whose solution the flags of that launch announce
  int chain_fuse_max_tasks = 4096;   // single-column levels with at most this many update tasks run panel + update as one launch (MVGX_BA_CHAIN_FUSE=0: never)
  bool fold_cand_now = false;   // this step: set by compute_step before the solve
  bool fold_candidate = true, candidate_cost_done = false;   // the candidate and its cost from the back-substitution pass of the point groups (compute_step)
  bool all_points_grouped = false;   // every point is in a group: the per-point kernels of the record path have nothing to do
  bool fail_clear = false;           // the device's fail word was cleared by the last Jacobian evaluation and nothing has run since that can set it
  int bs_chain_levels = 0;   // the top levels of the elimination tree with one tile column each (sp_backsolve_chain_kernel), 0: none
  bool plan_ready = false, plan_sparse = false;   // symbolic phase of the reduced solve done (mvgx_ba_create; again at the first iteration when a communicator was attached since)
  double x_sqerr = 0;   // sum of squared residuals at x (the RMSE's numerator), kept with x_cost
  int solver_mode = 0;             // MVGX_BA_SOLVER / mvgx_ba_set_linear_solver: 0 auto, 1 dense, 2 sparse, 3 sparse when its plan exists
  bool solver_from_env = false;    // MVGX_BA_SOLVER names the solver: mvgx_ba_set_linear_solver leaves it alone
  bool plan_tried = false, plan_ok = false;   // the last plan_solver attempted the sparse plan / it exists
  mvgx_sparse::Plan plan;          // host copy of the schedule (launch geometry per level)
  int update128_min_tiles = 128;   // tuning (MVGX_BA_UPDATE128_MIN_TILES): deferred updates with at least this many 128 x 128 tiles use them
  int two_level_min_n = 2048;   // tuning (MVGX_BA_TWO_LEVEL_MIN_N): reduced systems at least this wide factor with 256-column outer panels
  // phase timing (MVGX_BA_PHASE_TIMING=1 at create): HIP events around the five phases of an iteration, summed into phase_ms
  bool phase_timing = false;
  struct Mark { int id; hipEvent_t a, b; };
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<Mark> marks;
  hipEvent_t ph_a = nullptr;
  double phase_ms[5] = {0, 0, 0, 0, 0};   // jacobian, schur, solve, backsub, cost
};

namespace {

#define BA_LAUNCH_CHECK() MVGX_HIP(hipGetLastError())

// The scalars of a step reach the host through its page-locked, device-visible block: a one-wave kernel copies them there and then
// raises a sequence number the host polls - the copy engine + hipStreamSynchronize pair it replaces woke the host 10 - 15 us after
// the stream had drained (an LM iteration waits for exactly one such hand-over). MVGX_BA_POLL_SCALARS=0: the copy + synchronise form.
// Round 6: the kernel also takes the step's accept decision ITSELF (x_cost: the cost at x, known to the host when it launches the step; the
// test of trust_region_minimizer.cc:640-700 on the same doubles with the same three operations) and leaves it in `accept` - the Jacobian
// evaluation of the next iteration is launched right behind this kernel, before the host has seen the scalars, gated by that word
// (Dev::gate): the 20-25 us between the last kernel of an iteration and the first of the next - the host's poll, decision and launch -
// were 5 % of an iteration at C3. The host reads the same word with the scalars and follows it.
__global__ __launch_bounds__(64) void ba_publish_scalars_kernel(const double* __restrict__ scalars, double* __restrict__ host, unsigned long long seq,
                                                                int* __restrict__ accept, double x_cost, double min_relative_decrease, int decide,
                                                                double parameter_tolerance, double function_tolerance, int x_norm_valid) {
  const int t = threadIdx.x;
  if (t <= kSCount) host[t] = scalars[t];   // (the fail word rides in slot kSCount)
  if (t == 0 && decide) {
    const double model = scalars[kSModelPt] + scalars[kSModelCam];
    const int failed = *reinterpret_cast<const int*>(scalars + kSCount);
    const bool ok = !failed && isfinite(model);
    const double relative_decrease = (x_cost - scalars[kSCost]) / model;
    const int a = (ok && model > 0.0 && relative_decrease > min_relative_decrease) ? 1 : 0;
    // ... and the two tolerance tests that END the solve before the step is taken (lm_iteration): a solve of three iterations should not pay
    // for an evaluation nobody uses. The gate is the conjunction; the host follows `a` and checks it, and launches the evaluation itself
    // whenever the gate stayed shut.
    const double step_norm = sqrt(scalars[kSCamStepSq] + scalars[kSStepSq]);
    const double x_norm = x_norm_valid ? sqrt(scalars[kSCamXSq] + scalars[kSXSq]) : -1.0;
    const bool ends = step_norm <= parameter_tolerance * (x_norm + parameter_tolerance) || fabs(x_cost - scalars[kSCost]) <= function_tolerance * x_cost;
    const int g = a && !ends ? 1 : 0;
    *accept = g;
    host[kSCount + 2] = (double)(a + 2 * g);
  }
  __threadfence_system();
  __syncthreads();
  if (t == 0) *reinterpret_cast<volatile unsigned long long*>(host + kSCount + 1) = seq;
}
int read_scalars(mvgx_ba_ctx* c, int decide = 0, const mvgx_ba_options* opt = nullptr, int (*between)(mvgx_ba_ctx*, void*) = nullptr, void* between_arg = nullptr) {
  if (!c->poll_scalars) {
    MVGX_HIP(hipMemcpyAsync(c->h_scalars, c->d.scalars, (kSCount + 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MVGX_HIP(hipStreamSynchronize(c->stream));
    return MVGX_OK;
  }
  const unsigned long long seq = ++c->publish_seq;
  hipLaunchKernelGGL(ba_publish_scalars_kernel, dim3(1), dim3(64), 0, c->stream, c->d.scalars, c->h_scalars_dev, seq, c->d_accept, c->x_cost, opt ? opt->min_relative_decrease : 0.0, decide,
                     opt ? opt->parameter_tolerance : 0.0, opt ? opt->function_tolerance : 0.0, c->x_norm_valid ? 1 : 0);
  BA_LAUNCH_CHECK();
  if (between) { const int rc_b = between(c, between_arg); if (rc_b) return rc_b; }   // (work enqueued behind the publish kernel while the device still runs the step)
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->h_scalars + kSCount + 1);
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spins = 0; *flag != seq; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 0xFFFFF) == 0xFFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {   // a failed launch / device fault: ask the runtime
      MVGX_HIP(hipStreamSynchronize(c->stream));
      MVGX_REQUIRE(*flag == seq, MVGX_ERR_HIP, "the scalars of the step did not reach the host");
      break;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return MVGX_OK;
}

bool multi_rank(const mvgx_ba_ctx* c) { return c->rccl != nullptr || c->allreduce != nullptr; }

enum { kPhJacobian = 0, kPhSchur, kPhSolve, kPhBacksub, kPhCost };
hipEvent_t phase_event(mvgx_ba_ctx* c) {
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    c->ev_pool.push_back(e);
  }
  hipEvent_t e = c->ev_pool[c->ev_used++];
  (void)hipEventRecord(e, c->stream);
  return e;
}
void phase_begin(mvgx_ba_ctx* c) { if (c->phase_timing) c->ph_a = phase_event(c); }
void phase_end(mvgx_ba_ctx* c, int id) {
  if (!c->phase_timing || !c->ph_a) return;
  hipEvent_t b = phase_event(c);
  if (b) c->marks.push_back({id, c->ph_a, b});
  c->ph_a = nullptr;
}
void phase_collect(mvgx_ba_ctx* c) {   // after a stream synchronisation
  for (const auto& m : c->marks) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, m.a, m.b) == hipSuccess) c->phase_ms[m.id] += ms;
  }
  c->marks.clear();
  c->ev_used = 0;
}

int all_reduce(mvgx_ba_ctx* c, double* buf, uint64_t count, int op = MVGX_REDUCE_SUM) {
  if (c->rccl) return mvgx::rccl_allreduce_f64(c->rccl, buf, count, op, c->stream);
  if (!c->allreduce) return MVGX_OK;
  const int rc = c->allreduce(c->allreduce_user, buf, count, op, c->stream);
  MVGX_REQUIRE(rc == 0, MVGX_ERR_HIP, "all-reduce callback failed (%d)", rc);
  return MVGX_OK;
}

// cost (+ Jacobian) at the given parameters; scalars[kSCost], [kSSqErr]
template <bool kJac>
int eval(mvgx_ba_ctx* c, const double* poses, const double* intr, const double* pts) {
  Dev& d = c->d;
  if (d.n_obs)
    hipLaunchKernelGGL(ba_linearize_kernel<kJac>, dim3(c->grid_obs), dim3(256), 0, c->stream, d, poses, intr, pts, d.part);
  BA_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, c->stream, d.part, d.n_obs ? c->grid_obs : 0, 2, 2, d.scalars,
                     kSCost, 0);
  if (d.n_priors) hipLaunchKernelGGL(ba_prior_kernel<kJac>, dim3(1), dim3(256), 0, c->stream, d, poses, 1);
  BA_LAUNCH_CHECK();
  return all_reduce(c, d.scalars + kSCost, 2);
}

// launch of the fused point-group pass in one of its three modes
template <int MODE>
void launch_point_groups(mvgx_ba_ctx* c, double inv_radius, double dmin, double dmax, double* cand_part = nullptr) {
  Dev& d = c->d;
  double* const ppp = d.tpp.part + (size_t)d.tpp.n_chunks * kNVpp;
  double* const ppi = d.tpi.part + (size_t)d.tpi.n_chunks * kNVpi;
  double* const pii = d.tii.part + (size_t)d.tii.n_chunks * kNVii;
  // the supergroups of the usual form, then - a launch of its own - those of the wide form (G.sg_order lists them in that order)
  const uint32_t n_narrow = d.grp.n_sg - c->n_sg_wide, n_wide = c->n_sg_wide;
  if (c->pinhole_family) {   // every intrinsic is a polynomial model: the variant without the spherical / fisheye branches
    if (n_narrow) hipLaunchKernelGGL((ba_point_group_kernel<MODE, true, false>), dim3(n_narrow), dim3(kGroupThreads), group_lds_bytes<MODE>(), c->stream, d, d.grp, inv_radius, dmin, dmax, ppp, ppi, pii, cand_part, 0u, 1);
    if (n_wide) hipLaunchKernelGGL((ba_point_group_kernel<MODE, true, true>), dim3(n_wide), dim3(kGroupThreads), group_lds_bytes<MODE>(), c->stream, d, d.grp, inv_radius, dmin, dmax, ppp, ppi, pii, cand_part, n_narrow, n_narrow ? 0 : 1);
  } else {
    if (n_narrow) hipLaunchKernelGGL((ba_point_group_kernel<MODE, false, false>), dim3(n_narrow), dim3(kGroupThreads), group_lds_bytes<MODE>(), c->stream, d, d.grp, inv_radius, dmin, dmax, ppp, ppi, pii, cand_part, 0u, 1);
    if (n_wide) hipLaunchKernelGGL((ba_point_group_kernel<MODE, false, true>), dim3(n_wide), dim3(kGroupThreads), group_lds_bytes<MODE>(), c->stream, d, d.grp, inv_radius, dmin, dmax, ppp, ppi, pii, cand_part, n_narrow, n_narrow ? 0 : 1);
  }
}

// TrustRegionMinimizer::EvaluateGradientAndJacobian at the current x (its cost is known: c->x_cost - the cost pass of the
// candidate that became x, or of the start). What depends on the Jacobian alone is formed here: the Gram blocks of the camera
// columns with their column norms and gradient (ba_cam_gram_kernel evaluates the observations itself), the Jacobian records,
// column norms and gradients of the points OUTSIDE the point groups, the Jacobi scaling at iteration 0, the LM diagonal. The
// points of the groups have no stored Jacobian: their norms come from the kGroupNorms pass at iteration 0 (the scaling needs
// them first) and from the forward pass of the next compute_step afterwards, which also reports their max |gradient|
// (scalars[kSGmaxGrp]): c->gmax_pending tells lm_iteration to complete gradient_max_norm with it.
int evaluate_gradient_and_jacobian(mvgx_ba_ctx* c, const mvgx_ba_options* opt, bool iteration_zero) {
  Dev& d = c->d;
  phase_begin(c);
  int rc;
  if (d.n_obs) {
    if (!d.grp.n_sg) {
      hipLaunchKernelGGL(ba_linearize_kernel<true>, dim3(c->grid_obs), dim3(256), 0, c->stream, d, d.poses, d.intr, d.pts, d.part);
    } else if (d.grp.n_ungrouped) {
      hipLaunchKernelGGL(ba_linearize_list_kernel, dim3((d.grp.n_ungrouped + 255) / 256), dim3(256), 0, c->stream, d, d.grp.ungrouped, d.grp.n_ungrouped);
    }
  }
  if (d.n_priors) hipLaunchKernelGGL(ba_prior_kernel<true>, dim3(1), dim3(256), 0, c->stream, d, d.poses, 0);
  BA_LAUNCH_CHECK();
  // (always: a point without observations is on neither path's lists, and its norms / factor / step must still be defined)
  if (d.n_pts && !c->all_points_grouped) hipLaunchKernelGGL(ba_point_norms_kernel, dim3((d.n_pts + 255) / 256), dim3(256), 0, c->stream, d);
  if (d.n_pichunks) {   // (also clears the fail word)
    if (c->pinhole_family) hipLaunchKernelGGL(ba_cam_gram_kernel<true>, dim3(d.n_pichunks), dim3(256), 0, c->stream, d);
    else hipLaunchKernelGGL(ba_cam_gram_kernel<false>, dim3(d.n_pichunks), dim3(256), 0, c->stream, d);
  }
  c->fail_clear = d.n_pichunks != 0;
  c->dmin = opt->min_lm_diagonal; c->dmax = opt->max_lm_diagonal;
  // (one rank, not iteration zero, every point grouped: the finish launch also forms the LM diagonal and the gradient maximum of
  // the camera columns - there is nothing else for ba_lm_diag_kernel to do)
  const bool fold_diag = !iteration_zero && !multi_rank(c) && c->all_points_grouped && d.n_pichunks != 0;
  {
    const uint32_t wg_pi = ((uint32_t)d.n_pi + 7) / 8, wg_pose = (d.n_poses + 31) / 32, wg_all = wg_pi + wg_pose + d.n_intr * kIntrSlices;
    if (wg_all) hipLaunchKernelGGL(ba_gram_finish_kernel, dim3(wg_all), dim3(1024), 0, c->stream, d, wg_pi, wg_pose, fold_diag ? 1 : 0, c->dmin, c->dmax);
  }
  BA_LAUNCH_CHECK();
  if ((rc = all_reduce(c, d.cn_cam, d.N))) return rc;
  if ((rc = all_reduce(c, d.g_cam, d.N))) return rc;
  if (iteration_zero) {
    if (d.grp.n_sg) launch_point_groups<kGroupNorms>(c, 0.0, 0.0, 0.0);
    hipLaunchKernelGGL(ba_make_scaling_kernel, dim3(c->grid_vec), dim3(256), 0, c->stream, d, opt->jacobi_scaling);
    BA_LAUNCH_CHECK();
  }
  if (!fold_diag) {
    hipLaunchKernelGGL(ba_lm_diag_kernel, dim3(c->grid_vec), dim3(256), 0, c->stream, d, opt->min_lm_diagonal,
                       opt->max_lm_diagonal, d.part, iteration_zero ? 0 : 1);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, c->stream, d.part, c->grid_vec, 1, 1, d.scalars, kSGmax, 1);
    BA_LAUNCH_CHECK();
  }
  if ((rc = all_reduce(c, d.scalars + kSGmax, 1, MVGX_REDUCE_MAX))) return rc;   // point gradients are rank-local
  phase_end(c, kPhJacobian);
  c->gmax_pending = !iteration_zero;
  if (iteration_zero) {   // the first gradient_max_norm is tested before any step is computed
    if ((rc = read_scalars(c))) return rc;
    c->gradient_max_norm = c->h_scalars[kSGmax];
  }
  return MVGX_OK;
}

// Partial reduced camera system of this rank at the current radius (raw sums: LM diagonal not yet added)
int assemble_system(mvgx_ba_ctx* c, double inv_radius) {
  Dev& d = c->d;
  if (!c->fail_clear) MVGX_HIP(hipMemsetAsync(d.fail, 0, sizeof(int), c->stream));   // (a step that follows a rejected one: no Jacobian evaluation in between)
  c->fail_clear = false;
  const bool flat = !d.grp.n_sg || d.grp.n_ungrouped;   // some points are on the record-based path
  if (d.n_pts && !c->all_points_grouped) hipLaunchKernelGGL(ba_point_solve_kernel, dim3((d.n_pts + 255) / 256), dim3(256), 0, c->stream, d, inv_radius);
  if (d.grp.n_sg) {
    if (d.grp.n_ungrouped)
      hipLaunchKernelGGL(ba_obs_z_kernel, dim3((d.grp.n_ungrouped + 255) / 256), dim3(256), 0, c->stream, d, d.grp.ungrouped, (uint64_t)d.grp.n_ungrouped);
  } else if (d.n_obs) {
    hipLaunchKernelGGL(ba_obs_z_kernel, dim3(c->grid_obs), dim3(256), 0, c->stream, d, (const uint32_t*)nullptr, (uint64_t)d.n_obs);
  }
  if (d.n_islots && flat) hipLaunchKernelGGL(ba_slot_z_kernel, dim3((8 * d.n_islots + 255) / 256), dim3(256), 0, c->stream, d);
  BA_LAUNCH_CHECK();
  if (!d.grp.n_sg) {   // (with point groups their forward pass zeroes the system, a slice per workgroup)
    if (d.sp.enabled) MVGX_HIP(hipMemsetAsync(d.sp.A, 0, (size_t)d.sp.n_slots * 4096 * sizeof(double), c->stream));
    else MVGX_HIP(hipMemsetAsync(d.S, 0, (size_t)d.N * d.LD * sizeof(double), c->stream));
  }
  if (d.grp.n_sg) {
    launch_point_groups<kGroupForward>(c, inv_radius, c->dmin, c->dmax);   // (max |gradient| of its supergroups: reduced by the assemble launch below)
  }
  if (d.tpp.n_chunks)
    hipLaunchKernelGGL((ba_schur_products_kernel<6, 6>), dim3(8 * ((d.tpp.n_chunks + 7) / 8)), dim3(64), 0, c->stream, d.tpp, d.Zpose, d.Zpose, d.hp, d.opt);
  if (d.tpi.n_chunks)
    hipLaunchKernelGGL((ba_schur_products_kernel<6, 8>), dim3(8 * ((d.tpi.n_chunks + 7) / 8)), dim3(64), 0, c->stream, d.tpi, d.Zpose, d.Zint, d.hp, d.opt);
  if (d.tii.n_chunks)
    hipLaunchKernelGGL((ba_schur_products_kernel<8, 8>), dim3(8 * ((d.tii.n_chunks + 7) / 8)), dim3(64), 0, c->stream, d.tii, d.Zint, d.Zint, d.hp, d.slot_point);
  BA_LAUNCH_CHECK();
  {
    d.asm_inv_radius = c->diag_blocks_complete && !multi_rank(c) ? inv_radius : 0.0;
    const uint32_t wg_pp = (d.tpp.n_blocks + 7) / 8, wg_pi = (d.tpi.n_blocks + 7) / 8, wg_ii = d.tii.n_blocks;
    const uint32_t wg_all = wg_pp + wg_pi + wg_ii + (d.grp.n_sg ? 1u : 0u);
    if (wg_all) hipLaunchKernelGGL(ba_schur_assemble_all_kernel, dim3(wg_all), dim3(1024), 0, c->stream, d, wg_pp, wg_pi, wg_ii);
  }
  BA_LAUNCH_CHECK();
  return MVGX_OK;
}

// Cholesky of the summed system (rhs as extra row -> forward substitution) + back substitution -> zsol: three launches
// per 64-column block step plus one per back-substitution step, all on the solver's stream.
// Measured and rejected (profiles/round1_ba_chol_modes_call16.json): a one-step look-ahead that updates the next block
// column on the main stream and the remaining columns on a side stream (fork / join through events) was 8 % slower on
// C3 and 1.5 % faster on C5 - the cross-stream dependencies cost what the overlap gains; replaying the same sequence
// as a captured HIP graph cost 4-16 ms of instantiation per context, more than a whole small solve. An update kernel on
// 128 x 128 tiles (64 x 64 per wave) was 2.1x slower per launch than the 64 x 64 one at C5 (profiles/round1_ba_c5_update128_call18.json).
// Round 3, measured and not kept: the top levels of the elimination tree (one tile column each) as ONE launch - a single workgroup
// walking factor / panel / update level by level and straight back down the reverse sweep. The factor-and-invert body needs
// exactly four waves (its barriers count every wave of the workgroup), so the 16 sub-block tasks of a tile that the separate T / U
// launches spread over 16 waves on other CUs ran four rounds deep, two dependent memory round trips each: 4 top levels of C3
// took 235 us in the fused launch against ~140 us as 16 launches, 6 levels of C5 262 us against ~216 us.
int factor_and_solve_sparse(mvgx_ba_ctx* c) {
  Dev& d = c->d;
  const mvgx_sparse::Plan& pl = c->plan;
  if (c->lookahead) {
    // One launch per level (sp_level_kernel): the factorisation of level l beside the T / U tasks of level l - 1. A level whose diagonal
    // tiles collect more contributions of the level before than a factor workgroup takes (ba_sparse_plan.h: kMaxPre) waits for them:
    // its tasks get a launch of their own in front of its factorisation.
    auto level = [&](int f0, int nf, int pre, int u0, int nu, int t0, int nt) {
      const int per = kLevelThreads / 64, nb = nf + (nu + nt + per - 1) / per;
      if (nb) hipLaunchKernelGGL(sp_level_kernel, dim3(nb), dim3(kLevelThreads), nf ? kDiagLds : 0, c->stream, d.sp, f0, nf, pre, u0, nu, t0, nt, d.fail);
    };
    for (int l = 0; l <= pl.n_levels; ++l) {
      const int nf = l < pl.n_levels ? pl.f_start[l + 1] - pl.f_start[l] : 0, f0 = l < pl.n_levels ? pl.f_start[l] : 0;
      const bool pre = l > 0 && l < pl.n_levels && pl.lookahead[l];
      int u0 = 0, nu = 0, t0 = 0, nt = 0;
      if (l > 0) {
        u0 = pl.u_start[l - 1]; nu = (pre ? pl.u_defer_start[l - 1] : pl.u_start[l]) - u0;
        t0 = pl.t_start[l - 1]; nt = pl.t_start[l] - t0;
      }
      if (l == 0 || pre || l == pl.n_levels) level(f0, nf, pre ? 1 : 0, u0, nu, t0, nt);
      else { level(0, 0, 0, u0, nu, t0, nt); level(f0, nf, 0, 0, 0, 0, 0); }
    }
  } else
  for (int l = 0; l < pl.n_levels; ++l) {
    const int nf = pl.f_start[l + 1] - pl.f_start[l], nt = pl.t_start[l + 1] - pl.t_start[l], nu = pl.u_start[l + 1] - pl.u_start[l];
    hipLaunchKernelGGL(sp_factor_kernel, dim3(nf), dim3(256), kDiagLds, c->stream, d.sp, pl.f_start[l], d.fail);
    if (nf == 1 && nu && nu <= c->chain_fuse_max_tasks) {   // a chain level: panel and update tasks in one launch (sp_chain_level_kernel)
      hipLaunchKernelGGL(sp_chain_level_kernel, dim3((nt + nu + 3) / 4), dim3(256), 0, c->stream, d.sp, pl.t_start[l], nt, pl.u_start[l], nu,
                         pl.f_cols[pl.f_start[l]]);
      continue;
    }
    if (nt) hipLaunchKernelGGL(sp_gemm_kernel<false>, dim3((nt + 3) / 4), dim3(256), 0, c->stream, d.sp, pl.t_start[l], nt);
    if (nu) hipLaunchKernelGGL(sp_gemm_kernel<true>, dim3((nu + 3) / 4), dim3(256), 0, c->stream, d.sp, pl.u_start[l], nu);
  }
  BA_LAUNCH_CHECK();
  if (c->bs_one_launch && pl.nT > 0) {   // the whole reverse sweep as one launch, columns handed over through flags (sp_backsolve_all_kernel)
    c->bs_epoch += 1;
    if (c->bs_epoch == 0) c->bs_epoch = 1;   // (0 is the value the flags start from)
    hipLaunchKernelGGL(sp_backsolve_all_kernel, dim3(pl.nT), dim3(256), 0, c->stream, d.sp, c->bs_epoch, d.fail);
  } else {
  int l_top = pl.n_levels - 1;
  if (c->bs_chain_levels >= 2) {   // the chain at the top of the tree: one workgroup, one launch
    const int l0 = pl.n_levels - c->bs_chain_levels;
    hipLaunchKernelGGL(sp_backsolve_chain_kernel, dim3(1), dim3(256), 0, c->stream, d.sp, pl.f_start[l0], c->bs_chain_levels, l0);
    l_top = l0 - 1;
  }
  for (int l = l_top; l >= 0; --l)
    hipLaunchKernelGGL(sp_backsolve_kernel, dim3(pl.f_start[l + 1] - pl.f_start[l]), dim3(256), 0, c->stream, d.sp, pl.f_start[l]);
  }
  if (c->fold_cand_now)   // (compute_step: this step's back-substitution forms the candidate - the gather and the camera half of the step's sums are one launch)
    hipLaunchKernelGGL(ba_step_scalars_cam_kernel, dim3((d.N + 255) / 256), dim3(256), 0, c->stream, d, 1.0 / c->radius, d.part, 1);
  else
    hipLaunchKernelGGL(sp_gather_solution_kernel, dim3((d.N + 255) / 256), dim3(256), 0, c->stream, d);
  BA_LAUNCH_CHECK();
  return MVGX_OK;
}

int factor_and_solve(mvgx_ba_ctx* c) {
  Dev& d = c->d;
  if (!d.N) return MVGX_OK;
  if (d.sp.enabled) return factor_and_solve_sparse(c);
  // two-level blocking: outer panels of pw columns; inside a panel the classic 64-column steps update only the panel's
  // own columns, the rest of the trailing matrix gets one update per panel with K = pw. The trailing matrix is read and
  // written N / pw times instead of N / 64 (C5: the update is bound by that traffic). Small systems are launch-latency
  // bound instead and keep pw = 64 (= the classic right-looking sweep: no inner update, one full update per step).
  const int pw_cfg = d.N >= c->two_level_min_n ? 256 : 64;
  for (int p0 = 0; p0 < d.N; p0 += pw_cfg) {
    const int pend = std::min(d.N, p0 + pw_cfg);
    for (int k0 = p0; k0 < pend; k0 += 64) {
      const int kb = std::min(64, pend - k0);
      double* linv = d.linv + (size_t)(k0 / 64) * 8192;
      hipLaunchKernelGGL(chol_diag_inv_kernel, dim3(1), dim3(256), kDiagLds, c->stream, d.S, d.LD, k0, kb, linv, d.fail);
      const int rows_below = d.N + 1 - (k0 + kb);   // >= 1: the rhs row
      hipLaunchKernelGGL(chol_panel_mfma_kernel, dim3((rows_below + 63) / 64), dim3(256), kPanelLds, c->stream, d.S, d.N, d.LD, k0, kb, linv);
      if (k0 + kb < pend) {   // columns of this panel right of the block
        const int ntr = (rows_below + 63) / 64, ntc = (pend - (k0 + kb) + 63) / 64;
        hipLaunchKernelGGL(chol_update_mfma_kernel, dim3(ntr, ntc), dim3(256), 0, c->stream, d.S, d.N, d.LD, k0, kb, k0 + kb, pend);
      }
    }
    if (pend < d.N) {   // everything right of the panel, K = panel width
      const int nt = (d.N + 1 - pend + 63) / 64, nt2 = (d.N + 1 - pend + 127) / 128;
      if (pw_cfg > 64 && nt2 * (nt2 + 1) / 2 >= c->update128_min_tiles)
        hipLaunchKernelGGL(chol_update128_kernel, dim3(nt2, nt2), dim3(256), kUpd128Lds, c->stream, d.S, d.N, d.LD, p0, pend - p0, pend, d.N);
      else
        hipLaunchKernelGGL(chol_update_mfma_kernel, dim3(nt, nt), dim3(256), 0, c->stream, d.S, d.N, d.LD, p0, pend - p0, pend, d.N);
    }
  }
  BA_LAUNCH_CHECK();
  for (int b0 = ((d.N - 1) / 64) * 64; b0 >= 0; b0 -= 64) {
    const int kb = std::min(64, d.N - b0);
    hipLaunchKernelGGL(chol_backsolve_step_kernel, dim3(std::max(1, (b0 + 255) / 256)), dim3(256), 0, c->stream, d.S, d.N, d.LD, b0, kb,
                       d.linv + (size_t)(b0 / 64) * 8192 + 4096, d.zsol);
  }
  BA_LAUNCH_CHECK();
  return MVGX_OK;
}

// One-time agreement across ranks, before the first iteration:
//  * which camera components are free parameters with residuals SOMEWHERE (a pose or intrinsic whose observations all live
//    on other ranks is still part of the program: its Jacobi scale, LM diagonal and x-norm share must be the same on
//    every rank, or the ranks would factor different systems);
//  * the union of the non-zero camera blocks of S (see ba_mark_blocks_kernel): the packed layout of the per-iteration
//    exchange and the pattern the block-sparse solver is planned on.
__global__ void ba_masks_f64_kernel(Dev d, double* __restrict__ buf, int dir) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  if (dir == 0) { buf[i] = d.cam_active[i]; buf[d.N + i] = d.cam_counts[i]; }
  else { d.cam_active[i] = buf[i] != 0.0; d.cam_counts[i] = buf[d.N + i] != 0.0; }
}

int setup_block_exchange(mvgx_ba_ctx* c, std::vector<std::pair<uint32_t, uint32_t>>& blocks) {
  Dev& d = c->d;
  blocks = c->h_blocks;
  if (!multi_rank(c) || d.ublk_off) return MVGX_OK;
  int rc;
  if (d.N) {
    double* masks = nullptr;
    if ((rc = dev_alloc(c->pool, &masks, (size_t)2 * d.N))) return rc;
    hipLaunchKernelGGL(ba_masks_f64_kernel, dim3((d.N + 255) / 256), dim3(256), 0, c->stream, d, masks, 0);
    BA_LAUNCH_CHECK();
    if ((rc = all_reduce(c, masks, (uint64_t)2 * d.N, MVGX_REDUCE_MAX))) return rc;
    hipLaunchKernelGGL(ba_masks_f64_kernel, dim3((d.N + 255) / 256), dim3(256), 0, c->stream, d, masks, 1);
    BA_LAUNCH_CHECK();
  }
  const int n_cb = (int)d.n_poses + (int)d.n_intr;
  double* flags = nullptr;
  if ((rc = dev_alloc(c->pool, &flags, (size_t)n_cb * n_cb))) return rc;
  MVGX_HIP(hipMemsetAsync(flags, 0, (size_t)n_cb * n_cb * sizeof(double), c->stream));
  if (d.tpp.n_blocks) hipLaunchKernelGGL(ba_mark_blocks_kernel<0>, dim3((d.tpp.n_blocks + 255) / 256), dim3(256), 0, c->stream, d, d.tpp, flags, n_cb);
  if (d.tpi.n_blocks) hipLaunchKernelGGL(ba_mark_blocks_kernel<1>, dim3((d.tpi.n_blocks + 255) / 256), dim3(256), 0, c->stream, d, d.tpi, flags, n_cb);
  if (d.tii.n_blocks) hipLaunchKernelGGL(ba_mark_blocks_kernel<2>, dim3((d.tii.n_blocks + 255) / 256), dim3(256), 0, c->stream, d, d.tii, flags, n_cb);
  BA_LAUNCH_CHECK();
  if ((rc = all_reduce(c, flags, (uint64_t)n_cb * n_cb, MVGX_REDUCE_MAX))) return rc;
  std::vector<double> h((size_t)n_cb * n_cb);
  MVGX_HIP(hipMemcpyAsync(h.data(), flags, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MVGX_HIP(hipStreamSynchronize(c->stream));
  std::vector<uint32_t> brow, bcol;
  std::vector<uint8_t> bh, bw;
  std::vector<uint64_t> boff;
  uint64_t off = 0;
  auto first = [&](int cb) { return cb < (int)d.n_poses ? 6 * cb : 6 * (int)d.n_poses + 8 * (cb - (int)d.n_poses); };
  auto width = [&](int cb) { return cb < (int)d.n_poses ? 6 : 8; };
  blocks.clear();
  for (int r = 0; r < n_cb; ++r)
    for (int q = r; q < n_cb; ++q)
      if (h[(size_t)r * n_cb + q] != 0.0) {
        blocks.emplace_back((uint32_t)r, (uint32_t)q);
        brow.push_back((uint32_t)first(r)); bcol.push_back((uint32_t)first(q));
        bh.push_back((uint8_t)width(r)); bw.push_back((uint8_t)width(q));
        boff.push_back(off);
        off += (uint64_t)width(r) * width(q);
      }
  d.n_ublocks = (uint32_t)brow.size();
  d.n_packed = off + (uint64_t)d.N;   // + the rhs column
  boff.push_back(off);
  if ((rc = dev_upload(c->pool, &d.ublk_row, brow, c->stream))) return rc;
  if ((rc = dev_upload(c->pool, &d.ublk_col, bcol, c->stream))) return rc;
  if ((rc = dev_upload(c->pool, &d.ublk_h, bh, c->stream))) return rc;
  if ((rc = dev_upload(c->pool, &d.ublk_w, bw, c->stream))) return rc;
  if ((rc = dev_upload(c->pool, &d.ublk_off, boff, c->stream))) return rc;
  if ((rc = dev_alloc(c->pool, &d.packed, (size_t)d.n_packed))) return rc;
  MVGX_HIP(hipStreamSynchronize(c->stream));
  return MVGX_OK;
}

// Reduced-system solver, chosen once per context on the (union) block pattern: block-sparse tile Cholesky in a nested-
// dissection order when it needs fewer rounds of dependent launches than the dense sweep has block steps and no more tiles
// than the dense triangle, the dense blocked Cholesky otherwise.
// MVGX_BA_SOLVER=dense|sparse forces either; MVGX_BA_ND_LEAF_COLS tunes the dissection depth.
// Symbolic phase of the reduced solve (host only): nested-dissection ordering, tile structure of the factor, task lists. Depends on
// the block structure alone, so mvgx_ba_create runs it for the context's own blocks; a context that is bound to a communicator
// afterwards plans again at its first iteration, on the union of the ranks' blocks.
int plan_solver(mvgx_ba_ctx* c, const std::vector<std::pair<uint32_t, uint32_t>>& blocks) {
  Dev& d = c->d;
  if (c->plan_ready) return MVGX_OK;
  bool sparse = false;
  const uint32_t np = d.n_poses;
  if (d.N > 0 && c->solver_mode != 1) {
    mvgx_sparse::PlanParams prm;
    if (const char* env = getenv("MVGX_BA_ND_LEAF_COLS")) prm.leaf_cols = std::max(64, atoi(env));
    if (const char* env = getenv("MVGX_BA_ND_SEP_SLACK")) prm.sep_weight_slack = std::max(1.0, atof(env));   // (1: only level sets as light as the lightest compete on balance)
    if (const char* env = getenv("MVGX_BA_LOOKAHEAD_MAX_PRE")) prm.max_pre = std::max(0, atoi(env));   // (tests: the schedule's fall-back forms)
    const bool ok = mvgx_sparse::build_plan(
        (int)(np + d.n_intr), d.N, blocks, [&](int cb) { return cb < (int)np ? 6 : 8; },
        [&](int cb) { return cb < (int)np ? 6 * cb : 6 * (int)np + 8 * (cb - (int)np); }, prm, (uint64_t)1 << 25, c->plan);
    const uint64_t nd = (uint64_t)((d.N + 63) / 64);
    c->plan_tried = true; c->plan_ok = ok;
    // (round 6: "<=" - a camera graph without structure gives a plan of nd single-column levels, a dense factorisation run by the tile
    // kernels: a level is then two launches where the dense solver's block step is three, and the reverse sweep one launch where it
    // is nd - 0.59 against 0.93 ms at N = 1 203 on a scene with long tracks, call r6_14)
    sparse = ok && (c->solver_mode >= 2 || ((uint64_t)c->plan.n_levels <= nd && c->plan.n_fill_tiles <= nd * (nd + 1) / 2));
    MVGX_REQUIRE(ok || c->solver_mode != 2, MVGX_ERR_UNSUPPORTED, "MVGX_BA_SOLVER=sparse: the reduced system fills too much for the task lists");
  }
  if (sparse && getenv("MVGX_BA_PLAN_DEBUG")) {
    const mvgx_sparse::Plan& pl = c->plan;
    for (int l = 0; l < pl.n_levels; ++l) {
      int longest = 0; long total = 0;
      for (int t = pl.u_start[l]; t < pl.u_start[l + 1]; ++t) { const int n = pl.u_tasks[t].c1 - pl.u_tasks[t].c0; longest = std::max(longest, n); total += n; }
      if (l == 0) fprintf(stderr, "[mvgx ba plan] %d U task groups with split contributor lists (%d chunks)\n", pl.n_split_groups, pl.n_scratch_blocks);
      fprintf(stderr, "[mvgx ba plan] level %2d: %3d columns, %5d T tasks, %6d U tasks, %7ld contributions, longest list %d\n", l,
              pl.f_start[l + 1] - pl.f_start[l], pl.t_start[l + 1] - pl.t_start[l], pl.u_start[l + 1] - pl.u_start[l], total, longest);
    }
  }
  c->plan_sparse = sparse;
  c->plan_ready = true;
  return MVGX_OK;
}

int setup_solver(mvgx_ba_ctx* c, const std::vector<std::pair<uint32_t, uint32_t>>& blocks) {
  Dev& d = c->d;
  if (c->solver_ready) return MVGX_OK;
  int rc = plan_solver(c, blocks);
  if (rc) return rc;
  const bool sparse = c->plan_sparse;
  if (sparse) {
    const mvgx_sparse::Plan& pl = c->plan;
    SpSys& s = d.sp;
    s.nT = pl.nT; s.n_slots = pl.n_slots; s.n_levels = pl.n_levels;
    int32_t *pcol = nullptr, *tmap = nullptr, *tile_kb = nullptr, *f_cols = nullptr, *bs_start = nullptr, *bs_slot = nullptr, *bs_row = nullptr;
    mvgx_sparse::GemmTask *tt = nullptr, *ut = nullptr;
    mvgx_sparse::SlotPair *tp = nullptr, *up = nullptr;
    if ((rc = dev_upload(c->pool, &pcol, pl.pcol, c->stream))) return rc;
    if ((rc = dev_upload(c->pool, &tmap, pl.tmap, c->stream))) return rc;
    if ((rc = dev_upload(c->pool, &tile_kb, pl.tile_kb, c->stream))) return rc;
    if ((rc = dev_upload(c->pool, &f_cols, pl.f_cols, c->stream))) return rc;
    if ((rc = dev_upload(c->pool, &bs_start, pl.bs_start, c->stream))) return rc;
    if ((rc = dev_upload(c->pool, &bs_slot, pl.bs_slot, c->stream))) return rc;
    if ((rc = dev_upload(c->pool, &bs_row, pl.bs_row, c->stream))) return rc;
    int32_t* level_of = nullptr;
    if ((rc = dev_upload(c->pool, &level_of, pl.level_of, c->stream))) return rc;
    s.level_of = level_of;
    {
      int32_t *slot_col = nullptr, *pre_start = nullptr, *pre_slot = nullptr, *pre_col = nullptr;
      std::vector<int32_t> ps = pl.pre_slot, pc = pl.pre_col;
      if (ps.empty()) { ps.push_back(0); pc.push_back(0); }
      if ((rc = dev_upload(c->pool, &slot_col, pl.slot_col, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &pre_start, pl.pre_start, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &pre_slot, ps, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &pre_col, pc, c->stream))) return rc;
      MVGX_HIP(hipStreamSynchronize(c->stream));   // (ps / pc are locals)
      s.slot_col = slot_col; s.pre_start = pre_start; s.pre_slot = pre_slot; s.pre_col = pre_col;
      int32_t* bs_rec = nullptr;
      if ((rc = dev_upload(c->pool, &bs_rec, pl.bs_rec, c->stream))) return rc;
      s.bs_rec = bs_rec;
      if ((rc = dev_alloc(c->pool, &s.z_flag, (size_t)std::max(pl.nT, 1)))) return rc;
      MVGX_HIP(hipMemsetAsync(s.z_flag, 0, (size_t)std::max(pl.nT, 1) * sizeof(unsigned), c->stream));
      c->bs_epoch = 0;
    }
    {   // the chain at the top: as many single-column levels as the kernel's LDS tables hold (MVGX_BA_BACKSOLVE_CHAIN=0: none)
      const char* env = getenv("MVGX_BA_BACKSOLVE_CHAIN");
      int n = 0; size_t entries = 0;
      if (!(env && atoi(env) == 0))
        for (int l = pl.n_levels - 1; l >= 0 && pl.f_start[l + 1] - pl.f_start[l] == 1 && n < kChainMax; --l) {
          const int k = pl.f_cols[pl.f_start[l]];
          const size_t e = (size_t)(pl.bs_start[k + 1] - pl.bs_start[k]);
          if (entries + e > (size_t)kChainEntriesMax) break;
          entries += e; ++n;
        }
      c->bs_chain_levels = n;
      if (getenv("MVGX_BA_PLAN_DEBUG")) fprintf(stderr, "[mvgx ba plan] tile columns %d, levels %d, chain at the top %d levels (%zu tiles below its columns)\n", pl.nT, pl.n_levels, n, entries);
    }
    if ((rc = dev_upload(c->pool, &tt, pl.t_tasks, c->stream))) return rc;
    if ((rc = dev_upload(c->pool, &ut, pl.u_tasks, c->stream))) return rc;
    if ((rc = dev_upload(c->pool, &tp, pl.t_pairs, c->stream))) return rc;
    if ((rc = dev_upload(c->pool, &up, pl.u_pairs, c->stream))) return rc;
    s.pcol = pcol; s.tmap = tmap; s.tile_kb = tile_kb; s.f_cols = f_cols; s.bs_start = bs_start; s.bs_slot = bs_slot; s.bs_row = bs_row;
    s.t_tasks = tt; s.u_tasks = ut; s.t_pairs = tp; s.u_pairs = up;
    if ((rc = dev_alloc(c->pool, &s.A, (size_t)pl.n_slots * 4096))) return rc;
    if ((rc = dev_alloc(c->pool, &s.L, (size_t)pl.n_slots * 4096))) return rc;
    if ((rc = dev_alloc(c->pool, &s.Linv, (size_t)pl.nT * 4096))) return rc;
    if ((rc = dev_alloc(c->pool, &s.z, (size_t)pl.nT * 64))) return rc;
    if ((rc = dev_alloc(c->pool, &s.u_scratch, (size_t)std::max(pl.n_scratch_blocks, 1) * 256))) return rc;
    if ((rc = dev_alloc(c->pool, &s.u_counter, (size_t)std::max(pl.n_split_groups, 1)))) return rc;
    MVGX_HIP(hipMemsetAsync(s.u_counter, 0, (size_t)std::max(pl.n_split_groups, 1) * sizeof(unsigned), c->stream));
    // rows 1..63 of the rhs tiles and the padding are never written: zero once
    MVGX_HIP(hipMemsetAsync(s.L, 0, (size_t)pl.n_slots * 4096 * sizeof(double), c->stream));
    MVGX_HIP(hipMemsetAsync(s.z, 0, (size_t)pl.nT * 64 * sizeof(double), c->stream));
    s.enabled = 1;   // (the host vectors behind the uploads are the plan's: they live as long as the context)
  } else if (d.N > 0) {
    const size_t bytes = (size_t)d.N * d.LD * sizeof(double) + (size_t)((d.N + 63) / 64) * 8192 * sizeof(double);
    size_t free_b = 0, total_b = 0;
    mvgx::trim_device_cache();   // idle cached slabs would read as used memory
    MVGX_HIP(hipMemGetInfo(&free_b, &total_b));
    MVGX_REQUIRE(bytes < free_b, MVGX_ERR_UNSUPPORTED,
                 "reduced camera system of %d columns: its factor fills more than half of the dense triangle, and the dense "
                 "storage (%.1f GB) exceeds the free device memory (%.1f GB); the reference would use an iterative Schur "
                 "solver here, which this library does not provide", d.N, bytes / 1e9, free_b / 1e9);
    if ((rc = dev_alloc(c->pool, &d.S, (size_t)d.N * d.LD))) return rc;
    if ((rc = dev_alloc(c->pool, &d.linv, (size_t)((d.N + 63) / 64) * 8192))) return rc;
  }
  c->solver_ready = true;
  return MVGX_OK;
}

// sum of the partial reduced systems over the ranks: the one bulk exchange of the iteration (RCCL over xGMI)
int exchange_system(mvgx_ba_ctx* c) {
  Dev& d = c->d;
  if (!multi_rank(c) || !d.N) return MVGX_OK;
  hipLaunchKernelGGL(ba_pack_system_kernel, dim3(d.n_ublocks + 1), dim3(64), 0, c->stream, d, 0);
  BA_LAUNCH_CHECK();
  const int rc = all_reduce(c, d.packed, d.n_packed);
  if (rc) return rc;
  hipLaunchKernelGGL(ba_pack_system_kernel, dim3(d.n_ublocks + 1), dim3(64), 0, c->stream, d, 1);
  BA_LAUNCH_CHECK();
  return MVGX_OK;
}

// x + delta and its cost, enqueued behind the step that produced delta WITHOUT waiting for the step's verdict: an invalid step
// (failed factorisation, non-positive model cost change) is rare and only wastes these launches - the candidate arrays and
// scalars they write are not read in that case - while every valid step saves one host round trip per iteration.
int enqueue_candidate_and_cost(mvgx_ba_ctx* c, bool candidate_done) {
  Dev& d = c->d;
  if (c->candidate_cost_done) return all_reduce(c, d.scalars + kSCost, 2);   // (the back-substitution pass has formed both)
  phase_begin(c);
  int rc;
  if (!candidate_done) {   // (else ba_step_scalars_kernel has formed the candidate with the model cost)
    // the kernel writes EVERY entry of the three candidate arrays (x + 0 for constant parameters): no copy of x beforehand
    hipLaunchKernelGGL(ba_candidate_kernel, dim3(c->grid_vec), dim3(256), 0, c->stream, d, d.part);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, c->stream, d.part, c->grid_vec, 4, 4, d.scalars, kSCamStepSq, 0);
    BA_LAUNCH_CHECK();
    if ((rc = all_reduce(c, d.scalars + kSStepSq, 2))) return rc;   // point parts
  }
  if ((rc = eval<false>(c, d.cposes, d.cintr, d.cpts))) return rc;
  phase_end(c, kPhCost);
  return MVGX_OK;
}

// LevenbergMarquardtStrategy::ComputeStep + SchurComplementSolver::SolveImpl. ok=false <=> LINEAR_SOLVER_FAILURE.
int evaluate_gradient_and_jacobian(mvgx_ba_ctx* c, const mvgx_ba_options* opt, bool iteration_zero);
// The Jacobian evaluation of the NEXT iteration at x + delta, enqueued behind the publish kernel and gated by the device's accept word
// (ba_publish_scalars_kernel): the pointers of x and the candidate are exchanged for the launches and exchanged back - the host accepts,
// or does not, when it has read the scalars. Host-side flags the evaluation sets are restored: they become true only with the acceptance.
int speculative_evaluation(mvgx_ba_ctx* c, void* arg) {
  const mvgx_ba_options* opt = static_cast<const mvgx_ba_options*>(arg);
  Dev& d = c->d;
  const bool fail_clear = c->fail_clear, gmax_pending = c->gmax_pending;
  const double dmin = c->dmin, dmax = c->dmax;
  std::swap(d.poses, d.cposes); std::swap(d.intr, d.cintr); std::swap(d.pts, d.cpts);
  d.gate = c->d_accept;
  const int rc = evaluate_gradient_and_jacobian(c, opt, false);
  d.gate = nullptr;
  std::swap(d.poses, d.cposes); std::swap(d.intr, d.cintr); std::swap(d.pts, d.cpts);
  c->fail_clear = fail_clear; c->gmax_pending = gmax_pending; c->dmin = dmin; c->dmax = dmax;
  c->spec_done = rc == MVGX_OK;
  return rc;
}

int compute_step(mvgx_ba_ctx* c, bool* ok, double* model_cost_change, const mvgx_ba_options* opt) {
  Dev& d = c->d;
  const double inv_radius = 1.0 / c->radius;
  phase_begin(c);
  int rc = assemble_system(c, inv_radius);
  if (rc) return rc;
  if ((rc = exchange_system(c))) return rc;
  if (d.N && d.asm_inv_radius == 0.0) hipLaunchKernelGGL(ba_finish_system_kernel, dim3((d.N + 255) / 256), dim3(256), 0, c->stream, d, inv_radius);   // (else: done by the assemble launch)
  BA_LAUNCH_CHECK();
  phase_end(c, kPhSchur);
  phase_begin(c);
  // every point grouped: the back-substitution pass forms the candidate x + delta and its cost on the way (one pass over the
  // observations less per iteration); the camera half of the step's sums runs in front of it (MVGX_BA_SEPARATE_COST=1: the passes of old)
  const bool fold_cand = c->fold_candidate && c->all_points_grouped && d.grp.n_sg && !c->model_cost_from_jacobian;
  c->fold_cand_now = fold_cand;
  if ((rc = factor_and_solve(c))) return rc;
  phase_end(c, kPhSolve);
  phase_begin(c);
  if (!(d.sp.enabled && c->all_points_grouped))   // (sparse solve + every point grouped: the gather kernel has written the camera steps, no point is left for this kernel)
    hipLaunchKernelGGL(ba_backsub_kernel, dim3(c->grid_vec), dim3(256), 0, c->stream, d);   // camera steps; points of the record-based path
  c->candidate_cost_done = fold_cand;
  if (fold_cand) {
    const int n_cam_wg = (d.N + 255) / 256;
    if (!d.sp.enabled) hipLaunchKernelGGL(ba_step_scalars_cam_kernel, dim3(n_cam_wg), dim3(256), 0, c->stream, d, inv_radius, d.part, 0);   // (else: with the gather)
    launch_point_groups<kGroupBacksub>(c, inv_radius, c->dmin, c->dmax, d.grp_part);
    hipLaunchKernelGGL(ba_step_reduce_kernel, dim3(8), dim3(1024), 0, c->stream, d.part, n_cam_wg, d.grp_part, (int)d.grp.n_sg, d.scalars);
    if (d.n_priors) hipLaunchKernelGGL(ba_prior_kernel<false>, dim3(1), dim3(256), 0, c->stream, d, d.cposes, 1);
    BA_LAUNCH_CHECK();
    if ((rc = all_reduce(c, d.scalars + kSStepSq, 3))) return rc;   // (the cost's all-reduce: where the separate cost pass has it - ranks may differ in fold_cand)
  } else if (d.grp.n_sg) launch_point_groups<kGroupBacksub>(c, inv_radius, c->dmin, c->dmax);
  if (fold_cand) {
  } else if (c->model_cost_from_jacobian) {   // Ceres' own form (trust_region_minimizer.cc:402-405): one more pass over the Jacobian records
    if (d.n_obs) hipLaunchKernelGGL(ba_model_cost_kernel, dim3(c->grid_obs), dim3(256), 0, c->stream, d, d.part);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, c->stream, d.part, d.n_obs ? c->grid_obs : 0, 1, 1, d.scalars,
                       kSModel, 0);
    if (d.n_priors) hipLaunchKernelGGL(ba_prior_model_kernel, dim3(1), dim3(256), 0, c->stream, d);
    BA_LAUNCH_CHECK();
    if ((rc = all_reduce(c, d.scalars + kSModel, 1))) return rc;
  } else {
    hipLaunchKernelGGL(ba_step_scalars_kernel, dim3(c->grid_vec), dim3(256), 0, c->stream, d, inv_radius, d.part);   // model cost + candidate
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, c->stream, d.part, c->grid_vec, 6, 6, d.scalars, kSCamStepSq, 0);
    BA_LAUNCH_CHECK();
    if ((rc = all_reduce(c, d.scalars + kSStepSq, 3))) return rc;   // the rank-local point parts; the camera parts are the same on every rank
  }
  if (multi_rank(c)) {   // a point block that failed to invert on one rank fails the step everywhere
    hipLaunchKernelGGL(ba_pack_fail_kernel, dim3(1), dim3(1), 0, c->stream, d);
    BA_LAUNCH_CHECK();
    if ((rc = all_reduce(c, d.scalars + kSFail, 1, MVGX_REDUCE_MAX))) return rc;
  }
  if (c->gmax_pending && multi_rank(c)) {   // max |gradient| of the grouped points (rank-local), from this step's forward pass
    if ((rc = all_reduce(c, d.scalars + kSGmaxGrp, 1, MVGX_REDUCE_MAX))) return rc;
  }
  phase_end(c, kPhBacksub);
  if ((rc = enqueue_candidate_and_cost(c, !c->model_cost_from_jacobian))) return rc;
  // the usual case - one rank, every point grouped, no priors, the candidate's cost from the back-substitution pass - launches the next
  // iteration's Jacobian evaluation ahead of the decision (two gated kernels: Gram + finish)
  c->spec_done = false;
  const bool spec = c->speculate && c->poll_scalars && fold_cand && !multi_rank(c) && !d.n_priors && d.n_pichunks && !c->model_cost_from_jacobian &&
                    c->iteration < opt->max_num_iterations;
  if ((rc = read_scalars(c, spec ? 1 : 0, opt, spec ? &speculative_evaluation : nullptr, const_cast<mvgx_ba_options*>(opt)))) return rc;
  if (c->gmax_pending) {   // the Jacobian evaluation before this step left its max |gradient| on the device (see there)
    c->gradient_max_norm = std::max(c->h_scalars[kSGmax], c->h_scalars[kSGmaxGrp]);
    c->gmax_pending = false;
    c->gmax_resolved = true;
  }
  *model_cost_change = c->model_cost_from_jacobian ? c->h_scalars[kSModel] : c->h_scalars[kSModelPt] + c->h_scalars[kSModelCam];
  const bool failed = multi_rank(c) ? (c->h_scalars[kSFail] != 0.0) : (*c->h_fail != 0);
  *ok = !failed && std::isfinite(*model_cost_change);
  return MVGX_OK;
}

// step and parameter norms and the candidate's cost, from the scalars compute_step fetched
void candidate_results(const mvgx_ba_ctx* c, double* step_norm, double* x_norm, double* cand_cost) {
  *step_norm = std::sqrt(c->h_scalars[kSCamStepSq] + c->h_scalars[kSStepSq]);
  *x_norm = std::sqrt(c->h_scalars[kSCamXSq] + c->h_scalars[kSXSq]);
  *cand_cost = c->h_scalars[kSCost];
}

int accept_candidate(mvgx_ba_ctx* c) {
  Dev& d = c->d;   // passed by value to every launch: exchanging the pointers here is the whole acceptance, nothing is copied
  std::swap(d.poses, d.cposes);
  std::swap(d.intr, d.cintr);
  std::swap(d.pts, d.cpts);
  return MVGX_OK;
}

int global_obs_count(mvgx_ba_ctx* c) {
  c->n_obs_global = c->n_obs_rmse_local;
  if (!multi_rank(c)) return MVGX_OK;
  MVGX_HIP(hipMemcpyAsync(c->d.scalars + kSNobs, &c->n_obs_global, sizeof(double), hipMemcpyHostToDevice, c->stream));
  int rc = all_reduce(c, c->d.scalars + kSNobs, 1);
  if (rc) return rc;
  if ((rc = read_scalars(c))) return rc;
  c->n_obs_global = c->h_scalars[kSNobs];
  return MVGX_OK;
}

double rmse_from(const mvgx_ba_ctx* c) {   // of the current x
  return c->n_obs_global > 0 ? std::sqrt(c->x_sqerr / (2.0 * c->n_obs_global)) : 0.0;
}

int start(mvgx_ba_ctx* c, const mvgx_ba_options* opt) {
  int rc = global_obs_count(c);
  if (rc) return rc;
  {
    std::vector<std::pair<uint32_t, uint32_t>> blocks;
    if ((rc = setup_block_exchange(c, blocks))) return rc;
    if ((rc = setup_solver(c, blocks))) return rc;
  }
  if ((rc = eval<false>(c, c->d.poses, c->d.intr, c->d.pts))) return rc;
  if ((rc = read_scalars(c))) return rc;
  c->x_cost = c->h_scalars[kSCost];
  c->x_sqerr = c->h_scalars[kSSqErr];
  c->initial_rmse = rmse_from(c);
  c->radius = opt->initial_radius;
  c->decrease_factor = 2.0;
  c->reuse_diagonal = false; c->x_norm_valid = false; c->last_successful = true;
  c->iteration = 0; c->invalid = 0; c->successful = 0; c->termination = 1; c->finished = false;
  if ((rc = evaluate_gradient_and_jacobian(c, opt, true))) return rc;  // IterationZero
  c->initial_cost = c->x_cost;
  c->started = true;
  return MVGX_OK;
}

// One pass of the while-loop of TrustRegionMinimizer::Minimize. Sets c->finished when a termination test fires.
int lm_iteration(mvgx_ba_ctx* c, const mvgx_ba_options* opt) {
  // FinalizeIterationAndCheckIfMinimizerCanContinue
  if (c->last_successful) ++c->successful;
  if (c->iteration >= opt->max_num_iterations) { c->termination = 1; c->finished = true; return MVGX_OK; }
  if (c->last_successful && !c->gmax_pending && c->gradient_max_norm <= opt->gradient_tolerance) { c->termination = 0; c->finished = true; return MVGX_OK; }
  if (c->radius <= opt->min_radius) { c->termination = 0; c->finished = true; return MVGX_OK; }
  ++c->iteration;
  bool ok = false;
  double model_cost_change = 0;
  c->gmax_resolved = false;
  int rc = compute_step(c, &ok, &model_cost_change, opt);
  if (rc) return rc;
  // The gradient test of this iteration's start, made now that the max |gradient| of the last Jacobian evaluation has come back
  // with the step's scalars (one host round trip per iteration instead of two): the step just computed is dropped, as if the
  // test had fired before it.
  if (c->gmax_resolved && c->last_successful && c->gradient_max_norm <= opt->gradient_tolerance) {
    --c->iteration; c->termination = 0; c->finished = true; return MVGX_OK;
  }
  c->reuse_diagonal = true;
  if (!(ok && model_cost_change > 0.0)) {  // HandleInvalidStep + StepIsInvalid (= StepRejected(0))
    if (++c->invalid >= opt->max_consecutive_invalid_steps) { c->termination = 2; c->finished = true; return MVGX_OK; }
    c->radius = c->radius / c->decrease_factor; c->decrease_factor *= 2.0;
    c->last_successful = false;
    return MVGX_OK;
  }
  c->invalid = 0;
  double step_norm, x_norm, cand;
  candidate_results(c, &step_norm, &x_norm, &cand);
  if (!c->x_norm_valid) x_norm = -1.0;  // Init() leaves x_norm_ = -1 until the first accepted step
  if (opt->verbose)
    fprintf(stderr, "[mvgx ba] it %d cost %.9e cand %.9e model %.6e radius %.3e |step| %.3e\n", c->iteration, c->x_cost, cand,
            model_cost_change, c->radius, step_norm);
  if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) { c->termination = 0; c->finished = true; return MVGX_OK; }
  if (std::fabs(c->x_cost - cand) <= opt->function_tolerance * c->x_cost) { c->termination = 0; c->finished = true; return MVGX_OK; }
  const double relative_decrease = (c->x_cost - cand) / model_cost_change;
  const bool accept = relative_decrease > opt->min_relative_decrease;
  // (a Jacobian evaluation launched ahead runs on the device's own decision: the two are the same test on the same doubles)
  const int dev_word = c->spec_done ? (int)c->h_scalars[kSCount + 2] : 0;   // accept + 2 gate
  MVGX_REQUIRE(!c->spec_done || accept == ((dev_word & 1) != 0), MVGX_ERR_NUMERIC, "internal error: the device's accept decision differs from the host's");
  if (c->spec_done && !(dev_word & 2)) c->spec_done = false;   // (the gate stayed shut - the device saw the solve end here: the evaluation is launched below, as without the look-ahead)
  if (accept) {
    if ((rc = accept_candidate(c))) return rc;
    c->x_cost = cand;   // the cost pass of the candidate IS the cost at the new x
    c->x_sqerr = c->h_scalars[kSSqErr];
    if (c->spec_done) {   // the evaluation at the new x is on its way: what it leaves on the host side
      c->fail_clear = c->d.n_pichunks != 0;
      c->dmin = opt->min_lm_diagonal; c->dmax = opt->max_lm_diagonal;
      c->gmax_pending = true;
    } else if ((rc = evaluate_gradient_and_jacobian(c, opt, false))) return rc;
    c->radius = c->radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
    c->radius = std::min(opt->max_radius, c->radius);
    c->decrease_factor = 2.0; c->reuse_diagonal = false;
    c->last_successful = true; c->x_norm_valid = true;
  } else {
    c->radius = c->radius / c->decrease_factor; c->decrease_factor *= 2.0;
    c->last_successful = false;
  }
  return MVGX_OK;
}

int fill_summary(mvgx_ba_ctx* c, mvgx_ba_summary* s) {
  if (!s) return MVGX_OK;
  // (cost and squared error at the final x were computed when that x was evaluated as a candidate: no pass of their own)
  MVGX_HIP(hipStreamSynchronize(c->stream));
  s->num_iterations = c->iteration;
  s->num_successful_steps = c->successful;
  s->termination = c->termination;
  s->initial_cost = c->initial_cost;
  s->final_cost = c->x_cost;
  s->initial_rmse = c->initial_rmse;
  s->final_rmse = rmse_from(c);
  phase_collect(c);   // (the stream is idle)
  s->jacobian_ms = c->phase_ms[kPhJacobian]; s->schur_ms = c->phase_ms[kPhSchur]; s->solve_ms = c->phase_ms[kPhSolve];
  s->backsub_ms = c->phase_ms[kPhBacksub]; s->cost_ms = c->phase_ms[kPhCost];
  return MVGX_OK;
}

}  // namespace

namespace {
// Which components of the camera blocks are free parameters (cam_active) and which count in the norms of the termination tests
// (cam_counts: every component of a block that is in the program, sfm_data_BA_ceres.cpp:274-306, :321-344): from the caller's
// constant masks and the blocks that carry a residual.
void camera_component_flags(const mvgx_ba_problem* p, const std::vector<uint8_t>& pose_used, const std::vector<uint8_t>& intr_used,
                            std::vector<uint8_t>& cam_active, std::vector<uint8_t>& cam_counts) {
  const size_t N = 6 * (size_t)p->n_poses + 8 * (size_t)p->n_intrinsics;
  cam_active.assign(N, 0); cam_counts.assign(N, 0);
  for (uint32_t i = 0; i < p->n_poses; ++i) {
    const uint8_t m = p->pose_const_mask ? p->pose_const_mask[i] : 0;
    const bool in_program = pose_used[i] && ((m & 0x3F) != 0x3F);
    for (int cpt = 0; cpt < 6; ++cpt) {
      cam_active[6 * (size_t)i + cpt] = in_program && !((m >> cpt) & 1);
      cam_counts[6 * (size_t)i + cpt] = in_program;
    }
  }
  for (uint32_t k = 0; k < p->n_intrinsics; ++k) {
    const int K = mvgx_ba::intr_param_count(p->intr_model[k]);
    const uint8_t m = p->intr_const_mask ? p->intr_const_mask[k] : 0;
    const uint8_t full = (uint8_t)((1u << K) - 1u);
    const bool in_program = intr_used[k] && K > 0 && ((m & full) != full);
    for (int cpt = 0; cpt < 8; ++cpt) {
      cam_active[6 * (size_t)p->n_poses + 8 * (size_t)k + cpt] = in_program && cpt < K && !((m >> cpt) & 1);
      cam_counts[6 * (size_t)p->n_poses + 8 * (size_t)k + cpt] = in_program && cpt < K;
    }
  }
}
}  // namespace
namespace mvgx {
// 128 bits over everything mvgx_ba_create turns into structure (lists, groups, product lists, the plan of the reduced solve):
// counts, the three index lists of the observations, camera models, which points are free, which observations are control
// points, whether weights exist, which poses carry priors. NOT in it (values, re-bound by mvgx_ba_update): parameters, image
// points, weights' values, prior centres / weights, the constant masks of poses and intrinsics, the loss scales.
// Two multiply-xorshift chains with different constants; the observation lists are hashed grain by grain on the host threads
// and the grains' words chained in order.
BaFingerprint ba_fingerprint(const mvgx_ba_problem* p) {
  struct Chain {
    uint64_t a, b;
    void add(uint64_t v) {
      a = (a ^ v) * 0x9E3779B97F4A7C15ull; a ^= a >> 32;
      b = (b + v) * 0xC2B2AE3D27D4EB4Full; b ^= b >> 29;
    }
  };
  Chain h{0x243F6A8885A308D3ull, 0x13198A2E03707344ull};
  h.add(p->n_poses); h.add(p->n_intrinsics); h.add(p->n_points); h.add(p->n_obs); h.add(p->n_pose_priors);
  h.add((uint64_t)(p->points_constant != 0) | (uint64_t)(p->obs_weight != nullptr) << 1 | (uint64_t)(p->obs_is_control != nullptr) << 2 |
        (uint64_t)(p->point_const_mask != nullptr) << 3);
  for (uint32_t k = 0; k < p->n_intrinsics; ++k) h.add((uint64_t)(uint32_t)p->intr_model[k]);
  for (uint32_t k = 0; k < p->n_pose_priors; ++k) h.add(p->prior_pose[k]);
  constexpr uint64_t kGrain = 1u << 15;
  {
    const uint64_t n_grains = (p->n_obs + kGrain - 1) / kGrain;
    std::vector<Chain> part((size_t)n_grains);
    parallel_for_dynamic((size_t)n_grains, 1, host_threads(p->n_obs), [&](size_t g, unsigned) {
      Chain c{0x452821E638D01377ull + g, 0xBE5466CF34E90C6Cull ^ g};
      for (uint64_t k = g * kGrain, e = std::min<uint64_t>(p->n_obs, (g + 1) * kGrain); k < e; ++k) {
        c.add((uint64_t)p->obs_pose[k] << 32 | p->obs_point[k]);
        c.add((uint64_t)p->obs_intr[k] << 1 | (uint64_t)(p->obs_is_control && p->obs_is_control[k]));
      }
      part[g] = c;
    });
    for (const Chain& c : part) { h.add(c.a); h.add(c.b); }
  }
  if (p->point_const_mask) {
    const uint64_t n_grains = ((uint64_t)p->n_points + kGrain - 1) / kGrain;
    std::vector<Chain> part((size_t)n_grains);
    parallel_for_dynamic((size_t)n_grains, 1, host_threads(p->n_points), [&](size_t g, unsigned) {
      Chain c{0xC0AC29B7C97C50DDull + g, 0x3F84D5B5B5470917ull ^ g};
      for (uint64_t k = g * kGrain, e = std::min<uint64_t>(p->n_points, (g + 1) * kGrain); k < e; ++k) c.add(p->point_const_mask[k] != 0);
      part[g] = c;
    });
    for (const Chain& c : part) { h.add(c.a); h.add(c.b); }
  }
  return BaFingerprint{{h.a, h.b}};
}

// Argument checks of a problem description, shared by the single- and the multi-device entry points: bad input is MVGX_ERR_ARG /
// MVGX_ERR_UNSUPPORTED here, not a fault inside a worker thread later.
int ba_validate_problem(const mvgx_ba_problem* p) {
  MVGX_REQUIRE((p->poses || !p->n_poses) && (p->intrinsics || !p->n_intrinsics) && (p->points || !p->n_points), MVGX_ERR_ARG,
               "mvgx_ba_create: NULL parameter array");
  MVGX_REQUIRE(p->intr_model || !p->n_intrinsics, MVGX_ERR_ARG, "mvgx_ba_create: NULL camera-model array");
  MVGX_REQUIRE(!p->n_obs || (p->obs_pose && p->obs_intr && p->obs_point && p->obs_xy), MVGX_ERR_ARG,
               "mvgx_ba_create: NULL observation array");
  MVGX_REQUIRE(!p->n_pose_priors || (p->prior_pose && p->prior_center && p->prior_weight), MVGX_ERR_ARG,
               "mvgx_ba_create: NULL pose-prior array");
  for (uint32_t k = 0; k < p->n_intrinsics; ++k)
    MVGX_REQUIRE(mvgx_ba::intr_param_count(p->intr_model[k]) >= 0, MVGX_ERR_UNSUPPORTED,
                 "intrinsic %u: camera model %d has no cost functor (sfm_data_BA_ceres.cpp:84-108)", k, p->intr_model[k]);
  {
    constexpr uint64_t kGrain = 1u << 16;
    const uint64_t n_grains = (p->n_obs + kGrain - 1) / kGrain;
    std::atomic<uint64_t> bad{UINT64_MAX};
    parallel_for_dynamic((size_t)n_grains, 1, host_threads(p->n_obs), [&](size_t g, unsigned) {
      for (uint64_t k = g * kGrain, e = std::min<uint64_t>(p->n_obs, (g + 1) * kGrain); k < e; ++k)
        if (!(p->obs_pose[k] < p->n_poses && p->obs_intr[k] < p->n_intrinsics && p->obs_point[k] < p->n_points)) {
          uint64_t cur = bad.load();
          while (k < cur && !bad.compare_exchange_weak(cur, k)) {}
          return;
        }
    });
    MVGX_REQUIRE(bad.load() == UINT64_MAX, MVGX_ERR_ARG, "observation %llu references a block out of range", (unsigned long long)bad.load());
  }
  for (uint32_t k = 0; k < p->n_pose_priors; ++k)
    MVGX_REQUIRE(p->prior_pose[k] < p->n_poses, MVGX_ERR_ARG, "pose prior %u references a pose out of range", k);
  return MVGX_OK;
}
static thread_local bool g_defer_plan = false;
void ba_create_defer_plan(bool on) { g_defer_plan = on; }
bool ba_create_plan_deferred() { return g_defer_plan; }
void ba_ctx_comm_abort(mvgx_ba_ctx* c) {
  if (c && c->rccl) rccl_abort(c->rccl);
}
}  // namespace mvgx

namespace mvgx { bool ba_create_plan_deferred(); }

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

int mvgx_ba_create_multi(const int* devices, int n_devices, const mvgx_ba_problem* p, mvgx_ba_ctx** out) {
  MVGX_REQUIRE(devices && p && out && n_devices >= 1, MVGX_ERR_ARG, "mvgx_ba_create_multi: bad argument");
  { const int vrc = mvgx::ba_validate_problem(p); if (vrc) return vrc; }
  // never more shards than points (an empty shard would only add latency)
  n_devices = (int)std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)n_devices, p->n_points));
  if (n_devices == 1) return mvgx_ba_create(devices[0] < 0 ? -2 : devices[0], p, out);
  mvgx::BaMulti* m = nullptr;
  const int rc = mvgx::ba_multi_create(devices, n_devices, p, &m);
  if (rc) return rc;
  auto* c = new mvgx_ba_ctx();
  c->multi = m;
  c->device = devices[0];
  *out = c;
  return MVGX_OK;
}

int mvgx_ba_create(int device, const mvgx_ba_problem* p, mvgx_ba_ctx** out) {
  MVGX_REQUIRE(p && out, MVGX_ERR_ARG, "mvgx_ba_create: NULL argument");
  const auto t_enter = std::chrono::steady_clock::now();
  { const int vrc = mvgx::ba_validate_problem(p); if (vrc) return vrc; }   // before any dispatch to the multi-device path
  if (device == -1) {   // "no preference": MVGX_DEVICES may name the device(s); small problems stay on one device
    std::vector<int> devs;
    const int rc = mvgx::devices_from_env(devs);
    if (rc) return rc;
    uint64_t min_obs = 200000;   // below this an LM iteration is a few hundred microseconds of device work: the exchange would dominate
    if (const char* env = getenv("MVGX_BA_MULTI_MIN_OBS")) min_obs = strtoull(env, nullptr, 10);
    if (devs.size() >= 2 && p->n_obs >= min_obs) return mvgx_ba_create_multi(devs.data(), (int)devs.size(), p, out);
    if (!devs.empty()) device = devs[0];
  }
  if (device < 0) device = -1;
  MVGX_REQUIRE(p->n_obs < (1ull << 31), MVGX_ERR_ARG, "mvgx_ba_create: too many observations for one device shard");
  int rc = mvgx::select_device(device);
  if (rc) return rc;
  // MVGX_BA_CREATE_TIMING=1: phase times of this function on stderr (host structure vs allocation vs upload)
  const bool timing = getenv("MVGX_BA_CREATE_TIMING") != nullptr;
  auto t_prev = t_enter;
  auto tick = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[mvgx_ba_create] %-36s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  mvgx::HostArena ha;   // the large lists of the structure build and the sources of the uploads (declared before the guard: the
                        // guard's destroy drains the stream before the slabs go back to the cache)
  auto* c = new mvgx_ba_ctx();
  struct Guard {   // every early return below (allocation failure, bad product list) releases what was built so far
    mvgx_ba_ctx* c;
    ~Guard() { if (c) mvgx_ba_destroy(c); }
  } guard{c};
  c->fingerprint = mvgx::ba_fingerprint(p);
  tick("validation, fingerprint");
  MVGX_HIP(hipGetDevice(&c->device));
  if ((rc = mvgx::acquire_stream(&c->stream))) return rc;
  tick("stream");
  MVGX_HIP(hipEventCreate(&c->ev0));
  MVGX_HIP(hipEventCreate(&c->ev1));
  tick("events");
  MVGX_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_scalars), (kSCount + 3) * sizeof(double), hipHostMallocDefault));
  memset(c->h_scalars, 0, (kSCount + 3) * sizeof(double));
  if (const char* env = getenv("MVGX_BA_SPECULATE")) c->speculate = atoi(env) != 0;
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_scalars_dev), c->h_scalars, 0) != hipSuccess) { (void)hipGetLastError(); c->poll_scalars = false; }
  if (const char* env = getenv("MVGX_BA_POLL_SCALARS")) c->poll_scalars = c->poll_scalars && atoi(env) != 0;
  if (const char* env = getenv("MVGX_BA_SEPARATE_COST")) c->fold_candidate = atoi(env) == 0;
  if (const char* env = getenv("MVGX_BA_CHAIN_FUSE")) c->chain_fuse_max_tasks = atoi(env);
  if (const char* env = getenv("MVGX_BA_LOOKAHEAD")) c->lookahead = atoi(env) != 0;
  if (const char* env = getenv("MVGX_BA_BACKSOLVE_FLAGS")) c->bs_one_launch = atoi(env) != 0;
  tick("page-locked scalars");
  c->h_fail = reinterpret_cast<int*>(c->h_scalars + kSCount);
  Dev& d = c->d;
  d.n_poses = p->n_poses; d.n_intr = p->n_intrinsics; d.n_pts = p->n_points; d.n_obs = p->n_obs; d.n_priors = p->n_pose_priors;
  d.N = 6 * (int)d.n_poses + 8 * (int)d.n_intr; d.LD = d.N + 1;
  d.huber_a = p->huber_a;
  d.prior_huber_a = p->prior_huber_a;
  c->phase_timing = getenv("MVGX_BA_PHASE_TIMING") != nullptr;
  if (getenv("MVGX_BA_FACTOR_DEBUG")) { const int one = 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_factor_debug), &one, sizeof(one)); }
  {   // (a synchronous copy: only when the setting changes)
    static std::atomic<int> last_on{0};
    const int on = getenv("MVGX_BA_GROUP_DEBUG") ? 1 : 0;
    if (last_on.exchange(on) != on) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_group_debug), &on, sizeof(on));
    static std::atomic<int> last_compact{1};
    const char* env_c = getenv("MVGX_BA_GROUP_COMPACT");
    const int compact_on = env_c ? (atoi(env_c) != 0) : 1;
    if (last_compact.exchange(compact_on) != compact_on) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_group_compact), &compact_on, sizeof(compact_on));
  }
  if (const char* env = getenv("MVGX_BA_MODEL_COST")) c->model_cost_from_jacobian = !strcmp(env, "jacobian");
  if (const char* env = getenv("MVGX_BA_SOLVER")) {
    c->solver_mode = !strcmp(env, "dense") ? 1 : !strcmp(env, "sparse") ? 2 : 0;
    c->solver_from_env = true;   // ("auto" included: the library's rule, whatever the caller's options say)
  }
  if (const char* env = getenv("MVGX_BA_TWO_LEVEL_MIN_N")) c->two_level_min_n = std::max(1, atoi(env));
  if (const char* env = getenv("MVGX_BA_UPDATE128_MIN_TILES")) c->update128_min_tiles = std::max(1, atoi(env));
  const uint64_t no = d.n_obs;
  tick("settings");
  // ---- host-side structure (the analogue of Ceres' preprocessor: ordering, chunks, block structure) ----
  const unsigned T = host_threads(no);
  // observations sorted by point (stable): identity when the caller's list already is (a landmark-by-landmark export of an
  // SfM_Data scene), else one counting sort
  constexpr size_t kGrain = 16384;
  const size_t n_grains = (size_t)((no + kGrain - 1) / kGrain);
#define HA(type, name, n) type* name = nullptr; if ((rc = ha.array(&name, std::max<size_t>((size_t)(n), 1)))) return rc
#define UP(field, vec) if ((rc = dev_upload(c->pool, &d.field, vec, c->stream))) return rc
#define UPN(field, ptr, n) if ((rc = dev_upload_n(c->pool, &d.field, ptr, (size_t)(n), c->stream))) return rc
#define AL(field, n) if ((rc = dev_alloc(c->pool, &d.field, (size_t)(n)))) return rc
  std::vector<uint32_t> pt_start(d.n_pts + 1, 0);
  std::atomic<int> unsorted{0};
  parallel_for_dynamic(n_grains, 1, T, [&](size_t g, unsigned) {
    bool bad = false;
    for (uint64_t k = std::max<uint64_t>(1, g * kGrain), e = std::min<uint64_t>(no, (g + 1) * kGrain); k < e; ++k) bad = bad || p->obs_point[k - 1] > p->obs_point[k];
    if (bad) unsorted.store(1);
  });
  const bool sorted = !unsorted.load();
  std::vector<uint64_t> perm;   // only when the caller's list is not sorted by point
  if (sorted) {
    // CSR starts from the boundaries of the sorted list: observation k opens the points (obs_point[k - 1], obs_point[k]]
    parallel_for_dynamic(n_grains, 1, T, [&](size_t g, unsigned) {
      for (uint64_t k = g * kGrain, e = std::min<uint64_t>(no, (g + 1) * kGrain); k < e; ++k) {
        const uint32_t cur = p->obs_point[k], prev = k ? p->obs_point[k - 1] : UINT32_MAX;
        if (k == 0) { for (uint32_t q = 0; q <= cur; ++q) pt_start[q] = 0; }
        else if (cur != prev) { for (uint32_t q = prev + 1; q <= cur; ++q) pt_start[q] = (uint32_t)k; }
      }
    });
    const uint32_t last = no ? p->obs_point[no - 1] : UINT32_MAX;
    for (uint32_t q = (no ? last + 1 : 0); q <= d.n_pts; ++q) pt_start[q] = (uint32_t)no;
  } else {
    for (uint64_t k = 0; k < no; ++k) pt_start[p->obs_point[k] + 1]++;
    for (uint32_t j = 0; j < d.n_pts; ++j) pt_start[j + 1] += pt_start[j];
    perm.resize(no);
    std::vector<uint32_t> fill(pt_start.begin(), pt_start.end() - 1);
    for (uint64_t k = 0; k < no; ++k) perm[fill[p->obs_point[k]]++] = k;
  }
  // Uploads of the caller's arrays go through page-locked scratch (mvgx::HostArena), copied there by the host threads: from
  // pageable memory hipMemcpyAsync blocks the caller and moved 8 GB/s here (18 ms for the observation list of the 5M-observation
  // scene); from page-locked memory it is an asynchronous DMA at 50 GB/s that overlaps the rest of this function.
  auto staged_upload = [&](auto** dev, const auto* src, size_t n) -> int {
    using E = std::remove_cv_t<std::remove_reference_t<decltype(*src)>>;
    int rc_ = dev_alloc(c->pool, dev, n);
    if (rc_ || !n) return rc_;
    if (n * sizeof(E) < (1u << 20)) { MVGX_HIP(hipMemcpyAsync(*dev, src, n * sizeof(E), hipMemcpyHostToDevice, c->stream)); return MVGX_OK; }
    E* st = nullptr;
    if ((rc_ = ha.array(&st, n))) return rc_;
    const size_t per = (1u << 20) / sizeof(E), n_chunks = (n + per - 1) / per;
    parallel_for_dynamic(n_chunks, 1, T, [&](size_t ch, unsigned) { memcpy(st + ch * per, src + ch * per, std::min(per, n - ch * per) * sizeof(E)); });
    MVGX_HIP(hipMemcpyAsync(*dev, st, n * sizeof(E), hipMemcpyHostToDevice, c->stream));
    return MVGX_OK;
  };
#define UPS(field, ptr, n) if ((rc = staged_upload(&d.field, ptr, (size_t)(n)))) return rc
  UPS(poses, p->poses, (size_t)d.n_poses * 6); UPS(intr, p->intrinsics, (size_t)d.n_intr * 8); UPS(pts, p->points, (size_t)d.n_pts * 3);
  UPS(model, p->intr_model, d.n_intr);
  AL(cposes, d.n_poses * 6); AL(cintr, d.n_intr * 8); AL(cpts, (size_t)d.n_pts * 3);
  // The observation list in point order. A list that arrives sorted is read where it lies (oorig stays null: identity); else it
  // is gathered into the page-locked scratch.
  const uint32_t* opose = p->obs_pose; const uint32_t* ointr = p->obs_intr; const uint32_t* opt_ = p->obs_point;
  const double* oxy = p->obs_xy;
  HA(uint32_t, oslot, no);
  HA(double, oweight, p->obs_weight ? no : 0);
  HA(uint8_t, octrl, p->obs_is_control ? no : 0);
  if (!sorted) {
    HA(uint32_t, g_pose, no); HA(uint32_t, g_intr, no); HA(uint32_t, g_pt, no); HA(uint32_t, g_orig, no); HA(double, g_xy, 2 * no);
    parallel_for_dynamic(n_grains, 1, T, [&](size_t g, unsigned) {
      for (uint64_t k = g * kGrain, e = std::min<uint64_t>(no, (g + 1) * kGrain); k < e; ++k) {
        const uint64_t s_ = perm[k];
        g_pose[k] = p->obs_pose[s_]; g_intr[k] = p->obs_intr[s_]; g_pt[k] = p->obs_point[s_]; g_orig[k] = (uint32_t)s_;
        g_xy[2 * k] = p->obs_xy[2 * s_]; g_xy[2 * k + 1] = p->obs_xy[2 * s_ + 1];
      }
    });
    opose = g_pose; ointr = g_intr; opt_ = g_pt; oxy = g_xy;
    c->h_perm.assign(g_orig, g_orig + no);
    UPN(oorig, g_orig, no);
    UPN(opose, opose, no); UPN(ointr, ointr, no); UPN(opt, opt_, no); UPN(oxy, oxy, 2 * no);
  } else {
    UPS(opose, opose, no); UPS(ointr, ointr, no); UPS(opt, opt_, no); UPS(oxy, oxy, 2 * no);
  }
  std::vector<double> ctrl_count(n_grains, 0.0);
  if (p->obs_weight || p->obs_is_control) {
    parallel_for_dynamic(n_grains, 1, T, [&](size_t g, unsigned) {
      double nc = 0;
      for (uint64_t k = g * kGrain, e = std::min<uint64_t>(no, (g + 1) * kGrain); k < e; ++k) {
        const uint64_t s_ = sorted ? k : perm[k];
        if (p->obs_weight) oweight[k] = p->obs_weight[s_];
        if (p->obs_is_control) { octrl[k] = p->obs_is_control[s_] ? 1 : 0; nc += octrl[k]; }
      }
      ctrl_count[g] = nc;
    });
    if (p->obs_weight) { UPN(oweight, oweight, no); }
    if (p->obs_is_control) { UPN(octrl, octrl, no); }
  }
  double n_rmse = (double)no;
  for (double v : ctrl_count) n_rmse -= v;
  c->n_obs_rmse_local = n_rmse;
  c->n_obs_rmse_all = n_rmse;
  tick("sort by point + gather");
  std::vector<uint8_t> pose_used(d.n_poses, 0), intr_used(d.n_intr, 0), pt_free(d.n_pts, 0);
  parallel_for_dynamic(n_grains, 1, T, [&](size_t g, unsigned) {   // (concurrent stores of the same value 1: relaxed atomic accesses)
    auto set_flag = [](uint8_t& f) { if (!__atomic_load_n(&f, __ATOMIC_RELAXED)) __atomic_store_n(&f, (uint8_t)1, __ATOMIC_RELAXED); };
    for (uint64_t k = g * kGrain, e = std::min<uint64_t>(no, (g + 1) * kGrain); k < e; ++k) {
      set_flag(pose_used[opose[k]]);
      set_flag(intr_used[ointr[k]]);
      set_flag(pt_free[opt_[k]]);
    }
  });
  for (uint32_t k = 0; k < d.n_priors; ++k) pose_used[p->prior_pose[k]] = 1;
  for (uint32_t j = 0; j < d.n_pts; ++j)
    if (p->points_constant || (p->point_const_mask && p->point_const_mask[j])) pt_free[j] = 0;
  tick("  used flags");
  // (point, intrinsic) slots: the distinct intrinsics of a point in order of first appearance; counted, then filled
  std::vector<uint32_t> ptk_start(d.n_pts + 1, 0);
  const size_t n_pgrains = ((size_t)d.n_pts + 4095) / 4096;
  parallel_for_dynamic(n_pgrains, 1, T, [&](size_t g, unsigned) {
    for (uint32_t j = (uint32_t)(g * 4096), e = (uint32_t)std::min<size_t>(d.n_pts, (g + 1) * 4096); j < e; ++j) {
      uint32_t n = 0;
      for (uint32_t o = pt_start[j]; o < pt_start[j + 1]; ++o) {
        bool seen = false;
        for (uint32_t q = pt_start[j]; q < o && !seen; ++q) seen = ointr[q] == ointr[o];
        n += seen ? 0u : 1u;
      }
      ptk_start[j + 1] = n;
    }
  });
  for (uint32_t j = 0; j < d.n_pts; ++j) ptk_start[j + 1] += ptk_start[j];
  std::vector<uint32_t> slot_intr(ptk_start[d.n_pts]), slot_point(ptk_start[d.n_pts]);
  parallel_for_dynamic(n_pgrains, 1, T, [&](size_t g, unsigned) {
    for (uint32_t j = (uint32_t)(g * 4096), e = (uint32_t)std::min<size_t>(d.n_pts, (g + 1) * 4096); j < e; ++j) {
      const uint32_t first = ptk_start[j];
      uint32_t used = first;
      for (uint32_t o = pt_start[j]; o < pt_start[j + 1]; ++o) {
        uint32_t s_ = first;
        while (s_ < used && slot_intr[s_] != ointr[o]) ++s_;
        if (s_ == used) { slot_intr[used] = ointr[o]; slot_point[used] = j; ++used; }
        oslot[o] = s_;
      }
    }
  });
  d.n_islots = (int)slot_intr.size();
  tick("  point-intrinsic slots");
  // observations by pose (ascending observation index inside a pose): the rows of the product lists; then, inside a pose,
  // stably by intrinsic: the (pose, intrinsic) pairs of the Gram kernels, cut into chunks of kPiChunk observations
  std::vector<uint32_t> prow_start;
  HA(uint32_t, pose_obs, no);
  counting_sort_indices(no, d.n_poses, T, [&](uint64_t k) { return opose[k]; }, prow_start, pose_obs);
  tick("  counting sort by pose");
  // (the usual case - every view has one intrinsic - needs no second list)
  std::vector<uint8_t> pose_mixed(d.n_poses, 0);
  std::atomic<int> any_mixed{0};
  parallel_for_dynamic(d.n_poses, 1, T, [&](size_t i, unsigned) {
    bool one = true;
    for (uint32_t q = prow_start[i]; q < prow_start[i + 1] && one; ++q) one = ointr[pose_obs[q]] == ointr[pose_obs[prow_start[i]]];
    if (!one) { pose_mixed[i] = 1; any_mixed.store(1); }
  });
  uint32_t* pi_obs = pose_obs;
  if (any_mixed.load()) {
    HA(uint32_t, pi_sorted, no);
    pi_obs = pi_sorted;
    parallel_for_dynamic(d.n_poses, 1, T, [&](size_t i, unsigned) {
      uint32_t* b_ = pi_obs + prow_start[i]; uint32_t* e_ = pi_obs + prow_start[i + 1];
      std::copy(pose_obs + prow_start[i], pose_obs + prow_start[i + 1], b_);
      if (pose_mixed[i]) std::stable_sort(b_, e_, [&](uint32_t x, uint32_t y) { return ointr[x] < ointr[y]; });
    });
  }
  UPN(pi_obs, pi_obs, no);
  tick("  pi_obs");
  std::vector<uint32_t> pi_start, pi_intr, pose_pi_start(d.n_poses + 1, 0), pichunk_lo, pichunk_hi, pi_chunk0;
  {
    std::vector<std::vector<uint32_t>> per_pose(d.n_poses);   // the first list position of every (pose, intrinsic) pair of the pose
    parallel_for_dynamic(d.n_poses, 4, T, [&](size_t i, unsigned) {
      for (uint32_t q = prow_start[i]; q < prow_start[i + 1]; ++q)
        if (q == prow_start[i] || ointr[pi_obs[q]] != ointr[pi_obs[q - 1]]) per_pose[i].push_back(q);
    });
    for (uint32_t i = 0; i < d.n_poses; ++i) {
      for (uint32_t q : per_pose[i]) { pi_start.push_back(q); pi_intr.push_back(ointr[pi_obs[q]]); }
      pose_pi_start[i + 1] = (uint32_t)pi_start.size();
    }
  }
  d.n_pi = (int)pi_start.size();
  pi_start.push_back((uint32_t)no);
  for (int q = 0; q < d.n_pi; ++q) {
    pi_chunk0.push_back((uint32_t)pichunk_lo.size());
    for (uint32_t lo = pi_start[q]; lo < pi_start[q + 1]; lo += kPiChunk) {
      pichunk_lo.push_back(lo); pichunk_hi.push_back(std::min<uint32_t>(lo + kPiChunk, pi_start[q + 1]));
    }
  }
  pi_chunk0.push_back((uint32_t)pichunk_lo.size());
  d.n_pichunks = (int)pichunk_lo.size();
  // the chunks of every intrinsic (ascending chunk id): its own Gram blocks are summed from them on the matrix-core path
  std::vector<uint32_t> intr_pichunk_start(d.n_intr + 1, 0), intr_pichunk(pichunk_lo.size());
  {
    std::vector<uint32_t> chunk_intr(pichunk_lo.size());
    for (int q = 0; q < d.n_pi; ++q)
      for (uint32_t ch = pi_chunk0[q]; ch < pi_chunk0[q + 1]; ++ch) { chunk_intr[ch] = pi_intr[q]; intr_pichunk_start[pi_intr[q] + 1]++; }
    for (uint32_t k = 0; k < d.n_intr; ++k) intr_pichunk_start[k + 1] += intr_pichunk_start[k];
    std::vector<uint32_t> fill(intr_pichunk_start.begin(), intr_pichunk_start.end() - 1);
    for (uint32_t ch = 0; ch < (uint32_t)chunk_intr.size(); ++ch) intr_pichunk[fill[chunk_intr[ch]]++] = ch;
  }
  tick("  pairs and chunks");
  // the Gram kernel's inputs in (pose, intrinsic) order - the point and the image point of every observation, contiguous - are
  // gathered on the device from the lists uploaded above; the pair of every chunk
  AL(pi_pt, no); AL(pi_xy, no);
  if (no) hipLaunchKernelGGL(ba_gather_lists_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, c->stream, d.pi_obs, (uint32_t)no, d.opt, d.oxy, d.pi_pt, d.pi_xy);
  std::vector<uint32_t> pichunk_pose(pichunk_lo.size()), pichunk_intr(pichunk_lo.size());
  for (uint32_t i = 0; i < d.n_poses; ++i)
    for (uint32_t q = pose_pi_start[i]; q < pose_pi_start[i + 1]; ++q)
      for (uint32_t ch = pi_chunk0[q]; ch < pi_chunk0[q + 1]; ++ch) { pichunk_pose[ch] = i; pichunk_intr[ch] = pi_intr[q]; }
  tick("  chunk pairs");
  // pose-centre priors by pose
  std::vector<uint32_t> prior_pose(p->prior_pose, p->prior_pose + d.n_priors), pose_prior_start(d.n_poses + 1, 0), pose_prior_idx(d.n_priors);
  for (uint32_t k = 0; k < d.n_priors; ++k) pose_prior_start[prior_pose[k] + 1]++;
  for (uint32_t i = 0; i < d.n_poses; ++i) pose_prior_start[i + 1] += pose_prior_start[i];
  { std::vector<uint32_t> fill(pose_prior_start.begin(), pose_prior_start.end() - 1);
    for (uint32_t k = 0; k < d.n_priors; ++k) pose_prior_idx[fill[prior_pose[k]]++] = k; }
  tick("slots, pose / intrinsic lists");
  // Schur products by destination block, generated row by row on host threads. Constant / unused points have Z = 0: only
  // their (a, a) products are listed, so that every diagonal block exists (it carries the Gram block and the rhs).
  TripHost hpp, hpi, hii;
  // Point groups (GroupList): free points with at most kGroupCams distinct poses and kGroupIntr distinct intrinsics, ordered by
  // (lowest pose, highest pose, hash of the pose set); consecutive points join a group while the union of their poses stays within
  // kGroupCams, the union of their intrinsics within kGroupIntr and the group within kGroupPts points. A group that closes because
  // it is full hands its camera set to the next one: groups with the same set form a supergroup (one workgroup, one set of partial
  // blocks), up to kMaxSgGroups of them. Groups of fewer than kGroupMinPts points are dissolved unless they continue a supergroup
  // (their points stay on the record-based path, as do long tracks and constant points). MVGX_BA_GROUPS=0 disables the groups;
  // so does MVGX_BA_MODEL_COST=jacobian (Ceres' form of the model cost reads the Jacobian records of every observation).
  // A supergroup covers up to 2048 observations (8 groups); problems of 4 M observations and more - enough supergroups to fill the device
  // several times either way - up to 4096 (16 groups): fewer prologues, MFMA flushes and partial blocks. MVGX_BA_SG_GROUPS=n sets the
  // limit (sweeps of calls 83 / 84 / 86 with the largest-first launch order: C3 0.625 ms at 8, 0.654 at 12, 0.715 at 16, 0.726 at 6;
  // C5 1.747 ms at 8, 1.729 at 12, 1.714 at 16, 1.754 at 24, 1.914 at 6).
  int kMaxSgGroups = no >= 4000000ull ? 4096 / kGroupThreads : 2048 / kGroupThreads;
  if (const char* env = getenv("MVGX_BA_SG_GROUPS")) kMaxSgGroups = std::max(1, std::min(64, atoi(env)));
  std::vector<uint8_t> in_group(d.n_pts, 0);
  std::vector<uint32_t> sg_start{0}, g_obs_start{0}, g_pt_start{0}, g_pts, g_pt_estart, g_pt_ksplit, sg_cams, sg_intrs;
  uint32_t* g_eobs = nullptr; uint32_t* g_eq = nullptr;   // [entry]: observation, entry word (HostArena)
  size_t n_gentries = 0;
  std::vector<uint8_t> sg_pp, sg_pi, sg_ii;   // per supergroup: which destination blocks exist
  {
    const char* env = getenv("MVGX_BA_GROUPS");
    if (!(env && atoi(env) == 0) && !c->model_cost_from_jacobian && d.n_poses && d.n_pts) {
      std::vector<uint32_t> lo_pose(d.n_pts, UINT32_MAX), hi_pose(d.n_pts, 0);
      std::vector<uint64_t> set_key(d.n_pts, 0);   // (highest pose, order-independent hash of the pose set): equal sets become neighbours
      parallel_for_dynamic(n_pgrains, 1, T, [&](size_t g, unsigned) {
        for (uint32_t j = (uint32_t)(g * 4096), e = (uint32_t)std::min<size_t>(d.n_pts, (g + 1) * 4096); j < e; ++j) {
          const uint32_t o0 = pt_start[j], o1 = pt_start[j + 1];
          bool ok = pt_free[j] && o1 > o0 && o1 - o0 <= (uint32_t)kGroupCams && ptk_start[j + 1] - ptk_start[j] <= (uint32_t)kGroupIntr;
          for (uint32_t o = o0; o < o1 && ok; ++o) {
            for (uint32_t o2 = o0; o2 < o; ++o2) ok = ok && opose[o2] != opose[o];   // a pose seen twice (shared by two views): record path
            lo_pose[j] = std::min(lo_pose[j], opose[o]); hi_pose[j] = std::max(hi_pose[j], opose[o]);
            uint64_t h = (uint64_t)opose[o] * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            set_key[j] += h * 0xBF58476D1CE4E5B9ull;   // commutative combination
          }
          set_key[j] = ((uint64_t)hi_pose[j] << 32) | (uint32_t)(set_key[j] >> 32);
          if (!ok) lo_pose[j] = UINT32_MAX;
          in_group[j] = ok ? 1 : 0;   // (a thread's own range of the array; the few points of dissolved groups are taken out after the sweep:
                                      // marking points from the bucket sweep - every thread storing all over the array - was a third of that phase)
        }
      });
      tick("  point keys");
      std::vector<uint32_t> lo_start, order;   // eligible points by lowest pose (bucket n_poses: not eligible)
      counting_sort_indices(d.n_pts, d.n_poses + 1, T, [&](uint64_t j) { return lo_pose[j] == UINT32_MAX ? d.n_poses : lo_pose[j]; }, lo_start, order);
      parallel_for_dynamic(d.n_poses, 1, T, [&](size_t i, unsigned) {
        std::stable_sort(order.begin() + lo_start[i], order.begin() + lo_start[i + 1], [&](uint32_t x, uint32_t y) { return set_key[x] < set_key[y]; });
      });
      tick("  points by lowest pose, set key");
      // groups never span two lowest-pose buckets, so the buckets are swept independently (host threads) and their groups
      // stitched in bucket order: the result does not depend on the thread count
      struct BucketGroups {
        std::vector<uint32_t> sg_n, obs_n, eobs, eq, pt_n, pts, nk0, cams, intrs, dropped;   // sg_n: groups per supergroup; nk0: per point, its observations with local intrinsic 0; dropped: points of dissolved groups
        std::vector<uint8_t> pp, pi, ii;
      };
      std::vector<BucketGroups> per_bucket(d.n_poses);
      parallel_for_dynamic(d.n_poses, 1, T, [&](size_t bucket, unsigned) {
        BucketGroups& B = per_bucket[bucket];
        {   // (one allocation per list instead of a doubling series)
          size_t n_e = 0;
          const size_t n_p = lo_start[bucket + 1] - lo_start[bucket];
          for (uint32_t q = lo_start[bucket]; q < lo_start[bucket + 1]; ++q) n_e += pt_start[order[q] + 1] - pt_start[order[q]];
          B.eobs.reserve(n_e); B.eq.reserve(n_e); B.pts.reserve(n_p); B.nk0.reserve(n_p);
        }
        std::vector<uint32_t> cams, intrs, mc, mi, cur;
        std::vector<uint32_t> tail_cams, tail_intrs;   // sets of the supergroup under construction (sorted)
        bool sg_open = false;
        uint32_t cur_obs = 0;   // observations of `cur`
        auto local = [](const std::vector<uint32_t>& v, uint32_t id) { return (int)(std::lower_bound(v.begin(), v.end(), id) - v.begin()); };
        // which destination blocks the supergroup under construction has: a point with local poses P and local intrinsics K marks every
        // (x <= y) of P x P, every (x, k) of P x K, every (k <= k') of K x K. Collected as bit rows (a few ORs per point; marking the
        // flags themselves was |P|^2 stores per point and a third of this phase) and written out when the supergroup is complete.
        uint16_t acc_pp[kGroupCams] = {0}, acc_pi[kGroupCams] = {0}, acc_ii[kGroupIntr] = {0};
        static_assert(kGroupCams <= 16 && kGroupIntr <= 16, "bit rows of the destination-block flags");
        bool acc_open = false;
        auto flush_flags = [&]() {
          if (!acc_open) return;
          const size_t sgi = B.sg_n.size() - 1;
          for (int x = 0; x < kGroupCams; ++x) {
            for (int y = x; y < kGroupCams; ++y) if ((acc_pp[x] >> y) & 1u) B.pp[sgi * kGroupPairsPP + x * kGroupCams - x * (x - 1) / 2 + (y - x)] = 1;
            for (int k = 0; k < kGroupIntr; ++k) if ((acc_pi[x] >> k) & 1u) B.pi[sgi * kGroupPairsPI + x * kGroupIntr + k] = 1;
            acc_pp[x] = 0; acc_pi[x] = 0;
          }
          for (int k = 0; k < kGroupIntr; ++k) {
            for (int k2 = k; k2 < kGroupIntr; ++k2) if ((acc_ii[k] >> k2) & 1u) B.ii[sgi * kGroupPairsII + k * kGroupIntr - k * (k - 1) / 2 + (k2 - k)] = 1;
            acc_ii[k] = 0;
          }
          acc_open = false;
        };
        // emits `cur` as a group of the open supergroup (same camera / intrinsic sets) or as the first group of a new one
        auto emit_group = [&](bool continues) {
          if (!continues) {
            flush_flags();   // (of the supergroup before)
            B.sg_n.push_back(0);
            const size_t sgi = B.sg_n.size() - 1;
            B.cams.resize((sgi + 1) * kGroupCams, 0); B.intrs.resize((sgi + 1) * kGroupIntr, 0);
            std::copy(cams.begin(), cams.end(), B.cams.begin() + sgi * kGroupCams);
            std::copy(intrs.begin(), intrs.end(), B.intrs.begin() + sgi * kGroupIntr);
            B.pp.resize((sgi + 1) * kGroupPairsPP, 0); B.pi.resize((sgi + 1) * kGroupPairsPI, 0); B.ii.resize((sgi + 1) * kGroupPairsII, 0);
            tail_cams = cams; tail_intrs = intrs;
          }
          const size_t sgi = B.sg_n.size() - 1;
          B.sg_n[sgi]++;
          uint32_t n_obs_g = 0;
          for (size_t q = 0; q < cur.size(); ++q) {
            const uint32_t j = cur[q];
            B.pts.push_back(j);
            uint8_t xs[kGroupCams], ks[kGroupCams];
            const uint32_t o0 = pt_start[j];
            const int nx = (int)(pt_start[j + 1] - o0);
            uint32_t n_k0 = 0;
            for (int a = 0; a < nx; ++a) {   // local pose / intrinsic of every observation, once
              xs[a] = (uint8_t)local(tail_cams, opose[o0 + a]); ks[a] = (uint8_t)local(tail_intrs, ointr[o0 + a]);
              n_k0 += ks[a] == 0;
            }
            for (int pass = 0; pass < kGroupIntr; ++pass)   // the observations with local intrinsic 0 first, then those with 1
              for (int a = 0; a < nx; ++a) {
                if (ks[a] != pass) continue;
                B.eobs.push_back(o0 + (uint32_t)a);
                B.eq.push_back((uint32_t)q | ((uint32_t)xs[a] << 8) | ((uint32_t)ks[a] << 12));
              }
            n_obs_g += (uint32_t)nx;
            B.nk0.push_back(n_k0);
            uint16_t cm = 0, km = 0;
            for (int a = 0; a < nx; ++a) { cm |= (uint16_t)(1u << xs[a]); km |= (uint16_t)(1u << ks[a]); }
            for (int a = 0; a < nx; ++a) { acc_pp[xs[a]] |= cm; acc_pi[xs[a]] |= km; acc_ii[ks[a]] |= km; }
            acc_open = true;
          }
          (void)sgi;
          B.obs_n.push_back(n_obs_g);
          B.pt_n.push_back((uint32_t)cur.size());
        };
        auto close_group = [&]() {
          if (cams.size() > (size_t)kNarrowCams && !cur.empty()) {
            // a wide group takes the EXACT union of its points' sets (a group inherits the sets of the one before it - that is what lets
            // groups of one camera set form a supergroup - but a wide group never continues one, and its columns follow its poses in use)
            std::vector<uint32_t> ec, ei;
            for (uint32_t j : cur) {
              for (uint32_t o = pt_start[j]; o < pt_start[j + 1]; ++o) { ec.push_back(opose[o]); ei.push_back(ointr[o]); }
            }
            std::sort(ec.begin(), ec.end()); ec.erase(std::unique(ec.begin(), ec.end()), ec.end());
            std::sort(ei.begin(), ei.end()); ei.erase(std::unique(ei.begin(), ei.end()), ei.end());
            cams = ec; intrs = ei;
          }
          const bool continues = sg_open && cams == tail_cams && intrs == tail_intrs && B.sg_n.back() < (uint32_t)kMaxSgGroups;
          // (a wide group stays whatever its size: a handful of points left to the record-based path costs its dozen launches per iteration)
          if (cur.size() >= (size_t)kGroupMinPts || (continues && !cur.empty()) || (cams.size() > (size_t)kNarrowCams && !cur.empty())) { emit_group(continues); sg_open = true; }
          else { sg_open = false; B.dropped.insert(B.dropped.end(), cur.begin(), cur.end()); }
          if (cams.size() > (size_t)kNarrowCams) { cams.clear(); intrs.clear(); }   // (the next group starts from its own points: a wide group's sets are exact, and it continues the supergroup when they come out the same)
          cur.clear(); cur_obs = 0;
        };
        for (uint32_t q = lo_start[bucket]; q < lo_start[bucket + 1]; ++q) {
          const uint32_t j = order[q];
          // (the points of a bucket lie anywhere in the caller's order - a scene flattened from a hash map: their rows are fetched ahead,
          // the CSR entry sixteen points ahead, the rows it names eight points ahead; 1.7 -> ms of this phase at 1 M observations were misses)
          if (q + 16 < lo_start[bucket + 1]) { __builtin_prefetch(&pt_start[order[q + 16]]); __builtin_prefetch(&ptk_start[order[q + 16]]); }
          if (q + 8 < lo_start[bucket + 1]) {
            const uint32_t jn = order[q + 8], on = pt_start[jn];
            __builtin_prefetch(&opose[on]); __builtin_prefetch(&ointr[on]); __builtin_prefetch(&slot_intr[ptk_start[jn]]);
          }
          uint32_t pc[kGroupCams], pk[kGroupCams];
          int npc = 0, npk = 0;
          for (uint32_t o = pt_start[j]; o < pt_start[j + 1]; ++o) pc[npc++] = opose[o];
          for (uint32_t sl = ptk_start[j]; sl < ptk_start[j + 1]; ++sl) pk[npk++] = slot_intr[sl];
          std::sort(pc, pc + npc); std::sort(pk, pk + npk);
          // the two forms (see kNarrowCams): a group is wide once it holds a point of more than kNarrowCams poses - then up to kGroupCams
          // poses and kWidePts points; a group of short tracks never grows wide by union alone (it closes at kNarrowCams as before)
          const bool cur_wide = cams.size() > (size_t)kNarrowCams;
          const size_t max_pts = cur_wide ? (size_t)kWidePts : (size_t)kGroupPts;
          if (cur.size() >= max_pts || cur_obs + (uint32_t)npc > (uint32_t)kGroupThreads) close_group();   // full: the next group starts from the same sets (and may continue the supergroup)
          // (neighbours in the order mostly bring nothing new: the sets stay as they are - no union, no copies)
          const bool inside = std::includes(cams.begin(), cams.end(), pc, pc + npc) && std::includes(intrs.begin(), intrs.end(), pk, pk + npk);
          if (inside && (npc <= kNarrowCams || cur_wide)) { cur.push_back(j); cur_obs += (uint32_t)npc; continue; }
          mc.clear(); mi.clear();
          std::set_union(cams.begin(), cams.end(), pc, pc + npc, std::back_inserter(mc));
          std::set_union(intrs.begin(), intrs.end(), pk, pk + npk, std::back_inserter(mi));
          const bool to_wide = npc > kNarrowCams || cur_wide;
          const size_t cam_limit = to_wide ? (size_t)kGroupCams : (size_t)kNarrowCams;
          if (mc.size() > cam_limit || mi.size() > (size_t)kGroupIntr || (to_wide && !cur_wide && cur.size() >= (size_t)kWidePts)) {
            close_group();
            sg_open = false;
            mc.assign(pc, pc + npc); mi.assign(pk, pk + npk);
          }
          cams = mc; intrs = mi;
          cur.push_back(j); cur_obs += (uint32_t)npc;
        }
        close_group();
        flush_flags();
      });
      for (const BucketGroups& B : per_bucket) for (uint32_t j : B.dropped) in_group[j] = 0;
      tick("  greedy groups per bucket");
      // the buckets' groups stitched in bucket order: offsets from a serial prefix over the buckets, copies on host threads
      const size_t nb = per_bucket.size();
      std::vector<uint32_t> off_sg(nb + 1, 0), off_g(nb + 1, 0), off_e(nb + 1, 0), off_p(nb + 1, 0);
      for (size_t b = 0; b < nb; ++b) {
        const BucketGroups& B = per_bucket[b];
        off_sg[b + 1] = off_sg[b] + (uint32_t)B.sg_n.size(); off_g[b + 1] = off_g[b] + (uint32_t)B.obs_n.size();
        off_e[b + 1] = off_e[b] + (uint32_t)B.eobs.size(); off_p[b + 1] = off_p[b] + (uint32_t)B.pts.size();
      }
      const size_t tsg = off_sg[nb], tg = off_g[nb], te = off_e[nb], tp = off_p[nb];
      n_gentries = te;
      sg_start.resize(tsg + 1); g_obs_start.resize(tg + 1); g_pt_start.resize(tg + 1);
      g_pts.resize(tp); g_pt_estart.resize(tp + 1); g_pt_ksplit.resize(tp);
      sg_cams.resize(tsg * kGroupCams); sg_intrs.resize(tsg * kGroupIntr);
      sg_pp.resize(tsg * kGroupPairsPP); sg_pi.resize(tsg * kGroupPairsPI); sg_ii.resize(tsg * kGroupPairsII);
      if ((rc = ha.array(&g_eobs, std::max<size_t>(te, 1))) || (rc = ha.array(&g_eq, std::max<size_t>(te, 1)))) return rc;
      parallel_for_dynamic(nb, 1, T, [&](size_t b, unsigned) {
        const BucketGroups& B = per_bucket[b];
        uint32_t g = off_g[b];
        for (size_t sgi = 0; sgi < B.sg_n.size(); ++sgi) { g += B.sg_n[sgi]; sg_start[off_sg[b] + sgi + 1] = g; }
        uint32_t e = off_e[b], q = off_p[b];
        for (size_t k = 0; k < B.obs_n.size(); ++k) {
          e += B.obs_n[k]; q += B.pt_n[k];
          g_obs_start[off_g[b] + k + 1] = e; g_pt_start[off_g[b] + k + 1] = q;
        }
        std::copy(B.eobs.begin(), B.eobs.end(), g_eobs + off_e[b]);
        std::copy(B.eq.begin(), B.eq.end(), g_eq + off_e[b]);
        std::copy(B.pts.begin(), B.pts.end(), g_pts.begin() + off_p[b]);
        std::copy(B.cams.begin(), B.cams.end(), sg_cams.begin() + (size_t)off_sg[b] * kGroupCams);
        std::copy(B.intrs.begin(), B.intrs.end(), sg_intrs.begin() + (size_t)off_sg[b] * kGroupIntr);
        std::copy(B.pp.begin(), B.pp.end(), sg_pp.begin() + (size_t)off_sg[b] * kGroupPairsPP);
        std::copy(B.pi.begin(), B.pi.end(), sg_pi.begin() + (size_t)off_sg[b] * kGroupPairsPI);
        std::copy(B.ii.begin(), B.ii.end(), sg_ii.begin() + (size_t)off_sg[b] * kGroupPairsII);
        // first entry of every grouped point (its observations are consecutive entries), and where its local intrinsic 1 starts
        uint32_t ee = off_e[b];
        for (size_t k = 0; k < B.pts.size(); ++k) {
          g_pt_estart[off_p[b] + k] = ee; g_pt_ksplit[off_p[b] + k] = ee + B.nk0[k];
          ee += pt_start[B.pts[k] + 1] - pt_start[B.pts[k]];
        }
      });
      g_pt_estart[tp] = (uint32_t)te;
      tick("  stitch");
      // (the buckets' lists go back to the allocator on the host threads: released one after the other at the end of this block they were 0.5 / 1.9 ms)
      parallel_for_dynamic(nb, 4, T, [&](size_t b, unsigned) { per_bucket[b] = BucketGroups(); });
    }
  }
  const uint32_t n_groups = (uint32_t)g_pt_start.size() - 1, n_sg = (uint32_t)sg_start.size() - 1;
  tick("  entry starts");
  // destination blocks of the supergroups, per product family: (row block, column block) -> partial-block id
  TripExt gext_pp, gext_pi, gext_ii;
  {
    const size_t n_cb = (size_t)d.n_poses + d.n_intr;
    const uint32_t np_ = d.n_poses;
    gext_pp.rows.resize(n_cb); gext_pi.rows.resize(n_cb); gext_ii.rows.resize(n_cb);
    gext_pp.ext_row.assign((size_t)n_sg * kGroupPairsPP, kNoChunk);
    gext_pi.ext_row.assign((size_t)n_sg * kGroupPairsPI, kNoChunk);
    gext_ii.ext_row.assign((size_t)n_sg * kGroupPairsII, kNoChunk);
    for (uint32_t sgi = 0; sgi < n_sg; ++sgi) {
      const uint32_t* cm = sg_cams.data() + (size_t)sgi * kGroupCams;
      const uint32_t* im = sg_intrs.data() + (size_t)sgi * kGroupIntr;
      for (int x = 0, t = 0; x < kGroupCams; ++x)
        for (int y = x; y < kGroupCams; ++y, ++t)
          if (sg_pp[(size_t)sgi * kGroupPairsPP + t]) gext_pp.rows[cm[x]].emplace_back(cm[y], sgi * kGroupPairsPP + t);
      for (int x = 0; x < kGroupCams; ++x)
        for (int k = 0; k < kGroupIntr; ++k)
          if (sg_pi[(size_t)sgi * kGroupPairsPI + x * kGroupIntr + k]) gext_pi.rows[cm[x]].emplace_back(np_ + im[k], sgi * kGroupPairsPI + x * kGroupIntr + k);
      for (int k = 0, t = 0; k < kGroupIntr; ++k)
        for (int l = k; l < kGroupIntr; ++l, ++t)
          if (sg_ii[(size_t)sgi * kGroupPairsII + t]) gext_ii.rows[np_ + im[k]].emplace_back(np_ + im[l], sgi * kGroupPairsII + t);
    }
    tick("  destination rows");
    for (TripExt* e : {&gext_pp, &gext_pi, &gext_ii})
      parallel_for_dynamic(n_cb, 16, T, [&](size_t r, unsigned) { std::sort(e->rows[r].begin(), e->rows[r].end()); });
  }
  tick("point groups");
  // The observations of the points outside every group, ascending (counted per grain, then filled): the record-based path works
  // on these alone. By pose (ascending observation inside a pose) they are the rows of the product lists; their (point, intrinsic)
  // slots by intrinsic the rows of the intrinsic-intrinsic products.
  uint32_t* ungrouped = nullptr;
  size_t n_ungrouped = 0;
  {
    std::vector<uint32_t> cnt(n_grains + 1, 0);
    parallel_for_dynamic(n_grains, 1, T, [&](size_t g, unsigned) {
      uint32_t n = 0;
      for (uint64_t o = g * kGrain, e = std::min<uint64_t>(no, (g + 1) * kGrain); o < e; ++o) n += !in_group[opt_[o]];
      cnt[g + 1] = n;
    });
    for (size_t g = 0; g < n_grains; ++g) cnt[g + 1] += cnt[g];
    n_ungrouped = cnt[n_grains];
    if ((rc = ha.array(&ungrouped, std::max<size_t>(n_ungrouped, 1)))) return rc;
    if (n_ungrouped)
      parallel_for_dynamic(n_grains, 1, T, [&](size_t g, unsigned) {
        uint32_t at = cnt[g];
        if (at == cnt[g + 1]) return;
        for (uint64_t o = g * kGrain, e = std::min<uint64_t>(no, (g + 1) * kGrain); o < e; ++o) if (!in_group[opt_[o]]) ungrouped[at++] = (uint32_t)o;
      });
  }
  std::vector<uint32_t> urow_start, urow;   // [pose] -> positions in `ungrouped`
  counting_sort_indices(n_ungrouped, d.n_poses, T, [&](uint64_t i) { return opose[ungrouped[i]]; }, urow_start, urow);
  std::vector<uint32_t> uslots, islot_start, islot;   // slots of the points outside every group, by intrinsic (ascending slot)
  for (uint32_t q = 0; q < (uint32_t)slot_intr.size(); ++q) if (!in_group[slot_point[q]]) uslots.push_back(q);
  counting_sort_indices(uslots.size(), d.n_intr, T, [&](uint64_t i) { return slot_intr[uslots[i]]; }, islot_start, islot);
  tick("lists of the ungrouped points");
  {
    const size_t n_cb = (size_t)d.n_poses + d.n_intr;
    const uint32_t np = d.n_poses;
    if ((rc = build_trip_list(n_cb, [&](uint32_t r, auto emit) {   // pose-pose: Z_a^T Z_b, pose(a) <= pose(b), same point
          if (r >= np) return;
          for (uint32_t q = urow_start[r]; q < urow_start[r + 1]; ++q) {   // (all products of a grouped point are formed by its group)
            const uint32_t a = ungrouped[urow[q]], j = opt_[a];
            for (uint32_t b = pt_start[j]; b < pt_start[j + 1]; ++b)
              if (r <= opose[b] && (pt_free[j] || a == b)) emit(opose[b], a, b);
          }
        }, hpp, T, &gext_pp))) return rc;
    tick("pose-pose products");
    if ((rc = build_trip_list(n_cb, [&](uint32_t r, auto emit) {   // pose-intrinsic: Z_a^T Zint_slot
          if (r >= np) return;
          for (uint32_t q = urow_start[r]; q < urow_start[r + 1]; ++q) {
            const uint32_t a = ungrouped[urow[q]], j = opt_[a];
            for (uint32_t sl = ptk_start[j]; sl < ptk_start[j + 1]; ++sl)
              if (pt_free[j] || sl == oslot[a]) emit(np + slot_intr[sl], a, sl);
          }
        }, hpi, T, &gext_pi))) return rc;
    for (size_t b = 0; b < hpi.block_row.size(); ++b) {   // the (pose, intrinsic) pair whose Fc^T Fi belongs to the block
      const uint32_t i = hpi.block_row[b], k = hpi.block_col[b] - d.n_poses;
      for (uint32_t q = pose_pi_start[i]; q < pose_pi_start[i + 1]; ++q)
        if (pi_intr[q] == k) hpi.block_own[b] = (int32_t)q;
    }
    if ((rc = build_trip_list(n_cb, [&](uint32_t r, auto emit) {   // intrinsic-intrinsic
          if (r < np) return;
          const uint32_t k = r - np;
          for (uint32_t q = islot_start[k]; q < islot_start[k + 1]; ++q) {
            const uint32_t sa = uslots[islot[q]], j = slot_point[sa];
            for (uint32_t sb = ptk_start[j]; sb < ptk_start[j + 1]; ++sb)
              if (k <= slot_intr[sb] && (pt_free[j] || sa == sb)) emit(np + slot_intr[sb], sa, sb);
          }
        }, hii, T, &gext_ii))) return rc;
  }
  for (const TripHost* h : {&hpp, &hpi, &hii})
    for (size_t b = 0; b < h->block_row.size(); ++b) c->h_blocks.emplace_back(h->block_row[b], h->block_col[b]);
  {   // does every camera block have a diagonal destination in the assemble lists? (then the LM diagonal can be applied there)
    size_t n_diag = 0;
    for (const TripHost* h : {&hpp, &hii})
      for (size_t b = 0; b < h->block_row.size(); ++b) n_diag += h->block_row[b] == h->block_col[b];
    c->diag_blocks_complete = n_diag == (size_t)d.n_poses + d.n_intr;
  }
  tick("pose-intr / intr-intr products");
  // active / counted camera components
  std::vector<uint8_t> cam_active, cam_counts;
  c->h_pose_used = pose_used; c->h_intr_used = intr_used; c->h_pt_free = pt_free;
  camera_component_flags(p, pose_used, intr_used, cam_active, cam_counts);
  std::vector<double> h_pc(p->prior_center, p->prior_center + (size_t)d.n_priors * 3), h_pw(p->prior_weight, p->prior_weight + (size_t)d.n_priors * 3);
  tick("masks, parameter copies");
  UP(pt_start, pt_start); UP(ptk_start, ptk_start); UP(slot_intr, slot_intr); UP(slot_point, slot_point);
  UP(cam_active, cam_active); UP(cam_counts, cam_counts); UP(pt_free, pt_free);
  UP(pichunk_lo, pichunk_lo); UP(pichunk_hi, pichunk_hi); UP(pi_chunk0, pi_chunk0); UP(pose_pi_start, pose_pi_start);
  UP(pichunk_pose, pichunk_pose); UP(pichunk_intr, pichunk_intr);
  UP(intr_pichunk_start, intr_pichunk_start); UP(intr_pichunk, intr_pichunk);
  UP(prior_pose, prior_pose); UP(pose_prior_start, pose_prior_start); UP(pose_prior_idx, pose_prior_idx);
  UP(prior_center, h_pc); UP(prior_weight, h_pw); AL(Jprior, (size_t)d.n_priors * kPriorJ);
  tick("small allocations + uploads");
  // Jacobian records and stored Z blocks: only when some point is on the record-based path (they are indexed by observation:
  // with point groups only the entries of the other points are ever touched)
  const bool record_path = n_ungrouped != 0;
  const uint64_t nrec = record_path ? no : 0;
  AL(JA, (size_t)kJA * nrec); AL(JB, (size_t)kJB * nrec); AL(JC, (size_t)kJC * nrec);
  AL(cn_cam, d.N); AL(g_cam, d.N); AL(scale_cam, d.N); AL(diag_cam, d.N);
  AL(cn_pt, (size_t)d.n_pts * 3); AL(g_pt, (size_t)d.n_pts * 3); AL(scale_pt, (size_t)d.n_pts * 3); AL(diag_pt, (size_t)d.n_pts * 3);
  AL(pichunk_part, (size_t)d.n_pichunks * kPiGram); AL(pi_gram, (size_t)d.n_pi * kPiGram); AL(pose_gram, (size_t)d.n_poses * kPoseGram);
  AL(igram, (size_t)d.n_intr * kIntrGram);
  AL(pichunk_ipart, (size_t)d.n_pichunks * kIntrGram);
  AL(intr_slice_part, (size_t)d.n_intr * kIntrSlices * kIntrGram); AL(intr_arrivals, d.n_intr);
  MVGX_HIP(hipMemsetAsync(d.intr_arrivals, 0, (size_t)std::max<uint32_t>(d.n_intr, 1) * sizeof(unsigned), c->stream));
  AL(Linv3, (size_t)d.n_pts * 6); AL(hp, (size_t)d.n_pts * 3);
  AL(Zpose, (size_t)nrec * 18); AL(Zint, (size_t)(record_path ? d.n_islots : 0) * 24);
  AL(zsol, d.N); AL(step_cam, d.N); AL(step_pt, (size_t)d.n_pts * 3);
  tick("large scratch allocations");
  {
    struct { TripHost* h; TripList* l; int nv; } lists[3] = {{&hpp, &d.tpp, 6 * 6 + 6}, {&hpi, &d.tpi, 6 * 8 + 6}, {&hii, &d.tii, 8 * 8 + 8}};
    for (auto& e : lists) {
      TripHost& h = *e.h; TripList& l = *e.l;
      l.n_trips = (uint32_t)h.trips.size(); l.n_chunks = (uint32_t)h.chunk_lo.size(); l.n_blocks = (uint32_t)h.block_row.size();
      if ((rc = dev_upload(c->pool, &l.trips, h.trips, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &l.chunk_lo, h.chunk_lo, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &l.chunk_hi, h.chunk_hi, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &l.chunk_diag, h.chunk_diag, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &l.block_row, h.block_row, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &l.block_col, h.block_col, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &l.block_chunk0, h.block_chunk0, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &l.block_own, h.block_own, c->stream))) return rc;
      l.n_ext = h.n_ext;
      if (h.n_ext && (rc = dev_upload(c->pool, &l.block_ext0, h.block_ext0, c->stream))) return rc;
      if ((rc = dev_alloc(c->pool, &l.part, ((size_t)l.n_chunks + l.n_ext) * e.nv))) return rc;
    }
    tick("  trip lists");
    d.grp.n_groups = n_groups; d.grp.n_sg = n_sg;
    c->n_grouped_points = (uint32_t)g_pts.size();
    c->all_points_grouped = d.n_pts != 0 && g_pts.size() == (size_t)d.n_pts;
    if (n_sg) {
      d.grp.n_ungrouped = (uint32_t)n_ungrouped;
      if ((rc = dev_upload(c->pool, &d.grp.sg_start, sg_start, c->stream))) return rc;
      c->h_sg_order.resize(n_sg);
      for (uint32_t q = 0; q < (uint32_t)n_sg; ++q) c->h_sg_order[q] = q;
      if (!getenv("MVGX_BA_SG_INDEX_ORDER"))   // (index order: the order of construction, for A/B runs)
        std::stable_sort(c->h_sg_order.begin(), c->h_sg_order.end(), [&](uint32_t a, uint32_t b) {
          return g_obs_start[sg_start[a + 1]] - g_obs_start[sg_start[a]] > g_obs_start[sg_start[b + 1]] - g_obs_start[sg_start[b]];
        });
      // ... the supergroups of the usual form first, the wide ones (a local pose beyond kNarrowCams in use) behind them: two launches
      auto sg_is_wide = [&](uint32_t sgi) {
        for (int x = kNarrowCams; x < kGroupCams; ++x)
          if (sg_pp[(size_t)sgi * kGroupPairsPP + x * kGroupCams - x * (x - 1) / 2]) return true;
        return false;
      };
      std::stable_partition(c->h_sg_order.begin(), c->h_sg_order.end(), [&](uint32_t sgi) { return !sg_is_wide(sgi); });
      c->n_sg_wide = 0;
      for (uint32_t sgi = 0; sgi < (uint32_t)n_sg; ++sgi) c->n_sg_wide += sg_is_wide(sgi) ? 1u : 0u;
      if ((rc = dev_upload(c->pool, &d.grp.sg_order, c->h_sg_order, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &d.grp.obs_start, g_obs_start, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &d.grp.pt_start, g_pt_start, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &d.grp.pts, g_pts, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &d.grp.pt_estart, g_pt_estart, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &d.grp.pt_ksplit, g_pt_ksplit, c->stream))) return rc;
      if ((rc = dev_upload_n(c->pool, &d.grp.eq, g_eq, n_gentries, c->stream))) return rc;
      if ((rc = dev_upload_n(c->pool, &d.grp.eobs, g_eobs, n_gentries, c->stream))) return rc;
      // the image points of the entries: gathered on the device
      if ((rc = dev_alloc(c->pool, &d.grp.exy, n_gentries))) return rc;
      c->n_gentries = n_gentries;
      if (n_gentries)
        hipLaunchKernelGGL(ba_gather_lists_kernel, dim3((unsigned)((n_gentries + 255) / 256)), dim3(256), 0, c->stream, d.grp.eobs, (uint32_t)n_gentries,
                           d.opt, d.oxy, static_cast<uint32_t*>(nullptr), d.grp.exy);
      if ((rc = dev_upload(c->pool, &d.grp.cams, sg_cams, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &d.grp.intrs, sg_intrs, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &d.grp.chunk_pp, gext_pp.ext_row, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &d.grp.chunk_pi, gext_pi.ext_row, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &d.grp.chunk_ii, gext_ii.ext_row, c->stream))) return rc;
      if ((rc = dev_upload_n(c->pool, &d.grp.ungrouped, ungrouped, n_ungrouped, c->stream))) return rc;
      if ((rc = dev_upload(c->pool, &d.pt_grouped, in_group, c->stream))) return rc;
      if ((rc = dev_alloc(c->pool, &d.grp.gmax_part, (size_t)n_sg))) return rc;
      tick("  group uploads");
    }
  }
  c->grid_obs = (int)std::max<uint64_t>(1, (no + 255) / 256);
  c->grid_vec = (int)std::max<size_t>(1, (std::max<size_t>((size_t)d.N, (size_t)d.n_pts * 3) + 255) / 256);
  AL(part, (size_t)6 * std::max(c->grid_obs, c->grid_vec) + 16);   // up to six partial sums per workgroup (ba_step_scalars_kernel)
  AL(grp_part, (size_t)5 * d.grp.n_sg + 8);
  AL(scalars, kSCount + 1);   // the fail word lives in the slot after the scalars: one D2H copy fetches both
  if ((rc = dev_alloc(c->pool, &c->d_accept, (size_t)2))) return rc;
  d.fail = reinterpret_cast<int*>(d.scalars + kSCount);
#undef UP
#undef UPN
#undef UPS
#undef AL
#undef HA
  MVGX_HIP(hipMemsetAsync(d.scalars, 0, kSCount * sizeof(double), c->stream));
  MVGX_HIP(hipMemsetAsync(d.zsol, 0, (size_t)std::max(d.N, 1) * sizeof(double), c->stream));
#define MVGX_GROUP_ATTR(mode, fam, wide_) MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_point_group_kernel<mode, fam, wide_>), hipFuncAttributeMaxDynamicSharedMemorySize, kGroupLds))
  MVGX_GROUP_ATTR(kGroupNorms, false, false); MVGX_GROUP_ATTR(kGroupForward, false, false); MVGX_GROUP_ATTR(kGroupBacksub, false, false);
  MVGX_GROUP_ATTR(kGroupNorms, true, false); MVGX_GROUP_ATTR(kGroupForward, true, false); MVGX_GROUP_ATTR(kGroupBacksub, true, false);
  MVGX_GROUP_ATTR(kGroupNorms, false, true); MVGX_GROUP_ATTR(kGroupForward, false, true); MVGX_GROUP_ATTR(kGroupBacksub, false, true);
  MVGX_GROUP_ATTR(kGroupNorms, true, true); MVGX_GROUP_ATTR(kGroupForward, true, true); MVGX_GROUP_ATTR(kGroupBacksub, true, true);
#undef MVGX_GROUP_ATTR
  c->pinhole_family = true;
  for (uint32_t k = 0; k < d.n_intr; ++k) {
    const int m = p->intr_model[k];
    c->pinhole_family = c->pinhole_family && (m == mvgx_ba::kCamPinhole || m == mvgx_ba::kCamRadial1 || m == mvgx_ba::kCamRadial3 || m == mvgx_ba::kCamBrown);
  }
  if (const char* env = getenv("MVGX_BA_GENERIC_MODELS")) c->pinhole_family = c->pinhole_family && atoi(env) == 0;
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_diag_inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kDiagLds));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sp_factor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kDiagLds));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sp_level_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kDiagLds));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_panel_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPanelLds));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_update128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kUpd128Lds));
  tick("product lists upload (enqueue)");
  MVGX_HIP(hipStreamSynchronize(c->stream));
  tick("stream drain");
  if (!mvgx::ba_create_plan_deferred() && (rc = plan_solver(c, c->h_blocks))) return rc;
  tick("reduced solve: symbolic phase");
  if (timing) fprintf(stderr, "[mvgx_ba_create] %-36s %8.2f ms\n", "total (before the locals are released)", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count());
  guard.c = nullptr;
  *out = c;
  return MVGX_OK;
}

int mvgx_ba_destroy(mvgx_ba_ctx* c) {
  if (!c) return MVGX_OK;
  if (c->multi) {
    mvgx::ba_multi_destroy(c->multi);
    delete c;
    return MVGX_OK;
  }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (getenv("MVGX_BA_FACTOR_DEBUG")) {
    long long st[24];
    if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_factor_stamps), sizeof(st)) == hipSuccess) {
      fprintf(stderr, "[mvgx factor kernel, shader clocks] load %lld | panels (+ overlapped inverse) %lld %lld %lld %lld | inverse tail A %lld | tail B %lld | store %lld | total %lld\n",
              st[1] - st[0], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6], st[8] - st[7], st[9] - st[8], st[9] - st[0]);
      fprintf(stderr, "[mvgx factor kernel, second panel slot, wave 0] row + diagonal-block loads %lld | the sub-panels with their hand-overs %lld | "
              "square roots + write-back %lld | wait for the other waves %lld | rank-16 update + barrier %lld\n",
              st[11] - st[10], st[15] - st[11], st[16] - st[15], st[17] - st[16], st[18] - st[17]);
    }
  }
  if (getenv("MVGX_BA_GROUP_DEBUG")) {
    unsigned long long st[3][8];
    static const char* const mode_name[3] = {"norms", "forward", "back-substitution"};
    if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_group_stamps), sizeof(st)) == hipSuccess)
      for (int m = 0; m < 3; ++m)
        fprintf(stderr, "[mvgx point-group kernel, %s mode, shader clocks of thread 0 summed over workgroups and launches] observation %llu | point sums %llu | "
                "point factors %llu | slots %llu | back-substitution %llu | matrix staging %llu | mfma %llu | partial blocks out %llu\n",
                mode_name[m], st[m][0], st[m][1], st[m][2], st[m][3], st[m][4], st[m][5], st[m][6], st[m][7]);
  }
  c->pool.release();
  if (c->h_scalars) (void)hipHostFree(c->h_scalars);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  mvgx::rccl_destroy(c->rccl);
  mvgx::release_stream(c->device, c->stream);   // (synchronised above)
  delete c;
  return MVGX_OK;
}

// The host workers of the structure build, for the host side of a caller (include/mvgx.h).
int mvgx_host_parallel_for(uint64_t n_items, unsigned max_workers, mvgx_host_item_fn fn, void* user) {
  MVGX_REQUIRE(fn || !n_items, MVGX_ERR_ARG, "mvgx_host_parallel_for: NULL function");
  unsigned threads = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
  if (const char* env = getenv("MVGX_HOST_THREADS")) threads = (unsigned)std::min(64, std::max(1, atoi(env)));
  if (max_workers) threads = std::min(threads, max_workers);
  parallel_for_dynamic((size_t)n_items, 1, threads, [&](size_t i, unsigned tix) { fn(user, (uint64_t)i, tix); });
  return MVGX_OK;
}

// New values for the structure the context was built from (include/mvgx.h). What mvgx_ba_create derives from VALUES is exactly
// what is redone here: the three parameter arrays, the image points (and their two re-ordered copies: (pose, intrinsic) order for
// the camera Gram kernel, entry order for the point groups), weights, prior targets, the camera component flags, the loss scales.
// Everything else in the context is a function of the fingerprinted structure. The LM state is reset by mvgx_ba_solve's start().
static int ba_update_impl(mvgx_ba_ctx* c, const mvgx_ba_problem* p, const uint8_t* obs_enabled);
int mvgx_ba_update(mvgx_ba_ctx* c, const mvgx_ba_problem* p) { return ba_update_impl(c, p, nullptr); }
// ... and with a subset of the observations switched off (include/mvgx.h): the structure stays the context's; an observation that
// is off contributes nothing (Dev::odisabled), a point left without observations stops being a parameter (pt_free 0: scale 0, no
// step, not in |x|), a camera block left without observations leaves the program (camera_component_flags), the RMSE counts the
// observations that are on.
int mvgx_ba_update_subset(mvgx_ba_ctx* c, const mvgx_ba_problem* p, const uint8_t* obs_enabled) { return ba_update_impl(c, p, obs_enabled); }
static int ba_update_impl(mvgx_ba_ctx* c, const mvgx_ba_problem* p, const uint8_t* obs_enabled) {
  MVGX_REQUIRE(c && p, MVGX_ERR_ARG, "mvgx_ba_update: NULL argument");
  { const int vrc = mvgx::ba_validate_problem(p); if (vrc) return vrc; }
  if (c->multi) {
    MVGX_REQUIRE(!obs_enabled, MVGX_ERR_UNSUPPORTED, "mvgx_ba_update_subset: not available on a multi-device context");
    return mvgx::ba_multi_update(c->multi, p);
  }
  if (!(mvgx::ba_fingerprint(p) == c->fingerprint)) {
    set_error("mvgx_ba_update: the problem's structure is not the one this context was created from");
    return MVGX_ERR_STRUCTURE;
  }
  MVGX_HIP(hipSetDevice(c->device));
  Dev& d = c->d;
  const uint64_t no = d.n_obs;
  const unsigned T = host_threads(no);
  mvgx::HostArena ha;   // page-locked sources of the copies; the stream is drained before it goes out of scope
  // ... on EVERY way out: an early return used to free the arena under copies still in flight. A failed update also leaves the
  // context half re-bound: it refuses to solve until an update has gone through (update_incomplete).
  struct Drain {
    mvgx_ba_ctx* c; bool ok = false;
    ~Drain() { (void)hipStreamSynchronize(c->stream); c->update_incomplete = !ok; }
  } drain{c};
  int rc = MVGX_OK;
  auto staged_copy = [&](auto* dev, const auto* src, size_t n) -> int {
    using E = std::remove_cv_t<std::remove_reference_t<decltype(*src)>>;
    if (!n) return MVGX_OK;
    if (n * sizeof(E) < (1u << 20)) { MVGX_HIP(hipMemcpyAsync(dev, src, n * sizeof(E), hipMemcpyHostToDevice, c->stream)); return MVGX_OK; }
    E* st = nullptr;
    const int rc_ = ha.array(&st, n);
    if (rc_) return rc_;
    const size_t per = (1u << 20) / sizeof(E), n_chunks = (n + per - 1) / per;
    parallel_for_dynamic(n_chunks, 1, T, [&](size_t ch, unsigned) { memcpy(st + ch * per, src + ch * per, std::min(per, n - ch * per) * sizeof(E)); });
    MVGX_HIP(hipMemcpyAsync(dev, st, n * sizeof(E), hipMemcpyHostToDevice, c->stream));
    return MVGX_OK;
  };
  if ((rc = staged_copy(d.poses, p->poses, (size_t)d.n_poses * 6)) || (rc = staged_copy(d.intr, p->intrinsics, (size_t)d.n_intr * 8)) ||
      (rc = staged_copy(d.pts, p->points, (size_t)d.n_pts * 3)))
    return rc;
  if (c->h_perm.empty()) {
    if ((rc = staged_copy(d.oxy, p->obs_xy, (size_t)(2 * no)))) return rc;
    if (p->obs_weight && (rc = staged_copy(d.oweight, p->obs_weight, (size_t)no))) return rc;
  } else {   // the caller's list is not in point order: through the permutation mvgx_ba_create found
    double* g_xy = nullptr; double* g_w = nullptr;
    if ((rc = ha.array(&g_xy, (size_t)(2 * no))) || (p->obs_weight && (rc = ha.array(&g_w, (size_t)no)))) return rc;
    constexpr size_t kGrain = 16384;
    parallel_for_dynamic((size_t)((no + kGrain - 1) / kGrain), 1, T, [&](size_t g, unsigned) {
      for (uint64_t k = g * kGrain, e = std::min<uint64_t>(no, (g + 1) * kGrain); k < e; ++k) {
        const uint64_t s_ = c->h_perm[k];
        g_xy[2 * k] = p->obs_xy[2 * s_]; g_xy[2 * k + 1] = p->obs_xy[2 * s_ + 1];
        if (g_w) g_w[k] = p->obs_weight[s_];
      }
    });
    if (no) MVGX_HIP(hipMemcpyAsync(d.oxy, g_xy, (size_t)(2 * no) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (g_w && no) MVGX_HIP(hipMemcpyAsync(d.oweight, g_w, (size_t)no * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  if (d.n_priors) {
    MVGX_HIP(hipMemcpyAsync(d.prior_center, p->prior_center, (size_t)d.n_priors * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    MVGX_HIP(hipMemcpyAsync(d.prior_weight, p->prior_weight, (size_t)d.n_priors * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  // which blocks carry a residual: the structure's, or - with a subset - those of the observations that are on
  std::vector<uint8_t> pose_used_sub, intr_used_sub, pt_free_sub, odis;
  const std::vector<uint8_t>*pose_used = &c->h_pose_used, *intr_used = &c->h_intr_used, *pt_free = &c->h_pt_free;
  double n_rmse = c->n_obs_rmse_all;
  bool any_off = false;
  if (obs_enabled) {
    pose_used_sub.assign(d.n_poses, 0); intr_used_sub.assign(d.n_intr, 0);
    std::vector<uint8_t> pt_seen(d.n_pts, 0);
    odis.assign((size_t)std::max<uint64_t>(no, 1), 0);
    n_rmse = 0;
    {   // (host threads: the flags are bytes that only ever receive 1, the counts are summed per grain)
      constexpr uint64_t kGrain = 1u << 15;
      const size_t n_grains = (size_t)((no + kGrain - 1) / kGrain);
      std::vector<double> cnt(n_grains, 0.0);
      std::vector<uint8_t> off(n_grains, 0);
      parallel_for_dynamic(n_grains, 1, T, [&](size_t g, unsigned) {
        double n = 0; bool any = false;
        for (uint64_t k = g * kGrain, e = std::min<uint64_t>(no, (g + 1) * kGrain); k < e; ++k) {   // k: position in point order; s_: the caller's index
          const uint64_t s_ = c->h_perm.empty() ? k : c->h_perm[k];
          if (obs_enabled[s_]) {
            // (read before write: a few hundred flag bytes stored to by every thread for every observation bounce between the cores;
            // relaxed atomic accesses: several host threads set the same byte - the only value it ever receives is 1)
            auto set_flag = [](uint8_t& f) { if (!__atomic_load_n(&f, __ATOMIC_RELAXED)) __atomic_store_n(&f, (uint8_t)1, __ATOMIC_RELAXED); };
            set_flag(pose_used_sub[p->obs_pose[s_]]);
            set_flag(intr_used_sub[p->obs_intr[s_]]);
            set_flag(pt_seen[p->obs_point[s_]]);
            n += (p->obs_is_control && p->obs_is_control[s_]) ? 0.0 : 1.0;
          } else {
            odis[k] = 1; any = true;
          }
        }
        cnt[g] = n; off[g] = any;
      });
      for (size_t g = 0; g < n_grains; ++g) { n_rmse += cnt[g]; any_off = any_off || off[g]; }
    }
    for (uint32_t k = 0; k < d.n_priors; ++k) pose_used_sub[p->prior_pose[k]] = 1;
    pt_free_sub = c->h_pt_free;
    for (uint32_t j = 0; j < d.n_pts; ++j) pt_free_sub[j] = pt_free_sub[j] && pt_seen[j];
    pose_used = &pose_used_sub; intr_used = &intr_used_sub; pt_free = &pt_free_sub;
  }
  if (any_off) {
    if (!c->odisabled_buf && (rc = dev_alloc(c->pool, &c->odisabled_buf, (size_t)no))) return rc;
    MVGX_HIP(hipMemcpyAsync(c->odisabled_buf, odis.data(), (size_t)no, hipMemcpyHostToDevice, c->stream));
    d.odisabled = c->odisabled_buf;
  } else {
    d.odisabled = nullptr;
  }
  c->n_obs_rmse_local = n_rmse;
  c->n_obs_global = 0;   // (counted again by the next start() / mvgx_ba_evaluate: ADVICE r4 - evaluate() after an update divided by the old count)
  if (d.n_pts) MVGX_HIP(hipMemcpyAsync(d.pt_free, pt_free->data(), (size_t)d.n_pts, hipMemcpyHostToDevice, c->stream));
  std::vector<uint8_t> cam_active, cam_counts;
  camera_component_flags(p, *pose_used, *intr_used, cam_active, cam_counts);
  if (d.N) {
    MVGX_HIP(hipMemcpyAsync(d.cam_active, cam_active.data(), cam_active.size(), hipMemcpyHostToDevice, c->stream));
    MVGX_HIP(hipMemcpyAsync(d.cam_counts, cam_counts.data(), cam_counts.size(), hipMemcpyHostToDevice, c->stream));
  }
  d.huber_a = p->huber_a;
  d.prior_huber_a = p->prior_huber_a;
  // the two re-ordered copies of the image points
  if (no) hipLaunchKernelGGL(ba_gather_lists_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, c->stream, d.pi_obs, (uint32_t)no, d.opt, d.oxy,
                             static_cast<uint32_t*>(nullptr), d.pi_xy);
  if (c->n_gentries)
    hipLaunchKernelGGL(ba_gather_lists_kernel, dim3((unsigned)((c->n_gentries + 255) / 256)), dim3(256), 0, c->stream, d.grp.eobs, (uint32_t)c->n_gentries,
                       d.opt, d.oxy, static_cast<uint32_t*>(nullptr), d.grp.exy);
  BA_LAUNCH_CHECK();
  // as a new context starts: no scalars, no reduced solution, no failure word left from the previous solve
  MVGX_HIP(hipMemsetAsync(d.scalars, 0, (kSCount + 1) * sizeof(double), c->stream));
  MVGX_HIP(hipMemsetAsync(d.zsol, 0, (size_t)std::max(d.N, 1) * sizeof(double), c->stream));
  MVGX_HIP(hipStreamSynchronize(c->stream));
  c->started = false; c->finished = false;
  c->gmax_pending = false; c->gmax_resolved = false; c->fail_clear = false; c->fold_cand_now = false; c->candidate_cost_done = false;
  drain.ok = true;
  return MVGX_OK;
}

// Test hook (not declared in include/mvgx.h): factor-and-invert one kb x kb block (kb <= 64) with the kernel of the dense
// solver. a: column-major 64 x 64, lower triangle read; l_out: 64 x 64 column-major factor; linv_out: 2 x 4096 doubles as the
// kernel stores them ([k][c] = Linv[c][k], then row-major).
int mvgx_debug_factor64(const double* a, int kb, double* l_out, double* linv_out) {
  MVGX_REQUIRE(a && l_out && linv_out && kb > 0 && kb <= 64, MVGX_ERR_ARG, "mvgx_debug_factor64: bad argument");
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_diag_inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kDiagLds));
  double *dA = nullptr, *dLi = nullptr; int* dfail = nullptr; int fail = 0;
  MVGX_HIP(hipMalloc(reinterpret_cast<void**>(&dA), 4096 * sizeof(double)));
  MVGX_HIP(hipMalloc(reinterpret_cast<void**>(&dLi), 8192 * sizeof(double)));
  MVGX_HIP(hipMalloc(reinterpret_cast<void**>(&dfail), sizeof(int)));
  MVGX_HIP(hipMemcpy(dA, a, 4096 * sizeof(double), hipMemcpyHostToDevice));
  MVGX_HIP(hipMemset(dfail, 0, sizeof(int)));
  hipLaunchKernelGGL(chol_diag_inv_kernel, dim3(1), dim3(256), kDiagLds, 0, dA, 64, 0, kb, dLi, dfail);
  MVGX_HIP(hipStreamSynchronize(0));
  MVGX_HIP(hipMemcpy(l_out, dA, 4096 * sizeof(double), hipMemcpyDeviceToHost));
  MVGX_HIP(hipMemcpy(linv_out, dLi, 8192 * sizeof(double), hipMemcpyDeviceToHost));
  MVGX_HIP(hipMemcpy(&fail, dfail, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(dA); (void)hipFree(dLi); (void)hipFree(dfail);
  MVGX_REQUIRE(fail == 0, MVGX_ERR_NUMERIC, "mvgx_debug_factor64: not positive definite");
  return MVGX_OK;
}

int mvgx_ba_comm_init(mvgx_ba_ctx* c, int world, int rank, const void* unique_id128) {
  MVGX_REQUIRE(c && unique_id128, MVGX_ERR_ARG, "mvgx_ba_comm_init: NULL argument");
  MVGX_REQUIRE(!c->multi, MVGX_ERR_STATE, "mvgx_ba_comm_init: a multi-device context has its own exchange");
  MVGX_REQUIRE(!c->started, MVGX_ERR_STATE, "mvgx_ba_comm_init after the solve has started");
  MVGX_HIP(hipSetDevice(c->device));
  mvgx::rccl_destroy(c->rccl);
  c->rccl = nullptr;
  c->plan_ready = false;   // the plan is made on the union of the ranks' blocks
  int rc = mvgx::rccl_init(&c->rccl, world, rank, unique_id128);
  if (rc) return rc;
  if ((rc = mvgx::rccl_self_check(c->rccl, c->stream))) { mvgx::rccl_destroy(c->rccl); c->rccl = nullptr; }
  return rc;
}

int mvgx_ba_set_allreduce(mvgx_ba_ctx* c, mvgx_allreduce_f64 fn, void* user) {
  MVGX_REQUIRE(c, MVGX_ERR_ARG, "mvgx_ba_set_allreduce: NULL context");
  MVGX_REQUIRE(!c->multi, MVGX_ERR_STATE, "mvgx_ba_set_allreduce: a multi-device context has its own exchange");
  c->allreduce = fn;
  c->allreduce_user = user;
  c->plan_ready = false;   // the plan is made on the union of the ranks' blocks
  return MVGX_OK;
}

int mvgx_ba_lm_iteration(mvgx_ba_ctx* c, const mvgx_ba_options* opt, mvgx_ba_summary* summary) {
  MVGX_REQUIRE(c && opt, MVGX_ERR_ARG, "mvgx_ba_lm_iteration: NULL argument");
  if (c->multi) return mvgx::ba_multi_solve(c->multi, opt, summary, true);
  MVGX_REQUIRE(!c->update_incomplete, MVGX_ERR_STATE, "mvgx_ba_lm_iteration: the last mvgx_ba_update of this context failed half way - update it again (or create a new one)");
  MVGX_HIP(hipSetDevice(c->device));
  int rc;
  MVGX_HIP(hipEventRecord(c->ev0, c->stream));
  if (!c->started && (rc = start(c, opt))) return rc;
  if (!c->finished && (rc = lm_iteration(c, opt))) return rc;
  MVGX_HIP(hipEventRecord(c->ev1, c->stream));
  MVGX_HIP(hipEventSynchronize(c->ev1));
  float ms = 0;
  MVGX_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  if ((rc = fill_summary(c, summary))) return rc;
  if (summary) { summary->total_ms = ms; summary->iter_ms_mean = ms; }
  return c->termination == 2 ? MVGX_ERR_NUMERIC : MVGX_OK;
}

int mvgx_ba_solve(mvgx_ba_ctx* c, const mvgx_ba_options* opt, mvgx_ba_summary* summary) {
  MVGX_REQUIRE(c && opt, MVGX_ERR_ARG, "mvgx_ba_solve: NULL argument");
  if (c->multi) return mvgx::ba_multi_solve(c->multi, opt, summary, false);
  MVGX_REQUIRE(!c->update_incomplete, MVGX_ERR_STATE, "mvgx_ba_solve: the last mvgx_ba_update of this context failed half way - update it again (or create a new one)");
  MVGX_HIP(hipSetDevice(c->device));
  int rc;
  MVGX_HIP(hipEventRecord(c->ev0, c->stream));
  c->started = false;
  if ((rc = start(c, opt))) return rc;
  while (!c->finished)
    if ((rc = lm_iteration(c, opt))) return rc;
  MVGX_HIP(hipEventRecord(c->ev1, c->stream));
  MVGX_HIP(hipEventSynchronize(c->ev1));
  float ms = 0;
  MVGX_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  if ((rc = fill_summary(c, summary))) return rc;
  if (summary) { summary->total_ms = ms; summary->iter_ms_mean = c->iteration ? ms / c->iteration : ms; }
  if (c->termination == 2) { set_error("mvgx_ba_solve: %d consecutive invalid steps (linear solve failed or model cost did not decrease)", c->invalid); return MVGX_ERR_NUMERIC; }
  return MVGX_OK;
}

int mvgx_ba_read_params(mvgx_ba_ctx* c, double* poses, double* intrinsics, double* points) {
  MVGX_REQUIRE(c, MVGX_ERR_ARG, "mvgx_ba_read_params: NULL context");
  if (c->multi) return mvgx::ba_multi_read_params(c->multi, poses, intrinsics, points);
  MVGX_HIP(hipSetDevice(c->device));
  Dev& d = c->d;
  if (poses) MVGX_HIP(hipMemcpyAsync(poses, d.poses, (size_t)d.n_poses * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (intrinsics) MVGX_HIP(hipMemcpyAsync(intrinsics, d.intr, (size_t)d.n_intr * 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (points) MVGX_HIP(hipMemcpyAsync(points, d.pts, (size_t)d.n_pts * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MVGX_HIP(hipStreamSynchronize(c->stream));
  return MVGX_OK;
}

static int filter_scratch(mvgx_ba_ctx* c, double** out) {   // 3 n_obs doubles for mvgx_ba_residuals / mvgx_ba_track_angles, on first use
  if (!c->filter_scratch) {
    const int rc = dev_alloc(c->pool, &c->filter_scratch, (size_t)3 * (size_t)c->d.n_obs);
    if (rc) return rc;
  }
  *out = c->filter_scratch;
  return MVGX_OK;
}
int mvgx_ba_residuals(mvgx_ba_ctx* c, double* residual_norm) {
  MVGX_REQUIRE(c && residual_norm, MVGX_ERR_ARG, "mvgx_ba_residuals: NULL argument");
  if (c->multi) return mvgx::ba_multi_residuals(c->multi, residual_norm);
  MVGX_HIP(hipSetDevice(c->device));
  Dev& d = c->d;
  if (!d.n_obs) return MVGX_OK;
  // scratch of its own (3 doubles per observation, allocated at the first call): until round 4 this borrowed the Z array of the
  // record path, which a scene whose points are all grouped allocates EMPTY - the n_obs doubles then ran over the arrays the arena
  // placed behind it (step vectors, then the product lists), harmless only as long as the context was destroyed right after
  double* out = nullptr;
  { const int rc_ = filter_scratch(c, &out); if (rc_) return rc_; }
  hipLaunchKernelGGL(ba_residual_norm_kernel, dim3(c->grid_obs), dim3(256), 0, c->stream, d, d.oorig, out);
  BA_LAUNCH_CHECK();
  MVGX_HIP(hipMemcpyAsync(residual_norm, out, (size_t)d.n_obs * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MVGX_HIP(hipStreamSynchronize(c->stream));
  return MVGX_OK;
}

int mvgx_ba_track_angles(mvgx_ba_ctx* c, double* max_angle_deg) {
  MVGX_REQUIRE(c && max_angle_deg, MVGX_ERR_ARG, "mvgx_ba_track_angles: NULL argument");
  if (c->multi) return mvgx::ba_multi_track_angles(c->multi, max_angle_deg);
  MVGX_HIP(hipSetDevice(c->device));
  Dev& d = c->d;
  if (!d.n_pts) return MVGX_OK;
  if (!d.n_obs) {
    for (uint32_t p = 0; p < d.n_pts; ++p) max_angle_deg[p] = 0.0;
    return MVGX_OK;
  }
  // rays in the context's filter scratch (3 doubles per observation), angles in h_p (3 per point: free between LM iterations)
  double* rays = nullptr;
  { const int rc_ = filter_scratch(c, &rays); if (rc_) return rc_; }
  double* out = d.hp;
  hipLaunchKernelGGL(ba_obs_ray_kernel, dim3(c->grid_obs), dim3(256), 0, c->stream, d, rays);
  BA_LAUNCH_CHECK();
  hipLaunchKernelGGL(ba_track_angle_kernel, dim3((d.n_pts + 255) / 256), dim3(256), 0, c->stream, d, rays, out);
  BA_LAUNCH_CHECK();
  MVGX_HIP(hipMemcpyAsync(max_angle_deg, out, (size_t)d.n_pts * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MVGX_HIP(hipStreamSynchronize(c->stream));
  return MVGX_OK;
}

// Options::linear_solver_type of the caller (sfm_data_BA_ceres.cpp:132-146 default, :483 hand-over; sequential_SfM.cpp:1193-1205 picks
// DENSE_SCHUR / SPARSE_SCHUR by the pose count): both are direct Schur-complement solves, here the storage + factorisation of the
// reduced camera system. Before the first iteration the plan is simply made again; afterwards the call succeeds when the solver in
// place already is what `kind` asks for, else MVGX_ERR_STATE (the device arrays of the other solver were never allocated).
int mvgx_ba_set_linear_solver(mvgx_ba_ctx* c, int kind) {
  MVGX_REQUIRE(c, MVGX_ERR_ARG, "mvgx_ba_set_linear_solver: NULL context");
  MVGX_REQUIRE(kind >= MVGX_BA_LINEAR_SOLVER_AUTO && kind <= MVGX_BA_LINEAR_SOLVER_SPARSE_PREFERRED, MVGX_ERR_ARG, "mvgx_ba_set_linear_solver: kind %d", kind);
  if (c->multi) return mvgx::ba_multi_set_linear_solver(c->multi, kind);
  if (c->solver_from_env || kind == c->solver_mode) return MVGX_OK;
  if (c->solver_ready) {
    const bool sparse_now = c->d.sp.enabled != 0;
    const bool same = (kind == MVGX_BA_LINEAR_SOLVER_DENSE && !sparse_now) || (kind >= MVGX_BA_LINEAR_SOLVER_SPARSE && sparse_now) ||
                      (kind == MVGX_BA_LINEAR_SOLVER_SPARSE_PREFERRED && !sparse_now && c->plan_tried && !c->plan_ok) || c->d.N == 0;
    MVGX_REQUIRE(same, MVGX_ERR_STATE, "mvgx_ba_set_linear_solver(%d) after the first iteration: the %s solver is set up", kind, sparse_now ? "block-sparse" : "dense");
    c->solver_mode = kind;
    return MVGX_OK;
  }
  c->solver_mode = kind;
  c->plan_ready = false;
  if (multi_rank(c) || mvgx::ba_create_plan_deferred()) return MVGX_OK;   // (planned at the first iteration, on the union of the ranks' blocks)
  MVGX_HIP(hipSetDevice(c->device));
  return plan_solver(c, c->h_blocks);
}

int mvgx_ba_get_solver_info(mvgx_ba_ctx* c, mvgx_ba_solver_info* out) {
  MVGX_REQUIRE(c && out, MVGX_ERR_ARG, "mvgx_ba_get_solver_info: NULL argument");
  if (c->multi) return mvgx::ba_multi_solver_info(c->multi, out);
  MVGX_REQUIRE(c->solver_ready, MVGX_ERR_STATE, "mvgx_ba_get_solver_info before the first iteration");
  memset(out, 0, sizeof(*out));
  const int64_t nd = (c->d.N + 63) / 64;
  out->n_columns = c->d.N;
  out->n_dense_tiles = nd * (nd + 1) / 2;
  out->n_point_groups = (int32_t)c->d.grp.n_groups;
  out->n_grouped_points = (int32_t)c->n_grouped_points;
  if (c->d.sp.enabled) {
    const mvgx_sparse::Plan& pl = c->plan;
    out->sparse = 1; out->n_padded = pl.N_pad; out->n_parts = pl.n_parts; out->n_border_blocks = pl.n_border_blocks;
    out->n_levels = pl.n_levels; out->n_factor_tiles = (int64_t)pl.n_fill_tiles; out->flops = pl.flops;
  } else {
    out->n_padded = c->d.N; out->n_factor_tiles = out->n_dense_tiles; out->n_levels = (int32_t)nd;
    out->flops = (double)c->d.N * c->d.N * c->d.N / 3.0 + 2.0 * (double)c->d.N * c->d.N;
  }
  return MVGX_OK;
}

int mvgx_ba_evaluate(mvgx_ba_ctx* c, double* cost, double* rmse) {
  MVGX_REQUIRE(c, MVGX_ERR_ARG, "mvgx_ba_evaluate: NULL context");
  if (c->multi) return mvgx::ba_multi_evaluate(c->multi, cost, rmse);
  MVGX_REQUIRE(!c->update_incomplete, MVGX_ERR_STATE, "mvgx_ba_evaluate: the last mvgx_ba_update of this context failed half way");
  MVGX_HIP(hipSetDevice(c->device));
  int rc = eval<false>(c, c->d.poses, c->d.intr, c->d.pts);
  if (rc) return rc;
  if ((rc = read_scalars(c))) return rc;
  if (cost) *cost = c->h_scalars[kSCost];
  if (c->n_obs_global == 0 && (rc = global_obs_count(c))) return rc;
  if (rmse) *rmse = c->n_obs_global > 0 ? std::sqrt(c->h_scalars[kSSqErr] / (2.0 * c->n_obs_global)) : 0.0;
  return MVGX_OK;
}

}  // extern "C"
