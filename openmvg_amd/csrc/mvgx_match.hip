// libmvgx_hip.so — brute-force L2 2-NN + Lowe ratio matching of 128-D uint8 descriptors on gfx950.
//
// Reference semantics reproduced bit-exactly (paths under /root/reference/src/openMVG):
//   matching/metric.hpp:55-93            L2<uint8_t>: d = sum (a_k - b_k)^2 as int, exact
//   matching/matcher_brute_force.hpp:163-200  all nI distances per query, two smallest (duplicates count)
//   matching/matching_filters.hpp:39-60  keep query iff (float)d0 < ratio_sq * (float)d1   (fp32, one multiply)
//   matching/regions_matcher.hpp:198-204 emit IndMatch(i_in_I, j_in_J) in ascending j
//   matching_image_collection/Matcher_Regions.cpp:95-103  only non-empty pairs are reported
//
// Device formulation. a' = a - 128 (a uint8 -> int8 by flipping bit 7); the distance is translation
// invariant, so d = |a'|^2 + |b'|^2 - 2 a'.b'. The 128-long dot products run on the i8 matrix cores
// (v_mfma_i32_32x32x32_i8, exact int32 accumulation), norms are precomputed per descriptor.
//
// HBM layout (built once per image set by the prep kernels):
//   Rows of an image are PERMUTED into tile slots by the parity of |a'|^2: slot (tile t, row m) with
//   half(m) = (m >> 2) & 1 and j(m) = (m >> 3) * 4 + (m & 3) holds the (16 t + j)-th row of parity half(m)
//   (stable order), so the two lane halves of a 32x32 MFMA result (lane >> 5) see even-norm and odd-norm
//   database rows respectively. Empty slots are pad rows. An image owns ceil(max(n_even, n_odd) / 16) tiles
//   rounded up to a multiple of 8 (one LDS window = 8 tiles = 256 slots).
//   tiles  : int8, one 4 KiB block per 32 slots, "fragment-major": chunk c = s*64 + lane holds the 16
//            bytes lane `lane` feeds to MFMA k-step s (row = lane&31, k-half = lane>>5). A straight 1 KiB
//            coalesced read per wave-instruction therefore IS an MFMA operand; the same image serves as
//            A (database, through LDS) or B (queries, register-resident) operand.
//   rconst : int32 per slot, R = -(|a'|^2 << 8) - (slot index within its 256-slot window); pad slots hold
//            INT_MIN + 512 so they never win                                  (exact kernel, variants 1-3)
//   cinit  : int32 per slot, C = -floor(|a'|^2 / 2); pad slots hold kCPad     (filter kernel, variant 4)
//   qnorm  : int32 per slot, |a'|^2.
//   perm   : uint32 per slot, original row index (kNoMatch for pad slots); rowpos: slot of each original row.
//
// Kernel l2_top2_ratio: one 256-thread workgroup = 4 waves = 512 queries of image J against all of image I.
// Each wave keeps 4 query tiles (128 queries, 64 VGPRs) resident as MFMA B operands for its whole life and
// streams I through a double-buffered 33 KiB LDS window. Per 32x32 distance tile the epilogue costs three
// VALU ops per element:  T = (acc << 9) + R  (v_lshl_add_u32: -(d' << 8) - idx, d' = |a'|^2 - 2 a'.b'),
// T2 = med3(T1, T2, T), T1 = max(T1, T)  — an exact running top-2 of packed (distance, index) keys.
// Windows are folded into full-width (d0', argmin, d1') state; the two half-waves that share a query column
// are merged at the end, |b'|^2 is added, the ratio test is evaluated in fp32 exactly like the reference.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mvgx_common.h"

namespace {

using mvgx::set_error;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kDim = 128;
constexpr int kTileRows = 32;
constexpr int kTileBytes = kTileRows * kDim;  // 4096
constexpr int kWinTiles = 8;                  // tiles per LDS window
constexpr int kWinRows = kWinTiles * kTileRows;  // 256 -> 8 index bits in the packed key
constexpr int kWaves = 4;                     // waves per workgroup
constexpr int kNQ = 4;                        // query tiles per wave (register-resident)
constexpr int kBlockQTiles = kWaves * kNQ;    // 16 tiles = 512 queries per workgroup
constexpr int kStageBytes = kWinTiles * kTileBytes + kWinTiles * kTileRows * 4;  // 32768 + 1024
constexpr int kRPad = INT_MIN + 512;
constexpr int kCPad = -(1 << 27);             // accumulator init of pad slots (real values are > -2^22)
constexpr int kNegInit = -(1 << 28);          // "no value yet" for running maxima; 2 * kNegInit - 1 fits int32
constexpr uint32_t kNoMatch = 0xFFFFFFFFu;
constexpr int kTailTiles = 16;                // slack tiles after the last image (query over-read)

struct MatchParams {
  const int8_t* tiles;
  const int8_t* rows_slot;         // int8 rows, row-major in slot order
  const int* rconst;
  const int* cinit;
  const int* qnorm;
  const uint32_t* perm;            // slot -> original row (kNoMatch: pad)
  const uint32_t* rowpos;          // original row (global numbering) -> slot within its image
  const uint8_t* rows_u8;          // original row-major descriptors (naive check kernel only)
  const uint64_t* img_row_off;     // first row of each image in rows_u8
  const uint32_t* img_tile_off;    // first tile of each image (padded tile counts)
  const uint32_t* img_n;           // descriptors per image
  const uint32_t* img_ntiles;      // occupied tiles per image
  const uint32_t* img_neven;       // rows of even squared norm per image (slots of a parity half are filled in rank order)
  const uint2* pairs;              // batch-local (I, J)
  const uint2* work;               // (batch-local pair index, first query tile)
  const uint4* work8h;             // ... for l2_filter16h_kernel: one record per 8 query tiles (256 query slots)
  const uint4* work8;              // the same items as 32-byte records for l2_filter16_kernel: (pair, first query tile, tileI0, tileJ0), (ntI, ntJpad, 0, 0)
  uint32_t n_work;
  uint32_t* best;                  // [batch pairs][qstride], indexed by query SLOT: original index in I or kNoMatch
  int2* cd;                        // filter -> verify: (d0, upper bound of d1) of a candidate query slot
  uint32_t* count;                 // [batch pairs]: accepted matches
  uint32_t* errflag;               // internal-consistency violations seen by the verify kernel (must stay 0)
  uint32_t qstride;
  float ratio_sq;
};

// ------------------------------------------------------------------------------------------------
// prep 1/3: |a'|^2 of every original row + number of even-norm rows per image
// grid = (ceil(max n / 256), n_images), block = 256, one thread per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norms_kernel(const uint8_t* __restrict__ rows,
                                                        const uint64_t* __restrict__ img_row_off,
                                                        const uint32_t* __restrict__ img_n, int* __restrict__ rownorm,
                                                        uint32_t* __restrict__ n_even) {
  const uint32_t img = blockIdx.y;
  const uint32_t n = img_n[img];
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  bool even = false;
  if (r < n) {
    const uint4* src = reinterpret_cast<const uint4*>(rows + (img_row_off[img] + r) * kDim);
    int nn = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      uint4 v = src[c];
      const uint32_t w[4] = {v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u};
#pragma unroll
      for (int i = 0; i < 4; ++i) nn = __builtin_amdgcn_sdot4((int)w[i], (int)w[i], nn, false);
    }
    rownorm[img_row_off[img] + r] = nn;
    even = (nn & 1) == 0;
  }
  const unsigned long long m = __ballot(even);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&n_even[img], (uint32_t)__popcll(m));
}

// in-tile row of the j-th (0..15) slot of parity half h: matches the 32x32 MFMA result layout, where lane half h
// holds rows (r / 4) * 8 + h * 4 + r % 4 in accumulator register r
__device__ __host__ __forceinline__ int slot_row(int j, int h) { return (j >> 2) * 8 + h * 4 + (j & 3); }

// ------------------------------------------------------------------------------------------------
// prep 2/3: stable partition of an image's rows by norm parity into tile slots; per-slot constants
// grid = n_images, block = 1024 (one workgroup walks the image's rows in order)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void assign_slots_kernel(const uint64_t* __restrict__ img_row_off,
                                                            const uint32_t* __restrict__ img_tile_off,
                                                            const uint32_t* __restrict__ img_n,
                                                            const int* __restrict__ rownorm, uint32_t* __restrict__ rowpos,
                                                            uint32_t* __restrict__ perm, int* __restrict__ rconst,
                                                            int* __restrict__ cinit, int* __restrict__ qnorm) {
  __shared__ uint32_t s_cnt[16][2];
  __shared__ uint32_t s_base[2];
  const uint32_t img = blockIdx.x;
  const uint32_t n = img_n[img];
  const uint64_t row0 = img_row_off[img];
  const size_t slot0 = (size_t)img_tile_off[img] * kTileRows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 2) s_base[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t c0 = 0; c0 < n; c0 += blockDim.x) {
    const uint32_t r = c0 + threadIdx.x;
    const bool valid = r < n;
    const int nn = valid ? rownorm[row0 + r] : 0;
    const int par = nn & 1;
    const unsigned long long m_even = __ballot(valid && par == 0), m_odd = __ballot(valid && par == 1);
    if (lane == 0) { s_cnt[wave][0] = (uint32_t)__popcll(m_even); s_cnt[wave][1] = (uint32_t)__popcll(m_odd); }
    __syncthreads();
    if (valid) {
      uint32_t rank = s_base[par];
      for (int w = 0; w < wave; ++w) rank += s_cnt[w][par];
      rank += (uint32_t)__popcll((par ? m_odd : m_even) & ((1ull << lane) - 1ull));
      const uint32_t t = rank >> 4;
      const uint32_t pos = t * kTileRows + (uint32_t)slot_row((int)(rank & 15u), par);
      rowpos[row0 + r] = pos;
      perm[slot0 + pos] = r;
      rconst[slot0 + pos] = -(nn << 8) - (int)((t % kWinTiles) * kTileRows + (pos & 31u));
      cinit[slot0 + pos] = -(nn >> 1);
      qnorm[slot0 + pos] = nn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t e = 0, o = 0;
      for (int w = 0; w < 16; ++w) { e += s_cnt[w][0]; o += s_cnt[w][1]; }
      s_base[0] += e; s_base[1] += o;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// prep 3/3: gather the permuted rows into fragment-major int8 tiles
// grid = (max padded tiles per image, n_images), block = 256 (one 16-byte chunk per thread)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_tiles_kernel(const uint8_t* __restrict__ rows,
                                                          const uint64_t* __restrict__ img_row_off,
                                                          const uint32_t* __restrict__ img_tile_off,
                                                          const uint32_t* __restrict__ perm, int8_t* __restrict__ tiles,
                                                          int8_t* __restrict__ rows_slot) {
  const uint32_t img = blockIdx.y;
  const uint32_t ntiles_pad = img_tile_off[img + 1] - img_tile_off[img];
  const uint32_t t = blockIdx.x;
  if (t >= ntiles_pad) return;
  const uint32_t gt = img_tile_off[img] + t;
  const int c = threadIdx.x;        // chunk id = s*64 + lane
  const int s = c >> 6, lane = c & 63;
  const int m = lane & 31, h = lane >> 5;
  const uint32_t src_row = perm[(size_t)gt * kTileRows + m];
  uint4 v = make_uint4(0, 0, 0, 0);
  if (src_row != kNoMatch) {
    v = *reinterpret_cast<const uint4*>(rows + (img_row_off[img] + src_row) * kDim + (s * 2 + h) * 16);
    v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
  }
  *reinterpret_cast<uint4*>(tiles + (size_t)gt * kTileBytes + c * 16) = v;
  // the same int8 bytes once more, row-major in slot order (128 contiguous bytes per slot): what the verify stage reads
  *reinterpret_cast<uint4*>(rows_slot + ((size_t)gt * kTileRows + m) * kDim + (s * 2 + h) * 16) = v;
}

__global__ __launch_bounds__(256) void fill_i32_kernel(int* __restrict__ p, size_t n, int value) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = value;
}

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int med3i(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// bijective XCD-aware remap: consecutive work items stay on one XCD (block b runs on XCD b % 8), so the
// workgroups sharing a database image hit the same 4 MiB L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nwg) {
  const uint32_t q = nwg >> 3, r = nwg & 7u;
  const uint32_t xcd = b & 7u, pos = b >> 3;
  const uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + pos;
}

enum StageMode { kStageRegs = 1, kStageGlds = 2, kStageGldsAsm = 3 };

// A window = 8 tiles (32 KiB) + their rconst (1 KiB). Every wave moves a quarter of each: 8 x 1 KiB of tile data
// (16 B/lane) and 256 B of rconst (4 B/lane). No wave-dependent control flow.

// LDS-DMA pieces written as inline asm: hipcc does not count them, so it emits no vmcnt(0) in front of later
// ds_reads of the OTHER buffer; the wait is placed by hand before the window barrier.
// (M0 = wave-uniform LDS byte address; recipe of cdna_hip_programming.md section 5.7.)
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ void glds4_asm(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ void stage_window_glds_asm(char* buf, const int8_t* __restrict__ gtiles,
                                                      const int* __restrict__ grconst, int wave, int lane) {
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)buf;
#pragma unroll
  for (int i = 0; i < kWinTiles; ++i) {
    const int off = i * kTileBytes + wave * 1024;
    glds16_asm(gtiles + off + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + off));
  }
  glds4_asm(grconst + wave * 64 + lane, __builtin_amdgcn_readfirstlane(lds0 + kWinTiles * kTileBytes + wave * 256));
}

// A HALF window = 4 tiles (16 KiB) + their cinit (512 B) for l2_filter16h_kernel: every wave moves a quarter of each tile, waves 0 and 1 half of the
// constants each.
constexpr int kHalfTiles = kWinTiles / 2;
constexpr int kHalfStageBytes = kHalfTiles * kTileBytes + kHalfTiles * kTileRows * 4;  // 16384 + 512
__device__ __forceinline__ void stage_half_glds_asm(char* buf, const int8_t* __restrict__ gtiles, const int* __restrict__ grconst, int wave, int lane) {
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)buf;
#pragma unroll
  for (int i = 0; i < kHalfTiles; ++i) {
    const int off = i * kTileBytes + wave * 1024;
    glds16_asm(gtiles + off + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + off));
  }
  if (wave < 2) glds4_asm(grconst + wave * 64 + lane, __builtin_amdgcn_readfirstlane(lds0 + kHalfTiles * kTileBytes + wave * 256));
}

// LDS-DMA through the builtin (the compiler tracks it, and drains it before any later ds_read it cannot
// prove disjoint — kept as an A/B arm).
__device__ __forceinline__ void stage_window_glds(char* buf, const int8_t* __restrict__ gtiles,
                                                  const int* __restrict__ grconst, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < kWinTiles; ++i) {
    const int off = i * kTileBytes + wave * 1024;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gtiles + off + lane * 16),
                                     (void __attribute__((address_space(3)))*)(buf + off), 16, 0, 0);
  }
  __builtin_amdgcn_global_load_lds(
      (const void __attribute__((address_space(1)))*)(grconst + wave * 64 + lane),
      (void __attribute__((address_space(3)))*)(buf + kWinTiles * kTileBytes + wave * 256), 4, 0, 0);
}

// Through registers, split into issue (before the window's MFMA work) and commit (after it), so the HBM/L2
// latency hides under the compute.
struct StageRegs {
  uint4 v[kWinTiles];
  int r;
};
__device__ __forceinline__ void stage_issue(StageRegs& sr, const int8_t* __restrict__ gtiles,
                                            const int* __restrict__ grconst, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < kWinTiles; ++i)
    sr.v[i] = *reinterpret_cast<const uint4*>(gtiles + i * kTileBytes + wave * 1024 + lane * 16);
  sr.r = grconst[wave * 64 + lane];
}
__device__ __forceinline__ void stage_commit(char* buf, const StageRegs& sr, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < kWinTiles; ++i)
    *reinterpret_cast<uint4*>(buf + i * kTileBytes + wave * 1024 + lane * 16) = sr.v[i];
  *reinterpret_cast<int*>(buf + kWinTiles * kTileBytes + wave * 256 + lane * 4) = sr.r;
}

// ------------------------------------------------------------------------------------------------
// l2_top2_ratio: the hot kernel
// ------------------------------------------------------------------------------------------------
template <int kMode>
__global__ __launch_bounds__(256, 2) void l2_top2_ratio_kernel(MatchParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x kStageBytes

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5;

  const uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const uint2 wk = p.work[w];
  const uint2 ij = p.pairs[wk.x];
  const uint32_t I = ij.x, J = ij.y;
  const uint32_t tileI0 = p.img_tile_off[I], tileJ0 = p.img_tile_off[J];
  const uint32_t ntJpad = p.img_tile_off[J + 1] - tileJ0;
  const int ntI = (int)p.img_ntiles[I];
  const int nwin = (ntI + kWinTiles - 1) / kWinTiles;
  const uint32_t qt0 = wk.y + (uint32_t)wave * kNQ;

  // register-resident query fragments (B operands): 4 tiles x 4 k-steps x 16 B per lane
  v4i b[kNQ][4];
  {
    const int8_t* qsrc = p.tiles + (size_t)(tileJ0 + qt0) * kTileBytes + lane * 16;
#pragma unroll
    for (int n = 0; n < kNQ; ++n)
#pragma unroll
      for (int s = 0; s < 4; ++s) b[n][s] = *reinterpret_cast<const v4i*>(qsrc + n * kTileBytes + s * 1024);
  }

  const int8_t* gI = p.tiles + (size_t)tileI0 * kTileBytes;
  const int* gR = p.rconst + (size_t)tileI0 * kTileRows;

  int G1[kNQ], G2[kNQ], Gi[kNQ];
#pragma unroll
  for (int n = 0; n < kNQ; ++n) { G1[n] = INT_MAX; G2[n] = INT_MAX; Gi[n] = 0; }

  StageRegs sr;
  if constexpr (kMode == kStageGlds) {
    stage_window_glds(smem, gI, gR, wave, lane);
  } else if constexpr (kMode == kStageGldsAsm) {
    stage_window_glds_asm(smem, gI, gR, wave, lane);
  } else {
    stage_issue(sr, gI, gR, wave, lane);
    stage_commit(smem, sr, wave, lane);
  }

  // Pin the query fragments here: the compiler must retire their loads BEFORE the window loop, so no
  // s_waitcnt vmcnt lands inside the tile loop where it would also drain the next window's prefetch.
#pragma unroll
  for (int n = 0; n < kNQ; ++n)
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(b[n][s]));

  for (int win = 0; win < nwin; ++win) {
    // stage `win` has landed for every wave; everybody is done reading the other buffer
    if constexpr (kMode == kStageGldsAsm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* buf = smem + (win & 1) * kStageBytes;
    char* nbuf = smem + ((win + 1) & 1) * kStageBytes;
    // Prefetch the next window. Register mode: the last iteration re-fetches its own window into the idle
    // buffer (no conditional around the loads keeps the staging registers out of scratch). LDS-DMA modes:
    // nothing may be in flight towards LDS when the workgroup retires, so the last iteration issues nothing.
    if constexpr (kMode == kStageRegs) {
      const int wn = (win + 1 < nwin) ? win + 1 : win;
      stage_issue(sr, gI + (size_t)wn * kWinTiles * kTileBytes, gR + wn * kWinRows, wave, lane);
    } else if (win + 1 < nwin) {
      const int8_t* gt = gI + (size_t)(win + 1) * kWinTiles * kTileBytes;
      const int* gr = gR + (win + 1) * kWinRows;
      if constexpr (kMode == kStageGlds) stage_window_glds(nbuf, gt, gr, wave, lane);
      else stage_window_glds_asm(nbuf, gt, gr, wave, lane);
    }

    int T1[kNQ], T2[kNQ];
#pragma unroll
    for (int n = 0; n < kNQ; ++n) { T1[n] = INT_MIN; T2[n] = INT_MIN; }

    const int nt = min(kWinTiles, ntI - win * kWinTiles);
    for (int t = 0; t < nt; ++t) {
      const char* tb = buf + t * kTileBytes + lane * 16;
      v4i a[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const v4i*>(tb + s * 1024);
      const char* rb = buf + kWinTiles * kTileBytes + t * (kTileRows * 4) + h * 16;
      int R[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int4 rv = *reinterpret_cast<const int4*>(rb + g * 32);
        R[g * 4 + 0] = rv.x; R[g * 4 + 1] = rv.y; R[g * 4 + 2] = rv.z; R[g * 4 + 3] = rv.w;
      }
#pragma unroll
      for (int n = 0; n < kNQ; ++n) {
        v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[n][s], acc, 0, 0, 0);
        int t1 = T1[n], t2 = T2[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int T = (int)((unsigned)acc[r] << 9) + R[r];
          t2 = med3i(t1, t2, T);
          t1 = max(t1, T);
        }
        T1[n] = t1; T2[n] = t2;
      }
    }
    if constexpr (kMode == kStageRegs) stage_commit(nbuf, sr, wave, lane);
    // fold the window's packed top-2 into the full-width state
#pragma unroll
    for (int n = 0; n < kNQ; ++n) {
      const int k1 = -T1[n], k2 = -T2[n];  // ascending keys: (d' << 8) | idx8
      const int wd1 = k1 >> 8, widx = (k1 & 255) + win * kWinRows, wd2 = k2 >> 8;
      const bool better = wd1 < G1[n];
      const int g2 = better ? min(G1[n], wd2) : min(G2[n], wd1);
      Gi[n] = better ? widx : Gi[n];
      G1[n] = better ? wd1 : G1[n];
      G2[n] = g2;
    }
  }

  // merge the two half-waves (lane, lane^32 share a query column, disjoint database rows)
#pragma unroll
  for (int n = 0; n < kNQ; ++n) {
    const int P1 = __shfl_xor(G1[n], 32), P2 = __shfl_xor(G2[n], 32), Pi = __shfl_xor(Gi[n], 32);
    const bool better = P1 < G1[n];
    const int g2 = better ? min(G1[n], P2) : min(G2[n], P1);
    Gi[n] = better ? Pi : Gi[n];
    G1[n] = better ? P1 : G1[n];
    G2[n] = g2;
  }

  // ratio test + output (lanes 0..31 own one query slot each per tile); indices go back to original rows
#pragma unroll
  for (int n = 0; n < kNQ; ++n) {
    const uint32_t q = (qt0 + n) * kTileRows + (lane & 31);   // query slot within J
    const bool inb = (lane < 32) && (qt0 + n < ntJpad);
    bool ok = false;
    if (inb) {
      const bool valid = p.perm[(size_t)tileJ0 * kTileRows + q] != kNoMatch;
      const int nq = p.qnorm[(size_t)tileJ0 * kTileRows + q];
      const int d0 = G1[n] + nq, d1 = G2[n] + nq;
      ok = valid && (__int2float_rn(d0) < __fmul_rn(p.ratio_sq, __int2float_rn(d1)));
      p.best[(size_t)wk.x * p.qstride + q] = ok ? p.perm[(size_t)tileI0 * kTileRows + (uint32_t)Gi[n]] : kNoMatch;
    }
    const unsigned long long m = __ballot(ok);
    if (lane == 0 && m) atomicAdd(&p.count[wk.x], (uint32_t)__popcll(m));
  }
}

// ------------------------------------------------------------------------------------------------
// l2_filter (variant 4, stage 1): the same MFMA stream with a ONE-VALU-per-distance epilogue.
//
// The accumulator starts at C_i = -floor(|a'_i|^2 / 2), so a finished 32x32 tile holds v = a'.b' - floor(|a'|^2 / 2)
// and, because lane half h only sees rows of norm parity h,  w = 2 v - h = |b'|^2 - d  exactly (larger = nearer).
// Instead of a running top-2 (3 VALU per distance) each lane keeps running MAXIMA of two partitions of the rows it sees
// (2 x v_max3_i32 per two distances):
//   P-classes: accumulator register pair s = r / 2  (8 per lane)   -> rows at two fixed in-tile positions, all tiles
//   Q-classes: LDS window g (8 tiles)               (folded into a running top-2 over windows with argmax)
// Let x* be the nearest row, in P-class s1 and window g1 of half hw. Every other row is in a different P-class, a
// different window, or the other half, unless it shares (s1, g1, hw) with x* — at most 15 rows. Hence
//   V2 = max(second-best P-class, second-best window, best of the other half)   (all in w units, all exact)
// is the exact nearest distance among the rows OUTSIDE that 16-row cell and d1 <= d1_ub = |b'|^2 - V2.
//   * (float)d0 >= ratio_sq * (float)d1_ub  =>  the reference's test fails for the true d1 <= d1_ub too: REJECT, exactly.
//   * otherwise the query is a CANDIDATE: stage 2 recomputes the 16 rows of the cell exactly and finishes the test.
// Nothing is approximated; ties (equal maxima in two classes) make d1_ub <= d0 and are rejected like the reference does.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// kDbg bits 0..2: timing experiments only (results are wrong): bit 0 drops the epilogue, bit 1 the per-tile LDS fragment loads,
// bit 2 the per-window wait + barrier + staging - what each costs is the difference to kDbg = 0 (tools/filter_breakdown.py).
// kDbg = 8 / 16: earlier forms of the epilogue with valid results (see kFold below).
template <int kMode, int kDbg = 0>
__global__ __launch_bounds__(256, 2) void l2_filter_kernel(MatchParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x kStageBytes

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5;

  const uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const uint2 wk = p.work[w];
  const uint2 ij = p.pairs[wk.x];
  const uint32_t I = ij.x, J = ij.y;
  const uint32_t tileI0 = p.img_tile_off[I], tileJ0 = p.img_tile_off[J];
  const uint32_t ntJpad = p.img_tile_off[J + 1] - tileJ0;
  const int ntI = (int)p.img_ntiles[I];
  const int nwin = (ntI + kWinTiles - 1) / kWinTiles;
  const uint32_t qt0 = wk.y + (uint32_t)wave * kNQ;

  v4i b[kNQ][4];
  {
    const int8_t* qsrc = p.tiles + (size_t)(tileJ0 + qt0) * kTileBytes + lane * 16;
#pragma unroll
    for (int n = 0; n < kNQ; ++n)
#pragma unroll
      for (int s = 0; s < 4; ++s) b[n][s] = *reinterpret_cast<const v4i*>(qsrc + n * kTileBytes + s * 1024);
  }

  const int8_t* gI = p.tiles + (size_t)tileI0 * kTileBytes;
  const int* gC = p.cinit + (size_t)tileI0 * kTileRows;

  // The P-class maxima are kept per window (TW) and folded into the run-long maxima TP and into the window maximum once per
  // window: 8 v_max3 per chain instead of 16, +20 VALU per query tile and window (13.08 -> 12.77 ms per launch in one run,
  // profiles/round2_filter_epilogue_forms_call28.json). TW takes 32 registers (248 of 256 in use).
  // kDbg bit 3 (results VALID) selects the earlier epilogue (both partitions updated per chain); bit 4 single-buffers the
  // accumulator initialiser under the new one (re-fetched behind its last use in a tile: 235 registers, 12.85 ms). Both are
  // kept for comparison and as cross-checks of each other in the tests.
  constexpr bool kFold = (kDbg & 8) == 0;
  constexpr bool kOneCv = kFold && (kDbg & 16) != 0;
  int TP[kNQ][8];                      // P-class maxima
  int TW[kNQ][8];                      // kFold: P-class maxima of the current window
  int Q1[kNQ], Q2[kNQ], Qg[kNQ];       // best / second-best window maximum, best window
#pragma unroll
  for (int n = 0; n < kNQ; ++n) {
#pragma unroll
    for (int s = 0; s < 8; ++s) { TP[n][s] = kNegInit; TW[n][s] = kNegInit; }
    Q1[n] = kNegInit; Q2[n] = kNegInit; Qg[n] = 0;
  }

  StageRegs sr;
  if constexpr (kMode == kStageGlds) {
    stage_window_glds(smem, gI, gC, wave, lane);
  } else if constexpr (kMode == kStageGldsAsm) {
    stage_window_glds_asm(smem, gI, gC, wave, lane);
  } else {
    stage_issue(sr, gI, gC, wave, lane);
    stage_commit(smem, sr, wave, lane);
  }
#pragma unroll
  for (int n = 0; n < kNQ; ++n)
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(b[n][s]));

  for (int win = 0; win < nwin; ++win) {
    if constexpr (!(kDbg & 4) && kMode == kStageGldsAsm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!(kDbg & 4)) __syncthreads();
    char* buf = smem + (win & 1) * kStageBytes;
    char* nbuf = smem + ((win + 1) & 1) * kStageBytes;
    if constexpr (kDbg & 4) {
    } else if constexpr (kMode == kStageRegs) {
      const int wn = (win + 1 < nwin) ? win + 1 : win;
      stage_issue(sr, gI + (size_t)wn * kWinTiles * kTileBytes, gC + wn * kWinRows, wave, lane);
    } else if (win + 1 < nwin) {
      const int8_t* gt = gI + (size_t)(win + 1) * kWinTiles * kTileBytes;
      const int* gc = gC + (win + 1) * kWinRows;
      if constexpr (kMode == kStageGlds) stage_window_glds(nbuf, gt, gc, wave, lane);
      else stage_window_glds_asm(nbuf, gt, gc, wave, lane);
    }

    int TQ[kNQ];
#pragma unroll
    for (int n = 0; n < kNQ; ++n) TQ[n] = kNegInit;

    // Software pipeline over the (tile, query tile) chains of the window: the 4-MFMA chain of step k is issued, then
    // the 16 v_max3 of step k-1 run in its shadow (two accumulator sets ping-pong); the LDS fragments of tile t+1 are
    // fetched while tile t computes. The pipeline is drained at the end of every window (once per 32 chains).
    const int nt = min(kWinTiles, ntI - win * kWinTiles);
    const char* wb = buf + lane * 16;
    const char* wc = buf + kWinTiles * kTileBytes + h * 16;
    v4i a[4];
    v16i cv;
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const v4i*>(wb + s * 1024);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int4 c4 = *reinterpret_cast<const int4*>(wc + g * 32);
      cv[g * 4 + 0] = c4.x; cv[g * 4 + 1] = c4.y; cv[g * 4 + 2] = c4.z; cv[g * 4 + 3] = c4.w;
    }
    v16i accA, accB;   // accB starts as the neutral element of max: its first epilogue is a no-op
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[r] = kNegInit;

#define MVGX_EPILOGUE(ACC, N)                                          \
  if constexpr (kDbg & 1) { asm volatile("" : "+v"(ACC)); TQ[N] = max(TQ[N], ACC[0]); } else \
  if constexpr (kFold) {                                               \
    _Pragma("unroll") for (int s = 0; s < 8; ++s)                      \
      TW[N][s] = max(max(TW[N][s], ACC[2 * s]), ACC[2 * s + 1]);       \
  } else                                                               \
  {                                                                    \
    int tq = TQ[N];                                                    \
    _Pragma("unroll") for (int s = 0; s < 8; ++s) {                    \
      TP[N][s] = max(max(TP[N][s], ACC[2 * s]), ACC[2 * s + 1]);       \
      tq = max(max(tq, ACC[2 * s + 1]), ACC[2 * s]);                   \
    }                                                                  \
    TQ[N] = tq;                                                        \
  }
#define MVGX_CHAIN(ACC, N, A, CV)                                                            \
  ACC = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], b[N][0], CV, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1], b[N][1], ACC, 0, 0, 0);                  \
  ACC = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[2], b[N][2], ACC, 0, 0, 0);                  \
  ACC = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[3], b[N][3], ACC, 0, 0, 0);
#define MVGX_INTERLEAVE()                                                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                         \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                         \
    __builtin_amdgcn_sched_group_barrier(0x2, kFold ? 2 : 4, 0);                             \
  }

#define MVGX_TILE(A, CV, AN, CN, TN)                                                              \
  {                                                                                                \
    const int tn_ = (TN);                                                                          \
    MVGX_CHAIN(accA, 0, A, CV)                                                                     \
    if constexpr (kDbg & 2) { _Pragma("unroll") for (int s = 0; s < 4; ++s) AN[s] = A[s]; CN = CV; } else { \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                  \
        AN[s] = *reinterpret_cast<const v4i*>(wb + tn_ * kTileBytes + s * 1024);                   \
    }                                                                                              \
    MVGX_EPILOGUE(accB, 3)                                                                         \
    MVGX_INTERLEAVE()                                                                              \
    MVGX_CHAIN(accB, 1, A, CV)                                                                     \
    if constexpr (!(kDbg & 2) && !kOneCv) {                                                        \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                \
      const int4 c4 = *reinterpret_cast<const int4*>(wc + tn_ * (kTileRows * 4) + g * 32);        \
      CN[g * 4 + 0] = c4.x; CN[g * 4 + 1] = c4.y; CN[g * 4 + 2] = c4.z; CN[g * 4 + 3] = c4.w;     \
    }                                                                                              \
    }                                                                                              \
    MVGX_EPILOGUE(accA, 0)                                                                         \
    MVGX_INTERLEAVE()                                                                              \
    MVGX_CHAIN(accA, 2, A, CV)                                                                     \
    MVGX_EPILOGUE(accB, 1)                                                                         \
    MVGX_INTERLEAVE()                                                                              \
    if constexpr (kOneCv) {  /* CV is dead after the first MFMA of the tile's last chain: fetch the next tile's */ \
      accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], b[3][0], CV, 0, 0, 0);                   \
      if constexpr (!(kDbg & 2))                                                                   \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                              \
        const int4 c4 = *reinterpret_cast<const int4*>(wc + tn_ * (kTileRows * 4) + g * 32);      \
        CV[g * 4 + 0] = c4.x; CV[g * 4 + 1] = c4.y; CV[g * 4 + 2] = c4.z; CV[g * 4 + 3] = c4.w;   \
      }                                                                                            \
      accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1], b[3][1], accB, 0, 0, 0);                 \
      accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[2], b[3][2], accB, 0, 0, 0);                 \
      accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[3], b[3][3], accB, 0, 0, 0);                 \
    } else {                                                                                       \
      MVGX_CHAIN(accB, 3, A, CV)                                                                   \
    }                                                                                              \
    MVGX_EPILOGUE(accA, 2)                                                                         \
    MVGX_INTERLEAVE()                                                                              \
  }
    // two tiles per trip so that the fragment registers ping-pong (a, cv) <-> (an, cn) without copies; the fetch of a
    // tile past the end of the window re-reads the last tile (no branch around the loads)
    v4i an[4];
    v16i cn;
    int t = 0;
    if constexpr (kOneCv) {
      for (; t + 1 < nt; t += 2) {
        MVGX_TILE(a, cv, an, cv, t + 1)
        MVGX_TILE(an, cv, a, cv, min(t + 2, nt - 1))
      }
      if (t < nt) MVGX_TILE(a, cv, an, cv, t)
    } else {
      for (; t + 1 < nt; t += 2) {
        MVGX_TILE(a, cv, an, cn, t + 1)
        MVGX_TILE(an, cn, a, cv, min(t + 2, nt - 1))
      }
      if (t < nt) MVGX_TILE(a, cv, an, cn, t)
    }
    MVGX_EPILOGUE(accB, 3)   // drain
#undef MVGX_EPILOGUE
#undef MVGX_CHAIN
#undef MVGX_TILE
#undef MVGX_INTERLEAVE
    if constexpr (kMode == kStageRegs) stage_commit(nbuf, sr, wave, lane);
    if constexpr (kFold) {   // the window's class maxima: their maximum is the window maximum, then they join the run-long maxima
#pragma unroll
      for (int n = 0; n < kNQ; ++n) {
        int tq = max(max(TW[n][0], TW[n][1]), TW[n][2]);
        tq = max(max(tq, TW[n][3]), TW[n][4]);
        tq = max(max(tq, TW[n][5]), TW[n][6]);
        TQ[n] = max(tq, TW[n][7]);
#pragma unroll
        for (int s = 0; s < 8; ++s) { TP[n][s] = max(TP[n][s], TW[n][s]); TW[n][s] = kNegInit; }
      }
    }
    // fold the window maximum into the running (best, best window, second best) over windows
#pragma unroll
    for (int n = 0; n < kNQ; ++n) {
      const int v = TQ[n];
      const bool better = v > Q1[n];
      Q2[n] = better ? Q1[n] : max(Q2[n], v);
      Qg[n] = better ? win : Qg[n];
      Q1[n] = better ? v : Q1[n];
    }
  }

#pragma unroll
  for (int n = 0; n < kNQ; ++n) {
    // best / second-best P-class of this lane half
    int p1 = kNegInit, p2 = kNegInit, ps = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int v = TP[n][s];
      const bool better = v > p1;
      p2 = better ? p1 : max(p2, v);
      ps = better ? s : ps;
      p1 = better ? v : p1;
    }
    // exact w = |b'|^2 - d of: the half's best row, its runner-up outside the best P-class, outside the best window
    const int w1 = 2 * p1 - h, w2p = 2 * p2 - h, w2q = 2 * Q2[n] - h;
    const int o1 = __shfl_xor(w1, 32), o2p = __shfl_xor(w2p, 32), o2q = __shfl_xor(w2q, 32);
    const int os = __shfl_xor(ps, 32), og = __shfl_xor(Qg[n], 32);
    const bool mine = w1 > o1;   // parities differ between the halves, so w1 != o1
    const int W1 = mine ? w1 : o1;
    const int V2 = mine ? max3i(w2p, w2q, o1) : max3i(o2p, o2q, w1);
    const int s1 = mine ? ps : os, g1 = mine ? Qg[n] : og, hw = mine ? h : (h ^ 1);

    const uint32_t q = (qt0 + n) * kTileRows + (lane & 31);   // query slot within J
    if (lane < 32 && qt0 + n < ntJpad) {
      const bool valid = p.perm[(size_t)tileJ0 * kTileRows + q] != kNoMatch;
      const int nq = p.qnorm[(size_t)tileJ0 * kTileRows + q];
      const int d0 = nq - W1, d1ub = nq - V2;
      const bool cand = valid && (__int2float_rn(d0) < __fmul_rn(p.ratio_sq, __int2float_rn(d1ub)));
      const size_t o = (size_t)wk.x * p.qstride + q;
      p.best[o] = cand ? ((uint32_t)s1 | ((uint32_t)hw << 3) | ((uint32_t)g1 << 4)) : kNoMatch;
      if (cand) p.cd[o] = make_int2(d0, d1ub);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// l2_filter16 (variant 4, "filter_shape" 16; round 5): the filter on v_mfma_i32_16x16x64_i8.
//
// Why: on REAL descriptor bytes the device's power management holds a pure stream of 32x32x32 i8 MFMAs at 0.725 of the nominal
// 5 POPS and a stream of 16x16x64 at 0.88 (tools/mfma_i8_shapes.hip, profiles/round5_mfma_i8_shapes_call_r5_01.txt; both reach 0.99
// on zero operands): per MAC the 16x16 shape moves a quarter of the accumulator registers through the matrix pipe's result path.
// Same passes per distance, same VALU per distance, same LDS bytes - only the shape of one instruction changes, and with it
//   * the operand fetch: lane l feeds row (l & 15) of a 16-row block and the 16 k-values of quarter (l >> 4) of a 64-long k-step.
//     Those 16 bytes are chunk (2 ks + (l >> 5)) * 64 + ((l >> 4) & 1) * 32 + blk * 16 + (l & 15) of the SAME fragment-major tile the
//     32x32x32 kernels read: no second copy of the descriptors, one ds_read_b128 per lane and k-step as before (16 lanes = 256
//     contiguous bytes: conflict-free);
//   * the result: lane l holds D[4 (l >> 4) + r][l & 15], r = 0..3 - rows 4 g + r of the block for lane group g = l >> 4. In-tile rows
//     m carry norm parity (m >> 2) & 1 (slot_row), so group g sees parity g & 1 only: w = 2 v - (g & 1) is exact as before;
//   * the partition of the rows a lane sees (64 per window instead of 128): P-class c = 2 blk + (r >> 1), 4 classes of 16 rows per
//     window, so a (class, window, lane group) cell is 16 rows: two adjacent slots in each of the window's 8 tiles - the SAME cells
//     as the 32x32 kernel's (s1, half) cells under s1 = 4 (c >> 1) + 2 (g >> 1) + (c & 1), half = g & 1. The code word and the verify
//     stage are unchanged;
//   * a lane owns one query of each of 8 blocks of 16 (not one of each of 4 tiles of 32): 8 x (4 + 4) running maxima, the same 64
//     registers; four lane groups are merged at the end (two exchange rounds) instead of two halves.
// Exactness argument: the one of l2_filter_kernel with "half" read as "lane group" (other groups' best rows bound d1 from their side).
// ------------------------------------------------------------------------------------------------
constexpr int kNB16 = 2 * kNQ;   // query blocks of 16 per wave
__global__ __launch_bounds__(256, 2) void l2_filter16_kernel(MatchParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x kStageBytes

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g4 = lane >> 4, par = g4 & 1;

  const uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const uint4 wr0 = p.work8[2 * (size_t)w], wr1 = p.work8[2 * (size_t)w + 1];   // one 32-byte record: no dependent table walks before the first load of data
  struct { uint32_t x, y; } wk = {wr0.x, wr0.y};
  const uint32_t tileI0 = wr0.z, tileJ0 = wr0.w;
  const uint32_t ntJpad = wr1.y;
  const int ntI = (int)wr1.x;
  const int nwin = (ntI + kWinTiles - 1) / kWinTiles;
  const uint32_t qt0 = wk.y + (uint32_t)wave * kNQ;

  // byte offset of this lane's 16-byte chunk inside a tile, block 0, k-step 0 (see above)
  const int lane_chunk = ((lane >> 5) * 64 + ((lane >> 4) & 1) * 32 + (lane & 15)) * 16;
  v4i b[kNB16][2];
  {
    const int8_t* qsrc = p.tiles + (size_t)(tileJ0 + qt0) * kTileBytes + lane_chunk;
#pragma unroll
    for (int n = 0; n < kNB16; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b[n][ks] = *reinterpret_cast<const v4i*>(qsrc + (n >> 1) * kTileBytes + (n & 1) * 256 + ks * 2048);
  }

  const int8_t* gI = p.tiles + (size_t)tileI0 * kTileBytes;
  const int* gC = p.cinit + (size_t)tileI0 * kTileRows;

  // validity and squared norm of the lane's eight query slots: fetched now, used after the last window (they used to be loaded there,
  // one more trip to memory at the end of every workgroup)
  uint32_t qperm[kNB16];
  int qn[kNB16];
#pragma unroll
  for (int n = 0; n < kNB16; ++n) {
    const bool inb = lane < 16 && qt0 + (uint32_t)(n >> 1) < ntJpad;
    const size_t qs = (size_t)tileJ0 * kTileRows + (qt0 + (uint32_t)(n >> 1)) * kTileRows + (uint32_t)(n & 1) * 16 + (uint32_t)(lane & 15);
    qperm[n] = inb ? p.perm[qs] : kNoMatch;
    qn[n] = inb ? p.qnorm[qs] : 0;
  }

  int TP[kNB16][4];                         // P-class maxima of the run
  int TW[kNB16][4];                         // ... of the current window
  int Q1[kNB16], Q2[kNB16], Qg[kNB16];      // best / second-best window maximum, best window
#pragma unroll
  for (int n = 0; n < kNB16; ++n) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { TP[n][c] = kNegInit; TW[n][c] = kNegInit; }
    Q1[n] = kNegInit; Q2[n] = kNegInit; Qg[n] = 0;
  }

  stage_window_glds_asm(smem, gI, gC, wave, lane);
#pragma unroll
  for (int n = 0; n < kNB16; ++n) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(b[n][ks]));
    asm volatile("" : "+v"(qperm[n]), "+v"(qn[n]));
  }

  for (int win = 0; win < nwin; ++win) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* buf = smem + (win & 1) * kStageBytes;
    char* nbuf = smem + ((win + 1) & 1) * kStageBytes;
    if (win + 1 < nwin)
      stage_window_glds_asm(nbuf, gI + (size_t)(win + 1) * kWinTiles * kTileBytes, gC + (win + 1) * kWinRows, wave, lane);

    const int nt = min(kWinTiles, ntI - win * kWinTiles);
    const char* wb = buf + lane_chunk;
    const char* wc = buf + kWinTiles * kTileBytes + g4 * 16;
    // accB starts as the neutral element of max: its first epilogue is a no-op
    v4i accA[4], accB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accB[i] = v4i{kNegInit, kNegInit, kNegInit, kNegInit};

    // one 16-row block = 16 MFMAs in two groups of four query blocks; the eight v_max3 of the previous group run in the shadow of a group
#define MVGX_GROUP16(ACC, N0, A, CV)                                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                          \
    ACC[i_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], b[(N0) + i_][0], CV, 0, 0, 0);                    \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                          \
    ACC[i_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], b[(N0) + i_][1], ACC[i_], 0, 0, 0);
#define MVGX_EPI16(ACC, N0, BLK)                                                                            \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
    TW[(N0) + i_][2 * (BLK)] = max(max(TW[(N0) + i_][2 * (BLK)], ACC[i_][0]), ACC[i_][1]);                  \
    TW[(N0) + i_][2 * (BLK) + 1] = max(max(TW[(N0) + i_][2 * (BLK) + 1], ACC[i_][2]), ACC[i_][3]);          \
  }
// the schedule of half a block: eight (MFMA, v_max3) pairs with the half's LDS reads (NDS of them) right behind the first pairs - left to
// itself the compiler sinks a fragment load to just in front of its first use and waits there
#define MVGX_MIX16(NDS)                                                                                     \
  _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                        \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                        \
    __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);                                                        \
    if (i_ < (NDS)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
  }
#define MVGX_BLOCK16(A, CV, AN, CN, BLK, NEXT_OFF, NEXT_COFF)                                               \
  {                                                                                                         \
    MVGX_GROUP16(accA, 0, A, CV)                                                                            \
    AN[0] = *reinterpret_cast<const v4i*>(wb + (NEXT_OFF));                                                 \
    AN[1] = *reinterpret_cast<const v4i*>(wb + (NEXT_OFF) + 2048);                                          \
    MVGX_EPI16(accB, 4, 1 - (BLK))                                                                          \
    MVGX_MIX16(2)                                                                                           \
    MVGX_GROUP16(accB, 4, A, CV)                                                                            \
    CN = *reinterpret_cast<const v4i*>(wc + (NEXT_COFF));                                                   \
    MVGX_EPI16(accA, 0, BLK)                                                                                \
    MVGX_MIX16(1)                                                                                           \
  }
    v4i a0[2], a1[2], c0, c1;
    a0[0] = *reinterpret_cast<const v4i*>(wb);
    a0[1] = *reinterpret_cast<const v4i*>(wb + 2048);
    c0 = *reinterpret_cast<const v4i*>(wc);
    // (the first epilogue of a window folds accB = neutral into block class 1 of blocks 4..7: a no-op)
    // (round 6) the wave is given issue priority over its SIMD-mate for the tile loop and hands it back for the window's fold: the mate in
    // its MFMA stream then goes first while this wave is at a fold, a barrier, its start or its merge - 11.80 against 11.89 ms per launch at
    // 2 000 descriptors, 3.53 against 3.57 at 1 000 (calls r6_58 / r6_59; priority over the whole run of windows: half the gain; priority
    // in the OTHER phases instead: a little slower than none)
    __builtin_amdgcn_s_setprio(3);
    for (int t = 0; t < nt; ++t) {
      const int tn = min(t + 1, nt - 1);   // the fetch past the window's last tile re-reads it (no branch around the loads)
      MVGX_BLOCK16(a0, c0, a1, c1, 0, t * kTileBytes + 256, t * (kTileRows * 4) + 64)
      MVGX_BLOCK16(a1, c1, a0, c0, 1, tn * kTileBytes, tn * (kTileRows * 4))
    }
    MVGX_EPI16(accB, 4, 1)   // drain: the second group of the window's last block
    __builtin_amdgcn_s_setprio(0);
#undef MVGX_GROUP16
#undef MVGX_EPI16
#undef MVGX_MIX16
#undef MVGX_BLOCK16
    // the window's class maxima: their maximum is the window maximum, then they join the run-long maxima
#pragma unroll
    for (int n = 0; n < kNB16; ++n) {
      const int v = max(max(max(TW[n][0], TW[n][1]), TW[n][2]), TW[n][3]);
#pragma unroll
      for (int c = 0; c < 4; ++c) { TP[n][c] = max(TP[n][c], TW[n][c]); TW[n][c] = kNegInit; }
      const bool better = v > Q1[n];
      Q2[n] = better ? Q1[n] : max(Q2[n], v);
      Qg[n] = better ? win : Qg[n];
      Q1[n] = better ? v : Q1[n];
    }
  }

#pragma unroll
  for (int n = 0; n < kNB16; ++n) {
    // best / second-best P-class of this lane
    int p1 = kNegInit, p2 = kNegInit, pc = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int v = TP[n][c];
      const bool better = v > p1;
      p2 = better ? p1 : max(p2, v);
      pc = better ? c : pc;
      p1 = better ? v : p1;
    }
    // exact w = |b'|^2 - d of the lane's best row; v2: its runner-up outside the best row's (class, window) cell
    int W1 = 2 * p1 - par, V2 = max(2 * p2 - par, 2 * Q2[n] - par);
    int code = (4 * (pc >> 1) + 2 * (g4 >> 1) + (pc & 1)) | (par << 3) | (Qg[n] << 4);   // (s1, half, window) of the 32x32 kernel's cells
    // merge the four lane groups that share the query column: the winner's runner-up also has to beat the other groups' best rows.
    // (groups g and g ^ 2 have the same parity, so their best values can be equal: then V2 >= W1, d1_ub <= d0 and the query is
    // rejected, as the reference rejects a tie for the first place)
#pragma unroll
    for (int x = 16; x <= 32; x <<= 1) {
      const int o1 = __shfl_xor(W1, x), o2 = __shfl_xor(V2, x), oc = __shfl_xor(code, x);
      const bool mine = W1 > o1;
      V2 = mine ? max(V2, o1) : max(o2, W1);
      code = mine ? code : oc;
      W1 = mine ? W1 : o1;
    }
    const uint32_t q = (qt0 + (uint32_t)(n >> 1)) * kTileRows + (uint32_t)(n & 1) * 16 + (uint32_t)(lane & 15);   // query slot within J
    if (lane < 16 && qt0 + (uint32_t)(n >> 1) < ntJpad) {
      const bool valid = qperm[n] != kNoMatch;
      const int nq = qn[n];
      const int d0 = nq - W1, d1ub = nq - V2;
      const bool cand = valid && (__int2float_rn(d0) < __fmul_rn(p.ratio_sq, __int2float_rn(d1ub)));
      const size_t o = (size_t)wk.x * p.qstride + q;
      p.best[o] = cand ? (uint32_t)code : kNoMatch;
      if (cand) p.cd[o] = make_int2(d0, d1ub);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// l2_filter16h (variant 4, "filter_shape" 17; round 5): l2_filter16_kernel for THREE workgroups per CU. At two workgroups per CU (236 VGPRs, 66 KiB)
// the matrix pipe is 74.5 % busy: a workgroup lives 49 us at 2 000 descriptors of which ~12 are not MFMA stream (start: record -> query fragments +
// first window; eight window barriers; the four-group merge), and with two waves per SIMD nothing runs beside them. Here a wave keeps TWO query
// tiles (4 blocks of 16: 32 VGPRs of fragments, half the running maxima) and the database streams through HALF windows of 4 tiles (2 x 16.5 KiB of
// LDS per workgroup): <= 168 VGPRs and 33 KiB, three workgroups per CU. The Q-class of the partition stays the 8-tile window (the maxima are
// folded every second half window), so cells, code word and verify stage are those of the other filter kernels. A workgroup owns 256 query slots:
// its work records step by 8 query tiles (work8h), the verify stage keeps its 512-slot items.
// ------------------------------------------------------------------------------------------------
constexpr int kNQh = 2;                 // query tiles per wave
constexpr int kNBh = 2 * kNQh;          // query blocks of 16 per wave
constexpr int kBlockQTilesH = kWaves * kNQh;   // 8 tiles = 256 queries per workgroup
__global__ __launch_bounds__(256, 3) void l2_filter16h_kernel(MatchParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x kHalfStageBytes

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g4 = lane >> 4, par = g4 & 1;

  const uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const uint4 wr0 = p.work8h[2 * (size_t)w], wr1 = p.work8h[2 * (size_t)w + 1];
  const uint32_t pair = wr0.x;
  const uint32_t tileI0 = wr0.z, tileJ0 = wr0.w;
  const uint32_t ntJpad = wr1.y;
  const int ntI = (int)wr1.x;
  const int nhalf = (ntI + kHalfTiles - 1) / kHalfTiles;
  const uint32_t qt0 = wr0.y + (uint32_t)wave * kNQh;

  const int lane_chunk = ((lane >> 5) * 64 + ((lane >> 4) & 1) * 32 + (lane & 15)) * 16;
  v4i b[kNBh][2];
  {
    const int8_t* qsrc = p.tiles + (size_t)(tileJ0 + qt0) * kTileBytes + lane_chunk;
#pragma unroll
    for (int n = 0; n < kNBh; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b[n][ks] = *reinterpret_cast<const v4i*>(qsrc + (n >> 1) * kTileBytes + (n & 1) * 256 + ks * 2048);
  }
  const int8_t* gI = p.tiles + (size_t)tileI0 * kTileBytes;
  const int* gC = p.cinit + (size_t)tileI0 * kTileRows;

  uint32_t qperm[kNBh];
  int qn[kNBh];
#pragma unroll
  for (int n = 0; n < kNBh; ++n) {
    const bool inb = lane < 16 && qt0 + (uint32_t)(n >> 1) < ntJpad;
    const size_t qs = (size_t)tileJ0 * kTileRows + (qt0 + (uint32_t)(n >> 1)) * kTileRows + (uint32_t)(n & 1) * 16 + (uint32_t)(lane & 15);
    qperm[n] = inb ? p.perm[qs] : kNoMatch;
    qn[n] = inb ? p.qnorm[qs] : 0;
  }

  int TP[kNBh][4], TW[kNBh][4], Q1[kNBh], Q2[kNBh], Qg[kNBh];
#pragma unroll
  for (int n = 0; n < kNBh; ++n) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { TP[n][c] = kNegInit; TW[n][c] = kNegInit; }
    Q1[n] = kNegInit; Q2[n] = kNegInit; Qg[n] = 0;
  }

  stage_half_glds_asm(smem, gI, gC, wave, lane);
#pragma unroll
  for (int n = 0; n < kNBh; ++n) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(b[n][ks]));
    asm volatile("" : "+v"(qperm[n]), "+v"(qn[n]));
  }

  for (int hw = 0; hw < nhalf; ++hw) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* buf = smem + (hw & 1) * kHalfStageBytes;
    char* nbuf = smem + ((hw + 1) & 1) * kHalfStageBytes;
    if (hw + 1 < nhalf)
      stage_half_glds_asm(nbuf, gI + (size_t)(hw + 1) * kHalfTiles * kTileBytes, gC + (hw + 1) * kHalfTiles * kTileRows, wave, lane);

    const int nt = min(kHalfTiles, ntI - hw * kHalfTiles);
    const char* wb = buf + lane_chunk;
    const char* wc = buf + kHalfTiles * kTileBytes + g4 * 16;
    // two accumulator sets of four query blocks ping-pong over the 16-row blocks: the eight MFMAs of a block are issued, the eight v_max3 of
    // the block before run in their shadow. accB starts as the neutral element of max: its first epilogue is a no-op
    v4i accA[4], accB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accB[i] = v4i{kNegInit, kNegInit, kNegInit, kNegInit};
#define MVGX_GROUPH(ACC, A, CV)                                                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                          \
    ACC[i_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], b[i_][0], CV, 0, 0, 0);                           \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                          \
    ACC[i_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], b[i_][1], ACC[i_], 0, 0, 0);
#define MVGX_EPIH(ACC, BLK)                                                                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
    TW[i_][2 * (BLK)] = max(max(TW[i_][2 * (BLK)], ACC[i_][0]), ACC[i_][1]);                                \
    TW[i_][2 * (BLK) + 1] = max(max(TW[i_][2 * (BLK) + 1], ACC[i_][2]), ACC[i_][3]);                        \
  }
#define MVGX_MIXH(NDS)                                                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                        \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                        \
    __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);                                                        \
    if (i_ < (NDS)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
  }
    v4i a0[2], a1[2], c0, c1;
    a0[0] = *reinterpret_cast<const v4i*>(wb);
    a0[1] = *reinterpret_cast<const v4i*>(wb + 2048);
    c0 = *reinterpret_cast<const v4i*>(wc);
    for (int t = 0; t < nt; ++t) {
      const int tn = min(t + 1, nt - 1);   // the fetch past the last tile re-reads it (no branch around the loads)
      // block 0 of tile t (accA); the maxima of the block before (block 1, accB) folded beside it
      MVGX_GROUPH(accA, a0, c0)
      a1[0] = *reinterpret_cast<const v4i*>(wb + t * kTileBytes + 256);
      a1[1] = *reinterpret_cast<const v4i*>(wb + t * kTileBytes + 256 + 2048);
      c1 = *reinterpret_cast<const v4i*>(wc + t * (kTileRows * 4) + 64);
      MVGX_EPIH(accB, 1)
      MVGX_MIXH(3)
      // block 1 of tile t (accB); block 0's maxima folded
      MVGX_GROUPH(accB, a1, c1)
      a0[0] = *reinterpret_cast<const v4i*>(wb + tn * kTileBytes);
      a0[1] = *reinterpret_cast<const v4i*>(wb + tn * kTileBytes + 2048);
      c0 = *reinterpret_cast<const v4i*>(wc + tn * (kTileRows * 4));
      MVGX_EPIH(accA, 0)
      MVGX_MIXH(3)
    }
    MVGX_EPIH(accB, 1)   // drain: block 1 of the half window's last tile
#undef MVGX_GROUPH
#undef MVGX_EPIH
#undef MVGX_MIXH
    if ((hw & 1) || hw + 1 == nhalf) {   // (uniform) an 8-tile window is complete: its class maxima join the run-long ones, their maximum is the window maximum
      const int win = hw >> 1;
#pragma unroll
      for (int n = 0; n < kNBh; ++n) {
        const int v = max(max(max(TW[n][0], TW[n][1]), TW[n][2]), TW[n][3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) { TP[n][c] = max(TP[n][c], TW[n][c]); TW[n][c] = kNegInit; }
        const bool better = v > Q1[n];
        Q2[n] = better ? Q1[n] : max(Q2[n], v);
        Qg[n] = better ? win : Qg[n];
        Q1[n] = better ? v : Q1[n];
      }
    }
  }

#pragma unroll
  for (int n = 0; n < kNBh; ++n) {
    int p1 = kNegInit, p2 = kNegInit, pc = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int v = TP[n][c];
      const bool better = v > p1;
      p2 = better ? p1 : max(p2, v);
      pc = better ? c : pc;
      p1 = better ? v : p1;
    }
    int W1 = 2 * p1 - par, V2 = max(2 * p2 - par, 2 * Q2[n] - par);
    int code = (4 * (pc >> 1) + 2 * (g4 >> 1) + (pc & 1)) | (par << 3) | (Qg[n] << 4);
#pragma unroll
    for (int x = 16; x <= 32; x <<= 1) {
      const int o1 = __shfl_xor(W1, x), o2 = __shfl_xor(V2, x), oc = __shfl_xor(code, x);
      const bool mine = W1 > o1;
      V2 = mine ? max(V2, o1) : max(o2, W1);
      code = mine ? code : oc;
      W1 = mine ? W1 : o1;
    }
    const uint32_t q = (qt0 + (uint32_t)(n >> 1)) * kTileRows + (uint32_t)(n & 1) * 16 + (uint32_t)(lane & 15);
    if (lane < 16 && qt0 + (uint32_t)(n >> 1) < ntJpad) {
      const bool valid = qperm[n] != kNoMatch;
      const int nq = qn[n];
      const int d0 = nq - W1, d1ub = nq - V2;
      const bool cand = valid && (__int2float_rn(d0) < __fmul_rn(p.ratio_sq, __int2float_rn(d1ub)));
      const size_t o = (size_t)pair * p.qstride + q;
      p.best[o] = cand ? (uint32_t)code : kNoMatch;
      if (cand) p.cd[o] = make_int2(d0, d1ub);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// l2_verify (variant 4, stage 2): finishes the candidates of the filter. Same work items as the filter (a workgroup
// owns 512 query slots of one pair). 16 lanes per candidate recompute the exact distances of the 16 slots of its
// (P-class, window, half) cell with v_dot4_i32_i8 on the tile bytes, take the cell's best (must equal d0) and its
// runner-up, d1 = min(d1_ub, runner-up), and evaluate the reference's fp32 ratio test. best[] receives the ORIGINAL
// index in I or kNoMatch; count[] the accepted queries of the pair.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_verify_kernel(MatchParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const uint2 wk = p.work[blockIdx.x];
  const uint2 ij = p.pairs[wk.x];
  const uint32_t I = ij.x, J = ij.y;
  const uint32_t tileI0 = p.img_tile_off[I], tileJ0 = p.img_tile_off[J];
  const uint32_t ntJpad = p.img_tile_off[J + 1] - tileJ0;
  const uint32_t nEvenI = p.img_neven[I], nOddI = p.img_n[I] - nEvenI;
  const int l16 = lane & 15, sub = lane >> 4;
  uint32_t accepted = 0;
  for (int part = 0; part < 2; ++part) {
    const uint32_t q0 = (wk.y + (uint32_t)wave * kNQ) * kTileRows + (uint32_t)part * 64;
    const uint32_t q = q0 + lane;
    const bool inb = q < ntJpad * kTileRows;
    const uint32_t code = inb ? p.best[(size_t)wk.x * p.qstride + q] : kNoMatch;
    unsigned long long todo = __ballot(code != kNoMatch);
    while (todo) {
      // the sub-th pending candidate of this round goes to lane group `sub`
      unsigned long long m = todo;
      for (int i = 0; i < sub; ++i) m &= m - 1;
      const bool active = m != 0;
      const int src = active ? __builtin_ctzll(m) : 0;
      for (int i = 0; i < 4; ++i) todo &= todo - 1;
      const uint32_t cc = (uint32_t)__shfl((int)code, src);
      const uint32_t qs = q0 + (uint32_t)src;                      // query slot
      const int s1 = cc & 7, hw = (cc >> 3) & 1, g1 = (int)(cc >> 4);
      // Row rho (0..15) of the cell sits in tile g1*8 + (rho >> 1), in-tile row slot_row(2*s1 + (rho & 1), hw); rows
      // 2c and 2c+1 are adjacent slots = 256 contiguous bytes of rows_slot. Load c (0..7) of the group therefore reads
      // exactly those two rows, lane l16 taking 16-byte piece (l16 & 7) of row 2c + (l16 >> 3): 2-4 cache lines per
      // group and instruction instead of 16. Each lane needs only ONE piece of the query.
      const int piece = l16 & 7, rsel = l16 >> 3;
      const int mrow = slot_row(2 * s1 + rsel, hw);
      const size_t slot0 = (size_t)tileI0 * kTileRows + (active ? (size_t)g1 * kWinTiles * kTileRows + mrow : 0);
      const size_t slotQ = (size_t)tileJ0 * kTileRows + (active ? qs : 0);
      const size_t o = (size_t)wk.x * p.qstride + (active ? qs : 0);
      // the row this lane finally owns: rho = 2 * (l16 & 7) + (l16 >> 3)  (a permutation of 0..15)
      const uint32_t tt = (uint32_t)g1 * kWinTiles + (uint32_t)piece;
      const size_t slotI = active ? slot0 + (size_t)piece * kTileRows : slot0;
      // slots of a parity half are filled in rank order: slot (tile tt, j = 2*s1 + rsel) holds rank 16*tt + j
      const bool valid = active && (tt * 16u + (uint32_t)(2 * s1 + rsel)) < (hw ? nOddI : nEvenI);
      int4 va[8];
#pragma unroll
      for (int c = 0; c < 8; ++c)
        va[c] = *reinterpret_cast<const int4*>(p.rows_slot + (slot0 + (active ? (size_t)c * kTileRows : 0)) * kDim + piece * 16);
      const int4 vb = *reinterpret_cast<const int4*>(p.rows_slot + slotQ * kDim + piece * 16);
      const int2 f = p.cd[o];
      const int f_d0 = f.x, f_d1 = f.y;
      // |b'|^2 from the query piece itself (sum over the 8 lanes holding its pieces)
      int nb = __builtin_amdgcn_sdot4(vb.x, vb.x, 0, false);
      nb = __builtin_amdgcn_sdot4(vb.y, vb.y, nb, false);
      nb = __builtin_amdgcn_sdot4(vb.z, vb.z, nb, false);
      nb = __builtin_amdgcn_sdot4(vb.w, vb.w, nb, false);
      nb += __builtin_amdgcn_update_dpp(0, nb, 0xB1, 0xF, 0xF, true);
      nb += __builtin_amdgcn_update_dpp(0, nb, 0x4E, 0xF, 0xF, true);
      nb += __builtin_amdgcn_update_dpp(0, nb, 0x141, 0xF, 0xF, true);
      int dot = 0, na = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        int part = __builtin_amdgcn_sdot4(va[c].x, vb.x, 0, false);
        part = __builtin_amdgcn_sdot4(va[c].y, vb.y, part, false);
        part = __builtin_amdgcn_sdot4(va[c].z, vb.z, part, false);
        part = __builtin_amdgcn_sdot4(va[c].w, vb.w, part, false);
        // sum over the 8 lanes holding the 8 pieces of this row (DPP butterfly: xor 1, xor 2, half-row mirror)
        part += __builtin_amdgcn_update_dpp(0, part, 0xB1, 0xF, 0xF, true);
        part += __builtin_amdgcn_update_dpp(0, part, 0x4E, 0xF, 0xF, true);
        part += __builtin_amdgcn_update_dpp(0, part, 0x141, 0xF, 0xF, true);
        dot = (c == piece) ? part : dot;   // lane keeps the row whose tile index equals its piece index
        // |a'|^2 of the same row, same way (cheaper than a scattered load of the per-slot norm)
        int sq = __builtin_amdgcn_sdot4(va[c].x, va[c].x, 0, false);
        sq = __builtin_amdgcn_sdot4(va[c].y, va[c].y, sq, false);
        sq = __builtin_amdgcn_sdot4(va[c].z, va[c].z, sq, false);
        sq = __builtin_amdgcn_sdot4(va[c].w, va[c].w, sq, false);
        sq += __builtin_amdgcn_update_dpp(0, sq, 0xB1, 0xF, 0xF, true);
        sq += __builtin_amdgcn_update_dpp(0, sq, 0x4E, 0xF, 0xF, true);
        sq += __builtin_amdgcn_update_dpp(0, sq, 0x141, 0xF, 0xF, true);
        na = (c == piece) ? sq : na;
      }
      const int d = valid ? na + nb - 2 * dot : INT_MAX;
      // best of the cell (key = distance, lane) and runner-up over the 16 lanes of the group (DPP min butterfly)
      int key = valid ? ((d << 4) | l16) : INT_MAX;
      key = min(key, __builtin_amdgcn_update_dpp(INT_MAX, key, 0xB1, 0xF, 0xF, false));
      key = min(key, __builtin_amdgcn_update_dpp(INT_MAX, key, 0x4E, 0xF, 0xF, false));
      key = min(key, __builtin_amdgcn_update_dpp(INT_MAX, key, 0x141, 0xF, 0xF, false));
      key = min(key, __builtin_amdgcn_update_dpp(INT_MAX, key, 0x140, 0xF, 0xF, false));
      const int wl = key & 15, dbest = key >> 4;
      int second = (l16 == wl) ? INT_MAX : d;
      second = min(second, __builtin_amdgcn_update_dpp(INT_MAX, second, 0xB1, 0xF, 0xF, false));
      second = min(second, __builtin_amdgcn_update_dpp(INT_MAX, second, 0x4E, 0xF, 0xF, false));
      second = min(second, __builtin_amdgcn_update_dpp(INT_MAX, second, 0x141, 0xF, 0xF, false));
      second = min(second, __builtin_amdgcn_update_dpp(INT_MAX, second, 0x140, 0xF, 0xF, false));
      bool ok = false;
      if (active && l16 == wl) {   // the winner's lane finishes the test and translates its slot to the original row
        if (key == INT_MAX || dbest != f_d0) atomicAdd(p.errflag, 1u);
        const int d1 = min(f_d1, second);
        ok = __int2float_rn(dbest) < __fmul_rn(p.ratio_sq, __int2float_rn(d1));
        p.best[o] = ok ? p.perm[slotI] : kNoMatch;
      }
      accepted += (uint32_t)__popcll(__ballot(ok));
    }
  }
  if (lane == 0 && accepted) atomicAdd(&p.count[wk.x], accepted);
}

// statistics (profile mode only): candidates the filter hands to the verify stage. One atomic per 16 K slots — a single
// counter word saturates at ~90 atomics/us on this chip, which is why the verify kernel itself does not count.
__global__ __launch_bounds__(256) void count_candidates_kernel(const uint32_t* __restrict__ best, size_t n,
                                                               uint32_t* __restrict__ counter) {
  __shared__ uint32_t s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  uint32_t c = 0;
  const size_t base = (size_t)blockIdx.x * 16384;
  for (int k = 0; k < 64; ++k) {
    const size_t i = base + (size_t)k * 256 + threadIdx.x;
    if (i < n && best[i] != kNoMatch) ++c;
  }
  const unsigned long long m = __ballot(c != 0);
  (void)m;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_sum, c);
  __syncthreads();
  if (threadIdx.x == 0 && s_sum) atomicAdd(counter, s_sum);
}

// ------------------------------------------------------------------------------------------------
// naive check kernel (variant 0): one thread per query, plain integer loops over the ORIGINAL uint8 rows.
// No MFMA, no tiles: an independent on-device formulation used by the tests to localise faults.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_top2_ratio_naive_kernel(MatchParams p) {
  const uint2 wk = p.work[blockIdx.x];
  const uint2 ij = p.pairs[wk.x];
  const uint32_t I = ij.x, J = ij.y;
  const uint32_t nI = p.img_n[I], nJ = p.img_n[J];
  const uint8_t* rowsI = p.rows_u8 + p.img_row_off[I] * kDim;
  const uint8_t* rowsJ = p.rows_u8 + p.img_row_off[J] * kDim;
  for (uint32_t q = wk.y * kTileRows + threadIdx.x; q < min(nJ, (wk.y + kBlockQTiles) * kTileRows);
       q += blockDim.x) {
    uint32_t qv[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) qv[k] = reinterpret_cast<const uint32_t*>(rowsJ + (size_t)q * kDim)[k];
    int d0 = INT_MAX, d1 = INT_MAX;
    uint32_t i0 = 0;
    for (uint32_t i = 0; i < nI; ++i) {
      const uint32_t* dbr = reinterpret_cast<const uint32_t*>(rowsI + (size_t)i * kDim);
      int d = 0;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const uint32_t x = qv[k], y = dbr[k];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const int df = (int)((x >> (8 * bb)) & 0xFF) - (int)((y >> (8 * bb)) & 0xFF);
          d += df * df;
        }
      }
      if (d < d0) { d1 = d0; d0 = d; i0 = i; }
      else if (d < d1) { d1 = d; }
    }
    const bool ok = __int2float_rn(d0) < __fmul_rn(p.ratio_sq, __int2float_rn(d1));
    p.best[(size_t)wk.x * p.qstride + p.rowpos[p.img_row_off[J] + q]] = ok ? i0 : kNoMatch;
    if (ok) atomicAdd(&p.count[wk.x], 1u);
  }
}

// ------------------------------------------------------------------------------------------------
// compaction: exclusive scan of per-pair counts, then ordered (ascending j) gather of the accepted queries
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_counts_kernel(const uint32_t* __restrict__ count, uint32_t n,
                                                           uint32_t* __restrict__ offsets /* n+1 */) {
  __shared__ uint32_t s_part[1024];
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += count[i];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  // Hillis-Steele inclusive scan over 1024 partials
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const uint32_t v = (threadIdx.x >= d) ? s_part[threadIdx.x - d] : 0;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = s_part[threadIdx.x] - sum;
  for (uint32_t i = lo; i < hi; ++i) { offsets[i] = run; run += count[i]; }
  if (threadIdx.x == 1023) offsets[n] = s_part[1023];
}

__global__ __launch_bounds__(256) void compact_matches_kernel(const uint32_t* __restrict__ best,
                                                              const uint32_t* __restrict__ offsets,
                                                              const uint2* __restrict__ pairs,
                                                              const uint32_t* __restrict__ img_n,
                                                              const uint64_t* __restrict__ img_row_off,
                                                              const uint32_t* __restrict__ rowpos,
                                                              uint32_t n_pairs, uint32_t qstride,
                                                              uint2* __restrict__ out_ij) {
  const uint32_t pidx = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per image pair
  if (pidx >= n_pairs) return;
  const int lane = threadIdx.x & 63;
  const uint32_t off = offsets[pidx];
  if (offsets[pidx + 1] == off) return;
  const uint32_t J = pairs[pidx].y;
  const uint32_t nJ = img_n[J];
  const uint32_t* pos = rowpos + img_row_off[J];   // original query row -> slot (best[] is slot-indexed)
  uint32_t run = off;
  for (uint32_t q0 = 0; q0 < nJ; q0 += 64) {
    const uint32_t q = q0 + lane;
    const uint32_t v = (q < nJ) ? best[(size_t)pidx * qstride + pos[q]] : kNoMatch;
    const bool ok = v != kNoMatch;
    const unsigned long long m = __ballot(ok);
    if (ok) {
      const uint32_t pre = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      out_ij[run + pre] = make_uint2(v, q);
    }
    run += (uint32_t)__popcll(m);
  }
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  int ensure(size_t n) {
    if (n <= cap) return MVGX_OK;
    if (p) { MVGX_HIP(hipFree(p)); p = nullptr; cap = 0; }
    const size_t want = std::max<size_t>(n, 16);
    MVGX_HIP(mvgx::device_malloc(reinterpret_cast<void**>(&p), want * sizeof(T)));
    cap = want;
    return MVGX_OK;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
};

template <typename T>
struct PinnedBuf {
  T* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return MVGX_OK;
    if (p) { MVGX_HIP(hipHostFree(p)); p = nullptr; cap = 0; }
    const size_t want = std::max<size_t>(n, 16);
    MVGX_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), want * sizeof(T), hipHostMallocDefault));
    cap = want;
    return MVGX_OK;
  }
  bool pageable = false;   // true: plain malloc'd memory (cheap to obtain; D2H copies are staged by the runtime)
  // grow to at least n elements, keeping the first `used` (geometric growth: the caller appends batch after batch)
  int grow_keep(size_t n, size_t used) {
    if (n <= cap) return MVGX_OK;
    const size_t want = std::max<size_t>(std::max<size_t>(n, cap + cap / 2), 16);
    if (pageable) {
      T* q = static_cast<T*>(realloc(p, want * sizeof(T)));
      MVGX_REQUIRE(q != nullptr, MVGX_ERR_HIP, "out of host memory (%zu bytes)", want * sizeof(T));
      p = q;
      cap = want;
      return MVGX_OK;
    }
    T* q = nullptr;
    MVGX_HIP(hipHostMalloc(reinterpret_cast<void**>(&q), want * sizeof(T), hipHostMallocDefault));
    if (p) {
      if (used) memcpy(q, p, used * sizeof(T));
      (void)hipHostFree(p);
    }
    p = q;
    cap = want;
    return MVGX_OK;
  }
  void release() { if (p) { if (pageable) free(p); else (void)hipHostFree(p); p = nullptr; cap = 0; } }
};

}  // namespace

struct mvgx_match_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_total0 = nullptr, ev_total1 = nullptr;
  // options
  int variant = 4;          // 0 naive; exact top-2 kernel: 1 register-staged LDS, 2 LDS-DMA builtin, 3 LDS-DMA asm;
                            // 4 = filter (1 VALU / distance) + verify, LDS staging mode in `stage`
  int filter_shape = 16;    // MFMA shape of the variant-4 filter: 16 = v_mfma_i32_16x16x64_i8 (l2_filter16_kernel, round 5), 32 = 32x32x32
  int stage = 3;            // staging mode of variant 4 (1 / 2 / 3 as above); 3 measured fastest (sweep call 5)
  int profile = 0;
  int64_t batch_pairs = 1 << 15;   // 16 batches on the 1k-image set: short pipeline fill/drain, 262k workgroups per filter launch
  int keep_host_results = 1;
  int overlap = 1;   // 1: batch b's filter runs beside batch b-1's verify/scan/compaction/copies (two slots)
  int verify_alone = 0;   // 1: the next filter kernel also waits for this batch's verify kernel (the two never share the device)
  int debug_filter = 0;     // 1..7: timing experiments of l2_filter_kernel (see its kDbg), results invalid; 8, 16: earlier forms (valid)
  int stream_hold = 0;      // mvgx_match_run_stream: 1 = a batch's buffers survive two further sink calls (see Slot)
  int pinned_stream = 1;    // host buffers of the stream mode: pinned (contexts that are run repeatedly) or plain memory (one-shot)
  // regions
  uint32_t n_images = 0;
  uint32_t total_tiles = 0;
  uint32_t max_tiles_pad = 0;
  uint32_t qstride = 0;
  std::vector<uint32_t> h_n, h_tile_off, h_ntiles;
  std::vector<uint64_t> h_row_off;
  DevBuf<uint8_t> d_rows;
  bool rows_owned = true;
  const uint8_t* d_rows_view = nullptr;
  DevBuf<int8_t> d_tiles, d_rows_slot;
  DevBuf<int> d_rconst, d_cinit, d_qnorm, d_rownorm;
  DevBuf<uint64_t> d_row_off;
  DevBuf<uint32_t> d_tile_off, d_n, d_ntiles, d_perm, d_rowpos, d_neven, d_err;
  // batch scratch, two slots: while the filter kernel of batch b runs, batch b-1 is verified, scanned, compacted and
  // copied out on the other slot's stream
  struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev_scan = nullptr;     // offsets of the batch are on the host
    hipEvent_t ev_filter = nullptr;   // filter kernel of the batch has finished
    DevBuf<uint2> d_pairs, d_work, d_ij;
    DevBuf<uint4> d_work8, d_work8h;
    DevBuf<uint32_t> d_best, d_count, d_offsets;
    DevBuf<int2> d_cd;
    PinnedBuf<uint2> hp_pairs, hp_work;
    PinnedBuf<uint4> hp_work8, hp_work8h;
    PinnedBuf<uint32_t> hp_offsets;
    uint64_t p0 = 0;
    uint32_t nb = 0;
    // stream mode (mvgx_match_run_stream): the lists of the slot's finished batch wait here for the sink
    // Two sets of host buffers per slot, used alternately: with the option "stream_hold" the lists handed to the sink stay
    // valid until two further sink calls have returned (a caller can turn them into its own data structures on other threads
    // while the run goes on); without it only the first set is used
    hipEvent_t ev_copy = nullptr;     // match lists of the batch are on the host
    PinnedBuf<uint32_t> hp_ij[2];
    std::vector<uint32_t> h_off_out[2];
    int out_set = 0;
    uint64_t out_p0 = 0;
    uint32_t out_nb = 0;
  } slot[2];
  // results of the last run
  // results of the last run - or, with "double_buffer_results", of the last two runs (alternating): a caller may then
  // consume run k on another thread while run k + 1 executes
  struct Results {
    std::vector<uint64_t> offsets;
    PinnedBuf<uint32_t> ij;   // pinned: the D2H copies are plain DMA, nothing is zero-filled
    size_t ij_n = 0;
  } results[2];
  int cur = 0;
  int double_buffer = 0;
  std::vector<hipEvent_t> ev_pool;
  // a context over several devices (mvgx_match_create_multi / MVGX_DEVICES): it owns one ordinary context per device and
  // no device state of its own; descriptors are replicated, the batches of a run are shared out dynamically
  std::vector<mvgx_match_ctx*> children;
};

namespace {

int prep_regions(mvgx_match_ctx* c) {
  const uint32_t n_images = c->n_images;
  c->h_tile_off.assign(n_images + 1, 0);
  c->h_row_off.assign(n_images + 1, 0);
  c->h_ntiles.assign(n_images + 1, 0);
  uint32_t max_n = 0;
  for (uint32_t k = 0; k < n_images; ++k) {
    c->h_row_off[k + 1] = c->h_row_off[k] + c->h_n[k];
    max_n = std::max(max_n, c->h_n[k]);
  }
  const uint64_t total_rows = c->h_row_off[n_images];
  int rc;
  if ((rc = c->d_row_off.ensure(n_images + 1))) return rc;
  if ((rc = c->d_tile_off.ensure(n_images + 1))) return rc;
  if ((rc = c->d_n.ensure(n_images + 1))) return rc;
  if ((rc = c->d_ntiles.ensure(n_images + 1))) return rc;
  if ((rc = c->d_neven.ensure(n_images + 1))) return rc;
  if ((rc = c->d_err.ensure(2))) return rc;
  if ((rc = c->d_rownorm.ensure(std::max<uint64_t>(total_rows, 1)))) return rc;
  if ((rc = c->d_rowpos.ensure(std::max<uint64_t>(total_rows, 1)))) return rc;
  MVGX_HIP(hipMemcpyAsync(c->d_row_off.p, c->h_row_off.data(), (n_images + 1) * sizeof(uint64_t),
                          hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipMemcpyAsync(c->d_n.p, c->h_n.data(), n_images * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipMemsetAsync(c->d_neven.p, 0, (n_images + 1) * sizeof(uint32_t), c->stream));
  MVGX_HIP(hipMemsetAsync(c->d_err.p, 0, 2 * sizeof(uint32_t), c->stream));
  // 1/3: row norms + even-norm counts (the tile count of an image depends on its parity split)
  std::vector<uint32_t> h_even(n_images + 1, 0);
  if (total_rows > 0) {
    dim3 grid((max_n + 255) / 256, n_images);
    hipLaunchKernelGGL(row_norms_kernel, grid, dim3(256), 0, c->stream, c->d_rows_view, c->d_row_off.p, c->d_n.p,
                       c->d_rownorm.p, c->d_neven.p);
    MVGX_HIP(hipGetLastError());
    MVGX_HIP(hipMemcpyAsync(h_even.data(), c->d_neven.p, n_images * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  }
  MVGX_HIP(hipStreamSynchronize(c->stream));
  uint32_t max_pad = 0;
  for (uint32_t k = 0; k < n_images; ++k) {
    const uint32_t ne = h_even[k], no = c->h_n[k] - h_even[k];
    const uint32_t nt = (std::max(ne, no) + 15) / 16;
    const uint32_t pad = (nt + kWinTiles - 1) / kWinTiles * kWinTiles;
    c->h_ntiles[k] = nt;
    c->h_tile_off[k + 1] = c->h_tile_off[k] + pad;
    max_pad = std::max(max_pad, pad);
  }
  c->total_tiles = c->h_tile_off[n_images];
  c->max_tiles_pad = max_pad;
  c->qstride = max_pad * kTileRows;
  const size_t alloc_tiles = (size_t)c->total_tiles + kTailTiles;
  const size_t alloc_slots = alloc_tiles * kTileRows;
  if ((rc = c->d_tiles.ensure(alloc_tiles * kTileBytes))) return rc;
  if ((rc = c->d_rows_slot.ensure(alloc_tiles * kTileBytes))) return rc;
  if ((rc = c->d_rconst.ensure(alloc_slots))) return rc;
  if ((rc = c->d_cinit.ensure(alloc_slots))) return rc;
  if ((rc = c->d_qnorm.ensure(alloc_slots))) return rc;
  if ((rc = c->d_perm.ensure(alloc_slots))) return rc;
  MVGX_HIP(hipMemcpyAsync(c->d_tile_off.p, c->h_tile_off.data(), (n_images + 1) * sizeof(uint32_t),
                          hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipMemcpyAsync(c->d_ntiles.p, c->h_ntiles.data(), (n_images + 1) * sizeof(uint32_t),
                          hipMemcpyHostToDevice, c->stream));
  // every slot starts as a pad slot; slack tiles after the last image are zero data (their results are never used)
  const int fill_grid = (int)((alloc_slots + 255) / 256);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(fill_grid), dim3(256), 0, c->stream, c->d_rconst.p, alloc_slots, kRPad);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(fill_grid), dim3(256), 0, c->stream, c->d_cinit.p, alloc_slots, kCPad);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(fill_grid), dim3(256), 0, c->stream, c->d_qnorm.p, alloc_slots, 0);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(fill_grid), dim3(256), 0, c->stream, reinterpret_cast<int*>(c->d_perm.p),
                     alloc_slots, (int)kNoMatch);
  MVGX_HIP(hipGetLastError());
  MVGX_HIP(hipMemsetAsync(c->d_tiles.p + (size_t)c->total_tiles * kTileBytes, 0, (size_t)kTailTiles * kTileBytes,
                          c->stream));
  if (total_rows > 0 && max_pad > 0) {
    // 2/3: slots, 3/3: tiles
    hipLaunchKernelGGL(assign_slots_kernel, dim3(n_images), dim3(1024), 0, c->stream, c->d_row_off.p, c->d_tile_off.p,
                       c->d_n.p, c->d_rownorm.p, c->d_rowpos.p, c->d_perm.p, c->d_rconst.p, c->d_cinit.p, c->d_qnorm.p);
    MVGX_HIP(hipGetLastError());
    dim3 grid(max_pad, n_images);
    hipLaunchKernelGGL(build_tiles_kernel, grid, dim3(256), 0, c->stream, c->d_rows_view, c->d_row_off.p,
                       c->d_tile_off.p, c->d_perm.p, c->d_tiles.p, c->d_rows_slot.p);
    MVGX_HIP(hipGetLastError());
  }
  MVGX_HIP(hipStreamSynchronize(c->stream));
  return MVGX_OK;
}

hipEvent_t get_event(mvgx_match_ctx* c, size_t i) {
  while (c->ev_pool.size() <= i) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    c->ev_pool.push_back(e);
  }
  return c->ev_pool[i];
}

}  // namespace

extern "C" {

int mvgx_match_create_multi(const int* devices, int n_devices, mvgx_match_ctx** out) {
  MVGX_REQUIRE(out != nullptr && devices != nullptr && n_devices >= 1, MVGX_ERR_ARG, "mvgx_match_create_multi: bad argument");
  if (n_devices == 1) return mvgx_match_create(devices[0] < 0 ? -2 : devices[0], out);
  auto* c = new mvgx_match_ctx();
  c->device = devices[0];
  for (int k = 0; k < n_devices; ++k) {
    mvgx_match_ctx* child = nullptr;
    const int rc = mvgx_match_create(devices[k] < 0 ? -2 : devices[k], &child);   // -2: current device, MVGX_DEVICES not consulted
    if (rc) { mvgx_match_destroy(c); return rc; }
    c->children.push_back(child);
  }
  *out = c;
  return MVGX_OK;
}

int mvgx_match_create(int device, mvgx_match_ctx** out) {
  MVGX_REQUIRE(out != nullptr, MVGX_ERR_ARG, "mvgx_match_create: out is NULL");
  if (device == -1) {   // "the caller has no preference": MVGX_DEVICES may name the device(s) to use
    std::vector<int> devs;
    int rc = mvgx::devices_from_env(devs);
    if (rc) return rc;
    if (devs.size() >= 2) return mvgx_match_create_multi(devs.data(), (int)devs.size(), out);
    if (devs.size() == 1) device = devs[0];
  }
  if (device < 0) device = -1;
  int rc = mvgx::select_device(device);
  if (rc) return rc;
  auto* c = new mvgx_match_ctx();
  struct Guard {   // an early return below releases what was created so far
    mvgx_match_ctx* c;
    ~Guard() { if (c) mvgx_match_destroy(c); }
  } guard{c};
  MVGX_HIP(hipGetDevice(&c->device));
  MVGX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  MVGX_HIP(hipEventCreate(&c->ev_total0));
  MVGX_HIP(hipEventCreate(&c->ev_total1));
  for (auto& sl : c->slot) {
    MVGX_HIP(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
    MVGX_HIP(hipEventCreateWithFlags(&sl.ev_scan, hipEventDisableTiming));
    MVGX_HIP(hipEventCreateWithFlags(&sl.ev_filter, hipEventDisableTiming));
    MVGX_HIP(hipEventCreateWithFlags(&sl.ev_copy, hipEventDisableTiming));
  }
  // 2 x 33 KiB dynamic LDS for both MFMA variants
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_top2_ratio_kernel<kStageRegs>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_top2_ratio_kernel<kStageGlds>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_top2_ratio_kernel<kStageGldsAsm>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_filter16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_filter_kernel<kStageRegs>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_filter_kernel<kStageGlds>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_filter_kernel<kStageGldsAsm>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
#define MVGX_DBG_ATTR(D) MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_filter_kernel<kStageGldsAsm, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  MVGX_DBG_ATTR(1) MVGX_DBG_ATTR(2) MVGX_DBG_ATTR(3) MVGX_DBG_ATTR(4) MVGX_DBG_ATTR(5) MVGX_DBG_ATTR(6) MVGX_DBG_ATTR(7) MVGX_DBG_ATTR(8) MVGX_DBG_ATTR(16)
#undef MVGX_DBG_ATTR
  if (const char* e = getenv("MVGX_MATCH_FILTER")) {   // experiments: see l2_filter_kernel's kDbg; the same values the option accepts
    const int v = atoi(e);
#ifdef MVGX_FILTER_TIMING_VARIANTS
    const bool ok = (v >= 0 && v <= 8) || v == 16;
#else
    const bool ok = v == 0 || v == 8 || v == 16;
#endif
    if (ok) c->debug_filter = v;
    else fprintf(stderr, "[mvgx] MVGX_MATCH_FILTER=%s ignored (not a form this build of the library has)\n", e);
  }
  guard.c = nullptr;
  *out = c;
  return MVGX_OK;
}

int mvgx_match_destroy(mvgx_match_ctx* c) {
  if (!c) return MVGX_OK;
  if (!c->children.empty()) {
    for (mvgx_match_ctx* child : c->children) mvgx_match_destroy(child);
    for (auto& r : c->results) r.ij.release();
    delete c;
    return MVGX_OK;
  }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->d_rows.release(); c->d_tiles.release(); c->d_rconst.release(); c->d_qnorm.release();
  c->d_rows_slot.release(); c->d_cinit.release(); c->d_rownorm.release(); c->d_ntiles.release(); c->d_perm.release(); c->d_rowpos.release();
  c->d_neven.release(); c->d_err.release();
  c->d_row_off.release(); c->d_tile_off.release(); c->d_n.release();
  for (auto& r : c->results) r.ij.release();
  for (auto& sl : c->slot) {
    sl.d_work8.release(); sl.hp_work8.release(); sl.d_work8h.release(); sl.hp_work8h.release();
    sl.d_pairs.release(); sl.d_work.release(); sl.d_ij.release(); sl.d_cd.release();
    sl.d_best.release(); sl.d_count.release(); sl.d_offsets.release();
    sl.hp_pairs.release(); sl.hp_work.release(); sl.hp_offsets.release(); sl.hp_ij[0].release(); sl.hp_ij[1].release();
    if (sl.ev_copy) (void)hipEventDestroy(sl.ev_copy);
    if (sl.ev_scan) (void)hipEventDestroy(sl.ev_scan);
    if (sl.ev_filter) (void)hipEventDestroy(sl.ev_filter);
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
  }
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->ev_total0) (void)hipEventDestroy(c->ev_total0);
  if (c->ev_total1) (void)hipEventDestroy(c->ev_total1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return MVGX_OK;
}

int mvgx_match_set_option(mvgx_match_ctx* c, const char* key, int64_t value) {
  MVGX_REQUIRE(c && key, MVGX_ERR_ARG, "mvgx_match_set_option: NULL argument");
  if (strcmp(key, "pinned_results") && strcmp(key, "double_buffer_results") && strcmp(key, "keep_host_results"))
    for (mvgx_match_ctx* child : c->children) {   // per-device knobs; the result buffers belong to the parent
      const int rc = mvgx_match_set_option(child, key, value);
      if (rc) return rc;
    }
  if (!strcmp(key, "variant")) {
    MVGX_REQUIRE(value >= 0 && value <= 4, MVGX_ERR_ARG, "variant must be 0..4");
    c->variant = (int)value;
  } else if (!strcmp(key, "filter_shape")) {
    MVGX_REQUIRE(value == 16 || value == 17 || value == 32, MVGX_ERR_ARG, "filter_shape must be 16, 17 (16x16x64, three workgroups per CU) or 32");
    c->filter_shape = (int)value;
  } else if (!strcmp(key, "stage")) {
    MVGX_REQUIRE(value >= 1 && value <= 3, MVGX_ERR_ARG, "stage must be 1..3");
    c->stage = (int)value;
  } else if (!strcmp(key, "profile")) {
    c->profile = value < 0 ? 0 : (int)std::min<int64_t>(value, 2);   // 1: kernel timing events; 2: also count the filter's candidates
  } else if (!strcmp(key, "batch_pairs")) {
    MVGX_REQUIRE(value >= 1 && value <= (1 << 20), MVGX_ERR_ARG, "batch_pairs must be in [1, 2^20]");
    c->batch_pairs = value;
  } else if (!strcmp(key, "keep_host_results")) {
    c->keep_host_results = value != 0;
  } else if (!strcmp(key, "overlap")) {
    c->overlap = value != 0;
  } else if (!strcmp(key, "verify_alone")) {
    c->verify_alone = value != 0;
  } else if (!strcmp(key, "debug_filter")) {
#ifdef MVGX_FILTER_TIMING_VARIANTS
    MVGX_REQUIRE((value >= 0 && value <= 8) || value == 16, MVGX_ERR_ARG, "debug_filter must be 0..8 or 16");
#else
    MVGX_REQUIRE(value == 0 || value == 8 || value == 16, MVGX_ERR_ARG,
                 "debug_filter must be 0, 8 or 16 (the timing variants 1..7 return wrong lists and exist only in a library built with "
                 "-DMVGX_FILTER_TIMING_VARIANTS)");
#endif
    c->debug_filter = (int)value;
  } else if (!strcmp(key, "stream_hold")) {
    c->stream_hold = value != 0;
  } else if (!strcmp(key, "pinned_stream")) {
    c->pinned_stream = value != 0;
  } else if (!strcmp(key, "stream_reserve")) {
    // Page-locks the host buffers of the stream mode ahead of the run (value: uint32 words per buffer; with "stream_hold" four
    // buffers, else two): a caller does this on another thread while its regions are uploaded - pinning ~100 MB takes tens of
    // milliseconds, which the first batches of a run would otherwise wait for. Only with "pinned_stream"; buffers that are larger stay.
    MVGX_REQUIRE(value >= 0 && value <= (int64_t)1 << 31, MVGX_ERR_ARG, "stream_reserve: words per buffer in [0, 2^31]");
    if (c->pinned_stream && value > 0) {
      MVGX_HIP(hipSetDevice(c->device));
      for (auto& sl : c->slot)
        for (int set = 0; set < (c->stream_hold ? 2 : 1); ++set) {
          PinnedBuf<uint32_t>& hb = sl.hp_ij[set];
          if (!hb.p) hb.pageable = false;
          if (!hb.pageable && (size_t)value > hb.cap) { hb.release(); const int rc = hb.grow_keep((size_t)value, 0); if (rc) return rc; }
        }
    }
  } else if (!strcmp(key, "double_buffer_results")) {
    c->double_buffer = value != 0;
  } else if (!strcmp(key, "pinned_results")) {
    for (auto& r : c->results) {
      MVGX_REQUIRE(r.ij.p == nullptr || r.ij.pageable == (value == 0), MVGX_ERR_STATE, "pinned_results must be set before the first run");
      r.ij.pageable = value == 0;
    }
  } else {
    set_error("unknown option '%s'", key);
    return MVGX_ERR_ARG;
  }
  return MVGX_OK;
}

int mvgx_match_set_regions(mvgx_match_ctx* c, const uint8_t* const* desc_rows, const uint32_t* n_desc,
                           uint32_t n_images, uint32_t dim) {
  MVGX_REQUIRE(c && n_desc && (desc_rows || n_images == 0), MVGX_ERR_ARG, "mvgx_match_set_regions: NULL argument");
  MVGX_REQUIRE(dim == kDim, MVGX_ERR_UNSUPPORTED, "descriptor length %u unsupported (device path is 128-D uint8)", dim);
  if (!c->children.empty()) {   // replicate the descriptors: one uploading thread per device
    const size_t nd = c->children.size();
    std::vector<int> rcs(nd, MVGX_OK);
    std::vector<std::string> errs(nd);
    std::vector<std::thread> th;
    for (size_t d = 0; d < nd; ++d)
      th.emplace_back([&, d]() {
        rcs[d] = mvgx_match_set_regions(c->children[d], desc_rows, n_desc, n_images, dim);
        if (rcs[d]) errs[d] = mvgx_last_error();
      });
    for (auto& t : th) t.join();
    for (size_t d = 0; d < nd; ++d)
      if (rcs[d]) { set_error("device %d: %s", c->children[d]->device, errs[d].c_str()); return rcs[d]; }
    return MVGX_OK;
  }
  MVGX_HIP(hipSetDevice(c->device));
  c->n_images = n_images;
  c->h_n.assign(n_desc, n_desc + n_images);
  uint64_t total_rows = 0;
  for (uint32_t k = 0; k < n_images; ++k) {
    MVGX_REQUIRE(n_desc[k] == 0 || desc_rows[k] != nullptr, MVGX_ERR_ARG, "image %u: NULL descriptor array", k);
    total_rows += n_desc[k];
  }
  int rc = c->d_rows.ensure(std::max<uint64_t>(total_rows, 1) * kDim);
  if (rc) return rc;
  c->rows_owned = true;
  c->d_rows_view = c->d_rows.p;
  uint64_t row = 0;
  for (uint32_t k = 0; k < n_images; ++k) {
    if (n_desc[k])
      MVGX_HIP(hipMemcpyAsync(c->d_rows.p + row * kDim, desc_rows[k], (size_t)n_desc[k] * kDim,
                              hipMemcpyHostToDevice, c->stream));
    row += n_desc[k];
  }
  return prep_regions(c);
}

int mvgx_match_set_regions_device(mvgx_match_ctx* c, const void* d_desc_concat, const uint32_t* n_desc,
                                  uint32_t n_images, uint32_t dim) {
  MVGX_REQUIRE(c && n_desc && (d_desc_concat || n_images == 0), MVGX_ERR_ARG,
               "mvgx_match_set_regions_device: NULL argument");
  MVGX_REQUIRE(dim == kDim, MVGX_ERR_UNSUPPORTED, "descriptor length %u unsupported (device path is 128-D uint8)", dim);
  MVGX_REQUIRE(c->children.empty(), MVGX_ERR_UNSUPPORTED, "mvgx_match_set_regions_device: a device pointer belongs to one device; "
               "use mvgx_match_set_regions on a multi-device context");
  MVGX_HIP(hipSetDevice(c->device));
  c->n_images = n_images;
  c->h_n.assign(n_desc, n_desc + n_images);
  c->rows_owned = false;
  c->d_rows_view = static_cast<const uint8_t*>(d_desc_concat);
  return prep_regions(c);
}

}  // extern "C"

namespace {

// The pair list of one run, cut into batches that the devices of the context take one after the other (a single device: in
// order; several: whoever is free - the lists of a batch do not depend on which device computed them).
struct BatchFeed {
  const uint32_t* pairs_IJ = nullptr;
  uint64_t n_pairs = 0, B = 1;
  std::atomic<uint64_t> next{0};
  std::atomic<int> stop{0};   // a sink asked to cancel, or a device failed
  bool take(uint64_t& p0, uint32_t& nb) {
    if (stop.load(std::memory_order_relaxed)) return false;
    p0 = next.fetch_add(B);
    if (p0 >= n_pairs) return false;
    nb = (uint32_t)std::min<uint64_t>(B, n_pairs - p0);
    return true;
  }
};

// Called when the lists of a batch are in host memory: pairs [p0, p0 + nb), offsets[nb + 1] relative to the batch (in
// matches), ij = 2 uint32 per match. Non-zero return = stop the run.
using BatchSink = std::function<int(uint64_t p0, uint32_t nb, const uint32_t* offsets, const uint32_t* ij)>;

// One device's share of a run. sink == nullptr: the lists are appended to c->results[c->cur] (the feed must then be
// consumed by this device alone, in order); otherwise every batch is handed to `sink` from this thread as soon as its
// lists are on the host, and host memory stays O(batch): two pinned buffers per device.
int run_device(mvgx_match_ctx* c, BatchFeed& feed, float ratio_sq, const BatchSink* sink, mvgx_match_stats& st) {
  const uint32_t* pairs_IJ = feed.pairs_IJ;
  MVGX_HIP(hipSetDevice(c->device));
  mvgx_match_ctx::Results& res = c->results[c->cur];
  memset(&st, 0, sizeof(st));
  st.variant = (uint32_t)c->variant;
  size_t n_ev = 0;
  int rc;

  MVGX_HIP(hipEventRecord(c->ev_total0, c->slot[0].stream));

  // Stage 1 of a batch on its slot's stream: work list -> filter (+ verify) -> per-pair counts -> exclusive scan -> offsets to host
  auto issue = [&](mvgx_match_ctx::Slot& sl, mvgx_match_ctx::Slot* prev, uint64_t p0, uint32_t nb) -> int {
    int rc;
    hipStream_t stream = sl.stream;
    sl.p0 = p0; sl.nb = nb;
    if ((rc = sl.hp_pairs.ensure(nb))) return rc;
    // worst-case work items: ceil(max tiles / 16) per pair
    const uint32_t max_blocks_per_pair = std::max<uint32_t>(1, (c->max_tiles_pad + kBlockQTiles - 1) / kBlockQTiles);
    if ((rc = sl.hp_work.ensure((size_t)nb * max_blocks_per_pair))) return rc;
    const bool records = c->variant == 4 && c->filter_shape == 16 && c->stage == 3 && !c->debug_filter;
    const bool records_h = c->variant == 4 && c->filter_shape == 17 && c->stage == 3 && !c->debug_filter;
    if (records && (rc = sl.hp_work8.ensure((size_t)nb * max_blocks_per_pair * 2))) return rc;
    if (records_h && (rc = sl.hp_work8h.ensure((size_t)nb * max_blocks_per_pair * 4))) return rc;
    uint32_t n_work_h = 0;
    uint32_t n_work = 0;
    for (uint32_t k = 0; k < nb; ++k) {
      const uint32_t I = pairs_IJ[2 * (p0 + k)], J = pairs_IJ[2 * (p0 + k) + 1];
      sl.hp_pairs.p[k] = make_uint2(I, J);
      const uint32_t nI = c->h_n[I], nJ = c->h_n[J];
      // matcher_brute_force.hpp:108-113: NN(=2) > rows  -> no result; Matcher_Regions.cpp:65-69,85-90: empty regions skipped
      if (nI < 2 || nJ == 0) continue;
      const uint32_t ntJ = c->h_ntiles[J];   // occupied tiles of the query image
      for (uint32_t qt = 0; qt < ntJ; qt += kBlockQTiles) {
        if (records) {   // everything the filter's workgroup needs to start, in ONE scalar load (it used to walk work -> pairs -> three per-image tables)
          sl.hp_work8.p[2 * n_work] = make_uint4(k, qt, c->h_tile_off[I], c->h_tile_off[J]);
          sl.hp_work8.p[2 * n_work + 1] = make_uint4(c->h_ntiles[I], c->h_tile_off[J + 1] - c->h_tile_off[J], 0, 0);
        }
        if (records_h)
          for (uint32_t q2 = qt; q2 < std::min(ntJ, qt + (uint32_t)kBlockQTiles); q2 += kBlockQTilesH) {
            sl.hp_work8h.p[2 * n_work_h] = make_uint4(k, q2, c->h_tile_off[I], c->h_tile_off[J]);
            sl.hp_work8h.p[2 * n_work_h + 1] = make_uint4(c->h_ntiles[I], c->h_tile_off[J + 1] - c->h_tile_off[J], 0, 0);
            ++n_work_h;
          }
        sl.hp_work.p[n_work++] = make_uint2(k, qt);
      }
      st.n_pairs += 1;
      st.n_desc_pairs += (uint64_t)nI * nJ;
    }
    if ((rc = sl.d_pairs.ensure(nb))) return rc;
    if ((rc = sl.d_work.ensure(std::max<uint32_t>(n_work, 1)))) return rc;
    if (records && (rc = sl.d_work8.ensure((size_t)std::max<uint32_t>(n_work, 1) * 2))) return rc;
    if (records_h && (rc = sl.d_work8h.ensure((size_t)std::max<uint32_t>(n_work_h, 1) * 2))) return rc;
    if ((rc = sl.d_best.ensure((size_t)nb * c->qstride))) return rc;
    if (c->variant == 4) {
      if ((rc = sl.d_cd.ensure((size_t)nb * c->qstride))) return rc;
    }
    if ((rc = sl.d_count.ensure(nb))) return rc;
    if ((rc = sl.d_offsets.ensure((size_t)nb + 1))) return rc;
    if ((rc = sl.hp_offsets.ensure((size_t)nb + 1))) return rc;
    MVGX_HIP(hipMemcpyAsync(sl.d_pairs.p, sl.hp_pairs.p, nb * sizeof(uint2), hipMemcpyHostToDevice, stream));
    if (n_work)
      MVGX_HIP(hipMemcpyAsync(sl.d_work.p, sl.hp_work.p, n_work * sizeof(uint2), hipMemcpyHostToDevice, stream));
    if (n_work && records)
      MVGX_HIP(hipMemcpyAsync(sl.d_work8.p, sl.hp_work8.p, (size_t)n_work * 2 * sizeof(uint4), hipMemcpyHostToDevice, stream));
    if (n_work_h && records_h)
      MVGX_HIP(hipMemcpyAsync(sl.d_work8h.p, sl.hp_work8h.p, (size_t)n_work_h * 2 * sizeof(uint4), hipMemcpyHostToDevice, stream));
    MVGX_HIP(hipMemsetAsync(sl.d_count.p, 0, nb * sizeof(uint32_t), stream));

    MatchParams mp;
    mp.tiles = c->d_tiles.p; mp.rows_slot = c->d_rows_slot.p; mp.rconst = c->d_rconst.p; mp.cinit = c->d_cinit.p; mp.qnorm = c->d_qnorm.p;
    mp.perm = c->d_perm.p; mp.rowpos = c->d_rowpos.p;
    mp.rows_u8 = c->d_rows_view; mp.img_row_off = c->d_row_off.p;
    mp.img_tile_off = c->d_tile_off.p; mp.img_n = c->d_n.p; mp.img_ntiles = c->d_ntiles.p;
    mp.cd = sl.d_cd.p; mp.img_neven = c->d_neven.p; mp.errflag = c->d_err.p;
    mp.pairs = sl.d_pairs.p; mp.work = sl.d_work.p; mp.work8 = sl.d_work8.p; mp.work8h = sl.d_work8h.p; mp.n_work = n_work;
    mp.best = sl.d_best.p; mp.count = sl.d_count.p; mp.qstride = c->qstride; mp.ratio_sq = ratio_sq;

    // filter kernels run one after the other (each fills the device); everything else of batch b-1 runs beside filter b
    if (prev) MVGX_HIP(hipStreamWaitEvent(stream, prev->ev_filter, 0));
    if (n_work) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (c->profile) {
        e0 = get_event(c, n_ev++); e1 = get_event(c, n_ev++);
        MVGX_REQUIRE(e0 && e1, MVGX_ERR_HIP, "hipEventCreate failed");
        MVGX_HIP(hipEventRecord(e0, stream));
      }
      if (c->variant == 0) {
        hipLaunchKernelGGL(l2_top2_ratio_naive_kernel, dim3(n_work), dim3(256), 0, stream, mp);
      } else if (c->variant == 1) {
        hipLaunchKernelGGL(l2_top2_ratio_kernel<kStageRegs>, dim3(n_work), dim3(256), 2 * kStageBytes, stream, mp);
      } else if (c->variant == 2) {
        hipLaunchKernelGGL(l2_top2_ratio_kernel<kStageGlds>, dim3(n_work), dim3(256), 2 * kStageBytes, stream, mp);
      } else if (c->variant == 3) {
        hipLaunchKernelGGL(l2_top2_ratio_kernel<kStageGldsAsm>, dim3(n_work), dim3(256), 2 * kStageBytes, stream, mp);
      } else if (c->filter_shape == 16 && c->stage == 3 && !c->debug_filter) {
        hipLaunchKernelGGL(l2_filter16_kernel, dim3(n_work), dim3(256), 2 * kStageBytes, stream, mp);
      } else if (c->filter_shape == 17 && c->stage == 3 && !c->debug_filter) {
        hipLaunchKernelGGL(l2_filter16h_kernel, dim3(n_work_h), dim3(256), 2 * kHalfStageBytes, stream, mp);
      } else if (c->stage == 1) {
        hipLaunchKernelGGL(l2_filter_kernel<kStageRegs>, dim3(n_work), dim3(256), 2 * kStageBytes, stream, mp);
      } else if (c->stage == 2) {
        hipLaunchKernelGGL(l2_filter_kernel<kStageGlds>, dim3(n_work), dim3(256), 2 * kStageBytes, stream, mp);
      } else if (c->debug_filter) {   // timing experiments (wrong results; verify is skipped below)
#define MVGX_DBG_CASE(D) case D: hipLaunchKernelGGL((l2_filter_kernel<kStageGldsAsm, D>), dim3(n_work), dim3(256), 2 * kStageBytes, stream, mp); break;
        switch (c->debug_filter) {
#ifdef MVGX_FILTER_TIMING_VARIANTS   // parts of the kernel compiled out (wrong results): only in a build made for tools/filter_breakdown.py
          MVGX_DBG_CASE(1) MVGX_DBG_CASE(2) MVGX_DBG_CASE(3) MVGX_DBG_CASE(4) MVGX_DBG_CASE(5) MVGX_DBG_CASE(6) MVGX_DBG_CASE(7)
#endif
          MVGX_DBG_CASE(8) MVGX_DBG_CASE(16) default: break; }
#undef MVGX_DBG_CASE
      } else {
        hipLaunchKernelGGL(l2_filter_kernel<kStageGldsAsm>, dim3(n_work), dim3(256), 2 * kStageBytes, stream, mp);
      }
      MVGX_HIP(hipGetLastError());
      if (c->profile) MVGX_HIP(hipEventRecord(e1, stream));
      if (!c->verify_alone) MVGX_HIP(hipEventRecord(sl.ev_filter, stream));
      st.n_kernel_launches += 1;
      if (c->variant == 4 && !(c->debug_filter & 7)) {
        if (c->profile >= 2) {   // statistics pass (0.13 ms per batch): only on request, not in the timed runs of bench.py ("profile" 1)
          const size_t nslots = (size_t)nb * c->qstride;
          hipLaunchKernelGGL(count_candidates_kernel, dim3((unsigned)((nslots + 16383) / 16384)), dim3(256), 0, stream,
                             sl.d_best.p, nslots, c->d_err.p + 1);
        }
        hipLaunchKernelGGL(l2_verify_kernel, dim3(n_work), dim3(256), 0, stream, mp);
        MVGX_HIP(hipGetLastError());
      }
      if (c->verify_alone) MVGX_HIP(hipEventRecord(sl.ev_filter, stream));
    }
    if (!n_work) MVGX_HIP(hipEventRecord(sl.ev_filter, stream));
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, sl.d_count.p, nb, sl.d_offsets.p);
    MVGX_HIP(hipGetLastError());
    MVGX_HIP(hipMemcpyAsync(sl.hp_offsets.p, sl.d_offsets.p, ((size_t)nb + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    MVGX_HIP(hipEventRecord(sl.ev_scan, stream));
    return MVGX_OK;
  };
  // Stage 2, in batch order: totals -> ordered compaction -> copy of the match lists to the host (collect mode: appended to
  // the run's result buffer; stream mode: into the slot's own pinned buffer, delivered by stage 3)
  auto finish = [&](mvgx_match_ctx::Slot& sl) -> int {
    int rc;
    MVGX_HIP(hipEventSynchronize(sl.ev_scan));
    const uint32_t nb = sl.nb;
    const uint64_t p0 = sl.p0;
    const uint32_t total = sl.hp_offsets.p[nb];
    if (sink) {
      if (c->stream_hold) sl.out_set ^= 1;
      sl.h_off_out[sl.out_set].assign(sl.hp_offsets.p, sl.hp_offsets.p + nb + 1);   // hp_offsets is rewritten by the slot's next batch
    } else {
      const uint64_t base = res.offsets[p0];
      for (uint32_t k = 0; k <= nb; ++k) res.offsets[p0 + k] = base + sl.hp_offsets.p[k];
    }
    sl.out_p0 = p0; sl.out_nb = nb;
    st.n_matches += total;
    if (total) {
      if ((rc = sl.d_ij.ensure(total))) return rc;
      hipLaunchKernelGGL(compact_matches_kernel, dim3((nb + 3) / 4), dim3(256), 0, sl.stream, sl.d_best.p,
                         sl.d_offsets.p, sl.d_pairs.p, c->d_n.p, c->d_row_off.p, c->d_rowpos.p, nb, c->qstride,
                         sl.d_ij.p);
      MVGX_HIP(hipGetLastError());
      if (sink) {
        PinnedBuf<uint32_t>& hb = sl.hp_ij[sl.out_set];
        if (!hb.p) hb.pageable = !c->pinned_stream;
        // batch sizes vary: grow with headroom (re-pinning ~100 MB costs tens of milliseconds); nothing of the old content is kept
        if ((size_t)total * 2 > hb.cap) { hb.release(); if ((rc = hb.grow_keep(std::max<size_t>((size_t)total * 3, 4u << 20), 0))) return rc; }
        MVGX_HIP(hipMemcpyAsync(hb.p, sl.d_ij.p, (size_t)total * sizeof(uint2), hipMemcpyDeviceToHost, sl.stream));
      } else if (c->keep_host_results) {
        const size_t old = res.ij_n;
        if (old + (size_t)total * 2 > res.ij.cap)   // growth moves the lists: no copy into them may be in flight
          for (auto& o : c->slot) MVGX_HIP(hipStreamSynchronize(o.stream));
        if ((rc = res.ij.grow_keep(old + (size_t)total * 2, old))) return rc;
        res.ij_n = old + (size_t)total * 2;
        MVGX_HIP(hipMemcpyAsync(res.ij.p + old, sl.d_ij.p, (size_t)total * sizeof(uint2),
                                hipMemcpyDeviceToHost, sl.stream));   // completes before the run returns
      }
    }
    if (sink) MVGX_HIP(hipEventRecord(sl.ev_copy, sl.stream));
    return MVGX_OK;
  };
  // Stage 3 (stream mode): the lists of the slot's batch are on the host -> the sink, on this thread
  auto deliver = [&](mvgx_match_ctx::Slot& sl) -> int {
    MVGX_HIP(hipEventSynchronize(sl.ev_copy));
    if (feed.stop.load()) return MVGX_OK;   // cancelled: the sink is not entered again
    if ((*sink)(sl.out_p0, sl.out_nb, sl.h_off_out[sl.out_set].data(), sl.hp_ij[sl.out_set].p)) feed.stop.store(1);
    return MVGX_OK;
  };
  // software pipeline over the batches this device takes: issue(b) | finish(b-1) | deliver(b-2)
  uint64_t nbatch = 0, p0 = 0;
  uint32_t nb = 0;
  while (feed.take(p0, nb)) {
    mvgx_match_ctx::Slot& cur = c->slot[nbatch & 1];
    mvgx_match_ctx::Slot* prev = nbatch ? &c->slot[(nbatch - 1) & 1] : nullptr;
    if (!c->overlap) {
      if ((rc = issue(cur, nullptr, p0, nb)) || (rc = finish(cur))) return rc;
      MVGX_HIP(hipStreamSynchronize(cur.stream));
      if (sink && (rc = deliver(cur))) return rc;
      ++nbatch;
      continue;
    }
    if ((rc = issue(cur, prev, p0, nb))) return rc;   // batch b: filter on the device ...
    if (prev && (rc = finish(*prev))) return rc;      // ... while batch b-1 is compacted and copied out ...
    if (sink && nbatch >= 2 && (rc = deliver(cur))) return rc;   // ... and batch b-2 (this slot's previous one) is consumed
    // (deliver(cur) reads the slot's HOST buffers of batch b-2; the device side of the slot already works on batch b)
    ++nbatch;
  }
  if (c->overlap && nbatch) {
    mvgx_match_ctx::Slot& last = c->slot[(nbatch - 1) & 1];
    if (sink && nbatch >= 2 && (rc = deliver(c->slot[nbatch & 1]))) return rc;   // batch nbatch-2
    if ((rc = finish(last))) return rc;
    if (sink && (rc = deliver(last))) return rc;
  }
  for (auto& sl : c->slot) MVGX_HIP(hipStreamSynchronize(sl.stream));
  MVGX_HIP(hipEventRecord(c->ev_total1, c->slot[0].stream));
  MVGX_HIP(hipEventSynchronize(c->ev_total1));
  float ms = 0.f;
  MVGX_HIP(hipEventElapsedTime(&ms, c->ev_total0, c->ev_total1));
  st.total_ms = ms;
  if (c->variant == 4 && !(c->debug_filter & 7)) {   // the verify kernel cross-checks the filter's d0 against its own exact recomputation
    uint32_t flags[2] = {0, 0};
    MVGX_HIP(hipMemcpy(flags, c->d_err.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    (void)hipMemset(c->d_err.p, 0, 2 * sizeof(uint32_t));
    const uint32_t nerr = flags[0];
    st.kernel_vgprs = flags[1];   // (field reused) candidates verified in this run, modulo 2^32
    if (nerr) {
      set_error("matching filter/verify disagreement on %u candidates (internal error)", nerr);
      return MVGX_ERR_NUMERIC;
    }
  }
  if (c->profile) {
    for (size_t i = 0; i + 1 < n_ev; i += 2) {
      float k = 0.f;
      MVGX_HIP(hipEventElapsedTime(&k, c->ev_pool[i], c->ev_pool[i + 1]));
      st.kernel_ms += k;
    }
  }
  return MVGX_OK;
}

void add_stats(mvgx_match_stats& a, const mvgx_match_stats& b) {
  a.n_pairs += b.n_pairs; a.n_desc_pairs += b.n_desc_pairs; a.n_matches += b.n_matches;
  a.n_kernel_launches += b.n_kernel_launches; a.kernel_ms += b.kernel_ms;
  a.total_ms = std::max(a.total_ms, b.total_ms); a.kernel_vgprs += b.kernel_vgprs; a.variant = b.variant;
}

// Several devices: one host thread per device runs run_device() on the shared feed; finished batches are handed to the
// CALLING thread through a one-entry mailbox per device, so that `sink` is never entered from two threads nor from a
// thread the caller does not know (PairWiseMatchesContainer::insert is not thread safe, indMatch.hpp:70-75).
int run_multi(mvgx_match_ctx* c, BatchFeed& feed, float ratio_sq, const BatchSink& sink, mvgx_match_stats& st) {
  const size_t nd = c->children.size();
  struct Mail { uint64_t p0; uint32_t nb; const uint32_t* offsets; const uint32_t* ij; int result; bool full, done; };
  std::mutex mu;
  std::condition_variable cv_main, cv_dev;
  std::vector<Mail> mail(nd, Mail{0, 0, nullptr, nullptr, 0, false, false});
  std::vector<int> rcs(nd, MVGX_OK);
  std::vector<std::string> errs(nd);
  std::vector<mvgx_match_stats> sts(nd);
  size_t running = nd;
  std::vector<std::thread> workers;
  for (size_t d = 0; d < nd; ++d) {
    workers.emplace_back([&, d]() {
      BatchSink post = [&, d](uint64_t p0, uint32_t nb, const uint32_t* offsets, const uint32_t* ij) -> int {
        std::unique_lock<std::mutex> lk(mu);
        mail[d] = Mail{p0, nb, offsets, ij, 0, true, false};
        cv_main.notify_one();
        cv_dev.wait(lk, [&]() { return mail[d].done; });   // the buffers stay valid until the caller has consumed them
        mail[d].full = mail[d].done = false;
        return mail[d].result;
      };
      rcs[d] = run_device(c->children[d], feed, ratio_sq, &post, sts[d]);
      if (rcs[d] != MVGX_OK) { errs[d] = mvgx_last_error(); feed.stop.store(1); }
      std::lock_guard<std::mutex> lk(mu);
      --running;
      cv_main.notify_one();
    });
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      size_t ready = nd;
      cv_main.wait(lk, [&]() {
        for (size_t d = 0; d < nd; ++d)
          if (mail[d].full && !mail[d].done) { ready = d; return true; }
        return running == 0;
      });
      if (ready == nd) break;
      Mail m = mail[ready];
      if (feed.stop.load()) {   // a sink call already returned non-zero (or a device failed): include/mvgx.h promises the sink is not entered again
        mail[ready].result = 1;
        mail[ready].done = true;
        cv_dev.notify_all();
        continue;
      }
      lk.unlock();
      const int r = sink(m.p0, m.nb, m.offsets, m.ij);
      if (r != 0) feed.stop.store(1);
      lk.lock();
      mail[ready].result = r;
      mail[ready].done = true;
      cv_dev.notify_all();
    }
  }
  for (auto& w : workers) w.join();
  memset(&st, 0, sizeof(st));
  int rc = MVGX_OK;
  for (size_t d = 0; d < nd; ++d) {
    add_stats(st, sts[d]);
    if (rcs[d] != MVGX_OK && rc == MVGX_OK) { rc = rcs[d]; set_error("device %d: %s", c->children[d]->device, errs[d].c_str()); }
  }
  return rc;
}

int check_run_args(mvgx_match_ctx* c, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq, const char* who) {
  MVGX_REQUIRE(c && (pairs_IJ || n_pairs == 0), MVGX_ERR_ARG, "%s: NULL argument", who);
  const mvgx_match_ctx* r = c->children.empty() ? c : c->children[0];
  MVGX_REQUIRE(r->d_rows_view != nullptr || r->n_images == 0, MVGX_ERR_STATE, "%s before set_regions", who);
  MVGX_REQUIRE(ratio_sq <= 1.0f && ratio_sq >= 0.0f, MVGX_ERR_UNSUPPORTED,
               "ratio_sq = %g: the device path reproduces the reference only for 0 <= ratio^2 <= 1 "
               "(ties are libstdc++ partial_sort order beyond that)", (double)ratio_sq);
  for (uint64_t k = 0; k < n_pairs; ++k)
    MVGX_REQUIRE(pairs_IJ[2 * k] < r->n_images && pairs_IJ[2 * k + 1] < r->n_images, MVGX_ERR_ARG,
                 "pair %llu references image out of range", (unsigned long long)k);
  return MVGX_OK;
}

// pairs per batch: the option, capped so that the per-slot scratch (12 B per pair and query slot) stays near 6 GB when the
// images carry tens of thousands of descriptors
uint64_t batch_size(const mvgx_match_ctx* c) {
  const mvgx_match_ctx* r = c->children.empty() ? c : c->children[0];
  return std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)c->batch_pairs, std::max<uint64_t>(16, (1ull << 29) / std::max<uint32_t>(r->qstride, 1))));
}

}  // namespace

extern "C" {

int mvgx_match_run(mvgx_match_ctx* c, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq,
                   mvgx_match_stats* stats) {
  int rc = check_run_args(c, pairs_IJ, n_pairs, ratio_sq, "mvgx_match_run");
  if (rc) return rc;
  if (c->double_buffer) c->cur ^= 1;
  mvgx_match_ctx::Results& res = c->results[c->cur];
  res.offsets.assign(n_pairs + 1, 0);
  res.ij_n = 0;
  BatchFeed feed;
  feed.pairs_IJ = pairs_IJ; feed.n_pairs = n_pairs; feed.B = batch_size(c);
  mvgx_match_stats st;
  if (c->children.empty()) {
    rc = run_device(c, feed, ratio_sq, nullptr, st);
  } else {
    // several devices: batches arrive in any order; per-pair counts first, the lists are put in place once all offsets are known
    struct Piece { uint64_t p0; std::vector<uint32_t> ij; };
    std::vector<Piece> pieces;
    BatchSink collect = [&](uint64_t p0, uint32_t nb, const uint32_t* offsets, const uint32_t* ij) -> int {
      for (uint32_t k = 0; k < nb; ++k) res.offsets[p0 + k + 1] = offsets[k + 1] - offsets[k];
      if (c->keep_host_results && offsets[nb]) pieces.push_back(Piece{p0, std::vector<uint32_t>(ij, ij + 2 * (size_t)offsets[nb])});
      return 0;
    };
    rc = run_multi(c, feed, ratio_sq, collect, st);
    if (rc == MVGX_OK) {
      for (uint64_t k = 0; k < n_pairs; ++k) res.offsets[k + 1] += res.offsets[k];
      if (c->keep_host_results) {
        if ((rc = res.ij.grow_keep(2 * res.offsets[n_pairs], 0))) return rc;
        res.ij_n = 2 * res.offsets[n_pairs];
        for (const Piece& p : pieces) memcpy(res.ij.p + 2 * res.offsets[p.p0], p.ij.data(), p.ij.size() * sizeof(uint32_t));
      }
    }
  }
  if (rc == MVGX_OK && stats) *stats = st;
  return rc;
}

int mvgx_match_run_stream(mvgx_match_ctx* c, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq,
                          mvgx_match_batch_sink sink, void* user, mvgx_match_stats* stats) {
  int rc = check_run_args(c, pairs_IJ, n_pairs, ratio_sq, "mvgx_match_run_stream");
  if (rc) return rc;
  MVGX_REQUIRE(sink != nullptr, MVGX_ERR_ARG, "mvgx_match_run_stream: sink is NULL");
  BatchFeed feed;
  feed.pairs_IJ = pairs_IJ; feed.n_pairs = n_pairs; feed.B = batch_size(c);
  BatchSink fn = [&](uint64_t p0, uint32_t nb, const uint32_t* offsets, const uint32_t* ij) -> int {
    return sink(user, p0, nb, offsets, ij);
  };
  mvgx_match_stats st;
  rc = c->children.empty() ? run_device(c, feed, ratio_sq, &fn, st) : run_multi(c, feed, ratio_sq, fn, st);
  if (rc == MVGX_OK && stats) *stats = st;
  return rc;
}

int mvgx_match_results(mvgx_match_ctx* c, const uint64_t** offsets, const uint32_t** ij) {
  MVGX_REQUIRE(c && offsets && ij, MVGX_ERR_ARG, "mvgx_match_results: NULL argument");
  *offsets = c->results[c->cur].offsets.data();
  *ij = c->results[c->cur].ij.p;
  return MVGX_OK;
}

int mvgx_match_pairs_u8_l2(const uint8_t* const* desc_rows, const uint32_t* n_desc, uint32_t n_images,
                           uint32_t dim, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq, int device,
                           mvgx_match_sink sink, void* user) {
  mvgx_match_ctx* c = nullptr;
  int rc = mvgx_match_create(device, &c);
  if (rc) return rc;
  rc = mvgx_match_set_regions(c, desc_rows, n_desc, n_images, dim);
  if (!rc) rc = mvgx_match_run(c, pairs_IJ, n_pairs, ratio_sq, nullptr);
  if (!rc && sink) {
    for (uint64_t k = 0; k < n_pairs; ++k) {
      const uint64_t a = c->results[c->cur].offsets[k], b = c->results[c->cur].offsets[k + 1];
      if (b > a) sink(user, pairs_IJ[2 * k], pairs_IJ[2 * k + 1], c->results[c->cur].ij.p + 2 * a, (uint32_t)(b - a));
    }
  }
  mvgx_match_destroy(c);
  return rc;
}

}  // extern "C"
