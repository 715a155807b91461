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
// HBM layout (built once per image set by prep_tiles_kernel):
//   tiles  : int8, one 4 KiB block per 32 descriptors, "fragment-major": chunk c = s*64 + lane holds the 16
//            bytes lane `lane` feeds to MFMA k-step s (row = lane&31, k-half = lane>>5). A straight 1 KiB
//            coalesced read per wave-instruction therefore IS an MFMA operand; the same image serves as
//            A (database, through LDS) or B (queries, register-resident) operand.
//   rconst : int32 per descriptor, R = -(|a'|^2 << 8) - (row index within its 256-row window); pad rows hold
//            INT_MIN + 512 so they never win.
//   qnorm  : int32 per descriptor, |a'|^2.
// Each image owns ceil(n/32) tiles rounded up to a multiple of 8 (one LDS window = 8 tiles = 256 rows).
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
#include <climits>
#include <cstdint>
#include <cstring>
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
constexpr uint32_t kNoMatch = 0xFFFFFFFFu;
constexpr int kTailTiles = 16;                // slack tiles after the last image (query over-read)

struct MatchParams {
  const int8_t* tiles;
  const int* rconst;
  const int* qnorm;
  const uint8_t* rows_u8;          // original row-major descriptors (naive check kernel only)
  const uint64_t* img_row_off;     // first row of each image in rows_u8
  const uint32_t* img_tile_off;    // first tile of each image
  const uint32_t* img_n;           // descriptors per image
  const uint2* pairs;              // batch-local (I, J)
  const uint2* work;               // (batch-local pair index, first query tile)
  uint32_t n_work;
  uint32_t* best;                  // [batch pairs][qstride]: index in I of the accepted match or kNoMatch
  uint32_t* count;                 // [batch pairs]: accepted matches
  uint32_t qstride;
  float ratio_sq;
};

// ------------------------------------------------------------------------------------------------
// prep: row-major uint8 descriptors -> fragment-major int8 tiles + rconst + qnorm
// grid = (max padded tiles per image, n_images), block = 256
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_tiles_kernel(const uint8_t* __restrict__ rows,
                                                         const uint64_t* __restrict__ img_row_off,
                                                         const uint32_t* __restrict__ img_tile_off,
                                                         const uint32_t* __restrict__ img_n,
                                                         int8_t* __restrict__ tiles, int* __restrict__ rconst,
                                                         int* __restrict__ qnorm) {
  const uint32_t img = blockIdx.y;
  const uint32_t n = img_n[img];
  const uint32_t ntiles_pad = img_tile_off[img + 1] - img_tile_off[img];
  const uint32_t t = blockIdx.x;
  if (t >= ntiles_pad) return;
  const uint32_t gt = img_tile_off[img] + t;
  const uint8_t* src = rows + img_row_off[img] * kDim;
  __shared__ int s_norm[kTileRows][8];

  const int c = threadIdx.x;        // chunk id = s*64 + lane
  const int s = c >> 6, lane = c & 63;
  const int m = lane & 31, h = lane >> 5;
  const uint32_t row = t * kTileRows + m;
  const int kchunk = s * 2 + h;     // source bytes [16*kchunk, 16*kchunk + 16)
  uint4 v = make_uint4(0, 0, 0, 0);
  int part = 0;
  if (row < n) {
    v = *reinterpret_cast<const uint4*>(src + (size_t)row * kDim + kchunk * 16);
    v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int a = (int)(int8_t)((w[i] >> (8 * b)) & 0xFF);
        part += a * a;
      }
    }
  }
  *reinterpret_cast<uint4*>(tiles + (size_t)gt * kTileBytes + c * 16) = v;
  s_norm[m][kchunk] = part;
  __syncthreads();
  if (threadIdx.x < kTileRows) {
    const int mm = threadIdx.x;
    const uint32_t r = t * kTileRows + mm;
    int nn = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) nn += s_norm[mm][i];
    const bool valid = r < n;
    const int idx8 = (int)((t % kWinTiles) * kTileRows + mm);
    rconst[(size_t)gt * kTileRows + mm] = valid ? (-(nn << 8) - idx8) : kRPad;
    qnorm[(size_t)gt * kTileRows + mm] = valid ? nn : 0;
  }
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
  const uint32_t nI = p.img_n[I], nJ = p.img_n[J];
  const uint32_t tileI0 = p.img_tile_off[I], tileJ0 = p.img_tile_off[J];
  const int ntI = (int)((nI + kTileRows - 1) / kTileRows);
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

  // ratio test + output (lanes 0..31 own one query each per tile)
#pragma unroll
  for (int n = 0; n < kNQ; ++n) {
    const uint32_t q = (qt0 + n) * kTileRows + (lane & 31);
    const bool inb = (lane < 32) && (q < nJ);
    bool ok = false;
    if (inb) {
      const int nq = p.qnorm[(size_t)tileJ0 * kTileRows + q];
      const int d0 = G1[n] + nq, d1 = G2[n] + nq;
      ok = __int2float_rn(d0) < __fmul_rn(p.ratio_sq, __int2float_rn(d1));
      p.best[(size_t)wk.x * p.qstride + q] = ok ? (uint32_t)Gi[n] : kNoMatch;
    }
    const unsigned long long m = __ballot(ok);
    if (lane == 0 && m) atomicAdd(&p.count[wk.x], (uint32_t)__popcll(m));
  }
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
    p.best[(size_t)wk.x * p.qstride + q] = ok ? i0 : kNoMatch;
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
                                                              uint32_t n_pairs, uint32_t qstride,
                                                              uint2* __restrict__ out_ij) {
  const uint32_t pidx = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per image pair
  if (pidx >= n_pairs) return;
  const int lane = threadIdx.x & 63;
  const uint32_t off = offsets[pidx];
  if (offsets[pidx + 1] == off) return;
  const uint32_t nJ = img_n[pairs[pidx].y];
  uint32_t run = off;
  for (uint32_t q0 = 0; q0 < nJ; q0 += 64) {
    const uint32_t q = q0 + lane;
    const uint32_t v = (q < nJ) ? best[(size_t)pidx * qstride + q] : kNoMatch;
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
    MVGX_HIP(hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T)));
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
  void release() { if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; } }
};

}  // namespace

struct mvgx_match_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_total0 = nullptr, ev_total1 = nullptr;
  // options
  int variant = 1;          // 0 naive, 1 MFMA + register-staged LDS, 2 MFMA + LDS-DMA builtin, 3 MFMA + LDS-DMA asm
  int profile = 0;
  int64_t batch_pairs = 1 << 17;
  int keep_host_results = 1;
  // regions
  uint32_t n_images = 0;
  uint32_t total_tiles = 0;
  uint32_t max_tiles_pad = 0;
  uint32_t qstride = 0;
  std::vector<uint32_t> h_n, h_tile_off;
  std::vector<uint64_t> h_row_off;
  DevBuf<uint8_t> d_rows;
  bool rows_owned = true;
  const uint8_t* d_rows_view = nullptr;
  DevBuf<int8_t> d_tiles;
  DevBuf<int> d_rconst, d_qnorm;
  DevBuf<uint64_t> d_row_off;
  DevBuf<uint32_t> d_tile_off, d_n;
  // batch scratch
  DevBuf<uint2> d_pairs, d_work, d_ij;
  DevBuf<uint32_t> d_best, d_count, d_offsets;
  PinnedBuf<uint2> hp_pairs, hp_work;
  PinnedBuf<uint32_t> hp_offsets;
  // results of the last run
  std::vector<uint64_t> res_offsets;
  std::vector<uint32_t> res_ij;
  std::vector<hipEvent_t> ev_pool;
};

namespace {

int prep_regions(mvgx_match_ctx* c) {
  const uint32_t n_images = c->n_images;
  c->h_tile_off.assign(n_images + 1, 0);
  c->h_row_off.assign(n_images + 1, 0);
  uint32_t max_pad = 0, max_n = 0;
  for (uint32_t k = 0; k < n_images; ++k) {
    const uint32_t nt = (c->h_n[k] + kTileRows - 1) / kTileRows;
    const uint32_t pad = (nt + kWinTiles - 1) / kWinTiles * kWinTiles;
    c->h_tile_off[k + 1] = c->h_tile_off[k] + pad;
    c->h_row_off[k + 1] = c->h_row_off[k] + c->h_n[k];
    max_pad = std::max(max_pad, pad);
    max_n = std::max(max_n, c->h_n[k]);
  }
  c->total_tiles = c->h_tile_off[n_images];
  c->max_tiles_pad = max_pad;
  c->qstride = (max_n + kTileRows - 1) / kTileRows * kTileRows;
  const size_t alloc_tiles = (size_t)c->total_tiles + kTailTiles;
  int rc;
  if ((rc = c->d_tiles.ensure(alloc_tiles * kTileBytes))) return rc;
  if ((rc = c->d_rconst.ensure(alloc_tiles * kTileRows))) return rc;
  if ((rc = c->d_qnorm.ensure(alloc_tiles * kTileRows))) return rc;
  if ((rc = c->d_row_off.ensure(n_images + 1))) return rc;
  if ((rc = c->d_tile_off.ensure(n_images + 1))) return rc;
  if ((rc = c->d_n.ensure(n_images + 1))) return rc;
  MVGX_HIP(hipMemcpyAsync(c->d_row_off.p, c->h_row_off.data(), (n_images + 1) * sizeof(uint64_t),
                          hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipMemcpyAsync(c->d_tile_off.p, c->h_tile_off.data(), (n_images + 1) * sizeof(uint32_t),
                          hipMemcpyHostToDevice, c->stream));
  MVGX_HIP(hipMemcpyAsync(c->d_n.p, c->h_n.data(), n_images * sizeof(uint32_t), hipMemcpyHostToDevice,
                          c->stream));
  // slack tiles: zero data (their results are never written)
  MVGX_HIP(hipMemsetAsync(c->d_tiles.p + (size_t)c->total_tiles * kTileBytes, 0, (size_t)kTailTiles * kTileBytes,
                          c->stream));
  if (c->total_tiles > 0 && max_pad > 0) {
    dim3 grid(max_pad, n_images);
    hipLaunchKernelGGL(prep_tiles_kernel, grid, dim3(256), 0, c->stream, c->d_rows_view, c->d_row_off.p,
                       c->d_tile_off.p, c->d_n.p, c->d_tiles.p, c->d_rconst.p, c->d_qnorm.p);
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

int mvgx_match_create(int device, mvgx_match_ctx** out) {
  MVGX_REQUIRE(out != nullptr, MVGX_ERR_ARG, "mvgx_match_create: out is NULL");
  int rc = mvgx::select_device(device);
  if (rc) return rc;
  auto* c = new mvgx_match_ctx();
  MVGX_HIP(hipGetDevice(&c->device));
  MVGX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  MVGX_HIP(hipEventCreate(&c->ev_total0));
  MVGX_HIP(hipEventCreate(&c->ev_total1));
  // 2 x 33 KiB dynamic LDS for both MFMA variants
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_top2_ratio_kernel<kStageRegs>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_top2_ratio_kernel<kStageGlds>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  MVGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&l2_top2_ratio_kernel<kStageGldsAsm>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
  *out = c;
  return MVGX_OK;
}

int mvgx_match_destroy(mvgx_match_ctx* c) {
  if (!c) return MVGX_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->d_rows.release(); c->d_tiles.release(); c->d_rconst.release(); c->d_qnorm.release();
  c->d_row_off.release(); c->d_tile_off.release(); c->d_n.release();
  c->d_pairs.release(); c->d_work.release(); c->d_ij.release();
  c->d_best.release(); c->d_count.release(); c->d_offsets.release();
  c->hp_pairs.release(); c->hp_work.release(); c->hp_offsets.release();
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->ev_total0) (void)hipEventDestroy(c->ev_total0);
  if (c->ev_total1) (void)hipEventDestroy(c->ev_total1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return MVGX_OK;
}

int mvgx_match_set_option(mvgx_match_ctx* c, const char* key, int64_t value) {
  MVGX_REQUIRE(c && key, MVGX_ERR_ARG, "mvgx_match_set_option: NULL argument");
  if (!strcmp(key, "variant")) {
    MVGX_REQUIRE(value >= 0 && value <= 3, MVGX_ERR_ARG, "variant must be 0..3");
    c->variant = (int)value;
  } else if (!strcmp(key, "profile")) {
    c->profile = value != 0;
  } else if (!strcmp(key, "batch_pairs")) {
    MVGX_REQUIRE(value >= 1 && value <= (1 << 20), MVGX_ERR_ARG, "batch_pairs must be in [1, 2^20]");
    c->batch_pairs = value;
  } else if (!strcmp(key, "keep_host_results")) {
    c->keep_host_results = value != 0;
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
  MVGX_HIP(hipSetDevice(c->device));
  c->n_images = n_images;
  c->h_n.assign(n_desc, n_desc + n_images);
  c->rows_owned = false;
  c->d_rows_view = static_cast<const uint8_t*>(d_desc_concat);
  return prep_regions(c);
}

int mvgx_match_run(mvgx_match_ctx* c, const uint32_t* pairs_IJ, uint64_t n_pairs, float ratio_sq,
                   mvgx_match_stats* stats) {
  MVGX_REQUIRE(c && (pairs_IJ || n_pairs == 0), MVGX_ERR_ARG, "mvgx_match_run: NULL argument");
  MVGX_REQUIRE(c->d_rows_view != nullptr || c->n_images == 0, MVGX_ERR_STATE, "mvgx_match_run before set_regions");
  MVGX_REQUIRE(ratio_sq <= 1.0f && ratio_sq >= 0.0f, MVGX_ERR_UNSUPPORTED,
               "ratio_sq = %g: the device path reproduces the reference only for 0 <= ratio^2 <= 1 "
               "(ties are libstdc++ partial_sort order beyond that)", (double)ratio_sq);
  MVGX_HIP(hipSetDevice(c->device));
  for (uint64_t k = 0; k < n_pairs; ++k)
    MVGX_REQUIRE(pairs_IJ[2 * k] < c->n_images && pairs_IJ[2 * k + 1] < c->n_images, MVGX_ERR_ARG,
                 "pair %llu references image out of range", (unsigned long long)k);

  c->res_offsets.assign(n_pairs + 1, 0);
  c->res_ij.clear();
  mvgx_match_stats st;
  memset(&st, 0, sizeof(st));
  st.variant = (uint32_t)c->variant;
  size_t n_ev = 0;
  int rc;

  MVGX_HIP(hipEventRecord(c->ev_total0, c->stream));
  const uint64_t B = (uint64_t)c->batch_pairs;
  for (uint64_t p0 = 0; p0 < n_pairs; p0 += B) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(B, n_pairs - p0);
    if ((rc = c->hp_pairs.ensure(nb))) return rc;
    // worst-case work items: ceil(max tiles / 16) per pair
    const uint32_t max_blocks_per_pair = std::max<uint32_t>(1, (c->max_tiles_pad + kBlockQTiles - 1) / kBlockQTiles);
    if ((rc = c->hp_work.ensure((size_t)nb * max_blocks_per_pair))) return rc;
    uint32_t n_work = 0;
    for (uint32_t k = 0; k < nb; ++k) {
      const uint32_t I = pairs_IJ[2 * (p0 + k)], J = pairs_IJ[2 * (p0 + k) + 1];
      c->hp_pairs.p[k] = make_uint2(I, J);
      const uint32_t nI = c->h_n[I], nJ = c->h_n[J];
      // matcher_brute_force.hpp:108-113: NN(=2) > rows  -> no result; Matcher_Regions.cpp:65-69,85-90: empty regions skipped
      if (nI < 2 || nJ == 0) continue;
      const uint32_t ntJ = (nJ + kTileRows - 1) / kTileRows;
      for (uint32_t qt = 0; qt < ntJ; qt += kBlockQTiles) c->hp_work.p[n_work++] = make_uint2(k, qt);
      st.n_pairs += 1;
      st.n_desc_pairs += (uint64_t)nI * nJ;
    }
    if ((rc = c->d_pairs.ensure(nb))) return rc;
    if ((rc = c->d_work.ensure(std::max<uint32_t>(n_work, 1)))) return rc;
    if ((rc = c->d_best.ensure((size_t)nb * c->qstride))) return rc;
    if ((rc = c->d_count.ensure(nb))) return rc;
    if ((rc = c->d_offsets.ensure((size_t)nb + 1))) return rc;
    if ((rc = c->hp_offsets.ensure((size_t)nb + 1))) return rc;
    MVGX_HIP(hipMemcpyAsync(c->d_pairs.p, c->hp_pairs.p, nb * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
    if (n_work)
      MVGX_HIP(hipMemcpyAsync(c->d_work.p, c->hp_work.p, n_work * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
    MVGX_HIP(hipMemsetAsync(c->d_count.p, 0, nb * sizeof(uint32_t), c->stream));

    MatchParams mp;
    mp.tiles = c->d_tiles.p; mp.rconst = c->d_rconst.p; mp.qnorm = c->d_qnorm.p;
    mp.rows_u8 = c->d_rows_view; mp.img_row_off = c->d_row_off.p;
    mp.img_tile_off = c->d_tile_off.p; mp.img_n = c->d_n.p;
    mp.pairs = c->d_pairs.p; mp.work = c->d_work.p; mp.n_work = n_work;
    mp.best = c->d_best.p; mp.count = c->d_count.p; mp.qstride = c->qstride; mp.ratio_sq = ratio_sq;

    if (n_work) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (c->profile) {
        e0 = get_event(c, n_ev++); e1 = get_event(c, n_ev++);
        MVGX_REQUIRE(e0 && e1, MVGX_ERR_HIP, "hipEventCreate failed");
        MVGX_HIP(hipEventRecord(e0, c->stream));
      }
      if (c->variant == 0) {
        hipLaunchKernelGGL(l2_top2_ratio_naive_kernel, dim3(n_work), dim3(256), 0, c->stream, mp);
      } else if (c->variant == 1) {
        hipLaunchKernelGGL(l2_top2_ratio_kernel<kStageRegs>, dim3(n_work), dim3(256), 2 * kStageBytes, c->stream, mp);
      } else if (c->variant == 2) {
        hipLaunchKernelGGL(l2_top2_ratio_kernel<kStageGlds>, dim3(n_work), dim3(256), 2 * kStageBytes, c->stream, mp);
      } else {
        hipLaunchKernelGGL(l2_top2_ratio_kernel<kStageGldsAsm>, dim3(n_work), dim3(256), 2 * kStageBytes, c->stream, mp);
      }
      MVGX_HIP(hipGetLastError());
      if (c->profile) MVGX_HIP(hipEventRecord(e1, c->stream));
      st.n_kernel_launches += 1;
    }
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_count.p, nb, c->d_offsets.p);
    MVGX_HIP(hipGetLastError());
    MVGX_HIP(hipMemcpyAsync(c->hp_offsets.p, c->d_offsets.p, ((size_t)nb + 1) * sizeof(uint32_t),
                            hipMemcpyDeviceToHost, c->stream));
    MVGX_HIP(hipStreamSynchronize(c->stream));
    const uint32_t total = c->hp_offsets.p[nb];
    const uint64_t base = c->res_offsets[p0];
    for (uint32_t k = 0; k <= nb; ++k) c->res_offsets[p0 + k] = base + c->hp_offsets.p[k];
    st.n_matches += total;
    if (total) {
      if ((rc = c->d_ij.ensure(total))) return rc;
      hipLaunchKernelGGL(compact_matches_kernel, dim3((nb + 3) / 4), dim3(256), 0, c->stream, c->d_best.p,
                         c->d_offsets.p, c->d_pairs.p, c->d_n.p, nb, c->qstride, c->d_ij.p);
      MVGX_HIP(hipGetLastError());
      if (c->keep_host_results) {
        const size_t old = c->res_ij.size();
        c->res_ij.resize(old + (size_t)total * 2);
        MVGX_HIP(hipMemcpyAsync(c->res_ij.data() + old, c->d_ij.p, (size_t)total * sizeof(uint2),
                                hipMemcpyDeviceToHost, c->stream));
      }
      MVGX_HIP(hipStreamSynchronize(c->stream));
    }
  }
  MVGX_HIP(hipEventRecord(c->ev_total1, c->stream));
  MVGX_HIP(hipEventSynchronize(c->ev_total1));
  float ms = 0.f;
  MVGX_HIP(hipEventElapsedTime(&ms, c->ev_total0, c->ev_total1));
  st.total_ms = ms;
  if (c->profile) {
    for (size_t i = 0; i + 1 < n_ev; i += 2) {
      float k = 0.f;
      MVGX_HIP(hipEventElapsedTime(&k, c->ev_pool[i], c->ev_pool[i + 1]));
      st.kernel_ms += k;
    }
  }
  if (stats) *stats = st;
  return MVGX_OK;
}

int mvgx_match_results(mvgx_match_ctx* c, const uint64_t** offsets, const uint32_t** ij) {
  MVGX_REQUIRE(c && offsets && ij, MVGX_ERR_ARG, "mvgx_match_results: NULL argument");
  *offsets = c->res_offsets.data();
  *ij = c->res_ij.data();
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
      const uint64_t a = c->res_offsets[k], b = c->res_offsets[k + 1];
      if (b > a) sink(user, pairs_IJ[2 * k], pairs_IJ[2 * k + 1], c->res_ij.data() + 2 * a, (uint32_t)(b - a));
    }
  }
  mvgx_match_destroy(c);
  return rc;
}

}  // extern "C"
