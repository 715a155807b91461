// hipemu runtime (TEST INFRASTRUCTURE ONLY — see hip/hip_runtime.h): fiber scheduler + wave collectives, and the
// translation unit that compiles the BA solver's HIP source for the host.
#include <hip/hip_runtime.h>
#include <ucontext.h>

#include <cstdio>
#include <vector>

namespace hipemu {

thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;   // one emulated device per OS thread

namespace {
enum State { RUNNABLE, WAIT_WAVE, WAIT_BLOCK, DONE };
constexpr size_t kStack = 512 * 1024;
struct Fiber {
  ucontext_t ctx;
  State st = DONE;
  char* stack = nullptr;
};
thread_local std::vector<Fiber> g_fibers;
thread_local ucontext_t g_sched;
thread_local int g_cur = -1, g_n = 0;
thread_local const std::function<void()>* g_body = nullptr;
// wave-collective exchange slots, double-buffered: a lane may run ahead to its next collective (other buffer) while
// slower lanes still read this one; the buffer is reused only after another wave-wide rendezvous
thread_local double g_slot[2][1024];
thread_local double g_slot_b[2][1024];
thread_local unsigned char g_parity[1024];
// ballots: a lane that has already left the kernel must not vote, and a lane that runs ahead and finishes must still be
// counted by the slower lanes of the same ballot: votes are tagged with the lane's ballot sequence number
thread_local unsigned g_ballot_seq[1024], g_ballot_tag[2][1024];
// wave collectives met by every lane so far: the lanes of a wave that rendezvous must have met the same number - a collective in
// divergent code (undefined on the device, silent garbage here) is reported instead
thread_local unsigned g_coll_n[1024];
thread_local int g_site[1024];   // last MVGX_EMU_SITE(n) passed by the lane (optional source markers: lanes that rendezvous must agree)

void trampoline() {
  (*g_body)();
  g_fibers[g_cur].st = DONE;
  swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}

void yield(State s) {
  g_fibers[g_cur].st = s;
  swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}

void run_block(unsigned nthreads) {
  if (g_fibers.size() < nthreads) {
    const size_t old = g_fibers.size();
    g_fibers.resize(nthreads);
    for (size_t i = old; i < nthreads; ++i) g_fibers[i].stack = static_cast<char*>(malloc(kStack));
  }
  g_n = (int)nthreads;
  for (int i = 0; i < g_n; ++i) {
    Fiber& f = g_fibers[i];
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &g_sched;
    makecontext(&f.ctx, trampoline, 0);
    f.st = RUNNABLE;
    g_parity[i] = 0;
    g_ballot_seq[i] = 0;
    g_coll_n[i] = 0;
    g_site[i] = 0;
    g_ballot_tag[0][i] = g_ballot_tag[1][i] = 0;
  }
  for (;;) {
    bool progressed = false;
    for (int i = 0; i < g_n; ++i) {
      if (g_fibers[i].st != RUNNABLE) continue;
      g_cur = i;
      g_threadIdx = dim3((unsigned)i, 0, 0);
      swapcontext(&g_sched, &g_fibers[i].ctx);
      progressed = true;
    }
    // release waves whose live lanes all wait at a wave collective
    bool all_done = true, all_block = true;
    for (int w0 = 0; w0 < g_n; w0 += 64) {
      bool any = false, ok = true;
      for (int i = w0; i < std::min(g_n, w0 + 64); ++i) {
        if (g_fibers[i].st == DONE) continue;
        any = true;
        if (g_fibers[i].st != WAIT_WAVE) ok = false;
      }
      if (any && ok) {
        int first = -1;
        for (int i = w0; i < std::min(g_n, w0 + 64); ++i) {
          if (g_fibers[i].st != WAIT_WAVE) continue;
          if (first < 0) first = i;
          if (g_site[i] != g_site[first]) {
            fprintf(stderr, "hipemu: lanes %d and %d of a wave rendezvous at different places (site %d / %d, collectives %u / %u)\n", first & 63, i & 63, g_site[first], g_site[i], g_coll_n[first], g_coll_n[i]);
            abort();
          }
          if (g_parity[i] != g_parity[first]) {
            fprintf(stderr, "hipemu: exchange-buffer parity of lane %d differs from lane %d's (collectives %u / %u)\n", i & 63, first & 63, g_coll_n[i], g_coll_n[first]);
            abort();
          }
          if (g_coll_n[i] != g_coll_n[first]) {
            fprintf(stderr, "hipemu: wave collective in divergent code (block %u): lane %d is at its collective number %u, lane %d at %u\n",
                    g_blockIdx.x, first & 63, g_coll_n[first], i & 63, g_coll_n[i]);
            abort();
          }
        }
        for (int i = w0; i < std::min(g_n, w0 + 64); ++i)
          if (g_fibers[i].st == WAIT_WAVE) g_fibers[i].st = RUNNABLE;
        progressed = true;
      }
    }
    for (int i = 0; i < g_n; ++i) {
      if (g_fibers[i].st != DONE) all_done = false;
      if (g_fibers[i].st != DONE && g_fibers[i].st != WAIT_BLOCK) all_block = false;
    }
    if (all_done) break;
    if (all_block) {
      for (int i = 0; i < g_n; ++i)
        if (g_fibers[i].st == WAIT_BLOCK) g_fibers[i].st = RUNNABLE;
      progressed = true;
    }
    if (!progressed) {
      fprintf(stderr, "hipemu: deadlock (divergent barrier / wave collective) in block %u of %u x %u threads; fiber states"
              " (R runnable, w wave collective, B block barrier, . done):\n", g_blockIdx.x, g_gridDim.x, g_blockDim.x);
      for (int i = 0; i < g_n; ++i) {
        fputc("RwB."[g_fibers[i].st], stderr);
        if ((i & 63) == 63) fputc('\n', stderr);
      }
      fputc('\n', stderr);
      abort();
    }
  }
}
}  // namespace

}  // namespace hipemu

struct hipemuGraph {
  struct Node { dim3 grid, block; std::function<void()> body; };
  std::vector<Node> nodes;
  int refs = 1;
};

namespace hipemu {
namespace { thread_local hipemuGraph* g_capture = nullptr; }

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  if (g_capture) { g_capture->nodes.push_back({grid, block, body}); return; }
  g_body = &body;
  g_gridDim = grid;
  g_blockDim = block;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = dim3(bx, by, bz);
        run_block(block.x);   // one-dimensional workgroups only (all the solver uses)
      }
  g_body = nullptr;
}

void site(int n) { g_site[g_cur] = n; }

void block_barrier() {
  const int me = g_cur;
  yield(WAIT_BLOCK);
  g_cur = me;
  g_threadIdx = dim3((unsigned)me, 0, 0);
}

static void wave_sync() {
  const int me = g_cur;
  ++g_coll_n[me];
  yield(WAIT_WAVE);
  g_cur = me;
  g_threadIdx = dim3((unsigned)me, 0, 0);
}

int lane_xor(int lane, int m) { return lane ^ m; }
int lane_down(int lane, int d) { return lane + d; }
int lane_abs(int, int s) { return s; }

double shfl_f64(double v, int src_lane_of(int, int), int arg) {
  const int me = g_cur, w0 = me & ~63, lane = me & 63, par = g_parity[me];
  g_parity[me] ^= 1;
  g_slot[par][me] = v;
  wave_sync();
  const int src = src_lane_of(lane, arg);
  double r = v;
  if (src >= 0 && src < 64 && w0 + src < g_n) r = g_slot[par][w0 + src];
  return r;
}

int readlane_i32(int v, int src_lane) {
  double d = 0;
  memcpy(&d, &v, sizeof(int));
  d = shfl_f64(d, lane_abs, src_lane);
  int r;
  memcpy(&r, &d, sizeof(int));
  return r;
}

// v_mfma_f64_16x16x4_f64: lane l feeds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; receives
// D[i = (l >> 4) + 4 reg][j = l & 15], reg = 0..3   (cdna_hip_programming.md, "f64 MFMA does NOT use these maps")
d4 mfma_f64_16x16x4(double a, double b, d4 c, int, int, int) {
  const int me = g_cur, w0 = me & ~63, lane = me & 63, par = g_parity[me];
  g_parity[me] ^= 1;
  g_slot[par][me] = a;
  g_slot_b[par][me] = b;
  wave_sync();
  d4 out = c;
  const int j = lane & 15;
  for (int reg = 0; reg < 4; ++reg) {
    const int i = (lane >> 4) + 4 * reg;
    double s = 0;
    for (int k = 0; k < 4; ++k) s += g_slot[par][w0 + k * 16 + i] * g_slot_b[par][w0 + k * 16 + j];
    out[reg] += s;
  }
  return out;
}

// v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 products. Lane l feeds A_b[i = l & 3][k = l >> 4] and B_b[k = l >> 4][j = l & 3] of
// block b = (l & 15) >> 2 and receives D_b[i = l >> 4][j = l & 3] of the same block - read off the device with tools/mfma_f64_4x4x4_probe.hip
// (profiles/round6_mfma_f64_4x4x4_probe_call_r6_35.txt).
double mfma_f64_4x4x4(double a, double b, double c, int, int, int) {
  const int me = g_cur, w0 = me & ~63, lane = me & 63, par = g_parity[me];
  g_parity[me] ^= 1;
  g_slot[par][me] = a;
  g_slot_b[par][me] = b;
  wave_sync();
  const int i = lane >> 4, blk = (lane & 15) >> 2, j = lane & 3;
  double s = 0;
  for (int k = 0; k < 4; ++k) s += g_slot[par][w0 + 16 * k + 4 * blk + i] * g_slot_b[par][w0 + 16 * k + 4 * blk + j];
  return c + s;
}

// v_mfma_i32_32x32x32_i8 (gfx950): lane l feeds 16 bytes of A row i = l & 31 and of B column j = l & 31, covering
// k = 16 (l >> 5) .. + 15; it receives D[i = 8 (r >> 2) + 4 (l >> 5) + (r & 3)][j = l & 31], r = 0..15
// (MI355X_MICROARCH.md, 32x32 accumulator layout; the matching kernels were validated bit for bit on the device with it)
thread_local unsigned char g_a16[2][1024][16], g_b16[2][1024][16];
v16i_t mfma_i32_32x32x32_i8(v4i_t a, v4i_t b, v16i_t c, int, int, int) {
  const int me = g_cur, w0 = me & ~63, lane = me & 63, par = g_parity[me];
  g_parity[me] ^= 1;
  memcpy(g_a16[par][me], &a, 16);
  memcpy(g_b16[par][me], &b, 16);
  wave_sync();
  v16i_t out = c;
  const int j = lane & 31;
  for (int r = 0; r < 16; ++r) {
    const int i = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
    int s = 0;
    for (int k = 0; k < 32; ++k)
      s += (int)(signed char)g_a16[par][w0 + i + 32 * (k >> 4)][k & 15] * (int)(signed char)g_b16[par][w0 + j + 32 * (k >> 4)][k & 15];
    out[r] += s;
  }
  return out;
}

// v_mfma_i32_16x16x64_i8 (gfx950): lane l feeds 16 bytes of A row i = l & 15 and of B column j = l & 15, covering
// k = 16 (l >> 4) .. + 15; it receives D[i = 4 (l >> 4) + r][j = l & 15], r = 0..3 (the 16x16 accumulator layout of 32-bit results)
v4i_t mfma_i32_16x16x64_i8(v4i_t a, v4i_t b, v4i_t c, int, int, int) {
  const int me = g_cur, w0 = me & ~63, lane = me & 63, par = g_parity[me];
  g_parity[me] ^= 1;
  memcpy(g_a16[par][me], &a, 16);
  memcpy(g_b16[par][me], &b, 16);
  wave_sync();
  v4i_t out = c;
  const int j = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (lane >> 4) + r;
    int s = 0;
    for (int k = 0; k < 64; ++k)
      s += (int)(signed char)g_a16[par][w0 + i + 16 * (k >> 4)][k & 15] * (int)(signed char)g_b16[par][w0 + j + 16 * (k >> 4)][k & 15];
    out[r] += s;
  }
  return out;
}

unsigned long long ballot(bool pred) {
  const int me = g_cur, w0 = me & ~63, par = g_parity[me];
  g_parity[me] ^= 1;
  const unsigned seq = ++g_ballot_seq[me];
  g_slot[par][me] = pred ? 1.0 : 0.0;
  g_ballot_tag[par][me] = seq;
  wave_sync();
  unsigned long long m = 0;
  for (int l = 0; l < 64 && w0 + l < g_n; ++l)
    if (g_ballot_tag[par][w0 + l] == seq && g_slot[par][w0 + l] != 0.0) m |= 1ull << l;
  return m;
}

// DPP with full row / bank masks for the controls the kernels use: quad_perm (0x00-0xFF), row_mirror (0x140),
// row_half_mirror (0x141) - permutations inside a row of 16 lanes, so `old` / bound_ctrl never apply
int update_dpp(int old, int src, int dpp_ctrl, int, int, bool bound_ctrl) {
  const int lane = g_cur & 63;
  int from;
  if (dpp_ctrl >= 0x111 && dpp_ctrl <= 0x11F) {   // row_shr:n - lane i of a 16-lane row reads lane i - n of the same row; below the row's start: 0 (bound_ctrl) or `old`
    const int n = dpp_ctrl & 15;
    const bool valid = (lane & 15) >= n;
    const int v = readlane_i32(src, valid ? lane - n : lane);   // (every lane takes part in the exchange)
    return valid ? v : (bound_ctrl ? 0 : old);
  }
  if (dpp_ctrl >= 0 && dpp_ctrl <= 0xFF) from = (lane & ~3) + ((dpp_ctrl >> (2 * (lane & 3))) & 3);
  else if (dpp_ctrl == 0x140) from = (lane & ~15) + (15 - (lane & 15));
  else if (dpp_ctrl == 0x141) from = (lane & ~7) + (7 - (lane & 7));
  else { fprintf(stderr, "hipemu: DPP control 0x%x not emulated\n", dpp_ctrl); abort(); }
  (void)old;
  return readlane_i32(src, from);
}

}  // namespace hipemu

hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { hipemu::g_capture = new hipemuGraph(); return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = hipemu::g_capture; hipemu::g_capture = nullptr; return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) { ++g->refs; *e = g; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
  for (auto& n : e->nodes) hipemu::launch(n.grid, n.block, n.body);
  return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) { if (g && --g->refs == 0) delete g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { return hipGraphDestroy(e); }

#ifndef HIPEMU_NO_PRODUCT
// ---- the product's device + host code, compiled for the host against the shim ----
namespace {   // the kernels' `extern __shared__` arrays (same unnamed namespace as the kernels below)
thread_local __attribute__((aligned(16))) double panel[32768];
thread_local __attribute__((aligned(16))) double zs[32768];
thread_local __attribute__((aligned(16))) double lds[32768];
thread_local __attribute__((aligned(16))) uint32_t lds_u32[65536];
}  // namespace
#include "mvgx_common.hip"
namespace mvgx {   // no RCCL in the emulation: the callback transport of mvgx_ba_set_allreduce covers multi-rank tests
struct RcclComm {};
int rccl_unique_id(void*) { set_error("hipemu: no RCCL"); return MVGX_ERR_UNSUPPORTED; }
int rccl_init(RcclComm**, int, int, const void*) { set_error("hipemu: no RCCL"); return MVGX_ERR_UNSUPPORTED; }
void rccl_destroy(RcclComm*) {}
void rccl_abort(RcclComm*) {}
int rccl_allreduce_f64(RcclComm*, double*, uint64_t, int, hipStream_t) { return MVGX_ERR_UNSUPPORTED; }
int rccl_self_check(RcclComm*, hipStream_t) { return MVGX_ERR_UNSUPPORTED; }
}  // namespace mvgx
extern "C" int mvgx_comm_unique_id(void* out) { return mvgx::rccl_unique_id(out); }
#include "mvgx_ba.hip"
#include "mvgx_ba_multi.hip"
#include "mvgx_bruteforce.hip"
#include "mvgx_geofilter.hip"
#include "mvgx_guided.hip"
#endif  // HIPEMU_NO_PRODUCT
