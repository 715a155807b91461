// libmvgx_hip.so — bundle adjustment on several devices of ONE process (north_star: "BA residual/Jacobian
// evaluation shards observations across GPUs with an ... all-reduce ... of the reduced camera blocks"), so that an unchanged
// single-process caller (openMVG's main_SfM through Bundle_Adjustment_Ceres::Adjust of the replacement TU) reaches every
// GPU of the node. The one-process-per-GPU form (mvgx_ba_comm_init / mvgx_ba_set_allreduce on per-rank contexts, used by
// bench.py under torch.distributed.run) stays as it is; this file builds the same thing inside one process:
//
//   * the problem is cut exactly like openmvg_amd/sharding.py cuts it: every device gets ALL poses and intrinsics and a
//     disjoint subset of the points with ALL their observations (a point must be local to be eliminated, ceres
//     schur_eliminator_impl.h:114-151), balanced by sum L_p^2 (longest-processing-time dealing of the track-length classes);
//     control points travel with their points, pose-centre priors live on shard 0 only;
//   * one ordinary single-device context per shard, one host thread per device for every call;
//   * the exchange step: RCCL (ncclAllReduce on each context's stream; one communicator per device, created by the device
//     threads from one unique id) when every ordinal is distinct and librccl loads - or the in-process PEER transport
//     below: the devices' buffers are peer-mapped over xGMI (hipDeviceEnablePeerAccess) and every rank sums the slices of
//     all ranks in rank order with a plain kernel, so all ranks obtain bit-identical sums (they then factor the same reduced
//     system). MVGX_BA_TRANSPORT=rccl|peer forces either. The peer form also serves several contexts on one device, which is
//     how the single-GPU tests run this path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "mvgx_ba_multi.h"
#include "mvgx_common.h"

namespace mvgx {
namespace {

constexpr int kMaxPeers = 16;

// ---------------------------------------------------------------------------------------------------------------------
// peer transport
// ---------------------------------------------------------------------------------------------------------------------
struct Barrier {   // reusable, abortable: a rank that fails wakes the others with an error instead of leaving them waiting
  std::mutex mu;
  std::condition_variable cv;
  int n = 1, waiting = 0;
  uint64_t generation = 0;
  bool aborted = false;
  bool wait() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const uint64_t gen = generation;
    if (++waiting == n) {
      waiting = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    cv.wait(lk, [&]() { return generation != gen || aborted; });
    return !aborted;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = true;
    cv.notify_all();
  }
};

struct PeerPtrs { double* p[kMaxPeers]; };

struct PeerGroup {
  int n = 1;
  Barrier bar;
  PeerPtrs cur;                      // buffers of the collective in flight, one per rank
  std::vector<double*> tmp;          // per rank: private scratch on its device
  std::vector<size_t> tmp_cap;
};
struct PeerRank { PeerGroup* g; int rank; };

// out[i] = p[0][i] (+|max) p[1][i] ... in rank order, i in [lo, hi): every rank computes the same bits
__global__ void peer_reduce_kernel(PeerPtrs pp, int n, double* __restrict__ out, uint64_t lo, uint64_t hi, int op) {
  for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (uint64_t)gridDim.x * blockDim.x) {
    double acc = pp.p[0][i];
    for (int r = 1; r < n; ++r) {
      const double v = pp.p[r][i];
      acc = op == MVGX_REDUCE_MAX ? (v > acc ? v : acc) : acc + v;
    }
    out[i] = acc;
  }
}
// all-gather of the reduced slices: slice s of rank s's buffer -> this rank's buffer
__global__ void peer_gather_kernel(PeerPtrs pp, int n, int self, uint64_t count) {
  for (int s = 0; s < n; ++s) {
    if (s == self) continue;
    const uint64_t lo = count * s / n, hi = count * (s + 1) / n;
    for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (uint64_t)gridDim.x * blockDim.x)
      pp.p[self][i] = pp.p[s][i];
  }
}

constexpr uint64_t kPeerSliceMin = 1u << 20;   // doubles: below this every rank reduces the whole vector itself (2 barriers)

// the mvgx_allreduce_f64 callback of a shard's context; runs on that shard's host thread with its device current
int peer_allreduce(void* user, void* device_buffer, uint64_t count, int op, void* hip_stream) {
  PeerRank* me = static_cast<PeerRank*>(user);
  PeerGroup* g = me->g;
  const int r = me->rank, n = g->n;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  double* buf = static_cast<double*>(device_buffer);
  auto fail = [&](const char* what, hipError_t e) {
    set_error("peer all-reduce (rank %d of %d): %s -> %s", r, n, what, hipGetErrorString(e));
    g->bar.abort();
    return 1;
  };
  hipError_t e;
  g->cur.p[r] = buf;
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("hipStreamSynchronize", e);   // my contribution is complete
  if (!g->bar.wait()) return 1;                                                                    // ... and so is everybody's
  const PeerPtrs pp = g->cur;
  if (count < kPeerSliceMin) {
    if (g->tmp_cap[r] < count) {
      if (g->tmp[r]) (void)hipFree(g->tmp[r]);
      g->tmp[r] = nullptr; g->tmp_cap[r] = 0;
      const size_t want = std::max<uint64_t>(count, 4096);
      if ((e = device_malloc(reinterpret_cast<void**>(&g->tmp[r]), want * sizeof(double))) != hipSuccess) return fail("hipMalloc", e);
      g->tmp_cap[r] = want;
    }
    const unsigned grid = (unsigned)std::min<uint64_t>(1024, (count + 255) / 256);
    hipLaunchKernelGGL(peer_reduce_kernel, dim3(std::max(1u, grid)), dim3(256), 0, stream, pp, n, g->tmp[r], (uint64_t)0, count, op);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("peer_reduce_kernel", e);
    if (!g->bar.wait()) return 1;   // nobody reads my buffer any more: the sum may replace it
    if ((e = hipMemcpyAsync(buf, g->tmp[r], count * sizeof(double), hipMemcpyDeviceToDevice, stream)) != hipSuccess)
      return fail("hipMemcpyAsync", e);
    return 0;
  }
  // reduce-scatter (slice r in place: only rank r reads or writes slice r of any buffer in this step) + all-gather
  const uint64_t lo = count * r / n, hi = count * (r + 1) / n;
  if (hi > lo) {
    const unsigned grid = (unsigned)std::min<uint64_t>(2048, (hi - lo + 255) / 256);
    hipLaunchKernelGGL(peer_reduce_kernel, dim3(grid), dim3(256), 0, stream, pp, n, buf, lo, hi, op);
  }
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("peer_reduce_kernel", e);
  if (!g->bar.wait()) return 1;
  hipLaunchKernelGGL(peer_gather_kernel, dim3(2048), dim3(256), 0, stream, pp, n, r, count);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail("peer_gather_kernel", e);
  if (!g->bar.wait()) return 1;   // every rank has fetched my slice: my buffer is mine again
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// sharding (same rule as openmvg_amd/sharding.py: assign_points + shard_ba_scene)
// ---------------------------------------------------------------------------------------------------------------------
void assign_points(const mvgx_ba_problem& p, int world, std::vector<int32_t>& owner) {
  const uint32_t np = p.n_points;
  owner.assign(np, 0);
  if (world <= 1 || !np) return;
  std::vector<uint32_t> L(np, 0);
  for (uint64_t k = 0; k < p.n_obs; ++k) L[p.obs_point[k]]++;
  const uint32_t maxL = *std::max_element(L.begin(), L.end());
  // points grouped by track length, heaviest class first; a class is dealt round-robin starting at the lightest shard
  std::vector<uint32_t> start(maxL + 2, 0);
  for (uint32_t q = 0; q < np; ++q) start[maxL - L[q] + 1]++;
  for (uint32_t l = 0; l <= maxL; ++l) start[l + 1] += start[l];
  std::vector<uint32_t> order(np), fill(start.begin(), start.end() - 1);
  for (uint32_t q = 0; q < np; ++q) order[fill[maxL - L[q]]++] = q;
  std::vector<double> load(world, 0.0);
  std::vector<int> ranks(world);
  for (uint32_t cls = 0; cls <= maxL; ++cls) {
    const uint32_t a = start[cls], b = start[cls + 1];
    if (a == b) continue;
    const double cost = (double)(maxL - cls) * (double)(maxL - cls) + 1.0;
    std::iota(ranks.begin(), ranks.end(), 0);
    std::stable_sort(ranks.begin(), ranks.end(), [&](int x, int y) { return load[x] < load[y]; });
    for (uint32_t k = a; k < b; ++k) {
      const int rk = ranks[(k - a) % world];
      owner[order[k]] = rk;
      load[rk] += cost;
    }
  }
}

struct Shard {
  std::vector<uint32_t> pt_global;    // local point -> global point
  std::vector<uint64_t> obs_global;   // local observation -> global observation
  std::vector<double> points, obs_xy, obs_weight;
  std::vector<uint32_t> obs_pose, obs_intr, obs_point;
  std::vector<uint8_t> obs_is_control, point_const_mask;
};

}  // namespace

struct BaMulti {
  int n = 0;
  std::vector<int> devices;
  std::vector<mvgx_ba_ctx*> child;
  std::vector<Shard> shard;
  uint32_t n_points = 0;
  uint64_t n_obs = 0;
  BaFingerprint fingerprint;   // of the whole problem (mvgx_ba_update)
  bool use_rccl = false;
  bool comm_ready = false;   // the shards' communicators exist (RCCL transport): a failing shard aborts them
  std::mutex abort_mu;
  PeerGroup peers;
  std::vector<PeerRank> peer_rank;
};

namespace {

// runs fn(rank) on one host thread per shard; the first failing status is returned with its message
// A shard that fails must not leave the others waiting inside a collective: the peer transport's barrier is aborted; on the RCCL
// transport every communicator is aborted (ncclCommAbort fails the all-reduces in flight), after which the multi-device context
// can only be destroyed - its entry points report MVGX_ERR_STATE.
void abort_collectives(BaMulti* m) {
  m->peers.bar.abort();
  if (m->use_rccl && m->comm_ready) {
    std::lock_guard<std::mutex> lk(m->abort_mu);
    for (mvgx_ba_ctx* c : m->child) ba_ctx_comm_abort(c);
  }
}
template <class F>
int on_all(BaMulti* m, F fn) {
  std::vector<int> rc(m->n, MVGX_OK);
  std::vector<std::string> err(m->n);
  std::vector<std::thread> th;
  for (int r = 1; r < m->n; ++r)
    th.emplace_back([&, r]() {
      rc[r] = fn(r);
      if (rc[r]) { err[r] = mvgx_last_error(); abort_collectives(m); }
    });
  rc[0] = fn(0);
  if (rc[0]) { err[0] = mvgx_last_error(); abort_collectives(m); }
  for (auto& t : th) t.join();
  {   // a failed collective leaves the barrier aborted: re-arm it for the next call
    std::lock_guard<std::mutex> lk(m->peers.bar.mu);
    m->peers.bar.aborted = false;
    m->peers.bar.waiting = 0;
  }
  for (int r = 0; r < m->n; ++r)
    if (rc[r]) { set_error("device shard %d (device %d): %s", r, m->devices[r], err[r].c_str()); return rc[r]; }
  return MVGX_OK;
}

}  // namespace

void ba_multi_destroy(BaMulti* m) {
  if (!m) return;
  for (mvgx_ba_ctx* c : m->child) mvgx_ba_destroy(c);
  for (int r = 0; r < (int)m->peers.tmp.size(); ++r)
    if (m->peers.tmp[r]) { (void)hipSetDevice(m->devices[r]); (void)hipFree(m->peers.tmp[r]); }
  delete m;
}

int ba_multi_create(const int* devices, int n_devices, const mvgx_ba_problem* p, BaMulti** out) {
  MVGX_REQUIRE(devices && p && out && n_devices >= 2, MVGX_ERR_ARG, "ba_multi_create: bad argument");
  MVGX_REQUIRE(n_devices <= kMaxPeers, MVGX_ERR_ARG, "at most %d device shards", kMaxPeers);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    set_error("no HIP device visible (libmvgx_hip needs a gfx950 GPU; there is no CPU fallback)");
    return MVGX_ERR_NODEV;
  }
  for (int k = 0; k < n_devices; ++k)
    MVGX_REQUIRE(devices[k] >= 0 && devices[k] < count, MVGX_ERR_ARG, "device %d out of range (%d visible)", devices[k], count);
  for (uint64_t k = 0; k < p->n_obs; ++k)
    MVGX_REQUIRE(p->obs_point && p->obs_point[k] < p->n_points, MVGX_ERR_ARG, "observation %llu references a point out of range",
                 (unsigned long long)k);
  auto* m = new BaMulti();
  struct Guard { BaMulti* m; ~Guard() { if (m) ba_multi_destroy(m); } } guard{m};
  m->n = n_devices;
  m->devices.assign(devices, devices + n_devices);
  m->n_points = p->n_points;
  m->n_obs = p->n_obs;
  m->fingerprint = ba_fingerprint(p);
  m->shard.resize(m->n);
  m->child.assign(m->n, nullptr);
  // ---- shards ----
  std::vector<int32_t> owner;
  assign_points(*p, m->n, owner);
  std::vector<uint32_t> local_id(p->n_points);
  for (uint32_t q = 0; q < p->n_points; ++q) {
    Shard& s = m->shard[owner[q]];
    local_id[q] = (uint32_t)s.pt_global.size();
    s.pt_global.push_back(q);
  }
  for (int r = 0; r < m->n; ++r) {
    Shard& s = m->shard[r];
    const size_t np = s.pt_global.size();
    s.points.resize(3 * np);
    if (p->point_const_mask) s.point_const_mask.resize(np);
    for (size_t j = 0; j < np; ++j) {
      memcpy(&s.points[3 * j], p->points + 3 * (size_t)s.pt_global[j], 3 * sizeof(double));
      if (p->point_const_mask) s.point_const_mask[j] = p->point_const_mask[s.pt_global[j]];
    }
  }
  for (uint64_t k = 0; k < p->n_obs; ++k) {
    const uint32_t q = p->obs_point[k];
    Shard& s = m->shard[owner[q]];
    s.obs_global.push_back(k);
    s.obs_pose.push_back(p->obs_pose[k]);
    s.obs_intr.push_back(p->obs_intr[k]);
    s.obs_point.push_back(local_id[q]);
    s.obs_xy.push_back(p->obs_xy[2 * k]);
    s.obs_xy.push_back(p->obs_xy[2 * k + 1]);
    if (p->obs_weight) s.obs_weight.push_back(p->obs_weight[k]);
    if (p->obs_is_control) s.obs_is_control.push_back(p->obs_is_control[k]);
  }
  // ---- transport ----
  bool distinct = true;
  for (int a = 0; a < m->n; ++a)
    for (int b = a + 1; b < m->n; ++b) distinct = distinct && m->devices[a] != m->devices[b];
  const char* tr = getenv("MVGX_BA_TRANSPORT");
  uint8_t uid[128];
  if (tr && !strcmp(tr, "peer")) {
    m->use_rccl = false;
  } else if (tr && !strcmp(tr, "rccl")) {
    MVGX_REQUIRE(distinct, MVGX_ERR_ARG, "MVGX_BA_TRANSPORT=rccl needs distinct devices");
    const int rc = mvgx_comm_unique_id(uid);
    if (rc) return rc;
    m->use_rccl = true;
  } else {
    m->use_rccl = distinct && mvgx_comm_unique_id(uid) == MVGX_OK;
  }
  if (!m->use_rccl) {
    m->peers.n = m->n;
    m->peers.bar.n = m->n;
    m->peers.tmp.assign(m->n, nullptr);
    m->peers.tmp_cap.assign(m->n, 0);
    m->peer_rank.resize(m->n);
    for (int r = 0; r < m->n; ++r) m->peer_rank[r] = PeerRank{&m->peers, r};
    for (int a = 0; a < m->n; ++a)
      for (int b = 0; b < m->n; ++b) {
        if (m->devices[a] == m->devices[b]) continue;
        int can = 0;
        MVGX_HIP(hipDeviceCanAccessPeer(&can, m->devices[a], m->devices[b]));
        MVGX_REQUIRE(can, MVGX_ERR_UNSUPPORTED, "device %d cannot map the memory of device %d (peer transport)", m->devices[a], m->devices[b]);
        MVGX_HIP(hipSetDevice(m->devices[a]));
        const hipError_t e = hipDeviceEnablePeerAccess(m->devices[b], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) MVGX_HIP(e);
        (void)hipGetLastError();
      }
  }
  // ---- one context per shard, built by its own host thread (structure build + upload run side by side) ----
  const int rc = on_all(m, [&](int r) -> int {
    const Shard& s = m->shard[r];
    mvgx_ba_problem sp = *p;
    sp.n_points = (uint32_t)s.pt_global.size();
    sp.n_obs = s.obs_global.size();
    sp.points = s.points.data();
    sp.obs_pose = s.obs_pose.data(); sp.obs_intr = s.obs_intr.data(); sp.obs_point = s.obs_point.data(); sp.obs_xy = s.obs_xy.data();
    sp.obs_weight = p->obs_weight ? s.obs_weight.data() : nullptr;
    sp.obs_is_control = p->obs_is_control ? s.obs_is_control.data() : nullptr;
    sp.point_const_mask = p->point_const_mask ? s.point_const_mask.data() : nullptr;
    if (r != 0) {   // prior residuals touch replicated blocks only: exactly one shard may hold them
      sp.n_pose_priors = 0; sp.prior_pose = nullptr; sp.prior_center = nullptr; sp.prior_weight = nullptr;
    }
    ba_create_defer_plan(true);
    const int rc_ = mvgx_ba_create(m->devices[r], &sp, &m->child[r]);
    ba_create_defer_plan(false);
    return rc_;
  });
  if (rc) return rc;   // (a shard that cannot be built - e.g. out of memory on one device - ends the call here, before any rank waits in ncclCommInitRank)
  // ---- the transport, once every shard exists ----
  const int rc2 = on_all(m, [&](int r) -> int {
    if (m->use_rccl) return mvgx_ba_comm_init(m->child[r], m->n, r, uid);
    return mvgx_ba_set_allreduce(m->child[r], &peer_allreduce, &m->peer_rank[r]);
  });
  if (rc2) return rc2;
  m->comm_ready = true;
  guard.m = nullptr;
  *out = m;
  return MVGX_OK;
}

// mvgx_ba_update on a multi-device context: the same cut (it is a function of the fingerprinted structure), every shard's values
// gathered again from the caller's arrays, every child re-bound by its own host thread.
int ba_multi_update(BaMulti* m, const mvgx_ba_problem* p) {
  if (!(ba_fingerprint(p) == m->fingerprint)) {
    set_error("mvgx_ba_update: the problem's structure is not the one this context was created from");
    return MVGX_ERR_STRUCTURE;
  }
  return on_all(m, [&](int r) -> int {
    Shard& s = m->shard[r];
    for (size_t j = 0; j < s.pt_global.size(); ++j) memcpy(&s.points[3 * j], p->points + 3 * (size_t)s.pt_global[j], 3 * sizeof(double));
    for (size_t k = 0; k < s.obs_global.size(); ++k) {
      s.obs_xy[2 * k] = p->obs_xy[2 * s.obs_global[k]];
      s.obs_xy[2 * k + 1] = p->obs_xy[2 * s.obs_global[k] + 1];
      if (p->obs_weight) s.obs_weight[k] = p->obs_weight[s.obs_global[k]];
    }
    mvgx_ba_problem sp = *p;
    sp.n_points = (uint32_t)s.pt_global.size();
    sp.n_obs = s.obs_global.size();
    sp.points = s.points.data();
    sp.obs_pose = s.obs_pose.data(); sp.obs_intr = s.obs_intr.data(); sp.obs_point = s.obs_point.data(); sp.obs_xy = s.obs_xy.data();
    sp.obs_weight = p->obs_weight ? s.obs_weight.data() : nullptr;
    sp.obs_is_control = p->obs_is_control ? s.obs_is_control.data() : nullptr;
    sp.point_const_mask = p->point_const_mask ? s.point_const_mask.data() : nullptr;
    if (r != 0) { sp.n_pose_priors = 0; sp.prior_pose = nullptr; sp.prior_center = nullptr; sp.prior_weight = nullptr; }
    return mvgx_ba_update(m->child[r], &sp);
  });
}

int ba_multi_n_shards(const BaMulti* m) { return m->n; }
int ba_multi_transport_is_rccl(const BaMulti* m) { return m->use_rccl ? 1 : 0; }

int ba_multi_solve(BaMulti* m, const mvgx_ba_options* opt, mvgx_ba_summary* summary, bool one_iteration) {
  std::vector<mvgx_ba_summary> sum(m->n);
  const int rc = on_all(m, [&](int r) -> int {
    return one_iteration ? mvgx_ba_lm_iteration(m->child[r], opt, &sum[r]) : mvgx_ba_solve(m->child[r], opt, &sum[r]);
  });
  // the LM state is replicated: every shard reports the same iteration counts and costs; times are the slowest shard's
  if (summary && (rc == MVGX_OK || rc == MVGX_ERR_NUMERIC)) {
    *summary = sum[0];
    for (int r = 1; r < m->n; ++r) {
      summary->total_ms = std::max(summary->total_ms, sum[r].total_ms);
      summary->iter_ms_mean = std::max(summary->iter_ms_mean, sum[r].iter_ms_mean);
      summary->jacobian_ms = std::max(summary->jacobian_ms, sum[r].jacobian_ms);
      summary->schur_ms = std::max(summary->schur_ms, sum[r].schur_ms);
      summary->solve_ms = std::max(summary->solve_ms, sum[r].solve_ms);
      summary->backsub_ms = std::max(summary->backsub_ms, sum[r].backsub_ms);
      summary->cost_ms = std::max(summary->cost_ms, sum[r].cost_ms);
    }
  }
  return rc;
}

int ba_multi_evaluate(BaMulti* m, double* cost, double* rmse) {
  std::vector<double> c(m->n, 0.0), e(m->n, 0.0);
  const int rc = on_all(m, [&](int r) -> int { return mvgx_ba_evaluate(m->child[r], &c[r], &e[r]); });   // sums across the shards inside
  if (rc) return rc;
  if (cost) *cost = c[0];
  if (rmse) *rmse = e[0];
  return MVGX_OK;
}

int ba_multi_read_params(BaMulti* m, double* poses, double* intrinsics, double* points) {
  return on_all(m, [&](int r) -> int {
    const Shard& s = m->shard[r];
    std::vector<double> local(points ? 3 * s.pt_global.size() : 0);
    const int rc = mvgx_ba_read_params(m->child[r], r == 0 ? poses : nullptr, r == 0 ? intrinsics : nullptr, points ? local.data() : nullptr);
    if (rc) return rc;
    if (points)
      for (size_t j = 0; j < s.pt_global.size(); ++j) memcpy(points + 3 * (size_t)s.pt_global[j], &local[3 * j], 3 * sizeof(double));
    return MVGX_OK;
  });
}

int ba_multi_residuals(BaMulti* m, double* residual_norm) {
  return on_all(m, [&](int r) -> int {
    const Shard& s = m->shard[r];
    std::vector<double> local(s.obs_global.size());
    if (local.empty()) return MVGX_OK;
    const int rc = mvgx_ba_residuals(m->child[r], local.data());
    if (rc) return rc;
    for (size_t k = 0; k < local.size(); ++k) residual_norm[s.obs_global[k]] = local[k];
    return MVGX_OK;
  });
}

int ba_multi_track_angles(BaMulti* m, double* max_angle_deg) {
  return on_all(m, [&](int r) -> int {
    const Shard& s = m->shard[r];
    std::vector<double> local(s.pt_global.size());
    if (local.empty()) return MVGX_OK;
    const int rc = mvgx_ba_track_angles(m->child[r], local.data());
    if (rc) return rc;
    for (size_t j = 0; j < local.size(); ++j) max_angle_deg[s.pt_global[j]] = local[j];
    return MVGX_OK;
  });
}

int ba_multi_solver_info(BaMulti* m, mvgx_ba_solver_info* out) { return mvgx_ba_get_solver_info(m->child[0], out); }
int ba_multi_set_linear_solver(BaMulti* m, int kind) {   // (the shards plan on the union of their blocks: the same answer on each)
  for (int r = 0; r < m->n; ++r)
    if (const int rc = mvgx_ba_set_linear_solver(m->child[r], kind)) return rc;
  return MVGX_OK;
}

}  // namespace mvgx
