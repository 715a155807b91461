// Symbolic phase of the block-sparse reduced camera system (host code, no HIP types).
//
// What it replaces in the reference (paths under /root/reference/src/third_party/ceres-solver/internal/ceres):
//   schur_complement_solver.cc:241-347   SparseSchurComplementSolver: block-random-access storage of the non-zero camera
//                                        blocks of S, sparse Cholesky of the reduced system
//   reorder_program.cc:524-528           fill-reducing ordering of the camera blocks (AMD through the sparse backend)
//   eigensparse.cc:60-143                the symbolic + numeric factorisation (SimplicialLDLT) the vendored build ends in
// and openMVG/sfm/pipelines/sequential/sequential_SfM.cpp:1193-1205 (SPARSE_SCHUR above 100 poses).
//
// Design for the device (the numeric phase lives in mvgx_ba.hip):
//   * graph of camera blocks (poses: 6 columns, intrinsics: 8) from the non-zero blocks of S; blocks coupled to a large
//     share of all others (shared intrinsics) form a dense border that is ordered last;
//   * nested dissection of the rest by BFS level structures: separators split the graph into independent parts whose
//     factorisations run concurrently on the device - a sequential-capture scene is a (cyclic) band whose natural order
//     would be one long dependency chain;
//   * every part is padded to a multiple of 64 columns (identity diagonal), so a 64 x 64 tile never mixes two independent
//     parts; fill is computed on tiles; the elimination tree of the tile columns gives the level schedule: the columns
//     of one level are factored by one batched launch;
//   * per level three task lists: F (diagonal tiles to factor + invert), T (tiles below them: L_ik = A_ik L_kk^-T) and
//     U (16 x 16 sub-blocks of the target tiles A_ij -= sum_k L_ik L_jk^T, contributors in ascending k: a fixed summation
//     order, no atomics), plus the lists of the reverse sweep (back substitution);
//   * look-ahead (round 5): the contributions of the columns of level l - 1 to the DIAGONAL tile of a column of level l - the only
//     updates the next factorisation waits for - are applied by that column's factor workgroup itself, from A_kj and Linv_j
//     (pre_* lists); all other T / U tasks of level l - 1 then run BESIDE the factorisation of level l in the same launch (the U tasks
//     rebuild the strips of L they multiply from A and Linv, exactly as the T tasks form them). One launch per level instead of three.
#ifndef MVGX_BA_SPARSE_PLAN_H_
#define MVGX_BA_SPARSE_PLAN_H_

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace mvgx_sparse {

constexpr int kTile = 64;

struct GemmTask {   // one 16 x 16 sub-block of a tile: dst(bi, bj) (-)= sum over contributors [c0, c1) of X_rows(bi) Y_rows(bj)^T
  int32_t dst;      // slot of the destination tile
  int16_t bi, bj;   // sub-block row / column (0..3)
  int32_t c0, c1;   // contributor range in the pair list
  // A long contributor list (a border tile collects one contribution from EVERY column of a level) is cut into n_chunks tasks of
  // at most kSplitChunk contributors; chunk tasks leave their partial sums in scratch block `scratch + chunk`, and the one that
  // arrives last (counter `group`) adds them up in chunk order and applies them to the destination. n_chunks <= 1: an ordinary task.
  int16_t chunk = 0, n_chunks = 0;
  int32_t group = 0, scratch = 0;
};
constexpr int kSplitMin = 7, kSplitChunk = 4;   // lists of kSplitMin or more contributors are cut into chunks of kSplitChunk
constexpr int kBsInline = 6;
constexpr int kMaxPre = 6;   // look-ahead: a diagonal tile takes at most this many contributions of the previous level inside its factor workgroup
static_assert(kMaxPre < kSplitMin, "a deferred contributor list is never a split one");
struct SlotPair { int32_t a, b; };   // T: (slot of A_ik, tile column k -> Linv_k); U: (slot of L_ik, slot of L_jk)

struct PlanParams {
  int leaf_cols = 192;          // parts at most this wide are not dissected further
  double max_sep_frac = 0.34;   // a separator heavier than this share of its subgraph is refused (no dissection)
  double min_side_frac = 0.2;   // both sides of a separator must carry at least this share
  double sep_weight_slack = 1.25;   // separators up to this factor heavier than the lightest one compete on balance
  double dense_degree_factor = 4.0;   // border: degree > max(dense_degree_min, factor x median degree)
  int dense_degree_min = 16;
  int max_pre = kMaxPre;        // look-ahead: contributions of the level before a factor workgroup takes itself (<= kMaxPre; tests lower it)
};

struct Plan {
  int N = 0, N_pad = 0, nT = 0, n_slots = 0, n_levels = 0, n_parts = 0, n_border_blocks = 0;
  std::vector<int32_t> pcol;       // N: original scalar column -> padded, permuted column
  std::vector<int32_t> tmap;       // (nT + 1) x nT: tile (I, J), I >= J (row nT = rhs) -> slot, -1 = structurally zero
  std::vector<int32_t> tile_kb;    // nT: valid (non-padding) columns of the tile
  std::vector<int32_t> level_of;   // nT
  std::vector<int32_t> f_cols, f_start;                  // tile columns grouped by level; f_start[n_levels + 1]
  std::vector<GemmTask> t_tasks, u_tasks;                // grouped by level
  std::vector<int32_t> t_start, u_start;                 // [n_levels + 1]
  std::vector<SlotPair> t_pairs, u_pairs;
  std::vector<int32_t> bs_start, bs_slot, bs_row;        // per tile column: the factor tiles below it (slot, tile row)
  // the same per position in f_cols, as ONE 64-byte record a workgroup of the reverse sweep starts from (round 5: the sweep walked
  // f_cols -> bs_start / tmap -> the lists -> the tiles, one trip to memory each): [0] column, [1] slot of its rhs strip, [2] entries,
  // [3] first entry in bs_slot / bs_row, [4 + 2 i], [5 + 2 i]: (slot, row) of entry i < kBsInline
  std::vector<int32_t> bs_rec;                           // [nT][16]
  int n_split_groups = 0, n_scratch_blocks = 0;          // of the U tasks with split contributor lists (GemmTask)
  // look-ahead: lookahead[l] != 0 - the factor workgroups of level l apply the level-(l - 1) contributions to their diagonal tiles
  // themselves (pre_start[k] .. pre_start[k + 1]: slot of A_kj and column j, ascending j); the U tasks that would have done it are the
  // LAST ones of level l - 1 (u_defer_start[l - 1] .. u_start[l]): the level-by-level schedule runs them, the look-ahead schedule skips them
  std::vector<uint8_t> lookahead;                        // [n_levels]
  std::vector<int32_t> u_defer_start;                    // [n_levels]
  std::vector<int32_t> pre_start, pre_slot, pre_col;     // [nT + 1], entries
  std::vector<int32_t> slot_col;                         // [n_slots]: tile column of a slot (the rhs strip of column k counts as column k)
  uint64_t n_fill_tiles = 0;       // tiles of the factor (lower triangle incl. diagonal, without the rhs row)
  double flops = 0;                // multiply-adds x 2 of the numeric phase on the non-zero tiles
};

namespace detail {

struct Graph {
  int n = 0;
  std::vector<int32_t> start, adj;   // CSR, symmetric, no self loops
  std::vector<int32_t> w;            // columns per node
};

// BFS over the nodes whose stamp[v] == tag, from `root`; returns the visit order and fills level[] for visited nodes.
inline void bfs(const Graph& g, const std::vector<int32_t>& stamp, int32_t tag, int root, std::vector<int32_t>& order,
                std::vector<int32_t>& level, std::vector<int32_t>& seen, int32_t seen_tag) {
  order.clear();
  order.push_back(root);
  seen[root] = seen_tag;
  level[root] = 0;
  for (size_t h = 0; h < order.size(); ++h) {
    const int v = order[h];
    for (int32_t e = g.start[v]; e < g.start[v + 1]; ++e) {
      const int u = g.adj[e];
      if (stamp[u] != tag || seen[u] == seen_tag) continue;
      seen[u] = seen_tag;
      level[u] = level[v] + 1;
      order.push_back(u);
    }
  }
}

struct Part { std::vector<int32_t> nodes; int parent = -1; };

struct Dissector {
  const Graph& g;
  const PlanParams& prm;
  std::vector<Part>& parts;
  std::vector<int32_t> stamp, seen, level;
  int32_t next_tag = 1, next_seen = 1;
  Dissector(const Graph& g_, const PlanParams& p, std::vector<Part>& out)
      : g(g_), prm(p), parts(out), stamp(g_.n, 0), seen(g_.n, 0), level(g_.n, 0) {}

  int deg_in(int v, int32_t tag) const {
    int dgr = 0;
    for (int32_t e = g.start[v]; e < g.start[v + 1]; ++e) dgr += stamp[g.adj[e]] == tag;
    return dgr;
  }

  // nodes: one connected component of the current subgraph
  void dissect(std::vector<int32_t> nodes, int parent) {
    const int32_t tag = next_tag++;
    long W = 0;
    for (int v : nodes) { stamp[v] = tag; W += g.w[v]; }
    std::vector<int32_t> order;
    // pseudo-peripheral start: two BFS sweeps, each restarting from a minimum-degree node of the last level
    int root = nodes[0];
    for (int sweep = 0; sweep < 2; ++sweep) {
      bfs(g, stamp, tag, root, order, level, seen, next_seen++);
      const int last = level[order.back()];
      int best = order.back(), bd = deg_in(best, tag);
      for (size_t q = order.size(); q-- > 0 && level[order[q]] == last;) {
        const int dd = deg_in(order[q], tag);
        if (dd < bd || (dd == bd && order[q] < best)) { best = order[q]; bd = dd; }
      }
      root = best;
    }
    bfs(g, stamp, tag, root, order, level, seen, next_seen++);
    const int n_lv = level[order.back()] + 1;
    int sep = -1;
    if (W > prm.leaf_cols && n_lv >= 3) {
      std::vector<long> lw(n_lv, 0);
      for (int v : order) lw[level[v]] += g.w[v];
      // Among the level sets that qualify, the lightest - and among those within `sep_weight_slack` of the lightest, the one that cuts
      // the subgraph most evenly (round 5). Taking the FIRST lightest level, as rounds 2 - 4 did, cut a band - all of whose level sets
      // weigh the same - 20 : 80 at every step: an elimination tree of depth log_1.25 instead of log_2 (16 levels of dependent
      // launches for the 1 000-camera ring where 11 do).
      long before = 0, min_w = -1;
      for (int s = 0; s < n_lv; ++s) {
        const long after = W - before - lw[s];
        if (s > 0 && s < n_lv - 1 && before >= prm.min_side_frac * W && after >= prm.min_side_frac * W &&
            lw[s] <= prm.max_sep_frac * W && (min_w < 0 || lw[s] < min_w)) min_w = lw[s];
        before += lw[s];
      }
      before = 0;
      long best_gap = -1;
      for (int s = 0; s < n_lv; ++s) {
        const long after = W - before - lw[s];
        if (min_w >= 0 && s > 0 && s < n_lv - 1 && before >= prm.min_side_frac * W && after >= prm.min_side_frac * W &&
            lw[s] <= prm.max_sep_frac * W && lw[s] <= prm.sep_weight_slack * min_w) {
          const long gap = before > after ? before - after : after - before;
          if (best_gap < 0 || gap < best_gap) { best_gap = gap; sep = s; }
        }
        before += lw[s];
      }
    }
    const int me = (int)parts.size();
    parts.emplace_back();
    parts[me].parent = parent;
    if (sep < 0) {   // leaf: Cuthill-McKee-like order (BFS from the pseudo-peripheral node) keeps the band narrow
      parts[me].nodes = order;
      return;
    }
    std::vector<int32_t> rest;
    for (int v : order) {
      if (level[v] == sep) parts[me].nodes.push_back(v); else rest.push_back(v);
    }
    // connected components of what is left; each becomes a child subtree
    const int32_t rtag = next_tag++;
    for (int v : rest) stamp[v] = rtag;
    for (int v : parts[me].nodes) stamp[v] = -1;
    std::vector<std::vector<int32_t>> comps;
    {
      std::vector<int32_t> comp;
      const int32_t stag = next_seen++;
      for (int v : rest) {
        if (seen[v] == stag) continue;
        bfs(g, stamp, rtag, v, comp, level, seen, stag);
        comps.push_back(comp);
      }
    }
    for (auto& cmp : comps) dissect(std::move(cmp), me);
  }
};

}  // namespace detail

// blocks: (row block, col block) pairs of the non-zero camera blocks of S (any triangle, duplicates allowed).
// width(cb), first(cb): scalar width / first scalar column of camera block cb in the ORIGINAL numbering.
template <class WidthFn, class FirstFn>
inline bool build_plan(int n_cb, int N, const std::vector<std::pair<uint32_t, uint32_t>>& blocks, WidthFn width, FirstFn first,
                       const PlanParams& prm, uint64_t max_tasks, Plan& out) {
  using namespace detail;
  out = Plan();
  out.N = N;
  // ---- camera-block graph ----
  Graph g;
  g.n = n_cb;
  g.w.resize(n_cb);
  for (int v = 0; v < n_cb; ++v) g.w[v] = width(v);
  {
    std::vector<std::pair<uint32_t, uint32_t>> e;
    e.reserve(blocks.size() * 2);
    for (const auto& b : blocks)
      if (b.first != b.second) { e.emplace_back(b.first, b.second); e.emplace_back(b.second, b.first); }
    std::sort(e.begin(), e.end());
    e.erase(std::unique(e.begin(), e.end()), e.end());
    g.start.assign(n_cb + 1, 0);
    for (const auto& x : e) g.start[x.first + 1]++;
    for (int v = 0; v < n_cb; ++v) g.start[v + 1] += g.start[v];
    g.adj.resize(e.size());
    for (size_t q = 0; q < e.size(); ++q) g.adj[q] = (int32_t)e[q].second;   // sorted by (row, col): CSR order
  }
  // ---- dense border ----
  std::vector<uint8_t> is_border(n_cb, 0);
  {
    std::vector<int> deg(n_cb);
    for (int v = 0; v < n_cb; ++v) deg[v] = g.start[v + 1] - g.start[v];
    std::vector<int> sorted(deg);
    std::sort(sorted.begin(), sorted.end());
    const int med = n_cb ? sorted[n_cb / 2] : 0;
    const double thr = std::max<double>(prm.dense_degree_min, prm.dense_degree_factor * med);
    for (int v = 0; v < n_cb; ++v) is_border[v] = deg[v] > thr;
  }
  // graph without the border
  Graph h;
  h.n = n_cb;
  h.w = g.w;
  h.start.assign(n_cb + 1, 0);
  for (int v = 0; v < n_cb; ++v) {
    if (is_border[v]) continue;
    for (int32_t e = g.start[v]; e < g.start[v + 1]; ++e) h.start[v + 1] += !is_border[g.adj[e]];
  }
  for (int v = 0; v < n_cb; ++v) h.start[v + 1] += h.start[v];
  h.adj.resize(h.start[n_cb]);
  for (int v = 0; v < n_cb; ++v) {
    if (is_border[v]) continue;
    int32_t pos = h.start[v];
    for (int32_t e = g.start[v]; e < g.start[v + 1]; ++e)
      if (!is_border[g.adj[e]]) h.adj[pos++] = g.adj[e];
  }
  // ---- nested dissection of every connected component; part 0 = border (root) ----
  std::vector<Part> parts(1);
  for (int v = 0; v < n_cb; ++v)
    if (is_border[v]) { parts[0].nodes.push_back(v); out.n_border_blocks++; }
  {
    std::vector<std::vector<int32_t>> comps;
    {
      std::vector<int32_t> stamp(n_cb, 0), seen(n_cb, 0), level(n_cb, 0), comp;
      for (int v = 0; v < n_cb; ++v) stamp[v] = is_border[v] ? -1 : 0;
      for (int v = 0; v < n_cb; ++v) {
        if (is_border[v] || seen[v] == 1) continue;
        bfs(h, stamp, 0, v, comp, level, seen, 1);
        comps.push_back(comp);
      }
    }
    Dissector ds(h, prm, parts);   // stamps every node set it works on with a fresh tag
    for (auto& cmp : comps) ds.dissect(std::move(cmp), 0);
  }
  out.n_parts = (int)parts.size();
  // ---- post-order of the part tree (children in creation order), column assignment with padding ----
  std::vector<std::vector<int>> children(parts.size());
  for (size_t p = 1; p < parts.size(); ++p) children[parts[p].parent].push_back((int)p);
  std::vector<int> post;
  {
    std::vector<std::pair<int, size_t>> st;
    st.emplace_back(0, 0);
    while (!st.empty()) {
      auto& top = st.back();
      if (top.second < children[top.first].size()) { const int ch = children[top.first][top.second++]; st.emplace_back(ch, 0); }
      else { post.push_back(top.first); st.pop_back(); }
    }
  }
  out.pcol.assign(N, -1);
  std::vector<int32_t> tile_kb;
  int col = 0;
  for (int p : post) {
    if (parts[p].nodes.empty()) continue;
    const int col0 = col;
    for (int v : parts[p].nodes) {
      const int f = first(v), wv = g.w[v];
      for (int q = 0; q < wv; ++q) out.pcol[f + q] = col + q;
      col += wv;
    }
    const int used = col - col0;
    const int nt = (used + kTile - 1) / kTile;
    for (int t = 0; t < nt; ++t) tile_kb.push_back(std::min(kTile, used - t * kTile));
    col = col0 + nt * kTile;
  }
  for (int q = 0; q < N; ++q)
    if (out.pcol[q] < 0) return false;   // a block outside every part: cannot happen
  out.N_pad = col;
  const int nT = col / kTile;
  out.nT = nT;
  out.tile_kb = tile_kb;
  // ---- tile pattern of S and symbolic factorisation on tile columns ----
  std::vector<std::vector<int32_t>> below(nT);   // rows below the diagonal, per tile column
  for (const auto& b : blocks) {
    const int ra = out.pcol[first(b.first)], rb = ra + g.w[b.first] - 1;
    const int ca = out.pcol[first(b.second)], cb = ca + g.w[b.second] - 1;
    for (int tr = ra / kTile; tr <= rb / kTile; ++tr)
      for (int tc = ca / kTile; tc <= cb / kTile; ++tc) {
        const int hi = std::max(tr, tc), lo = std::min(tr, tc);
        if (hi != lo) below[lo].push_back(hi);
      }
  }
  std::vector<int32_t> parent(nT, -1);
  for (int k = 0; k < nT; ++k) {
    auto& s = below[k];
    std::sort(s.begin(), s.end());
    s.erase(std::unique(s.begin(), s.end()), s.end());
    if (s.empty()) continue;
    const int par = s[0];
    parent[k] = par;
    auto& ps = below[par];
    ps.insert(ps.end(), s.begin() + 1, s.end());   // merged (sorted, de-duplicated) when column `par` is reached
  }
  out.level_of.assign(nT, 0);
  for (int k = 0; k < nT; ++k)
    if (parent[k] >= 0) out.level_of[parent[k]] = std::max(out.level_of[parent[k]], out.level_of[k] + 1);
  out.n_levels = nT ? *std::max_element(out.level_of.begin(), out.level_of.end()) + 1 : 0;
  // ---- slots ----
  out.tmap.assign((size_t)(nT + 1) * nT, -1);
  int slots = 0;
  uint64_t n_u_targets_est = 0;
  for (int k = 0; k < nT; ++k) {
    out.tmap[(size_t)k * nT + k] = slots++;
    for (int i : below[k]) out.tmap[(size_t)i * nT + k] = slots++;
    out.tmap[(size_t)nT * nT + k] = slots++;   // rhs row
    const uint64_t m = below[k].size();
    n_u_targets_est += (m + 1) * (m + 2) / 2;
    out.n_fill_tiles += m + 1;
  }
  out.n_slots = slots;
  out.slot_col.assign(slots, 0);
  for (int k = 0; k < nT; ++k) {
    out.slot_col[out.tmap[(size_t)k * nT + k]] = k;
    for (int i : below[k]) out.slot_col[out.tmap[(size_t)i * nT + k]] = k;
    out.slot_col[out.tmap[(size_t)nT * nT + k]] = k;
  }
  if (n_u_targets_est * 16 > max_tasks) return false;   // too much fill for the task-list formulation: dense path
  // ---- level schedule ----
  out.f_start.assign(out.n_levels + 1, 0);
  for (int k = 0; k < nT; ++k) out.f_start[out.level_of[k] + 1]++;
  for (int l = 0; l < out.n_levels; ++l) out.f_start[l + 1] += out.f_start[l];
  out.f_cols.resize(nT);
  {
    std::vector<int32_t> fill(out.f_start.begin(), out.f_start.end() - 1);
    for (int k = 0; k < nT; ++k) out.f_cols[fill[out.level_of[k]]++] = k;
  }
  out.t_start.assign(out.n_levels + 1, 0);
  out.u_start.assign(out.n_levels + 1, 0);
  struct Contrib { int32_t dst; int32_t k; int32_t a, b; uint8_t rhs, diag; int32_t j; };
  std::vector<Contrib> contribs;
  out.lookahead.assign(out.n_levels, 0);
  out.u_defer_start.assign(out.n_levels, 0);
  std::vector<std::vector<std::pair<int32_t, int32_t>>> pre(nT);   // per tile column: (slot of A_kj, j)
  for (int l = 0; l < out.n_levels; ++l) {
    contribs.clear();
    int level_scratch = 0;
    for (int q = out.f_start[l]; q < out.f_start[l + 1]; ++q) {
      const int k = out.f_cols[q];
      const auto& s = below[k];
      // T: every tile below the diagonal of column k, and the rhs strip
      for (size_t a = 0; a <= s.size(); ++a) {
        const bool rhs = a == s.size();
        const int i = rhs ? nT : s[a];
        const int slot = out.tmap[(size_t)i * nT + k];
        for (int bi = 0; bi < (rhs ? 1 : 4); ++bi)
          for (int bj = 0; bj < 4; ++bj) {
            out.t_tasks.push_back(GemmTask{slot, (int16_t)bi, (int16_t)bj, (int32_t)out.t_pairs.size(), (int32_t)out.t_pairs.size() + 1});
            out.t_pairs.push_back(SlotPair{slot, k});
          }
        out.flops += 2.0 * (rhs ? 1 : 64) * 64 * 64 / 2;
      }
      // U: contributions of column k to the tiles (i, j), i >= j, both rows of column k
      for (size_t b = 0; b < s.size(); ++b) {
        const int j = s[b];
        const int sb = out.tmap[(size_t)j * nT + k];
        for (size_t a = b; a <= s.size(); ++a) {
          const bool rhs = a == s.size();
          const int i = rhs ? nT : s[a];
          const int dst = out.tmap[(size_t)i * nT + j];
          if (dst < 0) return false;   // symbolic factorisation is closed under these updates: cannot happen
          contribs.push_back(Contrib{dst, k, out.tmap[(size_t)i * nT + k], sb, (uint8_t)rhs, (uint8_t)(i == j), (int32_t)j});
          out.flops += 2.0 * (rhs ? 1 : 64) * 64 * 64;
        }
      }
      out.flops += 64.0 * 64 * 64 / 3 + 64.0 * 64 * 64 / 3;   // factor + inverse of the diagonal tile
    }
    std::sort(contribs.begin(), contribs.end(), [](const Contrib& x, const Contrib& y) { return x.dst != y.dst ? x.dst < y.dst : x.k < y.k; });
    // look-ahead of the next level: every diagonal tile of level l + 1 must take its level-l contributions inside its factor workgroup
    bool la = l + 1 < out.n_levels;
    for (size_t a = 0; a < contribs.size() && la;) {
      size_t b = a;
      while (b < contribs.size() && contribs[b].dst == contribs[a].dst) ++b;
      if (contribs[a].diag && out.level_of[contribs[a].j] == l + 1 && (int)(b - a) > std::min(prm.max_pre, kMaxPre)) la = false;
      a = b;
    }
    if (l + 1 < out.n_levels) out.lookahead[l + 1] = la;
    for (int pass = 0; pass < 2; ++pass) {   // 0: the tasks every schedule runs; 1: the ones the look-ahead schedule leaves to the factor workgroups
      if (pass == 1) out.u_defer_start[l] = (int32_t)out.u_tasks.size();
      for (size_t a = 0; a < contribs.size();) {
        size_t b = a;
        while (b < contribs.size() && contribs[b].dst == contribs[a].dst) ++b;
        const bool deferred = la && contribs[a].diag && out.level_of[contribs[a].j] == l + 1;
        if (deferred != (pass == 1)) { a = b; continue; }
        if (deferred)
          for (size_t q = a; q < b; ++q) pre[contribs[a].j].emplace_back(contribs[q].a, contribs[q].k);
        const int c0 = (int)out.u_pairs.size();
        for (size_t q = a; q < b; ++q) out.u_pairs.push_back(SlotPair{contribs[q].a, contribs[q].b});
        const int c1 = (int)out.u_pairs.size();
        const int n_c = c1 - c0;
        const int chunks = n_c >= kSplitMin ? (n_c + kSplitChunk - 1) / kSplitChunk : 1;
        if (chunks > 32767) return false;   // (GemmTask::n_chunks is 16 bits: a level of more than 131 000 columns - dense path)
        for (int bi = 0; bi < (contribs[a].rhs ? 1 : 4); ++bi)
          for (int bj = 0; bj < 4; ++bj) {
            if (contribs[a].diag && bj > bi) continue;   // lower triangle of a diagonal tile
            if (chunks == 1) {
              out.u_tasks.push_back(GemmTask{contribs[a].dst, (int16_t)bi, (int16_t)bj, c0, c1});
              continue;
            }
            for (int ch = 0; ch < chunks; ++ch) {
              GemmTask g{contribs[a].dst, (int16_t)bi, (int16_t)bj, c0 + ch * kSplitChunk, std::min(c1, c0 + (ch + 1) * kSplitChunk)};
              g.chunk = (int16_t)ch; g.n_chunks = (int16_t)chunks; g.group = out.n_split_groups; g.scratch = level_scratch;
              out.u_tasks.push_back(g);
            }
            out.n_split_groups += 1;
            level_scratch += chunks;   // (the scratch blocks of a level are free again when its update kernel has ended: levels share them)
            out.n_scratch_blocks = std::max(out.n_scratch_blocks, level_scratch);
          }
        a = b;
      }
    }
    out.t_start[l + 1] = (int32_t)out.t_tasks.size();
    out.u_start[l + 1] = (int32_t)out.u_tasks.size();
  }
  out.pre_start.assign(nT + 1, 0);
  for (int k = 0; k < nT; ++k) {
    for (const auto& e : pre[k]) { out.pre_slot.push_back(e.first); out.pre_col.push_back(e.second); }
    out.pre_start[k + 1] = (int32_t)out.pre_slot.size();
  }
  // ---- reverse sweep ----
  out.bs_start.assign(nT + 1, 0);
  for (int k = 0; k < nT; ++k) {
    // (round 6: highest level first. The one-launch reverse sweep - sp_backsolve_all_kernel - walks a column's list in rounds of four and
    // waits for each round's parts of z: the part solved LAST - the lowest level - now comes last, so only the final round waits, with
    // every other product done. The same order in every form of the sweep: the same sums.)
    std::vector<int> ord(below[k].begin(), below[k].end());
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return out.level_of[a] > out.level_of[b]; });
    for (int i : ord) { out.bs_slot.push_back(out.tmap[(size_t)i * nT + k]); out.bs_row.push_back(i); }
    out.bs_start[k + 1] = (int32_t)out.bs_slot.size();
  }
  out.bs_rec.assign((size_t)nT * 16, 0);
  for (int f = 0; f < nT; ++f) {
    const int k = out.f_cols[f];
    int32_t* r = &out.bs_rec[(size_t)f * 16];
    r[0] = k; r[1] = out.tmap[(size_t)nT * nT + k]; r[2] = out.bs_start[k + 1] - out.bs_start[k]; r[3] = out.bs_start[k];
    for (int i = 0; i < kBsInline && i < r[2]; ++i) { r[4 + 2 * i] = out.bs_slot[r[3] + i]; r[5 + 2 * i] = out.bs_row[r[3] + i]; }
  }
  return true;
}

}  // namespace mvgx_sparse

#endif  // MVGX_BA_SPARSE_PLAN_H_
