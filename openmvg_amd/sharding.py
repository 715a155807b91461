"""Work partitioning for one-process-per-GPU runs. Pure host logic, no device code.

Matching: image pairs are independent units; descriptors are replicated on every GPU and the (I, J)-ordered pair list is
cut into `world` contiguous ranges of equal descriptor-pair work (sum nI*nJ) — contiguous in I keeps a rank's database
images L2-resident, as Matcher_Regions.cpp:49-54 groups by I. No collective on the data path.

Bundle adjustment: 3-D points (with ALL their observations — a point's rows must be local to eliminate it,
ceres schur_eliminator_impl.h:114-151) are partitioned across ranks balanced by sum L_p^2 (the Schur outer-product work
of a track of length L_p); poses and intrinsics are replicated in the same order on every rank.
"""
import numpy as np


def shard_pairs(pairs, n_desc, rank, world):
    """Rows [lo, hi) of `pairs` (n x 2, I = database, J = query) for `rank`, balanced by cumulative nI * nJ."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    if world <= 1 or len(pairs) == 0:
        return pairs
    n_desc = np.asarray(n_desc, np.float64)
    w = n_desc[pairs[:, 0]] * n_desc[pairs[:, 1]] + 1.0   # +1: empty pairs still cost a slot
    cum = np.cumsum(w)
    cuts = np.searchsorted(cum, cum[-1] * np.arange(1, world) / world, side="left") + 1
    bounds = np.concatenate([[0], np.minimum(cuts, len(pairs)), [len(pairs)]])
    bounds = np.maximum.accumulate(bounds)
    return pairs[int(bounds[rank]):int(bounds[rank + 1])]


def assign_points(obs_point, n_points, world):
    """owner[p] in [0, world): greedy longest-processing-time assignment of points by L_p^2 (deterministic)."""
    L = np.bincount(np.asarray(obs_point, np.int64), minlength=int(n_points)).astype(np.float64)
    if world <= 1:
        return np.zeros(int(n_points), np.int32)
    cost = L * L + 1.0
    # LPT over (few) distinct costs: deal the points of each cost class round-robin, heaviest class first, starting
    # each class at the currently lightest rank. O(n log n), independent of world.
    order = np.argsort(-cost, kind="stable")
    owner = np.empty(int(n_points), np.int32)
    load = np.zeros(world)
    start = 0
    sc = cost[order]
    while start < len(order):
        end = start + int(np.searchsorted(-sc[start:], -sc[start], side="right"))
        ranks = np.argsort(load, kind="stable")
        k = np.arange(end - start)
        owner[order[start:end]] = ranks[k % world]
        load += np.bincount(ranks[k % world], minlength=world) * sc[start]
        start = end
    return owner


def shard_ba_scene(scene, rank, world, owner=None):
    """This rank's shard of a flat BA scene (the dict layout of openmvg_amd.ba / include/mvgx.h): all poses and
    intrinsics, the rank's points renumbered 0..n-1, and the observations of those points. Also returns the global ids
    of the local points (to scatter refined points back)."""
    if owner is None:
        owner = assign_points(scene["obs_point"], scene["n_points"], world)
    mine = np.flatnonzero(owner == rank)
    local_id = np.full(int(scene["n_points"]), -1, np.int64)
    local_id[mine] = np.arange(len(mine))
    op = np.asarray(scene["obs_point"], np.int64)
    keep = local_id[op] >= 0
    out = dict(scene)
    out["points"] = np.ascontiguousarray(np.asarray(scene["points"], np.float64)[mine])
    out["n_points"] = len(mine)
    out["obs_point"] = local_id[op[keep]].astype(np.uint32)
    for k in ("obs_pose", "obs_intr"):
        out[k] = np.ascontiguousarray(np.asarray(scene[k])[keep])
    out["obs_xy"] = np.ascontiguousarray(np.asarray(scene["obs_xy"], np.float64).reshape(-1, 2)[keep])
    out["n_obs"] = int(keep.sum())
    # control points shard like any other point (their residual rows only touch camera blocks, which are summed)
    for k in ("obs_weight", "obs_is_control"):
        if scene.get(k) is not None:
            out[k] = np.ascontiguousarray(np.asarray(scene[k])[keep])
    if scene.get("point_const_mask") is not None:
        out["point_const_mask"] = np.ascontiguousarray(np.asarray(scene["point_const_mask"])[mine])
    # pose-centre priors are residuals on replicated blocks: exactly one rank may hold them, or the summed reduced system
    # and cost would count them `world` times
    if rank != 0:
        for k in ("prior_pose", "prior_center", "prior_weight"):
            out.pop(k, None)
    return out, mine
