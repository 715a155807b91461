"""world_size-2 CPU (gloo) tests of the multi-GPU decomposition: the host-side partitioning is exact,
disjoint and balanced, and the quantities the BA ranks sum with one all-reduce are additive over the shards. The per-shard
values come from the CPU oracle (the checker) — the product's kernels run on GPUs only and are covered by the -m gpu tests;
what is tested here is the partition + the collective pattern (torch.distributed, same calls as the RCCL run)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openmvg_amd import matching, sharding, synth
from tests import _oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for port in [_free_port()] for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    return dict(ret)


def _pairs_job(rank, world):
    n_desc = np.array([300, 0, 280, 310, 1, 295, 305, 290, 300], np.int64)
    pairs = matching.exhaustive_pairs_array(len(n_desc))
    mine = sharding.shard_pairs(pairs, n_desc, rank, world)
    work = float((n_desc[mine[:, 0]] * n_desc[mine[:, 1]]).sum())
    # gather every rank's range: together they must be the full list, in order, without overlap
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(mine)]))
    total = torch.tensor([work], dtype=torch.float64)
    dist.all_reduce(total)
    lo = int(sum(int(s) for s in sizes[:rank]))
    assert np.array_equal(mine, pairs[lo:lo + len(mine)])
    assert int(sum(int(s) for s in sizes)) == len(pairs)
    full = float((n_desc[pairs[:, 0]] * n_desc[pairs[:, 1]]).sum())
    assert abs(float(total) - full) < 1e-6
    return work / full


def test_pair_sharding_is_a_balanced_partition():
    shares = _run(_pairs_job)
    assert abs(sum(shares.values()) - 1.0) < 1e-12
    assert max(shares.values()) < 0.62   # contiguous cut of a 36-pair list: within one pair of 50 %


def _ba_job(rank, world):
    sc = synth.ba_scene(12, 400, track_len=6, model=3, n_intr_groups=2, seed=77, outlier_frac=0.05)
    shard, mine = sharding.shard_ba_scene(sc, rank, world)
    # every rank holds all cameras; points are disjoint and complete
    assert shard["n_poses"] == sc["n_poses"] and shard["n_intrinsics"] == sc["n_intrinsics"]
    counts = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([shard["n_points"], shard["n_obs"]]))
    assert sum(int(c[0]) for c in counts) == sc["n_points"] and sum(int(c[1]) for c in counts) == sc["n_obs"]
    ids = [None] * world
    dist.all_gather_object(ids, mine.tolist())
    assert sorted(i for part in ids for i in part) == list(range(sc["n_points"]))
    # additivity of what the ranks all-reduce: cost (1/2 sum rho) and squared error -> global cost / RMSE
    cost, rmse = _oracle.port_ba_evaluate(shard)
    t = torch.tensor([cost, rmse * rmse * 2.0 * shard["n_obs"], float(shard["n_obs"])], dtype=torch.float64)
    dist.all_reduce(t)
    gcost, grmse = _oracle.port_ba_evaluate(sc)
    assert abs(float(t[0]) - gcost) <= 1e-12 * gcost
    assert abs(np.sqrt(float(t[1]) / (2.0 * float(t[2]))) - grmse) <= 1e-12 * grmse
    L = np.bincount(shard["obs_point"], minlength=shard["n_points"]).astype(np.float64)
    return float((L * L).sum())


def test_ba_point_sharding_is_a_balanced_partition_and_sums_are_additive():
    loads = _run(_ba_job)
    tot = sum(loads.values())
    assert max(loads.values()) / tot < 0.52


def test_assign_points_single_rank_and_determinism():
    sc = synth.ba_scene(6, 100, track_len=4, model=1, seed=5)
    assert np.all(sharding.assign_points(sc["obs_point"], sc["n_points"], 1) == 0)
    a = sharding.assign_points(sc["obs_point"], sc["n_points"], 4)
    b = sharding.assign_points(sc["obs_point"], sc["n_points"], 4)
    assert np.array_equal(a, b) and set(a.tolist()) == {0, 1, 2, 3}
