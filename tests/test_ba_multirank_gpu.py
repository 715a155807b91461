"""GPU tests of the BA exchange step on ONE device: (1) the RCCL binding with a 1-rank communicator,
(2) a 2-rank run — two contexts holding the two point shards, driven from two host threads, with a test transport that
implements all-reduce(sum / max) semantics through host memory. The sharded solve must reproduce the single-context solve
(same LM trajectory), which checks every cross-rank reduction the solver issues, with the real kernels."""
import ctypes as C
import threading

import numpy as np
import pytest

from openmvg_amd import ba, sharding, synth

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_communicator():
    sc = synth.ba_scene(10, 300, track_len=6, model=3, n_intr_groups=2, seed=91)
    ctx = ba.BaContext(sc); ref = ctx.solve(); ctx.close()
    ctx = ba.BaContext(sc)
    ctx.comm_init(1, 0, ba.comm_unique_id())
    s = ctx.solve()
    ctx.close()
    assert s.num_iterations == ref.num_iterations and abs(s.final_rmse - ref.final_rmse) < 1e-12


class _HostAllReduce:
    """all-reduce over `world` threads of one process: D2H, barrier, combine, H2D (test transport only)."""

    def __init__(self, world):
        self.hip = C.CDLL("libamdhip64.so")
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.calls = 0

    def make(self, rank):
        def fn(ptr, count, op, stream):
            self.hip.hipStreamSynchronize(C.c_void_p(stream))
            host = np.empty(count, np.float64)
            assert self.hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(ptr), C.c_size_t(8 * count), 2) == 0
            self.slots[rank] = host
            self.barrier.wait()
            tot = np.maximum.reduce(self.slots) if op == 1 else np.sum(self.slots, axis=0)
            self.barrier.wait()
            assert self.hip.hipMemcpy(C.c_void_p(ptr), C.c_void_p(tot.ctypes.data), C.c_size_t(8 * count), 1) == 0
            if rank == 0:
                self.calls += 1
            return 0
        return fn


@pytest.mark.parametrize("kw,strict", [
    (dict(n_cams=14, n_points=600, track_len=6, model=3, n_intr_groups=2, seed=92), True),
    # 40-iteration crawl along a Huber-flattened valley: the per-shard summation order differs from the single-rank one
    # (fp64 sums are not associative — the reference's own order is unspecified too), so the
    # function-tolerance test may fire one iteration apart; the result must still agree to the parity tolerance.
    (dict(n_cams=10, n_points=400, track_len=5, model=1, n_intr_groups=1, seed=93, outlier_frac=0.05), False),
])
def test_two_point_shards_reproduce_the_single_rank_solve(kw, strict):
    sc = synth.ba_scene(**kw)
    ctx = ba.BaContext(sc); ref = ctx.solve(); rposes, rintr, rpts = ctx.read_params(); ctx.close()

    world = 2
    tr = _HostAllReduce(world)
    owner = sharding.assign_points(sc["obs_point"], sc["n_points"], world)
    out = [None] * world

    def run(rank):
        shard, mine = sharding.shard_ba_scene(sc, rank, world, owner)
        c = ba.BaContext(shard)
        c.set_allreduce(tr.make(rank))
        s = c.solve()
        poses, intr, pts = c.read_params()
        c.close()
        out[rank] = (s, poses, intr, pts, mine)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert all(o is not None for o in out) and tr.calls > 0
    pts = np.zeros_like(rpts)
    for s, poses, intr, p, mine in out:
        assert abs(s.initial_rmse - ref.initial_rmse) < 1e-9 and abs(s.initial_cost - ref.initial_cost) <= 1e-12 * ref.initial_cost
        if strict:
            assert s.num_iterations == ref.num_iterations and s.num_successful_steps == ref.num_successful_steps
            assert abs(s.final_rmse - ref.final_rmse) < 1e-9
            assert abs(s.final_cost - ref.final_cost) <= 1e-9 * ref.final_cost
            assert np.allclose(poses, rposes, atol=1e-9) and np.allclose(intr, rintr, rtol=1e-9, atol=1e-9)
        else:
            # Along the plateau every LM step changes the cost by about function_tolerance (1e-6 RELATIVE), and the
            # termination test fires a few iterations apart for different summation orders (measured under emulation:
            # oracle 47 iterations, this solver 40, its earlier kernel generation 42 - all "converged"). What is
            # comparable is the cost level: within a few function tolerances.
            assert abs(s.final_cost - ref.final_cost) <= 2e-5 * ref.final_cost
            assert abs(s.final_rmse - ref.final_rmse) < 2e-5 * max(1.0, ref.final_rmse)
        pts[mine] = p
    if strict:
        assert np.allclose(pts, rpts, atol=1e-8)
    # camera parameters are bit-identical across ranks (every rank factors the same reduced system)
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


# ---- the in-process form: one context over several device shards, threads + peer-mapped sums inside the library ----
@pytest.mark.parametrize("devices,kw", [
    ([0, 0], dict(n_cams=14, n_points=600, track_len=6, model=3, n_intr_groups=2, seed=92)),
    ([0, 0, 0, 0], dict(n_cams=60, n_points=6000, track_len=8, model=3, n_intr_groups=4, seed=96)),
])
def test_in_process_multi_device_context_equals_the_single_device_solve(devices, kw):
    """mvgx_ba_create_multi with several contexts on this one GPU (peer transport: every rank sums all ranks' buffers in rank
    order): same LM trajectory as the single context, parameters / residuals / track angles in the caller's numbering."""
    sc = synth.ba_scene(**kw)
    one = ba.BaContext(sc); s1 = one.solve(); p1, i1, x1 = one.read_params(); r1, a1 = one.residuals(), one.track_angles(); one.close()
    many = ba.BaContext(sc, devices=devices)
    s2 = many.solve()
    p2, i2, x2 = many.read_params()
    r2, a2, e2 = many.residuals(), many.track_angles(), many.evaluate()
    many.close()
    assert s2.num_iterations == s1.num_iterations and s2.num_successful_steps == s1.num_successful_steps
    assert abs(s2.final_rmse - s1.final_rmse) < 1e-9 and abs(s2.final_cost - s1.final_cost) <= 1e-9 * s1.final_cost
    assert abs(e2[1] - s1.final_rmse) < 1e-9
    assert np.allclose(p2, p1, atol=1e-9) and np.allclose(i2, i1, rtol=1e-9, atol=1e-9) and np.allclose(x2, x1, atol=1e-8)
    assert np.allclose(r2, r1, atol=1e-8) and np.allclose(a2, a1, atol=1e-7)


def test_in_process_multi_device_large_exchange_takes_the_sliced_path(monkeypatch):
    """dense reduced system of 1.2 M doubles (> 2^20): the reduce-scatter + all-gather form of the peer transport"""
    monkeypatch.setenv("MVGX_BA_SOLVER", "dense")
    sc = synth.ba_scene(n_cams=180, n_points=4000, track_len=180, model=1, n_intr_groups=1, seed=99)   # every point sees every camera
    opt = ba.default_options(max_num_iterations=2)
    one = ba.BaContext(sc); s1 = one.solve(opt); p1, _, _ = one.read_params(); one.close()
    many = ba.BaContext(sc, devices=[0, 0, 0]); s2 = many.solve(opt); p2, _, _ = many.read_params(); many.close()
    assert s2.num_iterations == s1.num_iterations and abs(s2.final_cost - s1.final_cost) <= 1e-9 * s1.final_cost
    assert np.allclose(p2, p1, atol=1e-9)
