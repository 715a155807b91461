"""CPU tests of the BA solver's DEVICE code under the HIP execution-model emulation (tests/native/hipemu, tests/_emu.py).

No GPU exists where the CPU suite runs; these tests compile openmvg_amd/csrc/mvgx_ba.hip for the host against a shim
that emulates workgroups, waves, LDS, barriers, shuffles and v_mfma_f64_16x16x4_f64, and run the solver through the
same C ABI against the oracle. They check the index arithmetic of every kernel (sorted Schur products, Gram blocks,
blocked Cholesky with partial last block, back substitution, reductions) and the LM driver — what they cannot check is
gfx950 code generation and timing, which the `-m gpu` tests and bench.py cover on the MI355X. Scenes are tiny: the
emulation runs each workgroup as 64..1024 fibers."""
import ctypes as C
import threading

import numpy as np
import pytest

from openmvg_amd import ba, sharding, synth
from openmvg_amd import ba_options as bo
from tests import _emu, _oracle

RMSE_TOL = 1e-6


def _solve_emu(sc, options=None, **masks):
    with _emu.emulated():
        ctx = ba.BaContext(sc, **masks)
        s = ctx.solve(options)
        poses, intr, pts = ctx.read_params()
        ctx.close()
    return s, poses, intr, pts


def test_lm_trajectory_equals_oracle():
    """same iteration count, costs and parameters as the oracle"""
    sc = synth.ba_scene(n_cams=9, n_points=120, track_len=5, model=3, n_intr_groups=2, seed=21, rot_deg=0.3)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc)
    s, poses, intr, pts = _solve_emu(sc)
    assert rc == 0 and s.num_iterations == osum.num_iterations and s.num_successful_steps == osum.num_successful_steps
    assert s.termination == osum.termination
    assert abs(s.initial_cost - osum.initial_cost) <= 1e-10 * osum.initial_cost
    assert abs(s.final_cost - osum.final_cost) <= 1e-8 * osum.final_cost
    assert abs(s.final_rmse - osum.final_rmse) < RMSE_TOL
    assert np.allclose(pts, opx, atol=1e-8) and np.allclose(intr, opi, rtol=1e-8, atol=1e-8)


def test_multi_block_cholesky_and_wide_intrinsics():
    """reduced system of 3 block columns (partial last block), one intrinsic per camera (pose x intrinsic and
    intrinsic x intrinsic products both populated); two LM iterations against the oracle"""
    sc = synth.ba_scene(n_cams=12, n_points=150, track_len=4, model=2, n_intr_groups=12, seed=33)
    opt = dict(max_num_iterations=2)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc, options=_oracle.default_ba_options(**opt))
    s, poses, intr, pts = _solve_emu(sc, ba.default_options(**opt))
    assert 6 * 12 + 8 * 12 > 128
    assert s.num_iterations == osum.num_iterations == 2
    assert abs(s.final_cost - osum.final_cost) <= 1e-9 * osum.final_cost
    assert np.allclose(pts, opx, atol=1e-9) and np.allclose(poses, opp, atol=1e-9) and np.allclose(intr, opi, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("iopt,eopt,sopt", [(10, 2, 1), (14, 6, 0), (1, 1, 1)])
def test_subset_parameterizations(iopt, eopt, sopt):
    sc = synth.ba_scene(n_cams=8, n_points=100, track_len=5, model=3, n_intr_groups=2, seed=22, rot_deg=0.3)
    masks = bo.masks_for(sc, iopt, eopt, sopt)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc, **masks)
    s, poses, intr, pts = _solve_emu(sc, **masks)
    assert s.num_iterations == osum.num_iterations
    assert abs(s.final_rmse - osum.final_rmse) < RMSE_TOL * max(1.0, osum.final_rmse)
    if eopt == 1:
        assert np.array_equal(poses, sc["poses"])
    if sopt == 0:
        assert np.array_equal(pts, sc["points"])
    if iopt == 1:
        assert np.array_equal(intr, sc["intrinsics"])


class _HostAllReduce:
    """all-reduce over `world` threads of one process through host memory (the emulated device memory IS host memory)"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def make(self, rank):
        def fn(ptr, count, op, stream):
            buf = (C.c_double * count).from_address(ptr)
            self.slots[rank] = np.frombuffer(buf, np.float64).copy()
            self.barrier.wait()
            tot = np.maximum.reduce(self.slots) if op == 1 else np.sum(self.slots, axis=0)
            self.barrier.wait()
            np.frombuffer(buf, np.float64)[:] = tot
            return 0
        return fn


def test_two_point_shards_reproduce_the_single_rank_solve():
    """the exchange step of SURVEY 8(e) with the real device code: two emulated devices (two host threads), callback
    transport; every cross-rank reduction the solver issues is exercised"""
    sc = synth.ba_scene(n_cams=8, n_points=160, track_len=5, model=3, n_intr_groups=2, seed=92)
    opt = dict(max_num_iterations=3)
    ref, rposes, rintr, rpts = _solve_emu(sc, ba.default_options(**opt))
    world = 2
    tr = _HostAllReduce(world)
    owner = sharding.assign_points(sc["obs_point"], sc["n_points"], world)
    out = [None] * world

    def run(rank):
        shard, mine = sharding.shard_ba_scene(sc, rank, world, owner)
        c = ba.BaContext(shard)
        c.set_allreduce(tr.make(rank))
        s = c.solve(ba.default_options(**opt))
        poses, intr, pts = c.read_params()
        c.close()
        out[rank] = (s, poses, intr, pts, mine)

    with _emu.emulated():
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(600)
    assert all(o is not None for o in out)
    pts = np.zeros_like(rpts)
    for s, poses, intr, p, mine in out:
        assert s.num_iterations == ref.num_iterations and abs(s.final_cost - ref.final_cost) <= 1e-9 * ref.final_cost
        assert np.allclose(poses, rposes, atol=1e-9) and np.allclose(intr, rintr, rtol=1e-9, atol=1e-9)
        pts[mine] = p
    assert np.allclose(pts, rpts, atol=1e-8)
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


def _run_two_ranks(sc, owner, opt):
    world = 2
    tr = _HostAllReduce(world)
    out = [None] * world

    def run(rank):
        shard, mine = sharding.shard_ba_scene(sc, rank, world, owner)
        c = ba.BaContext(shard)
        c.set_allreduce(tr.make(rank))
        s = c.solve(ba.default_options(**opt))
        poses, intr, pts = c.read_params()
        c.close()
        out[rank] = (s, poses, intr, pts, mine)

    with _emu.emulated():
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(600)
    assert all(o is not None for o in out)
    return out


def test_a_camera_observed_on_one_rank_only_is_still_in_every_ranks_program():
    """round-1 advisor finding: every point seen by pose 0 lives on rank 0, so rank 1 has no residual on pose 0 (and none on
    intrinsic 0's ... pose 0 column block). Its 'free parameter with residuals' masks must come from all ranks, or rank 1
    overwrites the block's diagonal with 1 and the ranks factor different systems."""
    sc = synth.ba_scene(n_cams=8, n_points=160, track_len=5, model=3, n_intr_groups=2, seed=92)
    opt = dict(max_num_iterations=4)
    ref, rposes, rintr, rpts = _solve_emu(sc, ba.default_options(**opt))
    sees0 = np.zeros(sc["n_points"], bool)
    sees0[sc["obs_point"][sc["obs_pose"] == 0]] = True
    owner = np.where(sees0, 0, 1).astype(np.int32)
    rest = np.flatnonzero(~sees0)
    owner[rest[::3]] = 0   # rank 0 also gets a share of the other points
    assert not np.any(sc["obs_pose"][owner[sc["obs_point"]] == 1] == 0)
    out = _run_two_ranks(sc, owner, opt)
    for s, poses, intr, p, mine in out:
        assert s.num_iterations == ref.num_iterations and abs(s.final_cost - ref.final_cost) <= 1e-9 * ref.final_cost
        assert np.allclose(poses, rposes, atol=1e-9) and np.allclose(intr, rintr, rtol=1e-9, atol=1e-9)
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("model", [4, 5, 7])
def test_brown_fisheye_spherical_functors(model):
    sc = synth.ba_scene(n_cams=6, n_points=60, track_len=4, model=model, n_intr_groups=2, seed=70 + model, rot_deg=0.3)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc)
    s, poses, intr, pts = _solve_emu(sc)
    assert rc == 0 and s.num_iterations == osum.num_iterations and s.termination == osum.termination
    assert abs(s.final_cost - osum.final_cost) <= 1e-8 * osum.final_cost and abs(s.final_rmse - osum.final_rmse) < RMSE_TOL
    assert np.allclose(pts, opx, atol=1e-7) and np.allclose(intr, opi, rtol=1e-7, atol=1e-7) and np.allclose(poses, opp, atol=1e-7)


def test_control_points_and_pose_priors():
    """weighted loss-free residuals on constant points + PoseCenterConstraintCostFunction rows, together"""
    sc = synth.ba_scene(n_cams=8, n_points=80, track_len=5, model=3, n_intr_groups=2, seed=81, rot_deg=0.3)
    sc = synth.add_pose_priors(synth.add_control_points(sc, n_ctrl=5, weight=20.0), sigma=0.005, huber_a=2e-4, every=2)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc)
    s, poses, intr, pts = _solve_emu(sc)
    assert rc == 0 and s.num_iterations == osum.num_iterations and s.num_successful_steps == osum.num_successful_steps
    assert abs(s.initial_cost - osum.initial_cost) <= 1e-10 * osum.initial_cost
    assert abs(s.final_cost - osum.final_cost) <= 1e-8 * osum.final_cost
    assert abs(s.initial_rmse - osum.initial_rmse) < 1e-9 and abs(s.final_rmse - osum.final_rmse) < RMSE_TOL
    assert np.allclose(pts, opx, atol=1e-7) and np.allclose(poses, opp, atol=1e-7)
    ns = sc["n_structure_points"]
    assert np.array_equal(pts[ns:], sc["points"][ns:])


def test_two_shards_with_control_points_and_priors():
    """control points shard with the points; the pose priors live on rank 0 only (sharding.shard_ba_scene)"""
    sc = synth.ba_scene(n_cams=8, n_points=120, track_len=5, model=1, n_intr_groups=1, seed=95, rot_deg=0.3)
    sc = synth.add_pose_priors(synth.add_control_points(sc, n_ctrl=6, weight=20.0), sigma=0.005, huber_a=2e-4)
    opt = dict(max_num_iterations=3)
    ref, rposes, rintr, rpts = _solve_emu(sc, ba.default_options(**opt))
    world = 2
    tr = _HostAllReduce(world)
    owner = sharding.assign_points(sc["obs_point"], sc["n_points"], world)
    out = [None] * world

    def run(rank):
        shard, mine = sharding.shard_ba_scene(sc, rank, world, owner)
        c = ba.BaContext(shard)
        c.set_allreduce(tr.make(rank))
        s = c.solve(ba.default_options(**opt))
        poses, intr, pts = c.read_params()
        c.close()
        out[rank] = (s, poses, intr, pts, mine)

    with _emu.emulated():
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(600)
    assert all(o is not None for o in out)
    for s, poses, intr, p, mine in out:
        assert s.num_iterations == ref.num_iterations and abs(s.final_cost - ref.final_cost) <= 1e-9 * ref.final_cost
        assert abs(s.final_rmse - ref.final_rmse) < 1e-9 and abs(s.initial_rmse - ref.initial_rmse) < 1e-9
        assert np.allclose(poses, rposes, atol=1e-9)


def test_reduced_system_beyond_one_backsolve_group():
    """N = 6 * 42 + 8 * 3 = 276 > 256 columns: five Cholesky block steps (partial last one) and two back-substitution groups"""
    sc = synth.ba_scene(n_cams=42, n_points=260, track_len=4, model=3, n_intr_groups=3, seed=37)
    opt = dict(max_num_iterations=1)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc, options=_oracle.default_ba_options(**opt))
    s, poses, intr, pts = _solve_emu(sc, ba.default_options(**opt))
    assert s.num_iterations == osum.num_iterations == 1
    assert abs(s.final_cost - osum.final_cost) <= 1e-9 * osum.final_cost
    assert np.allclose(pts, opx, atol=1e-9) and np.allclose(poses, opp, atol=1e-9) and np.allclose(intr, opi, rtol=1e-9, atol=1e-9)


def test_pixel_residual_outlier_filter():
    """mvgx_ba_residuals + the mirror of RemoveOutliers_PixelResidualError (sfm_data_filters.cpp:40-73) against a numpy
    restatement that uses the projection of synth.project"""
    sc = synth.ba_scene(n_cams=7, n_points=90, track_len=4, model=3, n_intr_groups=2, seed=55, outlier_frac=0.1, rot_deg=0.0,
                        center_sigma=0.0, point_sigma=0.0)
    sc["intrinsics"] = sc["intrinsics_gt"].copy()
    perm = np.random.default_rng(1).permutation(sc["n_obs"])        # the caller's order is not the device's point order
    for k in ("obs_pose", "obs_intr", "obs_point"):
        sc[k] = sc[k][perm]
    sc["obs_xy"] = sc["obs_xy"][perm]
    xy = synth.project(3, sc["intrinsics"][sc["obs_intr"]], sc["poses"][sc["obs_pose"]], sc["points"][sc["obs_point"]])
    want = np.linalg.norm(xy - sc["obs_xy"], axis=1)
    with _emu.emulated():
        ctx = ba.BaContext(sc); got = ctx.residuals(); ctx.close()
        n_out, filtered = ba.RemoveOutliers_PixelResidualError(sc, 4.0, 3)
    assert np.allclose(got, want, rtol=1e-10, atol=1e-9)
    keep = want <= 4.0
    assert n_out == int((~keep).sum()) and n_out > 0
    cnt = np.bincount(sc["obs_point"][keep], minlength=sc["n_points"])
    keep &= cnt[sc["obs_point"]] >= 3
    assert filtered["n_obs"] == int(keep.sum()) and np.array_equal(filtered["obs_point"], sc["obs_point"][keep])



@pytest.mark.parametrize("tiles128", [1, 1000])
def test_two_level_cholesky(monkeypatch, tiles128):
    """256-column outer panels (used for reduced systems of >= 2048 columns) forced onto a 276-column system: inner steps
    update the panel's own columns, one deferred K = 256 update covers the rest (on 128 x 128 or on 64 x 64 tiles), the last
    panel is partial"""
    monkeypatch.setenv("MVGX_BA_TWO_LEVEL_MIN_N", "1")
    monkeypatch.setenv("MVGX_BA_UPDATE128_MIN_TILES", str(tiles128))
    sc = synth.ba_scene(n_cams=42, n_points=260, track_len=4, model=3, n_intr_groups=3, seed=37)
    opt = dict(max_num_iterations=1)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc, options=_oracle.default_ba_options(**opt))
    s, poses, intr, pts = _solve_emu(sc, ba.default_options(**opt))
    assert abs(s.final_cost - osum.final_cost) <= 1e-9 * osum.final_cost
    assert np.allclose(pts, opx, atol=1e-9) and np.allclose(poses, opp, atol=1e-9) and np.allclose(intr, opi, rtol=1e-9, atol=1e-9)


from tests._ba_cases import edge_scenes as _edge_scenes  # noqa: E402


@pytest.mark.parametrize("case", range(5))
def test_degenerate_problems_follow_the_oracle(case):
    name, sc, masks = _edge_scenes()[case]
    rc, osum, *_ = _oracle.port_ba_solve(sc, **masks)
    s, *_ = _solve_emu(sc, **masks)
    assert rc == 0, name
    assert (s.num_iterations, s.termination) == (osum.num_iterations, osum.termination), name
    assert abs(s.final_cost - osum.final_cost) <= 1e-8 * max(osum.final_cost, 1e-12) + 1e-18, name


@pytest.mark.parametrize("model", [3, 4, 5, 7])
def test_track_filters_follow_the_reference(model):
    """mvgx_ba_track_angles + the mirrors of RemoveOutliers_AngleError / badTrackRejector (device kernels under emulation)
    against the oracle and the reference's committed output (tests/golden/ba_filters.npz, make_filter_golden.py)"""
    import os
    from tests import _ba_cases
    sc = _ba_cases.filter_scene(model, n_cams=12, n_points=150) if model != 3 else _ba_cases.filter_scene(model)
    want = _oracle.port_ba_track_angles(sc)
    with _emu.emulated():
        ctx = ba.BaContext(sc); got = ctx.track_angles(); ctx.close()
        n_ang, f_ang = ba.RemoveOutliers_AngleError(sc, 2.0)
        again, f_both = ba.badTrackRejector(sc, 4.0, 50)
    assert np.abs(got - want).max() < 1e-9
    alive = np.bincount(sc["obs_point"], minlength=sc["n_points"]) > 0
    assert n_ang == int((alive & (want < 2.0)).sum()) > 0
    if model == 3:   # the scene of the committed reference output
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba_filters.npz"))
        assert n_ang == int(gold["m3_count_angle_only"]) and f_ang["n_obs"] == int(gold["m3_keep_angle_only"].sum())
        keep = gold["m3_keep"]
        assert again == (int(gold["m3_counts"].sum()) > 50)
        for k in ("obs_pose", "obs_point"):
            assert np.array_equal(f_both[k], sc[k][keep])
        assert np.array_equal(f_both["obs_xy"], sc["obs_xy"][keep])
    elif _oracle.have_ref_ba():
        keep, counts, _ = _oracle.ref_ba_filters(sc, 4.0, 2, 2.0)
        assert again == (sum(counts) > 50) and np.array_equal(f_both["obs_xy"], sc["obs_xy"][keep])


@pytest.mark.skipif(not _oracle.have_ref_ba(), reason="oracle/_ref/libref_ba.so not built")
def test_bundle_then_reject_loop_equals_the_reference_pipeline():
    """`do { BA } while (badTrackRejector(4.0, 0))` (sequential_SfM.cpp:206-210,1226-1232), emulated device code against the
    reference's Bundle_Adjustment_Ceres + sfm_data_filters.cpp driven by the same loop"""
    from tests import _ba_cases
    sc0 = synth.ba_scene(n_cams=8, n_points=90, track_len=4, model=3, n_intr_groups=2, seed=91, outlier_frac=0.05, n_rings=1)

    def ours_adjust(sc):
        sc = dict(sc)
        assert ba.Bundle_Adjustment_HIP().Adjust(sc)
        return sc

    def ref_adjust(sc):
        rc, st, poses, intr, pts = _oracle.ref_ba_adjust(sc)
        assert rc == 0
        out = dict(sc); out["poses"] = poses; out["intrinsics"] = intr; out["points"] = pts
        return out

    def ref_rejector(sc, prec, count):
        keep, counts, _ = _oracle.ref_ba_filters(sc, prec, 2, 2.0)
        return sum(counts) > count, ba._drop_observations(sc, keep)

    with _emu.emulated():
        ours, n_ours = _ba_cases.rejector_loop(ours_adjust, ba.badTrackRejector, sc0)
        ref, n_ref = _ba_cases.rejector_loop(ref_adjust, ref_rejector, sc0)
        assert n_ours == n_ref >= 2
        assert ours["n_obs"] == ref["n_obs"] < sc0["n_obs"]
        assert np.array_equal(ours["obs_xy"], ref["obs_xy"])
        c1 = ba.BaContext(ours); r1 = c1.evaluate()[1]; c1.close()
        c2 = ba.BaContext(ref); r2 = c2.evaluate()[1]; c2.close()
    assert abs(r1 - r2) < 1e-6


def test_structure_build_is_independent_of_the_host_thread_count(monkeypatch):
    """mvgx_ba_create builds the observation order, the slot lists and the three product lists on host threads (row-wise
    generation, per-thread histograms): the device sees the same arrays - bitwise equal costs after an LM iteration - for 1
    and 8 threads (MVGX_HOST_THREADS forces the parallel paths), for point-sorted and for shuffled observation lists"""
    sc = synth.ba_scene(n_cams=40, n_points=1500, track_len=6, model=3, n_intr_groups=3, seed=5)
    perm = np.random.default_rng(0).permutation(sc["n_obs"])
    shuffled = dict(sc)
    for k in ("obs_pose", "obs_intr", "obs_point"):
        shuffled[k] = sc[k][perm]
    shuffled["obs_xy"] = sc["obs_xy"][perm]
    assert sc["n_obs"] >= 4096   # counting_sort_indices' threshold for per-thread histograms
    for scene in (sc, shuffled):
        got = []
        for threads in ("1", "8"):
            monkeypatch.setenv("MVGX_HOST_THREADS", threads)
            with _emu.emulated():
                ctx = ba.BaContext(scene)
                s = ctx.lm_iteration()
                ctx.close()
            got.append((s.initial_cost, s.final_cost, s.final_rmse))
        assert got[0] == got[1]


def test_allocation_failures_during_create_are_reported_not_fatal(monkeypatch):
    """every early return of mvgx_ba_create / mvgx_match_create / the brute-force contexts releases the partially built
    context (guard objects): with the emulation's allocation-failure injection the calls raise MvgxError - no crash, no
    double free - wherever the failure lands, and a later healthy call still works"""
    from openmvg_amd import _capi, matching
    sc = synth.ba_scene(n_cams=4, n_points=20, track_len=3, model=3, n_intr_groups=2, seed=2)
    imgs = synth.image_descriptors(3, n_desc=40, seed=1)
    bins = synth.binary_descriptors(3, 40, seed=1)
    pairs = matching.exhaustive_pairs_array(3)
    failures = 0
    with _emu.emulated():
        for n in list(range(0, 40, 3)) + [55, 80, 110]:
            monkeypatch.setenv("HIPEMU_FAIL_MALLOC_AFTER", str(n))
            try:
                ctx = ba.BaContext(sc); ctx.solve(); ctx.close()
            except _capi.MvgxError:
                failures += 1
            monkeypatch.setenv("HIPEMU_FAIL_MALLOC_AFTER", str(n))
            try:
                m = matching.MatchContext(0); m.set_regions(imgs); m.run(pairs, np.float32(0.64)); m.close()
            except _capi.MvgxError:
                failures += 1
            monkeypatch.setenv("HIPEMU_FAIL_MALLOC_AFTER", str(n))
            try:
                h = matching.HammingContext(); h.set_regions(bins, 64); h.run(pairs, 0.8); h.close()
            except _capi.MvgxError:
                failures += 1
        monkeypatch.delenv("HIPEMU_FAIL_MALLOC_AFTER")
        ctx = ba.BaContext(sc); s = ctx.solve(); ctx.close()
        m = matching.MatchContext(0); m.set_regions(imgs); _, off, ij = m.run(pairs, np.float32(0.64)); m.close()
    assert failures >= 20 and s.termination == 0
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij)


# ---- block-sparse reduced camera system (ba_sparse_plan.h + the sp_* kernels) ---------------------------------------------
def _solve_emu_info(sc, options=None):
    with _emu.emulated():
        ctx = ba.BaContext(sc)
        s = ctx.solve(options)
        info = ctx.solver_info()
        poses, intr, pts = ctx.read_params()
        ctx.close()
    return s, info, poses, intr, pts


@pytest.mark.parametrize("leaf_cols", [64])
def test_block_sparse_solver_on_a_ring_equals_dense_and_oracle(monkeypatch, leaf_cols):
    """72 cameras on rings, tracks of 4 consecutive cameras, 3 shared intrinsics: S is a cyclic band + a dense border. The
    nested dissection must cut it into several parts (parallel levels), the shared intrinsics must go to the border, and the
    LM trajectory must equal the dense solver's and the oracle's."""
    sc = synth.ba_scene(n_cams=72, n_points=500, track_len=4, model=3, n_intr_groups=3, seed=71)
    opt = dict(max_num_iterations=3)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc, options=_oracle.default_ba_options(**opt))
    monkeypatch.setenv("MVGX_BA_SOLVER", "dense")
    sd, info_d, pd, idn, xd = _solve_emu_info(sc, ba.default_options(**opt))
    assert info_d.sparse == 0
    monkeypatch.setenv("MVGX_BA_SOLVER", "sparse")
    monkeypatch.setenv("MVGX_BA_ND_LEAF_COLS", str(leaf_cols))
    ss, info, ps, isn, xs = _solve_emu_info(sc, ba.default_options(**opt))
    assert info.sparse == 1 and info.n_border_blocks == 3 and info.n_columns == 6 * 72 + 8 * 3
    assert info.n_parts >= (7 if leaf_cols == 64 else 3) and info.n_levels < info.n_padded // 64
    assert ss.num_iterations == sd.num_iterations == osum.num_iterations == 3
    for s in (ss, sd):
        assert abs(s.final_cost - osum.final_cost) <= 1e-9 * osum.final_cost
    assert np.allclose(xs, xd, atol=1e-10) and np.allclose(ps, pd, atol=1e-10) and np.allclose(isn, idn, rtol=1e-10, atol=1e-10)
    assert np.allclose(xs, opx, atol=1e-9) and np.allclose(ps, opp, atol=1e-9) and np.allclose(isn, opi, rtol=1e-9, atol=1e-9)


def _lookahead_schedules_agree(monkeypatch, sc, iters, emulated):
    """The block-sparse solve under its three schedules: level by level (rounds 2 - 4: factor, T, U launches per level), look-ahead (round 5:
    one launch per level, the factor workgroups take the contributions of the level before themselves, the U tasks rebuild their strips
    of L), and look-ahead with every level on its fall-back form (MVGX_BA_LOOKAHEAD_MAX_PRE=0: tasks first, then the factorisation). The
    arithmetic and its order are the same by construction: every output must be equal bit for bit."""
    import contextlib
    from tests import _emu as emu
    out = {}
    for name, env in (("levels", {"MVGX_BA_LOOKAHEAD": "0"}), ("lookahead", {"MVGX_BA_LOOKAHEAD": "1"}),
                      ("fallback", {"MVGX_BA_LOOKAHEAD": "1", "MVGX_BA_LOOKAHEAD_MAX_PRE": "0"}), ("one", {"MVGX_BA_LOOKAHEAD": "1", "MVGX_BA_LOOKAHEAD_MAX_PRE": "1"}),
                      ("flags", {"MVGX_BA_BACKSOLVE_FLAGS": "1"})):   # (the reverse sweep as one launch, columns handed over through flags)
        monkeypatch.setenv("MVGX_BA_SOLVER", "sparse")
        monkeypatch.setenv("MVGX_BA_ND_LEAF_COLS", "64")
        for k in ("MVGX_BA_LOOKAHEAD", "MVGX_BA_LOOKAHEAD_MAX_PRE", "MVGX_BA_BACKSOLVE_FLAGS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with (emu.emulated() if emulated else contextlib.nullcontext()):
            ctx = ba.BaContext(sc)
            s = ctx.solve(ba.default_options(max_num_iterations=iters)); info = ctx.solver_info(); prm = ctx.read_params(); ctx.close()
        assert info.sparse == 1 and info.n_levels >= 3
        out[name] = (s.num_iterations, s.num_successful_steps, s.final_cost, s.final_rmse) + tuple(prm)
    for name in ("lookahead", "fallback", "one", "flags"):
        a, b = out["levels"], out[name]
        assert a[:4] == b[:4] and all(np.array_equal(x, y) for x, y in zip(a[4:], b[4:])), name


def test_lookahead_schedule_equals_the_level_by_level_schedule_emulated(monkeypatch):
    _lookahead_schedules_agree(monkeypatch, synth.ba_scene(n_cams=72, n_points=400, track_len=4, model=3, n_intr_groups=3, seed=71), 1, True)


def test_block_sparse_solver_is_the_default_when_the_factor_is_sparse():
    sc = synth.ba_scene(n_cams=120, n_points=700, track_len=4, model=1, n_intr_groups=1, seed=72)
    opt = dict(max_num_iterations=1)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc, options=_oracle.default_ba_options(**opt))
    s, info, poses, intr, pts = _solve_emu_info(sc, ba.default_options(**opt))
    assert info.sparse == 1 and info.n_factor_tiles < info.n_dense_tiles and info.n_levels < (info.n_columns + 63) // 64
    assert abs(s.final_cost - osum.final_cost) <= 1e-9 * osum.final_cost
    assert np.allclose(pts, opx, atol=1e-9) and np.allclose(poses, opp, atol=1e-9)


def test_block_sparse_solver_with_per_camera_intrinsics_and_constant_blocks(monkeypatch):
    """one intrinsic per camera (intrinsic blocks are NOT border: each couples with a few poses), some poses constant,
    some cameras unobserved: inactive blocks keep their unit diagonal inside the tiles"""
    sc = synth.ba_scene(n_cams=40, n_points=300, track_len=4, model=2, n_intr_groups=40, seed=73)
    keep = sc["obs_pose"] != 7            # camera 7 loses all its observations
    for k in ("obs_pose", "obs_intr", "obs_point"):
        sc[k] = sc[k][keep]
    sc["obs_xy"] = sc["obs_xy"][keep]; sc["n_obs"] = int(keep.sum())
    pm = np.zeros(40, np.uint8); pm[3] = 0x3F; pm[11] = 0x07
    opt = dict(max_num_iterations=2)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc, options=_oracle.default_ba_options(**opt), pose_const_mask=pm)
    monkeypatch.setenv("MVGX_BA_SOLVER", "sparse")
    monkeypatch.setenv("MVGX_BA_ND_LEAF_COLS", "64")
    with _emu.emulated():
        ctx = ba.BaContext(sc, pose_const_mask=pm)
        s = ctx.solve(ba.default_options(**opt)); info = ctx.solver_info(); poses, intr, pts = ctx.read_params(); ctx.close()
    assert info.sparse == 1 and info.n_border_blocks == 0 and info.n_parts > 3
    assert s.num_iterations == osum.num_iterations
    assert abs(s.final_cost - osum.final_cost) <= 1e-9 * osum.final_cost
    assert np.allclose(pts, opx, atol=1e-9) and np.allclose(poses, opp, atol=1e-9) and np.allclose(intr, opi, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_in_process_multi_device_context_equals_the_single_device_solve(devices):
    """mvgx_ba_create_multi: the library shards the problem itself (points by sum L_p^2, priors on the first shard), runs one
    host thread per shard and sums over peer-mapped buffers in rank order - here several emulated contexts of 'device 0'.
    Solve, read-back, residuals, track angles and evaluate come back in the caller's numbering."""
    sc = synth.ba_scene(n_cams=9, n_points=170, track_len=5, model=3, n_intr_groups=2, seed=97, rot_deg=0.3)
    sc = synth.add_pose_priors(synth.add_control_points(sc, n_ctrl=5, weight=20.0), sigma=0.005, huber_a=2e-4)
    perm = np.random.default_rng(4).permutation(sc["n_obs"])       # observations not sorted by point
    for k in ("obs_pose", "obs_intr", "obs_point", "obs_weight", "obs_is_control"):
        sc[k] = np.ascontiguousarray(np.asarray(sc[k])[perm])
    sc["obs_xy"] = np.ascontiguousarray(np.asarray(sc["obs_xy"]).reshape(-1, 2)[perm])
    opt = ba.default_options(max_num_iterations=4)
    with _emu.emulated():
        one = ba.BaContext(sc)
        s1 = one.solve(opt)
        p1, i1, x1 = one.read_params()
        r1, a1, e1 = one.residuals(), one.track_angles(), one.evaluate()
        one.close()
        many = ba.BaContext(sc, devices=devices)
        s2 = many.solve(opt)
        p2, i2, x2 = many.read_params()
        r2, a2, e2 = many.residuals(), many.track_angles(), many.evaluate()
        info = many.solver_info()
        many.close()
    assert s2.num_iterations == s1.num_iterations and s2.num_successful_steps == s1.num_successful_steps
    assert abs(s2.final_cost - s1.final_cost) <= 1e-9 * s1.final_cost and abs(s2.final_rmse - s1.final_rmse) < 1e-9
    assert abs(s2.initial_rmse - s1.initial_rmse) < 1e-12
    assert np.allclose(p2, p1, atol=1e-9) and np.allclose(i2, i1, rtol=1e-9, atol=1e-9) and np.allclose(x2, x1, atol=1e-8)
    assert np.allclose(r2, r1, atol=1e-8) and np.allclose(a2, a1, atol=1e-7)
    assert abs(e2[0] - e1[0]) <= 1e-9 * e1[0] and abs(e2[1] - e1[1]) < 1e-9
    assert info.n_columns == 6 * 9 + 8 * 2


def test_multi_device_from_the_environment_only_for_large_problems(monkeypatch):
    """mvgx_ba_create(-1): MVGX_DEVICES names the devices, MVGX_BA_MULTI_MIN_OBS the size from which a problem is sharded"""
    sc = synth.ba_scene(n_cams=6, n_points=80, track_len=4, model=1, n_intr_groups=1, seed=98)
    opt = ba.default_options(max_num_iterations=2)
    with _emu.emulated():
        c = ba.BaContext(sc); ref = c.solve(opt); c.close()
        monkeypatch.setenv("MVGX_DEVICES", "0,0")
        c = ba.BaContext(sc); s_small = c.solve(opt); c.close()          # below the default threshold: one device
        monkeypatch.setenv("MVGX_BA_MULTI_MIN_OBS", "10")
        c = ba.BaContext(sc); s_multi = c.solve(opt); c.close()
    for s in (s_small, s_multi):
        assert s.num_iterations == ref.num_iterations and abs(s.final_cost - ref.final_cost) <= 1e-9 * ref.final_cost


@pytest.mark.parametrize("kw", [
    dict(n_cams=12, n_points=300, track_len=6, model=3, n_intr_groups=2, seed=101),
    dict(n_cams=12, n_points=300, track_len=6, model=3, n_intr_groups=1, seed=108),       # one radial-K3 intrinsic: the strip form (columns 64 .. 67 on the 4 x 4 x 4 MFMA)
    dict(n_cams=14, n_points=260, track_len=10, model=2, n_intr_groups=1, seed=109),      # ... radial-K1 (four parameters), ten poses per point
    dict(n_cams=30, n_points=700, track_len=10, model=1, n_intr_groups=1, seed=102),      # ten poses per point: full groups
    dict(n_cams=16, n_points=200, track_len=12, model=1, n_intr_groups=1, seed=103),      # 11 .. 16 poses per point: the wide form of the groups (round 6)
    dict(n_cams=20, n_points=150, track_len=14, model=3, n_intr_groups=2, seed=106),      # ... with two local intrinsics
    dict(n_cams=20, n_points=120, track_len=18, model=1, n_intr_groups=1, seed=107),      # tracks longer than any group: flat list only
    dict(n_cams=10, n_points=260, track_len=10, model=3, n_intr_groups=2, seed=104),      # one camera set, two intrinsics: supergroups of several groups
    dict(n_cams=16, n_points=400, track_len=4, model=2, n_intr_groups=8, seed=105),       # many points with more than two intrinsics: both paths mixed
])
def test_point_groups_on_the_matrix_cores_equal_the_flat_product_list(kw, monkeypatch):
    """The fused point-group pass (ba_point_group_kernel: Jacobian evaluated in registers, per-point factors in LDS, Z^T Z of the
    group's dense matrix incl. the intrinsic columns on the f64 MFMA, partial blocks of all three product families, back-substitution
    by recomputation) must equal the record-based path (MVGX_BA_GROUPS=0: Jacobian records, flat product lists) and the oracle;
    constant points, a pose seen twice by one point, long tracks and points with more than two intrinsics stay on the record path"""
    sc = synth.ba_scene(**kw)
    sc = synth.add_control_points(sc, n_ctrl=4, weight=10.0)                  # constant points with observations
    sc["obs_pose"] = np.asarray(sc["obs_pose"]).copy()
    first = int(np.flatnonzero(np.asarray(sc["obs_point"]) == 3)[0])         # point 3: make two of its observations share a pose
    second = int(np.flatnonzero(np.asarray(sc["obs_point"]) == 3)[1])
    sc["obs_pose"][second] = sc["obs_pose"][first]
    opt = ba.default_options(max_num_iterations=3)
    with _emu.emulated():
        c = ba.BaContext(sc); s_g = c.solve(opt); pg, ig, xg = c.read_params(); info = c.solver_info(); c.close()
        monkeypatch.setenv("MVGX_BA_GROUPS", "0")
        c = ba.BaContext(sc); s_f = c.solve(opt); pf, if_, xf = c.read_params(); info_f = c.solver_info(); c.close()
    assert info_f.n_point_groups == 0
    if kw["n_intr_groups"] > 2:
        assert 0 < info.n_grouped_points < 0.9 * kw["n_points"]
    elif kw["track_len"] <= 16:
        assert info.n_point_groups > 0 and info.n_grouped_points > 0.8 * kw["n_points"]
    else:
        assert info.n_point_groups == 0
    assert s_g.num_iterations == s_f.num_iterations and abs(s_g.final_cost - s_f.final_cost) <= 1e-10 * s_f.final_cost
    assert np.allclose(pg, pf, atol=1e-9) and np.allclose(ig, if_, rtol=1e-9, atol=1e-9) and np.allclose(xg, xf, atol=1e-8)
    rc, osum, *_ = _oracle.port_ba_solve(sc, opt)
    assert abs(s_g.final_rmse - osum.final_rmse) < 1e-9


def test_candidate_cost_from_the_back_substitution_pass_equals_the_separate_passes(monkeypatch):
    """every point grouped: x + delta and its cost come from the back-substitution pass of the point groups (weights, a Huber-active
    outlier share, two intrinsics); MVGX_BA_SEPARATE_COST=1 keeps ba_step_scalars_kernel + ba_linearize_kernel<false> - same trajectory"""
    sc = synth.ba_scene(n_cams=8, n_points=120, track_len=5, model=3, n_intr_groups=2, seed=131, outlier_frac=0.05)
    opt = ba.default_options(max_num_iterations=4)
    with _emu.emulated():
        c = ba.BaContext(sc); s_a = c.solve(opt); pa, ia, xa = c.read_params(); info = c.solver_info(); c.close()
        monkeypatch.setenv("MVGX_BA_SEPARATE_COST", "1")
        c = ba.BaContext(sc); s_b = c.solve(opt); pb, ib, xb = c.read_params(); c.close()
    assert info.n_grouped_points == 120
    assert s_a.num_iterations == s_b.num_iterations and abs(s_a.final_cost - s_b.final_cost) <= 1e-12 * s_b.final_cost
    assert np.allclose(pa, pb, atol=1e-10) and np.allclose(ia, ib, rtol=1e-10, atol=1e-10) and np.allclose(xa, xb, atol=1e-9)
    rc, osum, *_ = _oracle.port_ba_solve(sc, opt)
    assert abs(s_a.final_rmse - osum.final_rmse) < 1e-9


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_model_cost_from_the_normal_equations_equals_the_jacobian_form(devices, monkeypatch):
    """model_cost_change of trust_region_minimizer.cc:402-405, -(J s)^T (r + J s / 2), is evaluated as (s^T D^2 s - s^T g) / 2 from the
    quantities the solver already holds (exact for a direct solve) instead of a pass over all observations; same LM trajectory as
    the Jacobian form (MVGX_BA_MODEL_COST=jacobian), with priors, control points, Huber-active outliers, rejected steps, 2 shards"""
    sc = synth.ba_scene(n_cams=9, n_points=160, track_len=5, model=3, n_intr_groups=2, seed=131, rot_deg=2.0, outlier_frac=0.05)
    sc = synth.add_pose_priors(synth.add_control_points(sc, n_ctrl=5, weight=20.0), sigma=0.005, huber_a=2e-4)
    opt = ba.default_options(max_num_iterations=8, initial_radius=1e7)     # a large first radius: some steps get rejected
    out = {}
    with _emu.emulated():
        for form in ("normal", "jacobian"):
            if form == "jacobian":
                monkeypatch.setenv("MVGX_BA_MODEL_COST", "jacobian")
            c = ba.BaContext(sc) if devices is None else ba.BaContext(sc, devices=devices)
            s = c.solve(opt)
            out[form] = (s, c.read_params())
            c.close()
    a, b = out["normal"][0], out["jacobian"][0]
    assert a.num_iterations == b.num_iterations and a.num_successful_steps == b.num_successful_steps
    assert abs(a.final_cost - b.final_cost) <= 1e-11 * b.final_cost
    for x, y in zip(out["normal"][1], out["jacobian"][1]):
        assert np.allclose(x, y, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("kw", [
    dict(n_cams=9, n_points=150, track_len=5, model=3, n_intr_groups=3, seed=141),
    dict(n_cams=6, n_points=700, track_len=6, model=4, n_intr_groups=1, seed=142),     # 4 200 observations of one intrinsic: several chunks, odd tails
])
def test_gram_blocks_on_the_matrix_cores_equal_the_oracle(kw):
    """Fc^T Fc, Fc^T Fi, Fi^T Fi and the gradients of a (pose, intrinsic) chunk as one F^T F on the f64 matrix cores, the rows
    evaluated by the kernel itself (ba_cam_gram_kernel, the intrinsic's blocks from the same pass; chunks of 512 observations with
    partial last rounds), with pose priors adding their rows: same LM trajectory as the oracle, parameters equal to rounding"""
    sc = synth.ba_scene(**kw)
    sc = synth.add_pose_priors(sc, sigma=0.005, huber_a=2e-4)
    opt = dict(max_num_iterations=4)
    with _emu.emulated():
        c = ba.BaContext(sc); s1 = c.solve(ba.default_options(**opt)); p1 = c.read_params(); c.close()
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc, options=_oracle.default_ba_options(**opt))
    assert rc == 0 and s1.num_iterations == osum.num_iterations and abs(s1.final_cost - osum.final_cost) <= 1e-9 * osum.final_cost
    assert abs(s1.initial_cost - osum.initial_cost) <= 1e-12 * osum.initial_cost
    for x, y in zip(p1, (opp, opi, opx)):
        assert np.allclose(x, y, rtol=1e-8, atol=1e-8)


def test_factor_and_invert_kernel_against_numpy():
    """the 64 x 64 Cholesky + inverse workgroup kernel under the emulation: full, partial (identity-padded) and tiny blocks"""
    from tests import _factor64
    _factor64.check_factor64(_emu.handle())
    _factor64.check_factor64_rejects_indefinite(_emu.handle())
