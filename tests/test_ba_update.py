"""mvgx_ba_update (include/mvgx.h): a context re-bound to new VALUES of the structure it was created from must behave exactly like a
context created from those values - same grouping, same summation order, hence bit-identical results - and must refuse a problem
of another structure without being disturbed. Call pattern in the reference: consecutive Bundle_Adjustment::Adjust calls on one
scene (global_SfM.cpp's refinement passes with growing parameter sets, sequential_SfM.cpp:1190-1232).
CPU tests run the device code under the HIP emulation (tests/_emu.py); the `gpu` tests run the same checks on the MI355X."""
import numpy as np
import pytest

from openmvg_amd import _capi, ba, synth
from openmvg_amd import ba_options as bo
from tests import _emu


_MAX_ITERATIONS = 4   # (the emulation runs an LM iteration in about a second: equality of the two sides does not need a converged solve)


def _perturbed(sc, seed):
    """same structure, other values: parameters, image points"""
    rng = np.random.default_rng(seed)
    out = dict(sc)
    out["poses"] = sc["poses"] + rng.normal(0, 2e-3, sc["poses"].shape)
    out["points"] = sc["points"] + rng.normal(0, 5e-3, sc["points"].shape)
    out["intrinsics"] = sc["intrinsics"].copy()
    out["intrinsics"][:, 0] *= 1.002
    out["obs_xy"] = sc["obs_xy"] + rng.normal(0, 0.3, sc["obs_xy"].shape)
    return out


def _run(ctx):
    s = ctx.solve(ba.default_options(max_num_iterations=_MAX_ITERATIONS))
    return (s.num_iterations, s.num_successful_steps, s.termination, s.initial_cost, s.final_cost, s.final_rmse) + tuple(ctx.read_params())


def _same(a, b):
    return a[:6] == b[:6] and all(np.array_equal(x, y) for x, y in zip(a[6:], b[6:]))


def _check_update_equals_create(sc, **ctx_args):
    sc2 = _perturbed(sc, 7)
    c = ba.BaContext(sc, **ctx_args)
    first = _run(c)
    assert c.update(sc2) is True
    second = _run(c)
    assert c.update(sc) is True          # and back: the first problem again, from the used context
    third = _run(c)
    c.close()
    f = ba.BaContext(sc2, **ctx_args)
    fresh = _run(f)
    f.close()
    assert _same(second, fresh), "updated context differs from a context created from the same values"
    assert _same(third, first), "re-bound to the first values, the context does not repeat its first solve"
    assert not _same(first, second)
    return first, second


def _check_masks_are_values(sc):
    """the refinement passes of global_SfM.cpp:  translations + structure  ->  everything: same structure, other constant masks"""
    m1 = bo.masks_for(sc, 1, 4, 1)    # intrinsics NONE (constant), ADJUST_TRANSLATION, structure ADJUST_ALL
    m2 = bo.masks_for(sc, 14, 6, 1)   # ADJUST_ALL intrinsics (focal | pp | distortion), ADJUST_ALL extrinsics
    c = ba.BaContext(sc, **m1)
    r1 = _run(c)
    assert c.update(sc, **m2) is True
    r2 = _run(c)
    c.close()
    f = ba.BaContext(sc, **m2)
    fresh = _run(f)
    f.close()
    assert _same(r2, fresh) and not _same(r1, r2)


def _check_structure_change_is_refused(sc):
    c = ba.BaContext(sc)
    first = _run(c)
    other = dict(sc)
    other["obs_pose"] = sc["obs_pose"].copy()
    k = int(np.flatnonzero(other["obs_pose"] != other["obs_pose"][0])[0])
    other["obs_pose"][k] = other["obs_pose"][0]          # one observation moved to another pose
    assert c.update(other) is False
    assert b"structure" in _capi.lib().mvgx_last_error()
    fewer = {k_: (v[:-1] if k_ in ("obs_pose", "obs_intr", "obs_point") else v[:-2] if k_ == "obs_xy" else v) for k_, v in sc.items()}
    fewer["n_obs"] = int(sc["n_obs"]) - 1
    assert c.update(fewer) is False
    assert c.update(sc, points_constant=True) is False   # which points are free is structure
    assert c.update(sc) is True                           # untouched by the refusals
    assert _same(_run(c), first)
    c.close()


def _scene():
    return synth.ba_scene(n_cams=6, n_points=48, track_len=4, model=3, n_intr_groups=2, seed=41, rot_deg=0.3)


def test_update_equals_create_emulated():
    with _emu.emulated():
        _check_update_equals_create(_scene())


def test_update_with_unsorted_observations_weights_and_priors_emulated():
    """the caller's list not in point order (the permutation of create is reused), control points with weights, pose priors"""
    sc = synth.ba_scene(n_cams=6, n_points=40, track_len=4, model=1, n_intr_groups=1, seed=43, rot_deg=0.3)
    rng = np.random.default_rng(5)
    perm = rng.permutation(int(sc["n_obs"]))
    for k in ("obs_pose", "obs_intr", "obs_point"):
        sc[k] = np.ascontiguousarray(sc[k][perm])
    sc["obs_xy"] = np.ascontiguousarray(sc["obs_xy"].reshape(-1, 2)[perm].reshape(-1))
    n_obs, n_pts = int(sc["n_obs"]), int(sc["n_points"])
    ctrl_pts = np.zeros(n_pts, np.uint8); ctrl_pts[:3] = 1
    sc["point_const_mask"] = ctrl_pts
    is_ctrl = ctrl_pts[sc["obs_point"]].astype(np.uint8)
    sc["obs_is_control"] = is_ctrl
    sc["obs_weight"] = np.where(is_ctrl, 20.0, 0.0)
    sc["prior_pose"] = np.arange(4, dtype=np.uint32)
    sc["prior_center"] = rng.normal(0, 1, 12)
    sc["prior_weight"] = np.full(12, 0.5)
    sc["prior_huber_a"] = 0.25
    with _emu.emulated():
        sc2 = _perturbed(sc, 9)
        sc2["obs_weight"] = np.where(is_ctrl, 35.0, 0.0)
        sc2["prior_center"] = sc["prior_center"] + 0.05
        c = ba.BaContext(sc)
        _run(c)
        assert c.update(sc2) is True
        second = _run(c)
        c.close()
        f = ba.BaContext(sc2)
        fresh = _run(f)
        f.close()
        assert _same(second, fresh)


def test_constant_masks_are_values_emulated():
    with _emu.emulated():
        _check_masks_are_values(_scene())


def test_structure_change_is_refused_emulated():
    with _emu.emulated():
        _check_structure_change_is_refused(_scene())


def test_update_of_a_two_shard_context_emulated(monkeypatch):
    monkeypatch.setenv("MVGX_BA_MULTI_MIN_OBS", "1")
    sc = _scene()
    with _emu.emulated():
        first, second = _check_update_equals_create(sc, devices=[0, 0])
        one = ba.BaContext(_perturbed(sc, 7))
        ref = _run(one)
        one.close()
    assert second[:3] == ref[:3] and abs(second[5] - ref[5]) < 1e-12


# ---------------------------------------------------------------------------------------------------- MI355X
@pytest.mark.gpu
def test_update_equals_create_on_the_device():
    sc = synth.ba_scene(n_cams=60, n_points=20000, track_len=8, model=3, n_intr_groups=4, seed=51)
    _check_update_equals_create(sc)
    _check_masks_are_values(sc)
    _check_structure_change_is_refused(sc)


@pytest.mark.gpu
def test_update_of_a_two_shard_context_on_the_device(monkeypatch):
    monkeypatch.setenv("MVGX_BA_MULTI_MIN_OBS", "1")
    monkeypatch.setenv("MVGX_BA_TRANSPORT", "peer")
    sc = synth.ba_scene(n_cams=40, n_points=8000, track_len=6, model=3, n_intr_groups=2, seed=52)
    _check_update_equals_create(sc, devices=[0, 0])


@pytest.mark.gpu
def test_update_is_cheaper_than_create_on_the_device():
    import time
    sc = synth.ba_scene(n_cams=200, n_points=100000, track_len=10, model=3, n_intr_groups=1, seed=53)
    sc2 = _perturbed(sc, 3)
    c = ba.BaContext(sc); c.solve(); c.close()      # warm: slab caches, host workers
    t0 = time.perf_counter(); c = ba.BaContext(sc); t_create = time.perf_counter() - t0
    c.solve()
    t0 = time.perf_counter(); assert c.update(sc2); t_update = time.perf_counter() - t0
    c.close()
    print(f"create {t_create * 1e3:.2f} ms, update {t_update * 1e3:.2f} ms")
    assert t_update < 0.5 * t_create


@pytest.mark.parametrize("case", range(4))
def test_update_of_degenerate_problems_emulated(case):
    """the edge scenes of tests/test_ba_emu_cpu.py (no observations, unused blocks, everything constant ...): re-binding each to
    itself repeats its solve; none of them is taken for another's structure"""
    from tests.test_ba_emu_cpu import _edge_scenes
    scenes = _edge_scenes()
    name, sc, masks = scenes[case]
    with _emu.emulated():
        c = ba.BaContext(sc, **masks)
        first = _run(c)
        assert c.update(sc, **masks) is True, name
        assert _same(_run(c), first), name
        other_name, other, other_masks = scenes[(case + 1) % 4]
        same_structure = all(np.array_equal(np.asarray(sc[k]), np.asarray(other[k])) for k in ("obs_pose", "obs_intr", "obs_point", "intr_model")) and \
            all(int(sc[k]) == int(other[k]) for k in ("n_poses", "n_intrinsics", "n_points")) and \
            bool(masks.get("points_constant", False)) == bool(other_masks.get("points_constant", False))
        assert c.update(other, **other_masks) is same_structure, (name, other_name)
        c.close()


# ------------------------------------------------------------------------------------------- mvgx_ba_update_subset
def _reduced(sc, enabled):
    """the scene a caller is left with after erasing the observations that are off (same point / pose numbering: a point or pose
    left without observations stays in the arrays, unobserved)"""
    out = dict(sc)
    keep = np.asarray(enabled, bool)
    for k in ("obs_pose", "obs_intr", "obs_point"):
        out[k] = np.ascontiguousarray(sc[k][keep])
    out["obs_xy"] = np.ascontiguousarray(sc["obs_xy"].reshape(-1, 2)[keep].reshape(-1))
    out["n_obs"] = int(keep.sum())
    return out


def _subset_mask(sc, seed):
    rng = np.random.default_rng(seed)
    en = rng.random(int(sc["n_obs"])) > 0.06
    en[sc["obs_point"] == 3] = False              # a track erased entirely
    en[sc["obs_point"] == 11] = False
    one = np.flatnonzero(sc["obs_point"] == 20)   # a track cut down to one observation
    en[one[1:]] = False
    return en


def _check_subset_equals_reduced_scene(sc, seed=5, **ctx_args):
    en = _subset_mask(sc, seed)
    c = ba.BaContext(sc, **ctx_args)
    full = _run(c)
    assert c.update(sc, obs_enabled=en, **{k: v for k, v in ctx_args.items() if k != "devices"}) is True
    sub = _run(c)
    res_sub = c.residuals(); ang_sub = c.track_angles()
    assert c.update(sc, **{k: v for k, v in ctx_args.items() if k != "devices"}) is True      # everything back on
    again = _run(c)
    c.close()
    red = _reduced(sc, en)
    f = ba.BaContext(red, **ctx_args)
    fresh = _run(f)
    res_f = f.residuals(); ang_f = f.track_angles()
    f.close()
    assert sub[:3] == fresh[:3], (sub[:3], fresh[:3])                      # iterations, successful steps, termination
    assert abs(sub[3] - fresh[3]) <= 1e-12 * fresh[3] and abs(sub[4] - fresh[4]) <= 1e-11 * fresh[4]
    assert abs(sub[5] - fresh[5]) < 1e-12                                    # final RMSE over the observations that are on
    for a, b in zip(sub[6:], fresh[6:]):   # (other point groups, other order of the sums: weakly determined distortion terms move in their 8th digit)
        assert np.allclose(a, b, rtol=1e-7, atol=1e-8), np.abs(a - b).max()
    assert np.allclose(res_sub[en], res_f, rtol=0, atol=1e-9)
    assert np.allclose(ang_sub, ang_f, rtol=0, atol=1e-9)
    assert _same(again, full), "switching every observation back on does not restore the full problem"
    return sub, full


def test_subset_equals_the_reduced_scene_emulated():
    with _emu.emulated():
        sub, full = _check_subset_equals_reduced_scene(_scene())
    assert sub[3] != full[3]


def _check_disabled_observations_with_degenerate_projections(sc):
    """an observation that is switched off must be SELECTED out, not multiplied by a zero weight - its projection is still
    evaluated, and a point at the camera's centre (0 / 0) or on its principal plane (x / 0) gave inf * 0 = NaN in the cost and the Gram
    blocks of the whole problem. Two points lose all their observations and are moved to such places; the solve must equal the one of
    the scene without those observations, and evaluate() must divide by the count of what is on."""
    sc = dict(sc)
    en = np.ones(int(sc["n_obs"]), bool)
    pts = np.array(sc["points"], float).reshape(-1, 3)
    poses = np.array(sc["poses"], float).reshape(-1, 6)
    # exactly degenerate projections need exact arithmetic: the pose that sees both points gets a zero rotation (X_cam = X + t, no rounding)
    cam = int(sc["obs_pose"][np.flatnonzero(sc["obs_point"] == 5)[0]])
    poses[cam, :3] = 0.0
    t = poses[cam, 3:]
    for which, ix in enumerate((5, 9)):
        en[sc["obs_point"] == ix] = False
        pts[ix] = -t if which == 0 else np.array([-t[0] + 0.75, -t[1] - 0.25, -t[2]])   # the camera's centre (0 / 0) / its principal plane (x / 0)
        assert (pts[ix] + t)[2] == 0.0
    extra = np.flatnonzero((sc["obs_pose"] == cam) & en)[:1]   # ... observed by that camera: append a disabled observation of each point
    for ix in (5, 9):
        for k in ("obs_pose", "obs_intr", "obs_point"):
            sc[k] = np.ascontiguousarray(np.concatenate([sc[k], [cam if k == "obs_pose" else sc[k][extra[0]] if k == "obs_intr" else ix]]).astype(sc[k].dtype))
        sc["obs_xy"] = np.ascontiguousarray(np.concatenate([np.asarray(sc["obs_xy"], float).reshape(-1), [10.0, 20.0]]))
        en = np.concatenate([en, [False]])
    sc["n_obs"] = int(len(en))
    sc["poses"] = np.ascontiguousarray(poses.reshape(-1))
    sc["points"] = np.ascontiguousarray(pts.reshape(-1))
    c = ba.BaContext(sc)
    assert c.update(sc, obs_enabled=en) is True
    cost0, rmse0 = c.evaluate()
    sub = _run(c)
    c.close()
    f = ba.BaContext(_reduced(sc, en))
    cost_f, rmse_f = f.evaluate()
    fresh = _run(f)
    f.close()
    assert np.isfinite(cost0) and np.isfinite(rmse0) and all(np.isfinite(v) for v in sub[3:6]), (cost0, rmse0, sub[3:6])
    assert abs(cost0 - cost_f) <= 1e-12 * cost_f and abs(rmse0 - rmse_f) <= 1e-12 * rmse_f, (cost0, cost_f, rmse0, rmse_f)   # (evaluate() counts what is on)
    assert sub[:3] == fresh[:3] and abs(sub[5] - fresh[5]) < 1e-12
    assert np.array_equal(sub[8][5], pts[5]) and np.array_equal(sub[8][9], pts[9])   # unobserved points do not move


def test_disabled_observations_with_degenerate_projections_emulated():
    with _emu.emulated():
        _check_disabled_observations_with_degenerate_projections(_scene())


@pytest.mark.gpu
def test_disabled_observations_with_degenerate_projections_on_the_device():
    _check_disabled_observations_with_degenerate_projections(synth.ba_scene(n_cams=40, n_points=6000, track_len=6, model=3, n_intr_groups=2, seed=58))


def test_subset_with_unsorted_observations_and_constant_intrinsics_emulated():
    sc = synth.ba_scene(n_cams=6, n_points=48, track_len=4, model=1, n_intr_groups=2, seed=47, rot_deg=0.3)
    perm = np.random.default_rng(2).permutation(int(sc["n_obs"]))
    for k in ("obs_pose", "obs_intr", "obs_point"):
        sc[k] = np.ascontiguousarray(sc[k][perm])
    sc["obs_xy"] = np.ascontiguousarray(sc["obs_xy"].reshape(-1, 2)[perm].reshape(-1))
    with _emu.emulated():
        _check_subset_equals_reduced_scene(sc, seed=8, **bo.masks_for(sc, 1, 6, 1))


def test_subset_that_empties_a_pose_emulated():
    """every observation of one pose off: its block leaves the program (unit diagonal, no step), as in a scene without it"""
    sc = _scene()
    en = np.ones(int(sc["n_obs"]), bool)
    en[sc["obs_pose"] == 2] = False
    with _emu.emulated():
        c = ba.BaContext(sc)
        assert c.update(sc, obs_enabled=en) is True
        sub = _run(c)
        c.close()
        f = ba.BaContext(_reduced(sc, en))
        fresh = _run(f)
        f.close()
    assert sub[:3] == fresh[:3] and abs(sub[5] - fresh[5]) < 1e-12
    assert np.array_equal(sub[6][2], sc["poses"][2]) and np.allclose(sub[6], fresh[6], rtol=1e-7, atol=1e-8)


@pytest.mark.gpu
def test_subset_equals_the_reduced_scene_on_the_device():
    sc = synth.ba_scene(n_cams=60, n_points=20000, track_len=8, model=3, n_intr_groups=4, seed=55)
    _check_subset_equals_reduced_scene(sc)


def test_residuals_and_angles_leave_the_context_intact_emulated():
    """mvgx_ba_residuals / mvgx_ba_track_angles on a scene whose points are all grouped, then the same solve again: until round 4
    both borrowed an array such a scene allocates empty and wrote over what lay behind it"""
    sc = _scene()
    with _emu.emulated():
        c = ba.BaContext(sc)
        first = _run(c)
        c.residuals(); c.track_angles()
        assert c.update(sc) is True
        assert _same(_run(c), first)
        c.residuals(); c.track_angles()
        c.close()
