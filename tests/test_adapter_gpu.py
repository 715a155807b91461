"""Drop-in boundary test: the SAME caller code that drives the reference (oracle/ref_shim_*.cpp: builds Regions /
SfM_Data, constructs Matcher_Regions / Bundle_Adjustment_Ceres by name, calls Match / Adjust) is linked against the
MI355X replacement translation units (openmvg_amd/adapter/) instead of the reference's Matcher_Regions.cpp and
sfm_data_BA_ceres.cpp. Results must equal the reference's: match lists bit-exactly, BA final RMSE within 1e-6."""
import os

import numpy as np
import pytest

from openmvg_amd import synth
from openmvg_amd import matching
from tests import _oracle

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _oracle.have_adapter(), reason="adapter library not built (needs the openMVG tree)")]


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def _adapter_counters(lib, reset=False):
    import ctypes as C
    out = (C.c_uint64 * 3)()
    lib.mvgx_adapter_counters(out, 1 if reset else 0)
    return int(out[0]), int(out[1]), int(out[2])   # device pairs, fallback pairs, device failures


def test_matcher_regions_replacement_equals_reference():
    descs = synth.image_descriptors(7, n_desc=450, seed=11)
    descs[3] = descs[3][:0]          # an image without regions (Matcher_Regions.cpp:65-69,85-90)
    descs[5] = descs[5][:1]          # a database of one descriptor: NN=2 > rows (matcher_brute_force.hpp:108-113)
    pairs = matching.exhaustive_pairs_array(7)
    _adapter_counters(_oracle.adapter(), reset=True)
    got = _oracle.ref_matcher_regions_match(descs, pairs, 0.8, lib=_oracle.adapter())
    # the container came from the device: every pair counted there, the reference route behind the error policy never entered
    assert _adapter_counters(_oracle.adapter(), reset=True) == (len(pairs), 0, 0)
    off, ij = _oracle.port_matcher_regions_match(descs, pairs, 0.8)
    _same(got, _oracle.offsets_to_dict(pairs, off, ij))
    if _oracle.have_ref_match():
        _same(got, _oracle.ref_matcher_regions_match(descs, pairs, 0.8))


def test_matcher_regions_replacement_ratio_above_one_uses_reference_route():
    """ratio > 1: tie order is libstdc++'s partial_sort — the replacement must route to the reference's own matcher."""
    if not _oracle.have_ref_match():
        pytest.skip("needs oracle/_ref")
    descs = synth.image_descriptors(3, n_desc=200, seed=12)
    pairs = matching.exhaustive_pairs_array(3)
    _same(_oracle.ref_matcher_regions_match(descs, pairs, 1.05, lib=_oracle.adapter()),
          _oracle.ref_matcher_regions_match(descs, pairs, 1.05))


def test_matcher_regions_replacement_hamming_equals_reference():
    """-n BRUTEFORCEHAMMING on AKAZE_Binary_Regions: the same caller shim linked against the reference TUs
    (oracle/_ref/libref_match.so) and against the MI355X replacement (adapter library)"""
    from openmvg_amd import matching, synth
    sizes = [300, 0, 257, 64, 1, 2, 500]
    imgs = synth.binary_descriptors(len(sizes), sizes, seed=9)
    pairs = matching.exhaustive_pairs_array(len(sizes))
    got = _oracle.ref_matcher_regions_match_binary64(imgs, pairs, 0.8, lib=_oracle.adapter())
    o_off, o_ij = _oracle.port_matcher_regions_match_hamming(imgs, pairs, 0.8)
    want = _oracle.offsets_to_dict(pairs, o_off, o_ij)
    assert sum(len(v) for v in want.values()) > 100
    _same(got, want)
    if _oracle.have_ref_match():
        _same(got, _oracle.ref_matcher_regions_match_binary64(imgs, pairs, 0.8))


def test_matcher_regions_replacement_float_equals_reference():
    """-n BRUTEFORCEL2 on AKAZE_Float_Regions (64 floats): same caller shim, reference TUs vs the MI355X replacement;
    the lists are identical because the device sums in the reference's order"""
    from openmvg_amd import matching, synth
    sizes = [300, 0, 257, 64, 1, 2, 500]
    imgs = synth.float_descriptors(len(sizes), sizes, seed=9)
    pairs = matching.exhaustive_pairs_array(len(sizes))
    got = _oracle.ref_matcher_regions_match_float64(imgs, pairs, 0.8, lib=_oracle.adapter())
    o_off, o_ij = _oracle.port_matcher_regions_match_f32(imgs, pairs, 0.8)
    want = _oracle.offsets_to_dict(pairs, o_off, o_ij)
    assert sum(len(v) for v in want.values()) > 100
    _same(got, want)
    if _oracle.have_ref_match():
        _same(got, _oracle.ref_matcher_regions_match_float64(imgs, pairs, 0.8))


def test_matcher_regions_replacement_liop_equals_reference():
    """-n BRUTEFORCEL2 on AKAZE_Liop_Regions (144 x uint8): same caller shim, reference TUs vs the MI355X replacement"""
    from tests.test_l2u8_cpu import liop_like
    sizes = [300, 0, 257, 64, 1, 2, 500]
    imgs = liop_like(sizes, 144, seed=21)
    pairs = matching.exhaustive_pairs_array(len(sizes))
    got = _oracle.ref_matcher_regions_match_liop144(imgs, pairs, 0.8, lib=_oracle.adapter())
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8, dim=144)
    want = _oracle.offsets_to_dict(pairs, o_off, o_ij)
    assert sum(len(v) for v in want.values()) > 50
    _same(got, want)
    if _oracle.have_ref_match():
        _same(got, _oracle.ref_matcher_regions_match_liop144(imgs, pairs, 0.8))


def _golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba_golden.npz"))


@pytest.mark.parametrize("tag", list(_golden()["case_names"]))
def test_bundle_adjustment_ceres_replacement_equals_golden(tag):
    z = _golden()
    keys = ("poses", "intrinsics", "intr_model", "points", "obs_pose", "obs_intr", "obs_point", "obs_xy")
    sc = {k: z[f"{tag}/{k}"].copy() for k in keys}
    sc["n_poses"] = len(sc["poses"]); sc["n_intrinsics"] = len(sc["intrinsics"]); sc["n_points"] = len(sc["points"])
    sc["n_obs"] = len(sc["obs_pose"])
    _, iopt, eopt, sopt = tag.split("|")
    ref_stats = z[f"{tag}/ref_stats"]     # {rmse_before, rmse_after, seconds, Adjust() return} from the reference
    rc, stats, poses, intr, pts = _oracle.ref_ba_adjust(sc, int(iopt), int(eopt), int(sopt), lib=_oracle.adapter())
    assert rc == 0 and stats[3] == ref_stats[3] == 1.0
    assert abs(stats[0] - ref_stats[0]) < 1e-9                      # same scene going in (reference's own RMSE helper)
    assert abs(stats[1] - ref_stats[1]) < 1e-6 * max(1.0, ref_stats[1]), (stats[1], ref_stats[1])
    if ref_stats[1] < 100:
        assert np.allclose(pts, z[f"{tag}/ref_points"], atol=1e-4)


@pytest.mark.parametrize("name", list(_golden()["ex_case_names"]))
def test_bundle_adjustment_ceres_replacement_functors_control_points_priors(name):
    """Brown / fisheye / spherical cameras, Control_Point_Parameter(20, true) and use_motion_priors_opt through the
    replacement Bundle_Adjustment_Ceres::Adjust, against the reference's outputs on the same SfM_Data"""
    from tests.test_ba_gpu import _ex_case
    z = _golden()
    tag, sc = _ex_case(z, name)
    iopt = int(name.split("|")[1])
    ref_stats = z[f"{tag}/ref_stats"]
    rc, stats, poses, intr, pts = _oracle.ref_ba_adjust_ex(sc, intrinsics_opt=iopt, lib=_oracle.adapter())
    assert rc == 0 and stats[3] == ref_stats[3] == 1.0
    assert abs(stats[0] - ref_stats[0]) < 1e-9
    assert abs(stats[1] - ref_stats[1]) < 1e-6, (stats[1], ref_stats[1])
    assert np.allclose(pts, z[f"{tag}/ref_points"], atol=1e-5)
    assert np.allclose(poses[:, 3:], z[f"{tag}/ref_poses"][:, 3:], atol=1e-5)


def test_bundle_adjustment_unsupported_model_returns_false():
    sc = synth.ba_scene(4, 30, track_len=3, model=1, seed=3)
    # the shim only builds pinhole / K1 / K3 cameras (-3 otherwise); an unsupported Adjust() is covered through the C ABI
    # in test_ba_gpu.py::test_error_behaviour. Here: Adjust on a healthy scene returns true and lowers the RMSE.
    rc, stats, *_ = _oracle.ref_ba_adjust(sc, lib=_oracle.adapter())
    assert rc == 0 and stats[1] < stats[0]


def test_matcher_regions_replacement_uses_the_devices_of_the_environment(monkeypatch):
    """Unchanged callers reach several GPUs through the environment: MVGX_DEVICES=0,0 makes Matcher_Regions::Match of the
    replacement TU run two device contexts (here on the one GPU of the box) - same container."""
    descs = synth.image_descriptors(30, n_desc=700, seed=13)
    descs[4] = descs[4][:0]
    pairs = matching.exhaustive_pairs_array(30)
    one = _oracle.ref_matcher_regions_match(descs, pairs, 0.8, lib=_oracle.adapter())
    monkeypatch.setenv("MVGX_DEVICES", "0,0")
    two = _oracle.ref_matcher_regions_match(descs, pairs, 0.8, lib=_oracle.adapter())
    _same(one, two)
    off, ij = _oracle.port_matcher_regions_match(descs, pairs, 0.8)
    _same(two, _oracle.offsets_to_dict(pairs, off, ij))


@pytest.mark.parametrize("name", list(_golden()["ex_case_names"])[-3:])
def test_bundle_adjustment_ceres_replacement_on_the_devices_of_the_environment(name, monkeypatch):
    """Unchanged callers of Bundle_Adjustment_Ceres::Adjust reach several GPUs through the environment (MVGX_DEVICES; here
    two shards on the one GPU of the box, threshold lowered so that the small golden scenes are sharded): same outputs as
    the reference's on control-point / prior scenes."""
    from tests.test_ba_gpu import _ex_case
    monkeypatch.setenv("MVGX_DEVICES", "0,0")
    monkeypatch.setenv("MVGX_BA_MULTI_MIN_OBS", "1")
    z = _golden()
    tag, sc = _ex_case(z, name)
    iopt = int(name.split("|")[1])
    ref_stats = z[f"{tag}/ref_stats"]
    rc, stats, poses, intr, pts = _oracle.ref_ba_adjust_ex(sc, intrinsics_opt=iopt, lib=_oracle.adapter())
    assert rc == 0 and stats[3] == ref_stats[3] == 1.0
    assert abs(stats[1] - ref_stats[1]) < 1e-6, (stats[1], ref_stats[1])
    assert np.allclose(pts, z[f"{tag}/ref_points"], atol=1e-5)


@pytest.mark.parametrize("tag", ["synthetic", "synthetic_grid", "sceaux"])
def test_cascade_hashing_replacement_equals_the_reference_lists(tag):
    """-n CASCADEHASHINGL2 (main_ComputeMatches' default) through the replacement TU: hashing by the reference's CascadeHasher on
    the host, matching stage on the MI355X, the reference's de-duplication classes - containers equal the reference's stored ones
    (synthetic set with an empty image, a grid of repeated feature positions, the real SceauxCastle regions), order included"""
    from tests.test_cascade import load
    descs, xy, hs, bs, pairs, ref = load(tag)
    for ratio in (0.8, 0.6):
        got = _oracle.ref_cascade_matcher_regions_match(descs, xy, pairs, ratio, lib=_oracle.adapter())
        _same(got, ref[int(ratio * 100)])
    if _oracle.have_ref_match():    # and live against the reference TU at another ratio
        _same(_oracle.ref_cascade_matcher_regions_match(descs, xy, pairs, 0.9, lib=_oracle.adapter()),
              _oracle.ref_cascade_matcher_regions_match(descs, xy, pairs, 0.9))
