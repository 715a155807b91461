"""Geometric filter, CPU side: the restatement against the compiled reference and its stored outputs, the
device code under the HIP emulation against the same, the host mirror, argument handling."""
import ctypes as C

import numpy as np
import pytest

from openmvg_amd import _capi, geofilter, synth
from tests import _emu, _geofilter_cases as gc, _oracle

GOLD = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "geofilter.npz"))


def _gold_tv(sel=None):
    start = GOLD["start"].astype(np.int64)
    pairs = range(len(start) - 1) if sel is None else sel
    xs_i, xs_j, st, wh = [], [], [0], []
    idx = []
    for p in pairs:
        xs_i.append(GOLD["xI"][start[p]:start[p + 1]]); xs_j.append(GOLD["xJ"][start[p]:start[p + 1]])
        st.append(st[-1] + int(start[p + 1] - start[p])); wh.append(GOLD["wh"][p]); idx.append(p)
    tv = dict(xI=np.concatenate(xs_i), xJ=np.concatenate(xs_j), start=np.asarray(st, np.uint64), wh=np.asarray(wh, np.uint32))
    ref = dict(mask=np.concatenate([GOLD["mask"][start[p]:start[p + 1]] for p in idx]), ok=GOLD["ok"][idx], F=GOLD["F"][idx],
               precision=GOLD["precision"][idx], nfa=GOLD["nfa"][idx])
    return tv, ref


def test_restatement_equals_the_stored_reference_outputs():
    """oracle/geofilter_oracle.cpp on the golden inputs: same inlier sets as the reference (stored), NFA / precision / F per policy;
    the count under policy (b) is bounded by the reference's own build-to-build spread (gc.allowed_differing)"""
    tv, ref = _gold_tv()
    got = _oracle.port_geofilter(tv, float(GOLD["precision_px"]), int(GOLD["max_iterations"]))
    differing, rep = gc.compare(tv["start"], ref, got["mask"], got["ok"], got["F"], got["precision"], got["nfa"])
    assert rep["pairs_ok_reference"] > 100 and len(differing) <= gc.allowed_differing(rep["pairs"]), rep


@pytest.mark.skipif(not _oracle.have_ref_geofilter(), reason="oracle/_ref/libref_geofilter.so not built (needs /root/reference)")
def test_restatement_equals_the_compiled_reference_live():
    tv = synth.two_view_matches(400, seed=77, n_max=250)
    ref = _oracle.ref_geofilter(tv)
    got = _oracle.port_geofilter(tv)
    differing, rep = gc.compare(tv["start"], ref, got["mask"], got["ok"], got["F"], got["precision"], got["nfa"])
    assert len(differing) <= gc.allowed_differing(rep["pairs"]), rep
    # few iterations: the max-consensus warm-up and its early exit decide (robust_estimator_ACRansac.hpp:445-451)
    ref2 = _oracle.ref_geofilter(tv, max_iterations=40); got2 = _oracle.port_geofilter(tv, max_iterations=40)
    differing2, rep2 = gc.compare(tv["start"], ref2, got2["mask"], got2["ok"], got2["F"], got2["precision"], got2["nfa"])
    assert len(differing2) <= gc.allowed_differing(rep2["pairs"]), rep2


def test_emulated_device_code_equals_the_stored_reference_outputs():
    """the kernel of openmvg_amd/csrc/mvgx_geofilter.hip under tests/native/hipemu (one fiber per lane) on a handful of golden pairs,
    incl. one without geometry, one below 8 correspondences"""
    start = GOLD["start"].astype(np.int64)
    n = np.diff(start)
    small = [int(p) for p in np.argsort(n) if n[p] <= 90]
    sel = small[:2] + [p for p in small if GOLD["ok"][p]][:4] + [p for p in small if not GOLD["ok"][p] and n[p] > 7][:2]
    tv, ref = _gold_tv(sel)
    with _emu.emulated():
        mask, res, st = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], geofilter.GeometricFilter_FMatrix_AC(4.0, 2048))
    differing, rep = gc.compare(tv["start"], ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"])
    assert not differing, (rep, differing)
    assert int(st.n_pairs_ok) == int(ref["ok"].sum())


def test_container_mirror_and_argument_errors():
    with _emu.emulated():
        assert geofilter.Robust_model_estimation({}, [], []) == {}
        xI = np.zeros((3, 2)); start = np.array([0, 3], np.uint64); wh = np.array([[100, 100, 100, 100]], np.uint32)
        mask, res, st = geofilter.filter_pairs(xI, xI, start, wh)   # fewer than 8 correspondences: rejected without estimation
        assert not mask.any() and not res["ok"][0] and res["n_inliers"][0] == 0 and np.array_equal(res["F"][0], np.eye(3))
        with pytest.raises(_capi.MvgxError) as e:
            geofilter.filter_pairs(xI, xI, start, wh, geofilter.GeometricFilter_FMatrix_AC(float("inf"), 1024))
        assert e.value.code == _capi.MVGX_ERR_UNSUPPORTED


def test_indexed_form_equals_the_gathered_form_under_emulation():
    """mvgx_geofilter_f_acransac_indexed (feature positions per image + index pairs, gathered on the device) returns what the
    gathered form returns on the same correspondences; indices out of range are argument errors"""
    start = GOLD["start"].astype(np.int64)
    n = np.diff(start)
    small = [int(p) for p in np.argsort(n) if 7 < n[p] <= 80]
    sel = small[:1] + [p for p in small if GOLD["ok"][p]][:3]
    tv, ref = _gold_tv(sel)
    st0 = tv["start"].astype(np.int64)
    feats, sizes, pairs, ij = [], [], [], []
    for p in range(len(sel)):   # images 2 p, 2 p + 1 hold the pair's features in shuffled order, plus three unused features each
        m = int(st0[p + 1] - st0[p])
        pi, pj = np.random.default_rng(p).permutation(m + 3)[:m], np.random.default_rng(50 + p).permutation(m + 3)[:m]
        fi = np.full((m + 3, 2), 7.0); fj = np.full((m + 3, 2), 9.0)
        fi[pi] = tv["xI"][st0[p]:st0[p + 1]]; fj[pj] = tv["xJ"][st0[p]:st0[p + 1]]
        feats += [fi, fj]; sizes += [tv["wh"][p][:2], tv["wh"][p][2:]]
        pairs.append((2 * p, 2 * p + 1)); ij.append(np.stack([pi, pj], 1))
    ij = np.concatenate(ij).astype(np.uint32)
    f = geofilter.GeometricFilter_FMatrix_AC(4.0, 2048)
    with _emu.emulated():
        mask_g, res_g, _ = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], f)
        mask_i, res_i, st = geofilter.filter_pairs_indexed(feats, np.array(sizes), np.array(pairs), tv["start"], ij, f)
        assert np.array_equal(mask_g, mask_i) and np.array_equal(res_g["ok"], res_i["ok"]) and np.array_equal(res_g["F"], res_i["F"])
        assert np.array_equal(res_g["nfa"], res_i["nfa"]) and int(st.n_pairs_ok) == int(res_g["ok"].sum())
        bad = ij.copy(); bad[0, 0] = len(feats[0])
        with pytest.raises(_capi.MvgxError) as e:
            geofilter.filter_pairs_indexed(feats, np.array(sizes), np.array(pairs), tv["start"], bad, f)
        assert e.value.code == _capi.MVGX_ERR_ARG
        badp = np.array(pairs); badp[1, 1] = len(feats)
        with pytest.raises(_capi.MvgxError) as e:
            geofilter.filter_pairs_indexed(feats, np.array(sizes), badp, tv["start"], ij, f)
        assert e.value.code == _capi.MVGX_ERR_ARG


def test_emulated_global_table_class_follows_the_restatement():
    """a pair with more than 12 000 correspondences (pool and log tables in global scratch like every pair above 256:
    geofilter_f_acransac_kernel<4, true>); few iterations: the emulation walks 12 001 residuals per model"""
    from openmvg_amd import synth
    tv = synth.two_view_matches_bulk(1, n=12001, seed=77, inlier_frac=(0.5, 0.7), no_geometry_frac=0.0)
    want = _oracle.port_geofilter(tv, max_iterations=12)
    with _emu.emulated():
        mask, res, st = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], geofilter.GeometricFilter_FMatrix_AC(4.0, 12))
    ref = dict(mask=want["mask"], ok=want["ok"], F=want["F"], precision=want["precision"], nfa=want["nfa"])
    differing, rep = gc.compare(tv["start"], ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"])
    assert bool(res["ok"][0]) == bool(want["ok"][0]) and not differing, (rep, differing)


def test_emulated_global_table_form_equals_the_lds_form():
    """MVGX_GEO_GLOBAL_ABOVE=8: pairs of more than eight correspondences keep their pool and log tables in global scratch (the form of every pair
    above 256 correspondences) - every output equal to the LDS form's on the same pairs"""
    import os
    tv = synth.two_view_matches(8, seed=21, n_max=60)
    fun = geofilter.GeometricFilter_FMatrix_AC(4.0, 512)
    out = {}
    saved = os.environ.get("MVGX_GEO_GLOBAL_ABOVE")
    try:
        for form in ("lds", "global"):
            if form == "global":
                os.environ["MVGX_GEO_GLOBAL_ABOVE"] = "8"
            else:
                os.environ.pop("MVGX_GEO_GLOBAL_ABOVE", None)
            with _emu.emulated():
                mask, res, st = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], fun)
            out[form] = (mask.copy(), res.copy(), int(st.n_iterations), int(st.n_models))
    finally:
        if saved is None:
            os.environ.pop("MVGX_GEO_GLOBAL_ABOVE", None)
        else:
            os.environ["MVGX_GEO_GLOBAL_ABOVE"] = saved
    assert out["lds"][2:] == out["global"][2:] and out["lds"][2] > 100
    assert np.array_equal(out["lds"][0], out["global"][0]) and out["lds"][1].tobytes() == out["global"][1].tobytes()


def test_adapter_specialisation_fills_the_container_like_the_reference_template():
    """ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_FMatrix_AC>: the same caller code linked once against
    the reference header's template and once against the explicit specialisation of openmvg_amd/adapter/mvgx_geometric_filter.cpp
    (device code under the HIP emulation): same pairs in the container, same match lists, with and without a distorting
    intrinsic (MatchesPairToMat undistorts the positions)"""
    from tests import _geofilter_scene
    ref_lib, emu_lib = _oracle.geofilter_container_lib("reference"), _oracle.geofilter_container_lib("adapter_emu")
    if ref_lib is None or emu_lib is None:
        pytest.skip("needs /root/reference (reference library and adapter harness)")
    feats, wh, putative = _geofilter_scene.collection(n_pairs=6, seed=9, n_min=40, n_max=70, inlier_frac=(0.6, 0.9), no_geometry_frac=0.2)
    for k1 in (0.0, 0.02):
        want = _oracle.geofilter_container("reference", feats, wh, putative, max_iterations=512, k1=k1)
        got = _oracle.geofilter_container("adapter_emu", feats, wh, putative, max_iterations=512, k1=k1)
        assert set(want) == set(got) and len(want) >= 2
        assert all(np.array_equal(want[k], got[k]) for k in want)


def test_adapter_specialisation_injected_device_failure_uses_the_reference_functor(monkeypatch, capfd):
    """error convention (openmvg_amd/adapter/mvgx_adapter_policy.hpp): a failing device call is logged once and the pairs run through
    the reference's own functor.Robust_estimation inside the replacement - same container as the reference template, no exception"""
    import ctypes as C
    if _oracle.geofilter_container_lib("adapter_emu") is None or _oracle.geofilter_container_lib("reference") is None:
        pytest.skip("adapter harness / reference library not present")
    from tests._geofilter_scene import collection
    feats, wh, putative = collection(n_pairs=8, seed=5, n_max=120)
    want = _oracle.geofilter_container("reference", feats, wh, putative)
    lib = _oracle.geofilter_container_lib("adapter_emu")
    out = (C.c_uint64 * 3)()
    lib.mvgx_adapter_counters(out, 1)
    monkeypatch.setenv("MVGX_ADAPTER_INJECT_FAILURE", "geofilter:run")
    got = _oracle.geofilter_container("adapter_emu", feats, wh, putative)
    monkeypatch.delenv("MVGX_ADAPTER_INJECT_FAILURE")
    lib.mvgx_adapter_counters(out, 1)
    assert (int(out[0]), int(out[1]), int(out[2])) == (0, len(putative), 1)
    assert got.keys() == want.keys() and all(np.array_equal(got[k], want[k]) for k in want)
    assert "continuing with the reference's own CPU code" in capfd.readouterr().err
