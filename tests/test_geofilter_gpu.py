"""Geometric filter on the MI355X: the device path through the C ABI against the compiled reference
(oracle/_ref/libref_geofilter.so travels with the snapshot) and against the stored reference outputs, per the parity policy of
tests/_geofilter_cases.py."""
import numpy as np
import pytest

from openmvg_amd import _capi, geofilter, synth
from tests import _geofilter_cases as gc, _oracle
from tests.test_geofilter_cpu import GOLD, _gold_tv

pytestmark = pytest.mark.gpu


def test_golden_fixture_inlier_sets():
    tv, ref = _gold_tv()
    mask, res, st = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], geofilter.GeometricFilter_FMatrix_AC(float(GOLD["precision_px"]), int(GOLD["max_iterations"])))
    differing, rep = gc.compare(tv["start"], ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"])
    assert rep["pairs_ok_reference"] > 100 and len(differing) <= gc.allowed_differing(rep["pairs"]), (rep, differing)
    assert int(st.n_pairs) == rep["pairs"] and st.kernel_ms > 0


@pytest.mark.parametrize("kw,iters", [(dict(seed=5, n_max=400), 2048), (dict(seed=6, n_max=120, inlier_frac=(0.15, 0.5)), 1024),
                                      (dict(seed=7, n_max=200), 37), (dict(seed=8, n_min=1100, n_max=1300, tiny_frac=0.0), 2048)])
def test_against_the_compiled_reference(kw, iters):
    """mixed sizes (three LDS classes at n > 1024), low inlier ratios, an iteration budget that ends inside the warm-up"""
    if not _oracle.have_ref_geofilter():
        pytest.skip("oracle/_ref/libref_geofilter.so not built")
    n_pairs = 60 if kw.get("n_min", 0) > 1000 else 1500
    tv = synth.two_view_matches(n_pairs, **kw)
    ref = _oracle.ref_geofilter(tv, 4.0, iters)
    mask, res, _ = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], geofilter.GeometricFilter_FMatrix_AC(4.0, iters))
    differing, rep = gc.compare(tv["start"], ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"])
    assert len(differing) <= gc.allowed_differing(rep["pairs"]), (rep, differing[:10])


def test_edge_cases_and_errors():
    rng = np.random.default_rng(3)
    # pairs with 0, 7, 8 correspondences; identical points; all correspondences on one line
    blocks = [np.zeros((0, 2)), rng.uniform(0, 900, (7, 2)), rng.uniform(0, 900, (8, 2)), np.tile([[100.0, 200.0]], (30, 1)),
              np.stack([np.linspace(0, 900, 40), np.linspace(0, 900, 40)], 1)]
    xI = np.concatenate(blocks); xJ = xI + rng.normal(0, 0.3, xI.shape)
    start = np.cumsum([0] + [len(b) for b in blocks]).astype(np.uint64)
    wh = np.tile(np.array([1000, 1000, 1000, 1000], np.uint32), (len(blocks), 1))
    mask, res, _ = geofilter.filter_pairs(xI, xJ, start, wh)
    assert not res["ok"][0] and not res["ok"][1] and np.array_equal(res["F"][0], np.eye(3)) and not mask[:7].any()
    if _oracle.have_ref_geofilter():
        ref = _oracle.ref_geofilter(dict(xI=xI, xJ=xJ, start=start, wh=wh))
        assert np.array_equal(ref["ok"][:3], res["ok"][:3])
    with pytest.raises(_capi.MvgxError) as e:
        geofilter.filter_pairs(xI, xJ, start, wh, geofilter.GeometricFilter_FMatrix_AC(float("inf"), 64))
    assert e.value.code == _capi.MVGX_ERR_UNSUPPORTED
    big = np.zeros(((1 << 20) + 1, 2))
    with pytest.raises(_capi.MvgxError):
        geofilter.filter_pairs(big, big, np.array([0, (1 << 20) + 1], np.uint64), wh[:1])


def test_pairs_beyond_the_lds_classes_against_the_compiled_reference():
    """more than 12 000 correspondences in a pair: the wave's sampling pool and log tables live in global scratch
    (geofilter_f_acransac_kernel<1, true>), mixed with pairs of the LDS classes in one call"""
    if not _oracle.have_ref_geofilter():
        pytest.skip("oracle/_ref/libref_geofilter.so not built")
    tvs = [synth.two_view_matches_bulk(3, n=12001, seed=41, no_geometry_frac=0.0), synth.two_view_matches_bulk(2, n=20000, seed=42, inlier_frac=(0.4, 0.6), no_geometry_frac=0.0),
           synth.two_view_matches_bulk(4, n=300, seed=43)]
    xI = np.concatenate([t["xI"] for t in tvs]); xJ = np.concatenate([t["xJ"] for t in tvs])
    counts = np.concatenate([np.diff(t["start"].astype(np.int64)) for t in tvs])
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    wh = np.concatenate([t["wh"] for t in tvs])
    tv = dict(xI=xI, xJ=xJ, start=start, wh=wh)
    ref = _oracle.ref_geofilter(tv, 4.0, 256)
    mask, res, st = geofilter.filter_pairs(xI, xJ, start, wh, geofilter.GeometricFilter_FMatrix_AC(4.0, 256))
    differing, rep = gc.compare(start, ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"])
    assert res["ok"][:5].all() and not differing, (rep, differing)


@pytest.mark.parametrize("model", ["f", "h", "e"])
def test_global_table_form_equals_the_lds_form(model):
    """MVGX_GEO_GLOBAL_ABOVE=8 puts every pair of more than eight correspondences into the global-table form (the form of every pair above
    256): every output equal to the LDS form's, for the three models, samples ahead included"""
    import os
    if model == "h":
        tv = synth.two_view_homography_matches(300, seed=31, n_min=8, n_max=250, tiny_frac=0.0)
        run = lambda: geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], geofilter.GeometricFilter_HMatrix_AC(4.0, 1024))   # noqa: E731
    elif model == "e":
        tv = synth.two_view_matches(300, seed=32, n_max=250)
        K = synth.two_view_calibration(tv)
        run = lambda: geofilter.filter_pairs_e(tv["xI"], tv["xJ"], tv["start"], tv["wh"], K, geofilter.GeometricFilter_EMatrix_AC(4.0, 1024))   # noqa: E731
    else:
        tv = synth.two_view_matches(300, seed=33, n_max=250)
        run = lambda: geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], geofilter.GeometricFilter_FMatrix_AC(4.0, 1024))   # noqa: E731
    out = {}
    saved = os.environ.get("MVGX_GEO_GLOBAL_ABOVE")
    try:
        for form in ("lds", "global"):
            if form == "global":
                os.environ["MVGX_GEO_GLOBAL_ABOVE"] = "8"
            else:
                os.environ.pop("MVGX_GEO_GLOBAL_ABOVE", None)
            mask, res, st = run()
            out[form] = (mask.copy(), res.copy(), int(st.n_iterations), int(st.n_models))
    finally:
        if saved is None:
            os.environ.pop("MVGX_GEO_GLOBAL_ABOVE", None)
        else:
            os.environ["MVGX_GEO_GLOBAL_ABOVE"] = saved
    assert out["lds"][2:] == out["global"][2:] and out["lds"][2] > 10000
    assert np.array_equal(out["lds"][0], out["global"][0]) and out["lds"][1].tobytes() == out["global"][1].tobytes()


def test_container_form():
    tv = synth.two_view_matches(30, seed=11, n_max=200, tiny_frac=0.0)
    start = tv["start"].astype(np.int64)
    feats, sizes, putative = [], [], {}
    for p in range(30):   # images 2 p and 2 p + 1 with the pair's features in shuffled order
        n = int(start[p + 1] - start[p])
        perm_i, perm_j = np.random.default_rng(p).permutation(n), np.random.default_rng(100 + p).permutation(n)
        fi = np.zeros((n, 2)); fj = np.zeros((n, 2))
        fi[perm_i] = tv["xI"][start[p]:start[p + 1]]; fj[perm_j] = tv["xJ"][start[p]:start[p + 1]]
        feats += [fi, fj]; sizes += [tuple(tv["wh"][p][:2]), tuple(tv["wh"][p][2:])]
        putative[(2 * p, 2 * p + 1)] = np.stack([perm_i, perm_j], 1).astype(np.uint32)
    geo = geofilter.Robust_model_estimation(putative, feats, sizes)
    mask, res, _ = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"])
    assert set(geo) == {(2 * p, 2 * p + 1) for p in range(30) if res["ok"][p]}
    for p in range(30):
        if res["ok"][p]:
            assert len(geo[(2 * p, 2 * p + 1)]) == int(mask[start[p]:start[p + 1]].sum())


@pytest.mark.parametrize("guided", [False, True])
def test_adapter_specialisation_against_the_reference_template(guided):
    """the drop-in (openmvg_amd/adapter/mvgx_geometric_filter.{hpp,cpp}) on the device against the reference's member template
    through identical caller code; guided matching runs the reference's own Geometry_guided_matching with the device's F"""
    from tests import _geofilter_scene
    ref_lib, dev_lib = _oracle.geofilter_container_lib("reference"), _oracle.geofilter_container_lib("adapter")
    if ref_lib is None or dev_lib is None:
        pytest.skip("reference library / adapter harness not built")
    feats, wh, putative = _geofilter_scene.collection(n_pairs=200, seed=21, n_max=250)
    descs = [np.random.default_rng(7 + k).integers(0, 256, (len(f), 128), dtype=np.uint8) for k, f in enumerate(feats)] if guided else None
    want = _oracle.geofilter_container("reference", feats, wh, putative, guided=guided, ratio=0.8, descs=descs)
    got = _oracle.geofilter_container("adapter", feats, wh, putative, guided=guided, ratio=0.8, descs=descs)
    differing = [k for k in set(want) | set(got) if k not in want or k not in got or not np.array_equal(want[k], got[k])]
    from tests import _geofilter_cases as gc
    assert len(want) > 100 and len(differing) <= gc.allowed_differing(len(want), "f"), (len(want), len(got), differing[:5])
