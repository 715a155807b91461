"""Real descriptors (BASELINE configs[0] substitute): SIFT regions of the two SceauxCastle JPGs that ship with openMVG,
extracted by the reference's own SIFT_Anatomy_Image_describer and matched by the reference's own Matcher_Regions
(tests/golden/make_sceaux_golden.py -> tests/golden/sceaux_sift.npz, "data": "real"). CPU: the restatement reproduces the stored
lists; GPU: the device path does, through the C ABI and through the Matcher_Regions mirror."""
import os

import numpy as np
import pytest

from openmvg_amd import matching
from tests import _oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sceaux_sift.npz")
PAIRS = np.array([[0, 1], [1, 0]], np.uint32)


def _load():
    z = np.load(GOLDEN)
    descs = [z["desc0"], z["desc1"]]
    want = {r: {(0, 1): z[f"matches_r{r}_0_1"], (1, 0): z[f"matches_r{r}_1_0"]} for r in (80, 60)}
    return z, descs, want


def test_fixture_is_real_sift():
    z, descs, want = _load()
    assert list(z["images"]) == ["100_7101.jpg", "100_7102.jpg"] and tuple(z["size0"]) == (1416, 1064)
    assert descs[0].shape == (2783, 128) and descs[1].shape == (2557, 128) and descs[0].dtype == np.uint8
    # RootSIFT quantisation of the reference (sift_DescriptorExtractor.hpp:484-494): sum of squares ~ 512^2
    n2 = (descs[0].astype(np.int64) ** 2).sum(axis=1)
    assert 0.9 * 512 ** 2 < np.median(n2) < 1.1 * 512 ** 2
    assert len(want[80][(0, 1)]) == 1378 and len(want[60][(1, 0)]) == 1046
    f = z["feat0"]
    assert f.shape == (2783, 4) and (f[:, 0] >= 0).all() and (f[:, 0] <= 1416).all() and (f[:, 1] <= 1064).all()


@pytest.mark.parametrize("ratio", [0.8, 0.6])
def test_restatement_reproduces_the_reference_on_real_descriptors(ratio):
    _, descs, want = _load()
    off, ij = _oracle.port_matcher_regions_match(descs, PAIRS, ratio)
    got = _oracle.offsets_to_dict(PAIRS, off, ij)
    w = want[int(round(ratio * 100))]
    assert got.keys() == w.keys() and all(np.array_equal(got[k], w[k]) for k in w)


@pytest.mark.gpu
@pytest.mark.parametrize("ratio", [0.8, 0.6])
@pytest.mark.parametrize("variant", [1, 43])
def test_device_path_reproduces_the_reference_on_real_descriptors(ratio, variant):
    from tests.test_matching_gpu import run_hip
    _, descs, want = _load()
    st, off, ij = run_hip(descs, PAIRS, ratio, variant)
    got = _oracle.offsets_to_dict(PAIRS, off, ij)
    w = want[int(round(ratio * 100))]
    assert got.keys() == w.keys() and all(np.array_equal(got[k], w[k]) for k in w)
    assert int(st.n_desc_pairs) == 2 * 2783 * 2557


@pytest.mark.gpu
def test_matcher_regions_mirror_on_real_descriptors():
    _, descs, want = _load()
    prov = matching.Regions_Provider({k: matching.Regions(d) for k, d in enumerate(descs)})
    out = matching.PairWiseMatches()
    matching.Matcher_Regions(0.8, matching.EMatcherType.BRUTE_FORCE_L2).Match(prov, [(0, 1)], out)
    assert list(out) == [(0, 1)] and np.array_equal(out[(0, 1)], want[80][(0, 1)])


def test_emulated_device_code_on_real_descriptors():
    """the matching kernels under the HIP emulation (tests/_emu.py), default variant, on the real regions"""
    from tests import _emu
    _, descs, want = _load()
    with _emu.emulated():
        ctx = matching.MatchContext(0)
        ctx.set_regions(descs)
        _, off, ij = ctx.run(PAIRS[:1], np.float32(0.8) * np.float32(0.8))
        ctx.close()
    assert np.array_equal(ij, want[80][(0, 1)])


# ---- the step after matching on the same real pair (the geometric filter had only ever seen synthetic matches) ----
GEO_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sceaux_geofilter.npz")


def _geo_case(model, sel=None):
    """(two-view arrays, stored reference outputs) of tests/golden/make_sceaux_geofilter_golden.py: the reference's putative lists
    between the two SceauxCastle images (both directions, ratio 0.8 / 0.6), filtered by the compiled reference (4 px, 2048 iterations)"""
    from tests.golden.make_sceaux_geofilter_golden import two_view
    z = np.load(GOLDEN); g = np.load(GEO_GOLDEN)
    tv = two_view(z)
    ref = dict(mask=g[f"{model}_mask"], ok=g[f"{model}_ok"], F=g[f"{model}_F"], precision=g[f"{model}_precision"], nfa=g[f"{model}_nfa"])
    if sel is not None:
        st = tv["start"].astype(np.int64)
        cut = lambda a: np.concatenate([a[st[p]:st[p + 1]] for p in sel])   # noqa: E731
        tv = dict(xI=cut(tv["xI"]), xJ=cut(tv["xJ"]), wh=tv["wh"][sel],
                  start=np.concatenate([[0], np.cumsum([st[p + 1] - st[p] for p in sel])]).astype(np.uint64))
        ref = dict(mask=cut(ref["mask"]), ok=ref["ok"][sel], F=ref["F"][sel], precision=ref["precision"][sel], nfa=ref["nfa"][sel])
    return tv, ref, g


def test_geofilter_fixture_is_the_reference_on_real_matches():
    tv, ref, g = _geo_case("f")
    assert list(np.diff(tv["start"].astype(np.int64))) == [1378, 1421, 1038, 1046] and ref["ok"].all()
    st = tv["start"].astype(np.int64)
    assert [int(ref["mask"][st[p]:st[p + 1]].sum()) for p in range(4)] == [1143, 1131, 946, 925]
    _, refh, _ = _geo_case("h")   # a homography explains the castle's facade only: fewer inliers than the epipolar model
    assert [int(refh["mask"][st[p]:st[p + 1]].sum()) for p in range(4)] == [527, 500, 448, 447]
    # the reference's own spread on this pair: its -mavx2 -mfma build ends with the same inlier sets
    assert g["f_same_inlier_set_in_the_avx2_fma_build"].all() and g["h_same_inlier_set_in_the_avx2_fma_build"].all()
    assert len(g["container_f_r80"]) == 1143 and len(g["container_h_r80"]) == 527


@pytest.mark.parametrize("model", ["f", "h"])
def test_geofilter_restatement_on_real_matches(model):
    from tests import _geofilter_cases as gc
    tv, ref, _ = _geo_case(model)
    got = (_oracle.port_geofilter if model == "f" else _oracle.port_geofilter_h)(tv)
    differing, rep = gc.compare(tv["start"], ref, got["mask"], got["ok"], got["F"], got["precision"], got["nfa"])
    assert not differing, rep


@pytest.mark.parametrize("model,sel", [("f", [2]), ("h", [0, 3])])
def test_geofilter_emulated_device_code_on_real_matches(model, sel):
    """the kernel of mvgx_geofilter.hip under the HIP emulation on real putative matches: identical inlier sets, NFA, precision, model"""
    from openmvg_amd import geofilter
    from tests import _emu, _geofilter_cases as gc
    tv, ref, _ = _geo_case(model, sel)
    fun = (geofilter.GeometricFilter_FMatrix_AC if model == "f" else geofilter.GeometricFilter_HMatrix_AC)(4.0, 2048)
    with _emu.emulated():
        mask, res, st = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], fun)
    differing, rep = gc.compare(tv["start"], ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"])
    assert not differing, rep
    assert int(st.n_iterations) > 0 and int(st.n_models) >= int(st.n_iterations)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["f", "h"])
def test_geofilter_device_on_real_matches(model):
    """reference SIFT -> (stored) reference lists == device lists (tests above) -> device F and H filter against the compiled
    reference's stored outputs: no pair may differ"""
    from openmvg_amd import geofilter
    from tests import _geofilter_cases as gc
    tv, ref, _ = _geo_case(model)
    fun = (geofilter.GeometricFilter_FMatrix_AC if model == "f" else geofilter.GeometricFilter_HMatrix_AC)(4.0, 2048)
    mask, res, st = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], fun)
    differing, rep = gc.compare(tv["start"], ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"])
    assert not differing, rep
    assert int(st.n_pairs_ok) == 4 and int(st.n_iterations) > 0 and int(st.wave_clocks) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["f", "h"])
def test_real_pair_end_to_end_match_then_filter_on_the_device(model):
    """device Match (Matcher_Regions mirror) -> device Robust_model_estimation (container form, positions gathered on the device) ==
    the container the reference's ImageCollectionGeometricFilter template produced from the reference's own lists"""
    from openmvg_amd import geofilter
    z = np.load(GOLDEN); g = np.load(GEO_GOLDEN)
    descs = [z["desc0"], z["desc1"]]
    prov = matching.Regions_Provider({k: matching.Regions(d) for k, d in enumerate(descs)})
    putative = matching.PairWiseMatches()
    matching.Matcher_Regions(0.8, matching.EMatcherType.BRUTE_FORCE_L2).Match(prov, [(0, 1)], putative)
    fun = (geofilter.GeometricFilter_FMatrix_AC if model == "f" else geofilter.GeometricFilter_HMatrix_AC)(4.0, 2048)
    feats = [z["feat0"][:, :2].astype(np.float64), z["feat1"][:, :2].astype(np.float64)]
    out = geofilter.Robust_model_estimation({k: putative[k] for k in putative}, feats, [tuple(z["size0"]), tuple(z["size1"])], fun)
    assert list(out) == [(0, 1)] and np.array_equal(out[(0, 1)], g[f"container_{model}_r80"])
