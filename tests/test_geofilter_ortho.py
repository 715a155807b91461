"""The orthographic essential model of the geometric filter (main_GeometricFilter -g o): GeometricFilter_EOMatrix_RA =
ACKernelAdaptorEssentialOrtho<ThreePointKernel, OrthographicSymmetricEpipolarDistanceError> + ACRANSAC (Eo_Robust.hpp:35-165).
The solver is closed form (+ - x / sqrt) and the device evaluates it in the reference's order without contraction, so on the SAME
inputs the models, NFA values, bounds and inlier sets are the reference's bit for bit - asserted here on every pair, no tolerance.
Checker: the compiled reference (oracle/_ref/libref_geofilter.so :: ref_geofilter_eo_acransac) live and through the stored fixture
tests/golden/geofilter_ortho.npz (make_geofilter_ortho_golden.py; the inputs are the reference cameras' own bearing vectors)."""
import ctypes as C
import os

import numpy as np
import pytest

from openmvg_amd import _capi, geofilter, synth
from tests import _emu, _oracle

GOLD_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geofilter_ortho.npz")
FUNCTOR = geofilter.GeometricFilter_EOMatrix_RA


def golden_scene():
    tv = synth.two_view_matches(150, seed=23, n_max=160)
    return tv, synth.two_view_calibration(tv)


def reference_inputs(tv, K, precision):
    """hnormalized bearing vectors of the reference's cameras (Pinhole_Intrinsic(w, h, K)(x).colwise().hnormalized()) and the per-pair bound"""
    bI, bJ = _oracle.ref_pinhole_bearings(tv, K)
    hI = np.ascontiguousarray(bI[:, :2] / bI[:, 2:3]); hJ = np.ascontiguousarray(bJ[:, :2] / bJ[:, 2:3])
    _, _, prec = geofilter.ortho_inputs(tv["xI"], tv["xJ"], tv["start"], np.asarray(K, np.float64).reshape(-1, 2, 3, 3), precision)
    return hI, hJ, prec


def run(hI, hJ, start, wh, prec, precision=2.0, iterations=1024):
    start = np.ascontiguousarray(start, np.uint64); wh = np.ascontiguousarray(wh, np.uint32)
    hI = np.ascontiguousarray(hI, np.float64); hJ = np.ascontiguousarray(hJ, np.float64); prec = np.ascontiguousarray(prec, np.float64)
    n_pairs = len(start) - 1
    mask = np.zeros(max(len(hI), 1), np.uint8)
    res = (_capi.GeofilterResult * max(n_pairs, 1))()
    st = _capi.GeofilterStats()
    opt = _capi.GeofilterOptions(precision, iterations)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    _capi.check(_capi.lib().mvgx_geofilter_eo_acransac(-1, P(hI), P(hJ), P(start), P(wh), P(prec), n_pairs, C.byref(opt), P(mask), C.cast(res, C.c_void_p),
                                                       C.byref(st)))
    return mask[:len(hI)].astype(bool), geofilter._results_array(res, n_pairs), st


def assert_identical(ref, mask, res):
    assert np.array_equal(mask, ref["mask"]) and np.array_equal(res["ok"].astype(bool), ref["ok"])
    ok = ref["ok"]
    assert np.array_equal(res["F"][ok], ref["F"][ok]) and np.array_equal(res["nfa"][ok], ref["nfa"][ok])
    assert np.array_equal(res["precision_robust"][ok], ref["precision"][ok])


def _gold(sel=None):
    g = np.load(GOLD_PATH)
    start = g["start"].astype(np.int64)
    pairs = list(range(len(start) - 1)) if sel is None else list(sel)
    cut = lambda a: np.concatenate([a[start[p]:start[p + 1]] for p in pairs])   # noqa: E731
    new_start = np.concatenate([[0], np.cumsum([start[p + 1] - start[p] for p in pairs])]).astype(np.uint64)
    ref = dict(mask=cut(g["mask"]), ok=g["ok"][pairs], F=g["F"][pairs], precision=g["precision"][pairs], nfa=g["nfa"][pairs])
    return cut(g["hI"]), cut(g["hJ"]), new_start, g["wh"][pairs], g["prec"][pairs], ref


def test_golden_fixture_is_the_reference():
    if not _oracle.have_ref_geofilter():
        pytest.skip("oracle/_ref/libref_geofilter.so not built (needs /root/reference)")
    tv, K = golden_scene()
    hI, hJ, start, wh, prec, ref = _gold()
    lI, lJ, lprec = reference_inputs(tv, K, 2.0)
    assert np.array_equal(hI, lI) and np.array_equal(hJ, lJ) and np.array_equal(prec, lprec) and np.array_equal(start, tv["start"].astype(np.uint64))
    live = _oracle.ref_geofilter_eo(tv, K, precision=2.0, max_iterations=1024)
    assert int(ref["ok"].sum()) > 80
    assert np.array_equal(live["mask"], ref["mask"]) and np.array_equal(live["ok"], ref["ok"]) and np.array_equal(live["F"], ref["F"])


def test_restatement_is_bit_identical_to_the_stored_reference_outputs():
    """oracle/geofilter_oracle.cpp (plain C++, -ffp-contract=off) on every stored pair: closed form, so no tolerance"""
    hI, hJ, start, wh, prec, ref = _gold()
    got = _oracle.port_geofilter_ortho(hI, hJ, start, wh, prec)
    res = dict(F=got["F"], ok=got["ok"], nfa=got["nfa"], precision_robust=got["precision"])
    assert_identical(ref, got["mask"], res)


def test_emulated_device_code_is_bit_identical_to_the_stored_reference_outputs():
    g = np.load(GOLD_PATH)
    n = np.diff(g["start"].astype(np.int64))
    small = [int(p) for p in np.argsort(n) if 10 < n[p] <= 70]
    sel = [p for p in small if g["ok"][p]][:3] + [p for p in small if not g["ok"][p]][:2] + [int(np.argmin(n))]
    hI, hJ, start, wh, prec, ref = _gold(sel)
    with _emu.emulated():
        mask, res, st = run(hI, hJ, start, wh, prec)
    assert_identical(ref, mask, res)
    assert int(st.n_models) == 2 * int(st.n_iterations) > 0   # (two models per sample)


def test_host_mirror_and_argument_errors_under_emulation():
    tv, K = golden_scene()
    hI, hJ, prec = reference_inputs(tv, K, 2.0) if _oracle.have_ref_geofilter() else (None, None, None)
    mI, mJ, mprec = geofilter.ortho_inputs(tv["xI"], tv["xJ"], tv["start"], np.asarray(K).reshape(-1, 2, 3, 3), 2.0)
    if hI is not None:   # numpy's bearings equal the reference cameras' to rounding
        assert np.abs(mI - hI).max() < 1e-13 and np.abs(mJ - hJ).max() < 1e-13
    with _emu.emulated():
        x = np.zeros((3, 2)); start = np.array([0, 3], np.uint64); wh = np.array([[100, 100, 100, 100]], np.uint32)
        Kp = np.tile(np.array([[90.0, 0, 50], [0, 90.0, 50], [0, 0, 1]]), (1, 2, 1, 1))
        mask, res, st = geofilter.filter_pairs_ortho(x, x, start, wh, Kp)   # not more correspondences than a sample: no estimation
        assert not mask.any() and not res["ok"][0] and np.array_equal(res["F"][0], np.eye(3))
        with pytest.raises(ValueError):
            geofilter.filter_pairs_ortho(x, x, start, wh, Kp[:0])
        with pytest.raises(_capi.MvgxError) as e:   # an unbounded precision is not reproduced on the device
            run(x, x, start, wh, np.array([np.inf]))
        assert e.value.code == _capi.MVGX_ERR_UNSUPPORTED


# ---- the drop-in: ImageCollectionGeometricFilter::Robust_model_estimation<GeometricFilter_EOMatrix_RA> ----
def _container_case(kind, guided=False):
    from tests import _geofilter_scene
    feats, wh, putative = _geofilter_scene.collection(n_pairs=5, seed=12, n_min=40, n_max=70, inlier_frac=(0.6, 0.9), no_geometry_frac=0.2, size=(1000, 1000))
    return _oracle.geofilter_container(kind, feats, wh, putative, precision=2.0, max_iterations=512, guided=guided, model="eo", focal=900.0)


@pytest.mark.parametrize("guided", [False, True])
def test_adapter_specialisation_fills_the_container_like_the_reference_template(guided):
    """(guided: the functor's Geometry_guided_matching returns no matches and the caller swaps them in - empty lists in both)"""
    ref_lib, lib = _oracle.geofilter_container_lib("reference"), _oracle.geofilter_container_lib("adapter_emu")
    if ref_lib is None or lib is None or not hasattr(ref_lib, "ref_geofilter_container_eo"):
        pytest.skip("needs /root/reference (reference library and adapter harness)")
    want, got = _container_case("reference", guided), _container_case("adapter_emu", guided)
    assert set(want) == set(got) and len(want) >= 2 and (8, 9) not in want
    assert all(np.array_equal(want[k], got[k]) for k in want)
    assert guided == all(len(v) == 0 for v in want.values())


# ---- MI355X ----
@pytest.mark.gpu
def test_device_is_bit_identical_to_the_stored_reference_outputs():
    hI, hJ, start, wh, prec, ref = _gold()
    mask, res, st = run(hI, hJ, start, wh, prec)
    assert int(ref["ok"].sum()) > 80
    assert_identical(ref, mask, res)


@pytest.mark.gpu
def test_device_is_bit_identical_to_the_compiled_reference_on_mixed_sizes():
    if not _oracle.have_ref_geofilter():
        pytest.skip("oracle/_ref/libref_geofilter.so not present")
    tv = synth.two_view_matches(600, seed=19, n_min=3, n_max=1500)
    K = synth.two_view_calibration(tv)
    for precision, iters in ((2.0, 1024), (4.0, 30)):
        hI, hJ, prec = reference_inputs(tv, K, precision)
        ref = _oracle.ref_geofilter_eo(tv, K, precision=precision, max_iterations=iters)
        mask, res, st = run(hI, hJ, tv["start"], tv["wh"], prec, precision, iters)
        assert_identical(ref, mask, res)


@pytest.mark.gpu
@pytest.mark.parametrize("guided", [False, True])
def test_adapter_specialisation_on_the_device(guided):
    ref_lib, lib = _oracle.geofilter_container_lib("reference"), _oracle.geofilter_container_lib("adapter")
    if ref_lib is None or lib is None or not hasattr(ref_lib, "ref_geofilter_container_eo"):
        pytest.skip("adapter harness / reference library not present")
    want, got = _container_case("reference", guided), _container_case("adapter", guided)
    assert set(want) == set(got) and len(want) >= 2
    assert all(np.array_equal(want[k], got[k]) for k in want)
