"""Guided matching - the second stage of the a-contrario geometric filters (VERDICT r4 "what's missing" 2).

  reference:   robust_estimation/guided_matching.hpp:178-227 through {F,H,E}_ACRobust.hpp's Geometry_guided_matching
  restatement: oracle/geofilter_oracle.cpp::port_guided_match, pinned here to the reference's own template (ref_guided_match)
  device:      openmvg_amd/csrc/mvgx_guided.hip (mvgx_guided_match_u8) - under the HIP emulation on the CPU, on the MI355X with -m gpu

Integer descriptor distances are exact; the geometric test is rounded operation by operation as the reference's build rounds it, so the
lists are compared entry by entry, no tolerance."""
import contextlib

import numpy as np
import pytest

from openmvg_amd import geofilter
from tests import _emu, _oracle


def _pair(rng, n_true, n_clutter_i, n_clutter_j, kind, wh=(3000, 2000), noise=1.0, desc_bytes=128, dup=0):
    """two feature sets related by a fundamental matrix (kind 0) / a homography (kind 1) + clutter; float-exact positions"""
    w, h = wh
    xi = np.stack([rng.uniform(0, w, n_true), rng.uniform(0, h, n_true)], 1)
    if kind == 1:
        H = np.array([[1.02, 0.03, 40.0], [-0.02, 0.98, -25.0], [1e-5, -2e-5, 1.0]])
        p = np.c_[xi, np.ones(n_true)] @ H.T
        xj = p[:, :2] / p[:, 2:3]
        M = H
    else:
        # a rectified-like geometry with a rotation: F from random cameras
        R = np.array([[0.995, -0.02, 0.09], [0.021, 0.9997, -0.01], [-0.0898, 0.012, 0.9959]])
        t = np.array([1.0, 0.1, 0.2])
        K = np.array([[2400.0, 0, w / 2], [0, 2400.0, h / 2], [0, 0, 1]])
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        M = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
        M = M / np.abs(M).max()
        depth = rng.uniform(4, 20, n_true)
        X = (np.linalg.inv(K) @ np.c_[xi, np.ones(n_true)].T) * depth
        p = (K @ (R @ X + t[:, None])).T
        xj = p[:, :2] / p[:, 2:3]
    xj = xj + rng.normal(0, noise, xj.shape)
    xi = np.r_[xi, np.stack([rng.uniform(0, w, n_clutter_i), rng.uniform(0, h, n_clutter_i)], 1)]
    xj = np.r_[xj, np.stack([rng.uniform(0, w, n_clutter_j), rng.uniform(0, h, n_clutter_j)], 1)]
    xi = xi.astype(np.float32).astype(np.float64); xj = xj.astype(np.float32).astype(np.float64)   # what a Regions object holds
    di = rng.integers(0, 256, (len(xi), desc_bytes), dtype=np.uint8)
    dj = rng.integers(0, 256, (len(xj), desc_bytes), dtype=np.uint8)
    dj[:n_true] = np.clip(di[:n_true].astype(int) + rng.integers(-12, 13, (n_true, desc_bytes)), 0, 255).astype(np.uint8)   # true matches look alike
    for k in range(dup):   # a right feature repeated (same position, same descriptor): best == second best -> the left feature is dropped
        xj = np.r_[xj, xj[k:k + 1]]; dj = np.r_[dj, dj[k:k + 1]]
    pi, pj = rng.permutation(len(xi)), rng.permutation(len(xj))
    return np.ascontiguousarray(xi[pi]), np.ascontiguousarray(di[pi]), np.ascontiguousarray(xj[pj]), np.ascontiguousarray(dj[pj]), M


@pytest.mark.skipif(not _oracle.have_ref_geofilter(), reason="oracle/_ref not built")
@pytest.mark.parametrize("kind", [0, 1])
def test_restatement_equals_the_reference_template(kind):
    rng = np.random.default_rng(31 + kind)
    total = 0
    for trial in range(6):
        xi, di, xj, dj, M = _pair(rng, 120, 150, 170, kind, dup=3 if trial % 2 else 0)
        for prec, ratio in ((4.0, 0.8), (1.5, 0.6), (30.0, 1.0)):
            want = _oracle.ref_guided_match(kind, M, xi, di, xj, dj, prec * prec, ratio * ratio)
            got = _oracle.port_guided_match(kind, M, xi, di, xj, dj, prec * prec, ratio * ratio)
            assert np.array_equal(want, got), (trial, prec, ratio, len(want), len(got))
            total += len(want)
    # (the scenes do produce guided matches; under a homography only the left features with TWO right features inside the bound can pass -
    # distanceRatio needs a second best - so that list is short by construction)
    assert total > (500 if kind == 0 else 50)


def _device_equals_restatement(emulated, n_pairs, sizes, seed):
    rng = np.random.default_rng(seed)
    for kind in (0, 1):
        feats, descs, pairs, models, precs = [], [], [], [], []
        for p in range(n_pairs):
            xi, di, xj, dj, M = _pair(rng, *sizes, kind, dup=2 if p % 3 == 0 else 0)
            feats += [xi, xj]; descs += [di, dj]
            pairs.append((2 * p, 2 * p + 1)); models.append(M)
            precs.append([4.0, 1.2, np.inf, 25.0][p % 4])
        # a pair with an empty left image, one with an empty right image, one that reuses images in the other direction
        feats += [np.zeros((0, 2)), feats[0]]; descs += [np.zeros((0, 128), np.uint8), descs[0]]
        e = len(feats) - 2
        pairs += [(e, 1), (0, e), (1, 0)]; models += [models[0], models[0], models[0].T if kind == 0 else np.linalg.inv(models[0])]; precs += [4.0, 4.0, 4.0]
        ratio = 0.8
        with (_emu.emulated() if emulated else contextlib.nullcontext()):
            got, st = geofilter.guided_matching(feats, descs, pairs, models, precs, ratio, kind)
        n = 0
        for p, (I, J) in enumerate(pairs):
            th = precs[p] * precs[p]
            want = _oracle.port_guided_match(kind, models[p], feats[I], descs[I], feats[J], descs[J], th, ratio * ratio) if np.isfinite(th) else np.zeros((0, 2), np.uint32)
            have = got.get((I, J), np.zeros((0, 2), np.uint32))
            assert np.array_equal(want, have), (kind, p, len(want), len(have))
            n += len(want)
        assert n > (20 * n_pairs / 4 if kind == 0 else 0) and st.n_matches == n and st.n_pairs == len(pairs)
        assert st.n_geometric_passed > 0 and st.n_geometric_tests > st.n_geometric_passed


def test_device_code_equals_the_restatement_emulated():
    _device_equals_restatement(True, 5, (40, 50, 45), 5)


@pytest.mark.gpu
def test_device_equals_the_restatement_on_the_mi355x():
    _device_equals_restatement(False, 24, (400, 900, 1100), 7)


@pytest.mark.gpu
@pytest.mark.skipif(not _oracle.have_ref_geofilter(), reason="oracle/_ref not built")
@pytest.mark.parametrize("kind", [0, 1])
def test_device_equals_the_reference_template_on_the_mi355x(kind):
    rng = np.random.default_rng(77 + kind)
    xi, di, xj, dj, M = _pair(rng, 700, 1300, 1500, kind, dup=5)
    for prec, ratio in ((4.0, 0.8), (2.0, 0.6)):
        want = _oracle.ref_guided_match(kind, M, xi, di, xj, dj, prec * prec, ratio * ratio)
        got, _ = geofilter.guided_matching([xi, xj], [di, dj], [(0, 1)], [M], [prec], ratio, kind)
        assert np.array_equal(want, got.get((0, 1), np.zeros((0, 2), np.uint32))) and len(want) >= (300 if kind == 0 else 3)


def test_other_descriptor_lengths_and_argument_checks_emulated():
    rng = np.random.default_rng(3)
    with _emu.emulated():
        for nb in (64, 144):
            xi, di, xj, dj, M = _pair(rng, 30, 20, 25, 0, desc_bytes=nb)
            got, _ = geofilter.guided_matching([xi, xj], [di, dj], [(0, 1)], [M], [4.0], 0.8, 0)
            want = _oracle.port_guided_match(0, M, xi, di, xj, dj, 16.0, 0.8 * 0.8)
            assert np.array_equal(want, got.get((0, 1), np.zeros((0, 2), np.uint32))) and len(want) > 2
        xi, di, xj, dj, M = _pair(rng, 10, 5, 5, 0, desc_bytes=32)
        with pytest.raises(Exception):
            geofilter.guided_matching([xi, xj], [di, dj], [(0, 1)], [M], [4.0], 0.8, 0)      # 32-byte descriptors: not built
        xi, di, xj, dj, M = _pair(rng, 10, 5, 5, 0)
        with pytest.raises(Exception):
            geofilter.guided_matching([xi, xj], [di, dj], [(0, 2)], [M], [4.0], 0.8, 0)      # image index out of range
        with pytest.raises(Exception):
            geofilter.guided_matching([xi, xj], [di, dj], [(0, 1)], [M], [4.0], 0.8, 2)      # unknown kind


# ---- the other region types (round 6; VERDICT r5 "what's missing" 5): AKAZE float (L2<float>) and AKAZE binary (squared Hamming) ----
def _typed_pair(rng, n_true, n_ci, n_cj, kind, desc_type, dup=0, n_el=64):
    """_pair with float rows or bit rows derived from its uint8 descriptors (true matches look alike there): float = the bytes scaled and
    normalised to unit length (AKAZE's M-SURF rows are unit vectors; near-ties in the last float bits are what the summation order decides),
    bits = the four high bits of every byte (true matches differ in a few bits, everything else in half of them)"""
    xi, bi, xj, bj, M = _pair(rng, n_true, n_ci, n_cj, kind, desc_bytes=n_el, dup=dup)
    if desc_type == 1:
        di = bi.astype(np.float32) / np.float32(255); dj = bj.astype(np.float32) / np.float32(255)
        di = (di / np.maximum(np.linalg.norm(di, axis=1, keepdims=True), 1e-6)).astype(np.float32)
        dj = (dj / np.maximum(np.linalg.norm(dj, axis=1, keepdims=True), 1e-6)).astype(np.float32)
    else:
        di = bi & 0xF0; dj = bj & 0xF0
    return xi, np.ascontiguousarray(di), xj, np.ascontiguousarray(dj), M


@pytest.mark.skipif(not _oracle.have_ref_geofilter(), reason="oracle/_ref not built")
@pytest.mark.parametrize("desc_type", [1, 2])
@pytest.mark.parametrize("kind", [0, 1])
def test_typed_restatement_equals_the_reference_template(kind, desc_type):
    rng = np.random.default_rng(131 + 2 * kind + desc_type)
    total = 0
    for trial in range(4):
        xi, di, xj, dj, M = _typed_pair(rng, 120, 150, 170, kind, desc_type, dup=3 if trial % 2 else 0)
        for prec, ratio in ((4.0, 0.8), (60.0, 0.9), (30.0, 1.0)):
            want = _oracle.ref_guided_match_typed(kind, desc_type, M, xi, di, xj, dj, prec * prec, ratio * ratio)
            got = _oracle.port_guided_match_typed(kind, desc_type, M, xi, di, xj, dj, prec * prec, ratio * ratio)
            assert np.array_equal(want, got), (trial, prec, ratio, len(want), len(got))
            total += len(want)
    assert total > 30


def _typed_device_equals_restatement(emulated, n_pairs, sizes, seed):
    rng = np.random.default_rng(seed)
    for desc_type, n_el in ((1, 64), (1, 128), (2, 64), (2, 32)):
        for kind in (0, 1):
            feats, descs, pairs, models, precs = [], [], [], [], []
            for p in range(n_pairs):
                xi, di, xj, dj, M = _typed_pair(rng, *sizes, kind, desc_type, dup=2 if p % 3 == 0 else 0, n_el=n_el)
                feats += [xi, xj]; descs += [di, dj]
                pairs.append((2 * p, 2 * p + 1)); models.append(M)
                precs.append([4.0, 40.0, np.inf, 25.0][p % 4])
            feats += [np.zeros((0, 2))]; descs += [np.zeros((0, n_el), descs[0].dtype)]
            pairs += [(len(feats) - 1, 1), (0, len(feats) - 1)]; models += [models[0], models[0]]; precs += [4.0, 4.0]
            ratio = 0.9
            with (_emu.emulated() if emulated else contextlib.nullcontext()):
                got, st = geofilter.guided_matching(feats, descs, pairs, models, precs, ratio, kind, desc_type=desc_type)
            n = 0
            for p, (I, J) in enumerate(pairs):
                th = precs[p] * precs[p]
                want = (_oracle.port_guided_match_typed(kind, desc_type, models[p], feats[I], descs[I], feats[J], descs[J], th, ratio * ratio)
                        if np.isfinite(th) else np.zeros((0, 2), np.uint32))
                have = got.get((I, J), np.zeros((0, 2), np.uint32))
                assert np.array_equal(want, have), (desc_type, n_el, kind, p, len(want), len(have))
                n += len(want)
            assert n > 0 and st.n_matches == n and st.n_geometric_passed > 0


def test_typed_device_code_equals_the_restatement_emulated():
    _typed_device_equals_restatement(True, 4, (40, 50, 45), 15)


@pytest.mark.gpu
def test_typed_device_equals_the_restatement_on_the_mi355x():
    _typed_device_equals_restatement(False, 12, (400, 900, 1100), 17)


@pytest.mark.gpu
@pytest.mark.skipif(not _oracle.have_ref_geofilter(), reason="oracle/_ref not built")
@pytest.mark.parametrize("desc_type", [1, 2])
@pytest.mark.parametrize("kind", [0, 1])
def test_typed_device_equals_the_reference_template_on_the_mi355x(kind, desc_type):
    rng = np.random.default_rng(177 + 2 * kind + desc_type)
    xi, di, xj, dj, M = _typed_pair(rng, 700, 1300, 1500, kind, desc_type, dup=5)
    for prec, ratio in ((4.0, 0.8), (50.0, 0.9)):
        want = _oracle.ref_guided_match_typed(kind, desc_type, M, xi, di, xj, dj, prec * prec, ratio * ratio)
        got, _ = geofilter.guided_matching([xi, xj], [di, dj], [(0, 1)], [M], [prec], ratio, kind, desc_type=desc_type)
        assert np.array_equal(want, got.get((0, 1), np.zeros((0, 2), np.uint32))), (prec, ratio, len(want))
    assert len(want) > 3


# ---- through the replacement TU: ImageCollectionGeometricFilter::Robust_model_estimation(functor, putative, b_guided_matching = true) ----
def _matching_descriptors(feats, putative, seed, desc_type=0):
    """descriptors under which the putative matches look alike (so that guided matching keeps many of them) and everything else is noise;
    desc_type 1 / 2: 64 floats (unit rows) / 64 bytes of bits derived from such bytes as in _typed_pair"""
    rng = np.random.default_rng(seed)
    n_el = 128 if desc_type == 0 else 64
    descs = [rng.integers(0, 256, (len(f), n_el), dtype=np.uint8) for f in feats]
    for (I, J), m in putative.items():
        descs[J][m[:, 1]] = np.clip(descs[I][m[:, 0]].astype(int) + rng.integers(-10, 11, (len(m), n_el)), 0, 255).astype(np.uint8)
    if desc_type == 1:
        descs = [d.astype(np.float32) / np.float32(255) for d in descs]
        descs = [np.ascontiguousarray((d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6)).astype(np.float32)) for d in descs]
    elif desc_type == 2:
        descs = [d & 0xF0 for d in descs]
    return descs


def _guided_counters(lib, reset=True):
    import ctypes as C
    out = (C.c_uint64 * 2)()
    lib.mvgx_adapter_guided_counters(out, 1 if reset else 0)
    return int(out[0]), int(out[1])


def _adapter_guided_case(kind, model, n_pairs, n_max, desc_type=0):
    from tests import _geofilter_scene
    ref_lib, lib = _oracle.geofilter_container_lib("reference"), _oracle.geofilter_container_lib(kind)
    if ref_lib is None or lib is None:
        pytest.skip("needs the reference library and the adapter harness (tools/prep_gpu.sh)")
    kw = dict(homography=True, inlier_frac=(0.6, 0.9)) if model == "h" else dict(size=(1000, 1000)) if model == "e" else {}
    feats, wh, putative = _geofilter_scene.collection(n_pairs=n_pairs, seed=33, n_min=40, n_max=n_max, no_geometry_frac=0.2, **kw)
    descs = _matching_descriptors(feats, putative, 5, desc_type)
    args = dict(max_iterations=512, guided=True, ratio=0.8, descs=descs, model=model, focal=900.0 if model == "e" else 0.0, desc_type=desc_type)
    want = _oracle.geofilter_container("reference", feats, wh, putative, **args)
    _guided_counters(lib)
    got = _oracle.geofilter_container(kind, feats, wh, putative, **args)
    on_device, on_host = _guided_counters(lib)
    assert set(want) == set(got) and len(want) >= 2
    # (a pair whose FIRST stage ends on another inlier set - the parity policy of the a-contrario stage, tests/_geofilter_cases.py - has
    # another model and so other guided matches: such pairs are counted, not compared)
    differing = [k for k in want if not np.array_equal(want[k], got[k])]
    from tests import _geofilter_cases as gc
    assert len(differing) <= gc.allowed_differing(len(want), model), (len(differing), len(want))
    assert on_device == len(got) and on_host == 0, (on_device, on_host, len(got))      # every accepted pair was guided on the device
    assert sum(len(v) for v in got.values()) > 0 or model == "h"   # (a left feature needs TWO right features inside the bound: short lists)
    return want, got


@pytest.mark.parametrize("model", ["f", "h", "e"])
def test_adapter_guided_matching_runs_the_device_code_emulated(model):
    _adapter_guided_case("adapter_emu", model, 5, 60)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["f", "h", "e"])
def test_adapter_guided_matching_on_the_mi355x(model):
    _adapter_guided_case("adapter", model, 60, 250)


# AKAZE float / AKAZE binary regions through the replacement TU (round 6): the regions' own metric on the device, no host pair
@pytest.mark.parametrize("desc_type,model", [(1, "f"), (2, "f"), (2, "h")])
def test_adapter_guided_matching_other_region_types_emulated(desc_type, model):
    _adapter_guided_case("adapter_emu", model, 4, 60, desc_type)


@pytest.mark.gpu
@pytest.mark.parametrize("desc_type", [1, 2])
@pytest.mark.parametrize("model", ["f", "h", "e"])
def test_adapter_guided_matching_other_region_types_on_the_mi355x(model, desc_type):
    _adapter_guided_case("adapter", model, 40, 250, desc_type)


def test_adapter_guided_matching_injected_failure_takes_the_reference_member_function(monkeypatch, capfd):
    from tests import _geofilter_scene
    ref_lib, lib = _oracle.geofilter_container_lib("reference"), _oracle.geofilter_container_lib("adapter_emu")
    if ref_lib is None or lib is None:
        pytest.skip("needs the reference library and the adapter harness")
    feats, wh, putative = _geofilter_scene.collection(n_pairs=4, seed=33, n_min=40, n_max=60, no_geometry_frac=0.0)
    descs = _matching_descriptors(feats, putative, 5)
    args = dict(max_iterations=512, guided=True, ratio=0.8, descs=descs)
    want = _oracle.geofilter_container("reference", feats, wh, putative, **args)
    _guided_counters(lib)
    monkeypatch.setenv("MVGX_ADAPTER_INJECT_FAILURE", "geofilter:guided")
    got = _oracle.geofilter_container("adapter_emu", feats, wh, putative, **args)
    monkeypatch.delenv("MVGX_ADAPTER_INJECT_FAILURE")
    on_device, on_host = _guided_counters(lib)
    import ctypes as C
    out = (C.c_uint64 * 3)()
    lib.mvgx_adapter_counters(out, 1)
    assert on_device == 0 and on_host == len(got) and int(out[2]) == 1
    assert set(want) == set(got) and all(np.array_equal(want[k], got[k]) for k in want)



#  main_GeometricFilter -g h passes ratio -1 with guided matching, which selects H_ACRobust.hpp:166-187's geometry-only matching.
# The replacement TU leaves such a call to the functor's own member function (it used to run the descriptor-ratio kernel with ratio^2 = 1).
# There is no test against the reference for it: the reference's own code for that branch cannot run - H_ACRobust.hpp:118-124 takes the
# output matrix as Eigen::Ref<Mat> and resizes it (an assertion in a build with assertions, a write through a null pointer without:
# python /tmp: "DenseBase::resize() does not actually allow to resize", observed with oracle/_ref's build of the file). What the adapter
# does with a negative ratio is therefore exactly what the host application's functor does.
