"""Cascade_Hashing_Matcher_Regions on the scalar region types other than 128-byte SIFT (VERDICT r3 missing #3): 144-byte uint8 rows
(AKAZE_Liop_Regions) and 64-float rows (AKAZE_Float_Regions). The replacement TU keeps their hashing stage with the reference's
CascadeHasher on the host and runs the matching stage (bucket candidates, Hamming ranking, exact L2 of the ten best, ratio test) on the
device through mvgx_cascade_set_regions_typed; containers must equal the reference's (Cascade_Hashing_Matcher_Regions.cpp:233-262
dispatch, cascade_hasher.hpp:253-420), order included. Golden containers: tests/golden/make_cascade_typed_golden.py."""
import os

import numpy as np
import pytest

from openmvg_amd import matching, synth
from tests import _oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "cascade_typed.npz")
KINDS = ["liop144", "float64"]
SIZES = [0, 1, 37, 600, 1100, 1500]


def typed_case(kind):
    """descriptors with true correspondences across the images, feature positions with repeats (the coordinate de-duplication has work)"""
    rng = np.random.default_rng(0xCA5CADE if kind == "liop144" else 0xF10A7)
    if kind == "float64":
        descs = synth.float_descriptors(len(SIZES), SIZES, seed=21, noise=0.08)
    else:
        from tests.test_l2u8_cpu import liop_like
        descs = liop_like(SIZES, 144, seed=21)
    xy = []
    for n in SIZES:
        p = (rng.random((n, 2)) * 900).astype(np.float32)
        if n > 30:
            p[rng.integers(0, n, n // 10)] = p[rng.integers(0, n, n // 10)]   # repeated positions
        xy.append(p)
    p = matching.exhaustive_pairs_array(len(SIZES))
    return descs, xy, p


def golden(kind, ratio):
    z = np.load(GOLD)
    key = f"{kind}/r{int(round(ratio * 100))}"
    return {tuple(int(v) for v in k): z[f"{key}/{k[0]}_{k[1]}"] for k in z[f"{key}/keys"]}


def _same(a, b):
    assert sorted(a.keys()) == sorted(b.keys()), (sorted(a.keys()), sorted(b.keys()))
    for k in a:
        assert np.array_equal(a[k], b[k]), (k, len(a[k]), len(b[k]))


@pytest.mark.parametrize("kind", KINDS)
def test_golden_containers_are_the_reference(kind):
    if not _oracle.have_ref_match():
        pytest.skip("oracle/_ref not built")
    descs, xy, pairs = typed_case(kind)
    want = golden(kind, 0.8)
    assert sum(len(v) for v in want.values()) > 300
    _same(_oracle.ref_cascade_matcher_regions_match_typed(kind, descs, xy, pairs, 0.8), want)


@pytest.mark.parametrize("kind", KINDS)
def test_replacement_matching_stage_emulated(kind):
    lib = _oracle.adapter_emu()
    descs, xy, pairs = typed_case(kind)
    before = _counters(lib, reset=True)
    got = _oracle.ref_cascade_matcher_regions_match_typed(kind, descs, xy, pairs, 0.8, lib=lib)
    _same(got, golden(kind, 0.8))
    c = _counters(lib)
    assert c["device_pairs"] > 0 and c["fallback_pairs"] == 0 and c["device_failures"] == 0, (before, c)


def _hash_check(lib, kind):
    """MVGX_CASCADE_HASH=check: the replacement TU hashes one view on the device AND with this build's own CascadeHasher and compares the
    codes and bucket ids (round 5: the hashing stage of the 144-byte and 64-float shapes runs on the device too)"""
    import ctypes as C
    descs, xy, pairs = typed_case(kind)
    saved = os.environ.get("MVGX_CASCADE_HASH")
    os.environ["MVGX_CASCADE_HASH"] = "check"
    try:
        got = _oracle.ref_cascade_matcher_regions_match_typed(kind, descs, xy, pairs, 0.8, lib=lib)
    finally:
        if saved is None:
            os.environ.pop("MVGX_CASCADE_HASH", None)
        else:
            os.environ["MVGX_CASCADE_HASH"] = saved
    fn = lib.mvgx_adapter_cascade_last_hash_check
    fn.restype = C.c_int
    assert fn() == 1, fn()   # 1: equal (2: differ, 0: the check did not run)
    _same(got, golden(kind, 0.8))


@pytest.mark.parametrize("kind", KINDS)
def test_device_hashing_equals_the_reference_hasher_emulated(kind):
    _hash_check(_oracle.adapter_emu(), kind)


def test_other_lengths_stay_on_the_reference_route():
    """the C-ABI refuses shapes outside (uint8, 128 | 144) and (float, 64) with MVGX_ERR_UNSUPPORTED; nothing is left half set"""
    from openmvg_amd import _capi
    from tests import _emu
    with _emu.emulated():
        ctx = matching.CascadeContext(0)
        try:
            d = [np.zeros((4, 96), np.uint8)]
            with pytest.raises(_capi.MvgxError) as e:
                ctx.set_regions(d, [np.zeros((4, 12), np.uint8)], [np.zeros((4, 6), np.uint16)], dtype=np.uint8, dim=96)
            assert e.value.code == _capi.MVGX_ERR_UNSUPPORTED
        finally:
            ctx.close()


def _counters(lib, reset=False):
    import ctypes as C
    out = (C.c_uint64 * 3)()
    lib.mvgx_adapter_counters(out, 1 if reset else 0)
    return {"device_pairs": int(out[0]), "fallback_pairs": int(out[1]), "device_failures": int(out[2])}


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_device_hashing_equals_the_reference_hasher_on_the_mi355x(kind):
    _hash_check(_oracle.adapter(), kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_replacement_matching_stage_on_the_mi355x(kind):
    lib = _oracle.adapter()
    descs, xy, pairs = typed_case(kind)
    _counters(lib, reset=True)
    for ratio in (0.8, 0.6):
        got = _oracle.ref_cascade_matcher_regions_match_typed(kind, descs, xy, pairs, ratio, lib=lib)
        _same(got, golden(kind, ratio))
    c = _counters(lib)
    assert c["device_pairs"] == 2 * len(pairs) - 2 * 5 and c["fallback_pairs"] == 0 and c["device_failures"] == 0, c
    if _oracle.have_ref_match():    # and live against the reference TU at another ratio
        _same(_oracle.ref_cascade_matcher_regions_match_typed(kind, descs, xy, pairs, 0.9, lib=lib),
              _oracle.ref_cascade_matcher_regions_match_typed(kind, descs, xy, pairs, 0.9))
