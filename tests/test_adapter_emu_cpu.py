"""CPU test of the matcher adapter's C++ code (openmvg_amd/adapter/mvgx_matcher_regions.cpp: region-type dispatch, pair
batching, delivery threads, container fill): the replacement TU, driven by the reference's caller code
(oracle/ref_shim_match.cpp), linked against the HIP emulation libraries instead of libmvgx_hip.so. Same entry points as
tests/test_adapter_gpu.py; needs the openMVG tree (build container)."""
import numpy as np
import pytest

from openmvg_amd import matching, synth
from tests import _oracle
from tests.test_l2u8_cpu import liop_like

pytestmark = pytest.mark.skipif(_oracle.adapter_emu() is None, reason="openMVG tree / adapter objects not present")


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_sift_uint8_route():
    descs = synth.image_descriptors(5, n_desc=150, seed=11)
    descs[3] = descs[3][:0]; descs[4] = descs[4][:1]
    pairs = matching.exhaustive_pairs_array(5)
    got = _oracle.ref_matcher_regions_match(descs, pairs, 0.8, lib=_oracle.adapter_emu())
    off, ij = _oracle.port_matcher_regions_match(descs, pairs, 0.8)
    want = _oracle.offsets_to_dict(pairs, off, ij)
    assert sum(len(v) for v in want.values()) > 20
    _same(got, want)


def test_sift_route_on_the_devices_of_the_environment(monkeypatch):
    """MVGX_DEVICES in the caller's environment: Match runs several device contexts (here emulated ones), the lists arrive
    batch by batch on the calling thread, the container equals the single-context one"""
    descs = synth.image_descriptors(9, n_desc=110, seed=12)
    descs[2] = descs[2][:0]
    pairs = matching.exhaustive_pairs_array(9)
    off, ij = _oracle.port_matcher_regions_match(descs, pairs, 0.8)
    want = _oracle.offsets_to_dict(pairs, off, ij)
    for env in ("0,0", "0,0,0", "all"):
        monkeypatch.setenv("MVGX_DEVICES", env)
        _same(_oracle.ref_matcher_regions_match(descs, pairs, 0.8, lib=_oracle.adapter_emu()), want)


def test_liop_uint8_144_route():
    sizes = [120, 0, 130, 64, 1, 2]
    imgs = liop_like(sizes, 144, seed=21)
    pairs = matching.exhaustive_pairs_array(len(sizes))
    got = _oracle.ref_matcher_regions_match_liop144(imgs, pairs, 0.8, lib=_oracle.adapter_emu())
    off, ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8, dim=144)
    want = _oracle.offsets_to_dict(pairs, off, ij)
    assert sum(len(v) for v in want.values()) > 20
    _same(got, want)
    if _oracle.have_ref_match():
        _same(got, _oracle.ref_matcher_regions_match_liop144(imgs, pairs, 0.8))


def test_binary_and_float_routes():
    sizes = [100, 0, 90, 1, 2]
    b = synth.binary_descriptors(len(sizes), sizes, seed=9)
    pairs = matching.exhaustive_pairs_array(len(sizes))
    got = _oracle.ref_matcher_regions_match_binary64(b, pairs, 0.8, lib=_oracle.adapter_emu())
    off, ij = _oracle.port_matcher_regions_match_hamming(b, pairs, 0.8)
    _same(got, _oracle.offsets_to_dict(pairs, off, ij))
    f = synth.float_descriptors(len(sizes), sizes, seed=9)
    got = _oracle.ref_matcher_regions_match_float64(f, pairs, 0.8, lib=_oracle.adapter_emu())
    off, ij = _oracle.port_matcher_regions_match_f32(f, pairs, 0.8)
    want = _oracle.offsets_to_dict(pairs, off, ij)
    assert sum(len(v) for v in want.values()) > 10
    _same(got, want)


def test_ratio_above_one_uses_the_reference_route():
    descs = synth.image_descriptors(3, n_desc=60, seed=5)
    pairs = matching.exhaustive_pairs_array(3)
    got = _oracle.ref_matcher_regions_match(descs, pairs, 1.05, lib=_oracle.adapter_emu())
    if _oracle.have_ref_match():
        _same(got, _oracle.ref_matcher_regions_match(descs, pairs, 1.05))


@pytest.mark.parametrize("tag", ["synthetic", "synthetic_grid", "sceaux"])
def test_cascade_hashing_replacement_equals_the_reference_lists(tag):
    """Cascade_Hashing_Matcher_Regions::Match of the replacement TU (hashing stage = the reference's CascadeHasher on the host,
    matching stage = emulated device code, de-duplication = the reference's classes) against the reference's stored containers:
    same pairs, same lists in the same order - including the cases where the coordinate de-duplication removes matches"""
    from tests.test_cascade import load
    descs, xy, hs, bs, pairs, ref = load(tag)
    for ratio in (0.8, 0.6):
        got = _oracle.ref_cascade_matcher_regions_match(descs, xy, pairs, ratio, lib=_oracle.adapter_emu())
        _same(got, ref[int(ratio * 100)])
