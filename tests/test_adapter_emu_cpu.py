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


# ---- error convention of the boundary (openmvg_amd/adapter/mvgx_adapter_policy.hpp) ----
def _counters(lib, reset=False):
    import ctypes as C
    out = (C.c_uint64 * 3)()
    lib.mvgx_adapter_counters(out, 1 if reset else 0)
    return {"device_pairs": int(out[0]), "fallback_pairs": int(out[1]), "device_failures": int(out[2])}


def test_device_route_is_the_route_of_the_parity_tests():
    """the counters of the adapter library prove which route produced a container: every pair from the (emulated) device, no failure"""
    lib = _oracle.adapter_emu()
    _counters(lib, reset=True)
    descs = synth.image_descriptors(4, n_desc=90, seed=3)
    pairs = matching.exhaustive_pairs_array(4)
    _oracle.ref_matcher_regions_match(descs, pairs, 0.8, lib=lib)
    c = _counters(lib, reset=True)
    assert c == {"device_pairs": len(pairs), "fallback_pairs": 0, "device_failures": 0}


@pytest.mark.parametrize("stage", ["match:create", "match:set_regions", "match:run"])
def test_injected_device_failure_finishes_on_the_reference_route(stage, monkeypatch, capfd):
    """a failing device call neither throws through Match() (the unchanged main_ComputeMatches has no handler) nor loses pairs: it
    is logged once and the reference's own RegionMatcherFactory route - linked into the application anyway - finishes the call"""
    lib = _oracle.adapter_emu()
    descs = synth.image_descriptors(5, n_desc=120, seed=11)
    descs[3] = descs[3][:0]
    pairs = matching.exhaustive_pairs_array(5)
    off, ij = _oracle.port_matcher_regions_match(descs, pairs, 0.8)
    want = _oracle.offsets_to_dict(pairs, off, ij)
    _counters(lib, reset=True)
    monkeypatch.setenv("MVGX_ADAPTER_INJECT_FAILURE", stage)
    got = _oracle.ref_matcher_regions_match(descs, pairs, 0.8, lib=lib)
    got2 = _oracle.ref_matcher_regions_match(descs, pairs, 0.8, lib=lib)   # second call: same result, no second log line
    monkeypatch.delenv("MVGX_ADAPTER_INJECT_FAILURE")
    _same(got, want); _same(got2, want)
    c = _counters(lib, reset=True)
    n_dev = int(sum(1 for a, b in pairs))   # every pair of this set is a device pair (empty regions are handled by the device path)
    assert c["device_failures"] == 2 and c["device_pairs"] == 0 and c["fallback_pairs"] == 2 * n_dev, c
    err = capfd.readouterr().err
    assert err.count("injected by MVGX_ADAPTER_INJECT_FAILURE") == 1 and "continuing with the reference's own CPU code" in err


def test_injected_failure_throws_when_asked(monkeypatch):
    """MVGX_ON_DEVICE_ERROR=throw: the caller prefers an exception (the shim reports it as a failed call)"""
    lib = _oracle.adapter_emu()
    descs = synth.image_descriptors(3, n_desc=40, seed=2)
    pairs = matching.exhaustive_pairs_array(3)
    monkeypatch.setenv("MVGX_ADAPTER_INJECT_FAILURE", "match:create")
    monkeypatch.setenv("MVGX_ON_DEVICE_ERROR", "throw")
    _counters(lib, reset=True)
    with pytest.raises(RuntimeError):
        _oracle.ref_matcher_regions_match(descs, pairs, 0.8, lib=lib)
    assert _counters(lib, reset=True)["fallback_pairs"] == 0


@pytest.mark.parametrize("stage", ["cascade:create", "cascade:hash", "cascade:run"])
def test_cascade_hashing_injected_failure_finishes_with_the_reference_classes(stage, monkeypatch):
    from tests.test_cascade import load
    descs, xy, hs, bs, pairs, ref = load("synthetic")
    lib = _oracle.adapter_emu()
    _counters(lib, reset=True)
    monkeypatch.setenv("MVGX_ADAPTER_INJECT_FAILURE", stage)
    got = _oracle.ref_cascade_matcher_regions_match(descs, xy, pairs, 0.8, lib=lib)
    monkeypatch.delenv("MVGX_ADAPTER_INJECT_FAILURE")
    _same(got, ref[80])
    c = _counters(lib, reset=True)
    assert c["device_failures"] == 1 and c["device_pairs"] == 0 and c["fallback_pairs"] > 0, c


@pytest.mark.parametrize("mode", ["host", "check"])
def test_cascade_hashing_host_hash_option(mode, monkeypatch):
    """MVGX_CASCADE_HASH=host keeps the hashing stage with the reference's CascadeHasher (for openMVG builds whose Eigen uses FMA),
    =check compares one view both ways first; either way the containers equal the reference's and the matching stage ran on the device"""
    from tests.test_cascade import load
    descs, xy, hs, bs, pairs, ref = load("synthetic")
    lib = _oracle.adapter_emu()
    _counters(lib, reset=True)
    monkeypatch.setenv("MVGX_CASCADE_HASH", mode)
    got = _oracle.ref_cascade_matcher_regions_match(descs, xy, pairs, 0.8, lib=lib)
    _same(got, ref[80])
    c = _counters(lib, reset=True)
    assert c["device_pairs"] > 0 and c["fallback_pairs"] == 0 and c["device_failures"] == 0, c
