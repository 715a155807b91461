"""CPU tests of the float-descriptor path (BRUTE_FORCE_L2 on AKAZE_Float_Regions,
matching/regions_matcher.cpp:119-124). Float addition is not associative, so 'parity' here means the reference's operation
order: the C restatement (oracle_l2_f32, metric.hpp:98-135) is pinned bit for bit to the reference's compiled L2<float> and
to its Matcher_Regions output; the device code of openmvg_amd/csrc/mvgx_bruteforce.hip (emulated) must then reproduce the
restatement's match lists exactly."""
import ctypes as C
import os

import numpy as np
import pytest

from openmvg_amd import matching, synth
from tests import _emu, _oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l2f_golden.npz")


def golden_case():
    sizes = [0, 1, 2, 3, 63, 64, 65, 255, 257, 300]
    imgs = synth.float_descriptors(len(sizes), sizes, seed=23)
    n = len(imgs)
    pairs = np.concatenate([matching.exhaustive_pairs_array(n), matching.exhaustive_pairs_array(n)[:, ::-1]])
    return imgs, pairs


def _same(a, b):
    return a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)


@pytest.mark.skipif(not _oracle.have_ref_match(), reason="oracle/_ref/libref_match.so not built")
def test_l2_float_metric_bitwise_equals_reference():
    rng = np.random.default_rng(4)
    L = _oracle.ref_match(); P = _oracle.port()
    L.ref_l2_f32.restype = C.c_float; L.ref_l2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    for size in (8, 64, 128):      # multiples of 8: the reference's scalar loop (metric.hpp:107-112)
        for _ in range(200):
            a = rng.standard_normal(size).astype(np.float32) * 10; b = rng.standard_normal(size).astype(np.float32) * 10
            r = np.float32(L.ref_l2_f32(a.ctypes.data, b.ctypes.data, size))
            o = np.float32(P.oracle_l2_f32(a.ctypes.data, b.ctypes.data, size))
            assert r.tobytes() == o.tobytes()


@pytest.mark.parametrize("ratio", [0.8, 0.6, 1.0])
def test_restatement_equals_reference_and_golden(ratio):
    imgs, pairs = golden_case()
    off, ij = _oracle.port_matcher_regions_match_f32(imgs, pairs, ratio)
    assert int(off[-1]) > 200
    g = np.load(GOLD)
    key = f"r{int(round(ratio * 100))}"
    assert np.array_equal(off, g[key + "_offsets"]) and np.array_equal(ij, g[key + "_ij"])
    if _oracle.have_ref_match():
        assert _same(_oracle.offsets_to_dict(pairs, off, ij), _oracle.ref_matcher_regions_match_float64(imgs, pairs, ratio))


def _run_emu(imgs, pairs, ratio_sq, batch_pairs=None):
    with _emu.emulated():
        ctx = matching.L2fContext()
        if batch_pairs:
            ctx.set_option("batch_pairs", batch_pairs)
        ctx.set_regions(imgs, 64)
        st, off, ij = ctx.run(pairs, ratio_sq)
        ctx.close()
    return st, off, ij


@pytest.mark.parametrize("ratio,batch", [(0.8, None), (1.0, 7), (0.6, None)])
def test_emulated_device_code_equals_restatement(ratio, batch):
    imgs, pairs = golden_case()
    o_off, o_ij = _oracle.port_matcher_regions_match_f32(imgs, pairs, ratio)
    _, off, ij = _run_emu(imgs, pairs, np.float32(ratio) * np.float32(ratio), batch)
    assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij)


def test_emulated_duplicates_large_magnitudes_and_mirror():
    """exact duplicates (d0 = d1 = 0), rows of very different magnitude (the summation order matters most), odd row counts
    (the pad row of the pairwise-interleaved layout must never become a neighbour), and the Matcher_Regions mirror"""
    rng = np.random.default_rng(8)
    a = (rng.standard_normal((71, 64)) * np.logspace(-3, 3, 64)).astype(np.float32)
    b = a[::-1].copy() + (1e-3 * rng.standard_normal((71, 64))).astype(np.float32)
    b[::2] = a[::2]; b[1] = b[3]
    c = np.zeros((3, 64), np.float32); c[1] = a[5]      # an all-zero row equals the pad row
    imgs = [a, b, c]
    pairs = np.array([[0, 1], [1, 0], [0, 2], [2, 0], [1, 2], [2, 1]], np.uint32)
    for ratio in (1.0, 0.5):
        o_off, o_ij = _oracle.port_matcher_regions_match_f32(imgs, pairs, ratio)
        _, off, ij = _run_emu(imgs, pairs, np.float32(ratio) * np.float32(ratio))
        assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij)
    with _emu.emulated():
        prov = matching.Regions_Provider({10 + k: matching.Float_Regions(d) for k, d in enumerate(imgs)})
        out = matching.PairWiseMatches()
        matching.Matcher_Regions(0.8, matching.EMatcherType.BRUTE_FORCE_L2).Match(prov, [(10, 11), (11, 12), (10, 12)], out)
    o_off, o_ij = _oracle.port_matcher_regions_match_f32(imgs, np.array([[0, 1], [0, 2], [1, 2]], np.uint32), 0.8)
    assert _same(dict(out), _oracle.offsets_to_dict(np.array([[10, 11], [10, 12], [11, 12]]), o_off, o_ij))


def test_emulated_error_behaviour():
    with _emu.emulated():
        ctx = matching.L2fContext()
        with pytest.raises(Exception):
            ctx.set_regions([np.zeros((3, 128), np.float32)], 128)    # only AKAZE_Float_Regions' 64 floats
        ctx.set_regions([np.zeros((3, 64), np.float32)] * 2, 64)
        with pytest.raises(Exception):
            ctx.run(np.array([[0, 1]], np.uint32), 1.5)
        ctx.close()
