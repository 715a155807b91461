"""CASCADE_HASHING_L2 (the default -n of main_ComputeMatches): parity of the matching stage. The hashing stage (per-descriptor hash
code + bucket ids: single-precision Eigen products of the reference's CascadeHasher) is the reference's own code everywhere - in
the openMVG adapter and, here, in the stored fixture tests/golden/cascade_hashing.npz (make_cascade_golden.py) - so the device
stage is integer work and must reproduce the reference exactly:
  * restatement (oracle/match_oracle.c) against the reference's final lists (CPU),
  * emulated device code against the restatement (CPU), device code against the restatement and the reference (GPU).
The reference finishes with a coordinate de-duplication whose std::set ordering (indMatchDecoratorXY.hpp:43-57) is only a strict
order when the left features have distinct x and distinct y; the 'synthetic' case is built that way and is compared list against
list (order = ascending y of the left feature), the 'grid' and real cases - where that step removes matches - through the adapter
(tests/test_adapter_*.py) and here as set inclusion."""
import os

import numpy as np
import pytest

from openmvg_amd import matching
from tests import _oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cascade_hashing.npz")


def load(tag):
    z = np.load(GOLDEN)
    n = int(z[f"{tag}/n_images"])
    descs = [z[f"{tag}/desc{k}"] for k in range(n)]
    xy = [z[f"{tag}/xy{k}"] for k in range(n)]
    hs = [z[f"{tag}/hash{k}"] for k in range(n)]
    bs = [z[f"{tag}/bids{k}"] for k in range(n)]
    pairs = z[f"{tag}/pairs"]
    ref = {r: {tuple(k): z[f"{tag}/r{r}/{k[0]}_{k[1]}"] for k in z[f"{tag}/r{r}/keys"]} for r in (80, 60)}
    return descs, xy, hs, bs, pairs, ref


def finish_distinct(m, xyI):
    """the reference's two de-duplication steps on data with distinct left coordinates: set of (i, j), then order by y of the left feature"""
    m = np.unique(np.asarray(m, np.uint32).reshape(-1, 2), axis=0)
    return m[np.argsort(xyI[m[:, 0], 1], kind="stable")]


def device_lists(descs, hs, bs, pairs, ratio, batch_pairs=None):
    ctx = matching.CascadeContext(0)
    try:
        if batch_pairs:
            ctx.set_option("batch_pairs", batch_pairs)
        ctx.set_regions(descs, hs, bs)
        r = np.float32(ratio)
        st, off, ij = ctx.run(pairs, r * r)
    finally:
        ctx.close()
    return st, _oracle.offsets_to_dict(pairs, off, ij)


@pytest.mark.parametrize("ratio", [0.8, 0.6])
def test_restatement_reproduces_the_reference(ratio):
    descs, xy, hs, bs, pairs, ref = load("synthetic")
    want = ref[int(ratio * 100)]
    got = {}
    for I, J in pairs:
        if len(descs[I]) == 0:
            continue
        m = _oracle.port_cascade_match_pair(descs[I], hs[I], bs[I], descs[J], hs[J], bs[J], ratio)
        if len(m):
            got[(int(I), int(J))] = finish_distinct(m, xy[I])
    assert got.keys() == want.keys() and sum(len(v) for v in want.values()) > 400
    for k in want:
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("tag", ["synthetic_grid", "sceaux"])
def test_restatement_contains_the_reference_where_coordinates_repeat(tag):
    descs, xy, hs, bs, pairs, ref = load(tag)
    removed = 0
    for (I, J), want in ref[80].items():
        m = _oracle.port_cascade_match_pair(descs[I], hs[I], bs[I], descs[J], hs[J], bs[J], 0.8)
        got = set(map(tuple, m))
        assert set(map(tuple, want)) <= got
        removed += len(got) - len(want)
    assert removed > 0        # the coordinate step did remove matches: these cases need the adapter for list equality


def _emulated_or_gpu(fn):
    return fn


def test_emulated_device_code_equals_the_restatement():
    from tests import _emu
    descs, xy, hs, bs, pairs, ref = load("synthetic")
    with _emu.emulated():
        st, got = device_lists(descs, hs, bs, pairs, 0.8, batch_pairs=3)
    want = {}
    for I, J in pairs:
        if len(descs[I]) and len(descs[J]):
            m = _oracle.port_cascade_match_pair(descs[I], hs[I], bs[I], descs[J], hs[J], bs[J], 0.8)
            if len(m):
                want[(int(I), int(J))] = m
    assert got.keys() == want.keys()
    for k in want:
        assert np.array_equal(got[k], want[k]), k        # before de-duplication: ascending query order, identical
    for k, v in ref[80].items():
        assert np.array_equal(finish_distinct(got[k], xy[k[0]]), v), k


def hashed_on_device(descs, fetch=True):
    ctx = matching.CascadeContext(0)
    try:
        return ctx.hash_regions(descs, fetch=fetch)
    finally:
        ctx.close()


@pytest.mark.skipif(not _oracle.have_ref_match(), reason="oracle/_ref not built")
def test_zero_mean_restatement_equals_the_compiled_reference():
    """matching.cascade_zero_mean (Eigen 3.4's reduction order restated in numpy float32) against the reference's
    GetZeroMeanDescriptor chain, bit for bit, over image counts that move the 32-byte alignment of the per-image matrix' columns"""
    from openmvg_amd import synth
    for n_img, n_desc in [(1, 50), (2, 40), (3, 33), (7, 10), (8, 9), (9, 60), (16, 20), (17, 20), (25, 7), (33, 3), (41, 12)]:
        descs = synth.random_descriptors(n_img, [n_desc + (k % 3) for k in range(n_img)], seed=n_img)
        if n_img > 2:
            descs[1] = np.zeros((0, 128), np.uint8)
        zr = _oracle.ref_cascade_zero_mean(descs)
        zm = matching.cascade_zero_mean(descs)
        assert np.array_equal(zr.view(np.uint32), zm.view(np.uint32)), (n_img, n_desc)
    descs, xy, hs, bs, pairs, ref = load("sceaux")
    assert np.array_equal(_oracle.ref_cascade_zero_mean(descs).view(np.uint32), matching.cascade_zero_mean(descs).view(np.uint32))


def test_emulated_hashing_stage_equals_the_stored_reference_codes():
    """mvgx_cascade_hash_regions (projections generated like CascadeHasher::Init, Eigen's product order per descriptor) under the
    HIP emulation: hash codes and bucket ids equal the ones the reference's CreateHashedDescriptions produced for the fixture, and
    the lists that follow equal the ones obtained from the reference's codes"""
    from tests import _emu
    descs, xy, hs, bs, pairs, ref = load("synthetic")
    with _emu.emulated():
        h, b = hashed_on_device(descs)
        ctx = matching.CascadeContext(0)
        try:
            ctx.hash_regions(descs)
            st, off, ij = ctx.run(pairs, np.float32(0.8) * np.float32(0.8))
        finally:
            ctx.close()
        st2, want = device_lists(descs, hs, bs, pairs, 0.8)
    for k in range(len(descs)):
        assert np.array_equal(h[k], hs[k]), k
        assert np.array_equal(b[k], bs[k]), k
    got = _oracle.offsets_to_dict(pairs, off, ij)
    assert got.keys() == want.keys() and all(np.array_equal(got[k], want[k]) for k in want)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["synthetic", "synthetic_grid", "sceaux"])
def test_hashing_stage_on_the_device_equals_the_stored_reference_codes(tag):
    descs, xy, hs, bs, pairs, ref = load(tag)
    h, b = hashed_on_device(descs)
    for k in range(len(descs)):
        assert np.array_equal(h[k], hs[k]), (tag, k)
        assert np.array_equal(b[k], bs[k]), (tag, k)
    ctx = matching.CascadeContext(0)
    try:
        ctx.hash_regions(descs)
        st, off, ij = ctx.run(pairs, np.float32(0.8) * np.float32(0.8))
    finally:
        ctx.close()
    st2, want = device_lists(descs, hs, bs, pairs, 0.8)
    got = _oracle.offsets_to_dict(pairs, off, ij)
    assert got.keys() == want.keys() and all(np.array_equal(got[k], want[k]) for k in want)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,ratio", [("synthetic", 0.8), ("synthetic", 0.6), ("synthetic_grid", 0.8), ("sceaux", 0.8), ("sceaux", 0.6)])
def test_device_code_equals_the_restatement_and_the_reference(tag, ratio):
    descs, xy, hs, bs, pairs, ref = load(tag)
    st, got = device_lists(descs, hs, bs, pairs, ratio, batch_pairs=4)
    want_ref = ref[int(ratio * 100)]
    for I, J in pairs:
        key = (int(I), int(J))
        m = _oracle.port_cascade_match_pair(descs[I], hs[I], bs[I], descs[J], hs[J], bs[J], ratio) if len(descs[I]) else np.zeros((0, 2), np.uint32)
        assert np.array_equal(got.get(key, np.zeros((0, 2), np.uint32)), m), key
        if key in want_ref:
            if tag == "synthetic":
                assert np.array_equal(finish_distinct(got[key], xy[I]), want_ref[key]), key
            else:
                assert set(map(tuple, want_ref[key])) <= set(map(tuple, got[key]))
    assert int(st.n_matches) == sum(len(v) for v in got.values()) > 0


@pytest.mark.gpu
def test_any_ratio_is_reproduced():
    """(distance, id) pairs are totally ordered: unlike the brute-force path there is no tie ambiguity, ratio > 1 included"""
    descs, xy, hs, bs, pairs, ref = load("synthetic")
    st, got = device_lists(descs, hs, bs, pairs[:3], 1.2)
    for I, J in pairs[:3]:
        m = _oracle.port_cascade_match_pair(descs[I], hs[I], bs[I], descs[J], hs[J], bs[J], 1.2)
        assert np.array_equal(got.get((int(I), int(J)), np.zeros((0, 2), np.uint32)), m)


# ---- bucket layouts that load the selection logic (tests/golden/cascade_layouts.npz, make_cascade_layouts_golden.py) -------------
LAYOUTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cascade_layouts.npz")


def _layout_cases():
    z = np.load(LAYOUTS)
    return [tuple(int(v) for v in row) for row in z["layouts"]]


def _layout(g, b):
    z = np.load(LAYOUTS)
    t = f"g{g}b{b}"
    return z["descI"], z["descJ"], z[f"{t}/hashI"], z[f"{t}/bidsI"], z[f"{t}/hashJ"], z[f"{t}/bidsJ"], {0.8: z[f"{t}/r8"], 1.3: z[f"{t}/r13"]}


@pytest.mark.parametrize("g,b", _layout_cases())
def test_restatement_on_crowded_buckets(g, b):
    dI, dJ, hI, bI, hJ, bJ, want = _layout(g, b)
    for ratio, w in want.items():
        assert np.array_equal(_oracle.port_cascade_match_pair(dI, hI, bI, dJ, hJ, bJ, ratio, g, b), w), (g, b, ratio)
    assert len(want[1.3]) >= len(want[0.8]) > (50 if b <= 10 else 0)     # 2^16 buckets: almost every bucket is empty


def _device_pair(dI, dJ, hI, bI, hJ, bJ, ratio, g, b):
    ctx = matching.CascadeContext(0)
    try:
        ctx.set_regions([dI, dJ], [hI, hJ], [bI, bJ], n_groups=g, bits_per_bucket=b)
        r = np.float32(ratio)
        _, off, ij = ctx.run(np.array([[0, 1]], np.uint32), r * r)
    finally:
        ctx.close()
    return ij


@pytest.mark.parametrize("g,b", [(6, 2), (3, 1), (8, 4), (8, 16)])
def test_emulated_device_code_on_crowded_buckets(g, b):
    from tests import _emu
    dI, dJ, hI, bI, hJ, bJ, want = _layout(g, b)
    with _emu.emulated():
        for ratio, w in want.items():
            assert np.array_equal(_device_pair(dI, dJ, hI, bI, hJ, bJ, ratio, g, b), w), (g, b, ratio)


@pytest.mark.gpu
@pytest.mark.parametrize("g,b", _layout_cases())
def test_device_code_on_crowded_buckets(g, b):
    dI, dJ, hI, bI, hJ, bJ, want = _layout(g, b)
    for ratio, w in want.items():
        assert np.array_equal(_device_pair(dI, dJ, hI, bI, hJ, bJ, ratio, g, b), w), (g, b, ratio)
