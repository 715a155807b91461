"""CPU tests of BRUTE_FORCE_L2 on uint8 descriptors of other lengths (AKAZE_Liop_Regions: 144): the C
restatement (oracle_matcher_regions_match_u8 with dim = 144) against the reference's own Matcher_Regions on AKAZE_Liop_Regions
and its committed output; the device code of mvgx_bruteforce.hip (emulated) against the restatement, incl. dim = 128 where the
MFMA path (also emulated) must agree too."""
import os

import numpy as np
import pytest

from openmvg_amd import matching, synth
from tests import _emu, _oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l2u8_liop_golden.npz")


def liop_like(sizes, dim=144, seed=0):
    """uint8 rows with planted near-duplicates across consecutive images (matches exist), full byte range"""
    rng = np.random.default_rng(seed)
    imgs = [rng.integers(0, 256, (n, dim), dtype=np.uint8) for n in sizes]
    for k in range(1, len(sizes)):
        m = min(sizes[k], sizes[k - 1])
        if m:
            imgs[k][:m] = np.clip(imgs[k - 1][:m].astype(np.int16) + rng.integers(-9, 10, (m, dim)), 0, 255).astype(np.uint8)
    return imgs


def golden_case():
    sizes = [0, 1, 2, 3, 63, 65, 255, 257, 300]
    imgs = liop_like(sizes, 144, seed=5)
    p = matching.exhaustive_pairs_array(len(sizes))
    return imgs, np.concatenate([p, p[:, ::-1]])


def _same(a, b):
    return a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("ratio", [0.8, 1.0])
def test_restatement_equals_reference_and_golden(ratio):
    imgs, pairs = golden_case()
    off, ij = _oracle.port_matcher_regions_match(imgs, pairs, ratio, dim=144)
    assert int(off[-1]) > 500
    g = np.load(GOLD)
    key = f"r{int(round(ratio * 100))}"
    assert np.array_equal(off, g[key + "_offsets"]) and np.array_equal(ij, g[key + "_ij"])
    if _oracle.have_ref_match():
        assert _same(_oracle.offsets_to_dict(pairs, off, ij), _oracle.ref_matcher_regions_match_liop144(imgs, pairs, ratio))


def _run_emu(imgs, pairs, ratio, dim, batch_pairs=None):
    with _emu.emulated():
        ctx = matching.L2u8Context()
        if batch_pairs:
            ctx.set_option("batch_pairs", batch_pairs)
        ctx.set_regions(imgs, dim)
        st, off, ij = ctx.run(pairs, np.float32(ratio) * np.float32(ratio))
        ctx.close()
    return st, off, ij


@pytest.mark.parametrize("ratio,batch", [(0.8, None), (1.0, 7)])
def test_emulated_device_code_equals_restatement(ratio, batch):
    imgs, pairs = golden_case()
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, ratio, dim=144)
    _, off, ij = _run_emu(imgs, pairs, ratio, 144, batch)
    assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij)


@pytest.mark.parametrize("dim", [64, 128])
def test_emulated_other_lengths_and_agreement_with_the_mfma_path(dim):
    sizes = [40, 0, 200, 5, 129]
    imgs = liop_like(sizes, dim, seed=9)
    pairs = np.array([(i, j) for i in range(5) for j in range(5) if i != j], np.uint32)
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8, dim=dim)
    _, off, ij = _run_emu(imgs, pairs, 0.8, dim)
    assert int(o_off[-1]) > 20 and np.array_equal(off, o_off) and np.array_equal(ij, o_ij)
    if dim == 128:   # the SIFT path on the matrix cores (emulated) gives the same lists
        with _emu.emulated():
            ctx = matching.MatchContext(0); ctx.set_regions(imgs)
            _, off2, ij2 = ctx.run(pairs, np.float32(0.8) * np.float32(0.8)); ctx.close()
        assert np.array_equal(off2, o_off) and np.array_equal(ij2, o_ij)


def test_emulated_extremes_mirror_and_errors():
    a = np.zeros((40, 144), np.uint8); b = np.full((40, 144), 255, np.uint8)     # d = 144 * 255^2
    rng = np.random.default_rng(1)
    a[::3] = rng.integers(0, 256, (14, 144), dtype=np.uint8); b[::2] = a[::2]; b[1] = b[3]
    imgs = [a, b, rng.integers(0, 256, (3, 144), dtype=np.uint8)]
    pairs = np.array([[0, 1], [1, 0], [0, 2], [2, 1]], np.uint32)
    for ratio in (1.0, 0.5):
        o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, ratio, dim=144)
        _, off, ij = _run_emu(imgs, pairs, ratio, 144)
        assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij)
    with _emu.emulated():
        prov = matching.Regions_Provider({k: matching.Regions(d) for k, d in enumerate(imgs)})
        out = matching.PairWiseMatches()
        matching.Matcher_Regions(0.8, matching.EMatcherType.BRUTE_FORCE_L2).Match(prov, [(0, 1), (0, 2), (1, 2)], out)
        ctx = matching.L2u8Context()
        with pytest.raises(Exception):
            ctx.set_regions([np.zeros((3, 100), np.uint8)], 100)
        ctx.close()
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, np.array([[0, 1], [0, 2], [1, 2]], np.uint32), 0.8, dim=144)
    assert _same(dict(out), _oracle.offsets_to_dict(np.array([[0, 1], [0, 2], [1, 2]]), o_off, o_ij))
