"""MI355X parity tests of the binary-descriptor path (mvgx_hamming_*, openmvg_amd/csrc/mvgx_bruteforce.hip) through the C ABI:
bit-exact against the C restatement of the reference (oracle/match_oracle.c), the reference's committed output
(tests/golden/hamming_golden.npz) and, when its build travelled, the reference itself."""
import numpy as np
import pytest

from openmvg_amd import matching, synth
from tests import _oracle
from tests.test_hamming_cpu import GOLD, golden_case

pytestmark = pytest.mark.gpu


def run_hip(imgs, pairs, ratio, L=None, batch_pairs=None):
    ctx = matching.HammingContext(0)
    try:
        if batch_pairs:
            ctx.set_option("batch_pairs", batch_pairs)
        ctx.set_regions(imgs, L)
        return ctx.run(pairs, ratio)
    finally:
        ctx.close()


@pytest.mark.parametrize("ratio", [0.8, 0.6, 1.0])
def test_golden_and_reference(ratio):
    imgs, pairs = golden_case()
    _, off, ij = run_hip(imgs, pairs, ratio, 64)
    g = np.load(GOLD)
    key = f"r{int(round(ratio * 100))}"
    assert np.array_equal(off, g[key + "_offsets"]) and np.array_equal(ij, g[key + "_ij"])
    if _oracle.have_ref_match():
        ref = _oracle.ref_matcher_regions_match_binary64(imgs, pairs, ratio)
        got = _oracle.offsets_to_dict(pairs, off, ij)
        assert got.keys() == ref.keys() and all(np.array_equal(got[k], ref[k]) for k in ref)


@pytest.mark.parametrize("L", [64, 32, 20, 61])
def test_descriptor_lengths_and_batching(L):
    sizes = [40, 0, 300, 5, 257, 1, 2]
    imgs = synth.binary_descriptors(len(sizes), sizes, n_bytes=L, seed=3, flip_bits=max(2, L // 2))
    pairs = np.array([(i, j) for i in range(len(sizes)) for j in range(len(sizes)) if i != j], np.uint32)
    o_off, o_ij = _oracle.port_matcher_regions_match_hamming(imgs, pairs, 0.9, L)
    for bp in (None, 5):
        _, off, ij = run_hip(imgs, pairs, 0.9, L, bp)
        assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij)


def test_akaze_like_2000_desc_sampled_vs_oracle():
    imgs = synth.binary_descriptors(12, 2000, seed=21)
    pairs = matching.exhaustive_pairs_array(12)
    st, off, ij = run_hip(imgs, pairs, 0.8, 64)
    assert st.n_desc_pairs == len(pairs) * 2000 * 2000 and int(off[-1]) > 1000
    sel = np.random.default_rng(0).choice(len(pairs), 8, replace=False)
    o_off, o_ij = _oracle.port_matcher_regions_match_hamming(imgs, pairs[sel], 0.8)
    for n, k in enumerate(sel):
        assert np.array_equal(ij[int(off[k]):int(off[k + 1])], o_ij[int(o_off[n]):int(o_off[n + 1])])
    # size-independent properties: every emitted j ascending and unique per pair, i in range
    for k in range(len(pairs)):
        m = ij[int(off[k]):int(off[k + 1])]
        assert np.all(np.diff(m[:, 1].astype(np.int64)) > 0) and (m[:, 0] < 2000).all()


def test_duplicates_extremes_and_mirror():
    rng = np.random.default_rng(8)
    a = np.zeros((70, 64), np.uint8); b = np.full((70, 64), 255, np.uint8)
    a[::3] = rng.integers(0, 256, (24, 64), dtype=np.uint8)
    b[::2] = a[::2]; b[1] = b[3]
    c = rng.integers(0, 256, (70, 64), dtype=np.uint8); c[10] = a[12]; c[11] = a[12]
    imgs = [a, b, c]
    pairs = np.array([[0, 1], [1, 0], [0, 2], [2, 0], [1, 2], [2, 1]], np.uint32)
    for ratio in (1.0, 0.5):
        o_off, o_ij = _oracle.port_matcher_regions_match_hamming(imgs, pairs, ratio)
        _, off, ij = run_hip(imgs, pairs, ratio, 64)
        assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij)
    prov = matching.Regions_Provider({10 + k: matching.Binary_Regions(d) for k, d in enumerate(imgs)})
    out = matching.PairWiseMatches()
    matching.Matcher_Regions(0.8, matching.EMatcherType.BRUTE_FORCE_HAMMING, device=0).Match(prov, [(10, 11), (11, 12), (10, 12)], out)
    o_off, o_ij = _oracle.port_matcher_regions_match_hamming(imgs, np.array([[0, 1], [0, 2], [1, 2]], np.uint32), 0.8)
    want = _oracle.offsets_to_dict(np.array([[10, 11], [10, 12], [11, 12]]), o_off, o_ij)
    assert dict(out).keys() == want.keys() and all(np.array_equal(out[k], want[k]) for k in want)


def test_error_behaviour():
    ctx = matching.HammingContext(0)
    with pytest.raises(Exception):
        ctx.set_regions([np.zeros((3, 65), np.uint8)], 65)
    ctx.set_regions([np.zeros((3, 64), np.uint8)] * 2, 64)
    with pytest.raises(Exception):
        ctx.run(np.array([[0, 1]], np.uint32), 1.5)
    with pytest.raises(Exception):
        ctx.run(np.array([[0, 2]], np.uint32), 0.8)
    ctx.close()
