"""MI355X parity tests of BRUTE_FORCE_L2 on uint8 descriptors of other lengths (mvgx_l2u8_*, AKAZE_Liop_Regions: 144) through
the C ABI: bit-exact against the C restatement, the reference's committed output and, when its build travelled, the reference."""
import numpy as np
import pytest

from openmvg_amd import matching
from tests import _oracle
from tests.test_l2u8_cpu import GOLD, golden_case, liop_like

pytestmark = pytest.mark.gpu


def run_hip(imgs, pairs, ratio, dim, batch_pairs=None):
    ctx = matching.L2u8Context(0)
    try:
        if batch_pairs:
            ctx.set_option("batch_pairs", batch_pairs)
        ctx.set_regions(imgs, dim)
        return ctx.run(pairs, np.float32(ratio) * np.float32(ratio))
    finally:
        ctx.close()


@pytest.mark.parametrize("ratio", [0.8, 1.0])
def test_golden_and_reference(ratio):
    imgs, pairs = golden_case()
    _, off, ij = run_hip(imgs, pairs, ratio, 144)
    g = np.load(GOLD)
    key = f"r{int(round(ratio * 100))}"
    assert np.array_equal(off, g[key + "_offsets"]) and np.array_equal(ij, g[key + "_ij"])
    if _oracle.have_ref_match():
        ref = _oracle.ref_matcher_regions_match_liop144(imgs, pairs, ratio)
        got = _oracle.offsets_to_dict(pairs, off, ij)
        assert got.keys() == ref.keys() and all(np.array_equal(got[k], ref[k]) for k in ref)


@pytest.mark.parametrize("dim", [64, 128, 144])
def test_lengths_batching_and_agreement_with_the_mfma_path(dim):
    sizes = [40, 0, 900, 5, 257, 1, 2]
    imgs = liop_like(sizes, dim, seed=9)
    pairs = np.array([(i, j) for i in range(len(sizes)) for j in range(len(sizes)) if i != j], np.uint32)
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8, dim=dim)
    for bp in (None, 5):
        _, off, ij = run_hip(imgs, pairs, 0.8, dim, bp)
        assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij)
    if dim == 128:
        ctx = matching.MatchContext(0); ctx.set_regions(imgs)
        _, off2, ij2 = ctx.run(pairs, np.float32(0.8) * np.float32(0.8)); ctx.close()
        assert np.array_equal(off2, o_off) and np.array_equal(ij2, o_ij)


def test_liop_like_2000_desc_sampled_vs_oracle_and_mirror():
    imgs = liop_like([2000] * 6, 144, seed=3)
    pairs = matching.exhaustive_pairs_array(6)
    st, off, ij = run_hip(imgs, pairs, 0.8, 144)
    assert st.n_desc_pairs == len(pairs) * 2000 * 2000 and int(off[-1]) > 1000
    sel = np.array([0, 7, 14])
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs[sel], 0.8, dim=144)
    for n, k in enumerate(sel):
        assert np.array_equal(ij[int(off[k]):int(off[k + 1])], o_ij[int(o_off[n]):int(o_off[n + 1])])
    small = [d[:200] for d in imgs[:3]]
    prov = matching.Regions_Provider({k: matching.Regions(d) for k, d in enumerate(small)})
    out = matching.PairWiseMatches()
    matching.Matcher_Regions(0.8, matching.EMatcherType.BRUTE_FORCE_L2, device=0).Match(prov, [(0, 1), (1, 2), (0, 2)], out)
    o_off, o_ij = _oracle.port_matcher_regions_match(small, np.array([[0, 1], [0, 2], [1, 2]], np.uint32), 0.8, dim=144)
    want = _oracle.offsets_to_dict(np.array([[0, 1], [0, 2], [1, 2]]), o_off, o_ij)
    assert dict(out).keys() == want.keys() and all(np.array_equal(out[k], want[k]) for k in want)
