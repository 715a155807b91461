"""MI355X parity tests of the float-descriptor path (mvgx_l2f_*, openmvg_amd/csrc/mvgx_bruteforce.hip) through the C ABI:
match lists bit-identical to the C restatement (oracle_l2_f32 = the reference's summation order), to the reference's committed
output (tests/golden/l2f_golden.npz) and, when its build travelled, to the reference itself."""
import numpy as np
import pytest

from openmvg_amd import matching, synth
from tests import _oracle
from tests.test_l2f_cpu import GOLD, golden_case

pytestmark = pytest.mark.gpu


def run_hip(imgs, pairs, ratio, batch_pairs=None):
    ctx = matching.L2fContext(0)
    try:
        if batch_pairs:
            ctx.set_option("batch_pairs", batch_pairs)
        ctx.set_regions(imgs, 64)
        return ctx.run(pairs, np.float32(ratio) * np.float32(ratio))
    finally:
        ctx.close()


@pytest.mark.parametrize("ratio", [0.8, 0.6, 1.0])
def test_golden_and_reference(ratio):
    imgs, pairs = golden_case()
    _, off, ij = run_hip(imgs, pairs, ratio)
    g = np.load(GOLD)
    key = f"r{int(round(ratio * 100))}"
    assert np.array_equal(off, g[key + "_offsets"]) and np.array_equal(ij, g[key + "_ij"])
    if _oracle.have_ref_match():
        ref = _oracle.ref_matcher_regions_match_float64(imgs, pairs, ratio)
        got = _oracle.offsets_to_dict(pairs, off, ij)
        assert got.keys() == ref.keys() and all(np.array_equal(got[k], ref[k]) for k in ref)


def test_ragged_batched_and_magnitudes():
    rng = np.random.default_rng(8)
    a = (rng.standard_normal((71, 64)) * np.logspace(-3, 3, 64)).astype(np.float32)
    b = a[::-1].copy() + (1e-3 * rng.standard_normal((71, 64))).astype(np.float32)
    b[::2] = a[::2]; b[1] = b[3]
    c = np.zeros((3, 64), np.float32); c[1] = a[5]
    imgs = [a, b, c] + synth.float_descriptors(4, [257, 0, 1, 300], seed=2)
    pairs = np.array([(i, j) for i in range(len(imgs)) for j in range(len(imgs)) if i != j], np.uint32)
    for ratio, bp in ((1.0, None), (0.5, 5), (0.8, None)):
        o_off, o_ij = _oracle.port_matcher_regions_match_f32(imgs, pairs, ratio)
        _, off, ij = run_hip(imgs, pairs, ratio, bp)
        assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij), ratio


def test_akaze_like_2000_desc_sampled_vs_oracle_and_mirror():
    imgs = synth.float_descriptors(10, 2000, seed=21)
    pairs = matching.exhaustive_pairs_array(10)
    st, off, ij = run_hip(imgs, pairs, 0.8)
    assert st.n_desc_pairs == len(pairs) * 2000 * 2000 and int(off[-1]) > 1000
    sel = np.random.default_rng(0).choice(len(pairs), 6, replace=False)
    o_off, o_ij = _oracle.port_matcher_regions_match_f32(imgs, pairs[sel], 0.8)
    for n, k in enumerate(sel):
        assert np.array_equal(ij[int(off[k]):int(off[k + 1])], o_ij[int(o_off[n]):int(o_off[n + 1])])
    prov = matching.Regions_Provider({k: matching.Float_Regions(d[:300]) for k, d in enumerate(imgs[:3])})
    out = matching.PairWiseMatches()
    matching.Matcher_Regions(0.8, matching.EMatcherType.BRUTE_FORCE_L2, device=0).Match(prov, [(0, 1), (1, 2), (0, 2)], out)
    small = [d[:300] for d in imgs[:3]]
    o_off, o_ij = _oracle.port_matcher_regions_match_f32(small, np.array([[0, 1], [0, 2], [1, 2]], np.uint32), 0.8)
    want = _oracle.offsets_to_dict(np.array([[0, 1], [0, 2], [1, 2]]), o_off, o_ij)
    assert dict(out).keys() == want.keys() and all(np.array_equal(out[k], want[k]) for k in want)


def test_error_behaviour():
    ctx = matching.L2fContext(0)
    with pytest.raises(Exception):
        ctx.set_regions([np.zeros((3, 128), np.float32)], 128)
    ctx.set_regions([np.zeros((3, 64), np.float32)] * 2, 64)
    with pytest.raises(Exception):
        ctx.run(np.array([[0, 1]], np.uint32), 1.5)
    ctx.close()
