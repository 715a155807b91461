"""Real descriptors (BASELINE configs[0] substitute): SIFT regions of the two SceauxCastle JPGs that ship with openMVG,
extracted by the reference's own SIFT_Anatomy_Image_describer and matched by the reference's own Matcher_Regions
(tests/golden/make_sceaux_golden.py -> tests/golden/sceaux_sift.npz, "data": "real"). CPU: the restatement reproduces the stored
lists; GPU: the device path does, through the C ABI and through the Matcher_Regions mirror."""
import os

import numpy as np
import pytest

from openmvg_amd import matching
from tests import _oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sceaux_sift.npz")
PAIRS = np.array([[0, 1], [1, 0]], np.uint32)


def _load():
    z = np.load(GOLDEN)
    descs = [z["desc0"], z["desc1"]]
    want = {r: {(0, 1): z[f"matches_r{r}_0_1"], (1, 0): z[f"matches_r{r}_1_0"]} for r in (80, 60)}
    return z, descs, want


def test_fixture_is_real_sift():
    z, descs, want = _load()
    assert list(z["images"]) == ["100_7101.jpg", "100_7102.jpg"] and tuple(z["size0"]) == (1416, 1064)
    assert descs[0].shape == (2783, 128) and descs[1].shape == (2557, 128) and descs[0].dtype == np.uint8
    # RootSIFT quantisation of the reference (sift_DescriptorExtractor.hpp:484-494): sum of squares ~ 512^2
    n2 = (descs[0].astype(np.int64) ** 2).sum(axis=1)
    assert 0.9 * 512 ** 2 < np.median(n2) < 1.1 * 512 ** 2
    assert len(want[80][(0, 1)]) == 1378 and len(want[60][(1, 0)]) == 1046
    f = z["feat0"]
    assert f.shape == (2783, 4) and (f[:, 0] >= 0).all() and (f[:, 0] <= 1416).all() and (f[:, 1] <= 1064).all()


@pytest.mark.parametrize("ratio", [0.8, 0.6])
def test_restatement_reproduces_the_reference_on_real_descriptors(ratio):
    _, descs, want = _load()
    off, ij = _oracle.port_matcher_regions_match(descs, PAIRS, ratio)
    got = _oracle.offsets_to_dict(PAIRS, off, ij)
    w = want[int(round(ratio * 100))]
    assert got.keys() == w.keys() and all(np.array_equal(got[k], w[k]) for k in w)


@pytest.mark.gpu
@pytest.mark.parametrize("ratio", [0.8, 0.6])
@pytest.mark.parametrize("variant", [1, 43])
def test_device_path_reproduces_the_reference_on_real_descriptors(ratio, variant):
    from tests.test_matching_gpu import run_hip
    _, descs, want = _load()
    st, off, ij = run_hip(descs, PAIRS, ratio, variant)
    got = _oracle.offsets_to_dict(PAIRS, off, ij)
    w = want[int(round(ratio * 100))]
    assert got.keys() == w.keys() and all(np.array_equal(got[k], w[k]) for k in w)
    assert int(st.n_desc_pairs) == 2 * 2783 * 2557


@pytest.mark.gpu
def test_matcher_regions_mirror_on_real_descriptors():
    _, descs, want = _load()
    prov = matching.Regions_Provider({k: matching.Regions(d) for k, d in enumerate(descs)})
    out = matching.PairWiseMatches()
    matching.Matcher_Regions(0.8, matching.EMatcherType.BRUTE_FORCE_L2).Match(prov, [(0, 1)], out)
    assert list(out) == [(0, 1)] and np.array_equal(out[(0, 1)], want[80][(0, 1)])


def test_emulated_device_code_on_real_descriptors():
    """the matching kernels under the HIP emulation (tests/_emu.py), default variant, on the real regions"""
    from tests import _emu
    _, descs, want = _load()
    with _emu.emulated():
        ctx = matching.MatchContext(0)
        ctx.set_regions(descs)
        _, off, ij = ctx.run(PAIRS[:1], np.float32(0.8) * np.float32(0.8))
        ctx.close()
    assert np.array_equal(ij, want[80][(0, 1)])
