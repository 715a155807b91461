"""CPU: the C oracle reproduces the committed golden fixtures (outputs of the reference's own Matcher_Regions)."""
import numpy as np
import pytest

from tests import _golden, _oracle


@pytest.mark.parametrize("case", _golden.CASES)
def test_port_oracle_reproduces_reference_golden(case):
    imgs, pairs, ratio, ref = _golden.load_case(case)
    offsets, ij = _oracle.port_matcher_regions_match(imgs, pairs, ratio)
    got = _oracle.offsets_to_dict(pairs, offsets, ij)
    assert set(got) == set(ref)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
