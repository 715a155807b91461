"""Writes tests/golden/cascade_typed.npz with the REFERENCE's own Cascade_Hashing_Matcher_Regions::Match on AKAZE_Liop_Regions and
AKAZE_Float_Regions (oracle/_ref/libref_match.so, oracle/ref_shim_match.cpp) for tests/test_cascade_typed.typed_case.
Run in the build container."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _oracle  # noqa: E402
from tests.test_cascade_typed import GOLD, KINDS, typed_case  # noqa: E402

out = {}
for kind in KINDS:
    descs, xy, pairs = typed_case(kind)
    for ratio in (0.8, 0.6):
        ref = _oracle.ref_cascade_matcher_regions_match_typed(kind, descs, xy, pairs, ratio)
        key = f"{kind}/r{int(round(ratio * 100))}"
        out[key + "/keys"] = np.array(sorted(ref.keys()), np.uint32).reshape(-1, 2)
        for (a, b), m in ref.items():
            out[f"{key}/{a}_{b}"] = m.astype(np.uint32)
        print(kind, ratio, len(ref), sum(len(m) for m in ref.values()))
np.savez_compressed(GOLD, **out)
