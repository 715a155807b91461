"""Writes tests/golden/l2u8_liop_golden.npz with the REFERENCE's own Matcher_Regions(BRUTE_FORCE_L2) on AKAZE_Liop_Regions
(oracle/_ref/libref_match.so) for tests/test_l2u8_cpu.golden_case. Run in the build container."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _oracle  # noqa: E402
from tests.test_l2u8_cpu import GOLD, golden_case  # noqa: E402

imgs, pairs = golden_case()
out = {}
for ratio in (0.8, 1.0):
    ref = _oracle.ref_matcher_regions_match_liop144(imgs, pairs, ratio)
    offsets = np.zeros(len(pairs) + 1, np.uint64)
    chunks = []
    for k, (a, b) in enumerate(pairs):
        m = ref.get((int(a), int(b)), np.zeros((0, 2), np.uint32))
        offsets[k + 1] = offsets[k] + len(m)
        chunks.append(m)
    key = f"r{int(round(ratio * 100))}"
    out[key + "_offsets"] = offsets
    out[key + "_ij"] = np.concatenate(chunks).astype(np.uint32)
np.savez_compressed(GOLD, **out)
print({k: v.shape for k, v in out.items()})
