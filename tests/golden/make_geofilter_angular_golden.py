"""Writes tests/golden/geofilter_angular.npz with the REFERENCE's own ACKernelAdaptor_AngularRadianError + ACRANSAC
(oracle/_ref/libref_geofilter.so :: ref_geofilter_e_angular_acransac, pose stage off) on tests/test_geofilter_angular.golden_case,
for the eight-point ("a_") and the three-point upright ("u_") solver. Run in the build container."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _oracle  # noqa: E402
from tests.test_geofilter_angular import GOLD_PATH, golden_case  # noqa: E402

bI, bJ, start = golden_case()
out = dict(bI=bI, bJ=bJ, start=start)
for tag, upright in (("a", False), ("u", True)):
    r = _oracle.ref_geofilter_angular(bI, bJ, start, upright=upright)
    out.update({tag + "_mask": r["mask"], tag + "_ok": r["ok"], tag + "_F": r["F"], tag + "_precision": r["precision"], tag + "_nfa": r["nfa"]})
    print(tag, int(r["ok"].sum()), "of", len(start) - 1, "pairs ok;", int(r["mask"].sum()), "inliers")
np.savez_compressed(GOLD_PATH, **out)
print(os.path.getsize(GOLD_PATH), "bytes")
