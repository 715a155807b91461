"""Writes tests/golden/geofilter_ortho.npz: the REFERENCE's GeometricFilter_EOMatrix_RA estimation (oracle/_ref/libref_geofilter.so ::
ref_geofilter_eo_acransac) on tests/test_geofilter_ortho.golden_scene, with the inputs the device entry takes - the hnormalized bearing
vectors of the reference's own cameras and the per-pair camera-plane bound. Run in the build container."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _oracle  # noqa: E402
from tests.test_geofilter_ortho import GOLD_PATH, golden_scene, reference_inputs  # noqa: E402

tv, K = golden_scene()
hI, hJ, prec = reference_inputs(tv, K, 2.0)
r = _oracle.ref_geofilter_eo(tv, K, precision=2.0, max_iterations=1024)
np.savez_compressed(GOLD_PATH, hI=hI, hJ=hJ, prec=prec, start=tv["start"].astype(np.uint64), wh=np.asarray(tv["wh"], np.uint32), mask=r["mask"], ok=r["ok"],
                    F=r["F"], precision=r["precision"], nfa=r["nfa"])
print(int(r["ok"].sum()), "of", len(tv["start"]) - 1, "pairs ok;", int(r["mask"].sum()), "inliers;", os.path.getsize(GOLD_PATH), "bytes")
