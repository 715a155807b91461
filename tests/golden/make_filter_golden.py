"""Writes tests/golden/ba_filters.npz with the REFERENCE's own RemoveOutliers_PixelResidualError / RemoveOutliers_AngleError
(oracle/_ref/libref_ba.so, sfm/sfm_data_filters.cpp) on tests/_ba_cases.filter_scene. Run in the build container."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _ba_cases, _oracle  # noqa: E402

out = {}
for model in (1, 2, 3, 4, 5, 7):
    sc = _ba_cases.filter_scene(model)
    keep, counts, ang = _oracle.ref_ba_filters(sc, 4.0, 2, 2.0)
    keep_a, counts_a, _ = _oracle.ref_ba_filters(sc, -1.0, 2, 2.0)   # angle filter alone
    out[f"m{model}_keep"] = keep; out[f"m{model}_counts"] = np.array(counts); out[f"m{model}_angles"] = ang
    out[f"m{model}_keep_angle_only"] = keep_a; out[f"m{model}_count_angle_only"] = np.array(counts_a[1])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ba_filters.npz"), **out)
print({k: (v.shape, v.sum()) for k, v in out.items() if k.endswith("counts")})
