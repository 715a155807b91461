"""Parity of the angular (-g a, -g u) and orthographic (-g o) essential models at scale: the device against the compiled reference
(oracle/_ref/libref_geofilter.so on the box's host threads) on 20 000 pairs x 250 matches and on 3 000 pairs of mixed sizes.
One JSON line per case: pairs, pairs accepted by the reference, pairs whose inlier set / verdict differs; for the orthographic model
(closed form) also whether models, NFA values and bounds are bit-identical on the reference cameras' own bearing vectors."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmvg_amd import geofilter, synth  # noqa: E402
from tests import _geofilter_cases as gc, _oracle  # noqa: E402

for label, tv in (("20000 x 250", synth.two_view_matches_bulk(20000, n=250, seed=0xA11CE)), ("3000 mixed 4..1500", synth.two_view_matches(3000, seed=77, n_min=4, n_max=1500))):
    K = synth.two_view_calibration(tv)
    bI, bJ = _oracle.ref_pinhole_bearings(tv, K)
    for upright in (False, True):
        t0 = time.time(); ref = _oracle.ref_geofilter_angular(bI, bJ, tv["start"], upright=upright); t_ref = time.time() - t0
        mask, res, st = geofilter.filter_pairs_angular(bI, bJ, tv["start"], geofilter.GeometricFilter_ESphericalMatrix_AC_Angular(4.0, 2048, upright))
        differing, rep = gc.compare(tv["start"], ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"], model_tol=1e-3)
        print(json.dumps(dict(model="u" if upright else "a", workload=label, **rep, reference_seconds=round(t_ref, 2), device_call_ms=round(st.total_ms, 1),
                              device_kernel_ms=round(st.kernel_ms, 1))), flush=True)
    hI = np.ascontiguousarray(bI[:, :2] / bI[:, 2:3]); hJ = np.ascontiguousarray(bJ[:, :2] / bJ[:, 2:3])
    _, _, prec = geofilter.ortho_inputs(tv["xI"], tv["xJ"], tv["start"], np.asarray(K, np.float64).reshape(-1, 2, 3, 3), 2.0)
    t0 = time.time(); ref = _oracle.ref_geofilter_eo(tv, K, precision=2.0, max_iterations=1024); t_ref = time.time() - t0
    mask, res, st = geofilter.filter_pairs_ortho_prepared(hI, hJ, tv["start"], tv["wh"], prec, geofilter.GeometricFilter_EOMatrix_RA(2.0, 1024))
    differing, rep = gc.compare(tv["start"], ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"])
    ok = ref["ok"]
    exact = bool(np.array_equal(mask, ref["mask"]) and np.array_equal(res["F"][ok], ref["F"][ok]) and np.array_equal(res["nfa"][ok], ref["nfa"][ok]) and
                 np.array_equal(res["precision_robust"][ok], ref["precision"][ok]))
    print(json.dumps(dict(model="o", workload=label, **rep, bit_identical_models_nfa_bounds_masks=exact, reference_seconds=round(t_ref, 2),
                          device_call_ms=round(st.total_ms, 1), device_kernel_ms=round(st.kernel_ms, 1))), flush=True)
