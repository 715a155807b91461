"""Random small image pairs through the EMULATED geometric-filter kernel (both models) against the compiled reference
(oracle/_ref/libref_geofilter.so), CPU only. Usage: fuzz_emulated_geofilter.py [seconds (default 300)] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import geofilter, synth
from tests import _emu, _geofilter_cases as gc, _oracle
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
t0 = time.time(); n_pairs = {"f": 0, "h": 0}; n_diff = {"f": 0, "h": 0}; bad = 0
while time.time() - t0 < secs:
    model = "h" if rng.random() < 0.6 else "f"
    kw = dict(seed=int(rng.integers(1 << 30)), n_max=int(rng.integers(12, 70)), noise_px=float(rng.choice([0.2, 0.5, 1.0])),
              inlier_frac=(0.2, 0.9), no_geometry_frac=0.2, tiny_frac=0.15)
    its = int(rng.choice([40, 256, 1024]))
    tv = (synth.two_view_homography_matches if model == "h" else synth.two_view_matches)(4, **kw)
    ref = (_oracle.ref_geofilter_h if model == "h" else _oracle.ref_geofilter)(tv, 4.0, its)
    fun = (geofilter.GeometricFilter_HMatrix_AC if model == "h" else geofilter.GeometricFilter_FMatrix_AC)(4.0, its)
    try:
        with _emu.emulated():
            mask, res, _ = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], fun)
        differing, rep = gc.compare(tv["start"], ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"])
    except AssertionError as e:
        print("POLICY", model, kw, its, repr(e)[:200], flush=True); bad += 1; continue
    n_pairs[model] += rep["pairs"]; n_diff[model] += len(differing)
    if differing: print("differs", model, kw, its, differing, flush=True)
print("pairs", n_pairs, "differing", n_diff, "policy violations", bad)
