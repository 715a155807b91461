#!/usr/bin/env python
"""Generates tests/golden/geofilter.npz: synthetic two-view correspondences (openmvg_amd.synth.two_view_matches, fixed seed) and what
the REFERENCE's own GeometricFilter_FMatrix_AC kernel + ACRANSAC return for them (oracle/_ref/libref_geofilter.so, compiled from
/root/reference by oracle/Makefile). Run in the build container: python tests/golden/make_geofilter_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openmvg_amd import synth
from tests import _oracle

tv = synth.two_view_matches(240, seed=2024, n_max=300)
ref = _oracle.ref_geofilter(tv, precision=4.0, max_iterations=2048)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "geofilter.npz"), xI=tv["xI"], xJ=tv["xJ"], start=tv["start"], wh=tv["wh"],
                    mask=ref["mask"], ok=ref["ok"], F=ref["F"], precision=ref["precision"], nfa=ref["nfa"],
                    precision_px=np.float64(4.0), max_iterations=np.uint32(2048))
print("pairs", len(ref["ok"]), "ok", int(ref["ok"].sum()), "inliers", int(ref["mask"].sum()))
