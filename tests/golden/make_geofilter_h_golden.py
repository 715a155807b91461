#!/usr/bin/env python
"""Generates tests/golden/geofilter_h.npz: synthetic two-view correspondences related by homographies
(openmvg_amd.synth.two_view_homography_matches, fixed seed) and what the REFERENCE's own GeometricFilter_HMatrix_AC kernel
(ACKernelAdaptor<FourPointSolver, AsymmetricError, UnnormalizerI>, point-to-point) + ACRANSAC return for them
(oracle/_ref/libref_geofilter.so::ref_geofilter_h_acransac, compiled from /root/reference by oracle/Makefile).
Run in the build container: python tests/golden/make_geofilter_h_golden.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openmvg_amd import synth
from tests import _oracle

tv = synth.two_view_homography_matches(240, seed=4242, n_max=220)
ref = _oracle.ref_geofilter_h(tv, precision=4.0, max_iterations=2048)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "geofilter_h.npz"), xI=tv["xI"], xJ=tv["xJ"], start=tv["start"], wh=tv["wh"],
                    mask=ref["mask"], ok=ref["ok"], F=ref["F"], precision=ref["precision"], nfa=ref["nfa"], precision_px=4.0, max_iterations=2048)
print("pairs", len(tv["start"]) - 1, "ok", int(ref["ok"].sum()), "inliers", int(ref["mask"].sum()), "of", len(ref["mask"]))
