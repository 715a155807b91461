#!/usr/bin/env python
"""Generates tests/golden/geofilter_e.npz: synthetic calibrated two-view correspondences (openmvg_amd.synth.two_view_matches, fixed seed,
with the cameras' calibration matrices synth.two_view_calibration) and what the REFERENCE's own GeometricFilter_EMatrix_AC kernel
(ACKernelAdaptorEssential<FivePointSolver, EpipolarDistanceError> + ACRANSAC, E_ACRobust.hpp:57-150) returns for them
(oracle/_ref/libref_geofilter.so::ref_geofilter_e_acransac, compiled from /root/reference by oracle/Makefile), the bearing vectors the
reference's Pinhole_Intrinsic::operator() produced (the device call's input), and known answers of the five-point solver alone
(ref_five_point on random samples). Run in the build container: python tests/golden/make_geofilter_e_golden.py"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from openmvg_amd import synth
from tests import _oracle

tv = synth.two_view_matches(240, seed=5151, n_max=220)
K = synth.two_view_calibration(tv)
ref = _oracle.ref_geofilter_e(tv, K, precision=4.0, max_iterations=2048)
bI, bJ = _oracle.ref_pinhole_bearings(tv, K)
# five-point known answers: 64 random relative poses, five points in front of both cameras
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"))
rng = np.random.default_rng(99)
fp_b1, fp_b2, fp_E, fp_n = [], [], [], []
for _ in range(64):
    w = rng.normal(size=3); w *= rng.uniform(0.05, 0.5) / np.linalg.norm(w)
    th = np.linalg.norm(w); k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = rng.normal(size=3); t /= np.linalg.norm(t)
    X = rng.uniform(-1, 1, (5, 3)) + np.array([0, 0, 4.0])
    b1 = np.ascontiguousarray(X / np.linalg.norm(X, axis=1, keepdims=True))
    X2 = X @ R.T + t
    b2 = np.ascontiguousarray(X2 / np.linalg.norm(X2, axis=1, keepdims=True))
    Es = np.zeros(90); n = C.c_int(0)
    lib.ref_five_point(b1.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p), Es.ctypes.data_as(C.c_void_p), C.byref(n))
    fp_b1.append(b1); fp_b2.append(b2); fp_E.append(Es.reshape(10, 9)); fp_n.append(n.value)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "geofilter_e.npz"), xI=tv["xI"], xJ=tv["xJ"], start=tv["start"], wh=tv["wh"], K=K, bI=bI, bJ=bJ,
                    mask=ref["mask"], ok=ref["ok"], F=ref["F"], precision=ref["precision"], nfa=ref["nfa"], precision_px=4.0, max_iterations=2048,
                    fp_b1=np.asarray(fp_b1), fp_b2=np.asarray(fp_b2), fp_E=np.asarray(fp_E), fp_n=np.asarray(fp_n, np.int32))
print("pairs", len(tv["start"]) - 1, "ok", int(ref["ok"].sum()), "inliers", int(ref["mask"].sum()), "of", len(ref["mask"]), "five-point solutions", fp_n)
