"""WHICH pairs end with another inlier set on the device than with the compiled reference, and why. For the bench sets of the
fundamental (f) and essential (e) functors: device and reference (oracle/_ref) on the first m pairs; for every differing pair the two results side by
side and a classification from the residuals of the pair's correspondences under BOTH final models (numpy, EpipolarDistanceError):
  * "edge":      the two models agree (normalised difference < 1e-7) and the inlier sets differ only in correspondences whose residual lies within
                 1e-9 (relative) of the reported bound - a residual on the edge of the threshold, decided by the last bit of the model
  * "trajectory": the final models differ - an earlier a-contrario decision went the other way (a better-NFA model found or not found); the tool
                 reports whether the reference's model, evaluated by the device's rule, would have had a better NFA than the device's (in which
                 case the device never saw that model: its solver gave another root set for that sample = conditioning) or a worse one (a tie /
                 rounding in the NFA comparison itself)
Usage: geofilter_differing_pairs.py [f|e] [pairs]   -> one JSON line per differing pair + a summary line"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import geofilter, synth
from tests import _geofilter_cases as gc, _oracle

model = sys.argv[1] if len(sys.argv) > 1 else "f"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
n = 250
tv = synth.two_view_matches_bulk(m, n=n, seed=0x6E0F)
K = synth.two_view_calibration(tv) if model == "e" else None
if model == "e":
    fun = geofilter.GeometricFilter_EMatrix_AC(4.0, 2048)
    mask, res, st = geofilter.filter_pairs_e(tv["xI"], tv["xJ"], tv["start"], tv["wh"], K, fun)
    ref = _oracle.ref_geofilter_e(tv, K)
else:
    fun = geofilter.GeometricFilter_FMatrix_AC(4.0, 2048)
    mask, res, st = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], fun)
    ref = _oracle.ref_geofilter(tv)
differing, rep = gc.compare(tv["start"], ref, mask, res["ok"], res["F"], res["precision_robust"], res["nfa"])
start = tv["start"].astype(np.int64)


def pixel_F(p, M):
    """the matrix whose epipolar error is taken in pixels: F itself, or K2^-T E K1^-1"""
    M = np.asarray(M, np.float64).reshape(3, 3)
    if model != "e":
        return M
    K1, K2 = np.asarray(K[p, 0], np.float64).reshape(3, 3), np.asarray(K[p, 1], np.float64).reshape(3, 3)
    return np.linalg.inv(K2).T @ M @ np.linalg.inv(K1)


def residuals(F, x, y):
    Fx = np.c_[x, np.ones(len(x))] @ F.T
    dt = (Fx * np.c_[y, np.ones(len(y))]).sum(1)
    return dt * dt / (Fx[:, 0] ** 2 + Fx[:, 1] ** 2)


out = []
for p in differing:
    lo, hi = start[p], start[p + 1]
    x, y = tv["xI"][lo:hi], tv["xJ"][lo:hi]
    mr, md = np.asarray(ref["mask"][lo:hi], bool), np.asarray(mask[lo:hi], bool)
    rec = {"model": model, "pair": int(p), "matches": int(hi - lo), "ok_reference": bool(ref["ok"][p]), "ok_device": bool(res["ok"][p]),
           "inliers_reference": int(mr.sum()), "inliers_device": int(md.sum()), "symmetric_difference": int((mr != md).sum()),
           "nfa_reference": float(ref["nfa"][p]), "nfa_device": float(res["nfa"][p]),
           "precision_reference": float(ref["precision"][p]), "precision_device": float(res["precision_robust"][p])}
    if rec["ok_reference"] and rec["ok_device"]:
        Fr, Fd = gc.normalised(ref["F"][p])[0], gc.normalised(res["F"][p])[0]
        rec["model_difference"] = float(np.abs(Fr - Fd).max())
        rr, rd = residuals(pixel_F(p, ref["F"][p]), x, y), residuals(pixel_F(p, res["F"][p]), x, y)
        # the bound the functor stores: precision (pixels) for F; the essential functor's is squared pixels (E_ACRobust.hpp)
        br = rec["precision_reference"] ** 2 if model != "e" else rec["precision_reference"]
        flips = np.flatnonzero(mr != md)
        rec["relative_distance_of_the_flipped_residuals_to_the_reference_bound"] = [float(abs(rr[i] - br) / br) for i in flips[:6]]
        same_model = rec["model_difference"] < 1e-7
        rec["class"] = ("edge (same model, residual on the bound)" if same_model else
                        "trajectory (another model won: nfa reference %.6f, device %.6f)" % (rec["nfa_reference"], rec["nfa_device"]))
        rec["device_model_is_better_by_nfa"] = bool(rec["nfa_device"] < rec["nfa_reference"])
    else:
        rec["class"] = "acceptance (one side found no model above the inlier count)"
    out.append(rec)
    print(json.dumps(rec), flush=True)
print(json.dumps({"model": model, "pairs": m, "pairs_differing": len(differing), "allowed_by_the_reference_spread": gc.allowed_differing(m, model),
                  "classes": {c: sum(1 for r in out if r["class"].startswith(c)) for c in ("edge", "trajectory", "acceptance")},
                  "device_better_nfa": sum(1 for r in out if r.get("device_model_is_better_by_nfa")),
                  "reference_better_nfa": sum(1 for r in out if r.get("device_model_is_better_by_nfa") is False)}), flush=True)
