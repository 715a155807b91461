"""Generates tests/golden/sceaux_geofilter.npz: the reference's geometric filter on REAL putative matches.

Input: tests/golden/sceaux_sift.npz - the reference's own SIFT regions of the two SceauxCastle JPGs and the reference's own
Matcher_Regions lists between them (both directions, ratio 0.8 and 0.6: four containers entries with 1 038 - 1 421 putative matches
each; real matches bring the planar facade / repeated windows a synthetic generator does not).
Output per entry and model (F: GeometricFilter_FMatrix_AC(4.0, 2048), H: GeometricFilter_HMatrix_AC(4.0, 2048) - the settings of
main_GeometricFilter): the reference's inlier mask, ok flag, model, precision and NFA from oracle/_ref/libref_geofilter.so (g++ -O3),
the geometric-match container the reference's ImageCollectionGeometricFilter template produces (ref_geofilter_container{,_h}), and -
as the reference's own spread - whether the -O3 -mavx2 -mfma build of the same sources (make -C oracle ref_geofilter_fma) ends with
the same inlier set. Runs in the build container only (needs oracle/_ref); the fixture travels.
  python tests/golden/make_sceaux_geofilter_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import _oracle  # noqa: E402

ENTRIES = [(80, 0, 1), (80, 1, 0), (60, 0, 1), (60, 1, 0)]   # (ratio x 100, I, J) of sceaux_sift.npz


def two_view(z):
    """the four entries as gathered correspondences (what MatchesPairToMat returns: feature positions as doubles)"""
    xI, xJ, start, wh = [], [], [0], []
    for r, i, j in ENTRIES:
        m = z[f"matches_r{r}_{i}_{j}"]
        xI.append(z[f"feat{i}"][m[:, 0], :2].astype(np.float64)); xJ.append(z[f"feat{j}"][m[:, 1], :2].astype(np.float64))
        start.append(start[-1] + len(m)); wh.append([*z[f"size{i}"], *z[f"size{j}"]])
    return dict(xI=np.concatenate(xI), xJ=np.concatenate(xJ), start=np.asarray(start, np.uint64), wh=np.asarray(wh, np.uint32))


def main():
    z = np.load(os.path.join(HERE, "sceaux_sift.npz"))
    tv = two_view(z)
    fma_path = os.path.join(ROOT, "oracle", "_ref", "libref_geofilter_fma.so")
    fma = C.CDLL(fma_path) if os.path.exists(fma_path) else None
    out = {"entries": np.asarray(ENTRIES, np.int32), "precision_px": np.float64(4.0), "max_iterations": np.int32(2048), "start": tv["start"]}
    for model, ref_fn, name in (("f", _oracle.ref_geofilter, "ref_geofilter_f_acransac"), ("h", _oracle.ref_geofilter_h, "ref_geofilter_h_acransac")):
        ref = ref_fn(tv)
        for k in ("mask", "ok", "F", "precision", "nfa"):
            out[f"{model}_{k}"] = ref[k]
        print(model, "ok", ref["ok"], "inliers", [int(ref["mask"][tv["start"][p]:tv["start"][p + 1]].sum()) for p in range(4)], "nfa", ref["nfa"])
        if fma is not None:
            other = _oracle._geofilter_call(getattr(fma, name), tv, 4.0, 2048, 0)
            same = [bool(other["ok"][p] == ref["ok"][p] and np.array_equal(other["mask"][tv["start"][p]:tv["start"][p + 1]], ref["mask"][tv["start"][p]:tv["start"][p + 1]]))
                    for p in range(4)]
            out[f"{model}_same_inlier_set_in_the_avx2_fma_build"] = np.asarray(same)
            print(model, "same inlier set in the -mavx2 -mfma build of the reference:", same)
    # container level: the two images as a collection, the ratio-0.8 lists of both directions as (0, 1) ... the reference's container is
    # keyed by (I, J) with I < J in practice; the direction (1, 0) is stored as a second collection
    for model in ("f", "h"):
        for r in (80, 60):
            feats = [z["feat0"][:, :2], z["feat1"][:, :2]]
            sizes = np.asarray([z["size0"], z["size1"]], np.uint32)
            got = _oracle.geofilter_container("reference", feats, sizes, {(0, 1): z[f"matches_r{r}_0_1"]}, model=model)
            out[f"container_{model}_r{r}"] = got.get((0, 1), np.zeros((0, 2), np.uint32))
            print("container", model, r, len(out[f"container_{model}_r{r}"]), "geometric matches")
    np.savez_compressed(os.path.join(HERE, "sceaux_geofilter.npz"), **out)


if __name__ == "__main__":
    main()
