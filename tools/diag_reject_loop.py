"""Diagnosis: the BA + outlier-rejection loop of tests/test_ba_gpu.py::test_bundle_then_reject_loop_equals_the_reference_pipeline with
both model-cost forms on the device and the reference at several thread counts: surviving observations per round."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import ba, synth
from tests import _ba_cases, _oracle

sc0 = synth.ba_scene(n_cams=24, n_points=1500, track_len=5, model=3, n_intr_groups=2, seed=91, outlier_frac=0.04, n_rings=1)


def loop(adjust, rejector):
    trace = []
    def adj(sc):
        out = adjust(sc)
        return out
    def rej(sc, prec, count):
        again, out = rejector(sc, prec, count)
        trace.append(int(out["n_obs"]))
        return again, out
    final, rounds = _ba_cases.rejector_loop(adj, rej, sc0)
    return final, rounds, trace


def ours_adjust(sc):
    sc = dict(sc)
    assert ba.Bundle_Adjustment_HIP().Adjust(sc)
    return sc


def make_ref(thr):
    def ref_adjust(sc):
        rc, st, poses, intr, pts = _oracle.ref_ba_adjust(sc, num_threads=thr)
        out = dict(sc); out["poses"] = poses; out["intrinsics"] = intr; out["points"] = pts
        return out
    return ref_adjust


def ref_rejector(sc, prec, count):
    keep, counts, _ = _oracle.ref_ba_filters(sc, prec, 2, 2.0)
    return sum(counts) > count, ba._drop_observations(sc, keep)


for form in ("normal", "jacobian"):
    os.environ["MVGX_BA_MODEL_COST"] = form
    for groups in ("1", "0"):
        os.environ["MVGX_BA_GROUPS"] = groups
        f, r, t = loop(ours_adjust, ba.badTrackRejector)
        print(f"device model-cost={form} groups={groups}: rounds {r}, surviving {t}", flush=True)
for thr in (1, 4, 16, 0):
    f, r, t = loop(make_ref(thr), ref_rejector)
    print(f"reference threads={thr or 'default'}: rounds {r}, surviving {t}", flush=True)
