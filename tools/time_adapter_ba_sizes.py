"""Wall time of Bundle_Adjustment_Ceres::Adjust for the problem sizes the SfM engines send most often (initial pair
sequential_SfM.cpp:593-596, per-resection local BAs, SfM_Localizer.cpp:386-387, the global BA :1190-1215): the MI355X
replacement TU against the reference TU (vendored Ceres, default threads), same caller code
(oracle/ref_shim_ba.cpp::ref_ba_adjust, out_stats[2] = seconds inside Adjust). Prints one JSON line per size."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvg_amd import synth
from tests import _oracle

SIZES = [(2, 800, 2), (3, 1500, 3), (10, 4000, 4), (50, 20000, 6), (200, 100000, 10)]
reps = 4
for n_cams, n_pts, tl in SIZES:
    sc = synth.ba_scene(n_cams=n_cams, n_points=n_pts, track_len=tl, model=3, n_intr_groups=1, seed=0xAD1A + n_cams)
    ours = []
    for _ in range(reps):
        rc, st, *_ = _oracle.ref_ba_adjust(sc, lib=_oracle.adapter())
        ours.append(st[2] * 1e3)
    rec = {"views": n_cams, "points": n_pts, "observations": int(sc["n_obs"]), "replacement_ms": [round(x, 2) for x in ours],
           "replacement_rmse": float(st[1])}
    if _oracle.have_ref_ba() and "--no-ref" not in sys.argv:
        # the reference's default is one thread per hardware thread (BA_Ceres_options: omp_get_max_threads) - on a 256-thread
        # host that costs seconds on a tiny problem; its best thread count is reported beside it
        best = None
        for thr in (1, 16, 0):
            t = []
            for _ in range(2 if thr else 1):
                rc, st, *_ = _oracle.ref_ba_adjust(sc, num_threads=thr)
                t.append(st[2] * 1e3)
            rec[f"reference_ms_threads_{thr if thr else 'default'}"] = [round(x, 2) for x in t]
            if thr:
                best = min(t) if best is None else min(best, min(t))
        rec.update(reference_rmse=float(st[1]), reference_best_ms=round(best, 2), speedup_vs_reference_best=round(best / min(ours[1:]), 2))
    print(json.dumps(rec), flush=True)
