"""End to end: putative matching of an image collection (exhaustive pairs, L2 + ratio test) followed by the geometric filter
(fundamental matrix, AC-RANSAC) of every pair that produced matches - Matcher_Regions::Match + Robust_model_estimation of
main_ComputeMatches / main_GeometricFilter on one synthetic collection whose neighbouring images share landmarks.
Usage: time_match_filter.py [n_images] [n_desc]   -> one JSON line"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import geofilter, matching, synth

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_desc = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
t = time.perf_counter()
S = synth.sfm_image_set(n_images, n_desc)
t_gen = time.perf_counter() - t
pairs = matching.exhaustive_pairs_array(n_images)
rec = {"n_images": n_images, "n_desc": n_desc, "n_pairs": int(len(pairs)), "scene_s": t_gen}
for rep in range(2):   # the second pass is the warm one (cached slabs, streams)
    t0 = time.perf_counter()
    ctx = matching.MatchContext()
    ctx.set_regions(S["desc"])
    t1 = time.perf_counter()
    st, off, ij = ctx.run(pairs, 0.8 * 0.8)
    t2 = time.perf_counter()
    ctx.close()
    # the container as the matcher returned it goes to the filter: pairs with matches, their index pairs, the feature positions
    # of every image (MatchesPairToMat runs on the device)
    cnt = np.diff(off.astype(np.int64))
    live = np.flatnonzero(cnt > 0)
    pair_of = np.repeat(np.arange(len(pairs)), cnt)
    start = np.concatenate([[0], np.cumsum(cnt[live])]).astype(np.uint64)
    sizes = np.tile(np.array(S["size"], np.uint32), (n_images, 1))
    t3 = time.perf_counter()
    mask, res, gst = geofilter.filter_pairs_indexed(S["xy"], sizes, pairs[live], start, ij)
    t4 = time.perf_counter()
    lm = np.stack(S["landmark"])
    true = (lm[pairs[pair_of, 0], ij[:, 0]] == lm[pairs[pair_of, 1], ij[:, 1]]) & (lm[pairs[pair_of, 0], ij[:, 0]] >= 0)
    rec[f"pass{rep}"] = {
        "set_regions_s": t1 - t0, "match_s": t2 - t1, "container_bookkeeping_s": t3 - t2, "geometric_filter_s": t4 - t3,
        "match_plus_filter_s": (t2 - t0) + (t4 - t2), "filter_kernel_ms": gst.kernel_ms,
        "putative_matches": int(len(ij)), "pairs_with_matches": int(len(live)), "pairs_estimated_ok": int(res["ok"].sum()),
        "putative_true_fraction": float(true.mean()) if len(ij) else 0.0,
        "inliers": int(mask.sum()), "inliers_true_fraction": float(true[mask].mean()) if mask.any() else 0.0,
        "true_matches_kept_fraction": float(mask[true].mean()) if true.any() else 0.0}
print(json.dumps(rec))
