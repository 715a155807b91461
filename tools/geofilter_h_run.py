"""The homography model of the geometric filter on a synthetic workload against the compiled reference on a sample of it:
throughput of both, pairs that differ. Usage: geofilter_h_run.py [n_pairs] [n_matches_max] [ref_sample]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import geofilter, synth
from tests import _geofilter_cases as gc, _oracle
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_max = int(sys.argv[2]) if len(sys.argv) > 2 else 250
n_ref = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
tv = synth.two_view_homography_matches(n_pairs, seed=0x6E0F, n_min=n_max, n_max=n_max, tiny_frac=0.0)
f = geofilter.GeometricFilter_HMatrix_AC(4.0, 2048)
geofilter.filter_pairs(tv["xI"][:10 * n_max], tv["xJ"][:10 * n_max], tv["start"][:11], tv["wh"][:10], f)   # kernels loaded
t0 = time.perf_counter(); mask, res, st = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], f); dt = time.perf_counter() - t0
out = {"model": "homography (GeometricFilter_HMatrix_AC, 4 px, 2048 iterations)", "pairs": n_pairs, "matches_per_pair": n_max, "kernel_ms": st.kernel_ms,
       "call_s": dt, "pairs_per_s_kernel": n_pairs / (st.kernel_ms * 1e-3), "pairs_ok": int(st.n_pairs_ok),
       "true_matches_kept": float((mask & tv["is_inlier"]).sum() / max(1, (tv["is_inlier"] & np.repeat(res["ok"], n_max)).sum())),
       "false_inliers": int((mask & ~tv["is_inlier"]).sum())}
if _oracle.have_ref_geofilter() and n_ref:
    sub = dict(xI=tv["xI"][:n_ref * n_max], xJ=tv["xJ"][:n_ref * n_max], start=tv["start"][:n_ref + 1], wh=tv["wh"][:n_ref])
    ref = _oracle.ref_geofilter_h(sub, 4.0, 2048)
    differing, rep = gc.compare(sub["start"], ref, mask[:n_ref * n_max], res["ok"][:n_ref], res["F"][:n_ref], res["precision_robust"][:n_ref], res["nfa"][:n_ref])
    out["reference"] = {"pairs": n_ref, "seconds": ref["seconds"], "pairs_per_s": n_ref / ref["seconds"], "threads": os.cpu_count(), **rep}
print(json.dumps(out))
