"""Essential filter on the PMC workload (20 000 pairs x 250): per-pair model counts and a digest of every output, written to a file -
to compare a run under rocprofv3 --pmc with a plain one (usage: geofilter_e_determinism.py out.npz)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import geofilter, synth
tv = synth.two_view_matches_bulk(20000, n=250, seed=0x6E0F)
K = synth.two_view_calibration(tv)
outs = []
for rep in range(2):
    mask, res, st = geofilter.filter_pairs_e(tv["xI"], tv["xJ"], tv["start"], tv["wh"], K, geofilter.GeometricFilter_EMatrix_AC(4.0, 2048))
    outs.append((mask.copy(), res.copy(), int(st.n_models), int(st.n_iterations)))
    print("rep", rep, "models", st.n_models, "iterations", st.n_iterations, "digest", hashlib.sha1(mask.tobytes() + res.tobytes()).hexdigest()[:16], flush=True)
print("two calls equal:", outs[0][1].tobytes() == outs[1][1].tobytes() and np.array_equal(outs[0][0], outs[1][0]))
np.savez(sys.argv[1], mask=outs[0][0], res=outs[0][1].view(np.uint8), models=outs[0][2])
if len(sys.argv) > 2:
    other = np.load(sys.argv[2])
    same = np.array_equal(other["mask"], outs[0][0]) and np.array_equal(other["res"], outs[0][1].view(np.uint8))
    print("equal to", sys.argv[2], ":", same, "models", int(other["models"]), outs[0][2])
    if not same:
        a = other["res"].view(outs[0][1].dtype); b = outs[0][1]
        diff = [p for p in range(len(b)) if a[p].tobytes() != b[p].tobytes()]
        print("pairs whose results differ:", len(diff), diff[:20])
