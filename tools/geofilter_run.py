"""One pass of the geometric filter over a synthetic workload (the command profiled by the rocprofv3 passes of the kernel).
Usage: geofilter_run.py [n_pairs] [n_matches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvg_amd import geofilter, synth
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 250
tv = synth.two_view_matches_bulk(n_pairs, n=n, seed=0x6E0F)
mask, res, st = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], geofilter.GeometricFilter_FMatrix_AC(4.0, 2048))
print("pairs", n_pairs, "kernel_ms", st.kernel_ms, "ok", int(st.n_pairs_ok))
