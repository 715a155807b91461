"""One pass of the geometric filter over a synthetic workload (the command profiled by the rocprofv3 passes of the kernel).
Usage: geofilter_run.py [n_pairs] [n_matches] [f|h|e|a|u]   (a / u: the angular essential models, eight-point / three-point upright)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvg_amd import geofilter, synth
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 250
model = sys.argv[3] if len(sys.argv) > 3 else "f"
if model == "h":
    tv = synth.two_view_homography_matches(n_pairs, seed=0x6E0F, n_min=n, n_max=n, tiny_frac=0.0)
    fun = geofilter.GeometricFilter_HMatrix_AC(4.0, 2048)
else:
    tv = synth.two_view_matches_bulk(n_pairs, n=n, seed=0x6E0F)
    fun = geofilter.GeometricFilter_FMatrix_AC(4.0, 2048)
if model in ("a", "u"):
    import numpy as np
    K = synth.two_view_calibration(tv)
    st_ = tv["start"].astype(np.int64)
    bI = np.zeros((len(tv["xI"]), 3)); bJ = np.zeros((len(tv["xJ"]), 3))
    for p in range(n_pairs):
        bI[st_[p]:st_[p + 1]] = geofilter.pinhole_bearings(K[p, 0], tv["xI"][st_[p]:st_[p + 1]])
        bJ[st_[p]:st_[p + 1]] = geofilter.pinhole_bearings(K[p, 1], tv["xJ"][st_[p]:st_[p + 1]])
    mask, res, st = geofilter.filter_pairs_angular(bI, bJ, tv["start"], geofilter.GeometricFilter_ESphericalMatrix_AC_Angular(4.0, 2048, model == "u"))
elif model == "e":
    mask, res, st = geofilter.filter_pairs_e(tv["xI"], tv["xJ"], tv["start"], tv["wh"], synth.two_view_calibration(tv), geofilter.GeometricFilter_EMatrix_AC(4.0, 2048))
else:
    mask, res, st = geofilter.filter_pairs(tv["xI"], tv["xJ"], tv["start"], tv["wh"], fun)
print("model", model, "pairs", n_pairs, "kernel_ms", st.kernel_ms, "ok", int(st.n_pairs_ok), "iterations", int(st.n_iterations), "models", int(st.n_models),
      "wave_clocks", int(st.wave_clocks))
# a measurement build of the library (-DMVGX_GEO_STAMPS, MVGX_LIB_PATH): shader clocks per stage of an a-contrario iteration
from openmvg_amd import _capi
if hasattr(_capi.lib(), "mvgx_debug_geo_stamps"):
    import ctypes as C
    out = (C.c_ulonglong * 8)()
    _capi.lib().mvgx_debug_geo_stamps(out, 1)
    names = ["sampling", "minimal solver", "residuals + histogram", "NFA over the bins", "inliers of a better model", "loop control, pool"]
    tot = float(sum(out[:6])) or 1.0
    print("stage clocks per iteration:", {nm: round(out[k] / max(int(st.n_iterations), 1)) for k, nm in enumerate(names)},
          "shares:", {nm: round(out[k] / tot, 3) for k, nm in enumerate(names)})
