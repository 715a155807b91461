import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from openmvg_amd import synth, geofilter
from tests import _oracle
n_pairs=2000
tv = synth.two_view_matches_bulk(n_pairs, n=250, seed=0x6E0F)
K = synth.two_view_calibration(tv)
st_ = tv["start"].astype(np.int64)
bI = np.zeros((len(tv["xI"]), 3)); bJ = np.zeros((len(tv["xJ"]), 3))
for p in range(n_pairs):
    bI[st_[p]:st_[p + 1]] = geofilter.pinhole_bearings(K[p, 0], tv["xI"][st_[p]:st_[p + 1]])
    bJ[st_[p]:st_[p + 1]] = geofilter.pinhole_bearings(K[p, 1], tv["xJ"][st_[p]:st_[p + 1]])
for up in (False, True):
    r = _oracle.ref_geofilter_angular(bI,bJ,tv["start"],upright=up)
    print("reference angular upright", up, "pairs", n_pairs, "seconds", round(r["seconds"],3), "pairs/s", round(n_pairs/r["seconds"],1), "ok", int(r["ok"].sum()))
