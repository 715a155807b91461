"""End-to-end wall time of Bundle_Adjustment_Ceres::Adjust (SfM_Data in, SfM_Data out): the MI355X replacement TU vs the
reference TU (vendored Ceres), same caller code (oracle/ref_shim_ba.cpp::ref_ba_adjust, out_stats[2])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_ba
from openmvg_amd import synth
from tests import _oracle
sc = synth.ba_scene(**bench_ba.ba_config(1))
for rep in range(2):
    rc, st, *_ = _oracle.ref_ba_adjust(sc, lib=_oracle.adapter())
    print(f"replacement Adjust(): rc {rc}  {st[2]*1e3:.1f} ms  RMSE {st[0]:.4f} -> {st[1]:.6f}", flush=True)
if _oracle.have_ref_ba() and "--ref" in sys.argv:
    rc, st, *_ = _oracle.ref_ba_adjust(sc)
    print(f"reference   Adjust(): rc {rc}  {st[2]*1e3:.1f} ms  RMSE {st[0]:.4f} -> {st[1]:.6f}", flush=True)
