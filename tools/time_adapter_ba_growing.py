"""Bundle adjustment of a scene that GROWS between two Adjust() calls - resection of a view and triangulation of its tracks, then BA
(sequential_SfM.cpp:206-210) - at 200 views / 100 000 tracks / 1 M observations: the last view, its ~5 000 observations and 2 000 of its
tracks join the SfM_Data between the calls. Wall time of either Adjust() through the replacement TU (the kept context of call 1 does not fit
the grown scene: call 2 rebuilds it - MVGX_ADAPTER_TIMING=1 prints its phases) and through the reference TU (Ceres, 16 threads), same caller
code (oracle/ref_shim_ba.cpp::ref_ba_adjust_growing). One JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench_ba
from openmvg_amd import synth
from tests import _oracle

sc = synth.ba_scene(**bench_ba.ba_config(1))
a = _oracle.adapter()
_oracle.ref_ba_adjust_growing(sc, 2000, lib=a)   # warm: slab caches, host workers
runs = []
for rep in range(3):
    a.mvgx_adapter_ba_release_context()
    runs.append(_oracle.ref_ba_adjust_growing(sc, 2000, lib=a))
ours = runs[-1]
rec = {"views": int(sc["n_poses"]), "tracks_call_1_2": [int(ours["counts"][1]), int(ours["counts"][3])], "observations_call_1_2": [int(ours["counts"][0]), int(ours["counts"][2])],
       "replacement_adjust_ms_call_1_2": [np.round(r["seconds"] * 1e3, 2).tolist() for r in runs], "replacement_rmse_before_after1_after2": ours["rmse"].tolist()}
if _oracle.have_ref_ba() and "--no-ref" not in sys.argv:
    ref = _oracle.ref_ba_adjust_growing(sc, 2000, num_threads=16)
    rec.update(reference_adjust_ms_call_1_2=np.round(ref["seconds"] * 1e3, 1).tolist(), reference_rmse_before_after1_after2=ref["rmse"].tolist(),
               rmse_diff_vs_reference=float(np.abs(ref["rmse"] - ours["rmse"]).max()),
               speedup_call_2=round(float(ref["seconds"][1] / ours["seconds"][1]), 1))
print(json.dumps(rec), flush=True)
