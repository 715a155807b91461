"""`do { BA } while (badTrackRejector)` (sequential_SfM.cpp:1190-1232) on ONE SfM_Data of 200 views / 100 000 tracks / 1 M observations with
1 % outlier observations: wall time of every Adjust() and of the two outlier filters per round, the replacement TUs (kept context:
re-bound, re-bound with observations switched off) against themselves with MVGX_BA_CONTEXT_CACHE=0 and against the reference TUs
(oracle/ref_shim_ba.cpp::ref_ba_reject_loop, same caller code). One JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import synth
from tests import _oracle

sc = synth.ba_scene(n_cams=200, n_points=100000, track_len=10, model=3, n_intr_groups=1, seed=0xAD1A + 200, outlier_frac=0.01)
a = _oracle.adapter()
_oracle.ref_ba_reject_loop(sc, lib=a)   # warm: slab caches, host workers
a.mvgx_adapter_ba_release_context()
ours = _oracle.ref_ba_reject_loop(sc, lib=a)
os.environ["MVGX_BA_CONTEXT_CACHE"] = "0"
plain = _oracle.ref_ba_reject_loop(sc, lib=a)
del os.environ["MVGX_BA_CONTEXT_CACHE"]
rec = {"views": 200, "tracks": int(sc["n_points"]), "observations": int(sc["n_obs"]), "rounds": ours["rounds"], "removed_per_round": ours["removed"].tolist(),
       "replacement_ms_per_round_adjust_residual_angle": np.round(ours["seconds"] * 1e3, 2).tolist(),
       "replacement_without_kept_context_ms": np.round(plain["seconds"] * 1e3, 2).tolist(),
       "replacement_loop_ms": round(float(ours["seconds"].sum() * 1e3), 2), "replacement_without_kept_context_loop_ms": round(float(plain["seconds"].sum() * 1e3), 2),
       "final_rmse": ours["rmse"], "same_result_without_kept_context": bool(np.array_equal(ours["keep"], plain["keep"]) and abs(ours["rmse"] - plain["rmse"]) < 1e-9)}
if _oracle.have_ref_ba() and "--no-ref" not in sys.argv:
    ref = _oracle.ref_ba_reject_loop(sc, num_threads=16)
    rec.update(reference_ms_per_round=np.round(ref["seconds"] * 1e3, 1).tolist(), reference_loop_ms=round(float(ref["seconds"].sum() * 1e3), 1),
               reference_rounds=ref["rounds"], same_observations_kept=bool(np.array_equal(ref["keep"], ours["keep"])),
               rmse_diff_vs_reference=abs(ref["rmse"] - ours["rmse"]))
print(json.dumps(rec), flush=True)
