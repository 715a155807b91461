"""Generates tests/golden/ba_plateau_seed93_reference.json: the compiled reference (oracle/_ref) on the Huber-plateau scene with
1/2/4/8 threads and three observation orders - the spread of ITS OWN final RMSE, which bounds what parity can mean there."""
import sys, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np
from openmvg_amd import synth
from tests import _oracle
kw = dict(n_cams=10, n_points=400, track_len=5, model=1, n_intr_groups=1, seed=93, outlier_frac=0.05)
sc = synth.ba_scene(**kw)
out = []
for thr in (1, 2, 4, 8):
    for perm_seed in (None, 1, 2):
        s2 = dict(sc)
        if perm_seed is not None:   # the reference's containers are hash maps: the residual-block order is unspecified
            p = np.random.default_rng(perm_seed).permutation(sc["n_obs"])
            for k in ("obs_pose", "obs_intr", "obs_point"):
                s2[k] = sc[k][p]
            s2["obs_xy"] = sc["obs_xy"][p]
        rc, st, *_ = _oracle.ref_ba_adjust(s2, num_threads=thr)
        out.append({"threads": thr, "obs_order": perm_seed, "rc": rc, "initial_rmse": st[0], "final_rmse": st[1]})
        print(out[-1], flush=True)
rc, osum, *_ = _oracle.port_ba_solve(sc)
print("oracle", osum.num_iterations, osum.final_rmse, osum.final_cost)
r = [o["final_rmse"] for o in out]
print("spread", max(r) - min(r), (max(r) - min(r)) / np.mean(r))
json.dump({"scene": kw, "runs": out, "oracle_restatement": {"iterations": osum.num_iterations, "final_rmse": osum.final_rmse, "final_cost": osum.final_cost},
           "reference_spread_abs": max(r) - min(r)}, open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'ba_plateau_seed93_reference.json'), 'w'), indent=1)
