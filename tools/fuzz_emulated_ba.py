import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import ba, synth
from openmvg_amd import ba_options as bo
from tests import _emu, _oracle
rng = np.random.default_rng(11)
t0=time.time(); n=0; bad=0
while time.time()-t0 < 300:
    model = int(rng.choice([1,2,3,4,5,7]))
    n_cams = int(rng.integers(3,14)); n_pts = int(rng.integers(5,90)); tl = int(rng.integers(2,6))
    groups = int(rng.integers(1, min(3,n_cams)+1))
    seed = int(rng.integers(1<<30))
    sc = synth.ba_scene(n_cams=n_cams, n_points=n_pts, track_len=tl, model=model, n_intr_groups=groups, seed=seed,
                        outlier_frac=float(rng.choice([0,0,0.1])), n_rings=1)
    extra = int(rng.integers(0,4))
    if extra == 1: sc = synth.add_control_points(sc, n_ctrl=3, views_per_point=min(3,n_cams), seed=seed&0xffff)
    if extra == 2 and n_cams >= 4: sc = synth.add_pose_priors(sc, seed=seed&0xffff) if hasattr(synth,'add_pose_priors') else sc
    iopt = int(rng.choice([14, 1, 2, 6, 8])) if model != 7 else 14
    eopt = int(rng.choice([6, 2, 4, 1]))
    sopt = int(rng.choice([1, 1, 0]))
    masks = bo.masks_for(sc, iopt, eopt, sopt)
    try:
        rc, osum, *_ = _oracle.port_ba_solve(sc, **masks)
        with _emu.emulated():
            ctx = ba.BaContext(sc, **masks); s = ctx.solve(); ctx.close()
    except Exception as e:
        print("EXC", model, n_cams, n_pts, tl, groups, seed, extra, iopt, eopt, sopt, repr(e)[:200], flush=True); bad += 1; n += 1; continue
    n += 1
    same = (s.num_iterations, s.termination) == (osum.num_iterations, osum.termination) and abs(s.final_cost-osum.final_cost) <= 1e-7*max(osum.final_cost,1e-12)+1e-15
    if not same:
        # iteration counts may legitimately differ by rounding at a plateau; flag only cost disagreements
        rel = abs(s.final_cost-osum.final_cost)/max(osum.final_cost,1e-12)
        tag = "DIFF" if rel > 1e-6 else "iters"
        if tag == "DIFF": bad += 1
        print(tag, model, n_cams, n_pts, tl, groups, seed, extra, iopt, eopt, sopt, (s.num_iterations, s.termination, s.final_cost), (osum.num_iterations, osum.termination, osum.final_cost), flush=True)
print("cases", n, "bad", bad)
