"""BA problems shared by the emulated (CPU) and the MI355X (GPU) parity tests."""
import numpy as np

from openmvg_amd import ba_options as bo
from openmvg_amd import synth


def edge_scenes():
    base = synth.ba_scene(n_cams=6, n_points=40, track_len=3, model=1, seed=3)
    out = []
    # a pose and a point that no observation references (Ceres drops unused blocks; ours keep a decoupled unit slot)
    sc = dict(base); sc["poses"] = np.vstack([base["poses"], base["poses"][:1]]); sc["n_poses"] = 7
    sc["points"] = np.vstack([base["points"], [[0.1, 0.2, 0.3]]]); sc["n_points"] = 41
    out.append(("unused_pose_and_point", sc, {}))
    # empty problem
    sc = dict(base)
    for k in ("obs_pose", "obs_intr", "obs_point"):
        sc[k] = base[k][:0]
    sc["obs_xy"] = base["obs_xy"][:0]; sc["n_obs"] = 0
    out.append(("no_observations", sc, {}))
    # nothing to optimise (sfm_data_BA_ceres.cpp: every block SetParameterBlockConstant)
    out.append(("all_constant", base, bo.masks_for(base, 1, 1, 0)))
    # one observation per point: V_p = EᵀE is rank 2, only the LM diagonal makes it invertible
    keep = np.concatenate([[True], base["obs_point"][1:] != base["obs_point"][:-1]])
    sc = dict(base)
    for k in ("obs_pose", "obs_intr", "obs_point"):
        sc[k] = base[k][keep]
    sc["obs_xy"] = base["obs_xy"][keep]; sc["n_obs"] = int(keep.sum())
    out.append(("single_observation_tracks", sc, {}))
    # a point far off every frustum: huge residuals, rejected steps, still the oracle's trajectory
    sc = dict(base); sc["points"] = base["points"].copy(); sc["points"][0] = [100.0, 0, 0]
    out.append(("wild_point", sc, {}))
    return out
