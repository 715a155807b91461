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


def filter_scene(model, seed=5, n_cams=24, n_points=600, track_len=3):
    """A 'post-BA' scene for the track filters (sfm_data_filters.cpp:40-121): parameters at their ground truth, 0.5 px
    noise, 6 % gross outliers (residual filter), 12 % of the points far away so that their rays are nearly parallel (angle
    filter), observations shuffled, a few tracks cut down to two observations."""
    sc = synth.ba_scene(n_cams=n_cams, n_points=n_points, track_len=track_len, model=model, n_intr_groups=3, seed=seed,
                        outlier_frac=0.06, n_rings=1)
    rng = np.random.default_rng(seed + 1000)
    sc["poses"] = sc["poses_gt"].copy(); sc["intrinsics"] = sc["intrinsics_gt"].copy(); sc["points"] = sc["points_gt"].copy()
    far = rng.random(n_points) < 0.12
    # far points sit on the optical axis side of their first camera: all the cameras of the track see them in front
    first = np.full(n_points, -1, np.int64)
    for o in range(len(sc["obs_point"]) - 1, -1, -1):
        first[sc["obs_point"][o]] = sc["obs_pose"][o]
    for p in np.nonzero(far)[0]:
        pose = sc["poses"][first[p]]
        R = synth._rodrigues(pose[None, :3])[0]
        C = -R.T @ pose[3:]
        sc["points"][p] = C + R.T @ np.array([0.02, -0.01, 1.0]) * rng.uniform(300, 900)
    sel = far[sc["obs_point"]]
    xy = synth.project(model, sc["intrinsics"][sc["obs_intr"][sel]], sc["poses"][sc["obs_pose"][sel]], sc["points"][sc["obs_point"][sel]])
    sc["obs_xy"][sel] = xy + 0.2 * rng.standard_normal(xy.shape)
    # ragged tracks: drop the last observation of every 7th point
    last = np.concatenate([sc["obs_point"][1:] != sc["obs_point"][:-1], [True]])
    keep = ~(last & (sc["obs_point"] % 7 == 0))
    perm = rng.permutation(int(keep.sum()))
    for k in ("obs_pose", "obs_intr", "obs_point"):
        sc[k] = np.ascontiguousarray(sc[k][keep][perm])
    sc["obs_xy"] = np.ascontiguousarray(sc["obs_xy"][keep][perm])
    sc["n_obs"] = len(perm)
    return sc


def rejector_loop(adjust, rejector, scene, dPrecision=4.0, count=0, max_rounds=5):
    """The pipeline loop `do { BA } while (badTrackRejector(dPrecision, count))` (sequential_SfM.cpp:206-210) over the flat
    scene: adjust(scene) -> scene', rejector(scene', dPrecision, count) -> (again, scene''). Returns the final scene and the
    number of bundle adjustments run."""
    rounds = 0
    while True:
        scene = adjust(scene)
        rounds += 1
        again, scene = rejector(scene, dPrecision, count)
        if not again or rounds >= max_rounds:
            return scene, rounds
