"""Generates tests/golden/ba_golden.npz by running the REFERENCE's Bundle_Adjustment_Ceres::Adjust (vendored Ceres 1.13,
compiled in place into oracle/_ref/libref_ba.so) on small fixed scenes. Run in the build container:

    python tests/golden/make_ba_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from openmvg_amd import ba_options as bo  # noqa: E402
from openmvg_amd import synth  # noqa: E402
from tests import _oracle  # noqa: E402
from tests.test_oracle_ba import SCENES  # noqa: E402

KEYS = ("poses", "intrinsics", "intr_model", "points", "obs_pose", "obs_intr", "obs_point", "obs_xy")
CASES = [(name, 14, 6, 1) for name in sorted(SCENES)] + [
    ("ring_k3_groups", int(bo.Intrinsic_Parameter_Type.ADJUST_FOCAL_LENGTH), 6, 1),
    ("ring_k3_groups", 14, int(bo.Extrinsic_Parameter_Type.ADJUST_ROTATION), 1),
    ("ring_k3_groups", 14, int(bo.Extrinsic_Parameter_Type.ADJUST_TRANSLATION), 1),
    ("ring_k3_groups", 14, 6, 0),
    ("ring_pinhole", 1, 6, 1),
]


# scenes beyond plain reprojection residuals: the other camera functors, ground control points, pose-centre priors
EX_KEYS = KEYS + ("obs_weight", "obs_is_control", "point_const_mask", "prior_pose", "prior_center", "prior_weight")


def ex_scenes():
    base = dict(n_cams=10, n_points=150, track_len=5, n_intr_groups=2, rot_deg=0.3)
    yield "brown", 14, synth.ba_scene(model=4, seed=104, **base)
    yield "fisheye", 14, synth.ba_scene(model=5, seed=105, **base)
    yield "spherical", 14, synth.ba_scene(model=7, seed=107, **base)
    yield "gcp_pinhole_fixed_intrinsics", 1, synth.add_control_points(synth.ba_scene(model=1, seed=108, **base), n_ctrl=6, weight=20.0)
    yield "gcp_k3", 14, synth.add_control_points(synth.ba_scene(model=3, seed=109, **base), n_ctrl=8, weight=20.0)
    yield "priors_pinhole", 14, synth.add_pose_priors(synth.ba_scene(model=1, seed=110, n_cams=12, n_points=150, track_len=6, rot_deg=0.3), sigma=0.01)


def main():
    out = {"case_names": np.array([f"{n}|{i}|{e}|{s}" for n, i, e, s in CASES])}
    ex_names = []
    for name, iopt, sc in ex_scenes():
        ex_names.append(f"{name}|{iopt}")
        tag = f"ex/{name}|{iopt}"
        rc, stats, rp, ri, rx = _oracle.ref_ba_adjust_ex(sc, intrinsics_opt=iopt)
        assert rc == 0 and stats[1] < stats[0]
        for k in EX_KEYS:
            if sc.get(k) is not None:
                out[f"{tag}/{k}"] = np.asarray(sc[k])
        out[f"{tag}/meta"] = np.array([sc.get("n_structure_points", sc["n_points"]), sc.get("control_weight", 0.0)], np.float64)
        out[f"{tag}/ref_stats"] = stats; out[f"{tag}/ref_poses"] = rp; out[f"{tag}/ref_intrinsics"] = ri; out[f"{tag}/ref_points"] = rx
        if sc.get("prior_pose") is not None:   # the problem the reference actually solves (after its registration step)
            usable, prep, centroid = _oracle.ref_ba_prior_prepare(sc)
            assert usable
            out[f"{tag}/prep_poses"] = prep["poses"]; out[f"{tag}/prep_points"] = prep["points"]
            out[f"{tag}/prep_prior_center"] = prep["prior_center"]
            out[f"{tag}/prep_meta"] = np.concatenate([[prep["prior_huber_a"]], centroid])
        print(tag, "rmse", stats[0], "->", stats[1])
    out["ex_case_names"] = np.array(ex_names)
    for name, iopt, eopt, sopt in CASES:
        tag = f"{name}|{iopt}|{eopt}|{sopt}"
        sc = synth.ba_scene(**SCENES[name])
        rc, stats, rp, ri, rx = _oracle.ref_ba_adjust(sc, intrinsics_opt=iopt, extrinsics_opt=eopt, structure_opt=sopt)
        assert rc == 0
        for k in KEYS:
            out[f"{tag}/{k}"] = sc[k]
        out[f"{tag}/ref_stats"] = stats
        out[f"{tag}/ref_poses"] = rp
        out[f"{tag}/ref_intrinsics"] = ri
        out[f"{tag}/ref_points"] = rx
        print(tag, "rmse", stats[0], "->", stats[1])
    path = os.path.join(ROOT, "tests", "golden", "ba_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
