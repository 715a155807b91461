"""CPU test of the BA adapter's C++ code (openmvg_amd/adapter/mvgx_bundle_adjustment{,_ceres}.cpp: SfM_Data -> mvgx_ba_problem,
Optimize_Options -> masks, control points, the prior registration through the reference's own Similarity3 / LMedS code,
write-back rules): the replacement Bundle_Adjustment_Ceres, driven by the reference's caller code (oracle/ref_shim_ba.cpp),
linked against the HIP emulation instead of libmvgx_hip.so; compared with the outputs of the reference stored in
tests/golden/ba_golden.npz. Same entry points as tests/test_adapter_gpu.py; needs the openMVG tree (build container)."""
import os

import numpy as np
import pytest

from tests import _oracle

pytestmark = pytest.mark.skipif(_oracle.adapter_ba_emu() is None, reason="openMVG tree / adapter objects not present")


def _golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba_golden.npz"))


@pytest.mark.parametrize("tag", ["tiny_pinhole|14|6|1", "tiny_k1|14|6|1", "tiny_k3|14|6|1", "ring_outliers|14|6|1", "ring_pinhole|1|6|1"])
def test_bundle_adjustment_ceres_replacement_equals_golden(tag):
    z = _golden()
    keys = ("poses", "intrinsics", "intr_model", "points", "obs_pose", "obs_intr", "obs_point", "obs_xy")
    sc = {k: z[f"{tag}/{k}"].copy() for k in keys}
    sc["n_poses"] = len(sc["poses"]); sc["n_intrinsics"] = len(sc["intrinsics"]); sc["n_points"] = len(sc["points"])
    sc["n_obs"] = len(sc["obs_pose"])
    _, iopt, eopt, sopt = tag.split("|")
    ref_stats = z[f"{tag}/ref_stats"]
    rc, stats, poses, intr, pts = _oracle.ref_ba_adjust(sc, int(iopt), int(eopt), int(sopt), lib=_oracle.adapter_ba_emu())
    assert rc == 0 and stats[3] == ref_stats[3] == 1.0
    assert abs(stats[0] - ref_stats[0]) < 1e-9
    assert abs(stats[1] - ref_stats[1]) < 1e-6 * max(1.0, ref_stats[1]), (stats[1], ref_stats[1])
    if ref_stats[1] < 100:
        assert np.allclose(pts, z[f"{tag}/ref_points"], atol=1e-4)


@pytest.mark.parametrize("name", list(_golden()["ex_case_names"]))
def test_functors_control_points_priors(name):
    from tests.test_ba_gpu import _ex_case
    z = _golden()
    tag, sc = _ex_case(z, name)
    iopt = int(name.split("|")[1])
    ref_stats = z[f"{tag}/ref_stats"]
    rc, stats, poses, intr, pts = _oracle.ref_ba_adjust_ex(sc, intrinsics_opt=iopt, lib=_oracle.adapter_ba_emu())
    assert rc == 0 and stats[3] == ref_stats[3] == 1.0
    assert abs(stats[0] - ref_stats[0]) < 1e-9
    assert abs(stats[1] - ref_stats[1]) < 1e-6, (stats[1], ref_stats[1])
    assert np.allclose(pts, z[f"{tag}/ref_points"], atol=1e-5)
    assert np.allclose(poses[:, 3:], z[f"{tag}/ref_poses"][:, 3:], atol=1e-5)


def test_injected_device_failure_makes_adjust_return_false(monkeypatch, capfd):
    """error convention of the boundary: Adjust() reports a failing device through its return value, like the reference
    reports an unusable solution (sfm_data_BA_ceres.cpp:503-507) - no exception, poses / intrinsics untouched"""
    z = _golden()
    tag = "tiny_pinhole|14|6|1"
    keys = ("poses", "intrinsics", "intr_model", "points", "obs_pose", "obs_intr", "obs_point", "obs_xy")
    sc = {k: z[f"{tag}/{k}"].copy() for k in keys}
    sc["n_poses"] = len(sc["poses"]); sc["n_intrinsics"] = len(sc["intrinsics"]); sc["n_points"] = len(sc["points"]); sc["n_obs"] = len(sc["obs_pose"])
    monkeypatch.setenv("MVGX_ADAPTER_INJECT_FAILURE", "ba:create")
    rc, stats, poses, intr, pts = _oracle.ref_ba_adjust(sc, 14, 6, 1, lib=_oracle.adapter_ba_emu())
    monkeypatch.delenv("MVGX_ADAPTER_INJECT_FAILURE")
    assert stats[3] == 0.0   # Adjust returned false
    assert np.array_equal(intr, sc["intrinsics"])
    assert "Adjust() returns false" in capfd.readouterr().err
    rc, stats, *_ = _oracle.ref_ba_adjust(sc, 14, 6, 1, lib=_oracle.adapter_ba_emu())   # and the next call works
    assert stats[3] == 1.0


def _stats(lib, reset=0):
    import ctypes as C
    out = (C.c_uint64 * 2)()
    lib.mvgx_adapter_ba_context_stats(out, C.c_int(reset))
    return int(out[0]), int(out[1])


def test_consecutive_adjust_calls_keep_the_context(monkeypatch):
    """the replacement TU offers its arrays to the context the previous Adjust() left idle (mvgx_ba_update): the same scene again -
    with the same or with other Optimize_Options, as global_SfM.cpp:379-446 does - re-binds it; another scene replaces it;
    MVGX_BA_CONTEXT_CACHE=0 restores create / destroy per call. Results do not depend on which happened."""
    lib = _oracle.adapter_ba_emu()
    z = _golden()
    keys = ("poses", "intrinsics", "intr_model", "points", "obs_pose", "obs_intr", "obs_point", "obs_xy")

    def scene(tag):
        sc = {k: z[f"{tag}/{k}"].copy() for k in keys}
        sc["n_poses"] = len(sc["poses"]); sc["n_intrinsics"] = len(sc["intrinsics"]); sc["n_points"] = len(sc["points"]); sc["n_obs"] = len(sc["obs_pose"])
        return sc
    def same(x, y):   # (stats[2] is the wall time of the call)
        return x[0] == y[0] and np.array_equal(np.delete(x[1], 2), np.delete(y[1], 2)) and all(np.array_equal(u, v) for u, v in zip(x[2:], y[2:]))
    lib.mvgx_adapter_ba_release_context()
    _stats(lib, reset=1)
    a = scene("tiny_k3|14|6|1")
    r1 = _oracle.ref_ba_adjust(a, 14, 6, 1, lib=lib)
    assert _stats(lib) == (1, 0)
    r2 = _oracle.ref_ba_adjust(a, 14, 6, 1, lib=lib)                  # same scene, same options
    assert _stats(lib) == (1, 1)
    assert r1[1][3] == r2[1][3] == 1.0 and same(r1, r2)
    r3 = _oracle.ref_ba_adjust(a, 1, 4, 1, lib=lib)                   # same scene, translations + structure only
    assert _stats(lib) == (1, 2)
    b = scene("tiny_pinhole|14|6|1")
    r4 = _oracle.ref_ba_adjust(b, 14, 6, 1, lib=lib)                  # another scene: a new context
    assert _stats(lib) == (2, 2) and r4[1][3] == 1.0
    monkeypatch.setenv("MVGX_BA_CONTEXT_CACHE", "0")
    r5 = _oracle.ref_ba_adjust(a, 1, 4, 1, lib=lib)
    r6 = _oracle.ref_ba_adjust(a, 14, 6, 1, lib=lib)
    assert _stats(lib) == (4, 2)
    assert same(r3, r5), "re-bound context and new context disagree (other options)"
    assert same(r1, r6)
    monkeypatch.delenv("MVGX_BA_CONTEXT_CACHE")
    lib.mvgx_adapter_ba_release_context()
