"""CPU: the closed-form Jacobians the HIP kernels use (openmvg_amd/csrc/ba_math.h, compiled for the host here) agree
with the oracle's forward-mode autodiff (the way Ceres differentiates the reference's functors)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from openmvg_amd import synth
from tests import _oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hostlib():
    src = os.path.join(ROOT, "tests", "native", "ba_math_host.cpp")
    hdr = os.path.join(ROOT, "openmvg_amd", "csrc", "ba_math.h")
    out = os.path.join(ROOT, "tests", "native", "_build", "libba_math_host.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(out):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.dirname(hdr), src, "-o", out], check=True)
    L = C.CDLL(out)
    L.host_eval_observation.argtypes = [C.c_int] + [C.c_void_p] * 8
    L.host_eval_residual.argtypes = [C.c_int] + [C.c_void_p] * 5
    L.host_eval_prior.argtypes = [C.c_void_p] * 5
    L.host_huber.argtypes = [C.c_double, C.c_double, C.c_void_p]
    L.host_invert_spd3.argtypes = [C.c_void_p, C.c_void_p]
    return L


def _cases(rng, n):
    for trial in range(n):
        model = (1, 2, 3, 4, 5, 7)[trial % 6]
        K = {1: 3, 2: 4, 3: 6, 4: 8, 5: 7, 7: 0}[model]
        intr = np.zeros(8); intr[:3] = [1000 + 30 * rng.normal(), 500 + 5 * rng.normal(), 500 + 5 * rng.normal()]
        intr[3:K] = 0.1 * rng.standard_normal(max(K - 3, 0))
        if model == 7:
            intr[:] = 0; intr[0] = 2000 + trial; intr[1] = 1000
        pose = np.concatenate([rng.standard_normal(3) * (3.0 if trial % 5 == 0 else 0.5), 0.2 * rng.standard_normal(3) + [0, 0, 2.5]])
        if trial % 7 == 0:
            pose[:3] = 1e-9 * rng.standard_normal(3)     # first-order branch of AngleAxisRotatePoint
        if trial % 11 == 0:
            pose[:3] = 0.0
        X = rng.uniform(-0.3, 0.3, 3)
        obs = rng.uniform(300, 700, 2)
        yield model, intr, pose, X, obs


def test_closed_form_jacobians_equal_autodiff(hostlib):
    rng = np.random.default_rng(1)
    worst = 0.0
    for model, intr, pose, X, obs in _cases(rng, 600):
        r0, Ji0, Jc0, Jp0 = _oracle.port_ba_eval_obs(model, intr, pose, X, obs)
        r = np.zeros(2); Ji = np.zeros((2, 8)); Jc = np.zeros((2, 6)); Jp = np.zeros((2, 3))
        hostlib.host_eval_observation(model, intr.ctypes.data, pose.ctypes.data, X.ctypes.data, obs.ctypes.data,
                                      r.ctypes.data, Ji.ctypes.data, Jc.ctypes.data, Jp.ctypes.data)
        for a, b in ((r, r0), (Ji, Ji0), (Jc, Jc0), (Jp, Jp0)):
            scale = max(1.0, np.abs(b).max())
            err = np.abs(a - b).max() / scale
            worst = max(worst, err)
            assert err < 1e-11, (model, pose, a, b)
        r2 = np.zeros(2)
        hostlib.host_eval_residual(model, intr.ctypes.data, pose.ctypes.data, X.ctypes.data, obs.ctypes.data, r2.ctypes.data)
        assert np.array_equal(r2, r)
    assert worst < 1e-11


def test_huber_and_spd3(hostlib):
    rho = np.zeros(3)
    hostlib.host_huber(16.0, 100.0, rho.ctypes.data)
    assert rho.tolist() == [100.0, 1.0, 0.0]
    hostlib.host_huber(16.0, 400.0, rho.ctypes.data)     # s > b = 256: rho = 2 a sqrt(s) - b
    assert np.allclose(rho, [2 * 16 * 20 - 256, 16 / 20, -(16 / 20) / 800])
    hostlib.host_huber(0.0, 400.0, rho.ctypes.data)      # no loss
    assert rho.tolist() == [400.0, 1.0, 0.0]
    rng = np.random.default_rng(2)
    for _ in range(50):
        B = rng.standard_normal((3, 3)); A = B @ B.T + 1e-3 * np.eye(3)
        v = np.array([A[0, 0], A[0, 1], A[0, 2], A[1, 1], A[1, 2], A[2, 2]]); inv = np.zeros(6)
        assert hostlib.host_invert_spd3(v.ctypes.data, inv.ctypes.data) == 1
        Ai = np.array([[inv[0], inv[1], inv[2]], [inv[1], inv[3], inv[4]], [inv[2], inv[4], inv[5]]])
        assert np.allclose(Ai @ A, np.eye(3), atol=1e-8)
    bad = np.array([1.0, 2.0, 0.0, 1.0, 0.0, 1.0]); inv = np.zeros(6)
    assert hostlib.host_invert_spd3(bad.ctypes.data, inv.ctypes.data) == 0


def test_pose_center_prior_closed_form_equals_autodiff(hostlib):
    """PoseCenterConstraintCostFunction (sfm_data_BA_ceres.cpp:44-80): closed form vs the oracle's Jets, both branches of
    AngleAxisRotatePoint"""
    rng = np.random.default_rng(5)
    for trial in range(200):
        pose = np.concatenate([rng.standard_normal(3) * (2.5 if trial % 4 == 0 else 0.4), rng.standard_normal(3)])
        if trial % 9 == 0:
            pose[:3] = 1e-9 * rng.standard_normal(3)
        if trial % 13 == 0:
            pose[:3] = 0.0
        center = rng.standard_normal(3); weight = rng.uniform(0.1, 3.0, 3)
        r0, J0 = _oracle.port_ba_eval_prior(pose, center, weight)
        r = np.zeros(3); J = np.zeros((3, 6))
        hostlib.host_eval_prior(pose.ctypes.data, center.ctypes.data, weight.ctypes.data, r.ctypes.data, J.ctypes.data)
        assert np.abs(r - r0).max() < 1e-12 and np.abs(J - J0).max() < 1e-11, (pose, J, J0)
        R = synth._rodrigues(pose[None, :3])[0]
        assert np.allclose(r, weight * (-R.T @ pose[3:] - center), atol=1e-12)   # C = -R^T t
