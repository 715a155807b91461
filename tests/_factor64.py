"""Test helper: the 64 x 64 factor-and-invert kernel of the BA solver (chol_diag_inv_body in openmvg_amd/csrc/mvgx_ba.hip) through
the test hook mvgx_debug_factor64, checked against numpy. Shared by the emulation test (CPU) and the GPU test."""
import ctypes as C

import numpy as np


def factor64_errors(handle, kb, seed):
    """-> (relative error of L, of the k-major inverse, of the row-major inverse) for one random SPD kb x kb block"""
    f = handle.mvgx_debug_factor64
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(seed)
    m = rng.standard_normal((kb, 2 * kb))
    a = m @ m.T + kb * np.eye(kb)
    full = np.zeros((64, 64), order="F")
    full[:kb, :kb] = a
    l_out = np.zeros((64, 64), order="F")
    linv = np.zeros(8192)
    rc = f(full.ctypes.data, kb, l_out.ctypes.data, linv.ctypes.data)
    assert rc == 0, rc
    l_ref = np.linalg.cholesky(a)
    li_ref = np.linalg.inv(l_ref)
    l_got = np.tril(l_out[:kb, :kb])
    li_km = linv[:4096].reshape(64, 64).T[:kb, :kb]      # [k][c] = Linv[c][k]
    li_rm = linv[4096:].reshape(64, 64)[:kb, :kb]
    scale_l, scale_i = np.abs(l_ref).max(), np.abs(li_ref).max()
    return (np.abs(l_got - l_ref).max() / scale_l, np.abs(li_km - li_ref).max() / scale_i, np.abs(li_rm - li_ref).max() / scale_i)


def check_factor64(handle):
    for seed, kb in enumerate((64, 64, 64, 63, 49, 48, 37, 33, 32, 17, 16, 5, 1)):
        errs = factor64_errors(handle, kb, 100 + seed)
        assert max(errs) < 1e-12, (kb, errs)


def check_factor64_rejects_indefinite(handle):
    """a block that is not positive definite must be reported (fail flag -> MVGX_ERR_NUMERIC), whichever panel meets the bad pivot"""
    f = handle.mvgx_debug_factor64
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(9)
    for bad in (0, 17, 40, 63):
        m = rng.standard_normal((64, 128))
        a = m @ m.T + 64 * np.eye(64)
        a[bad, bad] = -1.0
        full = np.asfortranarray(a)
        l_out = np.zeros((64, 64), order="F")
        linv = np.zeros(8192)
        assert f(full.ctypes.data, 64, l_out.ctypes.data, linv.ctypes.data) == 6, bad   # MVGX_ERR_NUMERIC
