"""The arithmetic that must stay un-fused on the device (DESIGN.md 3.7 "Contraction"): the residuals / normalisation / NFA sums of the
geometric-filter kernels and the projections of the cascade hashing stage reproduce a reference built without FMA, one rounded product
and one rounded sum at a time. The toolchain's __dmul_rn / __fadd_rn ... are plain operators that hipcc's default contraction fuses; the
library goes through OCML's rounded operations instead. These tests feed values on which fma(a, b, c) differs from round(round(a b) + c)
through the library's own helpers (test hooks) - a toolchain or flag change that fuses them again fails here, not in a rare inlier set."""
import ctypes as C

import numpy as np
import pytest

from openmvg_amd import _capi


@pytest.mark.gpu
def test_double_products_and_sums_are_not_contracted():
    lib = _capi.lib()
    lib.mvgx_debug_rounded_ops.restype = C.c_int
    a = 1.0 + 2.0 ** -30
    abc = np.array([a, a, -1.0])                       # a a = 1 + 2^-29 + 2^-60: the product rounds the last term away
    out = np.zeros(2)
    _capi.check(lib.mvgx_debug_rounded_ops(abc.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    assert out[0] == 2.0 ** -29 and out[1] == 2.0 ** -29 + 2.0 ** -60, out


@pytest.mark.gpu
def test_float_products_and_sums_of_the_hashing_stage_are_not_contracted():
    lib = _capi.lib()
    lib.mvgx_debug_rounded_ops_f32.restype = C.c_int
    a = np.float32(1.0 + 2.0 ** -12)
    abc = np.array([a, a, -1.0], np.float32)           # a a = 1 + 2^-11 + 2^-24: float32 keeps 2^-23
    out = np.zeros(2, np.float32)
    _capi.check(lib.mvgx_debug_rounded_ops_f32(abc.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    assert out[0] == np.float32(2.0 ** -11) and out[1] == np.float32(2.0 ** -11 + 2.0 ** -24), out
