"""CPU emulation build of the BA solver's HIP source — TEST INFRASTRUCTURE ONLY (tests/native/hipemu).

The product (openmvg_amd) has no CPU path and never loads this library. The CPU test-suite uses it to execute the
*same* device code (kernels + host driver of openmvg_amd/csrc/mvgx_ba.hip) under a fiber-based emulation of the HIP
execution model, so that index arithmetic, LDS reductions, wave collectives and the MFMA tile maps are checked against
the oracle here, where no GPU exists; the `-m gpu` tests then run the real thing on the MI355X.
"""
import contextlib
import ctypes as C
import os
import subprocess

from openmvg_amd import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SRC = os.path.join(_HERE, "native", "hipemu")
_OUT = os.path.join(_HERE, "native", "_build", "libmvgx_ba_emu.so")
_CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build(force=False):
    deps = [os.path.join(_SRC, "hipemu.cpp"), os.path.join(_SRC, "hip", "hip_runtime.h"),
            os.path.join(_ROOT, "include", "mvgx.h")]
    csrc = os.path.join(_ROOT, "openmvg_amd", "csrc")
    deps += [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if not force and os.path.exists(_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(_OUT) for d in deps):
        return _OUT
    os.makedirs(os.path.dirname(_OUT), exist_ok=True)
    cxx = _CLANG if os.path.exists(_CLANG) else "clang++"
    subprocess.run([cxx, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wno-psabi",
                    "-Wl,-Bsymbolic",   # never bind to same-named (weak, inline) symbols of libmvgx_hip.so loaded earlier
                    "-I" + _SRC,
                    "-I" + os.path.join(_ROOT, "include"), "-I" + csrc, os.path.join(_SRC, "hipemu.cpp"), "-o", _OUT], check=True)
    return _OUT


_handle = None


def handle():
    global _handle
    if _handle is None:
        h = C.CDLL(build())
        for name, (restype, argtypes) in _capi.PROTOTYPES.items():
            if name.startswith("mvgx_match"):
                continue   # the matching kernels are not emulated (inline gfx950 asm)
            fn = getattr(h, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _handle = h
    return _handle


@contextlib.contextmanager
def emulated():
    """Routes openmvg_amd._capi to the emulation library inside the block (tests only)."""
    saved = _capi._lib
    _capi._lib = handle()
    try:
        yield
    finally:
        _capi._lib = saved
