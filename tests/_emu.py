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
_CLANG = "/opt/rocm/lib/llvm/bin/clang++"
# MVGX_EMU_SANITIZE=1 (tools/sanitize_cpu.sh): the emulation libraries built with AddressSanitizer + UndefinedBehaviorSanitizer into files of
# their own; the interpreter must then run with the ASan runtime preloaded. GPU sanitizers do not exist on this pool - the device source is
# checked for out-of-bounds and undefined behaviour here, on the CPU build of the same code.
_SAN = os.environ.get("MVGX_EMU_SANITIZE") == "1"
_SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize=vptr,function", "-shared-libsan", "-fno-omit-frame-pointer", "-g", "-O1"] if _SAN else ["-O2"]
_OUT = os.path.join(_HERE, "native", "_build", "libmvgx_ba_emu_san.so" if _SAN else "libmvgx_ba_emu.so")


def build(force=False):
    deps = [os.path.join(_SRC, "hipemu.cpp"), os.path.join(_SRC, "hip", "hip_runtime.h"),
            os.path.join(_ROOT, "include", "mvgx.h")]
    csrc = os.path.join(_ROOT, "openmvg_amd", "csrc")
    deps += [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if not force and os.path.exists(_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(_OUT) for d in deps):
        return _OUT
    os.makedirs(os.path.dirname(_OUT), exist_ok=True)
    cxx = _CLANG if os.path.exists(_CLANG) else "clang++"
    subprocess.run([cxx, "-x", "c++", "-std=c++17", *_SAN_FLAGS, "-fPIC", "-shared", "-Wno-psabi",
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
                continue   # the uint8 matching path lives in the second library (handle_match)
            fn = getattr(h, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _handle = h
    return _handle


class _Router:
    """one handle over the two emulation libraries: mvgx_match_* -> libmvgx_match_emu.so (built on first use), the rest ->
    libmvgx_ba_emu.so; mvgx_last_error follows the library of the last call"""

    def __init__(self):
        object.__setattr__(self, "_last", None)

    def __getattr__(self, name):
        if name == "mvgx_last_error":
            return getattr(self._last or handle(), name)
        lib = handle_match() if name.startswith("mvgx_match") else handle()
        object.__setattr__(self, "_last", lib)
        return getattr(lib, name)


@contextlib.contextmanager
def emulated():
    """Routes openmvg_amd._capi to the emulation libraries inside the block (tests only)."""
    saved = _capi._lib
    _capi._lib = _Router()
    try:
        yield
    finally:
        _capi._lib = saved


# ---------------------------------------------------------------------------------------------------------
# the matching path (second emulation library)
# ---------------------------------------------------------------------------------------------------------
_OUT_MATCH = os.path.join(_HERE, "native", "_build", "libmvgx_match_emu_san.so" if _SAN else "libmvgx_match_emu.so")
_GEN_MATCH = os.path.join(_HERE, "native", "_build", "mvgx_match_emu.hip")

_HOST_STAGING = """
// (hipemu) LDS-DMA staging as per-lane copies: same source / destination addresses as the gfx950 instructions
__device__ __forceinline__ void stage_window_glds_asm(char* buf, const int8_t* __restrict__ gtiles,
                                                      const int* __restrict__ grconst, int wave, int lane) {
  for (int i = 0; i < kWinTiles; ++i) {
    const int off = i * kTileBytes + wave * 1024;
    memcpy(buf + off + lane * 16, gtiles + off + lane * 16, 16);
  }
  memcpy(buf + kWinTiles * kTileBytes + wave * 256 + lane * 4, grconst + wave * 64 + lane, 4);
}
__device__ __forceinline__ void stage_window_glds(char* buf, const int8_t* __restrict__ gtiles,
                                                  const int* __restrict__ grconst, int wave, int lane) {
  stage_window_glds_asm(buf, gtiles, grconst, wave, lane);
}
constexpr int kHalfTiles = kWinTiles / 2;
constexpr int kHalfStageBytes = kHalfTiles * kTileBytes + kHalfTiles * kTileRows * 4;
__device__ __forceinline__ void stage_half_glds_asm(char* buf, const int8_t* __restrict__ gtiles, const int* __restrict__ grconst, int wave, int lane) {
  for (int i = 0; i < kHalfTiles; ++i) {
    const int off = i * kTileBytes + wave * 1024;
    memcpy(buf + off + lane * 16, gtiles + off + lane * 16, 16);
  }
  if (wave < 2) memcpy(buf + kHalfTiles * kTileBytes + wave * 256 + lane * 4, grconst + wave * 64 + lane, 4);
}

"""


def generate_match_source():
    import re
    src = open(os.path.join(_ROOT, "openmvg_amd", "csrc", "mvgx_match.hip")).read()
    a = src.index("// LDS-DMA pieces written as inline asm")
    b = src.index("// Through registers, split into issue")
    src = src[:a] + _HOST_STAGING + src[b:]
    src = re.sub(r"asm volatile\([^;]*\);", ";", src)
    src = re.sub(r"__attribute__\(\(address_space\(\d+\)\)\)", "", src)
    assert "asm volatile" not in src and "global_load_lds" not in src
    os.makedirs(os.path.dirname(_GEN_MATCH), exist_ok=True)
    if not os.path.exists(_GEN_MATCH) or open(_GEN_MATCH).read() != src:
        open(_GEN_MATCH, "w").write(src)
    return _GEN_MATCH


def build_match(force=False):
    gen = generate_match_source()
    deps = [gen, os.path.join(_SRC, "hipemu.cpp"), os.path.join(_SRC, "hipemu_match.cpp"), os.path.join(_SRC, "hip", "hip_runtime.h"),
            os.path.join(_ROOT, "include", "mvgx.h"), os.path.join(_ROOT, "openmvg_amd", "csrc", "mvgx_common.hip"),
            os.path.join(_ROOT, "openmvg_amd", "csrc", "mvgx_common.h")]
    if not force and os.path.exists(_OUT_MATCH) and all(os.path.getmtime(d) <= os.path.getmtime(_OUT_MATCH) for d in deps):
        return _OUT_MATCH
    cxx = _CLANG if os.path.exists(_CLANG) else "clang++"
    subprocess.run([cxx, "-x", "c++", "-std=c++17", *_SAN_FLAGS, "-fPIC", "-shared", "-Wno-psabi", "-Wl,-Bsymbolic", "-I" + _SRC,
                    "-I" + os.path.dirname(gen), "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_ROOT, "openmvg_amd", "csrc"),
                    os.path.join(_SRC, "hipemu_match.cpp"), "-o", _OUT_MATCH], check=True)
    return _OUT_MATCH


_handle_match = None


def handle_match():
    global _handle_match
    if _handle_match is None:
        h = C.CDLL(build_match())
        for name, (restype, argtypes) in _capi.PROTOTYPES.items():
            if name.startswith("mvgx_match") or name in ("mvgx_last_error", "mvgx_abi_version", "mvgx_device_count"):
                fn = getattr(h, name)
                fn.restype = restype
                fn.argtypes = argtypes
        _handle_match = h
    return _handle_match
