"""Build helper: compiles the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU).

  openmvg_amd/csrc/*.hip  ->  openmvg_amd/lib/libmvgx_hip.so
"""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libmvgx_hip.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not _stale(LIB, deps):
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


def build_oracle(ref=True, verbose=False):
    """Builds the CPU checkers (test infrastructure): the C restatement always, oracle/_ref only when the
    reference tree is present (i.e. in the build container)."""
    odir = os.path.join(ROOT, "oracle")
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", odir, "port"], check=True, stdout=out)
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-C", odir, "-j8", "ref"], check=True, stdout=out)


def build_adapter(verbose=False):
    """openMVG-side adapter TUs (openmvg_amd/adapter/*.cpp: link-time replacements of Matcher_Regions.cpp and
    sfm_data_BA_ceres.cpp) compiled against the openMVG tree when it is present -> openmvg_amd/lib/adapter_obj/*.o."""
    if not os.path.isdir("/root/reference/src"):
        return None
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", os.path.join(_HERE, "adapter")], check=True, stdout=out)
    return os.path.join(LIBDIR, "adapter_obj")
