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
    """One object per translation unit (compiled side by side, rebuilt only when the unit or a header changed), linked into ONE shared
    library. No relocatable device code: no kernel calls a device function of another unit."""
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    extra = os.environ.get("MVGX_HIPCC_FLAGS", "").split()   # measurement builds (-DMVGX_..._STAMPS): tools/ only
    objs = [os.path.join(objdir, os.path.basename(s_)[:-4] + ".o") for s_ in srcs]
    todo = [(s_, o) for s_, o in zip(srcs, objs) if force or extra or _stale(o, [s_] + hdrs)]
    if todo:
        from concurrent.futures import ThreadPoolExecutor

        def one(so):
            cmd = [hipcc_path()] + flags + extra + ["-c", so[0], "-o", so[1]]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1)) as ex:
            list(ex.map(one, todo))
    if todo or _stale(LIB, objs):
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
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
