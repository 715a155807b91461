"""CPU: libmvgx_hip.so builds for gfx950, loads without a GPU, and exports every symbol include/mvgx.h declares."""
import ctypes as C
import os
import re

from openmvg_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "mvgx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mvgx_[a-z0-9_]+)\s*\(", txt)) - {"mvgx_match_sink", "mvgx_allreduce_f64"})


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    handle = C.CDLL(built_lib)
    syms = _header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in include/mvgx.h but not exported"
        assert s in _capi.PROTOTYPES, f"{s} has no ctypes prototype"
    assert set(_capi.PROTOTYPES) <= set(syms)


def test_library_loads_without_gpu_and_reports_version(built_lib):
    lib = _capi.lib()
    assert lib.mvgx_abi_version() == 12
    assert _capi.device_count() >= 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _capi.lib()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("missing extension must raise")


def test_every_built_shared_library_resolves_all_its_symbols():
    """dlopen(RTLD_NOW) of the product, the adapter and the reference checker libraries: a missing link dependency must
    show up here, not as a lazy-binding failure on the GPU box."""
    import ctypes
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = glob.glob(os.path.join(root, "openmvg_amd", "lib", "*.so")) + glob.glob(os.path.join(root, "oracle", "_ref", "*.so")) + \
        glob.glob(os.path.join(root, "oracle", "_build", "*.so")) + glob.glob(os.path.join(root, "tests", "native", "_build", "libmvgx_openmvg_adapter*.so"))
    assert any(p.endswith("libmvgx_hip.so") for p in libs)
    for p in sorted(libs):
        # (DEEPBIND: a library's own dependencies before whatever an earlier test put into the global scope - the *_emu adapter
        # libraries must bind to the emulation library they were linked with, not to a globally loaded libmvgx_hip.so)
        ctypes.CDLL(p, mode=os.RTLD_NOW | os.RTLD_DEEPBIND)
