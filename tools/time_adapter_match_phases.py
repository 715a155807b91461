"""Where the time of Matcher_Regions::Match goes at the boundary (replacement TU, 1 000 x 2 000 by default): the whole call, the call
without the container (MVGX_ADAPTER_DEBUG_SKIP=2: lists built, container untouched) and without lists (=1: device + transfers only).
Usage: time_adapter_match_phases.py [n_images] [repetitions]"""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    from openmvg_amd import synth
    from tests import _oracle
    n = int(sys.argv[2]); reps = int(sys.argv[3])
    descs = synth.image_descriptors(n, n_desc=2000, seed=0xC0FFEE00)
    arrs, ptrs, cnt = _oracle._desc_tables(descs)
    lib = C.CDLL(_oracle.ADAPTER_SO, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    lib.ref_matcher_regions_match_u8_timed.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_float, C.c_void_p]
    for r in range(reps):
        out = np.zeros(3)
        lib.ref_matcher_regions_match_u8_timed(ptrs, cnt, n, C.c_float(0.8), out.ctypes.data)
        print(f"RESULT rep {r} Match() {out[0]:.4f} s, {int(out[1])} matches in {int(out[2])} pairs, {n*(n-1)/2*4e6/out[0]:.3e} descriptor pairs/s", flush=True)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "1000"
reps = sys.argv[2] if len(sys.argv) > 2 else "3"
for label, skip in (("whole call", None), ("lists built, container untouched", "2"), ("device + transfers only", "1")):
    env = dict(os.environ, MVGX_ADAPTER_TIMING="1")
    if skip: env["MVGX_ADAPTER_DEBUG_SKIP"] = skip
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n, reps], env=env, capture_output=True, text=True)
    print(f"== {label}")
    for l in (p.stdout + p.stderr).splitlines():
        if "RESULT" in l or "mvgx Matcher_Regions" in l: print("  " + l)
